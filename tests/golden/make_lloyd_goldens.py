"""Generates tests/golden/lloyd_goldens.npz with the scikit-learn installed in the build container
(the third-party module that holds the reference's k-means arithmetic: KMeans(...).fit/.predict
at k-means-color-clustering/color_kmeans.py:66-78, KmeanGrids.py:300-304).  Only the resulting
vectors travel; sklearn is never imported on the GPU box.

Each case: X (as the caller holds it: u8 / f32 / f64), the fixed init C0, and sklearn's
cluster_centers_, labels_, predict(X), inertia_, n_iter_ for
KMeans(n_clusters=k, init=C0, n_init=1, max_iter=300, tol=1e-4).fit(X.astype(float64)).
"""
import os
import warnings

import numpy as np
import sklearn
from sklearn.cluster import KMeans

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lloyd_goldens.npz")


def run(name, X, C0, store, max_iter=300, tol=1e-4):
    Xd = X.astype(np.float64)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        km = KMeans(n_clusters=C0.shape[0], init=C0.astype(np.float64), n_init=1,
                    max_iter=max_iter, tol=tol).fit(Xd)
        pred = km.predict(Xd)
    store[f"{name}/X"] = X
    store[f"{name}/C0"] = C0.astype(np.float64)
    store[f"{name}/centers"] = km.cluster_centers_
    store[f"{name}/labels"] = km.labels_.astype(np.int32)
    store[f"{name}/predict"] = pred.astype(np.int32)
    store[f"{name}/inertia"] = np.float64(km.inertia_)
    store[f"{name}/n_iter"] = np.int32(km.n_iter_)
    store[f"{name}/max_iter"] = np.int32(max_iter)
    store[f"{name}/tol"] = np.float64(tol)
    print(name, X.shape, X.dtype, "k", C0.shape[0], "n_iter", km.n_iter_, "inertia", km.inertia_,
          "counts", np.bincount(km.labels_, minlength=C0.shape[0]))


def flow_vis_like(rng, n, frac=0.08):
    """RGBA rows shaped like a thresholded flow-visualisation cell: mostly black, some colour"""
    X = np.zeros((n, 4), np.uint8)
    m = rng.random(n) < frac
    hue = rng.integers(0, 3, m.sum())
    v = rng.integers(30, 256, m.sum())
    col = np.zeros((m.sum(), 3), np.uint8)
    col[np.arange(m.sum()), hue] = v
    col[np.arange(m.sum()), (hue + 1) % 3] = (v * rng.random(m.sum())).astype(np.uint8)
    col[col < 30] = 0
    X[m, :3] = col
    X[m, 3] = 255
    return X


def main():
    rng = np.random.default_rng(0)
    S = {}
    # cfg0: one 256x256 RGB frame -> preprocess-like RGBA rows, k=3
    img = rng.integers(0, 256, (256, 256, 3), dtype=np.uint8)
    img[img < 30] = 0
    X = np.concatenate([img.reshape(-1, 3), np.full((65536, 1), 255, np.uint8)], 1)
    X[(img.reshape(-1, 3) == 0).all(1), 3] = 0
    run("cfg0_u8_k3", X, X[rng.choice(65536, 3, replace=False)], S)
    # grid-cell shapes (51x51 = 2601 points), k = 1, 3, 8
    Xc = flow_vis_like(rng, 2601)
    for k in (1, 3, 8):
        uniq = np.unique(Xc, axis=0)
        C0 = uniq[rng.choice(len(uniq), k, replace=False)]
        run(f"cell_u8_k{k}", Xc, C0, S)
    # (u,v) vectors, f32, d=2, k=5: five motion populations + noise
    vel = rng.uniform(-4, 4, (5, 2))
    lab = rng.integers(0, 5, 65536)
    U = (vel[lab] + 0.35 * rng.standard_normal((65536, 2))).astype(np.float32)
    run("uv_f32_k5", U, U[rng.choice(65536, 5, replace=False)], S)
    run("uv_f32_k5_small", U[:2601], U[rng.choice(2601, 5, replace=False)], S)
    # stops by tol rather than strict convergence: one broad blob, k=3, loose tol
    B = rng.standard_normal((65536, 2)).astype(np.float32)
    run("blob_f32_k3_tol", B, B[:3], S, tol=1e-2)
    # max_iter exhausted
    run("blob_f32_k5_maxiter", B[:20000], B[:5], S, max_iter=4, tol=0.0)
    # empty-cluster relocation: one init centre far away from all data
    C0 = np.array([[0.0, 0.0], [1.0, 1.0], [500.0, 500.0]])
    run("reloc_f64_k3", B[:4096].astype(np.float64), C0, S)
    # several empty clusters in one iteration (sklearn hands far points to empty clusters in argpartition order; the C
    # oracle, lloyd_api.cpp and sharded.py go farthest-first: these pin that the two agree): 2 and 3 far-away centres
    C0 = np.array([[0.0, 0.0], [1.0, 1.0], [500.0, 500.0], [-400.0, 300.0]])
    run("reloc2_f64_k4", B[:4096].astype(np.float64), C0, S)
    C0 = np.array([[0.5, -0.5], [-1.0, 0.3], [600.0, 0.0], [0.0, -700.0], [-300.0, -300.0], [1.5, 1.5]])
    run("reloc3_f32_k6", B[4096:12288], C0, S)
    # f64 data, d=4, k=8
    D = rng.standard_normal((8192, 4)) * np.array([1, 2, 3, 4.0])
    run("gauss_f64_d4_k8", D, D[:8], S)
    S["sklearn_version"] = np.array(sklearn.__version__)
    np.savez_compressed(OUT, **S)
    print(OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
