"""Generates tests/golden/kpp_goldens.npz with the scikit-learn installed in the build container: what
KMeans(n_clusters=k, init='k-means++', n_init=1, random_state=seed).fit(X) -- the construction the reference uses at
k-means-color-clustering/color_kmeans.py:66 and KmeanGrids.py:300, plus a seed -- picks as seeds and converges to.
The seeding indices come from sklearn's own _kmeans_plusplus on the column-centred data with RandomState(seed), exactly
as KMeans.fit calls it (_kmeans.py:1478-1510).  Only the vectors travel; sklearn is never imported on the GPU box."""
import os
import warnings

import numpy as np
from sklearn.cluster import KMeans
from sklearn.cluster._kmeans import _kmeans_plusplus
from sklearn.utils.extmath import row_norms

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kpp_goldens.npz")


def case(name, X, k, seed, S):
    Xd = X.astype(np.float64)
    Xc = Xd - Xd.mean(axis=0)
    _, idx = _kmeans_plusplus(Xc, k, row_norms(Xc, squared=True), np.ones(len(Xc)), np.random.RandomState(seed))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        km = KMeans(n_clusters=k, init="k-means++", n_init=1, random_state=seed).fit(Xd)
    S[f"{name}/X"] = X
    S[f"{name}/k"] = np.int32(k)
    S[f"{name}/seed"] = np.int32(seed)
    S[f"{name}/indices"] = idx.astype(np.int64)
    S[f"{name}/centers"] = km.cluster_centers_
    S[f"{name}/labels"] = km.labels_.astype(np.int32)
    S[f"{name}/n_iter"] = np.int32(km.n_iter_)
    print(name, X.shape, X.dtype, "k", k, "seed", seed, "indices", idx, "n_iter", km.n_iter_)


def main():
    rng = np.random.default_rng(5)
    S = {}
    # cell-sized RGBA rows (the reference's shape: thresholded flow visualisation, mostly black)
    cell = np.zeros((2601, 4), np.uint8)
    m = rng.random(2601) < 0.3
    cell[m, :3] = rng.integers(30, 256, (m.sum(), 3))
    cell[m, 3] = 255
    for seed in (0, 1, 7, 42):
        case(f"cell_k3_s{seed}", cell, 3, seed, S)
    case("cell_k8_s3", cell, 8, 3, S)
    # blobs in 2-D (flow-vector like), f64
    cen = np.array([[-3, -3], [-1.5, 1], [0, 0], [1.5, -1], [3, 3]], np.float64)
    blob = (cen[rng.integers(0, 5, 6000)] + 0.35 * rng.standard_normal((6000, 2)))
    for seed in (0, 11):
        case(f"blob_k5_s{seed}", blob, 5, seed, S)
    # random u8 image rows (cfg0 shape, smaller)
    img = rng.integers(0, 256, (4096, 4), dtype=np.uint8)
    case("img_k3_s2", img, 3, 2, S)
    np.savez_compressed(OUT, **S)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
