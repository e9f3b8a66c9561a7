"""Generates tests/golden/kat_kpp_goldens.npz with the scikit-learn installed in the build container: what the reference's
cluster_colors (k-means-color-clustering/KmeanGrids.py:288-339, color_kmeans.py:54-135) computes for k = 3 on cells of
the reference's own recorded flow visualisation (tests/golden/kat_cells.npz, frame index 0) when its
KMeans(n_clusters=k) is given a seed:  KMeans(n_clusters=3, init='k-means++', n_init=1, random_state=SEED).fit(X) ->
predict -> bincount -> dominant cluster -> np.rint.  Only cells with at least 12 distinct rows are kept (fewer distinct
points than clusters make sklearn's own result thread-count dependent).  Only the vectors travel."""
import os
import sys
import warnings

import numpy as np
from sklearn.cluster import KMeans

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import oracle as O  # noqa: E402

SEED, K = 0, 3


def main():
    kat = np.load(os.path.join(HERE, "kat_cells.npz"))
    cells = kat["cells_rgb"][0]
    idx, cen, dom, nit = [], [], [], []
    for c in range(cells.shape[0]):
        X = O.preprocess_rgba(cells[c]).reshape(-1, 4)
        if len(np.unique(X, axis=0)) < 12:
            continue
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            km = KMeans(n_clusters=K, init="k-means++", n_init=1, random_state=SEED).fit(X.astype(np.float64))
        counts = np.bincount(km.predict(X.astype(np.float64)), minlength=K)
        order = sorted(range(K), key=lambda i: counts[i] / len(X), reverse=True)          # stable, as the reference
        idx.append(c)
        cen.append(km.cluster_centers_)
        dom.append(np.rint(km.cluster_centers_[order[0]]))
        nit.append(km.n_iter_)
        if len(idx) == 40:
            break
    np.savez_compressed(os.path.join(HERE, "kat_kpp_goldens.npz"), cell_index=np.array(idx, np.int32),
                        centers=np.array(cen), dominant_rint=np.array(dom), n_iter=np.array(nit, np.int32),
                        seed=np.int32(SEED), k=np.int32(K))
    print("cells", idx, "n_iter", nit)


if __name__ == "__main__":
    main()
