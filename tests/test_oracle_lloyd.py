"""The Lloyd oracle (oracle/lloyd_ref.c) against vectors produced by the installed scikit-learn
(tests/golden/make_lloyd_goldens.py) -- this is what pins it."""
import os

import numpy as np
import pytest

from oracle import oracle as O

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "lloyd_goldens.npz"))
CASES = sorted({k.split("/")[0] for k in Z.files if "/" in k})


@pytest.mark.parametrize("name", CASES)
def test_fit_matches_sklearn(name):
    X, C0 = Z[name + "/X"], Z[name + "/C0"]
    cen, lab, inertia, n_iter = O.kmeans_fit(X, C0, int(Z[name + "/max_iter"]), float(Z[name + "/tol"]))
    assert n_iter == int(Z[name + "/n_iter"])
    assert np.array_equal(lab, Z[name + "/labels"])            # bit-exact labels
    assert np.abs(cen - Z[name + "/centers"]).max() <= 1e-11   # sklearn's own thread-order noise is 6e-14
    assert abs(inertia - float(Z[name + "/inertia"])) <= 1e-12 * float(Z[name + "/inertia"])


@pytest.mark.parametrize("name", CASES)
def test_predict_matches_sklearn(name):
    X = Z[name + "/X"]
    assert np.array_equal(O.kmeans_predict(X, Z[name + "/centers"]), Z[name + "/predict"])


def test_too_few_samples_raises():
    with pytest.raises(ValueError):
        O.kmeans_fit(np.zeros((2, 4), np.uint8), np.zeros((3, 4)))


def test_partials_sum_to_full_step():
    """two shards' partials add up to the one-shard partials (exact: integer-valued data)"""
    X = Z["cell_u8_k3/X"]
    C0 = Z["cell_u8_k3/C0"]
    mean = np.zeros(4)
    full = O.lloyd_partials(X, mean, C0, np.full(len(X), -1, np.int32))
    a = O.lloyd_partials(X[:1000], mean, C0, np.full(1000, -1, np.int32))
    b = O.lloyd_partials(X[1000:], mean, C0, np.full(len(X) - 1000, -1, np.int32))
    assert np.array_equal(a + b, full)


KPP = np.load(os.path.join(os.path.dirname(__file__), "golden", "kpp_goldens.npz"))
KPP_CASES = sorted({k.split("/")[0] for k in KPP.files})


@pytest.mark.parametrize("name", KPP_CASES)
def test_kmeans_plusplus_host_logic_reproduces_sklearn_seeds(name):
    """the numpy-RandomState draw order, cumsum/searchsorted and greedy candidate choice of cluster.kmeans_plusplus,
    with the oracle standing in for the device step, pick the rows sklearn's _kmeans_plusplus picks"""
    from opticalflowclustering_amd.cluster import kmeans_plusplus
    X = KPP[name + "/X"]
    centers, idx = kmeans_plusplus(X, int(KPP[name + "/k"]), int(KPP[name + "/seed"]), _step=O.kpp_candidates)
    assert np.array_equal(idx, KPP[name + "/indices"])
    assert np.array_equal(centers, X[idx].astype(np.float64))


KATKPP = np.load(os.path.join(os.path.dirname(__file__), "golden", "kat_kpp_goldens.npz"))


def test_seeded_cluster_colors_on_reference_cells_matches_sklearn():
    """k = 3 on cells of the reference's recorded visualisation, KMeans(n_clusters=3, random_state=0) as the reference
    constructs it plus a seed (make_kat_kpp_goldens.py): k-means++ host logic + Lloyd oracle land on sklearn's centres"""
    from opticalflowclustering_amd.cluster import kmeans_plusplus
    kat = np.load(os.path.join(os.path.dirname(__file__), "golden", "kat_cells.npz"))
    k, seed = int(KATKPP["k"]), int(KATKPP["seed"])
    for j, c in enumerate(KATKPP["cell_index"]):
        X = O.preprocess_rgba(kat["cells_rgb"][0][c]).reshape(-1, 4)
        C0, _ = kmeans_plusplus(X, k, seed, _step=O.kpp_candidates)
        cen, lab, _, n_iter = O.kmeans_fit(X, C0)
        assert n_iter == int(KATKPP["n_iter"][j]), c
        assert np.abs(cen - KATKPP["centers"][j]).max() <= 1e-9, c
        counts = np.bincount(O.kmeans_predict(X, cen), minlength=k)
        assert np.array_equal(np.rint(cen[int(np.argmax(counts))]), KATKPP["dominant_rint"][j]), c
