"""KMeans -- the subset of sklearn.cluster.KMeans the reference's hot path uses
(color_kmeans.py:66-78, KmeanGrids.py:300-304: KMeans(n_clusters=k) -> .fit(X) -> .cluster_centers_,
.predict(X)), running Lloyd on the MI355X through libofc.

Semantics follow sklearn (cast to float64 math, centring, tol = 1e-4 * mean(var), strict-then-tol
convergence, final E-step, empty-cluster relocation).  One deliberate difference: the reference
constructs KMeans(n_clusters=k) with sklearn's default init='k-means++', random_state=None, i.e. a
non-deterministic seeding.  Here `init` is an explicit (k, d) array, 'seeded-rows'
(the default: k distinct rows of X picked by numpy's default_rng(random_state), seed 0 -- reproducible), or
'k-means++': sklearn's own seeding (_kmeans.py:174-272) with numpy-RandomState-compatible draws, so that
KMeans(n_clusters=k, init='k-means++', random_state=s) lands on the centres sklearn finds for the same seed
(random_state=None then means numpy's global RandomState, as in sklearn).  For the reference's documented k=1
the result does not depend on the seeding at all."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, load, ptr

_DT = {np.dtype(np.uint8): _lib.U8, np.dtype(np.float32): _lib.F32, np.dtype(np.float64): _lib.F64}


def _as_supported(X):
    X = np.asarray(X)
    if X.ndim != 2:
        raise ValueError(f"Expected 2D array, got {X.ndim}D array instead")
    if X.dtype not in _DT:
        X = X.astype(np.float64)          # sklearn: everything that is not f32/f64 becomes f64
    return np.ascontiguousarray(X)


def seeded_rows_init(X, k, random_state=0):
    """k distinct rows (by value, when there are enough distinct rows) picked by a seeded rng"""
    rng = np.random.default_rng(random_state)
    X = np.asarray(X)
    uniq = np.unique(X, axis=0)
    if len(uniq) >= k:
        return uniq[rng.choice(len(uniq), k, replace=False)].astype(np.float64)
    return X[rng.choice(len(X), k, replace=len(X) < k)].astype(np.float64)


def check_random_state(seed):
    """sklearn.utils.check_random_state: None -> numpy's global RandomState, int -> RandomState(seed)"""
    if seed is None or seed is np.random:
        return np.random.mtrand._rand
    if isinstance(seed, (int, np.integer)):
        return np.random.RandomState(int(seed))
    if isinstance(seed, np.random.RandomState):
        return seed
    raise ValueError(f"{seed!r} cannot be used to seed a numpy.random.RandomState instance")


def kmeans_plusplus(X, n_clusters, random_state=None, n_local_trials=None, device=0, _step=None):
    """sklearn's _kmeans_plusplus (_kmeans.py:174-272) as KMeans.fit runs it: on the column-centred data, unit sample
    weights.  The O(N d trials) part of every step -- distances of all samples to the candidate rows, the minimum
    with the running closest distance, the candidates' potentials -- is one libofc launch (ofc_kpp_candidates); the
    RandomState draws, np.cumsum and searchsorted are numpy's, in sklearn's order.  -> (centres (k,d) f64 rows of X,
    indices).  The distances are not bit-identical to BLAS's (different summation order inside the dot product), so
    agreement with sklearn is exact unless a random value falls within ~1e-16 (relative) of a cumulative-sum
    boundary; tests/golden/kpp_goldens.npz pins it for a spread of seeds and shapes."""
    X = _as_supported(X)
    N, d = X.shape
    rs = check_random_state(random_state)
    if n_local_trials is None:
        n_local_trials = 2 + int(np.log(n_clusters))                              # :217-221
    if n_local_trials > 8:
        raise ValueError("n_local_trials > 8 is not supported")
    mean = X.astype(np.float64).mean(axis=0) if X.dtype != np.float32 else X.mean(axis=0).astype(np.float64)
    weight = np.ones(N, np.float64)
    indices = np.full(n_clusters, -1, dtype=np.int64)
    indices[0] = rs.choice(N, p=weight / weight.sum())                            # :224

    def step(cand, closest):
        cand = np.ascontiguousarray(cand, np.int64)
        if _step is not None:                    # tests: the CPU oracle stands in for the device step
            return _step(X, mean, cand, closest)
        out = np.empty((len(cand), N), np.float64)
        pots = np.empty(len(cand), np.float64)
        check(load().ofc_kpp_candidates(device, ptr(X), _DT[X.dtype], N, d, ptr(mean), ptr(cand), len(cand),
                                        ptr(closest) if closest is not None else None, ptr(out), ptr(pots)))
        return out, pots

    out, pots = step(indices[:1], None)                                           # :233-236
    closest, current_pot = out[0], pots[0]
    for c in range(1, n_clusters):
        rand_vals = rs.uniform(size=n_local_trials) * current_pot                  # :242
        candidate_ids = np.searchsorted(np.cumsum(closest, dtype=np.float64), rand_vals)   # :243-245
        np.clip(candidate_ids, None, N - 1, out=candidate_ids)                    # :247
        out, pots = step(candidate_ids, closest)                                  # :250-256
        best = int(np.argmin(pots))                                               # :259
        current_pot, closest = pots[best], out[best]
        indices[c] = candidate_ids[best]
    return X[indices].astype(np.float64), indices


class KMeans:
    def __init__(self, n_clusters=8, *, init="seeded-rows", n_init=1, max_iter=300, tol=1e-4,
                 random_state=0, device=0, **_ignored):
        self.n_clusters, self.init, self.n_init = int(n_clusters), init, n_init
        self.max_iter, self.tol, self.random_state, self.device = int(max_iter), float(tol), random_state, device

    def _init_centers(self, X):
        if isinstance(self.init, str):
            if self.init == "k-means++":
                return kmeans_plusplus(X, self.n_clusters, self.random_state, device=self.device)[0]
            if self.init not in ("seeded-rows", "random"):
                raise ValueError(f"init should be an array, 'k-means++' or 'seeded-rows', got {self.init!r}")
            return seeded_rows_init(X, self.n_clusters, self.random_state if self.random_state is not None else 0)
        C0 = np.ascontiguousarray(self.init, np.float64)
        if C0.shape != (self.n_clusters, X.shape[1]):
            raise ValueError(f"The shape of the initial centers {C0.shape} does not match "
                             f"({self.n_clusters}, {X.shape[1]})")
        return C0

    def fit(self, X, y=None, sample_weight=None):
        if sample_weight is not None:
            raise ValueError("sample_weight is not supported (the reference never passes it)")
        X = _as_supported(X)
        N, d = X.shape
        k = self.n_clusters
        if N < k:
            raise ValueError(f"n_samples={N} should be >= n_clusters={k}.")
        C0 = self._init_centers(X)
        centers = np.empty((k, d), np.float64)
        labels = np.empty(N, np.int32)
        inertia, n_iter = C.c_double(), C.c_int()
        check(load().ofc_kmeans_fit(self.device, ptr(X), _DT[X.dtype], N, d, k, ptr(C0), self.max_iter,
                                    self.tol, ptr(centers), ptr(labels), C.byref(inertia), C.byref(n_iter)))
        self.cluster_centers_, self.labels_ = centers, labels
        self.inertia_, self.n_iter_ = inertia.value, n_iter.value
        self.n_features_in_ = d
        return self

    def predict(self, X):
        X = _as_supported(X)
        N, d = X.shape
        if d != self.cluster_centers_.shape[1]:
            raise ValueError(f"X has {d} features, but KMeans is expecting {self.cluster_centers_.shape[1]}")
        labels = np.empty(N, np.int32)
        cen = np.ascontiguousarray(self.cluster_centers_, np.float64)
        check(load().ofc_kmeans_predict(self.device, ptr(X), _DT[X.dtype], N, d, self.n_clusters, ptr(cen), ptr(labels)))
        return labels

    def fit_predict(self, X, y=None):
        return self.fit(X).labels_


def kmeans_fit_dev(X_ptr, dtype, N, d, init, max_iter=300, tol=1e-4, labels_ptr=None, device=0, colsum=None):
    """device-resident X (this rank's shard when a communicator is active).  colsum: this rank's column sums when the
    caller already has them (the flow engine emits sum(u), sum(v) with the field): the fit then skips that sweep.
    -> centers (k,d), inertia, n_iter"""
    C0 = np.ascontiguousarray(init, np.float64)
    k = C0.shape[0]
    centers = np.empty((k, d), np.float64)
    inertia, n_iter = C.c_double(), C.c_int()
    cs = np.ascontiguousarray(colsum, np.float64) if colsum is not None else None
    if cs is not None and cs.shape != (d,):
        raise ValueError(f"colsum must have shape ({d},)")
    check(load().ofc_kmeans_fit_dev_stats(device, C.c_void_p(X_ptr), dtype, N, d, k, ptr(C0), max_iter, tol, ptr(cs), ptr(centers),
                                          C.c_void_p(labels_ptr) if labels_ptr else None, C.byref(inertia), C.byref(n_iter)))
    return centers, inertia.value, n_iter.value


def prune_stats(device=0):
    """how the last kmeans_fit_dev on `device` swept its samples (ofc_lloyd_prune_stats): tile sweeps, how many of them
    pruned, and the share of tiles the pruned sweeps did not have to read"""
    out = np.zeros(6, np.float64)
    check(load().ofc_lloyd_prune_stats(device, ptr(out)))
    return {"tile_sweeps": int(out[0]), "pruned_sweeps": int(out[1]), "probe_sweeps": int(out[4]), "final_pruned": bool(out[5]),
            "skip_fraction": float(out[3] / out[2]) if out[2] > 0 else 0.0}
