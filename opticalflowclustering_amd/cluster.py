"""KMeans -- the subset of sklearn.cluster.KMeans the reference's hot path uses
(color_kmeans.py:66-78, KmeanGrids.py:300-304: KMeans(n_clusters=k) -> .fit(X) -> .cluster_centers_,
.predict(X)), running Lloyd on the MI355X through libofc.

Semantics follow sklearn (cast to float64 math, centring, tol = 1e-4 * mean(var), strict-then-tol
convergence, final E-step, empty-cluster relocation).  One deliberate difference: the reference
constructs KMeans(n_clusters=k) with sklearn's default init='k-means++', random_state=None, i.e. a
non-deterministic seeding.  Here `init` is either an explicit (k, d) array or
'seeded-rows' (k distinct rows of X picked by numpy's default_rng(random_state), default seed 0),
which keeps runs reproducible; for the reference's documented k=1 the result does not depend on
the seeding at all."""
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


class KMeans:
    def __init__(self, n_clusters=8, *, init="seeded-rows", n_init=1, max_iter=300, tol=1e-4,
                 random_state=0, device=0, **_ignored):
        self.n_clusters, self.init, self.n_init = int(n_clusters), init, n_init
        self.max_iter, self.tol, self.random_state, self.device = int(max_iter), float(tol), random_state, device

    def _init_centers(self, X):
        if isinstance(self.init, str):
            if self.init not in ("seeded-rows", "k-means++", "random"):
                raise ValueError(f"init should be an array or 'seeded-rows', got {self.init!r}")
            return seeded_rows_init(X, self.n_clusters, self.random_state)
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


def kmeans_fit_dev(X_ptr, dtype, N, d, init, max_iter=300, tol=1e-4, labels_ptr=None, device=0):
    """device-resident X (this rank's shard when a communicator is active).
    -> centers (k,d), inertia, n_iter"""
    C0 = np.ascontiguousarray(init, np.float64)
    k = C0.shape[0]
    centers = np.empty((k, d), np.float64)
    inertia, n_iter = C.c_double(), C.c_int()
    check(load().ofc_kmeans_fit_dev(device, C.c_void_p(X_ptr), dtype, N, d, k, ptr(C0), max_iter, tol, ptr(centers),
                                    C.c_void_p(labels_ptr) if labels_ptr else None, C.byref(inertia), C.byref(n_iter)))
    return centers, inertia.value, n_iter.value
