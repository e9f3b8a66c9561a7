"""Host-driven sharded Lloyd: the same control flow as libofc's in-library driver (lloyd_api.cpp ==
sklearn's _kmeans_single_lloyd, _kmeans.py:624-752), but with the per-iteration exchange handed to a
caller-supplied collective, so that shards can be combined by ANY transport -- torch.distributed with
gloo across nodes, or RCCL.  Every rank holds a contiguous shard of the rows (frame-sharded (u,v) vectors)
and all k centres; per iteration ONE all-reduce (sum) of [k*d sums | k counts | n_changed].

`backend` supplies the four shard-local passes; DeviceShard runs them on the MI355X through the C ABI.
(The CPU tests drive the identical function with an oracle-backed shard and a gloo all-reduce.)"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, load, ptr


def _np_sum_small(a):
    a = [float(v) for v in a]
    n = len(a)
    if n < 8:
        r = 0.0
        for v in a:
            r += v
        return r
    r = a[:8]
    i = 8
    while i < n - (n % 8):
        for j in range(8):
            r[j] += a[i + j]
        i += 8
    res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
    for v in a[i:]:
        res += v
    return res


class DeviceShard:
    """a device-resident shard X_dev (N x d of dtype) + its u8 label buffer"""

    def __init__(self, X_ptr, dtype, N, d, labels_ptr=None, device=0):
        self.X, self.dtype, self.N, self.d, self.device = X_ptr, dtype, int(N), int(d), device
        self._own = None
        if labels_ptr is None:
            self._own = _lib.DeviceBuffer(max(self.N, 1), device)
            labels_ptr = self._own.ptr
        self.labels = labels_ptr
        check(load().ofc_memset(device, C.c_void_p(self.labels), 0xFF, self.N))

    def colstats(self, mean, pass_):
        out = np.zeros(self.d, np.float64)
        m = np.ascontiguousarray(mean, np.float64) if mean is not None else None
        check(load().ofc_lloyd_colstats_dev(self.device, C.c_void_p(self.X), self.dtype, self.N, self.d, ptr(m), pass_, ptr(out)))
        return out

    def step(self, mean, centers_c, accumulate=True):
        k = len(centers_c)
        rec = np.zeros(k * self.d + k + 1, np.float64)
        check(load().ofc_lloyd_step_dev(self.device, C.c_void_p(self.X), self.dtype, self.N, self.d, k,
                                        ptr(np.ascontiguousarray(mean, np.float64)), ptr(np.ascontiguousarray(centers_c, np.float64)),
                                        C.c_void_p(self.labels), 1 if accumulate else 0, ptr(rec)))
        return rec

    def inertia(self, mean, centers_c):
        v = C.c_double()
        check(load().ofc_lloyd_inertia_dev(self.device, C.c_void_p(self.X), self.dtype, self.N, self.d, len(centers_c),
                                           ptr(np.ascontiguousarray(mean, np.float64)), ptr(np.ascontiguousarray(centers_c, np.float64)),
                                           C.c_void_p(self.labels), C.byref(v)))
        return v.value

    def farthest(self, mean, centers_c, excl):
        ex = np.ascontiguousarray(excl, np.int64) if len(excl) else None
        d2, idx, lab = C.c_double(), C.c_int64(), C.c_int()
        xc = np.zeros(self.d, np.float64)
        check(load().ofc_lloyd_farthest_dev(self.device, C.c_void_p(self.X), self.dtype, self.N, self.d, len(centers_c),
                                            ptr(np.ascontiguousarray(mean, np.float64)), ptr(np.ascontiguousarray(centers_c, np.float64)),
                                            C.c_void_p(self.labels), ptr(ex), len(excl), C.byref(d2), C.byref(idx), ptr(xc), C.byref(lab)))
        return d2.value, idx.value, xc, lab.value


def fit_sharded(shard, init, max_iter=300, tol=1e-4, allreduce=None, rank=0):
    """-> (cluster_centers_ (k,d), inertia_, n_iter_); labels stay in the shard.
    allreduce(np.float64 array, op) with op in {'sum','max','min'} returns the reduced array."""
    if allreduce is None:
        allreduce = lambda a, op: a                                        # noqa: E731  (single shard)
    init = np.ascontiguousarray(init, np.float64)
    k, d = init.shape
    s = allreduce(np.concatenate([shard.colstats(None, 0), [float(shard.N)]]), "sum")
    Ng = s[d]
    if Ng < k:
        raise ValueError(f"n_samples={int(Ng)} should be >= n_clusters={k}.")
    mean = s[:d] / Ng                                                      # X.mean(axis=0)
    var = allreduce(shard.colstats(mean, 1), "sum") / Ng
    tol_abs = 0.0 if tol == 0 else _np_sum_small(var) / d * tol            # _tolerance
    c = init - mean
    strict, it = False, 0
    for it in range(max_iter):
        rec = allreduce(shard.step(mean, c, True), "sum")
        sums, w, n_changed = rec[:k * d].reshape(k, d).copy(), rec[k * d:k * d + k].copy(), rec[k * d + k]
        if (w == 0).any():                                                 # _relocate_empty_clusters_dense
            excl, first = [], True
            for j in range(k):
                if w[j] != 0:
                    continue
                d2, idx, xc, lab = shard.farthest(mean, c, excl)
                g = allreduce(np.array([d2]), "max")[0]
                cand = float(rank) if (idx >= 0 and d2 == g) else 1e300
                owner = allreduce(np.array([cand]), "min")[0] == float(rank) and idx >= 0 and d2 == g
                if first and not g > 0:
                    break
                first = False
                msg = np.zeros(d + 1)
                if owner:
                    msg[:d], msg[d] = xc, float(lab)
                    excl.append(idx)
                msg = allreduce(msg, "sum")
                old = int(msg[d])
                sums[old] -= msg[:d]
                sums[j] = msg[:d]
                w[j] = 1.0
                w[old] -= 1.0
        amax = int(np.argmax(w))                                           # _average_centers (in-place quirk kept)
        cnew = sums.copy()
        for j in range(k):
            if w[j] > 0:
                cnew[j] = cnew[j] * (1.0 / w[j])
            else:
                cnew[j] = cnew[amax]
        sh2 = []
        for j in range(k):                                                 # _center_shift, 4-way grouped
            a, b, r, f = cnew[j], c[j], 0.0, 0
            while f + 4 <= d:
                r += ((a[f] - b[f]) ** 2 + (a[f + 1] - b[f + 1]) ** 2 + (a[f + 2] - b[f + 2]) ** 2 + (a[f + 3] - b[f + 3]) ** 2)
                f += 4
            while f < d:
                r += (a[f] - b[f]) ** 2
                f += 1
            sh2.append(np.sqrt(r) ** 2)
        c = cnew
        if n_changed == 0:
            strict = True
            break
        if _np_sum_small(sh2) <= tol_abs:
            break
    if not strict:
        shard.step(mean, c, False)
    inertia = allreduce(np.array([shard.inertia(mean, c)]), "sum")[0]
    return c + mean, inertia, it + 1
