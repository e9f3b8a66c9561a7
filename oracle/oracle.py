"""numpy/ctypes front-end of the CPU oracle (oracle/*.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never from the opticalflowclustering_amd package.

Every function restates one step of the reference's hot path on the CPU; see the C files for the
reference file:line each one follows.  Farneback flow is PARITY UNPINNED (no recorded flow in the
reference); Lloyd is pinned by sklearn goldens, the colour routines by the reference's recorded CSVs.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libofc_oracle.so")


class FbParams(C.Structure):
    _fields_ = [("pyr_scale", C.c_double), ("levels", C.c_int), ("winsize", C.c_int),
                ("iterations", C.c_int), ("poly_n", C.c_int), ("poly_sigma", C.c_double),
                ("flags", C.c_int)]


def default_params():
    # computeOpticalFlowModule.py:20-22
    return FbParams(0.5, 3, 15, 3, 5, 1.2, 0)


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
            for f in os.listdir(_HERE) if f.endswith((".c", ".h"))):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.ofc_ref_farneback.restype = C.c_int
        _lib.ofc_ref_kmeans_fit.restype = C.c_int
        _lib.ofc_ref_kmeans_predict.restype = C.c_int
        _lib.ofc_ref_lloyd_partials.restype = C.c_int
        _lib.ofc_ref_pyramid_levels.restype = C.c_int
    return _lib


_NATIVE_DIR = os.path.join(_HERE, "_native")


class _Native:
    """the same C files built -O3 -march=native ON THE HOST THAT RUNS THIS (bench.py's cpu_baseline leg): never shipped,
    never built ahead of time (oracle/_native/ is git- and gpurun-ignored: a -march=native object must not travel)."""

    def __init__(self):
        os.makedirs(_NATIVE_DIR, exist_ok=True)
        so = os.path.join(_NATIVE_DIR, "libofc_oracle_native.so")
        srcs = [os.path.join(_HERE, f) for f in ("farneback_ref.c", "lloyd_ref.c", "color_ref.c")]
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-shared",
                               "-o", so] + srcs + ["-lm"])
        self._lib = C.CDLL(so)
        for f in ("ofc_ref_farneback", "ofc_ref_kmeans_fit", "ofc_ref_lloyd_partials"):
            getattr(self._lib, f).restype = C.c_int

    def farneback(self, prev, nxt, params=None):
        return farneback(prev, nxt, params, _lib=self._lib)

    def kmeans_fit(self, X, init, max_iter=300, tol=1e-4):
        return kmeans_fit(X, init, max_iter, tol, _lib=self._lib)

    def lloyd_partials(self, X, mean, centers_c, labels):
        return lloyd_partials(X, mean, centers_c, labels, _lib=self._lib)


_native = None


def native():
    global _native
    if _native is None:
        _native = _Native()
    return _native


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


# ---------------------------------------------------------------- Farneback pieces
def gaussian_kernel(n, sigma):
    k = np.empty(n, np.float32)
    lib().ofc_ref_gaussian_kernel(C.c_int(n), C.c_double(sigma), _p(k))
    return k


def gaussian_blur(img, ksize, sigma):
    img = _f32(img)
    H, W = img.shape
    out = np.empty_like(img)
    lib().ofc_ref_gaussian_blur(_p(img), W, H, ksize, C.c_double(sigma), _p(out))
    return out


def resize_linear(img, dw, dh):
    img = _f32(img)
    cn = 1 if img.ndim == 2 else img.shape[2]
    sh, sw = img.shape[:2]
    out = np.empty((dh, dw) if img.ndim == 2 else (dh, dw, cn), np.float32)
    lib().ofc_ref_resize_linear(_p(img), sw, sh, cn, _p(out), dw, dh)
    return out


def polyexp_setup(n=5, sigma=1.2):
    g = np.zeros(n + 1, np.float32)
    xg = np.zeros(n + 1, np.float32)
    xxg = np.zeros(n + 1, np.float32)
    ig = np.zeros(4, np.float64)
    lib().ofc_ref_polyexp_setup(n, C.c_double(sigma), _p(g), _p(xg), _p(xxg), _p(ig))
    return g, xg, xxg, ig


def polyexp(img, n=5, sigma=1.2):
    """f32 HxW -> f32 HxWx5 {y-lin, x-lin, y^2, x^2, xy}"""
    img = _f32(img)
    H, W = img.shape
    out = np.empty((H, W, 5), np.float32)
    lib().ofc_ref_polyexp(_p(img), W, H, n, C.c_double(sigma), _p(out))
    return out


def update_matrices(R0, R1, flow):
    R0, R1, flow = _f32(R0), _f32(R1), _f32(flow)
    H, W = flow.shape[:2]
    M = np.empty((H, W, 5), np.float32)
    lib().ofc_ref_update_matrices(_p(R0), _p(R1), _p(flow), _p(M), W, H, 0, H)
    return M


def update_flow_blur(R0, R1, flow, M, block_size=15, update_mats=True):
    """returns (new_flow, new_M); inputs are not modified"""
    R0, R1 = _f32(R0), _f32(R1)
    flow = _f32(flow).copy()
    M = _f32(M).copy()
    H, W = flow.shape[:2]
    lib().ofc_ref_update_flow_blur(_p(R0), _p(R1), _p(flow), _p(M), W, H, block_size,
                                   1 if update_mats else 0)
    return flow, M


def level_geometry(W, H, k, params=None):
    p = params or default_params()
    w, h, ks = C.c_int(), C.c_int(), C.c_int()
    sg = C.c_double()
    lib().ofc_ref_level_geometry(W, H, C.byref(p), k, C.byref(w), C.byref(h), C.byref(ks),
                                 C.byref(sg))
    return w.value, h.value, ks.value, sg.value


def pyramid_levels(W, H, params=None):
    p = params or default_params()
    return lib().ofc_ref_pyramid_levels(W, H, C.byref(p))


def level_image(gray, k, params=None):
    p = params or default_params()
    gray = _u8(gray)
    H, W = gray.shape
    w, h, _, _ = level_geometry(W, H, k, p)
    out = np.empty((h, w), np.float32)
    lib().ofc_ref_level_image(_p(gray), W, H, C.byref(p), k, _p(out))
    return out


def farneback(prev, nxt, params=None, _lib=None):
    """cv2.calcOpticalFlowFarneback(prev, next, None, 0.5, 3, 15, 3, 5, 1.2, 0) -> HxWx2 f32"""
    p = params or default_params()
    prev, nxt = _u8(prev), _u8(nxt)
    assert prev.shape == nxt.shape and prev.ndim == 2
    H, W = prev.shape
    flow = np.empty((H, W, 2), np.float32)
    rc = (_lib or lib()).ofc_ref_farneback(_p(prev), _p(nxt), W, H, C.byref(p), _p(flow))
    if rc != 0:
        raise ValueError("oracle farneback: unsupported parameters")
    return flow


# ---------------------------------------------------------------- colour
def bgr2gray(bgr):
    bgr = _u8(bgr)
    out = np.empty(bgr.shape[:-1], np.uint8)
    lib().ofc_ref_bgr2gray(_p(bgr), C.c_int64(out.size), _p(out))
    return out


def bgr2hsv(bgr):
    bgr = _u8(bgr)
    out = np.empty_like(bgr)
    lib().ofc_ref_bgr2hsv(_p(bgr), C.c_int64(bgr.size // 3), _p(out))
    return out


def hsv2bgr(hsv):
    hsv = _u8(hsv)
    out = np.empty_like(hsv)
    lib().ofc_ref_hsv2bgr(_p(hsv), C.c_int64(hsv.size // 3), _p(out))
    return out


def cart_to_polar(x, y):
    x, y = _f32(x), _f32(y)
    mag, ang = np.empty_like(x), np.empty_like(x)
    lib().ofc_ref_cart_to_polar(_p(x), _p(y), C.c_int64(x.size), _p(mag), _p(ang))
    return mag, ang


def flow_to_bgr(flow):
    """computeOpticalFlowModule.py:25-33 -> (HxWx3 u8 BGR, mean magnitude)"""
    flow = _f32(flow)
    H, W = flow.shape[:2]
    out = np.empty((H, W, 3), np.uint8)
    mm = C.c_float()
    lib().ofc_ref_flow_to_bgr(_p(flow), W, H, _p(out), C.byref(mm))
    return out, mm.value


def grid_cell_means(bgr, rows=14, cols=25):
    bgr = _u8(bgr)
    H, W = bgr.shape[:2]
    mean = np.empty((rows * cols, 3), np.uint8)
    hsv = np.empty((rows * cols, 3), np.uint8)
    lib().ofc_ref_grid_cell_means(_p(bgr), W, H, rows, cols, _p(mean), _p(hsv))
    return mean, hsv


def extract_cell(bgr, cell, rows=14, cols=25):
    bgr = _u8(bgr)
    H, W = bgr.shape[:2]
    out = np.empty((H // rows, W // cols, 3), np.uint8)
    lib().ofc_ref_extract_cell(_p(bgr), W, H, rows, cols, cell, _p(out))
    return out


def preprocess_rgba(img3, thresh=30):
    img3 = _u8(img3)
    out = np.empty(img3.shape[:-1] + (4,), np.uint8)
    lib().ofc_ref_preprocess_rgba(_p(img3), C.c_int64(img3.size // 3), thresh, _p(out))
    return out


# ---------------------------------------------------------------- Lloyd
_DT = {np.dtype(np.uint8): 0, np.dtype(np.float32): 1, np.dtype(np.float64): 2}


def kmeans_fit(X, init, max_iter=300, tol=1e-4, _lib=None):
    """-> centers (k,d) f64, labels (N,) i32, inertia, n_iter"""
    X = np.ascontiguousarray(X)
    if X.dtype not in _DT:
        X = X.astype(np.float64)
    init = np.ascontiguousarray(init, np.float64)
    N, d = X.shape
    k = init.shape[0]
    centers = np.empty((k, d), np.float64)
    labels = np.empty(N, np.int32)
    inertia = C.c_double()
    n_iter = C.c_int()
    rc = (_lib or lib()).ofc_ref_kmeans_fit(_p(X), _DT[X.dtype], C.c_int64(N), d, k, _p(init), max_iter,
                                  C.c_double(tol), _p(centers), _p(labels), C.byref(inertia),
                                  C.byref(n_iter))
    if rc != 0:
        raise ValueError(f"n_samples={N} should be >= n_clusters={k}.")
    return centers, labels, inertia.value, n_iter.value


def kmeans_predict(X, centers):
    X = np.ascontiguousarray(X)
    if X.dtype not in _DT:
        X = X.astype(np.float64)
    centers = np.ascontiguousarray(centers, np.float64)
    N, d = X.shape
    labels = np.empty(N, np.int32)
    lib().ofc_ref_kmeans_predict(_p(X), _DT[X.dtype], C.c_int64(N), d, centers.shape[0],
                                 _p(centers), _p(labels))
    return labels


def lloyd_partials(X, mean, centers_c, labels, _lib=None):
    """shard step: labels (i32, in/out). -> [sums k*d | counts k | n_changed]"""
    X = np.ascontiguousarray(X)
    mean = np.ascontiguousarray(mean, np.float64)
    centers_c = np.ascontiguousarray(centers_c, np.float64)
    N, d = X.shape
    k = centers_c.shape[0]
    out = np.empty(k * d + k + 1, np.float64)
    (_lib or lib()).ofc_ref_lloyd_partials(_p(X), _DT[X.dtype], C.c_int64(N), d, k, _p(mean),
                                 _p(centers_c), _p(labels), _p(out))
    return out


def kpp_candidates(X, mean, cand, closest=None):
    """CPU restatement of one k-means++ seeding step (sklearn/cluster/_kmeans.py:250-256 with
    _euclidean_distances' expanded form, metrics/pairwise.py): -> (min(closest, dist) [n_cand][N], potentials)"""
    Xc = np.asarray(X, np.float64) - np.asarray(mean, np.float64)
    Y = Xc[np.asarray(cand, np.int64)]
    xx = (Xc * Xc).sum(axis=1)
    yy = (Y * Y).sum(axis=1)
    dist = np.maximum((-2.0 * (Y @ Xc.T) + yy[:, None]) + xx[None, :], 0.0)
    if closest is not None:
        dist = np.minimum(dist, np.asarray(closest, np.float64)[None, :])
    return dist, dist.sum(axis=1)
