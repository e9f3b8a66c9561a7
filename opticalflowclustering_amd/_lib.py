"""ctypes binding of libofc.so (include/ofc.h).  Thin: numpy arrays in, numpy arrays out, error codes
turned into Python exceptions.  There is no CPU fallback -- if the shared library is missing or no
gfx950 device is present the calls raise."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libofc.so")

OFC_OK, OFC_EINVAL, OFC_ENODEV, OFC_EHIP, OFC_ENOMEM, OFC_ENOTREADY, OFC_EUNSUPPORTED, OFC_ECOMM = \
    0, -1, -2, -3, -4, -5, -6, -7
U8, F32, F64 = 0, 1, 2
UNIQUE_ID_BYTES = 128


class OfcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libofc error {code}: {msg}")
        self.code = code


class FbParams(C.Structure):
    """ofc_fb_params; defaults = cv2.calcOpticalFlowFarneback(..., 0.5, 3, 15, 3, 5, 1.2, 0)
    as called at computeOpticalFlowModule.py:20-22"""
    _fields_ = [("pyr_scale", C.c_double), ("levels", C.c_int), ("winsize", C.c_int),
                ("iterations", C.c_int), ("poly_n", C.c_int), ("poly_sigma", C.c_double),
                ("flags", C.c_int)]

    def __init__(self, pyr_scale=0.5, levels=3, winsize=15, iterations=3, poly_n=5, poly_sigma=1.2,
                 flags=0):
        super().__init__(pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags)


_lib = None

_vp, _i, _i64, _f, _d, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double, C.c_size_t
_ip = C.POINTER(C.c_int)
_PROTOS = {
    "ofc_version": ([], _i),
    "ofc_last_error": ([], C.c_char_p),
    "ofc_device_count": ([_ip], _i),
    "ofc_malloc": ([_i, _sz, C.POINTER(_vp)], _i),
    "ofc_free": ([_i, _vp], _i),
    "ofc_memcpy_h2d": ([_i, _vp, _vp, _sz], _i),
    "ofc_memcpy_d2h": ([_i, _vp, _vp, _sz], _i),
    "ofc_memset": ([_i, _vp, _i, _sz], _i),
    "ofc_device_sync": ([_i], _i),
    "ofc_fb_default_params": ([C.POINTER(FbParams)], None),
    "ofc_flow_create": ([_i, _i, _i, C.POINTER(FbParams), _i, C.POINTER(_vp)], _i),
    "ofc_flow_destroy": ([_vp], None),
    "ofc_flow_calc": ([_vp, _vp, _vp, _vp], _i),
    "ofc_flow_calc_frames_dev": ([_vp, _vp, _i, _vp], _i),
    "ofc_flow_calc_frames_dev_stats": ([_vp, _vp, _i, _vp, _vp], _i),
    "ofc_flow_sync": ([_vp], _i),
    "ofc_flow_push_gray": ([_vp, _vp, _vp], _i),
    "ofc_flow_push_bgr": ([_vp, _vp, _vp, C.POINTER(_f), _vp], _i),
    "ofc_flow_last_vis_dev": ([_vp, C.POINTER(_vp)], _i),
    "ofc_level_image": ([_i, _vp, _i, _i, C.POINTER(FbParams), _i, _vp, _ip, _ip], _i),
    "ofc_polyexp": ([_i, _vp, _i, _i, _i, _d, _vp], _i),
    "ofc_polyexp_u8": ([_i, _vp, _i, _i, _i, _d, _vp], _i),
    "ofc_update_matrices": ([_i, _vp, _vp, _vp, _i, _i, _vp], _i),
    "ofc_box_solve": ([_i, _vp, _i, _i, _i, _vp], _i),
    "ofc_flow_resize": ([_i, _vp, _i, _i, _i, _i, _f, _vp], _i),
    "ofc_flow_iterate": ([_i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp], _i),
    "ofc_bench_flow_iters": ([_i, _i, _i, _i, _i, _i, C.POINTER(_f)], _i),
    "ofc_bench_polyexp": ([_i, _i, _i, _i, _i, _i, C.POINTER(_f)], _i),
    "ofc_bgr2gray": ([_i, _vp, _i, _i, _vp], _i),
    "ofc_flow_to_bgr": ([_i, _vp, _i, _i, _vp, C.POINTER(_f)], _i),
    "ofc_flow_to_bgr_dev": ([_i, _vp, _i, _i, _i, _vp, _vp], _i),
    "ofc_bgr2hsv": ([_i, _vp, _i64, _vp], _i),
    "ofc_preprocess_rgba": ([_i, _vp, _i64, _i, _vp], _i),
    "ofc_grid_cell_means": ([_i, _vp, _i, _i, _i, _i, _vp, _vp], _i),
    "ofc_kmeans_fit": ([_i, _vp, _i, _i64, _i, _i, _vp, _i, _d, _vp, _vp, C.POINTER(_d), _ip], _i),
    "ofc_kmeans_predict": ([_i, _vp, _i, _i64, _i, _i, _vp, _vp], _i),
    "ofc_kmeans_fit_dev": ([_i, _vp, _i, _i64, _i, _i, _vp, _i, _d, _vp, _vp, C.POINTER(_d), _ip], _i),
    "ofc_kmeans_fit_dev_stats": ([_i, _vp, _i, _i64, _i, _i, _vp, _i, _d, _vp, _vp, _vp, C.POINTER(_d), _ip], _i),
    "ofc_lloyd_prune_stats": ([_i, _vp], _i),
    "ofc_bench_lloyd_sweep": ([_i, _vp, _i64, _i, _vp, _vp, _i, _i, C.POINTER(_f)], _i),
    "ofc_lloyd_colstats_dev": ([_i, _vp, _i, _i64, _i, _vp, _i, _vp], _i),
    "ofc_lloyd_step_dev": ([_i, _vp, _i, _i64, _i, _i, _vp, _vp, _vp, _i, _vp], _i),
    "ofc_lloyd_inertia_dev": ([_i, _vp, _i, _i64, _i, _i, _vp, _vp, _vp, C.POINTER(_d)], _i),
    "ofc_lloyd_farthest_dev": ([_i, _vp, _i, _i64, _i, _i, _vp, _vp, _vp, _vp, _i, C.POINTER(_d), C.POINTER(_i64), _vp, _ip], _i),
    "ofc_kpp_candidates": ([_i, _vp, _i, _i64, _i, _vp, _vp, _i, _vp, _vp, _vp], _i),
    "ofc_kmeans_fit_batched": ([_i, _vp, _vp, _i, _i, _i, _vp, _i, _d, _vp, _vp, _vp, _vp], _i),
    "ofc_grid_kmeans": ([_i, _vp, _i, _i, _i, _i, _i, _vp, _i, _d, _i, _vp, _vp], _i),
    "ofc_grid_kmeans_dev": ([_i, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _d, _i, _vp, _vp], _i),
    "ofc_dist_unique_id": ([_vp], _i),
    "ofc_dist_init": ([_i, _i, _i, _vp], _i),
    "ofc_dist_allreduce_f64": ([_i, _vp, _i], _i),
    "ofc_dist_init_host": ([_i, _i, _i, _vp, _vp], _i),
    "ofc_dist_loopback": ([_i], _i),
    "ofc_dist_finalize": ([], _i),
    "ofc_stream_create": ([_i, _i, _i, C.POINTER(FbParams), _i, _i, _i, C.POINTER(_vp)], _i),
    "ofc_stream_push_gray": ([_vp, _vp, _ip], _i),
    "ofc_stream_finish": ([_vp, _vp, _i, _ip], _i),
    "ofc_stream_destroy": ([_vp], None),
    "ofc_grid_cell_mean_flow": ([_i, _vp, _i, _i, _i, _i, _vp], _i),
    "ofc_sliding_cosine": ([_i, _vp, _i, _vp, _i, _vp], _i),
    "ofc_synth_frames_dev": ([_i, _vp, _i, _i, _i, _i, _i], _i),
}
EXPORTS = tuple(_PROTOS)


def load():
    """dlopen libofc.so (built in-tree by __graft_entry__.build() / csrc/Makefile)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OfcError(OFC_ENODEV, f"{LIB_PATH} not built: run `make -C {_HERE}/csrc` "
                                       "(python -c 'import __graft_entry__ as g; g.build()')")
        lib = C.CDLL(LIB_PATH)
        for name, (args, res) in _PROTOS.items():
            fn = getattr(lib, name)          # AttributeError = missing export: fail loudly
            fn.argtypes = args
            fn.restype = res
        _lib = lib
    return _lib


def check(rc):
    if rc != OFC_OK:
        msg = load().ofc_last_error().decode("utf-8", "replace")
        if rc == OFC_EINVAL:
            raise ValueError(msg)
        raise OfcError(rc, msg)


def ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def device_count():
    n = C.c_int()
    check(load().ofc_device_count(C.byref(n)))
    return n.value


class DeviceBuffer:
    """a hipMalloc'ed buffer owned through the C ABI (inputs stay resident in HBM between calls)"""

    def __init__(self, nbytes, device=0):
        self.device, self.nbytes = device, int(nbytes)
        p = C.c_void_p()
        check(load().ofc_malloc(device, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, arr, offset=0):
        arr = np.ascontiguousarray(arr)
        assert offset + arr.nbytes <= self.nbytes
        check(load().ofc_memcpy_h2d(self.device, self.ptr + offset, ptr(arr), arr.nbytes))
        return self

    def download(self, shape, dtype, offset=0):
        out = np.empty(shape, dtype)
        assert offset + out.nbytes <= self.nbytes
        check(load().ofc_memcpy_d2h(self.device, ptr(out), self.ptr + offset, out.nbytes))
        return out

    def zero(self):
        check(load().ofc_memset(self.device, self.ptr, 0, self.nbytes))

    def free(self):
        if self.ptr:
            load().ofc_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
