"""numpy-level access to the individual Farneback stages of libofc (parity-test and bench hooks).
Layouts are OpenCV's internal interleaved ones so results compare 1:1 with the oracle."""
import ctypes as C

import numpy as np

from ._lib import FbParams, check, load, ptr


def level_image(gray, k, params=None, device=0):
    gray = np.ascontiguousarray(gray, np.uint8)
    H, W = gray.shape
    p = params or FbParams()
    out = np.empty((H, W), np.float32)
    w, h = C.c_int(), C.c_int()
    check(load().ofc_level_image(device, ptr(gray), W, H, C.byref(p), k, ptr(out), C.byref(w), C.byref(h)))
    return out.reshape(-1)[: w.value * h.value].reshape(h.value, w.value).copy()


def polyexp(img, n=5, sigma=1.2, device=0):
    img = np.ascontiguousarray(img, np.float32)
    H, W = img.shape
    out = np.empty((H, W, 5), np.float32)
    check(load().ofc_polyexp(device, ptr(img), W, H, n, sigma, ptr(out)))
    return out


def polyexp_u8(gray, n=5, sigma=1.2, device=0):
    """polyexp of pyramid level 0 straight from the u8 frame (the fused form the flow engine runs)"""
    gray = np.ascontiguousarray(gray, np.uint8)
    H, W = gray.shape
    out = np.empty((H, W, 5), np.float32)
    check(load().ofc_polyexp_u8(device, ptr(gray), W, H, n, sigma, ptr(out)))
    return out


def update_matrices(R0, R1, flow, device=0):
    R0, R1, flow = (np.ascontiguousarray(a, np.float32) for a in (R0, R1, flow))
    H, W = flow.shape[:2]
    M = np.empty((H, W, 5), np.float32)
    check(load().ofc_update_matrices(device, ptr(R0), ptr(R1), ptr(flow), W, H, ptr(M)))
    return M


def box_solve(M, winsize=15, device=0):
    M = np.ascontiguousarray(M, np.float32)
    H, W = M.shape[:2]
    flow = np.empty((H, W, 2), np.float32)
    check(load().ofc_box_solve(device, ptr(M), W, H, winsize, ptr(flow)))
    return flow


def flow_resize(flow, dw, dh, mul=1.0, device=0):
    flow = np.ascontiguousarray(flow, np.float32)
    sh, sw = flow.shape[:2]
    out = np.empty((dh, dw, 2), np.float32)
    check(load().ofc_flow_resize(device, ptr(flow), sw, sh, dw, dh, mul, ptr(out)))
    return out


def flow_iterate(R0, R1, flow_in, iters, winsize=15, mode=0, rows_per_block=0, device=0):
    """`iters` Farneback iterations from flow_in with the engine's fused kernels (mode 1: two iterations per launch)"""
    R0, R1, flow_in = (np.ascontiguousarray(a, np.float32) for a in (R0, R1, flow_in))
    H, W = flow_in.shape[:2]
    out = np.empty((H, W, 2), np.float32)
    check(load().ofc_flow_iterate(device, ptr(R0), ptr(R1), ptr(flow_in), W, H, winsize, iters, mode, rows_per_block,
                                  ptr(out)))
    return out


def bench_flow_iters(W, H, n_pairs, reps, mode, device=0):
    """ms per batch for the last two iterations of a level (mode 0: two launches, mode 1: the two-iteration kernel)"""
    ms = C.c_float()
    check(load().ofc_bench_flow_iters(device, W, H, n_pairs, reps, mode, C.byref(ms)))
    return ms.value


def bench_polyexp(W, H, n_images, iters, rows_per_block=0, device=0):
    ms = C.c_float()
    check(load().ofc_bench_polyexp(device, W, H, n_images, iters, rows_per_block, C.byref(ms)))
    return ms.value


def bench_lloyd_sweep(X_ptr, N, centers, mean, what, iters=10, device=0):
    """ms per launch of one Lloyd sweep over a resident (u,v) stream (ofc_bench_lloyd_sweep): what 0 full label-less sweep,
    1 pruned tile sweep, 2 metadata-building sweep, 3 final E-step"""
    cen = np.ascontiguousarray(centers, np.float64)
    mean = np.ascontiguousarray(mean, np.float64)
    ms = C.c_float()
    check(load().ofc_bench_lloyd_sweep(device, C.c_void_p(X_ptr), N, len(cen), ptr(cen), ptr(mean), what, iters, C.byref(ms)))
    return ms.value
