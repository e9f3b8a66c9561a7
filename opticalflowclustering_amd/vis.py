"""numpy-level access to the visualisation / grid / batched k-means entry points of libofc."""
import ctypes as C

import numpy as np

from ._lib import check, load, ptr


def bgr2gray(bgr, device=0):
    bgr = np.ascontiguousarray(bgr, np.uint8)
    H, W = bgr.shape[:2]
    out = np.empty((H, W), np.uint8)
    check(load().ofc_bgr2gray(device, ptr(bgr), W, H, ptr(out)))
    return out


def flow_to_bgr(flow, device=0):
    """computeOpticalFlowModule.py:25-33 -> (BGR uint8 HxWx3, np.mean(magnitude))"""
    flow = np.ascontiguousarray(flow, np.float32)
    H, W = flow.shape[:2]
    out = np.empty((H, W, 3), np.uint8)
    mm = C.c_float()
    check(load().ofc_flow_to_bgr(device, ptr(flow), W, H, ptr(out), C.byref(mm)))
    return out, mm.value


def grid_cell_means(bgr, rows=14, cols=25, device=0):
    bgr = np.ascontiguousarray(bgr, np.uint8)
    H, W = bgr.shape[:2]
    mean = np.empty((rows * cols, 3), np.uint8)
    hsv = np.empty((rows * cols, 3), np.uint8)
    check(load().ofc_grid_cell_means(device, ptr(bgr), W, H, rows, cols, ptr(mean), ptr(hsv)))
    return mean, hsv


def kmeans_fit_batched(X, offsets, k, init=None, max_iter=300, tol=1e-4, device=0):
    """many independent u8 RGBA problems in one launch.
    -> centers (P,k,4) f64, counts (P,k) = bincount(predict), labels (total,), n_iter (P,)"""
    X = np.ascontiguousarray(X, np.uint8)
    offsets = np.ascontiguousarray(offsets, np.int64)
    P = len(offsets) - 1
    if X.ndim != 2 or X.shape[1] != 4:
        raise ValueError("X must be (total, 4) uint8")
    if init is not None:
        init = np.ascontiguousarray(init, np.float64)
        if init.shape != (P, k, 4):
            raise ValueError(f"init must be ({P}, {k}, 4)")
    centers = np.empty((P, k, 4), np.float64)
    counts = np.empty((P, k), np.int32)
    labels = np.empty(len(X), np.int32)
    n_iter = np.empty(P, np.int32)
    check(load().ofc_kmeans_fit_batched(device, ptr(X), ptr(offsets), P, 4, k, ptr(init), max_iter, tol,
                                        ptr(centers), ptr(counts), ptr(labels), ptr(n_iter)))
    return centers, counts, labels, n_iter


def grid_kmeans(bgr, k=1, rows=14, cols=25, init=None, max_iter=300, tol=1e-4, channel_order=0, device=0):
    """KmeanGrids.py:376-392 for one frame: -> (rint'ed dominant centre (cells,4) f64, hsv (cells,3) u8)"""
    bgr = np.ascontiguousarray(bgr, np.uint8)
    H, W = bgr.shape[:2]
    nc = rows * cols
    if init is not None:
        init = np.ascontiguousarray(init, np.float64)
        if init.shape != (nc, k, 4):
            raise ValueError(f"init must be ({nc}, {k}, 4)")
    centers = np.empty((nc, 4), np.float64)
    hsv = np.empty((nc, 3), np.uint8)
    check(load().ofc_grid_kmeans(device, ptr(bgr), W, H, rows, cols, k, ptr(init), max_iter, tol, channel_order,
                                 ptr(centers), ptr(hsv)))
    return centers, hsv
