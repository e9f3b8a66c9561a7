"""Streaming ingest (BASELINE.json configs[4] shape): frames are pushed one at a time (as a decoder produces them),
packed into pinned ring buffers, uploaded with hipMemcpyAsync on a copy stream while the previous batch computes,
and reduced to the grid-cell averaged flow (rows*cols (u,v) means per pair).  k-means over those vectors is then a
small streaming-Lloyd problem (cluster.KMeans)."""
import ctypes as C

import numpy as np

from ._lib import FbParams, check, load, ptr


class FlowStream:
    def __init__(self, W, H, batch_pairs=8, rows=14, cols=25, params=None, device=0):
        self.W, self.H, self.rows, self.cols = W, H, rows, cols
        self.params = params or FbParams()
        h = C.c_void_p()
        check(load().ofc_stream_create(device, W, H, C.byref(self.params), batch_pairs, rows, cols, C.byref(h)))
        self._h = h
        self.pushed = 0

    def push(self, gray):
        gray = np.ascontiguousarray(gray, np.uint8)
        if gray.shape != (self.H, self.W):
            raise ValueError(f"expected a {self.H}x{self.W} uint8 frame")
        done = C.c_int()
        check(load().ofc_stream_push_gray(self._h, ptr(gray), C.byref(done)))
        self.pushed += 1
        return done.value

    def finish(self):
        """-> cell_uv (n_pairs, rows*cols, 2) float32"""
        n = max(self.pushed - 1, 0)
        out = np.empty((max(n, 1), self.rows * self.cols, 2), np.float32)
        got = C.c_int()
        check(load().ofc_stream_finish(self._h, ptr(out), max(n, 1), C.byref(got)))
        self.pushed = 0
        return out[:got.value]

    def close(self):
        if getattr(self, "_h", None):
            load().ofc_stream_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def grid_cell_mean_flow(flow, rows=14, cols=25, device=0):
    flow = np.ascontiguousarray(flow, np.float32)
    H, W = flow.shape[:2]
    out = np.empty((rows * cols, 2), np.float32)
    check(load().ofc_grid_cell_mean_flow(device, ptr(flow), W, H, rows, cols, ptr(out)))
    return out
