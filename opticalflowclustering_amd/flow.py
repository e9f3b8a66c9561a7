"""Farneback flow engine (libofc ofc_flow_*): the device-side replacement of
cv2.calcOpticalFlowFarneback as called at computeOpticalFlowModule.py:20-22."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import FbParams, check, load, ptr


class FlowEngine:
    """one engine per (device, resolution); owns all device scratch for `max_batch` frame pairs"""

    def __init__(self, W, H, params=None, max_batch=1, device=0):
        self.W, self.H, self.device, self.max_batch = int(W), int(H), device, int(max_batch)
        self.params = params or FbParams()
        h = C.c_void_p()
        check(load().ofc_flow_create(device, self.W, self.H, C.byref(self.params), self.max_batch, C.byref(h)))
        self._h = h

    def calc(self, prev_gray, next_gray):
        """one isolated pair, host arrays -> HxWx2 float32 (u = x-displacement, v = y-displacement)"""
        prev_gray = np.ascontiguousarray(prev_gray, np.uint8)
        next_gray = np.ascontiguousarray(next_gray, np.uint8)
        if prev_gray.shape != (self.H, self.W) or next_gray.shape != (self.H, self.W):
            raise ValueError(f"expected two {self.H}x{self.W} uint8 images")
        flow = np.empty((self.H, self.W, 2), np.float32)
        check(load().ofc_flow_calc(self._h, ptr(prev_gray), ptr(next_gray), ptr(flow)))
        return flow

    def push(self, gray):
        """streaming: returns None for the first frame, then the flow prev->gray"""
        gray = np.ascontiguousarray(gray, np.uint8)
        if gray.shape != (self.H, self.W):
            raise ValueError(f"expected a {self.H}x{self.W} uint8 image")
        flow = np.empty((self.H, self.W, 2), np.float32)
        rc = load().ofc_flow_push_gray(self._h, ptr(gray), ptr(flow))
        if rc == _lib.OFC_ENOTREADY:
            return None
        check(rc)
        return flow

    def calc_frames_dev(self, frames_ptr, n_frames, flow_ptr, sync=True, uv_sum_ptr=None):
        """device pointers: n_frames resident u8 frames -> n_frames-1 flows (async unless sync); uv_sum_ptr: device address
        of two doubles that receive sum(u), sum(v) of those flows (from the last iteration's epilogue)"""
        if uv_sum_ptr:
            check(load().ofc_flow_calc_frames_dev_stats(self._h, C.c_void_p(frames_ptr), n_frames, C.c_void_p(flow_ptr),
                                                        C.c_void_p(uv_sum_ptr)))
        else:
            check(load().ofc_flow_calc_frames_dev(self._h, C.c_void_p(frames_ptr), n_frames, C.c_void_p(flow_ptr)))
        if sync:
            self.sync()

    def sync(self):
        check(load().ofc_flow_sync(self._h))

    def close(self):
        if getattr(self, "_h", None):
            load().ofc_flow_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
