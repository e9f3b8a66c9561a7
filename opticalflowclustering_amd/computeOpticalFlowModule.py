"""Drop-in for k-means-color-clustering/computeOpticalFlowModule.py: same class name (typo included),
same constructor and compute() contract, the arithmetic on the MI355X.

    from opticalflowclustering_amd.computeOpticalFlowModule import ComputeOpticalFLow
    compflow = ComputeOpticalFLow(firstframe)          # KmeanGrids.py:178
    frame_optical = compflow.compute(frame_rgb)        # KmeanGrids.py:187
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import FbParams, check, load, ptr


class ComputeOpticalFLow:
    """reference: computeOpticalFlowModule.py:6-36.  Holds one frame of state (prev_gray) on the device;
    compute(frame) returns a NEW HxWx3 uint8 BGR array each call (callers mutate it, KmeanGrids.py:108,277):
    BGR2GRAY -> calcOpticalFlowFarneback(prev, gray, None, 0.5, 3, 15, 3, 5, 1.2, 0) -> cartToPolar ->
    H = angle*180/pi/2, S = 255, V = normalize(mag, 0, 255, MINMAX) (uint8 truncation) -> HSV2BGR."""

    def __init__(self, firstframe, device=0):
        self.firstframe = firstframe
        self.width = self.firstframe.shape[1]
        self.height = self.firstframe.shape[0]
        self.device = device
        self.last_mean_magnitude = None
        h = C.c_void_p()
        p = FbParams()
        check(load().ofc_flow_create(device, self.width, self.height, C.byref(p), 1, C.byref(h)))
        self._h = h
        rc = load().ofc_flow_push_bgr(self._h, ptr(self._as_bgr(firstframe)), None, None, None)
        if rc != _lib.OFC_ENOTREADY:
            check(rc)

    def _as_bgr(self, frame):
        if frame is None:
            raise ValueError("frame is None (end of stream?)")       # the reference crashes inside cvtColor here
        frame = np.ascontiguousarray(frame, np.uint8)
        if frame.shape != (self.height, self.width, 3):
            raise ValueError(f"expected a {self.height}x{self.width}x3 uint8 BGR frame, got {frame.shape}")
        return frame

    def compute(self, frame, return_flow=False):
        frame = self._as_bgr(frame)
        rgb = np.empty((self.height, self.width, 3), np.uint8)
        flow = np.empty((self.height, self.width, 2), np.float32) if return_flow else None
        mm = C.c_float()
        check(load().ofc_flow_push_bgr(self._h, ptr(frame), ptr(rgb), C.byref(mm), ptr(flow)))
        self.last_mean_magnitude = mm.value
        return (rgb, flow) if return_flow else rgb

    def vis_device_ptr(self):
        """device address of the visualisation compute() just produced (for ofc_grid_kmeans_dev)"""
        p = C.c_void_p()
        check(load().ofc_flow_last_vis_dev(self._h, C.byref(p)))
        return p.value

    def close(self):
        if getattr(self, "_h", None):
            load().ofc_flow_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def main(vid_path="video_lq.mp4"):
    """reference __main__ (computeOpticalFlowModule.py:38-50) without the imshow window"""
    from .frameio import FrameSource
    cap = FrameSource(vid_path)
    ret, firstframe = cap.read()
    compflow = ComputeOpticalFLow(firstframe)
    n = 0
    while cap.isOpened():
        ret, frame = cap.read()
        if not ret:                      # the reference does not check and crashes at end of stream
            break
        compflow.compute(frame)
        n += 1
    print("frames processed", n)


if __name__ == "__main__":
    import sys
    main(*sys.argv[1:2])
