"""Device-resident clip pipeline (BASELINE.json configs[2]/[3]): a clip's frames sit in HBM, dense
Farneback flow is computed for every consecutive pair in batches, and Lloyd's k-means runs over the
per-pixel (u,v) vectors of the whole clip without the flow ever leaving the device.  With a
communicator (dist.py) each rank owns a contiguous range of pairs (plus the one-frame halo) and the
Lloyd partial sums are all-reduced over RCCL once per iteration."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import DeviceBuffer, FbParams, check, load
from .cluster import kmeans_fit_dev
from .flow import FlowEngine


def shard_pairs(n_pairs, world, rank):
    """contiguous ranges of pair indices, sizes differing by at most one"""
    base, rem = divmod(n_pairs, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def batch_schedule(n_pairs, batch_pairs):
    """sizes of the flow launch sequences a clip of n_pairs is cut into.  batch_pairs: pairs per batch, or an explicit schedule.
    The short batch goes FIRST: at the end of a step both engines then finish on full batches instead of one of them running
    a half-empty launch sequence alone (36.7 against 37.0 ms per 299-pair step)"""
    if isinstance(batch_pairs, (list, tuple)):
        schedule = [int(b) for b in batch_pairs]
        if sum(schedule) != n_pairs or min(schedule) < 1:
            raise ValueError(f"batch schedule {schedule} does not cover {n_pairs} pairs")
        return schedule
    b = max(1, min(int(batch_pairs), n_pairs))
    return ([n_pairs % b] if n_pairs % b else []) + [b] * (n_pairs // b)


class ClipPipeline:
    def __init__(self, W, H, n_frames_local, batch_pairs=16, params=None, device=0, n_engines=2):
        self.W, self.H, self.device = W, H, device
        self.n_frames = int(n_frames_local)
        self.n_pairs = self.n_frames - 1
        self.schedule = batch_schedule(self.n_pairs, batch_pairs)
        self.batch = max(self.schedule)
        # two engines = two HIP streams: consecutive batches are independent, so their kernels overlap and fill
        # each other's tails / latency-bound phases
        self.engines = [FlowEngine(W, H, params or FbParams(), max_batch=self.batch, device=device)
                        for _ in range(max(1, min(n_engines, len(self.schedule))))]
        self.engine = self.engines[0]
        P = W * H
        self.frames = DeviceBuffer(self.n_frames * P, device)
        self.flows = DeviceBuffer(self.n_pairs * P * 8, device)
        self.labels = DeviceBuffer(self.n_pairs * P, device)
        # sum(u), sum(v) per batch, written by the flow engine with the field (epilogue of the last level-0 iteration):
        # run_kmeans adds them up in batch order instead of sweeping the 5 GB of vectors once more for the column means
        self.n_batches = len(self.schedule)
        self.uv_sums = DeviceBuffer(self.n_batches * 16, device)
        self._sums_valid = False

    def synth(self, t0=0, seed=0):
        """fill the resident clip with synthetic frames t0 .. t0+n_frames-1"""
        check(load().ofc_synth_frames_dev(self.device, C.c_void_p(self.frames.ptr), self.W, self.H,
                                          self.n_frames, t0, seed))

    def upload_frames(self, frames):
        frames = np.ascontiguousarray(frames, np.uint8)
        assert frames.shape == (self.n_frames, self.H, self.W)
        self.frames.upload(frames)

    def run_flow(self, sync=True, stats=True):
        P = self.W * self.H
        p0 = 0
        for i, n in enumerate(self.schedule):
            self.engines[i % len(self.engines)].calc_frames_dev(self.frames.ptr + p0 * P, n + 1,
                                                                self.flows.ptr + p0 * P * 8, sync=False,
                                                                uv_sum_ptr=self.uv_sums.ptr + 16 * i if stats else None)
            p0 += n
        self._sums_valid = stats
        if sync:
            self.sync()

    def sync(self):
        for e in self.engines:
            e.sync()

    def run_kmeans(self, init, max_iter=300, tol=1e-4):
        """Lloyd over all local (u,v) vectors (global when a communicator is active).
        -> centers (k,2), inertia, n_iter"""
        self.sync()
        N = self.n_pairs * self.W * self.H
        colsum = None
        if self._sums_valid:
            per_batch = self.uv_sums.download((self.n_batches, 2), np.float64)
            colsum = np.zeros(2)
            for b in range(self.n_batches):                 # fixed order
                colsum += per_batch[b]
        return kmeans_fit_dev(self.flows.ptr, _lib.F32, N, 2, init, max_iter, tol, self.labels.ptr, self.device, colsum=colsum)

    def sample_uv(self, idx):
        """host copy of a few (u,v) rows (for choosing the initial centres)"""
        out = np.empty((len(idx), 2), np.float32)
        for i, j in enumerate(idx):
            out[i] = self.flows.download((2,), np.float32, offset=int(j) * 8)
        return out

    def flows_host(self):
        return self.flows.download((self.n_pairs, self.H, self.W, 2), np.float32)

    def labels_host(self):
        return self.labels.download((self.n_pairs, self.H, self.W), np.uint8)

    def close(self):
        for e in self.engines:
            e.close()
        for b in (self.frames, self.flows, self.labels, self.uv_sums):
            b.free()
