#!/usr/bin/env python3
"""bench.py -- headline benchmark of the flow -> k-means hot path on MI355X.

Workload (BASELINE.json configs[2]; configs[3] when launched on N > 1 GPUs): a 300-frame synthetic
1080p clip resident in HBM -> dense Farneback flow for all 299 consecutive pairs -> Lloyd's k-means
(k=5, fixed init, max_iter=300, tol=1e-4, run to convergence) over the 6.2e8 per-pixel (u,v) vectors.
One "step" = one full pass (flow + k-means) over the clip.  With N ranks the pairs are sharded
contiguously (strong scaling: the clip is fixed) and the Lloyd partial sums are all-reduced over RCCL
once per iteration.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (metric = Mpixels/s of flow+k-means, whole job), including
  roofline     -- the polynomial-expansion kernel (24 B/px algorithmic) timed with HIP events on its own
                  stream over 64 distinct resident 1080p images, against the 8 TB/s HBM peak
  cpu_baseline -- the CPU oracle (C restatement, 1 thread) timed on a bounded sample of the same clip.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, CLIP_FRAMES, K_CLUSTERS = 1920, 1080, 300, 5
INIT = np.array([[-3.0, -3.0], [-1.5, 1.0], [0.0, 0.0], [1.5, -1.0], [3.0, 3.0]])
HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
POLYEXP_BYTES_PER_PX = 24  # 4 B read + 5 x 4 B written per pixel per image


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=32, help="frame pairs per flow launch sequence")
    ap.add_argument("--frames", type=int, default=CLIP_FRAMES, help="clip length (default = the named config)")
    ap.add_argument("--engines", type=int, default=2, help="flow engines (HIP streams) fed round-robin")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..."
                     % (args.gpus, args.gpus))
    force_dist = os.environ.get("OFC_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        import torch  # noqa: F401  (first: libofc then shares torch's HIP/RCCL runtime instances)

    from opticalflowclustering_amd import _lib, dist, stages
    from opticalflowclustering_amd.pipeline import ClipPipeline, shard_pairs

    lib = _lib.load()
    device = local_rank if world > 1 else 0
    rank, world, barrier, allreduce_max = dist.init_from_torch_env(device)

    n_pairs_total = args.frames - 1
    p0, p1 = shard_pairs(n_pairs_total, world, rank)
    pipe = ClipPipeline(W, H, p1 - p0 + 1, batch_pairs=args.batch, device=device, n_engines=args.engines)
    pipe.synth(t0=p0, seed=0)

    def step():
        pipe.run_flow(sync=False)
        return pipe.run_kmeans(INIT, max_iter=300, tol=1e-4)

    res = None
    for _ in range(args.warmup):
        res = step()
    barrier()
    _lib.check(lib.ofc_device_sync(device))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    _lib.check(lib.ofc_device_sync(device))
    barrier()
    dt = allreduce_max(time.perf_counter() - t0)
    centers, inertia, n_iter = res
    # extra, outside the timed region: the flow alone over the same resident clip (the configs[1] shape: many distinct
    # 1080p pairs per launch sequence) -- reported as config.flow_only_mpx_s, not part of `value`
    barrier()
    tf = time.perf_counter()
    for _ in range(2):
        pipe.run_flow(sync=True)
    barrier()
    dt_flow = allreduce_max(time.perf_counter() - tf) / 2

    out = None
    if rank == 0:
        mpx = args.steps * n_pairs_total * W * H / 1e6
        out = {
            "metric": "Mpixels/s dense flow+kmeans @1080p",
            "value": mpx / dt,
            "unit": "Mpixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32 storage, f64 accumulation",
            "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[%d]: %d-frame 1080p synthetic clip, Farneback flow "
                                   "(0.5,3,15,3,5,1.2,0) + Lloyd k=5 over (u,v), frames sharded over %d GPU(s)"
                                   % (2 if world == 1 else 3, args.frames, world),
                       "width": W, "height": H, "frames": args.frames, "pairs": n_pairs_total, "k": K_CLUSTERS,
                       "lloyd_iters": int(n_iter), "flow_batch_pairs": pipe.batch,
                       "flow_only_mpx_s": n_pairs_total * W * H / 1e6 / dt_flow,
                       "parallelism": "frames sharded x%d, RCCL all-reduce of k*(d+1)+1 f64 per Lloyd iteration" % world},
        }
        # ---- roofline leg: the polyexp kernel, HIP events on its own stream, 64 distinct 1080p images ----
        n_img, iters = 64, 20
        ms = stages.bench_polyexp(W, H, n_img, iters, 0, device)
        achieved = POLYEXP_BYTES_PER_PX * W * H * n_img / (ms * 1e-3) / 1e9
        out["roofline"] = {"kernel": "k_polyexp", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(),
                           "launch_ms": ms, "images_per_launch": n_img,
                           "algorithmic_bytes_per_launch": POLYEXP_BYTES_PER_PX * W * H * n_img}
        # ---- CPU baseline: the oracle (1 thread) on a bounded sample of the same clip ----
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(pipe, n_iter)
    pipe.close()
    if world > 1 or force_dist:
        dist.finalize()
    if rank == 0:
        print(json.dumps(out), flush=True)


def pmc_traffic():
    """HBM bytes per launch of the roofline kernel from the committed PMC passes (profiles/r01_polyexp_pmc.json:
    2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md); counters cannot be read from inside
    the benchmark process, so this is the recorded value for the same launch configuration, or null."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_polyexp_pmc.json")) as f:
            return json.load(f)["traffic_bytes_per_launch"]
    except Exception:
        return None


def cpu_baseline(pipe, n_iter, sample_pairs=16):
    """the CPU restatement of the same path (oracle/, C, single thread) on the first `sample_pairs`
    pairs of the clip: Farneback per pair, then Lloyd (same init, to convergence) over their (u,v)"""
    from oracle import oracle as O
    P = W * H
    frames = pipe.frames.download((sample_pairs + 1, H, W), np.uint8)
    t0 = time.perf_counter()
    flows = np.stack([O.farneback(frames[t], frames[t + 1]) for t in range(sample_pairs)])
    t_flow = time.perf_counter() - t0
    X = flows.reshape(-1, 2)
    t1 = time.perf_counter()
    _, _, _, it = O.kmeans_fit(X, INIT)
    t_km = time.perf_counter() - t1
    cores = 1
    return {"value": sample_pairs * P / 1e6 / (t_flow + t_km), "unit": "Mpixels/s", "cores": cores, "kind": "port",
            "sample": "first %d pairs of the clip (%d x 1080p): flow %.2f s + Lloyd k=5 %d iters %.2f s, "
                      "oracle/*.c built -O2 without -march=native, 1 thread"
                      % (sample_pairs, sample_pairs, t_flow, it, t_km),
            "host_cores_available": os.cpu_count()}


if __name__ == "__main__":
    main()
