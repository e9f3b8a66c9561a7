#!/usr/bin/env python3
"""bench.py -- headline benchmark of the flow -> k-means hot path on MI355X.

Default workload (BASELINE.json configs[2]; configs[3] when launched on N > 1 GPUs): a 300-frame synthetic
1080p clip resident in HBM -> dense Farneback flow for all 299 consecutive pairs -> Lloyd's k-means
(k=5, fixed init, max_iter=300, tol=1e-4, run to convergence) over the 6.2e8 per-pixel (u,v) vectors.
One "step" = one full pass (flow + k-means) over the clip.  With N ranks the pairs are sharded
contiguously (strong scaling: the clip is fixed) and the Lloyd partial sums are all-reduced over RCCL
once per iteration.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --workload cfg4            # configs[4] shape: 4K frames pushed from host memory, k=8 on cell flow

Prints ONE JSON line on rank 0 (metric = Mpixels/s of flow+k-means, whole job), including
  roofline     -- the polynomial-expansion kernel (24 B/px algorithmic) timed with HIP events on its own stream over 64
                  distinct resident 1080p images, against the 8 TB/s HBM peak; sub-objects `flow_iter` (the kernel that
                  dominates a step, 56 B/px), `lloyd` (the full label-less Lloyd sweep over the clip's 6.2e8 vectors, 8 B/point,
                  with the pruned tile sweep that replaces it beside it) and `pipeline` (whole flow, staged 478 B/px and
                  fused 267 B/px models)
  cpu_baseline -- the CPU oracle timed on a bounded sample of the same clip: single thread and all host cores, plus
                  scikit-learn's own KMeans when importable.
"""
import argparse
import hashlib
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
FLOW_ITER_BYTES_PER_PX = 56   # SURVEY.md 8d fused iteration: R0 20 + R1 20 (gathered) + flow 8 in + 8 out
STAGED_BYTES_PER_PX = 478     # whole Farneback per full-res pixel, staged kernels (the model priced against)
FUSED_BYTES_PER_PX = 267      # M never stored, R computed once per frame
LLOYD_BYTES_PER_POINT = 8     # label-less sweep of the f32 (u,v) stream (SURVEY.md 8d prices 10: + 1 B label read + 1 B written)
PMC_JSON = "r03_pmc.json"     # profiles/: HBM traffic per launch from the PMC passes of tools/roofline_pmc.sh
MAX_BATCH = 32


def auto_batch(n_pairs, max_batch=MAX_BATCH):
    """pairs per flow launch sequence.  A shard of a few batches is split evenly (38 pairs run 19 + 19 on the two
    engines, never 32 + 6: measured 7.9 ms against 8.3 for one 38-pair batch); a longer clip takes full 32-pair batches
    and one short tail (299 pairs: 46.8 ms with 9 x 32 + 11 against 47.2 with 10 x 30); a clip of 256 pairs or more takes
    64-pair batches (round 3: 36.5-36.9 against 37.5 ms per step -- at level 0 a 64-pair launch is exactly one work-group
    per pair and column tile, 512 = one full round of the chip, each marching the whole frame height with a single 14-row
    window warm-up; 48, 60, 75, 80, 100, 128, 150 pairs and a third engine were all slower)"""
    if n_pairs >= 256:
        return 2 * max_batch
    n_batches = -(-n_pairs // max_batch)
    return max_batch if n_batches >= 4 else -(-n_pairs // n_batches)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=0, help="frame pairs per flow launch sequence (0 = equal batches <= 32)")
    ap.add_argument("--frames", type=int, default=CLIP_FRAMES, help="clip length (default = the named config)")
    ap.add_argument("--engines", type=int, default=2, help="flow engines (HIP streams) fed round-robin")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extras", action="store_true", help="skip the informational legs outside the timed region")
    ap.add_argument("--workload", choices=("cfg2", "cfg4"), default="cfg2")
    ap.add_argument("--frames4k", type=int, default=1000, help="frames pushed per step of --workload cfg4 (configs[4]: 1000)")
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
    if os.environ.get("OFC_BENCH_DEVICE"):            # rehearsal of N > 1 on fewer GPUs (with OFC_DIST_TRANSPORT=gloo: RCCL
        device = int(os.environ["OFC_BENCH_DEVICE"])  # refuses two ranks on one device)
    rank, world, barrier, allreduce_max = dist.init_from_torch_env(device)
    if args.workload == "cfg4":
        out = bench_cfg4(args, device, rank, world, barrier, allreduce_max)
        if world > 1 or force_dist:
            barrier()
            dist.finalize()
        if rank == 0:
            print(json.dumps(out), flush=True)
        return

    n_pairs_total = args.frames - 1
    p0, p1 = shard_pairs(n_pairs_total, world, rank)
    batch = args.batch if args.batch > 0 else auto_batch(p1 - p0)
    if os.environ.get("OFC_BENCH_SCHEDULE"):          # experiment: an explicit schedule of batch sizes, e.g. "64,64,64,64,32,11"
        batch = [int(v) for v in os.environ["OFC_BENCH_SCHEDULE"].split(",")]
    pipe = ClipPipeline(W, H, p1 - p0 + 1, batch_pairs=batch, device=device, n_engines=args.engines)
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
                       "lloyd_iters": int(n_iter), "flow_batch_pairs": pipe.batch, "flow_batch_schedule": list(pipe.schedule),
                       "centers": [[float(v) for v in row] for row in centers], "inertia": float(inertia),
                       "parallelism": "frames sharded x%d, all-reduce of k*(d+1)+1 f64 per Lloyd iteration, transport %s"
                                      % (world, {"rccl": "RCCL", "gloo-host": "gloo (host fallback)", "none": "none (one rank)"}[dist.TRANSPORT])},
        }
    def guarded(slot, fn):
        """an informational leg must never cost the headline line that was already measured"""
        try:
            return fn()
        except Exception as e:            # noqa: BLE001
            import traceback
            traceback.print_exc()
            return {"error": "%s in %s: %s" % (e.__class__.__name__, slot, e)}

    if rank == 0:
        from opticalflowclustering_amd.cluster import prune_stats
        ps = prune_stats(device)
        out["config"]["lloyd_skip_fraction"] = ps["skip_fraction"]
        out["config"]["lloyd_sweeps"] = {"tile_sweeps": ps["tile_sweeps"], "pruned": ps["pruned_sweeps"], "probes": ps["probe_sweeps"],
                                         "what": "label-less Lloyd iterations over 64-sample tiles; a pruned sweep reads only the "
                                                 "tiles whose (u,v) box is not inside one Voronoi cell (skip fraction = tiles not read)"}
    if not args.no_extras:
        # ---- informational legs, outside the timed region ----
        # (1) the flow alone over the same resident clip (the configs[1] shape: many distinct 1080p pairs per launch sequence)
        barrier()
        tf = time.perf_counter()
        for _ in range(2):
            pipe.run_flow(sync=True)
        barrier()
        dt_flow = allreduce_max(time.perf_counter() - tf) / 2
        if rank == 0:
            out["config"]["flow_only_mpx_s"] = n_pairs_total * W * H / 1e6 / dt_flow
        if world == 1 and not force_dist:
            # (2) the pruned fit must be the unpruned one (same n_iter, same labels, centres <= 1e-9), and the labelled,
            # host-driven fit (sharded.fit_sharded over a DeviceShard: reads and writes labels every iteration) over the
            # resident flows must land where the label-less in-library fit did
            out["config"]["pruned_fit_check"] = guarded("pruned_fit_check", lambda: pruned_fit_check(pipe, centers, n_iter, inertia))
            out["config"]["labelled_fit_check"] = guarded("labelled_fit_check", lambda: labelled_fit_check(pipe, centers, n_iter, device))
            lloyd_roof = guarded("roofline.lloyd", lambda: lloyd_roofline(pipe, centers, device))
    pipe.close()
    if rank == 0 and world == 1 and not args.no_extras and args.frames == CLIP_FRAMES:
        # (3) what one rank of an 8-GPU run does per step, alone on this GPU (no collective: RCCL refuses two ranks on one
        # device) -- the numbers the >= 6x scaling target is priced on
        out["config"].update(guarded("shard_step_ms", lambda: shard_step_ms(device, args.engines, centers, int(n_iter))))
    if rank == 0:
        out["roofline"] = guarded("roofline", lambda: kernel_rooflines(device, out["config"].get("flow_only_mpx_s")))
        if world == 1 and not force_dist and not args.no_extras and isinstance(out["roofline"], dict):
            out["roofline"]["lloyd"] = lloyd_roof
        # ---- CPU baseline: the oracle on a bounded sample of the same clip ----
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = guarded("cpu_baseline", lambda: cpu_baseline(device))
    if world > 1 or force_dist:
        barrier()            # rank 0 ran the roofline legs meanwhile: tear the communicator down together
        dist.finalize()
    if rank == 0:
        print(json.dumps(out), flush=True)


def kernel_rooflines(device, flow_only_mpx_s):
    """roofline leg: the polyexp kernel, HIP events on its own stream, 64 distinct 1080p images; then the flow iteration"""
    from opticalflowclustering_amd import stages
    n_img, iters = 64, 20
    ms = stages.bench_polyexp(W, H, n_img, iters, 0, device)
    achieved = POLYEXP_BYTES_PER_PX * W * H * n_img / (ms * 1e-3) / 1e9
    roof = {"kernel": "k_polyexp", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic("k_polyexp"),
            "launch_ms": ms, "images_per_launch": n_img,
            "algorithmic_bytes_per_launch": POLYEXP_BYTES_PER_PX * W * H * n_img}
    # the kernel that dominates a step: one level-0 Farneback iteration over 32 resident pairs
    n_pairs_l = 32
    ms_it = stages.bench_flow_iters(W, H, n_pairs_l, 10, 0, device) / 2
    alg = FLOW_ITER_BYTES_PER_PX * W * H * n_pairs_l
    roof["flow_iter"] = {"kernel": "k_flow_iter<7,0>", "bound": "hbm",
                         "achieved": alg / (ms_it * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg / (ms_it * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "traffic": pmc_traffic("k_flow_iter"), "launch_ms": ms_it,
                         "pairs_per_launch": n_pairs_l, "algorithmic_bytes_per_launch": alg}
    if flow_only_mpx_s:
        f = flow_only_mpx_s * 1e6
        roof["pipeline"] = {
            "what": "whole Farneback flow (all levels, all kernels) over the resident clip, per full-res pixel",
            "model": "staged (SURVEY.md 8d: 478 B/px, M stored, every kernel separate) is the one priced; the fused "
                     "model (267 B/px: M never stored, R computed once per frame) is what the engine's kernels move",
            "staged_bytes_per_px": STAGED_BYTES_PER_PX, "fused_bytes_per_px": FUSED_BYTES_PER_PX,
            "achieved_staged": f * STAGED_BYTES_PER_PX / 1e9,
            "model_ratio_staged": f * STAGED_BYTES_PER_PX / 1e9 / HBM_PEAK_GBS,
            "model_ratio_staged_note": "NOT a fraction of peak that was moved: the staged model prices traffic (M, per-kernel "
                                       "R re-reads) that the fused engine never generates, so this ratio may exceed 1",
            "achieved_fused": f * FUSED_BYTES_PER_PX / 1e9, "frac_fused": f * FUSED_BYTES_PER_PX / 1e9 / HBM_PEAK_GBS,
            "peak": HBM_PEAK_GBS, "unit": "GB/s"}
    return roof


def pruned_fit_check(pipe, centers, n_iter, inertia):
    """the fit bench.py timed (tile sweeps, pruned) against the same fit with every sweep full (OFC_LLOYD_PRUNE=0) over
    the resident 6.2e8 vectors: n_iter equal, labels bit-equal, centres <= 1e-9 (VERDICT r02 #1's done-when)"""
    lab_pruned = pipe.labels_host()
    old = os.environ.get("OFC_LLOYD_PRUNE")
    os.environ["OFC_LLOYD_PRUNE"] = "0"
    try:
        c0, in0, it0 = pipe.run_kmeans(INIT, max_iter=300, tol=1e-4)
    finally:
        if old is None:
            os.environ.pop("OFC_LLOYD_PRUNE", None)
        else:
            os.environ["OFC_LLOYD_PRUNE"] = old
    same_labels = bool(np.array_equal(lab_pruned, pipe.labels_host()))
    dc = float(np.abs(c0 - centers).max())
    return {"ok": bool(int(it0) == int(n_iter) and same_labels and dc <= 1e-9), "n_iter": [int(n_iter), int(it0)],
            "labels_equal": same_labels, "max_abs_centre_diff": dc, "inertia_rel_diff": float(abs(in0 - inertia) / in0)}


def lloyd_roofline(pipe, centers, device):
    """the Lloyd sweeps over the clip's resident vectors, HIP events on the Lloyd stream, centres fixed at the converged ones"""
    from opticalflowclustering_amd import stages
    N = pipe.n_pairs * W * H
    from opticalflowclustering_amd import _lib
    colsum = np.zeros(2)
    _lib.check(_lib.load().ofc_lloyd_colstats_dev(device, pipe.flows.ptr, _lib.F32, N, 2, None, 0, _lib.ptr(colsum)))
    mean = colsum / N
    ms = {name: stages.bench_lloyd_sweep(pipe.flows.ptr, N, centers, mean, what, 10, device)
          for name, what in (("full", 0), ("pruned", 1), ("meta", 2), ("final", 3), ("final_pruned", 4))}
    alg = LLOYD_BYTES_PER_POINT * N
    return {"kernel": "k_lloyd_assign<2,5,float,3>", "bound": "hbm", "achieved": alg / (ms["full"] * 1e-3) / 1e9,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / (ms["full"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "traffic": pmc_traffic("k_lloyd_assign"), "launch_ms": ms["full"], "points_per_launch": N,
            "algorithmic_bytes_per_launch": alg,
            "frac_at_10_bytes_per_point": 10 * N / (ms["full"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "note": "the full label-less sweep (8 B/point; SURVEY.md 8d's 10 B/point form reads and writes a label byte as "
                    "well).  The default fit runs it never: one streaming pass builds the tile metadata (`meta_ms`: full read + "
                    "40 B written per 64 samples), every iteration is a pruned sweep (`pruned_ms`), the final E-step "
                    "`final_estep_pruned_ms` (`final_estep_ms` = its full form, every sample read)",
            "pruned_ms": ms["pruned"], "meta_ms": ms["meta"], "final_estep_ms": ms["final"],
            "final_estep_pruned_ms": ms["final_pruned"],
            "pruned_traffic": pmc_traffic("k_lloyd_tiles_pruned")}


def labelled_fit_check(pipe, centers, n_iter, device):
    from opticalflowclustering_amd import _lib
    from opticalflowclustering_amd.sharded import DeviceShard, fit_sharded
    N = pipe.n_pairs * W * H
    shard = DeviceShard(pipe.flows.ptr, _lib.F32, N, 2, pipe.labels.ptr, device)
    c2, _, it2 = fit_sharded(shard, INIT, max_iter=300, tol=1e-4)
    ok = int(it2) == int(n_iter) and float(np.abs(c2 - centers).max()) <= 1e-9
    return {"ok": bool(ok), "n_iter": [int(n_iter), int(it2)], "max_abs_centre_diff": float(np.abs(c2 - centers).max())}


def shard_step_ms(device, engines, global_centers, global_iters, steps=10):
    """rank 0's share of configs[3]: pairs [0, 38) of the clip.
    shard_step_ms       flow + a self-contained Lloyd fit over the shard's own vectors (its own iteration count; on this
                        clip the shard alone meets an empty cluster and takes the labelled path)
    shard_rank_step_ms  flow + exactly the device work the rank does inside the 8-GPU fit: the clip-wide fit's number of
                        label-less sweeps over its 38 pairs (started from the converged clip-wide centres, with a negative tol so
                        that exactly max_iter = that many run), column statistics and the final E-step: everything but the collectives"""
    from opticalflowclustering_amd import _lib
    from opticalflowclustering_amd.pipeline import ClipPipeline, shard_pairs
    p0, p1 = shard_pairs(CLIP_FRAMES - 1, 8, 0)
    pipe = ClipPipeline(W, H, p1 - p0 + 1, batch_pairs=auto_batch(p1 - p0), device=device, n_engines=engines)
    pipe.synth(t0=p0, seed=0)
    lib = _lib.load()
    res = {}
    for key, init, kw in (("shard_step_ms", INIT, dict(max_iter=300, tol=1e-4)),
                          ("shard_rank_step_ms", np.asarray(global_centers), dict(max_iter=global_iters, tol=-1.0))):
        for _ in range(2):
            pipe.run_flow(sync=False)
            r = pipe.run_kmeans(init, **kw)
        _lib.check(lib.ofc_device_sync(device))
        t0 = time.perf_counter()
        for _ in range(steps):
            pipe.run_flow(sync=False)
            r = pipe.run_kmeans(init, **kw)
        _lib.check(lib.ofc_device_sync(device))
        res[key] = (time.perf_counter() - t0) / steps * 1e3
        res[key.replace("_ms", "_lloyd_iters")] = int(r[2])
    pipe.close()
    return res


def source_sha16():
    """fingerprint of the kernel sources a recorded PMC pass belongs to"""
    h = hashlib.sha256()
    for name in ("flow_kernels.hip", "lloyd_kernels.hip", "lloyd_tiles.hip"):
        with open(os.path.join(ROOT, "opticalflowclustering_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/r03_pmc.json: 2 x FETCH_SIZE + WRITE_SIZE, the
    gfx950 correction of MI355X_MICROARCH.md).  Counters cannot be read from inside the benchmark process, so this is the
    value recorded for the same launch configuration -- and only while the kernel source is the one the passes were
    collected on (sha of the kernel sources recorded beside them); otherwise null."""
    try:
        with open(os.path.join(ROOT, "profiles", PMC_JSON)) as f:
            rec = json.load(f)
        if rec.get("source_sha16") != source_sha16():
            return None
        return rec["kernels"][kernel]["traffic_bytes_per_launch"]
    except Exception:
        return None


def cpu_baseline(device, single_pairs=4, multi_pairs=32):
    """the CPU restatement of the same path (oracle/*.c) on the first pairs of the same clip, built -O3 -march=native on
    this host: (a) ONE thread, `single_pairs` pairs (how OpenCV's serial row loops and a 1-thread sklearn would run);
    (b) ALL host cores this process may use, `multi_pairs` pairs: Farneback per pair on a thread pool (pairs are
    independent), Lloyd with the samples split over the threads and the partial sums added per iteration (the shape of
    sklearn's OpenMP chunks); (c) scikit-learn's own KMeans on the same vectors when it is importable here."""
    from concurrent.futures import ThreadPoolExecutor
    from opticalflowclustering_amd.pipeline import ClipPipeline
    from opticalflowclustering_amd.sharded import fit_sharded
    from oracle import oracle as O
    P = W * H
    pipe = ClipPipeline(W, H, multi_pairs + 1, batch_pairs=multi_pairs, device=device, n_engines=1)
    pipe.synth(t0=0, seed=0)
    frames = pipe.frames.download((multi_pairs + 1, H, W), np.uint8)
    pipe.close()
    quota = cpu_quota()
    # threads = the CPUs this process can really use: its affinity mask, capped by the cgroup's CPU quota when there is one
    # (a GPU box grants a 16-CPU share of its 256 cores: 64 threads on 16 CPUs only oversubscribe)
    cores = max(1, min(len(os.sched_getaffinity(0)), int(np.ceil(quota)) if quota else 64))
    N = O.native()                                      # -O3 -march=native build of the same C files, made here
    # ---- (a) one thread ----
    t0 = time.perf_counter()
    flows1 = np.stack([N.farneback(frames[t], frames[t + 1]) for t in range(single_pairs)])
    t_flow1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    _, _, _, it1 = N.kmeans_fit(flows1.reshape(-1, 2), INIT)
    t_km1 = time.perf_counter() - t0
    single = {"value": single_pairs * P / 1e6 / (t_flow1 + t_km1), "unit": "Mpixels/s", "cores": 1,
              "sample": "first %d pairs: flow %.2f s + Lloyd k=5 %d iterations %.2f s" % (single_pairs, t_flow1, it1, t_km1)}
    # ---- (b) all cores ----
    with ThreadPoolExecutor(cores) as pool:
        cpu0 = time.process_time()
        t0 = time.perf_counter()
        flows = np.stack(list(pool.map(lambda t: N.farneback(frames[t], frames[t + 1]), range(multi_pairs))))
        t_flow = time.perf_counter() - t0
        X = flows.reshape(-1, 2)
        shard = ThreadedOracleShard(X, cores, pool, N)
        t0 = time.perf_counter()
        _, _, it = fit_sharded(shard, INIT, max_iter=300, tol=1e-4)
        t_km = time.perf_counter() - t0
        cpu_used = time.process_time() - cpu0
    allc = {"value": multi_pairs * P / 1e6 / (t_flow + t_km), "unit": "Mpixels/s", "cores": cores,
            "cpu_seconds_per_wall_second": cpu_used / (t_flow + t_km),
            "sample": "first %d pairs: flow %.2f s + Lloyd k=5 %d iterations %.2f s, %d threads" % (multi_pairs, t_flow, it, t_km, cores)}
    # ---- (c) the reference's own Lloyd ----
    try:
        import warnings
        from sklearn.cluster import KMeans
        Xs = X[: 8 * P]
        t0 = time.perf_counter()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            km = KMeans(n_clusters=K_CLUSTERS, init=INIT, n_init=1, max_iter=300, tol=1e-4).fit(Xs)
        t_sk = time.perf_counter() - t0
        import sklearn
        sk = {"value": len(Xs) * km.n_iter_ / 1e6 / t_sk, "unit": "Mpoint-iterations/s (Lloyd only, all cores)",
              "seconds": t_sk, "n_iter": int(km.n_iter_), "version": sklearn.__version__,
              "sample": "the %d (u,v) vectors of the first 8 pairs, f32 as the flow produces them; for comparison the "
                        "oracle's all-core Lloyd above runs %.0f Mpoint-iterations/s" % (len(Xs), len(X) * it / 1e6 / t_km)}
    except Exception as e:                              # not importable on this box
        sk = "unavailable on this box (%s)" % e.__class__.__name__
    return {"value": allc["value"], "unit": "Mpixels/s", "cores": cores, "kind": "port",
            "sample": allc["sample"] + "; oracle/*.c built -O3 -march=native on this host",
            "single_thread": single, "all_cores": allc, "sklearn_kmeans": sk, "host_cores_available": os.cpu_count(),
            "cpu_quota_cores": quota,
            "cores_note": "`cores` = threads started (min(affinity, cgroup quota; 64 without a quota)); `cpu_quota_cores` = CPU time per wall second this "
                          "process's cgroup grants (cpu.max; null = no limit readable); `cpu_seconds_per_wall_second` in "
                          "all_cores = what the threaded leg actually got"}


def cpu_quota():
    """CPU share of this process's cgroup in cores (cgroup v2 cpu.max, v1 cfs quota), None when unlimited / unreadable"""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()
        return None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            per = float(f.read())
        return None if q <= 0 else q / per
    except Exception:
        return None


class ThreadedOracleShard:
    """the four shard-local passes of sharded.fit_sharded on the CPU oracle, rows split over a thread pool"""

    def __init__(self, X, n, pool, N):
        self.X, self.pool, self.Nlib = X, pool, N
        self.N, self.d = X.shape
        self.cuts = [(self.N * i) // n for i in range(n + 1)]
        self.labels = np.full(self.N, -1, np.int32)

    def _map(self, fn):
        return list(self.pool.map(lambda i: fn(self.cuts[i], self.cuts[i + 1]), range(len(self.cuts) - 1)))

    def colstats(self, mean, pass_):
        if pass_ == 0:
            return np.sum(self._map(lambda a, b: self.X[a:b].astype(np.float64).sum(0)), 0)
        return np.sum(self._map(lambda a, b: ((self.X[a:b].astype(np.float64) - mean) ** 2).sum(0)), 0)

    def step(self, mean, centers_c, accumulate=True):
        return np.sum(self._map(lambda a, b: self.Nlib.lloyd_partials(self.X[a:b], mean, centers_c, self.labels[a:b])), 0)

    def inertia(self, mean, centers_c):
        return float(np.sum(self._map(lambda a, b: ((self.X[a:b].astype(np.float64) - mean - centers_c[self.labels[a:b]]) ** 2).sum())))

    def farthest(self, mean, centers_c, excl):
        Xc = self.X.astype(np.float64) - mean
        d2 = ((Xc - centers_c[self.labels]) ** 2).sum(1)
        d2[list(excl)] = -2
        i = int(np.argmax(d2))
        return float(d2[i]), i, Xc[i], int(self.labels[i])


def bench_cfg4(args, device, rank=0, world=1, barrier=lambda: None, allreduce_max=lambda v: v):
    """BASELINE.json configs[4]: a 4K stream of `--frames4k` frames pushed one by one from host memory (as a decoder hands
    them over) through the pinned double-buffered hipMemcpyAsync ingest, Farneback per pair, reduced on the device to the
    14x25 grid-cell averaged flow (KmeanGrids' grid), then Lloyd k=8 over all cell vectors.  PCIe-inclusive by
    construction.  With N ranks the stream is cut into contiguous pair ranges (each rank also pushes its one halo frame)
    and the Lloyd fit over the cell vectors is the in-library one: every rank holds its own vectors, one all-reduce per
    iteration."""
    from opticalflowclustering_amd import _lib, synth
    from opticalflowclustering_amd.cluster import kmeans_fit_dev, seeded_rows_init
    from opticalflowclustering_amd.pipeline import shard_pairs
    from opticalflowclustering_amd.stream import FlowStream
    W4, H4, n = 3840, 2160, args.frames4k
    p = synth.texture_params(0)
    # eight vertical bands moving with eight different velocities (so that k = 8 has something to find); the stream walks
    # the eight distinct frames back and forth (0..7,6..1,0..): every consecutive pair is one motion step
    base = [synth.frame(W4, H4, *synth.population_motion(W4, H4, t, n_pop=8, seed=4)[:2], p).astype(np.uint8) for t in range(8)]
    order = list(range(8)) + list(range(6, 0, -1))
    p0, p1 = shard_pairs(n - 1, world, rank)                  # this rank's pairs [p0, p1): frames p0 .. p1
    fs = FlowStream(W4, H4, batch_pairs=8, device=device)
    buf = _lib.DeviceBuffer(max(p1 - p0, 1) * 350 * 8, device)
    # the same initial centres on every rank: 8 distinct rows of the first 14 pairs' cell vectors (those pairs cycle with
    # period 14, so every rank can compute them from frames it has anyway)
    warm = FlowStream(W4, H4, batch_pairs=8, device=device)
    for t in range(15):
        warm.push(base[order[t % len(order)]])
    init = seeded_rows_init(warm.finish().reshape(-1, 2), 8, 0)
    warm.close()

    def step():
        for t in range(p0, p1 + 1):
            fs.push(base[order[t % len(order)]])
        cells = fs.finish()
        buf.upload(cells)
        return cells, kmeans_fit_dev(buf.ptr, _lib.F32, cells.shape[0] * 350, 2, init, device=device)

    for _ in range(max(args.warmup, 1)):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        cells, (centers, inertia, n_iter) = step()
    barrier()
    dt = allreduce_max(time.perf_counter() - t0)
    fs.close()
    buf.free()
    from opticalflowclustering_amd import dist
    return {"metric": "Mpixels/s dense flow+kmeans @4K stream", "value": args.steps * (n - 1) * W4 * H4 / 1e6 / dt,
            "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 1),
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32 storage, f64 accumulation", "data": "synthetic, pushed from host memory (PCIe-inclusive)",
            "config": {"workload": "BASELINE.json configs[4]: %d-frame 4K stream sharded over %d GPU(s), pinned double-buffered "
                                   "ingest, Farneback, 14x25 grid-cell averaged flow, Lloyd k=8 over the cell vectors" % (n, world),
                       "width": W4, "height": H4, "frames": n, "pairs": n - 1, "k": 8, "lloyd_iters": int(n_iter),
                       "cell_vectors": int((n - 1) * 350), "ms_per_frame": 1e3 * dt / args.steps / n,
                       "upload_bytes_per_frame": W4 * H4, "centers": [[float(v) for v in row] for row in centers],
                       "inertia": float(inertia), "transport": dist.TRANSPORT}}


if __name__ == "__main__":
    main()
