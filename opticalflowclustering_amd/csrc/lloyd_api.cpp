// lloyd_api.cpp -- C ABI of libofc.so, part 2: Lloyd k-means drivers (streaming shape).
// Control flow = sklearn's _kmeans_single_lloyd (_kmeans.py:624-752) around the kernels of
// lloyd_kernels.hip; see include/ofc.h for the reference call sites.
#include "lloyd_common.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <mutex>

namespace ofc {

// work-groups of a streaming sweep: enough lanes to keep the memory system full, few enough that the fixed-order record
// reduction stays short (it is a visible share of an iteration on a 1/8 shard)
static int lloyd_grid(int64_t N)
{
    return (int)std::max<int64_t>(1, std::min<int64_t>(cdiv64(N / 4, 256), N >= (200ll << 20) ? 2048 : 1024));
}

static size_t dtype_size(int dtype) { return dtype == OFC_U8 ? 1 : (dtype == OFC_F32 ? 4 : 8); }

// Per-device scratch, created once and reused by every fit: creating/destroying a HIP stream (1.5-6 ms) and
// pinned memory per call dominated small fits (rocprof hip-trace of a 38-frame shard).
struct LloydScratch {
    DevBuf state, partial, tot, tot_local, excl, far, labels;
    DevBuf tile_box, tile_sum, tile_sq;  // lloyd_tiles.hip: 16 + 16 + 8 B per 64-sample tile, rebuilt by every fit (k_tile_meta)
    DevBuf tile_meta;                    // k_tile_meta's reduced record (4 doubles)
    double prune_stats[6] = {0, 0, 0, 0, 0, 0};   // of the last fit, see ofc_lloyd_prune_stats
    LloydStatus *status = nullptr;       // pinned, device-visible; one slot per iteration of a window
    LloydStatus *status_dev = nullptr;
    hipStream_t stream = nullptr;
    bool ready = false;
    std::mutex mu;                       // fits on one device share this scratch: serialised
    int init()
    {
        if (ready) return OFC_OK;
        constexpr int NVMAX = LLOYD_NVMAX;
        OFC_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        OFC_TRY(state.alloc(sizeof(LloydState)));
        OFC_TRY(partial.alloc(sizeof(double) * 2048 * NVMAX));
        OFC_TRY(tot.alloc(sizeof(double) * (NVMAX + 8)));
        OFC_TRY(tot_local.alloc(sizeof(double) * (NVMAX + 8)));
        OFC_TRY(excl.alloc(sizeof(int64_t) * LLOYD_KMAX));
        OFC_TRY(far.alloc(sizeof(double) * 2 * 2048));
        OFC_TRY(tile_meta.alloc(sizeof(double) * 8));
        OFC_HIP(hipHostMalloc((void **)&status, sizeof(LloydStatus) * LLOYD_WINDOW, hipHostMallocMapped));
        OFC_HIP(hipHostGetDevicePointer((void **)&status_dev, status, 0));
        ready = true;
        return OFC_OK;
    }
};

// OFC_LLOYD_PRUNE: 0 = off (every sweep is k_lloyd_assign's), 1 = auto (default: tile sweeps for f32 d=2 k<=8 streams of
// >= 2^20 samples, pruning switched on and off by the device-side policy in k_lloyd_update), 2 = as auto for any N,
// 3 = every tile sweep after the first runs pruned whatever the share of tiles that pass (tests: worst cases)
static int prune_policy_for(int dtype, int64_t N, int d, int k)
{
    if (!lloyd_tiles_supported(dtype, d, k) || N < 64) return LLOYD_PRUNE_OFF;
    const char *e = getenv("OFC_LLOYD_PRUNE");
    const int v = (e && e[0] >= '0' && e[0] <= '3' && !e[1]) ? e[0] - '0' : LLOYD_PRUNE_AUTO;
    if (v == LLOYD_PRUNE_AUTO && N < (1ll << 20)) return LLOYD_PRUNE_OFF;
    return v;
}

static LloydScratch &scratch_for(int device)
{
    static LloydScratch *pool = new LloydScratch[16];   // intentionally never destroyed (no HIP calls at exit)
    return pool[device & 15];
}

// empty-cluster relocation (_k_means_common.pyx:167-211) on the all-reduced totals `tot`.
// Returns with `tot` patched on the device (or untouched when the farthest distance is 0).
static int relocate_empty(LloydScratch &sc, const void *X, int dtype, int64_t N, int d, int k, int kmax,
                          int nblocks, const uint8_t *labels, const double *mean_h)
{
    hipStream_t s = sc.stream;
    const int NV = lloyd_record_len(kmax, d);
    std::vector<double> tot(NV);
    OFC_HIP(hipMemcpyAsync(tot.data(), sc.tot.p, sizeof(double) * NV, hipMemcpyDeviceToHost, s));
    OFC_HIP(hipStreamSynchronize(s));
    double *w = tot.data() + kmax * d;
    LloydState *st = sc.state.as<LloydState>();
    const double *c_old = st->centers;          // device address of the member array
    std::vector<int64_t> excl;
    std::vector<double> blk(2 * nblocks);
    bool first = true;
    for (int j = 0; j < k; j++) {
        if (w[j] != 0.0) continue;
        if (!excl.empty())
            OFC_HIP(hipMemcpyAsync(sc.excl.p, excl.data(), sizeof(int64_t) * excl.size(), hipMemcpyHostToDevice, s));
        OFC_TRY(launch_lloyd_farthest(X, dtype, N, d, st, c_old, labels, sc.excl.as<int64_t>(), (int)excl.size(),
                                      sc.far.as<double>(), nblocks, s));
        OFC_HIP(hipMemcpyAsync(blk.data(), sc.far.p, sizeof(double) * 2 * nblocks, hipMemcpyDeviceToHost, s));
        OFC_HIP(hipStreamSynchronize(s));
        double best = -1;
        int64_t bi = -1;
        for (int b = 0; b < nblocks; b++) {
            const double v = blk[2 * b];
            const int64_t i = (int64_t)blk[2 * b + 1];
            if (i < 0) continue;
            if (v > best || (v == best && i < bi)) { best = v; bi = i; }
        }
        // global winner across ranks: max distance, ties -> lowest rank (= lowest global index)
        double rec[LLOYD_DMAX + 3];   // [dist | rank-or-inf | x_c[d] | old label]
        int owner = 1;
        if (dist_active()) {
            double *dv = sc.far.as<double>();
            double g = best;
            OFC_HIP(hipMemcpyAsync(dv, &g, sizeof(double), hipMemcpyHostToDevice, s));
            OFC_TRY(dist_allreduce_f64(dv, 1, DIST_MAX, s));
            OFC_HIP(hipMemcpyAsync(&g, dv, sizeof(double), hipMemcpyDeviceToHost, s));
            OFC_HIP(hipStreamSynchronize(s));
            double r = (bi >= 0 && best == g) ? (double)dist_rank() : 1e300;
            OFC_HIP(hipMemcpyAsync(dv, &r, sizeof(double), hipMemcpyHostToDevice, s));
            OFC_TRY(dist_allreduce_f64(dv, 1, DIST_MIN, s));
            OFC_HIP(hipMemcpyAsync(&r, dv, sizeof(double), hipMemcpyDeviceToHost, s));
            OFC_HIP(hipStreamSynchronize(s));
            owner = ((int)r == dist_rank()) && bi >= 0 && best == g;
            best = g;
        }
        if (first && !(best > 0)) return OFC_OK;   // np.max(distances) == 0: relocating is pointless
        first = false;
        memset(rec, 0, sizeof(rec));
        if (owner) {
            unsigned char raw[LLOYD_DMAX * 8];
            uint8_t lab;
            OFC_HIP(hipMemcpyAsync(raw, (const char *)X + (size_t)bi * d * dtype_size(dtype), d * dtype_size(dtype),
                                   hipMemcpyDeviceToHost, s));
            OFC_HIP(hipMemcpyAsync(&lab, labels + bi, 1, hipMemcpyDeviceToHost, s));
            OFC_HIP(hipStreamSynchronize(s));
            for (int f = 0; f < d; f++) {
                double v = dtype == OFC_U8 ? (double)raw[f]
                         : dtype == OFC_F32 ? (double)((float *)raw)[f] : ((double *)raw)[f];
                rec[f] = v - mean_h[f];
            }
            rec[d] = (double)lab;
            excl.push_back(bi);
        }
        if (dist_active()) {
            double *dv = sc.far.as<double>();
            OFC_HIP(hipMemcpyAsync(dv, rec, sizeof(double) * (d + 1), hipMemcpyHostToDevice, s));
            OFC_TRY(dist_allreduce_f64(dv, d + 1, DIST_BCAST, s));
            OFC_HIP(hipMemcpyAsync(rec, dv, sizeof(double) * (d + 1), hipMemcpyDeviceToHost, s));
            OFC_HIP(hipStreamSynchronize(s));
        }
        const int old = (int)rec[d];
        for (int f = 0; f < d; f++) {
            tot[old * d + f] -= rec[f];
            tot[j * d + f] = rec[f];
        }
        w[j] = 1.0;
        w[old] -= 1.0;
    }
    OFC_HIP(hipMemcpyAsync(sc.tot.p, tot.data(), sizeof(double) * NV, hipMemcpyHostToDevice, s));
    OFC_HIP(hipStreamSynchronize(s));
    return OFC_OK;
}

static int lloyd_fit_dev(int device, const void *X, int dtype, int64_t N, int d, int k, const double *init,
                         int max_iter, double tol_rel, double *centers, uint8_t *labels_dev,
                         double *inertia, int *n_iter, const double *colsum = nullptr)
{
    OFC_REQUIRE(X && init && centers, "null pointer");
    OFC_REQUIRE(dtype >= OFC_U8 && dtype <= OFC_F64, "bad dtype %d", dtype);
    OFC_REQUIRE(d >= 1 && k >= 1 && max_iter >= 1 && N >= 0, "bad shape");
    if (d > LLOYD_DMAX || k > LLOYD_KMAX) {
        set_error("k=%d, d=%d outside the kernels' range (k <= %d, d <= %d)", k, d, LLOYD_KMAX, LLOYD_DMAX);
        return OFC_EUNSUPPORTED;
    }
    OFC_TRY(ensure_device(device));
    const int kmax = lloyd_kmax(k);
    const int NV = lloyd_record_len(kmax, d);
    LloydScratch &sc = scratch_for(device);
    std::lock_guard<std::mutex> lock(sc.mu);
    OFC_TRY(sc.init());
    hipStream_t s = sc.stream;
    const int nblocks = lloyd_grid(N);
    if (!labels_dev) {
        if (sc.labels.bytes < (size_t)std::max<int64_t>(N, 1)) OFC_TRY(sc.labels.alloc((size_t)std::max<int64_t>(N, 1)));
        labels_dev = sc.labels.as<uint8_t>();
    }
    LloydState *st = sc.state.as<LloydState>();
    OFC_HIP(hipMemsetAsync(st, 0, sizeof(LloydState), s));
    double *tot = sc.tot.as<double>();

    // ---- column mean (X.mean(axis=0), _kmeans.py:1478-1484) and tol (_tolerance, :279-287) ----
    double hbuf[LLOYD_DMAX + 1], mean_h[LLOYD_DMAX];
    if (colsum) {           // the caller already has this rank's column sums (the flow kernels' epilogue): no sweep
        OFC_HIP(hipMemcpyAsync(tot, colsum, sizeof(double) * d, hipMemcpyHostToDevice, s));
    } else {
        OFC_TRY(launch_lloyd_colstats(X, dtype, N, d, st->mean, 0, sc.partial.as<double>(), nblocks, s));
        OFC_TRY(launch_reduce_records(sc.partial.as<double>(), nblocks, d, tot, s));
    }
    double nloc = (double)N;
    OFC_HIP(hipMemcpyAsync(tot + d, &nloc, sizeof(double), hipMemcpyHostToDevice, s));
    OFC_TRY(dist_allreduce_f64(tot, d + 1, DIST_SUM, s));
    OFC_HIP(hipMemcpyAsync(hbuf, tot, sizeof(double) * (d + 1), hipMemcpyDeviceToHost, s));
    OFC_HIP(hipStreamSynchronize(s));
    const double Ng = hbuf[d];
    if (Ng < (double)k) {
        set_error("n_samples=%.0f should be >= n_clusters=%d.", Ng, k);
        return OFC_EINVAL;
    }
    for (int f = 0; f < d; f++) mean_h[f] = hbuf[f] / Ng;
    OFC_HIP(hipMemcpyAsync(st->mean, mean_h, sizeof(double) * d, hipMemcpyHostToDevice, s));

    // ---- centred init ----
    double c0[LLOYD_KMAX * LLOYD_DMAX];
    for (int j = 0; j < k * d; j++) c0[j] = init[j] - mean_h[j % d];
    OFC_HIP(hipMemcpyAsync(st->centers, c0, sizeof(double) * k * d, hipMemcpyHostToDevice, s));
    const int prune = prune_policy_for(dtype, N, d, k);
    if (prune) {
        const size_t need = (size_t)(N >> 6) * 16;
        if (sc.tile_box.bytes < need) {
            OFC_TRY(sc.tile_box.alloc(need));
            OFC_TRY(sc.tile_sum.alloc(need));
            OFC_TRY(sc.tile_sq.alloc(need / 2));
        }
    }
    for (double &v : sc.prune_stats) v = 0;
    OFC_TRY(launch_lloyd_set_centers(st, k, d, s, prune));
    // ---- Lloyd iterations ----
    // sklearn stops on `labels == labels_old` (strict) before it looks at the centre shift (_kmeans.py:716-728).  While
    // no cluster is empty that test is redundant: equal labels give bit-equal sums (fixed reduction order), hence
    // centers_new == centers, shift_tot == 0 <= tol, and the tol test stops in the SAME iteration; the E-step sklearn
    // then repeats (mode 2 below) reproduces the labels.  So the iterations stream X only (mode 3: no label read or
    // write, 8 instead of 10 B/point for float2 data).  The labelled form (mode 1) takes over from the first
    // iteration that meets an empty cluster: relocation needs the labels, and a relocated centre breaks the
    // equal-labels => zero-shift argument.  That iteration itself cannot be a strict stop: equal labels would mean the
    // previous iteration had the same empty cluster.
    //
    // The convergence test runs on the device (k_lloyd_update) and the host enqueues LLOYD_WINDOW iterations per
    // synchronisation (windows of 4, 8, 16, 16, ...): an iteration behind the one that converged (or met an empty cluster)
    // finds st->halt set and does nothing.  One host round trip per window instead of one per iteration (18 us each -- a tenth of an
    // iteration on a 1/8 shard); every rank takes the same decisions because they derive from all-reduced totals.
    bool strict = false, labelled = false, stop = false;
    int tiles_next = LLOYD_TILES_FULL;   // how the device would run the next tile sweep: decides the form of the final E-step
    const bool trace = getenv("OFC_LLOYD_TRACE") != nullptr;
    int it = 0;
    const int *halt = &st->halt;
    double *tot_local = dist_has_comm() ? sc.tot_local.as<double>() : tot;
    int window = 4;                       // 4, 8, 16, 16, ...: an 11-iteration fit costs two host round trips, and a fit
                                          // that meets its empty cluster early (iteration 1-2, typically) wastes few
    while (it < max_iter && !stop) {
        const int nwin = std::min(window, max_iter - it);
        window = std::min(2 * window, LLOYD_WINDOW);
        for (int w = 0; w < nwin; w++) sc.status[w].valid = 0;
        for (int w = 0; w < nwin; w++) {
            // label-less sweeps of a (u,v) stream go tile by tile: iteration 0 builds the tile metadata, the later ones
            // run in the mode k_lloyd_update chose from the previous iteration's tile counts (lloyd_tiles.hip)
            const int tiles = (prune && !labelled) ? (it + w == 0 ? 1 : 2) : 0;
            if (tiles == 1) {      // is the field coherent enough? then tile metadata + column sums of squares, then iteration 0
                OFC_TRY(launch_lloyd_tiles((const float *)X, N, k, st, nullptr, nullptr, nullptr, nullptr,
                                           sc.partial.as<double>(), nblocks, LLOYD_WHAT_PROBE, nullptr, s));
                OFC_TRY(launch_lloyd_tiles((const float *)X, N, k, st, sc.tile_box.p, sc.tile_sum.p, sc.tile_sq.p, nullptr,
                                           sc.partial.as<double>(), nblocks, LLOYD_WHAT_META, nullptr, s));
                OFC_TRY(launch_reduce_records(sc.partial.as<double>(), nblocks, 2, sc.tile_meta.as<double>(), s));
            }
            if (tiles)
                OFC_TRY(launch_lloyd_tiles((const float *)X, N, k, st, sc.tile_box.p, sc.tile_sum.p, sc.tile_sq.p, nullptr,
                                           sc.partial.as<double>(), nblocks, LLOYD_WHAT_SWEEP,
                                           tiles == 1 ? sc.tile_meta.as<double>() : nullptr, s));
            else
                OFC_TRY(launch_lloyd_assign(X, dtype, N, d, k, st, labels_dev, sc.partial.as<double>(), nblocks,
                                            labelled ? 1 : 3, it + w == 0, s));
            // with a communicator the local record goes to its own buffer and the collective writes `tot`: an iteration
            // behind the halt flag then re-reduces the same local records into the same totals (in place it would sum
            // the totals of all ranks again, and a stalled iteration's `tot` is what relocate_empty reads)
            OFC_TRY(launch_reduce_records(sc.partial.as<double>(), nblocks, NV, tot_local, s, halt));
            OFC_TRY(dist_allreduce_f64(tot_local, tot, NV, DIST_SUM, s));
            OFC_TRY(launch_lloyd_update(st, tot, k, d, 0, labelled, it + w == 0, Ng, tol_rel, sc.status_dev + w, s, tiles));
        }
        OFC_HIP(hipStreamSynchronize(s));
        int done = nwin;                      // iterations of this window that really ran
        for (int w = 0; w < nwin; w++) {
            LloydStatus &S = sc.status[w];
            if (!S.valid) { set_error("internal: Lloyd iteration %d did not run", it + w); return OFC_EHIP; }
            if (trace)
                fprintf(stderr, "[ofc lloyd] it %d tiles_mode %d tested %.0f pure %.0f shift %.3e empty %d\n", it + w,
                        S.tiles_mode, S.tiles_tested, S.tiles_pure, S.shift_tot, S.n_empty);
            if (S.tiles_mode != -1) {
                tiles_next = S.tiles_next;
                sc.prune_stats[0] += 1;                                   // sweeps that went tile by tile
                if (S.tiles_mode == LLOYD_TILES_PRUNED) {
                    sc.prune_stats[1] += 1;                               // ... of them pruned
                    sc.prune_stats[2] += S.tiles_tested;                  // tiles tested / skipped by the pruned sweeps
                    sc.prune_stats[3] += S.tiles_pure;
                } else if (S.tiles_mode == LLOYD_TILES_PROBE) {
                    sc.prune_stats[4] += 1;
                }
            }
            if (S.n_empty > 0) {              // the device stalled here; iterations w+1.. of the window were no-ops
                const bool was_labelled = labelled;
                if (!labelled) {   // materialise this iteration's labels (st->centers is still the E-step's input)
                    OFC_TRY(launch_lloyd_assign(X, dtype, N, d, k, st, labels_dev, nullptr, nblocks, 0, 0, s));
                    labelled = true;
                }
                OFC_TRY(relocate_empty(sc, X, dtype, N, d, k, kmax, nblocks, labels_dev, mean_h));
                S.valid = 0;
                OFC_TRY(launch_lloyd_update(st, tot, k, d, 1, was_labelled, it + w == 0, Ng, tol_rel, sc.status_dev + w, s));
                OFC_HIP(hipStreamSynchronize(s));
                done = w + 1;
                if (S.converged) { strict = S.strict != 0; it += w; stop = true; }
                break;
            }
            if (S.converged) { strict = S.strict != 0; it += w; stop = true; break; }
        }
        if (!stop) it += done;
    }
    if (!stop) it = max_iter - 1;        // ran out of iterations: n_iter = max_iter
    // ---- final E-step (only when not strictly converged) and inertia: one sweep ----
    const bool final_tiles = !strict && prune && !labelled && tiles_next == LLOYD_TILES_PRUNED;
    sc.prune_stats[5] = final_tiles ? 1 : 0;
    if (final_tiles)                        // the fit's tile metadata is valid and most tiles pass: those are not read
        OFC_TRY(launch_lloyd_tiles((const float *)X, N, k, st, sc.tile_box.p, sc.tile_sum.p, sc.tile_sq.p, labels_dev,
                                   sc.partial.as<double>(), nblocks, LLOYD_WHAT_FINAL, nullptr, s));
    else if (!strict)
        OFC_TRY(launch_lloyd_assign(X, dtype, N, d, k, st, labels_dev, sc.partial.as<double>(), nblocks, 2, 0, s));
    else
        OFC_TRY(launch_lloyd_inertia(X, dtype, N, d, st, labels_dev, sc.partial.as<double>(), nblocks, s));
    OFC_TRY(launch_reduce_records(sc.partial.as<double>(), nblocks, 1, tot, s));
    OFC_TRY(dist_allreduce_f64(tot, 1, DIST_SUM, s));
    double in = 0, cfin[LLOYD_KMAX * LLOYD_DMAX];
    OFC_HIP(hipMemcpyAsync(&in, tot, sizeof(double), hipMemcpyDeviceToHost, s));
    OFC_HIP(hipMemcpyAsync(cfin, st->centers, sizeof(double) * k * d, hipMemcpyDeviceToHost, s));
    OFC_HIP(hipStreamSynchronize(s));
    for (int j = 0; j < k * d; j++) centers[j] = cfin[j] + mean_h[j % d];
    if (inertia) *inertia = in;
    if (n_iter) *n_iter = it + 1;
    return OFC_OK;
}

static int lloyd_predict_dev(int device, const void *X, int dtype, int64_t N, int d, int k,
                             const double *centers, uint8_t *labels_dev)
{
    OFC_TRY(ensure_device(device));
    if (d > LLOYD_DMAX || k > LLOYD_KMAX || d < 1 || k < 1) {
        set_error("k=%d, d=%d outside the kernels' range (k <= %d, d <= %d)", k, d, LLOYD_KMAX, LLOYD_DMAX);
        return OFC_EUNSUPPORTED;
    }
    DevBuf state;
    OFC_TRY(state.alloc(sizeof(LloydState)));
    LloydState *st = state.as<LloydState>();
    OFC_HIP(hipMemset(st, 0, sizeof(LloydState)));
    OFC_HIP(hipMemcpy(st->centers, centers, sizeof(double) * k * d, hipMemcpyHostToDevice));
    OFC_TRY(launch_lloyd_set_centers(st, k, d, nullptr));
    const int nblocks = lloyd_grid(N);
    OFC_TRY(launch_lloyd_assign(X, dtype, N, d, k, st, labels_dev, nullptr, nblocks, 0, 0, nullptr));
    OFC_HIP(hipStreamSynchronize(nullptr));
    return OFC_OK;
}

static void widen_labels(const uint8_t *src, int64_t N, int32_t *dst)
{
    for (int64_t i = 0; i < N; i++) dst[i] = src[i];
}

}  // namespace ofc

using namespace ofc;

extern "C" {

int ofc_kmeans_fit_dev(int device, const void *X_dev, int dtype, int64_t N, int d, int k, const double *init,
                       int max_iter, double tol_rel, double *centers, uint8_t *labels_dev, double *inertia,
                       int *n_iter)
{
    return lloyd_fit_dev(device, X_dev, dtype, N, d, k, init, max_iter, tol_rel, centers, labels_dev, inertia, n_iter);
}

/* see include/ofc.h */
int ofc_bench_lloyd_sweep(int device, const float *X_dev, int64_t N, int k, const double *centers, const double *mean,
                          int what, int iters, float *ms_per_launch)
{
    OFC_REQUIRE(X_dev && centers && mean && ms_per_launch && iters >= 1 && N >= 64, "bad arguments");
    OFC_REQUIRE(what >= 0 && what <= 5, "what = %d outside 0..5", what);
    if (!lloyd_tiles_supported(OFC_F32, 2, k)) { set_error("k=%d outside 1..8", k); return OFC_EUNSUPPORTED; }
    OFC_TRY(ensure_device(device));
    LloydScratch &sc = scratch_for(device);
    std::lock_guard<std::mutex> lock(sc.mu);
    OFC_TRY(sc.init());
    hipStream_t s = sc.stream;
    const int nblocks = lloyd_grid(N);
    const size_t need = (size_t)(N >> 6) * 16;
    if (sc.tile_box.bytes < need) {
        OFC_TRY(sc.tile_box.alloc(need));
        OFC_TRY(sc.tile_sum.alloc(need));
        OFC_TRY(sc.tile_sq.alloc(need / 2));
    }
    if (sc.labels.bytes < (size_t)N) OFC_TRY(sc.labels.alloc((size_t)N));
    LloydState *st = sc.state.as<LloydState>();
    OFC_HIP(hipMemsetAsync(st, 0, sizeof(LloydState), s));
    double c0[LLOYD_KMAX * LLOYD_DMAX];
    for (int j = 0; j < k * 2; j++) c0[j] = centers[j] - mean[j % 2];
    OFC_HIP(hipMemcpyAsync(st->mean, mean, sizeof(double) * 2, hipMemcpyHostToDevice, s));
    OFC_HIP(hipMemcpyAsync(st->centers, c0, sizeof(double) * k * 2, hipMemcpyHostToDevice, s));
    OFC_TRY(launch_lloyd_set_centers(st, k, 2, s, LLOYD_PRUNE_ALWAYS));
    double *partial = sc.partial.as<double>();
    const float *X = X_dev;
    auto launch = [&]() -> int {
        switch (what) {
        case 0: return launch_lloyd_assign(X, OFC_F32, N, 2, k, st, sc.labels.as<uint8_t>(), partial, nblocks, 3, 0, s);
        case 1: case 5: return launch_lloyd_tiles(X, N, k, st, sc.tile_box.p, sc.tile_sum.p, sc.tile_sq.p, nullptr, partial, nblocks, LLOYD_WHAT_SWEEP, nullptr, s);
        case 2: return launch_lloyd_tiles(X, N, k, st, sc.tile_box.p, sc.tile_sum.p, sc.tile_sq.p, nullptr, partial, nblocks, LLOYD_WHAT_META, nullptr, s);
        case 3: return launch_lloyd_assign(X, OFC_F32, N, 2, k, st, sc.labels.as<uint8_t>(), partial, nblocks, 2, 0, s);
        default: return launch_lloyd_tiles(X, N, k, st, sc.tile_box.p, sc.tile_sum.p, sc.tile_sq.p, sc.labels.as<uint8_t>(), partial, nblocks, LLOYD_WHAT_FINAL, nullptr, s);
        }
    };
    if (what == 1 || what == 4 || what == 5) {      // the pruned sweeps need the tile metadata and the mode flag
        OFC_TRY(launch_lloyd_tiles(X, N, k, st, sc.tile_box.p, sc.tile_sum.p, sc.tile_sq.p, nullptr, partial, nblocks, LLOYD_WHAT_META, nullptr, s));
        const int mode = what == 5 ? LLOYD_TILES_FULL : LLOYD_TILES_PRUNED;
        OFC_HIP(hipMemcpyAsync(&st->prune_mode, &mode, sizeof(int), hipMemcpyHostToDevice, s));
    }
    hipEvent_t e0, e1;
    OFC_HIP(hipEventCreate(&e0));
    OFC_HIP(hipEventCreate(&e1));
    for (int w = 0; w < 2; w++) OFC_TRY(launch());
    OFC_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < iters; i++) OFC_TRY(launch());
    OFC_HIP(hipEventRecord(e1, s));
    OFC_HIP(hipEventSynchronize(e1));
    float ms = 0;
    OFC_HIP(hipEventElapsedTime(&ms, e0, e1));
    *ms_per_launch = ms / iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return OFC_OK;
}

/* see include/ofc.h */
int ofc_lloyd_prune_stats(int device, double *out6)
{
    OFC_REQUIRE(out6, "null pointer");
    OFC_TRY(ensure_device(device));
    LloydScratch &sc = scratch_for(device);
    std::lock_guard<std::mutex> lock(sc.mu);
    for (int i = 0; i < 6; i++) out6[i] = sc.prune_stats[i];
    return OFC_OK;
}

int ofc_kmeans_fit_dev_stats(int device, const void *X_dev, int dtype, int64_t N, int d, int k, const double *init,
                             int max_iter, double tol_rel, const double *colsum, double *centers, uint8_t *labels_dev,
                             double *inertia, int *n_iter)
{
    return lloyd_fit_dev(device, X_dev, dtype, N, d, k, init, max_iter, tol_rel, centers, labels_dev, inertia, n_iter, colsum);
}

// ---- host-driven building blocks ----
namespace {
struct StepCtx {
    DevBuf state, partial, tot, excl, far;
    int nblocks = 1;
    int init(int64_t N, int nv)
    {
        nblocks = lloyd_grid(N);
        OFC_TRY(state.alloc(sizeof(LloydState)));
        OFC_TRY(partial.alloc(sizeof(double) * (size_t)nblocks * std::max(nv, 2)));
        OFC_TRY(tot.alloc(sizeof(double) * (nv + 8)));
        OFC_HIP(hipMemset(state.p, 0, sizeof(LloydState)));
        return OFC_OK;
    }
    int set(const double *mean, const double *centers_c, int k, int d)
    {
        LloydState *st = state.as<LloydState>();
        if (mean) OFC_HIP(hipMemcpy(st->mean, mean, sizeof(double) * d, hipMemcpyHostToDevice));
        if (centers_c) {
            OFC_HIP(hipMemcpy(st->centers, centers_c, sizeof(double) * k * d, hipMemcpyHostToDevice));
            OFC_TRY(launch_lloyd_set_centers(st, k, d, nullptr));
        }
        return OFC_OK;
    }
};
int check_kd(int k, int d)
{
    if (d < 1 || k < 1 || d > LLOYD_DMAX || k > LLOYD_KMAX) {
        set_error("k=%d, d=%d outside the kernels' range (k <= %d, d <= %d)", k, d, LLOYD_KMAX, LLOYD_DMAX);
        return OFC_EUNSUPPORTED;
    }
    return OFC_OK;
}
}  // namespace

int ofc_lloyd_colstats_dev(int device, const void *X_dev, int dtype, int64_t N, int d, const double *mean, int pass,
                           double *out)
{
    OFC_REQUIRE(X_dev && out && N >= 0 && (pass == 0 || mean), "bad arguments");
    OFC_TRY(check_kd(1, d));
    OFC_TRY(ensure_device(device));
    StepCtx c;
    OFC_TRY(c.init(N, d));
    OFC_TRY(c.set(pass ? mean : nullptr, nullptr, 1, d));
    LloydState *st = c.state.as<LloydState>();
    OFC_TRY(launch_lloyd_colstats(X_dev, dtype, N, d, st->mean, pass, c.partial.as<double>(), c.nblocks, nullptr));
    OFC_TRY(launch_reduce_records(c.partial.as<double>(), c.nblocks, d, c.tot.as<double>(), nullptr));
    OFC_HIP(hipMemcpy(out, c.tot.p, sizeof(double) * d, hipMemcpyDeviceToHost));
    return OFC_OK;
}

int ofc_lloyd_step_dev(int device, const void *X_dev, int dtype, int64_t N, int d, int k, const double *mean,
                       const double *centers_c, uint8_t *labels_dev, int accumulate, double *record)
{
    OFC_REQUIRE(X_dev && mean && centers_c && labels_dev && (record || !accumulate) && N >= 0, "bad arguments");
    OFC_TRY(check_kd(k, d));
    OFC_TRY(ensure_device(device));
    const int kmax = lloyd_kmax(k), NV = lloyd_record_len(kmax, d);
    StepCtx c;
    OFC_TRY(c.init(N, NV));
    OFC_TRY(c.set(mean, centers_c, k, d));
    LloydState *st = c.state.as<LloydState>();
    OFC_TRY(launch_lloyd_assign(X_dev, dtype, N, d, k, st, labels_dev, c.partial.as<double>(), c.nblocks, accumulate ? 1 : 0, 0, nullptr));
    if (accumulate) {
        OFC_TRY(launch_reduce_records(c.partial.as<double>(), c.nblocks, NV, c.tot.as<double>(), nullptr));
        std::vector<double> t(NV);
        OFC_HIP(hipMemcpy(t.data(), c.tot.p, sizeof(double) * NV, hipMemcpyDeviceToHost));
        for (int j = 0; j < k; j++) {
            for (int f = 0; f < d; f++) record[j * d + f] = t[j * d + f];
            record[k * d + j] = t[kmax * d + j];
        }
        record[k * d + k] = t[kmax * d + kmax];
    } else {
        OFC_HIP(hipStreamSynchronize(nullptr));
    }
    return OFC_OK;
}

int ofc_lloyd_inertia_dev(int device, const void *X_dev, int dtype, int64_t N, int d, int k, const double *mean,
                          const double *centers_c, const uint8_t *labels_dev, double *inertia)
{
    OFC_REQUIRE(X_dev && mean && centers_c && labels_dev && inertia && N >= 0, "bad arguments");
    OFC_TRY(check_kd(k, d));
    OFC_TRY(ensure_device(device));
    StepCtx c;
    OFC_TRY(c.init(N, 2));
    OFC_TRY(c.set(mean, centers_c, k, d));
    OFC_TRY(launch_lloyd_inertia(X_dev, dtype, N, d, c.state.as<LloydState>(), labels_dev, c.partial.as<double>(), c.nblocks, nullptr));
    OFC_TRY(launch_reduce_records(c.partial.as<double>(), c.nblocks, 1, c.tot.as<double>(), nullptr));
    OFC_HIP(hipMemcpy(inertia, c.tot.p, sizeof(double), hipMemcpyDeviceToHost));
    return OFC_OK;
}

int ofc_lloyd_farthest_dev(int device, const void *X_dev, int dtype, int64_t N, int d, int k, const double *mean,
                           const double *centers_c, const uint8_t *labels_dev, const int64_t *excl, int n_excl,
                           double *dist2, int64_t *index, double *x_c, int *label)
{
    OFC_REQUIRE(X_dev && mean && centers_c && labels_dev && dist2 && index && x_c && label && N >= 0, "bad arguments");
    OFC_REQUIRE(n_excl >= 0 && n_excl <= LLOYD_KMAX && (n_excl == 0 || excl), "bad exclusion list");
    OFC_TRY(check_kd(k, d));
    OFC_TRY(ensure_device(device));
    StepCtx c;
    OFC_TRY(c.init(N, 2));
    OFC_TRY(c.set(mean, centers_c, k, d));
    OFC_TRY(c.excl.alloc(sizeof(int64_t) * LLOYD_KMAX));
    if (n_excl) OFC_HIP(hipMemcpy(c.excl.p, excl, sizeof(int64_t) * n_excl, hipMemcpyHostToDevice));
    LloydState *st = c.state.as<LloydState>();
    OFC_TRY(launch_lloyd_farthest(X_dev, dtype, N, d, st, st->centers, labels_dev, c.excl.as<int64_t>(), n_excl,
                                  c.partial.as<double>(), c.nblocks, nullptr));
    std::vector<double> blk(2 * c.nblocks);
    OFC_HIP(hipMemcpy(blk.data(), c.partial.p, sizeof(double) * 2 * c.nblocks, hipMemcpyDeviceToHost));
    double best = -1;
    int64_t bi = -1;
    for (int b = 0; b < c.nblocks; b++) {
        const double v = blk[2 * b];
        const int64_t i = (int64_t)blk[2 * b + 1];
        if (i < 0) continue;
        if (v > best || (v == best && i < bi)) { best = v; bi = i; }
    }
    *dist2 = bi >= 0 ? best : -1;
    *index = bi;
    *label = -1;
    if (bi >= 0) {
        unsigned char raw[LLOYD_DMAX * 8];
        uint8_t lab;
        OFC_HIP(hipMemcpy(raw, (const char *)X_dev + (size_t)bi * d * dtype_size(dtype), d * dtype_size(dtype), hipMemcpyDeviceToHost));
        OFC_HIP(hipMemcpy(&lab, labels_dev + bi, 1, hipMemcpyDeviceToHost));
        for (int f = 0; f < d; f++) {
            double v = dtype == OFC_U8 ? (double)raw[f] : dtype == OFC_F32 ? (double)((float *)raw)[f] : ((double *)raw)[f];
            x_c[f] = v - mean[f];
        }
        *label = lab;
    }
    return OFC_OK;
}

int ofc_kmeans_fit(int device, const void *X, int dtype, int64_t N, int d, int k, const double *init,
                   int max_iter, double tol_rel, double *centers, int32_t *labels, double *inertia, int *n_iter)
{
    OFC_REQUIRE(X && init && centers, "null pointer");
    OFC_REQUIRE(dtype >= OFC_U8 && dtype <= OFC_F64 && d >= 1 && N >= 0, "bad arguments");
    OFC_REQUIRE(N >= k, "n_samples=%lld should be >= n_clusters=%d.", (long long)N, k);
    OFC_TRY(ensure_device(device));
    DevBuf dX, dL;
    const size_t bytes = (size_t)N * d * dtype_size(dtype);
    OFC_TRY(dX.alloc(std::max<size_t>(bytes, 16)));
    OFC_TRY(dL.alloc((size_t)std::max<int64_t>(N, 1)));
    OFC_HIP(hipMemcpy(dX.p, X, bytes, hipMemcpyHostToDevice));
    OFC_TRY(lloyd_fit_dev(device, dX.p, dtype, N, d, k, init, max_iter, tol_rel, centers, dL.as<uint8_t>(), inertia, n_iter));
    if (labels) {
        std::vector<uint8_t> l8((size_t)N);
        OFC_HIP(hipMemcpy(l8.data(), dL.p, (size_t)N, hipMemcpyDeviceToHost));
        widen_labels(l8.data(), N, labels);
    }
    return OFC_OK;
}

int ofc_kmeans_predict(int device, const void *X, int dtype, int64_t N, int d, int k, const double *centers,
                       int32_t *labels)
{
    OFC_REQUIRE(X && centers && labels, "null pointer");
    OFC_REQUIRE(dtype >= OFC_U8 && dtype <= OFC_F64 && d >= 1 && N >= 0 && k >= 1, "bad arguments");
    OFC_TRY(ensure_device(device));
    DevBuf dX, dL;
    const size_t bytes = (size_t)N * d * dtype_size(dtype);
    OFC_TRY(dX.alloc(std::max<size_t>(bytes, 16)));
    OFC_TRY(dL.alloc((size_t)std::max<int64_t>(N, 1)));
    OFC_HIP(hipMemcpy(dX.p, X, bytes, hipMemcpyHostToDevice));
    OFC_TRY(lloyd_predict_dev(device, dX.p, dtype, N, d, k, centers, dL.as<uint8_t>()));
    std::vector<uint8_t> l8((size_t)N);
    OFC_HIP(hipMemcpy(l8.data(), dL.p, (size_t)N, hipMemcpyDeviceToHost));
    widen_labels(l8.data(), N, labels);
    return OFC_OK;
}

/* see include/ofc.h */
int ofc_kpp_candidates(int device, const void *X, int dtype, int64_t N, int d, const double *mean,
                       const int64_t *cand, int n_cand, const double *closest, double *out_min, double *pots)
{
    OFC_REQUIRE(X && mean && cand && out_min && pots && N >= 1, "bad arguments");
    OFC_REQUIRE(dtype >= OFC_U8 && dtype <= OFC_F64, "bad dtype %d", dtype);
    OFC_REQUIRE(n_cand >= 1 && n_cand <= 8, "n_cand %d outside 1..8", n_cand);
    OFC_TRY(check_kd(1, d));
    for (int c = 0; c < n_cand; c++) OFC_REQUIRE(cand[c] >= 0 && cand[c] < N, "candidate index out of range");
    OFC_TRY(ensure_device(device));
    const size_t es = dtype_size(dtype);
    std::vector<double> cc((size_t)n_cand * d);
    for (int c = 0; c < n_cand; c++)
        for (int f = 0; f < d; f++) {
            const char *p = (const char *)X + ((size_t)cand[c] * d + f) * es;
            const double v = dtype == OFC_U8 ? (double)*(const uint8_t *)p
                           : dtype == OFC_F32 ? (double)*(const float *)p : *(const double *)p;
            cc[(size_t)c * d + f] = v - mean[f];
        }
    const int nblocks = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv64(N, 256), 1024));
    DevBuf Xd, cl, out, partial, tot;
    OFC_TRY(Xd.alloc((size_t)N * d * es));
    OFC_TRY(out.alloc(sizeof(double) * (size_t)N * n_cand));
    OFC_TRY(partial.alloc(sizeof(double) * 8 * nblocks));
    OFC_TRY(tot.alloc(sizeof(double) * 8));
    OFC_HIP(hipMemcpy(Xd.p, X, (size_t)N * d * es, hipMemcpyHostToDevice));
    if (closest) {
        OFC_TRY(cl.alloc(sizeof(double) * N));
        OFC_HIP(hipMemcpy(cl.p, closest, sizeof(double) * N, hipMemcpyHostToDevice));
    }
    OFC_TRY(launch_kpp_candidates(Xd.p, dtype, N, d, mean, cc.data(), n_cand, closest ? cl.as<double>() : nullptr,
                                  out.as<double>(), partial.as<double>(), nblocks, nullptr));
    OFC_TRY(launch_reduce_records(partial.as<double>(), nblocks, 8, tot.as<double>(), nullptr));
    double t[8];
    OFC_HIP(hipMemcpy(t, tot.p, sizeof(t), hipMemcpyDeviceToHost));
    for (int c = 0; c < n_cand; c++) pots[c] = t[c];
    OFC_HIP(hipMemcpy(out_min, out.p, sizeof(double) * (size_t)N * n_cand, hipMemcpyDeviceToHost));
    return OFC_OK;
}

}  // extern "C"
