// lloyd_common.h -- device-side state and launcher prototypes of the Lloyd kernels.
#pragma once
#include "ofc_common.h"

namespace ofc {

constexpr int LLOYD_KMAX = 16;   // streaming / batched kernels keep per-lane partials in registers
constexpr int LLOYD_DMAX = 4;

// lives in device memory; centres are stored CENTRED (x - column mean), as sklearn iterates
struct LloydState {
    double mean[LLOYD_DMAX];
    double centers[LLOYD_KMAX * LLOYD_DMAX];      // current (old) centres, row-major k x d
    double centers_new[LLOYD_KMAX * LLOYD_DMAX];
    double cn[LLOYD_KMAX];                        // |c_j|^2 (FMA chain)
    double tol;                                   // sklearn's tol (mean column variance * tol_rel), set by iteration 0
    int halt;                                     // set by k_lloyd_update on convergence / empty cluster: the iterations
    int pad;                                      // that were enqueued speculatively behind it become no-ops
    // tile pruning (lloyd_tiles.hip): decided on the device, iteration by iteration, from the all-reduced tile counts
    int prune_policy;                             // LLOYD_PRUNE_*: set by the host before iteration 0
    int prune_mode;                               // how the NEXT k_lloyd_tiles sweep runs: LLOYD_TILES_*
    int prune_cooldown, prune_backoff;            // full sweeps left before the next probe / its current spacing
};

// how a k_lloyd_tiles sweep treats the 64-sample tiles
enum { LLOYD_TILES_FULL = 0,      // every tile sample by sample (no metadata read)
       LLOYD_TILES_PRUNED = 1,    // box test per tile: cached sums for tiles wholly inside one Voronoi cell, the rest by sample
       LLOYD_TILES_PROBE = 2 };   // box test only counted, every tile still by sample (decides whether to switch to PRUNED)
enum { LLOYD_PRUNE_OFF = 0, LLOYD_PRUNE_AUTO = 1, LLOYD_PRUNE_ON = 2, LLOYD_PRUNE_ALWAYS = 3 };

// written by k_lloyd_update into pinned host memory once per iteration (one slot per iteration of a window)
struct LloydStatus {
    double n_changed;
    double shift_tot;
    int n_empty;
    int valid;        // the iteration ran (0: it was a no-op behind a halt)
    int converged;    // strict or tol stop reached in this iteration
    int strict;       // ... by labels == labels_old
    double tol;
    double counts[LLOYD_KMAX];
    double sqsum[LLOYD_DMAX];     // sum (x-mean)^2 per column (first iteration only)
    double tiles_tested, tiles_pure;   // k_lloyd_tiles: tiles box-tested (or checked for uniform labels while the metadata
    int tiles_mode;                    // is built) / found inside one cell; mode the sweep ran in (-1: not a tile sweep)
    int tiles_next;                    // mode chosen for the next tile sweep (LLOYD_TILES_*)
};

int lloyd_kmax(int k);
int launch_lloyd_colstats(const void *X, int dtype, int64_t N, int d, const double *mean, int pass,
                          double *partial, int nblocks, hipStream_t s);
// halt != nullptr: skip when *halt != 0 (device flag)
int launch_reduce_records(const double *partial, int nblocks, int nv, double *out, hipStream_t s,
                          const int *halt = nullptr);
// mode 0: labels only; 1: labels + M-step record (+ column sums of (x-mean)^2 when first != 0); 2: labels + inertia;
// 3: M-step record only (labels untouched, n_changed = 0)
// record = [kmax*d sums][kmax counts][n_changed][LLOYD_DMAX squared sums][tiles tested][tiles pure]
int launch_lloyd_assign(const void *X, int dtype, int64_t N, int d, int k, const LloydState *st,
                        uint8_t *labels, double *partial, int nblocks, int mode, int first, hipStream_t s);
constexpr int LLOYD_REC_EXTRA = 1 + LLOYD_DMAX + 2;     // slots behind the sums and counts
constexpr int LLOYD_NVMAX = LLOYD_KMAX * LLOYD_DMAX + LLOYD_KMAX + LLOYD_REC_EXTRA;
inline int lloyd_record_len(int kmax, int d) { return kmax * d + kmax + LLOYD_REC_EXTRA; }
// (u,v) stream (f32, d = 2, k <= 8) in 64-sample tiles, see lloyd_tiles.hip.
//   LLOYD_WHAT_PROBE  before anything else: box test of a 1/64 sample of the tiles against the initial centres; sets
//                     st->prune_mode, and st->prune_policy = OFF on the device when the field is too incoherent to pay
//   LLOYD_WHAT_META   before iteration 0: box[N/64] (lo_u, lo_v, hi_u, hi_v), tsum[N/64] (sum of the centred samples),
//                     tsq[N/64] (scatter about the tile's own mean) written; partial[block][2] = column sums of (x-mean)^2
//                     (returns at once, records zeroed, when the probe switched the policy off)
//   LLOYD_WHAT_SWEEP  one label-less iteration in st->prune_mode; meta != nullptr: iteration 0 (meta = the reduced META
//                     record, device pointer: its sums of squares ride along).  Record as launch_lloyd_assign's mode 3
//   LLOYD_WHAT_FINAL  the final E-step (labels written, partial[block] = inertia share)
enum { LLOYD_WHAT_SWEEP = 0, LLOYD_WHAT_META = 1, LLOYD_WHAT_FINAL = 2, LLOYD_WHAT_PROBE = 3 };
bool lloyd_tiles_supported(int dtype, int d, int k);
int launch_lloyd_tiles(const float *X, int64_t N, int k, const LloydState *st, void *box, void *tsum, void *tsq,
                       uint8_t *labels, double *partial, int nblocks, int what, const double *meta, hipStream_t s);
// tiles: 0 = the sweep was not a k_lloyd_tiles one, otherwise it ran in st->prune_mode (1: iteration 0)
int launch_lloyd_update(LloydState *st, const double *tot, int k, int d, int after_reloc, int labelled, int first,
                        double n_total, double tol_rel, LloydStatus *status, hipStream_t s, int tiles = 0);
constexpr int LLOYD_WINDOW = 16;    // most iterations enqueued per host synchronisation (windows grow 4, 8, 16, 16, ...)
int launch_lloyd_set_centers(LloydState *st, int k, int d, hipStream_t s, int prune_policy = 0);
int launch_lloyd_inertia(const void *X, int dtype, int64_t N, int d, const LloydState *st,
                         const uint8_t *labels, double *partial, int nblocks, hipStream_t s);
int launch_lloyd_farthest(const void *X, int dtype, int64_t N, int d, const LloydState *st,
                          const double *c_old, const uint8_t *labels, const int64_t *excl, int n_excl,
                          double *out, int nblocks, hipStream_t s);

// k-means++ seeding step; partial: [nblocks][8] potentials per candidate
int launch_kpp_candidates(const void *X, int dtype, int64_t N, int d, const double *mean, const double *cand_centred,
                          int n_cand, const double *closest, double *out, double *partial, int nblocks, hipStream_t s);

// distributed plumbing (dist.cpp): no-ops when no communicator is set up
bool dist_active();
int dist_rank();
int dist_world();
enum { DIST_SUM = 0, DIST_MAX = 1, DIST_MIN = 2, DIST_BCAST = 3 /* a sum to which exactly one rank contributes */ };
int dist_allreduce_f64(double *buf_dev, int count, int op, hipStream_t s);                      // in place
int dist_allreduce_f64(const double *send_dev, double *recv_dev, int count, int op, hipStream_t s);
bool dist_has_comm();
int launch_loopback_reduce(const double *send, double *recv, int count, int world, int sum, hipStream_t s);

}  // namespace ofc
