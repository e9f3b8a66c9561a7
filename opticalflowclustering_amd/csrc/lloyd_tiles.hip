// lloyd_tiles.hip -- Lloyd sweeps over a resident (u,v) field (f32, d = 2, k <= 8) that do not re-read what cannot
// change its cluster (round 3).  Semantics stay sklearn's Lloyd (_kmeans.py:624-752, _k_means_lloyd.pyx:23-218) as
// lloyd_api.cpp restates it; this file only changes HOW the label-less M-step record of an iteration is formed.
//
// The samples are cut into tiles of 64 consecutive ones (512 B; a row segment of the flow field, whose vectors are
// close to each other inside a motion population).  Before iteration 0 one streaming pass (k_tile_meta) leaves per tile
//     box[t]  = (min u, min v, max u, max v)      f32, 16 B
//     tsum[t] = sum over the tile of (x - mean)   f64 x 2, 16 B      (the mean is fixed for the whole fit)
//     tsq[t]  = sum |x - mean(tile)|^2            f64, 8 B           (for the final E-step's inertia)
// and the column sums of (x - mean)^2 that sklearn's tol needs (_tolerance, _kmeans.py:279-287).
// From then on a tile is tested before it is read: D_j(x) - D_j'(x) = (|c_j|^2 - |c_j'|^2) + 2 x.(c_j' - c_j) is linear
// in x, so its maximum over the box sits at a corner; if it is negative (by a margin far above the rounding of the
// sample-by-sample arithmetic) for every j' != j, every sample of the tile has label j whatever its position in the
// box, and the tile contributes tsum[t] and 64 to cluster j without being read.  No drift tracking, no runner-up
// distances (round 2's Hamerly attempt lost on exactly those).  Tiles that fail the test are walked sample by sample
// by the same wave, 16 lanes per tile, four points per lane -- the arithmetic of k_lloyd_assign's mode 3.
// The final E-step uses the same test: a tile inside one cell of the FINAL centres gets its 64 label bytes without being
// read (the test proves that every sample's argmin is that cell), and its inertia share is tsq[t] + 64 |mean(tile) - c_j|^2.
// Whether any of this is used is decided on the device: k_tile_probe / k_tile_decide test a 1/64 sample of the tiles against
// the initial centres first (an incoherent field skips the metadata pass and sweeps in full), and k_lloyd_update
// (lloyd_kernels.hip) switches between pruned, full and counting-only sweeps from every iteration's all-reduced tile counts.
//
// Deterministic: a wave owns a fixed set of tile groups, every lane adds into its private LDS column in a fixed order,
// columns and work-groups are folded in a fixed order (as in lloyd_kernels.hip).  The sums differ from the unpruned
// sweep's in the last bits only (another summation order): centres agree to ~1e-15, labels and n_iter are the same
// (tests/test_gpu_lloyd_tiles.py, bench.py's config.pruned_fit_check).
#include "lloyd_common.h"

#include <algorithm>

namespace ofc {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

// value of lane (l + N) % 16 of the same 16-lane row (DPP row_ror: no LDS, no address register)
template <int N> __device__ __forceinline__ int row_ror(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, 0x120 + N, 0xf, 0xf, false);
}
template <int N> __device__ __forceinline__ float row_ror(float v) { return __int_as_float(row_ror<N>(__float_as_int(v))); }
template <int N> __device__ __forceinline__ double row_ror(double v)
{
    return __hiloint2double(row_ror<N>(__double2hiint(v)), row_ror<N>(__double2loint(v)));
}
// all-reduce over the 16 lanes of a row; every lane ends with the bit-identical result (each step combines a pair
// symmetrically)
__device__ __forceinline__ float row_min(float v)
{
    v = fminf(v, row_ror<8>(v)); v = fminf(v, row_ror<4>(v)); v = fminf(v, row_ror<2>(v)); return fminf(v, row_ror<1>(v));
}
__device__ __forceinline__ float row_max(float v)
{
    v = fmaxf(v, row_ror<8>(v)); v = fmaxf(v, row_ror<4>(v)); v = fmaxf(v, row_ror<2>(v)); return fmaxf(v, row_ror<1>(v));
}
__device__ __forceinline__ double row_sum(double v)
{
    v += row_ror<8>(v); v += row_ror<4>(v); v += row_ror<2>(v); return v + row_ror<1>(v);
}

// centres (centred, as sklearn iterates) in registers for the unrolled distance loop, and in LDS for a lane-varying index
// (a select chain over the register copies is turned into a dynamically indexed private array, i.e. scratch memory)
template <int KMAX> struct TileCtx {
    double c[KMAX * 2], cn[KMAX], m[2];
    const double (*s_cen)[4];                    // LDS: (c_x, c_y, |c|^2, -) per cluster

    __device__ __forceinline__ void load(const LloydState *st, double (*lds)[4])
    {
#pragma unroll
        for (int j = 0; j < KMAX; j++) {
            cn[j] = st->cn[j];
            c[2 * j] = st->centers[2 * j];
            c[2 * j + 1] = st->centers[2 * j + 1];
        }
        m[0] = st->mean[0];
        m[1] = st->mean[1];
        if (threadIdx.x < KMAX) {
            lds[threadIdx.x][0] = st->centers[threadIdx.x * 2];
            lds[threadIdx.x][1] = st->centers[threadIdx.x * 2 + 1];
            lds[threadIdx.x][2] = st->cn[threadIdx.x];
        }
        s_cen = lds;
        __syncthreads();
    }
    // first strict minimum of D_j = |c_j|^2 - 2 x.c_j (the arithmetic of lloyd_kernels.hip's assign_point for d = 2)
    __device__ __forceinline__ int label_of(double x0, double x1) const
    {
        double best = 0;
        int label = 0;
#pragma unroll
        for (int j = 0; j < KMAX; j++) {
            const double dj = cn[j] - 2.0 * fma(x1, c[2 * j + 1], x0 * c[2 * j]);
            if (j == 0 || dj < best) { best = dj; label = j; }
        }
        return label;
    }
    __device__ __forceinline__ void centre_of(int j, double &cjx, double &cjy, double &cnj) const
    {
        cjx = s_cen[j][0]; cjy = s_cen[j][1]; cnj = s_cen[j][2];
    }
    // is the (centred) box inside one Voronoi cell?  -> its label, or -1
    __device__ __forceinline__ int box_label(double lox, double loy, double hix, double hiy) const
    {
        const int j = label_of(lox, loy);       // the candidate; the test below proves or rejects it for the whole box
        double cjx, cjy, cnj;
        centre_of(j, cjx, cjy, cnj);
        const double ax = fmax(fabs(lox), fabs(hix)), ay = fmax(fabs(loy), fabs(hiy));
        bool pure = true;
#pragma unroll
        for (int q = 0; q < KMAX; q++) {
            const double gx = c[2 * q] - cjx, gy = c[2 * q + 1] - cjy;
            // max over the box of D_j - D_q
            const double wmax = (cnj - cn[q]) + 2.0 * (fmax(lox * gx, hix * gx) + fmax(loy * gy, hiy * gy));
            // every term that enters a sample's D_j, D_q, times 1e-12: >> their f64 rounding (~1e-15 of the same)
            const double mag = (fabs(cnj) + fabs(cn[q])) +
                               4.0 * (ax * (fabs(c[2 * q]) + fabs(cjx)) + ay * (fabs(c[2 * q + 1]) + fabs(cjy)));
            pure = pure && (q == j || wmax < -1e-12 * mag);
        }
        return pure ? j : -1;
    }
    // sklearn's per-sample squared distance for d = 2 (_euclidean_dense_dense, as lloyd_kernels.hip's sq_euclid_grouped)
    __device__ __forceinline__ double sq_dist(double x0, double x1, int l) const
    {
#pragma clang fp contract(off)
        double cjx, cjy, cnj;
        centre_of(l, cjx, cjy, cnj);
        double r = 0;
        r += (x0 - cjx) * (x0 - cjx);
        r += (x1 - cjy) * (x1 - cjy);
        return r;
    }
};

// ------------------------------------------------------------------------------------------------
// Is the field coherent enough for the tile test to pay?  k_tile_probe forms the boxes of a 1/64 sample of the tiles and
// tests them against the INITIAL centres (reads 1.6 % of the samples); k_tile_decide turns the policy off on the device
// when fewer than 45 % pass: k_tile_meta then returns at once, every sweep runs full, and an incoherent field (white
// noise over the populations) pays ~2 % of one sweep for having been asked instead of a whole metadata pass.
// 16 lanes <-> one tile, four samples per lane, 4 tiles per wave step (as k_tile_meta).
// ------------------------------------------------------------------------------------------------
template <int KMAX>
__global__ __launch_bounds__(256) void k_tile_probe(const float *__restrict__ X, int64_t N,
                                                    const LloydState *__restrict__ st, double *__restrict__ partial)
{
    __shared__ double s_cen[KMAX][4];
    __shared__ unsigned s_red[4][2];
    TileCtx<KMAX> cx;
    cx.load(st, s_cen);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, row = lane >> 4, r16 = lane & 15;
    const int64_t NT = N >> 6, NS = (NT + 3) >> 2;
    const int64_t nw = (int64_t)gridDim.x * 4, w = (int64_t)blockIdx.x * 4 + wave;
    const int64_t stride = NS > 64 * nw ? 64 : (NS > nw ? NS / nw : 1);      // every 64th step of a large field
    unsigned n_tested = 0, n_pure = 0;
    for (int64_t s = w * stride; s < NS; s += nw * stride) {
        const int64_t t = s * 4 + row;
        if (t < NT) {                               // row-uniform
            const v4f *p = reinterpret_cast<const v4f *>(X + t * 128 + r16 * 8);
            const v4f a = p[0], b = p[1];
            const float lu = row_min(fminf(fminf(a.x, a.z), fminf(b.x, b.z)));
            const float lv = row_min(fminf(fminf(a.y, a.w), fminf(b.y, b.w)));
            const float hu = row_max(fmaxf(fmaxf(a.x, a.z), fmaxf(b.x, b.z)));
            const float hv = row_max(fmaxf(fmaxf(a.y, a.w), fmaxf(b.y, b.w)));
            const int tl = cx.box_label((double)lu - cx.m[0], (double)lv - cx.m[1], (double)hu - cx.m[0], (double)hv - cx.m[1]);
            if (r16 == 0) {
                n_tested += 1u;
                n_pure += (tl >= 0);
            }
        }
    }
    for (int off = 32; off >= 1; off >>= 1) {
        n_tested += __shfl_down(n_tested, off, 64);
        n_pure += __shfl_down(n_pure, off, 64);
    }
    if (lane == 0) { s_red[wave][0] = n_tested; s_red[wave][1] = n_pure; }
    __syncthreads();
    if (tid < 2) partial[(size_t)blockIdx.x * 2 + tid] = (double)(s_red[0][tid] + s_red[1][tid] + s_red[2][tid] + s_red[3][tid]);
}

// one work-group: adds the probe's records and sets how the fit sweeps (st->prune_policy, st->prune_mode)
__global__ void k_tile_decide(LloydState *st, const double *__restrict__ partial, int nblocks)
{
    __shared__ double s[2][64];
    double a = 0, b = 0;
    for (int i = threadIdx.x; i < nblocks; i += 64) { a += partial[2 * i]; b += partial[2 * i + 1]; }
    s[0][threadIdx.x] = a;
    s[1][threadIdx.x] = b;
    __syncthreads();
    if (threadIdx.x != 0) return;
    double tested = 0, pure = 0;
    for (int i = 0; i < 64; i++) { tested += s[0][i]; pure += s[1][i]; }
    if (st->prune_policy != LLOYD_PRUNE_ALWAYS && !(pure >= 0.45 * tested && tested > 0)) st->prune_policy = LLOYD_PRUNE_OFF;
    st->prune_mode = st->prune_policy == LLOYD_PRUNE_OFF ? LLOYD_TILES_FULL : LLOYD_TILES_PRUNED;
}

// ------------------------------------------------------------------------------------------------
// one streaming pass over the samples before iteration 0: tile metadata and the column sums of (x - mean)^2.
// 16 lanes <-> one tile, four samples per lane; a wave walks 4 tiles per step, the next step's samples requested first.
// partial[block] = [sum (u-mean_u)^2, sum (v-mean_v)^2]; nothing is done when k_tile_decide switched the policy off.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tile_meta(const float *__restrict__ X, int64_t N,
                                                   const LloydState *__restrict__ st, v4f *__restrict__ box,
                                                   v2d *__restrict__ tsum, double *__restrict__ tsq,
                                                   double *__restrict__ partial)
{
    __shared__ double s_red[4][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, row = lane >> 4, r16 = lane & 15;
    if (st->prune_policy == LLOYD_PRUNE_OFF) {      // uniform
        if (tid < 2) partial[(size_t)blockIdx.x * 2 + tid] = 0.0;
        return;
    }
    const double m0 = st->mean[0], m1 = st->mean[1];
    const int64_t NT = N >> 6;
    const int64_t NS = (NT + 3) >> 2;               // steps of 4 tiles
    const int64_t s0 = (int64_t)blockIdx.x * 4 + wave, ss = (int64_t)gridDim.x * 4;
    double sq0 = 0, sq1 = 0;
    auto fetch = [&](int64_t s, v4f &a, v4f &b) {
        const int64_t t = s * 4 + row;
        if (s < NS && t < NT) {
            const v4f *p = reinterpret_cast<const v4f *>(X + t * 128 + r16 * 8);
            a = __builtin_nontemporal_load(p);
            b = __builtin_nontemporal_load(p + 1);
        }
    };
    v4f na = {0, 0, 0, 0}, nb = na;
    fetch(s0, na, nb);
    for (int64_t s = s0; s < NS; s += ss) {
        const v4f a = na, b = nb;
        fetch(s + ss, na, nb);
        const int64_t t = s * 4 + row;
        if (t < NT) {                               // row-uniform
            const double x00 = (double)a.x - m0, x01 = (double)a.y - m1, x10 = (double)a.z - m0, x11 = (double)a.w - m1;
            const double x20 = (double)b.x - m0, x21 = (double)b.y - m1, x30 = (double)b.z - m0, x31 = (double)b.w - m1;
            sq0 += x00 * x00; sq0 += x10 * x10; sq0 += x20 * x20; sq0 += x30 * x30;
            sq1 += x01 * x01; sq1 += x11 * x11; sq1 += x21 * x21; sq1 += x31 * x31;
            const float lu = row_min(fminf(fminf(a.x, a.z), fminf(b.x, b.z)));
            const float lv = row_min(fminf(fminf(a.y, a.w), fminf(b.y, b.w)));
            const float hu = row_max(fmaxf(fmaxf(a.x, a.z), fmaxf(b.x, b.z)));
            const float hv = row_max(fmaxf(fmaxf(a.y, a.w), fmaxf(b.y, b.w)));
            const double su = row_sum((x00 + x10) + (x20 + x30));
            const double sv = row_sum((x01 + x11) + (x21 + x31));
            // scatter about the tile's own mean (exact division by 64): what the final E-step needs for a skipped tile
            const double mu = su * 0.015625, mv = sv * 0.015625;
            double sc = (x00 - mu) * (x00 - mu) + (x01 - mv) * (x01 - mv);
            sc += (x10 - mu) * (x10 - mu) + (x11 - mv) * (x11 - mv);
            sc += (x20 - mu) * (x20 - mu) + (x21 - mv) * (x21 - mv);
            sc += (x30 - mu) * (x30 - mu) + (x31 - mv) * (x31 - mv);
            sc = row_sum(sc);
            if (r16 == 0) {
                v4f bx = {lu, lv, hu, hv};
                // a NaN anywhere in the tile poisons its sums: such a tile never passes the box test
                if (!(su == su && sv == sv)) bx = v4f{__builtin_inff(), __builtin_inff(), -__builtin_inff(), -__builtin_inff()};
                box[t] = bx;
                tsum[t] = v2d{su, sv};
                tsq[t] = sc;
            }
        }
    }
    if (blockIdx.x == 0 && tid < (int)(N - NT * 64)) {          // the samples behind the last full tile
        const int64_t i = NT * 64 + tid;
        const double x0 = (double)X[i * 2] - m0, x1 = (double)X[i * 2 + 1] - m1;
        sq0 += x0 * x0;
        sq1 += x1 * x1;
    }
    for (int off = 32; off >= 1; off >>= 1) {
        sq0 += __shfl_down(sq0, off, 64);
        sq1 += __shfl_down(sq1, off, 64);
    }
    if (lane == 0) { s_red[wave][0] = sq0; s_red[wave][1] = sq1; }
    __syncthreads();
    if (tid < 2) partial[(size_t)blockIdx.x * 2 + tid] = ((s_red[0][tid] + s_red[1][tid]) + s_red[2][tid]) + s_red[3][tid];
}

enum { TILES_SWEEP = 0, TILES_FINAL = 1 };

// WHAT = TILES_SWEEP: one label-less Lloyd iteration in st->prune_mode (full / pruned / probe).  meta != nullptr marks
//        iteration 0, which owes the record the column sums of (x - mean)^2: k_tile_meta's reduced record `meta` when the
//        metadata pass ran, formed here sample by sample when k_tile_decide switched the tile test off.
// WHAT = TILES_FINAL: the final E-step (labels + inertia, _kmeans.py:736-744): a tile inside one cell gets 64 equal label
//        bytes and adds  sum |x - c_j|^2 = scatter(tile) + 64 |mean(tile) - c_j|^2  (both terms >= 0: no cancellation) without
//        being read; the others are walked by sample with sklearn's per-sample distance.  partial[block] = inertia share.
template <int KMAX, int WHAT>
__global__ __launch_bounds__(256) void k_lloyd_tiles(const float *__restrict__ X, int64_t N,
                                                     const LloydState *__restrict__ st, const v4f *__restrict__ box,
                                                     const v2d *__restrict__ tsum, const double *__restrict__ tsq,
                                                     uint8_t *__restrict__ labels, double *__restrict__ partial,
                                                     const double *__restrict__ meta)
{
    constexpr int D = 2, k = KMAX;
    constexpr int NV = KMAX * D + KMAX + LLOYD_REC_EXTRA;
    constexpr bool FINAL = WHAT == TILES_FINAL, ACCUM = !FINAL;
    if (ACCUM && st->halt) return;      // speculatively enqueued behind the iteration that converged (uniform)
    const int mode = ACCUM ? st->prune_mode : LLOYD_TILES_PRUNED;
    const bool own_sq = ACCUM && meta && st->prune_policy == LLOYD_PRUNE_OFF;      // uniform
    double sq0 = 0, sq1 = 0;
    extern __shared__ __align__(16) unsigned char smem[];
    double *sacc = reinterpret_cast<double *>(smem);                             // [k*D][256]
    unsigned *scnt = reinterpret_cast<unsigned *>(sacc + (size_t)k * D * 256);   // [k][256]
    __shared__ unsigned s_tiles[4][2];
    __shared__ double s_cen[KMAX][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, row = lane >> 4, r16 = lane & 15;
    if (ACCUM) {
        for (int i = 0; i < k * D; i++) sacc[i * 256 + tid] = 0.0;
        for (int j = 0; j < k; j++) scnt[j * 256 + tid] = 0u;
    }
    TileCtx<KMAX> cx;
    cx.load(st, s_cen);
    double inert = 0;
    unsigned n_tested = 0, n_pure = 0;
    auto accumulate = [&](int l, double x0, double x1, unsigned n) {
        sacc[(l * D) * 256 + tid] += x0;
        sacc[(l * D + 1) * 256 + tid] += x1;
        scnt[l * 256 + tid] += n;
    };
    // four consecutive samples of tile t (this lane's quad of its row's tile)
    auto walk = [&](int64_t t, const v4f a, const v4f b) {
        double x[4][D];
        x[0][0] = a.x; x[0][1] = a.y; x[1][0] = a.z; x[1][1] = a.w;
        x[2][0] = b.x; x[2][1] = b.y; x[3][0] = b.z; x[3][1] = b.w;
        int nl[4];
#pragma unroll
        for (int p = 0; p < 4; p++) {
            x[p][0] -= cx.m[0];
            x[p][1] -= cx.m[1];
            nl[p] = cx.label_of(x[p][0], x[p][1]);
        }
        if (ACCUM) {
#pragma unroll
            for (int p = 0; p < 4; p++) accumulate(nl[p], x[p][0], x[p][1], 1u);
            if (own_sq) {
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    sq0 += x[p][0] * x[p][0];
                    sq1 += x[p][1] * x[p][1];
                }
            }
        }
        if (FINAL) {
#pragma unroll
            for (int p = 0; p < 4; p++) inert += cx.sq_dist(x[p][0], x[p][1], nl[p]);
            __builtin_nontemporal_store((unsigned)(nl[0] | (nl[1] << 8) | (nl[2] << 16) | (nl[3] << 24)),
                                        reinterpret_cast<unsigned *>(labels) + t * 16 + r16);
        }
    };

    const int64_t NT = N >> 6;                      // full tiles; the < 64 samples behind them: block 0, below
    const int64_t NG = (NT + 63) >> 6;              // groups of 64 tiles: one tile per lane in the box test
    const int64_t wg = (int64_t)blockIdx.x * 4 + wave, nw = (int64_t)gridDim.x * 4;
    for (int64_t g = wg; g < NG; g += nw) {
        const int64_t t = g * 64 + lane;
        const bool valid = t < NT;
        unsigned long long todo;                    // tiles of this group that are walked by sample (wave-uniform)
        if (mode == LLOYD_TILES_FULL) {
            todo = __ballot(valid);
        } else {
            int j = -1;
            if (valid) {
                const v4f b = box[t];
                j = cx.box_label((double)b.x - cx.m[0], (double)b.y - cx.m[1], (double)b.z - cx.m[0], (double)b.w - cx.m[1]);
                n_tested += 1u;
                n_pure += (j >= 0);
            }
            const bool skip = j >= 0 && mode == LLOYD_TILES_PRUNED;
            if (skip) {
                const v2d s = tsum[t];
                if (ACCUM) accumulate(j, s.x, s.y, 64u);
                if (FINAL) {
#pragma clang fp contract(off)
                    double cjx, cjy, cnj;
                    cx.centre_of(j, cjx, cjy, cnj);
                    const double du = s.x * 0.015625 - cjx, dv = s.y * 0.015625 - cjy;
                    inert += tsq[t] + 64.0 * (du * du + dv * dv);
                    const unsigned w4 = (unsigned)j * 0x01010101u;
                    typedef unsigned v4u __attribute__((ext_vector_type(4)));
                    v4u *lp = reinterpret_cast<v4u *>(labels + t * 64);
#pragma unroll
                    for (int q = 0; q < 4; q++) __builtin_nontemporal_store((v4u){w4, w4, w4, w4}, lp + q);
                }
            }
            todo = __ballot(valid && !skip);
        }
        // walk the remaining tiles: 4 per step (one per 16-lane row), two steps in flight
        while (todo) {
            int ta[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (todo) {
                    ta[i] = __builtin_ctzll(todo);
                    todo &= todo - 1;
                } else {
                    ta[i] = -1;
                }
            }
            const int tA = row == 0 ? ta[0] : row == 1 ? ta[1] : row == 2 ? ta[2] : ta[3];
            const int tB = row == 0 ? ta[4] : row == 1 ? ta[5] : row == 2 ? ta[6] : ta[7];
            const int64_t tileA = g * 64 + tA, tileB = g * 64 + tB;
            v4f a0 = {0, 0, 0, 0}, a1 = a0, b0 = a0, b1 = a0;
            if (tA >= 0) {
                const v4f *p = reinterpret_cast<const v4f *>(X + tileA * 128 + r16 * 8);
                a0 = __builtin_nontemporal_load(p);
                a1 = __builtin_nontemporal_load(p + 1);
            }
            if (tB >= 0) {
                const v4f *p = reinterpret_cast<const v4f *>(X + tileB * 128 + r16 * 8);
                b0 = __builtin_nontemporal_load(p);
                b1 = __builtin_nontemporal_load(p + 1);
            }
            if (tA >= 0) walk(tileA, a0, a1);
            if (tB >= 0) walk(tileB, b0, b1);
        }
    }
    // the samples behind the last full tile
    if (blockIdx.x == 0 && tid < (int)(N - NT * 64)) {
        const int64_t i = NT * 64 + tid;
        const double x0 = (double)X[i * 2] - cx.m[0], x1 = (double)X[i * 2 + 1] - cx.m[1];
        const int l = cx.label_of(x0, x1);
        if (ACCUM) accumulate(l, x0, x1, 1u);
        if (own_sq) {
            sq0 += x0 * x0;
            sq1 += x1 * x1;
        }
        if (FINAL) {
            inert += cx.sq_dist(x0, x1, l);
            labels[i] = (uint8_t)l;
        }
    }
    if (FINAL) {                                    // fixed order: shuffle tree, then waves 0..3
        __shared__ double lds_in[4];
        for (int off = 32; off >= 1; off >>= 1) inert += __shfl_down(inert, off, 64);
        if (lane == 0) lds_in[wave] = inert;
        __syncthreads();
        if (tid == 0) partial[blockIdx.x] = ((lds_in[0] + lds_in[1]) + lds_in[2]) + lds_in[3];
        return;
    }
    // ---- record: fold the 256 private columns in a fixed order (as k_lloyd_assign) ----
    double *rec = partial + (size_t)blockIdx.x * NV;
    if (tid < NV) rec[tid] = 0.0;
    for (int off = 32; off >= 1; off >>= 1) {
        n_tested += __shfl_down(n_tested, off, 64);
        n_pure += __shfl_down(n_pure, off, 64);
    }
    if (lane == 0) { s_tiles[wave][0] = n_tested; s_tiles[wave][1] = n_pure; }
    __syncthreads();
    for (int i = wave; i < k * D + k; i += 4) {
        double a;
        if (i < k * D) {
            const double *col = sacc + (size_t)i * 256;
            a = ((col[lane] + col[lane + 64]) + col[lane + 128]) + col[lane + 192];
        } else {
            const unsigned *col = scnt + (size_t)(i - k * D) * 256;
            a = (double)col[lane] + (double)col[lane + 64] + (double)col[lane + 128] + (double)col[lane + 192];
        }
        for (int off = 32; off >= 1; off >>= 1) a += __shfl_down(a, off, 64);
        if (lane == 0) rec[i < k * D ? i : KMAX * D + (i - k * D)] = a;
    }
    if (tid == 0) {
        rec[KMAX * D + KMAX + 1 + LLOYD_DMAX] = (double)(s_tiles[0][0] + s_tiles[1][0] + s_tiles[2][0] + s_tiles[3][0]);
        rec[KMAX * D + KMAX + 2 + LLOYD_DMAX] = (double)(s_tiles[0][1] + s_tiles[1][1] + s_tiles[2][1] + s_tiles[3][1]);
    }
    // iteration 0: sum (x-mean)^2 per column for sklearn's tol: formed by k_tile_meta (rides in work-group 0's record) or,
    // with the tile test switched off, by this sweep
    if (own_sq) {
        __shared__ double lds_sq[4][2];
        for (int off = 32; off >= 1; off >>= 1) {
            sq0 += __shfl_down(sq0, off, 64);
            sq1 += __shfl_down(sq1, off, 64);
        }
        if (lane == 0) { lds_sq[wave][0] = sq0; lds_sq[wave][1] = sq1; }
        __syncthreads();
        if (tid < D) rec[KMAX * D + KMAX + 1 + tid] = ((lds_sq[0][tid] + lds_sq[1][tid]) + lds_sq[2][tid]) + lds_sq[3][tid];
    } else if (meta && blockIdx.x == 0 && tid >= 64 && tid < 64 + D) {
        rec[KMAX * D + KMAX + 1 + (tid - 64)] = meta[tid - 64];
    }
}

bool lloyd_tiles_supported(int dtype, int d, int k) { return dtype == OFC_F32 && d == 2 && k >= 1 && k <= 8; }

template <int KMAX>
static void launch_tiles_k(const float *X, int64_t N, const LloydState *st, void *box, void *tsum, void *tsq, uint8_t *labels,
                           double *partial, int nblocks, int what, const double *meta, hipStream_t s)
{
    const size_t lds = (size_t)KMAX * (8 * 2 + 4) * 256;
    v4f *b = (v4f *)box;
    v2d *ts = (v2d *)tsum;
    double *tq = (double *)tsq;
    if (what == LLOYD_WHAT_PROBE) {
        const int pb = std::min(nblocks, 256);
        hipLaunchKernelGGL((k_tile_probe<KMAX>), dim3(pb), dim3(256), 0, s, X, N, st, partial);
        hipLaunchKernelGGL(k_tile_decide, dim3(1), dim3(64), 0, s, const_cast<LloydState *>(st), partial, pb);
    } else if (what == LLOYD_WHAT_META)
        hipLaunchKernelGGL(k_tile_meta, dim3(nblocks), dim3(256), 0, s, X, N, st, b, ts, tq, partial);
    else if (what == LLOYD_WHAT_FINAL)
        hipLaunchKernelGGL((k_lloyd_tiles<KMAX, TILES_FINAL>), dim3(nblocks), dim3(256), 0, s, X, N, st, b, ts, tq, labels, partial, nullptr);
    else
        hipLaunchKernelGGL((k_lloyd_tiles<KMAX, TILES_SWEEP>), dim3(nblocks), dim3(256), lds, s, X, N, st, b, ts, tq, labels, partial, meta);
}

int launch_lloyd_tiles(const float *X, int64_t N, int k, const LloydState *st, void *box, void *tsum, void *tsq,
                       uint8_t *labels, double *partial, int nblocks, int what, const double *meta, hipStream_t s)
{
    switch (k) {
    case 1: launch_tiles_k<1>(X, N, st, box, tsum, tsq, labels, partial, nblocks, what, meta, s); break;
    case 2: launch_tiles_k<2>(X, N, st, box, tsum, tsq, labels, partial, nblocks, what, meta, s); break;
    case 3: launch_tiles_k<3>(X, N, st, box, tsum, tsq, labels, partial, nblocks, what, meta, s); break;
    case 4: launch_tiles_k<4>(X, N, st, box, tsum, tsq, labels, partial, nblocks, what, meta, s); break;
    case 5: launch_tiles_k<5>(X, N, st, box, tsum, tsq, labels, partial, nblocks, what, meta, s); break;
    case 6: launch_tiles_k<6>(X, N, st, box, tsum, tsq, labels, partial, nblocks, what, meta, s); break;
    case 7: launch_tiles_k<7>(X, N, st, box, tsum, tsq, labels, partial, nblocks, what, meta, s); break;
    case 8: launch_tiles_k<8>(X, N, st, box, tsum, tsq, labels, partial, nblocks, what, meta, s); break;
    default: set_error("k=%d outside the tile sweep's range (1..8)", k); return OFC_EUNSUPPORTED;
    }
    OFC_HIP(hipGetLastError());
    return OFC_OK;
}

}  // namespace ofc
