// flow_experiments.hip -- the round-2 experiments on the Farneback iteration, kept with their parity tests as the record of
// what was measured ("Round 2"): k_flow_iter2 (two iterations per launch, OFC_FLOW_FUSE2) and
// k_flow_iter_w3 (3 waves per SIMD, OFC_FLOW_W3).  Both are SLOWER than k_flow_iter on MI355X and never run unless their
// switch is set; flow_kernels.hip holds what ships.
#include "flow_device.h"

namespace ofc {

// ------------------------------------------------------------------------------------------------
// K4+K5 x2: TWO consecutive Farneback iterations  flow_in -> (flow_mid) -> flow_out  in one launch, so that the second
// iteration finds R0/R1 in the L2 instead of in HBM (the three iterations of a level each used to stream both frames' R:
// 2/3 of k_flow_iter's traffic).  flow_mid never exists in memory.
//
// Work-group = 2*C threads: waves [0, C/64) are stage A (first iteration), waves [C/64, C/32) stage B (second), thread <->
// column of a C-column tile (C - 4m final outputs + 2m halo columns per side) exactly as in k_flow_iter: 16-row ring of M in
// registers, f64 vertical running sums, LDS exchange for the horizontal sums, regularised 2x2 solve.  The march advances
// TWO rows per step; stage A emits its flow rows into a 16-row ring in LDS ("link"), stage B runs LAG = 10 steps behind
// (its warm-up needs A's first 16 rows), reads its flow vectors from the link and writes the final rows.  In steady state
// B ingests the rows A ingested 12 rows earlier: a 256-column tile keeps ~(12 + 4 + |flow|) rows x 2 frames x 5 KB = 90 KB
// of R per work-group alive, 32 work-groups (32 consecutive pairs of one tile, sharing frames) per XCD = 3 MB of its 4 MB L2.
//
// Every wave runs the same phases in lockstep (one work-group per CU: 120 KB LDS, 2 waves per SIMD), so memory latency is
// hidden by software pipelining instead of by a second work-group: the gathers of step j+1 are issued right after the
// step's barrier and land during the horizontal pass; the vertical sums cross LDS double-buffered (one barrier per step).
//
// Arithmetic per iteration is that of k_flow_iter (um_load / um_math, exact f64 sums, same solve); the horizontal sums are
// formed for 2 outputs per lane instead of 4, so an f64 sum may round differently in its last bit (~1e-16 relative; a
// flow value changes by one f32 ulp about once per few million pixels).
// ------------------------------------------------------------------------------------------------
constexpr int FI2_LAG = 10;

// the flow fields are streamed (each vector read once and written once per launch): non-temporal, so that they do not push
// the R rows stage B is about to re-read out of the L2
__device__ __forceinline__ float2 ld_flow(const float2 *p)
{
    typedef float v2f __attribute__((ext_vector_type(2)));
    const v2f v = __builtin_nontemporal_load(reinterpret_cast<const v2f *>(p));
    return make_float2(v.x, v.y);
}
__device__ __forceinline__ void st_flow(float2 *p, float2 v)
{
    typedef float v2f __attribute__((ext_vector_type(2)));
    __builtin_nontemporal_store((v2f){v.x, v.y}, reinterpret_cast<v2f *>(p));
}

template <int M, int C>
__global__ __launch_bounds__(2 * C, 2) void k_flow_iter2(const float *__restrict__ Rb, size_t frame_stride_R,
                                                         const float *__restrict__ flow_inb,
                                                         float *__restrict__ flow_outb, int W, int H,
                                                         int rows_per_block /* even */, int tiles_x, int n_strips,
                                                         int npair)
{
    constexpr int NW = C / 64;                // waves per stage
    constexpr int TXO = C - 4 * M;            // final outputs per tile row
    constexpr int VP = C + 16;                // columns of a vertical-sum row (+ pad: a lane reads 16 consecutive)
    __shared__ __align__(16) double vs[2][2][5][2][VP];        // [stage][step parity][channel][row of the step][column]
    __shared__ __align__(16) float2 link[16][C];               // stage A's flow rows, slot = row & 15
    const int tid = threadIdx.x, lane = tid & 63;
    const int stage = __builtin_amdgcn_readfirstlane(tid / C);          // wave-uniform
    const int wl = __builtin_amdgcn_readfirstlane((tid / 64) % NW);     // wave within its stage
    const int c = tid % C;
    // XCD-aware work-group -> (tile, pair) map, as in k_flow_iter
    const int tiles = tiles_x * n_strips;
    const int group = blockIdx.x / (8 * npair), rem = blockIdx.x - group * (8 * npair);
    const int pair = rem >> 3, tile = group * 8 + (rem & 7);
    if (tile >= tiles) return;
    const int tile_x = tile % tiles_x, tile_y = tile / tiles_x;
    const int x0 = tile_x * TXO;
    const int y_begin = tile_y * rows_per_block;
    const int y_end = min(y_begin + rows_per_block, H);
    const size_t plane = (size_t)W * H;
    const float *R0 = Rb + (size_t)pair * frame_stride_R;
    const float *R1 = R0 + frame_stride_R;
    const float2 *flow_in = reinterpret_cast<const float2 *>(flow_inb) + (size_t)pair * plane;
    float2 *flow_out = reinterpret_cast<float2 *>(flow_outb) + (size_t)pair * plane;
    const int xc = min(max(x0 - 2 * M + c, 0), W - 1);                   // this thread's (clamped) image column
    const int cl = min(max(xc - (x0 - 2 * M), M), C - M - 1);            // where stage A's flow of that column sits in a link row
    const double scale = 1.0 / ((2 * M + 1) * (2 * M + 1));

    // stage geometry: local step j ingests image rows yb - 8 + 2j + {0,1} (clamped) and, from j = 8 on, emits rows
    // yb + 2(j-8) + {0,1}.  A starts 8 rows above B (B's first ingested row is A's first emitted one); at the top of the
    // image both start at row 0 (rows above replicate row 0).
    const int yb = stage ? y_begin : max(y_begin - 8, 0);
    const int nB = 8 + (y_end - y_begin + 1) / 2;
    const int nA = 8 + (min(y_begin + 2 * (nB - 8) + 7, H - 1) - yb) / 2 + 1;
    const int nst = stage ? nB : nA;
    const int lag = stage ? FI2_LAG : 0;
    const int G = FI2_LAG + nB;

    float ring[16][5];
    double v[5] = {0, 0, 0, 0, 0};
    UmIn u[2];
    float2 fl[2], fln[2];
    auto row_of = [&](int j, int r) -> int { return min(max(yb - 8 + 2 * j + r, 0), H - 1); };
    // prologue: stage A requests the operands of its step 0 (and the flow vectors of step 1)
    if (stage == 0) {
#pragma unroll
        for (int r = 0; r < 2; r++) fl[r] = ld_flow(flow_in + (size_t)row_of(0, r) * W + xc);
#pragma unroll
        for (int r = 0; r < 2; r++) um_load(R0, R1, plane, W, H, xc, row_of(0, r), fl[r], u[r]);
#pragma unroll
        for (int r = 0; r < 2; r++) fln[r] = ld_flow(flow_in + (size_t)row_of(1, r) * W + xc);
    }

    for (int g0 = 0; g0 < G; g0 += 8) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int g = g0 + q;
            if (g < G) {                                       // uniform
                const int j = g - lag;
                const bool act = j >= 0 && j < nst;            // wave-uniform
                if (act) {
                    float mi[2][5];
#pragma unroll
                    for (int r = 0; r < 2; r++) um_math(u[r], W, H, xc, row_of(j, r), fl[r], mi[r]);
                    if (j >= 8) {
#pragma unroll
                        for (int r = 0; r < 2; r++) {
                            const int s_in = (2 * q + r) & 15, s_out = (2 * q + r + 1) & 15;
#pragma unroll
                            for (int ch = 0; ch < 5; ch++) {
                                vs[stage][q & 1][ch][r][c] = v[ch];
                                v[ch] += (double)mi[r][ch] - (double)ring[s_out][ch];
                            }
#pragma unroll
                            for (int ch = 0; ch < 5; ch++) ring[s_in][ch] = mi[r][ch];
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 2; r++) {
                            const int s_in = (2 * q + r) & 15;
                            if (j > 0 || r > 0) {              // the very first row lies outside the first window
#pragma unroll
                                for (int ch = 0; ch < 5; ch++) v[ch] += (double)mi[r][ch];
                            }
#pragma unroll
                            for (int ch = 0; ch < 5; ch++) ring[s_in][ch] = mi[r][ch];
                        }
                    }
                }
                __syncthreads();
                // ---- operands of the next step: in flight during the horizontal pass ----
                if (j + 1 >= 0 && j + 1 < nst) {
                    if (stage == 0) {
#pragma unroll
                        for (int r = 0; r < 2; r++) fl[r] = fln[r];
                    } else {
#pragma unroll
                        for (int r = 0; r < 2; r++) fl[r] = link[row_of(j + 1, r) & 15][cl];
                    }
#pragma unroll
                    for (int r = 0; r < 2; r++) um_load(R0, R1, plane, W, H, xc, row_of(j + 1, r), fl[r], u[r]);
                    if (stage == 0) {
#pragma unroll
                        for (int r = 0; r < 2; r++) fln[r] = ld_flow(flow_in + (size_t)row_of(j + 2, r) * W + xc);
                    }
                }
                // ---- horizontal pass: 2 rows x C columns = NW waves x 64 lanes x 2 outputs ----
                if (act && j >= 8) {
                    const int item = 128 * wl + 2 * lane;
                    const int r = item / C, t = item - r * C;          // outputs t, t+1 <-> centre columns t+M, t+M+1
                    const int y = yb + 2 * (j - 8) + r;
                    double S[5][2];
#pragma unroll
                    for (int ch = 0; ch < 5; ch++) {
                        const double *base = &vs[stage][q & 1][ch][r][t];
                        double a[16];
#pragma unroll
                        for (int k = 0; k < 8; k++) {
                            const double2 d = *reinterpret_cast<const double2 *>(base + 2 * k);
                            a[2 * k] = d.x; a[2 * k + 1] = d.y;
                        }
                        double s = a[0];
#pragma unroll
                        for (int k = 1; k <= 2 * M; k++) s += a[k];
                        S[ch][0] = s;
                        s += a[2 * M + 1] - a[0];
                        S[ch][1] = s;
                    }
                    float2 fo[2];
#pragma unroll
                    for (int o = 0; o < 2; o++) {
                        const double g11 = S[0][o] * scale, g12 = S[1][o] * scale, g22 = S[2][o] * scale,
                                     h1 = S[3][o] * scale, h2 = S[4][o] * scale;
                        const double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
                        fo[o] = make_float2((float)((g11 * h2 - g12 * h1) * idet), (float)((g22 * h1 - g12 * h2) * idet));
                    }
                    if (stage == 0) {
                        if (t < C - 2 * M && y < H) {
                            link[y & 15][t + M] = fo[0];
                            link[y & 15][t + M + 1] = fo[1];
                        }
                    } else if (y < y_end) {
#pragma unroll
                        for (int o = 0; o < 2; o++) {
                            const int x = x0 - M + t + o;
                            if (t + o >= M && t + o < C - 3 * M && x < W) st_flow(flow_out + (size_t)y * W + x, fo[o]);
                        }
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K4+K5 fused, THREE work-groups per CU.  k_flow_iter keeps 244 VGPRs (80 of them the ring of M rows, 112 the gathered
// operands of four rows in flight) and therefore runs at 2 waves per SIMD; its counters say a wave issues instructions 39 %
// of its life, waits on memory / barriers 24 % and stalls on dependent (mostly f64) instructions 37 %, with 1.6 waves per SIMD
// on average: the SIMDs idle for lack of waves.  This form fits 168 VGPRs = 3 waves per SIMD:
//   * two rows per step (56 operand registers instead of 112),
//   * the ring split: 10 of its 16 row slots in registers, 6 in LDS ([slot][channel][column] floats: conflict-free
//     lane <-> column accesses, 30 KB), every slot index a compile-time constant after unrolling 8 steps,
//   * the horizontal pass on 2 outputs per lane (all four waves busy on the step's two rows).
// LDS 52.5 KB per work-group -> 3 per CU (157 KB).  Same arithmetic as k_flow_iter per pixel; the horizontal f64 sums
// group differently (2 outputs per lane), as in k_flow_iter2.
// ------------------------------------------------------------------------------------------------
template <int M>
__global__ __launch_bounds__(256, 3) void k_flow_iter_w3(const float *__restrict__ Rb, size_t frame_stride_R,
                                                         const float *__restrict__ flow_inb,
                                                         float *__restrict__ flow_outb, int W, int H,
                                                         int rows_per_block /* even */, int tiles_x, int n_strips,
                                                         int npair)
{
    constexpr int C = 256, TXO = C - 2 * M, VP = C + 16, RREG = 10, RLDS = 16 - RREG;
    __shared__ __align__(16) double vs[5][2][VP];
    __shared__ float lring[RLDS][5][C];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wl = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles = tiles_x * n_strips;
    const int group = blockIdx.x / (8 * npair), rem = blockIdx.x - group * (8 * npair);
    const int pair = rem >> 3, tile = group * 8 + (rem & 7);
    if (tile >= tiles) return;
    const int tile_x = tile % tiles_x, tile_y = tile / tiles_x;
    const int x0 = tile_x * TXO;
    const int y_begin = tile_y * rows_per_block;
    const int y_end = min(y_begin + rows_per_block, H);
    const size_t plane = (size_t)W * H;
    const float *R0 = Rb + (size_t)pair * frame_stride_R;
    const float *R1 = R0 + frame_stride_R;
    const float2 *flow_in = reinterpret_cast<const float2 *>(flow_inb) + (size_t)pair * plane;
    float2 *flow_out = reinterpret_cast<float2 *>(flow_outb) + (size_t)pair * plane;
    const int xc = min(max(x0 - M + tid, 0), W - 1);
    const double scale = 1.0 / ((2 * M + 1) * (2 * M + 1));
    const int nst = 8 + (y_end - y_begin + 1) / 2;       // local step j ingests rows y_begin - 8 + 2j + {0,1}; emits from j = 8

    float ring[RREG][5];
    double v[5] = {0, 0, 0, 0, 0};
    float2 fln[2];
    auto row_of = [&](int j, int r) -> int { return min(max(y_begin - 8 + 2 * j + r, 0), H - 1); };
#pragma unroll
    for (int r = 0; r < 2; r++) fln[r] = flow_in[(size_t)row_of(0, r) * W + xc];

    for (int j0 = 0; j0 < nst; j0 += 8) {
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const int j = j0 + q;
            if (j < nst) {                                     // uniform
                float mi[2][5];
                {
                    UmIn u[2];
                    float2 fl[2];
#pragma unroll
                    for (int r = 0; r < 2; r++) fl[r] = fln[r];
                    __builtin_amdgcn_s_setprio(3);
#pragma unroll
                    for (int r = 0; r < 2; r++) um_load(R0, R1, plane, W, H, xc, row_of(j, r), fl[r], u[r]);
#pragma unroll
                    for (int r = 0; r < 2; r++) fln[r] = flow_in[(size_t)row_of(j + 1, r) * W + xc];
                    __builtin_amdgcn_s_setprio(2);
#pragma unroll
                    for (int r = 0; r < 2; r++) um_math(u[r], W, H, xc, row_of(j, r), fl[r], mi[r]);
                }
                __syncthreads();                               // the previous step's horizontal pass is done with vs
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    const int s_in = (2 * q + r) & 15, s_out = (2 * q + r + 1) & 15;
                    if (j >= 8) {
                        float mo[5];
                        if (s_out < RREG) {
#pragma unroll
                            for (int ch = 0; ch < 5; ch++) mo[ch] = ring[s_out < RREG ? s_out : 0][ch];
                        } else {
#pragma unroll
                            for (int ch = 0; ch < 5; ch++) mo[ch] = lring[s_out >= RREG ? s_out - RREG : 0][ch][tid];
                        }
#pragma unroll
                        for (int ch = 0; ch < 5; ch++) {
                            vs[ch][r][tid] = v[ch];
                            v[ch] += (double)mi[r][ch] - (double)mo[ch];
                        }
                    } else if (j > 0 || r > 0) {               // warm-up; the very first row lies outside the first window
#pragma unroll
                        for (int ch = 0; ch < 5; ch++) v[ch] += (double)mi[r][ch];
                    }
                    if (s_in < RREG) {
#pragma unroll
                        for (int ch = 0; ch < 5; ch++) ring[s_in < RREG ? s_in : 0][ch] = mi[r][ch];
                    } else {
#pragma unroll
                        for (int ch = 0; ch < 5; ch++) lring[s_in >= RREG ? s_in - RREG : 0][ch][tid] = mi[r][ch];
                    }
                }
                __syncthreads();
                __builtin_amdgcn_s_setprio(0);
                if (j >= 8) {
                    const int item = 128 * wl + 2 * lane;
                    const int r = item / C, t = item - r * C;          // outputs t, t+1 <-> image columns x0 + t, x0 + t + 1
                    const int y = y_begin + 2 * (j - 8) + r;
                    if (y < y_end && t < TXO && x0 + t < W) {
                        double S[5][2];
#pragma unroll
                        for (int ch = 0; ch < 5; ch++) {
                            const double *base = &vs[ch][r][t];
                            double a[16];
#pragma unroll
                            for (int k = 0; k < 8; k++) {
                                const double2 d = *reinterpret_cast<const double2 *>(base + 2 * k);
                                a[2 * k] = d.x; a[2 * k + 1] = d.y;
                            }
                            double s = a[0];
#pragma unroll
                            for (int k = 1; k <= 2 * M; k++) s += a[k];
                            S[ch][0] = s;
                            s += a[2 * M + 1] - a[0];
                            S[ch][1] = s;
                        }
                        float2 fo[2];
#pragma unroll
                        for (int o = 0; o < 2; o++) {
                            const double g11 = S[0][o] * scale, g12 = S[1][o] * scale, g22 = S[2][o] * scale,
                                         h1 = S[3][o] * scale, h2 = S[4][o] * scale;
                            const double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
                            fo[o] = make_float2((float)((g11 * h2 - g12 * h1) * idet), (float)((g22 * h1 - g12 * h2) * idet));
                        }
                        float2 *dst = flow_out + (size_t)y * W + x0 + t;
                        if (t + 1 < TXO && x0 + t + 1 < W && (W & 1) == 0) {     // 16 contiguous, 16-B aligned bytes (x0, t even)
                            *reinterpret_cast<float4 *>(dst) = make_float4(fo[0].x, fo[0].y, fo[1].x, fo[1].y);
                        } else {
                            dst[0] = fo[0];
                            if (t + 1 < TXO && x0 + t + 1 < W) dst[1] = fo[1];
                        }
                    }
                }
            }
        }
    }
}

int flow_iter_w3_rows(int W, int H, int npair, int winsize)
{
    const int tiles_x = cdiv(W, 256 - (winsize - 1));
    const int resident = 3 * 256;
    int best_rows = cdiv(H, 2) * 2;
    int64_t best_cost = LLONG_MAX;
    for (int n = 1; n <= 64; n++) {
        const int rows = cdiv(cdiv(H, n), 2) * 2;
        if (rows < 16 && n > 1) break;
        const int64_t blocks = (int64_t)tiles_x * cdiv(H, rows) * npair;
        const int64_t cost = cdiv64(blocks, resident) * (rows + 16);
        if (cost < best_cost) { best_cost = cost; best_rows = rows; }
    }
    return best_rows;
}

int launch_flow_iter_w3(const float *R, size_t frame_stride_R, const float *flow_in, float *flow_out, int npair, int W,
                        int H, int winsize, hipStream_t s, int rows_per_block)
{
    if (winsize != 15) { set_error("k_flow_iter_w3 is built for winsize 15 only (got %d)", winsize); return OFC_EUNSUPPORTED; }
    if ((int64_t)W * H * 5 >= (1ll << 30)) { set_error("frame too large for 32-bit R offsets (%dx%d)", W, H); return OFC_EUNSUPPORTED; }
    if (rows_per_block <= 0) rows_per_block = flow_iter_w3_rows(W, H, npair, winsize);
    rows_per_block = cdiv(rows_per_block, 2) * 2;
    const int tx = cdiv(W, 256 - 14), ns = cdiv(H, rows_per_block);
    dim3 grid(cdiv(tx * ns, 8) * 8 * npair);
    hipLaunchKernelGGL((k_flow_iter_w3<7>), grid, dim3(256), 0, s, R, frame_stride_R, flow_in, flow_out, W, H,
                       rows_per_block, tx, ns, npair);
    OFC_HIP(hipGetLastError());
    return OFC_OK;
}

// strip height of the two-iteration kernel: rounds x (steps per strip), one work-group per CU
int flow_iter2_rows(int W, int H, int npair, int winsize, int C)
{
    const int tiles_x = cdiv(W, C - 2 * (winsize - 1));
    const int resident = 256;
    int best_rows = cdiv(H, 2) * 2;
    int64_t best_cost = LLONG_MAX;
    for (int n = 1; n <= 64; n++) {
        const int rows = cdiv(cdiv(H, n), 2) * 2;
        if (rows < 16 && n > 1) break;
        const int64_t blocks = (int64_t)tiles_x * cdiv(H, rows) * npair;
        const int64_t cost = cdiv64(blocks, resident) * (FI2_LAG + 8 + rows / 2);
        if (cost < best_cost) { best_cost = cost; best_rows = rows; }
    }
    return best_rows;
}

int launch_flow_iter2(const float *R, size_t frame_stride_R, const float *flow_in, float *flow_out, int npair, int W,
                      int H, int winsize, hipStream_t s, int rows_per_block)
{
    if (winsize != 15) { set_error("two-iteration kernel is built for winsize 15 only (got %d)", winsize); return OFC_EUNSUPPORTED; }
    if ((int64_t)W * H * 5 >= (1ll << 30)) { set_error("frame too large for 32-bit R offsets (%dx%d)", W, H); return OFC_EUNSUPPORTED; }
    constexpr int MM = 7, CC = 256;
    if (rows_per_block <= 0) rows_per_block = flow_iter2_rows(W, H, npair, winsize, CC);
    rows_per_block = cdiv(rows_per_block, 2) * 2;
    const int tx = cdiv(W, CC - 4 * MM), ns = cdiv(H, rows_per_block);
    dim3 grid(cdiv(tx * ns, 8) * 8 * npair);
    hipLaunchKernelGGL((k_flow_iter2<MM, CC>), grid, dim3(2 * CC), 0, s, R, frame_stride_R, flow_in, flow_out, W, H,
                       rows_per_block, tx, ns, npair);
    OFC_HIP(hipGetLastError());
    return OFC_OK;
}

}  // namespace ofc
