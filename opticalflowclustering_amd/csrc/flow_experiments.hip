// flow_experiments.hip -- the round-2 experiments on the Farneback iteration, kept with their parity tests as the record of
// what was measured ("Round 2"): k_flow_iter2 (two iterations per launch, OFC_FLOW_FUSE2) and
// k_flow_iter_w3 (3 waves per SIMD, OFC_FLOW_W3); and round 3's k_flow_iter_p (gathers software-pipelined across steps, ring
// partly in LDS, OFC_FLOW_PIPE).  All are SLOWER than k_flow_iter on MI355X and never run unless their switch is set;
// flow_kernels.hip holds what ships.
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

// ------------------------------------------------------------------------------------------------
// k_flow_iter, software-pipelined across steps (round 3; winsize 15, flow_in at this level's size).
//
// k_flow_iter's wave alternates between a memory phase (36 gathers of a 4-row step: issue, wait) and a compute phase
// (matrix arithmetic, exchange, f64 horizontal sums, solve), and with the 80-VGPR ring only two waves fit a SIMD: a wave's
// step lasts (memory + compute), and tools/gatherbench reproduces that additive law at this occupancy whatever the layout
// of R.  Here the gathers of the NEXT step's rows 0,1 leave before this step's exchange and horizontal pass, rows 2,3
// right after it, so that at most two rows of operands (56 VGPRs) are in flight while the horizontal pass needs its ~76
// temporaries.  The registers for that come from the ring: 7 of its 16 slots live in LDS (35 KB, beside the 43.5 KB
// exchange buffer: still two work-groups per CU).  Every slot index is a compile-time constant after unrolling.
// Arithmetic, operand order and summation order are k_flow_iter's: the output is bit-identical (tested).
//
// RESULT (MI355X, 32 x 1080p pairs, level 0): 1.81-1.92 ms per iteration against k_flow_iter's 0.72-0.75.  The schedule does
// not fit two waves per SIMD: with four rows of operands live across the step boundary (112 VGPRs), 45 ring registers, the
// f64 window sums, three generations of flow vectors and the matrix arithmetic's temporaries the allocator spills 90-110
// dwords per lane in every variant tried (two rows early / one row early / none early, channels of the horizontal pass
// serialised, fractions recomputed instead of carried), and the scratch traffic of the spills exceeds the gather traffic
// it was meant to hide.  Opt-in only (OFC_FLOW_PIPE=1); tools/flow_pipe_ab.py reproduces the comparison.
// ------------------------------------------------------------------------------------------------
constexpr int FP_NL = 7, FP_NR = 16 - FP_NL;      // ring slots FP_NR..15 in LDS, 0..FP_NR-1 in registers

template <bool SUMS>
__global__ __launch_bounds__(256, 2) void k_flow_iter_p(const float *__restrict__ Rb, size_t frame_stride_R,
                                                        const float *__restrict__ flow_inb,
                                                        float *__restrict__ flow_outb, int W, int H,
                                                        int rows_per_block /* multiple of 16 */, int tiles_x,
                                                        int n_strips, int npair, double *__restrict__ sums)
{
    constexpr int M = 7;
    constexpr int TXO = 256 - 2 * M;
    constexpr int NV = 2 * M + 4;
    constexpr int NV2 = (NV + 1) / 2;
    constexpr int PLD = 136, PITCH = 2 * PLD;
    extern __shared__ __align__(16) unsigned char fp_smem[];
    double (*vs)[BS_ROWS][PITCH] = reinterpret_cast<double (*)[BS_ROWS][PITCH]>(fp_smem);                       // [5][4][272]
    float (*ringL)[5][256] = reinterpret_cast<float (*)[5][256]>(fp_smem + sizeof(double) * 5 * BS_ROWS * PITCH);   // [FP_NL][5][256]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles = tiles_x * n_strips;
    const int group = blockIdx.x / (8 * npair), rem = blockIdx.x - group * (8 * npair);
    const int pair = rem >> 3, tile = group * 8 + (rem & 7);
    if (tile >= tiles) {
        if (SUMS && threadIdx.x < 2) sums[(size_t)blockIdx.x * 2 + threadIdx.x] = 0.0;     // padding work-group of the grid
        return;
    }
    double su = 0, sv = 0;
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
    const int vs_w = ((tid >> 1) & 1) * PLD + 2 * (tid >> 2) + (tid & 1);
    const double scale = 1.0 / ((2 * M + 1) * (2 * M + 1));

    float ringR[FP_NR][5];
    double v[5] = {0, 0, 0, 0, 0};
    auto ring_put = [&](int slot, const float (&m)[5]) {         // slot: a constant after unrolling
#pragma unroll
        for (int c = 0; c < 5; c++) {
            if (slot < FP_NR) ringR[slot < FP_NR ? slot : 0][c] = m[c];
            else ringL[slot >= FP_NR ? slot - FP_NR : 0][c][tid] = m[c];
        }
    };
    auto ring_get = [&](int slot, int c) -> float {
        return slot < FP_NR ? ringR[slot < FP_NR ? slot : 0][c] : ringL[slot >= FP_NR ? slot - FP_NR : 0][c][tid];
    };
    auto in_row = [&](int yc, int r) { return min(yc + r + 1 + M, H - 1); };     // row entering the window with output row yc + r

    // ---- warm-up: rows y_begin-M .. y_begin+M (replicate-clamped), four at a time, as k_flow_iter ----
#pragma unroll
    for (int j4 = -M; j4 <= M; j4 += BS_ROWS) {
        float2 flw[BS_ROWS];
        UmIn uw[BS_ROWS];
        __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int q = 0; q < BS_ROWS; q++) flw[q] = flow_in[(size_t)min(max(y_begin + j4 + q, 0), H - 1) * W + xc];
#pragma unroll
        for (int q = 0; q < BS_ROWS; q++)
            um_load(R0, R1, plane, W, H, xc, min(max(y_begin + j4 + q, 0), H - 1), flw[q], uw[q]);
        __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int q = 0; q < BS_ROWS; q++) {
            if (j4 + q <= M) {
                const int row = min(max(y_begin + j4 + q, 0), H - 1);
                float m[5];
                um_math(uw[q], W, H, xc, row, flw[q], m);
                ring_put((j4 + q + 16) & 15, m);
#pragma unroll
                for (int c = 0; c < 5; c++) v[c] += (double)m[c];
            }
        }
    }

    // ---- prologue of the pipeline: the first step's gathers in full, the flow vectors of the two steps behind it ----
    float2 fl[BS_ROWS], fn[BS_ROWS], f2[BS_ROWS];
    UmIn u[BS_ROWS];
#pragma unroll
    for (int r = 0; r < BS_ROWS; r++) fl[r] = flow_in[(size_t)in_row(y_begin, r) * W + xc];
#pragma unroll
    for (int r = 0; r < BS_ROWS; r++) fn[r] = flow_in[(size_t)in_row(y_begin + BS_ROWS, r) * W + xc];
    __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int r = 0; r < BS_ROWS; r++) um_load(R0, R1, plane, W, H, xc, in_row(y_begin, r), fl[r], u[r]);
    __builtin_amdgcn_s_setprio(0);

    for (int y16 = y_begin; y16 < y_end; y16 += 16) {
#pragma unroll
        for (int q4 = 0; q4 < 4; q4++) {
            const int yc = y16 + 4 * q4;
            if (yc < y_end) {                                   // uniform
                const bool more = yc + BS_ROWS < y_end;         // uniform: another step follows in this strip
                // the barrier that frees `vs` from the previous step's readers comes first here (k_flow_iter has it behind
                // the matrix arithmetic so that an early wave can start its gathers: here those are in flight already):
                // every row's matrix entries then go straight into the window sums, the exchange buffer and the ring
                __syncthreads();
                __builtin_amdgcn_s_setprio(2);
                auto consume = [&](int r) {                     // r: a constant after unrolling
                    float mi[5];
                    um_refrac(u[r], W, H, xc, in_row(yc, r), fl[r]);
                    um_math(u[r], W, H, xc, in_row(yc, r), fl[r], mi);
                    const int s_in = (4 * q4 + r + 1 + M) & 15, s_out = (4 * q4 + r + 16 - M) & 15;
#pragma unroll
                    for (int c = 0; c < 5; c++) {
                        vs[c][r][vs_w] = v[c];
                        v[c] += (double)mi[c] - (double)ring_get(s_out, c);
                    }
                    ring_put(s_in, mi);
                };
                // rows 0,1: requested before the previous step's exchange, long since here
                consume(0);
                consume(1);
                __builtin_amdgcn_sched_barrier(0);
                if (more) {     // their registers take the next step's rows 0,1 at once; the flow vectors two steps ahead follow
                    __builtin_amdgcn_s_setprio(3);
                    um_load(R0, R1, plane, W, H, xc, in_row(yc + BS_ROWS, 0), fn[0], u[0]);
                    um_load(R0, R1, plane, W, H, xc, in_row(yc + BS_ROWS, 1), fn[1], u[1]);
#pragma unroll
                    for (int r = 0; r < BS_ROWS; r++) f2[r] = flow_in[(size_t)in_row(yc + 2 * BS_ROWS, r) * W + xc];
                    __builtin_amdgcn_s_setprio(2);
                }
                __builtin_amdgcn_sched_barrier(0);
                // rows 2,3: requested after the previous step's horizontal pass
                consume(2);
                consume(3);
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();
                __builtin_amdgcn_s_setprio(0);
                const int y = yc + wave;
                const int xo = x0 + 4 * lane;
                if (y < y_end && 4 * lane < TXO && xo < W) {
                    double S[5][4];
#pragma unroll
                    for (int c = 0; c < 5; c++) {
                        double a[2 * NV2];
#pragma unroll
                        for (int q = 0; q < NV2; q++) {
                            double2 d = *reinterpret_cast<const double2 *>(&vs[c][wave][(q & 1) * PLD + 2 * (lane + (q >> 1))]);
                            a[2 * q] = d.x; a[2 * q + 1] = d.y;
                        }
                        double s = a[0];
#pragma unroll
                        for (int q = 1; q <= 2 * M; q++) s += a[q];
                        S[c][0] = s;
#pragma unroll
                        for (int o = 1; o < 4; o++) {
                            s += a[2 * M + o] - a[o - 1];
                            S[c][o] = s;
                        }
                        // one channel's 18 window values at a time: two rows of the next step's operands are in flight in
                        // registers meanwhile, and the scheduler would otherwise request all five channels' windows up front
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    float2 fo[4];
#pragma unroll
                    for (int o = 0; o < 4; o++) {
                        const double g11 = S[0][o] * scale, g12 = S[1][o] * scale, g22 = S[2][o] * scale,
                                     h1 = S[3][o] * scale, h2 = S[4][o] * scale;
                        const double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
                        fo[o] = make_float2((float)((g11 * h2 - g12 * h1) * idet), (float)((g22 * h1 - g12 * h2) * idet));
                    }
                    if (SUMS) {
#pragma unroll
                        for (int o = 0; o < 4; o++)
                            if (4 * lane + o < TXO && xo + o < W) { su += (double)fo[o].x; sv += (double)fo[o].y; }
                    }
                    float2 *dst = flow_out + (size_t)y * W + xo;
                    if (4 * lane + 3 < TXO && xo + 3 < W && (W & 1) == 0) {      // 32 contiguous, 16-B aligned bytes
                        reinterpret_cast<float4 *>(dst)[0] = make_float4(fo[0].x, fo[0].y, fo[1].x, fo[1].y);
                        reinterpret_cast<float4 *>(dst)[1] = make_float4(fo[2].x, fo[2].y, fo[3].x, fo[3].y);
                    } else {
#pragma unroll
                        for (int o = 0; o < 4; o++)
                            if (4 * lane + o < TXO && xo + o < W) dst[o] = fo[o];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (more) {                                     // the next step's rows 2,3
                    __builtin_amdgcn_s_setprio(3);
                    um_load(R0, R1, plane, W, H, xc, in_row(yc + BS_ROWS, 2), fn[2], u[2]);
                    um_load(R0, R1, plane, W, H, xc, in_row(yc + BS_ROWS, 3), fn[3], u[3]);
                    __builtin_amdgcn_s_setprio(0);
#pragma unroll
                    for (int r = 0; r < BS_ROWS; r++) { fl[r] = fn[r]; fn[r] = f2[r]; }
                }
            }
        }
    }
    if (SUMS) {                 // fixed order: shuffle tree inside the wave, then waves 0..3 (as k_flow_iter)
        for (int off = 32; off >= 1; off >>= 1) {
            su += __shfl_down(su, off, 64);
            sv += __shfl_down(sv, off, 64);
        }
        __syncthreads();
        if (lane == 0) { vs[0][0][2 * wave] = su; vs[0][0][2 * wave + 1] = sv; }
        __syncthreads();
        if (tid < 2) sums[(size_t)blockIdx.x * 2 + tid] = ((vs[0][0][tid] + vs[0][0][2 + tid]) + vs[0][0][4 + tid]) + vs[0][0][6 + tid];
    }
}

constexpr size_t FP_LDS = sizeof(double) * 5 * BS_ROWS * 272 + sizeof(float) * FP_NL * 5 * 256;

int launch_flow_iter_pipe(const float *R, size_t frame_stride_R, const float *flow_in, float *flow_out, int npair, int W,
                          int H, int rows_per_block, hipStream_t s, double *uv_sum, double *uv_scratch,
                          size_t uv_scratch_doubles)
{
    static bool attr_set = false;       // 79,360 B of dynamic LDS: above the 64 KB default
    if (!attr_set) {
        OFC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_flow_iter_p<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FP_LDS));
        OFC_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_flow_iter_p<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)FP_LDS));
        attr_set = true;
    }
    const int tx = cdiv(W, 256 - 14), ns = cdiv(H, rows_per_block);
    const int grid = cdiv(tx * ns, 8) * 8 * npair;
    if (uv_sum) {
        if ((size_t)grid * 2 > uv_scratch_doubles) { set_error("flow sums: scratch too small (%d work-groups)", grid); return OFC_EINVAL; }
        hipLaunchKernelGGL((k_flow_iter_p<true>), dim3(grid), dim3(256), FP_LDS, s, R, frame_stride_R, flow_in, flow_out, W, H,
                           rows_per_block, tx, ns, npair, uv_scratch);
        OFC_HIP(hipGetLastError());
        return launch_reduce_records(uv_scratch, grid, 2, uv_sum, s, nullptr);
    }
    hipLaunchKernelGGL((k_flow_iter_p<false>), dim3(grid), dim3(256), FP_LDS, s, R, frame_stride_R, flow_in, flow_out, W, H,
                       rows_per_block, tx, ns, npair, nullptr);
    OFC_HIP(hipGetLastError());
    return OFC_OK;
}

}  // namespace ofc
