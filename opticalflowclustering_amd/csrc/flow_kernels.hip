// flow_kernels.hip -- hand-written gfx950 kernels of the Farneback pyramid.
//
// All kernels are batched over images / frame pairs through blockIdx.z so that even the coarse
// pyramid levels (240x135 at 1080p) fill the 256 CUs.  Layouts in HBM:
//   frames  [n][H0][W0] u8            pyramid image I [n][h][w] f32
//   R       [n][h][w][5] f32 pixel-interleaved (5 coefficients = 20 contiguous bytes: the bilinear gather of the
//                                      flow iteration reads 40-byte runs; producer stores stay fully coalesced)
//   M       [p][5][h][w] f32 planar   flow [p][h][w][2] f32 (the ABI layout of cv2's output)
//
// Arithmetic follows the oracle (oracle/farneback_ref.c == SURVEY.md App. A) statement by statement.
// Kernels that are purely memory bound keep FP contraction OFF so that they reproduce the oracle
// bit for bit; polyexp and the box filter use FMAs / exact f64 sums (differences ~1e-7 relative).
#include "flow_device.h"      // um_load / um_math; lloyd_common.h: launch_reduce_records (fixed-order sum of records)

namespace ofc {

// ------------------------------------------------------------------------------------------------
// bilinear tap helpers (imgproc resize INTER_LINEAR)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void lin_tap_x(int dx, double scale, int sw, int &s0, float &a)
{
#pragma clang fp contract(off)
    float fx = (float)((dx + 0.5) * scale - 0.5);
    int sx = (int)floorf(fx);
    fx -= (float)sx;
    if (sx < 0) { fx = 0.f; sx = 0; }
    if (sx >= sw - 1) { fx = 0.f; sx = sw - 1; }
    s0 = sx;
    a = fx;
}

__device__ __forceinline__ void lin_tap_y(int dy, double scale, int sh, int &s0, int &s1, float &b)
{
#pragma clang fp contract(off)
    float fy = (float)((dy + 0.5) * scale - 0.5);
    int sy = (int)floorf(fy);
    fy -= (float)sy;
    s0 = min(max(sy, 0), sh - 1);
    s1 = min(max(sy + 1, 0), sh - 1);
    b = fy;
}

__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

// ------------------------------------------------------------------------------------------------
// K2  blur + decimate:  u8 full-res frame -> f32 level image.
// The separable Gaussian (REFLECT_101) is evaluated ONLY at the <=2x2 full-res positions each output
// pixel's bilinear taps touch: 1/4 .. 1/16 of the work of blurring the whole frame per level.
// LDS: u8 input tile (+halo) and the row-filtered samples.  HBM-bound on the u8 read (1 B/px of the
// full-res frame per level) -- SURVEY.md 8d.
// ------------------------------------------------------------------------------------------------
struct LevelArgs {
    int W0, H0, w, h;
    int r;
    int txo, tyo;
    int in_w, in_h;      // LDS tile extents (upper bounds)
    int in_pitch;        // u8 pitch, multiple of 4
    double sx, sy;
    float kern[32];
};

__global__ __launch_bounds__(256) void k_level_image(const uint8_t *__restrict__ src,
                                                     float *__restrict__ dst, LevelArgs p)
{
#pragma clang fp contract(off)
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x;
    const int ox0 = blockIdx.x * p.txo, oy0 = blockIdx.y * p.tyo;
    const int nox = min(p.txo, p.w - ox0), noy = min(p.tyo, p.h - oy0);
    const uint8_t *img = src + (size_t)blockIdx.z * p.W0 * p.H0;
    float *out = dst + (size_t)blockIdx.z * p.w * p.h;

    // LDS carve-up
    int *xs0 = reinterpret_cast<int *>(smem);                   // [txo]
    float *xa = reinterpret_cast<float *>(xs0 + p.txo);         // [txo]
    int *ys0 = reinterpret_cast<int *>(xa + p.txo);             // [tyo]
    int *ys1 = ys0 + p.tyo;                                     // [tyo]
    float *yb = reinterpret_cast<float *>(ys1 + p.tyo);         // [tyo]
    float *h1 = yb + p.tyo;                                     // [in_h][2*txo]
    unsigned char *tile = reinterpret_cast<unsigned char *>(h1 + (size_t)p.in_h * 2 * p.txo);

    if (tid < nox) {
        int s0; float a;
        lin_tap_x(ox0 + tid, p.sx, p.W0, s0, a);
        xs0[tid] = s0; xa[tid] = a;
    }
    if (tid >= 64 && tid - 64 < noy) {
        int s0, s1; float b;
        lin_tap_y(oy0 + tid - 64, p.sy, p.H0, s0, s1, b);
        ys0[tid - 64] = s0; ys1[tid - 64] = s1; yb[tid - 64] = b;
    }
    __syncthreads();
    const int x_first = xs0[0], x_last = min(xs0[nox - 1] + 1, p.W0 - 1);
    const int y_first = ys0[0], y_last = ys1[noy - 1];
    const int ix0 = x_first - p.r, iy0 = y_first - p.r;
    const int tw = x_last - x_first + 1 + 2 * p.r, th = y_last - y_first + 1 + 2 * p.r;

    for (int i = tid; i < tw * th; i += 256) {
        int ly = i / tw, lx = i - ly * tw;
        int gx = reflect101(ix0 + lx, p.W0), gy = reflect101(iy0 + ly, p.H0);
        tile[ly * p.in_pitch + lx] = img[(size_t)gy * p.W0 + gx];
    }
    __syncthreads();

    // row pass (taps summed left to right) at the two x positions of every output column
    const int ks = 2 * p.r + 1;
    for (int i = tid; i < th * nox * 2; i += 256) {
        int ly = i / (nox * 2), rem = i - ly * nox * 2;
        int j = rem >> 1, t = rem & 1;
        int x = min(xs0[j] + t, p.W0 - 1);
        const unsigned char *row = tile + ly * p.in_pitch + (x - x_first);
        float acc = p.kern[0] * (float)row[0];
        for (int q = 1; q < ks; q++) acc += p.kern[q] * (float)row[q];
        h1[(ly * p.txo + j) * 2 + t] = acc;
    }
    __syncthreads();

    // column pass (centre, then symmetric pairs) at the two y positions, then the bilinear mix
    for (int i = tid; i < noy * nox; i += 256) {
        int oy = i / nox, ox = i - oy * nox;
        float a1 = xa[ox], a0 = 1.f - a1, b1 = yb[oy], b0 = 1.f - b1;
        float hrow[2];
#pragma unroll
        for (int yt = 0; yt < 2; yt++) {
            int c = (yt ? ys1[oy] : ys0[oy]) - iy0;
            float v[2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                if (t == 1 && a1 == 0.f) { v[1] = 0.f; continue; }
                const float *col = h1 + (c * p.txo + ox) * 2 + t;
                float acc = p.kern[p.r] * col[0];
                for (int m = 1; m <= p.r; m++)
                    acc += p.kern[p.r + m] * (col[-m * p.txo * 2] + col[m * p.txo * 2]);
                v[t] = acc;
            }
            hrow[yt] = (a1 == 0.f) ? v[0] : v[0] * a0 + v[1] * a1;
        }
        out[(size_t)(oy0 + oy) * p.w + ox0 + ox] = hrow[0] * b0 + hrow[1] * b1;
    }
}

// ------------------------------------------------------------------------------------------------
// K2 fast paths for the pyramid the reference actually uses (pyr_scale 0.5: exact 1, 2, 4, 8x decimation).
// Same arithmetic, same order as k_level_image (bit-exact, contraction off); only the data movement differs:
// window bytes come from aligned dword loads and are unpacked with compile-time shifts (v_cvt_f32_ubyteN),
// nothing is staged through LDS as bytes.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float ubyte_of(uint32_t w, int n) { return (float)((w >> (8 * n)) & 255u); }

// level 0: 3x3 blur, no decimation.  Thread = 4 px x 8 rows; block tile 256 x 32.
__global__ __launch_bounds__(256) void k_level0(const uint8_t *__restrict__ src, float *__restrict__ dst,
                                                int W, int H, float k0, float k1, float k2)
{
#pragma clang fp contract(off)
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int x0 = blockIdx.x * 256 + 4 * tx, y0 = blockIdx.y * 32 + 8 * ty;
    if (x0 >= W || y0 >= H) return;
    const uint8_t *img = src + (size_t)blockIdx.z * W * H;
    float *out = dst + (size_t)blockIdx.z * W * H;
    const bool fast = x0 >= 4 && x0 + 8 <= W;         // W % 4 == 0 is guaranteed by the launcher
    float h[10][4];
#pragma unroll
    for (int j = 0; j < 10; j++) {
        const uint8_t *row = img + (size_t)reflect101(y0 - 1 + j, H) * W;
        float b[6];
        if (fast) {
            const uint32_t *rw = reinterpret_cast<const uint32_t *>(row + x0);
            const uint32_t wa = rw[-1], wb = rw[0], wc = rw[1];
            b[0] = ubyte_of(wa, 3);
            b[1] = ubyte_of(wb, 0); b[2] = ubyte_of(wb, 1); b[3] = ubyte_of(wb, 2); b[4] = ubyte_of(wb, 3);
            b[5] = ubyte_of(wc, 0);
        } else {
#pragma unroll
            for (int q = 0; q < 6; q++) b[q] = (float)row[reflect101(x0 - 1 + q, W)];
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {
            float acc = k0 * b[c];
            acc += k1 * b[c + 1];
            acc += k2 * b[c + 2];
            h[j][c] = acc;
        }
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        if (y0 + i < H) {
            float o[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                float d = k1 * h[i + 1][c];
                d += k2 * (h[i][c] + h[i + 2][c]);
                o[c] = d;
            }
            *reinterpret_cast<float4 *>(out + (size_t)(y0 + i) * W + x0) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

// exact S-fold decimation with a (2R+1)-tap Gaussian.  Row pass: one item = (input row, output column):
// both bilinear x taps (xa, xa+1) from ONE 2R+2 byte window held in registers.  Column pass: both y taps from one
// 2R+2 row window of the row-filtered samples in LDS.
template <int S, int R, int TXO, int TYO>
__global__ __launch_bounds__(256) void k_level_dec(const uint8_t *__restrict__ src, float *__restrict__ dst,
                                                   int W0, int H0, int w, int h, LevelArgs p)
{
#pragma clang fp contract(off)
    constexpr int ROWS = S * TYO + 2 * R;                 // input rows feeding the tile
    constexpr int NB = 2 * R + 2;                         // bytes per window
    constexpr int XOFF = (S / 2 - 1 - R);                 // window start relative to S*ox
    constexpr int AL = ((XOFF % 4) + 4) % 4;              // its offset inside the first aligned dword
    constexpr int NW = (AL + NB + 3) / 4;                 // aligned dwords covering the window
    __shared__ float hrow[ROWS][TXO][2];
    const int tid = threadIdx.x;
    const int ox0 = blockIdx.x * TXO, oy0 = blockIdx.y * TYO;
    const uint8_t *img = src + (size_t)blockIdx.z * W0 * H0;
    float *out = dst + (size_t)blockIdx.z * w * h;
    const int yfirst = S * oy0 + S / 2 - 1 - R;           // global row of tile row 0 (before reflection)

    // thread <-> (output column j, row group rg): the thread's RPG input rows are requested back to back before any of
    // them is used (the first version walked "items" one after the other: five dependent memory round trips per work-group
    // on the 1/4 level, which made this trivial kernel 8 % of the flow time)
    constexpr int NG = 256 / TXO, RPG = (ROWS + NG - 1) / NG;
    const int j = tid % TXO, rg = tid / TXO;
    const int ox = ox0 + j;
    if (ox < w) {
        const int xs = S * ox + XOFF;                     // first byte of the window
        const bool fast = xs - AL >= 0 && xs - AL + 4 * NW <= W0;
        auto refl = [](int p, int len) { return p < 0 ? -p : (p >= len ? 2 * (len - 1) - p : p); };   // |overshoot| < len (launcher)
        auto rowpass = [&](int i, const float (&b)[NB]) {
            const int ly = rg + NG * i;
            float a0 = p.kern[0] * b[0], a1 = p.kern[0] * b[1];
#pragma unroll
            for (int q = 1; q <= 2 * R; q++) {
                a0 += p.kern[q] * b[q];
                a1 += p.kern[q] * b[q + 1];
            }
            if (ly < ROWS) *reinterpret_cast<float2 *>(&hrow[ly][j][0]) = make_float2(a0, a1);
        };
        if (fast) {
            uint32_t wd[RPG][NW];
#pragma unroll
            for (int i = 0; i < RPG; i++) {
                const int ly = min(rg + NG * i, ROWS - 1);
                const uint32_t *rw = reinterpret_cast<const uint32_t *>(img + (size_t)refl(yfirst + ly, H0) * W0 + (xs - AL));
#pragma unroll
                for (int q = 0; q < NW; q++) wd[i][q] = rw[q];
            }
#pragma unroll
            for (int i = 0; i < RPG; i++) {
                float b[NB];
#pragma unroll
                for (int q = 0; q < NB; q++) b[q] = ubyte_of(wd[i][(AL + q) >> 2], (AL + q) & 3);
                rowpass(i, b);
            }
        } else {
#pragma unroll
            for (int i = 0; i < RPG; i++) {
                const int ly = min(rg + NG * i, ROWS - 1);
                const uint8_t *row = img + (size_t)refl(yfirst + ly, H0) * W0;
                float b[NB];
#pragma unroll
                for (int q = 0; q < NB; q++) b[q] = (float)row[reflect101(xs + q, W0)];
                rowpass(i, b);
            }
        }
    }
    __syncthreads();
    for (int it = tid; it < TYO * TXO; it += 256) {
        const int oy = it / TXO, j = it - oy * TXO;
        if (ox0 + j >= w || oy0 + oy >= h) continue;
        const int c0 = S * oy + R;                        // tile row of the first y tap's centre
        float2 win[2 * R + 2];
#pragma unroll
        for (int q = 0; q < 2 * R + 2; q++) win[q] = *reinterpret_cast<const float2 *>(&hrow[c0 - R + q][j][0]);
        float hy[2];
#pragma unroll
        for (int yt = 0; yt < 2; yt++) {
            float v0 = p.kern[R] * win[R + yt].x, v1 = p.kern[R] * win[R + yt].y;
#pragma unroll
            for (int m = 1; m <= R; m++) {
                v0 += p.kern[R + m] * (win[R + yt - m].x + win[R + yt + m].x);
                v1 += p.kern[R + m] * (win[R + yt - m].y + win[R + yt + m].y);
            }
            hy[yt] = v0 * 0.5f + v1 * 0.5f;
        }
        out[(size_t)(oy0 + oy) * w + ox0 + j] = hy[0] * 0.5f + hy[1] * 0.5f;
    }
}

int launch_level_image(const uint8_t *src, float *dst, int nimg, int W0, int H0,
                       const LevelGeom &g, hipStream_t s)
{
    LevelArgs a;
    memset(&a, 0, sizeof(a));
    a.W0 = W0; a.H0 = H0; a.w = g.w; a.h = g.h;
    a.r = g.ksize / 2;
    if (g.ksize > 31) { set_error("blur kernel size %d > 31 unsupported", g.ksize); return OFC_EUNSUPPORTED; }
    gaussian_kernel(g.ksize, g.sigma, a.kern);
    a.sx = (double)W0 / g.w; a.sy = (double)H0 / g.h;
    // fast paths: exact decimation by 1 / 2 / 4 / 8 with the reference's tap counts, rows dword-addressable
    const bool aligned = (W0 % 4) == 0 && ((uintptr_t)src % 4) == 0 && (((size_t)W0 * H0) % 4) == 0;
    if (aligned && g.w == W0 && g.h == H0 && a.r == 1) {
        hipLaunchKernelGGL(k_level0, dim3(cdiv(W0, 256), cdiv(H0, 32), nimg), dim3(256), 0, s, src, dst, W0, H0,
                           a.kern[0], a.kern[1], a.kern[2]);
        OFC_HIP(hipGetLastError());
        return OFC_OK;
    }
#define OFC_DEC(S_, R_, TX_, TY_)                                                                          \
    if (aligned && g.w * S_ == W0 && g.h * S_ == H0 && a.r == R_ && H0 > 2 * R_ + S_ && W0 > 2 * R_ + 8) {                                        \
        hipLaunchKernelGGL((k_level_dec<S_, R_, TX_, TY_>), dim3(cdiv(g.w, TX_), cdiv(g.h, TY_), nimg),    \
                           dim3(256), 0, s, src, dst, W0, H0, g.w, g.h, a);                                \
        OFC_HIP(hipGetLastError());                                                                        \
        return OFC_OK;                                                                                     \
    }
    OFC_DEC(2, 1, 64, 16)
    OFC_DEC(4, 4, 32, 8)
    OFC_DEC(8, 9, 32, 8)
#undef OFC_DEC
    // general path (any scale / tap count): LDS-staged tile
    if (a.sx <= 2.5) { a.txo = 64; a.tyo = 16; } else { a.txo = 32; a.tyo = 8; }
    a.in_w = (int)(a.txo * a.sx) + 2 * a.r + 4;
    a.in_h = (int)(a.tyo * a.sy) + 2 * a.r + 4;
    a.in_pitch = (a.in_w + 3) & ~3;
    size_t lds = (size_t)(2 * a.txo + 3 * a.tyo) * 4 + (size_t)a.in_h * 2 * a.txo * 4 +
                 (size_t)a.in_h * a.in_pitch;
    if (lds > 160 * 1024) { set_error("level_image tile needs %zu B LDS", lds); return OFC_EUNSUPPORTED; }
    dim3 grid(cdiv(g.w, a.txo), cdiv(g.h, a.tyo), nimg);
    hipLaunchKernelGGL(k_level_image, grid, dim3(256), lds, s, src, dst, a);
    OFC_HIP(hipGetLastError());
    return OFC_OK;
}

// ------------------------------------------------------------------------------------------------
// K3  polynomial expansion (the north-star kernel).  24 B/px algorithmic: 4 read + 20 written.
//
// Work-group = 256 threads = 4 waves, output strip 240 columns x rows_per_block rows, walked in chunks
// of 8 rows:
//   vertical pass   thread <-> column (250 = 240 + 2x5 halo columns, replicate-clamped).  Each thread keeps an
//                   18-row register window of its column that slides down 8 rows per chunk; the 8 new rows
//                   are requested right after the barrier (lanes <-> x: coalesced), so their latency hides
//                   under the horizontal pass.  The window yields the three f32 moments t0,t1,t2 of 8 rows with
//                   the symmetric / antisymmetric tap pairing of the reference -> LDS [3][8][256].
//   horizontal pass wave <-> row, lane <-> 4 consecutive x: 12 ds_read_b128 (forced whole, see lds_read4),
//                   accumulators as described in the loop, 5 float4 stores per lane (1 KiB/wave-instruction).
// 24 KB LDS, 146 VGPRs -> 3 waves/SIMD.  Measured ceilings on MI355X for this traffic shape (tools/membench):
// 1 read : 5 write streams = 4.9 TB/s; this kernel's compute alone (stores off) = 5.9 TB/s-equivalent.
// ------------------------------------------------------------------------------------------------
// 16-byte LDS read that hipcc may not narrow: a plain float4 load whose tail elements are unused is split into
// ds_read2_b32 / ds_read2_b64, and with a 16-byte lane stride those are 8-way bank conflicts
// (SQ_LDS_BANK_CONFLICT 1.3e8 cycles per launch before this).  volatile keeps the access whole -> ds_read_b128.
__device__ __forceinline__ float4 lds_read4(const float *p)
{
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef const volatile v4f __attribute__((address_space(3))) * lds_v4f_ptr;
    const v4f v = *(lds_v4f_ptr)(p);          // explicit LDS address space: ds_read_b128, not flat_load
    return make_float4(v.x, v.y, v.z, v.w);
}

constexpr int PE_N = 5;
constexpr int PE_TX = 240;
constexpr int PE_VW = PE_TX + 2 * PE_N;   // 250
constexpr int PE_CH = 8;
constexpr int PE_PLANE = 66;    // float4 per (row, pixel & 3) plane of the moments tile: 64 + 2 -> bank-staggered

struct PolyArgs {
    int W, H, rows_per_block;
    float g[PE_N + 1], xg[PE_N + 1], xxg[PE_N + 1];
    double ig11, ig03, ig33, ig55;
};

// TAG only gives the bench hook's launches their own symbol in rocprof's kernel statistics.
// U8IN: the pyramid's level 0 (3x3 [1/4 1/2 1/4] blur of the u8 frame, REFLECT_101, no decimation) is evaluated on the
// fly from the frame instead of being read back as an f32 image: 1 B/px read instead of 4, and the level-0 image
// kernel with its 5 B/px disappears.  Every intermediate value of that blur is a multiple of 1/16 below 256, exactly
// representable in f32 whatever the evaluation order, so the result is bit-identical to k_level0 + this kernel.
// F64H: the six horizontal sums exactly as the reference forms them (double accumulators; b1/b4 from double products, the
// other four from float products) -- a study / fallback build (OFC_POLYEXP_F64=1): see DESIGN.md section 2.
template <int TAG, bool U8IN, bool F64H = false>
__global__ __launch_bounds__(256, 3) void k_polyexp(const void *__restrict__ Iv, float *__restrict__ R,
                                                 PolyArgs p)
{
    // vertical moments of the chunk, one float4 (t0, t1, t2, t1) per pixel: the horizontal pass consumes them as the
    // register pairs (t0,t1) and (t2,t1), which is what lets it run on packed-f32 instructions.  Pixel p of the strip
    // lives at [p & 3][p >> 2]: lane l of the horizontal pass reads pixels 4l+j, so for a fixed j the 64 lanes touch 64
    // consecutive float4 of one plane (conflict-free ds_read_b128); planes are 66 float4 apart (264 dwords = 8 mod 32
    // banks) so that the vertical pass's writes, lane <-> pixel, spread over the banks as well.
    __shared__ __align__(16) float4 t4[PE_CH][4][PE_PLANE];
    __shared__ __align__(16) float stage[4][PE_TX * 5 + 16];   // per-wave output row (240 px x 5 coefficients)
    typedef float v2f __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int W = p.W, H = p.H;
    const int x0 = blockIdx.x * PE_TX;
    const int y_begin = blockIdx.y * p.rows_per_block;
    const int y_end = min(y_begin + p.rows_per_block, H);
    const size_t plane = (size_t)W * H;
    const float *img = U8IN ? nullptr : reinterpret_cast<const float *>(Iv) + (size_t)blockIdx.z * plane;
    const uint8_t *img8 = U8IN ? reinterpret_cast<const uint8_t *>(Iv) + (size_t)blockIdx.z * plane : nullptr;
    float *out = R + (size_t)blockIdx.z * 5 * plane;
    const int xc = min(max(x0 - PE_N + tid, 0), W - 1);   // this thread's (clamped) column
    const bool vec_ok = (W & 3) == 0;

    // U8IN: one (unaligned) dword per frame row holds the three horizontal taps of this column: bytes
    // reflect101(xc-1), xc, reflect101(xc+1) all lie in [base, base+3] for base = clamp(xc-1, 0, W-4)
    typedef uint32_t u32u __attribute__((aligned(1)));
    const int base = U8IN ? min(max(xc - 1, 0), W - 4) : 0;
    const int sh0 = U8IN ? 8 * (reflect101(xc - 1, W) - base) : 0, sh1 = U8IN ? 8 * (xc - base) : 0,
              sh2 = U8IN ? 8 * (reflect101(xc + 1, W) - base) : 0;
    auto hword = [&](int r) -> uint32_t { return *reinterpret_cast<const u32u *>(img8 + (size_t)r * W + base); };
    auto hval = [&](uint32_t w) -> float {          // row-blurred sample (exact: a multiple of 1/4)
        return 0.25f * (float)((w >> sh0) & 255u) + 0.5f * (float)((w >> sh1) & 255u) + 0.25f * (float)((w >> sh2) & 255u);
    };
    auto i0_any = [&](int r) -> float {             // level-0 image at (valid) row r, any position: 3 row loads
        return 0.25f * hval(hword(reflect101(r - 1, H))) + 0.5f * hval(hword(r)) + 0.25f * hval(hword(reflect101(r + 1, H)));
    };
    // strips whose rows (incl. the +-1 blur rows) need neither clamping nor reflection share the row-blurred samples
    // between consecutive level-0 rows: 20 + 8 per chunk dword loads per thread, the count of the f32 path
    const bool interior = U8IN && y_begin >= PE_N + 1 && y_begin + p.rows_per_block + 2 * PE_N + 2 <= H - 1;

    // register window of this thread's column: rows yc-5 .. yc+12; slides down 8 rows per chunk, the 8 new
    // rows are requested right after the barrier so that their HBM latency hides under the horizontal pass
    float s[8 + 2 * PE_N], nxt[8];
    uint32_t nxtw[8];
    float hk0 = 0.f, hk1 = 0.f;                         // U8IN interior: row-blurred samples of window rows 18, 19
    if (!U8IN) {
#pragma unroll
        for (int j = 0; j < 8 + 2 * PE_N; j++)
            s[j] = img[(size_t)min(max(y_begin - PE_N + j, 0), H - 1) * W + xc];
    } else if (interior) {
        float hv[8 + 2 * PE_N + 2];
#pragma unroll
        for (int j = 0; j < 8 + 2 * PE_N + 2; j++) hv[j] = hval(hword(y_begin - PE_N - 1 + j));
#pragma unroll
        for (int j = 0; j < 8 + 2 * PE_N; j++) s[j] = 0.25f * hv[j] + 0.5f * hv[j + 1] + 0.25f * hv[j + 2];
        hk0 = hv[8 + 2 * PE_N];
        hk1 = hv[8 + 2 * PE_N + 1];
    } else {
#pragma unroll
        for (int j = 0; j < 8 + 2 * PE_N; j++) s[j] = i0_any(min(max(y_begin - PE_N + j, 0), H - 1));
    }

    for (int yc = y_begin; yc < y_end; yc += PE_CH) {
        // ---- vertical pass ----
        // per tap pair: (a+b, b-a) in one packed add, (t0,t2) += (g,xxg)*(a+b) in one packed fma, t1 in a scalar fma:
        // 3 VALU instructions instead of 5, every lane-operation identical to the scalar form (same roundings)
        if (tid < PE_VW) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                v2f t02 = {s[i + PE_N] * p.g[0], 0.f};
                float t1 = 0.f;
#pragma unroll
                for (int k = 1; k <= PE_N; k++) {
                    const float a = s[i + PE_N - k], b = s[i + PE_N + k];
                    const v2f pd = (v2f){b, b} + (v2f){a, -a};                     // (a + b, b - a)
                    t02 = __builtin_elementwise_fma((v2f){p.g[k], p.xxg[k]}, (v2f){pd.x, pd.x}, t02);
                    t1 = fmaf(p.xg[k], pd.y, t1);
                }
                t4[i][tid & 3][tid >> 2] = make_float4(t02.x, t1, t02.y, t1);
            }
        }
        __syncthreads();
        if (yc + PE_CH < y_end) {
            if (!U8IN) {
#pragma unroll
                for (int j = 0; j < 8; j++)
                    nxt[j] = img[(size_t)min(yc + PE_CH + PE_N + j, H - 1) * W + xc];
            } else if (interior) {
#pragma unroll
                for (int j = 0; j < 8; j++) nxtw[j] = hword(yc + PE_CH + PE_N + 1 + j);   // frame rows yc+14 .. yc+21
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++) nxt[j] = i0_any(min(yc + PE_CH + PE_N + j, H - 1));
            }
        }
        // ---- horizontal pass ----
        for (int rr = wave; rr < PE_CH; rr += 4) {
            const int y = yc + rr;
            if (y >= y_end) break;
            const int xo = x0 + 4 * lane;
            if (lane < PE_TX / 4 && xo < W) {
                v2f A[14], Q[14];          // (t0, t1) and (t2, t1) of strip pixels 4*lane .. 4*lane + 13
#pragma unroll
                for (int j = 0; j < 14; j++) {
                    const float4 v = lds_read4(reinterpret_cast<const float *>(&t4[rr][j & 3][lane + (j >> 2)]));
                    A[j] = (v2f){v.x, v.y};
                    Q[j] = (v2f){v.z, v.w};
                }
                float r0[4], r1[4], r2[4], r3[4], r4[4];
#pragma unroll
                for (int o = 0; o < 4; o++) {
                    const int c = o + PE_N;
                    // The reference accumulates these six 11-tap sums in double.  Its inputs (the vertical moments) are
                    // f32 already, each sum has only 6 terms, and the one place where rounding is amplified -- b1*ig03
                    // cancelling against b4|b5*ig33 in the second-derivative coefficients -- is evaluated in f64 below.
                    // f32 FMA accumulation costs <= ~3 ulp per sum; measured effect on the flow vs the oracle:
                    // 4.7e-7 relative / 1.1e-4 px worst case incl. flat-bright and half-black frames (bar 1e-4 / 1e-3),
                    // the same as with f64 accumulators, for 40 % less time in this VALU-bound pass.
                    // packed over PAIRS OF SUMS (not pairs of pixels, whose operands would straddle register pairs):
                    // per tap pair 3 packed adds + 3 packed fmas instead of 5 + 6 scalar ones
                    float b1, b2, b3, b4, b5, b6;
                    double d1 = 0, d4 = 0, d5 = 0;
                    if (F64H) {
#pragma clang fp contract(off)
                        double e1 = (double)(A[c].x * p.g[0]), e2 = 0, e3 = (double)(A[c].y * p.g[0]), e4 = 0,
                               e5 = (double)(Q[c].x * p.g[0]), e6 = 0;
#pragma unroll
                        for (int k = 1; k <= PE_N; k++) {
                            const double tg = (double)(A[c + k].x + A[c - k].x);
                            e1 += tg * (double)p.g[k];
                            e4 += tg * (double)p.xxg[k];
                            e2 += (double)((A[c + k].x - A[c - k].x) * p.xg[k]);
                            e3 += (double)((A[c + k].y + A[c - k].y) * p.g[k]);
                            e6 += (double)((A[c + k].y - A[c - k].y) * p.xg[k]);
                            e5 += (double)((Q[c + k].x + Q[c - k].x) * p.g[k]);
                        }
                        d1 = e1; d4 = e4; d5 = e5;
                        b1 = (float)e1; b4 = (float)e4; b5 = (float)e5;
                        r1[o] = (float)(e2 * p.ig11);
                        r0[o] = (float)(e3 * p.ig11);
                        r4[o] = (float)(e6 * p.ig55);
                        b2 = b3 = b6 = 0.f;
                    } else {
                    v2f b14 = {A[c].x * p.g[0], 0.f};                    // (b1, b4)
                    v2f b26 = {0.f, 0.f};                                // (b2, b6)
                    v2f b53 = Q[c] * (v2f){p.g[0], p.g[0]};              // (b5, b3)
#pragma unroll
                    for (int k = 1; k <= PE_N; k++) {
                        const v2f sm = A[c + k] + A[c - k];              // (t0 sum, -)
                        const v2f df = A[c + k] - A[c - k];              // (t0 diff, t1 diff)
                        const v2f sq = Q[c + k] + Q[c - k];              // (t2 sum, t1 sum)
                        b14 = __builtin_elementwise_fma((v2f){sm.x, sm.x}, (v2f){p.g[k], p.xxg[k]}, b14);
                        b26 = __builtin_elementwise_fma(df, (v2f){p.xg[k], p.xg[k]}, b26);
                        b53 = __builtin_elementwise_fma(sq, (v2f){p.g[k], p.g[k]}, b53);
                    }
                    b1 = b14.x; b4 = b14.y; b2 = b26.x; b6 = b26.y; b5 = b53.x; b3 = b53.y;
                    }
                    if (F64H) {
#pragma clang fp contract(off)
                        r3[o] = (float)(d1 * p.ig03 + d4 * p.ig33);
                        r2[o] = (float)(d1 * p.ig03 + d5 * p.ig33);
                    } else {
                        r1[o] = b2 * (float)p.ig11;
                        r0[o] = b3 * (float)p.ig11;
                        r3[o] = (float)((double)b1 * p.ig03 + (double)b4 * p.ig33);
                        r2[o] = (float)((double)b1 * p.ig03 + (double)b5 * p.ig33);
                        r4[o] = b6 * (float)p.ig55;
                    }
                }
                // R is pixel-interleaved ([y][x][5], as OpenCV keeps it) so that the consumer's bilinear taps are two
                // 40-byte runs instead of 20 scattered dwords.  A lane's 4 px x 5 coefficients are 80 contiguous bytes;
                // storing them directly would make every wave-instruction write 16 B out of each 80 (measured: 5x
                // slower).  They cross a per-wave LDS row instead (conflict-free at the 80-B lane stride) and leave as
                // five fully contiguous 1-KiB wave stores.
                if (vec_ok) {
                    float *sg = &stage[wave][20 * lane];
                    *reinterpret_cast<float4 *>(sg) = make_float4(r0[0], r1[0], r2[0], r3[0]);
                    *reinterpret_cast<float4 *>(sg + 4) = make_float4(r4[0], r0[1], r1[1], r2[1]);
                    *reinterpret_cast<float4 *>(sg + 8) = make_float4(r3[1], r4[1], r0[2], r1[2]);
                    *reinterpret_cast<float4 *>(sg + 12) = make_float4(r2[2], r3[2], r4[2], r0[3]);
                    *reinterpret_cast<float4 *>(sg + 16) = make_float4(r1[3], r2[3], r3[3], r4[3]);
                } else {
                    float *o0 = out + ((size_t)y * W + xo) * 5;
#pragma unroll
                    for (int o = 0; o < 4; o++)
                        if (xo + o < W) {
                            o0[o * 5] = r0[o]; o0[o * 5 + 1] = r1[o]; o0[o * 5 + 2] = r2[o];
                            o0[o * 5 + 3] = r3[o]; o0[o * 5 + 4] = r4[o];
                        }
                }
            }
            if (vec_ok) {     // wave-uniform; the staging row belongs to this wave only
                __builtin_amdgcn_wave_barrier();
                typedef float v4f __attribute__((ext_vector_type(4)));
                const int nf4 = min(PE_TX, W - x0) * 5 / 4;
                v4f *orow = reinterpret_cast<v4f *>(out + ((size_t)y * W + x0) * 5);
#pragma unroll
                for (int j = 0; j < 5; j++) {
                    const int idx = lane + 64 * j;
                    if (idx < nf4) {
                        const float4 v = lds_read4(&stage[wave][4 * idx]);
                        __builtin_nontemporal_store((v4f){v.x, v.y, v.z, v.w}, orow + idx);
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2 * PE_N; j++) s[j] = s[j + 8];
        if (U8IN && interior) {
            float hn[10];
            hn[0] = hk0;
            hn[1] = hk1;
#pragma unroll
            for (int j = 0; j < 8; j++) hn[2 + j] = hval(nxtw[j]);
#pragma unroll
            for (int j = 0; j < 8; j++) s[2 * PE_N + j] = 0.25f * hn[j] + 0.5f * hn[j + 1] + 0.25f * hn[j + 2];
            hk0 = hn[8];
            hk1 = hn[9];
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) s[2 * PE_N + j] = nxt[j];
        }
    }
}


// OFC_POLYEXP_F64=1: horizontal sums in double, as the reference (read at every launch: a test toggles it)
static bool polyexp_f64()
{
    const char *e = getenv("OFC_POLYEXP_F64");
    return e && e[0] == '1';
}

int polyexp_default_rows(int W, int H, int nimg)
{
    // 16-row strips: the 10-row window warm-up re-reads input rows that the strip above is reading at about the same
    // time (L2 hits), and the shorter strips keep more independent work-groups in flight -- measured 4.61 TB/s against
    // 4.38 (64 rows) .. 4.53 (32-48 rows) on 64 1080p images.  Shrink further only when the launch would not give
    // every CU a few work-groups.
    int tiles_x = cdiv(W, PE_TX);
    int rows = 16;
    while (rows > PE_CH && (int64_t)tiles_x * cdiv(H, rows) * nimg < 1024) rows >>= 1;
    return rows;
}

static void polyexp_args(PolyArgs &a, int W, int H, int nimg, const PolyConsts &c, int rows_per_block)
{
    a.W = W; a.H = H;
    for (int i = 0; i <= PE_N; i++) { a.g[i] = c.g[i]; a.xg[i] = c.xg[i]; a.xxg[i] = c.xxg[i]; }
    a.ig11 = c.ig11; a.ig03 = c.ig03; a.ig33 = c.ig33; a.ig55 = c.ig55;
    if (rows_per_block <= 0) rows_per_block = polyexp_default_rows(W, H, nimg);
    a.rows_per_block = cdiv(rows_per_block, PE_CH) * PE_CH;
}

// polyexp of pyramid level 0 straight from the u8 frames (see U8IN); the caller checks polyexp_u8_ok()
bool polyexp_u8_ok(int W, int H, const LevelGeom &g)
{
    if (g.w != W || g.h != H || g.ksize != 3 || W < 4 || H < 2) return false;
    float k[3];
    gaussian_kernel(3, g.sigma, k);
    return k[0] == 0.25f && k[1] == 0.5f && k[2] == 0.25f;
}

int launch_polyexp_u8(const uint8_t *frames, float *R, int nimg, int W, int H, const PolyConsts &c, hipStream_t s)
{
    PolyArgs a;
    polyexp_args(a, W, H, nimg, c, 0);
    dim3 grid(cdiv(W, PE_TX), cdiv(H, a.rows_per_block), nimg);
    if (polyexp_f64()) hipLaunchKernelGGL((k_polyexp<0, true, true>), grid, dim3(256), 0, s, frames, R, a);
    else hipLaunchKernelGGL((k_polyexp<0, true>), grid, dim3(256), 0, s, frames, R, a);
    OFC_HIP(hipGetLastError());
    return OFC_OK;
}

int launch_polyexp(const float *I, float *R, int nimg, int W, int H, const PolyConsts &c,
                   int rows_per_block, hipStream_t s, bool bench_tag)
{
    PolyArgs a;
    polyexp_args(a, W, H, nimg, c, rows_per_block);
    dim3 grid(cdiv(W, PE_TX), cdiv(H, a.rows_per_block), nimg);
    if (polyexp_f64()) hipLaunchKernelGGL((k_polyexp<2, false, true>), grid, dim3(256), 0, s, I, R, a);
    else if (bench_tag) hipLaunchKernelGGL((k_polyexp<1, false>), grid, dim3(256), 0, s, I, R, a);
    else hipLaunchKernelGGL((k_polyexp<0, false>), grid, dim3(256), 0, s, I, R, a);
    OFC_HIP(hipGetLastError());
    return OFC_OK;
}

// (the per-pixel update-matrices arithmetic -- um_load / um_math -- lives in flow_device.h)

__global__ __launch_bounds__(256) void k_update_matrices(const float *__restrict__ R0b,
                                                         const float *__restrict__ R1b,
                                                         size_t pair_stride_R,
                                                         const float *__restrict__ flowb,
                                                         float *__restrict__ Mb, int W, int H)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const size_t plane = (size_t)W * H;
    const float *R0 = R0b + blockIdx.z * pair_stride_R;
    const float *R1 = R1b + blockIdx.z * pair_stride_R;
    const float2 fl = reinterpret_cast<const float2 *>(flowb)[(size_t)blockIdx.z * plane + (size_t)y * W + x];
    float *M = Mb + (size_t)blockIdx.z * 5 * plane;
    float m[5];
    update_matrices_px(R0, R1, plane, W, H, x, y, fl, m);
    const size_t idx = (size_t)y * W + x;
#pragma unroll
    for (int c = 0; c < 5; c++) M[c * plane + idx] = m[c];
}

int launch_update_matrices(const float *R0, const float *R1, size_t pair_stride_R,
                           const float *flow, float *M, int npair, int W, int H, hipStream_t s)
{
    if ((int64_t)W * H * 5 >= (1ll << 30)) { set_error("frame too large for 32-bit R offsets (%dx%d)", W, H); return OFC_EUNSUPPORTED; }
    dim3 grid(cdiv(W, 256), H, npair);
    hipLaunchKernelGGL(k_update_matrices, grid, dim3(256), 0, s, R0, R1, pair_stride_R, flow, M, W, H);
    OFC_HIP(hipGetLastError());
    return OFC_OK;
}

// ------------------------------------------------------------------------------------------------
// K5  box mean (winsize x winsize, replicate border) of the 5 M planes + regularised 2x2 solve.
// 28 B/px algorithmic (20 read + 8 written).
//
// "Column march": work-group = 256 threads <-> 256 columns (256 - 2m outputs + halo), marching down a
// strip of rows.  Every thread keeps the vertical (2m+1)-row sums of its column for the 5 channels as
// f64 running sums in registers (+ new row - old row: 2 loads per channel per row, L2-resident), and
// every 4 rows the work-group exchanges them through LDS (41 KB -> 3 work-groups per CU) so
// that wave <-> row, lane <-> 4 consecutive x can form the horizontal sums by sliding (2m+1 + 3*2 adds
// for 4 outputs) from 9 ds_read_b128 per channel, solve, and store 4 float2.  The sums are exact f64
// sums of f32 values (the reference's running sums round the f32 differences: ~1e-7 relative).
// ------------------------------------------------------------------------------------------------

template <int M>
__global__ __launch_bounds__(256) void k_box_solve(const float *__restrict__ Mb,
                                                   float *__restrict__ flowb, int W, int H,
                                                   int rows_per_block)
{
    constexpr int TXO = 256 - 2 * M;          // outputs per work-group row
    constexpr int NV = 2 * M + 4;             // vsum values a lane needs for its 4 outputs
    constexpr int NV2 = (NV + 1) / 2;         // as double2 reads
    constexpr int PITCH = 256 + 2;            // doubles; keeps rows 16-B aligned
    __shared__ __align__(16) double vs[5][BS_ROWS][PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int x0 = blockIdx.x * TXO;
    const int y_begin = blockIdx.y * rows_per_block;
    const int y_end = min(y_begin + rows_per_block, H);
    const size_t plane = (size_t)W * H;
    const float *Mp = Mb + (size_t)blockIdx.z * 5 * plane;
    float2 *flow = reinterpret_cast<float2 *>(flowb) + (size_t)blockIdx.z * plane;
    const int xc = min(max(x0 - M + tid, 0), W - 1);
    const double scale = 1.0 / ((2 * M + 1) * (2 * M + 1));

    // vertical sums of rows [y_begin-M, y_begin+M] (replicate)
    double v[5] = {0, 0, 0, 0, 0};
    for (int j = -M; j <= M; j++) {
        const size_t o = (size_t)min(max(y_begin + j, 0), H - 1) * W + xc;
#pragma unroll
        for (int c = 0; c < 5; c++) v[c] += (double)Mp[c * plane + o];
    }
    for (int yc = y_begin; yc < y_end; yc += BS_ROWS) {
#pragma unroll
        for (int r = 0; r < BS_ROWS; r++) {
#pragma unroll
            for (int c = 0; c < 5; c++) vs[c][r][tid] = v[c];
            // advance to row yc + r + 1
            const int y = yc + r;
            const size_t oa = (size_t)min(y + 1 + M, H - 1) * W + xc;
            const size_t os = (size_t)max(y - M, 0) * W + xc;
#pragma unroll
            for (int c = 0; c < 5; c++)
                v[c] += (double)Mp[c * plane + oa] - (double)Mp[c * plane + os];
        }
        __syncthreads();
        const int y = yc + wave;
        const int xo = x0 + 4 * lane;
        if (y < y_end && 4 * lane < TXO && xo < W) {
            double S[5][4];
#pragma unroll
            for (int c = 0; c < 5; c++) {
                double a[2 * NV2];
#pragma unroll
                for (int q = 0; q < NV2; q++) {
                    double2 d = *reinterpret_cast<const double2 *>(&vs[c][wave][4 * lane + 2 * q]);
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
            }
#pragma unroll
            for (int o = 0; o < 4; o++) {
                if (4 * lane + o < TXO && xo + o < W) {
                    const double g11 = S[0][o] * scale, g12 = S[1][o] * scale, g22 = S[2][o] * scale,
                                 h1 = S[3][o] * scale, h2 = S[4][o] * scale;
                    const double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
                    flow[(size_t)y * W + xo + o] = make_float2((float)((g11 * h2 - g12 * h1) * idet),
                                                               (float)((g22 * h1 - g12 * h2) * idet));
                }
            }
        }
        __syncthreads();   // vs is rewritten by the next step
    }
}

int box_default_rows(int W, int H, int npair)
{
    int tiles_x = cdiv(W, 256 - 14);
    int rows = 128;
    while (rows > 16 && (int64_t)tiles_x * cdiv(H, rows) * npair < 1536) rows >>= 1;
    return rows;
}

int launch_box_solve(const float *M, float *flow, int npair, int W, int H, int winsize,
                     int rows_per_block, hipStream_t s)
{
    if (rows_per_block <= 0) rows_per_block = box_default_rows(W, H, npair);
    rows_per_block = cdiv(rows_per_block, BS_ROWS) * BS_ROWS;
    dim3 block(256);
#define OFC_BOX_CASE(MM)                                                                         \
    case 2 * MM + 1: {                                                                           \
        dim3 grid(cdiv(W, 256 - 2 * MM), cdiv(H, rows_per_block), npair);                        \
        hipLaunchKernelGGL(k_box_solve<MM>, grid, block, 0, s, M, flow, W, H, rows_per_block);   \
        break;                                                                                   \
    }
    switch (winsize) {
        OFC_BOX_CASE(2)
        OFC_BOX_CASE(3)
        OFC_BOX_CASE(4)
        OFC_BOX_CASE(5)
        OFC_BOX_CASE(6)
        OFC_BOX_CASE(7)
        OFC_BOX_CASE(8)
    default:
        set_error("winsize %d unsupported (odd 5..17)", winsize);
        return OFC_EUNSUPPORTED;
    }
#undef OFC_BOX_CASE
    OFC_HIP(hipGetLastError());
    return OFC_OK;
}

// one vector of resize(prevFlow, (w,h), INTER_LINEAR) * mul at (x, y): the arithmetic of k_flow_resize (bit-exact)
struct UpsArgs {
    const float *src;     // coarse flow [pair][sh][sw][2]; nullptr = flow_in is already at this level's size
    int sw, sh;
    double scx, scy;
    float mul;
};

__device__ __forceinline__ float2 upsampled_flow(const float2 *__restrict__ s, int sw, int sh, double scx, double scy,
                                                 float mul, int x, int y)
{
#pragma clang fp contract(off)
    int sx0, sy0, sy1;
    float a1, b1;
    if (scx == 0.5 && scy == 0.5) {
        // exact x2 (the reference's pyr_scale 0.5 on even sizes): (d+0.5)*0.5-0.5 = 0.5d-0.25 is exact in f32 too,
        // so the taps come from integer arithmetic instead of two f64 evaluations per pixel -- same values
        const int sx = (x + 1) / 2 - 1, sy = (y + 1) / 2 - 1;          // floor(0.5d - 0.25)
        a1 = (x & 1) ? 0.25f : 0.75f;
        b1 = (y & 1) ? 0.25f : 0.75f;
        sx0 = sx;
        if (sx < 0) { a1 = 0.f; sx0 = 0; }
        if (sx >= sw - 1) { a1 = 0.f; sx0 = sw - 1; }
        sy0 = min(max(sy, 0), sh - 1);
        sy1 = min(max(sy + 1, 0), sh - 1);
    } else {
        lin_tap_x(x, scx, sw, sx0, a1);
        lin_tap_y(y, scy, sh, sy0, sy1, b1);
    }
    const float a0 = 1.f - a1, b0 = 1.f - b1;
    const int sx1 = (a1 == 0.f) ? sx0 : sx0 + 1;
    const float2 p00 = s[(size_t)sy0 * sw + sx0], p01 = s[(size_t)sy0 * sw + sx1];
    const float2 p10 = s[(size_t)sy1 * sw + sx0], p11 = s[(size_t)sy1 * sw + sx1];
    float h0x, h0y, h1x, h1y;
    if (a1 == 0.f) {
        h0x = p00.x; h0y = p00.y; h1x = p10.x; h1y = p10.y;
    } else {
        h0x = p00.x * a0 + p01.x * a1; h0y = p00.y * a0 + p01.y * a1;
        h1x = p10.x * a0 + p11.x * a1; h1y = p10.y * a0 + p11.y * a1;
    }
    return make_float2((h0x * b0 + h1x * b1) * mul, (h0y * b0 + h1y * b1) * mul);
}

// ------------------------------------------------------------------------------------------------
// K4+K5 fused: one Farneback iteration  flow_in -> flow_out  without M ever existing in memory.
// Same column march as k_box_solve, but the incoming row of M is COMPUTED (um_load/um_math: R0, flow_in,
// bilinear gather of R1) instead of loaded, and the outgoing row comes back from a 16-row ring that lives
// in REGISTERS (80 VGPRs per lane): the march is unrolled over super-steps of 16 rows so that every ring
// slot has a compile-time index.  (A first version kept the ring in a work-group-private global buffer:
// rocprof showed 2.2 GB of WRITE_SIZE and a 50 % L2 miss rate per level-0 launch -- the ring traffic alone
// was as large as the algorithmic traffic.)
// HBM traffic per pixel: 20 (R0) + 20 (R1, gathered) + 8 (flow in) + 8 (flow out) = 56 B, versus
// 68 + 28 (+20 for the second read of M) for the separate kernels.
// ------------------------------------------------------------------------------------------------
// UPS: the first iteration of a level reads its initial flow straight from the coarser level (bilinear x2, times
// 1/pyr_scale) instead of from a materialised upsampled copy -- K6 fused in, 16 B/px less traffic and one launch less.
// UPS = 0: flow_in is at this level's size; 1: bilinear upsample of the coarser level with general taps; 2: the exact x2
// pyramid (pyr_scale 0.5 on even sizes), whose taps follow a fixed parity pattern (see below)
// STAMP: diagnostic build (tools/fi_stamps.py): s_memtime stamps around the phases of a step, summed per wave into `dbg`
// [block][wave][8] -- where a step's cycles go; never used by the engine
// SUMS: also emit, per work-group, the f64 sums of the flow vectors it stores (`sums`: [grid][2]) -- the column sums Lloyd's
// centring needs, taken from the last level-0 iteration's epilogue instead of from an extra 5 GB sweep over the clip
template <int M, int UPS, bool STAMP = false, bool SUMS = false>
__global__ __launch_bounds__(256, 2) void k_flow_iter(const float *__restrict__ Rb, size_t frame_stride_R,
                                                      const float *__restrict__ flow_inb,
                                                      float *__restrict__ flow_outb, int W, int H,
                                                      int rows_per_block /* multiple of 16 */, UpsArgs ups,
                                                      int tiles_x, int n_strips, int npair,
                                                      unsigned long long *__restrict__ dbg = nullptr,
                                                      double *__restrict__ sums = nullptr)
{
    constexpr int TXO = 256 - 2 * M;
    constexpr int NV = 2 * M + 4;
    constexpr int NV2 = (NV + 1) / 2;
    // Vertical sums of one row cross LDS as PAIRS of doubles, even pairs in one plane and odd pairs in another: the
    // horizontal pass's lane l reads pairs 2l .. 2l+8, so for a fixed q the 64 lanes read 64 consecutive 16-byte pairs
    // of one plane (conflict-free ds_read_b128; with the plain [column] layout the 32-byte lane stride was a 4-way bank
    // conflict: SQ_LDS_BANK_CONFLICT 1.3e8 of 2.2e8 LDS-active cycles per level-0 launch).  Plane pitch 136 doubles
    // = 64 B mod 128: the writers' two interleaved 16-byte streams then cover every bank exactly twice.
    constexpr int PLD = 136, PITCH = 2 * PLD;
    __shared__ __align__(16) double vs[5][BS_ROWS][PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware work-group -> (tile, pair) map.  Work-groups are dealt round-robin over the 8 XCDs, so ids L and
    // L+8 share an L2.  Consecutive same-XCD ids take the SAME tile of CONSECUTIVE pairs: frame t's R is R1 of pair
    // t-1 and R0 of pair t, and the two work-groups march down their strips in step, so the second read hits L2
    // instead of HBM.  (speed only: any placement gives the same result)
    const int tiles = tiles_x * n_strips;
    const int group = blockIdx.x / (8 * npair), rem = blockIdx.x - group * (8 * npair);
    const int pair = rem >> 3, tile = group * 8 + (rem & 7);
    if (tile >= tiles) {
        if (SUMS && threadIdx.x < 2) sums[(size_t)blockIdx.x * 2 + threadIdx.x] = 0.0;     // padding work-group of the grid
        return;
    }
    double su = 0, sv = 0;                          // SUMS: this thread's share of sum(u), sum(v)
    const int tile_x = tile % tiles_x, tile_y = tile / tiles_x;
    const int x0 = tile_x * TXO;
    const int y_begin = tile_y * rows_per_block;
    const int y_end = min(y_begin + rows_per_block, H);
    const size_t plane = (size_t)W * H;
    const float *R0 = Rb + (size_t)pair * frame_stride_R;
    const float *R1 = R0 + frame_stride_R;
    const float2 *flow_in = UPS ? reinterpret_cast<const float2 *>(ups.src) + (size_t)pair * ups.sw * ups.sh
                                : reinterpret_cast<const float2 *>(flow_inb) + (size_t)pair * plane;
    float2 *flow_out = reinterpret_cast<float2 *>(flow_outb) + (size_t)pair * plane;
    const int xc = min(max(x0 - M + tid, 0), W - 1);
    const int vs_w = ((tid >> 1) & 1) * PLD + 2 * (tid >> 2) + (tid & 1);     // this column's slot in a row of vs
    const double scale = 1.0 / ((2 * M + 1) * (2 * M + 1));
    auto flow_at = [&](int row) -> float2 {
        if (UPS) return upsampled_flow(flow_in, ups.sw, ups.sh, ups.scx, ups.scy, ups.mul, xc, row);
        return flow_in[(size_t)row * W + xc];
    };
    // UPS with the exact x2 pyramid: this thread's horizontal taps never change, and the 4 fine rows of a step touch only
    // 3 or 4 distinct coarse rows -- interpolate each coarse row once per step (the arithmetic of upsampled_flow, shared)
    constexpr bool ups2 = UPS == 2;
    int usx0 = 0, usx1 = 0;
    float ua1 = 0.f;
    if (ups2) {
        const int sx = (xc + 1) / 2 - 1;
        ua1 = (xc & 1) ? 0.25f : 0.75f;
        usx0 = sx;
        if (sx < 0) { ua1 = 0.f; usx0 = 0; }
        if (sx >= ups.sw - 1) { ua1 = 0.f; usx0 = ups.sw - 1; }
        usx1 = (ua1 == 0.f) ? usx0 : usx0 + 1;
    }
    auto coarse_taps = [&](int sy, float2 &p0, float2 &p1) {       // this thread's two taps of (clamped) coarse row sy
        const float2 *rowp = flow_in + (size_t)min(max(sy, 0), ups.sh - 1) * ups.sw;
        p0 = rowp[usx0];
        p1 = rowp[usx1];
    };
    auto coarse_lerp = [&](float2 p0, float2 p1) -> float2 {        // horizontal interpolation of upsampled_flow
#pragma clang fp contract(off)
        if (ua1 == 0.f) return p0;
        const float a0 = 1.f - ua1;
        return make_float2(p0.x * a0 + p1.x * ua1, p0.y * a0 + p1.y * ua1);
    };
    // consecutive steps advance by two coarse rows: rows s0+2, s0+3 of a step are rows s0, s0+1 of the next (carried,
    // interpolated), and the taps of the next step's two new rows are requested behind this step's gathers so that they
    // travel during the horizontal pass, like the flow vectors of the non-UPS form
    float2 hcar[2], pn[2][2];
    bool have_next = false;
    float2 fl_last = ups2 ? flow_at(H - 1) : make_float2(0.f, 0.f);   // flow of the last image row this thread has seen

    float ring[16][5];          // ring[row & 15] = M(clamp(row)); statically indexed everywhere below
    double v[5] = {0, 0, 0, 0, 0};

    // ---- warm-up: rows y_begin-M .. y_begin+M (replicate-clamped), four at a time (all flow vectors of a batch first,
    //      then all gathers: two memory round trips per four rows; on the short strips of the coarse levels the warm-up
    //      is a third of a work-group's life) ----
#pragma unroll
    for (int j4 = -M; j4 <= M; j4 += BS_ROWS) {
        float2 fl[BS_ROWS];
        UmIn u[BS_ROWS];
        __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int q = 0; q < BS_ROWS; q++) fl[q] = flow_at(min(max(y_begin + j4 + q, 0), H - 1));
#pragma unroll
        for (int q = 0; q < BS_ROWS; q++)
            um_load(R0, R1, plane, W, H, xc, min(max(y_begin + j4 + q, 0), H - 1), fl[q], u[q]);
        __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int q = 0; q < BS_ROWS; q++) {
            if (j4 + q <= M) {
                const int row = min(max(y_begin + j4 + q, 0), H - 1);
                float m[5];
                um_math(u[q], W, H, xc, row, fl[q], m);
#pragma unroll
                for (int c = 0; c < 5; c++) {
                    ring[(j4 + q + 16) & 15][c] = m[c];
                    v[c] += (double)m[c];
                }
            }
        }
    }

    // the flow vectors steer the gather addresses: fetched one step ahead (behind the current step's gathers) so that a
    // step does not start with a second, dependent memory round trip.  (UPS computes them from 4 coarse taps: not
    // worth 16 more loads in flight.)
    float2 fln[BS_ROWS];
    if (!UPS) {
#pragma unroll
        for (int r = 0; r < BS_ROWS; r++) fln[r] = flow_in[(size_t)min(y_begin + r + 1 + M, H - 1) * W + xc];
    }
    unsigned long long acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    auto stamp = [&](int slot) {
        if (STAMP) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_sched_barrier(0);
            if (slot >= 0) acc_t[slot] += t - tprev;
            tprev = t;
        }
    };
    stamp(-1);
    for (int y16 = y_begin; y16 < y_end; y16 += 16) {
#pragma unroll
        for (int q4 = 0; q4 < 4; q4++) {
            const int yc = y16 + 4 * q4;
            if (yc < y_end) {                                   // uniform
                // ---- advance the window over the 4 rows of this step; loads batched two rows at a time ----
                float2 fl[BS_ROWS];
                float mi[BS_ROWS][5];
                if (ups2) {
#pragma clang fp contract(off)
                    const int e0 = yc + 1 + M, s0 = (e0 + 1) / 2 - 1;      // floor(0.5 e - 0.25) of the first fine row
                    float2 hc[BS_ROWS];
                    if (have_next) {                            // uniform
                        hc[0] = hcar[0];
                        hc[1] = hcar[1];
                        hc[2] = coarse_lerp(pn[0][0], pn[0][1]);
                        hc[3] = coarse_lerp(pn[1][0], pn[1][1]);
                    } else {
#pragma unroll
                        for (int j = 0; j < BS_ROWS; j++) {
                            float2 p0, p1;
                            coarse_taps(s0 + j, p0, p1);
                            hc[j] = coarse_lerp(p0, p1);
                        }
                    }
                    hcar[0] = hc[2];
                    hcar[1] = hc[3];
                    have_next = true;
#pragma unroll
                    for (int r = 0; r < BS_ROWS; r++) {
                        // yc = 0 (mod 4), so the parity of e0 is that of 1 + M: fine rows e0+r sit between coarse rows
                        // s0 + off, s0 + off + 1 with off = 0,1,1,2 (e0 even) or 0,0,1,1 (e0 odd)
                        constexpr bool E0_EVEN = ((1 + M) & 1) == 0;
                        const int off = E0_EVEN ? (r + 1) / 2 : r / 2;
                        const float b1 = (E0_EVEN == ((r & 1) == 0)) ? 0.75f : 0.25f, b0 = 1.f - b1;
                        const float2 h0 = hc[off], h1 = hc[off + 1];
                        fl[r] = make_float2((h0.x * b0 + h1.x * b1) * ups.mul, (h0.y * b0 + h1.y * b1) * ups.mul);
                        // rows below the image replicate row H-1 (the gathers are clamped to it): so does their flow
                        if (e0 + r > H - 1) fl[r] = fl_last; else fl_last = fl[r];      // uniform
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < BS_ROWS; r++)
                        fl[r] = UPS ? flow_at(min(yc + r + 1 + M, H - 1)) : fln[r];
                }
                {
                    UmIn u[BS_ROWS];
                    // a wave that is ready to issue its gathers goes first: while its SIMD neighbour grinds through a
                    // horizontal pass, all 36 requests of this step leave at once instead of trickling out between the
                    // neighbour's VALU instructions (level-0 launch 783 -> 655 us)
                    stamp(0);                                   // 0: loop head, flow vectors / coarse taps of this step
                    __builtin_amdgcn_s_setprio(3);
#pragma unroll
                    for (int q = 0; q < BS_ROWS; q++)            // the gathers of all four rows in flight together
                        um_load(R0, R1, plane, W, H, xc, min(yc + q + 1 + M, H - 1), fl[q], u[q]);
                    if (!UPS) {
#pragma unroll
                        for (int r = 0; r < BS_ROWS; r++)
                            fln[r] = flow_in[(size_t)min(yc + BS_ROWS + r + 1 + M, H - 1) * W + xc];
                    }
                    stamp(1);                                   // 1: issuing the gathers
                    if (STAMP) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        stamp(2);                               // 2: waiting for the operands
                    }
                    __builtin_amdgcn_s_setprio(2);
#pragma unroll
                    // rows below the image replicate row H-1: the loads above were clamped to it, and the same inputs give
                    // the same matrix entries again (no carried copy, no select)
                    for (int r = 0; r < BS_ROWS; r++)
                        um_math(u[r], W, H, xc, min(yc + r + 1 + M, H - 1), fl[r], mi[r]);
                }
                if (ups2) {                 // taps of coarse rows s0'+2, s0'+3 of the next step (the gathered operands are dead now)
                    const int s0n = (yc + BS_ROWS + 1 + M + 1) / 2 - 1;
                    coarse_taps(s0n + 2, pn[0][0], pn[0][1]);
                    coarse_taps(s0n + 3, pn[1][0], pn[1][1]);
                }
                stamp(7);                                       // 7: matrix arithmetic
                // the barrier that protects `vs` from the previous step's readers sits HERE, after this step's loads and
                // matrix arithmetic: a wave that finished its horizontal pass early starts its gathers without waiting
                __syncthreads();
                stamp(3);                                       // 3: first barrier
#pragma unroll
                for (int r = 0; r < BS_ROWS; r++) {
                    const int s_in = (4 * q4 + r + 1 + M) & 15, s_out = (4 * q4 + r + 16 - M) & 15;
#pragma unroll
                    for (int c = 0; c < 5; c++) {
                        vs[c][r][vs_w] = v[c];
                        v[c] += (double)mi[r][c] - (double)ring[s_out][c];
                    }
#pragma unroll
                    for (int c = 0; c < 5; c++) ring[s_in][c] = mi[r][c];
                }
                stamp(4);                                       // 4: ring / vertical sums / exchange writes
                __syncthreads();
                stamp(5);                                       // 5: second barrier
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
                stamp(6);                                       // 6: horizontal sums + solve + stores
            }
        }
    }
    if (STAMP && dbg && lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; i++) dbg[((size_t)blockIdx.x * 4 + wave) * 8 + i] = acc_t[i];
    }
    if (SUMS) {                 // fixed order: shuffle tree inside the wave, then waves 0..3
        for (int off = 32; off >= 1; off >>= 1) {
            su += __shfl_down(su, off, 64);
            sv += __shfl_down(sv, off, 64);
        }
        __syncthreads();        // the last step's readers are done with vs
        if (lane == 0) { vs[0][0][2 * wave] = su; vs[0][0][2 * wave + 1] = sv; }
        __syncthreads();
        if (tid < 2) sums[(size_t)blockIdx.x * 2 + tid] = ((vs[0][0][tid] + vs[0][0][2 + tid]) + vs[0][0][4 + tid]) + vs[0][0][6 + tid];
    }
}

int flow_iter_rows(int W, int H, int npair, int winsize)
{
    // cost model: a launch runs in ceil(blocks / resident) rounds (224 VGPRs -> 2 work-groups per CU), each as
    // long as one strip incl. its 2m-row window warm-up.  Pick the strip count that minimises rounds x strip.
    const int tiles_x = cdiv(W, 256 - (winsize - 1));
    const int resident = 2 * 256;
    int best_rows = cdiv(H, 16) * 16;
    int64_t best_cost = LLONG_MAX;
    for (int n = 1; n <= 32; n++) {
        const int rows = cdiv(cdiv(H, n), 16) * 16;
        if (rows < 16 && n > 1) break;
        const int64_t blocks = (int64_t)tiles_x * cdiv(H, rows) * npair;
        const int64_t cost = cdiv64(blocks, resident) * (rows + winsize - 1);
        if (cost < best_cost) { best_cost = cost; best_rows = rows; }
    }
    return best_rows;
}

int flow_iter_max_grid(int W, int H, int npair, int winsize)
{
    const int rows = flow_iter_rows(W, H, npair, winsize);
    return cdiv(cdiv(W, 256 - (winsize - 1)) * cdiv(H, rows), 8) * 8 * npair;
}

int launch_flow_iter(const float *R, size_t frame_stride_R, const float *flow_in, float *flow_out,
                     int npair, int W, int H, int winsize, hipStream_t s, const float *coarse, int sw, int sh,
                     float mul, double *uv_sum, double *uv_scratch, size_t uv_scratch_doubles)
{
    if (winsize > 15) { set_error("fused iteration supports winsize <= 15 (ring of 16 rows)"); return OFC_EUNSUPPORTED; }
    if ((int64_t)W * H * 5 >= (1ll << 30)) { set_error("frame too large for 32-bit R offsets (%dx%d)", W, H); return OFC_EUNSUPPORTED; }
    const int rows_per_block = flow_iter_rows(W, H, npair, winsize);
    UpsArgs u;
    u.src = coarse; u.sw = sw; u.sh = sh; u.mul = mul;
    u.scx = coarse ? (double)sw / W : 1.0; u.scy = coarse ? (double)sh / H : 1.0;
    // OFC_FLOW_PIPE=1: the software-pipelined experiment of flow_experiments.hip (bit-identical output, 2.6x slower: it
    // does not fit 256 VGPRs); read at every launch because its parity test toggles it
    {
        const char *e = getenv("OFC_FLOW_PIPE");
        if (e && e[0] == '1' && winsize == 15 && !coarse)
            return launch_flow_iter_pipe(R, frame_stride_R, flow_in, flow_out, npair, W, H, rows_per_block, s, uv_sum, uv_scratch,
                                         uv_scratch_doubles);
    }
    if (uv_sum) {               // the last iteration of level 0: also sum(u), sum(v) of the field it writes -> uv_sum[2]
        if (coarse || winsize != 15) { set_error("flow sums are emitted by the plain winsize-15 iteration only"); return OFC_EUNSUPPORTED; }
        const int tx = cdiv(W, 256 - 14), ns = cdiv(H, rows_per_block);
        const int grid = cdiv(tx * ns, 8) * 8 * npair;
        if ((size_t)grid * 2 > uv_scratch_doubles) { set_error("flow sums: scratch too small (%d work-groups)", grid); return OFC_EINVAL; }
        hipLaunchKernelGGL((k_flow_iter<7, 0, false, true>), dim3(grid), dim3(256), 0, s, R, frame_stride_R, flow_in, flow_out, W, H,
                           rows_per_block, u, tx, ns, npair, nullptr, uv_scratch);
        OFC_HIP(hipGetLastError());
        return launch_reduce_records(uv_scratch, grid, 2, uv_sum, s, nullptr);
    }
    dim3 block(256);
#define OFC_FI_CASE(MM)                                                                                  \
    case 2 * MM + 1: {                                                                                   \
        const int tx = cdiv(W, 256 - 2 * MM), ns = cdiv(H, rows_per_block);                              \
        dim3 grid(cdiv(tx * ns, 8) * 8 * npair);                                                         \
        if (coarse && u.scx == 0.5 && u.scy == 0.5)                                                      \
            hipLaunchKernelGGL((k_flow_iter<MM, 2>), grid, block, 0, s, R, frame_stride_R, flow_in,      \
                               flow_out, W, H, rows_per_block, u, tx, ns, npair);                        \
        else if (coarse)                                                                                 \
            hipLaunchKernelGGL((k_flow_iter<MM, 1>), grid, block, 0, s, R, frame_stride_R, flow_in,      \
                               flow_out, W, H, rows_per_block, u, tx, ns, npair);                        \
        else                                                                                             \
            hipLaunchKernelGGL((k_flow_iter<MM, 0>), grid, block, 0, s, R, frame_stride_R, flow_in,      \
                               flow_out, W, H, rows_per_block, u, tx, ns, npair);                        \
        break;                                                                                           \
    }
    switch (winsize) {
        OFC_FI_CASE(2)
        OFC_FI_CASE(3)
        OFC_FI_CASE(4)
        OFC_FI_CASE(5)
        OFC_FI_CASE(6)
        OFC_FI_CASE(7)
    default:
        set_error("winsize %d unsupported by the fused iteration (odd 5..15)", winsize);
        return OFC_EUNSUPPORTED;
    }
#undef OFC_FI_CASE
    OFC_HIP(hipGetLastError());
    return OFC_OK;
}

// diagnostic launch of the stamped build (non-UPS, winsize 15); dbg: [grid][4][8] u64, zeroed by the caller
int launch_flow_iter_stamped(const float *R, size_t frame_stride_R, const float *flow_in, float *flow_out, int npair, int W,
                             int H, hipStream_t s, unsigned long long *dbg, int *grid_out)
{
    const int rows_per_block = flow_iter_rows(W, H, npair, 15);
    UpsArgs u;
    u.src = nullptr; u.sw = 0; u.sh = 0; u.mul = 1.f; u.scx = 1.0; u.scy = 1.0;
    const int tx = cdiv(W, 256 - 14), ns = cdiv(H, rows_per_block);
    dim3 grid(cdiv(tx * ns, 8) * 8 * npair);
    if (grid_out) *grid_out = (int)grid.x;
    if (dbg)
        hipLaunchKernelGGL((k_flow_iter<7, 0, true>), grid, dim3(256), 0, s, R, frame_stride_R, flow_in, flow_out, W, H,
                           rows_per_block, u, tx, ns, npair, dbg);
    OFC_HIP(hipGetLastError());
    return OFC_OK;
}

// ------------------------------------------------------------------------------------------------
// K6  flow upsample: resize(prevFlow, (w,h), INTER_LINEAR) * (1/pyr_scale).  10 B/px.  Bit-exact.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_flow_resize(const float2 *__restrict__ src,
                                                     float2 *__restrict__ dst, int sw, int sh, int dw,
                                                     int dh, double scx, double scy, float mul)
{
#pragma clang fp contract(off)
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= dw) return;
    const float2 *s = src + (size_t)blockIdx.z * sw * sh;
    float2 *d = dst + (size_t)blockIdx.z * dw * dh;
    if (sw == dw && sh == dh) {
        float2 v = s[(size_t)y * sw + x];
        d[(size_t)y * dw + x] = make_float2(v.x * mul, v.y * mul);
        return;
    }
    int sx0, sy0, sy1;
    float a1, b1;
    lin_tap_x(x, scx, sw, sx0, a1);
    lin_tap_y(y, scy, sh, sy0, sy1, b1);
    const float a0 = 1.f - a1, b0 = 1.f - b1;
    const int sx1 = (a1 == 0.f) ? sx0 : sx0 + 1;
    const float2 p00 = s[(size_t)sy0 * sw + sx0], p01 = s[(size_t)sy0 * sw + sx1];
    const float2 p10 = s[(size_t)sy1 * sw + sx0], p11 = s[(size_t)sy1 * sw + sx1];
    float h0x, h0y, h1x, h1y;
    if (a1 == 0.f) {
        h0x = p00.x; h0y = p00.y; h1x = p10.x; h1y = p10.y;
    } else {
        h0x = p00.x * a0 + p01.x * a1; h0y = p00.y * a0 + p01.y * a1;
        h1x = p10.x * a0 + p11.x * a1; h1y = p10.y * a0 + p11.y * a1;
    }
    d[(size_t)y * dw + x] = make_float2((h0x * b0 + h1x * b1) * mul, (h0y * b0 + h1y * b1) * mul);
}

int launch_flow_resize(const float *src, float *dst, int npair, int sw, int sh, int dw, int dh,
                       float mul, hipStream_t s)
{
    dim3 grid(cdiv(dw, 256), dh, npair);
    hipLaunchKernelGGL(k_flow_resize, grid, dim3(256), 0, s, reinterpret_cast<const float2 *>(src),
                       reinterpret_cast<float2 *>(dst), sw, sh, dw, dh, (double)sw / dw,
                       (double)sh / dh, mul);
    OFC_HIP(hipGetLastError());
    return OFC_OK;
}

}  // namespace ofc
