// color_kernels.hip -- the small 8-bit kernels either side of the flow:
// BGR2GRAY, flow -> HSV-coded BGR visualisation (cartToPolar, per-frame MINMAX normalisation, uint8
// truncation, HSV2BGR), per-grid-cell mean colour -> BGR2HSV.  All HBM-bound byte kernels; FP contraction
// is OFF so they reproduce the oracle (oracle/color_ref.c == SURVEY.md App. C) bit for bit.
#include "color_common.h"

#include <cfloat>

namespace ofc {

// ------------------------------------------------------------------------------------------------
// K1  cvtColor(BGR2GRAY) u8: 15-bit fixed point.  4 px per lane: 12 B in, 4 B out.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned gray_of(unsigned b, unsigned g, unsigned r)
{
    return (b * 3735u + g * 19235u + r * 9798u + (1u << 14)) >> 15;
}

__global__ __launch_bounds__(256) void k_bgr2gray(const uint8_t *__restrict__ bgr, uint8_t *__restrict__ gray,
                                                  int64_t npix)
{
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i = q * 4;
    if (i + 3 < npix) {
        const uint32_t *p = reinterpret_cast<const uint32_t *>(bgr + i * 3);
        const uint32_t w0 = p[0], w1 = p[1], w2 = p[2];
        const unsigned g0 = gray_of(w0 & 255, (w0 >> 8) & 255, (w0 >> 16) & 255);
        const unsigned g1 = gray_of(w0 >> 24, w1 & 255, (w1 >> 8) & 255);
        const unsigned g2 = gray_of((w1 >> 16) & 255, w1 >> 24, w2 & 255);
        const unsigned g3 = gray_of((w2 >> 8) & 255, (w2 >> 16) & 255, w2 >> 24);
        reinterpret_cast<uint32_t *>(gray)[q] = g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
    } else {
        for (int64_t j = i; j < npix; j++) gray[j] = (uint8_t)gray_of(bgr[j * 3], bgr[j * 3 + 1], bgr[j * 3 + 2]);
    }
}

int launch_bgr2gray(const uint8_t *bgr, uint8_t *gray, int64_t npix, hipStream_t s)
{
    const int64_t nq = cdiv64(npix, 4);
    hipLaunchKernelGGL(k_bgr2gray, dim3((unsigned)cdiv64(nq, 256)), dim3(256), 0, s, bgr, gray, npix);
    OFC_HIP(hipGetLastError());
    return OFC_OK;
}

__global__ __launch_bounds__(256) void k_bgr2hsv(const uint8_t *__restrict__ bgr, uint8_t *__restrict__ hsv,
                                                 int64_t npix)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    unsigned h, s, v;
    bgr2hsv_u8(bgr[i * 3], bgr[i * 3 + 1], bgr[i * 3 + 2], h, s, v);
    hsv[i * 3] = h; hsv[i * 3 + 1] = s; hsv[i * 3 + 2] = v;
}

int launch_bgr2hsv(const uint8_t *bgr, uint8_t *hsv, int64_t npix, hipStream_t s)
{
    hipLaunchKernelGGL(k_bgr2hsv, dim3((unsigned)cdiv64(npix, 256)), dim3(256), 0, s, bgr, hsv, npix);
    OFC_HIP(hipGetLastError());
    return OFC_OK;
}

// K10  preprocess_image: per-channel threshold, alpha from the grey value; 3 B in, 4 B out per pixel
__global__ __launch_bounds__(256) void k_preprocess_rgba(const uint8_t *__restrict__ img, uint32_t *__restrict__ rgba,
                                                         int64_t npix, int thresh)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    unsigned c0 = img[i * 3], c1 = img[i * 3 + 1], c2 = img[i * 3 + 2];
    c0 = c0 < (unsigned)thresh ? 0u : c0;
    c1 = c1 < (unsigned)thresh ? 0u : c1;
    c2 = c2 < (unsigned)thresh ? 0u : c2;
    const unsigned alpha = gray_of(c0, c1, c2) > 0 ? 255u : 0u;
    rgba[i] = c0 | (c1 << 8) | (c2 << 16) | (alpha << 24);
}

int launch_preprocess_rgba(const uint8_t *img, uint8_t *rgba, int64_t npix, int thresh, hipStream_t s)
{
    hipLaunchKernelGGL(k_preprocess_rgba, dim3((unsigned)cdiv64(npix, 256)), dim3(256), 0, s, img,
                       reinterpret_cast<uint32_t *>(rgba), npix, thresh);
    OFC_HIP(hipGetLastError());
    return OFC_OK;
}

// ------------------------------------------------------------------------------------------------
// K7/K8  flow -> BGR.  Pass 1: per-frame min/max of the magnitude + f64 sum (for np.mean(magnitude)).
// Pass 2: recompute magnitude/angle, normalise, truncate, HSV2BGR.  8 B/px read twice + 3 B/px written.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float mag_of(float x, float y)
{
#pragma clang fp contract(off)
    return sqrtf(x * x + y * y);
}

__device__ __forceinline__ float atan_deg(float y, float x)
{
#pragma clang fp contract(off)
    const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
    const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
    const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
    const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// record per block: {min, max, sum}
__global__ __launch_bounds__(256) void k_mag_stats(const float2 *__restrict__ flow, int64_t npix,
                                                   double *__restrict__ partial)
{
    __shared__ float smn[4], smx[4];
    __shared__ double ssum[4];
    const float2 *f = flow + (size_t)blockIdx.y * npix;
    float mn = FLT_MAX, mx = -FLT_MAX;
    double sum = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
        const float2 v = f[i];
        const float m = mag_of(v.x, v.y);
        mn = fminf(mn, m);
        mx = fmaxf(mx, m);
        sum += (double)m;
    }
    for (int off = 32; off >= 1; off >>= 1) {
        mn = fminf(mn, __shfl_down(mn, off, 64));
        mx = fmaxf(mx, __shfl_down(mx, off, 64));
        sum += __shfl_down(sum, off, 64);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { smn[wave] = mn; smx[wave] = mx; ssum[wave] = sum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double *o = partial + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 3;
        o[0] = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
        o[1] = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
        o[2] = ((ssum[0] + ssum[1]) + ssum[2]) + ssum[3];
    }
}

// one block per frame: fold the block records, derive normalize()'s scale/shift
__global__ __launch_bounds__(64) void k_mag_finish(const double *__restrict__ partial, int nblocks,
                                                   int64_t npix, VisFrameStats *__restrict__ stats,
                                                   float *__restrict__ mean_mag)
{
    if (threadIdx.x != 0) return;
    const double *p = partial + (size_t)blockIdx.x * nblocks * 3;
    double mn = p[0], mx = p[1], sum = 0;
    for (int b = 0; b < nblocks; b++) {
        mn = fmin(mn, p[3 * b]);
        mx = fmax(mx, p[3 * b + 1]);
        sum += p[3 * b + 2];
    }
    const double dscale = 255. * (mx - mn > DBL_EPSILON ? 1. / (mx - mn) : 0);
    const double dshift = 0. - mn * dscale;
    stats[blockIdx.x].a = (float)dscale;
    stats[blockIdx.x].b = (float)dshift;
    if (mean_mag) mean_mag[blockIdx.x] = (float)(sum / (double)npix);
}

__device__ __forceinline__ void hsv2bgr_u8(unsigned H, unsigned S, unsigned V, unsigned &B, unsigned &G, unsigned &R)
{
#pragma clang fp contract(off)
    float h = (float)H, s = (float)S * (1.f / 255.f), v = (float)V * (1.f / 255.f);
    float b, g, r;
    if (s == 0) {
        b = g = r = v;
    } else {
        h *= (6.f / 180.f);
        if (h < 0) do h += 6; while (h < 0);
        else if (h >= 6) do h -= 6; while (h >= 6);
        int sector = (int)floorf(h);
        h -= (float)sector;
        if ((unsigned)sector >= 6u) { sector = 0; h = 0.f; }
        const float t0 = v, t1 = v * (1.f - s), t2 = v * (1.f - s * h), t3 = v * (1.f - s * (1.f - h));
        // sector_data = {1,3,0},{1,0,2},{3,0,1},{0,2,1},{0,1,3},{2,1,0}  (b,g,r indices into tab)
        switch (sector) {
        case 0: b = t1; g = t3; r = t0; break;
        case 1: b = t1; g = t0; r = t2; break;
        case 2: b = t3; g = t0; r = t1; break;
        case 3: b = t0; g = t2; r = t1; break;
        case 4: b = t0; g = t1; r = t3; break;
        default: b = t2; g = t1; r = t0; break;
        }
    }
    const float fb = b * 255.f, fg = g * 255.f, fr = r * 255.f;
    B = fb < 0 ? 0u : fb > 255 ? 255u : (unsigned)(int)fb;
    G = fg < 0 ? 0u : fg > 255 ? 255u : (unsigned)(int)fg;
    R = fr < 0 ? 0u : fr > 255 ? 255u : (unsigned)(int)fr;
}

// 4 px per lane: 32 B read, 12 B written
__global__ __launch_bounds__(256) void k_flow_colorize(const float2 *__restrict__ flow, int64_t npix,
                                                       const VisFrameStats *__restrict__ stats,
                                                       uint8_t *__restrict__ bgr)
{
#pragma clang fp contract(off)
    const float2 *f = flow + (size_t)blockIdx.y * npix;
    uint8_t *o = bgr + (size_t)blockIdx.y * npix * 3;
    const float a = stats[blockIdx.y].a, b = stats[blockIdx.y].b;
    const float pif = (float)M_PI, rad = (float)(M_PI / 180);
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i0 = q * 4;
    if (i0 >= npix) return;
    unsigned px[4][3];
    const int n = (int)min((int64_t)4, npix - i0);
    for (int p = 0; p < n; p++) {
        const float2 v = f[i0 + p];
        const float m = mag_of(v.x, v.y);
        const float ang = atan_deg(v.y, v.x) * rad;
        const float hf = ang * 180.f / pif / 2.f;
        const float vf = m * a + b;
        const unsigned Hh = (unsigned)(int)hf & 255u;
        const unsigned Vv = vf < 0 ? 0u : vf > 255 ? 255u : (unsigned)(int)vf;
        hsv2bgr_u8(Hh, 255u, Vv, px[p][0], px[p][1], px[p][2]);
    }
    if (n == 4) {
        uint32_t *w = reinterpret_cast<uint32_t *>(o + i0 * 3);
        w[0] = px[0][0] | (px[0][1] << 8) | (px[0][2] << 16) | (px[1][0] << 24);
        w[1] = px[1][1] | (px[1][2] << 8) | (px[2][0] << 16) | (px[2][1] << 24);
        w[2] = px[2][2] | (px[3][0] << 8) | (px[3][1] << 16) | (px[3][2] << 24);
    } else {
        for (int p = 0; p < n; p++)
            for (int c = 0; c < 3; c++) o[(i0 + p) * 3 + c] = (uint8_t)px[p][c];
    }
}

int launch_flow_to_bgr(const float *flow, int W, int H, int nframes, uint8_t *bgr, float *mean_mag_dev,
                       double *partial /* nframes*VIS_BLOCKS*3 */, VisFrameStats *stats, hipStream_t s)
{
    const int64_t npix = (int64_t)W * H;
    hipLaunchKernelGGL(k_mag_stats, dim3(VIS_BLOCKS, nframes), dim3(256), 0, s,
                       reinterpret_cast<const float2 *>(flow), npix, partial);
    hipLaunchKernelGGL(k_mag_finish, dim3(nframes), dim3(64), 0, s, partial, VIS_BLOCKS, npix, stats, mean_mag_dev);
    hipLaunchKernelGGL(k_flow_colorize, dim3((unsigned)cdiv64(cdiv64(npix, 4), 256), nframes), dim3(256), 0, s,
                       reinterpret_cast<const float2 *>(flow), npix, stats, bgr);
    OFC_HIP(hipGetLastError());
    return OFC_OK;
}

// ------------------------------------------------------------------------------------------------
// K9  grid cells: per-cell mean BGR -> uint8 (truncation) -> BGR2HSV.  One work-group per cell.
// The white grid lines cv2.rectangle has already painted when a cell is averaged are applied
// analytically (row 0 white iff cy >= 1, column 0 iff cx >= 1) -- SURVEY.md App. C.7.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_grid_cell_means(const uint8_t *__restrict__ bgr, int W, int H,
                                                         int rows, int cols, uint8_t *__restrict__ mean_bgr,
                                                         uint8_t *__restrict__ hsv)
{
    __shared__ unsigned long long ssum[4][3];
    const int cell = blockIdx.x, cy = cell / cols, cx = cell % cols;
    const int xs = W / cols, ys = H / rows;
    const uint8_t *frame = bgr + (size_t)blockIdx.y * W * H * 3;
    unsigned s0 = 0, s1 = 0, s2 = 0;
    for (int i = threadIdx.x; i < xs * ys; i += 256) {
        const int ly = i / xs, lx = i - ly * xs;
        const bool white = (cy >= 1 && ly == 0) || (cx >= 1 && lx == 0);
        const uint8_t *p = frame + ((size_t)(cy * ys + ly) * W + cx * xs + lx) * 3;
        s0 += white ? 255u : p[0];
        s1 += white ? 255u : p[1];
        s2 += white ? 255u : p[2];
    }
    unsigned long long t0 = s0, t1 = s1, t2 = s2;
    for (int off = 32; off >= 1; off >>= 1) {
        t0 += __shfl_down(t0, off, 64);
        t1 += __shfl_down(t1, off, 64);
        t2 += __shfl_down(t2, off, 64);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { ssum[wave][0] = t0; ssum[wave][1] = t1; ssum[wave][2] = t2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned m[3];
        const double cnt = (double)xs * ys;
        for (int c = 0; c < 3; c++) {
            const unsigned long long tot = ssum[0][c] + ssum[1][c] + ssum[2][c] + ssum[3][c];
            m[c] = (unsigned)(int)((double)tot / cnt);
        }
        unsigned h, sv, v;
        bgr2hsv_u8(m[0], m[1], m[2], h, sv, v);
        const size_t o = ((size_t)blockIdx.y * rows * cols + cell) * 3;
        mean_bgr[o] = m[0]; mean_bgr[o + 1] = m[1]; mean_bgr[o + 2] = m[2];
        hsv[o] = h; hsv[o + 1] = sv; hsv[o + 2] = v;
    }
}

int launch_grid_cell_means(const uint8_t *bgr, int W, int H, int nframes, int rows, int cols,
                           uint8_t *mean_bgr, uint8_t *hsv, hipStream_t s)
{
    hipLaunchKernelGGL(k_grid_cell_means, dim3(rows * cols, nframes), dim3(256), 0, s, bgr, W, H, rows, cols,
                       mean_bgr, hsv);
    OFC_HIP(hipGetLastError());
    return OFC_OK;
}

// ------------------------------------------------------------------------------------------------
// grid-cell averaged flow: per cell the mean (u, v) over its x_step x y_step pixels (f64 sums, f32 result).
// One work-group per (cell, pair); 8 B/px read.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_grid_cell_mean_flow(const float2 *__restrict__ flow, int W, int H, int rows,
                                                             int cols, float2 *__restrict__ out)
{
    __shared__ double su[4], sv[4];
    const int cell = blockIdx.x, cy = cell / cols, cx = cell % cols;
    const int xs = W / cols, ys = H / rows;
    const float2 *f = flow + (size_t)blockIdx.y * W * H;
    double u = 0, v = 0;
    for (int i = threadIdx.x; i < xs * ys; i += 256) {
        const int ly = i / xs, lx = i - ly * xs;
        const float2 p = f[(size_t)(cy * ys + ly) * W + cx * xs + lx];
        u += (double)p.x; v += (double)p.y;
    }
    for (int off = 32; off >= 1; off >>= 1) { u += __shfl_down(u, off, 64); v += __shfl_down(v, off, 64); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { su[wave] = u; sv[wave] = v; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double n = (double)xs * ys;
        out[(size_t)blockIdx.y * rows * cols + cell] =
            make_float2((float)((((su[0] + su[1]) + su[2]) + su[3]) / n), (float)((((sv[0] + sv[1]) + sv[2]) + sv[3]) / n));
    }
}

int launch_grid_cell_mean_flow(const float *flow, int W, int H, int npair, int rows, int cols, float *out, hipStream_t s)
{
    hipLaunchKernelGGL(k_grid_cell_mean_flow, dim3(rows * cols, npair), dim3(256), 0, s,
                       reinterpret_cast<const float2 *>(flow), W, H, rows, cols, reinterpret_cast<float2 *>(out));
    OFC_HIP(hipGetLastError());
    return OFC_OK;
}

// ------------------------------------------------------------------------------------------------
// sliding-window cosine similarity (findCosineDifferentVectors.py:5-61): one work-group per window offset.
// np.dot / np.linalg.norm on the integer hue columns are exact integer sums; a value v with |v| < 2^31 and integral
// goes through the int64 accumulators, anything else through f64 (tree order).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sliding_cosine(const double *__restrict__ a, int na,
                                                        const double *__restrict__ b, double *__restrict__ sims,
                                                        int all_int)
{
    __shared__ double sd[3][4];
    __shared__ long long si[3][4];
    const double *w = b + blockIdx.x;
    double fd = 0, fa = 0, fb = 0;
    long long id = 0, ia = 0, ib = 0;
    for (int i = threadIdx.x; i < na; i += 256) {
        if (all_int) {
            const long long x = (long long)a[i], y = (long long)w[i];
            id += x * y; ia += x * x; ib += y * y;
        } else {
            fd += a[i] * w[i]; fa += a[i] * a[i]; fb += w[i] * w[i];
        }
    }
    for (int off = 32; off >= 1; off >>= 1) {
        fd += __shfl_down(fd, off, 64); fa += __shfl_down(fa, off, 64); fb += __shfl_down(fb, off, 64);
        id += __shfl_down(id, off, 64); ia += __shfl_down(ia, off, 64); ib += __shfl_down(ib, off, 64);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { sd[0][wave] = fd; sd[1][wave] = fa; sd[2][wave] = fb; si[0][wave] = id; si[1][wave] = ia; si[2][wave] = ib; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double dot, na2, nb2;
        if (all_int) {
            dot = (double)(si[0][0] + si[0][1] + si[0][2] + si[0][3]);
            na2 = (double)(si[1][0] + si[1][1] + si[1][2] + si[1][3]);
            nb2 = (double)(si[2][0] + si[2][1] + si[2][2] + si[2][3]);
        } else {
            dot = ((sd[0][0] + sd[0][1]) + sd[0][2]) + sd[0][3];
            na2 = ((sd[1][0] + sd[1][1]) + sd[1][2]) + sd[1][3];
            nb2 = ((sd[2][0] + sd[2][1]) + sd[2][2]) + sd[2][3];
        }
        const double n1 = sqrt(na2), n2 = sqrt(nb2);
        sims[blockIdx.x] = (n1 == 0 || n2 == 0) ? 0.0 : dot / (n1 * n2);
    }
}

int launch_sliding_cosine(const double *a, int na, const double *b, int nwin, double *sims, int all_int, hipStream_t s)
{
    hipLaunchKernelGGL(k_sliding_cosine, dim3(nwin), dim3(256), 0, s, a, na, b, sims, all_int);
    OFC_HIP(hipGetLastError());
    return OFC_OK;
}

// ------------------------------------------------------------------------------------------------
// synthetic frames for bench.py (not part of the reference path): analytic multi-sinusoid texture,
// five motion populations in vertical bands, frame t displaced by t * velocity.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_synth_frames(uint8_t *__restrict__ frames, int W, int H, int t0,
                                                      SynthParams sp)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, t = t0 + blockIdx.z;
    if (x >= W) return;
    const int band = min(x * SYNTH_POP / W, SYNTH_POP - 1);
    const float fx = (float)x - sp.vx[band] * (float)t, fy = (float)y - sp.vy[band] * (float)t;
    float acc = 0.f;
#pragma unroll 4
    for (int i = 0; i < SYNTH_WAVES; i++)
        acc += sp.a[i] * __sinf(6.2831853f * (sp.fx[i] * fx + sp.fy[i] * fy) + sp.ph[i]);
    float v = 127.5f + 100.f * acc * sp.inv_norm;
    v = fminf(fmaxf(v, 0.f), 255.f);
    frames[((size_t)blockIdx.z * H + y) * W + x] = (uint8_t)(int)v;
}

int launch_synth_frames(uint8_t *frames, int W, int H, int nframes, int t0, const SynthParams &sp, hipStream_t s)
{
    hipLaunchKernelGGL(k_synth_frames, dim3(cdiv(W, 256), H, nframes), dim3(256), 0, s, frames, W, H, t0, sp);
    OFC_HIP(hipGetLastError());
    return OFC_OK;
}

}  // namespace ofc
