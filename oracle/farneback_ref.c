/*
 * farneback_ref.c -- CPU oracle for cv2.calcOpticalFlowFarneback as the reference calls it
 * (k-means-color-clustering/computeOpticalFlowModule.py:20-22, computeOpticalFlow.py:99-101:
 *  pyr_scale .5, levels 3, winsize 15, iterations 3, poly_n 5, poly_sigma 1.2, flags 0).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle_ref.h).
 *
 * The algorithm lives in OpenCV (opencv-python, unpinned, absent from /root/reference and from
 * this image).  This file restates OpenCV's published CPU path -- modules/video/src/optflowgf.cpp
 * (FarnebackPrepareGaussian / FarnebackPolyExp / FarnebackUpdateMatrices /
 * FarnebackUpdateFlow_Blur / FarnebackOpticalFlowImpl::calc), imgproc GaussianBlur + resize --
 * loop for loop, with the same float/double placement and borders, as catalogued in
 * SURVEY.md App. A.1-A.5.  PARITY UNPINNED: no flow field is recorded anywhere in the reference;
 * acceptance numbers are the known-translation figures of SURVEY.md App. A.9.
 *
 * Built with -ffp-contract=off so every product and sum rounds once, in the order written.
 */
#include "oracle_ref.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

static int iround_even(double v) { return (int)rint(v); }           /* cvRound */
static int ifloor(float v) { int i = (int)v; return i - (v < (float)i); } /* cvFloor */

static int reflect101(int p, int len)
{
    /* BORDER_REFLECT_101: gfedcb|abcdefgh|gfedcba */
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    }
    return p;
}

/* imgproc getGaussianKernel(n, sigma, CV_32F) -- SURVEY.md App. A.2 */
void ofc_ref_gaussian_kernel(int n, double sigma, float *k)
{
    static const float small_tab[4][7] = {
        {1.f},
        {0.25f, 0.5f, 0.25f},
        {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f},
        {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f}};
    const float *fixed = (n % 2 == 1 && n <= 7 && sigma <= 0) ? small_tab[n >> 1] : 0;
    double sigmaX = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    double scale2X = -0.5 / (sigmaX * sigmaX);
    double sum = 0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        double t = fixed ? (double)fixed[i] : exp(scale2X * x * x);
        k[i] = (float)t;
        sum += k[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) k[i] = (float)(k[i] * sum);
}

/* GaussianBlur on a 32F image: separable, row pass then column pass, f32 accumulation,
 * BORDER_REFLECT_101.  Row pass sums the taps left to right; column pass is the symmetric
 * form (centre, then pairs) -- SURVEY.md App. A.2. */
void ofc_ref_gaussian_blur(const float *src, int W, int H, int ksize, double sigma, float *dst)
{
    if (ksize == 1) { memcpy(dst, src, sizeof(float) * (size_t)W * H); return; }
    float kern[64];
    ofc_ref_gaussian_kernel(ksize, sigma, kern);
    int r = ksize / 2;
    float *tmp = (float *)malloc(sizeof(float) * (size_t)W * H);
    int *xi = (int *)malloc(sizeof(int) * (size_t)(W + 2 * r));
    for (int x = -r; x < W + r; x++) xi[x + r] = reflect101(x, W);
    for (int y = 0; y < H; y++) {
        const float *s = src + (size_t)y * W;
        float *t = tmp + (size_t)y * W;
        for (int x = 0; x < W; x++) {
            float acc = kern[0] * s[xi[x]];
            for (int j = 1; j < ksize; j++) acc += kern[j] * s[xi[x + j]];
            t[x] = acc;
        }
    }
    for (int y = 0; y < H; y++) {
        float *d = dst + (size_t)y * W;
        const float *c = tmp + (size_t)y * W;
        for (int x = 0; x < W; x++) d[x] = kern[r] * c[x];
        for (int j = 1; j <= r; j++) {
            const float *a = tmp + (size_t)reflect101(y - j, H) * W;
            const float *b = tmp + (size_t)reflect101(y + j, H) * W;
            float kj = kern[r + j];
            for (int x = 0; x < W; x++) d[x] += kj * (a[x] + b[x]);
        }
    }
    free(xi);
    free(tmp);
}

/* imgproc resize(..., INTER_LINEAR) for 32F, cn interleaved channels -- SURVEY.md App. A.2.
 * Horizontal interpolation first, then vertical, weights in f32. */
void ofc_ref_resize_linear(const float *src, int sw, int sh, int cn, float *dst, int dw, int dh)
{
    if (sw == dw && sh == dh) {
        memcpy(dst, src, sizeof(float) * (size_t)sw * sh * cn);
        return;
    }
    double scale_x = (double)sw / dw, scale_y = (double)sh / dh;
    int *xofs = (int *)malloc(sizeof(int) * dw);
    float *xa = (float *)malloc(sizeof(float) * dw);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = ifloor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        xa[dx] = fx;
    }
    float *h0 = (float *)malloc(sizeof(float) * (size_t)dw * cn);
    float *h1 = (float *)malloc(sizeof(float) * (size_t)dw * cn);
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = ifloor(fy);
        fy -= sy;
        int sy0 = sy < 0 ? 0 : (sy > sh - 1 ? sh - 1 : sy);
        int sy1 = sy + 1 < 0 ? 0 : (sy + 1 > sh - 1 ? sh - 1 : sy + 1);
        float b0 = 1.f - fy, b1 = fy;
        const float *r0 = src + (size_t)sy0 * sw * cn, *r1 = src + (size_t)sy1 * sw * cn;
        for (int dx = 0; dx < dw; dx++) {
            int sx = xofs[dx];
            float a1 = xa[dx], a0 = 1.f - a1;
            int sx1 = (a1 == 0.f) ? sx : sx + 1;
            for (int c = 0; c < cn; c++) {
                if (a1 == 0.f) {
                    h0[dx * cn + c] = r0[sx * cn + c];
                    h1[dx * cn + c] = r1[sx * cn + c];
                } else {
                    h0[dx * cn + c] = r0[sx * cn + c] * a0 + r0[sx1 * cn + c] * a1;
                    h1[dx * cn + c] = r1[sx * cn + c] * a0 + r1[sx1 * cn + c] * a1;
                }
            }
        }
        float *d = dst + (size_t)dy * dw * cn;
        for (int i = 0; i < dw * cn; i++) d[i] = h0[i] * b0 + h1[i] * b1;
    }
    free(h0); free(h1); free(xofs); free(xa);
}

/* FarnebackPrepareGaussian -- SURVEY.md App. A.3.  g/xg/xxg are indexed [0..n] (symmetric).
 * ig4 = {ig11, ig03, ig33, ig55}. */
void ofc_ref_polyexp_setup(int n, double sigma, float *g, float *xg, float *xxg, double *ig4)
{
    if (sigma < FLT_EPSILON) sigma = n * 0.3;
    double s = 0.;
    float gf[2 * 16 + 1];
    for (int x = -n; x <= n; x++) {
        gf[x + n] = (float)exp(-x * x / (2 * sigma * sigma));
        s += gf[x + n];
    }
    s = 1. / s;
    for (int x = -n; x <= n; x++) gf[x + n] = (float)(gf[x + n] * s);
    for (int x = 0; x <= n; x++) {
        g[x] = gf[x + n];
        xg[x] = (float)(x * g[x]);
        xxg[x] = (float)(x * x * g[x]);
    }
    double G00 = 0, G11 = 0, G33 = 0, G55 = 0;
    for (int y = -n; y <= n; y++)
        for (int x = -n; x <= n; x++) {
            float gg = gf[y + n] * gf[x + n];
            G00 += gg;
            G11 += gg * x * x;
            G33 += gg * x * x * x * x;
            G55 += gg * x * x * y * y;
        }
    /* G couples {0,3,4}: [[G00,G11,G11],[G11,G33,G55],[G11,G55,G33]]; 1,2 -> G11; 5 -> G55.
     * invG by cofactors of the symmetric 3x3 block (OpenCV: Cholesky; same to ~1e-16). */
    double a = G00, b = G11, c = G33, e = G55;
    double det = a * (c * c - e * e) - 2 * b * b * (c - e);
    ig4[0] = 1. / G11;                       /* ig11 */
    ig4[1] = -b * (c - e) / det;             /* ig03 */
    ig4[2] = (a * c - b * b) / det;          /* ig33 */
    ig4[3] = 1. / G55;                       /* ig55 */
}

/* FarnebackPolyExp -- SURVEY.md App. A.3.  dst is HxWx5 interleaved:
 * {y-linear, x-linear, y^2, x^2, xy}. */
void ofc_ref_polyexp(const float *src, int W, int H, int n, double sigma, float *dst5)
{
    float g[17], xg[17], xxg[17];
    double ig[4];
    ofc_ref_polyexp_setup(n, sigma, g, xg, xxg, ig);
    double ig11 = ig[0], ig03 = ig[1], ig33 = ig[2], ig55 = ig[3];
    float *rowbuf = (float *)malloc(sizeof(float) * (size_t)(W + 2 * n) * 3);
    float *row = rowbuf + n * 3;
    for (int y = 0; y < H; y++) {
        float g0 = g[0], g1, g2;
        const float *srow0 = src + (size_t)y * W, *srow1;
        float *drow = dst5 + (size_t)y * W * 5;
        for (int x = 0; x < W; x++) {
            row[x * 3] = srow0[x] * g0;
            row[x * 3 + 1] = row[x * 3 + 2] = 0.f;
        }
        for (int k = 1; k <= n; k++) {
            g0 = g[k]; g1 = xg[k]; g2 = xxg[k];
            srow0 = src + (size_t)(y - k > 0 ? y - k : 0) * W;
            srow1 = src + (size_t)(y + k < H - 1 ? y + k : H - 1) * W;
            for (int x = 0; x < W; x++) {
                float p = srow0[x] + srow1[x];
                float t0 = row[x * 3] + g0 * p;
                float t1 = row[x * 3 + 1] + g1 * (srow1[x] - srow0[x]);
                float t2 = row[x * 3 + 2] + g2 * p;
                row[x * 3] = t0; row[x * 3 + 1] = t1; row[x * 3 + 2] = t2;
            }
        }
        for (int x = 0; x < n * 3; x++) {
            row[-1 - x] = row[2 - x];
            row[W * 3 + x] = row[W * 3 + x - 3];
        }
        for (int x = 0; x < W; x++) {
            g0 = g[0];
            double b1 = row[x * 3] * g0, b2 = 0, b3 = row[x * 3 + 1] * g0,
                   b4 = 0, b5 = row[x * 3 + 2] * g0, b6 = 0;
            for (int k = 1; k <= n; k++) {
                double tg = row[(x + k) * 3] + row[(x - k) * 3];
                g0 = g[k];
                b1 += tg * g0;
                b4 += tg * xxg[k];
                b2 += (row[(x + k) * 3] - row[(x - k) * 3]) * xg[k];
                b3 += (row[(x + k) * 3 + 1] + row[(x - k) * 3 + 1]) * g0;
                b6 += (row[(x + k) * 3 + 1] - row[(x - k) * 3 + 1]) * xg[k];
                b5 += (row[(x + k) * 3 + 2] + row[(x - k) * 3 + 2]) * g0;
            }
            drow[x * 5 + 1] = (float)(b2 * ig11);
            drow[x * 5] = (float)(b3 * ig11);
            drow[x * 5 + 3] = (float)(b1 * ig03 + b4 * ig33);
            drow[x * 5 + 2] = (float)(b1 * ig03 + b5 * ig33);
            drow[x * 5 + 4] = (float)(b6 * ig55);
        }
    }
    free(rowbuf);
}

/* FarnebackUpdateMatrices -- SURVEY.md App. A.4.  R*, M: HxWx5 interleaved; flow HxWx2. */
void ofc_ref_update_matrices(const float *R0_, const float *R1, const float *flow_, float *M_,
                             int W, int H, int y0, int y1_)
{
    enum { BORDER = 5 };
    static const float border[BORDER] = {0.14f, 0.14f, 0.4472f, 0.4472f, 0.4472f};
    size_t step1 = (size_t)W * 5;
    for (int y = y0; y < y1_; y++) {
        const float *flow = flow_ + (size_t)y * W * 2;
        const float *R0 = R0_ + (size_t)y * W * 5;
        float *M = M_ + (size_t)y * W * 5;
        for (int x = 0; x < W; x++) {
            float dx = flow[x * 2], dy = flow[x * 2 + 1];
            float fx = x + dx, fy = y + dy;
            int x1 = ifloor(fx), y1 = ifloor(fy);
            float r2, r3, r4, r5, r6;
            fx -= x1; fy -= y1;
            if ((unsigned)x1 < (unsigned)(W - 1) && (unsigned)y1 < (unsigned)(H - 1)) {
                const float *ptr = R1 + (size_t)y1 * step1 + (size_t)x1 * 5;
                float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy),
                      a10 = (1.f - fx) * fy, a11 = fx * fy;
                r2 = a00 * ptr[0] + a01 * ptr[5] + a10 * ptr[step1] + a11 * ptr[step1 + 5];
                r3 = a00 * ptr[1] + a01 * ptr[6] + a10 * ptr[step1 + 1] + a11 * ptr[step1 + 6];
                r4 = a00 * ptr[2] + a01 * ptr[7] + a10 * ptr[step1 + 2] + a11 * ptr[step1 + 7];
                r5 = a00 * ptr[3] + a01 * ptr[8] + a10 * ptr[step1 + 3] + a11 * ptr[step1 + 8];
                r6 = a00 * ptr[4] + a01 * ptr[9] + a10 * ptr[step1 + 4] + a11 * ptr[step1 + 9];
                r4 = (R0[x * 5 + 2] + r4) * 0.5f;
                r5 = (R0[x * 5 + 3] + r5) * 0.5f;
                r6 = (R0[x * 5 + 4] + r6) * 0.25f;
            } else {
                r2 = r3 = 0.f;
                r4 = R0[x * 5 + 2];
                r5 = R0[x * 5 + 3];
                r6 = R0[x * 5 + 4] * 0.5f;
            }
            r2 = (R0[x * 5] - r2) * 0.5f;
            r3 = (R0[x * 5 + 1] - r3) * 0.5f;
            r2 += r4 * dy + r6 * dx;
            r3 += r6 * dy + r5 * dx;
            if ((unsigned)(x - BORDER) >= (unsigned)(W - BORDER * 2) ||
                (unsigned)(y - BORDER) >= (unsigned)(H - BORDER * 2)) {
                float scale = (x < BORDER ? border[x] : 1.f) *
                              (x >= W - BORDER ? border[W - x - 1] : 1.f) *
                              (y < BORDER ? border[y] : 1.f) *
                              (y >= H - BORDER ? border[H - y - 1] : 1.f);
                r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
            }
            M[x * 5] = r4 * r4 + r6 * r6;
            M[x * 5 + 1] = (r4 + r5) * r6;
            M[x * 5 + 2] = r5 * r5 + r6 * r6;
            M[x * 5 + 3] = r4 * r2 + r6 * r3;
            M[x * 5 + 4] = r6 * r2 + r5 * r3;
        }
    }
}

/* FarnebackUpdateFlow_Blur -- SURVEY.md App. A.5: double running sums (vertical then
 * horizontal, replicate border), scale 1/block^2, +1e-3 regulariser, striped matrix update. */
void ofc_ref_update_flow_blur(const float *R0, const float *R1, float *flow_, float *matM,
                              int W, int H, int block_size, int update_mats)
{
    int m = block_size / 2;
    int y0 = 0, y1;
    int min_update_stripe = (1 << 10) / W > block_size ? (1 << 10) / W : block_size;
    double scale = 1. / (block_size * block_size);
    double *vbuf = (double *)malloc(sizeof(double) * (size_t)(W + m * 2 + 2) * 5);
    double *vsum = vbuf + (m + 1) * 5;
    const float *srow0 = matM;
    for (int x = 0; x < W * 5; x++) vsum[x] = srow0[x] * (m + 2);
    for (int y = 1; y < m; y++) {
        srow0 = matM + (size_t)(y < H - 1 ? y : H - 1) * W * 5;
        for (int x = 0; x < W * 5; x++) vsum[x] += srow0[x];
    }
    for (int y = 0; y < H; y++) {
        double g11, g12, g22, h1, h2;
        float *flow = flow_ + (size_t)y * W * 2;
        srow0 = matM + (size_t)(y - m - 1 > 0 ? y - m - 1 : 0) * W * 5;
        const float *srow1 = matM + (size_t)(y + m < H - 1 ? y + m : H - 1) * W * 5;
        for (int x = 0; x < W * 5; x++) vsum[x] += srow1[x] - srow0[x];
        for (int x = 0; x < (m + 1) * 5; x++) {
            vsum[-1 - x] = vsum[4 - x];
            vsum[W * 5 + x] = vsum[W * 5 + x - 5];
        }
        g11 = vsum[0] * (m + 2);
        g12 = vsum[1] * (m + 2);
        g22 = vsum[2] * (m + 2);
        h1 = vsum[3] * (m + 2);
        h2 = vsum[4] * (m + 2);
        for (int x = 1; x < m; x++) {
            g11 += vsum[x * 5];
            g12 += vsum[x * 5 + 1];
            g22 += vsum[x * 5 + 2];
            h1 += vsum[x * 5 + 3];
            h2 += vsum[x * 5 + 4];
        }
        for (int x = 0; x < W; x++) {
            g11 += vsum[(x + m) * 5] - vsum[(x - m) * 5 - 5];
            g12 += vsum[(x + m) * 5 + 1] - vsum[(x - m) * 5 - 4];
            g22 += vsum[(x + m) * 5 + 2] - vsum[(x - m) * 5 - 3];
            h1 += vsum[(x + m) * 5 + 3] - vsum[(x - m) * 5 - 2];
            h2 += vsum[(x + m) * 5 + 4] - vsum[(x - m) * 5 - 1];
            double g11_ = g11 * scale, g12_ = g12 * scale, g22_ = g22 * scale;
            double h1_ = h1 * scale, h2_ = h2 * scale;
            double idet = 1. / (g11_ * g22_ - g12_ * g12_ + 1e-3);
            flow[x * 2] = (float)((g11_ * h2_ - g12_ * h1_) * idet);
            flow[x * 2 + 1] = (float)((g22_ * h1_ - g12_ * h2_) * idet);
        }
        y1 = y == H - 1 ? H : y - block_size;
        if (update_mats && (y1 == H || y1 >= y0 + min_update_stripe)) {
            ofc_ref_update_matrices(R0, R1, flow_, matM, W, H, y0, y1);
            y0 = y1;
        }
    }
    free(vbuf);
}

/* pyramid driver -- SURVEY.md App. A.1 */
int ofc_ref_pyramid_levels(int W, int H, const ofc_ref_fb_params *p)
{
    const int min_size = 32;
    int k;
    double scale = 1;
    for (k = 0; k < p->levels; k++) {
        scale *= p->pyr_scale;
        if (W * scale < min_size || H * scale < min_size) break;
    }
    return k;
}

void ofc_ref_level_geometry(int W, int H, const ofc_ref_fb_params *p, int k,
                            int *w, int *h, int *ksize, double *sigma)
{
    double scale = 1;
    for (int i = 0; i < k; i++) scale *= p->pyr_scale;
    double sg = (1. / scale - 1) * 0.5;
    int sz = iround_even(sg * 5) | 1;
    if (sz < 3) sz = 3;
    *sigma = sg;
    *ksize = sz;
    *w = iround_even(W * scale);
    *h = iround_even(H * scale);
}

/* u8 -> f32, GaussianBlur at full resolution, resize to the level size */
void ofc_ref_level_image(const uint8_t *img, int W, int H, const ofc_ref_fb_params *p, int k,
                         float *I)
{
    int w, h, ksize;
    double sigma;
    ofc_ref_level_geometry(W, H, p, k, &w, &h, &ksize, &sigma);
    size_t n = (size_t)W * H;
    float *f = (float *)malloc(sizeof(float) * n);
    float *b = (float *)malloc(sizeof(float) * n);
    for (size_t i = 0; i < n; i++) f[i] = (float)img[i];
    ofc_ref_gaussian_blur(f, W, H, ksize, sigma, b);
    ofc_ref_resize_linear(b, W, H, 1, I, w, h);
    free(f); free(b);
}

int ofc_ref_farneback(const uint8_t *prev, const uint8_t *next, int W, int H,
                      const ofc_ref_fb_params *p, float *flow_out)
{
    if (p->flags != 0 || p->poly_n > 15 || p->winsize < 1) return -1;
    int levels = ofc_ref_pyramid_levels(W, H, p);
    const uint8_t *img[2] = {prev, next};
    float *prevFlow = 0;
    int pw = 0, ph = 0;
    for (int k = levels; k >= 0; k--) {
        int w, h, ksize;
        double sigma;
        ofc_ref_level_geometry(W, H, p, k, &w, &h, &ksize, &sigma);
        size_t np = (size_t)w * h;
        float *flow = (k > 0) ? (float *)malloc(sizeof(float) * np * 2) : flow_out;
        if (!prevFlow) {
            memset(flow, 0, sizeof(float) * np * 2);
        } else {
            ofc_ref_resize_linear(prevFlow, pw, ph, 2, flow, w, h);
            float mul = (float)(1. / p->pyr_scale);
            for (size_t i = 0; i < np * 2; i++) flow[i] *= mul;
        }
        float *R[2], *I = (float *)malloc(sizeof(float) * np);
        float *M = (float *)malloc(sizeof(float) * np * 5);
        for (int i = 0; i < 2; i++) {
            R[i] = (float *)malloc(sizeof(float) * np * 5);
            ofc_ref_level_image(img[i], W, H, p, k, I);
            ofc_ref_polyexp(I, w, h, p->poly_n, p->poly_sigma, R[i]);
        }
        ofc_ref_update_matrices(R[0], R[1], flow, M, w, h, 0, h);
        for (int i = 0; i < p->iterations; i++)
            ofc_ref_update_flow_blur(R[0], R[1], flow, M, w, h, p->winsize,
                                     i < p->iterations - 1);
        free(R[0]); free(R[1]); free(I); free(M);
        if (prevFlow) free(prevFlow);
        prevFlow = (k > 0) ? flow : 0;
        pw = w; ph = h;
    }
    return 0;
}
