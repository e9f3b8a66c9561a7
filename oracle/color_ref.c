/*
 * color_ref.c -- CPU oracle for the small 8-bit colour / visualisation routines either side of
 * the flow: cv2.cvtColor(BGR2GRAY / BGR2HSV / HSV2BGR), cv2.cartToPolar,
 * cv2.normalize(NORM_MINMAX), the uint8 truncations of computeOpticalFlowModule.py:25-33, the
 * 14x25 grid geometry of KmeanGrids.py:52-113 and preprocess_image (KmeanGrids.py:269-286).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle_ref.h).
 *
 * Pins: BGR2HSV is bit-exact on 6 300 + 1 872 recorded values of the reference
 * (tests/golden/kat_cells.npz); HSV2BGR truncation and the per-frame min/max normalisation are
 * pinned structurally by the recorded flow-visualisation PNGs.  BGR2GRAY's
 * coefficient set and cartToPolar's polynomial are restated from OpenCV 4.x; parity unpinned.
 */
#include "oracle_ref.h"
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>

/* C.1: OpenCV 4.x fixed point, 15 bit */
void ofc_ref_bgr2gray(const uint8_t *bgr, int64_t npix, uint8_t *gray)
{
    for (int64_t i = 0; i < npix; i++) {
        int b = bgr[i * 3], g = bgr[i * 3 + 1], r = bgr[i * 3 + 2];
        gray[i] = (uint8_t)((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15);
    }
}

/* C.5: 8-bit BGR2HSV, H in [0,180) */
void ofc_ref_bgr2hsv(const uint8_t *bgr, int64_t npix, uint8_t *hsv)
{
    static int sdiv[256], hdiv[256], init = 0;
    if (!init) {
        sdiv[0] = hdiv[0] = 0;
        for (int i = 1; i < 256; i++) {
            sdiv[i] = (int)rint((255 << 12) / (1. * i));
            hdiv[i] = (int)rint((180 << 12) / (6. * i));
        }
        init = 1;
    }
    for (int64_t i = 0; i < npix; i++) {
        int b = bgr[i * 3], g = bgr[i * 3 + 1], r = bgr[i * 3 + 2];
        int v = b > g ? b : g; if (r > v) v = r;
        int vmin = b < g ? b : g; if (r < vmin) vmin = r;
        int diff = v - vmin;
        int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
        int s = (diff * sdiv[v] + (1 << 11)) >> 12;
        int h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
        h = (h * hdiv[diff] + (1 << 11)) >> 12;
        h += h < 0 ? 180 : 0;
        hsv[i * 3] = (uint8_t)h;
        hsv[i * 3 + 1] = (uint8_t)s;
        hsv[i * 3 + 2] = (uint8_t)v;
    }
}

/* C.4: 8-bit HSV2BGR through the f32 sector formula; output byte = floor(x*255) as the
 * recorded PNGs show. */
void ofc_ref_hsv2bgr(const uint8_t *hsv, int64_t npix, uint8_t *bgr)
{
    static const int sector_data[6][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};
    const float hscale = 6.f / 180.f;
    for (int64_t i = 0; i < npix; i++) {
        float h = (float)hsv[i * 3], s = hsv[i * 3 + 1] * (1.f / 255.f), v = hsv[i * 3 + 2] * (1.f / 255.f);
        float b, g, r;
        if (s == 0) {
            b = g = r = v;
        } else {
            float tab[4];
            h *= hscale;
            if (h < 0) do h += 6; while (h < 0);
            else if (h >= 6) do h -= 6; while (h >= 6);
            int sector = (int)floorf(h);
            h -= sector;
            if ((unsigned)sector >= 6u) { sector = 0; h = 0.f; }
            tab[0] = v;
            tab[1] = v * (1.f - s);
            tab[2] = v * (1.f - s * h);
            tab[3] = v * (1.f - s * (1.f - h));
            b = tab[sector_data[sector][0]];
            g = tab[sector_data[sector][1]];
            r = tab[sector_data[sector][2]];
        }
        float fb = b * 255.f, fg = g * 255.f, fr = r * 255.f;
        bgr[i * 3] = (uint8_t)(fb < 0 ? 0 : fb > 255 ? 255 : (int)fb);
        bgr[i * 3 + 1] = (uint8_t)(fg < 0 ? 0 : fg > 255 ? 255 : (int)fg);
        bgr[i * 3 + 2] = (uint8_t)(fr < 0 ? 0 : fr > 255 ? 255 : (int)fr);
    }
}

/* C.2: magnitude + fastAtan2 polynomial, angle in radians */
static float atan_deg_f32(float y, float x)
{
    const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
    const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
    const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
    const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
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

void ofc_ref_cart_to_polar(const float *x, const float *y, int64_t n, float *mag, float *ang)
{
    const float scale = (float)(M_PI / 180);
    for (int64_t i = 0; i < n; i++) {
        mag[i] = sqrtf(x[i] * x[i] + y[i] * y[i]);
        ang[i] = atan_deg_f32(y[i], x[i]) * scale;
    }
}

/* computeOpticalFlowModule.py:25-33 (== computeOpticalFlow.py:104-120):
 * H = u8(angle*180/pi/2) (truncation), S = 255, V = u8(normalize(mag, 0, 255, MINMAX)) (truncation),
 * then HSV2BGR.  mean_mag = np.mean(magnitude) (computeOpticalFlow.py:114-117), here as the f64
 * mean of the f32 magnitudes rounded to f32. */
void ofc_ref_flow_to_bgr(const float *flow, int W, int H, uint8_t *bgr, float *mean_mag)
{
    int64_t n = (int64_t)W * H;
    if (n <= 0) return;
    float *mag = (float *)malloc(sizeof(float) * n), *ang = (float *)malloc(sizeof(float) * n);
    float *u = (float *)calloc(n, sizeof(float)), *v = (float *)calloc(n, sizeof(float));
    uint8_t *hsv = (uint8_t *)malloc((size_t)n * 3);
    for (int64_t i = 0; i < n; i++) { u[i] = flow[i * 2]; v[i] = flow[i * 2 + 1]; }
    ofc_ref_cart_to_polar(u, v, n, mag, ang);
    float mn = mag[0], mx = mag[0];
    double sum = 0;
    for (int64_t i = 0; i < n; i++) {
        if (mag[i] < mn) mn = mag[i];
        if (mag[i] > mx) mx = mag[i];
        sum += mag[i];
    }
    if (mean_mag) *mean_mag = (float)(sum / (double)n);
    double dscale = 255. * ((double)mx - (double)mn > DBL_EPSILON ? 1. / ((double)mx - (double)mn) : 0);
    double dshift = 0. - (double)mn * dscale;
    float a = (float)dscale, b = (float)dshift;
    const float pif = (float)M_PI;
    for (int64_t i = 0; i < n; i++) {
        float hf = ang[i] * 180.f / pif / 2.f;
        float vf = mag[i] * a + b;
        hsv[i * 3] = (uint8_t)(int)hf;
        hsv[i * 3 + 1] = 255;
        hsv[i * 3 + 2] = (uint8_t)(vf < 0 ? 0 : vf > 255 ? 255 : (int)vf);
    }
    ofc_ref_hsv2bgr(hsv, n, bgr);
    free(mag); free(ang); free(u); free(v); free(hsv);
}

/* KmeanGrids.py:52-113 (mean variant live in drawGridsAndOutputCSV*.py:84-99).  Cells are
 * visited row-major; cv2.rectangle (KmeanGrids.py:108) paints each visited cell's outline white
 * on the shared frame, so when cell (cy,cx) is averaged its row 0 is already white iff cy>=1 and
 * its column 0 iff cx>=1.  mean -> astype(uint8) truncates -> BGR2HSV. */
static int cell_pixel_is_white_at_mean(int cy, int cx, int ly, int lx)
{
    return (cy >= 1 && ly == 0) || (cx >= 1 && lx == 0);
}

void ofc_ref_grid_cell_means(const uint8_t *bgr, int W, int H, int rows, int cols,
                             uint8_t *mean_bgr, uint8_t *hsv)
{
    int xs = W / cols, ys = H / rows;
    for (int cy = 0; cy < rows; cy++)
        for (int cx = 0; cx < cols; cx++) {
            uint64_t s[3] = {0, 0, 0};
            for (int ly = 0; ly < ys; ly++)
                for (int lx = 0; lx < xs; lx++) {
                    const uint8_t *p = bgr + ((size_t)(cy * ys + ly) * W + cx * xs + lx) * 3;
                    int white = cell_pixel_is_white_at_mean(cy, cx, ly, lx);
                    for (int c = 0; c < 3; c++) s[c] += white ? 255 : p[c];
                }
            uint8_t *m = mean_bgr + (size_t)(cy * cols + cx) * 3;
            double cnt = (double)xs * ys;
            for (int c = 0; c < 3; c++) m[c] = (uint8_t)((double)s[c] / cnt);
        }
    ofc_ref_bgr2hsv(mean_bgr, (int64_t)rows * cols, hsv);
}

/* the cell as KmeanGrids.py:113 stores it and :385 later reads it: after every rectangle has
 * been drawn, i.e. with a white row 0 and column 0 */
void ofc_ref_extract_cell(const uint8_t *bgr, int W, int H, int rows, int cols, int cell,
                          uint8_t *cell_bgr)
{
    int xs = W / cols, ys = H / rows;
    int cy = cell / cols, cx = cell % cols;
    for (int ly = 0; ly < ys; ly++)
        for (int lx = 0; lx < xs; lx++) {
            const uint8_t *p = bgr + ((size_t)(cy * ys + ly) * W + cx * xs + lx) * 3;
            uint8_t *q = cell_bgr + ((size_t)ly * xs + lx) * 3;
            int white = (ly == 0) || (lx == 0);
            for (int c = 0; c < 3; c++) q[c] = white ? 255 : p[c];
        }
}

/* preprocess_image, KmeanGrids.py:269-286: per-channel <thresh -> 0, alpha = 255 where the
 * grey value is > 0, channels kept in their incoming order */
void ofc_ref_preprocess_rgba(const uint8_t *bgr, int64_t npix, int thresh, uint8_t *rgba)
{
    for (int64_t i = 0; i < npix; i++) {
        uint8_t c3[3];
        for (int c = 0; c < 3; c++) c3[c] = bgr[i * 3 + c] < thresh ? 0 : bgr[i * 3 + c];
        uint8_t gray;
        ofc_ref_bgr2gray(c3, 1, &gray);
        rgba[i * 4] = c3[0]; rgba[i * 4 + 1] = c3[1]; rgba[i * 4 + 2] = c3[2];
        rgba[i * 4 + 3] = gray > 0 ? 255 : 0;
    }
}
