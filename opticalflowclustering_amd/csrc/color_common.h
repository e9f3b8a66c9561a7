// color_common.h -- shared device helpers + launcher prototypes of the colour / grid kernels.
#pragma once
#include "ofc_common.h"

namespace ofc {

struct VisFrameStats { float a, b; };   // normalize(): out = mag * a + b
constexpr int VIS_BLOCKS = 256;         // work-groups per frame in the min/max pass

constexpr int SYNTH_WAVES = 24;
constexpr int SYNTH_POP = 5;
struct SynthParams {
    float fx[SYNTH_WAVES], fy[SYNTH_WAVES], a[SYNTH_WAVES], ph[SYNTH_WAVES];
    float vx[SYNTH_POP], vy[SYNTH_POP];
    float inv_norm;
};

// 8-bit BGR2HSV, H in [0,180): integer tables sdiv[i] = round((255<<12)/i), hdiv[i] = round((180<<12)/(6i))
// (bit-exact on the reference's recorded CSVs)
__device__ __forceinline__ void bgr2hsv_u8(unsigned b, unsigned g, unsigned r, unsigned &h, unsigned &s, unsigned &v)
{
    const int ib = (int)b, ig = (int)g, ir = (int)r;
    const int vv = max(ib, max(ig, ir)), vmin = min(ib, min(ig, ir));
    const int diff = vv - vmin;
    const int sdiv = vv ? (int)rint((255 << 12) / (1. * vv)) : 0;
    const int hdiv = diff ? (int)rint((180 << 12) / (6. * diff)) : 0;
    const int ss = (diff * sdiv + (1 << 11)) >> 12;
    int hh = (vv == ir) ? (ig - ib) : ((vv == ig) ? (ib - ir + 2 * diff) : (ir - ig + 4 * diff));
    hh = (hh * hdiv + (1 << 11)) >> 12;
    if (hh < 0) hh += 180;
    h = (unsigned)hh; s = (unsigned)ss; v = (unsigned)vv;
}

int launch_bgr2gray(const uint8_t *bgr, uint8_t *gray, int64_t npix, hipStream_t s);
int launch_bgr2hsv(const uint8_t *bgr, uint8_t *hsv, int64_t npix, hipStream_t s);
int launch_preprocess_rgba(const uint8_t *img, uint8_t *rgba, int64_t npix, int thresh, hipStream_t s);
int launch_flow_to_bgr(const float *flow, int W, int H, int nframes, uint8_t *bgr, float *mean_mag_dev,
                       double *partial, VisFrameStats *stats, hipStream_t s);
int launch_grid_cell_means(const uint8_t *bgr, int W, int H, int nframes, int rows, int cols,
                           uint8_t *mean_bgr, uint8_t *hsv, hipStream_t s);
int launch_grid_cell_mean_flow(const float *flow, int W, int H, int npair, int rows, int cols, float *out, hipStream_t s);
int launch_sliding_cosine(const double *a, int na, const double *b, int nwin, double *sims, int all_int, hipStream_t s);
int launch_synth_frames(uint8_t *frames, int W, int H, int nframes, int t0, const SynthParams &sp, hipStream_t s);

// batched small-problem Lloyd (lloyd_batched.hip)
struct BatchedArgs {
    // source A: packed rows
    const uint8_t *X;          // [total][4]
    const int64_t *offsets;    // [P+1]
    // source B: grid cells of BGR frames (X == nullptr)
    const uint8_t *bgr;
    int W, H, rows, cols, channel_order, thresh;
    // problem
    int k, max_iter, n_problems;
    double tol_rel;
    const double *init;        // [P][k][4] or nullptr -> deterministic maximin seeding on the device
    // outputs (any may be nullptr)
    double *centers;           // [P][k][4]
    int32_t *counts;           // [P][k]   bincount(predict(X))
    int32_t *labels;           // [total]  (source A only)
    int32_t *n_iter;           // [P]
    double *dom_center;        // [P][4]   rint(dominant centre)
    uint8_t *dom_hsv;          // [P][3]   BGR2HSV of its first three components
};
int launch_lloyd_batched(const BatchedArgs &a, int max_points, hipStream_t s);

}  // namespace ofc
