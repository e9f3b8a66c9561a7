// color_api.cpp -- C ABI of libofc.so, part 3: visualisation, grid, batched per-cell k-means, synthetic
// frames.  See include/ofc.h for the reference call sites each entry point replaces.
#include "color_common.h"
#include "lloyd_common.h"

#include <algorithm>
#include <cmath>

using namespace ofc;

extern "C" {

int ofc_bgr2gray(int device, const uint8_t *bgr, int W, int H, uint8_t *gray)
{
    OFC_REQUIRE(bgr && gray && W >= 1 && H >= 1, "bad arguments");
    OFC_TRY(ensure_device(device));
    const size_t P = (size_t)W * H;
    DevBuf a, b;
    OFC_TRY(a.alloc(P * 3 + 16));
    OFC_TRY(b.alloc(P + 16));
    OFC_HIP(hipMemcpy(a.p, bgr, P * 3, hipMemcpyHostToDevice));
    OFC_TRY(launch_bgr2gray(a.as<uint8_t>(), b.as<uint8_t>(), (int64_t)P, nullptr));
    OFC_HIP(hipMemcpy(gray, b.p, P, hipMemcpyDeviceToHost));
    return OFC_OK;
}

int ofc_bgr2hsv(int device, const uint8_t *bgr, int64_t npix, uint8_t *hsv)
{
    OFC_REQUIRE(bgr && hsv && npix >= 1, "bad arguments");
    OFC_TRY(ensure_device(device));
    DevBuf a, b;
    OFC_TRY(a.alloc((size_t)npix * 3));
    OFC_TRY(b.alloc((size_t)npix * 3));
    OFC_HIP(hipMemcpy(a.p, bgr, (size_t)npix * 3, hipMemcpyHostToDevice));
    OFC_TRY(launch_bgr2hsv(a.as<uint8_t>(), b.as<uint8_t>(), npix, nullptr));
    OFC_HIP(hipMemcpy(hsv, b.p, (size_t)npix * 3, hipMemcpyDeviceToHost));
    return OFC_OK;
}

int ofc_preprocess_rgba(int device, const uint8_t *img3, int64_t npix, int thresh, uint8_t *rgba)
{
    OFC_REQUIRE(img3 && rgba && npix >= 1, "bad arguments");
    OFC_TRY(ensure_device(device));
    DevBuf a, b;
    OFC_TRY(a.alloc((size_t)npix * 3));
    OFC_TRY(b.alloc((size_t)npix * 4));
    OFC_HIP(hipMemcpy(a.p, img3, (size_t)npix * 3, hipMemcpyHostToDevice));
    OFC_TRY(launch_preprocess_rgba(a.as<uint8_t>(), b.as<uint8_t>(), npix, thresh, nullptr));
    OFC_HIP(hipMemcpy(rgba, b.p, (size_t)npix * 4, hipMemcpyDeviceToHost));
    return OFC_OK;
}

int ofc_flow_to_bgr_dev(int device, const float *flow_dev, int W, int H, int n_frames, uint8_t *bgr_dev,
                        float *mean_mag_dev)
{
    OFC_REQUIRE(flow_dev && bgr_dev && W >= 1 && H >= 1 && n_frames >= 1, "bad arguments");
    OFC_TRY(ensure_device(device));
    DevBuf partial, stats;
    OFC_TRY(partial.alloc(sizeof(double) * 3 * VIS_BLOCKS * (size_t)n_frames));
    OFC_TRY(stats.alloc(sizeof(VisFrameStats) * (size_t)n_frames));
    OFC_TRY(launch_flow_to_bgr(flow_dev, W, H, n_frames, bgr_dev, mean_mag_dev, partial.as<double>(),
                               stats.as<VisFrameStats>(), nullptr));
    OFC_HIP(hipStreamSynchronize(nullptr));
    return OFC_OK;
}

int ofc_flow_to_bgr(int device, const float *flow, int W, int H, uint8_t *bgr, float *mean_mag)
{
    OFC_REQUIRE(flow && bgr && W >= 1 && H >= 1, "bad arguments");
    OFC_TRY(ensure_device(device));
    const size_t P = (size_t)W * H;
    DevBuf f, o, m;
    OFC_TRY(f.alloc(P * 8));
    OFC_TRY(o.alloc(P * 3 + 16));
    OFC_TRY(m.alloc(sizeof(float)));
    OFC_HIP(hipMemcpy(f.p, flow, P * 8, hipMemcpyHostToDevice));
    OFC_TRY(ofc_flow_to_bgr_dev(device, f.as<float>(), W, H, 1, o.as<uint8_t>(), m.as<float>()));
    OFC_HIP(hipMemcpy(bgr, o.p, P * 3, hipMemcpyDeviceToHost));
    if (mean_mag) OFC_HIP(hipMemcpy(mean_mag, m.p, sizeof(float), hipMemcpyDeviceToHost));
    return OFC_OK;
}

int ofc_grid_cell_means(int device, const uint8_t *bgr, int W, int H, int rows, int cols, uint8_t *mean_bgr,
                        uint8_t *hsv)
{
    OFC_REQUIRE(bgr && mean_bgr && hsv, "null pointer");
    OFC_REQUIRE(rows >= 1 && cols >= 1 && W >= cols && H >= rows, "grid %dx%d does not fit %dx%d", rows, cols, W, H);
    OFC_TRY(ensure_device(device));
    const size_t P = (size_t)W * H, nc = (size_t)rows * cols;
    DevBuf f, a, b;
    OFC_TRY(f.alloc(P * 3));
    OFC_TRY(a.alloc(nc * 3));
    OFC_TRY(b.alloc(nc * 3));
    OFC_HIP(hipMemcpy(f.p, bgr, P * 3, hipMemcpyHostToDevice));
    OFC_TRY(launch_grid_cell_means(f.as<uint8_t>(), W, H, 1, rows, cols, a.as<uint8_t>(), b.as<uint8_t>(), nullptr));
    OFC_HIP(hipMemcpy(mean_bgr, a.p, nc * 3, hipMemcpyDeviceToHost));
    OFC_HIP(hipMemcpy(hsv, b.p, nc * 3, hipMemcpyDeviceToHost));
    return OFC_OK;
}

int ofc_kmeans_fit_batched(int device, const uint8_t *X, const int64_t *offsets, int n_problems, int d, int k,
                           const double *init, int max_iter, double tol_rel, double *centers, int32_t *counts,
                           int32_t *labels, int *n_iter)
{
    OFC_REQUIRE(X && offsets && n_problems >= 1 && max_iter >= 1, "bad arguments");
    if (d != 4) { set_error("the batched kernel is specialised for d=4 (RGBA rows); got d=%d", d); return OFC_EUNSUPPORTED; }
    OFC_TRY(ensure_device(device));
    const int64_t total = offsets[n_problems] - offsets[0];
    OFC_REQUIRE(offsets[0] == 0 && total >= 0, "offsets must start at 0 and be non-decreasing");
    int64_t maxn = 0;
    for (int p = 0; p < n_problems; p++) {
        OFC_REQUIRE(offsets[p + 1] >= offsets[p], "offsets must be non-decreasing");
        maxn = std::max(maxn, offsets[p + 1] - offsets[p]);
    }
    DevBuf dX, dOff, dInit, dCen, dCnt, dLab, dIt;
    OFC_TRY(dX.alloc((size_t)std::max<int64_t>(total, 1) * 4));
    OFC_TRY(dOff.alloc(sizeof(int64_t) * (n_problems + 1)));
    OFC_TRY(dCen.alloc(sizeof(double) * (size_t)n_problems * k * 4));
    OFC_TRY(dCnt.alloc(sizeof(int32_t) * (size_t)n_problems * k));
    OFC_TRY(dLab.alloc(sizeof(int32_t) * (size_t)std::max<int64_t>(total, 1)));
    OFC_TRY(dIt.alloc(sizeof(int32_t) * n_problems));
    OFC_HIP(hipMemcpy(dX.p, X, (size_t)total * 4, hipMemcpyHostToDevice));
    OFC_HIP(hipMemcpy(dOff.p, offsets, sizeof(int64_t) * (n_problems + 1), hipMemcpyHostToDevice));
    if (init) {
        OFC_TRY(dInit.alloc(sizeof(double) * (size_t)n_problems * k * 4));
        OFC_HIP(hipMemcpy(dInit.p, init, sizeof(double) * (size_t)n_problems * k * 4, hipMemcpyHostToDevice));
    }
    BatchedArgs a;
    memset(&a, 0, sizeof(a));
    a.X = dX.as<uint8_t>(); a.offsets = dOff.as<int64_t>();
    a.k = k; a.max_iter = max_iter; a.n_problems = n_problems; a.tol_rel = tol_rel;
    a.init = init ? dInit.as<double>() : nullptr;
    a.centers = dCen.as<double>(); a.counts = dCnt.as<int32_t>(); a.labels = dLab.as<int32_t>();
    a.n_iter = dIt.as<int32_t>();
    OFC_TRY(launch_lloyd_batched(a, (int)std::max<int64_t>(maxn, 1), nullptr));
    std::vector<int32_t> its(n_problems);
    OFC_HIP(hipMemcpy(its.data(), dIt.p, sizeof(int32_t) * n_problems, hipMemcpyDeviceToHost));
    for (int p = 0; p < n_problems; p++)
        if (its[p] < 0) {
            set_error("problem %d: n_samples=%lld should be >= n_clusters=%d.", p,
                      (long long)(offsets[p + 1] - offsets[p]), k);
            return OFC_EINVAL;
        }
    if (n_iter) memcpy(n_iter, its.data(), sizeof(int32_t) * n_problems);
    if (centers) OFC_HIP(hipMemcpy(centers, dCen.p, sizeof(double) * (size_t)n_problems * k * 4, hipMemcpyDeviceToHost));
    if (counts) OFC_HIP(hipMemcpy(counts, dCnt.p, sizeof(int32_t) * (size_t)n_problems * k, hipMemcpyDeviceToHost));
    if (labels && total > 0) OFC_HIP(hipMemcpy(labels, dLab.p, sizeof(int32_t) * (size_t)total, hipMemcpyDeviceToHost));
    return OFC_OK;
}

int ofc_grid_kmeans_dev(int device, const uint8_t *bgr_dev, int W, int H, int n_frames, int rows, int cols,
                        int k, const double *init, int max_iter, double tol_rel, int channel_order,
                        double *centers, uint8_t *hsv)
{
    OFC_REQUIRE(bgr_dev && centers && hsv, "null pointer");
    OFC_REQUIRE(rows >= 1 && cols >= 1 && W >= cols && H >= rows && max_iter >= 1 && n_frames >= 1, "bad arguments");
    OFC_TRY(ensure_device(device));
    const size_t nc = (size_t)rows * cols * n_frames;
    const int npts = (W / cols) * (H / rows);
    OFC_REQUIRE(npts >= k, "n_samples=%d should be >= n_clusters=%d.", npts, k);
    DevBuf dInit, dDom, dHsv;
    OFC_TRY(dDom.alloc(sizeof(double) * nc * 4));
    OFC_TRY(dHsv.alloc(nc * 3));
    if (init) {
        OFC_TRY(dInit.alloc(sizeof(double) * nc * k * 4));
        OFC_HIP(hipMemcpy(dInit.p, init, sizeof(double) * nc * k * 4, hipMemcpyHostToDevice));
    }
    BatchedArgs a;
    memset(&a, 0, sizeof(a));
    a.bgr = bgr_dev; a.W = W; a.H = H; a.rows = rows; a.cols = cols;
    a.channel_order = channel_order; a.thresh = 30;
    a.k = k; a.max_iter = max_iter; a.n_problems = (int)nc; a.tol_rel = tol_rel;
    a.init = init ? dInit.as<double>() : nullptr;
    a.dom_center = dDom.as<double>(); a.dom_hsv = dHsv.as<uint8_t>();
    OFC_TRY(launch_lloyd_batched(a, npts, nullptr));
    OFC_HIP(hipMemcpy(centers, dDom.p, sizeof(double) * nc * 4, hipMemcpyDeviceToHost));
    OFC_HIP(hipMemcpy(hsv, dHsv.p, nc * 3, hipMemcpyDeviceToHost));
    return OFC_OK;
}

int ofc_grid_kmeans(int device, const uint8_t *bgr, int W, int H, int rows, int cols, int k, const double *init,
                    int max_iter, double tol_rel, int channel_order, double *centers, uint8_t *hsv)
{
    OFC_REQUIRE(bgr && centers && hsv, "null pointer");
    OFC_REQUIRE(W >= 1 && H >= 1, "bad size");
    OFC_TRY(ensure_device(device));
    DevBuf f;
    OFC_TRY(f.alloc((size_t)W * H * 3));
    OFC_HIP(hipMemcpy(f.p, bgr, (size_t)W * H * 3, hipMemcpyHostToDevice));
    return ofc_grid_kmeans_dev(device, f.as<uint8_t>(), W, H, 1, rows, cols, k, init, max_iter, tol_rel,
                               channel_order, centers, hsv);
}

int ofc_sliding_cosine(int device, const double *small_v, int n_small, const double *large_v, int n_large, double *sims)
{
    OFC_REQUIRE(small_v && large_v && sims, "null pointer");
    OFC_REQUIRE(n_small >= 1 && n_large >= n_small, "need 1 <= n_small <= n_large (got %d, %d)", n_small, n_large);
    OFC_TRY(ensure_device(device));
    const int nwin = n_large - n_small + 1;
    int all_int = 1;
    for (int i = 0; i < n_small && all_int; i++) all_int = std::fabs(small_v[i]) < 2147483648.0 && small_v[i] == std::floor(small_v[i]);
    for (int i = 0; i < n_large && all_int; i++) all_int = std::fabs(large_v[i]) < 2147483648.0 && large_v[i] == std::floor(large_v[i]);
    DevBuf a, b, o;
    OFC_TRY(a.alloc(sizeof(double) * n_small));
    OFC_TRY(b.alloc(sizeof(double) * n_large));
    OFC_TRY(o.alloc(sizeof(double) * nwin));
    OFC_HIP(hipMemcpy(a.p, small_v, sizeof(double) * n_small, hipMemcpyHostToDevice));
    OFC_HIP(hipMemcpy(b.p, large_v, sizeof(double) * n_large, hipMemcpyHostToDevice));
    OFC_TRY(launch_sliding_cosine(a.as<double>(), n_small, b.as<double>(), nwin, o.as<double>(), all_int, nullptr));
    OFC_HIP(hipMemcpy(sims, o.p, sizeof(double) * nwin, hipMemcpyDeviceToHost));
    return OFC_OK;
}

int ofc_synth_frames_dev(int device, uint8_t *frames_dev, int W, int H, int n_frames, int t0, int seed)
{
    OFC_REQUIRE(frames_dev && W >= 1 && H >= 1 && n_frames >= 1, "bad arguments");
    OFC_TRY(ensure_device(device));
    SynthParams sp;
    uint32_t st = 0x9E3779B9u * (uint32_t)(seed + 1);
    auto rnd = [&st]() { st = st * 1664525u + 1013904223u; return (float)(st >> 8) * (1.0f / 16777216.0f); };
    float asum = 0;
    for (int i = 0; i < SYNTH_WAVES; i++) {
        sp.fx[i] = (0.01f + 0.11f * rnd()) * (rnd() < 0.5f ? -1.f : 1.f);
        sp.fy[i] = 0.01f + 0.11f * rnd();
        sp.a[i] = 0.3f + 0.7f * rnd();
        sp.ph[i] = 6.2831853f * rnd();
        asum += sp.a[i];
    }
    sp.inv_norm = 2.5f / asum;
    for (int j = 0; j < SYNTH_POP; j++) { sp.vx[j] = -4.f + 8.f * rnd(); sp.vy[j] = -4.f + 8.f * rnd(); }
    OFC_TRY(launch_synth_frames(frames_dev, W, H, n_frames, t0, sp, nullptr));
    OFC_HIP(hipStreamSynchronize(nullptr));
    return OFC_OK;
}

}  // extern "C"
