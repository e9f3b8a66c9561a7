// ofc_api.cpp -- C ABI of libofc.so, part 1: runtime plumbing and the Farneback flow engine.
// Declared in include/ofc.h (which cites the reference call each entry point replaces).
#include "color_common.h"
#include "lloyd_common.h"

#include <cfloat>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <memory>

namespace ofc {

static thread_local std::string g_err;

void set_error(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

int ensure_device(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no usable GPU (hipGetDeviceCount: %s, count %d); libofc has no CPU fallback",
                  hipGetErrorString(e), n);
        return OFC_ENODEV;
    }
    if (device < 0 || device >= n) {
        set_error("device %d out of range (have %d)", device, n);
        return OFC_ENODEV;
    }
    e = hipSetDevice(device);
    if (e != hipSuccess) {
        set_error("hipSetDevice(%d): %s", device, hipGetErrorString(e));
        return OFC_ENODEV;
    }
    return OFC_OK;
}

// ---- geometry & constants (host) ----
int pyramid_levels(int W, int H, const ofc_fb_params &p)
{
    const int min_size = 32;
    int k;
    double scale = 1;
    for (k = 0; k < p.levels; k++) {
        scale *= p.pyr_scale;
        if (W * scale < min_size || H * scale < min_size) break;
    }
    return k;
}

LevelGeom level_geometry(int W, int H, const ofc_fb_params &p, int k)
{
    double scale = 1;
    for (int i = 0; i < k; i++) scale *= p.pyr_scale;
    LevelGeom g;
    g.sigma = (1. / scale - 1) * 0.5;
    int sz = (int)std::nearbyint(g.sigma * 5) | 1;   // cvRound: half to even
    g.ksize = sz < 3 ? 3 : sz;
    g.w = (int)std::nearbyint(W * scale);
    g.h = (int)std::nearbyint(H * scale);
    return g;
}

void gaussian_kernel(int n, double sigma, float *k)
{
    static const float tab[4][7] = {{1.f},
                                    {0.25f, 0.5f, 0.25f},
                                    {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f},
                                    {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f}};
    const float *fixed = (n % 2 == 1 && n <= 7 && sigma <= 0) ? tab[n >> 1] : nullptr;
    double sigmaX = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    double scale2X = -0.5 / (sigmaX * sigmaX), sum = 0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        double t = fixed ? (double)fixed[i] : std::exp(scale2X * x * x);
        k[i] = (float)t;
        sum += k[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) k[i] = (float)(k[i] * sum);
}

void polyexp_setup(int n, double sigma, PolyConsts &c)
{
    if (sigma < FLT_EPSILON) sigma = n * 0.3;
    std::vector<float> gf(2 * n + 1);
    double s = 0;
    for (int x = -n; x <= n; x++) {
        gf[x + n] = (float)std::exp(-x * x / (2 * sigma * sigma));
        s += gf[x + n];
    }
    s = 1. / s;
    for (int x = -n; x <= n; x++) gf[x + n] = (float)(gf[x + n] * s);
    memset(&c, 0, sizeof(c));
    for (int x = 0; x <= n && x < 8; x++) {
        c.g[x] = gf[x + n];
        c.xg[x] = (float)(x * c.g[x]);
        c.xxg[x] = (float)(x * x * c.g[x]);
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
    double a = G00, b = G11, cc = G33, e = G55;
    double det = a * (cc * cc - e * e) - 2 * b * b * (cc - e);
    c.ig11 = 1. / G11;
    c.ig03 = -b * (cc - e) / det;
    c.ig33 = (a * cc - b * b) / det;
    c.ig55 = 1. / G55;
}

static int check_params(const ofc_fb_params &p, int W, int H)
{
    OFC_REQUIRE(W >= 16 && H >= 16 && W <= 16384 && H <= 16384, "frame size %dx%d out of range", W, H);
    OFC_REQUIRE(p.pyr_scale > 0 && p.pyr_scale < 1, "pyr_scale must be in (0,1)");
    OFC_REQUIRE(p.levels >= 0 && p.levels <= 16, "levels out of range");
    OFC_REQUIRE(p.iterations >= 1 && p.iterations <= 64, "iterations out of range");
    if (p.flags != 0) {
        set_error("flags=%d: only the box-filter variant (flags 0) the reference uses is implemented", p.flags);
        return OFC_EUNSUPPORTED;
    }
    if (p.poly_n != 5) {
        set_error("poly_n=%d: the polyexp kernel is specialised for poly_n=5 (the reference's value)", p.poly_n);
        return OFC_EUNSUPPORTED;
    }
    return OFC_OK;
}

}  // namespace ofc

using namespace ofc;

// =================================================================================================
// plumbing
// =================================================================================================
extern "C" {

int ofc_version(void) { return OFC_VERSION; }
const char *ofc_last_error(void) { return g_err.c_str(); }

int ofc_device_count(int *n)
{
    OFC_REQUIRE(n, "null pointer");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    *n = (e == hipSuccess) ? c : 0;
    return OFC_OK;
}

int ofc_malloc(int device, size_t bytes, void **dptr)
{
    OFC_REQUIRE(dptr, "null pointer");
    OFC_TRY(ensure_device(device));
    OFC_HIP(hipMalloc(dptr, bytes));
    return OFC_OK;
}

int ofc_free(int device, void *dptr)
{
    OFC_TRY(ensure_device(device));
    OFC_HIP(hipFree(dptr));
    return OFC_OK;
}

int ofc_memcpy_h2d(int device, void *dst, const void *src, size_t bytes)
{
    OFC_TRY(ensure_device(device));
    OFC_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return OFC_OK;
}

int ofc_memcpy_d2h(int device, void *dst, const void *src, size_t bytes)
{
    OFC_TRY(ensure_device(device));
    OFC_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return OFC_OK;
}

int ofc_memset(int device, void *dst, int value, size_t bytes)
{
    OFC_TRY(ensure_device(device));
    OFC_HIP(hipMemset(dst, value, bytes));
    return OFC_OK;
}

int ofc_device_sync(int device)
{
    OFC_TRY(ensure_device(device));
    OFC_HIP(hipDeviceSynchronize());
    return OFC_OK;
}

void ofc_fb_default_params(ofc_fb_params *p)
{
    if (!p) return;
    p->pyr_scale = 0.5; p->levels = 3; p->winsize = 15; p->iterations = 3;
    p->poly_n = 5; p->poly_sigma = 1.2; p->flags = 0;
}

}  // extern "C"

// =================================================================================================
// flow engine
// =================================================================================================
struct ofc_flow {
    int device = 0, W = 0, H = 0, max_batch = 0, levels = 0;
    ofc_fb_params prm{};
    PolyConsts pc{};
    std::vector<LevelGeom> geom;    // [0..levels]
    hipStream_t stream = nullptr;
    DevBuf I, R, M, flowA, flowB, flowC;   // scratch sized for level 0 and max_batch
    bool fused = true;              // update-matrices fused into the box/solve kernel (winsize <= 15)
    bool fuse_level0 = true;        // polyexp of level 0 reads the u8 frames (no f32 level-0 image)
    bool fuse2 = false;             // iterations 2+3 of a level in one launch (k_flow_iter2): measured SLOWER than two launches
                                    // on MI355X, kept as an opt-in experiment: OFC_FLOW_FUSE2=1 ...
    int fuse2_min_w = 0;            // ... or OFC_FLOW_FUSE2=<N > 1>: only at pyramid levels at least N pixels wide
    bool w3 = false;                // iterations 2.. of a level with the 3-waves-per-SIMD kernel (k_flow_iter_w3): OFC_FLOW_W3
    int w3_min_w = 0;
    DevBuf frames2, flow1;          // staging for the host-pointer entry points (batch of 1)
    DevBuf prev_gray;               // streaming state
    bool have_prev = false;
    DevBuf bgr_in, vis, vis_partial, vis_stats, mean_mag;   // ofc_flow_push_bgr
    DevBuf uv_scratch;              // per-work-group (sum u, sum v) records of the last level-0 iteration (ofc_flow_calc_frames_dev_stats)
    // A batch's launch sequence (4 pyramid levels x {level image, expansion, 3 iterations}: ~20 dependent launches, most of
    // them a few microseconds long on the coarse levels) is captured once per distinct argument set into a HIP graph and
    // replayed: a clip is processed with the same resident buffers step after step.  Opt-in (OFC_FLOW_GRAPH=1): measured on
    // MI355X it changes nothing -- 36.90 against 36.90 ms per 300-frame step, 5.33-5.41 ms per 38-pair shard either way
    // (two engines on two streams already cover each other's launch gaps; rocprof shows the GPU busy 99 % of a step).
    struct Captured {
        const uint8_t *frames = nullptr;
        int n_frames = 0;
        float *flow = nullptr;
        double *uv_sum = nullptr;
        int hits = 0;
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
    };
    std::vector<Captured> graphs;
    bool use_graph = false;
};

namespace ofc {

static int flow_run(ofc_flow *f, const uint8_t *frames_dev, int n_frames, float *flow_dev, double *uv_sum_dev = nullptr)
{
    const int npair = n_frames - 1;
    const int W = f->W, H = f->H;
    hipStream_t s = f->stream;
    float *I = f->I.as<float>(), *R = f->R.as<float>();
    const int iters = f->prm.iterations;
    float *prevFlow = nullptr;
    int pw = 0, ph = 0;
    for (int k = f->levels; k >= 0; k--) {
        const LevelGeom &g = f->geom[k];
        const size_t P = (size_t)g.w * g.h;
        // the level's final flow lands in `dst` (level 0: the caller's buffer); iterations ping-pong between
        // dst and tmp, so the initial flow goes to whichever makes the last iteration write dst
        float *dst = (k == 0) ? flow_dev : (((f->levels - k) & 1) ? f->flowB.as<float>() : f->flowA.as<float>());
        float *tmp = f->flowC.as<float>();
        if (k == 0 && f->fuse_level0 && polyexp_u8_ok(W, H, g)) {
            OFC_TRY(launch_polyexp_u8(frames_dev, R, n_frames, W, H, f->pc, s));
        } else {
            OFC_TRY(launch_level_image(frames_dev, I, n_frames, W, H, g, s));
            OFC_TRY(launch_polyexp(I, R, n_frames, g.w, g.h, f->pc, 0, s));
        }
        const size_t strideR = 5 * P;
        const float mul = (float)(1. / f->prm.pyr_scale);
        if (f->fused) {
            // launch plan: the first iteration alone (it reads the coarser level directly -- upsample fused -- or, at the
            // top of the pyramid, a zeroed buffer), then two iterations per launch while the level is wide enough for the
            // two-iteration kernel's 228-column tiles, single launches otherwise.  Launch l of L writes dst when (L-1-l)
            // is even, tmp otherwise, so that the last one lands in dst.
            const bool two = f->fuse2 && f->prm.winsize == 15 && g.w >= f->fuse2_min_w;
            int plan[64], L = 0;
            for (int i = 0; i < iters;) {
                const int n = (two && i > 0 && iters - i >= 2) ? 2 : 1;
                plan[L++] = n;
                i += n;
            }
            const float *cur = nullptr;
            for (int l = 0; l < L; l++) {
                float *nxt = ((L - 1 - l) & 1) ? tmp : dst;
                if (l == 0 && prevFlow) {
                    OFC_TRY(launch_flow_iter(R, strideR, nullptr, nxt, npair, g.w, g.h, f->prm.winsize, s, prevFlow, pw, ph, mul));
                } else {
                    if (l == 0) {
                        float *z = (nxt == dst) ? tmp : dst;
                        OFC_HIP(hipMemsetAsync(z, 0, sizeof(float) * 2 * P * npair, s));
                        cur = z;
                    }
                    if (uv_sum_dev && k == 0 && l == L - 1 && plan[l] == 1 && f->prm.winsize == 15 && !(f->w3 && g.w >= f->w3_min_w)) {
                        // the field's column sums ride in the epilogue of the iteration that writes it
                        const size_t need = (size_t)flow_iter_max_grid(g.w, g.h, npair, f->prm.winsize) * 2;
                        if (f->uv_scratch.bytes < need * sizeof(double)) OFC_TRY(f->uv_scratch.alloc(need * sizeof(double)));
                        OFC_TRY(launch_flow_iter(R, strideR, cur, nxt, npair, g.w, g.h, f->prm.winsize, s, nullptr, 0, 0, 1.f,
                                                 uv_sum_dev, f->uv_scratch.as<double>(), need));
                        uv_sum_dev = nullptr;
                    } else
                    if (plan[l] == 2) OFC_TRY(launch_flow_iter2(R, strideR, cur, nxt, npair, g.w, g.h, f->prm.winsize, s));
                    else if (f->w3 && f->prm.winsize == 15 && g.w >= f->w3_min_w)
                        OFC_TRY(launch_flow_iter_w3(R, strideR, cur, nxt, npair, g.w, g.h, f->prm.winsize, s));
                    else OFC_TRY(launch_flow_iter(R, strideR, cur, nxt, npair, g.w, g.h, f->prm.winsize, s));
                }
                cur = nxt;
            }
        } else {
            float *cur = dst;
            if (!prevFlow) {
                OFC_HIP(hipMemsetAsync(cur, 0, sizeof(float) * 2 * P * npair, s));
            } else {
                OFC_TRY(launch_flow_resize(prevFlow, cur, npair, pw, ph, g.w, g.h, mul, s));
            }
            float *M = f->M.as<float>();
            OFC_TRY(launch_update_matrices(R, R + strideR, strideR, cur, M, npair, g.w, g.h, s));
            for (int i = 0; i < iters; i++) {
                OFC_TRY(launch_box_solve(M, cur, npair, g.w, g.h, f->prm.winsize, 0, s));
                if (i < iters - 1)
                    OFC_TRY(launch_update_matrices(R, R + strideR, strideR, cur, M, npair, g.w, g.h, s));
            }
        }
        prevFlow = dst;
        pw = g.w; ph = g.h;
    }
    if (uv_sum_dev) {       // no epilogue carried them (one iteration per level, staged mode, another winsize, one of the
                            // experimental engines): one sweep over the finished field gives the same two sums
        const int64_t N = (int64_t)npair * W * H;
        const int nblocks = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv64(N / 4, 256), 1024));
        if (f->uv_scratch.bytes < sizeof(double) * 2 * nblocks) OFC_TRY(f->uv_scratch.alloc(sizeof(double) * 2 * nblocks));
        OFC_TRY(launch_lloyd_colstats(flow_dev, OFC_F32, N, 2, nullptr, 0, f->uv_scratch.as<double>(), nblocks, s));
        OFC_TRY(launch_reduce_records(f->uv_scratch.as<double>(), nblocks, 2, uv_sum_dev, s));
    }
    return OFC_OK;
}

// flow_run through a captured graph: the first call with an argument set runs directly, the second captures and
// instantiates, later ones replay.  (Two direct runs first: one-off calls -- tests, the streaming entry points with their
// rotating buffers -- never pay for an instantiation.)
static int flow_run_cached(ofc_flow *f, const uint8_t *frames_dev, int n_frames, float *flow_dev, double *uv_sum_dev)
{
    if (!f->use_graph) return flow_run(f, frames_dev, n_frames, flow_dev, uv_sum_dev);
    ofc_flow::Captured *c = nullptr;
    for (auto &g : f->graphs)
        if (g.frames == frames_dev && g.n_frames == n_frames && g.flow == flow_dev && g.uv_sum == uv_sum_dev) { c = &g; break; }
    if (!c) {
        if (f->graphs.size() >= 64) {               // a caller that never repeats itself: stop bookkeeping
            for (auto &g : f->graphs) {
                if (g.exec) (void)hipGraphExecDestroy(g.exec);
                if (g.graph) (void)hipGraphDestroy(g.graph);
            }
            f->graphs.clear();
        }
        f->graphs.emplace_back();
        c = &f->graphs.back();
        c->frames = frames_dev; c->n_frames = n_frames; c->flow = flow_dev; c->uv_sum = uv_sum_dev;
    }
    c->hits++;
    if (c->exec) {
        OFC_HIP(hipGraphLaunch(c->exec, f->stream));
        return OFC_OK;
    }
    if (c->hits < 2) return flow_run(f, frames_dev, n_frames, flow_dev, uv_sum_dev);   // (also sizes uv_scratch: no allocation inside a capture)
    OFC_HIP(hipStreamBeginCapture(f->stream, hipStreamCaptureModeThreadLocal));
    const int rc = flow_run(f, frames_dev, n_frames, flow_dev, uv_sum_dev);
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(f->stream, &g);
    if (rc != OFC_OK || e != hipSuccess || !g) {
        if (g) (void)hipGraphDestroy(g);
        f->use_graph = false;                       // whatever could not be captured: plain launches from now on
        (void)hipGetLastError();
        return flow_run(f, frames_dev, n_frames, flow_dev, uv_sum_dev);
    }
    hipGraphExec_t ex = nullptr;
    if (hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) != hipSuccess || !ex) {
        (void)hipGraphDestroy(g);
        f->use_graph = false;
        (void)hipGetLastError();
        return flow_run(f, frames_dev, n_frames, flow_dev, uv_sum_dev);
    }
    c->graph = g;
    c->exec = ex;
    OFC_HIP(hipGraphLaunch(c->exec, f->stream));
    return OFC_OK;
}

}  // namespace ofc

extern "C" {

int ofc_flow_create(int device, int W, int H, const ofc_fb_params *p, int max_batch, ofc_flow_t **out)
{
    OFC_REQUIRE(out, "null out pointer");
    *out = nullptr;
    ofc_fb_params prm;
    if (p) prm = *p; else ofc_fb_default_params(&prm);
    OFC_TRY(check_params(prm, W, H));
    OFC_REQUIRE(max_batch >= 1 && max_batch <= 4096, "max_batch out of range");
    OFC_REQUIRE((prm.winsize & 1) && prm.winsize >= 5, "winsize must be odd and >= 5");
    OFC_TRY(ensure_device(device));
    std::unique_ptr<ofc_flow> f(new ofc_flow);
    f->device = device; f->W = W; f->H = H; f->max_batch = max_batch; f->prm = prm;
    f->levels = pyramid_levels(W, H, prm);
    for (int k = 0; k <= f->levels; k++) f->geom.push_back(level_geometry(W, H, prm, k));
    polyexp_setup(prm.poly_n, prm.poly_sigma, f->pc);
    OFC_HIP(hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking));
    const size_t P0 = (size_t)W * H, nb = (size_t)max_batch;
    OFC_TRY(f->I.alloc(sizeof(float) * P0 * (nb + 1)));
    OFC_TRY(f->R.alloc(sizeof(float) * 5 * P0 * (nb + 1)));
    {
        const char *e = getenv("OFC_FLOW_STAGED");      // debugging aid: force the separate K4 / K5 kernels
        f->fused = prm.winsize <= 15 && !(e && e[0] == '1');
        f->fuse_level0 = !(e && e[0] == '1');            // the staged mode also keeps the separate level-0 image
        const char *e3 = getenv("OFC_FLOW_W3");           // 1: everywhere; N > 1: at levels >= N wide; 0: off
        if (e3 && e3[0]) {
            const int v = atoi(e3);
            f->w3 = v != 0;
            f->w3_min_w = v > 1 ? v : 0;
        }
        const char *eg = getenv("OFC_FLOW_GRAPH");
        f->use_graph = eg && eg[0] == '1';
        const char *e2 = getenv("OFC_FLOW_FUSE2");       // 1: two iterations per launch; N > 1: at levels >= N wide
        if (e2 && e2[0]) {
            const int v = atoi(e2);
            f->fuse2 = v != 0;
            if (v > 1) f->fuse2_min_w = v;
        }
    }
    if (f->fused) {
        OFC_TRY(f->flowC.alloc(sizeof(float) * 2 * P0 * nb));
    } else {
        OFC_TRY(f->M.alloc(sizeof(float) * 5 * P0 * nb));
    }
    const size_t P1 = f->levels >= 1 ? (size_t)f->geom[1].w * f->geom[1].h : 1;
    OFC_TRY(f->flowA.alloc(sizeof(float) * 2 * P1 * nb));
    OFC_TRY(f->flowB.alloc(sizeof(float) * 2 * P1 * nb));
    OFC_TRY(f->frames2.alloc(2 * P0));
    OFC_TRY(f->flow1.alloc(sizeof(float) * 2 * P0));
    OFC_TRY(f->prev_gray.alloc(P0));
    *out = f.release();
    return OFC_OK;
}

void ofc_flow_destroy(ofc_flow_t *f)
{
    if (!f) return;
    (void)hipSetDevice(f->device);
    if (f->stream) {
        (void)hipStreamSynchronize(f->stream);
        (void)hipStreamDestroy(f->stream);
    }
    for (auto &g : f->graphs) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    delete f;
}

int ofc_flow_calc_frames_dev(ofc_flow_t *f, const uint8_t *frames_dev, int n_frames, float *flow_dev)
{
    OFC_REQUIRE(f && frames_dev && flow_dev, "null pointer");
    OFC_REQUIRE(n_frames >= 2 && n_frames - 1 <= f->max_batch, "n_frames-1 = %d pairs not in [1, max_batch=%d]",
                n_frames - 1, f->max_batch);
    OFC_TRY(ensure_device(f->device));
    return flow_run_cached(f, frames_dev, n_frames, flow_dev, nullptr);
}

int ofc_flow_calc_frames_dev_stats(ofc_flow_t *f, const uint8_t *frames_dev, int n_frames, float *flow_dev, double *uv_sum_dev)
{
    OFC_REQUIRE(f && frames_dev && flow_dev && uv_sum_dev, "null pointer");
    OFC_REQUIRE(n_frames >= 2 && n_frames - 1 <= f->max_batch, "n_frames-1 = %d pairs not in [1, max_batch=%d]",
                n_frames - 1, f->max_batch);
    OFC_TRY(ensure_device(f->device));
    return flow_run_cached(f, frames_dev, n_frames, flow_dev, uv_sum_dev);
}

hipStream_t ofc_flow_stream_internal(ofc_flow_t *f) { return f->stream; }   // for stream_api.cpp (not exported in ofc.h)

int ofc_flow_sync(ofc_flow_t *f)
{
    OFC_REQUIRE(f, "null pointer");
    OFC_TRY(ensure_device(f->device));
    OFC_HIP(hipStreamSynchronize(f->stream));
    return OFC_OK;
}

int ofc_flow_calc(ofc_flow_t *f, const uint8_t *prev_gray, const uint8_t *next_gray, float *flow_out)
{
    OFC_REQUIRE(f && prev_gray && next_gray && flow_out, "null pointer");
    OFC_TRY(ensure_device(f->device));
    const size_t P0 = (size_t)f->W * f->H;
    uint8_t *fr = f->frames2.as<uint8_t>();
    OFC_HIP(hipMemcpyAsync(fr, prev_gray, P0, hipMemcpyHostToDevice, f->stream));
    OFC_HIP(hipMemcpyAsync(fr + P0, next_gray, P0, hipMemcpyHostToDevice, f->stream));
    OFC_TRY(flow_run(f, fr, 2, f->flow1.as<float>()));
    OFC_HIP(hipMemcpyAsync(flow_out, f->flow1.p, sizeof(float) * 2 * P0, hipMemcpyDeviceToHost, f->stream));
    OFC_HIP(hipStreamSynchronize(f->stream));
    return OFC_OK;
}

int ofc_flow_push_gray(ofc_flow_t *f, const uint8_t *gray, float *flow_out)
{
    OFC_REQUIRE(f && gray, "null pointer");
    OFC_TRY(ensure_device(f->device));
    const size_t P0 = (size_t)f->W * f->H;
    uint8_t *fr = f->frames2.as<uint8_t>();
    if (!f->have_prev) {
        OFC_HIP(hipMemcpyAsync(f->prev_gray.p, gray, P0, hipMemcpyHostToDevice, f->stream));
        OFC_HIP(hipStreamSynchronize(f->stream));
        f->have_prev = true;
        set_error("first frame pushed: no pair yet");
        return OFC_ENOTREADY;
    }
    OFC_REQUIRE(flow_out, "null flow_out");
    OFC_HIP(hipMemcpyAsync(fr, f->prev_gray.p, P0, hipMemcpyDeviceToDevice, f->stream));
    OFC_HIP(hipMemcpyAsync(fr + P0, gray, P0, hipMemcpyHostToDevice, f->stream));
    OFC_HIP(hipMemcpyAsync(f->prev_gray.p, fr + P0, P0, hipMemcpyDeviceToDevice, f->stream));
    OFC_TRY(flow_run(f, fr, 2, f->flow1.as<float>()));
    OFC_HIP(hipMemcpyAsync(flow_out, f->flow1.p, sizeof(float) * 2 * P0, hipMemcpyDeviceToHost, f->stream));
    OFC_HIP(hipStreamSynchronize(f->stream));
    return OFC_OK;
}

int ofc_flow_push_bgr(ofc_flow_t *f, const uint8_t *bgr, uint8_t *vis_out, float *mean_mag, float *flow_out)
{
    OFC_REQUIRE(f && bgr, "null pointer");
    OFC_TRY(ensure_device(f->device));
    const size_t P0 = (size_t)f->W * f->H;
    hipStream_t s = f->stream;
    if (!f->bgr_in.p) {
        OFC_TRY(f->bgr_in.alloc(P0 * 3 + 16));
        OFC_TRY(f->vis.alloc(P0 * 3 + 16));
        OFC_TRY(f->vis_partial.alloc(sizeof(double) * 3 * VIS_BLOCKS));
        OFC_TRY(f->vis_stats.alloc(sizeof(VisFrameStats)));
        OFC_TRY(f->mean_mag.alloc(sizeof(float)));
    }
    uint8_t *fr = f->frames2.as<uint8_t>();
    OFC_HIP(hipMemcpyAsync(f->bgr_in.p, bgr, P0 * 3, hipMemcpyHostToDevice, s));
    if (!f->have_prev) {
        OFC_TRY(launch_bgr2gray(f->bgr_in.as<uint8_t>(), f->prev_gray.as<uint8_t>(), (int64_t)P0, s));
        OFC_HIP(hipStreamSynchronize(s));
        f->have_prev = true;
        set_error("first frame pushed: no pair yet");
        return OFC_ENOTREADY;
    }
    OFC_HIP(hipMemcpyAsync(fr, f->prev_gray.p, P0, hipMemcpyDeviceToDevice, s));
    OFC_TRY(launch_bgr2gray(f->bgr_in.as<uint8_t>(), fr + P0, (int64_t)P0, s));
    OFC_HIP(hipMemcpyAsync(f->prev_gray.p, fr + P0, P0, hipMemcpyDeviceToDevice, s));
    OFC_TRY(flow_run(f, fr, 2, f->flow1.as<float>()));
    OFC_TRY(launch_flow_to_bgr(f->flow1.as<float>(), f->W, f->H, 1, f->vis.as<uint8_t>(), f->mean_mag.as<float>(),
                               f->vis_partial.as<double>(), f->vis_stats.as<VisFrameStats>(), s));
    if (vis_out) OFC_HIP(hipMemcpyAsync(vis_out, f->vis.p, P0 * 3, hipMemcpyDeviceToHost, s));
    if (mean_mag) OFC_HIP(hipMemcpyAsync(mean_mag, f->mean_mag.p, sizeof(float), hipMemcpyDeviceToHost, s));
    if (flow_out) OFC_HIP(hipMemcpyAsync(flow_out, f->flow1.p, sizeof(float) * 2 * P0, hipMemcpyDeviceToHost, s));
    OFC_HIP(hipStreamSynchronize(s));
    return OFC_OK;
}

int ofc_flow_last_vis_dev(ofc_flow_t *f, const uint8_t **vis_dev)
{
    OFC_REQUIRE(f && vis_dev, "null pointer");
    OFC_REQUIRE(f->vis.p, "no visualisation yet: call ofc_flow_push_bgr twice first");
    *vis_dev = f->vis.as<uint8_t>();
    return OFC_OK;
}

// =================================================================================================
// single stages (host buffers; parity-test hooks).  R is pixel-interleaved on the device as well; only M (planar in
// the staged fallback kernels) is converted on the host.
// =================================================================================================
static void interleave5(const float *planar, size_t P, float *inter)
{
    for (int c = 0; c < 5; c++)
        for (size_t i = 0; i < P; i++) inter[i * 5 + c] = planar[c * P + i];
}
static void planarize5(const float *inter, size_t P, float *planar)
{
    for (int c = 0; c < 5; c++)
        for (size_t i = 0; i < P; i++) planar[c * P + i] = inter[i * 5 + c];
}

int ofc_level_image(int device, const uint8_t *gray, int W, int H, const ofc_fb_params *p, int k,
                    float *out, int *w_out, int *h_out)
{
    OFC_REQUIRE(gray && out, "null pointer");
    ofc_fb_params prm;
    if (p) prm = *p; else ofc_fb_default_params(&prm);
    OFC_TRY(check_params(prm, W, H));
    OFC_REQUIRE(k >= 0 && k <= pyramid_levels(W, H, prm), "level %d out of range", k);
    OFC_TRY(ensure_device(device));
    LevelGeom g = level_geometry(W, H, prm, k);
    DevBuf src, dst;
    OFC_TRY(src.alloc((size_t)W * H));
    OFC_TRY(dst.alloc(sizeof(float) * g.w * g.h));
    OFC_HIP(hipMemcpy(src.p, gray, (size_t)W * H, hipMemcpyHostToDevice));
    OFC_TRY(launch_level_image(src.as<uint8_t>(), dst.as<float>(), 1, W, H, g, nullptr));
    OFC_HIP(hipMemcpy(out, dst.p, sizeof(float) * g.w * g.h, hipMemcpyDeviceToHost));
    if (w_out) *w_out = g.w;
    if (h_out) *h_out = g.h;
    return OFC_OK;
}

int ofc_polyexp(int device, const float *img, int W, int H, int n, double sigma, float *R5)
{
    OFC_REQUIRE(img && R5, "null pointer");
    OFC_REQUIRE(W >= 1 && H >= 1, "bad size");
    if (n != 5) { set_error("poly_n=%d unsupported (kernel specialised for 5)", n); return OFC_EUNSUPPORTED; }
    OFC_TRY(ensure_device(device));
    const size_t P = (size_t)W * H;
    PolyConsts pc;
    polyexp_setup(n, sigma, pc);
    DevBuf I, R;
    OFC_TRY(I.alloc(sizeof(float) * P));
    OFC_TRY(R.alloc(sizeof(float) * 5 * P));
    OFC_HIP(hipMemcpy(I.p, img, sizeof(float) * P, hipMemcpyHostToDevice));
    OFC_TRY(launch_polyexp(I.as<float>(), R.as<float>(), 1, W, H, pc, 0, nullptr));
    OFC_HIP(hipMemcpy(R5, R.p, sizeof(float) * 5 * P, hipMemcpyDeviceToHost));   // already pixel-interleaved
    return OFC_OK;
}

int ofc_polyexp_u8(int device, const uint8_t *gray, int W, int H, int n, double sigma, float *R5)
{
    OFC_REQUIRE(gray && R5 && W >= 1 && H >= 1, "bad arguments");
    if (n != 5) { set_error("poly_n %d unsupported (5 only)", n); return OFC_EUNSUPPORTED; }
    OFC_TRY(ensure_device(device));
    ofc_fb_params prm = {0.5, 0, 15, 3, 5, sigma, 0};
    const LevelGeom g = level_geometry(W, H, prm, 0);
    if (!polyexp_u8_ok(W, H, g)) { set_error("fused level-0 expansion needs W >= 4 and H >= 2"); return OFC_EUNSUPPORTED; }
    const size_t P = (size_t)W * H;
    PolyConsts pc;
    polyexp_setup(n, sigma, pc);
    DevBuf I, R;
    OFC_TRY(I.alloc(P));
    OFC_TRY(R.alloc(sizeof(float) * 5 * P));
    OFC_HIP(hipMemcpy(I.p, gray, P, hipMemcpyHostToDevice));
    OFC_TRY(launch_polyexp_u8(I.as<uint8_t>(), R.as<float>(), 1, W, H, pc, nullptr));
    OFC_HIP(hipMemcpy(R5, R.p, sizeof(float) * 5 * P, hipMemcpyDeviceToHost));
    return OFC_OK;
}

int ofc_update_matrices(int device, const float *R0, const float *R1, const float *flow, int W, int H,
                        float *M)
{
    OFC_REQUIRE(R0 && R1 && flow && M, "null pointer");
    OFC_REQUIRE(W >= 2 && H >= 2, "bad size");
    OFC_TRY(ensure_device(device));
    const size_t P = (size_t)W * H;
    std::vector<float> tmp(5 * P);
    DevBuf dR, dF, dM;
    OFC_TRY(dR.alloc(sizeof(float) * 10 * P));
    OFC_TRY(dF.alloc(sizeof(float) * 2 * P));
    OFC_TRY(dM.alloc(sizeof(float) * 5 * P));
    OFC_HIP(hipMemcpy(dR.p, R0, sizeof(float) * 5 * P, hipMemcpyHostToDevice));                       // R is pixel-interleaved
    OFC_HIP(hipMemcpy(dR.as<float>() + 5 * P, R1, sizeof(float) * 5 * P, hipMemcpyHostToDevice));
    OFC_HIP(hipMemcpy(dF.p, flow, sizeof(float) * 2 * P, hipMemcpyHostToDevice));
    OFC_TRY(launch_update_matrices(dR.as<float>(), dR.as<float>() + 5 * P, 5 * P, dF.as<float>(),
                                   dM.as<float>(), 1, W, H, nullptr));
    OFC_HIP(hipMemcpy(tmp.data(), dM.p, sizeof(float) * 5 * P, hipMemcpyDeviceToHost));
    interleave5(tmp.data(), P, M);
    return OFC_OK;
}

int ofc_box_solve(int device, const float *M, int W, int H, int winsize, float *flow)
{
    OFC_REQUIRE(M && flow, "null pointer");
    OFC_REQUIRE(W >= 1 && H >= 1, "bad size");
    OFC_TRY(ensure_device(device));
    const size_t P = (size_t)W * H;
    std::vector<float> tmp(5 * P);
    planarize5(M, P, tmp.data());
    DevBuf dM, dF;
    OFC_TRY(dM.alloc(sizeof(float) * 5 * P));
    OFC_TRY(dF.alloc(sizeof(float) * 2 * P));
    OFC_HIP(hipMemcpy(dM.p, tmp.data(), sizeof(float) * 5 * P, hipMemcpyHostToDevice));
    OFC_TRY(launch_box_solve(dM.as<float>(), dF.as<float>(), 1, W, H, winsize, 0, nullptr));
    OFC_HIP(hipMemcpy(flow, dF.p, sizeof(float) * 2 * P, hipMemcpyDeviceToHost));
    return OFC_OK;
}

int ofc_flow_resize(int device, const float *flow, int sw, int sh, int dw, int dh, float mul, float *out)
{
    OFC_REQUIRE(flow && out, "null pointer");
    OFC_REQUIRE(sw >= 1 && sh >= 1 && dw >= 1 && dh >= 1, "bad size");
    OFC_TRY(ensure_device(device));
    DevBuf s, d;
    OFC_TRY(s.alloc(sizeof(float) * 2 * sw * sh));
    OFC_TRY(d.alloc(sizeof(float) * 2 * dw * dh));
    OFC_HIP(hipMemcpy(s.p, flow, sizeof(float) * 2 * sw * sh, hipMemcpyHostToDevice));
    OFC_TRY(launch_flow_resize(s.as<float>(), d.as<float>(), 1, sw, sh, dw, dh, mul, nullptr));
    OFC_HIP(hipMemcpy(out, d.p, sizeof(float) * 2 * dw * dh, hipMemcpyDeviceToHost));
    return OFC_OK;
}

int ofc_flow_iterate(int device, const float *R0, const float *R1, const float *flow_in, int W, int H,
                     int winsize, int iters, int mode, int rows_per_block, float *flow_out)
{
    OFC_REQUIRE(R0 && R1 && flow_in && flow_out, "null pointer");
    OFC_REQUIRE(W >= 2 && H >= 2 && iters >= 1 && iters <= 64, "bad size / iteration count");
    OFC_TRY(ensure_device(device));
    const size_t P = (size_t)W * H;
    DevBuf dR, dA, dB;
    OFC_TRY(dR.alloc(sizeof(float) * 10 * P));
    OFC_TRY(dA.alloc(sizeof(float) * 2 * P));
    OFC_TRY(dB.alloc(sizeof(float) * 2 * P));
    OFC_HIP(hipMemcpy(dR.p, R0, sizeof(float) * 5 * P, hipMemcpyHostToDevice));
    OFC_HIP(hipMemcpy(dR.as<float>() + 5 * P, R1, sizeof(float) * 5 * P, hipMemcpyHostToDevice));
    OFC_HIP(hipMemcpy(dA.p, flow_in, sizeof(float) * 2 * P, hipMemcpyHostToDevice));
    float *cur = dA.as<float>(), *nxt = dB.as<float>();
    for (int i = 0; i < iters;) {
        const int n = (mode == 1 && iters - i >= 2) ? 2 : 1;
        if (n == 2) OFC_TRY(launch_flow_iter2(dR.as<float>(), 5 * P, cur, nxt, 1, W, H, winsize, nullptr, rows_per_block));
        else if (mode == 2) OFC_TRY(launch_flow_iter_w3(dR.as<float>(), 5 * P, cur, nxt, 1, W, H, winsize, nullptr, rows_per_block));
        else OFC_TRY(launch_flow_iter(dR.as<float>(), 5 * P, cur, nxt, 1, W, H, winsize, nullptr));
        std::swap(cur, nxt);
        i += n;
    }
    OFC_HIP(hipMemcpy(flow_out, cur, sizeof(float) * 2 * P, hipMemcpyDeviceToHost));
    return OFC_OK;
}

int ofc_bench_flow_iters(int device, int W, int H, int n_pairs, int reps, int mode, float *ms_out)
{
    OFC_REQUIRE(ms_out && n_pairs >= 1 && reps >= 1, "bad arguments");
    OFC_TRY(ensure_device(device));
    const size_t P = (size_t)W * H;
    const int nf = n_pairs + 1;
    PolyConsts pc;
    polyexp_setup(5, 1.2, pc);
    DevBuf fr, R, fa, fb;
    OFC_TRY(fr.alloc(P * nf));
    OFC_TRY(R.alloc(sizeof(float) * 5 * P * nf));
    OFC_TRY(fa.alloc(sizeof(float) * 2 * P * n_pairs));
    OFC_TRY(fb.alloc(sizeof(float) * 2 * P * n_pairs));
    hipStream_t s;
    OFC_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    OFC_TRY(ofc_synth_frames_dev(device, fr.as<uint8_t>(), W, H, nf, 0, 0));
    OFC_HIP(hipDeviceSynchronize());
    OFC_TRY(launch_polyexp_u8(fr.as<uint8_t>(), R.as<float>(), nf, W, H, pc, s));
    OFC_HIP(hipMemsetAsync(fb.p, 0, sizeof(float) * 2 * P * n_pairs, s));
    // a realistic flow_in: one iteration from zero flow
    OFC_TRY(launch_flow_iter(R.as<float>(), 5 * P, fb.as<float>(), fa.as<float>(), n_pairs, W, H, 15, s));
    hipEvent_t e0, e1;
    OFC_HIP(hipEventCreate(&e0));
    OFC_HIP(hipEventCreate(&e1));
    if (mode == 3) {            // diagnostic: the stamped build of k_flow_iter, phase shares printed to stdout
        int grid = 0;
        OFC_TRY(launch_flow_iter_stamped(R.as<float>(), 5 * P, fa.as<float>(), fb.as<float>(), n_pairs, W, H, s, nullptr, &grid));
        DevBuf dbg;
        OFC_TRY(dbg.alloc(sizeof(unsigned long long) * (size_t)grid * 4 * 8));
        OFC_HIP(hipMemsetAsync(dbg.p, 0, dbg.bytes, s));
        OFC_TRY(launch_flow_iter_stamped(R.as<float>(), 5 * P, fa.as<float>(), fb.as<float>(), n_pairs, W, H, s,
                                         dbg.as<unsigned long long>(), &grid));
        OFC_HIP(hipStreamSynchronize(s));
        std::vector<unsigned long long> h((size_t)grid * 4 * 8);
        OFC_HIP(hipMemcpy(h.data(), dbg.p, dbg.bytes, hipMemcpyDeviceToHost));
        double tot[8] = {0, 0, 0, 0, 0, 0, 0, 0}, all = 0;
        for (size_t i = 0; i < h.size(); i++) { tot[i & 7] += (double)h[i]; all += (double)h[i]; }
        static const char *name[8] = {"loop head / flow vectors", "issuing the gathers", "waiting for the operands", "first barrier",
                                      "ring + vertical sums + exchange writes", "second barrier", "horizontal sums + solve + stores",
                                      "matrix arithmetic"};
        printf("k_flow_iter<7,0> stamped build, %dx%d x %d pairs: share of the waves' in-loop cycles\n", W, H, n_pairs);
        for (int i = 0; i < 8; i++) printf("  %-42s %5.1f %%\n", name[i], 100.0 * tot[i] / all);
        fflush(stdout);
        *ms_out = 0.f;
        (void)hipStreamDestroy(s);
        return OFC_OK;
    }
    auto two = [&]() -> int {
        if (mode == 1) return launch_flow_iter2(R.as<float>(), 5 * P, fa.as<float>(), fb.as<float>(), n_pairs, W, H, 15, s);
        if (mode == 2) {
            OFC_TRY(launch_flow_iter_w3(R.as<float>(), 5 * P, fa.as<float>(), fb.as<float>(), n_pairs, W, H, 15, s));
            return launch_flow_iter_w3(R.as<float>(), 5 * P, fb.as<float>(), fa.as<float>(), n_pairs, W, H, 15, s);
        }
        OFC_TRY(launch_flow_iter(R.as<float>(), 5 * P, fa.as<float>(), fb.as<float>(), n_pairs, W, H, 15, s));
        // second iteration in place of the pair's scratch: fb -> fa would overwrite the input; write a third pass back to fb
        return launch_flow_iter(R.as<float>(), 5 * P, fb.as<float>(), fa.as<float>(), n_pairs, W, H, 15, s);
    };
    OFC_TRY(two());
    OFC_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < reps; i++) OFC_TRY(two());
    OFC_HIP(hipEventRecord(e1, s));
    OFC_HIP(hipEventSynchronize(e1));
    float ms = 0;
    OFC_HIP(hipEventElapsedTime(&ms, e0, e1));
    *ms_out = ms / reps;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipStreamDestroy(s);
    return OFC_OK;
}

// bench hook: polyexp over n_images distinct resident images, HIP events on the launch stream
int ofc_bench_polyexp(int device, int W, int H, int n_images, int iters, int rows_per_block,
                      float *ms_per_launch)
{
    OFC_REQUIRE(ms_per_launch && n_images >= 1 && iters >= 1, "bad arguments");
    OFC_TRY(ensure_device(device));
    const size_t P = (size_t)W * H;
    PolyConsts pc;
    polyexp_setup(5, 1.2, pc);
    DevBuf I, R;
    OFC_TRY(I.alloc(sizeof(float) * P * n_images));
    OFC_TRY(R.alloc(sizeof(float) * 5 * P * n_images));
    {   // non-trivial, image-dependent content (random-ish texture), generated on the host once
        std::vector<float> h(P);
        uint32_t st = 12345u;
        for (int im = 0; im < n_images; im++) {
            for (size_t i = 0; i < P; i++) {
                st = st * 1664525u + 1013904223u;
                h[i] = (float)(st >> 24);
            }
            OFC_HIP(hipMemcpy(I.as<float>() + (size_t)im * P, h.data(), sizeof(float) * P, hipMemcpyHostToDevice));
        }
    }
    hipStream_t s;
    OFC_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    OFC_HIP(hipEventCreate(&e0));
    OFC_HIP(hipEventCreate(&e1));
    for (int w = 0; w < 2; w++) OFC_TRY(launch_polyexp(I.as<float>(), R.as<float>(), n_images, W, H, pc, rows_per_block, s, true));
    OFC_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < iters; i++) OFC_TRY(launch_polyexp(I.as<float>(), R.as<float>(), n_images, W, H, pc, rows_per_block, s, true));
    OFC_HIP(hipEventRecord(e1, s));
    OFC_HIP(hipEventSynchronize(e1));
    float ms = 0;
    OFC_HIP(hipEventElapsedTime(&ms, e0, e1));
    *ms_per_launch = ms / iters;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipStreamDestroy(s);
    return OFC_OK;
}

}  // extern "C"
