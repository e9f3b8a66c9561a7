// ofc_common.h -- shared host-side plumbing of libofc.so (error reporting, HIP checks, device
// buffers).  gfx950 only; no CUDA-compat paths.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ofc.h"

namespace ofc {

void set_error(const char *fmt, ...);
int ensure_device(int device);   // validates + hipSetDevice; returns OFC_OK / OFC_ENODEV

#define OFC_HIP(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            ofc::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,   \
                           __LINE__);                                                         \
            return _e == hipErrorOutOfMemory ? OFC_ENOMEM : OFC_EHIP;                         \
        }                                                                                     \
    } while (0)

#define OFC_TRY(expr)                    \
    do {                                 \
        int _rc = (expr);                \
        if (_rc != OFC_OK) return _rc;   \
    } while (0)

#define OFC_REQUIRE(cond, ...)           \
    do {                                 \
        if (!(cond)) {                   \
            ofc::set_error(__VA_ARGS__); \
            return OFC_EINVAL;           \
        }                                \
    } while (0)

// RAII device buffer (host-side bookkeeping only)
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    int alloc(size_t n)
    {
        release();
        if (n == 0) return OFC_OK;
        OFC_HIP(hipMalloc(&p, n));
        bytes = n;
        return OFC_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- Farneback level geometry (host) -- SURVEY.md App. A.1 ----
struct LevelGeom {
    int w, h, ksize;
    double sigma;
};
int pyramid_levels(int W, int H, const ofc_fb_params &p);
LevelGeom level_geometry(int W, int H, const ofc_fb_params &p, int k);
void gaussian_kernel(int n, double sigma, float *k);           // getGaussianKernel, CV_32F
struct PolyConsts {
    float g[8], xg[8], xxg[8];
    double ig11, ig03, ig33, ig55;
};
void polyexp_setup(int n, double sigma, PolyConsts &c);        // FarnebackPrepareGaussian

// ---- kernel launchers (flow_kernels.hip) : device pointers, asynchronous on `s` ----
// src: [nimg][H0][W0] u8 -> dst: [nimg][h][w] f32
int launch_level_image(const uint8_t *src, float *dst, int nimg, int W0, int H0,
                       const LevelGeom &g, hipStream_t s);
// level 0 fused into the polynomial expansion (frames u8 -> R), when the level is the plain 3x3 blur of the frame
bool polyexp_u8_ok(int W, int H, const LevelGeom &g);
int launch_polyexp_u8(const uint8_t *frames, float *R, int nimg, int W, int H, const PolyConsts &c, hipStream_t s);
// I: [nimg][H][W] f32 -> R: [nimg][H][W][5] f32 (pixel-interleaved)
int launch_polyexp(const float *I, float *R, int nimg, int W, int H, const PolyConsts &c,
                   int rows_per_block, hipStream_t s, bool bench_tag = false);
// R0 = R + pair*strideR, R1 = R0 + strideR (consecutive frames, interleaved); flow [npair][H][W][2]; M [npair][5][H][W] planar
int launch_update_matrices(const float *R0, const float *R1, size_t pair_stride_R,
                           const float *flow, float *M, int npair, int W, int H, hipStream_t s);
int launch_box_solve(const float *M, float *flow, int npair, int W, int H, int winsize,
                     int rows_per_block, hipStream_t s);
// src [npair][sh][sw][2] -> dst [npair][dh][dw][2], * mul
int launch_flow_resize(const float *src, float *dst, int npair, int sw, int sh, int dw, int dh,
                       float mul, hipStream_t s);
// fused iteration (update matrices + box mean + solve): R [npair+1][5][H][W]; flow_in != flow_out
// coarse != nullptr: the initial flow is resize(coarse [npair][sh][sw][2], (W,H)) * mul, sampled on the fly
int launch_flow_iter(const float *R, size_t frame_stride_R, const float *flow_in, float *flow_out,
                     int npair, int W, int H, int winsize, hipStream_t s, const float *coarse = nullptr,
                     int sw = 0, int sh = 0, float mul = 1.f, double *uv_sum = nullptr, double *uv_scratch = nullptr,
                     size_t uv_scratch_doubles = 0);
// upper bound of the iteration kernel's grid (work-groups) for a level of this size: sizes the uv_scratch above
int flow_iter_max_grid(int W, int H, int npair, int winsize);
// two fused iterations flow_in -> flow_out (winsize 15; flow_in at this level's size, != flow_out); rows_per_block 0 = auto
int launch_flow_iter2(const float *R, size_t frame_stride_R, const float *flow_in, float *flow_out, int npair, int W,
                      int H, int winsize, hipStream_t s, int rows_per_block = 0);
// one iteration, 3-waves-per-SIMD form (winsize 15, flow_in at this level's size)
int launch_flow_iter_w3(const float *R, size_t frame_stride_R, const float *flow_in, float *flow_out, int npair, int W,
                        int H, int winsize, hipStream_t s, int rows_per_block = 0);
// round-3 experiment (flow_experiments.hip): k_flow_iter with the next step's gathers issued across the exchange (winsize 15)
int launch_flow_iter_pipe(const float *R, size_t frame_stride_R, const float *flow_in, float *flow_out, int npair, int W,
                          int H, int rows_per_block, hipStream_t s, double *uv_sum, double *uv_scratch,
                          size_t uv_scratch_doubles);
int launch_flow_iter_stamped(const float *R, size_t frame_stride_R, const float *flow_in, float *flow_out, int npair, int W,
                             int H, hipStream_t s, unsigned long long *dbg, int *grid_out);
int polyexp_default_rows(int W, int H, int nimg);
int box_default_rows(int W, int H, int npair);

}  // namespace ofc
