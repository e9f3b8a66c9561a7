// flow_device.h -- device-side pieces of the Farneback iteration shared by flow_kernels.hip (what ships) and
// flow_experiments.hip (the opt-in round-2 experiments): the per-pixel update-matrices arithmetic and the column march's
// step height.  Included inside namespace-free translation units; everything lives in namespace ofc.
#pragma once
#include "lloyd_common.h"

#include <algorithm>
#include <climits>

namespace ofc {

constexpr int BS_ROWS = 4;      // rows per step of the column march (k_box_solve, k_flow_iter)

// ------------------------------------------------------------------------------------------------
// K4  update matrices: bilinear warp-sample of R1 at x+flow, G/h products, border attenuation.
// 68 B/px algorithmic (20 R0 + 20 R1 + 8 flow read, 20 M written); gather-coalesced because the
// flow is smooth.  Contraction off: bit-exact with the oracle.
// ------------------------------------------------------------------------------------------------
// the per-pixel arithmetic of FarnebackUpdateMatrices, split into a branch-free LOAD
// part (so that a caller can issue the gathers of several pixels back to back: a load inside a divergent
// branch cannot be hoisted by the compiler and would serialise full memory latencies) and a pure-ALU part
// with contraction off -> bit-exact with the oracle.
struct UmIn {
    float r0[5];        // R0 at the pixel
    float g[4][5];      // R1 at the 4 bilinear taps (garbage-but-valid when out of range)
    float fx, fy;
    bool inr;
};

__device__ __forceinline__ void um_load(const float *__restrict__ R0, const float *__restrict__ R1, size_t /*plane*/,
                                        int W, int H, int x, int y, float2 fl, UmIn &u)
{
#pragma clang fp contract(off)
    // 32-bit BYTE offsets from the (uniform) frame base: global_load with an SGPR base and a 32-bit VGPR offset
    // instead of a 64-bit address per lane.  The launchers check 20*W*H < 2^32.
    const unsigned idx = (unsigned)y * (unsigned)W + (unsigned)x;
    float fx = (float)x + fl.x, fy = (float)y + fl.y;
    const int x1 = (int)floorf(fx), y1 = (int)floorf(fy);
    u.fx = fx - (float)x1;
    u.fy = fy - (float)y1;
    u.inr = (unsigned)x1 < (unsigned)(W - 1) && (unsigned)y1 < (unsigned)(H - 1);
    // R is pixel-interleaved: this pixel's 5 coefficients of R0 are 20 contiguous bytes; the two upper and the two lower
    // bilinear taps of R1 are 40 contiguous bytes each.  dword-aligned vector loads (global_load_dwordx4/x2).
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
    const float *q = reinterpret_cast<const float *>(reinterpret_cast<const char *>(R0) + idx * 20u);
    const f4u q0 = *reinterpret_cast<const f4u *>(q);
    u.r0[0] = q0.x; u.r0[1] = q0.y; u.r0[2] = q0.z; u.r0[3] = q0.w; u.r0[4] = q[4];
    const unsigned o1 = u.inr ? ((unsigned)y1 * (unsigned)W + (unsigned)x1) * 20u : 0u;
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
        const float *pr = reinterpret_cast<const float *>(reinterpret_cast<const char *>(R1) +
                                                          (o1 + (unsigned)rr * (unsigned)W * 20u));
        const f4u a = *reinterpret_cast<const f4u *>(pr), b = *reinterpret_cast<const f4u *>(pr + 4);
        const f2u c2 = *reinterpret_cast<const f2u *>(pr + 8);
        u.g[2 * rr][0] = a.x; u.g[2 * rr][1] = a.y; u.g[2 * rr][2] = a.z; u.g[2 * rr][3] = a.w; u.g[2 * rr][4] = b.x;
        u.g[2 * rr + 1][0] = b.y; u.g[2 * rr + 1][1] = b.z; u.g[2 * rr + 1][2] = b.w;
        u.g[2 * rr + 1][3] = c2.x; u.g[2 * rr + 1][4] = c2.y;
    }
}

// the fractions and the in-range flag again, from the same operands with the same operations as um_load: a caller that keeps
// gathers in flight across other work need not hold these three registers per row meanwhile (the empty asm keeps the
// compiler from recognising the common subexpression and carrying the first evaluation along)
__device__ __forceinline__ void um_refrac(UmIn &u, int W, int H, int x, int y, float2 fl)
{
#pragma clang fp contract(off)
    float flx = fl.x, fly = fl.y;
    asm volatile("" : "+v"(flx), "+v"(fly));
    float fx = (float)x + flx, fy = (float)y + fly;
    const int x1 = (int)floorf(fx), y1 = (int)floorf(fy);
    u.fx = fx - (float)x1;
    u.fy = fy - (float)y1;
    u.inr = (unsigned)x1 < (unsigned)(W - 1) && (unsigned)y1 < (unsigned)(H - 1);
}

__device__ __forceinline__ void um_math(const UmIn &u, int W, int H, int x, int y, float2 fl, float (&m)[5])
{
#pragma clang fp contract(off)
    const float dx = fl.x, dy = fl.y, fx = u.fx, fy = u.fy;
    float r2, r3, r4, r5, r6;
    if (u.inr) {
        const float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy,
                    a11 = fx * fy;
        r2 = a00 * u.g[0][0] + a01 * u.g[1][0] + a10 * u.g[2][0] + a11 * u.g[3][0];
        r3 = a00 * u.g[0][1] + a01 * u.g[1][1] + a10 * u.g[2][1] + a11 * u.g[3][1];
        r4 = a00 * u.g[0][2] + a01 * u.g[1][2] + a10 * u.g[2][2] + a11 * u.g[3][2];
        r5 = a00 * u.g[0][3] + a01 * u.g[1][3] + a10 * u.g[2][3] + a11 * u.g[3][3];
        r6 = a00 * u.g[0][4] + a01 * u.g[1][4] + a10 * u.g[2][4] + a11 * u.g[3][4];
        r4 = (u.r0[2] + r4) * 0.5f;
        r5 = (u.r0[3] + r5) * 0.5f;
        r6 = (u.r0[4] + r6) * 0.25f;
    } else {
        r2 = r3 = 0.f;
        r4 = u.r0[2]; r5 = u.r0[3]; r6 = u.r0[4] * 0.5f;
    }
    r2 = (u.r0[0] - r2) * 0.5f;
    r3 = (u.r0[1] - r3) * 0.5f;
    r2 += r4 * dy + r6 * dx;
    r3 += r6 * dy + r5 * dx;
    constexpr int BORDER = 5;
    if ((unsigned)(x - BORDER) >= (unsigned)(W - BORDER * 2) ||
        (unsigned)(y - BORDER) >= (unsigned)(H - BORDER * 2)) {
        // border[] = {0.14, 0.14, 0.4472, 0.4472, 0.4472} (select form: no dynamically indexed array)
        auto bw = [](int i) { return i < 2 ? 0.14f : 0.4472f; };
        float scale = (x < BORDER ? bw(x) : 1.f) * (x >= W - BORDER ? bw(W - x - 1) : 1.f) *
                      (y < BORDER ? bw(y) : 1.f) * (y >= H - BORDER ? bw(H - y - 1) : 1.f);
        r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
    }
    m[0] = r4 * r4 + r6 * r6;
    m[1] = (r4 + r5) * r6;
    m[2] = r5 * r5 + r6 * r6;
    m[3] = r4 * r2 + r6 * r3;
    m[4] = r6 * r2 + r5 * r3;
}

__device__ __forceinline__ void update_matrices_px(const float *__restrict__ R0, const float *__restrict__ R1,
                                                   size_t plane, int W, int H, int x, int y, float2 fl,
                                                   float (&m)[5])
{
    UmIn u;
    um_load(R0, R1, plane, W, H, x, y, fl, u);
    um_math(u, W, H, x, y, fl, m);
}

}  // namespace ofc
