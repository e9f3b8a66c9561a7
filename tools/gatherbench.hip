// gatherbench.hip -- what the vector-memory pipeline (TA/TCP/TD) of a CU delivers for the flow iteration's operand
// fetch, by layout of R.  Every lane fetches, for one pixel, R0 (5 floats at the pixel) and the 2x2 bilinear taps of R1
// (5 floats each) at a displaced position -- the loads of um_load, nothing else -- for
//   A  pixel-interleaved R [px][5] (20-B stride: dwordx4 loads at 4-B alignment; what the engine uses)
//   B  split R: RA [px] float4 (16-B aligned dwordx4) + RB [px] float (the fifth coefficient)
//   C  padded R [px][8] (32-B stride, two aligned dwordx4 per pixel, 60 % more bytes)
// build: /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/gatherbench.hip -o tools/gatherbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
typedef float f4a __attribute__((ext_vector_type(4)));

constexpr int W = 1920, H = 1080;

__device__ __forceinline__ void disp(int x, int y, int &x1, int &y1)
{
    x1 = min(max(x + 2 + ((x >> 6) & 1), 0), W - 2);      // smooth displacement, as a flow field gives
    y1 = min(max(y - 1, 0), H - 2);
}

// block = 256 threads <-> 256 columns, marches down ROWS rows (4 rows of loads in flight, as k_flow_iter)
// work > 0: after the loads of a 4-row step every lane runs `work` dependent FMAs per fetched value pair and the work-group
// crosses two barriers, as a step of k_flow_iter does (matrix arithmetic, exchange, horizontal pass): with the dynamic LDS
// request limiting a CU to two work-groups this reproduces the kernel's burst / compute alternation at 2 waves per SIMD
template <int LAYOUT>
__global__ __launch_bounds__(256) void k_gather(const float *__restrict__ R, const float *__restrict__ RB, float *out, int rows, int work, int stagger, int sh)
{
    extern __shared__ float dyn_lds[];
    // stagger > 0: half of the work-groups (bit `sh` of the linear id) start `stagger` x 640 clocks late, so that the chip's
    // work-groups do not all load and all compute at the same time (the convoy that forms when every work-group starts
    // together and HBM completes everybody's step at about the same time)
    {
        const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (stagger && ((lin >> sh) & 1))
            for (int i = 0; i < stagger; i++) __builtin_amdgcn_s_sleep(10);
    }
    const int x = min(blockIdx.x * 242 + threadIdx.x, W - 1);
    const int y0 = blockIdx.y * rows;
    const size_t plane = (size_t)W * H;
    const float *R0 = R + (size_t)blockIdx.z * plane * (LAYOUT == 0 ? 5 : LAYOUT == 1 ? 4 : 8);
    const float *R1 = R0 + plane * (LAYOUT == 0 ? 5 : LAYOUT == 1 ? 4 : 8);
    const float *B0 = RB + (size_t)blockIdx.z * plane, *B1 = B0 + plane;
    float acc = 0;
    for (int yb = y0; yb < min(y0 + rows, H); yb += 4) {
        float v[4][25];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int y = min(yb + r, H - 1);
            int x1, y1;
            disp(x, y, x1, y1);
            const unsigned i0 = (unsigned)y * W + x, i1 = (unsigned)y1 * W + x1;
            if (LAYOUT == 0) {
                const float *q = R0 + i0 * 5u;
                const f4u a = *reinterpret_cast<const f4u *>(q);
                v[r][0] = a.x; v[r][1] = a.y; v[r][2] = a.z; v[r][3] = a.w; v[r][4] = q[4];
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const float *p = R1 + (i1 + (unsigned)t * W) * 5u;
                    const f4u b = *reinterpret_cast<const f4u *>(p), c = *reinterpret_cast<const f4u *>(p + 4);
                    const f2u d = *reinterpret_cast<const f2u *>(p + 8);
                    float *o = &v[r][5 + 10 * t];
                    o[0] = b.x; o[1] = b.y; o[2] = b.z; o[3] = b.w; o[4] = c.x; o[5] = c.y; o[6] = c.z; o[7] = c.w; o[8] = d.x; o[9] = d.y;
                }
            } else if (LAYOUT == 1) {
                const f4a a = *reinterpret_cast<const f4a *>(R0 + i0 * 4u);
                v[r][0] = a.x; v[r][1] = a.y; v[r][2] = a.z; v[r][3] = a.w; v[r][4] = B0[i0];
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const unsigned i = i1 + (unsigned)t * W;
                    const f4a b = *reinterpret_cast<const f4a *>(R1 + i * 4u), c = *reinterpret_cast<const f4a *>(R1 + i * 4u + 4);
                    const f2u d = *reinterpret_cast<const f2u *>(B1 + i);
                    float *o = &v[r][5 + 10 * t];
                    o[0] = b.x; o[1] = b.y; o[2] = b.z; o[3] = b.w; o[4] = c.x; o[5] = c.y; o[6] = c.z; o[7] = c.w; o[8] = d.x; o[9] = d.y;
                }
            } else {
                const f4a a = *reinterpret_cast<const f4a *>(R0 + i0 * 8u);
                v[r][0] = a.x; v[r][1] = a.y; v[r][2] = a.z; v[r][3] = a.w; v[r][4] = R0[i0 * 8u + 4];
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const float *p = R1 + (i1 + (unsigned)t * W) * 8u;
                    const f4a b = *reinterpret_cast<const f4a *>(p), c = *reinterpret_cast<const f4a *>(p + 8);
                    float *o = &v[r][5 + 10 * t];
                    o[0] = b.x; o[1] = b.y; o[2] = b.z; o[3] = b.w; o[4] = p[4]; o[5] = c.x; o[6] = c.y; o[7] = c.z; o[8] = c.w; o[9] = p[12];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < 25; i++) acc += v[r][i];
        if (work) {
            __syncthreads();
            float a0 = acc, a1 = acc + 1.f, a2 = acc + 2.f, a3 = acc + 3.f;
            for (int i = 0; i < work; i++) {
                a0 = fmaf(a0, 1.0001f, 0.5f); a1 = fmaf(a1, 1.0001f, 0.5f);
                a2 = fmaf(a2, 1.0001f, 0.5f); a3 = fmaf(a3, 1.0001f, 0.5f);
            }
            acc = (a0 + a1) + (a2 + a3);
            if (acc == 777.f) dyn_lds[threadIdx.x] = acc;
            __syncthreads();
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}


// Ping-pong form: ONE 512-thread work-group per CU, its two halves (two 256-column tiles) in enforced anti-phase: while
// one half runs a step's arithmetic (between the same two barriers a step of k_flow_iter has), the other half issues the
// gathers of its next step and waits for them.  Every wave executes the same number of s_barrier per phase.  Same
// registers per thread as the 256-thread form, same LDS per CU -- only the phases of the two tiles are tied together.
__global__ __launch_bounds__(512) void k_pingpong(const float *__restrict__ R, float *out, int rows, int work)
{
    extern __shared__ float dyn_lds[];
    const int half = threadIdx.x >> 8, t = threadIdx.x & 255;
    const int x = min((blockIdx.x * 2 + half) * 242 + t, W - 1);
    const int y0 = blockIdx.y * rows;
    const size_t plane = (size_t)W * H;
    const float *R0 = R + (size_t)blockIdx.z * plane * 5;
    const float *R1 = R0 + plane * 5;
    float acc = 0;
    float v[4][25];
    auto load = [&](int yb) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int y = min(yb + r, H - 1);
            int x1, y1;
            disp(x, y, x1, y1);
            const unsigned i0 = (unsigned)y * W + x, i1 = (unsigned)y1 * W + x1;
            const float *q = R0 + i0 * 5u;
            const f4u a = *reinterpret_cast<const f4u *>(q);
            v[r][0] = a.x; v[r][1] = a.y; v[r][2] = a.z; v[r][3] = a.w; v[r][4] = q[4];
#pragma unroll
            for (int tt = 0; tt < 2; tt++) {
                const float *p = R1 + (i1 + (unsigned)tt * W) * 5u;
                const f4u b = *reinterpret_cast<const f4u *>(p), c = *reinterpret_cast<const f4u *>(p + 4);
                const f2u d = *reinterpret_cast<const f2u *>(p + 8);
                float *o = &v[r][5 + 10 * tt];
                o[0] = b.x; o[1] = b.y; o[2] = b.z; o[3] = b.w; o[4] = c.x; o[5] = c.y; o[6] = c.z; o[7] = c.w; o[8] = d.x; o[9] = d.y;
            }
        }
    };
    const int y_end = min(y0 + rows, H);
    const int nsteps = (y_end - y0 + 3) / 4;
    if (half == 0) load(y0);
    __syncthreads();
    for (int ph = 0; ph <= 2 * nsteps; ph++) {
        const int step = (ph - half) >> 1;                  // the step this half computes (even phases: half 0)
        if (((ph & 1) == half) && step >= 0 && step < nsteps) {
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int i = 0; i < 25; i++) acc += v[r][i];
            __builtin_amdgcn_s_barrier();
            float a0 = acc, a1 = acc + 1.f, a2 = acc + 2.f, a3 = acc + 3.f;
            for (int i = 0; i < work; i++) {
                a0 = fmaf(a0, 1.0001f, 0.5f); a1 = fmaf(a1, 1.0001f, 0.5f);
                a2 = fmaf(a2, 1.0001f, 0.5f); a3 = fmaf(a3, 1.0001f, 0.5f);
            }
            acc = (a0 + a1) + (a2 + a3);
            if (acc == 777.f) dyn_lds[threadIdx.x] = acc;
            __builtin_amdgcn_s_barrier();
        } else {
            const int nstep = (ph + 1 - half) >> 1;          // the step this half computes in the NEXT phase
            if (nstep >= 0 && nstep < nsteps && ((ph + 1) & 1) == half) load(y0 + 4 * nstep);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_s_barrier();
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <class F> float timeit(F f, int iters)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); f();
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; i++) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / iters;
}

int main(int argc, char **argv)
{
    const int lds = argc > 1 ? atoi(argv[1]) : 0;        // dynamic LDS bytes per work-group (occupancy limiter)
    const int work = argc > 2 ? atoi(argv[2]) : 0;       // dependent FMA quadruples per 4-row step
    const int stagger = argc > 3 ? atoi(argv[3]) : 0, sh = argc > 4 ? atoi(argv[4]) : 0;
    const int NP = 32;                       // pairs; NP + 1 frames of R
    const size_t plane = (size_t)W * H;
    float *R, *RB, *out;
    CK(hipMalloc(&R, sizeof(float) * 8 * plane * (NP + 1)));
    CK(hipMalloc(&RB, sizeof(float) * plane * (NP + 1)));
    CK(hipMalloc(&out, 4));
    CK(hipMemset(R, 0, sizeof(float) * 8 * plane * (NP + 1)));
    CK(hipMemset(RB, 0, sizeof(float) * plane * (NP + 1)));
    const int rows = 272;
    dim3 grid((W + 241) / 242, (H + rows - 1) / rows, NP);
    if (lds > 48 * 1024) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gather<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gather<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_gather<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    }
    printf("dynamic LDS %d B per work-group, %d FMA quadruples per step, stagger %d x 640 clk on bit %d\n", lds, work, stagger, sh);
    float ms;
    ms = timeit([&] { hipLaunchKernelGGL(k_gather<0>, grid, dim3(256), lds, 0, R, RB, out, rows, work, stagger, sh); }, 10);
    printf("A interleaved [px][5], unaligned x4   %.3f ms  (%.0f GB/s of 40 B/px unique)\n", ms, 40.0 * plane * NP / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(k_gather<1>, grid, dim3(256), lds, 0, R, RB, out, rows, work, stagger, sh); }, 10);
    printf("B split float4 + float, aligned x4    %.3f ms  (%.0f GB/s of 40 B/px unique)\n", ms, 40.0 * plane * NP / ms / 1e6);
    if (work) {
        const int lds2 = 90000;             // one 512-thread work-group per CU
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_pingpong), hipFuncAttributeMaxDynamicSharedMemorySize, lds2));
        dim3 grid2(((W + 241) / 242 + 1) / 2, (H + rows - 1) / rows, NP);
        ms = timeit([&] { hipLaunchKernelGGL(k_pingpong, grid2, dim3(512), lds2, 0, R, out, rows, work); }, 10);
        printf("P ping-pong halves (interleaved R), 1 WG/CU  %.3f ms  (%.0f GB/s of 40 B/px unique)\n", ms, 40.0 * plane * NP / ms / 1e6);
    }
    ms = timeit([&] { hipLaunchKernelGGL(k_gather<2>, grid, dim3(256), lds, 0, R, RB, out, rows, work, stagger, sh); }, 10);
    printf("C padded [px][8], aligned x4          %.3f ms  (%.0f GB/s of 64 B/px unique)\n", ms, 64.0 * plane * NP / ms / 1e6);
    return 0;
}
