// membench.hip -- what the HBM can do for polyexp's traffic shape (1 f32 plane read, 5 f32 planes written)
// build: /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/membench.hip -o tools/membench   (the binary is not tracked)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_copy(const float4 *a, float4 *b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
// read 1 plane, write 5 planes, each lane 4 consecutive px (float4), contiguous
__global__ void k_1to5(const float4 *a, float4 *b, size_t n4, size_t plane4)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = a[i];
        size_t img = i / plane4, r = i - img * plane4;
        float4 *o = b + img * 5 * plane4 + r;
        o[0] = v; v.x += 1; o[plane4] = v; v.y += 1; o[2 * plane4] = v; v.z += 1; o[3 * plane4] = v; v.w += 1; o[4 * plane4] = v;
    }
}
// write only (5 planes)
__global__ void k_w5(float4 *b, size_t n4, size_t plane4)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = make_float4(i, 1, 2, 3);
        size_t img = i / plane4, r = i - img * plane4;
        float4 *o = b + img * 5 * plane4 + r;
        o[0] = v; o[plane4] = v; o[2 * plane4] = v; o[3 * plane4] = v; o[4 * plane4] = v;
    }
}
// tiled like polyexp: block = 256 threads, wave <-> row, lane<60 <-> 4 px, tile 240 wide x 16 rows
__global__ void k_tiled(const float *a, float *b, int W, int H)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t plane = (size_t)W * H;
    const float *img = a + blockIdx.z * plane;
    float *out = b + blockIdx.z * 5 * plane;
    const int x0 = blockIdx.x * 240, y0 = blockIdx.y * 16;
    for (int rr = wave; rr < 16; rr += 4) {
        int y = y0 + rr;
        if (y >= H || lane >= 60) continue;
        size_t o = (size_t)y * W + x0 + 4 * lane;
        float4 v = *(const float4 *)(img + o);
        *(float4 *)(out + o) = v; *(float4 *)(out + plane + o) = v; *(float4 *)(out + 2 * plane + o) = v;
        *(float4 *)(out + 3 * plane + o) = v; *(float4 *)(out + 4 * plane + o) = v;
    }
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
int main()
{
    const int W = 1920, H = 1080, N = 64;
    const size_t plane = (size_t)W * H, n = plane * N;
    float *a, *b;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 20));
    CK(hipMemset(a, 1, n * 4)); CK(hipMemset(b, 0, n * 20));
    float ms;
    ms = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, (const float4 *)b, (float4 *)b + n * 5 / 8, n * 5 / 8); }, 10);
    printf("copy float4 (%.2f GB moved)      %.3f ms  %.0f GB/s\n", n * 5 / 8 * 16 * 2 / 1e9, ms, n * 5 / 8 * 16 * 2 / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(k_1to5, dim3(8192), dim3(256), 0, 0, (const float4 *)a, (float4 *)b, n / 4, plane / 4); }, 10);
    printf("1 read : 5 write contiguous        %.3f ms  %.0f GB/s\n", ms, n * 24 / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(k_w5, dim3(8192), dim3(256), 0, 0, (float4 *)b, n / 4, plane / 4); }, 10);
    printf("5 write only                       %.3f ms  %.0f GB/s\n", ms, n * 20 / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(k_tiled, dim3(8, 68, N), dim3(256), 0, 0, a, b, W, H); }, 10);
    printf("1:5 tiled 240x16 like polyexp      %.3f ms  %.0f GB/s\n", ms, n * 24 / ms / 1e6);
    return 0;
}
