// readbench.hip -- read bandwidth as a function of the working-set size (does the 256 MB Infinity Cache serve re-reads
// faster than HBM?): each launch sums a buffer of S bytes; launches repeat back to back over the same buffer.
// build: /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/readbench.hip -o tools/readbench   (the binary is not tracked)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_sum(const float4 *a, size_t n, float *out)
{
    float s = 0;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, st = (size_t)gridDim.x * 256;
    for (; i + st < n; i += 2 * st) {
        const float4 u = a[i], v = a[i + st];
        s += u.x + u.y + u.z + u.w + v.x + v.y + v.z + v.w;
    }
    for (; i < n; i += st) { const float4 u = a[i]; s += u.x + u.y + u.z + u.w; }
    if (s == 12345.678f) out[0] = s;
}

int main()
{
    const size_t maxb = (size_t)4 << 30;
    float4 *a; float *o;
    CK(hipMalloc(&a, maxb)); CK(hipMalloc(&o, 4));
    CK(hipMemset(a, 0, maxb));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t sizes_mb[] = {16, 32, 64, 128, 192, 256, 384, 512, 1024, 4096};
    for (size_t mb : sizes_mb) {
        const size_t n = (mb << 20) / 16;
        const int reps = (int)(16384 / mb) + 4;
        for (int w = 0; w < 3; w++) hipLaunchKernelGGL(k_sum, dim3(2048), dim3(256), 0, 0, a, n, o);
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_sum, dim3(2048), dim3(256), 0, 0, a, n, o);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("working set %5zu MB: %.0f GB/s\n", mb, (double)(mb << 20) * reps / ms / 1e6);
    }
    return 0;
}
