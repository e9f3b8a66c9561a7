// membench2.hip -- what this MI355X's HBM delivers for plain streams, by launch shape and cache policy.
// build: /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/membench2.hip -o tools/membench2
// (the guide quotes 6.29 TB/s for a float4 copy; round 1's tools/membench saw 4.7 with 8192 grid-stride blocks)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_copy(const v4f *__restrict__ a, v4f *__restrict__ b, size_t n)
{
    const size_t st = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * st < n; i += U * st) {
        v4f v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = NTL ? __builtin_nontemporal_load(a + i + u * st) : a[i + u * st];
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (NTS) __builtin_nontemporal_store(v[u], b + i + u * st); else b[i + u * st] = v[u];
        }
    }
    for (; i < n; i += st) b[i] = a[i];
}

template <int U, bool NTL>
__global__ __launch_bounds__(256) void k_read(const v4f *__restrict__ a, size_t n, float *out)
{
    const size_t st = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float s = 0;
    for (; i + (U - 1) * st < n; i += U * st) {
        v4f v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = NTL ? __builtin_nontemporal_load(a + i + u * st) : a[i + u * st];
#pragma unroll
        for (int u = 0; u < U; u++) s += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (s == 12345.678f) out[0] = s;
}

template <bool NTS>
__global__ __launch_bounds__(256) void k_write(v4f *__restrict__ b, size_t n)
{
    const size_t st = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += st) {
        v4f v = {(float)i, 1, 2, 3};
        if (NTS) __builtin_nontemporal_store(v, b + i); else b[i] = v;
    }
}

// chunked copy: each block owns a contiguous chunk (instead of grid-striding): DRAM page locality per block
template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_copy_chunk(const v4f *__restrict__ a, v4f *__restrict__ b, size_t n)
{
    const size_t per = (n + gridDim.x - 1) / gridDim.x;
    const size_t lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    for (size_t i = lo + threadIdx.x; i < hi; i += 256) {
        v4f v = NTL ? __builtin_nontemporal_load(a + i) : a[i];
        if (NTS) __builtin_nontemporal_store(v, b + i); else b[i] = v;
    }
}

// chunked read: each block owns a contiguous chunk, two float4 per lane per iteration with the next pair requested ahead
// (the Lloyd sweep's load structure)
template <bool CHUNK>
__global__ __launch_bounds__(256) void k_read_like_lloyd(const v4f *__restrict__ a, size_t n, float *out)
{
    const size_t n2 = n / 2;                                   // pairs of float4 ("quads" of (u,v) points)
    size_t q, qe, qs;
    if (CHUNK) {
        const size_t per = (n2 + gridDim.x - 1) / gridDim.x;
        q = blockIdx.x * per + threadIdx.x; qe = (blockIdx.x + 1) * per < n2 ? (blockIdx.x + 1) * per : n2; qs = 256;
    } else {
        q = (size_t)blockIdx.x * 256 + threadIdx.x; qe = n2; qs = (size_t)gridDim.x * 256;
    }
    float s = 0;
    v4f na = {0, 0, 0, 0}, nb = {0, 0, 0, 0};
    if (q < qe) { na = __builtin_nontemporal_load(a + 2 * q); nb = __builtin_nontemporal_load(a + 2 * q + 1); }
    for (; q < qe; q += qs) {
        const v4f x = na, y = nb;
        if (q + qs < qe) { na = __builtin_nontemporal_load(a + 2 * (q + qs)); nb = __builtin_nontemporal_load(a + 2 * (q + qs) + 1); }
        s += x.x + x.y + x.z + x.w + y.x + y.y + y.z + y.w;
    }
    if (s == 12345.678f) out[0] = s;
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
    const size_t bytes = (size_t)2 << 30;            // 2 GiB in, 2 GiB out: far beyond the 256 MB Infinity Cache
    const size_t n = bytes / 16;
    v4f *a, *b; float *o;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, 4));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    const int grids[] = {512, 1024, 2048, 4096, 8192, 16384};
    for (int g : grids) {
        float ms;
#define RUN(name, moved, ...) ms = timeit([&] { __VA_ARGS__; }, 10); printf("grid %5d %-28s %.3f ms  %.0f GB/s\n", g, name, ms, (moved) / ms / 1e6);
        RUN("copy u1", 2.0 * bytes, hipLaunchKernelGGL((k_copy<1, false, false>), dim3(g), dim3(256), 0, 0, a, b, n))
        RUN("copy u4", 2.0 * bytes, hipLaunchKernelGGL((k_copy<4, false, false>), dim3(g), dim3(256), 0, 0, a, b, n))
        RUN("copy u4 nt-load", 2.0 * bytes, hipLaunchKernelGGL((k_copy<4, true, false>), dim3(g), dim3(256), 0, 0, a, b, n))
        RUN("copy u4 nt-store", 2.0 * bytes, hipLaunchKernelGGL((k_copy<4, false, true>), dim3(g), dim3(256), 0, 0, a, b, n))
        RUN("copy u4 nt-both", 2.0 * bytes, hipLaunchKernelGGL((k_copy<4, true, true>), dim3(g), dim3(256), 0, 0, a, b, n))
        RUN("copy chunked", 2.0 * bytes, hipLaunchKernelGGL((k_copy_chunk<false, false>), dim3(g), dim3(256), 0, 0, a, b, n))
        RUN("copy chunked nt-both", 2.0 * bytes, hipLaunchKernelGGL((k_copy_chunk<true, true>), dim3(g), dim3(256), 0, 0, a, b, n))
        RUN("read u4", 1.0 * bytes, hipLaunchKernelGGL((k_read<4, false>), dim3(g), dim3(256), 0, 0, a, n, o))
        RUN("read u8 nt", 1.0 * bytes, hipLaunchKernelGGL((k_read<8, true>), dim3(g), dim3(256), 0, 0, a, n, o))
        RUN("read lloyd-like strided", 1.0 * bytes, hipLaunchKernelGGL((k_read_like_lloyd<false>), dim3(g), dim3(256), 0, 0, a, n, o))
        RUN("read lloyd-like chunked", 1.0 * bytes, hipLaunchKernelGGL((k_read_like_lloyd<true>), dim3(g), dim3(256), 0, 0, a, n, o))
        RUN("write", 1.0 * bytes, hipLaunchKernelGGL((k_write<false>), dim3(g), dim3(256), 0, 0, b, n))
        RUN("write nt", 1.0 * bytes, hipLaunchKernelGGL((k_write<true>), dim3(g), dim3(256), 0, 0, b, n))
    }
    // hipMemcpyAsync D2D as the runtime does it
    float ms = timeit([&] { CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); }, 10);
    printf("hipMemcpy D2D                      %.3f ms  %.0f GB/s\n", ms, 2.0 * bytes / ms / 1e6);
    return 0;
}
