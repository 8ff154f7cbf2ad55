// atomic_bench.hip - L2 fp32 atomic throughput on gfx950: scattered dwords vs 16 lanes per 64-B line
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(float *buf, int mode, int npix, int iters)
{
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned s = gid * 2654435761u + 12345u;
    for (int it = 0; it < iters; ++it) {
        s = s * 1664525u + 1013904223u;
        unsigned idx;
        if (mode == 0) idx = (s >> 4) % (unsigned)(npix * 16);                         // every lane its own random dword
        else if (mode == 1) { unsigned g = __shfl((int)s, (threadIdx.x & 63) & ~15); idx = ((g >> 4) % (unsigned)npix) * 16 + (threadIdx.x & 15); }  // 16 lanes share a 64-B pixel
        else { unsigned g = __shfl((int)s, (threadIdx.x & 63) & ~3); idx = ((g >> 4) % (unsigned)(npix * 4)) * 4 + (threadIdx.x & 3); }           // 4 lanes share 16 B
        unsafeAtomicAdd(buf + idx, 1.0f);
    }
}
int main()
{
    const int npix = 614400;   // 4 samples x 5 views x 240x128
    float *buf; hipMalloc(&buf, (size_t)npix * 64); hipMemset(buf, 0, (size_t)npix * 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode) {
        const int blocks = 4096, iters = 64;
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, buf, mode, npix, iters);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, buf, mode, npix, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double n = (double)blocks * 256 * iters;
        printf("mode %d: %.1f us, %.1f G lane-atomics/s\n", mode, ms * 1e3, n / ms / 1e6);
    }
    return 0;
}
