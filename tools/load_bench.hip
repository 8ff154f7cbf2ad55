// load_bench.hip - dwordx4 gather throughput vs how many lanes share a contiguous span (gfx950):
//   mode 0: 4 lanes per random 64-B pixel (the unprojection kernel's tap pattern)
//   mode 1: 8 lanes per random 128-B pixel pair (x0,x0+1 in one instruction), pair start at any pixel
//   mode 2: 16 lanes per random 256-B span;  mode 3: fully coalesced 1-KiB per wave
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k(const float4 *__restrict__ tab, float4 *out, int mode, unsigned npix, int iters)
{
    const unsigned lane = threadIdx.x, wave = blockIdx.x;
    unsigned s = wave * 2654435761u + 977u;
    float4 acc = make_float4(0, 0, 0, 0);
    const int lpg = mode == 0 ? 4 : (mode == 1 ? 8 : (mode == 2 ? 16 : 64));      // lanes per group
    for (int it = 0; it < iters; ++it) {
        float4 v[16];
#pragma unroll
        for (int l = 0; l < 16; ++l) {
            s = s * 1664525u + 1013904223u;
            const unsigned grp = lane / lpg;
            unsigned h = (s ^ (grp * 0x9E3779B9u)) * 2246822519u;
            h ^= h >> 15;
            const unsigned pix = h % (npix - 16);                                  // group's first pixel (64-B units)
            v[l] = tab[(size_t)pix * 4 + (lane % lpg)];
        }
#pragma unroll
        for (int l = 0; l < 16; ++l) { acc.x += v[l].x; acc.y += v[l].y; acc.z += v[l].z; acc.w += v[l].w; }
    }
    if (acc.x == 123.f) out[0] = acc;
}
int main()
{
    const unsigned npix = 153600;     // one sample: 5 views x 240x128 pixels x 64 B = 9.8 MB
    float4 *tab, *out; hipMalloc(&tab, (size_t)npix * 64); hipMalloc(&out, 64); hipMemset(tab, 0, (size_t)npix * 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 4; ++mode) {
        const int waves = 8192, iters = 3;
        hipLaunchKernelGGL(k, dim3(waves), dim3(64), 0, 0, tab, out, mode, npix, iters);
        hipEventRecord(e0);
        for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k, dim3(waves), dim3(64), 0, 0, tab, out, mode, npix, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 100.0, loads = (double)waves * iters * 16;
        printf("mode %d: %.1f us, %.0f wave-loads, %.2f TB/s, %.1f CU-cycles per wave-load\n", mode, us, loads,
               loads * 1024 / us / 1e6, us * 2400.0 * 256 / loads);
    }
    return 0;
}
