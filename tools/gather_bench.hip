// gather_bench.hip - microbenchmark: how fast can gfx950 gather random 64-B pixels (one per 4-lane group,
// 16 B per lane) from an N-MB table?  Sets the ceiling for the unprojection kernel's gather phase.
//   hipcc --offload-arch=gfx950 -O3 tools/gather_bench.hip -o gpurun_out/gather_bench && gpurun_out/gather_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int LOADS>
__global__ __launch_bounds__(256) void gather(const float4 *__restrict__ tab, const int *__restrict__ idx, float4 *out,
                                               int iters, int npix)
{
    const int lane = threadIdx.x & 63, q = lane & 3, g = lane >> 2;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    float4 acc = make_float4(0, 0, 0, 0);
    const int *ip = idx + (size_t)wave * iters * LOADS * 16;
    for (int it = 0; it < iters; ++it) {
        float4 v[LOADS];
#pragma unroll
        for (int l = 0; l < LOADS; ++l) {
            const int p = ip[(it * LOADS + l) * 16 + g];
            v[l] = tab[(size_t)p * 4 + q];
        }
#pragma unroll
        for (int l = 0; l < LOADS; ++l) { acc.x += v[l].x; acc.y += v[l].y; acc.z += v[l].z; acc.w += v[l].w; }
    }
    if (acc.x == 123.f) out[0] = acc;
}

int main()
{
    const int h = 128, w = 240, planes = 20;          // 5 views x 4 samples, 64-B pixels = 39 MB
    const int npix = h * w * planes;
    const int waves = 8000, iters = 3, LOADS = 16;     // ~ the coarse B=4 launch: 8000 waves x 48 loads
    std::vector<int> hidx((size_t)waves * iters * LOADS * 16);
    srand(1);
    for (int mode = 0; mode < 3; ++mode) {
        // mode 0: fully random pixels; 1: 2x2 footprints (pairs adjacent in x, rows adjacent) at random places
        // 2: footprints on a 4-px lattice walk (like neighbouring voxels)
        for (size_t wv = 0; wv < (size_t)waves; ++wv)
            for (int it = 0; it < iters; ++it)
                for (int l = 0; l < LOADS; l += 4)
                    for (int g = 0; g < 16; ++g) {
                        int pl = rand() % planes, y = rand() % (h - 1), x = rand() % (w - 1);
                        if (mode == 2) { pl = (int)(wv % planes); y = (int)((wv * 7 + g * 4 + it * 3) % (h - 1)); x = (int)((wv * 13 + l * 4 + g) % (w - 1)); }
                        const int base = (pl * h + y) * w + x;
                        const size_t o = ((wv * iters + it) * LOADS + l) * 16 + g;
                        if (mode == 0) { for (int k = 0; k < 4; ++k) hidx[o + 16 * k] = rand() % npix; }
                        else { hidx[o] = base; hidx[o + 16] = base + 1; hidx[o + 32] = base + w; hidx[o + 48] = base + w + 1; }
                    }
        float4 *tab, *out; int *didx;
        hipMalloc(&tab, (size_t)npix * 64); hipMalloc(&out, 64); hipMalloc(&didx, hidx.size() * 4);
        hipMemset(tab, 0, (size_t)npix * 64);
        hipMemcpy(didx, hidx.data(), hidx.size() * 4, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int blk : {64, 256}) {
            const int blocks = waves * 64 / blk;
            for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(gather<16>, dim3(blocks), dim3(blk), 0, 0, tab, didx, out, iters, npix);
            hipEventRecord(e0);
            const int reps = 20;
            for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(gather<16>, dim3(blocks), dim3(blk), 0, 0, tab, didx, out, iters, npix);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double us = ms * 1e3 / reps, bytes = (double)waves * iters * LOADS * 1024.0;
            printf("mode %d block %3d: %.2f us per launch, %.0f wave-loads, %.1f MB gathered, %.2f TB/s, %.1f cycles/load/CU\n",
                   mode, blk, us, (double)waves * iters * LOADS, bytes / 1e6, bytes / us / 1e6,
                   us * 2400.0 / ((double)waves * iters * LOADS / 256.0));
        }
        hipFree(tab); hipFree(out); hipFree(didx);
    }
    return 0;
}
