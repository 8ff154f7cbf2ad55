// gather_bench2.hip - what does one 16-quad dwordx4 wave-load (4 lanes = one 64-B pixel) cost on gfx950, as a function
// of where its lines live?  Sets the per-load price list for the unprojection kernel's gather.
//   hipcc --offload-arch=gfx950 -O3 tools/gather_bench2.hip -o gpurun_out/gather_bench2 && gpurun_out/gather_bench2
// Patterns (per wave-load of 16 quads):
//   same      all quads read one pixel
//   l1        16 distinct pixels inside a 16 KB window private to the wave's workgroup (L1-resident after first touch)
//   l2        16 random pixels inside a 2 MB window (fits the XCD's L2, not the L1)
//   mall      16 random pixels inside the whole 39 MB table (bench workload: 20 planes of 240x128)
//   pairs_*   quads come as horizontally adjacent pixel pairs issued in two consecutive loads (t00 / t10 of the kernel)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

template <int LOADS>
__global__ __launch_bounds__(64) void gather(const float4 *__restrict__ tab, const int *__restrict__ idx, float4 *out, int iters)
{
    const int lane = threadIdx.x & 63, q = lane & 3, g = lane >> 2;
    const int wave = blockIdx.x;
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
    const int npix = 128 * 240 * 20;
    const int waves = 8192, iters = 3, LOADS = 16;
    std::vector<int> hidx((size_t)waves * iters * LOADS * 16);
    float4 *tab, *out; int *didx;
    hipMalloc(&tab, (size_t)npix * 64); hipMalloc(&out, 64); hipMalloc(&didx, hidx.size() * 4);
    hipMemset(tab, 0, (size_t)npix * 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[] = {"same", "l1", "l2", "mall", "pairs_l2", "pairs_mall", "pairs_mall_aligned", "l2_lines_only", "mall_stride2"};
    for (int mode = 0; mode < 9; ++mode) {
        srand(7);
        for (size_t wv = 0; wv < (size_t)waves; ++wv)
            for (int it = 0; it < iters; ++it)
                for (int l = 0; l < LOADS; ++l)
                    for (int g = 0; g < 16; ++g) {
                        int p = 0;
                        switch (mode) {
                        case 0: p = (int)(wv % 1000) * 64; break;
                        case 1: p = (int)((wv % 997) * 256 + (rand() % 256)); break;                 // 16 KB window
                        case 2: p = (int)((wv % 16) * 32768 + (rand() % 32768)); break;              // 2 MB window
                        case 3: p = rand() % npix; break;
                        case 7: p = (int)((wv % 16) * 32768 + (rand() % 16384) * 2); break;          // always the even half of a line
                        case 8: p = (rand() % (npix / 2)) * 2; break;
                        default: {
                            if (l & 1) { p = hidx[(((wv * iters + it) * LOADS) + l - 1) * 16 + g] + 1; break; }
                            int base = (mode == 4) ? (int)((wv % 16) * 32768 + (rand() % 32767)) : rand() % (npix - 1);
                            if (mode == 6) base &= ~1;
                            p = base;
                        }
                        }
                        hidx[((wv * iters + it) * LOADS + l) * 16 + g] = p;
                    }
        hipMemcpy(didx, hidx.data(), hidx.size() * 4, hipMemcpyHostToDevice);
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(gather<16>, dim3(waves), dim3(64), 0, 0, tab, didx, out, iters);
        hipEventRecord(e0);
        const int reps = 20;
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(gather<16>, dim3(waves), dim3(64), 0, 0, tab, didx, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / reps, nl = (double)waves * iters * LOADS;
        printf("%-20s %8.2f us per launch, %.0f wave-loads, %.2f TB/s gathered, %.1f ns/load/CU (%.1f clk @2.4GHz)\n", names[mode], us, nl,
               nl * 1024.0 / us / 1e6, us * 1e3 / (nl / 256.0), us * 2400.0 / (nl / 256.0));
    }
    return 0;
}
