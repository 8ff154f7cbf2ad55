// mfma_bench.hip - sustained rate of v_mfma_f32_32x32x16_bf16 / v_mfma_f32_16x16x32_bf16 / v_mfma_f32_32x32x2_f32 on this chip:
// NACC independent accumulators per wave, W waves per SIMD, operands in registers, nothing else in the loop.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_bench.hip -o /tmp/mfma_bench && /tmp/mfma_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC, int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, unsigned seed)
{
    union { bf16x8 v; unsigned u[4]; } a, b;
    for (int i = 0; i < 4; ++i) { a.u[i] = 0x3f803f80u + threadIdx.x + i; b.u[i] = 0x3f803f80u + seed + i; }
    f32x16 acc[NACC];
    f32x4 acc4[NACC];
    for (int n = 0; n < NACC; ++n) { for (int v = 0; v < 16; ++v) acc[n][v] = 0.f; for (int v = 0; v < 4; ++v) acc4[n][v] = 0.f; }
    float fa = __uint_as_float(a.u[0]), fb = __uint_as_float(b.u[0]);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int n = 0; n < NACC; ++n) {
                if (KIND == 0) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc[n], 0, 0, 0);
                if (KIND == 1) acc4[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, acc4[n], 0, 0, 0);
                if (KIND == 2) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[n], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) { for (int v = 0; v < 16; ++v) s += acc[n][v]; for (int v = 0; v < 4; ++v) s += acc4[n][v]; }
    if (s == 123.456f) out[0] = s;
}

template <int NACC, int KIND>
void run(const char *name, int waves_per_simd, double flops_per_inst)
{
    float *out; hipMalloc(&out, 4);
    const int iters = 2000, blocks = 256 * waves_per_simd;       // 256 threads = 4 waves = one per SIMD of a CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, KIND>), dim3(blocks), dim3(256), 0, 0, out, 10, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, KIND>), dim3(blocks), dim3(256), 0, 0, out, iters, 1u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)blocks * 4 * iters * 4 * NACC;
    printf("%-28s acc=%d waves/SIMD=%d: %8.1f us  %7.1f TFLOP/s  %6.1f ns per instruction per SIMD\n", name, NACC, waves_per_simd,
           ms * 1e3, insts * flops_per_inst / (ms * 1e-3) / 1e12, ms * 1e6 / (insts / 1024.0));
    hipFree(out);
}

int main()
{
    run<4, 0>("32x32x16 bf16", 1, 32768); run<4, 0>("32x32x16 bf16", 2, 32768); run<2, 0>("32x32x16 bf16", 2, 32768); run<8, 0>("32x32x16 bf16", 1, 32768);
    run<4, 1>("16x16x32 bf16", 1, 16384); run<4, 1>("16x16x32 bf16", 2, 16384); run<8, 1>("16x16x32 bf16", 2, 16384);
    run<4, 2>("32x32x2 f32", 1, 4096); run<4, 2>("32x32x2 f32", 2, 4096);
    return 0;
}
