// mfma_pk_hazard.hip - minimal form of the round-4 GPU-sharing finding (profiles/r04_gpu_sharing_finding.md):
// do waves that execute matrix instructions disturb PACKED fp32 VALU instructions (v_pk_fma_f32) of OTHER waves on the same CU?
//
// Stream A loops an "aggressor" kernel that does nothing but matrix instructions on registers (three kinds: 16x16x32 bf16,
// 32x32x16 bf16, 32x32x2 f32); stream B runs a "victim" kernel - a register-only recurrence of v_pk_fma_f32 (or, as control,
// the same arithmetic with v_fma_f32) - whose result is compared bit for bit with its own result obtained alone.  No LDS,
// no memory traffic inside the loops, no shared data between the two kernels.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_pk_hazard.hip -o build_tools/mfma_pk_hazard && build_tools/mfma_pk_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float v2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void aggressor(float *out, int iters, unsigned seed)
{
    union { bf16x8 v; unsigned u[4]; } a, b;
    for (int i = 0; i < 4; ++i) { a.u[i] = 0x3f803f80u + (threadIdx.x & 7) + i; b.u[i] = 0x3f803f80u + (seed & 3) + i; }
    f32x16 acc[2]; f32x4 acc4[4];
    for (int n = 0; n < 2; ++n) for (int v = 0; v < 16; ++v) acc[n][v] = 0.f;
    for (int n = 0; n < 4; ++n) for (int v = 0; v < 4; ++v) acc4[n][v] = 0.f;
    const float fa = __uint_as_float(a.u[0]), fb = __uint_as_float(b.u[0]);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (KIND == 0) { for (int n = 0; n < 4; ++n) acc4[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, acc4[n], 0, 0, 0); }
            if (KIND == 1) { for (int n = 0; n < 2; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc[n], 0, 0, 0); }
            if (KIND == 2) { for (int n = 0; n < 2; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[n], 0, 0, 0); }
            if (KIND == 3) { for (int n = 0; n < 4; ++n) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(acc4[n][0]) : "v"(fa)); }   // control: VALU only
        }
        if (KIND != 3) for (int n = 0; n < 4; ++n) for (int v = 0; v < 4; ++v) acc4[n][v] *= 0.5f;      // keep the values finite
        if (KIND != 3) for (int n = 0; n < 2; ++n) for (int v = 0; v < 16; ++v) acc[n][v] *= 0.5f;
    }
    float s = 0.f;
    for (int n = 0; n < 2; ++n) for (int v = 0; v < 16; ++v) s += acc[n][v];
    for (int n = 0; n < 4; ++n) for (int v = 0; v < 4; ++v) s += acc4[n][v];
    if (s == 123.456f) out[0] = s;
}

template <bool PACKED>
__global__ __launch_bounds__(256) void victim(v2 *out, int iters)
{
    const unsigned t = blockIdx.x * 256 + threadIdx.x;
    v2 x = {1.0f + (float)(t & 1023) * 1e-3f, 0.5f + (float)(t & 511) * 2e-3f};
    v2 y = {0.25f, -0.75f};
    const v2 m = {0.9990234375f, -0.99951171875f}, c = {1e-3f, -2e-3f};
    for (int i = 0; i < iters; ++i) {
        if (PACKED) {
            asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %0\n v_pk_mul_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %3\n"
                         : "+v"(x), "+v"(y) : "v"(m), "v"(c));
        } else {
            asm volatile("v_fma_f32 %0, %0, %4, %6\n v_fma_f32 %1, %1, %5, %7\n v_fma_f32 %2, %2, %4, %0\n v_fma_f32 %3, %3, %5, %1\n"
                         "v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %5\n v_add_f32 %2, %2, %6\n v_add_f32 %3, %3, %7\n"
                         : "+v"(x.x), "+v"(x.y), "+v"(y.x), "+v"(y.y) : "v"(m.x), "v"(m.y), "v"(c.x), "v"(c.y));
        }
    }
    out[2 * t] = x; out[2 * t + 1] = y;
}

template <int KIND, bool PACKED>
static void run(const char *aname, const char *vname)
{
    const int vb = 2048, n = vb * 256 * 2;
    v2 *ref, *got; float *dummy;
    hipMalloc(&ref, n * sizeof(v2)); hipMalloc(&got, n * sizeof(v2)); hipMalloc(&dummy, 4);
    hipStream_t sa, sb; hipStreamCreate(&sa); hipStreamCreate(&sb);
    hipLaunchKernelGGL((victim<PACKED>), dim3(vb), dim3(256), 0, sb, ref, 4000);
    hipDeviceSynchronize();
    std::vector<v2> h0(n), h1(n);
    hipMemcpy(h0.data(), ref, n * sizeof(v2), hipMemcpyDeviceToHost);
    int bad_runs = 0; long bad_vals = 0; const int reps = 20;
    for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL((aggressor<KIND>), dim3(512), dim3(256), 0, sa, dummy, 12000, (unsigned)r);      // 2 workgroups per CU, a few ms: the victim fits beside them     // ~ms of matrix instructions
        hipLaunchKernelGGL((victim<PACKED>), dim3(vb), dim3(256), 0, sb, got, 4000);
        hipDeviceSynchronize();
        hipMemcpy(h1.data(), got, n * sizeof(v2), hipMemcpyDeviceToHost);
        long b = 0;
        for (int i = 0; i < n; ++i) b += memcmp(&h0[i], &h1[i], sizeof(v2)) != 0;
        bad_runs += b != 0; bad_vals += b;
    }
    printf("aggressor %-28s victim %-14s: %2d of %d runs differ from the victim alone (%ld values)\n", aname, vname, bad_runs, reps, bad_vals);
    hipFree(ref); hipFree(got); hipFree(dummy); hipStreamDestroy(sa); hipStreamDestroy(sb);
}

int main()
{
    run<0, true>("v_mfma_f32_16x16x32_bf16", "v_pk_*_f32");
    run<0, false>("v_mfma_f32_16x16x32_bf16", "v_fma/mul/add");
    run<1, true>("v_mfma_f32_32x32x16_bf16", "v_pk_*_f32");
    run<2, true>("v_mfma_f32_32x32x2_f32", "v_pk_*_f32");
    run<3, true>("v_fma_f32 only (control)", "v_pk_*_f32");
    return 0;
}
