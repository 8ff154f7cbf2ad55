// mfma_pk_hazard2.hip - second attempt at a stand-alone form of the GPU-sharing finding (profiles/r04_gpu_sharing_finding.md,
// profiles/r05_shared_gpu.md).  tools/mfma_pk_hazard.hip (register-only operands, a few VGPRs) stays clean; this one adds, one at a
// time, what the real kernels have and that one lacks:
//   aggressor (v_mfma_f32_16x16x32_bf16 loop, stream A):  A0 few VGPRs, registers only | A1 ~250 VGPRs (the occupancy pattern of
//       wino_fused16_kernel: one workgroup takes half of every SIMD's register file) | A2 = A1 + 52 KB LDS, operands through
//       ds_read_b128 | A3 = A2 + the B operand re-loaded from global memory
//   victim (v_pk_{fma,mul,add}_f32 recurrence, stream B): P0 registers only | P1 an SGPR-pair source | P2 op_sel / neg modifiers |
//       P3 one operand loaded from global memory per step | P4 one operand through LDS per step
// Every victim result is compared bit for bit with the same launch alone.  Prints a matrix of "runs that differ / runs".
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_pk_hazard2.hip -o /tmp/mfma_pk_hazard2 && /tmp/mfma_pk_hazard2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float v2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NACC, bool LDS, bool GLB>
__global__ __launch_bounds__(256, 2) void aggressor(float *out, const u32x4 *wts, int iters, unsigned seed)
{
    __shared__ u32x4 tile[LDS ? 3328 : 1];                               // 52 KB when used
    union { bf16x8 v; u32x4 u; } a, b;
    for (int i = 0; i < 4; ++i) { a.u[i] = 0x3f803f80u + (threadIdx.x & 7) + i; b.u[i] = 0x3f803f80u + (seed & 3) + i; }
    if (LDS) { for (int i = threadIdx.x; i < 3328; i += 256) tile[i] = a.u + (unsigned)(i & 3); __syncthreads(); }
    f32x4 acc[NACC];
#pragma unroll
    for (int n = 0; n < NACC; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        if (GLB) b.u = wts[(it * 64 + (threadIdx.x & 63)) & 4095];
#pragma unroll
        for (int n = 0; n < NACC; ++n) {
            if (LDS && (n & 3) == 0) a.u = tile[(threadIdx.x + 67 * n + 13 * it) % 3328];
            acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, acc[n], 0, 0, 0);
        }
#pragma unroll
        for (int n = 0; n < NACC; ++n) acc[n] *= 0.5f;                    // VALU on the accumulators: they live in VGPRs
    }
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < NACC; ++n) s += acc[n];
    if (s[0] + s[1] + s[2] + s[3] == 123.456f) out[0] = s[0];
}

template <int P>
__global__ __launch_bounds__(256) void victim(v2 *out, const v2 *tab, int iters)
{
    __shared__ v2 lt[512];
    const unsigned t = blockIdx.x * 256 + threadIdx.x;
    v2 x = {1.0f + (float)(t & 1023) * 1e-3f, 0.5f + (float)(t & 511) * 2e-3f};
    v2 y = {0.25f, -0.75f};
    const v2 m = {0.9990234375f, -0.99951171875f}, c = {1e-3f, -2e-3f};
    if (P == 4) { lt[threadIdx.x] = tab[threadIdx.x]; lt[threadIdx.x + 256] = tab[threadIdx.x + 256]; __syncthreads(); }
    for (int i = 0; i < iters; ++i) {
        if (P == 0)
            asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %0\n v_pk_mul_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %3\n"
                         : "+v"(x), "+v"(y) : "v"(m), "v"(c));
        if (P == 1)
            asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %0\n v_pk_mul_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %3\n"
                         : "+v"(x), "+v"(y) : "s"(m), "v"(c));
        if (P == 2)
            asm volatile("v_pk_fma_f32 %0, %0, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %1, %2, %0 neg_lo:[0,1,0]\n"
                         "v_pk_mul_f32 %0, %0, %2 op_sel_hi:[0,1]\n v_pk_add_f32 %1, %1, %3 neg_hi:[0,1]\n"
                         : "+v"(x), "+v"(y) : "v"(m), "v"(c));
        if (P == 3 || P == 4) {
            const v2 w = P == 3 ? tab[(t + 37u * i) & 511] : lt[(threadIdx.x + 37u * i) & 511];      // values near 1: the recurrence stays finite
            asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %4, %0\n v_pk_mul_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %3\n"
                         : "+v"(x), "+v"(y) : "v"(m), "v"(c), "v"(w));
        }
    }
    out[2 * t] = x; out[2 * t + 1] = y;
}

static v2 *g_tab; static u32x4 *g_wts; static float *g_dummy;

template <int NACC, bool LDS, bool GLB, int P>
static void cell(int agg_blocks)
{
    const int vb = 2048, n = vb * 256 * 2, viters = P >= 3 ? 1500 : 4000;
    v2 *ref, *got;
    hipMalloc(&ref, n * sizeof(v2)); hipMalloc(&got, n * sizeof(v2));
    hipStream_t sa, sb; hipStreamCreate(&sa); hipStreamCreate(&sb);
    hipLaunchKernelGGL((victim<P>), dim3(vb), dim3(256), 0, sb, ref, g_tab, viters);
    hipDeviceSynchronize();
    std::vector<v2> h0(n), h1(n);
    hipMemcpy(h0.data(), ref, n * sizeof(v2), hipMemcpyDeviceToHost);
    int bad_runs = 0; const int reps = 12;
    for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL((aggressor<NACC, LDS, GLB>), dim3(agg_blocks), dim3(256), 0, sa, g_dummy, g_wts, 384000 / NACC, (unsigned)r);
        hipLaunchKernelGGL((victim<P>), dim3(vb), dim3(256), 0, sb, got, g_tab, viters);
        hipDeviceSynchronize();
        hipMemcpy(h1.data(), got, n * sizeof(v2), hipMemcpyDeviceToHost);
        bad_runs += memcmp(h0.data(), h1.data(), n * sizeof(v2)) != 0;
    }
    printf(" %2d/%d", bad_runs, reps); fflush(stdout);
    hipFree(ref); hipFree(got); hipStreamDestroy(sa); hipStreamDestroy(sb);
}

template <int NACC, bool LDS, bool GLB>
static void row(const char *name, int agg_blocks)
{
    printf("%-44s", name);
    cell<NACC, LDS, GLB, 0>(agg_blocks); cell<NACC, LDS, GLB, 1>(agg_blocks); cell<NACC, LDS, GLB, 2>(agg_blocks);
    cell<NACC, LDS, GLB, 3>(agg_blocks); cell<NACC, LDS, GLB, 4>(agg_blocks);
    printf("\n");
}

int main()
{
    std::vector<v2> tab(512);
    for (int i = 0; i < 512; ++i) tab[i] = v2{1.0f - (float)(i % 7) * 1e-4f, -1.0f + (float)(i % 5) * 1e-4f};
    std::vector<unsigned> w(4096 * 4);
    for (size_t i = 0; i < w.size(); ++i) w[i] = 0x3f803f80u + (unsigned)(i % 5);
    hipMalloc(&g_tab, 512 * sizeof(v2)); hipMemcpy(g_tab, tab.data(), 512 * sizeof(v2), hipMemcpyHostToDevice);
    hipMalloc(&g_wts, w.size() * 4); hipMemcpy(g_wts, w.data(), w.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&g_dummy, 4);
    printf("runs whose victim result differs from the victim alone / runs\n%-44s%6s%6s%6s%6s%6s\n", "aggressor \\ victim", "P0", "P1", "P2", "P3", "P4");
    row<4, false, false>("A0 16 acc VGPRs, registers only, 512 wg", 512);
    row<56, false, false>("A1 224 acc VGPRs, registers only, 256 wg", 256);
    row<56, false, false>("A1 224 acc VGPRs, registers only, 512 wg", 512);
    row<56, true, false>("A2 A1 + 52 KB LDS operand reads, 256 wg", 256);
    row<56, true, false>("A2 A1 + 52 KB LDS operand reads, 512 wg", 512);
    row<56, true, true>("A3 A2 + B operand from global, 512 wg", 512);
    return 0;
}
