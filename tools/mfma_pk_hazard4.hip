// mfma_pk_hazard4.hip - third step of the stand-alone search (hazard2: a 4-instruction packed recurrence WITH op_sel / neg modifiers
// differs next to a matrix-instruction loop, 12 of 12; hazard3: the same modifiers one instruction at a time between other VALU work,
// checked in place: 0 of 4.7e9).  Here the recurrence itself is varied - which of its four instructions carries its modifier,
// s_nop between them, short chains - and every aggressor kind has a "none" control that checks the victim alone is reproducible.
//   hipcc --offload-arch=gfx950 -O2 -w tools/mfma_pk_hazard4.hip -o /tmp/h4 && /tmp/h4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float v2 __attribute__((ext_vector_type(2)));

// K: 0 no matrix instruction, 2 v_mfma_f32_16x16x32_bf16, 3 v_mfma_f32_32x32x16_bf16, 4 v_mfma_f32_32x32x2_f32 (4 per step, then the
// accumulators are halved).  SC: how they are halved - 0 `acc *= 0.5f` (the compiler emits v_pk_mul_f32 v, v, 0.5 op_sel_hi:[1,0] as in
// hazard2), 1 v_mul_f32 only, 2 v_pk_mul_f32 with a VGPR pair and no modifier, 3 v_pk_mul_f32 with op_sel_hi:[1,0] on a VGPR pair
template <int K, int SC>
__global__ __launch_bounds__(256, 2) void aggressor(float *out, int iters, unsigned seed)
{
    union { bf16x8 b; unsigned u[4]; float f[4]; } a, b;
    for (int i = 0; i < 4; ++i) { a.u[i] = 0x3f803f80u + (threadIdx.x & 7) + i; b.u[i] = 0x3f803f80u + (seed & 3) + i; }
    f32x4 acc[4]; f32x16 big = {0.f};
    for (int n = 0; n < 4; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const v2 half2 = {0.5f, 0.5f};
    for (int it = 0; it < iters; ++it) {
        if (K == 2) for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.b, b.b, acc[n], 0, 0, 0);
        if (K == 3) { big = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.b, b.b, big, 0, 0, 0); for (int n = 0; n < 4; ++n) for (int v = 0; v < 4; ++v) acc[n][v] += big[4 * n + v]; }
        if (K == 4) { big = __builtin_amdgcn_mfma_f32_32x32x2f32(a.f[0], b.f[0], big, 0, 0, 0); for (int n = 0; n < 4; ++n) for (int v = 0; v < 4; ++v) acc[n][v] += big[4 * n + v]; }
        if (K != 2 && K != 0) for (int v = 0; v < 16; ++v) big[v] *= 0.5f;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            if (SC == 0) acc[n] *= 0.5f;
            if (SC == 1) for (int v = 0; v < 4; ++v) asm volatile("v_mul_f32 %0, 0.5, %0" : "+v"(acc[n][v]));
            if (SC >= 2) {
                v2 lo = {acc[n][0], acc[n][1]}, hi = {acc[n][2], acc[n][3]};
                if (SC == 2) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(lo) : "v"(half2)); asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(hi) : "v"(half2)); }
                if (SC == 3) { asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(lo) : "v"(half2)); asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(hi) : "v"(half2)); }
                acc[n] = f32x4{lo.x, lo.y, hi.x, hi.y};
            }
        }
    }
    f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
    if (s[0] + s[1] + s[2] + s[3] + big[0] == 123.456f) out[0] = s[0];
}

#define I1M "v_pk_fma_f32 %0, %0, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]\n"
#define I1P "v_pk_fma_f32 %0, %0, %2, %3\n"
#define I2M "v_pk_fma_f32 %1, %1, %2, %0 neg_lo:[0,1,0]\n"
#define I2P "v_pk_fma_f32 %1, %1, %2, %0\n"
#define I3M "v_pk_mul_f32 %0, %0, %2 op_sel_hi:[0,1]\n"
#define I3P "v_pk_mul_f32 %0, %0, %2\n"
#define I4M "v_pk_add_f32 %1, %1, %3 neg_hi:[0,1]\n"
#define I4P "v_pk_add_f32 %1, %1, %3\n"
#define NOP "s_nop 7\n"
#define SEQ(text) asm volatile(text : "+v"(x), "+v"(y) : "v"(m), "v"(c))

static const char *const VNAME[] = {"S0 all four modified (hazard2's P2)", "S1 only fma op_sel/op_sel_hi", "S2 only fma neg_lo", "S3 only mul op_sel_hi",
                                    "S4 only add neg_hi", "S5 S0 with s_nop 7 between", "S6 S0, 16 steps only", "S7 no modifiers"};
template <int S>
__global__ __launch_bounds__(256) void victim(v2 *out, int iters)
{
    const unsigned t = blockIdx.x * 256 + threadIdx.x;
    v2 x = {1.0f + (float)(t & 1023) * 1e-3f, 0.5f + (float)(t & 511) * 2e-3f};
    v2 y = {0.25f, -0.75f};
    const v2 m = {0.9990234375f, -0.99951171875f}, c = {1e-3f, -2e-3f};
    for (int i = 0; i < iters; ++i) {
        if (S == 0 || S == 6) SEQ(I1M I2M I3M I4M);
        if (S == 1) SEQ(I1M I2P I3P I4P);
        if (S == 2) SEQ(I1P I2M I3P I4P);
        if (S == 3) SEQ(I1P I2P I3M I4P);
        if (S == 4) SEQ(I1P I2P I3P I4M);
        if (S == 5) SEQ(I1M NOP I2M NOP I3M NOP I4M NOP);
        if (S == 7) SEQ(I1P I2P I3P I4P);
    }
    out[2 * t] = x; out[2 * t + 1] = y;
}

static float *g_dummy;

template <int K, int SC, int S>
static void cell()
{
    const int vb = 2048, n = vb * 256 * 2, viters = S == 6 ? 16 : 4000;
    v2 *ref, *got;
    hipMalloc(&ref, n * sizeof(v2)); hipMalloc(&got, n * sizeof(v2));
    hipStream_t sa, sb; hipStreamCreate(&sa); hipStreamCreate(&sb);
    hipLaunchKernelGGL((victim<S>), dim3(vb), dim3(256), 0, sb, ref, viters);
    hipDeviceSynchronize();
    std::vector<v2> h0(n), h1(n);
    hipMemcpy(h0.data(), ref, n * sizeof(v2), hipMemcpyDeviceToHost);
    int bad_runs = 0; long bad_vals = 0; const int reps = 8; bool shown = false;
    for (int r = 0; r < reps; ++r) {
        if (K >= 0) hipLaunchKernelGGL((aggressor<(K < 0 ? 0 : K), SC>), dim3(512), dim3(256), 0, sa, g_dummy, 96000, (unsigned)r);
        hipLaunchKernelGGL((victim<S>), dim3(vb), dim3(256), 0, sb, got, viters);
        hipDeviceSynchronize();
        hipMemcpy(h1.data(), got, n * sizeof(v2), hipMemcpyDeviceToHost);
        long b = 0;
        for (int i = 0; i < n; ++i)
            if (memcmp(&h0[i], &h1[i], sizeof(v2))) {
                if (!shown && b < 3)
                    fprintf(stderr, "   [K%d SC%d %s] thread %d (lane %d) %s: alone (%.9g, %.9g) shared (%.9g, %.9g)\n", K, SC, VNAME[S], i / 2, (i / 2) & 63,
                            i & 1 ? "y" : "x", h0[i].x, h0[i].y, h1[i].x, h1[i].y);
                ++b;
            }
        if (b) shown = true;
        bad_runs += b != 0; bad_vals += b;
    }
    printf(" %d/%d:%-8ld", bad_runs, reps, bad_vals); fflush(stdout);
    hipFree(ref); hipFree(got); hipStreamDestroy(sa); hipStreamDestroy(sb);
}

template <int K, int SC>
static void row(const char *name)
{
    printf("%-50s", name);
    cell<K, SC, 0>(); cell<K, SC, 1>(); cell<K, SC, 2>(); cell<K, SC, 3>(); cell<K, SC, 4>(); cell<K, SC, 5>(); cell<K, SC, 6>(); cell<K, SC, 7>();
    printf("\n");
}

int main()
{
    hipMalloc(&g_dummy, 4);
    printf("cells: runs whose result differs from the first launch / runs : differing values (of %d per run)\n", 2048 * 256 * 2);
    for (int s = 0; s < 8; ++s) printf("  %s\n", VNAME[s]);
    printf("%-50s%-13s%-13s%-13s%-13s%-13s%-13s%-13s%-13s\n", "aggressor \\ victim", " S0", " S1", " S2", " S3", " S4", " S5", " S6", " S7");
    row<-1, 0>("none (victim alone again)");
    row<0, 0>("no matrix instr., v_pk_mul 0.5 op_sel_hi:[1,0]");
    row<2, 0>("16x16x32_bf16 + v_pk_mul 0.5 op_sel_hi:[1,0]");
    row<2, 1>("16x16x32_bf16 + v_mul_f32 only");
    row<2, 2>("16x16x32_bf16 + v_pk_mul VGPR pair, no modifier");
    row<2, 3>("16x16x32_bf16 + v_pk_mul VGPR pair op_sel_hi:[1,0]");
    row<3, 0>("32x32x16_bf16 + v_pk_mul 0.5 op_sel_hi:[1,0]");
    row<4, 0>("32x32x2_f32 + v_pk_mul 0.5 op_sel_hi:[1,0]");
    return 0;
}
