// mfma_pk_hazard3.hip - which packed-fp32 instruction forms go wrong next to which matrix instructions (follows
// tools/mfma_pk_hazard2.hip, whose only failing victim was the one with op_sel / neg modifiers).
// Stream A loops an aggressor that issues one kind of matrix instruction on registers; stream B runs a CHECKING victim: every
// step it executes ONE v_pk_{fma,mul,add}_f32 with a given set of modifiers on fresh pseudo-random operands and compares the two
// result halves with the same arithmetic done by plain v_fma_f32 / v_mul_f32 / v_add_f32 (sign flips as integer xor, half
// selection as register moves).  Mismatches are counted; the first one per cell is printed with its operands.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_pk_hazard3.hip -o /tmp/h3 && /tmp/h3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float v2 __attribute__((ext_vector_type(2)));

template <int K>
__global__ __launch_bounds__(256) void aggressor(float *out, int iters, unsigned seed)
{
    union { bf16x8 b; f16x8 h; unsigned u[4]; float f[4]; } a, b;
    for (int i = 0; i < 4; ++i) { a.u[i] = 0x3c003c00u + (threadIdx.x & 7) + i; b.u[i] = 0x3c003c00u + (seed & 3) + i; }
    f32x4 acc4[4]; f32x16 acc16[2];
    for (int n = 0; n < 4; ++n) acc4[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int n = 0; n < 2; ++n) for (int v = 0; v < 16; ++v) acc16[n][v] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (K == 1) for (int n = 0; n < 4; ++n) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(acc4[n][0]) : "v"(a.f[0]));
            if (K == 2) for (int n = 0; n < 4; ++n) acc4[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.b, b.b, acc4[n], 0, 0, 0);
            if (K == 3) for (int n = 0; n < 2; ++n) acc16[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.b, b.b, acc16[n], 0, 0, 0);
            if (K == 4) for (int n = 0; n < 2; ++n) acc16[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.f[0], b.f[0], acc16[n], 0, 0, 0);
            if (K == 5) for (int n = 0; n < 4; ++n) acc4[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.h, b.h, acc4[n], 0, 0, 0);
            if (K == 6) for (int n = 0; n < 4; ++n) acc4[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.f[0], b.f[0], acc4[n], 0, 0, 0);
            if (K == 7) for (int n = 0; n < 2; ++n) acc16[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h, b.h, acc16[n], 0, 0, 0);
        }
        if (K != 1) { for (int n = 0; n < 4; ++n) acc4[n] *= 0.25f; for (int n = 0; n < 2; ++n) acc16[n] *= 0.25f; }
    }
    float s = 0.f;
    for (int n = 0; n < 4; ++n) for (int v = 0; v < 4; ++v) s += acc4[n][v];
    for (int n = 0; n < 2; ++n) for (int v = 0; v < 16; ++v) s += acc16[n][v];
    if (s == 123.456f) out[0] = s;
}

// modifiers of victim form Q: kind 0 fma 1 mul 2 add; sl/sh = op_sel / op_sel_hi per source; nl/nh = neg_lo / neg_hi per source
struct Form { int kind, sl[3], sh[3], nl[3], nh[3]; const char *text; };
__host__ __device__ constexpr Form form(int q)
{
    switch (q) {
    case 0: return {0, {0, 0, 0}, {1, 1, 1}, {0, 0, 0}, {0, 0, 0}, "v_pk_fma_f32 (no modifiers)"};
    case 1: return {0, {0, 0, 0}, {0, 1, 1}, {0, 0, 0}, {0, 0, 0}, "v_pk_fma_f32 op_sel_hi:[0,1,1]"};
    case 2: return {0, {1, 0, 0}, {1, 1, 1}, {0, 0, 0}, {0, 0, 0}, "v_pk_fma_f32 op_sel:[1,0,0]"};
    case 3: return {0, {0, 0, 0}, {1, 1, 1}, {0, 1, 0}, {0, 0, 0}, "v_pk_fma_f32 neg_lo:[0,1,0]"};
    case 4: return {0, {0, 0, 0}, {1, 1, 1}, {0, 0, 0}, {0, 1, 0}, "v_pk_fma_f32 neg_hi:[0,1,0]"};
    case 5: return {1, {0, 0, 0}, {0, 1, 1}, {0, 0, 0}, {0, 0, 0}, "v_pk_mul_f32 op_sel_hi:[0,1]"};
    case 6: return {2, {0, 0, 0}, {1, 1, 1}, {0, 0, 0}, {0, 1, 0}, "v_pk_add_f32 neg_hi:[0,1]"};
    default: return {1, {1, 0, 0}, {0, 1, 1}, {0, 0, 0}, {0, 0, 0}, "v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[0,1]"};
    }
}

__device__ inline float flip(float v, int neg) { return __uint_as_float(__float_as_uint(v) ^ (neg ? 0x80000000u : 0u)); }

template <int Q>
__global__ __launch_bounds__(256) void victim(unsigned *count, unsigned *first, int iters)
{
    constexpr Form F = form(Q);
    unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u, bad = 0;
    for (int i = 0; i < iters; ++i) {
        unsigned r[6];
        for (int k = 0; k < 6; ++k) { s = s * 1664525u + 1013904223u; r[k] = 0x3f800000u | (s >> 9); }
        const v2 x = {__uint_as_float(r[0]), __uint_as_float(r[1])}, m = {__uint_as_float(r[2]), __uint_as_float(r[3])},
                 c = {__uint_as_float(r[4]), __uint_as_float(r[5])};
        v2 got;
        if (Q == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(got) : "v"(x), "v"(m), "v"(c));
        if (Q == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(got) : "v"(x), "v"(m), "v"(c));
        if (Q == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(got) : "v"(x), "v"(m), "v"(c));
        if (Q == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,1,0]" : "=v"(got) : "v"(x), "v"(m), "v"(c));
        if (Q == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_hi:[0,1,0]" : "=v"(got) : "v"(x), "v"(m), "v"(c));
        if (Q == 5) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(got) : "v"(x), "v"(m));
        if (Q == 6) asm volatile("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(got) : "v"(x), "v"(m));
        if (Q == 7) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(got) : "v"(x), "v"(m));
        const float a0 = F.sl[0] ? x.y : x.x, a1 = F.sh[0] ? x.y : x.x;
        const float b0 = flip(F.sl[1] ? m.y : m.x, F.nl[1]), b1 = flip(F.sh[1] ? m.y : m.x, F.nh[1]);
        const float c0 = F.sl[2] ? c.y : c.x, c1 = F.sh[2] ? c.y : c.x;
        float w0, w1;
        if (F.kind == 0) { asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(w0) : "v"(a0), "v"(b0), "v"(c0));
                           asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(w1) : "v"(a1), "v"(b1), "v"(c1)); }
        if (F.kind == 1) { asm volatile("v_mul_f32 %0, %1, %2" : "=v"(w0) : "v"(a0), "v"(b0));
                           asm volatile("v_mul_f32 %0, %1, %2" : "=v"(w1) : "v"(a1), "v"(b1)); }
        if (F.kind == 2) { asm volatile("v_add_f32 %0, %1, %2" : "=v"(w0) : "v"(a0), "v"(b0));
                           asm volatile("v_add_f32 %0, %1, %2" : "=v"(w1) : "v"(a1), "v"(b1)); }
        if (__float_as_uint(w0) != __float_as_uint(got.x) || __float_as_uint(w1) != __float_as_uint(got.y)) {
            if (bad++ == 0 && atomicAdd(&first[0], 1u) == 0) {
                const float rec[10] = {x.x, x.y, m.x, m.y, c.x, c.y, got.x, got.y, w0, w1};
                for (int k = 0; k < 10; ++k) first[1 + k] = __float_as_uint(rec[k]);
            }
        }
    }
    if (bad) atomicAdd(count, bad);
}

static unsigned *g_cnt, *g_first; static float *g_dummy;

template <int K, int Q>
static void cell()
{
    hipStream_t sa, sb; hipStreamCreate(&sa); hipStreamCreate(&sb);
    hipMemset(g_cnt, 0, 4); hipMemset(g_first, 0, 64); hipDeviceSynchronize();
    const int reps = 6;
    for (int r = 0; r < reps; ++r) {
        if (K) hipLaunchKernelGGL((aggressor<K>), dim3(512), dim3(256), 0, sa, g_dummy, 40000, (unsigned)r);
        hipLaunchKernelGGL((victim<Q>), dim3(2048), dim3(256), 0, sb, g_cnt, g_first, 1500);
        hipDeviceSynchronize();
    }
    unsigned cnt, f[16];
    hipMemcpy(&cnt, g_cnt, 4, hipMemcpyDeviceToHost); hipMemcpy(f, g_first, 64, hipMemcpyDeviceToHost);
    printf(" %9u", cnt); fflush(stdout);
    if (cnt) {
        float v[10]; memcpy(v, f + 1, 40);
        fprintf(stderr, "  first mismatch [%s]: x=(%.9g,%.9g) m=(%.9g,%.9g) c=(%.9g,%.9g) got=(%.9g,%.9g) want=(%.9g,%.9g)\n", form(Q).text,
                v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], v[9]);
    }
    hipStreamDestroy(sa); hipStreamDestroy(sb);
}

template <int K>
static void row(const char *name)
{
    printf("%-28s", name);
    cell<K, 0>(); cell<K, 1>(); cell<K, 2>(); cell<K, 3>(); cell<K, 4>(); cell<K, 5>(); cell<K, 6>(); cell<K, 7>();
    printf("\n");
}

int main()
{
    hipMalloc(&g_cnt, 4); hipMalloc(&g_first, 64); hipMalloc(&g_dummy, 4);
    printf("wrong packed results out of %.0f checked per cell (6 launches x 2048 x 256 threads x 1500 steps)\n", 6.0 * 2048 * 256 * 1500);
    for (int q = 0; q < 8; ++q) printf("  Q%d = %s\n", q, form(q).text);
    printf("%-28s%10s%10s%10s%10s%10s%10s%10s%10s\n", "aggressor \\ victim", "Q0", "Q1", "Q2", "Q3", "Q4", "Q5", "Q6", "Q7");
    row<0>("none");
    row<1>("v_fma_f32 only");
    row<2>("v_mfma_f32_16x16x32_bf16");
    row<3>("v_mfma_f32_32x32x16_bf16");
    row<4>("v_mfma_f32_32x32x2_f32");
    row<5>("v_mfma_f32_16x16x32_f16");
    row<6>("v_mfma_f32_16x16x4_f32");
    row<7>("v_mfma_f32_32x32x16_f16");
    return 0;
}
