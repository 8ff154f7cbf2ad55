// mfma_pk_hazard5.hip - the stand-alone form of the GPU-sharing finding, in place (follows hazard4: a packed recurrence whose
// v_pk_fma_f32 swaps the halves of one source - op_sel:[0,1,0] op_sel_hi:[1,0,1] - gives different results next to a loop of bf16
// matrix instructions; fp32 matrix instructions and plain VALU neighbours do not).
// Stream A: an aggressor that issues ONE kind of matrix instruction (4 per step, accumulators halved by v_mul_f32).  Stream B: a
// checking victim - bursts of 8 independent v_pk_{fma,mul,add}_f32 with one modifier form on pseudo-random operands, each result half
// compared with plain v_fma_f32 / v_mul_f32 / v_add_f32 on the selected halves.  Wrong results are counted per form and per lane;
// the first few are printed with their operands.
//   hipcc --offload-arch=gfx950 -O2 -w tools/mfma_pk_hazard5.hip -o /tmp/h5 && /tmp/h5
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float v2 __attribute__((ext_vector_type(2)));

static const char *const KNAME[] = {"none", "v_mfma_f32_16x16x32_bf16", "v_mfma_f32_32x32x16_bf16", "v_mfma_f32_16x16x32_f16", "v_mfma_f32_16x16x16_f16",
                                    "v_mfma_f32_16x16x16_bf16 (1k)", "v_mfma_i32_16x16x64_i8", "v_mfma_f32_16x16x4_f32", "v_mfma_f32_32x32x2_f32",
                                    "v_mfma_f64_16x16x4_f64", "v_fma_f32 only"};
template <int K>
__global__ __launch_bounds__(256, 2) void aggressor(float *out, int iters, unsigned seed)
{
    union { bf16x8 b; f16x8 h; f16x4 h4; s16x4 s4; i32x4 i; unsigned u[4]; float f[4]; double d[2]; } a, b;
    for (int i = 0; i < 4; ++i) { a.u[i] = 0x3c003c00u + (threadIdx.x & 7) + i; b.u[i] = 0x3c003c00u + (seed & 3) + i; }
    f32x4 acc[4]; f32x16 big = {0.f}; f64x4 dac = {0.0}; i32x4 iac = {0};
    for (int n = 0; n < 4; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            if (K == 1) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.b, b.b, acc[n], 0, 0, 0);
            if (K == 2) big = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.b, b.b, big, 0, 0, 0);
            if (K == 3) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.h, b.h, acc[n], 0, 0, 0);
            if (K == 4) acc[n] = __builtin_amdgcn_mfma_f32_16x16x16f16(a.h4, b.h4, acc[n], 0, 0, 0);
            if (K == 5) acc[n] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a.s4, b.s4, acc[n], 0, 0, 0);
            if (K == 6) iac = __builtin_amdgcn_mfma_i32_16x16x64_i8(a.i, b.i, iac, 0, 0, 0);
            if (K == 7) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.f[0], b.f[0], acc[n], 0, 0, 0);
            if (K == 8) big = __builtin_amdgcn_mfma_f32_32x32x2f32(a.f[0], b.f[0], big, 0, 0, 0);
            if (K == 9) dac = __builtin_amdgcn_mfma_f64_16x16x4f64(a.d[0], b.d[0], dac, 0, 0, 0);
            if (K == 10) for (int v = 0; v < 4; ++v) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(acc[n][v]) : "v"(a.f[0]));
        }
        for (int n = 0; n < 4; ++n) for (int v = 0; v < 4; ++v) asm volatile("v_mul_f32 %0, 0.5, %0" : "+v"(acc[n][v]));
        if (K == 2 || K == 8) for (int v = 0; v < 16; ++v) asm volatile("v_mul_f32 %0, 0.5, %0" : "+v"(big[v]));
        if (K == 6) for (int v = 0; v < 4; ++v) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(iac[v]));
        if (K == 9) for (int v = 0; v < 4; ++v) asm volatile("v_mul_f64 %0, 0.5, %0" : "+v"(dac[v]));
    }
    f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
    if (s[0] + s[1] + s[2] + s[3] + big[0] + (float)dac[0] + (float)iac[0] == 123.456f) out[0] = s[0];
}

// victim forms: kind 0 fma 1 mul 2 add; sl / sh = which half (0 low, 1 high) of each source feeds the low / high result
struct Form { int kind, sl[3], sh[3]; const char *text; int sg = -1; int neg[3] = {0, 0, 0}; float k1 = 0.f; };
// sg: which source is a scalar-register pair; neg: source negated (neg_lo and neg_hi both); k1 != 0: source 1 is that inline constant
__host__ __device__ constexpr Form form(int q)
{
    switch (q) {
    case 0: return {0, {0, 0, 0}, {1, 1, 1}, "v_pk_fma_f32 (no modifiers)"};
    case 1: return {0, {0, 1, 0}, {1, 0, 1}, "v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[1,0,1]  (src1 halves swapped)"};
    case 2: return {0, {0, 1, 0}, {1, 1, 1}, "v_pk_fma_f32 op_sel:[0,1,0]                    (src1 high half to both)"};
    case 3: return {0, {0, 0, 0}, {1, 0, 1}, "v_pk_fma_f32 op_sel_hi:[1,0,1]                 (src1 low half to both)"};
    case 4: return {0, {1, 0, 0}, {0, 1, 1}, "v_pk_fma_f32 op_sel:[1,0,0] op_sel_hi:[0,1,1]  (src0 halves swapped)"};
    case 5: return {0, {0, 0, 1}, {1, 1, 0}, "v_pk_fma_f32 op_sel:[0,0,1] op_sel_hi:[1,1,0]  (src2 halves swapped)"};
    case 6: return {1, {0, 1, 0}, {1, 0, 0}, "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]      (src1 halves swapped)"};
    case 7: return {2, {0, 1, 0}, {1, 0, 0}, "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]      (src1 halves swapped)"};
    case 8: return {1, {0, 1, 0}, {0, 0, 0}, "v_pk_mul_f32 v, v, S op_sel:[0,1] op_sel_hi:[0,0]  (as the compiler wrote it)", 1};
    case 9: return {1, {1, 0, 0}, {0, 0, 0}, "v_pk_mul_f32 v, S, v op_sel:[1,0] op_sel_hi:[0,0]  (Q8 with sources exchanged)", 0};
    case 10: return {0, {0, 1, 0}, {1, 1, 1}, "v_pk_fma_f32 v, S, v, v op_sel:[0,1,0]             (as the compiler wrote it)", 0};
    case 11: return {0, {1, 0, 0}, {1, 1, 1}, "v_pk_fma_f32 v, v, S, v op_sel:[1,0,0]             (Q10 with sources exchanged)", 1};
    // the other forms libsp3d.so contains (llvm-objdump of the finished library): negated sources, inline constants
    case 12: return {0, {0, 0, 0}, {1, 0, 1}, "v_pk_fma_f32 op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]", -1, {1, 0, 0}};
    case 13: return {2, {0, 0, 0}, {1, 1, 0}, "v_pk_add_f32 neg_lo:[0,1] neg_hi:[0,1]                 (a subtraction)", -1, {0, 1, 0}};
    case 14: return {1, {0, 0, 0}, {1, 0, 0}, "v_pk_mul_f32 v, v, 0.5 op_sel_hi:[1,0]", -1, {0, 0, 0}, 0.5f};
    default: return {2, {0, 0, 0}, {1, 0, 0}, "v_pk_add_f32 v, v, 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]", -1, {1, 0, 0}, 1.0f};
    }
}
template <int Q>
__device__ inline void burst(v2 (&g)[8], const v2 (&x)[8], const v2 (&m)[8], const v2 (&c)[8], const v2 sk)
{
#define OPS "=v"(g[0]), "=v"(g[1]), "=v"(g[2]), "=v"(g[3]), "=v"(g[4]), "=v"(g[5]), "=v"(g[6]), "=v"(g[7])
#define INS "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), \
            "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]), "v"(c[6]), "v"(c[7])
#define B3(op, mod) asm volatile(op " %0, %8, %16, %24 " mod "\n" op " %1, %9, %17, %25 " mod "\n" op " %2, %10, %18, %26 " mod "\n" op " %3, %11, %19, %27 " mod "\n" \
                                 op " %4, %12, %20, %28 " mod "\n" op " %5, %13, %21, %29 " mod "\n" op " %6, %14, %22, %30 " mod "\n" op " %7, %15, %23, %31 " mod "\n" : OPS : INS)
#define B2(op, mod) asm volatile(op " %0, %8, %16 " mod "\n" op " %1, %9, %17 " mod "\n" op " %2, %10, %18 " mod "\n" op " %3, %11, %19 " mod "\n" \
                                 op " %4, %12, %20 " mod "\n" op " %5, %13, %21 " mod "\n" op " %6, %14, %22 " mod "\n" op " %7, %15, %23 " mod "\n" : OPS : INS)
    if (Q == 0) B3("v_pk_fma_f32", "");
    if (Q == 1) B3("v_pk_fma_f32", "op_sel:[0,1,0] op_sel_hi:[1,0,1]");
    if (Q == 2) B3("v_pk_fma_f32", "op_sel:[0,1,0]");
    if (Q == 3) B3("v_pk_fma_f32", "op_sel_hi:[1,0,1]");
    if (Q == 4) B3("v_pk_fma_f32", "op_sel:[1,0,0] op_sel_hi:[0,1,1]");
    if (Q == 5) B3("v_pk_fma_f32", "op_sel:[0,0,1] op_sel_hi:[1,1,0]");
    if (Q == 6) B2("v_pk_mul_f32", "op_sel:[0,1] op_sel_hi:[1,0]");
    if (Q == 7) B2("v_pk_add_f32", "op_sel:[0,1] op_sel_hi:[1,0]");
    // scalar-register forms: the pair sk rides in as operand 32; x / m are the vector sources in their order of appearance
#define INSS INS, "s"(sk)
#define B2S1(op, mod) asm volatile(op " %0, %8, %32 " mod "\n" op " %1, %9, %32 " mod "\n" op " %2, %10, %32 " mod "\n" op " %3, %11, %32 " mod "\n" \
                                   op " %4, %12, %32 " mod "\n" op " %5, %13, %32 " mod "\n" op " %6, %14, %32 " mod "\n" op " %7, %15, %32 " mod "\n" : OPS : INSS)
#define B2S0(op, mod) asm volatile(op " %0, %32, %8 " mod "\n" op " %1, %32, %9 " mod "\n" op " %2, %32, %10 " mod "\n" op " %3, %32, %11 " mod "\n" \
                                   op " %4, %32, %12 " mod "\n" op " %5, %32, %13 " mod "\n" op " %6, %32, %14 " mod "\n" op " %7, %32, %15 " mod "\n" : OPS : INSS)
#define B3S0(op, mod) asm volatile(op " %0, %32, %8, %24 " mod "\n" op " %1, %32, %9, %25 " mod "\n" op " %2, %32, %10, %26 " mod "\n" op " %3, %32, %11, %27 " mod "\n" \
                                   op " %4, %32, %12, %28 " mod "\n" op " %5, %32, %13, %29 " mod "\n" op " %6, %32, %14, %30 " mod "\n" op " %7, %32, %15, %31 " mod "\n" : OPS : INSS)
#define B3S1(op, mod) asm volatile(op " %0, %8, %32, %24 " mod "\n" op " %1, %9, %32, %25 " mod "\n" op " %2, %10, %32, %26 " mod "\n" op " %3, %11, %32, %27 " mod "\n" \
                                   op " %4, %12, %32, %28 " mod "\n" op " %5, %13, %32, %29 " mod "\n" op " %6, %14, %32, %30 " mod "\n" op " %7, %15, %32, %31 " mod "\n" : OPS : INSS)
    if (Q == 8) B2S1("v_pk_mul_f32", "op_sel:[0,1] op_sel_hi:[0,0]");
    if (Q == 9) B2S0("v_pk_mul_f32", "op_sel:[1,0] op_sel_hi:[0,0]");
    if (Q == 10) B3S0("v_pk_fma_f32", "op_sel:[0,1,0]");
    if (Q == 11) B3S1("v_pk_fma_f32", "op_sel:[1,0,0]");
    if (Q == 12) B3("v_pk_fma_f32", "op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]");
    if (Q == 13) B2("v_pk_add_f32", "neg_lo:[0,1] neg_hi:[0,1]");
#define B2K(op, k, mod) asm volatile(op " %0, %8, " k " " mod "\n" op " %1, %9, " k " " mod "\n" op " %2, %10, " k " " mod "\n" op " %3, %11, " k " " mod "\n" \
                                     op " %4, %12, " k " " mod "\n" op " %5, %13, " k " " mod "\n" op " %6, %14, " k " " mod "\n" op " %7, %15, " k " " mod "\n" : OPS : INS)
    if (Q == 14) B2K("v_pk_mul_f32", "0.5", "op_sel_hi:[1,0]");
    if (Q == 15) B2K("v_pk_add_f32", "1.0", "op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]");
}

// out: [0] wrong results, [1] records taken, [2..65] wrong results per lane, [66 ..] up to 4 records of 10 floats
template <int Q>
__global__ __launch_bounds__(256) void victim(unsigned *out, int iters, float sk0, float sk1)
{
    constexpr Form F = form(Q);
    unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u, bad = 0;
    for (int i = 0; i < iters; ++i) {
        v2 x[8], m[8], c[8], g[8];
        for (int k = 0; k < 8; ++k) {
            unsigned r[6];
            for (int j = 0; j < 6; ++j) { s = s * 1664525u + 1013904223u; r[j] = 0x3f800000u | (s >> 9); }
            x[k] = v2{__uint_as_float(r[0]), __uint_as_float(r[1])}; m[k] = v2{__uint_as_float(r[2]), __uint_as_float(r[3])};
            c[k] = v2{__uint_as_float(r[4]), __uint_as_float(r[5])};
        }
        const v2 sk = {sk0, sk1};
        burst<Q>(g, x, m, c, sk);
        for (int k = 0; k < 8; ++k) {
            const v2 kk = {F.k1, F.k1};
            const v2 s0 = F.sg == 0 ? sk : x[k], s1 = F.k1 != 0.f ? kk : F.sg == 1 ? sk : (F.sg == 0 ? x[k] : m[k]);   // what the instruction had as source 0 / 1
            const unsigned n0 = F.neg[0] ? 0x80000000u : 0u, n1 = F.neg[1] ? 0x80000000u : 0u;
            const float a0 = __uint_as_float(__float_as_uint(F.sl[0] ? s0.y : s0.x) ^ n0), a1 = __uint_as_float(__float_as_uint(F.sh[0] ? s0.y : s0.x) ^ n0);
            const float b0 = __uint_as_float(__float_as_uint(F.sl[1] ? s1.y : s1.x) ^ n1), b1 = __uint_as_float(__float_as_uint(F.sh[1] ? s1.y : s1.x) ^ n1);
            const float c0 = F.sl[2] ? c[k].y : c[k].x, c1 = F.sh[2] ? c[k].y : c[k].x;
            float w0, w1;
            if (F.kind == 0) { asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(w0) : "v"(a0), "v"(b0), "v"(c0));
                               asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(w1) : "v"(a1), "v"(b1), "v"(c1)); }
            if (F.kind == 1) { asm volatile("v_mul_f32 %0, %1, %2" : "=v"(w0) : "v"(a0), "v"(b0)); asm volatile("v_mul_f32 %0, %1, %2" : "=v"(w1) : "v"(a1), "v"(b1)); }
            if (F.kind == 2) { asm volatile("v_add_f32 %0, %1, %2" : "=v"(w0) : "v"(a0), "v"(b0)); asm volatile("v_add_f32 %0, %1, %2" : "=v"(w1) : "v"(a1), "v"(b1)); }
            if (__float_as_uint(w0) != __float_as_uint(g[k].x) || __float_as_uint(w1) != __float_as_uint(g[k].y)) {
                ++bad;
                atomicAdd(&out[2 + (threadIdx.x & 63)], 1u);
                const unsigned slot = atomicAdd(&out[1], 1u);
                if (slot < 4) {
                    const float rec[10] = {s0.x, s0.y, s1.x, s1.y, c[k].x, c[k].y, g[k].x, g[k].y, w0, w1};
                    for (int j = 0; j < 10; ++j) out[66 + 10 * slot + j] = __float_as_uint(rec[j]);
                }
            }
        }
    }
    if (bad) atomicAdd(&out[0], bad);
}

static unsigned *g_out; static float *g_dummy;
static std::vector<unsigned> g_counts;      // every cell in print order, for the JSON line

template <int K, int Q>
static void cell()
{
    hipStream_t sa, sb; hipStreamCreate(&sa); hipStreamCreate(&sb);
    hipMemset(g_out, 0, 128 * 4); hipDeviceSynchronize();
    for (int r = 0; r < 4; ++r) {
        if (K) hipLaunchKernelGGL((aggressor<K>), dim3(512), dim3(256), 0, sa, g_dummy, 96000, (unsigned)r);
        hipLaunchKernelGGL((victim<Q>), dim3(2048), dim3(256), 0, sb, g_out, 300, 1.25f, 1.75f);
        hipDeviceSynchronize();
    }
    unsigned o[128];
    hipMemcpy(o, g_out, sizeof(o), hipMemcpyDeviceToHost);
    printf(" %9u", o[0]); fflush(stdout);
    g_counts.push_back(o[0]);
    if (o[0]) {
        fprintf(stderr, "[%s | %s]\n  wrong results per lane:", KNAME[K], form(Q).text);
        for (int l = 0; l < 64; ++l) fprintf(stderr, "%s%u", l % 16 ? " " : "\n    ", o[2 + l]);
        fprintf(stderr, "\n");
        for (unsigned k = 0; k < (o[1] < 4 ? o[1] : 4); ++k) {
            float v[10]; memcpy(v, o + 66 + 10 * k, 40);
            fprintf(stderr, "  src0=(%.9g, %.9g) src1=(%.9g, %.9g) src2=(%.9g, %.9g) got=(%.9g, %.9g) want=(%.9g, %.9g)\n", v[0], v[1], v[2], v[3], v[4], v[5],
                    v[6], v[7], v[8], v[9]);
        }
    }
    hipStreamDestroy(sa); hipStreamDestroy(sb);
}

template <int K>
static void row()
{
    printf("%-32s", KNAME[K]);
    cell<K, 0>(); cell<K, 1>(); cell<K, 2>(); cell<K, 3>(); cell<K, 4>(); cell<K, 5>(); cell<K, 6>(); cell<K, 7>(); cell<K, 8>(); cell<K, 9>(); cell<K, 10>(); cell<K, 11>(); cell<K, 12>(); cell<K, 13>(); cell<K, 14>(); cell<K, 15>();
    printf("\n");
}

int main()
{
    hipMalloc(&g_out, 128 * 4); hipMalloc(&g_dummy, 4);
    printf("wrong packed results out of %.0f checked per cell (4 launches x 2048 x 256 threads x 300 bursts of 8)\n", 4.0 * 2048 * 256 * 300 * 8);
    for (int q = 0; q < 16; ++q) printf("  Q%d = %s\n", q, form(q).text);
    printf("%-32s%10s%10s%10s%10s%10s%10s%10s%10s%10s%10s%10s%10s%10s%10s%10s%10s\n", "neighbour on stream A", "Q0", "Q1", "Q2", "Q3", "Q4", "Q5", "Q6", "Q7", "Q8", "Q9", "Q10",
           "Q11", "Q12", "Q13", "Q14", "Q15");
    row<0>(); row<10>(); row<1>(); row<2>(); row<3>(); row<4>(); row<5>(); row<6>(); row<7>(); row<8>(); row<9>();
    const int order[11] = {0, 10, 1, 2, 3, 4, 5, 6, 7, 8, 9};
    printf("JSON {");                                            // one line for tests/test_gpu_shared_gpu.py
    for (int r = 0; r < 11; ++r) {
        printf("%s\"%s\": [", r ? ", " : "", KNAME[order[r]]);
        for (int q = 0; q < 16; ++q) printf("%s%u", q ? ", " : "", g_counts[16 * r + q]);
        printf("]");
    }
    printf("}\n");
    return 0;
}
