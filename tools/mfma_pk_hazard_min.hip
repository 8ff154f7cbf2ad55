// Minimal reproducer (MI355X, gfx950, ROCm 7.2): a packed-fp32 VALU instruction whose LOW result reads the HIGH half of source 1
// (op_sel:[0,1]) returns that operand as ZERO in lanes 48-63 while waves of another kernel on the same CU execute
// v_mfma_f32_16x16x32_bf16.  Alone, or with op_sel on source 0 instead, it is correct.  Full matrix: tools/mfma_pk_hazard5.hip.
//   hipcc --offload-arch=gfx950 -O2 -w tools/mfma_pk_hazard_min.hip -o /tmp/hmin && /tmp/hmin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float v2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void neighbour(float *out, int iters)            // matrix instructions on registers, nothing else
{
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + (threadIdx.x & 7)); b[i] = (__bf16)(1.0f + i); }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        for (int n = 0; n < 4; ++n) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
        for (int v = 0; v < 4; ++v) asm volatile("v_mul_f32 %0, 0.5, %0" : "+v"(acc[v]));
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[0] = acc[0];
}

template <bool SRC1>
__global__ __launch_bounds__(256) void checker(unsigned *wrong /* [64]: per lane */, int iters)
{
    unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 1u;
    for (int i = 0; i < iters; ++i) {
        s = s * 1664525u + 1013904223u; const float x0 = __uint_as_float(0x3f800000u | (s >> 9));
        s = s * 1664525u + 1013904223u; const float x1 = __uint_as_float(0x3f800000u | (s >> 9));
        s = s * 1664525u + 1013904223u; const float m0 = __uint_as_float(0x3f800000u | (s >> 9));
        s = s * 1664525u + 1013904223u; const float m1 = __uint_as_float(0x3f800000u | (s >> 9));
        const v2 x = {x0, x1}, m = {m0, m1};
        v2 got; float lo, hi;                                                      // wanted: (x0 * m1, x1 * m0)
        if (SRC1) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(got) : "v"(x), "v"(m));
        else      asm volatile("v_pk_mul_f32 %0, %2, %1 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(got) : "v"(x), "v"(m));   // same product, m as source 0
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(lo) : "v"(x0), "v"(m1));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(hi) : "v"(x1), "v"(m0));
        if (got.x != lo || got.y != hi) atomicAdd(&wrong[threadIdx.x & 63], 1u);
    }
}

template <bool SRC1>
static void run(const char *what, bool with_neighbour, float *dummy, unsigned *wrong)
{
    hipStream_t sa, sb; (void)hipStreamCreate(&sa); (void)hipStreamCreate(&sb);
    (void)hipMemset(wrong, 0, 256);
    if (with_neighbour) hipLaunchKernelGGL(neighbour, dim3(512), dim3(256), 0, sa, dummy, 200000);
    hipLaunchKernelGGL(checker<SRC1>, dim3(2048), dim3(256), 0, sb, wrong, 2000);
    (void)hipDeviceSynchronize();
    unsigned h[64]; (void)hipMemcpy(h, wrong, 256, hipMemcpyDeviceToHost);
    unsigned long q[4] = {0, 0, 0, 0};
    for (int l = 0; l < 64; ++l) q[l / 16] += h[l];
    printf("%-58s wrong results in lanes 0-15 / 16-31 / 32-47 / 48-63: %lu / %lu / %lu / %lu\n", what, q[0], q[1], q[2], q[3]);
    (void)hipStreamDestroy(sa); (void)hipStreamDestroy(sb);
}

int main()
{
    float *dummy; unsigned *wrong; (void)hipMalloc(&dummy, 4); (void)hipMalloc(&wrong, 256);
    run<true>("op_sel on source 1, GPU to itself:", false, dummy, wrong);
    run<true>("op_sel on source 1, next to v_mfma_f32_16x16x32_bf16:", true, dummy, wrong);
    run<false>("op_sel on source 0, next to v_mfma_f32_16x16x32_bf16:", true, dummy, wrong);
    return 0;
}
