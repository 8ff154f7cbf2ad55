// Global fp32 atomic-add rate on gfx950 by address pattern (measurement tool for the backward scatter,
// profiles/r04_backward_kernels.md).  The buffer is the size of the packed heat-map gradient (5 views x 2 x 128 x 240 pixels
// of 16 channels = 64 bytes each); every wave instruction adds to 4 "pixels" (16 lanes each) chosen by the pattern.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/global_atomic_bench.hip -o build_tools/global_atomic_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

enum { G_SCATTER4, G_ROW4, G_PAIR2x2, G_ONE16, G_ROW4_UNALIGNED, G_SQUARE, G_SAME, G_ROW4_INT64HALF, G_U32, G_U64, G_F64, G_PKBF16, G_U64_ALL, G_WG_XCD, G_WG_XCD_ROW, G_AGENT_XCD, G_STORE_XCD, NPAT };
static const char *pat_name[] = {"4 scattered pixels per instruction", "4 adjacent pixels, 256-byte aligned", "2 + 2: two 128-byte aligned pairs, pairs scattered",
                                 "1 pixel (16 lanes active)", "4 adjacent pixels, 64-byte aligned only", "2x2 square (x0, x0+1 on rows y0, y0+1), x0 any",
                                 "4 lane groups on the SAME pixel", "2x2 square, x0 even (pairs 128-byte aligned)",
                                 "u32 add, 4 scattered pixels", "u64 add, 4 scattered pixels x 8 lanes (32 lanes active)", "f64 add, 4 scattered pixels x 8 lanes (32 lanes active)",
                                 "pk_add_bf16, 4 scattered pixels x 8 lanes (32 lanes active)", "u64 add, 8 scattered pixels x 8 lanes (64 lanes active)",
                                 "WORKGROUP-scope f32 add, 4 scattered pixels inside the XCD's own eighth of the buffer", "WORKGROUP-scope f32 add, 4 adjacent pixels inside the XCD's own eighth",
                                 "agent-scope f32 add, 4 scattered pixels inside the XCD's own eighth", "plain 64-byte stores, 4 scattered pixels inside the XCD's own eighth"};

__device__ inline uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int PAT>
__global__ __launch_bounds__(64) void k(float *buf, int npix, int roww, int iters)
{
    const int lane = threadIdx.x, grp = lane >> 4, ch = lane & 15;
    uint32_t s = hash(blockIdx.x * 977u + 13u);
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
        s = hash(s + i);
        int px;
        if (PAT == G_SCATTER4) px = hash(s + grp) % npix;
        else if (PAT == G_ROW4) px = ((s % (npix / 4)) * 4) + grp;
        else if (PAT == G_PAIR2x2) px = ((hash(s + (grp >> 1)) % (npix / 2)) * 2) + (grp & 1);
        else if (PAT == G_ONE16) px = s % npix;
        else if (PAT == G_ROW4_UNALIGNED) px = (s % (npix - 4)) + grp;
        else if (PAT == G_SQUARE) px = (s % (npix - roww - 2)) + (grp & 1) + (grp >> 1) * roww;
        else if (PAT == G_ROW4_INT64HALF) px = ((s % ((npix - roww - 2) / 2)) * 2) + (grp & 1) + (grp >> 1) * roww;
        else px = s % npix;
        if (PAT == G_ONE16 && grp != 0) continue;
        if (PAT == G_U32) { px = hash(s + grp) % npix; atomicAdd(reinterpret_cast<unsigned *>(buf) + (size_t)px * 16 + ch, 1u); continue; }
        if (PAT == G_U64 || PAT == G_F64 || PAT == G_PKBF16) {
            px = hash(s + grp) % npix;
            if (ch >= 8) continue;
            if (PAT == G_U64) atomicAdd(reinterpret_cast<unsigned long long *>(buf) + (size_t)px * 8 + ch, 1ull);
            else if (PAT == G_F64) unsafeAtomicAdd(reinterpret_cast<double *>(buf) + (size_t)px * 8 + ch, 1.0);
            else {
                typedef short bf2 __attribute__((ext_vector_type(2)));
                bf2 v = {0x3f80, 0x3f80};
                __builtin_amdgcn_global_atomic_fadd_v2bf16(reinterpret_cast<__attribute__((address_space(1))) bf2 *>(reinterpret_cast<uintptr_t>(buf)) + (size_t)px * 16 + ch, v);
            }
            continue;
        }
        if (PAT == G_WG_XCD || PAT == G_WG_XCD_ROW || PAT == G_AGENT_XCD || PAT == G_STORE_XCD) {
            const int xcc = __builtin_amdgcn_s_getreg(63508) & 7, part = npix / 8;      // XCC_ID
            px = xcc * part + (PAT == G_WG_XCD_ROW ? (s % (part / 4)) * 4 + grp : hash(s + grp) % part);
            float *q = buf + (size_t)px * 16 + ch;
            if (PAT == G_AGENT_XCD) unsafeAtomicAdd(q, 1.0f);
            else if (PAT == G_STORE_XCD) __builtin_nontemporal_store((float)i, q);
            else __hip_atomic_fetch_add(q, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            continue;
        }
        if (PAT == G_U64_ALL) { px = hash(s + (lane >> 3)) % npix; atomicAdd(reinterpret_cast<unsigned long long *>(buf) + (size_t)px * 8 + (lane & 7), 1ull); continue; }
        unsafeAtomicAdd(buf + (size_t)px * 16 + ch, 1.0f);
    }
}

template <int PAT>
static void run(float *buf, int npix, int roww, int blocks, int iters)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<PAT><<<blocks, 64>>>(buf, npix, roww, 4);
    (void)hipEventRecord(e0);
    k<PAT><<<blocks, 64>>>(buf, npix, roww, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)blocks * iters, pix = insts * (PAT == G_ONE16 ? 1 : 4);
    printf("%-62s %8.1f us  %6.2f G wave-instructions/s  %6.2f G pixel-adds/s  (%5.1f shader cycles per instruction per CU at 2.4 GHz)\n",
           pat_name[PAT], ms * 1e3, insts / ms * 1e-6, pix / ms * 1e-6, ms * 1e-3 * 2.4e9 * 256 / insts);
}

int main()
{
    const int roww = 240, npix = 5 * 2 * 128 * 240;
    float *buf;
    (void)hipMalloc(&buf, (size_t)npix * 64);
    (void)hipMemset(buf, 0, (size_t)npix * 64);
    const int blocks = 256 * 32, iters = 640;      // 5.2 M wave instructions: the fine backward's count
    run<G_SCATTER4>(buf, npix, roww, blocks, iters);
    run<G_SQUARE>(buf, npix, roww, blocks, iters);
    run<G_ROW4_INT64HALF>(buf, npix, roww, blocks, iters);
    run<G_PAIR2x2>(buf, npix, roww, blocks, iters);
    run<G_ROW4>(buf, npix, roww, blocks, iters);
    run<G_ROW4_UNALIGNED>(buf, npix, roww, blocks, iters);
    run<G_ONE16>(buf, npix, roww, blocks, iters);
    run<G_SAME>(buf, npix, roww, blocks, iters);
    run<G_U32>(buf, npix, roww, blocks, iters);
    run<G_U64>(buf, npix, roww, blocks, iters);
    run<G_U64_ALL>(buf, npix, roww, blocks, iters);
    run<G_F64>(buf, npix, roww, blocks, iters);
    run<G_PKBF16>(buf, npix, roww, blocks, iters);
    run<G_AGENT_XCD>(buf, npix, roww, blocks, iters);
    run<G_WG_XCD>(buf, npix, roww, blocks, iters);
    run<G_WG_XCD_ROW>(buf, npix, roww, blocks, iters);
    run<G_STORE_XCD>(buf, npix, roww, blocks, iters);
    printf("fewer workgroups (the same work per workgroup): CU-side or L2-side limit?\n");
    for (int nb : {64, 256, 1024, 2048}) { printf("  %5d workgroups: ", nb); run<G_SCATTER4>(buf, npix, roww, nb, iters); }
    for (int nb : {64, 256, 1024, 2048}) { printf("  %5d workgroups: ", nb); run<G_U64_ALL>(buf, npix, roww, nb, iters); }
    return 0;
}
