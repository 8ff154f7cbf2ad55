// LDS float-atomic rate on gfx950 (measurement tool for the backward kernel's LDS patch, profiles/r04_backward_kernels.md).
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/lds_atomic_bench.hip -o build_tools/lds_atomic_bench
// One workgroup of W waves per CU slot; every wave issues ITERS x 4 LDS operations with one of the address patterns
// below; cycles per wave-instruction from s_memtime of wave 0, and the chip-wide time from HIP events.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

enum { P_DISTINCT, P_SAME4, P_NEIGH4, P_FAR4, P_SAME64, P_WRITE, P_READ, P_INT, P_U64, P_F64, NPAT };
static const char *pat_name[] = {"ds_add_f32, 64 distinct consecutive words", "ds_add_f32, 4 lane groups on the SAME 16 words",
                                 "ds_add_f32, 4 lane groups on 4 neighbouring pixels (16 words each, consecutive)",
                                 "ds_add_f32, 4 lane groups 21 pixels apart (bank-aliased pixels)", "ds_add_f32, all 64 lanes one word",
                                 "ds_write_b32, 64 distinct consecutive words", "ds_read_b32, 64 distinct consecutive words",
                                 "ds_add_u32, 4 lane groups on the SAME 16 words",
                                 "ds_add_u64, 4 lane groups on the SAME 16 double-words", "ds_add_f64, 4 lane groups on the SAME 16 double-words"};

template <int PAT>
__global__ __launch_bounds__(256) void k(float *out, long long *cyc, int iters)
{
    extern __shared__ float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = lane >> 4, ch = lane & 15;
    for (int e = tid; e < 8192; e += 256) sm[e] = 0.0f;
    __syncthreads();
    int a;
    if (PAT == P_DISTINCT || PAT == P_WRITE || PAT == P_READ) a = lane;
    else if (PAT == P_SAME4 || PAT == P_INT) a = ch;
    else if (PAT == P_U64 || PAT == P_F64) a = 2 * ch;
    else if (PAT == P_NEIGH4) a = grp * 16 + ch;
    else if (PAT == P_FAR4) a = grp * 21 * 16 * 4 / 4 * 1 + ch + grp * 64 * 5;   // multiples of 64 words apart: same banks
    else a = 0;
    a += wave * 2048;
    float *p = sm + a;
    float acc = 0.0f;
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
        const float v = (float)(i & 3) + 1.0f;
        if (PAT == P_WRITE) {
            p[0] = v; p[256] = v; p[512] = v; p[768] = v;
            asm volatile("" ::: "memory");
        } else if (PAT == P_READ) {
            acc += p[0] + p[256] + p[512] + p[768];
            asm volatile("" ::: "memory");
        } else if (PAT == P_INT) {
            atomicAdd(reinterpret_cast<unsigned *>(p), 1u); atomicAdd(reinterpret_cast<unsigned *>(p + 256), 1u);
            atomicAdd(reinterpret_cast<unsigned *>(p + 512), 1u); atomicAdd(reinterpret_cast<unsigned *>(p + 768), 1u);
        } else if (PAT == P_U64) {
            unsigned long long *q = reinterpret_cast<unsigned long long *>(p);
            atomicAdd(q, 1ull); atomicAdd(q + 128, 1ull); atomicAdd(q + 256, 1ull); atomicAdd(q + 384, 1ull);
        } else if (PAT == P_F64) {
            double *q = reinterpret_cast<double *>(p);
            atomicAdd(q, 1.0); atomicAdd(q + 128, 1.0); atomicAdd(q + 256, 1.0); atomicAdd(q + 384, 1.0);
        } else {
            atomicAdd(p, v); atomicAdd(p + 256, v); atomicAdd(p + 512, v); atomicAdd(p + 768, v);
        }
    }
    __syncthreads();
    const long long t1 = __builtin_readcyclecounter();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    float s = acc;
    for (int e = tid; e < 8192; e += 256) s += sm[e];
    if (s == 123.456f) out[0] = s;
}

template <int PAT>
static void run(int blocks, int iters, float *out, long long *cyc)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<PAT><<<blocks, 256, 8192 * 4>>>(out, cyc, 10);
    hipEventRecord(e0);
    k<PAT><<<blocks, 256, 8192 * 4>>>(out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : h) mean += (double)v;
    mean /= blocks;
    const double wave_insts = (double)iters * 4;
    printf("%-82s blocks %5d  %8.1f us  %7.1f shader cycles per wave-instruction (4 waves of a workgroup interleaved: x1/4 per CU)\n",
           pat_name[PAT], blocks, ms * 1e3, mean / wave_insts);
}

int main()
{
    float *out; long long *cyc;
    hipMalloc(&out, 4); hipMalloc(&cyc, 8192 * 8);
    for (int blocks : {1, 256}) {
        run<P_DISTINCT>(blocks, 2000, out, cyc); run<P_SAME4>(blocks, 2000, out, cyc); run<P_NEIGH4>(blocks, 2000, out, cyc);
        run<P_FAR4>(blocks, 2000, out, cyc); run<P_SAME64>(blocks, 2000, out, cyc); run<P_WRITE>(blocks, 2000, out, cyc);
        run<P_READ>(blocks, 2000, out, cyc); run<P_INT>(blocks, 2000, out, cyc);
        run<P_U64>(blocks, 2000, out, cyc); run<P_F64>(blocks, 2000, out, cyc);
    }
    return 0;
}
