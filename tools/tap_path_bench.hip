// Where do gathered taps arrive fastest?  global_load_dwordx4 into VGPRs against global_load_lds_dwordx4 (LDS-DMA: the
// texture path writes LDS, no VGPR return) for the unprojection's tap pattern: a wave instruction fetches 16 "pixels" of
// 64 bytes (4 lanes x 16 bytes each) scattered over a heat-map-sized region.  (profiles/r04_issue_model.md, section 4)
//   hipcc --offload-arch=gfx950 -O3 tools/tap_path_bench.hip -o build_tools/tap_path_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ inline uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__device__ __forceinline__ void lds_dma16(uint32_t voff, uint32_t lds_dst, const char *gbase)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_dst), "s"(gbase) : "memory");
}

// MODE 0: loads into VGPRs, summed.  MODE 1: LDS-DMA, never read back.  MODE 2: LDS-DMA, then ds_read_b128 of the lane's own 16 bytes.
// MODE 3: as 0, but a wave-load = 8 voxels x the two x-neighbours of a tap pair (128 contiguous bytes, 64-byte aligned start)
template <int MODE>
__global__ __launch_bounds__(64) void k(const float4 *__restrict__ src, float *out, int npix, int iters, int lds_bytes_per_wave)
{
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const int lane = threadIdx.x;
    uint32_t s = hash(blockIdx.x * 977u + 13u);
    float4 acc = {0, 0, 0, 0};
    const char *gbase = reinterpret_cast<const char *>(src);
    const uint32_t lds0 = (uint32_t)(size_t)sm;
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
        // 8 wave-loads in flight, as half a view's taps of the brick kernel
        uint32_t off[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s = hash(s + j);
            if (MODE == 3) {
                const uint32_t px = hash(s + (lane >> 3)) % (uint32_t)(npix - 1);   // 8 pixel pairs per instruction
                off[j] = px * 64u + (lane & 7) * 16u;
            } else {
                const uint32_t px = hash(s + (lane >> 2)) % (uint32_t)npix;          // 16 pixels per instruction
                off[j] = px * 64u + (lane & 3) * 16u;
            }
        }
        if (MODE == 0 || MODE == 3) {
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float4 *>(gbase + off[j]);
#pragma unroll
            for (int j = 0; j < 8; ++j) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) lds_dma16(off[j], lds0 + j * 1024, gbase);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (MODE == 2) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float4 v = *reinterpret_cast<const float4 *>(sm + j * 1024 + lane * 16);
                    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                }
            }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}

// Round 5 (review item: footprint-major bf16 layout).  One "slot" = the taps of 32 voxels in one view.
// FOOT 0: today's bf16 maps, two lanes per 32-byte pixel: 4 wave-loads per slot (the 2x2 taps: pixel p, p+1, p+W, p+W+1 of
//         each voxel) - 32 scattered 32-byte pieces per wave-load, 2-4 distinct 128-byte lines per voxel.
// FOOT 1: footprint-major records: per tap origin one 128-byte record holding the whole 2x2 footprint of all 16 channels
//         (channel pair major: lane l of 8 reads channels 2l, 2l+1 of all four taps) - 4 wave-loads per slot of 8 voxels
//         each, every wave-load = 8 whole cache lines.
// Same bytes into the VGPRs (4 KB per slot), same number of wave-loads; only the number of lines per wave-load differs.
template <int FOOT>
__global__ __launch_bounds__(64) void kfoot(const uint4 *__restrict__ src, float *out, int npix, int W, int iters)
{
    const int lane = threadIdx.x;
    uint32_t s = hash(blockIdx.x * 977u + 13u);
    uint4 acc = {0, 0, 0, 0};
    const char *gbase = reinterpret_cast<const char *>(src);
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
        uint32_t off[8];                  // two slots in flight = 8 wave-loads, as in the brick kernel
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            s = hash(s + sl);
            if (FOOT == 0) {
                const uint32_t px = hash(s + (lane >> 1)) % (uint32_t)(npix - W - 1);      // 32 voxels per wave-load
                const uint32_t b = px * 32u + (lane & 1) * 16u;
                off[4 * sl + 0] = b; off[4 * sl + 1] = b + 32u; off[4 * sl + 2] = b + (uint32_t)W * 32u; off[4 * sl + 3] = b + (uint32_t)W * 32u + 32u;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {                                              // 8 voxels per wave-load
                    const uint32_t px = hash(s + j * 131u + (lane >> 3)) % (uint32_t)npix;
                    off[4 * sl + j] = px * 128u + (lane & 7) * 16u;
                }
            }
        }
        uint4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const uint4 *>(gbase + off[j]);
#pragma unroll
        for (int j = 0; j < 8; ++j) { acc.x += v[j].x; acc.y ^= v[j].y; acc.z += v[j].z; acc.w ^= v[j].w; }
    }
    if (acc.x + acc.y + acc.z + acc.w == 123456u) out[0] = (float)acc.x;
}

template <int FOOT>
static void run_foot(const char *name, const uint4 *src, float *out, int npix, int W, int waves, int iters)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    kfoot<FOOT><<<waves, 64>>>(src, out, npix, W, 2);
    (void)hipEventRecord(e0);
    kfoot<FOOT><<<waves, 64>>>(src, out, npix, W, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double slots = (double)waves * iters * 2;
    printf("%-64s %6d waves  %8.1f us  %6.1f cycles per slot of 32 voxel-views per CU (2.4 GHz)  %5.1f per wave-load\n", name, waves,
           ms * 1e3, ms * 1e-3 * 2.4e9 * 256 / slots, ms * 1e-3 * 2.4e9 * 256 / (slots * 4));
}

template <int MODE>
static void run(const char *name, const float4 *src, float *out, int npix, int waves, int iters)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<waves, 64, 8192>>>(src, out, npix, 2, 8192);
    (void)hipEventRecord(e0);
    k<MODE><<<waves, 64, 8192>>>(src, out, npix, iters, 8192);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double loads = (double)waves * iters * 8;
    printf("%-58s %6d waves  %8.1f us  %6.2f G wave-loads/s  %5.1f cycles per wave-load per CU (2.4 GHz)  %5.2f TB/s into the CUs\n", name, waves,
           ms * 1e3, loads / ms * 1e-6, ms * 1e-3 * 2.4e9 * 256 / loads, loads * 1024 / ms * 1e-9);
}

int main()
{
    for (int npix : {4 * 128 * 240, 256}) {        // one view of 4 samples (7.9 MB: L2 / Infinity Cache), or 16 KB (L1-resident)
        float4 *src; float *out;
        (void)hipMalloc(&src, (size_t)npix * 64); (void)hipMemset(src, 0, (size_t)npix * 64); (void)hipMalloc(&out, 4);
        printf("region %d pixels of 64 bytes\n", npix);
        for (int waves : {4096, 16384}) {
            run<0>("global_load_dwordx4 -> VGPRs", src, out, npix, waves, 200);
            run<1>("global_load_lds_dwordx4 -> LDS (not read back)", src, out, npix, waves, 200);
            run<2>("global_load_lds_dwordx4 -> LDS, ds_read_b128 of own 16 bytes", src, out, npix, waves, 200);
            run<3>("global_load_dwordx4 -> VGPRs, 8 voxels x an x-pair (128 B runs)", src, out, npix, waves, 200);
        }
        (void)hipFree(src); (void)hipFree(out);
    }
    // bf16: one view of one sample (128 x 240 pixels) as 32-byte pixels (0.98 MB) against 128-byte footprint records (3.9 MB);
    // the same for the part of a view one 64^3 person cube projects to (~64 x 64 pixels), and an L1-sized region
    for (int hw : {128 * 240, 64 * 64, 128}) {
        const int W = hw == 128 * 240 ? 240 : (hw == 64 * 64 ? 64 : 8);
        uint4 *src; float *out;
        (void)hipMalloc(&src, (size_t)hw * 128); (void)hipMemset(src, 0, (size_t)hw * 128); (void)hipMalloc(&out, 4);
        printf("bf16 taps, region of %d pixels (%d KB as 32-byte pixels, %d KB as footprint records)\n", hw, hw * 32 / 1024, hw * 128 / 1024);
        for (int waves : {4096, 16384}) {
            run_foot<0>("bf16 pixels, two lanes per pixel: 4 taps = 4 wave-loads of 32 voxels", src, out, hw, W, waves, 200);
            run_foot<1>("footprint records: 4 wave-loads of 8 voxels, one line per voxel", src, out, hw, W, waves, 200);
        }
        (void)hipFree(src); (void)hipFree(out);
    }
    return 0;
}
