// sp3d_unproject.hip - gfx950 kernels for ProjectLayer.get_voxel (forward, backward) and the
// heat-map re-tiling pass, plus their C-ABI entry points (include/sp3d.h).
//
// Reference path: /root/reference/lib/models/project_layer.py:42-102 and its helpers
// lib/utils/cameras.py:27-55, lib/utils/transforms.py:119-123.  The reference runs this as a
// Python batch x view loop of ~90 tiny kernels; here one launch covers the whole batch.
//
// Kernels
//   unproject_planar_kernel  lane = voxel, heat-maps in the reference's planar (B,J,h,w)
//                            layout; simple, exact, gather of 4*J scattered dwords / view.
//   pack_nhwc_kernel         (B,J,h,w) x V  ->  (V,B,h,w,Jp): LDS-tiled transpose so that a
//                            bilinear tap becomes ONE contiguous Jp*4-byte read.
//   unproject_nhwc_kernel    the fast path: phase 1 (lane = voxel) projects the tile's
//                            voxels through every camera and stages the sample positions in
//                            LDS; phase 2 (4 lanes = one voxel, each lane one 16-byte channel
//                            quad) gathers the taps with dwordx4 loads whose 4-lane groups
//                            read 64 contiguous bytes; phase 3 stores the (J, tile) result
//                            through LDS as coalesced dwordx4 rows.
//   unproject_bwd_kernel     recomputes the forward value (clamp mask) and scatters
//                            g * w_tap with hardware fp32 atomics.
#include <type_traits>

#include <string.h>
#include "sp3d_device.h"
#include "sp3d_proj_pk.h"
#include "sp3d_twiddles.h"

namespace sp3d {

// measurement only (tools/wave_timeline.py): when non-null the pipelined kernel stores s_memtime stamps
// per wave: [start, after P1(0), after view 0..V-1, end] (18 slots per wave)
#ifdef SP3D_TIMELINE        // 1: wave start / end only (light), 2: + per-view stamps
__device__ unsigned long long *g_timeline = nullptr;
#define SP3D_STAMP_ALWAYS(slot) do { if (tl && lane == 0) tl[slot] = __builtin_readcyclecounter(); } while (0)
#if SP3D_TIMELINE >= 2
#define SP3D_STAMP(slot) SP3D_STAMP_ALWAYS(slot)
#else
#define SP3D_STAMP(slot) do { } while (0)
#endif
#else
#define SP3D_STAMP(slot) do { } while (0)
#define SP3D_STAMP_ALWAYS(slot) do { } while (0)
#endif

// measurement only (tools/diag_ablate.py builds one library per -DSP3D_ABLATE=<mask>, never shipped): compile-time
// switches that REMOVE one part of the pipelined kernels (results are then wrong on purpose) to see how the parts
// compose in time.  1: no result stores  2: no tap loads (FMAs run on zeros)  4: no projection (synthetic tap
// records)  8: staggered start (s_sleep by wave slot)  16: no FMAs
#ifndef SP3D_ABLATE
#define SP3D_ABLATE 0
#endif
#define SP3D_DIAG_ON(bit) ((SP3D_ABLATE) & (bit))
#define SP3D_DIAG_FLAGS() do { } while (0)
#if SP3D_ABLATE
#define SP3D_DIAG
#endif

constexpr int TILE = 256; // voxels per workgroup (= threads per workgroup)

// ------------------------------------------------------------------------------------------
// planar-layout forward: lane = voxel.  JC = channels accumulated per pass.
// ------------------------------------------------------------------------------------------
template <int JC>
__global__ __launch_bounds__(TILE) void unproject_planar_kernel(Views hm, const float *__restrict__ cam,
                                                               const float *__restrict__ centers,
                                                               const uint8_t *__restrict__ valid,
                                                               float *__restrict__ cubes, float *__restrict__ grids,
                                                               Geom g)
{
    const int b = blockIdx.y;
    const int bs = g.sample_of ? g.sample_of[b] : b;   // row of the heat-map batch / camera table this cube reads
    const int n = blockIdx.x * TILE + threadIdx.x;
    if (n >= g.N) return;
    float *cb = cubes + (size_t)b * g.J * g.N;
    if (!valid[b]) { // project_layer.py:48,51,54 - skipped sample stays zero
        for (int j = 0; j < g.J; ++j) cb[(size_t)j * g.N + n] = 0.0f;
        if (grids) {
            float *gp = grids + ((size_t)b * g.N + n) * 3;
            gp[0] = 0.0f; gp[1] = 0.0f; gp[2] = 0.0f;
        }
        return;
    }
    const int vx = n / g.YZ, rem = n - vx * g.YZ, vy = rem / g.Z, vz = rem - vy * g.Z;
    const float x = linspace_at(g.Lx, g.X, vx) + centers[3 * b + 0];
    const float y = linspace_at(g.Ly, g.Y, vy) + centers[3 * b + 1];
    const float z = linspace_at(g.Lz, g.Z, vz) + centers[3 * b + 2];
    if (grids) {
        float *gp = grids + ((size_t)b * g.N + n) * 3;
        gp[0] = x; gp[1] = y; gp[2] = z;
    }
    const float W_in = (float)g.W_in, H_in = (float)g.H_in;
    const size_t plane = (size_t)g.h * g.w;
    for (int j0 = 0; j0 < g.J; j0 += JC) {
        float acc[JC];
#pragma unroll
        for (int k = 0; k < JC; ++k) acc[k] = 0.0f;
        float cnt = 0.0f;
        bool bad = false;
        for (int c = 0; c < g.V; ++c) {
            const float *cm = cam + ((size_t)bs * g.V + c) * SP3D_CAM_STRIDE;
            float ix, iy;
            const bool bound = sample_pos(cm, x, y, z, g.w, g.h, W_in, H_in, ix, iy);
            cnt += bound ? 1.0f : 0.0f;
            if (ix != ix || iy != iy) { bad = true; continue; } // NaN sample -> NaN -> 0 (project_layer.py:98)
            if (!bound) continue;                                // val * 0
            const Bilin bl = bilin(ix, iy);
            const bool x0ok = bl.x0 >= 0 && bl.x0 <= g.w - 1, x1ok = bl.x0 + 1 >= 0 && bl.x0 + 1 <= g.w - 1;
            const bool y0ok = bl.y0 >= 0 && bl.y0 <= g.h - 1, y1ok = bl.y0 + 1 >= 0 && bl.y0 + 1 <= g.h - 1;
            const float *base = hm.p[c] + ((size_t)bs * g.J + j0) * plane + (ptrdiff_t)bl.y0 * g.w + bl.x0;
#pragma unroll
            for (int k = 0; k < JC; ++k) {
                if (j0 + k < g.J) {
                    const float *pl = base + (size_t)k * plane;
                    const float t00 = (x0ok && y0ok) ? pl[0] : 0.0f;
                    const float t10 = (x1ok && y0ok) ? pl[1] : 0.0f;
                    const float t01 = (x0ok && y1ok) ? pl[g.w] : 0.0f;
                    const float t11 = (x1ok && y1ok) ? pl[g.w + 1] : 0.0f;
                    float v = t00 * bl.wnw;
                    v = fmaf(t10, bl.wne, v);
                    v = fmaf(t01, bl.wsw, v);
                    v = fmaf(t11, bl.wse, v);
                    acc[k] = acc[k] + v;
                }
            }
        }
        const float den = cnt + 1e-6f;
#pragma unroll
        for (int k = 0; k < JC; ++k)
            if (j0 + k < g.J) cb[(size_t)(j0 + k) * g.N + n] = bad ? 0.0f : fuse(acc[k], den);
    }
}

// ------------------------------------------------------------------------------------------
// (B,J,h,w) x V  ->  (V,B,h,w,JP) re-tiling.  One workgroup = 256 pixels of one (view,sample).
// ------------------------------------------------------------------------------------------
constexpr int PSTR = 260; // LDS row stride (floats): rows 16-B aligned, <=2-way write conflicts

template <int JP, typename TI = float, typename TO = float>
__global__ __launch_bounds__(256) void pack_nhwc_kernel(Views hm, float *__restrict__ packed_, int B, int J, int HW)
{
    __shared__ float tile[JP][PSTR];
    const int tid = threadIdx.x;
    const int p0 = blockIdx.x * 256;
    const int b = blockIdx.y, v = blockIdx.z;
    const TI *src = reinterpret_cast<const TI *>(hm.p[v]) + (size_t)b * J * HW;
    TO *packed = reinterpret_cast<TO *>(packed_);
    const int p = p0 + tid;
    // all J plane loads in flight before the first LDS write: with the load inside `if (p < HW)` the compiler emitted
    // branch -> load -> s_waitcnt vmcnt(0) -> ds_write per channel, JP dependent round trips per workgroup.  The pixel index is
    // clamped instead (a lane past the end re-reads the last pixel and writes zero).
    const int pc = p < HW ? p : HW - 1;
    float vals[JP];
    // channel index clamped too (planes j >= J re-read plane J - 1 and are zeroed below): straight-line code, no branch between
    // the loads - behind a wave-uniform `j < J` branch the bf16 variant still waited for every load before widening it
    if constexpr (sizeof(TI) == 2) {
        uint32_t raw[JP];
#pragma unroll
        for (int j = 0; j < JP; ++j) raw[j] = (uint32_t)reinterpret_cast<const uint16_t *>(src)[(size_t)min(j, J - 1) * HW + pc];
#pragma unroll
        for (int j = 0; j < JP; ++j) vals[j] = j < J ? __uint_as_float(raw[j] << 16) : 0.0f;
    } else {
#pragma unroll
        for (int j = 0; j < JP; ++j) vals[j] = reinterpret_cast<const float *>(src)[(size_t)min(j, J - 1) * HW + pc];
#pragma unroll
        for (int j = 0; j < JP; ++j) vals[j] = j < J ? vals[j] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < JP; ++j) tile[j][tid] = p < HW ? vals[j] : 0.0f;
    __syncthreads();
    constexpr int NQ = JP / 4;
    TO *dst = packed + (((size_t)v * B + b) * HW + p0) * JP;
    for (int e = tid; e < 256 * NQ; e += 256) {
        const int px = e / NQ, q = e - px * NQ;
        if (p0 + px < HW) {
            float4 o;
            o.x = tile[4 * q + 0][px]; o.y = tile[4 * q + 1][px];
            o.z = tile[4 * q + 2][px]; o.w = tile[4 * q + 3][px];
            Store4<TO>::store(dst + (size_t)px * JP + 4 * q, o);
        }
    }
}

// ------------------------------------------------------------------------------------------
// channels-last forward (the hot kernel).
//   JP   = floats per pixel (channel stride), multiple of 4, <= 16 per pass
//   LDS  = sIx,sIy [V][TILE] sample positions, sMask[TILE] bound bits (+bit31 NaN flag),
//          sOut [JP][OSTR] result tile
// ------------------------------------------------------------------------------------------
constexpr int OSTR = 260;

template <int JP, bool XCD, int U>
__global__ __launch_bounds__(TILE) void unproject_nhwc_kernel(Views hm, const float *__restrict__ cam,
                                                             const float *__restrict__ centers,
                                                             const uint8_t *__restrict__ valid,
                                                             float *__restrict__ cubes, float *__restrict__ grids,
                                                             Geom g, int tiles_per_sample, int total_tiles)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *sOut = smem;                                   // [JP][OSTR]
    float *sIx = sOut + JP * OSTR;                        // [V][TILE]
    float *sIy = sIx + g.V * TILE;                        // [V][TILE]
    uint32_t *sMask = reinterpret_cast<uint32_t *>(sIy + g.V * TILE); // [TILE]

    int b, tile;
    if (XCD) {
        if (!xcd_map(blockIdx.x, g.B, tiles_per_sample, g.xcd_chunk, b, tile)) return;
    } else {
        b = blockIdx.x / tiles_per_sample;
        tile = blockIdx.x - b * tiles_per_sample;
    }
    (void)total_tiles;
    const int bs = g.sample_of ? g.sample_of[b] : b;
    const int n0 = tile * TILE;
    const int tid = threadIdx.x;
    const int nvox = min(TILE, g.N - n0);
    float *cb = cubes + (size_t)b * g.J * g.N;
    constexpr int NQ = JP / 4;

    if (!valid[b]) { // skipped sample: zeros (project_layer.py:48,51,54)
        for (int j = 0; j < g.J; ++j)
            if (tid < nvox) cb[(size_t)j * g.N + n0 + tid] = 0.0f;
        if (grids && tid < nvox) {
            float *gp = grids + ((size_t)b * g.N + n0 + tid) * 3;
            gp[0] = 0.0f; gp[1] = 0.0f; gp[2] = 0.0f;
        }
        return;
    }

    // ---- phase 1: lane = voxel; project through every camera, stage sample positions
    {
        const int n = n0 + tid;
        uint32_t mask = 0;
        if (tid < nvox) {
            const int vx = n / g.YZ, rem = n - vx * g.YZ, vy = rem / g.Z, vz = rem - vy * g.Z;
            const float x = linspace_at(g.Lx, g.X, vx) + centers[3 * b + 0];
            const float y = linspace_at(g.Ly, g.Y, vy) + centers[3 * b + 1];
            const float z = linspace_at(g.Lz, g.Z, vz) + centers[3 * b + 2];
            if (grids) {
                float *gp = grids + ((size_t)b * g.N + n) * 3;
                gp[0] = x; gp[1] = y; gp[2] = z;
            }
            const float W_in = (float)g.W_in, H_in = (float)g.H_in;
            for (int c = 0; c < g.V; ++c) {
                const float *cm = cam + ((size_t)bs * g.V + c) * SP3D_CAM_STRIDE;
                float ix, iy;
                const bool bound = sample_pos(cm, x, y, z, g.w, g.h, W_in, H_in, ix, iy);
                if (bound) mask |= (1u << c);
                if (ix != ix || iy != iy) mask |= 0x80000000u;
                sIx[c * TILE + tid] = ix;
                sIy[c * TILE + tid] = iy;
            }
        }
        sMask[tid] = mask;
    }
    __syncthreads();

    // ---- phase 2: 4 lanes = one voxel; lane q owns channels [4q, 4q+4).  U voxels are in
    //      flight per lane (4*U dwordx4 loads issued back to back before the first use).
    {
        constexpr int LPV = 4;                 // lanes per voxel
        constexpr int GROUPS = TILE / LPV;     // 64 voxel groups per workgroup
        constexpr int VPG = TILE / GROUPS;     // 4 voxels per group
        const int grp = tid / LPV, q = tid % LPV;
        const bool qact = q < NQ;              // JP < 16: upper lanes idle
        const size_t rowf = (size_t)g.w * JP;  // floats per heat-map row
#pragma unroll 1
        for (int i0 = 0; i0 < VPG; i0 += U) {
            float acc[U][4];
            uint32_t msk[U];
            uint32_t any = 0;
#pragma unroll
            for (int i = 0; i < U; ++i) {
                msk[i] = sMask[(i0 + i) * GROUPS + grp];
                if (msk[i] & 0x80000000u) msk[i] = 0x80000000u;   // NaN position: voxel is zero, skip gathers
                any |= msk[i];
                acc[i][0] = acc[i][1] = acc[i][2] = acc[i][3] = 0.0f;
            }
            uint32_t cnt[U];
#pragma unroll
            for (int i = 0; i < U; ++i) cnt[i] = sMask[(i0 + i) * GROUPS + grp];
#pragma unroll 1
            for (int c = 0; c < g.V; ++c) {
                if (!__any((any >> c) & 1u)) continue;         // wave-uniform skip
                const float *vb = hm.p[c] + (size_t)bs * g.h * rowf + 4 * q;
                float4 t00[U], t10[U], t01[U], t11[U];
                float wnw[U], wne[U], wsw[U], wse[U];
                // Branch-free gather: every lane always loads.  A tap outside the heat-map (zeros
                // padding) or a lane whose voxel is not in view c gets weight 0 and a clamped /
                // parked address (pixel (0,0): all parked lanes hit one cache line).
#pragma unroll
                for (int i = 0; i < U; ++i) {
                    const bool on = qact && ((msk[i] >> c) & 1u);
                    const int t = (i0 + i) * GROUPS + grp;
                    const Bilin bl = bilin(sIx[c * TILE + t], sIy[c * TILE + t]);
                    const bool x0ok = on && bl.x0 >= 0 && bl.x0 <= g.w - 1;
                    const bool x1ok = on && bl.x0 + 1 >= 0 && bl.x0 + 1 <= g.w - 1;
                    const bool y0ok = bl.y0 >= 0 && bl.y0 <= g.h - 1;
                    const bool y1ok = bl.y0 + 1 >= 0 && bl.y0 + 1 <= g.h - 1;
                    wnw[i] = (x0ok && y0ok) ? bl.wnw : 0.0f;
                    wne[i] = (x1ok && y0ok) ? bl.wne : 0.0f;
                    wsw[i] = (x0ok && y1ok) ? bl.wsw : 0.0f;
                    wse[i] = (x1ok && y1ok) ? bl.wse : 0.0f;
                    const int xa = on ? min(max(bl.x0, 0), g.w - 1) : 0, xb = on ? min(max(bl.x0 + 1, 0), g.w - 1) : 0;
                    const int ya = on ? min(max(bl.y0, 0), g.h - 1) : 0, yb = on ? min(max(bl.y0 + 1, 0), g.h - 1) : 0;
                    const float *ra = vb + (size_t)ya * rowf, *rb = vb + (size_t)yb * rowf;
                    t00[i] = *reinterpret_cast<const float4 *>(ra + xa * JP);
                    t10[i] = *reinterpret_cast<const float4 *>(ra + xb * JP);
                    t01[i] = *reinterpret_cast<const float4 *>(rb + xa * JP);
                    t11[i] = *reinterpret_cast<const float4 *>(rb + xb * JP);
                }
#pragma unroll
                for (int i = 0; i < U; ++i) {
                    float v;
                    v = t00[i].x * wnw[i]; v = fmaf(t10[i].x, wne[i], v); v = fmaf(t01[i].x, wsw[i], v); v = fmaf(t11[i].x, wse[i], v); acc[i][0] = acc[i][0] + v;
                    v = t00[i].y * wnw[i]; v = fmaf(t10[i].y, wne[i], v); v = fmaf(t01[i].y, wsw[i], v); v = fmaf(t11[i].y, wse[i], v); acc[i][1] = acc[i][1] + v;
                    v = t00[i].z * wnw[i]; v = fmaf(t10[i].z, wne[i], v); v = fmaf(t01[i].z, wsw[i], v); v = fmaf(t11[i].z, wse[i], v); acc[i][2] = acc[i][2] + v;
                    v = t00[i].w * wnw[i]; v = fmaf(t10[i].w, wne[i], v); v = fmaf(t01[i].w, wsw[i], v); v = fmaf(t11[i].w, wse[i], v); acc[i][3] = acc[i][3] + v;
                }
            }
            if (qact) {
#pragma unroll
                for (int i = 0; i < U; ++i) {
                    const int t = (i0 + i) * GROUPS + grp;
                    const bool bad = (cnt[i] & 0x80000000u) != 0;
                    const float den = (float)__popc(cnt[i] & 0x7fffffffu) + 1e-6f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) sOut[(4 * q + k) * OSTR + t] = bad ? 0.0f : fuse(acc[i][k], den);
                }
            }
        }
    }
    __syncthreads();

    // ---- phase 3: coalesced store of the (J, tile) block, 16 B per lane where aligned
    if (((g.N & 3) == 0) && nvox == TILE) {
        for (int e = tid; e < g.J * (TILE / 4); e += TILE) {
            const int j = e / (TILE / 4), u = e - j * (TILE / 4);
            const float4 o = *reinterpret_cast<const float4 *>(&sOut[j * OSTR + 4 * u]);
            *reinterpret_cast<float4 *>(cb + (size_t)j * g.N + n0 + 4 * u) = o;
        }
    } else {
        for (int j = 0; j < g.J; ++j)
            if (tid < nvox) cb[(size_t)j * g.N + n0 + tid] = sOut[j * OSTR + tid];
    }
}

// ------------------------------------------------------------------------------------------
// channels-last forward, software-pipelined per wave ("pipe" kernel).
//
// A wave owns 64 consecutive voxels and never synchronises with the other waves of its
// workgroup.  For every view c it alternates two lane mappings:
//   P1(c)   lane l = voxel l : project through camera c, reduce the sample position to one
//           record {offset of the 2x2 tap block, 4 slot weights} staged in the wave's LDS slice
//           (double buffered by view parity).  Taps outside the heat-map (zeros padding) and
//           voxels not seen by camera c become zero WEIGHTS on in-range addresses, so the gather
//           is branch free; the 2x2 block is clamped inside the image and the weights move to
//           the slot whose pixel they belong to (order of the non-zero terms of ATen's
//           bilinear FMA chain is preserved => same bits as the oracle).
//   G(c)    lane (g,q) = voxels {g, 16+g, 32+g, 48+g}, channel quad q : 16 dwordx4 loads (each
//           4-lane group reads 64 contiguous bytes) issued back to back, THEN P1(c+1) runs on the
//           VALU while they are in flight, then the 64 FMAs of view c.
// The result tile goes through the wave's LDS slice once and leaves as dwordx4 rows.
// ------------------------------------------------------------------------------------------
constexpr int WREC = 2 * 5 * 64;           // floats: weights [buf][voxel][4] (16-byte records), then offsets [buf][voxel]
constexpr int WOFF = 2 * 4 * 64;           // first offset word
constexpr int WOSTR = 68;                  // sOut row stride (floats), rows 16-B aligned

struct Rec {
    int off;
    float w00, w10, w01, w11;
};

// scalar form of make_record_pk (sp3d_proj_pk.h), used by the backward scatter kernel
// ESZ: the record's offset is in units of 1/ESZ elements (ESZ = sizeof(element) gives byte offsets)
template <int JP, int ESZ = 1>
__device__ __forceinline__ Rec make_record(bool use, float ix, float iy, int w, int h)
{
    const RecPk p = make_record_pk(use, v2f{ix, iy}, w, h);
    Rec r;
    r.off = (p.y0 * w + p.x0) * (JP * ESZ);
    r.w00 = p.wt.x; r.w10 = p.wt.y; r.w01 = p.wb.x; r.w11 = p.wb.y;
    return r;
}

// The view loop shared by the pipelined kernels: P1 (lane = voxel) and G (lane = (voxel-of-4, channel quad)) for the 64
// voxels of this wave; `x,y,z` is this lane's voxel centre, `inb` whether the lane has a voxel at all.  On return
// acc[i][k] holds sum over views of the bilinear samples of voxel slot 16*i + lane/4, channel 4*(lane%4) + k, and
// mymask = number of views that see the lane's own voxel (+ bit 31: NaN sample position).
// Round 3: the projection runs on packed fp32 pairs (sp3d_proj_pk.h), a tap record is one 16-byte weight quad + one
// offset word (2 LDS instructions per slot instead of 5), the interpolation is written on channel pairs.
template <int JP, typename TI, int U = 4>
__device__ __forceinline__ void pipe_views(const Views &hm, const float *__restrict__ cam, const Geom &g, int bs, float x,
                                           float y, float z, bool inb, float *ws, int lane, float (&acc)[4][4],
                                           uint32_t &mymask, unsigned long long *tl, bool vsync = false)
{
    // vsync (tuning bit 10, brick kernel only; round-5 L1-residency experiment): a workgroup barrier per view, so that all
    // waves of a workgroup gather from the SAME view at any time (every wave of the workgroup runs all V iterations)
    constexpr int NQ = JP / 4;
    int *wsi = reinterpret_cast<int *>(ws);
    float4 *ws4 = reinterpret_cast<float4 *>(ws);
    (void)tl;
    SP3D_DIAG_FLAGS();
#ifdef SP3D_DIAG
    if (SP3D_DIAG_ON(8)) {      // stagger: waves of one SIMD start up to ~1.5k cycles apart
        const unsigned hw = __builtin_amdgcn_s_getreg(63492);   // HW_ID: wave_id[3:0]
        for (unsigned k = 0; k < (hw & 3u); ++k) __builtin_amdgcn_s_sleep(8);
    }
#endif
    const unsigned long long inbm = __builtin_amdgcn_ballot_w64(inb);
    auto P1 = [&](int c) -> bool {
        const float *cm = cam + ((size_t)bs * g.V + c) * SP3D_CAM_STRIDE;
#ifdef SP3D_DIAG
        if (SP3D_DIAG_ON(4)) {      // no projection: a fixed record per lane (distinct pixels, in range)
            if (inb) mymask += 1u;
            const int v = (c & 1) * 64 + lane;
            wsi[WOFF + v] = (int)((unsigned)(lane * 37 + c * 4001 + 1000 + (int)(x * 0.01f)) % (unsigned)(g.w * (g.h - 2))) * (JP * (int)sizeof(TI));
            ws4[v] = make_float4(0.25f, 0.25f, 0.25f, 0.25f);
            return true;
        }
#endif
        P1State st;
        const bool go = project_pk(cm, g, x, y, z, inbm, st);
        add_mask(mymask, st.bm);
        if (st.nm != 0ull && lane_of(st.nm)) mymask |= 0x80000000u;
        if (!go) return false;
        const unsigned long long um = st.bm & ~st.nm;
        if (um == 0ull) return false;           // no voxel of this wave sees camera c
        const RecPk r = make_record_pk(lane_of(um), st.i, g.w, g.h);
        const int v = (c & 1) * 64 + lane;
        wsi[WOFF + v] = (int)__umul24((unsigned)(JP * (int)sizeof(TI)), __umul24((unsigned)r.y0, (unsigned)g.w) + (unsigned)r.x0);
        ws4[v] = make_float4(r.wt.x, r.wt.y, r.wb.x, r.wb.y);
        return true;
    };

    // gather mapping
    const int g16 = lane >> 2, q = lane & 3;
    const bool qact = q < NQ;
    const uint32_t qoff = qact ? 4u * (uint32_t)sizeof(TI) * (uint32_t)q : 0u;      // this lane's channel quad, bytes
    const size_t rowf = (size_t)g.w * JP;
    bool have = P1(0);
    SP3D_STAMP(1);
#pragma unroll 1
    for (int c = 0; c < g.V; ++c) {
        SP3D_STAMP(2 + 4 * (c < 7 ? c : 6));
        if (vsync) __builtin_amdgcn_s_barrier();
        const bool cur = have;
        // wave-uniform row bases (SGPR pairs) + one 32-bit element offset per lane: the four taps of a slot are
        // {vb, vb2} + off (+ JP as an immediate), no 64-bit VALU address arithmetic
        const char *vb = reinterpret_cast<const char *>(reinterpret_cast<const TI *>(hm.p[c]) + (size_t)bs * g.h * rowf);
        const char *vb2 = vb + rowf * sizeof(TI);
        const int rb = (c & 1) * 64 + g16;
        if (cur) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        // the 4 voxel slots of this lane group are gathered U at a time (4*U dwordx4 loads in flight);
        // P1(c+1) is scheduled between the first group's loads and its FMAs
#pragma unroll
        for (int gi = 0; gi < 4 / U; ++gi) {
            float4 t00[U], t10[U], t01[U], t11[U];
#ifdef SP3D_DIAG
            if (SP3D_DIAG_ON(2)) {
#pragma unroll
                for (int k = 0; k < U; ++k) t00[k] = t10[k] = t01[k] = t11[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            } else
#endif
            if (cur) {
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    const uint32_t off = (uint32_t)wsi[WOFF + rb + 16 * (gi * U + k)] + qoff;      // bytes
                    // (issued in the reverse of the order the interpolation consumes them: loads return in order, so
                    // the wait for t00 covers the slot's other three and the chain needs one s_waitcnt per slot, not four)
                    t11[k] = Store4<TI>::load(reinterpret_cast<const TI *>(vb2 + off) + JP);
                    t01[k] = Store4<TI>::load(reinterpret_cast<const TI *>(vb2 + off));
                    t10[k] = Store4<TI>::load(reinterpret_cast<const TI *>(vb + off) + JP);
                    t00[k] = Store4<TI>::load(reinterpret_cast<const TI *>(vb + off));
                }
            }
            if (gi == 0) {
                __builtin_amdgcn_sched_barrier(0);
                SP3D_STAMP(3 + 4 * (c < 7 ? c : 6));     // all tap loads issued
                if (c + 1 < g.V) have = P1(c + 1);       // VALU work while the taps are in flight
                __builtin_amdgcn_sched_barrier(0);
                SP3D_STAMP(4 + 4 * (c < 7 ? c : 6));     // next view projected
            }
            if (cur && !SP3D_DIAG_ON(16)) {
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    const int i = gi * U + k;
                    const float4 wq = ws4[rb + 16 * i];                 // (w00, w10, w01, w11)
                    // ATen's bilinear chain per channel: fma(se, wse, fma(sw, wsw, fma(ne, wne, nw * wnw)))
                    v2f lo = v2f{t00[k].x, t00[k].y} * pk2(wq.x), hi = v2f{t00[k].z, t00[k].w} * pk2(wq.x);
                    lo = pk_fma(v2f{t10[k].x, t10[k].y}, pk2(wq.y), lo); hi = pk_fma(v2f{t10[k].z, t10[k].w}, pk2(wq.y), hi);
                    lo = pk_fma(v2f{t01[k].x, t01[k].y}, pk2(wq.z), lo); hi = pk_fma(v2f{t01[k].z, t01[k].w}, pk2(wq.z), hi);
                    lo = pk_fma(v2f{t11[k].x, t11[k].y}, pk2(wq.w), lo); hi = pk_fma(v2f{t11[k].z, t11[k].w}, pk2(wq.w), hi);
                    const v2f a0 = v2f{acc[i][0], acc[i][1]} + lo, a1 = v2f{acc[i][2], acc[i][3]} + hi;
                    acc[i][0] = a0.x; acc[i][1] = a0.y; acc[i][2] = a1.x; acc[i][3] = a1.y;
                }
            }
        }
    }
}

template <int JP, int NW, bool OUTCL, typename TI, typename TO, int U = 4>
__device__ __forceinline__ void pipe_tile(const Views &hm, const float *__restrict__ cam, const float *__restrict__ centers,
                                          const uint8_t *__restrict__ valid, float *__restrict__ cubes,
                                          float *__restrict__ grids, const Geom &g, int b, int tile, float *smem,
                                          unsigned wid)
{
    constexpr int NQ = JP / 4;
    // U = voxel slots gathered per batch of loads (4, 2 and 1 measured equal in the one-tile-per-wave kernel)
    constexpr int WLDS = (JP * WOSTR > WREC) ? JP * WOSTR : WREC;   // per-wave LDS floats (sOut aliases the records)
    (void)wid;
    SP3D_DIAG_FLAGS();
    const int bs = g.sample_of ? g.sample_of[b] : b;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = tile * (64 * NW) + wave * 64;                          // first voxel of this wave
    if (n0 >= g.N) return;
    const int nvox = min(64, g.N - n0);
    TO *cb = reinterpret_cast<TO *>(cubes) + (OUTCL ? (size_t)b * g.J * g.N : (size_t)b * g.sB);
    float *ws = smem + wave * WLDS;
    // offset of voxel n inside one channel plane of a planar result (== n for the dense layout)
    auto plane_off = [&](int n) -> size_t {
        if (g.dense) return (size_t)n;
        int vx, rem, vy, vz;
        udiv_magic((uint32_t)n, (uint32_t)g.YZ, g.magicYZ, vx, rem);
        udiv_magic((uint32_t)rem, (uint32_t)g.Z, g.magicZ, vy, vz);
        return (size_t)vx * g.sX + (size_t)vy * g.sY + vz;
    };

    if (!valid[b]) { // skipped sample: zeros (project_layer.py:48,51,54)
        const size_t zo = OUTCL ? 0 : plane_off(n0 + (lane < nvox ? lane : 0));
        for (int j = 0; j < g.J; ++j)
            if (lane < nvox) Store4<TO>::store1(cb + (OUTCL ? ((size_t)(n0 + lane) * g.J + j) : ((size_t)j * g.sJ + zo)), 0.0f);
        if (grids && lane < nvox) {
            float *gp = grids + ((size_t)b * g.N + n0 + lane) * 3;
            gp[0] = 0.0f; gp[1] = 0.0f; gp[2] = 0.0f;
        }
        if (g.pass_mask && lane < nvox) g.pass_mask[(size_t)b * g.N + n0 + lane] = 0;
        return;
    }

    // this lane's voxel (P1 mapping)
    const bool inb = lane < nvox;
    const int n = n0 + (inb ? lane : 0);
    int vx, rem, vy, vz;
    udiv_magic((uint32_t)n, (uint32_t)g.YZ, g.magicYZ, vx, rem);
    udiv_magic((uint32_t)rem, (uint32_t)g.Z, g.magicZ, vy, vz);
    const float x = linspace_step(g.Lx, g.stepx, g.X, vx) + centers[3 * b + 0];
    const float y = linspace_step(g.Ly, g.stepy, g.Y, vy) + centers[3 * b + 1];
    const float z = linspace_step(g.Lz, g.stepz, g.Z, vz) + centers[3 * b + 2];
    if (grids && inb) {
        float *gp = grids + ((size_t)b * g.N + n) * 3;
        gp[0] = x; gp[1] = y; gp[2] = z;
    }
    uint32_t mymask = 0;                        // bound bits of MY voxel (+ bit 31: NaN position)
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i][0] = acc[i][1] = acc[i][2] = acc[i][3] = 0.0f;
    const int g16 = lane >> 2, q = lane & 3;
    const bool qact = q < NQ;

#ifdef SP3D_TIMELINE
    unsigned long long *tl = g_timeline ? g_timeline + ((size_t)wid * NW + wave) * 32 : nullptr;
#else
    unsigned long long *tl = nullptr;
#endif
    SP3D_STAMP_ALWAYS(0);
#ifdef SP3D_TIMELINE
    if (tl && lane == 0) tl[26] = wall_clock64();       // chip-wide 100 MHz clock (cycle counters are per XCD)
#endif
    pipe_views<JP, TI, U>(hm, cam, g, bs, x, y, z, inb, ws, lane, acc, mymask, tl);

    // view fusion (project_layer.py:96-99) on the gather mapping, result tile -> LDS
    __builtin_amdgcn_wave_barrier();
    SP3D_STAMP_ALWAYS(30);
#ifdef SP3D_TIMELINE
    if (tl && lane == 0) {      // where it ran: HW_ID (wave/simd/cu/se) and XCC_ID
        tl[25] = wall_clock64();                           // view loop done, epilogue starts
        tl[28] = (unsigned long long)__builtin_amdgcn_s_getreg(63492);
        tl[29] = (unsigned long long)__builtin_amdgcn_s_getreg(63508);
    }
    if (tl && lane == 0) tl[31] = (unsigned long long)(mymask & 0x7fffffffu);
#endif
    // per voxel (P1 mapping, once): den = #views seeing it + 1e-6, rden = RN(1/den), 0 for a NaN sample position
    const float den_l = (float)(mymask & 0x7fffffffu) + 1e-6f;
    const float rden_l = (mymask & 0x80000000u) ? 0.0f : 1.0f / den_l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float den = __shfl(den_l, 16 * i + g16);
        const float rden = __shfl(rden_l, 16 * i + g16);      // rden = 0 makes fuse_rcp return exactly 0
        const bool bad = rden == 0.0f;                        // NaN sample position: voxel is zero
        if (g.pass_mask) {
            // gradient pass mask (torch.clamp backward: 0 <= pre <= 1; NaN-zeroed voxels block it)
            uint32_t bits = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float pre = fuse_pre(acc[i][k], den, rden);
                if (!bad && pre >= 0.0f && pre <= 1.0f) bits |= 1u << (4 * q + k);
            }
            if (!qact) bits = 0;
            bits |= (uint32_t)__shfl_xor((int)bits, 1);
            bits |= (uint32_t)__shfl_xor((int)bits, 2);
            const int nn = 16 * i + g16;
            if (q == 0 && nn < nvox) g.pass_mask[(size_t)b * g.N + n0 + nn] = (uint16_t)bits;
        }
        if (OUTCL) {
            // channels-last result (B, N, J): this lane's 4 channels are 16 contiguous bytes, the
            // 4 lanes of a voxel 64 B, the wave's 16 voxels of slot i 1 KiB - no LDS transpose.
            const int nn = 16 * i + g16;
            if (qact && 4 * q < g.J && nn < nvox) {
                float4 o;
                o.x = fuse_rcp(acc[i][0], den, rden); o.y = fuse_rcp(acc[i][1], den, rden);
                o.z = fuse_rcp(acc[i][2], den, rden); o.w = fuse_rcp(acc[i][3], den, rden);
                if (!SP3D_DIAG_ON(1) || o.x == 123456.0f) Store4<TO>::store_nt(cb + (size_t)(n0 + nn) * g.J + 4 * q, o);
            }
        } else if (qact) {
#pragma unroll
            for (int k = 0; k < 4; ++k) ws[(4 * q + k) * WOSTR + 16 * i + g16] = fuse_rcp(acc[i][k], den, rden);
        }
    }
#ifdef SP3D_TIMELINE
    if (OUTCL) {
        __builtin_amdgcn_s_waitcnt(0);                          // vmcnt(0): the result stores have left the wave
        if (tl && lane == 0) tl[27] = wall_clock64();
    }
#endif
    if (OUTCL) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // 4 consecutive voxels form one 16-byte piece when they lie in one z-column (dense: any 4; strided: Z % 4 == 0)
    if (g.vec4 && ((g.N & 3) == 0) && nvox == 64 && (g.dense || (g.Z & 3) == 0)) {
        // lane -> (channel j = pass*4 + lane/16, voxel quad u = lane%16): 256 B contiguous per channel
        const int u = lane & 15;
        const size_t po = plane_off(n0 + 4 * u);
        for (int j = lane >> 4; j < g.J; j += 4) {
            const float4 o = *reinterpret_cast<const float4 *>(&ws[j * WOSTR + 4 * u]);
            if (!SP3D_DIAG_ON(1) || o.x == 123456.0f) Store4<TO>::store_nt(cb + (size_t)j * g.sJ + po, o);
        }
    } else {
        const size_t po = plane_off(n0 + (lane < nvox ? lane : 0));
        for (int j = 0; j < g.J; ++j)
            if (lane < nvox) Store4<TO>::store1(cb + (size_t)j * g.sJ + po, ws[j * WOSTR + lane]);
    }
}

// NW = waves per workgroup (waves are independent; NW only sets the dispatch granularity)
// TI / TO: storage type of the packed heat-maps / of the cubes (float or bf16_t); math is fp32.
template <int JP, bool XCD, int NW, bool OUTCL, typename TI = float, typename TO = float>
__global__ __launch_bounds__(64 * NW) void unproject_pipe_kernel(Views hm, const float *__restrict__ cam,
                                                             const float *__restrict__ centers,
                                                             const uint8_t *__restrict__ valid,
                                                             float *__restrict__ cubes, float *__restrict__ grids,
                                                             Geom g, int tiles_per_sample, int total_tiles)
{
    constexpr int WLDS = (JP * WOSTR > WREC) ? JP * WOSTR : WREC;
    __shared__ __attribute__((aligned(16))) float smem[NW * WLDS];
    int b, tile;
    if (XCD) {
        if (!xcd_map_fast(blockIdx.x, g, b, tile)) return;
    } else {
        b = blockIdx.x / tiles_per_sample;
        tile = blockIdx.x - b * tiles_per_sample;
    }
    (void)total_tiles;
    pipe_tile<JP, NW, OUTCL, TI, TO>(hm, cam, centers, valid, cubes, grids, g, b, tile, smem, blockIdx.x);
}

// ------------------------------------------------------------------------------------------
// "brick" kernel: the same per-wave pipeline, but a wave owns a 4x4x4 block of voxels instead of 64
// consecutive ones, and a workgroup is a stack of `zw` such bricks along z.
//
// Why: the gather is bound by L1 misses, not bytes (profiles/r01_pmc_unproject_coarse_b4.json: 9 L2
// requests per 16-quad wave-load, TA busy 76 % of the kernel).  64 consecutive voxels are 3.2 z-columns:
// their projections in one view form 3 well separated vertical runs, and no two voxels of the wave
// share a 128-B line (2.5 distinct lines per voxel-view on the root grid, 2.3 on the 64^3 person
// cubes).  A compact brick always has neighbours along every camera's line of sight; those project
// onto nearly the same pixels: 1.7 lines per voxel-view on the root grid (80 mm voxels, ~4 px apart),
// 0.6-0.8 on the 64^3 cubes and the 160x160x40 grid (tools/sim_l1.py).
//
// Lane -> voxel: lx = lane/16, ly = (lane/4)%4, lz = lane%4 (z fastest, as in memory).  Gather slot i
// of lane group g16 is voxel 16*i + g16, i.e. (lx, ly, lz) = (i, g16/4, g16%4).
// Planar results: every wave leaves its (J x 64) tile in LDS, then the workgroup stores whole z-runs:
// a 16-byte piece = 4 z of one (channel, column), `zw` pieces in a row are contiguous, and so are the
// 4 y-neighbouring columns (when Y pitch == Z): 4*zw*16-byte runs.  Channels-last results leave from
// the gather mapping directly (64 B per voxel).
// ------------------------------------------------------------------------------------------
constexpr int BR = 4;

#ifndef SP3D_BRICK_U
#define SP3D_BRICK_U 4          // voxel slots gathered per batch of tap loads (16 dwordx4 in flight at 4)
#endif
#ifndef SP3D_BRICK_MINW
#define SP3D_BRICK_MINW 4
#endif
// the ZD form's own gather depth / occupancy target (its workgroups are 5 waves: 3 fit a CU at 4 waves per SIMD, 4 at 5)
// U = 2: 88 VGPRs -> 5 waves per SIMD -> FOUR 5-wave workgroups per CU instead of three: 37.5 -> 32.9 us warm, 51.8 -> 48.1
// behind a cache flush (U = 1 / 6 waves: 33.2 / 51.7).  The one-brick workgroups of the other forms keep U = 4 (26.1 vs 24.4 us).
#ifndef SP3D_ZD_U
#define SP3D_ZD_U 2
#endif
#ifndef SP3D_ZD_MINW
#define SP3D_ZD_MINW SP3D_BRICK_MINW
#endif
// ZD (round 6, root grid only: Z == ZDZ voxels = the whole z extent in ONE stack, JP == 16, float in / out): the workgroup
// does not store its cubes at all.  Its 4 x 4 columns x Z x J values stay in LDS and leave as the z-SPECTRUM the opening
// 7^3 conv wants (the direct ZDZ -> ZDSZ/2+1 point DFT of zdft_fwd_cl_kernel, sp3d_fft.hip: same table, same FMA order,
// same bits), in a layout whose unit is this workgroup's 4 x 4 tile: (B, J, K, X/4, Y/4, 16) complex, so every store is
// one whole 128-byte line.  `cubes` then points at that spectrum.  Deletes the cubes' write + re-read (2 x 32.8 MB at
// B = 4) and one launch from the root-net step; cfft2d_88_kernel un-tiles while it loads a plane into LDS.
constexpr int ZDZ = 20, ZDSZ = 28;
constexpr int SP3D_VARIANT_ZD = 1 << 24;        // launch_nhwc `variant` bit: brick stacks emit the z-spectrum
constexpr int SP3D_VARIANT_CHUNKS = 1 << 22;    // launch_nhwc `variant` bit: round-5 chunk map of the bricks instead of blocks / octants
template <int JP, bool OUTCL, typename TI = float, typename TO = float, bool ZD = false>
__global__ __launch_bounds__(512, ZD ? SP3D_ZD_MINW : SP3D_BRICK_MINW) void unproject_brick_kernel(Views hm, const float *__restrict__ cam,
                                                                const float *__restrict__ centers,
                                                                const uint8_t *__restrict__ valid,
                                                                float *__restrict__ cubes, float *__restrict__ grids,
                                                                Geom g, int wgs_per_sample, int nby, int nzc, int zw)
{
    constexpr int NQ = JP / 4;
    constexpr int WLDS = (JP * WOSTR > WREC) ? JP * WOSTR : WREC;
    extern __shared__ __attribute__((aligned(16))) float bsmem[];
    int b, wg;
    if (!xcd_map_fast(blockIdx.x, g, b, wg)) return;
    SP3D_DIAG_FLAGS();
    int zc, t;
    if (!(g.xcd_order & 2)) {   // default (round 3): z slowest - consecutive workgroups sweep (y, x) inside one z-layer of bricks
        // and an XCD's chunk is a z-slab: -3 % on all three grids (profiles/r03_ab_zslab.json)
        udiv_magic((uint32_t)wg, (uint32_t)g.bk_nxy, g.bk_magic_nxy, zc, t);
    } else {                    // bit 8 of the tuning `variant`: round 2's order, z fastest
        zc = wg % nzc; t = wg / nzc;
    }
    int bx, by;
    udiv_magic((uint32_t)t, (uint32_t)g.bk_nby, g.bk_magic_nby, bx, by);
    const int bs = g.sample_of ? g.sample_of[b] : b;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int x0 = bx * BR, y0 = by * BR, zbase = zc * zw * BR, z0 = zbase + wave * BR;
    TO *cb = reinterpret_cast<TO *>(cubes) + (OUTCL ? (size_t)b * g.J * g.N : (size_t)b * g.sB);
    float *ws = bsmem + wave * WLDS;

    // P1 mapping: this lane's voxel
    const int lx = lane >> 4, ly = (lane >> 2) & 3, lz = lane & 3;
    const int vx = x0 + lx, vy = y0 + ly, vz = z0 + lz;
    const bool inb = vx < g.X && vy < g.Y && vz < g.Z;
    const int n = (min(vx, g.X - 1) * g.Y + min(vy, g.Y - 1)) * g.Z + min(vz, g.Z - 1);
    // gather mapping: slot i of this lane is voxel (x0 + i, y0 + g16/4, z0 + g16%4)
    const int g16 = lane >> 2, q = lane & 3;
    const bool qact = q < NQ;
    const int gy = y0 + (g16 >> 2), gz = z0 + (g16 & 3);
    const bool ginb = gy < g.Y && gz < g.Z;
    const int gn0 = (x0 * g.Y + min(gy, g.Y - 1)) * g.Z + min(gz, g.Z - 1);        // + i * YZ

    const bool dead = ZD && !valid[b];      // ZD: a skipped sample's workgroups still emit their (all-zero) spectrum lines
    if (!ZD && !valid[b]) { // skipped sample: zeros (project_layer.py:48,51,54)
        if (inb) {
            const size_t zo = (size_t)vx * g.sX + (size_t)vy * g.sY + vz;
            for (int j = 0; j < g.J; ++j)
                Store4<TO>::store1(cb + (OUTCL ? ((size_t)n * g.J + j) : ((size_t)j * g.sJ + zo)), 0.0f);
            if (grids) {
                float *gp = grids + ((size_t)b * g.N + n) * 3;
                gp[0] = 0.0f; gp[1] = 0.0f; gp[2] = 0.0f;
            }
            if (g.pass_mask) g.pass_mask[(size_t)b * g.N + n] = 0;
        }
        return;
    }

    if (dead) {
        for (int i = lane; i < JP * WOSTR; i += 64) ws[i] = 0.0f;
    } else if (z0 < g.Z) {      // (a stack's last waves may lie above the volume: they only join the barrier)
        const float x = linspace_step(g.Lx, g.stepx, g.X, min(vx, g.X - 1)) + centers[3 * b + 0];
        const float y = linspace_step(g.Ly, g.stepy, g.Y, min(vy, g.Y - 1)) + centers[3 * b + 1];
        const float z = linspace_step(g.Lz, g.stepz, g.Z, min(vz, g.Z - 1)) + centers[3 * b + 2];
        if (grids && inb) {
            float *gp = grids + ((size_t)b * g.N + n) * 3;
            gp[0] = x; gp[1] = y; gp[2] = z;
        }
        uint32_t mymask = 0;
        float acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][0] = acc[i][1] = acc[i][2] = acc[i][3] = 0.0f;
#ifdef SP3D_TIMELINE
        unsigned long long *tl = g_timeline ? g_timeline + ((size_t)blockIdx.x * zw + wave) * 32 : nullptr;
        SP3D_STAMP_ALWAYS(0);
        if (tl && lane == 0) tl[26] = wall_clock64();
#else
        unsigned long long *tl = nullptr;
#endif
        pipe_views<JP, TI, ZD ? SP3D_ZD_U : SP3D_BRICK_U>(hm, cam, g, bs, x, y, z, inb, ws, lane, acc, mymask, tl, (g.xcd_order & 4) != 0);

        // view fusion (project_layer.py:96-99) on the gather mapping
        __builtin_amdgcn_wave_barrier();
        SP3D_STAMP_ALWAYS(30);
#ifdef SP3D_TIMELINE
        if (tl && lane == 0) {
            tl[25] = wall_clock64();
            tl[28] = (unsigned long long)__builtin_amdgcn_s_getreg(63492);
            tl[29] = (unsigned long long)__builtin_amdgcn_s_getreg(63508);
            tl[31] = (unsigned long long)(mymask & 0x7fffffffu);
        }
#endif
        const float den_l = (float)(mymask & 0x7fffffffu) + 1e-6f;
        const float rden_l = (mymask & 0x80000000u) ? 0.0f : 1.0f / den_l;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float den = __shfl(den_l, 16 * i + g16);
            const float rden = __shfl(rden_l, 16 * i + g16);      // rden = 0 makes fuse_rcp return exactly 0
            const bool bad = rden == 0.0f;
            const bool vin = ginb && (x0 + i < g.X);
            const int gn = gn0 + i * g.YZ;
            if (g.pass_mask) {
                uint32_t bits = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float pre = fuse_pre(acc[i][k], den, rden);
                    if (!bad && pre >= 0.0f && pre <= 1.0f) bits |= 1u << (4 * q + k);
                }
                if (!qact) bits = 0;
                bits |= (uint32_t)__shfl_xor((int)bits, 1);
                bits |= (uint32_t)__shfl_xor((int)bits, 2);
                if (q == 0 && vin) g.pass_mask[(size_t)b * g.N + gn] = (uint16_t)bits;
            }
            if (OUTCL) {
                if (qact && 4 * q < g.J && vin) {
                    float4 o;
                    o.x = fuse_rcp(acc[i][0], den, rden); o.y = fuse_rcp(acc[i][1], den, rden);
                    o.z = fuse_rcp(acc[i][2], den, rden); o.w = fuse_rcp(acc[i][3], den, rden);
                    if (!SP3D_DIAG_ON(1) || o.x == 123456.0f) Store4<TO>::store_nt(cb + (size_t)gn * g.J + 4 * q, o);
                }
            } else if (qact) {
#pragma unroll
                for (int k = 0; k < 4; ++k) ws[(4 * q + k) * WOSTR + 16 * i + g16] = fuse_rcp(acc[i][k], den, rden);
            }
        }
#ifdef SP3D_TIMELINE
        if (OUTCL) {
            __builtin_amdgcn_s_waitcnt(0);                      // the result stores have left the wave
            if (tl && lane == 0) tl[27] = wall_clock64();
        }
#endif
    }
    if (OUTCL) return;
    __syncthreads();
    if constexpr (ZD) {
        // thread -> (column pos = 4 * lx + ly of the tile, channel c): a wave holds 4 channels x 16 columns, its 16-lane groups
        // store 16 complex values = one 128-byte line per (c, kz)
        constexpr int K = ZDSZ / 2 + 1;
        const int pos = tid & 15, c = tid >> 4;
        if (c >= g.J) return;
        float v[ZDZ];
#pragma unroll
        for (int wz = 0; wz < ZDZ / BR; ++wz) {
            const float4 q4 = *reinterpret_cast<const float4 *>(bsmem + wz * WLDS + c * WOSTR + pos * 4);
            v[4 * wz] = q4.x; v[4 * wz + 1] = q4.y; v[4 * wz + 2] = q4.z; v[4 * wz + 3] = q4.w;
        }
        float2 *o = reinterpret_cast<float2 *>(cubes) + ((((size_t)b * g.J + c) * K) * (size_t)g.bk_nxy + (size_t)t) * 16 + pos;
        const size_t kstride = (size_t)g.bk_nxy * 16;
        float re[K], im[K];
        zdft_real<ZDZ, ZDSZ>(v, re, im);
#pragma unroll
        for (int k = 0; k < K; ++k) o[(size_t)k * kstride] = make_float2(re[k], im[k]);
        return;
    }
    // workgroup store of the (J, 4, 4, 4*zw) block: thread -> (channel phase jj, column, brick of the stack)
    const float rzw = 1.0f / (float)zw;
    const int per = 16 * zw;                                       // (column, brick) pairs = threads per channel phase
    const int jj = (int)(((float)(tid >> 4) + 0.5f) * rzw);        // tid / per            (0..3)
    const int cw = tid - jj * per;
    const int col = (int)(((float)cw + 0.5f) * rzw), wz = cw - col * zw;
    const int sx = x0 + (col >> 2), sy = y0 + (col & 3), sz = zbase + wz * BR;
    if (sx >= g.X || sy >= g.Y || sz >= g.Z) return;
    const float *tile = bsmem + wz * WLDS + col * 4;
    TO *dst = cb + (size_t)sx * g.sX + (size_t)sy * g.sY + sz;
    if (g.vec4 && (g.Z & 3) == 0) {
        for (int j = jj; j < g.J; j += 4) {
            const float4 o = *reinterpret_cast<const float4 *>(tile + j * WOSTR);
            if (!SP3D_DIAG_ON(1) || o.x == 123456.0f) Store4<TO>::store_nt(dst + (size_t)j * g.sJ, o);
        }
    } else {
        const int nz = min(BR, g.Z - sz);
        for (int j = jj; j < g.J; j += 4)
            for (int k = 0; k < nz; ++k) Store4<TO>::store1(dst + (size_t)j * g.sJ + k, tile[j * WOSTR + k]);
    }
}

// ------------------------------------------------------------------------------------------
// bf16 heat-maps (BASELINE configs[4]) on bricks with TWO lanes per pixel (round 4).
//
// The fp32 kernels give a 64-byte pixel to 4 lanes (16 B each).  With bf16 storage the same mapping loads 8 B per lane:
// half the bytes, the SAME 16 tap wave-loads per view - and the gather is bound by wave-loads through the texture path and
// by L1 line fills, not by bytes (profiles/r04_issue_model.md), so bf16 storage bought nothing and the conversion made it
// slower than fp32 (99.5 vs 90.3 us, ten 64^3 cubes, 4 views).  Here a 32-byte bf16 pixel goes to 2 lanes, 16 B = 8 channels
// each: a wave-load covers 32 voxels instead of 16, a view needs 8 wave-loads instead of 16, and every lane still owns 16
// accumulators (2 voxel slots x 8 channels instead of 4 x 4).  Arithmetic: the bf16 values are widened exactly (<< 16) and
// go through the same fp32 chain in the same order => the same bits as the 4-lane kernel and the oracle on the rounded maps.
// Lane -> voxel for P1 as in the fp32 brick kernel (lane = lx*16 + ly*4 + lz); gather slot i of lane pair g32 = lane/2 is
// voxel 32*i + g32.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void bf16x8_to_f32(const uint4 r, float (&f)[8])
{
    f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
    f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
    f[4] = __uint_as_float(r.z << 16); f[5] = __uint_as_float(r.z & 0xffff0000u);
    f[6] = __uint_as_float(r.w << 16); f[7] = __uint_as_float(r.w & 0xffff0000u);
}

__device__ __forceinline__ void pipe_views_h(const Views &hm, const float *__restrict__ cam, const Geom &g, int bs, float x,
                                             float y, float z, bool inb, float *ws, int lane, float (&acc)[2][8],
                                             uint32_t &mymask)
{
    constexpr int JP = 16;
    int *wsi = reinterpret_cast<int *>(ws);
    float4 *ws4 = reinterpret_cast<float4 *>(ws);
    const unsigned long long inbm = __builtin_amdgcn_ballot_w64(inb);
    auto P1 = [&](int c) -> bool {
        const float *cm = cam + ((size_t)bs * g.V + c) * SP3D_CAM_STRIDE;
        P1State st;
        const bool go = project_pk(cm, g, x, y, z, inbm, st);
        add_mask(mymask, st.bm);
        if (st.nm != 0ull && lane_of(st.nm)) mymask |= 0x80000000u;
        if (!go) return false;
        const unsigned long long um = st.bm & ~st.nm;
        if (um == 0ull) return false;
        const RecPk r = make_record_pk(lane_of(um), st.i, g.w, g.h);
        const int v = (c & 1) * 64 + lane;
        wsi[WOFF + v] = (int)__umul24((unsigned)(JP * 2), __umul24((unsigned)r.y0, (unsigned)g.w) + (unsigned)r.x0);     // bytes
        ws4[v] = make_float4(r.wt.x, r.wt.y, r.wb.x, r.wb.y);
        return true;
    };
    const int g32 = lane >> 1, q = lane & 1;
    const uint32_t qoff = 16u * (uint32_t)q;                    // this lane's 8 channels, bytes
    const size_t row_bytes = (size_t)g.w * JP * 2;
    bool have = P1(0);
#pragma unroll 1
    for (int c = 0; c < g.V; ++c) {
        const bool cur = have;
        const char *vb = reinterpret_cast<const char *>(hm.p[c]) + (size_t)bs * g.h * row_bytes;
        const char *vb2 = vb + row_bytes;
        const int rb = (c & 1) * 64 + g32;
        if (cur) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        uint4 t00[2], t10[2], t01[2], t11[2];
        if (cur) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const uint32_t off = (uint32_t)wsi[WOFF + rb + 32 * i] + qoff;
                t11[i] = *reinterpret_cast<const uint4 *>(vb2 + off + JP * 2);
                t01[i] = *reinterpret_cast<const uint4 *>(vb2 + off);
                t10[i] = *reinterpret_cast<const uint4 *>(vb + off + JP * 2);
                t00[i] = *reinterpret_cast<const uint4 *>(vb + off);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (c + 1 < g.V) have = P1(c + 1);       // VALU work while the taps are in flight
        __builtin_amdgcn_sched_barrier(0);
        if (cur) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float4 wq = ws4[rb + 32 * i];                 // (w00, w10, w01, w11)
                float a[8], b[8], cc[8], d[8];
                bf16x8_to_f32(t00[i], a); bf16x8_to_f32(t10[i], b); bf16x8_to_f32(t01[i], cc); bf16x8_to_f32(t11[i], d);
#pragma unroll
                for (int k = 0; k < 8; k += 2) {
                    // ATen's bilinear chain per channel: fma(se, wse, fma(sw, wsw, fma(ne, wne, nw * wnw)))
                    v2f v = v2f{a[k], a[k + 1]} * pk2(wq.x);
                    v = pk_fma(v2f{b[k], b[k + 1]}, pk2(wq.y), v);
                    v = pk_fma(v2f{cc[k], cc[k + 1]}, pk2(wq.z), v);
                    v = pk_fma(v2f{d[k], d[k + 1]}, pk2(wq.w), v);
                    const v2f s2 = v2f{acc[i][k], acc[i][k + 1]} + v;
                    acc[i][k] = s2.x; acc[i][k + 1] = s2.y;
                }
            }
        }
    }
}

template <bool OUTCL, typename TO>
__global__ __launch_bounds__(512, SP3D_BRICK_MINW) void unproject_brick_h_kernel(Views hm, const float *__restrict__ cam,
                                                                  const float *__restrict__ centers,
                                                                  const uint8_t *__restrict__ valid,
                                                                  float *__restrict__ cubes, float *__restrict__ grids,
                                                                  Geom g, int wgs_per_sample, int nby, int nzc, int zw)
{
    constexpr int JP = 16;
    constexpr int WLDS = (JP * WOSTR > WREC) ? JP * WOSTR : WREC;
    extern __shared__ __attribute__((aligned(16))) float bsmem[];
    int b, wg;
    if (!xcd_map_fast(blockIdx.x, g, b, wg)) return;
    int zc, t;
    if (!(g.xcd_order & 2)) udiv_magic((uint32_t)wg, (uint32_t)g.bk_nxy, g.bk_magic_nxy, zc, t);
    else { zc = wg % nzc; t = wg / nzc; }
    int bx, by;
    udiv_magic((uint32_t)t, (uint32_t)g.bk_nby, g.bk_magic_nby, bx, by);
    const int bs = g.sample_of ? g.sample_of[b] : b;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int x0 = bx * BR, y0 = by * BR, zbase = zc * zw * BR, z0 = zbase + wave * BR;
    TO *cb = reinterpret_cast<TO *>(cubes) + (OUTCL ? (size_t)b * g.J * g.N : (size_t)b * g.sB);
    float *ws = bsmem + wave * WLDS;

    // P1 mapping: this lane's voxel
    const int lx = lane >> 4, ly = (lane >> 2) & 3, lz = lane & 3;
    const int vx = x0 + lx, vy = y0 + ly, vz = z0 + lz;
    const bool inb = vx < g.X && vy < g.Y && vz < g.Z;
    const int n = (min(vx, g.X - 1) * g.Y + min(vy, g.Y - 1)) * g.Z + min(vz, g.Z - 1);
    const int g32 = lane >> 1, q = lane & 1;

    if (!valid[b]) { // skipped sample: zeros (project_layer.py:48,51,54)
        if (inb) {
            const size_t zo = (size_t)vx * g.sX + (size_t)vy * g.sY + vz;
            for (int j = 0; j < g.J; ++j)
                Store4<TO>::store1(cb + (OUTCL ? ((size_t)n * g.J + j) : ((size_t)j * g.sJ + zo)), 0.0f);
            if (grids) {
                float *gp = grids + ((size_t)b * g.N + n) * 3;
                gp[0] = 0.0f; gp[1] = 0.0f; gp[2] = 0.0f;
            }
            if (g.pass_mask) g.pass_mask[(size_t)b * g.N + n] = 0;
        }
        return;
    }

    if (z0 < g.Z) {
        const float x = linspace_step(g.Lx, g.stepx, g.X, min(vx, g.X - 1)) + centers[3 * b + 0];
        const float y = linspace_step(g.Ly, g.stepy, g.Y, min(vy, g.Y - 1)) + centers[3 * b + 1];
        const float z = linspace_step(g.Lz, g.stepz, g.Z, min(vz, g.Z - 1)) + centers[3 * b + 2];
        if (grids && inb) {
            float *gp = grids + ((size_t)b * g.N + n) * 3;
            gp[0] = x; gp[1] = y; gp[2] = z;
        }
        uint32_t mymask = 0;
        float acc[2][8];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[i][k] = 0.0f;
        pipe_views_h(hm, cam, g, bs, x, y, z, inb, ws, lane, acc, mymask);

        __builtin_amdgcn_wave_barrier();
        const float den_l = (float)(mymask & 0x7fffffffu) + 1e-6f;
        const float rden_l = (mymask & 0x80000000u) ? 0.0f : 1.0f / den_l;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int v = 32 * i + g32;                             // this slot's voxel inside the brick
            const float den = __shfl(den_l, v);
            const float rden = __shfl(rden_l, v);
            const bool bad = rden == 0.0f;
            const int gx = x0 + (v >> 4), gy = y0 + ((v >> 2) & 3), gz = z0 + (v & 3);
            const bool vin = gx < g.X && gy < g.Y && gz < g.Z;
            const int gn = (min(gx, g.X - 1) * g.Y + min(gy, g.Y - 1)) * g.Z + min(gz, g.Z - 1);
            float o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = fuse_rcp(acc[i][k], den, rden);
            if (g.pass_mask) {
                uint32_t bits = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float pre = fuse_pre(acc[i][k], den, rden);
                    if (!bad && pre >= 0.0f && pre <= 1.0f) bits |= 1u << (8 * q + k);
                }
                bits |= (uint32_t)__shfl_xor((int)bits, 1);
                if (q == 0 && vin) g.pass_mask[(size_t)b * g.N + gn] = (uint16_t)bits;
            }
            if (OUTCL) {
                if (vin) {
                    TO *dst = cb + (size_t)gn * g.J + 8 * q;
                    if (sizeof(TO) == 2 && 8 * q + 4 < g.J) {
                        // bf16 cubes: the lane's 8 channels as ONE 16-byte store (round 5: two 8-byte pieces made the L2
                        // write 136 MB for 84 MB of cubes, profiles/r05_pmc_configs4_bf16_v4.json)
                        typedef uint32_t v4u __attribute__((ext_vector_type(4)));
                        const uint2 lo = Store4<bf16_t>::pack4(make_float4(o[0], o[1], o[2], o[3]));
                        const uint2 hi = Store4<bf16_t>::pack4(make_float4(o[4], o[5], o[6], o[7]));
                        v4u t4; t4.x = lo.x; t4.y = lo.y; t4.z = hi.x; t4.w = hi.y;
                        __builtin_nontemporal_store(t4, reinterpret_cast<v4u *>(dst));
                    } else {
                        if (8 * q < g.J) Store4<TO>::store_nt(dst, make_float4(o[0], o[1], o[2], o[3]));
                        if (8 * q + 4 < g.J) Store4<TO>::store_nt(dst + 4, make_float4(o[4], o[5], o[6], o[7]));
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) ws[(8 * q + k) * WOSTR + v] = o[k];
            }
        }
    }
    if (OUTCL) return;
    __syncthreads();
    // workgroup store of the (J, 4, 4, 4*zw) block: thread -> (channel phase jj, column, brick of the stack); the LDS tile
    // is indexed by the voxel's brick-local number lx*16 + ly*4 + lz, as the fp32 kernel's (slot*16 + g16)
    const float rzw = 1.0f / (float)zw;
    const int per = 16 * zw;
    const int jj = (int)(((float)(tid >> 4) + 0.5f) * rzw);
    const int cw = tid - jj * per;
    const int col = (int)(((float)cw + 0.5f) * rzw), wz = cw - col * zw;
    const int sx = x0 + (col >> 2), sy = y0 + (col & 3), sz = zbase + wz * BR;
    if (sx >= g.X || sy >= g.Y || sz >= g.Z) return;
    const float *tile = bsmem + wz * WLDS + col * 4;
    TO *dst = cb + (size_t)sx * g.sX + (size_t)sy * g.sY + sz;
    if (g.vec4 && (g.Z & 3) == 0) {
        for (int j = jj; j < g.J; j += 4) {
            const float4 o = *reinterpret_cast<const float4 *>(tile + j * WOSTR);
            Store4<TO>::store_nt(dst + (size_t)j * g.sJ, o);
        }
    } else {
        const int nz = min(BR, g.Z - sz);
        for (int j = jj; j < g.J; j += 4)
            for (int k = 0; k < nz; ++k) Store4<TO>::store1(dst + (size_t)j * g.sJ + k, tile[j * WOSTR + k]);
    }
}

// ------------------------------------------------------------------------------------------
// backward: lane = voxel, planar layout.  Pass 1 recomputes the pre-clamp forward value (the
// clamp mask: grad flows where 0 <= pre <= 1, torch.clamp backward), pass 2 scatters.
// ------------------------------------------------------------------------------------------
template <int JC>
__global__ __launch_bounds__(TILE) void unproject_bwd_kernel(Views hm, const float *__restrict__ cam,
                                                            const float *__restrict__ centers,
                                                            const uint8_t *__restrict__ valid,
                                                            const float *__restrict__ grad_cubes, ViewsMut ghm,
                                                            Geom g)
{
    const int b = blockIdx.y;
    const int bs = g.sample_of ? g.sample_of[b] : b;
    const int n = blockIdx.x * TILE + threadIdx.x;
    if (n >= g.N || !valid[b]) return;
    const int vx = n / g.YZ, rem = n - vx * g.YZ, vy = rem / g.Z, vz = rem - vy * g.Z;
    const float x = linspace_at(g.Lx, g.X, vx) + centers[3 * b + 0];
    const float y = linspace_at(g.Ly, g.Y, vy) + centers[3 * b + 1];
    const float z = linspace_at(g.Lz, g.Z, vz) + centers[3 * b + 2];
    const float W_in = (float)g.W_in, H_in = (float)g.H_in;
    const size_t plane = (size_t)g.h * g.w;
    const float *gc = grad_cubes + (size_t)b * g.J * g.N + n;
    for (int j0 = 0; j0 < g.J; j0 += JC) {
        float acc[JC];
#pragma unroll
        for (int k = 0; k < JC; ++k) acc[k] = 0.0f;
        float cnt = 0.0f;
        bool bad = false;
        for (int c = 0; c < g.V; ++c) {
            const float *cm = cam + ((size_t)bs * g.V + c) * SP3D_CAM_STRIDE;
            float ix, iy;
            const bool bound = sample_pos(cm, x, y, z, g.w, g.h, W_in, H_in, ix, iy);
            cnt += bound ? 1.0f : 0.0f;
            if (ix != ix || iy != iy) { bad = true; continue; }
            if (!bound) continue;
            const Bilin bl = bilin(ix, iy);
            const bool x0ok = bl.x0 >= 0 && bl.x0 <= g.w - 1, x1ok = bl.x0 + 1 >= 0 && bl.x0 + 1 <= g.w - 1;
            const bool y0ok = bl.y0 >= 0 && bl.y0 <= g.h - 1, y1ok = bl.y0 + 1 >= 0 && bl.y0 + 1 <= g.h - 1;
            const float *base = hm.p[c] + ((size_t)bs * g.J + j0) * plane + (ptrdiff_t)bl.y0 * g.w + bl.x0;
#pragma unroll
            for (int k = 0; k < JC; ++k) {
                if (j0 + k < g.J) {
                    const float *pl = base + (size_t)k * plane;
                    const float t00 = (x0ok && y0ok) ? pl[0] : 0.0f;
                    const float t10 = (x1ok && y0ok) ? pl[1] : 0.0f;
                    const float t01 = (x0ok && y1ok) ? pl[g.w] : 0.0f;
                    const float t11 = (x1ok && y1ok) ? pl[g.w + 1] : 0.0f;
                    float v = t00 * bl.wnw;
                    v = fmaf(t10, bl.wne, v);
                    v = fmaf(t01, bl.wsw, v);
                    v = fmaf(t11, bl.wse, v);
                    acc[k] = acc[k] + v;
                }
            }
        }
        if (bad) continue;
        const float den = cnt + 1e-6f;
        float gs[JC];
        bool anyg = false;
#pragma unroll
        for (int k = 0; k < JC; ++k) {
            gs[k] = 0.0f;
            if (j0 + k < g.J) {
                const float pre = acc[k] / den;
                if (pre >= 0.0f && pre <= 1.0f) {
                    gs[k] = gc[(size_t)(j0 + k) * g.N] / den;
                    anyg = anyg || (gs[k] != 0.0f);
                }
            }
        }
        if (!anyg) continue;
        for (int c = 0; c < g.V; ++c) {
            const float *cm = cam + ((size_t)bs * g.V + c) * SP3D_CAM_STRIDE;
            float ix, iy;
            const bool bound = sample_pos(cm, x, y, z, g.w, g.h, W_in, H_in, ix, iy);
            if (!bound) continue;
            const Bilin bl = bilin(ix, iy);
            const bool x0ok = bl.x0 >= 0 && bl.x0 <= g.w - 1, x1ok = bl.x0 + 1 >= 0 && bl.x0 + 1 <= g.w - 1;
            const bool y0ok = bl.y0 >= 0 && bl.y0 <= g.h - 1, y1ok = bl.y0 + 1 >= 0 && bl.y0 + 1 <= g.h - 1;
            float *base = ghm.p[c] + ((size_t)bs * g.J + j0) * plane + (ptrdiff_t)bl.y0 * g.w + bl.x0;
#pragma unroll
            for (int k = 0; k < JC; ++k) {
                if (j0 + k < g.J && gs[k] != 0.0f) {
                    float *pl = base + (size_t)k * plane;
                    if (x0ok && y0ok) unsafeAtomicAdd(pl, gs[k] * bl.wnw);
                    if (x1ok && y0ok) unsafeAtomicAdd(pl + 1, gs[k] * bl.wne);
                    if (x0ok && y1ok) unsafeAtomicAdd(pl + g.w, gs[k] * bl.wsw);
                    if (x1ok && y1ok) unsafeAtomicAdd(pl + g.w + 1, gs[k] * bl.wse);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward, line-coalesced scatter ("bwd2").  Needs the pass mask written by the forward pipe kernel,
// so no heat-map is re-read.  L2 fp32 atomics are one transaction per (instruction, cache line): 64
// scattered lanes run at 21 G atomics/s, 16 lanes on the 16 channels of one 64-B pixel at 325 G/s
// (tools/atomic_bench.hip).  Hence: gradients accumulate into a channels-last (V,B,h,w,16) buffer and
// the scatter maps lane = (voxel-of-4, channel): one atomic instruction = 4 pixels x 16 channels.
//   P1   lane = voxel: sample records of every view -> LDS (same code as the forward kernel)
//   load grad tile (J rows of 64 voxels, coalesced) -> LDS, pass mask / view masks per voxel -> LDS
//   S    lane = (v4, ch): for its 16 voxels, g = pass ? grad / den : 0, then per bound view 4 atomics
// ------------------------------------------------------------------------------------------
// DET: accumulate in 64-bit FIXED POINT (value * *scale, rounded to nearest) with integer atomics.  Integer addition is
// associative, so the result does not depend on the order in which the hardware retires the atomics: bit-identical
// run to run (what SURVEY.md §5 asks for, since the reference's grid_sampler_2d_backward is order-dependent too);
// sp3d_fixed_to_float converts back.  *scale = 2^k chosen by the caller from max|grad| so that 2^40 steps span it.
template <int JP, bool XCD, bool DET = false>
__global__ __launch_bounds__(64) void unproject_bwd2_kernel(const float *__restrict__ cam,
                                                           const float *__restrict__ centers,
                                                           const uint8_t *__restrict__ valid,
                                                           const float *__restrict__ grad_cubes,
                                                           const uint16_t *__restrict__ pass_mask,
                                                           void *__restrict__ grad_packed_, size_t view_stride,
                                                           Geom g, int tiles_per_sample, const float *__restrict__ scale_p)
{
    using ACC = typename std::conditional<DET, unsigned long long, float>::type;
    ACC *grad_packed = reinterpret_cast<ACC *>(grad_packed_);
    const double scale = DET ? (double)*scale_p : 1.0;
    auto add = [&](ACC *p, float val) {
        if constexpr (DET) atomicAdd(p, (unsigned long long)__double2ll_rn((double)val * scale));
        else unsafeAtomicAdd(p, val);
    };
    extern __shared__ __attribute__((aligned(16))) float bsm[];
    float *rec = bsm;                                  // [V][5][64]
    int *reci = reinterpret_cast<int *>(rec);
    float *gt = bsm + g.V * 320;                       // [JP][64] gradient tile (0 where masked / beyond J)
    uint32_t *vm = reinterpret_cast<uint32_t *>(gt + JP * 64);   // [64] view bits per voxel (bit 31: NaN)
    int b, tile;
    if (XCD) {
        if (!xcd_map(blockIdx.x, g.B, tiles_per_sample, g.xcd_chunk, b, tile)) return;
    } else {
        b = blockIdx.x / tiles_per_sample;
        tile = blockIdx.x - b * tiles_per_sample;
    }
    const int n0 = tile * 64;
    if (n0 >= g.N || !valid[b]) return;
    const int bs = g.sample_of ? g.sample_of[b] : b;
    const int lane = threadIdx.x;
    const int nvox = min(64, g.N - n0);
    const bool inb = lane < nvox;
    const int n = n0 + (inb ? lane : 0);
    int vx, rem, vy, vz;
    udiv_magic((uint32_t)n, (uint32_t)g.YZ, g.magicYZ, vx, rem);
    udiv_magic((uint32_t)rem, (uint32_t)g.Z, g.magicZ, vy, vz);
    const float x = linspace_step(g.Lx, g.stepx, g.X, vx) + centers[3 * b + 0];
    const float y = linspace_step(g.Ly, g.stepy, g.Y, vy) + centers[3 * b + 1];
    const float z = linspace_step(g.Lz, g.stepz, g.Z, vz) + centers[3 * b + 2];
    uint32_t mymask = 0;
    for (int c = 0; c < g.V; ++c) {
        const float *cm = cam + ((size_t)bs * g.V + c) * SP3D_CAM_STRIDE;
        float ix, iy;
        bool isnan;
        const bool bound = sample_pos_fast(cm, x, y, z, g, ix, iy, isnan) && inb;
        if (bound) mymask |= (1u << c);
        if (isnan && inb) mymask |= 0x80000000u;
        const Rec r = make_record<JP>(bound && !isnan, isnan ? 0.0f : ix, isnan ? 0.0f : iy, g.w, g.h);
        const int base = c * 320 + lane;
        reci[base] = r.off;
        rec[base + 64] = r.w00; rec[base + 128] = r.w10; rec[base + 192] = r.w01; rec[base + 256] = r.w11;
    }
    // gradient tile: g = pass ? grad / den : 0     (autograd of project_layer.py:96-99)
    const uint32_t pm = inb ? (uint32_t)pass_mask[(size_t)b * g.N + n] : 0u;
    const float den = (float)__popc(mymask & 0x7fffffffu) + 1e-6f;
    const bool dead = (mymask & 0x80000000u) != 0 || (mymask & 0x7fffffffu) == 0;
    const float *gc = grad_cubes + (size_t)b * g.J * g.N + n;
    bool any = false;
    // the J gradient loads of a voxel in flight together (n is a valid voxel for every lane): inside the per-channel condition
    // they were JP dependent round trips per wave
    float gl[JP];
#pragma unroll
    for (int j = 0; j < JP; ++j) gl[j] = (j < g.J) ? gc[(size_t)j * g.N] : 0.0f;
    const bool live = inb && !dead;
#pragma unroll
    for (int j = 0; j < JP; ++j) {
        float v = 0.0f;
        if (j < g.J && live && ((pm >> j) & 1u)) v = gl[j] / den;
        any = any || (v != 0.0f);
        gt[j * 64 + lane] = v;
    }
    vm[lane] = any ? (mymask & 0x7fffffffu) : 0u;      // voxels without gradient scatter nothing
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // scatter: lane = (v4, ch)
    const int v4 = lane >> 4, ch = lane & 15;
    if (ch >= JP) return;
    const size_t rowf = (size_t)g.w * JP;
    ACC *gbase = grad_packed + (size_t)bs * g.h * rowf + ch;
#pragma unroll 1
    for (int m = 0; m < 16; ++m) {
        const int v = 4 * m + v4;
        uint32_t views = vm[v];
        const float gv = gt[ch * 64 + v];
        while (views) {
            const int c = __ffs((int)views) - 1;
            views &= views - 1;
            const int rb = c * 320 + v;
            ACC *p = gbase + (size_t)c * view_stride + reci[rb];
            const float w00 = rec[rb + 64], w10 = rec[rb + 128], w01 = rec[rb + 192], w11 = rec[rb + 256];
            if (w00 != 0.0f) add(p, gv * w00);
            if (w10 != 0.0f) add(p + JP, gv * w10);
            if (w01 != 0.0f) add(p + rowf, gv * w01);
            if (w11 != 0.0f) add(p + rowf + JP, gv * w11);
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward on DENSE grids (round 4, "bwd3"): a workgroup owns an 8x8x4 block of voxels and, per view, merges the block's tap
// gradients in an LDS patch of the heat-map gradient before they go to memory.
//
// What bounds the scatter (tools/global_atomic_bench.hip, profiles/r04_backward_kernels.md): memory atomics retire at
// ~20.7 G (instruction, 64-byte segment) pairs per second chip-wide - whatever the type (f32, u32, u64, f64, packed bf16),
// the scope, or how the segments of one instruction lie to each other; plain stores of the same segments are 4.6x
// faster.  bwd2 issues one segment per tap (4 per voxel and view).  On the 64^3 person cubes (31.7 mm pitch, ~1.7 heat-map
// pixels) the 1 024 taps of a block fall on ~180 distinct pixels of a ~18x12 rectangle: merged first, 4-5x fewer
// segments leave the CU.
// The merge cannot use fp32 LDS atomics: ds_add_f32 costs ~190 cycles per wave instruction per CU on gfx950, ds_add_u32 /
// ds_add_u64 cost 7 (tools/lds_atomic_bench.hip).  So the patch is 64-bit FIXED POINT: tap value * 2^k, rounded to
// nearest, with k from the block's largest |g| so that 2^50 steps span it (256 taps per pixel at most: no overflow); the
// patch sums are exact, and one rounding to fp32 happens when a pixel leaves (bwd2 rounds after every tap).  With the
// caller's global scale instead (DET) the flush adds the 64-bit sums to the fixed-point buffer with integer atomics:
// bit-identical to bwd2<DET>, run to run and to each other (integer addition is associative).
//   lane = voxel throughout (vz fastest).  gradient of the 16 channels in registers (one round trip: 16 loads in flight)
//   pass 1: project every view -> view mask / den, rectangle of the view's 2x2 tap blocks (LDS atomicMin/Max)
//   per view, per window of <= B3_PX pixels of the rectangle (almost always one): ds_add_u64 into patch[ch][pixel];
//   barrier; flush-and-clear, lanes = (pixel-of-4, channel): 64-byte segments, untouched pixels are skipped.
//   LDS: patch [JP][B3_PXS] int64 | rectangles [MAX_VIEWS][4] | block max           (33.5 KB: 4 workgroups per CU)
// ------------------------------------------------------------------------------------------
constexpr int B3_BX = 8, B3_BY = 8, B3_BZ = 4;
#ifndef SP3D_B3_PX
#define SP3D_B3_PX 256        // A/B on one box, us fp32 / deterministic: 128: 300 / 352, 192: 268 / 350, 256: 274 / 330, 384: 289 / 327, 512: 354 / 346
#endif
constexpr int B3_PX = SP3D_B3_PX;    // pixels of a patch window (a multiple of 32)
constexpr int B3_PXS = B3_PX + 4;    // plane stride (int64 words): == 4 mod 32, so the flush's (pixel-of-4, channel) lanes spread over the banks
#ifndef SP3D_B3_ABL
#define SP3D_B3_ABL 0            // measurement builds only: 1 no flush atomics, 2 no tap adds, 8 no view loop, 16 no gradient loads,
                                 // 32 no pass-1 projection, 64 no patch clear, 128 no divisions
#endif

template <int JP, bool DET>
__global__ __launch_bounds__(256, 4) void unproject_bwd3_kernel(const float *__restrict__ cam, const float *__restrict__ centers,
                                                            const uint8_t *__restrict__ valid,
                                                            const float *__restrict__ grad_cubes,
                                                            const uint16_t *__restrict__ pass_mask,
                                                            void *__restrict__ grad_acc_, size_t view_stride, Geom g,
                                                            int nbx, int nby, int nbz, const float *__restrict__ scale_p)
{
    using ACC = typename std::conditional<DET, unsigned long long, float>::type;
    ACC *grad_acc = reinterpret_cast<ACC *>(grad_acc_);
    extern __shared__ __attribute__((aligned(16))) unsigned long long psm3[];
    unsigned long long *patch = psm3;                                   // [JP][B3_PXS]
    int *rect = reinterpret_cast<int *>(patch + JP * B3_PXS);           // [MAX_VIEWS][4]: min x0, min y0, max x0 + 1, max y0 + 1
    uint32_t *bmax = reinterpret_cast<uint32_t *>(rect + 4 * SP3D_MAX_VIEWS);
    const int blocks_per_sample = nbx * nby * nbz;
    int b, blk;
    if (!xcd_map(blockIdx.x, g.B, blocks_per_sample, g.xcd_chunk, b, blk)) return;
    if (!valid[b]) return;
    const int bs = g.sample_of ? g.sample_of[b] : b;
    const int tid = threadIdx.x;
    const int bz = blk % nbz, by = (blk / nbz) % nby, bx = blk / (nbz * nby);
    const int vx = bx * B3_BX + (tid >> 5), vy = by * B3_BY + ((tid >> 2) & 7), vz = bz * B3_BZ + (tid & 3);
    const bool inb = vx < g.X && vy < g.Y && vz < g.Z;
    const int n = (min(vx, g.X - 1) * g.Y + min(vy, g.Y - 1)) * g.Z + min(vz, g.Z - 1);
    // gradient of this voxel, all channels: issued first, consumed after pass 1
    const float *gc = grad_cubes + (size_t)b * g.J * g.N + n;
    float gq[JP];
#pragma unroll
    for (int j = 0; j < JP; ++j) gq[j] = (SP3D_B3_ABL & 16) ? (float)(j + tid) : gc[(size_t)min(j, g.J - 1) * g.N];
    const uint32_t pm = inb ? (uint32_t)pass_mask[(size_t)b * g.N + n] : 0u;
    if (!(SP3D_B3_ABL & 64))
    for (int e = tid; e < JP * B3_PXS; e += 256) patch[e] = 0ull;
    if (tid < 4 * SP3D_MAX_VIEWS) rect[tid] = (tid & 3) < 2 ? 0x7fffffff : -1;
    if (tid == 0) *bmax = 0u;
    __syncthreads();

    const float x = linspace_step(g.Lx, g.stepx, g.X, min(vx, g.X - 1)) + centers[3 * b + 0];
    const float y = linspace_step(g.Ly, g.stepy, g.Y, min(vy, g.Y - 1)) + centers[3 * b + 1];
    const float z = linspace_step(g.Lz, g.stepz, g.Z, min(vz, g.Z - 1)) + centers[3 * b + 2];
    uint32_t mymask = (SP3D_B3_ABL & 32) ? 31u : 0u;
    for (int c = 0; c < ((SP3D_B3_ABL & 32) ? 0 : g.V); ++c) {
        const float *cm = cam + ((size_t)bs * g.V + c) * SP3D_CAM_STRIDE;
        float ix, iy;
        bool isnan;
        const bool bound = sample_pos_fast(cm, x, y, z, g, ix, iy, isnan) && inb;
        if (bound) mymask |= (1u << c);
        if (isnan && inb) mymask |= 0x80000000u;
        const bool use = bound && !isnan;
        const RecPk r = make_record_pk(use, v2f{isnan ? 0.0f : ix, isnan ? 0.0f : iy}, g.w, g.h);
        // rectangle: reduce in the wave first (64 lanes on ONE LDS word serialise: 230 us of the kernel when every
        // lane issued its own atomicMin/Max)
        int lo_x = use ? r.x0 : 0x7fffffff, lo_y = use ? r.y0 : 0x7fffffff, hi_x = use ? r.x0 + 1 : -1, hi_y = use ? r.y0 + 1 : -1;
        for (int o = 32; o > 0; o >>= 1) {
            lo_x = min(lo_x, __shfl_xor(lo_x, o)); lo_y = min(lo_y, __shfl_xor(lo_y, o));
            hi_x = max(hi_x, __shfl_xor(hi_x, o)); hi_y = max(hi_y, __shfl_xor(hi_y, o));
        }
        if ((tid & 63) == 0 && hi_x >= 0) {
            atomicMin(&rect[4 * c + 0], lo_x); atomicMin(&rect[4 * c + 1], lo_y);
            atomicMax(&rect[4 * c + 2], hi_x); atomicMax(&rect[4 * c + 3], hi_y);
        }
    }
    // g = pass ? grad / den : 0     (autograd of project_layer.py:96-99)
    const float den = (float)__popc(mymask & 0x7fffffffu) + 1e-6f;
    const bool dead = (mymask & 0x80000000u) != 0 || (mymask & 0x7fffffffu) == 0;
    uint32_t amax = 0u;
#pragma unroll
    for (int j = 0; j < JP; ++j) {
        float v = 0.0f;
        if (j < g.J && inb && !dead && ((pm >> j) & 1u)) v = (SP3D_B3_ABL & 128) ? gq[j] * den : gq[j] / den;
        gq[j] = v;
        amax = max(amax, __float_as_uint(v) & 0x7fffffffu);
    }
    const bool any = amax != 0u;                       // voxels without gradient scatter nothing
    if (!DET) {
        for (int o = 32; o > 0; o >>= 1) amax = max(amax, (uint32_t)__shfl_xor((int)amax, o));
        if ((tid & 63) == 0 && amax) atomicMax(bmax, amax);
    }
    __syncthreads();
    double scale, inv_scale = 1.0;
    bool nonfinite = false;     // uniform
    if (DET) {
        scale = (double)*scale_p;
    } else {
        const uint32_t m = *bmax;
        if (m == 0u) return;                           // no gradient anywhere in this block (uniform)
        if ((m >> 23) == 0xffu) {                      // Inf / NaN gradient in this block: no scale exists
            nonfinite = true;
            scale = 1.0;
        } else {
            const int k = min(50 - ((int)(m >> 23) - 126), 200);         // |g| < 2^(E - 126)  ->  |g| * 2^k < 2^50
            scale = __longlong_as_double((long long)(k + 1023) << 52);
            inv_scale = __longlong_as_double((long long)(1023 - k) << 52);
        }
    }
    if (SP3D_B3_ABL & 8) return;

    const size_t rowf = (size_t)g.w * JP;
#pragma unroll 1
    for (int c = 0; c < g.V; ++c) {
        const int rx0 = rect[4 * c + 0], ry0 = rect[4 * c + 1], rx1 = rect[4 * c + 2], ry1 = rect[4 * c + 3];
        if (rx1 < 0) continue;                          // nobody of this block sees view c (uniform)
        const float *cm = cam + ((size_t)bs * g.V + c) * SP3D_CAM_STRIDE;
        float ix, iy;
        bool isnan;
        const bool bound = sample_pos_fast(cm, x, y, z, g, ix, iy, isnan) && inb;
        const bool use = bound && !isnan;
        const RecPk r = make_record_pk(use, v2f{isnan ? 0.0f : ix, isnan ? 0.0f : iy}, g.w, g.h);
        const bool act = use && any;
        const float wts[4] = {r.wt.x, r.wt.y, r.wb.x, r.wb.y};
        ACC *gview = grad_acc + (size_t)c * view_stride + (size_t)bs * g.h * rowf;
        if (nonfinite) {
            // the block holds an Inf / NaN gradient: per-tap fp32 atomics straight to memory, as bwd2 adds them (the
            // non-finite value reaches exactly the pixels its voxel touches)
            if constexpr (!DET) {
                if (act) {
                    ACC *p0 = gview + ((size_t)r.y0 * g.w + r.x0) * JP;
#pragma unroll
                    for (int j = 0; j < JP; ++j) {
                        if (j >= g.J) break;
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            if (wts[t] != 0.0f) unsafeAtomicAdd(p0 + (size_t)(t >> 1) * rowf + (t & 1) * JP + j, gq[j] * wts[t]);
                    }
                }
            }
            continue;
        }
        // windows of the rectangle (one, unless the block's footprint in this view is unusually large)
        const int pw = rx1 - rx0 + 1, ph = ry1 - ry0 + 1;
        const int ww = min(pw, B3_PX), wh = min(ph, B3_PX / ww);
        const float rww = 1.0f / (float)ww;
#pragma unroll 1
        for (int wy0 = ry0; wy0 <= ry1; wy0 += wh) {
#pragma unroll 1
            for (int wx0 = rx0; wx0 <= rx1; wx0 += ww) {
                if (act && !(SP3D_B3_ABL & 2)) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int tx = r.x0 + (t & 1) - wx0, ty = r.y0 + (t >> 1) - wy0;
                        if (wts[t] == 0.0f || (unsigned)tx >= (unsigned)ww || (unsigned)ty >= (unsigned)wh) continue;
                        unsigned long long *pp = patch + ty * ww + tx;
                        // round-to-nearest-even of t * 2^k to int64 without a conversion sequence: the sum with 1.5 * 2^52
                        // holds the integer in its low mantissa bits (|t * 2^k| < 2^50); taking the constant's bit
                        // pattern off again touches only the high word.  == __double2ll_rn((double)t * scale) of bwd2<DET>.
#pragma unroll
                        for (int j = 0; j < JP; ++j) {
                            if (j >= g.J) break;                                    // uniform: the pad channels carry nothing (testing only the last three is 5 % slower)
                            const double d = __builtin_fma((double)(gq[j] * wts[t]), scale, 6755399441055744.0);
                            atomicAdd(pp + j * B3_PXS, (unsigned long long)__double_as_longlong(d) - 0x4338000000000000ull);
                        }
                    }
                }
                __syncthreads();
                // flush and clear: element e = (pixel, channel), 64 lanes = 4 pixels x 16 channels = 4 segments of 64 bytes
                const int nwx = min(ww, rx1 - wx0 + 1), nwy = min(wh, ry1 - wy0 + 1);
                const int nel = nwy * ww * 16;
                for (int e = tid; e < nel; e += 256) {
                    const int px = e >> 4, ch = e & 15;
                    if (ch >= JP) continue;
                    const long long val = (long long)patch[ch * B3_PXS + px];
                    if (val == 0) continue;
                    patch[ch * B3_PXS + px] = 0ull;
                    const int ty = (int)(((float)px + 0.5f) * rww), tx = px - ty * ww;
                    if (tx >= nwx || (SP3D_B3_ABL & 1)) continue;
                    ACC *dst = gview + ((size_t)(wy0 + ty) * g.w + (wx0 + tx)) * JP + ch;
                    if constexpr (DET) atomicAdd(dst, (unsigned long long)val);
                    else unsafeAtomicAdd(dst, (float)((double)val * inv_scale));
                }
                __syncthreads();
            }
        }
    }
}

__global__ __launch_bounds__(256) void fixed_to_float_kernel(const long long *__restrict__ acc, float *__restrict__ out,
                                                            const float *__restrict__ scale_p, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = (float)((double)acc[i] / (double)*scale_p);
}

// ------------------------------------------------------------------------------------------
// host-side helpers
// ------------------------------------------------------------------------------------------
static int make_geom(Geom &g, int B, int V, int J, int h, int w, int X, int Y, int Z, const float *grid_size,
                     int W_in, int H_in)
{
    if (B <= 0 || V <= 0 || J <= 0 || h <= 0 || w <= 0 || X <= 0 || Y <= 0 || Z <= 0 || W_in <= 0 || H_in <= 0)
        return SP3D_EINVAL;
    if (V > SP3D_MAX_VIEWS) return SP3D_EINVAL;
    if (!grid_size) return SP3D_ENULL;
    const int64_t N = (int64_t)X * Y * Z;
    if (N > (int64_t)0x7fffffff - TILE) return SP3D_ERANGE;
    if ((int64_t)B * ((N + TILE - 1) / TILE) > (int64_t)0x7fffffff - 8) return SP3D_ERANGE;
    if ((int64_t)h * w * 16 > (int64_t)0x7fffffff) return SP3D_ERANGE;
    g.B = B; g.V = V; g.J = J; g.h = h; g.w = w; g.X = X; g.Y = Y; g.Z = Z;
    g.sample_of = nullptr;
    g.pass_mask = nullptr;
    g.xcd_chunk = 1;
    g.xcd_order = 0;
    g.xm_mode = 2; g.xm_log2xps = g.xm_log2K = g.xm_rows = 0; g.xm_tiles = 1; g.xm_magic_tiles = 0;
    g.bk_nxy = g.bk_nby = 1; g.bk_magic_nxy = g.bk_magic_nby = 0;
    g.blk_log2py = 0; g.blk_w = g.blk_h = g.blk_nbx = g.blk_nzc = 1; g.blk_magic_wh = g.blk_magic_h = 0;
    g.N = (int)N; g.YZ = Y * Z; g.W_in = W_in; g.H_in = H_in;
    g.sB = (long long)J * N; g.sJ = (int)N; g.sX = Y * Z; g.sY = Z; g.dense = 1; g.vec4 = 1;
    g.Lx = grid_size[0]; g.Ly = grid_size[1]; g.Lz = grid_size[2];
    g.rW_in = 1.0f / (float)W_in; g.rH_in = 1.0f / (float)H_in;
    g.rw1 = w > 1 ? 1.0f / (float)(w - 1) : 0.0f; g.rh1 = h > 1 ? 1.0f / (float)(h - 1) : 0.0f;
    {   // torch.linspace step in fp32: (end - start) / (n - 1) with start = -(L/2), end = L/2
        const float L[3] = {g.Lx, g.Ly, g.Lz};
        const int n[3] = {X, Y, Z};
        float st[3];
        for (int a = 0; a < 3; ++a) {
            volatile float start = -(L[a] / 2.0f), end = L[a] / 2.0f;
            volatile float diff = end - start;
            st[a] = n[a] > 1 ? diff / (float)(n[a] - 1) : 0.0f;
        }
        g.stepx = st[0]; g.stepy = st[1]; g.stepz = st[2];
    }
    g.magicYZ = (uint32_t)((0x100000000ull / (uint64_t)(Y * Z)) + 1ull);
    g.magicZ = (uint32_t)((0x100000000ull / (uint64_t)Z) + 1ull);
    return SP3D_OK;
}

static int load_views(Views &v, const float *const *hm_views, int V)
{
    if (!hm_views) return SP3D_ENULL;
    for (int c = 0; c < SP3D_MAX_VIEWS; ++c) v.p[c] = nullptr;
    for (int c = 0; c < V; ++c) {
        if (!hm_views[c]) return SP3D_ENULL;
        v.p[c] = hm_views[c];
    }
    return SP3D_OK;
}

static int launch_status()
{
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SP3D_OK : (int)e;
}

// variant: bits[1:0] voxels in flight per lane (0:1, 1:2, 2:4); bit 2: disable the XCD-aware tile map;
// bit 3: per-wave software-pipelined kernel (unproject_pipe_kernel); bit 4: one wave per workgroup
// default: pipelined kernel; XCD-aware tile map only when several samples share the chip
// Default kernel per result layout and grid (profiles/r02_ab_brick.json, same-box A/B, bit-identical results):
//   channels-last result     4x4x4 bricks, one brick per workgroup (120): -5 % on the root grid, -20 % on 64^3 person
//                            cubes, -30 % on the 160x160x40 grid - fewer distinct 128-B lines per wave-load
//   planar result, Z % 32==0 brick stacks of 8 (56): the workgroup store writes whole 128-B z-runs (64^3 cubes: -20 %)
//   planar result, other Z   64 consecutive voxels per wave (24): a 20- or 40-voxel z-run does not fill store lines from a
//                            brick stack, and the stack's barrier costs more than the gather saves (root grid: +30 %)
static int default_variant(const Geom &g, bool out_cl)
{
    if (g.w < 2 || g.h < 2) return 24;
    if (out_cl) return 120;
    return (g.Z % 32 == 0) ? 56 : 24;
}

template <int JP>
static int launch_nhwc_jp(const Views &v, const float *cam, const float *centers, const uint8_t *valid, float *cubes,
                          float *grids, const Geom &g, int variant, bool out_cl, int io, hipStream_t s)
{
    const int tiles = (g.N + TILE - 1) / TILE;
    const int total = tiles * g.B;
    const size_t lds = (size_t)(JP * OSTR + 2 * g.V * TILE + TILE) * sizeof(float);
    const bool xcd = !(variant & 4);
    dim3 grid(xcd ? xcd_grid_blocks(g.B, tiles, g.xcd_chunk) : total), block(TILE);
    if (variant & 32) {      // brick kernel: 4x4x4 voxels per wave, a z-stack of bricks per workgroup
        const int nbx = (g.X + BR - 1) / BR, nby = (g.Y + BR - 1) / BR, nwz = (g.Z + BR - 1) / BR;
        int nzc = (nwz + 7) / 8, zw = (nwz + nzc - 1) / nzc;
        if (variant & 64) { zw = 1; nzc = nwz; }                  // tuning: every brick its own workgroup
        const int wgs = nbx * nby * nzc;
        Geom gb = g;
        {   // 2-4 chunks of consecutive workgroups (x-slabs of the volume) per serving XCD
            const int xps = (g.B <= 8 && (8 % g.B) == 0) ? 8 / g.B : 1;
            int k = 1;
            while (k * 2 * xps * 2 <= wgs) k *= 2;
            if ((variant >> 17) & 15) k = 1 << (((variant >> 17) & 15) - 1);
            gb.xcd_chunk = k;
        }
        set_xcd_fields(gb, wgs);
        set_brick_fields(gb, nbx * nby, nby);
        // default since round 6 (B in {1, 2, 4}): one block of brick columns per XCD - octants at B = 1, quadrants at B = 2,
        // halves at B = 4 - instead of round-robin chunks; same results, L2 fills 138 -> 60 MB on the 160x160x40 grid,
        // 75 -> 56 MB on the root grid at B = 4 (profiles/r06_pmc_blocks.json).  Tuning bit 22 restores the chunk map.
        const int block_grid = (variant & SP3D_VARIANT_CHUNKS) || (variant & 256) ? 0 : set_block_fields(gb, nbx, nby, nzc);
        constexpr int WLDS = (JP * WOSTR > WREC) ? JP * WOSTR : WREC;
        size_t blds = (size_t)zw * WLDS * sizeof(float);
        // round-5 L1-residency experiment (measurement only): tuning bit 10 = view-synchronous workgroups (only when every
        // wave of every workgroup lies inside the volume, so that all of them reach the per-view barrier), bits 11-13 = n:
        // n * 20 KB of unused LDS per workgroup, which caps the workgroups resident on a CU
        if (((variant >> 10) & 1) && nwz % zw == 0 && nzc * zw == nwz) gb.xcd_order |= 4;
        const int ballast = (variant >> 11) & 7;
        if (ballast) {
            blds += (size_t)ballast * 20480;
            if (blds > 65536) {
                const void *fn = out_cl ? reinterpret_cast<const void *>(unproject_brick_kernel<JP, true, float, float>)
                                        : reinterpret_cast<const void *>(unproject_brick_kernel<JP, false, float, float>);
                const hipError_t ea = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)blds);
                if (ea != hipSuccess) return (int)ea;
            }
        }
        dim3 bgrid(block_grid ? block_grid : xcd_grid_blocks(gb.B, wgs, gb.xcd_chunk)), bblock(64 * zw);
        if (variant & SP3D_VARIANT_ZD) {   // the stack's cubes leave as their z-spectrum (sp3d_unproject_fwd_zdft)
            if constexpr (JP == 16) {
                if (io || out_cl || grids || g.Z != ZDZ || nzc != 1 || zw != ZDZ / BR || (g.X % BR) || (g.Y % BR) || g.pass_mask)
                    return SP3D_EUNSUPPORTED;
                hipLaunchKernelGGL((unproject_brick_kernel<16, false, float, float, true>), bgrid, bblock, blds, s, v, cam, centers,
                                   valid, cubes, grids, gb, wgs, nby, nzc, zw);
                return SP3D_OK;
            } else {
                return SP3D_EUNSUPPORTED;
            }
        }
#define SP3D_BRICK(CL_, TI_, TO_) \
    hipLaunchKernelGGL((unproject_brick_kernel<JP, CL_, TI_, TO_>), bgrid, bblock, blds, s, v, cam, centers, valid, cubes, grids, gb, wgs, nby, nzc, zw)
        if ((io & 1) && JP == 16 && !((variant >> 9) & 1)) {
            // bf16 heat-maps: two lanes per pixel (tuning bit 9: keep the four-lane kernel below, for A/B)
#define SP3D_BRICK_H(CL_, TO_) \
    hipLaunchKernelGGL((unproject_brick_h_kernel<CL_, TO_>), bgrid, bblock, blds, s, v, cam, centers, valid, cubes, grids, gb, wgs, nby, nzc, zw)
            if (io & 2) { if (out_cl) SP3D_BRICK_H(true, bf16_t); else SP3D_BRICK_H(false, bf16_t); }
            else { if (out_cl) SP3D_BRICK_H(true, float); else SP3D_BRICK_H(false, float); }
#undef SP3D_BRICK_H
            return SP3D_OK;
        }
        if (io != 0) {
            if constexpr (JP == 16) {
                switch ((io & 3) * 2 + (out_cl ? 1 : 0)) {
                case 2: SP3D_BRICK(false, bf16_t, float); break;
                case 3: SP3D_BRICK(true, bf16_t, float); break;
                case 4: SP3D_BRICK(false, float, bf16_t); break;
                case 5: SP3D_BRICK(true, float, bf16_t); break;
                case 6: SP3D_BRICK(false, bf16_t, bf16_t); break;
                default: SP3D_BRICK(true, bf16_t, bf16_t); break;
                }
                return SP3D_OK;
            } else {
                return SP3D_EUNSUPPORTED;
            }
        }
        if (out_cl) SP3D_BRICK(true, float, float); else SP3D_BRICK(false, float, float);
#undef SP3D_BRICK
        return SP3D_OK;
    }
    if (variant & 8) {
        const int nw = (variant & 16) ? 1 : 4;
        const int ptiles = (g.N + 64 * nw - 1) / (64 * nw);
        const int ptotal = ptiles * g.B;
        dim3 pgrid(xcd ? xcd_grid_blocks(g.B, ptiles, g.xcd_chunk) : ptotal), pblock(64 * nw);
        Geom gp = g;
        set_xcd_fields(gp, ptiles);
#define SP3D_PIPE(XCD_, NW_, CL_) \
    hipLaunchKernelGGL((unproject_pipe_kernel<JP, XCD_, NW_, CL_>), pgrid, pblock, 0, s, v, cam, centers, valid, cubes, grids, gp, ptiles, ptotal)
#define SP3D_PIPE_T(XCD_, CL_, TI_, TO_) \
    hipLaunchKernelGGL((unproject_pipe_kernel<JP, XCD_, 1, CL_, TI_, TO_>), pgrid, pblock, 0, s, v, cam, centers, valid, cubes, grids, gp, ptiles, ptotal)
        if (io != 0) {   // bf16 storage variants: JP == 16, one wave per workgroup only
            if constexpr (JP == 16) {
                if (nw != 1) return SP3D_EUNSUPPORTED;
                const int sel = (io & 3) * 4 + (out_cl ? 2 : 0) + (xcd ? 1 : 0);
                switch (sel) {
                case 4: SP3D_PIPE_T(false, false, bf16_t, float); break;
                case 5: SP3D_PIPE_T(true, false, bf16_t, float); break;
                case 6: SP3D_PIPE_T(false, true, bf16_t, float); break;
                case 7: SP3D_PIPE_T(true, true, bf16_t, float); break;
                case 8: SP3D_PIPE_T(false, false, float, bf16_t); break;
                case 9: SP3D_PIPE_T(true, false, float, bf16_t); break;
                case 10: SP3D_PIPE_T(false, true, float, bf16_t); break;
                case 11: SP3D_PIPE_T(true, true, float, bf16_t); break;
                case 12: SP3D_PIPE_T(false, false, bf16_t, bf16_t); break;
                case 13: SP3D_PIPE_T(true, false, bf16_t, bf16_t); break;
                case 14: SP3D_PIPE_T(false, true, bf16_t, bf16_t); break;
                default: SP3D_PIPE_T(true, true, bf16_t, bf16_t); break;
                }
                return SP3D_OK;
            } else {
                return SP3D_EUNSUPPORTED;
            }
        }
        if (out_cl) {
            if (nw == 1) { if (xcd) SP3D_PIPE(true, 1, true); else SP3D_PIPE(false, 1, true); }
            else { if (xcd) SP3D_PIPE(true, 4, true); else SP3D_PIPE(false, 4, true); }
        } else {
            if (nw == 1) { if (xcd) SP3D_PIPE(true, 1, false); else SP3D_PIPE(false, 1, false); }
            else { if (xcd) SP3D_PIPE(true, 4, false); else SP3D_PIPE(false, 4, false); }
        }
#undef SP3D_PIPE_T
#undef SP3D_PIPE
        return SP3D_OK;
    }
    if (out_cl || io) return SP3D_EUNSUPPORTED;
#define SP3D_LAUNCH(XCD_, U_) \
    hipLaunchKernelGGL((unproject_nhwc_kernel<JP, XCD_, U_>), grid, block, lds, s, v, cam, centers, valid, cubes, grids, g, tiles, total)
    switch (variant & 3) {
    case 0: if (xcd) SP3D_LAUNCH(true, 1); else SP3D_LAUNCH(false, 1); break;
    case 2: if (xcd) SP3D_LAUNCH(true, 4); else SP3D_LAUNCH(false, 4); break;
    default: if (xcd) SP3D_LAUNCH(true, 2); else SP3D_LAUNCH(false, 2); break;
    }
#undef SP3D_LAUNCH
    return SP3D_OK;
}

// io: bit 0 = packed heat-maps are bf16, bit 1 = cubes are bf16
static int launch_nhwc(const Views &v, int Jp, const float *cam, const float *centers, const uint8_t *valid,
                       float *cubes, float *grids, const Geom &g_in, int variant, bool out_cl, int io, hipStream_t s)
{
    Geom g = g_in;
    // chunk order: centre of the volume first (cheap edge tiles form the tail) when a sample is spread over >= 4 XCDs
    // (-2.5 % at B = 1); with 2 XCDs per sample it buys no time and costs L2 locality (HBM-side reads 75 -> 92 MB on
    // the bench workload), so the plain sweep stays there.  Tuning bit 21 forces the sweep.
    g.xcd_order = ((!((variant >> 21) & 1) && g.B <= 2) ? 1 : 0) | ((variant & 256) ? 2 : 0);
    if ((variant >> 17) & 15) {
        g.xcd_chunk = 1 << (((variant >> 17) & 15) - 1);   // tuning bits 17-20: log2(K)+1
    } else {
        // default: 2-4 chunks per serving XCD - compact enough that an XCD's L2 sees a fraction of each
        // view (fabric reads 93 MB -> 81 MB on the bench workload), fine enough to balance visibility
        const int xps = (g.B <= 8 && (8 % g.B) == 0) ? 8 / g.B : 1;
        const int t64 = (g.N + 63) / 64;
        int k = 1;
        while (k * 2 * xps * 2 <= t64) k *= 2;
        g.xcd_chunk = k;
    }
    if (Jp < g.J || (Jp & 3) || Jp > 16) return SP3D_EUNSUPPORTED;
    if (out_cl && (g.J & 3)) return SP3D_EUNSUPPORTED;           // channels-last rows must be 16-B multiples
    if (g.w < 2 || g.h < 2) variant &= ~(8 | 32);                // the clamped 2x2 block needs a 2x2 image
    if ((int64_t)g.h * g.w > (1 << 24)) variant &= ~(8 | 32);    // the pipelined kernels form pixel indices with 24-bit multiplies
    variant &= ~128;                                             // (round-3 LDS patch kernels: measured slower, removed in round 4)
    if (variant & 32) variant |= 8;
    if (io && !(variant & 8)) return SP3D_EUNSUPPORTED;
    if (io) variant |= 16;
    int rc;
    switch (Jp) {
    case 4: rc = launch_nhwc_jp<4>(v, cam, centers, valid, cubes, grids, g, variant, out_cl, io, s); break;
    case 8: rc = launch_nhwc_jp<8>(v, cam, centers, valid, cubes, grids, g, variant, out_cl, io, s); break;
    case 12: rc = launch_nhwc_jp<12>(v, cam, centers, valid, cubes, grids, g, variant, out_cl, io, s); break;
    case 16: rc = launch_nhwc_jp<16>(v, cam, centers, valid, cubes, grids, g, variant, out_cl, io, s); break;
    default: return SP3D_EUNSUPPORTED;
    }
    return rc ? rc : launch_status();
}

} // namespace sp3d

using namespace sp3d;

extern "C" int sp3d_abi_version(void) { return SP3D_ABI_VERSION; }

extern "C" int sp3d_camera_finish(float *t, int records)
{
    if (!t) return SP3D_ENULL;
    if (records < 0) return SP3D_EINVAL;
    for (int r = 0; r < records; ++r, t += SP3D_CAM_STRIDE) {
        for (int c = 0; c < 3; ++c) {
            t[SP3D_CAM_RXY + 2 * c] = t[SP3D_CAM_R + c]; t[SP3D_CAM_RXY + 2 * c + 1] = t[SP3D_CAM_R + 3 + c];
            t[SP3D_CAM_AXY + 2 * c] = t[SP3D_CAM_A + c]; t[SP3D_CAM_AXY + 2 * c + 1] = t[SP3D_CAM_A + 3 + c];
            t[SP3D_CAM_RZ + c] = t[SP3D_CAM_R + 6 + c];
            t[SP3D_CAM_K2 + c] = t[SP3D_CAM_K + c];
        }
        t[SP3D_CAM_TXY] = t[SP3D_CAM_T]; t[SP3D_CAM_TXY + 1] = t[SP3D_CAM_T + 1]; t[SP3D_CAM_TZ] = t[SP3D_CAM_T + 2];
        t[SP3D_CAM_P2] = t[SP3D_CAM_P]; t[SP3D_CAM_P2 + 1] = t[SP3D_CAM_P + 1];
        t[SP3D_CAM_F2] = t[SP3D_CAM_F]; t[SP3D_CAM_F2 + 1] = t[SP3D_CAM_F + 1];
        t[SP3D_CAM_C2] = t[SP3D_CAM_C]; t[SP3D_CAM_C2 + 1] = t[SP3D_CAM_C + 1];
        t[SP3D_CAM_WH] = t[SP3D_CAM_W0]; t[SP3D_CAM_WH + 1] = t[SP3D_CAM_H0];
        t[SP3D_CAM_FLIP2] = t[SP3D_CAM_FLIP];
        uint32_t m = 0;                     // the per-view test of the projection: every |A| <= 1e30f, as integers (NaN / inf are larger)
        for (int i = 0; i < 6; ++i) {
            uint32_t a;
            memcpy(&a, &t[SP3D_CAM_A + i], 4);
            a &= 0x7fffffffu;
            m = a > m ? a : m;
        }
        t[SP3D_CAM_TAME] = m <= 0x7149f2cau ? 1.0f : 0.0f;
        t[30] = t[31] = t[63] = 0.0f;
    }
    return SP3D_OK;
}

extern "C" const char *sp3d_error_string(int code)
{
    switch (code) {
    case SP3D_OK: return "ok";
    case SP3D_EINVAL: return "invalid argument (dimension <= 0, too many views, unknown layout)";
    case SP3D_ENULL: return "required pointer is NULL";
    case SP3D_ERANGE: return "size overflows 32-bit kernel indexing";
    case SP3D_EUNSUPPORTED: return "unsupported combination";
    case SP3D_EFFT: return "hipFFT plan creation or execution failed";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown sp3d error";
    }
}

extern "C" int sp3d_pack_heatmaps_ex(const void *const *hm_views, void *packed, int in_bf16, int out_bf16, int B, int V,
                                     int J, int Jp, int h, int w, void *stream)
{
    if (B <= 0 || V <= 0 || J <= 0 || h <= 0 || w <= 0 || V > SP3D_MAX_VIEWS) return SP3D_EINVAL;
    if (!packed) return SP3D_ENULL;
    if (Jp < J || (Jp & 3)) return SP3D_EUNSUPPORTED;
    Views v;
    int rc = load_views(v, reinterpret_cast<const float *const *>(hm_views), V);
    if (rc) return rc;
    const int HW = h * w;
    dim3 grid((HW + 255) / 256, B, V), block(256);
    hipStream_t s = (hipStream_t)stream;
    float *pk = reinterpret_cast<float *>(packed);
    if (!in_bf16 && !out_bf16) {
        switch (Jp) {
        case 4: hipLaunchKernelGGL(pack_nhwc_kernel<4>, grid, block, 0, s, v, pk, B, J, HW); break;
        case 8: hipLaunchKernelGGL(pack_nhwc_kernel<8>, grid, block, 0, s, v, pk, B, J, HW); break;
        case 12: hipLaunchKernelGGL(pack_nhwc_kernel<12>, grid, block, 0, s, v, pk, B, J, HW); break;
        case 16: hipLaunchKernelGGL(pack_nhwc_kernel<16>, grid, block, 0, s, v, pk, B, J, HW); break;
        default: return SP3D_EUNSUPPORTED;
        }
    } else {
        if (Jp != 16) return SP3D_EUNSUPPORTED;
        if (in_bf16 && out_bf16) hipLaunchKernelGGL((pack_nhwc_kernel<16, bf16_t, bf16_t>), grid, block, 0, s, v, pk, B, J, HW);
        else if (in_bf16) hipLaunchKernelGGL((pack_nhwc_kernel<16, bf16_t, float>), grid, block, 0, s, v, pk, B, J, HW);
        else hipLaunchKernelGGL((pack_nhwc_kernel<16, float, bf16_t>), grid, block, 0, s, v, pk, B, J, HW);
    }
    return launch_status();
}

extern "C" int sp3d_pack_heatmaps(const float *const *hm_views, float *packed, int B, int V, int J, int Jp, int h,
                                  int w, void *stream)
{
    return sp3d_pack_heatmaps_ex(reinterpret_cast<const void *const *>(hm_views), packed, 0, 0, B, V, J, Jp, h, w, stream);
}

extern "C" int sp3d_unproject_fwd_indexed(const float *const *hm_views, int hm_layout, int Jp, const float *cam,
                                          const int32_t *sample_of, const float *centers, const uint8_t *valid,
                                          float *cubes, float *grids, int P, int V, int J, int h, int w, int X, int Y,
                                          int Z, const float *grid_size, int W_in, int H_in, void *stream)
{
    Geom g;
    int rc = make_geom(g, P, V, J, h, w, X, Y, Z, grid_size, W_in, H_in);
    if (rc) return rc;
    if (!cam || !centers || !valid || !cubes) return SP3D_ENULL;
    g.sample_of = sample_of;
    Views v;
    rc = load_views(v, hm_views, V);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int tiles = (g.N + TILE - 1) / TILE;
    const bool out_cl = (hm_layout & SP3D_OUT_CHANNELS_LAST) != 0;
    const int io = ((hm_layout & SP3D_HM_BF16) ? 1 : 0) | ((hm_layout & SP3D_OUT_BF16) ? 2 : 0);
    hm_layout &= 0xff;
    if (hm_layout == SP3D_LAYOUT_PLANAR) {
        if (out_cl || io) return SP3D_EUNSUPPORTED;
        dim3 grid(tiles, P), block(TILE);
        if (J == 1)
            hipLaunchKernelGGL(unproject_planar_kernel<1>, grid, block, 0, s, v, cam, centers, valid, cubes, grids, g);
        else if (J <= 4)
            hipLaunchKernelGGL(unproject_planar_kernel<4>, grid, block, 0, s, v, cam, centers, valid, cubes, grids, g);
        else
            hipLaunchKernelGGL(unproject_planar_kernel<16>, grid, block, 0, s, v, cam, centers, valid, cubes, grids, g);
        return launch_status();
    }
    if (hm_layout == SP3D_LAYOUT_NHWC) return launch_nhwc(v, Jp, cam, centers, valid, cubes, grids, g, default_variant(g, out_cl), out_cl, io, s);
    return SP3D_EINVAL;
}

extern "C" int sp3d_unproject_fwd_strided(const float *const *hm_views, int hm_layout, int Jp, const float *cam,
                                          const int32_t *sample_of, const float *centers, const uint8_t *valid,
                                          float *cubes, const int64_t *out_strides, int P, int V, int J, int h, int w,
                                          int X, int Y, int Z, const float *grid_size, int W_in, int H_in, void *stream)
{
    Geom g;
    int rc = make_geom(g, P, V, J, h, w, X, Y, Z, grid_size, W_in, H_in);
    if (rc) return rc;
    if (!cam || !centers || !valid || !cubes || !out_strides) return SP3D_ENULL;
    const bool out_cl = (hm_layout & SP3D_OUT_CHANNELS_LAST) != 0;
    const int io = ((hm_layout & SP3D_HM_BF16) ? 1 : 0) | ((hm_layout & SP3D_OUT_BF16) ? 2 : 0);
    if ((hm_layout & 0xff) != SP3D_LAYOUT_NHWC || out_cl || w < 2 || h < 2) return SP3D_EUNSUPPORTED;
    const int64_t sB = out_strides[0], sJ = out_strides[1], sX = out_strides[2], sY = out_strides[3];
    // the planes must not overlap and must fit 32-bit in-plane offsets
    if (sY < Z || sX < (int64_t)Y * sY || sJ < (int64_t)X * sX || sB < (int64_t)J * sJ) return SP3D_EINVAL;
    if (sJ > 0x7fffffff) return SP3D_ERANGE;
    g.sB = sB; g.sJ = (int)sJ; g.sX = (int)sX; g.sY = (int)sY;
    g.dense = (sY == Z && sX == (int64_t)Y * Z && sJ == (int64_t)g.N && sB == (int64_t)J * g.N) ? 1 : 0;
    // 16-byte pieces need 4-element aligned rows (and a 16-byte aligned base pointer); otherwise the kernels take their
    // scalar store path, which is correct for any stride but slow
    g.sample_of = sample_of;
    Views v;
    rc = load_views(v, hm_views, V);
    if (rc) return rc;
    g.vec4 = ((g.dense || ((sY | sX | sJ | sB) & 3) == 0) && ((uintptr_t)cubes & 15) == 0) ? 1 : 0;
    return launch_nhwc(v, Jp, cam, centers, valid, cubes, nullptr, g, default_variant(g, false), false, io, (hipStream_t)stream);
}

extern "C" int sp3d_unproject_fwd_zdft(const float *const *hm_views, int Jp, const float *cam, const float *centers,
                                       const uint8_t *valid, float *spec, int B, int V, int J, int h, int w, int X, int Y,
                                       int Z, const float *grid_size, int W_in, int H_in, int SZ, void *stream)
{
    Geom g;
    int rc = make_geom(g, B, V, J, h, w, X, Y, Z, grid_size, W_in, H_in);
    if (rc) return rc;
    if (!cam || !centers || !valid || !spec) return SP3D_ENULL;
    if (Jp != 16 || Z != ZDZ || SZ != ZDSZ || (X % BR) || (Y % BR) || w < 2 || h < 2 || (int64_t)h * w > (1 << 24) ||
        ((uintptr_t)spec & 127))
        return SP3D_EUNSUPPORTED;
    Views v;
    rc = load_views(v, hm_views, V);
    if (rc) return rc;
    return launch_nhwc(v, Jp, cam, centers, valid, spec, nullptr, g, 56 | SP3D_VARIANT_ZD, false, 0, (hipStream_t)stream);
}

extern "C" int sp3d_unproject_fwd(const float *const *hm_views, int hm_layout, int Jp, const float *cam,
                                  const float *centers, const uint8_t *valid, float *cubes, float *grids, int B,
                                  int V, int J, int h, int w, int X, int Y, int Z, const float *grid_size, int W_in,
                                  int H_in, void *stream)
{
    return sp3d_unproject_fwd_indexed(hm_views, hm_layout, Jp, cam, nullptr, centers, valid, cubes, grids, B, V, J, h,
                                      w, X, Y, Z, grid_size, W_in, H_in, stream);
}

extern "C" int sp3d_unproject_bwd_indexed(const float *const *hm_views, const float *cam, const int32_t *sample_of,
                                          const float *centers, const uint8_t *valid, const float *grad_cubes,
                                          float *const *grad_hm_views, int P, int V, int J, int h, int w, int X, int Y,
                                          int Z, const float *grid_size, int W_in, int H_in, void *stream)
{
    Geom g;
    int rc = make_geom(g, P, V, J, h, w, X, Y, Z, grid_size, W_in, H_in);
    if (rc) return rc;
    if (!cam || !centers || !valid || !grad_cubes || !grad_hm_views) return SP3D_ENULL;
    g.sample_of = sample_of;
    Views v;
    rc = load_views(v, hm_views, V);
    if (rc) return rc;
    ViewsMut gv;
    for (int c = 0; c < SP3D_MAX_VIEWS; ++c) gv.p[c] = nullptr;
    for (int c = 0; c < V; ++c) {
        if (!grad_hm_views[c]) return SP3D_ENULL;
        gv.p[c] = grad_hm_views[c];
    }
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((g.N + TILE - 1) / TILE, P), block(TILE);
    if (J == 1)
        hipLaunchKernelGGL(unproject_bwd_kernel<1>, grid, block, 0, s, v, cam, centers, valid, grad_cubes, gv, g);
    else if (J <= 4)
        hipLaunchKernelGGL(unproject_bwd_kernel<4>, grid, block, 0, s, v, cam, centers, valid, grad_cubes, gv, g);
    else
        hipLaunchKernelGGL(unproject_bwd_kernel<16>, grid, block, 0, s, v, cam, centers, valid, grad_cubes, gv, g);
    return launch_status();
}

extern "C" int sp3d_unproject_bwd(const float *const *hm_views, const float *cam, const float *centers,
                                  const uint8_t *valid, const float *grad_cubes, float *const *grad_hm_views, int B,
                                  int V, int J, int h, int w, int X, int Y, int Z, const float *grid_size, int W_in,
                                  int H_in, void *stream)
{
    return sp3d_unproject_bwd_indexed(hm_views, cam, nullptr, centers, valid, grad_cubes, grad_hm_views, B, V, J, h, w,
                                      X, Y, Z, grid_size, W_in, H_in, stream);
}

// Not part of the drop-in ABI (declared in csrc/sp3d_tuning.h): same as sp3d_unproject_fwd for
// the NHWC layout, with an explicit kernel variant, for A/B measurements (tools/ab_variants.py).
extern "C" int sp3d_unproject_fwd_variant(const float *const *hm_views, int Jp, const float *cam, const float *centers,
                                          const uint8_t *valid, float *cubes, float *grids, int B, int V, int J,
                                          int h, int w, int X, int Y, int Z, const float *grid_size, int W_in,
                                          int H_in, int variant, void *stream)
{
    Geom g;
    int rc = make_geom(g, B, V, J, h, w, X, Y, Z, grid_size, W_in, H_in);
    if (rc) return rc;
    if (!cam || !centers || !valid || !cubes) return SP3D_ENULL;
    Views v;
    rc = load_views(v, hm_views, V);
    if (rc) return rc;
    return launch_nhwc(v, Jp, cam, centers, valid, cubes, grids, g, variant & 0xffffff, (variant & 0x1000000) != 0, 0, (hipStream_t)stream);
}

// forward with the gradient pass mask (uint16 per voxel, bit j = channel j passes gradient) for
// sp3d_unproject_bwd_packed.  NHWC fp32 input only (the pipelined kernel).
extern "C" int sp3d_unproject_fwd_train(const float *const *hm_views, int hm_layout, int Jp, const float *cam,
                                        const int32_t *sample_of, const float *centers, const uint8_t *valid,
                                        float *cubes, float *grids, uint16_t *pass_mask, int P, int V, int J, int h,
                                        int w, int X, int Y, int Z, const float *grid_size, int W_in, int H_in,
                                        void *stream)
{
    Geom g;
    int rc = make_geom(g, P, V, J, h, w, X, Y, Z, grid_size, W_in, H_in);
    if (rc) return rc;
    if (!cam || !centers || !valid || !cubes || !pass_mask) return SP3D_ENULL;
    if ((hm_layout & 0xff) != SP3D_LAYOUT_NHWC || (hm_layout & (SP3D_HM_BF16 | SP3D_OUT_BF16)) || w < 2 || h < 2)
        return SP3D_EUNSUPPORTED;
    g.sample_of = sample_of;
    g.pass_mask = pass_mask;
    Views v;
    rc = load_views(v, hm_views, V);
    if (rc) return rc;
    const bool train_cl = (hm_layout & SP3D_OUT_CHANNELS_LAST) != 0;
    return launch_nhwc(v, Jp, cam, centers, valid, cubes, grids, g, default_variant(g, train_cl), train_cl, 0,
                       (hipStream_t)stream);
}

// scatter: which kernel sp3d_unproject_bwd_packed[_det] launches - SP3D_SCATTER_AUTO (by voxel pitch), _PER_TAP (bwd2),
// _MERGE (bwd3).  A per-call argument: the library keeps no selector state (include/sp3d.h "no global state").
static int bwd_packed_impl(const float *cam, const int32_t *sample_of, const float *centers, const uint8_t *valid,
                           const float *grad_cubes, const uint16_t *pass_mask, void *grad_acc, const float *scale, int B,
                           int P, int V, int J, int Jp, int h, int w, int X, int Y, int Z, const float *grid_size,
                           int W_in, int H_in, int scatter, void *stream)
{
    Geom g;
    int rc = make_geom(g, P, V, J, h, w, X, Y, Z, grid_size, W_in, H_in);
    if (rc) return rc;
    if (B <= 0) return SP3D_EINVAL;
    if (scatter != SP3D_SCATTER_AUTO && scatter != SP3D_SCATTER_PER_TAP && scatter != SP3D_SCATTER_MERGE) return SP3D_EINVAL;
    if (!cam || !centers || !valid || !grad_cubes || !pass_mask || !grad_acc) return SP3D_ENULL;
    if (Jp < J || (Jp & 3) || Jp > 16 || w < 2 || h < 2) return SP3D_EUNSUPPORTED;
    g.sample_of = sample_of;
    const int tiles = (g.N + 63) / 64;
    const size_t view_stride = (size_t)B * h * w * Jp;
    // dense grids (the 64^3 person cubes at 31.7 mm: voxels ~1.7 heat-map pixels apart): block-wise LDS merge, bwd3.
    // The pixel pitch depends on the cameras (device data); what the host knows is the voxel pitch in mm: <= 50 mm.
    const bool dense = X >= 2 && Y >= 2 && Z >= 2 && (double)grid_size[0] / (X - 1) <= 50.0 &&
                       (double)grid_size[1] / (Y - 1) <= 50.0 && (double)grid_size[2] / (Z - 1) <= 50.0;
    if (scatter == SP3D_SCATTER_MERGE || (scatter == SP3D_SCATTER_AUTO && dense)) {
        const int nbx = (X + B3_BX - 1) / B3_BX, nby = (Y + B3_BY - 1) / B3_BY, nbz = (Z + B3_BZ - 1) / B3_BZ;
        const size_t lds3 = (size_t)Jp * B3_PXS * sizeof(unsigned long long) + (4 * SP3D_MAX_VIEWS + 4) * sizeof(int);
        // the 16-byte z runs a block reads of the gradient volume share their 256-byte rows with the blocks above and
        // below: keep a whole z column of blocks on one XCD, back to back in dispatch order (chunk = nbz when a power of two)
        if ((nbz & (nbz - 1)) == 0) g.xcd_chunk = nbz;
        dim3 grid3(xcd_grid_blocks(P, nbx * nby * nbz, g.xcd_chunk)), block3(256);
        hipStream_t s3 = (hipStream_t)stream;
#define SP3D_B3(JP_, DET_) hipLaunchKernelGGL((unproject_bwd3_kernel<JP_, DET_>), grid3, block3, lds3, s3, cam, centers, valid, grad_cubes, pass_mask, grad_acc, view_stride, g, nbx, nby, nbz, scale)
        if (scale) {
            switch (Jp) { case 4: SP3D_B3(4, true); break; case 8: SP3D_B3(8, true); break; case 12: SP3D_B3(12, true); break; default: SP3D_B3(16, true); break; }
        } else {
            switch (Jp) { case 4: SP3D_B3(4, false); break; case 8: SP3D_B3(8, false); break; case 12: SP3D_B3(12, false); break; default: SP3D_B3(16, false); break; }
        }
#undef SP3D_B3
        return launch_status();
    }
    const size_t lds = (size_t)(V * 320 + Jp * 64 + 64) * sizeof(float);
    dim3 grid(xcd_grid_blocks(P, tiles, g.xcd_chunk)), block(64);
    hipStream_t s = (hipStream_t)stream;
#define SP3D_B2(JP_, DET_) hipLaunchKernelGGL((unproject_bwd2_kernel<JP_, true, DET_>), grid, block, lds, s, cam, centers, valid, grad_cubes, pass_mask, grad_acc, view_stride, g, tiles, scale)
    if (scale) {
        switch (Jp) { case 4: SP3D_B2(4, true); break; case 8: SP3D_B2(8, true); break; case 12: SP3D_B2(12, true); break; default: SP3D_B2(16, true); break; }
    } else {
        switch (Jp) { case 4: SP3D_B2(4, false); break; case 8: SP3D_B2(8, false); break; case 12: SP3D_B2(12, false); break; default: SP3D_B2(16, false); break; }
    }
#undef SP3D_B2
    return launch_status();
}

extern "C" int sp3d_unproject_bwd_packed(const float *cam, const int32_t *sample_of, const float *centers,
                                         const uint8_t *valid, const float *grad_cubes, const uint16_t *pass_mask,
                                         float *grad_packed, int B, int P, int V, int J, int Jp, int h, int w, int X,
                                         int Y, int Z, const float *grid_size, int W_in, int H_in, int scatter,
                                         void *stream)
{
    return bwd_packed_impl(cam, sample_of, centers, valid, grad_cubes, pass_mask, grad_packed, nullptr, B, P, V, J, Jp, h, w,
                           X, Y, Z, grid_size, W_in, H_in, scatter, stream);
}

extern "C" int sp3d_unproject_bwd_packed_det(const float *cam, const int32_t *sample_of, const float *centers,
                                             const uint8_t *valid, const float *grad_cubes, const uint16_t *pass_mask,
                                             int64_t *grad_fixed, const float *scale, int B, int P, int V, int J, int Jp,
                                             int h, int w, int X, int Y, int Z, const float *grid_size, int W_in, int H_in,
                                             int scatter, void *stream)
{
    if (!scale) return SP3D_ENULL;
    return bwd_packed_impl(cam, sample_of, centers, valid, grad_cubes, pass_mask, grad_fixed, scale, B, P, V, J, Jp, h, w, X,
                           Y, Z, grid_size, W_in, H_in, scatter, stream);
}

extern "C" int sp3d_fixed_to_float(const int64_t *acc, float *out, const float *scale, int64_t n, void *stream)
{
    if (n <= 0) return SP3D_EINVAL;
    if (!acc || !out || !scale) return SP3D_ENULL;
    if ((n + 255) / 256 > 0x7fffffff) return SP3D_ERANGE;
    hipLaunchKernelGGL(fixed_to_float_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const long long *>(acc), out, scale, (size_t)n);
    return launch_status();
}

// measurement only: one thread writes the chip-wide 100 MHz clock (s_memrealtime) to *slot.  Two of them around a kernel
// inside a captured HIP graph give the kernel's time in the replayed step (bench.py roofline.in_step_graph_stamps; PyTorch's
// ROCm build refuses timing events inside a capture).
namespace sp3d {
__global__ void stamp_kernel(unsigned long long *slot) { *slot = wall_clock64(); }
} // namespace sp3d
extern "C" int sp3d_debug_stamp(uint64_t *slot, void *stream)
{
    if (!slot) return SP3D_ENULL;
    hipLaunchKernelGGL(sp3d::stamp_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, reinterpret_cast<unsigned long long *>(slot));
    return launch_status();
}

// measurement only: which -DSP3D_ABLATE mask this library carries (tools/diag_ablate.py)
extern "C" int sp3d_debug_set_diag(int flags)
{
    (void)flags;
    return SP3D_ABLATE;      // the ablation mask this library was compiled with (0 = the shipped kernels)
}

// measurement only: set / clear the per-wave timeline buffer of the pipelined kernel
extern "C" int sp3d_debug_set_timeline(void *dev_buffer)
{
#ifdef SP3D_TIMELINE
    unsigned long long *p = (unsigned long long *)dev_buffer;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_timeline), &p, sizeof(p));
#else
    (void)dev_buffer;
    return SP3D_EUNSUPPORTED;       // the shipped library carries no stamps; tools/wave_timeline.py builds its own
#endif
}
