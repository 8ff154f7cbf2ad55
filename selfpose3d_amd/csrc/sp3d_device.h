// sp3d_device.h - device-side math shared by the gfx950 unprojection kernels.
//
// Arithmetic contract: every operation below is the fp32 operation the
// reference performs, in the reference's order; the only fused multiply-adds are the ones
// torch's K=3 `mm` performs (rigid transform, crop affine) and ATen's bilinear chain, and
// they are written as explicit fmaf().  The translation unit is compiled with
// -ffp-contract=off so the compiler adds none of its own; division is IEEE (hipcc default
// -fhip-fp32-correctly-rounded-divide-sqrt).  Result: bit-identical to oracle/sp3d_oracle.c.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sp3d.h"

namespace sp3d {

struct Views {
    const float *p[SP3D_MAX_VIEWS];
};
struct ViewsMut {
    float *p[SP3D_MAX_VIEWS];
};

struct Geom {
    int B, V, J, h, w, X, Y, Z;
    int N;          // X*Y*Z
    int YZ;         // Y*Z
    int W_in, H_in; // network input size (cfg.NETWORK.IMAGE_SIZE)
    float Lx, Ly, Lz;
    float rW_in, rH_in, rw1, rh1; // correctly rounded 1/W_in, 1/H_in, 1/(w-1), 1/(h-1)
    float stepx, stepy, stepz;    // fp32 (end-start)/(n-1) of torch.linspace, computed on the host (same IEEE ops)
    uint32_t magicYZ, magicZ;     // floor(2^32/d)+1 for d = Y*Z, Z (exact n/d with one fix-up, see udiv_magic)
    const int *sample_of;         // optional (P): heat-map / camera row each output cube reads (NULL: identity)
    int xcd_chunk;                // tiles per chunk of the XCD-affine tile map (power of two)
    int xcd_order;                // 0: chunks in sweep order, 1: centre of the volume first
    // planar result addressing (pipelined / brick kernels): element (p, j, x, y, z) lives at p*sB + j*sJ + x*sX + y*sY + z.
    // Dense (B,J,X,Y,Z): sB = J*N, sJ = N, sX = Y*Z, sY = Z.  Other strides let the kernel write straight into a larger
    // buffer (the zero-padded input of the frequency-domain opening conv, sp3d_unproject_fwd_strided).
    long long sB;
    int sJ, sX, sY;
    int dense;                    // 1: the strides are the dense ones (flat voxel index == offset)
    int vec4;                     // 1: rows are 4-element aligned, 16-byte result pieces are allowed
    uint16_t *pass_mask;          // optional (P,N): bit j set iff 0 <= pre-clamp value of channel j <= 1 and the
                                  // voxel is not NaN-zeroed (where torch's clamp / index_put_ let gradient through)
    // blockIdx -> (sample, tile) without run-time integer divisions (set_xcd_fields; round 3: the ~10 division sequences
    // of the old prologue were ~250 of a brick wave's ~1 350 instructions, and these kernels are instruction-issue bound)
    int xm_mode;                  // 0: B <= 8 and 8 % B == 0 (XCD groups per sample), 1: B % 8 == 0, 2: plain interleave
    int xm_log2xps, xm_log2K;     // mode 0: XCDs per sample and tiles per chunk, both powers of two
    int xm_rows;                  // mode 0: chunk rows per serving XCD (centre-out order)
    int xm_tiles;                 // tiles (workgroups) per sample the fields were computed for
    uint32_t xm_magic_tiles;      // floor(2^32 / tiles) + 1
    // brick decode: wg -> (z chunk, y brick, x brick)
    int bk_nxy, bk_nby;
    uint32_t bk_magic_nxy, bk_magic_nby;
    // xm_mode 3 (round 6, brick kernels, B in {1, 2, 4}): the 8 / B XCDs of a sample each own ONE rectangular (x, y) block
    // of brick columns over the whole z extent instead of round-robin chunks (set_block_fields): an XCD's L2 then sees the
    // part of every view its block projects to, once (tools/sim_xcd_map.py: L2 fills 6.9x -> 2.6x the heat-maps at B = 1)
    int blk_log2py, blk_w, blk_h, blk_nbx, blk_nzc;
    uint32_t blk_magic_wh, blk_magic_h;
};

// torch.linspace(-L/2, L/2, n)[i] in fp32 (project_layer.py:28-30; ATen CPU kernel form)
__device__ __forceinline__ float linspace_at(float L, int n, int i)
{
    const float start = -(L / 2.0f), end = L / 2.0f;
    if (n == 1) return start;
    const float step = (end - start) / (float)(n - 1);
    return (i < n / 2) ? fmaf(step, (float)i, start) : fmaf(-step, (float)(n - 1 - i), end);
}

// linspace with the step precomputed on the host (bit-identical to linspace_at)
__device__ __forceinline__ float linspace_step(float L, float step, int n, int i)
{
    const float start = -(L / 2.0f), end = L / 2.0f;
    if (n == 1) return start;
    return (i < n / 2) ? fmaf(step, (float)i, start) : fmaf(-step, (float)(n - 1 - i), end);
}

// exact n / d for a uniform divisor d with magic = floor(2^32/d)+1: umulhi over-estimates by at most 1
__device__ __forceinline__ void udiv_magic(uint32_t n, uint32_t d, uint32_t magic, int &q, int &r)
{
    uint32_t qq = d == 1 ? n : __umulhi(n, magic);
    int rr = (int)(n - qq * d);
    if (rr < 0) { qq -= 1; rr += (int)d; }
    q = (int)qq; r = rr;
}

// clamp for values already known not to be NaN-relevant: one v_med3_f32 (NaN -> lo)
__device__ __forceinline__ float clamp_fast(float v, float lo, float hi) { return __builtin_amdgcn_fmed3f(v, lo, hi); }

__device__ __forceinline__ float clamp_nan(float v, float lo, float hi)
{
    // torch.clamp: NaN propagates
    return (v != v) ? v : (v < lo ? lo : (v > hi ? hi : v));
}

// Voxel centre -> sample position in heat-map pixels for one view.
// project_layer.py:76-90 (project_pose = cameras.py:27-55) + ATen unnormalize(align_corners).
// `cm` is wave-uniform (one sample, one view): the compiler keeps it in SGPRs via s_load.
__device__ __forceinline__ bool sample_pos(const float *__restrict__ cm, float x, float y, float z, int w, int h,
                                           float W_in, float H_in, float &ix, float &iy)
{
    const float dx = x - cm[SP3D_CAM_T + 0], dy = y - cm[SP3D_CAM_T + 1], dz = z - cm[SP3D_CAM_T + 2];
    const float xc = fmaf(cm[2], dz, fmaf(cm[1], dy, cm[0] * dx));
    const float yc = fmaf(cm[5], dz, fmaf(cm[4], dy, cm[3] * dx));
    const float zc = fmaf(cm[8], dz, fmaf(cm[7], dy, cm[6] * dx));
    const float den = zc + 1e-5f;
    const float y0 = xc / den, y1 = yc / den;
    float r2 = y0 * y0 + y1 * y1;
    r2 = r2 > 1e10f ? 1e10f : r2;
    const float r4 = r2 * r2, r6 = r4 * r2;
    const float radial = 1.0f + ((cm[SP3D_CAM_K] * r2 + cm[SP3D_CAM_K + 1] * r4) + cm[SP3D_CAM_K + 2] * r6);
    const float tan = cm[SP3D_CAM_P] * y1 + cm[SP3D_CAM_P + 1] * y0;
    const float corr = radial + 2.0f * tan;
    const float u0 = y0 * corr + cm[SP3D_CAM_P + 1] * r2;
    const float u1 = y1 * corr + cm[SP3D_CAM_P] * r2;
    float px = cm[SP3D_CAM_F] * u0 + cm[SP3D_CAM_C];
    float py = cm[SP3D_CAM_F + 1] * u1 + cm[SP3D_CAM_C + 1];
    const float W0 = cm[SP3D_CAM_W0], H0 = cm[SP3D_CAM_H0];
    const bool bound = (px >= 0.0f) && (py >= 0.0f) && (px < W0) && (py < H0);
    const float mx = W0 > H0 ? W0 : H0;
    px = clamp_nan(px, -1.0f, mx);
    py = clamp_nan(py, -1.0f, mx);
    float qx = fmaf(cm[SP3D_CAM_A + 2], 1.0f, fmaf(cm[SP3D_CAM_A + 1], py, cm[SP3D_CAM_A + 0] * px));
    const float qy = fmaf(cm[SP3D_CAM_A + 5], 1.0f, fmaf(cm[SP3D_CAM_A + 4], py, cm[SP3D_CAM_A + 3] * px));
    if (cm[SP3D_CAM_FLIP] != 0.0f) qx = W_in - qx;
    const float ux = qx * (float)w / W_in;
    const float uy = qy * (float)h / H_in;
    float gx = ux / (float)(w - 1) * 2.0f - 1.0f;
    float gy = uy / (float)(h - 1) * 2.0f - 1.0f;
    gx = clamp_nan(gx, -1.1f, 1.1f);
    gy = clamp_nan(gy, -1.1f, 1.1f);
    ix = (gx + 1.0f) * ((float)(w - 1) / 2.0f);
    iy = (gy + 1.0f) * ((float)(h - 1) / 2.0f);
    return bound;
}

// x / c for a wave-uniform constant c with rc = RN(1/c): one Newton correction on the residual
// gives the correctly rounded quotient (verified exhaustively-by-sampling against IEEE division
// for every constant used, tests/test_host_cabi.py::test_const_division_is_exact); 3 VALU ops
// instead of the ~10 of the IEEE expansion.  x must be finite (it is: px,py are clamped).
__device__ __forceinline__ float div_const(float x, float c, float rc)
{
    const float q = x * rc;
    const float r = fmaf(-q, c, x);
    return fmaf(r, rc, q);
}

// (x0/d, x1/d) with one shared reciprocal: the IEEE f32 division expansion (rcp, one Newton step on
// the reciprocal, quotient, two residual corrections) minus its range scaling, which is the
// identity for |d| in [2^-96, 2^96] and finite non-overflowing quotients.  d = depth + 1e-5 (mm):
// only d == 0 or non-finite operands leave that range, and those take the plain IEEE path.
__device__ __forceinline__ void div_pair(float x0, float x1, float d, float &q0, float &q1)
{
    const float ad = __builtin_fabsf(d);
    if (__builtin_expect(!(ad >= 1e-20f && ad <= 1e20f), 0)) { q0 = x0 / d; q1 = x1 / d; return; }
    float r = __builtin_amdgcn_rcpf(d);
    const float e = fmaf(-d, r, 1.0f);
    r = fmaf(e, r, r);
    float a = x0 * r, b = x1 * r;
    a = fmaf(fmaf(-d, a, x0), r, a);
    b = fmaf(fmaf(-d, b, x1), r, b);
    q0 = fmaf(fmaf(-d, a, x0), r, a);
    q1 = fmaf(fmaf(-d, b, x1), r, b);
}

// sample_pos with (a) the four divisions by image-size constants done by div_const and (b) NaN
// tracked as a flag: a NaN can only be born before the first clamp (everything after it is a
// bounded affine map of clamped values), so `isnan` = NaN(px)|NaN(py) and the clamps become
// single v_med3_f32.  Same bits as sample_pos for ix, iy whenever isnan is false; when it is
// true the caller zeroes the voxel exactly as a NaN sample position does in the reference.
// The three stages of sample_pos_fast, separately callable so that a kernel can interleave them with
// memory instructions (unproject_pipe_kernel ILV): a = camera transform + perspective division,
// b = distortion, pixel, bound test, first clamp, c = crop affine .. heat-map sample position.
__device__ __forceinline__ void proj_a(const float *__restrict__ cm, float x, float y, float z, float &y0, float &y1)
{
    const float dx = x - cm[SP3D_CAM_T + 0], dy = y - cm[SP3D_CAM_T + 1], dz = z - cm[SP3D_CAM_T + 2];
    const float xc = fmaf(cm[2], dz, fmaf(cm[1], dy, cm[0] * dx));
    const float yc = fmaf(cm[5], dz, fmaf(cm[4], dy, cm[3] * dx));
    const float zc = fmaf(cm[8], dz, fmaf(cm[7], dy, cm[6] * dx));
    const float den = zc + 1e-5f;
    div_pair(xc, yc, den, y0, y1);
}

__device__ __forceinline__ bool proj_b(const float *__restrict__ cm, float y0, float y1, float &px, float &py, bool &isnan)
{
    float r2 = y0 * y0 + y1 * y1;
    r2 = fminf(r2, 1e10f);        // NaN r2 -> 1e10, harmless: y0|y1 NaN already makes px,py NaN
    const float r4 = r2 * r2, r6 = r4 * r2;
    const float radial = 1.0f + ((cm[SP3D_CAM_K] * r2 + cm[SP3D_CAM_K + 1] * r4) + cm[SP3D_CAM_K + 2] * r6);
    const float tan = cm[SP3D_CAM_P] * y1 + cm[SP3D_CAM_P + 1] * y0;
    const float corr = radial + 2.0f * tan;
    const float u0 = y0 * corr + cm[SP3D_CAM_P + 1] * r2;
    const float u1 = y1 * corr + cm[SP3D_CAM_P] * r2;
    px = cm[SP3D_CAM_F] * u0 + cm[SP3D_CAM_C];
    py = cm[SP3D_CAM_F + 1] * u1 + cm[SP3D_CAM_C + 1];
    const float W0 = cm[SP3D_CAM_W0], H0 = cm[SP3D_CAM_H0];
    // four compares straight into lane masks, combined on the scalar unit (the plain && chain is turned into a
    // <4 x float> compare + 16-bit shuffling by the SLP vectoriser: 15 VALU ops instead of 4)
    const bool bound = __builtin_amdgcn_inverse_ballot_w64(
        __builtin_amdgcn_ballot_w64(px >= 0.0f) & __builtin_amdgcn_ballot_w64(py >= 0.0f) &
        __builtin_amdgcn_ballot_w64(px < W0) & __builtin_amdgcn_ballot_w64(py < H0));
    isnan = (px != px) || (py != py);
    const float mx = W0 > H0 ? W0 : H0;
    px = clamp_fast(px, -1.0f, mx);
    py = clamp_fast(py, -1.0f, mx);
    return bound;
}

// all six entries of the crop affine finite?  (wave-uniform: integer tests on the scalar unit)
__device__ __forceinline__ bool affine_finite(const float *__restrict__ cm)
{
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 6; ++i) ok = ok && ((__float_as_uint(cm[SP3D_CAM_A + i]) & 0x7f800000u) != 0x7f800000u);
    return ok;
}

__device__ __forceinline__ void proj_c(const float *__restrict__ cm, const Geom &g, float px, float py, float &ix,
                                       float &iy, bool &isnan)
{
    float qx = fmaf(cm[SP3D_CAM_A + 2], 1.0f, fmaf(cm[SP3D_CAM_A + 1], py, cm[SP3D_CAM_A + 0] * px));
    const float qy = fmaf(cm[SP3D_CAM_A + 5], 1.0f, fmaf(cm[SP3D_CAM_A + 4], py, cm[SP3D_CAM_A + 3] * px));
    const float W_in = (float)g.W_in, H_in = (float)g.H_in;
    if (cm[SP3D_CAM_FLIP] != 0.0f) qx = W_in - qx;
    float gx, gy;
    if (__builtin_expect(affine_finite(cm), 1)) {
        const float ux = div_const(qx * (float)g.w, W_in, g.rW_in);
        const float uy = div_const(qy * (float)g.h, H_in, g.rH_in);
        gx = div_const(ux, (float)(g.w - 1), g.rw1) * 2.0f - 1.0f;
        gy = div_const(uy, (float)(g.h - 1), g.rh1) * 2.0f - 1.0f;
    } else {                                        // infinite q: div_const would turn it into NaN, IEEE keeps it
        gx = qx * (float)g.w / W_in / (float)(g.w - 1) * 2.0f - 1.0f;
        gy = qy * (float)g.h / H_in / (float)(g.h - 1) * 2.0f - 1.0f;
    }
    isnan = isnan || (gx != gx) || (gy != gy);      // non-finite camera tables only
    gx = clamp_fast(gx, -1.1f, 1.1f);
    gy = clamp_fast(gy, -1.1f, 1.1f);
    ix = (gx + 1.0f) * ((float)(g.w - 1) / 2.0f);
    iy = (gy + 1.0f) * ((float)(g.h - 1) / 2.0f);
}

__device__ __forceinline__ bool sample_pos_fast(const float *__restrict__ cm, float x, float y, float z, const Geom &g,
                                                float &ix, float &iy, bool &isnan)
{
    float y0, y1, px, py;
    proj_a(cm, x, y, z, y0, y1);
    const bool bound = proj_b(cm, y0, y1, px, py, isnan);
    proj_c(cm, g, px, py, ix, iy, isnan);
    return bound;
}

// bilinear weights, ATen CPU form: w = ix - floor(ix), e = 1 - w, ...
struct Bilin {
    int x0, y0;
    float wnw, wne, wsw, wse;
};
__device__ __forceinline__ Bilin bilin(float ix, float iy)
{
    Bilin b;
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const float wx = ix - fx0, ex = 1.0f - wx, ny = iy - fy0, sy = 1.0f - ny;
    b.wnw = sy * ex; b.wne = sy * wx; b.wsw = ny * ex; b.wse = ny * wx;
    b.x0 = (int)fx0; b.y0 = (int)fy0;
    return b;
}

// output of the view fusion for one channel (project_layer.py:96-99)
__device__ __forceinline__ float fuse(float acc, float den)
{
    float o = acc / den;
    if (o != o) o = 0.0f;
    return o < 0.0f ? 0.0f : (o > 1.0f ? 1.0f : o);
}

// acc / den with rden = RN(1/den): den = (#views seeing the voxel) + 1e-6f takes at most 17 values
// (SP3D_MAX_VIEWS); the corrected quotient is the IEEE one for each of them (tested like div_const).
__device__ __forceinline__ float fuse_rcp(float acc, float den, float rden)
{
    const float q = acc * rden;
    const float r = fmaf(-q, den, acc);
    const float o = fmaf(r, rden, q);
    return __builtin_amdgcn_fmed3f(o, 0.0f, 1.0f);   // med3(NaN,0,1) = 0: the NaN->0 rule of project_layer.py:98
}

// pre-clamp quotient of fuse_rcp (for the gradient pass mask)
__device__ __forceinline__ float fuse_pre(float acc, float den, float rden)
{
    const float q = acc * rden;
    return fmaf(fmaf(-q, den, acc), rden, q);
}

// ---- storage types: fp32 or bf16 (math is always fp32) -----------------------------------------
struct bf16_t { uint16_t v; };

__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f)       // round to nearest even (values here are finite)
{
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

template <typename T> struct Store4;
template <> struct Store4<float> {
    __device__ __forceinline__ static float4 load(const float *p) { return *reinterpret_cast<const float4 *>(p); }
    __device__ __forceinline__ static void store(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
    // the cubes are written once and read by a later kernel: non-temporal stores keep them from evicting the
    // heat-map lines the gather re-uses out of the XCD's L2 (-2.5 % on the bench workload, -8 % when the maps fit).
    // Not for the packed heat-maps: those are re-read by the very next kernel and should stay cached.
    __device__ __forceinline__ static void store_nt(float *p, float4 v)
    {
        typedef float v4f __attribute__((ext_vector_type(4)));
        v4f t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
        __builtin_nontemporal_store(t, reinterpret_cast<v4f *>(p));
    }
    __device__ __forceinline__ static void store1(float *p, float v) { *p = v; }
};
template <> struct Store4<bf16_t> {
    __device__ __forceinline__ static float4 load(const bf16_t *p)
    {
        const uint2 r = *reinterpret_cast<const uint2 *>(p);
        return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                           __uint_as_float(r.y & 0xffff0000u));
    }
    __device__ __forceinline__ static uint2 pack4(float4 v)
    {
        uint2 r;
        r.x = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
        r.y = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
        return r;
    }
    __device__ __forceinline__ static void store(bf16_t *p, float4 v) { *reinterpret_cast<uint2 *>(p) = pack4(v); }
    __device__ __forceinline__ static void store_nt(bf16_t *p, float4 v)
    {
        typedef uint32_t v2u __attribute__((ext_vector_type(2)));
        const uint2 r = pack4(v);
        v2u t; t.x = r.x; t.y = r.y;
        __builtin_nontemporal_store(t, reinterpret_cast<v2u *>(p));
    }
    __device__ __forceinline__ static void store1(bf16_t *p, float v) { p->v = f32_to_bf16(v); }
};

// blockIdx -> (sample, tile) with XCD affinity AND balance.  Observed placement: block `bid` runs on
// XCD bid % 8 (speed only - any placement gives the same results).  Each XCD has its own 4 MiB L2, so
// all tiles of one sample should run on as few XCDs as possible (its heat-maps are then fetched into
// those L2s only); but voxel tiles differ a lot in cost (camera visibility), so the tiles of a sample
// are dealt round-robin to the XCDs that serve it instead of in contiguous ranges:
//   B <= 8, 8 % B == 0 : sample b is served by the 8/B XCDs {b*8/B ...}, tile t by XCD (t % (8/B))
//   B  > 8, B % 8 == 0 : XCD x serves samples x, x+8, ... whole
//   otherwise          : plain interleave (tile-major order), still correct, less affinity
// returns false if this block has no work.  grid size: xcd_grid_blocks().
// K = tiles per dealt chunk (power of two): a sample's tiles go to its XCDs in chunks of K consecutive
// tiles - larger K keeps each XCD on a compact part of the volume (fewer distinct heat-map lines per
// L2), smaller K balances the load better.  K comes from Geom::xcd_chunk.
__host__ __device__ __forceinline__ int xcd_slots_per_xcd(int B, int tiles, int K)
{
    if (B <= 8 && (8 % B) == 0) {
        const int xps = 8 / B;
        const int chunks = (tiles + K - 1) / K;
        return ((chunks + xps - 1) / xps) * K;
    }
    if (B > 8 && (B % 8) == 0) return (B / 8) * tiles;
    return (B * tiles + 7) / 8;
}
__host__ __device__ __forceinline__ int xcd_grid_blocks(int B, int tiles, int K) { return 8 * xcd_slots_per_xcd(B, tiles, K); }

__device__ __forceinline__ bool xcd_map(int bid, int B, int tiles, int K, int &b, int &tile, int order = 0)
{
    const int x = bid & 7, slot = bid >> 3;
    if (B <= 8 && (8 % B) == 0) {
        const int xps = 8 / B;
        b = x / xps;
        const int sub = x - b * xps;
        int row = slot / K;                                   // chunk rows in dispatch order
        if (order & 1) {                                      // centre-out: the cheap edge-of-volume tiles run last
            const int rows = ((tiles + K - 1) / K + xps - 1) / xps, mid = rows / 2;
            row = mid + ((row & 1) ? -((row + 1) >> 1) : (row >> 1));
        }
        const int chunk = row * xps + sub;
        tile = chunk * K + (slot & (K - 1));
        return tile < tiles;
    }
    if (B > 8 && (B % 8) == 0) {
        const int j = slot / tiles;
        b = x + 8 * j;
        tile = slot - j * tiles;
        return b < B;
    }
    const int lt = slot * 8 + x;          // plain order
    b = lt / tiles;
    tile = lt - b * tiles;
    return b < B;
}

// host: fill the division-free map fields for `tiles` workgroups per sample and chunk size K = g.xcd_chunk
inline void set_xcd_fields(Geom &g, int tiles)
{
    auto ilog2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return l; };
    g.xm_tiles = tiles;
    g.xm_magic_tiles = (uint32_t)((0x100000000ull / (uint64_t)(tiles > 0 ? tiles : 1)) + 1ull);
    // the shift-and-mask decode below equals xcd_map() only for a power-of-two chunk size: any other K (no call site
    // produces one) takes the plain interleave, whose decode is exact for every grid >= B * tiles blocks
    if (g.B <= 8 && (8 % g.B) == 0 && g.xcd_chunk > 0 && (g.xcd_chunk & (g.xcd_chunk - 1)) == 0) {
        const int xps = 8 / g.B, K = g.xcd_chunk;
        g.xm_mode = 0; g.xm_log2xps = ilog2(xps); g.xm_log2K = ilog2(K);
        g.xm_rows = ((tiles + K - 1) / K + xps - 1) / xps;
    } else {
        g.xm_mode = (g.B > 8 && (g.B % 8) == 0) ? 1 : 2;
        g.xm_log2xps = g.xm_log2K = 0; g.xm_rows = 0;
    }
}
// host: block map (xm_mode 3) for nbx x nby x nzc brick stacks per sample; returns the grid size in workgroups, or 0 when
// the batch size has no whole number of XCDs per sample (the caller keeps the chunk map)
inline int set_block_fields(Geom &g, int nbx, int nby, int nzc)
{
    if (!(g.B == 1 || g.B == 2 || g.B == 4)) return 0;
    if (g.B == 1 && nbx == nby && (nbx & 1) == 0 && nbx >= 4) {          // octants (xm_mode 4)
        const int h = nbx / 2, per = h * (h - 1) / 2 + (h + 1) / 2;
        g.xm_mode = 4;
        g.blk_w = h; g.blk_h = per; g.blk_nbx = nbx; g.blk_nzc = nzc; g.blk_log2py = 0;
        g.blk_magic_h = (uint32_t)((0x100000000ull / (uint64_t)per) + 1ull);
        g.blk_magic_wh = 0;
        return 8 * nzc * per;
    }
    const int xps = 8 / g.B;
    const int px = xps == 8 ? 2 : (xps == 4 ? 2 : 1), py = xps / px;       // 2 x 4, 2 x 2, 1 x 2 blocks
    auto ilog2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return l; };
    g.xm_mode = 3;
    g.xm_log2xps = ilog2(xps);
    g.blk_log2py = ilog2(py);
    g.blk_w = (nbx + px - 1) / px; g.blk_h = (nby + py - 1) / py; g.blk_nbx = nbx; g.blk_nzc = nzc;
    g.blk_magic_wh = (uint32_t)((0x100000000ull / (uint64_t)(g.blk_w * g.blk_h)) + 1ull);
    g.blk_magic_h = (uint32_t)((0x100000000ull / (uint64_t)g.blk_h) + 1ull);
    return 8 * nzc * g.blk_w * g.blk_h;
}
inline void set_brick_fields(Geom &g, int nxy, int nby)
{
    g.bk_nxy = nxy; g.bk_nby = nby;
    g.bk_magic_nxy = (uint32_t)((0x100000000ull / (uint64_t)(nxy > 0 ? nxy : 1)) + 1ull);
    g.bk_magic_nby = (uint32_t)((0x100000000ull / (uint64_t)(nby > 0 ? nby : 1)) + 1ull);
}

// xcd_map() with the host-prepared fields: shifts and at most one magic division (same mapping, bit for bit)
__device__ __forceinline__ bool xcd_map_fast(int bid, const Geom &g, int &b, int &tile)
{
    const int x = bid & 7, slot = bid >> 3;
    if (g.xm_mode == 0) {
        b = x >> g.xm_log2xps;
        const int sub = x & ((1 << g.xm_log2xps) - 1);
        int row = slot >> g.xm_log2K;
        if (g.xcd_order & 1) {                                // centre-out: the cheap edge-of-volume tiles run last
            const int mid = g.xm_rows / 2;
            row = mid + ((row & 1) ? -((row + 1) >> 1) : (row >> 1));
        }
        const int chunk = (row << g.xm_log2xps) + sub;
        tile = (chunk << g.xm_log2K) + (slot & ((1 << g.xm_log2K) - 1));
        return tile < g.xm_tiles;
    }
    int q, r;
    if (g.xm_mode == 3) {
        b = x >> g.xm_log2xps;
        const int sub = x & ((1 << g.xm_log2xps) - 1);
        const int pxi = sub >> g.blk_log2py, pyi = sub & ((1 << g.blk_log2py) - 1);
        int zc, t, lx, ly;
        udiv_magic((uint32_t)slot, (uint32_t)(g.blk_w * g.blk_h), g.blk_magic_wh, zc, t);      // z slowest, then x, then y
        udiv_magic((uint32_t)t, (uint32_t)g.blk_h, g.blk_magic_h, lx, ly);
        const int bx = pxi * g.blk_w + lx, by = pyi * g.blk_h + ly;
        tile = (zc * g.blk_nbx + bx) * g.bk_nby + by;
        return zc < g.blk_nzc && bx < g.blk_nbx && by < g.bk_nby;
    }
    if (g.xm_mode == 4) {
        // B = 1: the 8 XCDs own the 8 OCTANTS of the (x, y) plane of brick columns - quadrants around the grid's centre cut
        // along their diagonal - over the whole z extent.  Equal column counts, and for camera rigs that surround the scene
        // about equal work (a rectangular 2 x 4 cut gave the inner blocks 21 % more work than the mean: +9 % time).
        // (u, v) = a column's distances from the two centre lines; upper octant: v < u, or v == u with u even.
        b = 0;
        const int quad = x >> 1, upper = x & 1, h = g.blk_w, tri = (h * (h - 1)) >> 1;
        int zc, k;
        udiv_magic((uint32_t)slot, (uint32_t)g.blk_h, g.blk_magic_h, zc, k);           // blk_h = columns per octant (max)
        int u, v;
        if (k < tri) {
            u = (int)((1.0f + __builtin_sqrtf(1.0f + 8.0f * (float)k)) * 0.5f);
            while (((u * (u - 1)) >> 1) > k) --u;
            while ((((u + 1) * u) >> 1) <= k) ++u;
            v = k - ((u * (u - 1)) >> 1);                                            // 0 <= v < u
        } else {
            u = v = 2 * (k - tri) + (upper ? 0 : 1);                                    // its share of the diagonal
        }
        if (!upper) { const int tmp = u; u = v; v = tmp; }
        const int cx = g.blk_nbx >> 1;
        const int bx = (quad & 2) ? cx + u : cx - 1 - u, by = (quad & 1) ? cx + v : cx - 1 - v;
        tile = (zc * g.blk_nbx + bx) * g.bk_nby + by;
        return zc < g.blk_nzc && u < h && v < h;
    }
    if (g.xm_mode == 1) {
        udiv_magic((uint32_t)slot, (uint32_t)g.xm_tiles, g.xm_magic_tiles, q, r);
        b = x + 8 * q; tile = r;
        return b < g.B;
    }
    udiv_magic((uint32_t)(slot * 8 + x), (uint32_t)g.xm_tiles, g.xm_magic_tiles, q, r);
    b = q; tile = r;
    return b < g.B;
}

} // namespace sp3d
