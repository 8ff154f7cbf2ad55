// sp3d_proj_pk.h - the per-voxel projection of sp3d_device.h (proj_a / proj_b / proj_c / make_record) written on
// explicit 2-wide vectors, so that every (x, y) pair of the reference's arithmetic is ONE packed-fp32 instruction
// (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32: two IEEE fp32 operations per lane in the issue slot of one).
//
// Why (round 3): a plain VALU instruction occupies its SIMD for 4 cycles per wave64, packed or not
// (profiles/r02_pmc_unproject_fine64.json: SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 1.01 quad-cycles), and on the dense
// grids the pipelined kernels spend 56-61 % of the launch issuing VALU - they are instruction-issue bound, not
// bandwidth bound.  The scalar form of the projection was ~175 instructions per (wave, view); this form is ~95.
//
// Arithmetic contract unchanged: each lane-half performs exactly the fp32 operation sequence of
// sample_pos() in sp3d_device.h = the reference's (project_layer.py:76-90, cameras.py:27-55); packing changes which
// instruction carries an operation, never the operation.  The three places where an fma replaces a (mul, add) pair
// are exact by construction: the multiplier is 2 or 1 (the product is exact, so the single rounding of the fma is the
// rounding of the add).  Bit-equality with the oracle is asserted by every GPU parity test.
#pragma once
#include "sp3d_device.h"

namespace sp3d {

#ifndef SP3D_NO_PK
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
#else
// -DSP3D_NO_PK (libsp3d_nopk.so, built with -fno-slp-vectorize): the same operations as one plain VALU instruction per
// component - no v_pk_*_f32 anywhere in the library: the conservative flavour for a GPU that is SHARED (two streams, two
// processes).  Same bits, ~6-9 % slower.  Background: ONE packed form - low result <- high half of source 1, both
// multiplicands in vector registers - is wrong in lanes 48-63 while waves of another kernel execute gfx950's double-rate
// matrix instructions on the same CU (profiles/r05_mfma_pk_hazard.md).  The compiler, not this header, picks operand order
// and op_sel bits of the instructions it forms from the arithmetic below, so the BUILD removes that form from the default
// flavour (selfpose3d_amd/pk_src1.py exchanges source 0 and source 1 in the generated assembly).
struct v2f {
    float x, y;
};
__device__ __forceinline__ v2f operator+(v2f a, v2f b) { return v2f{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ v2f operator-(v2f a, v2f b) { return v2f{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ v2f operator*(v2f a, v2f b) { return v2f{a.x * b.x, a.y * b.y}; }
__device__ __forceinline__ v2f operator-(v2f a) { return v2f{-a.x, -a.y}; }
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return v2f{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)}; }
#endif
__device__ __forceinline__ v2f pk2(float a) { return v2f{a, a}; }

// One voxel's sample position in view `cm`: everything of project_layer.py:76-90 up to the un-normalised heat-map
// coordinate.  Returns false (early out, wave-uniform) when no voxel of the wave lies inside the camera image.
struct P1State {
    v2f i;              // (ix, iy): un-normalised sample position (always finite: it is a clamped value's affine image)
    unsigned long long bm;      // lanes inside the ORIGINAL image (project_layer.py:78-79) that own a voxel
    unsigned long long nm;      // lanes whose reference sample position is NaN -> output NaN -> 0 (project_layer.py:98)
};

// any lane set? / this lane set?  (scalar-unit forms: `__any(pred)` costs two VALU instructions here)
__device__ __forceinline__ bool lane_of(unsigned long long m) { return __builtin_amdgcn_inverse_ballot_w64(m); }

// x += (lane in m): one v_addc_co_u32 with the lane mask as carry-in
__device__ __forceinline__ void add_mask(uint32_t &x, unsigned long long m)
{
    asm volatile("v_addc_co_u32_e64 %0, vcc, 0, %0, %1" : "+v"(x) : "s"(m) : "vcc");
}

// max(a, b) of two wave-uniform floats on the scalar unit when both are non-negative numbers (integer order == float order)
__device__ __forceinline__ float umax_f32(float a, float b)
{
    const uint32_t ab = __float_as_uint(a), bb = __float_as_uint(b);
    if (ab <= 0x7f800000u && bb <= 0x7f800000u) return __uint_as_float(ab > bb ? ab : bb);
    return a > b ? a : b;
}

// all six entries of the crop affine of magnitude <= 1e30?  (wave-uniform: integer tests on the scalar unit.)  Then
// every value of the fast path below stays finite (|q| <= 2e30 * 1e5, times w,h < 2^15), so no NaN can be born in it
// and div_const's reciprocal form equals the IEEE division; larger or non-finite rows take the IEEE form.
__device__ __forceinline__ bool affine_tame(const float *__restrict__ cm)
{
    // computed once per record on the host (sp3d_camera_finish: integer max of the six magnitudes <= 0x7149f2ca = 1e30f);
    // in the kernel the test was 12 scalar instructions per view and wave
    return __float_as_uint(cm[SP3D_CAM_TAME]) != 0u;
}

// inbm: lanes that own a voxel (ballot of `inb`, computed once per wave by the caller)
template <bool EARLY_OUT = true>
__device__ __forceinline__ bool project_pk(const float *__restrict__ cm, const Geom &g, float x, float y, float z,
                                           unsigned long long inbm, P1State &o)
{
    // --- camera transform + perspective division (cameras.py:40-42); torch's K=3 mm is an fma chain
    const v2f dxy = v2f{x, y} - v2f{cm[SP3D_CAM_TXY], cm[SP3D_CAM_TXY + 1]};
    const float dz = z - cm[SP3D_CAM_TZ];
    // operand pairs from the record's derived block: aligned scalar-register pairs, no moves
    v2f c2 = v2f{cm[SP3D_CAM_RXY + 0], cm[SP3D_CAM_RXY + 1]} * pk2(dxy.x);
    c2 = pk_fma(v2f{cm[SP3D_CAM_RXY + 2], cm[SP3D_CAM_RXY + 3]}, pk2(dxy.y), c2);
    c2 = pk_fma(v2f{cm[SP3D_CAM_RXY + 4], cm[SP3D_CAM_RXY + 5]}, pk2(dz), c2);
    const float zc = fmaf(cm[SP3D_CAM_RZ + 2], dz, fmaf(cm[SP3D_CAM_RZ + 1], dxy.y, cm[SP3D_CAM_RZ] * dxy.x));
    const float den = zc + 1e-5f;
    v2f yn;
    {   // div_pair (sp3d_device.h) on the pair; one lane out of the reciprocal form's range sends the whole wave through
        // the IEEE division (same results, no exec-mask juggling on the common path)
        const float ad = __builtin_fabsf(den);
        const unsigned long long okm = __builtin_amdgcn_ballot_w64(ad >= 1e-20f) & __builtin_amdgcn_ballot_w64(ad <= 1e20f);
        if (__builtin_expect(okm != __builtin_amdgcn_ballot_w64(true), 0)) {
            yn = v2f{c2.x / den, c2.y / den};
        } else {
            float r = __builtin_amdgcn_rcpf(den);
            const float e = fmaf(-den, r, 1.0f);
            r = fmaf(e, r, r);
            const v2f nd = pk2(-den), rr = pk2(r);
            v2f q = c2 * rr;
            q = pk_fma(pk_fma(nd, q, c2), rr, q);
            yn = pk_fma(pk_fma(nd, q, c2), rr, q);
        }
    }
    // --- distortion, pixel (cameras.py:44-55)
    const v2f y2 = yn * yn;
    float r2 = y2.x + y2.y;
    r2 = fminf(r2, 1e10f);        // NaN r2 -> 1e10, harmless: a NaN yn already makes px,py NaN
    const float r4 = r2 * r2, r6 = r4 * r2;
    const v2f kr = v2f{cm[SP3D_CAM_K2], cm[SP3D_CAM_K2 + 1]} * v2f{r2, r4};
    const float radial = 1.0f + ((kr.x + kr.y) + cm[SP3D_CAM_K2 + 2] * r6);
    const v2f pt = v2f{cm[SP3D_CAM_P2], cm[SP3D_CAM_P2 + 1]} * v2f{yn.y, yn.x};
    const float tan = pt.x + pt.y;
    const float corr = fmaf(2.0f, tan, radial);       // == radial + 2*tan (2*tan is exact)
    const v2f u = yn * pk2(corr) + v2f{cm[SP3D_CAM_P2 + 1], cm[SP3D_CAM_P2]} * pk2(r2);
    v2f p = v2f{cm[SP3D_CAM_F2], cm[SP3D_CAM_F2 + 1]} * u + v2f{cm[SP3D_CAM_C2], cm[SP3D_CAM_C2 + 1]};
    const float W0 = cm[SP3D_CAM_WH], H0 = cm[SP3D_CAM_WH + 1];
    // in-image test on the unclamped pixel (project_layer.py:78-79): four compares combined on the scalar unit
    const unsigned long long bm = __builtin_amdgcn_ballot_w64(p.x >= 0.0f) & __builtin_amdgcn_ballot_w64(p.y >= 0.0f) &
                                  __builtin_amdgcn_ballot_w64(p.x < W0) & __builtin_amdgcn_ballot_w64(p.y < H0) & inbm;
    const unsigned long long nm = (__builtin_amdgcn_ballot_w64(p.x != p.x) | __builtin_amdgcn_ballot_w64(p.y != p.y)) & inbm;
    o.bm = bm; o.nm = nm;
    const bool aff_ok = affine_tame(cm);
    // nobody in this wave is inside image c: the only thing the rest could still add is a NaN born in the affine stage,
    // which needs a non-finite affine row
    if (EARLY_OUT && (bm & ~nm) == 0ull && aff_ok) return false;
    const float mx = umax_f32(W0, H0);
    p.x = clamp_fast(p.x, -1.0f, mx);             // med3: NaN -> lo, so p is finite from here on
    p.y = clamp_fast(p.y, -1.0f, mx);
    // --- crop affine, flip, heat-map coordinate (project_layer.py:81-90, transforms.py:119-123)
    v2f q = v2f{cm[SP3D_CAM_AXY + 0], cm[SP3D_CAM_AXY + 1]} * pk2(p.x);
    q = pk_fma(v2f{cm[SP3D_CAM_AXY + 2], cm[SP3D_CAM_AXY + 3]}, pk2(p.y), q);
    q = q + v2f{cm[SP3D_CAM_AXY + 4], cm[SP3D_CAM_AXY + 5]};         // == fma(A2, 1.0f, q)
    const float W_in = (float)g.W_in, H_in = (float)g.H_in;
    if (cm[SP3D_CAM_FLIP2] != 0.0f) q.x = W_in - q.x;                // wave-uniform
    v2f gn;
    if (__builtin_expect(aff_ok, 1)) {
        // div_const (sp3d_device.h) on the pair: x / c == fma(fma(-q, c, x), rc, q) with q = x * rc
        const v2f un = q * v2f{(float)g.w, (float)g.h};
        const v2f cin = v2f{W_in, H_in}, rin = v2f{g.rW_in, g.rH_in};
        v2f t = un * rin;
        const v2f uu = pk_fma(pk_fma(-t, cin, un), rin, t);
        const v2f c1 = v2f{(float)(g.w - 1), (float)(g.h - 1)}, r1 = v2f{g.rw1, g.rh1};
        t = uu * r1;
        const v2f gg = pk_fma(pk_fma(-t, c1, uu), r1, t);
        gn = pk_fma(gg, pk2(2.0f), pk2(-1.0f));                      // == gg * 2 - 1 (gg * 2 is exact)
    } else {                                                          // infinite q: div_const would turn it into NaN, IEEE keeps it
        gn.x = q.x * (float)g.w / W_in / (float)(g.w - 1) * 2.0f - 1.0f;
        gn.y = q.y * (float)g.h / H_in / (float)(g.h - 1) * 2.0f - 1.0f;
        o.nm |= (__builtin_amdgcn_ballot_w64(gn.x != gn.x) | __builtin_amdgcn_ballot_w64(gn.y != gn.y)) & inbm;   // non-finite camera tables only
    }
    gn.x = clamp_fast(gn.x, -1.1f, 1.1f);
    gn.y = clamp_fast(gn.y, -1.1f, 1.1f);
    o.i = (gn + pk2(1.0f)) * v2f{(float)(g.w - 1) / 2.0f, (float)(g.h - 1) / 2.0f};
    return true;
}

// Tap record of one voxel in one view: the 2x2 block's origin pixel and the four slot weights (see make_record in
// sp3d_unproject.hip for the clamping rule of blocks that touch the zero padding).
struct RecPk {
    int x0, y0;         // origin of the (clamped) 2x2 block
    v2f wt, wb;         // (w00, w10) top row, (w01, w11) bottom row
};

// x_def, y_def: origin reported for voxels the view does not see (their weights are zero; any in-range pixel will do)
__device__ __forceinline__ RecPk make_record_pk(bool use, v2f i, int w, int h, int x_def = 0, int y_def = 0)
{
    RecPk r;
    const v2f f0 = v2f{floorf(i.x), floorf(i.y)};
    const v2f fr = i - f0;                         // (wx, ny)
    const v2f om = pk2(1.0f) - fr;                 // (ex, sy)
    const int x0 = (int)f0.x, y0 = (int)f0.y;
    const unsigned long long edge = __builtin_amdgcn_ballot_w64(use) &
                                    (__builtin_amdgcn_ballot_w64((unsigned)x0 > (unsigned)(w - 2)) |
                                     __builtin_amdgcn_ballot_w64((unsigned)y0 > (unsigned)(h - 2)));
    if (edge == 0ull) {
        // common case: all four taps in range; voxels not seen by this camera get zero x-weights at the origin
        const v2f fx = v2f{use ? om.x : 0.0f, use ? fr.x : 0.0f};      // (left, right)
        r.wt = pk2(om.y) * fx;
        r.wb = pk2(fr.y) * fx;
        r.x0 = use ? x0 : x_def;
        r.y0 = use ? y0 : y_def;
        return r;
    }
    const int x0c = min(max(x0, 0), w - 2), y0c = min(max(y0, 0), h - 2);
    const int dxs = use ? x0c - x0 : 99, dys = y0c - y0;
    const float fxl = dxs == 0 ? om.x : (dxs == 1 ? fr.x : 0.0f);
    const float fxr = dxs == 0 ? fr.x : (dxs == -1 ? om.x : 0.0f);
    const float fyt = dys == 0 ? om.y : (dys == 1 ? fr.y : 0.0f);
    const float fyb = dys == 0 ? fr.y : (dys == -1 ? om.y : 0.0f);
    r.wt = pk2(fyt) * v2f{fxl, fxr};
    r.wb = pk2(fyb) * v2f{fxl, fxr};
    r.x0 = use ? x0c : x_def; r.y0 = use ? y0c : y_def;
    return r;
}

} // namespace sp3d
