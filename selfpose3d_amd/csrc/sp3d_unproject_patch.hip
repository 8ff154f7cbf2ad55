// sp3d_unproject_patch.hip - unprojection for DENSE voxel grids: each wave stages the image patch its brick samples in LDS.
//
// Reference path: /root/reference/lib/models/project_layer.py:42-102 (same arithmetic as sp3d_unproject.hip, bit for bit).
//
// Why a second kernel.  On the per-person 64^3 cubes (31 mm voxels, ~1.8 heat-map pixels apart) and the 160x160x40 stress
// grid (50 mm, ~2.7 px) neighbouring voxels sample the same pixels over and over: the 256 bilinear taps of a 4x4x4 brick
// fall into a patch of ~75 pixels (tools/sim_patch.py).  The brick kernel (sp3d_unproject.hip) still sends every tap
// through the texture-address path: 16 one-KiB wave-loads per (wave, view), 2.9 M of them for ten 64^3 cubes, and the TA
// is what binds there (60 % busy next to 61 % VALU-busy, profiles/r02_pmc_unproject_fine64.json).  Here the patch
// crosses the TA once, as ~9 row-wide LDS-DMA instructions (global_load_lds_dwordx4: SGPR row base + lane * 16, no VGPR
// round trip, no per-lane address arithmetic), and the taps are ds_read_b128 from LDS.
//
// History: round 2 built a WORKGROUP-shared patch (8x8x4 voxels, single buffer, two barriers per view) and found it
// latency-bound; round 3 first rebuilt it double-buffered with one barrier per view and LDS-DMA issued a
// view ahead (git 713daf7): still 188 us against 110 us for bricks on ten 64^3 cubes - the per-view barrier chains four
// waves' LDS round trips (tools/patch_timeline.py: 6 000 cycles per view against 2 300 of VALU work).  This form has no
// barrier at all: the patch is WAVE-private (one 4x4x4 brick = one single-wave workgroup, as in the brick kernel).
//
//   P1(c+1)   lane = voxel: packed-fp32 projection (sp3d_proj_pk.h), tap record -> the wave's LDS slice.  The patch
//             rectangle of view c+1 = bounding box of the tap blocks of the brick's 8 corner voxels (8 v_readlane +
//             scalar min/max): the perspective image of a convex brick is the convex hull of its corners' images, so the
//             box holds every tap - up to lens distortion and rounding, which is why every voxel's tap block is
//             VERIFIED against the box (two compares) and the wave falls back to the direct global gather for that view
//             if one is outside, if a corner is not inside the image, or if the box exceeds the 8 KB buffer.
//   view c    P1(c+1) runs while the LDS-DMAs of patch(c) are in flight; wait for them (vmcnt: the wave's own), read the
//             taps (ds_read_b128), and as soon as they are in registers issue the DMAs of patch(c+1) into the same
//             buffer; interpolate (they fly during the FMAs and the next projection).
//   epilogue  view fusion and channels-last result stores as in the brick kernel (64 B per voxel from the gather mapping).
//
// Results are bit-identical to the other forward kernels (tests/test_gpu_parity.py::test_patch_kernel_bit_identical).
#include "sp3d_device.h"
#include "sp3d_proj_pk.h"
#include "sp3d_unproject_patch.h"

namespace sp3d {

// measurement only (tools/patch_timeline.py builds a copy of the library with -DSP3D_PATCH_TL): s_memtime stamps per wave
#ifdef SP3D_PATCH_TL
__device__ unsigned long long *g_patch_tl = nullptr;
#define PTL(slot) do { if (tl && lane == 0) tl[slot] = __builtin_readcyclecounter(); } while (0)
#else
#define PTL(slot) do { } while (0)
#endif

constexpr int WCAP = 8192;                 // bytes of a wave's patch buffer
constexpr int PREC = 2 * 5 * 64;           // tap records, floats: weights [buf][voxel][4], then offsets [buf][voxel]
constexpr int POFF = 2 * 4 * 64;
constexpr int WLDS_BYTES = WCAP + PREC * 4;
// lanes of the brick's corner voxels: lane = lx * 16 + ly * 4 + lz, (lx, ly, lz) in {0, 3}^3
constexpr unsigned long long CORNERS = (1ull << 0) | (1ull << 3) | (1ull << 12) | (1ull << 15) | (1ull << 48) |
                                       (1ull << 51) | (1ull << 60) | (1ull << 63);

__device__ __forceinline__ void lds_dma16(uint32_t voff, uint32_t lds_dst, const char *gbase)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(lds_dst), "s"(gbase)
                 : "memory");
}

// a wave-uniform pointer the compiler cannot prove uniform -> SGPR pair
__device__ __forceinline__ const char *uniform_ptr(const char *p)
{
    const unsigned long long u = (unsigned long long)(size_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    return reinterpret_cast<const char *>((size_t)(((unsigned long long)hi << 32) | lo));
}

// min / max of wave-uniform values on the scalar unit (the compiler otherwise moves such chains to the VALU)
__device__ __forceinline__ uint32_t smin_u32(uint32_t a, uint32_t b) { uint32_t r; asm("s_min_u32 %0, %1, %2" : "=s"(r) : "s"(a), "s"(b) : "scc"); return r; }
__device__ __forceinline__ uint32_t smax_u32(uint32_t a, uint32_t b) { uint32_t r; asm("s_max_u32 %0, %1, %2" : "=s"(r) : "s"(a), "s"(b) : "scc"); return r; }

// patch rectangle of one (wave, view), wave-uniform
struct PatchBox {
    int x0, y0, pw, ph;
    uint32_t pitch;         // bytes between patch rows in LDS: pw * PXB, or 512 when two rows travel per DMA instruction
};

template <int JP, bool OUTCL, typename TI, typename TO>
__global__ __launch_bounds__(64, 4) void unproject_wpatch_kernel(Views hm, const float *__restrict__ cam,
                                                                const float *__restrict__ centers,
                                                                const uint8_t *__restrict__ valid,
                                                                float *__restrict__ cubes, float *__restrict__ grids,
                                                                Geom g, int wgs_per_sample, int nby, int nbz)
{
    constexpr int NQ = JP / 4;
    constexpr int PXB = JP * (int)sizeof(TI);             // bytes per pixel
    static_assert(PXB % 16 == 0, "a pixel must be a whole number of 16-byte chunks");
    constexpr int QB = 4 * (int)sizeof(TI);               // bytes of one lane's channel quad
    static_assert(OUTCL, "channels-last results only (planar results: brick stacks, sp3d_unproject.hip)");
    extern __shared__ __attribute__((aligned(16))) char psm[];
    char *pbuf = psm;                                                       // [WCAP]
    float *ws = reinterpret_cast<float *>(psm + WCAP);                      // [PREC]
    int *wsi = reinterpret_cast<int *>(ws);
    float4 *ws4 = reinterpret_cast<float4 *>(ws);

    int b, wg;
    if (!xcd_map_fast(blockIdx.x, g, b, wg)) return;
#ifdef SP3D_PATCH_TL
    unsigned long long *tl = g_patch_tl ? g_patch_tl + (size_t)blockIdx.x * 32 : nullptr;
#endif
    const int lane = threadIdx.x;
    PTL(0);
    int bz, t, bx, by;                                 // z slowest, as in the brick kernel
    udiv_magic((uint32_t)wg, (uint32_t)g.bk_nxy, g.bk_magic_nxy, bz, t);
    udiv_magic((uint32_t)t, (uint32_t)g.bk_nby, g.bk_magic_nby, bx, by);
    const int bs = __builtin_amdgcn_readfirstlane(g.sample_of ? g.sample_of[b] : b);
    const int x0 = bx * 4, y0 = by * 4, z0 = bz * 4;
    TO *cb = reinterpret_cast<TO *>(cubes) + (size_t)b * g.J * g.N;

    // P1 mapping: this lane's voxel; gather mapping: slot i of lane group g16 is voxel (x0 + i, y0 + g16/4, z0 + g16%4)
    const int lx = lane >> 4, ly = (lane >> 2) & 3, lz = lane & 3;
    const int vx = x0 + lx, vy = y0 + ly, vz = z0 + lz;
    const bool inb = vx < g.X && vy < g.Y && vz < g.Z;
    const int n = (min(vx, g.X - 1) * g.Y + min(vy, g.Y - 1)) * g.Z + min(vz, g.Z - 1);
    const int g16 = lane >> 2, q = lane & 3;
    const bool qact = q < NQ;
    const int gy = y0 + (g16 >> 2), gz = z0 + (g16 & 3);
    const bool ginb = gy < g.Y && gz < g.Z;
    const int gn0 = (x0 * g.Y + min(gy, g.Y - 1)) * g.Z + min(gz, g.Z - 1);        // + i * YZ

    if (!valid[b]) { // skipped sample: zeros (project_layer.py:48,51,54)
        if (inb) {
            for (int j = 0; j < g.J; ++j) Store4<TO>::store1(cb + (size_t)n * g.J + j, 0.0f);
            if (grids) {
                float *gp = grids + ((size_t)b * g.N + n) * 3;
                gp[0] = 0.0f; gp[1] = 0.0f; gp[2] = 0.0f;
            }
            if (g.pass_mask) g.pass_mask[(size_t)b * g.N + n] = 0;
        }
        return;
    }

    const size_t rowb = (size_t)g.w * PXB;                  // bytes per heat-map row
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
    const uint32_t lane16 = (uint32_t)lane * 16u;

    const unsigned long long inbm = __builtin_amdgcn_ballot_w64(inb);
    // P1(c): tap records of the brick in view c.  Returns 0: no voxel sees the view, 1: the records hold global byte
    // offsets (direct gather), 2: they hold byte offsets inside the LDS patch described by `box`.
    auto P1 = [&](int c, PatchBox &box) -> int {
        const float *cm = cam + ((size_t)bs * g.V + c) * SP3D_CAM_STRIDE;
        P1State st;
        const bool go = project_pk(cm, g, x, y, z, inbm, st);
        add_mask(mymask, st.bm);
        if (st.nm != 0ull && lane_of(st.nm)) mymask |= 0x80000000u;
        if (!go) return 0;
        const unsigned long long um = st.bm & ~st.nm;
        if (um == 0ull) return 0;
        const bool use = lane_of(um);
        const RecPk r = make_record_pk(use, st.i, g.w, g.h);
        const int v = (c & 1) * 64 + lane;
        ws4[v] = make_float4(r.wt.x, r.wt.y, r.wb.x, r.wb.y);
        if ((um & CORNERS) == CORNERS) {
            // bounding box of the 8 corner voxels' tap origins, on the scalar unit
            const int xy = r.x0 | (r.y0 << 16);
            uint32_t xmin = 0xffffu, xmax = 0, ymin = 0xffffu, ymax = 0;
#define SP3D_CORNER(L_) { const uint32_t s_ = (uint32_t)__builtin_amdgcn_readlane(xy, L_); const uint32_t cx_ = s_ & 0xffffu, cy_ = s_ >> 16; \
                          xmin = smin_u32(xmin, cx_); xmax = smax_u32(xmax, cx_); ymin = smin_u32(ymin, cy_); ymax = smax_u32(ymax, cy_); }
            SP3D_CORNER(0) SP3D_CORNER(3) SP3D_CORNER(12) SP3D_CORNER(15) SP3D_CORNER(48) SP3D_CORNER(51) SP3D_CORNER(60) SP3D_CORNER(63)
#undef SP3D_CORNER
            const int pw = (int)(xmax - xmin) + 2, ph = (int)(ymax - ymin) + 2;
            const uint32_t rb = (uint32_t)pw * (uint32_t)PXB;
            const uint32_t pitch = rb <= 512u ? 512u : rb;
            if (pitch * (uint32_t)ph <= (uint32_t)WCAP) {
                const uint32_t xr = (uint32_t)r.x0 - xmin, yr = (uint32_t)r.y0 - ymin;
                const unsigned long long out = um & (__builtin_amdgcn_ballot_w64(xr > (uint32_t)(pw - 2)) |
                                                     __builtin_amdgcn_ballot_w64(yr > (uint32_t)(ph - 2)));
                if (out == 0ull) {
                    box.x0 = (int)xmin; box.y0 = (int)ymin; box.pw = pw; box.ph = ph; box.pitch = pitch;
                    wsi[POFF + v] = use ? (int)(__umul24(yr, pitch) + __umul24(xr, (uint32_t)PXB)) : 0;
                    return 2;
                }
            }
        }
        wsi[POFF + v] = (int)__umul24((unsigned)PXB, __umul24((unsigned)r.y0, (unsigned)g.w) + (unsigned)r.x0);
        return 1;
    };

    // LDS-DMA of view c's patch: one instruction = one patch row (SGPR row base + lane * 16 -> M0 + lane * 16), or two
    // rows when a row is at most 512 bytes (lanes 32-63 fetch the next row; LDS row pitch 512)
    auto issue_patch = [&](int c, const PatchBox &bx) {
        const char *gb = uniform_ptr(reinterpret_cast<const char *>(hm.p[c]) + ((size_t)bs * g.h + bx.y0) * rowb + (size_t)bx.x0 * PXB);
        const uint32_t rb = (uint32_t)bx.pw * (uint32_t)PXB;                  // bytes per patch row
        const uint32_t dst0 = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)pbuf);
        if (bx.pitch == 512u) {
            const uint32_t l16 = (uint32_t)(lane & 31) * 16u;
            const uint32_t voff = l16 + ((lane & 32) ? (uint32_t)rowb : 0u);
            const bool hi = (lane & 32) != 0;
            for (int r = 0; r < bx.ph; r += 2)
                if (l16 < rb && (!hi || r + 1 < bx.ph)) lds_dma16(voff, dst0 + (uint32_t)r * 512u, gb + (size_t)r * rowb);
        } else if (rb <= 1024u) {
            if (lane16 < rb)
                for (int r = 0; r < bx.ph; ++r) lds_dma16(lane16, dst0 + (uint32_t)r * rb, gb + (size_t)r * rowb);
        } else {
            for (int r = 0; r < bx.ph; ++r)
                for (uint32_t o = 0; o < rb; o += 1024u)
                    if (lane16 + o < rb) lds_dma16(lane16, dst0 + (uint32_t)r * rb + o, gb + (size_t)r * rowb + o);
        }
    };

    const uint32_t qoff = qact ? (uint32_t)QB * (uint32_t)q : 0u;      // this lane's channel quad, bytes
    PatchBox box, nbox;
    box.x0 = box.y0 = box.pw = box.ph = 0; box.pitch = 0; nbox = box;
    int have = P1(0, box);
    if (have == 2) issue_patch(0, box);
    PTL(1);
#pragma unroll 1
    for (int c = 0; c < g.V; ++c) {
        const int cur = have;
        // next view's projection while this view's patch is on its way
        if (c + 1 < g.V) have = P1(c + 1, nbox);
        PTL(2 + 4 * (c < 6 ? c : 5));
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int rb = (c & 1) * 64 + g16;
        const bool next_patch = c + 1 < g.V && have == 2;
        // interpolation of view c from the taps in registers (ATen's bilinear chain per channel:
        // fma(se, wse, fma(sw, wsw, fma(ne, wne, nw * wnw)))), accumulated over the views in order
        auto interp = [&](const float4 (&t00)[4], const float4 (&t10)[4], const float4 (&t01)[4], const float4 (&t11)[4]) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 wq = ws4[rb + 16 * i];                 // (w00, w10, w01, w11)
                v2f lo = v2f{t00[i].x, t00[i].y} * pk2(wq.x), hi = v2f{t00[i].z, t00[i].w} * pk2(wq.x);
                lo = pk_fma(v2f{t10[i].x, t10[i].y}, pk2(wq.y), lo); hi = pk_fma(v2f{t10[i].z, t10[i].w}, pk2(wq.y), hi);
                lo = pk_fma(v2f{t01[i].x, t01[i].y}, pk2(wq.z), lo); hi = pk_fma(v2f{t01[i].z, t01[i].w}, pk2(wq.z), hi);
                lo = pk_fma(v2f{t11[i].x, t11[i].y}, pk2(wq.w), lo); hi = pk_fma(v2f{t11[i].z, t11[i].w}, pk2(wq.w), hi);
                const v2f a0 = v2f{acc[i][0], acc[i][1]} + lo, a1 = v2f{acc[i][2], acc[i][3]} + hi;
                acc[i][0] = a0.x; acc[i][1] = a0.y; acc[i][2] = a1.x; acc[i][3] = a1.y;
            }
        };
        // (the three cases are separate code paths on purpose: tap registers that are live across a merge point get
        // copied or zero-filled on EVERY path by the register allocator - 64 extra VALU instructions per view)
        if (cur == 2) {
            float4 t00[4], t10[4], t01[4], t11[4];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // patch(c): this wave's own DMAs
            const uint32_t prow = box.pitch;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t a0 = (uint32_t)wsi[POFF + rb + 16 * k] + qoff;       // byte offset inside the patch
                const uint32_t a1 = a0 + prow;
                t00[k] = Store4<TI>::load(reinterpret_cast<const TI *>(pbuf + a0));
                t10[k] = Store4<TI>::load(reinterpret_cast<const TI *>(pbuf + a0) + JP);
                t01[k] = Store4<TI>::load(reinterpret_cast<const TI *>(pbuf + a1));
                t11[k] = Store4<TI>::load(reinterpret_cast<const TI *>(pbuf + a1) + JP);
            }
            PTL(3 + 4 * (c < 6 ? c : 5));
            if (next_patch) {
                // the taps of view c are in registers: the buffer is free for patch(c+1)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                issue_patch(c + 1, nbox);
            }
            PTL(4 + 4 * (c < 6 ? c : 5));
            interp(t00, t10, t01, t11);
        } else if (cur == 1) {
            if (next_patch) issue_patch(c + 1, nbox);                   // nobody reads the buffer in this view
            float4 t00[4], t10[4], t01[4], t11[4];
            const char *vb = reinterpret_cast<const char *>(hm.p[c]) + (size_t)bs * g.h * rowb;
            const char *vb2 = vb + rowb;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t off = (uint32_t)wsi[POFF + rb + 16 * k] + qoff;
                t00[k] = Store4<TI>::load(reinterpret_cast<const TI *>(vb + off));
                t10[k] = Store4<TI>::load(reinterpret_cast<const TI *>(vb + off) + JP);
                t01[k] = Store4<TI>::load(reinterpret_cast<const TI *>(vb2 + off));
                t11[k] = Store4<TI>::load(reinterpret_cast<const TI *>(vb2 + off) + JP);
            }
            interp(t00, t10, t01, t11);
        } else if (next_patch) {
            issue_patch(c + 1, nbox);
        }
        if (next_patch) box = nbox;
        PTL(5 + 4 * (c < 6 ? c : 5));
    }
    PTL(28);

    // ---- view fusion (project_layer.py:96-99) on the gather mapping -----------------------------------------------
    __builtin_amdgcn_wave_barrier();
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
        if (qact && 4 * q < g.J && vin) {
            float4 o;
            o.x = fuse_rcp(acc[i][0], den, rden); o.y = fuse_rcp(acc[i][1], den, rden);
            o.z = fuse_rcp(acc[i][2], den, rden); o.w = fuse_rcp(acc[i][3], den, rden);
            Store4<TO>::store_nt(cb + (size_t)gn * g.J + 4 * q, o);
        }
    }
#ifdef SP3D_PATCH_TL
    __builtin_amdgcn_s_waitcnt(0);
    PTL(29);
#endif
}

int launch_patch(const Views &v, int Jp, const float *cam, const float *centers, const uint8_t *valid, float *cubes,
                 float *grids, const Geom &g, bool out_cl, int io, hipStream_t s)
{
    if (Jp != 16 || !out_cl) return SP3D_EUNSUPPORTED;
    const int nbx = (g.X + 3) / 4, nby = (g.Y + 3) / 4, nbz = (g.Z + 3) / 4;
    const int wgs = nbx * nby * nbz;
    Geom gb = g;
    {   // 2-4 chunks of consecutive workgroups (x-slabs of the volume) per serving XCD
        const int xps = (g.B <= 8 && (8 % g.B) == 0) ? 8 / g.B : 1;
        int k = 1;
        while (k * 2 * xps * 2 <= wgs) k *= 2;
        gb.xcd_chunk = k;
    }
    set_xcd_fields(gb, wgs);
    set_brick_fields(gb, nbx * nby, nby);
    const size_t lds = WLDS_BYTES;
    dim3 grid(xcd_grid_blocks(gb.B, wgs, gb.xcd_chunk)), block(64);
#define SP3D_PATCH(TI_, TO_) \
    hipLaunchKernelGGL((unproject_wpatch_kernel<16, true, TI_, TO_>), grid, block, lds, s, v, cam, centers, valid, cubes, grids, gb, wgs, nby, nbz)
    switch (io & 3) {
    case 0: SP3D_PATCH(float, float); break;
    case 1: SP3D_PATCH(bf16_t, float); break;
    case 2: SP3D_PATCH(float, bf16_t); break;
    default: SP3D_PATCH(bf16_t, bf16_t); break;
    }
#undef SP3D_PATCH
    return SP3D_OK;
}

} // namespace sp3d

#ifdef SP3D_PATCH_TL
extern "C" int sp3d_debug_set_patch_timeline(void *dev_buffer)
{
    unsigned long long *p = (unsigned long long *)dev_buffer;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(sp3d::g_patch_tl), &p, sizeof(p));
}
#endif
