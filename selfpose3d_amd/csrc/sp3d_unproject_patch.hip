// sp3d_unproject_patch.hip - unprojection for DENSE voxel grids: the image patch a voxel block samples is staged in LDS.
//
// Reference path: /root/reference/lib/models/project_layer.py:42-102 (same arithmetic as sp3d_unproject.hip, bit for bit).
//
// Why a second kernel.  On the per-person 64^3 cubes (31 mm voxels, ~1.8 heat-map pixels apart) and the 160x160x40 stress
// grid (50 mm, ~2.7 px) neighbouring voxels sample the same pixels over and over: an 8x8x4 voxel block issues 1024 bilinear
// taps per view into a patch of ~190 pixels (tools/sim_patch.py).  The brick kernel (sp3d_unproject.hip) sends every tap
// through the texture-address path (64 B per tap = 2.9 M 1-KiB wave-loads for ten 64^3 cubes: 60 % TA-busy next to 61 %
// VALU-busy, profiles/r02_pmc_unproject_fine64.json).  Here the patch crosses the TA once, as full-width contiguous
// LDS-DMA rows (global_load_lds_dwordx4: no VGPR round trip), and the taps are ds_read_b128 from LDS (4 cycles per
// wave-instruction instead of >= 16).  Round 2 built this single-buffered behind two barriers and found it latency-bound
// (DESIGN.md 7b); this form is a pipeline:
//
//   workgroup = 4 waves = 2x2x1 bricks of 4x4x4 voxels (8x8x4 block); LDS = 2 patch buffers (20 KB each) + tap records
//   prologue  wave w projects the block's 8 CORNER voxels through cameras w, w+4, ... (lane & 7 = corner, camera record
//             in SGPRs) -> patch rectangle per view = bounding box of the corners' tap blocks.
//             A perspective image of a convex block is the convex hull of its corners' images, so this box holds every
//             tap of the block - up to lens distortion and rounding, which is why every wave VERIFIES its own taps
//             against the box (two compares per voxel) and falls back to the direct global gather for that view if one
//             is outside.  Blocks whose corners are not all inside the image, or whose box exceeds the buffer, use the
//             direct gather too (workgroup-uniform).  No cross-lane reduction in the view loop.
//   view c    wait for patch(c) [own LDS-DMAs: vmcnt(0)], ONE workgroup barrier, issue the LDS-DMAs of patch(c+1) into
//             the other buffer (one instruction = one patch row, SGPR base + lane * 16: no per-lane address arithmetic),
//             fetch camera c+1 into SGPRs, gather view c's taps from LDS, project view c+1 (packed-fp32 projection,
//             sp3d_proj_pk.h) while they arrive, interpolate.  The DMAs of patch(c+1) fly during the whole of view c's
//             work.  (The camera fetch sits BEFORE the tap reads: scalar loads and LDS reads share lgkmcnt, a fetch
//             behind them would make the projection wait for every tap.)
//   epilogue  view fusion and result stores as in the brick kernel (channels-last: 64 B per voxel from the gather mapping;
//             planar: through LDS, 16-byte z-runs).
//
// Results are bit-identical to the other forward kernels (tests/test_gpu_parity.py::test_nhwc_variants_bit_identical).
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

constexpr int PNW = 4;                     // waves per workgroup
constexpr int PBX = 8, PBY = 8, PBZ = 4;   // voxel block of a workgroup
constexpr int PCAP_BYTES = 20480;          // one patch buffer
constexpr int PREC = 2 * 5 * 64;           // per-wave tap records, floats: weights [buf][voxel][4], then offsets [buf][voxel]
constexpr int POFF = 2 * 4 * 64;
constexpr int PDESC = 2;                   // ints per view descriptor

// view descriptor (LDS, written once per workgroup by the corner pass; read back through v_readfirstlane)
//   [0] px0 | py0 << 16   [1] pw | ph << 16, 0 = direct gather (no patch)

__device__ __forceinline__ void lds_dma16(uint32_t voff, uint32_t lds_dst, const char *gbase)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(lds_dst), "s"(gbase)
                 : "memory");
}

template <int JP, bool OUTCL, typename TI, typename TO>
__global__ __launch_bounds__(64 * PNW, 3) void unproject_patch_kernel(Views hm, const float *__restrict__ cam,
                                                                     const float *__restrict__ centers,
                                                                     const uint8_t *__restrict__ valid,
                                                                     float *__restrict__ cubes, float *__restrict__ grids,
                                                                     Geom g, int wgs_per_sample, int nby, int nbz)
{
    constexpr int NQ = JP / 4;
    constexpr int PXB = JP * (int)sizeof(TI);             // bytes per pixel
    static_assert(PXB % 16 == 0, "a pixel must be a whole number of 16-byte chunks");
    constexpr int QB = 4 * (int)sizeof(TI);               // bytes of one lane's channel quad
    extern __shared__ __attribute__((aligned(16))) char psm[];
    char *pbuf = psm;                                                       // [2][PCAP_BYTES]
    float *recs = reinterpret_cast<float *>(psm + 2 * PCAP_BYTES);          // [PNW][PREC]
    int *desc = reinterpret_cast<int *>(psm + 2 * PCAP_BYTES + PNW * PREC * 4);   // [SP3D_MAX_VIEWS][PDESC]

    int b, wg;
    if (!xcd_map(blockIdx.x, g.B, wgs_per_sample, g.xcd_chunk, b, wg, g.xcd_order)) return;
#ifdef SP3D_PATCH_TL
    unsigned long long *tl = g_patch_tl ? g_patch_tl + ((size_t)blockIdx.x * PNW + (threadIdx.x >> 6)) * 32 : nullptr;
    const int lane_tl = threadIdx.x & 63;
    if (tl && lane_tl == 0) tl[0] = __builtin_readcyclecounter();
#endif
    const int bz = wg % nbz, t = wg / nbz, by = t % nby, bx = t / nby;
    const int bs = g.sample_of ? g.sample_of[b] : b;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int X0 = bx * PBX, Y0 = by * PBY, Z0 = bz * PBZ;
    const int x0 = X0 + (wave & 1) * 4, y0 = Y0 + (wave >> 1) * 4, z0 = Z0;
    TO *cb = reinterpret_cast<TO *>(cubes) + (OUTCL ? (size_t)b * g.J * g.N : (size_t)b * g.sB);
    float *ws = recs + wave * PREC;
    int *wsi = reinterpret_cast<int *>(ws);
    float4 *ws4 = reinterpret_cast<float4 *>(ws);

    // P1 mapping: this lane's voxel; gather mapping: slot i of lane group g16 is voxel (x0 + i, y0 + g16/4, z0 + g16%4)
    const int lx = lane >> 4, ly = (lane >> 2) & 3, lz = lane & 3;
    const int vx = x0 + lx, vy = y0 + ly, vz = z0 + lz;
    const bool inb = vx < g.X && vy < g.Y && vz < g.Z;
    const int n = (min(vx, g.X - 1) * g.Y + min(vy, g.Y - 1)) * g.Z + min(vz, g.Z - 1);
    const int g16 = lane >> 2, q = lane & 3;
    const bool qact = q < NQ;
    const int gy = y0 + (g16 >> 2), gz = z0 + (g16 & 3);
    const bool ginb = gy < g.Y && gz < g.Z;
    const int gn0 = (min(x0, g.X - 1) * g.Y + min(gy, g.Y - 1)) * g.Z + min(gz, g.Z - 1);        // + i * YZ

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

    const float ctrx = centers[3 * b + 0], ctry = centers[3 * b + 1], ctrz = centers[3 * b + 2];
    const size_t rowb = (size_t)g.w * PXB;                  // bytes per heat-map row

    // ---- corner pass: patch rectangle per view (wave w: views w, w + PNW, ...) ------------------------------------
    {
        const int k = lane & 7;
        const int cx = min(X0 + ((k & 1) ? PBX - 1 : 0), g.X - 1), cy = min(Y0 + ((k & 2) ? PBY - 1 : 0), g.Y - 1);
        const int cz = min(Z0 + ((k & 4) ? PBZ - 1 : 0), g.Z - 1);
        const float px = linspace_step(g.Lx, g.stepx, g.X, cx) + ctrx;
        const float py = linspace_step(g.Ly, g.stepy, g.Y, cy) + ctry;
        const float pz = linspace_step(g.Lz, g.stepz, g.Z, cz) + ctrz;
        for (int v = wave; v < g.V; v += PNW) {
            const float *cm = cam + ((size_t)bs * g.V + v) * SP3D_CAM_STRIDE;
            P1State st;
            project_pk<false>(cm, g, px, py, pz, true, st);
            const unsigned long long um = st.bm & ~st.nm;
            const RecPk r = make_record_pk(lane_of(um), st.i, g.w, g.h);
            int xmin = r.x0, xmax = r.x0, ymin = r.y0, ymax = r.y0;
#pragma unroll
            for (int m = 1; m < 8; m <<= 1) {
                xmin = min(xmin, __shfl_xor(xmin, m)); xmax = max(xmax, __shfl_xor(xmax, m));
                ymin = min(ymin, __shfl_xor(ymin, m)); ymax = max(ymax, __shfl_xor(ymax, m));
            }
            if (lane == 0) {
                const int pw = xmax - xmin + 2, ph = ymax - ymin + 2;
                const bool fits = (um == ~0ull) && pw * ph * PXB <= PCAP_BYTES && xmax < 65535 && ymax < 65535;
                desc[v * PDESC + 0] = xmin | (ymin << 16);
                desc[v * PDESC + 1] = fits ? (pw | (ph << 16)) : 0;
            }
        }
    }
    __syncthreads();
    PTL(1);

    // LDS-DMA of view c's patch: one instruction = (a 1-KiB piece of) one patch row: global address = SGPR row base +
    // lane * 16, LDS destination = M0 + lane * 16 (rows packed back to back: row pitch pw * PXB bytes).  Rows are dealt
    // round-robin to the waves.
    const uint32_t lane16 = (uint32_t)lane * 16u;
    auto issue_patch = [&](int c) {
        const int *d = desc + c * PDESC;
        const int wh = __builtin_amdgcn_readfirstlane(d[1]);
        if (!wh) return;
        const int xy = __builtin_amdgcn_readfirstlane(d[0]);
        const int px0 = xy & 0xffff, py0 = (int)((unsigned)xy >> 16), pw = wh & 0xffff, ph = (int)((unsigned)wh >> 16);
        const char *gb = reinterpret_cast<const char *>(hm.p[c]) + ((size_t)bs * g.h + py0) * rowb + (size_t)px0 * PXB;
        const uint32_t rb = (uint32_t)pw * (uint32_t)PXB;                     // bytes per patch row
        const uint32_t dst0 = (uint32_t)(size_t)(pbuf + (c & 1) * PCAP_BYTES);
        if (rb <= 1024u) {
            if (lane16 < rb)
                for (int r = wave; r < ph; r += PNW) lds_dma16(lane16, dst0 + (uint32_t)r * rb, gb + (size_t)r * rowb);
        } else {
            for (int r = wave; r < ph; r += PNW)
                for (uint32_t o = 0; o < rb; o += 1024u)
                    if (lane16 + o < rb) lds_dma16(lane16, dst0 + (uint32_t)r * rb + o, gb + (size_t)r * rowb + o);
        }
    };

    const float x = linspace_step(g.Lx, g.stepx, g.X, min(vx, g.X - 1)) + ctrx;
    const float y = linspace_step(g.Ly, g.stepy, g.Y, min(vy, g.Y - 1)) + ctry;
    const float z = linspace_step(g.Lz, g.stepz, g.Z, min(vz, g.Z - 1)) + ctrz;
    if (grids && inb) {
        float *gp = grids + ((size_t)b * g.N + n) * 3;
        gp[0] = x; gp[1] = y; gp[2] = z;
    }
    uint32_t mymask = 0;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i][0] = acc[i][1] = acc[i][2] = acc[i][3] = 0.0f;

    // P1(c): tap records of this wave's voxels in view c.  Returns 0: no voxel of the wave sees the view, 1: records
    // hold global byte offsets (direct gather), 2: records hold LDS byte addresses inside patch buffer c & 1.
    auto P1 = [&](int c, const float *cm) -> int {
        const int *d = desc + c * PDESC;
        P1State st;
        const bool go = project_pk(cm, g, x, y, z, inb, st);
        add_mask(mymask, st.bm);
        if (st.nm != 0ull && lane_of(st.nm)) mymask |= 0x80000000u;
        if (!go) return 0;
        const unsigned long long um = st.bm & ~st.nm;
        if (um == 0ull) return 0;
        const bool use = lane_of(um);
        const int xy = __builtin_amdgcn_readfirstlane(d[0]), wh = __builtin_amdgcn_readfirstlane(d[1]);
        const int px0 = xy & 0xffff, py0 = (int)((unsigned)xy >> 16), pw = wh & 0xffff, ph = (int)((unsigned)wh >> 16);
        const bool patch = wh != 0;
        const RecPk r = make_record_pk(use, st.i, g.w, g.h, patch ? px0 : 0, patch ? py0 : 0);
        const int v = (c & 1) * 64 + lane;
        ws4[v] = make_float4(r.wt.x, r.wt.y, r.wb.x, r.wb.y);
        if (patch) {
            const uint32_t xr = (uint32_t)(r.x0 - px0), yr = (uint32_t)(r.y0 - py0);
            const unsigned long long out = __builtin_amdgcn_ballot_w64(xr > (uint32_t)(pw - 2)) |
                                           __builtin_amdgcn_ballot_w64(yr > (uint32_t)(ph - 2));
            if (out == 0ull) {
                wsi[POFF + v] = (int)(__umul24(__umul24(yr, (uint32_t)pw) + xr, (uint32_t)PXB) + (uint32_t)((c & 1) * PCAP_BYTES));
                return 2;
            }
        }
        wsi[POFF + v] = (int)__umul24((unsigned)PXB, __umul24((unsigned)r.y0, (unsigned)g.w) + (unsigned)r.x0);
        return 1;
    };

    const uint32_t qoff = qact ? (uint32_t)QB * (uint32_t)q : 0u;      // this lane's channel quad, bytes
    issue_patch(0);
    int have = P1(0, cam + ((size_t)bs * g.V) * SP3D_CAM_STRIDE);
    PTL(2);
#pragma unroll 1
    for (int c = 0; c < g.V; ++c) {
        const int cur = have;
        // patch(c) has landed (this wave's DMAs) for every wave, and every wave is done reading the other buffer
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        PTL(3 + 5 * (c < 5 ? c : 4));
        if (c + 1 < g.V) issue_patch(c + 1);
        // camera c+1 -> SGPRs now, so that the projection below does not wait (lgkmcnt) behind the tap reads
        float cmr[SP3D_CAM_STRIDE];
        {
            const float *cm = cam + ((size_t)bs * g.V + min(c + 1, g.V - 1)) * SP3D_CAM_STRIDE;
#pragma unroll
            for (int i = 0; i < SP3D_CAM_STRIDE; ++i) cmr[i] = cm[i];
            asm volatile("; camera record resident" :: "s"(cmr[0]), "s"(cmr[SP3D_CAM_STRIDE - 1]));
        }
        __builtin_amdgcn_sched_barrier(0);
        PTL(4 + 5 * (c < 5 ? c : 4));
        const int rb = (c & 1) * 64 + g16;
        float4 t00[4], t10[4], t01[4], t11[4];
        if (cur == 2) {
            // patch row pitch, bytes
            const uint32_t prow = (uint32_t)(__builtin_amdgcn_readfirstlane(desc[c * PDESC + 1]) & 0xffff) * (uint32_t)PXB;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t a0 = (uint32_t)wsi[POFF + rb + 16 * k] + qoff;       // byte offset inside psm
                const uint32_t a1 = a0 + prow;
                t00[k] = Store4<TI>::load(reinterpret_cast<const TI *>(psm + a0));
                t10[k] = Store4<TI>::load(reinterpret_cast<const TI *>(psm + a0) + JP);
                t01[k] = Store4<TI>::load(reinterpret_cast<const TI *>(psm + a1));
                t11[k] = Store4<TI>::load(reinterpret_cast<const TI *>(psm + a1) + JP);
            }
        } else if (cur == 1) {
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
        } else {        // (defined on every path: otherwise the registers of the previous view are carried along by copies)
#pragma unroll
            for (int k = 0; k < 4; ++k) t00[k] = t10[k] = t01[k] = t11[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        __builtin_amdgcn_sched_barrier(0);
        PTL(5 + 5 * (c < 5 ? c : 4));
        if (c + 1 < g.V) have = P1(c + 1, cmr);      // VALU work while the taps arrive
        __builtin_amdgcn_sched_barrier(0);
        PTL(6 + 5 * (c < 5 ? c : 4));
        if (cur) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 wq = ws4[rb + 16 * i];                 // (w00, w10, w01, w11)
                // ATen's bilinear chain per channel: fma(se, wse, fma(sw, wsw, fma(ne, wne, nw * wnw)))
                v2f lo = v2f{t00[i].x, t00[i].y} * pk2(wq.x), hi = v2f{t00[i].z, t00[i].w} * pk2(wq.x);
                lo = pk_fma(v2f{t10[i].x, t10[i].y}, pk2(wq.y), lo); hi = pk_fma(v2f{t10[i].z, t10[i].w}, pk2(wq.y), hi);
                lo = pk_fma(v2f{t01[i].x, t01[i].y}, pk2(wq.z), lo); hi = pk_fma(v2f{t01[i].z, t01[i].w}, pk2(wq.z), hi);
                lo = pk_fma(v2f{t11[i].x, t11[i].y}, pk2(wq.w), lo); hi = pk_fma(v2f{t11[i].z, t11[i].w}, pk2(wq.w), hi);
                const v2f a0 = v2f{acc[i][0], acc[i][1]} + lo, a1 = v2f{acc[i][2], acc[i][3]} + hi;
                acc[i][0] = a0.x; acc[i][1] = a0.y; acc[i][2] = a1.x; acc[i][3] = a1.y;
            }
        }
#ifdef SP3D_PATCH_TL
        if (tl && lane == 0) { tl[7 + 5 * (c < 5 ? c : 4)] = __builtin_readcyclecounter(); tl[31] = (unsigned long long)cur; }
#endif
    }
    PTL(28);

    // ---- view fusion (project_layer.py:96-99) on the gather mapping -----------------------------------------------
    __builtin_amdgcn_wave_barrier();
    const float den_l = (float)(mymask & 0x7fffffffu) + 1e-6f;
    const float rden_l = (mymask & 0x80000000u) ? 0.0f : 1.0f / den_l;
    if (!OUTCL) __syncthreads();                    // every wave is done with the patch buffers: they become the result tile
    float *tile = reinterpret_cast<float *>(psm) + wave * (JP * 64);       // planar: [channel][voxel of the brick]
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
                Store4<TO>::store_nt(cb + (size_t)gn * g.J + 4 * q, o);
            }
        } else if (qact) {
#pragma unroll
            for (int k = 0; k < 4; ++k) tile[(4 * q + k) * 64 + 16 * i + g16] = fuse_rcp(acc[i][k], den, rden);
        }
    }
#ifdef SP3D_PATCH_TL
    __builtin_amdgcn_s_waitcnt(0);
    PTL(29);
#endif
    if (OUTCL) return;
    __syncthreads();
    // planar result: thread -> (channel phase jj = tid / 64, brick w = (tid / 16) % 4, column of the brick = tid % 16);
    // a piece = the 4 z of one (channel, column)
    {
        const int jj = tid >> 6, w4 = (tid >> 4) & 3, col = tid & 15;
        const int sx = X0 + (w4 & 1) * 4 + (col >> 2), sy = Y0 + (w4 >> 1) * 4 + (col & 3), sz = Z0;
        if (sx >= g.X || sy >= g.Y || sz >= g.Z) return;
        const float *src = reinterpret_cast<const float *>(psm) + w4 * (JP * 64) + col * 4;
        TO *dst = cb + (size_t)sx * g.sX + (size_t)sy * g.sY + sz;
        if (g.vec4 && (g.Z & 3) == 0) {
            for (int j = jj; j < g.J; j += 4) {
                const float4 o = *reinterpret_cast<const float4 *>(src + j * 64);
                Store4<TO>::store_nt(dst + (size_t)j * g.sJ, o);
            }
        } else {
            const int nz = min(PBZ, g.Z - sz);
            for (int j = jj; j < g.J; j += 4)
                for (int k = 0; k < nz; ++k) Store4<TO>::store1(dst + (size_t)j * g.sJ + k, src[j * 64 + k]);
        }
    }
}

int launch_patch(const Views &v, int Jp, const float *cam, const float *centers, const uint8_t *valid, float *cubes,
                 float *grids, const Geom &g, bool out_cl, int io, hipStream_t s)
{
    if (Jp != 16) return SP3D_EUNSUPPORTED;
    const int nbx = (g.X + PBX - 1) / PBX, nby = (g.Y + PBY - 1) / PBY, nbz = (g.Z + PBZ - 1) / PBZ;
    const int wgs = nbx * nby * nbz;
    Geom gb = g;
    {   // 2-4 chunks of consecutive workgroups (x-slabs of the volume) per serving XCD
        const int xps = (g.B <= 8 && (8 % g.B) == 0) ? 8 / g.B : 1;
        int k = 1;
        while (k * 2 * xps * 2 <= wgs) k *= 2;
        gb.xcd_chunk = k;
    }
    const size_t lds = 2 * PCAP_BYTES + PNW * PREC * 4 + SP3D_MAX_VIEWS * PDESC * 4;
    dim3 grid(xcd_grid_blocks(gb.B, wgs, gb.xcd_chunk)), block(64 * PNW);
#define SP3D_PATCH(CL_, TI_, TO_) \
    hipLaunchKernelGGL((unproject_patch_kernel<16, CL_, TI_, TO_>), grid, block, lds, s, v, cam, centers, valid, cubes, grids, gb, wgs, nby, nbz)
    switch ((io & 3) * 2 + (out_cl ? 1 : 0)) {
    case 0: SP3D_PATCH(false, float, float); break;
    case 1: SP3D_PATCH(true, float, float); break;
    case 2: SP3D_PATCH(false, bf16_t, float); break;
    case 3: SP3D_PATCH(true, bf16_t, float); break;
    case 4: SP3D_PATCH(false, float, bf16_t); break;
    case 5: SP3D_PATCH(true, float, bf16_t); break;
    case 6: SP3D_PATCH(false, bf16_t, bf16_t); break;
    default: SP3D_PATCH(true, bf16_t, bf16_t); break;
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
