// sp3d_proposal.hip - the two small reductions either side of the unprojection:
//   * 3D NMS + top-k + index->mm   (core/proposal.py:28-48, cuboid_proposal_net.py:42-52)
//   * soft-argmax over a fine cube  (pose_regression_net.py:19-28)
// Reference: /root/reference/lib/core/proposal.py, lib/models/cuboid_proposal_net.py,
// lib/models/pose_regression_net.py.  The reference issues max_pool3d + eq + mul + topk +
// 6 index ops (and softmax + mul + sum) as separate library kernels with (B,N) temporaries.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/sp3d.h"

namespace sp3d {

constexpr int NMS_THREADS = 256;
// tile of one workgroup: 4 x 8 x 32 voxels (z fastest, as in memory), 4 voxels per thread
constexpr int NT_X = 4, NT_Y = 8, NT_Z = 32;
constexpr int NMS_PER_THREAD = NT_X * NT_Y * NT_Z / NMS_THREADS;

struct Cand {
    float v;
    int i;
};

// Total order of the top-k: larger value first, then LOWER flat index (torch leaves ties unspecified - SURVEY.md
// App. D-6; the oracle applies the same rule); -0 == +0 (a tie), NaN never wins.  As ONE unsigned 64-bit key whose
// maximum is the best candidate: high word = the value's bits mapped to unsigned order (zeros merged), low word =
// ~index; 0 = "no candidate".
__device__ __forceinline__ unsigned long long cand_key(float v, int i)
{
    if (v != v) return 0ull;
    uint32_t b = __float_as_uint(v == 0.0f ? 0.0f : v);
    b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);          // -inf -> 0x007fffff ... +inf -> 0xff800000
    return ((unsigned long long)b << 32) | (uint32_t)(~(uint32_t)i);
}
__device__ __forceinline__ int key_index(unsigned long long k) { return (int)(~(uint32_t)k); }

// wave-wide maximum of a 64-bit key with DPP moves (no LDS round trips: the ds_bpermute form of this reduction was what
// the selection rounds spent their time on); result in every lane
template <int CTRL>
__device__ __forceinline__ unsigned long long key_dpp(unsigned long long k)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)k, CTRL, 0xf, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(k >> 32), CTRL, 0xf, 0xf, false);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long wave_max_key(unsigned long long k)
{
    unsigned long long o;
    o = key_dpp<0xb1>(k); k = o > k ? o : k;      // quad_perm [1,0,3,2]
    o = key_dpp<0x4e>(k); k = o > k ? o : k;      // quad_perm [2,3,0,1]
    o = key_dpp<0x141>(k); k = o > k ? o : k;     // row_half_mirror
    o = key_dpp<0x140>(k); k = o > k ? o : k;     // row_mirror: every lane of a 16-lane row holds the row maximum
    // across the four rows: lanes 15, 31, 47, 63 on the scalar unit
    const uint32_t lo = (uint32_t)k, hi = (uint32_t)(k >> 32);
    unsigned long long r = 0ull;
#pragma unroll
    for (int l = 15; l < 64; l += 16) {
        const unsigned long long v = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)hi, l) << 32) |
                                     (uint32_t)__builtin_amdgcn_readlane((int)lo, l);
        r = v > r ? v : r;
    }
    return r;
}

// Top-k of the candidates a WAVE holds (NPT per lane) by k selection rounds that stay inside the wave: no barrier, no
// LDS round trip (the workgroup-wide rounds this replaces cost two barriers each: 25 + 23 us for the two NMS kernels in
// round 2).  Winner t: key -> okey[t] (0 = none left), value as computed -> oval[t].
template <int NPT>
__device__ __forceinline__ void wave_topk(unsigned long long (&mk)[NPT], const float (&mv)[NPT], int k,
                                          unsigned long long *okey, float *oval)
{
    const int lane = threadIdx.x & 63;
    for (int t = 0; t < k; ++t) {
        unsigned long long best = mk[0];
#pragma unroll
        for (int e = 1; e < NPT; ++e) best = mk[e] > best ? mk[e] : best;
        const unsigned long long win = wave_max_key(best);
#pragma unroll
        for (int e = 0; e < NPT; ++e)
            if (win != 0ull && mk[e] == win) { oval[t] = mv[e]; mk[e] = 0ull; }      // keys are unique: one owner
        if (lane == 0) okey[t] = win;
    }
}

// second stage, ONE wave: the NW * k winners of the workgroup's waves (in LDS) -> the workgroup's top-k
constexpr int NMS_WAVES = NMS_THREADS / 64;
constexpr int TOPK2_PER_LANE = (NMS_WAVES * SP3D_MAX_TOPK + 63) / 64;

// NaN-propagating maximum (the scan form of max_pool3d's window maximum, `m = (v > m || v != v) ? v : m`, is NaN as
// soon as the window holds one): associative, so the 3x3x3 window separates into three 3-tap passes
__device__ __forceinline__ float pmax(float a, float b)
{
    return (a != a || b != b) ? __builtin_nanf("") : (a > b ? a : b);
}

// stage 1: a workgroup owns a 4 x 8 x 32 tile of one sample.  The tile + a one-voxel halo (-inf outside the volume =
// max_pool3d's padding) goes to LDS, the 27-tap window maximum is three separable passes there (6 LDS reads per voxel
// instead of 27 scattered global loads with bounds tests), nms = (x == max) * x (core/proposal.py:28-32), then k
// selection rounds emit the tile's top-k.
__global__ __launch_bounds__(NMS_THREADS) void nms_chunk_topk_kernel(const float *__restrict__ cubes, int X, int Y,
                                                                    int Z, int k, int ntx, int nty, int ntz,
                                                                    Cand *__restrict__ ws)
{
    constexpr int HX = NT_X + 2, HY = NT_Y + 2, HZ = NT_Z + 2;
    __shared__ float sa[HX * HY * HZ];          // tile + halo
    __shared__ float sb[HX * HY * NT_Z];        // after the z pass
    __shared__ float sc[HX * NT_Y * NT_Z];      // after the y pass
    __shared__ unsigned long long skey[NMS_WAVES * SP3D_MAX_TOPK], fkey[SP3D_MAX_TOPK];
    __shared__ float sval[NMS_WAVES * SP3D_MAX_TOPK], fval[SP3D_MAX_TOPK];
    const int b = blockIdx.y, chunk = blockIdx.x, nchunks = gridDim.x;
    const int tz = chunk % ntz, t1 = chunk / ntz, ty = t1 % nty, tx = t1 / nty;
    const int x0 = tx * NT_X, y0 = ty * NT_Y, z0 = tz * NT_Z;
    const float *c = cubes + (size_t)b * X * Y * Z;
    const float ninf = -INFINITY;
    // all loads of the thread first, then the LDS stores: one memory round trip instead of one per iteration (16.3 -> 11.6 us)
    constexpr int NLOAD = (HX * HY * HZ + NMS_THREADS - 1) / NMS_THREADS;
    float ld[NLOAD];
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
        const int e = i * NMS_THREADS + threadIdx.x;
        const int hz = e % HZ, r = e / HZ, hy = r % HY, hx = r / HY;
        const int x = x0 + hx - 1, y = y0 + hy - 1, z = z0 + hz - 1;
        const bool in = e < HX * HY * HZ && (unsigned)x < (unsigned)X && (unsigned)y < (unsigned)Y && (unsigned)z < (unsigned)Z;
        ld[i] = in ? c[((size_t)x * Y + y) * Z + z] : ninf;
    }
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
        const int e = i * NMS_THREADS + threadIdx.x;
        if (e < HX * HY * HZ) sa[e] = ld[i];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < HX * HY * NT_Z; e += NMS_THREADS) {
        const int z = e % NT_Z, r = e / NT_Z;                       // r = hx * HY + hy
        const float *p = sa + r * HZ + z;
        sb[e] = pmax(pmax(p[0], p[1]), p[2]);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < HX * NT_Y * NT_Z; e += NMS_THREADS) {
        const int z = e % NT_Z, r = e / NT_Z, y = r % NT_Y, hx = r / NT_Y;
        const float *p = sb + (hx * HY + y) * NT_Z + z;
        sc[e] = pmax(pmax(p[0], p[NT_Z]), p[2 * NT_Z]);
    }
    __syncthreads();
    float mv[NMS_PER_THREAD];
    int mi[NMS_PER_THREAD];
    unsigned long long mk[NMS_PER_THREAD];
#pragma unroll
    for (int e = 0; e < NMS_PER_THREAD; ++e) {
        const int t = e * NMS_THREADS + threadIdx.x;
        const int z = t % NT_Z, r = t / NT_Z, y = r % NT_Y, x = r / NT_Y;
        const int gx = x0 + x, gy = y0 + y, gz = z0 + z;
        const bool in = gx < X && gy < Y && gz < Z;
        const float *p = sc + (x * NT_Y + y) * NT_Z + z;
        const float m = pmax(pmax(p[0], p[NT_Y * NT_Z]), p[2 * NT_Y * NT_Z]);
        const float self = sa[((x + 1) * HY + (y + 1)) * HZ + z + 1];
        const float keep = (self == m) ? 1.0f : 0.0f;
        mv[e] = keep * self;
        mi[e] = (gx * Y + gy) * Z + gz;
        mk[e] = in ? cand_key(mv[e], mi[e]) : 0ull;
    }
    // per-wave top-k, then wave 0 merges the four lists
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    wave_topk<NMS_PER_THREAD>(mk, mv, k, skey + wave * SP3D_MAX_TOPK, sval + wave * SP3D_MAX_TOPK);
    __syncthreads();
    if (wave != 0) return;
    unsigned long long k2[TOPK2_PER_LANE];
    float v2[TOPK2_PER_LANE];
#pragma unroll
    for (int e = 0; e < TOPK2_PER_LANE; ++e) {
        const int j = e * 64 + lane, w = j / k, t = j - w * k;          // candidate t of wave w
        const bool have = j < NMS_WAVES * k;
        k2[e] = have ? skey[w * SP3D_MAX_TOPK + t] : 0ull;
        v2[e] = have ? sval[w * SP3D_MAX_TOPK + t] : 0.0f;
    }
    wave_topk<TOPK2_PER_LANE>(k2, v2, k, fkey, fval);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    Cand *out = ws + ((size_t)b * nchunks + chunk) * k;
    if (lane < k) {
        Cand w;
        const unsigned long long key = fkey[lane];
        w.v = key != 0ull ? fval[lane] : -INFINITY;
        w.i = key != 0ull ? key_index(key) : 0x7fffffff;
        out[lane] = w;
    }
}

// (Round 4 tried ONE launch - the last tile of a sample to arrive at an atomicInc ticket runs stage 2 in place: bit-equal,
// but 25.3 us against 11.6 + 10.7 us for the two kernels back to back.  Every tile pays an agent-scope release fence (an L2
// write-back) to publish eight candidates across XCDs, and the queued second launch already starts ~2 us after the first
// drains.  Not shipped; profiles/r04_nms_one_launch.md.)
// stage 2: one workgroup per sample merges nchunks*k candidates (per-wave top-k over a strided share held in registers,
// then wave 0 merges the four lists), and lanes 0..k-1 unravel their winner and convert it to mm in parallel.
// Candidate sets beyond NMS_THREADS * MERGE_PER_THREAD are reduced in passes of that size (carry = the running top-k).
// candidates per thread and pass: 16 (4096 per pass), or 8 when everything fits one pass of 2048 (the bench's 2000: half the
// selection work of every round)
template <int MERGE_PER_THREAD>
__global__ __launch_bounds__(NMS_THREADS) void nms_merge_kernel(Cand *__restrict__ ws, int ncand, int X, int Y, int Z,
                                                               int k, float Lx, float Ly, float Lz, float cx, float cy,
                                                               float cz, float *__restrict__ vals,
                                                               int64_t *__restrict__ idx, float *__restrict__ locs,
                                                               float *__restrict__ gcent, float threshold)
{
    __shared__ unsigned long long skey[NMS_WAVES * SP3D_MAX_TOPK], fkey[SP3D_MAX_TOPK];
    __shared__ float sval[NMS_WAVES * SP3D_MAX_TOPK], fval[SP3D_MAX_TOPK];
    const int b = blockIdx.x;
    const Cand *cand = ws + (size_t)b * ncand;
    const int YZ = Y * Z;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x < SP3D_MAX_TOPK) { fkey[threadIdx.x] = 0ull; fval[threadIdx.x] = 0.0f; }
    __syncthreads();
    for (int base = 0; base < ncand; base += NMS_THREADS * MERGE_PER_THREAD) {
        float mv[MERGE_PER_THREAD];
        unsigned long long mk[MERGE_PER_THREAD];
        // all of a thread's candidates in flight first - unconditional loads from clamped indices, one basic block - then the
        // selects: with the load inside `if (i < ncand)` the compiler emitted branch -> load -> s_waitcnt vmcnt(0) per element,
        // MERGE_PER_THREAD dependent round trips in a kernel of one workgroup per sample
        Cand cds[MERGE_PER_THREAD];
#pragma unroll
        for (int e = 0; e < MERGE_PER_THREAD; ++e) {
            const int i = base + e * NMS_THREADS + threadIdx.x;
            cds[e] = cand[min(i, ncand - 1)];
        }
#pragma unroll
        for (int e = 0; e < MERGE_PER_THREAD; ++e) {
            const int i = base + e * NMS_THREADS + threadIdx.x;
            const bool have = i < ncand && cds[e].i != 0x7fffffff;
            mv[e] = i < ncand ? cds[e].v : -INFINITY;
            mk[e] = have ? cand_key(cds[e].v, cds[e].i) : 0ull;
        }
        wave_topk<MERGE_PER_THREAD>(mk, mv, k, skey + wave * SP3D_MAX_TOPK, sval + wave * SP3D_MAX_TOPK);
        __syncthreads();
        if (wave == 0) {
            // the four lists + the running top-k of the earlier passes
            constexpr int NP = ((NMS_WAVES + 1) * SP3D_MAX_TOPK + 63) / 64;
            unsigned long long k2[NP];
            float v2[NP];
#pragma unroll
            for (int e = 0; e < NP; ++e) {
                const int j = e * 64 + lane, w = j / k, t = j - w * k;
                const bool have = j < (NMS_WAVES + 1) * k;
                k2[e] = !have ? 0ull : (w < NMS_WAVES ? skey[w * SP3D_MAX_TOPK + t] : fkey[t]);
                v2[e] = !have ? 0.0f : (w < NMS_WAVES ? sval[w * SP3D_MAX_TOPK + t] : fval[t]);
            }
            __builtin_amdgcn_wave_barrier();
            wave_topk<NP>(k2, v2, k, fkey, fval);
        }
        __syncthreads();
    }
    if (threadIdx.x < k) {
        const int t = threadIdx.x;
        const unsigned long long win = fkey[t];
        const bool have = win != 0ull;
        const int n = have ? key_index(win) : 0;
        const float v = have ? fval[t] : 0.0f;
        const int ix = n / YZ, iy = (n % YZ) / Z, iz = n % Z;          // core/proposal.py:21-23
        vals[(size_t)b * k + t] = v;
        int64_t *ip = idx + ((size_t)b * k + t) * 3;
        ip[0] = ix; ip[1] = iy; ip[2] = iz;
        if (locs) {                                                      // cuboid_proposal_net.py:47-51
            float *lp = locs + ((size_t)b * k + t) * 3;
            lp[0] = ((float)ix / (float)(X - 1) * Lx + cx) - Lx / 2.0f;
            lp[1] = ((float)iy / (float)(Y - 1) * Ly + cy) - Ly / 2.0f;
            lp[2] = ((float)iz / (float)(Z - 1) * Lz + cz) - Lz / 2.0f;
            if (gcent) {   // ProposalLayer.forward in eval (cuboid_proposal_net.py:62-81): [x,y,z, (score>thr)-1, score]
                float *gp = gcent + ((size_t)b * k + t) * 5;
                gp[0] = lp[0]; gp[1] = lp[1]; gp[2] = lp[2];
                gp[3] = (v > threshold ? 1.0f : 0.0f) - 1.0f;
                gp[4] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// soft-argmax: one workgroup per (sample, joint); pass 1 max, pass 2 exp-sum and moments
// ------------------------------------------------------------------------------------------
constexpr int SA_THREADS = 1024;

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}

// GRID = true: voxel centres are regenerated from (centre, size, bins) exactly as the unprojection
// kernel writes them (same fp32 linspace form), so `grids` (12 B/voxel) is never materialised.
struct GridSpec {
    int X, Y, Z;
    float Lx, Ly, Lz, sx, sy, sz;   // box size and fp32 linspace steps (host-computed)
};

__device__ __forceinline__ float lin_at(float L, float step, int n, int i)
{
    const float start = -(L / 2.0f), end = L / 2.0f;
    if (n == 1) return start;
    return (i < n / 2) ? fmaf(step, (float)i, start) : fmaf(-step, (float)(n - 1 - i), end);
}

template <bool GRID>
__global__ __launch_bounds__(SA_THREADS) void soft_argmax_kernel(const float *__restrict__ x,
                                                                const float *__restrict__ grids,
                                                                const float *__restrict__ centers, GridSpec gs,
                                                                float *__restrict__ out, int J, int64_t N, float beta,
                                                                float *__restrict__ stats = nullptr)
{
    __shared__ float red[4][SA_THREADS / 64];
    const int j = blockIdx.x, b = blockIdx.y;
    const float *xv = x + ((size_t)b * J + j) * N;
    const float *gv = GRID ? nullptr : grids + (size_t)b * N * 3;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float m = -INFINITY;
    for (int64_t n = threadIdx.x; n < N; n += SA_THREADS) m = fmaxf(m, beta * xv[n]);
    m = wave_max(m);
    if (lane == 0) red[0][wave] = m;
    __syncthreads();
    m = red[0][0];
#pragma unroll
    for (int w = 1; w < SA_THREADS / 64; ++w) m = fmaxf(m, red[0][w]);
    __syncthreads();
    float cx = 0.f, cy = 0.f, cz = 0.f;
    if (GRID) { cx = centers[3 * b]; cy = centers[3 * b + 1]; cz = centers[3 * b + 2]; }
    const int YZ = gs.Y * gs.Z;
    float s = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int64_t n = threadIdx.x; n < N; n += SA_THREADS) {
        const float e = __expf(beta * xv[n] - m);
        float g0, g1, g2;
        if (GRID) {
            const int nn = (int)n;
            const int ix = nn / YZ, r = nn - ix * YZ, iy = r / gs.Z, iz = r - iy * gs.Z;
            g0 = lin_at(gs.Lx, gs.sx, gs.X, ix) + cx;
            g1 = lin_at(gs.Ly, gs.sy, gs.Y, iy) + cy;
            g2 = lin_at(gs.Lz, gs.sz, gs.Z, iz) + cz;
        } else {
            g0 = gv[3 * n + 0]; g1 = gv[3 * n + 1]; g2 = gv[3 * n + 2];
        }
        s += e;
        a0 = fmaf(e, g0, a0);
        a1 = fmaf(e, g1, a1);
        a2 = fmaf(e, g2, a2);
    }
    s = wave_sum(s); a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
    if (lane == 0) { red[0][wave] = s; red[1][wave] = a0; red[2][wave] = a1; red[3][wave] = a2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float S = 0.f, A0 = 0.f, A1 = 0.f, A2 = 0.f;
        for (int w = 0; w < SA_THREADS / 64; ++w) { S += red[0][w]; A0 += red[1][w]; A1 += red[2][w]; A2 += red[3][w]; }
        float *o = out + ((size_t)b * J + j) * 3;
        o[0] = A0 / S; o[1] = A1 / S; o[2] = A2 / S;
        if (stats) {            // (max of beta x, sum of exp): the backward pass needs no reduction of its own
            stats[((size_t)b * J + j) * 2 + 0] = m;
            stats[((size_t)b * J + j) * 2 + 1] = S;
        }
    }
}

// backward of the soft-argmax w.r.t. x (the voxel centres carry no gradient: proposals are detached):
//   out_d = sum_n p_n grid_nd,  p_n = exp(beta x_n - m) / S   =>   dx_n = beta p_n (g . grid_n - g . out)
// With (m, S) kept by the forward pass this is one elementwise pass: read x, write dx (8 bytes per element).
__global__ __launch_bounds__(256) void soft_argmax_grid_bwd_kernel(const float *__restrict__ x, const float *__restrict__ centers,
                                                                   GridSpec gs, const float *__restrict__ out,
                                                                   const float *__restrict__ stats,
                                                                   const float *__restrict__ grad_out, float *__restrict__ dx,
                                                                   int J, int64_t N, float beta)
{
    const int j = blockIdx.y, b = blockIdx.z;
    const size_t bj = (size_t)b * J + j;
    const float m = stats[bj * 2], rS = 1.0f / stats[bj * 2 + 1];
    const float g0 = grad_out[bj * 3], g1 = grad_out[bj * 3 + 1], g2 = grad_out[bj * 3 + 2];
    const float go = g0 * out[bj * 3] + g1 * out[bj * 3 + 1] + g2 * out[bj * 3 + 2];
    const float cx = centers[3 * b], cy = centers[3 * b + 1], cz = centers[3 * b + 2];
    const int YZ = gs.Y * gs.Z;
    const float *xv = x + bj * N;
    float *dv = dx + bj * N;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
        const int nn = (int)n;
        const int ix = nn / YZ, r = nn - ix * YZ, iy = r / gs.Z, iz = r - iy * gs.Z;
        const float q0 = lin_at(gs.Lx, gs.sx, gs.X, ix) + cx, q1 = lin_at(gs.Ly, gs.sy, gs.Y, iy) + cy,
                    q2 = lin_at(gs.Lz, gs.sz, gs.Z, iz) + cz;
        const float p = __expf(beta * xv[n] - m) * rS;
        dv[n] = beta * p * (fmaf(g0, q0, fmaf(g1, q1, g2 * q2)) - go);
    }
}

} // namespace sp3d

using namespace sp3d;

extern "C" int64_t sp3d_nms_topk_workspace_bytes(int B, int X, int Y, int Z, int k)
{
    if (B <= 0 || X <= 0 || Y <= 0 || Z <= 0 || k <= 0) return 0;
    const int64_t nchunks = (int64_t)((X + NT_X - 1) / NT_X) * ((Y + NT_Y - 1) / NT_Y) * ((Z + NT_Z - 1) / NT_Z);
    return (int64_t)B * nchunks * k * (int64_t)sizeof(Cand);
}

extern "C" int sp3d_nms_proposals(const float *root_cubes, int B, int X, int Y, int Z, int k, const float *grid_size,
                                  const float *grid_center, float threshold, float *vals, int64_t *idx, float *locs,
                                  float *grid_centers, void *workspace, void *stream)
{
    if (B <= 0 || X <= 0 || Y <= 0 || Z <= 0 || k <= 0 || k > SP3D_MAX_TOPK) return SP3D_EINVAL;
    if (!root_cubes || !vals || !idx || !workspace) return SP3D_ENULL;
    if (locs && (!grid_size || !grid_center)) return SP3D_ENULL;
    if (grid_centers && !locs) return SP3D_ENULL;
    const int64_t N = (int64_t)X * Y * Z;
    if (N > 0x7ffffffe) return SP3D_ERANGE;
    const int ntx = (X + NT_X - 1) / NT_X, nty = (Y + NT_Y - 1) / NT_Y, ntz = (Z + NT_Z - 1) / NT_Z;
    const int64_t nch = (int64_t)ntx * nty * ntz;
    if (nch > 0x7fffffff / (SP3D_MAX_TOPK + 1) || B > 65535) return SP3D_ERANGE;
    const int nchunks = (int)nch;
    hipStream_t s = (hipStream_t)stream;
    Cand *ws = (Cand *)workspace;
    hipLaunchKernelGGL(nms_chunk_topk_kernel, dim3(nchunks, B), dim3(NMS_THREADS), 0, s, root_cubes, X, Y, Z, k, ntx, nty,
                       ntz, ws);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    const float L[3] = {locs ? grid_size[0] : 0.f, locs ? grid_size[1] : 0.f, locs ? grid_size[2] : 0.f};
    const float C[3] = {locs ? grid_center[0] : 0.f, locs ? grid_center[1] : 0.f, locs ? grid_center[2] : 0.f};
    if (nchunks * k <= NMS_THREADS * 8)
        hipLaunchKernelGGL(nms_merge_kernel<8>, dim3(B), dim3(NMS_THREADS), 0, s, ws, nchunks * k, X, Y, Z, k, L[0], L[1],
                           L[2], C[0], C[1], C[2], vals, idx, locs, grid_centers, threshold);
    else
        hipLaunchKernelGGL(nms_merge_kernel<16>, dim3(B), dim3(NMS_THREADS), 0, s, ws, nchunks * k, X, Y, Z, k, L[0], L[1],
                           L[2], C[0], C[1], C[2], vals, idx, locs, grid_centers, threshold);
    e = hipGetLastError();
    return e == hipSuccess ? SP3D_OK : (int)e;
}

extern "C" int sp3d_nms_topk(const float *root_cubes, int B, int X, int Y, int Z, int k, const float *grid_size,
                             const float *grid_center, float *vals, int64_t *idx, float *locs, void *workspace,
                             void *stream)
{
    return sp3d_nms_proposals(root_cubes, B, X, Y, Z, k, grid_size, grid_center, 0.0f, vals, idx, locs, nullptr, workspace,
                              stream);
}

extern "C" int sp3d_soft_argmax(const float *x, const float *grids, float *out, int Bv, int J, int64_t N, float beta,
                                void *stream)
{
    if (Bv <= 0 || J <= 0 || N <= 0) return SP3D_EINVAL;
    if (!x || !grids || !out) return SP3D_ENULL;
    GridSpec gs = {};
    hipLaunchKernelGGL(soft_argmax_kernel<false>, dim3(J, Bv), dim3(SA_THREADS), 0, (hipStream_t)stream, x, grids,
                       (const float *)nullptr, gs, out, J, N, beta);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SP3D_OK : (int)e;
}

static GridSpec make_grid_spec(const float *grid_size, int X, int Y, int Z)
{
    GridSpec gs;
    gs.X = X; gs.Y = Y; gs.Z = Z; gs.Lx = grid_size[0]; gs.Ly = grid_size[1]; gs.Lz = grid_size[2];
    const int n[3] = {X, Y, Z};
    float st[3];
    for (int a = 0; a < 3; ++a) {
        volatile float start = -(grid_size[a] / 2.0f), end = grid_size[a] / 2.0f;
        volatile float diff = end - start;
        st[a] = n[a] > 1 ? diff / (float)(n[a] - 1) : 0.0f;
    }
    gs.sx = st[0]; gs.sy = st[1]; gs.sz = st[2];
    return gs;
}

extern "C" int sp3d_soft_argmax_grid_train(const float *x, const float *centers, const float *grid_size, int X, int Y, int Z,
                                           float *out, float *stats, int Bv, int J, float beta, void *stream)
{
    if (Bv <= 0 || J <= 0 || X <= 0 || Y <= 0 || Z <= 0 || Bv > 65535) return SP3D_EINVAL;
    if (!x || !centers || !grid_size || !out) return SP3D_ENULL;
    const int64_t N = (int64_t)X * Y * Z;
    if (N > 0x7ffffffe) return SP3D_ERANGE;
    const GridSpec gs = make_grid_spec(grid_size, X, Y, Z);
    hipLaunchKernelGGL(soft_argmax_kernel<true>, dim3(J, Bv), dim3(SA_THREADS), 0, (hipStream_t)stream, x,
                       (const float *)nullptr, centers, gs, out, J, N, beta, stats);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SP3D_OK : (int)e;
}

extern "C" int sp3d_soft_argmax_grid(const float *x, const float *centers, const float *grid_size, int X, int Y, int Z,
                                     float *out, int Bv, int J, float beta, void *stream)
{
    return sp3d_soft_argmax_grid_train(x, centers, grid_size, X, Y, Z, out, nullptr, Bv, J, beta, stream);
}

extern "C" int sp3d_soft_argmax_grid_bwd(const float *x, const float *centers, const float *grid_size, int X, int Y, int Z,
                                         const float *out, const float *stats, const float *grad_out, float *grad_x, int Bv,
                                         int J, float beta, void *stream)
{
    if (Bv <= 0 || J <= 0 || X <= 0 || Y <= 0 || Z <= 0 || Bv > 65535 || J > 65535) return SP3D_EINVAL;
    if (!x || !centers || !grid_size || !out || !stats || !grad_out || !grad_x) return SP3D_ENULL;
    const int64_t N = (int64_t)X * Y * Z;
    if (N > 0x7ffffffe) return SP3D_ERANGE;
    const GridSpec gs = make_grid_spec(grid_size, X, Y, Z);
    int64_t nb = (N + 256 * 8 - 1) / (256 * 8);
    if (nb > 256) nb = 256;
    hipLaunchKernelGGL(soft_argmax_grid_bwd_kernel, dim3((unsigned)nb, (unsigned)J, (unsigned)Bv), dim3(256), 0,
                       (hipStream_t)stream, x, centers, gs, out, stats, grad_out, grad_x, J, N, beta);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SP3D_OK : (int)e;
}
