// Batched 3-D real transforms for the frequency-domain 7x7x7 opening conv of V2V-Net (v2v_net.py:10-20 of the reference
// is the conv they replace; see v2v_net._front_fft).  Thin layer over hipFFT/rocFFT plans, kept behind the C ABI so the
// caller controls buffers: torch.fft clones its operand before every real transform on ROCm (the library may overwrite
// C2R inputs) - three ~55 MB copies per root-net step that the plan here does not need, because the padded input
// buffer is only read by the R2C plan and the spectrum is scratch that the C2R plan is allowed to destroy.
//
// Plans are cached per (device, STREAM, direction, batch, SX, SY, SZ); the first call of a shape on a stream builds the
// plan (and its work buffer), so it must not happen inside a stream capture - later calls only enqueue kernels on
// `stream`.  A plan's work buffer belongs to one stream: transforms of the same shape enqueued on two streams (an eager
// call next to a graph captured elsewhere) use different plans and cannot race on scratch memory (advisor, round 2).
// A captured forward borrows the plan its eager warm-up built (a capturing stream cannot allocate): do not run eager
// transforms of that shape on the warm-up stream while a replay of such a graph is in flight on another stream.
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <stdint.h>

#include <map>
#include <mutex>
#include <tuple>

#include "../../include/sp3d.h"
#include "sp3d_twiddles.h"

namespace {
// device, stream, kind (0 R2C 3-D, 1 C2R 3-D, 2 C2C 2-D), batch, SX, SY, SZ
using Key = std::tuple<int, uintptr_t, int, int, int, int, int>;
std::mutex g_mu;
std::map<Key, hipfftHandle> g_plans;

int get_plan(int inverse, int batch, int SX, int SY, int SZ, void *stream, hipfftHandle *out)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    const Key k{dev, (uintptr_t)stream, inverse, batch, SX, SY, SZ};
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_plans.find(k);
    if (it != g_plans.end()) {
        *out = it->second;
        return SP3D_OK;
    }
    // a stream that is being captured cannot build a plan (work-buffer allocation): it borrows the plan an eager
    // (warm-up) call of the same shape built on this device.  PyTorch captures on a private side stream, so this is the
    // normal path of a captured forward.
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (stream && hipStreamIsCapturing((hipStream_t)stream, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive) {
        for (auto &kv : g_plans) {
            const Key &o = kv.first;
            if (std::get<0>(o) == dev && std::get<2>(o) == inverse && std::get<3>(o) == batch && std::get<4>(o) == SX &&
                std::get<5>(o) == SY && std::get<6>(o) == SZ) {
                *out = kv.second;
                return SP3D_OK;
            }
        }
        return SP3D_EFFT;       // no plan of this shape yet: run the forward once eagerly before capturing it
    }
    hipfftHandle p;
    int n[3] = {SX, SY, SZ};
    // dense default layouts: real (batch,SX,SY,SZ), complex (batch,SX,SY,SZ/2+1); kind 2: complex (batch,SX,SY) planes
    const hipfftResult r = inverse == 2 ? hipfftPlanMany(&p, 2, n, nullptr, 1, 0, nullptr, 1, 0, HIPFFT_C2C, batch)
                                        : hipfftPlanMany(&p, 3, n, nullptr, 1, 0, nullptr, 1, 0, inverse ? HIPFFT_C2R : HIPFFT_R2C, batch);
    if (r != HIPFFT_SUCCESS) return SP3D_EFFT;
    g_plans.emplace(k, p);
    *out = p;
    return SP3D_OK;
}

int check_args(const void *a, const void *b, int batch, int SX, int SY, int SZ)
{
    if (batch <= 0 || SX <= 0 || SY <= 0 || SZ <= 0) return SP3D_EINVAL;
    if (!a || !b) return SP3D_ENULL;
    if ((int64_t)batch * SX * SY * SZ > (int64_t)1 << 40) return SP3D_ERANGE;
    return SP3D_OK;
}
} // namespace

extern "C" int sp3d_rfft3d(const float *in, float *out, int batch, int SX, int SY, int SZ, void *stream)
{
    int rc = check_args(in, out, batch, SX, SY, SZ);
    if (rc) return rc;
    hipfftHandle p;
    if ((rc = get_plan(0, batch, SX, SY, SZ, stream, &p))) return rc;
    std::lock_guard<std::mutex> lock(g_mu);              // SetStream + Exec of one plan must not interleave
    if (hipfftSetStream(p, (hipStream_t)stream) != HIPFFT_SUCCESS) return SP3D_EFFT;
    if (hipfftExecR2C(p, const_cast<float *>(in), reinterpret_cast<hipfftComplex *>(out)) != HIPFFT_SUCCESS) return SP3D_EFFT;
    return SP3D_OK;
}

extern "C" int sp3d_irfft3d(float *in, float *out, int batch, int SX, int SY, int SZ, void *stream)
{
    int rc = check_args(in, out, batch, SX, SY, SZ);
    if (rc) return rc;
    hipfftHandle p;
    if ((rc = get_plan(1, batch, SX, SY, SZ, stream, &p))) return rc;
    std::lock_guard<std::mutex> lock(g_mu);
    if (hipfftSetStream(p, (hipStream_t)stream) != HIPFFT_SUCCESS) return SP3D_EFFT;
    if (hipfftExecC2R(p, reinterpret_cast<hipfftComplex *>(in), out) != HIPFFT_SUCCESS) return SP3D_EFFT;
    return SP3D_OK;
}

extern "C" int sp3d_cfft2d(float *data, int batch, int SX, int SY, int inverse, void *stream)
{
    int rc = check_args(data, data, batch, SX, SY, 1);
    if (rc) return rc;
    hipfftHandle p;
    if ((rc = get_plan(2, batch, SX, SY, 0, stream, &p))) return rc;
    std::lock_guard<std::mutex> lock(g_mu);
    if (hipfftSetStream(p, (hipStream_t)stream) != HIPFFT_SUCCESS) return SP3D_EFFT;
    hipfftComplex *d = reinterpret_cast<hipfftComplex *>(data);
    if (hipfftExecC2C(p, d, d, inverse ? HIPFFT_BACKWARD : HIPFFT_FORWARD) != HIPFFT_SUCCESS) return SP3D_EFFT;
    return SP3D_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// z passes of the root grid's opening conv as direct DFTs.  On the 80x80x20 grid the z rows are 20 samples padded to 28:
// a direct 20 -> 15 (complex) DFT with compile-time twiddles is 600 FMAs per row - nothing next to the memory pass -
// and, unlike the library's z pass, it can (a) read the unprojection's channels-last result (the fastest form of that
// kernel) instead of a zero-padded planar buffer, (b) skip the padding rows, (c) write the spectrum with kz as the SLOWEST
// frequency index, so that the x,y passes are dense 2-D complex transforms (sp3d_cfft2d: half the time of the strided
// passes of the 3-D real plan), and on the way back (d) produce only the [0:X,0:Y,0:Z] corner, with the folded-BN shift,
// ReLU and the channels-last layout the 3x3x3 layers want - the crop/epilogue pass disappears.
//   spectrum layout: (B, C, SZ/2+1, SX, SY) complex
// ---------------------------------------------------------------------------------------------------------------
namespace sp3d {

constexpr int ZD_TY = 16;                                 // y rows per workgroup
template <int Z, int C> constexpr int zd_pitch() { return Z * C + 4; }

// x (B,X,Y,Z,C) dense channels-last real -> out (B,Cout,SZ/2+1,SX,SY) complex, Cout <= C channels kept.
// A workgroup owns 16 consecutive positions p = x*SY + y of one sample's (padded) x,y plane - whatever row they fall in -
// so that each of its stores is one whole, aligned 128-byte line of a (c,kz) plane (16 complex values); with tiles cut per
// row every other row's lines were written in two halves by different workgroups (SY*8 B = 5.5 lines): 26 -> 17 us.
template <int Z, int SZ, int C>
__global__ __launch_bounds__(256) void zdft_fwd_cl_kernel(const float *__restrict__ x, float2 *__restrict__ out, int X, int Y,
                                                         int SX, int SY, int Cout)
{
    static_assert(C == 16 && (Z * C) % 4 == 0, "16 channels x 16 positions = 256 threads");
    constexpr int K = SZ / 2 + 1, P = zd_pitch<Z, C>();
    __shared__ __attribute__((aligned(16))) float tile[ZD_TY * P];
    const int plane = SX * SY;
    const int ntile = (plane + ZD_TY - 1) / ZD_TY;
    const int b = blockIdx.x / ntile;
    const int p0 = (blockIdx.x - b * ntile) * ZD_TY;
    const int t = threadIdx.x, yy = t & 15, c = ((t >> 6) << 2) + ((t >> 4) & 3);
    constexpr int R4 = Z * C / 4;
    for (int i = t; i < ZD_TY * R4; i += 256) {
        const int row = i / R4, rem = i - row * R4;
        const int p = p0 + row, px = p / SY, py = p - px * SY;
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (px < X && py < Y) v = reinterpret_cast<const float4 *>(x + ((((int64_t)b * X + px) * Y + py) * Z) * C)[rem];
        *reinterpret_cast<float4 *>(tile + row * P + 4 * rem) = v;
    }
    __syncthreads();
    const int p = p0 + yy, px = p / SY, py = p - px * SY;
    float re[K], im[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { re[k] = 0.0f; im[k] = 0.0f; }
    if (px < X && py < Y && c < Cout) {
        float v[Z];
#pragma unroll
        for (int z = 0; z < Z; ++z) v[z] = tile[yy * P + z * C + c];
        zdft_real<Z, SZ>(v, re, im);
    }
    if (c < Cout && p < plane) {
        float2 *o = out + (((int64_t)b * Cout + c) * K) * plane + p;
#pragma unroll
        for (int k = 0; k < K; ++k) o[(int64_t)k * plane] = make_float2(re[k], im[k]);
    }
}

// in (B,O,SZ/2+1,SX,SY) complex (after the inverse x,y passes) -> y (B,X,Y,Z,O) channels-last real = act(shift[o] + C2R_z(in))
template <int Z, int SZ, int O>
__global__ __launch_bounds__(256) void zdft_inv_cl_kernel(const float2 *__restrict__ in, float *__restrict__ y,
                                                         const float *__restrict__ shift, int X, int Y, int SX, int SY,
                                                         int relu)
{
    static_assert(O == 16 && (Z * O) % 4 == 0, "16 channels x 16 rows = 256 threads");
    constexpr int K = SZ / 2 + 1, P = zd_pitch<Z, O>();
    constexpr Twiddles<SZ> tw{};
    __shared__ __attribute__((aligned(16))) float tile[ZD_TY * P];
    const int nyt = (Y + ZD_TY - 1) / ZD_TY;
    int r = blockIdx.x;
    const int yt = r % nyt; r /= nyt;
    const int xx = r % X;
    const int b = r / X;
    const int y0 = yt * ZD_TY;
    const int t = threadIdx.x, yy = t & 15, o = ((t >> 6) << 2) + ((t >> 4) & 3);
    const int ny = min(ZD_TY, Y - y0);
    if (yy < ny) {
        const float2 *p = in + ((((int64_t)b * O + o) * K) * SX + xx) * SY + y0 + yy;
        float2 f[K];
#pragma unroll
        for (int k = 0; k < K; ++k) f[k] = p[(int64_t)k * SX * SY];
        const float sh = shift[o];
#pragma unroll
        for (int z = 0; z < Z; ++z) {
            // Hermitian input: bins 0 and (SZ even) SZ/2 count once, the others twice; their imaginary parts are ignored
            float acc = f[0].x;
            if (SZ % 2 == 0) acc = (z & 1) ? acc - f[K - 1].x : acc + f[K - 1].x;
            constexpr int KL = (SZ % 2 == 0) ? K - 1 : K;
#pragma unroll
            for (int k = 1; k < KL; ++k) {
                const int m = (k * z) % SZ;
                acc = fmaf(2.0f * tw.c[m], f[k].x, acc);
                acc = fmaf(-2.0f * tw.s[m], f[k].y, acc);
            }
            acc += sh;
            if (relu) acc = acc < 0.0f ? 0.0f : acc;              // NaN propagates like torch.relu
            tile[yy * P + z * O + o] = acc;
        }
    }
    __syncthreads();
    float4 *dst = reinterpret_cast<float4 *>(y + ((((int64_t)b * X + xx) * Y + y0) * Z) * O);
    constexpr int R4 = Z * O / 4;
    for (int i = t; i < ny * R4; i += 256) {
        const int row = i / R4, rem = i - row * R4;
        dst[i] = *reinterpret_cast<const float4 *>(tile + row * P + 4 * rem);
    }
}


// ---------------------------------------------------------------------------------------------------------------
// 88 x 88 complex plane transform in ONE kernel (the x,y passes of the root grid's opening conv): a workgroup keeps a
// whole plane (62 KB) in LDS, so the plane crosses HBM once each way - the library's plan runs two strided out-of-place
// passes (2 x 32 us for 960 planes).  88 = 11 x 8, Cooley-Tukey: n = 8 n1 + n2, k = k1 + 11 k2:
//   A: for every n2 an 11-point DFT over n1 (symmetric form: 100 real FMAs), times W88^(n2 k1), stored at 8 k1 + n2;
//   B: for every k1 an 8-point DFT over n2 (radix 2), stored at k1 + 11 k2 (all reads of B before any of its writes).
// Row pass then column pass; in both the 64 lanes of a wave work on different lines (rows: LDS stride 89 complex -
// conflict-free for 8-byte accesses; columns: consecutive addresses).  inverse: conj in, conj out.  Unnormalised.
// ---------------------------------------------------------------------------------------------------------------
constexpr int F88 = 88, F88_PITCH = 89, F88_NT = 512;

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x)); }
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }          // a * (-i)

template <bool COLS> __device__ __forceinline__ int f88_at(int line, int pos) { return COLS ? pos * F88_PITCH + line : line * F88_PITCH + pos; }

template <bool COLS>
__device__ __forceinline__ void f88_pass(float2 *pl, const float2 *tws, int tid, int nlines)
{
    constexpr Twiddles<11> t11{};
    // only lines < nlines are transformed (row pass: the others are all-zero inputs or outputs nobody reads)
    // A: tasks (line, n2)
    for (int task = tid; task < nlines * 8; task += F88_NT) {
        const int line = task % nlines, n2 = task / nlines;
        float2 a[11];
#pragma unroll
        for (int n1 = 0; n1 < 11; ++n1) a[n1] = pl[f88_at<COLS>(line, 8 * n1 + n2)];
        float2 sm[5], df[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) { sm[j] = cadd(a[j + 1], a[10 - j]); df[j] = csub(a[j + 1], a[10 - j]); }
        float2 out[11];
        out[0] = a[0];
#pragma unroll
        for (int j = 0; j < 5; ++j) out[0] = cadd(out[0], sm[j]);
#pragma unroll
        for (int k = 1; k <= 5; ++k) {
            float2 cpart = a[0], spart = make_float2(0.0f, 0.0f);
#pragma unroll
            for (int j = 1; j <= 5; ++j) {
                const int m = (j * k) % 11;
                cpart.x = fmaf(sm[j - 1].x, t11.c[m], cpart.x); cpart.y = fmaf(sm[j - 1].y, t11.c[m], cpart.y);
                spart.x = fmaf(df[j - 1].x, t11.s[m], spart.x); spart.y = fmaf(df[j - 1].y, t11.s[m], spart.y);
            }
            // X_k = C - i S,  X_{11-k} = C + i S
            out[k] = make_float2(cpart.x + spart.y, cpart.y - spart.x);
            out[11 - k] = make_float2(cpart.x - spart.y, cpart.y + spart.x);
        }
        pl[f88_at<COLS>(line, n2)] = out[0];
#pragma unroll
        for (int k1 = 1; k1 < 11; ++k1)
            pl[f88_at<COLS>(line, 8 * k1 + n2)] = n2 == 0 ? out[k1] : cmul(out[k1], tws[n2 * k1]);
    }
    __syncthreads();
    // B: tasks (line, k1); two tasks per thread at most (88 * 11 = 968 <= 2 * 512)
    float2 b[2][8];
    const float r = 0.70710678118654752440f;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int task = tid + u * F88_NT;
        if (task < nlines * 11) {
            const int line = task % nlines, k1 = task / nlines;
#pragma unroll
            for (int n2 = 0; n2 < 8; ++n2) b[u][n2] = pl[f88_at<COLS>(line, 8 * k1 + n2)];
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int task = tid + u * F88_NT;
        if (task < nlines * 11) {
            const int line = task % nlines, k1 = task / nlines;
            const float2 *q = b[u];
            const float2 u0 = cadd(q[0], q[4]), u1 = csub(q[0], q[4]), u2 = cadd(q[2], q[6]), u3 = mul_mi(csub(q[2], q[6]));
            const float2 u4 = cadd(q[1], q[5]), u5 = csub(q[1], q[5]), u6 = cadd(q[3], q[7]), u7 = mul_mi(csub(q[3], q[7]));
            const float2 v0 = cadd(u0, u2), v2 = csub(u0, u2), v1 = cadd(u1, u3), v3 = csub(u1, u3);
            const float2 v4 = cadd(u4, u6), v6 = csub(u4, u6), v5 = cadd(u5, u7), v7 = csub(u5, u7);
            const float2 w5 = make_float2(r * (v5.x + v5.y), r * (v5.y - v5.x));        // v5 * (1 - i)/sqrt2
            const float2 w6 = mul_mi(v6);
            const float2 w7 = make_float2(r * (v7.y - v7.x), -r * (v7.x + v7.y));       // v7 * (-1 - i)/sqrt2
            pl[f88_at<COLS>(line, k1)] = cadd(v0, v4);
            pl[f88_at<COLS>(line, k1 + 11)] = cadd(v1, w5);
            pl[f88_at<COLS>(line, k1 + 22)] = cadd(v2, w6);
            pl[f88_at<COLS>(line, k1 + 33)] = cadd(v3, w7);
            pl[f88_at<COLS>(line, k1 + 44)] = csub(v0, v4);
            pl[f88_at<COLS>(line, k1 + 55)] = csub(v1, w5);
            pl[f88_at<COLS>(line, k1 + 66)] = csub(v2, w6);
            pl[f88_at<COLS>(line, k1 + 77)] = csub(v3, w7);
        }
    }
    __syncthreads();
}

// rows_in: only the first rows_in rows of every input plane are non-zero (forward transform of zero-padded data; the
// others are not even read); rows_out: only the first rows_out rows of the result are needed (the others are not written)
// tiled != nullptr (forward only): the plane's signal comes from the 4 x 4-tiled z-spectrum the fused unprojection kernel
// writes (sp3d_unproject_fwd_zdft: per plane (rows_in/4, nby, 16) complex, tile-major, no padding) instead of from `data`;
// the zero padding (columns >= 4 * nby, rows >= rows_in) is produced here.  Every 128-byte line read is one tile.
__global__ __launch_bounds__(F88_NT) void cfft2d_88_kernel(float2 *__restrict__ data, int inverse, int rows_in, int rows_out,
                                                           const float2 *__restrict__ tiled, int nby)
{
    extern __shared__ float2 lds88[];
    float2 *pl = lds88, *tws = lds88 + F88 * F88_PITCH;
    constexpr Twiddles<88> t88{};
    const int tid = threadIdx.x;
    float2 *base = data + (int64_t)blockIdx.x * F88 * F88;
    if (tid < F88) tws[tid] = make_float2(t88.c[tid], -t88.s[tid]);
    if (tiled) {
        for (int i = tid; i < F88 * F88_PITCH; i += F88_NT) pl[i] = make_float2(0.0f, 0.0f);
        __syncthreads();
        const int ntile = (rows_in / 4) * nby;
        const float4 *src = reinterpret_cast<const float4 *>(tiled + (int64_t)blockIdx.x * ntile * 16);
        for (int i = tid; i < ntile * 8; i += F88_NT) {          // 8 float4 (= 2 complex each) per tile
            const int tile = i >> 3, w = i & 7;
            const int tx = tile / nby, ty = tile - tx * nby;
            const int rr = 4 * tx + (w >> 1), cc = 4 * ty + 2 * (w & 1);
            const float4 v = src[i];
            pl[rr * F88_PITCH + cc] = make_float2(v.x, v.y);
            pl[rr * F88_PITCH + cc + 1] = make_float2(v.z, v.w);
        }
    } else
    for (int i = tid; i < F88 * F88 / 2; i += F88_NT) {
        const int e = 2 * i, rr = e / F88, cc = e - rr * F88;
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (rr < rows_in) v = reinterpret_cast<const float4 *>(base)[i];
        if (inverse) { v.y = -v.y; v.w = -v.w; }
        pl[rr * F88_PITCH + cc] = make_float2(v.x, v.y);
        pl[rr * F88_PITCH + cc + 1] = make_float2(v.z, v.w);
    }
    __syncthreads();
    if (rows_out < F88) {                           // way back: along x first, then along y for the rows that are read
        f88_pass<true>(pl, tws, tid, F88);
        f88_pass<false>(pl, tws, tid, rows_out);
    } else {                                        // way in: along y for the rows that carry signal, then along x
        f88_pass<false>(pl, tws, tid, rows_in);
        f88_pass<true>(pl, tws, tid, F88);
    }
    for (int i = tid; i < rows_out * F88 / 2; i += F88_NT) {
        const int e = 2 * i, rr = e / F88, cc = e - rr * F88;
        const float2 p0 = pl[rr * F88_PITCH + cc], p1 = pl[rr * F88_PITCH + cc + 1];
        float4 v = make_float4(p0.x, p0.y, p1.x, p1.y);
        if (inverse) { v.y = -v.y; v.w = -v.w; }
        reinterpret_cast<float4 *>(base)[i] = v;
    }
}
} // namespace sp3d

extern "C" int sp3d_zdft_fwd_cl(const float *x, float *spec, int B, int C, int Cout, int X, int Y, int Z, int SX, int SY, int SZ,
                                void *stream)
{
    if (B <= 0 || C <= 0 || Cout <= 0 || Cout > C || X <= 0 || Y <= 0 || Z <= 0 || SX < X || SY < Y || SZ < Z) return SP3D_EINVAL;
    if (!x || !spec) return SP3D_ENULL;
    if (!(C == 16 && Z == 20 && SZ == 28) || (reinterpret_cast<uintptr_t>(x) & 15)) return SP3D_EUNSUPPORTED;
    const int64_t blocks = (int64_t)B * (((int64_t)SX * SY + sp3d::ZD_TY - 1) / sp3d::ZD_TY);
    if (blocks > 0x7fffffff) return SP3D_ERANGE;
    hipLaunchKernelGGL((sp3d::zdft_fwd_cl_kernel<20, 28, 16>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x,
                       reinterpret_cast<float2 *>(spec), X, Y, SX, SY, Cout);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SP3D_OK : (int)e;
}

extern "C" int sp3d_zdft_inv_cl(const float *spec, float *y, const float *shift, int B, int O, int X, int Y, int Z, int SX, int SY,
                                int SZ, int relu, void *stream)
{
    if (B <= 0 || O <= 0 || X <= 0 || Y <= 0 || Z <= 0 || SX < X || SY < Y || SZ < Z) return SP3D_EINVAL;
    if (!spec || !y || !shift) return SP3D_ENULL;
    if (!(O == 16 && Z == 20 && SZ == 28) || (reinterpret_cast<uintptr_t>(y) & 15)) return SP3D_EUNSUPPORTED;
    const int64_t blocks = (int64_t)B * X * ((Y + sp3d::ZD_TY - 1) / sp3d::ZD_TY);
    if (blocks > 0x7fffffff) return SP3D_ERANGE;
    hipLaunchKernelGGL((sp3d::zdft_inv_cl_kernel<20, 28, 16>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float2 *>(spec), y, shift, X, Y, SX, SY, relu);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SP3D_OK : (int)e;
}

// dynamic LDS of cfft2d_88_kernel (a plane + the twiddle row: 63 KB, above the default limit: raised once per process)
static int f88_lds(size_t *bytes)
{
    *bytes = (size_t)(sp3d::F88 * sp3d::F88_PITCH + sp3d::F88) * sizeof(float2);
    static bool attr_set = false;
    if (!attr_set) {
        const hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(sp3d::cfft2d_88_kernel),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)*bytes);
        if (ea != hipSuccess) return (int)ea;
        attr_set = true;
    }
    return SP3D_OK;
}

// sp3d_cfft2d with the zero-padding knowledge of its caller: only the first rows_in rows (x) of every input plane are
// non-zero, only the first rows_out rows of every output plane are needed.  88 x 88 planes run in the single-kernel LDS
// transform above; other sizes fall back to the hipFFT plan (full planes).
extern "C" int sp3d_cfft2d_ex(float *data, int batch, int SX, int SY, int inverse, int rows_in, int rows_out, void *stream)
{
    if (batch <= 0 || SX <= 0 || SY <= 0 || rows_in <= 0 || rows_out <= 0 || rows_in > SX || rows_out > SX) return SP3D_EINVAL;
    if (!data) return SP3D_ENULL;
    if (SX != sp3d::F88 || SY != sp3d::F88 || (reinterpret_cast<uintptr_t>(data) & 15))
        return sp3d_cfft2d(data, batch, SX, SY, inverse, stream);
    size_t lds;
    if (const int ra = f88_lds(&lds)) return ra;
    hipLaunchKernelGGL(sp3d::cfft2d_88_kernel, dim3((unsigned)batch), dim3(sp3d::F88_NT), lds, (hipStream_t)stream,
                       reinterpret_cast<float2 *>(data), inverse, rows_in, rows_out, (const float2 *)nullptr, 0);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SP3D_OK : (int)e;
}

// forward 88 x 88 plane transforms whose input is the 4 x 4-tiled z-spectrum of sp3d_unproject_fwd_zdft:
// tiled (batch, X/4, Y/4, 16) complex -> planes (batch, 88, 88) complex (all 88 rows written)
extern "C" int sp3d_cfft2d_88_tiled(const float *tiled, float *planes, int batch, int X, int Y, void *stream)
{
    if (batch <= 0 || X <= 0 || Y <= 0) return SP3D_EINVAL;
    if (!tiled || !planes) return SP3D_ENULL;
    if ((X & 3) || (Y & 3) || X > sp3d::F88 || Y > sp3d::F88 || (reinterpret_cast<uintptr_t>(tiled) & 15) ||
        (reinterpret_cast<uintptr_t>(planes) & 15))
        return SP3D_EUNSUPPORTED;
    size_t lds;
    if (const int ra = f88_lds(&lds)) return ra;
    hipLaunchKernelGGL(sp3d::cfft2d_88_kernel, dim3((unsigned)batch), dim3(sp3d::F88_NT), lds, (hipStream_t)stream,
                       reinterpret_cast<float2 *>(planes), 0, X, sp3d::F88, reinterpret_cast<const float2 *>(tiled), Y / 4);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SP3D_OK : (int)e;
}
