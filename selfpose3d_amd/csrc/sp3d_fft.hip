// Batched 3-D real transforms for the frequency-domain 7x7x7 opening conv of V2V-Net (v2v_net.py:10-20 of the reference
// is the conv they replace; see v2v_net._front_fft).  Thin layer over hipFFT/rocFFT plans, kept behind the C ABI so the
// caller controls buffers: torch.fft clones its operand before every real transform on ROCm (the library may overwrite
// C2R inputs) - three ~55 MB copies per root-net step that the plan here does not need, because the padded input
// buffer is only read by the R2C plan and the spectrum is scratch that the C2R plan is allowed to destroy.
//
// Plans are cached per (device, direction, batch, SX, SY, SZ); the first call of a shape builds the plan (and its work
// buffer), so it must not happen inside a stream capture - later calls only enqueue kernels on `stream`.
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <stdint.h>

#include <map>
#include <mutex>
#include <tuple>

#include "../../include/sp3d.h"

namespace {
using Key = std::tuple<int, int, int, int, int, int>;       // device, kind (0 R2C 3-D, 1 C2R 3-D, 2 C2C 2-D), batch, SX, SY, SZ
std::mutex g_mu;
std::map<Key, hipfftHandle> g_plans;

int get_plan(int inverse, int batch, int SX, int SY, int SZ, hipfftHandle *out)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    const Key k{dev, inverse, batch, SX, SY, SZ};
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_plans.find(k);
    if (it != g_plans.end()) {
        *out = it->second;
        return SP3D_OK;
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
    if ((rc = get_plan(0, batch, SX, SY, SZ, &p))) return rc;
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
    if ((rc = get_plan(1, batch, SX, SY, SZ, &p))) return rc;
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
    if ((rc = get_plan(2, batch, SX, SY, 0, &p))) return rc;
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

constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr double cx_sin(double x) { double x2 = x * x, t = x, s = x; for (int n = 1; n < 16; ++n) { t *= -x2 / ((2.0 * n) * (2.0 * n + 1.0)); s += t; } return s; }
constexpr double cx_cos(double x) { double x2 = x * x, t = 1.0, s = 1.0; for (int n = 1; n < 16; ++n) { t *= -x2 / ((2.0 * n - 1.0) * (2.0 * n)); s += t; } return s; }
template <int SZ> struct Twiddles {
    float c[SZ], s[SZ];
    constexpr Twiddles() : c(), s()
    {
        for (int m = 0; m < SZ; ++m) {
            double a = 2.0 * kPi * m / SZ;
            if (a > kPi) a -= 2.0 * kPi;
            double cv = cx_cos(a), sv = cx_sin(a);
            if (cv < 1e-13 && cv > -1e-13) cv = 0.0;
            if (sv < 1e-13 && sv > -1e-13) sv = 0.0;
            c[m] = (float)cv; s[m] = (float)sv;
        }
    }
};

constexpr int ZD_TY = 16;                                 // y rows per workgroup
template <int Z, int C> constexpr int zd_pitch() { return Z * C + 4; }

// x (B,X,Y,Z,C) dense channels-last real -> out (B,Cout,SZ/2+1,SX,SY) complex, Cout <= C channels kept
template <int Z, int SZ, int C>
__global__ __launch_bounds__(256) void zdft_fwd_cl_kernel(const float *__restrict__ x, float2 *__restrict__ out, int X, int Y,
                                                         int SX, int SY, int Cout)
{
    static_assert(C == 16 && (Z * C) % 4 == 0, "16 channels x 16 rows = 256 threads");
    constexpr int K = SZ / 2 + 1, P = zd_pitch<Z, C>();
    constexpr Twiddles<SZ> tw{};
    __shared__ __attribute__((aligned(16))) float tile[ZD_TY * P];
    const int nyt = (SY + ZD_TY - 1) / ZD_TY;
    int r = blockIdx.x;
    const int yt = r % nyt; r /= nyt;
    const int xx = r % SX;
    const int b = r / SX;
    const int y0 = yt * ZD_TY;
    const int t = threadIdx.x, yy = t & 15, c = ((t >> 6) << 2) + ((t >> 4) & 3);
    const int ny = (xx < X) ? max(0, min(ZD_TY, Y - y0)) : 0;          // rows of this tile that carry signal
    if (ny > 0) {
        const float4 *src = reinterpret_cast<const float4 *>(x + ((((int64_t)b * X + xx) * Y + y0) * Z) * C);
        constexpr int R4 = Z * C / 4;
        for (int i = t; i < ny * R4; i += 256) {
            const int row = i / R4, rem = i - row * R4;
            *reinterpret_cast<float4 *>(tile + row * P + 4 * rem) = src[i];
        }
    }
    __syncthreads();
    float re[K], im[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { re[k] = 0.0f; im[k] = 0.0f; }
    if (yy < ny && c < Cout) {
        float v[Z];
#pragma unroll
        for (int z = 0; z < Z; ++z) v[z] = tile[yy * P + z * C + c];
#pragma unroll
        for (int k = 0; k < K; ++k) {
#pragma unroll
            for (int z = 0; z < Z; ++z) {
                const int m = (k * z) % SZ;
                re[k] = fmaf(v[z], tw.c[m], re[k]);
                im[k] = fmaf(v[z], -tw.s[m], im[k]);
            }
        }
    }
    if (c < Cout && y0 + yy < SY) {
        float2 *o = out + ((((int64_t)b * Cout + c) * K) * SX + xx) * SY + y0 + yy;
#pragma unroll
        for (int k = 0; k < K; ++k) o[(int64_t)k * SX * SY] = make_float2(re[k], im[k]);
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

} // namespace sp3d

extern "C" int sp3d_zdft_fwd_cl(const float *x, float *spec, int B, int C, int Cout, int X, int Y, int Z, int SX, int SY, int SZ,
                                void *stream)
{
    if (B <= 0 || C <= 0 || Cout <= 0 || Cout > C || X <= 0 || Y <= 0 || Z <= 0 || SX < X || SY < Y || SZ < Z) return SP3D_EINVAL;
    if (!x || !spec) return SP3D_ENULL;
    if (!(C == 16 && Z == 20 && SZ == 28) || (reinterpret_cast<uintptr_t>(x) & 15)) return SP3D_EUNSUPPORTED;
    const int64_t blocks = (int64_t)B * SX * ((SY + sp3d::ZD_TY - 1) / sp3d::ZD_TY);
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
