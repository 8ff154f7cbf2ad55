// sp3d_fftconv.hip - the channel contraction of an FFT-domain convolution.
//
// The V2V nets open with a 7x7x7 stride-1 Conv3d (reference: lib/models/v2v_net.py:113-117 `Basic3DBlock(in, 16, 7)`);
// on the 80x80x20 root grid that one layer is 90 GFLOP per batch of 4 and a third of the whole root-net step as a
// direct convolution.  In inference the build runs it in the frequency domain instead (selfpose3d_amd/v2v_net.py:
// real FFT of the zero-padded input through torch.fft / rocFFT, this kernel, inverse FFT): 20x fewer flops, and
// closer to the float64 result than the direct fp32 convolution.  What is left between the two FFTs is
//     Y[b, o, f] = sum_c X[b, c, f] * W[o, c, f]            (complex, f = frequency bin, W = conj(FFT(weights)))
// a memory-bound pass over W (O*C*F*8 bytes, read once per batch chunk) - no MFMA: per bin it is a 4x16x16 product.
//
// lane = frequency bin (consecutive bins -> 512-B coalesced float2 loads of every (b,c) / (o,c) plane); one workgroup
// = 256 bins x OG output channels x up to BB batch entries; accumulators stay in registers.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sp3d.h"

namespace sp3d {

// Y[i, j, f] = sum_k P[i, k, f] * Q[j, k, f]   (optionally conj(P) / conj(Q)); every operand has the frequency bins
// contiguous, its two leading dimensions at the given strides (in complex elements).  The three products of the
// frequency-domain convolution are instances of it:
//   forward      Y[b,o] = sum_c X[b,c] conj(W^)[o,c]          P = X, Q = W^ (conjQ)
//   grad input   Gx[b,c] = sum_o Gy[b,o] W^[o,c]              P = Gy, Q = W^ read as [c][o]
//   grad weight  Gw[o,c] = sum_b conj(Gy[b,o]) X[b,c]         P = Gy read as [o][b] (conjP), Q = X read as [c][b]
template <int BB, int OG, bool CONJP, bool CONJQ>
__global__ __launch_bounds__(256) void freq_contract_kernel(const float2 *__restrict__ P, const float2 *__restrict__ Q,
                                                           float2 *__restrict__ Y, int I, int J, int K, int64_t F,
                                                           int64_t sPi, int64_t sPk, int64_t sQj, int64_t sQk)
{
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (f >= F) return;
    const int j0 = blockIdx.y * OG, i0 = blockIdx.z * BB;
    float2 acc[BB][OG];
#pragma unroll
    for (int b = 0; b < BB; ++b)
#pragma unroll
        for (int o = 0; o < OG; ++o) acc[b][o] = make_float2(0.0f, 0.0f);
    for (int k = 0; k < K; ++k) {
        float2 x[BB], w[OG];
#pragma unroll
        for (int b = 0; b < BB; ++b) {
            x[b] = (i0 + b < I) ? P[(int64_t)(i0 + b) * sPi + (int64_t)k * sPk + f] : make_float2(0.0f, 0.0f);
            if (CONJP) x[b].y = -x[b].y;
        }
#pragma unroll
        for (int o = 0; o < OG; ++o) {
            w[o] = (j0 + o < J) ? Q[(int64_t)(j0 + o) * sQj + (int64_t)k * sQk + f] : make_float2(0.0f, 0.0f);
            if (CONJQ) w[o].y = -w[o].y;
        }
#pragma unroll
        for (int b = 0; b < BB; ++b)
#pragma unroll
            for (int o = 0; o < OG; ++o) {
                acc[b][o].x = fmaf(x[b].x, w[o].x, acc[b][o].x);
                acc[b][o].x = fmaf(-x[b].y, w[o].y, acc[b][o].x);
                acc[b][o].y = fmaf(x[b].x, w[o].y, acc[b][o].y);
                acc[b][o].y = fmaf(x[b].y, w[o].x, acc[b][o].y);
            }
    }
#pragma unroll
    for (int b = 0; b < BB; ++b)
#pragma unroll
        for (int o = 0; o < OG; ++o)
            if (i0 + b < I && j0 + o < J) Y[((int64_t)(i0 + b) * J + (j0 + o)) * F + f] = acc[b][o];
}

template <bool CP, bool CQ>
static void launch_contract(const float2 *p, const float2 *q, float2 *y, int I, int J, int K, int64_t F, int64_t sPi,
                            int64_t sPk, int64_t sQj, int64_t sQk, hipStream_t s)
{
    const unsigned gx = (unsigned)((F + 255) / 256);
    if (I >= 3)
        hipLaunchKernelGGL((freq_contract_kernel<4, 8, CP, CQ>), dim3(gx, (J + 7) / 8, (I + 3) / 4), dim3(256), 0, s, p, q, y, I, J, K, F, sPi, sPk, sQj, sQk);
    else if (I == 2)
        hipLaunchKernelGGL((freq_contract_kernel<2, 8, CP, CQ>), dim3(gx, (J + 7) / 8, 1), dim3(256), 0, s, p, q, y, I, J, K, F, sPi, sPk, sQj, sQk);
    else
        hipLaunchKernelGGL((freq_contract_kernel<1, 16, CP, CQ>), dim3(gx, (J + 15) / 16, 1), dim3(256), 0, s, p, q, y, I, J, K, F, sPi, sPk, sQj, sQk);
}

// ---------------------------------------------------------------------------------------------------------------
// The forward contraction with the weight spectrum's LAST transform done on the fly (round 6).
// freq_contract_kernel streams W^ (O*C*F*8 bytes: 223 MB for the root grid's 16 x 15 x (15 x 88 x 88) bins) - two thirds of
// all the bytes of the opening conv.  But W^ is the spectrum of a k-tap kernel (k = 7): along y it is a trigonometric
// polynomial of degree (k-1)/2 in the bin index.  With G[o,c,row,ty] the weights transformed along z and x only
// (row = (kz, kx); 17.7 MB), centred taps u = ty - p and E_u(ky) = exp(+2 pi i ky u / SY) = c_u + i s_u:
//     W^[o,c,row,ky] = G_p + sum_{u=1..p} (G_{p+u} + G_{p-u}) c_u + i (G_{p+u} - G_{p-u}) s_u
// i.e. 4 FMAs per u with the sums S_u and differences D_u tabulated: 12 FMAs per (o, c, bin) for k = 7, next to the
// 16 of the 4-sample contraction itself - and 12.6x fewer weight bytes.  T[row][o][c][2 + 4p] = (G_p, S_1, D_1, ...).
// thread = (row of the block, ky); a block stages its rows' table slice (OG output channels) in LDS, lanes of one row
// read the same words (broadcast).
// ---------------------------------------------------------------------------------------------------------------
#ifndef SP3D_FT_OG
#define SP3D_FT_OG 4
#endif
#ifndef SP3D_FT_MINW
#define SP3D_FT_MINW 1
#endif
constexpr int FT_ROWS = 4, FT_P = 3, FT_W = 2 + 4 * FT_P, FT_CMAX = 16, FT_NT = 384;

template <int BB, int OG>
__global__ __launch_bounds__(FT_NT, SP3D_FT_MINW) void freq_contract_ty_kernel(const float2 *__restrict__ X, const float *__restrict__ T,
                                                                 const float *__restrict__ tw, float2 *__restrict__ Y, int B,
                                                                 int C, int O, int rows, int SY)
{
    __shared__ __attribute__((aligned(16))) float tab[FT_ROWS * OG * FT_CMAX * FT_W];
    // workgroup i runs on XCD i % 8: the O / OG output groups of one block of rows get ids 8 apart, i.e. the SAME XCD one after
    // the other, so that X is fetched into that L2 once (2-D grid: every group landed on another XCD and X crossed the fabric
    // O / OG times - 236 MB per launch for 56 MB of X, PMC)
    const int nog = (O + OG - 1) / OG;
    const int gid = blockIdx.x, rb = (gid / (8 * nog)) * 8 + (gid & 7), ogi = (gid >> 3) % nog;
    const int row0 = rb * FT_ROWS, o0 = ogi * OG, b0 = blockIdx.z * BB;
    if (row0 >= rows) return;
    const int t = threadIdx.x;
    const int og = min(OG, O - o0);
    // stage: per row a contiguous run of og * C * FT_W floats of T
    const int per = og * C * FT_W;
    for (int r = 0; r < FT_ROWS; ++r) {
        if (row0 + r >= rows) break;
        const float *src = T + ((size_t)(row0 + r) * O + o0) * C * FT_W;
        for (int i = t; i < per; i += FT_NT) tab[r * OG * FT_CMAX * FT_W + i] = src[i];
    }
    __syncthreads();
    const int r = t / SY, ky = t - r * SY;
    if (r >= FT_ROWS || row0 + r >= rows) return;
    float cs[FT_P], sn[FT_P];
#pragma unroll
    for (int u = 0; u < FT_P; ++u) { cs[u] = tw[(ky * FT_P + u) * 2]; sn[u] = tw[(ky * FT_P + u) * 2 + 1]; }
    const size_t F = (size_t)rows * SY, f = (size_t)(row0 + r) * SY + ky;
    float2 acc[BB][OG];
#pragma unroll
    for (int b = 0; b < BB; ++b)
#pragma unroll
        for (int o = 0; o < OG; ++o) acc[b][o] = make_float2(0.0f, 0.0f);
    const float *trow = tab + r * OG * FT_CMAX * FT_W;
    for (int c = 0; c < C; ++c) {
        float2 x[BB];
#pragma unroll
        for (int b = 0; b < BB; ++b) x[b] = (b0 + b < B) ? X[((size_t)(b0 + b) * C + c) * F + f] : make_float2(0.0f, 0.0f);
#pragma unroll
        for (int o = 0; o < OG; ++o) {
            if (o >= og) break;
            const float *p = trow + (o * C + c) * FT_W;
            float wr = p[0], wi = p[1];
#pragma unroll
            for (int u = 0; u < FT_P; ++u) {          // (S.re, S.im, D.re, D.im)
                wr = fmaf(p[2 + 4 * u], cs[u], wr);
                wr = fmaf(-p[2 + 4 * u + 3], sn[u], wr);
                wi = fmaf(p[2 + 4 * u + 1], cs[u], wi);
                wi = fmaf(p[2 + 4 * u + 2], sn[u], wi);
            }
#pragma unroll
            for (int b = 0; b < BB; ++b) {
                acc[b][o].x = fmaf(x[b].x, wr, acc[b][o].x);
                acc[b][o].x = fmaf(-x[b].y, wi, acc[b][o].x);
                acc[b][o].y = fmaf(x[b].x, wi, acc[b][o].y);
                acc[b][o].y = fmaf(x[b].y, wr, acc[b][o].y);
            }
        }
    }
#pragma unroll
    for (int b = 0; b < BB; ++b)
#pragma unroll
        for (int o = 0; o < OG; ++o)
            if (b0 + b < B && o < og) Y[((size_t)(b0 + b) * O + (o0 + o)) * F + f] = acc[b][o];
}

} // namespace sp3d

using namespace sp3d;

// X (B,C,rows,SY) complex, T (rows,O,C,14) real table of the 7-tap weights transformed along the other two axes, tw (SY,3,2)
// = (cos, sin)(2 pi ky u / SY) for u = 1..3 -> Y (B,O,rows,SY) complex = sum_c X * W^ with W^ rebuilt per bin (see above)
extern "C" int sp3d_freq_contract_ty(const float *X, const float *T, const float *tw, float *Y, int B, int C, int O, int rows,
                                     int SY, void *stream)
{
    if (B <= 0 || C <= 0 || O <= 0 || rows <= 0 || SY <= 0) return SP3D_EINVAL;
    if (!X || !T || !tw || !Y) return SP3D_ENULL;
    if (C > FT_CMAX || FT_ROWS * SY > FT_NT) return SP3D_EUNSUPPORTED;
    const int nrb = (rows + FT_ROWS - 1) / FT_ROWS, nog = (O + SP3D_FT_OG - 1) / SP3D_FT_OG;
    const dim3 grid((unsigned)(((nrb + 7) / 8) * 8 * nog), 1, (unsigned)((B + 3) / 4));
    hipLaunchKernelGGL((freq_contract_ty_kernel<4, SP3D_FT_OG>), grid, dim3(FT_NT), 0, (hipStream_t)stream, reinterpret_cast<const float2 *>(X), T,
                       tw, reinterpret_cast<float2 *>(Y), B, C, O, rows, SY);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SP3D_OK : (int)e;
}

extern "C" int sp3d_freq_contract_ex(const float *P, const float *Q, float *Y, int I, int J, int K, int64_t F,
                                     int64_t sPi, int64_t sPk, int64_t sQj, int64_t sQk, int conj_p, int conj_q,
                                     void *stream)
{
    if (I <= 0 || J <= 0 || K <= 0 || F <= 0) return SP3D_EINVAL;
    if (!P || !Q || !Y) return SP3D_ENULL;
    if ((F + 255) / 256 > 0x7fffffff || I > 65535 * 4 || J > 65535 * 8) return SP3D_ERANGE;
    const float2 *p = reinterpret_cast<const float2 *>(P), *q = reinterpret_cast<const float2 *>(Q);
    float2 *y = reinterpret_cast<float2 *>(Y);
    hipStream_t s = (hipStream_t)stream;
    if (conj_p) { if (conj_q) launch_contract<true, true>(p, q, y, I, J, K, F, sPi, sPk, sQj, sQk, s); else launch_contract<true, false>(p, q, y, I, J, K, F, sPi, sPk, sQj, sQk, s); }
    else { if (conj_q) launch_contract<false, true>(p, q, y, I, J, K, F, sPi, sPk, sQj, sQk, s); else launch_contract<false, false>(p, q, y, I, J, K, F, sPi, sPk, sQj, sQk, s); }
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SP3D_OK : (int)e;
}

extern "C" int sp3d_freq_contract(const float *X, const float *W, float *Y, int B, int C, int O, int64_t F, void *stream)
{
    return sp3d_freq_contract_ex(X, W, Y, B, O, C, F, (int64_t)C * F, F, (int64_t)C * F, F, 0, 0, stream);
}
