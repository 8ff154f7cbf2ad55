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

} // namespace sp3d

using namespace sp3d;

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
