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

template <int BB, int OG>
__global__ __launch_bounds__(256) void freq_contract_kernel(const float2 *__restrict__ X, const float2 *__restrict__ W,
                                                           float2 *__restrict__ Y, int B, int C, int O, int64_t F)
{
    const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (f >= F) return;
    const int o0 = blockIdx.y * OG, b0 = blockIdx.z * BB;
    float2 acc[BB][OG];
#pragma unroll
    for (int b = 0; b < BB; ++b)
#pragma unroll
        for (int o = 0; o < OG; ++o) acc[b][o] = make_float2(0.0f, 0.0f);
    for (int c = 0; c < C; ++c) {
        float2 x[BB], w[OG];
#pragma unroll
        for (int b = 0; b < BB; ++b)
            x[b] = (b0 + b < B) ? X[((int64_t)(b0 + b) * C + c) * F + f] : make_float2(0.0f, 0.0f);
#pragma unroll
        for (int o = 0; o < OG; ++o)
            w[o] = (o0 + o < O) ? W[((int64_t)(o0 + o) * C + c) * F + f] : make_float2(0.0f, 0.0f);
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
            if (b0 + b < B && o0 + o < O) Y[((int64_t)(b0 + b) * O + (o0 + o)) * F + f] = acc[b][o];
}

} // namespace sp3d

using namespace sp3d;

extern "C" int sp3d_freq_contract(const float *X, const float *W, float *Y, int B, int C, int O, int64_t F, void *stream)
{
    if (B <= 0 || C <= 0 || O <= 0 || F <= 0) return SP3D_EINVAL;
    if (!X || !W || !Y) return SP3D_ENULL;
    if ((F + 255) / 256 > 0x7fffffff || B > 65535 * 4) return SP3D_ERANGE;
    const float2 *x = reinterpret_cast<const float2 *>(X), *w = reinterpret_cast<const float2 *>(W);
    float2 *y = reinterpret_cast<float2 *>(Y);
    hipStream_t s = (hipStream_t)stream;
    const unsigned gx = (unsigned)((F + 255) / 256);
    if (B >= 3) {
        hipLaunchKernelGGL((freq_contract_kernel<4, 8>), dim3(gx, (O + 7) / 8, (B + 3) / 4), dim3(256), 0, s, x, w, y, B, C, O, F);
    } else if (B == 2) {
        hipLaunchKernelGGL((freq_contract_kernel<2, 8>), dim3(gx, (O + 7) / 8, 1), dim3(256), 0, s, x, w, y, B, C, O, F);
    } else {
        hipLaunchKernelGGL((freq_contract_kernel<1, 16>), dim3(gx, (O + 15) / 16, 1), dim3(256), 0, s, x, w, y, B, C, O, F);
    }
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SP3D_OK : (int)e;
}
