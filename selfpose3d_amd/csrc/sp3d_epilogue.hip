// sp3d_epilogue.hip - fused per-channel shift (+ residual) (+ ReLU) epilogue for the V2V conv stack
// in inference.  The reference runs Conv3d -> BatchNorm3d -> ReLU (and `relu(res + skip)`,
// `upsample + skip`) as separate library kernels (/root/reference/lib/models/v2v_net.py:13-17,
// 26-45, 60-69, 100-108).  With BatchNorm folded into the conv weights on the host
// (selfpose3d_amd/v2v_net.py::_FoldedV2V) what is left per layer is ONE memory-bound pass:
//     mode 0: y = y + shift[c]
//     mode 1: y = relu(y + shift[c])
//     mode 2: y = relu(y + shift[c] + r)          (residual block output)
//     mode 3: y = relu(y + shift[c]) + r          (up-sampling block + skip connection)
// in place, dwordx4 per lane, channels-last (c = element % C) or planar (c = (element / inner) % C).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sp3d.h"

namespace sp3d {

template <int MODE, bool CL>
__global__ __launch_bounds__(256) void channel_shift_act_kernel(float *__restrict__ y, const float *__restrict__ shift,
                                                               const float *__restrict__ res, int64_t n4, int C,
                                                               int64_t inner)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 v = reinterpret_cast<const float4 *>(y)[i];
        float s0, s1, s2, s3;
        if (CL) {   // C % 4 == 0: the four elements are four consecutive channels
            const int c = (int)((i * 4) % C);
            const float4 s = *reinterpret_cast<const float4 *>(shift + c);
            s0 = s.x; s1 = s.y; s2 = s.z; s3 = s.w;
        } else {    // inner % 4 == 0: the four elements share one channel
            const int c = (int)(((i * 4) / inner) % C);
            s0 = s1 = s2 = s3 = shift[c];
        }
        v.x += s0; v.y += s1; v.z += s2; v.w += s3;
        if (MODE == 2) {
            const float4 r = reinterpret_cast<const float4 *>(res)[i];
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        if (MODE >= 1) {
            v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f);
        }
        if (MODE == 3) {
            const float4 r = reinterpret_cast<const float4 *>(res)[i];
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        reinterpret_cast<float4 *>(y)[i] = v;
    }
}

} // namespace sp3d

using namespace sp3d;

extern "C" int sp3d_channel_shift_act(float *y, const float *shift, const float *residual, int mode, int64_t batch,
                                      int C, int64_t inner, int channels_last, void *stream)
{
    if (batch <= 0 || C <= 0 || inner <= 0 || mode < 0 || mode > 3) return SP3D_EINVAL;
    if (!y || !shift || ((mode >= 2) && !residual)) return SP3D_ENULL;
    if (channels_last ? (C & 3) : (inner & 3)) return SP3D_EUNSUPPORTED;
    const int64_t n4 = batch * C * inner / 4;
    const int blocks = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    hipStream_t s = (hipStream_t)stream;
#define SP3D_EPI(M, CL) hipLaunchKernelGGL((channel_shift_act_kernel<M, CL>), dim3(blocks), dim3(256), 0, s, y, shift, residual, n4, C, inner)
    if (channels_last) {
        switch (mode) { case 0: SP3D_EPI(0, true); break; case 1: SP3D_EPI(1, true); break; case 2: SP3D_EPI(2, true); break; default: SP3D_EPI(3, true); }
    } else {
        switch (mode) { case 0: SP3D_EPI(0, false); break; case 1: SP3D_EPI(1, false); break; case 2: SP3D_EPI(2, false); break; default: SP3D_EPI(3, false); }
    }
#undef SP3D_EPI
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SP3D_OK : (int)e;
}

// ------------------------------------------------------------------------------------------
// ConvTranspose3d(kernel 2, stride 2) (lib/models/v2v_net.py:57-69) has no overlapping taps: every output voxel
// (2x+i, 2y+j, 2z+k) depends on input voxel (x,y,z) only, so the layer is ONE GEMM  G[n, (i,j,k,o)] = X[n,:] . W[:, (i,j,k,o)]
// (rocBLAS through torch.matmul on the channels-last view) followed by this scatter, which also applies the layer's
// epilogue: out = relu(G + shift[o]) + skip   (mode 3 of sp3d_channel_shift_act), channels-last in and out.
// ------------------------------------------------------------------------------------------
namespace sp3d {

__global__ __launch_bounds__(256) void upsample2x_scatter_kernel(const float *__restrict__ G, float *__restrict__ out,
                                                                const float *__restrict__ shift,
                                                                const float *__restrict__ skip, int64_t n_in, int X, int Y,
                                                                int Z, int O)
{
    const int64_t total = n_in * 8 * (O / 4);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int q = O / 4;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int o4 = (int)(e % q);
        int64_t r = e / q;
        const int ijk = (int)(r & 7);
        const int64_t n = r >> 3;                              // input voxel (b,x,y,z) flattened
        const int z = (int)(n % Z); int64_t m = n / Z;
        const int y = (int)(m % Y); m /= Y;
        const int x = (int)(m % X);
        const int64_t b = m / X;
        const int xo = 2 * x + (ijk >> 2), yo = 2 * y + ((ijk >> 1) & 1), zo = 2 * z + (ijk & 1);
        const int64_t oi = ((((b * (2 * X) + xo) * (2 * Y) + yo) * (2 * Z) + zo) * O) + 4 * o4;
        const float4 g = *reinterpret_cast<const float4 *>(G + (n * 8 + ijk) * O + 4 * o4);
        const float4 s = *reinterpret_cast<const float4 *>(shift + 4 * o4);
        const float4 k = *reinterpret_cast<const float4 *>(skip + oi);
        float4 v;
        v.x = fmaxf(g.x + s.x, 0.0f) + k.x; v.y = fmaxf(g.y + s.y, 0.0f) + k.y;
        v.z = fmaxf(g.z + s.z, 0.0f) + k.z; v.w = fmaxf(g.w + s.w, 0.0f) + k.w;
        *reinterpret_cast<float4 *>(out + oi) = v;
    }
}


// The LAST up-sampling layer feeds only the 1x1x1 output conv (v2v_net.py:128-133: output_layer(encoder_decoder(x))), so
// its O = 32-channel result never needs to exist: the 8 lanes that hold one output voxel's channels (4 each) reduce
//   head[j] = bias[j] + sum_o Wout[j][o] * (relu(G + shift[o]) + skip[o])
// with three xor-shuffles per head channel and write (batch,2X,2Y,2Z,J) channels-last.  Saves the 65 MB write, the N = 1
// (root net) / N = 15 GEMM that re-reads it, and the bias pass.
__global__ __launch_bounds__(256) void upsample2x_scatter_head_kernel(const float *__restrict__ G, float *__restrict__ head,
                                                                     const float *__restrict__ shift,
                                                                     const float *__restrict__ skip,
                                                                     const float *__restrict__ wout,
                                                                     const float *__restrict__ bout, int64_t n_in, int X,
                                                                     int Y, int Z, int J)
{
    constexpr int O = 32, q = 8;
    const int64_t total = n_in * 8 * q;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int o4 = (int)(e & (q - 1));
        int64_t r = e >> 3;
        const int ijk = (int)(r & 7);
        const int64_t n = r >> 3;
        const int z = (int)(n % Z); int64_t m = n / Z;
        const int y = (int)(m % Y); m /= Y;
        const int x = (int)(m % X);
        const int64_t b = m / X;
        const int xo = 2 * x + (ijk >> 2), yo = 2 * y + ((ijk >> 1) & 1), zo = 2 * z + (ijk & 1);
        const int64_t vox = ((b * (2 * X) + xo) * (2 * Y) + yo) * (2 * Z) + zo;
        const float4 g = *reinterpret_cast<const float4 *>(G + (n * 8 + ijk) * O + 4 * o4);
        const float4 s = *reinterpret_cast<const float4 *>(shift + 4 * o4);
        const float4 k = *reinterpret_cast<const float4 *>(skip + vox * O + 4 * o4);
        float4 v;
        v.x = fmaxf(g.x + s.x, 0.0f) + k.x; v.y = fmaxf(g.y + s.y, 0.0f) + k.y;
        v.z = fmaxf(g.z + s.z, 0.0f) + k.z; v.w = fmaxf(g.w + s.w, 0.0f) + k.w;
        for (int j0 = 0; j0 < J; j0 += q) {
            float mine = 0.0f;                                   // lane o4 keeps head channel j0 + o4
#pragma unroll
            for (int jj = 0; jj < q; ++jj) {
                const int j = j0 + jj;
                if (j >= J) break;                               // uniform
                const float4 w = *reinterpret_cast<const float4 *>(wout + (int64_t)j * O + 4 * o4);
                float p = fmaf(v.w, w.w, fmaf(v.z, w.z, fmaf(v.y, w.y, v.x * w.x)));
                p += __shfl_xor(p, 1);
                p += __shfl_xor(p, 2);
                p += __shfl_xor(p, 4);
                if (jj == o4) mine = p;
            }
            const int j = j0 + o4;
            if (j < J) head[vox * J + j] = mine + bout[j];
        }
    }
}
} // namespace sp3d

extern "C" int sp3d_upsample2x_scatter(const float *G, float *out, const float *shift, const float *skip, int64_t batch, int X,
                                       int Y, int Z, int O, void *stream)
{
    if (batch <= 0 || X <= 0 || Y <= 0 || Z <= 0 || O <= 0) return SP3D_EINVAL;
    if (!G || !out || !shift || !skip) return SP3D_ENULL;
    if (O & 3) return SP3D_EUNSUPPORTED;
    const int64_t n_in = batch * X * Y * Z;
    const int64_t total = n_in * 8 * (O / 4);
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(sp3d::upsample2x_scatter_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, G, out, shift, skip, n_in,
                       X, Y, Z, O);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SP3D_OK : (int)e;
}


extern "C" int sp3d_upsample2x_scatter_head(const float *G, float *head, const float *shift, const float *skip, const float *wout,
                                            const float *bout, int64_t batch, int X, int Y, int Z, int O, int J, void *stream)
{
    if (batch <= 0 || X <= 0 || Y <= 0 || Z <= 0 || O <= 0 || J <= 0) return SP3D_EINVAL;
    if (!G || !head || !shift || !skip || !wout || !bout) return SP3D_ENULL;
    if (O != 32) return SP3D_EUNSUPPORTED;
    const int64_t n_in = batch * X * Y * Z;
    const int64_t total = n_in * 8 * 8;
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(sp3d::upsample2x_scatter_head_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, G, head, shift,
                       skip, wout, bout, n_in, X, Y, Z, J);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SP3D_OK : (int)e;
}


// ------------------------------------------------------------------------------------------
// sp3d_fetch_ring: the per-batch camera table of a HIP-GRAPHED step.
//
// A captured step cannot take a different source address per replay, and an asynchronous host->device copy enqueued
// between two graph launches costs ~90 us of idle GPU (copy-command latency between the two graphs,
// profiles/r02_bench_kernel_stats.md).  Instead the graph's first node is this one-workgroup kernel: the host writes
// the table of step t into slot t % R of a PINNED host ring before launching replay t; the kernel keeps its own step
// counter in device memory, reads slot counter % R straight from host memory (2.5 KB over PCIe, a few us) into the
// fixed device table the other kernels read, and increments the counter.  Replays execute in stream order, so replay t
// always reads slot t % R; the host may run R - 1 launches ahead (it waits on the event of replay t - R before
// rewriting a slot).
// ------------------------------------------------------------------------------------------
namespace sp3d {
__global__ __launch_bounds__(256) void fetch_ring_kernel(const float *__restrict__ ring, float *__restrict__ dst,
                                                        unsigned int *__restrict__ counter, int R, int n)
{
    __shared__ unsigned int slot;
    if (threadIdx.x == 0) slot = *counter % (unsigned int)R;
    __syncthreads();
    // host memory: every load is a PCIe round trip (~2 us), so all of a thread's loads are issued before its first store and
    // they are 16 bytes wide - one round trip for tables up to 4096 floats (the 64-float camera records of B*V <= 64 views;
    // dword loads in a dependent loop cost one round trip per 256 floats: 8.6 us for the bench's 1280 floats)
    const float *src = ring + (size_t)slot * n;
    if ((n & 3) == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0) {
        typedef float f4v __attribute__((ext_vector_type(4)));
        const f4v *s4 = reinterpret_cast<const f4v *>(src);
        f4v *d4 = reinterpret_cast<f4v *>(dst);
        const int n4 = n >> 2;
        for (int base = 0; base < n4; base += 4 * 256) {
            f4v v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = base + k * 256 + (int)threadIdx.x;
                if (i < n4) v[k] = __builtin_nontemporal_load(s4 + i);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = base + k * 256 + (int)threadIdx.x;
                if (i < n4) d4[i] = v[k];
            }
        }
    } else {
        const volatile float *vs = src;
        for (int i = threadIdx.x; i < n; i += 256) dst[i] = vs[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) *counter = *counter + 1u;
}
} // namespace sp3d

extern "C" int sp3d_fetch_ring(const float *ring, float *dst, uint32_t *counter, int R, int n, void *stream)
{
    if (R <= 0 || n <= 0) return SP3D_EINVAL;
    if (!ring || !dst || !counter) return SP3D_ENULL;
    hipLaunchKernelGGL(sp3d::fetch_ring_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ring, dst, counter, R, n);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SP3D_OK : (int)e;
}


// ------------------------------------------------------------------------------------------
// sp3d_maxpool2x_cl: MaxPool3d(2, 2) of the V2V encoder (v2v_net.py:48-54, 76-80) on channels-last activations:
// (B,X,Y,Z,C) -> (B,X/2,Y/2,Z/2,C), one lane = one output voxel x 4 channels, 8 dwordx4 reads, one write.  torch's
// kernel also produces the arg-max indices nobody reads in inference (26 + 11 us per step against 11 + 4 here).
// ------------------------------------------------------------------------------------------
namespace sp3d {
__global__ __launch_bounds__(256) void maxpool2x_cl_kernel(const float *__restrict__ x, float *__restrict__ y, int B, int X,
                                                          int Y, int Z, int C)
{
    const int C4 = C >> 2, Xo = X >> 1, Yo = Y >> 1, Zo = Z >> 1;
    const int64_t n = (int64_t)B * Xo * Yo * Zo * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        int64_t r = i / C4;
        const int zo = (int)(r % Zo); r /= Zo;
        const int yo = (int)(r % Yo); r /= Yo;
        const int xo = (int)(r % Xo);
        const int b = (int)(r / Xo);
        const float *p = x + ((((int64_t)b * X + 2 * xo) * Y + 2 * yo) * Z + 2 * zo) * C + 4 * c4;
        float4 m = *reinterpret_cast<const float4 *>(p);
#pragma unroll
        for (int t = 1; t < 8; ++t) {
            const float4 v = *reinterpret_cast<const float4 *>(p + ((int64_t)((t >> 2) & 1) * Y * Z + (int64_t)((t >> 1) & 1) * Z + (t & 1)) * C);
            // torch.max semantics: NaN propagates
            m.x = (v.x > m.x || v.x != v.x) ? v.x : m.x; m.y = (v.y > m.y || v.y != v.y) ? v.y : m.y;
            m.z = (v.z > m.z || v.z != v.z) ? v.z : m.z; m.w = (v.w > m.w || v.w != v.w) ? v.w : m.w;
        }
        *reinterpret_cast<float4 *>(y + i * 4) = m;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// crop + shift (+ReLU) + planar -> channels-last, the tail of the frequency-domain opening conv: the inverse transform
// leaves (B,C,SX,SY,SZ) planar volumes whose [0:X,0:Y,0:Z] corner is the result.  torch needs a strided copy (crop +
// layout change) and then the in-place epilogue; this is one pass: a work-group takes NY consecutive y columns of one
// (b,x), reads 16-byte pieces of the z runs of every channel, transposes through LDS and writes NY*Z*C contiguous floats.
constexpr int CROP_NY = 4;
__global__ __launch_bounds__(256) void crop_shift_act_cl_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                               const float *__restrict__ shift, int X, int Y, int Z, int C,
                                                               int SX, int SY, int SZ, int relu)
{
    extern __shared__ float lds[];                      // [CROP_NY][Z][C + 4]
    const int P = C + 4, Z4 = Z >> 2, C4 = C >> 2;
    const int ygroups = (Y + CROP_NY - 1) / CROP_NY;
    int r = blockIdx.x;
    const int yg = r % ygroups; r /= ygroups;
    const int x = r % X;
    const int b = r / X;
    const int y0 = yg * CROP_NY;
    const int ny = min(CROP_NY, Y - y0);
    const int per_col = C * Z4;
    for (int e = threadIdx.x; e < ny * per_col; e += 256) {
        const int col = e / per_col, q = e - col * per_col;
        const int c = q / Z4, z4 = q - c * Z4;
        const float4 v = *reinterpret_cast<const float4 *>(
            src + ((((int64_t)b * C + c) * SX + x) * SY + (y0 + col)) * SZ + 4 * z4);
        float *o = lds + ((col * Z) + 4 * z4) * P + c;
        o[0] = v.x; o[P] = v.y; o[2 * P] = v.z; o[3 * P] = v.w;
    }
    __syncthreads();
    float *out = dst + (((int64_t)b * X + x) * Y + y0) * Z * C;
    for (int e = threadIdx.x; e < ny * Z * C4; e += 256) {
        const int vox = e / C4, q = e - vox * C4;
        float4 v = *reinterpret_cast<const float4 *>(lds + vox * P + 4 * q);
        const float4 s = *reinterpret_cast<const float4 *>(shift + 4 * q);
        v.x += s.x; v.y += s.y; v.z += s.z; v.w += s.w;
        if (relu) {                                      // NaN propagates like torch.relu
            v.x = v.x < 0.f ? 0.f : v.x; v.y = v.y < 0.f ? 0.f : v.y;
            v.z = v.z < 0.f ? 0.f : v.z; v.w = v.w < 0.f ? 0.f : v.w;
        }
        *reinterpret_cast<float4 *>(out + (int64_t)e * 4) = v;
    }
}
} // namespace sp3d

extern "C" int sp3d_maxpool2x_cl(const float *x, float *y, int B, int X, int Y, int Z, int C, void *stream)
{
    if (B <= 0 || X <= 1 || Y <= 1 || Z <= 1 || C <= 0) return SP3D_EINVAL;
    if (!x || !y) return SP3D_ENULL;
    if ((C & 3) || (X & 1) || (Y & 1) || (Z & 1)) return SP3D_EUNSUPPORTED;
    const int64_t n = (int64_t)B * (X / 2) * (Y / 2) * (Z / 2) * (C / 4);
    const int64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(sp3d::maxpool2x_cl_kernel, dim3((unsigned)(blocks > 65535 * 16 ? 65535 * 16 : blocks)), dim3(256), 0,
                       (hipStream_t)stream, x, y, B, X, Y, Z, C);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SP3D_OK : (int)e;
}


extern "C" int sp3d_crop_shift_act_cl(const float *src, float *dst, const float *shift, int B, int C, int X, int Y, int Z,
                                      int SX, int SY, int SZ, int relu, void *stream)
{
    if (B <= 0 || C <= 0 || X <= 0 || Y <= 0 || Z <= 0 || SX < X || SY < Y || SZ < Z) return SP3D_EINVAL;
    if (!src || !dst || !shift) return SP3D_ENULL;
    if ((C & 3) || (Z & 3) || (SZ & 3) || (reinterpret_cast<uintptr_t>(src) & 15) || (reinterpret_cast<uintptr_t>(dst) & 15))
        return SP3D_EUNSUPPORTED;
    const size_t lds = (size_t)sp3d::CROP_NY * Z * (C + 4) * sizeof(float);
    if (lds > 64 * 1024) return SP3D_EUNSUPPORTED;
    const int64_t blocks = (int64_t)B * X * ((Y + sp3d::CROP_NY - 1) / sp3d::CROP_NY);
    if (blocks > 0x7fffffff) return SP3D_EINVAL;
    hipLaunchKernelGGL(sp3d::crop_shift_act_cl_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, src, dst,
                       shift, X, Y, Z, C, SX, SY, SZ, relu);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SP3D_OK : (int)e;
}
