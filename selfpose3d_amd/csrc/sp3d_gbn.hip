// sp3d_gbn.hip - GROUPED training-mode batch normalisation on channels-last tensors (round 5).
//
// The reference trains its pose net one candidate slot at a time (/root/reference/lib/models/multi_person_posenet.py:84-88,
// multi_person_posenet_ssv.py:354-383): up to MAX_PEOPLE_NUM calls of V2VNet, each a BatchNorm3d batch of its own
// (/root/reference/lib/models/v2v_net.py:14,28,31,38,64: statistics over the cubes of ONE call), and its 2-D backbone one
// camera at a time (multi_person_posenet.py:44-47; pose_resnet.py BatchNorm2d).  Running all slots (all cameras) as ONE
// batch is only the same function if every BatchNorm layer keeps the statistics of each slot's (camera's) samples apart.
// These kernels do that in the layout the convolutions run fastest in: x is (N, S, C) with C contiguous
// (torch.channels_last / channels_last_3d), group_of[n] in [0, G) names the group of sample n (any assignment: ragged
// contiguous runs for slots, n % V for cameras), and per (group, channel)
//     mean = E[x], var = E[x^2] - mean^2 over the group's samples x S,   y = (x - mean) / sqrt(var + eps) * w + b [ReLU]
// with the running statistics receiving the groups' momentum updates one after the other in group order - what the loop does.
//
// Four streaming passes, all HBM-bound (16-byte accesses, a thread keeps its channel columns for all rows it visits):
//   gbn_stats_kernel    x       -> per-(replica, group, channel) sum and sum of squares, accumulated in FLOAT64 per thread,
//                                  merged with float64 memory atomics into R = 16 replicas (contention spread)
//   gbn_finalize_kernel         -> mean / invstd / scale / shift per (group, channel); running statistics; 16 lanes per
//                                  channel reduce the replicas and ZERO them again (the workspace is zero-filled once, by
//                                  the caller; every call leaves it zero-filled)
//   gbn_apply_kernel    x       -> y = fma(x, scale, shift) [max 0]
//   gbn_bwd_stats_kernel x, dy  -> sum dy, sum dy * xhat  (dy masked by the recomputed ReLU when fused)
//   gbn_bwd_finalize_kernel     -> k1, k2, k3 per (group, channel); grad_weight, grad_bias
//   gbn_bwd_apply_kernel x, dy  -> dx = k1 * dy + k2 * x + k3
// Algorithmic bytes per element: forward 3 x sizeof(T) (two reads, one write), backward 5 x sizeof(T).
// T = float (product) or double (the float64 equivalence tests against the per-slot loop).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sp3d.h"

namespace sp3d {

constexpr int GBN_TPB = 256;
constexpr int GBN_CPT = 2;          // channel columns per thread: C <= 2 * 256 vectors of 16 bytes (2048 floats)

template <typename T> struct GbnVec;
template <> struct GbnVec<float> {
    static constexpr int W = 4;
    typedef float4 V;
    static __device__ __forceinline__ void unpack(const V &v, float (&a)[4]) { a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w; }
    static __device__ __forceinline__ V pack(const float (&a)[4]) { return make_float4(a[0], a[1], a[2], a[3]); }
};
template <> struct GbnVec<double> {
    static constexpr int W = 2;
    typedef double2 V;
    static __device__ __forceinline__ void unpack(const V &v, double (&a)[2]) { a[0] = v.x; a[1] = v.y; }
    static __device__ __forceinline__ V pack(const double (&a)[2]) { return make_double2(a[0], a[1]); }
};

struct GbnMap {            // thread -> (channel column, row phase) of a block that walks rows of CV 16-byte columns
    int col0, ro, RP, ncol;
    bool active;
    __device__ __forceinline__ GbnMap(int CV)
    {
        const int cpp = CV < GBN_TPB ? CV : GBN_TPB;        // columns per pass
        RP = GBN_TPB / cpp;
        col0 = threadIdx.x % cpp;
        ro = threadIdx.x / cpp;
        active = ro < RP;
        ncol = (CV + GBN_TPB - 1) / GBN_TPB;
    }
};

// sum over the RP row phases of one block, then one float64 atomic per (channel, quantity)
template <int W>
__device__ __forceinline__ void gbn_block_merge(const GbnMap &m, int CV, double (&a)[GBN_CPT][W], double (&b)[GBN_CPT][W],
                                                double *__restrict__ acc_gc /* (C, 2) of this replica and group */)
{
    __shared__ double red[GBN_TPB * 4 * 2];
    for (int k = 0; k < m.ncol; ++k) {
        const int col = m.col0 + k * GBN_TPB;
        __syncthreads();
        if (m.active)
            for (int e = 0; e < W; ++e) {
                red[(threadIdx.x * W + e) * 2 + 0] = a[k][e];
                red[(threadIdx.x * W + e) * 2 + 1] = b[k][e];
            }
        __syncthreads();
        if (m.active && m.ro == 0 && col < CV) {
            const int cpp = CV < GBN_TPB ? CV : GBN_TPB;
            for (int e = 0; e < W; ++e) {
                double s = 0.0, q = 0.0;
                for (int r = 0; r < m.RP; ++r) {
                    s += red[((r * cpp + m.col0) * W + e) * 2 + 0];
                    q += red[((r * cpp + m.col0) * W + e) * 2 + 1];
                }
                atomicAdd(acc_gc + (size_t)(col * W + e) * 2 + 0, s);
                atomicAdd(acc_gc + (size_t)(col * W + e) * 2 + 1, q);
            }
        }
    }
}

// grid (chunks, N): block (chunk, n) visits rows [chunk * rows, ...) of sample n
template <typename T>
__global__ __launch_bounds__(GBN_TPB) void gbn_stats_kernel(const T *__restrict__ x, const int32_t *__restrict__ group_of,
                                                            int64_t S, int C, int G, int R, int64_t rows,
                                                            double *__restrict__ acc)
{
    constexpr int W = GbnVec<T>::W;
    typedef typename GbnVec<T>::V V;
    const int CV = C / W;
    const GbnMap m(CV);
    const int n = blockIdx.y, g = group_of[n];
    const int64_t r0 = (int64_t)blockIdx.x * rows, r1 = (r0 + rows < S) ? r0 + rows : S;
    double s[GBN_CPT][W], q[GBN_CPT][W];
    for (int k = 0; k < GBN_CPT; ++k)
        for (int e = 0; e < W; ++e) s[k][e] = q[k][e] = 0.0;
    if (m.active) {
        const T *base = x + (size_t)n * S * C;
        for (int k = 0; k < m.ncol; ++k) {
            const int col = m.col0 + k * GBN_TPB;
            if (col >= CV) break;
            const V *p = reinterpret_cast<const V *>(base) + col;
#pragma unroll 4
            for (int64_t r = r0 + m.ro; r < r1; r += m.RP) {
                T a[W];
                GbnVec<T>::unpack(p[(size_t)r * CV], a);
                for (int e = 0; e < W; ++e) {
                    const double v = (double)a[e];
                    s[k][e] += v;
                    q[k][e] = fma(v, v, q[k][e]);
                }
            }
        }
    }
    const int rep = (int)((blockIdx.x + (unsigned)n * 7u) % (unsigned)R);
    gbn_block_merge<W>(m, CV, s, q, acc + ((size_t)rep * G + g) * C * 2);
}

// Sum of the R = 16 replicas of (group g, channel c): sixteen lanes per channel, one replica each, xor-shuffle reduction.
// Every lane ZEROES the accumulator word it has read: the workspace is left zero-filled for the next call (the caller
// fills it once).  Returns the totals in all sixteen lanes.
constexpr int GBN_GB = 8;           // groups whose replica words are fetched together (independent loads, one latency)
__device__ __forceinline__ void gbn_replica_fetch(double *__restrict__ acc, int G, int C, int g0, int c, int r, bool ok,
                                                  double2 (&v)[GBN_GB])
{
#pragma unroll
    for (int j = 0; j < GBN_GB; ++j) {
        v[j] = make_double2(0.0, 0.0);
        if (ok && g0 + j < G) v[j] = *reinterpret_cast<const double2 *>(acc + (((size_t)r * G + g0 + j) * C + c) * 2);
    }
#pragma unroll
    for (int j = 0; j < GBN_GB; ++j)
        if (ok && g0 + j < G) *reinterpret_cast<double2 *>(acc + (((size_t)r * G + g0 + j) * C + c) * 2) = make_double2(0.0, 0.0);
}
__device__ __forceinline__ void gbn_replica_sum(const double2 v, double &s, double &q)
{
    s = v.x; q = v.y;
#pragma unroll
    for (int m = 1; m < SP3D_GBN_REPLICAS; m <<= 1) {
        s += __shfl_xor(s, m);
        q += __shfl_xor(q, m);
    }
}

// 256 threads = 16 channels x 16 replica lanes; groups in order (the running statistics see the loop's sequence of updates)
template <typename T>
__global__ __launch_bounds__(256) void gbn_finalize_kernel(double *__restrict__ acc, const int32_t *__restrict__ group_samples,
                                                           int64_t S, int C, int G, int G_update,
                                                           const T *__restrict__ weight, const T *__restrict__ bias,
                                                           T *__restrict__ running_mean, T *__restrict__ running_var,
                                                           double eps, double momentum, T *__restrict__ mean,
                                                           T *__restrict__ invstd, T *__restrict__ scale, T *__restrict__ shift)
{
    static_assert(SP3D_GBN_REPLICAS == 16, "sixteen replica lanes per channel");
    const int r = threadIdx.x & 15, c = blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool ok = c < C;
    const double w = (ok && weight) ? (double)weight[c] : 1.0, b = (ok && bias) ? (double)bias[c] : 0.0;
    double rm = (ok && running_mean) ? (double)running_mean[c] : 0.0, rv = (ok && running_var) ? (double)running_var[c] : 0.0;
    for (int g0 = 0; g0 < G; g0 += GBN_GB) {
      double2 v[GBN_GB];
      gbn_replica_fetch(acc, G, C, g0, c, r, ok, v);
#pragma unroll
      for (int j = 0; j < GBN_GB; ++j) {
        const int g = g0 + j;
        if (g >= G) break;
        double s, q;
        gbn_replica_sum(v[j], s, q);
        const double cnt = (double)group_samples[g] * (double)S;
        double mu = 0.0, var = 0.0;
        if (cnt > 0.0) {
            mu = s / cnt;
            var = q / cnt - mu * mu;
            if (var < 0.0) var = 0.0;
        }
        const double is = 1.0 / sqrt(var + eps);
        // the T-rounded values are THE statistics: forward and backward both use them
        const T mu_t = (T)mu, is_t = (T)is;
        if (ok && r == 0) {
            mean[(size_t)g * C + c] = mu_t;
            invstd[(size_t)g * C + c] = is_t;
            const double sc = w * (double)is_t;
            scale[(size_t)g * C + c] = (T)sc;
            shift[(size_t)g * C + c] = (T)(b - (double)mu_t * sc);
        }
        if (g < G_update && cnt > 1.0) {     // torch: running_var takes the UNBIASED batch variance
            rm = (1.0 - momentum) * rm + momentum * mu;
            rv = (1.0 - momentum) * rv + momentum * var * (cnt / (cnt - 1.0));
        }
      }
    }
    if (ok && r == 0) {
        if (running_mean) running_mean[c] = (T)rm;
        if (running_var) running_var[c] = (T)rv;
    }
}

// grid (blocks, N): y = fma(x, scale[g], shift[g]) [max 0]; the (scale, shift) rows of the sample's group sit in LDS
// MODE 0: y = bn(x); 1: y = max(0, bn(x)); 2: y = max(0, bn(x) + residual)  (the tail of a residual block, v2v_net.py:42-45,
// pose_resnet.py BasicBlock / Bottleneck: BatchNorm, add and ReLU in one pass)
template <typename T, int MODE>
__global__ __launch_bounds__(GBN_TPB) void gbn_apply_kernel(const T *__restrict__ x, const int32_t *__restrict__ group_of,
                                                            const T *__restrict__ scale, const T *__restrict__ shift,
                                                            const T *__restrict__ res, T *__restrict__ y, int64_t S, int C)
{
    constexpr bool RELU = MODE >= 1;
    constexpr int W = GbnVec<T>::W;
    typedef typename GbnVec<T>::V V;
    extern __shared__ __attribute__((aligned(16))) unsigned char gbn_smem[];
    T *tab = reinterpret_cast<T *>(gbn_smem);               // [2][C]
    const int n = blockIdx.y, g = group_of[n], CV = C / W;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        tab[c] = scale[(size_t)g * C + c];
        tab[C + c] = shift[(size_t)g * C + c];
    }
    __syncthreads();
    const int64_t nv = S * CV, stride = (int64_t)gridDim.x * blockDim.x;
    const V *xp = reinterpret_cast<const V *>(x + (size_t)n * S * C);
    const V *rp = MODE == 2 ? reinterpret_cast<const V *>(res + (size_t)n * S * C) : nullptr;
    V *yp = reinterpret_cast<V *>(y + (size_t)n * S * C);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
        const int col = (int)(i % CV);
        T a[W], sc[W], sh[W], rr[W];
        GbnVec<T>::unpack(xp[i], a);
        if (MODE == 2) GbnVec<T>::unpack(rp[i], rr);
        GbnVec<T>::unpack(*reinterpret_cast<const V *>(tab + col * W), sc);
        GbnVec<T>::unpack(*reinterpret_cast<const V *>(tab + C + col * W), sh);
        for (int e = 0; e < W; ++e) {
            a[e] = fma(a[e], sc[e], sh[e]);
            if (MODE == 2) a[e] = a[e] + rr[e];
            if (RELU) a[e] = a[e] > (T)0 ? a[e] : (T)0;
        }
        yp[i] = GbnVec<T>::pack(a);
    }
}

// sum dy and sum dy * xhat per (replica, group, channel); with RELU dy counts only where fma(x, scale, shift) > 0
// (MODE 2: dy counts where the block's OUTPUT y is positive - y is what the next layer kept anyway)
template <typename T, int MODE>
__global__ __launch_bounds__(GBN_TPB) void gbn_bwd_stats_kernel(const T *__restrict__ x, const T *__restrict__ dy,
                                                                const T *__restrict__ yout,
                                                                const int32_t *__restrict__ group_of,
                                                                const T *__restrict__ mean, const T *__restrict__ invstd,
                                                                const T *__restrict__ scale, const T *__restrict__ shift,
                                                                int64_t S, int C, int G, int R, int64_t rows,
                                                                double *__restrict__ acc)
{
    constexpr bool RELU = MODE == 1;
    constexpr int W = GbnVec<T>::W;
    typedef typename GbnVec<T>::V V;
    const int CV = C / W;
    const GbnMap m(CV);
    const int n = blockIdx.y, g = group_of[n];
    const int64_t r0 = (int64_t)blockIdx.x * rows, r1 = (r0 + rows < S) ? r0 + rows : S;
    double s[GBN_CPT][W], q[GBN_CPT][W];
    for (int k = 0; k < GBN_CPT; ++k)
        for (int e = 0; e < W; ++e) s[k][e] = q[k][e] = 0.0;
    if (m.active) {
        const size_t off = (size_t)n * S * C;
        for (int k = 0; k < m.ncol; ++k) {
            const int col = m.col0 + k * GBN_TPB;
            if (col >= CV) break;
            T mu[W], is[W], sc[W], sh[W];
            GbnVec<T>::unpack(*reinterpret_cast<const V *>(mean + (size_t)g * C + col * W), mu);
            GbnVec<T>::unpack(*reinterpret_cast<const V *>(invstd + (size_t)g * C + col * W), is);
            if (RELU) {
                GbnVec<T>::unpack(*reinterpret_cast<const V *>(scale + (size_t)g * C + col * W), sc);
                GbnVec<T>::unpack(*reinterpret_cast<const V *>(shift + (size_t)g * C + col * W), sh);
            }
            const V *px = reinterpret_cast<const V *>(x + off) + col;
            const V *pd = reinterpret_cast<const V *>(dy + off) + col;
            const V *py = MODE == 2 ? reinterpret_cast<const V *>(yout + off) + col : nullptr;
#pragma unroll 2
            for (int64_t r = r0 + m.ro; r < r1; r += m.RP) {
                T a[W], d[W], yo[W];
                GbnVec<T>::unpack(px[(size_t)r * CV], a);
                GbnVec<T>::unpack(pd[(size_t)r * CV], d);
                if (MODE == 2) GbnVec<T>::unpack(py[(size_t)r * CV], yo);
                for (int e = 0; e < W; ++e) {
                    if (RELU && !(fma(a[e], sc[e], sh[e]) > (T)0)) continue;
                    if (MODE == 2 && !(yo[e] > (T)0)) continue;
                    const double dv = (double)d[e];
                    s[k][e] += dv;
                    q[k][e] = fma(dv, ((double)a[e] - (double)mu[e]) * (double)is[e], q[k][e]);
                }
            }
        }
    }
    const int rep = (int)((blockIdx.x + (unsigned)n * 7u) % (unsigned)R);
    gbn_block_merge<W>(m, CV, s, q, acc + ((size_t)rep * G + g) * C * 2);
}

// dx = w * invstd * (dy - mean(dy) - xhat * mean(dy * xhat)) = k1 * dy + k2 * x + k3   (16 channels x 16 replica lanes)
template <typename T>
__global__ __launch_bounds__(256) void gbn_bwd_finalize_kernel(double *__restrict__ acc, const int32_t *__restrict__ group_samples,
                                                               int64_t S, int C, int G, const T *__restrict__ weight,
                                                               const T *__restrict__ mean, const T *__restrict__ invstd,
                                                               T *__restrict__ k1, T *__restrict__ k2, T *__restrict__ k3,
                                                               T *__restrict__ grad_weight, T *__restrict__ grad_bias)
{
    const int r = threadIdx.x & 15, c = blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool ok = c < C;
    const double w = (ok && weight) ? (double)weight[c] : 1.0;
    double gw = 0.0, gb = 0.0;
    for (int g0 = 0; g0 < G; g0 += GBN_GB) {
      double2 v[GBN_GB];
      gbn_replica_fetch(acc, G, C, g0, c, r, ok, v);
#pragma unroll
      for (int j = 0; j < GBN_GB; ++j) {
        const int g = g0 + j;
        if (g >= G) break;
        double s, q;
        gbn_replica_sum(v[j], s, q);
        gb += s;
        gw += q;
        if (ok && r == 0) {
            const double cnt = (double)group_samples[g] * (double)S;
            const double is = (double)invstd[(size_t)g * C + c], mu = (double)mean[(size_t)g * C + c];
            const double a = cnt > 0.0 ? s / cnt : 0.0, b = cnt > 0.0 ? q / cnt : 0.0;
            const double c1 = w * is, c2 = -c1 * b * is;
            k1[(size_t)g * C + c] = (T)c1;
            k2[(size_t)g * C + c] = (T)c2;
            k3[(size_t)g * C + c] = (T)(-c1 * a - c2 * mu);
        }
      }
    }
    if (ok && r == 0) {
        if (grad_weight) grad_weight[c] = (T)gw;
        if (grad_bias) grad_bias[c] = (T)gb;
    }
}

template <typename T, int MODE>
__global__ __launch_bounds__(GBN_TPB) void gbn_bwd_apply_kernel(const T *__restrict__ x, const T *__restrict__ dy,
                                                                const T *__restrict__ yout,
                                                                const int32_t *__restrict__ group_of,
                                                                const T *__restrict__ k1, const T *__restrict__ k2,
                                                                const T *__restrict__ k3, const T *__restrict__ scale,
                                                                const T *__restrict__ shift, T *__restrict__ dx,
                                                                T *__restrict__ dres, int64_t S, int C)
{
    constexpr bool RELU = MODE == 1;
    constexpr int W = GbnVec<T>::W;
    typedef typename GbnVec<T>::V V;
    extern __shared__ __attribute__((aligned(16))) unsigned char gbn_smem[];
    T *tab = reinterpret_cast<T *>(gbn_smem);               // [5][C]: k1 k2 k3 scale shift
    const int n = blockIdx.y, g = group_of[n], CV = C / W;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        tab[c] = k1[(size_t)g * C + c];
        tab[C + c] = k2[(size_t)g * C + c];
        tab[2 * C + c] = k3[(size_t)g * C + c];
        if (RELU) {
            tab[3 * C + c] = scale[(size_t)g * C + c];
            tab[4 * C + c] = shift[(size_t)g * C + c];
        }
    }
    __syncthreads();
    const int64_t nv = S * CV, stride = (int64_t)gridDim.x * blockDim.x;
    const size_t off = (size_t)n * S * C;
    const V *xp = reinterpret_cast<const V *>(x + off);
    const V *dp = reinterpret_cast<const V *>(dy + off);
    const V *yp = MODE == 2 ? reinterpret_cast<const V *>(yout + off) : nullptr;
    V *op = reinterpret_cast<V *>(dx + off);
    V *rp = MODE == 2 ? reinterpret_cast<V *>(dres + off) : nullptr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
        const int col = (int)(i % CV);
        T a[W], d[W], c1[W], c2[W], c3[W], o[W];
        GbnVec<T>::unpack(xp[i], a);
        GbnVec<T>::unpack(dp[i], d);
        if (MODE == 2) {                    // the gradient that passes the block's ReLU: also the residual branch's gradient
            T yo[W];
            GbnVec<T>::unpack(yp[i], yo);
            for (int e = 0; e < W; ++e)
                if (!(yo[e] > (T)0)) d[e] = (T)0;
            rp[i] = GbnVec<T>::pack(d);
        }
        GbnVec<T>::unpack(*reinterpret_cast<const V *>(tab + col * W), c1);
        GbnVec<T>::unpack(*reinterpret_cast<const V *>(tab + C + col * W), c2);
        GbnVec<T>::unpack(*reinterpret_cast<const V *>(tab + 2 * C + col * W), c3);
        if (RELU) {
            T sc[W], sh[W];
            GbnVec<T>::unpack(*reinterpret_cast<const V *>(tab + 3 * C + col * W), sc);
            GbnVec<T>::unpack(*reinterpret_cast<const V *>(tab + 4 * C + col * W), sh);
            for (int e = 0; e < W; ++e)
                if (!(fma(a[e], sc[e], sh[e]) > (T)0)) d[e] = (T)0;
        }
        for (int e = 0; e < W; ++e) o[e] = fma(c1[e], d[e], fma(c2[e], a[e], c3[e]));
        op[i] = GbnVec<T>::pack(o);
    }
}

static int gbn_check(int dtype, int N, int64_t S, int C, int G, int R)
{
    if (N <= 0 || S <= 0 || C <= 0 || G <= 0 || R <= 0 || (dtype != SP3D_GBN_F32 && dtype != SP3D_GBN_F64)) return SP3D_EINVAL;
    const int W = dtype == SP3D_GBN_F32 ? 4 : 2;
    if (C % W || C / W > GBN_CPT * GBN_TPB) return SP3D_EUNSUPPORTED;
    if (N > 65535 || S > ((int64_t)1 << 40)) return SP3D_ERANGE;
    return 0;
}

// rows per block: ~2048 blocks in flight over the whole tensor, whole multiples of the block's row phases
static int64_t gbn_rows(int N, int64_t S, int CV)
{
    const int cpp = CV < GBN_TPB ? CV : GBN_TPB;
    const int64_t RP = GBN_TPB / cpp;
    int64_t rows = (N * S + 2047) / 2048;
    rows = (rows + RP * 8 - 1) / (RP * 8) * (RP * 8);
    return rows < RP * 8 ? RP * 8 : rows;
}

static int gbn_status()
{
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

} // namespace sp3d

using namespace sp3d;

extern "C" int64_t sp3d_gbn_workspace_bytes(int G, int C)
{
    if (G <= 0 || C <= 0) return SP3D_EINVAL;
    return (int64_t)SP3D_GBN_REPLICAS * G * C * 2 * (int64_t)sizeof(double);
}

extern "C" int sp3d_gbn_forward(const void *x, void *y, int dtype, const int32_t *group_of, const int32_t *group_samples,
                                int N, int64_t S, int C, int G, int G_update, const void *weight, const void *bias,
                                void *running_mean, void *running_var, double eps, double momentum, int mode,
                                const void *residual, void *mean, void *invstd, void *scale, void *shift, double *workspace,
                                void *stream)
{
    if (mode < 0 || mode > 2) return SP3D_EINVAL;
    if (mode == 2 && !residual) return SP3D_ENULL;
    const int R = SP3D_GBN_REPLICAS;
    int rc = gbn_check(dtype, N, S, C, G, R);
    if (rc) return rc;
    if (G_update < 0 || G_update > G) return SP3D_EINVAL;
    if (!x || !y || !group_of || !group_samples || !mean || !invstd || !scale || !shift || !workspace) return SP3D_ENULL;
    hipStream_t s = (hipStream_t)stream;
    const int W = dtype == SP3D_GBN_F32 ? 4 : 2, CV = C / W;
    const int64_t rows = gbn_rows(N, S, CV), chunks = (S + rows - 1) / rows;
    const int64_t nv = S * CV;
    int64_t ab = (nv + GBN_TPB * 4 - 1) / (GBN_TPB * 4);
    const int64_t cap = (4096 + N - 1) / N;
    if (ab > cap) ab = cap;
    if (ab < 1) ab = 1;
    if (chunks > 0x7fffffff) return SP3D_ERANGE;
    const dim3 gs((unsigned)chunks, (unsigned)N), ga((unsigned)ab, (unsigned)N), gf((unsigned)((C + 15) / 16));
    if (dtype == SP3D_GBN_F32) {
        typedef float T;
        const size_t lds = 2 * (size_t)C * sizeof(T);
        hipLaunchKernelGGL(gbn_stats_kernel<T>, gs, dim3(GBN_TPB), 0, s, (const T *)x, group_of, S, C, G, R, rows, workspace);
        hipLaunchKernelGGL(gbn_finalize_kernel<T>, gf, dim3(256), 0, s, workspace, group_samples, S, C, G, G_update,
                           (const T *)weight, (const T *)bias, (T *)running_mean, (T *)running_var, eps, momentum, (T *)mean,
                           (T *)invstd, (T *)scale, (T *)shift);
        if (mode == 2) hipLaunchKernelGGL((gbn_apply_kernel<T, 2>), ga, dim3(GBN_TPB), lds, s, (const T *)x, group_of, (const T *)scale, (const T *)shift, (const T *)residual, (T *)y, S, C);
        else if (mode == 1) hipLaunchKernelGGL((gbn_apply_kernel<T, 1>), ga, dim3(GBN_TPB), lds, s, (const T *)x, group_of, (const T *)scale, (const T *)shift, (const T *)nullptr, (T *)y, S, C);
        else hipLaunchKernelGGL((gbn_apply_kernel<T, 0>), ga, dim3(GBN_TPB), lds, s, (const T *)x, group_of, (const T *)scale, (const T *)shift, (const T *)nullptr, (T *)y, S, C);
    } else {
        typedef double T;
        const size_t lds = 2 * (size_t)C * sizeof(T);
        hipLaunchKernelGGL(gbn_stats_kernel<T>, gs, dim3(GBN_TPB), 0, s, (const T *)x, group_of, S, C, G, R, rows, workspace);
        hipLaunchKernelGGL(gbn_finalize_kernel<T>, gf, dim3(256), 0, s, workspace, group_samples, S, C, G, G_update,
                           (const T *)weight, (const T *)bias, (T *)running_mean, (T *)running_var, eps, momentum, (T *)mean,
                           (T *)invstd, (T *)scale, (T *)shift);
        if (mode == 2) hipLaunchKernelGGL((gbn_apply_kernel<T, 2>), ga, dim3(GBN_TPB), lds, s, (const T *)x, group_of, (const T *)scale, (const T *)shift, (const T *)residual, (T *)y, S, C);
        else if (mode == 1) hipLaunchKernelGGL((gbn_apply_kernel<T, 1>), ga, dim3(GBN_TPB), lds, s, (const T *)x, group_of, (const T *)scale, (const T *)shift, (const T *)nullptr, (T *)y, S, C);
        else hipLaunchKernelGGL((gbn_apply_kernel<T, 0>), ga, dim3(GBN_TPB), lds, s, (const T *)x, group_of, (const T *)scale, (const T *)shift, (const T *)nullptr, (T *)y, S, C);
    }
    return gbn_status();
}

extern "C" int sp3d_gbn_backward(const void *x, const void *dy, void *dx, int dtype, const int32_t *group_of,
                                 const int32_t *group_samples, int N, int64_t S, int C, int G, const void *weight,
                                 const void *mean, const void *invstd, const void *scale, const void *shift, int mode,
                                 const void *y, void *grad_residual, void *grad_weight, void *grad_bias, void *k123,
                                 double *workspace, void *stream)
{
    if (mode < 0 || mode > 2) return SP3D_EINVAL;
    if (mode == 2 && (!y || !grad_residual)) return SP3D_ENULL;
    const int relu = mode == 1;
    const int R = SP3D_GBN_REPLICAS;
    int rc = gbn_check(dtype, N, S, C, G, R);
    if (rc) return rc;
    if (!x || !dy || !dx || !group_of || !group_samples || !mean || !invstd || !k123 || !workspace) return SP3D_ENULL;
    if (relu && (!scale || !shift)) return SP3D_ENULL;
    hipStream_t s = (hipStream_t)stream;
    const int W = dtype == SP3D_GBN_F32 ? 4 : 2, CV = C / W;
    const int64_t rows = gbn_rows(N, S, CV), chunks = (S + rows - 1) / rows;
    const int64_t nv = S * CV;
    int64_t ab = (nv + GBN_TPB * 4 - 1) / (GBN_TPB * 4);
    const int64_t cap = (4096 + N - 1) / N;
    if (ab > cap) ab = cap;
    if (ab < 1) ab = 1;
    if (chunks > 0x7fffffff) return SP3D_ERANGE;
    const dim3 gs((unsigned)chunks, (unsigned)N), ga((unsigned)ab, (unsigned)N), gf((unsigned)((C + 15) / 16));
    const size_t gc = (size_t)G * C;
#define SP3D_GBN_BWD(T_)                                                                                                          \
    {                                                                                                                             \
        typedef T_ T;                                                                                                             \
        T *k1 = (T *)k123, *k2 = k1 + gc, *k3 = k2 + gc;                                                                          \
        const size_t lds = 5 * (size_t)C * sizeof(T);                                                                             \
        if (mode == 2) hipLaunchKernelGGL((gbn_bwd_stats_kernel<T, 2>), gs, dim3(GBN_TPB), 0, s, (const T *)x, (const T *)dy, (const T *)y, group_of, (const T *)mean, (const T *)invstd, (const T *)scale, (const T *)shift, S, C, G, R, rows, workspace); \
        else if (relu) hipLaunchKernelGGL((gbn_bwd_stats_kernel<T, 1>), gs, dim3(GBN_TPB), 0, s, (const T *)x, (const T *)dy, (const T *)nullptr, group_of, (const T *)mean, (const T *)invstd, (const T *)scale, (const T *)shift, S, C, G, R, rows, workspace); \
        else hipLaunchKernelGGL((gbn_bwd_stats_kernel<T, 0>), gs, dim3(GBN_TPB), 0, s, (const T *)x, (const T *)dy, (const T *)nullptr, group_of, (const T *)mean, (const T *)invstd, (const T *)scale, (const T *)shift, S, C, G, R, rows, workspace); \
        hipLaunchKernelGGL(gbn_bwd_finalize_kernel<T>, gf, dim3(256), 0, s, workspace, group_samples, S, C, G, (const T *)weight, (const T *)mean, (const T *)invstd, k1, k2, k3, (T *)grad_weight, (T *)grad_bias); \
        if (mode == 2) hipLaunchKernelGGL((gbn_bwd_apply_kernel<T, 2>), ga, dim3(GBN_TPB), lds, s, (const T *)x, (const T *)dy, (const T *)y, group_of, k1, k2, k3, (const T *)scale, (const T *)shift, (T *)dx, (T *)grad_residual, S, C); \
        else if (relu) hipLaunchKernelGGL((gbn_bwd_apply_kernel<T, 1>), ga, dim3(GBN_TPB), lds, s, (const T *)x, (const T *)dy, (const T *)nullptr, group_of, k1, k2, k3, (const T *)scale, (const T *)shift, (T *)dx, (T *)nullptr, S, C); \
        else hipLaunchKernelGGL((gbn_bwd_apply_kernel<T, 0>), ga, dim3(GBN_TPB), lds, s, (const T *)x, (const T *)dy, (const T *)nullptr, group_of, k1, k2, k3, (const T *)scale, (const T *)shift, (T *)dx, (T *)nullptr, S, C); \
    }
    if (dtype == SP3D_GBN_F32) SP3D_GBN_BWD(float)
    else SP3D_GBN_BWD(double)
#undef SP3D_GBN_BWD
    return gbn_status();
}
