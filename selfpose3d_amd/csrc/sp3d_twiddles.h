// Compile-time DFT twiddle tables, shared by the z-DFT kernels of sp3d_fft.hip and the unprojection kernel that emits the
// z-spectrum directly (sp3d_unproject.hip, unproject_brick_kernel<..., ZD>): ONE table, so both produce the same bits.
#pragma once

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

// Z real samples (zero-padded to SZ, SZ even) -> the SZ/2+1 bins of their DFT, used by BOTH z-pass producers (zdft_fwd_cl_kernel
// and the fused unprojection epilogue) so that they give the same bits.  Even and odd samples are summed apart:
// X_k = E_k + O_k and X_{SZ/2-k} = conj(E_k) - conj(O_k) ((-1)^z = W^{(SZ/2) z}), so only the bins k <= SZ/4 are evaluated:
// 2*Z FMAs per pair of bins instead of 4*Z (SZ = 28, Z = 20: 320 FMAs + 28 adds instead of 600 FMAs).
template <int Z, int SZ>
__device__ __forceinline__ void zdft_real(const float (&v)[Z], float (&re)[SZ / 2 + 1], float (&im)[SZ / 2 + 1])
{
    static_assert(SZ % 2 == 0, "even transform length");
    constexpr Twiddles<SZ> tw{};
    constexpr int H = SZ / 2;
#pragma unroll
    for (int k = 0; 2 * k <= H; ++k) {
        float er = 0.0f, ei = 0.0f, orr = 0.0f, oi = 0.0f;
#pragma unroll
        for (int z = 0; z < Z; z += 2) {
            const int m = (k * z) % SZ;
            er = fmaf(v[z], tw.c[m], er);
            ei = fmaf(v[z], -tw.s[m], ei);
        }
#pragma unroll
        for (int z = 1; z < Z; z += 2) {
            const int m = (k * z) % SZ;
            orr = fmaf(v[z], tw.c[m], orr);
            oi = fmaf(v[z], -tw.s[m], oi);
        }
        re[k] = er + orr;
        im[k] = ei + oi;
        if (2 * k != H) {
            re[H - k] = er - orr;
            im[H - k] = oi - ei;
        }
    }
}

} // namespace sp3d
