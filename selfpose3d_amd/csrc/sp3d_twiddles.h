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

} // namespace sp3d
