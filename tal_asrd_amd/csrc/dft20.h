// 20-point complex DFT in registers (float64): the building block of the 400 = 20 x 20 log-mel transform (csrc/logmel.hip).
// Prime-factor form (4 and 5 are coprime: no twiddles between the two passes):
//     n = (5 n1 + 4 n2) mod 20,   k = (5 k1 + 16 k2) mod 20
//     X[k] = sum_{n2} W5^{n2 k2} sum_{n1} W4^{n1 k1} x[n],        W_N = exp(-2 pi i / N)
// five 4-point transforms, then four 5-point transforms, in place; every index below is a compile-time constant once the
// loops are unrolled, so the arrays live in registers and the two index maps cost nothing.  ~330 float64 adds / multiplies.
#pragma once

#ifndef TAL_HD
#ifdef __HIPCC__
#define TAL_HD __host__ __device__ __forceinline__
#else
#define TAL_HD inline
#endif
#endif

namespace tal {

TAL_HD void dft4(double& r0, double& i0, double& r1, double& i1, double& r2, double& i2, double& r3, double& i3) {
    const double ar = r0 + r2, ai = i0 + i2, br = r0 - r2, bi = i0 - i2;
    const double cr = r1 + r3, ci = i1 + i3, dr = r1 - r3, di = i1 - i3;
    r0 = ar + cr; i0 = ai + ci;
    r2 = ar - cr; i2 = ai - ci;
    r1 = br + di; i1 = bi - dr;          // b - i d
    r3 = br - di; i3 = bi + dr;          // b + i d
}

TAL_HD void dft5(double& r0, double& i0, double& r1, double& i1, double& r2, double& i2, double& r3, double& i3, double& r4,
                 double& i4) {
    constexpr double C1 = 0.30901699437494742410229341718282;     // cos(2 pi / 5)
    constexpr double C2 = -0.80901699437494742410229341718282;    // cos(4 pi / 5)
    constexpr double S1 = 0.95105651629515357211643933337938;     // sin(2 pi / 5)
    constexpr double S2 = 0.58778525229247312916870595463907;     // sin(4 pi / 5)
    const double t1r = r1 + r4, t1i = i1 + i4, t2r = r2 + r3, t2i = i2 + i3;
    const double t3r = r1 - r4, t3i = i1 - i4, t4r = r2 - r3, t4i = i2 - i3;
    const double m1r = r0 + (C1 * t1r + C2 * t2r), m1i = i0 + (C1 * t1i + C2 * t2i);
    const double m2r = r0 + (C2 * t1r + C1 * t2r), m2i = i0 + (C2 * t1i + C1 * t2i);
    const double p1r = S1 * t3r + S2 * t4r, p1i = S1 * t3i + S2 * t4i;
    const double p2r = S2 * t3r - S1 * t4r, p2i = S2 * t3i - S1 * t4i;
    r0 = r0 + (t1r + t2r); i0 = i0 + (t1i + t2i);
    r1 = m1r + p1i; i1 = m1i - p1r;      // m1 - i p1
    r4 = m1r - p1i; i4 = m1i + p1r;
    r2 = m2r + p2i; i2 = m2i - p2r;      // m2 - i p2
    r3 = m2r - p2i; i3 = m2i + p2r;
}

// x[0..19] (natural order) -> X[0..19] (natural order); `xr / xi` are overwritten with scratch.
TAL_HD void dft20(double (&xr)[20], double (&xi)[20], double (&yr)[20], double (&yi)[20]) {
#pragma unroll
    for (int n2 = 0; n2 < 5; ++n2) {
        const int a = (4 * n2) % 20, b = (5 + 4 * n2) % 20, c = (10 + 4 * n2) % 20, d = (15 + 4 * n2) % 20;
        dft4(xr[a], xi[a], xr[b], xi[b], xr[c], xi[c], xr[d], xi[d]);
    }
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
        const int a = (5 * k1) % 20, b = (5 * k1 + 4) % 20, c = (5 * k1 + 8) % 20, d = (5 * k1 + 12) % 20, e = (5 * k1 + 16) % 20;
        dft5(xr[a], xi[a], xr[b], xi[b], xr[c], xi[c], xr[d], xi[d], xr[e], xi[e]);
    }
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
        for (int k2 = 0; k2 < 5; ++k2) {
            yr[(5 * k1 + 16 * k2) % 20] = xr[(5 * k1 + 4 * k2) % 20];
            yi[(5 * k1 + 16 * k2) % 20] = xi[(5 * k1 + 4 * k2) % 20];
        }
}

}  // namespace tal
