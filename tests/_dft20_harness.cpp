// Host harness for tal_asrd_amd/csrc/dft20.h (the header compiles for the host too): reads complex vectors of 20 values
// (re im per line) from stdin, writes their 20-point DFTs; `400` as argv[1]: reads 2 x 400 real samples + 400 window taps and
// writes the 201-bin power spectra of the two frames the way csrc/logmel.hip combines dft20 (pair trick, 20 x 20, twiddles
// by the W^q / W^4q recurrence).  tests/test_dft20_cpu.py drives it.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../tal_asrd_amd/csrc/dft20.h"

int main(int argc, char** argv) {
    using namespace tal;
    if (argc > 1 && !strcmp(argv[1], "400")) {
        std::vector<double> a(400), b(400), w(400);
        for (auto* v : {&a, &b, &w})
            for (int n = 0; n < 400; ++n)
                if (scanf("%lf", &(*v)[n]) != 1) return 1;
        const double two_pi = 6.283185307179586476925286766559;
        std::vector<double> Yr(400), Yi(400), Zr(400), Zi(400);
        double xr[20], xi[20], yr[20], yi[20];
        for (int q = 0; q < 20; ++q) {
            for (int n1 = 0; n1 < 20; ++n1) { xr[n1] = w[20 * n1 + q] * a[20 * n1 + q]; xi[n1] = w[20 * n1 + q] * b[20 * n1 + q]; }
            dft20(xr, xi, yr, yi);
            const double t1r = cos(two_pi * q / 400.0), t1i = -sin(two_pi * q / 400.0);
            const double t4r = cos(two_pi * 4 * q / 400.0), t4i = -sin(two_pi * 4 * q / 400.0);
            double tr[4] = {1.0, t1r, t1r * t1r - t1i * t1i, 0.0}, ti[4] = {0.0, t1i, 2.0 * t1r * t1i, 0.0};
            tr[3] = tr[2] * t1r - ti[2] * t1i; ti[3] = tr[2] * t1i + ti[2] * t1r;
            for (int aa = 0; aa < 5; ++aa)
                for (int c = 0; c < 4; ++c) {
                    const int k1 = 4 * aa + c;
                    Yr[k1 * 20 + q] = yr[k1] * tr[c] - yi[k1] * ti[c];
                    Yi[k1 * 20 + q] = yr[k1] * ti[c] + yi[k1] * tr[c];
                    const double nr = tr[c] * t4r - ti[c] * t4i, ni = tr[c] * t4i + ti[c] * t4r;
                    tr[c] = nr; ti[c] = ni;
                }
        }
        for (int k1 = 0; k1 < 20; ++k1) {
            for (int n2 = 0; n2 < 20; ++n2) { xr[n2] = Yr[k1 * 20 + n2]; xi[n2] = Yi[k1 * 20 + n2]; }
            dft20(xr, xi, yr, yi);
            for (int k2 = 0; k2 < 20; ++k2) { Zr[k1 + 20 * k2] = yr[k2]; Zi[k1 + 20 * k2] = yi[k2]; }
        }
        for (int k = 0; k <= 200; ++k) {
            const int m = k ? 400 - k : 0;
            const double ar = Zr[k] + Zr[m], ai = Zi[k] - Zi[m], br = Zi[k] + Zi[m], bi = Zr[k] - Zr[m];
            printf("%.17g %.17g\n", 0.25 * (ar * ar + ai * ai), 0.25 * (br * br + bi * bi));
        }
        return 0;
    }
    double xr[20], xi[20], yr[20], yi[20];
    for (;;) {
        for (int i = 0; i < 20; ++i)
            if (scanf("%lf %lf", &xr[i], &xi[i]) != 2) return 0;
        dft20(xr, xi, yr, yi);
        for (int i = 0; i < 20; ++i) printf("%.17g %.17g\n", yr[i], yi[i]);
    }
}
