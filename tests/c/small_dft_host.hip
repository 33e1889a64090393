// Host-side check of csrc/small_dft.hpp (no GPU needed: the in-register DFTs are __host__ __device__): every supported
// size, forward and inverse, float and double, against a direct O(C^2) sum in double; frequency k is read at spos(C, k).
//   hipcc -std=c++17 -O1 tests/c/small_dft_host.hip -o small_dft_host && ./small_dft_host
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "../../grafx_amd/csrc/small_dft.hpp"

using cxf = float __attribute__((ext_vector_type(2)));
using cxd = double __attribute__((ext_vector_type(2)));

static double worst = 0.0;

template <int C, bool INV, typename V>
static void check(const char* what) {
    V v[C];
    double xr[C], xi[C];
    for (int n = 0; n < C; ++n) {
        xr[n] = (double)rand() / RAND_MAX - 0.5;
        xi[n] = (double)rand() / RAND_MAX - 0.5;
        v[n] = V{(typename gfx::sd_scalar<V>::type)xr[n], (typename gfx::sd_scalar<V>::type)xi[n]};
        xr[n] = v[n].x;     // the rounded inputs
        xi[n] = v[n].y;
    }
    gfx::sdft<C, INV>(v);
    double err = 0.0, mag = 0.0;
    for (int k = 0; k < C; ++k) {
        double sr = 0.0, si = 0.0;
        for (int n = 0; n < C; ++n) {
            const double a = (INV ? 2.0 : -2.0) * M_PI * (double)((n * k) % C) / C;
            sr += xr[n] * cos(a) - xi[n] * sin(a);
            si += xr[n] * sin(a) + xi[n] * cos(a);
        }
        const V got = v[gfx::spos(C, k)];
        err = fmax(err, fmax(fabs(got.x - sr), fabs(got.y - si)));
        mag = fmax(mag, fmax(fabs(sr), fabs(si)));
    }
    const double rel = err / mag, tol = sizeof(V) == 8 ? 2e-6 : 2e-15;
    worst = fmax(worst, rel / tol);
    if (rel > tol) {
        printf("FAIL %s C=%d inv=%d rel %.3e\n", what, C, (int)INV, rel);
        exit(1);
    }
}

template <int C>
static void size() {
    static_assert(gfx::sd_supported(C), "size");
    check<C, false, cxf>("float");
    check<C, true, cxf>("float");
    check<C, false, cxd>("double");
    check<C, true, cxd>("double");
}

int main() {
    srand(1);
#define X(C) size<C>();
    X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(12) X(14) X(15) X(16) X(18) X(20) X(21) X(24) X(25) X(27) X(28) X(30) X(32)
    X(35) X(36) X(40) X(42) X(45) X(48) X(49) X(50) X(54) X(56) X(60) X(63)   // the pair form's column passes (czt_pair.hip)
#undef X
    // spos is a permutation
    for (int C : {6, 9, 25, 27, 28, 30}) {
        int seen[64] = {0};
        for (int k = 0; k < C; ++k) seen[gfx::spos(C, k)]++;
        for (int k = 0; k < C; ++k)
            if (seen[k] != 1) {
                printf("FAIL spos(%d) is not a permutation\n", C);
                return 1;
            }
    }
    printf("SMALL_DFT_OK worst error / tolerance %.3f\n", worst);
    return 0;
}
