// In-register DFTs of small composite sizes (factors 2, 3, 5, 7) for the column passes of the chirp-z transforms
// (czt.hip): with C in {3, 5, 6, 7, 9, 10, 12, ...} tiles per signal row next to the powers of two, the transform size
// C x 8192 follows the (3P - 1) / 2 points a chirp-z convolution needs much more closely than the next power of two does
// (P = 135 071: 25 tiles instead of 32; P = 191 071: 36 instead of 64).
//
// sdft<C, INV>(v) is an in-place decimation-in-frequency Cooley-Tukey over the smallest prime factor p of C:
//   n = i m + j,  k = q + p k':   X[q + p k'] = sum_j [ (sum_i x[i m + j] W_p^(i q)) W_C^(j q) ] W_m^(j k')
// (p-point butterflies over stride m, twiddle, then m-point transforms of the p contiguous blocks), so frequency k ends
// up at position spos(C, k) = (k % p) m + spos(m, k / p) -- the bit reversal when C is a power of two.  All roots of
// unity are compile-time constants (constexpr tables, evaluated in double: Taylor series after an exact octant reduction
// in integers), so that after full unrolling every twiddle is an immediate.
#pragma once
#include <hip/hip_runtime.h>

namespace gfx {

constexpr double kSdPi = 3.141592653589793238462643383279502884;

constexpr double sd_sin_small(double x) {   // |x| <= pi / 4
    double term = x, sum = x;
    const double x2 = x * x;
    for (int n = 1; n < 12; ++n) {
        term *= -x2 / (double)((2 * n) * (2 * n + 1));
        sum += term;
    }
    return sum;
}
constexpr double sd_cos_small(double x) {   // |x| <= pi / 4
    double term = 1.0, sum = 1.0;
    const double x2 = x * x;
    for (int n = 1; n < 12; ++n) {
        term *= -x2 / (double)((2 * n - 1) * (2 * n));
        sum += term;
    }
    return sum;
}
// cos / sin of 2 pi num / den: the nearest quarter turn is taken off in integers, the rest (|.| <= 1/8 turn) by series
constexpr double sd_unit(int num, int den, bool want_sin) {
    int r = num % den;
    if (r < 0) r += den;
    const int q = (8 * r + den) / (2 * den);                       // nearest quarter turn, 0..4
    const double a = kSdPi * (double)(4 * r - q * den) / (double)(2 * den);
    const double c = sd_cos_small(a), s = sd_sin_small(a);
    switch (q & 3) {
        case 0: return want_sin ? s : c;
        case 1: return want_sin ? c : -s;
        case 2: return want_sin ? -s : -c;
        default: return want_sin ? -c : s;
    }
}

template <int N>
struct SdRoots {            // e^{+2 pi i k / N}
    double re[N], im[N];
    constexpr SdRoots() : re{}, im{} {
        for (int k = 0; k < N; ++k) {
            re[k] = sd_unit(k, N, false);
            im[k] = sd_unit(k, N, true);
        }
    }
};
// (used as function-local `constexpr SdRoots<N> R{}` objects: a __device__ global would be externally initialisable and
// its loads would not fold into immediates)

constexpr int sd_factor(int n) { return n % 2 == 0 ? 2 : n % 3 == 0 ? 3 : n % 5 == 0 ? 5 : n % 7 == 0 ? 7 : n; }
constexpr bool sd_supported(int n) {
    while (n > 1) {
        const int p = sd_factor(n);
        if (p > 7) return false;
        n /= p;
    }
    return true;
}
// position of frequency k after sdft<N>  (a loop of FIXED length, so that the optimiser unrolls it and folds the result
// once N and k are constants: a register array indexed with it must not end up in scratch memory)
constexpr int spos(int N, int k) {
    int pos = 0;
    for (int it = 0; it < 6; ++it) {           // N <= 64 has at most six prime factors
        if (N > 1) {
            const int p = sd_factor(N), m = N / p;
            pos += (k % p) * m;
            k /= p;
            N = m;
        }
    }
    return pos;
}

// scalar type of a two-element complex vector (decltype(v.x) of an ext_vector element is not plain float / double)
template <typename V> struct sd_scalar { using type = double; };
template <> struct sd_scalar<float __attribute__((ext_vector_type(2)))> { using type = float; };

// a * (wr + i wi) with compile-time wr, wi (S: float or double, V: the matching two-element vector)
template <typename V, typename S>
__host__ __device__ __forceinline__ V sd_mulk(V a, S wr, S wi) {
    return V{a.x * wr - a.y * wi, a.x * wi + a.y * wr};
}

// p-point DFT over v[base + i * stride], i < P, in place; exponent sign - (forward) or + (INV)
template <int P, bool INV, typename V, int N>
__host__ __device__ __forceinline__ void sd_butterfly(V (&v)[N], int base, int stride) {
    using S = typename sd_scalar<V>::type;
    if constexpr (P == 2) {
        const V a = v[base], b = v[base + stride];
        v[base] = a + b;
        v[base + stride] = a - b;
    } else {
        // pairs (i, P - i): S_i = a_i + a_{P-i}, D_i = a_i - a_{P-i};  out_q = a_0 + sum_i cos(2 pi i q / P) S_i -+ i sin(.) D_i
        constexpr int H = (P - 1) / 2;
        constexpr SdRoots<P> R{};
        V a0 = v[base], s[H], d[H];
#pragma unroll
        for (int i = 1; i <= H; ++i) {
            const V x = v[base + i * stride], y = v[base + (P - i) * stride];
            s[i - 1] = x + y;
            d[i - 1] = x - y;
        }
        V sum = a0;
#pragma unroll
        for (int i = 0; i < H; ++i) sum += s[i];
        v[base] = sum;
#pragma unroll
        for (int q = 1; q <= H; ++q) {
            V re = a0, im = V{0, 0};
#pragma unroll
            for (int i = 1; i <= H; ++i) {
                const S c = (S)R.re[(i * q) % P], sn = (S)R.im[(i * q) % P];
                re += s[i - 1] * c;
                im += d[i - 1] * sn;
            }
            // forward (e^{-i theta}): out_q = re - i im, out_{P-q} = re + i im;  inverse: the other way round
            const V iim = V{-im.y, im.x};     // i * im
            v[base + q * stride] = INV ? re + iim : re - iim;
            v[base + (P - q) * stride] = INV ? re - iim : re + iim;
        }
    }
}

template <int C, bool INV, typename V, int N>
__host__ __device__ __forceinline__ void sdft_at(V (&v)[N], int base) {
    using S = typename sd_scalar<V>::type;
    if constexpr (C > 1) {
        constexpr int P = sd_factor(C), M = C / P;
        static_assert(P <= 7, "sdft: sizes with prime factors up to 7");
#pragma unroll
        for (int j = 0; j < M; ++j) sd_butterfly<P, INV>(v, base + j, M);
        if constexpr (M > 1) {
            constexpr SdRoots<C> R{};
#pragma unroll
            for (int q = 1; q < P; ++q) {
#pragma unroll
                for (int j = 1; j < M; ++j) {
                    const S wr = (S)R.re[(j * q) % C], wi = (S)R.im[(j * q) % C];
                    v[base + q * M + j] = sd_mulk(v[base + q * M + j], wr, INV ? wi : -wi);
                }
            }
#pragma unroll
            for (int q = 0; q < P; ++q) sdft_at<M, INV>(v, base + q * M);
        }
    }
}

// DFT of v[0 .. C) in place; frequency k at v[spos(C, k)]
template <int C, bool INV, typename V>
__host__ __device__ __forceinline__ void sdft(V (&v)[C]) {
    sdft_at<C, INV>(v, 0);
}

}  // namespace gfx
