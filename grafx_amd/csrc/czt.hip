// The reference's odd-length aliasing, y = irfft_{P-1}(rfft_P(z)), as two chirp-z transforms on the LDS FFT tile.
//
// convolve() (core/convolution.py:119-134) pads to P = Lx + Lh - 1, multiplies rffts and calls irfft WITHOUT n, i.e.
// with length 2 (P // 2).  For odd P that inverts a P-point spectrum on a (P-1)-point grid -- a global resampling of the
// linear convolution z.  Every reference default (4000 / 16384 / 60000 taps with even
// audio lengths) takes that path, so reproducing it is part of parity.  P is arbitrary (131072 + 4000 - 1 = 135071 is
// not even composite-friendly), hence Bluestein:
//
//   Z[k] = cP[k]  sum_m (z[m] cP[m]) bP[k - m],   cP[k] = e^{-i pi k^2 / P},  bP[j] = e^{+i pi j^2 / P},  k < K = (P+1)/2
//   y[n] = Re( cQ[n] sum_k (w_k Z[k] cQ[k]) bQ[n - k] ) / Q,   cQ[k] = e^{+i pi k^2 / Q},  bQ[j] = e^{-i pi j^2 / Q},
//          Q = P - 1,  w_0 = w_{K-1} = 1, else 2   (taking the real part drops Im Z[0], Im Z[K-1] like a c2r transform)
//
// two circular convolutions of NFFT = C x 8192 >= (3P - 1) / 2 points (C = the smallest count of tiles with prime factors
// up to 7 that covers them: 25 for P = 135 071, where the next power of two is 32), each a "four-step" FFT around the tile:
//   cols_fwd : per column n2 (of 8192) a C-point DFT over n1 (registers) and the twiddle W_NFFT^(n2 k1)
//   rows     : per row k1 one tile: forward, multiply by the chirp's spectrum (thread layout, precomputed), inverse
//   cols_inv : conjugate twiddle, C-point inverse DFT, then the chirp / weights of the next step
// all in place on one NFFT-point complex buffer per signal row.  fp32 throughout; the chirps are evaluated once per P
// in double with the phase reduced exactly (k^2 mod 2P in integers).  C <= 32 covers P <= 174,763; beyond, up to three
// OUTER radix-4 levels split the transform into 4 / 16 / 64 sub-transforms of C x 8192 points (czt_outer_*: 4-point DFT
// over the quarters + twiddle W_N^(n' k3), in place, applied recursively): NFFT <= 2^24, P <= 11,184,811 (233 s of audio at
// 48 kHz plus the filter).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <mutex>

#include "../../include/grafx_amd.h"
#include "fft_tile.hpp"
#include "fft_tile_f64.hpp"
#include "small_dft.hpp"

namespace gfx {

constexpr int CZT_MAXC = 32;
// tiles per (sub-)transform: every size up to 32 with prime factors up to 7 (czt_geom picks the smallest that covers P)
#define GFX_CZT_SIZES(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(12) X(14) X(15) X(16) X(18) X(20) X(21) X(24) \
    X(25) X(27) X(28) X(30) X(32)

// Working precision of the transforms (the data in and out is fp32 either way): float = the packed-FP32 tile, double =
// the `precise` form for the energy envelope (fft_tile_f64.hpp).  Tables, spectra and the workspace are T2 per point.
template <typename T> struct Prec;
template <> struct Prec<float> {
    using cxt = cx;
    using T2 = float2;
    using Tw = TileTw;
    static constexpr int lds_bytes = TILE_LDS_BYTES;
    static __device__ __forceinline__ T2 make(float x, float y) { return make_float2(x, y); }
};
template <> struct Prec<double> {
    using cxt = cxd;
    using T2 = double2;
    using Tw = TileTwD;
    static constexpr int lds_bytes = TILE_LDS_BYTES_F64;
    static __device__ __forceinline__ T2 make(double x, double y) { return make_double2(x, y); }
};

struct CztGeom {
    int64_t P, Q, K, NFFT;   // NFFT = S * C * 8192
    int C, S;                // C <= 32 columns per sub-transform; S = 4^levels sub-transforms under `levels` outer radix-4 levels
    int levels;
    // where the real output rows go: row r of the call at y + r * ldy (yC == 0), or -- gfx_odd_alias_rows_f32 -- row
    // (row0 + r) of a (rows / yC, yC, len) signal addressed through a row map (a strided view of the render's buffer)
    int yC;
    int64_t row0;
    gfx_rowmap_t ymap;
};

__device__ __forceinline__ float* czt_out_row(const CztGeom& g, float* y, int64_t ldy, int64_t row) {
    if (g.yC == 0) return y + row * ldy;
    const int64_t q = g.row0 + row;
    const unsigned r = (unsigned)(q / g.yC);
    const int c = (int)(q - (int64_t)r * g.yC);
    const unsigned inner = (unsigned)g.ymap.inner, o = r / inner, rem = r - o * inner;
    return y + (int64_t)o * g.ymap.stride_outer + (int64_t)rem * g.ymap.stride_inner + (int64_t)c * g.ymap.stride_ch;
}

constexpr int CZT_MAX_LEVELS = 3;   // NFFT <= 2^24: P <= 11,184,811 (233 s of audio at 48 kHz plus the filter)

static inline bool czt_geom(int64_t P, CztGeom& g) {
    if (P < 3 || (P & 1) == 0) return false;
    g.yC = 0;
    g.row0 = 0;
    g.ymap = gfx_rowmap_t{1, 0, 0, 0};
    g.P = P;
    g.Q = P - 1;
    g.K = (P + 1) / 2;
    const int64_t need = P + g.K - 1;
    const int64_t tiles = (need + TILE_M - 1) / TILE_M;
    // C <= 32 tiles per (sub-)transform with prime factors up to 7 (small_dft.hpp); beyond 32, outer radix-4 levels
    g.levels = 0;
    int64_t per = tiles;
    while (per > CZT_MAXC) {
        per = (per + 3) / 4;
        ++g.levels;
    }
    int64_t C = per;
    while (!sd_supported((int)C)) ++C;       // (32 is supported: the loop ends)
    if (g.levels > CZT_MAX_LEVELS || C < 1) return false;
    g.S = 1 << (2 * g.levels);
    g.C = (int)C;
    g.NFFT = (int64_t)g.S * C * TILE_M;
    return true;
}

// plan layout (float2 units): cP[P] | cQ[Q] | spectra [NFFT] each of bP, bQ (forward) and of bQ, bP placed for the
// adjoint (same chirps, mirrored support: the adjoint's first transform has Q inputs and K outputs, its second K
// inputs and P outputs)
static inline size_t czt_plan_f2(const CztGeom& g) { return (size_t)(g.P + g.Q + 4 * g.NFFT); }

// b[j] = exp(sign i pi j^2 / den) at circular index j mod NFFT for j in [-lo, hi], zero elsewhere
struct ChirpSeq {
    int64_t lo, hi, den;
    double sign;
};

template <typename T>
__device__ __forceinline__ typename Prec<T>::T2 chirp_d(int64_t j, int64_t den, double sign) {   // exp(sign i pi j^2 / den)
    const int64_t r = (j * j) % (2 * den);
    double s, c;
    sincospi((double)r / (double)den, &s, &c);
    return Prec<T>::make((T)c, (T)(sign * s));
}

template <typename T>
__global__ void czt_chirp_table_kernel(typename Prec<T>::T2* __restrict__ tab, int64_t n, int64_t den, float sign) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) tab[i] = chirp_d<T>(i, den, sign);
}

// e^{-+ 2 pi i r / N} for 0 <= r < N, N a multiple of 4 (N = C x 8192, not a power of two in general): the quarter turn is
// taken off in integers and the rest, rem / (N / 2) <= 1/2 half-turns with rem < 2^24 exact, goes to sincospif -- the phase
// is good to 1e-7 rad whatever N (for a power of two the quotient is exact, as before).
__device__ __forceinline__ cx unit_root_f(int r, int N, bool conj) {
    const int quarter = N >> 2;
    const int q = r / quarter, rem = r - q * quarter;
    float s, c;
    sincospif((float)rem / (float)(N >> 1), &s, &c);
    const float cr = (q & 1) ? ((q & 2) ? s : -s) : ((q & 2) ? -c : c);     // cos of the full angle
    const float sr = (q & 1) ? ((q & 2) ? -c : c) : ((q & 2) ? -s : s);     // sin of the full angle
    return cx{cr, conj ? sr : -sr};
}
__device__ __forceinline__ cxd col_twiddle(int n2, int k1, double inv_half_nfft, bool conj) {
    double s, c;
    sincospi((double)(n2 * k1) * inv_half_nfft, &s, &c);
    return cxd{c, conj ? s : -s};
}

// The column twiddles W_NS^(n2 k1), k1 = 1, 2, ... C-1 IN TURN.  float: n2 k1 mod NS carried incrementally, one sincospif
// each; double: powers of W^(n2) by repeated multiplication (31 steps lose ~1e-15, and a double sincospi per point would
// make the column kernels compute-bound).
template <typename T> struct ColTw;
template <> struct ColTw<float> {
    int n2, NS, r;
    bool conj;
    __device__ __forceinline__ ColTw(int n2_, int NS_, bool conj_) : n2(n2_), NS(NS_), r(0), conj(conj_) {}
    __device__ __forceinline__ cx at(int) {
        r += n2;
        r -= r >= NS ? NS : 0;
        return unit_root_f(r, NS, conj);
    }
};
template <> struct ColTw<double> {
    cxd w1, w;
    __device__ __forceinline__ ColTw(int n2, int NS, bool conj) : w1(col_twiddle(n2, 1, 2.0 / (double)NS, conj)), w(cxd{1.0, 0.0}) {}
    __device__ __forceinline__ cxd at(int) { w = cmul(w, w1); return w; }
};

// C-point DFT of a column in registers: the power-of-two sizes on the tile's radix-2 codelet (bit-identical with the
// rounds before), the others on small_dft.hpp; either way frequency k ends up at v[spos(C, k)].
template <int C, bool INV, typename V>
__device__ __forceinline__ void col_dft(V (&v)[C]) {
    if constexpr ((C & (C - 1)) == 0) dif<C, INV>(v);
    else sdft<C, INV>(v);
}

// MODE 0: real rows, z[row, i - lo] tab[i] for lo <= i < lo + len (row stride ldz), zero elsewhere;  MODE 1: the complex
// buffer itself;  MODE 2: a chirp sequence (plan building)
template <typename T, int C, int MODE>
__global__ __launch_bounds__(256) void czt_cols_fwd_kernel(const float* __restrict__ z,
                                                          const typename Prec<T>::T2* __restrict__ tab,
                                                          typename Prec<T>::T2* __restrict__ buf, CztGeom g, int64_t ldz,
                                                          int64_t lo, int64_t len, ChirpSeq cs) {
    using cx = typename Prec<T>::cxt;
    const int n2 = blockIdx.x * 256 + threadIdx.x;          // column 0..8191
    const int64_t row = blockIdx.y;                        // signal row * S + sub-transform
    const int64_t NS = g.NFFT / g.S;                       // points of one sub-transform
    typename Prec<T>::T2* b = buf + row * NS;
    cx v[C];
#pragma unroll
    for (int n1 = 0; n1 < C; ++n1) {
        const int64_t i = (int64_t)n1 * TILE_M + n2;
        cx e = {0, 0};
        if (MODE == 0) {
            if (i >= lo && i < lo + len) e = to_cx(tab[i]) * (T)z[row * ldz + (i - lo)];
        } else if (MODE == 1) {
            e = to_cx(b[i]);
        } else {
            if (i <= cs.hi) e = to_cx(chirp_d<T>(i, cs.den, cs.sign));
            else if (i >= g.NFFT - cs.lo) e = to_cx(chirp_d<T>(g.NFFT - i, cs.den, cs.sign));
        }
        v[n1] = e;
    }
    col_dft<C, false>(v);
    ColTw<T> tw(n2, (int)NS, false);
#pragma unroll
    for (int k1 = 0; k1 < C; ++k1) {
        const cx e = v[spos(C, k1)];
        const cx o = k1 == 0 ? e : cmul(e, tw.at(k1));
        b[(int64_t)k1 * TILE_M + n2] = Prec<T>::make(o.x, o.y);
    }
}

// One tile per (row, k1): forward, times the chirp spectrum, inverse -- in place.  PLAN: forward only, spectrum stored
// in thread layout.
template <typename T, bool PLAN>
__global__ __launch_bounds__(TILE_T, sizeof(T) == 4 ? 2 : 1) void czt_rows_kernel(
    typename Prec<T>::T2* __restrict__ buf, const typename Prec<T>::T2* __restrict__ spec,
    typename Prec<T>::T2* __restrict__ spec_out, int C, const typename Prec<T>::T2* __restrict__ twtab) {
    using cx = typename Prec<T>::cxt;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    cx* lds = reinterpret_cast<cx*>(lds_raw);
    const int t = threadIdx.x;
    const int64_t tile = blockIdx.x;                    // row * C + k1
    const int k1 = (int)(tile % C);                       // C here = tiles per signal row = S * C
    cx* b = reinterpret_cast<cx*>(buf) + tile * TILE_M;
    typename Prec<T>::Tw tw;
    tile_twiddles(tw, twtab, t);
    cx v[32], w[2][16];
#pragma unroll
    for (int a = 0; a < 32; ++a) v[a] = b[t + 256 * a];
    tile_forward(v, w, tw, lds, t);
    if (PLAN) {
        cx* o = reinterpret_cast<cx*>(spec_out) + (int64_t)k1 * TILE_M;
#pragma unroll
        for (int q = 0; q < 32; ++q) o[q * TILE_T + t] = w[q >> 4][q & 15];
        return;
    }
    const cx* sp = reinterpret_cast<const cx*>(spec) + (int64_t)k1 * TILE_M;
#pragma unroll
    for (int q = 0; q < 32; ++q) w[q >> 4][q & 15] = cmul(w[q >> 4][q & 15], sp[q * TILE_T + t]);
    __syncthreads();
    tile_inverse(w, v, tw, lds, t);
#pragma unroll
    for (int a = 0; a < 32; ++a) b[t + 256 * a] = v[brev(a, 5)];
}

// MODE 0 (after the first convolution): buf[k] <- conv[k] cP[k] w_k cQ[k] / NFFT for k < K, zero beyond
// MODE 1 (after the second):            y[row, n - lo] <- Re(conv[n] cQ[n]) / (NFFT Q)  for lo <= n < lo + len
// MODE 3: MODE 0 and the second convolution's column pass (cols_fwd MODE 1) in one -- the same thread owns the column in
//         both, so the buffer is read and written once instead of twice
// (the adjoint passes cP in cQ's place for MODE 1: its output lives on the P grid, still divided by Q)
template <typename T, int C, int MODE>
__global__ __launch_bounds__(256) void czt_cols_inv_kernel(typename Prec<T>::T2* __restrict__ buf,
                                                          const typename Prec<T>::T2* __restrict__ cP,
                                                          const typename Prec<T>::T2* __restrict__ cQ,
                                                          float* __restrict__ y, int64_t ldy, int64_t lo, int64_t len,
                                                          CztGeom g) {
    using cx = typename Prec<T>::cxt;
    const int n2 = blockIdx.x * 256 + threadIdx.x;
    const int64_t row = blockIdx.y;
    const int64_t NS = g.NFFT / g.S;
    typename Prec<T>::T2* b = buf + row * NS;
    cx v[C];
    ColTw<T> twi(n2, (int)NS, true);
#pragma unroll
    for (int k1 = 0; k1 < C; ++k1) {
        const cx e = to_cx(b[(int64_t)k1 * TILE_M + n2]);
        v[k1] = k1 == 0 ? e : cmul(e, twi.at(k1));
    }
    col_dft<C, true>(v);
    const T sc = (T)1 / (T)NS;
    if (MODE == 3) {   // the step between the two convolutions (S = 1): MODE 0's values, then cols_fwd's MODE 1 on them
        cx u[C];
#pragma unroll
        for (int n1 = 0; n1 < C; ++n1) {
            const int64_t i = (int64_t)n1 * TILE_M + n2;
            cx o = {0, 0};
            if (i < g.K) {
                const T wk = (i == 0 || i == g.K - 1) ? (T)1 : (T)2;
                o = cmul(cmul(v[spos(C, n1)] * sc, to_cx(cP[i])), to_cx(cQ[i])) * wk;
            }
            u[n1] = o;
        }
        col_dft<C, false>(u);
        ColTw<T> twf(n2, (int)NS, false);
#pragma unroll
        for (int k1 = 0; k1 < C; ++k1) {
            const cx e = u[spos(C, k1)];
            const cx o = k1 == 0 ? e : cmul(e, twf.at(k1));
            b[(int64_t)k1 * TILE_M + n2] = Prec<T>::make(o.x, o.y);
        }
        return;
    }
#pragma unroll
    for (int n1 = 0; n1 < C; ++n1) {
        const int64_t i = (int64_t)n1 * TILE_M + n2;
        const cx e = v[spos(C, n1)] * sc;
        if (MODE == 2) {                                   // plain inverse of a sub-transform (outer level follows)
            b[i] = Prec<T>::make(e.x, e.y);
        } else if (MODE == 0) {
            cx o = {0, 0};
            if (i < g.K) {
                const T wk = (i == 0 || i == g.K - 1) ? (T)1 : (T)2;
                o = cmul(cmul(e, to_cx(cP[i])), to_cx(cQ[i])) * wk;
            }
            b[i] = Prec<T>::make(o.x, o.y);
        } else {
            if (i >= lo && i < lo + len) {
                const cx c = to_cx(cQ[i]);
                czt_out_row(g, y, ldy, row)[i - lo] = (float)((e.x * c.x - e.y * c.y) / (T)g.Q);
            }
        }
    }
}

// ---- outer radix-4 level (S = 4): NFFT = 4 NS, n = n3 NS + n', k = k3 + 4 k' ---------------------------------------
//   forward:  sub[k3][n'] = ( sum_n3 x[n3 NS + n'] W_4^(n3 k3) ) W_NFFT^(n' k3)      then four NS-point transforms
//   inverse:  x[n3 NS + n'] = (1/4) sum_k3 ( sub[k3][n'] conj W_NFFT^(n' k3) ) W_4^(-n3 k3)
// in place (a thread owns the four positions n' + n3 NS).  Input / output modes as in the column kernels.
template <typename T>
__device__ __forceinline__ typename Prec<T>::cxt outer_twiddle(int64_t np, int k3, int64_t NFFT, bool conj) {
    // W_NFFT^(np k3), np k3 < NFFT <= 2^24
    if constexpr (sizeof(T) == 4) {
        return unit_root_f((int)(np * k3), (int)NFFT, conj);
    } else {
        double s, c;
        sincospi(2.0 * (double)(np * k3) / (double)NFFT, &s, &c);
        return cxd{c, conj ? s : -s};
    }
}

template <typename T, int MODE>   // 0: real rows times a chirp table; 1: complex buffer; 2: a chirp sequence (plan)
__global__ __launch_bounds__(256) void czt_outer_fwd_kernel(const float* __restrict__ z,
                                                           const typename Prec<T>::T2* __restrict__ tab,
                                                           typename Prec<T>::T2* __restrict__ buf, CztGeom g, int64_t ldz,
                                                           int64_t lo, int64_t len, ChirpSeq cs) {
    using cx = typename Prec<T>::cxt;
    const int64_t NS = g.NFFT / 4;
    const int64_t np = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = blockIdx.y;
    typename Prec<T>::T2* b = buf + row * g.NFFT;
    cx v[4];
#pragma unroll
    for (int n3 = 0; n3 < 4; ++n3) {
        const int64_t i = n3 * NS + np;
        cx e = {0, 0};
        if (MODE == 0) {
            if (i >= lo && i < lo + len) e = to_cx(tab[i]) * (T)z[row * ldz + (i - lo)];
        } else if (MODE == 1) {
            e = to_cx(b[i]);
        } else {
            if (i <= cs.hi) e = to_cx(chirp_d<T>(i, cs.den, cs.sign));
            else if (i >= g.NFFT - cs.lo) e = to_cx(chirp_d<T>(g.NFFT - i, cs.den, cs.sign));
        }
        v[n3] = e;
    }
    dif<4, false>(v);                                     // result for k3 at v[brev(k3, 2)]
#pragma unroll
    for (int k3 = 0; k3 < 4; ++k3) {
        const cx e = v[brev(k3, 2)];
        const cx o = k3 == 0 ? e : cmul(e, outer_twiddle<T>(np, k3, g.NFFT, false));
        b[k3 * NS + np] = Prec<T>::make(o.x, o.y);
    }
}

template <typename T, int MODE>   // 0: next step's input (times cP w_k cQ for k < K, zero beyond); 1: real output slice
__global__ __launch_bounds__(256) void czt_outer_inv_kernel(typename Prec<T>::T2* __restrict__ buf,
                                                           const typename Prec<T>::T2* __restrict__ cP,
                                                           const typename Prec<T>::T2* __restrict__ cQ,
                                                           float* __restrict__ y, int64_t ldy, int64_t lo, int64_t len,
                                                           CztGeom g) {
    using cx = typename Prec<T>::cxt;
    const int64_t NS = g.NFFT / 4;
    const int64_t np = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = blockIdx.y;
    typename Prec<T>::T2* b = buf + row * g.NFFT;
    cx v[4];
#pragma unroll
    for (int k3 = 0; k3 < 4; ++k3) {
        const cx e = to_cx(b[k3 * NS + np]);
        v[k3] = k3 == 0 ? e : cmul(e, outer_twiddle<T>(np, k3, g.NFFT, true));
    }
    dif<4, true>(v);
    if (MODE == 2) {   // MODE 0's values, then czt_outer_fwd_kernel's MODE 1 on them: one pass over the buffer
        cx u[4];
#pragma unroll
        for (int n3 = 0; n3 < 4; ++n3) {
            const int64_t i = n3 * NS + np;
            cx o = {0, 0};
            if (i < g.K) {
                const T wk = (i == 0 || i == g.K - 1) ? (T)1 : (T)2;
                o = cmul(cmul(v[brev(n3, 2)] * (T)0.25, to_cx(cP[i])), to_cx(cQ[i])) * wk;
            }
            u[n3] = o;
        }
        dif<4, false>(u);
#pragma unroll
        for (int k3 = 0; k3 < 4; ++k3) {
            const cx e = u[brev(k3, 2)];
            const cx o = k3 == 0 ? e : cmul(e, outer_twiddle<T>(np, k3, g.NFFT, false));
            b[k3 * NS + np] = Prec<T>::make(o.x, o.y);
        }
        return;
    }
#pragma unroll
    for (int n3 = 0; n3 < 4; ++n3) {
        const int64_t i = n3 * NS + np;
        const cx e = v[brev(n3, 2)] * (T)0.25;
        if (MODE == 3) {                                   // plain inverse of an inner level
            b[i] = Prec<T>::make(e.x, e.y);
        } else if (MODE == 0) {
            cx o = {0, 0};
            if (i < g.K) {
                const T wk = (i == 0 || i == g.K - 1) ? (T)1 : (T)2;
                o = cmul(cmul(e, to_cx(cP[i])), to_cx(cQ[i])) * wk;
            }
            b[i] = Prec<T>::make(o.x, o.y);
        } else {
            if (i >= lo && i < lo + len) {
                const cx c = to_cx(cQ[i]);
                czt_out_row(g, y, ldy, row)[i - lo] = (float)((e.x * c.x - e.y * c.y) / (T)g.Q);
            }
        }
    }
}

template <typename T, int MODE>
static void launch_cols_fwd(const CztGeom& g, const float* z, const typename Prec<T>::T2* cP, typename Prec<T>::T2* buf,
                            int64_t rows, hipStream_t st, int64_t ldz = 0, int64_t lo = 0, int64_t len = 0,
                            ChirpSeq cs = ChirpSeq{0, 0, 1, 1.0}) {
    const dim3 grid(TILE_M / 256, (unsigned)(rows * g.S)), blk(256);   // one "row" per sub-transform
#define GFX_CF(CC) case CC: hipLaunchKernelGGL((czt_cols_fwd_kernel<T, CC, MODE>), grid, blk, 0, st, z, cP, buf, g, ldz, lo, len, cs); break;
    switch (g.C) { GFX_CZT_SIZES(GFX_CF) default: break; }
#undef GFX_CF
}

template <typename T, int MODE>
static void launch_cols_inv(const CztGeom& g, typename Prec<T>::T2* buf, const typename Prec<T>::T2* cP,
                            const typename Prec<T>::T2* cQ, float* y, int64_t ldy, int64_t lo, int64_t len, int64_t rows,
                            hipStream_t st) {
    const dim3 grid(TILE_M / 256, (unsigned)(rows * g.S)), blk(256);
#define GFX_CI(CC) case CC: hipLaunchKernelGGL((czt_cols_inv_kernel<T, CC, MODE>), grid, blk, 0, st, buf, cP, cQ, y, ldy, lo, len, g); break;
    switch (g.C) { GFX_CZT_SIZES(GFX_CI) default: break; }
#undef GFX_CI
}

template <typename K>
static bool czt_allow_lds(K kernel, int bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) ==
           hipSuccess;
}

// the double tile's twiddles: TW_ROWS x 256 double2 (rows as in fft_tile.hpp), one table per device
__global__ void czt_twiddle_table_f64_kernel(double2* __restrict__ table) {
    const int t = threadIdx.x, row = blockIdx.x;
    int num;
    double den;
    if (row < 4) { num = t * row; den = 8192.0; }
    else if (row < 12) { num = t * 4 * (row - 4); den = 8192.0; }
    else if (row < 16) { num = (t & 15) * (row - 12); den = 256.0; }
    else { num = (t & 15) * 4 * (row - 16); den = 256.0; }
    double s, c;
    sincospi(2.0 * (double)num / den, &s, &c);
    table[row * TILE_T + t] = make_double2(c, -s);
}

static const double2* tile_twiddle_table_f64(hipStream_t stream) {
    static std::mutex mu;
    static double2* tables[64] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!tables[dev]) {
        double2* p = nullptr;
        if (hipMalloc(&p, sizeof(double2) * TW_ROWS * TILE_T) != hipSuccess) return nullptr;
        hipLaunchKernelGGL(czt_twiddle_table_f64_kernel, dim3(TW_ROWS), dim3(TILE_T), 0, stream, p);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) {
            hipFree(p);
            return nullptr;
        }
        tables[dev] = p;
    }
    return tables[dev];
}

template <typename T> static const typename Prec<T>::T2* czt_twiddles(hipStream_t st);
template <> const float2* czt_twiddles<float>(hipStream_t st) { return tile_twiddle_table(st); }
template <> const double2* czt_twiddles<double>(hipStream_t st) { return tile_twiddle_table_f64(st); }

// The levels below the outermost one and the column pass: every 4^lvl-th part of the buffer is its own transform
template <typename T>
static void czt_inner_fwd(const CztGeom& g, typename Prec<T>::T2* buf, int64_t rows, hipStream_t st) {
    using T2 = typename Prec<T>::T2;
    const ChirpSeq none{0, 0, 1, 1.0};
    for (int lvl = 1; lvl < g.levels; ++lvl) {
        CztGeom gl = g;
        gl.NFFT = g.NFFT >> (2 * lvl);
        const dim3 grid((unsigned)(gl.NFFT / 4 / 256), (unsigned)(rows << (2 * lvl)));
        hipLaunchKernelGGL((czt_outer_fwd_kernel<T, 1>), grid, dim3(256), 0, st, (const float*)nullptr, (const T2*)nullptr, buf,
                           gl, (int64_t)0, (int64_t)0, (int64_t)0, none);
    }
    launch_cols_fwd<T, 1>(g, nullptr, nullptr, buf, rows, st);
}

template <typename T>
static void czt_inner_inv(const CztGeom& g, typename Prec<T>::T2* buf, const typename Prec<T>::T2* cP,
                          const typename Prec<T>::T2* cQ, int64_t rows, hipStream_t st) {
    launch_cols_inv<T, 2>(g, buf, cP, cQ, nullptr, 0, 0, 0, rows, st);
    for (int lvl = g.levels - 1; lvl >= 1; --lvl) {
        CztGeom gl = g;
        gl.NFFT = g.NFFT >> (2 * lvl);
        const dim3 grid((unsigned)(gl.NFFT / 4 / 256), (unsigned)(rows << (2 * lvl)));
        hipLaunchKernelGGL((czt_outer_inv_kernel<T, 3>), grid, dim3(256), 0, st, buf, cP, cQ, (float*)nullptr, (int64_t)0,
                           (int64_t)0, (int64_t)0, gl);
    }
}

template <typename T>
static int czt_plan(void* plan, int64_t P, void* ws, size_t ws_bytes, hipStream_t st) {
    using T2 = typename Prec<T>::T2;
    CztGeom g;
    if (!plan || !czt_geom(P, g)) return GFX_EINVAL;
    if (!ws || ws_bytes < (size_t)g.NFFT * sizeof(T2)) return GFX_ENOSPC;
    const T2* tw = czt_twiddles<T>(st);
    if (!tw || !czt_allow_lds(czt_rows_kernel<T, true>, Prec<T>::lds_bytes)) return GFX_ELAUNCH;
    T2* cP = (T2*)plan;
    T2* cQ = cP + g.P;
    T2* spec = cQ + g.Q;
    hipLaunchKernelGGL(czt_chirp_table_kernel<T>, dim3((unsigned)((g.P + 255) / 256)), dim3(256), 0, st, cP, g.P, g.P, -1.0f);
    hipLaunchKernelGGL(czt_chirp_table_kernel<T>, dim3((unsigned)((g.Q + 255) / 256)), dim3(256), 0, st, cQ, g.Q, g.Q, 1.0f);
    T2* buf = (T2*)ws;
    const int ctot = g.S * g.C;
    const dim3 og((unsigned)(g.NFFT / 4 / 256), 1);
    // forward: bP over [-(P-1), K-1], bQ over [-(K-1), Q-1];  adjoint: bQ over [-(Q-1), K-1], bP over [-(K-1), P-1]
    const ChirpSeq seqs[4] = {{g.P - 1, g.K - 1, g.P, 1.0}, {g.K - 1, g.Q - 1, g.Q, -1.0},
                              {g.Q - 1, g.K - 1, g.Q, -1.0}, {g.K - 1, g.P - 1, g.P, 1.0}};
    for (int i = 0; i < 4; ++i) {
        if (g.levels == 0) launch_cols_fwd<T, 2>(g, nullptr, nullptr, buf, 1, st, 0, 0, 0, seqs[i]);
        else {
            hipLaunchKernelGGL((czt_outer_fwd_kernel<T, 2>), og, dim3(256), 0, st, (const float*)nullptr, (const T2*)nullptr,
                               buf, g, (int64_t)0, (int64_t)0, (int64_t)0, seqs[i]);
            czt_inner_fwd<T>(g, buf, 1, st);
        }
        hipLaunchKernelGGL((czt_rows_kernel<T, true>), dim3((unsigned)ctot), dim3(TILE_T), Prec<T>::lds_bytes, st, buf,
                           (const T2*)nullptr, spec + (int64_t)i * g.NFFT, ctot, tw);
    }
    return hipGetLastError() == hipSuccess ? GFX_OK : GFX_ELAUNCH;
}

// The two chirp-z transforms of one direction: rows of `in` (slice [ilo, ilo + ilen) of the first grid, times tab1) ->
// convolution with spec1 -> times cP cQ w_k on the K bins -> convolution with spec2 -> Re(. tab2) / Q into the slice
// [olo, olo + olen) of the second grid.
template <typename T>
static int czt_run(const CztGeom& g, const float* in, int64_t ldi, int64_t ilo, int64_t ilen,
                   const typename Prec<T>::T2* tab1, const typename Prec<T>::T2* spec1, const typename Prec<T>::T2* spec2,
                   const typename Prec<T>::T2* tab2, float* out, int64_t ldo, int64_t olo, int64_t olen, int64_t rows,
                   const typename Prec<T>::T2* cP, const typename Prec<T>::T2* cQ, typename Prec<T>::T2* buf, hipStream_t st) {
    using T2 = typename Prec<T>::T2;
    const T2* tw = czt_twiddles<T>(st);
    constexpr int LDS = Prec<T>::lds_bytes;
    if (!tw || !czt_allow_lds(czt_rows_kernel<T, false>, LDS)) return GFX_ELAUNCH;
    const int ctot = g.S * g.C;
    const unsigned tiles = (unsigned)(rows * ctot);
    const ChirpSeq none{0, 0, 1, 1.0};
    if (g.levels == 0) {
        launch_cols_fwd<T, 0>(g, in, tab1, buf, rows, st, ldi, ilo, ilen);
        hipLaunchKernelGGL((czt_rows_kernel<T, false>), dim3(tiles), dim3(TILE_T), LDS, st, buf, spec1, (T2*)nullptr, ctot, tw);
        launch_cols_inv<T, 3>(g, buf, cP, cQ, nullptr, 0, 0, 0, rows, st);
        hipLaunchKernelGGL((czt_rows_kernel<T, false>), dim3(tiles), dim3(TILE_T), LDS, st, buf, spec2, (T2*)nullptr, ctot, tw);
        launch_cols_inv<T, 1>(g, buf, cP, tab2, out, ldo, olo, olen, rows, st);
        return hipGetLastError() == hipSuccess ? GFX_OK : GFX_ELAUNCH;
    }
    // outer radix-4 levels around 4^levels sub-transforms of C x 8192 points per row
    const dim3 og((unsigned)(g.NFFT / 4 / 256), (unsigned)rows);
    hipLaunchKernelGGL((czt_outer_fwd_kernel<T, 0>), og, dim3(256), 0, st, in, tab1, buf, g, ldi, ilo, ilen, none);
    czt_inner_fwd<T>(g, buf, rows, st);
    hipLaunchKernelGGL((czt_rows_kernel<T, false>), dim3(tiles), dim3(TILE_T), LDS, st, buf, spec1, (T2*)nullptr, ctot, tw);
    czt_inner_inv<T>(g, buf, cP, cQ, rows, st);
    hipLaunchKernelGGL((czt_outer_inv_kernel<T, 2>), og, dim3(256), 0, st, buf, cP, cQ, (float*)nullptr, (int64_t)0,
                       (int64_t)0, (int64_t)0, g);
    czt_inner_fwd<T>(g, buf, rows, st);
    hipLaunchKernelGGL((czt_rows_kernel<T, false>), dim3(tiles), dim3(TILE_T), LDS, st, buf, spec2, (T2*)nullptr, ctot, tw);
    czt_inner_inv<T>(g, buf, cP, cQ, rows, st);
    hipLaunchKernelGGL((czt_outer_inv_kernel<T, 1>), og, dim3(256), 0, st, buf, cP, tab2, out, ldo, olo, olen, g);
    return hipGetLastError() == hipSuccess ? GFX_OK : GFX_ELAUNCH;
}

template <typename T>
static int czt_alias(const float* z, float* y, int64_t ldy, int64_t lo, int64_t len, int64_t rows, int64_t P,
                     const void* plan, void* ws, size_t ws_bytes, void* stream, bool adjoint,
                     const gfx_rowmap_t* ymap = nullptr, int yC = 0, int64_t row0 = 0) {
    using T2 = typename Prec<T>::T2;
    CztGeom g;
    if (!z || !y || !plan || rows <= 0 || rows > 16383 || !czt_geom(P, g)) return GFX_EINVAL;
    if (ymap) {
        if (adjoint || yC < 1 || row0 < 0 || ymap->inner <= 0 || ymap->inner > 0x7fffffffLL || (row0 + rows) / yC > 0x7fffffffLL)
            return GFX_EINVAL;
        g.yC = yC;
        g.ymap = *ymap;
    }
    if (lo < 0 || len < 1 || lo + len > g.Q || ldy < len) return GFX_EINVAL;
    if (!ws || ws_bytes < (size_t)rows * g.NFFT * sizeof(T2)) return GFX_ENOSPC;
    const T2* cP = (const T2*)plan;
    const T2* cQ = cP + g.P;
    const T2* spec = cQ + g.Q;
    const int64_t step = 65535 / g.S;        // rows * S sub-transforms ride on a grid dimension
    for (int64_t r0 = 0; r0 < rows; r0 += step) {
        const int64_t n = rows - r0 < step ? rows - r0 : step;
        T2* buf = (T2*)ws + r0 * g.NFFT;
        int rc;
        g.row0 = row0 + r0;
        if (!adjoint)   // z (rows x P) -> y[:, lo : lo + len]
            rc = czt_run<T>(g, z + r0 * g.P, g.P, 0, g.P, cP, spec, spec + g.NFFT, cQ, ymap ? y : y + r0 * ldy, ldy, lo, len, n, cP,
                            cQ, buf, (hipStream_t)stream);
        else   // `z` is the gradient gy (row stride ldy) of y[:, lo : lo + len], `y` the gradient gz (rows x P)
            rc = czt_run<T>(g, z + r0 * ldy, ldy, lo, len, cQ, spec + 2 * g.NFFT, spec + 3 * g.NFFT, cP, y + r0 * g.P, g.P, 0,
                            g.P, n, cP, cQ, buf, (hipStream_t)stream);
        if (rc != GFX_OK) return rc;
    }
    return GFX_OK;
}

}  // namespace gfx

using namespace gfx;

extern "C" {

size_t gfx_odd_alias_plan_bytes(int64_t P) {
    CztGeom g;
    return czt_geom(P, g) ? czt_plan_f2(g) * sizeof(float2) : 0;
}

size_t gfx_odd_alias_workspace_bytes(int64_t rows, int64_t P) {
    CztGeom g;
    if (rows <= 0 || !czt_geom(P, g)) return 0;
    return (size_t)rows * g.NFFT * sizeof(float2);
}

int gfx_odd_alias_plan_f32(void* plan, int64_t P, void* ws, size_t ws_bytes, void* stream) {
    return czt_plan<float>(plan, P, ws, ws_bytes, (hipStream_t)stream);
}

int gfx_odd_alias_f32(const float* z, float* y, int64_t ldy, int64_t lo, int64_t len, int64_t rows, int64_t P,
                      const void* plan, void* ws, size_t ws_bytes, void* stream) {
    return czt_alias<float>(z, y, ldy, lo, len, rows, P, plan, ws, ws_bytes, stream, false);
}

int gfx_odd_alias_rows_f32(const float* z, float* y, gfx_rowmap_t ymap, int64_t C, int64_t row0, int64_t lo, int64_t len,
                           int64_t rows, int64_t P, const void* plan, void* ws, size_t ws_bytes, void* stream) {
    if (C < 1 || C > 0x7fffffffLL) return GFX_EINVAL;
    return czt_alias<float>(z, y, len, lo, len, rows, P, plan, ws, ws_bytes, stream, false, &ymap, (int)C, row0);
}

// Transpose of gfx_odd_alias_f32 (the gradient of the aliasing step): with G'[k] = sum_n gy[n] e^{+2 pi i k n / Q},
//   gz[m] = Re( sum_{k < K} w_k G'[k] e^{-2 pi i k m / P} ) / Q
// i.e. the same two chirp-z transforms in the other order -- cQ / bQ first (Q inputs, K bins), cP / bP second (P
// outputs) -- with the chirps' supports mirrored (the plan's third and fourth spectrum).
int gfx_odd_alias_adjoint_f32(const float* gy, int64_t ldg, int64_t lo, int64_t len, float* gz, int64_t rows, int64_t P,
                              const void* plan, void* ws, size_t ws_bytes, void* stream) {
    return czt_alias<float>(gy, gz, ldg, lo, len, rows, P, plan, ws, ws_bytes, stream, true);
}

// ---- the same with double-precision transforms (fp32 in and out): plan and workspace are twice the size ------------
size_t gfx_odd_alias_precise_plan_bytes(int64_t P) { return 2 * gfx_odd_alias_plan_bytes(P); }

size_t gfx_odd_alias_precise_workspace_bytes(int64_t rows, int64_t P) { return 2 * gfx_odd_alias_workspace_bytes(rows, P); }

int gfx_odd_alias_precise_plan_f32(void* plan, int64_t P, void* ws, size_t ws_bytes, void* stream) {
    return czt_plan<double>(plan, P, ws, ws_bytes, (hipStream_t)stream);
}

int gfx_odd_alias_precise_f32(const float* z, float* y, int64_t ldy, int64_t lo, int64_t len, int64_t rows, int64_t P,
                              const void* plan, void* ws, size_t ws_bytes, void* stream) {
    return czt_alias<double>(z, y, ldy, lo, len, rows, P, plan, ws, ws_bytes, stream, false);
}

int gfx_odd_alias_precise_adjoint_f32(const float* gy, int64_t ldg, int64_t lo, int64_t len, float* gz, int64_t rows,
                                      int64_t P, const void* plan, void* ws, size_t ws_bytes, void* stream) {
    return czt_alias<double>(gy, gz, ldg, lo, len, rows, P, plan, ws, ws_bytes, stream, true);
}

}  // extern "C"
