// The odd-length aliasing y = irfft_{P-1}(rfft_P(z)) (czt.hip) for TWO real rows per complex chirp-z transform.
//
// The aliasing A is a real-linear map of a row, and the transforms of czt.hip are complex anyway (Bluestein), so the
// pair z1 + i z2 goes through as ONE complex row and comes out as A z1 + i A z2 -- provided the spectrum is kept on both
// sides (a real row's one-sided bins k < K with weights 1, 2, ..., 2, 1 rely on Z[-k] = conj Z[k], which a complex row
// does not have):
//
//   Z[k'] = cP[|k'|] sum_m (z[m] cP[m]) bP[k' - m],          k' = -(K-1) .. K-1  (P = 2K - 1 bins),  z = z1 + i z2
//   v[n]  = cQ[n] sum_k' (w_k' Z[k'] cQ[|k'|]) bQ[n - k'] / Q,   w = 1/2 at |k'| = K - 1 (the Nyquist bin of the Q grid), else 1
//   y1 = Re v,  y2 = Im v
//
// (chirps as in czt.hip; reference: core/convolution.py:119-134, rfft(n=P) ... irfft() without n).  With the bins stored
// at position k' + (K - 1), the first convolution takes its input at positions m + (K - 1) and the second delivers y[n] at
// position n + (K - 1): both chirp kernels span 2P - 1 points and NFFT = C x 8192 >= 2P - 1 is enough (no wrapped term
// reaches a position that is read).  Against one transform per row that is 2P - 1 points per PAIR instead of
// 2 x (3P - 1) / 2: a third fewer bytes through every pass (P = 135 071: 35 tiles per pair against 2 x 25), for the same
// five passes.  Up to 63 tiles in one column pass (P <= 258 048: 5.4 s of audio at 48 kHz); longer rows take outer radix-4
// levels around czt.hip's column passes of up to 32 tiles, as there (2P - 1 <= 2^24: P <= 8 388 607); the adjoint stays on
// czt.hip.  A transform's rounding error is eps times its LARGER component, so the rows of a pair are brought to one binade
// first: czt_rowmax_kernel takes max |z| of every row, the first column pass multiplies the second row by the power of
// two 2^(e1 - e2) (exact), the last one divides it out again (exact), and a row that is all zero comes out all zero.
// Every row keeps an error relative to ITS OWN peak, as the reference's independent rows do; a pair of equally loud rows
// (same binade) is bit-identical to the unscaled pair.  (GRAFX_ALIAS_PAIR_SCALE=0: no scaling, round 5's behaviour.)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "czt_core.hpp"

namespace gfx {

constexpr int CZT_PAIR_MAXC = 63;
#ifdef GFX_PAIR_DEV_SIZES
#define GFX_CZT_PAIR_SIZES(X) X(2) X(3) X(35) X(36) X(48)
#else
#define GFX_CZT_PAIR_SIZES(X) GFX_CZT_SIZES(X) X(35) X(36) X(40) X(42) X(45) X(48) X(49) X(50) X(54) X(56) X(60) X(63)
#endif

// the middle pass keeps two columns of C points in registers when fused with the next column pass: above 48 tiles (36 in
// double precision) that no longer fits two waves per SIMD and it goes in two passes instead (one more sweep over the buffer)
template <typename T, int C> constexpr bool pair_mid_fused() { return sizeof(T) == 4 ? C <= 48 : C <= 36; }

static inline bool czt_pair_geom(int64_t P, CztGeom& g) {
    if (P < 3 || (P & 1) == 0) return false;
    g.yC = 0;
    g.row0 = 0;
    g.rmax = nullptr;
    g.relu = 0;
    g.ymap = gfx_rowmap_t{1, 0, 0, 0};
    g.P = P;
    g.Q = P - 1;
    g.K = (P + 1) / 2;
    const int64_t tiles = (2 * P - 1 + TILE_M - 1) / TILE_M;
    // up to 63 tiles in one column pass; beyond, outer radix-4 levels around sub-transforms of C <= 32 tiles as in czt.hip
    // (the column kernels of that size range are czt.hip's)
    g.levels = 0;
    int64_t per = tiles;
    if (tiles > CZT_PAIR_MAXC)
        while (per > CZT_MAXC) {
            per = (per + 3) / 4;
            ++g.levels;
        }
    if (g.levels > CZT_MAX_LEVELS) return false;
    int C = (int)per;
    while (!sd_supported(C) || C == 64) ++C;      // (63 = 3 * 3 * 7 and 32 are supported: the loop ends at or below them)
    g.S = 1 << (2 * g.levels);
    g.C = C;
    g.NFFT = (int64_t)g.S * C * TILE_M;
    return true;
}

// plan layout (T2 units): cP[P] | cQ[Q] | mid[P] | spectrum of bP over [-(P + K - 2), K - 1] | spectrum of bQ over
// [-(K - 1), Q + K - 2];  mid[i] = cP[|k'|] cQ[|k'|] w_k' for the bin k' = i - (K - 1) stored at position i (times
// 1 / NFFT when the transform has no outer levels: its inverse column pass leaves the scaling to this table)
// the workspace's tail: one word per row (max |z|), rounded up to 256 bytes
static inline size_t pair_rmax_bytes(int64_t rows) { return ((size_t)rows * sizeof(uint32_t) + 255) & ~(size_t)255; }
static inline size_t czt_pair_plan_t2(const CztGeom& g) { return (size_t)(g.P + g.Q + g.P + 2 * g.NFFT); }

template <typename T>
__global__ void czt_pair_mid_table_kernel(typename Prec<T>::T2* __restrict__ mid, int64_t P, double scale) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const int64_t off = (P - 1) / 2, kp = i - off, a = kp < 0 ? -kp : kp;
    const double2 p = chirp_d<double>(a, P, -1.0), q = chirp_d<double>(a, P - 1, 1.0);
    const double w = (a == off ? 0.5 : 1.0) * scale;
    mid[i] = Prec<T>::make((T)((p.x * q.x - p.y * q.y) * w), (T)((p.x * q.y + p.y * q.x) * w));
}

// the instruction scheduler may not move anything across this point: keeps the loads and twiddles of the NEXT phase of a
// column kernel from being hoisted above the C-point transform, where they would be live on top of the column itself
__device__ __forceinline__ void pair_sched_fence() { __builtin_amdgcn_sched_barrier(0); }


// ---- per-row scaling of a pair -------------------------------------------------------------------------------------
// bits of max |z[r, :]| (non-negative floats order like their bit patterns; a NaN sorts above everything)
constexpr int RMAX_SPAN = 8192;     // samples of a row per workgroup: eight 16-byte loads per thread, all in flight
typedef float rmax_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t rmax_abs(float x) { return __float_as_uint(x) & 0x7fffffffu; }
template <typename T>     // (T: one instance per translation unit that includes this file)
__global__ __launch_bounds__(256) void czt_rowmax_kernel(const float* __restrict__ z, int64_t P, uint32_t* __restrict__ rmax) {
    const float* row = z + (int64_t)blockIdx.y * P;
    // rows of odd length start at any multiple of four bytes: quads are taken from the 16-byte boundary below the row,
    // the first and the last quad of a row element by element
    const int mis = (int)(((uintptr_t)row >> 2) & 3);
    const rmax_f4* quads = reinterpret_cast<const rmax_f4*>(row - mis);
    const int64_t nq = (P + mis + 3) >> 2, q0 = (int64_t)blockIdx.x * (RMAX_SPAN / 4) + threadIdx.x;
    rmax_f4 v[RMAX_SPAN / 1024];
#pragma unroll
    for (int j = 0; j < RMAX_SPAN / 1024; ++j) {
        const int64_t q = q0 + j * 256;
        v[j] = rmax_f4{0.f, 0.f, 0.f, 0.f};
        if (q > 0 && q < nq - 1) {
            v[j] = quads[q];
        } else if (q < nq) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int64_t i = 4 * q - mis + c;
                if (i >= 0 && i < P) v[j][c] = row[i];
            }
        }
    }
    uint32_t m = 0;
#pragma unroll
    for (int j = 0; j < RMAX_SPAN / 1024; ++j) {
        const uint32_t a = rmax_abs(v[j][0]), b = rmax_abs(v[j][1]), c = rmax_abs(v[j][2]), d = rmax_abs(v[j][3]);
        const uint32_t ab = a > b ? a : b, cd = c > d ? c : d, e = ab > cd ? ab : cd;
        m = e > m ? e : m;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const uint32_t t = (uint32_t)__shfl_xor((int)m, o);
        m = t > m ? t : m;
    }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(rmax + blockIdx.y, m);
}

struct PairScale {
    int d;            // the second row goes in as z2 2^d and comes out as y2 2^-d
    bool nz1, nz2;    // a row that is all zero comes out all zero
};
__device__ __forceinline__ PairScale pair_scale(const CztGeom& g, int64_t pr, bool two) {
    PairScale s{0, true, true};
    if (g.rmax) {
        const uint32_t m1 = g.rmax[2 * pr], m2 = two ? g.rmax[2 * pr + 1] : 0u;     // (uniform: scalar loads)
        s.nz1 = m1 != 0;
        s.nz2 = m2 != 0;
        int e1, e2;
        (void)frexpf(__uint_as_float(m1), &e1);
        (void)frexpf(__uint_as_float(m2), &e2);
        // (a row holding an infinity or a NaN poisons its partner whatever the scale: left alone)
        s.d = (m1 != 0 && m2 != 0 && m1 < 0x7f800000u && m2 < 0x7f800000u) ? e1 - e2 : 0;
    }
    return s;
}
__device__ __forceinline__ float pair_out(const CztGeom& g, float v) { return g.relu ? fmaxf(v, 0.0f) : v; }
template <typename T> __device__ __forceinline__ T pair_ldexp(T x, int d) {
    if constexpr (sizeof(T) == 4) return ldexpf(x, d); else return ldexp(x, d);
}

// first column pass: (z1[m] + i z2[m] 2^d) cP[m] at position m + K - 1, zero elsewhere
template <typename T, int C>
__global__ __launch_bounds__(256) void czt_pair_in_kernel(const float* __restrict__ z, const typename Prec<T>::T2* __restrict__ cP,
                                                         typename Prec<T>::T2* __restrict__ buf, CztGeom g, int64_t rows) {
    using cx = typename Prec<T>::cxt;
    const int n2 = blockIdx.x * 256 + threadIdx.x;
    const int64_t pr = blockIdx.y;
    const ColBuf<T> cb(buf + pr * g.NFFT, g.NFFT);
    const int64_t off = g.K - 1;
    const bool two = 2 * pr + 1 < rows;
    const float* z1 = z + 2 * pr * g.P;
    const float* z2 = z1 + (two ? g.P : 0);
    const PairScale ps = pair_scale(g, pr, two);
    cx v[C];
#pragma unroll
    for (int n1 = 0; n1 < C; ++n1) {
        const int64_t m = (int64_t)n1 * TILE_M + n2 - off;
        cx e = {0, 0};
        if (m >= 0 && m < g.P) {
            const cx c = to_cx(cP[m]);
            const T a = (T)z1[m], bb = two ? pair_ldexp((T)z2[m], ps.d) : (T)0;
            e = cx{c.x * a - c.y * bb, c.x * bb + c.y * a};
        }
        v[n1] = e;
    }
    col_dft<C, false>(v);
    ColTw<T, C> tw(n2, (int)g.NFFT, false);
#pragma unroll
    for (int k1 = 0; k1 < C; ++k1) {
        const cx e = v[spos(C, k1)];
        const cx o = k1 == 0 ? e : cmul(e, tw.at(k1));
        cb.st(k1, n2, o);
    }
}

// (launch bounds of the two kernels below, measured at 4096 rows, tools/alias_bench.py: in float two workgroups per CU with a
// few spilled registers beat one without -- 13.8 against 15.6 ms at P = 135 071; in double it is the other way round, 14.6
// against 15.3 ms per 2048 rows, and 20.0 against 27.6 at 48 tiles)
// between the convolutions: the inverse column pass of the first, the bins' factors mid[i] (zero beyond the P bins), and
// -- FUSED -- the forward column pass of the second
template <typename T, int C, bool FUSED>
__global__ __launch_bounds__(256, sizeof(T) == 4 ? 2 : 1) void czt_pair_mid_kernel(typename Prec<T>::T2* __restrict__ buf,
                                                          const typename Prec<T>::T2* __restrict__ mid, CztGeom g) {
    using cx = typename Prec<T>::cxt;
    const int n2 = blockIdx.x * 256 + threadIdx.x;
    const ColBuf<T> cb(buf + (int64_t)blockIdx.y * g.NFFT, g.NFFT);
    cx v[C];
    ColTw<T, C> twi(n2, (int)g.NFFT, true);
#pragma unroll
    for (int k1 = 0; k1 < C; ++k1) {
        const cx e = cb.ld(k1, n2);
        v[k1] = k1 == 0 ? e : cmul(e, twi.at(k1));
    }
    pair_sched_fence();
    col_dft<C, true>(v);
    pair_sched_fence();   // (the factors' loads and the next pass's twiddles stay behind the transform: they would be live across it)
    if constexpr (FUSED) {
        cx u[C];
#pragma unroll
        for (int n1 = 0; n1 < C; ++n1) {
            const int64_t i = (int64_t)n1 * TILE_M + n2;
            u[n1] = i < g.P ? cmul(v[spos(C, n1)], to_cx(mid[i])) : cx{0, 0};
        }
        pair_sched_fence();
        col_dft<C, false>(u);
        pair_sched_fence();
        ColTw<T, C> twf(n2, (int)g.NFFT, false);
#pragma unroll
        for (int k1 = 0; k1 < C; ++k1) {
            const cx e = u[spos(C, k1)];
            const cx o = k1 == 0 ? e : cmul(e, twf.at(k1));
            cb.st(k1, n2, o);
        }
    } else {
#pragma unroll
        for (int n1 = 0; n1 < C; ++n1) {
            const int64_t i = (int64_t)n1 * TILE_M + n2;
            const cx o = i < g.P ? cmul(v[spos(C, n1)], to_cx(mid[i])) : cx{0, 0};
            cb.st(n1, n2, o);
        }
    }
}

// last column pass: v[n] = conv[n + K - 1] cQ[n] / (NFFT Q) for lo <= n < lo + len; real part to the pair's first row,
// imaginary part to its second
template <typename T, int C>
__global__ __launch_bounds__(256, sizeof(T) == 4 ? 2 : 1) void czt_pair_out_kernel(const typename Prec<T>::T2* __restrict__ buf,
                                                          const typename Prec<T>::T2* __restrict__ cQ, float* __restrict__ y,
                                                          int64_t ldy, int64_t lo, int64_t len, CztGeom g, int64_t rows) {
    using cx = typename Prec<T>::cxt;
    const int n2 = blockIdx.x * 256 + threadIdx.x;
    const int64_t pr = blockIdx.y;
    const ColBuf<T> cb(buf + pr * g.NFFT, g.NFFT);
    const int64_t off = g.K - 1;
    const bool two = 2 * pr + 1 < rows;
    cx v[C];
    ColTw<T, C> twi(n2, (int)g.NFFT, true);
#pragma unroll
    for (int k1 = 0; k1 < C; ++k1) {
        const cx e = cb.ld(k1, n2);
        v[k1] = k1 == 0 ? e : cmul(e, twi.at(k1));
    }
    pair_sched_fence();
    col_dft<C, true>(v);
    pair_sched_fence();
    const T sc = (T)1 / ((T)g.NFFT * (T)g.Q);
    float* y1 = czt_out_row(g, y, ldy, 2 * pr);
    float* y2 = two ? czt_out_row(g, y, ldy, 2 * pr + 1) : y1;
    const PairScale ps = pair_scale(g, pr, two);
#pragma unroll
    for (int n1 = 0; n1 < C; ++n1) {
        const int64_t n = (int64_t)n1 * TILE_M + n2 - off;
        if (n >= lo && n < lo + len) {
            const cx o = cmul(v[spos(C, n1)], to_cx(cQ[n])) * sc;
            y1[n - lo] = pair_out(g, ps.nz1 ? (float)o.x : 0.0f);
            if (two) y2[n - lo] = pair_out(g, ps.nz2 ? (float)pair_ldexp(o.y, -ps.d) : 0.0f);
        }
    }
}

// ---- transforms with outer radix-4 levels (more than 63 tiles per pair): the outermost level's three passes in pair form;
// the levels below and the column passes are czt.hip's (czt_levels_fwd / czt_levels_inv) ---------------------------------
template <typename T>
__global__ __launch_bounds__(256) void czt_pair_outer_in_kernel(const float* __restrict__ z, const typename Prec<T>::T2* __restrict__ cP,
                                                               typename Prec<T>::T2* __restrict__ buf, CztGeom g, int64_t rows) {
    using cx = typename Prec<T>::cxt;
    const int64_t NS = g.NFFT / 4;
    const int64_t np = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t pr = blockIdx.y;
    typename Prec<T>::T2* b = buf + pr * g.NFFT;
    const int64_t off = g.K - 1;
    const bool two = 2 * pr + 1 < rows;
    const float* z1 = z + 2 * pr * g.P;
    const float* z2 = z1 + (two ? g.P : 0);
    const PairScale ps = pair_scale(g, pr, two);
    cx v[4];
#pragma unroll
    for (int n3 = 0; n3 < 4; ++n3) {
        const int64_t m = n3 * NS + np - off;
        cx e = {0, 0};
        if (m >= 0 && m < g.P) {
            const cx c = to_cx(cP[m]);
            const T a = (T)z1[m], bb = two ? pair_ldexp((T)z2[m], ps.d) : (T)0;
            e = cx{c.x * a - c.y * bb, c.x * bb + c.y * a};
        }
        v[n3] = e;
    }
    dif<4, false>(v);
#pragma unroll
    for (int k3 = 0; k3 < 4; ++k3) {
        const cx e = v[brev(k3, 2)];
        const cx o = k3 == 0 ? e : cmul(e, outer_twiddle<T>(np, k3, g.NFFT, false));
        buf_store<T>(&b[k3 * NS + np], o);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void czt_pair_outer_mid_kernel(typename Prec<T>::T2* __restrict__ buf,
                                                                const typename Prec<T>::T2* __restrict__ mid, CztGeom g) {
    using cx = typename Prec<T>::cxt;
    const int64_t NS = g.NFFT / 4;
    const int64_t np = (int64_t)blockIdx.x * 256 + threadIdx.x;
    typename Prec<T>::T2* b = buf + (int64_t)blockIdx.y * g.NFFT;
    cx v[4], u[4];
#pragma unroll
    for (int k3 = 0; k3 < 4; ++k3) {
        const cx e = buf_load<T>(&b[k3 * NS + np]);
        v[k3] = k3 == 0 ? e : cmul(e, outer_twiddle<T>(np, k3, g.NFFT, true));
    }
    dif<4, true>(v);
#pragma unroll
    for (int n3 = 0; n3 < 4; ++n3) {
        const int64_t i = n3 * NS + np;
        u[n3] = i < g.P ? cmul(v[brev(n3, 2)] * (T)0.25, to_cx(mid[i])) : cx{0, 0};
    }
    dif<4, false>(u);
#pragma unroll
    for (int k3 = 0; k3 < 4; ++k3) {
        const cx e = u[brev(k3, 2)];
        const cx o = k3 == 0 ? e : cmul(e, outer_twiddle<T>(np, k3, g.NFFT, false));
        buf_store<T>(&b[k3 * NS + np], o);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void czt_pair_outer_out_kernel(const typename Prec<T>::T2* __restrict__ buf,
                                                                const typename Prec<T>::T2* __restrict__ cQ, float* __restrict__ y,
                                                                int64_t ldy, int64_t lo, int64_t len, CztGeom g, int64_t rows) {
    using cx = typename Prec<T>::cxt;
    const int64_t NS = g.NFFT / 4;
    const int64_t np = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t pr = blockIdx.y;
    const typename Prec<T>::T2* b = buf + pr * g.NFFT;
    const int64_t off = g.K - 1;
    const bool two = 2 * pr + 1 < rows;
    cx v[4];
#pragma unroll
    for (int k3 = 0; k3 < 4; ++k3) {
        const cx e = buf_load<T>(&b[k3 * NS + np]);
        v[k3] = k3 == 0 ? e : cmul(e, outer_twiddle<T>(np, k3, g.NFFT, true));
    }
    dif<4, true>(v);
    const T sc = (T)0.25 / (T)g.Q;
    float* y1 = czt_out_row(g, y, ldy, 2 * pr);
    float* y2 = two ? czt_out_row(g, y, ldy, 2 * pr + 1) : y1;
    const PairScale ps = pair_scale(g, pr, two);
#pragma unroll
    for (int n3 = 0; n3 < 4; ++n3) {
        const int64_t n = n3 * NS + np - off;
        if (n >= lo && n < lo + len) {
            const cx o = cmul(v[brev(n3, 2)], to_cx(cQ[n])) * sc;
            y1[n - lo] = pair_out(g, ps.nz1 ? (float)o.x : 0.0f);
            if (two) y2[n - lo] = pair_out(g, ps.nz2 ? (float)pair_ldexp(o.y, -ps.d) : 0.0f);
        }
    }
}

template <typename T>
static void pair_chain_levels(const CztGeom& g, const float* z, float* y, int64_t ldy, int64_t lo, int64_t len, int64_t rows,
                              const typename Prec<T>::T2* cP, const typename Prec<T>::T2* cQ, const typename Prec<T>::T2* mid,
                              const typename Prec<T>::T2* spec, typename Prec<T>::T2* buf, const typename Prec<T>::T2* tw,
                              hipStream_t st) {
    using T2 = typename Prec<T>::T2;
    constexpr int LDS = Prec<T>::lds_bytes;
    const int64_t pairs = (rows + 1) / 2;
    const dim3 og((unsigned)(g.NFFT / 4 / 256), (unsigned)pairs), blk(256);
    const int ctot = g.S * g.C;
    const unsigned tiles = (unsigned)(pairs * ctot);
    hipLaunchKernelGGL((czt_pair_outer_in_kernel<T>), og, blk, 0, st, z, cP, buf, g, rows);
    czt_levels_fwd(g, buf, pairs, st);
    hipLaunchKernelGGL((czt_rows_kernel<T, false>), dim3(tiles), dim3(TILE_T), LDS, st, buf, spec, (T2*)nullptr, ctot, tw);
    czt_levels_inv(g, buf, pairs, st);
    hipLaunchKernelGGL((czt_pair_outer_mid_kernel<T>), og, blk, 0, st, buf, mid, g);
    czt_levels_fwd(g, buf, pairs, st);
    hipLaunchKernelGGL((czt_rows_kernel<T, false>), dim3(tiles), dim3(TILE_T), LDS, st, buf, spec + g.NFFT, (T2*)nullptr, ctot, tw);
    czt_levels_inv(g, buf, pairs, st);
    hipLaunchKernelGGL((czt_pair_outer_out_kernel<T>), og, blk, 0, st, (const T2*)buf, cQ, y, ldy, lo, len, g, rows);
}

// ---- one outer level, fused (float): the outer radix-4 pass and the column pass of its four sub-transforms in ONE kernel ----
// With one outer level a pair's transform is 4 sub-transforms of C x 8192 points; the chain above sweeps the workspace nine
// times (outer in, columns, tiles, columns, outer mid, columns, tiles, columns, outer out).  Here the four waves of a
// workgroup own the same 64 columns n2, wave q playing sub-transform k3 = q in the column passes and quarter n3 = q in the
// radix-4 passes; the values change hands through LDS ([n1][q][column], C x 2 KB).  Five sweeps, as without the level:
//   lv_in : z -> radix-4 over the quarters, twiddle W_NFFT^(n' k3) -> C-point DFT over n1, twiddle -> workspace
//   lv_mid: workspace -> inverse column pass -> inverse radix-4 -> bins' factors -> radix-4 -> column pass -> workspace
//   lv_out: workspace -> inverse column pass -> inverse radix-4 -> chirp, real / imaginary part to the pair's rows
// The outer twiddle factors as W_NFFT^(n2 k3) (one per thread) times W_4C^(n1 k3) (a 4C-entry table in LDS, read at
// wave-uniform addresses): n' = n1 8192 + n2 and NFFT = 4C x 8192.
constexpr int LV_COLS = 64;

template <typename T, int QUARTER> __device__ __forceinline__ typename Prec<T>::cxt lv_rot(typename Prec<T>::cxt t) {
    using cx = typename Prec<T>::cxt;        // t times i^QUARTER: a swap and signs, free once QUARTER is a constant
    constexpr int r = QUARTER & 3;
    return r == 0 ? t : r == 1 ? cx{-t.y, t.x} : r == 2 ? cx{-t.x, -t.y} : cx{t.y, -t.x};
}

template <typename T, int C> struct LvShared {
    using cx = typename Prec<T>::cxt;
    cx* ex;        // [n1][q][col]
    cx* tab;       // e^(-2 pi i j / 4C), j < 4C
    __device__ __forceinline__ LvShared(unsigned char* raw, int exrows = C) : ex(reinterpret_cast<cx*>(raw)), tab(ex + exrows * 4 * LV_COLS) {}
    __device__ __forceinline__ void fill_table(int tid) {
        for (int j = tid; j < 4 * C; j += 256) {
            float sn, cs;
            sincospif(2.0f * (float)j / (float)(4 * C), &sn, &cs);
            tab[j] = cx{(T)cs, (T)(-sn)};
        }
    }
    __device__ __forceinline__ cx& at(int n1, int q, int col) { return ex[(n1 * 4 + q) * LV_COLS + col]; }
};
template <int C> constexpr size_t lv_lds_bytes(int exrows = C) { return (size_t)(exrows * 4 * LV_COLS + 4 * C) * sizeof(cx); }
// the last kernel exchanges its column in two halves of n1 (half the LDS: three workgroups per CU instead of two)
template <int C> constexpr int lv_half() { return (C + 1) / 2; }

// forward radix-4 of the four quarters' values of (n1, column) for sub-transform K3 (the wave's role: a template parameter,
// chosen by a uniform switch -- with a run-time role the rotations are selects and the middle kernel was ALU-bound at
// 2.7 TB/s), with the outer twiddle
template <typename T, int C, int K3>
__device__ __forceinline__ typename Prec<T>::cxt lv_fwd4(LvShared<T, C>& sh, int row, int n1, int col, typename Prec<T>::cxt a0) {
    using cx = typename Prec<T>::cxt;       // (row: where the exchange holds n1's values)
    const cx x = sh.at(row, 0, col) + lv_rot<T, 4 - ((1 * K3) & 3)>(sh.at(row, 1, col)) + lv_rot<T, 4 - ((2 * K3) & 3)>(sh.at(row, 2, col)) +
                 lv_rot<T, 4 - ((3 * K3) & 3)>(sh.at(row, 3, col));                                    // (-i)^(n3 k3)
    return K3 == 0 ? x : cmul(x, cmul(a0, sh.tab[n1 * K3]));
}
// inverse radix-4: the value of quarter N3 from the four sub-transforms' (already conj-twiddled) values
template <typename T, int C, int N3>
__device__ __forceinline__ typename Prec<T>::cxt lv_inv4(LvShared<T, C>& sh, int n1, int col) {
    using cx = typename Prec<T>::cxt;
    const cx x = sh.at(n1, 0, col) + lv_rot<T, (1 * N3) & 3>(sh.at(n1, 1, col)) + lv_rot<T, (2 * N3) & 3>(sh.at(n1, 2, col)) +
                 lv_rot<T, (3 * N3) & 3>(sh.at(n1, 3, col));                                          // i^(n3 k3)
    return x * (T)0.25;
}
// the column of sub-transform K3, conj-twiddled, to LDS for the inverse radix-4 (scaled by sc)
template <typename T, int C, int K3, int A = 0, int B = C>
__device__ __forceinline__ void lv_put_inverse(LvShared<T, C>& sh, const typename Prec<T>::cxt (&v)[C], int col,
                                               typename Prec<T>::cxt a0, T sc) {
    using cx = typename Prec<T>::cxt;       // rows n1 in [A, B), stored at n1 - A
#pragma unroll
    for (int n1 = A; n1 < B; ++n1) {
        const cx e = v[spos(C, n1)] * sc;
        if (K3 == 0) {
            sh.at(n1 - A, 0, col) = e;
        } else {
            const cx w = cmul(a0, sh.tab[n1 * K3]);               // conj of the forward outer twiddle
            sh.at(n1 - A, K3, col) = cmul(e, cx{w.x, -w.y});
        }
    }
}

template <typename T, int C, int K3, int A = 0, int B = C>
__device__ __forceinline__ void lv_take_forward(LvShared<T, C>& sh, typename Prec<T>::cxt (&v)[C], int col, typename Prec<T>::cxt a0) {
#pragma unroll
    for (int n1 = A; n1 < B; ++n1) v[n1] = lv_fwd4<T, C, K3>(sh, n1 - A, n1, col, a0);
}
template <typename T, int C, int N3>
__device__ __forceinline__ void lv_take_inverse(LvShared<T, C>& sh, typename Prec<T>::cxt (&u)[C], int col) {
#pragma unroll
    for (int n1 = 0; n1 < C; ++n1) u[n1] = lv_inv4<T, C, N3>(sh, n1, col);
}

#define GFX_LV_ROLE(q, ...) switch (q) { case 0: { constexpr int Q = 0; __VA_ARGS__; } break; case 1: { constexpr int Q = 1; __VA_ARGS__; } break; \
                                         case 2: { constexpr int Q = 2; __VA_ARGS__; } break; default: { constexpr int Q = 3; __VA_ARGS__; } break; }

template <typename T, int C>
__global__ __launch_bounds__(256) void czt_pair_lv_in_kernel(const float* __restrict__ z, const typename Prec<T>::T2* __restrict__ cP,
                                                            typename Prec<T>::T2* __restrict__ buf, CztGeom g, int64_t rows) {
    using cx = typename Prec<T>::cxt;
    extern __shared__ __attribute__((aligned(16))) unsigned char lv_raw[];
    constexpr int CH = lv_half<C>();
    LvShared<T, C> sh(lv_raw, CH);
    const int tid = threadIdx.x, col = tid & (LV_COLS - 1), q = __builtin_amdgcn_readfirstlane(tid >> 6);   // (the wave's role: scalar)
    const int n2 = blockIdx.x * LV_COLS + col;
    const int64_t pr = blockIdx.y, NS = g.NFFT / 4, off = g.K - 1;
    const ColBuf<T> cb(buf + pr * g.NFFT, g.NFFT);
    const bool two = 2 * pr + 1 < rows;
    const float* z1 = z + 2 * pr * g.P;
    const float* z2 = z1 + (two ? g.P : 0);
    const PairScale ps = pair_scale(g, pr, two);
    sh.fill_table(tid);
    const cx a0 = unit_root_f(n2 * q, (int)g.NFFT, false);      // W_NFFT^(n2 k3), n2 k3 < 3 x 8192 < NFFT
    cx v[C];
    // Per half of n1 (half the LDS: more workgroups per CU).  The input covers two of the four quarters: the (quarter, n1)
    // slots of a column are dealt round the four waves.  All loads of a thread go out first, from clamped addresses (a load
    // inside a per-slot test is a round trip per slot: 2.4 against 1.65 ms for this kernel), the tests are applied to the values
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int A = h * CH, NH = h == 0 ? CH : C - CH;        // rows [A, A + NH) of n1
        cx ce[CH];
        float za[CH], zb[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int slot = 4 * j + q, n3 = slot / NH, n1 = A + slot - n3 * NH;     // (scalar)
            const int64_t m0 = n3 * NS + (int64_t)n1 * TILE_M + blockIdx.x * LV_COLS - off;   // first column's sample
            ce[j] = cx{0, 0};
            za[j] = zb[j] = 0.0f;
            if (slot < 4 * NH && m0 + LV_COLS > 0 && m0 < g.P) {                    // (uniform: the slot meets the input)
                const int64_t m = m0 + col, mm = m < 0 ? 0 : (m >= g.P ? g.P - 1 : m);
                ce[j] = to_cx(cP[mm]);
                za[j] = z1[mm];
                zb[j] = z2[mm];
            }
        }
        if (h) __syncthreads();        // (the first half's values have been taken)
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int slot = 4 * j + q, n3 = slot / NH, n1l = slot - n3 * NH;
            if (slot < 4 * NH) {
                const int64_t m = n3 * NS + (int64_t)(A + n1l) * TILE_M + n2 - off;
                const bool ok = m >= 0 && m < g.P;
                const T a = ok ? (T)za[j] : (T)0, bb = ok && two ? pair_ldexp((T)zb[j], ps.d) : (T)0;
                sh.at(n1l, n3, col) = cx{ce[j].x * a - ce[j].y * bb, ce[j].x * bb + ce[j].y * a};
            }
        }
        __syncthreads();
        if (h == 0) { GFX_LV_ROLE(q, lv_take_forward<T, C, Q, 0, CH>(sh, v, col, a0)) }
        else { GFX_LV_ROLE(q, lv_take_forward<T, C, Q, CH, C>(sh, v, col, a0)) }
    }
    col_dft<C, false>(v);
    ColTw<T, C> tw(n2, (int)NS, false);
#pragma unroll
    for (int k1 = 0; k1 < C; ++k1) {
        const cx e = v[spos(C, k1)];
        cb.st(q * C + k1, n2, k1 == 0 ? e : cmul(e, tw.at(k1)));
    }
}

template <typename T, int C>
__global__ __launch_bounds__(256) void czt_pair_lv_mid_kernel(typename Prec<T>::T2* __restrict__ buf,
                                                             const typename Prec<T>::T2* __restrict__ mid, CztGeom g) {
    using cx = typename Prec<T>::cxt;
    extern __shared__ __attribute__((aligned(16))) unsigned char lv_raw[];
    LvShared<T, C> sh(lv_raw);
    const int tid = threadIdx.x, col = tid & (LV_COLS - 1), q = __builtin_amdgcn_readfirstlane(tid >> 6);   // (the wave's role: scalar)
    const int n2 = blockIdx.x * LV_COLS + col;
    const int64_t NS = g.NFFT / 4;
    const ColBuf<T> cb(buf + (int64_t)blockIdx.y * g.NFFT, g.NFFT);
    sh.fill_table(tid);
    cx v[C];
    {
        ColTw<T, C> twi(n2, (int)NS, true);
#pragma unroll
        for (int k1 = 0; k1 < C; ++k1) {
            const cx e = cb.ld(q * C + k1, n2);
            v[k1] = k1 == 0 ? e : cmul(e, twi.at(k1));
        }
    }
    col_dft<C, true>(v);
    __syncthreads();      // (the table is there)
    const cx a0 = unit_root_f(n2 * q, (int)g.NFFT, false);
    GFX_LV_ROLE(q, lv_put_inverse<T, C, Q>(sh, v, col, a0, (T)1 / (T)NS))
    __syncthreads();
    cx u[C];
    GFX_LV_ROLE(q, lv_take_inverse<T, C, Q>(sh, u, col))
    {
        cx f[C];                                                  // quarter q: the bins' factors (zero beyond the P bins),
#pragma unroll
        for (int n1 = 0; n1 < C; ++n1) {                          // loaded from clamped addresses, all at once
            const int64_t i = q * NS + (int64_t)n1 * TILE_M + n2;
            f[n1] = to_cx(mid[i < g.P ? i : g.P - 1]);
        }
#pragma unroll
        for (int n1 = 0; n1 < C; ++n1) {
            const int64_t i = q * NS + (int64_t)n1 * TILE_M + n2;
            u[n1] = i < g.P ? cmul(u[n1], f[n1]) : cx{0, 0};
        }
    }
    __syncthreads();
#pragma unroll
    for (int n1 = 0; n1 < C; ++n1) sh.at(n1, q, col) = u[n1];
    __syncthreads();
    GFX_LV_ROLE(q, lv_take_forward<T, C, Q>(sh, v, col, a0))
    col_dft<C, false>(v);
    ColTw<T, C> twf(n2, (int)NS, false);
#pragma unroll
    for (int k1 = 0; k1 < C; ++k1) {
        const cx e = v[spos(C, k1)];
        cb.st(q * C + k1, n2, k1 == 0 ? e : cmul(e, twf.at(k1)));
    }
}

template <typename T, int C>
__global__ __launch_bounds__(256) void czt_pair_lv_out_kernel(const typename Prec<T>::T2* __restrict__ buf,
                                                             const typename Prec<T>::T2* __restrict__ cQ, float* __restrict__ y,
                                                             int64_t ldy, int64_t lo, int64_t len, CztGeom g, int64_t rows) {
    using cx = typename Prec<T>::cxt;
    extern __shared__ __attribute__((aligned(16))) unsigned char lv_raw[];
    constexpr int CH = lv_half<C>();
    LvShared<T, C> sh(lv_raw, CH);
    const int tid = threadIdx.x, col = tid & (LV_COLS - 1), q = __builtin_amdgcn_readfirstlane(tid >> 6);   // (the wave's role: scalar)
    const int n2 = blockIdx.x * LV_COLS + col;
    const int64_t pr = blockIdx.y, NS = g.NFFT / 4, off = g.K - 1;
    const ColBuf<T> cb(buf + pr * g.NFFT, g.NFFT);
    const bool two = 2 * pr + 1 < rows;
    const PairScale ps = pair_scale(g, pr, two);
    sh.fill_table(tid);
    cx v[C];
    {
        ColTw<T, C> twi(n2, (int)NS, true);
#pragma unroll
        for (int k1 = 0; k1 < C; ++k1) {
            const cx e = cb.ld(q * C + k1, n2);
            v[k1] = k1 == 0 ? e : cmul(e, twi.at(k1));
        }
    }
    col_dft<C, true>(v);
    __syncthreads();
    const cx a0 = unit_root_f(n2 * q, (int)g.NFFT, false);
    const T sc = (T)1 / ((T)NS * (T)g.Q);
    float* y1 = czt_out_row(g, y, ldy, 2 * pr);
    float* y2 = two ? czt_out_row(g, y, ldy, 2 * pr + 1) : y1;
    // Per half of n1: the column to LDS, then -- the output slice covers about two of the four quarters -- the 4 x CH
    // (quarter, n1) slots dealt round the four waves, the chirp's loads out together from clamped addresses, products, stores
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        constexpr int dummy = 0;
        (void)dummy;
        const int A = h * CH, NH = h == 0 ? CH : C - CH;        // rows [A, A + NH) of n1
        if (h) __syncthreads();
        if (h == 0) { GFX_LV_ROLE(q, lv_put_inverse<T, C, Q, 0, CH>(sh, v, col, a0, sc)) }
        else { GFX_LV_ROLE(q, lv_put_inverse<T, C, Q, CH, C>(sh, v, col, a0, sc)) }
        __syncthreads();
        cx cq[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int slot = 4 * j + q, n3 = slot / NH, n1 = A + slot - n3 * NH;     // (scalar)
            const int64_t n0 = n3 * NS + (int64_t)n1 * TILE_M + blockIdx.x * LV_COLS - off;
            cq[j] = cx{0, 0};
            if (slot < 4 * NH && n0 + LV_COLS > lo && n0 < lo + len) {               // (uniform: the slot meets the slice)
                const int64_t n = n0 + col;
                cq[j] = to_cx(cQ[n < 0 ? 0 : (n >= g.Q ? g.Q - 1 : n)]);
            }
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int slot = 4 * j + q, n3 = slot / NH, n1l = slot - n3 * NH;
            const int64_t n0 = n3 * NS + (int64_t)(A + n1l) * TILE_M + blockIdx.x * LV_COLS - off;
            if (slot < 4 * NH && n0 + LV_COLS > lo && n0 < lo + len) {
                cx x;
                GFX_LV_ROLE(n3, x = lv_inv4<T, C, Q>(sh, n1l, col))
                const int64_t n = n0 + col;
                if (n >= lo && n < lo + len) {
                    const cx o = cmul(x, cq[j]);
                    y1[n - lo] = pair_out(g, ps.nz1 ? (float)o.x : 0.0f);
                    if (two) y2[n - lo] = pair_out(g, ps.nz2 ? (float)pair_ldexp(o.y, -ps.d) : 0.0f);
                }
            }
        }
    }
}

template <typename T, int CC>
static int pair_chain_one_level(const CztGeom& g, const float* z, float* y, int64_t ldy, int64_t lo, int64_t len, int64_t rows,
                                const typename Prec<T>::T2* cP, const typename Prec<T>::T2* cQ, const typename Prec<T>::T2* mid,
                                const typename Prec<T>::T2* spec, typename Prec<T>::T2* buf, const typename Prec<T>::T2* tw,
                                hipStream_t st) {
    using T2 = typename Prec<T>::T2;
    constexpr int LDS = Prec<T>::lds_bytes;
    constexpr size_t LV = lv_lds_bytes<CC>(), LVH = lv_lds_bytes<CC>(lv_half<CC>());
    if (!czt_allow_lds(czt_pair_lv_in_kernel<T, CC>, (int)LVH) || !czt_allow_lds(czt_pair_lv_mid_kernel<T, CC>, (int)LV) ||
        !czt_allow_lds(czt_pair_lv_out_kernel<T, CC>, (int)LVH))
        return GFX_ELAUNCH;
    const int64_t pairs = (rows + 1) / 2;
    const dim3 grid(TILE_M / LV_COLS, (unsigned)pairs), blk(256);
    const int ctot = 4 * CC;
    const unsigned tiles = (unsigned)(pairs * ctot);
    hipLaunchKernelGGL((czt_pair_lv_in_kernel<T, CC>), grid, blk, LVH, st, z, cP, buf, g, rows);
    hipLaunchKernelGGL((czt_rows_kernel<T, false>), dim3(tiles), dim3(TILE_T), LDS, st, buf, spec, (T2*)nullptr, ctot, tw);
    hipLaunchKernelGGL((czt_pair_lv_mid_kernel<T, CC>), grid, blk, LV, st, buf, mid, g);
    hipLaunchKernelGGL((czt_rows_kernel<T, false>), dim3(tiles), dim3(TILE_T), LDS, st, buf, spec + g.NFFT, (T2*)nullptr, ctot, tw);
    hipLaunchKernelGGL((czt_pair_lv_out_kernel<T, CC>), grid, blk, LVH, st, (const T2*)buf, cQ, y, ldy, lo, len, g, rows);
    return GFX_OK;
}

template <typename T, int MODE>
static void pair_cols_fwd(const CztGeom& g, typename Prec<T>::T2* buf, int64_t pairs, hipStream_t st, ChirpSeq cs) {
    using T2 = typename Prec<T>::T2;
    const dim3 grid(TILE_M / 256, (unsigned)pairs), blk(256);
#define GFX_PF(CC) case CC: hipLaunchKernelGGL((czt_cols_fwd_kernel<T, CC, MODE>), grid, blk, 0, st, (const float*)nullptr, (const T2*)nullptr, \
                                               buf, g, (int64_t)0, (int64_t)0, (int64_t)0, cs); break;
    switch (g.C) { GFX_CZT_PAIR_SIZES(GFX_PF) default: break; }
#undef GFX_PF
}

template <typename T>
static int czt_pair_plan(void* plan, int64_t P, void* ws, size_t ws_bytes, hipStream_t st) {
    using T2 = typename Prec<T>::T2;
    CztGeom g;
    if (!plan || !czt_pair_geom(P, g)) return GFX_EINVAL;
    if (!ws || ws_bytes < (size_t)g.NFFT * sizeof(T2)) return GFX_ENOSPC;
    const T2* tw = czt_twiddles<T>(st);
    if (!tw || !czt_allow_lds(czt_rows_kernel<T, true>, Prec<T>::lds_bytes)) return GFX_ELAUNCH;
    T2* cP = (T2*)plan;
    T2* cQ = cP + g.P;
    T2* mid = cQ + g.Q;
    T2* spec = mid + g.P;
    hipLaunchKernelGGL(czt_pair_mid_table_kernel<T>, dim3((unsigned)((g.P + 255) / 256)), dim3(256), 0, st, mid, g.P,
                       g.levels == 0 ? 1.0 / (double)g.NFFT : 1.0);
    hipLaunchKernelGGL(czt_chirp_table_kernel<T>, dim3((unsigned)((g.P + 255) / 256)), dim3(256), 0, st, cP, g.P, g.P, -1.0f);
    hipLaunchKernelGGL(czt_chirp_table_kernel<T>, dim3((unsigned)((g.Q + 255) / 256)), dim3(256), 0, st, cQ, g.Q, g.Q, 1.0f);
    T2* buf = (T2*)ws;
    const ChirpSeq seqs[2] = {{g.P + g.K - 2, g.K - 1, g.P, 1.0}, {g.K - 1, g.Q + g.K - 2, g.Q, -1.0}};
    for (int i = 0; i < 2; ++i) {
        if (g.levels > 0) {
            czt_levels_chirp_spectrum(g, seqs[i], buf, spec + (int64_t)i * g.NFFT, tw, st);
            continue;
        }
        pair_cols_fwd<T, 2>(g, buf, 1, st, seqs[i]);
        hipLaunchKernelGGL((czt_rows_kernel<T, true>), dim3((unsigned)g.C), dim3(TILE_T), Prec<T>::lds_bytes, st, buf,
                           (const T2*)nullptr, spec + (int64_t)i * g.NFFT, g.C, tw);
    }
    return hipGetLastError() == hipSuccess ? GFX_OK : GFX_ELAUNCH;
}

template <typename T, int CC>
static void pair_chain(const CztGeom& g, const float* z, float* y, int64_t ldy, int64_t lo, int64_t len, int64_t rows,
                       const typename Prec<T>::T2* cP, const typename Prec<T>::T2* cQ, const typename Prec<T>::T2* mid,
                       const typename Prec<T>::T2* spec,
                       typename Prec<T>::T2* buf, const typename Prec<T>::T2* tw, hipStream_t st) {
    using T2 = typename Prec<T>::T2;
    constexpr int LDS = Prec<T>::lds_bytes;
    constexpr bool FUSED = pair_mid_fused<T, CC>();
    const int64_t pairs = (rows + 1) / 2;
    const dim3 grid(TILE_M / 256, (unsigned)pairs), blk(256);
    const unsigned tiles = (unsigned)(pairs * g.C);
    hipLaunchKernelGGL((czt_pair_in_kernel<T, CC>), grid, blk, 0, st, z, cP, buf, g, rows);
    hipLaunchKernelGGL((czt_rows_kernel<T, false>), dim3(tiles), dim3(TILE_T), LDS, st, buf, spec, (T2*)nullptr, g.C, tw);
    hipLaunchKernelGGL((czt_pair_mid_kernel<T, CC, FUSED>), grid, blk, 0, st, buf, mid, g);
    if constexpr (!FUSED)
        hipLaunchKernelGGL((czt_cols_fwd_kernel<T, CC, 1>), grid, blk, 0, st, (const float*)nullptr, (const T2*)nullptr, buf, g,
                           (int64_t)0, (int64_t)0, (int64_t)0, ChirpSeq{0, 0, 1, 1.0});
    hipLaunchKernelGGL((czt_rows_kernel<T, false>), dim3(tiles), dim3(TILE_T), LDS, st, buf, spec + g.NFFT, (T2*)nullptr, g.C, tw);
    hipLaunchKernelGGL((czt_pair_out_kernel<T, CC>), grid, blk, 0, st, (const T2*)buf, cQ, y, ldy, lo, len, g, rows);
}

template <typename T>
static int czt_pair_alias(const float* z, float* y, int64_t ldy, int64_t lo, int64_t len, int64_t rows, int64_t P,
                          const void* plan, void* ws, size_t ws_bytes, void* stream, const gfx_rowmap_t* ymap = nullptr,
                          int yC = 0, int64_t row0 = 0, const uint32_t* given_max = nullptr, int relu = 0) {
    using T2 = typename Prec<T>::T2;
    CztGeom g;
    if (!z || !y || !plan || rows <= 0 || !czt_pair_geom(P, g)) return GFX_EINVAL;
    if (ymap) {
        if (yC < 1 || row0 < 0 || ymap->inner <= 0 || ymap->inner > 0x7fffffffLL || (row0 + rows) / yC > 0x7fffffffLL)
            return GFX_EINVAL;
        g.yC = yC;
        g.ymap = *ymap;
    }
    if (lo < 0 || len < 1 || lo + len > g.Q || ldy < len) return GFX_EINVAL;
    g.relu = relu ? 1 : 0;
    const int64_t pairs = (rows + 1) / 2;
    if (!ws || ws_bytes < (size_t)pairs * g.NFFT * sizeof(T2) + pair_rmax_bytes(rows)) return GFX_ENOSPC;
    hipStream_t st = (hipStream_t)stream;
    static const bool scaled = [] { const char* e = getenv("GRAFX_ALIAS_PAIR_SCALE"); return !(e && e[0] == '0'); }();
    uint32_t* rmax = nullptr;
    if (scaled && rows > 1 && given_max) {
        rmax = const_cast<uint32_t*>(given_max);     // the producer of z left them (gfx_fftconv_rowmax_f32)
    } else if (scaled && rows > 1) {
        // max |z| of every row, behind the transforms' own workspace
        rmax = (uint32_t*)((T2*)ws + pairs * g.NFFT);
        if (hipMemsetAsync(rmax, 0, (size_t)rows * sizeof(uint32_t), st) != hipSuccess) return GFX_ELAUNCH;
        for (int64_t r0 = 0; r0 < rows; r0 += 65535) {
            const int64_t n = rows - r0 < 65535 ? rows - r0 : 65535;
            hipLaunchKernelGGL(czt_rowmax_kernel<T>, dim3((unsigned)((g.P + 3 + RMAX_SPAN - 1) / RMAX_SPAN), (unsigned)n), dim3(256), 0, st,
                               z + r0 * g.P, g.P, rmax + r0);
        }
    }
    const T2* tw = czt_twiddles<T>(st);
    if (!tw || !czt_allow_lds(czt_rows_kernel<T, false>, Prec<T>::lds_bytes)) return GFX_ELAUNCH;
    const T2* cP = (const T2*)plan;
    const T2* cQ = cP + g.P;
    const T2* mid = cQ + g.Q;
    const T2* spec = mid + g.P;
    const int64_t step = 65535 / g.S;        // pairs * S sub-transforms ride on a grid dimension
    for (int64_t p0 = 0; p0 < pairs; p0 += step) {
        const int64_t n = rows - 2 * p0 < 2 * step ? rows - 2 * p0 : 2 * step;   // rows of this launch chain
        const float* zc = z + 2 * p0 * g.P;
        float* yc = ymap ? y : y + 2 * p0 * ldy;
        T2* buf = (T2*)ws + p0 * g.NFFT;
        g.row0 = row0 + 2 * p0;
        g.rmax = rmax ? rmax + 2 * p0 : nullptr;
        if (g.levels > 0) {
            if constexpr (sizeof(T) == 4) {
                static const bool fused = [] { const char* e = getenv("GRAFX_CZT_FUSED_LEVEL"); return !(e && e[0] == '0'); }();
                if (g.levels == 1 && fused) {
                    int rc = GFX_EINVAL;
#define GFX_PL(CC) case CC: rc = pair_chain_one_level<T, CC>(g, zc, yc, ldy, lo, len, n, cP, cQ, mid, spec, buf, tw, st); break;
                    switch (g.C) { GFX_CZT_SIZES(GFX_PL) default: break; }
#undef GFX_PL
                    if (rc != GFX_OK) return rc;
                    continue;
                }
            }
            pair_chain_levels<T>(g, zc, yc, ldy, lo, len, n, cP, cQ, mid, spec, buf, tw, st);
            continue;
        }
#define GFX_PC(CC) case CC: pair_chain<T, CC>(g, zc, yc, ldy, lo, len, n, cP, cQ, mid, spec, buf, tw, st); break;
        switch (g.C) { GFX_CZT_PAIR_SIZES(GFX_PC) default: return GFX_EINVAL; }
#undef GFX_PC
    }
    return hipGetLastError() == hipSuccess ? GFX_OK : GFX_ELAUNCH;
}

}  // namespace gfx

using namespace gfx;

extern "C" {

#ifndef GFX_CZT_PAIR_F64   // this file compiles the float transforms; czt_pair_f64.hip includes it for the double ones

size_t gfx_odd_alias_pair_plan_bytes(int64_t P) {
    CztGeom g;
    return czt_pair_geom(P, g) ? czt_pair_plan_t2(g) * sizeof(float2) : 0;
}

size_t gfx_odd_alias_pair_workspace_bytes(int64_t rows, int64_t P) {
    CztGeom g;
    if (rows <= 0 || !czt_pair_geom(P, g)) return 0;
    return (size_t)((rows + 1) / 2) * g.NFFT * sizeof(float2) + pair_rmax_bytes(rows);
}

int gfx_odd_alias_pair_plan_f32(void* plan, int64_t P, void* ws, size_t ws_bytes, void* stream) {
    return czt_pair_plan<float>(plan, P, ws, ws_bytes, (hipStream_t)stream);
}

int gfx_odd_alias_pair_f32(const float* z, float* y, int64_t ldy, int64_t lo, int64_t len, int64_t rows, int64_t P,
                           const void* plan, void* ws, size_t ws_bytes, void* stream) {
    return czt_pair_alias<float>(z, y, ldy, lo, len, rows, P, plan, ws, ws_bytes, stream);
}

int gfx_odd_alias_pair_rows_f32(const float* z, float* y, gfx_rowmap_t ymap, int64_t C, int64_t row0, int64_t lo, int64_t len,
                                int64_t rows, int64_t P, const void* plan, void* ws, size_t ws_bytes, void* stream) {
    if (C < 1 || C > 0x7fffffffLL || (row0 & 1)) return GFX_EINVAL;
    return czt_pair_alias<float>(z, y, len, lo, len, rows, P, plan, ws, ws_bytes, stream, &ymap, (int)C, row0);
}

int gfx_odd_alias_pair_max_f32(const float* z, float* y, int64_t ldy, int64_t lo, int64_t len, int64_t rows, int64_t P,
                               const void* plan, void* ws, size_t ws_bytes, const uint32_t* rowmax, void* stream) {
    return czt_pair_alias<float>(z, y, ldy, lo, len, rows, P, plan, ws, ws_bytes, stream, nullptr, 0, 0, rowmax);
}

int gfx_odd_alias_pair_rows_max_f32(const float* z, float* y, gfx_rowmap_t ymap, int64_t C, int64_t row0, int64_t lo, int64_t len,
                                    int64_t rows, int64_t P, const void* plan, void* ws, size_t ws_bytes,
                                    const uint32_t* rowmax, void* stream) {
    if (C < 1 || C > 0x7fffffffLL || (row0 & 1)) return GFX_EINVAL;
    return czt_pair_alias<float>(z, y, len, lo, len, rows, P, plan, ws, ws_bytes, stream, &ymap, (int)C, row0, rowmax);
}

#else

size_t gfx_odd_alias_pair_precise_plan_bytes(int64_t P) { return 2 * gfx_odd_alias_pair_plan_bytes(P); }

size_t gfx_odd_alias_pair_precise_workspace_bytes(int64_t rows, int64_t P) { return 2 * gfx_odd_alias_pair_workspace_bytes(rows, P); }

int gfx_odd_alias_pair_precise_plan_f32(void* plan, int64_t P, void* ws, size_t ws_bytes, void* stream) {
    return czt_pair_plan<double>(plan, P, ws, ws_bytes, (hipStream_t)stream);
}

int gfx_odd_alias_pair_precise_f32(const float* z, float* y, int64_t ldy, int64_t lo, int64_t len, int64_t rows, int64_t P,
                                   const void* plan, void* ws, size_t ws_bytes, void* stream) {
    return czt_pair_alias<double>(z, y, ldy, lo, len, rows, P, plan, ws, ws_bytes, stream);
}

int gfx_odd_alias_pair_precise_max_f32(const float* z, float* y, int64_t ldy, int64_t lo, int64_t len, int64_t rows, int64_t P,
                                       const void* plan, void* ws, size_t ws_bytes, const uint32_t* rowmax, int relu,
                                       void* stream) {
    return czt_pair_alias<double>(z, y, ldy, lo, len, rows, P, plan, ws, ws_bytes, stream, nullptr, 0, 0, rowmax, relu);
}

#endif

}  // extern "C"
