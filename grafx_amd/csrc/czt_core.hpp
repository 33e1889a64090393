// Shared pieces of the chirp-z transforms (czt.hip: one real row per transform; czt_pair.hip: two real rows packed
// into one complex transform): precision traits, the geometry, the chirp tables and the column / tile kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/grafx_amd.h"
#include "fft_tile.hpp"
#include "fft_tile_f64.hpp"
#include "small_dft.hpp"

namespace gfx {

// the workspace streams through every pass once: non-temporal accesses keep it from evicting what IS reused (the chirp
// spectrum, the twiddle tables).  -DGFX_CZT_NT=0 for A/B.
#ifndef GFX_CZT_EARLY_SPEC_F64
#define GFX_CZT_EARLY_SPEC_F64 0
#endif
#ifndef GFX_CZT_EARLY_SPEC
#define GFX_CZT_EARLY_SPEC 1
#endif
#ifndef GFX_CZT_NT
#define GFX_CZT_NT 1
#endif
#if GFX_CZT_NT
#define GFX_CZT_LOAD(p) __builtin_nontemporal_load(p)
#define GFX_CZT_STORE(v, p) __builtin_nontemporal_store(v, p)
#else
#define GFX_CZT_LOAD(p) (*(p))
#define GFX_CZT_STORE(v, p) (*(p) = (v))
#endif


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
    // two rows per transform (czt_pair.hip): bits of max |z| of every row of the launch chain, or null -- the second row
    // of a pair goes through scaled by the power of two that brings it to the first row's binade (pair_scale)
    const uint32_t* rmax;
    int relu;                // the last pass stores max(y, 0) (the envelope smoother's relu, core/envelope.py:48, fused in)
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
    g.rmax = nullptr;
    g.relu = 0;
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

// Tile accesses through a buffer descriptor: base in SGPRs, one lane offset, the register row as the scalar offset -- a
// global access with a 2048 a byte offset needs a 64-bit VGPR address per row (the immediate field ends at 4095), 64
// registers of addresses per phase.  AUX 2 = non-temporal.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const void* base, uint32_t bytes) {
    const uint64_t p = reinterpret_cast<uint64_t>(base);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p), hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, bytes, 0x00020000);
}
template <int AUX> __device__ __forceinline__ cx tile_ld(__amdgpu_buffer_rsrc_t r, int t, int row, cx*) {
    return __builtin_bit_cast(cx, __builtin_amdgcn_raw_buffer_load_b64(r, 8u * (uint32_t)t, (uint32_t)(row * 256 * 8), AUX));
}
template <int AUX> __device__ __forceinline__ cxd tile_ld(__amdgpu_buffer_rsrc_t r, int t, int row, cxd*) {
    return __builtin_bit_cast(cxd, __builtin_amdgcn_raw_buffer_load_b128(r, 16u * (uint32_t)t, (uint32_t)(row * 256 * 16), AUX));
}
template <int AUX> __device__ __forceinline__ void tile_st(__amdgpu_buffer_rsrc_t r, int t, int row, cx v) {
    using u2 = unsigned __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, v), r, 8u * (uint32_t)t, (uint32_t)(row * 256 * 8), AUX);
}
template <int AUX> __device__ __forceinline__ void tile_st(__amdgpu_buffer_rsrc_t r, int t, int row, cxd v) {
    using u4 = unsigned __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), r, 16u * (uint32_t)t, (uint32_t)(row * 256 * 16), AUX);
}

// the workspace's points in the column passes (T2 in memory, the tile's complex type in registers)
template <typename T>
__device__ __forceinline__ typename Prec<T>::cxt buf_load(const typename Prec<T>::T2* p) {
    return GFX_CZT_LOAD(reinterpret_cast<const typename Prec<T>::cxt*>(p));
}
template <typename T>
__device__ __forceinline__ void buf_store(typename Prec<T>::T2* p, typename Prec<T>::cxt v) {
    GFX_CZT_STORE(v, reinterpret_cast<typename Prec<T>::cxt*>(p));
}

// A column pass's view of one transform's points: point (k1, n2) = b[k1 * 8192 + n2].  float: through a descriptor (lane
// offset 8 n2, row k1 as the scalar offset) -- a global access per row costs a 64-bit VGPR address and its carry chain per
// point, in kernels that are bound by vector-memory issue and the vector ALU; double: global accesses (measured equal).
template <typename T> struct ColBuf {
    using cx = typename Prec<T>::cxt;
    static constexpr int NT = GFX_CZT_NT ? 2 : 0;
    typename Prec<T>::T2* b;
    __amdgpu_buffer_rsrc_t r;
    __device__ __forceinline__ ColBuf(const typename Prec<T>::T2* base, int64_t points)
        : b(const_cast<typename Prec<T>::T2*>(base)), r(tile_rsrc(base, (uint32_t)(points * (int64_t)sizeof(typename Prec<T>::T2)))) {}
    __device__ __forceinline__ cx ld(int k1, int n2) const {
        if constexpr (sizeof(T) == 4)
            return __builtin_bit_cast(cx, __builtin_amdgcn_raw_buffer_load_b64(r, 8u * (uint32_t)n2, (uint32_t)k1 * (TILE_M * 8u), NT));
        else
            return buf_load<T>(&b[(int64_t)k1 * TILE_M + n2]);
    }
    __device__ __forceinline__ void st(int k1, int n2, cx v) const {
        if constexpr (sizeof(T) == 4) {
            using u2 = unsigned __attribute__((ext_vector_type(2)));
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, v), r, 8u * (uint32_t)n2, (uint32_t)k1 * (TILE_M * 8u), NT);
        } else {
            buf_store<T>(&b[(int64_t)k1 * TILE_M + n2], v);
        }
    }
};

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

// The column twiddles W_NS^(n2 k1), k1 = 1 .. C-1 (asked for in turn, k1 a constant after unrolling).  float: two digits,
// k1 = 8 a + b -- W^(n2 b) for b < 8 and W^(8 n2 a), each one sincospif on an exactly reduced phase, and one complex product
// for the rest: 11 evaluations instead of 34 at 35 tiles (one per point made the float column passes instruction-bound:
// ~50 of ~80 instructions per point; one more rounding, 6e-8).  double: powers of W^(n2) by repeated multiplication (31 steps
// lose ~1e-15, and a double sincospi per point would make the column kernels compute-bound).
template <typename T, int C> struct ColTw;
template <int C> struct ColTw<float, C> {
    static constexpr int NLO = C < 8 ? C : 8, NHI = (C - 1) / 8 + 1;
    cx lo[NLO], hi[NHI];
    __device__ __forceinline__ ColTw(int n2, int NS, bool conj) {
        lo[0] = hi[0] = cx{1.0f, 0.0f};
        int r = 0;
#pragma unroll
        for (int b = 1; b < NLO; ++b) {
            r += n2;                                       // n2 b < 8 x 8192 <= NS whenever C >= 8; below, b < C keeps it under NS
            lo[b] = unit_root_f(r, NS, conj);
        }
        r = 0;
#pragma unroll
        for (int a = 1; a < NHI; ++a) {
            r += 8 * n2;                                   // (8 n2 < 65536 <= NS here: one subtraction reduces)
            r -= r >= NS ? NS : 0;
            hi[a] = unit_root_f(r, NS, conj);
        }
    }
    __device__ __forceinline__ cx at(int k1) const {
        const int a = k1 >> 3, b = k1 & 7;
        return a == 0 ? lo[b] : b == 0 ? hi[a] : cmul(hi[a], lo[b]);
    }
};
template <int C> struct ColTw<double, C> {
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
    const ColBuf<T> cb(b, NS);
    cx v[C];
#pragma unroll
    for (int n1 = 0; n1 < C; ++n1) {
        const int64_t i = (int64_t)n1 * TILE_M + n2;
        cx e = {0, 0};
        if (MODE == 0) {
            if (i >= lo && i < lo + len) e = to_cx(tab[i]) * (T)z[row * ldz + (i - lo)];
        } else if (MODE == 1) {
            e = cb.ld(n1, n2);
        } else {
            if (i <= cs.hi) e = to_cx(chirp_d<T>(i, cs.den, cs.sign));
            else if (i >= g.NFFT - cs.lo) e = to_cx(chirp_d<T>(g.NFFT - i, cs.den, cs.sign));
        }
        v[n1] = e;
    }
    col_dft<C, false>(v);
    ColTw<T, C> tw(n2, (int)NS, false);
#pragma unroll
    for (int k1 = 0; k1 < C; ++k1) {
        const cx e = v[spos(C, k1)];
        const cx o = k1 == 0 ? e : cmul(e, tw.at(k1));
        cb.st(k1, n2, o);
    }
}

// One tile per (row, k1): forward, times the chirp spectrum, inverse -- in place.  PLAN: forward only, spectrum stored
// in thread layout.
template <typename T, bool PLAN>
__global__ __launch_bounds__(TILE_T, (sizeof(T) == 4 || GFX_F64_SPLIT) ? 2 : 1) void czt_rows_kernel(
    typename Prec<T>::T2* __restrict__ buf, const typename Prec<T>::T2* __restrict__ spec,
    typename Prec<T>::T2* __restrict__ spec_out, int C, const typename Prec<T>::T2* __restrict__ twtab) {
    using cx = typename Prec<T>::cxt;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    cx* lds = reinterpret_cast<cx*>(lds_raw);
    const int t = threadIdx.x;
    const int64_t tile = blockIdx.x;                    // row * C + k1
    const int k1 = (int)(tile % C);                       // C here = tiles per signal row = S * C
    constexpr uint32_t TILE_BYTES = TILE_M * sizeof(cx);
    constexpr int NT = GFX_CZT_NT ? 2 : 0;
    const __amdgpu_buffer_rsrc_t rb = tile_rsrc(reinterpret_cast<cx*>(buf) + tile * TILE_M, TILE_BYTES);
    constexpr bool SPLIT = sizeof(T) == 8 && GFX_F64_SPLIT;     // (fft_tile_f64.hpp: exchanges in two rounds, twiddles per pass)
    typename Prec<T>::Tw tw;
    if constexpr (!SPLIT) tile_twiddles(tw, twtab, t);
    cx v[32], w[2][16];
    // (descriptor accesses in float: 10.1 -> 9.3 ms per 4096 rows; in double the global form measured 1 % better)
    cx* b = reinterpret_cast<cx*>(buf) + tile * TILE_M;
#pragma unroll
    for (int a = 0; a < 32; ++a) {
        if constexpr (sizeof(T) == 4) v[a] = tile_ld<NT>(rb, t, a, (cx*)nullptr);
        else v[a] = GFX_CZT_LOAD(&b[t + 256 * a]);
    }
    // float: the spectrum's loads go out with the tile's, up front, as in fftconv1_kernel -- left to itself the compiler
    // issues each one right before its product and waits for it, 32 L2 round trips in the middle of the tile (in double the
    // 128 registers are not there)
    constexpr bool EARLY = !PLAN && (sizeof(T) == 4 || GFX_CZT_EARLY_SPEC_F64) && GFX_CZT_EARLY_SPEC;
    cx sreg[EARLY ? 32 : 1];
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc(reinterpret_cast<const cx*>(spec) + (PLAN ? 0 : (int64_t)k1 * TILE_M), TILE_BYTES);
    if constexpr (EARLY) {
#pragma unroll
        for (int q = 0; q < 32; ++q) sreg[q] = tile_ld<0>(rs, t, q, (cx*)nullptr);
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (SPLIT) tile_forward(v, w, twtab, lds, t);
    else tile_forward(v, w, tw, lds, t);
    if (PLAN) {
        cx* o = reinterpret_cast<cx*>(spec_out) + (int64_t)k1 * TILE_M;
#pragma unroll
        for (int q = 0; q < 32; ++q) o[q * TILE_T + t] = w[q >> 4][q & 15];
        return;
    }
    if constexpr (SPLIT) {
        // eight spectrum values at a time, fenced: left to itself the scheduler requests all 32 (128 registers) on top of the
        // 128 of the tile, and at two workgroups per CU there are 256 in all
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            __builtin_amdgcn_sched_barrier(0);
            cx sp[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) sp[q] = tile_ld<0>(rs, t, 8 * g + q, (cx*)nullptr);
#pragma unroll
            for (int q = 0; q < 8; ++q) w[(8 * g + q) >> 4][(8 * g + q) & 15] = cmul(w[(8 * g + q) >> 4][(8 * g + q) & 15], sp[q]);
        }
        __builtin_amdgcn_sched_barrier(0);
    } else {
#pragma unroll
        for (int q = 0; q < 32; ++q)
            w[q >> 4][q & 15] = cmul(w[q >> 4][q & 15], EARLY ? sreg[EARLY ? q : 0] : tile_ld<0>(rs, t, q, (cx*)nullptr));
    }
    __syncthreads();
    if constexpr (SPLIT) tile_inverse(w, v, twtab, lds, t);
    else tile_inverse(w, v, tw, lds, t);
#pragma unroll
    for (int a = 0; a < 32; ++a) {
        if constexpr (sizeof(T) == 4) tile_st<NT>(rb, t, a, v[brev(a, 5)]);
        else GFX_CZT_STORE(v[brev(a, 5)], &b[t + 256 * a]);
    }
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
    const ColBuf<T> cb(b, NS);
    cx v[C];
    ColTw<T, C> twi(n2, (int)NS, true);
#pragma unroll
    for (int k1 = 0; k1 < C; ++k1) {
        const cx e = cb.ld(k1, n2);
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
        ColTw<T, C> twf(n2, (int)NS, false);
#pragma unroll
        for (int k1 = 0; k1 < C; ++k1) {
            const cx e = u[spos(C, k1)];
            const cx o = k1 == 0 ? e : cmul(e, twf.at(k1));
            cb.st(k1, n2, o);
        }
        return;
    }
#pragma unroll
    for (int n1 = 0; n1 < C; ++n1) {
        const int64_t i = (int64_t)n1 * TILE_M + n2;
        const cx e = v[spos(C, n1)] * sc;
        if (MODE == 2) {                                   // plain inverse of a sub-transform (outer level follows)
            cb.st(n1, n2, e);
        } else if (MODE == 0) {
            cx o = {0, 0};
            if (i < g.K) {
                const T wk = (i == 0 || i == g.K - 1) ? (T)1 : (T)2;
                o = cmul(cmul(e, to_cx(cP[i])), to_cx(cQ[i])) * wk;
            }
            cb.st(n1, n2, o);
        } else {
            if (i >= lo && i < lo + len) {
                const cx c = to_cx(cQ[i]);
                czt_out_row(g, y, ldy, row)[i - lo] = (float)((e.x * c.x - e.y * c.y) / (T)g.Q);
            }
        }
    }
}

// ---- outer radix-4 levels (NFFT = 4 NS, n = n3 NS + n', k = k3 + 4 k'): the twiddle W_NFFT^(n' k3) --------------------
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

// the passes below the outermost level of a transform with outer levels (g.levels >= 1), in place on `rows` buffers of
// g.NFFT points: inner radix-4 levels and the column pass (forward), the inverse column pass and the inner levels
// (inverse, scaled by 4^(levels-1) / NFFT: the outermost level's 1/4 is the caller's); and a chirp sequence's spectrum
// in the tile pass's layout.  Defined in czt.hip.
void czt_levels_fwd(const CztGeom& g, float2* buf, int64_t rows, hipStream_t st);
void czt_levels_fwd(const CztGeom& g, double2* buf, int64_t rows, hipStream_t st);
void czt_levels_inv(const CztGeom& g, float2* buf, int64_t rows, hipStream_t st);
void czt_levels_inv(const CztGeom& g, double2* buf, int64_t rows, hipStream_t st);
void czt_levels_chirp_spectrum(const CztGeom& g, ChirpSeq cs, float2* buf, float2* spec, const float2* tw, hipStream_t st);
void czt_levels_chirp_spectrum(const CztGeom& g, ChirpSeq cs, double2* buf, double2* spec, const double2* tw, hipStream_t st);

template <typename K>
static bool czt_allow_lds(K kernel, int bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) ==
           hipSuccess;
}

// the double tile's twiddles, one table per device (czt.hip)
const double2* tile_twiddle_table_f64(hipStream_t stream);

template <typename T> inline const typename Prec<T>::T2* czt_twiddles(hipStream_t st);
template <> inline const float2* czt_twiddles<float>(hipStream_t st) { return tile_twiddle_table(st); }
template <> inline const double2* czt_twiddles<double>(hipStream_t st) { return tile_twiddle_table_f64(st); }

}  // namespace gfx
