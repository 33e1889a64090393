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

#include "czt_core.hpp"

namespace gfx {

// ---- outer radix-4 level (S = 4): NFFT = 4 NS, n = n3 NS + n', k = k3 + 4 k' ---------------------------------------
//   forward:  sub[k3][n'] = ( sum_n3 x[n3 NS + n'] W_4^(n3 k3) ) W_NFFT^(n' k3)      then four NS-point transforms
//   inverse:  x[n3 NS + n'] = (1/4) sum_k3 ( sub[k3][n'] conj W_NFFT^(n' k3) ) W_4^(-n3 k3)
// in place (a thread owns the four positions n' + n3 NS).  Input / output modes as in the column kernels.
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
            e = buf_load<T>(&b[i]);
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
        buf_store<T>(&b[k3 * NS + np], o);
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
        const cx e = buf_load<T>(&b[k3 * NS + np]);
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
            buf_store<T>(&b[k3 * NS + np], o);
        }
        return;
    }
#pragma unroll
    for (int n3 = 0; n3 < 4; ++n3) {
        const int64_t i = n3 * NS + np;
        const cx e = v[brev(n3, 2)] * (T)0.25;
        if (MODE == 3) {                                   // plain inverse of an inner level
            buf_store<T>(&b[i], e);
        } else if (MODE == 0) {
            cx o = {0, 0};
            if (i < g.K) {
                const T wk = (i == 0 || i == g.K - 1) ? (T)1 : (T)2;
                o = cmul(cmul(e, to_cx(cP[i])), to_cx(cQ[i])) * wk;
            }
            buf_store<T>(&b[i], o);
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

const double2* tile_twiddle_table_f64(hipStream_t stream) {
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

// what czt_pair.hip shares with this file when its transforms need outer levels (declared in czt_core.hpp)
template <typename T>
static void czt_chirp_spectrum(const CztGeom& g, ChirpSeq cs, typename Prec<T>::T2* buf, typename Prec<T>::T2* spec,
                               const typename Prec<T>::T2* tw, hipStream_t st) {
    using T2 = typename Prec<T>::T2;
    const int ctot = g.S * g.C;
    hipLaunchKernelGGL((czt_outer_fwd_kernel<T, 2>), dim3((unsigned)(g.NFFT / 4 / 256), 1), dim3(256), 0, st, (const float*)nullptr,
                       (const T2*)nullptr, buf, g, (int64_t)0, (int64_t)0, (int64_t)0, cs);
    czt_inner_fwd<T>(g, buf, 1, st);
    hipLaunchKernelGGL((czt_rows_kernel<T, true>), dim3((unsigned)ctot), dim3(TILE_T), Prec<T>::lds_bytes, st, buf,
                       (const T2*)nullptr, spec, ctot, tw);
}
void czt_levels_fwd(const CztGeom& g, float2* buf, int64_t rows, hipStream_t st) { czt_inner_fwd<float>(g, buf, rows, st); }
void czt_levels_fwd(const CztGeom& g, double2* buf, int64_t rows, hipStream_t st) { czt_inner_fwd<double>(g, buf, rows, st); }
void czt_levels_inv(const CztGeom& g, float2* buf, int64_t rows, hipStream_t st) {
    czt_inner_inv<float>(g, buf, nullptr, nullptr, rows, st);
}
void czt_levels_inv(const CztGeom& g, double2* buf, int64_t rows, hipStream_t st) {
    czt_inner_inv<double>(g, buf, nullptr, nullptr, rows, st);
}
void czt_levels_chirp_spectrum(const CztGeom& g, ChirpSeq cs, float2* buf, float2* spec, const float2* tw, hipStream_t st) {
    czt_chirp_spectrum<float>(g, cs, buf, spec, tw, st);
}
void czt_levels_chirp_spectrum(const CztGeom& g, ChirpSeq cs, double2* buf, double2* spec, const double2* tw, hipStream_t st) {
    czt_chirp_spectrum<double>(g, cs, buf, spec, tw, st);
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
