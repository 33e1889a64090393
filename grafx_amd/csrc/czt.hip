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
// two circular convolutions of NFFT = C x 8192 >= (3P - 1) / 2 points, each a "four-step" FFT around the tile:
//   cols_fwd : per column n2 (of 8192) a C-point DFT over n1 (registers) and the twiddle W_NFFT^(n2 k1)
//   rows     : per row k1 one tile: forward, multiply by the chirp's spectrum (thread layout, precomputed), inverse
//   cols_inv : conjugate twiddle, C-point inverse DFT, then the chirp / weights of the next step
// all in place on one NFFT-point complex buffer per signal row.  fp32 throughout; the chirps are evaluated once per P
// in double with the phase reduced exactly (k^2 mod 2P in integers).  C <= 32 covers P <= 174,763; up to P <= 699,051
// (10 s of audio at 48 kHz plus the filter) an OUTER radix-4 level splits the 2^20-point transform into four 2^18-point
// ones (czt_outer_*: 4-point DFT over the quarters + twiddle W_N^(n' k3), in place).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/grafx_amd.h"
#include "fft_tile.hpp"

namespace gfx {

constexpr int CZT_MAXC = 32;

struct CztGeom {
    int64_t P, Q, K, NFFT;   // NFFT = S * C * 8192
    int C, S;                // C <= 32 columns per sub-transform; S = 1, or 4 sub-transforms under an outer radix-4 level
};

static inline bool czt_geom(int64_t P, CztGeom& g) {
    if (P < 3 || (P & 1) == 0) return false;
    g.P = P;
    g.Q = P - 1;
    g.K = (P + 1) / 2;
    const int64_t need = P + g.K - 1;
    int C = 1;
    while ((int64_t)C * TILE_M < need) C *= 2;
    g.S = 1;
    if (C > CZT_MAXC) {
        if (C > 4 * CZT_MAXC) return false;
        g.S = 4;
        C = CZT_MAXC;
    }
    g.C = C;
    g.NFFT = (int64_t)g.S * C * TILE_M;
    return true;
}

// plan layout (float2 units): cP[P] | cQ[Q] | spectrum of bP [NFFT] | spectrum of bQ [NFFT]
static inline size_t czt_plan_f2(const CztGeom& g) { return (size_t)(g.P + g.Q + 2 * g.NFFT); }

__device__ __forceinline__ float2 chirp_d(int64_t j, int64_t den, double sign) {   // exp(sign i pi j^2 / den)
    const int64_t r = (j * j) % (2 * den);
    double s, c;
    sincospi((double)r / (double)den, &s, &c);
    return make_float2((float)c, (float)(sign * s));
}

__global__ void czt_chirp_table_kernel(float2* __restrict__ tab, int64_t n, int64_t den, float sign) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) tab[i] = chirp_d(i, den, sign);
}

// W_NFFT^(n2 k1): the argument 2 n2 k1 / NFFT is exact in float (NFFT is a power of two, n2 k1 < 2^18)
__device__ __forceinline__ cx col_twiddle(int n2, int k1, float inv_half_nfft, bool conj) {
    float s, c;
    sincospif((float)(n2 * k1) * inv_half_nfft, &s, &c);
    return cx{c, conj ? s : -s};
}

// MODE 0: real rows z (rows x P) times cP;  MODE 1: the complex buffer itself;  MODE 2 / 3: the chirp sequences bP / bQ
// (plan building: b[j] at circular index j mod NFFT for j in [-(P-1), K-1] resp. [-(K-1), Q-1], zero elsewhere)
template <int C, int MODE>
__global__ __launch_bounds__(256) void czt_cols_fwd_kernel(const float* __restrict__ z, const float2* __restrict__ cP,
                                                          float2* __restrict__ buf, CztGeom g) {
    const int n2 = blockIdx.x * 256 + threadIdx.x;          // column 0..8191
    const int64_t row = blockIdx.y;                        // signal row * S + sub-transform
    const int64_t NS = g.NFFT / g.S;                       // points of one sub-transform
    float2* b = buf + row * NS;
    cx v[C];
#pragma unroll
    for (int n1 = 0; n1 < C; ++n1) {
        const int64_t i = (int64_t)n1 * TILE_M + n2;
        cx e = {0.0f, 0.0f};
        if (MODE == 0) {
            if (i < g.P) e = to_cx(cP[i]) * z[row * g.P + i];
        } else if (MODE == 1) {
            e = to_cx(b[i]);
        } else {
            const int64_t lo = MODE == 2 ? g.P - 1 : g.K - 1, hi = MODE == 2 ? g.K - 1 : g.Q - 1;
            const int64_t den = MODE == 2 ? g.P : g.Q;
            const double sign = MODE == 2 ? 1.0 : -1.0;
            if (i <= hi) e = to_cx(chirp_d(i, den, sign));
            else if (i >= g.NFFT - lo) e = to_cx(chirp_d(g.NFFT - i, den, sign));
        }
        v[n1] = e;
    }
    dif<C, false>(v);
    const float ihn = 2.0f / (float)NS;
    constexpr int LOGC = C == 1 ? 0 : C == 2 ? 1 : C == 4 ? 2 : C == 8 ? 3 : C == 16 ? 4 : 5;
#pragma unroll
    for (int k1 = 0; k1 < C; ++k1) {
        const cx e = v[brev(k1, LOGC)];
        const cx o = k1 == 0 ? e : cmul(e, col_twiddle(n2, k1, ihn, false));
        b[(int64_t)k1 * TILE_M + n2] = make_float2(o.x, o.y);
    }
}

// One tile per (row, k1): forward, times the chirp spectrum, inverse -- in place.  PLAN: forward only, spectrum stored
// in thread layout.
template <bool PLAN>
__global__ __launch_bounds__(TILE_T, 2) void czt_rows_kernel(float2* __restrict__ buf, const float2* __restrict__ spec,
                                                             float2* __restrict__ spec_out, int C,
                                                             const float2* __restrict__ twtab) {
    extern __shared__ __attribute__((aligned(16))) cx lds[];
    const int t = threadIdx.x;
    const int64_t tile = blockIdx.x;                    // row * C + k1
    const int k1 = (int)(tile % C);                       // C here = tiles per signal row = S * C
    cx* b = reinterpret_cast<cx*>(buf) + tile * TILE_M;
    TileTw tw;
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
template <int C, int MODE>
__global__ __launch_bounds__(256) void czt_cols_inv_kernel(float2* __restrict__ buf, const float2* __restrict__ cP,
                                                          const float2* __restrict__ cQ, float* __restrict__ y,
                                                          int64_t ldy, int64_t lo, int64_t len, CztGeom g) {
    const int n2 = blockIdx.x * 256 + threadIdx.x;
    const int64_t row = blockIdx.y;
    const int64_t NS = g.NFFT / g.S;
    float2* b = buf + row * NS;
    const float ihn = 2.0f / (float)NS;
    cx v[C];
#pragma unroll
    for (int k1 = 0; k1 < C; ++k1) {
        const cx e = to_cx(b[(int64_t)k1 * TILE_M + n2]);
        v[k1] = k1 == 0 ? e : cmul(e, col_twiddle(n2, k1, ihn, true));
    }
    dif<C, true>(v);
    constexpr int LOGC = C == 1 ? 0 : C == 2 ? 1 : C == 4 ? 2 : C == 8 ? 3 : C == 16 ? 4 : 5;
    const float sc = 1.0f / (float)NS;
#pragma unroll
    for (int n1 = 0; n1 < C; ++n1) {
        const int64_t i = (int64_t)n1 * TILE_M + n2;
        const cx e = v[brev(n1, LOGC)] * sc;
        if (MODE == 2) {                                   // plain inverse of a sub-transform (outer level follows)
            b[i] = make_float2(e.x, e.y);
        } else if (MODE == 0) {
            cx o = {0.0f, 0.0f};
            if (i < g.K) {
                const float wk = (i == 0 || i == g.K - 1) ? 1.0f : 2.0f;
                o = cmul(cmul(e, to_cx(cP[i])), to_cx(cQ[i])) * wk;
            }
            b[i] = make_float2(o.x, o.y);
        } else {
            if (i >= lo && i < lo + len) {
                const cx c = to_cx(cQ[i]);
                y[row * ldy + (i - lo)] = (e.x * c.x - e.y * c.y) / (float)g.Q;
            }
        }
    }
}

// ---- outer radix-4 level (S = 4): NFFT = 4 NS, n = n3 NS + n', k = k3 + 4 k' ---------------------------------------
//   forward:  sub[k3][n'] = ( sum_n3 x[n3 NS + n'] W_4^(n3 k3) ) W_NFFT^(n' k3)      then four NS-point transforms
//   inverse:  x[n3 NS + n'] = (1/4) sum_k3 ( sub[k3][n'] conj W_NFFT^(n' k3) ) W_4^(-n3 k3)
// in place (a thread owns the four positions n' + n3 NS).  Input / output modes as in the column kernels.
__device__ __forceinline__ cx outer_twiddle(int64_t np, int k3, int64_t NFFT, bool conj) {
    // W_NFFT^(np k3): np k3 < 3 * 2^18 is exact in float, and so is the quotient by the power of two NFFT
    float s, c;
    sincospif(2.0f * (float)(np * k3) / (float)NFFT, &s, &c);
    return cx{c, conj ? s : -s};
}

template <int MODE>   // 0: real z times cP; 1: complex buffer; 2 / 3: chirp sequences bP / bQ (plan)
__global__ __launch_bounds__(256) void czt_outer_fwd_kernel(const float* __restrict__ z, const float2* __restrict__ cP,
                                                           float2* __restrict__ buf, CztGeom g) {
    const int64_t NS = g.NFFT / 4;
    const int64_t np = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = blockIdx.y;
    float2* b = buf + row * g.NFFT;
    cx v[4];
#pragma unroll
    for (int n3 = 0; n3 < 4; ++n3) {
        const int64_t i = n3 * NS + np;
        cx e = {0.0f, 0.0f};
        if (MODE == 0) {
            if (i < g.P) e = to_cx(cP[i]) * z[row * g.P + i];
        } else if (MODE == 1) {
            e = to_cx(b[i]);
        } else {
            const int64_t lo = MODE == 2 ? g.P - 1 : g.K - 1, hi = MODE == 2 ? g.K - 1 : g.Q - 1;
            const int64_t den = MODE == 2 ? g.P : g.Q;
            const double sign = MODE == 2 ? 1.0 : -1.0;
            if (i <= hi) e = to_cx(chirp_d(i, den, sign));
            else if (i >= g.NFFT - lo) e = to_cx(chirp_d(g.NFFT - i, den, sign));
        }
        v[n3] = e;
    }
    dif<4, false>(v);                                     // result for k3 at v[brev(k3, 2)]
#pragma unroll
    for (int k3 = 0; k3 < 4; ++k3) {
        const cx e = v[brev(k3, 2)];
        const cx o = k3 == 0 ? e : cmul(e, outer_twiddle(np, k3, g.NFFT, false));
        b[k3 * NS + np] = make_float2(o.x, o.y);
    }
}

template <int MODE>   // 0: next step's input (times cP w_k cQ for k < K, zero beyond); 1: real output slice
__global__ __launch_bounds__(256) void czt_outer_inv_kernel(float2* __restrict__ buf, const float2* __restrict__ cP,
                                                           const float2* __restrict__ cQ, float* __restrict__ y,
                                                           int64_t ldy, int64_t lo, int64_t len, CztGeom g) {
    const int64_t NS = g.NFFT / 4;
    const int64_t np = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = blockIdx.y;
    float2* b = buf + row * g.NFFT;
    cx v[4];
#pragma unroll
    for (int k3 = 0; k3 < 4; ++k3) {
        const cx e = to_cx(b[k3 * NS + np]);
        v[k3] = k3 == 0 ? e : cmul(e, outer_twiddle(np, k3, g.NFFT, true));
    }
    dif<4, true>(v);
#pragma unroll
    for (int n3 = 0; n3 < 4; ++n3) {
        const int64_t i = n3 * NS + np;
        const cx e = v[brev(n3, 2)] * 0.25f;
        if (MODE == 0) {
            cx o = {0.0f, 0.0f};
            if (i < g.K) {
                const float wk = (i == 0 || i == g.K - 1) ? 1.0f : 2.0f;
                o = cmul(cmul(e, to_cx(cP[i])), to_cx(cQ[i])) * wk;
            }
            b[i] = make_float2(o.x, o.y);
        } else {
            if (i >= lo && i < lo + len) {
                const cx c = to_cx(cQ[i]);
                y[row * ldy + (i - lo)] = (e.x * c.x - e.y * c.y) / (float)g.Q;
            }
        }
    }
}

template <int MODE>
static void launch_cols_fwd(const CztGeom& g, const float* z, const float2* cP, float2* buf, int64_t rows, hipStream_t st) {
    const dim3 grid(TILE_M / 256, (unsigned)(rows * g.S)), blk(256);   // one "row" per sub-transform
    switch (g.C) {
        case 1: hipLaunchKernelGGL((czt_cols_fwd_kernel<1, MODE>), grid, blk, 0, st, z, cP, buf, g); break;
        case 2: hipLaunchKernelGGL((czt_cols_fwd_kernel<2, MODE>), grid, blk, 0, st, z, cP, buf, g); break;
        case 4: hipLaunchKernelGGL((czt_cols_fwd_kernel<4, MODE>), grid, blk, 0, st, z, cP, buf, g); break;
        case 8: hipLaunchKernelGGL((czt_cols_fwd_kernel<8, MODE>), grid, blk, 0, st, z, cP, buf, g); break;
        case 16: hipLaunchKernelGGL((czt_cols_fwd_kernel<16, MODE>), grid, blk, 0, st, z, cP, buf, g); break;
        default: hipLaunchKernelGGL((czt_cols_fwd_kernel<32, MODE>), grid, blk, 0, st, z, cP, buf, g); break;
    }
}

template <int MODE>
static void launch_cols_inv(const CztGeom& g, float2* buf, const float2* cP, const float2* cQ, float* y, int64_t ldy,
                            int64_t lo, int64_t len, int64_t rows, hipStream_t st) {
    const dim3 grid(TILE_M / 256, (unsigned)(rows * g.S)), blk(256);
    switch (g.C) {
        case 1: hipLaunchKernelGGL((czt_cols_inv_kernel<1, MODE>), grid, blk, 0, st, buf, cP, cQ, y, ldy, lo, len, g); break;
        case 2: hipLaunchKernelGGL((czt_cols_inv_kernel<2, MODE>), grid, blk, 0, st, buf, cP, cQ, y, ldy, lo, len, g); break;
        case 4: hipLaunchKernelGGL((czt_cols_inv_kernel<4, MODE>), grid, blk, 0, st, buf, cP, cQ, y, ldy, lo, len, g); break;
        case 8: hipLaunchKernelGGL((czt_cols_inv_kernel<8, MODE>), grid, blk, 0, st, buf, cP, cQ, y, ldy, lo, len, g); break;
        case 16: hipLaunchKernelGGL((czt_cols_inv_kernel<16, MODE>), grid, blk, 0, st, buf, cP, cQ, y, ldy, lo, len, g); break;
        default: hipLaunchKernelGGL((czt_cols_inv_kernel<32, MODE>), grid, blk, 0, st, buf, cP, cQ, y, ldy, lo, len, g); break;
    }
}

template <typename K>
static bool czt_allow_lds(K kernel) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               TILE_LDS_BYTES) == hipSuccess;
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
    CztGeom g;
    if (!plan || !czt_geom(P, g)) return GFX_EINVAL;
    if (!ws || ws_bytes < (size_t)g.NFFT * sizeof(float2)) return GFX_ENOSPC;
    hipStream_t st = (hipStream_t)stream;
    const float2* tw = tile_twiddle_table(st);
    if (!tw || !czt_allow_lds(czt_rows_kernel<true>)) return GFX_ELAUNCH;
    float2* cP = (float2*)plan;
    float2* cQ = cP + g.P;
    float2* sP = cQ + g.Q;
    float2* sQ = sP + g.NFFT;
    hipLaunchKernelGGL(czt_chirp_table_kernel, dim3((unsigned)((g.P + 255) / 256)), dim3(256), 0, st, cP, g.P, g.P, -1.0f);
    hipLaunchKernelGGL(czt_chirp_table_kernel, dim3((unsigned)((g.Q + 255) / 256)), dim3(256), 0, st, cQ, g.Q, g.Q, 1.0f);
    float2* buf = (float2*)ws;
    const int ctot = g.S * g.C;
    const dim3 og((unsigned)(g.NFFT / 4 / 256), 1);
    if (g.S == 1) launch_cols_fwd<2>(g, nullptr, nullptr, buf, 1, st);
    else {
        hipLaunchKernelGGL(czt_outer_fwd_kernel<2>, og, dim3(256), 0, st, (const float*)nullptr, (const float2*)nullptr, buf, g);
        launch_cols_fwd<1>(g, nullptr, nullptr, buf, 1, st);
    }
    hipLaunchKernelGGL(czt_rows_kernel<true>, dim3((unsigned)ctot), dim3(TILE_T), TILE_LDS_BYTES, st, buf,
                       (const float2*)nullptr, sP, ctot, tw);
    if (g.S == 1) launch_cols_fwd<3>(g, nullptr, nullptr, buf, 1, st);
    else {
        hipLaunchKernelGGL(czt_outer_fwd_kernel<3>, og, dim3(256), 0, st, (const float*)nullptr, (const float2*)nullptr, buf, g);
        launch_cols_fwd<1>(g, nullptr, nullptr, buf, 1, st);
    }
    hipLaunchKernelGGL(czt_rows_kernel<true>, dim3((unsigned)ctot), dim3(TILE_T), TILE_LDS_BYTES, st, buf,
                       (const float2*)nullptr, sQ, ctot, tw);
    return hipGetLastError() == hipSuccess ? GFX_OK : GFX_ELAUNCH;
}

int gfx_odd_alias_f32(const float* z, float* y, int64_t ldy, int64_t lo, int64_t len, int64_t rows, int64_t P,
                      const void* plan, void* ws, size_t ws_bytes, void* stream) {
    CztGeom g;
    if (!z || !y || !plan || rows <= 0 || rows > 16383 || !czt_geom(P, g)) return GFX_EINVAL;
    if (lo < 0 || len < 1 || lo + len > g.Q || ldy < len) return GFX_EINVAL;
    if (!ws || ws_bytes < (size_t)rows * g.NFFT * sizeof(float2)) return GFX_ENOSPC;
    hipStream_t st = (hipStream_t)stream;
    const float2* tw = tile_twiddle_table(st);
    if (!tw || !czt_allow_lds(czt_rows_kernel<false>)) return GFX_ELAUNCH;
    const float2* cP = (const float2*)plan;
    const float2* cQ = cP + g.P;
    const float2* sP = cQ + g.Q;
    const float2* sQ = sP + g.NFFT;
    float2* buf = (float2*)ws;
    const int ctot = g.S * g.C;
    const unsigned tiles = (unsigned)(rows * ctot);
    if (g.S == 1) {
        launch_cols_fwd<0>(g, z, cP, buf, rows, st);
        hipLaunchKernelGGL(czt_rows_kernel<false>, dim3(tiles), dim3(TILE_T), TILE_LDS_BYTES, st, buf, sP, (float2*)nullptr,
                           ctot, tw);
        launch_cols_inv<0>(g, buf, cP, cQ, nullptr, 0, 0, 0, rows, st);
        launch_cols_fwd<1>(g, nullptr, nullptr, buf, rows, st);
        hipLaunchKernelGGL(czt_rows_kernel<false>, dim3(tiles), dim3(TILE_T), TILE_LDS_BYTES, st, buf, sQ, (float2*)nullptr,
                           ctot, tw);
        launch_cols_inv<1>(g, buf, cP, cQ, y, ldy, lo, len, rows, st);
        return hipGetLastError() == hipSuccess ? GFX_OK : GFX_ELAUNCH;
    }
    // outer radix-4 level around four 2^18-point transforms per row
    const dim3 og((unsigned)(g.NFFT / 4 / 256), (unsigned)rows);
    hipLaunchKernelGGL(czt_outer_fwd_kernel<0>, og, dim3(256), 0, st, z, cP, buf, g);
    launch_cols_fwd<1>(g, nullptr, nullptr, buf, rows, st);
    hipLaunchKernelGGL(czt_rows_kernel<false>, dim3(tiles), dim3(TILE_T), TILE_LDS_BYTES, st, buf, sP, (float2*)nullptr, ctot,
                       tw);
    launch_cols_inv<2>(g, buf, cP, cQ, nullptr, 0, 0, 0, rows, st);
    hipLaunchKernelGGL(czt_outer_inv_kernel<0>, og, dim3(256), 0, st, buf, cP, cQ, (float*)nullptr, (int64_t)0, (int64_t)0,
                       (int64_t)0, g);
    hipLaunchKernelGGL(czt_outer_fwd_kernel<1>, og, dim3(256), 0, st, (const float*)nullptr, (const float2*)nullptr, buf, g);
    launch_cols_fwd<1>(g, nullptr, nullptr, buf, rows, st);
    hipLaunchKernelGGL(czt_rows_kernel<false>, dim3(tiles), dim3(TILE_T), TILE_LDS_BYTES, st, buf, sQ, (float2*)nullptr, ctot,
                       tw);
    launch_cols_inv<2>(g, buf, cP, cQ, nullptr, 0, 0, 0, rows, st);
    hipLaunchKernelGGL(czt_outer_inv_kernel<1>, og, dim3(256), 0, st, buf, cP, cQ, y, ldy, lo, len, g);
    return hipGetLastError() == hipSuccess ? GFX_OK : GFX_ELAUNCH;
}

}  // extern "C"
