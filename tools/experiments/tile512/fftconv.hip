// Overlap-save FIR convolution on LDS FFT tiles (gfx950).
//
// Replaces the reference's convolve() — core/convolution.py:119-134 — for the case where
// it equals a true linear convolution (P = L + N - 1 even; see DESIGN.md for the odd-P quirk,
// which is layered on top of the *full* convolution this file produces).
//
// Kernels
//   hspec_kernel    taps -> per-partition tile spectra (He, Ho pairs in thread layout)
//   fftconv1_kernel N <= 8193: load x tile -> FFT -> x H -> IFFT -> store valid samples (fused)
//   xspec_kernel    N  > 8193: x window -> FFT -> spectra to HBM/L2
//   macinv_kernel   N  > 8193: sum_p X[i-p] * H[p] in registers -> IFFT -> store
//
// Algorithmic bytes: 4*(C_in + C_out) per output frame per row (read x once, write y once);
// the tile overlap (N-1 of 16384 samples) is re-read through L2.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/grafx_amd.h"
#include "fft_tile.hpp"

namespace gfx {

struct ConvGeom {
    int64_t nparts, part_len, O, V, ntiles;
};

static inline ConvGeom conv_geom(int64_t N, int64_t Lout) {
    ConvGeom g;
    if (N <= TILE_M + 1) {
        g.nparts = 1;
        g.part_len = N;
        g.O = (N - 1 + 1) & ~int64_t(1);  // overlap >= N-1, even so tiles stay 8-byte aligned
    } else {
        g.part_len = TILE_M;
        g.nparts = (N + TILE_M - 1) / TILE_M;
        g.O = TILE_M;
    }
    g.V = TILE_F - g.O;
    g.ntiles = (Lout + g.V - 1) / g.V;
    return g;
}

struct ConvArgs {
    gfx_rowmap_t xmap, ymap;
    int64_t L, Lout, off;    // signal length, outputs per row, output offset into the full convolution
    int64_t O, V;            // overlap and valid samples per tile
    int64_t ntiles, nblocks; // tiles per row-channel, total workgroups of real work
    int nparts;
    int Cin, Cf, Cout;
};

// rows and row counts fit 32 bits (checked by the launchers): 32-bit division is ~5x cheaper than
// the 64-bit software divide and stays on the scalar unit.
__device__ __forceinline__ int64_t row_off(const gfx_rowmap_t& m, unsigned r, int c) {
    const unsigned inner = (unsigned)m.inner;
    const unsigned q = r / inner, rem = r - q * inner;
    return (int64_t)q * m.stride_outer + (int64_t)rem * m.stride_inner + (int64_t)c * m.stride_ch;
}

// workgroup b runs on XCD b % 8 (observed): give each XCD a contiguous run of logical
// indices so tiles of one row-channel (which share the filter spectrum) meet in one L2.
__device__ __forceinline__ unsigned xcd_logical_block() {
    const unsigned per_xcd = gridDim.x >> 3;
    return (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
}

// v[a] = (x[s + 2m], x[s + 2m + 1]), m = tile_point(t, a); x is zero outside [0, L)
__device__ __forceinline__ void load_window(float2 (&v)[16], const float* __restrict__ row, int64_t s, int64_t L,
                                            int t, float gain) {
    const bool inside = s >= 0 && s + TILE_F <= L;
    const bool aligned = ((reinterpret_cast<uintptr_t>(row + s)) & 7) == 0;
    if (inside && aligned) {
        // streamed once: non-temporal so the signal does not evict the filter spectra from L2
        const double* p = reinterpret_cast<const double*>(row + s);
#pragma unroll
        for (int a = 0; a < 16; ++a) {
#ifdef GFX_NO_NT
            const double raw = p[tile_point(t, a)];
#else
            const double raw = __builtin_nontemporal_load(p + tile_point(t, a));
#endif
            const float2 e = *reinterpret_cast<const float2*>(&raw);
            v[a] = make_float2(e.x * gain, e.y * gain);
        }
    } else {
#pragma unroll
        for (int a = 0; a < 16; ++a) {
            const int64_t n = s + 2 * tile_point(t, a);
            const float e0 = (n >= 0 && n < L) ? row[n] : 0.0f;
            const float e1 = (n + 1 >= 0 && n + 1 < L) ? row[n + 1] : 0.0f;
            v[a] = make_float2(e0 * gain, e1 * gain);
        }
    }
}

// y[n0 + (2m - O)] for 2m >= O, n < Lout;  v[a] = (z'[2m], z'[2m+1]), m = tile_point(t, a)
__device__ __forceinline__ void store_valid(const float2 (&v)[16], float* __restrict__ row, int64_t n0, int64_t O,
                                            int64_t Lout, int t) {
    const bool aligned = ((reinterpret_cast<uintptr_t>(row + n0)) & 7) == 0 && (O & 1) == 0;
    const bool whole = n0 + (TILE_F - O) <= Lout;
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        const int64_t q = 2 * tile_point(t, a);
        if (q < O) continue;
        const int64_t n = n0 + q - O;
        if (whole && aligned) {
#ifdef GFX_NO_NT
            *reinterpret_cast<float2*>(row + n) = v[a];
#else
            __builtin_nontemporal_store(*reinterpret_cast<const double*>(&v[a]), reinterpret_cast<double*>(row + n));
#endif
        } else {
            if (n < Lout) row[n] = v[a].x;
            if (n + 1 < Lout) row[n + 1] = v[a].y;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// taps -> {alpha, beta} per bin in the thread layout (see spectral_product)
__global__ __launch_bounds__(TILE_T, 4) void hspec_kernel(const float* __restrict__ h, const float* __restrict__ gain,
                                                          int64_t gain_div, float4* __restrict__ Hs, int64_t N,
                                                          int nparts, int64_t part_len,
                                                          const float2* __restrict__ twtab) {
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int t = threadIdx.x;
    const unsigned b = blockIdx.x;
    const unsigned rc = b / (unsigned)nparts;
    const int p = (int)(b - rc * (unsigned)nparts);
    const int64_t start = (int64_t)p * part_len;
    const int64_t len = min(part_len, N - start);
    const float g = gain ? gain[rc / (unsigned)gain_div] : 1.0f;

    float2 v[16], w[16], q[16];
    load_window(v, h + (int64_t)rc * N + start, 0, len, t, g);
    tile_forward(v, w, twtab, lds, t);
    tile_mirror(w, q, t);

    float4* out = Hs + (int64_t)b * H_TILE_F4;
    const float sc = 1.0f / (2.0f * TILE_M);
    const float2 wj = tile_wj(twtab, t);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float2 A = w[r], Bc = cconj(q[r]);
        const float2 he2 = cadd(A, Bc);                      // 2 He
        const float2 ho = mul_neg_i(csub(A, Bc));            // 2 Ho
        const float2 wk = cmul(wj, w16(brev(r, 4)));      // W_M^k, k = j + 512*k3
        const float2 iho = mul_pos_i(ho);                    // 2 i Ho
        // alpha = (2He + i(1-W)Ho)/(2M), beta = i(1+W)Ho/(2M), with ho = 2Ho: halve the Ho terms
        const float2 wiho = cmul(wk, iho);
        const float2 al = make_float2((he2.x + 0.5f * (iho.x - wiho.x)) * sc, (he2.y + 0.5f * (iho.y - wiho.y)) * sc);
        const float2 be = make_float2(0.5f * (iho.x + wiho.x) * sc, 0.5f * (iho.y + wiho.y) * sc);
        out[r * TILE_T + t] = make_float4(al.x, al.y, be.x, be.y);
    }
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TILE_T, 4) void fftconv1_kernel(const float* __restrict__ x, const float4* __restrict__ Hs,
                                                             float* __restrict__ y, ConvArgs a,
                                                             const float2* __restrict__ twtab) {
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int t = threadIdx.x;
    const unsigned lb = xcd_logical_block();
    if (lb >= (unsigned)a.nblocks) return;
    const unsigned ntiles = (unsigned)a.ntiles;
    const unsigned rco = lb / ntiles;
    const int64_t tile = lb - rco * ntiles;
    const unsigned r = rco / (unsigned)a.Cout;
    const int c = (int)(rco - r * (unsigned)a.Cout);
    const float* xrow = x + row_off(a.xmap, r, a.Cin == 1 ? 0 : c);
    float* yrow = y + row_off(a.ymap, r, c);
    const float4* H = Hs + ((int64_t)r * a.Cf + (a.Cf == 1 ? 0 : c)) * H_TILE_F4;

    float2 v[16], w[16], q[16];
#ifdef GFX_NOX
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = make_float2((float)(t + i), (float)(t - i));
#else
    load_window(v, xrow, a.off + tile * a.V - a.O, a.L, t, 1.0f);
#endif
#if !defined(GFX_ABLATE) || GFX_ABLATE == 1
    tile_forward_a(v, twtab, lds, t);
#endif
    float4 hA[8];  // first half of the row's spectrum: in flight under the last forward pass
#pragma unroll
#ifdef GFX_NOH
    for (int i = 0; i < 8; ++i) hA[i] = make_float4(1.0f, 0.0f, 0.5f, 0.25f);
#else
    for (int i = 0; i < 8; ++i) hA[i] = H[i * TILE_T + t];
#endif
#if !defined(GFX_ABLATE) || GFX_ABLATE == 1
    tile_forward_b(w, lds, t);
#else
#pragma unroll
    for (int i = 0; i < 16; ++i) w[i] = v[i];
#endif
    tile_mirror(w, q, t);
    float4 hB[8];
#pragma unroll
#ifdef GFX_NOH
    for (int i = 0; i < 8; ++i) hB[i] = make_float4(1.0f, 0.0f, 0.5f, 0.25f);
#else
    for (int i = 0; i < 8; ++i) hB[i] = H[(8 + i) * TILE_T + t];
#endif
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = spectral_product(w[i], q[i], hA[i]);
#pragma unroll
    for (int i = 0; i < 8; ++i) w[8 + i] = spectral_product(w[8 + i], q[8 + i], hB[i]);
    __syncthreads();  // every thread is done reading S2 before the inverse overwrites it
#if !defined(GFX_ABLATE) || GFX_ABLATE == 2
    tile_inverse(w, v, twtab, lds, t);
#else
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = w[i];
#endif
#ifdef GFX_NOST
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += v[i].x + v[i].y;
    if (acc == 12345.678f) yrow[t] = acc;
#else
    store_valid(v, yrow, tile * a.V, a.O, a.Lout, t);
#endif
}

// ------------------------------------------------------------------------------------------------
// window j (j = jj - (nparts-1)) of x starts at off - O + j*V; windows that miss [0, L) are skipped.
__device__ __forceinline__ bool window_live(int64_t s, int64_t L) { return s + TILE_F > 0 && s < L; }

__global__ __launch_bounds__(TILE_T, 4) void xspec_kernel(const float* __restrict__ x, float2* __restrict__ Zs,
                                                          ConvArgs a, int64_t nwin,
                                                          const float2* __restrict__ twtab) {
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int t = threadIdx.x;
    const unsigned lb = xcd_logical_block();
    if (lb >= (unsigned)a.nblocks) return;
    const unsigned rcx = lb / (unsigned)nwin;
    const int64_t jj = lb - rcx * (unsigned)nwin;
    const int64_t s = a.off - a.O + (jj - (a.nparts - 1)) * a.V;
    if (!window_live(s, a.L)) return;
    const unsigned xr = rcx / (unsigned)a.Cin;
    const float* xrow = x + row_off(a.xmap, xr, (int)(rcx - xr * (unsigned)a.Cin));
    float2 v[16], w[16];
    load_window(v, xrow, s, a.L, t, 1.0f);
    tile_forward(v, w, twtab, lds, t);
    float2* out = Zs + (int64_t)lb * TILE_M;
#pragma unroll
    for (int q = 0; q < 16; ++q) out[q * TILE_T + t] = w[q];
}

__global__ __launch_bounds__(TILE_T, 4) void macinv_kernel(const float2* __restrict__ Zs, const float4* __restrict__ Hs,
                                                           float* __restrict__ y, ConvArgs a, int64_t nwin,
                                                           const float2* __restrict__ twtab) {
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int t = threadIdx.x;
    const unsigned lb = xcd_logical_block();
    if (lb >= (unsigned)a.nblocks) return;
    const unsigned ntiles = (unsigned)a.ntiles;
    const unsigned rco = lb / ntiles;
    const int64_t tile = lb - rco * ntiles;
    const unsigned r = rco / (unsigned)a.Cout;
    const int c = (int)(rco - r * (unsigned)a.Cout);
    float* yrow = y + row_off(a.ymap, r, c);
    const float4* H = Hs + ((int64_t)r * a.Cf + (a.Cf == 1 ? 0 : c)) * a.nparts * H_TILE_F4;
    const float2* Z = Zs + ((int64_t)r * a.Cin + (a.Cin == 1 ? 0 : c)) * nwin * TILE_M;

    float2 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = make_float2(0.0f, 0.0f);

    for (int p = 0; p < a.nparts; ++p) {
        const int64_t j = tile - p;
        if (!window_live(a.off - a.O + j * a.V, a.L)) continue;
        const float2* Zj = Z + (j + a.nparts - 1) * TILE_M;
        const float4* Hp = H + (int64_t)p * H_TILE_F4;
        float2 w[16], q[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) w[i] = Zj[i * TILE_T + t];
        tile_mirror(w, q, t);
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = cadd(acc[i], spectral_product(w[i], q[i], Hp[i * TILE_T + t]));
    }

    float2 v[16];
    tile_inverse(acc, v, twtab, lds, t);
    store_valid(v, yrow, tile * a.V, a.O, a.Lout, t);
}

static inline unsigned pad8(int64_t n) { return (unsigned)(((n + 7) / 8) * 8); }

template <typename K>
static int allow_lds(K kernel) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               TILE_LDS_BYTES) == hipSuccess
               ? 0
               : GFX_ELAUNCH;
}

}  // namespace gfx

using namespace gfx;

extern "C" {

int64_t gfx_fftconv_nparts(int64_t N) { return N <= 0 ? 0 : conv_geom(N, 1).nparts; }

size_t gfx_fir_spectrum_bytes(int64_t RCf, int64_t N) {
    if (RCf <= 0 || N <= 0) return 0;
    return (size_t)RCf * conv_geom(N, 1).nparts * H_TILE_F4 * sizeof(float4);
}

size_t gfx_fftconv_workspace_bytes(int64_t R, int64_t C_in, int64_t L, int64_t Lout, int64_t off, int64_t N) {
    (void)L;
    (void)off;
    if (R <= 0 || N <= 0 || Lout <= 0) return 0;
    const ConvGeom g = conv_geom(N, Lout);
    if (g.nparts == 1) return 0;
    return (size_t)R * C_in * (g.ntiles + g.nparts - 1) * TILE_M * sizeof(float2);
}

int gfx_fir_spectrum_f32(const float* h, const float* gain, int64_t gain_div, void* Hs, int64_t RCf, int64_t N,
                         void* stream) {
    if (!h || !Hs || RCf <= 0 || N <= 0 || (gain && gain_div <= 0)) return GFX_EINVAL;
    const ConvGeom g = conv_geom(N, 1);
    if (RCf * g.nparts > 0x7fffffffLL) return GFX_EINVAL;
    if (allow_lds(hspec_kernel)) return GFX_ELAUNCH;
    const float2* tw = tile_twiddle_table((hipStream_t)stream);
    if (!tw) return GFX_ELAUNCH;
    hipLaunchKernelGGL(hspec_kernel, dim3((unsigned)(RCf * g.nparts)), dim3(TILE_T), TILE_LDS_BYTES,
                       (hipStream_t)stream, h, gain, gain_div, (float4*)Hs, N, (int)g.nparts, g.part_len, tw);
    return hipGetLastError() == hipSuccess ? GFX_OK : GFX_ELAUNCH;
}

int gfx_fftconv_f32(const float* x, gfx_rowmap_t xmap, const void* Hs, float* y, gfx_rowmap_t ymap, int64_t R,
                    int64_t C_in, int64_t C_f, int64_t L, int64_t Lout, int64_t off, int64_t N, void* ws,
                    size_t ws_bytes, void* stream) {
    if (!x || !Hs || !y || R <= 0 || L <= 0 || Lout <= 0 || N <= 0) return GFX_EINVAL;
    if (C_in < 1 || C_f < 1 || (C_in != C_f && C_in != 1 && C_f != 1)) return GFX_EINVAL;
    if (xmap.inner <= 0 || ymap.inner <= 0 || xmap.inner > 0x7fffffffLL || ymap.inner > 0x7fffffffLL) return GFX_EINVAL;
    const ConvGeom g = conv_geom(N, Lout);
    ConvArgs a;
    a.xmap = xmap;
    a.ymap = ymap;
    a.L = L;
    a.Lout = Lout;
    a.off = off;
    a.O = g.O;
    a.V = g.V;
    a.ntiles = g.ntiles;
    a.nparts = (int)g.nparts;
    a.Cin = (int)C_in;
    a.Cf = (int)C_f;
    a.Cout = (int)(C_in > C_f ? C_in : C_f);
    a.nblocks = R * a.Cout * g.ntiles;
    if (a.nblocks > 0x7ffffff0LL) return GFX_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const float2* tw = tile_twiddle_table(st);
    if (!tw) return GFX_ELAUNCH;

    if (g.nparts == 1) {
        if (allow_lds(fftconv1_kernel)) return GFX_ELAUNCH;
        hipLaunchKernelGGL(fftconv1_kernel, dim3(pad8(a.nblocks)), dim3(TILE_T), TILE_LDS_BYTES, st, x,
                           (const float4*)Hs, y, a, tw);
        return hipGetLastError() == hipSuccess ? GFX_OK : GFX_ELAUNCH;
    }
    const int64_t nwin = g.ntiles + g.nparts - 1;
    const size_t need = (size_t)R * C_in * nwin * TILE_M * sizeof(float2);
    if (!ws || ws_bytes < need) return GFX_ENOSPC;
    if (allow_lds(xspec_kernel) || allow_lds(macinv_kernel)) return GFX_ELAUNCH;
    ConvArgs ax = a;
    ax.nblocks = R * C_in * nwin;
    if (ax.nblocks > 0x7ffffff0LL) return GFX_EINVAL;
    hipLaunchKernelGGL(xspec_kernel, dim3(pad8(ax.nblocks)), dim3(TILE_T), TILE_LDS_BYTES, st, x, (float2*)ws, ax,
                       nwin, tw);
    hipLaunchKernelGGL(macinv_kernel, dim3(pad8(a.nblocks)), dim3(TILE_T), TILE_LDS_BYTES, st, (const float2*)ws,
                       (const float4*)Hs, y, a, nwin, tw);
    return hipGetLastError() == hipSuccess ? GFX_OK : GFX_ELAUNCH;
}

}  // extern "C"
