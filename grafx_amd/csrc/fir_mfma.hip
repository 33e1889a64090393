// Short FIR filters as batched Toeplitz GEMMs on the fp32 matrix cores (gfx950).
//
// The north star's "MFMA only for the FIR-as-batched-GEMM path": what the reference reaches through
// FlashFFTConv's matrix-unit convolution (core/convolution.py:60-65, 85-106) and needs for its short-filter
// processors (filter.py:34-39 FIRFilter, the small-bin ZeroPhaseFIREqualizer, multitap delays).  For a handful of
// taps a 16384-sample FFT tile is mostly overhead; a direct form is 2 N flops per sample and the matrix cores do it in
// exact fp32 (v_mfma_f32_16x16x4_f32 == an fmaf chain, MI355X_MICROARCH.md).
//
//   y[n] = sum_k h[k] x[n + off - k],  per row its own filter.
//
// Outputs are tiled 256 at a time as Y[i][r] = y[n0 + 16 i + r].  With u = k - r + 15 (0 <= u < N + 15):
//   Y = A B,   A[i][u] = x[n0 + off + 16 i + 15 - u]              (16 x (N+15): the signal window, reversed)
//              B[u][r] = h[u + r - 15] if 0 <= u + r - 15 < N else 0   ((N+15) x 16: one Toeplitz band)
// i.e. ceil((N + 15) / 4) MFMA k-steps per output tile: N / (N + 15) of the issued flops are useful (the band's
// triangular ends are the only padding).  Both operands come from LDS: the row's window of x in 16-sample blocks
// padded to 17 floats (the A fragment's lanes are 16 samples apart), the taps once per workgroup with 15 zeros on
// either side so that B needs no mask.  One workgroup = 4 waves x 4 output tiles = 4096 samples of one row-channel; a
// B fragment is read once and used for the wave's 4 tiles.
//
// Roofline: MFMA-bound above ~64 taps (N / 64 matrix-core cycles per output sample and CU), HBM-bound below; the FFT
// tile kernel takes over beyond the crossover measured in profiles/r2/fir_mfma_crossover.txt.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/grafx_amd.h"

namespace gfx {

constexpr int FM_T = 256;                  // threads per workgroup (4 waves)
constexpr int FM_SEG = 4096;               // output samples per workgroup
constexpr int FM_MAXQ = 32;                // N <= 16 * FM_MAXQ = 512 taps

using f32x4 = float __attribute__((ext_vector_type(4)));

struct FirArgs {
    gfx_rowmap_t xmap, ymap;
    int64_t L, Lout, off, N;
    int nub, Cin, Cf, Cout, nseg;   // nub = ceil((N + 15) / 16) blocks of 16 along u
    unsigned hrows;
};

__device__ __forceinline__ int64_t fm_row_off(const gfx_rowmap_t& m, unsigned r, int c) {
    const unsigned inner = (unsigned)m.inner;
    const unsigned q = r / inner, rem = r - q * inner;
    return (int64_t)q * m.stride_outer + (int64_t)rem * m.stride_inner + (int64_t)c * m.stride_ch;
}

__global__ __launch_bounds__(FM_T) void fir_mfma_kernel(const float* __restrict__ x, const float* __restrict__ h,
                                                        float* __restrict__ y, FirArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nub = a.nub;
    const int nblk = FM_SEG / 16 + nub;                 // 16-sample blocks of the x window (each padded to 17 floats)
    float* xs = smem;                                   // [nblk][17]
    float* hs = smem + nblk * 17;                       // [16 nub + 16]: hs[15 + k] = h[k], zeros around
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned b = blockIdx.x;
    const unsigned rco = b / (unsigned)a.nseg;
    const int seg = (int)(b - rco * (unsigned)a.nseg);
    const unsigned r = rco / (unsigned)a.Cout;
    const int c = (int)(rco - r * (unsigned)a.Cout);
    const float* xrow = x + fm_row_off(a.xmap, r, a.Cin == 1 ? 0 : c);
    float* yrow = y + fm_row_off(a.ymap, r, c);
    const float* hrow = h + ((int64_t)(r % a.hrows) * a.Cf + (a.Cf == 1 ? 0 : c)) * a.N;
    const int64_t seg0 = (int64_t)seg * FM_SEG;
    const int64_t xbase = seg0 + a.off - 16 * (nub - 1);      // signal index of window position 0

    for (int i = tid; i < 16 * nub + 16; i += FM_T) {
        const int k = i - 15;
        hs[i] = (k >= 0 && k < a.N) ? hrow[k] : 0.0f;
    }
    // signal window -> 17-float blocks, zero outside [0, L)
    const int W = nblk * 16;
    for (int m = 4 * tid; m < W; m += 4 * FM_T) {
        const int64_t n = xbase + m;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (n + e >= 0 && n + e < a.L) ? xrow[n + e] : 0.0f;
        float* dst = xs + (m >> 4) * 17 + (m & 15);
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[e] = v[e];
    }
    __syncthreads();

    const int i16 = lane & 15, g = lane >> 4;
    // A fragment (signal) of tile T, u = 16 ub + 4 kap + g:  xs[17 (64 wave + 16 T + i + nub-1-ub) + 15 - 4 kap - g]
    const float* xa = xs + 17 * (64 * wave + i16 + nub - 1) + 3 - g;
    // B fragment (taps): hs[u + r] = hs[16 ub + 4 kap + g + r]
    const float* hb = hs + (i16 + g);
    f32x4 acc[4];
#pragma unroll
    for (int T = 0; T < 4; ++T) acc[T] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    for (int ub = 0; ub < nub; ++ub) {
        const float* xq = xa - 17 * ub;
        const float* hq = hb + 16 * ub;
#pragma unroll
        for (int kap = 0; kap < 4; ++kap) {
            const float bv = hq[4 * kap];
#pragma unroll
            for (int T = 0; T < 4; ++T) {
                const float av = xq[17 * 16 * T + 12 - 4 * kap];
                acc[T] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[T], 0, 0, 0);
            }
        }
    }
    // D[4 g + v][r] = y[n0 + 16 (4 g + v) + r]
#pragma unroll
    for (int T = 0; T < 4; ++T) {
        const int64_t n0 = seg0 + 256 * (4 * wave + T);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int64_t n = n0 + 16 * (4 * g + v) + i16;
            if (n < a.Lout) yrow[n] = acc[T][v];
        }
    }
}

}  // namespace gfx

using namespace gfx;

extern "C" int64_t gfx_fir_direct_max_taps(void) { return 16 * FM_MAXQ; }

extern "C" int gfx_fir_direct_f32(const float* x, gfx_rowmap_t xmap, const float* h, int64_t h_rows, float* y,
                                  gfx_rowmap_t ymap, int64_t R, int64_t C_in, int64_t C_f, int64_t L, int64_t Lout,
                                  int64_t off, int64_t N, void* stream) {
    if (!x || !h || !y || R <= 0 || L <= 0 || Lout <= 0 || N <= 0 || N > 16 * FM_MAXQ) return GFX_EINVAL;
    if (h_rows < 1 || h_rows > R || h_rows > 0x7fffffffLL) return GFX_EINVAL;
    if (C_in < 1 || C_f < 1 || (C_in != C_f && C_in != 1 && C_f != 1)) return GFX_EINVAL;
    if (xmap.inner <= 0 || ymap.inner <= 0 || xmap.inner > 0x7fffffffLL || ymap.inner > 0x7fffffffLL) return GFX_EINVAL;
    if (off < -(int64_t(1) << 40) || off > (int64_t(1) << 40)) return GFX_EINVAL;
    FirArgs a;
    a.xmap = xmap;
    a.ymap = ymap;
    a.L = L;
    a.Lout = Lout;
    a.off = off;
    a.N = N;
    a.nub = (int)((N + 15 + 15) / 16);
    a.Cin = (int)C_in;
    a.Cf = (int)C_f;
    a.Cout = (int)(C_in > C_f ? C_in : C_f);
    a.nseg = (int)((Lout + FM_SEG - 1) / FM_SEG);
    a.hrows = (unsigned)h_rows;
    const int64_t blocks = R * a.Cout * a.nseg;
    if (blocks > 0x7fffffffLL) return GFX_EINVAL;
    const size_t lds = (size_t)((FM_SEG / 16 + a.nub) * 17 + 16 * a.nub + 16) * sizeof(float);
    hipLaunchKernelGGL(fir_mfma_kernel, dim3((unsigned)blocks), dim3(FM_T), lds, (hipStream_t)stream, x, h, y, a);
    return hipGetLastError() == hipSuccess ? GFX_OK : GFX_ELAUNCH;
}
