// STFT-masked noise reverb: impulse-response synthesis.
//
// Replaces STFTMaskedNoiseReverb.compute_stft_mask / compute_ir (reverb.py:161-200, torch.istft),
// ms_to_lr (core/midside.py:4-8) and normalize_impulse (core/utils.py:14-18):
//   M[r,c,k,m] = exp((H0[r,c,k] - softplus(Hd[r,c,k]) * m [+ G[r,c,m]]) / 8)
//   ir[r,c]    = istft(V[c] * M[r,c])             (n_fft, hop, window; centred, trimmed to ir_len)
//   [ir = (mid+side, mid-side)]                   ("pseudo_midside")
//   gain[r]    = 1 / sqrt(mean_c sum_t ir^2 + 1e-12)   (applied when the taps are turned into spectra)
//
// The per-frame inverse real DFT (n_fft = 384 = 2^7*3 by default) is GEMM-shaped:
//   frames[(r,c,m), n] = sum_kk A[(r,c,m), kk] * Basis[kk, n],   kk = (bin, re|im)
// with the window and the 1/n_fft, Hermitian weights folded into Basis.  It runs on the fp32
// matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 FMA chains); the A operand (masked noise
// spectrum) is generated in registers, never stored.  A second kernel overlap-adds the frames,
// divides by the window envelope, applies mid/side -> left/right and accumulates the energy.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "../../include/grafx_amd.h"
#include "small_dft.hpp"

namespace gfx {

using f32x4 = __attribute__((ext_vector_type(4))) float;

__host__ __device__ inline int64_t kpad_of(int64_t n_fft) { return ((2 * (n_fft / 2 + 1)) + 3) / 4 * 4; }

// Basis[kk][n]: kk = 2k -> coefficient of Re S[k], kk = 2k+1 -> coefficient of Im S[k]; window folded in.
__global__ void istft_basis_kernel(const float* __restrict__ window, float* __restrict__ basis, int n_fft, int kpad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)kpad * n_fft) return;
    const int kk = (int)(i / n_fft), n = (int)(i % n_fft);
    const int k = kk >> 1, half = n_fft / 2;
    float v = 0.0f;
    if (k <= half) {
        const bool edge = (k == 0 || k == half);
        const int r = (int)(((int64_t)k * n) % n_fft);
        float s, c;
        sincospif(2.0f * (float)r / (float)n_fft, &s, &c);
        const float wgt = (edge ? 1.0f : 2.0f) / (float)n_fft;
        v = (kk & 1) ? (edge ? 0.0f : -wgt * s) : wgt * c;   // c2r ignores Im of DC / Nyquist
        v *= window[n];
    }
    basis[i] = v;
}

// Half basis behind the full one (same buffer): hb[k][p][n], n < NC = roundup(n_fft/2 + 1, 16), p = 0: (1|2)/N cos(2 pi k n / N),
// p = 1: -(1|2)/N sin(2 pi k n / N), NOT windowed.  Columns n and N - n of the inverse real DFT share these products
// (cos is even in n, sin odd), so frame[n] = w[n] (C + S), frame[N-n] = w[N-n] (C - S): half the matrix multiply.
__host__ __device__ inline int64_t half_cols(int64_t n_fft) { return ((n_fft / 2 + 1) + 15) / 16 * 16; }
__global__ void istft_half_basis_kernel(float* __restrict__ hb, int n_fft, int NC) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int K = n_fft / 2 + 1;
    if (i >= (int64_t)K * 2 * NC) return;
    const int n = (int)(i % NC), p = (int)((i / NC) & 1), k = (int)(i / (2 * NC));
    float v = 0.0f;
    if (n <= n_fft / 2) {
        const bool edge = (k == 0 || k == n_fft / 2);
        const int r = (int)(((int64_t)k * n) % n_fft);
        float sn, cs;
        sincospif(2.0f * (float)r / (float)n_fft, &sn, &cs);
        const float wgt = (edge ? 1.0f : 2.0f) / (float)n_fft;
        v = p ? (edge ? 0.0f : -wgt * sn) : wgt * cs;
    }
    hb[i] = v;
}

__device__ __forceinline__ float softplus_t(float v) { return v > 20.0f ? v : log1pf(expf(v)); }

struct IstftArgs {
    int64_t R;
    int n_fft, hop, T, K, kpad;   // K = n_fft/2+1 bins, T frames
    int64_t ir_len;
    int64_t nstride;              // floats between the noise spectra of consecutive rows (0: one spectrum shared by all)
};

// grid: (column chunks of 128, frame tiles of 64, R*2); 4 waves, each 16 frames x 128 columns.
__global__ __launch_bounds__(256) void istft_frames_kernel(const float* __restrict__ noise_stft,  // (2,K,T,2)
                                                           const float* __restrict__ init_lm,     // (R,2,K)
                                                           const float* __restrict__ delta_lm,    // (R,2,K)
                                                           const float* __restrict__ gain_env,    // (R,2,T) or null
                                                           const float* __restrict__ basis,       // (kpad,n_fft)
                                                           float* __restrict__ frames,            // (R*2,T,n_fft)
                                                           IstftArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t rc = blockIdx.z;
    const int c = (int)(rc & 1);
    const int col0 = blockIdx.x * 128;
    const int m = blockIdx.y * 64 + wave * 16 + (lane & 15);   // this lane's A row (frame)
    const int kq = lane >> 4;
    const bool m_ok = m < a.T;
    const float* H0 = init_lm + rc * a.K;
    const float* Hd = delta_lm + rc * a.K;
    const float genv = (gain_env && m_ok) ? gain_env[rc * a.T + m] : 0.0f;
    const float mf = (float)m;

    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    const int ncol = (a.n_fft - col0 + 15) / 16;  // live column tiles in this chunk (<= 8)

    for (int kk0 = 0; kk0 < a.kpad; kk0 += 4) {
        const int kk = kk0 + kq, k = kk >> 1;
        float av = 0.0f;
        if (m_ok && k < a.K) {
            // reverb.py:192-199: mask = exp((init + (-softplus(delta)) * m [+ gain_env]) / 8)
            const float slope = -softplus_t(Hd[k]);
            float lm = __fadd_rn(H0[k], __fmul_rn(slope, mf));
            if (gain_env) lm = __fadd_rn(lm, genv);
            const float mask = expf(lm / 8.0f);
            av = noise_stft[(rc >> 1) * a.nstride + (((int64_t)c * a.K + k) * a.T + m) * 2 + (kk & 1)] * mask;
        }
        const float* brow = basis + (int64_t)kk * a.n_fft + col0 + (lane & 15);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j < ncol) {
                const int col = col0 + 16 * j + (lane & 15);
                const float bv = col < a.n_fft ? brow[16 * j] : 0.0f;
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[j], 0, 0, 0);
            }
        }
    }
    // C layout: row = (lane >> 4) * 4 + reg, col = lane & 15
    const int mrow0 = blockIdx.y * 64 + wave * 16 + (lane >> 4) * 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int col = col0 + 16 * j + (lane & 15);
        if (j < ncol && col < a.n_fft) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int mm = mrow0 + q;
                if (mm < a.T) frames[(rc * a.T + mm) * a.n_fft + col] = acc[j][q];
            }
        }
    }
}

// Block-tiled variant for n_fft <= 384: one workgroup = 64 frames x all columns.  The A operand (masked
// noise spectrum, one exp per bin) is generated ONCE per 32-deep k-chunk into LDS and shared by the 4
// waves; each wave owns a 96-column slab (6 column tiles) for all 4 row tiles -> 24 MFMAs per k-step
// against 4 LDS reads + 6 basis loads (the per-wave kernel above does 8 MFMAs per 9 loads and
// regenerates A three times).
constexpr int IST_KC = 32;           // k-depth per LDS chunk (8 MFMA k-steps)
constexpr int IST_AROW = 80;         // floats per kk row of the A chunk (64 frames + pad: rows 16 banks apart)
__global__ __launch_bounds__(256) void istft_frames_tiled_kernel(const float* __restrict__ noise_stft,
                                                                 const float* __restrict__ init_lm,
                                                                 const float* __restrict__ delta_lm,
                                                                 const float* __restrict__ gain_env,
                                                                 const float* __restrict__ basis,
                                                                 float* __restrict__ frames, IstftArgs a) {
    __shared__ float As[2][IST_KC * IST_AROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t rc = blockIdx.y;
    const int c = (int)(rc & 1);
    const int m0 = blockIdx.x * 64;
    const float* H0 = init_lm + rc * a.K;
    const float* Hd = delta_lm + rc * a.K;
    // A generation role: frame fm = tid & 63, bins kb = (tid >> 6) * 4 .. +3 of the chunk's 16 bins
    const int fm = tid & 63, m = m0 + fm;
    const bool m_ok = m < a.T;
    const float mf = (float)m;
    const float genv = (gain_env && m_ok) ? gain_env[rc * a.T + m] : 0.0f;

    f32x4 acc[4][6];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 6; ++ct) acc[rt][ct] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

    const int kq = lane >> 4, li = lane & 15;
    const int colw = wave * 96 + li;
    const int nchunks = (a.kpad + IST_KC - 1) / IST_KC;
    for (int ch = 0; ch < nchunks; ++ch) {
        float* A = As[ch & 1];
        const int kk0 = ch * IST_KC;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = (kk0 >> 1) + (tid >> 6) * 4 + q;
            float re = 0.0f, im = 0.0f;
            if (m_ok && k < a.K) {
                const float slope = -softplus_t(Hd[k]);
                float lm = __fadd_rn(H0[k], __fmul_rn(slope, mf));
                if (gain_env) lm = __fadd_rn(lm, genv);
                const float mask = expf(lm / 8.0f);
                const float2 nz = *reinterpret_cast<const float2*>(noise_stft + (rc >> 1) * a.nstride + (((int64_t)c * a.K + k) * a.T + m) * 2);
                re = nz.x * mask;
                im = nz.y * mask;
            }
            const int kkl = 2 * ((tid >> 6) * 4 + q);
            A[kkl * IST_AROW + fm] = re;
            A[(kkl + 1) * IST_AROW + fm] = im;
        }
        __syncthreads();  // chunk ch visible; the other buffer is free again two barriers later
#pragma unroll
        for (int ks = 0; ks < IST_KC / 4; ++ks) {
            const int kk = kk0 + ks * 4 + kq;
            float av[4], bv[6];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) av[rt] = A[(ks * 4 + kq) * IST_AROW + 16 * rt + li];
            const float* brow = basis + (int64_t)kk * a.n_fft + colw;
#pragma unroll
            for (int ct = 0; ct < 6; ++ct) bv[ct] = (kk < a.kpad && colw + 16 * ct < a.n_fft) ? brow[16 * ct] : 0.0f;
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int ct = 0; ct < 6; ++ct)
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt], bv[ct], acc[rt][ct], 0, 0, 0);
        }
    }
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 6; ++ct) {
            const int col = colw + 16 * ct;
            if (col < a.n_fft) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int mm = m0 + 16 * rt + kq * 4 + q;
                    if (mm < a.T) frames[(rc * a.T + mm) * a.n_fft + col] = acc[rt][ct][q];
                }
            }
        }
}

// The tiled kernel on the half basis (n_fft <= 384): per k-step of 4 bins a wave multiplies the real parts into its C tiles
// and the imaginary parts into its S tiles, then writes frame[n] = w[n] (C + S) and frame[N-n] = w[N-n] (C - S):
// 26 column tiles x 49 k-steps instead of 24 x 97.
__global__ __launch_bounds__(256) void istft_frames_half_kernel(const float* __restrict__ noise_stft,
                                                                const float* __restrict__ init_lm,
                                                                const float* __restrict__ delta_lm,
                                                                const float* __restrict__ gain_env,
                                                                const float* __restrict__ hb,
                                                                const float* __restrict__ window,
                                                                float* __restrict__ frames, IstftArgs a) {
    __shared__ float As[2][IST_KC * IST_AROW];
    __shared__ float slope_s[256];   // -softplus(delta) per bin: the same for all 64 frames of the tile (K <= 193)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t rc = blockIdx.y;
    const int c = (int)(rc & 1);
    const int m0 = blockIdx.x * 64;
    const float* H0 = init_lm + rc * a.K;
    const float* Hd = delta_lm + rc * a.K;
    const int fm = tid & 63, m = m0 + fm;
    const bool m_ok = m < a.T;
    const float mf = (float)m;
    const float genv = (gain_env && m_ok) ? gain_env[rc * a.T + m] : 0.0f;
    if (tid < a.K) slope_s[tid] = -softplus_t(Hd[tid]);
    __syncthreads();
    const int NC = (int)half_cols(a.n_fft), ntile = NC / 16;
    // wave w: row tiles 2 (w & 1), +1 (32 of the 64 frames) x one half of the column tiles (7 + 6 of 13 at n_fft = 384):
    // 2 x 7 x {C, S} = 28 accumulator tiles = 112 registers, so that two workgroups fit a CU
    constexpr int HCT = 7;
    const int rt0 = 2 * (wave & 1);
    const int ct0 = (wave >> 1) * ((ntile + 1) / 2);
    const int nct = (wave >> 1) ? ntile - ct0 : ct0 + (ntile + 1) / 2 > ntile ? ntile : (ntile + 1) / 2;

    f32x4 accC[2][HCT], accS[2][HCT];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < HCT; ++ct) accC[rt][ct] = accS[rt][ct] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

    const int kq = lane >> 4, li = lane & 15;
    const int colw = ct0 * 16 + li;
    const int nchunks = (2 * a.K + IST_KC - 1) / IST_KC;
    for (int ch = 0; ch < nchunks; ++ch) {
        float* A = As[ch & 1];
        const int kk0 = ch * IST_KC;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = (kk0 >> 1) + (tid >> 6) * 4 + q;
            float re = 0.0f, im = 0.0f;
            if (m_ok && k < a.K) {
                float lm = __fadd_rn(H0[k], __fmul_rn(slope_s[k], mf));
                if (gain_env) lm = __fadd_rn(lm, genv);
                const float mask = expf(lm / 8.0f);
                const float2 nz = *reinterpret_cast<const float2*>(noise_stft + (rc >> 1) * a.nstride + (((int64_t)c * a.K + k) * a.T + m) * 2);
                re = nz.x * mask;
                im = nz.y * mask;
            }
            const int kkl = 2 * ((tid >> 6) * 4 + q);
            A[kkl * IST_AROW + fm] = re;
            A[(kkl + 1) * IST_AROW + fm] = im;
        }
        __syncthreads();  // chunk ch visible; the other buffer is free again two barriers later
#pragma unroll
        for (int ks = 0; ks < IST_KC / 8; ++ks) {   // 4 bins per step
            const int bl = ks * 4 + kq;              // bin within the chunk
            const int k = (kk0 >> 1) + bl;
            float are[2], aim[2], bc[HCT], bs[HCT];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                are[rt] = A[(2 * bl) * IST_AROW + 16 * (rt0 + rt) + li];
                aim[rt] = A[(2 * bl + 1) * IST_AROW + 16 * (rt0 + rt) + li];
            }
            const float* brow = hb + (int64_t)k * 2 * NC + colw;
#pragma unroll
            for (int ct = 0; ct < HCT; ++ct) {
                const bool ok = k < a.K && ct < nct;
                bc[ct] = ok ? brow[16 * ct] : 0.0f;
                bs[ct] = ok ? brow[NC + 16 * ct] : 0.0f;
            }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < HCT; ++ct)
                    if (ct < nct) {   // (wave-uniform)
                        accC[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(are[rt], bc[ct], accC[rt][ct], 0, 0, 0);
                        accS[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(aim[rt], bs[ct], accS[rt][ct], 0, 0, 0);
                    }
        }
    }
    const int half = a.n_fft / 2;
#pragma unroll
    for (int ct = 0; ct < HCT; ++ct) {
        const int n = colw + 16 * ct;
        if (ct < nct && n <= half) {
            const float wn = window[n], wm = window[(a.n_fft - n) % a.n_fft];
            const bool mirror = n > 0 && n < half;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int mm = m0 + 16 * (rt0 + rt) + kq * 4 + q;
                    if (mm < a.T) {
                        float* fr = frames + (rc * a.T + mm) * a.n_fft;
                        fr[n] = wn * (accC[rt][ct][q] + accS[rt][ct][q]);
                        if (mirror) fr[a.n_fft - n] = wm * (accC[rt][ct][q] - accS[rt][ct][q]);
                    }
                }
        }
    }
}

// overlap-add + envelope division + trim (centre) + optional ms->lr, and the block's share of the row's energy
// sum_c sum_t ir[r,c,t]^2 (core/utils.py:16): partial[r][block], summed in a fixed order (a shuffle tree per wave, the four
// waves in index order) -- the impulse response is not read again for its normalisation.
__global__ __launch_bounds__(256) void istft_ola_kernel(const float* __restrict__ frames, const float* __restrict__ window,
                                                        float* __restrict__ ir, float* __restrict__ partial, IstftArgs a,
                                                        int ms_to_lr) {
    __shared__ float red[4];
    const int64_t r = blockIdx.y;
    const int64_t tp = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // output index after trimming
    float e = 0.0f;
    if (tp < a.ir_len) {
        const int64_t t = tp + a.n_fft / 2;
        int64_t m_hi = t / a.hop;
        if (m_hi > a.T - 1) m_hi = a.T - 1;
        int64_t m_lo = (t - a.n_fft + a.hop) / a.hop;  // ceil((t - n_fft + 1) / hop)
        if (m_lo < 0) m_lo = 0;
        float v0 = 0.0f, v1 = 0.0f, env = 0.0f;
        for (int64_t m = m_lo; m <= m_hi; ++m) {
            const int n = (int)(t - m * a.hop);
            const float w = window[n];
            env = fmaf(w, w, env);
            v0 += frames[((r * 2 + 0) * a.T + m) * a.n_fft + n];
            v1 += frames[((r * 2 + 1) * a.T + m) * a.n_fft + n];
        }
        v0 /= env;
        v1 /= env;
        if (ms_to_lr) {
            const float l = v0 + v1, rr = v0 - v1;
            v0 = l;
            v1 = rr;
        }
        ir[(r * 2 + 0) * a.ir_len + tp] = v0;
        ir[(r * 2 + 1) * a.ir_len + tp] = v1;
        e = fmaf(v1, v1, v0 * v0);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) e += __shfl_down(e, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = e;
    __syncthreads();
    if (threadIdx.x == 0) partial[r * gridDim.x + blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

// row_gain[r] = 1 / sqrt(mean_c sum_t ir[r,c,t]^2 + 1e-12)  (core/utils.py:16-17) from the overlap-add kernel's block
// partials: one wave per row, lane l takes partials l, l + 64, ... in order, then a shuffle tree -- a fixed order, so the
// gain (a factor of every sample of the reverb's output) is the same bits from run to run.
__global__ __launch_bounds__(64) void ir_energy_gain_kernel(const float* __restrict__ partial, float* __restrict__ gain,
                                                            int nblk) {
    const float* p = partial + (int64_t)blockIdx.x * nblk;
    float e = 0.0f;
    for (int b = threadIdx.x; b < nblk; b += 64) e += p[b];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) e += __shfl_down(e, d, 64);
    if (threadIdx.x == 0) gain[blockIdx.x] = 1.0f / sqrtf(0.5f * e + 1e-12f);
}

// ---- n_fft = 384, hop = 192 (the reference's defaults): frames by FFT, overlap-add in LDS --------------------------------
// The matrix-core form above spends 2 x 193 x 384 multiply-adds on a frame; a 384-point inverse real FFT needs ~4000.  At
// cfg3 (512 rows x 2 x 313 frames) that is 95 GFLOP = 0.9 ms of fp32 MFMA against microseconds of vector arithmetic, and
// the frames no longer leave the chip: one workgroup = one row, both channels, FR consecutive frames, finishing the FR - 1
// blocks of `hop` samples that lie between them.
//   x[n] = 1/N sum_k Xh[k] e^(+2 pi i k n / N)  (Xh: the Hermitian extension of the K = 193 bins; Im Xh[0], Im Xh[192] ignored,
//   as torch.istft / irfft do) is computed as one 192-point complex transform of z[j] = x[2j] + i x[2j+1]:
//   Z[k] = E[k] + i O[k],  E[k] = (X[k] + conj X[192-k]) / 2,  O[k] = (X[k] - conj X[192-k]) / 2 * e^(+2 pi i k / 384),
//   z[j] = 1/192 sum_k Z[k] e^(+2 pi i j k / 192).
// 192 = 8 x 24: eight lanes per frame; lane l transforms Z[l + 8 j], j < 24, in registers (small_dft.hpp), multiplies by
// e^(2 pi i b l / 192) and hands the 24 x 8 intermediate values over through LDS; then three 8-point transforms per lane
// give z[24 a + b], b = l, l + 8, l + 16.  The windowed samples are added into an LDS image of the workgroup's output blocks
// (a block = second half of one frame + first half of the next: every LDS word has one owner per pass, no atomics), then
// envelope division, mid/side -> left/right, store and energy partial as in istft_ola_kernel.
using cxf = float __attribute__((ext_vector_type(2)));

// Behind the two bases (same buffer): e^(+2 pi i k / n_fft), k < n_fft / 2, then e^(+2 pi i p / (n_fft / 2)), p < n_fft / 2 -- the
// factors of the even/odd split and of the two-stage transform, read through the (vector) L1 by every workgroup.
__global__ void istft_fft_tables_kernel(float* __restrict__ tab, int n_fft) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, M = n_fft / 2;
    if (i >= n_fft) return;
    float sn, cs;
    if (i < M) sincospif(2.0f * (float)i / (float)n_fft, &sn, &cs);
    else sincospif(2.0f * (float)(i - M) / (float)M, &sn, &cs);
    tab[2 * i] = cs;
    tab[2 * i + 1] = sn;
}

// LN lanes work on a frame (LN = 8: 192 = 8 x 24, LN = 16: 192 = 16 x 12) and a wave on 64 / LN frames, each lane touching
// 8-byte (spectrum) or 2 x 4-byte (overlap-add image) entries at l, l + LN, ...: a frame pitch of 2 LN banks (mod 64) puts
// the dwords of neighbouring frames' lanes side by side, so that a wave's access takes the four cycles its 512 bytes need
// anyway.  With the natural pitches (193 entries / 192 floats) lanes of neighbouring frames met in the same banks, up to
// eight deep.
template <int LN> constexpr int f3_pitch() { return LN == 8 ? 200 : 208; }       // 8-byte entries per frame: 2 LN dwords mod 64
template <int LN> constexpr int f3_ola_pitch() { return LN == 8 ? 208 : 224; }   // floats per block of 192 samples: 2 LN mod 64

template <int FR, int LN>
constexpr size_t fft384_lds_bytes() {
    return (size_t)(2 * FR * f3_pitch<LN>() + 2 * (FR - 1) * f3_ola_pitch<LN>() + 4 * 192 + 400 + 384 + 8) * sizeof(float);
}

template <int FR, int LN>
__global__ __launch_bounds__(FR * LN) void stft_ir_fft384_kernel(const float* __restrict__ noise_stft,
                                                                 const float* __restrict__ init_lm,
                                                                 const float* __restrict__ delta_lm,
                                                                 const float* __restrict__ gain_env,
                                                                 const float* __restrict__ window,
                                                                 const float* __restrict__ tables, float* __restrict__ ir,
                                                                 float* __restrict__ partial, IstftArgs a, int ms_to_lr) {
    constexpr int M = 192, C1 = M / LN, NT = FR * LN, PITCH = f3_pitch<LN>(), OP = f3_ola_pitch<LN>(), NB = FR - 1, K = 193;
    constexpr int QG = (K + LN - 1) / LN;    // bins a lane generates
    constexpr int QN = (C1 + LN - 1) / LN;   // second-stage transforms a lane may own (b = l + LN q < C1)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    cxf* Xs = reinterpret_cast<cxf*>(smem);                 // [FR][PITCH]: the masked spectra, then the stage-1 results
    float* ola = smem + 2 * FR * PITCH;                     // [2][NB][OP]
    cxf* tw384 = reinterpret_cast<cxf*>(ola + 2 * NB * OP); // e^(+2 pi i k / 384), k < 192
    cxf* tw192 = tw384 + M;                                 // e^(+2 pi i p / 192), p < 192
    float* slope_s = reinterpret_cast<float*>(tw192 + M);   // -softplus(delta) per bin (200 floats)
    float* h0_s = slope_s + 200;                            // the initial log-magnitudes (200 floats)
    float* win_s = h0_s + 200;                              // the window (384 floats)
    float* red = win_s + 384;
    const int tid = threadIdx.x;
    const int64_t r = blockIdx.y;
    const int mfirst = blockIdx.x * NB;                     // first frame; the blocks are mfirst + 1 .. mfirst + NB
    // (the tables could be read from the copy behind the bases directly -- 4.6 KB less LDS -- but per-lane 8-byte loads
    // through L1 cost more than the occupancy gives: cfg3 3.00 against 2.92 ms)
    for (int i = tid; i < 2 * M; i += NT) {
        if (i < M) {
            tw384[i] = reinterpret_cast<const cxf*>(tables)[i];
            tw192[i] = reinterpret_cast<const cxf*>(tables)[M + i];
        }
        win_s[i] = window[i];
    }
    const int f = tid / LN, l = tid % LN;                   // LN lanes per frame
    for (int c = 0; c < 2; ++c) {
        const int64_t rc = r * 2 + c;
        for (int i = tid; i < K; i += NT) {
            slope_s[i] = -softplus_t(delta_lm[rc * K + i]);
            h0_s[i] = init_lm[rc * K + i];
        }
        __syncthreads();   // (also: the previous channel's last reads of Xs are done)
        {
            const int m = mfirst + f;
            const bool m_ok = m < a.T;
            const float mf = (float)m;
            const float genv = (gain_env && m_ok) ? gain_env[rc * a.T + m] : 0.0f;
            const float* nz0 = noise_stft + r * a.nstride + ((int64_t)c * K * a.T + (m_ok ? m : 0)) * 2;
            // all noise loads of the lane in flight together (one at a time, each would cost a trip to L2: the workgroup
            // has only a few waves to hide it behind); lane l takes bins l, l + LN, ...
            float2 nz[QG];
#pragma unroll
            for (int q = 0; q < QG; ++q) {
                const int k = l + LN * q;
                nz[q] = (m_ok && k < K) ? *reinterpret_cast<const float2*>(nz0 + (int64_t)k * a.T * 2) : make_float2(0.0f, 0.0f);
            }
#pragma unroll
            for (int q = 0; q < QG; ++q) {
                const int k = l + LN * q;
                if (k < K) {
                    float lm = __fadd_rn(h0_s[k], __fmul_rn(slope_s[k], mf));
                    if (gain_env) lm = __fadd_rn(lm, genv);
                    const float mask = m_ok ? expf(lm / 8.0f) : 0.0f;
                    Xs[f * PITCH + k] = cxf{nz[q].x * mask, (k == 0 || k == M) ? 0.0f : nz[q].y * mask};
                }
            }
        }
        __syncthreads();
        cxf v[C1];
        {
            const cxf* X = Xs + f * PITCH;
#pragma unroll
            for (int j = 0; j < C1; ++j) {
                const int k = l + LN * j;
                const cxf A = X[k], Bc = X[M - k];                          // B = conj(Bc)
                const cxf e = cxf{A.x + Bc.x, A.y - Bc.y};                  // A + B
                const cxf d = cxf{A.x - Bc.x, A.y + Bc.y};                  // A - B
                const cxf w = tw384[k];
                const cxf o = cxf{d.x * w.x - d.y * w.y, d.x * w.y + d.y * w.x};
                v[j] = cxf{e.x - o.y, e.y + o.x} * (1.0f / 384.0f);         // (E + i O) / 192
            }
        }
        sdft<C1, true>(v);
        __syncthreads();   // every lane has read its spectrum
#pragma unroll
        for (int b = 0; b < C1; ++b) {
            const cxf y = v[spos(C1, b)], w = tw192[b * l];
            Xs[f * PITCH + b * LN + l] = cxf{y.x * w.x - y.y * w.y, y.x * w.y + y.y * w.x};
        }
        __syncthreads();
        cxf u[QN][LN];
#pragma unroll
        for (int q = 0; q < QN; ++q) {
            const int b = l + LN * q;
            if (b < C1) {
                const cxf* Y = Xs + f * PITCH + b * LN;
#pragma unroll
                for (int i = 0; i < LN; ++i) u[q][i] = Y[i];
                sdft<LN, true>(u[q]);
            }
        }
        float* o = ola + c * NB * OP;
        // first halves (a < LN / 2: n = 2 (C1 a + b) < 192) of frames 1 .. FR - 1 open their block
        if (f >= 1) {
#pragma unroll
            for (int q = 0; q < QN; ++q)
                if (l + LN * q < C1) {
#pragma unroll
                    for (int aa = 0; aa < LN / 2; ++aa) {
                        const int n = 2 * (C1 * aa + l + LN * q);
                        const cxf z = u[q][spos(LN, aa)];
                        *reinterpret_cast<float2*>(o + (f - 1) * OP + n) = make_float2(win_s[n] * z.x, win_s[n + 1] * z.y);
                    }
                }
        }
        __syncthreads();
        // second halves of frames 0 .. FR - 2 complete the block the next frame opened
        if (f <= FR - 2) {
#pragma unroll
            for (int q = 0; q < QN; ++q)
                if (l + LN * q < C1) {
#pragma unroll
                    for (int aa = LN / 2; aa < LN; ++aa) {
                        const int n = 2 * (C1 * aa + l + LN * q);
                        const cxf z = u[q][spos(LN, aa)];
                        float2* dst = reinterpret_cast<float2*>(o + f * OP + (n - M));
                        const float2 old = *dst;
                        // (frame m - 1 first, then frame m: the order of istft_ola_kernel's sum)
                        *dst = make_float2(win_s[n] * z.x + old.x, win_s[n + 1] * z.y + old.y);
                    }
                }
        }
    }
    __syncthreads();
    float e = 0.0f;
    for (int i = tid; i < NB * M; i += NT) {
        const int j = i / M, s = i - j * M;
        const int jb = mfirst + 1 + j;                      // block: frames jb - 1 (second half) and jb (first half)
        const int64_t tp = (int64_t)(jb - 1) * M + s;       // output index after the centre trim of n_fft / 2
        if (tp < a.ir_len && jb - 1 < a.T) {
            const float w1 = win_s[s + M], w0 = win_s[s];
            float env = fmaf(w1, w1, 0.0f);
            if (jb < a.T) env = fmaf(w0, w0, env);
            float v0 = ola[j * OP + s] / env, v1 = ola[NB * OP + j * OP + s] / env;
            if (ms_to_lr) {
                const float lft = v0 + v1, rgt = v0 - v1;
                v0 = lft;
                v1 = rgt;
            }
            ir[(r * 2 + 0) * a.ir_len + tp] = v0;
            ir[(r * 2 + 1) * a.ir_len + tp] = v1;
            e += fmaf(v1, v1, v0 * v0);
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) e += __shfl_down(e, d, 64);
    if ((tid & 63) == 0) red[tid >> 6] = e;
    __syncthreads();
    if (tid == 0) {
        float tot = red[0];
        for (int w = 1; w < NT / 64; ++w) tot += red[w];
        partial[r * gridDim.x + blockIdx.x] = tot;
    }
}

// ---- FilteredNoiseShapingReverb impulse response (reverb.py:343-366) -----------------------------------
//   ir[r,c,t] = sum_k noise[c,k,t] * gain[r,c,k] * (exp(t*d) - fg * exp(t*f))
//   d  = sigmoid(log_decay)*(max_decay - min_decay) + min_decay
//   f  = sigmoid(log_fade_in)*(d - min_decay) + min_decay,  fg = sigmoid(z_fade_in_gain)   (fade-in optional)
// and `gain` is the raw log_gain parameter, as upstream multiplies by it un-exponentiated (reverb.py:361).
// The reference materialises the (R,C,K,ir_len) envelope; here the K band terms are summed in registers.
constexpr int NS_MAX_K = 64;

__global__ __launch_bounds__(256) void noise_shaping_ir_kernel(const float* __restrict__ noise, int64_t noise_stride,
                                                               const float* __restrict__ log_decay,
                                                               const float* __restrict__ gain,
                                                               const float* __restrict__ log_fade_in,
                                                               const float* __restrict__ z_fade_gain,
                                                               float* __restrict__ ir, int C, int K, int64_t ir_len,
                                                               float min_decay, float max_decay) {
    __shared__ float sd[NS_MAX_K], sg[NS_MAX_K], sf[NS_MAX_K], sfg[NS_MAX_K];
    const int64_t rc = blockIdx.y;
    const int c = (int)(rc % C);
    if (threadIdx.x < K) {
        const int64_t i = rc * K + threadIdx.x;
        const float d = (max_decay - min_decay) / (1.0f + expf(-log_decay[i])) + min_decay;
        sd[threadIdx.x] = d;
        sg[threadIdx.x] = gain[i];
        if (log_fade_in) {
            sf[threadIdx.x] = (d - min_decay) / (1.0f + expf(-log_fade_in[i])) + min_decay;
            sfg[threadIdx.x] = 1.0f / (1.0f + expf(-z_fade_gain[i]));
        }
    }
    __syncthreads();
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ir_len) return;
    const float ft = (float)t;
    const float* nz = noise + (int64_t)c * K * noise_stride + t;
    float acc = 0.0f;
    for (int k = 0; k < K; ++k) {
        float env = expf(ft * sd[k]);
        if (log_fade_in) env -= expf(ft * sf[k]) * sfg[k];
        acc += nz[k * noise_stride] * (env * sg[k]);
    }
    ir[rc * ir_len + t] = acc;
}

}  // namespace gfx

using namespace gfx;


namespace gfx {
// ---- forward STFT of real rows (fixed_noise=False: fresh noise per call, reverb.py:116-128) ---------------------------
// torch.stft(x, n_fft, hop, window, center=True, pad_mode="reflect", return_complex=True) as a direct sum: one workgroup
// per (row, frame) holds the windowed frame and the n_fft roots of unity in LDS, thread k sums bin k (phase index j k mod
// n_fft carried incrementally: exact).  n_fft <= 2048 (384 in the reverb: 313 frames x 193 bins x 384 terms per row of
// 60 000 samples -- microseconds; the FFT library this replaces was the last one on the reverb's forward path).
__global__ __launch_bounds__(256) void stft_frames_kernel(const float* __restrict__ x, const float* __restrict__ window,
                                                          float2* __restrict__ out, int64_t T, int n_fft, int hop,
                                                          int frames) {
    extern __shared__ float2 stft_lds[];                 // tab[n_fft], then the windowed frame
    float2* tab = stft_lds;
    float* fr = reinterpret_cast<float*>(tab + n_fft);
    const int64_t row = blockIdx.y;
    const int m = blockIdx.x;
    const int bins = n_fft / 2 + 1, half = n_fft / 2;
    const float* xr = x + row * T;
    for (int j = threadIdx.x; j < n_fft; j += blockDim.x) {
        double s, c;
        sincospi(2.0 * (double)j / (double)n_fft, &s, &c);
        tab[j] = make_float2((float)c, (float)s);
        int64_t n = (int64_t)m * hop - half + j;        // reflect padding (no edge repeat), as torch / numpy "reflect"
        if (n < 0) n = -n;
        if (n >= T) n = 2 * (T - 1) - n;
        fr[j] = window[j] * xr[n];
    }
    __syncthreads();
    for (int k = threadIdx.x; k < bins; k += blockDim.x) {
        float re = 0.0f, im = 0.0f;
        int idx = 0;
        for (int j = 0; j < n_fft; ++j) {
            const float2 w = tab[idx];
            re = fmaf(fr[j], w.x, re);
            im = fmaf(-fr[j], w.y, im);
            idx += k;
            if (idx >= n_fft) idx -= n_fft;
        }
        out[((int64_t)row * bins + k) * frames + m] = make_float2(re, im);
    }
}
}  // namespace gfx

extern "C" {

int gfx_stft_f32(const float* x, const float* window, float* out, int64_t rows, int64_t T, int64_t n_fft, int64_t hop,
                 void* stream) {
    if (!x || !window || !out || rows <= 0 || rows > 65535 || n_fft < 2 || n_fft > 2048 || (n_fft & 1) || hop < 1 ||
        T <= n_fft / 2)      // (reflect padding needs more than n_fft / 2 samples, as torch.stft does)
        return GFX_EINVAL;
    const int64_t frames = 1 + T / hop;
    if (frames > 0x7fffffffLL) return GFX_EINVAL;
    const size_t lds = (size_t)n_fft * (sizeof(float2) + sizeof(float));
    hipLaunchKernelGGL(gfx::stft_frames_kernel, dim3((unsigned)frames, (unsigned)rows), dim3(256), lds, (hipStream_t)stream, x,
                       window, (float2*)out, T, (int)n_fft, (int)hop, (int)frames);
    return hipGetLastError() == hipSuccess ? GFX_OK : GFX_ELAUNCH;
}

int gfx_noise_shaping_ir_f32(const float* noise, int64_t noise_stride, const float* log_decay, const float* log_gain,
                             const float* log_fade_in, const float* z_fade_in_gain, float* ir, int64_t R, int64_t C,
                             int64_t K, int64_t ir_len, float min_decay, float max_decay, void* stream) {
    if (!noise || !log_decay || !log_gain || !ir || R <= 0 || C <= 0 || K <= 0 || K > NS_MAX_K || ir_len <= 0)
        return GFX_EINVAL;
    if ((log_fade_in == nullptr) != (z_fade_in_gain == nullptr) || noise_stride < ir_len || R * C > 0x7fffffffLL)
        return GFX_EINVAL;
    const int64_t rows = R * C;
    for (int64_t done = 0; done < rows; done += 65535) {  // grid.y limit
        const int64_t n = rows - done < 65535 ? rows - done : 65535;
        if (done % C != 0) return GFX_EINVAL;  // unreachable for C in {1,2,3,5,...}; keeps the channel phase
        hipLaunchKernelGGL(noise_shaping_ir_kernel, dim3((unsigned)((ir_len + 255) / 256), (unsigned)n), dim3(256), 0,
                           (hipStream_t)stream, noise, noise_stride, log_decay + done * K, log_gain + done * K,
                           log_fade_in ? log_fade_in + done * K : nullptr,
                           z_fade_in_gain ? z_fade_in_gain + done * K : nullptr, ir + done * ir_len, (int)C, (int)K,
                           ir_len, min_decay, max_decay);
    }
    return hipGetLastError() == hipSuccess ? GFX_OK : GFX_ELAUNCH;
}

size_t gfx_istft_basis_bytes(int64_t n_fft) {
    if (n_fft < 2 || (n_fft & 1)) return 0;
    return ((size_t)kpad_of(n_fft) * n_fft + (size_t)(n_fft / 2 + 1) * 2 * half_cols(n_fft) + 2 * (size_t)n_fft) * sizeof(float);
}

int gfx_istft_basis_f32(const float* window, float* basis, int64_t n_fft, void* stream) {
    if (!window || !basis || n_fft < 2 || (n_fft & 1) || n_fft > 65536) return GFX_EINVAL;
    const int64_t total = kpad_of(n_fft) * n_fft;
    hipLaunchKernelGGL(istft_basis_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       window, basis, (int)n_fft, (int)kpad_of(n_fft));
    const int64_t htotal = (n_fft / 2 + 1) * 2 * half_cols(n_fft);
    hipLaunchKernelGGL(istft_half_basis_kernel, dim3((unsigned)((htotal + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       basis + total, (int)n_fft, (int)half_cols(n_fft));
    hipLaunchKernelGGL(istft_fft_tables_kernel, dim3((unsigned)((n_fft + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       basis + total + htotal, (int)n_fft);
    return hipGetLastError() == hipSuccess ? GFX_OK : GFX_ELAUNCH;
}

size_t gfx_stft_reverb_workspace_bytes(int64_t R, int64_t n_fft, int64_t num_frames) {
    if (R <= 0 || n_fft <= 0 || num_frames <= 0) return 0;
    // the frames, then one energy partial per 256 impulse-response samples and row (ir_len < num_frames * hop <= num_frames * n_fft)
    return ((size_t)R * 2 * num_frames * n_fft + (size_t)R * ((num_frames * n_fft + 255) / 256)) * sizeof(float);
}

size_t gfx_stft_reverb_workspace_bytes_sched(int64_t R, int64_t ir_len, int64_t n_fft, int64_t hop, int64_t num_frames,
                                             int schedule) {
    if (R <= 0 || ir_len <= 0 || n_fft <= 0 || num_frames <= 0) return 0;
    if (n_fft == 384 && hop == 192 && schedule != GFX_ISTFT_GEMM)   // the FFT form: one energy partial per workgroup
        return (size_t)R * (size_t)(((ir_len + 191) / 192 + 14) / 15) * sizeof(float);
    return gfx_stft_reverb_workspace_bytes(R, n_fft, num_frames);
}

int gfx_stft_reverb_ir_f32(const float* noise_stft, const float* init_log_magnitude, const float* delta_log_magnitude,
                           const float* gain_env_log_magnitude, const float* window, const float* basis, float* ir,
                           float* row_gain, int64_t R, int64_t ir_len, int64_t n_fft, int64_t hop, int64_t num_frames,
                           int ms_to_lr, void* ws, size_t ws_bytes, void* stream) {
    return gfx_stft_reverb_ir_ex_f32(noise_stft, 1, init_log_magnitude, delta_log_magnitude, gain_env_log_magnitude, window,
                                     basis, ir, row_gain, R, ir_len, n_fft, hop, num_frames, ms_to_lr, ws, ws_bytes, stream);
}

int gfx_stft_reverb_ir_ex_f32(const float* noise_stft, int64_t noise_rows, const float* init_log_magnitude,
                              const float* delta_log_magnitude, const float* gain_env_log_magnitude, const float* window,
                              const float* basis, float* ir, float* row_gain, int64_t R, int64_t ir_len, int64_t n_fft,
                              int64_t hop, int64_t num_frames, int ms_to_lr, void* ws, size_t ws_bytes, void* stream) {
    return gfx_stft_reverb_ir_sched_f32(noise_stft, noise_rows, init_log_magnitude, delta_log_magnitude,
                                        gain_env_log_magnitude, window, basis, ir, row_gain, R, ir_len, n_fft, hop,
                                        num_frames, ms_to_lr, ws, ws_bytes, GFX_ISTFT_AUTO, stream);
}

int gfx_stft_reverb_ir_sched_f32(const float* noise_stft, int64_t noise_rows, const float* init_log_magnitude,
                                 const float* delta_log_magnitude, const float* gain_env_log_magnitude,
                                 const float* window, const float* basis, float* ir, float* row_gain, int64_t R,
                                 int64_t ir_len, int64_t n_fft, int64_t hop, int64_t num_frames, int ms_to_lr, void* ws,
                                 size_t ws_bytes, int schedule, void* stream) {
    if (noise_rows != 1 && noise_rows != R) return GFX_EINVAL;
    if (!noise_stft || !init_log_magnitude || !delta_log_magnitude || !window || !basis || !ir || !row_gain)
        return GFX_EINVAL;
    if (R <= 0 || ir_len <= 0 || n_fft < 2 || (n_fft & 1) || hop < 1 || hop > n_fft || num_frames < 1) return GFX_EINVAL;
    if (R * 2 > 65535) return GFX_EINVAL;
    const size_t need = gfx_stft_reverb_workspace_bytes_sched(R, ir_len, n_fft, hop, num_frames, schedule);
    if (!ws || ws_bytes < need) return GFX_ENOSPC;
    IstftArgs a;
    a.R = R;
    a.n_fft = (int)n_fft;
    a.hop = (int)hop;
    a.T = (int)num_frames;
    a.K = (int)(n_fft / 2 + 1);
    a.kpad = (int)kpad_of(n_fft);
    a.ir_len = ir_len;
    a.nstride = noise_rows == 1 ? 0 : 2 * (n_fft / 2 + 1) * num_frames * 2;
    hipStream_t st = (hipStream_t)stream;
    if (schedule != GFX_ISTFT_AUTO && schedule != GFX_ISTFT_GEMM && schedule != GFX_ISTFT_FFT) return GFX_EINVAL;
    const bool fft_ok = n_fft == 384 && hop == 192;
    if (schedule == GFX_ISTFT_FFT && !fft_ok) return GFX_EINVAL;
    if (fft_ok && schedule != GFX_ISTFT_GEMM) {
        // frames by FFT, overlap-add in LDS: one workgroup = one row, 31 blocks of 192 samples
        if (ir_len > num_frames * n_fft || R > 65535) return GFX_EINVAL;
        const int64_t nblocks = (ir_len + 191) / 192;
        float* partial = (float*)ws;
        auto launch = [&](auto FRc, auto LNc) -> int {
            constexpr int FR = decltype(FRc)::value, LN = decltype(LNc)::value;
            constexpr size_t lds = fft384_lds_bytes<FR, LN>();
            static_assert(lds <= 160 * 1024, "LDS");
            const unsigned nwg = (unsigned)((nblocks + FR - 2) / (FR - 1));
            auto kern = stft_ir_fft384_kernel<FR, LN>;
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds) != hipSuccess)
                return GFX_ELAUNCH;
            hipLaunchKernelGGL(kern, dim3(nwg, (unsigned)R), dim3(FR * LN), lds, st, noise_stft, init_log_magnitude,
                               delta_log_magnitude, gain_env_log_magnitude, window,
                               basis + kpad_of(n_fft) * n_fft + (n_fft / 2 + 1) * 2 * half_cols(n_fft), ir, partial, a,
                               ms_to_lr);
            hipLaunchKernelGGL(ir_energy_gain_kernel, dim3((unsigned)R), dim3(64), 0, st, (const float*)partial, row_gain,
                               (int)nwg);
            return GFX_OK;
        };
        static const int fr_env = [] { const char* e = getenv("GRAFX_ISTFT_FR"); return e ? atoi(e) : 0; }();
        using I8 = std::integral_constant<int, 8>;
        using I16 = std::integral_constant<int, 16>;
        using I32 = std::integral_constant<int, 32>;
        const int rc = fr_env == 32 ? launch(I32{}, I8{}) : fr_env == 16 ? launch(I16{}, I8{}) : launch(I16{}, I16{});
        if (rc != GFX_OK) return rc;
        return hipGetLastError() == hipSuccess ? GFX_OK : GFX_ELAUNCH;
    }
    if (n_fft <= 384) {
        dim3 g1((unsigned)((num_frames + 63) / 64), (unsigned)(R * 2));
#ifdef GFX_ISTFT_FULL   // round 1's full-matrix form, kept for A/B timing
        hipLaunchKernelGGL(istft_frames_tiled_kernel, g1, dim3(256), 0, st, noise_stft, init_log_magnitude,
                           delta_log_magnitude, gain_env_log_magnitude, basis, (float*)ws, a);
#else
        hipLaunchKernelGGL(istft_frames_half_kernel, g1, dim3(256), 0, st, noise_stft, init_log_magnitude,
                           delta_log_magnitude, gain_env_log_magnitude, basis + kpad_of(n_fft) * n_fft, window, (float*)ws, a);
#endif
    } else {
        dim3 g1((unsigned)((n_fft + 127) / 128), (unsigned)((num_frames + 63) / 64), (unsigned)(R * 2));
        hipLaunchKernelGGL(istft_frames_kernel, g1, dim3(256), 0, st, noise_stft, init_log_magnitude,
                           delta_log_magnitude, gain_env_log_magnitude, basis, (float*)ws, a);
    }
    if (ir_len > num_frames * n_fft) return GFX_EINVAL;   // (the partials' share of the workspace is sized by this bound)
    dim3 g2((unsigned)((ir_len + 255) / 256), (unsigned)R);
    float* partial = (float*)ws + (size_t)R * 2 * num_frames * n_fft;
    hipLaunchKernelGGL(istft_ola_kernel, g2, dim3(256), 0, st, (const float*)ws, window, ir, partial, a, ms_to_lr);
    hipLaunchKernelGGL(ir_energy_gain_kernel, dim3((unsigned)R), dim3(64), 0, st, (const float*)partial, row_gain, (int)g2.x);
    return hipGetLastError() == hipSuccess ? GFX_OK : GFX_ELAUNCH;
}

}  // extern "C"
