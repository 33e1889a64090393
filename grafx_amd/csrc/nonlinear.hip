// Memoryless waveshapers (gfx950): one streaming pass, 8 B per channel-sample, HBM-bound.
//
// Replaces the forward() bodies of grafx.processors.nonlinear (reference nonlinear.py):
//   TanhDistortion           46-79    y = post * (tanh(pre*(x-dc) + b) - tanh(b))
//   PiecewiseTanhDistortion  120-175  tanh in the middle, rescaled tanh branches beyond +kp / -kn
//   PowerDistortion          210-233  y = sum_k tanh(w_k) * f((pre*(x-dc))^k)
//   ChebyshevDistortion      270-307  y = sum_k tanh(w_k) * f(T_k(pre*(x-dc)))
// where f = tanh or identity, dc = the row-channel mean when remove_dc is set (row_mean_kernel),
// pre = exp(log_pre_gain), post = exp(log_post_gain) or 1/pre.  The upstream torch code materialises
// K full-size tensors for the two polynomial shapers; here the K terms live in registers.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/grafx_amd.h"

namespace gfx {

constexpr int WS_MAX_K = 32;

__device__ __forceinline__ int64_t nrow_off(const gfx_rowmap_t& m, int64_t r, int c) {
    const unsigned inner = (unsigned)m.inner, rr = (unsigned)r;
    const unsigned q = rr / inner, rem = rr - q * inner;
    return (int64_t)q * m.stride_outer + (int64_t)rem * m.stride_inner + (int64_t)c * m.stride_ch;
}

struct WsArgs {
    gfx_rowmap_t xmap, ymap;
    int64_t R, L;
    int C, mode, K;
    int use_tanh, inverse_post;
};

struct WsRow {
    float pre, post, dc;
    float b, tb;                    // tanh: bias, tanh(bias)
    float kp, kn, gp, gn, ap, an, bp, bn;  // piecewise
    const float* w;                 // polynomial weights (already tanh'ed), in LDS: a run-time-indexed register
};                                  // array would live in scratch memory

template <int MODE>
__device__ __forceinline__ float shape(float x, const WsRow& q, int K, bool use_tanh) {
    const float u = (x - q.dc) * q.pre;
    float y;
    if (MODE == GFX_WS_TANH) {
        y = tanhf(u + q.b) - q.tb;
    } else if (MODE == GFX_WS_PIECEWISE) {
        if (u > q.kp)
            y = q.ap * tanhf(q.gp * (u - q.kp)) + q.bp;
        else if (u < -q.kn)
            y = q.an * tanhf(q.gn * (u + q.kn)) + q.bn;
        else
            y = tanhf(u);
    } else if (MODE == GFX_WS_POWER) {
        float p = 1.0f;
        y = q.w[0] * (use_tanh ? tanhf(1.0f) : 1.0f);
        for (int k = 1; k < K; ++k) {
            p *= u;
            y += q.w[k] * (use_tanh ? tanhf(p) : p);
        }
    } else {  // Chebyshev
        float t0 = 1.0f, t1 = u;
        y = q.w[0] * (use_tanh ? tanhf(1.0f) : 1.0f);
        if (K > 1) y += q.w[1] * (use_tanh ? tanhf(u) : u);
        for (int k = 2; k < K; ++k) {
            const float t2 = 2.0f * u * t1 - t0;
            y += q.w[k] * (use_tanh ? tanhf(t2) : t2);
            t0 = t1;
            t1 = t2;
        }
    }
    return y * q.post;
}

template <int MODE>
__global__ __launch_bounds__(256) void waveshaper_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         const float* __restrict__ log_pre,
                                                         const float* __restrict__ log_post,
                                                         const float* __restrict__ p0, const float* __restrict__ p1,
                                                         const float* __restrict__ dc, WsArgs a, int vec) {
    __shared__ float sw[WS_MAX_K];
    for (int64_t r = blockIdx.y; r < a.R; r += gridDim.y) {
        WsRow q;
        q.w = sw;
        q.pre = log_pre ? expf(log_pre[r]) : 1.0f;
        q.post = a.inverse_post ? 1.0f / q.pre : (log_post ? expf(log_post[r]) : 1.0f);
        q.b = q.tb = 0.0f;
        if (MODE == GFX_WS_TANH && p0) {
            q.b = p0[r];
            q.tb = tanhf(q.b);
        }
        if (MODE == GFX_WS_PIECEWISE) {
            // nonlinear.py:163-166: threshold splits as (kn, kp), hardness as (gp, gn)
            q.gp = expf(p0[2 * r]);
            q.gn = expf(p0[2 * r + 1]);
            q.kn = 1.0f / (1.0f + expf(-p1[2 * r]));
            q.kp = 1.0f / (1.0f + expf(-p1[2 * r + 1]));
            q.bp = tanhf(q.kp);
            q.bn = -tanhf(q.kn);
            q.ap = (1.0f - q.bp) / q.gp;
            q.an = (1.0f + q.bn) / q.gn;
        }
        if (MODE == GFX_WS_POWER || MODE == GFX_WS_CHEBYSHEV) {
            __syncthreads();  // previous row's readers are done
            if ((int)threadIdx.x < a.K) sw[threadIdx.x] = tanhf(p0[r * a.K + threadIdx.x]);
            __syncthreads();
        }
        for (int c = 0; c < a.C; ++c) {
            q.dc = dc ? dc[r * a.C + c] : 0.0f;
            const float* xr = x + nrow_off(a.xmap, r, c);
            float* yr = y + nrow_off(a.ymap, r, c);
            const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (int64_t)gridDim.x * blockDim.x;
            if (vec) {
                using f4 = float __attribute__((ext_vector_type(4)));
                for (int64_t i = tid; i < a.L / 4; i += nthr) {
                    const f4 v = reinterpret_cast<const f4*>(xr)[i];
                    f4 o;
                    o.x = shape<MODE>(v.x, q, a.K, a.use_tanh);
                    o.y = shape<MODE>(v.y, q, a.K, a.use_tanh);
                    o.z = shape<MODE>(v.z, q, a.K, a.use_tanh);
                    o.w = shape<MODE>(v.w, q, a.K, a.use_tanh);
                    __builtin_nontemporal_store(o, reinterpret_cast<f4*>(yr) + i);
                }
                for (int64_t n = (a.L & ~int64_t(3)) + tid; n < a.L; n += nthr) yr[n] = shape<MODE>(xr[n], q, a.K, a.use_tanh);
            } else {
                for (int64_t n = tid; n < a.L; n += nthr) yr[n] = shape<MODE>(xr[n], q, a.K, a.use_tanh);
            }
        }
    }
}

// mean over time of every row-channel (the remove_dc option): one workgroup per row-channel
__global__ __launch_bounds__(256) void row_mean_kernel(const float* __restrict__ x, gfx_rowmap_t xmap,
                                                       float* __restrict__ mean, int64_t R, int C, int64_t L) {
    __shared__ float part[4];
    for (int64_t rc = blockIdx.x; rc < R * C; rc += gridDim.x) {
        const int64_t r = rc / C;
        const float* xr = x + nrow_off(xmap, r, (int)(rc - r * C));
        float s = 0.0f;
        for (int64_t n = threadIdx.x; n < L; n += 256) s += xr[n];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) mean[rc] = (part[0] + part[1] + part[2] + part[3]) / (float)L;
        __syncthreads();
    }
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline bool map_vec(const gfx_rowmap_t& m) {
    return m.stride_outer % 4 == 0 && m.stride_inner % 4 == 0 && m.stride_ch % 4 == 0;
}

}  // namespace gfx

using namespace gfx;

extern "C" {

int gfx_row_mean_f32(const float* x, gfx_rowmap_t xmap, float* mean, int64_t R, int64_t C, int64_t L, void* stream) {
    if (!x || !mean || R <= 0 || C <= 0 || L <= 0 || xmap.inner <= 0 || R > 0x7fffffffLL) return GFX_EINVAL;
    const int64_t n = R * C;
    hipLaunchKernelGGL(row_mean_kernel, dim3((unsigned)(n > 65535 * 16 ? 65535 * 16 : n)), dim3(256), 0,
                       (hipStream_t)stream, x, xmap, mean, R, (int)C, L);
    return hipGetLastError() == hipSuccess ? GFX_OK : GFX_ELAUNCH;
}

int gfx_waveshaper_f32(const float* x, gfx_rowmap_t xmap, float* y, gfx_rowmap_t ymap, int64_t R, int64_t C, int64_t L,
                       int mode, int use_tanh, int inverse_post_gain, const float* log_pre_gain,
                       const float* log_post_gain, const float* p0, const float* p1, int64_t K, const float* dc,
                       void* stream) {
    if (!x || !y || R <= 0 || C <= 0 || L <= 0 || R > 0x7fffffffLL || xmap.inner <= 0 || ymap.inner <= 0) return GFX_EINVAL;
    if (mode < GFX_WS_TANH || mode > GFX_WS_CHEBYSHEV) return GFX_EINVAL;
    if (inverse_post_gain && !log_pre_gain) return GFX_EINVAL;
    if (mode == GFX_WS_PIECEWISE && (!p0 || !p1)) return GFX_EINVAL;
    if ((mode == GFX_WS_POWER || mode == GFX_WS_CHEBYSHEV) && (!p0 || K < 1 || K > WS_MAX_K)) return GFX_EINVAL;
    WsArgs a;
    a.xmap = xmap; a.ymap = ymap; a.R = R; a.L = L; a.C = (int)C; a.mode = mode; a.K = (int)K;
    a.use_tanh = use_tanh; a.inverse_post = inverse_post_gain;
    const int vec = aligned16(x) && aligned16(y) && map_vec(xmap) && map_vec(ymap);
    int64_t bx = (L / 4 + 255) / 256;
    if (bx > 64) bx = 64;
    if (bx < 1) bx = 1;
    const dim3 grid((unsigned)bx, (unsigned)(R > 65535 ? 65535 : R));
    hipStream_t st = (hipStream_t)stream;
    switch (mode) {
        case GFX_WS_TANH:
            hipLaunchKernelGGL(waveshaper_kernel<GFX_WS_TANH>, grid, dim3(256), 0, st, x, y, log_pre_gain, log_post_gain, p0, p1, dc, a, vec);
            break;
        case GFX_WS_PIECEWISE:
            hipLaunchKernelGGL(waveshaper_kernel<GFX_WS_PIECEWISE>, grid, dim3(256), 0, st, x, y, log_pre_gain, log_post_gain, p0, p1, dc, a, vec);
            break;
        case GFX_WS_POWER:
            hipLaunchKernelGGL(waveshaper_kernel<GFX_WS_POWER>, grid, dim3(256), 0, st, x, y, log_pre_gain, log_post_gain, p0, p1, dc, a, vec);
            break;
        default:
            hipLaunchKernelGGL(waveshaper_kernel<GFX_WS_CHEBYSHEV>, grid, dim3(256), 0, st, x, y, log_pre_gain, log_post_gain, p0, p1, dc, a, vec);
            break;
    }
    return hipGetLastError() == hipSuccess ? GFX_OK : GFX_ELAUNCH;
}

}  // extern "C"

// ---- small inverse real DFT (parameter-side front-ends) --------------------------------------------------------------
// y = irfft(X, n) for short transforms of ANY length n <= 8192 as a direct sum: the zero-phase FIR design
// (core/fir.py:20-27: n = 2 bins - 1 = 2047, odd) and the surrogate delay's soft impulse (core/delay.py:73-76).  These
// are per-node front-ends (R x ~1024 bins), where a direct O(n K) sum with the twiddles e^{2 pi i j / n} tabulated
// in LDS (phase index k m mod n carried incrementally: exact) is a few microseconds per row and replaces the FFT library.
//   y[(m + roll) mod n] = w[(m + roll) mod n] / n * ( Re X[0] + 2 sum_{0<k<n/2} Re(X[k] e^{2 pi i k m / n})
//                                                      + [n even] Re X[n/2] (-1)^m )
namespace gfx {

__global__ __launch_bounds__(256) void irdft_kernel(const float* __restrict__ X, int is_real, float* __restrict__ y,
                                                    int K, int n, int roll, const float* __restrict__ window) {
    extern __shared__ float2 tab[];                       // tab[j] = e^{2 pi i j / n}, then the row's spectrum
    float2* spec = tab + n;
    const int64_t row = blockIdx.x;
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        double s, c;                                       // in double: j / n is not exact in float (6e-7 rad at n = 4001)
        sincospi(2.0 * (double)j / (double)n, &s, &c);
        tab[j] = make_float2((float)c, (float)s);
    }
    for (int k = threadIdx.x; k < K; k += blockDim.x)
        spec[k] = is_real ? make_float2(X[row * K + k], 0.0f)
                          : make_float2(X[(row * K + k) * 2], X[(row * K + k) * 2 + 1]);
    __syncthreads();
    const int half = n / 2;
    const bool even = (n & 1) == 0;
    const int kmax = even ? half - 1 : half;              // bins with weight 2
    for (int m = threadIdx.x; m < n; m += blockDim.x) {
        double acc = 0.0;                                  // thousands of terms: accumulate in double (a tiny kernel)
        int idx = 0;
        for (int k = 1; k <= kmax; ++k) {
            idx += m;
            if (idx >= n) idx -= n;
            const float2 w = tab[idx], x = spec[k];
            acc += (double)x.x * (double)w.x - (double)x.y * (double)w.y;
        }
        float v = (float)((double)spec[0].x + 2.0 * acc);
        if (even) v += (m & 1) ? -spec[half].x : spec[half].x;
        int o = m + roll;
        o -= (o >= n) ? n : 0;
        v /= (float)n;
        y[row * n + o] = window ? v * window[o] : v;
    }
}

// X[k] = sum_m x[m] e^{-2 pi i k m / n}, k = 0 .. n/2: the forward twin (rfft) of irdft_kernel, same table, same exact
// phase index, double accumulation.  The gradient of an inverse real DFT is this transform of the incoming gradient
// (autograd.IrdftFn, autograd.FsmFirFn.backward): with it the parameter-side front-ends of the training path stay off the
// FFT library for every length the inverse kernel covers.
// One workgroup per (row, 32 bins): 8 lanes share a bin, each sums every 8th sample, then a shuffle tree -- the rows are
// few (one per filter), so the work of a row has to spread over many workgroups (a first cut with one workgroup per row
// and the whole sum over m in one thread took 0.65 ms for 32 filters of 4001 taps, all of it latency).
__global__ __launch_bounds__(256) void rdft_kernel(const float* __restrict__ x, float* __restrict__ X, int K, int n) {
    extern __shared__ float2 tab[];                       // tab[j] = e^{2 pi i j / n}, then the row's samples
    float* sig = reinterpret_cast<float*>(tab + n);
    const int64_t row = blockIdx.y;
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        double s, c;
        sincospi(2.0 * (double)j / (double)n, &s, &c);
        tab[j] = make_float2((float)c, (float)s);
        sig[j] = x[row * n + j];
    }
    __syncthreads();
    const int part = threadIdx.x & 7;
    const int k = blockIdx.x * 32 + (threadIdx.x >> 3);
    double re = 0.0, im = 0.0;
    if (k < K) {
        int idx = (int)(((int64_t)k * part) % n);          // k m mod n for m = part, part + 8, ...
        const int step = (int)(((int64_t)k * 8) % n);
        for (int m = part; m < n; m += 8) {
            const float2 w = tab[idx];
            re += (double)sig[m] * (double)w.x;
            im -= (double)sig[m] * (double)w.y;
            idx += step;
            if (idx >= n) idx -= n;
        }
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) {                       // the 8 lanes of a bin are adjacent
        re += __shfl_down(re, o, 8);
        im += __shfl_down(im, o, 8);
    }
    if (k < K && part == 0) {
        X[(row * K + k) * 2] = (float)re;
        X[(row * K + k) * 2 + 1] = (float)im;
    }
}

}  // namespace gfx

extern "C" int gfx_rdft_f32(const float* x, float* X, int64_t rows, int64_t K, int64_t n, void* stream) {
    if (!x || !X || rows <= 0 || rows > 65535 || n < 1 || n > 8192 || K != n / 2 + 1) return GFX_EINVAL;
    const size_t lds = (size_t)n * (sizeof(float2) + sizeof(float));
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(gfx::rdft_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
        return GFX_ELAUNCH;
    hipLaunchKernelGGL(gfx::rdft_kernel, dim3((unsigned)((K + 31) / 32), (unsigned)rows), dim3(256), lds, (hipStream_t)stream, x, X,
                       (int)K, (int)n);
    return hipGetLastError() == hipSuccess ? GFX_OK : GFX_ELAUNCH;
}

extern "C" int gfx_irdft_f32(const float* X, int is_real, float* y, int64_t rows, int64_t K, int64_t n, int64_t roll,
                             const float* window, void* stream) {
    if (!X || !y || rows <= 0 || rows > 0x7fffffffLL || n < 1 || n > 8192 || K != n / 2 + 1 || roll < 0 || roll >= n)
        return GFX_EINVAL;
    const size_t lds = (size_t)(n + K) * sizeof(float2);
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(gfx::irdft_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
        return GFX_ELAUNCH;
    hipLaunchKernelGGL(gfx::irdft_kernel, dim3((unsigned)rows), dim3(256), lds, (hipStream_t)stream, X, is_real, y, (int)K,
                       (int)n, (int)roll, window);
    return hipGetLastError() == hipSuccess ? GFX_OK : GFX_ELAUNCH;
}
