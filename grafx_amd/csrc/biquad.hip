// Exact time-domain biquad cascade as a parallel scan (gfx950).
//
// Replaces IIRFilter._process_lfilter / _process_ssm — reference core/iir.py:154-261 — whose upstream
// implementations call torchaudio.functional.lfilter / torchlpc.sample_wise_lpc (neither is installed here):
// K second-order sections applied in series to every row-channel,
//     w[n] = x[n] - a1 w[n-1] - a2 w[n-2],   y[n] = b0 w[n] + b1 w[n-1] + b2 w[n-2]      (a0-normalised)
// with zero initial state.  No FFT, no truncation of the impulse response (the FSM backend aliases it to
// fsm_fir_len taps), 8 B of HBM traffic per channel-sample.
//
// Parallelisation: one WAVE walks a row-channel in 512-sample tiles (64 lanes x 8 samples), four row-channels
// per workgroup, no barriers in the time loop.  The recursion is the linear system s[n] = M s[n-1] + (x[n], 0),
// M = [[-a1, -a2], [1, 0]], s = (w[n], w[n-1]):
//   1. each lane runs its 8 samples from a zero state                 -> end state e_t
//   2. Hillis-Steele scan over the 64 lanes with M^(8*2^d)            -> state at the end of every chunk
//   3. every lane adds M^(8*lane) * (carry entering the tile), reruns its 8 samples from the true state and
//      applies the numerator; the tile's end state (lane 63, plus M^512 * carry) is the next carry.
// Matrix powers are formed in double per (row-channel, section) when the wave starts and live in LDS.
//
// ssm_quirk: upstream's "ssm" backend drives the recursive part of every section with the ORIGINAL input
// instead of the previous section's output (core/iir.py:226-246 index `input_signal`, not `x`); for K = 1
// the two backends agree, for K > 1 this flag reproduces what "ssm" actually returns.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/grafx_amd.h"

namespace gfx {

constexpr int BQ_T = 256;            // threads per workgroup = 4 independent waves
constexpr int BQ_W = BQ_T / 64;      // row-channels per workgroup
constexpr int BQ_E = 8;              // samples per lane
constexpr int BQ_TILE = 64 * BQ_E;   // samples per wave tile
constexpr int BQ_MAX_K = 32;  // 4 waves x K x (160 B constants + 1 KB lane powers) of LDS: 148 KB at K = 32

struct M2 {  // 2x2 matrix, row-major
    float a, b, c, d;
};
struct M2d {
    double a, b, c, d;
};
__device__ __forceinline__ M2d mul(const M2d& x, const M2d& y) {
    return {x.a * y.a + x.b * y.c, x.a * y.b + x.b * y.d, x.c * y.a + x.d * y.c, x.c * y.b + x.d * y.d};
}
__device__ __forceinline__ M2 narrow(const M2d& m) { return {(float)m.a, (float)m.b, (float)m.c, (float)m.d}; }
__device__ __forceinline__ float2 apply(const M2& m, float2 s) {
    return make_float2(fmaf(m.a, s.x, m.b * s.y), fmaf(m.c, s.x, m.d * s.y));
}

struct SecConst {       // per (wave, section), in LDS
    M2 step[6];         // M^(E * 2^d)
    M2 wave;            // M^(E * 64)
    float b0, b1, b2, a1, a2, pad0, pad1, pad2;
    float2 carry;       // state entering the current tile
    float2 pad3;
};
// LDS: SecConst sec[4][K]; M2 lanepow[4][K][64]

__device__ __forceinline__ int64_t brow_off(const gfx_rowmap_t& m, int64_t r, int c) {
    const unsigned inner = (unsigned)m.inner, rr = (unsigned)r;
    const unsigned q = rr / inner, rem = rr - q * inner;
    return (int64_t)q * m.stride_outer + (int64_t)rem * m.stride_inner + (int64_t)c * m.stride_ch;
}

struct BqArgs {
    gfx_rowmap_t xmap, ymap;
    int64_t L, total;
    int Cin, Cf, Cout, K, quirk, vec;
};

__global__ __launch_bounds__(BQ_T) void biquad_cascade_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                              const float* __restrict__ Bs,
                                                              const float* __restrict__ As, BqArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    SecConst* sec = reinterpret_cast<SecConst*>(smem) + wave * a.K;
    M2* lanepow = reinterpret_cast<M2*>(reinterpret_cast<SecConst*>(smem) + BQ_W * a.K) + (size_t)wave * a.K * 64;

    const int64_t rc = (int64_t)blockIdx.x * BQ_W + wave;
    const bool live = rc < a.total;
    const int64_t r = live ? rc / a.Cout : 0;
    const int c = live ? (int)(rc - r * a.Cout) : 0;
    const float* xr = x + brow_off(a.xmap, r, a.Cin == 1 ? 0 : c);
    float* yr = y + brow_off(a.ymap, r, c);
    const float* B = Bs + ((r * a.Cf + (a.Cf == 1 ? 0 : c)) * a.K) * 3;
    const float* A = As + ((r * a.Cf + (a.Cf == 1 ? 0 : c)) * a.K) * 3;

    for (int k = 0; k < a.K; ++k) {
        const float a0 = A[3 * k];
        const float a1 = A[3 * k + 1] / a0, a2 = A[3 * k + 2] / a0;
        M2d m = {-(double)a1, -(double)a2, 1.0, 0.0};
        m = mul(m, m);
        m = mul(m, m);
        M2d s = mul(m, m);  // M^8
        M2d p = {1.0, 0.0, 0.0, 1.0};
#pragma unroll
        for (int d = 0; d < 6; ++d) {
            if (lane == 0) sec[k].step[d] = narrow(s);
            if ((lane >> d) & 1) p = mul(p, s);
            s = mul(s, s);
        }
        lanepow[k * 64 + lane] = narrow(p);  // M^(8*lane)
        if (lane == 0) {
            sec[k].wave = narrow(s);  // M^512
            sec[k].b0 = B[3 * k] / a0;
            sec[k].b1 = B[3 * k + 1] / a0;
            sec[k].b2 = B[3 * k + 2] / a0;
            sec[k].a1 = a1;
            sec[k].a2 = a2;
            sec[k].carry = make_float2(0.0f, 0.0f);
        }
    }
    __syncthreads();  // the only barrier: tables written, every wave now works alone
    if (!live) return;

    using f4 = float __attribute__((ext_vector_type(4)));
    for (int64_t n0 = 0; n0 < a.L; n0 += BQ_TILE) {
        const int64_t n = n0 + BQ_E * lane;
        float v[BQ_E], x0[BQ_E];
        if (a.vec && n + BQ_E <= a.L) {
#pragma unroll
            for (int j = 0; j < BQ_E / 4; ++j) {
                const f4 q = *reinterpret_cast<const f4*>(xr + n + 4 * j);
                v[4 * j] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < BQ_E; ++i) v[i] = n + i < a.L ? xr[n + i] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < BQ_E; ++i) x0[i] = v[i];

        for (int k = 0; k < a.K; ++k) {
            SecConst& q = sec[k];
            const float a1 = q.a1, a2 = q.a2;
            float in[BQ_E];
#pragma unroll
            for (int i = 0; i < BQ_E; ++i) in[i] = a.quirk ? x0[i] : v[i];
            // 1. zero-state run of this lane's chunk
            float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
            for (int i = 0; i < BQ_E; ++i) {
                const float w = in[i] - a1 * s1 - a2 * s2;
                s2 = s1;
                s1 = w;
            }
            // 2. inclusive scan of the chunk end states across the wave
            float2 inc = make_float2(s1, s2);
#pragma unroll
            for (int d = 0; d < 6; ++d) {
                const float ux = __shfl_up(inc.x, 1 << d, 64), uy = __shfl_up(inc.y, 1 << d, 64);
                if (lane >= (1 << d)) {
                    const float2 m = apply(q.step[d], make_float2(ux, uy));
                    inc.x += m.x;
                    inc.y += m.y;
                }
            }
            float2 excl = make_float2(__shfl_up(inc.x, 1, 64), __shfl_up(inc.y, 1, 64));
            if (lane == 0) excl = make_float2(0.0f, 0.0f);
            // 3. carry: state entering the tile; its successor is lane 63's total plus M^512 * carry
            const float2 carry = q.carry;
            const float2 h = apply(lanepow[k * 64 + lane], carry);
            const float2 adv = apply(q.wave, carry);
            const float2 next = make_float2(__shfl(inc.x, 63, 64) + adv.x, __shfl(inc.y, 63, 64) + adv.y);
            if (lane == 0) q.carry = next;  // same-wave LDS accesses are ordered: read above, write here
            // 4. true state before this lane's first sample, rerun, numerator
            s1 = h.x + excl.x;
            s2 = h.y + excl.y;
            if (!a.quirk) {
#pragma unroll
                for (int i = 0; i < BQ_E; ++i) {
                    const float w = in[i] - a1 * s1 - a2 * s2;
                    v[i] = q.b0 * w + q.b1 * s1 + q.b2 * s2;
                    s2 = s1;
                    s1 = w;
                }
            } else {
                const float c1 = q.b1 - q.b0 * a1, c2 = q.b2 - q.b0 * a2;  // strictly proper part
#pragma unroll
                for (int i = 0; i < BQ_E; ++i) {
                    const float w = in[i] - a1 * s1 - a2 * s2;
                    v[i] = q.b0 * v[i] + c1 * s1 + c2 * s2;
                    s2 = s1;
                    s1 = w;
                }
            }
        }
        if (a.vec && n + BQ_E <= a.L) {
#pragma unroll
            for (int j = 0; j < BQ_E / 4; ++j)
                __builtin_nontemporal_store(f4{v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]},
                                            reinterpret_cast<f4*>(yr + n + 4 * j));
        } else {
#pragma unroll
            for (int i = 0; i < BQ_E; ++i)
                if (n + i < a.L) yr[n + i] = v[i];
        }
    }
}

static inline size_t bq_lds_bytes(int64_t K) {
    return (size_t)BQ_W * K * sizeof(SecConst) + (size_t)BQ_W * K * 64 * sizeof(M2);
}
static inline bool bq_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline bool bq_map_vec(const gfx_rowmap_t& m) {
    return m.stride_outer % 4 == 0 && m.stride_inner % 4 == 0 && m.stride_ch % 4 == 0;
}

}  // namespace gfx

using namespace gfx;

extern "C" {

int gfx_biquad_cascade_f32(const float* x, gfx_rowmap_t xmap, float* y, gfx_rowmap_t ymap, const float* Bs,
                           const float* As, int64_t R, int64_t C_in, int64_t C_f, int64_t K, int64_t L, int ssm_quirk,
                           void* stream) {
    if (!x || !y || !Bs || !As || R <= 0 || L <= 0 || K < 1 || K > BQ_MAX_K) return GFX_EINVAL;
    if (C_in < 1 || C_f < 1 || (C_in != C_f && C_in != 1 && C_f != 1)) return GFX_EINVAL;
    if (xmap.inner <= 0 || ymap.inner <= 0 || R > 0x7fffffffLL) return GFX_EINVAL;
    BqArgs a;
    a.xmap = xmap; a.ymap = ymap; a.L = L;
    a.Cin = (int)C_in; a.Cf = (int)C_f; a.Cout = (int)(C_in > C_f ? C_in : C_f);
    a.K = (int)K; a.quirk = ssm_quirk ? 1 : 0;
    a.vec = bq_aligned16(x) && bq_aligned16(y) && bq_map_vec(xmap) && bq_map_vec(ymap);
    a.total = R * a.Cout;
    const int64_t blocks = (a.total + BQ_W - 1) / BQ_W;
    if (blocks > 0x7fffffffLL) return GFX_EINVAL;
    const size_t lds = bq_lds_bytes(K);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(biquad_cascade_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return GFX_ELAUNCH;
    hipLaunchKernelGGL(biquad_cascade_kernel, dim3((unsigned)blocks), dim3(BQ_T), lds, (hipStream_t)stream, x, y, Bs, As, a);
    return hipGetLastError() == hipSuccess ? GFX_OK : GFX_ELAUNCH;
}

}  // extern "C"
