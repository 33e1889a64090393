// Exact time-domain biquad cascade as a parallel scan (gfx950).
//
// Replaces IIRFilter._process_lfilter / _process_ssm — reference core/iir.py:154-261 — whose upstream
// implementations call torchaudio.functional.lfilter / torchlpc.sample_wise_lpc (neither is installed here):
// K second-order sections applied in series to every row-channel,
//     w[n] = x[n] - a1 w[n-1] - a2 w[n-2],   y[n] = b0 w[n] + b1 w[n-1] + b2 w[n-2]      (a0-normalised)
// with zero initial state.  No FFT, no truncation of the impulse response (the FSM backend aliases it to
// fsm_fir_len taps), 8 B of HBM traffic per channel-sample.
//
// Parallelisation: RL lanes walk TWO row-channels (packed fp32: the two channels of a stereo row, or two neighbouring mono
// rows) in tiles of RL x 8 samples, no barriers and no cross-lane LDS traffic in the time loop.  RL = 64 (one wave per pair,
// 512-sample tiles, the next tile requested ahead) unless there are pairs enough (>= 8192) to fill the chip with RL = 16
// (four pairs per wave, 128-sample tiles): a pair's tiles are a sequential chain, and with few pairs the time is that
// chain's latency.  The recursion is the linear system s[n] = M s[n-1] + (x[n], 0), M = [[-a1, -a2], [1, 0]],
// s = (w[n], w[n-1]):
//   1. each lane runs its 8 samples from a zero state                 -> end state e_t
//   2. Hillis-Steele scan over the 16 lanes of a DPP row with M^(8*2^d), d < 4 -> state at the end of every chunk.  The
//      shifted operands come from DPP row shifts (row_shr:1/2/4/8, zero fill: the lanes a step does not reach add zero),
//      i.e. from the vector ALU's own lane crossbar.  RL = 64: two more steps carry the rows' totals across (row_bcast:15
//      into rows 1 and 3, row_bcast:31 into rows 2 and 3), weighted per lane with M^(8 (l % 16 + 1)) from a 16-entry table.
//      The state entering the tile (the carry) is injected at lane 0 (its end state += M^8 * carry), so the scan delivers
//      every lane's TRUE entering state (one lane below: row_shr:1 / wave_shr:1) and the last lane's total is the next
//      tile's carry (row_ror:1 / v_readlane)
//   3. every lane reruns its 8 samples from that state and applies the numerator.
// The matrix powers are formed in double per (row-channel, section) when the workgroup starts and live in LDS (256 B per
// pair and section, + 512 B for RL = 64).  Rounds 2-3 scanned over the 64 lanes with __shfl_up = ds_bpermute_b32: 16 trips
// through the LDS crossbar per section and tile of a row-channel, ~9 cycles each of the CU's ONE LDS unit -- that, not
// arithmetic or HBM, was the kernel's time (K = 6 at 8192 x 2 x 131072: 7.3 ms = 2.3 TB/s; two row-channels per wave in
// packed fp32 alone changed nothing).
//
// ssm_quirk: upstream's "ssm" backend drives the recursive part of every section with the ORIGINAL input
// instead of the previous section's output (core/iir.py:226-246 index `input_signal`, not `x`); for K = 1
// the two backends agree, for K > 1 this flag reproduces what "ssm" actually returns.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/grafx_amd.h"

namespace gfx {

constexpr int BQ_T = 64;             // threads per workgroup: one wave
constexpr int BQ_E = 8;              // samples per lane
constexpr int BQ_MAX_K = 32;
// Lanes per pair of row-channels, RL: 64 (the whole wave) or 16 (one DPP row; four pairs per wave) -- see the head of the file;
// 2048 stereo rows, K = 6: 1.5 ms with RL = 64 against 3.6 with RL = 16; 8192 rows, K = 2: 3.7 with RL = 16 against 4.5.

using f2 = float __attribute__((ext_vector_type(2)));   // (row-channel 2p, row-channel 2p + 1)
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

struct M2d {
    double a, b, c, d;
};
__device__ __forceinline__ M2d mul(const M2d& x, const M2d& y) {
    return {x.a * y.a + x.b * y.c, x.a * y.b + x.b * y.d, x.c * y.a + x.d * y.c, x.c * y.b + x.d * y.d};
}

struct SecConst {       // per (pair, section), in LDS; [4] = the matrix entries a, b, c, d
    f2 step[6][4];      // M^(E * 2^d)  (RL = 16 uses d < 4)
    f2 b0, b1, b2, a1, a2;
    f2 carry1, carry2;  // state (w[n-1], w[n-2]) entering the current tile
    f2 pad;
};
static_assert(sizeof(SecConst) == 256, "SecConst layout");
// LDS: SecConst sec[64 / RL][K]; for RL = 64 also f2 rowpow[K][16][4] = M^(E (j + 1)), j < 16

// (s1', s2') = M (s1, s2), the product rounded once before the fused multiply-add (as the scalar form did)
__device__ __forceinline__ void apply2(const f2 (&m)[4], f2 s1, f2 s2, f2& o1, f2& o2) {
    o1 = fma2(m[0], s1, m[1] * s2);
    o2 = fma2(m[2], s1, m[3] * s2);
}
// the value of the lane CTRL selects: 0x110 + n: n lanes below within the 16-lane row, 0x121: the row rotated right by one
// (lane 0 reads lane 15), 0x138: one lane below in the whole wave, 0x142 / 0x143: lane 15 of a row to the next row / lane 31
// to rows 2 and 3; zero where there is no such lane and in the rows ROWS (a mask of the wave's four) leaves out
template <int CTRL, int ROWS = 0xf>
__device__ __forceinline__ f2 dpp2(f2 v) {
    // (one 64-bit move, which the back end splits into two v_mov_b32_dpp: given two 32-bit builtins on .x and .y, hipcc
    // 7.2 folds the second into a copy of the first)
    const long long w = __builtin_amdgcn_update_dpp(0LL, __builtin_bit_cast(long long, v), CTRL, ROWS, 0xf, ROWS == 0xf);
    return __builtin_bit_cast(f2, w);
}

// lane 63's value, wave-uniform.  (The halves go through named floats: __builtin_bit_cast(int, v.y) of a vector element
// reads v.x with hipcc 7.2 -- the same front-end slip that made two 32-bit DPP builtins on .x / .y one.)
__device__ __forceinline__ f2 last_lane(f2 v) {
    const float x = v.x, y = v.y;
    return f2{__int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63)),
              __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y), 63))};
}

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

template <int RL, bool AHEAD>
__global__ __launch_bounds__(BQ_T) void biquad_cascade_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                              const float* __restrict__ Bs,
                                                              const float* __restrict__ As, BqArgs a) {
    constexpr int BQ_PW = BQ_T / RL, BQ_TILE = RL * BQ_E, STEPS = RL == 16 ? 4 : 5;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int t = threadIdx.x, rl = t & (RL - 1), pw = t / RL;
    SecConst* sec = reinterpret_cast<SecConst*>(smem) + pw * a.K;
    f2* rowpow = reinterpret_cast<f2*>(reinterpret_cast<SecConst*>(smem) + BQ_PW * a.K);   // (RL = 64)

    const int64_t rc0 = 2 * ((int64_t)blockIdx.x * BQ_PW + pw);
    const bool live[2] = {rc0 < a.total, rc0 + 1 < a.total};
    const float* xr[2];
    float* yr[2];
    const float *B[2], *A[2];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
        const int64_t rc = live[ch] ? rc0 + ch : (live[0] ? rc0 : 0);   // a missing partner re-reads its neighbour, stores nothing
        const int64_t r = rc / a.Cout;
        const int c = (int)(rc - r * a.Cout);
        xr[ch] = x + brow_off(a.xmap, r, a.Cin == 1 ? 0 : c);
        yr[ch] = y + brow_off(a.ymap, r, c);
        B[ch] = Bs + ((r * a.Cf + (a.Cf == 1 ? 0 : c)) * a.K) * 3;
        A[ch] = As + ((r * a.Cf + (a.Cf == 1 ? 0 : c)) * a.K) * 3;
    }

    // constants: lane (k, ch) of the pair's lanes takes section k + (RL / 2) j of row-channel ch
    for (int k = rl >> 1; k < a.K; k += RL / 2) {
        const int ch = rl & 1;
        const float a0 = A[ch][3 * k];
        const float a1 = A[ch][3 * k + 1] / a0, a2 = A[ch][3 * k + 2] / a0;
        M2d s = {-(double)a1, -(double)a2, 1.0, 0.0};
#pragma unroll
        for (int e = 1; e < BQ_E; e *= 2) s = mul(s, s);  // M^E
        if constexpr (RL == 64) {
            M2d pw_j = s;
            for (int j = 0; j < 16; ++j) {   // M^(E (j + 1))
                float* dst = reinterpret_cast<float*>(rowpow + (size_t)(k * 16 + j) * 4);
                dst[0 + ch] = (float)pw_j.a;
                dst[2 + ch] = (float)pw_j.b;
                dst[4 + ch] = (float)pw_j.c;
                dst[6 + ch] = (float)pw_j.d;
                pw_j = mul(pw_j, s);
            }
        }
#pragma unroll
        for (int d = 0; d < STEPS; ++d) {
            reinterpret_cast<float*>(&sec[k].step[d][0])[ch] = (float)s.a;
            reinterpret_cast<float*>(&sec[k].step[d][1])[ch] = (float)s.b;
            reinterpret_cast<float*>(&sec[k].step[d][2])[ch] = (float)s.c;
            reinterpret_cast<float*>(&sec[k].step[d][3])[ch] = (float)s.d;
            s = mul(s, s);
        }
        reinterpret_cast<float*>(&sec[k].b0)[ch] = B[ch][3 * k] / a0;
        reinterpret_cast<float*>(&sec[k].b1)[ch] = B[ch][3 * k + 1] / a0;
        reinterpret_cast<float*>(&sec[k].b2)[ch] = B[ch][3 * k + 2] / a0;
        reinterpret_cast<float*>(&sec[k].a1)[ch] = a1;
        reinterpret_cast<float*>(&sec[k].a2)[ch] = a2;
        reinterpret_cast<float*>(&sec[k].carry1)[ch] = 0.0f;
        reinterpret_cast<float*>(&sec[k].carry2)[ch] = 0.0f;
    }
    __syncthreads();  // the only barrier: tables written, every pair's lanes now work alone

    using f4 = float __attribute__((ext_vector_type(4)));
    const int64_t len = live[0] ? a.L : 0;   // (a pair past the end walks nothing: the DPP rows of a wave are independent)
    // the next tile's samples are requested before the running tile's sections are worked through: a pair's tiles are a
    // sequential chain (1024 of them at L = 131072), and with few rows there are no other waves to hide a load behind
    auto load = [&](int64_t n0, f2 (&w)[BQ_E]) {
        const int64_t n = n0 + BQ_E * rl;
        const bool whole = a.vec && n + BQ_E <= a.L;
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            float e[BQ_E];
            if (whole) {
#pragma unroll
                for (int j = 0; j < BQ_E / 4; ++j) {
                    const f4 q = *reinterpret_cast<const f4*>(xr[ch] + n + 4 * j);
                    e[4 * j] = q.x; e[4 * j + 1] = q.y; e[4 * j + 2] = q.z; e[4 * j + 3] = q.w;
                }
            } else {
#pragma unroll
                for (int i = 0; i < BQ_E; ++i) e[i] = n + i < a.L ? xr[ch][n + i] : 0.0f;
            }
#pragma unroll
            for (int i = 0; i < BQ_E; ++i) reinterpret_cast<float*>(&w[i])[ch] = e[i];
        }
    };
    // AHEAD (few rows): the next tile's samples are requested before the running tile's sections are worked through --
    // there are no other waves to hide a load behind.  (With rows enough to fill the chip the other waves do that, and the
    // extra registers cost more than the request ahead brings: K = 1 at 8192 stereo rows 4.2 against 5.1 ms.)
    f2 nxt[BQ_E];
    if (AHEAD && len > 0) load(0, nxt);
    for (int64_t n0 = 0; n0 < len; n0 += BQ_TILE) {
        const int64_t n = n0 + BQ_E * rl;
        const bool whole = a.vec && n + BQ_E <= a.L;
        f2 v[BQ_E], x0[BQ_E];
        if constexpr (AHEAD) {
#pragma unroll
            for (int i = 0; i < BQ_E; ++i) v[i] = nxt[i];
            if (n0 + BQ_TILE < len) load(n0 + BQ_TILE, nxt);
        } else {
            load(n0, v);
        }
#pragma unroll
        for (int i = 0; i < BQ_E; ++i) x0[i] = v[i];

        for (int k = 0; k < a.K; ++k) {
            SecConst& q = sec[k];
            const f2 a1 = q.a1, a2 = q.a2;
            f2 in[BQ_E];
#pragma unroll
            for (int i = 0; i < BQ_E; ++i) in[i] = a.quirk ? x0[i] : v[i];
            // 1. zero-state run of this lane's chunk
            f2 s1 = {0.0f, 0.0f}, s2 = {0.0f, 0.0f};
#pragma unroll
            for (int i = 0; i < BQ_E; ++i) {
                const f2 w = fma2(-a2, s2, fma2(-a1, s1, in[i]));
                s2 = s1;
                s1 = w;
            }
            // 2. inclusive scan of the chunk end states over the sixteen lanes, the tile's entering state riding along
            const f2 c1 = q.carry1, c2 = q.carry2;
            f2 i1 = s1, i2 = s2, m1, m2;
            if (rl == 0) {
                apply2(q.step[0], c1, c2, m1, m2);
                i1 += m1;
                i2 += m2;
            }
            apply2(q.step[0], dpp2<0x111>(i1), dpp2<0x111>(i2), m1, m2);   // within the 16-lane rows: the lanes a step does
            i1 += m1;                                                       // not reach read zeros
            i2 += m2;
            apply2(q.step[1], dpp2<0x112>(i1), dpp2<0x112>(i2), m1, m2);
            i1 += m1;
            i2 += m2;
            apply2(q.step[2], dpp2<0x114>(i1), dpp2<0x114>(i2), m1, m2);
            i1 += m1;
            i2 += m2;
            apply2(q.step[3], dpp2<0x118>(i1), dpp2<0x118>(i2), m1, m2);
            i1 += m1;
            i2 += m2;
            f2 nx1, nx2;
            if constexpr (RL == 64) {
                // across the four rows: lane l of a row still lacks M^(8 (l % 16 + 1)) x (the state at the end of the row
                // before).  Rows 1 and 3 take their neighbour's total (lane 15 -> next row); then lane 31 holds the true
                // state at the end of row 1, which rows 2 and 3 take (row 3 through the 128 samples of row 2).
                const f2* rp = rowpow + (size_t)(k * 16 + (rl & 15)) * 4;
                const f2 w[4] = {rp[0], rp[1], rp[2], rp[3]};
                apply2(w, dpp2<0x142, 0xa>(i1), dpp2<0x142, 0xa>(i2), m1, m2);
                i1 += m1;
                i2 += m2;
                f2 u1 = dpp2<0x143, 0xc>(i1), u2 = dpp2<0x143, 0xc>(i2);
                apply2(q.step[4], u1, u2, m1, m2);   // M^128
                if (rl >= 48) {
                    u1 = m1;
                    u2 = m2;
                }
                apply2(w, u1, u2, m1, m2);
                i1 += m1;
                i2 += m2;
                // this lane's entering state: the inclusive total of the lane below (lane 0: the carry itself)
                s1 = dpp2<0x138>(i1);
                s2 = dpp2<0x138>(i2);
                nx1 = last_lane(i1);   // the next tile's carry
                nx2 = last_lane(i2);
            } else {
                s1 = dpp2<0x111>(i1);
                s2 = dpp2<0x111>(i2);
                nx1 = dpp2<0x121>(i1);   // lane 0 <- lane 15: the next tile's carry
                nx2 = dpp2<0x121>(i2);
            }
            if (rl == 0) {
                s1 = c1;
                s2 = c2;
                q.carry1 = nx1;   // same-wave LDS accesses are ordered: read above, write here
                q.carry2 = nx2;
            }
            // 3. rerun from the true state, numerator
            if (!a.quirk) {
                const f2 b0 = q.b0, b1 = q.b1, b2 = q.b2;
#pragma unroll
                for (int i = 0; i < BQ_E; ++i) {
                    const f2 w = fma2(-a2, s2, fma2(-a1, s1, in[i]));
                    v[i] = fma2(b0, w, fma2(b1, s1, b2 * s2));
                    s2 = s1;
                    s1 = w;
                }
            } else {
                const f2 b0 = q.b0, cc1 = q.b1 - q.b0 * a1, cc2 = q.b2 - q.b0 * a2;  // strictly proper part
#pragma unroll
                for (int i = 0; i < BQ_E; ++i) {
                    const f2 w = fma2(-a2, s2, fma2(-a1, s1, in[i]));
                    v[i] = fma2(b0, v[i], fma2(cc1, s1, cc2 * s2));
                    s2 = s1;
                    s1 = w;
                }
            }
        }
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
            if (!live[ch]) continue;
            float e[BQ_E];
#pragma unroll
            for (int i = 0; i < BQ_E; ++i) e[i] = ch ? v[i].y : v[i].x;
            if (whole) {
#pragma unroll
                for (int j = 0; j < BQ_E / 4; ++j)
                    __builtin_nontemporal_store(f4{e[4 * j], e[4 * j + 1], e[4 * j + 2], e[4 * j + 3]},
                                                reinterpret_cast<f4*>(yr[ch] + n + 4 * j));
            } else {
#pragma unroll
                for (int i = 0; i < BQ_E; ++i)
                    if (n + i < a.L) yr[ch][n + i] = e[i];
            }
        }
    }
}

static inline size_t bq_lds_bytes(int64_t K, int RL) {
    return (size_t)(BQ_T / RL) * K * sizeof(SecConst) + (RL == 64 ? (size_t)K * 16 * 4 * sizeof(f2) : 0);
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
    const int64_t pairs = (a.total + 1) / 2;
    // sixteen lanes per pair where such waves fill the chip, else the whole wave
    const bool narrow = pairs >= 8192;
    const int RL = narrow ? 16 : 64;
    const int64_t blocks = (pairs + BQ_T / RL - 1) / (BQ_T / RL);
    if (blocks > 0x7fffffffLL) return GFX_EINVAL;
    const size_t lds = bq_lds_bytes(K, RL);
    auto kern = narrow ? biquad_cascade_kernel<16, false> : biquad_cascade_kernel<64, true>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess)
        return GFX_ELAUNCH;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(BQ_T), lds, (hipStream_t)stream, x, y, Bs, As, a);
    return hipGetLastError() == hipSuccess ? GFX_OK : GFX_ELAUNCH;
}

}  // extern "C"
