// The 8192-point LDS FFT tile of fft_tile.hpp on a 512-thread workgroup, 16 complex points per thread ("wide" tile).
//
//   Same transform, same 73,728-byte exchange image (S1 32 x 272, S2 512 x 18, aliased), but half the registers per
//   thread: a kernel built on it fits 128 VGPRs, i.e. FOUR waves per SIMD (2 workgroups x 8 waves per CU) where the
//   256-thread tile holds two.  The radix-32 pass is split over lane pairs (t, t^1): a radix-2 step across the pair
//   (one DPP quad-perm move per dword) and a radix-16 codelet in each lane.
//
//   Roles of thread t:   pass 1: column b = t >> 1, half h = t & 1   (points A = a' + 16 h, a' = 0..15; the lane
//                                 ends up with the outputs k1 = 2 m + h)
//                        pass 2: k1 = t >> 4, d = t & 15             (16 values c, b = 16 c + d)
//                        pass 3: S2 row t = rho(j): lanes (2 i, 2 i + 1) hold the butterflies j = i and 512 - i
//   Thread layout of the spectrum: w[brev4(k3)] = Z[j(t) + 512 k3].  The mirror bin M - k of the bin in w[r] is register
//   15 - r of lane t ^ 1 -- one DPP move away; lanes 0 and 1 hold the two self-mirrored butterflies j = 0 and 256.
//
//   The spectral product uses the one-output form  Z'[k] = alpha_k Z[k] + beta_k conj(Z[M-k])  with
//   alpha = 2 He + i (1 - W^k) Ho, beta = i (1 + W^k) Ho  (He, Ho as stored by hspec_kernel, i.e. scaled by 1/(4M)):
//   every lane computes its own 16 bins from its own (alpha, beta) pairs, 4 packed instructions per bin.
//
// tools/fft_tile512_model.py is the numpy model of this index math (round 1's tools/experiments/tile512 measured the
// scalar-arithmetic version of the idea).
#pragma once
#include <hip/hip_runtime.h>

#include "fft_tile.hpp"

namespace gfx {
namespace wide {

constexpr int WT = 512;                 // threads per workgroup
constexpr int WE = 16;                  // complex points per thread
constexpr int W_TW_ROWS = 17;           // twiddle table rows (x 512 float2)
constexpr int W_XQ_F2 = 32;             // float2 slots behind the tile image: spectra of the two self-mirrored butterflies
constexpr int W_TW2_F2 = 128;           // float2 slots of the pass-2 twiddle table in LDS (8 values x 16 d)
constexpr int W_LDS_BYTES = TILE_LDS_BYTES + (W_XQ_F2 + W_TW2_F2) * 8;         // one tile per workgroup
constexpr int W_H_F4 = WE * WT;         // float4 {alpha, beta} per filter (128 KB)

// value of the same register in lane t^1 (DPP quad_perm [1,0,3,2])
__device__ __forceinline__ float lane_xor1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ cx lane_xor1(cx v) { return cx{lane_xor1(v.x), lane_xor1(v.y)}; }

// acc + a * conj(w)
__device__ __forceinline__ cx cmacc(cx acc, cx a, cx w) {
    cx t, r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(t) : "v"(a), "v"(w), "v"(acc));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}

__host__ __device__ __forceinline__ int j_of(int t) {
    return t == 0 ? 0 : (t == 1 ? 256 : ((t & 1) ? 512 - (t >> 1) : (t >> 1)));
}
__device__ __forceinline__ int rho(int j) { return j == 0 ? 0 : (j == 256 ? 1 : (j < 256 ? 2 * j : 2 * (512 - j) + 1)); }

// Twiddle table (W_TW_ROWS x 512 float2, filled once per device in double precision, common.hip):
//   rows 0-3  W_8192^(b (h + 2 i))   rows 4-7  W_8192^(8 b i)   rows 8-11 W_256^(d i)   rows 12-15 W_256^(4 d i)
//   row 16    W_8192^(j(t))          (b = t >> 1, h = t & 1, d = t & 15)
const float2* tile512_twiddle_table(hipStream_t stream);

// two-level twiddle set W^(x i) = lo[i & 3] * hi[i >> 2]: rows 0-7 of the table are the pass-1 set (tw1), rows 8-15 the
// pass-2 set (tw2).  The callers fetch both ahead of the filter spectrum (vector loads return in order: a twiddle
// requested behind the 16 spectrum loads would wait for all of them) and again at the start of the inverse transform
struct Tw4x4 {
    cx lo[4], hi[4];
    template <bool CONJ>
    __device__ __forceinline__ cx apply(cx e, int i, bool lo_is_one) const {
        const bool hi_is_one = (i >> 2) == 0;
        if (lo_is_one && hi_is_one) return e;
        const cx w = lo_is_one ? hi[i >> 2] : (hi_is_one ? lo[i & 3] : cmul(lo[i & 3], hi[i >> 2]));
        return CONJ ? cmulc(e, w) : cmul(e, w);
    }
};
__device__ __forceinline__ void load_tw(Tw4x4& tw, const float2* __restrict__ table, int first_row, int t) {
    const cx* tab = reinterpret_cast<const cx*>(table);
#pragma unroll
    for (int i = 0; i < 4; ++i) tw.lo[i] = tab[(first_row + i) * WT + t];
#pragma unroll
    for (int i = 0; i < 4; ++i) tw.hi[i] = tab[(first_row + 4 + i) * WT + t];
}

// The pass-2 set depends on d = t & 15 only: 8 x 16 values, kept in LDS (filled once per workgroup, fill_tw2; visible after
// the first barrier of forward_2).  A pass-2 twiddle fetched from memory would queue behind the sixteen spectrum loads
// (vector loads return in order) and stall the second pass until the whole spectrum has arrived.
__device__ __forceinline__ void fill_tw2(cx* tw2tab, const float2* __restrict__ table, int t) {
    if (t < W_TW2_F2) tw2tab[t] = reinterpret_cast<const cx*>(table)[(8 + (t >> 4)) * WT + (t & 15)];
}
__device__ __forceinline__ void read_tw2(Tw4x4& tw2, const cx* tw2tab, int d) {
#pragma unroll
    for (int i = 0; i < 4; ++i) tw2.lo[i] = tw2tab[16 * i + d];
#pragma unroll
    for (int i = 0; i < 4; ++i) tw2.hi[i] = tw2tab[16 * (4 + i) + d];
}

// ---- the passes, for NT tiles at once (NT = 1 or 2).  Tile i uses the LDS image at lds + i * W_IMG_F2.  With NT = 2
// every barrier serves both tiles and each wave carries two independent dependency chains (the dual kernel).
constexpr int W_IMG_F2 = TILE_LDS_F2 + W_XQ_F2;   // one exchange image + its xq slots, float2 units

// Forward, first pass: p[i][a'] = z_i[256 (a' + 16 h) + b]  ->  S1.  After it p is dead (the caller may issue loads here).
template <int NT>
__device__ __forceinline__ void forward_1(cx (&p)[NT][16], const Tw4x4& tw1, cx* lds, int t) {
    const int b = t >> 1;
    const bool odd = t & 1;
    // radix-2 across the lane pair: the even lane keeps z[a'] + z[a'+16], the odd lane (z[a'] - z[a'+16]) W_32^a'
    const float sg = odd ? -1.0f : 1.0f;
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int a = 0; a < 16; ++a) p[i][a] = p[i][a] * cx{sg, sg} + lane_xor1(p[i][a]);
    if (odd) {
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int a = 1; a < 16; ++a) p[i][a] = tw32<false>(p[i][a], a);
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        dif<16, false>(p[i]);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = brev(r, 4);
            lds[i * W_IMG_F2 + s1_at(2 * m + (t & 1), b)] = tw1.apply<false>(p[i][r], m, false);
        }
    }
}
// Forward, second pass: S1 -> S2.  Ends with a barrier; afterwards thread t reads S2 row t (forward_3).
template <int NT>
__device__ __forceinline__ void forward_2(const cx* tw2tab, cx* lds, int t) {
    __syncthreads();
    const int k1 = t >> 4, d = t & 15;
    Tw4x4 tw2;
    read_tw2(tw2, tw2tab, d);
    cx u[NT][16];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int c = 0; c < 16; ++c) u[i][c] = lds[i * W_IMG_F2 + s1_at(k1, 16 * c + d)];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        dif<16, false>(u[i]);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k2 = brev(r, 4);
            lds[i * W_IMG_F2 + rho(k1 + 32 * k2) * S2_ROW + d] = tw2.apply<false>(u[i][r], k2, (k2 & 3) == 0);
        }
    }
    __syncthreads();
}
// Forward, third pass: w[i][brev4(k3)] = Z_i[j(t) + 512 k3].  Lanes 0 / 1 also leave their spectra in the xq slots.
template <int NT>
__device__ __forceinline__ void forward_3(cx (&w)[NT][16], cx* lds, int t) {
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const f4v* row = reinterpret_cast<const f4v*>(lds + i * W_IMG_F2 + t * S2_ROW);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f4v v = row[q];
            w[i][2 * q] = v.lo;
            w[i][2 * q + 1] = v.hi;
        }
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        dif<16, false>(w[i]);
        if (t < 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) lds[i * W_IMG_F2 + TILE_LDS_F2 + 16 * t + r] = w[i][r];
        }
    }
}

// Z[M - k] for the bin k in register r: register 15 - r of lane t ^ 1; lanes 0 / 1 read their own saved spectra.
// `wpartner` must be the ORIGINAL spectrum register 15 - r (both lanes of a pair update r and 15 - r in lockstep).
__device__ __forceinline__ cx mirror_of(cx wpartner, int r, const cx* img, int t) {
    cx q = lane_xor1(wpartner);
    if (t < 2) q = img[TILE_LDS_F2 + 16 * t + (t == 0 ? brev((16 - brev(r, 4)) & 15, 4) : 15 - r)];
    return q;
}

// w <- alpha w + beta conj(mirror), in place, pairs (r, 15 - r) together.  hab[r] = {alpha, beta} of register r.
__device__ __forceinline__ void spectral_product(cx (&w)[16], const f4v (&hab)[16], const cx* img, int t) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int r2 = 15 - r;
        const cx qa = mirror_of(w[r2], r, img, t), qb = mirror_of(w[r], r2, img, t);
        w[r] = cmacc(cmul(hab[r].lo, w[r]), hab[r].hi, qa);
        w[r2] = cmacc(cmul(hab[r2].lo, w[r2]), hab[r2].hi, qb);
    }
}

// Inverse (unnormalised): w[i][brev4(k3)] = Z'_i[j(t) + 512 k3]  ->  v[i][a'] = z'_i[256 (a' + 16 h) + b].
// Thread t writes S2 row t, the row it alone read in forward_3: no barrier needed in between.
template <int NT>
__device__ __forceinline__ void inverse(cx (&w)[NT][16], cx (&v)[NT][16], const cx* tw2tab, const Tw4x4& tw1, cx* lds,
                                        int t) {
    Tw4x4 tw2;
    read_tw2(tw2, tw2tab, t & 15);
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        cx p[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) p[k] = w[i][brev(k, 4)];  // register renaming only
        dif<16, true>(p);
        f4v* row = reinterpret_cast<f4v*>(lds + i * W_IMG_F2 + t * S2_ROW);
#pragma unroll
        for (int q = 0; q < 8; ++q) row[q] = __builtin_shufflevector(p[brev(2 * q, 4)], p[brev(2 * q + 1, 4)], 0, 1, 2, 3);
    }
    __syncthreads();
    const int k1 = t >> 4, d = t & 15;
    cx u[NT][16];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2)
            u[i][k2] = tw2.apply<true>(lds[i * W_IMG_F2 + rho(k1 + 32 * k2) * S2_ROW + d], k2, (k2 & 3) == 0);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        dif<16, true>(u[i]);
#pragma unroll
        for (int r = 0; r < 16; ++r) lds[i * W_IMG_F2 + s1_at(k1, 16 * brev(r, 4) + d)] = u[i][r];
    }
    __syncthreads();
    const int b = t >> 1;
    const bool odd = t & 1;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        cx g[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) g[m] = tw1.apply<true>(lds[i * W_IMG_F2 + s1_at(2 * m + (t & 1), b)], m, false);
        dif<16, true>(g);
#pragma unroll
        for (int a = 0; a < 16; ++a) v[i][a] = g[brev(a, 4)];
    }
    // the even lane holds E[a'], the odd lane O[a']: z[a'] = E + conj(W_32^a') O, z[a'+16] = E - conj(W_32^a') O
    if (odd) {
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int a = 1; a < 16; ++a) v[i][a] = tw32<true>(v[i][a], a);
    }
    const float sg = odd ? -1.0f : 1.0f;
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int a = 0; a < 16; ++a) v[i][a] = v[i][a] * cx{sg, sg} + lane_xor1(v[i][a]);
}

}  // namespace wide
}  // namespace gfx
