// The 8192-point LDS FFT tile of fft_tile.hpp in double precision -- same decomposition (8192 = 32 x 16 x 16, 256
// threads), same LDS images (S1 [32][272], S2 [512][18], here of double2: 147,456 B, one workgroup per CU), same
// register layout on entry and exit, plain v_fma_f64 arithmetic (the packed-FP32 tricks have no 64-bit counterpart;
// the vector FP64 rate of gfx950 is half the packed FP32 rate).  Used by the `precise` form of the odd-length aliasing
// (czt.hip): the energy envelope of the dynamics processors feeds log() and a gain curve, where the ~1e-6-of-peak
// noise floor of any fp32 transform pair is amplified beyond the parity bound on quiet passages.  Not on the hot path.
#pragma once
#include <hip/hip_runtime.h>

#include "fft_tile.hpp"

namespace gfx {

using cxd = double __attribute__((ext_vector_type(2)));
// GFX_F64_SPLIT (round 6, default): every LDS exchange of the tile moves the real parts and the imaginary parts in two
// rounds through an image of DOUBLES -- 73,728 B, the float tile's footprint, so that TWO workgroups share a CU and one's
// loads and stores run under the other's arithmetic (with the 147,456-byte double2 images a CU held one workgroup whose
// load, transform and store phases ran strictly one after the other: 20 us per tile, half memory, half arithmetic).  The
// price is four barriers per exchange instead of one or two, and 256 registers per thread instead of 461: the twiddles are
// fetched per pass (tile_twiddles_1 / _2) instead of living through the tile.  -DGFX_F64_SPLIT=0: the one-round images.
#ifndef GFX_F64_SPLIT
#define GFX_F64_SPLIT 1
#endif
constexpr int TILE_LDS_BYTES_F64 = TILE_LDS_F2 * (GFX_F64_SPLIT ? 8 : 16);      // 73,728 B (147,456 B in one round)

__device__ __forceinline__ cxd to_cx(double2 a) { return cxd{a.x, a.y}; }
__device__ __forceinline__ cxd cmul(cxd a, cxd w) { return cxd{a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x}; }
__device__ __forceinline__ cxd cmulc(cxd a, cxd w) { return cxd{a.x * w.x + a.y * w.y, a.y * w.x - a.x * w.y}; }
__device__ __forceinline__ cxd mul_neg_i(cxd a) { return cxd{a.y, -a.x}; }
__device__ __forceinline__ cxd mul_pos_i(cxd a) { return cxd{-a.y, a.x}; }

__device__ constexpr double kCos32d[16] = {
    1.0, 0.98078528040323044913, 0.92387953251128675613, 0.83146961230254523708,
    0.70710678118654752440, 0.55557023301960222474, 0.38268343236508977173, 0.19509032201612826785,
    0.0, -0.19509032201612826785, -0.38268343236508977173, -0.55557023301960222474,
    -0.70710678118654752440, -0.83146961230254523708, -0.92387953251128675613, -0.98078528040323044913};
__device__ constexpr double kSin32d[16] = {
    0.0, 0.19509032201612826785, 0.38268343236508977173, 0.55557023301960222474,
    0.70710678118654752440, 0.83146961230254523708, 0.92387953251128675613, 0.98078528040323044913,
    1.0, 0.98078528040323044913, 0.92387953251128675613, 0.83146961230254523708,
    0.70710678118654752440, 0.55557023301960222474, 0.38268343236508977173, 0.19509032201612826785};

template <bool INV>
__device__ __forceinline__ cxd tw32(cxd d, int idx32) {
    if (idx32 == 0) return d;
    if (idx32 == 16) return -d;
    if (idx32 == 8) return INV ? mul_pos_i(d) : mul_neg_i(d);
    if (idx32 == 24) return INV ? mul_neg_i(d) : mul_pos_i(d);
    const double sg = idx32 >= 16 ? -1.0 : 1.0;
    const double c = sg * kCos32d[idx32 & 15], s = sg * (INV ? kSin32d[idx32 & 15] : -kSin32d[idx32 & 15]);
    return cxd{d.x * c - d.y * s, d.y * c + d.x * s};
}

// In-register radix-2 DIF DFT of N points; result for frequency k at v[brev(k)]
template <int N, bool INV>
__device__ __forceinline__ void dif(cxd (&v)[N]) {
#pragma unroll
    for (int len = N; len >= 2; len >>= 1) {
        const int half = len >> 1;
#pragma unroll
        for (int base = 0; base < N; base += len) {
#pragma unroll
            for (int j = 0; j < half; ++j) {
                const cxd a = v[base + j], b = v[base + j + half];
                v[base + j] = a + b;
                v[base + j + half] = tw32<INV>(a - b, j * (32 / len));
            }
        }
    }
}

struct TileTwD {
    cxd lo1[4], hi1[8];  // W_8192^(t*i), W_8192^(t*4i)
    cxd lo2[4], hi2[4];  // W_256^(d*i),  W_256^(d*4i), d = t & 15

    __device__ __forceinline__ cxd fwd1(cxd e, int k1) const { return apply<false>(e, lo1[k1 & 3], hi1[k1 >> 2], k1 & 3, k1 >> 2); }
    __device__ __forceinline__ cxd inv1(cxd e, int k1) const { return apply<true>(e, lo1[k1 & 3], hi1[k1 >> 2], k1 & 3, k1 >> 2); }
    __device__ __forceinline__ cxd fwd2(cxd e, int k2) const { return apply<false>(e, lo2[k2 & 3], hi2[k2 >> 2], k2 & 3, k2 >> 2); }
    __device__ __forceinline__ cxd inv2(cxd e, int k2) const { return apply<true>(e, lo2[k2 & 3], hi2[k2 >> 2], k2 & 3, k2 >> 2); }

    template <bool CONJ>
    static __device__ __forceinline__ cxd apply(cxd e, cxd lo, cxd hi, int il, int ih) {
        if (il == 0 && ih == 0) return e;
        const cxd w = il == 0 ? hi : (ih == 0 ? lo : cmul(lo, hi));
        return CONJ ? cmulc(e, w) : cmul(e, w);
    }
};

// table: TW_ROWS x 256 double2, row-major [row][t]; rows as in fft_tile.hpp (0-3 lo1, 4-11 hi1, 12-15 lo2, 16-19 hi2)
__device__ __forceinline__ void tile_twiddles(TileTwD& tw, const double2* __restrict__ table, int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) tw.lo1[i] = to_cx(table[i * TILE_T + t]);
#pragma unroll
    for (int i = 0; i < 8; ++i) tw.hi1[i] = to_cx(table[(4 + i) * TILE_T + t]);
#pragma unroll
    for (int i = 0; i < 4; ++i) tw.lo2[i] = to_cx(table[(12 + i) * TILE_T + t]);
#pragma unroll
    for (int i = 0; i < 4; ++i) tw.hi2[i] = to_cx(table[(16 + i) * TILE_T + t]);
}

__device__ __forceinline__ void tile_twiddles_1(TileTwD& tw, const double2* __restrict__ table, int t) {   // pass 1's
#pragma unroll
    for (int i = 0; i < 4; ++i) tw.lo1[i] = to_cx(table[i * TILE_T + t]);
#pragma unroll
    for (int i = 0; i < 8; ++i) tw.hi1[i] = to_cx(table[(4 + i) * TILE_T + t]);
}
__device__ __forceinline__ void tile_twiddles_2(TileTwD& tw, const double2* __restrict__ table, int t) {   // pass 2's
#pragma unroll
    for (int i = 0; i < 4; ++i) tw.lo2[i] = to_cx(table[(12 + i) * TILE_T + t]);
#pragma unroll
    for (int i = 0; i < 4; ++i) tw.hi2[i] = to_cx(table[(16 + i) * TILE_T + t]);
}

#if GFX_F64_SPLIT
// (the scheduler may not carry loads or twiddles of a later phase across these points: at 256 registers there is no room)
__device__ __forceinline__ void f64_fence() { __builtin_amdgcn_sched_barrier(0); }
// The same two transforms with every exchange in two rounds (real parts, then imaginary parts) through `lds` read as an
// image of doubles; `table`: the twiddle table, from which each pass fetches what it needs.
__device__ __forceinline__ void tile_forward(cxd (&v)[32], cxd (&w)[2][16], const double2* __restrict__ table, cxd* lds_c, int t) {
    double* lds = reinterpret_cast<double*>(lds_c);
    TileTwD tw;
    tile_twiddles_1(tw, table, t);
    dif<32, false>(v);
#pragma unroll
    for (int r = 0; r < 32; ++r) v[r] = tw.fwd1(v[r], brev(r, 5));
    const int kk = t >> 4, d = t & 15;
#pragma unroll
    for (int part = 0; part < 2; ++part) {
#pragma unroll
        for (int r = 0; r < 32; ++r) lds[s1_at(brev(r, 5), t)] = v[r][part];
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int c = 0; c < 16; ++c) w[s][c][part] = lds[s1_at(kk + 16 * s, 16 * c + d)];
        }
        __syncthreads();
    }
    f64_fence();
    tile_twiddles_2(tw, table, t);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        dif<16, false>(w[s]);
#pragma unroll
        for (int r = 0; r < 16; ++r) w[s][r] = tw.fwd2(w[s][r], brev(r, 4));
    }
    f64_fence();
    cxd u[2][16];
#pragma unroll
    for (int part = 0; part < 2; ++part) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int r = 0; r < 16; ++r) lds[s2_row(brev(r, 4), kk + 16 * s) + d] = w[s][r][part];
        }
        __syncthreads();
#pragma unroll
        for (int bf = 0; bf < 2; ++bf) {
            const int j = bf ? bf_b(t) : bf_a(t);
            const double* row = lds + s2_row(j >> 5, j & 31);
#pragma unroll
            for (int q = 0; q < 16; ++q) u[bf][q][part] = row[q];
        }
        __syncthreads();
    }
    f64_fence();
#pragma unroll
    for (int bf = 0; bf < 2; ++bf) {
#pragma unroll
        for (int q = 0; q < 16; ++q) w[bf][q] = u[bf][q];
        dif<16, false>(w[bf]);
    }
}

__device__ __forceinline__ void tile_inverse(cxd (&w)[2][16], cxd (&v)[32], const double2* __restrict__ table, cxd* lds_c, int t) {
    double* lds = reinterpret_cast<double*>(lds_c);
    TileTwD tw;
    cxd p[2][16];
#pragma unroll
    for (int bf = 0; bf < 2; ++bf) {
#pragma unroll
        for (int k = 0; k < 16; ++k) p[bf][k] = w[bf][brev(k, 4)];
        dif<16, true>(p[bf]);
    }
    const int kk = t >> 4, d = t & 15;
    f64_fence();
#pragma unroll
    for (int part = 0; part < 2; ++part) {
#pragma unroll
        for (int bf = 0; bf < 2; ++bf) {
            const int j = bf ? bf_b(t) : bf_a(t);
            double* row = lds + s2_row(j >> 5, j & 31);
#pragma unroll
            for (int q = 0; q < 16; ++q) row[q] = p[bf][brev(q, 4)][part];
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int k2 = 0; k2 < 16; ++k2) w[s][k2][part] = lds[s2_row(k2, kk + 16 * s) + d];
        }
        __syncthreads();
    }
    f64_fence();
    tile_twiddles_2(tw, table, t);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) w[s][k2] = tw.inv2(w[s][k2], k2);
        dif<16, true>(w[s]);
    }
    f64_fence();
#pragma unroll
    for (int part = 0; part < 2; ++part) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int r = 0; r < 16; ++r) lds[s1_at(kk + 16 * s, 16 * brev(r, 4) + d)] = w[s][r][part];
        }
        __syncthreads();
#pragma unroll
        for (int k1 = 0; k1 < 32; ++k1) v[k1][part] = lds[s1_at(k1, t)];
        __syncthreads();
    }
    f64_fence();
    tile_twiddles_1(tw, table, t);
#pragma unroll
    for (int k1 = 0; k1 < 32; ++k1) v[k1] = tw.inv1(v[k1], k1);
    dif<32, true>(v);
}
#endif

// Forward: v[a] = z[t + 256*a]  ->  w[bf][brev4(k3)] = Z[j_bf + 512*k3]   (see fft_tile.hpp)
__device__ __forceinline__ void tile_forward(cxd (&v)[32], cxd (&w)[2][16], const TileTwD& tw, cxd* lds, int t) {
    dif<32, false>(v);
#pragma unroll
    for (int r = 0; r < 32; ++r) {
        const int k1 = brev(r, 5);
        lds[s1_at(k1, t)] = tw.fwd1(v[r], k1);
    }
    __syncthreads();
    const int kk = t >> 4, d = t & 15;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int c = 0; c < 16; ++c) w[s][c] = lds[s1_at(kk + 16 * s, 16 * c + d)];
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        dif<16, false>(w[s]);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k2 = brev(r, 4);
            lds[s2_row(k2, kk + 16 * s) + d] = tw.fwd2(w[s][r], k2);
        }
    }
    __syncthreads();
#pragma unroll
    for (int bf = 0; bf < 2; ++bf) {
        const int j = bf ? bf_b(t) : bf_a(t);
        const cxd* row = lds + s2_row(j >> 5, j & 31);
#pragma unroll
        for (int q = 0; q < 16; ++q) w[bf][q] = row[q];
        dif<16, false>(w[bf]);
    }
}

// Inverse (unnormalised): the layout tile_forward leaves -> v[brev5(a)] = z'[t + 256*a]
__device__ __forceinline__ void tile_inverse(cxd (&w)[2][16], cxd (&v)[32], const TileTwD& tw, cxd* lds, int t) {
#pragma unroll
    for (int bf = 0; bf < 2; ++bf) {
        cxd p[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) p[k] = w[bf][brev(k, 4)];
        dif<16, true>(p);
        const int j = bf ? bf_b(t) : bf_a(t);
        cxd* row = lds + s2_row(j >> 5, j & 31);
#pragma unroll
        for (int q = 0; q < 16; ++q) row[q] = p[brev(q, 4)];
    }
    __syncthreads();
    const int kk = t >> 4, d = t & 15;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) w[s][k2] = tw.inv2(lds[s2_row(k2, kk + 16 * s) + d], k2);
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        dif<16, true>(w[s]);
#pragma unroll
        for (int r = 0; r < 16; ++r) lds[s1_at(kk + 16 * s, 16 * brev(r, 4) + d)] = w[s][r];
    }
    __syncthreads();
#pragma unroll
    for (int k1 = 0; k1 < 32; ++k1) v[k1] = tw.inv1(lds[s1_at(k1, t)], k1);
    dif<32, true>(v);
}

}  // namespace gfx
