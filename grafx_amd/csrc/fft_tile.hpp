// LDS FFT tile for gfx950: one 8192-point complex FFT (= one 16384-sample real
// tile, two samples packed per complex point) per 256-thread workgroup.
//
//   8192 = 32 x 16 x 16.  Each thread owns 32 complex values in VGPRs; the three
//   radix passes run in registers and exchange through LDS twice per direction.
//   Forward is decimation-in-frequency, inverse is its mirror, so no digit
//   reversal is ever materialised: the spectrum lives in a private "thread
//   layout" (thread t holds butterflies j=t and j=512-t, j = k1 + 32*k2, bins
//   k = j + 512*k3) that pairs every bin k with its mirror M-k in the same
//   thread, which is what the real-signal (polyphase) product needs.
//
//   LDS images (float2 units), padded so that every access pattern below is
//   bank-conflict free for its instruction's lane groups (MI355X_MICROARCH §LDS):
//     S1[k1][b]      at k1*272 + b              (32 rows of 256 + 16 pad)
//     S2[k2][k1][d]  at (k2*32 + k1)*18 + d     (512 rows of 16 + 2 pad)
//   S1 and S2 alias the same 73,728-byte buffer -> 2 workgroups per CU.
//
// tools/fft_tile_model.py is the numpy model of exactly this index math.
#pragma once
#include <hip/hip_runtime.h>

namespace gfx {

constexpr int TILE_M = 8192;         // complex points per tile
constexpr int TILE_F = 16384;        // real samples per tile
constexpr int TILE_T = 256;          // threads per workgroup
constexpr int S1_ROW = 272;          // padded row of S1 (float2 units)
constexpr int S2_ROW = 18;           // padded row of S2 (float2 units)
constexpr int TILE_LDS_F2 = 512 * S2_ROW;             // 9216 float2
constexpr int TILE_LDS_BYTES = TILE_LDS_F2 * 8;       // 73,728 B
constexpr int H_SLOTS = 17;          // (He,Ho) pair slots per thread (slot 16: thread 0 only)
constexpr int H_TILE_F4 = H_SLOTS * TILE_T;           // float4 per (row-channel, partition)

// ---- complex arithmetic on packed FP32 (v_pk_*_f32) ---------------------------------------------
// A complex value is one even-aligned VGPR pair (re, im).  CDNA3/4 execute v_pk_add/mul/fma_f32 on
// both halves at the rate of one scalar op, and their op_sel / neg modifiers pick and negate halves for
// free, so a complex add is ONE instruction, a complex multiply TWO, and multiplications by +-i or a
// conjugation fold into the neighbouring add.  The compiler maps plain vector expressions (a + b,
// a * const) to the packed forms by itself; the forms that need a half-swap plus a one-sided negation
// it does not find, so those are spelled out below.  (Build with -fno-slp-vectorize: the automatic
// packing of the *scalar* code is slower than scalar code.)
using cx = float __attribute__((ext_vector_type(2)));
using f4v = float __attribute__((ext_vector_type(4)));

__device__ __forceinline__ cx to_cx(float2 a) { return cx{a.x, a.y}; }
__device__ __forceinline__ cx cadd(cx a, cx b) { return a + b; }
__device__ __forceinline__ cx csub(cx a, cx b) { return a - b; }
__device__ __forceinline__ cx cswap(cx a) { return __builtin_shufflevector(a, a, 1, 0); }
__device__ __forceinline__ cx cconj(cx a) { return a * cx{1.0f, -1.0f}; }
__device__ __forceinline__ cx mul_neg_i(cx a) { return cswap(a) * cx{1.0f, -1.0f}; }  // a * (-i)
__device__ __forceinline__ cx mul_pos_i(cx a) { return cswap(a) * cx{-1.0f, 1.0f}; }  // a * (+i)

#define GFX_PK2(name, insn)                                                  \
    __device__ __forceinline__ cx name(cx a, cx b) {                         \
        cx r;                                                                \
        asm(insn : "=v"(r) : "v"(a), "v"(b));                                \
        return r;                                                            \
    }
// (a - b) * (-i) = (a.y - b.y, b.x - a.x)
GFX_PK2(sub_mul_neg_i, "v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_lo:[0,1] neg_hi:[1,0]")
// (a - b) * (+i) = (b.y - a.y, a.x - b.x)
GFX_PK2(sub_mul_pos_i, "v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_lo:[1,0] neg_hi:[0,1]")
// a + conj(b) = (a.x + b.x, a.y - b.y)
GFX_PK2(add_conj, "v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]")
// (a - conj(b)) * (-i) = (a.y + b.y, b.x - a.x)
GFX_PK2(sub_conj_mul_neg_i, "v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_hi:[1,0]")
// a + i*b = (a.x - b.y, a.y + b.x)
GFX_PK2(add_mul_pos_i, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]")
// conj(a - i*b) = (a.x + b.y, b.x - a.y)
GFX_PK2(conj_sub_mul_pos_i, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[1,0]")
#undef GFX_PK2

// a * w:  t = (a.x w.x, a.y w.x);  r = (a.y * -w.y + t.x, a.x * w.y + t.y)
__device__ __forceinline__ cx cmul(cx a, cx w) {
    cx t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
// a * conj(w)
__device__ __forceinline__ cx cmulc(cx a, cx w) {
    cx t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
// acc + a * w
__device__ __forceinline__ cx cmac(cx acc, cx a, cx w) {
    cx t, r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(t) : "v"(a), "v"(w), "v"(acc));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}

// W_32^j = exp(-2*pi*i*j/32), j = 0..15 (cos, sin magnitudes)
__device__ constexpr float kCos32[16] = {
    1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
    0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f,
    0.0f, -0.19509032201612826785f, -0.38268343236508977173f, -0.55557023301960222474f,
    -0.70710678118654752440f, -0.83146961230254523708f, -0.92387953251128675613f, -0.98078528040323044913f};
__device__ constexpr float kSin32[16] = {
    0.0f, 0.19509032201612826785f, 0.38268343236508977173f, 0.55557023301960222474f,
    0.70710678118654752440f, 0.83146961230254523708f, 0.92387953251128675613f, 0.98078528040323044913f,
    1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
    0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f};

// d * W_32^(+-idx32), idx32 in [0, 32): forward uses exp(-i..), INV uses exp(+i..).  idx32 is a compile-time
// constant at every call site, so the factors become scalar-register constants of two packed instructions.
template <bool INV>
__device__ __forceinline__ cx tw32(cx d, int idx32) {
    if (idx32 == 0) return d;
    if (idx32 == 16) return d * cx{-1.0f, -1.0f};
    if (idx32 == 8) return INV ? mul_pos_i(d) : mul_neg_i(d);
    if (idx32 == 24) return INV ? mul_neg_i(d) : mul_pos_i(d);
    const float sg = idx32 >= 16 ? -1.0f : 1.0f;
    const float c = sg * kCos32[idx32 & 15], s = sg * (INV ? kSin32[idx32 & 15] : -kSin32[idx32 & 15]);  // w = c + i*s
    return d * cx{c, c} + cswap(d) * cx{-s, s};
}
// (a - b) * W_32^(+-idx32), idx32 in [0, 16)
template <bool INV>
__device__ __forceinline__ cx tw32_sub(cx a, cx b, int idx32) {
    if (idx32 == 8) return INV ? sub_mul_pos_i(a, b) : sub_mul_neg_i(a, b);
    return tw32<INV>(a - b, idx32);
}

constexpr __host__ __device__ int brev(int v, int bits) {
    int r = 0;
    for (int i = 0; i < bits; ++i) r |= ((v >> i) & 1) << (bits - 1 - i);
    return r;
}

// In-register radix-2 DIF DFT of N (16 or 32) points; result for frequency k is at v[brev(k)].
template <int N, bool INV>
__device__ __forceinline__ void dif(cx (&v)[N]) {
#pragma unroll
    for (int len = N; len >= 2; len >>= 1) {
        const int half = len >> 1;
#pragma unroll
        for (int base = 0; base < N; base += len) {
#pragma unroll
            for (int j = 0; j < half; ++j) {
                const cx a = v[base + j], b = v[base + j + half];
                v[base + j] = cadd(a, b);
                v[base + j + half] = tw32_sub<INV>(a, b, j * (32 / len));
            }
        }
    }
}

// Per-thread twiddles, kept two-level to save VGPRs: W^(t*k1) = lo[k1 & 3] * hi[k1 >> 2]
// (one extra complex multiply per use, one extra rounding ~6e-8).
struct TileTw {
    cx lo1[4], hi1[8];  // W_8192^(t*i), W_8192^(t*4i)
    cx lo2[4], hi2[4];  // W_256^(d*i),  W_256^(d*4i), d = t & 15

    __device__ __forceinline__ cx fwd1(cx e, int k1) const { return apply<false>(e, lo1[k1 & 3], hi1[k1 >> 2], k1 & 3, k1 >> 2); }
    __device__ __forceinline__ cx inv1(cx e, int k1) const { return apply<true>(e, lo1[k1 & 3], hi1[k1 >> 2], k1 & 3, k1 >> 2); }
    __device__ __forceinline__ cx fwd2(cx e, int k2) const { return apply<false>(e, lo2[k2 & 3], hi2[k2 >> 2], k2 & 3, k2 >> 2); }
    __device__ __forceinline__ cx inv2(cx e, int k2) const { return apply<true>(e, lo2[k2 & 3], hi2[k2 >> 2], k2 & 3, k2 >> 2); }
    __device__ __forceinline__ cx base() const { return lo1[1]; }  // W_8192^t

    template <bool CONJ>
    static __device__ __forceinline__ cx apply(cx e, cx lo, cx hi, int il, int ih) {
        if (il == 0 && ih == 0) return e;
        const cx w = il == 0 ? hi : (ih == 0 ? lo : cmul(lo, hi));
        return CONJ ? cmulc(e, w) : cmul(e, w);
    }
};

__device__ __forceinline__ float2 unit_root(int num, float inv_half_den) {
    // exp(-2*pi*i*num/den), 2/den = inv_half_den (a power of two, so the argument is exact)
    float s, c;
    sincospif((float)num * inv_half_den, &s, &c);
    return make_float2(c, -s);
}

// Twiddle table: TW_ROWS x 256 float2, row-major [row][t]; rows 0-3 lo1, 4-11 hi1, 12-15 lo2, 16-19 hi2 -- followed by the
// same values as TW_ROWS / 2 x 256 float4 (rows 2p and 2p+1 side by side), which is what tile_twiddles fetches: ten 16-byte
// loads per thread instead of twenty 8-byte ones (a vector-memory instruction costs the CU's address unit ~22 cycles
// whatever its width; MI355X_MICROARCH.md).  Filled once per device by tile_twiddle_table_kernel in common.hip (double
// precision, rounded once).
constexpr int TW_ROWS = 20;
constexpr int TW_TABLE_F2 = 2 * TW_ROWS * 256;   // float2 units, both copies

__device__ __forceinline__ void tile_twiddles(TileTw& tw, const float2* __restrict__ table, int t) {
    // buffer loads: descriptor in SGPRs, one lane offset, row stride as the scalar offset
    const uint64_t p = reinterpret_cast<uint64_t>(table + TW_ROWS * 256);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p), hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, TW_ROWS * 256 * 8, 0x00020000);
    auto pair = [&](int i) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, 16u * (uint32_t)t, (uint32_t)(i * 256 * 16), 0);
        return __builtin_bit_cast(f4v, v);
    };
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const f4v q = pair(i);
        tw.lo1[2 * i] = q.lo;
        tw.lo1[2 * i + 1] = q.hi;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f4v q = pair(2 + i);
        tw.hi1[2 * i] = q.lo;
        tw.hi1[2 * i + 1] = q.hi;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const f4v q = pair(6 + i);
        tw.lo2[2 * i] = q.lo;
        tw.lo2[2 * i + 1] = q.hi;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const f4v q = pair(8 + i);
        tw.hi2[2 * i] = q.lo;
        tw.hi2[2 * i + 1] = q.hi;
    }
}

// Host side: one table per device, created on first use (the only allocation the library makes;
// 80 KB, lives for the process).  Returns nullptr on failure.
const float2* tile_twiddle_table(hipStream_t stream);

__device__ __forceinline__ int s1_at(int k1, int b) { return k1 * S1_ROW + b; }
__device__ __forceinline__ int s2_row(int k2, int k1) { return (k2 * 32 + k1) * S2_ROW; }

// Butterfly ownership: thread t holds j = t and 512 - t; thread 0 holds the self-mirrored j = 0 and 256.
__device__ __forceinline__ int bf_a(int t) { return t; }
__device__ __forceinline__ int bf_b(int t) { return t == 0 ? 256 : 512 - t; }

// Forward: v[a] = z[t + 256*a] (natural order)  ->  w[bf][brev4(k3)] = Z[j_bf + 512*k3].
// Uses 3 barriers; on return other threads may still be reading S2.
__device__ __forceinline__ void tile_forward(cx (&v)[32], cx (&w)[2][16], const TileTw& tw, cx* lds, int t) {
    dif<32, false>(v);
#pragma unroll
    for (int r = 0; r < 32; ++r) {
        const int k1 = brev(r, 5);
        lds[s1_at(k1, t)] = tw.fwd1(v[r], k1);
    }
    __syncthreads();
    const int kk = t >> 4, d = t & 15;
    cx u[2][16];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int c = 0; c < 16; ++c) u[s][c] = lds[s1_at(kk + 16 * s, 16 * c + d)];
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        dif<16, false>(u[s]);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k2 = brev(r, 4);
            lds[s2_row(k2, kk + 16 * s) + d] = tw.fwd2(u[s][r], k2);
        }
    }
    __syncthreads();
#pragma unroll
    for (int bf = 0; bf < 2; ++bf) {
        const int j = bf ? bf_b(t) : bf_a(t);
        const f4v* row = reinterpret_cast<const f4v*>(lds + s2_row(j >> 5, j & 31));
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f4v p = row[q];
            w[bf][2 * q] = p.lo;
            w[bf][2 * q + 1] = p.hi;
        }
        dif<16, false>(w[bf]);
    }
}

// Inverse (unnormalised): w[bf][brev4(k3)] = Z'[j_bf + 512*k3] (the layout tile_forward leaves)
//   ->  v[brev5(a)] = z'[t + 256*a].
// The caller must have a barrier between the last S2 read of tile_forward and this call.
__device__ __forceinline__ void tile_inverse(cx (&w)[2][16], cx (&v)[32], const TileTw& tw, cx* lds, int t) {
#pragma unroll
    for (int bf = 0; bf < 2; ++bf) {
        cx p[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) p[k] = w[bf][brev(k, 4)];  // register renaming only
        dif<16, true>(p);
        const int j = bf ? bf_b(t) : bf_a(t);
        f4v* row = reinterpret_cast<f4v*>(lds + s2_row(j >> 5, j & 31));
#pragma unroll
        for (int q = 0; q < 8; ++q) row[q] = __builtin_shufflevector(p[brev(2 * q, 4)], p[brev(2 * q + 1, 4)], 0, 1, 2, 3);
    }
    __syncthreads();
    const int kk = t >> 4, d = t & 15;
    cx u[2][16];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) {
            u[s][k2] = tw.inv2(lds[s2_row(k2, kk + 16 * s) + d], k2);
        }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        dif<16, true>(u[s]);
#pragma unroll
        for (int r = 0; r < 16; ++r) lds[s1_at(kk + 16 * s, 16 * brev(r, 4) + d)] = u[s][r];
    }
    __syncthreads();
#pragma unroll
    for (int k1 = 0; k1 < 32; ++k1) {
        v[k1] = tw.inv1(lds[s1_at(k1, t)], k1);
    }
    dif<32, true>(v);
}

// ---- real-signal (polyphase) product on mirrored bin pairs -------------------------------------
// Z packs a real tile as z[m] = x[2m] + i*x[2m+1].  With A = Z[k], B = conj(Z[M-k]):
//   Xe = A + B, Xo = -i (A - B)   (the halves are folded into the stored filter)
//   Ye = He*Xe + W_M^k*Ho*Xo,  Yo = Ho*Xe + He*Xo
//   Z'[k] = Ye + i*Yo,  Z'[M-k] = conj(Ye - i*Yo)
// where (He, Ho) are the spectra of the even / odd filter taps scaled by 1/(4M).
__device__ __forceinline__ void pair_split(cx za, cx zb, cx& xe, cx& xo) {
    xe = add_conj(za, zb);
    xo = sub_conj_mul_neg_i(za, zb);
}
__device__ __forceinline__ void pair_product(cx xe, cx xo, f4v h, cx wk, cx& ye, cx& yo) {
    const cx he = h.lo, ho = h.hi;
    const cx who = cmul(wk, ho);
    ye = cmac(cmul(he, xe), who, xo);
    yo = cmac(cmul(ho, xe), he, xo);
}
// (ye, yo) += the product (partitioned convolution accumulates over partitions)
__device__ __forceinline__ void pair_product_acc(cx xe, cx xo, f4v h, cx wk, cx& ye, cx& yo) {
    const cx he = h.lo, ho = h.hi;
    const cx who = cmul(wk, ho);
    ye = cmac(cmac(ye, he, xe), who, xo);
    yo = cmac(cmac(yo, ho, xe), he, xo);
}
__device__ __forceinline__ void pair_merge(cx ye, cx yo, cx& za, cx& zb) {
    za = add_mul_pos_i(ye, yo);
    zb = conj_sub_mul_pos_i(ye, yo);
}

// a * W_16^k3 (forward sign), k3 a compile-time constant
__device__ __forceinline__ cx mul_w16(cx a, int k3) { return tw32<false>(a, 2 * k3); }

// Visit every mirrored pair held by thread t.  fn(slot, ia, ib, wk, self): ia/ib index into the
// flattened [2][16] natural-order arrays (bf*16 + k3); wk = W_M^k for k = bin of ia.
template <typename Fn>
__device__ __forceinline__ void for_each_pair(int t, cx wj, Fn&& fn) {
    if (t != 0) {
#pragma unroll
        for (int k3 = 0; k3 < 16; ++k3) fn(k3, k3, 16 + (15 - k3), mul_w16(wj, k3), false);
    } else {
        const cx one = {1.0f, 0.0f};
        fn(0, 0, 0, one, true);               // k = 0
        fn(8, 8, 8, mul_w16(one, 8), true);   // k = M/2
#pragma unroll
        for (int k3 = 1; k3 < 8; ++k3) fn(k3, k3, 16 - k3, mul_w16(one, k3), false);
#pragma unroll
        for (int k3 = 0; k3 < 8; ++k3) fn(9 + k3, 16 + k3, 16 + (15 - k3), tw32<false>(one, 1 + 2 * k3), false);  // W_8192^256 = W_32^1
    }
}

}  // namespace gfx
