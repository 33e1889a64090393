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

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) {  // a * conj(b)
    return make_float2(fmaf(a.x, b.x, a.y * b.y), fmaf(a.y, b.x, -a.x * b.y));
}
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }
__device__ __forceinline__ float2 mul_neg_i(float2 a) { return make_float2(a.y, -a.x); }  // a * (-i)
__device__ __forceinline__ float2 mul_pos_i(float2 a) { return make_float2(-a.y, a.x); }  // a * (+i)

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

// d * W_32^(+-idx32): forward uses exp(-i..), INV uses exp(+i..)
template <bool INV>
__device__ __forceinline__ float2 tw32(float2 d, int idx32) {
    if (idx32 == 0) return d;
    if (idx32 == 8) return INV ? mul_pos_i(d) : mul_neg_i(d);
    const float c = kCos32[idx32], s = INV ? kSin32[idx32] : -kSin32[idx32];  // w = c + i*s
    return make_float2(fmaf(d.x, c, -d.y * s), fmaf(d.x, s, d.y * c));
}

constexpr __host__ __device__ int brev(int v, int bits) {
    int r = 0;
    for (int i = 0; i < bits; ++i) r |= ((v >> i) & 1) << (bits - 1 - i);
    return r;
}

// In-register radix-2 DIF DFT of N (16 or 32) points; result for frequency k is at v[brev(k)].
template <int N, bool INV>
__device__ __forceinline__ void dif(float2 (&v)[N]) {
#pragma unroll
    for (int len = N; len >= 2; len >>= 1) {
        const int half = len >> 1;
#pragma unroll
        for (int base = 0; base < N; base += len) {
#pragma unroll
            for (int j = 0; j < half; ++j) {
                const float2 a = v[base + j], b = v[base + j + half];
                v[base + j] = cadd(a, b);
                v[base + j + half] = tw32<INV>(csub(a, b), j * (32 / len));
            }
        }
    }
}

// Per-thread twiddles, kept two-level to save VGPRs: W^(t*k1) = lo[k1 & 3] * hi[k1 >> 2]
// (one extra complex multiply per use, one extra rounding ~6e-8).
struct TileTw {
    float2 lo1[4], hi1[8];  // W_8192^(t*i), W_8192^(t*4i)
    float2 lo2[4], hi2[4];  // W_256^(d*i),  W_256^(d*4i), d = t & 15

    __device__ __forceinline__ float2 fwd1(float2 e, int k1) const { return apply<false>(e, lo1[k1 & 3], hi1[k1 >> 2], k1 & 3, k1 >> 2); }
    __device__ __forceinline__ float2 inv1(float2 e, int k1) const { return apply<true>(e, lo1[k1 & 3], hi1[k1 >> 2], k1 & 3, k1 >> 2); }
    __device__ __forceinline__ float2 fwd2(float2 e, int k2) const { return apply<false>(e, lo2[k2 & 3], hi2[k2 >> 2], k2 & 3, k2 >> 2); }
    __device__ __forceinline__ float2 inv2(float2 e, int k2) const { return apply<true>(e, lo2[k2 & 3], hi2[k2 >> 2], k2 & 3, k2 >> 2); }
    __device__ __forceinline__ float2 base() const { return lo1[1]; }  // W_8192^t

    template <bool CONJ>
    static __device__ __forceinline__ float2 apply(float2 e, float2 lo, float2 hi, int il, int ih) {
        if (il == 0 && ih == 0) return e;
        const float2 w = il == 0 ? hi : (ih == 0 ? lo : cmul(lo, hi));
        return CONJ ? cmulc(e, w) : cmul(e, w);
    }
};

__device__ __forceinline__ float2 unit_root(int num, float inv_half_den) {
    // exp(-2*pi*i*num/den), 2/den = inv_half_den (a power of two, so the argument is exact)
    float s, c;
    sincospif((float)num * inv_half_den, &s, &c);
    return make_float2(c, -s);
}

// Twiddle table: TW_ROWS x 256 float2, row-major [row][t]; rows 0-3 lo1, 4-11 hi1, 12-15 lo2, 16-19 hi2.
// Filled once per device by tile_twiddle_table_kernel in common.hip (double precision, rounded once); every tile
// then fetches its 20 values with coalesced 8-byte loads instead of 20 sincos evaluations.
constexpr int TW_ROWS = 20;

__device__ __forceinline__ void tile_twiddles(TileTw& tw, const float2* __restrict__ table, int t) {
    // buffer loads: descriptor in SGPRs, one lane offset, row stride as the scalar offset
    const uint64_t p = reinterpret_cast<uint64_t>(table);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p), hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, TW_ROWS * TILE_T * 8, 0x00020000);
    auto row = [&](int i) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b64(r, 8u * (uint32_t)t, (uint32_t)(i * TILE_T * 8), 0);
        return make_float2(__uint_as_float(v[0]), __uint_as_float(v[1]));
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) tw.lo1[i] = row(i);
#pragma unroll
    for (int i = 0; i < 8; ++i) tw.hi1[i] = row(4 + i);
#pragma unroll
    for (int i = 0; i < 4; ++i) tw.lo2[i] = row(12 + i);
#pragma unroll
    for (int i = 0; i < 4; ++i) tw.hi2[i] = row(16 + i);
}

// Host side: one table per device, created on first use (the only allocation the library makes;
// 40 KB, lives for the process).  Returns nullptr on failure.
const float2* tile_twiddle_table(hipStream_t stream);

__device__ __forceinline__ int s1_at(int k1, int b) { return k1 * S1_ROW + b; }
__device__ __forceinline__ int s2_row(int k2, int k1) { return (k2 * 32 + k1) * S2_ROW; }

// Butterfly ownership: thread t holds j = t and 512 - t; thread 0 holds the self-mirrored j = 0 and 256.
__device__ __forceinline__ int bf_a(int t) { return t; }
__device__ __forceinline__ int bf_b(int t) { return t == 0 ? 256 : 512 - t; }

// Forward: v[a] = z[t + 256*a] (natural order)  ->  w[bf][brev4(k3)] = Z[j_bf + 512*k3].
// Uses 3 barriers; on return other threads may still be reading S2.
__device__ __forceinline__ void tile_forward(float2 (&v)[32], float2 (&w)[2][16], const TileTw& tw, float2* lds, int t) {
    dif<32, false>(v);
#pragma unroll
    for (int r = 0; r < 32; ++r) {
        const int k1 = brev(r, 5);
        lds[s1_at(k1, t)] = tw.fwd1(v[r], k1);
    }
    __syncthreads();
    const int kk = t >> 4, d = t & 15;
    float2 u[2][16];
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
        const float4* row = reinterpret_cast<const float4*>(lds + s2_row(j >> 5, j & 31));
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 p = row[q];
            w[bf][2 * q] = make_float2(p.x, p.y);
            w[bf][2 * q + 1] = make_float2(p.z, p.w);
        }
        dif<16, false>(w[bf]);
    }
}

// Inverse (unnormalised): w[bf][brev4(k3)] = Z'[j_bf + 512*k3] (the layout tile_forward leaves)
//   ->  v[brev5(a)] = z'[t + 256*a].
// The caller must have a barrier between the last S2 read of tile_forward and this call.
__device__ __forceinline__ void tile_inverse(float2 (&w)[2][16], float2 (&v)[32], const TileTw& tw, float2* lds, int t) {
#pragma unroll
    for (int bf = 0; bf < 2; ++bf) {
        float2 p[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) p[k] = w[bf][brev(k, 4)];  // register renaming only
        dif<16, true>(p);
        const int j = bf ? bf_b(t) : bf_a(t);
        float4* row = reinterpret_cast<float4*>(lds + s2_row(j >> 5, j & 31));
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float2 e0 = p[brev(2 * q, 4)], e1 = p[brev(2 * q + 1, 4)];
            row[q] = make_float4(e0.x, e0.y, e1.x, e1.y);
        }
    }
    __syncthreads();
    const int kk = t >> 4, d = t & 15;
    float2 u[2][16];
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
__device__ __forceinline__ void pair_split(float2 za, float2 zb, float2& xe, float2& xo) {
    const float2 b = cconj(zb);
    xe = cadd(za, b);
    xo = mul_neg_i(csub(za, b));
}
__device__ __forceinline__ void pair_product(float2 xe, float2 xo, float4 h, float2 wk, float2& ye, float2& yo) {
    const float2 he = make_float2(h.x, h.y), ho = make_float2(h.z, h.w);
    const float2 who = cmul(wk, ho);
    ye = cadd(cmul(he, xe), cmul(who, xo));
    yo = cadd(cmul(ho, xe), cmul(he, xo));
}
__device__ __forceinline__ void pair_merge(float2 ye, float2 yo, float2& za, float2& zb) {
    const float2 iyo = mul_pos_i(yo);
    za = cadd(ye, iyo);
    zb = cconj(csub(ye, iyo));
}

// W_16^k3 (forward sign) as a compile-time-foldable constant
__device__ __forceinline__ float2 w16(int k3) {
    const float2 w = tw32<false>(make_float2(1.0f, 0.0f), 2 * (k3 & 7));
    return (k3 & 8) ? make_float2(-w.x, -w.y) : w;
}

// Visit every mirrored pair held by thread t.  fn(slot, ia, ib, wk, self): ia/ib index into the
// flattened [2][16] natural-order arrays (bf*16 + k3); wk = W_M^k for k = bin of ia.
template <typename Fn>
__device__ __forceinline__ void for_each_pair(int t, float2 wj, Fn&& fn) {
    if (t != 0) {
#pragma unroll
        for (int k3 = 0; k3 < 16; ++k3) fn(k3, k3, 16 + (15 - k3), cmul(wj, w16(k3)), false);
    } else {
        fn(0, 0, 0, make_float2(1.0f, 0.0f), true);    // k = 0
        fn(8, 8, 8, make_float2(-1.0f, 0.0f), true);   // k = M/2
#pragma unroll
        for (int k3 = 1; k3 < 8; ++k3) fn(k3, k3, 16 - k3, w16(k3), false);
        const float2 w256 = tw32<false>(make_float2(1.0f, 0.0f), 1);  // W_8192^256 = W_32^1
#pragma unroll
        for (int k3 = 0; k3 < 8; ++k3) fn(9 + k3, 16 + k3, 16 + (15 - k3), cmul(w256, w16(k3)), false);
    }
}

}  // namespace gfx
