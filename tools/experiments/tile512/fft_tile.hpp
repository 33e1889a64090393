// LDS FFT tile for gfx950: one 8192-point complex FFT (= one 16384-sample real tile, two samples
// packed per complex point) per 512-thread workgroup, 16 points per thread.
//
//   8192 = 32 x 16 x 16.  Three radix passes run in registers and exchange through LDS twice per
//   direction; the radix-32 pass is split over lane pairs (t, t^1): a radix-2 step across the pair
//   (one DPP quad-perm move per value) followed by a radix-16 codelet in each lane.  16 points per
//   thread keeps the kernel under 128 VGPRs -> 4 waves per SIMD (2 workgroups x 8 waves per CU),
//   twice the latency hiding of a 32-points-per-thread layout.
//
//   Forward is decimation-in-frequency, the inverse is its mirror: no digit reversal is ever
//   materialised.  The spectrum lives in a private "thread layout": thread t holds the radix-16
//   butterfly j(t) (bins k = j + 512*k3, k3 at register brev4(k3)) and the mirror butterfly
//   512 - j sits in lane t^1, so Z[M-k] — what the product of real-signal spectra needs — is one
//   DPP move away (lanes 0 and 1 hold the two self-mirrored butterflies j = 0 and 256).
//
//   Roles of thread t:   pass 1: column b = t>>1, half h = t&1   (points a = a' + 16h, a' = 0..15)
//                        pass 2: k1 = t>>4, d = t&15
//                        pass 3: LDS row t = rho(j)
//   LDS images (float2 units), padded so the access patterns are bank-conflict free:
//     S1[k1][b]   at k1*272 + b          (32 rows of 256 + 16 pad)
//     S2[row][d]  at row*18 + d          (512 rows of 16 + 2 pad), row = rho(k1 + 32*k2)
//   S1 and S2 alias one 73,728-byte buffer -> 2 workgroups per CU.
//
// tools/fft_tile512_model.py is the numpy model of exactly this index math.
#pragma once
#include <hip/hip_runtime.h>

namespace gfx {

constexpr int TILE_M = 8192;          // complex points per tile
constexpr int TILE_F = 16384;         // real samples per tile
constexpr int TILE_T = 512;           // threads per workgroup
constexpr int TILE_E = 16;            // complex points per thread
constexpr int S1_ROW = 272;           // padded row of S1 (float2 units)
constexpr int S2_ROW = 18;            // padded row of S2 (float2 units)
constexpr int TILE_LDS_F2 = 512 * S2_ROW;              // 9216 float2
constexpr int TILE_LDS_BYTES = TILE_LDS_F2 * 8;        // 73,728 B
constexpr int H_TILE_F4 = TILE_E * TILE_T;             // float4 {alpha, beta} per (row-channel, partition)

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

// value of the same register in lane t^1 (DPP quad_perm [1,0,3,2]; no LDS traffic)
__device__ __forceinline__ float lane_xor1(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float2 lane_xor1(float2 v) { return make_float2(lane_xor1(v.x), lane_xor1(v.y)); }

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

// In-register radix-2 DIF DFT of N = 16 points; result for frequency k is at v[brev(k)].
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

// Per-thread twiddles, two-level: W^(b(2m+h)) = lo1[m & 3] * hi1[m >> 2], W_256^(d*k2) = lo2[k2 & 3] * hi2[k2 >> 2].
// Fetched from the per-device table (TW_ROWS x 512 float2, double-precision evaluated, common.hip):
//   rows 0-3  W_8192^(b(h+2i))   rows 4-7  W_8192^(8 b i)   rows 8-11 W_256^(d i)   rows 12-15 W_256^(4 d i)
//   row 16    W_8192^(j(t))      (b = t>>1, h = t&1, d = t&15)
constexpr int TW_ROWS = 17;

// pass-1 / pass-2 twiddle sets, loaded right where they are used so they never stay live across the
// spectral product (keeps the kernels under 128 VGPRs without spills)
struct Tw4x4 {
    float2 lo[4], hi[4];
    template <bool CONJ>
    __device__ __forceinline__ float2 apply(float2 e, int i, bool lo_is_one) const {
        const bool hi_is_one = (i >> 2) == 0;
        if (lo_is_one && hi_is_one) return e;
        const float2 w = lo_is_one ? hi[i >> 2] : (hi_is_one ? lo[i & 3] : cmul(lo[i & 3], hi[i >> 2]));
        return CONJ ? cmulc(e, w) : cmul(e, w);
    }
};
__device__ __forceinline__ void load_tw(Tw4x4& tw, const float2* __restrict__ table, int first_row, int t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) tw.lo[i] = table[(first_row + i) * TILE_T + t];
#pragma unroll
    for (int i = 0; i < 4; ++i) tw.hi[i] = table[(first_row + 4 + i) * TILE_T + t];
}
__device__ __forceinline__ float2 tile_wj(const float2* __restrict__ table, int t) { return table[16 * TILE_T + t]; }

// Host side: one table per device, created on first use (the only allocation the library makes;
// 70 KB, lives for the process).  Returns nullptr on failure.
const float2* tile_twiddle_table(hipStream_t stream);

__host__ __device__ __forceinline__ int tile_j_of(int t) {
    return t == 0 ? 0 : (t == 1 ? 256 : ((t & 1) ? 512 - (t >> 1) : (t >> 1)));
}
__device__ __forceinline__ int tile_rho(int j) { return j == 0 ? 0 : (j == 256 ? 1 : (j < 256 ? 2 * j : 2 * (512 - j) + 1)); }
__device__ __forceinline__ int s1_at(int k1, int b) { return k1 * S1_ROW + b; }

// complex-point index inside the tile of this thread's a'-th value: 256*(a' + 16h) + b
__device__ __forceinline__ int tile_point(int t, int a) { return 256 * (a + 16 * (t & 1)) + (t >> 1); }

// Forward: p[a'] = z[tile_point(t, a')]  ->  w[brev4(k3)] = Z[j(t) + 512*k3].
// 3 barriers; on return other threads may still be reading S2.
// Split in two so callers can issue their spectrum loads between the halves (latency hiding).
__device__ __forceinline__ void tile_forward_a(float2 (&p)[16], const float2* __restrict__ table, float2* lds, int t) {
    const int b = t >> 1;
    const bool odd = t & 1;
    Tw4x4 tw1;  // W_8192^(b(2m+h)); lo[0] = W^(b h) is not 1 in odd lanes.  Issued first: lands under the radix work
    load_tw(tw1, table, 0, t);
    // radix-2 across the lane pair: even lane keeps z[a'] + z[a'+16], odd lane (z[a'] - z[a'+16]) * W_32^a'
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        const float2 q = lane_xor1(p[a]);
        const float2 e = odd ? csub(q, p[a]) : cadd(p[a], q);
        const float c = odd ? kCos32[a] : 1.0f, s = odd ? -kSin32[a] : 0.0f;
        p[a] = (a == 0) ? e : make_float2(fmaf(e.x, c, -e.y * s), fmaf(e.x, s, e.y * c));
    }
    dif<16, false>(p);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = brev(r, 4);
        lds[s1_at(2 * m + (t & 1), b)] = tw1.apply<false>(p[r], m, false);
    }
    Tw4x4 tw2;  // W_256^(d k2): in flight across the barrier, the LDS reads and the next codelet
    load_tw(tw2, table, 8, t);
    __syncthreads();
    const int k1 = t >> 4, d = t & 15;
    float2 u[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) u[c] = lds[s1_at(k1, 16 * c + d)];
    __syncthreads();
    dif<16, false>(u);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int k2 = brev(r, 4);
        lds[tile_rho(k1 + 32 * k2) * S2_ROW + d] = tw2.apply<false>(u[r], k2, (k2 & 3) == 0);
    }
    __syncthreads();
}
__device__ __forceinline__ void tile_forward_b(float2 (&w)[16], const float2* lds, int t) {
    const float4* row = reinterpret_cast<const float4*>(lds + t * S2_ROW);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float4 v = row[q];
        w[2 * q] = make_float2(v.x, v.y);
        w[2 * q + 1] = make_float2(v.z, v.w);
    }
    dif<16, false>(w);
}
__device__ __forceinline__ void tile_forward(float2 (&p)[16], float2 (&w)[16], const float2* __restrict__ table,
                                             float2* lds, int t) {
    tile_forward_a(p, table, lds, t);
    tile_forward_b(w, lds, t);
}

// q[r] = Z[M - k] for the bin k held in w[r]: register 15-r of lane t^1 (lanes 0/1: own registers).
__device__ __forceinline__ void tile_mirror(const float2 (&w)[16], float2 (&q)[16], int t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) q[r] = lane_xor1(w[15 - r]);
    if (t < 2) {  // wave 0 only
#pragma unroll
        for (int r = 0; r < 16; ++r) q[r] = (t == 0) ? w[brev((16 - brev(r, 4)) & 15, 4)] : w[15 - r];
    }
}

// Inverse (unnormalised): w[brev4(k3)] = Z'[j(t) + 512*k3]  ->  v[a'] = z'[tile_point(t, a')].
// The caller must have a barrier between the last S2 read of tile_forward and this call.
__device__ __forceinline__ void tile_inverse(float2 (&w)[16], float2 (&v)[16], const float2* __restrict__ table,
                                             float2* lds, int t) {
    Tw4x4 tw2;
    load_tw(tw2, table, 8, t);
    {
        float2 p[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) p[k] = w[brev(k, 4)];  // register renaming only
        dif<16, true>(p);
        float4* row = reinterpret_cast<float4*>(lds + t * S2_ROW);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float2 e0 = p[brev(2 * q, 4)], e1 = p[brev(2 * q + 1, 4)];
            row[q] = make_float4(e0.x, e0.y, e1.x, e1.y);
        }
    }
    __syncthreads();
    const int k1 = t >> 4, d = t & 15;
    float2 u[16];
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2)
        u[k2] = tw2.apply<true>(lds[tile_rho(k1 + 32 * k2) * S2_ROW + d], k2, (k2 & 3) == 0);
    Tw4x4 tw1;
    load_tw(tw1, table, 0, t);
    __syncthreads();
    dif<16, true>(u);
#pragma unroll
    for (int r = 0; r < 16; ++r) lds[s1_at(k1, 16 * brev(r, 4) + d)] = u[r];
    __syncthreads();
    const int b = t >> 1;
    const bool odd = t & 1;
    float2 g[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) g[m] = tw1.apply<true>(lds[s1_at(2 * m + (t & 1), b)], m, false);
    dif<16, true>(g);
    // even lane holds E[a'], odd lane O[a']: z[a'] = E + conj(W_32^a') O, z[a'+16] = E - conj(W_32^a') O
#pragma unroll
    for (int a = 0; a < 16; ++a) {
        float2 e = g[brev(a, 4)];
        if (a != 0) {
            const float c = odd ? kCos32[a] : 1.0f, s = odd ? kSin32[a] : 0.0f;
            e = make_float2(fmaf(e.x, c, -e.y * s), fmaf(e.x, s, e.y * c));
        }
        const float2 q = lane_xor1(e);
        v[a] = odd ? csub(q, e) : cadd(e, q);
    }
}

// W_16^k3 (forward sign)
__device__ __forceinline__ float2 w16(int k3) {
    const float2 w = tw32<false>(make_float2(1.0f, 0.0f), 2 * (k3 & 7));
    return (k3 & 8) ? make_float2(-w.x, -w.y) : w;
}

// Product of packed real-signal spectra in the thread layout.  With A = Z[k], Bc = conj(Z[M-k]):
//   Z'[k] = alpha_k * A + beta_k * Bc,
//   alpha = (2 He + i (1 - W_M^k) Ho) / (2M),  beta = i (1 + W_M^k) Ho / (2M),
// He / Ho = spectra of the even / odd filter taps (polyphase form of the real convolution).
__device__ __forceinline__ float2 spectral_product(float2 a, float2 qmirror, float4 ab) {
    const float2 al = make_float2(ab.x, ab.y), be = make_float2(ab.z, ab.w);
    return cadd(cmul(al, a), cmulc(be, qmirror));
}

}  // namespace gfx
