// Overlap-save FIR convolution on LDS FFT tiles (gfx950).
//
// Replaces the reference's convolve() — core/convolution.py:119-134 — for the case where
// it equals a true linear convolution (P = L + N - 1 even; see DESIGN.md for the odd-P quirk,
// which is layered on top of the *full* convolution this file produces).
//
// Kernels
//   hspec_kernel    taps -> per-partition tile spectra (He, Ho pairs in thread layout)
//   fftconv1_kernel N <= 8193: load x tile -> FFT -> x H -> IFFT -> store valid samples (fused)
//   xspec_kernel    N  > 8193: x window -> FFT -> spectra to HBM/L2
//   macinv_kernel   N  > 8193: sum_p X[i-p] * H[p] in registers -> IFFT -> store
//
// Algorithmic bytes: 4*(C_in + C_out) per output frame per row (read x once, write y once);
// the tile overlap (N-1 of 16384 samples) is re-read through L2.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/grafx_amd.h"
#include "fft_tile.hpp"

namespace gfx {

struct ConvGeom {
    int64_t nparts, part_len, O, V, ntiles, hop;
    bool ok;
};

// part_len = 0: the default geometry (one tile-sized filter up to 8193 taps, else 8192-tap partitions with windows
// hopping by 8192).  A longer partition (8192 < part_len <= 16384 - Lout, even) is legal when one output tile covers
// all Lout samples: the windows of the partitions then hop by part_len while the tile keeps 16384 - part_len valid
// samples -- fewer partitions and windows for "long filter, short output" problems (the filter gradient).
static inline ConvGeom conv_geom(int64_t N, int64_t Lout, int64_t part_len = 0) {
    ConvGeom g;
    g.ok = true;
    if (N <= TILE_M + 1) {
        g.nparts = 1;
        g.part_len = N;
        // overlap >= N-1, rounded up to whole 512-sample register rows: loads and stores of a tile then need no per-lane
        // masks (load_window / store_valid), ~15 % fewer vector-ALU instructions per tile for <1 % more tiles
        g.O = (N - 1 + 511) & ~int64_t(511);
        g.ok = part_len == 0;
    } else if (part_len == 0 || part_len == TILE_M) {
        g.part_len = TILE_M;
        g.nparts = (N + TILE_M - 1) / TILE_M;
        g.O = TILE_M;
    } else {
        g.ok = part_len > TILE_M && (part_len & 1) == 0 && part_len + Lout <= TILE_F;
        g.part_len = part_len;
        g.nparts = (N + part_len - 1) / part_len;
        g.O = part_len;
    }
    g.V = TILE_F - g.O;
    g.hop = g.nparts == 1 ? g.V : g.part_len;
    g.ntiles = (Lout + g.V - 1) / g.V;
    if (g.hop != g.V && g.ntiles != 1) g.ok = false;
    return g;
}

struct ConvArgs {
    gfx_rowmap_t xmap, ymap, cmap;  // cmap: rows of the optional input copy (fftconv1 only)
    int64_t L, Lout, off;    // signal length, outputs per row, output offset into the full convolution
    int64_t O, V, hop;       // overlap and valid samples per tile; start-to-start distance of the partition windows
    int64_t ntiles, nblocks; // tiles per row-channel, total workgroups of real work
    int nparts;
    int Cin, Cf, Cout;
    unsigned hrows;          // filter rows: row r convolves with filter r % hrows (batch-major rows sharing filters)
};

// rows and row counts fit 32 bits (checked by the launchers): 32-bit division is ~5x cheaper than
// the 64-bit software divide and stays on the scalar unit.
__device__ __forceinline__ int64_t row_off(const gfx_rowmap_t& m, unsigned r, int c) {
    const unsigned inner = (unsigned)m.inner;
    const unsigned q = r / inner, rem = r - q * inner;
    return (int64_t)q * m.stride_outer + (int64_t)rem * m.stride_inner + (int64_t)c * m.stride_ch;
}

// workgroup b runs on XCD b % 8 (observed): give each XCD a contiguous run of logical
// indices so tiles of one row-channel (which share the filter spectrum) meet in one L2.
__device__ __forceinline__ unsigned xcd_logical_block() {
    const unsigned per_xcd = gridDim.x >> 3;
    return (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
}

// ---- raw buffer access (SRSRC descriptor in SGPRs + one 32-bit lane offset) --------------------------
// All per-tile traffic goes through buffer instructions: the row base and the per-`a` strides live in
// SGPRs, every lane carries ONE 32-bit byte offset, and the hardware range check supplies the zeros
// outside [0, L) / drops the stores past the row end — no 64-bit per-lane addresses, no bounds code.
using rsrc_t = __amdgpu_buffer_rsrc_t;

__device__ __forceinline__ rsrc_t make_rsrc(const void* base, int64_t bytes) {
    // descriptor inputs must be provably wave-uniform (else hipcc wraps every access in a waterfall loop)
    const uint64_t p = reinterpret_cast<uint64_t>(base);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p), hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
    const int64_t nb = bytes < 0 ? 0 : (bytes > 0x7fffffffLL ? 0x7fffffffLL : bytes);
    const uint32_t n = __builtin_amdgcn_readfirstlane((uint32_t)nb);
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, n, 0x00020000);
}
__device__ __forceinline__ cx buf_load_f2(rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(cx, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
__device__ __forceinline__ f4v buf_load_f4(rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
// outputs are streamed: written once, read by a later stage long after they left the 4 MB L2s.  The
// non-temporal hint keeps them from evicting the window overlaps and filter spectra the tiles re-read
// (tee kernel at 8192 rows: 7.3 -> 6.5 ms).
#ifndef GFX_STORE_AUX
#define GFX_STORE_AUX 2  // nt
#endif
__device__ __forceinline__ void buf_store_f2(rsrc_t r, uint32_t voff, uint32_t soff, cx e) {
    using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, e), r, voff, soff, GFX_STORE_AUX);
}
constexpr uint32_t OOB = 0xffffffffu;  // lane offset that the range check always rejects

// v[a] = (x[s + 2m], x[s + 2m + 1]), m = t + 256 a; x is zero outside [0, L).  `row` = start of the signal row.
// Straight-line in the common case: the two special cases (a window starting before the row, a pair straddling
// sample 0) sit behind one wave-uniform branch after all 32 loads have been issued.
template <bool FAST = true>
__device__ __forceinline__ void load_window(cx (&v)[32], const float* __restrict__ row, int64_t s, int64_t L,
                                            int t, float gain) {
    // descriptor starts at the window (possibly before the row for the first tile: those lanes are masked)
    const rsrc_t r = make_rsrc(row + s, (L - s) * 4);
    if (FAST && (s >= 0 || (s & 511) == 0)) {
        // No per-lane clipping (uniform test): the window starts inside the row, or a whole number of 512-sample register
        // rows before it (tile geometry with O a multiple of 512) -- every load is "lane offset 8 t" or skipped as a
        // row, and the range check supplies the zeros past the row end.  Saves ~100 compare / select instructions
        // per window on the vector ALU, which is what bounds these kernels.
        const int a_lo = s < 0 ? (int)((-s) >> 9) : 0;
#pragma unroll
        for (int a = 0; a < 32; ++a) {
            if (a < a_lo) v[a] = cx{0.0f, 0.0f};
            else v[a] = buf_load_f2(r, 8u * (uint32_t)t, 2048u * a);
        }
        if (gain != 1.0f) {
#pragma unroll
            for (int a = 0; a < 32; ++a) v[a] *= gain;
        }
        return;
    }
    const bool clip = s < 0;                                   // uniform; only a first tile can start before the row
    const int s32 = clip ? (int)s : 0;                         // |s| <= 16384 there
#pragma unroll
    for (int a = 0; a < 32; ++a) {
        const int n = s32 + 2 * (t + 256 * a);                 // sample index of the pair's first element (if clip)
        const uint32_t voff = (clip && n < 0) ? OOB : 8u * (uint32_t)t;
        v[a] = buf_load_f2(r, voff, 2048u * a);
    }
    if (clip && (s32 & 1)) {                                   // odd start: the pair (x[-1], x[0]) was masked as a whole
        const float x0 = row[0];
#pragma unroll
        for (int a = 0; a < 32; ++a)
            if (s32 + 2 * (t + 256 * a) == -1) v[a] = cx{0.0f, x0};
    }
    if (gain != 1.0f) {
#pragma unroll
        for (int a = 0; a < 32; ++a) v[a] *= gain;
    }
}

// Time-reversed window of a segment of `len` (<= 16384) samples: v[a] = (seg[len-1-2m], seg[len-2-2m]), m = t + 256 a,
// zero below the segment.  Lets the filter-gradient correlation use the signal as its own "filter" without a flipped
// copy of it.
__device__ __forceinline__ void load_window_rev(cx (&v)[32], const float* __restrict__ seg, int64_t len, int t) {
    const rsrc_t r = make_rsrc(seg, len * 4);
    const int base = (int)len - 2;
#pragma unroll
    for (int a = 0; a < 32; ++a) {
        const int e = base - 2 * (t + 256 * a);                // index of the pair's lower sample
        v[a] = cswap(buf_load_f2(r, e < 0 ? OOB : 4u * (uint32_t)e, 0));
    }
    if (len & 1) {                                             // uniform: the pair (seg[-1], seg[0]) was masked as a whole
        const float x0 = seg[0];
#pragma unroll
        for (int a = 0; a < 32; ++a)
            if (base - 2 * (t + 256 * a) == -1) v[a] = cx{x0, 0.0f};
    }
}

// y[n0 + (2m - O)] for 2m >= O, n < Lout;  v[brev5(a)] = (z'[2m], z'[2m+1])  (NATURAL: v[a] instead)
template <bool NATURAL = false>
__device__ __forceinline__ void store_valid(const cx (&v)[32], float* __restrict__ row, int64_t n0, int64_t O,
                                            int64_t Lout, int t) {
    // opaque copy of the lane index: the tee store (before the transforms) and the output store (after them) use
    // the same 32 lane masks, and without this the compiler keeps them alive across the whole kernel (16 spilled
    // dwords = 2.8 GB of scratch write-back per 8192-row launch) instead of recomputing 32 compares
    asm volatile("" : "+v"(t));
    const int64_t room = Lout - (n0 - O);                      // samples from the descriptor base to the row end
    const rsrc_t r = make_rsrc(row + (n0 - O), room * 4);
    if ((O & 511) == 0 && !((room & 1) && room < TILE_F)) {
        // Row-uniform form (uniform test): the overlap is a whole number of 512-sample register rows and no sample pair
        // straddles the row end, so a row is either skipped or stored with "lane offset 8 t" (pairs past the row end
        // are dropped by the range check): no per-lane masks.
        const int a_lo = (int)(O >> 9);
#pragma unroll
        for (int a = 0; a < 32; ++a)
            if (a >= a_lo) buf_store_f2(r, 8u * (uint32_t)t, 2048u * a, v[NATURAL ? a : brev(a, 5)]);
        return;
    }
    const int o32 = (int)O;
    const int tail = (room & 1) && room < TILE_F ? (int)room - 1 : -1;  // a pair straddling the row end starts here
#pragma unroll
    for (int a = 0; a < 32; ++a) {
        const int q = 2 * (t + 256 * a);
        const uint32_t voff = (q < o32 || q == tail) ? OOB : 8u * (uint32_t)t;
        buf_store_f2(r, voff, 2048u * a, v[NATURAL ? a : brev(a, 5)]);
    }
    if (tail >= o32) {                                         // uniform: single trailing sample of an odd-length row
        float last = 0.0f;
        bool mine = false;
#pragma unroll
        for (int a = 0; a < 32; ++a)
            if (2 * (t + 256 * a) == tail) {
                last = v[NATURAL ? a : brev(a, 5)].x;
                mine = true;
            }
        if (mine) row[n0 - O + tail] = last;
    }
}

}  // namespace gfx
#include "fftconv_wide.hpp"
namespace gfx {

#define NAT(arr, i) arr[(i) >> 4][brev((i) & 15, 4)]

// ------------------------------------------------------------------------------------------------
// REV: the taps of filter (r, c) are row (r, c) of `h` read backwards through `hmap` (tap k = h[r, c, N-1-k]).
template <bool REV>
__global__ __launch_bounds__(TILE_T, 2) void hspec_kernel(const float* __restrict__ h, const float* __restrict__ gain,
                                                          int64_t gain_div, float4* __restrict__ Hs, int64_t N,
                                                          int nparts, int64_t part_len, gfx_rowmap_t hmap, int C,
                                                          const float2* __restrict__ twtab) {
    extern __shared__ __attribute__((aligned(16))) cx lds[];
    const int t = threadIdx.x;
    const unsigned b = blockIdx.x;
    const unsigned rc = b / (unsigned)nparts;
    const int p = (int)(b - rc * (unsigned)nparts);
    const int64_t start = (int64_t)p * part_len;
    const int64_t len = min(part_len, N - start);

    TileTw tw;
    cx v[32], w[2][16];
    if (REV) {
        const unsigned r = rc / (unsigned)C;
        load_window_rev(v, h + row_off(hmap, r, (int)(rc - r * (unsigned)C)) + (N - start - len), len, t);
    } else {
        load_window(v, h + (int64_t)rc * N + start, 0, len, t, gain ? gain[rc / (unsigned)gain_div] : 1.0f);
    }
    tile_twiddles(tw, twtab, t);
    tile_forward(v, w, tw, lds, t);

    f4v* out = reinterpret_cast<f4v*>(Hs) + (int64_t)b * H_TILE_F4;
    const float sc = 1.0f / (4.0f * TILE_M);
    for_each_pair(t, tw.base(), [&](int slot, int ia, int ib, cx, bool) {
        cx he, ho;
        pair_split(NAT(w, ia), NAT(w, ib), he, ho);
        out[slot * TILE_T + t] = __builtin_shufflevector(he * sc, ho * sc, 0, 1, 2, 3);
    });
}

// ------------------------------------------------------------------------------------------------
template <bool TEE>
__global__ __launch_bounds__(TILE_T, 2) void fftconv1_kernel(const float* __restrict__ x, const float4* __restrict__ Hs,
                                                             float* __restrict__ y, float* __restrict__ xcopy,
                                                             ConvArgs a, const float2* __restrict__ twtab) {
    extern __shared__ __attribute__((aligned(16))) cx lds[];
    const int t = threadIdx.x;
    const unsigned lb = xcd_logical_block();
    if (lb >= (unsigned)a.nblocks) return;
#ifdef GFX_STAGGER
    if (blockIdx.x >= 256 && blockIdx.x < 512) {
#pragma unroll
        for (int i = 0; i < GFX_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
    }
#endif
#if GFX_CONV_PRIO == 1
    if ((blockIdx.x >> 3) & 1) __builtin_amdgcn_s_setprio(1);
#elif GFX_CONV_PRIO == 2
    if ((blockIdx.x >> 3) & 1) __builtin_amdgcn_s_setprio(3);
#elif GFX_CONV_PRIO == 3
    if ((blockIdx.x >> 11) & 1) __builtin_amdgcn_s_setprio(2);
#endif
    const unsigned ntiles = (unsigned)a.ntiles;
    const unsigned rco = lb / ntiles;
    const int64_t tile = lb - rco * ntiles;
    const unsigned r = rco / (unsigned)a.Cout;
    const int c = (int)(rco - r * (unsigned)a.Cout);
    const float* xrow = x + row_off(a.xmap, r, a.Cin == 1 ? 0 : c);
    float* yrow = y + row_off(a.ymap, r, c);
    const rsrc_t H = make_rsrc(Hs + ((int64_t)(r % a.hrows) * a.Cf + (a.Cf == 1 ? 0 : c)) * H_TILE_F4, H_TILE_F4 * 16);

    // Every global load of the tile is issued up front: the window, the twiddles, and the filter spectrum
    // (needed only after the forward transform, by which time it has long arrived).  Left to itself the
    // compiler issues each spectrum load right before its use and waits for it: 16 serialised L2 round trips.
    TileTw tw;
    cx v[32], w[2][16];
    f4v hreg[H_SLOTS];
#ifdef GFX_T_STAMP
    // experiment: phase timestamps (100 MHz wall clock) of every 64th workgroup into `xcopy` (non-tee launches only)
    unsigned long long stamp[8];
#define T_STAMP(i) stamp[i] = __builtin_amdgcn_s_memrealtime()
#define T_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define T_STAMP(i)
#define T_DRAIN()
#endif
    T_STAMP(0);
    load_window(v, xrow, a.off + tile * a.V - a.O, a.L, t, 1.0f);
    tile_twiddles(tw, twtab, t);
#pragma unroll
    for (int q = 0; q < H_SLOTS; ++q) hreg[q] = buf_load_f4(H, 16u * (uint32_t)t, 4096u * q);
    __builtin_amdgcn_sched_barrier(0);
    T_STAMP(1);
    T_DRAIN();
    T_STAMP(2);
    // off == 0 here: the window's valid part is x[tile*V, tile*V + V) itself
    if (TEE) store_valid<true>(v, xcopy + row_off(a.cmap, r, c), tile * a.V, a.O, a.L, t);
    tile_forward(v, w, tw, lds, t);
    T_STAMP(3);
    for_each_pair(t, tw.base(), [&](int slot, int ia, int ib, cx wk, bool self) {
        cx xe, xo, ye, yo, za, zb;
        pair_split(NAT(w, ia), NAT(w, ib), xe, xo);
        pair_product(xe, xo, hreg[slot], wk, ye, yo);
        pair_merge(ye, yo, za, zb);
        NAT(w, ia) = za;
        if (!self) NAT(w, ib) = zb;
    });
    T_STAMP(4);
    // no barrier here: the inverse starts by writing S2 rows j = t and 512 - t, the very rows (and the only rows)
    // this thread read at the end of the forward transform -- nobody else touches them in between
    tile_inverse(w, v, tw, lds, t);
    T_STAMP(5);
    store_valid(v, yrow, tile * a.V, a.O, a.Lout, t);
    T_STAMP(6);
    T_DRAIN();
    T_STAMP(7);
#ifdef GFX_T_STAMP
    if (!TEE && xcopy && t == 0 && (lb & 63) == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(xcopy) + (lb >> 6) * 12;
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = stamp[q];
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// window j (j = jj - (nparts-1)) of x starts at off - O + j*V; windows that miss [0, L) are skipped.
__device__ __forceinline__ bool window_live(int64_t s, int64_t L) { return s + TILE_F > 0 && s < L; }

__global__ __launch_bounds__(TILE_T, 2) void xspec_kernel(const float* __restrict__ x, float2* __restrict__ Zs,
                                                          ConvArgs a, int64_t nwin,
                                                          const float2* __restrict__ twtab) {
    extern __shared__ __attribute__((aligned(16))) cx lds[];
    const int t = threadIdx.x;
    const unsigned lb = xcd_logical_block();
    if (lb >= (unsigned)a.nblocks) return;
    const unsigned rcx = lb / (unsigned)nwin;
    const int64_t jj = lb - rcx * (unsigned)nwin;
    const int64_t s = a.off - a.O + (jj - (a.nparts - 1)) * a.hop;
    if (!window_live(s, a.L)) return;
    const unsigned xr = rcx / (unsigned)a.Cin;
    const float* xrow = x + row_off(a.xmap, xr, (int)(rcx - xr * (unsigned)a.Cin));
    TileTw tw;
    cx v[32], w[2][16];
    load_window(v, xrow, s, a.L, t, 1.0f);
    tile_twiddles(tw, twtab, t);
    tile_forward(v, w, tw, lds, t);
    cx* out = reinterpret_cast<cx*>(Zs) + (int64_t)lb * TILE_M;
#pragma unroll
    for (int q = 0; q < 32; ++q) out[q * TILE_T + t] = w[q >> 4][q & 15];
}

__global__ __launch_bounds__(TILE_T, 2) void macinv_kernel(const float2* __restrict__ Zs, const float4* __restrict__ Hs,
                                                           float* __restrict__ y, ConvArgs a, int64_t nwin,
                                                           const float2* __restrict__ twtab) {
    extern __shared__ __attribute__((aligned(16))) cx lds[];
    const int t = threadIdx.x;
    const unsigned lb = xcd_logical_block();
    if (lb >= (unsigned)a.nblocks) return;
    const unsigned ntiles = (unsigned)a.ntiles;
    const unsigned rco = lb / ntiles;
    const int64_t tile = lb - rco * ntiles;
    const unsigned r = rco / (unsigned)a.Cout;
    const int c = (int)(rco - r * (unsigned)a.Cout);
    float* yrow = y + row_off(a.ymap, r, c);
    const f4v* H = reinterpret_cast<const f4v*>(Hs) + ((int64_t)(r % a.hrows) * a.Cf + (a.Cf == 1 ? 0 : c)) * a.nparts * H_TILE_F4;
    const cx* Z = reinterpret_cast<const cx*>(Zs) + ((int64_t)r * a.Cin + (a.Cin == 1 ? 0 : c)) * nwin * TILE_M;

    cx ye[H_SLOTS], yo[H_SLOTS];
#pragma unroll
    for (int s = 0; s < H_SLOTS; ++s) ye[s] = yo[s] = cx{0.0f, 0.0f};
    const cx wj = to_cx(twtab[TILE_T + t]);  // W_8192^t

    for (int p = 0; p < a.nparts; ++p) {
        const int64_t j = tile - p;
        if (!window_live(a.off - a.O + j * a.hop, a.L)) continue;
        const cx* Zj = Z + (j + a.nparts - 1) * TILE_M;
        const f4v* Hp = H + (int64_t)p * H_TILE_F4;
        cx w[2][16];
#pragma unroll
        for (int q = 0; q < 32; ++q) w[q >> 4][q & 15] = Zj[q * TILE_T + t];
        for_each_pair(t, wj, [&](int slot, int ia, int ib, cx wk, bool) {
            cx xe, xo;
            pair_split(NAT(w, ia), NAT(w, ib), xe, xo);
            pair_product_acc(xe, xo, Hp[slot * TILE_T + t], wk, ye[slot], yo[slot]);
        });
    }

    cx pz[2][16], v[32];
    for_each_pair(t, wj, [&](int slot, int ia, int ib, cx, bool self) {
        cx za, zb;
        pair_merge(ye[slot], yo[slot], za, zb);
        NAT(pz, ia) = za;
        if (!self) NAT(pz, ib) = zb;
    });
    TileTw tw;
    tile_twiddles(tw, twtab, t);
    tile_inverse(pz, v, tw, lds, t);
    store_valid(v, yrow, tile * a.V, a.O, a.Lout, t);
}

// One output tile per row (ntiles == 1, the filter-gradient shape: a long "filter", few outputs): every signal window
// meets exactly one filter partition, so its spectrum is used once -- transform it here instead of writing it to
// a workspace (xspec_kernel) and reading it back (macinv_kernel).
__global__ __launch_bounds__(TILE_T, 2) void winmac_kernel(const float* __restrict__ x, const float4* __restrict__ Hs,
                                                           float* __restrict__ y, ConvArgs a,
                                                           const float2* __restrict__ twtab) {
    extern __shared__ __attribute__((aligned(16))) cx lds[];
    const int t = threadIdx.x;
    const unsigned rco = xcd_logical_block();
    if (rco >= (unsigned)a.nblocks) return;
    const unsigned r = rco / (unsigned)a.Cout;
    const int c = (int)(rco - r * (unsigned)a.Cout);
    const float* xrow = x + row_off(a.xmap, r, a.Cin == 1 ? 0 : c);
    float* yrow = y + row_off(a.ymap, r, c);
    const f4v* H = reinterpret_cast<const f4v*>(Hs) + ((int64_t)(r % a.hrows) * a.Cf + (a.Cf == 1 ? 0 : c)) * a.nparts * H_TILE_F4;

    cx ye[H_SLOTS], yo[H_SLOTS];
#pragma unroll
    for (int s = 0; s < H_SLOTS; ++s) ye[s] = yo[s] = cx{0.0f, 0.0f};
    TileTw tw;
    tile_twiddles(tw, twtab, t);
    for (int p = 0; p < a.nparts; ++p) {
        const int64_t s = a.off - a.O - (int64_t)p * a.hop;  // window of partition p for output tile 0
        if (!window_live(s, a.L)) continue;
        const f4v* Hp = H + (int64_t)p * H_TILE_F4;
        cx v[32], w[2][16];
        load_window(v, xrow, s, a.L, t, 1.0f);
        tile_forward(v, w, tw, lds, t);
        for_each_pair(t, tw.base(), [&](int slot, int ia, int ib, cx wk, bool) {
            cx xe, xo;
            pair_split(NAT(w, ia), NAT(w, ib), xe, xo);
            pair_product_acc(xe, xo, Hp[slot * TILE_T + t], wk, ye[slot], yo[slot]);
        });
        __syncthreads();  // S2 reads of this window are done before the next window's S1 writes
    }
    cx pz[2][16], v[32];
    for_each_pair(t, tw.base(), [&](int slot, int ia, int ib, cx, bool self) {
        cx za, zb;
        pair_merge(ye[slot], yo[slot], za, zb);
        NAT(pz, ia) = za;
        if (!self) NAT(pz, ib) = zb;
    });
    tile_inverse(pz, v, tw, lds, t);
    store_valid(v, yrow, 0, a.O, a.Lout, t);
}

// Filter gradient of the short-filter convolution, gh[r,c,k] = sum_n g[r,cg,n] x[r,cx,n+off-k], k < N <= 8193, as a
// tile-wise circular correlation: per tile i the V-sample slice x[iV, iV+V) (zero-padded to the tile) is correlated with
// the window g[iV-off, iV-off+16384) -- C = conj(X) G, written with the same polyphase product as the convolution
// (he = conj(Xe), ho = conj(Xo) conj(W^k)) -- summed over the tiles of the row; the first N lags of the inverse
// transform are the gradient.  x and g are read exactly once and nothing else touches memory (the partitioned form
// writes and re-reads 12.5 GB of spectra at the console sizes).
struct CorrArgs {
    gfx_rowmap_t xmap, gmap;
    int64_t L, Lg, off, N, V, ntiles, nblocks;
    int Cx, Cg, Cout;
};

// The tiles are summed in the FREQUENCY domain: conj(X_i) G_i is accumulated over the tiles of the row in registers
// (68 VGPRs of polyphase accumulators) and inverse-transformed once per row -- two transforms per tile.  The twiddles are
// re-fetched for every transform (L2-resident table) instead of being held, which is what lets the accumulators fit
// (256 VGPRs + 160 B of scratch; summing the inverse transforms of every tile in the output row instead -- three
// transforms per tile, no accumulator registers -- is 9 % slower: 8.0 vs 7.3 ms at 8192 rows).
__global__ __launch_bounds__(TILE_T, 2) void corr1_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                           float* __restrict__ gh, CorrArgs a,
                                                           const float2* __restrict__ twtab) {
    extern __shared__ __attribute__((aligned(16))) cx lds[];
    const int t = threadIdx.x;
    const unsigned rco = xcd_logical_block();
    if (rco >= (unsigned)a.nblocks) return;
    const unsigned r = rco / (unsigned)a.Cout;
    const int c = (int)(rco - r * (unsigned)a.Cout);
    const float* xrow = x + row_off(a.xmap, r, a.Cx == 1 ? 0 : c);
    const float* grow = g + row_off(a.gmap, r, a.Cg == 1 ? 0 : c);
    const cx wj = to_cx(twtab[TILE_T + t]);  // W_8192^t
    const float sc = 1.0f / (4.0f * TILE_M);
    cx ye[H_SLOTS], yo[H_SLOTS];
#pragma unroll
    for (int q = 0; q < H_SLOTS; ++q) ye[q] = yo[q] = cx{0.0f, 0.0f};
    auto fresh_twiddles = [&](TileTw& tw) {
        int tt = t;
        asm volatile("" : "+v"(tt));  // a new load every time: the compiler must not keep 32 VGPRs of twiddles alive
        tile_twiddles(tw, twtab, tt);
    };
    for (int64_t i = 0; i < a.ntiles; ++i) {
        const int64_t s = i * a.V;
        f4v hreg[H_SLOTS];
        {
            cx v[32], w[2][16];
            TileTw tw;
            const int64_t xend = s + a.V < a.L ? s + a.V : a.L;
            load_window<false>(v, xrow, s, xend, t, 1.0f);   // (the two-form loader costs this kernel 500 B of scratch)
            fresh_twiddles(tw);
            tile_forward(v, w, tw, lds, t);
            for_each_pair(t, wj, [&](int slot, int ia, int ib, cx wk, bool) {
                cx xe, xo;
                pair_split(NAT(w, ia), NAT(w, ib), xe, xo);
                hreg[slot] = __builtin_shufflevector(cconj(xe) * sc, cmulc(cconj(xo), wk) * sc, 0, 1, 2, 3);
            });
        }
        __syncthreads();  // S2 reads of the x transform are done before the g transform's S1 writes
        {
            cx v[32], w[2][16];
            TileTw tw;
            load_window<false>(v, grow, s - a.off, a.Lg, t, 1.0f);
            fresh_twiddles(tw);
            tile_forward(v, w, tw, lds, t);
            for_each_pair(t, wj, [&](int slot, int ia, int ib, cx wk, bool) {
                cx ge, go;
                pair_split(NAT(w, ia), NAT(w, ib), ge, go);
                pair_product_acc(ge, go, hreg[slot], wk, ye[slot], yo[slot]);
            });
        }
        __syncthreads();
    }
    cx pz[2][16], v[32];
    for_each_pair(t, wj, [&](int slot, int ia, int ib, cx, bool self) {
        cx za, zb;
        pair_merge(ye[slot], yo[slot], za, zb);
        NAT(pz, ia) = za;
        if (!self) NAT(pz, ib) = zb;
    });
    TileTw tw;
    fresh_twiddles(tw);
    tile_inverse(pz, v, tw, lds, t);
    store_valid(v, gh + ((int64_t)r * a.Cout + c) * a.N, 0, 0, a.N, t);
}

// ------------------------------------------------------------------------------------------------
// fftconv1pp_kernel: the same tile arithmetic as fftconv1_kernel, organised for large launches as ONE persistent
// 512-thread workgroup per CU whose two halves (4 waves each, one per SIMD) work half a tile apart ("ping-pong"):
//
//   * every half-period one half runs the FIRST part of a tile (window -> registers, input copy, forward passes 1-3,
//     spectral product: role X) while the other runs the SECOND part of the previous tile (inverse passes, output
//     stores: role Y), then the roles swap.  Nine barriers per half-period order the four LDS exchanges of the two
//     halves (X: 1, 2; Y: 3, 4) on ONE shared exchange image, so that in every interval one half moves data through
//     the LDS while the other issues packed-FP32 arithmetic: the LDS round trips of one half hide under the
//     arithmetic of the other by construction instead of by the chance alignment of two independent workgroups.
//   * the 64 KB the second exchange image used to take is a circular buffer of the input row ("ring").  Windows
//     arrive by LDS-DMA (buffer_load_dword ... lds: no VGPRs, no wait), requested one tile ahead by the half that has
//     just lifted its window into registers.  Consecutive windows of a row overlap by O samples and the ring keeps
//     them: every input sample crosses HBM -> LDS exactly once, the overlap is never re-read.
//   * a workgroup walks (filter row, channel, batch) units in filter-major order, so its filter spectrum (68 VGPRs)
//     and twiddles (32 VGPRs) are loaded once and stay in registers; nothing but the window stream and the output
//     stores touches the vector-memory queue in steady state.
//
// Geometry: O = N - 1 rounded up to 512 samples (V = 16384 - O a multiple of 512), so that a window starts on a ring
// row (256 complex values) and a DMA chunk (64 samples = 256 B) never straddles the ring's wrap-around.
constexpr int PP_T = 512;
constexpr int PP_LDS_BYTES = TILE_LDS_BYTES + TILE_F * 4;   // exchange image + ring = 139,264 B -> one workgroup per CU
constexpr int PP_CHUNK = 256;                                // samples per window-request chunk (16 B per lane)

struct PPArgs {
    ConvArgs a;        // with the ping-pong geometry (O, V, ntiles)
    unsigned units;    // R * Cout (row, output channel) pairs
    unsigned bh;       // rows per filter row: R / hrows; units are ordered (filter row, channel, batch)
};

#ifndef GFX_CONV_PRIO
#define GFX_CONV_PRIO 0   // experiment: static wave priority for every other workgroup of fftconv1_kernel
#endif
#ifndef GFX_PP_ABLATE
#define GFX_PP_ABLATE 0   // timing experiments only: 1 no window requests, 2 no stores, 4 no spectrum loads, 8 no compute
#endif
#ifndef GFX_PP_EXP
#define GFX_PP_EXP 0      // timing experiments: 1 head into fresh registers (store WAR), 2 drop the i9 barrier (WRONG results)
#endif
#define LDSP(p) reinterpret_cast<__attribute__((address_space(3))) void*>((__attribute__((address_space(3))) char*)(p))

// Barrier that orders LDS traffic only: pending global stores and LDS-DMA loads stay in flight across it.  Written as
// inline asm (with the LDS wait it needs) because a compiler-visible fence or barrier makes hipcc drain the
// vector-memory queue -- it treats in-flight LDS-DMA as LDS writes the barrier must publish.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
// s_waitcnt vmcnt(N) through the builtin (gfx9 encoding: vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt 15 << 8 | vmcnt[5:4] << 14), so
// that hipcc's own wait bookkeeping sees it: after an inline-asm wait it still believes earlier loads are pending and
// sprinkles vmcnt(0) over the loop, each of which drains the window requests and stores in flight.
template <int N>
__device__ __forceinline__ void wait_vmem_le() {
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void wait_vmem() { wait_vmem_le<0>(); }

struct PPTile {          // where tile `s` of the workgroup's stream lives (all wave-uniform)
    unsigned r, tile, hidx;
    int c;
};

__device__ __forceinline__ PPTile pp_decode(const PPArgs& pa, unsigned u0, unsigned s) {
    const unsigned ntiles = (unsigned)pa.a.ntiles;
    const unsigned un = s / ntiles;
    PPTile q;
    q.tile = s - un * ntiles;
    const unsigned u = u0 + un;
    const unsigned b = u % pa.bh, fc = u / pa.bh;
    const unsigned j = fc / (unsigned)pa.a.Cout;
    q.c = (int)(fc - j * (unsigned)pa.a.Cout);
    q.r = b * pa.a.hrows + j;
    q.hidx = j * (unsigned)pa.a.Cf + (pa.a.Cf == 1 ? 0u : (unsigned)q.c);
    return q;
}

// One LDS-DMA instruction: lane l fetches the dword at (descriptor base + voff + soff) into LDS at lds_addr + 4 l.
// Inline asm, so that hipcc does not know about it: for a builtin LDS-DMA it makes every later LDS read and every
// barrier of the wave wait for the transfer (vmcnt(0)), which is exactly the latency this kernel is built to hide.
// The waves that issue requests wait for them explicitly (wait_vmem) before the barrier that opens the next half-period.
using i32x4 = int __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 dma_rsrc(const void* base, int64_t bytes) {
    const uint64_t p = reinterpret_cast<uint64_t>(base);
    const int64_t nb = bytes < 0 ? 0 : (bytes > 0x7fffffffLL ? 0x7fffffffLL : bytes);
    return i32x4{(int)__builtin_amdgcn_readfirstlane((uint32_t)p), (int)__builtin_amdgcn_readfirstlane((uint32_t)(p >> 32) & 0xffffu),
                 (int)__builtin_amdgcn_readfirstlane((uint32_t)nb), 0x00020000};
}
#define GFX_DMA_ASM(insn)                                                                                     \
    unsigned keep;                                                                                            \
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t" insn " %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0" \
                 : "=&s"(keep)                                                                                \
                 : "v"(voff), "s"(rs), "s"(lds_addr), "s"(soff)                                               \
                 : "memory")
// s_nop 4: the scalar operands may have just been written by a VALU instruction (v_readfirstlane, or v_readlane of a
// spilled SGPR); s_nop 0: M0 write -> LDS-DMA read.  M0 is saved and restored around the instruction.
__device__ __forceinline__ void dma_dword(i32x4 rs, uint32_t lds_addr, uint32_t voff, uint32_t soff) {
    GFX_DMA_ASM("buffer_load_dword");
}
__device__ __forceinline__ void dma_dwordx4(i32x4 rs, uint32_t lds_addr, uint32_t voff, uint32_t soff) {
    GFX_DMA_ASM("buffer_load_dwordx4");
}
#undef GFX_DMA_ASM

// Request the part of tile q's window that the ring does not hold yet: everything for the first tile of a row, the V new
// samples otherwise, in chunks of 256 samples (1 KB); chunk k of the request goes to wave slot k % nslots.  Interior
// chunks are one 16-byte-per-lane transfer with a scalar offset (only 4-byte alignment is needed on the global side);
// chunks that touch the row's ends are four dword transfers with per-lane offsets, and the buffer range check supplies
// the zeros (LDS-DMA writes a zero for a lane whose offset fails the check: tools/ubench/lds_dma.hip).
__device__ __forceinline__ void pp_request(const float* __restrict__ x, uint32_t ring_lds, const PPArgs& pa,
                                           const PPTile& q, int slot, int nslots, int lane) {
    const ConvArgs& a = pa.a;
    const float* xrow = x + row_off(a.xmap, q.r, a.Cin == 1 ? 0 : q.c);
    const i32x4 rs = dma_rsrc(xrow, a.L * 4);
    const int vch = (int)(a.V / PP_CHUNK);
    const int nch = q.tile == 0 ? TILE_F / PP_CHUNK : vch;
    // first source sample and first ring chunk of the request
    const int64_t n_first = q.tile == 0 ? a.off - a.O : a.off + (int64_t)q.tile * a.V;
    const int rc_first = q.tile == 0 ? 0 : (int)(((int64_t)(q.tile - 1) * vch) & (TILE_F / PP_CHUNK - 1));
    for (int k = slot; k < nch; k += nslots) {
        const int64_t n0 = n_first + (int64_t)k * PP_CHUNK;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(
            ring_lds + (uint32_t)(((rc_first + k) & (TILE_F / PP_CHUNK - 1)) * (PP_CHUNK * 4)));
        if (n0 >= 0 && n0 + PP_CHUNK <= a.L) {
            dma_dwordx4(rs, dst, 16u * (uint32_t)lane, __builtin_amdgcn_readfirstlane((uint32_t)(4 * n0)));
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t n = n0 + 64 * j + lane;
                dma_dword(rs, dst + 256u * j, (n >= 0 && n < a.L) ? (uint32_t)(4 * n) : OOB, 0u);
            }
        }
    }
}

// Twiddles of the ping-pong kernel live in LDS (rows lo1[1..3], hi1[1..7] of the per-device table: 10 x 256 values; and
// lo2[1..3], hi2[1..3] for d = 0..15): a segment reads the handful it needs right before using them, so no twiddle
// register survives a barrier, and nothing but window requests and stores enters the vector-memory queue.
constexpr int PP_TW1_F2 = 10 * 256;
constexpr int PP_TW2_F2 = 6 * 16;
constexpr int PP_LDS_TOTAL = PP_LDS_BYTES + (PP_TW1_F2 + PP_TW2_F2) * 8;   // 160,512 B of the CU's 163,840

__device__ __forceinline__ void pp_tw1(TileTw& tw, const cx* tw1, int t) {
    tw.lo1[0] = tw.hi1[0] = cx{1.0f, 0.0f};
#pragma unroll
    for (int i = 1; i < 4; ++i) tw.lo1[i] = tw1[(i - 1) * 256 + t];
#pragma unroll
    for (int i = 1; i < 8; ++i) tw.hi1[i] = tw1[(2 + i) * 256 + t];
}
__device__ __forceinline__ void pp_tw2(TileTw& tw, const cx* tw2, int d) {
    tw.lo2[0] = tw.hi2[0] = cx{1.0f, 0.0f};
#pragma unroll
    for (int i = 1; i < 4; ++i) tw.lo2[i] = tw2[(i - 1) * 16 + d];
#pragma unroll
    for (int i = 1; i < 4; ++i) tw.hi2[i] = tw2[(2 + i) * 16 + d];
}

template <bool TEE>
__global__ __launch_bounds__(PP_T, 2) void fftconv1pp_kernel(const float* __restrict__ x, const float4* __restrict__ Hs,
                                                             float* __restrict__ y, float* __restrict__ xcopy, PPArgs pa,
                                                             const float2* __restrict__ twtab) {
    extern __shared__ __attribute__((aligned(16))) cx lds[];
    float* ring = reinterpret_cast<float*>(lds + TILE_LDS_F2);
    const cx* ringc = reinterpret_cast<const cx*>(ring);
    cx* tw1 = lds + TILE_LDS_F2 + TILE_M;
    cx* tw2 = tw1 + PP_TW1_F2;
    const ConvArgs& a = pa.a;
    const int tid = threadIdx.x;
    const int t = tid & 255, lane = tid & 63;
    const int half = __builtin_amdgcn_readfirstlane(tid >> 8);
    const int wih = __builtin_amdgcn_readfirstlane((tid >> 6) & 3);   // wave within the half
    const unsigned G = gridDim.x, wg = blockIdx.x;
    const unsigned u0 = (unsigned)((uint64_t)wg * pa.units / G), u1 = (unsigned)((uint64_t)(wg + 1) * pa.units / G);
    const unsigned S = (u1 - u0) * (unsigned)a.ntiles;   // tiles in this workgroup's stream
    if (S == 0) return;
    const int vrows = (int)(a.V / 512);                  // ring rows (256 complex) a window start advances per tile

    // twiddle tables -> LDS (table rows: 0-3 lo1, 4-11 hi1, 12-15 lo2, 16-19 hi2; rows 0, 4, 12, 16 are all ones)
    for (int i = tid; i < PP_TW1_F2; i += PP_T) {
        const int row = i >> 8;
        tw1[i] = to_cx(twtab[(row < 3 ? 1 + row : 2 + row) * TILE_T + (i & 255)]);
    }
    if (tid < PP_TW2_F2) {
        const int row = tid >> 4;
        tw2[tid] = to_cx(twtab[(row < 3 ? 13 + row : 14 + row) * TILE_T + (tid & 15)]);
    }

    const int kk = t >> 4, d = t & 15;
    const uint32_t ring_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)LDSP(ring));   // LDS byte address

    // Loop-carried registers of a half: `v` = the head of its next tile (window lifted from the ring, first radix pass and
    // its twiddles done, ready for exchange 1); `pz` = its tile after the spectral product and the first inverse pass,
    // ready for exchange 3.  Only one of the two is live at a period boundary.
    // One 64-VGPR array carries both (as far as the compiler can tell, two arrays would both be live at the loop header).
    cx st[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) st[q] = cx{0.0f, 0.0f};
    PPTile cur = pp_decode(pa, u0, 0);                   // the tile this half is working on

    // window -> registers, input copy, first forward pass (role Y', last interval; and the prologue for tile 0)
    auto head = [&](const PPTile& q, bool live) {
        const int a0 = (int)((q.tile * (unsigned)vrows) & 31u);               // ring row of the window's first sample
#pragma unroll
        for (int i = 0; i < 32; ++i) st[i] = ringc[(((i + a0) & 31) << 8) + t];
        if (TEE && live && !(GFX_PP_ABLATE & 2))
            store_valid<true>(st, xcopy + row_off(a.cmap, q.r, q.c), (int64_t)q.tile * a.V, a.O, a.L, t);
        dif<32, false>(st);
        TileTw tw;
        pp_tw1(tw, tw1, t);
#pragma unroll
        for (int r = 0; r < 32; ++r) st[r] = tw.fwd1(st[r], brev(r, 5));
    };

    // All workgroups start together and do identical work, so left alone they stay in phase: every CU would request its
    // window, and store its tile, at the same moment -- bursts of 12 MB that the memory system serves while the whole
    // chip waits, then idles while the whole chip computes.  Spread the phases over one tile period once, at the start
    // (the relative phases persist: every workgroup runs at the same pace).
#ifndef GFX_PP_STAGGER
#define GFX_PP_STAGGER 200   // cycles per phase step, 64 steps
#endif
    {
        const unsigned steps = ((wg * 0x9E3779B1u) >> 26) * (GFX_PP_STAGGER / 64 + 1);   // 0..63 phase steps
        for (unsigned i = 0; i < steps; ++i) __builtin_amdgcn_s_sleep(1);                // 64 cycles each
    }

    // prologue: the first window, requested by all eight waves; half 0 lifts it
    pp_request(x, ring_lds, pa, cur, half * 4 + wih, 8, lane);
    wait_vmem();
    lds_barrier();
    if (half == 0) head(cur, true);

    // Period p: one half runs role X' on tile p (exchanges 1 and 2, last forward pass, spectral product, first inverse
    // pass); the other runs role Y' (exchanges 3 and 4 and the last inverse pass of tile p-1, its output stores, then the
    // head of tile p+1).  The four exchanges take turns on the one exchange image (3, 1, 4, 2: intervals i1..i8); from
    // i8 on role Y' is done with the LDS image and runs its store-heavy tail + head against role X's arithmetic-only
    // product stretch.  Tiles outside the stream (role Y' tail at p = 0, role X' at p = S, heads past the end) run the
    // arithmetic on whatever the registers hold and skip only their memory side effects.
    for (unsigned p = 0; p <= S; ++p) {
        if ((int)(p & 1) == half) {
            // ------------------------------------------------------------------------ role X': tile p (cur), v ready
            cx u[2][16], w[2][16];
            // the tile's filter spectrum (L2-resident; used after i8).  Fetched per tile rather than kept across the loop: 68
            // VGPRs held through both roles push the allocator into scratch, and a scratch reload is a vector-memory
            // load that waits behind everything in flight.
            f4v hreg[H_SLOTS];
            {
                const rsrc_t H = make_rsrc(Hs + (int64_t)cur.hidx * H_TILE_F4, H_TILE_F4 * 16);
#pragma unroll
                for (int q = 0; q < H_SLOTS; ++q)
                    hreg[q] = (GFX_PP_ABLATE & 4) ? f4v{1.0f, 0.0f, 0.5f, 0.0f} : buf_load_f4(H, 16u * (uint32_t)t, 4096u * q);
            }
            lds_barrier();                                                    // i1 (other half: W3, window request)
            lds_barrier();                                                    // i2 (other half: R3)
            lds_barrier();                                                    // i3: W1
#pragma unroll
            for (int r = 0; r < 32; ++r) lds[s1_at(brev(r, 5), t)] = st[r];
            lds_barrier();                                                    // i4: R1, first radix-16 set
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int c = 0; c < 16; ++c) u[s][c] = lds[s1_at(kk + 16 * s, 16 * c + d)];
            dif<16, false>(u[0]);
            lds_barrier();                                                    // i5 (other half: W4)
            dif<16, false>(u[1]);
            {
                TileTw tw;
                pp_tw2(tw, tw2, d);
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int r = 0; r < 16; ++r) u[s][r] = tw.fwd2(u[s][r], brev(r, 4));
            }
            lds_barrier();                                                    // i6 (other half: R4)
            lds_barrier();                                                    // i7: W2
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int r = 0; r < 16; ++r) lds[s2_row(brev(r, 4), kk + 16 * s) + d] = u[s][r];
            lds_barrier();                                                    // i8: R2, last forward pass
#pragma unroll
            for (int bf = 0; bf < 2; ++bf) {
                const int j = bf ? bf_b(t) : bf_a(t);
                const f4v* row = reinterpret_cast<const f4v*>(lds + s2_row(j >> 5, j & 31));
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const f4v pq = row[q];
                    w[bf][2 * q] = pq.lo;
                    w[bf][2 * q + 1] = pq.hi;
                }
            }
            dif<16, false>(w[0]);
            dif<16, false>(w[1]);
            if (!(GFX_PP_EXP & 2)) lds_barrier();                             // i9 (other half: next window has landed)
            // take delivery of the spectrum HERE, on every path: the product below sits in two lane-divergent blocks that a
            // wave may skip, and hipcc would otherwise carry "loads may be pending" into the other role and guard its
            // registers with vmcnt waits -- each of which drains the window request in flight
#pragma unroll
            for (int q = 0; q < H_SLOTS; ++q) asm volatile("" : "+v"(hreg[q]));
            for_each_pair(t, tw1[t], [&](int slot, int ia, int ib, cx wk, bool self) {
                cx xe, xo, ye, yo, za, zb;
                pair_split(NAT(w, ia), NAT(w, ib), xe, xo);
                pair_product(xe, xo, hreg[slot], wk, ye, yo);
                pair_merge(ye, yo, za, zb);
                NAT(w, ia) = za;
                if (!self) NAT(w, ib) = zb;
            });
#pragma unroll
            for (int bf = 0; bf < 2; ++bf) {
                cx pz[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) pz[k] = w[bf][brev(k, 4)];      // register renaming only
                dif<16, true>(pz);
#pragma unroll
                for (int k = 0; k < 16; ++k) st[bf * 16 + k] = pz[k];
            }
        } else {
            // ------------------------------------------------------------------------ role Y': tail of tile p-1, head of p+1
            const bool tail = p >= 1, more = p + 1 < S;
            cx u[2][16];
            lds_barrier();                                                    // i1: W3; the ring can take window p+1
#pragma unroll
            for (int bf = 0; bf < 2; ++bf) {
                const int j = bf ? bf_b(t) : bf_a(t);
                f4v* row = reinterpret_cast<f4v*>(lds + s2_row(j >> 5, j & 31));
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    row[q] = __builtin_shufflevector(st[bf * 16 + brev(2 * q, 4)], st[bf * 16 + brev(2 * q + 1, 4)], 0, 1, 2, 3);
            }
            const PPTile nxt = pp_decode(pa, u0, more ? p + 1 : S - 1);
            if (!(GFX_PP_ABLATE & 1) && more) pp_request(x, ring_lds, pa, nxt, wih, 4, lane);
            lds_barrier();                                                    // i2: R3, first inverse radix-16 set
            {
                TileTw tw;
                pp_tw2(tw, tw2, d);
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int k2 = 0; k2 < 16; ++k2) u[s][k2] = tw.inv2(lds[s2_row(k2, kk + 16 * s) + d], k2);
            }
            dif<16, true>(u[0]);
            lds_barrier();                                                    // i3 (other half: W1)
            dif<16, true>(u[1]);
            lds_barrier();                                                    // i4 (other half: R1)
            lds_barrier();                                                    // i5: W4
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int r = 0; r < 16; ++r) lds[s1_at(kk + 16 * s, 16 * brev(r, 4) + d)] = u[s][r];
            lds_barrier();                                                    // i6: R4
            {
                TileTw tw;
                pp_tw1(tw, tw1, t);
#pragma unroll
                for (int k1 = 0; k1 < 32; ++k1) st[k1] = tw.inv1(lds[s1_at(k1, t)], k1);
            }
            lds_barrier();                                                    // i7 (other half: W2): last inverse pass
            dif<32, true>(st);
            lds_barrier();                                                    // i8 (other half: R2 ...): output stores
            if (tail && !(GFX_PP_ABLATE & 2))
                store_valid(st, y + row_off(a.ymap, cur.r, cur.c), (int64_t)cur.tile * a.V, a.O, a.Lout, t);
            // This half requested window p+1 itself (i1) and has since issued only these output stores (32 buffer stores,
            // plus a single-sample store on odd row ends): the vector-memory queue completes in order, so "at most 32
            // outstanding" means the request has landed, without waiting for the stores.  (p = 0: nothing was stored.)
            if (GFX_PP_EXP & 4) {}                                             // timing experiment: no wait at all
            else if (tail && !(GFX_PP_ABLATE & 2)) wait_vmem_le<32>();
            else wait_vmem();
            if (!(GFX_PP_EXP & 2)) lds_barrier();                             // i9: window p+1 is in the ring: head
            cur = nxt;
            if (GFX_PP_EXP & 1) {
                cx keep[32];
#pragma unroll
                for (int q = 0; q < 32; ++q) keep[q] = st[q];
                head(cur, more);
#pragma unroll
                for (int q = 0; q < 32; ++q) asm volatile("" : "+v"(keep[q]));   // the store sources stay untouched until here
            } else {
                head(cur, more);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// fftconv1h_kernel: the tile arithmetic of fftconv1_kernel with HALF-SIZE LDS exchanges, so that THREE workgroups fit a
// CU (the production kernel's 73.7 KB exchange image allows two, and two resident tiles leave the vector ALU ~60 % used).
// Every exchange moves its 32 values per thread in two rounds of 16 through an image of half the size, split along an
// index that lets every thread read exactly half of what it needs per round:
//   exchange 1 / 4 (S1 image, 16 of its 32 k1 rows): round s in {0,1} carries rows k1 in [16 s, 16 s + 16) -- the second
//                 pass's two radix-16 sets;
//   exchange 2 / 3 (S2 image, 8 of its 16 k2 planes): round 0 carries butterflies j < 256 (the thread's j = t),
//                 round 1 j >= 256 (its mirror 512 - t).
// Same LDS instruction count as the full exchanges (8-byte accesses), twice the barriers (15 per tile instead of 7) --
// round 1 measured a re/im split (twice the instructions AND barriers) at -28 %.  Twiddles: stage 2 and the eight
// "hi" stage-1 rows live in LDS next to the image (14.8 KB), only lo1[1..3] stay in registers, so that the kernel fits
// 168 VGPRs.  LDS: 36,864 + 15,104 = 51,968 B per workgroup.
constexpr int HX_IMG_F2 = 256 * S2_ROW;                 // 4608 float2 = 36,864 B (S1 half: 16 x 272 = 4352 fits too)
constexpr int HX_TW1_F2 = 7 * 256;                      // hi1[1..7][t]
constexpr int HX_TW2_F2 = 6 * 16;                       // lo2[1..3][d], hi2[1..3][d]
constexpr int HX_LDS_BYTES = (HX_IMG_F2 + HX_TW1_F2 + HX_TW2_F2) * 8;

__device__ __forceinline__ int s1h_at(int k1h, int b) { return k1h * S1_ROW + b; }
__device__ __forceinline__ int s2h_row(int k2h, int k1) { return (k2h * 32 + k1) * S2_ROW; }

struct HxTw {
    cx lo1[4];             // registers: W_8192^(t i), i < 4
    const cx* hi1;         // LDS: hi1[(i - 1) * 256 + t] = W_8192^(4 i t), i = 1..7
    const cx* tw2;         // LDS: lo2[1..3][d], hi2[1..3][d]
    __device__ __forceinline__ void stage1(TileTw& tw, int t) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) tw.lo1[i] = lo1[i];
        tw.hi1[0] = cx{1.0f, 0.0f};
#pragma unroll
        for (int i = 1; i < 8; ++i) tw.hi1[i] = hi1[(i - 1) * 256 + t];
    }
    __device__ __forceinline__ void stage2(TileTw& tw, int d) const {
        tw.lo2[0] = tw.hi2[0] = cx{1.0f, 0.0f};
#pragma unroll
        for (int i = 1; i < 4; ++i) tw.lo2[i] = tw2[(i - 1) * 16 + d];
#pragma unroll
        for (int i = 1; i < 4; ++i) tw.hi2[i] = tw2[(2 + i) * 16 + d];
    }
};

// `mid()` runs once the first exchange is done (the 32 window registers are dead, the second pass not yet started):
// the caller requests the filter spectrum there, so that its 68 registers are not live during the first pass.
template <typename Mid>
__device__ __forceinline__ void tile_forward_hx(cx (&v)[32], cx (&w)[2][16], const HxTw& hx, cx* lds, int t, Mid&& mid) {
    const int kk = t >> 4, d = t & 15;
    dif<32, false>(v);
    {
        TileTw tw;
        hx.stage1(tw, t);
#pragma unroll
        for (int r = 0; r < 32; ++r) v[r] = tw.fwd1(v[r], brev(r, 5));
    }
    cx u[2][16];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const int k1 = brev(r, 5);
            if ((k1 >> 4) == s) lds[s1h_at(k1 & 15, t)] = v[r];
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 16; ++c) u[s][c] = lds[s1h_at(kk, 16 * c + d)];
        __syncthreads();
        if (s == 1) {
            __builtin_amdgcn_sched_barrier(0);
            mid();
            __builtin_amdgcn_sched_barrier(0);
            dif<16, false>(u[0]);
        }
    }
    dif<16, false>(u[1]);
    {
        TileTw tw;
        hx.stage2(tw, d);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) u[s][r] = tw.fwd2(u[s][r], brev(r, 4));
    }
#pragma unroll
    for (int bf = 0; bf < 2; ++bf) {                    // round bf carries planes k2 in [8 bf, 8 bf + 8)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k2 = brev(r, 4);
                if ((k2 >> 3) == bf) lds[s2h_row(k2 & 7, kk + 16 * s) + d] = u[s][r];
            }
        __syncthreads();
        const int j = bf ? bf_b(t) : bf_a(t);
        const f4v* row = reinterpret_cast<const f4v*>(lds + s2h_row((j >> 5) & 7, j & 31));
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f4v pq = row[q];
            w[bf][2 * q] = pq.lo;
            w[bf][2 * q + 1] = pq.hi;
        }
        __syncthreads();
        if (bf == 1) dif<16, false>(w[0]);
    }
    dif<16, false>(w[1]);
}

// the last barrier of tile_forward_hx already separates its reads from the writes below
__device__ __forceinline__ void tile_inverse_hx(cx (&w)[2][16], cx (&v)[32], const HxTw& hx, cx* lds, int t) {
    const int kk = t >> 4, d = t & 15;
    cx u[2][16];
    TileTw tw2;
    hx.stage2(tw2, d);
#pragma unroll
    for (int bf = 0; bf < 2; ++bf) {
        cx p[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) p[k] = w[bf][brev(k, 4)];  // register renaming only
        dif<16, true>(p);
        const int j = bf ? bf_b(t) : bf_a(t);
        f4v* row = reinterpret_cast<f4v*>(lds + s2h_row((j >> 5) & 7, j & 31));
#pragma unroll
        for (int q = 0; q < 8; ++q) row[q] = __builtin_shufflevector(p[brev(2 * q, 4)], p[brev(2 * q + 1, 4)], 0, 1, 2, 3);
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int k2h = 0; k2h < 8; ++k2h)
                u[s][8 * bf + k2h] = tw2.inv2(lds[s2h_row(k2h, kk + 16 * s) + d], 8 * bf + k2h);
        __syncthreads();
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        dif<16, true>(u[s]);
#pragma unroll
        for (int r = 0; r < 16; ++r) lds[s1h_at(kk, 16 * brev(r, 4) + d)] = u[s][r];
        __syncthreads();
#pragma unroll
        for (int k1h = 0; k1h < 16; ++k1h) v[16 * s + k1h] = lds[s1h_at(k1h, t)];
        if (s == 0) __syncthreads();
    }
    {
        TileTw tw;
        hx.stage1(tw, t);
#pragma unroll
        for (int k1 = 0; k1 < 32; ++k1) v[k1] = tw.inv1(v[k1], k1);
    }
    dif<32, true>(v);
}

template <bool TEE>
__global__ __launch_bounds__(TILE_T, 3) void fftconv1h_kernel(const float* __restrict__ x, const float4* __restrict__ Hs,
                                                              float* __restrict__ y, float* __restrict__ xcopy,
                                                              ConvArgs a, const float2* __restrict__ twtab) {
    extern __shared__ __attribute__((aligned(16))) cx lds[];
    const int t = threadIdx.x;
    const unsigned lb = xcd_logical_block();
    if (lb >= (unsigned)a.nblocks) return;
    const unsigned ntiles = (unsigned)a.ntiles;
    const unsigned rco = lb / ntiles;
    const int64_t tile = lb - rco * ntiles;
    const unsigned r = rco / (unsigned)a.Cout;
    const int c = (int)(rco - r * (unsigned)a.Cout);
    const float* xrow = x + row_off(a.xmap, r, a.Cin == 1 ? 0 : c);
    float* yrow = y + row_off(a.ymap, r, c);
    const rsrc_t H = make_rsrc(Hs + ((int64_t)(r % a.hrows) * a.Cf + (a.Cf == 1 ? 0 : c)) * H_TILE_F4, H_TILE_F4 * 16);

    cx* tw1 = lds + HX_IMG_F2;
    cx* tw2 = tw1 + HX_TW1_F2;
    HxTw hx;
    hx.hi1 = tw1;
    hx.tw2 = tw2;
    cx v[32], w[2][16];
    f4v hreg[H_SLOTS];
    load_window(v, xrow, a.off + tile * a.V - a.O, a.L, t, 1.0f);
    // twiddle table rows: 0-3 lo1, 4-11 hi1, 12-15 lo2, 16-19 hi2
    hx.lo1[0] = cx{1.0f, 0.0f};
#pragma unroll
    for (int i = 1; i < 4; ++i) hx.lo1[i] = to_cx(twtab[i * TILE_T + t]);
    cx hi[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) hi[i] = to_cx(twtab[(5 + i) * TILE_T + t]);
    cx t2 = {0.0f, 0.0f};
    if (t < HX_TW2_F2) {
        const int row = t >> 4;
        t2 = to_cx(twtab[(row < 3 ? 13 + row : 14 + row) * TILE_T + (t & 15)]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 7; ++i) tw1[i * 256 + t] = hi[i];
    if (t < HX_TW2_F2) tw2[t] = t2;
    if (TEE) store_valid<true>(v, xcopy + row_off(a.cmap, r, c), tile * a.V, a.O, a.L, t);
    __syncthreads();                                   // twiddle tables visible
    tile_forward_hx(v, w, hx, lds, t, [&]() {
#pragma unroll
        for (int q = 0; q < H_SLOTS; ++q) hreg[q] = buf_load_f4(H, 16u * (uint32_t)t, 4096u * q);
    });
    for_each_pair(t, hx.lo1[1], [&](int slot, int ia, int ib, cx wk, bool self) {
        cx xe, xo, ye, yo, za, zb;
        pair_split(NAT(w, ia), NAT(w, ib), xe, xo);
        pair_product(xe, xo, hreg[slot], wk, ye, yo);
        pair_merge(ye, yo, za, zb);
        NAT(w, ia) = za;
        if (!self) NAT(w, ib) = zb;
    });
    tile_inverse_hx(w, v, hx, lds, t);
    store_valid(v, yrow, tile * a.V, a.O, a.Lout, t);
}

static inline unsigned pad8(int64_t n) { return (unsigned)(((n + 7) / 8) * 8); }

template <typename K>
static int allow_lds(K kernel) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               TILE_LDS_BYTES) == hipSuccess
               ? 0
               : GFX_ELAUNCH;
}

}  // namespace gfx

using namespace gfx;

extern "C" {

int64_t gfx_fftconv_nparts(int64_t N) { return N <= 0 ? 0 : conv_geom(N, 1).nparts; }

int64_t gfx_fftconv_part_len(int64_t N, int64_t Lout) {
    if (N <= TILE_M + 1 || Lout <= 0 || Lout > TILE_M - 2) return 0;  // default geometry
    return (TILE_F - Lout) & ~int64_t(1);
}

// experiment builds (-DGFX_W_STAMP): a 4 MB device buffer the wide kernel writes its phase timestamps into
static void* gfx_dbg_stamp_buffer() {
#if defined(GFX_W_STAMP) || defined(GFX_T_STAMP)
    static void* p = nullptr;
    if (!p && hipMalloc(&p, 4 << 20) != hipSuccess) p = nullptr;
    return p;
#else
    return nullptr;
#endif
}
#if defined(GFX_W_STAMP) || defined(GFX_T_STAMP)
int gfx_dbg_stamp_read(void* host, size_t bytes) {
    void* p = gfx_dbg_stamp_buffer();
    if (!p || bytes > (4u << 20)) return GFX_EINVAL;
    return hipMemcpy(host, p, bytes, hipMemcpyDeviceToHost) == hipSuccess ? GFX_OK : GFX_ELAUNCH;
}
#endif

size_t gfx_fftconv_wide_ws_bytes(int64_t h_rows, int64_t C_f) {
    return (h_rows <= 0 || C_f <= 0) ? 0 : (size_t)h_rows * C_f * wide::W_H_F4 * 16;
}

size_t gfx_fir_spectrum_bytes(int64_t RCf, int64_t N) { return gfx_fir_spectrum_bytes_ex(RCf, N, 0); }

size_t gfx_fir_spectrum_bytes_ex(int64_t RCf, int64_t N, int64_t part_len) {
    if (RCf <= 0 || N <= 0) return 0;
    const ConvGeom g = conv_geom(N, 1, part_len);
    return g.ok ? (size_t)RCf * g.nparts * H_TILE_F4 * sizeof(float4) : 0;
}

size_t gfx_fftconv_workspace_bytes(int64_t R, int64_t C_in, int64_t L, int64_t Lout, int64_t off, int64_t N) {
    return gfx_fftconv_workspace_bytes_ex(R, C_in, L, Lout, off, N, 0);
}

size_t gfx_fftconv_workspace_bytes_ex(int64_t R, int64_t C_in, int64_t L, int64_t Lout, int64_t off, int64_t N,
                                      int64_t part_len) {
    (void)L;
    (void)off;
    if (R <= 0 || N <= 0 || Lout <= 0) return 0;
    const ConvGeom g = conv_geom(N, Lout, part_len);
    if (!g.ok || g.nparts == 1 || g.ntiles == 1) return 0;  // one output tile: windows are transformed in place
    return (size_t)R * C_in * (g.ntiles + g.nparts - 1) * TILE_M * sizeof(float2);
}

int gfx_fir_spectrum_f32(const float* h, const float* gain, int64_t gain_div, void* Hs, int64_t RCf, int64_t N,
                         void* stream) {
    return gfx_fir_spectrum_ex_f32(h, gain, gain_div, Hs, RCf, N, 0, stream);
}

int gfx_fir_spectrum_ex_f32(const float* h, const float* gain, int64_t gain_div, void* Hs, int64_t RCf, int64_t N,
                            int64_t part_len, void* stream) {
    if (!h || !Hs || RCf <= 0 || N <= 0 || (gain && gain_div <= 0)) return GFX_EINVAL;
    const ConvGeom g = conv_geom(N, 1, part_len);
    if (!g.ok) return GFX_EINVAL;
    if (RCf * g.nparts > 0x7fffffffLL) return GFX_EINVAL;
    if (allow_lds(hspec_kernel<false>)) return GFX_ELAUNCH;
    const float2* tw = tile_twiddle_table((hipStream_t)stream);
    if (!tw) return GFX_ELAUNCH;
    const gfx_rowmap_t none = {1, 0, 0, 0};
    hipLaunchKernelGGL(hspec_kernel<false>, dim3((unsigned)(RCf * g.nparts)), dim3(TILE_T), TILE_LDS_BYTES,
                       (hipStream_t)stream, h, gain, gain_div, (float4*)Hs, N, (int)g.nparts, g.part_len, none, 1, tw);
    return hipGetLastError() == hipSuccess ? GFX_OK : GFX_ELAUNCH;
}

int gfx_fir_spectrum_rev_f32(const float* x, gfx_rowmap_t xmap, int64_t R, int64_t C, int64_t L, int64_t part_len,
                             void* Hs, void* stream) {
    if (!x || !Hs || R <= 0 || C <= 0 || L <= 0 || xmap.inner <= 0 || xmap.inner > 0x7fffffffLL) return GFX_EINVAL;
    const ConvGeom g = conv_geom(L, 1, part_len);
    if (!g.ok) return GFX_EINVAL;
    if (R * C * g.nparts > 0x7fffffffLL) return GFX_EINVAL;
    if (allow_lds(hspec_kernel<true>)) return GFX_ELAUNCH;
    const float2* tw = tile_twiddle_table((hipStream_t)stream);
    if (!tw) return GFX_ELAUNCH;
    hipLaunchKernelGGL(hspec_kernel<true>, dim3((unsigned)(R * C * g.nparts)), dim3(TILE_T), TILE_LDS_BYTES,
                       (hipStream_t)stream, x, (const float*)nullptr, (int64_t)1, (float4*)Hs, L, (int)g.nparts,
                       g.part_len, xmap, (int)C, tw);
    return hipGetLastError() == hipSuccess ? GFX_OK : GFX_ELAUNCH;
}

int gfx_fir_grad_f32(const float* x, gfx_rowmap_t xmap, const float* g, gfx_rowmap_t gmap, float* gh, int64_t R,
                     int64_t C_x, int64_t C_g, int64_t L, int64_t Lg, int64_t N, int64_t off, void* stream) {
    if (!x || !g || !gh || R <= 0 || L <= 0 || Lg <= 0 || N <= 0 || N > TILE_M + 1) return GFX_EINVAL;
    if (C_x < 1 || C_g < 1 || (C_x != C_g && C_x != 1 && C_g != 1)) return GFX_EINVAL;
    if (xmap.inner <= 0 || gmap.inner <= 0 || xmap.inner > 0x7fffffffLL || gmap.inner > 0x7fffffffLL) return GFX_EINVAL;
    if (off < -TILE_M || off > TILE_M) return GFX_EINVAL;  // |window start - tile start| stays within the 32-bit clip math
    CorrArgs a;
    a.xmap = xmap;
    a.gmap = gmap;
    a.L = L;
    a.Lg = Lg;
    a.off = off;
    a.N = N;
    a.V = TILE_F - (N & ~int64_t(1));  // tile slice: V + N - 1 <= 16384, V even
    a.ntiles = (L + a.V - 1) / a.V;
    a.Cx = (int)C_x;
    a.Cg = (int)C_g;
    a.Cout = (int)(C_x > C_g ? C_x : C_g);
    a.nblocks = R * a.Cout;
    if (a.nblocks > 0x7ffffff0LL) return GFX_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const float2* tw = tile_twiddle_table(st);
    if (!tw) return GFX_ELAUNCH;
    if (allow_lds(corr1_kernel)) return GFX_ELAUNCH;
    hipLaunchKernelGGL(corr1_kernel, dim3(pad8(a.nblocks)), dim3(TILE_T), TILE_LDS_BYTES, st, x, g, gh, a, tw);
    return hipGetLastError() == hipSuccess ? GFX_OK : GFX_ELAUNCH;
}

int gfx_fftconv_f32(const float* x, gfx_rowmap_t xmap, const void* Hs, float* y, gfx_rowmap_t ymap, int64_t R,
                    int64_t C_in, int64_t C_f, int64_t L, int64_t Lout, int64_t off, int64_t N, void* ws,
                    size_t ws_bytes, void* stream) {
    const gfx_rowmap_t none = {1, 0, 0, 0};
    return gfx_fftconv_ex_f32(x, xmap, Hs, R, 0, y, ymap, nullptr, none, R, C_in, C_f, L, Lout, off, N, ws, ws_bytes,
                              stream);
}

int gfx_fftconv_tee_f32(const float* x, gfx_rowmap_t xmap, const void* Hs, float* y, gfx_rowmap_t ymap, float* xcopy,
                        gfx_rowmap_t cmap, int64_t R, int64_t C_in, int64_t C_f, int64_t L, int64_t Lout, int64_t off,
                        int64_t N, void* ws, size_t ws_bytes, void* stream) {
    if (!xcopy) return GFX_EINVAL;
    return gfx_fftconv_ex_f32(x, xmap, Hs, R, 0, y, ymap, xcopy, cmap, R, C_in, C_f, L, Lout, off, N, ws, ws_bytes,
                              stream);
}

int gfx_fftconv_ex_f32(const float* x, gfx_rowmap_t xmap, const void* Hs, int64_t h_rows, int64_t part_len, float* y,
                       gfx_rowmap_t ymap, float* xcopy, gfx_rowmap_t cmap, int64_t R, int64_t C_in, int64_t C_f,
                       int64_t L, int64_t Lout, int64_t off, int64_t N, void* ws, size_t ws_bytes, void* stream) {
    return gfx_fftconv_sched_f32(x, xmap, Hs, h_rows, part_len, y, ymap, xcopy, cmap, R, C_in, C_f, L, Lout, off, N, ws,
                                 ws_bytes, GFX_SCHED_AUTO, stream);
}

// Ping-pong schedule: worth it once every CU gets a long stream of tiles (it runs ONE workgroup per CU, and a unit --
// a (row, channel) pair -- is walked serially by its workgroup).
static bool pp_applicable(int64_t R, int64_t h_rows, int64_t N, int64_t part_len) {
    return N <= TILE_M + 1 && part_len == 0 && R % h_rows == 0;
}

static int device_cus() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (!cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 0;
        cus[dev] = n;
    }
    return cus[dev];
}

int gfx_fftconv_sched_f32(const float* x, gfx_rowmap_t xmap, const void* Hs, int64_t h_rows, int64_t part_len, float* y,
                          gfx_rowmap_t ymap, float* xcopy, gfx_rowmap_t cmap, int64_t R, int64_t C_in, int64_t C_f,
                          int64_t L, int64_t Lout, int64_t off, int64_t N, void* ws, size_t ws_bytes, int schedule,
                          void* stream) {
    if (schedule != GFX_SCHED_AUTO && schedule != GFX_SCHED_TILE && schedule != GFX_SCHED_PINGPONG &&
        schedule != GFX_SCHED_HALFX && schedule != GFX_SCHED_WIDE)
        return GFX_EINVAL;
    if (!x || !Hs || !y || R <= 0 || L <= 0 || Lout <= 0 || N <= 0) return GFX_EINVAL;
    if (h_rows < 1 || h_rows > R || h_rows > 0x7fffffffLL) return GFX_EINVAL;
    if (xcopy && (off != 0 || Lout != L || C_in < C_f || N > TILE_M + 1 || cmap.inner <= 0 || cmap.inner > 0x7fffffffLL))
        return GFX_EINVAL;
    if (C_in < 1 || C_f < 1 || (C_in != C_f && C_in != 1 && C_f != 1)) return GFX_EINVAL;
    if (xmap.inner <= 0 || ymap.inner <= 0 || xmap.inner > 0x7fffffffLL || ymap.inner > 0x7fffffffLL) return GFX_EINVAL;
    const ConvGeom g = conv_geom(N, Lout, part_len);
    if (!g.ok) return GFX_EINVAL;
    ConvArgs a;
    a.xmap = xmap;
    a.ymap = ymap;
    a.cmap = cmap;
    a.hrows = (unsigned)h_rows;
    a.L = L;
    a.Lout = Lout;
    a.off = off;
    a.O = g.O;
    a.V = g.V;
    a.hop = g.hop;
    a.ntiles = g.ntiles;
    a.nparts = (int)g.nparts;
    a.Cin = (int)C_in;
    a.Cf = (int)C_f;
    a.Cout = (int)(C_in > C_f ? C_in : C_f);
    a.nblocks = R * a.Cout * g.ntiles;
    if (a.nblocks > 0x7ffffff0LL) return GFX_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const float2* tw = tile_twiddle_table(st);
    if (!tw) return GFX_ELAUNCH;

    if (schedule == GFX_SCHED_WIDE) {
        // 512-thread tile: the spectra are first rewritten into its thread layout (workspace: gfx_fftconv_wide_ws_bytes)
        const int64_t nf = h_rows * C_f;
        if (!wide::applicable(g, L, Lout, off, N) || part_len != 0 || !ws || ws_bytes < (size_t)nf * wide::W_H_F4 * 16 ||
            nf > 0x7fffffffLL)
            return GFX_EINVAL;
        const float2* tww = wide::tile512_twiddle_table(st);
        if (!tww) return GFX_ELAUNCH;
        const void* k = xcopy ? reinterpret_cast<const void*>(wide::fftconv1w_kernel<true>)
                              : reinterpret_cast<const void*>(wide::fftconv1w_kernel<false>);
        if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, wide::W_LDS_BYTES) != hipSuccess)
            return GFX_ELAUNCH;
        hipLaunchKernelGGL(wide::hconv_kernel, dim3((unsigned)nf), dim3(wide::WT), 0, st, (const float4*)Hs, (float4*)ws);
        if (xcopy)
            hipLaunchKernelGGL(wide::fftconv1w_kernel<true>, dim3(pad8(a.nblocks)), dim3(wide::WT), wide::W_LDS_BYTES, st, x,
                               (const float4*)ws, y, xcopy, a, tww);
        else
            hipLaunchKernelGGL(wide::fftconv1w_kernel<false>, dim3(pad8(a.nblocks)), dim3(wide::WT), wide::W_LDS_BYTES, st, x,
                               (const float4*)ws, y, (float*)gfx_dbg_stamp_buffer(), a, tww);
        return hipGetLastError() == hipSuccess ? GFX_OK : GFX_ELAUNCH;
    }
    if (schedule == GFX_SCHED_PINGPONG && !(g.nparts == 1 && pp_applicable(R, h_rows, N, part_len))) return GFX_EINVAL;
    if (g.nparts == 1 && schedule != GFX_SCHED_TILE && pp_applicable(R, h_rows, N, part_len)) {
        PPArgs pa;
        pa.a = a;
        pa.a.O = ((N - 1) + 511) & ~int64_t(511);
        pa.a.V = TILE_F - pa.a.O;
        pa.a.hop = pa.a.V;
        pa.a.ntiles = (Lout + pa.a.V - 1) / pa.a.V;
        const int64_t units = R * a.Cout;
        const int cus = device_cus();
        if (cus <= 0) return GFX_ELAUNCH;
        // AUTO never picks it on this hardware generation: measured on MI355X (profiles/r2/pingpong_ablation.md) the
        // one-tile-per-workgroup kernel is faster at every size (5.1 vs 7.0 ms at 8192 stereo rows), so the ping-pong
        // schedule runs only when asked for by name
        if (units <= 0x7fffffffLL / (pa.a.ntiles + 1) && schedule == GFX_SCHED_PINGPONG) {
            pa.units = (unsigned)units;
            pa.bh = (unsigned)(R / h_rows);
            const unsigned grid = (unsigned)(units < cus ? units : cus);
            const void* k = xcopy ? reinterpret_cast<const void*>(fftconv1pp_kernel<true>)
                                  : reinterpret_cast<const void*>(fftconv1pp_kernel<false>);
            if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_TOTAL) != hipSuccess)
                return GFX_ELAUNCH;
            if (xcopy)
                hipLaunchKernelGGL(fftconv1pp_kernel<true>, dim3(grid), dim3(PP_T), PP_LDS_TOTAL, st, x, (const float4*)Hs,
                                   y, xcopy, pa, tw);
            else
                hipLaunchKernelGGL(fftconv1pp_kernel<false>, dim3(grid), dim3(PP_T), PP_LDS_TOTAL, st, x, (const float4*)Hs,
                                   y, xcopy, pa, tw);
            return hipGetLastError() == hipSuccess ? GFX_OK : GFX_ELAUNCH;
        }
    }
    if (schedule == GFX_SCHED_HALFX && g.nparts != 1) return GFX_EINVAL;
    if (g.nparts == 1 && schedule == GFX_SCHED_HALFX) {
        const void* k = xcopy ? reinterpret_cast<const void*>(fftconv1h_kernel<true>)
                              : reinterpret_cast<const void*>(fftconv1h_kernel<false>);
        if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, HX_LDS_BYTES) != hipSuccess) return GFX_ELAUNCH;
        if (xcopy)
            hipLaunchKernelGGL(fftconv1h_kernel<true>, dim3(pad8(a.nblocks)), dim3(TILE_T), HX_LDS_BYTES, st, x,
                               (const float4*)Hs, y, xcopy, a, tw);
        else
            hipLaunchKernelGGL(fftconv1h_kernel<false>, dim3(pad8(a.nblocks)), dim3(TILE_T), HX_LDS_BYTES, st, x,
                               (const float4*)Hs, y, xcopy, a, tw);
        return hipGetLastError() == hipSuccess ? GFX_OK : GFX_ELAUNCH;
    }
    if (g.nparts == 1) {
        if (allow_lds(fftconv1_kernel<false>) || allow_lds(fftconv1_kernel<true>)) return GFX_ELAUNCH;
        if (xcopy)
            hipLaunchKernelGGL(fftconv1_kernel<true>, dim3(pad8(a.nblocks)), dim3(TILE_T), TILE_LDS_BYTES, st, x,
                               (const float4*)Hs, y, xcopy, a, tw);
        else
            hipLaunchKernelGGL(fftconv1_kernel<false>, dim3(pad8(a.nblocks)), dim3(TILE_T), TILE_LDS_BYTES, st, x,
                               (const float4*)Hs, y, (float*)gfx_dbg_stamp_buffer(), a, tw);
        return hipGetLastError() == hipSuccess ? GFX_OK : GFX_ELAUNCH;
    }
    if (g.ntiles == 1) {
        if (allow_lds(winmac_kernel)) return GFX_ELAUNCH;
        hipLaunchKernelGGL(winmac_kernel, dim3(pad8(a.nblocks)), dim3(TILE_T), TILE_LDS_BYTES, st, x, (const float4*)Hs, y,
                           a, tw);
        return hipGetLastError() == hipSuccess ? GFX_OK : GFX_ELAUNCH;
    }
    const int64_t nwin = g.ntiles + g.nparts - 1;
    const size_t need = (size_t)R * C_in * nwin * TILE_M * sizeof(float2);
    if (!ws || ws_bytes < need) return GFX_ENOSPC;
    if (allow_lds(xspec_kernel) || allow_lds(macinv_kernel)) return GFX_ELAUNCH;
    ConvArgs ax = a;
    ax.nblocks = R * C_in * nwin;
    if (ax.nblocks > 0x7ffffff0LL) return GFX_EINVAL;
    hipLaunchKernelGGL(xspec_kernel, dim3(pad8(ax.nblocks)), dim3(TILE_T), TILE_LDS_BYTES, st, x, (float2*)ws, ax,
                       nwin, tw);
    hipLaunchKernelGGL(macinv_kernel, dim3(pad8(a.nblocks)), dim3(TILE_T), TILE_LDS_BYTES, st, (const float2*)ws,
                       (const float4*)Hs, y, a, nwin, tw);
    return hipGetLastError() == hipSuccess ? GFX_OK : GFX_ELAUNCH;
}

}  // extern "C"
