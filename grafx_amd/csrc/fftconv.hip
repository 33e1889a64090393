// Overlap-save FIR convolution on LDS FFT tiles (gfx950).
//
// Replaces the reference's convolve() — core/convolution.py:119-134 — for the case where
// it equals a true linear convolution (P = L + N - 1 even; see DESIGN.md for the odd-P quirk,
// which is layered on top of the *full* convolution this file produces).
//
// Kernels
//   hspec_kernel    taps -> per-partition tile spectra (He, Ho pairs in thread layout)
//   fftconv1_kernel N <= 8193: load x tile -> FFT -> x H -> IFFT -> store valid samples (fused)
//   xspec_kernel    N  > 8193: x window -> FFT -> split spectra (Xe, Xo pairs in thread layout) to HBM/L2
//   macinv_pair_kernel  N > 8193: two output tiles per 512-thread workgroup, sum_p X[i-p] * H[p] in registers -> 2 x IFFT -> store
//   macinv_kernel   the same with one output tile per 256-thread workgroup (GFX_SCHED_TILE)
//   winmac_kernel / corr1_kernel / gfx_fftconv_pipe_*, gfx_corr_pipe (generated assembly): see below
//
// Algorithmic bytes: 4*(C_in + C_out) per output frame per row (read x once, write y once);
// the tile overlap (N-1 of 16384 samples) is re-read through L2.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

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


// max |y| over what store_valid has just stored of this tile (positions O <= q < room of the tile), into rowmax[rco]: the
// by-product for the odd-length aliasing's pair scaling (round 6; gfx_fftconv_rowmax_f32).  Non-negative floats order like
// their bit patterns: a wave reduction and one atomic maximum per wave and tile.
__device__ __forceinline__ void tile_rowmax(const cx (&v)[32], uint32_t* __restrict__ rowmax, unsigned rco, int64_t n0, int64_t O,
                                            int64_t Lout, int t) {
    const int64_t room = Lout - (n0 - O);
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const int64_t q = 2 * (t + 256 * k);
        const cx e = v[brev(k, 5)];
        const uint32_t ex = __float_as_uint(e.x) & 0x7fffffffu, ey = __float_as_uint(e.y) & 0x7fffffffu;
        if (q >= O && q < room) m = ex > m ? ex : m;
        if (q >= O && q + 1 < room) m = ey > m ? ey : m;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const uint32_t u = (uint32_t)__shfl_xor((int)m, o);
        m = u > m ? u : m;
    }
    if ((t & 63) == 0 && m) atomicMax(rowmax + rco, m);
}

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
                                                             ConvArgs a, const float2* __restrict__ twtab,
                                                             uint32_t* __restrict__ rowmax = nullptr) {
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

    // Every global load of the tile is issued up front: the window, the twiddles, and the filter spectrum
    // (needed only after the forward transform, by which time it has long arrived).  Left to itself the
    // compiler issues each spectrum load right before its use and waits for it: 16 serialised L2 round trips.
    TileTw tw;
    cx v[32], w[2][16];
    f4v hreg[H_SLOTS];
    load_window(v, xrow, a.off + tile * a.V - a.O, a.L, t, 1.0f);
    tile_twiddles(tw, twtab, t);
#pragma unroll
    for (int q = 0; q < H_SLOTS; ++q) hreg[q] = buf_load_f4(H, 16u * (uint32_t)t, 4096u * q);
    __builtin_amdgcn_sched_barrier(0);
    // off == 0 here: the window's valid part is x[tile*V, tile*V + V) itself
    if (TEE) store_valid<true>(v, xcopy + row_off(a.cmap, r, c), tile * a.V, a.O, a.L, t);
    tile_forward(v, w, tw, lds, t);
    for_each_pair(t, tw.base(), [&](int slot, int ia, int ib, cx wk, bool self) {
        cx xe, xo, ye, yo, za, zb;
        pair_split(NAT(w, ia), NAT(w, ib), xe, xo);
        pair_product(xe, xo, hreg[slot], wk, ye, yo);
        pair_merge(ye, yo, za, zb);
        NAT(w, ia) = za;
        if (!self) NAT(w, ib) = zb;
    });
    // no barrier here: the inverse starts by writing S2 rows j = t and 512 - t, the very rows (and the only rows)
    // this thread read at the end of the forward transform -- nobody else touches them in between
    tile_inverse(w, v, tw, lds, t);
    store_valid(v, yrow, tile * a.V, a.O, a.Lout, t);
    if (rowmax) tile_rowmax(v, rowmax, rco, tile * a.V, a.O, a.Lout, t);   // (see tile_rowmax)
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
    // stored the way the filter spectra are: per mirrored bin pair one 16-byte entry (Xe, Xo) at [slot][t] (the seventeenth
    // slot: thread 0 only), already split -- the product kernels fetch a pair with ONE instruction where two 8-byte rows
    // cost the CU's address unit twice as much (a vector-memory instruction occupies it ~22 cycles whatever its width, and
    // that is what bounds their loop), and thread 0's different pairing is settled here, once, instead of in every turn.
    f4v* out = reinterpret_cast<f4v*>(Zs) + (int64_t)lb * H_TILE_F4;
    for_each_pair(t, tw.base(), [&](int slot, int ia, int ib, cx, bool) {
        cx xe, xo;
        pair_split(NAT(w, ia), NAT(w, ib), xe, xo);
        // (streamed: the product kernel starts after ALL windows are written, by when only the last tenth is still in
        // a cache -- cfg3 2.05 -> 2.02 ms with the product kernel)
        __builtin_nontemporal_store(__builtin_shufflevector(xe, xo, 0, 1, 2, 3), &out[slot * TILE_T + t]);
    });
}

__global__ __launch_bounds__(TILE_T, 2) void macinv_kernel(const float2* __restrict__ Zs, const float4* __restrict__ Hs,
                                                           float* __restrict__ y, ConvArgs a, int64_t nwin,
                                                           const float2* __restrict__ twtab,
                                                           uint32_t* __restrict__ rowmax = nullptr) {
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
    const f4v* Z = reinterpret_cast<const f4v*>(Zs) + ((int64_t)r * a.Cin + (a.Cin == 1 ? 0 : c)) * nwin * H_TILE_F4;

    cx ye[H_SLOTS], yo[H_SLOTS];
#pragma unroll
    for (int s = 0; s < H_SLOTS; ++s) ye[s] = yo[s] = cx{0.0f, 0.0f};
    const cx wj = to_cx(twtab[TILE_T + t]);  // W_8192^t

    for (int p = 0; p < a.nparts; ++p) {
        const int64_t j = tile - p;
        if (!window_live(a.off - a.O + j * a.hop, a.L)) continue;
        // every operand of the turn requested up front (left to itself the compiler fetches each pair right before its
        // product and waits for it: 34 serialised trips to L2 per turn, 3.8 instead of 1.9 ms at cfg3)
        const rsrc_t zr = make_rsrc(Z + (j + a.nparts - 1) * H_TILE_F4, (int64_t)H_TILE_F4 * 16);
        const rsrc_t hr = make_rsrc(H + (int64_t)p * H_TILE_F4, (int64_t)H_TILE_F4 * 16);
        f4v xreg[H_SLOTS], hreg[H_SLOTS];
#pragma unroll
        for (int q = 0; q < H_SLOTS; ++q) {
            xreg[q] = buf_load_f4(zr, 16u * (uint32_t)t, 4096u * q);
            hreg[q] = buf_load_f4(hr, 16u * (uint32_t)t, 4096u * q);
        }
        __builtin_amdgcn_sched_barrier(0);
        for_each_pair(t, wj, [&](int slot, int, int, cx wk, bool) {
            pair_product_acc(xreg[slot].lo, xreg[slot].hi, hreg[slot], wk, ye[slot], yo[slot]);
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
    if (rowmax) tile_rowmax(v, rowmax, rco, tile * a.V, a.O, a.Lout, t);
}

// ---- two consecutive output tiles per 512-thread workgroup ---------------------------------------------------------
// Output tiles i and i + 1 of a row meet the windows i + 1, i, ..., i + 1 - nparts; window i + 1 - k multiplies partition k
// for tile i + 1 and partition k - 1 for tile i.  One workgroup of two 256-thread groups walks those nparts + 1 windows:
// per turn it fetches ONE window spectrum and ONE filter partition (the other partition is the one of the turn before, kept
// in registers), i.e. (nparts + 1) + nparts tile-sized operands per tile PAIR where macinv_kernel fetches 2 nparts per tile
// (60 001 taps: 17 against 32) -- the loop is bound by what the CUs' L1 can take in, not by arithmetic.
// Registers decide the shape: two accumulator sets, a window and two partitions are 340 registers per thread for a
// 256-thread tile (round 4's first attempt, tools/experiments/r4_macinv2: spills, serialised loads, 1.6x slower).  Here
// the mirrored bin pairs of a thread are split between the two groups -- group G multiplies pairs 8G .. 8G + 7 of BOTH
// tiles -- so everything halves: 2 x 36 accumulators, 2 x 36 for two window sets (the running one and the next), 2 x 36
// for the partitions.  After the loop the groups swap halves through LDS (group 0 gets tile i complete, group 1 tile i + 1) and run
// the two inverse transforms side by side.  Every tile adds the same products in the same order as in macinv_kernel (equal
// to the last bit or two: the compiler contracts the twiddle arithmetic of the two kernels differently).
//
// Mirrored pairs of a group, the same straight-line code for every lane: slot i < 8 of group G is pair 8 G + i of threads
// t != 0 -- bins (row 0, k3) and (row 1, 15 - k3), k3 = 8 G + i, rows = the two butterflies bf of tile_forward's layout --
// and, for thread 0, whose bins pair up INSIDE a row, (row 0: k3 = i with 16 - i; i = 0 with itself) in group 0 and
// (row 1: k3 = i with 15 - i) in group 1; group 0 has a ninth slot for thread 0's self-paired bin (row 0, k3 = 8).  These
// are for_each_pair's slots, in which both the filter partitions (hspec_kernel) and the window spectra (xspec_kernel) are
// stored, so the difference is data, not control flow: a per-lane slot base in the load offset (thread 0 of group 1:
// slots 9.. instead of 8..), W^(t + 512 k3) from a per-lane base (thread 0 of group 1: W_32^17, so that base x
// W_32^(16 + 2 i) = W_32^(1 + 2 i), its pairs' factor), and selects where the finished halves are put into place.  A
// divergent branch for thread 0 would double the code and, worse, blur the compiler's wait counts at its join (every turn
// would start by waiting for all outstanding loads).
//
// Flat 16-entry layout of a group's half tile: e < 8 "lower" = row G, k3 = e;  e >= 8 "upper" = k3 = e of row
// (G == 0) == (t != 0).
template <int G>
__device__ __forceinline__ void macinv_pair_half(const f4v* __restrict__ Z, const f4v* __restrict__ H, const ConvArgs& a,
                                                 int64_t tile, bool two, int t, cx wj, cx* lds_other,
                                                 cx (&own)[16]) {
    constexpr int NS = G ? 8 : 9;
    cx aye[NS], ayo[NS], bye[NS], byo[NS];   // tile i ("a") and tile i + 1 ("b")
#pragma unroll
    for (int s = 0; s < NS; ++s) aye[s] = ayo[s] = bye[s] = byo[s] = cx{0.0f, 0.0f};

    const bool z = t == 0;
    // window and partition slots alike (16-byte entries, slot * 256 + t): group 1 multiplies slots 8.. (thread 0: 9..)
    const uint32_t hv = 16u * (uint32_t)(t + (G ? (z ? 9 : 8) * TILE_T : 0));
    const cx w32_17 = {-0.98078528040323044913f, 0.19509032201612826785f};
    const cx wbase = (G && z) ? w32_17 : wj;

    // turns k_lo .. k_hi: the windows tile + 1 - k that overlap the signal (window starts fall with k) and that one of the
    // two tiles uses (turn 0: tile i + 1 only, turn nparts: tile i only)
    const int64_t s0 = a.off - a.O + (tile + 1) * a.hop;                  // start of the window of turn 0
    int k_lo = two ? 0 : 1, k_hi = a.nparts;
#ifdef GFX_PAIR_NOLOOP
    k_hi = -1;   // timing experiment: the walk skipped, the rest of the kernel as it is
#endif
    if (s0 >= a.L) k_lo = max(k_lo, (int)((s0 - a.L) / a.hop) + 1);       // first k with s0 - k hop < L
    if (s0 + TILE_F <= (int64_t)k_hi * a.hop) k_hi = (int)((s0 + TILE_F - 1) / a.hop);   // last k with s0 - k hop + TILE_F > 0

    // Operands come through two range-checked descriptors: the nparts partitions of this row's filter (partition k at
    // k * 68 KB: a request for partition nparts -- the last turn has none for tile i + 1 -- falls outside and reads as
    // zeros), and one window spectrum at a time (an empty descriptor for the turn after the last).  So every turn is
    // the same straight-line code, with no test around a load or a product.
    // (Tried and dropped: the turns in a cyclic order that makes all pairs of a row read the same window in the same turn
    // -- the L2 hit rate was not what held the loop back: 2.36 against 2.25 ms with xspec at cfg3, all nparts + 1 turns
    // taken by every pair.)
    const rsrc_t hr = make_rsrc(H, (int64_t)a.nparts * H_TILE_F4 * 16);
    auto window = [&](int k) {
        return make_rsrc(Z + (tile + 1 - k + a.nparts - 1) * H_TILE_F4, k <= k_hi ? (int64_t)H_TILE_F4 * 16 : 0);
    };
    // One turn: the window of turn k (w) times partition k - 1 (ha) for tile i, then times partition k (hb) for tile i + 1.
    // The operands of turn k + 1 are requested a full turn ahead, in two bursts: its window into the second window
    // register set at the start of the turn, its NEW partition (k + 1, tile i + 1's) into the registers of ha once all of
    // tile i's products are done -- so the next turn runs with (ha, hb) and the two window sets exchanged.  (Requests
    // slot by slot IN PLACE -- a slot's next operands into its own registers right after its products, no second window
    // set -- measured the same to 1.5 %: the loop is bound by what the address unit moves, not by when it is asked.)
    if (k_lo <= k_hi) {
        f4v w0[NS], w1[NS], h0[NS], h1[NS];
        auto req_w = [&](f4v (&w)[NS], int k) {
            const rsrc_t zr = window(k);
#pragma unroll
            for (int i = 0; i < NS; ++i) w[i] = buf_load_f4(zr, hv, 4096u * i);
        };
        auto req_h = [&](f4v (&h)[NS], int k) {   // k = -1 / nparts: outside the descriptor, zeros
            const uint32_t hoff = k < 0 ? 0x40000000u : (uint32_t)k * (uint32_t)(H_TILE_F4 * 16);
#pragma unroll
            for (int i = 0; i < NS; ++i) h[i] = buf_load_f4(hr, hv, hoff + 4096u * i);
        };
        auto bturn = [&](int k, f4v (&w)[NS], f4v (&wn)[NS], f4v (&ha)[NS], f4v (&hb)[NS]) {
            req_w(wn, k + 1);
            __builtin_amdgcn_sched_barrier(0);
            cx wjt = wbase;
            asm volatile("" : "+v"(wjt));   // the slots' W^(t + 512 k3) recomputed every turn (2 instructions each), not kept: 16 registers
#pragma unroll
            for (int i = 0; i < NS; ++i)
                pair_product_acc(w[i].lo, w[i].hi, ha[i], mul_w16(wjt, 8 * G + i), aye[i], ayo[i]);
            __builtin_amdgcn_sched_barrier(0);
            req_h(ha, k + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NS; ++i)
                pair_product_acc(w[i].lo, w[i].hi, hb[i], mul_w16(wjt, 8 * G + i), bye[i], byo[i]);
            __builtin_amdgcn_sched_barrier(0);
        };
        req_h(h0, k_lo - 1);
        req_w(w0, k_lo);
        req_h(h1, k_lo);
        __builtin_amdgcn_sched_barrier(0);
        for (int k = k_lo; k <= k_hi; k += 2) {
            bturn(k, w0, w1, h0, h1);
            if (k + 1 <= k_hi) bturn(k + 1, w1, w0, h1, h0);
        }
    }

    // the finished half tiles in the flat layout: this group's half of ITS tile stays in registers (own), its half of the
    // other group's tile goes through LDS
    auto place = [&](const cx (&ye)[NS], const cx (&yo)[NS], cx (&flat)[16]) {
        cx za[NS], zb[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) pair_merge(ye[i], yo[i], za[i], zb[i]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (G == 0) {
                flat[j] = za[j];
                flat[8 + j] = z ? (j == 0 ? za[NS - 1] : zb[8 - j]) : zb[7 - j];
            } else {
                flat[j] = z ? za[j] : zb[7 - j];
                flat[8 + j] = z ? zb[7 - j] : za[j];
            }
        }
    };
    cx other[16];
    if (G == 0) {
        place(aye, ayo, own);
        place(bye, byo, other);
    } else {
        place(bye, byo, own);
        place(aye, ayo, other);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) lds_other[e * TILE_T + t] = other[e];
}

// own / got: the flat rows of this group's pairs and of the other group's (see above) -> the [2][16] layout tile_inverse takes
template <int G>
__device__ __forceinline__ void macinv_pair_gather(const cx (&own)[16], const cx* lds_mine, int t, cx (&pz)[2][16]) {
    cx got[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) got[e] = lds_mine[e * TILE_T + t];
    const bool z = t == 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        pz[G][brev(i, 4)] = own[i];
        pz[1 - G][brev(i, 4)] = got[i];
        // upper halves: own is row (G == 0) == (t != 0), got the other one
        const cx up_own = own[8 + i], up_got = got[8 + i];
        const bool own_is_row1 = (G == 0) != z;
        pz[1][brev(8 + i, 4)] = own_is_row1 ? up_own : up_got;
        pz[0][brev(8 + i, 4)] = own_is_row1 ? up_got : up_own;
    }
}

__global__ __launch_bounds__(2 * TILE_T, 1) void macinv_pair_kernel(const float2* __restrict__ Zs,
                                                                    const float4* __restrict__ Hs,
                                                                    float* __restrict__ y, ConvArgs a, int64_t nwin,
                                                                    const float2* __restrict__ twtab,
                                                                    uint32_t* __restrict__ rowmax = nullptr) {
    extern __shared__ __attribute__((aligned(16))) cx lds[];
    const int t = threadIdx.x & (TILE_T - 1);
    const int grp = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
    const unsigned lb = xcd_logical_block();
    if (lb >= (unsigned)a.nblocks) return;
    const unsigned npairs = (unsigned)((a.ntiles + 1) >> 1);
    const unsigned rco = lb / npairs;
    const int64_t tile = 2 * (int64_t)(lb - rco * npairs);
    const bool two = tile + 1 < a.ntiles;
    const unsigned r = rco / (unsigned)a.Cout;
    const int c = (int)(rco - r * (unsigned)a.Cout);
    float* yrow = y + row_off(a.ymap, r, c);
    const f4v* H = reinterpret_cast<const f4v*>(Hs) + ((int64_t)(r % a.hrows) * a.Cf + (a.Cf == 1 ? 0 : c)) * a.nparts * H_TILE_F4;
    const f4v* Z = reinterpret_cast<const f4v*>(Zs) + ((int64_t)r * a.Cin + (a.Cin == 1 ? 0 : c)) * nwin * H_TILE_F4;
    const cx wj = to_cx(twtab[TILE_T + t]);  // W_8192^t
    cx* lds_mine = lds + grp * TILE_LDS_F2;
    cx* lds_other = lds + (1 - grp) * TILE_LDS_F2;

    cx own[16], pz[2][16], v[32];
    if (grp == 0) macinv_pair_half<0>(Z, H, a, tile, two, t, wj, lds_other, own);
    else macinv_pair_half<1>(Z, H, a, tile, two, t, wj, lds_other, own);
    __syncthreads();
    if (grp == 0) macinv_pair_gather<0>(own, lds_mine, t, pz);
    else macinv_pair_gather<1>(own, lds_mine, t, pz);
    __syncthreads();
    TileTw tw;
    tile_twiddles(tw, twtab, t);
    tile_inverse(pz, v, tw, lds_mine, t);
    if (grp == 0 || two) store_valid(v, yrow, (tile + grp) * a.V, a.O, a.Lout, t);
    if (rowmax && (grp == 0 || two)) tile_rowmax(v, rowmax, rco, (tile + grp) * a.V, a.O, a.Lout, t);
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


// ---- the hand-scheduled persistent kernels (csrc/asm/gen_fftconv_pipe.py) ---------------------------------------
// Assembled at build time into one gfx950 code object that is embedded here and loaded per device on first use
// (hipModuleLoadData: no file on disk, nothing to ship besides the library).
}  // namespace gfx
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "fftconv_pipe_args.inc"
#include "fftconv_pipe_hsaco.inc"
namespace gfx {

static inline unsigned pad8(int64_t n) { return (unsigned)(((n + 7) / 8) * 8); }

struct PipeModule {
    hipModule_t mod = nullptr;
    hipFunction_t fn[sizeof(kPipeVariants) / sizeof(kPipeVariants[0])] = {};
    hipFunction_t corr = nullptr;
    int cus = 0;
    bool tried = false, ok = false;
};

static PipeModule* pipe_module() {
    static std::mutex mu;
    static PipeModule mods[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    PipeModule& m = mods[dev];
    if (!m.tried) {
        m.tried = true;
        // GRAFX_PIPE_HSACO=<file>: load another build of the same kernels instead of the embedded one (schedule tuning:
        // `python -m grafx_amd.csrc.asm.gen_fftconv_pipe --hsaco f.hsaco <knobs>` needs no rebuild of the library)
        const char* alt = getenv("GRAFX_PIPE_HSACO");
        bool ok = (alt && *alt ? hipModuleLoad(&m.mod, alt) : hipModuleLoadData(&m.mod, kPipeCodeObject)) == hipSuccess;
        for (size_t i = 0; ok && i < sizeof(kPipeVariants) / sizeof(kPipeVariants[0]); ++i)
            ok = hipModuleGetFunction(&m.fn[i], m.mod, kPipeVariants[i].name) == hipSuccess;
        ok = ok && hipModuleGetFunction(&m.corr, m.mod, kCorrKernelName) == hipSuccess;
        ok = ok && hipDeviceGetAttribute(&m.cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && m.cus >= 8;
        m.ok = ok;
        (void)hipGetLastError();
    }
    return m.ok ? &m : nullptr;
}

// q_est = mulhi(n, m) >> sh is floor(n / d) or one less for every 32-bit n (the kernel corrects by one)
static inline void pipe_magic(uint32_t d, uint32_t& m, uint32_t& sh) {
    sh = 31u - (uint32_t)__builtin_clz(d);
    const unsigned __int128 q = ((unsigned __int128)1 << (32 + sh)) / d;
    m = q > 0xffffffffu ? 0xffffffffu : (uint32_t)q;
}

// What the persistent kernels cover: one partition, no output offset, even row lengths (stores are whole sample pairs),
// the same batch-major row grouping on every tensor, byte strides inside 32 bits, and an overlap the build has a variant
// for.  Everything else runs on fftconv1_kernel.
static int pipe_variant(const ConvArgs& a, const ConvGeom& g, bool tee, int64_t N) {
    if (g.nparts != 1 || a.off != 0 || (a.Lout & 1) || (tee && ((a.L & 1) || a.Lout != a.L)) || N > TILE_M + 1) return -1;
    if (a.L * 4 >= (int64_t(1) << 30) || a.Lout * 4 >= (int64_t(1) << 30)) return -1;
    if (a.xmap.inner != a.ymap.inner || (tee && a.cmap.inner != a.xmap.inner)) return -1;
    auto fits = [](const gfx_rowmap_t& mp) {
        return mp.stride_inner >= 0 && mp.stride_ch >= 0 && mp.stride_outer >= 0 && mp.stride_inner * 4 < (int64_t(1) << 32) &&
               mp.stride_ch * 4 < (int64_t(1) << 32);
    };
    if (!fits(a.xmap) || !fits(a.ymap) || (tee && !fits(a.cmap))) return -1;
    for (size_t i = 0; i < sizeof(kPipeVariants) / sizeof(kPipeVariants[0]); ++i)
        if (kPipeVariants[i].tee == tee && (int64_t)kPipeVariants[i].a_lo * 512 == a.O) return (int)i;
    return -1;
}

// `rowmax` (nullable): R * Cout zeroed words that receive the bits of max |y| of every output row-channel (see the generator:
// rowmax_accumulate / rowmax_flush); passed to the kernel as a ready buffer descriptor, all zeros when not wanted
static int launch_pipe(PipeModule* pm, int variant, const float* x, const void* Hs, float* y, float* xcopy,
                       const ConvArgs& a, const float2* tw, hipStream_t st, uint32_t* rowmax = nullptr, int64_t rowmax_words = 0) {
    PipeKernArgs k;
    memset(&k, 0, sizeof(k));
    if (rowmax) {
        k.rm_lo = (uint32_t)reinterpret_cast<uint64_t>(rowmax);
        k.rm_hi = (uint32_t)(reinterpret_cast<uint64_t>(rowmax) >> 32) & 0xffffu;
        k.rm_rec = (uint32_t)(rowmax_words * 4);
        k.rm_flags = 0x00020000u;
    }
    auto lo = [](const void* p) { return (uint32_t)reinterpret_cast<uint64_t>(p); };
    auto hi = [](const void* p) { return (uint32_t)(reinterpret_cast<uint64_t>(p) >> 32); };
    k.x_lo = lo(x); k.x_hi = hi(x); k.h_lo = lo(Hs); k.h_hi = hi(Hs); k.y_lo = lo(y); k.y_hi = hi(y);
    k.c_lo = lo(xcopy); k.c_hi = hi(xcopy); k.tw_lo = lo(tw); k.tw_hi = hi(tw);
    k.L_bytes = (uint32_t)(a.L * 4); k.Lout_bytes = (uint32_t)(a.Lout * 4);
    k.V_bytes = (uint32_t)(a.V * 4); k.O_bytes = (uint32_t)(a.O * 4);
    k.ntiles = (uint32_t)a.ntiles; k.nblocks = (uint32_t)a.nblocks;
    pipe_magic(k.ntiles, k.m_ntiles, k.sh_ntiles);
    k.inner = (uint32_t)a.xmap.inner;
    pipe_magic(k.inner, k.m_inner, k.sh_inner);
    k.hrows = a.hrows;
    pipe_magic(k.hrows, k.m_hrows, k.sh_hrows);
    k.cout_shift = a.Cout == 2 ? 1 : 0; k.cout_mask = a.Cout == 2 ? 1 : 0;
    k.cin_mask = a.Cin == 2 ? 1 : 0; k.cf_mask = a.Cf == 2 ? 1 : 0; k.Cf = (uint32_t)a.Cf;
    const unsigned grid = (unsigned)(2 * pm->cus) & ~7u;     // two resident workgroups per CU
    k.per_xcd = (uint32_t)((a.nblocks + grid - 1) / grid);   // tiles per workgroup: consecutive runs (the overlap is carried in registers)
    k.wgs_per_xcd = grid / 8;
    auto strides = [](const gfx_rowmap_t& mp, uint32_t& olo, uint32_t& ohi, uint32_t& in, uint32_t& ch) {
        const uint64_t o = (uint64_t)mp.stride_outer * 4;
        olo = (uint32_t)o; ohi = (uint32_t)(o >> 32); in = (uint32_t)(mp.stride_inner * 4); ch = (uint32_t)(mp.stride_ch * 4);
    };
    strides(a.xmap, k.xs_outer_lo, k.xs_outer_hi, k.xs_inner, k.xs_ch);
    strides(a.ymap, k.ys_outer_lo, k.ys_outer_hi, k.ys_inner, k.ys_ch);
    if (xcopy) strides(a.cmap, k.cs_outer_lo, k.cs_outer_hi, k.cs_inner, k.cs_ch);
    size_t size = sizeof(k);
    void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &k, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    // GRAFX_PIPE_THREADS: block size for an experimental code object loaded through GRAFX_PIPE_HSACO (timing experiments
    // with other tile shapes, tools/experiments/r4_tile16); the shipped kernels are 256-thread tiles
    // -- honoured ONLY together with GRAFX_PIPE_HSACO: the embedded code object never runs with another block size
    static const unsigned threads = [] {
        const char* alt = getenv("GRAFX_PIPE_HSACO");
        const char* e = getenv("GRAFX_PIPE_THREADS");
        const int n = (alt && *alt && e) ? atoi(e) : 0;
        return n == 512 ? 512u : (unsigned)TILE_T;
    }();
    return hipModuleLaunchKernel(pm->fn[variant], grid, 1, 1, threads, 1, 1, 0, st, nullptr, config) == hipSuccess ? GFX_OK
                                                                                                               : GFX_ELAUNCH;
}

// The hand-scheduled form of corr1_kernel (csrc/asm/gen_corr_pipe.py) covers off = 0 with the same row grouping on x and
// g and byte strides inside 32 bits -- the equaliser's filter gradient; GRAFX_FFTCONV_SCHED=tile keeps corr1_kernel.
static bool corr_pipe_ok(const CorrArgs& a) {
    if (a.off != 0 || a.N > TILE_M + 1 || a.xmap.inner != a.gmap.inner) return false;
    if (a.L * 4 >= (int64_t(1) << 30) || a.Lg * 4 >= (int64_t(1) << 30)) return false;
    auto fits = [](const gfx_rowmap_t& mp) {
        return mp.stride_inner >= 0 && mp.stride_ch >= 0 && mp.stride_outer >= 0 && mp.stride_inner * 4 < (int64_t(1) << 32) &&
               mp.stride_ch * 4 < (int64_t(1) << 32);
    };
    return fits(a.xmap) && fits(a.gmap);
}

static int launch_corr_pipe(PipeModule* pm, const float* x, const float* g, float* gh, const CorrArgs& a, const float2* tw,
                            hipStream_t st) {
    CorrKernArgs k;
    memset(&k, 0, sizeof(k));
    auto lo = [](const void* p) { return (uint32_t)reinterpret_cast<uint64_t>(p); };
    auto hi = [](const void* p) { return (uint32_t)(reinterpret_cast<uint64_t>(p) >> 32); };
    k.x_lo = lo(x); k.x_hi = hi(x); k.g_lo = lo(g); k.g_hi = hi(g); k.o_lo = lo(gh); k.o_hi = hi(gh);
    k.tw_lo = lo(tw); k.tw_hi = hi(tw);
    k.L_bytes = (uint32_t)(a.L * 4); k.Lg_bytes = (uint32_t)(a.Lg * 4); k.V_bytes = (uint32_t)(a.V * 4);
    k.N_even_bytes = (uint32_t)((a.N & ~int64_t(1)) * 4);
    k.ntiles = (uint32_t)a.ntiles; k.nblocks = (uint32_t)a.nblocks;
    k.inner = (uint32_t)a.xmap.inner;
    pipe_magic(k.inner, k.m_inner, k.sh_inner);
    k.cout_shift = a.Cout == 2 ? 1 : 0; k.cout_mask = a.Cout == 2 ? 1 : 0;
    k.cx_mask = a.Cx == 2 ? 1 : 0; k.cg_mask = a.Cg == 2 ? 1 : 0;
    k.out_row_bytes = (uint32_t)(a.N * 4);
    if (a.N & 1) {       // the last lag is half of a sample pair: stored on its own (row, lane, row offset of that pair)
        const uint32_t m = (uint32_t)((a.N - 1) / 2);
        k.tail_row = m >> 8; k.tail_lane = m & 255; k.tail_off = 2048u * (m >> 8);
    } else {
        k.tail_row = 255;
    }
    const float sc = 1.0f / (4.0f * TILE_M);
    memcpy(&k.scale, &sc, 4);
    const unsigned grid = pad8(a.nblocks);
    k.pad0 = grid / 8;
    auto strides = [](const gfx_rowmap_t& mp, uint32_t& olo, uint32_t& ohi, uint32_t& in, uint32_t& ch) {
        const uint64_t o = (uint64_t)mp.stride_outer * 4;
        olo = (uint32_t)o; ohi = (uint32_t)(o >> 32); in = (uint32_t)(mp.stride_inner * 4); ch = (uint32_t)(mp.stride_ch * 4);
    };
    strides(a.xmap, k.xs_outer_lo, k.xs_outer_hi, k.xs_inner, k.xs_ch);
    strides(a.gmap, k.gs_outer_lo, k.gs_outer_hi, k.gs_inner, k.gs_ch);
    size_t size = sizeof(k);
    void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &k, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    return hipModuleLaunchKernel(pm->corr, grid, 1, 1, TILE_T, 1, 1, 0, st, nullptr, config) == hipSuccess ? GFX_OK : GFX_ELAUNCH;
}

// Whether GFX_SCHED_AUTO may pick the persistent kernel for large launches (decided by measurement)
#ifndef GFX_PIPE_AUTO
#define GFX_PIPE_AUTO 1
#endif
// GFX_SCHED_AUTO: which of the two kernels a launch gets.  GRAFX_FFTCONV_SCHED=tile|pipe overrides (A/B measurements).
static int auto_schedule() {
    static int cached = -1;
    if (cached < 0) {
        const char* e = getenv("GRAFX_FFTCONV_SCHED");
        cached = e && !strcmp(e, "pipe") ? GFX_SCHED_PIPE : (e && !strcmp(e, "tile") ? GFX_SCHED_TILE : GFX_SCHED_AUTO);
    }
    return cached;
}


static thread_local const char* t_last_kernel = "";   // see gfx_fftconv_last_kernel

template <typename K>
static int allow_lds(K kernel, int bytes = TILE_LDS_BYTES) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               bytes) == hipSuccess
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
    return (size_t)R * C_in * (g.ntiles + g.nparts - 1) * H_TILE_F4 * sizeof(float4);
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
    PipeModule* pm = pipe_module();
    if (pm && corr_pipe_ok(a) && auto_schedule() != GFX_SCHED_TILE) return launch_corr_pipe(pm, x, g, gh, a, tw, st);
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


static int fftconv_sched(const float* x, gfx_rowmap_t xmap, const void* Hs, int64_t h_rows, int64_t part_len, float* y,
                         gfx_rowmap_t ymap, float* xcopy, gfx_rowmap_t cmap, int64_t R, int64_t C_in, int64_t C_f,
                         int64_t L, int64_t Lout, int64_t off, int64_t N, void* ws, size_t ws_bytes, int schedule,
                         void* stream, uint32_t* rowmax, int* rowmax_written);

int gfx_fftconv_sched_f32(const float* x, gfx_rowmap_t xmap, const void* Hs, int64_t h_rows, int64_t part_len, float* y,
                          gfx_rowmap_t ymap, float* xcopy, gfx_rowmap_t cmap, int64_t R, int64_t C_in, int64_t C_f,
                          int64_t L, int64_t Lout, int64_t off, int64_t N, void* ws, size_t ws_bytes, int schedule,
                          void* stream) {
    return fftconv_sched(x, xmap, Hs, h_rows, part_len, y, ymap, xcopy, cmap, R, C_in, C_f, L, Lout, off, N, ws, ws_bytes,
                         schedule, stream, nullptr, nullptr);
}

int gfx_fftconv_rowmax_f32(const float* x, gfx_rowmap_t xmap, const void* Hs, int64_t h_rows, int64_t part_len, float* y,
                           gfx_rowmap_t ymap, float* xcopy, gfx_rowmap_t cmap, int64_t R, int64_t C_in, int64_t C_f,
                           int64_t L, int64_t Lout, int64_t off, int64_t N, void* ws, size_t ws_bytes, uint32_t* rowmax,
                           int* rowmax_written, void* stream) {
    if (!rowmax || !rowmax_written) return GFX_EINVAL;
    *rowmax_written = 0;
    return fftconv_sched(x, xmap, Hs, h_rows, part_len, y, ymap, xcopy, cmap, R, C_in, C_f, L, Lout, off, N, ws, ws_bytes,
                         GFX_SCHED_AUTO, stream, rowmax, rowmax_written);
}

static int fftconv_sched(const float* x, gfx_rowmap_t xmap, const void* Hs, int64_t h_rows, int64_t part_len, float* y,
                         gfx_rowmap_t ymap, float* xcopy, gfx_rowmap_t cmap, int64_t R, int64_t C_in, int64_t C_f,
                         int64_t L, int64_t Lout, int64_t off, int64_t N, void* ws, size_t ws_bytes, int schedule,
                         void* stream, uint32_t* rowmax, int* rowmax_written) {
    if (schedule != GFX_SCHED_AUTO && schedule != GFX_SCHED_TILE && schedule != GFX_SCHED_PIPE) return GFX_EINVAL;
    if (!x || !Hs || !y || R <= 0 || L <= 0 || Lout <= 0 || N <= 0) return GFX_EINVAL;
    if (h_rows < 1 || h_rows > R || h_rows > 0x7fffffffLL) return GFX_EINVAL;
    if (xcopy && (off != 0 || Lout < L || C_in < C_f || N > TILE_M + 1 || cmap.inner <= 0 || cmap.inner > 0x7fffffffLL))
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

    // the persistent hand-scheduled kernel: by name, or by AUTO once a launch is large enough to fill its pipeline
    PipeModule* pm = pipe_module();   // (loaded on the first call on a device, whatever the schedule: never mid-capture)
    if (schedule == GFX_SCHED_AUTO && auto_schedule() != GFX_SCHED_AUTO) schedule = auto_schedule();
    const int pv = pm ? pipe_variant(a, g, xcopy != nullptr, N) : -1;
    if (schedule == GFX_SCHED_PIPE && pv < 0) {
        if (auto_schedule() == GFX_SCHED_PIPE) schedule = GFX_SCHED_TILE;   // the override only steers what can be steered
        else return GFX_EINVAL;
    }
    if (schedule == GFX_SCHED_PIPE || (schedule == GFX_SCHED_AUTO && pv >= 0 && GFX_PIPE_AUTO && a.nblocks >= 16 * pm->cus))
    {
        // (the shipped code object is generated without the rowmax knob: this kernel takes even output lengths only, and the
        // one consumer of the maxima -- the odd-length aliasing -- has an odd one)
        const int rc = launch_pipe(pm, pv, x, Hs, y, xcopy, a, tw, st);
        if (rc == GFX_OK) t_last_kernel = kPipeVariants[pv].name;
        return rc;
    }
    if (g.nparts == 1) {
        if (allow_lds(fftconv1_kernel<false>) || allow_lds(fftconv1_kernel<true>)) return GFX_ELAUNCH;
        if (xcopy)
            hipLaunchKernelGGL(fftconv1_kernel<true>, dim3(pad8(a.nblocks)), dim3(TILE_T), TILE_LDS_BYTES, st, x,
                               (const float4*)Hs, y, xcopy, a, tw, rowmax);
        else
            hipLaunchKernelGGL(fftconv1_kernel<false>, dim3(pad8(a.nblocks)), dim3(TILE_T), TILE_LDS_BYTES, st, x,
                               (const float4*)Hs, y, (float*)nullptr, a, tw, rowmax);
        if (hipGetLastError() != hipSuccess) return GFX_ELAUNCH;
        t_last_kernel = xcopy ? "fftconv1_kernel<true>" : "fftconv1_kernel<false>";
        if (rowmax && rowmax_written) *rowmax_written = 1;
        return GFX_OK;
    }
    if (g.ntiles == 1) {
        if (allow_lds(winmac_kernel)) return GFX_ELAUNCH;
        hipLaunchKernelGGL(winmac_kernel, dim3(pad8(a.nblocks)), dim3(TILE_T), TILE_LDS_BYTES, st, x, (const float4*)Hs, y,
                           a, tw);
        if (hipGetLastError() != hipSuccess) return GFX_ELAUNCH;
        t_last_kernel = "winmac_kernel";
        return GFX_OK;
    }
    const int64_t nwin = g.ntiles + g.nparts - 1;
    const size_t need = (size_t)R * C_in * nwin * H_TILE_F4 * sizeof(float4);
    if (!ws || ws_bytes < need) return GFX_ENOSPC;
    if (allow_lds(xspec_kernel) || allow_lds(macinv_kernel)) return GFX_ELAUNCH;
    ConvArgs ax = a;
    ax.nblocks = R * C_in * nwin;
    if (ax.nblocks > 0x7ffffff0LL) return GFX_EINVAL;
    hipLaunchKernelGGL(xspec_kernel, dim3(pad8(ax.nblocks)), dim3(TILE_T), TILE_LDS_BYTES, st, x, (float2*)ws, ax,
                       nwin, tw);
    // (the pair kernel asks for "partition -1" at byte offset 0x40000000 and counts on the descriptor's range check to
    // return zeros: the filter's partitions must end below that offset -- ~126 M taps; longer filters take macinv_kernel)
    if (schedule != GFX_SCHED_TILE && (int64_t)(g.nparts + 1) * H_TILE_F4 * 16 < 0x40000000LL) {
        // two consecutive output tiles per 512-thread workgroup: each window spectrum and filter partition fetched once a pair
        if (allow_lds(macinv_pair_kernel, 2 * TILE_LDS_BYTES)) return GFX_ELAUNCH;
        ConvArgs ap = a;
        ap.nblocks = R * a.Cout * ((g.ntiles + 1) / 2);
        hipLaunchKernelGGL(macinv_pair_kernel, dim3(pad8(ap.nblocks)), dim3(2 * TILE_T), 2 * TILE_LDS_BYTES, st,
                           (const float2*)ws, (const float4*)Hs, y, ap, nwin, tw, rowmax);
        if (hipGetLastError() != hipSuccess) return GFX_ELAUNCH;
        t_last_kernel = "xspec_kernel+macinv_pair_kernel";
        if (rowmax && rowmax_written) *rowmax_written = 1;
        return GFX_OK;
    }
    hipLaunchKernelGGL(macinv_kernel, dim3(pad8(a.nblocks)), dim3(TILE_T), TILE_LDS_BYTES, st, (const float2*)ws,
                       (const float4*)Hs, y, a, nwin, tw, rowmax);
    if (hipGetLastError() != hipSuccess) return GFX_ELAUNCH;
    t_last_kernel = "xspec_kernel+macinv_kernel";
    if (rowmax && rowmax_written) *rowmax_written = 1;
    return GFX_OK;
}

const char* gfx_fftconv_last_kernel(void) { return t_last_kernel; }

}  // extern "C"
