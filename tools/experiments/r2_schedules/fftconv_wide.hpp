// fftconv1 on the 512-thread "wide" tile (fft_tile512.hpp): four waves per SIMD instead of two.
// Included by fftconv.hip after ConvArgs / the buffer helpers; same tile geometry, same arguments.
#pragma once
#include "fft_tile512.hpp"

namespace gfx {
namespace wide {

// Filter spectra in the wide thread layout: {alpha, beta} of the bin in register r of thread t at [r * 512 + t], from the
// (He, Ho) pairs hspec_kernel stores.  One workgroup per (filter row-channel).
__global__ __launch_bounds__(WT) void hconv_kernel(const float4* __restrict__ Hs, float4* __restrict__ Hw) {
    const int t = threadIdx.x;
    const float4* src = Hs + (int64_t)blockIdx.x * H_TILE_F4;
    float4* dst = Hw + (int64_t)blockIdx.x * W_H_F4;
    const int j = j_of(t);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int k3 = brev(r, 4);
        const int k = j + 512 * k3;          // this lane's bin
        int slot, th, kc;                    // canonical pair: slot / thread of the 256-thread layout, its bin kc
        bool side_b;                         // this bin is the mirror (M - kc) of the stored pair
        if (j == 0) {
            th = 0;
            side_b = k3 > 8;
            slot = side_b ? 16 - k3 : k3;
            kc = 512 * slot;
        } else if (j == 256) {
            th = 0;
            side_b = k3 > 7;
            const int q = side_b ? 15 - k3 : k3;
            slot = 9 + q;
            kc = 256 + 512 * q;
        } else if (j < 256) {
            th = j;
            slot = k3;
            side_b = false;
            kc = k;
        } else {
            th = 512 - j;
            slot = 15 - k3;
            side_b = true;
            kc = TILE_M - k;
        }
        const float4 h = src[slot * TILE_T + th];
        double sn, cs;
        sincospi(2.0 * (double)kc / (double)TILE_M, &sn, &cs);
        float hex = h.x, hey = h.y, hox = h.z, hoy = h.w, wx = (float)cs, wy = (float)(-sn);  // W = exp(-2 pi i kc / M)
        if (side_b) {  // mirror bin: conj(He), conj(Ho), conj(W)
            hey = -hey;
            hoy = -hoy;
            wy = -wy;
        }
        // alpha = 2 He + i (1 - W) Ho,  beta = i (1 + W) Ho
        const float ax = 1.0f - wx, ay = -wy;   // 1 - W
        const float bx = 1.0f + wx, by = wy;    // 1 + W
        const float px = ax * hox - ay * hoy, py = ax * hoy + ay * hox;   // (1 - W) Ho
        const float qx = bx * hox - by * hoy, qy = bx * hoy + by * hox;   // (1 + W) Ho
        dst[r * WT + t] = make_float4(2.0f * hex - py, 2.0f * hey + px, -qy, qx);
    }
}

// p[a'] = (x[s + 2m], x[s + 2m + 1]), m = 256 (a' + 16 h) + b.  Needs s >= 0 or s a multiple of -512 (whole register
// rows before the row start): rows are loaded whole or skipped, the range check supplies the zeros past the row end.
__device__ __forceinline__ void load_window(cx (&p)[16], const float* __restrict__ row, int64_t s, int64_t L, int t) {
    const rsrc_t r = make_rsrc(row + s, (L - s) * 4);
    const uint32_t voff = 8u * (uint32_t)(t >> 1) + ((t & 1) ? 32768u : 0u);
    if (s >= 0) {
#pragma unroll
        for (int a = 0; a < 16; ++a) p[a] = buf_load_f2(r, voff, 2048u * a);
    } else {
        const int a_lo = (int)((-s) >> 9);                    // register rows A < a_lo lie before the row (a_lo <= 16)
        const uint32_t voff_m = (t & 1) ? voff : OOB;         // odd lanes hold A >= 16
#pragma unroll
        for (int a = 0; a < 16; ++a) p[a] = buf_load_f2(r, a < a_lo ? voff_m : voff, 2048u * a);
    }
}

// y[n0 + (2m - O)] for 2m >= O (O a multiple of 512, <= 8192), dropped past the row end by the range check;
// v[a'] = (z'[2m], z'[2m+1]).  Needs no sample pair straddling the row end (even `room`, or room >= 16384).
__device__ __forceinline__ void store_valid(const cx (&v)[16], float* __restrict__ row, int64_t n0, int64_t O, int64_t Lout,
                                            int t) {
    asm volatile("" : "+v"(t));
    const int64_t room = Lout - (n0 - O);
    const rsrc_t r = make_rsrc(row + (n0 - O), room * 4);
    const uint32_t voff = 8u * (uint32_t)(t >> 1) + ((t & 1) ? 32768u : 0u);
    const uint32_t voff_m = (t & 1) ? voff : OOB;
    const int a_lo = (int)(O >> 9);
#pragma unroll
    for (int a = 0; a < 16; ++a) buf_store_f2(r, a < a_lo ? voff_m : voff, 2048u * a, v[a]);
}

template <bool TEE>
__global__ __launch_bounds__(WT, 4) void fftconv1w_kernel(const float* __restrict__ x, const float4* __restrict__ Hw,
                                                          float* __restrict__ y, float* __restrict__ xcopy, ConvArgs a,
                                                          const float2* __restrict__ table) {
    extern __shared__ __attribute__((aligned(16))) cx lds[];
    const int t = threadIdx.x;
    const unsigned lb = xcd_logical_block();
    if (lb >= (unsigned)a.nblocks) return;
#ifdef GFX_STAGGER
    // experiment: the second workgroup of every CU starts late, so that the two resident tiles run out of phase
    if (blockIdx.x >= 256 && blockIdx.x < 512) {
#pragma unroll
        for (int i = 0; i < GFX_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
    }
#endif
    const unsigned ntiles = (unsigned)a.ntiles;
    const unsigned rco = lb / ntiles;
    const int64_t tile = lb - rco * ntiles;
    const unsigned r = rco / (unsigned)a.Cout;
    const int c = (int)(rco - r * (unsigned)a.Cout);
    const float* xrow = x + row_off(a.xmap, r, a.Cin == 1 ? 0 : c);
    float* yrow = y + row_off(a.ymap, r, c);
    const rsrc_t H = make_rsrc(Hw + ((int64_t)(r % a.hrows) * a.Cf + (a.Cf == 1 ? 0 : c)) * W_H_F4, W_H_F4 * 16);

#ifndef GFX_W_ABLATE
#define GFX_W_ABLATE 0   // experiments: 1 no window loads, 2 no spectrum loads, 4 no stores, 8 no transforms
#endif
#ifdef GFX_W_STAMP
    // experiment: phase timestamps (100 MHz wall clock) of every 64th workgroup into `xcopy` (non-tee launches only)
    unsigned long long stamp[11];
#define W_STAMP(i) stamp[i] = __builtin_amdgcn_s_memrealtime()
#define W_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define W_STAMP(i)
#define W_DRAIN()
#endif
    W_STAMP(0);
    cx p[1][16], w[1][16];
    f4v hab[16];
    if (GFX_W_ABLATE & 1) {
#pragma unroll
        for (int q = 0; q < 16; ++q) p[0][q] = cx{(float)(t + q), (float)(t ^ q)};
    } else {
        load_window(p[0], xrow, tile * a.V - a.O, a.L, t);
    }
    Tw4x4 tw1;
    load_tw(tw1, table, 0, t);
    cx* tw2tab = lds + W_IMG_F2;
    fill_tw2(tw2tab, table, t);
    W_STAMP(1);
    W_DRAIN();
    W_STAMP(2);
    if (TEE && !(GFX_W_ABLATE & 4)) store_valid(p[0], xcopy + row_off(a.cmap, r, c), tile * a.V, a.O, a.L, t);
    if (!(GFX_W_ABLATE & 8)) forward_1<1>(p, tw1, lds, t);
    W_STAMP(3);
    // the filter spectrum is requested here, where the window registers have just died: it arrives under passes 2 and 3
#pragma unroll
    for (int q = 0; q < 16; ++q)
        hab[q] = (GFX_W_ABLATE & 2) ? f4v{1.0f, (float)q, 0.5f, (float)t} : buf_load_f4(H, 16u * (uint32_t)t, (uint32_t)(WT * 16) * q);
    __builtin_amdgcn_sched_barrier(0);
    W_STAMP(4);
    if (!(GFX_W_ABLATE & 8)) {
        forward_2<1>(tw2tab, lds, t);
        W_STAMP(5);
        forward_3<1>(w, lds, t);
    } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) w[0][q] = p[0][q];
    }
    W_STAMP(6);
    spectral_product(w[0], hab, lds, t);
    W_STAMP(7);
    if (!(GFX_W_ABLATE & 8)) {
        load_tw(tw1, table, 0, t);
        inverse<1>(w, p, tw2tab, tw1, lds, t);
    } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) p[0][q] = w[0][q];
    }
    W_STAMP(8);
    if (!(GFX_W_ABLATE & 4)) store_valid(p[0], yrow, tile * a.V, a.O, a.Lout, t);
    else if (p[0][3].x == 1.2345f) yrow[t] = p[0][5].y;   // keep the arithmetic alive
    W_STAMP(9);
    W_DRAIN();
    W_STAMP(10);
#ifdef GFX_W_STAMP
    if (!TEE && xcopy && t == 0 && (lb & 63) == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(xcopy) + (lb >> 6) * 12;
#pragma unroll
        for (int q = 0; q < 11; ++q) o[q] = stamp[q];
        o[11] = blockIdx.x;
    }
#endif
}

// the wide kernel handles the plain causal geometry: one partition, no output offset, even row ends
static inline bool applicable(const ConvGeom& g, int64_t L, int64_t Lout, int64_t off, int64_t N) {
    return g.nparts == 1 && off == 0 && N <= TILE_M + 1 && (g.O & 511) == 0 && g.O <= 8192 && (Lout & 1) == 0 && (L & 1) == 0;
}

}  // namespace wide
}  // namespace gfx
