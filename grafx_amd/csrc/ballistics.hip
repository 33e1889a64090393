// Ballistics: the attack / release one-pole recursion of the dynamics processors' "ballistics" smoothers.
//
// Replaces (reference src/grafx/processors):
//   Ballistics.forward -> torchcomp.compressor_core(x, zi = 1, at, rt)      core/envelope.py:84-101
//   (the energy that feeds it: energy = x.square().mean(-2), dynamics.py:390)
//
//     y[-1] = 1;   c[n] = at if x[n] < y[n-1] else rt;   y[n] = (1 - c[n]) y[n-1] + c[n] x[n]
//
// The coefficient depends on the state, so this is not an associative scan.  What the recursion does have
// is CONTRACTION: every step maps the state through a non-decreasing piecewise-linear function of slope 1 - at or
// 1 - rt < 1, so two trajectories over the same input approach each other by that factor per sample and, in float32,
// become THE SAME BITS after a few time constants.  The kernel uses that to cut rows into chunks without giving up one
// bit of the sequential result:
//
//   pass 0  a row is cut into up to 64 chunks, one lane each (the chunks of a row sit in neighbouring lanes of one
//           wave).  A lane starts `warm` samples before its chunk from a guess (the first sample it sees), walks the
//           recursion -- the same two products and one sum per step, rounded separately, as the reference's CPU loop --
//           and keeps the state it ENTERS its chunk with.  After the chunk the lanes compare, bit for bit, that entry
//           state with the state the lane to the left LEFT its chunk with (one DPP shift).  Chunk 0 starts from
//           y[-1] = 1, so by induction a row whose comparisons all hold is exactly the sequential recursion; a row
//           with a mismatch, or whose slower coefficient needs a longer warm-up than a chunk, is flagged.
//   pass 1  flagged rows are walked whole, one lane per row (16 / 32 / 64 rows per wave).
//
// The output is therefore ALWAYS the float32 sequential recursion; speculation only decides how fast it is produced.
// Tiles of 64 rows x 64 samples go through LDS so that HBM sees 256-byte segments while each lane walks its own row;
// the next tile's loads are in flight (registers) while the current one is walked.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/grafx_amd.h"
#include "dyn_gain.hpp"

// every product and sum of this file rounds on its own, as numpy / numba / torch's CPU kernels do: a fused multiply-add
// in the recursion (or in the energy x_l^2 + x_r^2) would change bits
#pragma clang fp contract(off)

namespace gfx {

constexpr int BTMAX = 64;      // chunk lengths and warm-ups are multiples of this (tiles are 64 or 32 samples per row)

typedef float v2f __attribute__((ext_vector_type(2)));
using f4 = float __attribute__((ext_vector_type(4)));

struct BlArgs {
    const float* u;      // SRC 0: (R, L) rows.  SRC 1: signal x addressed through xmap, the rows are mean_c x^2
    gfx_rowmap_t xmap;
    int C;
    float* y;            // (R, L)
    const float* z;      // (R, 2): z_alpha (at, rt = sigmoid, envelope.py:97-99), or with is_coef the coefficients themselves
    int is_coef;
    unsigned* flag;      // (R): written by pass 0 (1 = the row is left to pass 1), read by pass 1
    int64_t R, L;
    int lg;              // log2(chunks per row), 0..6
    int T;               // chunk length, a multiple of BT (also the longest warm-up)
    int pass;            // 0: chunks, verified;  1: flagged rows, whole;  2: all rows, whole (no flags)
    int from_flags;      // pass 0 again, with longer chunks, over the rows an earlier pass 0 flagged
    int vec;             // 16-byte accesses are legal (alignment, L % 4 == 0)
    // DST 1 (the whole compressor / gate in this kernel): y is the OUTPUT SIGNAL, addressed through ymap like x, and the
    // gain computer's per-row parameters (row r reads parameter row r % prows; z likewise)
    gfx_rowmap_t ymap;
    const float *log_threshold, *log_ratio, *log_knee;
    int knee, gate;
    unsigned prows;
};

__device__ __forceinline__ int64_t row_off(const gfx_rowmap_t& m, int64_t r, int c) {
    const int64_t q = r / m.inner, rem = r - q * m.inner;
    return q * m.stride_outer + rem * m.stride_inner + (int64_t)c * m.stride_ch;
}

// one step; both candidates are formed and the comparison picks one -- the same values as
// `c = x < y ? at : rt; y = (1 - c) * y + c * x` with separately rounded products, on a dependency chain of three
// instructions (product, sum, select; the comparison runs beside the product).  Plain v_mul / v_add: the packed-FP32 forms
// of the same arithmetic (v_pk_mul_f32 / v_pk_add_f32 on (at, rt) pairs) measured 84 cycles per step on this chain.
__device__ __forceinline__ float bstep(float s, float x, v2f c, v2f om) {
    const float ya = om.x * s + c.x * x;
    const float yr = om.y * s + c.y * x;
    return x < s ? ya : yr;
}

// The workgroup is ONE wave: its LDS instructions execute in order, so the cooperative copies and the per-lane walk need no
// s_barrier between them -- and must not get __syncthreads(), whose fence is `s_waitcnt vmcnt(0)`: it would wait for the
// NEXT tile's loads, requested just before, and put an HBM round trip (2.5 us) into every 64-step tile (measured: the
// whole-row walk at 39 ns per step).  A compiler-level barrier keeps the order of the LDS accesses.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// four samples of a row at position n.  VEC: one 16-byte access (n % 4 == 0, L % 4 == 0, aligned rows) that is always
// issued -- at a safe address when the position is outside the row or the row is not walked, and then returns samples
// nobody uses (no select on the loaded value: it would make the request wait for its own data) -- so that the
// cooperative passes compile to straight-line code; otherwise element by element, zero outside [0, L).
template <bool VEC>
__device__ __forceinline__ f4 load_f4(const float* __restrict__ base, int64_t off, int64_t n, int64_t L, bool live) {
    f4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    const bool in = live && n >= 0 && n < L;
    if (VEC) {
        v = __builtin_nontemporal_load(reinterpret_cast<const f4*>(base + (in ? off : 0)));
    } else if (in) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (n + i < L) v[i] = base[off + i];
    }
    return v;
}

// RPW rows per wave, tiles of BT samples per row (64 rows x 32 samples or 16 / 32 rows x 64: at most 8 cooperative passes,
// i.e. 32 + 32 prefetch registers and 32 of offsets -- the 64 x 64 tile needs 256 VGPRs with the energy source)
// PF tiles are requested ahead of the one being walked (register sets): the whole-row walk is a handful of waves with
// nothing else on their SIMD to cover a load's round trip, and one tile ahead (a 64-step walk, ~0.4 us) is shorter than it.
// DST 0: the smoothed rows are the output.  DST 1 (with SRC 1): the gain computer and the gain stage ride on the walk --
// the tile keeps the signal's channels in LDS instead of their energy, a lane forms the energy of its sample, steps the
// recursion and at once turns the new state into the gain (log -> knee -> exp, dynamics.py:394-405) that scales the
// sample in place; the tile then leaves as the compressor's output.  Energy -> envelope -> gain -> output in one pass over
// x: 16 bytes per stereo sample instead of the 32 of "envelope kernel + gain kernel" (48 through round 4's four launches).
template <int RPW, int BT, int SRC, bool VEC, int PF, int DST>
__global__ __launch_bounds__(64) void ballistics_walk_kernel(BlArgs a) {
    constexpr int PITCH = BT + 4;              // LDS row pitch in floats: 16-byte row reads of 16 neighbouring lanes cover all banks
    constexpr int LPR = BT / 4;                // lanes per row in a cooperative pass
    constexpr int RPP = 64 / LPR;              // rows per pass
    constexpr int NP = RPW / RPP;              // cooperative passes
    constexpr int PLANE = RPW * PITCH;         // DST 1: one LDS plane per channel
    __shared__ __attribute__((aligned(16))) float tile[(DST == 1 ? 2 : 1) * PLANE];
    const int lane = threadIdx.x;
    const int nchunk = 1 << a.lg;
    const int64_t nvr = a.R << a.lg;             // virtual rows: (row, chunk)
    const int64_t vbase = (int64_t)blockIdx.x * RPW;
    const int64_t vr = vbase + lane;
    const bool valid = lane < RPW && vr < nvr;
    const int64_t row = valid ? (vr >> a.lg) : 0;
    const int chunk = (int)(vr & (nchunk - 1));
    v2f c = {0.0f, 0.0f}, om = {1.0f, 1.0f};
    Knee q;
    if (DST == 1) {
        const unsigned pr = valid ? (unsigned)(row % a.prows) : 0u;
        knee_setup(q, a.log_threshold[pr], a.log_ratio[pr], a.log_knee ? a.log_knee[pr] : 0.0f, a.knee, a.gate);
    }
    if (valid) {
        const int64_t zr = DST == 1 ? row % a.prows : row;
        float at = a.z[2 * zr], rt = a.z[2 * zr + 1];
        if (!a.is_coef) {
            at = 1.0f / (1.0f + expf(-at));
            rt = 1.0f / (1.0f + expf(-rt));
        }
        c = v2f{at, rt};
        om = v2f{1.0f - at, 1.0f - rt};
    }
    bool alive = valid;
    int warm = 0;
    if (a.pass == 0 && a.from_flags) alive = valid && a.flag[row] != 0u;   // (rows an earlier pass finished are not touched)
    if (a.pass == 0) {
        // warm-up: two things have to happen before a lane's state is THE row's state.  An error as large as the signal
        // contracts below an ulp in 18 / c steps at the slower coefficient c (2^-26); from there the two trajectories are
        // neighbouring floats, and a step rounds neighbours to the same float with probability ~c -- another 16 / c steps
        // leave 1e-7 of the boundaries unmerged (one unmerged boundary sends its row to the whole-row walk, and one such
        // row makes that launch take its full 4.4 ms: the margin is what keeps the verified chunks worth having at small
        // c; with 18 / c alone every row of a launch at c = 2.5e-3 failed the check).  Rounded up to whole tiles; rows
        // that need more than a chunk are not cut.
        const float cmin = fminf(c.x, c.y);
        float need = 3.0e38f;
        if (cmin >= 1.0f) need = 8.0f;
        else if (cmin > 0.0f) need = ceilf(-34.0f / log1pf(-cmin)) + 8.0f;
        const bool slow = !(need <= (float)a.T);
        if (alive && slow) {
            alive = false;
            if (chunk == 0) a.flag[row] = 1u;
        }
        warm = alive ? ((int)need + BTMAX - 1) / BTMAX * BTMAX : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) warm = max(warm, __shfl_xor(warm, o, 64));
    } else if (a.pass == 1) {
        alive = valid && a.flag[row] != 0u;
    }
    const int64_t cstart = (int64_t)chunk * a.T;
    const int64_t cend = min(cstart + (int64_t)a.T, a.L);
    alive = alive && cstart < a.L;               // (a chunk past the end of the row has nothing to do)
    const uint64_t alive_mask = __ballot(alive);
    if (alive_mask == 0) return;
    const int warm_tiles = warm / BT;         // (warm and T are multiples of BTMAX)
    const int ntile = warm_tiles + a.T / BT;
    const int cr = lane / LPR, cc = (lane % LPR) * 4;   // cooperative copies: row within a pass, first sample
    const float invC = 1.0f / (float)a.C;

    // Cooperative pass p moves the tile row of virtual row vbase + RPP p + cr.  Per pass and lane, kept in registers for the
    // whole walk: the element offset of this lane's four samples at tile 0 in the source and in y; the position of
    // those samples within their row follows from the chunk index with three 32-bit operations.
    int64_t soff[SRC == 1 ? NP : 1], yoff[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int64_t v = vbase + p * RPP + cr;
        const int64_t rrow = v >> a.lg;
        const int64_t pos0 = (int64_t)(int)(v & (nchunk - 1)) * a.T - warm + cc;
        yoff[p] = DST == 1 ? row_off(a.ymap, rrow < a.R ? rrow : 0, 0) + pos0 : rrow * a.L + pos0;
        if (SRC == 1) soff[p] = row_off(a.xmap, rrow < a.R ? rrow : 0, 0) + pos0;
    }
    const unsigned chunk_c = (unsigned)((vbase + cr) & (nchunk - 1));
    auto pass_pos = [&](int p, int ti, bool& live) {      // position in its row of pass p's samples at tile ti
        live = ((alive_mask >> (p * RPP)) >> cr) & 1ull;
        const unsigned ch = (chunk_c + (unsigned)(RPP * p)) & (unsigned)(nchunk - 1);
        return (int64_t)((int)(ch * (unsigned)a.T) - warm + cc + ti * BT);
    };
    f4 nx[PF][NP], nz[PF][SRC == 1 ? NP : 1];
    auto request = [&](int ti, f4 (&bx)[NP], f4 (&bz)[SRC == 1 ? NP : 1]) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            bool live;
            const int64_t n = pass_pos(p, ti, live);
            const int64_t o = (SRC == 1 ? soff[p] : yoff[p]) + (int64_t)ti * BT;
            bx[p] = load_f4<VEC>(a.u, o, n, a.L, live);
            if (SRC == 1 && a.C == 2) bz[p] = load_f4<VEC>(a.u, o + a.xmap.stride_ch, n, a.L, live);
        }
    };
#pragma unroll
    for (int u = 0; u < PF; ++u)
        if (u < ntile) request(u, nx[u], nz[u]);
    float s = 1.0f;            // zi = 1 (envelope.py:98); chunks > 0 replace it by the first sample they see
    float entry = 1.0f;
    float* mine = &tile[(lane < RPW ? lane : 0) * PITCH];   // (lanes >= RPW only copy)
    // one tile: the register set that holds it -> LDS (and re-requested for the tile PF ahead), walk, store
    auto do_tile = [&](int ti, f4 (&bx)[NP], f4 (&bz)[SRC == 1 ? NP : 1]) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            f4 v = bx[p];
            if (DST == 1) {   // the channels themselves: the walk forms the energy and scales them in place
                if (a.C == 2) *reinterpret_cast<f4*>(&tile[PLANE + (p * RPP + cr) * PITCH + cc]) = bz[p];
            } else if (SRC == 1) {   // energy = mean_c x^2: squares, sum and the division by C each rounded (dynamics.py:390)
                const f4 w = bz[p];
                v = a.C == 2 ? (v * v + w * w) * invC : v * v;
            }
            *reinterpret_cast<f4*>(&tile[(p * RPP + cr) * PITCH + cc]) = v;
        }
        if (ti + PF < ntile) request(ti + PF, bx, bz);
        wave_lds_sync();
        const bool warming = ti < warm_tiles;
        const int64_t n0 = cstart - warm + (int64_t)ti * BT;          // this lane's position at step 0 of the tile
        int nv = 0;                                                    // steps this lane takes in this tile
        if (alive) nv = warming ? (chunk > 0 ? BT : 0) : (int)max((int64_t)0, min((int64_t)BT, cend - n0));
        if (ti == 0 && chunk > 0) {
            if (DST == 1) {
                const float xa = mine[0], xb = mine[PLANE];
                s = a.C == 2 ? (xa * xa + xb * xb) * invC : xa * xa;
            } else {
                s = mine[0];
            }
        }
        if (ti == warm_tiles) entry = s;
        const bool whole = __all(nv == BT || nv == 0);
        if (lane >= RPW) {
        } else if (DST == 1) {
            for (int j = 0; j < BT; j += 4) {
                f4 xa = *reinterpret_cast<f4*>(mine + j), xb = {0.0f, 0.0f, 0.0f, 0.0f};
                if (a.C == 2) xb = *reinterpret_cast<f4*>(mine + PLANE + j);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float e = a.C == 2 ? (xa[i] * xa[i] + xb[i] * xb[i]) * invC : xa[i] * xa[i];
                    const float t = bstep(s, e, c, om);
                    s = j + i < nv ? t : s;
                    if (!warming) {   // uniform
                        const float g = FastMath::exp(log_gain_m<FastMath>(q, FastMath::log(s + 1e-5f)));   // dynamics.py:394-403
                        xa[i] *= g;
                        xb[i] *= g;
                    }
                }
                if (!warming) {
                    *reinterpret_cast<f4*>(mine + j) = xa;
                    if (a.C == 2) *reinterpret_cast<f4*>(mine + PLANE + j) = xb;
                }
            }
        } else if (whole) {
            float t = s;
#pragma unroll 4
            for (int j = 0; j < BT; j += 4) {
                f4 q = *reinterpret_cast<f4*>(mine + j);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    t = bstep(t, q[i], c, om);
                    q[i] = t;
                }
                *reinterpret_cast<f4*>(mine + j) = q;
            }
            s = nv ? t : s;
        } else {
            for (int j = 0; j < BT; j += 4) {
                f4 q = *reinterpret_cast<f4*>(mine + j);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float t = bstep(s, q[i], c, om);
                    s = j + i < nv ? t : s;
                    q[i] = s;
                }
                *reinterpret_cast<f4*>(mine + j) = q;
            }
        }
        wave_lds_sync();
        if (!warming) {
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                bool live;
                const int64_t n = pass_pos(p, ti, live);
                if (live && n < a.L) {
#pragma unroll
                    for (int ch = 0; ch < (DST == 1 ? 2 : 1); ++ch) {
                        if (ch == 1 && a.C != 2) break;
                        const f4 v = *reinterpret_cast<const f4*>(&tile[ch * PLANE + (p * RPP + cr) * PITCH + cc]);
                        float* o = a.y + yoff[p] + (int64_t)ti * BT + (ch ? a.ymap.stride_ch : 0);
                        if (VEC) {
                            __builtin_nontemporal_store(v, reinterpret_cast<f4*>(o));
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (n + i < a.L) o[i] = v[i];
                        }
                    }
                }
            }
            wave_lds_sync();
        }
    };
    for (int ti0 = 0; ti0 < ntile; ti0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u)
            if (ti0 + u < ntile) do_tile(ti0 + u, nx[u], nz[u]);
    }
    if (a.pass == 0) {
        // lane k entered its chunk with the state its warm-up reached; the row's true state there is what lane k - 1
        // left its chunk with, PROVIDED lane k - 1 was right itself -- chunk 0 is (it starts from zi), so a row without a
        // mismatch is the sequential recursion bit for bit
        const float left = __shfl_up(s, 1, 64);
        const bool ok = !alive || chunk == 0 || __float_as_uint(entry) == __float_as_uint(left);
        const uint64_t bad = __ballot(!ok);
        if (valid && chunk == 0 && ((alive_mask >> lane) & 1ull)) {
            const uint64_t m = (nchunk == 64 ? ~0ull : ((1ull << nchunk) - 1ull)) << lane;   // the row's lanes
            a.flag[row] = (bad & m) != 0ull ? 1u : 0u;
        }
    }
}

template <int SRC, int DST>
static int launch_walk(int rpw, const BlArgs& a, hipStream_t st) {
    const int64_t nvr = a.R << a.lg;
    const int64_t blocks = (nvr + rpw - 1) / rpw;
    if (blocks > 0x7fffffffLL) return GFX_EINVAL;
    const dim3 grid((unsigned)blocks), blk(64);
    if (!a.vec) {   // unaligned rows / L % 4 != 0: the element-wise form, one shape
        const dim3 g64((unsigned)((nvr + 63) / 64));
        hipLaunchKernelGGL((ballistics_walk_kernel<64, 32, SRC, false, 1, DST>), g64, blk, 0, st, a);
    } else if (rpw == 16) hipLaunchKernelGGL((ballistics_walk_kernel<16, 64, SRC, true, DST ? 2 : 4, DST>), grid, blk, 0, st, a);
    else if (rpw == 32) hipLaunchKernelGGL((ballistics_walk_kernel<32, 64, SRC, true, DST ? 1 : 2, DST>), grid, blk, 0, st, a);
    else hipLaunchKernelGGL((ballistics_walk_kernel<64, 32, SRC, true, 1, DST>), grid, blk, 0, st, a);
    return GFX_OK;
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// u: SRC 0 rows / SRC 1 signal.  Without a workspace (the flags of the two-pass form) every row is walked whole.
template <int SRC, int DST = 0>
static int ballistics_run(const float* u, gfx_rowmap_t xmap, int C, const float* z, int is_coef, float* y, int64_t R, int64_t L,
                   void* ws, size_t ws_bytes, hipStream_t st, const BlArgs* gain = nullptr) {
    if (!u || !z || !y || R <= 0 || L <= 0 || R > 0x3fffffffLL || L > 0x7fffff00LL) return GFX_EINVAL;
    if (SRC == 1 && ((C != 1 && C != 2) || xmap.inner <= 0)) return GFX_EINVAL;
    if (ws && ws_bytes < gfx_ballistics_ws_bytes(R)) return GFX_ENOSPC;
    BlArgs a;
    if (gain) a = *gain;    // DST 1: ymap and the gain computer's parameters
    else { a.ymap = xmap; a.log_threshold = a.log_ratio = a.log_knee = nullptr; a.knee = a.gate = 0; a.prows = 1; }
    a.u = u; a.xmap = xmap; a.C = C; a.y = y; a.z = z; a.is_coef = is_coef; a.flag = reinterpret_cast<unsigned*>(ws);
    a.R = R; a.L = L;
    a.vec = (L % 4 == 0) && aligned16(u) && aligned16(y);
    if (SRC == 1) a.vec = a.vec && xmap.stride_outer % 4 == 0 && xmap.stride_inner % 4 == 0 && xmap.stride_ch % 4 == 0;
    if (DST == 1) a.vec = a.vec && a.ymap.stride_outer % 4 == 0 && a.ymap.stride_inner % 4 == 0 && a.ymap.stride_ch % 4 == 0;
    // chunks per row: enough virtual rows for ~4 waves per SIMD, chunks no shorter than 256 samples, at most one wave per row
    int lg = 0;
    while (ws && lg < 6 && (R << lg) < 262144 && (L >> (lg + 1)) >= 256) ++lg;
    const int64_t per = (L + (1LL << lg) - 1) >> lg;
    a.lg = lg;
    a.T = (int)((per + BTMAX - 1) / BTMAX * BTMAX);
    // rows per wave of the whole-row walk: the fewest that still leave every wave a SIMD of its own
    const int rpw = R <= 16 * 1024 ? 16 : (R <= 32 * 1024 ? 32 : 64);
    int rc;
    a.from_flags = 0;
    if (lg == 0) {
        a.pass = 2;
        rc = launch_walk<SRC, DST>(rpw, a, st);
    } else {
        a.pass = 0;
        rc = launch_walk<SRC, DST>(64, a, st);
        if (rc != GFX_OK) return rc;
        if (lg >= 4) {
            // second try for the rows whose warm-up did not fit (or did not converge in) a chunk: chunks eight times as
            // long -- a coefficient of 2.5e-3 warms up in 7 200 samples, no use for 4096-sample chunks, a fifth of a
            // 32768-sample one; whole-row walks are bound by the latency of one step (34 ns: 4.4 ms for 131072 samples
            // however few the rows), chunks divide it
            a.lg = lg - 3;
            const int64_t per2 = (L + (1LL << a.lg) - 1) >> a.lg;
            a.T = (int)((per2 + BTMAX - 1) / BTMAX * BTMAX);
            a.from_flags = 1;
            rc = launch_walk<SRC, DST>(64, a, st);
            if (rc != GFX_OK) return rc;
            a.from_flags = 0;
        }
        a.pass = 1;
        a.lg = 0;
        a.T = (int)((L + BTMAX - 1) / BTMAX * BTMAX);
        rc = launch_walk<SRC, DST>(rpw, a, st);
    }
    if (rc != GFX_OK) return rc;
    return hipGetLastError() == hipSuccess ? GFX_OK : GFX_ELAUNCH;
}

}  // namespace gfx

using namespace gfx;

extern "C" {

size_t gfx_ballistics_ws_bytes(int64_t R) { return R <= 0 ? 0 : (size_t)R * sizeof(unsigned); }

int gfx_ballistics_f32(const float* u, const float* z_alpha, float* y, int64_t R, int64_t L, void* stream) {
    gfx_rowmap_t none = {1, 0, 0, 0};
    return ballistics_run<0>(u, none, 1, z_alpha, 0, y, R, L, nullptr, 0, (hipStream_t)stream);
}

int gfx_ballistics_ws_f32(const float* u, const float* z_alpha, int is_coef, float* y, int64_t R, int64_t L, void* ws,
                          size_t ws_bytes, void* stream) {
    gfx_rowmap_t none = {1, 0, 0, 0};
    return ballistics_run<0>(u, none, 1, z_alpha, is_coef, y, R, L, ws, ws_bytes, (hipStream_t)stream);
}

int gfx_dynamics_ballistics_f32(const float* x, gfx_rowmap_t xmap, float* y, gfx_rowmap_t ymap, const float* log_threshold,
                                const float* log_ratio, const float* log_knee, const float* z_alpha, int64_t param_rows,
                                int64_t R, int64_t C, int64_t L, int knee, int gate, void* ws, size_t ws_bytes, void* stream) {
    if (!log_threshold || !log_ratio || knee < 0 || knee > 2 || (knee != 0 && !log_knee)) return GFX_EINVAL;
    if (param_rows < 1 || param_rows > R || xmap.inner <= 0 || ymap.inner <= 0) return GFX_EINVAL;
    BlArgs g;
    g.ymap = ymap; g.log_threshold = log_threshold; g.log_ratio = log_ratio; g.log_knee = log_knee;
    g.knee = knee; g.gate = gate; g.prows = (unsigned)param_rows;
    return ballistics_run<1, 1>(x, xmap, (int)C, z_alpha, 0, y, R, L, ws, ws_bytes, (hipStream_t)stream, &g);
}

int gfx_ballistics_energy_f32(const float* x, gfx_rowmap_t xmap, int64_t C, const float* z_alpha, int is_coef, float* env,
                              int64_t R, int64_t L, void* ws, size_t ws_bytes, void* stream) {
    return ballistics_run<1>(x, xmap, (int)C, z_alpha, is_coef, env, R, L, ws, ws_bytes, (hipStream_t)stream);
}

}  // extern "C"
