// Dynamics path: energy envelope, one-pole / ballistics smoothing, gain computer, gain apply.
//
// Replaces (reference src/grafx/processors):
//   Compressor.forward / NoiseGate.forward + gain_{hard,quad,exp}_knee   dynamics.py:361-489, 598-721
//   TruncatedOnePoleIIRFilter (h = (1-a) a^n, n < N; relu(convolve))       core/envelope.py:34-60
//   Ballistics -> torchcomp.compressor_core (third-party recursion)        core/envelope.py:84-101
//
// The truncated one-pole FIR is applied as its exact recursive form
//     y[n] = (1-a) * (u[n] - a^N * u[n-N]),   u[n] = a*u[n-1] + e[n]
// with a workgroup-wide prefix scan (wave shuffles + one LDS hop) over 1024-sample tiles, one
// workgroup streaming each row: x is read once and y written once (8 B per channel-sample).
// The a^N correction only runs for rows where it is not negligible (a^N > 1e-9).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/grafx_amd.h"
#include "dyn_gain.hpp"

#ifdef GFX_NT_OFF
#define GFX_NT_STORE(...) gfx_plain_store(__VA_ARGS__)
template <typename T> __device__ __forceinline__ void gfx_plain_store(T v, T* p) { *p = v; }
#else
#define GFX_NT_STORE(...) __builtin_nontemporal_store(__VA_ARGS__)
#endif

#ifndef GFX_DYN_PF
#define GFX_DYN_PF 1               // tiles requested ahead of the one being scanned (dyn_fused)
#endif

namespace gfx {

constexpr int DT = 256;            // threads per workgroup
constexpr int DE = 4;              // samples per thread per tile
constexpr int DTILE = DT * DE;     // 1024 samples per tile
// per-parameter-row pole table (dyn_pole_table_kernel), floats per row:
//   a^(4 l) l < 64 | a_step[6] | a_wave | a_N | ap[0..4] | a | 1 - a | trunc | one-shot | H | look-back | M | a^(512 i) i < 64
#ifndef GFX_DYN_BWD_FAST
#define GFX_DYN_BWD_FAST true     // hardware log / exp / reciprocal in the backward tiles (false: the library functions)
#endif
constexpr int DP_TAB = 148;
constexpr int DP_ONESHOT = 80, DP_HIST = 81, DP_LOOKBACK = 82, DP_LB_TILES = 83, DP_LB_W = 84;

__device__ __forceinline__ int64_t drow_off(const gfx_rowmap_t& m, int64_t r, int c) {
    const unsigned inner = (unsigned)m.inner, rr = (unsigned)r;  // both fit 32 bits (launchers check)
    const unsigned q = rr / inner, rem = rr - q * inner;
    return (int64_t)q * m.stride_outer + (int64_t)rem * m.stride_inner + (int64_t)c * m.stride_ch;
}

// a^k for integer k >= 0, rounded once from double (keeps long decays accurate)
__device__ __forceinline__ float powk(double log_a, double k) { return (float)exp(k * log_a); }

struct OnePole {
    float a;          // pole (already clamped)
    float one_m_a;    // 1 - a
    float ap[DE + 1]; // a^0 .. a^DE
    float a_lane;     // a^(DE * lane)
    float a_step[6];  // a^(DE * 2^d), d = 0..5 (in-wave scan offsets)
    float a_wave;     // a^(DE * 64)
    float a_N;        // a^N
    bool trunc;       // a^N not negligible
};

__device__ __forceinline__ void onepole_setup(OnePole& p, float z_alpha, int64_t N, int lane) {
    // core/envelope.py:51-54: alpha = clamp(sigmoid(z), max = 1 - 1e-5)
    p.a = fminf(sigmoidf(z_alpha), 1.0f - 1e-5f);
    p.one_m_a = 1.0f - p.a;
    const double la = log((double)p.a);
#pragma unroll
    for (int i = 0; i <= DE; ++i) p.ap[i] = powk(la, i);
    p.a_lane = powk(la, DE * lane);
#pragma unroll
    for (int d = 0; d < 6; ++d) p.a_step[d] = powk(la, DE << d);
    p.a_wave = powk(la, DE * 64);
    p.a_N = powk(la, (double)N);
    p.trunc = p.a_N > 1e-9f;
}

// One tile of the recursion u[n] = a u[n-1] + e[n] across the workgroup.
//   e[0..DE)  : this thread's inputs (tile-local positions DE*t .. DE*t+DE-1)
//   carry     : u at the end of the previous tile (same in every thread); updated
//   slots     : 4 floats of LDS for this tile parity
// returns u for the thread's DE positions.
__device__ __forceinline__ void scan_tile(const OnePole& p, const float (&e)[DE], float (&u)[DE], float& carry,
                                          float* slots, int lane, int wave) {
    float loc[DE];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < DE; ++i) {
        s = fmaf(p.a, s, e[i]);
        loc[i] = s;
    }
    // inclusive scan of thread totals inside the wave: S_t += a^(DE*2^d) * S_(t - 2^d)
    float inc = s;
#pragma unroll
    for (int d = 0; d < 6; ++d) {
        const float up = __shfl_up(inc, 1 << d, 64);
        if (lane >= (1 << d)) inc = fmaf(p.a_step[d], up, inc);
    }
    if (lane == 63) slots[wave] = inc;
    float excl = __shfl_up(inc, 1, 64);
    if (lane == 0) excl = 0.0f;
    __syncthreads();
    float state = carry;  // u entering wave 0
    float entering = state;
#pragma unroll
    for (int w = 0; w < DT / 64; ++w) {
        if (w == wave) entering = state;
        state = fmaf(p.a_wave, state, slots[w]);
    }
    carry = state;
    const float pre = fmaf(p.a_lane, entering, excl);  // u just before this thread's first sample
#pragma unroll
    for (int i = 0; i < DE; ++i) u[i] = fmaf(p.ap[i + 1], pre, loc[i]);
}

// ---- loads / stores of 4 consecutive samples with bounds -----------------------------------------
// samples [n, n+4) of a row, zero outside [lo, L)
__device__ __forceinline__ void load4(const float* __restrict__ row, int64_t n, int64_t L, bool vec, float (&v)[DE],
                                      int64_t lo = 0) {
    if (vec && n + DE <= L && n >= lo) {
        const float4 q = *reinterpret_cast<const float4*>(row + n);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
#pragma unroll
        for (int i = 0; i < DE; ++i) v[i] = (n + i >= lo && n + i < L) ? row[n + i] : 0.0f;
    }
}
__device__ __forceinline__ void store4(float* __restrict__ row, int64_t n, int64_t L, bool vec, const float (&v)[DE]) {
    if (vec && n + DE <= L) {
        using f4 = float __attribute__((ext_vector_type(4)));
        GFX_NT_STORE(f4{v[0], v[1], v[2], v[3]}, reinterpret_cast<f4*>(row + n));  // streamed output
    } else {
#pragma unroll
        for (int i = 0; i < DE; ++i)
            if (n + i < L) row[n + i] = v[i];
    }
}
__device__ __forceinline__ bool vec_ok(const float* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
// the same for rows known to be 16-byte aligned with L % 4 == 0 and n % 4 == 0: one predicated 16-byte access, no
// element-wise path (which is most of the code of a kernel that inlines a dozen of these)
template <bool AL>
__device__ __forceinline__ void ld4(const float* __restrict__ row, int64_t n, int64_t L, bool vec, float (&v)[DE]) {
    if (AL) {
        float4 q = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (n >= 0 && n < L) q = *reinterpret_cast<const float4*>(row + n);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
        load4(row, n, L, vec, v);
    }
}
template <bool AL>
__device__ __forceinline__ void st4(float* __restrict__ row, int64_t n, int64_t L, bool vec, const float (&v)[DE]) {
    if (AL) {
        using f4 = float __attribute__((ext_vector_type(4)));
        if (n < L) GFX_NT_STORE(f4{v[0], v[1], v[2], v[3]}, reinterpret_cast<f4*>(row + n));
    } else {
        store4(row, n, L, vec, v);
    }
}

struct DynArgs {
    gfx_rowmap_t xmap, ymap;
    int64_t R, L, N;       // rows, length, one-pole FIR length (iir smoother)
    int C;                 // channels
    int smoother;          // 0 none, 1 truncated one-pole
    int knee, gate;
    unsigned prows;        // parameter rows: row r uses parameters r % prows
    int nchunks;           // workgroups per row (time chunks; > 1 only with few rows, see the launcher)
    int64_t chunk_tiles;   // tiles per chunk
};

// ---- fused compressor / gate: energy -> one-pole -> log -> knee -> exp -> multiply -----------------
// Tiles [t_lo, t_hi) of the row are produced.  The smoother is a truncated FIR (N taps), so a chunk that does not
// start at the row start is exact if its scans start N samples early from a zero state: tiles [t_warm, t_lo) are
// scanned without producing output, and samples before `s0 = t_warm * DTILE` count as zero for both scans.
// u1row (training forward, whole rows only): also store (1-a) x the UN-truncated scan of the energy, which is what the
// backward pass needs (gfx_dynamics_bwd_u1_f32) -- one extra 4-byte store per sample here instead of a pass over x there.
template <bool TRUNC>
__device__ __forceinline__ void dyn_stream(const DynArgs& a, const OnePole& p, const Knee& q, const float* x0,
                                           const float* x1, float* y0, float* y1, float* slots, int t,
                                           int64_t t_warm, int64_t t_lo, int64_t t_hi, float* u1row = nullptr) {
    const int lane = t & 63, wave = t >> 6;
    const bool vx = vec_ok(x0) && vec_ok(x1) && vec_ok(y0) && vec_ok(y1);
    const bool vu = (a.L % 4) == 0;
    float carry_u = 0.0f;
    const float invC = 1.0f / (float)a.C;
    float carry = 0.0f, carry2 = 0.0f;
    const int64_t s0 = t_warm * DTILE;
    // software prefetch: the next tile's samples are requested before the current tile is scanned, so the
    // HBM round trip overlaps the scan / log / exp work (the barrier inside scan_tile would otherwise fence it)
    // (GFX_DYN_PF tiles ahead: one workgroup streams one row, so its memory parallelism is what it keeps in flight itself)
    float na[GFX_DYN_PF][DE], nb[GFX_DYN_PF][DE];
#pragma unroll
    for (int k = 0; k < GFX_DYN_PF; ++k) {
#pragma unroll
        for (int i = 0; i < DE; ++i) na[k][i] = nb[k][i] = 0.0f;
        if (t_warm + k < t_hi) {
            load4(x0, (t_warm + k) * DTILE + (int64_t)DE * t, a.L, vx, na[k]);
            if (a.C == 2) load4(x1, (t_warm + k) * DTILE + (int64_t)DE * t, a.L, vx, nb[k]);
        }
    }
    for (int64_t tile = t_warm; tile < t_hi; ++tile) {
        const int64_t n = tile * DTILE + DE * t;
        float xa[DE], xb[DE], e[DE], env[DE];
#pragma unroll
        for (int i = 0; i < DE; ++i) {
            xa[i] = na[0][i];
            xb[i] = nb[0][i];
        }
#pragma unroll
        for (int k = 0; k + 1 < GFX_DYN_PF; ++k)
#pragma unroll
            for (int i = 0; i < DE; ++i) {
                na[k][i] = na[k + 1][i];
                nb[k][i] = nb[k + 1][i];
            }
        if (tile + GFX_DYN_PF < t_hi) {
            load4(x0, n + (int64_t)GFX_DYN_PF * DTILE, a.L, vx, na[GFX_DYN_PF - 1]);
            if (a.C == 2) load4(x1, n + (int64_t)GFX_DYN_PF * DTILE, a.L, vx, nb[GFX_DYN_PF - 1]);
        }
#pragma unroll
        for (int i = 0; i < DE; ++i) {
            const float sq = a.C == 2 ? (xa[i] * xa[i] + xb[i] * xb[i]) : xa[i] * xa[i];
            e[i] = sq * invC;  // dynamics.py:390 energy = x.square().mean(-2)
        }
        if (a.smoother == 1) {
            float u[DE];
            if (TRUNC && u1row) {   // the un-truncated scan, for the backward pass only (uniform branch)
                float uf[DE], raw[DE];
                scan_tile(p, e, uf, carry_u, slots + 8 * (tile & 1) + 4, lane, wave);
#pragma unroll
                for (int i = 0; i < DE; ++i) raw[i] = p.one_m_a * uf[i];
                store4(u1row, n, a.L, vu, raw);
            }
            if (TRUNC) {
                // ONE scan of e[n] - a^N e[n-N]: the scan is linear, and subtracting before accumulating keeps the
                // truncation exact where U[n] - a^N U[n-N] would cancel (short filters, poles near one)
                float da[DE], db[DE];
                load4(x0, n - a.N, a.L, false, da, s0);
                if (a.C == 2) load4(x1, n - a.N, a.L, false, db, s0);
#pragma unroll
                for (int i = 0; i < DE; ++i)
                    e[i] = fmaf(-p.a_N, (a.C == 2 ? (da[i] * da[i] + db[i] * db[i]) : da[i] * da[i]) * invC, e[i]);
            }
            scan_tile(p, e, u, carry, slots + 8 * (tile & 1), lane, wave);
            if (tile < t_lo) continue;  // warm-up tile: only the scan state matters (uniform branch)
            if (!TRUNC && u1row) {
                float raw[DE];
#pragma unroll
                for (int i = 0; i < DE; ++i) raw[i] = p.one_m_a * u[i];
                store4(u1row, n, a.L, vu, raw);
            }
#pragma unroll
            for (int i = 0; i < DE; ++i) env[i] = fmaxf(p.one_m_a * u[i], 0.0f);  // relu, envelope.py:48
        } else {
#pragma unroll
            for (int i = 0; i < DE; ++i) env[i] = e[i];
        }
        float ga[DE], gb[DE];
#pragma unroll
        for (int i = 0; i < DE; ++i) {
            const float G = FastMath::log(env[i] + 1e-5f);                  // dynamics.py:394
            const float g = FastMath::exp(log_gain_m<FastMath>(q, G));      // 402-403
            ga[i] = g * xa[i];
            gb[i] = g * xb[i];
        }
        store4(y0, n, a.L, vx, ga);
        if (a.C == 2) store4(y1, n, a.L, vx, gb);
    }
}

#ifndef GFX_DYN_WAVES
#define GFX_DYN_WAVES 1
#endif
__global__ __launch_bounds__(DT, GFX_DYN_WAVES) void dyn_fused_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                       const float* __restrict__ log_threshold,
                                                       const float* __restrict__ log_ratio,
                                                       const float* __restrict__ log_knee,
                                                       const float* __restrict__ z_alpha, DynArgs a,
                                                       float* __restrict__ u1, const float* __restrict__ oneshot_tab) {
    __shared__ float slots[16];
    const int t = threadIdx.x;
    const int64_t r = blockIdx.x / a.nchunks;
    const int chunk = (int)(blockIdx.x - r * a.nchunks);
    OnePole p;
    p.trunc = false;
    const unsigned pr = (unsigned)r % a.prows;
    // rows the one-shot grid takes (dyn_oneshot_kernel, same pole table, complementary test) are not produced here
    if (oneshot_tab && (oneshot_tab[(size_t)pr * DP_TAB + DP_ONESHOT] != 0.0f ||
                        oneshot_tab[(size_t)pr * DP_TAB + DP_LOOKBACK] != 0.0f)) return;
    if (a.smoother == 1) onepole_setup(p, z_alpha[pr], a.N, t & 63);
    Knee q;
    knee_setup(q, log_threshold[pr], log_ratio[pr], log_knee ? log_knee[pr] : 0.0f, a.knee, a.gate);
    const float* x0 = x + drow_off(a.xmap, r, 0);
    const float* x1 = x + drow_off(a.xmap, r, a.C == 2 ? 1 : 0);
    float* y0 = y + drow_off(a.ymap, r, 0);
    float* y1 = y + drow_off(a.ymap, r, a.C == 2 ? 1 : 0);
    const int64_t ntiles = (a.L + DTILE - 1) / DTILE;
    const int64_t t_lo = chunk * a.chunk_tiles, t_hi = min(t_lo + a.chunk_tiles, ntiles);
    if (t_lo >= t_hi) return;
    // warm-up: N taps of history (none without a smoother), whole tiles, not before the row start
    const int64_t warm_tiles = a.smoother == 1 ? (a.N + DTILE - 1) / DTILE : 0;
    const int64_t t_warm = t_lo > warm_tiles ? t_lo - warm_tiles : 0;
    float* u1row = u1 ? u1 + r * a.L : nullptr;   // (launched with nchunks = 1 then)
    if (p.trunc)
        dyn_stream<true>(a, p, q, x0, x1, y0, y1, slots, t, t_warm, t_lo, t_hi, u1row);
    else
        dyn_stream<false>(a, p, q, x0, x1, y0, y1, slots, t, t_warm, t_lo, t_hi, u1row);
}

// ---- the same fused compressor / gate as dependency-free ONE-SHOT tiles ---------------------------------------------
// dyn_fused_kernel streams a row per workgroup: a few thousand long-lived streams at scattered addresses, the access
// shape that tops out at 4.8-5.5 TB/s on this chip where a one-shot copy reaches 6.2-6.6 (profiles/r2/
// stream2_copy_ceiling.txt).  Here every 512-sample tile of every row belongs to one WAVE of a short-lived workgroup
// (four consecutive tiles per workgroup, workgroups of a row on consecutive logical block indices of one XCD, so the
// chip sweeps memory front to back) and NO state crosses tiles: the smoother is a FIR, h[k] = (1-a) a^k, so the scan
// state entering a tile is the weighted sum of the H most recent energies before it, u[s-1] = sum_{k<H} a^k e[s-1-k],
// with H the number of taps above 1e-12.  A tile re-reads those H samples (lane l takes taps 4l .. 4l+3 as one
// predicated 16-byte load per channel) and reduces them with six shuffles: no LDS, no barrier.
// Which rows qualify is decided ON THE DEVICE from the pole table (no host synchronisation): a row is taken here when its
// truncation term is dead (a^N <= 1e-12) and H <= DYN_OS_HMAX; every other row leaves this grid at once and is produced
// by dyn_fused_kernel, launched over the same rows with the complementary test.
#ifndef GFX_DYN_OS_HMAX
#define GFX_DYN_OS_HMAX 256   // taps of history a one-shot tile may re-read (tile: 1024 samples)
#endif

// `any_lb` (nullable): set to 1 when some row takes the look-back tiles -- only then do the tile grids draw tickets
// (see LbArgs); without it no row is given to the look-back (the backward pass, callers without the larger workspace).
__global__ void dyn_pole_table_kernel(const float* __restrict__ z_alpha, float* __restrict__ tab, int64_t rows, int64_t N,
                                      unsigned* __restrict__ any_lb) {
    const int64_t r = blockIdx.x;
    const int lane = threadIdx.x;   // 64 threads
    if (r >= rows) return;
    OnePole p;
    onepole_setup(p, z_alpha[r], N, lane);
    float* t = tab + r * DP_TAB;
    t[lane] = p.a_lane;
    const double la = log((double)p.a);
    t[DP_LB_W + lane] = powk(la, 512.0 * lane);   // weight of the tile `lane + 1` tiles back in the state entering a tile
    if (lane == 0) {
#pragma unroll
        for (int d = 0; d < 6; ++d) t[64 + d] = p.a_step[d];
        t[70] = p.a_wave;
        t[71] = p.a_N;
#pragma unroll
        for (int i = 0; i <= DE; ++i) t[72 + i] = p.ap[i];
        t[77] = p.a;
        t[78] = p.one_m_a;
        t[79] = p.trunc ? 1.0f : 0.0f;
        // taps above 1e-12: H = ceil(log 1e-12 / log a); the row is one-shot material when the N-tap truncation is
        // beyond that (so H <= N and the truncation term is dead) and the history fits the re-read budget; with a longer
        // history (up to 64 tiles of 512 samples) the tiles get their entry state from their predecessors' aggregates
        const double h = ceil(-27.631021115928547 / la);
        const bool dead = h <= (double)N;
        const bool os = dead && h <= (double)GFX_DYN_OS_HMAX;
        const double m = ceil(h / 512.0);
        const bool lb = any_lb != nullptr && dead && !os && m <= 64.0;
        t[DP_ONESHOT] = os ? 1.0f : 0.0f;
        t[DP_HIST] = os ? (float)h : 0.0f;
        t[DP_LOOKBACK] = lb ? 1.0f : 0.0f;
        t[DP_LB_TILES] = lb ? (float)m : 0.0f;
        if (lb) *any_lb = 1u;
    }
}

constexpr int OS_SUB = 2;                  // 256-sample sub-tiles per wave tile (1: 5.1, 2: 6.0, 4: 5.6 TB/s -- profiles/r3/dyn_oneshot_ablation.txt)
constexpr int OS_WTILE = 64 * DE * OS_SUB; // 512 samples per wave (a multiple of 256: the history offsets assume it)
constexpr int OS_GTILE = OS_WTILE * (DT / 64);   // 2048 samples per workgroup

// One WAVE per 512-sample tile, four tiles per workgroup, no LDS and no barrier: the wave scans two 256-sample
// sub-tiles (each lane 4 consecutive samples, 6 shuffle steps per sub-tile), chains them through one scalar carry, and
// gets the state entering its tile from the history dot product (lanes 4 l < H, one predicated 16-byte load per channel).
// One wave tile of one row, in two steps so that a caller walking several rows can have the next row's loads in flight
// while it works on this one.  os_load: x and the H samples of history before the tile (lanes 4 l < H; none before the row
// start; s is a multiple of 512).  os_finish: the scans, the gain, the stores of y (and of the scan); returns what it stored.
struct OsIn {
    float xa[OS_SUB][DE], xb[OS_SUB][DE], ha[DE], hb[DE];
};

// LOOK-BACK tiles (round 5): a smoother memory longer than the history a tile may re-read (H > GFX_DYN_OS_HMAX samples,
// up to 64 tiles) does not send the row to the row kernel any more.  The scan is linear, so the state entering tile j is
//     u[s - 1] = sum_{i >= 1} a^(512 (i - 1)) A[j - i],      A[t] = the state tile t leaves from a ZERO entry state,
// and A[t] depends on tile t's own samples only: every tile publishes its aggregate as soon as its local scans are done
// -- before it needs anything from anybody -- and then reads the M = ceil(H / 512) aggregates before it (lane i polls tile
// j - 1 - i; a 64-lane weighted sum).  One hop of latency, no chain along the row, 8 bytes of traffic per tile and row.
//   * hand-off: one naturally aligned 8-byte {aggregate, 1} granule per (row, tile), written by ONE agent-scope relaxed
//     store and polled with agent-scope relaxed loads (both bypass the CU's L1; an 8-byte granule needs no fence);
//     the granules are zeroed by a memset node in front of the launch.
//   * progress: a tile only ever waits for tiles of the same row with smaller indices, which live in workgroups with
//     smaller logical indices.  Logical indices are handed out by TICKETS (one counter per blockIdx & 7, so that a
//     workgroup's tiles stay on the XCD the block index maps to): whoever holds ticket t knows tickets < t were drawn by
//     workgroups that are running or done -- no assumption about the order in which the hardware starts workgroups.
//     Tickets are drawn only when the pole table found a look-back row (ctrl[8]).
struct LbArgs {
    unsigned long long* gran;   // [row][tile]; nullptr: no look-back in this launch
    unsigned* ctrl;             // [0..7] tickets, [8] "some row looks back"
    unsigned ntiles;            // 512-sample tiles per row
    int split;                  // the routing-sum kernel is launched in both forms (plain / deferred walk), see there
};

__device__ __forceinline__ unsigned lb_block_index(const LbArgs& lb, unsigned per_xcd) {
    __shared__ unsigned ticket;
    unsigned bi = blockIdx.x >> 3;
    if (lb.gran && lb.ctrl[8] != 0u) {     // uniform over the grid
        if (threadIdx.x == 0) ticket = atomicAdd(&lb.ctrl[blockIdx.x & 7u], 1u);
        __syncthreads();
        bi = ticket;
    }
    return (blockIdx.x & 7u) * per_xcd + bi;
}

template <bool AL>
__device__ __forceinline__ void os_load(const DynArgs& a, const float* __restrict__ x0, const float* __restrict__ x1,
                                        bool vx, const float* __restrict__ tb, int64_t s, int lane, OsIn& in) {
    const bool stereo = a.C == 2;
    const int64_t n0 = s + DE * lane;
#pragma unroll
    for (int k = 0; k < OS_SUB; ++k) {
        ld4<AL>(x0, n0 + 256 * k, a.L, vx, in.xa[k]);
        if (stereo) ld4<AL>(x1, n0 + 256 * k, a.L, vx, in.xb[k]);
        else {
#pragma unroll
            for (int i = 0; i < DE; ++i) in.xb[k][i] = 0.0f;
        }
    }
    // history taps 4 l .. 4 l + 3 = samples s - 4 (l + 1) .. s - 4 l - 1
    const int H = (int)tb[DP_HIST];
    if (s != 0 && DE * lane < H) {
        ld4<AL>(x0, s - DE * (lane + 1), a.L, vx, in.ha);
        if (stereo) ld4<AL>(x1, s - DE * (lane + 1), a.L, vx, in.hb);
    }
}

// A tile in three steps, so that a caller walking several rows can put other rows' work between them:
//   os_scan    energies, the sub-tiles' local and in-wave scans; the state entering the tile from the re-read history
//              (one-shot rows) -- or, for a look-back row, the tile's aggregate PUBLISHED (see LbArgs)
//   os_lookback  the entry state of a look-back row from the aggregates of the tiles before it (polls them)
//   os_emit    envelope -> gain -> outputs, stores
struct OsMid {
    float loc[OS_SUB][DE], excl[OS_SUB], total[OS_SUB], carry;
};

template <bool AL>
__device__ __forceinline__ void os_scan(const DynArgs& a, const OsIn& in, const float* __restrict__ tb, int64_t s, int lane,
                                        OsMid& mid, unsigned long long* __restrict__ grow) {
    const bool stereo = a.C == 2;
    const float (&xa)[OS_SUB][DE] = in.xa;
    const float (&xb)[OS_SUB][DE] = in.xb;
    const int H = (int)tb[DP_HIST];
    const bool hist = s != 0 && DE * lane < H;
    const float a1 = tb[77], a_sub = tb[70];                             // a, a^256
    const float a_lane = tb[lane];                                       // a^(4 lane)
    float a_step[6];
#pragma unroll
    for (int d = 0; d < 6; ++d) a_step[d] = tb[64 + d];
    const float invC = 1.0f / (float)a.C;
    // the sub-tiles' local and in-wave scans do not depend on each other
#pragma unroll
    for (int k = 0; k < OS_SUB; ++k) {
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < DE; ++i) {
            const float e = (stereo ? (xa[k][i] * xa[k][i] + xb[k][i] * xb[k][i]) : xa[k][i] * xa[k][i]) * invC;
            acc = fmaf(a1, acc, e);
            mid.loc[k][i] = acc;
        }
        float inc = acc;
#pragma unroll
        for (int d = 0; d < 6; ++d) {
            const float up = __shfl_up(inc, 1 << d, 64);
            if (lane >= (1 << d)) inc = fmaf(a_step[d], up, inc);
        }
        const float ex = __shfl_up(inc, 1, 64);
        mid.excl[k] = lane == 0 ? 0.0f : ex;
        mid.total[k] = __shfl(inc, 63, 64);      // the sub-tile's aggregate, uniform
    }
    // state entering the tile: sum over the live taps of a^k e[s-1-k], k = 4 lane + (3 - i)
    float carry = 0.0f;
    if (grow) {                                   // look-back row (uniform): publish the tile's aggregate
        float agg = mid.total[0];
#pragma unroll
        for (int k = 1; k < OS_SUB; ++k) agg = fmaf(a_sub, agg, mid.total[k]);
        if (lane == 0)
            __hip_atomic_store(grow + (int)(s / OS_WTILE), (1ull << 32) | (unsigned long long)__float_as_uint(agg),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (s != 0 && H > 0) {                 // uniform
        float hs = 0.0f;
        if (hist) {
            float w = 0.0f;                       // Horner, oldest first: ((e0 a + e1) a + e2) a + e3
#pragma unroll
            for (int i = 0; i < DE; ++i) {
                const float eh = (stereo ? (in.ha[i] * in.ha[i] + in.hb[i] * in.hb[i]) : in.ha[i] * in.ha[i]) * invC;
                w = fmaf(a1, w, eh);
            }
            hs = w * a_lane;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) hs += __shfl_xor(hs, d, 64);
        carry = hs;
    }
    mid.carry = carry;
}

__device__ __forceinline__ float os_lookback(const float* __restrict__ tb, int64_t s, int lane,
                                              const unsigned long long* __restrict__ grow) {
    const int j = (int)(s / OS_WTILE);
    const int M = (int)tb[DP_LB_TILES];
    float hs = 0.0f;
    if (lane < M && lane < j) {
        const unsigned long long* g = grow + (j - 1 - lane);
        unsigned long long v = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while ((v >> 32) == 0ull) {
            __builtin_amdgcn_s_sleep(2);
            v = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        hs = __uint_as_float((unsigned)v) * tb[DP_LB_W + lane];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) hs += __shfl_xor(hs, d, 64);
    return hs;
}

template <bool AL>
__device__ __forceinline__ void os_emit(const DynArgs& a, const float (&xa)[OS_SUB][DE], const float (&xb)[OS_SUB][DE],
                                        const OsMid& mid, float carry, float* __restrict__ y0, float* __restrict__ y1,
                                        bool vx, float* __restrict__ u1row, const float* __restrict__ tb, const Knee& q,
                                        int64_t s, int lane, float (&ga)[OS_SUB][DE], float (&gb)[OS_SUB][DE]) {
    const bool stereo = a.C == 2;
    const int64_t n0 = s + DE * lane;
    const float one_m_a = tb[78], a_sub = tb[70];
    const float apk[DE] = {tb[73], tb[74], tb[75], tb[76]};              // a^1 .. a^4
    const float a_lane = tb[lane];
    const bool vu = (a.L % 4) == 0;
#pragma unroll
    for (int k = 0; k < OS_SUB; ++k) {
        const float pre = fmaf(a_lane, carry, mid.excl[k]);   // u just before this lane's first sample of sub-tile k
        carry = fmaf(a_sub, carry, mid.total[k]);
        float raw[DE];
#pragma unroll
        for (int i = 0; i < DE; ++i) {
            const float u = fmaf(apk[i], pre, mid.loc[k][i]);
            raw[i] = one_m_a * u;
            const float env = fmaxf(raw[i], 0.0f);                       // relu, envelope.py:48
            const float G = FastMath::log(env + 1e-5f);                  // dynamics.py:394
            const float g = FastMath::exp(log_gain_m<FastMath>(q, G));   // 402-403
            ga[k][i] = g * xa[k][i];
            gb[k][i] = g * xb[k][i];
        }
        const int64_t n = n0 + 256 * k;
        if (u1row) st4<AL>(u1row, n, a.L, vu, raw);
        if (y0) {                                     // (null: an output-only render that only sums this row)
            st4<AL>(y0, n, a.L, vx, ga[k]);
            if (stereo) st4<AL>(y1, n, a.L, vx, gb[k]);
        }
    }
}

// os_emit for a row whose local scans were NOT kept (the deferred walk of the routing-sum kernel keeps the samples, the
// exclusive in-wave scans and the sub-tile totals of its pending row, 18 registers + 2 scalars, and rebuilds the
// four-sample local scans here -- the same fused multiply-adds in the same order, so the same bits)
template <bool AL>
__device__ __forceinline__ void os_emit_lean(const DynArgs& a, const float (&xa)[OS_SUB][DE], const float (&xb)[OS_SUB][DE],
                                             const float (&excl)[OS_SUB], const float (&total)[OS_SUB], float carry,
                                             float* __restrict__ y0, float* __restrict__ y1, float* __restrict__ u1row,
                                             const float* __restrict__ tb, const Knee& q, int64_t s, int lane,
                                             float (&ga)[OS_SUB][DE], float (&gb)[OS_SUB][DE]) {
    const bool stereo = a.C == 2;
    const int64_t n0 = s + DE * lane;
    const float a1 = tb[77], one_m_a = tb[78], a_sub = tb[70];
    const float apk[DE] = {tb[73], tb[74], tb[75], tb[76]};              // a^1 .. a^4
    const float a_lane = tb[lane];
    const float invC = 1.0f / (float)a.C;
#pragma unroll
    for (int k = 0; k < OS_SUB; ++k) {
        const float pre = fmaf(a_lane, carry, excl[k]);
        carry = fmaf(a_sub, carry, total[k]);
        float raw[DE], acc = 0.0f;
#pragma unroll
        for (int i = 0; i < DE; ++i) {
            const float e = (stereo ? (xa[k][i] * xa[k][i] + xb[k][i] * xb[k][i]) : xa[k][i] * xa[k][i]) * invC;
            acc = fmaf(a1, acc, e);
            const float u = fmaf(apk[i], pre, acc);
            raw[i] = one_m_a * u;
            const float env = fmaxf(raw[i], 0.0f);
            const float G = FastMath::log(env + 1e-5f);
            const float g = FastMath::exp(log_gain_m<FastMath>(q, G));
            ga[k][i] = g * xa[k][i];
            gb[k][i] = g * xb[k][i];
        }
        const int64_t n = n0 + 256 * k;
        if (u1row) st4<AL>(u1row, n, a.L, true, raw);
        if (y0) {
            st4<AL>(y0, n, a.L, true, ga[k]);
            if (stereo) st4<AL>(y1, n, a.L, true, gb[k]);
        }
    }
}

template <bool AL>
__device__ __forceinline__ void os_finish(const DynArgs& a, const OsIn& in, float* __restrict__ y0, float* __restrict__ y1,
                                          bool vx, float* __restrict__ u1row, const float* __restrict__ tb, const Knee& q,
                                          int64_t s, int lane, float (&ga)[OS_SUB][DE], float (&gb)[OS_SUB][DE],
                                          unsigned long long* __restrict__ grow = nullptr) {
    OsMid mid;
    os_scan<AL>(a, in, tb, s, lane, mid, grow);
    const float carry = grow ? os_lookback(tb, s, lane, grow) : mid.carry;
    os_emit<AL>(a, in.xa, in.xb, mid, carry, y0, y1, vx, u1row, tb, q, s, lane, ga, gb);
}

__device__ __forceinline__ void os_tile(const DynArgs& a, const float* __restrict__ x0, const float* __restrict__ x1,
                                        float* __restrict__ y0, float* __restrict__ y1, float* __restrict__ u1row,
                                        const float* __restrict__ tb, const Knee& q, int64_t s, int lane,
                                        float (&ga)[OS_SUB][DE], float (&gb)[OS_SUB][DE],
                                        unsigned long long* __restrict__ grow) {
    const bool vx = vec_ok(x0) && vec_ok(x1) && vec_ok(y0) && vec_ok(y1);
    OsIn in;
    os_load<false>(a, x0, x1, vx, tb, s, lane, in);
    os_finish<false>(a, in, y0, y1, vx, u1row, tb, q, s, lane, ga, gb, grow);
}

__global__ __launch_bounds__(DT) void dyn_oneshot_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         const float* __restrict__ log_threshold,
                                                         const float* __restrict__ log_ratio,
                                                         const float* __restrict__ log_knee,
                                                         const float* __restrict__ tab, DynArgs a, unsigned ngroups,
                                                         unsigned nblocks, float* __restrict__ u1, LbArgs lb) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // workgroup b runs on XCD b % 8: give each XCD a contiguous run of tiles (a tile's history is its neighbour's data);
    // the runs are whole rows (the launcher pads the grid), so a row's tiles never straddle two runs
    const unsigned per_xcd = gridDim.x >> 3;
    const unsigned b = lb_block_index(lb, per_xcd);
    if (b >= nblocks) return;
    const unsigned r = b / ngroups;
    const unsigned grp = b - r * ngroups;
    const unsigned pr = r % a.prows;
    const float* tb = tab + (size_t)pr * DP_TAB;
    const bool looks_back = lb.gran && tb[DP_LOOKBACK] != 0.0f;   // (the launchers hand look-back rows to the row-group walk)
    if (tb[DP_ONESHOT] == 0.0f && !looks_back) return;          // produced by dyn_fused_kernel / the row-group walk (uniform)
    const int64_t s = (int64_t)grp * OS_GTILE + (int64_t)wave * OS_WTILE;     // first sample of this wave's tile
    if (s >= a.L) return;
    Knee q;
    knee_setup(q, log_threshold[pr], log_ratio[pr], log_knee ? log_knee[pr] : 0.0f, a.knee, a.gate);
    float ga[OS_SUB][DE], gb[OS_SUB][DE];
    os_tile(a, x + drow_off(a.xmap, r, 0), x + drow_off(a.xmap, r, a.C == 2 ? 1 : 0), y + drow_off(a.ymap, r, 0),
            y + drow_off(a.ymap, r, a.C == 2 ? 1 : 0), u1 ? u1 + (int64_t)r * a.L : nullptr, tb, q, s, lane, ga, gb,
            looks_back ? lb.gran + (size_t)r * lb.ntiles : nullptr);
}

// The one-shot tiles with the ROUTING SUM that follows fused in (render/core.py:36-112: the "mix" stage whose sources are
// exactly this stage's rows).  The rows of one graph (`inner` consecutive rows, r = g * inner + j) feed the mix
// destinations; a wave owns one 512-sample tile of ALL rows of its graph, walks them in increasing j -- the summation order
// of the gather-sum kernels, from 0.0f, so the sums are bit-identical to the separate pass -- producing each row's output
// as os_tile does and adding it to the accumulators of the destinations it feeds: the mix stage's re-read of every row
// (its whole traffic but the output rows) disappears.  A destination occupies one of NA accumulators only between its
// first and its last source (the host colours the live ranges: the console's four buses take turns in one accumulator,
// the send bus holds the other), and is stored right after its last source -- 32 accumulator registers instead of 16 per
// destination: 115 VGPRs, four waves per SIMD.  sched[j], per row of a graph: bits 0..3 = accumulators
// the row is added to; byte 1 + a = (destination + 1) to store accumulator a to, and clear it, after this row (0: none).
// Rows the pole table sends to dyn_fused_kernel have been written by it BEFORE this grid (the launcher orders it so) and
// are read back here.
struct MixArgs {
    const int64_t* sched;                 // [inner]
    float* out;                           // mix destinations: out + g * sb + j * sv + c * sc
    int64_t sb, sv, sc;
    int inner;
    // sources of the routing sum that are NOT rows of this stage (finished rows of the same buffer, e.g. the reverb return
    // next to the bus compressors in a master sum): pairs (row offset from the first destination row, code as in sched),
    // n_pre of them added before the stage's rows and n_post after -- the sum stays in increasing row order
    const int64_t* extras;
    int n_pre, n_post;
    // the stage's own rows are consumed by the routing sums alone (an output-only render: nobody reads them afterwards), so
    // the tiles do not store them -- rows the ROW kernel produces are still written (the tiles read them back from y)
    int skip_rows;
};

// Knee kind and compressor / gate are template parameters: a wave runs the row body `inner` times, and with every gain curve
// (and the element-wise tail paths of load4 / store4) inlined it is 6 k instructions, more than the instruction cache holds
// -- 5.0 ms for 8192 rows where this form takes 3.3-3.5.  Requesting rows ahead of the one being scanned was measured too
// (register rings of 2-4 rows): slower at every depth, the registers cost more waves than the loads in flight gain
// (EXPERIMENTS.md).
#ifndef GFX_DEFER_WAVES
#define GFX_DEFER_WAVES 1     // waves per SIMD the deferred walk is compiled for: left to the compiler (155 VGPRs, three waves,
                              // 4.19 ms for the 8192-row stage at a = 0.9975); 4 forces 128 VGPRs and 84 bytes of scratch: 4.31 ms
#endif
template <int NA, bool STEREO, int KIND, bool GATE, bool DEFER>
__global__ __launch_bounds__(DT, (DEFER && NA <= 2) ? GFX_DEFER_WAVES : 1) void dyn_oneshot_mix_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                             const float* __restrict__ log_threshold,
                                                             const float* __restrict__ log_ratio,
                                                             const float* __restrict__ log_knee,
                                                             const float* __restrict__ tab, DynArgs a, unsigned ngroups,
                                                             unsigned nblocks, float* __restrict__ u1, MixArgs m, LbArgs lb) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    // the launcher starts both forms of this kernel: the deferred walk takes the call when some row looks back, the plain
    // walk when none does (uniform over the grid; the other grid leaves at once)
    if (lb.split && (lb.gran && lb.ctrl[8] != 0u) != DEFER) return;
    const unsigned per_xcd = gridDim.x >> 3;
    const unsigned b = lb_block_index(lb, per_xcd);
    if (b >= nblocks) return;
    const unsigned g = b / ngroups;               // graph (batch index)
    const unsigned grp = b - g * ngroups;
    const int64_t s = (int64_t)grp * OS_GTILE + (int64_t)wave * OS_WTILE;
    if (s >= a.L) return;
    const int64_t n0 = s + DE * lane;
    // NA == 0: no routing sum at all -- the walk over groups of `inner` rows that produces the LOOK-BACK rows of a call
    // without a fused sum (every other row belongs to dyn_oneshot_kernel / dyn_fused_kernel and is skipped here)
    // The routing sums' accumulators: registers in the plain walk; in the deferred walk -- whose pending row costs it a wave
    // per SIMD otherwise (155 VGPRs) -- in LDS, which these kernels do not use for anything else: one 16-byte entry per
    // lane, accumulator, sub-tile and channel, lane-contiguous (conflict-free), 32 KB per workgroup for two stereo
    // accumulators.  A row touches the accumulators it feeds with one read-add-write each: ~16 LDS instructions per row
    // and lane against ~200 vector instructions.
#ifdef GFX_ACC_LDS_ALL   // A/B: the plain walk with LDS accumulators too
    constexpr bool ACC_LDS = NA > 0;
#else
    constexpr bool ACC_LDS = DEFER && NA > 0;
#endif
    constexpr int NCH = STEREO ? 2 : 1;
    using f4 = float __attribute__((ext_vector_type(4)));
    __shared__ f4 acc_lds[ACC_LDS ? NA * OS_SUB * NCH : 1][DT];
    float acc0[(!ACC_LDS && NA) ? NA : 1][OS_SUB][DE], acc1[(!ACC_LDS && STEREO && NA) ? NA : 1][OS_SUB][DE];
    if (ACC_LDS) {
#pragma unroll
        for (int i = 0; i < NA * OS_SUB * NCH; ++i) acc_lds[i][t] = f4{0.0f, 0.0f, 0.0f, 0.0f};
    } else {
#pragma unroll
        for (int c = 0; c < NA; ++c)
#pragma unroll
            for (int k = 0; k < OS_SUB; ++k)
#pragma unroll
                for (int i = 0; i < DE; ++i) {
                    acc0[c][k][i] = 0.0f;
                    if (STEREO) acc1[c][k][i] = 0.0f;
                }
    }
    float* const obase = NA ? m.out + (int64_t)g * m.sb : nullptr;
    // add one row's tile to the accumulators `code` names, then store and clear the destinations it completes
    auto settle = [&](uint64_t code, const float (&ga)[OS_SUB][DE], const float (&gb)[OS_SUB][DE]) {
        const unsigned add = (unsigned)code & 15u;
#pragma unroll
        for (int c = 0; c < NA; ++c) {
            const unsigned fl = (unsigned)(code >> (8 + 8 * c)) & 255u;
            if (ACC_LDS) {
                if ((((add >> c) & 1u) | fl) == 0u) continue;      // uniform: the row neither feeds nor completes it
                float* o0 = obase + (int64_t)(fl ? fl - 1u : 0u) * m.sv;
                float* o1 = o0 + m.sc;
#pragma unroll
                for (int k = 0; k < OS_SUB; ++k) {
                    f4 v0 = acc_lds[(c * OS_SUB + k) * NCH][t], v1 = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (STEREO) v1 = acc_lds[(c * OS_SUB + k) * NCH + 1][t];
                    if ((add >> c) & 1u) {
#pragma unroll
                        for (int i = 0; i < DE; ++i) {
                            v0[i] += ga[k][i];
                            if (STEREO) v1[i] += gb[k][i];
                        }
                    }
                    if (fl != 0u) {                   // this destination is complete: store, clear
                        const float s0[DE] = {v0[0], v0[1], v0[2], v0[3]}, s1[DE] = {v1[0], v1[1], v1[2], v1[3]};
                        st4<true>(o0, n0 + 256 * k, a.L, true, s0);
                        if (STEREO) st4<true>(o1, n0 + 256 * k, a.L, true, s1);
                        v0 = f4{0.0f, 0.0f, 0.0f, 0.0f};
                        v1 = v0;
                    }
                    acc_lds[(c * OS_SUB + k) * NCH][t] = v0;
                    if (STEREO) acc_lds[(c * OS_SUB + k) * NCH + 1][t] = v1;
                }
                continue;
            }
            if ((add >> c) & 1u) {                // uniform
#pragma unroll
                for (int k = 0; k < OS_SUB; ++k)
#pragma unroll
                    for (int i = 0; i < DE; ++i) {
                        acc0[c][k][i] += ga[k][i];
                        if (STEREO) acc1[c][k][i] += gb[k][i];
                    }
            }
            if (fl != 0u) {                       // uniform: this destination is complete
                float* o0 = obase + (int64_t)(fl - 1u) * m.sv;
                float* o1 = o0 + m.sc;
#pragma unroll
                for (int k = 0; k < OS_SUB; ++k) {
                    st4<true>(o0, n0 + 256 * k, a.L, true, acc0[c][k]);
                    if (STEREO) st4<true>(o1, n0 + 256 * k, a.L, true, acc1[c][k]);
#pragma unroll
                    for (int i = 0; i < DE; ++i) {
                        acc0[c][k][i] = 0.0f;
                        if (STEREO) acc1[c][k][i] = 0.0f;
                    }
                }
            }
        }
    };
    auto extra = [&](int e) {                     // a finished row of the buffer that joins the sum
        const float* p0 = obase + m.extras[2 * e] * m.sv;
        const float* p1 = p0 + m.sc;
        float ga[OS_SUB][DE], gb[OS_SUB][DE];
#pragma unroll
        for (int k = 0; k < OS_SUB; ++k) {
            ld4<true>(p0, n0 + 256 * k, a.L, true, ga[k]);
            if (STEREO) ld4<true>(p1, n0 + 256 * k, a.L, true, gb[k]);
        }
        settle((uint64_t)m.extras[2 * e + 1], ga, gb);
    };
    for (int e = 0; e < m.n_pre; ++e) extra(e);
    if (!DEFER) {
        for (int jr = 0; jr < m.inner; ++jr) {
            const unsigned r = g * (unsigned)m.inner + (unsigned)jr;
            const unsigned pr = r % a.prows;
            const float* tb = tab + (size_t)pr * DP_TAB;
            const uint64_t code = (uint64_t)m.sched[jr];
            float* y0 = y + drow_off(a.ymap, r, 0);
            float* y1 = y + drow_off(a.ymap, r, STEREO ? 1 : 0);
            float ga[OS_SUB][DE], gb[OS_SUB][DE];
            const bool looks_back = tb[DP_LOOKBACK] != 0.0f;
            if (tb[DP_ONESHOT] != 0.0f || looks_back) {   // uniform
                Knee q;
                knee_setup(q, log_threshold[pr], log_ratio[pr], KIND != 0 ? log_knee[pr] : 0.0f, KIND, GATE ? 1 : 0);
                OsIn in;
                os_load<true>(a, x + drow_off(a.xmap, r, 0), x + drow_off(a.xmap, r, STEREO ? 1 : 0), true, tb, s, lane, in);
                os_finish<true>(a, in, m.skip_rows ? nullptr : y0, y1, true, u1 ? u1 + (int64_t)r * a.L : nullptr, tb, q, s, lane,
                                ga, gb, looks_back ? lb.gran + (size_t)r * lb.ntiles : nullptr);
            } else if (((unsigned)code & 15u) != 0u) {   // the row kernel's row: read back what it wrote
#pragma unroll
                for (int k = 0; k < OS_SUB; ++k) {
                    ld4<true>(y0, n0 + 256 * k, a.L, true, ga[k]);
                    if (STEREO) ld4<true>(y1, n0 + 256 * k, a.L, true, gb[k]);
                }
            }
            settle(code, ga, gb);
        }
    } else {
        // The same walk, one row DEFERRED: row jr is loaded, scanned and -- if it looks back -- its aggregate published
        // before row jr - 1 is finished.  A look-back row then asks for its predecessors' aggregates a whole row time
        // after they were published (the waves of a graph walk the rows side by side), instead of right behind its own
        // publication, when the tiles before it are at the same point of the same row: without the deferral every row of
        // every wave waits out a hand-off (1-3 us of a ~6 us row).  The rows are still finished, and added to the
        // routing sums, in increasing order.
        float pxa[OS_SUB][DE], pxb[OS_SUB][DE], pexcl[OS_SUB];
        float ptotal[OS_SUB], pcarry = 0.0f;      // uniform
        int pkind = 0;                            // 0 nothing pending, 1 computed row, 2 look-back row, 3 read-back row
        unsigned prow = 0;
        uint64_t pcode = 0;
        auto finish = [&]() {
            if (pkind == 1 || pkind == 2) {       // uniform
                const unsigned pr = prow % a.prows;
                const float* tb = tab + (size_t)pr * DP_TAB;
                Knee q;
                knee_setup(q, log_threshold[pr], log_ratio[pr], KIND != 0 ? log_knee[pr] : 0.0f, KIND, GATE ? 1 : 0);
                const float carry = pkind == 2 ? os_lookback(tb, s, lane, lb.gran + (size_t)prow * lb.ntiles) : pcarry;
                float ga[OS_SUB][DE], gb[OS_SUB][DE];
                os_emit_lean<true>(a, pxa, pxb, pexcl, ptotal, carry, m.skip_rows ? nullptr : y + drow_off(a.ymap, prow, 0),
                                   y + drow_off(a.ymap, prow, STEREO ? 1 : 0), u1 ? u1 + (int64_t)prow * a.L : nullptr, tb, q, s,
                                   lane, ga, gb);
                settle(pcode, ga, gb);
            } else {
                settle(pcode, pxa, pxb);          // a row the row kernel wrote (read back below), or one nobody sums
            }
        };
        int walked = 0;
        for (int jr = 0; jr < m.inner; ++jr) {
            const unsigned r = g * (unsigned)m.inner + (unsigned)jr;
            if (NA == 0 && (int64_t)r >= a.R) break;      // (the last group of a call without a sum may be short)
            const unsigned pr = r % a.prows;
            const float* tb = tab + (size_t)pr * DP_TAB;
            const uint64_t code = NA ? (uint64_t)m.sched[jr] : 0ull;
            const bool looks_back = tb[DP_LOOKBACK] != 0.0f;
            OsIn in;
            float cexcl[OS_SUB], ctotal[OS_SUB], ccarry = 0.0f;
            int ckind = 3;
            if ((NA != 0 && tb[DP_ONESHOT] != 0.0f) || looks_back) {   // uniform
                os_load<true>(a, x + drow_off(a.xmap, r, 0), x + drow_off(a.xmap, r, STEREO ? 1 : 0), true, tb, s, lane, in);
                OsMid cmid;
                os_scan<true>(a, in, tb, s, lane, cmid, looks_back ? lb.gran + (size_t)r * lb.ntiles : nullptr);
#pragma unroll
                for (int k = 0; k < OS_SUB; ++k) {
                    cexcl[k] = cmid.excl[k];
                    ctotal[k] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, cmid.total[k])));
                }
                ccarry = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, cmid.carry)));
                ckind = looks_back ? 2 : 1;
            } else if (NA != 0 && ((unsigned)code & 15u) != 0u) {   // the row kernel's row: read back what it wrote
                const float* y0 = y + drow_off(a.ymap, r, 0);
                const float* y1 = y + drow_off(a.ymap, r, STEREO ? 1 : 0);
#pragma unroll
                for (int k = 0; k < OS_SUB; ++k) {
                    ld4<true>(y0, n0 + 256 * k, a.L, true, in.xa[k]);
                    if (STEREO) ld4<true>(y1, n0 + 256 * k, a.L, true, in.xb[k]);
                }
            }
            if (walked > 0) finish();
            ++walked;
#pragma unroll
            for (int k = 0; k < OS_SUB; ++k) {
#pragma unroll
                for (int i = 0; i < DE; ++i) {
                    pxa[k][i] = in.xa[k][i];
                    pxb[k][i] = in.xb[k][i];
                }
                pexcl[k] = cexcl[k];
                ptotal[k] = ctotal[k];
            }
            pcarry = ccarry;
            pkind = ckind;
            prow = r;
            pcode = code;
        }
        if (walked > 0) finish();
    }
    for (int e = m.n_pre; e < m.n_pre + m.n_post; ++e) extra(e);
}

// ---- standalone pieces (used when a configuration cannot take the fused kernel) --------------------
// energy: e[r,n] = mean_c x[r,c,n]^2
__global__ void energy_kernel(const float* __restrict__ x, gfx_rowmap_t xmap, float* __restrict__ e, int64_t R, int64_t L, int C) {
    const float invC = 1.0f / (float)C;
    for (int64_t r = blockIdx.y; r < R; r += gridDim.y) {
        const float* x0 = x + drow_off(xmap, r, 0);
        const float* x1 = x + drow_off(xmap, r, C == 2 ? 1 : 0);
        for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < L; n += (int64_t)gridDim.x * blockDim.x) {
            const float a = x0[n], b = x1[n];
            e[r * L + n] = (C == 2 ? (a * a + b * b) : a * a) * invC;
        }
    }
}

// truncated one-pole on (R, L) rows -> (R, Lout); Lout may extend to L + N - 1 (full convolution)
// ESRC (round 6): the rows are the energy mean_c x^2 of a signal read in place (dynamics.py:390) -- the envelope of a
// compressor whose smoother's convolve() aliases (upstream's default tap counts) no longer goes through an energy buffer.
// rowmax (nullable): receives the bits of max |out| of the row (one workgroup walks the row: a plain store), the by-product
// the odd-length aliasing's pair scaling asks for (czt_pair.hip).
template <bool TRUNC, bool ESRC>
__device__ __forceinline__ void onepole_stream(const OnePole& p, const float* u_in, const float* x1, int C, float* out, int64_t L,
                                               int64_t Lout, int64_t N, int relu, float* slots, int t, uint32_t* rowmax) {
    const int lane = t & 63, wave = t >> 6;
    const bool vi = vec_ok(u_in) && (!ESRC || vec_ok(x1)), vo = vec_ok(out);
    const float invC = 1.0f / (float)C;
    auto load_e = [&](int64_t n, bool vec, float (&e)[DE]) {
        load4(u_in, n, L, vec, e);
        if (ESRC) {
            float b[DE] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (C == 2) load4(x1, n, L, vec, b);
#pragma unroll
            for (int i = 0; i < DE; ++i) e[i] = (C == 2 ? (e[i] * e[i] + b[i] * b[i]) : e[i] * e[i]) * invC;
        }
    };
    float carry = 0.0f;
    uint32_t mx = 0;
    const int64_t ntiles = (Lout + DTILE - 1) / DTILE;
    float ne[DE];  // software prefetch of the next tile (see dyn_stream)
    load_e((int64_t)DE * t, vi, ne);
    for (int64_t tile = 0; tile < ntiles; ++tile) {
        const int64_t n = tile * DTILE + DE * t;
        float e[DE], u[DE];
#pragma unroll
        for (int i = 0; i < DE; ++i) e[i] = ne[i];
        if (tile + 1 < ntiles) load_e(n + DTILE, vi, ne);
        if (TRUNC) {  // one scan of e[n] - a^N e[n-N] (see dyn_stream)
            float e2[DE];
            load_e(n - N, false, e2);
#pragma unroll
            for (int i = 0; i < DE; ++i) e[i] = fmaf(-p.a_N, e2[i], e[i]);
        }
        scan_tile(p, e, u, carry, slots + 8 * (tile & 1), lane, wave);
#pragma unroll
        for (int i = 0; i < DE; ++i) {
            u[i] = p.one_m_a * u[i];
            if (relu) u[i] = fmaxf(u[i], 0.0f);
            const uint32_t b = __float_as_uint(u[i]) & 0x7fffffffu;
            if (rowmax && n + i < Lout) mx = b > mx ? b : mx;
        }
        store4(out, n, Lout, vo, u);
    }
    if (rowmax) {      // (uniform)
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const uint32_t v = (uint32_t)__shfl_xor((int)mx, o);
            mx = v > mx ? v : mx;
        }
        __syncthreads();
        if (lane == 0) slots[wave] = __uint_as_float(mx);
        __syncthreads();
        if (t == 0) {
            uint32_t m = 0;
            for (int w = 0; w < DT / 64; ++w) {
                const uint32_t v = __float_as_uint(slots[w]);
                m = v > m ? v : m;
            }
            *rowmax = m;
        }
    }
}

template <bool ESRC>
__global__ __launch_bounds__(DT) void onepole_kernel(const float* __restrict__ u, gfx_rowmap_t xmap, int C,
                                                     const float* __restrict__ z_alpha, float* __restrict__ out, int64_t L,
                                                     int64_t Lout, int64_t N, int relu, uint32_t* __restrict__ rowmax) {
    __shared__ float slots[16];
    const int t = threadIdx.x;
    const int64_t r = blockIdx.x;
    OnePole p;
    onepole_setup(p, z_alpha[r], N, t & 63);
    const float* in0 = ESRC ? u + drow_off(xmap, r, 0) : u + r * L;
    const float* in1 = ESRC ? u + drow_off(xmap, r, C == 2 ? 1 : 0) : nullptr;
    uint32_t* rm = rowmax ? rowmax + r : nullptr;
    // the FIR has exactly N taps: when Lout > L the tail still needs the a^N term once n >= N
    if (p.trunc)
        onepole_stream<true, ESRC>(p, in0, in1, C, out + r * Lout, L, Lout, N, relu, slots, t, rm);
    else
        onepole_stream<false, ESRC>(p, in0, in1, C, out + r * Lout, L, Lout, N, relu, slots, t, rm);
}

// one-pole FIR taps themselves, h[n] = (1-a) * exp(n * log a)  (envelope.py:51-60), for the generic conv path
__global__ void onepole_fir_kernel(const float* __restrict__ z_alpha, float* __restrict__ h, int64_t R, int64_t N) {
    for (int64_t r = blockIdx.y; r < R; r += gridDim.y) {
        const float a = fminf(sigmoidf(z_alpha[r]), 1.0f - 1e-5f);
        const float la = logf(a);
        for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x)
            h[r * N + n] = (1.0f - a) * expf((float)n * la);
    }
}

// (the ballistics recursion itself lives in ballistics.hip; its adjoint below keeps the 64 x 64 LDS tiling)
constexpr int BROWS = 64;   // (columns per tile: a template parameter of the kernel)

// Adjoint of the ballistics recursion (the branch choice c[n] is piecewise constant in the inputs):
//   lambda[n] = g[n] + (1 - c[n+1]) lambda[n+1];   gx[n] = c[n] lambda[n];
//   d/d(at) = sum over attack steps of lambda[n] (x[n] - y[n-1]),  d/d(rt) likewise over release steps,
// walked backwards in time with the same tiling as the forward kernel (x, y, g tiles in LDS, one lane per row).
// Chunked (gridDim.y > 1): the adjoint is a LINEAR recursion once the branch pattern is known, and a contraction -- a carry
// entering n steps later has shrunk by prod (1 - c) <= (1 - c_min)^n.  Workgroup (b, k) walks chunk k of its 64 rows and
// starts `warm` samples LATER in time with a zero carry (nothing stored, nothing summed there): the carry it reaches its own
// chunk with is exact to (1 - c_min)^warm <= 6e-10, warm = 21.2 / -log(1 - c_min) of the group's slowest coefficient, at most
// 2048 samples (c_min >= BWD_CMIN).  A group with a slower row (wave vote) takes the exact two-pass form of a linear scan
// instead: chunk aggregates, a chain over the chunks, then the walk with the true carries.  Per-chunk sums of the two coefficient gradients go to `part` (chunks x R x 2) and are
// added in chunk order by ballistics_bwd_finish_kernel: the same bits from run to run.  part == nullptr: one chunk, gz
// written directly (gfx_ballistics_bwd_f32).
constexpr float BWD_CMIN = 0.0103f;   // (1 - 0.0103)^2048 = 6e-10
constexpr int64_t BWD_WARM = 2048;

template <int BC, bool AGG = false>
__global__ __launch_bounds__(64) void ballistics_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                            const float* __restrict__ g,
                                                            const float* __restrict__ z_alpha, float* __restrict__ gx,
                                                            float* __restrict__ gz, int64_t R, int64_t L, int64_t chunk,
                                                            float* __restrict__ part, float* __restrict__ agg = nullptr) {
    constexpr int BCOLS = BC, BPAD = BC + 4, LPR = BC / 4, RPP = 64 / LPR;   // lanes per row, rows per cooperative pass
    __shared__ __attribute__((aligned(16))) float tx[BROWS * BPAD], ty[BROWS * BPAD], tg[BROWS * BPAD];
    const int lane = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * BROWS;
    const int64_t my = r0 + lane;
    float at = 0.0f, rt = 0.0f;
    if (my < R) {
        at = sigmoidf(z_alpha[2 * my]);
        rt = sigmoidf(z_alpha[2 * my + 1]);
    }
    float carry = 0.0f, sa = 0.0f, sr = 0.0f, prod = 1.0f;
    const int cr = lane / LPR, cc = (lane % LPR) * 4;
    const bool vec = (L % 4 == 0) && vec_ok(x) && vec_ok(y) && vec_ok(g) && vec_ok(gx);
    const int64_t ntiles = (L + BCOLS - 1) / BCOLS;
    // this workgroup's range of tiles: [t_lo, t_hi) are its own, [t_hi, t_top) the warm-up (chunk a multiple of BCOLS)
    int64_t t_lo = 0, t_hi = ntiles, t_top = ntiles;
    if (part) {
        const bool slow = __any(my < R && fminf(at, rt) < BWD_CMIN);
        const int64_t k = blockIdx.y, per = chunk / BCOLS;
        if (AGG && !slow) return;
        if (slow) {
            // no warm-up reaches far enough: the chunk's own tiles with a zero carry first (AGG: nothing stored; the product
            // of its (1 - c) and the carry it ends with go to `agg`), ballistics_bwd_carry_kernel chains the chunks of a row,
            // and the second launch starts every chunk with the carry that really enters it
            t_lo = k * per;
            t_hi = t_top = t_lo + per < ntiles ? t_lo + per : ntiles;
            if (!AGG && my < R) carry = agg[(k * R + my) * 2];
        } else {
            // warm-up of this group: (1 - c_min)^warm <= e^-21.2 = 6e-10 for its slowest coefficient, at most BWD_WARM
            float cmin = my < R ? fminf(at, rt) : 1.0f;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) cmin = fminf(cmin, __shfl_xor(cmin, d, 64));
            int64_t wt = (int64_t)ceilf(-21.2f / (log1pf(-fminf(cmin, 0.999f)) * (float)BCOLS));
            wt = wt < 1 ? 1 : (wt > BWD_WARM / BCOLS ? BWD_WARM / BCOLS : wt);
            t_lo = k * per;
            t_hi = t_lo + per < ntiles ? t_lo + per : ntiles;
            t_top = t_hi + wt < ntiles ? t_hi + wt : ntiles;
        }
    }
    for (int64_t tile = t_top - 1; tile >= t_lo; --tile) {
        const int64_t n0 = tile * BCOLS;
        const bool own = tile < t_hi;
#pragma unroll 4
        for (int pass = 0; pass < BROWS / RPP; ++pass) {
            const int row = pass * RPP + cr;
            const int64_t rr = r0 + row;
            float a[4] = {0.0f, 0.0f, 0.0f, 0.0f}, b[4] = {0.0f, 0.0f, 0.0f, 0.0f}, c[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (rr < R) {
                load4(x + rr * L, n0 + cc, L, vec, a);
                load4(y + rr * L, n0 + cc, L, vec, b);
                load4(g + rr * L, n0 + cc, L, vec, c);
            }
            *reinterpret_cast<float4*>(&tx[row * BPAD + cc]) = make_float4(a[0], a[1], a[2], a[3]);
            *reinterpret_cast<float4*>(&ty[row * BPAD + cc]) = make_float4(b[0], b[1], b[2], b[3]);
            *reinterpret_cast<float4*>(&tg[row * BPAD + cc]) = make_float4(c[0], c[1], c[2], c[3]);
        }
        const float y_before = (my < R && n0 > 0) ? y[my * L + n0 - 1] : 1.0f;  // y[-1] = 1
        __syncthreads();
        // four steps per LDS access (rows are 68 floats apart: sixteen lanes' 16-byte accesses tile the 64 banks); past the
        // row's end the tiles hold x = y = g = 0 -- a zero gradient entering a zero carry, whatever the branch
        const float ta = own ? 1.0f : 0.0f;
#pragma unroll 2
        for (int j4 = BCOLS / 4 - 1; j4 >= 0; --j4) {
            const float4 xq = *reinterpret_cast<const float4*>(&tx[lane * BPAD + 4 * j4]);
            const float4 yq = *reinterpret_cast<const float4*>(&ty[lane * BPAD + 4 * j4]);
            const float4 gq = *reinterpret_cast<const float4*>(&tg[lane * BPAD + 4 * j4]);
            const float y0 = j4 > 0 ? ty[lane * BPAD + 4 * j4 - 1] : y_before;
            const float xs[4] = {xq.x, xq.y, xq.z, xq.w}, yp[4] = {y0, yq.x, yq.y, yq.z}, gs[4] = {gq.x, gq.y, gq.z, gq.w};
            float o[4];
#pragma unroll
            for (int i = 3; i >= 0; --i) {
                const bool attack = xs[i] < yp[i];
                const float c = attack ? at : rt;
                const float lam = gs[i] + carry;
                o[i] = c * lam;
                const float d = ta * lam * (xs[i] - yp[i]);
                sa += attack ? d : 0.0f;
                sr += attack ? 0.0f : d;
                carry = (1.0f - c) * lam;
                if (AGG) prod *= 1.0f - c;
            }
            if (!AGG) *reinterpret_cast<float4*>(&tg[lane * BPAD + 4 * j4]) = make_float4(o[0], o[1], o[2], o[3]);
        }
        if (AGG) {
            __syncthreads();
            continue;
        }
        __syncthreads();
#pragma unroll 4
        for (int pass = 0; pass < BROWS / RPP; ++pass) {
            const int row = pass * RPP + cr;
            const int64_t rr = r0 + row;
            if (rr < R && own) {
                const float4 q = *reinterpret_cast<const float4*>(&tg[row * BPAD + cc]);
                const float v[4] = {q.x, q.y, q.z, q.w};
                store4(gx + rr * L, n0 + cc, L, vec, v);
            }
        }
        __syncthreads();
    }
    if (AGG) {
        if (my < R) {
            agg[((int64_t)blockIdx.y * R + my) * 2] = prod;
            agg[((int64_t)blockIdx.y * R + my) * 2 + 1] = carry;
        }
        return;
    }
    if (my < R) {
        if (part) {
            part[((int64_t)blockIdx.y * R + my) * 2] = sa;
            part[((int64_t)blockIdx.y * R + my) * 2 + 1] = sr;
        } else {
            gz[2 * my] = sa * at * (1.0f - at);
            gz[2 * my + 1] = sr * rt * (1.0f - rt);
        }
    }
}

// agg[k][r] = (product of (1 - c) over chunk k, the carry chunk k ends with from a zero carry)  ->  agg[k][r][0] = the carry
// that enters chunk k: carry_in[last] = 0, carry_in[k] = end[k + 1] + prod[k + 1] carry_in[k + 1]  (rows of fast groups hold
// nothing meaningful and are not read)
__global__ void ballistics_bwd_carry_kernel(float* __restrict__ agg, int64_t R, int chunks) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    float cin = 0.0f;
    for (int k = chunks - 1; k >= 0; --k) {
        const float p = agg[((int64_t)k * R + r) * 2], e = agg[((int64_t)k * R + r) * 2 + 1];
        agg[((int64_t)k * R + r) * 2] = cin;
        cin = e + p * cin;
    }
}

__global__ void ballistics_bwd_finish_kernel(const float* __restrict__ part, const float* __restrict__ z_alpha,
                                             float* __restrict__ gz, int64_t R, int chunks) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // (row, which coefficient)
    if (i >= 2 * R) return;
    float s = 0.0f;
    for (int k = 0; k < chunks; ++k) s += part[(int64_t)k * 2 * R + i];
    const float c = sigmoidf(z_alpha[i]);
    gz[i] = s * c * (1.0f - c);
}

// env (R,L) -> gain (R,L):  g = log_gain(log(env + 1e-5));  out = exp(g) or g (log_out)
__global__ void dyn_gain_kernel(const float* __restrict__ env, float* __restrict__ gain,
                                const float* __restrict__ log_threshold, const float* __restrict__ log_ratio,
                                const float* __restrict__ log_knee, int64_t R, int64_t L, int knee, int gate,
                                int log_out) {
    for (int64_t r = blockIdx.y; r < R; r += gridDim.y) {
        Knee q;
        knee_setup(q, log_threshold[r], log_ratio[r], log_knee ? log_knee[r] : 0.0f, knee, gate);
        for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < L; n += (int64_t)gridDim.x * blockDim.x) {
            const float g = log_gain(q, logf(env[r * L + n] + 1e-5f));
            gain[r * L + n] = log_out ? g : expf(g);
        }
    }
}

// y[r,c,n] = (exp_gain ? exp(g[r,n]) : g[r,n]) * x[r,c,n]
__global__ void apply_gain_kernel(const float* __restrict__ x, gfx_rowmap_t xmap, const float* __restrict__ g,
                                  float* __restrict__ y, gfx_rowmap_t ymap, int64_t R, int64_t L, int C,
                                  int exp_gain) {
    for (int64_t r = blockIdx.y; r < R; r += gridDim.y)
    for (int c = 0; c < C; ++c) {
        const float* xr = x + drow_off(xmap, r, c);
        float* yr = y + drow_off(ymap, r, c);
        for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < L; n += (int64_t)gridDim.x * blockDim.x) {
            const float gv = g[r * L + n];
            yr[n] = (exp_gain ? expf(gv) : gv) * xr[n];
        }
    }
}

// y[r,c,n] = exp(log_gain(log(env[r,n] + 1e-5))) * x[r,c,n]: gain computer and gain stage in one pass over an envelope
// that a smoother kernel left in memory (the ballistics configurations: dynamics.py:394-405 behind core/envelope.py:84-101).
// Four samples per thread, 16-byte accesses when the rows allow it.
__global__ __launch_bounds__(256) void dyn_gain_apply_kernel(const float* __restrict__ x, gfx_rowmap_t xmap,
                                                             const float* __restrict__ env, float* __restrict__ y,
                                                             gfx_rowmap_t ymap, const float* __restrict__ log_threshold,
                                                             const float* __restrict__ log_ratio,
                                                             const float* __restrict__ log_knee, int64_t R, int64_t L, int C,
                                                             int knee, int gate, unsigned prows, int vec) {
    for (int64_t r = blockIdx.y; r < R; r += gridDim.y) {
        const unsigned pr = (unsigned)r % prows;
        Knee q;
        knee_setup(q, log_threshold[pr], log_ratio[pr], log_knee ? log_knee[pr] : 0.0f, knee, gate);
        const float* x0 = x + drow_off(xmap, r, 0);
        const float* x1 = x + drow_off(xmap, r, C == 2 ? 1 : 0);
        float* y0 = y + drow_off(ymap, r, 0);
        float* y1 = y + drow_off(ymap, r, C == 2 ? 1 : 0);
        const float* er = env + r * L;
        for (int64_t n = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * DE; n < L; n += (int64_t)gridDim.x * blockDim.x * DE) {
            float e[DE], a[DE], b[DE] = {0.0f, 0.0f, 0.0f, 0.0f};
            load4(er, n, L, vec, e);
            load4(x0, n, L, vec, a);
            if (C == 2) load4(x1, n, L, vec, b);
#pragma unroll
            for (int i = 0; i < DE; ++i) {
                const float g = expf(log_gain(q, logf(e[i] + 1e-5f)));
                a[i] *= g;
                b[i] *= g;
            }
            store4(y0, n, L, vec, a);
            if (C == 2) store4(y1, n, L, vec, b);
        }
    }
}

// ---- backward of the gain computer (training path of Compressor / NoiseGate) --------------------------------
// Partial derivatives of g = log_gain(G) (dynamics.py:444-489 compressor, 676-721 gate) w.r.t. G, the threshold T,
// log_ratio and log_knee.  The region masks are piecewise constant, as in torch's autograd of the same expressions.
struct KneeGrad {
    float dG, dT, dlr, dlk;
};
__device__ __forceinline__ KneeGrad log_gain_grad(const Knee& q, float G) {
    KneeGrad o = {0.0f, 0.0f, 0.0f, 0.0f};
    const float d = G - q.T;
    if (!q.gate) {
        const float c = q.invR - 1.0f;  // (1/R - 1)
        if (q.kind == 0 || (q.kind == 1 && G > q.T + q.W)) {
            if (q.kind == 1 || d > 0.0f) {  // above the threshold: g = (1/R - 1)(G - T)
                o.dG = c;
                o.dT = -c;
                o.dlr = -d * q.invR * q.invR * q.er;
            }
        } else if (q.kind == 1) {
            if (!(G < q.T - q.W)) {  // knee region: g = c s^2 / (4W), s = G - T + W
                const float s = d + q.W, h = s / (2.0f * q.W);
                o.dG = c * h;
                o.dT = -c * h;
                o.dlr = -q.invR * q.invR * s * s / (4.0f * q.W) * q.er;
                o.dlk = c * (h - h * h) * q.W;  // dg/dW * dW/dlk,  W = exp(lk)/2
            }
        } else {  // exponential: g = c softplus(k d) / k
            const float v = q.k * d;
            const float sp = softplusf(v), sg = v > 20.0f ? 1.0f : sigmoidf(v);
            o.dG = c * sg;
            o.dT = -c * sg;
            o.dlr = -q.invR * q.invR * sp / q.k * q.er;
            o.dlk = c * (sg * v - sp) / q.k;  // dg/dk * k
        }
    } else {
        const float c = 1.0f - q.R;  // (1 - R) = -exp(lr)
        if (q.kind == 0 || (q.kind == 1 && G < q.T - q.W)) {
            if (q.kind == 1 || d < 0.0f) {  // below the threshold: g = (R - 1)(G - T)
                o.dG = -c;
                o.dT = c;
                o.dlr = d * q.er;
            }
        } else if (q.kind == 1) {
            if (!(G > q.T + q.W)) {  // knee region: g = c s^2 / (4W), s = G - T - W
                const float s = d - q.W, h = s / (2.0f * q.W);
                o.dG = c * h;
                o.dT = -c * h;
                o.dlr = -s * s / (4.0f * q.W) * q.er;
                o.dlk = c * (-h - h * h) * q.W;
            }
        } else {  // exponential: g = -er softplus(k (T - G)) / k
            const float v = -q.k * d;
            const float sp = softplusf(v), sg = v > 20.0f ? 1.0f : sigmoidf(v);
            o.dG = q.er * sg;
            o.dT = -q.er * sg;
            o.dlr = -q.er * sp / q.k;
            o.dlk = -q.er * (sg * v - sp) / q.k;
        }
    }
    return o;
}

// ---- fused backward of the compressor / gate with the one-pole energy smoother ------------------------------
// Pass A, forward in time, one workgroup per row: recompute energy -> smoothed energy -> gain exactly as the
// forward kernel does, and from the output gradient gy emit
//   denv[n] = dL/d(smoothed energy), relu-masked       (R, L)
//   u1[n]   = (1-a) * (untruncated scan of the energy) (R, L)   -- input of the pole-gradient reduction
//   gparams[r, 0..2] = dL/d(log_threshold, log_ratio, log_knee)
template <bool TRUNC>
__device__ __forceinline__ void dyn_bwd_a_stream(const DynArgs& a, const OnePole& p, const Knee& q, const float* x0,
                                                 const float* x1, const float* g0, const float* g1,
                                                 float* denv, float* u1, float* slots, int t, float (&acc)[3]) {
    const int lane = t & 63, wave = t >> 6;
    const bool vx = vec_ok(x0) && vec_ok(x1) && vec_ok(g0) && vec_ok(g1), vo = (a.L % 4) == 0;
    const float invC = 1.0f / (float)a.C;
    float carry = 0.0f, carry2 = 0.0f;
    const int64_t ntiles = (a.L + DTILE - 1) / DTILE;
    // software prefetch, as in dyn_stream: the next tile's samples are requested before this tile is scanned
    float nxa[DE], nxb[DE] = {0.0f, 0.0f, 0.0f, 0.0f}, nga[DE], ngb[DE] = {0.0f, 0.0f, 0.0f, 0.0f};
    load4(x0, (int64_t)DE * t, a.L, vx, nxa);
    load4(g0, (int64_t)DE * t, a.L, vx, nga);
    if (a.C == 2) {
        load4(x1, (int64_t)DE * t, a.L, vx, nxb);
        load4(g1, (int64_t)DE * t, a.L, vx, ngb);
    }
    for (int64_t tile = 0; tile < ntiles; ++tile) {
        const int64_t n = tile * DTILE + DE * t;
        float xa[DE], xb[DE], ga[DE], gb[DE], e[DE], u[DE];
#pragma unroll
        for (int i = 0; i < DE; ++i) {
            xa[i] = nxa[i];
            xb[i] = nxb[i];
            ga[i] = nga[i];
            gb[i] = ngb[i];
        }
        if (tile + 1 < ntiles) {
            load4(x0, n + DTILE, a.L, vx, nxa);
            load4(g0, n + DTILE, a.L, vx, nga);
            if (a.C == 2) {
                load4(x1, n + DTILE, a.L, vx, nxb);
                load4(g1, n + DTILE, a.L, vx, ngb);
            }
        }
#pragma unroll
        for (int i = 0; i < DE; ++i) e[i] = (a.C == 2 ? (xa[i] * xa[i] + xb[i] * xb[i]) : xa[i] * xa[i]) * invC;
        scan_tile(p, e, u, carry, slots + 8 * (tile & 1), lane, wave);
        float lin[DE], raw[DE];
#pragma unroll
        for (int i = 0; i < DE; ++i) raw[i] = p.one_m_a * u[i];
        if (TRUNC) {
            float da[DE], db[DE], e2[DE], u2[DE];
            load4(x0, n - a.N, a.L, false, da);
            if (a.C == 2) load4(x1, n - a.N, a.L, false, db);
#pragma unroll
            for (int i = 0; i < DE; ++i)
                e2[i] = (a.C == 2 ? (da[i] * da[i] + db[i] * db[i]) : da[i] * da[i]) * invC;
            scan_tile(p, e2, u2, carry2, slots + 8 * (tile & 1) + 4, lane, wave);
#pragma unroll
            for (int i = 0; i < DE; ++i) u[i] = fmaf(-p.a_N, u2[i], u[i]);
        }
        float dv[DE];
#pragma unroll
        for (int i = 0; i < DE; ++i) {
            lin[i] = p.one_m_a * u[i];
            const float env = fmaxf(lin[i], 0.0f);
            const float G = logf(env + 1e-5f);
            const float gn = expf(log_gain(q, G));
            const float dgain = a.C == 2 ? (ga[i] * xa[i] + gb[i] * xb[i]) : ga[i] * xa[i];
            const float dg = dgain * gn;
            const KneeGrad k = log_gain_grad(q, G);
            dv[i] = lin[i] > 0.0f ? dg * k.dG / (env + 1e-5f) : 0.0f;
            if (n + i < a.L) {
                acc[0] += dg * k.dT;
                acc[1] += dg * k.dlr;
                acc[2] += dg * k.dlk;
            }
        }
        store4(denv, n, a.L, vo, dv);
        store4(u1, n, a.L, vo, raw);
    }
}

__global__ __launch_bounds__(DT) void dyn_bwd_a_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                       gfx_rowmap_t gmap, const float* __restrict__ log_threshold,
                                                       const float* __restrict__ log_ratio,
                                                       const float* __restrict__ log_knee,
                                                       const float* __restrict__ z_alpha,
                                                       float* __restrict__ denv, float* __restrict__ u1,
                                                       float* __restrict__ gparams, DynArgs a) {
    __shared__ float slots[16];
    __shared__ float red[3][4];
    const int t = threadIdx.x;
    const int64_t r = blockIdx.x;
    OnePole p;
    onepole_setup(p, z_alpha[r], a.N, t & 63);
    Knee q;
    knee_setup(q, log_threshold[r], log_ratio[r], log_knee ? log_knee[r] : 0.0f, a.knee, a.gate);
    const float* x0 = x + drow_off(a.xmap, r, 0);
    const float* x1 = x + drow_off(a.xmap, r, a.C == 2 ? 1 : 0);
    const float* g0 = gy + drow_off(gmap, r, 0);
    const float* g1 = gy + drow_off(gmap, r, a.C == 2 ? 1 : 0);
    float acc[3] = {0.0f, 0.0f, 0.0f};
    if (p.trunc)
        dyn_bwd_a_stream<true>(a, p, q, x0, x1, g0, g1, denv + r * a.L, u1 + r * a.L, slots, t, acc);
    else
        dyn_bwd_a_stream<false>(a, p, q, x0, x1, g0, g1, denv + r * a.L, u1 + r * a.L, slots, t, acc);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float v = acc[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if ((t & 63) == 0) red[k][t >> 6] = v;
    }
    __syncthreads();
    if (t < 3) gparams[3 * r + t] = red[t][0] + red[t][1] + red[t][2] + red[t][3];
}

// Pass B, backward in time: de[m] = sum_{k<N} h[k] denv[m+k] (the smoother's adjoint = the same scan on the
// reversed sequence), then gx = gain * gy + (2/C) * de * x with the gain recomputed from pass A's scan
// (env = relu(u1[m] - a^N u1[m-N])): cheaper than a (R, L) gain array written by A and read back here.
// Position j of the reversed walk is sample L-1-j.
__device__ __forceinline__ void rload4(const float* __restrict__ row, int64_t j, int64_t L, bool vec, float (&v)[DE]) {
    // v[i] = row[L-1-(j+i)], zero outside [0, L)
    const int64_t hi = L - 1 - j;  // sample of v[0]
    if (vec && hi - 3 >= 0 && hi < L) {
        const float4 q = *reinterpret_cast<const float4*>(row + hi - 3);
        v[0] = q.w; v[1] = q.z; v[2] = q.y; v[3] = q.x;
    } else {
#pragma unroll
        for (int i = 0; i < DE; ++i) v[i] = (hi - i >= 0 && hi - i < L) ? row[hi - i] : 0.0f;
    }
}
__device__ __forceinline__ void rstore4(float* __restrict__ row, int64_t j, int64_t L, bool vec, const float (&v)[DE]) {
    const int64_t hi = L - 1 - j;
    if (vec && hi - 3 >= 0 && hi < L) {
        *reinterpret_cast<float4*>(row + hi - 3) = make_float4(v[3], v[2], v[1], v[0]);
    } else {
#pragma unroll
        for (int i = 0; i < DE; ++i)
            if (hi - i >= 0 && hi - i < L) row[hi - i] = v[i];
    }
}

// POLE: also accumulate the pole gradient of the truncated smoother.  With U = u1 / (1-a) (pass A's un-truncated scan),
// g = denv and de = this pass's adjoint scan,
//   dL/da = sum_m  -g[m] U[m] + (a^N - (1-a) N a^(N-1)) g[m] U[m-N] + de[m] U[m-1]
// (the last term is sum_n g[n] (1-a) (D[n] - a^N D[n-N]), D = dU/da, moved onto the adjoint scan: D is a scan of U,
// so pairing it with g equals pairing U with the backward scan of g, which is de one sample later).
template <bool TRUNC, bool POLE>
__device__ __forceinline__ void dyn_bwd_b_stream(const DynArgs& a, const OnePole& p, const Knee& q, const float* x0,
                                                 const float* x1, const float* g0, const float* g1,
                                                 const float* denv, const float* u1, float* o0, float* o1,
                                                 float* slots, int t, float& pole) {
    const int lane = t & 63, wave = t >> 6;
    const bool al = (a.L % 4) == 0;  // reversed float4 groups stay 16-byte aligned only then
    const bool vx = al && vec_ok(x0) && vec_ok(x1) && vec_ok(g0) && vec_ok(g1), vo = al;
    const bool vgx = al && vec_ok(o0) && vec_ok(o1);
    const float k2 = 2.0f / (float)a.C;
    const float pole_c2 = p.a_N - p.one_m_a * (float)a.N * (p.a_N / p.a);
    float carry = 0.0f, carry2 = 0.0f;
    const int64_t ntiles = (a.L + DTILE - 1) / DTILE;
    // software prefetch: this tile's other operands and the next tile's denv are requested before the scan (whose
    // barrier would otherwise fence them), so their HBM round trips overlap the scan and the gain arithmetic
    float nd[DE];
    rload4(denv, (int64_t)DE * t, a.L, vo, nd);
    for (int64_t tile = 0; tile < ntiles; ++tile) {
        const int64_t j = tile * DTILE + DE * t;
        float d[DE], u[DE];
#pragma unroll
        for (int i = 0; i < DE; ++i) d[i] = nd[i];
        if (tile + 1 < ntiles) rload4(denv, j + DTILE, a.L, vo, nd);
        float uu[DE], xa[DE], ga[DE], xb[DE] = {0.0f, 0.0f, 0.0f, 0.0f}, gb[DE] = {0.0f, 0.0f, 0.0f, 0.0f};
        rload4(u1, j, a.L, vo, uu);
        rload4(x0, j, a.L, vx, xa);
        rload4(g0, j, a.L, vx, ga);
        if (a.C == 2) {
            rload4(x1, j, a.L, vx, xb);
            rload4(g1, j, a.L, vx, gb);
        }
        scan_tile(p, d, u, carry, slots + 8 * (tile & 1), lane, wave);
        if (TRUNC) {
            float d2[DE], u2[DE];
            rload4(denv, j - a.N, a.L, false, d2);
            scan_tile(p, d2, u2, carry2, slots + 8 * (tile & 1) + 4, lane, wave);
#pragma unroll
            for (int i = 0; i < DE; ++i) u[i] = fmaf(-p.a_N, u2[i], u[i]);
        }
        float un[DE] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (TRUNC) rload4(u1, j + a.N, a.L, false, un);
        if (POLE) {
            const int64_t below = a.L - 1 - j - DE;  // sample under this thread's four
            const float um = (below >= 0 && below < a.L) ? u1[below] : 0.0f;
#pragma unroll
            for (int i = 0; i < DE; ++i) {
                const float prev = i + 1 < DE ? uu[i + 1] : um;
                pole += p.one_m_a * u[i] * prev - d[i] * uu[i];
                if (TRUNC) pole = fmaf(pole_c2 * d[i], un[i], pole);
            }
        }
        float gn[DE], oa[DE];
#pragma unroll
        for (int i = 0; i < DE; ++i) {
            const float lin = TRUNC ? fmaf(-p.a_N, un[i], uu[i]) : uu[i];
            gn[i] = expf(log_gain(q, logf(fmaxf(lin, 0.0f) + 1e-5f)));
        }
#pragma unroll
        for (int i = 0; i < DE; ++i) oa[i] = fmaf(gn[i], ga[i], k2 * p.one_m_a * u[i] * xa[i]);
        rstore4(o0, j, a.L, vgx, oa);
        if (a.C == 2) {
            float ob[DE];
#pragma unroll
            for (int i = 0; i < DE; ++i) ob[i] = fmaf(gn[i], gb[i], k2 * p.one_m_a * u[i] * xb[i]);
            rstore4(o1, j, a.L, vgx, ob);
        }
    }
}

__global__ __launch_bounds__(DT) void dyn_bwd_b_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                       gfx_rowmap_t gmap, const float* __restrict__ log_threshold,
                                                       const float* __restrict__ log_ratio,
                                                       const float* __restrict__ log_knee,
                                                       const float* __restrict__ z_alpha, const float* __restrict__ denv,
                                                       const float* __restrict__ u1, float* __restrict__ dalpha,
                                                       float* __restrict__ gx, DynArgs a) {
    __shared__ float slots[16];
    __shared__ float red[4];
    const int t = threadIdx.x;
    const int64_t r = blockIdx.x;
    OnePole p;
    onepole_setup(p, z_alpha[r], a.N, t & 63);
    const float* x0 = x + drow_off(a.xmap, r, 0);
    const float* x1 = x + drow_off(a.xmap, r, a.C == 2 ? 1 : 0);
    const float* g0 = gy + drow_off(gmap, r, 0);
    const float* g1 = gy + drow_off(gmap, r, a.C == 2 ? 1 : 0);
    float* o0 = gx + drow_off(a.ymap, r, 0);
    float* o1 = gx + drow_off(a.ymap, r, a.C == 2 ? 1 : 0);
    Knee q;
    knee_setup(q, log_threshold[r], log_ratio[r], log_knee ? log_knee[r] : 0.0f, a.knee, a.gate);
    float pole = 0.0f;
    const float* ur = u1 + r * a.L;
    if (dalpha) {
        if (p.trunc)
            dyn_bwd_b_stream<true, true>(a, p, q, x0, x1, g0, g1, denv + r * a.L, ur, o0, o1, slots, t, pole);
        else
            dyn_bwd_b_stream<false, true>(a, p, q, x0, x1, g0, g1, denv + r * a.L, ur, o0, o1, slots, t, pole);
        for (int o = 32; o > 0; o >>= 1) pole += __shfl_down(pole, o, 64);
        if ((t & 63) == 0) red[t >> 6] = pole;
        __syncthreads();
        if (t == 0) dalpha[r] = (red[0] + red[1] + red[2] + red[3]) / p.one_m_a;  // u1 = (1-a) U
    } else if (p.trunc) {
        dyn_bwd_b_stream<true, false>(a, p, q, x0, x1, g0, g1, denv + r * a.L, ur, o0, o1, slots, t, pole);
    } else {
        dyn_bwd_b_stream<false, false>(a, p, q, x0, x1, g0, g1, denv + r * a.L, ur, o0, o1, slots, t, pole);
    }
}

// ---- the same backward in fewer bytes (round 2): pass A' only scans the energy (x -> u1: no gy, no denv), and pass C --
// pass B with the gain computer's derivatives folded in -- recomputes denv where it needs it from (x, gy, u1):
//   A' reads 8 B and writes 4 B per stereo sample; C reads 20 B (x, gy, u1) and writes 8 B: 40 B instead of 56 B.
// For rows whose truncation term is live, C also needs denv N samples later for its second scan and recomputes it from
// a second set of loads there.
__device__ __forceinline__ void dyn_bwd_u1_stream(const DynArgs& a, const OnePole& p, const float* x0, const float* x1,
                                                  float* u1, float* slots, int t) {
    const int lane = t & 63, wave = t >> 6;
    const bool vx = vec_ok(x0) && vec_ok(x1), vo = (a.L % 4) == 0;
    const float invC = 1.0f / (float)a.C;
    float carry = 0.0f;
    const int64_t ntiles = (a.L + DTILE - 1) / DTILE;
    float nxa[DE], nxb[DE] = {0.0f, 0.0f, 0.0f, 0.0f};
    load4(x0, (int64_t)DE * t, a.L, vx, nxa);
    if (a.C == 2) load4(x1, (int64_t)DE * t, a.L, vx, nxb);
    for (int64_t tile = 0; tile < ntiles; ++tile) {
        const int64_t n = tile * DTILE + DE * t;
        float xa[DE], xb[DE], e[DE], u[DE], raw[DE];
#pragma unroll
        for (int i = 0; i < DE; ++i) {
            xa[i] = nxa[i];
            xb[i] = nxb[i];
        }
        if (tile + 1 < ntiles) {
            load4(x0, n + DTILE, a.L, vx, nxa);
            if (a.C == 2) load4(x1, n + DTILE, a.L, vx, nxb);
        }
#pragma unroll
        for (int i = 0; i < DE; ++i) e[i] = (a.C == 2 ? (xa[i] * xa[i] + xb[i] * xb[i]) : xa[i] * xa[i]) * invC;
        scan_tile(p, e, u, carry, slots + 8 * (tile & 1), lane, wave);
#pragma unroll
        for (int i = 0; i < DE; ++i) raw[i] = p.one_m_a * u[i];
        store4(u1, n, a.L, vo, raw);
    }
}

__global__ __launch_bounds__(DT) void dyn_bwd_u1_kernel(const float* __restrict__ x, const float* __restrict__ z_alpha,
                                                        float* __restrict__ u1, DynArgs a,
                                                        const float* __restrict__ tab = nullptr) {
    __shared__ float slots[16];
    const int t = threadIdx.x;
    const int64_t r = blockIdx.x;
    if (tab && tab[(size_t)r * DP_TAB + DP_ONESHOT] != 0.0f) return;   // a one-shot row rebuilds its scan in its own tiles
    OnePole p;
    onepole_setup(p, z_alpha[r], a.N, t & 63);
    dyn_bwd_u1_stream(a, p, x + drow_off(a.xmap, r, 0), x + drow_off(a.xmap, r, a.C == 2 ? 1 : 0), u1 + r * a.L, slots, t);
}

// dL/d(smoothed energy) at four (reversed-walk) positions from the samples, output gradients and scan values there;
// also returns the gain and, when `acc` is given, adds the parameter-gradient terms (as pass A did).
// (A: float in the tiles -- eight terms per thread and launch, the sums continue in double --, double in the row kernel, where
// a thread adds hundreds of terms of both signs)
template <bool FAST = false, typename A = float>
__device__ __forceinline__ void dyn_denv4(const DynArgs& a, const Knee& q, const float (&xa)[DE], const float (&xb)[DE],
                                          const float (&ga)[DE], const float (&gb)[DE], const float (&lin)[DE],
                                          float (&dv)[DE], float (&gn)[DE], A* acc) {
#pragma unroll
    for (int i = 0; i < DE; ++i) {
        const float env = fmaxf(lin[i], 0.0f);
        const float G = FAST ? FastMath::log(env + 1e-5f) : logf(env + 1e-5f);
        gn[i] = FAST ? FastMath::exp(log_gain_m<FastMath>(q, G)) : expf(log_gain(q, G));
        const float dgain = a.C == 2 ? (ga[i] * xa[i] + gb[i] * xb[i]) : ga[i] * xa[i];
        const float dg = dgain * gn[i];
        const KneeGrad k = log_gain_grad(q, G);
        dv[i] = lin[i] > 0.0f ? (FAST ? dg * k.dG * __builtin_amdgcn_rcpf(env + 1e-5f) : dg * k.dG / (env + 1e-5f)) : 0.0f;
        if (acc) {   // samples outside the row have x = gy = 0, hence dg = 0
            acc[0] += dg * k.dT;
            acc[1] += dg * k.dlr;
            acc[2] += dg * k.dlk;
        }
    }
}

template <bool TRUNC, bool POLE>
__device__ __forceinline__ void dyn_bwd_c_stream(const DynArgs& a, const OnePole& p, const Knee& q, const float* x0,
                                                 const float* x1, const float* g0, const float* g1, const float* u1,
                                                 float* o0, float* o1, float* slots, int t, double& pole,
                                                 double (&acc)[3]) {
    const int lane = t & 63, wave = t >> 6;
    const bool al = (a.L % 4) == 0;  // reversed float4 groups stay 16-byte aligned only then
    const bool vx = al && vec_ok(x0) && vec_ok(x1) && vec_ok(g0) && vec_ok(g1), vo = al;
    const bool vgx = al && vec_ok(o0) && vec_ok(o1);
    const float k2 = 2.0f / (float)a.C;
    const float pole_c2 = p.a_N - p.one_m_a * (float)a.N * (p.a_N / p.a);
    float carry = 0.0f, carry2 = 0.0f;
    const int64_t ntiles = (a.L + DTILE - 1) / DTILE;
    // software prefetch of the next tile's operands (the scan's barrier would otherwise fence the loads)
    float nu[DE], nxa[DE], nga[DE], nxb[DE] = {0.0f, 0.0f, 0.0f, 0.0f}, ngb[DE] = {0.0f, 0.0f, 0.0f, 0.0f};
    rload4(u1, (int64_t)DE * t, a.L, vo, nu);
    rload4(x0, (int64_t)DE * t, a.L, vx, nxa);
    rload4(g0, (int64_t)DE * t, a.L, vx, nga);
    if (a.C == 2) {
        rload4(x1, (int64_t)DE * t, a.L, vx, nxb);
        rload4(g1, (int64_t)DE * t, a.L, vx, ngb);
    }
    for (int64_t tile = 0; tile < ntiles; ++tile) {
        const int64_t j = tile * DTILE + DE * t;
        float uu[DE], xa[DE], ga[DE], xb[DE], gb[DE];
#pragma unroll
        for (int i = 0; i < DE; ++i) {
            uu[i] = nu[i];
            xa[i] = nxa[i];
            ga[i] = nga[i];
            xb[i] = nxb[i];
            gb[i] = ngb[i];
        }
        if (tile + 1 < ntiles) {
            rload4(u1, j + DTILE, a.L, vo, nu);
            rload4(x0, j + DTILE, a.L, vx, nxa);
            rload4(g0, j + DTILE, a.L, vx, nga);
            if (a.C == 2) {
                rload4(x1, j + DTILE, a.L, vx, nxb);
                rload4(g1, j + DTILE, a.L, vx, ngb);
            }
        }
        float un[DE] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (TRUNC) rload4(u1, j + a.N, a.L, false, un);
        float lin[DE], d[DE], gn[DE], u[DE];
#pragma unroll
        for (int i = 0; i < DE; ++i) lin[i] = TRUNC ? fmaf(-p.a_N, un[i], uu[i]) : uu[i];
        dyn_denv4(a, q, xa, xb, ga, gb, lin, d, gn, acc);
        scan_tile(p, d, u, carry, slots + 8 * (tile & 1), lane, wave);
        if (TRUNC) {
            // denv at the walk position j - N (N samples later in time), recomputed from its own operands; its lagged scan
            // value is u1 at (j - N) + N = j, i.e. uu
            float u_l[DE], xa2[DE], ga2[DE], xb2[DE] = {0.0f, 0.0f, 0.0f, 0.0f}, gb2[DE] = {0.0f, 0.0f, 0.0f, 0.0f};
            rload4(u1, j - a.N, a.L, false, u_l);
            rload4(x0, j - a.N, a.L, false, xa2);
            rload4(g0, j - a.N, a.L, false, ga2);
            if (a.C == 2) {
                rload4(x1, j - a.N, a.L, false, xb2);
                rload4(g1, j - a.N, a.L, false, gb2);
            }
            float lin2[DE], d2[DE], gn2[DE], u2[DE];
#pragma unroll
            for (int i = 0; i < DE; ++i) lin2[i] = fmaf(-p.a_N, uu[i], u_l[i]);
            dyn_denv4(a, q, xa2, xb2, ga2, gb2, lin2, d2, gn2, (float*)nullptr);
            scan_tile(p, d2, u2, carry2, slots + 8 * (tile & 1) + 4, lane, wave);
#pragma unroll
            for (int i = 0; i < DE; ++i) u[i] = fmaf(-p.a_N, u2[i], u[i]);
        }
        if (POLE) {
            const int64_t below = a.L - 1 - j - DE;  // sample under this thread's four
            const float um = (below >= 0 && below < a.L) ? u1[below] : 0.0f;
#pragma unroll
            for (int i = 0; i < DE; ++i) {
                const float prev = i + 1 < DE ? uu[i + 1] : um;
                pole += (double)(p.one_m_a * u[i] * prev - d[i] * uu[i]);
                if (TRUNC) pole += (double)(pole_c2 * d[i] * un[i]);
            }
        }
        float oa[DE];
#pragma unroll
        for (int i = 0; i < DE; ++i) oa[i] = fmaf(gn[i], ga[i], k2 * p.one_m_a * u[i] * xa[i]);
        rstore4(o0, j, a.L, vgx, oa);
        if (a.C == 2) {
            float ob[DE];
#pragma unroll
            for (int i = 0; i < DE; ++i) ob[i] = fmaf(gn[i], gb[i], k2 * p.one_m_a * u[i] * xb[i]);
            rstore4(o1, j, a.L, vgx, ob);
        }
    }
}

__global__ __launch_bounds__(DT) void dyn_bwd_c_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                       gfx_rowmap_t gmap, const float* __restrict__ log_threshold,
                                                       const float* __restrict__ log_ratio,
                                                       const float* __restrict__ log_knee,
                                                       const float* __restrict__ z_alpha, const float* __restrict__ u1,
                                                       float* __restrict__ dalpha, float* __restrict__ gparams,
                                                       float* __restrict__ gx, DynArgs a,
                                                       const float* __restrict__ oneshot_tab) {
    __shared__ float slots[16];
    __shared__ double red[4][4];
    const int t = threadIdx.x;
    const int64_t r = blockIdx.x;
    if (oneshot_tab && oneshot_tab[(size_t)r * DP_TAB + DP_ONESHOT] != 0.0f) return;   // dyn_bwd_oneshot_kernel's row
    OnePole p;
    onepole_setup(p, z_alpha[r], a.N, t & 63);
    const float* x0 = x + drow_off(a.xmap, r, 0);
    const float* x1 = x + drow_off(a.xmap, r, a.C == 2 ? 1 : 0);
    const float* g0 = gy + drow_off(gmap, r, 0);
    const float* g1 = gy + drow_off(gmap, r, a.C == 2 ? 1 : 0);
    float* o0 = gx + drow_off(a.ymap, r, 0);
    float* o1 = gx + drow_off(a.ymap, r, a.C == 2 ? 1 : 0);
    Knee q;
    knee_setup(q, log_threshold[r], log_ratio[r], log_knee ? log_knee[r] : 0.0f, a.knee, a.gate);
    double pole = 0.0, acc[3] = {0.0, 0.0, 0.0};    // per-thread sums over the whole row: double (hundreds of terms of both signs)
    const float* ur = u1 + r * a.L;
    if (dalpha) {
        if (p.trunc)
            dyn_bwd_c_stream<true, true>(a, p, q, x0, x1, g0, g1, ur, o0, o1, slots, t, pole, acc);
        else
            dyn_bwd_c_stream<false, true>(a, p, q, x0, x1, g0, g1, ur, o0, o1, slots, t, pole, acc);
    } else if (p.trunc) {
        dyn_bwd_c_stream<true, false>(a, p, q, x0, x1, g0, g1, ur, o0, o1, slots, t, pole, acc);
    } else {
        dyn_bwd_c_stream<false, false>(a, p, q, x0, x1, g0, g1, ur, o0, o1, slots, t, pole, acc);
    }
    double v4[4] = {acc[0], acc[1], acc[2], pole};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        double v = v4[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if ((t & 63) == 0) red[k][t >> 6] = v;
    }
    __syncthreads();
    if (t < 3) gparams[3 * r + t] = (float)(red[t][0] + red[t][1] + red[t][2] + red[t][3]);
    if (t == 3 && dalpha) dalpha[r] = (float)((red[3][0] + red[3][1] + red[3][2] + red[3][3]) / (double)p.one_m_a);  // u1 = (1-a) U
}

// The backward-in-time pass as dependency-free one-shot tiles (the backward twin of dyn_oneshot_kernel): in the reversed
// "walk" coordinates of dyn_bwd_c_stream the adjoint of the smoother is the same one-pole scan, so a wave takes 512 walk
// positions, rebuilds the scan state entering them from the H positions before (= the H samples LATER in time: lanes
// 4 l < H recompute denv there from their own predicated loads) and needs nothing from any other tile.  Rows are chosen
// on the device from the same pole table; per-row sums (knee parameters, pole) are reduced per workgroup, written to
// `partial` [row][group][4] and added up in group order by dyn_bwd_sums_kernel.  gx means what it means in dyn_bwd_c_kernel.
// Knee kind and compressor / gate are template parameters (one gain-curve path per instantiation: the generic code is
// 15 k instructions, more than the instruction cache holds), every access is a whole aligned float4 (the launcher only
// takes this path for 16-byte aligned rows of a length divisible by four), and log / exp / the reciprocal are the hardware
// forms as in the forward tiles (6.0 vs 6.4 ms with the library functions, 6.9-7.1 for the row kernel, at 8192 rows).
// samples L-4-j .. L-1-j in walk order (v[0] = the latest), zero when the group is outside [0, L)
__device__ __forceinline__ void rl4(const float* __restrict__ row, int64_t j, int64_t L, float (&v)[DE]) {
    const int64_t n = L - 4 - j;
    float4 q = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (n >= 0 && n + 4 <= L) q = *reinterpret_cast<const float4*>(row + n);
    v[0] = q.w; v[1] = q.z; v[2] = q.y; v[3] = q.x;
}

// RESCAN (round 6): the smoother's scan u1 is not read but REBUILT from x -- in walk coordinates the forward-in-time scan is
// a SUFFIX scan (u1[j] depends on the positions after j = the samples before it in time): the two sub-tiles' local and
// in-wave scans run with the shuffles mirrored, the state entering the tile from its far end is the dot product of the H
// samples beyond it (x only: one more predicated 16-byte load per channel), and the H positions in front of the tile (whose
// denv the adjoint scan needs) continue the scan from the tile's first value.  4 of the 28 bytes per stereo sample go away
// here, and the forward pass of a training step does not have to store the scan at all (4 of its 20).
// (three waves per SIMD = 168 VGPRs: the rescan's 169-172 would otherwise cost a whole wave of occupancy)
template <int KIND, bool GATE, bool RESCAN>
__global__ __launch_bounds__(DT, 3) void dyn_bwd_oneshot_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                             gfx_rowmap_t gmap, const float* __restrict__ log_threshold,
                                                             const float* __restrict__ log_ratio,
                                                             const float* __restrict__ log_knee,
                                                             const float* __restrict__ tab, const float* __restrict__ u1,
                                                             const float* __restrict__ dalpha /* only: wanted? */,
                                                             double* __restrict__ partial,
                                                             float* __restrict__ gx, DynArgs a, unsigned ngroups,
                                                             unsigned nblocks) {
    __shared__ double red[4][4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const unsigned per_xcd = gridDim.x >> 3;
    const unsigned b = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (b >= nblocks) return;
    const unsigned r = b / ngroups;
    const unsigned grp = b - r * ngroups;
    const float* tb = tab + (size_t)r * DP_TAB;
    if (tb[DP_ONESHOT] == 0.0f) return;                 // dyn_bwd_c_kernel's row (uniform)
    const int64_t s = (int64_t)grp * OS_GTILE + (int64_t)wave * OS_WTILE;     // first WALK position of this wave's tile
    const int64_t L = s < a.L ? a.L : 0;                // a wave past the row end reads zeros and stores nothing
    const float* x0 = x + drow_off(a.xmap, r, 0);
    const float* x1 = x + drow_off(a.xmap, r, a.C == 2 ? 1 : 0);
    const float* g0 = gy + drow_off(gmap, r, 0);
    const float* g1 = gy + drow_off(gmap, r, a.C == 2 ? 1 : 0);
    float* o0 = gx + drow_off(a.ymap, r, 0);
    float* o1 = gx + drow_off(a.ymap, r, a.C == 2 ? 1 : 0);
    const float* ur = RESCAN ? nullptr : u1 + (int64_t)r * a.L;
    const bool stereo = a.C == 2;
    const int64_t j0 = s + DE * lane;

    float uu[OS_SUB][DE], xa[OS_SUB][DE], xb[OS_SUB][DE], ga[OS_SUB][DE], gb[OS_SUB][DE];
#pragma unroll
    for (int k = 0; k < OS_SUB; ++k) {
        if constexpr (!RESCAN) rl4(ur, j0 + 256 * k, L, uu[k]);
        rl4(x0, j0 + 256 * k, L, xa[k]);
        rl4(g0, j0 + 256 * k, L, ga[k]);
        rl4(x1, j0 + 256 * k, stereo ? L : 0, xb[k]);
        rl4(g1, j0 + 256 * k, stereo ? L : 0, gb[k]);
    }
    // walk positions s - 4 (l + 1) .. s - 4 l - 1 = taps 4 l + 3 .. 4 l of the state entering the tile
    const int H = (int)tb[DP_HIST];
    const bool hist = s != 0 && DE * lane < H;
    const int64_t Lh = hist ? L : 0, jh = s - DE * (lane + 1);
    float hu[DE], hxa[DE], hxb[DE], hga[DE], hgb[DE];
    if constexpr (!RESCAN) rl4(ur, jh, Lh, hu);
    rl4(x0, jh, Lh, hxa);
    rl4(g0, jh, Lh, hga);
    rl4(x1, jh, stereo ? Lh : 0, hxb);
    rl4(g1, jh, stereo ? Lh : 0, hgb);
    // u1 one walk position past the tile (the pole term pairs every position with the next one)
    const int64_t edge = a.L - 1 - (s + OS_WTILE);
    float u_edge = 0.0f;
    if constexpr (!RESCAN) u_edge = (dalpha && L != 0 && edge >= 0) ? ur[edge] : 0.0f;
    const float a1 = tb[77], one_m_a = tb[78], a_sub = tb[70];
    const float apk[DE] = {tb[73], tb[74], tb[75], tb[76]};
    const float a_lane = tb[lane];
    float a_step[6];
#pragma unroll
    for (int d = 0; d < 6; ++d) a_step[d] = tb[64 + d];
    if constexpr (RESCAN) {
        const float invC = 1.0f / (float)a.C;
        // the H samples beyond the far end of the tile (walk positions s + 512 + 4 lane + i: EARLIER in time; zeros past
        // the row start), taps a^(4 lane + i)
        float fxa[DE], fxb[DE];
        const int64_t Lf = DE * lane < H ? L : 0;
        rl4(x0, s + OS_WTILE + DE * lane, Lf, fxa);
        rl4(x1, s + OS_WTILE + DE * lane, stereo ? Lf : 0, fxb);
        // local and in-wave SUFFIX scans of the two sub-tiles (independent of each other)
        float fl[OS_SUB][DE], fexcl[OS_SUB], ftot[OS_SUB];
#pragma unroll
        for (int k = 0; k < OS_SUB; ++k) {
            float acc = 0.0f;
#pragma unroll
            for (int i = DE - 1; i >= 0; --i) {
                const float e = (stereo ? (xa[k][i] * xa[k][i] + xb[k][i] * xb[k][i]) : xa[k][i] * xa[k][i]) * invC;
                acc = fmaf(a1, acc, e);
                fl[k][i] = acc;
            }
            float inc = acc;
#pragma unroll
            for (int st = 0; st < 6; ++st) {
                const float dn = __shfl_down(inc, 1 << st, 64);
                if (lane + (1 << st) < 64) inc = fmaf(a_step[st], dn, inc);
            }
            const float ex = __shfl_down(inc, 1, 64);
            fexcl[k] = lane == 63 ? 0.0f : ex;
            ftot[k] = __shfl(inc, 0, 64);
        }
        float w = 0.0f;
#pragma unroll
        for (int i = DE - 1; i >= 0; --i) {
            const float e = (stereo ? (fxa[i] * fxa[i] + fxb[i] * fxb[i]) : fxa[i] * fxa[i]) * invC;
            w = fmaf(a1, w, e);
        }
        float fc = w * a_lane;                    // (lanes without a live tap loaded zeros)
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) fc += __shfl_xor(fc, o, 64);
        u_edge = one_m_a * fc;                    // the scan one walk position past the tile (0 past the row start)
        const float a_far = tb[63 - lane];        // a^(4 (63 - lane)): from the sub-tile's far end to this lane's
#pragma unroll
        for (int k = OS_SUB - 1; k >= 0; --k) {
            const float pre = fmaf(a_far, fc, fexcl[k]);
            fc = fmaf(a_sub, fc, ftot[k]);
#pragma unroll
            for (int i = 0; i < DE; ++i) uu[k][i] = one_m_a * fmaf(apk[DE - 1 - i], pre, fl[k][i]);
        }
        // the scan continued over the H positions in front of the tile (lane l: s - 4 (l + 1) + i), from its value at s
        float hl[DE], acc = 0.0f;
#pragma unroll
        for (int i = DE - 1; i >= 0; --i) {
            const float e = (stereo ? (hxa[i] * hxa[i] + hxb[i] * hxb[i]) : hxa[i] * hxa[i]) * invC;
            acc = fmaf(a1, acc, e);
            hl[i] = acc;
        }
        float inc = acc;
#pragma unroll
        for (int st = 0; st < 6; ++st) {
            const float up = __shfl_up(inc, 1 << st, 64);
            if (lane >= (1 << st)) inc = fmaf(a_step[st], up, inc);
        }
        const float ex = __shfl_up(inc, 1, 64);
        const float pre = fmaf(a_lane, fc, lane == 0 ? 0.0f : ex);
#pragma unroll
        for (int i = 0; i < DE; ++i) hu[i] = one_m_a * fmaf(apk[DE - 1 - i], pre, hl[i]);
    }
    Knee q;
    knee_setup(q, log_threshold[r], log_ratio[r], log_knee ? log_knee[r] : 0.0f, KIND, GATE ? 1 : 0);
    q.kind = KIND;
    q.gate = GATE ? 1 : 0;
    const float k2 = 2.0f / (float)a.C;

    float acc[3] = {0.0f, 0.0f, 0.0f}, pole = 0.0f;
    float d[OS_SUB][DE], gn[OS_SUB][DE], loc[OS_SUB][DE], excl[OS_SUB], total[OS_SUB];
#pragma unroll
    for (int k = 0; k < OS_SUB; ++k) {
        dyn_denv4<GFX_DYN_BWD_FAST>(a, q, xa[k], xb[k], ga[k], gb[k], uu[k], d[k], gn[k], acc);
        float run = 0.0f;
#pragma unroll
        for (int i = 0; i < DE; ++i) {
            run = fmaf(a1, run, d[k][i]);
            loc[k][i] = run;
        }
        float inc = run;
#pragma unroll
        for (int st = 0; st < 6; ++st) {
            const float up = __shfl_up(inc, 1 << st, 64);
            if (lane >= (1 << st)) inc = fmaf(a_step[st], up, inc);
        }
        const float ex = __shfl_up(inc, 1, 64);
        excl[k] = lane == 0 ? 0.0f : ex;
        total[k] = __shfl(inc, 63, 64);
    }
    float carry = 0.0f;
    if (s != 0 && H > 0) {                       // uniform
        float hd[DE], hgn[DE];
        dyn_denv4<GFX_DYN_BWD_FAST>(a, q, hxa, hxb, hga, hgb, hu, hd, hgn, (float*)nullptr);   // (lanes without a live tap hold zeros: denv = 0)
        float w = 0.0f;                          // Horner, farthest walk position first
#pragma unroll
        for (int i = 0; i < DE; ++i) w = fmaf(a1, w, hd[i]);
        float hs = hist ? w * a_lane : 0.0f;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) hs += __shfl_xor(hs, o, 64);
        carry = hs;
    }
#pragma unroll
    for (int k = 0; k < OS_SUB; ++k) {
        const float pre = fmaf(a_lane, carry, excl[k]);
        carry = fmaf(a_sub, carry, total[k]);
        float u[DE];
#pragma unroll
        for (int i = 0; i < DE; ++i) u[i] = fmaf(apk[i], pre, loc[k][i]);        // the adjoint scan ("de")
        if (dalpha) {
            // u1 at the next walk position: the neighbouring lane's first value, the next sub-tile's, or the one past the tile
            float nxt = __shfl_down(uu[k][0], 1, 64);
            const float first_next = k + 1 < OS_SUB ? __shfl(uu[k + 1 < OS_SUB ? k + 1 : k][0], 0, 64) : u_edge;
            if (lane == 63) nxt = first_next;
#pragma unroll
            for (int i = 0; i < DE; ++i) {
                const float prev = i + 1 < DE ? uu[k][i + 1] : nxt;
                pole += one_m_a * u[i] * prev - d[k][i] * uu[k][i];
            }
        }
        const int64_t n = L - 4 - (j0 + 256 * k);
        if (n >= 0 && n + 4 <= L) {
            using f4 = float __attribute__((ext_vector_type(4)));
            f4 oa, ob;
#pragma unroll
            for (int i = 0; i < DE; ++i) {
                oa[3 - i] = fmaf(gn[k][i], ga[k][i], k2 * one_m_a * u[i] * xa[k][i]);
                ob[3 - i] = fmaf(gn[k][i], gb[k][i], k2 * one_m_a * u[i] * xb[k][i]);
            }
            *reinterpret_cast<f4*>(o0 + n) = oa;
            if (stereo) *reinterpret_cast<f4*>(o1 + n) = ob;
        }
    }
    double v4[4] = {acc[0], acc[1], acc[2], pole};      // eight terms per thread in float, everything above them in double
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        double v = v4[k];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) red[k][wave] = v;
    }
    __syncthreads();
    // this workgroup's share of the row's four sums; dyn_bwd_sums_kernel adds the shares in group order (no atomics: the
    // parameter gradients are the same bits from run to run)
    if (t < 4) partial[((size_t)r * ngroups + grp) * 4 + t] = red[t][0] + red[t][1] + red[t][2] + red[t][3];
}

// gparams[r] (3 sums) and dalpha[r] of the rows dyn_bwd_oneshot_kernel took: its workgroups' partials in group order, one
// wave per row (lane l adds groups l, l + 64, ... in order, then a shuffle tree).
__global__ __launch_bounds__(64) void dyn_bwd_sums_kernel(const double* __restrict__ partial, const float* __restrict__ tab,
                                                          float* __restrict__ gparams, float* __restrict__ dalpha,
                                                          unsigned ngroups) {
    const unsigned r = blockIdx.x;
    const float* tb = tab + (size_t)r * DP_TAB;
    if (tb[DP_ONESHOT] == 0.0f) return;                 // dyn_bwd_c_kernel wrote this row's sums itself
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    for (unsigned g = threadIdx.x; g < ngroups; g += 64) {
        const double* p = partial + ((size_t)r * ngroups + g) * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] += p[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_down(v[k], o, 64);
    if (threadIdx.x == 0) {
        gparams[3 * (size_t)r + 0] = (float)v[0];
        gparams[3 * (size_t)r + 1] = (float)v[1];
        gparams[3 * (size_t)r + 2] = (float)v[2];
        if (dalpha) dalpha[r] = (float)(v[3] / (double)tb[78]);   // u1 = (1 - a) U
    }
}

// One pass over (x, gy, env): gain = exp(g(log(env + 1e-5))),  dgain = sum_c gy x,  dg = dgain * gain,
//   denv = dg * dg/dG / (env + 1e-5),   gparams[r] += sum_n dg * (dg/dT, dg/dlog_ratio, dg/dlog_knee).
__global__ __launch_bounds__(256) void dyn_gain_bwd_kernel(const float* __restrict__ x, gfx_rowmap_t xmap,
                                                           const float* __restrict__ gy, gfx_rowmap_t gmap,
                                                           const float* __restrict__ env,
                                                           const float* __restrict__ log_threshold,
                                                           const float* __restrict__ log_ratio,
                                                           const float* __restrict__ log_knee, int64_t R, int64_t L,
                                                           int C, int knee, int gate, float* __restrict__ gain,
                                                           float* __restrict__ denv, float* __restrict__ gparams) {
    __shared__ float red[3][4];
    for (int64_t r = blockIdx.y; r < R; r += gridDim.y) {
        Knee q;
        knee_setup(q, log_threshold[r], log_ratio[r], log_knee ? log_knee[r] : 0.0f, knee, gate);
        const float* x0 = x + drow_off(xmap, r, 0);
        const float* x1 = x + drow_off(xmap, r, C == 2 ? 1 : 0);
        const float* g0 = gy + drow_off(gmap, r, 0);
        const float* g1 = gy + drow_off(gmap, r, C == 2 ? 1 : 0);
        float sT = 0.0f, sR = 0.0f, sK = 0.0f;
        for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < L; n += (int64_t)gridDim.x * blockDim.x) {
            const float e = env[r * L + n];
            const float G = logf(e + 1e-5f);
            const float gn = expf(log_gain(q, G));
            const float dgain = C == 2 ? (g0[n] * x0[n] + g1[n] * x1[n]) : g0[n] * x0[n];
            const float dg = dgain * gn;
            const KneeGrad k = log_gain_grad(q, G);
            gain[r * L + n] = gn;
            denv[r * L + n] = dg * k.dG / (e + 1e-5f);
            sT += dg * k.dT;
            sR += dg * k.dlr;
            sK += dg * k.dlk;
        }
        for (int o = 32; o > 0; o >>= 1) {
            sT += __shfl_down(sT, o, 64);
            sR += __shfl_down(sR, o, 64);
            sK += __shfl_down(sK, o, 64);
        }
        if ((threadIdx.x & 63) == 0) {
            red[0][threadIdx.x >> 6] = sT;
            red[1][threadIdx.x >> 6] = sR;
            red[2][threadIdx.x >> 6] = sK;
        }
        __syncthreads();
        if (threadIdx.x < 3)
            atomicAdd(&gparams[3 * r + threadIdx.x],
                      red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
        __syncthreads();
    }
}

// da[r] = sum_n g[n] (c0 U[n] + c2 U[n-N]) + g[n+1] (c1 S[n] + c3 S[n-N]),  U/S zero before the row start, g[L] = 0:
// the pole gradient of the truncated one-pole smoother from its two scans (see autograd.pole_gradient; the
// one-sample shift pairs g[n+1] with S[n] = D[n+1]).
__global__ __launch_bounds__(256) void onepole_dz_kernel(const float* __restrict__ g, const float* __restrict__ U,
                                                         const float* __restrict__ S, const float* __restrict__ coef,
                                                         float* __restrict__ da, int64_t L, int64_t N) {
    __shared__ float part[4];
    const int64_t r = blockIdx.x;
    const float c0 = coef[4 * r], c1 = coef[4 * r + 1], c2 = coef[4 * r + 2], c3 = coef[4 * r + 3];
    const float* gr = g + r * L;
    const float* Ur = U + r * L;
    const float* Sr = S + r * L;
    float s = 0.0f;
    for (int64_t n = threadIdx.x; n < L; n += 256) {
        float u = c0 * Ur[n], d = c1 * Sr[n];
        if (n >= N) {
            u += c2 * Ur[n - N];
            d += c3 * Sr[n - N];
        }
        s = fmaf(gr[n], u, s);
        if (n + 1 < L) s = fmaf(gr[n + 1], d, s);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) da[r] = part[0] + part[1] + part[2] + part[3];
}

// gx[r,c,n] = gain[r,n] * gy[r,c,n] + (2/C) * de[r,n] * x[r,c,n]   (de = dL/d energy, energy = mean_c x^2)
__global__ void dyn_dx_kernel(const float* __restrict__ x, gfx_rowmap_t xmap, const float* __restrict__ gy,
                              gfx_rowmap_t gmap, const float* __restrict__ gain, const float* __restrict__ de,
                              float* __restrict__ gx, int64_t R, int64_t L, int C) {
    const float k = 2.0f / (float)C;
    for (int64_t r = blockIdx.y; r < R; r += gridDim.y)
    for (int c = 0; c < C; ++c) {
        const float* xr = x + drow_off(xmap, r, c);
        const float* gr = gy + drow_off(gmap, r, c);
        float* o = gx + (r * C + c) * L;
        for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < L; n += (int64_t)gridDim.x * blockDim.x)
            o[n] = fmaf(gain[r * L + n], gr[n], k * de[r * L + n] * xr[n]);
    }
}

// StereoGain: y[r,c,n] = x[r,cx,n] * exp(log_gain[r,c])   (stereo.py:38-41; mono input broadcasts to 2 channels)
__global__ void stereo_gain_kernel(const float* __restrict__ x, gfx_rowmap_t xmap, const float* __restrict__ log_gain,
                                   float* __restrict__ y, gfx_rowmap_t ymap, int64_t R, int64_t L, int Cin) {
    for (int64_t r = blockIdx.y; r < R; r += gridDim.y)
    for (int c = 0; c < 2; ++c) {
        const float g = expf(log_gain[2 * r + c]);
        const float* xr = x + drow_off(xmap, r, Cin == 2 ? c : 0);
        float* yr = y + drow_off(ymap, r, c);
        for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < L; n += (int64_t)gridDim.x * blockDim.x)
            yr[n] = xr[n] * g;
    }
}

static inline dim3 row_grid(int64_t R, int64_t L) {
    int64_t bx = (L + 255) / 256;
    if (bx > 64) bx = 64;
    return dim3((unsigned)bx, (unsigned)(R > 65535 ? 65535 : R));
}

}  // namespace gfx

using namespace gfx;

#define GFX_LAUNCH_OK() (hipGetLastError() == hipSuccess ? GFX_OK : GFX_ELAUNCH)

extern "C" {

int gfx_dynamics_fused_f32(const float* x, gfx_rowmap_t xmap, float* y, gfx_rowmap_t ymap, const float* log_threshold,
                           const float* log_ratio, const float* log_knee, const float* z_alpha, int64_t R, int64_t C,
                           int64_t L, int smoother, int64_t iir_len, int knee, int gate, void* stream) {
    return gfx_dynamics_fused_ex_f32(x, xmap, y, ymap, log_threshold, log_ratio, log_knee, z_alpha, R, R, C, L, smoother,
                                     iir_len, knee, gate, stream);
}

int gfx_dynamics_fused_ex_f32(const float* x, gfx_rowmap_t xmap, float* y, gfx_rowmap_t ymap, const float* log_threshold,
                              const float* log_ratio, const float* log_knee, const float* z_alpha, int64_t param_rows,
                              int64_t R, int64_t C, int64_t L, int smoother, int64_t iir_len, int knee, int gate,
                              void* stream) {
    return gfx_dynamics_fused_u1_f32(x, xmap, y, ymap, log_threshold, log_ratio, log_knee, z_alpha, param_rows, R, C, L,
                                     smoother, iir_len, knee, gate, nullptr, stream);
}

int gfx_dynamics_fused_u1_f32(const float* x, gfx_rowmap_t xmap, float* y, gfx_rowmap_t ymap, const float* log_threshold,
                              const float* log_ratio, const float* log_knee, const float* z_alpha, int64_t param_rows,
                              int64_t R, int64_t C, int64_t L, int smoother, int64_t iir_len, int knee, int gate,
                              float* u1, void* stream) {
    return gfx_dynamics_fused_ws_f32(x, xmap, y, ymap, log_threshold, log_ratio, log_knee, z_alpha, param_rows, R, C, L,
                                     smoother, iir_len, knee, gate, u1, nullptr, 0, stream);
}

size_t gfx_dynamics_ws_bytes(int64_t param_rows) {
    return param_rows <= 0 ? 0 : (size_t)param_rows * DP_TAB * sizeof(float);
}

// pole table | 16 control words (tickets, "some row looks back") | one 8-byte granule per row and 512-sample tile
static size_t dyn_lb_offset(int64_t param_rows) { return ((size_t)param_rows * DP_TAB * sizeof(float) + 255) & ~(size_t)255; }
size_t gfx_dynamics_ws_bytes_ex(int64_t param_rows, int64_t R, int64_t L) {
    if (param_rows <= 0 || R <= 0 || L <= 0) return 0;
    return dyn_lb_offset(param_rows) + 64 + (size_t)R * (size_t)((L + OS_WTILE - 1) / OS_WTILE) * 8;
}

size_t gfx_dynamics_bwd_ws_bytes(int64_t R, int64_t L) {   // the pole table + four partial sums per one-shot workgroup
    if (R <= 0 || L <= 0) return 0;
    // (the table padded to 8 bytes: the partial sums behind it are doubles)
    return (((size_t)R * DP_TAB + 1) & ~(size_t)1) * sizeof(float) + (size_t)R * (size_t)((L + OS_GTILE - 1) / OS_GTILE) * 4 * sizeof(double);
}

static thread_local const char* t_dyn_last_kernel = "";   // see gfx_dynamics_last_kernel

static int dynamics_fused_launch(const float* x, gfx_rowmap_t xmap, float* y, gfx_rowmap_t ymap, const float* log_threshold,
                                 const float* log_ratio, const float* log_knee, const float* z_alpha, int64_t param_rows,
                                 int64_t R, int64_t C, int64_t L, int smoother, int64_t iir_len, int knee, int gate,
                                 float* u1, void* ws, size_t ws_bytes, void* stream, const MixArgs* mix, int mix_na = 0) {
    if (param_rows < 1 || param_rows > R) return GFX_EINVAL;
    if (u1 && smoother != 1) return GFX_EINVAL;
    if (!x || !y || !log_threshold || !log_ratio || R <= 0 || L <= 0 || (C != 1 && C != 2)) return GFX_EINVAL;
    if (knee < 0 || knee > 2 || (knee != 0 && !log_knee)) return GFX_EINVAL;
    if (smoother != 0 && smoother != 1) return GFX_EINVAL;
    if (smoother == 1 && (!z_alpha || iir_len < 1)) return GFX_EINVAL;
    if (R > 0x7fffffffLL) return GFX_EINVAL;
    if (ws && ws_bytes < gfx_dynamics_ws_bytes(param_rows)) return GFX_ENOSPC;
    DynArgs a;
    a.xmap = xmap; a.ymap = ymap; a.R = R; a.L = L; a.N = iir_len; a.C = (int)C;
    a.smoother = smoother; a.knee = knee; a.gate = gate;
    a.prows = (unsigned)param_rows;
    hipStream_t st = (hipStream_t)stream;
    const int64_t ntiles = (L + DTILE - 1) / DTILE;
    // With a workspace and a smoother: the pole table, then the dependency-free one-shot grid for the rows whose
    // history fits (decided per row on the device), then the row kernel for the others (same table, complementary test).
    const float* tab = nullptr;
    const int64_t ngroups = (L + OS_GTILE - 1) / OS_GTILE;
    // (tile grids: 8 runs of whole rows / graphs, one per XCD -- see lb_block_index)
    const bool oneshot = ws && smoother == 1 && L > OS_WTILE && (R + 8) * ngroups <= 0x7ffffff0LL;
    if (mix && !oneshot) return GFX_EINVAL;
    LbArgs lb = {nullptr, nullptr, 0, 0};
    auto tile_grid = [&](int64_t units) { return dim3((unsigned)(((units + 7) / 8) * ngroups * 8)); };
    if (oneshot) {
        float* t = (float*)ws;
        // the look-back walks move whole aligned float4 (as the fused routing sum does: its entry point checked already)
        auto al16 = [](const void* p, const gfx_rowmap_t& mp) {
            return ((uintptr_t)p & 15) == 0 && ((mp.stride_outer | mp.stride_inner | mp.stride_ch) & 3) == 0;
        };
        const bool lb_ok = mix || (al16(x, xmap) && al16(y, ymap) && ((uintptr_t)u1 & 15) == 0 && (L & 3) == 0);
        if (lb_ok && ws_bytes >= gfx_dynamics_ws_bytes_ex(param_rows, R, L)) {   // room for the look-back granules: rows with
            char* base = (char*)ws + dyn_lb_offset(param_rows);                  // a long smoother memory stay on the tile grid
            lb.ctrl = (unsigned*)base;
            lb.gran = (unsigned long long*)(base + 64);
            lb.ntiles = (unsigned)((L + OS_WTILE - 1) / OS_WTILE);
            if (hipMemsetAsync(base, 0, 64 + (size_t)R * lb.ntiles * 8, st) != hipSuccess) return GFX_ELAUNCH;
        }
        hipLaunchKernelGGL(dyn_pole_table_kernel, dim3((unsigned)param_rows), dim3(64), 0, st, z_alpha, t, param_rows, iir_len,
                           lb.gran ? lb.ctrl + 8 : (unsigned*)nullptr);
        tab = t;
        if (!mix) {
            const unsigned nblocks = (unsigned)(R * ngroups);
            a.nchunks = 1;
            a.chunk_tiles = 1;
            const LbArgs none = {nullptr, nullptr, 0, 0};   // (look-back rows are produced by the row-group walk below)
            hipLaunchKernelGGL(dyn_oneshot_kernel, tile_grid(R), dim3(DT), 0, st, x, y, log_threshold,
                               log_ratio, log_knee, (const float*)t, a, (unsigned)ngroups, nblocks, u1, none);
        }
    }
    // Few rows: one workgroup per row walks the whole length serially (~2 us per tile) and the launch is bound by
    // that latency, not by bandwidth.  Split every row into time chunks then; a chunk re-scans N samples of history
    // (exact, the smoother is an N-tap FIR), so chunks are kept at least as long as that history.
    const int64_t warm = smoother == 1 ? (iir_len + DTILE - 1) / DTILE : 0;
    int64_t nchunks = 1;
    while (!u1 && R * nchunks < 2048 && nchunks < 16 && ntiles / (2 * nchunks) >= (warm > 4 ? warm : 4)) nchunks *= 2;
    a.nchunks = (int)nchunks;
    a.chunk_tiles = (ntiles + nchunks - 1) / nchunks;
    if (R * nchunks > 0x7fffffffLL) return GFX_EINVAL;
    hipLaunchKernelGGL(dyn_fused_kernel, dim3((unsigned)(R * nchunks)), dim3(DT), 0, st, x, y,
                       log_threshold, log_ratio, log_knee, z_alpha, a, u1, tab);
    if (!mix && lb.gran) {
        // look-back rows of a call without a routing sum: waves walk groups of 16 rows tile by tile, one row deferred (the
        // schedule of the fused sum without accumulators); leaves at once when the pole table found no such row
        constexpr int LB_GROUP = 16;
        const int64_t units = (R + LB_GROUP - 1) / LB_GROUP;
        const unsigned nblocks = (unsigned)(units * ngroups);
        MixArgs none;
        none.sched = nullptr; none.out = nullptr; none.sb = none.sv = none.sc = 0; none.inner = LB_GROUP;
        none.extras = nullptr; none.n_pre = none.n_post = 0; none.skip_rows = 0;
        a.nchunks = 1;
        a.chunk_tiles = 1;
        lb.split = 1;
        const dim3 grid = tile_grid(units), blk(DT);
#define GFX_LBW3(ST, KN, GT)                                                                                               \
    hipLaunchKernelGGL((dyn_oneshot_mix_kernel<0, ST, KN, GT, true>), grid, blk, 0, st, x, y, log_threshold, log_ratio,   \
                       log_knee, tab, a, (unsigned)ngroups, nblocks, u1, none, lb)
#define GFX_LBW2(ST, KN)                    \
    do {                                    \
        if (gate) GFX_LBW3(ST, KN, true);   \
        else GFX_LBW3(ST, KN, false);       \
    } while (0)
#define GFX_LBW(ST)                            \
    do {                                       \
        if (knee == 0) GFX_LBW2(ST, 0);        \
        else if (knee == 1) GFX_LBW2(ST, 1);   \
        else GFX_LBW2(ST, 2);                  \
    } while (0)
        if (C == 2) GFX_LBW(true);
        else GFX_LBW(false);
#undef GFX_LBW
#undef GFX_LBW2
#undef GFX_LBW3
    }
    if (mix) {   // after the row kernel: its rows are read back by the tiles that sum them
        const unsigned nblocks = (unsigned)((R / mix->inner) * ngroups);
        a.nchunks = 1;
        a.chunk_tiles = 1;
        const dim3 grid = tile_grid(R / mix->inner), blk(DT);
        // GRAFX_DYN_DEFER (experiments): 0 = the plain walk only, 1 = the deferred walk only; default: both, chosen on the device
        static const int defer_mode = [] {
            const char* e = getenv("GRAFX_DYN_DEFER");
            return e && *e ? atoi(e) : -1;
        }();
        lb.split = (defer_mode < 0 && lb.gran) ? 1 : 0;
#define GFX_MIX3(NA_, ST, KN, GT)                                                                                            \
    do {                                                                                                                     \
        if (defer_mode != 1)                                                                                                 \
            hipLaunchKernelGGL((dyn_oneshot_mix_kernel<NA_, ST, KN, GT, false>), grid, blk, 0, st, x, y, log_threshold,     \
                               log_ratio, log_knee, tab, a, (unsigned)ngroups, nblocks, u1, *mix, lb);                       \
        if (defer_mode == 1 || lb.split)                                                                                     \
            hipLaunchKernelGGL((dyn_oneshot_mix_kernel<NA_, ST, KN, GT, true>), grid, blk, 0, st, x, y, log_threshold,      \
                               log_ratio, log_knee, tab, a, (unsigned)ngroups, nblocks, u1, *mix, lb);                       \
    } while (0)
#define GFX_MIX2(NA_, ST, KN)           \
    do {                                \
        if (gate) GFX_MIX3(NA_, ST, KN, true); \
        else GFX_MIX3(NA_, ST, KN, false);     \
    } while (0)
#define GFX_MIX(NA_, ST)                       \
    do {                                       \
        if (knee == 0) GFX_MIX2(NA_, ST, 0);   \
        else if (knee == 1) GFX_MIX2(NA_, ST, 1); \
        else GFX_MIX2(NA_, ST, 2);             \
    } while (0)
        if (C == 2) {
            if (mix_na <= 2) GFX_MIX(2, true);
            else GFX_MIX(4, true);
        } else {
            if (mix_na <= 2) GFX_MIX(2, false);
            else GFX_MIX(4, false);
        }
#undef GFX_MIX2
#undef GFX_MIX3
#undef GFX_MIX
    }
    const int rc = GFX_LAUNCH_OK();
    // (a call without a routing sum whose workspace holds the look-back granules also launches the row-group walk,
    // dyn_oneshot_mix_kernel<0, ...>: which of the two produced the rows is decided on the device, so both are named)
    if (rc == GFX_OK)
        t_dyn_last_kernel = mix ? "dyn_oneshot_mix_kernel"
                                : (oneshot ? (lb.gran ? "dyn_oneshot_kernel+dyn_oneshot_mix_kernel" : "dyn_oneshot_kernel")
                                           : "dyn_fused_kernel");
    return rc;
}

const char* gfx_dynamics_last_kernel(void) { return t_dyn_last_kernel; }

int gfx_dynamics_fused_ws_f32(const float* x, gfx_rowmap_t xmap, float* y, gfx_rowmap_t ymap, const float* log_threshold,
                              const float* log_ratio, const float* log_knee, const float* z_alpha, int64_t param_rows,
                              int64_t R, int64_t C, int64_t L, int smoother, int64_t iir_len, int knee, int gate,
                              float* u1, void* ws, size_t ws_bytes, void* stream) {
    return dynamics_fused_launch(x, xmap, y, ymap, log_threshold, log_ratio, log_knee, z_alpha, param_rows, R, C, L, smoother,
                                 iir_len, knee, gate, u1, ws, ws_bytes, stream, nullptr);
}

int gfx_dynamics_fused_mix_f32(const float* x, gfx_rowmap_t xmap, float* y, gfx_rowmap_t ymap, const float* log_threshold,
                               const float* log_ratio, const float* log_knee, const float* z_alpha, int64_t param_rows,
                               int64_t R, int64_t C, int64_t L, int smoother, int64_t iir_len, int knee, int gate,
                               float* u1, void* ws, size_t ws_bytes, const int64_t* sched, int64_t inner, int64_t n_acc,
                               float* mix, int64_t mix_sb, int64_t mix_sv, int64_t mix_sc, const int64_t* extras,
                               int64_t n_pre, int64_t n_post, void* stream) {
    return gfx_dynamics_fused_mix_flags_f32(x, xmap, y, ymap, log_threshold, log_ratio, log_knee, z_alpha, param_rows, R, C, L,
                                            smoother, iir_len, knee, gate, u1, ws, ws_bytes, sched, inner, n_acc, mix, mix_sb,
                                            mix_sv, mix_sc, extras, n_pre, n_post, 0, stream);
}

int gfx_dynamics_fused_mix_flags_f32(const float* x, gfx_rowmap_t xmap, float* y, gfx_rowmap_t ymap, const float* log_threshold,
                                     const float* log_ratio, const float* log_knee, const float* z_alpha, int64_t param_rows,
                                     int64_t R, int64_t C, int64_t L, int smoother, int64_t iir_len, int knee, int gate,
                                     float* u1, void* ws, size_t ws_bytes, const int64_t* sched, int64_t inner, int64_t n_acc,
                                     float* mix, int64_t mix_sb, int64_t mix_sv, int64_t mix_sc, const int64_t* extras,
                                     int64_t n_pre, int64_t n_post, int flags, void* stream) {
    if (!sched || !mix || inner < 1 || inner > 65535 || n_acc < 1 || n_acc > 4 || R % inner != 0 || !ws || smoother != 1)
        return GFX_EINVAL;
    if (n_pre < 0 || n_post < 0 || n_pre + n_post > 65535 || (n_pre + n_post > 0 && !extras)) return GFX_EINVAL;
    // every access of the fused kernel is a whole aligned float4 (the element-wise paths would triple its code size)
    const int64_t strides = xmap.stride_outer | xmap.stride_inner | xmap.stride_ch | ymap.stride_outer | ymap.stride_inner |
                            ymap.stride_ch | mix_sb | mix_sv | mix_sc;
    if ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)mix | (uintptr_t)u1) & 15) != 0 || (strides & 3) != 0 || (L & 3) != 0)
        return GFX_EINVAL;
    MixArgs m;
    m.sched = sched; m.out = mix; m.sb = mix_sb; m.sv = mix_sv; m.sc = mix_sc; m.inner = (int)inner;
    m.extras = extras; m.n_pre = (int)n_pre; m.n_post = (int)n_post;
    m.skip_rows = (flags & GFX_MIX_SKIP_ROWS) ? 1 : 0;
    return dynamics_fused_launch(x, xmap, y, ymap, log_threshold, log_ratio, log_knee, z_alpha, param_rows, R, C, L, smoother,
                                 iir_len, knee, gate, u1, ws, ws_bytes, stream, &m, (int)n_acc);
}

int gfx_energy_f32(const float* x, gfx_rowmap_t xmap, float* e, int64_t R, int64_t C, int64_t L, void* stream) {
    if (!x || !e || R <= 0 || L <= 0 || (C != 1 && C != 2)) return GFX_EINVAL;
    hipLaunchKernelGGL(energy_kernel, row_grid(R, L), dim3(256), 0, (hipStream_t)stream, x, xmap, e, R, L, (int)C);
    return GFX_LAUNCH_OK();
}

int gfx_onepole_f32(const float* u, const float* z_alpha, float* out, int64_t R, int64_t L, int64_t Lout,
                    int64_t iir_len, int relu, void* stream) {
    if (!u || !z_alpha || !out || R <= 0 || L <= 0 || Lout <= 0 || iir_len < 1 || R > 0x7fffffffLL) return GFX_EINVAL;
    const gfx_rowmap_t none = {1, 0, 0, 0};
    hipLaunchKernelGGL(onepole_kernel<false>, dim3((unsigned)R), dim3(DT), 0, (hipStream_t)stream, u, none, 1, z_alpha, out, L,
                       Lout, iir_len, relu, (uint32_t*)nullptr);
    return GFX_LAUNCH_OK();
}

int gfx_onepole_energy_f32(const float* x, gfx_rowmap_t xmap, int64_t C, const float* z_alpha, float* out, int64_t R,
                           int64_t L, int64_t Lout, int64_t iir_len, int relu, uint32_t* rowmax, void* stream) {
    if (!x || !z_alpha || !out || R <= 0 || L <= 0 || Lout <= 0 || iir_len < 1 || R > 0x7fffffffLL || (C != 1 && C != 2) ||
        xmap.inner <= 0)
        return GFX_EINVAL;
    hipLaunchKernelGGL(onepole_kernel<true>, dim3((unsigned)R), dim3(DT), 0, (hipStream_t)stream, x, xmap, (int)C, z_alpha, out,
                       L, Lout, iir_len, relu, rowmax);
    return GFX_LAUNCH_OK();
}

int gfx_onepole_fir_f32(const float* z_alpha, float* h, int64_t R, int64_t iir_len, void* stream) {
    if (!z_alpha || !h || R <= 0 || iir_len < 1) return GFX_EINVAL;
    hipLaunchKernelGGL(onepole_fir_kernel, row_grid(R, iir_len), dim3(256), 0, (hipStream_t)stream, z_alpha, h, R, iir_len);
    return GFX_LAUNCH_OK();
}

int gfx_ballistics_bwd_f32(const float* x, const float* y, const float* g, const float* z_alpha, float* gx, float* gz,
                           int64_t R, int64_t L, void* stream) {
    if (!x || !y || !g || !z_alpha || !gx || !gz || R <= 0 || L <= 0) return GFX_EINVAL;
    hipLaunchKernelGGL(ballistics_bwd_kernel<64>, dim3((unsigned)((R + BROWS - 1) / BROWS)), dim3(64), 0,
                       (hipStream_t)stream, x, y, g, z_alpha, gx, gz, R, L, (int64_t)0, (float*)nullptr);
    return GFX_LAUNCH_OK();
}

// chunks of the chunked adjoint: enough workgroups for ~16 waves per CU, chunks of at least 4096 samples (the 2048-sample
// warm-up is walked on top of every chunk), a multiple of the 64-sample tile
static int64_t ballistics_bwd_chunk(int64_t R, int64_t L, int* chunks) {
    const int64_t groups = (R + BROWS - 1) / BROWS;
    int64_t want = (4096 + groups - 1) / groups;
    const int64_t most = L / 4096 > 1 ? L / 4096 : 1;
    if (want > most) want = most;
    if (want > 1024) want = 1024;
    if (want < 1) want = 1;
    int64_t chunk = (L + want - 1) / want;
    chunk = (chunk + 63) / 64 * 64;
    *chunks = (int)((L + chunk - 1) / chunk);
    return chunk;
}

size_t gfx_ballistics_bwd_ws_bytes(int64_t R, int64_t L) {
    if (R <= 0 || L <= 0) return 0;
    int chunks;
    ballistics_bwd_chunk(R, L, &chunks);
    return (size_t)chunks * R * 4 * sizeof(float);   // partial sums and chunk aggregates, (chunks, R, 2) each
}

int gfx_ballistics_bwd_ws_f32(const float* x, const float* y, const float* g, const float* z_alpha, float* gx, float* gz,
                              int64_t R, int64_t L, void* ws, size_t ws_bytes, void* stream) {
    if (!x || !y || !g || !z_alpha || !gx || !gz || R <= 0 || L <= 0) return GFX_EINVAL;
    int chunks;
    const int64_t chunk = ballistics_bwd_chunk(R, L, &chunks);
    if (chunks <= 1) return gfx_ballistics_bwd_f32(x, y, g, z_alpha, gx, gz, R, L, stream);
    if (!ws || ws_bytes < (size_t)chunks * R * 4 * sizeof(float)) return GFX_ENOSPC;
    float* part = (float*)ws;
    float* agg = part + (size_t)chunks * R * 2;
    const dim3 grid((unsigned)((R + BROWS - 1) / BROWS), (unsigned)chunks);
    hipStream_t st = (hipStream_t)stream;
    // (32-column tiles: 28 KB of LDS per one-wave workgroup, five per CU; 64 columns 6.4 ms, 32 4.9, 16 5.6 at 9216 rows)
    hipLaunchKernelGGL((ballistics_bwd_kernel<32, true>), grid, dim3(64), 0, st, x, y, g, z_alpha, gx, gz, R, L, chunk, part, agg);
    hipLaunchKernelGGL(ballistics_bwd_carry_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st, agg, R, chunks);
    hipLaunchKernelGGL((ballistics_bwd_kernel<32, false>), grid, dim3(64), 0, st, x, y, g, z_alpha, gx, gz, R, L, chunk, part, agg);
    hipLaunchKernelGGL(ballistics_bwd_finish_kernel, dim3((unsigned)((2 * R + 255) / 256)), dim3(256), 0, st, (const float*)part,
                       z_alpha, gz, R, chunks);
    return GFX_LAUNCH_OK();
}

int gfx_dyn_gain_f32(const float* env, float* gain, const float* log_threshold, const float* log_ratio,
                     const float* log_knee, int64_t R, int64_t L, int knee, int gate, int log_out, void* stream) {
    if (!env || !gain || !log_threshold || !log_ratio || R <= 0 || L <= 0) return GFX_EINVAL;
    if (knee < 0 || knee > 2 || (knee != 0 && !log_knee)) return GFX_EINVAL;
    hipLaunchKernelGGL(dyn_gain_kernel, row_grid(R, L), dim3(256), 0, (hipStream_t)stream, env, gain, log_threshold,
                       log_ratio, log_knee, R, L, knee, gate, log_out);
    return GFX_LAUNCH_OK();
}

int gfx_dyn_gain_bwd_f32(const float* x, gfx_rowmap_t xmap, const float* gy, gfx_rowmap_t gmap, const float* env,
                         const float* log_threshold, const float* log_ratio, const float* log_knee, int64_t R, int64_t C,
                         int64_t L, int knee, int gate, float* gain, float* denv, float* gparams, void* stream) {
    if (!x || !gy || !env || !log_threshold || !log_ratio || !gain || !denv || !gparams) return GFX_EINVAL;
    if (R <= 0 || L <= 0 || (C != 1 && C != 2) || knee < 0 || knee > 2 || (knee != 0 && !log_knee)) return GFX_EINVAL;
    hipLaunchKernelGGL(dyn_gain_bwd_kernel, row_grid(R, L), dim3(256), 0, (hipStream_t)stream, x, xmap, gy, gmap, env,
                       log_threshold, log_ratio, log_knee, R, L, (int)C, knee, gate, gain, denv, gparams);
    return GFX_LAUNCH_OK();
}

int gfx_dynamics_bwd_f32(const float* x, gfx_rowmap_t xmap, const float* gy, gfx_rowmap_t gmap,
                         const float* log_threshold, const float* log_ratio, const float* log_knee,
                         const float* z_alpha, int64_t R, int64_t C, int64_t L, int64_t iir_len, int knee, int gate,
                         float* gx, gfx_rowmap_t gxmap, float* gparams, float* denv, float* u1, float* dalpha,
                         void* stream) {
    if (!x || !gy || !log_threshold || !log_ratio || !z_alpha || !gx || !gparams || !u1) return GFX_EINVAL;
#ifdef GFX_DYN_BWD_AB
    if (!denv) return GFX_EINVAL;
#endif
    if (R <= 0 || L <= 0 || (C != 1 && C != 2) || iir_len < 1 || knee < 0 || knee > 2 || (knee != 0 && !log_knee))
        return GFX_EINVAL;
    if (R > 0x7fffffffLL || xmap.inner <= 0 || gmap.inner <= 0 || gxmap.inner <= 0) return GFX_EINVAL;
    DynArgs a;
    a.xmap = xmap; a.ymap = gxmap; a.R = R; a.L = L; a.N = iir_len; a.C = (int)C;
    a.smoother = 1; a.knee = knee; a.gate = gate; a.prows = (unsigned)R; a.nchunks = 1; a.chunk_tiles = 0;
    hipStream_t st = (hipStream_t)stream;
#ifdef GFX_DYN_BWD_AB   // round 1's two passes (A writes denv and u1, B reads them back), kept for A/B timing
    hipLaunchKernelGGL(dyn_bwd_a_kernel, dim3((unsigned)R), dim3(DT), 0, st, x, gy, gmap, log_threshold, log_ratio,
                       log_knee, z_alpha, denv, u1, gparams, a);
    hipLaunchKernelGGL(dyn_bwd_b_kernel, dim3((unsigned)R), dim3(DT), 0, st, x, gy, gmap, log_threshold, log_ratio,
                       log_knee, z_alpha, denv, u1, dalpha, gx, a);
#else
    (void)denv;
    hipLaunchKernelGGL(dyn_bwd_u1_kernel, dim3((unsigned)R), dim3(DT), 0, st, x, z_alpha, u1, a);
    hipLaunchKernelGGL(dyn_bwd_c_kernel, dim3((unsigned)R), dim3(DT), 0, st, x, gy, gmap, log_threshold, log_ratio,
                       log_knee, z_alpha, u1, dalpha, gparams, gx, a, (const float*)nullptr);
#endif
    return GFX_LAUNCH_OK();
}

int gfx_dynamics_bwd_u1_f32(const float* x, gfx_rowmap_t xmap, const float* gy, gfx_rowmap_t gmap,
                            const float* log_threshold, const float* log_ratio, const float* log_knee,
                            const float* z_alpha, int64_t R, int64_t C, int64_t L, int64_t iir_len, int knee, int gate,
                            float* gx, gfx_rowmap_t gxmap, float* gparams, const float* u1, float* dalpha, void* stream) {
    return gfx_dynamics_bwd_u1_ws_f32(x, xmap, gy, gmap, log_threshold, log_ratio, log_knee, z_alpha, R, C, L, iir_len, knee,
                                      gate, gx, gxmap, gparams, u1, dalpha, nullptr, 0, stream);
}

// `rescan`: u1 is SCRATCH (R x L floats) -- one-shot rows rebuild the scan inside their tiles and never touch it, the rows of
// the row kernel get theirs from dyn_bwd_u1_kernel first
static int dyn_bwd_launch(const float* x, gfx_rowmap_t xmap, const float* gy, gfx_rowmap_t gmap,
                          const float* log_threshold, const float* log_ratio, const float* log_knee,
                          const float* z_alpha, int64_t R, int64_t C, int64_t L, int64_t iir_len, int knee, int gate,
                          float* gx, gfx_rowmap_t gxmap, float* gparams, float* u1, float* dalpha, void* ws,
                          size_t ws_bytes, void* stream, bool rescan) {
    if (!x || !gy || !log_threshold || !log_ratio || !z_alpha || !gx || !gparams || !u1) return GFX_EINVAL;
    if (R <= 0 || L <= 0 || (C != 1 && C != 2) || iir_len < 1 || knee < 0 || knee > 2 || (knee != 0 && !log_knee))
        return GFX_EINVAL;
    if (R > 0x7fffffffLL || xmap.inner <= 0 || gmap.inner <= 0 || gxmap.inner <= 0) return GFX_EINVAL;
    if (ws && ws_bytes < gfx_dynamics_bwd_ws_bytes(R, L)) return GFX_ENOSPC;
    DynArgs a;
    a.xmap = xmap; a.ymap = gxmap; a.R = R; a.L = L; a.N = iir_len; a.C = (int)C;
    a.smoother = 1; a.knee = knee; a.gate = gate; a.prows = (unsigned)R; a.nchunks = 1; a.chunk_tiles = 0;
    hipStream_t st = (hipStream_t)stream;
    const float* tab = nullptr;
    const int64_t ngroups = (L + OS_GTILE - 1) / OS_GTILE;
    auto aligned = [&](const float* p, const gfx_rowmap_t& m) {
        return ((uintptr_t)p & 15) == 0 && m.stride_outer % 4 == 0 && m.stride_inner % 4 == 0 && m.stride_ch % 4 == 0;
    };
    const bool vec = L % 4 == 0 && aligned(x, xmap) && aligned(gy, gmap) && aligned(gx, gxmap) && ((uintptr_t)u1 & 15) == 0;
    if (ws && vec && L > OS_WTILE && R * ngroups <= 0x7ffffff0LL && ws_bytes >= gfx_dynamics_bwd_ws_bytes(R, L)) {
        // rows with a short smoother memory (chosen on the device, as in gfx_dynamics_fused_ws_f32) run as one-shot tiles
        // whose workgroups leave partial sums behind the pole table; the row kernel writes the other rows
        float* t = (float*)ws;
        double* partial = reinterpret_cast<double*>(t + (((size_t)R * DP_TAB + 1) & ~(size_t)1));
        hipLaunchKernelGGL(dyn_pole_table_kernel, dim3((unsigned)R), dim3(64), 0, st, z_alpha, t, R, iir_len, (unsigned*)nullptr);
        const unsigned nblocks = (unsigned)(R * ngroups);
        const dim3 grid((nblocks + 7u) & ~7u);
#define GFX_BWD_OS(K, G)                                                                                                \
    do {                                                                                                                \
        if (rescan)                                                                                                     \
            hipLaunchKernelGGL((dyn_bwd_oneshot_kernel<K, G, true>), grid, dim3(DT), 0, st, x, gy, gmap, log_threshold,  \
                               log_ratio, log_knee, (const float*)t, (const float*)u1, (const float*)dalpha, partial, gx, \
                               a, (unsigned)ngroups, nblocks);                                                          \
        else                                                                                                            \
            hipLaunchKernelGGL((dyn_bwd_oneshot_kernel<K, G, false>), grid, dim3(DT), 0, st, x, gy, gmap, log_threshold, \
                               log_ratio, log_knee, (const float*)t, (const float*)u1, (const float*)dalpha, partial, gx, \
                               a, (unsigned)ngroups, nblocks);                                                          \
    } while (0)
        if (gate) {
            if (knee == 0) GFX_BWD_OS(0, true); else if (knee == 1) GFX_BWD_OS(1, true); else GFX_BWD_OS(2, true);
        } else {
            if (knee == 0) GFX_BWD_OS(0, false); else if (knee == 1) GFX_BWD_OS(1, false); else GFX_BWD_OS(2, false);
        }
#undef GFX_BWD_OS
        hipLaunchKernelGGL(dyn_bwd_sums_kernel, dim3((unsigned)R), dim3(64), 0, st, (const double*)partial, (const float*)t,
                           gparams, dalpha, (unsigned)ngroups);
        tab = t;
    }
    if (rescan) hipLaunchKernelGGL(dyn_bwd_u1_kernel, dim3((unsigned)R), dim3(DT), 0, st, x, z_alpha, u1, a, tab);
    hipLaunchKernelGGL(dyn_bwd_c_kernel, dim3((unsigned)R), dim3(DT), 0, st, x, gy, gmap, log_threshold,
                       log_ratio, log_knee, z_alpha, u1, dalpha, gparams, gx, a, tab);
    return GFX_LAUNCH_OK();
}

int gfx_dynamics_bwd_u1_ws_f32(const float* x, gfx_rowmap_t xmap, const float* gy, gfx_rowmap_t gmap,
                               const float* log_threshold, const float* log_ratio, const float* log_knee,
                               const float* z_alpha, int64_t R, int64_t C, int64_t L, int64_t iir_len, int knee, int gate,
                               float* gx, gfx_rowmap_t gxmap, float* gparams, const float* u1, float* dalpha, void* ws,
                               size_t ws_bytes, void* stream) {
    return dyn_bwd_launch(x, xmap, gy, gmap, log_threshold, log_ratio, log_knee, z_alpha, R, C, L, iir_len, knee, gate, gx, gxmap,
                          gparams, const_cast<float*>(u1), dalpha, ws, ws_bytes, stream, false);
}

int gfx_dynamics_bwd_rescan_ws_f32(const float* x, gfx_rowmap_t xmap, const float* gy, gfx_rowmap_t gmap,
                                   const float* log_threshold, const float* log_ratio, const float* log_knee,
                                   const float* z_alpha, int64_t R, int64_t C, int64_t L, int64_t iir_len, int knee, int gate,
                                   float* gx, gfx_rowmap_t gxmap, float* gparams, float* u1_scratch, float* dalpha, void* ws,
                                   size_t ws_bytes, void* stream) {
    return dyn_bwd_launch(x, xmap, gy, gmap, log_threshold, log_ratio, log_knee, z_alpha, R, C, L, iir_len, knee, gate, gx, gxmap,
                          gparams, u1_scratch, dalpha, ws, ws_bytes, stream, true);
}

int gfx_onepole_dz_f32(const float* g, const float* U, const float* D, const float* coef, float* da, int64_t R,
                       int64_t L, int64_t N, void* stream) {
    if (!g || !U || !D || !coef || !da || R <= 0 || L <= 0 || N < 1 || R > 0x7fffffffLL) return GFX_EINVAL;
    hipLaunchKernelGGL(onepole_dz_kernel, dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, g, U, D, coef, da, L, N);
    return GFX_LAUNCH_OK();
}

int gfx_dyn_dx_f32(const float* x, gfx_rowmap_t xmap, const float* gy, gfx_rowmap_t gmap, const float* gain,
                   const float* de, float* gx, int64_t R, int64_t C, int64_t L, void* stream) {
    if (!x || !gy || !gain || !de || !gx || R <= 0 || L <= 0 || (C != 1 && C != 2)) return GFX_EINVAL;
    hipLaunchKernelGGL(dyn_dx_kernel, row_grid(R, L), dim3(256), 0, (hipStream_t)stream, x, xmap, gy, gmap, gain, de, gx,
                       R, L, (int)C);
    return GFX_LAUNCH_OK();
}

int gfx_apply_gain_f32(const float* x, gfx_rowmap_t xmap, const float* g, float* y, gfx_rowmap_t ymap, int64_t R,
                       int64_t C, int64_t L, int exp_gain, void* stream) {
    if (!x || !g || !y || R <= 0 || L <= 0 || C < 1) return GFX_EINVAL;
    hipLaunchKernelGGL(apply_gain_kernel, row_grid(R, L), dim3(256), 0, (hipStream_t)stream, x, xmap, g, y, ymap, R, L,
                       (int)C, exp_gain);
    return GFX_LAUNCH_OK();
}

int gfx_dyn_gain_apply_f32(const float* x, gfx_rowmap_t xmap, const float* env, float* y, gfx_rowmap_t ymap,
                           const float* log_threshold, const float* log_ratio, const float* log_knee, int64_t param_rows,
                           int64_t R, int64_t C, int64_t L, int knee, int gate, void* stream) {
    if (!x || !env || !y || !log_threshold || !log_ratio || R <= 0 || L <= 0 || (C != 1 && C != 2)) return GFX_EINVAL;
    if (knee < 0 || knee > 2 || (knee != 0 && !log_knee) || param_rows < 1 || param_rows > R) return GFX_EINVAL;
    auto al = [](const void* p, const gfx_rowmap_t& m) {
        return ((uintptr_t)p & 15) == 0 && m.stride_outer % 4 == 0 && m.stride_inner % 4 == 0 && m.stride_ch % 4 == 0;
    };
    const int vec = L % 4 == 0 && al(x, xmap) && al(y, ymap) && ((uintptr_t)env & 15) == 0;
    int64_t bx = (L + 4 * 256 - 1) / (4 * 256);
    if (bx > 128) bx = 128;
    hipLaunchKernelGGL(dyn_gain_apply_kernel, dim3((unsigned)bx, (unsigned)(R > 65535 ? 65535 : R)), dim3(256), 0,
                       (hipStream_t)stream, x, xmap, env, y, ymap, log_threshold, log_ratio, log_knee, R, L, (int)C, knee,
                       gate, (unsigned)param_rows, vec);
    return GFX_LAUNCH_OK();
}

int gfx_stereo_gain_f32(const float* x, gfx_rowmap_t xmap, const float* log_gain, float* y, gfx_rowmap_t ymap,
                        int64_t R, int64_t C_in, int64_t L, void* stream) {
    if (!x || !log_gain || !y || R <= 0 || L <= 0 || (C_in != 1 && C_in != 2)) return GFX_EINVAL;
    hipLaunchKernelGGL(stereo_gain_kernel, row_grid(R, L), dim3(256), 0, (hipStream_t)stream, x, xmap, log_gain, y,
                       ymap, R, L, (int)C_in);
    return GFX_LAUNCH_OK();
}

}  // extern "C"
