// Signal routing for render_grafx: gather rows of the signal buffer and sum them per destination.
//
// Replaces, in one pass over HBM, the reference's read -> aggregate -> write chain for `mix`/`out`
// nodes and for any node with indexed or multiple inputs:
//   read_single_tensor ("index": index_select)            render/core.py:36-50
//   aggregate_tensor   ("sum" / "scatter" via PyG scatter) render/core.py:101-112
//   inplace_write_tensor                                    render/core.py:80-98
//   out[b, j, c, n] = sum_{e in [seg[j], seg[j+1])} buf[b, src[e], c, n]
// Edges arrive sorted by (destination, source) (prepare.py:115-119), so every destination's
// sources are one contiguous run and are added in that order.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/grafx_amd.h"

#ifdef GFX_NT_OFF
#define GFX_NT_STORE(...) gfx_plain_store(__VA_ARGS__)
template <typename T> __device__ __forceinline__ void gfx_plain_store(T v, T* p) { *p = v; }
#else
#define GFX_NT_STORE(...) __builtin_nontemporal_store(__VA_ARGS__)
#endif

#ifndef GFX_GATHER_UNROLL
#define GFX_GATHER_UNROLL 1
#endif

namespace gfx {

// streamed outputs: non-temporal, so they do not push the rows still being read out of L2
__device__ __forceinline__ void nt_store(float4* p, float4 v) {
    using f4 = float __attribute__((ext_vector_type(4)));
    GFX_NT_STORE(f4{v.x, v.y, v.z, v.w}, reinterpret_cast<f4*>(p));
}

__global__ __launch_bounds__(256) void gather_sum_kernel(const float* __restrict__ buf, int64_t buf_sb, int64_t buf_sv,
                                                         int64_t buf_sc, const int64_t* __restrict__ src,
                                                         const int64_t* __restrict__ seg, float* __restrict__ out,
                                                         int64_t out_sb, int64_t out_sv, int64_t out_sc, int C,
                                                         int64_t L, int vec) {
    const int j = blockIdx.y / C, c = blockIdx.y % C;
    const int64_t b = blockIdx.z;
    const int64_t e0 = seg[j], e1 = seg[j + 1];
    const float* base = buf + b * buf_sb + (int64_t)c * buf_sc;
    float* dst = out + b * out_sb + (int64_t)j * out_sv + (int64_t)c * out_sc;
    if (vec) {
        const int64_t L4 = L >> 2;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < L4; i += (int64_t)gridDim.x * blockDim.x) {
            float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            for (int64_t e = e0; e < e1; ++e) {
                const float4 v = reinterpret_cast<const float4*>(base + src[e] * buf_sv)[i];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            nt_store(reinterpret_cast<float4*>(dst) + i, acc);
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < L; i += (int64_t)gridDim.x * blockDim.x) {
            float acc = 0.0f;
            for (int64_t e = e0; e < e1; ++e) acc += base[src[e] * buf_sv + i];
            dst[i] = acc;
        }
    }
}

// Source-major variant for fan-out routing (a source feeding several destinations, e.g. every channel
// strip -> its bus AND the send): each source row is read ONCE and added to up to NJ destination
// accumulators selected by a per-source bit mask.  Per destination the sources still arrive in
// increasing order, i.e. the same summation order as the destination-major kernel.
template <int NJ>
__global__ __launch_bounds__(256) void gather_sum_fanout_kernel(const float* __restrict__ buf, int64_t buf_sb,
                                                                int64_t buf_sv, int64_t buf_sc,
                                                                const int64_t* __restrict__ usrc,
                                                                const int64_t* __restrict__ dmask, int U,
                                                                float* __restrict__ out, int64_t out_sb,
                                                                int64_t out_sv, int64_t out_sc, int J, int64_t L4) {
    const int c = blockIdx.y;
    const int64_t b = blockIdx.z;
    const float* base = buf + b * buf_sb + (int64_t)c * buf_sc;
    float* dst = out + b * out_sb + (int64_t)c * out_sc;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < L4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 acc[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        // The rows are read once: non-temporal loads (2.02 vs 2.14 ms per step for the two routing stages).  Requesting
        // GFX_GATHER_UNROLL = 2 / 4 / 8 source rows before the first addition was measured and is NOT faster (2.11-2.14
        // ms): with 65 k one-shot workgroups the kernel is not short of loads in flight.  Additions stay in source order.
        using f4 = float __attribute__((ext_vector_type(4)));
        auto row = [&](int u) { return __builtin_nontemporal_load(reinterpret_cast<const f4*>(base + usrc[u] * buf_sv) + i); };
        auto add = [&](f4 v, unsigned mask) {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                if ((mask >> j) & 1u) {
                    acc[j].x += v.x; acc[j].y += v.y; acc[j].z += v.z; acc[j].w += v.w;
                }
        };
        int u = 0;
        for (; u + GFX_GATHER_UNROLL <= U; u += GFX_GATHER_UNROLL) {
            f4 v[GFX_GATHER_UNROLL];
#pragma unroll
            for (int k = 0; k < GFX_GATHER_UNROLL; ++k) v[k] = row(u + k);
#pragma unroll
            for (int k = 0; k < GFX_GATHER_UNROLL; ++k) add(v[k], (unsigned)dmask[u + k]);
        }
        for (; u < U; ++u) add(row(u), (unsigned)dmask[u]);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            if (j < J) nt_store(reinterpret_cast<float4*>(dst + (int64_t)j * out_sv) + i, acc[j]);
    }
}

// StereoGain (stereo.py:25-48: y[r,c,n] = x[r,cx,n] exp(log_gain[r,c])) with the routing sum that follows fused in -- the
// gain / pan stage in front of a bus, the elementwise twin of dyn_oneshot_mix_kernel (dynamics.hip: same schedule words,
// same extras, same increasing summation order, bit-identical sums).  One thread per 16-byte column of a graph: it walks
// the `inner` rows, scales, stores the row and adds it to the accumulators of the destinations the row feeds.
struct GainMixArgs {
    const float* x;
    float* y;
    gfx_rowmap_t xmap, ymap;
    const float* log_gain;                // (R, 2)
    const int64_t* sched;                 // [inner], see gfx_dynamics_fused_mix_f32
    float* out;
    int64_t sb, sv, sc;
    const int64_t* extras;
    int inner, n_pre, n_post, Cin;
    int64_t L4;
};

__device__ __forceinline__ int64_t gm_row_off(const gfx_rowmap_t& m, unsigned r, int c) {
    const unsigned inner = (unsigned)m.inner;
    const unsigned q = r / inner, rem = r - q * inner;
    return (int64_t)q * m.stride_outer + (int64_t)rem * m.stride_inner + (int64_t)c * m.stride_ch;
}

template <int NA>
__global__ __launch_bounds__(256) void gain_mix_kernel(GainMixArgs a) {
    const unsigned g = blockIdx.y;
    float* const obase = a.out + (int64_t)g * a.sb;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.L4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 acc0[NA], acc1[NA];
#pragma unroll
        for (int c = 0; c < NA; ++c) acc0[c] = acc1[c] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        auto settle = [&](uint64_t code, float4 v0, float4 v1) {
            const unsigned add = (unsigned)code & 15u;
#pragma unroll
            for (int c = 0; c < NA; ++c) {
                if ((add >> c) & 1u) {
                    acc0[c].x += v0.x; acc0[c].y += v0.y; acc0[c].z += v0.z; acc0[c].w += v0.w;
                    acc1[c].x += v1.x; acc1[c].y += v1.y; acc1[c].z += v1.z; acc1[c].w += v1.w;
                }
                const unsigned fl = (unsigned)(code >> (8 + 8 * c)) & 255u;
                if (fl != 0u) {
                    float* o = obase + (int64_t)(fl - 1u) * a.sv;
                    nt_store(reinterpret_cast<float4*>(o) + i, acc0[c]);
                    nt_store(reinterpret_cast<float4*>(o + a.sc) + i, acc1[c]);
                    acc0[c] = acc1[c] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                }
            }
        };
        auto extra = [&](int e) {
            const float* p = obase + a.extras[2 * e] * a.sv;
            settle((uint64_t)a.extras[2 * e + 1], reinterpret_cast<const float4*>(p)[i],
                   reinterpret_cast<const float4*>(p + a.sc)[i]);
        };
        for (int e = 0; e < a.n_pre; ++e) extra(e);
        for (int j = 0; j < a.inner; ++j) {
            const unsigned r = g * (unsigned)a.inner + (unsigned)j;
            const float g0 = expf(a.log_gain[2 * (int64_t)r]), g1 = expf(a.log_gain[2 * (int64_t)r + 1]);
            const float4 x0 = reinterpret_cast<const float4*>(a.x + gm_row_off(a.xmap, r, 0))[i];
            const float4 x1 = a.Cin == 2 ? reinterpret_cast<const float4*>(a.x + gm_row_off(a.xmap, r, 1))[i] : x0;
            const float4 y0 = make_float4(x0.x * g0, x0.y * g0, x0.z * g0, x0.w * g0);
            const float4 y1 = make_float4(x1.x * g1, x1.y * g1, x1.z * g1, x1.w * g1);
            nt_store(reinterpret_cast<float4*>(a.y + gm_row_off(a.ymap, r, 0)) + i, y0);
            nt_store(reinterpret_cast<float4*>(a.y + gm_row_off(a.ymap, r, 1)) + i, y1);
            settle((uint64_t)a.sched[j], y0, y1);
        }
        for (int e = a.n_pre; e < a.n_pre + a.n_post; ++e) extra(e);
    }
}

}  // namespace gfx

extern "C" int gfx_gather_sum_fanout_f32(const float* buf, int64_t buf_sb, int64_t buf_sv, int64_t buf_sc,
                                         const int64_t* unique_src, const int64_t* dest_mask, int64_t U, float* out,
                                         int64_t out_sb, int64_t out_sv, int64_t out_sc, int64_t B, int64_t J,
                                         int64_t C, int64_t L, void* stream) {
    if (!buf || !unique_src || !dest_mask || !out || B <= 0 || J <= 0 || J > 32 || C <= 0 || L <= 0 || U <= 0)
        return GFX_EINVAL;
    if (B > 65535 || C > 65535) return GFX_EINVAL;
    const bool aligned = (((uintptr_t)buf | (uintptr_t)out) & 15) == 0 && (L % 4 == 0) &&
                         ((buf_sb | buf_sv | buf_sc | out_sb | out_sv | out_sc) % 4 == 0);
    if (!aligned) return GFX_EINVAL;  // callers fall back to gfx_gather_sum_f32
    int64_t bx = (L / 4 + 255) / 256;
    if (bx > 512) bx = 512;
    // up to 8 destinations: the forward routing sums (many strips -> a few buses); up to 32: their adjoints (a few bus
    // gradients -> every strip), 32 accumulators = 128 VGPRs
    if (J <= 8)
        hipLaunchKernelGGL(gfx::gather_sum_fanout_kernel<8>, dim3((unsigned)bx, (unsigned)C, (unsigned)B), dim3(256), 0,
                           (hipStream_t)stream, buf, buf_sb, buf_sv, buf_sc, unique_src, dest_mask, (int)U, out, out_sb,
                           out_sv, out_sc, (int)J, L / 4);
    else
        hipLaunchKernelGGL(gfx::gather_sum_fanout_kernel<32>, dim3((unsigned)bx, (unsigned)C, (unsigned)B), dim3(256), 0,
                           (hipStream_t)stream, buf, buf_sb, buf_sv, buf_sc, unique_src, dest_mask, (int)U, out, out_sb,
                           out_sv, out_sc, (int)J, L / 4);
    return hipGetLastError() == hipSuccess ? GFX_OK : GFX_ELAUNCH;
}

extern "C" int gfx_gather_sum_f32(const float* buf, int64_t buf_sb, int64_t buf_sv, int64_t buf_sc, const int64_t* src_idx,
                                  const int64_t* seg_ptr, float* out, int64_t out_sb, int64_t out_sv, int64_t out_sc,
                                  int64_t B, int64_t J, int64_t C, int64_t L, void* stream) {
    if (!buf || !src_idx || !seg_ptr || !out || B <= 0 || J <= 0 || C <= 0 || L <= 0) return GFX_EINVAL;
    if (B > 65535 || J * C > 65535) return GFX_EINVAL;
    const bool aligned = (((uintptr_t)buf | (uintptr_t)out) & 15) == 0 && (L % 4 == 0) &&
                         ((buf_sb | buf_sv | buf_sc | out_sb | out_sv | out_sc) % 4 == 0);
    const int64_t work = aligned ? L / 4 : L;
    int64_t bx = (work + 255) / 256;
    if (bx > 512) bx = 512;
    hipLaunchKernelGGL(gfx::gather_sum_kernel, dim3((unsigned)bx, (unsigned)(J * C), (unsigned)B), dim3(256), 0,
                       (hipStream_t)stream, buf, buf_sb, buf_sv, buf_sc, src_idx, seg_ptr, out, out_sb, out_sv, out_sc,
                       (int)C, L, aligned ? 1 : 0);
    return hipGetLastError() == hipSuccess ? GFX_OK : GFX_ELAUNCH;
}

extern "C" int gfx_stereo_gain_mix_f32(const float* x, gfx_rowmap_t xmap, const float* log_gain, float* y, gfx_rowmap_t ymap,
                                       int64_t R, int64_t C_in, int64_t L, const int64_t* sched, int64_t inner,
                                       int64_t n_acc, float* mix, int64_t mix_sb, int64_t mix_sv, int64_t mix_sc,
                                       const int64_t* extras, int64_t n_pre, int64_t n_post, void* stream) {
    if (!x || !log_gain || !y || !sched || !mix || R <= 0 || L <= 0 || (C_in != 1 && C_in != 2)) return GFX_EINVAL;
    if (inner < 1 || R % inner != 0 || R / inner > 65535 || n_acc < 1 || n_acc > 4) return GFX_EINVAL;
    if (n_pre < 0 || n_post < 0 || (n_pre + n_post > 0 && !extras)) return GFX_EINVAL;
    if (xmap.inner != inner || ymap.inner != inner) return GFX_EINVAL;
    const int64_t strides = xmap.stride_outer | xmap.stride_inner | xmap.stride_ch | ymap.stride_outer | ymap.stride_inner |
                            ymap.stride_ch | mix_sb | mix_sv | mix_sc;
    if ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)mix) & 15) != 0 || (strides & 3) != 0 || (L & 3) != 0) return GFX_EINVAL;
    gfx::GainMixArgs a;
    a.x = x; a.y = y; a.xmap = xmap; a.ymap = ymap; a.log_gain = log_gain; a.sched = sched; a.out = mix;
    a.sb = mix_sb; a.sv = mix_sv; a.sc = mix_sc; a.extras = extras; a.inner = (int)inner; a.n_pre = (int)n_pre;
    a.n_post = (int)n_post; a.Cin = (int)C_in; a.L4 = L / 4;
    int64_t bx = (L / 4 + 255) / 256;
    if (bx > 512) bx = 512;
    const dim3 grid((unsigned)bx, (unsigned)(R / inner));
    if (n_acc <= 2) hipLaunchKernelGGL(gfx::gain_mix_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(gfx::gain_mix_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? GFX_OK : GFX_ELAUNCH;
}
