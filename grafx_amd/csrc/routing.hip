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
