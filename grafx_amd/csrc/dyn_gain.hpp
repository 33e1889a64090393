// The gain computer of the dynamics processors (Compressor / NoiseGate knees, reference dynamics.py:444-489, 676-721) and
// the hardware log / exp forms the forward kernels use -- shared by dynamics.hip and ballistics.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace gfx {

__device__ __forceinline__ float sigmoidf(float z) { return 1.0f / (1.0f + expf(-z)); }
__device__ __forceinline__ float softplusf(float v) { return v > 20.0f ? v : log1pf(expf(v)); }  // torch threshold=20


// ---- gain computer -----------------------------------------------------------------------------
struct Knee {
    float T, R, invR, W, k, er;  // threshold (already -6), ratio, 1/ratio, half knee width, exp knee, exp(log_ratio)
    float inv4W, invk;           // 1 / (4 W), 1 / k: the per-sample divisions of the gain curves are multiplications by
                                 // these per-row reciprocals (an IEEE division is ~11 instructions)
    int kind;                    // 0 hard, 1 quadratic, 2 exponential
    int gate;                    // 0 compressor, 1 noise gate
};

__device__ __forceinline__ void knee_setup(Knee& q, float log_threshold, float log_ratio, float log_knee, int kind,
                                           int gate) {
    q.T = log_threshold - 6.0f;            // dynamics.py:395 / 630
    q.er = expf(log_ratio);
    q.R = 1.0f + q.er;
    q.invR = 1.0f / q.R;
    q.k = expf(log_knee);
    q.W = q.k / 2.0f;
    q.inv4W = 1.0f / (4.0f * q.W);
    q.invk = 1.0f / q.k;
    q.kind = kind;
    q.gate = gate;
}

// log-gain g(G) for log-energy G.  (The nested conditionals compile to an exec-masked region per sample; selects between
// values computed for every sample were measured and are slower, 3.49 vs 3.32 ms for 8192 rows: a wave whose samples all
// sit outside the knee skips the quadratic arm.)
__device__ __forceinline__ float log_gain(const Knee& q, float G) {
    const float d = G - q.T;
    if (!q.gate) {
        if (q.kind == 0) return fminf(G, q.T + d * q.invR) - G;                                // dynamics.py:444-453
        if (q.kind == 1) {                                                                     // 456-475
            const bool below = G < (q.T - q.W), above = G > (q.T + q.W);
            const float mid = G + (q.invR - 1.0f) * ((d + q.W) * (d + q.W)) * q.inv4W;
            const float out = below ? G : (above ? (q.T + d * q.invR) : mid);
            return out - G;
        }
        return (q.invR - 1.0f) * softplusf(q.k * d) * q.invk;                                  // 478-489
    }
    if (q.kind == 0) return fminf(G, q.R * d + q.T) - G;                                       // 676-686
    if (q.kind == 1) {                                                                         // 688-707
        const bool below = G < (q.T - q.W), above = G > (q.T + q.W);
        const float mid = G + (1.0f - q.R) * ((d - q.W) * (d - q.W)) * q.inv4W;
        const float out = below ? (q.R * d + q.T) : (above ? G : mid);
        return out - G;
    }
    return -q.er * softplusf(q.k * (-d)) * q.invk;                                             // 709-721
}

// Hardware log2 / exp2 forms (v_log_f32, v_exp_f32) for the forward kernels: the one-shot tiles run at copy speed once the
// arithmetic is out of the way -- with the library logf / expf they take 3.46 ms where the grid moving the same bytes takes
// 2.72 (8192 stereo rows, profiles/r3/dyn_oneshot_ablation.txt) -- and the row kernel uses the same forms so that a row
// gives the same samples whichever kernel produces it.  Accuracy: the
// envelope is >= 1e-5, so log() sees no denormals and is good to ~1e-7 absolute; exp(g) is good to (2 + |g| log2 e) ulp,
// i.e. a relative 6e-8 |g| on a GAIN that is itself e^g: large |g| means a proportionally small output.
struct FastMath {
    // __logf without its special cases: the same v_log_f32 and the same compensated product with ln 2 (bit-identical for
    // normal finite arguments), minus the denormal pre-scaling and the inf / nan pass-through -- 5 instructions instead of
    // 12, on an argument that is env + 1e-5 >= 1e-5 (an infinite envelope gives nan here, as the gain curve would anyway).
    // Worth 0.3 % on the one-shot tiles (3.308 vs 3.317 ms, same box): they are not bound by their instruction count.
    static __device__ __forceinline__ float log(float v) {
        const float y = __builtin_amdgcn_logf(v);
        const float c = 0x1.62e42ep-1f, cl = 0x1.efa39ep-25f;
        const float r = c * y;
        float t = fmaf(y, c, -r);
        t = fmaf(cl, y, t);
        return fmaf(c, y, t);
    }
    static __device__ __forceinline__ float exp(float v) { return __expf(v); }
    // softplus, torch threshold 20; below -15 log1p(e^v) = e^v to fp32 (and 1 + e^v would round to 1)
    static __device__ __forceinline__ float softplus(float v) {
        return v > 20.0f ? v : (v < -15.0f ? __expf(v) : __logf(1.0f + __expf(v)));
    }
};

template <typename M>
__device__ __forceinline__ float log_gain_m(const Knee& q, float G) {
    if (q.kind != 2) return log_gain(q, G);       // hard / quadratic knees: no transcendental
    const float d = G - q.T;
    if (!q.gate) return (q.invR - 1.0f) * M::softplus(q.k * d) * q.invk;   // dynamics.py:478-489
    return -q.er * M::softplus(q.k * (-d)) * q.invk;                         // 709-721
}

}  // namespace gfx
