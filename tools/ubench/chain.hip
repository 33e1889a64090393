// Latency of the ballistics step as a dependent chain, one wave, registers only:
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/chain.hip -o tools/ubench/bin/chain && tools/ubench/bin/chain
// Variants: (a) two candidates + select (v_mul, v_add, v_cndmask on the chain), (b) select the coefficient first
// (v_cmp, 2 x v_cndmask, v_mul, v_add), (c) a single fused multiply-add chain (reference for the machine's dependent latency),
// each with 16 / 64 active lanes and with 1 / 2 / 4 independent chains per lane.  Reports shader cycles per step
// (clock64) and the shader clock (clock64 / wall_clock64 at 100 MHz).
#include <hip/hip_runtime.h>

#include <cstdio>
#pragma clang fp contract(off)

template <int VAR, int ILP>
__global__ void chain(const float* __restrict__ xin, float* __restrict__ out, long long* __restrict__ cyc, int n, int lanes,
                      float at, float rt) {
    const int lane = threadIdx.x;
    if (lane >= lanes) return;
    float s[ILP];
    for (int k = 0; k < ILP; ++k) s[k] = 1.0f + 0.01f * k + 0.001f * lane;
    const float oa = 1.0f - at, orr = 1.0f - rt;
    float x = xin[lane];
    const long long w0 = wall_clock64();
    const long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < ILP; ++k) {
            if (VAR == 0) {
                const float ya = oa * s[k] + at * x, yr = orr * s[k] + rt * x;
                s[k] = x < s[k] ? ya : yr;
            } else if (VAR == 1) {
                const bool a = x < s[k];
                const float c = a ? at : rt, o = a ? oa : orr;
                s[k] = o * s[k] + c * x;
            } else {
                s[k] = __builtin_fmaf(oa, s[k], x);
            }
        }
        x = x * 1.0001f + 0.37f;   // (off the chains)
        if (x > 3.0f) x -= 2.9f;
    }
    const long long t1 = clock64();
    const long long w1 = wall_clock64();
    float acc = 0.0f;
    for (int k = 0; k < ILP; ++k) acc += s[k];
    out[lane] = acc;
    if (lane == 0) {
        cyc[0] = t1 - t0;
        cyc[1] = w1 - w0;
    }
}

template <int VAR, int ILP>
void run(const char* name, int lanes) {
    float *x, *o;
    long long* c;
    hipMalloc(&x, 256);
    hipMalloc(&o, 256);
    hipMalloc(&c, 16);
    hipMemset(x, 0, 256);
    const int n = 200000;
    chain<VAR, ILP><<<1, 64>>>(x, o, c, n, lanes, 0.3f, 0.6f);
    chain<VAR, ILP><<<1, 64>>>(x, o, c, n, lanes, 0.3f, 0.6f);
    long long h[2];
    hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
    printf("%-34s lanes %2d  chains/lane %d: %6.1f cycles per step and chain-set, shader clock %.0f MHz\n", name, lanes, ILP,
           (double)h[0] / n, (double)h[0] / (double)h[1] * 100.0);
    hipFree(x); hipFree(o); hipFree(c);
}

int main() {
    run<2, 1>("fma chain", 64);
    run<0, 1>("two candidates + select", 64);
    run<0, 1>("two candidates + select", 16);
    run<0, 2>("two candidates + select", 16);
    run<0, 4>("two candidates + select", 16);
    run<1, 1>("select coefficient, then step", 16);
    run<1, 4>("select coefficient, then step", 16);
    return 0;
}
