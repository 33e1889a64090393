// micro-benchmark: f32 VALU issue rate on gfx950 at a given waves/SIMD (run on the GPU box)
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ILP>
__global__ __launch_bounds__(256) void fma_chain(float* out, float a, float b, int iters) {
    extern __shared__ float lds[];
    float acc[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) acc[i] = threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int i = 0; i < ILP; ++i) acc[i] = fmaf(acc[i], a, b);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int ILP>
void run(int blocks_per_cu, size_t lds) {
    float* out;
    hipMalloc(&out, 4096 * 256 * sizeof(float));
    int iters = 2000;
    hipFuncSetAttribute((const void*)fma_chain<ILP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    int blocks = 256 * blocks_per_cu;
    fma_chain<ILP><<<blocks, 256, lds>>>(out, 1.0001f, 0.5f, 10);
    hipEventRecord(a);
    fma_chain<ILP><<<blocks, 256, lds>>>(out, 1.0001f, 0.5f, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double instr_per_wave = (double)iters * 16 * ILP;
    // each SIMD hosts blocks_per_cu waves (4 waves/block over 4 SIMDs)
    double cyc = ms * 1e-3 * 2.4e9;  // at nominal 2.4 GHz
    printf("ILP=%d waves/SIMD=%d: %.3f ms -> %.2f cycles (at 2.4GHz) per VALU instr per SIMD\n", ILP, blocks_per_cu, ms,
           cyc / (instr_per_wave * blocks_per_cu));
    hipFree(out);
}
int main() {
    run<1>(1, 100000); run<4>(1, 100000); run<8>(1, 100000);
    run<1>(2, 70000); run<4>(2, 70000); run<8>(2, 70000);
    run<8>(4, 36000); run<8>(8, 16000);
    return 0;
}
