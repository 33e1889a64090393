// micro-benchmark: packed-FP32 (v_pk_fma_f32 / v_pk_add_f32) issue rate vs scalar v_fma_f32 on gfx950
//   hipcc --offload-arch=gfx950 -O3 pk_rate.hip -o pk_rate && ./pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2 __attribute__((ext_vector_type(2)));
enum { SCALAR = 0, PK_VGPR = 1, PK_SGPR = 2, PK_OPSEL = 3, PK_ADD = 4, PK_ADD_OPSEL = 5 };
template <int MODE, int ILP>
__global__ __launch_bounds__(256) void chain(float* out, float a, float b, int iters) {
    extern __shared__ float lds[];
    v2 acc[ILP];
    v2 va = {a, a * 1.0001f}, vb = {b, b * 0.5f};
#pragma unroll
    for (int i = 0; i < ILP; ++i) acc[i] = v2{(float)threadIdx.x + i, (float)i};
    if (MODE == PK_VGPR || MODE == PK_OPSEL || MODE == PK_ADD || MODE == PK_ADD_OPSEL) {  // force operands into VGPRs
        va.x += threadIdx.x * 1e-9f;
        vb.x += threadIdx.x * 1e-9f;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                if (MODE == SCALAR) {
                    acc[i].x = fmaf(acc[i].x, a, b);
                } else if (MODE == PK_VGPR || MODE == PK_SGPR) {
                    acc[i] = __builtin_elementwise_fma(acc[i], va, vb);
                } else if (MODE == PK_OPSEL) {
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(acc[i]) : "v"(acc[i]), "v"(va), "v"(vb));
                } else if (MODE == PK_ADD) {
                    acc[i] = acc[i] + va;
                } else {
                    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_lo:[0,1] neg_hi:[1,0]" : "=v"(acc[i]) : "v"(acc[i]), "v"(va));
                }
            }
    }
    v2 s = {0, 0};
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}
template <int MODE, int ILP>
void run(const char* name, int blocks_per_cu, size_t lds) {
    float* out;
    hipMalloc(&out, 4096 * 256 * sizeof(float));
    int iters = 2000;
    hipFuncSetAttribute((const void*)chain<MODE, ILP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    int blocks = 256 * blocks_per_cu;
    chain<MODE, ILP><<<blocks, 256, lds>>>(out, 1.0001f, 0.5f, 10);
    hipEventRecord(a);
    chain<MODE, ILP><<<blocks, 256, lds>>>(out, 1.0001f, 0.5f, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double instr_per_wave = (double)iters * 16 * ILP;
    double cyc = ms * 1e-3 * 2.4e9;
    printf("%-14s ILP=%d waves/SIMD=%d: %.3f ms -> %.2f cycles (2.4GHz) per instr per SIMD\n", name, ILP, blocks_per_cu, ms,
           cyc / (instr_per_wave * blocks_per_cu));
    hipFree(out);
}
#define ALL(MODE, name) run<MODE, 1>(name, 2, 70000); run<MODE, 4>(name, 2, 70000); run<MODE, 8>(name, 2, 70000); run<MODE, 8>(name, 1, 100000);
int main() {
    ALL(SCALAR, "v_fma_f32")
    ALL(PK_VGPR, "pk_fma vgpr")
    ALL(PK_SGPR, "pk_fma sgpr")
    ALL(PK_OPSEL, "pk_fma opsel")
    ALL(PK_ADD, "pk_add")
    ALL(PK_ADD_OPSEL, "pk_add opsel")
    return 0;
}
