// Library-level entry points of the C ABI (include/grafx_amd.h).
#include <hip/hip_runtime.h>

#include <mutex>

#include "../../include/grafx_amd.h"
#include "fft_tile.hpp"

namespace gfx {

__global__ void tile_twiddle_table_kernel(float2* __restrict__ table) {
    const int t = threadIdx.x;
    const int row = blockIdx.x;
    int num;
    double den;
    if (row < 4) { num = t * row; den = 8192.0; }
    else if (row < 12) { num = t * 4 * (row - 4); den = 8192.0; }
    else if (row < 16) { num = (t & 15) * (row - 12); den = 256.0; }
    else { num = (t & 15) * 4 * (row - 16); den = 256.0; }
    double s, c;
    sincospi(2.0 * (double)num / den, &s, &c);
    const float2 v = make_float2((float)c, (float)(-s));
    table[row * TILE_T + t] = v;
    table[TW_ROWS * TILE_T + (row >> 1) * 2 * TILE_T + 2 * t + (row & 1)] = v;   // the float4-pair copy
}

const float2* tile_twiddle_table(hipStream_t stream) {
    static std::mutex mu;
    static float2* tables[64] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!tables[dev]) {
        float2* p = nullptr;
        if (hipMalloc(&p, sizeof(float2) * TW_TABLE_F2) != hipSuccess) return nullptr;
        hipLaunchKernelGGL(tile_twiddle_table_kernel, dim3(TW_ROWS), dim3(TILE_T), 0, stream, p);
        // make the table visible to every stream before anyone else can use it
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) {
            hipFree(p);
            return nullptr;
        }
        tables[dev] = p;
    }
    return tables[dev];
}



}  // namespace gfx

extern "C" {

int gfx_abi_version(void) { return 1; }

int gfx_device_info(int* n_cu, size_t* lds_bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return GFX_ELAUNCH;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return GFX_ELAUNCH;
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (lds_bytes) *lds_bytes = prop.maxSharedMemoryPerMultiProcessor;
    return GFX_OK;
}

}  // extern "C"
