// Library-level entry points of the C ABI (include/grafx_amd.h).
#include <hip/hip_runtime.h>

#include "../../include/grafx_amd.h"

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
