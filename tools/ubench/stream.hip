// Streaming-copy variants on gfx950: what bandwidth can a read+write kernel reach on this box?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/stream.hip -o tools/ubench/bin/stream && tools/ubench/bin/stream
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

using f4 = float __attribute__((ext_vector_type(4)));

template <int UNROLL, bool NT_LD, bool NT_ST>
__global__ __launch_bounds__(256) void copy_kernel(const f4* __restrict__ src, f4* __restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        f4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT_LD ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (NT_ST) __builtin_nontemporal_store(v[u], dst + i + u * stride);
            else dst[i + u * stride] = v[u];
        }
    }
    for (; i < n; i += stride) dst[i] = src[i];
}

// one workgroup per contiguous chunk (the per-row pattern of the scan kernels)
template <bool NT_ST>
__global__ __launch_bounds__(256) void chunk_kernel(const f4* __restrict__ src, f4* __restrict__ dst, size_t chunk) {
    const f4* s = src + (size_t)blockIdx.x * chunk;
    f4* d = dst + (size_t)blockIdx.x * chunk;
    for (size_t i = threadIdx.x; i < chunk; i += 512) {
        const f4 a = s[i];
        const f4 b = i + 256 < chunk ? s[i + 256] : f4{0, 0, 0, 0};
        if (NT_ST) {
            __builtin_nontemporal_store(a, d + i);
            if (i + 256 < chunk) __builtin_nontemporal_store(b, d + i + 256);
        } else {
            d[i] = a;
            if (i + 256 < chunk) d[i + 256] = b;
        }
    }
}

template <typename F>
static float time_ms(F&& launch, int iters = 5) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}

int main() {
    const size_t bytes = (size_t)8 << 30;  // 8 GiB each way
    const size_t n = bytes / sizeof(f4);
    f4 *src, *dst;
    hipMalloc(&src, bytes);
    hipMalloc(&dst, bytes);
    hipMemset(src, 1, bytes);
    hipMemset(dst, 0, bytes);
    auto report = [&](const char* name, float ms) { printf("%-44s %7.3f ms  %7.1f GB/s\n", name, ms, 2.0 * bytes / ms / 1e6); };
    report("hipMemcpy d2d", time_ms([&] { hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0); }));
    for (int grid : {2048, 8192, 32768, 131072}) {
        char nm[96];
        snprintf(nm, sizeof nm, "grid %6d unroll 1", grid);
        report(nm, time_ms([&] { copy_kernel<1, false, false><<<grid, 256>>>(src, dst, n); }));
        snprintf(nm, sizeof nm, "grid %6d unroll 4", grid);
        report(nm, time_ms([&] { copy_kernel<4, false, false><<<grid, 256>>>(src, dst, n); }));
        snprintf(nm, sizeof nm, "grid %6d unroll 4, nt store", grid);
        report(nm, time_ms([&] { copy_kernel<4, false, true><<<grid, 256>>>(src, dst, n); }));
        snprintf(nm, sizeof nm, "grid %6d unroll 4, nt load + nt store", grid);
        report(nm, time_ms([&] { copy_kernel<4, true, true><<<grid, 256>>>(src, dst, n); }));
        snprintf(nm, sizeof nm, "grid %6d unroll 8, nt load + nt store", grid);
        report(nm, time_ms([&] { copy_kernel<8, true, true><<<grid, 256>>>(src, dst, n); }));
    }
    const size_t chunk = 131072 / 4;  // one 512 KB row per workgroup
    report("one workgroup per 512 KB row", time_ms([&] { chunk_kernel<false><<<(unsigned)(n / chunk), 256>>>(src, dst, chunk); }));
    report("one workgroup per 512 KB row, nt store", time_ms([&] { chunk_kernel<true><<<(unsigned)(n / chunk), 256>>>(src, dst, chunk); }));
    return 0;
}
