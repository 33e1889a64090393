// Semantics probe for LDS-DMA (buffer_load_dword ... lds) on gfx950, as used by the ping-pong convolution kernel:
//   1. basic: 64 lanes x 4 B land contiguously at the LDS address given
//   2. lanes whose offset fails the buffer range check: do they write 0 to LDS, or leave it untouched?
//   3. lanes switched off in EXEC: untouched?
//   4. a 4-byte aligned (not 16-byte aligned) global address with the dwordx4 form
//   5. s_waitcnt vmcnt(0) + barrier is enough for other waves to read the data
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_dma.hip -o tools/ubench/bin/lds_dma && tools/ubench/bin/lds_dma
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

using rsrc_t = __amdgpu_buffer_rsrc_t;
#define LDSP(p) reinterpret_cast<__attribute__((address_space(3))) void*>((__attribute__((address_space(3))) char*)(p))

__device__ __forceinline__ rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    const uint64_t p = reinterpret_cast<uint64_t>(base);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p), hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

__global__ void probe(const float* __restrict__ src, float* __restrict__ out, int nsrc) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (int i = t; i < 2048; i += blockDim.x) lds[i] = -7.0f;  // sentinel
    __syncthreads();
    const rsrc_t r = make_rsrc(src, (uint32_t)nsrc * 4);
    if (wave == 0) {
        // 1. basic, soffset stepping: chunk 0 <- src[0..64), chunk 1 <- src[100..164) (4-byte aligned only)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDSP(lds + 0), 4, 4u * lane, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDSP(lds + 64), 4, 4u * lane, 400, 0, 0);
        // 2. range check: lanes >= 32 get an offset beyond num_records
        const uint32_t voff = lane < 32 ? 4u * lane : 0xffffffffu;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDSP(lds + 128), 4, voff, 0, 0, 0);
        // 2b. offsets just past the end (nsrc - 16 + lane): lanes >= 16 out of range
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDSP(lds + 192), 4, 4u * (uint32_t)(nsrc - 16 + lane), 0, 0, 0);
        // 3. EXEC-masked lanes
        if (lane < 20) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDSP(lds + 256), 4, 4u * lane, 0, 0, 0);
        // 4. dwordx4, source only 4-byte aligned (src + 1): lane l <- src[1 + 4l .. 5 + 4l)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDSP(lds + 512), 16, 16u * lane, 4, 0, 0);
        // 4b. dwordx4 with the tail lanes out of range (lane l covers floats nsrc-32+4l..+4: lanes >= 8 are out)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDSP(lds + 1024), 16, 4u * (uint32_t)(nsrc - 32) + 16u * lane, 0, 0, 0);
        __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0) only (expcnt 7, lgkmcnt 15 left alone): gfx9 encoding
    }
    __syncthreads();
    // 5. every wave reads everything back
    for (int i = t; i < 2048; i += blockDim.x) out[i] = lds[i];
}

int main() {
    const int nsrc = 4096;
    std::vector<float> h(nsrc);
    for (int i = 0; i < nsrc; ++i) h[i] = 1000.0f + i;
    float *src, *out;
    hipMalloc(&src, nsrc * 4);
    hipMalloc(&out, 2048 * 4);
    hipMemcpy(src, h.data(), nsrc * 4, hipMemcpyHostToDevice);
    probe<<<1, 256, 2048 * 4>>>(src, out, nsrc);
    std::vector<float> o(2048);
    hipError_t e = hipMemcpy(o.data(), out, 2048 * 4, hipMemcpyDeviceToHost);
    printf("status: %s\n", hipGetErrorString(e));
    auto show = [&](const char* name, int at, int n) {
        printf("%-34s", name);
        for (int i = 0; i < n; ++i) printf(" %g", o[at + i]);
        printf("\n");
    };
    show("1 basic [0..4)", 0, 4);
    show("1 basic [62..66)", 62, 4);
    show("1 soffset chunk end [126..128)", 126, 2);
    show("2 in-range lanes [128+30..+34)", 158, 4);
    show("2 OOB lanes (0 or -7?) [128+62..]", 190, 2);
    show("2b tail [192+14..+18)", 206, 4);
    show("3 exec [256+18..+22)", 274, 4);
    show("4 x4 unaligned [512..516)", 512, 4);
    show("4 x4 [512+252..+256)", 764, 4);
    show("4b x4 tail [1024+28..+36)", 1052, 8);
    show("4b x4 tail far [1024+250..]", 1274, 4);
    return 0;
}
