#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ uint32_t lds_b32(uint32_t byte_addr) {
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(byte_addr));
    return v;
}
__global__ void k_probe(uint32_t* out) {
    __shared__ uint16_t s[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) s[i] = (uint16_t)i;
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)s;   // LDS offset (low 32 bits of the shared-aperture address)
    out[threadIdx.x] = lds_b32(base + 2 * threadIdx.x + 2 * (threadIdx.x & 1) * 0 + 2);     // odd halfword index for every lane: unaligned for even (lane+1)
    out[64 + threadIdx.x] = lds_b32(base + 2 * (threadIdx.x * 7 % 500) + 2);
}
// throughput: random pair reads, aligned vs unaligned addresses
template <int UNALIGNED>
__global__ __launch_bounds__(1024) void k_tp(const uint32_t* idx, uint32_t* out, int iters) {
    extern __shared__ uint16_t st[];
    for (int i = threadIdx.x; i < 36000; i += 1024) st[i] = (uint16_t)i;
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)st;
    uint32_t a = idx[blockIdx.x * 1024 + threadIdx.x] % 35000u, acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint32_t addr = UNALIGNED ? (a * 2u) : ((a * 2u) & ~3u);
        uint32_t v;
        asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(base + addr));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        acc += v;
        a = (a * 1664525u + 1013904223u + v) % 35000u;
    }
    out[blockIdx.x * 1024 + threadIdx.x] = acc;
}
int main() {
    uint32_t* d; CK(hipMalloc(&d, 4 * 128));
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d);
    uint32_t h[128]; CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l) { uint32_t i = l + 1; uint32_t e = i | ((i + 1) << 16); if (h[l] != e) { if (bad < 5) printf("lane %d got %08x expect %08x\n", l, h[l], e); ++bad; } }
    for (int l = 0; l < 64; ++l) { uint32_t i = (l * 7 % 500) + 1; uint32_t e = i | ((i + 1) << 16); if (h[64 + l] != e) { if (bad < 10) printf("lane %d got %08x expect %08x\n", l, h[64 + l], e); ++bad; } }
    printf("unaligned ds_read_b32: %d mismatches of 128\n", bad);
    const int nb = 256, iters = 2000;
    std::vector<uint32_t> idx(nb * 1024); uint32_t s = 12345; for (auto& v : idx) { s = s * 1664525u + 1013904223u; v = s >> 8; }
    uint32_t *di, *dout; CK(hipMalloc(&di, idx.size() * 4)); CK(hipMalloc(&dout, idx.size() * 4));
    CK(hipMemcpy(di, idx.data(), idx.size() * 4, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void*)k_tp<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 72000));
    CK(hipFuncSetAttribute((const void*)k_tp<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 72000));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int u = 0; u < 2; ++u) for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        if (u) hipLaunchKernelGGL(k_tp<1>, dim3(nb), dim3(1024), 72000, 0, di, dout, iters);
        else hipLaunchKernelGGL(k_tp<0>, dim3(nb), dim3(1024), 72000, 0, di, dout, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        // per CU: 16 waves x iters wave-instructions
        if (rep) printf("%s random ds_read_b32: %.1f cycles per wave-instruction per CU (16 waves, dependent chain; 2.4 GHz)\n", u ? "unaligned" : "aligned  ", ms * 1e-3 * 2.4e9 / (16.0 * iters));
    }
    return 0;
}
