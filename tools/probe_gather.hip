#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
// every lane: ITER independent gathers of W dwords from a table of `span` 16-byte slots; pattern: 0 = all lanes same slot
// (changes per iteration), 1 = lanes in groups of 8 share a slot, 2 = random per lane
template <int W>
__global__ __launch_bounds__(256) void k_gather(const uint4* tab, uint32_t span_mask, int pattern, int iters, uint32_t* out) {
    uint32_t s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it += 4) {
        uint32_t idx[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s = s * 1664525u + 1013904223u;
            uint32_t r = s >> 8;
            if (pattern == 0) r = __builtin_amdgcn_readfirstlane(r);
            else if (pattern == 1) r = __shfl(r, lane & ~7u);
            idx[k] = r & span_mask;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (W == 4) { const uint4 v = tab[idx[k]]; acc += v.x ^ v.y ^ v.z ^ v.w; }
            else if (W == 2) { const uint2 v = *reinterpret_cast<const uint2*>(tab + idx[k]); acc += v.x ^ v.y; }
            else { acc += *reinterpret_cast<const uint32_t*>(tab + idx[k]); }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main() {
    const size_t slots = 1u << 22;   // 64 MB
    uint4* tab; CK(hipMalloc(&tab, slots * 16)); CK(hipMemset(tab, 1, slots * 16));
    uint32_t* out; CK(hipMalloc(&out, 2048 * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 512, nb = 2048;
    const uint32_t spans[] = { 255u, 2047u, 32767u, 131071u, (1u << 22) - 1 };   // 4 KB, 32 KB, 512 KB, 2 MB, 64 MB
    const char* sn[] = { "4 KB", "32 KB", "512 KB", "2 MB", "64 MB" };
    for (int W : { 4, 2, 1 })
        for (int pattern = 0; pattern < 3; ++pattern)
            for (int si = 0; si < 5; ++si) {
                for (int rep = 0; rep < 2; ++rep) {
                    CK(hipEventRecord(e0));
                    if (W == 4) hipLaunchKernelGGL(k_gather<4>, dim3(nb), dim3(256), 0, 0, tab, spans[si], pattern, iters, out);
                    else if (W == 2) hipLaunchKernelGGL(k_gather<2>, dim3(nb), dim3(256), 0, 0, tab, spans[si], pattern, iters, out);
                    else hipLaunchKernelGGL(k_gather<1>, dim3(nb), dim3(256), 0, 0, tab, spans[si], pattern, iters, out);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    // wave-gathers per CU = nb * 4 waves * iters / 256 CUs
                    const double per = ms * 1e-3 * 2.4e9 / ((double)nb * 4 * iters / 256);
                    if (rep) printf("dwordx%d pattern %s span %-6s : %6.1f cycles (2.4 GHz) per wave-gather per CU\n", W, pattern == 0 ? "uniform " : (pattern == 1 ? "groups-8" : "random  "), sn[si], per);
                }
            }
    return 0;
}
