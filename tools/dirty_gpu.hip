// Fills the LDS of every CU and a large part of the free HBM with signalling garbage (NaN bit patterns), then exits.  Device memory is not
// cleared between kernels, and LDS is never cleared: a kernel that reads a location it did not write gets whatever ran before.  Run before
// a test to turn such a read into a visible failure:  tools/dirty_gpu [GiB] && python -m pytest tests -m gpu ...
// Build: hipcc --offload-arch=gfx950 -O2 tools/dirty_gpu.hip -o tools/dirty_gpu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ __launch_bounds__(1024) void k_dirty_lds(unsigned* sink) {
    extern __shared__ unsigned lds[];
    const int n = 160 * 1024 / 4;
    for (int i = threadIdx.x; i < n; i += 1024) lds[i] = 0x7fc0dead;
    __syncthreads();
    if (lds[(threadIdx.x * 37) % n] == 1u) sink[0] = 1;    // keep the stores alive
    // stay resident for a while so that the workgroups spread over all CUs
    for (int k = 0; k < 2000; ++k) __builtin_amdgcn_s_sleep(64);
}
__global__ void k_fill(unsigned* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0xffc0beefu;
}
int main(int argc, char** argv) {
    const size_t gib = argc > 1 ? (size_t)std::atol(argv[1]) : 32;
    unsigned* sink; (void)hipMalloc(&sink, 4);
    (void)hipFuncSetAttribute((const void*)k_dirty_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 4; ++rep) k_dirty_lds<<<1024, 1024, 160 * 1024>>>(sink);
    hipError_t e = hipDeviceSynchronize();
    std::printf("lds: %s\n", hipGetErrorString(e));
    std::vector<unsigned*> blocks;
    for (size_t g = 0; g < gib; ++g) {
        unsigned* p;
        if (hipMalloc(&p, 1ull << 30) != hipSuccess) break;
        k_fill<<<2048, 256>>>(p, (1ull << 30) / 4);
        blocks.push_back(p);
    }
    e = hipDeviceSynchronize();
    std::printf("hbm: %zu GiB filled, %s\n", blocks.size(), hipGetErrorString(e));
    for (unsigned* p : blocks) (void)hipFree(p);
    return 0;
}
