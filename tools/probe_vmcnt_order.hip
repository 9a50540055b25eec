// Does s_waitcnt vmcnt(N) on gfx950 retire vector-memory operations IN ORDER when loads and stores are mixed?
// Every wave issues a load that misses every cache (a line of a 4 GiB buffer nobody touched since it was written), then a store to
// a line that is hot in L2, then s_waitcnt vmcnt(1) -- "at most the store is still outstanding" if the counter decrements in issue
// order -- and copies the load's destination register.  A copy that still holds the sentinel is a load that had NOT returned when
// vmcnt(1) let the wave through: the store was retired before the older load.
// Build/run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/probe_vmcnt_order.hip -o tools/probe_vmcnt_order && tools/probe_vmcnt_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef int rsrc_t __attribute__((ext_vector_type(4)));

__global__ void k_fill(uint32_t* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (uint32_t)(i * 2654435761u) | 1u;
}

template <int MODE>   // 0: global load + global store, 1: buffer load + buffer store, 2: buffer load + buffer store whose lanes are ALL out of range
__global__ void k_probe(const uint32_t* cold, size_t stride_dw, uint32_t* hot, uint32_t* bad, int rounds, uint32_t bytes_cold, uint32_t bytes_hot) {
    const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    uint32_t nbad = 0;
    for (int r = 0; r < rounds; ++r) {
        const size_t idx = ((wave * (size_t)rounds + r) * stride_dw + lane) ;
        const uint32_t want = (uint32_t)(idx * 2654435761u) | 1u;
        uint32_t ld = 0, early = 0;
        uint32_t* hp = hot + (threadIdx.x + blockIdx.x * blockDim.x) % 4096;
        if (MODE == 0) {
            const uint32_t* cp = cold + idx;
            asm volatile("v_mov_b32 %0, 0\n\t"
                         "global_load_dword %0, %2, off\n\t"
                         "global_store_dword %3, %4, off\n\t"
                         "s_waitcnt vmcnt(1)\n\t"
                         "v_mov_b32 %1, %0\n\t"
                         "s_waitcnt vmcnt(0)"
                         : "=&v"(ld), "=&v"(early) : "v"(cp), "v"(hp), "v"(want) : "memory");
        } else {
            rsrc_t rc, rh;
            const uint64_t bc = (uint64_t)cold, bh = (uint64_t)hot;
            rc.x = (int)bc; rc.y = (int)(bc >> 32) & 0xffff; rc.z = (int)0xffffffffu; rc.w = 0x00020000;
            rh.x = (int)bh; rh.y = (int)(bh >> 32) & 0xffff; rh.z = (int)bytes_hot; rh.w = 0x00020000;
            // cold buffer: 64-bit base + 32-bit offset: fold the high part of the offset into the base
            const uint64_t off = idx * 4ull, hi = off & ~0x3fffffffull;
            const uint64_t b2 = bc + hi;
            rc.x = (int)b2; rc.y = (int)(b2 >> 32) & 0xffff;
            const uint32_t voc = (uint32_t)(off - hi);
            const uint32_t voh = (MODE == 2) ? 0x80000000u : (uint32_t)(((threadIdx.x + blockIdx.x * blockDim.x) % 4096) * 4);
            rsrc_t rcs, rhs;
            rcs.x = __builtin_amdgcn_readfirstlane(rc.x); rcs.y = __builtin_amdgcn_readfirstlane(rc.y); rcs.z = rc.z; rcs.w = rc.w;
            rhs.x = __builtin_amdgcn_readfirstlane(rh.x); rhs.y = __builtin_amdgcn_readfirstlane(rh.y); rhs.z = __builtin_amdgcn_readfirstlane(rh.z); rhs.w = rh.w;
            asm volatile("v_mov_b32 %0, 0\n\t"
                         "buffer_load_dword %0, %2, %5, 0 offen\n\t"
                         "buffer_store_dword %4, %3, %6, 0 offen\n\t"
                         "s_waitcnt vmcnt(1)\n\t"
                         "v_mov_b32 %1, %0\n\t"
                         "s_waitcnt vmcnt(0)"
                         : "=&v"(ld), "=&v"(early) : "v"(voc), "v"(voh), "v"(want), "s"(rcs), "s"(rhs) : "memory");
        }
        if (ld != want) nbad += 1u << 16;        // the load itself (after vmcnt(0)): must never happen
        if (early != want) nbad += 1;
    }
    if (nbad & 0xffffu) atomicAdd(bad, nbad & 0xffffu);
    if (nbad >> 16) atomicAdd(bad + 1, nbad >> 16);
}

int main() {
    const size_t cold_bytes = 6ull << 30;
    uint32_t *cold, *hot, *bad;
    if (hipMalloc(&cold, cold_bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    (void)hipMalloc(&hot, 4096 * 4); (void)hipMalloc(&bad, 8);
    k_fill<<<4096, 256>>>(cold, cold_bytes / 4);
    (void)hipDeviceSynchronize();
    const int waves = 256 * 4 * 8, rounds = 16;
    const size_t stride_dw = (cold_bytes / 4) / ((size_t)waves * rounds) & ~(size_t)63;   // every (wave, round) its own far-apart line
    printf("stride between probes: %zu KiB\n", stride_dw * 4 / 1024);
    const char* names[3] = {"global load, global store (hot)", "buffer load, buffer store (hot)", "buffer load, buffer store with every lane out of range"};
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            // evict: stream over the cold buffer again so nothing of it sits in L2 / Infinity Cache in probe order
            k_fill<<<4096, 256>>>(cold, cold_bytes / 4);
            (void)hipMemset(bad, 0, 8);
            (void)hipDeviceSynchronize();
            if (mode == 0) k_probe<0><<<waves / 4, 256>>>(cold, stride_dw, hot, bad, rounds, 0, 4096 * 4);
            if (mode == 1) k_probe<1><<<waves / 4, 256>>>(cold, stride_dw, hot, bad, rounds, 0, 4096 * 4);
            if (mode == 2) k_probe<2><<<waves / 4, 256>>>(cold, stride_dw, hot, bad, rounds, 0, 4096 * 4);
            hipError_t e = hipDeviceSynchronize();
            uint32_t hb[2];
            (void)hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost);
            printf("%-62s rep %d: %u of %zu lane-probes saw the register before the load returned (after vmcnt(0): %u wrong) %s\n", names[mode], rep, hb[0],
                   (size_t)waves * 64 * rounds, hb[1], e == hipSuccess ? "" : hipGetErrorString(e));
        }
    }
    return 0;
}
