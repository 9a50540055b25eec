// HBM streaming ceiling of ONE MI355X measured with hand-written kernels (round 5; the round-2 "ceiling" was torch's copy kernel).
// Sweeps, on 1 GiB footprints (4 x the Infinity Cache):
//   op      read / write / copy / triad (2 in, 1 out) / planes (8 + 16 B in, 12 B out: the byte shape of k_lap_final_v4)
//   W       4, 8, 12, 16 bytes per lane and access (dword .. dwordx4; 12 = the BGR group of four pixels)
//   U       1, 2, 4, 8 accesses in flight per lane (all loads of a chunk are issued before the first use)
//   policy  plain / nontemporal loads and stores
//   k       1, 2, 4, 8 workgroups of 256 threads per CU (= waves per SIMD), enforced with dynamic LDS
//   grid    persistent (256 k workgroups): chunks round-robin over workgroups ("stride"), one contiguous range per workgroup
//           ("contig"), one contiguous eighth per XCD with its workgroups round-robin inside ("xcd");
//           one-shot (a workgroup per chunk): launch order ("oneshot") or XCD-contiguous ("oneshot_xcd")
// Prints one line per configuration (GB/s of read + written bytes, HIP events over REPS launches) and the best of every op.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_stream.hip -o tools/ubench_stream
//   tools/ubench_stream                 full sweep
//   tools/ubench_stream only OP W U NT K MODE [reps]   one configuration (for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int W> struct V;
template <> struct V<4> { uint32_t d[1]; };
template <> struct __attribute__((aligned(8))) V<8> { uint32_t d[2]; };
template <> struct __attribute__((packed, aligned(4))) V<12> { uint32_t d[3]; };
template <> struct __attribute__((aligned(16))) V<16> { uint32_t d[4]; };

template <int W, int NT> __device__ __forceinline__ V<W> ld(const V<W>* p) {
    V<W> v;
    if (NT) {
        if constexpr (W == 4) v.d[0] = __builtin_nontemporal_load(&p->d[0]);
        else if constexpr (W == 8) { const auto t = __builtin_nontemporal_load(reinterpret_cast<const __attribute__((ext_vector_type(2))) uint32_t*>(p)); v.d[0] = t.x; v.d[1] = t.y; }
        else if constexpr (W == 12) { const auto t = __builtin_nontemporal_load(reinterpret_cast<const __attribute__((ext_vector_type(3), aligned(4))) uint32_t*>(p)); v.d[0] = t.x; v.d[1] = t.y; v.d[2] = t.z; }
        else { const auto t = __builtin_nontemporal_load(reinterpret_cast<const __attribute__((ext_vector_type(4))) uint32_t*>(p)); v.d[0] = t.x; v.d[1] = t.y; v.d[2] = t.z; v.d[3] = t.w; }
    } else v = *p;
    return v;
}
template <int W, int NT> __device__ __forceinline__ void st(V<W>* p, const V<W>& v) {
    if (NT) {
        if constexpr (W == 4) __builtin_nontemporal_store(v.d[0], &p->d[0]);
        else if constexpr (W == 8) { __attribute__((ext_vector_type(2))) uint32_t t = {v.d[0], v.d[1]}; __builtin_nontemporal_store(t, reinterpret_cast<__attribute__((ext_vector_type(2))) uint32_t*>(p)); }
        else if constexpr (W == 12) { typedef __attribute__((ext_vector_type(3), aligned(4))) uint32_t T3; T3 t = {v.d[0], v.d[1], v.d[2]}; __builtin_nontemporal_store(t, reinterpret_cast<T3*>(p)); }
        else { __attribute__((ext_vector_type(4))) uint32_t t = {v.d[0], v.d[1], v.d[2], v.d[3]}; __builtin_nontemporal_store(t, reinterpret_cast<__attribute__((ext_vector_type(4))) uint32_t*>(p)); }
    } else *p = v;
}

enum { OP_READ = 0, OP_WRITE = 1, OP_COPY = 2, OP_TRIAD = 3 };
enum { M_STRIDE = 0, M_CONTIG = 1, M_XCD = 2, M_ONESHOT = 3, M_ONESHOT_XCD = 4 };
static const char* kOp[] = {"read", "write", "copy", "triad", "planes"};
static const char* kMode[] = {"stride", "contig", "xcd", "oneshot", "oneshot_xcd"};

// which chunks workgroup b of G visits: first, step, end (exclusive)
__device__ __forceinline__ void chunk_range(int mode, size_t nchunks, size_t& c, size_t& step, size_t& end) {
    const size_t b = blockIdx.x, G = gridDim.x;
    if (mode == M_STRIDE || mode == M_ONESHOT) { c = b; step = G; end = nchunks; }
    else if (mode == M_CONTIG) { c = b * nchunks / G; step = 1; end = (b + 1) * nchunks / G; }
    else if (mode == M_XCD) { const size_t x = b & 7, j = b >> 3, per = nchunks / 8; c = x * per + j; step = G / 8; end = (x + 1) * per; }
    else { const size_t x = b & 7, j = b >> 3, per = nchunks / 8; c = x * per + j; step = nchunks; end = j < per ? nchunks : 0; }
}

template <int OP, int W, int U, int NT>
__global__ __launch_bounds__(256) void k_stream(const V<W>* __restrict__ a, const V<W>* __restrict__ b, V<W>* __restrict__ c, size_t nchunks, int mode, uint32_t* sink) {
    extern __shared__ uint32_t lds_pad[];
    size_t ch, step, end;
    chunk_range(mode, nchunks, ch, step, end);
    uint32_t acc = 0;
    for (; ch < end; ch += step) {
        const size_t base = ch * (size_t)(256 * U) + threadIdx.x;
        V<W> x[U], y[U];
        if (OP != OP_WRITE) {
#pragma unroll
            for (int j = 0; j < U; ++j) x[j] = ld<W, NT>(a + base + j * 256);
        }
        if (OP == OP_TRIAD) {
#pragma unroll
            for (int j = 0; j < U; ++j) y[j] = ld<W, NT>(b + base + j * 256);
        }
#pragma unroll
        for (int j = 0; j < U; ++j) {
            if (OP == OP_READ) { for (int q = 0; q < W / 4; ++q) acc += x[j].d[q]; }
            else {
                V<W> v;
                for (int q = 0; q < W / 4; ++q) v.d[q] = OP == OP_WRITE ? (uint32_t)base + q : OP == OP_COPY ? x[j].d[q] : x[j].d[q] + y[j].d[q];
                st<W, NT>(c + base + j * 256, v);
            }
        }
    }
    if (OP == OP_READ && acc == 0x12345678u) sink[0] = acc + lds_pad[0];
}

// the byte shape of the last Laplace kernel: per 4-pixel group 8 B (uint16 x 4) + 16 B (dword x 4) in, 12 B out
template <int U, int NT>
__global__ __launch_bounds__(256) void k_planes(const V<8>* __restrict__ a, const V<16>* __restrict__ b, V<12>* __restrict__ c, size_t nchunks, int mode, uint32_t* sink) {
    extern __shared__ uint32_t lds_pad[];
    size_t ch, step, end;
    chunk_range(mode, nchunks, ch, step, end);
    for (; ch < end; ch += step) {
        const size_t base = ch * (size_t)(256 * U) + threadIdx.x;
        V<8> x[U]; V<16> y[U];
#pragma unroll
        for (int j = 0; j < U; ++j) x[j] = ld<8, NT>(a + base + j * 256);
#pragma unroll
        for (int j = 0; j < U; ++j) y[j] = ld<16, NT>(b + base + j * 256);
#pragma unroll
        for (int j = 0; j < U; ++j) {
            V<12> v;
            v.d[0] = x[j].d[0] + y[j].d[0]; v.d[1] = x[j].d[1] + y[j].d[1]; v.d[2] = y[j].d[2] ^ y[j].d[3];
            st<12, NT>(c + base + j * 256, v);
        }
    }
}

struct Cfg { int op, W, U, nt, k, mode; double gbs, us; int occ; };
static void *g_a, *g_b, *g_c; static uint32_t* g_sink;
static size_t kFoot = (size_t)1 << 30;      // UBENCH_FOOT_MB overrides (cache-resident footprints)
static int g_cus = 256;

template <int OP, int W, int U, int NT>
static void run_one(Cfg& c, int reps) {
    const size_t nel = kFoot / W, nchunks = (nel / (256 * U)) / 8 * 8;
    const bool oneshot = c.mode >= M_ONESHOT;
    // dynamic LDS so that at most k workgroups fit a CU (160 KiB)
    const size_t lds = c.k >= 8 ? 0 : (size_t)(160 * 1024 / c.k) - (c.k == 1 ? 0 : 1024);
    auto kern = k_stream<OP, W, U, NT>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, lds)); c.occ = occ;
    const size_t grid = oneshot ? nchunks : (size_t)g_cus * std::min(c.k, occ);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, (const V<W>*)g_a, (const V<W>*)g_b, (V<W>*)g_c, nchunks, c.mode, g_sink);
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, (const V<W>*)g_a, (const V<W>*)g_b, (V<W>*)g_c, nchunks, c.mode, g_sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)nchunks * 256 * U * W * (OP == OP_READ || OP == OP_WRITE ? 1 : OP == OP_COPY ? 2 : 3);
    c.us = ms * 1e3 / reps; c.gbs = bytes / (c.us * 1e-6) / 1e9;
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}
template <int U, int NT>
static void run_planes(Cfg& c, int reps) {
    const size_t nel = kFoot / 16, nchunks = (nel / (256 * U)) / 8 * 8;
    const bool oneshot = c.mode >= M_ONESHOT;
    const size_t lds = c.k >= 8 ? 0 : (size_t)(160 * 1024 / c.k) - (c.k == 1 ? 0 : 1024);
    auto kern = k_planes<U, NT>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, lds)); c.occ = occ;
    const size_t grid = oneshot ? nchunks : (size_t)g_cus * std::min(c.k, occ);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, (const V<8>*)g_a, (const V<16>*)g_b, (V<12>*)g_c, nchunks, c.mode, g_sink);
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, (const V<8>*)g_a, (const V<16>*)g_b, (V<12>*)g_c, nchunks, c.mode, g_sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)nchunks * 256 * U * 36;
    c.us = ms * 1e3 / reps; c.gbs = bytes / (c.us * 1e-6) / 1e9;
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

template <int OP, int W, int U> static void disp_nt(Cfg& c, int reps) { if (c.nt) run_one<OP, W, U, 1>(c, reps); else run_one<OP, W, U, 0>(c, reps); }
template <int OP, int W> static void disp_u(Cfg& c, int reps) {
    switch (c.U) { case 1: disp_nt<OP, W, 1>(c, reps); break; case 2: disp_nt<OP, W, 2>(c, reps); break; case 4: disp_nt<OP, W, 4>(c, reps); break; default: disp_nt<OP, W, 8>(c, reps); }
}
template <int OP> static void disp_w(Cfg& c, int reps) {
    switch (c.W) { case 4: disp_u<OP, 4>(c, reps); break; case 8: disp_u<OP, 8>(c, reps); break; case 12: disp_u<OP, 12>(c, reps); break; default: disp_u<OP, 16>(c, reps); }
}
static void dispatch(Cfg& c, int reps) {
    if (c.op == 4) {
        switch (c.U) {
            case 1: c.nt ? run_planes<1, 1>(c, reps) : run_planes<1, 0>(c, reps); break;
            case 2: c.nt ? run_planes<2, 1>(c, reps) : run_planes<2, 0>(c, reps); break;
            case 4: c.nt ? run_planes<4, 1>(c, reps) : run_planes<4, 0>(c, reps); break;
            default: c.nt ? run_planes<8, 1>(c, reps) : run_planes<8, 0>(c, reps);
        }
        return;
    }
    switch (c.op) { case 0: disp_w<0>(c, reps); break; case 1: disp_w<1>(c, reps); break; case 2: disp_w<2>(c, reps); break; default: disp_w<3>(c, reps); }
}
static void print(const Cfg& c) {
    printf("%-6s W=%2d U=%d %-5s k=%d(occ %d) %-11s %8.1f us  %7.1f GB/s\n", kOp[c.op], c.W, c.U, c.nt ? "nt" : "plain", c.k, c.occ, kMode[c.mode], c.us, c.gbs);
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0)); g_cus = prop.multiProcessorCount;
    if (const char* e = getenv("UBENCH_FOOT_MB")) kFoot = (size_t)atol(e) << 20;
    CK(hipMalloc(&g_a, kFoot + 4096)); CK(hipMalloc(&g_b, kFoot + 4096)); CK(hipMalloc(&g_c, kFoot + 4096)); CK(hipMalloc((void**)&g_sink, 4096));
    CK(hipMemset(g_a, 1, kFoot)); CK(hipMemset(g_b, 2, kFoot)); CK(hipMemset(g_c, 3, kFoot));
    if (argc >= 8 && !strcmp(argv[1], "only")) {
        Cfg c{atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), atoi(argv[7]), 0, 0, 0};
        dispatch(c, argc > 8 ? atoi(argv[8]) : 5); print(c);
        return 0;
    }
    printf("# %s, %d CUs; footprint %zu MiB per buffer; GB/s = (read + written bytes) / HIP-event time of 5 launches\n", prop.name, g_cus, kFoot >> 20);
    // clock ramp
    { Cfg c{2, 16, 4, 0, 4, 0, 0, 0, 0}; dispatch(c, 60); }
    std::vector<Cfg> all;
    const int Ws[] = {4, 8, 12, 16}, Us[] = {1, 2, 4, 8}, Ks[] = {1, 2, 4, 8};
    for (int op = 0; op <= 4; ++op)
        for (int W : Ws) {
            if (op == 4 && W != 16) continue;
            for (int U : Us) for (int nt = 0; nt < 2; ++nt) {
                for (int k : Ks) for (int mode = 0; mode <= 2; ++mode) { Cfg c{op, W, U, nt, k, mode, 0, 0, 0}; dispatch(c, 5); print(c); all.push_back(c); }
                for (int mode = 3; mode <= 4; ++mode) { Cfg c{op, W, U, nt, 8, mode, 0, 0, 0}; dispatch(c, 5); print(c); all.push_back(c); }
            }
        }
    printf("\n# best per op and width\n");
    for (int op = 0; op <= 4; ++op) for (int W : Ws) {
        const Cfg* b = nullptr;
        for (const Cfg& c : all) if (c.op == op && c.W == W && (!b || c.gbs > b->gbs)) b = &c;
        if (b) print(*b);
    }
    printf("\n# best per op, width and grid shape (plain | nt)\n");
    for (int op = 0; op <= 4; ++op) for (int W : Ws) for (int mode = 0; mode <= 4; ++mode) for (int nt = 0; nt < 2; ++nt) {
        const Cfg* b = nullptr;
        for (const Cfg& c : all) if (c.op == op && c.W == W && c.mode == mode && c.nt == nt && (!b || c.gbs > b->gbs)) b = &c;
        if (b) print(*b);
    }
    return 0;
}
