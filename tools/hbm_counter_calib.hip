// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on MI355X against KNOWN byte counts: streaming kernels that read
// (write) W = 4, 8, 12, 16 bytes per lane, coalesced, over a 1 GiB buffer (4 x the Infinity Cache) and over a 64 MiB buffer
// (cache resident across launches), plus a 16-byte-per-lane gather.  Prints one JSON line per kernel with the bytes ONE
// launch moves; tools/hbm_counter_calib.py joins it with the counter CSVs of the --pmc passes (tools/calib.sh) into
// profiles/r03_hbm_counter_calibration.json, whose factors tools/pmc_traffic.py applies.
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_counter_calib.hip -o tools/hbm_counter_calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int W> struct Vec;
template <> struct Vec<4> { uint32_t a; };
template <> struct Vec<8> { uint32_t a, b; };
template <> struct __attribute__((packed, aligned(4))) Vec<12> { uint32_t a, b, c; };
template <> struct __attribute__((aligned(16))) Vec<16> { uint32_t a, b, c, d; };

#define KREAD(W, NAME)                                                                                         \
    __global__ __launch_bounds__(256) void NAME(const Vec<W>* __restrict__ src, size_t n, uint32_t* out) {        \
        uint32_t acc = 0;                                                                                     \
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {         \
            const Vec<W> v = src[i];                                                                          \
            acc += v.a;                                                                                       \
            if (W >= 8) acc += reinterpret_cast<const uint32_t*>(&v)[W / 4 - 1];                                \
        }                                                                                                     \
        if (acc == 0x12345678u) out[0] = acc;                                                                  \
    }
#define KWRITE(W, NAME)                                                                                        \
    __global__ __launch_bounds__(256) void NAME(Vec<W>* __restrict__ dst, size_t n, uint32_t seed) {             \
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {         \
            Vec<W> v;                                                                                         \
            uint32_t* p = reinterpret_cast<uint32_t*>(&v);                                                     \
            for (int k = 0; k < W / 4; ++k) p[k] = seed + (uint32_t)i + k;                                     \
            dst[i] = v;                                                                                       \
        }                                                                                                     \
    }
KREAD(4, calib_read4) KREAD(8, calib_read8) KREAD(12, calib_read12) KREAD(16, calib_read16)
KREAD(4, calib_read4_cached) KREAD(12, calib_read12_cached) KREAD(16, calib_read16_cached)
KWRITE(4, calib_write4) KWRITE(8, calib_write8) KWRITE(12, calib_write12) KWRITE(16, calib_write16)
KWRITE(16, calib_write16_cached)
// 2-byte and mixed accesses of the integer Lab planes: 8 bytes (4 x u16) + 16 bytes (4 x u32) per lane
__global__ __launch_bounds__(256) void calib_read_planes(const uint2* __restrict__ a, const uint4* __restrict__ b, size_t n, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const uint2 x = a[i]; const uint4 y = b[i]; acc += x.x + x.y + y.x + y.w; }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const size_t big = (size_t)1 << 30, small = (size_t)64 << 20;
    void* buf; uint32_t* out;
    CK(hipMalloc(&buf, big + 4096)); CK(hipMalloc(&out, 4096)); CK(hipMemset(buf, 1, big));
    const int grid = 256 * 16, reps = 6;
#define RUNR(W, NAME, BYTES) for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(NAME, dim3(grid), dim3(256), 0, 0, (const Vec<W>*)buf, (size_t)(BYTES) / W, out); \
    printf("{\"kernel\": \"%s\", \"read_bytes\": %zu, \"write_bytes\": 0, \"bytes_per_lane\": %d, \"footprint\": %zu}\n", #NAME, ((size_t)(BYTES) / W) * W, W, (size_t)(BYTES));
#define RUNW(W, NAME, BYTES) for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(NAME, dim3(grid), dim3(256), 0, 0, (Vec<W>*)buf, (size_t)(BYTES) / W, (uint32_t)r); \
    printf("{\"kernel\": \"%s\", \"read_bytes\": 0, \"write_bytes\": %zu, \"bytes_per_lane\": %d, \"footprint\": %zu}\n", #NAME, ((size_t)(BYTES) / W) * W, W, (size_t)(BYTES));
    RUNR(4, calib_read4, big) RUNR(8, calib_read8, big) RUNR(12, calib_read12, big) RUNR(16, calib_read16, big)
    RUNR(4, calib_read4_cached, small) RUNR(12, calib_read12_cached, small) RUNR(16, calib_read16_cached, small)
    RUNW(4, calib_write4, big) RUNW(8, calib_write8, big) RUNW(12, calib_write12, big) RUNW(16, calib_write16, big)
    RUNW(16, calib_write16_cached, small)
    {
        const size_t n = big / 32;      // 8 + 16 bytes per element: planes of n x 4 pixels
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(calib_read_planes, dim3(grid), dim3(256), 0, 0, (const uint2*)buf, (const uint4*)((char*)buf + n * 8), n, out);
        printf("{\"kernel\": \"calib_read_planes\", \"read_bytes\": %zu, \"write_bytes\": 0, \"bytes_per_lane\": 24, \"footprint\": %zu}\n", n * 24, n * 24);
    }
    CK(hipDeviceSynchronize());
    return 0;
}
