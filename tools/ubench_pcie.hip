// PCIe between page-locked host memory and HBM on one MI355X box (round 5, for the host -> host surfaces lvm_process / lvm_export_frames):
// DMA copies against kernels that read / write the page-locked buffer directly (zero-copy), one direction and both at once.
// Frame-sized transfers (1080p BGR = 6.2 MB) and 32-frame batches.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_pcie.hip -o tools/ubench_pcie
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
struct __attribute__((packed, aligned(4))) B12 { uint32_t a, b, c; };
__global__ __launch_bounds__(256) void k_copy12(const B12* __restrict__ src, B12* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const size_t frame = 1920 * 1080 * 3, batch = 32 * frame;
    void *h_in, *h_out, *d_a, *d_b;
    CK(hipHostMalloc(&h_in, batch, 0)); CK(hipHostMalloc(&h_out, batch, 0));
    CK(hipMalloc(&d_a, batch)); CK(hipMalloc(&d_b, batch));
    for (size_t i = 0; i < batch; i += 4096) ((char*)h_in)[i] = 1, ((char*)h_out)[i] = 2;
    CK(hipMemset(d_a, 1, batch)); CK(hipMemset(d_b, 2, batch));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    auto rate = [&](const char* name, size_t bytes_moved, int reps, auto fn) {
        for (int r = 0; r < 3; ++r) fn();
        CK(hipDeviceSynchronize());
        const double t0 = now();
        for (int r = 0; r < reps; ++r) fn();
        CK(hipDeviceSynchronize());
        const double dt = (now() - t0) / reps;
        printf("%-78s %8.1f us  %6.1f GB/s\n", name, dt * 1e6, bytes_moved / dt / 1e9);
    };
    for (size_t sz : {frame, batch}) {
        const int reps = sz == frame ? 200 : 20;
        printf("# %zu bytes per transfer (%s)\n", sz, sz == frame ? "one 1080p BGR frame" : "32 frames");
        rate("DMA host -> device (hipMemcpyAsync), back to back", sz, reps, [&] { CK(hipMemcpyAsync(d_a, h_in, sz, hipMemcpyHostToDevice, s1)); });
        rate("DMA device -> host", sz, reps, [&] { CK(hipMemcpyAsync(h_out, d_b, sz, hipMemcpyDeviceToHost, s1)); });
        rate("DMA both directions at once (two streams), bytes of both", 2 * sz, reps, [&] {
            CK(hipMemcpyAsync(d_a, h_in, sz, hipMemcpyHostToDevice, s1)); CK(hipMemcpyAsync(h_out, d_b, sz, hipMemcpyDeviceToHost, s2)); });
        rate("DMA host -> device, synchronised after every copy (one frame in flight)", sz, reps, [&] {
            CK(hipMemcpyAsync(d_a, h_in, sz, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); });
        for (int grid : {256, 1024, 4096}) {
            char nm[128];
            snprintf(nm, sizeof nm, "kernel reads the page-locked buffer (12 B per lane), grid %d", grid);
            rate(nm, sz, reps, [&] { hipLaunchKernelGGL(k_copy12, dim3(grid), dim3(256), 0, s1, (const B12*)h_in, (B12*)d_a, sz / 12); });
            snprintf(nm, sizeof nm, "kernel reads the page-locked buffer (16 B per lane), grid %d", grid);
            rate(nm, sz, reps, [&] { hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(256), 0, s1, (const uint4*)h_in, (uint4*)d_a, sz / 16); });
            snprintf(nm, sizeof nm, "kernel writes the page-locked buffer (12 B per lane), grid %d", grid);
            rate(nm, sz, reps, [&] { hipLaunchKernelGGL(k_copy12, dim3(grid), dim3(256), 0, s1, (const B12*)d_b, (B12*)h_out, sz / 12); });
        }
        rate("kernel reads + another kernel writes page-locked buffers at once, bytes of both", 2 * sz, reps, [&] {
            hipLaunchKernelGGL(k_copy12, dim3(512), dim3(256), 0, s1, (const B12*)h_in, (B12*)d_a, sz / 12);
            hipLaunchKernelGGL(k_copy12, dim3(512), dim3(256), 0, s2, (const B12*)d_b, (B12*)h_out, sz / 12); });
        rate("DMA up + kernel writes page-locked buffer at once, bytes of both", 2 * sz, reps, [&] {
            CK(hipMemcpyAsync(d_a, h_in, sz, hipMemcpyHostToDevice, s1));
            hipLaunchKernelGGL(k_copy12, dim3(512), dim3(256), 0, s2, (const B12*)d_b, (B12*)h_out, sz / 12); });
    }
    // chunked: one frame as 4 / 8 row chunks, upload chunk k+1 under "kernel" k (a device copy standing in), download behind
    for (int nch : {1, 2, 4, 8}) {
        const size_t cs = frame / nch / 12 * 12;
        hipEvent_t ev[8]; for (int i = 0; i < 8; ++i) CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        char nm[128]; snprintf(nm, sizeof nm, "one frame up in %d chunk(s), device pass per chunk on a second stream, synchronised", nch);
        rate(nm, frame, 200, [&] {
            for (int k = 0; k < nch; ++k) {
                CK(hipMemcpyAsync((char*)d_a + k * cs, (char*)h_in + k * cs, cs, hipMemcpyHostToDevice, s1));
                CK(hipEventRecord(ev[k], s1)); CK(hipStreamWaitEvent(s2, ev[k], 0));
                hipLaunchKernelGGL(k_copy12, dim3(256), dim3(256), 0, s2, (const B12*)((char*)d_a + k * cs), (B12*)((char*)d_b + k * cs), cs / 12);
            }
            CK(hipStreamSynchronize(s2)); });
    }
    return 0;
}
