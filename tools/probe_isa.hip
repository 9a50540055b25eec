// Probe of two gfx950 instruction semantics the kernels would like to rely on (run on the GPU box):
//   v_cvt_pk_u8_f32  -- rounding mode and saturation
//   DPP wave_shr:1 / wave_shl:1 -- cross-lane shifts over the whole 64-lane wave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k_cvt(const float* in, unsigned* out, int n) {
    const int i = threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 1, 0xAABBCCDDu);
}
__global__ void k_dpp(int* out) {
    const int i = threadIdx.x;
    const int v = 1000 + i;
    out[i] = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false);        // wave_shr:1
    out[64 + i] = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xf, 0xf, false);   // wave_shl:1
    out[128 + i] = __builtin_amdgcn_update_dpp(-1, v, 0x111, 0xf, 0xf, false);  // row_shr:1
}
int main() {
    const float h[] = {0.f, 0.4f, 0.5f, 0.6f, 1.5f, 2.5f, 3.5f, 254.4f, 254.5f, 254.6f, 255.4f, 255.5f, 256.f, 300.f, 1e9f, -0.4f, -0.6f, -5.f, NAN, INFINITY};
    const int n = sizeof(h) / sizeof(h[0]);
    float* d; unsigned* o; int* q;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, n * 4); hipMalloc(&q, 192 * 4);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_cvt, dim3(1), dim3(64), 0, 0, d, o, n);
    hipLaunchKernelGGL(k_dpp, dim3(1), dim3(64), 0, 0, q);
    unsigned ho[32]; int hq[192];
    hipMemcpy(ho, o, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hq, q, sizeof(hq), hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("cvt_pk_u8(%g) -> %08x (byte1 = %u)\n", h[i], ho[i], (ho[i] >> 8) & 255);
    printf("wave_shr:1 lanes 0,1,15,16,17,31,32,33,63: %d %d %d %d %d %d %d %d %d\n", hq[0], hq[1], hq[15], hq[16], hq[17], hq[31], hq[32], hq[33], hq[63]);
    printf("wave_shl:1 lanes 0,1,15,16,31,32,62,63: %d %d %d %d %d %d %d %d\n", hq[64], hq[65], hq[79], hq[80], hq[95], hq[96], hq[126], hq[127]);
    printf("row_shr:1  lanes 0,1,15,16,17: %d %d %d %d %d\n", hq[128], hq[129], hq[143], hq[144], hq[145]);
    return 0;
}
