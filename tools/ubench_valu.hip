// Issue-rate microbenchmark of the gfx950 vector instructions the colour kernels are built from (run on the GPU
// box: `hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench && /tmp/ubench`).  Every kernel runs
// 16 independent dependency chains of ONE instruction per lane, 8 waves per SIMD on every CU, and reports the time
// per wave-instruction per SIMD relative to v_fma_f32.  Numbers feed the instruction budget in DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define CHAINS 16
#define ITERS 2048

#define KERNEL32(NAME, ASM)                                                                         \
    __global__ __launch_bounds__(256) void NAME(float* out, float a, float b) {                    \
        float r[CHAINS];                                                                            \
        _Pragma("unroll") for (int k = 0; k < CHAINS; ++k) r[k] = a * (threadIdx.x + k) + b;        \
        for (int it = 0; it < ITERS; ++it) {                                                        \
            _Pragma("unroll") for (int k = 0; k < CHAINS; ++k)                                      \
                asm volatile(ASM : "+v"(r[k]) : "v"(a), "v"(b));                                    \
        }                                                                                           \
        float s = 0.f;                                                                              \
        _Pragma("unroll") for (int k = 0; k < CHAINS; ++k) s += r[k];                               \
        if (s == 12345.678f) out[0] = s;                                                            \
    }
#define KERNEL64(NAME, ASM)                                                                         \
    __global__ __launch_bounds__(256) void NAME(float* out, float a, float b) {                    \
        double r[CHAINS];                                                                           \
        const double da = __hiloint2double(__float_as_int(a), __float_as_int(a));                   \
        const double db = __hiloint2double(__float_as_int(b), __float_as_int(b));                   \
        _Pragma("unroll") for (int k = 0; k < CHAINS; ++k) r[k] = __hiloint2double(threadIdx.x + k, k); \
        for (int it = 0; it < ITERS; ++it) {                                                        \
            _Pragma("unroll") for (int k = 0; k < CHAINS; ++k)                                      \
                asm volatile(ASM : "+v"(r[k]) : "v"(da), "v"(db));                                  \
        }                                                                                           \
        int s = 0;                                                                                  \
        _Pragma("unroll") for (int k = 0; k < CHAINS; ++k) s += __double2hiint(r[k]) ^ __double2loint(r[k]); \
        if (s == 123456789) out[0] = (float)s;                                                      \
    }

KERNEL32(k_fma, "v_fma_f32 %0, %0, %1, %2")
KERNEL32(k_mul, "v_mul_f32 %0, %0, %1")
KERNEL32(k_add, "v_add_f32 %0, %0, %1")
KERNEL32(k_mul_lit, "v_mul_f32 %0, 0x3f8ccccd, %0")
KERNEL32(k_fma_lit, "v_fmaak_f32 %0, %0, %1, 0x3f8ccccd")
KERNEL32(k_exp, "v_exp_f32 %0, %0")
KERNEL32(k_log, "v_log_f32 %0, %0")
KERNEL32(k_rcp, "v_rcp_f32 %0, %0")
KERNEL32(k_rsq, "v_rsq_f32 %0, %0")
KERNEL32(k_sqrt, "v_sqrt_f32 %0, %0")
KERNEL32(k_sin, "v_sin_f32 %0, %0")
KERNEL32(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_cmp, "v_cmp_gt_f32 vcc, %0, %1")
KERNEL32(k_cvt_pk_u8, "v_cvt_pk_u8_f32 %0, %1, 1, %0")
KERNEL32(k_cvt_ubyte, "v_cvt_f32_ubyte1 %0, %0")
KERNEL32(k_bfe, "v_bfe_u32 %0, %0, 8, 8")
KERNEL32(k_med3, "v_med3_f32 %0, %0, %1, %2")
KERNEL32(k_fract, "v_fract_f32 %0, %0")
KERNEL32(k_cvt_i32, "v_cvt_i32_f32 %0, %0")
KERNEL32(k_cvt_u32, "v_cvt_u32_f32 %0, %0")
KERNEL32(k_dpp_shr, "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL32(k_dpp_rowshr, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL32(k_add_dpp, "v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL32(k_lshl_add, "v_lshl_add_u32 %0, %0, 2, %1")
KERNEL32(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
KERNEL32(k_perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL32(k_rndne, "v_rndne_f32 %0, %0")
KERNEL32(k_ldexp, "v_ldexp_f32 %0, %0, %1")
KERNEL32(k_max, "v_max_f32 %0, %0, %1")
KERNEL32(k_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL32(k_swmmac_nop, "v_nop")
KERNEL32(k_cndmask_e64, "v_cndmask_b32_e64 %0, %0, %1, s[10:11]")
KERNEL32(k_cndmask_fma, "v_cndmask_b32 %0, %0, %1, vcc\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2")
KERNEL32(k_fma4, "v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %1, %2")
KERNEL32(k_cmp_cnd, "v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc")
KERNEL32(k_mov, "v_mov_b32 %0, %1")
KERNEL32(k_xor, "v_xor_b32 %0, %0, %1")
KERNEL32(k_min3, "v_min3_f32 %0, %0, %1, %2")
KERNEL32(k_snop, "s_nop 0")
KERNEL32(k_sdwa, "v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1")
KERNEL64(k_pk_fma, "v_pk_fma_f32 %0, %0, %1, %2")
KERNEL64(k_pk_mul, "v_pk_mul_f32 %0, %0, %1")
KERNEL64(k_pk_add, "v_pk_add_f32 %0, %0, %1")
KERNEL64(k_pk_mov, "v_pk_mov_b32 %0, %0, %1")
KERNEL64(k_fma64, "v_fma_f64 %0, %0, %1, %2")
KERNEL64(k_mul64, "v_mul_f64 %0, %0, %1")
KERNEL64(k_add64, "v_add_f64 %0, %0, %1")

// LDS table lookups with a per-lane random index: ds_read_b32 from a 256-entry table (the u8 gamma LUT) and
// ds_read_b128 from a 1024 x float4 table (the inverse-gamma spline), plus a bank-private layout of the former
// (entry i of lane l at dword 32 * i + (l & 31): every lane of a 32-lane group owns a bank).
template <int MODE>
__global__ __launch_bounds__(256) void k_lds(float* out, const unsigned* idx, int n) {
    __shared__ __attribute__((aligned(16))) float tab[MODE == 2 ? 8192 : 4096];
    for (int i = threadIdx.x; i < (MODE == 2 ? 8192 : 4096); i += 256) tab[i] = (float)i;
    __syncthreads();
    unsigned ix[CHAINS];
    for (int k = 0; k < CHAINS; ++k) ix[k] = idx[(blockIdx.x * 256 + threadIdx.x + 977 * k) % n];
    float s = 0.f;
    for (int it = 0; it < ITERS / 4; ++it) {
#pragma unroll
        for (int k = 0; k < CHAINS; ++k) {
            if (MODE == 0) { s += tab[ix[k] & 255]; }
            else if (MODE == 1) { const float4 t = *reinterpret_cast<const float4*>(&tab[(ix[k] & 1023) * 4]); s += t.x + t.w; }
            else { s += tab[(ix[k] & 255) * 32 + (threadIdx.x & 31)]; }
            ix[k] = ix[k] * 1664525u + 1013904223u + (unsigned)s;     // next index depends on the loaded value: no hoisting
        }
    }
    if (s == 12345.678f) out[0] = s;
}

// occupancy x ILP sweep: C independent v_fma_f32 chains per lane, launched with W waves per SIMD
template <int C>
__global__ __launch_bounds__(256) void k_fma_ilp(float* out, float a, float b) {
    float r[C];
#pragma unroll
    for (int k = 0; k < C; ++k) r[k] = a * (threadIdx.x + k) + b;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int rep = 0; rep < 16 / C; ++rep)
#pragma unroll
            for (int k = 0; k < C; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(a), "v"(b));
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < C; ++k) s += r[k];
    if (s == 12345.678f) out[0] = s;
}

// The same v_fma_f32 loop with both counters read around it (round 6): s_memtime ticks with the shader clock, s_memrealtime at 100 MHz.  A wave's
// elapsed shader cycles / (instructions it issued x waves sharing its SIMD) = issue cycles per wave-instruction WITHOUT assuming a clock, and
// the ratio of the two counters is the clock the part really ran at under this load.
__global__ __launch_bounds__(256) void k_fma_clock(float* out, float a, float b, unsigned long long* stamps) {
    float r[CHAINS];
#pragma unroll
    for (int k = 0; k < CHAINS; ++k) r[k] = a * (threadIdx.x + k) + b;
    const unsigned long long c0 = __builtin_readcyclecounter(), t0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int k = 0; k < CHAINS; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(a), "v"(b));
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), t1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < CHAINS; ++k) s += r[k];
    if (s == 12345.678f) out[0] = s;
    if ((threadIdx.x & 63) == 0) {
        const size_t w = ((size_t)blockIdx.x * 256 + threadIdx.x) / 64;
        stamps[2 * w] = c1 - c0; stamps[2 * w + 1] = t1 - t0;
    }
}

typedef void (*kern_t)(float*, float, float);
struct Entry { const char* name; kern_t k; };

int main() {
    float* d_out; hipMalloc(&d_out, 64);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * 8;                       // 8 x 256 threads per CU = 8 waves per SIMD
    printf("device %s, %d CUs, clock %.0f MHz\n", prop.gcnArchName, cus, prop.clockRate / 1000.0);
    Entry es[] = {
        {"v_fma_f32", k_fma}, {"v_mul_f32", k_mul}, {"v_add_f32", k_add}, {"v_mul_f32 literal", k_mul_lit}, {"v_fmaak_f32", k_fma_lit},
        {"v_pk_fma_f32", k_pk_fma}, {"v_pk_mul_f32", k_pk_mul}, {"v_pk_add_f32", k_pk_add}, {"v_pk_mov_b32", k_pk_mov},
        {"v_exp_f32", k_exp}, {"v_log_f32", k_log}, {"v_rcp_f32", k_rcp}, {"v_rsq_f32", k_rsq}, {"v_sqrt_f32", k_sqrt}, {"v_sin_f32", k_sin},
        {"v_cndmask_b32", k_cndmask}, {"v_cmp_gt_f32", k_cmp}, {"v_cvt_pk_u8_f32", k_cvt_pk_u8}, {"v_cvt_f32_ubyte1", k_cvt_ubyte},
        {"v_bfe_u32", k_bfe}, {"v_med3_f32", k_med3}, {"v_fract_f32", k_fract}, {"v_cvt_i32_f32", k_cvt_i32}, {"v_cvt_u32_f32", k_cvt_u32},
        {"v_mov_b32 dpp wave_shr", k_dpp_shr}, {"v_mov_b32 dpp row_shr", k_dpp_rowshr}, {"v_add_f32 dpp row_shr", k_add_dpp},
        {"v_lshl_add_u32", k_lshl_add}, {"v_and_or_b32", k_and_or}, {"v_perm_b32", k_perm}, {"v_rndne_f32", k_rndne}, {"v_ldexp_f32", k_ldexp},
        {"v_max_f32", k_max}, {"v_mad_u32_u24", k_mad_u32_u24}, {"v_nop", k_swmmac_nop},
        {"v_cndmask_b32_e64 sgpr", k_cndmask_e64}, {"cndmask + 3 fma (per 4)", k_cndmask_fma}, {"4 fma (per 4)", k_fma4}, {"cmp + cndmask (per 2)", k_cmp_cnd},
        {"v_mov_b32", k_mov}, {"v_xor_b32", k_xor}, {"v_min3_f32", k_min3}, {"s_nop 0", k_snop}, {"v_lshlrev_b32_sdwa", k_sdwa},
        {"v_fma_f64", k_fma64}, {"v_mul_f64", k_mul64}, {"v_add_f64", k_add64},
    };
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    {   // the clock first: every later line quotes "cycles at 2.4 GHz", this block says what a cycle really was under an all-fma load
        const size_t nw = (size_t)blocks * 4;
        unsigned long long* d_st; hipMalloc(&d_st, nw * 16);
        hipLaunchKernelGGL(k_fma_clock, dim3(blocks), dim3(256), 0, 0, d_out, 1.0001f, 0.5f, d_st);
        hipDeviceSynchronize();
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_fma_clock, dim3(blocks), dim3(256), 0, 0, d_out, 1.0001f, 0.5f, d_st);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        std::vector<unsigned long long> st(nw * 2);
        hipMemcpy(st.data(), d_st, nw * 16, hipMemcpyDeviceToHost);
        double cyc = 0, tick = 0;
        for (size_t w = 0; w < nw; ++w) { cyc += (double)st[2 * w]; tick += (double)st[2 * w + 1]; }
        cyc /= nw; tick /= nw;
        const double mhz = cyc / tick * 100.0, ns = (double)best * 1e6 / (8.0 * ITERS * CHAINS);
        printf("clock under the all-v_fma_f32 kernel: %.0f MHz (s_memtime / s_memrealtime x 100 MHz, mean over %zu waves); kernel %.3f ms = %.3f ns per "
               "wave-instruction per SIMD = %.2f shader cycles at that clock; a wave was resident for %.0f %% of the kernel's duration\n",
               mhz, nw, best, ns, ns * mhz * 1e-3, 100.0 * (tick * 1e-5) / best);
        hipFree(d_st);
    }
    double base = 0;
    for (auto& e : es) {
        hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d_out, 1.0001f, 0.5f);
        hipDeviceSynchronize();
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, d_out, 1.0001f, 0.5f);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        // wave-instructions per SIMD: 8 blocks/CU x 4 waves / 4 SIMDs = 8 waves per SIMD
        const double per = (double)best * 1e6 / (8.0 * ITERS * CHAINS);     // ns per wave-instruction per SIMD
        if (base == 0) base = per;
        printf("%-26s %8.3f ms  %6.3f ns/wave-instr/SIMD  x%.2f of v_fma_f32  (%.2f cycles at 2.4 GHz)\n", e.name, best, per, per / base, per * 2.4);
    }
    // occupancy x ILP: ns per wave-instruction per SIMD when a SIMD holds W waves of C independent chains each
    {
        kern_t ks[5] = {k_fma_ilp<1>, k_fma_ilp<2>, k_fma_ilp<4>, k_fma_ilp<8>, k_fma_ilp<16>};
        const int cs[5] = {1, 2, 4, 8, 16};
        for (int W : {1, 2, 4, 8}) {
            for (int ci = 0; ci < 5; ++ci) {
                float best = 1e30f;
                for (int rep = 0; rep < 3; ++rep) {
                    hipEventRecord(e0, 0);
                    hipLaunchKernelGGL(ks[ci], dim3(cus * W), dim3(256), 0, 0, d_out, 1.0001f, 0.5f);
                    hipEventRecord(e1, 0);
                    hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (rep && ms < best) best = ms;
                }
                printf("v_fma_f32  W = %d waves/SIMD, %2d chains: %7.3f ns per wave-instr per SIMD\n", W, cs[ci], (double)best * 1e6 / ((double)W * ITERS * 16));
            }
        }
    }
    // LDS lookups
    std::vector<unsigned> h(1 << 16);
    unsigned x = 12345u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x >> 8; }
    unsigned* d_idx; hipMalloc(&d_idx, h.size() * 4); hipMemcpy(d_idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const char* names[3] = {"ds_read_b32 random/256", "ds_read_b128 random/1024", "ds_read_b32 bank-private"};
    for (int m = 0; m < 3; ++m) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, 0);
            if (m == 0) hipLaunchKernelGGL(k_lds<0>, dim3(blocks / 2), dim3(256), 0, 0, d_out, d_idx, (int)h.size());
            else if (m == 1) hipLaunchKernelGGL(k_lds<1>, dim3(blocks / 2), dim3(256), 0, 0, d_out, d_idx, (int)h.size());
            else hipLaunchKernelGGL(k_lds<2>, dim3(blocks / 2), dim3(256), 0, 0, d_out, d_idx, (int)h.size());
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        // 4 blocks per CU x 4 waves = 16 waves per CU, each ITERS/4 * CHAINS lookups (+ ~4 VALU each)
        const double per = (double)best * 1e6 / (16.0 * (ITERS / 4) * CHAINS);
        printf("%-26s %8.3f ms  %6.3f ns per wave-lookup per CU (%.2f cycles at 2.4 GHz)\n", names[m], best, per, per * 2.4);
    }
    return 0;
}
