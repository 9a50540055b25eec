// TEST INFRASTRUCTURE ONLY -- the CPU spelling of the product's lvm_gfx950.h (csrc/lvm_gfx950.h: constant address space, raw buffer
// loads / stores, v_dot2_i32_i16 / v_perm_b32 / v_mul_u32_u24), same names and semantics, for the emulation build of tests/emu.
// tests/emu/build_emu.sh puts this directory first on the include path, so <lvm_gfx950.h> resolves here.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace lvm {

template <class T> using const_tab = const T*;
template <class T> __device__ __forceinline__ const_tab<T> as_const_tab(const T* p) { return p; }
struct B96 { uint32_t a, b, c; };
typedef float lvm_f2 __attribute__((vector_size(8)));
struct BufRsrc { char* base; uint32_t bytes; };
// Range check of a raw buffer access as the HARDWARE does it: only the vector offset (voff) is compared with the resource's size -- the
// scalar offset (soff) is added afterwards, unchecked.  An access whose voff is in range but whose voff + soff is not would read /
// write outside the buffer on the GPU: the emulation aborts on it instead of quietly returning 0 (ADVICE, round 4).
__device__ __forceinline__ bool buf_in_range(const BufRsrc& r, uint32_t voff, uint32_t soff, uint32_t n) {
    if ((uint64_t)voff + n > r.bytes) return false;                          // dropped / reads 0, like the hardware
    if ((uint64_t)voff + soff + n > r.bytes) { std::fprintf(stderr, "hip-emu: raw buffer access with voff %u in range but voff + soff = %llu outside %u bytes: out of bounds on the GPU\n", voff, (unsigned long long)voff + soff, r.bytes); std::abort(); }
    return true;
}
__device__ __forceinline__ BufRsrc buf_rsrc(const void* base, uint32_t bytes) { return BufRsrc{(char*)base, bytes}; }
__device__ __forceinline__ float buf_ld_f32(const BufRsrc& r, uint32_t voff, uint32_t soff) {
    const uint64_t o = (uint64_t)voff + soff;
    float v = 0.f;
    if (buf_in_range(r, voff, soff, 4)) std::memcpy(&v, r.base + o, 4);
    return v;
}
__device__ __forceinline__ B96 buf_ld_b96(const BufRsrc& r, uint32_t voff, uint32_t soff) {
    const uint64_t o = (uint64_t)voff + soff;
    B96 v{0, 0, 0};
    if (buf_in_range(r, voff, soff, 12)) std::memcpy(&v, r.base + o, 12);
    return v;
}
__device__ __forceinline__ float4 buf_ld_f32x4(const BufRsrc& r, uint32_t voff, uint32_t soff) {
    const uint64_t o = (uint64_t)voff + soff;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (buf_in_range(r, voff, soff, 16)) std::memcpy(v, r.base + o, 16);
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ lvm_f2 buf_ld_f32x2(const BufRsrc& r, uint32_t voff, uint32_t soff) {
    const uint64_t o = (uint64_t)voff + soff;
    lvm_f2 v = {0.f, 0.f};
    if (buf_in_range(r, voff, soff, 8)) std::memcpy(&v, r.base + o, 8);
    return v;
}
__device__ __forceinline__ void buf_st_f32x4(float a, float b, float c, float d, const BufRsrc& r, uint32_t voff, uint32_t soff) {
    const uint64_t o = (uint64_t)voff + soff;
    const float v[4] = {a, b, c, d};
    if (o + 16 <= r.bytes) std::memcpy(r.base + o, v, 16);       // (the product puts voff + soff into the vector register here: the SUM is range-checked)
}
__device__ __forceinline__ void buf_st_f32x2(float a, float b, const BufRsrc& r, uint32_t voff, uint32_t soff) {
    const uint64_t o = (uint64_t)voff + soff;
    const float v[2] = {a, b};
    if (buf_in_range(r, voff, soff, 8)) std::memcpy(r.base + o, v, 8);
}
__device__ __forceinline__ void buf_st_b96(const B96& v, const BufRsrc& r, uint32_t voff, uint32_t soff) {
    const uint64_t o = (uint64_t)voff + soff;
    if (o + 12 <= r.bytes) std::memcpy(r.base + o, &v, 12);      // (likewise: whole offset in the vector register)
}
// streaming (nontemporal) forms: a cache hint only -- the plain accesses here
__device__ __forceinline__ float ld_stream_f32(const void* p) { float v; std::memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint2 ld_stream_u32x2(const void* p) { uint2 v; std::memcpy(&v, p, 8); return v; }
__device__ __forceinline__ uint4 ld_stream_u32x4(const void* p) { uint4 v; std::memcpy(&v, p, 16); return v; }
__device__ __forceinline__ float4 ld_stream_f32x4(const void* p) { float4 v; std::memcpy(&v, p, 16); return v; }
__device__ __forceinline__ void st_stream_b96(void* p, uint32_t a, uint32_t b, uint32_t c) { const uint32_t v[3] = {a, b, c}; std::memcpy(p, v, 12); }
__device__ __forceinline__ void st_stream_f32x4(void* p, float a, float b, float c, float d) { const float v[4] = {a, b, c, d}; std::memcpy(p, v, 16); }
__device__ __forceinline__ float buf_lds_f32(const BufRsrc& r, uint32_t voff, uint32_t soff) { return buf_ld_f32(r, voff, soff); }
__device__ __forceinline__ uint2 buf_lds_u32x2(const BufRsrc& r, uint32_t voff, uint32_t soff) {
    const uint64_t o = (uint64_t)voff + soff; uint2 v = make_uint2(0, 0);
    if (buf_in_range(r, voff, soff, 8)) std::memcpy(&v, r.base + o, 8);
    return v;
}
__device__ __forceinline__ uint4 buf_lds_u32x4(const BufRsrc& r, uint32_t voff, uint32_t soff) {
    const uint64_t o = (uint64_t)voff + soff; uint4 v = make_uint4(0, 0, 0, 0);
    if (buf_in_range(r, voff, soff, 16)) std::memcpy(&v, r.base + o, 16);
    return v;
}
__device__ __forceinline__ B96 buf_lds_b96(const BufRsrc& r, uint32_t voff, uint32_t soff) { return buf_ld_b96(r, voff, soff); }
__device__ __forceinline__ float4 buf_lds_f32x4(const BufRsrc& r, uint32_t voff, uint32_t soff) { return buf_ld_f32x4(r, voff, soff); }
__device__ __forceinline__ lvm_f2 buf_lds_f32x2(const BufRsrc& r, uint32_t voff, uint32_t soff) { return buf_ld_f32x2(r, voff, soff); }
__device__ __forceinline__ void buf_sts_f32x4(float a, float b, float c, float d, const BufRsrc& r, uint32_t voff, uint32_t soff) { buf_st_f32x4(a, b, c, d, r, voff, soff); }
__device__ __forceinline__ void buf_sts_f32x2(float a, float b, const BufRsrc& r, uint32_t voff, uint32_t soff) { buf_st_f32x2(a, b, r, voff, soff); }
__device__ __forceinline__ void buf_sts_b96(const B96& v, const BufRsrc& r, uint32_t voff, uint32_t soff) { buf_st_b96(v, r, voff, soff); }
__device__ __forceinline__ lvm_f2 f2_fma(lvm_f2 a, lvm_f2 b, lvm_f2 c) { lvm_f2 v = {__builtin_fmaf(a[0], b[0], c[0]), __builtin_fmaf(a[1], b[1], c[1])}; return v; }
__device__ __forceinline__ void lvm_pin(lvm_f2&, lvm_f2&) {}
__device__ __forceinline__ void lvm_pin(float&, float&) {}
__device__ __forceinline__ void lvm_issue_fence() {}
__device__ __forceinline__ uint32_t lut_dot2(uint32_t pair, uint32_t wts, uint32_t acc) {
    return acc + (pair & 0xffffu) * (wts & 0xffffu) + (pair >> 16) * (wts >> 16);
}
__device__ __forceinline__ uint32_t lut_lo2(uint32_t d, uint32_t e) { return (d & 0xffffu) | (e << 16); }
__device__ __forceinline__ uint32_t lut_hi2(uint32_t d, uint32_t e) { return (d >> 16) | (e & 0xffff0000u); }
__device__ __forceinline__ uint32_t lut_mul24(uint32_t a, uint32_t b) { return (a & 0xffffffu) * (b & 0xffffffu); }

__device__ __forceinline__ uint32_t lvm_pack_b4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return (a & 255u) | ((b & 255u) << 8) | ((c & 255u) << 16) | (d << 24); }

// (no clocks on the CPU: two counters at a fixed ratio of 20 : 1, so that the probe's plumbing reports "2000 MHz")
inline unsigned long long& emu_ticks() { static unsigned long long t = 0; return t; }
__device__ __forceinline__ unsigned long long lvm_clock_real() { return emu_ticks() += 100; }
__device__ __forceinline__ unsigned long long lvm_clock_core() { return emu_ticks() * 20; }
__device__ __forceinline__ void lvm_sleep() {}

// a real cross-lane ballot (the header's __builtin_amdgcn_ballot_w64 stand-in is for wave-uniform predicates only)
__device__ __forceinline__ unsigned long long lvm_ballot64(bool pred) { return hipemu::wave_ballot(pred ? 1 : 0); }
__device__ __forceinline__ void lvm_wave_lds_sync() { hipemu::sync(); }
__device__ __forceinline__ int lvm_wave_prefix_add(int v, int lane) {      // the same sums by ds_bpermute: lane i reads lane i - d
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __builtin_amdgcn_ds_bpermute((lane - d) * 4, v);
        if (lane >= d) v += t;
    }
    return v;
}
__device__ __forceinline__ int lvm_wave_last(int v) { return __builtin_amdgcn_ds_bpermute(63 * 4, v); }
}  // namespace lvm
