// lvm_gfx950.h -- the gfx950-only primitives of liblvm_hip.so: instructions and address spaces that have no portable spelling.
// Included as <lvm_gfx950.h> (the Makefile passes -I.).  The CPU emulation build of the TEST SUITE (tests/emu, test infrastructure)
// puts its own lvm_gfx950.h first on the include path and never sees this file; nothing in the product refers to the emulation.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace lvm {

// Read-only tables (written by the host before the launch, never by a kernel) read through the CONSTANT address space: in a kernel
// that also stores, a plain global pointer cannot be assumed unmodified, so a wave-uniform table look-up becomes a vector load +
// s_waitcnt vmcnt(0) + v_readfirstlane -- which drains every load the wave has in flight.  Through address space 4 it is an s_load.
template <class T> using const_tab = const T __attribute__((address_space(4)))*;
template <class T> __device__ __forceinline__ const_tab<T> as_const_tab(const T* p) { return (const_tab<T>)p; }

// Buffer-resource loads / stores (raw buffer ops, stride 0): address = resource base (four SGPRs, built from wave-uniform values) +
// per-lane 32-bit byte offset (VGPR) + wave-uniform 32-bit byte offset (SGPR).  No 64-bit address arithmetic in vector registers
// (the generic form costs a VGPR pair and a v_lshl_add_u64 per load).  RANGE CHECK: the hardware compares only the VECTOR offset (+ the
// instruction's immediate) with the resource's size -- an access whose voff lies outside [0, bytes) reads 0 / is dropped, which is what the
// kernels' "drop" sentinels rely on (always placed in voff); the scalar offset is added unchecked, so voff + soff must stay inside the
// allocation whenever voff is in range (the emulation header aborts on an access that would not).
struct B96 { uint32_t a, b, c; };
typedef __amdgpu_buffer_rsrc_t BufRsrc;
typedef unsigned int lvm_u32x3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ BufRsrc buf_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float buf_ld_f32(const BufRsrc& r, uint32_t voff, uint32_t soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ B96 buf_ld_b96(const BufRsrc& r, uint32_t voff, uint32_t soff) {
    const lvm_u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(r, (int)voff, (int)soff, 0);
    return B96{v.x, v.y, v.z};
}
typedef float lvm_f2 __attribute__((vector_size(8)));
typedef float lvm_f32x4 __attribute__((ext_vector_type(4)));
typedef float lvm_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int lvm_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int lvm_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 buf_ld_f32x4(const BufRsrc& r, uint32_t voff, uint32_t soff) {
    const lvm_f32x4 v = __builtin_bit_cast(lvm_f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ lvm_f2 buf_ld_f32x2(const BufRsrc& r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(lvm_f2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0));
}
// Stores of MORE than 64 bits put the whole offset into the vector register and leave the scalar-offset field at zero.  The store reads
// its data registers some cycles after it issues; the compiler's hazard recogniser pads a following write to those registers with
// wait states ONLY for stores without a register in the soffset field (GCNHazardRecognizer: "this hazard only exists if the
// instruction is not using a register in the soffset field") -- and on gfx950 the hazard is there with an SGPR soffset too: measured in
// round 4, `buffer_store_dwordx3 v[4:6], v64, s[40:43], s14 offen` directly followed by `v_mov_b64 v[6:7], ...` stored the NEW v6 for
// lanes 12-15 of every row of 16 lanes (the last ones the store reads), a few hundred pixels of a 4K frame, not reproducible in the
// emulation build.  One v_add per store buys the compiler's own padding.
__device__ __forceinline__ void buf_st_f32x4(float a, float b, float c, float d, const BufRsrc& r, uint32_t voff, uint32_t soff) {
    lvm_f32x4 q; q.x = a; q.y = b; q.z = c; q.w = d;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(lvm_u32x4, q), r, (int)(voff + soff), 0, 0);
}
__device__ __forceinline__ void buf_st_f32x2(float a, float b, const BufRsrc& r, uint32_t voff, uint32_t soff) {
    lvm_f32x2 q; q.x = a; q.y = b;
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(lvm_u32x2, q), r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ void buf_st_b96(const B96& v, const BufRsrc& r, uint32_t voff, uint32_t soff) {
    lvm_u32x3 q; q.x = v.a; q.y = v.b; q.z = v.c;
    __builtin_amdgcn_raw_buffer_store_b96(q, r, (int)(voff + soff), 0, 0);      // (offset in the vector register: see buf_st_f32x4)
}

// ---- streaming cache policy (round 5) ----
// Data that a kernel reads ONCE or writes for a much later launch (planes of a 32-frame batch, output frames) carries the nontemporal
// hint (`nt`): tools/ubench_stream.hip measured the byte mix of the last Laplace kernel (8 + 16 B in, 12 B out per lane) at 5.9 TB/s
// with plain and 6.4-6.5 TB/s with nontemporal accesses on MI355X (profiles/r05_ubench_stream.txt), the strip skeleton of the same kernel
// 9 % faster (tools/ubench_strips.hip).  -DLVM_NT=0 builds the plain forms (A/B measurements).
#ifndef LVM_NT
#define LVM_NT 1
#endif
constexpr int kAuxStream = LVM_NT ? 2 : 0;          // aux / cache-policy operand of the raw buffer builtins: bit 1 = nt on gfx950
__device__ __forceinline__ float ld_stream_f32(const void* p) {
#if LVM_NT
    return __builtin_nontemporal_load(reinterpret_cast<const float*>(p));
#else
    return *reinterpret_cast<const float*>(p);
#endif
}
__device__ __forceinline__ uint2 ld_stream_u32x2(const void* p) {
#if LVM_NT
    const lvm_u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const lvm_u32x2*>(p));
    return make_uint2(v.x, v.y);
#else
    return *reinterpret_cast<const uint2*>(p);
#endif
}
__device__ __forceinline__ uint4 ld_stream_u32x4(const void* p) {
#if LVM_NT
    const lvm_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const lvm_u32x4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
#else
    return *reinterpret_cast<const uint4*>(p);
#endif
}
__device__ __forceinline__ float4 ld_stream_f32x4(const void* p) {
#if LVM_NT
    const lvm_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const lvm_f32x4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *reinterpret_cast<const float4*>(p);
#endif
}
__device__ __forceinline__ void st_stream_b96(void* p, uint32_t a, uint32_t b, uint32_t c) {      // 4-byte aligned
    typedef unsigned int u3a4 __attribute__((ext_vector_type(3), aligned(4)));
    u3a4 q; q.x = a; q.y = b; q.z = c;
#if LVM_NT
    __builtin_nontemporal_store(q, reinterpret_cast<u3a4*>(p));
#else
    *reinterpret_cast<u3a4*>(p) = q;
#endif
}
__device__ __forceinline__ void st_stream_f32x4(void* p, float a, float b, float c, float d) {
    lvm_f32x4 q; q.x = a; q.y = b; q.z = c; q.w = d;
#if LVM_NT
    __builtin_nontemporal_store(q, reinterpret_cast<lvm_f32x4*>(p));
#else
    *reinterpret_cast<lvm_f32x4*>(p) = q;
#endif
}
// the buffer-resource forms of the same
__device__ __forceinline__ float buf_lds_f32(const BufRsrc& r, uint32_t voff, uint32_t soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, kAuxStream));
}
__device__ __forceinline__ B96 buf_lds_b96(const BufRsrc& r, uint32_t voff, uint32_t soff) {
    const lvm_u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(r, (int)voff, (int)soff, kAuxStream);
    return B96{v.x, v.y, v.z};
}
__device__ __forceinline__ float4 buf_lds_f32x4(const BufRsrc& r, uint32_t voff, uint32_t soff) {
    const lvm_f32x4 v = __builtin_bit_cast(lvm_f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, kAuxStream));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint2 buf_lds_u32x2(const BufRsrc& r, uint32_t voff, uint32_t soff) {
    const lvm_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, kAuxStream);
    return make_uint2(v.x, v.y);
}
__device__ __forceinline__ uint4 buf_lds_u32x4(const BufRsrc& r, uint32_t voff, uint32_t soff) {
    const lvm_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, kAuxStream);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ lvm_f2 buf_lds_f32x2(const BufRsrc& r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(lvm_f2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, kAuxStream));
}
__device__ __forceinline__ void buf_sts_f32x4(float a, float b, float c, float d, const BufRsrc& r, uint32_t voff, uint32_t soff) {
    lvm_f32x4 q; q.x = a; q.y = b; q.z = c; q.w = d;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(lvm_u32x4, q), r, (int)(voff + soff), 0, kAuxStream);     // (offset in the vector register: see buf_st_f32x4)
}
__device__ __forceinline__ void buf_sts_f32x2(float a, float b, const BufRsrc& r, uint32_t voff, uint32_t soff) {
    lvm_f32x2 q; q.x = a; q.y = b;
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(lvm_u32x2, q), r, (int)voff, (int)soff, kAuxStream);
}
__device__ __forceinline__ void buf_sts_b96(const B96& v, const BufRsrc& r, uint32_t voff, uint32_t soff) {
    lvm_u32x3 q; q.x = v.a; q.y = v.b; q.z = v.c;
    __builtin_amdgcn_raw_buffer_store_b96(q, r, (int)(voff + soff), 0, kAuxStream);
}

// ---- packed FP32: two floats in a register pair, one v_pk_{add,mul,fma}_f32 per operation (full rate on gfx950) ----
__device__ __forceinline__ lvm_f2 f2_fma(lvm_f2 a, lvm_f2 b, lvm_f2 c) { return __builtin_elementwise_fma(a, b, c); }
// A zero-instruction fence on VALUES: the operands pass through an empty asm statement, so everything that produces them is
// emitted before this point and everything that consumes them after it.  (Instruction selection orders a basic block by data
// dependence, not by source order: without it the long fma chains of k_rz_split_rows are emitted next to the store that needs them,
// with the operand rows of up to nine steps still live.)
__device__ __forceinline__ void lvm_pin(lvm_f2& a, lvm_f2& b) { asm volatile("" : "+v"(a), "+v"(b)); }
// Memory instructions stay on their side of this point (a compiler-level fence, no instruction): a prefetching load is not sunk
// towards its first use.
__device__ __forceinline__ void lvm_issue_fence() { asm volatile("" ::: "memory"); }
__device__ __forceinline__ void lvm_pin(float& a, float& b) { asm volatile("" : "+v"(a), "+v"(b)); }

// ---- forward Lab table (lab_lut.h) ----
typedef unsigned short lut_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t lut_dot2(uint32_t pair, uint32_t wts, uint32_t acc) {      // v_dot2_u32_u16
    return __builtin_amdgcn_udot2(__builtin_bit_cast(lut_u2, pair), __builtin_bit_cast(lut_u2, wts), acc, false);
}
// (lo16(d), lo16(e)) and (hi16(d), hi16(e)) of two dwords
__device__ __forceinline__ uint32_t lut_lo2(uint32_t d, uint32_t e) { return __builtin_amdgcn_perm(e, d, 0x05040100u); }
__device__ __forceinline__ uint32_t lut_hi2(uint32_t d, uint32_t e) { return __builtin_amdgcn_perm(e, d, 0x07060302u); }
// 24-bit x 24-bit -> low 32 bits as ONE full-rate v_mul_u32_u24.  Spelled as an instruction: a plain product is re-associated by the
// compiler ((x0 y0) wz -> (wz x0) y0, whose intermediate no longer fits 24 bits) and then becomes the quarter-rate v_mul_lo_u32 --
// four of them per pixel in round 3's ISA, ~12 issue slots of the conversion's ~94
__device__ __forceinline__ uint32_t lut_mul24(uint32_t a, uint32_t b) { uint32_t r; asm("v_mul_u32_u24_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// four values < 256 -> one dword (a in byte 0) as three v_perm_b32: a | b << 8 | c << 16 | d << 24 is otherwise built from two shifts, a
// v_or3_b32 and a v_lshl_or_b32; an inline-asm v_lshl_or_b32 chain costs an s_nop per statement (the hazard recogniser pads what it cannot see)
__device__ __forceinline__ uint32_t lvm_pack_b4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    const uint32_t lo = __builtin_amdgcn_perm(b, a, 0x0c0c0400u), hi = __builtin_amdgcn_perm(d, c, 0x0c0c0400u);
    return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
}

// the two counters of a wave: s_memtime ticks with the shader clock (it slows down when the part throttles), s_memrealtime at a
// constant 100 MHz -- their ratio over an interval is the average shader clock of that interval (lvm_debug_clock_probe_*)
__device__ __forceinline__ unsigned long long lvm_clock_core() { return __builtin_readcyclecounter(); }
__device__ __forceinline__ unsigned long long lvm_clock_real() { return __builtin_amdgcn_s_memrealtime(); }
__device__ __forceinline__ void lvm_sleep() { __builtin_amdgcn_s_sleep(127); }

// the lanes of the wave whose predicate holds (v_cmp into an SGPR pair); every lane of the wave must execute it
__device__ __forceinline__ unsigned long long lvm_ballot64(bool pred) { return __builtin_amdgcn_ballot_w64(pred); }
// Orders the LDS accesses of ONE wave: what its lanes wrote before is what its lanes read after (the LDS serves a wave's requests in
// issue order; the wait + the compiler barrier keep loads from moving up).  No other wave is waited for.  Every lane of every wave of the
// workgroup must execute it the same number of times (the emulation spells it as a yield).
__device__ __forceinline__ void lvm_wave_lds_sync() {
    __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
// Inclusive prefix sum over the 64 lanes of a wave in six DPP steps: Hillis-Steele inside the rows of 16 (row_shr 1, 2, 4, 8), then lane 15
// of rows 0 / 2 added to rows 1 / 3 (row_bcast:15, row mask 0xA) and lane 31 to rows 2 and 3 (row_bcast:31, row mask 0xC).  Lanes without a
// source, and rows outside the mask, receive 0.
__device__ __forceinline__ int lvm_wave_prefix_add(int v, int /*lane*/) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);
    return v;
}
// the value of lane 63, in every lane (v_readlane into an SGPR)
__device__ __forceinline__ int lvm_wave_last(int v) { return __builtin_amdgcn_readlane(v, 63); }
}  // namespace lvm
