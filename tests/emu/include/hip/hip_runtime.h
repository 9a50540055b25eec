// TEST INFRASTRUCTURE ONLY -- a tiny fiber-based emulation of the HIP subset used by
// live-video-magnification_amd/csrc, so the *kernel logic* (tiling, border rules, operation
// order) can be exercised by the CPU test-suite (`-m "not gpu"`) in a container without a GPU.
// The build lives in tests/emu/_build/ and is only ever loaded by tests/; the product library
// (liblvm_hip.so) is always the hipcc/gfx950 build and has no CPU path.
// One workgroup runs at a time; its work-items are fibres (own stacks, a 20-instruction switch: hip_emu.cpp) and __syncthreads() yields
// to a round-robin scheduler.  Device allocations are filled with 0xFF (NaN floats) so reads of
// uninitialised memory surface in the parity checks.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600 };
typedef struct ihipStream_t* hipStream_t;
typedef struct ihipEvent_t* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1 };

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#ifdef HIPEMU_POISON_LDS
// Special build (tools/emu_lds_poison.sh): every __shared__ array lives in one linker section that the launcher fills with 0xFF (NaN as
// float) before each workgroup -- on the GPU a workgroup finds whatever the previous one left in LDS.  Single-threaded use only.
#define __shared__ static __attribute__((section("lds_emu")))
#else
#define __shared__ static thread_local
#endif
#define __launch_bounds__(...)
#define HIP_KERNEL_NAME(...) __VA_ARGS__
#define __builtin_unpredictable(x) (x)   // clang-only optimiser hint

namespace hipemu {
struct State { dim3 tid, bid, bdim, gdim; };
extern thread_local State st;
void sync();
void launch(const std::function<void()>& body, dim3 grid, dim3 block);
int dpp_wave_shift(int old, int src, int ctrl);
int wave_bpermute(int byte_addr, int src);
unsigned long long wave_ballot(int pred);
}  // namespace hipemu
#define threadIdx (hipemu::st.tid)
#define blockIdx (hipemu::st.bid)
#define blockDim (hipemu::st.bdim)
#define gridDim (hipemu::st.gdim)
static inline void __syncthreads() { hipemu::sync(); }

template <class K, class... A>
static inline void hipLaunchKernelGGL(K kern, dim3 grid, dim3 block, size_t, hipStream_t, A... args) {
    hipemu::launch([=]() { kern(args...); }, grid, block);
}

struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
static inline uint2 make_uint2(unsigned x, unsigned y) { uint2 r; r.x = x; r.y = y; return r; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { uint4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }



struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
static inline float2 make_float2(float x, float y) { float2 r; r.x = x; r.y = y; return r; }
static inline float4 make_float4(float x, float y, float z, float w) { float4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
#ifndef __clang__        // (clang has both as builtins: tools/emu_uninit.sh builds this emulation with clang for -ftrivial-auto-var-init)
template <class T> static inline T __builtin_nontemporal_load(const T* p) { return *p; }
template <class T, class U> static inline void __builtin_nontemporal_store(U v, T* p) { *p = (T)v; }
#endif
static inline float __fdividef(float a, float b) { return a / b; }
static inline void __builtin_amdgcn_sched_barrier(int) {}
static inline float __builtin_amdgcn_logf(float x) { return log2f(x); }
static inline float __builtin_amdgcn_fmed3f(float a, float b, float c) {      // v_med3_f32: the MINIMUM of the non-NaN inputs when an input is NaN
    if (a != a || b != b || c != c) return fminf(fminf(a, b), c);
    return a < b ? (b < c ? b : (a < c ? c : a)) : (a < c ? a : (b < c ? c : b));
}
static inline float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }
static inline float __builtin_amdgcn_fractf(float x) { float f = x - floorf(x); return f < 0.99999994f ? f : 0.99999994f; }   // v_fract_f32
static inline unsigned __builtin_amdgcn_cvt_pk_u8_f32(float v, unsigned byte, unsigned old) {   // RNE, saturating, NaN -> 0
    float r = (v == v) ? nearbyintf(v) : 0.f;
    r = r < 0.f ? 0.f : (r > 255.f ? 255.f : r);
    return (old & ~(0xffu << (8 * byte))) | ((unsigned)r << (8 * byte));
}
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline double __builtin_amdgcn_rcp(double x) { return 1.0 / x; }
static inline float __builtin_amdgcn_sqrtf(float x) { return sqrtf(x); }
static inline float __builtin_amdgcn_rsqf(float x) { return 1.0f / sqrtf(x); }
static inline float __builtin_amdgcn_sinf(float rev) { return sinf(rev * 6.283185307179586f); }   // v_sin_f32: argument in revolutions
static inline float __builtin_amdgcn_cosf(float rev) { return cosf(rev * 6.283185307179586f); }
// DPP wave_shr:1 (0x138) / wave_shl:1 (0x130): every lane of the wave must execute it (uniform control flow)
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int, int, bool) { return hipemu::dpp_wave_shift(old, src, ctrl); }
static inline int __builtin_amdgcn_ds_bpermute(int byte_addr, int src) { return hipemu::wave_bpermute(byte_addr, src); }
static inline int __builtin_amdgcn_readfirstlane(int x) { return x; }
// wave-uniform shortcuts compute what the general path selects, so a per-lane answer is equivalent here
static inline unsigned long long __builtin_amdgcn_ballot_w64(bool c) { return c ? 1ull : 0ull; }   // only ever applied to wave-uniform values
static inline int __float_as_int(float f) { int i; std::memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; std::memcpy(&f, &i, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned i; std::memcpy(&i, &f, 4); return i; }
static inline float __uint_as_float(unsigned i) { float f; std::memcpy(&f, &i, 4); return f; }

static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 3; return hipSuccess; }   // three "CUs": the persistent kernels loop
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "hip-emu error"; }
// HIPEMU_GUARD=1: every "device" buffer is its own mmap region with 16 MiB of PROT_NONE either side and its end on a page end, so
// that an access far outside a buffer faults like it does on the GPU (malloc neighbours would silently absorb it; ASan's redzones
// only catch near misses)
namespace hipemu { void* guard_malloc(size_t n); bool guard_free(void* p); }
static inline hipError_t hipMalloc(void** p, size_t n) {
    *p = hipemu::guard_malloc(n ? n : 1);
    if (!*p) *p = std::malloc(n ? n : 1);
    if (!*p) return hipErrorOutOfMemory;
    std::memset(*p, 0xFF, n); return hipSuccess;
}
static inline hipError_t hipFree(void* p) { if (!hipemu::guard_free(p)) std::free(p); return hipSuccess; }
// page-locked host memory: tracked, so that hipPointerGetAttributes can tell it from pageable memory (the product's zero-copy path)
namespace hipemu { void host_track(void* p, size_t n); void host_untrack(void* p); void* host_lookup(const void* p); }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) {
    *p = std::malloc(n ? n : 1);
    if (!*p) return hipErrorOutOfMemory;
    hipemu::host_track(*p, n ? n : 1); return hipSuccess;
}
static inline hipError_t hipHostFree(void* p) { hipemu::host_untrack(p); std::free(p); return hipSuccess; }
enum hipMemoryType { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2 };
struct hipPointerAttribute_t { hipMemoryType type; int device; void* devicePointer; void* hostPointer; };
static inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p) {
    void* q = hipemu::host_lookup(p);
    a->type = q ? hipMemoryTypeHost : hipMemoryTypeUnregistered; a->device = 0; a->devicePointer = q; a->hostPointer = q;
    return hipSuccess;
}
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { std::memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { std::memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t wbytes, size_t h, hipMemcpyKind, hipStream_t = nullptr) {
    for (size_t y = 0; y < h; ++y) std::memcpy((char*)d + y * dp, (const char*)s + y * sp, wbytes);
    return hipSuccess;
}
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)(uintptr_t)1; return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = (hipStream_t)(uintptr_t)1; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }   // the emulation runs every launch to completion
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline int atomicMin(int* p, int v) { int o = *p; if (v < o) *p = v; return o; }
static inline int atomicMax(int* p, int v) { int o = *p; if (v > o) *p = v; return o; }
static inline unsigned atomicOr(unsigned* p, unsigned v) { unsigned o = *p; *p = o | v; return o; }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p = o + v; return o; }
static inline unsigned long long atomicMin(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; if (v < o) *p = v; return o; }
static inline void __threadfence() {}
// graphs are not emulated: capture reports failure and the library falls back to plain launches
typedef struct ihipGraph_t* hipGraph_t;
typedef struct ihipGraphExec_t* hipGraphExec_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal, hipStreamCaptureModeThreadLocal, hipStreamCaptureModeRelaxed };
static inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorInvalidValue; }
static inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return hipErrorInvalidValue; }
static inline hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t, void*, void*, size_t) { *e = nullptr; return hipErrorInvalidValue; }
static inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorInvalidValue; }
static inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
enum { hipEventDisableTiming = 2 };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
