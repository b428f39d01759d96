// hip_emu.h -- TEST-ONLY CPU emulation of the small HIP subset the kernels in
// triton_vm_amd/csrc use.  It exists so that the *same kernel source* that hipcc compiles for
// gfx950 can be executed in the GPU-less build container by the `-m "not gpu"` tests
// (tests/test_emu_*.py) and compared with the oracle before any GPU minute is spent.
//
// It is NOT a product path and NOT a CPU fallback: the product library libtriton_hip.so is built by
// hipcc only and fails loudly without a GPU.  The emulated library lives in tests/emu/ and is
// loaded by tests only.
//
// Model: one workgroup at a time; every work-item is a ucontext fiber; __syncthreads() and the
// wave-level exchange primitives park the fiber until its group has arrived.
#pragma once
#include <ucontext.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define TVM_EMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __restrict__ __restrict
#define __shared__ static thread_local   /* one copy per worker thread = per workgroup in flight */

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_emu { unsigned x, y, z; };

namespace emu {
struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    uint3_emu tid{0, 0, 0};
    bool done = false;
};
struct Group { int count = 0; int gen = 0; };
// workgroups run in parallel on a pool of host threads: everything a workgroup touches is per thread
extern thread_local Fiber* cur;
extern thread_local uint3_emu block_idx;
extern dim3 block_dim, grid_dim;   // the same for every workgroup of a launch
extern thread_local unsigned char* dyn_smem;
extern thread_local uint64_t xchg[4096];  // per-work-item exchange slots for shuffles
void yield();
void sync_block();
void sync_wave();
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
int lane_id();
}  // namespace emu

#define threadIdx (emu::cur->tid)
#define blockIdx (emu::block_idx)
#define blockDim (emu::block_dim)
#define gridDim (emu::grid_dim)

static inline void __syncthreads() { emu::sync_block(); }
static inline uint64_t __umul64hi(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) >> 64); }
static inline unsigned __brev(unsigned x) {
    unsigned r = 0;
    for (int i = 0; i < 32; i++) r |= ((x >> i) & 1u) << (31 - i);
    return r;
}
// 64-lane wave shuffles (all lanes of the wave must participate)
static inline uint64_t __shfl_xor(uint64_t v, int mask, int width = 64) {
    (void)width;
    int flat = emu::cur->tid.x + emu::block_dim.x * (emu::cur->tid.y + emu::block_dim.y * emu::cur->tid.z);
    emu::xchg[flat] = v;
    emu::sync_wave();
    uint64_t r = emu::xchg[flat ^ mask];
    emu::sync_wave();
    return r;
}
static inline uint64_t __shfl(uint64_t v, int src, int width = 64) {
    (void)width;
    int flat = emu::cur->tid.x + emu::block_dim.x * (emu::cur->tid.y + emu::block_dim.y * emu::cur->tid.z);
    emu::xchg[flat] = v;
    emu::sync_wave();
    uint64_t r = emu::xchg[(flat & ~63) | (src & 63)];
    emu::sync_wave();
    return r;
}
// every lane of a wave deposits `bytes` (<= 64) bytes; `all` receives the 64 deposits of the wave in lane order
namespace emu { extern thread_local unsigned char xchg_wide[4096][64]; }
static inline void emu_wave_gather(const void* mine, size_t bytes, void* all) {
    int flat = emu::cur->tid.x + emu::block_dim.x * (emu::cur->tid.y + emu::block_dim.y * emu::cur->tid.z);
    memcpy(emu::xchg_wide[flat], mine, bytes);
    emu::sync_wave();
    for (int l = 0; l < 64; l++) memcpy((unsigned char*)all + l * bytes, emu::xchg_wide[(flat & ~63) | l], bytes);
    emu::sync_wave();
}

// ---- runtime API subset -------------------------------------------------------------------
typedef int hipError_t;
typedef void* hipStream_t;
typedef struct emu_event { double t; }* hipEvent_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
static inline hipError_t hipMalloc(void** p, size_t n) {
    *p = malloc(n ? n : 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
// (the emulation executes every "asynchronous" call at once, so a second stream and waits on events order nothing: what the CPU
// suite checks of the side lane -- tvm_side_*, capi.hip -- is the call sequence and the results, the GPU suite checks the ordering)
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)(void*)1; return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new emu_event{0}; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new emu_event{0}; return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = nullptr);
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)((b->t - a->t) * 1e3); return hipSuccess; }
static inline hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = *t = (size_t)8 << 30; return hipSuccess; }
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };

#define TVM_LAUNCH(kernel, grid, block, shmem, stream, ...) \
    emu::launch((grid), (block), (shmem), [=]() { kernel(__VA_ARGS__); })
#define TVM_DYN_SMEM(T, name) T* name = reinterpret_cast<T*>(emu::dyn_smem)
