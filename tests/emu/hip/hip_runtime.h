// TEST INFRASTRUCTURE: a host stand-in for <hip/hip_runtime.h>, just large enough to compile
// cyclevae-vc_amd/csrc/*.{h,hip} as plain C++ and run every kernel on host fibers (one fiber per GPU
// thread; see emu_rt.cpp).  "Device pointers" are host pointers.  Used only by tests/ (-m "not gpu").
#pragma once
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <functional>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float4 {
    float x, y, z, w;
};

typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorUnknown = 999 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 1 };

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)

namespace emu {
struct Thread;
struct ThreadView {
    dim3 tid, bid, bdim, gdim;
};
extern ThreadView* cur_view;
void launch(const std::function<void()>& body, dim3 grid, dim3 block, size_t smem, bool all_blocks_resident);
void syncthreads();
unsigned char* smem();
void yield();
}  // namespace emu

#define threadIdx (emu::cur_view->tid)
#define blockIdx (emu::cur_view->bid)
#define blockDim (emu::cur_view->bdim)
#define gridDim (emu::cur_view->gdim)
#define __syncthreads() emu::syncthreads()

#define hipLaunchKernelGGL(kernel, grid, block, smem, stream, ...) \
    emu::launch([=]() { kernel(__VA_ARGS__); }, (grid), (block), (smem), false)

static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }
typedef void* hipEvent_t;
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
enum { hipEventDisableTiming = 2 };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)1; return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
