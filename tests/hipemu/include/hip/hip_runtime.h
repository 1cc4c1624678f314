// TEST INFRASTRUCTURE ONLY -- a minimal single-threaded emulation of the HIP execution model (fibers per thread of a
// workgroup, __syncthreads / wave shuffles as fiber barriers) so that the indexing / halo / boundary logic of the
// kernels in phiflow_amd/csrc can be exercised in a container without a GPU. It is never shipped, never loaded by the
// package `phiflow_amd` (which loads phiflow_amd/lib/libphihip.so only and fails loudly without a HIP device), and it
// says nothing about performance. Build: tests/hipemu/build_emu.sh -> tests/hipemu/libphihip_emu.so
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <functional>

using std::max;
using std::min;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

typedef void* hipStream_t;
typedef struct hipemu_event* hipEvent_t;
enum hipError_t { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorLaunchFailure = 719 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
#define hipHostMallocDefault 0
#define hipHostMallocMapped 2
struct hipDeviceProp_t {
    int multiProcessorCount;
    char name[64];
};

namespace hipemu {
struct Fiber;
extern Fiber* g_cur;
extern dim3 g_block_idx, g_block_dim, g_grid_dim;
dim3& cur_thread_idx();
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
void launch_resident(dim3 grid, dim3 block, const std::function<void()>& body);   // all workgroups alive at once (persistent kernels)
void spin_yield();           // a fiber polling another workgroup's flag lets the others run
void sync_block();
void sync_wave();
void* wave_slot(int lane);   // 16-byte exchange slot of `lane` in the calling fiber's wave
int cur_lane();
void* dynamic_lds();         // 160 KB shared by the (one) running workgroup: `extern __shared__` of the device build
}  // namespace hipemu

#define threadIdx (hipemu::cur_thread_idx())
#define blockIdx (hipemu::g_block_idx)
#define blockDim (hipemu::g_block_dim)
#define gridDim (hipemu::g_grid_dim)

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipemu::launch((grid), (block), [&]() { kernel(__VA_ARGS__); })
// launch of a kernel whose workgroups synchronise with each other (device build: a plain launch of a grid that is resident as a whole)
#define hipemuLaunchResident(kernel, grid, block, shmem, stream, ...) \
    hipemu::launch_resident((grid), (block), [&]() { kernel(__VA_ARGS__); })

inline void __syncthreads() { hipemu::sync_block(); }
inline void __threadfence() {}        // one fiber runs at a time, in program order
inline int __mul24(int a, int b) { return a * b; }

// fibers of the emulation run one at a time: a plain read-modify-write is atomic
template <typename T>
inline T atomicAdd(T* p, T v) { const T old = *p; *p = old + v; return old; }

template <typename T>
inline T __shfl_down(T v, unsigned delta, int width = 64) {
    static_assert(sizeof(T) <= 16, "shuffle payload too large");
    const int lane = hipemu::cur_lane();
    memcpy(hipemu::wave_slot(lane), &v, sizeof(T));
    hipemu::sync_wave();
    T r = v;
    const int src = lane + (int)delta;
    if (src / width == lane / width && src < 64) memcpy(&r, hipemu::wave_slot(src), sizeof(T));
    hipemu::sync_wave();
    return r;
}

template <typename T>
inline T __shfl_xor(T v, int mask, int width = 64) {
    const int lane = hipemu::cur_lane();
    memcpy(hipemu::wave_slot(lane), &v, sizeof(T));
    hipemu::sync_wave();
    T r = v;
    const int src = lane ^ mask;
    if (src / width == lane / width && src < 64) memcpy(&r, hipemu::wave_slot(src), sizeof(T));
    hipemu::sync_wave();
    return r;
}

hipError_t hipMalloc(void** p, size_t bytes);
hipError_t hipFree(void* p);
hipError_t hipMemGetInfo(size_t* free_bytes, size_t* total_bytes);
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned flags);
hipError_t hipHostFree(void* p);
inline hipError_t hipHostGetDevicePointer(void** dev, void* host, unsigned) { *dev = host; return hipSuccess; }   // one address space
hipError_t hipMemsetAsync(void* p, int value, size_t bytes, hipStream_t s);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipDeviceSynchronize();
hipError_t hipGetLastError();
const char* hipGetErrorString(hipError_t e);
hipError_t hipGetDeviceCount(int* n);
hipError_t hipSetDevice(int d);
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* prop, int d);
// emulation: pretend 4 resident workgroups per CU for every kernel
template <typename F>
inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, F, int, size_t) { *n = 4; return hipSuccess; }
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int bytes) { return bytes <= 160 * 1024 ? hipSuccess : hipErrorInvalidValue; }
// emulation: streams never capture (the library asks before it synchronises inside a call, e.g. the first-call autotune)
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone = 0, hipStreamCaptureStatusActive = 1, hipStreamCaptureStatusInvalidated = 2 };
inline hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* st) { *st = hipStreamCaptureStatusNone; return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e);
#define hipEventDisableTiming 2
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventQuery(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
