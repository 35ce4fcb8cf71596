// TEST INFRASTRUCTURE (tests/cpu_emul): a host-only stand-in for the part of the HIP runtime API the product uses, so that the product's own
// sources (rtwin_capi.cpp, rtw_device.hip with every kernel and launch wrapper) compile for the CPU and run under AddressSanitizer / UBSan
// (GPU sanitizers are not available on the pool).  "Device memory" is the host heap -- every hipMalloc is its own exactly-sized allocation,
// which is what makes an out-of-bounds access of a kernel visible -- streams are in order, launches run to completion before they return.
// Never part of librtwin.so; the product build does not see this directory.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600, hipErrorUnknown = 999 };
typedef struct hipemuStream* hipStream_t;
typedef struct hipemuEvent* hipEvent_t;
typedef struct hipemuGraph* hipGraph_t;
typedef struct hipemuGraphExec* hipGraphExec_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0 };
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal, hipStreamCaptureModeRelaxed };
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 1 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };

#ifndef HIPEMU_COMPUTE_UNITS
#define HIPEMU_COMPUTE_UNITS 2          // persistent kernels size their grids by this: small, the emulation runs one lane at a time
#endif

static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "emulated HIP runtime: error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidValue; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = HIPEMU_COMPUTE_UNITS; return hipSuccess; }
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) { *p = (T*)std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <class T> static inline hipError_t hipHostMalloc(T** p, size_t n, unsigned = 0) { *p = (T*)std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
static inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) std::memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { if (n) std::memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { if (n) std::memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { if (n) std::memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)std::malloc(8); return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { std::free(s); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)std::malloc(8); return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)std::malloc(8); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { std::free(e); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }
static inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
// no stream capture: launches have already run when the call returns
static inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorUnknown; }
static inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return hipErrorUnknown; }
static inline hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t, void*, void*, size_t) { *e = nullptr; return hipErrorUnknown; }
static inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorUnknown; }
static inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
