// rtw_host_emul.h -- DEBUG AID ONLY (tests/): lets the device code of rtw_device.hip compile as plain
// host C++ so that AddressSanitizer / UBSan can run over the kernels' memory accesses on the CPU
// (GPU sanitizers are not available on the pool).  Never linked into librtwin.so.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#define __device__
#define __global__
#define __forceinline__ inline
#define __restrict__
#define __shared__ static thread_local
#define __launch_bounds__(...)
struct float4 { float x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { float4 r = { x, y, z, w }; return r; }
struct dim3e { unsigned x, y, z; };
static thread_local dim3e blockIdx, threadIdx, blockDim, gridDim;
static inline void __syncthreads() {}
static inline float __uint_as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline uint32_t __float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline int __float_as_int(float f) { int u; std::memcpy(&u, &f, 4); return u; }
static inline float __int_as_float(int u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }
static inline float __builtin_amdgcn_logf(float x) { return log2f(x); }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p += v; return o; }
static inline uint32_t atomicAdd(uint32_t* p, uint32_t v) { uint32_t o = *p; *p += v; return o; }
typedef void* hipStream_t;
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
