// TEST INFRASTRUCTURE (tests/cpu_emul): the device side of the HIP stand-in.  A kernel launch runs its blocks one after another; the threads
// of a block are fibers on one OS thread, a wave is 64 consecutive fibers.  A cross-lane operation (__ballot, __shfl, readlane, ds_bpermute,
// DPP, wave barrier) is a RENDEZVOUS of the wave's live lanes: every lane deposits its value and yields; when all of them have arrived -- at
// the same source line, or the run aborts with "divergent wave operation", which is itself a check of the kernels' wave-uniform control flow
// -- they go on with each other's values.  __syncthreads is the same for the block.  LDS is a heap block of exactly the size the launch asked
// for, global memory is the host heap: AddressSanitizer sees every access of every kernel.  See hipemu_rt.cpp.
#pragma once
#include "hip_runtime_api.h"
#include <cmath>
#include <cstdio>
#include <functional>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define HIP_DYNAMIC_SHARED(type, var) type* var = (type*)hipemu::dynamic_lds()
#define HIP_SYMBOL(x) x

struct float4 { float x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { float4 r = { x, y, z, w }; return r; }
struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };

namespace hipemu {
extern dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()>& body);
void* dynamic_lds();
// rendezvous of the wave: deposits v, returns the 64 deposited values (valid until this lane's next rendezvous) and the mask of live lanes
const uint64_t* wave_exchange(uint64_t v, const char* what, int line, uint64_t* live);
void block_barrier(int line);
int lane();
}
#define threadIdx hipemu::t_threadIdx
#define blockIdx hipemu::t_blockIdx
#define blockDim hipemu::t_blockDim
#define gridDim hipemu::t_gridDim
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) hipemu::launch((grid), (block), (lds), [=]() { kernel(__VA_ARGS__); })
static inline hipError_t hipMemcpyFromSymbol(void* d, const void* s, size_t n) { std::memcpy(d, s, n); return hipSuccess; }

static inline float __uint_as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline uint32_t __float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline int __float_as_int(float f) { int u; std::memcpy(&u, &f, 4); return u; }
static inline float __int_as_float(int u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned long long wall_clock64() { return 0ull; }
// one fiber runs at a time: plain read-modify-write is atomic
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p += v; return o; }
static inline uint32_t atomicAdd(uint32_t* p, uint32_t v) { uint32_t o = *p; *p += v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p += v; return o; }
static inline unsigned long long atomicMax(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; if (v > o) *p = v; return o; }
static inline uint32_t atomicMax(uint32_t* p, uint32_t v) { uint32_t o = *p; if (v > o) *p = v; return o; }
static inline uint32_t atomicMin(uint32_t* p, uint32_t v) { uint32_t o = *p; if (v < o) *p = v; return o; }
static inline void __syncthreads(int line = __builtin_LINE()) { hipemu::block_barrier(line); }

namespace hipemu {
template <class T> static inline uint64_t bits_of(T v) { static_assert(sizeof(T) <= 8, "wave operand"); uint64_t u = 0; std::memcpy(&u, &v, sizeof(T)); return u; }
template <class T> static inline T from_bits(uint64_t u) { T v; std::memcpy(&v, &u, sizeof(T)); return v; }
static inline int first_live(uint64_t live) { return __builtin_ffsll((long long)live) - 1; }
}
static inline unsigned long long __ballot(int pred, int line = __builtin_LINE())
{
    uint64_t live; const uint64_t* v = hipemu::wave_exchange(pred ? 1u : 0u, "__ballot", line, &live);
    unsigned long long m = 0;
    for (int l = 0; l < 64; l++) if (((live >> l) & 1u) && v[l]) m |= 1ull << l;
    return m;
}
template <class T> static inline T __shfl(T x, int src, int width = 64, int line = __builtin_LINE())
{
    (void)width;
    uint64_t live; const uint64_t* v = hipemu::wave_exchange(hipemu::bits_of(x), "__shfl", line, &live);
    return hipemu::from_bits<T>(v[src & 63]);
}
template <class T> static inline T __shfl_xor(T x, int mask, int width = 64, int line = __builtin_LINE())
{
    (void)width;
    uint64_t live; const uint64_t* v = hipemu::wave_exchange(hipemu::bits_of(x), "__shfl_xor", line, &live);
    return hipemu::from_bits<T>(v[(hipemu::lane() ^ mask) & 63]);
}
static inline int hipemu_readlane(int x, int l, int line = __builtin_LINE())
{
    uint64_t live; const uint64_t* v = hipemu::wave_exchange(hipemu::bits_of(x), "readlane", line, &live);
    return hipemu::from_bits<int>(v[l & 63]);
}
static inline int hipemu_readfirstlane(int x, int line = __builtin_LINE())
{
    uint64_t live; const uint64_t* v = hipemu::wave_exchange(hipemu::bits_of(x), "readfirstlane", line, &live);
    return hipemu::from_bits<int>(v[hipemu::first_live(live)]);
}
static inline int hipemu_ds_bpermute(int byte_addr, int x, int line = __builtin_LINE())
{
    uint64_t live; const uint64_t* v = hipemu::wave_exchange(hipemu::bits_of(x), "ds_bpermute", line, &live);
    return hipemu::from_bits<int>(v[(byte_addr >> 2) & 63]);
}
static inline int hipemu_mov_dpp(int x, int ctrl, int row_mask, int bank_mask, bool bound_ctrl, int line = __builtin_LINE())
{
    (void)row_mask; (void)bank_mask; (void)bound_ctrl;
    uint64_t live; const uint64_t* v = hipemu::wave_exchange(hipemu::bits_of(x), "mov_dpp", line, &live);
    if (ctrl < 0 || ctrl > 0xFF) { std::fprintf(stderr, "hipemu: DPP control 0x%x is not a quad_perm\n", ctrl); std::abort(); }
    const int l = hipemu::lane();
    return hipemu::from_bits<int>(v[(l & ~3) | ((ctrl >> (2 * (l & 3))) & 3)]);
}
static inline void hipemu_wave_barrier(int line = __builtin_LINE()) { uint64_t live; (void)hipemu::wave_exchange(0, "wave_barrier", line, &live); }
static inline uint32_t hipemu_mbcnt_lo(uint32_t mask, uint32_t add) { const int l = hipemu::lane(); return add + (uint32_t)__builtin_popcount(l >= 32 ? mask : (mask & ((1u << l) - 1u))); }
static inline uint32_t hipemu_mbcnt_hi(uint32_t mask, uint32_t add) { const int l = hipemu::lane(); return add + (l <= 32 ? 0u : (uint32_t)__builtin_popcount(mask & ((1u << (l - 32)) - 1u))); }
#define __builtin_amdgcn_readlane(x, l) hipemu_readlane((x), (l))
#define __builtin_amdgcn_readfirstlane(x) hipemu_readfirstlane((x))
#define __builtin_amdgcn_ds_bpermute(a, x) hipemu_ds_bpermute((a), (x))
#define __builtin_amdgcn_mov_dpp(x, c, r, b, bc) hipemu_mov_dpp((x), (c), (r), (b), (bc))
#define __builtin_amdgcn_wave_barrier() hipemu_wave_barrier()
#define __builtin_amdgcn_mbcnt_lo(m, a) hipemu_mbcnt_lo((m), (a))
#define __builtin_amdgcn_mbcnt_hi(m, a) hipemu_mbcnt_hi((m), (a))
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_amdgcn_logf(x) log2f(x)
