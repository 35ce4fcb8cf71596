// TEST INFRASTRUCTURE: does the harness catch what it is there to catch?  Four tiny kernels against the HIP stand-in of shim/:
//   ok          a wave-level prefix sum through __ballot / __shfl / LDS + __syncthreads, checked on the host
//   oob_global  writes one element past a hipMalloc'd buffer        -> AddressSanitizer must abort the run
//   oob_lds     reads one word past the launch's dynamic LDS         -> AddressSanitizer must abort the run
//   divergent   half of a wave calls one __ballot more than the rest -> the scheduler must abort ("divergent wave operation")
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ void scan_kernel(const uint32_t* in, uint32_t* out, int n)
{
    HIP_DYNAMIC_SHARED(uint32_t, part);
    const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x), lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    uint32_t v = t < n ? in[t] : 0u;
    for (int o = 1; o < 64; o <<= 1) { const uint32_t u = (uint32_t)__shfl((int)v, lane - o); if (lane >= o) v += u; }
    if (lane == 63) part[wave] = v;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; w++) base += part[w];
    const unsigned long long odd = __ballot((v & 1u) != 0u);
    if (t < n) out[t] = base + v + (uint32_t)__popcll(odd) * 0u;
}
__global__ void oob_global_kernel(uint32_t* out, int n) { out[threadIdx.x == 5 ? n : (int)threadIdx.x] = 1u; }
__global__ void oob_lds_kernel(uint32_t* out, int words)
{
    HIP_DYNAMIC_SHARED(uint32_t, lds);
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    out[threadIdx.x] = lds[threadIdx.x == 7 ? words : (int)threadIdx.x];
}
__global__ void divergent_kernel(uint32_t* out)
{
    unsigned long long m = 0;
    if ((threadIdx.x & 63u) < 32u) m = __ballot(true);
    m += __ballot(threadIdx.x > 3u);        // the other half arrives here first: two different operations wait in one wave
    out[threadIdx.x] = (uint32_t)m;
}

int main(int argc, char** argv)
{
    const char* mode = argc > 1 ? argv[1] : "ok";
    const int n = 256;
    uint32_t *in = nullptr, *out = nullptr;
    if (hipMalloc(&in, n * 4) != hipSuccess || hipMalloc(&out, n * 4) != hipSuccess) return 2;
    for (int i = 0; i < n; i++) in[i] = (uint32_t)(i * 7 + 1);
    if (!std::strcmp(mode, "ok")) {
        hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(256), 4 * sizeof(uint32_t), nullptr, in, out, n);
        uint32_t s = 0;
        for (int i = 0; i < n; i++) { s += in[i]; if (out[i] != s) { std::printf("selftest: scan differs at %d\n", i); return 1; } }
        std::printf("selftest ok\n");
        hipFree(in); hipFree(out);
        return 0;
    }
    if (!std::strcmp(mode, "oob_global")) hipLaunchKernelGGL(oob_global_kernel, dim3(1), dim3(64), 0, nullptr, out, n);
    else if (!std::strcmp(mode, "oob_lds")) hipLaunchKernelGGL(oob_lds_kernel, dim3(1), dim3(64), 64 * sizeof(uint32_t), nullptr, out, 64);
    else if (!std::strcmp(mode, "divergent")) hipLaunchKernelGGL(divergent_kernel, dim3(1), dim3(64), 0, nullptr, out);
    std::printf("selftest: %s went through unnoticed\n", mode);
    return 0;
}
