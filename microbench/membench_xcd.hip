// microbench/membench_xcd.hip -- streaming ceilings with and without the XCD-private workgroup order, at the
// bench's working-set sizes.  2:1 nontemporal read:write mix (the forward kernel's traffic), read-only, write-only, 1:2 mix (the inverse's).
// Build: hipcc --offload-arch=gfx950 -O3 -o microbench/membench_xcd microbench/membench_xcd.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ size_t wg_index(int logr)
{
    const unsigned i = blockIdx.x;
    if (logr < 0) return i;
    const unsigned j = i >> 3, x = i & 7u;
    return ((size_t)((j >> logr) * 8 + x) << logr) + (j & ((1u << logr) - 1));
}
__global__ __launch_bounds__(256) void k_mix(const f32x4 *__restrict__ in, f32x4 *__restrict__ out, size_t n4out, int logr)
{
    const size_t i = wg_index(logr) * 256 + threadIdx.x;
    if (i < n4out) {
        const f32x4 a = __builtin_nontemporal_load(&in[2 * (i - threadIdx.x) + threadIdx.x]);
        const f32x4 b = __builtin_nontemporal_load(&in[2 * (i - threadIdx.x) + 256 + threadIdx.x]);
        __builtin_nontemporal_store(a + b, &out[i]);
    }
}
// the inverse kernel's traffic: one 16-byte read, two 16-byte writes per thread (1:2)
__global__ __launch_bounds__(256) void k_mix12(const f32x4 *__restrict__ in, f32x4 *__restrict__ out, size_t n4in, int logr)
{
    const size_t i = wg_index(logr) * 256 + threadIdx.x;
    if (i < n4in) {
        const f32x4 a = __builtin_nontemporal_load(&in[i]);
        __builtin_nontemporal_store(a, &out[2 * (i - threadIdx.x) + threadIdx.x]);
        __builtin_nontemporal_store(a + a, &out[2 * (i - threadIdx.x) + 256 + threadIdx.x]);
    }
}
__global__ __launch_bounds__(256) void k_rd(const f32x4 *__restrict__ in, size_t n4, float *sink, int logr)
{
    const size_t i = wg_index(logr) * 256 + threadIdx.x;
    if (i < n4) { const f32x4 a = __builtin_nontemporal_load(&in[i]); if (a.x + a.y + a.z + a.w == 123456.789f) sink[0] = a.x; }
}
__global__ __launch_bounds__(256) void k_wr(f32x4 *__restrict__ out, size_t n4, int logr)
{
    const size_t i = wg_index(logr) * 256 + threadIdx.x;
    if (i < n4) __builtin_nontemporal_store(f32x4{1.f, 2.f, 3.f, 4.f}, &out[i]);
}
template <typename F> double time_ms(F f, int iters)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) f();
    std::vector<float> t;
    for (int i = 0; i < iters; ++i) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms); }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}
int main()
{
    for (size_t gib_in : {1ull, 16ull, 64ull}) {
        const size_t nin = gib_in << 28;                 // floats
        const size_t n4in = nin / 4, n4out = n4in / 2;
        float *in, *out, *sink;
        CK(hipMalloc(&in, nin * 4)); CK(hipMalloc(&out, nin * 2)); CK(hipMalloc(&sink, 64));
        CK(hipMemset(in, 0, nin * 4)); CK(hipMemset(out, 0, nin * 2));
        const int iters = gib_in >= 16 ? 5 : 20;
        for (int logr : {-1, 9}) {
            const unsigned gm = (unsigned)(((n4out + 255) / 256 + 4095) / 4096 * 4096), gr = (unsigned)(((n4in + 255) / 256 + 4095) / 4096 * 4096);
            double ms = time_ms([&] { hipLaunchKernelGGL(k_mix, dim3(gm), dim3(256), 0, 0, (const f32x4 *)in, (f32x4 *)out, n4out, logr); }, iters);
            printf("%2zu GiB in  %-12s mix 2:1 nt  %.1f GB/s", gib_in, logr < 0 ? "round-robin" : "xcd-private", 1.5 * nin * 4 / ms / 1e6);
            ms = time_ms([&] { hipLaunchKernelGGL(k_rd, dim3(gr), dim3(256), 0, 0, (const f32x4 *)in, n4in, sink, logr); }, iters);
            printf("   read-only %.1f GB/s", 1.0 * nin * 4 / ms / 1e6);
            ms = time_ms([&] { hipLaunchKernelGGL(k_wr, dim3(gm), dim3(256), 0, 0, (f32x4 *)out, n4out, logr); }, iters);
            printf("   write-only %.1f GB/s", 0.5 * nin * 4 / ms / 1e6);
            // 1:2: the half-size buffer is read, the full-size buffer written (the fused inverse: int16 in, fp32 out)
            ms = time_ms([&] { hipLaunchKernelGGL(k_mix12, dim3(gm), dim3(256), 0, 0, (const f32x4 *)out, (f32x4 *)in, n4out, logr); }, iters);
            printf("   mix 1:2 nt %.1f GB/s\n", 1.5 * nin * 4 / ms / 1e6);
        }
        CK(hipFree(in)); CK(hipFree(out)); CK(hipFree(sink));
    }
    return 0;
}
