// microbench/membench.hip -- streaming-bandwidth ceilings on this GPU for the fused kernel's traffic mix.
// Not part of the product: it calibrates what "HBM roofline" is reachable for a 2:1 read:write
// stream (the forward kernel reads 256 B and writes 128 B per block) and calibrates rocprofv3's
// FETCH_SIZE for the kernel's access pattern (16 B per lane at a 32 B lane stride).
// Build: hipcc --offload-arch=gfx950 -O3 -o microbench/membench microbench/membench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// plain float4 copy: n4 float4 elements
__global__ void k_copy(const float4* __restrict__ in, float4* __restrict__ out, size_t n4)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) out[i] = in[i];
}
// 2:1 mix: read two float4, write one (sum) -- coalesced 16 B per lane
__global__ void k_mix21(const float4* __restrict__ in, float4* __restrict__ out, size_t n4out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4out) {
        float4 a = in[i], b = in[i + n4out];
        out[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k_mix21_nt(const float4* __restrict__ in_, float4* __restrict__ out_, size_t n4out)
{
    const f32x4* in = reinterpret_cast<const f32x4*>(in_);
    f32x4* out = reinterpret_cast<f32x4*>(out_);
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4out) {
        f32x4 a = __builtin_nontemporal_load(&in[i]), b = __builtin_nontemporal_load(&in[i + n4out]);
        __builtin_nontemporal_store(a + b, &out[i]);
    }
}
// read-only with the fused kernel's pattern: lane-per-block rows, 2 x 16 B per row at 32 B lane stride
__global__ __launch_bounds__(64) void k_read_blocks(const float* __restrict__ in, size_t pitch, int wb, int nblk, float* __restrict__ sink)
{
    int g = blockIdx.x * 64 + threadIdx.x;
    if (g >= nblk) return;
    int by = g / wb, bx = g - by * wb;
    const float* src = in + (size_t)by * 8 * pitch + (size_t)bx * 8;
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const float4* row = reinterpret_cast<const float4*>(src + (size_t)r * pitch);
        float4 lo = row[0], hi = row[1];
        acc += lo.x + lo.y + lo.z + lo.w + hi.x + hi.y + hi.z + hi.w;
    }
    if (acc == 123456.789f) sink[g] = acc;   // never true; keeps the loads alive
}
__global__ void k_read(const float4* __restrict__ in, size_t n4, float* __restrict__ sink)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) { float4 a = in[i]; if (a.x + a.y + a.z + a.w == 123456.789f) sink[0] = a.x; }
}
__global__ void k_write(float4* __restrict__ out, size_t n4)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) out[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

template <typename F> double time_ms(F f, int iters = 20)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    std::vector<float> t;
    for (int i = 0; i < iters; ++i) {
        CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main(int argc, char** argv)
{
    const int H = 4096 * 16, W = 4096;                 // same footprint as bench.py's default step
    const size_t nin = (size_t)H * W, n4in = nin / 4;  // 1 GiB of fp32
    const size_t n4out = n4in / 2;                     // 0.5 GiB
    float *in, *out, *sink;
    CK(hipMalloc(&in, nin * 4)); CK(hipMalloc(&out, nin * 4)); CK(hipMalloc(&sink, nin / 16));
    CK(hipMemset(in, 0, nin * 4)); CK(hipMemset(out, 0, nin * 4));
    const int nblk = (H / 8) * (W / 8), wb = W / 8;
    double ms;
    ms = time_ms([&] { hipLaunchKernelGGL(k_copy, dim3((n4in + 255) / 256), dim3(256), 0, 0, (const float4*)in, (float4*)out, n4in); });
    printf("copy float4 1GiB->1GiB      : %.3f ms  %.1f GB/s (read+write)\n", ms, 2.0 * nin * 4 / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_mix21, dim3((n4out + 255) / 256), dim3(256), 0, 0, (const float4*)in, (float4*)out, n4out); });
    printf("mix 2:1 read 1GiB write .5GiB: %.3f ms  %.1f GB/s (read+write)\n", ms, 1.5 * nin * 4 / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_mix21_nt, dim3((n4out + 255) / 256), dim3(256), 0, 0, (const float4*)in, (float4*)out, n4out); });
    printf("mix 2:1 nontemporal          : %.3f ms  %.1f GB/s (read+write)\n", ms, 1.5 * nin * 4 / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_read, dim3((n4in + 255) / 256), dim3(256), 0, 0, (const float4*)in, n4in, sink); });
    printf("read-only float4 1GiB        : %.3f ms  %.1f GB/s\n", ms, 1.0 * nin * 4 / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_read_blocks, dim3((nblk + 63) / 64), dim3(64), 0, 0, in, (size_t)W, wb, nblk, sink); });
    printf("read-only block pattern 1GiB : %.3f ms  %.1f GB/s\n", ms, 1.0 * nin * 4 / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_write, dim3((n4out + 255) / 256), dim3(256), 0, 0, (float4*)out, n4out); });
    printf("write-only float4 .5GiB      : %.3f ms  %.1f GB/s\n", ms, 0.5 * nin * 4 / ms / 1e6);
    return 0;
}
