// Issue rate of the float64 VALU instructions the exact tier is made of, on one MI355X: a wave-level
// instruction count per second for v_fma_f64 / v_mul_f64 / v_add_f64 / v_cvt_f64_f32 / v_rndne_f64 (and
// v_fma_f32 as the yardstick), 8 independent chains per lane, 8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/f64_rate microbench/f64_rate.hip && /tmp/f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ __launch_bounds__(256) void k_rate(double *out, int iters, double seed)
{
    double a[8];
    float f[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[k] = seed + k + threadIdx.x * 1e-3; f[k] = (float)a[k]; }
    const double m = 1.0000001, c = 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
                if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[k]) : "v"(m));
                if (OP == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[k]) : "v"(c));
                if (OP == 3) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[k]) : "v"(f[k]));
                if (OP == 4) asm volatile("v_rndne_f64 %0, %0" : "+v"(a[k]));
                if (OP == 5) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[k]) : "v"((float)m), "v"((float)c));
                if (OP == 6) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(f[k]) : "v"(a[k]));
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += a[k] + f[k];
    if (s == 12345.678) out[0] = s;
}

template <int OP>
void run(const char *name, double *d, int waves_per_simd = 8)
{
    const int iters = 4096, grid = 256 * waves_per_simd;            // workgroups of 4 waves, one wave per SIMD each
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_rate<OP>), dim3(grid), dim3(256), 0, 0, d, 16, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_rate<OP>), dim3(grid), dim3(256), 0, 0, d, iters, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = (double)grid * 4 * iters * 32;           // wave-level instructions issued
    const double per_simd_per_s = wave_instr / (ms * 1e-3) / (256 * 4);
    printf("%-16s %d waves/SIMD %8.3f ms  %7.2f G wave-instr/s per SIMD  = %5.2f cycles per instruction at 2.4 GHz\n", name, waves_per_simd, ms,
           per_simd_per_s * 1e-9, 2.4e9 / per_simd_per_s);
}

int main()
{
    double *d;
    hipMalloc(&d, 64);
    run<5>("v_fma_f32", d);
    run<0>("v_fma_f64", d);
    run<1>("v_mul_f64", d);
    run<2>("v_add_f64", d);
    run<3>("v_cvt_f64_f32", d);
    run<4>("v_rndne_f64", d);
    run<6>("v_cvt_i32_f64", d);
    for (int w : {1, 2, 3, 4, 6}) run<0>("v_fma_f64", d, w);
    for (int w : {1, 2, 4}) run<5>("v_fma_f32", d, w);
    return 0;
}
