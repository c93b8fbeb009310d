// Micro-benchmark: f32 VALU issue rate on gfx950, scalar v_fma_f32 vs packed v_pk_fma_f32, at 1..8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define ITERS 4096
__global__ void k_scalar(float *out, float a, float b)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < ITERS; ++i)
    {
        x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
        x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ void k_packed(float *out, float a, float b)
{
    f2 av = {a, a}, bv = {b, b};
    f2 x0 = {(float)threadIdx.x, 1.f}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
    for (int i = 0; i < ITERS; ++i)
    {
        x0 = __builtin_elementwise_fma(x0, av, bv); x1 = __builtin_elementwise_fma(x1, av, bv);
        x2 = __builtin_elementwise_fma(x2, av, bv); x3 = __builtin_elementwise_fma(x3, av, bv);
    }
    f2 s = x0 + x1 + x2 + x3;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
int main()
{
    float *d; (void)hipMalloc(&d, 256 * 8 * 256 * sizeof(float) * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int wps = 1; wps <= 8; wps *= 2)
    {
        for (int packed = 0; packed < 2; ++packed)
        {
            const int blocks = 256 * wps;   // 256-thread blocks: 1 wave per SIMD per block
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep)
            {
                (void)hipEventRecord(e0);
                if (packed) { hipLaunchKernelGGL(k_packed, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f); }
                else { hipLaunchKernelGGL(k_scalar, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f); }
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) { best = ms; }
            }
            const double fma_lane_ops = (double)blocks * 256 * ITERS * 8;     // both kernels do 8 f32 fma per lane per iteration
            const double instr_per_wave = (double)ITERS * (packed ? 4 : 8);
            printf("waves/SIMD %d %s: %.3f ms  %.1f TFLOP/s  %.2f cycles/instr/SIMD at 2.4 GHz\n", wps, packed ? "v_pk_fma_f32" : "v_fma_f32   ", best,
                   2.0 * fma_lane_ops / (best * 1e-3) / 1e12, best * 1e-3 * 2.4e9 / (instr_per_wave * wps));
        }
    }
    return 0;
}
