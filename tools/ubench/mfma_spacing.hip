// Follow-up to mfma_valu_overlap.hip: is the VALU of a SIMD blocked by another wave's matrix-core work as such, or only while
// that wave has its NEXT v_mfma waiting at the issue stage for the busy matrix pipe?  Waves 0-3 (one per SIMD) run nv iterations
// of 8 independent v_pk_fma_f32; waves 4-7 (one per SIMD) run nm iterations of { v_mfma_f32_16x16x4_f32 ; K x s_nop 7 } on ACC
// accumulators in turn.  With K = 0 the MFMA wave keeps the pipe oversubscribed; with K >= 4 (32+ idle cycles of that wave
// between its MFMAs) the pipe is free when the next MFMA arrives.  Printed per configuration: cycles per iteration of the VALU
// wave alone, of the MFMA wave alone, and of both when they share the SIMDs.  mode 2: ONE wave per SIMD runs
// { v_mfma ; J x v_pk_fma_f32 } -- does a wave's own vector work issue under its own MFMA?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int K, int ACC, int J>
__global__ __launch_bounds__(512) void k(float *d, int nv, int nm, int who)     // who: 1 = VALU waves only, 2 = MFMA waves only, 3 = both, 4 = mixed in one wave (waves 0-3)
{
    const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
    f2 v[8];
    for (int i = 0; i < 8; ++i) { v[i] = f2{l * 0.001f + i, 1.0f + l * 0.002f}; }
    const f2 m = {1.0000001f, 0.9999999f}, c = {1e-7f, -1e-7f};
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const float a = l * 0.001f, b = 1.0f + l * 0.002f;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    int iters = 1;
    if (who == 4)
    {
        if (wv < 4)
        {
            iters = nm;
            for (int s = 0; s < nm; ++s)
            {
#pragma unroll
                for (int q = 0; q < ACC; ++q)
                {
                    acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[q], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < J; ++i) { v[i] = __builtin_elementwise_fma(v[i], m, c); }
                }
            }
        }
    }
    else if (wv < 4)
    {
        if (who & 1)
        {
            iters = nv;
            for (int s = 0; s < nv; ++s)
            {
#pragma unroll
                for (int i = 0; i < 8; ++i) { v[i] = __builtin_elementwise_fma(v[i], m, c); }
            }
        }
    }
    else if (who & 2)
    {
        iters = nm;
        for (int s = 0; s < nm; ++s)
        {
#pragma unroll
            for (int q = 0; q < ACC; ++q)
            {
                acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[q], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < K; ++i) { asm volatile("s_nop 7"); }
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float r = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    for (int i = 0; i < 8; ++i) { r += v[i].x + v[i].y; }
    d[blockIdx.x * 512 + threadIdx.x] = r;
    if (blockIdx.x == 0 && l == 0) { d[gridDim.x * 512 + wv] = (float)(t1 - t0) / (float)iters; }
}

template <int K, int ACC>
void run(float *d, int grid)
{
    const int nv = 20000, nm = 20000;
    float h[8];
    float res[3][2];
    for (int who = 1; who <= 3; ++who)
    {
        // with both kinds on the chip the MFMA waves run 8 x as long, so the VALU waves never run alone
        const int nmm = (who == 3) ? 8 * nm : nm;
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<K, ACC, 0>), dim3(grid), dim3(512), 0, 0, d, nv, nmm, who); }
        hipDeviceSynchronize();
        hipMemcpy(h, d + grid * 512, sizeof(h), hipMemcpyDeviceToHost);
        res[who - 1][0] = h[0]; res[who - 1][1] = h[4];
    }
    printf("K=%d s_nop 7 after each MFMA, %d accumulators: VALU wave alone %.1f cycles/iter (8 v_pk_fma), MFMA wave alone %.1f cycles/iter (%d MFMA); sharing a SIMD: VALU wave %.1f, MFMA wave %.1f\n",
           K, ACC, res[0][0], res[1][1], ACC, res[2][0], res[2][1]);
}

template <int J>
void run_mixed(float *d, int grid)
{
    float h[8];
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<0, 2, J>), dim3(grid), dim3(512), 0, 0, d, 0, 20000, 4); }
    hipDeviceSynchronize();
    hipMemcpy(h, d + grid * 512, sizeof(h), hipMemcpyDeviceToHost);
    printf("one wave per SIMD, 2 x { v_mfma ; %d x v_pk_fma } per iteration: %.1f cycles/iter\n", J, h[0]);
}

int main()
{
    const int grid = 256;
    float *d; hipMalloc(&d, (grid * 512 + 64) * sizeof(float));
    run<0, 2>(d, grid); run<1, 2>(d, grid); run<2, 2>(d, grid); run<3, 2>(d, grid); run<4, 2>(d, grid); run<5, 2>(d, grid); run<6, 2>(d, grid);
    run<0, 4>(d, grid); run<4, 4>(d, grid);
    run<0, 1>(d, grid); run<4, 1>(d, grid); run<5, 1>(d, grid);
    run_mixed<0>(d, grid); run_mixed<2>(d, grid); run_mixed<4>(d, grid); run_mixed<6>(d, grid); run_mixed<8>(d, grid);
    return 0;
}
