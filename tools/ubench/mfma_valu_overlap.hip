// Do the matrix cores and the vector ALU of a SIMD work at the same time on gfx950?  A workgroup of 8 waves (2 per SIMD);
// mode 0: every wave runs a chain of independent v_pk_fma_f32; mode 1: every wave runs v_mfma_f32_16x16x4_f32 (two
// accumulators); mode 2: the even waves run the VALU chain and the odd waves the MFMA chain, each the same length as in
// modes 0 / 1.  If the units overlap, mode 2 takes max(mode 0, mode 1) per wave pair; if MFMA issue blocks the VALU, the sum.
// Prints cycles per iteration of each mode (s_memtime of wave 0 of block 0, the grid fills the chip: 256 CUs x 1 workgroup).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <bool SMALL>
__global__ __launch_bounds__(512) void k(float *d, int n, int mode)
{
    const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
    const bool valu = (mode == 0) || (mode == 2 && (wv & 1) == 0) || (mode == 3 && (wv & 4) == 0);
    f2 v[8];
    for (int i = 0; i < 8; ++i) { v[i] = f2{l * 0.001f + i, 1.0f + l * 0.002f}; }
    const f2 m = {1.0000001f, 0.9999999f}, c = {1e-7f, -1e-7f};
    f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0}, a3 = {0, 0, 0, 0};
    const float a = l * 0.001f, b = 1.0f + l * 0.002f;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (valu)
    {
        for (int s = 0; s < n; ++s)
        {
#pragma unroll
            for (int i = 0; i < 8; ++i) { v[i] = __builtin_elementwise_fma(v[i], m, c); }      // 8 independent v_pk_fma_f32
        }
    }
    else
    {
        for (int s = 0; s < n; ++s)
        {
            if (SMALL)      // eight v_mfma_f32_4x4x1_16B_f32 on four accumulators: the same 2 x 2048 flop... no: 8 x 512 flops
            {
                a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(b, a, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(b, a, a3, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(b, a, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_4x4x1f32(b, a, a3, 0, 0, 0);
            }
            else
            {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, a1, 0, 0, 0);
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    const long long t2 = __builtin_amdgcn_s_memtime();
    float r = a0[0] + a1[1] + a2[2] + a3[3];
    for (int i = 0; i < 8; ++i) { r += v[i].x + v[i].y; }
    d[blockIdx.x * 512 + threadIdx.x] = r;
    if (blockIdx.x == 0 && l == 0) { d[gridDim.x * 512 + 2 * wv] = (float)(t1 - t0) / (float)n; d[gridDim.x * 512 + 2 * wv + 1] = (float)(t2 - t0) / (float)n; }
}

int main()
{
    const int grid = 256, n = 20000;
    float *d; hipMalloc(&d, (grid * 512 + 64) * sizeof(float));
    float h[16];
    const char *names[] = {"VALU only (8 v_pk_fma_f32 per iteration, 8 waves)", "MFMA only (2 v_mfma_f32_16x16x4_f32 per iteration, 8 waves)",
                           "even waves VALU, odd waves MFMA (each SIMD: one of each)", "waves 0-3 VALU, waves 4-7 MFMA (each SIMD: one of each)"};
    for (int mode = 0; mode < 8; ++mode)
    {
        if (mode >= 4 && (mode & 3) != 1 && (mode & 3) != 3) { continue; }
        if (mode >= 4) { printf("the same with eight v_mfma_f32_4x4x1_16B_f32 (four accumulators) per iteration instead of two 16x16x4:\n"); }
        for (int rep = 0; rep < 2; ++rep)
        {
            if (mode >= 4) { hipLaunchKernelGGL(k<true>, dim3(grid), dim3(512), 0, 0, d, n, mode & 3); }
            else { hipLaunchKernelGGL(k<false>, dim3(grid), dim3(512), 0, 0, d, n, mode); }
        }
        hipDeviceSynchronize();
        hipMemcpy(h, d + grid * 512, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d  %s\n   s_memtime ticks (100 MHz) per iteration, per wave (own loop | until the workgroup's last wave is done):", mode, names[mode & 3]);
        for (int w = 0; w < 8; ++w) { printf("  %.3f|%.3f", h[2 * w], h[2 * w + 1]); }
        printf("\n");
    }
    return 0;
}
