// Layout check of v_mfma_f32_4x4x1_16B_f32 on gfx950: 16 blocks, D_b[i][j] += A_b[i] * B_b[j].
// Hypothesis: lane l supplies A_b[i] with b = l / 4, i = l % 4, and B_b[j] with b = l / 4, j = l % 4;
// D_b[i][j] is VGPR i of lane 4 b + j.  Prints the number of mismatches (0 = hypothesis holds) and a timing of a chain.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *a, const float *b, float *d)
{
    const int l = threadIdx.x;
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[64 + l], b[64 + l], acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) { d[i * 64 + l] = acc[i]; }
}
template <int NC>
__global__ void tc(float *d, int n)
{
    const int l = threadIdx.x;
    f32x4 acc[NC];
    for (int c = 0; c < NC; ++c) { acc[c] = f32x4{0, 0, 0, 0}; }
    float a = l * 0.001f, b = 1.0f + l * 0.002f;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < n; ++s)
    {
#pragma unroll
        for (int c = 0; c < NC; ++c) { acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[c], 0, 0, 0); }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0; for (int c = 0; c < NC; ++c) { r += acc[c][0]; }
    d[l] = r;
    if (l == 0 && blockIdx.x == 0) { d[64] = (float)(t1 - t0) / (float)(n * NC); }
}
__global__ void t(float *d, int n)
{
    const int l = threadIdx.x;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    float a = l * 0.001f, b = 1.0f + l * 0.002f;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < n; ++s)
    {
        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(b, a, acc1, 0, 0, 0);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    d[l] = acc0[0] + acc1[1];
    if (l == 0 && blockIdx.x == 0) { d[64] = (float)(t1 - t0); }
}
int main()
{
    float ha[128], hb[128], hd[256], *da, *db, *dd;
    for (int i = 0; i < 128; ++i) { ha[i] = 1.0f + i; hb[i] = 0.5f + 3.0f * i; }
    hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dd, sizeof(hd));
    hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd);
    hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int blk = 0; blk < 16; ++blk) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j)
    {
        const float want = ha[4 * blk + i] * hb[4 * blk + j] + ha[64 + 4 * blk + i] * hb[64 + 4 * blk + j];
        const float got = hd[i * 64 + 4 * blk + j];
        if (want != got) { if (bad < 5) printf("mismatch blk %d i %d j %d want %g got %g\n", blk, i, j, want, got); ++bad; }
    }
    printf("layout mismatches: %d\n", bad);
    hipLaunchKernelGGL(t, dim3(1), dim3(64), 0, 0, dd, 1000);
    hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
    printf("cycles per 4x4x1 mfma (one wave, two chains): %.2f\n", hd[64] / 2000.0);
    hipLaunchKernelGGL(t, dim3(1), dim3(256), 0, 0, dd, 1000);
    hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
    printf("cycles per 4x4x1 mfma (4 waves on 4 SIMDs): %.2f\n", hd[64] / 2000.0);
    hipLaunchKernelGGL(tc<1>, dim3(1), dim3(64), 0, 0, dd, 1000); hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost); printf("1 chain: %.2f cycles per mfma\n", hd[64]);
    hipLaunchKernelGGL(tc<2>, dim3(1), dim3(64), 0, 0, dd, 1000); hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost); printf("2 chains: %.2f\n", hd[64]);
    hipLaunchKernelGGL(tc<4>, dim3(1), dim3(64), 0, 0, dd, 1000); hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost); printf("4 chains: %.2f\n", hd[64]);
    hipLaunchKernelGGL(tc<8>, dim3(1), dim3(64), 0, 0, dd, 1000); hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost); printf("8 chains: %.2f\n", hd[64]);
    hipLaunchKernelGGL(tc<2>, dim3(1), dim3(512), 0, 0, dd, 1000); hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost); printf("2 chains, 8 waves (2 per SIMD): %.2f\n", hd[64]);
    return 0;
}
