// Micro-benchmark: the memory traffic of the headline spectrogram workload with no arithmetic -- each 256-thread
// workgroup reads 64 frames' worth of audio (10,240 floats) and writes 64 x 257 floats, dword per lane as the kernel does.
// Gives the memory-only floor for this read:write mix (0.64 GB : 1.03 GB per launch).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ __launch_bounds__(256) void rw(const float *x, float *y, int in_per_wg, int out_per_wg, int mode)
{
    const float *xi = x + (size_t)blockIdx.x * in_per_wg;
    float *yo = y + (size_t)blockIdx.x * out_per_wg;
    float acc = 0.0f;
    if (mode & 1) { for (int i = threadIdx.x; i < in_per_wg; i += 256) { acc += xi[i]; } }
    if (mode & 2) { for (int i = threadIdx.x; i < out_per_wg; i += 256) { yo[i] = acc + (float)i; } }
    else if (acc == 12345.678f) { yo[threadIdx.x] = acc; }
}
// the same bytes with 16-byte accesses, four per thread in flight: what the memory system does when the kernel shape is ideal
__global__ __launch_bounds__(256) void rw4(const float4 *x, float4 *y, size_t nin4, size_t nout4, int mode)
{
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nth = (size_t)gridDim.x * 256;
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    if (mode & 1)
    {
        for (size_t i = tid; i < nin4; i += 4 * nth)
        {
            float4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { const size_t k = i + j * nth; v[j] = (k < nin4) ? x[k] : acc; }
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
        }
    }
    if (mode & 2) { for (size_t i = tid; i < nout4; i += nth) { y[i] = acc; } }
    else if (acc.x == 12345.678f) { y[tid] = acc; }
}

int main()
{
    const int nwg = 15625, in_per_wg = 10240, out_per_wg = 64 * 257;
    float *x, *y;
    CHECK(hipMalloc(&x, (size_t)nwg * in_per_wg * 4)); CHECK(hipMalloc(&y, (size_t)nwg * out_per_wg * 4));
    CHECK(hipMemset(x, 0, (size_t)nwg * in_per_wg * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int mode = 1; mode <= 3; ++mode)
    {
        float best = 1e9f;
        for (int rep = 0; rep < 10; ++rep)
        {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(rw, dim3(nwg), dim3(256), 0, 0, x, y, in_per_wg, out_per_wg, mode);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) { best = ms; }
        }
        // sustained: 20 launches back to back (an isolated launch ends with part of its writes still in L2 / MALL)
        CHECK(hipEventRecord(e0));
        for (int rep = 0; rep < 20; ++rep) { hipLaunchKernelGGL(rw, dim3(nwg), dim3(256), 0, 0, x, y, in_per_wg, out_per_wg, mode); }
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float sus; CHECK(hipEventElapsedTime(&sus, e0, e1)); sus /= 20.0f;
        const double bytes = ((mode & 1) ? (double)nwg * in_per_wg * 4 : 0.0) + ((mode & 2) ? (double)nwg * out_per_wg * 4 : 0.0);
        printf("%s: isolated %.3f ms, %.2f TB/s; back to back %.3f ms, %.2f TB/s\n", mode == 1 ? "read 0.64 GB" : mode == 2 ? "write 1.03 GB" : "read 0.64 GB + write 1.03 GB",
               best, bytes / (best * 1e-3) / 1e12, sus, bytes / (sus * 1e-3) / 1e12);
    }
    const size_t nin4 = (size_t)nwg * in_per_wg / 4, nout4 = (size_t)nwg * out_per_wg / 4;
    for (int mode = 1; mode <= 3; ++mode)
    {
        float best = 1e9f;
        for (int rep = 0; rep < 10; ++rep)
        {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(rw4, dim3(256 * 16), dim3(256), 0, 0, (const float4 *)x, (float4 *)y, nin4, nout4, mode);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) { best = ms; }
        }
        CHECK(hipEventRecord(e0));
        for (int rep = 0; rep < 20; ++rep) { hipLaunchKernelGGL(rw4, dim3(256 * 16), dim3(256), 0, 0, (const float4 *)x, (float4 *)y, nin4, nout4, mode); }
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float sus; CHECK(hipEventElapsedTime(&sus, e0, e1)); sus /= 20.0f;
        const double bytes = ((mode & 1) ? (double)nin4 * 16 : 0.0) + ((mode & 2) ? (double)nout4 * 16 : 0.0);
        printf("16-byte grid-stride, %s: isolated %.3f ms, %.2f TB/s; back to back %.3f ms, %.2f TB/s\n", mode == 1 ? "read 0.64 GB" : mode == 2 ? "write 1.03 GB" : "read + write",
               best, bytes / (best * 1e-3) / 1e12, sus, bytes / (sus * 1e-3) / 1e12);
    }
    return 0;
}
