// What v_permlane32_swap_b32 / v_permlane16_swap_b32 (gfx950) do to a pair of registers: prints, per lane, the two results
// for inputs a = lane, b = 100 + lane.   hipcc --offload-arch=gfx950 -O3 tools/ubench/permlane_swap.hip -o /tmp/pls && /tmp/pls
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned *p)
{
    const unsigned a = threadIdx.x, b = 100u + threadIdx.x;
    const u2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    const u2 q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    p[threadIdx.x] = r.x; p[64 + threadIdx.x] = r.y; p[128 + threadIdx.x] = q.x; p[192 + threadIdx.x] = q.y;
}
int main()
{
    unsigned *d, h[256];
    if (hipMalloc((void **)&d, sizeof(h)) != hipSuccess) { return 1; }
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) { return 1; }
    const char *names[4] = {"swap32 .x", "swap32 .y", "swap16 .x", "swap16 .y"};
    for (int r = 0; r < 4; ++r) { printf("%s:", names[r]); for (int i = 0; i < 64; i += 8) { printf(" [%d]=%u", i, h[64 * r + i]); } printf("\n"); }
    return 0;
}
