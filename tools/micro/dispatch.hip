// How fast does the device start workgroups?  N one-wave workgroups against N / 4 four-wave workgroups doing the same tiny
// amount of work per wave (one load, a few dependent shuffles, one store) -- the shape of the batch regime's one-wave-per-query
// selections (14433 workgroups of 64 threads at CATH size).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void work(const unsigned *in, unsigned *out, int nwaves, int spin)
{
    const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= nwaves) return;
    unsigned v = in[(size_t)w * 64 + lane];
    for (int i = 0; i < spin; i++) v = v * 1664525u + (unsigned)__shfl_xor((int)v, 1 + (i & 31), 64);
    out[(size_t)w * 64 + lane] = v;
}
int main()
{
    const int n = 14433;
    unsigned *in, *out;
    CK(hipMalloc(&in, (size_t)n * 256)); CK(hipMalloc(&out, (size_t)n * 256));
    CK(hipMemset(in, 1, (size_t)n * 256));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int spin : {0, 50, 400, 2000})
        for (int nt : {64, 256, 1024}) {
            const int per = nt / 64, grid = (n + per - 1) / per;
            float best = 1e9f;
            for (int rep = 0; rep < 6; rep++) {
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(work, dim3(grid), dim3(nt), 0, 0, in, out, n, spin);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            printf("%d waves, %4d dependent steps each, %4d threads per workgroup (%5d workgroups): %.1f us\n", n, spin, nt, grid, 1e3f * best);
        }
    return 0;
}
