// Read-bandwidth ceiling probe: grid-stride float4 reads of a buffer of a given size.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void rd(const f32x4* __restrict__ p, size_t n4, float* out, int unroll_dummy)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    f32x4 acc = {0, 0, 0, 0};
    for (; i + 3 * stride < n4; i += 4 * stride) {
        f32x4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
        acc += a + b + c + d;
    }
    for (; i < n4; i += stride) acc += p[i];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}
int main(int argc, char** argv)
{
    for (int a = 1; a < argc; a++) {
        double gb = atof(argv[a]);
        size_t bytes = (size_t)(gb * 1e9) / 16 * 16;
        f32x4* p; float* o;
        if (hipMalloc(&p, bytes) != hipSuccess) { printf("alloc %g GB failed\n", gb); continue; }
        hipMalloc(&o, 4);
        hipMemset(p, 0, bytes);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int grid : {2048, 4096, 8192}) {
            float best = 1e9;
            for (int r = 0; r < 5; r++) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(rd, dim3(grid), dim3(256), 0, 0, p, bytes / 16, o, 0);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            printf("%.1f GB grid=%d: %.3f ms  %.1f GB/s\n", gb, grid, best, bytes / best / 1e6);
        }
        hipFree(p); hipFree(o);
    }
    return 0;
}
