// Does the scan kernel's access pattern cost HBM bandwidth?  Each workgroup (256 threads) walks
// tiles of 256 rows x 4 KB: per K step every wave-instruction reads 8 rows x one 128-B line
// (row-major layout: lines 4 KB apart) -- versus the same bytes from a tile-major layout where a
// K step's 256 x 128 B are contiguous.  Loads go to registers and are summed.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <bool TILEMAJOR>
__global__ __launch_bounds__(256, 2) void rd(const float* __restrict__ p, long nrows, int tiles_per_wg, float* out)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x4 acc = {0, 0, 0, 0};
    for (int t = 0; t < tiles_per_wg; t++) {
        long tile = (long)blockIdx.x * tiles_per_wg + t;
        long row0 = tile * 256;
        if (row0 + 256 > nrows) break;
        for (int kt = 0; kt < 32; kt++) {
#pragma unroll
            for (int n = 0; n < 8; n++) {
                int ii = wave + 4 * n;             // 32 instructions of 8 rows
                int r = 8 * ii + (lane >> 3), s = lane & 7;
                const float* src;
                if (TILEMAJOR) src = p + (row0 * 1024) + (long)kt * (256 * 32) + r * 32 + s * 4;
                else src = p + (row0 + r) * 1024 + kt * 32 + s * 4;
                acc += *(const f32x4*)src;
            }
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}
int main(int argc, char** argv)
{
    double gb = argc > 1 ? atof(argv[1]) : 41.0;
    long nrows = (long)(gb * 1e9 / 4096) / 256 * 256;
    size_t bytes = (size_t)nrows * 4096;
    float *p, *o;
    if (hipMalloc(&p, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&o, 4);
    hipMemset(p, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    long ntiles = nrows / 256;
    for (int wgs : {512, 1024, 2048}) {
        int tpw = (int)((ntiles + wgs - 1) / wgs);
        for (int mode = 0; mode < 2; mode++) {
            float best = 1e9;
            for (int r = 0; r < 5; r++) {
                hipEventRecord(e0);
                if (mode) hipLaunchKernelGGL(rd<true>, dim3(wgs), dim3(256), 0, 0, p, nrows, tpw, o);
                else hipLaunchKernelGGL(rd<false>, dim3(wgs), dim3(256), 0, 0, p, nrows, tpw, o);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            printf("%.1f GB wgs=%d %s: %.3f ms  %.1f GB/s\n", gb, wgs, mode ? "tile-major" : "row-major ", best, bytes / best / 1e6);
        }
    }
    return 0;
}
