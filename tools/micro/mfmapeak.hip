// fp32 MFMA ceiling probe: every wave issues back-to-back fp32 MFMAs on independent accumulators
// (operands in registers, no memory traffic) -- what the matrix pipes of THIS box sustain, to set
// beside the 157.3 TFLOP/s datasheet figure.  Two shapes (32x32x2, 16x16x4) x two data sets
// (constant-ish, random): the chip lowers its clock under load and random operands toggle more.
// The in-kernel clock is d(s_memtime)/d(s_memrealtime) x 100 MHz.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ inline float rnd(unsigned s)
{
    s = s * 747796405u + 2891336453u;
    s = ((s >> ((s >> 28) + 4)) ^ s) * 277803737u;
    s = (s >> 22) ^ s;
    return (float)(s & 0xFFFFFF) / 8388608.0f - 1.0f;
}

template <int SHAPE>
__global__ __launch_bounds__(256) void mf(float* out, int iters, int random, unsigned long long* clk)
{
    float x[8], y[8];
    const unsigned gid = blockIdx.x * 256 + threadIdx.x;
    for (int i = 0; i < 8; i++) {
        x[i] = random ? rnd(gid * 16 + i) : 1.0f;
        y[i] = random ? rnd(gid * 16 + 8 + i) : 0.5f;
    }
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    if constexpr (SHAPE == 32) {
        f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[u], y[u], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[u + 1], y[u], a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[u], y[u + 1], a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x[u + 1], y[u + 1], a3, 0, 0, 0);
            }
        }
        for (int r = 0; r < 16; r++) s += a0[r] + a1[r] + a2[r] + a3[r];
    } else {
        f32x4 a[8] = {};
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                // same flops as one 32x32x2 group of four: 8 x (16*16*4*2) = 4 x (32*32*2*2) / 2 -> run twice
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    a[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[u], y[u], a[0], 0, 0, 0);
                    a[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[u + 1], y[u], a[1], 0, 0, 0);
                    a[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[u], y[u + 1], a[2], 0, 0, 0);
                    a[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[u + 1], y[u + 1], a[3], 0, 0, 0);
                    a[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(y[u], x[u], a[4], 0, 0, 0);
                    a[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(y[u + 1], x[u], a[5], 0, 0, 0);
                    a[6] = __builtin_amdgcn_mfma_f32_16x16x4f32(y[u], x[u + 1], a[6], 0, 0, 0);
                    a[7] = __builtin_amdgcn_mfma_f32_16x16x4f32(y[u + 1], x[u + 1], a[7], 0, 0, 0);
                }
            }
        }
        for (int r = 0; r < 8; r++) s += a[r][0] + a[r][1] + a[r][2] + a[r][3];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main(int argc, char** argv)
{
    float* o; unsigned long long* c;
    const int grid = 512;
    CK(hipMalloc(&o, 4)); CK(hipMalloc(&c, 16 * grid));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 40000;
    static unsigned long long h[2 * grid];
    for (int shape : {32, 16}) for (int random : {0, 1}) {
        float best = 1e9; double mhz = 0;
        // warm: ~2 s of back-to-back launches so DVFS settles
        for (int r = 0; r < 60; r++) {
            CK(hipEventRecord(e0));
            if (shape == 32) hipLaunchKernelGGL(mf<32>, dim3(grid), dim3(256), 0, 0, o, iters, random, c);
            else hipLaunchKernelGGL(mf<16>, dim3(grid), dim3(256), 0, 0, o, iters, random, c);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 50 && ms < best) {
                best = ms;
                CK(hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost));
                double acc = 0; for (int b = 0; b < grid; b++) acc += (double)h[2 * b] / (double)h[2 * b + 1] * 100.0;
                mhz = acc / grid;
            }
        }
        // (per loop iteration a wave issues 16 MFMAs of the 32x32x2 shape or 64 of the 16x16x4 shape: 65536 / 131072 flop)
        double flops = (double)grid * 4 * iters * (shape == 32 ? 16.0 * (32 * 32 * 2 * 2) : 64.0 * (16 * 16 * 4 * 2));
        printf("shape %s  %s operands: %.3f ms  %.1f TFLOP/s  in-kernel clock %.0f MHz\n", shape == 32 ? "32x32x2" : "16x16x4",
               random ? "random  " : "constant", best, flops / best / 1e9, mhz);
    }
    return 0;
}
