// v_mfma_f32_4x4x1_16B_f32: sixteen 4 x 4 outer products per instruction (64 A values x their block's 4 B values), one fma
// per output.  (1) which lanes meet where; (2) is a chain of them a k-ordered fmaf chain, bit for bit; (3) issue rate beside
// v_mfma_f32_32x32x2_f32 (same 32 fma / cycle / SIMD on paper); (4) v_permlane32_swap moves the upper half of one register
// into the lower half of another (what turns two 32 x 32 x 2 A fragments into 64 rows x one k per register).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void k4(const float *A, const float *B, float *D, int steps)
{
    const int lane = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < steps; s++) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(A[s * 64 + lane], B[s * 64 + lane], acc, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[lane * 4 + r] = acc[r];
}
__global__ void kswap(unsigned *out)
{
    const unsigned lane = threadIdx.x;
    unsigned x = 1000 + lane, y = 2000 + lane;
    auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    out[lane] = r[0]; out[64 + lane] = r[1];
}
template <int MODE> __global__ __launch_bounds__(256) void krate(float *out, int iters)
{
    f32x16 big[2] = {};
    f32x4 small[4] = {};
    const float a = threadIdx.x * 0.001f, b = 1.0f - threadIdx.x * 0.002f;
    for (int it = 0; it < iters; it++) {
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int n = 0; n < 4; n++) big[n & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, big[n & 1], 0, 0, 0);
        }
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int n = 0; n < 8; n++) small[n & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, small[n & 3], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int n = 0; n < 16; n++) s += big[0][n] + big[1][n];
    for (int n = 0; n < 4; n++) s += small[n][0] + small[n][1] + small[n][2] + small[n][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main()
{
    const int steps = 64;
    float *hA = (float *)malloc(steps * 256), *hB = (float *)malloc(steps * 256), hD[256];
    float *dA, *dB, *dD;
    CK(hipMalloc(&dA, steps * 256)); CK(hipMalloc(&dB, steps * 256)); CK(hipMalloc(&dD, 1 << 22));
    // (1) layout: one step, D = A[la] * B[lb]
    for (int l = 0; l < 64; l++) { hA[l] = 1.0f + l / 64.0f; hB[l] = ldexpf(1.0f, l - 32); } // (mantissa = the A lane, exponent = the B lane)
    CK(hipMemcpy(dA, hA, 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB, 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k4, dim3(1), dim3(64), 0, 0, dA, dB, dD, 1);
    CK(hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost));
    int la_of[64][4], lb_of[64][4], regular = 1;
    for (int l = 0; l < 64; l++)
        for (int r = 0; r < 4; r++) {
            la_of[l][r] = lb_of[l][r] = -1;
            for (int la = 0; la < 64; la++)
                for (int lb = 0; lb < 64; lb++)
                    if (hD[l * 4 + r] == hA[la] * hB[lb]) { la_of[l][r] = la; lb_of[l][r] = lb; }
            regular &= la_of[l][r] == 4 * (l / 4) + r && lb_of[l][r] == l;
        }
    printf("4x4x1 16B layout: lane 0: D[r] = A[%d,%d,%d,%d] * B[%d,%d,%d,%d]; lane 5: A[%d,%d,%d,%d] * B[%d,%d,%d,%d]; lane 63: A[%d..%d] * B[%d]\n",
           la_of[0][0], la_of[0][1], la_of[0][2], la_of[0][3], lb_of[0][0], lb_of[0][1], lb_of[0][2], lb_of[0][3],
           la_of[5][0], la_of[5][1], la_of[5][2], la_of[5][3], lb_of[5][0], lb_of[5][1], lb_of[5][2], lb_of[5][3], la_of[63][0], la_of[63][3], lb_of[63][0]);
    printf("   lane l, register r = A[4 (l / 4) + r] * B[l] for every (l, r): %s\n", regular ? "yes" : "NO");
    // (2) chain
    srand(7);
    for (int n = 0; n < steps * 64; n++) {
        hA[n] = ldexpf((float)rand() / RAND_MAX - 0.5f, rand() % 24 - 12);
        hB[n] = ldexpf((float)rand() / RAND_MAX - 0.5f, rand() % 24 - 12);
    }
    CK(hipMemcpy(dA, hA, steps * 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB, steps * 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k4, dim3(1), dim3(64), 0, 0, dA, dB, dD, steps);
    CK(hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; l++)
        for (int r = 0; r < 4; r++) {
            float c = 0.f;
            for (int s = 0; s < steps; s++) c = fmaf(hA[s * 64 + la_of[l][r]], hB[s * 64 + lb_of[l][r]], c);
            bad += memcmp(&c, &hD[l * 4 + r], 4) != 0;
        }
    printf("4x4x1 over %d chained instructions: mismatches vs fmaf chain: %d / 256\n", steps, bad);
    // (4) permlane32 swap
    unsigned hs[128], *ds;
    CK(hipMalloc(&ds, 512));
    hipLaunchKernelGGL(kswap, dim3(1), dim3(64), 0, 0, ds);
    CK(hipMemcpy(hs, ds, 512, hipMemcpyDeviceToHost));
    printf("v_permlane32_swap x, y (builtin; x = 1000 + lane, y = 2000 + lane): x[0] %u x[31] %u x[32] %u x[63] %u | y[0] %u y[31] %u y[32] %u y[63] %u\n",
           hs[0], hs[31], hs[32], hs[63], hs[64], hs[95], hs[96], hs[127]);
    // (3) rate: 1024 workgroups of 4 waves, iters rounds of {4 x 32x32x2 = 256 cycles} / {8 x 4x4x1 = 64 cycles on paper} / both
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    float ms[3];
    for (int mode = 0; mode < 3; mode++)
        for (int rep = 0; rep < 2; rep++) {
            CK(hipEventRecord(e0, 0));
            if (mode == 0) hipLaunchKernelGGL(krate<0>, dim3(1024), dim3(256), 0, 0, dD, iters);
            if (mode == 1) hipLaunchKernelGGL(krate<1>, dim3(1024), dim3(256), 0, 0, dD, iters);
            if (mode == 2) hipLaunchKernelGGL(krate<2>, dim3(1024), dim3(256), 0, 0, dD, iters);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms[mode], e0, e1));
        }
    printf("rate, 1024 x 256 threads x %d rounds: 4 x (32x32x2) %.2f ms | 8 x (4x4x1) %.2f ms | both %.2f ms  (paper: 4 : 1 : 5)\n", iters, ms[0], ms[1], ms[2]);
    return bad != 0;
}
