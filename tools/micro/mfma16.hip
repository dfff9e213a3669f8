// Is v_mfma_f32_16x16x4_f32 a k-ordered fmaf chain, bit for bit?  One wave: D = C + A (16x4) * B (4x16), lane (i = lane & 15,
// g = lane >> 4) feeds A[i][g] and B[g][i]; lane holds D[4 (lane >> 4) + r][lane & 15], r = 0..3.  Compared on the host with
// fmaf(a3, b3, fmaf(a2, b2, fmaf(a1, b1, fmaf(a0, b0, c)))) and with the other association orders; chained over many
// instructions with operands of wide dynamic range so that any other rounding order shows.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void k16(const float *A, const float *B, float *D, int steps)
{
    const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < steps; s++) {
        const float a = A[(s * 16 + i) * 4 + g];   // A_s[i][g]
        const float b = B[(s * 4 + g) * 16 + i];   // B_s[g][i]
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; r++) D[(4 * g + r) * 16 + i] = acc[r];
}
int main()
{
    const int steps = 64;
    float *hA = (float *)malloc(steps * 64 * 4), *hB = (float *)malloc(steps * 64 * 4), hD[256];
    srand(7);
    for (int n = 0; n < steps * 64; n++) {
        hA[n] = ldexpf((float)rand() / RAND_MAX - 0.5f, rand() % 24 - 12);
        hB[n] = ldexpf((float)rand() / RAND_MAX - 0.5f, rand() % 24 - 12);
    }
    float *dA, *dB, *dD;
    CK(hipMalloc(&dA, steps * 256)); CK(hipMalloc(&dB, steps * 256)); CK(hipMalloc(&dD, 1024));
    CK(hipMemcpy(dA, hA, steps * 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB, steps * 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dD, steps);
    CK(hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost));
    int bad_chain = 0, bad_rev = 0, bad_pair = 0;
    for (int m = 0; m < 16; m++)
        for (int n = 0; n < 16; n++) {
            float c0 = 0.f, c1 = 0.f, c2 = 0.f;
            for (int s = 0; s < steps; s++) {
                const float *a = hA + (s * 16 + m) * 4;
                float b[4];
                for (int k = 0; k < 4; k++) b[k] = hB[(s * 4 + k) * 16 + n];
                for (int k = 0; k < 4; k++) c0 = fmaf(a[k], b[k], c0);              // k = 0,1,2,3
                for (int k = 3; k >= 0; k--) c1 = fmaf(a[k], b[k], c1);             // reversed
                c2 = c2 + (fmaf(a[1], b[1], a[0] * b[0]) + fmaf(a[3], b[3], a[2] * b[2])); // tree
            }
            unsigned u, v0, v1, v2;
            memcpy(&u, &hD[m * 16 + n], 4); memcpy(&v0, &c0, 4); memcpy(&v1, &c1, 4); memcpy(&v2, &c2, 4);
            bad_chain += u != v0; bad_rev += u != v1; bad_pair += u != v2;
        }
    printf("16x16x4 f32 over %d chained instructions: mismatches vs fmaf chain k=0..3: %d / 256; vs reversed: %d; vs tree: %d\n", steps, bad_chain, bad_rev, bad_pair);
    return bad_chain != 0;
}
