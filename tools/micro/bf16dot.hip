// micro-test of the bf16 beam scoring: to_bf16 conversion + lane-partial dot as in hnsw_beam_kernel<2, ., true>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>
#include <cstring>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) f32x4 *gptr4;
__global__ void to_bf16_kernel(const float *__restrict__ src, int64_t total, __bf16 *__restrict__ dst)
{
    int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    for (; i < total; i += (int64_t)gridDim.x * blockDim.x * 4) {
        const f32x4 v = *(const f32x4 *)(src + i);
        dst[i] = (__bf16)v[0]; dst[i + 1] = (__bf16)v[1]; dst[i + 2] = (__bf16)v[2]; dst[i + 3] = (__bf16)v[3];
    }
}
__device__ __forceinline__ float wave_sum_f32(float x)
{
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}
__global__ void dot_kernel(const float *xq, const uint16_t *xb16, int dp, int nrows, float *out, float *parts)
{
    constexpr int NCH = 2;
    const int lane = threadIdx.x & 63;
    const int nchunks = dp >> 3;
    f32x4 qv[NCH * 2];
    for (int i = 0; i < NCH; i++) {
        const int c = i * 64 + lane;
        for (int w = 0; w < 2; w++) qv[i * 2 + w] = c < nchunks ? *(const f32x4 *)(xq + 4 * (c * 2 + w)) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
        const float *src = (const float *)(xb16 + (size_t)row * dp);
        f32x4 v[NCH];
        for (int i = 0; i < NCH; i++) {
            const int c = i * 64 + lane;
            v[i] = c < nchunks ? *(gptr4)(src + 4 * c) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        float part = 0.f;
        for (int i = 0; i < NCH; i++)
            for (int e = 0; e < 4; e++) {
                const float fe = v[i][e];
                const uint32_t w2 = __float_as_uint(fe);
                part = __builtin_fmaf(__builtin_bit_cast(float, w2 << 16), qv[2 * i + (e >> 1)][(2 * e) & 3], part);
                part = __builtin_fmaf(__builtin_bit_cast(float, w2 & 0xFFFF0000u), qv[2 * i + (e >> 1)][(2 * e + 1) & 3], part);
            }
        if (row == 0) parts[lane] = part;
        if (row == 0 && lane == 0) { for (int i = 0; i < 2; i++) for (int e = 0; e < 4; e++) { parts[64 + i * 4 + e] = v[i][e]; } for (int k2 = 0; k2 < 4; k2++) for (int e = 0; e < 4; e++) parts[80 + k2 * 4 + e] = qv[k2][e]; }
        const float dot = wave_sum_f32(part);
        if (lane == 0) out[row] = dot;
    }
}
int main()
{
    const int n = 512, d = 1024;
    std::vector<float> x((size_t)n * d), q(d);
    srand(1);
    for (auto &v : x) v = (rand() / (float)RAND_MAX - 0.5f) / 16.f;
    for (auto &v : q) v = (rand() / (float)RAND_MAX - 0.5f) / 16.f;
    float *dx, *dq, *dout; uint16_t *d16;
    hipMalloc(&dx, x.size() * 4); hipMalloc(&dq, d * 4); hipMalloc(&dout, n * 4); hipMalloc(&d16, x.size() * 2);
    hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dq, q.data(), d * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(to_bf16_kernel, dim3(64), dim3(256), 0, 0, dx, (int64_t)n * d, (__bf16 *)d16);
    float *dparts; hipMalloc(&dparts, 128 * 4);
    hipLaunchKernelGGL(dot_kernel, dim3(64), dim3(64), 0, 0, dq, d16, d, n, dout, dparts);
    std::vector<float> out(n);
    hipMemcpy(out.data(), dout, n * 4, hipMemcpyDeviceToHost);
    std::vector<uint16_t> h16(x.size());
    hipMemcpy(h16.data(), d16, x.size() * 2, hipMemcpyDeviceToHost);
    int badconv = 0;
    for (size_t i = 0; i < x.size(); i++) {
        uint32_t u; memcpy(&u, &x[i], 4);
        const uint16_t t = (uint16_t)(u >> 16), r = (uint16_t)((u + 0x7FFF + ((u >> 16) & 1)) >> 16);
        if (h16[i] != r && h16[i] != t) { if (badconv < 5) printf("conv mismatch at %zu: got %04x want %04x/%04x\n", i, h16[i], r, t); badconv++; }
    }
    printf("conversion mismatches: %d of %zu\n", badconv, x.size());
    std::vector<float> parts(128);
    hipMemcpy(parts.data(), dparts, 512, hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; i++) { uint32_t u; memcpy(&u, &parts[64 + i], 4); const int c = (i / 4) * 64; const int e = i % 4; printf("v[%d][%d] = %08x, host pair (%04x lo, %04x hi)\n", i / 4, e, u, h16[8 * c + 2 * e], h16[8 * c + 2 * e + 1]); }
    for (int k2 = 0; k2 < 4; k2++) printf("qv[%d] = %g %g %g %g ; host q[%d..] = %g %g %g %g\n", k2, parts[80 + 4 * k2], parts[81 + 4 * k2], parts[82 + 4 * k2], parts[83 + 4 * k2], (k2 / 2) * 512 + (k2 % 2) * 4, q[(k2 / 2) * 512 + (k2 % 2) * 4], q[(k2 / 2) * 512 + (k2 % 2) * 4 + 1], q[(k2 / 2) * 512 + (k2 % 2) * 4 + 2], q[(k2 / 2) * 512 + (k2 % 2) * 4 + 3]);
    for (int lane = 0; lane < 64; lane += 21) {
        double ref = 0;
        for (int i = 0; i < 2; i++) {
            const int c = i * 64 + lane;
            for (int j = 0; j < 8; j++) { uint32_t u = (uint32_t)h16[8 * c + j] << 16; float f; memcpy(&f, &u, 4); ref += (double)f * q[8 * c + j]; }
        }
        double h2 = 0, h3 = 0, h4 = 0;
        for (int i = 0; i < 2; i++) {
            const int c = i * 64 + lane;
            for (int j = 0; j < 8; j++) {
                auto bf = [&](size_t idx) { uint32_t u = (uint32_t)h16[idx] << 16; float f; memcpy(&f, &u, 4); return (double)f; };
                h2 += bf(8 * c + (j ^ 1)) * q[8 * c + j];            // halves swapped
                h3 += bf(8 * c + j) * q[8 * c + (j % 4) + 4 * ((j / 2) % 2)]; // some q mix
                h4 += (double)x[8 * c + j] * q[8 * c + j];
            }
        }
        printf("lane %d: gpu partial %g, expected %g, swapped %g, fp32 %g\n", lane, parts[lane], ref, h2, h4);
    }
    double worst2 = 0;
    for (int r = 0; r < n; r++) {
        double ref = 0;
        for (int i = 0; i < d; i++) { uint32_t u = (uint32_t)h16[(size_t)r * d + i] << 16; float f; memcpy(&f, &u, 4); ref += (double)f * q[i]; }
        worst2 = fmax(worst2, fabs(ref - out[r]));
    }
    printf("max |gpu dot - cpu dot of the same bf16 data|: %g\n", worst2);
    double worst = 0;
    for (int r = 0; r < n; r++) {
        double ref = 0;
        for (int i = 0; i < d; i++) ref += (double)x[(size_t)r * d + i] * q[i];
        worst = fmax(worst, fabs(ref - out[r]));
    }
    printf("max |bf16 dot - fp64 dot| over %d rows: %g (scores ~ %g)\n", n, worst, 1.0 / 16 / 16 / 12 * sqrt((double)d));
    return 0;
}
