// Developer tool: fp32 matrix-core issue rate on this device, register operands only.
//   ./mfma_rate   -> TFLOP/s for {32x32x2, 16x16x4} x {1, 2, 4 accumulator chains} x waves per SIMD
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a, float b) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = (float)threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a, float b) {
    f32x4 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 4; ++r) acc[c][r] = (float)threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 4; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
double timeit(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
    hipEventRecord(e0, 0);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3 / 5;
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 8 * 256 * 4 * sizeof(float));
    const int iters = 2000;
    for (int wps = 1; wps <= 4; wps *= 2) {  // waves per SIMD = blocks per CU (256-thread blocks)
        const int blocks = 256 * wps;
        double t;
        t = timeit([&] { hipLaunchKernelGGL(k32<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f); });
        printf("32x32x2 chains=1 waves/SIMD=%d: %.1f TF/s\n", wps, blocks * 4.0 * iters * 8 * 1 * 4096 / t / 1e12);
        t = timeit([&] { hipLaunchKernelGGL(k32<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f); });
        printf("32x32x2 chains=2 waves/SIMD=%d: %.1f TF/s\n", wps, blocks * 4.0 * iters * 8 * 2 * 4096 / t / 1e12);
        t = timeit([&] { hipLaunchKernelGGL(k16<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f); });
        printf("16x16x4 chains=1 waves/SIMD=%d: %.1f TF/s\n", wps, blocks * 4.0 * iters * 8 * 1 * 2048 / t / 1e12);
        t = timeit([&] { hipLaunchKernelGGL(k16<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f); });
        printf("16x16x4 chains=2 waves/SIMD=%d: %.1f TF/s\n", wps, blocks * 4.0 * iters * 8 * 2 * 2048 / t / 1e12);
        t = timeit([&] { hipLaunchKernelGGL(k16<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f); });
        printf("16x16x4 chains=4 waves/SIMD=%d: %.1f TF/s\n", wps, blocks * 4.0 * iters * 8 * 4 * 2048 / t / 1e12);
    }
    return 0;
}
