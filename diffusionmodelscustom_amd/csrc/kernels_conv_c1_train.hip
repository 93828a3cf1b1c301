// The network's last convolution in a training step: 3x3, stride 1, padding 1, Cin channels -> ONE output channel
// (DecoderBlock(last_layer) under modules_DANRA_conditional.py:415-416: nn.Conv2d(64, out_channels = 1, 3, padding = 1)),
// on contiguous NCHW fp32 tensors: forward, data gradient and weight gradient.  With one output channel there is no matrix
// product to speak of (9 Cin multiply-adds per output value); all three are bound by the one pass over the Cin-channel
// tensor they read or write and run on the vector units:
//   forward        y[b][y][x]     = sum_{c, kh, kw} x[b][c][y + kh - 1][x + kw - 1] w[c][kh][kw]
//   data gradient  dx[b][c][y][x] = sum_{kh, kw} dy[b][y - kh + 1][x - kw + 1] w[c][kh][kw]
//   weight grad.   dw[c][kh][kw]  = sum_{b, y, x} dy[b][y][x] x[b][c][y + kh - 1][x + kw - 1]
// The weight gradient is summed per (image, channel) first and over the images by a second kernel, in a fixed order.
#include "danra_kernels.h"

namespace danra {

namespace {
constexpr int kC1Threads = 256;
}

__global__ __launch_bounds__(kC1Threads) void k_conv_c1_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                            float* __restrict__ y, int Cin, int H, int W) {
    extern __shared__ float ws[];   // [Cin][9]
    for (int i = threadIdx.x; i < Cin * 9; i += kC1Threads) ws[i] = w[i];
    __syncthreads();
    const int HW = H * W, b = blockIdx.y;
    const int p = blockIdx.x * kC1Threads + threadIdx.x;
    if (p >= HW) return;
    const int py = p / W, px = p - py * W;
    bool ok[9];
    int off[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int iy = py + t / 3 - 1, ix = px + t % 3 - 1;
        ok[t] = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        off[t] = ok[t] ? iy * W + ix : 0;
    }
    const float* xb = x + (size_t)b * Cin * HW;
    float acc = 0.f;
    for (int c = 0; c < Cin; ++c) {
        const float* xc = xb + (size_t)c * HW;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float v = ok[t] ? xc[off[t]] : 0.f;
            acc = fmaf(v, ws[c * 9 + t], acc);
        }
    }
    y[(size_t)b * HW + p] = acc;
}

__global__ __launch_bounds__(kC1Threads) void k_conv_c1_dgrad(const float* __restrict__ dy, const float* __restrict__ w,
                                                              float* __restrict__ dx, int Cin, int H, int W) {
    extern __shared__ float ws[];   // [Cin][9]
    for (int i = threadIdx.x; i < Cin * 9; i += kC1Threads) ws[i] = w[i];
    __syncthreads();
    const int HW = H * W, b = blockIdx.y;
    const int p = blockIdx.x * kC1Threads + threadIdx.x;
    if (p >= HW) return;
    const int py = p / W, px = p - py * W;
    float g[9];   // dy under tap t of the output that reads this input value: output (py - kh + 1, px - kw + 1)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int oy = py - (t / 3) + 1, ox = px - (t % 3) + 1;
        g[t] = ((unsigned)oy < (unsigned)H && (unsigned)ox < (unsigned)W) ? dy[(size_t)b * HW + oy * W + ox] : 0.f;
    }
    float* db = dx + (size_t)b * Cin * HW + p;
    for (int c = 0; c < Cin; ++c) {
        float a = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) a = fmaf(g[t], ws[c * 9 + t], a);
        db[(size_t)c * HW] = a;
    }
}

// partial[b][c][9]: the sums over the pixels of image b
__global__ __launch_bounds__(kC1Threads) void k_conv_c1_wgrad(const float* __restrict__ x, const float* __restrict__ dy,
                                                              float* __restrict__ partial, int Cin, int H, int W) {
    __shared__ float red[kC1Threads / 64][9];
    const int HW = H * W, c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const float* xc = x + ((size_t)b * Cin + c) * HW;
    const float* dyb = dy + (size_t)b * HW;
    float acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = 0.f;
    for (int p = tid; p < HW; p += kC1Threads) {
        const int py = p / W, px = p - py * W;
        const float g = dyb[p];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int iy = py + t / 3 - 1, ix = px + t % 3 - 1;
            const float v = ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? xc[iy * W + ix] : 0.f;
            acc[t] = fmaf(g, v, acc[t]);
        }
    }
    // wave sums (fixed butterfly order), then the four waves in order
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float v = acc[t];
        for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s);
        if ((tid & 63) == 0) red[tid >> 6][t] = v;
    }
    __syncthreads();
    if (tid < 9) partial[((size_t)b * Cin + c) * 9 + tid] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
}

__global__ __launch_bounds__(kC1Threads) void k_conv_c1_wgrad_sum(const float* __restrict__ partial, float* __restrict__ dw, int n, int B) {
    const int i = blockIdx.x * kC1Threads + threadIdx.x;
    if (i >= n) return;
    float a = 0.f;
    for (int b = 0; b < B; ++b) a += partial[(size_t)b * n + i];
    dw[i] = a;
}

bool conv_c1_train_supported(int B, int Cin, int H, int W) {
    return B > 0 && B <= 65535 && Cin > 0 && Cin <= 1024 && H > 0 && W > 0 && (long long)B * Cin * H * W < (1LL << 31);
}
size_t conv_c1_train_workspace_floats(int B, int Cin, int H, int W) {
    return conv_c1_train_supported(B, Cin, H, W) ? (size_t)B * Cin * 9 : 0;
}

hipError_t launch_conv_c1_train_fwd(const float* x, const float* w, float* y, int B, int Cin, int H, int W, hipStream_t stream) {
    if (!conv_c1_train_supported(B, Cin, H, W)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_conv_c1_fwd, dim3((H * W + kC1Threads - 1) / kC1Threads, B), dim3(kC1Threads), (size_t)Cin * 9 * sizeof(float),
                       stream, x, w, y, Cin, H, W);
    return hipGetLastError();
}
hipError_t launch_conv_c1_train_dgrad(const float* dy, const float* w, float* dx, int B, int Cin, int H, int W, hipStream_t stream) {
    if (!conv_c1_train_supported(B, Cin, H, W)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_conv_c1_dgrad, dim3((H * W + kC1Threads - 1) / kC1Threads, B), dim3(kC1Threads), (size_t)Cin * 9 * sizeof(float),
                       stream, dy, w, dx, Cin, H, W);
    return hipGetLastError();
}
hipError_t launch_conv_c1_train_wgrad(const float* x, const float* dy, float* dw, int B, int Cin, int H, int W, float* workspace,
                                      hipStream_t stream) {
    if (!conv_c1_train_supported(B, Cin, H, W) || !workspace) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_conv_c1_wgrad, dim3(Cin, B), dim3(kC1Threads), 0, stream, x, dy, workspace, Cin, H, W);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int n = Cin * 9;
    hipLaunchKernelGGL(k_conv_c1_wgrad_sum, dim3((n + kC1Threads - 1) / kC1Threads), dim3(kC1Threads), 0, stream, workspace, dw, n, B);
    return hipGetLastError();
}

}  // namespace danra
