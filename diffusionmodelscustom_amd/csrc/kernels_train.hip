// Training-step kernels outside the attention core: BatchNorm2d in train mode (batch statistics), forward and backward,
// on NCHW fp32 tensors as the differentiable torch composition holds them.
//
// One workgroup per channel.  A channel's B*HW values (a few thousand to a few ten thousand at the network's sizes) are
// walked three times in the forward pass (sum -> centred sum of squares -> normalise) and twice in the backward pass; after
// the first walk they come from L2.  Two-pass statistics: the biased variance equals a direct evaluation, as in torch.
#include "danra_kernels.h"

namespace danra {

// Sum of `v` over the 256 threads of the workgroup, returned to every thread.
__device__ __forceinline__ float block_sum256(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();  // the previous use of `red` has been read
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// y = (x - mean) * rstd * gamma + beta [then ReLU]; mean / rstd of the batch are kept for the backward pass; the running
// statistics move by `momentum` towards the batch mean and the unbiased batch variance (torch.nn.BatchNorm2d, training).
__global__ __launch_bounds__(256) void k_bn_train_fwd(const float* __restrict__ x, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float* __restrict__ y,
                                                      float* __restrict__ save_mean, float* __restrict__ save_rstd,
                                                      float* __restrict__ running_mean, float* __restrict__ running_var,
                                                      int B, int C, int HW, float eps, float momentum, int relu) {
    __shared__ float red[4];
    const int c = blockIdx.x, tid = threadIdx.x;
    const int hw4 = HW / 4;            // HW is a multiple of 4 (checked by the launcher)
    const int per = B * hw4;           // float4 items of this channel
    const float n = (float)B * (float)HW;
    auto item = [&](int i) -> size_t {  // float offset of item i: image i / hw4, position 4 * (i % hw4)
        const int b = i / hw4, r = i - b * hw4;
        return ((size_t)b * C + c) * HW + 4 * (size_t)r;
    };
    float s = 0.f;
    for (int i = tid; i < per; i += 256) {
        const float4 v = *reinterpret_cast<const float4*>(x + item(i));
        s += (v.x + v.y) + (v.z + v.w);
    }
    const float mean = block_sum256(s, red) / n;
    float q = 0.f;
    for (int i = tid; i < per; i += 256) {
        const float4 v = *reinterpret_cast<const float4*>(x + item(i));
        const float a = v.x - mean, b2 = v.y - mean, c2 = v.z - mean, d = v.w - mean;
        q += (a * a + b2 * b2) + (c2 * c2 + d * d);
    }
    const float var = block_sum256(q, red) / n;
    const float rstd = 1.0f / sqrtf(var + eps);
    const float g = gamma[c] * rstd, sh = beta[c];
    for (int i = tid; i < per; i += 256) {
        const size_t o = item(i);
        const float4 v = *reinterpret_cast<const float4*>(x + o);
        float4 r = make_float4((v.x - mean) * g + sh, (v.y - mean) * g + sh, (v.z - mean) * g + sh, (v.w - mean) * g + sh);
        if (relu) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
        *reinterpret_cast<float4*>(y + o) = r;
    }
    if (tid == 0) {
        save_mean[c] = mean;
        save_rstd[c] = rstd;
        if (running_mean) {
            running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mean;
            running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (var * (n / fmaxf(n - 1.0f, 1.0f)));
        }
    }
}

// g = dy (masked by y > 0 when the forward applied ReLU); dbeta = sum g; dgamma = sum g * xhat;
// dx = gamma * rstd * (g - dbeta / n - xhat * dgamma / n).
__global__ __launch_bounds__(256) void k_bn_train_bwd(const float* __restrict__ x, const float* __restrict__ dy,
                                                      const float* __restrict__ y, const float* __restrict__ gamma,
                                                      const float* __restrict__ save_mean,
                                                      const float* __restrict__ save_rstd, float* __restrict__ dx,
                                                      float* __restrict__ dgamma, float* __restrict__ dbeta, int B, int C,
                                                      int HW, int relu) {
    __shared__ float red[4];
    const int c = blockIdx.x, tid = threadIdx.x;
    const int hw4 = HW / 4;
    const int per = B * hw4;
    const float n = (float)B * (float)HW;
    const float mean = save_mean[c], rstd = save_rstd[c];
    auto item = [&](int i) -> size_t {
        const int b = i / hw4, r = i - b * hw4;
        return ((size_t)b * C + c) * HW + 4 * (size_t)r;
    };
    auto masked = [&](size_t o) -> float4 {
        float4 g = *reinterpret_cast<const float4*>(dy + o);
        if (relu) {
            const float4 yy = *reinterpret_cast<const float4*>(y + o);
            g.x = yy.x > 0.f ? g.x : 0.f; g.y = yy.y > 0.f ? g.y : 0.f; g.z = yy.z > 0.f ? g.z : 0.f; g.w = yy.w > 0.f ? g.w : 0.f;
        }
        return g;
    };
    float sg = 0.f, sgx = 0.f;
    for (int i = tid; i < per; i += 256) {
        const size_t o = item(i);
        const float4 g = masked(o);
        const float4 v = *reinterpret_cast<const float4*>(x + o);
        sg += (g.x + g.y) + (g.z + g.w);
        sgx += (g.x * ((v.x - mean) * rstd) + g.y * ((v.y - mean) * rstd)) + (g.z * ((v.z - mean) * rstd) + g.w * ((v.w - mean) * rstd));
    }
    const float db = block_sum256(sg, red);
    const float dg = block_sum256(sgx, red);
    if (tid == 0) {
        dgamma[c] = dg;
        dbeta[c] = db;
    }
    const float k = gamma[c] * rstd, mb = db / n, mg = dg / n;
    for (int i = tid; i < per; i += 256) {
        const size_t o = item(i);
        const float4 g = masked(o);
        const float4 v = *reinterpret_cast<const float4*>(x + o);
        float4 r;
        r.x = k * (g.x - mb - (v.x - mean) * rstd * mg);
        r.y = k * (g.y - mb - (v.y - mean) * rstd * mg);
        r.z = k * (g.z - mb - (v.z - mean) * rstd * mg);
        r.w = k * (g.w - mb - (v.w - mean) * rstd * mg);
        *reinterpret_cast<float4*>(dx + o) = r;
    }
}

// InstanceNorm2d(affine=False) on NCHW fp32: one wave per (image, channel) plane of HW values (HW a multiple of 4; the
// planes of this network hold 16 .. 16384 values), two-pass statistics, mean / rstd kept per plane for the backward pass.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(256) void k_in_train_fwd(const float* __restrict__ x, float* __restrict__ y,
                                                      float* __restrict__ save_mean, float* __restrict__ save_rstd,
                                                      int planes, int HW, float eps) {
    const int plane = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (plane >= planes) return;
    const float* src = x + (size_t)plane * HW;
    float* dst = y + (size_t)plane * HW;
    const int n4 = HW / 4;
    float s = 0.f;
    for (int i = lane; i < n4; i += 64) {
        const float4 v = *reinterpret_cast<const float4*>(src + 4 * i);
        s += (v.x + v.y) + (v.z + v.w);
    }
    const float mean = wave_sum(s) / (float)HW;
    float q = 0.f;
    for (int i = lane; i < n4; i += 64) {
        const float4 v = *reinterpret_cast<const float4*>(src + 4 * i);
        const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
        q += (a * a + b * b) + (c * c + d * d);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)HW + eps);
    for (int i = lane; i < n4; i += 64) {
        const float4 v = *reinterpret_cast<const float4*>(src + 4 * i);
        *reinterpret_cast<float4*>(dst + 4 * i) =
            make_float4((v.x - mean) * rstd, (v.y - mean) * rstd, (v.z - mean) * rstd, (v.w - mean) * rstd);
    }
    if (lane == 0) {
        save_mean[plane] = mean;
        save_rstd[plane] = rstd;
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)) per plane
__global__ __launch_bounds__(256) void k_in_train_bwd(const float* __restrict__ x, const float* __restrict__ dy,
                                                      const float* __restrict__ save_mean,
                                                      const float* __restrict__ save_rstd, float* __restrict__ dx,
                                                      int planes, int HW) {
    const int plane = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (plane >= planes) return;
    const float* src = x + (size_t)plane * HW;
    const float* g0 = dy + (size_t)plane * HW;
    float* dst = dx + (size_t)plane * HW;
    const int n4 = HW / 4;
    const float mean = save_mean[plane], rstd = save_rstd[plane];
    float sg = 0.f, sgx = 0.f;
    for (int i = lane; i < n4; i += 64) {
        const float4 v = *reinterpret_cast<const float4*>(src + 4 * i);
        const float4 g = *reinterpret_cast<const float4*>(g0 + 4 * i);
        sg += (g.x + g.y) + (g.z + g.w);
        sgx += (g.x * ((v.x - mean) * rstd) + g.y * ((v.y - mean) * rstd)) + (g.z * ((v.z - mean) * rstd) + g.w * ((v.w - mean) * rstd));
    }
    const float mg = wave_sum(sg) / (float)HW, mgx = wave_sum(sgx) / (float)HW;
    for (int i = lane; i < n4; i += 64) {
        const float4 v = *reinterpret_cast<const float4*>(src + 4 * i);
        const float4 g = *reinterpret_cast<const float4*>(g0 + 4 * i);
        float4 r;
        r.x = rstd * (g.x - mg - (v.x - mean) * rstd * mgx);
        r.y = rstd * (g.y - mg - (v.y - mean) * rstd * mgx);
        r.z = rstd * (g.z - mg - (v.z - mean) * rstd * mgx);
        r.w = rstd * (g.w - mg - (v.w - mean) * rstd * mgx);
        *reinterpret_cast<float4*>(dst + 4 * i) = r;
    }
}

hipError_t launch_in_train_fwd(const float* x, float* y, float* save_mean, float* save_rstd, int planes, int HW, float eps,
                               hipStream_t stream) {
    if (HW % 4 || planes <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_in_train_fwd, dim3((planes + 3) / 4), dim3(256), 0, stream, x, y, save_mean, save_rstd, planes, HW, eps);
    return hipGetLastError();
}

hipError_t launch_in_train_bwd(const float* x, const float* dy, const float* save_mean, const float* save_rstd, float* dx,
                               int planes, int HW, hipStream_t stream) {
    if (HW % 4 || planes <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_in_train_bwd, dim3((planes + 3) / 4), dim3(256), 0, stream, x, dy, save_mean, save_rstd, dx, planes, HW);
    return hipGetLastError();
}

hipError_t launch_bn_train_fwd(const float* x, const float* gamma, const float* beta, float* y, float* save_mean,
                               float* save_rstd, float* running_mean, float* running_var, int B, int C, int HW, float eps,
                               float momentum, int relu, hipStream_t stream) {
    if (HW % 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_bn_train_fwd, dim3(C), dim3(256), 0, stream, x, gamma, beta, y, save_mean, save_rstd, running_mean,
                       running_var, B, C, HW, eps, momentum, relu);
    return hipGetLastError();
}

hipError_t launch_bn_train_bwd(const float* x, const float* dy, const float* y, const float* gamma, const float* save_mean,
                               const float* save_rstd, float* dx, float* dgamma, float* dbeta, int B, int C, int HW, int relu,
                               hipStream_t stream) {
    if (HW % 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_bn_train_bwd, dim3(C), dim3(256), 0, stream, x, dy, y, gamma, save_mean, save_rstd, dx, dgamma, dbeta,
                       B, C, HW, relu);
    return hipGetLastError();
}

}  // namespace danra
