// Training-step kernels outside the attention core: BatchNorm2d in train mode (batch statistics), forward and backward,
// on NCHW fp32 tensors as the differentiable torch composition holds them.
//
// One workgroup per channel.  A channel's B*HW values (a few thousand to a few ten thousand at the network's sizes) are
// walked three times in the forward pass (sum -> centred sum of squares -> normalise) and twice in the backward pass; after
// the first walk they come from L2.  Two-pass statistics: the biased variance equals a direct evaluation, as in torch.
#include <type_traits>
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
                                                      int B, int C, int HW, float eps, float momentum, int relu,
                                                      const float* __restrict__ res, long long* __restrict__ batches) {
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
        if (res) {  // the block's shortcut, added in front of the ReLU
            const float4 a = *reinterpret_cast<const float4*>(res + o);
            r.x += a.x; r.y += a.y; r.z += a.z; r.w += a.w;
        }
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
        if (batches && c == 0) *batches += 1;   // nn.BatchNorm2d.num_batches_tracked (one writer per launch)
    }
}

// g = dy (masked by y > 0 when the forward applied ReLU); dbeta = sum g; dgamma = sum g * xhat;
// dx = gamma * rstd * (g - dbeta / n - xhat * dgamma / n).
__global__ __launch_bounds__(256) void k_bn_train_bwd(const float* __restrict__ x, const float* __restrict__ dy,
                                                      const float* __restrict__ y, const float* __restrict__ gamma,
                                                      const float* __restrict__ save_mean,
                                                      const float* __restrict__ save_rstd, float* __restrict__ dx,
                                                      float* __restrict__ dgamma, float* __restrict__ dbeta, int B, int C,
                                                      int HW, int relu, float* __restrict__ dres) {
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
        if (dres) *reinterpret_cast<float4*>(dres + o) = g;  // the shortcut's gradient: dy behind the ReLU mask
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
                               float momentum, int relu, hipStream_t stream, const float* res, long long* batches) {
    if (HW % 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_bn_train_fwd, dim3(C), dim3(256), 0, stream, x, gamma, beta, y, save_mean, save_rstd, running_mean,
                       running_var, B, C, HW, eps, momentum, relu, res, batches);
    return hipGetLastError();
}

// out[c] = sum over images and positions of x[b][c][.] (the gradient of a convolution's bias).  Workgroup (c, s): channel c,
// images s, s + S, ...; partial sums per thread in a fixed order, then the block's tree; with S > 1 the S partial sums of
// a channel (in `part`) are added in order by k_channel_sum_merge: the same bits on every call.
__global__ __launch_bounds__(256) void k_channel_sum(const float* __restrict__ x, float* __restrict__ out, int B, int C, int HW) {
    __shared__ float red[4];
    const int c = blockIdx.x, s = blockIdx.y, S = gridDim.y, tid = threadIdx.x;
    const int hw4 = HW / 4;
    const int nb = (B - s + S - 1) / S;       // images of this workgroup
    const int per = nb * hw4;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;   // four chains: the loads of four iterations in flight
    auto at = [&](int i) -> float4 {
        const int bi = i / hw4, r = i - bi * hw4;
        return *reinterpret_cast<const float4*>(x + ((size_t)(s + bi * S) * C + c) * HW + 4 * (size_t)r);
    };
    int i = tid;
    for (; i + 768 < per; i += 1024) {
        const float4 a = at(i), b4 = at(i + 256), c4 = at(i + 512), d4 = at(i + 768);
        s0 += (a.x + a.y) + (a.z + a.w);
        s1 += (b4.x + b4.y) + (b4.z + b4.w);
        s2 += (c4.x + c4.y) + (c4.z + c4.w);
        s3 += (d4.x + d4.y) + (d4.z + d4.w);
    }
    for (; i < per; i += 256) {
        const float4 a = at(i);
        s0 += (a.x + a.y) + (a.z + a.w);
    }
    const float total = block_sum256((s0 + s1) + (s2 + s3), red);
    if (tid == 0) out[(size_t)s * C + c] = total;
}

__global__ void k_channel_sum_merge(const float* __restrict__ part, float* __restrict__ out, int C, int S) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float t = part[c];
    for (int s = 1; s < S; ++s) t += part[(size_t)s * C + c];
    out[c] = t;
}

// images per channel split over this many workgroups (1: no second kernel, no workspace)
int channel_sum_slices(int B, int C, int HW) {
    const long long bytes = (long long)B * HW * 4;            // per channel
    if (bytes < (1 << 17) || C >= 256) return 1;
    int S = 256 / C > 1 ? 512 / C : 1;                         // about 512 workgroups
    if (S > 16) S = 16;
    if (S > B) S = B;
    return S < 1 ? 1 : S;
}

hipError_t launch_channel_sum(const float* x, float* out, int B, int C, int HW, float* workspace, hipStream_t stream) {
    if (B <= 0 || C <= 0 || HW <= 0 || HW % 4) return hipErrorInvalidValue;
    const int S = workspace ? channel_sum_slices(B, C, HW) : 1;
    hipLaunchKernelGGL(k_channel_sum, dim3(C, S), dim3(256), 0, stream, x, S > 1 ? workspace : out, B, C, HW);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || S == 1) return e;
    hipLaunchKernelGGL(k_channel_sum_merge, dim3((C + 255) / 256), dim3(256), 0, stream, workspace, out, C, S);
    return hipGetLastError();
}

hipError_t launch_bn_train_bwd(const float* x, const float* dy, const float* y, const float* gamma, const float* save_mean,
                               const float* save_rstd, float* dx, float* dgamma, float* dbeta, int B, int C, int HW, int relu,
                               hipStream_t stream, float* dres) {
    if (HW % 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_bn_train_bwd, dim3(C), dim3(256), 0, stream, x, dy, y, gamma, save_mean, save_rstd, dx, dgamma, dbeta,
                       B, C, HW, relu, dres);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of a linear layer over tokens: dW[n][k] = sum_m dY[m][n] X[m][k], db[n] = sum_m dY[m][n]
// (torch.nn.functional.linear backward for y = x W^T + b; the attention projections of the 64-channel levels:
// M = batch x tokens = 8192 ... 131072 rows, N = 64 or 192, K = 64).  A GEMM whose reduction runs over the long
// dimension and whose result is tiny: the rows are cut into slices, one workgroup per slice forms the whole N x K
// product of its rows, and a second kernel adds the slices in a fixed order (bitwise reproducible).
//
// Both operands are read straight from memory in their natural layout: for v_mfma_f32_32x32x2_f32 lane (i, h)
// supplies A[i][k = h] and B[k = h][j = i], and here the reduction index k is the row m, so lanes 0-31 read row m,
// lanes 32-63 row m + 1, each lane WF consecutive floats of it (one dwordx4 / dwordx2 load): component c of that
// load is the operand of a 32 x 32 tile whose rows are the columns n = base + WF i + c of dY - the tiles interleave
// in n, which only matters when the result is written out.  The N/32 x K/32 tiles are dealt to the four waves
// (three each for N = 192), every wave walks all row pairs of the slice, eight pairs per batch of loads with the next
// batch in flight behind the matrix instructions of the current one, and writes its tiles itself: no LDS, no barrier.
template <int NF, int KF, int W>  // N = 32 NF, K = 32 KF, W = wave
__device__ __forceinline__ void wgrad_wave(const float* __restrict__ dY, const float* __restrict__ X, float* __restrict__ pw,
                                           float* __restrict__ pb, int m_begin, int m_end, int lane) {
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    constexpr int N = 32 * NF, K = 32 * KF;
    constexpr int NA4 = NF / 4, NA2 = (NF % 4) / 2, NA1 = NF % 2;  // pieces of a dY row: 128, 64 and 32 columns wide
    constexpr int KB4 = KF / 4, KB2 = (KF % 4) / 2, KB1 = KF % 2;  // ... of an X row
    constexpr int TPW = NF * KF / 4, T0 = W * TPW;                // this wave's tiles: t = a KF + b, T0 <= t < T0 + TPW
    static_assert((NF * KF) % 4 == 0, "tiles are dealt evenly to four waves");
    constexpr int U = 8;  // row pairs per batch of loads
    const int i = lane & 31, h = lane >> 5;
    f32x16 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum[NF];  // db share of the tiles (a, 0) this wave owns: column base + WF i + c, this lane's row parity
#pragma unroll
    for (int a = 0; a < NF; ++a) bsum[a] = 0.f;
    float ra[2][U][NF], rb[2][U][KF];

    auto load_batch = [&](auto buf_tag, int sp0) {
        constexpr int BUF = decltype(buf_tag)::value;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int m = m_begin + 2 * (sp0 + u) + h;
            const bool live = m < m_end;  // (past the slice, or the second row of an odd tail: zeros)
            const float* yrow = dY + (size_t)(live ? m : m_end - 1) * N;
            const float* xrow = X + (size_t)(live ? m : m_end - 1) * K;
            int o = 0;
#pragma unroll
            for (int p = 0; p < NA4; ++p) {
                const float4 v = *reinterpret_cast<const float4*>(yrow + 128 * p + 4 * i);
                ra[BUF][u][o++] = v.x; ra[BUF][u][o++] = v.y; ra[BUF][u][o++] = v.z; ra[BUF][u][o++] = v.w;
            }
#pragma unroll
            for (int p = 0; p < NA2; ++p) {
                const float2 v = *reinterpret_cast<const float2*>(yrow + 128 * NA4 + 64 * p + 2 * i);
                ra[BUF][u][o++] = v.x; ra[BUF][u][o++] = v.y;
            }
            if (NA1) ra[BUF][u][o++] = yrow[128 * NA4 + 64 * NA2 + i];
            o = 0;
#pragma unroll
            for (int p = 0; p < KB4; ++p) {
                const float4 v = *reinterpret_cast<const float4*>(xrow + 128 * p + 4 * i);
                rb[BUF][u][o++] = v.x; rb[BUF][u][o++] = v.y; rb[BUF][u][o++] = v.z; rb[BUF][u][o++] = v.w;
            }
#pragma unroll
            for (int p = 0; p < KB2; ++p) {
                const float2 v = *reinterpret_cast<const float2*>(xrow + 128 * KB4 + 64 * p + 2 * i);
                rb[BUF][u][o++] = v.x; rb[BUF][u][o++] = v.y;
            }
            if (KB1) rb[BUF][u][o++] = xrow[128 * KB4 + 64 * KB2 + i];
            // only dY needs the zeros: a product with a zero operand contributes nothing whatever X holds (finite: it
            // is a row of the slice)
#pragma unroll
            for (int a = 0; a < NF; ++a) ra[BUF][u][a] = live ? ra[BUF][u][a] : 0.f;
        }
    };
    auto multiply_batch = [&](auto buf_tag) {
        constexpr int BUF = decltype(buf_tag)::value;
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                constexpr int dummy = 0;
                (void)dummy;
                const int a = (T0 + t) / KF, b = (T0 + t) % KF;
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[BUF][u][a], rb[BUF][u][b], acc[t], 0, 0, 0);
                if (b == 0) bsum[a] += ra[BUF][u][a];
            }
        }
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    const int pairs = (m_end - m_begin + 1) / 2;
    load_batch(B0{}, 0);
    for (int sp = 0; sp < pairs; sp += 2 * U) {
        load_batch(B1{}, sp + U);
        multiply_batch(B0{});
        load_batch(B0{}, sp + 2 * U);
        multiply_batch(B1{});
    }

    // column of dY (row of dW) that tile a holds in its row ii, and column of X that tile b holds in its column jj
    auto n_of = [&](int a, int ii) {
        if (a < 4 * NA4) return 128 * (a / 4) + 4 * ii + (a % 4);
        if (a < 4 * NA4 + 2 * NA2) return 128 * NA4 + 64 * ((a - 4 * NA4) / 2) + 2 * ii + ((a - 4 * NA4) % 2);
        return 128 * NA4 + 64 * NA2 + ii;
    };
    auto k_of = [&](int b, int jj) {
        if (b < 4 * KB4) return 128 * (b / 4) + 4 * jj + (b % 4);
        if (b < 4 * KB4 + 2 * KB2) return 128 * KB4 + 64 * ((b - 4 * KB4) / 2) + 2 * jj + ((b - 4 * KB4) % 2);
        return 128 * KB4 + 64 * KB2 + jj;
    };
    // acc[t][r] = D[row (r & 3) + 8 (r >> 2) + 4 h][column i]
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int a = (T0 + t) / KF, b = (T0 + t) % KF;
#pragma unroll
        for (int r = 0; r < 16; ++r) pw[n_of(a, (r & 3) + 8 * (r >> 2) + 4 * h) * K + k_of(b, i)] = acc[t][r];
        if (b == 0 && pb) {
            const float both = bsum[a] + __shfl_xor(bsum[a], 32);  // the two rows of a pair
            if (h == 0) pb[n_of(a, i)] = both;
        }
    }
}

template <int NF, int KF>
__global__ __launch_bounds__(256) void k_linear_wgrad(const float* __restrict__ dY, const float* __restrict__ X,
                                                      float* __restrict__ part_w, float* __restrict__ part_b, int M,
                                                      int rows_per_wg) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m_begin = blockIdx.x * rows_per_wg;
    const int m_end = min(M, m_begin + rows_per_wg);
    float* pw = part_w + (size_t)blockIdx.x * (32 * NF) * (32 * KF);
    float* pb = part_b ? part_b + (size_t)blockIdx.x * (32 * NF) : nullptr;
    if (wave == 0) wgrad_wave<NF, KF, 0>(dY, X, pw, pb, m_begin, m_end, lane);
    else if (wave == 1) wgrad_wave<NF, KF, 1>(dY, X, pw, pb, m_begin, m_end, lane);
    else if (wave == 2) wgrad_wave<NF, KF, 2>(dY, X, pw, pb, m_begin, m_end, lane);
    else wgrad_wave<NF, KF, 3>(dY, X, pw, pb, m_begin, m_end, lane);
}

// out[q] = sum over the slices of part[s * count + q], four values per thread.  A block owns 32 float4 columns and
// eight lanes per column: lane g adds slices g, g + 8, ... (eight loads in flight), the eight sums are added in lane
// order through LDS - a fixed order whatever the timing.
__global__ __launch_bounds__(256) void k_sum_slices(const float* __restrict__ part, float* __restrict__ out, int count4,
                                                    int slices) {
    __shared__ float4 red[8][32];
    const int col = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int q = blockIdx.x * 32 + col;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < count4) {
        const float4* p = reinterpret_cast<const float4*>(part) + q;
        for (int s0 = g; s0 < slices; s0 += 64) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(size_t)min(s0 + 8 * u, slices - 1) * count4];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (s0 + 8 * u < slices) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
    }
    red[g][col] = acc;
    __syncthreads();
    if (g == 0 && q < count4) {
        float4 t = red[0][col];
#pragma unroll
        for (int u = 1; u < 8; ++u) { t.x += red[u][col].x; t.y += red[u][col].y; t.z += red[u][col].z; t.w += red[u][col].w; }
        reinterpret_cast<float4*>(out)[q] = t;
    }
}

static int wgrad_slices(int M) {
    // one slice per CU where the rows allow it; a slice is a multiple of 8 rows (4 waves x 2 rows) and at least 64
    int rows = (M + 255) / 256;
    rows = (rows + 7) / 8 * 8;
    if (rows < 64) rows = 64;
    return rows;
}
size_t linear_wgrad_workspace_floats(int M, int N, int K) {
    const int rows = wgrad_slices(M);
    const size_t slices = (size_t)(M + rows - 1) / rows;
    return slices * ((size_t)N * K + N);
}
bool linear_wgrad_supported(int M, int N, int K) { return M > 0 && K == 64 && (N == 64 || N == 128 || N == 192); }

hipError_t launch_linear_wgrad(const float* dy, const float* x, float* dw, float* db, int M, int N, int K, float* workspace,
                               hipStream_t stream) {
    if (!linear_wgrad_supported(M, N, K)) return hipErrorInvalidValue;
    const int rows = wgrad_slices(M);
    const int slices = (M + rows - 1) / rows;
    float* part_w = workspace;
    float* part_b = db ? workspace + (size_t)slices * N * K : nullptr;
    if (N == 192) hipLaunchKernelGGL((k_linear_wgrad<6, 2>), dim3(slices), dim3(256), 0, stream, dy, x, part_w, part_b, M, rows);
    else if (N == 128) hipLaunchKernelGGL((k_linear_wgrad<4, 2>), dim3(slices), dim3(256), 0, stream, dy, x, part_w, part_b, M, rows);
    else hipLaunchKernelGGL((k_linear_wgrad<2, 2>), dim3(slices), dim3(256), 0, stream, dy, x, part_w, part_b, M, rows);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_sum_slices, dim3((N * K / 4 + 31) / 32), dim3(256), 0, stream, part_w, dw, N * K / 4, slices);
    if (db) hipLaunchKernelGGL(k_sum_slices, dim3((N / 4 + 31) / 32), dim3(256), 0, stream, part_b, db, N / 4, slices);
    return hipGetLastError();
}


// ------------------------------------------------------------------------------------------------------------------
// The entry of an attention block in the training step (ImageSelfAttention.forward, modules_DANRA_conditional.py:93-98):
// NCHW feature map -> tokens [B][L][C] -> LayerNorm over C, as ONE kernel in either direction, so that neither the
// transposed copy of the map nor a strided view of it exists.  A workgroup owns TL tokens of one image (TL = 32, or all L
// tokens of a smaller map): the map's values for those tokens go through an LDS tile [TL][C + 1] (read along the pixels,
// written along the channels).
//
// forward : normed[b][l][c] = (x[b][c][l] - mean) * rstd * gamma[c] + beta[c];  mean / rstd [B*L] are kept.
// backward: dx[b][c][l] = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat)),  g = dn * gamma;
//           part[tile][0][c] = sum_t dn * xhat, part[tile][1][c] = sum_t dn   (summed over the tiles by k_sum_slices)
namespace {
struct LnTokGeom {
    int C, L, TL, tiles_per_img;
};
bool ln_tok_geom(int B, int C, int L, LnTokGeom& g) {
    if (B <= 0 || C <= 0 || L <= 0 || C % 4 || C > 512) return false;
    if (L % 32 == 0) g.TL = 32;
    else if (L >= 4 && L < 32 && (L & (L - 1)) == 0) g.TL = L;  // (a token's channel sums stay inside one wave)
    else return false;
    if ((size_t)2 * g.TL * (C + 1) * sizeof(float) > 64 * 1024) return false;
    g.C = C; g.L = L; g.tiles_per_img = L / g.TL;
    return (long long)B * C * L < (1LL << 31);
}
}  // namespace

__global__ __launch_bounds__(256) void k_ln_tokens_fwd(const float* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ normed,
                                                       float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                       const LnTokGeom g, float eps) {
    extern __shared__ float lds[];
    const int C = g.C, L = g.L, TL = g.TL, LD = C + 1;
    float* tile = lds;               // [TL][C + 1]
    float* stat = lds + TL * LD;     // [TL][2]
    const int tid = threadIdx.x;
    const int b = blockIdx.x / g.tiles_per_img, l0 = (blockIdx.x - b * g.tiles_per_img) * TL;
    const float* xb = x + (size_t)b * C * L + l0;
    for (int e = tid; e < TL * C; e += 256) {
        const int c = e / TL, t = e - c * TL;
        tile[t * LD + c] = xb[(size_t)c * L + t];
    }
    __syncthreads();
    const int TPT = 256 / TL;  // threads per token: consecutive lanes of one wave
    const int t = tid / TPT, j = tid - t * TPT;
    float s = 0.f;
    for (int c = j; c < C; c += TPT) s += tile[t * LD + c];
    for (int o = 1; o < TPT; o <<= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float q = 0.f;
    for (int c = j; c < C; c += TPT) {
        const float d = tile[t * LD + c] - mean;
        q += d * d;
    }
    for (int o = 1; o < TPT; o <<= 1) q += __shfl_xor(q, o);
    const float rstd = 1.0f / sqrtf(q / (float)C + eps);
    if (j == 0) {
        stat[2 * t] = mean;
        stat[2 * t + 1] = rstd;
        mean_out[(size_t)b * L + l0 + t] = mean;
        rstd_out[(size_t)b * L + l0 + t] = rstd;
    }
    __syncthreads();
    float* nb = normed + ((size_t)b * L + l0) * C;
    for (int e = tid; e < TL * C; e += 256) {
        const int tt = e / C, c = e - tt * C;
        nb[e] = (tile[tt * LD + c] - stat[2 * tt]) * stat[2 * tt + 1] * gamma[c] + beta[c];
    }
}

__global__ __launch_bounds__(256) void k_ln_tokens_bwd(const float* __restrict__ dn, const float* __restrict__ x,
                                                       const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                       const float* __restrict__ gamma, float* __restrict__ dx,
                                                       float* __restrict__ part, const LnTokGeom g) {
    extern __shared__ float lds[];
    const int C = g.C, L = g.L, TL = g.TL, LD = C + 1;
    float* xh = lds;                 // [TL][C + 1]: xhat, then dx
    float* dt = lds + TL * LD;       // [TL][C + 1]: dn
    __shared__ float stat[32][2];
    const int tid = threadIdx.x;
    const int b = blockIdx.x / g.tiles_per_img, l0 = (blockIdx.x - b * g.tiles_per_img) * TL;
    const float* xb = x + (size_t)b * C * L + l0;
    const float* mb = mean_in + (size_t)b * L + l0;
    const float* rb = rstd_in + (size_t)b * L + l0;
    for (int e = tid; e < TL * C; e += 256) {
        const int c = e / TL, t = e - c * TL;
        xh[t * LD + c] = (xb[(size_t)c * L + t] - mb[t]) * rb[t];
    }
    const float* db = dn + ((size_t)b * L + l0) * C;
    for (int e = tid; e < TL * C; e += 256) {
        const int tt = e / C, c = e - tt * C;
        dt[tt * LD + c] = db[e];
    }
    __syncthreads();
    // per-channel sums over this tile's tokens (fixed order)
    for (int c = tid; c < C; c += 256) {
        float dg = 0.f, dbv = 0.f;
        for (int tt = 0; tt < TL; ++tt) {
            const float d = dt[tt * LD + c];
            dg += d * xh[tt * LD + c];
            dbv += d;
        }
        part[(size_t)blockIdx.x * 2 * C + c] = dg;
        part[(size_t)blockIdx.x * 2 * C + C + c] = dbv;
    }
    // per-token sums over the channels
    const int TPT = 256 / TL;
    const int t = tid / TPT, j = tid - t * TPT;
    float s1 = 0.f, s2 = 0.f;
    for (int c = j; c < C; c += TPT) {
        const float gv = dt[t * LD + c] * gamma[c];
        s1 += gv;
        s2 += gv * xh[t * LD + c];
    }
    for (int o = 1; o < TPT; o <<= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
    }
    if (j == 0) {
        stat[t][0] = s1 / (float)C;
        stat[t][1] = s2 / (float)C;
    }
    __syncthreads();
    float* dxb = dx + (size_t)b * C * L + l0;
    for (int e = tid; e < TL * C; e += 256) {
        const int c = e / TL, tt = e - c * TL;
        const float gv = dt[tt * LD + c] * gamma[c];
        dxb[(size_t)c * L + tt] = rb[tt] * (gv - stat[tt][0] - xh[tt * LD + c] * stat[tt][1]);
    }
}

// dst[b][c][l] = add[b][c][l] + src[b][l][c]: the exit of an attention block (tokens + residual back to NCHW).
__global__ __launch_bounds__(256) void k_tokens_to_nchw_add(const float* __restrict__ src, const float* __restrict__ add,
                                                            float* __restrict__ dst, int L, int C) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int l0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* s = src + (size_t)b * L * C;
    for (int i = ty; i < 32; i += 8)
        if (l0 + i < L && c0 + tx < C) tile[i][tx] = s[(size_t)(l0 + i) * C + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < C && l0 + tx < L) {
            const size_t o = ((size_t)b * C + c0 + i) * L + l0 + tx;
            dst[o] = (add ? add[o] : 0.f) + tile[tx][i];
        }
}

bool ln_tokens_supported(int B, int C, int L) {
    LnTokGeom g;
    return ln_tok_geom(B, C, L, g);
}
size_t ln_tokens_workspace_floats(int B, int C, int L) {
    LnTokGeom g;
    return ln_tok_geom(B, C, L, g) ? (size_t)B * g.tiles_per_img * 2 * C : 0;
}
hipError_t launch_ln_tokens_fwd(const float* x, const float* gamma, const float* beta, float* normed, float* mean, float* rstd,
                                int B, int C, int L, float eps, hipStream_t stream) {
    LnTokGeom g;
    if (!ln_tok_geom(B, C, L, g)) return hipErrorInvalidValue;
    const size_t lds = ((size_t)g.TL * (C + 1) + 2 * g.TL) * sizeof(float);
    hipLaunchKernelGGL(k_ln_tokens_fwd, dim3(B * g.tiles_per_img), dim3(256), lds, stream, x, gamma, beta, normed, mean, rstd, g,
                       eps);
    return hipGetLastError();
}
hipError_t launch_ln_tokens_bwd(const float* dn, const float* x, const float* mean, const float* rstd, const float* gamma,
                                float* dx, float* dgamma_dbeta, int B, int C, int L, float* workspace, hipStream_t stream) {
    LnTokGeom g;
    if (!ln_tok_geom(B, C, L, g)) return hipErrorInvalidValue;
    const size_t lds = (size_t)2 * g.TL * (C + 1) * sizeof(float);
    const int tiles = B * g.tiles_per_img;
    hipLaunchKernelGGL(k_ln_tokens_bwd, dim3(tiles), dim3(256), lds, stream, dn, x, mean, rstd, gamma, dx, workspace, g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_sum_slices, dim3((2 * C / 4 + 31) / 32), dim3(256), 0, stream, workspace, dgamma_dbeta, 2 * C / 4, tiles);
    return hipGetLastError();
}
hipError_t launch_tokens_to_nchw_add(const float* tokens, const float* add, float* out, int B, int C, int L, hipStream_t stream) {
    hipLaunchKernelGGL(k_tokens_to_nchw_add, dim3((C + 31) / 32, (L + 31) / 32, B), dim3(256), 0, stream, tokens, add, out, L, C);
    return hipGetLastError();
}

}  // namespace danra
