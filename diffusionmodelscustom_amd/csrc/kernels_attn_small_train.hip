// The scaled-dot-product core of the short-sequence attention blocks for training (the deep levels of the network:
// 64 tokens with head_dim 32, 16 with 64, 4 with 128 at a 64x64 input), forward and backward: softmax(q k^T / sqrt(hd)) v
// per (image, head), with q, k, v [B*heads][L][hd] and the output in token order [B][L][heads*hd] (what the block's output
// projection reads).  One workgroup per (image, head); q, k, v (and dO) and the L x L matrices live in LDS; plain fp32
// multiply-adds: a pair costs 10 L^2 hd flops - 1.3 megaflops at most - and the launches are bound by their LDS reads and
// their latency, not by arithmetic.  The backward kernel forms the probabilities again from q and k instead of reading saved ones.
//   dV = P^T dO,  dP = dO V^T,  dS = P o (dP - rowsum(P o dP)),  dQ = dS K / sqrt(hd),  dK = dS^T Q / sqrt(hd)
#include "danra_kernels.h"

namespace danra {

namespace {

constexpr int kSmallThreads = 256;

// rows of hd floats are padded by four floats in LDS: 16-byte reads of four consecutive columns stay aligned, and rows
// 144 / 272 / 528 bytes apart put eight consecutive rows on all 32 banks
__device__ __forceinline__ int pad(int hd) { return hd + 4; }

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

__device__ __forceinline__ float dot4(const float4 a, const float4 b, float acc) {
    return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, fmaf(a.x, b.x, acc))));
}

// out[i][j] = scale * sum_d A[i][d] B[j][d] (rows of L + 1 floats): one element per thread and pass, four columns per read
__device__ void rows_dot_rows(const float* As, const float* Bs, float* out, int L, int hd, float scale) {
    const int ld = pad(hd), ls = L + 1;
    for (int e = threadIdx.x; e < L * L; e += kSmallThreads) {
        const int i = e / L, j = e - i * L;
        float a = 0.f;
        for (int d = 0; d < hd; d += 4) a = dot4(ld4(As + i * ld + d), ld4(Bs + j * ld + d), a);
        out[i * ls + j] = a * scale;
    }
}

// S[i][j] = scale * sum_d Q[i][d] K[j][d], then each row's softmax in place (four lanes per row)
__device__ void scores_softmax(const float* Qs, const float* Ks, float* S, int L, int hd, float scale) {
    const int tid = threadIdx.x, ls = L + 1;
    rows_dot_rows(Qs, Ks, S, L, hd, scale);
    __syncthreads();
    {
        const int r = tid >> 2, q = tid & 3;          // (rows past L idle along: the shuffles below stay inside a group of four lanes)
        float* row = S + min(r, L - 1) * ls;
        float m = -INFINITY;
        for (int j = q; j < L; j += 4) m = fmaxf(m, row[j]);
        m = fmaxf(m, __shfl_xor(m, 1));
        m = fmaxf(m, __shfl_xor(m, 2));
        float sum = 0.f, e[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int j = q + 4 * k;
            e[k] = j < L ? expf(row[j] - m) : 0.f;
            sum += e[k];
        }
        sum += __shfl_xor(sum, 1);
        sum += __shfl_xor(sum, 2);
        const float inv = 1.0f / sum;
        if (r < L) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int j = q + 4 * k;
                if (j < L) row[j] = e[k] * inv;
            }
        }
    }
    __syncthreads();
}

// L rows of hd floats, `rs` floats apart -> padded rows, four floats at a time
__device__ void load_rows(const float* src, float* dst, int L, int hd, int rs) {
    const int hq = hd >> 2;
    for (int e = threadIdx.x; e < L * hq; e += kSmallThreads) {
        const int i = e / hq, dq = e - i * hq;
        *reinterpret_cast<float4*>(dst + i * pad(hd) + 4 * dq) = ld4(src + (size_t)i * rs + 4 * dq);
    }
}

}  // namespace

// A thread of the output loops owns four consecutive columns of one row: per step of a sum one 16-byte read of the
// streamed operand serves four multiply-adds.
__global__ __launch_bounds__(kSmallThreads) void k_attn_small_fwd(const float* __restrict__ Q, const float* __restrict__ K,
                                                                  const float* __restrict__ V, float* __restrict__ O, int heads,
                                                                  int L, int hd, float scale, const QkvLayout lay) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int ld = pad(hd), ls = L + 1, hq = hd >> 2;
    float *Qs = sm, *Ks = Qs + L * ld, *Vs = Ks + L * ld, *S = Vs + L * ld;
    const int bh = blockIdx.x, b = bh / heads, head = bh - b * heads;
    const size_t base = (size_t)b * lay.img + (size_t)head * lay.hs;
    load_rows(Q + base, Qs, L, hd, lay.rs);
    load_rows(K + base, Ks, L, hd, lay.rs);
    load_rows(V + base, Vs, L, hd, lay.rs);
    __syncthreads();
    scores_softmax(Qs, Ks, S, L, hd, scale);
    for (int e = threadIdx.x; e < L * hq; e += kSmallThreads) {
        const int i = e / hq, dq = e - i * hq;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < L; ++j) {
            const float p = S[i * ls + j];
            const float4 v = ld4(Vs + j * ld + 4 * dq);
            a.x = fmaf(p, v.x, a.x); a.y = fmaf(p, v.y, a.y); a.z = fmaf(p, v.z, a.z); a.w = fmaf(p, v.w, a.w);
        }
        *reinterpret_cast<float4*>(O + ((size_t)b * L + i) * (heads * hd) + head * hd + 4 * dq) = a;
    }
}

__global__ __launch_bounds__(kSmallThreads) void k_attn_small_bwd(const float* __restrict__ Q, const float* __restrict__ K,
                                                                  const float* __restrict__ V, const float* __restrict__ dO,
                                                                  float* __restrict__ dQ, float* __restrict__ dK,
                                                                  float* __restrict__ dV, int heads, int L, int hd, float scale,
                                                                  const QkvLayout lay) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, ld = pad(hd), ls = L + 1, hq = hd >> 2;
    float *Qs = sm, *Ks = Qs + L * ld, *Vs = Ks + L * ld, *Gs = Vs + L * ld, *P = Gs + L * ld, *dS = P + L * ls;
    const int bh = blockIdx.x, b = bh / heads, head = bh - b * heads;
    const size_t base = (size_t)b * lay.img + (size_t)head * lay.hs;
    load_rows(Q + base, Qs, L, hd, lay.rs);
    load_rows(K + base, Ks, L, hd, lay.rs);
    load_rows(V + base, Vs, L, hd, lay.rs);
    for (int e = tid; e < L * hq; e += kSmallThreads) {   // dO arrives in token order
        const int i = e / hq, dq = e - i * hq;
        *reinterpret_cast<float4*>(Gs + i * ld + 4 * dq) = ld4(dO + ((size_t)b * L + i) * (heads * hd) + head * hd + 4 * dq);
    }
    __syncthreads();
    scores_softmax(Qs, Ks, P, L, hd, scale);
    rows_dot_rows(Gs, Vs, dS, L, hd, 1.0f);   // dP
    __syncthreads();
    {   // dS = P o (dP - rowsum(P o dP)) in place, four lanes per row
        const int r = tid >> 2, q = tid & 3, rr = min(r, L - 1);
        float delta = 0.f;
        for (int j = q; j < L; j += 4) delta = fmaf(P[rr * ls + j], dS[rr * ls + j], delta);
        delta += __shfl_xor(delta, 1);
        delta += __shfl_xor(delta, 2);
        if (r < L)
            for (int j = q; j < L; j += 4) dS[r * ls + j] = P[r * ls + j] * (dS[r * ls + j] - delta);
    }
    __syncthreads();
    for (int e = tid; e < L * hq; e += kSmallThreads) {
        const int r = e / hq, dq = e - r * hq;
        float4 av = make_float4(0.f, 0.f, 0.f, 0.f), aq = av, ak = av;
        for (int j = 0; j < L; ++j) {
            const float p = P[j * ls + r];      // dV[r][d] = sum_i P[i][r] dO[i][d]
            const float s = dS[r * ls + j];     // dQ[r][d] = sum_j dS[r][j] K[j][d]
            const float t = dS[j * ls + r];     // dK[r][d] = sum_i dS[i][r] Q[i][d]
            const float4 g = ld4(Gs + j * ld + 4 * dq), k = ld4(Ks + j * ld + 4 * dq), qv = ld4(Qs + j * ld + 4 * dq);
            av.x = fmaf(p, g.x, av.x); av.y = fmaf(p, g.y, av.y); av.z = fmaf(p, g.z, av.z); av.w = fmaf(p, g.w, av.w);
            aq.x = fmaf(s, k.x, aq.x); aq.y = fmaf(s, k.y, aq.y); aq.z = fmaf(s, k.z, aq.z); aq.w = fmaf(s, k.w, aq.w);
            ak.x = fmaf(t, qv.x, ak.x); ak.y = fmaf(t, qv.y, ak.y); ak.z = fmaf(t, qv.z, ak.z); ak.w = fmaf(t, qv.w, ak.w);
        }
        const size_t o = base + (size_t)r * lay.rs + 4 * dq;
        *reinterpret_cast<float4*>(dV + o) = av;
        *reinterpret_cast<float4*>(dQ + o) = make_float4(aq.x * scale, aq.y * scale, aq.z * scale, aq.w * scale);
        *reinterpret_cast<float4*>(dK + o) = make_float4(ak.x * scale, ak.y * scale, ak.z * scale, ak.w * scale);
    }
}

// 1 <= L <= 64 tokens, head_dim a multiple of 4 up to 128, at most 2048 values per (image, head) operand
bool attention_small_train_supported(int B, int heads, int L, int hd) {
    return B > 0 && heads > 0 && L >= 1 && L <= 64 && hd >= 4 && hd <= 128 && hd % 4 == 0 && L * hd <= 2048 &&
           (long long)B * heads < (1LL << 24);
}

static size_t small_lds(int L, int hd, bool bwd) {
    return ((size_t)(bwd ? 4 : 3) * L * (hd + 4) + (size_t)(bwd ? 2 : 1) * L * (L + 1)) * sizeof(float);
}

static hipError_t small_fwd(const float* q, const float* k, const float* v, float* out, int B, int heads, int L, int hd,
                            const QkvLayout lay, hipStream_t stream) {
    if (!attention_small_train_supported(B, heads, L, hd)) return hipErrorInvalidValue;
    const size_t lds = small_lds(L, hd, false);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_small_fwd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_attn_small_fwd, dim3(B * heads), dim3(kSmallThreads), lds, stream, q, k, v, out, heads, L, hd,
                       1.0f / sqrtf((float)hd), lay);
    return hipGetLastError();
}

static hipError_t small_bwd(const float* q, const float* k, const float* v, const float* dout, float* dq, float* dk, float* dv,
                            int B, int heads, int L, int hd, const QkvLayout lay, hipStream_t stream) {
    if (!attention_small_train_supported(B, heads, L, hd)) return hipErrorInvalidValue;
    const size_t lds = small_lds(L, hd, true);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_small_bwd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_attn_small_bwd, dim3(B * heads), dim3(kSmallThreads), lds, stream, q, k, v, dout, dq, dk, dv, heads, L, hd,
                       1.0f / sqrtf((float)hd), lay);
    return hipGetLastError();
}

hipError_t launch_attention_small_fwd(const float* q, const float* k, const float* v, float* out, int B, int heads, int L, int hd,
                                      hipStream_t stream) {
    return small_fwd(q, k, v, out, B, heads, L, hd, QkvLayout{hd, (long long)heads * L * hd, L * hd}, stream);
}

hipError_t launch_attention_small_bwd(const float* q, const float* k, const float* v, const float* dout, float* dq, float* dk,
                                      float* dv, int B, int heads, int L, int hd, hipStream_t stream) {
    return small_bwd(q, k, v, dout, dq, dk, dv, B, heads, L, hd, QkvLayout{hd, (long long)heads * L * hd, L * hd}, stream);
}

// q, k, v read from, and dq, dk, dv written to, the output layout of a fused QKV projection: qkv, dqkv [B][L][3][heads * hd]
hipError_t launch_attention_small_qkv_fwd(const float* qkv, float* out, int B, int heads, int L, int hd, hipStream_t stream) {
    const int C = heads * hd;
    return small_fwd(qkv, qkv + C, qkv + 2 * C, out, B, heads, L, hd, QkvLayout{3 * C, (long long)L * 3 * C, hd}, stream);
}

hipError_t launch_attention_small_qkv_bwd(const float* qkv, const float* dout, float* dqkv, int B, int heads, int L, int hd,
                                          hipStream_t stream) {
    const int C = heads * hd;
    return small_bwd(qkv, qkv + C, qkv + 2 * C, dout, dqkv, dqkv + C, dqkv + 2 * C, B, heads, L, hd,
                     QkvLayout{3 * C, (long long)L * 3 * C, hd}, stream);
}

}  // namespace danra
