// The scaled-dot-product core of the short-sequence attention blocks for training (the deep levels of the network:
// 64 tokens with head_dim 32, 16 with 64, 4 with 128 at a 64x64 input), forward and backward: softmax(q k^T / sqrt(hd)) v
// per (image, head), with q, k, v [B*heads][L][hd] and the output in token order [B][L][heads*hd] (what the block's output
// projection reads).  One workgroup per (image, head); q, k, v (and dO) and the L x L matrices live in LDS; plain fp32
// multiply-adds: a pair costs 10 L^2 hd flops - a third of a megaflop at most - and the launches are bound by their latency,
// not by arithmetic.  The backward kernel forms the probabilities again from q and k instead of reading saved ones.
//   dV = P^T dO,  dP = dO V^T,  dS = P o (dP - rowsum(P o dP)),  dQ = dS K / sqrt(hd),  dK = dS^T Q / sqrt(hd)
#include "danra_kernels.h"

namespace danra {

namespace {

constexpr int kSmallThreads = 256;

// rows of hd floats are padded by one float in LDS (a column walk then steps through the banks)
__device__ __forceinline__ int pad(int hd) { return hd + 1; }

// S[i][j] = scale * sum_d Q[i][d] K[j][d], then each row's softmax in place (one thread per row)
__device__ void scores_softmax(const float* Qs, const float* Ks, float* S, int L, int hd, float scale) {
    const int tid = threadIdx.x, ld = pad(hd), ls = L + 1;
    for (int e = tid; e < L * L; e += kSmallThreads) {
        const int i = e / L, j = e - i * L;
        float a = 0.f;
        for (int d = 0; d < hd; ++d) a = fmaf(Qs[i * ld + d], Ks[j * ld + d], a);
        S[i * ls + j] = a * scale;
    }
    __syncthreads();
    if (tid < L) {
        float* row = S + tid * ls;
        float m = row[0];
        for (int j = 1; j < L; ++j) m = fmaxf(m, row[j]);
        float sum = 0.f;
        for (int j = 0; j < L; ++j) {
            const float p = expf(row[j] - m);
            row[j] = p;
            sum += p;
        }
        const float inv = 1.0f / sum;
        for (int j = 0; j < L; ++j) row[j] *= inv;
    }
    __syncthreads();
}

__device__ void load_rows(const float* src, float* dst, int L, int hd) {   // [L][hd] -> padded rows
    for (int e = threadIdx.x; e < L * hd; e += kSmallThreads) {
        const int i = e / hd, d = e - i * hd;
        dst[i * pad(hd) + d] = src[e];
    }
}

}  // namespace

__global__ __launch_bounds__(kSmallThreads) void k_attn_small_fwd(const float* __restrict__ Q, const float* __restrict__ K,
                                                                  const float* __restrict__ V, float* __restrict__ O, int heads,
                                                                  int L, int hd, float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int ld = pad(hd), ls = L + 1;
    float *Qs = sm, *Ks = Qs + L * ld, *Vs = Ks + L * ld, *S = Vs + L * ld;
    const int bh = blockIdx.x, b = bh / heads, head = bh - b * heads;
    const size_t base = (size_t)bh * L * hd;
    load_rows(Q + base, Qs, L, hd);
    load_rows(K + base, Ks, L, hd);
    load_rows(V + base, Vs, L, hd);
    __syncthreads();
    scores_softmax(Qs, Ks, S, L, hd, scale);
    for (int e = threadIdx.x; e < L * hd; e += kSmallThreads) {
        const int i = e / hd, d = e - i * hd;
        float a = 0.f;
        for (int j = 0; j < L; ++j) a = fmaf(S[i * ls + j], Vs[j * ld + d], a);
        O[((size_t)b * L + i) * (heads * hd) + head * hd + d] = a;
    }
}

__global__ __launch_bounds__(kSmallThreads) void k_attn_small_bwd(const float* __restrict__ Q, const float* __restrict__ K,
                                                                  const float* __restrict__ V, const float* __restrict__ dO,
                                                                  float* __restrict__ dQ, float* __restrict__ dK,
                                                                  float* __restrict__ dV, int heads, int L, int hd, float scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, ld = pad(hd), ls = L + 1;
    float *Qs = sm, *Ks = Qs + L * ld, *Vs = Ks + L * ld, *Gs = Vs + L * ld, *P = Gs + L * ld, *dS = P + L * ls;
    const int bh = blockIdx.x, b = bh / heads, head = bh - b * heads;
    const size_t base = (size_t)bh * L * hd;
    load_rows(Q + base, Qs, L, hd);
    load_rows(K + base, Ks, L, hd);
    load_rows(V + base, Vs, L, hd);
    for (int e = tid; e < L * hd; e += kSmallThreads) {   // dO arrives in token order
        const int i = e / hd, d = e - i * hd;
        Gs[i * ld + d] = dO[((size_t)b * L + i) * (heads * hd) + head * hd + d];
    }
    __syncthreads();
    scores_softmax(Qs, Ks, P, L, hd, scale);
    // dP, then dS = P o (dP - rowsum(P o dP)) in place
    for (int e = tid; e < L * L; e += kSmallThreads) {
        const int i = e / L, j = e - i * L;
        float a = 0.f;
        for (int d = 0; d < hd; ++d) a = fmaf(Gs[i * ld + d], Vs[j * ld + d], a);
        dS[i * ls + j] = a;
    }
    __syncthreads();
    if (tid < L) {
        float delta = 0.f;
        for (int j = 0; j < L; ++j) delta = fmaf(P[tid * ls + j], dS[tid * ls + j], delta);
        for (int j = 0; j < L; ++j) dS[tid * ls + j] = P[tid * ls + j] * (dS[tid * ls + j] - delta);
    }
    __syncthreads();
    for (int e = tid; e < L * hd; e += kSmallThreads) {
        const int r = e / hd, d = e - r * hd;
        float av = 0.f, aq = 0.f, ak = 0.f;
        for (int j = 0; j < L; ++j) {
            av = fmaf(P[j * ls + r], Gs[j * ld + d], av);     // dV[r][d] = sum_i P[i][r] dO[i][d]
            aq = fmaf(dS[r * ls + j], Ks[j * ld + d], aq);    // dQ[r][d] = sum_j dS[r][j] K[j][d]
            ak = fmaf(dS[j * ls + r], Qs[j * ld + d], ak);    // dK[r][d] = sum_i dS[i][r] Q[i][d]
        }
        dV[base + e] = av;
        dQ[base + e] = aq * scale;
        dK[base + e] = ak * scale;
    }
}

// 1 <= L <= 64 tokens, head_dim a multiple of 4 up to 128, at most 2048 values per (image, head) operand
bool attention_small_train_supported(int B, int heads, int L, int hd) {
    return B > 0 && heads > 0 && L >= 1 && L <= 64 && hd >= 4 && hd <= 128 && hd % 4 == 0 && L * hd <= 2048 &&
           (long long)B * heads < (1LL << 24);
}

static size_t small_lds(int L, int hd, bool bwd) {
    return ((size_t)(bwd ? 4 : 3) * L * (hd + 1) + (size_t)(bwd ? 2 : 1) * L * (L + 1)) * sizeof(float);
}

hipError_t launch_attention_small_fwd(const float* q, const float* k, const float* v, float* out, int B, int heads, int L, int hd,
                                      hipStream_t stream) {
    if (!attention_small_train_supported(B, heads, L, hd)) return hipErrorInvalidValue;
    const size_t lds = small_lds(L, hd, false);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_small_fwd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_attn_small_fwd, dim3(B * heads), dim3(kSmallThreads), lds, stream, q, k, v, out, heads, L, hd,
                       1.0f / sqrtf((float)hd));
    return hipGetLastError();
}

hipError_t launch_attention_small_bwd(const float* q, const float* k, const float* v, const float* dout, float* dq, float* dk,
                                      float* dv, int B, int heads, int L, int hd, hipStream_t stream) {
    if (!attention_small_train_supported(B, heads, L, hd)) return hipErrorInvalidValue;
    const size_t lds = small_lds(L, hd, true);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_small_bwd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_attn_small_bwd, dim3(B * heads), dim3(kSmallThreads), lds, stream, q, k, v, dout, dq, dk, dv, heads, L, hd,
                       1.0f / sqrtf((float)hd));
    return hipGetLastError();
}

}  // namespace danra
