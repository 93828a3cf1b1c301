// Backward pass of the head_dim-16 flash attention core (the training step's counterpart of k_flash_hd16).
//
// Forward, per (image, head):  S = (Q K^T) / 4,  P = softmax_rows(S),  O = P V.  k_flash_hd16 also leaves
// LSE2[q] = log2(sum_k exp(S[q][k])) (the row's log-sum-exp in the exp2 domain), so that the backward kernels
// rebuild any probability tile as P = exp2(S * log2(e) - LSE2) without a second softmax pass.
//
// Backward, with D[q] = sum_d dO[q][d] O[q][d]:
//   dV = P^T dO,   dP = dO V^T,   dS = P o (dP - D),   dQ = dS K / 4,   dK = dS^T Q / 4.
//
// Two kernels, no atomics, so the result is deterministic:
//   k_flash_hd16_bwd_dq  - a wave owns 16*QT queries and walks the keys (the forward's structure): score tiles are
//                          formed transposed (lane = query), -LSE2 and -D enter as the C operands of the matrix
//                          products, dQ^T = K^T dS^T accumulates in registers.  It also writes D for the second kernel.
//   k_flash_hd16_bwd_dkv - a wave owns 16*KT keys and walks the queries: score tiles are formed untransposed
//                          (lane = key) so that the contraction of dV^T = dO^T P and dK^T = Q^T dS runs over the
//                          register index, and dK / dV accumulate in registers.
// fp32 matrix cores (v_mfma_f32_16x16x4_f32) throughout: operand A lane (i, k-slot), operand B lane (j, k-slot),
// result lane (j, g) holds rows 4g .. 4g+3 of column j.
#include "danra_kernels.h"
#include "split_f16.h"

namespace danra {

typedef float f32x4 __attribute__((ext_vector_type(4)));

static constexpr int BT = 64;    // rows of the streamed operand per LDS tile
static constexpr int BROW = 20;  // padded LDS row (floats) of a 16-float row: conflict-free b32 column reads

#define MFMA4(acc, a4, b4)                                                        \
    do {                                                                          \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32((a4).x, (b4).x, acc, 0, 0, 0); \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32((a4).y, (b4).y, acc, 0, 0, 0); \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32((a4).z, (b4).z, acc, 0, 0, 0); \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32((a4).w, (b4).w, acc, 0, 0, 0); \
    } while (0)

// Workgroup id -> (image*heads + head, block within it); all blocks of one (image, head) on one XCD (ids are dealt
// round-robin over the 8 XCDs), so its streamed rows are fetched into one L2.
__device__ __forceinline__ void decode_block(int id, int nblk, int n_bh, int& bh, int& blk) {
    if (n_bh % 8 == 0) {
        bh = (id / (8 * nblk)) * 8 + (id & 7);
        blk = (id >> 3) % nblk;
    } else {
        bh = id / nblk;
        blk = id - bh * nblk;
    }
}

// Q, K, V, dQ: [n_bh][L][16];  O, dO: tokens [B][L][heads*16] at column head*16;  LSE2, Dout: [n_bh][L].
template <int QT>
__global__ __launch_bounds__(256) void k_flash_hd16_bwd_dq(const float* __restrict__ Q, const float* __restrict__ K,
                                                           const float* __restrict__ V, const float* __restrict__ O,
                                                           const float* __restrict__ dO, const float* __restrict__ LSE2,
                                                           float* __restrict__ dQ, float* __restrict__ Dout, int L,
                                                           int heads, int n_bh, float scale, const QkvLayout lay) {
    __shared__ __attribute__((aligned(16))) float Ks[2][BT * BROW];
    __shared__ __attribute__((aligned(16))) float Vs[2][BT * BROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bh, qblk;
    decode_block(blockIdx.x, L / (4 * QT * 16), n_bh, bh, qblk);
    const int q0 = qblk * (4 * QT * 16) + wave * (QT * 16);
    const int lq = lane & 15, g = lane >> 4;
    const int b = bh / heads, head = bh - b * heads, C = heads * 16;
    const float scale_log2e = scale * 1.4426950408889634f;

    const int RS = lay.rs;   // floats between two rows of q, k, v, dq (16, or the row of a fused QKV projection)
    const size_t base = (size_t)b * lay.img + (size_t)head * lay.hs;
    const float* Qb = Q + base;
    const float* Kb = K + base;
    const float* Vb = V + base;

    // resident operands of the wave's queries: lane (q, g) holds columns 4g .. 4g+3 of row q
    float4 qf[QT], dof[QT];
    f32x4 neglse[QT], negd[QT], dq[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int q = q0 + t * 16 + lq;
        qf[t] = *reinterpret_cast<const float4*>(Qb + (size_t)q * RS + 4 * g);
        qf[t].x *= scale_log2e; qf[t].y *= scale_log2e; qf[t].z *= scale_log2e; qf[t].w *= scale_log2e;
        const size_t tok = ((size_t)b * L + q) * C + head * 16 + 4 * g;
        dof[t] = *reinterpret_cast<const float4*>(dO + tok);
        const float4 of = *reinterpret_cast<const float4*>(O + tok);
        float d = dof[t].x * of.x + dof[t].y * of.y + dof[t].z * of.z + dof[t].w * of.w;
        d += __shfl_xor(d, 16);
        d += __shfl_xor(d, 32);
        if (g == 0) Dout[(size_t)bh * L + q] = d;
        const float l2 = LSE2[(size_t)bh * L + q];
        neglse[t] = f32x4{-l2, -l2, -l2, -l2};
        negd[t] = f32x4{-d, -d, -d, -d};
        dq[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const int srow = tid >> 2, sc4 = (tid & 3) * 4;
    float4 kreg = *reinterpret_cast<const float4*>(Kb + (size_t)srow * RS + sc4);
    float4 vreg = *reinterpret_cast<const float4*>(Vb + (size_t)srow * RS + sc4);
    *reinterpret_cast<float4*>(&Ks[0][srow * BROW + sc4]) = kreg;
    *reinterpret_cast<float4*>(&Vs[0][srow * BROW + sc4]) = vreg;
    __syncthreads();

    const int ntiles = L / BT;
    for (int kt = 0; kt < ntiles; ++kt) {
        const int buf = kt & 1;
        {   // unconditional prefetch (the last tile re-reads itself)
            const int nxt = min(kt + 1, ntiles - 1);
            kreg = *reinterpret_cast<const float4*>(Kb + (size_t)(nxt * BT + srow) * RS + sc4);
            vreg = *reinterpret_cast<const float4*>(Vb + (size_t)(nxt * BT + srow) * RS + sc4);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // 16 keys at a time
            const float4 kf = *reinterpret_cast<const float4*>(&Ks[buf][(j * 16 + lq) * BROW + 4 * g]);
            const float4 vf = *reinterpret_cast<const float4*>(&Vs[buf][(j * 16 + lq) * BROW + 4 * g]);
            float kt4[4];  // K^T: lane (d, g) holds K[key 4g + r][d]
#pragma unroll
            for (int r = 0; r < 4; ++r) kt4[r] = Ks[buf][(j * 16 + 4 * g + r) * BROW + lq];
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                f32x4 s = neglse[t], dp = negd[t];
                MFMA4(s, kf, qf[t]);    // S^T log2(e) - LSE2 : lane (q, g) holds keys 4g .. 4g+3
                MFMA4(dp, vf, dof[t]);  // dP^T - D
                f32x4 ds;
#pragma unroll
                for (int r = 0; r < 4; ++r) ds[r] = __builtin_amdgcn_exp2f(s[r]) * dp[r];
#pragma unroll
                for (int r = 0; r < 4; ++r) dq[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kt4[r], ds[r], dq[t], 0, 0, 0);
            }
        }
        *reinterpret_cast<float4*>(&Ks[buf ^ 1][srow * BROW + sc4]) = kreg;
        *reinterpret_cast<float4*>(&Vs[buf ^ 1][srow * BROW + sc4]) = vreg;
        __syncthreads();
    }

#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const float4 o = make_float4(dq[t][0] * scale, dq[t][1] * scale, dq[t][2] * scale, dq[t][3] * scale);
        *reinterpret_cast<float4*>(dQ + base + (size_t)(q0 + t * 16 + lq) * RS + 4 * g) = o;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The same gradient with the three products on the fp16 matrix cores, operands split in two fp16 halves as the forward
// kernel splits them (x = h + 2^-11 l, h rounded towards zero; 22 significant bits, the l l term dropped):
//   S^T  = [Kh | Kl] [Qh | 2^-11 Qh]^T + [Kh | Kl] [2^-11 Ql | 0]^T                 two 32-deep instructions per 16 keys
//   dP^T = [Vh | Vl] [Gh | 2^-11 Gh]^T + [Vh | Vl] [2^-11 Gl | 0]^T,  G = 2^-e dO    (e: so that max |G| of the wave's 16
//          queries lies in [8, 16): gradients are small numbers, and fp16 keeps 22 bits only above 2^-14)
//   dQ^T += Kh^T dSh + Kh^T dSl + 2^-11 Kl^T dSh,  dS = P o (dP - D) split as hi + lo   three instructions per 32 keys
// A tile is multiplied this way when every |K|, |V| staged for it is below 2^15, the wave's own q and dO are finite and
// in range, and 256 max|V| + max|2^-e D| < 2^15 (then no dS leaves the fp16 range); otherwise - per wave and tile - with
// the fp32 instructions of k_flash_hd16_bwd_dq from the fp32 copy of the tile.  K arrives in LDS three times: rows in
// fp32, rows as halves [h4 | l4] x 4, and as the A fragments of the third product ([hi, lo][32-key half][d][g][slot],
// slot 4j + r <-> key 16j + 4g + r: the order in which a lane holds its dS values); V twice.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

template <int QT>
__global__ __launch_bounds__(256) void k_flash_hd16_bwd_dq_h(const float* __restrict__ Q, const float* __restrict__ K,
                                                             const float* __restrict__ V, const float* __restrict__ O,
                                                             const float* __restrict__ dO, const float* __restrict__ LSE2,
                                                             float* __restrict__ dQ, float* __restrict__ Dout, int L,
                                                             int heads, int n_bh, float scale, const QkvLayout lay) {
    __shared__ __attribute__((aligned(16))) float Ks[2][BT * BROW];
    __shared__ __attribute__((aligned(16))) float Vs[2][BT * BROW];
    __shared__ __attribute__((aligned(16))) _Float16 Kh[2][BT * 32];
    __shared__ __attribute__((aligned(16))) _Float16 Vh[2][BT * 32];
    __shared__ __attribute__((aligned(16))) _Float16 Kt[2][2][2][16 * 4 * 8];
    __shared__ int slow_s[2][4];      // per staging wave: a value of its 16 keys leaves the range of the halves
    __shared__ float vmax_s[2][4];    // per staging wave: max |V| of its 16 keys
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bh, qblk;
    decode_block(blockIdx.x, L / (4 * QT * 16), n_bh, bh, qblk);
    const int q0 = qblk * (4 * QT * 16) + wave * (QT * 16);
    const int lq = lane & 15, g = lane >> 4;
    const int b = bh / heads, head = bh - b * heads, C = heads * 16;
    const float scale_log2e = scale * 1.4426950408889634f;
    const int RS = lay.rs;
    const size_t base = (size_t)b * lay.img + (size_t)head * lay.hs;
    const float* Qb = Q + base;
    const float* Kb = K + base;
    const float* Vb = V + base;

    // resident operands of the wave's queries: lane (q, g) holds columns 4g .. 4g+3 of row q, in fp32 and as halves
    float4 qf[QT], dof[QT];
    f16x8 qb1[QT], qb2[QT], gb1[QT], gb2[QT];
    f32x4 neglse[QT], negd[QT], dq[QT], dq2[QT];
    float unscale[QT];
    bool bounded = true;
    float dmax = 0.f;
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int q = q0 + t * 16 + lq;
        qf[t] = *reinterpret_cast<const float4*>(Qb + (size_t)q * RS + 4 * g);
        qf[t].x *= scale_log2e; qf[t].y *= scale_log2e; qf[t].z *= scale_log2e; qf[t].w *= scale_log2e;
        const size_t tok = ((size_t)b * L + q) * C + head * 16 + 4 * g;
        dof[t] = *reinterpret_cast<const float4*>(dO + tok);
        const float4 of = *reinterpret_cast<const float4*>(O + tok);
        float d = dof[t].x * of.x + dof[t].y * of.y + dof[t].z * of.z + dof[t].w * of.w;
        d += __shfl_xor(d, 16);
        d += __shfl_xor(d, 32);
        if (g == 0) Dout[(size_t)bh * L + q] = d;
        // 2^-e: brings the largest |dO| of these 16 queries into [8, 16) (1 where they are all zero or not finite)
        const float m = wave_max(fmaxf(fmaxf(fabsf(dof[t].x), fabsf(dof[t].y)), fmaxf(fabsf(dof[t].z), fabsf(dof[t].w))));
        const int ex = (int)((__float_as_uint(m) >> 23) & 0xff);   // biased exponent of the largest |dO|
        const bool usable = ex >= 5 && ex <= 250;
        const float sc = usable ? __uint_as_float((unsigned)(257 - ex) << 23) : 1.0f;     // 2^(3 - (ex - 127))
        unscale[t] = usable ? __uint_as_float((unsigned)(ex - 3) << 23) : 1.0f;          // 2^((ex - 127) - 3)
        bounded = bounded && (usable || m == 0.f);   // (not finite, or too small / large to rescale: fp32 instructions)
        dof[t].x *= sc; dof[t].y *= sc; dof[t].z *= sc; dof[t].w *= sc;
        d *= sc;
        dmax = fmaxf(dmax, fabsf(d));
        const float l2 = LSE2[(size_t)bh * L + q];
        neglse[t] = f32x4{-l2, -l2, -l2, -l2};
        negd[t] = f32x4{-d, -d, -d, -d};
        dq[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        dq2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float qq[4] = {qf[t].x, qf[t].y, qf[t].z, qf[t].w}, gg[4] = {dof[t].x, dof[t].y, dof[t].z, dof[t].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const _Float16 qh = (_Float16)qq[i], gh = (_Float16)gg[i];
            qb1[t][i] = qh;
            qb1[t][4 + i] = (_Float16)((float)qh * 0.00048828125f);
            qb2[t][i] = (_Float16)(qq[i] - (float)qh);
            qb2[t][4 + i] = (_Float16)0.f;
            gb1[t][i] = gh;
            gb1[t][4 + i] = (_Float16)((float)gh * 0.00048828125f);
            gb2[t][i] = (_Float16)(gg[i] - (float)gh);
            gb2[t][4 + i] = (_Float16)0.f;
            bounded = bounded && (fabsf(qq[i]) < 32768.f);
        }
    }
    dmax = wave_max(dmax);
    const bool wave_slow = __any(!bounded || !(dmax < 3.0e38f)) != 0;

    // staging: K and V rows (one float4 each per thread) and, for the A fragments of the third product, one column vd of
    // the four keys that are neighbours in a lane's slot order
    const int srow = tid >> 2, sc4 = (tid & 3) * 4;
    const int vd = lane >> 2, vg = lane & 3, vj = wave & 1, vhalf = wave >> 1;
    const int vkey = 32 * vhalf + 16 * vj + 4 * vg;
    float4 kreg, vreg, ktreg;
    auto load_kv = [&](int tile) {
        kreg = *reinterpret_cast<const float4*>(Kb + (size_t)(tile * BT + srow) * RS + sc4);
        vreg = *reinterpret_cast<const float4*>(Vb + (size_t)(tile * BT + srow) * RS + sc4);
        const float* kp = Kb + (size_t)(tile * BT + vkey) * RS + vd;
        ktreg = make_float4(kp[0], kp[RS], kp[2 * RS], kp[3 * RS]);
    };
    auto stage = [&](int buf) {
        *reinterpret_cast<float4*>(&Ks[buf][srow * BROW + sc4]) = kreg;
        *reinterpret_cast<float4*>(&Vs[buf][srow * BROW + sc4]) = vreg;
        unsigned h[2], l[2];
        split_scaled_pair_f16(kreg.x, kreg.y, h[0], l[0]);
        split_scaled_pair_f16(kreg.z, kreg.w, h[1], l[1]);
        *reinterpret_cast<u32x4*>(&Kh[buf][srow * 32 + 2 * sc4]) = u32x4{h[0], h[1], l[0], l[1]};
        split_scaled_pair_f16(vreg.x, vreg.y, h[0], l[0]);
        split_scaled_pair_f16(vreg.z, vreg.w, h[1], l[1]);
        *reinterpret_cast<u32x4*>(&Vh[buf][srow * 32 + 2 * sc4]) = u32x4{h[0], h[1], l[0], l[1]};
        split_scaled_pair_f16(ktreg.x, ktreg.y, h[0], l[0]);
        split_scaled_pair_f16(ktreg.z, ktreg.w, h[1], l[1]);
        const int at = (vd * 4 + vg) * 8 + 4 * vj;
        *reinterpret_cast<u32x2*>(&Kt[buf][0][vhalf][at]) = u32x2{h[0], h[1]};
        *reinterpret_cast<u32x2*>(&Kt[buf][1][vhalf][at]) = u32x2{l[0], l[1]};
        // (a wave stages 16 keys x 16 columns of the row forms, and the same 16 keys of the column form)
        const float kmax = fmaxf(fmaxf(fabsf(kreg.x), fabsf(kreg.y)), fmaxf(fabsf(kreg.z), fabsf(kreg.w)));
        const float vmax = fmaxf(fmaxf(fabsf(vreg.x), fabsf(vreg.y)), fmaxf(fabsf(vreg.z), fabsf(vreg.w)));
        const bool out = !(kmax < 32768.f) || !(vmax < 32768.f);
        const int any_out = __any(out);
        const float wmax = wave_max(out ? 0.f : vmax);
        if (lane == 0) {
            slow_s[buf][wave] = any_out ? 1 : 0;
            vmax_s[buf][wave] = wmax;
        }
    };
    load_kv(0);
    stage(0);
    __syncthreads();

    const int ntiles = L / BT;
    for (int kt = 0; kt < ntiles; ++kt) {
        const int buf = kt & 1;
        const int flagged = slow_s[buf][0] | slow_s[buf][1] | slow_s[buf][2] | slow_s[buf][3];
        const float vtile = fmaxf(fmaxf(vmax_s[buf][0], vmax_s[buf][1]), fmaxf(vmax_s[buf][2], vmax_s[buf][3]));
        load_kv(min(kt + 1, ntiles - 1));   // unconditional prefetch (the last tile re-reads itself)
        const float vt = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(vtile)));
        const bool fast = !wave_slow && __builtin_amdgcn_readfirstlane(flagged) == 0 && (256.f * vt + dmax < 32768.f);
        if (fast) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                f32x4 s[QT][2], dp[QT][2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const f16x8 ka = *reinterpret_cast<const f16x8*>(&Kh[buf][((2 * hf + j) * 16 + lq) * 32 + 8 * g]);
                    const f16x8 va = *reinterpret_cast<const f16x8*>(&Vh[buf][((2 * hf + j) * 16 + lq) * 32 + 8 * g]);
#pragma unroll
                    for (int t = 0; t < QT; ++t) {
                        f32x4 a = __builtin_amdgcn_mfma_f32_16x16x32_f16(ka, qb1[t], neglse[t], 0, 0, 0);
                        s[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ka, qb2[t], a, 0, 0, 0);
                        f32x4 c = __builtin_amdgcn_mfma_f32_16x16x32_f16(va, gb1[t], negd[t], 0, 0, 0);
                        dp[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(va, gb2[t], c, 0, 0, 0);
                    }
                }
                const f16x8 kthi = *reinterpret_cast<const f16x8*>(&Kt[buf][0][hf][(lq * 4 + g) * 8]);
                const f16x8 ktlo = *reinterpret_cast<const f16x8*>(&Kt[buf][1][hf][(lq * 4 + g) * 8]);
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    u32x4 sh, sl;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        float ds[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) ds[r] = __builtin_amdgcn_exp2f(s[t][j][r]) * dp[t][j][r];
                        unsigned h0, l0, h1, l1;
                        split_pair_f16(ds[0], ds[1], h0, l0);
                        split_pair_f16(ds[2], ds[3], h1, l1);
                        sh[2 * j] = h0; sh[2 * j + 1] = h1;
                        sl[2 * j] = l0; sl[2 * j + 1] = l1;
                    }
                    const f16x8 dsh = __builtin_bit_cast(f16x8, sh), dsl = __builtin_bit_cast(f16x8, sl);
                    dq2[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ktlo, dsh, dq2[t], 0, 0, 0);
                    dq[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kthi, dsl, dq[t], 0, 0, 0);
                    dq[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kthi, dsh, dq[t], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {  // 16 keys at a time, fp32 instructions (dO and D in the scaled form here too)
                const float4 kf = *reinterpret_cast<const float4*>(&Ks[buf][(j * 16 + lq) * BROW + 4 * g]);
                const float4 vf = *reinterpret_cast<const float4*>(&Vs[buf][(j * 16 + lq) * BROW + 4 * g]);
                float kt4[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) kt4[r] = Ks[buf][(j * 16 + 4 * g + r) * BROW + lq];
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    f32x4 s = neglse[t], dp = negd[t];
                    MFMA4(s, kf, qf[t]);
                    MFMA4(dp, vf, dof[t]);
                    f32x4 ds;
#pragma unroll
                    for (int r = 0; r < 4; ++r) ds[r] = __builtin_amdgcn_exp2f(s[r]) * dp[r];
#pragma unroll
                    for (int r = 0; r < 4; ++r) dq[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kt4[r], ds[r], dq[t], 0, 0, 0);
                }
            }
        }
        stage(buf ^ 1);
        __syncthreads();
    }

#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const float f = scale * unscale[t];
        const float4 o = make_float4((dq[t][0] + dq2[t][0] * 0.00048828125f) * f, (dq[t][1] + dq2[t][1] * 0.00048828125f) * f,
                                     (dq[t][2] + dq2[t][2] * 0.00048828125f) * f, (dq[t][3] + dq2[t][3] * 0.00048828125f) * f);
        *reinterpret_cast<float4*>(dQ + base + (size_t)(q0 + t * 16 + lq) * RS + 4 * g) = o;
    }
}

// Q, K, V, dK, dV: [n_bh][L][16];  dO: tokens [B][L][heads*16];  LSE2, D: [n_bh][L].
template <int KT>
__global__ __launch_bounds__(256) void k_flash_hd16_bwd_dkv(const float* __restrict__ Q, const float* __restrict__ K,
                                                            const float* __restrict__ V, const float* __restrict__ dO,
                                                            const float* __restrict__ LSE2, const float* __restrict__ D,
                                                            float* __restrict__ dK, float* __restrict__ dV, int L,
                                                            int heads, int n_bh, float scale, const QkvLayout lay) {
    __shared__ __attribute__((aligned(16))) float Qs[2][BT * BROW];
    __shared__ __attribute__((aligned(16))) float dOs[2][BT * BROW];
    __shared__ __attribute__((aligned(16))) float NLs[2][BT];  // -LSE2 of the tile's queries
    __shared__ __attribute__((aligned(16))) float NDs[2][BT];  // -D
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bh, kblk;
    decode_block(blockIdx.x, L / (4 * KT * 16), n_bh, bh, kblk);
    const int k0 = kblk * (4 * KT * 16) + wave * (KT * 16);
    const int lq = lane & 15, g = lane >> 4;
    const int b = bh / heads, head = bh - b * heads, C = heads * 16;
    const float scale_log2e = scale * 1.4426950408889634f;

    const int RS = lay.rs;
    const size_t base = (size_t)b * lay.img + (size_t)head * lay.hs;
    const float* Qb = Q + base;
    const float* dOb = dO + (size_t)b * L * C + head * 16;
    const float* Lb = LSE2 + (size_t)bh * L;
    const float* Db = D + (size_t)bh * L;

    // resident operands of the wave's keys: lane (key, g) holds columns 4g .. 4g+3 of row key
    float4 kf[KT], vf[KT];
    f32x4 dk[KT], dv[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        const size_t row = base + (size_t)(k0 + t * 16 + lq) * RS + 4 * g;
        kf[t] = *reinterpret_cast<const float4*>(K + row);
        kf[t].x *= scale_log2e; kf[t].y *= scale_log2e; kf[t].z *= scale_log2e; kf[t].w *= scale_log2e;
        vf[t] = *reinterpret_cast<const float4*>(V + row);
        dk[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        dv[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // staging: 64 queries x 16 floats of Q and dO (one float4 each per thread); -LSE2 and -D by the first 64 threads
    const int srow = tid >> 2, sc4 = (tid & 3) * 4, r64 = tid & 63;
    float4 qreg = *reinterpret_cast<const float4*>(Qb + (size_t)srow * RS + sc4);
    float4 doreg = *reinterpret_cast<const float4*>(dOb + (size_t)srow * C + sc4);
    float lreg = Lb[r64], dreg = Db[r64];
    *reinterpret_cast<float4*>(&Qs[0][srow * BROW + sc4]) = qreg;
    *reinterpret_cast<float4*>(&dOs[0][srow * BROW + sc4]) = doreg;
    if (tid < 64) {
        NLs[0][r64] = -lreg;
        NDs[0][r64] = -dreg;
    }
    __syncthreads();

    const int ntiles = L / BT;
    for (int qt = 0; qt < ntiles; ++qt) {
        const int buf = qt & 1;
        {
            const int nxt = min(qt + 1, ntiles - 1);
            qreg = *reinterpret_cast<const float4*>(Qb + (size_t)(nxt * BT + srow) * RS + sc4);
            doreg = *reinterpret_cast<const float4*>(dOb + (size_t)(nxt * BT + srow) * C + sc4);
            lreg = Lb[nxt * BT + r64];
            dreg = Db[nxt * BT + r64];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // 16 queries at a time
            const float4 qa = *reinterpret_cast<const float4*>(&Qs[buf][(j * 16 + lq) * BROW + 4 * g]);
            const float4 doa = *reinterpret_cast<const float4*>(&dOs[buf][(j * 16 + lq) * BROW + 4 * g]);
            const f32x4 nl = *reinterpret_cast<const f32x4*>(&NLs[buf][j * 16 + 4 * g]);  // -LSE2 of queries 4g .. 4g+3
            const f32x4 nd = *reinterpret_cast<const f32x4*>(&NDs[buf][j * 16 + 4 * g]);
            float qt4[4], dot4[4];  // Q^T, dO^T: lane (d, g) holds row (query 4g + r), column d
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                qt4[r] = Qs[buf][(j * 16 + 4 * g + r) * BROW + lq];
                dot4[r] = dOs[buf][(j * 16 + 4 * g + r) * BROW + lq];
            }
#pragma unroll
            for (int t = 0; t < KT; ++t) {
                f32x4 s = nl, dp = nd;
                MFMA4(s, qa, kf[t]);    // S log2(e) - LSE2 : lane (key, g) holds queries 4g .. 4g+3
                MFMA4(dp, doa, vf[t]);  // dP - D
                f32x4 p, ds;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p[r] = __builtin_amdgcn_exp2f(s[r]);
                    ds[r] = p[r] * dp[r];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dv[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(dot4[r], p[r], dv[t], 0, 0, 0);
                    dk[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(qt4[r], ds[r], dk[t], 0, 0, 0);
                }
            }
        }
        *reinterpret_cast<float4*>(&Qs[buf ^ 1][srow * BROW + sc4]) = qreg;
        *reinterpret_cast<float4*>(&dOs[buf ^ 1][srow * BROW + sc4]) = doreg;
        if (tid < 64) {
            NLs[buf ^ 1][r64] = -lreg;
            NDs[buf ^ 1][r64] = -dreg;
        }
        __syncthreads();
    }

    // lane (key, g) holds dK[key][4g .. 4g+3] and dV[key][4g .. 4g+3]
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        const size_t row = base + (size_t)(k0 + t * 16 + lq) * RS + 4 * g;
        *reinterpret_cast<float4*>(dK + row) = make_float4(dk[t][0] * scale, dk[t][1] * scale, dk[t][2] * scale, dk[t][3] * scale);
        *reinterpret_cast<float4*>(dV + row) = make_float4(dv[t][0], dv[t][1], dv[t][2], dv[t][3]);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// dK and dV with the four products on the fp16 matrix cores (the counterpart of k_flash_hd16_bwd_dq_h): the wave's keys
// are resident as halves ([h | 2^-11 h] and [x - h | 0], K after its multiplication by hd^-1/2 log2 e), the queries
// stream through LDS in tiles of 64, each in five forms: Q and dO rows in fp32 (fall-back), Q and G = 2^-e dO rows as
// halves [h4 | l4] x 4 (A operands of S and dP), and Q and G as the A fragments of the two accumulating products
// ([hi, lo][32-query half][d][g][slot], slot 4j + r <-> query 16j + 4g + r).  e is common to the 64 queries of a tile
// (the products that accumulate dV and dK contract over queries): the staging waves exchange their maxima through LDS,
// one more barrier per tile.  A tile's contribution is accumulated on its own and added to dK / dV times 2^e.
//   S  = [Qh | Ql] [Kh | 2^-11 Kh]^T + [Qh | Ql] [Kl | 0]^T          dP = the same with G and V, C operand -2^-e D
//   dV^T += Gh^T Ph + Gh^T Pl + 2^-11 Gl^T Ph                         dK^T += Qh^T dSh + Qh^T dSl + 2^-11 Ql^T dSh
// Fall-back to the fp32 instructions of k_flash_hd16_bwd_dkv: the whole wave when one of its keys leaves the range of the
// halves, a tile when a staged q or dO does, when the tile's dO cannot be rescaled, or when 256 max|V| + max|2^-e D| >= 2^15.
template <int KT>
__global__ __launch_bounds__(256) void k_flash_hd16_bwd_dkv_h(const float* __restrict__ Q, const float* __restrict__ K,
                                                              const float* __restrict__ V, const float* __restrict__ dO,
                                                              const float* __restrict__ LSE2, const float* __restrict__ D,
                                                              float* __restrict__ dK, float* __restrict__ dV, int L,
                                                              int heads, int n_bh, float scale, const QkvLayout lay) {
    __shared__ __attribute__((aligned(16))) float Qs[2][BT * BROW];
    __shared__ __attribute__((aligned(16))) float dOs[2][BT * BROW];
    __shared__ __attribute__((aligned(16))) _Float16 Qh[2][BT * 32];
    __shared__ __attribute__((aligned(16))) _Float16 Gh[2][BT * 32];
    __shared__ __attribute__((aligned(16))) _Float16 Qt[2][2][2][16 * 4 * 8];
    __shared__ __attribute__((aligned(16))) _Float16 Gt[2][2][2][16 * 4 * 8];
    __shared__ __attribute__((aligned(16))) float NLs[2][BT];
    __shared__ __attribute__((aligned(16))) float NDs[2][BT];
    __shared__ float gmax_s[2][4];   // per staging wave: max |dO| of its 16 queries (before the tile is staged)
    __shared__ float dmax_s[2][4];   // per staging wave: max |D| of its 16 queries
    __shared__ int slow_s[2][4];     // per staging wave: a q or dO that the halves cannot hold
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bh, kblk;
    decode_block(blockIdx.x, L / (4 * KT * 16), n_bh, bh, kblk);
    const int k0 = kblk * (4 * KT * 16) + wave * (KT * 16);
    const int lq = lane & 15, g = lane >> 4;
    const int b = bh / heads, head = bh - b * heads, C = heads * 16;
    const float scale_log2e = scale * 1.4426950408889634f;
    const int RS = lay.rs;
    const size_t base = (size_t)b * lay.img + (size_t)head * lay.hs;
    const float* Qb = Q + base;
    const float* dOb = dO + (size_t)b * L * C + head * 16;
    const float* Lb = LSE2 + (size_t)bh * L;
    const float* Db = D + (size_t)bh * L;

    // resident operands of the wave's keys: lane (key, g) holds columns 4g .. 4g+3 of row key
    float4 kf[KT], vf[KT];
    f16x8 kb1[KT], kb2[KT], vb1[KT], vb2[KT];
    f32x4 dk[KT], dv[KT];
    bool bounded = true;
    float vmax = 0.f;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        const size_t row = base + (size_t)(k0 + t * 16 + lq) * RS + 4 * g;
        kf[t] = *reinterpret_cast<const float4*>(K + row);
        kf[t].x *= scale_log2e; kf[t].y *= scale_log2e; kf[t].z *= scale_log2e; kf[t].w *= scale_log2e;
        vf[t] = *reinterpret_cast<const float4*>(V + row);
        dk[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        dv[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float kk[4] = {kf[t].x, kf[t].y, kf[t].z, kf[t].w}, vv[4] = {vf[t].x, vf[t].y, vf[t].z, vf[t].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const _Float16 kh = (_Float16)kk[i], vh = (_Float16)vv[i];
            kb1[t][i] = kh;
            kb1[t][4 + i] = (_Float16)((float)kh * 0.00048828125f);
            kb2[t][i] = (_Float16)(kk[i] - (float)kh);
            kb2[t][4 + i] = (_Float16)0.f;
            vb1[t][i] = vh;
            vb1[t][4 + i] = (_Float16)((float)vh * 0.00048828125f);
            vb2[t][i] = (_Float16)(vv[i] - (float)vh);
            vb2[t][4 + i] = (_Float16)0.f;
            bounded = bounded && (fabsf(kk[i]) < 32768.f) && (fabsf(vv[i]) < 32768.f);
            vmax = fmaxf(vmax, fabsf(vv[i]));
        }
    }
    vmax = wave_max(vmax);
    const bool wave_slow = __any(!bounded) != 0;

    // staging: 64 queries x 16 floats of Q and dO as rows (one float4 each per thread) and as columns (one column vd of the
    // four queries that are neighbours in a lane's slot order, read back from the fp32 rows in LDS); -LSE2 and -D by the
    // first 64 threads.
    const int srow = tid >> 2, sc4 = (tid & 3) * 4, r64 = tid & 63;
    const int vd = lane >> 2, vg = lane & 3, vj = wave & 1, vhalf = wave >> 1;
    const int vq = 32 * vhalf + 16 * vj + 4 * vg;
    float4 qreg, greg;
    float lreg, dreg;
    auto load_tile = [&](int tile) {
        qreg = *reinterpret_cast<const float4*>(Qb + (size_t)(tile * BT + srow) * RS + sc4);
        greg = *reinterpret_cast<const float4*>(dOb + (size_t)(tile * BT + srow) * C + sc4);
        lreg = Lb[tile * BT + r64];
        dreg = Db[tile * BT + r64];
    };
    // what the staging waves tell one another before a tile is staged (read behind a barrier)
    auto announce = [&](int buf) {
        *reinterpret_cast<float4*>(&Qs[buf][srow * BROW + sc4]) = qreg;    // (the fp32 rows: stage() reads its columns from them)
        *reinterpret_cast<float4*>(&dOs[buf][srow * BROW + sc4]) = greg;
        const float gm = wave_max(fmaxf(fmaxf(fabsf(greg.x), fabsf(greg.y)), fmaxf(fabsf(greg.z), fabsf(greg.w))));
        if (lane == 0) gmax_s[buf][wave] = gm;
    };
    auto stage = [&](int buf) {
        const float m = fmaxf(fmaxf(gmax_s[buf][0], gmax_s[buf][1]), fmaxf(gmax_s[buf][2], gmax_s[buf][3]));
        const int ex = (int)((__float_as_uint(m) >> 23) & 0xff);
        const bool usable = ex >= 5 && ex <= 250;
        const float sc = usable ? __uint_as_float((unsigned)(257 - ex) << 23) : 1.0f;   // 2^-e: max |2^-e dO| in [8, 16)
        unsigned h[2], l[2];
        split_scaled_pair_f16(qreg.x, qreg.y, h[0], l[0]);
        split_scaled_pair_f16(qreg.z, qreg.w, h[1], l[1]);
        *reinterpret_cast<u32x4*>(&Qh[buf][srow * 32 + 2 * sc4]) = u32x4{h[0], h[1], l[0], l[1]};
        split_scaled_pair_f16(greg.x * sc, greg.y * sc, h[0], l[0]);
        split_scaled_pair_f16(greg.z * sc, greg.w * sc, h[1], l[1]);
        *reinterpret_cast<u32x4*>(&Gh[buf][srow * 32 + 2 * sc4]) = u32x4{h[0], h[1], l[0], l[1]};
        const int at = (vd * 4 + vg) * 8 + 4 * vj;
        const float* qc = &Qs[buf][vq * BROW + vd];    // one column of four neighbouring queries, from the fp32 rows
        const float* gc = &dOs[buf][vq * BROW + vd];
        const float4 qtreg = make_float4(qc[0], qc[BROW], qc[2 * BROW], qc[3 * BROW]);
        const float4 gtreg = make_float4(gc[0], gc[BROW], gc[2 * BROW], gc[3 * BROW]);
        split_scaled_pair_f16(qtreg.x, qtreg.y, h[0], l[0]);
        split_scaled_pair_f16(qtreg.z, qtreg.w, h[1], l[1]);
        *reinterpret_cast<u32x2*>(&Qt[buf][0][vhalf][at]) = u32x2{h[0], h[1]};
        *reinterpret_cast<u32x2*>(&Qt[buf][1][vhalf][at]) = u32x2{l[0], l[1]};
        split_scaled_pair_f16(gtreg.x * sc, gtreg.y * sc, h[0], l[0]);
        split_scaled_pair_f16(gtreg.z * sc, gtreg.w * sc, h[1], l[1]);
        *reinterpret_cast<u32x2*>(&Gt[buf][0][vhalf][at]) = u32x2{h[0], h[1]};
        *reinterpret_cast<u32x2*>(&Gt[buf][1][vhalf][at]) = u32x2{l[0], l[1]};
        if (tid < 64) {
            NLs[buf][r64] = -lreg;
            NDs[buf][r64] = -dreg;
        }
        const float qm = fmaxf(fmaxf(fabsf(qreg.x), fabsf(qreg.y)), fmaxf(fabsf(qreg.z), fabsf(qreg.w)));
        const bool out = !(qm < 32768.f) || !(usable || m == 0.f);
        const int any_out = __any(out);
        const float dm = wave_max(fabsf(dreg));   // (every wave loaded the tile's 64 values of D)
        if (lane == 0) {
            slow_s[buf][wave] = any_out ? 1 : 0;
            dmax_s[buf][wave] = dm;
        }
    };
    load_tile(0);
    announce(0);
    __syncthreads();
    stage(0);
    __syncthreads();

    const int ntiles = L / BT;
    for (int qt = 0; qt < ntiles; ++qt) {
        const int buf = qt & 1;
        load_tile(min(qt + 1, ntiles - 1));
        // the tile's scale again (as stage() formed it), and whether the halves can take it
        const float m = fmaxf(fmaxf(gmax_s[buf][0], gmax_s[buf][1]), fmaxf(gmax_s[buf][2], gmax_s[buf][3]));
        const int ex = (int)((__float_as_uint(m) >> 23) & 0xff);
        const bool usable = ex >= 5 && ex <= 250;
        const float sc = usable ? __uint_as_float((unsigned)(257 - ex) << 23) : 1.0f;
        const float un = usable ? __uint_as_float((unsigned)(ex - 3) << 23) : 1.0f;
        const int flagged = slow_s[buf][0] | slow_s[buf][1] | slow_s[buf][2] | slow_s[buf][3];
        const float dm = dmax_s[buf][0] * sc;
        const bool fast = !wave_slow && __builtin_amdgcn_readfirstlane(flagged) == 0 &&
                          (256.f * vmax + __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(dm))) < 32768.f);
        if (fast) {
            f32x4 dvt[KT], dvt2[KT], dkt[KT], dkt2[KT];
#pragma unroll
            for (int t = 0; t < KT; ++t) {
                dvt[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dvt2[t] = dvt[t]; dkt[t] = dvt[t]; dkt2[t] = dvt[t];
            }
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                f32x4 s[KT][2], dp[KT][2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int jj = 2 * hf + j;
                    const f16x8 qa = *reinterpret_cast<const f16x8*>(&Qh[buf][(jj * 16 + lq) * 32 + 8 * g]);
                    const f16x8 ga = *reinterpret_cast<const f16x8*>(&Gh[buf][(jj * 16 + lq) * 32 + 8 * g]);
                    const f32x4 nl = *reinterpret_cast<const f32x4*>(&NLs[buf][jj * 16 + 4 * g]);
                    f32x4 nd = *reinterpret_cast<const f32x4*>(&NDs[buf][jj * 16 + 4 * g]);
                    nd[0] *= sc; nd[1] *= sc; nd[2] *= sc; nd[3] *= sc;
#pragma unroll
                    for (int t = 0; t < KT; ++t) {
                        f32x4 a = __builtin_amdgcn_mfma_f32_16x16x32_f16(qa, kb1[t], nl, 0, 0, 0);
                        s[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qa, kb2[t], a, 0, 0, 0);
                        f32x4 c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ga, vb1[t], nd, 0, 0, 0);
                        dp[t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ga, vb2[t], c, 0, 0, 0);
                    }
                }
                const f16x8 gthi = *reinterpret_cast<const f16x8*>(&Gt[buf][0][hf][(lq * 4 + g) * 8]);
                const f16x8 gtlo = *reinterpret_cast<const f16x8*>(&Gt[buf][1][hf][(lq * 4 + g) * 8]);
                const f16x8 qthi = *reinterpret_cast<const f16x8*>(&Qt[buf][0][hf][(lq * 4 + g) * 8]);
                const f16x8 qtlo = *reinterpret_cast<const f16x8*>(&Qt[buf][1][hf][(lq * 4 + g) * 8]);
#pragma unroll
                for (int t = 0; t < KT; ++t) {
                    u32x4 ph, pl, sh, sl;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        float p[4], ds[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            p[r] = __builtin_amdgcn_exp2f(s[t][j][r]);
                            ds[r] = p[r] * dp[t][j][r];
                        }
                        unsigned h0, l0, h1, l1;
                        split_pair_f16(p[0], p[1], h0, l0);
                        split_pair_f16(p[2], p[3], h1, l1);
                        ph[2 * j] = h0; ph[2 * j + 1] = h1;
                        pl[2 * j] = l0; pl[2 * j + 1] = l1;
                        split_pair_f16(ds[0], ds[1], h0, l0);
                        split_pair_f16(ds[2], ds[3], h1, l1);
                        sh[2 * j] = h0; sh[2 * j + 1] = h1;
                        sl[2 * j] = l0; sl[2 * j + 1] = l1;
                    }
                    const f16x8 phi = __builtin_bit_cast(f16x8, ph), plo = __builtin_bit_cast(f16x8, pl);
                    const f16x8 dsh = __builtin_bit_cast(f16x8, sh), dsl = __builtin_bit_cast(f16x8, sl);
                    dvt2[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gtlo, phi, dvt2[t], 0, 0, 0);
                    dvt[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gthi, plo, dvt[t], 0, 0, 0);
                    dvt[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gthi, phi, dvt[t], 0, 0, 0);
                    dkt2[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qtlo, dsh, dkt2[t], 0, 0, 0);
                    dkt[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qthi, dsl, dkt[t], 0, 0, 0);
                    dkt[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qthi, dsh, dkt[t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    dv[t][r] += (dvt[t][r] + dvt2[t][r] * 0.00048828125f) * un;
                    dk[t][r] += (dkt[t][r] + dkt2[t][r] * 0.00048828125f) * un;
                }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {  // 16 queries at a time, fp32 instructions on the unscaled tile
                const float4 qa = *reinterpret_cast<const float4*>(&Qs[buf][(j * 16 + lq) * BROW + 4 * g]);
                const float4 doa = *reinterpret_cast<const float4*>(&dOs[buf][(j * 16 + lq) * BROW + 4 * g]);
                const f32x4 nl = *reinterpret_cast<const f32x4*>(&NLs[buf][j * 16 + 4 * g]);
                const f32x4 nd = *reinterpret_cast<const f32x4*>(&NDs[buf][j * 16 + 4 * g]);
                float qt4[4], dot4[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    qt4[r] = Qs[buf][(j * 16 + 4 * g + r) * BROW + lq];
                    dot4[r] = dOs[buf][(j * 16 + 4 * g + r) * BROW + lq];
                }
#pragma unroll
                for (int t = 0; t < KT; ++t) {
                    f32x4 s = nl, dp = nd;
                    MFMA4(s, qa, kf[t]);
                    MFMA4(dp, doa, vf[t]);
                    f32x4 p, ds;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        p[r] = __builtin_amdgcn_exp2f(s[r]);
                        ds[r] = p[r] * dp[r];
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        dv[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(dot4[r], p[r], dv[t], 0, 0, 0);
                        dk[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(qt4[r], ds[r], dk[t], 0, 0, 0);
                    }
                }
            }
        }
        announce(buf ^ 1);
        __syncthreads();   // the maxima of the next tile are there (and every wave is done with this iteration's reads of them)
        stage(buf ^ 1);
        __syncthreads();
    }

#pragma unroll
    for (int t = 0; t < KT; ++t) {
        const size_t row = base + (size_t)(k0 + t * 16 + lq) * RS + 4 * g;
        *reinterpret_cast<float4*>(dK + row) = make_float4(dk[t][0] * scale, dk[t][1] * scale, dk[t][2] * scale, dk[t][3] * scale);
        *reinterpret_cast<float4*>(dV + row) = make_float4(dv[t][0], dv[t][1], dv[t][2], dv[t][3]);
    }
}

static hipError_t launch_bwd(const float* q, const float* k, const float* v, const float* o, const float* dout,
                             const float* lse2, float* dq, float* dk, float* dv, float* delta, int B, int heads, int L,
                             const QkvLayout lay, hipStream_t stream) {
    if (L % 128 != 0) return hipErrorInvalidValue;
    const float scale = 0.25f;  // head_dim^-1/2
    const dim3 grid((L / 128) * B * heads);
    if (attention_hd16_fp32_products())
        hipLaunchKernelGGL(k_flash_hd16_bwd_dq<2>, grid, dim3(256), 0, stream, q, k, v, o, dout, lse2, dq, delta, L, heads,
                           B * heads, scale, lay);
    else
        hipLaunchKernelGGL(k_flash_hd16_bwd_dq_h<2>, grid, dim3(256), 0, stream, q, k, v, o, dout, lse2, dq, delta, L, heads,
                           B * heads, scale, lay);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (attention_hd16_fp32_products())
        hipLaunchKernelGGL(k_flash_hd16_bwd_dkv<2>, grid, dim3(256), 0, stream, q, k, v, dout, lse2, delta, dk, dv, L, heads,
                           B * heads, scale, lay);
    else
        hipLaunchKernelGGL(k_flash_hd16_bwd_dkv_h<2>, grid, dim3(256), 0, stream, q, k, v, dout, lse2, delta, dk, dv, L, heads,
                           B * heads, scale, lay);
    return hipGetLastError();
}

hipError_t launch_attention_hd16_bwd(const float* q, const float* k, const float* v, const float* o, const float* dout,
                                     const float* lse2, float* dq, float* dk, float* dv, float* delta, int B, int heads,
                                     int L, hipStream_t stream) {
    return launch_bwd(q, k, v, o, dout, lse2, dq, dk, dv, delta, B, heads, L,
                      QkvLayout{16, (long long)heads * L * 16, L * 16}, stream);
}

// q, k, v read from, and dq, dk, dv written to, the layout of a fused QKV projection: qkv, dqkv [B][L][3][heads * 16]
hipError_t launch_attention_hd16_bwd_tokens(const float* qkv, const float* o, const float* dout, const float* lse2,
                                            float* dqkv, float* delta, int B, int heads, int L, hipStream_t stream) {
    const int C = heads * 16;
    return launch_bwd(qkv, qkv + C, qkv + 2 * C, o, dout, lse2, dqkv, dqkv + C, dqkv + 2 * C, delta, B, heads, L,
                      QkvLayout{3 * C, (long long)L * 3 * C, 16}, stream);
}

}  // namespace danra
