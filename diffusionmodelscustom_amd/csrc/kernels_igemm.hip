// Implicit-GEMM kernel on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32, exact fp32 FMA
// chains).  One kernel serves every dense contraction of the network: the 3x3 / 8x8 / 1x1 NHWC
// convolutions (k = tap-major, channel-minor), the transposed 2x2 stride-2 convolutions (four 1x1
// GEMM slabs + pixel-shuffle store), and the attention in/out projections (with the LayerNorm or
// InstanceNorm of the operand applied while it is staged into LDS).
//
// Tile: BM x 64 outputs per 256-thread workgroup, K walked in chunks of 32 through two LDS buffers;
// global loads of chunk c+1 are in flight while chunk c is multiplied.  LDS rows are padded to 36
// floats: 16-byte aligned for ds_write_b128 and conflict-free for the ds_read_b128 fragment reads
// (lane (i, h) reads k = 8u+4h..8u+4h+3 of row i, so MFMA step r consumes k in {8u+r, 8u+4+r} from
// both operands).
#include "danra_kernels.h"

namespace danra {

typedef float f32x16 __attribute__((ext_vector_type(16)));

static constexpr int BN = 64;
static constexpr int BK = 32;
static constexpr int LDK = 36;

__device__ __forceinline__ void epilogue_store(const IGemmArgs& p, int m, int n, float v) {
    if (p.bias) v += p.bias[n];
    if (p.res) v += p.res[(size_t)m * p.ldr + n];
    if (p.relu) v = fmaxf(v, 0.f);
    if (p.tbias) v += p.tbias[(size_t)(m / p.rows_per_img) * p.tb_stride + n];
    if (p.st_mode == ST_ROW) {
        p.out[(size_t)m * p.ldc + n] = v;
    } else if (p.st_mode == ST_QKV) {
        const int which = n / p.qkv_C;
        const int cc = n - which * p.qkv_C;
        const int head = cc / p.qkv_hd;
        const int d = cc - head * p.qkv_hd;
        const int b = m / p.qkv_L;
        const int l = m - b * p.qkv_L;
        p.out[(size_t)which * p.qkv_plane + ((size_t)(b * p.qkv_heads + head) * p.qkv_L + l) * p.qkv_hd + d] = v;
    } else {  // ST_CONVT: n = (dy, dx, co), m = (b, y, x) of the input grid
        const int q = n / p.ct_C;
        const int co = n - q * p.ct_C;
        const int hw = p.ct_H * p.ct_W;
        const int b = m / hw;
        const int r = m - b * hw;
        const int y = r / p.ct_W;
        const int x = r - y * p.ct_W;
        const size_t oy = 2 * y + (q >> 1), ox = 2 * x + (q & 1);
        p.out[(((size_t)b * 2 * p.ct_H + oy) * 2 * p.ct_W + ox) * p.ct_C + co] = v;
    }
}

template <int BM>
__global__ __launch_bounds__(256) void k_igemm(const IGemmArgs p) {
    constexpr int AR = BM / 32;  // float4 A loads per thread and chunk
    constexpr int NT = (BM == 128) ? 2 : 1;
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LDK + 2 * BM];
    float* const As0 = smem;
    float* const Bs0 = smem + 2 * BM * LDK;
    float* const rowMean = smem + 2 * (BM + BN) * LDK;
    float* const rowRstd = rowMean + BM;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int row0 = (BM == 128) ? wave * 32 : (wave >> 1) * 32;
    const int col0 = (BM == 128) ? 0 : (wave & 1) * 32;

    const int total_chunks = p.K / BK;
    int c_begin = 0, c_end = total_chunks;
    if (p.ksplit > 1) {
        c_begin = blockIdx.z * p.chunks_per_split;
        c_end = min(total_chunks, c_begin + p.chunks_per_split);
    }

    // ---- per-thread staging geometry
    const int lrow = tid >> 3;  // + 32*j
    const int c4 = (tid & 7) * 4;
    int rb[AR], riy[AR], rix[AR];
    bool rvalid[AR];
    float rmean[AR], rrstd[AR];
    const int howo = p.Ho * p.Wo;
#pragma unroll
    for (int j = 0; j < AR; ++j) {
        const int m = m0 + lrow + 32 * j;
        rvalid[j] = m < p.M;
        rmean[j] = 0.f;
        rrstd[j] = 1.f;
        if (p.a_mode != A_GEMM) {
            const int mm = rvalid[j] ? m : 0;
            const int b = mm / howo;
            const int r = mm - b * howo;
            const int oy = r / p.Wo;
            const int ox = r - oy * p.Wo;
            rb[j] = b;
            riy[j] = oy * p.stride - p.pad;
            rix[j] = ox * p.stride - p.pad;
        } else {
            rb[j] = 0;
            riy[j] = 0;
            rix[j] = 0;
        }
    }

    // ---- LayerNorm statistics of this block's rows (two-pass, fp32), K = channel count
    if (p.ln) {
        for (int r = wave * (BM / 4); r < (wave + 1) * (BM / 4); ++r) {
            const int m = m0 + r;
            float mean = 0.f, rstd = 1.f;
            if (m < p.M) {
                const float* src = p.A + (size_t)m * p.lda;
                float s = 0.f;
                for (int k = lane; k < p.K; k += 64) s += src[k];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
                mean = s / (float)p.K;
                float q = 0.f;
                for (int k = lane; k < p.K; k += 64) {
                    const float d = src[k] - mean;
                    q += d * d;
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
                rstd = 1.0f / sqrtf(q / (float)p.K + 1e-5f);
            }
            if (lane == 0) {
                rowMean[r] = mean;
                rowRstd[r] = rstd;
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < AR; ++j) {
            rmean[j] = rowMean[lrow + 32 * j];
            rrstd[j] = rowRstd[lrow + 32 * j];
        }
    }

    float4 areg[AR], breg[2];

    auto load_chunk = [&](int kc) {
        const int k0 = kc * BK;
        if (p.a_mode == A_GEMM) {
#pragma unroll
            for (int j = 0; j < AR; ++j) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (rvalid[j]) {
                    v = *reinterpret_cast<const float4*>(p.A + (size_t)(m0 + lrow + 32 * j) * p.lda + k0 + c4);
                    if (p.ln) {
                        v.x = (v.x - rmean[j]) * rrstd[j];
                        v.y = (v.y - rmean[j]) * rrstd[j];
                        v.z = (v.z - rmean[j]) * rrstd[j];
                        v.w = (v.w - rmean[j]) * rrstd[j];
                    }
                }
                areg[j] = v;
            }
        } else if (p.a_mode == A_CONV) {
            const int tap = k0 / p.Cin;
            const int cc = k0 - tap * p.Cin + c4;
            const int kh = tap / p.KW;
            const int kw = tap - kh * p.KW;
#pragma unroll
            for (int j = 0; j < AR; ++j) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                const int iy = riy[j] + kh, ix = rix[j] + kw;
                if (rvalid[j] && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi) {
                    v = *reinterpret_cast<const float4*>(p.A + ((size_t)(rb[j] * p.Hi + iy) * p.Wi + ix) * p.Cin + cc);
                    if (p.nmean) {
                        const float4 mu = *reinterpret_cast<const float4*>(p.nmean + (size_t)rb[j] * p.Cin + cc);
                        const float4 rs = *reinterpret_cast<const float4*>(p.nrstd + (size_t)rb[j] * p.Cin + cc);
                        v.x = (v.x - mu.x) * rs.x;
                        v.y = (v.y - mu.y) * rs.y;
                        v.z = (v.z - mu.z) * rs.z;
                        v.w = (v.w - mu.w) * rs.w;
                    }
                }
                areg[j] = v;
            }
        } else {  // A_CONV1: k = (plane, kh, kw); a 32-chunk = one plane, four kernel rows, eight columns
            const int plane = k0 / (p.KH * p.KW);
            const int kh = (k0 - plane * p.KH * p.KW) / p.KW + (c4 >> 3);
            const int kw = c4 & 7;
            const float* src = p.planes[plane];
#pragma unroll
            for (int j = 0; j < AR; ++j) {
                float e[4] = {0.f, 0.f, 0.f, 0.f};
                const int iy = riy[j] + kh;
                if (rvalid[j] && iy >= 0 && iy < p.Hi) {
                    const float* rowp = src + ((size_t)rb[j] * p.Hi + iy) * p.Wi;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int ix = rix[j] + kw + q;
                        if (ix >= 0 && ix < p.Wi) e[q] = rowp[ix];
                    }
                }
                areg[j] = make_float4(e[0], e[1], e[2], e[3]);
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
            breg[j] = *reinterpret_cast<const float4*>(p.W + (size_t)(n0 + lrow + 32 * j) * p.K + k0 + c4);
    };

    auto store_chunk = [&](int buf) {
        float* As = As0 + buf * BM * LDK;
        float* Bs = Bs0 + buf * BN * LDK;
#pragma unroll
        for (int j = 0; j < AR; ++j) *reinterpret_cast<float4*>(As + (lrow + 32 * j) * LDK + c4) = areg[j];
#pragma unroll
        for (int j = 0; j < 2; ++j) *reinterpret_cast<float4*>(Bs + (lrow + 32 * j) * LDK + c4) = breg[j];
    };

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int fr = lane & 31;
    const int fh = (lane >> 5) * 4;

    if (c_begin < c_end) {
        load_chunk(c_begin);
        store_chunk(0);
    }
    __syncthreads();
    for (int kc = c_begin; kc < c_end; ++kc) {
        const int buf = (kc - c_begin) & 1;
        if (kc + 1 < c_end) load_chunk(kc + 1);
        const float* As = As0 + buf * BM * LDK + (row0 + fr) * LDK + fh;
        const float* Bs = Bs0 + buf * BN * LDK + (col0 + fr) * LDK + fh;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const float4 a = *reinterpret_cast<const float4*>(As + kb * 8);
            float4 b[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) b[t] = *reinterpret_cast<const float4*>(Bs + t * 32 * LDK + kb * 8);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[t].x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[t].y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[t].z, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[t].w, acc[t], 0, 0, 0);
            }
        }
        if (kc + 1 < c_end) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: C/D layout of the 32x32 tile: col = lane & 31, row = (r & 3) + 8*(r >> 2) + 4*(lane >> 5)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = n0 + col0 + t * 32 + fr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + row0 + (r & 3) + 8 * (r >> 2) + fh;
            if (m < p.M) {
                if (p.ksplit > 1)
                    p.partial[((size_t)blockIdx.z * p.M + m) * p.N + n] = acc[t][r];
                else
                    epilogue_store(p, m, n, acc[t][r]);
            }
        }
    }
}

// Sums the split-K slabs in slab order (bitwise reproducible) and applies the epilogue.
__global__ __launch_bounds__(256) void k_splitk_epilogue(const IGemmArgs p) {
    const size_t total = (size_t)p.M * p.N;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        float v = 0.f;
        for (int z = 0; z < p.ksplit; ++z) v += p.partial[(size_t)z * total + idx];
        const int m = (int)(idx / p.N);
        const int n = (int)(idx - (size_t)m * p.N);
        epilogue_store(p, m, n, v);
    }
}

hipError_t launch_igemm(const IGemmArgs& a, hipStream_t stream) {
    const int ks = a.ksplit > 1 ? a.ksplit : 1;
    // Small problems use the 64-row tile so that more workgroups exist.
    const bool small = ((a.M + 127) / 128) * (a.N / BN) * ks < 256;
    if (small) {
        dim3 grid((a.M + 63) / 64, a.N / BN, ks);
        hipLaunchKernelGGL(k_igemm<64>, grid, dim3(256), 0, stream, a);
    } else {
        dim3 grid((a.M + 127) / 128, a.N / BN, ks);
        hipLaunchKernelGGL(k_igemm<128>, grid, dim3(256), 0, stream, a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (ks > 1) {
        const size_t total = (size_t)a.M * a.N;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(k_splitk_epilogue, dim3(blocks), dim3(256), 0, stream, a);
        e = hipGetLastError();
    }
    return e;
}

}  // namespace danra
