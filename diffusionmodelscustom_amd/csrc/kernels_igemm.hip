// Implicit-GEMM kernel on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32, exact fp32 FMA
// chains).  One kernel serves every dense contraction of the network: the 3x3 / 8x8 / 1x1 NHWC
// convolutions (k = tap-major, channel-minor), the transposed 2x2 stride-2 convolutions (four 1x1
// GEMM slabs + pixel-shuffle store), and the attention in/out projections (with the LayerNorm or
// InstanceNorm of the operand applied while it is staged into LDS).
//
// Tile: BM x 64 outputs per workgroup, K walked in chunks of BK through two LDS buffers; the global
// loads of chunk c+1 are in flight while chunk c is multiplied.  A workgroup has 4*KW waves: four
// own the 32x32 (BM=64) or 32x64 (BM=128) output sub-tiles, and each group of four takes every KW-th
// 8-deep k-block of a chunk, so small problems (few output tiles) still put 2-4 waves on every SIMD
// without an extra kernel; the KW partial accumulators are summed through LDS in a fixed order.
// Every global load is unconditional (out-of-range rows / padding taps read a clamped address and
// are zeroed afterwards): a load under a condition makes the compiler wait for it on the spot,
// which serialises the prefetch.  LDS rows are padded by 4 floats: 16-byte aligned for
// ds_write_b128 and conflict-free for the ds_read_b128 fragment reads (lane (i, h) reads
// k = 8u+4h..8u+4h+3 of row i, so MFMA step r consumes k in {8u+r, 8u+4+r} from both operands).
#include <type_traits>

#include "danra_kernels.h"
#include "split_f16.h"

namespace danra {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// The four values of a staged float4 as fp16 halves (see split_pair_f16).
__device__ __forceinline__ void split4(const float4 v, f16x4& hi, f16x4& lo) {
    unsigned h[2], l[2];
    split_pair_f16(v.x, v.y, h[0], l[0]);
    split_pair_f16(v.z, v.w, h[1], l[1]);
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    hi = __builtin_bit_cast(f16x4, (u32x2){h[0], h[1]});
    lo = __builtin_bit_cast(f16x4, (u32x2){l[0], l[1]});
}

static constexpr int BN = 64;

// process-wide switch: every product of this file with fp32 matrix instructions (danra_set_fp32_products)
static int g_igemm_fp32_products = 0;
void set_igemm_fp32_products(int on) { g_igemm_fp32_products = on ? 1 : 0; }
bool igemm_fp32_products_forced() { return g_igemm_fp32_products != 0; }

// x / d for x >= 0, d > 0: a shift when the host found d to be a power of two (sh = its log2, else -1).  Every
// feature map of the networks this library runs has power-of-two sides; an integer division by a run-time value
// costs some forty instructions, and the set-up of a workgroup does a dozen of them before its first load.
__device__ __forceinline__ int div_by(int x, int d, int sh) { return sh >= 0 ? (x >> sh) : x / d; }

// Output address = row part (depends on m) + column part (depends on n), for every store mode.
__device__ __forceinline__ unsigned long long row_offset(const IGemmArgs& p, int m, int& img) {
    img = div_by(m, p.rows_per_img, p.sh_rpi);
    if (p.st_mode == ST_ROW) return (unsigned long long)m * p.ldc;
    if (p.st_mode == ST_QKV) {
        const int b = div_by(m, p.qkv_L, p.sh_qkv_l);
        const int l = m - b * p.qkv_L;
        return ((unsigned long long)b * p.qkv_L * p.qkv_C) + (unsigned long long)l * p.qkv_hd;
    }
    const int hw = p.ct_H * p.ct_W;  // ST_CONVT: m = (b, y, x) of the input grid
    const int b = div_by(m, hw, p.sh_ct_hw);
    const int r = m - b * hw;
    const int y = div_by(r, p.ct_W, p.sh_ct_w);
    const int x = r - y * p.ct_W;
    return (((unsigned long long)b * 2 * p.ct_H + 2 * y) * 2 * p.ct_W + 2 * x) * p.ct_C;
}

// Row m of the GEMM -> row of the output tensor.  Convolutions on tiny feature maps walk their rows pixel-major
// (m = pixel * batch + image), so that the rows of a tile share one output position and with it the set of
// kernel taps that fall inside the image; everything else is image-major (identity).
__device__ __forceinline__ int out_row(const IGemmArgs& p, int m) {
    if (!p.pix_major) return m;
    const int r = div_by(m, p.batch, p.sh_batch);
    const int b = m - r * p.batch;
    return b * (p.Ho * p.Wo) + r;
}

__device__ __forceinline__ unsigned long long col_offset(const IGemmArgs& p, int n) {
    if (p.st_mode == ST_ROW) return n;
    if (p.st_mode == ST_QKV) {
        const int which = n / p.qkv_C;
        const int cc = n - which * p.qkv_C;
        const int head = cc / p.qkv_hd;
        const int d = cc - head * p.qkv_hd;
        return (unsigned long long)which * p.qkv_plane + (unsigned long long)head * p.qkv_L * p.qkv_hd + d;
    }
    const int q = n / p.ct_C;  // n = (dy, dx, co)
    const int co = n - q * p.ct_C;
    return ((unsigned long long)(q >> 1) * 2 * p.ct_W + (q & 1)) * p.ct_C + co;
}

// Exact (erf) GELU, as torch.nn.GELU() evaluates it.
__device__ __forceinline__ float gelu(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }
__device__ __forceinline__ float activate(int act, float v) {
    if (act == ACT_RELU) return fmaxf(v, 0.f);
    if (act == ACT_GELU) return gelu(v);
    return v;
}
__device__ __forceinline__ float4 activate(int act, float4 v) {
    if (act == ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    else if (act == ACT_GELU) { v.x = gelu(v.x); v.y = gelu(v.y); v.z = gelu(v.z); v.w = gelu(v.w); }
    return v;
}

__device__ __forceinline__ float finish(const IGemmArgs& p, float v, float bias, int m, int n, int img) {
    v += bias;
    if (p.out2 && n >= p.n_split) return v;  // second output: bias only
    if (p.res) v += p.res[(size_t)m * p.ldr + n];
    v = activate(p.act, v);
    if (p.tbias) v += p.tbias[(size_t)img * p.tb_stride + n];
    return v;
}

__device__ __forceinline__ float component(const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }

// Timing ablations (tune_igemm builds this file with -DDANRA_IGEMM_DEBUG=1; results are then wrong when
// p.debug != 0).  In the library build the flags are compiled out of the loop.
#ifndef DANRA_IGEMM_DEBUG
#define DANRA_IGEMM_DEBUG 0
#endif

constexpr int A_CONV_IN = 3;  // kernel-internal: A_CONV with InstanceNorm applied on load
constexpr int A_GEMM_LN = 4;  // kernel-internal: A_GEMM with LayerNorm applied on load

// HALF: the products run on the fp16 matrix cores (v_mfma_f32_32x32x16_f16, fp32 accumulation) with both operands split
// in two fp16 halves by the staging threads: W A ~ Wh Ah + Wh Al + Wl Ah (the dropped Wl Al term is 2^-22 of a
// product): three instructions per 16-deep k-step at 16x the fp32 instruction's rate.  An LDS row then holds
// [BK halves hi | BK halves lo] in the bytes of the BK floats it holds otherwise.
// SPEC (with HALF, KW = 4 and BM = 64 only): twelve waves in two roles.  Waves 0-3 only multiply: each owns the WHOLE
// 64x64 tile (four accumulator tiles) for one of the four 16-deep k-steps of every chunk, so a fragment read from LDS
// feeds two matrix instructions per term instead of one (the LDS read traffic of the 4 sub-tiles x 4 k-parts
// arrangement is twice as high, and with the split operands it is the largest item of an iteration).  Waves 4-11 only
// stage (global loads, conversion to fp16 halves, LDS writes).  Waves that all do both go through those phases
// together - matrix cores idle while everybody converts, and the other way round; with the roles apart the phases
// of a SIMD's waves overlap.  The four partial tiles are summed in the epilogue (fixed order), by all twelve waves.
template <int BM, int KW, int AMODE, bool HALF = false, bool SPEC = false>
__global__ __launch_bounds__(SPEC ? 768 : 256 * KW) void k_igemm(const IGemmArgs p) {
    static_assert(!SPEC || (HALF && KW == 4 && BM == 64), "wave roles exist for the 64-row split-operand kernel only");
    constexpr int KWC = KW;                    // k-parts among the multiplying waves
    constexpr int TS = SPEC ? 512 : 256 * KW;  // staging threads
    constexpr bool CONV = (AMODE == A_CONV || AMODE == A_CONV_IN);
    constexpr bool NORM = (AMODE == A_CONV_IN);
    constexpr bool GEMM = (AMODE == A_GEMM || AMODE == A_GEMM_LN);
    constexpr bool LNORM = (AMODE == A_GEMM_LN);
    const int dbg = DANRA_IGEMM_DEBUG ? p.debug : 0;
    constexpr int T = SPEC ? 768 : 256 * KW;
    constexpr int BK = (KW == 4 || HALF) ? 64 : 32;
    constexpr int LDK = BK + 4;
    constexpr int TPR = BK / 4;        // threads per staged row
    constexpr int SWEEP = TS / TPR;    // rows staged per sweep
    constexpr int AR = BM / SWEEP;     // float4 A loads per thread and chunk
    constexpr int BR = BN / SWEEP;     // float4 B loads per thread and chunk
    constexpr int NT = (BM == 128) ? 2 : 1;
    constexpr int NTR = SPEC ? 2 : 1;   // 32x32 accumulator tiles of a multiplying wave: rows x columns
    constexpr int NTC = SPEC ? 2 : NT;
    constexpr int NACC = NTR * NTC;
    constexpr int KB = BK / 8;         // 8-deep k-blocks per chunk
    static_assert(AR >= 1 && BR >= 1, "tile/threads mismatch");
    constexpr int STAGE_FLOATS = 2 * (BM + BN) * LDK;
    constexpr int RED_FLOATS = SPEC ? 4 * BM * (BN + 4) : (KWC - 1) * 4 * NT * 16 * 64;
    __shared__ __attribute__((aligned(16))) float smem[STAGE_FLOATS > RED_FLOATS ? STAGE_FLOATS : RED_FLOATS];
    __shared__ unsigned long long rowOff[BM];
    __shared__ int rowImg[BM];
    __shared__ int rowOut[BM];   // row of the output tensor (differs from m0 + row in pixel-major order)
    __shared__ int rangeFlag[3];  // HALF: a value too large for the split / a normal-sized value / a non-zero value was staged
    __shared__ int tapBox[BM / 64][4];  // convolutions, per 64 rows: kh_lo, kh_hi, kw_lo, kw_hi = taps that touch the image for some row
    float* const As0 = smem;
    float* const Bs0 = smem + 2 * BM * LDK;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    if (HALF && tid < 3) rangeFlag[tid] = 0;  // (ordered before its first use by the barriers of the pipeline)
    // Timing build only, debug bit 64: workgroup 0 leaves clock readings of the kernel's phases behind the per-iteration ones
    auto phase_stamp = [&](int k) {
#if DANRA_IGEMM_DEBUG
        if ((dbg & 64) && blockIdx.x == 0) {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (lane == 0) reinterpret_cast<unsigned long long*>(p.partial)[16 * 64 * 4 + wave * 8 + k] = t;
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
    };
    phase_stamp(0);
    const int tw = wave & 3;    // which output sub-tile
    const int kpart = SPEC ? (wave & 3) : (wave >> 2);  // which k-blocks of every chunk
    // SPEC: waves 4-11 stage, waves 0-3 multiply (a wave-uniform scalar; without SPEC every wave does both)
    const bool stager = !SPEC || __builtin_amdgcn_readfirstlane(wave) >= 4;
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int row0 = SPEC ? 0 : (BM == 128) ? tw * 32 : (tw >> 1) * 32;
    const int col0 = (SPEC || BM == 128) ? 0 : (tw & 1) * 32;

    // ---- output row table (a few integer divisions per row, done once per block)
    if (tid < BM) {
        const int m = min(m0 + tid, p.M - 1);
        const int mo = out_row(p, m);
        int img;
        rowOff[tid] = row_offset(p, mo, img);
        rowImg[tid] = img;
        rowOut[tid] = mo;
        if (CONV && p.pix_major) {  // taps of this row that read inside the image: iy = oy*stride - pad + kh in [0, Hi)
            const int r = mo - img * (p.Ho * p.Wo);
            const int oy = div_by(r, p.Wo, p.sh_wo), ox = r - oy * p.Wo;
            const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
            int lo_h = max(0, -iy0), hi_h = min(p.KH - 1, p.Hi - 1 - iy0);
            int lo_w = max(0, -ix0), hi_w = min(p.KW - 1, p.Wi - 1 - ix0);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {  // the 64 rows of this wave
                lo_h = min(lo_h, __shfl_xor(lo_h, o));
                hi_h = max(hi_h, __shfl_xor(hi_h, o));
                lo_w = min(lo_w, __shfl_xor(lo_w, o));
                hi_w = max(hi_w, __shfl_xor(hi_w, o));
            }
            if (lane == 0) {
                tapBox[wave][0] = lo_h; tapBox[wave][1] = hi_h; tapBox[wave][2] = lo_w; tapBox[wave][3] = hi_w;
            }
        }
    }
    if (CONV && p.pix_major) __syncthreads();
    // A pixel-major convolution only walks the taps of that box (on a 2x2 map that is 4 of the 9 taps of a 3x3 kernel);
    // chunks are numbered tap-major inside the box, and split-K slices are cut from that numbering.
    // (read through readfirstlane: values loaded from LDS are per-lane to the compiler, and the whole chunk walk
    // would otherwise be treated as divergent)
    int box[4] = {0, p.KH - 1, 0, p.KW - 1};
    if (CONV && p.pix_major) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int v = tapBox[0][i];
            if (BM == 128) v = (i & 1) ? max(v, tapBox[BM / 64 - 1][i]) : min(v, tapBox[BM / 64 - 1][i]);
            box[i] = __builtin_amdgcn_readfirstlane(v);
        }
    }
    const int kh_lo = box[0], kw_lo = box[2];
    const int nkh = max(box[1] - kh_lo + 1, 1), nkw = max(box[3] - kw_lo + 1, 1);
    const int cpt = CONV ? p.Cin / BK : 1;  // chunks per tap
    const int total_chunks = CONV ? nkh * nkw * cpt : p.K / BK;
    int c_begin = 0, c_end = total_chunks;
    if (p.ksplit > 1) {
        const int cps = CONV ? (total_chunks + p.ksplit - 1) / p.ksplit
                             : p.chunks_per_split * 32 / BK;  // (GEMM, stem) the host counts 32-deep chunks
        c_begin = min((int)blockIdx.z * cps, total_chunks);
        c_end = min(total_chunks, c_begin + cps);
    }

    // ---- per-thread staging geometry
    const int stid = SPEC ? ((tid - 256) & 511) : tid;  // index among the staging threads
    const int lrow = stid / TPR;  // + SWEEP*j
    const int c4 = (stid % TPR) * 4;
    // Row j of this thread: 32-bit element offsets (every tensor of the network is < 2^31 elements).
    //   conv : abase = offset of tap (0,0) of the row's window (may point before the row's image for
    //          padded taps: only used when the tap is in bounds)
    //   gemm : abase = m*lda + c4
    int rb[AR], riy[AR], rix[AR], rm[AR], abase[AR], nbase[AR];
    bool rvalid[AR];
    float rmean[AR], rrstd[AR];
    const int howo = p.Ho * p.Wo;
#pragma unroll
    for (int j = 0; j < AR; ++j) {
        const int m = m0 + lrow + SWEEP * j;
        rvalid[j] = m < p.M;
        rm[j] = min(m, p.M - 1);
        rmean[j] = 0.f;
        rrstd[j] = 1.f;
        rb[j] = 0;
        riy[j] = 0;
        rix[j] = 0;
        abase[j] = rm[j] * p.lda + c4;
        nbase[j] = 0;
        if (!GEMM) {
            const int mo = out_row(p, rm[j]);
            const int b = div_by(mo, howo, p.sh_howo);
            const int r = mo - b * howo;
            const int oy = div_by(r, p.Wo, p.sh_wo);
            const int ox = r - oy * p.Wo;
            rb[j] = b;
            riy[j] = oy * p.stride - p.pad;
            rix[j] = ox * p.stride - p.pad;
            abase[j] = ((b * p.Hi + riy[j]) * p.Wi + rix[j]) * p.Cin + c4;
            nbase[j] = b * p.Cin + c4;
        }
    }
    int wbase[BR];
    const int ldw = p.ldw ? p.ldw : p.K;  // weight row stride (a column slice of a wider matrix has ldw > K)
#pragma unroll
    for (int j = 0; j < BR; ++j) wbase[j] = (n0 + lrow + SWEEP * j) * ldw + c4;

    // ---- LayerNorm statistics of this thread's rows (two-pass, fp32) over the whole K = channel
    //      count: TPR lanes share a row, same thread->row mapping as the staging loads
    if (LNORM && stager) {
#pragma unroll
        for (int j = 0; j < AR; ++j) {
            const float* src = p.A + abase[j];
            float s = 0.f;
            for (int k = 0; k < p.K; k += BK) {
                const float4 v = *reinterpret_cast<const float4*>(src + k);
                s += (v.x + v.y) + (v.z + v.w);
            }
#pragma unroll
            for (int o = 1; o < TPR; o <<= 1) s += __shfl_xor(s, o);
            const float mean = s / (float)p.K;
            float q = 0.f;
            for (int k = 0; k < p.K; k += BK) {
                const float4 v = *reinterpret_cast<const float4*>(src + k);
                const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
                q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            }
#pragma unroll
            for (int o = 1; o < TPR; o <<= 1) q += __shfl_xor(q, o);
            rmean[j] = mean;
            rrstd[j] = 1.0f / sqrtf(q / (float)p.K + 1e-5f);
        }
    }

    // ---- what the epilogue adds to this thread's results.  With wave roles every thread has registers to spare and
    // the kernels are short (a dozen chunks on the small maps): bias, residual and embedding bias are fetched now,
    // which takes a trip to memory off the end of the kernel.
    constexpr int ESWEEP = SPEC ? T / 16 : 16;  // output rows per sweep of the threads that take part in the epilogue
    constexpr int NEJ = (BM + ESWEEP - 1) / ESWEEP;
    const int oc4 = (tid & 15) * 4;
    const int n = n0 + oc4;
    // columns from n_split on belong to the second output (same rows, its own column 0, bias only); BN divides
    // n_split, so a workgroup's columns lie on one side
    const bool second = p.out2 && n0 >= p.n_split;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (SPEC && p.bias && p.ksplit <= 1) bv = *reinterpret_cast<const float4*>(p.bias + n);
    float4 pre_res[NEJ], pre_tb[NEJ];
    if (SPEC && p.ksplit <= 1 && !second) {
#pragma unroll
        for (int j = 0; j < NEJ; ++j) {
            const int row = (tid >> 4) + ESWEEP * j;
            pre_res[j] = pre_tb[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row >= BM || m0 + row >= p.M) continue;
            const int mo = out_row(p, m0 + row);
            if (p.res) pre_res[j] = *reinterpret_cast<const float4*>(p.res + (size_t)mo * p.ldr + n);
            if (p.tbias)
                pre_tb[j] = *reinterpret_cast<const float4*>(p.tbias + (size_t)div_by(mo, p.rows_per_img, p.sh_rpi) * p.tb_stride + n);
        }
    }

    f32x16 acc[NACC];  // tile (tr, tc) at index tr * NTC + tc
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int fr = lane & 31;
    const int fh = (lane >> 5) * 4;

    // Software pipeline over the chunks, one barrier per chunk.  Iteration `it` reads its operand fragments of chunk
    // `it` from LDS buffer it&1 and issues the first k-block's matrix instructions; then it writes chunk it+1 (whose
    // global loads were issued a whole iteration ago, so the wait for them is free) to buffer (it+1)&1 and issues the
    // loads of chunk it+2; then the remaining matrix instructions and the barrier.  The staging work thus sits in the
    // shadow of the wave's own matrix instructions, and between the last of them and the barrier there is nothing left
    // to do.  (With the load wait and the LDS writes at the end of the iteration the 8x8 stride-2 convolution took
    // 107 us in tune_igemm, with them at its start 99.5 us, in the middle 95 us; fragment reads ahead of the staging
    // work: 109 us.)  One load site and one store site, so the staging registers stay in VGPRs; the loads past the
    // last chunk re-read it (clamped): every load stays unconditional.
    const int n_chunks = c_end - c_begin;
    // conv: (kh, kw, cc) of the chunk being loaded, advanced incrementally (wave-uniform scalars)
    //       and delta = the tap's element offset inside the NHWC window, (kh*Wi + kw)*Cin + cc
    //       and wk = the same position in the weight row, (kh*KW + kw)*Cin + cc
    int t_kh = 0, t_kw = 0, t_cc = 0, delta = 0, wk = 0;
    int loaded = 0;       // chunks whose loads have been issued
    auto rewind = [&]() {  // to the first chunk of this workgroup's K range
        loaded = 0;
        if (CONV) {
            const int q = min(c_begin, total_chunks - 1);
            const int tap = q / cpt;
            t_cc = (q - tap * cpt) * BK;
            t_kh = kh_lo + tap / nkw;
            t_kw = kw_lo + tap % nkw;
            delta = (t_kh * p.Wi + t_kw) * p.Cin + t_cc;
            wk = (t_kh * p.KW + t_kw) * p.Cin + t_cc;
        }
    };
    const int row_jump = (p.Wi - nkw) * p.Cin;   // from the box's last tap of a row to its first tap of the next
    const int wrow_jump = (p.KW - nkw) * p.Cin;
    const int kw_end = kw_lo + nkw;

    typedef float v4f __attribute__((ext_vector_type(4)));  // (HIP's float4 is a struct: arrays of it are not always promoted to registers)
    // two sets of staging registers: the split-operand pipeline keeps the loads of two chunks in flight (set = chunk & 1)
    v4f araw[2][AR], amu[2][AR], ars[2][AR], breg[2][BR];
    const float* Wsrc = HALF ? p.Wh : p.W;
    unsigned okmask[2] = {0, 0};  // stem only: 4 bits per staged row, which of the four elements are real data
    float amax = 0.f;     // HALF: largest |value| this thread has staged (range check of the fp16 split)

    // -- load phase: issue global loads only; nothing here may consume a loaded value
    auto load_chunk = [&](auto set_tag) {
        constexpr int SET = decltype(set_tag)::value;
        const int k0 = (dbg & 1) ? c_begin * BK : min(c_begin + loaded, c_end - 1) * BK;  // GEMM, stem
        const int wk_load = wk;
        okmask[SET] = 0;
        if (GEMM) {
            // rows past M re-read row M-1: they only feed output rows that are never stored
#pragma unroll
            for (int j = 0; j < AR; ++j) araw[SET][j] = *reinterpret_cast<const v4f*>(p.A + (unsigned)(abase[j] + k0));
        } else if (CONV) {
            // a padding tap reads a block of zeros instead (pointer select, the load itself stays unconditional);
            // with InstanceNorm on load its 1/std comes from the zeros too, so (0 - mean) * 0 = 0 needs no select
#pragma unroll
            for (int j = 0; j < AR; ++j) {
                const bool ok = (unsigned)(riy[j] + t_kh) < (unsigned)p.Hi && (unsigned)(rix[j] + t_kw) < (unsigned)p.Wi;
                const float* src = ok ? p.A + (unsigned)(abase[j] + delta) : p.zeros;
                araw[SET][j] = *reinterpret_cast<const v4f*>(src);
                if (NORM) {
                    const float* rs = ok ? p.nrstd + (unsigned)(nbase[j] + t_cc) : p.zeros;
                    amu[SET][j] = *reinterpret_cast<const v4f*>(p.nmean + (unsigned)(nbase[j] + t_cc));
                    ars[SET][j] = *reinterpret_cast<const v4f*>(rs);
                }
            }
            if (loaded + 1 < n_chunks && !(dbg & 1)) {  // advance to the next chunk (loads past the end re-read the last one)
                t_cc += BK;
                delta += BK;  // channels, then the next pixel of the row, are contiguous in NHWC
                wk += BK;     // ... and in the [kh][kw][Cin] weight row
                if (t_cc >= p.Cin) {
                    t_cc = 0;
                    if (++t_kw == kw_end) {
                        t_kw = kw_lo;
                        ++t_kh;
                        delta += row_jump;
                        wk += wrow_jump;
                    }
                }
            }
        } else {  // A_CONV1: k = (plane, kh, kw) with an 8x8 kernel: a chunk is BK/8 kernel rows of one plane
            const int plane = k0 >> 6;
            const int kh = ((k0 & 63) >> 3) + (c4 >> 3);
            const int kw = c4 & 7;
            const float* src = plane == 0 ? p.planes[0] : plane == 1 ? p.planes[1] : plane == 2 ? p.planes[2] : p.planes[3];
#pragma unroll
            for (int j = 0; j < AR; ++j) {
                const int iy = riy[j] + kh;
                const bool rowok = rvalid[j] && iy >= 0 && iy < p.Hi;
                const float* rowp = src + (rb[j] * p.Hi + min(max(iy, 0), p.Hi - 1)) * p.Wi;
                const int ix0 = rix[j] + kw;
                araw[SET][j].x = rowp[min(max(ix0, 0), p.Wi - 1)];
                araw[SET][j].y = rowp[min(max(ix0 + 1, 0), p.Wi - 1)];
                araw[SET][j].z = rowp[min(max(ix0 + 2, 0), p.Wi - 1)];
                araw[SET][j].w = rowp[min(max(ix0 + 3, 0), p.Wi - 1)];
                unsigned m4 = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) m4 |= (rowok && ix0 + q >= 0 && ix0 + q < p.Wi) ? (1u << q) : 0u;
                okmask[SET] |= m4 << (4 * j);
            }
        }
#pragma unroll
        for (int j = 0; j < BR; ++j) breg[SET][j] = *reinterpret_cast<const v4f*>(Wsrc + (unsigned)(wbase[j] + (CONV ? wk_load : k0)));
        ++loaded;
    };

    // -- store phase: normalise / zero-fill the staged operand, then write it to LDS buffer `buf`
    // (`h` is std::true_type for the split-operand fp16 form, std::false_type for fp32 instructions: a HALF kernel
    // holds both, see the range check behind the pipeline)
    auto store_chunk = [&](auto h, auto set_tag, int buf) {
        constexpr bool H = decltype(h)::value;
        constexpr int SET = decltype(set_tag)::value;
        float* As = As0 + buf * BM * LDK;
        float* Bs = Bs0 + buf * BN * LDK;
#pragma unroll
        for (int j = 0; j < AR; ++j) {
            float4 v = make_float4(araw[SET][j].x, araw[SET][j].y, araw[SET][j].z, araw[SET][j].w);
            if (LNORM) {
                v.x = (v.x - rmean[j]) * rrstd[j];
                v.y = (v.y - rmean[j]) * rrstd[j];
                v.z = (v.z - rmean[j]) * rrstd[j];
                v.w = (v.w - rmean[j]) * rrstd[j];
            }
            if (NORM) {
                v.x = (v.x - amu[SET][j].x) * ars[SET][j].x;
                v.y = (v.y - amu[SET][j].y) * ars[SET][j].y;
                v.z = (v.z - amu[SET][j].z) * ars[SET][j].z;
                v.w = (v.w - amu[SET][j].w) * ars[SET][j].w;
            }
            if (AMODE == A_CONV1) {
                const unsigned m4 = okmask[SET] >> (4 * j);
                v.x = (m4 & 1u) ? v.x : 0.f;
                v.y = (m4 & 2u) ? v.y : 0.f;
                v.z = (m4 & 4u) ? v.z : 0.f;
                v.w = (m4 & 8u) ? v.w : 0.f;
            }
            if (H) {
                amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                f16x4 hi, lo;
                split4(v, hi, lo);
                _Float16* row = reinterpret_cast<_Float16*>(As + (lrow + SWEEP * j) * LDK);
                *reinterpret_cast<f16x4*>(row + c4) = hi;
                *reinterpret_cast<f16x4*>(row + BK + c4) = lo;
            } else {
                *reinterpret_cast<float4*>(As + (lrow + SWEEP * j) * LDK + c4) = v;
            }
        }
#pragma unroll
        // (HALF: the weights were split when they were packed - every 64 floats of a row hold [64 halves hi | 64 halves
        // lo] of those 64 weights, the layout of an LDS row - so their staging is the same plain copy)
        for (int j = 0; j < BR; ++j) *reinterpret_cast<v4f*>(Bs + (lrow + SWEEP * j) * LDK + c4) = breg[SET][j];
    };

    // -- matrix phase, in two parts: fetch this wave's operand fragments of the chunk in LDS buffer `buf` ...
    // (HALF: a "k-block" is a 16-deep step: lane (i, h) holds halves 16s + 8h .. + 7 of row i, hi and lo)
    constexpr int NKB32 = KB / KWC;                      // k-blocks of a chunk that this wave multiplies: fp32 form
    constexpr int NKB16 = HALF ? (BK / 16) / KWC : 1;    // ... split-operand form
    static_assert(NKB32 >= 1 && NKB16 >= 1, "more k-parts than k-blocks in a chunk");
    float4 fa[NKB32][NTR], fb[NKB32][NTC];
    f16x8 ha[NKB16][NTR][2], hb[NKB16][NTC][2];
    auto read_fragments = [&](auto h, int buf) {
        constexpr bool H = decltype(h)::value;
        constexpr int NKB = H ? NKB16 : NKB32;
        const float* As = As0 + buf * BM * LDK + (row0 + fr) * LDK + fh;
        const float* Bs = Bs0 + buf * BN * LDK + (col0 + fr) * LDK + fh;
#pragma unroll
        for (int kbi = 0; kbi < NKB; ++kbi) {
            const int kb = kpart + kbi * KWC;
            if (H) {
                // float offset fh = 4h is half offset 8h; step kb starts at half 16 kb = float 8 kb
#pragma unroll
                for (int t = 0; t < NTR; ++t) {
                    ha[kbi][t][0] = *reinterpret_cast<const f16x8*>(As + t * 32 * LDK + kb * 8);
                    ha[kbi][t][1] = *reinterpret_cast<const f16x8*>(As + t * 32 * LDK + kb * 8 + BK / 2);
                }
#pragma unroll
                for (int t = 0; t < NTC; ++t) {
                    hb[kbi][t][0] = *reinterpret_cast<const f16x8*>(Bs + t * 32 * LDK + kb * 8);
                    hb[kbi][t][1] = *reinterpret_cast<const f16x8*>(Bs + t * 32 * LDK + kb * 8 + BK / 2);
                }
            } else {
#pragma unroll
                for (int t = 0; t < NTR; ++t) fa[kbi][t] = *reinterpret_cast<const float4*>(As + t * 32 * LDK + kb * 8);
#pragma unroll
                for (int t = 0; t < NTC; ++t) fb[kbi][t] = *reinterpret_cast<const float4*>(Bs + t * 32 * LDK + kb * 8);
            }
        }
    };
    // ... and multiply them.  Weights are the MFMA A operand: the product comes out transposed, so that a lane owns
    // runs of four consecutive output columns of one output row (16-byte epilogue).
    auto multiply_fragment = [&](auto h, int kbi) {
        // (term by term over the wave's tiles: consecutive instructions then write different accumulators; per
        // accumulator the order of the sum is lo-hi, hi-lo, hi-hi / x, y, z, w whatever the number of tiles)
        if (decltype(h)::value) {
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int tr = 0; tr < NTR; ++tr)
#pragma unroll
                    for (int tc = 0; tc < NTC; ++tc)
                        acc[tr * NTC + tc] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hb[kbi][tc][term == 0 ? 1 : 0], ha[kbi][tr][term == 1 ? 1 : 0],
                                                                                    acc[tr * NTC + tc], 0, 0, 0);
            return;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int tr = 0; tr < NTR; ++tr)
#pragma unroll
                for (int tc = 0; tc < NTC; ++tc)
                    acc[tr * NTC + tc] = __builtin_amdgcn_mfma_f32_32x32x2f32(component(fb[kbi][tc], e), component(fa[kbi][tr], e), acc[tr * NTC + tc], 0, 0, 0);
    };

    // one iteration: multiply chunk `it` (LDS buffer it & 1), write chunk it+1 from register set SET into the other
    // buffer and reuse that set for the loads of the chunk that is PF further on
    // (`role`: 0 = this wave stages and multiplies, 1 = it only stages, 2 = it only multiplies)
    // Timing build only, debug bit 64: workgroup 0 leaves clock readings in p.partial, four per wave and iteration.
    auto stamp = [&](int it, int k) {
#if DANRA_IGEMM_DEBUG
        if ((dbg & 64) && blockIdx.x == 0 && it < 64) {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): what was issued so far has landed (vmcnt, expcnt left alone)
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (lane == 0) reinterpret_cast<unsigned long long*>(p.partial)[(wave * 64 + it) * 4 + k] = t;
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
    };
    // Range check of the split (HALF): a staged value of 2^15 or more (or an infinity) does not fit the fp16 high half,
    // and a tile whose largest value is below 2^-8 would be kept to an absolute 2^-25 only, i.e. to fewer than 17 bits
    // (an all-zero tile is exact).  The staging waves cast their votes ahead of the last barrier of the loop - the last
    // iteration stages nothing new, so `amax` is final by then - and everybody reads the result behind it.
    auto vote = [&](int it) {
        if (it != n_chunks - 1) return;
        const bool too_big = !(amax < 32768.f);
        const bool normal = amax >= 0.00390625f;
        if (__any(too_big) && lane == 0) rangeFlag[0] = 1;
        if (__any(normal) && lane == 0) rangeFlag[1] = 1;
        if (__any(amax > 0.f) && lane == 0) rangeFlag[2] = 1;
    };
    auto iteration = [&](auto h, auto set_tag, auto load_tag, int it, auto role) {
        constexpr int NKB = decltype(h)::value ? NKB16 : NKB32;
        constexpr int ROLE = decltype(role)::value;
        if (ROLE == 1) {
            // (loads issued ahead of the conversions and LDS writes, or by half of the staging waves only, took longer)
            stamp(it, 0);
            store_chunk(h, set_tag, (it + 1) & 1);
            if (decltype(h)::value) vote(it);
            stamp(it, 1);
            load_chunk(load_tag);
            stamp(it, 2);
            __syncthreads();
            stamp(it, 3);
            return;
        }
        if (ROLE == 2) {
            stamp(it, 0);
            read_fragments(h, it & 1);
            stamp(it, 1);
#pragma unroll
            for (int kbi = 0; kbi < NKB; ++kbi) multiply_fragment(h, kbi);
            stamp(it, 2);
            __syncthreads();
            stamp(it, 3);
            return;
        }
        stamp(it, 0);
        if (!(dbg & 2)) {
            read_fragments(h, (dbg & 4) ? 0 : (it & 1));
            multiply_fragment(h, 0);
        }
        // the staging work sits between the first and the remaining matrix instructions of the wave
        __builtin_amdgcn_sched_barrier(0);
        stamp(it, 1);
        if (!(dbg & 4)) store_chunk(h, set_tag, (it + 1) & 1);  // (in the last iteration a copy of the last chunk that nobody reads)
        if (decltype(h)::value) vote(it);
        stamp(it, 2);
        if (!(dbg & 16)) load_chunk(load_tag);
        __builtin_amdgcn_sched_barrier(0);
        if (!(dbg & 2)) {
#pragma unroll
            for (int kbi = 1; kbi < NKB; ++kbi) multiply_fragment(h, kbi);
        }
        __builtin_amdgcn_sched_barrier(0);
        stamp(it, 3);
        if (!(dbg & 4)) __syncthreads();
    };
    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, 1>;
    auto pipeline = [&](auto h, auto role) {
        constexpr bool stager = decltype(role)::value != 2;
        phase_stamp(1);
        rewind();
        if (decltype(h)::value && decltype(role)::value != 0) {
            // A staging-only wave has registers to spare and a short iteration: it keeps the loads of TWO chunks in
            // flight.  Chunk c goes through register set c & 1; iteration `it` stores chunk it+1 and issues the
            // loads of chunk it+3.  (For waves that also multiply the second set bought nothing and cost the
            // 8-wave kernel its second workgroup per CU.)
            if (stager) {
                load_chunk(Set0{});     // chunk 0
                store_chunk(h, Set0{}, 0);
                load_chunk(Set1{});     // chunk 1
                load_chunk(Set0{});     // chunk 2
            }
            __syncthreads();
            phase_stamp(2);
            for (int it = 0; it < n_chunks; it += 2) {
                iteration(h, Set1{}, Set1{}, it, role);
                if (it + 1 < n_chunks) iteration(h, Set0{}, Set0{}, it + 1, role);
            }
            phase_stamp(3);
            return;
        }
        if (stager) {
            load_chunk(Set0{});    // chunk 0
            store_chunk(h, Set0{}, 0);
            load_chunk(Set0{});    // chunk 1
        }
        __syncthreads();
        for (int it = 0; it < n_chunks; ++it) iteration(h, Set0{}, Set0{}, it, role);  // stores chunk it+1, loads chunk it+2
    };
    float out_scale = HALF ? p.wscale : 1.0f;  // the split weights carry a power-of-two scale
    // The whole product, as one role sees it.  Every role goes through the same sequence of barriers.
    auto product = [&](auto role) {
        pipeline(std::integral_constant<bool, HALF>{}, role);
        if (!HALF) return;
        // A failed range check (see `vote`): the workgroup multiplies its tile again with fp32 instructions, from the
        // fp32 weights.  (A NaN comes out as a NaN in both forms.)
        const bool redo = rangeFlag[0] != 0 || (rangeFlag[2] != 0 && rangeFlag[1] == 0);
        if (__builtin_amdgcn_readfirstlane(redo ? 1 : 0)) {
#pragma unroll
            for (int t = 0; t < NACC; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
            Wsrc = p.W;
            out_scale = 1.0f;
            pipeline(std::false_type{}, role);
        }
    };
    if (!SPEC) product(std::integral_constant<int, 0>{});

    // ---- sum the partial accumulators of the k-parts (fixed order: k-part 0, 1, 2, 3)
    constexpr int LDO = BN + 4;
    float* tile = smem;
    if (SPEC) {
        // every multiplying wave leaves its whole partial tile in LDS; the epilogue below adds the four up.  (The
        // trailing __syncthreads() of the loop has retired every read of the staging buffers.)
        if (stager) {
            product(std::integral_constant<int, 1>{});
        } else {
            product(std::integral_constant<int, 2>{});
            float* mine = smem + kpart * BM * LDO;
#pragma unroll
            for (int tr = 0; tr < NTR; ++tr)
#pragma unroll
                for (int tc = 0; tc < NTC; ++tc)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<float4*>(mine + (tr * 32 + fr) * LDO + tc * 32 + 8 * g + fh) =
                            make_float4(acc[tr * NTC + tc][4 * g], acc[tr * NTC + tc][4 * g + 1], acc[tr * NTC + tc][4 * g + 2], acc[tr * NTC + tc][4 * g + 3]);
        }
        __syncthreads();
    } else {
        if (KWC > 1) {
            // the trailing __syncthreads() of the loop has retired every read of the staging buffers
            float* red = smem;
            if (kpart > 0) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[((((kpart - 1) * 4 + tw) * NT + t) * 16 + r) * 64 + lane] = acc[t][r];
            }
            __syncthreads();
            if (kpart > 0) return;  // (a wave that has ended no longer counts at a barrier)
#pragma unroll
            for (int kp = 1; kp < KWC; ++kp)
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] += red[((((kp - 1) * 4 + tw) * NT + t) * 16 + r) * 64 + lane];
        }

        // ---- epilogue.  acc[t][r] = out[row m0 + row0 + (lane & 31)][column n0 + col0 + 32t + (r & 3) + 8*(r >> 2) + 4*(lane >> 5)].
        // The tile goes through LDS once so that 16 consecutive lanes own one whole 64-column output row: residual
        // loads and stores are then 256-byte contiguous per row, and a thread needs one column offset for all its rows.
        __syncthreads();  // (the waves still alive) every read of the staging / reduction buffers has retired
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(tile + (row0 + fr) * LDO + col0 + t * 32 + 8 * g + fh) =
                    make_float4(acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]);
        __syncthreads();
    }
    phase_stamp(4);
    if (!SPEC && p.bias && p.ksplit <= 1) bv = *reinterpret_cast<const float4*>(p.bias + n);
    float* const outp = second ? p.out2 : p.out;
    const unsigned long long co = col_offset(p, second ? n - p.n_split : n);
#pragma unroll
    for (int j = 0; j < (BM + ESWEEP - 1) / ESWEEP; ++j) {
        const int row = (tid >> 4) + ESWEEP * j;
        const int m = m0 + row;
        if ((BM % ESWEEP) && row >= BM) continue;
        if (m >= p.M) continue;
        float4 v = *reinterpret_cast<const float4*>(tile + row * LDO + oc4);
        if (SPEC) {
#pragma unroll
            for (int kp = 1; kp < 4; ++kp) {
                const float4 o = *reinterpret_cast<const float4*>(tile + (kp * BM + row) * LDO + oc4);
                v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            }
        }
        if (HALF) {
            v.x *= out_scale; v.y *= out_scale; v.z *= out_scale; v.w *= out_scale;
        }
        if (p.ksplit > 1) {
            *reinterpret_cast<float4*>(p.partial + ((size_t)blockIdx.z * p.M + m) * p.N + n) = v;
            continue;
        }
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        if (second) {
            *reinterpret_cast<float4*>(outp + rowOff[row] + co) = v;
            continue;
        }
        if (p.res) {
            const float4 rv = SPEC ? pre_res[j] : *reinterpret_cast<const float4*>(p.res + (size_t)rowOut[row] * p.ldr + n);
            v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
        }
        v = activate(p.act, v);
        if (p.tbias) {
            const float4 tv = SPEC ? pre_tb[j] : *reinterpret_cast<const float4*>(p.tbias + (size_t)rowImg[row] * p.tb_stride + n);
            v.x += tv.x; v.y += tv.y; v.z += tv.z; v.w += tv.w;
        }
        *reinterpret_cast<float4*>(outp + rowOff[row] + co) = v;
    }
#if DANRA_IGEMM_DEBUG
    if (dbg & 64) __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0): the stores have been acknowledged
#endif
    phase_stamp(5);
}

// Sums the split-K slabs in slab order (bitwise reproducible) and applies the epilogue.
// One thread per four consecutive columns of one row (they lie on one side of n_split and are contiguous in every
// store mode: launch_igemm checks).  A thread has one or two such groups, so the kernel is a chain of memory
// latencies: every load of a group - up to eight slabs, bias, residual, embedding bias - is issued before the first
// value is used (with the slab loads inside a loop of unknown length each waited for the one before: 5-6 us per
// launch on the 4x4 and 2x2 maps, against 3 with the loads in flight together).
__global__ __launch_bounds__(256) void k_splitk_epilogue(const IGemmArgs p) {
    const size_t total4 = (size_t)p.M * p.N / 4;
    const size_t slab = (size_t)p.M * p.N;
    constexpr int KS = 8;  // slabs loaded together
    for (size_t i4 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i4 < total4; i4 += (size_t)gridDim.x * blockDim.x) {
        const size_t idx = i4 * 4;
        float4 part[KS];
#pragma unroll
        for (int z = 0; z < KS; ++z)
            if (z < p.ksplit) part[z] = *reinterpret_cast<const float4*>(p.partial + (size_t)z * slab + idx);
        const int mt = (int)((unsigned)idx / (unsigned)p.N);  // row of the GEMM; the slabs are in that order (M N < 2^31: launch_igemm checks)
        const int n = (int)((unsigned)idx - (unsigned)mt * (unsigned)p.N);
        const int m = out_row(p, mt);
        int img;
        const unsigned long long ro = row_offset(p, m, img);
        const bool second = p.out2 && n >= p.n_split;  // second output: bias only
        float4 bias = make_float4(0.f, 0.f, 0.f, 0.f), res = bias, tb = bias;
        if (p.bias) bias = *reinterpret_cast<const float4*>(p.bias + n);
        if (p.res && !second) res = *reinterpret_cast<const float4*>(p.res + (size_t)m * p.ldr + n);
        if (p.tbias && !second) tb = *reinterpret_cast<const float4*>(p.tbias + (size_t)img * p.tb_stride + n);
        float4 v = part[0];
#pragma unroll
        for (int z = 1; z < KS; ++z)
            if (z < p.ksplit) { v.x += part[z].x; v.y += part[z].y; v.z += part[z].z; v.w += part[z].w; }
        for (int z = KS; z < p.ksplit; ++z) {
            const float4 o = *reinterpret_cast<const float4*>(p.partial + (size_t)z * slab + idx);
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        v.x += bias.x; v.y += bias.y; v.z += bias.z; v.w += bias.w;
        if (!second) {
            if (p.res) { v.x += res.x; v.y += res.y; v.z += res.z; v.w += res.w; }
            v = activate(p.act, v);
            if (p.tbias) { v.x += tb.x; v.y += tb.y; v.z += tb.z; v.w += tb.w; }
        }
        *reinterpret_cast<float4*>((second ? p.out2 : p.out) + ro + col_offset(p, second ? n - p.n_split : n)) = v;
    }
}

// ------------------------------------------------------------------------------------------------
// Row-stationary GEMM for K == 64 (the 64-channel attention projections and transposed convolutions:
// M is tens of thousands of tokens, N = 64..256, so there is no K loop to pipeline and the work is
// bound by moving A and the output).  The whole weight matrix sits in LDS ([N][68] floats, loaded once
// per workgroup, every load in flight before the first LDS write); after that single barrier every
// wave works alone on its own 32 rows: it loads them straight into MFMA fragment registers (lane
// (i, h) holds k = 8u+4h..8u+4h+3 of row i, the same convention as k_igemm), applies the LayerNorm in
// registers (a row's 64 values live in lanes i and i+32), then walks N in 32-column steps of 32 MFMAs +
// epilogue.  The product is formed transposed (weights as the MFMA A operand), so a lane ends up with
// runs of four consecutive output columns of ONE output row: 16-byte stores, 4x fewer store
// instructions than the row-major accumulator layout, and the lane's own row offset serves them all.
// With no barrier in the walk the waves of a CU drift apart, so loads, matrix work and stores of
// different waves overlap.
// ---- fragment helpers of the K = 64 kernels (k_gemm_k64, k_attn_tail64).  A wave holds 32 rows of 64 values as
// float4 a[8]: lane (row, h) has k = 8 u + 4 h .. + 3 in a[u].
// The split-operand form of those rows: k-step s takes a[2s] and a[2s + 1], i.e. k = 16 s + 4 h + {0..3, 8..11}, the
// order in which launch_split_weights(frag64) lays out the halves of a weight row.  Returns (wave-uniform) whether
// the rows are in the range where the split keeps 22 bits: nothing at or above 2^15 (or non-finite), and something at
// or above 2^-8 unless all is zero.
__device__ __forceinline__ bool split_frag64(const float4 (&a)[8], f16x8 (&ah)[4], f16x8 (&al)[4]) {
    float amax = 0.f;
#pragma unroll
    for (int kb = 0; kb < 8; ++kb)
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(a[kb].x), fabsf(a[kb].y))), fmaxf(fabsf(a[kb].z), fabsf(a[kb].w)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        unsigned hi[4], lo[4];
        split_pair_f16(a[2 * st].x, a[2 * st].y, hi[0], lo[0]);
        split_pair_f16(a[2 * st].z, a[2 * st].w, hi[1], lo[1]);
        split_pair_f16(a[2 * st + 1].x, a[2 * st + 1].y, hi[2], lo[2]);
        split_pair_f16(a[2 * st + 1].z, a[2 * st + 1].w, hi[3], lo[3]);
        ah[st] = __builtin_bit_cast(f16x8, (u32x4){hi[0], hi[1], hi[2], hi[3]});
        al[st] = __builtin_bit_cast(f16x8, (u32x4){lo[0], lo[1], lo[2], lo[3]});
    }
    return !__any(!(amax < 32768.f)) && (__any(amax >= 0.00390625f) || !__any(amax > 0.f));
}
// One 32-column tile of W a^T (weights as the MFMA A operand: the product comes out transposed).  Bs = the tile's
// first weight row + this lane's row and half: in the split form a row holds [64 halves hi | 64 halves lo] and
// halves (2 s + h) 8 .. + 7 start at float 8 s + 4 h.
__device__ __forceinline__ f32x16 tile64_half(const float* Bs, const f16x8 (&ah)[4], const f16x8 (&al)[4]) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        const f16x8 wh = *reinterpret_cast<const f16x8*>(Bs + st * 8);
        const f16x8 wl = *reinterpret_cast<const f16x8*>(Bs + st * 8 + 32);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, ah[st], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, al[st], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, ah[st], acc, 0, 0, 0);
    }
    return acc;
}
// The same products untransposed (activations as the A operand): lane (column, h) ends up with rows 8 u + 4 h + e of its
// column in acc[4 u + e] - four consecutive rows of one column per register group, the order in which the attention
// core's V fragments hold their keys.  Identical arithmetic, term by term.
__device__ __forceinline__ f32x16 tile64_half_rows(const float* Bs, const f16x8 (&ah)[4], const f16x8 (&al)[4]) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        const f16x8 wh = *reinterpret_cast<const f16x8*>(Bs + st * 8);
        const f16x8 wl = *reinterpret_cast<const f16x8*>(Bs + st * 8 + 32);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[st], wl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[st], wh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[st], wh, acc, 0, 0, 0);
    }
    return acc;
}
__device__ __forceinline__ f32x16 tile64_f32_rows(const float* Bs, const float4 (&a)[8]) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
        const float4 b = *reinterpret_cast<const float4*>(Bs + kb * 8);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kb].x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kb].y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kb].z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kb].w, b.w, acc, 0, 0, 0);
    }
    return acc;
}
__device__ __forceinline__ f32x16 tile64_f32(const float* Bs, const float4 (&a)[8]) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
        const float4 b = *reinterpret_cast<const float4*>(Bs + kb * 8);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, a[kb].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, a[kb].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, a[kb].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, a[kb].w, acc, 0, 0, 0);
    }
    return acc;
}

// HALF: the products run as split-operand fp16 products (see k_igemm) on v_mfma_f32_32x32x16_f16: 72 matrix
// instructions of 32 cycles per 32 x 192 task instead of 192 of 64 cycles, which leaves the kernel to the memory
// system.  p.Wh then holds the weights split at pack time in fragment order: per row 64 halves hi, then 64 halves lo,
// the halves of k-step s and lane half h being k = 16 s + 4 h + {0..3, 8..11} - the k values whose activations lane
// (row, h) already holds in a[2s] and a[2s + 1], so the activations are split where they are.  A wave whose rows hold
// a value of 2^15 or more, or nothing above 2^-8 (and not all zeros), multiplies with fp32 instructions from the fp32
// weights in global memory instead.
template <int NW4, bool HALF>  // NW4 = weight float4 loads a thread keeps in flight
__global__ __launch_bounds__(1024) void k_gemm_k64(const IGemmArgs p) {
    constexpr int LDK = 68;
    extern __shared__ __attribute__((aligned(16))) float Wsm[];  // [N][LDK] then the column-offset table
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int waves = blockDim.x >> 6;
    const int fr = lane & 31;
    const int fh = (lane >> 5) * 4;
    // wave task = (32-row tile, 1/nsplit of the columns); the tasks of one tile sit in adjacent waves
    const int nsplit = p.force_kw > 0 ? p.force_kw : 1;
    const int task = blockIdx.x * waves + wave;
    const int m0 = (task / nsplit) * 32;
    const int nt_per = p.N / 32 / nsplit;
    const int nt_begin = (task % nsplit) * nt_per;
    unsigned long long* coloff = reinterpret_cast<unsigned long long*>(Wsm + (size_t)p.N * LDK);  // [N/4]
    float* bias_s = reinterpret_cast<float*>(coloff + p.N / 4);                                  // [N]
    constexpr int LDO = 36;
    float* stage = bias_s + p.N + wave * 32 * LDO;  // this wave's 32x32 output staging tile

    // ---- this wave's 32 rows, in fragment order (clamped rows are computed and never stored)
    const int mrow = m0 + fr;
    const int arow = min(mrow, p.M - 1);
    float4 a[8];
    {
        const float* src = p.A + (size_t)arow * p.lda + fh;
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) a[kb] = *reinterpret_cast<const float4*>(src + kb * 8);
    }
    // ---- the weights: N rows of 64 floats -> LDS
    const int total = p.N * 16;
    for (int q0 = tid; q0 < total; q0 += NW4 * (int)blockDim.x) {
        float4 w[NW4];
#pragma unroll
        for (int j = 0; j < NW4; ++j) {
            const int q = min(q0 + j * (int)blockDim.x, total - 1);
            w[j] = *reinterpret_cast<const float4*>((HALF ? p.Wh : p.W) + (size_t)q * 4);
        }
#pragma unroll
        for (int j = 0; j < NW4; ++j) {
            const int q = q0 + j * (int)blockDim.x;
            if (q < total) *reinterpret_cast<float4*>(Wsm + (q >> 4) * LDK + (q & 15) * 4) = w[j];
        }
    }
    for (int g = tid; g < p.N / 4; g += blockDim.x) coloff[g] = col_offset(p, g * 4);
    for (int n = tid; n < p.N; n += blockDim.x) bias_s[n] = p.bias ? p.bias[n] : 0.f;
    int img;
    const unsigned long long my_off = row_offset(p, arow, img);
    if (p.ln) {  // two-pass LayerNorm statistics; lanes i and i+32 hold the two halves of row i
        float s = 0.f;
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) s += (a[kb].x + a[kb].y) + (a[kb].z + a[kb].w);
        s += __shfl_xor(s, 32);
        const float mean = s * (1.0f / 64.0f);
        float q = 0.f;
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            const float dx = a[kb].x - mean, dy = a[kb].y - mean, dz = a[kb].z - mean, dw = a[kb].w - mean;
            q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
        q += __shfl_xor(q, 32);
        const float rstd = 1.0f / sqrtf(q * (1.0f / 64.0f) + 1e-5f);
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            a[kb].x = (a[kb].x - mean) * rstd;
            a[kb].y = (a[kb].y - mean) * rstd;
            a[kb].z = (a[kb].z - mean) * rstd;
            a[kb].w = (a[kb].w - mean) * rstd;
        }
    }
    // the walk stores through a wave-private LDS tile so that 8 consecutive lanes own 32 consecutive columns
    // of one row (128-byte runs); lane l then serves rows (l >> 3) + 8j
    const int orow = lane >> 3;
    const int oc4 = (lane & 7) * 4;
    unsigned long long roff4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned lo = __shfl((unsigned)my_off, orow + 8 * j);
        const unsigned hi = __shfl((unsigned)(my_off >> 32), orow + 8 * j);
        roff4[j] = ((unsigned long long)hi << 32) | lo;
    }
    __syncthreads();
    if (m0 >= p.M) return;

    // HALF: the wave's activations as fp16 halves, and whether they are in the range where that keeps 22 bits
    f16x8 ah[HALF ? 4 : 1], al[HALF ? 4 : 1];
    bool split_ok = false;
    if constexpr (HALF) split_ok = split_frag64(a, ah, al);
    const float out_scale = (HALF && split_ok) ? p.wscale : 1.0f;  // the split weights carry a power-of-two scale
    const int ldw = p.ldw ? p.ldw : 64;

    // ---- QKV projection of a head_dim-16 block with K and V left pre-split for the attention core (p.pre_*): a tile
    // of 32 columns is two heads of q, k or v; the wave's 32 rows are two groups of 16 keys of one image
    const bool presplit = HALF && p.pre_flags != nullptr;
    const int pre_b = presplit ? div_by(m0, p.qkv_L, p.sh_qkv_l) : 0, pre_l0 = m0 - pre_b * p.qkv_L;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

    for (int nt = nt_begin; nt < nt_begin + nt_per; ++nt) {
        if constexpr (HALF) {
            if (presplit && nt * 32 >= 2 * p.qkv_C) {   // ---- V: formed untransposed, split and stored from the registers
                f32x16 acc = split_ok ? tile64_half_rows(Wsm + (nt * 32 + fr) * LDK + fh, ah, al)
                                      : tile64_f32_rows(p.W + (size_t)(nt * 32 + fr) * ldw + fh, a);
                const float bcol = bias_s[nt * 32 + fr];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = acc[r] * out_scale + bcol;
                // lane (column fr, h): acc[4 u + e] = v[key 8 u + 4 h + e][column]; head = 2 (nt & 1) + (fr >> 4), d = fr & 15
                const int head = 2 * ((nt * 32 - 2 * p.qkv_C) >> 5) + (fr >> 4), d = fr & 15, hh = lane >> 5;
                const int bh = pre_b * p.qkv_heads + head;
                float mj[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float m = 0.f;
#pragma unroll
                    for (int r = 0; r < 8; ++r) m = fmaxf(m, fabsf(acc[8 * j + r]));
                    mj[j] = m;
                }
                // flags of (head, group of 16 keys j): lanes of the head are those with the same bit 4
                int flag[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const unsigned long long big = __ballot(!(mj[j] < 32768.f)), small = __ballot(mj[j] < 0.0009765625f);
                    const unsigned long long mine = (fr >> 4) ? 0xffff0000ffff0000ull : 0x0000ffff0000ffffull;
                    flag[j] = ((big & mine) != 0 || (small & mine) == mine) ? 1 : 0;
                }
                if ((lane & 15) == 0 && hh == 0) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        p.pre_flags[((size_t)bh * (p.qkv_L >> 4) + (pre_l0 >> 4) + j) * 2 + 1] = flag[j];
                }
                // slot group g of the 32-key half: keys 4 g + e (j = 0) and 16 + 4 g + e (j = 1); this lane holds g = hh and 2 + hh
                float* vt = p.pre_vs + ((size_t)bh * p.qkv_L + (pre_l0 & ~63)) * 16;           // the 64-key tile's 4 KB
                const int half = (pre_l0 >> 5) & 1;
#pragma unroll
                for (int gg = 0; gg < 2; ++gg) {
                    const int g = 2 * gg + hh, u0 = gg, u1 = 2 + gg;   // rows 8 u + 4 h + e: u = gg -> keys 4 g + e, u = 2 + gg -> 16 + 4 g + e
                    unsigned h4[4], l4[4];
                    split_scaled_pair_f16(acc[4 * u0], acc[4 * u0 + 1], h4[0], l4[0]);
                    split_scaled_pair_f16(acc[4 * u0 + 2], acc[4 * u0 + 3], h4[1], l4[1]);
                    split_scaled_pair_f16(acc[4 * u1], acc[4 * u1 + 1], h4[2], l4[2]);
                    split_scaled_pair_f16(acc[4 * u1 + 2], acc[4 * u1 + 3], h4[3], l4[3]);
                    const int at = (half * 512 + (d * 4 + g) * 8) / 2;   // floats; hi block, the lo block 512 floats further
                    *reinterpret_cast<u32x4*>(vt + at) = u32x4{h4[0], h4[1], h4[2], h4[3]};
                    *reinterpret_cast<u32x4*>(vt + 512 + at) = u32x4{l4[0], l4[1], l4[2], l4[3]};
                }
                if (flag[0] | flag[1]) {   // (rare) the fp32 values of the flagged head's keys, [key][16]
                    float* v32 = p.out + 2 * p.qkv_plane + ((size_t)bh * p.qkv_L + pre_l0) * 16 + d;
#pragma unroll
                    for (int r = 0; r < 16; ++r) v32[(size_t)(8 * (r >> 2) + 4 * hh + (r & 3)) * 16] = acc[r];
                }
                continue;
            }
        }
        // acc[r] = out[row m0 + (lane & 31)][column nt*32 + (r & 3) + 8*(r >> 2) + 4*(lane >> 5)]
        f32x16 acc;
        if constexpr (HALF) {
            // (a wave out of range - rare - takes the fp32 instructions with the fp32 weights from global memory)
            if (split_ok) acc = tile64_half(Wsm + (nt * 32 + fr) * LDK + fh, ah, al);
            else acc = tile64_f32(p.W + (size_t)(nt * 32 + fr) * ldw + fh, a);
        } else {
            acc = tile64_f32(Wsm + (nt * 32 + fr) * LDK + fh, a);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(stage + fr * LDO + 8 * g + fh) =
                make_float4(acc[4 * g] * out_scale, acc[4 * g + 1] * out_scale, acc[4 * g + 2] * out_scale, acc[4 * g + 3] * out_scale);
        const int n = nt * 32 + oc4;
        const float4 bv = *reinterpret_cast<const float4*>(bias_s + n);
        const unsigned long long co = coloff[n >> 2];
        if constexpr (HALF) {
            if (presplit && nt * 32 >= p.qkv_C) {   // ---- K: split on the way out; lane = columns oc4 .. + 3 of rows orow + 8 j
                const int head = 2 * ((nt * 32 - p.qkv_C) >> 5) + (oc4 >> 4), d4 = oc4 & 15;
                const int bh = pre_b * p.qkv_heads + head;
                float4 v[4];
                float mj[2] = {0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = *reinterpret_cast<const float4*>(stage + (orow + 8 * j) * LDO + oc4);
                    v[j].x += bv.x; v[j].y += bv.y; v[j].z += bv.z; v[j].w += bv.w;
                    mj[j >> 1] = fmaxf(fmaxf(mj[j >> 1], fmaxf(fabsf(v[j].x), fabsf(v[j].y))), fmaxf(fabsf(v[j].z), fabsf(v[j].w)));
                    unsigned h2[2], l2[2];
                    split_scaled_pair_f16(v[j].x, v[j].y, h2[0], l2[0]);
                    split_scaled_pair_f16(v[j].z, v[j].w, h2[1], l2[1]);
                    *reinterpret_cast<u32x4*>(p.pre_ks + ((size_t)bh * p.qkv_L + pre_l0 + orow + 8 * j) * 16 + d4) =
                        u32x4{h2[0], h2[1], l2[0], l2[1]};
                }
                // flags of (head, group of 16 keys): rows orow + 8 j with j < 2 are the first group; the head's lanes have the same bit 2
                const unsigned long long mine = (oc4 >> 4) ? 0xf0f0f0f0f0f0f0f0ull : 0x0f0f0f0f0f0f0f0full;
                int flag[2];
#pragma unroll
                for (int gq = 0; gq < 2; ++gq) flag[gq] = (__ballot(!(mj[gq] < 32768.f)) & mine) != 0 ? 1 : 0;
                if (lane == 0 || lane == 4) {
#pragma unroll
                    for (int gq = 0; gq < 2; ++gq)
                        p.pre_flags[((size_t)bh * (p.qkv_L >> 4) + (pre_l0 >> 4) + gq) * 2] = flag[gq];
                }
                if (flag[0] | flag[1]) {   // (rare) the fp32 values of the flagged head's keys
#pragma unroll
                    for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(p.out + roff4[j] + co) = v[j];
                }
                continue;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float4 v = *reinterpret_cast<const float4*>(stage + (orow + 8 * j) * LDO + oc4);
            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            v = activate(p.act, v);
            if (m0 + orow + 8 * j < p.M) *reinterpret_cast<float4*>(p.out + roff4[j] + co) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Tail of a 64-channel attention block with a feed-forward part, one kernel for three chained 64x64 GEMMs:
//   a   = o Wo^T + bo + x                       (output projection + residual)
//   h   = GELU(LayerNorm(a) W1^T + b1)          (LayerNorm affine already folded into W1, b1)
//   out = act(a + h W2^T + b2)
// Every step is local to a token row.  A wave owns 32 rows; with the transposed product of k_gemm_k64 the
// accumulator layout of one GEMM (lane (row, h) holds columns 8u + 4h .. 8u + 4h + 3, u = 0..7) is exactly the
// operand-fragment layout of the next one, so the chain runs in registers: no intermediate leaves the wave, the
// LayerNorm statistics come from the two lanes that share a row, and HBM sees o and x once and out once.
// The three weight matrices sit in LDS ([64][68] each); the result goes through a wave-private LDS tile so that
// rows are stored as contiguous 256-byte runs.
// out = a W^T for a 64 x 64 matrix in LDS (HALF: its split form, with Wg the fp32 matrix in global memory for a wave
// whose rows are out of the split's range, and `scale` undoing the weights' power-of-two scale).
template <bool HALF>
__device__ __forceinline__ void gemm64_frag(const float* __restrict__ Ws, const float* __restrict__ Wg, float scale,
                                            const float4 (&a)[8], float4 (&out)[8], int fr, int fh) {
    constexpr int LDK = 68;
    f16x8 ah[HALF ? 4 : 1], al[HALF ? 4 : 1];
    bool split_ok = false;
    if constexpr (HALF) split_ok = split_frag64(a, ah, al);
    const float sc = (HALF && split_ok) ? scale : 1.0f;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        f32x16 acc;
        if constexpr (HALF) {
            if (split_ok) acc = tile64_half(Ws + (nt * 32 + fr) * LDK + fh, ah, al);
            else acc = tile64_f32(Wg + (size_t)(nt * 32 + fr) * 64 + fh, a);
        } else {
            acc = tile64_f32(Ws + (nt * 32 + fr) * LDK + fh, a);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            out[4 * nt + q] = make_float4(acc[4 * q] * sc, acc[4 * q + 1] * sc, acc[4 * q + 2] * sc, acc[4 * q + 3] * sc);
    }
}

// FF: the feed-forward part is present (else the chain is the output projection alone).
// CT: the result is not stored but fed, still in registers, to the transposed convolution that consumes it
//     (kernel 2, stride 2, 64 -> 64 channels = a 64 -> 256 GEMM whose column block q = (dy, dx) lands on output
//     pixel (2y + dy, 2x + dx)): up[b][2y+dy][2x+dx][co] = sum_ci v[b][y][x][ci] * Wt[q*64 + co][ci] + bt[q*64 + co].
// HALF: every product of the chain as a split-operand fp16 product (k_gemm_k64 HALF): `sp` holds the matrices split at pack
// time in fragment order and their scales; each stage checks the range of its own operand rows.
template <bool FF, bool CT, bool HALF>
__global__ __launch_bounds__(512) void k_attn_tail64(const float* __restrict__ O, const float* __restrict__ X,
                                                     const float* __restrict__ Wo, const float* __restrict__ bo,
                                                     const float* __restrict__ W1, const float* __restrict__ b1,
                                                     const float* __restrict__ W2, const float* __restrict__ b2,
                                                     const float* __restrict__ Wt, const float* __restrict__ bt,
                                                     float* __restrict__ out, int M, int act, int ct_H, int ct_W,
                                                     float* __restrict__ part, const Tail64Split sp, int ct_stats_only) {
    constexpr int LDK = 68;
    constexpr int NW = FF ? 3 : 1;             // 64x64 matrices of the chain
    constexpr int WROWS = NW * 64 + (CT ? 256 : 0);
    constexpr int LDO = CT ? 36 : LDK;         // row pitch of the wave's output tile (32 or 64 columns wide)
    extern __shared__ __attribute__((aligned(16))) float tail_s[];  // [WROWS][LDK] weights | [WROWS] biases | output tiles
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, waves = blockDim.x >> 6;
    const int fr = lane & 31, fh = (lane >> 5) * 4;
    float* Ws = tail_s;
    float* bias_s = tail_s + WROWS * LDK;
    float* stage = bias_s + WROWS + wave * 32 * LDO;
    const int m0 = (blockIdx.x * waves + wave) * 32;
    const int arow = min(m0 + fr, M - 1);
    // this wave's rows of o and x in fragment order: every load is in flight before the weights are staged
    float4 a[8], xr[8];
    {
        const float* so = O + (size_t)arow * 64 + fh;
        const float* sx = X + (size_t)arow * 64 + fh;
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) a[kb] = *reinterpret_cast<const float4*>(so + kb * 8);
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) xr[kb] = *reinterpret_cast<const float4*>(sx + kb * 8);
    }
    // the weight matrices -> LDS, six 16-byte loads per thread in flight at a time
    {
        constexpr int NB = 6;
        constexpr int total = WROWS * 16;
        for (int q0 = tid; q0 < total; q0 += NB * (int)blockDim.x) {
            float4 w[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int q = min(q0 + j * (int)blockDim.x, total - 1);
                const float* src;
                int r = q;
                if (q < 1024) src = HALF ? sp.wo : Wo;
                else if (FF && q < 2048) { src = HALF ? sp.w1 : W1; r = q - 1024; }
                else if (FF && q < 3072) { src = HALF ? sp.w2 : W2; r = q - 2048; }
                else { src = HALF ? sp.wt : Wt; r = q - NW * 1024; }
                w[j] = *reinterpret_cast<const float4*>(src + (size_t)r * 4);
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int q = q0 + j * (int)blockDim.x;
                if (q < total) *reinterpret_cast<float4*>(Ws + (q >> 4) * LDK + (q & 15) * 4) = w[j];
            }
        }
        if (tid < 64) {
            bias_s[tid] = bo[tid];
            if (FF) {
                bias_s[64 + tid] = b1[tid];
                bias_s[128 + tid] = b2[tid];
            }
        }
        if (CT && tid < 256) bias_s[NW * 64 + tid] = bt[tid];
    }
    __syncthreads();
    if (m0 >= M) return;

    float4 att[8];
    gemm64_frag<HALF>(Ws, Wo, sp.so, a, att, fr, fh);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const float4 bv = *reinterpret_cast<const float4*>(bias_s + 8 * u + fh);
        att[u].x += bv.x + xr[u].x; att[u].y += bv.y + xr[u].y; att[u].z += bv.z + xr[u].z; att[u].w += bv.w + xr[u].w;
    }
    if (FF) {
        // LayerNorm over the row's 64 values (two-pass, lanes i and i + 32 hold its two halves)
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) s += (att[u].x + att[u].y) + (att[u].z + att[u].w);
        s += __shfl_xor(s, 32);
        const float mean = s * (1.0f / 64.0f);
        float q2 = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float dx = att[u].x - mean, dy = att[u].y - mean, dz = att[u].z - mean, dw = att[u].w - mean;
            q2 += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
        q2 += __shfl_xor(q2, 32);
        const float rstd = 1.0f / sqrtf(q2 * (1.0f / 64.0f) + 1e-5f);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            a[u] = make_float4((att[u].x - mean) * rstd, (att[u].y - mean) * rstd, (att[u].z - mean) * rstd,
                               (att[u].w - mean) * rstd);
        float4 hid[8];
        gemm64_frag<HALF>(Ws + 64 * LDK, W1, sp.s1, a, hid, fr, fh);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float4 bv = *reinterpret_cast<const float4*>(bias_s + 64 + 8 * u + fh);
            hid[u] = make_float4(gelu(hid[u].x + bv.x), gelu(hid[u].y + bv.y), gelu(hid[u].z + bv.z), gelu(hid[u].w + bv.w));
        }
        gemm64_frag<HALF>(Ws + 2 * 64 * LDK, W2, sp.s2, hid, a, fr, fh);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float4 bv = *reinterpret_cast<const float4*>(bias_s + 128 + 8 * u + fh);
            att[u] = make_float4(a[u].x + bv.x + att[u].x, a[u].y + bv.y + att[u].y, a[u].z + bv.z + att[u].z,
                                 a[u].w + bv.w + att[u].w);
        }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) att[u] = activate(act, att[u]);

    if (!CT) {
        // wave-private tile -> memory: 16 consecutive lanes store one whole 256-byte row
#pragma unroll
        for (int u = 0; u < 8; ++u) *reinterpret_cast<float4*>(stage + fr * LDO + 8 * u + fh) = att[u];
        const int orow = lane >> 4, oc4 = (lane & 15) * 4;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = orow + 4 * j;
            if (m0 + row < M)
                *reinterpret_cast<float4*>(out + (size_t)(m0 + row) * 64 + oc4) =
                    *reinterpret_cast<const float4*>(stage + row * LDO + oc4);
        }
        return;
    }
    // ---- transposed convolution: 8 column blocks of 32; block nt = (tap q = nt / 2, channels 32*(nt & 1) ..)
    const int orow = lane >> 3, oc4 = (lane & 7) * 4;  // 8 consecutive lanes store 128 contiguous bytes of one row
    unsigned long long rbase[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = min(m0 + orow + 8 * j, M - 1);
        const int hw = ct_H * ct_W;
        const int bimg = m / hw, r = m - bimg * hw, y = r / ct_W, x = r - y * ct_W;
        rbase[j] = (((unsigned long long)bimg * 2 * ct_H + 2 * y) * 2 * ct_W + 2 * x) * 64;
    }
    const float* Wts = Ws + NW * 64 * LDK;
    const float* bts = bias_s + NW * 64;
    f16x8 th[HALF ? 4 : 1], tl[HALF ? 4 : 1];
    bool ct_ok = false;
    if constexpr (HALF) ct_ok = split_frag64(att, th, tl);
    const float ct_scale = (HALF && ct_ok) ? sp.st : 1.0f;
    for (int nt = 0; nt < 8; ++nt) {
        f32x16 acc;
        if constexpr (HALF) {
            if (ct_ok) acc = tile64_half(Wts + (nt * 32 + fr) * LDK + fh, th, tl);
            else acc = tile64_f32(Wt + (size_t)(nt * 32 + fr) * 64 + fh, att);
        } else {
            acc = tile64_f32(Wts + (nt * 32 + fr) * LDK + fh, att);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(stage + fr * LDO + 8 * g + fh) =
                make_float4(acc[4 * g] * ct_scale, acc[4 * g + 1] * ct_scale, acc[4 * g + 2] * ct_scale, acc[4 * g + 3] * ct_scale);
        const float4 bv = *reinterpret_cast<const float4*>(bts + nt * 32 + oc4);
        const int q = nt >> 1;
        if (part) {
            // InstanceNorm statistics of the transposed convolution's output, for its consumer: this tile is one slice
            // (32 pixels of one image, tap q) of 32 channels; its mean and centred sum of squares (two passes over
            // the tile, lanes l and l + 32 share a column) go to part[image][slice][{mean, m2}][64].
            const int col = lane & 31, r0 = (lane >> 5) * 16;
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += stage[(r0 + r) * LDO + col];
            sum += __shfl_xor(sum, 32);
            const float mu = sum * (1.0f / 32.0f);
            float m2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float d = stage[(r0 + r) * LDO + col] - mu;
                m2 += d * d;
            }
            m2 += __shfl_xor(m2, 32);
            if (lane < 32) {
                const int hw = ct_H * ct_W;
                const int bimg = m0 / hw;
                const int slice = ((m0 - bimg * hw) >> 5) * 4 + q;
                const int S = (hw >> 5) * 4;
                float* dst = part + ((size_t)bimg * S + slice) * 128 + (nt & 1) * 32 + col;
                dst[0] = mu + bts[nt * 32 + col];
                dst[64] = m2;
            }
        }
        const unsigned long long co = ((unsigned long long)(q >> 1) * 2 * ct_W + (q & 1)) * 64 + (nt & 1) * 32 + oc4;
        if (ct_stats_only) continue;  // the consumer rebuilds this tile from the block's output (below) and the statistics
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float4 v = *reinterpret_cast<const float4*>(stage + (orow + 8 * j) * LDO + oc4);
            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            if (m0 + orow + 8 * j < M) *reinterpret_cast<float4*>(out + rbase[j] + co) = v;
        }
    }
    if (ct_stats_only) {
        // the block's own output rows [M][64] instead of the transposed convolution's: two passes of 32 columns through
        // the wave's tile, 8 consecutive lanes store 128 contiguous bytes of a row
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int u = 0; u < 4; ++u) *reinterpret_cast<float4*>(stage + fr * LDO + 8 * u + fh) = att[4 * half + u];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (m0 + orow + 8 * j < M)
                    *reinterpret_cast<float4*>(out + (size_t)(m0 + orow + 8 * j) * 64 + 32 * half + oc4) =
                        *reinterpret_cast<const float4*>(stage + (orow + 8 * j) * LDO + oc4);
        }
    }
}

template <bool FF, bool CT, bool HALF>
static hipError_t launch_attn_tail64_t(const float* o, const float* x, const float* Wo, const float* bo, const float* W1,
                                       const float* b1, const float* W2, const float* b2, const float* Wt, const float* bt,
                                       float* out, int M, int act, int ct_H, int ct_W, float* part, const Tail64Split& sp,
                                       int ct_stats_only, hipStream_t stream) {
    const int tasks = (M + 31) / 32;
    if (part && (!CT || M % 32 || (ct_H * ct_W) % 32)) return hipErrorInvalidValue;  // a slice = 32 rows of one image
    if (ct_stats_only && !part) return hipErrorInvalidValue;
    int waves = (FF && CT) ? 4 : 8;  // LDS budget; 4 also when there are few row tiles (fewer only stretch the staging)
    while (waves > 4 && tasks < 128 * waves) waves /= 2;
    const int wrows = (FF ? 3 : 1) * 64 + (CT ? 256 : 0);
    const size_t lds = (size_t)(wrows * 68 + wrows + waves * 32 * (CT ? 36 : 68)) * sizeof(float);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_tail64<FF, CT, HALF>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_attn_tail64<FF, CT, HALF>), dim3((tasks + waves - 1) / waves), dim3(64 * waves), lds, stream, o, x,
                       Wo, bo, W1, b1, W2, b2, Wt, bt, out, M, act, ct_H, ct_W, part, sp, ct_stats_only);
    return hipGetLastError();
}

hipError_t launch_attn_tail64(const float* o, const float* x, const float* Wo, const float* bo, const float* W1,
                              const float* b1, const float* W2, const float* b2, const float* Wt, const float* bt,
                              float* out, int M, int act, int ct_H, int ct_W, float* part, hipStream_t stream,
                              const Tail64Split* split, int ct_stats_only) {
    // the split-operand form needs the split copy of every matrix of the chain
    const bool half = split && !g_igemm_fp32_products && split->wo && (!W1 || (split->w1 && split->w2)) && (!Wt || split->wt);
    const Tail64Split sp = half ? *split : Tail64Split{};
#define DANRA_TAIL64(FF_, CT_)                                                                                              \
    (half ? launch_attn_tail64_t<FF_, CT_, true>(o, x, Wo, bo, W1, b1, W2, b2, Wt, bt, out, M, act, ct_H, ct_W, part, sp,       \
                                                 ct_stats_only, stream)                                                       \
          : launch_attn_tail64_t<FF_, CT_, false>(o, x, Wo, bo, W1, b1, W2, b2, Wt, bt, out, M, act, ct_H, ct_W, part, sp,      \
                                                  ct_stats_only, stream))
    if (W1 && Wt) return DANRA_TAIL64(true, true);
    if (W1) { part = nullptr; ct_stats_only = 0; return DANRA_TAIL64(true, false); }
    if (Wt) return DANRA_TAIL64(false, true);
#undef DANRA_TAIL64
    return hipErrorInvalidValue;  // the plain output projection goes through launch_igemm
}

// ------------------------------------------------------------------------------------------------
// The x-plane share of the 8x8 stride-2 stem (one input plane, 64 output channels, K = 64 taps):
//   out[b][oy][ox][n] = sum_{kh,kw} x[b][2oy-3+kh][2ox-3+kw] * W[n][kh*8+kw] + res[b][oy][ox][n] + tbias[b][n].
// A workgroup owns an 8x8 patch of output pixels of one image.  Its 22x22 input window (zero padded) and the
// 64x64 weight slice are staged in LDS once; the MFMA operand of a pixel is then read straight from the window
// (lane (pixel, h) needs taps kw = 4h..4h+3 of kernel row u: four consecutive floats), so there is no
// per-tap address arithmetic, bounds test or staging loop at all.  Same fragment convention as k_igemm.
__global__ __launch_bounds__(256) void k_stem_x(const float* __restrict__ x, const float* __restrict__ Wt, int ldw,
                                                const float* __restrict__ res, const float* __restrict__ tbias,
                                                int tb_stride, float* __restrict__ out, int H, int W) {
    constexpr int PW = 24;   // padded window row (floats)
    constexpr int LDK = 68;
    __shared__ __attribute__((aligned(16))) float win[22 * PW];
    __shared__ __attribute__((aligned(16))) float Ws[64 * LDK];   // reused as the output tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Ho = H / 2, Wo = W / 2;
    const int tiles_x = Wo / 8;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int b = blockIdx.y;
    // ---- weights: 64 rows x 64 taps
    {
        const int r = tid >> 4, c4 = (tid & 15) * 4;
        float4 w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = *reinterpret_cast<const float4*>(Wt + (size_t)(r + 16 * j) * ldw + c4);
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(Ws + (r + 16 * j) * LDK + c4) = w[j];
    }
    // ---- input window: rows 16ty-3 .. 16ty+18, columns 16tx-3 .. 16tx+18
    const float* xb = x + (size_t)b * H * W;
    for (int i = tid; i < 22 * 22; i += 256) {
        const int r = i / 22, c = i - r * 22;
        const int iy = 16 * ty - 3 + r, ix = 16 * tx - 3 + c;
        const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        const float v = xb[ok ? iy * W + ix : 0];
        win[r * PW + c] = ok ? v : 0.f;
    }
    __syncthreads();
    const int fr = lane & 31, fh = (lane >> 5) * 4;
    const int row0 = (wave >> 1) * 32, col0 = (wave & 1) * 32;
    const int pix = row0 + fr;                 // pixel of the 8x8 patch: (pix >> 3, pix & 7)
    const float* ap = win + (2 * (pix >> 3)) * PW + 2 * (pix & 7) + fh;
    const float* bp = Ws + (col0 + fr) * LDK + fh;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {   // kernel row u: k = 8u + 4h + {0..3}
        const float2 a01 = *reinterpret_cast<const float2*>(ap + u * PW);
        const float2 a23 = *reinterpret_cast<const float2*>(ap + u * PW + 2);
        const float4 bw = *reinterpret_cast<const float4*>(bp + u * 8);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bw.x, a01.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bw.y, a01.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bw.z, a23.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bw.w, a23.y, acc, 0, 0, 0);
    }
    // ---- epilogue through LDS: acc[r] = out[pixel row0 + (lane & 31)][channel col0 + (r & 3) + 8*(r >> 2) + 4*(lane >> 5)]
    __syncthreads();
    float* tile = Ws;
#pragma unroll
    for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(tile + pix * LDK + col0 + 8 * g + fh) =
            make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
    __syncthreads();
    const int oc4 = (tid & 15) * 4;
    float4 tb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tbias) tb = *reinterpret_cast<const float4*>(tbias + (size_t)b * tb_stride + oc4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = (tid >> 4) + 16 * j;
        const size_t o = (((size_t)b * Ho + 8 * ty + (p >> 3)) * Wo + 8 * tx + (p & 7)) * 64 + oc4;
        float4 v = *reinterpret_cast<const float4*>(tile + p * LDK + oc4);
        if (res) {
            const float4 rv = *reinterpret_cast<const float4*>(res + o);
            v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
        }
        v.x += tb.x; v.y += tb.y; v.z += tb.z; v.w += tb.w;
        *reinterpret_cast<float4*>(out + o) = v;
    }
}

hipError_t launch_stem_x(const float* x, const float* Wt, int ldw, const float* res, const float* tbias, int tb_stride,
                         float* out, int B, int H, int W, hipStream_t stream) {
    if (H % 16 || W % 16 || ldw % 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_stem_x, dim3((H / 16) * (W / 16), B), dim3(256), 0, stream, x, Wt, ldw, res, tbias, tb_stride, out,
                       H, W);
    return hipGetLastError();
}


// Whether launch_igemm runs these arguments in the split-operand fp16 form (the row-stationary K = 64 kernel never does).
// Whether launch_igemm hands these arguments to the row-stationary K = 64 kernel.
static bool takes_k64_path(const IGemmArgs& a) {
    const int ks = a.ksplit > 1 ? a.ksplit : 1;
    return a.a_mode == A_GEMM && a.K == 64 && a.lda % 4 == 0 && ks == 1 && !a.force_bm && a.N <= 256 && a.N >= 128 &&
           a.M >= 32768 && !a.tbias && !a.res && !a.out2;
}

bool igemm_uses_split(const IGemmArgs& a) {
    const int ks = a.ksplit > 1 ? a.ksplit : 1;
    if (takes_k64_path(a)) return a.half && a.Wh && !g_igemm_fp32_products && !a.ldw;
    return a.half && a.Wh && !g_igemm_fp32_products && a.a_mode != A_CONV1 && a.K % 64 == 0 && a.K >= 128 &&
           (a.a_mode != A_CONV || a.Cin % 64 == 0) && (ks == 1 || a.chunks_per_split % 2 == 0);
}

bool gemm_k64_presplits(const IGemmArgs& a) {
    return takes_k64_path(a) && igemm_uses_split(a) && a.st_mode == ST_QKV && a.qkv_hd == 16 && a.qkv_C == 64 && a.N == 192 &&
           a.qkv_heads == 4 && a.M % 32 == 0 && a.qkv_L % 64 == 0 && a.act == ACT_NONE;
}

static int log2_if_pow2(long long d) {
    if (d <= 0 || (d & (d - 1))) return -1;
    int s = 0;
    while ((1ll << s) < d) ++s;
    return s;
}

hipError_t launch_igemm(const IGemmArgs& args, hipStream_t stream) {
    IGemmArgs a = args;
    a.sh_rpi = log2_if_pow2(a.rows_per_img);
    a.sh_batch = log2_if_pow2(a.batch);
    a.sh_howo = log2_if_pow2((long long)a.Ho * a.Wo);
    a.sh_wo = log2_if_pow2(a.Wo);
    a.sh_qkv_l = log2_if_pow2(a.qkv_L);
    a.sh_ct_hw = log2_if_pow2((long long)a.ct_H * a.ct_W);
    a.sh_ct_w = log2_if_pow2(a.ct_W);
    const int ks = a.ksplit > 1 ? a.ksplit : 1;
    // every store mode keeps runs of four output columns contiguous when its inner extent is a multiple of 4
    const bool vec4 = a.ldc % 4 == 0 && (a.st_mode != ST_QKV || a.qkv_hd % 4 == 0) && (a.st_mode != ST_CONVT || a.ct_C % 4 == 0) &&
                      (!a.res || a.ldr % 4 == 0) && (!a.tbias || a.tb_stride % 4 == 0);
    if (!vec4) return hipErrorInvalidValue;  // the epilogues store 16 bytes at a time
    if (a.a_mode == A_CONV && !a.zeros) return hipErrorInvalidValue;
    if ((long long)a.M * a.N >= (1ll << 31)) return hipErrorInvalidValue;  // 32-bit element indices within an output or slab
    if (a.out2 && (a.n_split <= 0 || a.n_split % BN || a.n_split >= a.N || a.st_mode != ST_ROW)) return hipErrorInvalidValue;
    if ((a.pre_ks || a.pre_vs || a.pre_flags) && !(a.pre_ks && a.pre_vs && a.pre_flags && gemm_k64_presplits(a)))
        return hipErrorInvalidValue;
    if (takes_k64_path(a)) {
        // one 32-row tile per wave; fewer waves per workgroup when there are too few tiles for 256 CUs
        const int nsplit = (a.N % 64 == 0 && a.N >= 128) ? 2 : 1;
        const int tiles = (a.M + 31) / 32 * nsplit;
        IGemmArgs b = a;
        b.force_kw = nsplit;  // reused as the column split of a wave task
        int waves = 1;  // as many waves share one copy of the weights as keeps every CU busy
        while (waves < 16 && tiles > 256 * waves) waves *= 2;
        const size_t lds = (size_t)a.N * 68 * sizeof(float) + (size_t)(a.N / 4) * sizeof(unsigned long long) +
                           (size_t)a.N * sizeof(float) + (size_t)waves * 32 * 36 * sizeof(float);
        const int nw4 = (a.N * 16 + 64 * waves - 1) / (64 * waves);
        const dim3 grid((tiles + waves - 1) / waves), block(64 * waves);
#define DANRA_LAUNCH_K64(NW4_, H_)                                                                          \
    do {                                                                                                    \
        if (lds > 48 * 1024) {                                                                              \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_k64<NW4_, H_>),         \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);       \
            if (e != hipSuccess) return e;                                                                  \
        }                                                                                                   \
        DANRA_LAUNCH_KERNEL((k_gemm_k64<NW4_, H_>), grid, block, lds, stream, b);                            \
    } while (0)
        // (for K = 64 `Wh` is in the fragment order of this kernel: launch_split_weights with frag64)
        const bool half64 = igemm_uses_split(a);
        if (half64) {
            if (nw4 <= 4) DANRA_LAUNCH_K64(4, true);
            else DANRA_LAUNCH_K64(16, true);
        } else if (nw4 <= 4) DANRA_LAUNCH_K64(4, false);
        else DANRA_LAUNCH_K64(16, false);
#undef DANRA_LAUNCH_K64
        return hipGetLastError();
    }
    int bm = a.force_bm, kw = a.force_kw;
    if (!bm) {
        // Enough wave-level parallelism for 256 CUs x 4 SIMDs is the first concern; operand reuse second.
        // Measured on MI355X at batch 64 (tune_igemm): the 64-row tile beats the 128-row one on every shape of
        // the network, 8 waves per workgroup win when there are many tiles, 16 when there are few.
        const long t64 = (long)((a.M + 63) / 64) * (a.N / BN) * ks;
        bm = 64;
        kw = (t64 >= 512) ? 2 : 4;
    }
    // the split-operand fp16 form walks K in 64-deep chunks: channel counts (and the host's 32-deep split-K chunk
    // counts) must be multiples of that
    const bool half = igemm_uses_split(a);
    if (kw == 4 && (a.K % 64 || (ks > 1 && a.chunks_per_split % 2))) kw = 2;
    if (kw == 12 && !half) kw = 4;
    if (kw == 4 && half && bm == 64 && !a.force_kw) kw = 12;  // 16 waves that all stage and multiply -> 8 staging + 4 multiplying waves
#define DANRA_LAUNCH_IGEMM(BM_, KW_, H_)                                                                  \
    do {                                                                                                  \
        dim3 grid((a.M + BM_ - 1) / BM_, a.N / BN, ks);                                                   \
        if (a.a_mode == A_GEMM && a.ln) DANRA_LAUNCH_KERNEL((k_igemm<BM_, KW_, A_GEMM_LN, H_>), grid, dim3(256 * KW_), 0, stream, a); \
        else if (a.a_mode == A_GEMM) DANRA_LAUNCH_KERNEL((k_igemm<BM_, KW_, A_GEMM, H_>), grid, dim3(256 * KW_), 0, stream, a);   \
        else if (a.a_mode == A_CONV && a.nmean) DANRA_LAUNCH_KERNEL((k_igemm<BM_, KW_, A_CONV_IN, H_>), grid, dim3(256 * KW_), 0, stream, a); \
        else if (a.a_mode == A_CONV) DANRA_LAUNCH_KERNEL((k_igemm<BM_, KW_, A_CONV, H_>), grid, dim3(256 * KW_), 0, stream, a);   \
        else if (!H_) DANRA_LAUNCH_KERNEL((k_igemm<BM_, KW_, A_CONV1, false>), grid, dim3(256 * KW_), 0, stream, a); \
    } while (0)
    if (half && kw == 12 && bm == 64) {  // wave roles: 4 multiplying + 8 staging waves
        dim3 grid((a.M + 63) / 64, a.N / BN, ks);
        if (a.a_mode == A_GEMM && a.ln) DANRA_LAUNCH_KERNEL((k_igemm<64, 4, A_GEMM_LN, true, true>), grid, dim3(768), 0, stream, a);
        else if (a.a_mode == A_GEMM) DANRA_LAUNCH_KERNEL((k_igemm<64, 4, A_GEMM, true, true>), grid, dim3(768), 0, stream, a);
        else if (a.nmean) DANRA_LAUNCH_KERNEL((k_igemm<64, 4, A_CONV_IN, true, true>), grid, dim3(768), 0, stream, a);
        else DANRA_LAUNCH_KERNEL((k_igemm<64, 4, A_CONV, true, true>), grid, dim3(768), 0, stream, a);
    } else if (half) {
        if (bm == 128) DANRA_LAUNCH_IGEMM(128, 1, true);
        else if (kw == 4) DANRA_LAUNCH_IGEMM(64, 4, true);
        else if (kw == 2) DANRA_LAUNCH_IGEMM(64, 2, true);
        else DANRA_LAUNCH_IGEMM(64, 1, true);
    } else if (bm == 128) DANRA_LAUNCH_IGEMM(128, 1, false);
    else if (kw == 4) DANRA_LAUNCH_IGEMM(64, 4, false);
    else if (kw == 2) DANRA_LAUNCH_IGEMM(64, 2, false);
    else DANRA_LAUNCH_IGEMM(64, 1, false);
#undef DANRA_LAUNCH_IGEMM
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (ks > 1 && !a.keep_slabs) {
        const size_t total4 = (size_t)a.M * a.N / 4;
        int blocks = (int)((total4 + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        DANRA_LAUNCH_KERNEL(k_splitk_epilogue, dim3(blocks), dim3(256), 0, stream, a);
        e = hipGetLastError();
    }
    return e;
}

}  // namespace danra
