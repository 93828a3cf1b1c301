// The 8x8 stride-2 padding-3 convolution of the encoder (Encoder.conv2 behind the first attention block: 64 -> 64
// channels, 32x32 -> 16x16 at a 64x64 input) on NHWC fp32 maps, as split-operand fp16 matrix products
// (v_mfma_f32_32x32x16_f16, fp32 accumulation, W x ~ Wh xh + Wh xl + Wl xh: see split_f16.h).  The scheme of
// kernels_conv_win.hip carried over to a strided window:
//
//  * A workgroup owns 64 output pixels (4 rows x 16 columns) x 64 output channels.  The input WINDOW behind them
//    (14 x 38 pixels, zeros where it leaves the map) is staged in LDS once per 64 input channels, split in fp16 halves,
//    and serves all 64 taps: an activation is fetched once per workgroup, not once per tap that touches it (16 of the
//    64 on a stride-2 grid), and every tap's operand lies at a fixed offset behind the lane's own cell - the loop holds
//    no address arithmetic.  Window columns are stored de-interleaved (even columns, then odd ones): the lanes of a
//    fragment read hold consecutive output columns, i.e. every second input column of one parity, and find them side by
//    side, so the reads are free of bank conflicts like those of the stride-1 kernel.
//  * Weights stream from memory in fragment order straight into the matrix instruction's registers
//    (launch_pack_conv_frag with 64 taps), each by exactly one wave: wave w owns input channels 16 (w & 3) ..+15 of a
//    64-channel block and output channels 32 (w >> 2) ..+31 of the tile.
//  * The four partial tiles of a channel half are added in slice order through LDS: no split-K slabs, no reduction
//    kernel, bitwise reproducible results.
#include "conv_window.h"

namespace danra {

using namespace convwin;

namespace {

constexpr int S2_TR = 4, S2_TC = 16;                    // output rows, columns of a tile
constexpr int S2_ROWS = 2 * S2_TR + 6;                  // window rows
constexpr int S2_COLS = 2 * S2_TC + 6;                  // window columns
constexpr int S2_HALF = S2_COLS / 2;                    // cells of one column parity in a window row
constexpr int S2_PS = pixel_bytes(1);
constexpr int S2_WINDOW = S2_ROWS * S2_COLS * S2_PS;    // bytes
constexpr int S2_RED = 4 * 64 * (64 + 4) * 4;           // four partial tiles of 64 x 64 floats (padded rows)
constexpr int S2_LDS = S2_WINDOW > S2_RED ? S2_WINDOW : S2_RED;

}  // namespace

__global__ __launch_bounds__(512) void k_conv8s2(const ConvWinArgs p) {
    constexpr int TR = S2_TR, TC = S2_TC, TP = 64, TN = 64, TAPS = 64;
    constexpr int RS = S2_COLS, HALF = S2_HALF, PS = S2_PS;
    constexpr int NSTEPS = 64;            // k-steps of a wave per 64-channel block: the taps
    // k-steps of weights in flight per wave (2 KB each).  A step of two waves per SIMD lasts about 400 cycles and a trip
    // to a loaded L2 2000-6000: the ring is as deep as the registers allow
    constexpr int PD = 16;
    constexpr int LDO = TN + 4;
    constexpr int T = 512;
    constexpr int MAXITEMS = (S2_ROWS * S2_COLS * 16 + T - 1) / T;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int flags[3];              // a staged value too large for the split / of normal size / non-zero

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = wave & 3, ct = wave >> 2;   // slice of input channels, 32-channel tile of the outputs
    const int H = p.H, W = p.W, Ho = H >> 1, Wo = W >> 1, Cin = p.Cin, NCB = Cin >> 6;
    const int n0 = blockIdx.y * TN;
    if (tid < 3) flags[tid] = 0;          // (ordered before the vote by the barriers below)
    WIN_STAMP(0);

    // ---- tile geometry (wave-uniform): the window's first cell is input pixel (iy0, ix0)
    const int tcols = Wo / TC, trows = Ho / TR;
    const int tile = blockIdx.x;
    const int tx = tile % tcols, ty = (tile / tcols) % trows, b = tile / (tcols * trows);
    const int oy0 = ty * TR, ox0 = tx * TC;
    const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;

    // ---- this wave's weight stream: fragments [channel tile][slice s][block][tap][hi | lo][lane]
    // (uniform base + one 32-bit lane offset: the loads take the scalar-base form and need no address registers of their own)
    const char* wsrc = static_cast<const char*>(p.wf) + ((size_t)(((n0 >> 5) + ct) * 4 + s) * NCB * TAPS * 2) * 1024;
    const unsigned lane16 = lane * 16;
    auto wload = [&](f16x8 (&dst)[2], int g) {      // step g = block * 64 + tap
        const int cb = min(g >> 6, NCB - 1), tap = g & 63;
        const char* src = wsrc + (size_t)((cb * TAPS + tap) * 2) * 1024;
        dst[0] = *reinterpret_cast<const f16x8*>(src + lane16);
        dst[1] = *reinterpret_cast<const f16x8*>(src + 1024 + lane16);
    };
    f16x8 wq[PD][2];
#pragma unroll
    for (int d = 0; d < PD / 2; ++d) wload(wq[d], d);           // (the rest once the staging registers are free)

    // ---- staging: item i = tid + j T is (window cell i / 16 in row-major order, channels 4 (i % 16) ..+3 of the block);
    // cells outside the map receive zeros
    const int c4 = (tid & 15) * 4;
    v4f raw[MAXITEMS];
    float amax = 0.f;
    auto cell_of = [&](int j, int& r, int& c) -> bool {          // false: no such cell
        const int u = (tid >> 4) + j * (T / 16);
        r = u / RS; c = u - r * RS;
        return u < S2_ROWS * S2_COLS;
    };
    auto load_window = [&](int cb) {
#pragma unroll
        for (int j = 0; j < MAXITEMS; ++j) {
            int r, c;
            const bool cell = cell_of(j, r, c);
            const int iy = iy0 + r, ix = ix0 + c;
            const bool in = cell && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            const int at = __mul24(__mul24(b * H + iy, W) + ix, Cin) + cb * 64 + c4;
            raw[j] = *reinterpret_cast<const v4f*>(p.x + (unsigned)(in ? at : 0));
            if (!in) raw[j] = v4f{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_window = [&]() {
#pragma unroll
        for (int j = 0; j < MAXITEMS; ++j) {
            int r, c;
            if (!cell_of(j, r, c)) continue;
            const v4f v = raw[j];
            amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
            f16x4 hi, lo;
            split4w(v, hi, lo);
            unsigned char* dst = smem + (r * RS + (c & 1) * HALF + (c >> 1)) * PS + c4 * 2;
            *reinterpret_cast<f16x4*>(dst) = hi;
            *reinterpret_cast<f16x4*>(dst + 128) = lo;
        }
    };
    WIN_STAMP(1);
    load_window(0);
    WIN_STAMP(2);

    // ---- lane geometry of the B operand (activations): lane (r, h) of pixel tile t reads 8 halves of its pixel's
    // neighbour under every tap.  Tap (kh, kw) of output pixel (oyl, oxl) of the tile is window cell (2 oyl + kh, column
    // 2 oxl + kw) = parity kw & 1, half-column oxl + (kw >> 1): a fixed offset behind the lane's own cell.
    const int fr = lane & 31, fh = lane >> 5;
    int lb[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int q = 32 * t + fr, oyl = q / TC, oxl = q % TC;
        lb[t] = (2 * oyl * RS + oxl) * PS + s * 32 + fh * 16;
    }

    f32x16 acc[2];              // [pixel tile]
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    store_window();
    WIN_STAMP(3);
#pragma unroll
    for (int d = PD / 2; d < PD; ++d) wload(wq[d], d);
    for (int cb = 0; cb < NCB; ++cb) {
        if (cb > 0) {
            lds_barrier();                // everybody has read the previous block's window
            load_window(cb);
            store_window();
        }
        lds_barrier();
        WIN_STAMP(4);
#pragma unroll
        for (int i = 0; i < NSTEPS; ++i) {
            if (i == 16) WIN_STAMP(9);
            if (i == 32) WIN_STAMP(10);
            // (keeps each step's reads, products and weight load together: left to itself the scheduler spreads the 64
            // unrolled steps over one another and the loop takes half as long again)
            __builtin_amdgcn_sched_barrier(0);
            const int kh = i >> 3, kw = i & 7;
            const int coff = (kh * RS + (kw & 1) * HALF + (kw >> 1)) * PS;
            f16x8 ah[2], al[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                ah[t] = *reinterpret_cast<const f16x8*>(smem + lb[t] + coff);
                al[t] = *reinterpret_cast<const f16x8*>(smem + lb[t] + coff + 128);
            }
            const f16x8 wh = wq[i % PD][0], wl = wq[i % PD][1];
            // (term by term over the tiles: consecutive instructions write different accumulators; per accumulator lo-hi, hi-lo, hi-hi)
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, ah[t], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, al[t], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, ah[t], acc[t], 0, 0, 0);
            wload(wq[i % PD], cb * NSTEPS + i + PD);           // (past the last block: its last steps again, never used)
        }
    }
    WIN_STAMP(5);
    lds_barrier();                        // every window read has retired: the partial tiles reuse its LDS

    // ---- range vote of the split (every staged value is in `amax` by now)
    {
        const bool too_big = !(amax < 32768.f);
        if (__any(too_big) && lane == 0) flags[0] = 1;
        if (__any(amax >= 0.00390625f) && lane == 0) flags[1] = 1;
        if (__any(amax > 0.f) && lane == 0) flags[2] = 1;
    }
    // ---- partial tiles -> LDS: slice s, columns 32 ct ..+31.  The final sum has four terms, in slice order.
    float* red = reinterpret_cast<float*>(smem);
    float* mine = red + s * TP * LDO + 32 * ct;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(mine + (32 * t + fr) * LDO + 8 * g + 4 * fh) =
                make_float4(acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]);
    lds_barrier();
    WIN_STAMP(6);
    const bool redo = flags[0] != 0 || (flags[2] != 0 && flags[1] == 0);

    // ---- epilogue: 16 threads per pixel, 4 channels each, 2 pixels per thread
    const int oc4 = (tid & 15) * 4, n = n0 + oc4;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) bv = *reinterpret_cast<const float4*>(p.bias + n);
    auto finish_store = [&](int m, float4 v) {
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        if (p.res) {
            const float4 rv = *reinterpret_cast<const float4*>(p.res + (size_t)m * p.Cout + n);
            v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
        }
        v.x = act_apply(p.act, v.x); v.y = act_apply(p.act, v.y); v.z = act_apply(p.act, v.z); v.w = act_apply(p.act, v.w);
        if (p.tbias) {
            const float4 tv = *reinterpret_cast<const float4*>(p.tbias + (size_t)b * p.tb_stride + n);
            v.x += tv.x; v.y += tv.y; v.z += tv.z; v.w += tv.w;
        }
        *reinterpret_cast<float4*>(p.out + (size_t)m * p.Cout + n) = v;
    };
    const float scale = p.wscale;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int q = (tid >> 4) + e * 32, oy = oy0 + q / TC, ox = ox0 + q % TC;
        const int m = (b * Ho + oy) * Wo + ox;
        if (!__builtin_amdgcn_readfirstlane(redo ? 1 : 0)) {
            float4 v = *reinterpret_cast<const float4*>(red + q * LDO + oc4);
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float4 o = *reinterpret_cast<const float4*>(red + (w * TP + q) * LDO + oc4);
                v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            }
            v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
            finish_store(m, v);
            continue;
        }
        // a staged value left the range in which the split holds 22 bits (2^15 or more, non-finite, or nothing above
        // 2^-8): this workgroup forms its tile again with fp32 multiply-adds from the fp32 weights.  Rare, and slow.
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        const int K = TAPS * Cin;
        for (int kh = 0; kh < 8; ++kh)
            for (int kw = 0; kw < 8; ++kw) {
                const int iy = 2 * oy + kh - 3, ix = 2 * ox + kw - 3;
                if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) continue;
                const float* xr = p.x + ((size_t)(b * H + iy) * W + ix) * Cin;
                const float* wr = p.w32 + (size_t)n * K + (kh * 8 + kw) * Cin;
                for (int c = 0; c < Cin; ++c) {
                    const float xv = xr[c];
#pragma unroll
                    for (int k = 0; k < 4; ++k) a[k] = fmaf(wr[(size_t)k * K + c], xv, a[k]);
                }
            }
        finish_store(m, make_float4(a[0], a[1], a[2], a[3]));
    }
#if DANRA_WIN_DEBUG
    __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0): the stores have been acknowledged
#endif
    WIN_STAMP(7);
}

namespace {

bool pow2(int v) { return v > 0 && !(v & (v - 1)); }

}  // namespace

bool conv8s2_supported(int B, int H, int W, int Cin, int Cout) {
    if (B <= 0 || !pow2(H) || !pow2(W) || H < 8 || W < 32 || Cin % 64 || Cout % 64 || Cin <= 0 || Cout <= 0) return false;
    const long M = (long)B * H * W;
    return M * (long)(Cin > Cout ? Cin : Cout) < (1l << 31) && M < (1l << 23);
}

hipError_t launch_conv8s2(const ConvWinArgs& a, hipStream_t stream) {
    if (!conv8s2_supported(a.B, a.H, a.W, a.Cin, a.Cout) || !a.x || !a.wf || !a.w32 || !a.out || a.nmean || a.nrstd)
        return hipErrorInvalidValue;
    if (a.tbias && a.tb_stride % 4) return hipErrorInvalidValue;
    const int tiles = a.B * (a.H / 2) * (a.W / 2) / 64;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv8s2), hipFuncAttributeMaxDynamicSharedMemorySize, S2_LDS);
    if (e != hipSuccess) return e;
    DANRA_LAUNCH_KERNEL(k_conv8s2, dim3(tiles, a.Cout / 64), dim3(512), S2_LDS, stream, a);
    return hipGetLastError();
}

}  // namespace danra
