// 3x3 stride-1 convolutions on NHWC fp32 maps as split-operand fp16 matrix products (v_mfma_f32_32x32x16_f16, fp32
// accumulation, W x ~ Wh xh + Wh xl + Wl xh: see split_f16.h), built around what bounds a convolution of this network
// on gfx950: the bytes a CU pulls through its vector memory path.
//
//  * The input WINDOW of a workgroup's pixel tile (its pixels, plus the map row above and below when the tile is part
//    of an image) is staged in LDS once per stage of 64 ... 256 input channels, already split in fp16 halves, as the
//    plain run of NHWC pixels it is in memory, and serves all nine taps at fixed offsets: an activation is fetched once
//    per workgroup instead of once per tap.  There are no padding pixels: a lane whose tap falls outside its image
//    reads a pixel of zeros instead (its nine addresses are formed once, before the loop).
//  * The weights never pass through LDS.  They are packed in FRAGMENT ORDER (launch_pack_conv_frag): the 16 bytes lane
//    l of a wave feeds to the matrix instruction as its A operand for (32 output channels, one tap, 16 input
//    channels) lie at byte 16 l of a contiguous kilobyte, so a wave's weight load is one fully coalesced instruction
//    whose result is the operand, and each weight is fetched by exactly one wave of a workgroup.
//  * The reduction dimension is split over the waves of ONE workgroup (wave w owns input channels 16 (w & 3) ..+15 of
//    every 64-channel block, and with eight waves every second tap); the partial tiles are added in wave order
//    through LDS, so no launch needs split-K slabs or a reduction kernel, and results are bitwise reproducible.
//  * One barrier per 64-channel block (none per tap): the next block's window is in flight while this one is multiplied.
//
// Tiles: 32 NPT output pixels x 32 output channels.  Pixels are consecutive rows of the NHWC tensor (whole map rows of
// one image, or several whole images of a small map); on maps of at most four pixels the rows of a tile are the same
// output pixel of 32 consecutive images ("pixel-major"), which makes the taps that fall outside the map the same for
// the whole tile: they are skipped, weights included (5 of 9 on a 2x2 map).
#include <type_traits>

#include "conv_window.h"

namespace danra {

using namespace convwin;

#if DANRA_WIN_DEBUG
__device__ unsigned long long g_win_stamps[8 * 16];
#endif

// NPT: 32-pixel tiles per workgroup.  NW = 8 waves: wave w owns input channels 16 (w & 3) ..+15 of every 64-channel
// block; the two waves of a slice share its k-steps (cb, tap) alternately (their partial tiles are added before the final
// sum) or, with PSPLIT, each takes half of the pixel tiles for every k-step (no sum between them; both fetch the
// slice's weights, the second fetch from the CU's own cache).  WCAP: most window pixels the staging registers are sized
// for.  CBS: 64-channel blocks staged together (one barrier and one trip to memory per CBS blocks: on the small maps,
// whose tiles do little arithmetic per block, a trip to memory per block is what the loop would wait for).  NORM:
// InstanceNorm of the input applied while the window is staged ((x - mean[b][c]) * rstd[b][c], in-bounds pixels only).
// PIXM: pixel-major tiles (head of the file).
// S2: stride 2 (padding 1) on a map of at most 4 TP pixels: the tile's output pixels are whole images, the window their input
// images as they lie in memory, and a lane's own cell is input pixel (2 oy, 2 ox) of its output pixel - the taps keep their
// offsets.  (Lanes then read every second cell of a row: two of them share a bank, which these launches, bound by their
// set-up and their weights, do not notice.)  With p.out2 the output channels from p.n_split on go to a second tensor with
// their bias only: the 1x1 stride-2 shortcut of a BasicBlock rides along as the centre tap of a 3x3 kernel (the other
// eight taps of those channels are taken to be zero: they are neither fetched nor multiplied).
template <int NPT, int WCAP, int CBS, bool NORM, bool PIXM, bool PSPLIT, bool S2 = false>
__global__ __launch_bounds__(512) void k_conv_win(const ConvWinArgs p) {
    constexpr int NW = 8, T = 64 * NW;
    constexpr int TP = 32 * NPT;          // pixels of the tile
    constexpr int LTP = NPT == 4 ? 7 : NPT == 2 ? 6 : 5;
    constexpr int TN = 32;                // output channels of the tile
    constexpr int TGK = PSPLIT ? 1 : 2;   // wave groups that share the k-steps of a slice
    constexpr int NPTW = PSPLIT ? NPT / 2 : NPT;  // pixel tiles a wave multiplies
    constexpr int NTAPS = PIXM ? 4 : 9;   // taps a tile walks (pixel-major: a 2x2 map, four taps touch it)
    constexpr int KSTEPS = CBS * NTAPS;   // k-steps of a stage, per slice
    constexpr int NSTEPS = (KSTEPS + TGK - 1) / TGK;   // ... per wave (9 taps of ONE block: five for group 0, four for group 1)
    // k-steps of weights a wave keeps in flight (2 KB each; a trip to memory takes 1500-2000 cycles and a step of the
    // small tiles 100-200): the ring is indexed statically, so it divides the steps of a stage unless there is one stage
    constexpr int PD = PIXM ? (NSTEPS >= 8 ? 8 : NSTEPS) : (NPT == 4 && NSTEPS % 6 == 0) ? 6 : (NSTEPS % 9 == 0 ? 9 : 3);
    constexpr int LDO = TN + 4;           // row of a partial tile in LDS (floats)
    constexpr int PS = pixel_bytes(CBS);
    static_assert(!PSPLIT || NPT % 2 == 0, "the pixel tiles are dealt to two wave groups");
    static_assert(!S2 || (!NORM && !PIXM && !PSPLIT), "stride 2: whole-image tiles, no norm on load");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int flags[3];              // a staged value too large for the split / of normal size / non-zero

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = wave & 3, grp = wave >> 2;
    // (Numbering the workgroups so that those of one XCD share a channel tile - and with it the weights in that XCD's L2 -
    // was measured and dropped: the 32 CUs of an XCD then ask the same L2 lines at the same time, 5-10 % slower on the
    // 16x16 ... 4x4 maps, 5 % faster on the 2x2 ones, 1.8 % slower over a whole evaluation.)
    // (... except where a launch has few pixel tiles and many channel tiles - the 2x2 maps: there the channel tile is the
    // fast index, each XCD's workgroups share an eighth of the channel tiles, and the weights - 9 MB against 0.5 MB of
    // input - cross the fabric once instead of eight times: launch_one swaps the roles of the grid's axes.)
    // (Giving an XCD a contiguous eighth of the pixel tiles of the large maps - so that the halo rows neighbouring tiles
    // share, or those and the window the channel tiles of one pixel tile share, are asked from one L2 - was measured too:
    // 0.9 % slower over a whole evaluation in either form.)
    const int tile = p.swap_grid ? blockIdx.y : blockIdx.x;
    const int n0 = (p.swap_grid ? blockIdx.x : blockIdx.y) * TN;
    const int H = p.H, W = p.W, Cin = p.Cin, HW = H * W;
    const int NCB = Cin >> 6;
    const int NST = NCB / CBS;            // stages
    if (tid < 3) flags[tid] = 0;          // (ordered before the vote by the barriers of the pipeline)
    WIN_STAMP(0);

    // ---- tile geometry (all wave-uniform; every extent of a supported map is a power of two).  The window is the run
    // of nwin NHWC pixels from global pixel gstart on; those in [glo, ghi) are staged (the others are never read).
    const int Min = p.B * HW;                    // pixels of the input
    const int sh_out = p.sh_hw - (S2 ? 2 : 0);   // log2 of the pixels of an output map
    const int M = S2 ? Min >> 2 : Min;           // pixels (rows) of the output
    int b0 = 0, pos = 0, halo = 0, gstart, glo = 0, ghi = Min, nwin;
    if (S2) {                                     // whole output images; the window is their input images
        b0 = (tile << LTP) >> sh_out;
        gstart = b0 << p.sh_hw; nwin = 4 * TP;
    } else if (PIXM) {
        const int tpp = p.B >> LTP;               // tiles per output pixel
        pos = tile / tpp;
        b0 = (tile - pos * tpp) << LTP;
        gstart = b0 * 4; nwin = TP * 4;           // the tile's 32 images of 2x2 pixels
    } else if (HW > TP) {                         // rows of one image, and the row above and below
        halo = W;
        b0 = (tile << LTP) >> p.sh_hw;
        gstart = (tile << LTP) - W; nwin = TP + 2 * W;
        glo = b0 << p.sh_hw; ghi = glo + HW;
    } else {                                      // whole images
        b0 = (tile << LTP) >> p.sh_hw;
        gstart = tile << LTP; nwin = TP;
    }
    const int IS = p.lds_img;                     // pixel-major: bytes between the images of the window in LDS
    // ---- this wave's weight stream: fragments [channel tile][slice s][k-step (cb, tap)][hi | lo][lane].  Its steps
    // are numbered g = stage * NSTEPS + i; step i of a stage is k-step j = i (PSPLIT) or grp + 2 i of the stage's
    // CBS * NTAPS (block, tap) pairs.  The first PD steps are requested before anything else.
    const f16x8* wsrc = static_cast<const f16x8*>(p.wf) + ((size_t)((n0 >> 5) * 4 + s) * NCB * 9 * 2) * 64 + lane;
    const int oy = pos >> p.sh_w, ox = pos & (W - 1);   // pixel-major: the tile's output pixel
    auto tap_id = [&](int ti) -> int {                  // index of the stage's tap ti in the 3x3 kernel
        return PIXM ? ((ti >> 1) - oy + 1) * 3 + (ti & 1) - ox + 1 : ti;
    };
    // (tile-uniform) these output channels belong to the second output: a 1x1 kernel carried as the centre tap of a 3x3
    // one, whose other eight taps - zeros - are neither fetched nor multiplied
    const bool second = S2 && p.out2 != nullptr && n0 >= p.n_split;
    auto wload = [&](f16x8 (&dst)[2], int st, int i) {
        const int j = min(PSPLIT ? i : grp + 2 * i, KSTEPS - 1);   // (past the end: the last one again, never used)
        const int cbl = NTAPS == 4 ? j >> 2 : (j * 57) >> 9, ti = j - cbl * NTAPS;
        if (S2 && second && ti != 4) return;
        const int cb = min(st * CBS + cbl, NCB - 1);
        const f16x8* src = wsrc + (size_t)((cb * 9 + tap_id(ti)) * 2) * 64;
        dst[0] = src[0];
        dst[1] = src[64];
    };
    f16x8 wq[PD][2];
#pragma unroll
    for (int d = 0; d < PD; ++d) wload(wq[d], d / NSTEPS, d % NSTEPS);
    WIN_STAMP(9);

    // ---- staging sweep.  Item i = tid + j T of a stage is (window pixel i / (16 CBS), block (i / 16) % CBS of the stage,
    // channels 4 (i % 16) ..+3 of it); the window is a run of pixels in memory and (but for the pixel-major image
    // stride) in LDS, so item j of a thread lies a fixed step behind item j - 1 in both.
    constexpr int LPI = CBS == 4 ? 6 : CBS == 2 ? 5 : 4;   // log2 of the items of a pixel
    constexpr int USTEP = T >> LPI;                         // window pixels between a thread's items
    constexpr int MAXITEMS = (WCAP + USTEP - 1) / USTEP;
    const int nitems = (nwin + USTEP - 1) / USTEP;
    const int c4 = (tid & 15) * 4, cbl_t = (tid >> 4) & (CBS - 1), u0 = tid >> LPI;
    const int goff0 = __mul24(gstart + u0, Cin) + cbl_t * 64 + c4;   // (signed: negative above the first image, where nothing is loaded)
    const int gstep = USTEP * Cin;
    const int loff0 = (PIXM ? (int)__umul24(u0 >> 2, IS) + (u0 & 3) * PS : u0 * PS) + cbl_t * 256 + c4 * 2;
    const int lstep = PIXM ? (USTEP >> 2) * IS : USTEP * PS;
    unsigned staged = 0;       // bit j: item j is a pixel of the window that lies in the map
#pragma unroll
    for (int j = 0; j < MAXITEMS; ++j) {
        const int u = u0 + j * USTEP, g = gstart + u;
        if (u < nwin && g >= glo && g < ghi) staged |= 1u << j;
    }
    WIN_STAMP(1);
    v4f raw[MAXITEMS];
    v4f mean4 = {0.f, 0.f, 0.f, 0.f}, rstd4 = {1.f, 1.f, 1.f, 1.f};
    float amax = 0.f;
    auto load_window = [&](int st) {
#pragma unroll
        for (int j = 0; j < MAXITEMS; ++j)
            if (j < nitems)
                raw[j] = *reinterpret_cast<const v4f*>(p.x + (unsigned)(((staged >> j) & 1u) ? goff0 + j * gstep + st * (CBS * 64) : 0));
        if (NORM) {
            mean4 = *reinterpret_cast<const v4f*>(p.nmean + (unsigned)(b0 * Cin + (st * CBS + cbl_t) * 64 + c4));
            rstd4 = *reinterpret_cast<const v4f*>(p.nrstd + (unsigned)(b0 * Cin + (st * CBS + cbl_t) * 64 + c4));
        }
    };
    auto store_window = [&](int buf_off) {
#pragma unroll
        for (int j = 0; j < MAXITEMS; ++j) {
            if (j >= nitems || !((staged >> j) & 1u)) continue;
            v4f v = raw[j];
            if (NORM) v = (v - mean4) * rstd4;
            amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
            f16x4 hi, lo;
            split4w(v, hi, lo);
            unsigned char* dst = smem + buf_off + loff0 + j * lstep;
            *reinterpret_cast<f16x4*>(dst) = hi;
            *reinterpret_cast<f16x4*>(dst + 128) = lo;
        }
    };
    // the pixel of zeros behind the window of each buffer
    const int zoff = p.lds_window - PS;
    if (tid < 2 * (PS / 16)) {
        const int bsel = tid / (PS / 16);
        if (bsel == 0 || p.lds_bufs == 2)
            *reinterpret_cast<float4*>(smem + bsel * p.lds_window + zoff + (tid - bsel * (PS / 16)) * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    WIN_STAMP(2);
    load_window(0);
    WIN_STAMP(3);

    // ---- lane geometry of the B operand (activations): lane (r, h) of pixel tile t reads 8 halves of its pixel's
    // neighbour under every tap; the nine addresses (the zero pixel where the neighbour is outside the image)
    const int fr = lane & 31, fh = lane >> 5;
    const int t0 = PSPLIT ? grp * NPTW : 0;       // first pixel tile of this wave
    int lbase[NPTW];            // byte address of the lane's own pixel (its fragment of slice s)
    unsigned tapok[NPTW];       // bit ti: the neighbour under tap ti lies inside the image
    const int zaddr = zoff + s * 32 + fh * 16;
#pragma unroll
    for (int t = 0; t < NPTW; ++t) {
        const int q = 32 * (t0 + t) + fr;         // pixel of the tile
        if (S2) {
            const int il = q >> sh_out, r = q & ((1 << sh_out) - 1), oy = r >> (p.sh_w - 1), ox = r & ((W >> 1) - 1);
            lbase[t] = ((il << p.sh_hw) + ((2 * oy) << p.sh_w) + 2 * ox) * PS + s * 32 + fh * 16;
            // input rows 2 oy - 1 .. 2 oy + 1, columns likewise: only the first of each can leave the map
            const unsigned cols = (ox > 0 ? 1u : 0u) | 6u;
            tapok[t] = (oy > 0 ? cols : 0u) | cols << 3 | cols << 6;
        } else if (PIXM) {
            lbase[t] = (int)__umul24(q, IS) + s * 32 + fh * 16;
            tapok[t] = 0xfu;
        } else {
            const int g = (tile << LTP) + q, y = (g >> p.sh_w) & (H - 1), x = g & (W - 1);
            lbase[t] = (halo + q) * PS + s * 32 + fh * 16;
            const unsigned rows = (y > 0 ? 1u : 0u) | 2u | (y < H - 1 ? 4u : 0u);        // kh = 0, 1, 2
            const unsigned cols = (x > 0 ? 1u : 0u) | 2u | (x < W - 1 ? 4u : 0u);        // kw = 0, 1, 2
            tapok[t] = ((rows & 1u) ? cols : 0u) | ((rows & 2u) ? cols << 3 : 0u) | ((rows & 4u) ? cols << 6 : 0u);
        }
    }
    const int wbytes = p.lds_window;              // one window buffer
    const bool two_bufs = p.lds_bufs == 2;

    // ---- what the epilogue adds: 8 threads per pixel row, 4 channels each; bias, residual and embedding bias are
    // fetched while the last stage is multiplied (its staging registers are free by then)
    constexpr int NROWS = (TP * 8 + T - 1) / T;
    const int oc4 = (tid & 7) * 4;
    const int n = n0 + oc4;
    const float scale = p.wscale;
    const int ldc = p.out2 ? p.n_split : p.Cout;   // row of the first output
    auto row_of = [&](int q, int& img) -> int {   // row of the output tensor of tile pixel q, or -1
        if (PIXM) { img = b0 + q; return img * HW + pos; }
        const int m = tile * TP + q;
        img = min(m >> sh_out, p.B - 1);           // (rows past the batch in a ragged last tile: never stored)
        return m < M ? m : -1;
    };
    int erow[NROWS], eimg[NROWS];
    float4 pre_res[NROWS], pre_tb[NROWS];
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int e = 0; e < NROWS; ++e) {
        const int q = (tid >> 3) + e * (T / 8);
        eimg[e] = 0;
        erow[e] = q < TP ? row_of(q, eimg[e]) : -1;
        pre_res[e] = pre_tb[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    auto epilogue_prefetch = [&]() {
        if (p.bias) bv = *reinterpret_cast<const float4*>(p.bias + n);
#pragma unroll
        for (int e = 0; e < NROWS; ++e) {
            const int m = max(erow[e], 0);
            if (p.res) pre_res[e] = *reinterpret_cast<const float4*>(p.res + (size_t)m * ldc + n);
            if (p.tbias) pre_tb[e] = *reinterpret_cast<const float4*>(p.tbias + (size_t)eimg[e] * p.tb_stride + n);
        }
    };

    f32x16 acc[NPTW];
#pragma unroll
    for (int t = 0; t < NPTW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    store_window(0);
    WIN_STAMP(4);
    lds_barrier();
    WIN_STAMP(5);

    for (int st = 0; st < NST; ++st) {
        const int buf = (two_bufs && (st & 1)) ? wbytes : 0;
        const bool more = st + 1 < NST;
        if (more) load_window(st + 1);
        else epilogue_prefetch();
#pragma unroll
        for (int i = 0; i < NSTEPS; ++i) {
            const int j = PSPLIT ? i : grp + 2 * i;
            if ((KSTEPS % TGK) && i == NSTEPS - 1 && j >= KSTEPS) break;   // (one block of nine taps: group 1 has four steps)
            // (keeps a step's reads, products and weight load together where the scheduler otherwise spreads the unrolled
            // steps over one another: 1-1.5 us on the kernels with many steps per stage, nothing or a loss on the others)
            if constexpr (CBS >= 4 || NORM) __builtin_amdgcn_sched_barrier(0);
            const int cbl = NTAPS == 4 ? j >> 2 : (j * 57) >> 9, ti = j - cbl * NTAPS;
            if (S2 && second && ti != 4) continue;
            // where the tap's neighbour lies behind the lane's own pixel (wave-uniform)
            int toff;
            if (PIXM) toff = ti * PS;
            else { const int kh = (ti * 11) >> 5; toff = ((kh - 1) * W + ti - 3 * kh - 1) * PS; }
            const int boff = cbl * 256 + buf;
            f16x8 ah[NPTW], al[NPTW];
#pragma unroll
            for (int t = 0; t < NPTW; ++t) {
                const int at = (((tapok[t] >> ti) & 1u) ? lbase[t] + toff : zaddr) + boff;
                ah[t] = *reinterpret_cast<const f16x8*>(smem + at);
                al[t] = *reinterpret_cast<const f16x8*>(smem + at + 128);
            }
            const f16x8 wh = wq[i % PD][0], wl = wq[i % PD][1];
            // (term by term over the tiles: consecutive instructions write different accumulators; per accumulator lo-hi, hi-lo, hi-hi)
#pragma unroll
            for (int t = 0; t < NPTW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, ah[t], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NPTW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, al[t], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NPTW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, ah[t], acc[t], 0, 0, 0);
            // this slot's next step is PD steps on: in this stage, or in the next one
            if (i + PD < NSTEPS) wload(wq[i % PD], st, i + PD);
            else wload(wq[i % PD], st + 1, i + PD - NSTEPS);
        }
        if (more) {
            if (!two_bufs) lds_barrier();        // one buffer: everybody has read this stage before the next one lands
            store_window(two_bufs && !(st & 1) ? wbytes : 0);
        }
        lds_barrier();
    }

    WIN_STAMP(6);
    // ---- range vote of the split (every staged value is in `amax` by now)
    {
        const bool too_big = !(amax < 32768.f);
        if (__any(too_big) && lane == 0) flags[0] = 1;
        if (__any(amax >= 0.00390625f) && lane == 0) flags[1] = 1;
        if (__any(amax > 0.f) && lane == 0) flags[2] = 1;
    }
    // ---- partial tiles -> LDS (the loop's last barrier has retired every window read).  Waves that shared k-steps:
    // wave 4 + s first hands its tile to wave s, which adds it to its own.  The final sum has four terms, in slice order.
    float* red = reinterpret_cast<float*>(smem);
    float* mine = red + s * TP * LDO;
    auto put_tile = [&]() {
#pragma unroll
        for (int t = 0; t < NPTW; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(mine + (32 * (t0 + t) + fr) * LDO + 8 * g + 4 * fh) =
                    make_float4(acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]);
    };
    if (!PSPLIT) {
        if (grp == 1) put_tile();
        lds_barrier();
        if (grp == 0) {
#pragma unroll
            for (int t = 0; t < NPTW; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 o = *reinterpret_cast<const float4*>(mine + (32 * t + fr) * LDO + 8 * g + 4 * fh);
                    acc[t][4 * g] += o.x; acc[t][4 * g + 1] += o.y; acc[t][4 * g + 2] += o.z; acc[t][4 * g + 3] += o.w;
                }
            put_tile();
        }
    } else {
        put_tile();
    }
    lds_barrier();
    WIN_STAMP(7);
    const bool redo = flags[0] != 0 || (flags[2] != 0 && flags[1] == 0);

    auto finish_store = [&](int m, int img, float4 v, float4 rv, float4 tv) {
        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        if (second) {
            *reinterpret_cast<float4*>(p.out2 + (size_t)m * (p.Cout - p.n_split) + (n - p.n_split)) = v;
            return;
        }
        v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
        v.x = act_apply(p.act, v.x); v.y = act_apply(p.act, v.y); v.z = act_apply(p.act, v.z); v.w = act_apply(p.act, v.w);
        v.x += tv.x; v.y += tv.y; v.z += tv.z; v.w += tv.w;
        *reinterpret_cast<float4*>(p.out + (size_t)m * ldc + n) = v;
    };
    if (!__builtin_amdgcn_readfirstlane(redo ? 1 : 0)) {
#pragma unroll
        for (int e = 0; e < NROWS; ++e) {
            const int q = (tid >> 3) + e * (T / 8);
            if (q >= TP || erow[e] < 0) continue;
            float4 v = *reinterpret_cast<const float4*>(red + q * LDO + oc4);
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float4 o = *reinterpret_cast<const float4*>(red + (w * TP + q) * LDO + oc4);
                v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            }
            v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
            finish_store(erow[e], eimg[e], v, pre_res[e], pre_tb[e]);
        }
#if DANRA_WIN_DEBUG
        __builtin_amdgcn_s_waitcnt(0x0070);  // vmcnt(0): the stores have been acknowledged
#endif
        WIN_STAMP(8);
        return;
    }

    // ---- a staged value left the range in which the split holds 22 bits (2^15 or more, non-finite, or nothing above
    // 2^-8): this workgroup forms its tile again with fp32 multiply-adds from the fp32 weights.  Rare, and slow.
    for (int q = tid >> 3; q < TP; q += T / 8) {
        int img;
        const int m = row_of(q, img);
        if (m < 0) continue;
        const int r = m - (img << sh_out), oy = S2 ? r >> (p.sh_w - 1) : r >> p.sh_w, ox = S2 ? r & ((W >> 1) - 1) : r & (W - 1);
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        const int K = 9 * Cin;
        for (int kh = 0; kh < 3; ++kh)
            for (int kw = 0; kw < 3; ++kw) {
                if (S2 && second && (kh != 1 || kw != 1)) continue;
                const int iy = (S2 ? 2 * oy : oy) + kh - 1, ix = (S2 ? 2 * ox : ox) + kw - 1;
                if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) continue;
                const float* xr = p.x + ((size_t)(img * H + iy) * W + ix) * Cin;
                const float* wr = p.w32 + (size_t)n * K + (kh * 3 + kw) * Cin;
                for (int c = 0; c < Cin; ++c) {
                    float xv = xr[c];
                    if (NORM) xv = (xv - p.nmean[(size_t)img * Cin + c]) * p.nrstd[(size_t)img * Cin + c];
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] = fmaf(wr[(size_t)e * K + c], xv, a[e]);
                }
            }
        float4 rv = make_float4(0.f, 0.f, 0.f, 0.f), tv = rv;
        if (p.res) rv = *reinterpret_cast<const float4*>(p.res + (size_t)m * ldc + n);
        if (p.tbias) tv = *reinterpret_cast<const float4*>(p.tbias + (size_t)img * p.tb_stride + n);
        finish_store(m, img, make_float4(a[0], a[1], a[2], a[3]), rv, tv);
    }
}

// w [Cout][taps][Cin] fp32 (taps row-major) times `scale` (a power of two) -> fragment order of the window kernels:
// out[(((ct * 4 + s) * NCB + cb) * taps + tap) * 2 + hl][lane][e], value = channel 32 ct + (lane & 31), tap, input channel
// 64 cb + 16 s + 8 (lane >> 5) + e; hi = rounded towards zero to fp16, lo = the remainder (as k_split_weights forms them).
__global__ __launch_bounds__(256) void k_pack_conv_frag(const float* __restrict__ w, _Float16* __restrict__ out, int Cout,
                                                        int Cin, float scale, int taps) {
    typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
    const int NCB = Cin >> 6;
    const size_t total = (size_t)Cout * taps * Cin;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int nrow = (int)(i / ((size_t)taps * Cin));
        const int rem = (int)(i - (size_t)nrow * taps * Cin);
        const int tap = rem / Cin, c = rem - tap * Cin;
        const float v = w[i] * scale;
        const fp16x2 h = __builtin_amdgcn_cvt_pkrtz(v, 0.f);
        const int ct = nrow >> 5, cb = c >> 6, s = (c >> 4) & 3, hh = (c >> 3) & 1, e = c & 7;
        const int ln = (nrow & 31) + 32 * hh;
        const size_t frag = ((size_t)((ct * 4 + s) * NCB + cb) * taps + tap) * 2;
        out[(frag * 64 + ln) * 8 + e] = (_Float16)h[0];
        out[((frag + 1) * 64 + ln) * 8 + e] = (_Float16)(v - (float)h[0]);
    }
}

hipError_t launch_pack_conv_frag(const float* w, float* out, int Cout, int Cin, float scale, hipStream_t stream, int taps) {
    if (Cout % 32 || Cin % 64 || taps <= 0) return hipErrorInvalidValue;
    int blocks = (int)(((size_t)Cout * taps * Cin + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_pack_conv_frag, dim3(blocks), dim3(256), 0, stream, w, reinterpret_cast<_Float16*>(out), Cout, Cin, scale, taps);
    return hipGetLastError();
}

namespace {

int log2_pow2(int d) {
    if (d <= 0 || (d & (d - 1))) return -1;
    int s = 0;
    while ((1 << s) < d) ++s;
    return s;
}

struct WinPlan {
    int npt = 0, cbs = 1, bufs = 1, cap = 0;
    bool pixm = false;
    int tiles = 0, nwin = 0;
    int lds_img = 0, lds_window = 0;  // bytes: between the images of a pixel-major window, of one window buffer
    size_t lds = 0;
};

// smallest multiple of 16 that is >= least and congruent to 16 * slots modulo 256
int stride_for(int least, int slots) {
    int v = (least + 15) / 16 * 16;
    while ((v / 16) % 16 != slots % 16) v += 16;
    return v;
}

// window capacity of a kernel instance (WCAP) x channel blocks per stage: what the staging registers of a thread hold
// (12 items of four values)
constexpr int kMaxCapBlocks = 384;
int cap_class(int npt, bool pixm, int nwin) {
    if (pixm) return 128;
    if (npt == 4) return nwin <= 160 ? 160 : nwin <= 192 ? 192 : 256;
    if (npt == 2) return nwin <= 64 ? 64 : nwin <= 128 ? 128 : 192;
    return nwin <= 32 ? 32 : 96;
}
constexpr size_t kLdsBudget = 160 * 1024 - 1024;   // dynamic LDS a workgroup may ask for

// Stride 2: tiles of whole output images (at most 32 NPT pixels each, their input images in the window).
WinPlan plan_conv_win_s2(int B, int H, int W, int Cin, int Cout) {
    WinPlan pl;
    if (log2_pow2(W) < 0 || log2_pow2(H) < 0 || H < 2 || W < 2 || Cin % 64 || Cout % 32 || B <= 0) return pl;
    const int HWo = H * W / 4;
    const long M = (long)B * HWo;
    if ((long)B * H * W * (long)(Cin > Cout ? Cin : Cout) >= (1l << 31) || (long)B * H * W >= (1l << 23)) return pl;
    const int NCB = Cin / 64;
    for (int npt : {2, 1}) {
        const int TP = 32 * npt;
        if (HWo > TP) continue;
        if (npt == 2 && HWo <= 32 && (M + TP - 1) / TP * (Cout / 32) < 200) continue;   // (rather twice as many workgroups)
        pl.nwin = 4 * TP;
        pl.cap = pl.nwin;                       // the instances there are: 256 window pixels for NPT = 2, 128 for NPT = 1
        for (int cbs : {2, 1}) {
            if (NCB % cbs || pl.cap * cbs > 256) continue;
            if (cbs == 1 && NCB > 1) continue;  // (as at stride 1: two wave groups would deal nine taps of one block)
            const int PSB = pixel_bytes(cbs);
            const int lds_window = (pl.nwin * PSB + PSB + 255) / 256 * 256;
            const int stages = NCB / cbs;
            int bufs = stages > 1 ? 2 : 1;
            if ((size_t)lds_window * bufs > kLdsBudget) bufs = 1;
            if ((size_t)lds_window > kLdsBudget) continue;
            pl.npt = npt; pl.cbs = cbs; pl.bufs = bufs; pl.pixm = false;
            pl.tiles = (int)((M + TP - 1) / TP);
            pl.lds_img = 0; pl.lds_window = lds_window;
            const size_t red = (size_t)4 * TP * (32 + 4) * sizeof(float), win = (size_t)lds_window * bufs;
            pl.lds = win > red ? win : red;
            return pl;
        }
    }
    pl.npt = 0;
    return pl;
}

// The tile shape for a convolution, or npt = 0 when k_conv_win does not take it.
WinPlan plan_conv_win(int B, int H, int W, int Cin, int Cout, bool norm) {
    WinPlan pl;
    if (log2_pow2(W) < 0 || log2_pow2(H) < 0 || W > 64 || Cin % 64 || Cout % 32 || B <= 0) return pl;
    const int HW = H * W;
    const long M = (long)B * HW;
    if (M * (long)(Cin > Cout ? Cin : Cout) >= (1l << 31) || M >= (1l << 23)) return pl;   // 32-bit element offsets, 24-bit pixel indices
    auto fits = [&](int npt, bool pixm) -> bool {
        const int TP = 32 * npt;
        if (pixm) return H == 2 && W == 2 && B % TP == 0 && !norm;
        if (HW > TP) return TP % W == 0;                                 // whole rows of one image (+ the rows above and below)
        return !norm;                                                    // whole images (their statistics differ: no norm on load)
    };
    const int ctiles = Cout / 32;
    int npt = 0;
    bool pixm = false;
    if (fits(1, true)) { npt = 1; pixm = true; }
    else {
        // the largest pixel tile that still leaves a workgroup for (nearly) every CU.  (Measured around that rule at 64 images:
        // twice as many workgroups of half the size - two per CU, one's set-up behind the other's loop - and half as many of
        // twice the size both cost 6 % of an evaluation.)
        for (int cand : {4, 2, 1}) {
            if (!fits(cand, false)) continue;
            const long wgs = (M + 32 * cand - 1) / (32 * cand) * ctiles;
            if (wgs >= 200 || cand == 1) { npt = cand; break; }
        }
        if (!npt) return pl;
    }
    pl.npt = npt; pl.pixm = pixm;
    const int TP = 32 * npt;
    pl.tiles = pixm ? (B / TP) * HW : (int)((M + TP - 1) / TP);
    pl.nwin = pixm ? TP * 4 : HW > TP ? TP + 2 * W : TP;
    pl.cap = cap_class(npt, pixm, pl.nwin);
    const size_t red = (size_t)4 * TP * (32 + 4) * sizeof(float);
    // channel blocks per stage: as many as LDS and the staging registers hold
    const int NCB = Cin / 64;
    for (int cbs : {4, 2, 1}) {
        if (NCB % cbs) continue;
        if (pl.cap * cbs > kMaxCapBlocks) continue;
        // (two wave groups that deal ONE block's nine taps between them: five and four, with no second stage to even it out)
        if (cbs == 1 && NCB > 1 && !pixm && npt != 4) continue;
        // LDS image of the window: the pixels one after the other, 16 cbs + 1 sixteen-byte slots each.  A fragment read
        // (ds_read_b128) is served 16 lanes at a time and is conflict-free when those lanes' slots differ modulo 16:
        // consecutive lanes read consecutive pixels, so they do.  The lanes of a pixel-major tile read consecutive
        // IMAGES: the image stride is padded to one slot modulo 16.
        const int PSB = pixel_bytes(cbs);
        const int lds_img = pixm ? stride_for(4 * PSB, 1) : 0;
        const int lds_window = ((pixm ? TP * lds_img : pl.nwin * PSB) + PSB + 255) / 256 * 256;   // + the pixel of zeros
        const int stages = NCB / cbs;
        int bufs = stages > 1 ? 2 : 1;
        if ((size_t)lds_window * bufs > kLdsBudget) bufs = 1;
        if ((size_t)lds_window > kLdsBudget) continue;
        pl.cbs = cbs; pl.bufs = bufs;
        pl.lds_img = lds_img; pl.lds_window = lds_window;
        const size_t win = (size_t)lds_window * bufs;
        pl.lds = win > red ? win : red;
        return pl;
    }
    pl.npt = 0;
    return pl;
}

template <int NPT, int WCAP, int CBS, bool NORM, bool PIXM, bool PSPLIT, bool S2 = false>
hipError_t launch_one(const ConvWinArgs& a, const WinPlan& pl, hipStream_t stream) {
    auto kern = k_conv_win<NPT, WCAP, CBS, NORM, PIXM, PSPLIT, S2>;
    if (pl.lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds);
        if (e != hipSuccess) return e;
    }
    ConvWinArgs b = a;
    b.swap_grid = (PIXM || (S2 && pl.tiles <= 8)) ? 1 : 0;
    const dim3 grid = b.swap_grid ? dim3(a.Cout / 32, pl.tiles) : dim3(pl.tiles, a.Cout / 32);
    DANRA_LAUNCH_KERNEL(kern, grid, dim3(512), pl.lds, stream, b);
    return hipGetLastError();
}

template <int NPT, int WCAP, bool NORM, bool PIXM, bool PSPLIT>
hipError_t launch_cbs(const ConvWinArgs& a, const WinPlan& pl, hipStream_t stream) {
    if constexpr (WCAP * 4 <= kMaxCapBlocks) {
        if (pl.cbs == 4) return launch_one<NPT, WCAP, 4, NORM, PIXM, PSPLIT>(a, pl, stream);
    }
    if constexpr (WCAP * 2 <= kMaxCapBlocks) {
        if (pl.cbs == 2) return launch_one<NPT, WCAP, 2, NORM, PIXM, PSPLIT>(a, pl, stream);
    }
    if (pl.cbs != 1) return hipErrorInvalidValue;
    return launch_one<NPT, WCAP, 1, NORM, PIXM, PSPLIT>(a, pl, stream);
}

}  // namespace

bool conv_win_supported(int B, int H, int W, int Cin, int Cout, bool norm, int stride) {
    if (stride == 2) return !norm && plan_conv_win_s2(B, H, W, Cin, Cout).npt != 0;
    return stride == 1 && plan_conv_win(B, H, W, Cin, Cout, norm).npt != 0;
}

hipError_t launch_conv_win(const ConvWinArgs& args, hipStream_t stream) {
    ConvWinArgs a = args;
    const bool norm = a.nmean != nullptr;
    if (a.stride == 0) a.stride = 1;
    if (a.stride != 1 && a.stride != 2) return hipErrorInvalidValue;
    if (a.out2 && (a.stride != 2 || a.n_split <= 0 || a.n_split % 32 || a.n_split >= a.Cout || a.res || a.tbias)) return hipErrorInvalidValue;
    if (!a.out2) a.n_split = 0;
    if (a.stride == 2 && norm) return hipErrorInvalidValue;
    const WinPlan pl = a.stride == 2 ? plan_conv_win_s2(a.B, a.H, a.W, a.Cin, a.Cout) : plan_conv_win(a.B, a.H, a.W, a.Cin, a.Cout, norm);
    if (!pl.npt || !a.wf || !a.w32) return hipErrorInvalidValue;
    if (a.tbias && a.tb_stride % 4) return hipErrorInvalidValue;
    a.sh_w = log2_pow2(a.W);
    a.sh_hw = log2_pow2(a.H * a.W);
    a.lds_img = pl.lds_img; a.lds_window = pl.lds_window; a.lds_bufs = pl.bufs;
    if (a.stride == 2) {
        if (pl.npt == 2 && pl.cbs == 1) return launch_one<2, 256, 1, false, false, false, true>(a, pl, stream);
        if (pl.npt == 1 && pl.cbs == 2) return launch_one<1, 128, 2, false, false, false, true>(a, pl, stream);
        if (pl.npt == 1 && pl.cbs == 1) return launch_one<1, 128, 1, false, false, false, true>(a, pl, stream);
        return hipErrorInvalidValue;
    }
#define DANRA_WIN(NPT_, CAP_, PS_)                                                                                    \
    if (pl.npt == NPT_ && pl.cap == CAP_)                                                                             \
        return norm ? launch_cbs<NPT_, CAP_, true, false, PS_>(a, pl, stream) : launch_cbs<NPT_, CAP_, false, false, PS_>(a, pl, stream);
    if (pl.pixm) return launch_cbs<1, 128, false, true, false>(a, pl, stream);
    DANRA_WIN(4, 160, true)
    DANRA_WIN(4, 192, true)
    DANRA_WIN(4, 256, true)
    DANRA_WIN(2, 64, false)
    DANRA_WIN(2, 128, false)
    DANRA_WIN(2, 192, false)
    DANRA_WIN(1, 32, false)
    DANRA_WIN(1, 96, false)
#undef DANRA_WIN
    return hipErrorInvalidValue;
}

}  // namespace danra
