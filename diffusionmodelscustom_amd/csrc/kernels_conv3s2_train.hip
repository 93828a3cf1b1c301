// Training-step kernels for the 3x3, stride-2, padding-1 convolutions of the network (the first convolution of the
// BasicBlocks that halve the map), on contiguous NCHW fp32 tensors with even sides and the [Cout][Cin][3][3] weights as
// torch holds them; all products on v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains), every sum over workgroups through
// slabs added in slab order (bit-reproducible, no atomics).
//
//   forward        y[b][co][oy][ox]  = sum_{ci,kh,kw} w[co][ci][kh][kw] x[b][ci][2 oy - 1 + kh][2 ox - 1 + kw]
//   data gradient  dx[b][ci][iy][ix] = sum_{co} sum_{kh,kw of the parity of (iy + 1, ix + 1)}
//                                      w[co][ci][kh][kw] dy[b][co][(iy + 1 - kh) / 2][(ix + 1 - kw) / 2]
//   weight grad.   dw[co][ci][kh][kw] = sum_{b,oy,ox} dy[b][co][oy][ox] x[b][ci][2 oy - 1 + kh][2 ox - 1 + kw]
//
// A "tile" is 64 consecutive pixels of the HALF-resolution map (Ho x Wo = H/2 x W/2, the forward output and the grid of
// 2x2 cells of the input) in (image, row, column) order: 64 / Wo whole rows of one image, or 64 / (Ho Wo) whole images
// where an image has fewer than 64 such pixels.  What a tile needs of the other tensor is staged in LDS as a window with
// a zero halo (the full-resolution window has its halo above and to the left: row 2 oy + 1 never leaves the image; the
// half-resolution window of the data gradient below and to the right), so the product loops hold no bounds check.
//
// The data gradient takes the four pixels of an input 2x2 cell (u, v) as four outputs of the cell: pixel parity (py, px)
// uses kh = 1 at row u for py = 0, and kh = 0 at row u + 1, kh = 2 at row u for py = 1 (columns alike): the nine taps are
// each used by exactly one parity, so the four accumulators cost nine matrix instructions per pair of channels.
#include "danra_kernels.h"

namespace danra {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int S2_KC = 16;                 // reduction channels staged per chunk (forward: ci; data gradient: co)
constexpr int S2_LDW = S2_KC * 9 + 1;     // LDS row of the staged weights (odd)
constexpr int S2_NO = 32;                 // output channels per workgroup (forward / data gradient)

struct S2Geom {
    int B, H, W;           // the full-resolution tensor (x, dx)
    int Ho, Wo, logWo;     // the half-resolution tensor (y, dy)
    int NI, RH;            // images and half-resolution rows per image of a tile
    int tiles_per_img;     // NI == 1: Ho / RH
    int n_tiles;
    int WW, WIMG, WIN;     // full-resolution window: row W + 1, per image (2 RH + 1) rows, per channel NI images (odd total)
    int DW, DIMG, DWIN;    // half-resolution window: row Wo + 1, per image (RH + 1) rows
};

bool s2_geom(int B, int H, int W, S2Geom& g) {
    if (B <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1)) return false;
    g.B = B; g.H = H; g.W = W;
    g.Ho = H / 2; g.Wo = W / 2;
    if (g.Wo < 2 || g.Wo > 64 || (g.Wo & (g.Wo - 1))) return false;   // (a pixel pair of the weight gradient shares a row)
    g.logWo = 0;
    while ((1 << g.logWo) < g.Wo) ++g.logWo;
    const int hwo = g.Ho * g.Wo;
    if (hwo >= 64) {
        g.NI = 1;
        g.RH = 64 / g.Wo;
        if (g.Ho % g.RH) return false;
        g.tiles_per_img = g.Ho / g.RH;
        g.n_tiles = B * g.tiles_per_img;
    } else {
        if (64 % hwo) return false;
        g.NI = 64 / hwo;
        g.RH = g.Ho;
        g.tiles_per_img = 0;
        g.n_tiles = (B + g.NI - 1) / g.NI;
    }
    g.WW = W + 1;
    g.WIMG = (2 * g.RH + 1) * g.WW;
    g.WIN = (g.NI * g.WIMG) | 1;
    g.DW = g.Wo + 1;
    g.DIMG = (g.RH + 1) * g.DW;
    g.DWIN = (g.NI * g.DIMG) | 1;
    return g.WIN <= 512 && g.DWIN <= 256;
}

// pixel p (0..63) of a tile: image of the tile, row of that image's share, column
__device__ __forceinline__ void tile_px(const S2Geom& g, int p, int& il, int& r, int& c) {
    const int R = p >> g.logWo;
    c = p & (g.Wo - 1);
    il = g.NI == 1 ? 0 : R / g.RH;
    r = R - il * g.RH;
}

// first image and first half-resolution row of a tile
__device__ __forceinline__ void tile_origin(const S2Geom& g, int tile, int& b0, int& r0) {
    if (g.NI == 1) {
        b0 = tile / g.tiles_per_img;
        r0 = (tile - b0 * g.tiles_per_img) * g.RH;
    } else {
        b0 = tile * g.NI;
        r0 = 0;
    }
}

// source of full-resolution window element e (< NI * WIMG) of a tile: offset inside a channel plane of image b, or -1 (halo /
// past the last image)
__device__ __forceinline__ int full_window_source(const S2Geom& g, int b0, int r0, int e, int& b) {
    const int il = e / g.WIMG, rem = e - il * g.WIMG;
    const int row = rem / g.WW, col = rem - row * g.WW;
    const int iy = 2 * r0 - 1 + row, ix = col - 1;
    b = b0 + il;
    return (b < g.B && iy >= 0 && ix >= 0) ? iy * g.W + ix : -1;   // (iy <= H - 1 and ix <= W - 1 by construction)
}

__device__ __forceinline__ int half_window_source(const S2Geom& g, int b0, int r0, int e, int& b) {
    const int il = e / g.DIMG, rem = e - il * g.DIMG;
    const int row = rem / g.DW, col = rem - row * g.DW;
    const int oy = r0 + row, ox = col;
    b = b0 + il;
    return (b < g.B && oy < g.Ho && ox < g.Wo) ? oy * g.Wo + ox : -1;
}

__device__ __forceinline__ void zero16(f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// Forward.  Workgroup: one tile x 32 output channels x a range of 16-channel chunks (blockIdx.z: slab z).  Wave w: pixel
// half w & 1 and half w >> 1 of a chunk's 72 two-deep steps (step = (channel pair, tap); the lane half picks the channel).
__global__ __launch_bounds__(256) void k_conv3s2_train_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                           float* __restrict__ y, const S2Geom g, int Cin, int Cout,
                                                           int chunks_per_split) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wts = smem;                     // [32][S2_LDW]
    float* win = smem + S2_NO * S2_LDW;    // [S2_KC][WIN]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pt = wave & 1, kh = wave >> 1;
    const int l32 = lane & 31, kk = lane >> 5;
    const int tile = blockIdx.x, n0 = blockIdx.y * S2_NO;
    const int nchunks = Cin / S2_KC;
    const int c_begin = blockIdx.z * chunks_per_split, c_end = min(nchunks, c_begin + chunks_per_split);
    const int hw = g.H * g.W, hwo = g.Ho * g.Wo;
    int b0, r0;
    tile_origin(g, tile, b0, r0);

    // window positions tid and tid + 256: offset of the element inside x for channel 0 of a chunk, -1 halo, -2 none
    int wsrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = tid + 256 * i;
        wsrc[i] = -2;
        if (e < g.NI * g.WIMG) {
            int b;
            const int o = full_window_source(g, b0, r0, e, b);
            wsrc[i] = o >= 0 ? b * Cin * hw + o : -1;
        } else if (e < g.WIN) {
            wsrc[i] = -1;
        }
    }
    constexpr int NA = S2_NO * S2_KC * 9 / 256;  // 18
    float xreg[S2_KC][2], areg[NA];
    auto load_chunk = [&](int c) {
        const float* src = x + (size_t)c * S2_KC * hw;
#pragma unroll
        for (int cl = 0; cl < S2_KC; ++cl)
#pragma unroll
            for (int i = 0; i < 2; ++i) xreg[cl][i] = src[(size_t)cl * hw + max(wsrc[i], 0)];
#pragma unroll
        for (int i = 0; i < NA; ++i) {   // element e: (n, r = (channel of the chunk, tap)): consecutive floats of w
            const int e = tid + 256 * i, n = e / (S2_KC * 9), r = e - n * (S2_KC * 9);
            areg[i] = w[((size_t)(n0 + n) * Cin + c * S2_KC) * 9 + r];
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int cl = 0; cl < S2_KC; ++cl)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (wsrc[i] > -2) win[cl * g.WIN + tid + 256 * i] = wsrc[i] >= 0 ? xreg[cl][i] : 0.f;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = tid + 256 * i, n = e / (S2_KC * 9), r = e - n * (S2_KC * 9);
            wts[n * S2_LDW + r] = areg[i];
        }
    };

    int il, rl, cl0;
    tile_px(g, pt * 32 + l32, il, rl, cl0);
    const float* arow = wts + l32 * S2_LDW + kk * 9;
    const float* bcol = win + kk * g.WIN + il * g.WIMG + 2 * rl * g.WW + 2 * cl0;   // tap (kh, kw): + kh * WW + kw
    int tapoff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) tapoff[t] = (t / 3) * g.WW + (t % 3);
    f32x16 acc;
    zero16(acc);

    if (c_begin < c_end) load_chunk(c_begin);
    for (int c = c_begin; c < c_end; ++c) {
        __syncthreads();
        store_chunk();
        __syncthreads();
        if (c + 1 < c_end) load_chunk(c + 1);
#pragma unroll
        for (int s = 0; s < S2_KC / 4 * 9; ++s) {
            const int pair = kh * (S2_KC / 4) + s / 9, t = s % 9;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[pair * 18 + t], bcol[pair * 2 * g.WIN + tapoff[t]], acc, 0, 0, 0);
        }
    }

    __syncthreads();
    float* red = smem;  // 2 x 16 x 64 floats
    if (kh == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(pt * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (kh == 1) return;
    const int b = b0 + il;
    if (b >= g.B) return;
    float* dst = y + (size_t)blockIdx.z * ((size_t)g.B * Cout * hwo) + ((size_t)b * Cout + n0) * hwo + (r0 + rl) * g.Wo + cl0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = (r >> 2) * 8 + kk * 4 + (r & 3);
        dst[(size_t)n * hwo] = acc[r] + red[(pt * 16 + r) * 64 + lane];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Data gradient.  Workgroup: one tile of 64 input 2x2 cells x 32 input channels x a range of 16-channel chunks of the
// output channels (blockIdx.z: slab z).  Wave w: cell half w & 1, four accumulators (the cell's pixels), half w >> 1 of
// a chunk's channel pairs; the lane half picks the channel of a pair.
__global__ __launch_bounds__(256) void k_conv3s2_train_dgrad(const float* __restrict__ dy, const float* __restrict__ w,
                                                             float* __restrict__ dx, const S2Geom g, int Cin, int Cout,
                                                             int chunks_per_split) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wts = smem;                     // [32 ci][S2_LDW]: (co of the chunk, tap)
    float* dwin = smem + S2_NO * S2_LDW;   // [S2_KC][DWIN]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pt = wave & 1, kh = wave >> 1;
    const int l32 = lane & 31, kk = lane >> 5;
    const int tile = blockIdx.x, n0 = blockIdx.y * S2_NO;
    const int nchunks = Cout / S2_KC;
    const int c_begin = blockIdx.z * chunks_per_split, c_end = min(nchunks, c_begin + chunks_per_split);
    const int hw = g.H * g.W, hwo = g.Ho * g.Wo;
    int b0, r0;
    tile_origin(g, tile, b0, r0);

    int wsrc = -2;
    if (tid < g.NI * g.DIMG) {
        int b;
        const int o = half_window_source(g, b0, r0, tid, b);
        wsrc = o >= 0 ? b * Cout * hwo + o : -1;
    } else if (tid < g.DWIN) {
        wsrc = -1;
    }
    constexpr int NA = S2_NO * S2_KC * 9 / 256;  // 18
    float dreg[S2_KC], areg[NA];
    auto load_chunk = [&](int c) {
        const float* src = dy + (size_t)c * S2_KC * hwo + max(wsrc, 0);
#pragma unroll
        for (int cl = 0; cl < S2_KC; ++cl) dreg[cl] = src[(size_t)cl * hwo];
#pragma unroll
        for (int i = 0; i < NA; ++i) {   // element e: (co of the chunk, ci, tap): for one co 32 x 9 consecutive floats
            const int e = tid + 256 * i, cl = e / 288, rem = e - cl * 288;
            areg[i] = w[((size_t)(c * S2_KC + cl) * Cin + n0) * 9 + rem];
        }
    };
    auto store_chunk = [&]() {
        if (wsrc > -2) {
#pragma unroll
            for (int cl = 0; cl < S2_KC; ++cl) dwin[cl * g.DWIN + tid] = wsrc >= 0 ? dreg[cl] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = tid + 256 * i, cl = e / 288, rem = e - cl * 288, n = rem / 9, t = rem - n * 9;
            wts[n * S2_LDW + cl * 9 + t] = areg[i];
        }
    };

    int il, ul, vl;
    tile_px(g, pt * 32 + l32, il, ul, vl);
    const float* arow = wts + l32 * S2_LDW + kk * 9;                              // + pair * 18 + tap
    const float* bcol = dwin + kk * g.DWIN + il * g.DIMG + ul * g.DW + vl;        // + pair * 2 * DWIN + rowoff * DW + coloff
    f32x16 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) zero16(acc[q]);

    if (c_begin < c_end) load_chunk(c_begin);
    for (int c = c_begin; c < c_end; ++c) {
        __syncthreads();
        store_chunk();
        __syncthreads();
        if (c + 1 < c_end) load_chunk(c + 1);
#pragma unroll
        for (int s = 0; s < S2_KC / 4; ++s) {
            const int pair = kh * (S2_KC / 4) + s;
            const float* a = arow + pair * 18;
            const float* b = bcol + pair * 2 * g.DWIN;
#pragma unroll
            for (int th = 0; th < 3; ++th) {
                const int py = th == 1 ? 0 : 1, ro = th == 0 ? g.DW : 0;   // kh = 1: parity 0, row u; kh = 0: row u + 1; kh = 2: row u
#pragma unroll
                for (int tw = 0; tw < 3; ++tw) {
                    const int px = tw == 1 ? 0 : 1, co = tw == 0 ? 1 : 0;
                    acc[py * 2 + px] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[th * 3 + tw], b[ro + co], acc[py * 2 + px], 0, 0, 0);
                }
            }
        }
    }

    __syncthreads();
    float* red = smem;  // [2 cell halves][4 pixels][16][64 lanes] = 8192 floats
    if (kh == 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((pt * 4 + q) * 16 + r) * 64 + lane] = acc[q][r];
    }
    __syncthreads();
    if (kh == 1) return;
    const int b = b0 + il;
    if (b >= g.B) return;
    float* dst = dx + (size_t)blockIdx.z * ((size_t)g.B * Cin * hw) + ((size_t)b * Cin + n0) * hw +
                 (size_t)(2 * (r0 + ul)) * g.W + 2 * vl;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = (r >> 2) * 8 + kk * 4 + (r & 3);
#pragma unroll
        for (int py = 0; py < 2; ++py) {
            float2 v;
            v.x = acc[py * 2][r] + red[((pt * 4 + py * 2) * 16 + r) * 64 + lane];
            v.y = acc[py * 2 + 1][r] + red[((pt * 4 + py * 2 + 1) * 16 + r) * 64 + lane];
            *reinterpret_cast<float2*>(dst + (size_t)n * hw + py * g.W) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Weight gradient.  Workgroup: 64 output channels x 32 input channels x a range of tiles (blockIdx.z: slab z).  Wave w:
// output-channel half w & 1, half w >> 1 of a tile's 32 pixel pairs; nine accumulators (taps; rows = co, columns = ci).
__global__ __launch_bounds__(256) void k_conv3s2_train_wgrad(const float* __restrict__ x, const float* __restrict__ dy,
                                                             float* __restrict__ dw, const S2Geom g, int Cin, int Cout,
                                                             int tiles_per_split) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int LDY = 65;
    float* dyt = smem;              // [64 co][65]
    float* win = smem + 64 * LDY;   // [32 ci][WIN] (WIN odd)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cot = wave & 1, ph = wave >> 1;
    const int l32 = lane & 31, kk = lane >> 5;
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 64;
    const int t_begin = blockIdx.z * tiles_per_split, t_end = min(g.n_tiles, t_begin + tiles_per_split);
    const int hw = g.H * g.W, hwo = g.Ho * g.Wo;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) zero16(acc[t]);
    int tapoff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) tapoff[t] = (t / 3) * g.WW + (t % 3);

    for (int tile = t_begin; tile < t_end; ++tile) {
        int b0, r0;
        tile_origin(g, tile, b0, r0);
        __syncthreads();  // the previous tile has been multiplied
        {   // dy tile: 64 channels x 64 pixels; a thread keeps one pixel and walks 16 channels
            const int p = tid & 63, cq = tid >> 6;
            int il, rl, cl;
            tile_px(g, p, il, rl, cl);
            const bool ok = b0 + il < g.B;
            const float* src = dy + ((size_t)(ok ? b0 + il : 0) * Cout + co0 + cq) * hwo + (r0 + rl) * g.Wo + cl;
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = src[(size_t)(4 * i) * hwo];
#pragma unroll
            for (int i = 0; i < 16; ++i) dyt[(cq + 4 * i) * LDY + p] = ok ? v[i] : 0.f;
        }
        for (int e = tid; e < g.WIN; e += 256) {   // window: a thread keeps one position and walks the 32 channels
            int b = 0, o = -1;
            if (e < g.NI * g.WIMG) o = full_window_source(g, b0, r0, e, b);
            const float* src = x + (o >= 0 ? ((size_t)b * Cin + ci0) * hw + o : 0);
            float v[32];
#pragma unroll
            for (int c = 0; c < 32; ++c) v[c] = src[(size_t)c * hw];
#pragma unroll
            for (int c = 0; c < 32; ++c) win[c * g.WIN + e] = o >= 0 ? v[c] : 0.f;
        }
        __syncthreads();
        const float* arow = dyt + (cot * 32 + l32) * LDY + kk;
        const float* bcol = win + l32 * g.WIN + 2 * kk;
#pragma unroll 2
        for (int s = 0; s < 16; ++s) {
            const int p = 2 * (ph * 16 + s);   // pixels p and p + 1 share a row (Wo is even)
            int il, rl, cl;
            tile_px(g, p, il, rl, cl);
            const float* bp = bcol + il * g.WIMG + 2 * rl * g.WW + 2 * cl;
            const float a = arow[p];
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[tapoff[t]], acc[t], 0, 0, 0);
        }
    }

    float* dst = dw + (size_t)blockIdx.z * ((size_t)Cout * Cin * 9);
    float* red = smem;  // [9 taps][16 registers][64 lanes]
    for (int half = 0; half < 2; ++half) {
        __syncthreads();
        if (cot == half && ph == 1) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(t * 16 + r) * 64 + lane] = acc[t][r];
        }
        __syncthreads();
        if (cot == half && ph == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + half * 32 + (r >> 2) * 8 + kk * 4 + (r & 3);
                float* o = dst + ((size_t)co * Cin + ci0 + l32) * 9;
#pragma unroll
                for (int t = 0; t < 9; ++t) o[t] = acc[t][r] + red[(t * 16 + r) * 64 + lane];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// host side
namespace {

void s2_plan_split(const S2Geom& g, int Cred, int Nout, int& ksplit, int& cps) {
    const int nchunks = Cred / S2_KC;
    const long base = (long)g.n_tiles * (Nout / S2_NO);
    ksplit = 1;
    while (base * ksplit < 256 && nchunks / (ksplit * 2) >= 2 && ksplit < 16) ksplit *= 2;
    cps = (nchunks + ksplit - 1) / ksplit;
    ksplit = (nchunks + cps - 1) / cps;
}

void s2_plan_wgrad(const S2Geom& g, int Cin, int Cout, int& slices, int& tps) {
    const long tiles_out = (long)(Cin / 32) * (Cout / 64);
    long s = tiles_out >= 128 ? 1 : (512 + tiles_out - 1) / tiles_out;
    if (s > g.n_tiles) s = g.n_tiles;
    if (s > 128) s = 128;
    if (s < 1) s = 1;
    tps = (int)((g.n_tiles + s - 1) / s);
    slices = (g.n_tiles + tps - 1) / tps;
}

hipError_t s2_sum(const float* part, float* out, size_t count, int slices, hipStream_t stream) {
    hipLaunchKernelGGL(k_sum_slices, dim3((unsigned)((count / 4 + 31) / 32)), dim3(256), 0, stream, part, out, (int)(count / 4),
                       slices);
    return hipGetLastError();
}

template <typename K>
hipError_t s2_lds(K kernel, size_t bytes) {
    if (bytes <= 48 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace

// x [B][Cin][H][W], H and W even, W/2 a power of two up to 64, H/2 * W/2 a divisor or a multiple (by whole rows) of 64;
// Cin a multiple of 32, Cout a multiple of 64
bool conv3x3s2_train_supported(int B, int Cin, int Cout, int H, int W) {
    S2Geom g;
    if (!s2_geom(B, H, W, g)) return false;
    if (Cin <= 0 || Cout <= 0 || Cin % 32 || Cout % 64) return false;
    if (((size_t)B * Cout * g.Ho * g.Wo) % 4 || ((size_t)B * Cin * H * W) % 4) return false;
    const long long big = (long long)B * (Cin > Cout ? Cin : Cout) * H * W;
    return big < (1LL << 31) && (long long)Cin * Cout * 9 < (1LL << 31);
}

size_t conv3x3s2_train_workspace_floats(int B, int Cin, int Cout, int H, int W) {
    S2Geom g;
    if (!s2_geom(B, H, W, g)) return 0;
    int ks, cps, sl, tps;
    size_t need = 4;
    s2_plan_split(g, Cin, Cout, ks, cps);
    if (ks > 1) need = std::max(need, (size_t)ks * B * Cout * g.Ho * g.Wo);
    s2_plan_split(g, Cout, Cin, ks, cps);
    if (ks > 1) need = std::max(need, (size_t)ks * B * Cin * H * W);
    s2_plan_wgrad(g, Cin, Cout, sl, tps);
    if (sl > 1) need = std::max(need, (size_t)sl * Cin * Cout * 9);
    return need;
}

hipError_t launch_conv3x3s2_train_fwd(const float* x, const float* w, float* y, int B, int Cin, int Cout, int H, int W,
                                      float* workspace, hipStream_t stream) {
    S2Geom g;
    if (!conv3x3s2_train_supported(B, Cin, Cout, H, W) || !s2_geom(B, H, W, g)) return hipErrorInvalidValue;
    int ks, cps;
    s2_plan_split(g, Cin, Cout, ks, cps);
    const size_t lds = std::max((size_t)S2_NO * S2_LDW + (size_t)S2_KC * g.WIN, (size_t)2048) * sizeof(float);
    hipError_t e = s2_lds(k_conv3s2_train_fwd, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_conv3s2_train_fwd, dim3(g.n_tiles, Cout / S2_NO, ks), dim3(256), lds, stream, x, w, ks > 1 ? workspace : y,
                       g, Cin, Cout, cps);
    e = hipGetLastError();
    if (e != hipSuccess || ks == 1) return e;
    return s2_sum(workspace, y, (size_t)B * Cout * g.Ho * g.Wo, ks, stream);
}

hipError_t launch_conv3x3s2_train_dgrad(const float* dy, const float* w, float* dx, int B, int Cin, int Cout, int H, int W,
                                        float* workspace, hipStream_t stream) {
    S2Geom g;
    if (!conv3x3s2_train_supported(B, Cin, Cout, H, W) || !s2_geom(B, H, W, g)) return hipErrorInvalidValue;
    int ks, cps;
    s2_plan_split(g, Cout, Cin, ks, cps);
    const size_t lds = std::max((size_t)S2_NO * S2_LDW + (size_t)S2_KC * g.DWIN, (size_t)8192) * sizeof(float);
    hipError_t e = s2_lds(k_conv3s2_train_dgrad, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_conv3s2_train_dgrad, dim3(g.n_tiles, Cin / S2_NO, ks), dim3(256), lds, stream, dy, w,
                       ks > 1 ? workspace : dx, g, Cin, Cout, cps);
    e = hipGetLastError();
    if (e != hipSuccess || ks == 1) return e;
    return s2_sum(workspace, dx, (size_t)B * Cin * H * W, ks, stream);
}

hipError_t launch_conv3x3s2_train_wgrad(const float* x, const float* dy, float* dw, int B, int Cin, int Cout, int H, int W,
                                        float* workspace, hipStream_t stream) {
    S2Geom g;
    if (!conv3x3s2_train_supported(B, Cin, Cout, H, W) || !s2_geom(B, H, W, g)) return hipErrorInvalidValue;
    int sl, tps;
    s2_plan_wgrad(g, Cin, Cout, sl, tps);
    const size_t lds = std::max((size_t)64 * 65 + (size_t)32 * g.WIN, (size_t)9 * 16 * 64) * sizeof(float);
    hipError_t e = s2_lds(k_conv3s2_train_wgrad, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_conv3s2_train_wgrad, dim3(Cin / 32, Cout / 64, sl), dim3(256), lds, stream, x, dy, sl > 1 ? workspace : dw,
                       g, Cin, Cout, tps);
    e = hipGetLastError();
    if (e != hipSuccess || sl == 1) return e;
    return s2_sum(workspace, dw, (size_t)Cin * Cout * 9, sl, stream);
}

}  // namespace danra
