// Training-step kernels for the two 8x8, stride-2, padding-3 convolutions of the encoder's stem (conv1 on the 1-4 input
// planes, conv2 64 -> 64), on contiguous NCHW fp32 tensors and the [Cout][Cin][8][8] weights as torch holds them; all
// products on v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains), every sum over workgroups through slabs added in slab order
// (bit-reproducible, no atomics).
//
//   forward        y[b][co][oy][ox]  = sum_{ci,kh,kw} w[co][ci][kh][kw] x[b][ci][2 oy - 3 + kh][2 ox - 3 + kw]
//   data gradient  dx[b][ci][iy][ix] = sum_{co} sum_{kh,kw of the parity of (iy + 3, ix + 3)}
//                                      w[co][ci][kh][kw] dy[b][co][(iy + 3 - kh) / 2][(ix + 3 - kw) / 2]
//   weight grad.   dw[co][ci][kh][kw] = sum_{b,oy,ox} dy[b][co][oy][ox] x[b][ci][2 oy - 3 + kh][2 ox - 3 + kw]
//
// A "tile" is 64 consecutive pixels of the HALF-resolution map (H/2 x W/2: the forward output, and the grid of 2x2 cells
// of the input): 64 / (W/2) whole rows of one image.  What a tile needs of the other tensor is staged in LDS as a window
// with a zero halo, so that every tap of every pixel is a fixed offset from the pixel's own position and the product
// loops hold no bounds check.
//
// The data gradient treats the four pixels of an input 2x2 cell as four outputs of the cell: pixel parity (py, px) uses
// the 4x4 taps kh = 2a + (1 - py), kw = 2b + (1 - px) and reads dy at row u + (py ? 2 - a : 1 - a) of the cell's row u
// (columns alike), so one staged window of dy and one staged slice of the weights serve all four.
#include "danra_kernels.h"

namespace danra {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int C8_KC = 4;                 // channels of the reduction staged per chunk (forward: ci; data gradient: co)
constexpr int C8_LDW = C8_KC * 64 + 1;   // LDS row of the staged weights (odd: rows walk the banks)
constexpr int C8_NP = 5;                 // window positions per thread at most: (2 + 6) x (128 + 6) = 1072 <= 5 x 256

struct C8Geom {
    int B, H, W;         // the full-resolution tensor (x, dx)
    int Ho, Wo, logWo;   // the half-resolution tensor (y, dy)
    int RH;              // half-resolution rows per tile
    int tiles_per_img, n_tiles;
};

bool c8_geom(int B, int H, int W, C8Geom& g) {
    if (B <= 0 || H < 2 || W < 2 || (H & 1) || (W & 1)) return false;
    g.B = B; g.H = H; g.W = W;
    g.Ho = H / 2; g.Wo = W / 2;
    if (g.Wo < 8 || g.Wo > 64 || (g.Wo & (g.Wo - 1))) return false;
    g.logWo = 0;
    while ((1 << g.logWo) < g.Wo) ++g.logWo;
    g.RH = 64 / g.Wo;
    if (g.Ho % g.RH) return false;
    g.tiles_per_img = g.Ho / g.RH;
    g.n_tiles = B * g.tiles_per_img;
    return true;
}

__device__ __forceinline__ void zero16(f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// Forward.  Workgroup: one tile x 32 output channels; chunks of 4 input channels: their window (2 RH + 6 rows of W + 6)
// and the 32 x (4 x 64) weights in LDS, the next chunk's loads in flight while this one is multiplied.  Wave w: pixel half
// w & 1 (one 32x32 accumulator: rows = output channels, columns = pixels) and kernel rows 4 (w >> 1) .. + 3; the lane half
// picks the column of a pair (kw = 2 j + half).  The two halves of the kernel rows are added through LDS at the end.
__global__ __launch_bounds__(256) void k_conv8_train_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                         float* __restrict__ y, const C8Geom g, int Cin, int Cout) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int WW = g.W + 6, WIN = (2 * g.RH + 6) * WW;
    float* wts = smem;                  // [32][C8_LDW]
    float* win = smem + 32 * C8_LDW;    // [C8_KC][WIN]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pt = wave & 1, khalf = wave >> 1;
    const int l32 = lane & 31, kk = lane >> 5;
    const int tile = blockIdx.x, n0 = blockIdx.y * 32;
    const int b = tile / g.tiles_per_img, oy0 = (tile - b * g.tiles_per_img) * g.RH;
    const int hw = g.H * g.W;
    const int nchunks = (Cin + C8_KC - 1) / C8_KC;

    // what this thread stages per chunk: window positions tid + 256 i (source offset inside a channel; -1: halo, -2: none)
    int wsrc[C8_NP];
#pragma unroll
    for (int i = 0; i < C8_NP; ++i) {
        const int r = tid + 256 * i;
        wsrc[i] = -2;
        if (r < WIN) {
            const int row = r / WW, col = r - row * WW;
            const int iy = 2 * oy0 - 3 + row, ix = col - 3;
            wsrc[i] = ((unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W) ? iy * g.W + ix : -1;
        }
    }
    // weights: 32 rows x 64 float4 per chunk; thread: rows (tid >> 6) + 4 i, float4 tid & 63
    const int wrow = tid >> 6, wf4 = tid & 63;
    float xreg[C8_KC][C8_NP];
    float4 areg[8];
    auto load_chunk = [&](int c) {
        const int cnt = min(C8_KC, Cin - c * C8_KC);
#pragma unroll
        for (int cl = 0; cl < C8_KC; ++cl) {
            const float* src = x + ((size_t)b * Cin + min(c * C8_KC + cl, Cin - 1)) * hw;
#pragma unroll
            for (int i = 0; i < C8_NP; ++i) xreg[cl][i] = src[max(wsrc[i], 0)];   // unconditional (clamped), zeroed when stored
        }
        const bool have = wf4 * 4 < cnt * 64;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float* src = w + ((size_t)(n0 + wrow + 4 * i) * Cin + c * C8_KC) * 64 + (have ? wf4 * 4 : 0);
            areg[i] = *reinterpret_cast<const float4*>(src);
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int cl = 0; cl < C8_KC; ++cl)
#pragma unroll
            for (int i = 0; i < C8_NP; ++i)
                if (wsrc[i] > -2) win[cl * WIN + tid + 256 * i] = wsrc[i] >= 0 ? xreg[cl][i] : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float* d = wts + (wrow + 4 * i) * C8_LDW + wf4 * 4;
            d[0] = areg[i].x; d[1] = areg[i].y; d[2] = areg[i].z; d[3] = areg[i].w;
        }
    };

    const int p = pt * 32 + l32, oyl = p >> g.logWo, ox = p & (g.Wo - 1);
    const float* arow = wts + l32 * C8_LDW + khalf * 32 + kk;
    const float* bcol = win + (2 * oyl + khalf * 4) * WW + 2 * ox + kk;
    f32x16 acc;
    zero16(acc);

    load_chunk(0);
    for (int c = 0; c < nchunks; ++c) {
        __syncthreads();  // the previous chunk has been multiplied
        store_chunk();
        __syncthreads();
        if (c + 1 < nchunks) load_chunk(c + 1);
        const int cnt = min(C8_KC, Cin - c * C8_KC);
        for (int cl = 0; cl < cnt; ++cl) {
#pragma unroll
            for (int khh = 0; khh < 4; ++khh)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[cl * 64 + khh * 8 + 2 * j], bcol[cl * WIN + khh * WW + 2 * j],
                                                               acc, 0, 0, 0);
        }
    }

    __syncthreads();
    float* red = smem;  // 2 x 16 x 64 floats
    if (khalf == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(pt * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (khalf == 1) return;
    const int hwo = g.Ho * g.Wo;
    float* dst = y + ((size_t)b * Cout + n0) * hwo + oy0 * g.Wo + p;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = (r >> 2) * 8 + kk * 4 + (r & 3);
        dst[(size_t)n * hwo] = acc[r] + red[(pt * 16 + r) * 64 + lane];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Data gradient.  Workgroup: one tile of 64 input 2x2 cells x 32 input channels; chunks of 4 output channels: their dy
// window (RH + 4 rows of W/2 + 4) and the 32 x (4 x 64) weights in LDS.  Wave w: cell half w & 1, four 32x32 accumulators
// (one per pixel of the cell: rows = input channels, columns = cells) and the tap rows a = 2 (w >> 1), + 1 of each parity;
// the lane half picks the tap column of a pair (b = 2 j + half).
__global__ __launch_bounds__(256) void k_conv8_train_dgrad(const float* __restrict__ dy, const float* __restrict__ w,
                                                           float* __restrict__ dx, const C8Geom g, int Cin, int Cout) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int WWd = g.Wo + 4, WIND = (g.RH + 4) * WWd;   // at most 5 x 68 = 340 <= 2 x 256
    float* wts = smem;                  // [32 ci][C8_LDW]: (co of the chunk, kh, kw)
    float* dwin = smem + 32 * C8_LDW;   // [C8_KC][WIND]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pt = wave & 1, ah = wave >> 1;
    const int l32 = lane & 31, kk = lane >> 5;
    const int tile = blockIdx.x, n0 = blockIdx.y * 32;
    const int b = tile / g.tiles_per_img, u0 = (tile - b * g.tiles_per_img) * g.RH;
    const int hwo = g.Ho * g.Wo;
    const int nchunks = Cout / C8_KC;

    int wsrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = tid + 256 * i;
        wsrc[i] = -2;
        if (r < WIND) {
            const int row = r / WWd, col = r - row * WWd;
            const int oy = u0 - 2 + row, ox = col - 2;
            wsrc[i] = ((unsigned)oy < (unsigned)g.Ho && (unsigned)ox < (unsigned)g.Wo) ? oy * g.Wo + ox : -1;
        }
    }
    float dreg[C8_KC][2];
    float4 areg[8];
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int cl = 0; cl < C8_KC; ++cl) {
            const float* src = dy + ((size_t)b * Cout + c * C8_KC + cl) * hwo;
#pragma unroll
            for (int i = 0; i < 2; ++i) dreg[cl][i] = src[max(wsrc[i], 0)];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {   // float4 e of 2048: (co of the chunk, ci, 16 float4 of the 64 taps)
            const int e = tid + 256 * i, cl = e >> 9, ci = (e >> 4) & 31, f4 = e & 15;
            areg[i] = *reinterpret_cast<const float4*>(w + ((size_t)(c * C8_KC + cl) * Cin + n0 + ci) * 64 + f4 * 4);
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int cl = 0; cl < C8_KC; ++cl)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (wsrc[i] > -2) dwin[cl * WIND + tid + 256 * i] = wsrc[i] >= 0 ? dreg[cl][i] : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + 256 * i, cl = e >> 9, ci = (e >> 4) & 31, f4 = e & 15;
            float* d = wts + ci * C8_LDW + cl * 64 + f4 * 4;
            d[0] = areg[i].x; d[1] = areg[i].y; d[2] = areg[i].z; d[3] = areg[i].w;
        }
    };

    const int p = pt * 32 + l32, ul = p >> g.logWo, vl = p & (g.Wo - 1);
    const float* arow = wts + l32 * C8_LDW + 2 * kk;                 // + co * 64 + kh * 8 + 4 j + (1 - px)
    const float* bcol = dwin + (ul + 2) * WWd + vl + 2 - kk;         // + co * WIND + rowoff * WWd + (px ? 2 : 1) - 2 j
    f32x16 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) zero16(acc[q]);

    load_chunk(0);
    for (int c = 0; c < nchunks; ++c) {
        __syncthreads();
        store_chunk();
        __syncthreads();
        if (c + 1 < nchunks) load_chunk(c + 1);
#pragma unroll
        for (int cl = 0; cl < C8_KC; ++cl) {
#pragma unroll
            for (int aa = 0; aa < 2; ++aa) {
                const int a = 2 * ah + aa;   // (wave-uniform)
#pragma unroll
                for (int py = 0; py < 2; ++py) {
                    const int kh = 2 * a + (1 - py), rowoff = py ? 2 - a : 1 - a;
#pragma unroll
                    for (int px = 0; px < 2; ++px)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[py * 2 + px] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                arow[cl * 64 + kh * 8 + 4 * j + (1 - px)], bcol[cl * WIND + rowoff * WWd + (px ? 2 : 1) - 2 * j],
                                acc[py * 2 + px], 0, 0, 0);
                }
            }
        }
    }

    __syncthreads();
    float* red = smem;  // [2 cell halves][4 pixels][16][64 lanes] = 8192 floats <= the weight stage
    if (ah == 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((pt * 4 + q) * 16 + r) * 64 + lane] = acc[q][r];
    }
    __syncthreads();
    if (ah == 1) return;
    const int hw = g.H * g.W;
    float* dst = dx + ((size_t)b * Cin + n0) * hw + (size_t)(2 * (u0 + ul)) * g.W + 2 * vl;
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
// Weight gradient.  Rows of the product = 64 output channels (two wave halves), reduction = pixels (pairs; the lane half
// picks the pixel of a pair), eight 32x32 accumulators per wave:
//   SMALL = false (Cin a multiple of 32): columns = 32 input channels, accumulators = the 8 kernel columns kw of ONE
//     kernel row kh (blockIdx.x = (ci tile, kh)); the window holds the RH input rows 2 oy - 3 + kh of the tile.
//   SMALL = true (Cin <= 4): columns = (ci, kw) (8 per input channel), accumulators = the 8 kernel rows kh; the window
//     holds all 2 RH + 6 rows, its channel stride is 8 mod 32 (the four channels' columns fall on different banks).
// blockIdx.z: a range of tiles, result to slab z.  Wave w: output-channel half w & 1, half w >> 1 of a tile's 32 pixel pairs.
template <bool SMALL>
__global__ __launch_bounds__(256) void k_conv8_train_wgrad(const float* __restrict__ x, const float* __restrict__ dy,
                                                           float* __restrict__ dw, const C8Geom g, int Cin, int Cout,
                                                           int tiles_per_split) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int LDY = 65;
    const int WW = g.W + 6;
    const int WINK = (SMALL ? 2 * g.RH + 6 : g.RH) * WW;
    const int WINP = SMALL ? ((WINK + 23) / 32) * 32 + 8 : (WINK | 1);
    float* dyt = smem;               // [64 co][65]
    float* win = smem + 64 * LDY;    // [32 or 4 ci][WINP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cot = wave & 1, ph = wave >> 1;
    const int l32 = lane & 31, kk = lane >> 5;
    const int ci0 = SMALL ? 0 : (blockIdx.x >> 3) * 32, kh_wg = SMALL ? 0 : (blockIdx.x & 7);
    const int co0 = blockIdx.y * 64;
    const int t_begin = blockIdx.z * tiles_per_split, t_end = min(g.n_tiles, t_begin + tiles_per_split);
    const int hw = g.H * g.W, hwo = g.Ho * g.Wo;

    f32x16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) zero16(acc[t]);

    for (int tile = t_begin; tile < t_end; ++tile) {
        const int b = tile / g.tiles_per_img, oy0 = (tile - b * g.tiles_per_img) * g.RH;
        __syncthreads();  // the previous tile has been multiplied
        {   // dy tile: 64 channels x 64 pixels; a thread keeps one pixel and walks 16 channels
            const int p = tid & 63, cq = tid >> 6;
            const float* src = dy + ((size_t)b * Cout + co0 + cq) * hwo + oy0 * g.Wo + p;
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = src[(size_t)(4 * i) * hwo];
#pragma unroll
            for (int i = 0; i < 16; ++i) dyt[(cq + 4 * i) * LDY + p] = v[i];
        }
        if (!SMALL) {
            if (tid < WINK) {   // a thread keeps one window position and walks the 32 channels
                const int row = tid / WW, col = tid - row * WW;
                const int iy = 2 * (oy0 + row) - 3 + kh_wg, ix = col - 3;
                const bool ok = (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
                const float* src = x + ((size_t)b * Cin + ci0) * hw + (ok ? iy * g.W + ix : 0);
                float v[32];
#pragma unroll
                for (int c = 0; c < 32; ++c) v[c] = src[(size_t)c * hw];
#pragma unroll
                for (int c = 0; c < 32; ++c) win[c * WINP + tid] = ok ? v[c] : 0.f;
            }
        } else {
            float v[4][C8_NP];
            int src[C8_NP];
#pragma unroll
            for (int i = 0; i < C8_NP; ++i) {
                const int r = tid + 256 * i;
                src[i] = -2;
                if (r < WINK) {
                    const int row = r / WW, col = r - row * WW;
                    const int iy = 2 * oy0 - 3 + row, ix = col - 3;
                    src[i] = ((unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W) ? iy * g.W + ix : -1;
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float* base = x + ((size_t)b * Cin + min(c, Cin - 1)) * hw;
#pragma unroll
                for (int i = 0; i < C8_NP; ++i) v[c][i] = base[max(src[i], 0)];
            }
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int i = 0; i < C8_NP; ++i)
                    if (src[i] > -2) win[c * WINP + tid + 256 * i] = (src[i] >= 0 && c < Cin) ? v[c][i] : 0.f;
        }
        __syncthreads();
        const float* arow = dyt + (cot * 32 + l32) * LDY + kk;
        const float* bcol = SMALL ? win + (l32 >> 3) * WINP + (l32 & 7) + 2 * kk : win + l32 * WINP + 2 * kk;
#pragma unroll 2
        for (int s = 0; s < 16; ++s) {
            const int p = 2 * (ph * 16 + s);
            const int oyl = p >> g.logWo, ox = p & (g.Wo - 1);
            const float a = arow[p];
            if (SMALL) {
                const float* bp = bcol + 2 * oyl * WW + 2 * ox;
#pragma unroll
                for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[t * WW], acc[t], 0, 0, 0);
            } else {
                const float* bp = bcol + oyl * WW + 2 * ox;
#pragma unroll
                for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[t], acc[t], 0, 0, 0);
            }
        }
    }

    // the two pixel halves are added in a fixed order through LDS, one output-channel half at a time
    float* dst = dw + (size_t)blockIdx.z * ((size_t)Cout * Cin * 64);
    float* red = smem;  // [8][16 registers][64 lanes]
    for (int half = 0; half < 2; ++half) {
        __syncthreads();
        if (cot == half && ph == 1) {
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(t * 16 + r) * 64 + lane] = acc[t][r];
        }
        __syncthreads();
        if (cot == half && ph == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + half * 32 + (r >> 2) * 8 + kk * 4 + (r & 3);
                if (SMALL) {   // lane = (ci, kw), accumulator = kh
                    const int ci = l32 >> 3, kw = l32 & 7;
                    if (ci < Cin) {
                        float* o = dst + ((size_t)co * Cin + ci) * 64 + kw;
#pragma unroll
                        for (int t = 0; t < 8; ++t) o[t * 8] = acc[t][r] + red[(t * 16 + r) * 64 + lane];
                    }
                } else {       // lane = ci, accumulator = kw: 32 consecutive bytes
                    float* o = dst + ((size_t)co * Cin + ci0 + l32) * 64 + kh_wg * 8;
                    float4 v0, v1;
                    v0.x = acc[0][r] + red[(0 * 16 + r) * 64 + lane];
                    v0.y = acc[1][r] + red[(1 * 16 + r) * 64 + lane];
                    v0.z = acc[2][r] + red[(2 * 16 + r) * 64 + lane];
                    v0.w = acc[3][r] + red[(3 * 16 + r) * 64 + lane];
                    v1.x = acc[4][r] + red[(4 * 16 + r) * 64 + lane];
                    v1.y = acc[5][r] + red[(5 * 16 + r) * 64 + lane];
                    v1.z = acc[6][r] + red[(6 * 16 + r) * 64 + lane];
                    v1.w = acc[7][r] + red[(7 * 16 + r) * 64 + lane];
                    *reinterpret_cast<float4*>(o) = v0;
                    *reinterpret_cast<float4*>(o + 4) = v1;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// host side
namespace {

bool c8_small(int Cin) { return Cin <= 4; }

void c8_plan_wgrad(const C8Geom& g, int Cin, int Cout, int& slices, int& tps) {
    const long base = c8_small(Cin) ? Cout / 64 : (long)(Cin / 32) * 8 * (Cout / 64);
    long s = (512 + base - 1) / base;
    if (s > g.n_tiles) s = g.n_tiles;
    if (s > 256) s = 256;
    if (s < 1) s = 1;
    tps = (int)((g.n_tiles + s - 1) / s);
    slices = (g.n_tiles + tps - 1) / tps;
}

template <typename K>
hipError_t c8_lds(K kernel, size_t bytes) {
    if (bytes <= 48 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace

// x [B][Cin][H][W] (H, W even; W/2 a power of two in 8..64; H/2 a multiple of 64 / (W/2)), w [Cout][Cin][8][8], Cout a
// multiple of 64, Cin a multiple of 32 or at most 4
bool conv8s2_train_supported(int B, int Cin, int Cout, int H, int W) {
    C8Geom g;
    if (!c8_geom(B, H, W, g)) return false;
    if (Cin <= 0 || Cout <= 0 || Cout % 64 || !(c8_small(Cin) || Cin % 32 == 0)) return false;
    const long long big = (long long)B * (Cin > Cout ? Cin : Cout) * H * W;
    return big < (1LL << 31) && (long long)Cin * Cout * 64 < (1LL << 31) && g.n_tiles <= 65535 * 16;
}

// the data gradient exists for the wide form only (the gradient with respect to the network's input planes is never asked for)
bool conv8s2_train_dgrad_supported(int B, int Cin, int Cout, int H, int W) {
    return conv8s2_train_supported(B, Cin, Cout, H, W) && Cin % 32 == 0;
}

size_t conv8s2_train_workspace_floats(int B, int Cin, int Cout, int H, int W) {
    C8Geom g;
    if (!c8_geom(B, H, W, g)) return 0;
    int sl, tps;
    c8_plan_wgrad(g, Cin, Cout, sl, tps);
    return sl > 1 ? (size_t)sl * Cin * Cout * 64 : 4;
}

hipError_t launch_conv8s2_train_fwd(const float* x, const float* w, float* y, int B, int Cin, int Cout, int H, int W,
                                    hipStream_t stream) {
    C8Geom g;
    if (!conv8s2_train_supported(B, Cin, Cout, H, W) || !c8_geom(B, H, W, g)) return hipErrorInvalidValue;
    const size_t lds = (32 * C8_LDW + (size_t)C8_KC * (2 * g.RH + 6) * (g.W + 6)) * sizeof(float);
    hipError_t e = c8_lds(k_conv8_train_fwd, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_conv8_train_fwd, dim3(g.n_tiles, Cout / 32), dim3(256), lds, stream, x, w, y, g, Cin, Cout);
    return hipGetLastError();
}

hipError_t launch_conv8s2_train_dgrad(const float* dy, const float* w, float* dx, int B, int Cin, int Cout, int H, int W,
                                      hipStream_t stream) {
    C8Geom g;
    if (!conv8s2_train_dgrad_supported(B, Cin, Cout, H, W) || !c8_geom(B, H, W, g)) return hipErrorInvalidValue;
    const size_t lds = (32 * C8_LDW + (size_t)C8_KC * (g.RH + 4) * (g.Wo + 4)) * sizeof(float);
    hipError_t e = c8_lds(k_conv8_train_dgrad, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_conv8_train_dgrad, dim3(g.n_tiles, Cin / 32), dim3(256), lds, stream, dy, w, dx, g, Cin, Cout);
    return hipGetLastError();
}

hipError_t launch_conv8s2_train_wgrad(const float* x, const float* dy, float* dw, int B, int Cin, int Cout, int H, int W,
                                      float* workspace, hipStream_t stream) {
    C8Geom g;
    if (!conv8s2_train_supported(B, Cin, Cout, H, W) || !c8_geom(B, H, W, g)) return hipErrorInvalidValue;
    int sl, tps;
    c8_plan_wgrad(g, Cin, Cout, sl, tps);
    const bool small = c8_small(Cin);
    const int WW = g.W + 6;
    const int wink = (small ? 2 * g.RH + 6 : g.RH) * WW;
    const int winp = small ? ((wink + 23) / 32) * 32 + 8 : (wink | 1);
    const size_t loop_floats = 64 * 65 + (size_t)(small ? 4 : 32) * winp;
    const size_t lds = std::max(loop_floats, (size_t)8 * 16 * 64) * sizeof(float);
    float* dst = sl > 1 ? workspace : dw;
    hipError_t e;
    if (small) {
        e = c8_lds(k_conv8_train_wgrad<true>, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_conv8_train_wgrad<true>, dim3(1, Cout / 64, sl), dim3(256), lds, stream, x, dy, dst, g, Cin, Cout, tps);
    } else {
        e = c8_lds(k_conv8_train_wgrad<false>, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_conv8_train_wgrad<false>, dim3(Cin / 32 * 8, Cout / 64, sl), dim3(256), lds, stream, x, dy, dst, g, Cin,
                           Cout, tps);
    }
    e = hipGetLastError();
    if (e != hipSuccess || sl == 1) return e;
    const size_t count = (size_t)Cin * Cout * 64;
    hipLaunchKernelGGL(k_sum_slices, dim3((unsigned)((count / 4 + 31) / 32)), dim3(256), 0, stream, dst, dw, (int)(count / 4), sl);
    return hipGetLastError();
}

}  // namespace danra
