// Training-step kernels for the 3x3, stride-1, padding-1 convolutions of the network (the sixteen BasicBlock
// convolutions of stride 1 and the decoder blocks' convolutions), on contiguous NCHW fp32 tensors and the
// [Cout][Cin][3][3] weights exactly as the differentiable torch composition holds them: no layout conversion and no
// weight repack around a call.  Three products, all on v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains):
//
//   forward        y[b][co][p]  = sum_{ci,tap} w[co][ci][tap] x[b][ci][p + tap]
//   data gradient  dx[b][ci][p] = sum_{co,tap} w[co][ci][8 - tap] dy[b][co][p + tap]     (same kernel, weights re-indexed
//                                                                                          while they are staged)
//   weight grad.   dw[co][ci][tap] = sum_{b,p} dy[b][co][p] x[b][ci][p + tap]
//
// Geometry shared by the three: a "pixel tile" is 64 consecutive pixels in (image, row, column) order: 64 / W whole
// rows of one image when the image has at least 64 pixels, else 64 / (H W) whole images.  The input channels a
// workgroup needs for its tile are staged in LDS as a WINDOW with a zero halo: per channel, per image of the tile,
// (rows + 2) x (W + 2) floats, so that tap (ty, tx) of a pixel is the window value at a fixed offset from the pixel's
// own position and no bounds check exists in the product loop.
//
// Weights are the MFMA A operand (rows = output channels), pixels the B operand: the 32x32 accumulator of a lane holds
// 16 output channels of ONE pixel, and for a fixed register the 32 lanes of a half-wave are 32 consecutive pixels: NCHW
// stores are 128-byte runs.  Sums that run over workgroups (split reduction channels, pixel slices of the weight
// gradient) go to slabs that k_sum_slices adds in slab order: nothing is accumulated with atomics, results are
// bit-reproducible.
#include "danra_kernels.h"

namespace danra {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int CT_PX = 64;    // pixels per tile
constexpr int CT_KC = 16;    // reduction channels per chunk (forward / data gradient)
constexpr int CT_NO = 32;    // output channels per workgroup (forward / data gradient)
constexpr int CT_LDW = CT_KC * 9 + 1;  // LDS row of the staged weights: 16 channels x 9 taps + 1 (odd: conflict-free column reads)
constexpr int CT_MAXWIN = 256;  // largest window per channel: 16 images of 2x2 -> 16 x 4 x 4

struct ConvTrainGeom {
    int B, H, W, logW;
    int NI, RH;        // images and rows per image of a pixel tile
    int WIN;           // window floats per channel = NI * (RH + 2) * (W + 2)
    int tiles_per_img; // NI == 1: H / RH
    int n_tiles;
    int M;             // B * H * W
};

bool make_geom(int B, int H, int W, ConvTrainGeom& g) {
    if (B <= 0 || H <= 0 || W < 2 || W > 64 || (W & (W - 1)) || CT_PX % W) return false;
    g.B = B; g.H = H; g.W = W;
    g.logW = 0;
    while ((1 << g.logW) < W) ++g.logW;
    if (H * W >= CT_PX) {
        g.NI = 1;
        g.RH = CT_PX / W;
        if (H % g.RH) return false;
        g.tiles_per_img = H / g.RH;
        g.n_tiles = B * g.tiles_per_img;
    } else {
        if (CT_PX % (H * W)) return false;
        g.NI = CT_PX / (H * W);
        g.RH = H;
        g.tiles_per_img = 0;
        g.n_tiles = (B + g.NI - 1) / g.NI;
    }
    g.WIN = g.NI * (g.RH + 2) * (W + 2);
    g.M = B * H * W;
    return g.WIN <= CT_MAXWIN;
}

// Window position of pixel p (0..63) of a tile: row-of-tile R = p / W; NI == 1: window row R + 1, else image R / H,
// row R % H of that image.
__device__ __forceinline__ int window_pos(const ConvTrainGeom& g, int p) {
    const int R = p >> g.logW, x = p & (g.W - 1);
    int wrow;
    if (g.NI == 1) {
        wrow = R + 1;
    } else {
        const int il = R / g.RH;
        wrow = il * (g.RH + 2) + (R - il * g.RH) + 1;
    }
    return wrow * (g.W + 2) + x + 1;
}

// (image, row, column) of window element r (0..WIN-1) of tile `tile`; false when it lies outside every image (halo).
__device__ __forceinline__ bool window_source(const ConvTrainGeom& g, int tile, int r, int& b, int& y, int& x) {
    const int wrow = r / (g.W + 2);
    x = r - wrow * (g.W + 2) - 1;
    const int il = wrow / (g.RH + 2);
    const int yy = wrow - il * (g.RH + 2);
    if (g.NI == 1) {
        b = tile / g.tiles_per_img;
        y = (tile - b * g.tiles_per_img) * g.RH + yy - 1;
    } else {
        b = tile * g.NI + il;
        y = yy - 1;
    }
    return b < g.B && (unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W;
}

// (image, pixel offset inside the image) of pixel p of a tile; false past the last image.
__device__ __forceinline__ bool tile_pixel(const ConvTrainGeom& g, int tile, int p, int& b, int& off) {
    const int hw = g.H * g.W;
    if (g.NI == 1) {
        b = tile / g.tiles_per_img;
        off = (tile - b * g.tiles_per_img) * CT_PX + p;
    } else {
        const int il = p / hw;
        b = tile * g.NI + il;
        off = p - il * hw;
    }
    return b < g.B;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// Forward and data gradient.  out[b][n][p] = sum over reduction channels c and taps of A[n][(c, tap)] in[b][c][p + tap],
//   forward (DGRAD = false): A[n][(c, tap)] = w[n][c][tap]         (w is [Nout][Cred][9])
//   data gradient (true)   : A[n][(c, tap)] = w[c][n][8 - tap]     (w is [Cred][Nout][9]: the layer's own weight)
// Workgroup: one pixel tile x 32 output channels x a range of 16-channel chunks (blockIdx.z: split reduction, slab z).
// Wave w: pixel half w & 1 (one 32x32 accumulator), and half w >> 1 of a chunk's 72 two-deep steps (step = (channel
// pair, tap); the lane half picks the channel of the pair); the two halves are added through LDS at the end.
template <bool DGRAD>
__global__ __launch_bounds__(256) void k_conv3x3_train(const float* __restrict__ in, const float* __restrict__ w,
                                                       float* __restrict__ out, const ConvTrainGeom g, int Cred, int Nout,
                                                       int chunks_per_split) {
    __shared__ float win[CT_KC * CT_MAXWIN];
    __shared__ float wts[CT_NO * CT_LDW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pt = wave & 1, kh = wave >> 1;
    const int l32 = lane & 31, kk = lane >> 5;
    const int tile = blockIdx.x, n0 = blockIdx.y * CT_NO;
    const int nchunks = Cred / CT_KC;
    const int c_begin = blockIdx.z * chunks_per_split, c_end = min(nchunks, c_begin + chunks_per_split);
    const int hw = g.H * g.W;

    // ---- what this thread stages per chunk.  Window: the thread owns ONE window position (tid % WP, WP = 128 or 256
    //      >= WIN) and walks the chunk's channels with stride 256 / WP (offset -1 = halo or past the tensor: zero); one
    //      position decode per thread.  Weights: 18 values.
    const int wp_log = g.WIN <= 128 ? 7 : 8;
    const int wr = tid & ((1 << wp_log) - 1), wc0 = tid >> wp_log, wcs = 256 >> wp_log;  // position, first channel, stride
    constexpr int NE = CT_KC;  // loads per thread when one thread group walks all channels (WP = 256)
    int wsrc = -1;
    if (wr < g.WIN) {
        int b, y, x;
        if (window_source(g, tile, wr, b, y, x)) wsrc = ((b * Cred + wc0) * g.H + y) * g.W + x;
    }
    const int n_mine = CT_KC / wcs;  // channels this thread loads per chunk: 16 or 8
    constexpr int NA = CT_NO * CT_KC * 9 / 256;  // 18
    int asrc[NA], adst[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int e = tid + 256 * i;
        if (!DGRAD) {
            const int n = e / (CT_KC * 9), r = e - n * (CT_KC * 9);  // r = (channel of the chunk, tap): consecutive floats of w
            asrc[i] = (n0 + n) * Cred * 9 + r;
            adst[i] = n * CT_LDW + r;
        } else {
            const int c = e / 288, rem = e - c * 288;  // for one reduction channel: 32 output channels x 9 taps, consecutive
            const int n = rem / 9, t = rem - n * 9;
            asrc[i] = (c * Nout + n0 + n) * 9 + t;
            adst[i] = n * CT_LDW + c * 9 + (8 - t);
        }
    }
    float wreg[NE], areg[NA];
    auto load_chunk = [&](int c) {
        const float* src = in + c * CT_KC * hw + max(wsrc, 0);  // unconditional (clamped) loads, zeroed when stored
        const int w_off = DGRAD ? c * CT_KC * Nout * 9 : c * CT_KC * 9;
#pragma unroll
        for (int i = 0; i < NE; ++i)
            if (i < n_mine) wreg[i] = src[i * wcs * hw];
#pragma unroll
        for (int i = 0; i < NA; ++i) areg[i] = w[asrc[i] + w_off];
    };
    auto store_chunk = [&]() {
        if (wr < g.WIN) {
#pragma unroll
            for (int i = 0; i < NE; ++i)
                if (i < n_mine) win[(wc0 + i * wcs) * g.WIN + wr] = wsrc >= 0 ? wreg[i] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) wts[adst[i]] = areg[i];
    };

    // ---- product loop
    const int pos = window_pos(g, pt * 32 + l32);
    int tapoff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) tapoff[t] = (t / 3 - 1) * (g.W + 2) + (t % 3 - 1);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* arow = wts + l32 * CT_LDW + kk * 9;
    const float* bcol = win + kk * g.WIN + pos;

    if (c_begin < c_end) load_chunk(c_begin);
    for (int c = c_begin; c < c_end; ++c) {
        __syncthreads();  // the previous chunk has been multiplied
        store_chunk();
        __syncthreads();
        if (c + 1 < c_end) load_chunk(c + 1);  // in flight while this chunk is multiplied
#pragma unroll
        for (int s = 0; s < CT_KC / 4 * 9; ++s) {
            const int pair = kh * (CT_KC / 4) + s / 9, t = s % 9;  // this wave's half of the (channel pair, tap) steps
            const float a = arow[pair * 18 + t];
            const float b = bcol[pair * 2 * g.WIN + tapoff[t]];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }

    // ---- add the two halves of the steps (fixed order) and store
    __syncthreads();
    float* red = win;  // 2 x 16 x 64 floats
    if (kh == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(pt * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (kh == 1) return;
    int b, off;
    const bool ok = tile_pixel(g, tile, pt * 32 + l32, b, off);
    float* dst = out + (size_t)blockIdx.z * ((size_t)g.M * Nout);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = n0 + (r >> 2) * 8 + kk * 4 + (r & 3);
        const float v = acc[r] + red[(pt * 16 + r) * 64 + lane];
        if (ok) dst[((size_t)b * Nout + n) * hw + off] = v;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Weight gradient.  dw[co][ci][tap] = sum over pixels of dy[.][co][p] x[.][ci][p + tap].
// Workgroup: 64 output channels x 32 input channels x a range of pixel tiles (blockIdx.z: slab z).  Wave w: output
// channel half w & 1, and half w >> 1 of a tile's 32 pixel pairs; nine 32x32 accumulators (one per tap: rows = co,
// columns = ci).  Per pixel pair one dy fragment serves nine matrix instructions.  The window rows are padded to an odd
// length (lanes differ in ci).
__global__ __launch_bounds__(256) void k_conv3x3_wgrad_train(const float* __restrict__ x, const float* __restrict__ dy,
                                                             float* __restrict__ dw, const ConvTrainGeom g, int Cin, int Cout,
                                                             int tiles_per_split) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int LDY = CT_PX + 1;
    const int WINP = g.WIN | 1;
    float* dyt = smem;                  // [64 co][65]
    float* win = smem + 64 * LDY;       // [32 ci][WINP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cot = wave & 1, kh = wave >> 1;
    const int l32 = lane & 31, kk = lane >> 5;
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 64;
    const int t_begin = blockIdx.z * tiles_per_split, t_end = min(g.n_tiles, t_begin + tiles_per_split);
    const int hw = g.H * g.W;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    int tapoff[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) tapoff[t] = (t / 3 - 1) * (g.W + 2) + (t % 3 - 1);

    for (int tile = t_begin; tile < t_end; ++tile) {
        __syncthreads();  // the previous tile has been multiplied
        {   // dy tile: 64 channels x 64 pixels.  A thread keeps one pixel and walks 16 channels (channel stride H W)
            const int p = tid & 63, cq = tid >> 6;
            int b, off;
            const bool ok = tile_pixel(g, tile, p, b, off);
            const float* src = dy + ((size_t)(ok ? b : 0) * Cout + co0 + cq) * hw + off;
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = src[(size_t)(4 * i) * hw];  // unconditional: past the last image it re-reads image 0
#pragma unroll
            for (int i = 0; i < 16; ++i) dyt[(cq + 4 * i) * LDY + p] = ok ? v[i] : 0.f;
        }
        if (tid < g.WIN) {  // window: a thread keeps one window position and walks the 32 channels
            int b, y, xx;
            const bool ok = window_source(g, tile, tid, b, y, xx);
            const float* src = ok ? x + (((size_t)b * Cin + ci0) * g.H + y) * g.W + xx : x;
            float v[32];  // all 32 loads in flight before the first LDS write (an unroll by 8 paid the latency four times)
#pragma unroll
            for (int c = 0; c < 32; ++c) v[c] = src[(size_t)c * hw];
#pragma unroll
            for (int c = 0; c < 32; ++c) win[c * WINP + tid] = ok ? v[c] : 0.f;
        }
        __syncthreads();
        const float* arow = dyt + (cot * 32 + l32) * LDY + kk;
        const float* bcol = win + l32 * WINP + kk;
#pragma unroll 2
        for (int s = 0; s < 16; ++s) {
            const int p = 2 * (kh * 16 + s);
            const int pos = __builtin_amdgcn_readfirstlane(window_pos(g, p));
            const float a = arow[p];
#pragma unroll
            for (int t = 0; t < 9; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bcol[pos + tapoff[t]], acc[t], 0, 0, 0);
        }
    }

    // ---- the two pixel halves are added in a fixed order through LDS, register by register in lane order (conflict-free),
    //      one output-channel half at a time; the sums are stored from the registers: lane (ci, kk) writes the nine taps
    //      of its (co, ci) pairs, 36 consecutive bytes each (the stores of a pair merge in L2)
    float* dst = dw + (size_t)blockIdx.z * ((size_t)Cout * Cin * 9);
    float* red = smem;  // [9 taps][16 registers][64 lanes]
    for (int half = 0; half < 2; ++half) {
        __syncthreads();
        if (cot == half && kh == 1) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(t * 16 + r) * 64 + lane] = acc[t][r];
        }
        __syncthreads();
        if (cot == half && kh == 0) {
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

struct ConvTrainPlan {
    ConvTrainGeom g;
    int ksplit, chunks_per_split;  // forward / data gradient over Cred
};

// split the reduction channels until the grid has a few hundred workgroups (two chunks per workgroup at least)
void plan_split(const ConvTrainGeom& g, int Cred, int Nout, int& ksplit, int& cps) {
    const int nchunks = Cred / CT_KC;
    const long base = (long)g.n_tiles * (Nout / CT_NO);
    ksplit = 1;
    while (base * ksplit < 256 && nchunks / (ksplit * 2) >= 4 && ksplit < 16) ksplit *= 2;
    cps = (nchunks + ksplit - 1) / ksplit;
    ksplit = (nchunks + cps - 1) / cps;
}

void plan_wgrad(const ConvTrainGeom& g, int Cin, int Cout, int& slices, int& tps) {
    const long tiles_out = (long)(Cin / 32) * (Cout / 64);
    long s = tiles_out >= 128 ? 1 : (512 + tiles_out - 1) / tiles_out;  // (slabs of a 512 x 512 x 9 gradient are 9.4 MB each)
    if (s > g.n_tiles) s = g.n_tiles;
    if (s > 128) s = 128;
    if (s < 1) s = 1;
    tps = (int)((g.n_tiles + s - 1) / s);
    slices = (g.n_tiles + tps - 1) / tps;
}

hipError_t sum_slabs(const float* part, float* out, size_t count, int slices, hipStream_t stream) {
    hipLaunchKernelGGL(k_sum_slices, dim3((unsigned)((count / 4 + 31) / 32)), dim3(256), 0, stream, part, out, (int)(count / 4),
                       slices);
    return hipGetLastError();
}

}  // namespace

bool conv3x3_train_supported(int B, int Cin, int Cout, int H, int W) {
    ConvTrainGeom g;
    if (!make_geom(B, H, W, g)) return false;
    // forward reduces over Cin (chunks of 8) into Cout (tiles of 32); the data gradient the other way round; the weight
    // gradient works on 64 x 32 (Cout x Cin) tiles
    if (Cin % 32 || Cout % 64) return false;
    const long long big = (long long)B * (Cin > Cout ? Cin : Cout) * H * W;
    return big < (1LL << 31) && (long long)Cin * Cout * 9 < (1LL << 31);
}

size_t conv3x3_train_workspace_floats(int B, int Cin, int Cout, int H, int W) {
    ConvTrainGeom g;
    if (!make_geom(B, H, W, g)) return 0;
    int ks, cps, sl, tps;
    size_t need = 4;
    plan_split(g, Cin, Cout, ks, cps);
    if (ks > 1) need = std::max(need, (size_t)ks * g.M * Cout);
    plan_split(g, Cout, Cin, ks, cps);
    if (ks > 1) need = std::max(need, (size_t)ks * g.M * Cin);
    plan_wgrad(g, Cin, Cout, sl, tps);
    if (sl > 1) need = std::max(need, (size_t)sl * Cin * Cout * 9);
    return need;
}

// dgrad = false: in = x [B][Cin][H][W], out = y [B][Cout][H][W]; true: in = dy [B][Cout][H][W], out = dx [B][Cin][H][W]
hipError_t launch_conv3x3_train(bool dgrad, const float* in, const float* w, float* out, int B, int Cin, int Cout, int H,
                                int W, float* workspace, hipStream_t stream) {
    ConvTrainGeom g;
    if (!make_geom(B, H, W, g)) return hipErrorInvalidValue;
    const int Cred = dgrad ? Cout : Cin, Nout = dgrad ? Cin : Cout;
    int ks, cps;
    plan_split(g, Cred, Nout, ks, cps);
    float* dst = ks > 1 ? workspace : out;
    const dim3 grid(g.n_tiles, Nout / CT_NO, ks);
    if (dgrad) hipLaunchKernelGGL(k_conv3x3_train<true>, grid, dim3(256), 0, stream, in, w, dst, g, Cred, Nout, cps);
    else hipLaunchKernelGGL(k_conv3x3_train<false>, grid, dim3(256), 0, stream, in, w, dst, g, Cred, Nout, cps);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || ks == 1) return e;
    return sum_slabs(workspace, out, (size_t)g.M * Nout, ks, stream);
}

hipError_t launch_conv3x3_wgrad_train(const float* x, const float* dy, float* dw, int B, int Cin, int Cout, int H, int W,
                                      float* workspace, hipStream_t stream) {
    ConvTrainGeom g;
    if (!make_geom(B, H, W, g)) return hipErrorInvalidValue;
    int sl, tps;
    plan_wgrad(g, Cin, Cout, sl, tps);
    const size_t loop_floats = 64 * (CT_PX + 1) + 32 * (size_t)(g.WIN | 1);
    const size_t out_floats = 9 * 16 * 64;
    const size_t lds = std::max(loop_floats, out_floats) * sizeof(float);
    float* dst = sl > 1 ? workspace : dw;
    hipLaunchKernelGGL(k_conv3x3_wgrad_train, dim3(Cin / 32, Cout / 64, sl), dim3(256), lds, stream, x, dy, dst, g, Cin, Cout,
                       tps);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || sl == 1) return e;
    return sum_slabs(workspace, dw, (size_t)Cin * Cout * 9, sl, stream);
}


// ==================================================================================================================
// The two "pointwise" convolutions of the training step: the 1x1 stride-2 shortcut convolutions of the BasicBlocks that
// halve the map (torchvision's downsample[0]) and the kernel-2 stride-2 transposed convolutions of the decoder blocks.
// Both are plain matrix products over the pixels of the COARSE grid (the smaller of the layer's two maps), with the fine
// map addressed either at every second pixel (1x1 stride 2) or as four sub-positions (dy, dx) per coarse pixel that act
// as four "virtual channels" per real channel (transposed convolution: v = 4 * channel + 2 * dy + dx).  One kernel forms
// out[b][n][p] = sum_c A[n][c] in[b][c][p] for any combination of the two addressings and either orientation of the
// weight matrix, a second one the weight gradient dW[n][c] = sum_{b,p} Y[b][n][p] X[b][c][p].
namespace {

struct PwTensor {    // how a (virtual channel, coarse pixel) of one operand is addressed in its NCHW tensor
    int C;           // real channels
    int H, W;        // the tensor's own map
    int sub;         // 1: four virtual channels per real channel (pixel (2y + dy, 2x + dx) of coarse pixel (y, x))
    int stride;      // sub == 0: pixel (stride * y, stride * x)
};

__device__ __forceinline__ size_t pw_addr(const PwTensor& t, int b, int v, int y, int x) {
    if (t.sub) return (((size_t)b * t.C + (v >> 2)) * t.H + 2 * y + ((v >> 1) & 1)) * t.W + 2 * x + (v & 1);
    return (((size_t)b * t.C + v) * t.H + t.stride * y) * t.W + t.stride * x;
}

constexpr int PW_KC = 64;  // (virtual) reduction channels per chunk
constexpr int PW_LDA = PW_KC + 1;

}  // namespace

// out[b][n][p] (+ bias[real channel of n]) = sum_c w[n * sn + c * sc] in[b][c][p] over the coarse grid Hc x Wc.
// Workgroup: 64 coarse pixels (any 64 consecutive ones of the flattened (image, y, x) order) x 32 output (virtual)
// channels; waves as in k_conv3x3_train: pixel half x half of a chunk's 16 two-deep steps.
__global__ __launch_bounds__(256) void k_pw_train(const float* __restrict__ in, const float* __restrict__ w,
                                                  const float* __restrict__ bias, float* __restrict__ out, const PwTensor ti,
                                                  const PwTensor to, int B, int Hc, int Wc, int Cv, int Nv, int sn, int sc,
                                                  int chunks_per_split) {
    __shared__ float bt[PW_KC * 64];       // [c][pixel]
    __shared__ float at[32 * PW_LDA];      // [n][c]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pt = wave & 1, kh = wave >> 1;
    const int l32 = lane & 31, kk = lane >> 5;
    const int hw = Hc * Wc, M = B * hw;
    const int p0 = blockIdx.x * 64, n0 = blockIdx.y * 32;
    // this thread's pixel of the B tile (the same for its 8 channels) and of the output
    auto decode = [&](int p, int& b, int& y, int& x) {
        const int q = min(p, M - 1);
        b = q / hw;
        const int r = q - b * hw;
        y = r / Wc;
        x = r - y * Wc;
    };
    int sb, sy, sx;
    decode(p0 + (tid & 63), sb, sy, sx);
    constexpr int NB = PW_KC * 64 / 256, NA = 32 * PW_KC / 256;  // B and A values a thread stages per chunk
    const int c_first = tid >> 6;  // channels c_first + 4 i, i = 0..NB-1
    // A tile: coalesced along whichever index has stride 1
    int asrc[NA], adst[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int e = tid + 256 * i;
        const int n = sc == 1 ? e / PW_KC : e % 32, c = sc == 1 ? e % PW_KC : e / 32;
        asrc[i] = (n0 + n) * sn + c * sc;
        adst[i] = n * PW_LDA + c;
    }
    float breg[NB], areg[NA];
    auto load_chunk = [&](int c0) {
#pragma unroll
        for (int i = 0; i < NB; ++i) breg[i] = in[pw_addr(ti, sb, c0 + c_first + 4 * i, sy, sx)];
#pragma unroll
        for (int i = 0; i < NA; ++i) areg[i] = w[asrc[i] + c0 * sc];
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* arow = at + l32 * PW_LDA + kk;
    const float* bcol = bt + kk * 64 + pt * 32 + l32;
    // blockIdx.z: a range of chunks, result to slab z (plain output tensors only: the host sums the slabs)
    const int c_lo = blockIdx.z * chunks_per_split * PW_KC, c_hi = min(Cv, c_lo + chunks_per_split * PW_KC);
    out += (size_t)blockIdx.z * ((size_t)M * Nv);
    load_chunk(min(c_lo, Cv - PW_KC));
    for (int c0 = c_lo; c0 < c_hi; c0 += PW_KC) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NB; ++i) bt[(c_first + 4 * i) * 64 + (tid & 63)] = breg[i];
#pragma unroll
        for (int i = 0; i < NA; ++i) at[adst[i]] = areg[i];
        __syncthreads();
        if (c0 + PW_KC < c_hi) load_chunk(c0 + PW_KC);
#pragma unroll
        for (int s = 0; s < PW_KC / 4; ++s) {
            const int pair = kh * (PW_KC / 4) + s;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[2 * pair], bcol[2 * pair * 64], acc, 0, 0, 0);
        }
    }
    __syncthreads();
    float* red = bt;  // 2 x 16 x 64 floats
    if (kh == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(pt * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (kh == 1) return;
    const int p = p0 + pt * 32 + l32;
    int ob, oy, ox;
    decode(p, ob, oy, ox);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = n0 + (r >> 2) * 8 + kk * 4 + (r & 3);
        float v = acc[r] + red[(pt * 16 + r) * 64 + lane];
        if (bias) v += bias[to.sub ? (n >> 2) : n];
        if (p < M) {
            float* o = out + pw_addr(to, ob, n, oy, ox);
            o[0] = v;
            if (!to.sub && to.stride == 2) {  // the three pixels of the 2x2 cell that the strided product does not reach
                o[1] = 0.f;
                o[to.W] = 0.f;
                o[to.W + 1] = 0.f;
            }
        }
    }
}

// dW[n * Cv + c] = sum over images and coarse pixels of Y[b][n][p] X[b][c][p].  Workgroup: 64 x 32 values of dW x a
// range of 64-pixel chunks (blockIdx.z: slab z); wave: half of the 64 rows x half of a chunk's pixel pairs.
__global__ __launch_bounds__(256) void k_pw_wgrad_train(const float* __restrict__ Y, const float* __restrict__ X,
                                                        float* __restrict__ dw, const PwTensor ty, const PwTensor tx, int B,
                                                        int Hc, int Wc, int Nv, int Cv, int chunks_per_split) {
    constexpr int LD = 65;
    __shared__ float yt[64 * LD];  // [n][pixel]
    __shared__ float xt[32 * LD];  // [c][pixel]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = wave & 1, kh = wave >> 1;
    const int l32 = lane & 31, kk = lane >> 5;
    const int hw = Hc * Wc, M = B * hw;
    const int c0 = blockIdx.x * 32, n0 = blockIdx.y * 64;
    const int n_chunks = (M + 63) / 64;
    const int q_begin = blockIdx.z * chunks_per_split, q_end = min(n_chunks, q_begin + chunks_per_split);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int px = tid & 63, cq = tid >> 6;
    float yv[16], xv[8];
    auto load_chunk = [&](int q) {  // (a pixel past the end re-reads the last one; zeroed when stored)
        const int pc = min(q * 64 + px, M - 1);
        const int b = pc / hw, r = pc - b * hw, y = r / Wc, x = r - y * Wc;
#pragma unroll
        for (int i = 0; i < 16; ++i) yv[i] = Y[pw_addr(ty, b, n0 + cq + 4 * i, y, x)];
#pragma unroll
        for (int i = 0; i < 8; ++i) xv[i] = X[pw_addr(tx, b, c0 + cq + 4 * i, y, x)];
    };
    if (q_begin < q_end) load_chunk(q_begin);
    for (int q = q_begin; q < q_end; ++q) {
        const bool ok = q * 64 + px < M;
        __syncthreads();  // the previous chunk has been multiplied
#pragma unroll
        for (int i = 0; i < 16; ++i) yt[(cq + 4 * i) * LD + px] = ok ? yv[i] : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) xt[(cq + 4 * i) * LD + px] = ok ? xv[i] : 0.f;
        __syncthreads();
        if (q + 1 < q_end) load_chunk(q + 1);
        const float* arow = yt + (nt * 32 + l32) * LD + kk;
        const float* bcol = xt + l32 * LD + kk;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int pp = 2 * (kh * 16 + s);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[pp], bcol[pp], acc, 0, 0, 0);
        }
    }
    __syncthreads();
    float* red = yt;  // 2 x 16 x 64 floats
    if (kh == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(nt * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (kh == 1) return;
    float* dst = dw + (size_t)blockIdx.z * ((size_t)Nv * Cv);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = n0 + nt * 32 + (r >> 2) * 8 + kk * 4 + (r & 3);
        dst[(size_t)n * Cv + c0 + l32] = acc[r] + red[(nt * 16 + r) * 64 + lane];
    }
}

namespace {

// kind 0: nn.Conv2d(Cin, Cout, 1, stride 2) on x [B][Cin][H][W] (H, W even) -> y [B][Cout][H/2][W/2], w [Cout][Cin]
// kind 1: nn.ConvTranspose2d(Cin, Cout, 2, stride 2) on x [B][Cin][H][W] -> y [B][Cout][2H][2W], w [Cin][Cout][2][2]
struct PwPlan {
    int Hc, Wc;                    // coarse grid
    PwTensor x, y;                 // the layer's input and output tensors as operands
    int xv, yv;                    // their virtual channel counts
};

bool pw_plan(int kind, int B, int Cin, int Cout, int H, int W, PwPlan& p) {
    if (B <= 0 || Cin <= 0 || Cout <= 0 || H <= 0 || W <= 0) return false;
    if (kind == 0) {
        if ((H & 1) || (W & 1) || Cin % 64 || Cout % 64) return false;
        p.Hc = H / 2; p.Wc = W / 2;
        p.x = PwTensor{Cin, H, W, 0, 2};
        p.y = PwTensor{Cout, H / 2, W / 2, 0, 1};
        p.xv = Cin; p.yv = Cout;
    } else if (kind == 1) {
        if (Cin % 64 || Cout % 16) return false;
        p.Hc = H; p.Wc = W;
        p.x = PwTensor{Cin, H, W, 0, 1};
        p.y = PwTensor{Cout, 2 * H, 2 * W, 1, 1};
        p.xv = Cin; p.yv = 4 * Cout;
    } else {
        return false;
    }
    const long long big = (long long)B * (p.xv > p.yv ? p.xv : p.yv) * p.Hc * p.Wc;
    return big < (1LL << 31) && (long long)p.xv * p.yv < (1LL << 31);
}

void pw_wgrad_split(const PwPlan& p, int B, int Nv, int Cv, int& slices, int& cps) {
    const int n_chunks = (B * p.Hc * p.Wc + 63) / 64;
    const long tiles = (long)(Nv / 64) * (Cv / 32);
    long s = tiles >= 128 ? 1 : (256 + tiles - 1) / tiles;
    if (s > n_chunks) s = n_chunks;
    if (s > 128) s = 128;
    cps = (int)((n_chunks + s - 1) / s);
    slices = (n_chunks + cps - 1) / cps;
}

}  // namespace

static void pw_dgrad_split(const PwPlan& p, int kind, int B, int& ks, int& cps);

bool pointwise_train_supported(int kind, int B, int Cin, int Cout, int H, int W) {
    PwPlan p;
    return pw_plan(kind, B, Cin, Cout, H, W, p);
}

size_t pointwise_train_workspace_floats(int kind, int B, int Cin, int Cout, int H, int W) {
    PwPlan p;
    if (!pw_plan(kind, B, Cin, Cout, H, W, p)) return 0;
    // weight gradient: rows = the operand with at least 64 (virtual) channels per tile row (see launch_pointwise_wgrad_train)
    const int Nv = kind == 0 ? p.yv : p.xv, Cv = kind == 0 ? p.xv : p.yv;
    int sl, cps, ks;
    pw_wgrad_split(p, B, Nv, Cv, sl, cps);
    size_t need = sl > 1 ? (size_t)sl * Nv * Cv : 4;
    pw_dgrad_split(p, kind, B, ks, cps);
    if (ks > 1) need = std::max(need, (size_t)ks * B * p.Hc * p.Wc * p.xv);
    return need;
}

hipError_t launch_pointwise_train_fwd(int kind, const float* x, const float* w, const float* bias, float* y, int B, int Cin,
                                      int Cout, int H, int W, hipStream_t stream) {
    PwPlan p;
    if (!pw_plan(kind, B, Cin, Cout, H, W, p)) return hipErrorInvalidValue;
    const int M = B * p.Hc * p.Wc;
    const dim3 grid((M + 63) / 64, p.yv / 32);
    // kind 0: A[n = co][c = ci] = w[co * Cin + ci]; kind 1: A[n = (co, d)][c = ci] = w[ci * 4 Cout + n]
    hipLaunchKernelGGL(k_pw_train, grid, dim3(256), 0, stream, x, w, bias, y, p.x, p.y, B, p.Hc, p.Wc, p.xv, p.yv,
                       kind == 0 ? Cin : 1, kind == 0 ? 1 : 4 * Cout, p.xv / PW_KC);
    return hipGetLastError();
}

// the data gradient of a transposed convolution reduces over 4 Cout virtual channels (up to 2048) into a plain tensor: on the
// small maps that reduction is split over workgroups (slabs in `workspace`, added in slab order)
static void pw_dgrad_split(const PwPlan& p, int kind, int B, int& ks, int& cps) {
    const int nchunks = p.yv / PW_KC;
    const long base = (long)((B * p.Hc * p.Wc + 63) / 64) * (p.xv / 32);
    ks = 1;
    if (kind == 1)
        while (base * ks < 256 && nchunks / (ks * 2) >= 2 && ks < 16) ks *= 2;
    cps = (nchunks + ks - 1) / ks;
    ks = (nchunks + cps - 1) / cps;
}

hipError_t launch_pointwise_train_dgrad(int kind, const float* dy, const float* w, float* dx, int B, int Cin, int Cout, int H,
                                        int W, float* workspace, hipStream_t stream) {
    PwPlan p;
    if (!pw_plan(kind, B, Cin, Cout, H, W, p)) return hipErrorInvalidValue;
    const int M = B * p.Hc * p.Wc;
    // (kind 0: only every second pixel of every second row receives a gradient; the kernel writes the zeros between them)
    int ks, cps;
    pw_dgrad_split(p, kind, B, ks, cps);
    const dim3 grid((M + 63) / 64, p.xv / 32, ks);
    // kind 0: A[n = ci][c = co] = w[co * Cin + ci]; kind 1: A[n = ci][c = (co, d)] = w[ci * 4 Cout + c]
    hipLaunchKernelGGL(k_pw_train, grid, dim3(256), 0, stream, dy, w, (const float*)nullptr, ks > 1 ? workspace : dx, p.y, p.x, B,
                       p.Hc, p.Wc, p.yv, p.xv, kind == 0 ? 1 : 4 * Cout, kind == 0 ? Cin : 1, cps);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || ks == 1) return e;
    return sum_slabs(workspace, dx, (size_t)M * p.xv, ks, stream);
}

hipError_t launch_pointwise_train_wgrad(int kind, const float* x, const float* dy, float* dw, int B, int Cin, int Cout, int H,
                                        int W, float* workspace, hipStream_t stream) {
    PwPlan p;
    if (!pw_plan(kind, B, Cin, Cout, H, W, p)) return hipErrorInvalidValue;
    // kind 0: dw[co][ci]: rows from dy, columns from x; kind 1: dw[ci][(co, d)]: rows from x, columns from dy
    const float* rows = kind == 0 ? dy : x;
    const float* cols = kind == 0 ? x : dy;
    const PwTensor tr = kind == 0 ? p.y : p.x, tc = kind == 0 ? p.x : p.y;
    const int Nv = kind == 0 ? p.yv : p.xv, Cv = kind == 0 ? p.xv : p.yv;
    int sl, cps;
    pw_wgrad_split(p, B, Nv, Cv, sl, cps);
    float* dst = sl > 1 ? workspace : dw;
    hipLaunchKernelGGL(k_pw_wgrad_train, dim3(Cv / 32, Nv / 64, sl), dim3(256), 0, stream, rows, cols, dst, tr, tc, B, p.Hc,
                       p.Wc, Nv, Cv, cps);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || sl == 1) return e;
    return sum_slabs(workspace, dw, (size_t)Nv * Cv, sl, stream);
}

}  // namespace danra
