// Internal launch interface between the host runtime (danra_api.hip) and the gfx950 kernels.
// Every launcher enqueues on `stream` and returns immediately; shapes are validated by the caller.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stddef.h>
#include <stdint.h>

namespace danra {

// Timing hook for the kernel families that can be profiled from the C ABI.  While a source is set (per host thread),
// every launch through DANRA_LAUNCH_KERNEL takes an event pair from it and attaches the pair to the dispatch itself
// (hipExtLaunchKernelGGL): the events then carry the kernel's own begin and end times, what a kernel trace reports
// as its duration.  Events recorded around a launch would add the two marker packets and the launch latency, 5-7 us
// on kernels that run for 10.
struct LaunchEvents {
    void* ctx;
    int (*take)(void* ctx, hipEvent_t* start, hipEvent_t* stop);  // 0 = a pair was handed out
};
extern thread_local const LaunchEvents* t_launch_events;  // (kernels_misc.hip)

#define DANRA_LAUNCH_KERNEL(kernel, grid, block, lds, stream, ...)                                                    \
    do {                                                                                                              \
        hipEvent_t ev0_ = nullptr, ev1_ = nullptr;                                                                    \
        const ::danra::LaunchEvents* src_ = ::danra::t_launch_events;                                                 \
        if (src_ && src_->take(src_->ctx, &ev0_, &ev1_) == 0)                                                         \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, ev0_, ev1_, 0, __VA_ARGS__);                      \
        else                                                                                                          \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                        \
    } while (0)

// ---------------------------------------------------------------- implicit GEMM (convs, linears)
enum AMode { A_GEMM = 0, A_CONV = 1, A_CONV1 = 2 };
enum StoreMode { ST_ROW = 0, ST_QKV = 1, ST_CONVT = 2 };
enum Activation { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2 };

// out[m][n] = epilogue( sum_k A(m,k) * W[n][k] ).
//   A_GEMM : A(m,k) = A[m*lda + k], optionally LayerNorm-ed per row over the whole K (ln=1).
//   A_CONV : NHWC input, k = (kh, kw, c), zero padding; optional per-(image,channel) normalisation
//            (x - nmean) * nrstd of the in-bounds taps (InstanceNorm applied on load).
//   A_CONV1: up to four [B][Hi][Wi] planes, k = (plane, kh, kw)  (the network's 8x8 stride-2 stem).
// epilogue: v += bias[n]; v += res[m*ldr+n]; act (none / ReLU / exact GELU); v += tbias[(m / rows_per_img)*tb_stride + n]; store.
struct IGemmArgs {
    const float* A;
    const float* planes[4];
    const float* W;
    const float* Wh;      // the same matrix split in fp16 halves (launch_split_weights), or null; used when `half` is set
    float wscale;         // 2^-s where Wh holds 2^s * W
    const float* bias;
    const float* res;
    const float* tbias;
    const float* nmean;
    const float* nrstd;
    const float* zeros;  // >= 16 bytes of zeros: what a convolution's padding taps read
    float* out;
    float* out2;     // optional second output (ST_ROW only): columns n >= n_split go to out2[m*ldc + n - n_split], bias only
    float* partial;  // split-K slabs [ksplit][M][N]
    int M, N, K;
    int n_split;     // with out2: first column of the second output (a multiple of 64)
    int lda;
    int ldw;  // row stride of W in floats; 0 = K
    int a_mode;
    int ln;
    int Hi, Wi, Cin, Ho, Wo, KH, KW, stride, pad;
    int ldr, act, tb_stride, rows_per_img;
    int st_mode, ldc;
    int qkv_C, qkv_hd, qkv_L, qkv_heads;
    long long qkv_plane;
    int ct_C, ct_H, ct_W;
    int pix_major, batch;          // convolutions: rows ordered pixel-major (m = pixel*batch + image) when pix_major != 0
    int ksplit, chunks_per_split;  // chunks are counted 32-deep
    int force_bm, force_kw;        // 0 = let the launcher choose the tile (tuning / tests)
    int keep_slabs;                // split-K only: leave the slabs in `partial` (no reduction kernel); the consumer sums them
    int half;                      // != 0: multiply on the fp16 matrix cores with operands split in two fp16 halves (k_igemm HALF)
    // ST_QKV with head_dim 16 through the K = 64 kernel (gemm_k64_presplits): K and V leave the kernel in the pre-split form
    // of the attention core (launch_presplit_kv describes it), with their range flags; the fp32 K / V planes of `out` are
    // then written only for the groups of keys whose flag is set.  All three or none.
    float* pre_ks;
    float* pre_vs;
    int* pre_flags;
    // filled in by launch_igemm: log2 of a divisor of the row decomposition when it is a power of two, else -1
    // (rows_per_img, batch, Ho*Wo, Wo, qkv_L, ct_H*ct_W, ct_W): the kernels then shift instead of dividing
    int sh_rpi, sh_batch, sh_howo, sh_wo, sh_qkv_l, sh_ct_hw, sh_ct_w;
    int debug;                     // timing ablations of tune_igemm (results are wrong): 1 = always stage chunk 0, 2 = no MFMA, 4 = no LDS store/barrier
};
hipError_t launch_igemm(const IGemmArgs& a, hipStream_t stream);
// whether launch_igemm would run `a` (a QKV projection) through the kernel that can leave K and V pre-split
bool gemm_k64_presplits(const IGemmArgs& a);
// on != 0: every later launch_igemm multiplies with fp32 matrix instructions whatever `half` says.  Process-wide; for
// numerical cross-checks.
void set_igemm_fp32_products(int on);
bool igemm_uses_split(const IGemmArgs& a);

// Tail of a 64-channel attention block as one kernel (M token rows of 64 floats, every step local to a row):
//   a = o Wo^T + bo + x;  with W1: a += GELU(LayerNorm(a) W1^T + b1) W2^T + b2 (LayerNorm affine folded into W1/b1);
//   v = act(a);  with Wt ([256][64], rows (dy, dx, co)): the kernel-2 stride-2 transposed convolution that consumes v is
//   applied as well and `out` is its [B][2*ct_H][2*ct_W][64] result (v itself is not stored); else out = v.
// With Wt and `part` (M and ct_H*ct_W multiples of 32) the kernel also leaves the InstanceNorm statistics of its output in
// slices: part[image][S = ct_H*ct_W/8][{mean, centred sum of squares}][64], 32 samples per slice, for
// launch_in_stats_from_slices.  At least one of W1 and Wt must be given.
// ct_stats_only (with Wt and `part`): the transposed convolution is formed for its statistics only and `out` receives
// the block's own output rows [M][64] - for a consumer that rebuilds the transposed convolution (launch_final_up_conv).
// `split` (optional): the matrices split in fp16 halves in fragment order (launch_split_weights with frag64) and the
// inverse of each one's power-of-two scale; the chain then multiplies on the fp16 matrix cores (fp32 results, fp32
// instructions for rows out of the split's range).
struct Tail64Split {
    const float *wo = nullptr, *w1 = nullptr, *w2 = nullptr, *wt = nullptr;
    float so = 1.f, s1 = 1.f, s2 = 1.f, st = 1.f;
};
hipError_t launch_attn_tail64(const float* o, const float* x, const float* Wo, const float* bo, const float* W1,
                              const float* b1, const float* W2, const float* b2, const float* Wt, const float* bt,
                              float* out, int M, int act, int ct_H, int ct_W, float* part, hipStream_t stream,
                              const Tail64Split* split = nullptr, int ct_stats_only = 0);

// One input plane's share of the 8x8 stride-2 pad-3 stem: out = conv(x; Wt[64][8x8], row stride ldw) [+ res] [+ tbias[b]],
// x: [B][H][W], out/res: [B][H/2][W/2][64]; H and W multiples of 16.
hipError_t launch_stem_x(const float* x, const float* Wt, int ldw, const float* res, const float* tbias, int tb_stride,
                         float* out, int B, int H, int W, hipStream_t stream);

// ---------------------------------------------------------------- attention
// q,k,v: [B*heads][L][hd]; out tokens [B][L][heads*hd] at column head*hd.
// The 3x3 stride-1 padding-1 convolution to ONE output channel in a training step, NCHW fp32 (kernels_conv_c1_train.hip):
// x, dx [B][Cin][H][W]; w, dw [1][Cin][3][3]; y, dy [B][1][H][W]; workspace: conv_c1_train_workspace_floats floats.
bool conv_c1_train_supported(int B, int Cin, int H, int W);
size_t conv_c1_train_workspace_floats(int B, int Cin, int H, int W);
hipError_t launch_conv_c1_train_fwd(const float* x, const float* w, float* y, int B, int Cin, int H, int W, hipStream_t stream);
hipError_t launch_conv_c1_train_dgrad(const float* dy, const float* w, float* dx, int B, int Cin, int H, int W, hipStream_t stream);
hipError_t launch_conv_c1_train_wgrad(const float* x, const float* dy, float* dw, int B, int Cin, int H, int W, float* workspace,
                                      hipStream_t stream);
// Where q, k, v of one (image, head) lie: rows `rs` floats apart, the pair (b, head) at b * img + head * hs floats.
// [B*heads][L][16]: {16, heads * L * 16, L * 16}; the output of a fused QKV projection [B][L][3][heads * 16], read in
// place: {3 * heads * 16, L * 3 * heads * 16, 16} with k = q + heads * 16, v = q + 2 * heads * 16.
struct QkvLayout {
    int rs;
    long long img;
    int hs;
};
// The 3x3 stride-2 padding-1 convolutions in a training step, NCHW fp32 (kernels_conv3s2_train.hip): x, dx [B][Cin][H][W];
// w, dw [Cout][Cin][3][3]; y, dy [B][Cout][H/2][W/2]; workspace: conv3x3s2_train_workspace_floats floats.
bool conv3x3s2_train_supported(int B, int Cin, int Cout, int H, int W);
size_t conv3x3s2_train_workspace_floats(int B, int Cin, int Cout, int H, int W);
hipError_t launch_conv3x3s2_train_fwd(const float* x, const float* w, float* y, int B, int Cin, int Cout, int H, int W,
                                      float* workspace, hipStream_t stream);
hipError_t launch_conv3x3s2_train_dgrad(const float* dy, const float* w, float* dx, int B, int Cin, int Cout, int H, int W,
                                        float* workspace, hipStream_t stream);
hipError_t launch_conv3x3s2_train_wgrad(const float* x, const float* dy, float* dw, int B, int Cin, int Cout, int H, int W,
                                        float* workspace, hipStream_t stream);
// The 8x8 stride-2 padding-3 convolutions of the stem in a training step, NCHW fp32 (kernels_conv8_train.hip): x, dx
// [B][Cin][H][W]; w, dw [Cout][Cin][8][8]; y, dy [B][Cout][H/2][W/2]; workspace: conv8s2_train_workspace_floats floats.
bool conv8s2_train_supported(int B, int Cin, int Cout, int H, int W);
bool conv8s2_train_dgrad_supported(int B, int Cin, int Cout, int H, int W);
size_t conv8s2_train_workspace_floats(int B, int Cin, int Cout, int H, int W);
hipError_t launch_conv8s2_train_fwd(const float* x, const float* w, float* y, int B, int Cin, int Cout, int H, int W,
                                    hipStream_t stream);
hipError_t launch_conv8s2_train_dgrad(const float* dy, const float* w, float* dx, int B, int Cin, int Cout, int H, int W,
                                      hipStream_t stream);
hipError_t launch_conv8s2_train_wgrad(const float* x, const float* dy, float* dw, int B, int Cin, int Cout, int H, int W,
                                      float* workspace, hipStream_t stream);
// softmax(q k^T / sqrt(hd)) v of the short-sequence attention blocks for training (kernels_attn_small_train.hip): q, k, v,
// dq, dk, dv [B*heads][L][hd]; out, dout [B][L][heads*hd].
bool attention_small_train_supported(int B, int heads, int L, int hd);
hipError_t launch_attention_small_fwd(const float* q, const float* k, const float* v, float* out, int B, int heads, int L, int hd,
                                      hipStream_t stream);
hipError_t launch_attention_small_bwd(const float* q, const float* k, const float* v, const float* dout, float* dq, float* dk,
                                      float* dv, int B, int heads, int L, int hd, hipStream_t stream);
// both on the output of a fused QKV projection used in place: qkv, dqkv [B][L][3][heads*hd] (see QkvLayout)
hipError_t launch_attention_small_qkv_fwd(const float* qkv, float* out, int B, int heads, int L, int hd, hipStream_t stream);
hipError_t launch_attention_small_qkv_bwd(const float* qkv, const float* dout, float* dqkv, int B, int heads, int L, int hd,
                                          hipStream_t stream);
// K, V [n_bh][L][16] -> the pre-split form of the head_dim-16 core (ks, vs: as many bytes as k, v; flags: 2 * n_bh * L / 16
// ints), and that core on it; k32 / v32 are read only where a flag is set (kernels_attn.hip, k_flash_hd16 PRE).
hipError_t launch_presplit_kv(const float* k, const float* v, float* ks, float* vs, int* flags, int n_bh, int L,
                              hipStream_t stream);
hipError_t launch_attention_hd16_pre(const float* q, const float* ks, const float* vs, const float* k32, const float* v32,
                                     const int* flags, float* out, int B, int heads, int L, hipStream_t stream);
hipError_t launch_attention(const float* q, const float* k, const float* v, float* out, int B, int heads,
                            int L, int hd, hipStream_t stream);
// LayerNorm + QKV projection of one head + that head's attention as one kernel per (image, head), for the short
// sequences of the deep levels (see attn_block_small_supported): x [B][L][C] tokens, w_in_frag = in_proj [3C][C] with
// the LayerNorm affine folded in, split in fp16 halves after a scale of 1 / wscale (launch_split_weights) and brought
// into fragment order (launch_split_to_frag16), b_in [3C], o [B][L][C] head outputs (the operand of the output
// projection).
bool attn_block_small_supported(int B, int L, int C, int heads);
hipError_t launch_attn_block_small(const float* x, const float* w_in_frag, float wscale, const float* b_in, float* o,
                                   int B, int L, int C, int heads, hipStream_t stream);
// split [rows][K] (launch_split_weights without frag64; rows % 16 == 0, K % 64 == 0) -> out, same size: for 16-row tile T
// and 32-deep step s the 64 x 16 bytes of lanes (row 16 T + (lane & 15), k = 32 s + 8 (lane >> 4) ..) hi, then lo.
hipError_t launch_split_to_frag16(const float* split, float* out, int rows, int K, hipStream_t stream);
bool igemm_fp32_products_forced();
// out = act(A W^T + bias + res) for products with few rows (K = 128 / 256 / 512, M <= 8192; see proj_small_supported):
// the output projections of the short-sequence attention blocks.  w_frag = W split in fp16 halves in fragment order
// (scale 1 / wscale), w = the fp32 matrix (for rows outside the split's range).
// (K = 64 as well, up to 65536 rows: the output projections of the 64-channel blocks)
bool proj_small_supported(int M, int N, int K);
// The LayerNorm-fused QKV projection of an attention block on few tokens (C = 64: <= 16384, C = 128 / 256 / 512: <= 8192)
// through the same kernel: x [M][C], w_frag / w / bias = in_proj [3C][C] with the LayerNorm affine folded in,
// qkv [3][M / L * heads][L][hd].
bool qkv_small_supported(int M, int C, int L, int hd);
// pre_ks / pre_vs / pre_flags (all or none; qkv_small_presplits): K and V leave the kernel in the pre-split form of the
// head_dim-16 attention core (launch_presplit_kv), the fp32 K / V planes of `qkv` are written only where a flag is set.
bool qkv_small_presplits(int M, int C, int L, int hd);
hipError_t launch_qkv_small(const float* x, const float* w_frag, const float* w, float wscale, const float* bias, float* qkv,
                            int M, int C, int L, int hd, hipStream_t stream, float* pre_ks = nullptr, float* pre_vs = nullptr,
                            int* pre_flags = nullptr);
// w [rows][K] fp32 -> scale * w split in fp16 halves in the fragment order of launch_split_to_frag16 (rows % 16 == 0, K % 32 == 0)
hipError_t launch_frag16_from_f32(const float* w, float* out, int rows, int K, float scale, hipStream_t stream);
// With ct_W != 0: a kernel-2 stride-2 transposed convolution on a ct_H x ct_W map (N = 4 x channels), rows stored
// pixel-shuffled (see k_proj_small).
// act: an Activation; layernorm: the rows are normalised over their K channels first (K >= 128; the affine part is
// folded into w and bias), as the feed-forward part of an attention block needs it.
hipError_t launch_proj_small(const float* a, const float* w_frag, const float* w, float wscale, const float* bias,
                             const float* res, float* out, int M, int N, int K, int act, hipStream_t stream, int ct_H = 0,
                             int ct_W = 0, bool layernorm = false);
// head_dim 16, L a multiple of 128.  Forward as above; lse2 (may be null): [B*heads][L], the log-sum-exp of every
// query's scaled scores in the exp2 domain, which the backward pass needs.
hipError_t launch_attention_hd16(const float* q, const float* k, const float* v, float* out, float* lse2, int B,
                                 int heads, int L, hipStream_t stream);
// on != 0: every later launch_attention_hd16 multiplies with fp32 matrix instructions only (v_mfma_f32_16x16x4_f32)
// instead of the split-operand fp16 products.  Process-wide; for numerical cross-checks.
void set_attention_hd16_fp32_products(int on);
int attention_hd16_fp32_products();
// Gradients of that forward: dout tokens [B][L][heads*16] -> dq, dk, dv [B*heads][L][16]; delta: [B*heads][L] scratch
// (receives sum_d dout*out per query).  Two kernels, no atomics.
hipError_t launch_attention_hd16_bwd(const float* q, const float* k, const float* v, const float* out, const float* dout,
                                     const float* lse2, float* dq, float* dk, float* dv, float* delta, int B, int heads,
                                     int L, hipStream_t stream);

// Both on the output of a fused QKV projection used in place (the training step's in_proj): qkv, dqkv [B][L][3][heads*16].
hipError_t launch_attention_hd16_tokens(const float* qkv, float* out, float* lse2, int B, int heads, int L, hipStream_t stream);
hipError_t launch_attention_hd16_bwd_tokens(const float* qkv, const float* out, const float* dout, const float* lse2,
                                            float* dqkv, float* delta, int B, int heads, int L, hipStream_t stream);

// ---------------------------------------------------------------- small kernels
// Device-resident state of the sampling loop: one denoise step can be enqueued (or replayed from a
// hipGraph) without any per-step host argument.  k_time_mlp reads t_fwd and publishes it as t_upd; the
// update kernel reads t_upd and publishes t_upd - 1 as t_fwd, so no kernel reads a field it writes (the update
// kernel that also gathers the next step's time projections is the exception, see NextGather).
struct SampleState {
    int t_fwd;                 // timestep of the next UNet evaluation
    int t_upd;                 // timestep of the next update
    int t_max;                 // T - 1: the timestep of the first draw
    int philox;                // 1: z drawn in-kernel (Philox), 0: z read from `noise`
    unsigned long long seed;
    const float* noise;        // host-noise mode: [T-2][n] draws in draw order (timestep T-1 first), or null
    float cfg_scale;           // classifier-free guidance weight (used when the update is given two predictions)
    float noise_scale;         // every draw is multiplied by this (1, or 0.005 for the "scaled data" mode)
    unsigned long long quad_offset;  // Philox counter of this call's first element quad (a batch slice of a larger batch)
    unsigned int ticket;       // blocks of the update kernel that are done (the last one hands the timestep over)
};
// What launch_tbias_gather does for the next evaluation, as a tail of the update kernel.
struct NextGather {
    const float* table;        // [T][classes][width] time projections
    const int64_t* y;          // [B] labels or null
    float* tbias;              // [B][width]
    int classes, width, B;
};
hipError_t launch_init_state(SampleState* s, int t_start, int philox, unsigned long long seed, const float* noise,
                             float cfg_scale, float noise_scale, unsigned long long quad_offset, hipStream_t stream);
// x <- c1*(x - c2*eps) + c3*(noise_scale*z) with (c1, c2, c3) = coef[state->t_upd] and z = 0 at timestep 1.
// With eps_uncond != null, eps is first replaced by lerp(eps_uncond, eps, cfg_scale).
// With `next` the kernel also gathers tbias for the evaluation at timestep t_upd - 1 and publishes that timestep as
// t_upd as well as t_fwd (no gather kernel runs at the head of the next step).
hipError_t launch_ddpm_update_state(float* x, const float* eps, const float* eps_uncond, const float4* coef,
                                    SampleState* state, size_t n, hipStream_t stream, const NextGather* next = nullptr);

struct TimeMlpArgs {
    const int64_t* t;          // [B], or null when `state` supplies one timestep for every row
    SampleState* state;        // may be null
    const int64_t* y;          // may be null
    const float* label_emb;    // [num_classes][dim] or null
    const float* enc_inv_freq; // [dim/2]
    const float* dec_div;      // [dim/2]
    const float* W;            // [width][dim]
    const float* b;            // [width]
    float* enc_emb_out;        // [B][dim] or null
    float* dec_emb_out;        // [B][dim] or null
    float* tbias;              // [B][width] or null
    int B, dim, width, enc_width, num_classes;
    int group;                 // set by the launcher
};
hipError_t launch_time_mlp(const TimeMlpArgs& a, hipStream_t stream);
// tbias[b] = table[state->t_fwd][y_b] (table: [T][classes][width]); also publishes state->t_upd = state->t_fwd
hipError_t launch_tbias_gather(SampleState* state, const float* table, const int64_t* y, float* tbias, int classes,
                               int width, int B, hipStream_t stream);

// `part`: scratch of part_floats floats for sliced statistics (may be null: one block per image and 32 channels)
// With `out` != null the normalised tensor (+ skip, + tbias[b][c]) is written as well (one fused pass when an
// image's pixels fit a block's registers, statistics + k_in_apply otherwise).
// `slabs` (unsliced launches only, i.e. HW <= 1024): x is not one tensor but the split-K slabs of the GEMM that produces
// it - value = slab[0] + slab[1] + ... (in that order) + bias[c], each slab `stride` floats after the previous one,
// rows in the GEMM's order (pixel-major: row = pixel * B + image) - so the reduction kernel of that GEMM is not needed.
struct InSlabs {
    int count;          // number of slabs (>= 2)
    size_t stride;      // floats between slabs (M * N of the GEMM)
    const float* bias;  // [C] or null
    int pix_major;      // row order of the GEMM
    int ct_W;           // != 0: the GEMM is a kernel-2 stride-2 transposed convolution on a map ct_W pixels wide (rows = coarse
                        // pixels, columns = (dy, dx, channel)); needs `out`, which then receives the normalised tensor
};
hipError_t launch_in_stats(const float* x, float* mean, float* rstd, float* part, size_t part_floats, int B, int HW,
                           int C, hipStream_t stream, float* out = nullptr, const float* skip = nullptr,
                           const float* tbias = nullptr, int tb_stride = 0, const InSlabs* slabs = nullptr);
// mean / rstd of B images x 64 channels from S equal slices of `count` samples each (part[b][s][{mean, m2}][64])
hipError_t launch_in_stats_from_slices(const float* part, float* mean, float* rstd, int B, int S, int count,
                                       hipStream_t stream);
// out = (x - mean)*rstd [+ skip] [+ tbias[b][c]]
hipError_t launch_in_apply(const float* x, const float* mean, const float* rstd, const float* skip,
                           const float* tbias, int tb_stride, float* out, int B, int HW, int C, hipStream_t stream);
// 3x3 pad-1 conv Cin -> Cout (Cout <= 4) with InstanceNorm on load, NHWC in, NCHW out.
// With `part` (only where final_conv_merges_slices says so) the statistics are not read from mean / rstd but merged in
// the kernel from S slices of `count` samples each, part[image][S][{mean, m2}][64] (see launch_in_stats_from_slices).
bool final_conv_merges_slices(int H, int W, int Cin);
hipError_t launch_final_conv(const float* x, const float* mean, const float* rstd, const float* w, const float* bias,
                             float* out, int B, int H, int W, int Cin, int Cout, hipStream_t stream,
                             const float* part = nullptr, int S = 0, int count = 0);
// The final block from its input hin [B][Hh][Ww][64] without the upsampled tensor: transposed convolution (Wt [256][64],
// bt [256] as launch_pack_convT lays them out) + InstanceNorm (statistics of the transposed convolution's output in
// slices, part[image][S][{mean, m2}][64], `count` samples each) + 3x3 convolution w [Cout][9][64] -> out [B][Cout][2Hh][2Ww].
bool final_up_conv_supported(int Hh, int Ww);
hipError_t launch_final_up_conv(const float* hin, const float* Wt, const float* bt, const float* w, const float* bias,
                                float* out, int B, int Hh, int Ww, int Cout, const float* part, int S, int count,
                                hipStream_t stream);
hipError_t launch_ddpm_update(float* x, const float* eps, const float* noise, float c1, float c2, float c3,
                              size_t n, int philox, uint64_t seed, uint64_t offset, hipStream_t stream);
// noisy = sqrt(alpha_hat[t_b]) * x + sqrt(1 - alpha_hat[t_b]) * z; z read from `noise` or drawn (and written there)
hipError_t launch_noise_image(const float* x, const int64_t* t, const float* alpha_hat, int T, float* noisy, float* noise,
                              int B, size_t per_image, int philox, uint64_t seed, uint64_t offset, hipStream_t stream);
hipError_t launch_nchw_to_nhwc(const float* src, float* dst, int B, int C, int HW, hipStream_t stream);
hipError_t launch_nhwc_to_nchw(const float* src, float* dst, int B, int C, int HW, hipStream_t stream);

// ---------------------------------------------------------------- elucidated (EDM) sampler steps, kernels_edm.hip
// (NCHW fp32 images; `mask` is one byte per pixel and image, shared by the channels; `thr` one float per image)
hipError_t launch_edm_perturb(const float* images, const float* noise, const float* inpaint, const unsigned char* mask,
                              float* images_hat, float* net_in, int B, int C, int HW, float s_noise, float k, float c_in,
                              int philox, uint64_t seed, uint64_t offset, hipStream_t stream);
hipError_t launch_edm_precondition(const float* x, const float* net, float* out, size_t n, float c_skip, float c_out,
                                   int clamp, hipStream_t stream);
// workspace: edm_quantile_workspace_bytes(B, per_image) bytes (0 for small images: then it may be null)
size_t edm_quantile_workspace_bytes(int B, size_t per_image);
hipError_t launch_edm_quantile_abs(const float* x, float* out, int B, size_t per_image, float q, float floor_value,
                                   void* workspace, hipStream_t stream);
hipError_t launch_edm_euler(const float* images_hat, const float* net, float* model_out, const float* thr, float* d_out,
                            float* images_next, float* net_in_next, int B, size_t per_image, float c_skip, float c_out,
                            int clamp, float sigma_hat, float dt, float c_in_next, hipStream_t stream);
hipError_t launch_edm_heun(const float* images_hat, const float* images_next, const float* net, float* model_out,
                           const float* thr, const float* d_in, float* images_out, int B, size_t per_image, float c_skip,
                           float c_out, int clamp, float sigma_next, float half_dt, hipStream_t stream);
hipError_t launch_edm_renoise(float* images, const float* noise, size_t n, float scale, int philox, uint64_t seed,
                              uint64_t offset, hipStream_t stream);
hipError_t launch_edm_finish(float* images, const float* inpaint, const unsigned char* mask, int B, int C, int HW,
                             int unnormalize, hipStream_t stream);

// ---------------------------------------------------------------- training step
// BatchNorm2d in train mode on NCHW fp32 (HW a multiple of 4): batch statistics, optional fused ReLU; save_mean / save_rstd
// [C] feed the backward pass; running_mean / running_var (may be null) are updated with `momentum` (unbiased variance);
// *batches (may be null; the module's int64 num_batches_tracked) is incremented by one.
hipError_t launch_bn_train_fwd(const float* x, const float* gamma, const float* beta, float* y, float* save_mean,
                               float* save_rstd, float* running_mean, float* running_var, int B, int C, int HW, float eps,
                               float momentum, int relu, hipStream_t stream, const float* res = nullptr,
                               long long* batches = nullptr);
// out[c] = sum_{b, hw} x[b][c][hw] on NCHW fp32 (HW a multiple of 4): a convolution's bias gradient; fixed summation order.
// workspace (may be null: one workgroup per channel): channel_sum_slices(B, C, HW) * C floats, large tensors are then split
// over that many workgroups per channel and merged in order by a second kernel.
int channel_sum_slices(int B, int C, int HW);
hipError_t launch_channel_sum(const float* x, float* out, int B, int C, int HW, float* workspace, hipStream_t stream);
// y is the forward's output (read only with relu, for the mask)
hipError_t launch_bn_train_bwd(const float* x, const float* dy, const float* y, const float* gamma, const float* save_mean,
                               const float* save_rstd, float* dx, float* dgamma, float* dbeta, int B, int C, int HW, int relu,
                               hipStream_t stream, float* dres = nullptr);

// InstanceNorm2d(affine=False) on NCHW fp32, `planes` = B*C planes of HW values (HW a multiple of 4), with its backward
hipError_t launch_in_train_fwd(const float* x, float* y, float* save_mean, float* save_rstd, int planes, int HW, float eps,
                               hipStream_t stream);
hipError_t launch_in_train_bwd(const float* x, const float* dy, const float* save_mean, const float* save_rstd, float* dx,
                               int planes, int HW, hipStream_t stream);
// dW = dY^T X, db = column sums of dY (linear layer over M rows; K = 64, N = 64 / 128 / 192).  `workspace`:
// linear_wgrad_workspace_floats(M, N, K) floats.
size_t linear_wgrad_workspace_floats(int M, int N, int K);
bool linear_wgrad_supported(int M, int N, int K);
hipError_t launch_linear_wgrad(const float* dy, const float* x, float* dw, float* db, int M, int N, int K, float* workspace,
                               hipStream_t stream);

// 3x3 stride-1 padding-1 convolution of the training step on NCHW fp32 tensors with [Cout][Cin][3][3] weights
// (kernels_conv_train.hip): forward, data gradient (dgrad = true: in = dy, out = dx) and weight gradient.  workspace:
// conv3x3_train_workspace_floats(...) floats (slabs of the sums that run over workgroups).
bool conv3x3_train_supported(int B, int Cin, int Cout, int H, int W);
size_t conv3x3_train_workspace_floats(int B, int Cin, int Cout, int H, int W);
hipError_t launch_conv3x3_train(bool dgrad, const float* in, const float* w, float* out, int B, int Cin, int Cout, int H, int W,
                                float* workspace, hipStream_t stream);
hipError_t launch_conv3x3_wgrad_train(const float* x, const float* dy, float* dw, int B, int Cin, int Cout, int H, int W,
                                      float* workspace, hipStream_t stream);
// Entry and exit of an attention block in the training step (kernels_train.hip): NCHW map -> LayerNorm'ed tokens and its
// backward pass (dgamma_dbeta: [2][C]; workspace: ln_tokens_workspace_floats floats), tokens (+ NCHW residual) -> NCHW map.
bool ln_tokens_supported(int B, int C, int L);
size_t ln_tokens_workspace_floats(int B, int C, int L);
hipError_t launch_ln_tokens_fwd(const float* x, const float* gamma, const float* beta, float* normed, float* mean, float* rstd,
                                int B, int C, int L, float eps, hipStream_t stream);
hipError_t launch_ln_tokens_bwd(const float* dn, const float* x, const float* mean, const float* rstd, const float* gamma,
                                float* dx, float* dgamma_dbeta, int B, int C, int L, float* workspace, hipStream_t stream);
hipError_t launch_tokens_to_nchw_add(const float* tokens, const float* add, float* out, int B, int C, int L, hipStream_t stream);
// The 1x1 stride-2 convolutions (kind 0) and the kernel-2 stride-2 transposed convolutions (kind 1) of the training step
// on NCHW fp32 tensors with torch's weight layouts ([Cout][Cin][1][1] / [Cin][Cout][2][2]); H, W: the layer's INPUT map.
bool pointwise_train_supported(int kind, int B, int Cin, int Cout, int H, int W);
size_t pointwise_train_workspace_floats(int kind, int B, int Cin, int Cout, int H, int W);
hipError_t launch_pointwise_train_fwd(int kind, const float* x, const float* w, const float* bias, float* y, int B, int Cin,
                                      int Cout, int H, int W, hipStream_t stream);
hipError_t launch_pointwise_train_dgrad(int kind, const float* dy, const float* w, float* dx, int B, int Cin, int Cout, int H,
                                        int W, float* workspace, hipStream_t stream);
hipError_t launch_pointwise_train_wgrad(int kind, const float* x, const float* dy, float* dw, int B, int Cin, int Cout, int H,
                                        int W, float* workspace, hipStream_t stream);
__global__ void k_sum_slices(const float* __restrict__ part, float* __restrict__ out, int count4, int slices);

// ---------------------------------------------------------------- weight packing
// max |w| over n floats -> *out (one float; not atomic across launches: one launch per destination)
hipError_t launch_maxabs(const float* w, size_t n, float* out, hipStream_t stream);
// Split-operand form of a packed weight matrix for k_igemm's HALF variant: n floats (rows of K, K a multiple of 64) ->
// n float-sized slots; every 64 consecutive weights w become [64 halves hi | 64 halves lo] of scale * w with
// hi = fp16(scale*w) (rounded towards zero) and lo = fp16(scale*w - hi).  `scale` is a power of two.
hipError_t launch_split_weights(const float* w, float* out, size_t n, float scale, hipStream_t stream, bool frag64 = false);

// ---------------------------------------------------------------- 3x3 stride-1 convolutions with the input window in LDS
// out[b][y][x][n] = epilogue( sum_{kh, kw, c} x[b][y + kh - 1][x + kw - 1][c] * w[n][kh][kw][c] ), NHWC fp32, zero padding,
// as split-operand fp16 products (kernels_conv_win.hip).  epilogue as k_igemm's: + bias[n]; + res[row][n]; activation;
// + tbias[b * tb_stride + n].  nmean / nrstd ([B][Cin], both or neither): InstanceNorm of the input applied on load.
struct ConvWinArgs {
    const float* x;
    const void* wf;      // the weights in fragment order (launch_pack_conv_frag), scaled by 1 / wscale
    const float* w32;    // the same weights as fp32 [Cout][3][3][Cin]: what a tile out of the split's range multiplies with
    float wscale;
    const float* bias;
    const float* res;
    const float* tbias;
    const float* nmean;
    const float* nrstd;
    float* out;
    int B, H, W, Cin, Cout;   // H, W: the input map; Cout: every output column (both outputs with out2)
    int act, tb_stride;
    int stride;          // 1, or 2 (k_conv_win only: padding 1, output H/2 x W/2; no norm on load)
    float* out2;         // stride 2 only: columns n_split .. Cout - 1 go here ([rows][Cout - n_split]) with their bias only,
    int n_split;         // columns 0 .. n_split - 1 to `out` ([rows][n_split]); n_split a multiple of 32; no res / tbias then
    // filled in by launch_conv_win
    int sh_w, sh_hw, lds_img, lds_window, lds_bufs, swap_grid;
};
// H, W powers of two (W <= 64), Cin % 64 == 0, Cout % 32 == 0 and a tile shape that fits: otherwise the caller uses launch_igemm
bool conv_win_supported(int B, int H, int W, int Cin, int Cout, bool norm, int stride = 1);
hipError_t launch_conv_win(const ConvWinArgs& a, hipStream_t stream);
// w [Cout][taps][Cin] fp32 -> out (same number of bytes): scale * w split in fp16 halves, in the fragment order of the
// window kernels (k_conv_win: 9 taps; k_conv8s2: 64)
hipError_t launch_pack_conv_frag(const float* w, float* out, int Cout, int Cin, float scale, hipStream_t stream, int taps = 9);
// out[b][oy][ox][n] = epilogue( sum_{kh, kw, c} x[b][2 oy + kh - 3][2 ox + kw - 3][c] * w[n][kh][kw][c] ): the 8x8 stride-2
// padding-3 convolution (kernels_conv_s2.hip), arguments as launch_conv_win's (H, W: the INPUT map; no norm on load;
// w32 is [Cout][8][8][Cin]).  H, W powers of two >= 16, Cin % 64 == 0, Cout % 64 == 0.
bool conv8s2_supported(int B, int H, int W, int Cin, int Cout);
hipError_t launch_conv8s2(const ConvWinArgs& a, hipStream_t stream);
hipError_t launch_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var,
                          float* scale, float* shift, int C, hipStream_t stream);
hipError_t launch_pack_conv(const float* w_oihw, const float* scale, float* out_ohwi, int O, int I, int KH, int KW,
                            hipStream_t stream);
hipError_t launch_pack_convT(const float* w_iokk, const float* bias, float* w_out, float* bias_out, int Ci, int Co,
                             hipStream_t stream);
// LayerNorm(C) affine folded into the Linear(C -> rows) behind it
hipError_t launch_pack_inproj(const float* w_in, const float* b_in, const float* ln_g, const float* ln_b,
                              float* w_out, float* b_out, int rows, int C, hipStream_t stream);

}  // namespace danra
