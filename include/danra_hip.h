/*
 * danra_hip.h - C ABI of libdanra_hip.so: the MI355X (gfx950) implementation of the DANRA
 * DDPM sampling loop and attention-UNet forward pass.
 *
 * The reference (TheaQG/DiffusionModelsCustom) is pure Python: it has no FFI of its own, so each
 * entry point below names the reference Python interface it replaces (file:line relative to the
 * reference tree, DDPM_DANRA_conditional/).  All pointers named *_dev are device pointers owned
 * by the caller (PyTorch tensors in the shipped binding); the library borrows them for the call.
 * Everything is fp32 except the integer timestep / class tensors (int64, as torch.long).  Storage and accumulation
 * are fp32 throughout.  The large matrix products - every convolution but the stem, every projection and transposed
 * convolution with 64 or more input channels, and both products of the head_dim-16 attention core - are by default
 * evaluated on the fp16 matrix cores as three fp16 products of operands split in two fp16 halves (22 significant
 * bits per operand, the low x low term dropped, fp32 accumulation), with a per-tile fallback to fp32 matrix
 * instructions when a staged value leaves the range in which the split holds 22 bits (at or above 2^15, non-finite,
 * or a tile that is uniformly tiny).  danra_set_fp32_products(1) makes every product an fp32 instruction.
 * Every function returns 0 on success or a negative DANRA_E* code and never throws;
 * danra_last_error() returns the message of the calling thread's last failure.
 * All kernels are enqueued on the hipStream_t passed as `stream` (void*, 0 = default stream);
 * no call synchronises the host unless its comment says so.
 */
#ifndef DANRA_HIP_H
#define DANRA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DANRA_OK 0
#define DANRA_EINVAL (-1)   /* bad argument / unsupported shape */
#define DANRA_ESTATE (-2)   /* call order violated (e.g. forward before pack) */
#define DANRA_EHIP (-3)     /* a HIP runtime call failed */
#define DANRA_ENOMEM (-4)
#define DANRA_ENAME (-5)    /* unknown or missing state_dict entry */

typedef struct danra_handle_s* danra_handle_t;

/* Model geometry = the constructor arguments of Encoder/Decoder
 * (modules_DANRA_conditional.py:124-135, 472-478) after the input-channel arithmetic of :157-164. */
typedef struct {
    int has_lsm;         /* encoder.lsm buffer registered  -> lsm_cond plane is concatenated  */
    int has_topo;        /* encoder.elevation registered   -> topo_cond plane is concatenated */
    int has_cond;        /* cond_on_img                    -> cond_img plane is concatenated  */
    int num_classes;     /* 0 = no label embedding */
    int time_embedding;  /* 256 in every driver; must be even and <= 512 */
    int out_channels;    /* Decoder output_channels (1) */
    int n_heads;         /* heads of every attention block: 1, 2, 4 or 8 (4 in the DANRA drivers; the clean application's
                          * configuration default is 1, DDPM_clean_application/config.yaml:51) */
    int block_layers[4]; /* BasicBlocks per ResNet layer ([2,2,2,2]) */
    int attn_ff;         /* 0: the network of DDPM_DANRA_conditional/modules_DANRA_conditional.py.
                          * 1: the network of DDPM_clean_application/src/unet.py: every ImageSelfAttention names its
                          *    MultiheadAttention `.mha` (unet.py:87) and ends with the feed-forward block
                          *    x + Linear(GELU(Linear(LayerNorm(x)))) (`.ff`, unet.py:91-96,116); has_lsm / has_topo /
                          *    has_cond then mean Encoder(cond_on_lsm / cond_on_topo / cond_on_img) (unet.py:163-168) */
} danra_config_t;

const char* danra_last_error(void);

/* Replaces Encoder.__init__/Decoder.__init__/DiffusionNet.__init__ (modules:124-199, 472-509, 576-595):
 * creates the parameter table, no device memory yet. */
int danra_create(const danra_config_t* cfg, danra_handle_t* out);
int danra_destroy(danra_handle_t h);

/* The state_dict contract (keys/shapes of the reference's DiffusionNet.state_dict(); drop-in
 * checkpoint compatibility, training_DANRA_conditional.py:755-772).  Entries that carry no
 * arithmetic (num_batches_tracked, encoder.lsm/elevation, final_layer.time_projection_layer) are
 * not listed and are ignored by danra_set_param. */
int danra_num_params(danra_handle_t h);
const char* danra_param_name(danra_handle_t h, int index);
size_t danra_param_numel(danra_handle_t h, int index);

/* Replaces model.load_state_dict(...) (generation_DANRA_conditional.py:354-359): register the device
 * pointer of one state_dict entry in its native PyTorch layout.  The pointer is read by the next
 * danra_pack_weights() only. */
int danra_set_param(danra_handle_t h, const char* name, const float* value_dev, size_t numel);

/* Re-lay every registered parameter into the library's own arena: conv weights OIHW -> O(HW)I with
 * eval-mode BatchNorm folded in (modules:271 + BasicBlock), transposed-conv weights as four 1x1
 * GEMM slabs, LayerNorm affine folded into in_proj.  Must be called after all danra_set_param
 * calls and again whenever a parameter changes. */
int danra_pack_weights(danra_handle_t h, void* stream);

/* Optional override of the two sinusoidal frequency tables (n = time_embedding/2 floats each, host
 * pointers): enc_inv_freq[k] = 1/1000^(2k/dim) (modules:204-207) and dec_div[k] = 10000^(2k/dim)
 * rounded to fp32 (modules:57).  By default the library computes both in double precision. */
int danra_set_freq_tables(danra_handle_t h, const float* enc_inv_freq, const float* dec_div, int n);

/* Replaces DiffusionNet.forward(x, t, y, cond_img, lsm_cond, topo_cond) (modules:597-616) in eval
 * mode.  x/cond/lsm/topo: [B,1,H,W] planes; t_dev: int64[B]; y_dev: int64[B] or NULL (labels outside
 * [0, num_classes) are clamped into it: the caller validates them, as the shipped binding does);
 * eps_out_dev: [B,out_channels,H,W].  H and W must be multiples of 32 (five stride-2 stages) and B*H*W below 2^24
 * pixels per call (e.g. batch 1023 at 128x128), else DANRA_EINVAL. */
int danra_unet_forward(danra_handle_t h, const float* x_dev, const int64_t* t_dev, const int64_t* y_dev,
                       const float* cond_dev, const float* lsm_dev, const float* topo_dev,
                       float* eps_out_dev, int B, int H, int W, void* stream);

/* Copies the five encoder feature maps of the last danra_unet_forward (Encoder.forward's return
 * value, modules:312) to NCHW device buffers; any pointer may be NULL.  For parity tests. */
int danra_get_fmaps(danra_handle_t h, float* f1, float* f2, float* f3, float* f4, float* f5, void* stream);

/* Replaces diffusion_DANRA_conditional.py:155,157: x = c1*(x - c2*eps); x = x + c3*noise with every
 * operation rounded to fp32 as torch evaluates it (noise_dev NULL = zeros, the i==1 step). */
int danra_ddpm_update(float* x_dev, const float* eps_dev, const float* noise_dev, float c1, float c2,
                      float c3, size_t n, void* stream);

/* The same update with z ~ N(0,1) generated in the kernel: Philox4x32-10 keyed by `seed`, counter =
 * (element index / 4, step), Box-Muller.  Stands in for torch.randn_like on the device
 * (diffusion_DANRA_conditional.py:150); n must be a multiple of 4. */
int danra_ddpm_update_philox(float* x_dev, const float* eps_dev, float c1, float c2, float c3, size_t n,
                             uint64_t seed, uint64_t step, void* stream);

#define DANRA_NOISE_HOST 0   /* noise_dev holds (T-2) pre-drawn [B,1,H,W] tensors in loop order (i = T-1 ... 2) */
#define DANRA_NOISE_PHILOX 1 /* Gaussian noise generated inside the update kernel from `seed` */

/* Replaces DiffusionUtils.sample (diffusion_DANRA_conditional.py:105-159): the whole reverse loop
 * i = T-1 ... 1 on the device, x updated in place.  betas/alphas/alpha_hat are HOST fp32 arrays of
 * length T (DiffusionUtils.betas/.alphas/.alpha_hat, diffusion:47-51).  trace_dev (may be NULL)
 * receives x after the update of each timestep listed in trace_steps (host int array, n_trace
 * entries, [n_trace,B,1,H,W] floats).  Asynchronous on `stream`.
 * The loop state (timestep, coefficients, noise cursor) lives on the device, so one denoise step needs
 * no host argument: by default the step (every kernel of one UNet evaluation + the update) is captured
 * once per (B, H, W) into a hipGraph and replayed T-1 times; inputs are copied into the handle's
 * workspace first, so the captured step never refers to caller memory. */
int danra_sample(danra_handle_t h, float* x_dev, const int64_t* y_dev, const float* cond_dev,
                 const float* lsm_dev, const float* topo_dev, int B, int H, int W, int T,
                 const float* betas, const float* alphas, const float* alpha_hat, int noise_mode,
                 const float* noise_dev, uint64_t seed, const int* trace_steps, int n_trace,
                 float* trace_dev, void* stream);

/* Replaces DiffusionUtils.sample of DDPM_clean_application/src/diffusion_modules.py:101-186: the same loop with
 *  - classifier-free guidance (cfg_scale > 0, :156-160): every step evaluates the network twice, with and without the
 *    label (no conditioning planes in either call, so the network must have been built without any), and uses
 *    torch.lerp(eps_uncond, eps_cond, cfg_scale);
 *  - noise_scale: every Gaussian draw is multiplied by it before use (`noise *= 0.005` when data_scaled, :173-174;
 *    pass 1 otherwise).  Scaling the start image (:131-132) is the caller's job, x_dev is taken as given.
 * danra_sample(...) == danra_sample_guided(..., cfg_scale = 0, noise_scale = 1, ...). */
int danra_sample_guided(danra_handle_t h, float* x_dev, const int64_t* y_dev, const float* cond_dev,
                        const float* lsm_dev, const float* topo_dev, int B, int H, int W, int T,
                        const float* betas, const float* alphas, const float* alpha_hat, int noise_mode,
                        const float* noise_dev, uint64_t seed, float cfg_scale, float noise_scale,
                        const int* trace_steps, int n_trace, float* trace_dev, void* stream);

/* Replaces DiffusionUtils.noiseImage (diffusion_DANRA_conditional.py:85-103), the forward process used by the
 * training step: noisy = sqrt(alpha_hat[t_b]) * x + sqrt(1 - alpha_hat[t_b]) * noise for image b, every operation
 * rounded to fp32 as torch evaluates the three tensor expressions.  alpha_hat_dev: DiffusionUtils.alpha_hat (device,
 * T floats); t_dev: B int64 timesteps (clamped to [0, T-1]); per_image = C*H*W, a multiple of 4.
 * DANRA_NOISE_HOST reads the Gaussian draws from noise_dev ([B, C, H, W]); DANRA_NOISE_PHILOX draws them in the
 * kernel (Philox4x32-10 keyed by `seed`, counter = (element index / 4, offset), Box-Muller) and writes them to
 * noise_dev, which is the second value noiseImage returns (the regression target of the loss). */
int danra_noise_image(const float* x_dev, const int64_t* t_dev, const float* alpha_hat_dev, int T, float* noisy_out,
                      float* noise_dev, int B, size_t per_image, int noise_mode, uint64_t seed, uint64_t offset,
                      void* stream);

/* The scaled-dot-product core of ImageSelfAttention's nn.MultiheadAttention (modules_DANRA_conditional.py:100-104:
 * softmax(q k^T / sqrt(head_dim)) v per head, dropout 0) for the TRAINING step (training_DANRA_conditional.py:963-984),
 * where the in/out projections stay differentiable torch ops and this pair replaces the core's forward and its autograd
 * backward without ever forming the L x L matrices.  head_dim must be 16 (the 64-channel levels of the 4-head network:
 * L = 256 / 1024 at 64x64, 1024 / 4096 at 128x128) and L a multiple of 128, else DANRA_EINVAL.
 * q, k, v: [B*heads][L][16] (projection outputs, head-split, unscaled); out: tokens [B][L][heads*16], head h at columns
 * 16h..16h+15 (the layout out_proj consumes); lse2 (may be NULL when no backward follows): [B*heads][L], the
 * log-sum-exp of every query's scaled scores in the exp2 domain. */
int danra_attention_core_fwd(const float* q_dev, const float* k_dev, const float* v_dev, float* out_dev, float* lse2_dev,
                             int B, int heads, int L, int head_dim, void* stream);
/* The head_dim-16 core in the form the network's inference path runs it (modules:100-107, between the in- and out-projection
 * of nn.MultiheadAttention): K and V are first brought to the pre-split layout the QKV projection kernels of the network
 * write directly (fp16 halves in the order of the core's tiles, one range flag per 16 keys), then the core reads those.
 * Same arguments and result as danra_attention_core_fwd without the log-sum-exp; head_dim 16, L a multiple of 128.
 * Allocates a temporary buffer and waits for the stream: an entry for unit tests. */
int danra_attention_core_presplit_fwd(const float* q_dev, const float* k_dev, const float* v_dev, float* out_dev, int B,
                                      int heads, int L, int head_dim, void* stream);
/* Numerics switch (no counterpart in the reference; process-wide).  Two kernel families evaluate their fp32 products
 * on the fp16 matrix cores with every operand split in two fp16 halves (22 significant bits per operand, fp32
 * accumulation; operands outside the range where that holds are multiplied with fp32 instructions): the head_dim-16
 * attention core and the implicit GEMM behind the convolutions and projections with K >= 128.  fp32_products != 0
 * makes every later launch of both (danra_unet_forward, danra_sample, danra_attention_core_fwd, the per-op entry
 * points) use fp32 matrix instructions only (slower); 0 restores the default.  Both forms meet the same tolerances
 * against float64; the switch exists so that a caller can cross-check one against the other. */
int danra_set_fp32_products(int fp32_products);
/* Gradients of danra_attention_core_fwd: dout_dev = dLoss/dout (layout of out) -> dq, dk, dv (layout of q, k, v).
 * out_dev and lse2_dev are the forward's results; delta_dev: [B*heads][L] scratch.  Two kernels (one accumulates dq
 * over the keys, one dk and dv over the queries), no atomics: results are deterministic. */
int danra_attention_core_bwd(const float* q_dev, const float* k_dev, const float* v_dev, const float* out_dev,
                             const float* dout_dev, const float* lse2_dev, float* dq_dev, float* dk_dev, float* dv_dev,
                             float* delta_dev, int B, int heads, int L, int head_dim, void* stream);
/* The same core, forward and backward, on the output of the block's fused QKV projection used in place (no copy into
 * per-head tensors): qkv_dev, dqkv_dev [B][L][3][heads*16] - what nn.MultiheadAttention's in_proj produces for batch_first
 * tokens (torch.nn.functional.multi_head_attention_forward's _in_projection_packed) - q of head h at columns 16h .. 16h+15
 * of part 0, k and v in parts 1 and 2.  out, dout, lse2, delta as above. */
int danra_attention_core_qkv_fwd(const float* qkv_dev, float* out_dev, float* lse2_dev, int B, int heads, int L, int head_dim,
                                 void* stream);
int danra_attention_core_qkv_bwd(const float* qkv_dev, const float* out_dev, const float* dout_dev, const float* lse2_dev,
                                 float* dqkv_dev, float* delta_dev, int B, int heads, int L, int head_dim, void* stream);

/* The scaled-dot-product core of the short-sequence attention blocks in a training step (the deep levels: nn.MultiheadAttention
 * between its in- and out-projection, modules:100-107, at 64 / 16 / 4 tokens with head_dim 32 / 64 / 128): softmax(q k^T /
 * sqrt(head_dim)) v per (image, head), forward and the gradients of q, k and v.  q, k, v, dq, dk, dv [B*heads][L][head_dim];
 * out, dout [B][L][heads*head_dim] (token order).  1 <= L <= 64, head_dim a multiple of 4 up to 128, L * head_dim <= 2048. */
int danra_attention_small_supported(int B, int heads, int L, int head_dim);
int danra_attention_small_fwd(const float* q_dev, const float* k_dev, const float* v_dev, float* out_dev, int B, int heads, int L,
                              int head_dim, void* stream);
int danra_attention_small_bwd(const float* q_dev, const float* k_dev, const float* v_dev, const float* dout_dev, float* dq_dev,
                              float* dk_dev, float* dv_dev, int B, int heads, int L, int head_dim, void* stream);
/* Both on the output of the block's fused QKV projection used in place, qkv_dev, dqkv_dev [B][L][3][heads*head_dim] (see
 * danra_attention_core_qkv_fwd). */
int danra_attention_small_qkv_fwd(const float* qkv_dev, float* out_dev, int B, int heads, int L, int head_dim, void* stream);
int danra_attention_small_qkv_bwd(const float* qkv_dev, const float* dout_dev, float* dqkv_dev, int B, int heads, int L,
                                  int head_dim, void* stream);
/* nn.BatchNorm2d in TRAIN mode (the encoder's bn1 and the BasicBlocks' bn1 / bn2 / downsample.1 after model.train(),
 * training_DANRA_conditional.py:517; modules_DANRA_conditional.py:271 and torchvision's BasicBlock.forward): batch
 * statistics per channel over (B, H, W), y = (x - mean) / sqrt(var_biased + eps) * gamma + beta, optionally followed by the
 * ReLU that every such norm of the network except bn2 is followed by; running_mean / running_var (may both be NULL) move by
 * `momentum` towards the batch mean / unbiased batch variance.  x, y: NCHW fp32 [B][C][HW] contiguous, HW a multiple
 * of 4; save_mean / save_rstd: [C], kept for the backward pass; num_batches_tracked_dev (may be NULL): the module's int64
 * counter of that name, incremented by one on the device (torch.nn.modules.batchnorm._BatchNorm.forward does so per call). */
int danra_batchnorm_train_fwd(const float* x_dev, const float* gamma_dev, const float* beta_dev, float* y_dev,
                              float* save_mean_dev, float* save_rstd_dev, float* running_mean_dev, float* running_var_dev,
                              int B, int C, int HW, float eps, float momentum, int relu, long long* num_batches_tracked_dev,
                              void* stream);
/* Its autograd backward: dy -> dx, dgamma, dbeta.  y_dev (the forward's output) is read only when relu != 0. */
int danra_batchnorm_train_bwd(const float* x_dev, const float* dy_dev, const float* y_dev, const float* gamma_dev,
                              const float* save_mean_dev, const float* save_rstd_dev, float* dx_dev, float* dgamma_dev,
                              float* dbeta_dev, int B, int C, int HW, int relu, void* stream);

/* The tail of a torchvision BasicBlock in train mode, relu(bn2(z) + shortcut) (modules_DANRA_conditional.py:154,276-303), as
 * one kernel in either direction: y = relu(BatchNorm_train(x) + res); backward: dres = dy where y > 0 (else 0), dx / dgamma /
 * dbeta as danra_batchnorm_train_bwd for that masked gradient.  Same shapes and restrictions as danra_batchnorm_train_fwd. */
int danra_batchnorm_train_add_relu_fwd(const float* x_dev, const float* res_dev, const float* gamma_dev, const float* beta_dev,
                                       float* y_dev, float* save_mean_dev, float* save_rstd_dev, float* running_mean_dev,
                                       float* running_var_dev, int B, int C, int HW, float eps, float momentum,
                                       long long* num_batches_tracked_dev, void* stream);
int danra_batchnorm_train_add_relu_bwd(const float* x_dev, const float* dy_dev, const float* y_dev, const float* gamma_dev,
                                       const float* save_mean_dev, const float* save_rstd_dev, float* dx_dev, float* dres_dev,
                                       float* dgamma_dev, float* dbeta_dev, int B, int C, int HW, void* stream);

/* out[c] = sum over b, h, w of x[b][c][h][w] on contiguous NCHW fp32 (HW a multiple of 4): the bias gradient of the decoder's
 * nn.Conv2d / nn.ConvTranspose2d layers (modules_DANRA_conditional.py:407-417; autograd's dy.sum((0, 2, 3))), summed in a fixed
 * order.  workspace_dev (may be NULL): danra_channel_sum_workspace(B, C, HW) bytes; with it a large tensor is summed by several
 * workgroups per channel and their partial sums are merged in order by a second kernel (0 bytes: not needed for the shape). */
size_t danra_channel_sum_workspace(int B, int C, int HW);
int danra_channel_sum(const float* x_dev, float* out_dev, int B, int C, int HW, void* workspace_dev, void* stream);

/* nn.InstanceNorm2d(affine=False) as the decoder blocks use it (modules_DANRA_conditional.py:412,419) on NCHW fp32 with
 * its autograd backward, for the training step: `planes` = B*C planes of HW contiguous values each (HW a multiple of 4);
 * y = (x - mean) / sqrt(var_biased + eps) per plane; save_mean / save_rstd: [planes], kept for the backward pass
 * (dx = rstd * (dy - mean(dy) - xhat * mean(dy * xhat))). */
int danra_instancenorm_fwd(const float* x_dev, float* y_dev, float* save_mean_dev, float* save_rstd_dev, int planes, int HW,
                           float eps, void* stream);
int danra_instancenorm_bwd(const float* x_dev, const float* dy_dev, const float* save_mean_dev, const float* save_rstd_dev,
                           float* dx_dev, int planes, int HW, void* stream);

/* Weight and bias gradient of y = x W^T + b over M rows (torch.nn.functional.linear's backward for its weight and bias, as
 * autograd runs it for the attention projections nn.MultiheadAttention holds, modules_DANRA_conditional.py:84,105, in the
 * training step training_DANRA_conditional.py:962-984): dw[N][K] = dy^T x, db[N] = column sums of dy (db may be NULL).
 * dy: [M][N], x: [M][K], row-major fp32.  Supported: K = 64 and N = 64, 128 or 192 (danra_linear_wgrad_supported);
 * workspace: danra_linear_wgrad_workspace(M, N, K) bytes of device memory, the caller's (nothing is allocated). */
int danra_linear_wgrad_supported(int M, int N, int K);
size_t danra_linear_wgrad_workspace(int M, int N, int K);
int danra_linear_wgrad(const float* dy_dev, const float* x_dev, float* dw_dev, float* db_dev, int M, int N, int K,
                       void* workspace_dev, void* stream);

/* y = x W^T + b on M rows: torch.nn.functional.linear as nn.MultiheadAttention applies its in_proj / out_proj
 * (modules_DANRA_conditional.py:84,105), and - called with W^T - the input gradient dx = dy W of such a layer in the
 * training step (training_DANRA_conditional.py:962-984).  x: [M][K], w: [N][K], bias: [N] or NULL, y: [M][N], row-major
 * fp32; fp32 matrix instructions.  Supported: N a multiple of 64, K a multiple of 32 (danra_linear_supported). */
int danra_linear_supported(int M, int N, int K);
int danra_linear(const float* x_dev, const float* w_dev, const float* bias_dev, float* y_dev, int M, int N, int K,
                 void* stream);

/* The 3x3, stride-1, padding-1 convolutions of the training step (torchvision BasicBlock conv1 / conv2 of stride 1,
 * modules_DANRA_conditional.py:154,276-303, and the decoder blocks' nn.Conv2d(k=3, padding=1), modules_DANRA_conditional.py:417)
 * as autograd runs them in training_DANRA_conditional.py:962-984: forward y = conv(x, w), data gradient
 * dx = conv_transpose(dy, w), weight gradient dw = sum over batch and pixels of dy (x) x-windows.  Contiguous NCHW fp32
 * tensors and the [Cout][Cin][3][3] weight as nn.Conv2d holds it; no bias (the caller adds it).  Supported
 * (danra_conv3x3_train_supported): Cin a multiple of 32, Cout a multiple of 64, W a power of two in 2..64, and either
 * H*W >= 64 with H a multiple of 64/W, or H*W a divisor of 64.  workspace: danra_conv3x3_train_workspace(...) bytes of
 * device memory, the caller's (nothing is allocated); sums over workgroups are formed in a fixed order (bit-reproducible). */
int danra_conv3x3_train_supported(int B, int Cin, int Cout, int H, int W);
size_t danra_conv3x3_train_workspace(int B, int Cin, int Cout, int H, int W);
int danra_conv3x3_train_fwd(const float* x_dev, const float* w_dev, float* y_dev, int B, int Cin, int Cout, int H, int W,
                            void* workspace_dev, void* stream);
int danra_conv3x3_train_dgrad(const float* dy_dev, const float* w_dev, float* dx_dev, int B, int Cin, int Cout, int H, int W,
                              void* workspace_dev, void* stream);
int danra_conv3x3_train_wgrad(const float* x_dev, const float* dy_dev, float* dw_dev, int B, int Cin, int Cout, int H, int W,
                              void* workspace_dev, void* stream);

/* The network's last convolution in a training step: 3x3, stride 1, padding 1, Cin channels -> one output channel
 * (modules:415-416 with last_layer: nn.Conv2d(64, 1, 3, padding=1)), on contiguous NCHW fp32 tensors, forward, data gradient
 * and weight gradient (the bias is torch's): x, dx [B][Cin][H][W]; w, dw [1][Cin][3][3]; y, dy [B][1][H][W].
 * workspace (weight gradient only): danra_conv3x3_c1_train_workspace(...) bytes of device memory. */
int danra_conv3x3_c1_train_supported(int B, int Cin, int H, int W);
size_t danra_conv3x3_c1_train_workspace(int B, int Cin, int H, int W);
int danra_conv3x3_c1_train_fwd(const float* x_dev, const float* w_dev, float* y_dev, int B, int Cin, int H, int W, void* stream);
int danra_conv3x3_c1_train_dgrad(const float* dy_dev, const float* w_dev, float* dx_dev, int B, int Cin, int H, int W, void* stream);
int danra_conv3x3_c1_train_wgrad(const float* x_dev, const float* dy_dev, float* dw_dev, int B, int Cin, int H, int W,
                                 void* workspace, void* stream);
/* The 3x3, stride-2, padding-1 convolutions of the training step (conv1 of the first BasicBlock of layer2 / layer3 / layer4,
 * modules_DANRA_conditional.py:154 through torchvision's _make_layer) on contiguous NCHW fp32 tensors with the
 * [Cout][Cin][3][3] weight as torch holds it: x, dx [B][Cin][H][W]; y, dy [B][Cout][H/2][W/2].  Shapes
 * (danra_conv3x3s2_train_supported): H, W even, W/2 a power of two in 2..64, (H/2)(W/2) a divisor of 64 or a multiple of it by
 * whole rows, Cin a multiple of 32, Cout a multiple of 64.  workspace: danra_conv3x3s2_train_workspace(...) bytes of device
 * memory (slabs of the sums that run over workgroups, added in a fixed order: bit-reproducible). */
int danra_conv3x3s2_train_supported(int B, int Cin, int Cout, int H, int W);
size_t danra_conv3x3s2_train_workspace(int B, int Cin, int Cout, int H, int W);
int danra_conv3x3s2_train_fwd(const float* x_dev, const float* w_dev, float* y_dev, int B, int Cin, int Cout, int H, int W,
                              void* workspace_dev, void* stream);
int danra_conv3x3s2_train_dgrad(const float* dy_dev, const float* w_dev, float* dx_dev, int B, int Cin, int Cout, int H, int W,
                                void* workspace_dev, void* stream);
int danra_conv3x3s2_train_wgrad(const float* x_dev, const float* dy_dev, float* dw_dev, int B, int Cin, int Cout, int H, int W,
                                void* workspace_dev, void* stream);

/* The two stem convolutions of the encoder in a training step: 8x8, stride 2, padding 3, no bias
 * (modules_DANRA_conditional.py:186-191: conv1 on the 1-4 input planes, conv2 64 -> 64), on contiguous NCHW fp32 tensors with
 * the [Cout][Cin][8][8] weight as torch holds it: x, dx [B][Cin][H][W]; y, dy [B][Cout][H/2][W/2].  Shapes
 * (danra_conv8x8s2_train_supported): H, W even, W/2 a power of two in 8..64, H/2 a multiple of 64 / (W/2), Cout a multiple of
 * 64, Cin a multiple of 32 or at most 4; the data gradient exists for Cin a multiple of 32 only (_dgrad_supported: the
 * gradient with respect to the network's input planes is never asked for).  workspace (weight gradient only):
 * danra_conv8x8s2_train_workspace(...) bytes of device memory; sums over workgroups in a fixed order (bit-reproducible). */
int danra_conv8x8s2_train_supported(int B, int Cin, int Cout, int H, int W);
int danra_conv8x8s2_train_dgrad_supported(int B, int Cin, int Cout, int H, int W);
size_t danra_conv8x8s2_train_workspace(int B, int Cin, int Cout, int H, int W);
int danra_conv8x8s2_train_fwd(const float* x_dev, const float* w_dev, float* y_dev, int B, int Cin, int Cout, int H, int W,
                              void* stream);
int danra_conv8x8s2_train_dgrad(const float* dy_dev, const float* w_dev, float* dx_dev, int B, int Cin, int Cout, int H, int W,
                                void* stream);
int danra_conv8x8s2_train_wgrad(const float* x_dev, const float* dy_dev, float* dw_dev, int B, int Cin, int Cout, int H, int W,
                                void* workspace_dev, void* stream);
/* Entry and exit of ImageSelfAttention.forward (modules_DANRA_conditional.py:93-110) in the training step, on the NCHW map the
 * block receives and returns (L = H*W tokens of C channels):
 *   danra_ln_tokens_fwd : normed[b][l][c] = LayerNorm_C(x[b][.][l])[c] * gamma[c] + beta[c]   (x.view(B, C, L).swapaxes(1, 2)
 *                         followed by nn.LayerNorm(C), :93-97); mean / rstd: [B*L], kept for the backward pass;
 *   danra_ln_tokens_bwd : dx (NCHW), and dgamma_dbeta [2][C], from the gradient of `normed`; workspace:
 *                         danra_ln_tokens_workspace(B, C, L) bytes of the caller's device memory;
 *   danra_tokens_to_nchw_add : out[b][c][l] = add[b][c][l] + tokens[b][l][c]   (the residual and the way back, :107-110;
 *                         add may be NULL).  Its backward pass is danra_nchw_to_nhwc of the incoming gradient.
 * Supported (danra_ln_tokens_supported): C a multiple of 4, at most 512; L a multiple of 32, or 4, 8 or 16. */
int danra_ln_tokens_supported(int B, int C, int L);
size_t danra_ln_tokens_workspace(int B, int C, int L);
int danra_ln_tokens_fwd(const float* x_dev, const float* gamma_dev, const float* beta_dev, float* normed_dev, float* mean_dev,
                        float* rstd_dev, int B, int C, int L, float eps, void* stream);
int danra_ln_tokens_bwd(const float* dn_dev, const float* x_dev, const float* mean_dev, const float* rstd_dev,
                        const float* gamma_dev, float* dx_dev, float* dgamma_dbeta_dev, int B, int C, int L, void* workspace_dev,
                        void* stream);
int danra_tokens_to_nchw_add(const float* tokens_dev, const float* add_dev, float* out_dev, int B, int C, int L, void* stream);

/* The two stride-2 "pointwise" convolutions of the training step (training_DANRA_conditional.py:962-984), forward, data
 * gradient and weight gradient, on contiguous NCHW fp32 tensors with torch's weight layouts; H, W = the layer's INPUT map:
 *   kind DANRA_PW_CONV1X1_S2 (0): nn.Conv2d(Cin, Cout, 1, stride 2, bias=False), the shortcut of a BasicBlock that halves
 *     the map (torchvision's downsample[0], built by modules_DANRA_conditional.py:154); w [Cout][Cin][1][1];
 *     y [B][Cout][H/2][W/2].  Supported: H, W even, Cin and Cout multiples of 64.
 *   kind DANRA_PW_CONVT2X2 (1): nn.ConvTranspose2d(Cin, Cout, 2, stride 2), the up-sampling of a decoder block
 *     (modules_DANRA_conditional.py:408-411); w [Cin][Cout][2][2]; y [B][Cout][2H][2W]; `bias_dev` ([Cout] or NULL) is added
 *     by the forward call.  Supported: Cin a multiple of 64, Cout of 16.
 * The two gradients take danra_pointwise_train_workspace(...) bytes of the caller's device memory (slabs added in a fixed
 * order); nothing is allocated. */
#define DANRA_PW_CONV1X1_S2 0
#define DANRA_PW_CONVT2X2 1
int danra_pointwise_train_supported(int kind, int B, int Cin, int Cout, int H, int W);
size_t danra_pointwise_train_workspace(int kind, int B, int Cin, int Cout, int H, int W);
int danra_pointwise_train_fwd(int kind, const float* x_dev, const float* w_dev, const float* bias_dev, float* y_dev, int B,
                              int Cin, int Cout, int H, int W, void* stream);
int danra_pointwise_train_dgrad(int kind, const float* dy_dev, const float* w_dev, float* dx_dev, int B, int Cin, int Cout,
                                int H, int W, void* workspace_dev, void* stream);
int danra_pointwise_train_wgrad(int kind, const float* x_dev, const float* dy_dev, float* dw_dev, int B, int Cin, int Cout,
                                int H, int W, void* workspace_dev, void* stream);

/* Batch sharding of the in-kernel noise (no counterpart in the reference, which samples on one device): the calls
 * to danra_sample / danra_sample_guided that follow treat their image 0 as image `first_image` of a larger batch:
 * with DANRA_NOISE_PHILOX the draw for element e of local image b at a timestep is that of global element
 * (first_image + b) * H*W + e, so the slices of one batch sampled by several processes with one seed draw
 * different noise, and the same noise as the unsliced batch would on one device.  Default 0.  Sticky per handle. */
int danra_set_first_image(danra_handle_t h, uint64_t first_image);

/* Sampling-loop execution mode (no counterpart in the reference): enable != 0 replays a captured
 * hipGraph per denoise step (default; the environment variable DANRA_GRAPH=0 at danra_create time turns
 * it off), 0 enqueues every kernel of every step.  Both modes run the same kernels on the same data and
 * give bit-identical results. */
int danra_set_graph_mode(danra_handle_t h, int enable);

/* Measurement hook (no counterpart in the reference, which holds no timing code): between _begin and
 * _end every launch of the named kernel family is bracketed by HIP events on the stream it is launched
 * on.  _end waits for the last event (host sync) and returns the summed kernel time, the number of
 * launches and their algorithmic FLOPs (2 per multiply-add of QK^T and PV).  danra_unet_forward times every
 * launch; danra_sample times the launches of every 64th denoise step (that step is enqueued kernel by kernel, the
 * others are replayed from the graph as usual), so that the event records do not slow the loop being measured. */
#define DANRA_PROF_FLASH_ATTENTION 1 /* the flash attention kernels (head_dim 16; head_dim 32/64 with 2 heads or 1 head) */
#define DANRA_PROF_IGEMM 2           /* the implicit-GEMM family: every convolution (but the stem's x plane), transposed
                                      * convolution and projection, each launch with its split-K reduction kernel;
                                      * algorithmic FLOPs = 2*M*N*K per launch */
int danra_profile_begin(danra_handle_t h, int kind);
int danra_profile_end(danra_handle_t h, double* total_ms, long long* launches, double* total_flops);
/* Of the launches counted by the last _begin/_end pair: the FLOPs the issued matrix instructions executed (a
 * split-operand launch executes three fp16 products per algorithmic fp32 one, the head_dim-16 attention core 3.5) and
 * the algorithmic FLOPs that ran in the split-operand form.  Either pointer may be NULL. */
int danra_profile_executed(danra_handle_t h, double* executed_flops, double* split_flops);
/* Valid after danra_profile_end: the algorithmic bytes of the timed launches - every operand, weight and result of a
 * launch counted once at 4 bytes per element (what a launch must move if nothing were cached across launches and
 * nothing re-read within one). */
int danra_profile_bytes(danra_handle_t h, double* algorithmic_bytes);

/* ---- per-op entry points (unit parity tests of the kernels; NHWC = [B][H][W][C] fp32) ---- */

/* Sinusoidal embeddings + label embedding + the nine Sequential(SiLU, Linear) projections
 * (modules:42-63, 203-211, 243-256, 262-305, 449-451).  enc_emb_out/dec_emb_out: [B][time_embedding]
 * (may be NULL); tbias_out: [B][danra_tbias_width()] laid out encoder level 0..4 then decoder block
 * 0..3 (may be NULL). */
int danra_tbias_width(danra_handle_t h);
int danra_time_embed(danra_handle_t h, const int64_t* t_dev, const int64_t* y_dev, int B,
                     float* enc_emb_out, float* dec_emb_out, float* tbias_out, void* stream);

/* ImageSelfAttention.forward (modules:91-110; with attn_ff, DDPM_clean_application/src/unet.py:98-119) of attention
 * layer `index` (0..4 encoder levels, 5..8 decoder blocks) on NHWC tokens; relu = the decoder's trailing
 * activation (modules:459). */
int danra_attention(danra_handle_t h, int index, const float* x_nhwc, float* out_nhwc, int B, int HW,
                    int relu, void* stream);

/* Encoder.conv1 (modules:178-183, 260) on up to four [B,1,H,W] planes in concatenation order;
 * out NHWC [B][H/2][W/2][64]. */
int danra_conv1(danra_handle_t h, const float* x, const float* lsm, const float* topo, const float* cond,
                float* out_nhwc, int B, int H, int W, void* stream);
/* Encoder.conv2 + bn1 + relu (modules:186-191, 269-273): NHWC [B][H][W][64] -> [B][H/2][W/2][64]. */
int danra_conv2_bn_relu(danra_handle_t h, const float* x_nhwc, float* out_nhwc, int B, int H, int W, void* stream);
/* ResNet layer `level` (1..4) = its BasicBlocks (modules:276,285,294,303): NHWC in/out. */
int danra_resnet_layer(danra_handle_t h, int level, const float* x_nhwc, float* out_nhwc, int B, int H, int W, void* stream);
/* DecoderBlock.forward (modules:425-460): block 0..3 = residual_layers, 4 = final_layer (skip/t ignored).
 * fmap NHWC [B][H][W][Cin] -> out NHWC [B][2H][2W][Cout]; t_dev int64[B]. */
int danra_decoder_block(danra_handle_t h, int block, const float* fmap_nhwc, const float* skip_nhwc,
                        const int64_t* t_dev, float* out_nhwc, int B, int H, int W, void* stream);
/* One 3x3 stride-1 padding-1 convolution as the BasicBlocks (torchvision resnet.BasicBlock under modules:276-303) and
 * the DecoderBlocks (modules:415-416, 437) apply it, through the window-staged kernel, for unit tests: NHWC x
 * [B][H][W][Cin], w [Cout][3][3][Cin] (OIHW permuted to taps-then-channels), optional bias [Cout], residual (NHWC, added
 * before the optional ReLU) and InstanceNorm of the input on load (nmean, nrstd [B][Cin]: (x - mean) * rstd, zero
 * padding applied after it, as modules:436-437 compose them).  H, W powers of two (W <= 64), Cin % 64 == 0,
 * Cout % 32 == 0.  Allocates a temporary copy of the weights and waits for the stream. */
int danra_conv3x3(const float* x_nhwc, const float* w, const float* bias, const float* res_nhwc, const float* nmean,
                  const float* nrstd, float* out_nhwc, int B, int H, int W, int Cin, int Cout, int relu, void* stream);
/* One 3x3 stride-2 padding-1 convolution as the first convolution of a down-sampling BasicBlock applies it (torchvision
 * resnet.BasicBlock.conv1 under modules:285-303), through the window-staged kernel, for unit tests: NHWC x [B][H][W][Cin]
 * -> [B][H/2][W/2][.], w [Cout][3][3][Cin], optional bias and ReLU.  With out2 the output channels from n_split on go to
 * out2 ([B][H/2][W/2][Cout - n_split]) with their bias and no ReLU, the first n_split to out: how the library runs the
 * block's 1x1 stride-2 shortcut in the same launch (its weights as the centre tap of a 3x3 kernel; the other eight taps of
 * those channels are taken to be zero and are not read).
 * H, W powers of two with H*W/4 <= 64, Cin % 64 == 0, Cout % 32 == 0.  Allocates a temporary copy of the weights and waits
 * for the stream. */
int danra_conv3x3s2(const float* x_nhwc, const float* w, const float* bias, float* out_nhwc, float* out2_nhwc, int B, int H,
                    int W, int Cin, int Cout, int n_split, int relu, void* stream);
/* One 8x8 stride-2 padding-3 convolution as Encoder.conv2 applies it (modules:186-191, 269-273; bn1 folded into w and
 * bias by the caller), through the strided window kernel, for unit tests: NHWC x [B][H][W][Cin] -> [B][H/2][W/2][Cout],
 * w [Cout][8][8][Cin], optional bias, residual (NHWC, added before the optional ReLU).  H, W powers of two >= 16,
 * Cin % 64 == 0, Cout % 64 == 0.  Allocates a temporary copy of the weights and waits for the stream. */
int danra_conv8x8s2(const float* x_nhwc, const float* w, const float* bias, const float* res_nhwc, float* out_nhwc, int B,
                    int H, int W, int Cin, int Cout, int relu, void* stream);
/* InstanceNorm2d(affine=False, eps=1e-5) (modules:412,419) on NHWC. */
int danra_instance_norm(const float* x_nhwc, float* out_nhwc, int B, int HW, int C, void* stream);
/* danra_instance_norm for callers that cannot allocate device memory (code under hipGraph capture, serving loops,
 * harnesses that forbid hipMalloc).  It has no counterpart in the reference.  Same arithmetic and results as
 * danra_instance_norm, but the statistics scratch is a 4 MiB array built into the library (one copy per GPU, held until
 * the process ends) instead of a hipMalloc per call, and the call is asynchronous on `stream` (no host synchronisation).
 * Refused with DANRA_EINVAL when 2*B*C + 16384 (+ 2*B*C*S for maps above 1024 pixels, S = HW/1024 rounded up to a
 * power of two) exceeds 2^20 floats.  All calls on one GPU share the array: calls that overlap on one GPU (different
 * streams or threads) race; calls on one stream are ordered.  The array's address is resolved once per device, by
 * danra_create or by the first call on that device (which must therefore not be made under stream capture). */
int danra_instance_norm_noalloc(const float* x_nhwc, float* out_nhwc, int B, int HW, int C, void* stream);
/* ---- The elucidated (Karras et al., EDM) sampler of the cascaded model: the arithmetic of one sampling step around
 * the network evaluations (CascadedDDPM/elucidated_imagen.py:393-545, ElucidatedImagen.one_unet_sample).  Images are
 * NCHW fp32 on the device, [B][C][HW]; an inpainting mask is one byte per image and pixel, [B][HW], shared by the
 * channels (the reference's `b 1 h w` boolean, :462).  The network itself is the caller's: between these calls it
 * maps net_in to net_out.  Scalars (c_in, c_skip, c_out of :325-332, step sizes) are formed by the host exactly as the
 * reference forms them and passed as floats.  Every statement is evaluated in fp32, one rounding per operation, in the
 * reference's order. */
/* :489-499 and :359.  eps = s_noise * z; added = noise_scale * eps (noise_scale = sqrt(sigma_hat^2 - sigma^2));
 * images_hat = images + added, with inpaint + added in the masked pixels when inpaint/mask are given (both or neither);
 * net_in = c_in * images_hat.  z is read from noise_dev (DANRA_NOISE_HOST) or drawn from Philox(seed, offset). */
int danra_edm_perturb(const float* images_dev, const float* noise_dev, const float* inpaint_dev,
                      const unsigned char* mask_dev, float* images_hat_dev, float* net_in_dev, int B, int C, int HW,
                      float s_noise, float noise_scale, float c_in, int noise_mode, uint64_t seed, uint64_t offset,
                      void* stream);
/* preconditioned_network_forward, :364-367 (+ :311 when clamp != 0): model_out = c_skip * x + c_out * net_out,
 * clamped to [-1, 1] if `clamp`.  Used on its own when a dynamic threshold must be taken from model_out before the
 * step; otherwise danra_edm_euler / danra_edm_heun form it themselves from net_out. */
int danra_edm_precondition(const float* x_dev, const float* net_out_dev, float* model_out_dev, size_t n, float c_skip,
                           float c_out, int clamp, void* stream);
/* threshold_x_start, :313-319: s[b] = max(quantile(|x[b]|, percentile), 1) with torch.quantile's linear
 * interpolation between the two neighbouring order statistics, which are found exactly (radix select).
 * workspace_dev: danra_edm_dynamic_threshold_workspace(B, per_image) bytes of device memory, contents irrelevant before
 * and after the call (0 bytes for images of at most 32768 values: then it may be null). */
size_t danra_edm_dynamic_threshold_workspace(int B, size_t per_image);
int danra_edm_dynamic_threshold(const float* x_dev, float* s_dev, int B, size_t per_image, float percentile,
                                void* workspace_dev, void* stream);
/* :509-511 (+ :321, :359).  model_out: formed from net_out_dev as in danra_edm_precondition when net_out_dev is given,
 * else read from model_out_dev; thresholds_dev (nullable, one float per image) applies clamp(-s, s) / s to it; the
 * result is written back to model_out_dev.  d = (images_hat - model_out) / sigma_hat (to d_dev, nullable);
 * images_next = images_hat + dt * d; net_in_next = c_in_next * images_next (nullable: the last step). */
int danra_edm_euler(const float* images_hat_dev, const float* net_out_dev, float* model_out_dev,
                    const float* thresholds_dev, float* d_dev, float* images_next_dev, float* net_in_next_dev, int B,
                    size_t per_image, float c_skip, float c_out, int clamp, float sigma_hat, float dt, float c_in_next,
                    void* stream);
/* :520-529.  model_out (of the evaluation at images_next, sigma_next) as in danra_edm_euler, with images_next as x;
 * d' = (images_next - model_out) / sigma_next; images = images_hat + half_dt * (d + d'). */
int danra_edm_heun(const float* images_hat_dev, const float* images_next_dev, const float* net_out_dev,
                   float* model_out_dev, const float* thresholds_dev, const float* d_dev, float* images_dev, int B,
                   size_t per_image, float c_skip, float c_out, int clamp, float sigma_next, float half_dt, void* stream);
/* :535-536 (repaint): images += scale * z. */
int danra_edm_renoise(float* images_dev, const float* noise_dev, size_t n, float scale, int noise_mode, uint64_t seed,
                      uint64_t offset, void* stream);
/* :540-545: clamp to [-1, 1]; inpaint in the masked pixels (both or neither); (x + 1) / 2 if `unnormalize`. */
int danra_edm_finish(float* images_dev, const float* inpaint_dev, const unsigned char* mask_dev, int B, int C, int HW,
                     int unnormalize, void* stream);
/* NCHW <-> NHWC helpers for the tests. */
int danra_nchw_to_nhwc(const float* src, float* dst, int B, int C, int HW, void* stream);
int danra_nhwc_to_nchw(const float* src, float* dst, int B, int C, int HW, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DANRA_HIP_H */
