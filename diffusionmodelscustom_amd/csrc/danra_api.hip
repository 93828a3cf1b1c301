// Host runtime of libdanra_hip.so: parameter table, weight packing, workspace, the UNet forward
// schedule and the DDPM sampling loop.  C ABI declared in include/danra_hip.h.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/danra_hip.h"
#include "danra_kernels.h"

using namespace danra;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return fail(DANRA_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)
#define RC_TRY(expr)           \
    do {                       \
        int rc__ = (expr);     \
        if (rc__ != DANRA_OK) return rc__; \
    } while (0)

const int kFmap[5] = {64, 64, 128, 256, 512};  // modules_DANRA_conditional.py:170

struct ParamEntry {
    std::string name;
    size_t numel;
    const float* src = nullptr;
};

// A packed weight matrix that k_igemm may multiply in its split-operand fp16 form: a second copy in that form
// (launch_split_weights) and the power-of-two scale it carries.  Null when the matrix is not eligible.
struct SplitW {
    float* wh = nullptr;
    float inv_scale = 1.0f;
};

struct ConvP {
    SplitW split;
    float* wfrag = nullptr; // the split weights in the fragment order of the window kernels, or null
    int wrows = 0;          // rows of w / wfrag when they are not Cout (conv1 of a down-sampling block carries the shortcut's rows)
    float* w = nullptr;     // [Cout][KH][KW][Cin] (stem: [Cout][Cin][KH][KW])
    float* bias = nullptr;  // folded BatchNorm shift or conv bias (may be null)
    int Cin = 0, Cout = 0, KH = 0, KW = 0, stride = 1, pad = 0;
};
struct AttnP {
    float* w_in = nullptr;   // [3C][C], LayerNorm scale folded in
    float* b_in = nullptr;   // [3C],    LayerNorm shift folded in
    float* w_out = nullptr;  // [C][C]
    float* b_out = nullptr;
    // feed-forward block of the clean-application network (src/unet.py:91-96): LayerNorm folded into the first Linear
    float *w_ff1 = nullptr, *b_ff1 = nullptr, *w_ff2 = nullptr, *b_ff2 = nullptr;
    SplitW s_in, s_out, s_ff1, s_ff2;
    float* w_in_frag = nullptr;  // s_in in fragment order for the fused short-sequence kernel (launch_split_to_frag16), or null
    float* w_out_frag = nullptr; // s_out in fragment order for k_proj_small, or null
    float *w_ff1_frag = nullptr, *w_ff2_frag = nullptr;  // the same for the feed-forward matrices (C >= 128)
    int C = 0;
};
struct BlockP {
    ConvP c1, c2, sc;
    bool has_sc = false;
    // Blocks with a 1x1 stride-2 shortcut: the shortcut reads exactly the centre tap of the 3x3 stride-2 conv1 window,
    // so both are one implicit GEMM with 2*planes output columns (rows planes.. hold the shortcut weights in the
    // centre tap's K columns and zeros elsewhere).  c1.w / c1.bias point at the first half of these.
    float *w_both = nullptr, *b_both = nullptr;
};
struct DecP {
    float* wt = nullptr;  // [4*Cin][Cin]
    float* bt = nullptr;  // [4*Cin]
    SplitW s_wt;
    float* wt_frag = nullptr;  // the transposed convolution's matrix in fragment order for k_proj_small (Cin = 64), or null
    ConvP conv;
    AttnP attn;
    int Cin = 0, Cout = 0;
    int tb_off = 0;
};

struct Workspace {
    int B = 0, H = 0, W = 0;
    void* base = nullptr;
    size_t bytes = 0;
    float *tbias, *f1pre, *c2, *rt[4], *fpre[5], *fmap[5], *qkv, *kv_split, *kv_flags, *obuf, *up, *cv, *pre, *outA, *outB;
    float *mean1, *rstd1, *mean2, *rstd2, *partial, *eps, *eps_uncond, *stats_part, *zeros, *stem_static;
    size_t stats_part_floats = 0;
    size_t partial_floats = 0;
    // sampler: device-resident loop state, per-timestep update coefficients, private copies of the inputs
    // (so that a captured step never refers to caller memory) and the captured step itself
    SampleState* state;
    float4* coef;
    float *xs, *conds, *lsms, *topos;
    int64_t* ys;
    hipGraph_t step_graph = nullptr;
    hipGraphExec_t step_exec = nullptr;
    int step_exec_mode = -1;  // what the captured step was built for: bit 0 = labels, bit 1 = guided (two evaluations),
                              // bit 2 = every product with fp32 instructions (danra_set_fp32_products)
};
constexpr int kMaxTimesteps = 8192;  // rows of the coefficient table

}  // namespace

struct danra_handle_s {
    danra_config_t cfg;
    int in_channels = 1;
    int tb_width = 0, enc_width = 0;
    int tb_off_enc[5];
    std::vector<ParamEntry> params;
    std::unordered_map<std::string, int> index;
    // packed model
    void* arena = nullptr;
    size_t arena_bytes = 0;
    void* arena_h = nullptr;  // the split-operand copies of the eligible matrices, then one float per matrix (max |w|)
    size_t arena_h_bytes = 0;
    bool packed = false;
    ConvP stem, conv2, final_conv;
    std::vector<BlockP> blocks[4];
    AttnP attn[9];
    DecP dec[4];
    float *final_wt = nullptr, *final_bt = nullptr;
    // DANRA_FUSED_SMALL_ATTENTION=0 keeps the three-kernel form of the short-sequence attention blocks (comparison runs)
    bool fused_small_attention = !(getenv("DANRA_FUSED_SMALL_ATTENTION") && atoi(getenv("DANRA_FUSED_SMALL_ATTENTION")) == 0);
    // DANRA_CONV_WIN=0: every convolution through the implicit-GEMM kernel (comparison runs)
    // DANRA_PRESPLIT=0: the QKV projections leave fp32 K and V and the attention core splits them tile by tile (comparison runs)
    bool presplit = !(getenv("DANRA_PRESPLIT") && atoi(getenv("DANRA_PRESPLIT")) == 0);
    bool conv_win = !(getenv("DANRA_CONV_WIN") && atoi(getenv("DANRA_CONV_WIN")) == 0);
    bool conv_s2 = !(getenv("DANRA_CONV_S2") && atoi(getenv("DANRA_CONV_S2")) == 0);   // ... only the 8x8 stride-2 one
    SplitW final_s_wt;
    float *tw = nullptr, *tb = nullptr;  // time projections [tb_width][dim], [tb_width]
    float* label_emb = nullptr;
    float *enc_inv_freq = nullptr, *dec_div = nullptr;
    std::vector<float> h_enc_inv_freq, h_dec_div;
    bool freq_dirty = true;
    Workspace ws;
    // optional HIP-event timing of one kernel family (danra_profile_begin/_end)
    int prof_kind = 0;  // 0 = off
    bool prof_now = true;  // danra_sample times one denoise step in kProfStride and replays the others from the graph
    std::vector<hipEvent_t> prof_events;
    size_t prof_used = 0;
    double prof_flops = 0.0;
    double prof_exec_flops = 0.0;   // multiply-adds the instructions executed (x2), see igemm()
    double prof_split_flops = 0.0;  // the part of prof_flops that ran in the split-operand fp16 form
    double prof_bytes = 0.0;        // operands, weights and results of the timed launches, each counted once
    // sampling loop: 1 = replay one captured denoise step (hipGraph), 0 = enqueue every kernel
    int graph_mode = 1;
    hipStream_t cap_stream = nullptr;
    std::vector<float4> h_coef;
    // time projections of every (timestep, class) pair, rebuilt by each danra_sample call
    float* tb_table = nullptr;
    int64_t* tb_index = nullptr;  // [2][rows]: the timestep and the class of every table row
    size_t tb_rows_cap = 0;
    float* tb_table_u = nullptr;  // label-free rows [T][width] for the unconditional evaluations of a guided loop
    size_t tb_u_cap = 0;
    std::vector<int64_t> h_tb_index;
    // global index of the first image of the next danra_sample call (Philox counters are per global element)
    uint64_t first_image = 0;
};

namespace {

void add_param(danra_handle_s* h, const std::string& name, size_t numel) {
    h->index[name] = (int)h->params.size();
    h->params.push_back({name, numel, nullptr});
}
void add_bn(danra_handle_s* h, const std::string& p, int C) {
    add_param(h, p + ".weight", C);
    add_param(h, p + ".bias", C);
    add_param(h, p + ".running_mean", C);
    add_param(h, p + ".running_var", C);
}
// The MultiheadAttention module is `.attention` in modules_DANRA_conditional.py:87 and `.mha` in
// DDPM_clean_application/src/unet.py:87, which also adds the `.ff` Sequential (:91-96).
const char* mha_name(const danra_handle_s* h) { return h->cfg.attn_ff ? ".mha" : ".attention"; }
void add_attn(danra_handle_s* h, const std::string& p, int C) {
    const std::string m = p + mha_name(h);
    if (h->cfg.attn_ff) {
        add_param(h, m + ".in_proj_weight", (size_t)3 * C * C);
        add_param(h, m + ".in_proj_bias", (size_t)3 * C);
        add_param(h, m + ".out_proj.weight", (size_t)C * C);
        add_param(h, m + ".out_proj.bias", C);
        add_param(h, p + ".layernorm.weight", C);
        add_param(h, p + ".layernorm.bias", C);
        add_param(h, p + ".ff.0.weight", C);
        add_param(h, p + ".ff.0.bias", C);
        add_param(h, p + ".ff.1.weight", (size_t)C * C);
        add_param(h, p + ".ff.1.bias", C);
        add_param(h, p + ".ff.3.weight", (size_t)C * C);
        add_param(h, p + ".ff.3.bias", C);
        return;
    }
    add_param(h, p + ".layernorm.weight", C);
    add_param(h, p + ".layernorm.bias", C);
    add_param(h, m + ".in_proj_weight", (size_t)3 * C * C);
    add_param(h, m + ".in_proj_bias", (size_t)3 * C);
    add_param(h, m + ".out_proj.weight", (size_t)C * C);
    add_param(h, m + ".out_proj.bias", C);
}

void build_param_table(danra_handle_s* h) {
    const int dim = h->cfg.time_embedding;
    add_param(h, "encoder.conv1.weight", (size_t)64 * h->in_channels * 64);
    add_bn(h, "encoder.bn1", 64);
    int inpl = 64;
    for (int L = 1; L <= 4; ++L) {
        const int planes = 64 << (L - 1);
        for (int b = 0; b < h->cfg.block_layers[L - 1]; ++b) {
            const std::string p = "encoder.layer" + std::to_string(L) + "." + std::to_string(b);
            add_param(h, p + ".conv1.weight", (size_t)planes * inpl * 9);
            add_bn(h, p + ".bn1", planes);
            add_param(h, p + ".conv2.weight", (size_t)planes * planes * 9);
            add_bn(h, p + ".bn2", planes);
            if (b == 0 && (L > 1 || inpl != planes)) {
                add_param(h, p + ".downsample.0.weight", (size_t)planes * inpl);
                add_bn(h, p + ".downsample.1", planes);
            }
            inpl = planes;
        }
    }
    for (int i = 0; i < 5; ++i) {
        const std::string p = "encoder.time_projection_layers." + std::to_string(i) + ".1";
        add_param(h, p + ".weight", (size_t)kFmap[i] * dim);
        add_param(h, p + ".bias", kFmap[i]);
    }
    for (int i = 0; i < 5; ++i) add_attn(h, "encoder.attention_layers." + std::to_string(i), kFmap[i]);
    add_param(h, "encoder.conv2.weight", (size_t)64 * 64 * 64);
    if (h->cfg.num_classes > 0) add_param(h, "encoder.label_emb.weight", (size_t)h->cfg.num_classes * dim);
    for (int k = 0; k < 4; ++k) {
        const int cin = kFmap[4 - k], cout = kFmap[3 - k];
        const std::string p = "decoder.residual_layers." + std::to_string(k);
        add_attn(h, p + ".attention", cout);
        add_param(h, p + ".time_projection_layer.1.weight", (size_t)cout * dim);
        add_param(h, p + ".time_projection_layer.1.bias", cout);
        add_param(h, p + ".transpose.weight", (size_t)cin * cin * 4);
        add_param(h, p + ".transpose.bias", cin);
        add_param(h, p + ".conv.weight", (size_t)cout * cin * 9);
        add_param(h, p + ".conv.bias", cout);
    }
    add_param(h, "decoder.final_layer.transpose.weight", (size_t)64 * 64 * 4);
    add_param(h, "decoder.final_layer.transpose.bias", 64);
    add_param(h, "decoder.final_layer.conv.weight", (size_t)h->cfg.out_channels * 64 * 9);
    add_param(h, "decoder.final_layer.conv.bias", h->cfg.out_channels);
}

struct Bump {
    char* base;
    size_t off = 0;
    explicit Bump(void* b) : base((char*)b) {}
    float* take(size_t floats) {
        float* p = base ? (float*)(base + off) : nullptr;
        off += ((floats * sizeof(float) + 255) / 256) * 256;
        return p;
    }
};

const float* src_of(danra_handle_s* h, const std::string& name) {
    auto it = h->index.find(name);
    return it == h->index.end() ? nullptr : h->params[it->second].src;
}

// A matrix registered for splitting while the arena is laid out; split_registered() fills the copies afterwards.
struct SplitJob {
    const float* w;
    size_t n;
    SplitW* dst;
    bool frag64;  // rows of 64 weights in the fragment order of the row-stationary K = 64 GEMM
};

// Lays out (and on the second pass fills) the packed arena; `bh` lays out the arena of split-operand copies.
int layout_and_pack(danra_handle_s* h, Bump& bp, Bump& bh, std::vector<SplitJob>& jobs, bool fill, hipStream_t st) {
    const int dim = h->cfg.time_embedding;
    // k_igemm's split-operand form needs rows of whole 64-deep chunks and pays from K = 128 on (the K = 64 products
    // go through the row-stationary kernels, which are bound by memory, not by the matrix cores)
    // (K = 64 only where asked for: the token projections that go through k_gemm_k64, which has a layout of its own)
    auto want_split = [&](SplitW& s, float* w, size_t rows, size_t K, bool k64 = false) {
        s = SplitW();
        if (K % 64 || (K < 128 && !(k64 && K == 64))) return;
        s.wh = bh.take(rows * K);
        jobs.push_back({w, rows * K, &s, K == 64});
    };
    float* scratch_scale = bp.take(512);
    auto pack_conv_bn = [&](ConvP& c, const std::string& wname, const std::string& bn, int Cin, int Cout, int k,
                            int stride, int pad) -> int {
        c.Cin = Cin; c.Cout = Cout; c.KH = c.KW = k; c.stride = stride; c.pad = pad;
        c.w = bp.take((size_t)Cout * Cin * k * k);
        c.bias = bp.take(Cout);
        if (Cin % 64 == 0) want_split(c.split, c.w, Cout, (size_t)Cin * k * k);
        c.wfrag = ((k == 3 && stride == 1 && Cin % 64 == 0 && Cout % 32 == 0) || (k == 8 && stride == 2 && pad == 3 && Cin % 64 == 0 && Cout % 64 == 0))
                      ? bh.take((size_t)Cout * Cin * k * k) : nullptr;
        if (fill) {
            // scratch_scale is reused by consecutive launches on one stream: ordered, no hazard
            HIP_TRY(launch_bn_fold(src_of(h, bn + ".weight"), src_of(h, bn + ".bias"), src_of(h, bn + ".running_mean"),
                                   src_of(h, bn + ".running_var"), scratch_scale, c.bias, Cout, st));
            HIP_TRY(launch_pack_conv(src_of(h, wname), scratch_scale, c.w, Cout, Cin, k, k, st));
        }
        return DANRA_OK;
    };
    auto pack_attn = [&](AttnP& a, const std::string& p, int C) -> int {
        a.C = C;
        a.w_in = bp.take((size_t)3 * C * C);
        a.b_in = bp.take((size_t)3 * C);
        a.w_out = bp.take((size_t)C * C);
        a.b_out = bp.take(C);
        want_split(a.s_in, a.w_in, (size_t)3 * C, C, true);
        want_split(a.s_out, a.w_out, C, C, true);
        // (filled behind split_registered; the shapes attn_block_small_supported knows)
        a.w_in_frag = (C == 128 || C == 256 || C == 512) ? bh.take((size_t)3 * C * C) : nullptr;
        a.w_out_frag = (C == 128 || C == 256 || C == 512) ? bh.take((size_t)C * C) : nullptr;
        if (C == 64) {  // (the K = 64 split copies are in k_gemm_k64's own order: these come from the fp32 matrices)
            a.w_in_frag = bh.take((size_t)3 * C * C);
            a.w_out_frag = bh.take((size_t)C * C);
        }
        if (h->cfg.attn_ff) {
            a.w_ff1 = bp.take((size_t)C * C);
            a.b_ff1 = bp.take(C);
            a.w_ff2 = bp.take((size_t)C * C);
            a.b_ff2 = bp.take(C);
            want_split(a.s_ff1, a.w_ff1, C, C, true);
            want_split(a.s_ff2, a.w_ff2, C, C, true);
            if (C == 128 || C == 256 || C == 512) {
                a.w_ff1_frag = bh.take((size_t)C * C);
                a.w_ff2_frag = bh.take((size_t)C * C);
            }
        }
        if (fill) {
            const std::string m = p + mha_name(h);
            HIP_TRY(launch_pack_inproj(src_of(h, m + ".in_proj_weight"), src_of(h, m + ".in_proj_bias"),
                                       src_of(h, p + ".layernorm.weight"), src_of(h, p + ".layernorm.bias"), a.w_in,
                                       a.b_in, 3 * C, C, st));
            HIP_TRY(hipMemcpyAsync(a.w_out, src_of(h, m + ".out_proj.weight"), (size_t)C * C * 4,
                                   hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipMemcpyAsync(a.b_out, src_of(h, m + ".out_proj.bias"), (size_t)C * 4,
                                   hipMemcpyDeviceToDevice, st));
            if (h->cfg.attn_ff) {
                HIP_TRY(launch_pack_inproj(src_of(h, p + ".ff.1.weight"), src_of(h, p + ".ff.1.bias"),
                                           src_of(h, p + ".ff.0.weight"), src_of(h, p + ".ff.0.bias"), a.w_ff1, a.b_ff1, C,
                                           C, st));
                HIP_TRY(hipMemcpyAsync(a.w_ff2, src_of(h, p + ".ff.3.weight"), (size_t)C * C * 4,
                                       hipMemcpyDeviceToDevice, st));
                HIP_TRY(hipMemcpyAsync(a.b_ff2, src_of(h, p + ".ff.3.bias"), (size_t)C * 4, hipMemcpyDeviceToDevice, st));
            }
        }
        return DANRA_OK;
    };

    // stem: native OIHW flattening is already [Cout][plane][kh][kw]
    h->stem.Cin = h->in_channels; h->stem.Cout = 64; h->stem.KH = h->stem.KW = 8; h->stem.stride = 2; h->stem.pad = 3;
    h->stem.w = bp.take((size_t)64 * h->in_channels * 64);
    h->stem.bias = nullptr;
    if (fill)
        HIP_TRY(hipMemcpyAsync(h->stem.w, src_of(h, "encoder.conv1.weight"), (size_t)64 * h->in_channels * 64 * 4,
                               hipMemcpyDeviceToDevice, st));
    RC_TRY(pack_conv_bn(h->conv2, "encoder.conv2.weight", "encoder.bn1", 64, 64, 8, 2, 3));
    int inpl = 64;
    for (int L = 1; L <= 4; ++L) {
        const int planes = 64 << (L - 1);
        h->blocks[L - 1].resize(h->cfg.block_layers[L - 1]);
        for (int b = 0; b < h->cfg.block_layers[L - 1]; ++b) {
            BlockP& blk = h->blocks[L - 1][b];
            const std::string p = "encoder.layer" + std::to_string(L) + "." + std::to_string(b);
            const int stride = (b == 0 && L > 1) ? 2 : 1;
            blk.has_sc = (b == 0 && (L > 1 || inpl != planes));
            const bool both = blk.has_sc && stride == 2;
            if (both) {  // conv1's packed rows are followed by the shortcut's rows; the two bias vectors likewise
                blk.w_both = bp.take((size_t)2 * planes * inpl * 9);
                blk.b_both = bp.take((size_t)2 * planes);
                blk.c1.Cin = inpl; blk.c1.Cout = planes; blk.c1.KH = blk.c1.KW = 3; blk.c1.stride = stride; blk.c1.pad = 1;
                blk.c1.w = blk.w_both;
                blk.c1.bias = blk.b_both;
                if (inpl % 64 == 0) want_split(blk.c1.split, blk.w_both, (size_t)2 * planes, (size_t)9 * inpl);
                blk.c1.wrows = 2 * planes;
                blk.c1.wfrag = (inpl % 64 == 0) ? bh.take((size_t)2 * planes * inpl * 9) : nullptr;
                if (fill) {
                    HIP_TRY(launch_bn_fold(src_of(h, p + ".bn1.weight"), src_of(h, p + ".bn1.bias"),
                                           src_of(h, p + ".bn1.running_mean"), src_of(h, p + ".bn1.running_var"),
                                           scratch_scale, blk.c1.bias, planes, st));
                    HIP_TRY(launch_pack_conv(src_of(h, p + ".conv1.weight"), scratch_scale, blk.c1.w, planes, inpl, 3, 3, st));
                }
            } else {
                RC_TRY(pack_conv_bn(blk.c1, p + ".conv1.weight", p + ".bn1", inpl, planes, 3, stride, 1));
            }
            RC_TRY(pack_conv_bn(blk.c2, p + ".conv2.weight", p + ".bn2", planes, planes, 3, 1, 1));
            if (blk.has_sc)
                RC_TRY(pack_conv_bn(blk.sc, p + ".downsample.0.weight", p + ".downsample.1", inpl, planes, 1, stride, 0));
            if (both && fill) {
                const size_t K = (size_t)9 * inpl;
                float* rows2 = blk.w_both + (size_t)planes * K;
                HIP_TRY(hipMemsetAsync(rows2, 0, (size_t)planes * K * sizeof(float), st));
                HIP_TRY(hipMemcpy2DAsync(rows2 + (size_t)4 * inpl, K * sizeof(float), blk.sc.w, (size_t)inpl * sizeof(float),
                                         (size_t)inpl * sizeof(float), planes, hipMemcpyDeviceToDevice, st));
                HIP_TRY(hipMemcpyAsync(blk.b_both + planes, blk.sc.bias, (size_t)planes * sizeof(float),
                                       hipMemcpyDeviceToDevice, st));
            }
            inpl = planes;
        }
    }
    // time projections: rows = encoder levels 0..4 then decoder blocks 0..3
    h->tw = bp.take((size_t)h->tb_width * dim);
    h->tb = bp.take(h->tb_width);
    if (fill) {
        for (int i = 0; i < 5; ++i) {
            const std::string p = "encoder.time_projection_layers." + std::to_string(i) + ".1";
            HIP_TRY(hipMemcpyAsync(h->tw + (size_t)h->tb_off_enc[i] * dim, src_of(h, p + ".weight"),
                                   (size_t)kFmap[i] * dim * 4, hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipMemcpyAsync(h->tb + h->tb_off_enc[i], src_of(h, p + ".bias"), (size_t)kFmap[i] * 4,
                                   hipMemcpyDeviceToDevice, st));
        }
    }
    for (int i = 0; i < 5; ++i) RC_TRY(pack_attn(h->attn[i], "encoder.attention_layers." + std::to_string(i), kFmap[i]));
    h->label_emb = nullptr;
    if (h->cfg.num_classes > 0) {
        h->label_emb = bp.take((size_t)h->cfg.num_classes * dim);
        if (fill)
            HIP_TRY(hipMemcpyAsync(h->label_emb, src_of(h, "encoder.label_emb.weight"),
                                   (size_t)h->cfg.num_classes * dim * 4, hipMemcpyDeviceToDevice, st));
    }
    for (int k = 0; k < 4; ++k) {
        DecP& d = h->dec[k];
        d.Cin = kFmap[4 - k];
        d.Cout = kFmap[3 - k];
        const std::string p = "decoder.residual_layers." + std::to_string(k);
        d.wt = bp.take((size_t)4 * d.Cin * d.Cin);
        d.bt = bp.take((size_t)4 * d.Cin);
        d.conv.Cin = d.Cin; d.conv.Cout = d.Cout; d.conv.KH = d.conv.KW = 3; d.conv.stride = 1; d.conv.pad = 1;
        d.conv.w = bp.take((size_t)d.Cout * d.Cin * 9);
        d.conv.bias = bp.take(d.Cout);
        want_split(d.s_wt, d.wt, (size_t)4 * d.Cin, d.Cin, true);
        d.wt_frag = d.Cin == 64 ? bh.take((size_t)4 * d.Cin * d.Cin) : nullptr;
        want_split(d.conv.split, d.conv.w, d.Cout, (size_t)9 * d.Cin);
        d.conv.wfrag = (d.Cin % 64 == 0 && d.Cout % 32 == 0) ? bh.take((size_t)d.Cout * d.Cin * 9) : nullptr;
        if (fill) {
            HIP_TRY(launch_pack_convT(src_of(h, p + ".transpose.weight"), src_of(h, p + ".transpose.bias"), d.wt, d.bt,
                                      d.Cin, d.Cin, st));
            HIP_TRY(launch_pack_conv(src_of(h, p + ".conv.weight"), nullptr, d.conv.w, d.Cout, d.Cin, 3, 3, st));
            HIP_TRY(hipMemcpyAsync(d.conv.bias, src_of(h, p + ".conv.bias"), (size_t)d.Cout * 4, hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipMemcpyAsync(h->tw + (size_t)d.tb_off * dim, src_of(h, p + ".time_projection_layer.1.weight"),
                                   (size_t)d.Cout * dim * 4, hipMemcpyDeviceToDevice, st));
            HIP_TRY(hipMemcpyAsync(h->tb + d.tb_off, src_of(h, p + ".time_projection_layer.1.bias"), (size_t)d.Cout * 4,
                                   hipMemcpyDeviceToDevice, st));
        }
        RC_TRY(pack_attn(d.attn, p + ".attention", d.Cout));
        h->attn[5 + k] = d.attn;
    }
    h->final_wt = bp.take((size_t)4 * 64 * 64);
    want_split(h->final_s_wt, h->final_wt, (size_t)4 * 64, 64, true);
    h->final_bt = bp.take(4 * 64);
    h->final_conv.Cin = 64; h->final_conv.Cout = h->cfg.out_channels; h->final_conv.KH = h->final_conv.KW = 3;
    h->final_conv.stride = 1; h->final_conv.pad = 1;
    h->final_conv.w = bp.take((size_t)h->cfg.out_channels * 64 * 9);
    h->final_conv.bias = bp.take(h->cfg.out_channels);
    if (fill) {
        HIP_TRY(launch_pack_convT(src_of(h, "decoder.final_layer.transpose.weight"),
                                  src_of(h, "decoder.final_layer.transpose.bias"), h->final_wt, h->final_bt, 64, 64, st));
        HIP_TRY(launch_pack_conv(src_of(h, "decoder.final_layer.conv.weight"), nullptr, h->final_conv.w,
                                 h->cfg.out_channels, 64, 3, 3, st));
        HIP_TRY(hipMemcpyAsync(h->final_conv.bias, src_of(h, "decoder.final_layer.conv.bias"),
                               (size_t)h->cfg.out_channels * 4, hipMemcpyDeviceToDevice, st));
    }
    h->enc_inv_freq = bp.take(dim / 2);
    h->dec_div = bp.take(dim / 2);
    return DANRA_OK;
}

// The split-operand copies of the registered matrices.  Each is scaled by the power of two that brings its largest
// |w| into [2^12, 2^13): the low halves of all but vanishing weights then stay normal fp16 numbers.  A matrix that is
// all zero or holds a non-finite value keeps the fp32 instructions.  Waits for the stream (the maxima are read back).
int split_registered(danra_handle_s* h, std::vector<SplitJob>& jobs, float* maxima_dev, hipStream_t st) {
    if (jobs.empty()) return DANRA_OK;
    for (size_t i = 0; i < jobs.size(); ++i) HIP_TRY(launch_maxabs(jobs[i].w, jobs[i].n, maxima_dev + i, st));
    std::vector<float> maxima(jobs.size());
    HIP_TRY(hipMemcpyAsync(maxima.data(), maxima_dev, jobs.size() * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (size_t i = 0; i < jobs.size(); ++i) {
        const float mx = maxima[i];
        if (!(mx > 0.f) || !(mx <= 3.0e38f)) {
            jobs[i].dst->wh = nullptr;
            continue;
        }
        int e = 0;
        (void)frexpf(mx, &e);            // mx = m * 2^e, m in [0.5, 1)
        int s = 13 - e;                  // 2^s * mx in [2^12, 2^13)
        s = s > 100 ? 100 : (s < -100 ? -100 : s);
        jobs[i].dst->inv_scale = ldexpf(1.0f, -s);
        HIP_TRY(launch_split_weights(jobs[i].w, jobs[i].dst->wh, jobs[i].n, ldexpf(1.0f, s), st, jobs[i].frag64));
    }
    return DANRA_OK;
}

int upload_freq(danra_handle_s* h, hipStream_t st) {
    if (!h->freq_dirty) return DANRA_OK;
    const size_t n = h->cfg.time_embedding / 2;
    HIP_TRY(hipMemcpyAsync(h->enc_inv_freq, h->h_enc_inv_freq.data(), n * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(h->dec_div, h->h_dec_div.data(), n * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));  // host vectors may change afterwards
    h->freq_dirty = false;
    return DANRA_OK;
}

void drop_step_graph(Workspace& ws) {
    if (ws.step_exec) (void)hipGraphExecDestroy(ws.step_exec);
    if (ws.step_graph) (void)hipGraphDestroy(ws.step_graph);
    ws.step_exec = nullptr;
    ws.step_graph = nullptr;
    ws.step_exec_mode = -1;
}

int ensure_workspace(danra_handle_s* h, int B, int H, int W) {
    Workspace& ws = h->ws;
    if (ws.base && ws.B == B && ws.H == H && ws.W == W) return DANRA_OK;
    if (ws.base) {
        HIP_TRY(hipDeviceSynchronize());
        drop_step_graph(ws);
        HIP_TRY(hipFree(ws.base));
        ws.base = nullptr;
    }
    for (int pass = 0; pass < 2; ++pass) {
        Bump bp(pass ? ws.base : nullptr);
        const size_t hw2 = (size_t)(H / 2) * (W / 2), hw4 = (size_t)(H / 4) * (W / 4);
        ws.tbias = bp.take((size_t)B * h->tb_width);
        ws.state = (SampleState*)bp.take(64);
        ws.coef = (float4*)bp.take((size_t)kMaxTimesteps * 4);
        ws.ys = (int64_t*)bp.take((size_t)B * 2);
        ws.xs = bp.take((size_t)B * H * W);
        ws.conds = bp.take((size_t)B * H * W);
        ws.lsms = bp.take((size_t)B * H * W);
        ws.topos = bp.take((size_t)B * H * W);
        ws.f1pre = bp.take(B * hw2 * 64);
        ws.c2 = bp.take(B * hw4 * 64);
        for (int i = 0; i < 4; ++i) ws.rt[i] = bp.take(B * hw4 * 64);
        size_t hw = hw2;
        for (int i = 0; i < 5; ++i) {
            ws.fpre[i] = (i == 0) ? ws.f1pre : bp.take(B * hw * kFmap[i]);
            ws.fmap[i] = bp.take(B * hw * kFmap[i]);
            hw /= 4;
        }
        ws.qkv = bp.take(3 * B * hw2 * 64);
        ws.kv_split = bp.take(2 * B * hw2 * 64);   // K and V of the 64-channel blocks as their attention core reads them
        ws.kv_flags = bp.take(B * hw2 / 2 + 64);   // ... and their range flags (ints): two per group of 16 keys and head
        ws.obuf = bp.take(B * hw2 * 64);
        ws.up = bp.take((size_t)B * H * W * 64);
        ws.cv = bp.take(B * hw2 * 64);
        ws.pre = bp.take(B * hw2 * 64);
        ws.outA = bp.take(B * hw2 * 64);
        ws.outB = bp.take(B * hw2 * 64);
        ws.mean1 = bp.take((size_t)B * 512);
        ws.rstd1 = bp.take((size_t)B * 512);
        ws.mean2 = bp.take((size_t)B * 512);
        ws.rstd2 = bp.take((size_t)B * 512);
        ws.eps = bp.take((size_t)B * H * W * h->cfg.out_channels);
        ws.eps_uncond = bp.take((size_t)B * H * W * h->cfg.out_channels);
        ws.stats_part_floats = (size_t)1 << 19;  // sliced statistics; the fused producer leaves 4 floats per output pixel
        if (ws.stats_part_floats < (size_t)4 * B * H * W) ws.stats_part_floats = (size_t)4 * B * H * W;
        ws.stats_part = bp.take(ws.stats_part_floats);
        ws.partial_floats = (size_t)16 << 20;  // 64 MiB of split-K slabs
        ws.partial = bp.take(ws.partial_floats);
        ws.zeros = bp.take(64);
        ws.stem_static = bp.take(B * hw2 * 64);
        if (!pass) {
            ws.bytes = bp.off;
            if (hipMalloc(&ws.base, ws.bytes) != hipSuccess) {
                ws.base = nullptr;
                return fail(DANRA_ENOMEM, "workspace allocation of %zu bytes failed", ws.bytes);
            }
        }
    }
    HIP_TRY(hipMemset(ws.zeros, 0, 64 * sizeof(float)));
    ws.B = B; ws.H = H; ws.W = W;
    return DANRA_OK;
}

// ------------------------------------------------------------------------------------------------
// op launchers shared by the forward schedule and the per-op entry points

void choose_split(IGemmArgs& a, const Workspace& ws) {
    // Split K across workgroups only when there are too few output tiles to give every CU one
    // (the kernel itself already spreads a chunk's k-blocks over up to 16 waves per workgroup).
    a.ksplit = 1;
    a.chunks_per_split = a.K / 32;
    const int chunks = a.K / 32;
    const long tiles = (long)((a.M + 63) / 64) * (a.N / 64);
    if (tiles >= 256 || chunks < 16 || !ws.partial) return;
    int want = (int)((256 + tiles - 1) / tiles);
    const int maxs = chunks / 8;
    if (want > maxs) want = maxs;
    while (want > 1 && (size_t)want * a.M * a.N > ws.partial_floats) --want;
    if (want <= 1) return;
    int cps = (chunks + want - 1) / want;
    cps += cps & 1;  // even, so that 64-deep staging chunks line up
    a.chunks_per_split = cps;
    a.ksplit = (chunks + cps - 1) / cps;
    a.partial = ws.partial;
}

int igemm(danra_handle_s* h, const IGemmArgs& a, hipStream_t st);
int conv_win(danra_handle_s* h, const ConvWinArgs& w, double flops, double bytes, hipStream_t st, bool strided8 = false);

void use_split(IGemmArgs& a, const SplitW& s) {
    a.Wh = s.wh;
    a.wscale = s.inv_scale;
    a.half = s.wh != nullptr;
}

IGemmArgs blank_args() {
    IGemmArgs a;
    memset(&a, 0, sizeof(a));
    a.ksplit = 1;
    a.rows_per_img = 1;
    return a;
}

// out2 != null: the weight matrix holds 2*Cout rows (see BlockP::w_both); the second Cout columns go to out2, bias only.
// slabs != null: if the convolution is split over K, its reduction kernel is left out and *slabs describes the slabs
// (the caller's next kernel sums them); slabs->count stays 0 when the convolution was not split and `out` is complete.
int run_conv(danra_handle_s* h, const ConvP& c, const float* in, float* out, int B, int Hi, int Wi, const float* res,
             int relu, const float* tbias, const float* nmean, const float* nrstd, hipStream_t st, float* out2 = nullptr,
             InSlabs* slabs = nullptr) {
    IGemmArgs a = blank_args();
    const int Ho = (Hi + 2 * c.pad - c.KH) / c.stride + 1, Wo = (Wi + 2 * c.pad - c.KW) / c.stride + 1;
    a.A = in; a.W = c.w; a.bias = c.bias; a.res = res; a.tbias = tbias; a.out = out;
    use_split(a, c.split);
    a.nmean = nmean; a.nrstd = nrstd; a.zeros = h->ws.zeros;
    a.M = B * Ho * Wo; a.N = out2 ? 2 * c.Cout : c.Cout; a.K = c.KH * c.KW * c.Cin;
    a.out2 = out2; a.n_split = out2 ? c.Cout : 0;
    a.a_mode = A_CONV;
    a.Hi = Hi; a.Wi = Wi; a.Cin = c.Cin; a.Ho = Ho; a.Wo = Wo; a.KH = c.KH; a.KW = c.KW; a.stride = c.stride; a.pad = c.pad;
    a.ldr = c.Cout; a.act = relu ? ACT_RELU : ACT_NONE; a.tb_stride = h->tb_width; a.rows_per_img = Ho * Wo;
    a.st_mode = ST_ROW; a.ldc = c.Cout;
    // tiny maps: pixel-major rows, so that a tile's rows share their valid taps and the padding taps are skipped
    // (the strided rows of that order cost nothing measurable: limits of 4, 16 and 64 pixels give the same step time)
    a.pix_major = (c.KH > 1 && Ho * Wo <= 64) ? 1 : 0;
    a.batch = B;
    if (c.Cin % 32 || c.Cout % 64) return fail(DANRA_EINVAL, "conv channels %d->%d unsupported", c.Cin, c.Cout);
    // 3x3 stride-1 convolutions on maps with power-of-two sides: the kernel that stages the input window in LDS and
    // streams the weights in fragment order (never split over workgroups, so it leaves no slabs)
    if (c.wfrag && c.KH == 3 && c.KW == 3 && c.stride == 1 && c.pad == 1 && !out2 && h->conv_win && !igemm_fp32_products_forced() &&
        conv_win_supported(B, Hi, Wi, c.Cin, c.Cout, nmean != nullptr)) {
        ConvWinArgs w;
        memset(&w, 0, sizeof(w));
        w.x = in; w.wf = c.wfrag; w.w32 = c.w; w.wscale = c.split.inv_scale; w.bias = c.bias; w.res = res; w.tbias = tbias;
        w.nmean = nmean; w.nrstd = nrstd; w.out = out;
        w.B = B; w.H = Hi; w.W = Wi; w.Cin = c.Cin; w.Cout = c.Cout; w.act = a.act; w.tb_stride = h->tb_width;
        if (slabs) slabs->count = 0;
        return conv_win(h, w, 2.0 * (double)a.M * (double)a.N * (double)a.K,
                        4.0 * ((double)a.N * a.K + (double)a.M * a.N * (res ? 2.0 : 1.0) + (double)B * Hi * Wi * c.Cin), st);
    }
    // 3x3 stride-2 convolutions on small maps (conv1 of a down-sampling BasicBlock, with the block's shortcut as further rows)
    if (c.wfrag && c.KH == 3 && c.KW == 3 && c.stride == 2 && c.pad == 1 && !nmean && !res && !tbias && h->conv_win && h->conv_s2 &&
        !igemm_fp32_products_forced() && (out2 == nullptr) == (c.wrows == 0 || c.wrows == c.Cout) &&
        conv_win_supported(B, Hi, Wi, c.Cin, a.N, false, 2)) {
        ConvWinArgs w;
        memset(&w, 0, sizeof(w));
        w.x = in; w.wf = c.wfrag; w.w32 = c.w; w.wscale = c.split.inv_scale; w.bias = c.bias; w.out = out;
        w.out2 = out2; w.n_split = out2 ? c.Cout : 0; w.stride = 2;
        w.B = B; w.H = Hi; w.W = Wi; w.Cin = c.Cin; w.Cout = a.N; w.act = a.act; w.tb_stride = h->tb_width;
        if (slabs) slabs->count = 0;
        return conv_win(h, w, 2.0 * (double)a.M * (double)a.N * (double)a.K,
                        4.0 * ((double)a.N * a.K + (double)a.M * a.N + (double)B * Hi * Wi * c.Cin), st);
    }
    // the 8x8 stride-2 convolution: the strided window kernel
    if (c.wfrag && c.KH == 8 && c.KW == 8 && c.stride == 2 && c.pad == 3 && !out2 && !nmean && h->conv_win && h->conv_s2 && !igemm_fp32_products_forced() &&
        conv8s2_supported(B, Hi, Wi, c.Cin, c.Cout)) {
        ConvWinArgs w;
        memset(&w, 0, sizeof(w));
        w.x = in; w.wf = c.wfrag; w.w32 = c.w; w.wscale = c.split.inv_scale; w.bias = c.bias; w.res = res; w.tbias = tbias;
        w.out = out;
        w.B = B; w.H = Hi; w.W = Wi; w.Cin = c.Cin; w.Cout = c.Cout; w.act = a.act; w.tb_stride = h->tb_width;
        if (slabs) slabs->count = 0;
        return conv_win(h, w, 2.0 * (double)a.M * (double)a.N * (double)a.K,
                        4.0 * ((double)a.N * a.K + (double)a.M * a.N * (res ? 2.0 : 1.0) + (double)B * Hi * Wi * c.Cin), st, true);
    }
    choose_split(a, h->ws);
    if (slabs) {
        slabs->count = 0;
        if (a.ksplit > 1 && !res && !relu && !tbias && !out2) {
            a.keep_slabs = 1;
            slabs->count = a.ksplit;
            slabs->stride = (size_t)a.M * a.N;
            slabs->bias = c.bias;
            slabs->pix_major = a.pix_major;
            slabs->ct_W = 0;
        }
    }
    RC_TRY(igemm(h, a, st));
    return DANRA_OK;
}

// The 8x8 stride-2 stem on input planes [first, first + count) of the concatenated input: a column slice of
// the stem's weight matrix ([64][Cin*64], 64 columns per plane).  `res` (NHWC, 64 channels) is added to the result.
int run_stem_planes(danra_handle_s* h, const float* const planes[4], int first, int count, float* out, int B, int H,
                    int W, const float* res, const float* tbias, hipStream_t st) {
    IGemmArgs a = blank_args();
    const ConvP& c = h->stem;
    for (int i = 0; i < count; ++i) a.planes[i] = planes[first + i];
    a.W = c.w + 64 * first; a.ldw = 64 * c.Cin; a.tbias = tbias; a.out = out;
    a.res = res; a.ldr = 64;
    a.M = B * (H / 2) * (W / 2); a.N = 64; a.K = 64 * count;
    a.a_mode = A_CONV1;
    a.Hi = H; a.Wi = W; a.Cin = count; a.Ho = H / 2; a.Wo = W / 2; a.KH = 8; a.KW = 8; a.stride = 2; a.pad = 3;
    a.tb_stride = h->tb_width; a.rows_per_img = (H / 2) * (W / 2);
    a.st_mode = ST_ROW; a.ldc = 64;
    RC_TRY(igemm(h, a, st));
    return DANRA_OK;
}

// Encoder.conv1 on cat([x, lsm, topo, cond]) (modules:228-238,260).  The convolution is linear in its input
// planes, so it is evaluated as (conditioning planes) + (x plane): the first term does not change during a
// sampling loop and is then computed once per danra_sample call (`static_ready`), leaving a K = 64 convolution
// per denoise step.  Every caller uses the same two-term evaluation, so results do not depend on who calls.
int run_stem(danra_handle_s* h, const float* const planes[4], float* out, int B, int H, int W, const float* tbias,
             bool static_ready, hipStream_t st) {
    const int Cin = h->stem.Cin;
    const float* res = nullptr;
    if (Cin > 1) {
        if (!static_ready)
            RC_TRY(run_stem_planes(h, planes, 1, Cin - 1, h->ws.stem_static, B, H, W, nullptr, nullptr, st));
        res = h->ws.stem_static;
    }
    // the x plane: a dedicated kernel (H, W are multiples of 32 here)
    HIP_TRY(launch_stem_x(planes[0], h->stem.w, 64 * Cin, res, tbias, h->tb_width, out, B, H, W, st));
    return DANRA_OK;
}

// Hands the next two profiling events to a launch (LaunchEvents::take): the pair is attached to the dispatch and
// carries the kernel's begin and end times.
int prof_take(void* ctx, hipEvent_t* start, hipEvent_t* stop) {
    danra_handle_s* h = static_cast<danra_handle_s*>(ctx);
    while (h->prof_events.size() < h->prof_used + 2) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return 1;  // (the launch then goes untimed)
        h->prof_events.push_back(e);
    }
    *start = h->prof_events[h->prof_used];
    *stop = h->prof_events[h->prof_used + 1];
    h->prof_used += 2;
    return 0;
}

// Scope in which the launches of this thread are timed for handle `h`.
struct ProfScope {
    LaunchEvents src;
    bool on;
    ProfScope(danra_handle_s* h, bool enable) : src{h, prof_take}, on(enable) {
        if (on) t_launch_events = &src;
    }
    ~ProfScope() {
        if (on) t_launch_events = nullptr;
    }
};

// Every implicit GEMM of the forward schedule goes through here: with the family being timed (danra_profile_begin),
// each of its kernels - the split-K reduction kernel is one - carries its own pair of events.
int igemm(danra_handle_s* h, const IGemmArgs& a, hipStream_t st) {
    const bool timed = h->prof_kind == DANRA_PROF_IGEMM && h->prof_now;
    {
        ProfScope scope(h, timed);
        HIP_TRY(launch_igemm(a, st));
    }
    if (timed) {
        const double fl = 2.0 * (double)a.M * (double)a.N * (double)a.K;
        h->prof_flops += fl;
        // the split-operand form issues three fp16 products per algorithmic one
        const bool split = igemm_uses_split(a);
        h->prof_exec_flops += split ? 3.0 * fl : fl;
        if (split) h->prof_split_flops += fl;
        // one pass over the input tensor (a convolution reads every input element, however many windows hold it),
        // the weights, the result and what the epilogue adds to it
        double elems = (double)a.N * a.K + (double)a.M * a.N * (a.res ? 2.0 : 1.0);
        if (a.a_mode == A_GEMM) elems += (double)a.M * a.K;
        else if (a.a_mode == A_CONV) elems += (double)(a.M / (a.Ho * a.Wo)) * a.Hi * a.Wi * a.Cin;
        else elems += (double)(a.M / (a.Ho * a.Wo)) * a.Hi * a.Wi * (a.K / 64);  // stem: K = planes x 8 x 8
        h->prof_bytes += 4.0 * elems;
    }
    return DANRA_OK;
}

// The window-staged 3x3 convolutions belong to the same timed family (every product of theirs is a split-operand one).
int conv_win(danra_handle_s* h, const ConvWinArgs& w, double flops, double bytes, hipStream_t st, bool strided8) {
    const bool timed = h->prof_kind == DANRA_PROF_IGEMM && h->prof_now;
    {
        ProfScope scope(h, timed);
        HIP_TRY(strided8 ? launch_conv8s2(w, st) : launch_conv_win(w, st));
    }
    if (timed) {
        h->prof_flops += flops;
        h->prof_exec_flops += 3.0 * flops;
        h->prof_split_flops += flops;
        h->prof_bytes += bytes;
    }
    return DANRA_OK;
}

// With ct_w given (64-channel blocks only) the transposed convolution that consumes the block's output is applied
// in the same kernel as the output projection: ct_out receives [B][2*ct_H][2*ct_W][64] and `out` is not written.
int run_attention(danra_handle_s* h, int index, const float* x, float* out, int B, int L, int relu, hipStream_t st,
                  const float* ct_w = nullptr, const float* ct_b = nullptr, float* ct_out = nullptr, int ct_H = 0,
                  int ct_W = 0, const SplitW* ct_split = nullptr, bool ct_stats_only = false) {
    const AttnP& at = h->attn[index];
    const int C = at.C, heads = h->cfg.n_heads, hd = C / heads;
    Workspace& ws = h->ws;
    if (attn_block_small_supported(B, L, C, heads) && h->fused_small_attention && at.w_in_frag && !igemm_fp32_products_forced()) {
        // short sequences of the deep levels: LayerNorm, QKV projection and attention as one kernel per (image, head)
        // (a split-operand product; with fp32 products forced the three-kernel form below runs)
        HIP_TRY(launch_attn_block_small(x, at.w_in_frag, at.s_in.inv_scale, at.b_in, ws.obuf, B, L, C, heads, st));
    } else {
        const bool lean_qkv = at.w_in_frag && h->fused_small_attention && !igemm_fp32_products_forced() &&
                              qkv_small_supported(B * L, C, L, hd);
        IGemmArgs a = blank_args();
        a.A = x; a.lda = C; a.ln = 1; a.a_mode = A_GEMM;
        a.W = at.w_in; a.bias = at.b_in; a.out = ws.qkv;
        use_split(a, at.s_in);
        a.M = B * L; a.N = 3 * C; a.K = C;
        a.st_mode = ST_QKV; a.qkv_C = C; a.qkv_hd = hd; a.qkv_L = L; a.qkv_heads = heads;
        a.qkv_plane = (long long)B * L * C;
        choose_split(a, ws);
        const size_t plane = (size_t)B * L * C;
        // head_dim 16 at L % 128 == 0 (the flash core): K and V leave the projection kernel already split, in the order of
        // the core's LDS tiles, with their range flags; the fp32 K / V planes are written only for flagged groups of keys
        const bool pre = h->presplit && hd == 16 && L % 128 == 0 && !igemm_fp32_products_forced() &&
                         (lean_qkv ? qkv_small_presplits(B * L, C, L, hd) : gemm_k64_presplits(a));
        float* ks = pre ? ws.kv_split : nullptr;
        float* vs = pre ? ws.kv_split + plane : nullptr;
        int* kvf = pre ? reinterpret_cast<int*>(ws.kv_flags) : nullptr;
        if (lean_qkv)  // few tokens: one trip to memory per wave instead of a tiled GEMM
            HIP_TRY(launch_qkv_small(x, at.w_in_frag, at.w_in, at.s_in.inv_scale, at.b_in, ws.qkv, B * L, C, L, hd, st, ks, vs, kvf));
        else {
            a.pre_ks = ks; a.pre_vs = vs; a.pre_flags = kvf;
            RC_TRY(igemm(h, a, st));
        }
        const bool flash = (hd == 16 && L % 128 == 0) || ((hd == 32 || hd == 64) && L % 64 == 0 && L >= 256);
        const bool timed = (h->prof_kind == DANRA_PROF_FLASH_ATTENTION && h->prof_now && flash);
        {
            ProfScope scope(h, timed);
            if (pre)
                HIP_TRY(launch_attention_hd16_pre(ws.qkv, ks, vs, ws.qkv + plane, ws.qkv + 2 * plane, kvf, ws.obuf, B, heads, L, st));
            else
                HIP_TRY(launch_attention(ws.qkv, ws.qkv + plane, ws.qkv + 2 * plane, ws.obuf, B, heads, L, hd, st));
        }
        if (timed) {
            const double fl = 4.0 * (double)B * heads * (double)L * (double)L * hd;  // QK^T and PV, 2 flop per MAC
            h->prof_flops += fl;
            h->prof_exec_flops += (hd == 16 && L % 128 == 0) ? 3.5 * fl : fl;  // head_dim 16: 4 + 3 fp16 instructions per 2 algorithmic products
            h->prof_bytes += 4.0 * 4.0 * (double)plane;  // q, k, v read once, the head outputs written once
        }
    }
    // With the feed-forward block the attended tokens a = out_proj(o) + x go to the (now idle) q/k/v buffer,
    // h = GELU(Linear1(LayerNorm(a))) to the (now idle) head-output buffer, and out = act(a + Linear2(h)).
    const bool ff = at.w_ff1 != nullptr;
    if (ct_w && C != 64) return fail(DANRA_EINVAL, "fused transposed convolution needs a 64-channel block");
    if (C == 64 && (ff || ct_w)) {  // the row-local GEMMs behind the attention core of a 64-channel block run as one kernel
        Tail64Split sp;
        sp.wo = at.s_out.wh; sp.so = at.s_out.inv_scale;
        sp.w1 = at.s_ff1.wh; sp.s1 = at.s_ff1.inv_scale;
        sp.w2 = at.s_ff2.wh; sp.s2 = at.s_ff2.inv_scale;
        if (ct_split) { sp.wt = ct_split->wh; sp.st = ct_split->inv_scale; }
        HIP_TRY(launch_attn_tail64(ws.obuf, x, at.w_out, at.b_out, at.w_ff1, at.b_ff1, at.w_ff2, at.b_ff2, ct_w, ct_b,
                                   (ct_w && !ct_stats_only) ? ct_out : out, B * L, relu ? ACT_RELU : ACT_NONE, ct_H, ct_W,
                                   ct_w ? ws.stats_part : nullptr, st, &sp, ct_stats_only ? 1 : 0));
        return DANRA_OK;
    }
    float* attended = ff ? ws.qkv : out;
    if (at.w_out_frag && h->fused_small_attention && !igemm_fp32_products_forced() && proj_small_supported(B * L, C, C)) {
        // few rows: the whole product as one trip to memory per wave instead of a tiled GEMM (+ split-K reduction)
        HIP_TRY(launch_proj_small(ws.obuf, at.w_out_frag, at.w_out, at.s_out.inv_scale, at.b_out, x, attended, B * L, C, C,
                                  (relu && !ff) ? ACT_RELU : ACT_NONE, st));
        if (!ff) return DANRA_OK;
    } else {
    IGemmArgs o = blank_args();
    o.A = ws.obuf; o.lda = C; o.a_mode = A_GEMM;
    use_split(o, at.s_out);
    o.W = at.w_out; o.bias = at.b_out; o.res = x; o.ldr = C; o.act = (relu && !ff) ? ACT_RELU : ACT_NONE; o.out = attended;
    o.M = B * L; o.N = C; o.K = C; o.st_mode = ST_ROW; o.ldc = C;
    choose_split(o, ws);
    RC_TRY(igemm(h, o, st));
    if (!ff) return DANRA_OK;
    }
    if (at.w_ff1_frag && at.w_ff2_frag && h->fused_small_attention && !igemm_fp32_products_forced() &&
        proj_small_supported(B * L, C, C)) {  // few rows: both linears of the feed-forward part as one-trip kernels
        HIP_TRY(launch_proj_small(attended, at.w_ff1_frag, at.w_ff1, at.s_ff1.inv_scale, at.b_ff1, nullptr, ws.obuf, B * L, C, C,
                                  ACT_GELU, st, 0, 0, true));
        HIP_TRY(launch_proj_small(ws.obuf, at.w_ff2_frag, at.w_ff2, at.s_ff2.inv_scale, at.b_ff2, attended, out, B * L, C, C,
                                  relu ? ACT_RELU : ACT_NONE, st));
        return DANRA_OK;
    }
    IGemmArgs f1 = blank_args();
    f1.A = attended; f1.lda = C; f1.ln = 1; f1.a_mode = A_GEMM;
    use_split(f1, at.s_ff1);
    f1.W = at.w_ff1; f1.bias = at.b_ff1; f1.act = ACT_GELU; f1.out = ws.obuf;
    f1.M = B * L; f1.N = C; f1.K = C; f1.st_mode = ST_ROW; f1.ldc = C;
    choose_split(f1, ws);
    RC_TRY(igemm(h, f1, st));
    IGemmArgs f2 = blank_args();
    f2.A = ws.obuf; f2.lda = C; f2.a_mode = A_GEMM;
    use_split(f2, at.s_ff2);
    f2.W = at.w_ff2; f2.bias = at.b_ff2; f2.res = attended; f2.ldr = C; f2.act = relu ? ACT_RELU : ACT_NONE; f2.out = out;
    f2.M = B * L; f2.N = C; f2.K = C; f2.st_mode = ST_ROW; f2.ldc = C;
    choose_split(f2, ws);
    RC_TRY(igemm(h, f2, st));
    return DANRA_OK;
}

// slabs != null: if the GEMM is split over K, its reduction kernel is left out and *slabs describes the slabs (the
// caller's InstanceNorm statistics kernel sums them and writes `out`); slabs->count stays 0 otherwise.
int run_convT(danra_handle_s* h, const float* wt, const float* bt, int C, const float* in, float* out, int B, int Hh,
              int Ww, hipStream_t st, const SplitW* split = nullptr, InSlabs* slabs = nullptr) {
    IGemmArgs a = blank_args();
    if (split) use_split(a, *split);
    a.A = in; a.lda = C; a.a_mode = A_GEMM;
    a.W = wt; a.bias = bt; a.out = out;
    a.M = B * Hh * Ww; a.N = 4 * C; a.K = C;
    a.st_mode = ST_CONVT; a.ct_C = C; a.ct_H = Hh; a.ct_W = Ww;
    choose_split(a, h->ws);
    if (slabs) {
        slabs->count = 0;
        if (a.ksplit > 1) {
            a.keep_slabs = 1;
            slabs->count = a.ksplit;
            slabs->stride = (size_t)a.M * a.N;
            slabs->bias = bt;  // (the packed bias repeats the C values for each of the four sub-pixels)
            slabs->pix_major = 0;
            slabs->ct_W = Ww;
        }
    }
    RC_TRY(igemm(h, a, st));
    return DANRA_OK;
}

// One ResNet layer; the output (with the level's time bias added when tbias != null) goes to `out`.
int run_layer(danra_handle_s* h, int level, const float* x, float* out, int B, int Hi, int Wi, const float* tbias,
              hipStream_t st) {
    Workspace& ws = h->ws;
    const std::vector<BlockP>& blks = h->blocks[level - 1];
    const float* cur = x;
    int Hc = Hi, Wc = Wi;
    for (size_t b = 0; b < blks.size(); ++b) {
        const BlockP& blk = blks[b];
        const int Ho = (Hc + 2 - 3) / blk.c1.stride + 1, Wo = (Wc + 2 - 3) / blk.c1.stride + 1;
        const bool last = (b + 1 == blks.size());
        float* t1 = ws.rt[0];
        float* sc = ws.rt[1];
        float* dst = last ? out : ws.rt[2 + (b & 1)];
        const float* idn = cur;
        if (blk.w_both) {  // conv1 and the shortcut in one launch
            RC_TRY(run_conv(h, blk.c1, cur, t1, B, Hc, Wc, nullptr, 1, nullptr, nullptr, nullptr, st, sc));
            idn = sc;
        } else {
            RC_TRY(run_conv(h, blk.c1, cur, t1, B, Hc, Wc, nullptr, 1, nullptr, nullptr, nullptr, st));
            if (blk.has_sc) {
                RC_TRY(run_conv(h, blk.sc, cur, sc, B, Hc, Wc, nullptr, 0, nullptr, nullptr, nullptr, st));
                idn = sc;
            }
        }
        RC_TRY(run_conv(h, blk.c2, t1, dst, B, Ho, Wo, idn, 1, last ? tbias : nullptr, nullptr, nullptr, st));
        cur = dst;
        Hc = Ho; Wc = Wo;
    }
    return DANRA_OK;
}

int run_time_mlp(danra_handle_s* h, const int64_t* t, SampleState* state, const int64_t* y, int B, float* enc_out,
                 float* dec_out, float* tbias, hipStream_t st) {
    RC_TRY(upload_freq(h, st));
    TimeMlpArgs a;
    a.t = t; a.state = state; a.y = (h->cfg.num_classes > 0) ? y : nullptr; a.label_emb = h->label_emb;
    a.enc_inv_freq = h->enc_inv_freq; a.dec_div = h->dec_div; a.W = h->tw; a.b = h->tb;
    a.enc_emb_out = enc_out; a.dec_emb_out = dec_out; a.tbias = tbias;
    a.B = B; a.dim = h->cfg.time_embedding; a.width = h->tb_width; a.enc_width = h->enc_width;
    a.num_classes = h->cfg.num_classes;
    HIP_TRY(launch_time_mlp(a, st));
    return DANRA_OK;
}

// DecoderBlock k (0..3) given tbias already computed; in [B][Hh][Ww][Cin] -> out [B][2Hh][2Ww][Cout]
// up_ready: ws.up already holds this block's transposed convolution (the previous block applied it, see run_attention);
// next_wt/next_bt: apply the NEXT block's transposed convolution to this block's output (into ws.up, `out` unused).
int run_decoder_block(danra_handle_s* h, int k, const float* in, const float* skip, const float* tbias, float* out,
                      int B, int Hh, int Ww, hipStream_t st, bool up_ready = false, const float* next_wt = nullptr,
                      const float* next_bt = nullptr, const SplitW* next_split = nullptr, bool next_stats_only = false) {
    Workspace& ws = h->ws;
    const DecP& d = h->dec[k];
    const int H2 = 2 * Hh, W2 = 2 * Ww;
    // The first InstanceNorm of the small maps: the block that forms an image's statistics holds its pixels in
    // registers, so it also writes the normalised values back (in place) and the convolution reads a plain tensor.
    // Normalising in the convolution's operand loader instead costs it two more 16-byte loads (mean, 1/std) per
    // activation load, and the loader is what its loop waits for: 21 against 16 us per launch at 64 images on a 4x4 map.
    bool normalised = false;
    if (!up_ready) {
        InSlabs up_slabs{0, 0, nullptr, 0, 0};
        // (measured at 64 images: pays up to 8x8 maps; from 16x16 on the extra store costs the statistics kernel what
        // the convolution gains, 2 us each, and at 32x32 more: +6 against -3.4 us)
        normalised = H2 * W2 <= 64;
        // (... and when the transposed convolution is split over K, that block sums its slabs as well)
        if (d.wt_frag && h->fused_small_attention && !igemm_fp32_products_forced() &&
            proj_small_supported(B * Hh * Ww, 4 * d.Cin, d.Cin))  // 64 channels: bound by its store, one trip to memory per wave
            HIP_TRY(launch_proj_small(in, d.wt_frag, d.wt, d.s_wt.inv_scale, d.bt, nullptr, ws.up, B * Hh * Ww, 4 * d.Cin, d.Cin,
                                      0, st, Hh, Ww));
        else
            RC_TRY(run_convT(h, d.wt, d.bt, d.Cin, in, ws.up, B, Hh, Ww, st, &d.s_wt, normalised ? &up_slabs : nullptr));
        HIP_TRY(launch_in_stats(up_slabs.count ? ws.partial : ws.up, ws.mean1, ws.rstd1, ws.stats_part, ws.stats_part_floats,
                                B, H2 * W2, d.Cin, st, normalised ? ws.up : nullptr, nullptr, nullptr, 0,
                                up_slabs.count ? &up_slabs : nullptr));
    } else {  // the producer of ws.up left its statistics in slices of 32 samples
        HIP_TRY(launch_in_stats_from_slices(ws.stats_part, ws.mean1, ws.rstd1, B, Hh * Ww / 8, 32, st));
    }
    // second InstanceNorm + skip + time bias in one pass (the block that forms an image's statistics holds its pixels);
    // when the convolution is split over K and an image fits one block, that pass also sums the convolution's slabs
    InSlabs slabs{0, 0, nullptr, 0, 0};
    RC_TRY(run_conv(h, d.conv, ws.up, ws.cv, B, H2, W2, nullptr, 0, nullptr, normalised ? nullptr : ws.mean1,
                    normalised ? nullptr : ws.rstd1, st, nullptr, H2 * W2 <= 1024 ? &slabs : nullptr));
    HIP_TRY(launch_in_stats(slabs.count ? ws.partial : ws.cv, ws.mean2, ws.rstd2, ws.stats_part, ws.stats_part_floats, B,
                            H2 * W2, d.Cout, st, ws.pre, skip, tbias ? tbias + d.tb_off : nullptr, h->tb_width,
                            slabs.count ? &slabs : nullptr));
    RC_TRY(run_attention(h, 5 + k, ws.pre, out, B, H2 * W2, 1, st, next_wt, next_bt, ws.up, H2, W2, next_split,
                         next_stats_only));
    return DANRA_OK;
}

// up_ready: ws.up holds the transposed convolution and ws.stats_part its statistics in slices (the previous block's
// attention tail applied it); stats_only: only the statistics are there and `in` is that block's output - the
// transposed convolution is rebuilt inside the convolution kernel and the 64-channel tensor at full resolution never exists.
int run_final_block(danra_handle_s* h, const float* in, float* eps_nchw, int B, int Hh, int Ww, hipStream_t st,
                    bool up_ready = false, bool stats_only = false) {
    Workspace& ws = h->ws;
    if (stats_only) {
        HIP_TRY(launch_final_up_conv(in, h->final_wt, h->final_bt, h->final_conv.w, h->final_conv.bias, eps_nchw, B, Hh, Ww,
                                     h->cfg.out_channels, ws.stats_part, Hh * Ww / 8, 32, st));
        return DANRA_OK;
    }
    if (!up_ready) {
        RC_TRY(run_convT(h, h->final_wt, h->final_bt, 64, in, ws.up, B, Hh, Ww, st, &h->final_s_wt));
        HIP_TRY(launch_in_stats(ws.up, ws.mean1, ws.rstd1, ws.stats_part, ws.stats_part_floats, B, 4 * Hh * Ww, 64, st));
    } else if (final_conv_merges_slices(2 * Hh, 2 * Ww, 64)) {
        // the producer of ws.up left its statistics in slices of 32 samples; the convolution merges them itself
        HIP_TRY(launch_final_conv(ws.up, nullptr, nullptr, h->final_conv.w, h->final_conv.bias, eps_nchw, B, 2 * Hh, 2 * Ww,
                                  64, h->cfg.out_channels, st, ws.stats_part, Hh * Ww / 8, 32));
        return DANRA_OK;
    } else {
        HIP_TRY(launch_in_stats_from_slices(ws.stats_part, ws.mean1, ws.rstd1, B, Hh * Ww / 8, 32, st));
    }
    HIP_TRY(launch_final_conv(ws.up, ws.mean1, ws.rstd1, h->final_conv.w, h->final_conv.bias, eps_nchw, B, 2 * Hh, 2 * Ww,
                              64, h->cfg.out_channels, st));
    return DANRA_OK;
}

// `t` gives one timestep per image; with t == null every image uses the sampler state's current one.
int forward_impl(danra_handle_s* h, const float* x, const int64_t* t, SampleState* state, const int64_t* y,
                 const float* cond, const float* lsm, const float* topo, float* eps, int B, int H, int W,
                 hipStream_t st, bool stem_static_ready = false, const float* tb_table = nullptr,
                 bool tbias_ready = false) {
    Workspace& ws = h->ws;
    if (tbias_ready) {
        // sampling loop, one evaluation per step: the previous step's update kernel gathered this step's rows
    } else if (tb_table)  // sampling loop: the projections were tabulated for every timestep before the loop
        HIP_TRY(launch_tbias_gather(state, tb_table, y, ws.tbias, h->cfg.num_classes > 0 && y ? h->cfg.num_classes : 1,
                                    h->tb_width, B, st));
    else
        RC_TRY(run_time_mlp(h, t, state, y, B, nullptr, nullptr, ws.tbias, st));
    const float* planes[4] = {nullptr, nullptr, nullptr, nullptr};
    int np = 0;
    planes[np++] = x;
    if (h->cfg.has_lsm) planes[np++] = lsm;
    if (h->cfg.has_topo) planes[np++] = topo;
    if (h->cfg.has_cond) planes[np++] = cond;
    RC_TRY(run_stem(h, planes, ws.f1pre, B, H, W, ws.tbias + h->tb_off_enc[0], stem_static_ready, st));
    int Hc = H / 2, Wc = W / 2;
    RC_TRY(run_attention(h, 0, ws.f1pre, ws.fmap[0], B, Hc * Wc, 0, st));
    RC_TRY(run_conv(h, h->conv2, ws.fmap[0], ws.c2, B, Hc, Wc, nullptr, 1, nullptr, nullptr, nullptr, st));
    Hc /= 2; Wc /= 2;
    const float* cur = ws.c2;
    for (int L = 1; L <= 4; ++L) {
        RC_TRY(run_layer(h, L, cur, ws.fpre[L], B, Hc, Wc, ws.tbias + h->tb_off_enc[L], st));
        if (L > 1) { Hc /= 2; Wc /= 2; }
        RC_TRY(run_attention(h, L, ws.fpre[L], ws.fmap[L], B, Hc * Wc, 0, st));
        cur = ws.fmap[L];
    }
    // A 64-channel block's output feeds only the next block's transposed convolution, which is row-local like the
    // block's own output projection: the two run as one kernel and the block's output never reaches memory.
    const float* din = ws.fmap[4];
    bool up_ready = false, stats_only = false;
    for (int k = 0; k < 4; ++k) {
        float* dout = (k & 1) ? ws.outB : ws.outA;
        const float *next_wt = nullptr, *next_bt = nullptr;
        const SplitW* next_split = nullptr;
        // (measured: pays from 32768 token rows on; below that the per-workgroup weight staging outweighs the saved pass)
        if (h->dec[k].Cout == 64 && (long)B * 4 * Hc * Wc >= 32768) {
            next_wt = (k < 3) ? h->dec[k + 1].wt : h->final_wt;
            next_bt = (k < 3) ? h->dec[k + 1].bt : h->final_bt;
            next_split = (k < 3) ? &h->dec[k + 1].s_wt : &h->final_s_wt;
            // the final block rebuilds its transposed convolution from this block's output: only the statistics are formed here
            stats_only = (k == 3) && final_up_conv_supported(2 * Hc, 2 * Wc);
        }
        RC_TRY(run_decoder_block(h, k, din, ws.fmap[3 - k], ws.tbias, dout, B, Hc, Wc, st, up_ready, next_wt, next_bt,
                                 next_split, stats_only));
        up_ready = next_wt != nullptr;
        din = dout;
        Hc *= 2; Wc *= 2;
    }
    RC_TRY(run_final_block(h, din, eps, B, Hc, Wc, st, up_ready, stats_only));
    return DANRA_OK;
}

int check_ready(danra_handle_s* h, int B, int H, int W) {
    if (!h) return fail(DANRA_EINVAL, "null handle");
    if (!h->packed) return fail(DANRA_ESTATE, "danra_pack_weights has not been called");
    if (B <= 0 || H <= 0 || W <= 0 || H % 32 || W % 32)
        return fail(DANRA_EINVAL, "B=%d H=%d W=%d: H and W must be positive multiples of 32", B, H, W);
    // the kernels address every activation tensor with 32-bit byte offsets; the largest one holds 64 floats per pixel
    if ((unsigned long long)B * H * W >= (1ull << 24))
        return fail(DANRA_EINVAL, "B=%d H=%d W=%d: B*H*W must stay below 2^24 pixels per call (split the batch)", B, H, W);
    return DANRA_OK;
}

}  // namespace

static int resolve_scratch(float** out);

// ================================================================================================
extern "C" {

const char* danra_last_error(void) { return g_err; }

int danra_create(const danra_config_t* cfg, danra_handle_t* out) {
    if (!cfg || !out) return fail(DANRA_EINVAL, "null argument");
    if (cfg->time_embedding <= 0 || cfg->time_embedding % 8 || cfg->time_embedding > 512)
        return fail(DANRA_EINVAL, "time_embedding=%d must be a multiple of 8 and <= 512", cfg->time_embedding);
    if (cfg->n_heads != 1 && cfg->n_heads != 2 && cfg->n_heads != 4 && cfg->n_heads != 8)
        return fail(DANRA_EINVAL, "n_heads=%d unsupported (1, 2, 4 or 8)", cfg->n_heads);
    if (cfg->out_channels < 1 || cfg->out_channels > 4) return fail(DANRA_EINVAL, "out_channels must be 1..4");
    for (int i = 0; i < 4; ++i)
        if (cfg->block_layers[i] < 1 || cfg->block_layers[i] > 8) return fail(DANRA_EINVAL, "block_layers out of range");
    danra_handle_s* h = new danra_handle_s();
    h->cfg = *cfg;
    h->in_channels = 1 + (cfg->has_lsm ? 1 : 0) + (cfg->has_topo ? 1 : 0) + (cfg->has_cond ? 1 : 0);
    int off = 0;
    for (int i = 0; i < 5; ++i) { h->tb_off_enc[i] = off; off += kFmap[i]; }
    h->enc_width = off;
    for (int k = 0; k < 4; ++k) { h->dec[k].tb_off = off; off += kFmap[3 - k]; }
    h->tb_width = off;
    build_param_table(h);
    if (const char* g = getenv("DANRA_GRAPH")) h->graph_mode = (g[0] == '0') ? 0 : 1;
    const int half = cfg->time_embedding / 2;
    h->h_enc_inv_freq.resize(half);
    h->h_dec_div.resize(half);
    for (int k = 0; k < half; ++k) {
        const float e = (float)(2 * k) / (float)cfg->time_embedding;  // arange(0, dim, 2).float() / dim
        h->h_enc_inv_freq[k] = 1.0f / (float)pow(1000.0, (double)e);
        h->h_dec_div[k] = (float)pow(10000.0, (double)(2 * k) / (double)cfg->time_embedding);
    }
    // library set-up for this device: the address of the built-in scratch array (danra_instance_norm_noalloc)
    int dev = -1;
    if (hipGetDevice(&dev) == hipSuccess) {
        float* unused = nullptr;
        const int rc = resolve_scratch(&unused);
        if (rc != DANRA_OK) { delete h; return rc; }
    } else {
        (void)hipGetLastError();  // no device in this process: nothing to resolve
    }
    *out = h;
    return DANRA_OK;
}

int danra_destroy(danra_handle_t h) {
    if (!h) return DANRA_OK;
    if (h->arena) (void)hipFree(h->arena);
    if (h->arena_h) (void)hipFree(h->arena_h);
    drop_step_graph(h->ws);
    if (h->ws.base) (void)hipFree(h->ws.base);
    if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
    if (h->tb_table) (void)hipFree(h->tb_table);
    if (h->tb_table_u) (void)hipFree(h->tb_table_u);
    if (h->tb_index) (void)hipFree(h->tb_index);
    for (hipEvent_t e : h->prof_events) (void)hipEventDestroy(e);
    delete h;
    return DANRA_OK;
}

int danra_num_params(danra_handle_t h) { return h ? (int)h->params.size() : 0; }
const char* danra_param_name(danra_handle_t h, int i) {
    return (h && i >= 0 && i < (int)h->params.size()) ? h->params[i].name.c_str() : nullptr;
}
size_t danra_param_numel(danra_handle_t h, int i) {
    return (h && i >= 0 && i < (int)h->params.size()) ? h->params[i].numel : 0;
}

int danra_set_param(danra_handle_t h, const char* name, const float* value_dev, size_t numel) {
    if (!h || !name) return fail(DANRA_EINVAL, "null argument");
    auto it = h->index.find(name);
    if (it == h->index.end()) return fail(DANRA_ENAME, "unknown state_dict entry '%s'", name);
    ParamEntry& p = h->params[it->second];
    if (p.numel != numel) return fail(DANRA_EINVAL, "'%s': expected %zu elements, got %zu", name, p.numel, numel);
    if (!value_dev) return fail(DANRA_EINVAL, "'%s': null pointer", name);
    p.src = value_dev;
    return DANRA_OK;
}

int danra_pack_weights(danra_handle_t h, void* stream) {
    if (!h) return fail(DANRA_EINVAL, "null handle");
    for (const ParamEntry& p : h->params)
        if (!p.src) return fail(DANRA_ENAME, "state_dict entry '%s' was never set", p.name.c_str());
    hipStream_t st = (hipStream_t)stream;
    std::vector<SplitJob> jobs;
    if (!h->arena) {
        Bump probe(nullptr), probe_h(nullptr);
        RC_TRY(layout_and_pack(h, probe, probe_h, jobs, false, st));
        h->arena_bytes = probe.off;
        h->arena_h_bytes = probe_h.off + jobs.size() * sizeof(float);
        if (hipMalloc(&h->arena, h->arena_bytes) != hipSuccess) {
            h->arena = nullptr;
            return fail(DANRA_ENOMEM, "weight arena allocation of %zu bytes failed", h->arena_bytes);
        }
        if (hipMalloc(&h->arena_h, h->arena_h_bytes) != hipSuccess) {
            (void)hipFree(h->arena);
            h->arena = nullptr; h->arena_h = nullptr;
            return fail(DANRA_ENOMEM, "split-weight arena allocation of %zu bytes failed", h->arena_h_bytes);
        }
        jobs.clear();
    }
    Bump bp(h->arena), bh(h->arena_h);
    RC_TRY(layout_and_pack(h, bp, bh, jobs, true, st));
    RC_TRY(split_registered(h, jobs, (float*)((char*)h->arena_h + bh.off), st));
    for (int k = 0; k < 4; ++k) h->attn[5 + k] = h->dec[k].attn;  // (the copies made during the layout predate the scales)
    {   // the weights of the convolutions that go through the window kernels in fragment order, with the scale of their split copy
        auto frag = [&](ConvP& c) -> int {
            if (c.wfrag && !c.split.wh) c.wfrag = nullptr;
            if (c.wfrag) HIP_TRY(launch_pack_conv_frag(c.w, c.wfrag, c.wrows ? c.wrows : c.Cout, c.Cin, 1.0f / c.split.inv_scale, st, c.KH * c.KW));
            return DANRA_OK;
        };
        RC_TRY(frag(h->conv2));
        for (auto& layer : h->blocks)
            for (BlockP& blk : layer) { RC_TRY(frag(blk.c1)); RC_TRY(frag(blk.c2)); }
        for (DecP& d : h->dec) RC_TRY(frag(d.conv));
    }
    for (DecP& d : h->dec) {
        if (d.wt_frag && !d.s_wt.wh) d.wt_frag = nullptr;
        if (d.wt_frag) HIP_TRY(launch_frag16_from_f32(d.wt, d.wt_frag, 4 * d.Cin, d.Cin, 1.0f / d.s_wt.inv_scale, st));
    }
    for (AttnP& at : h->attn) {  // (an all-zero or non-finite matrix has no split copy and keeps the fp32 form)
        if (at.w_in_frag && !at.s_in.wh) at.w_in_frag = nullptr;
        if (at.w_out_frag && !at.s_out.wh) at.w_out_frag = nullptr;
        if (at.C == 64) {
            if (at.w_in_frag) HIP_TRY(launch_frag16_from_f32(at.w_in, at.w_in_frag, 3 * at.C, at.C, 1.0f / at.s_in.inv_scale, st));
            if (at.w_out_frag) HIP_TRY(launch_frag16_from_f32(at.w_out, at.w_out_frag, at.C, at.C, 1.0f / at.s_out.inv_scale, st));
            continue;
        }
        if (at.w_in_frag) HIP_TRY(launch_split_to_frag16(at.s_in.wh, at.w_in_frag, 3 * at.C, at.C, st));
        if (at.w_out_frag) HIP_TRY(launch_split_to_frag16(at.s_out.wh, at.w_out_frag, at.C, at.C, st));
        if (at.w_ff1_frag && !at.s_ff1.wh) at.w_ff1_frag = nullptr;
        if (at.w_ff2_frag && !at.s_ff2.wh) at.w_ff2_frag = nullptr;
        if (at.w_ff1_frag) HIP_TRY(launch_split_to_frag16(at.s_ff1.wh, at.w_ff1_frag, at.C, at.C, st));
        if (at.w_ff2_frag) HIP_TRY(launch_split_to_frag16(at.s_ff2.wh, at.w_ff2_frag, at.C, at.C, st));
    }
    h->freq_dirty = true;
    RC_TRY(upload_freq(h, st));
    for (ParamEntry& p : h->params) p.src = nullptr;  // borrowed pointers are not kept
    h->packed = true;
    return DANRA_OK;
}

int danra_set_freq_tables(danra_handle_t h, const float* enc_inv_freq, const float* dec_div, int n) {
    if (!h || !enc_inv_freq || !dec_div) return fail(DANRA_EINVAL, "null argument");
    if (n != h->cfg.time_embedding / 2) return fail(DANRA_EINVAL, "expected %d entries", h->cfg.time_embedding / 2);
    memcpy(h->h_enc_inv_freq.data(), enc_inv_freq, (size_t)n * 4);
    memcpy(h->h_dec_div.data(), dec_div, (size_t)n * 4);
    h->freq_dirty = true;
    return DANRA_OK;
}

int danra_unet_forward(danra_handle_t h, const float* x, const int64_t* t, const int64_t* y, const float* cond,
                       const float* lsm, const float* topo, float* eps, int B, int H, int W, void* stream) {
    RC_TRY(check_ready(h, B, H, W));
    if (!x || !t || !eps) return fail(DANRA_EINVAL, "x, t and eps_out are required");
    if (h->cfg.has_lsm && !lsm) return fail(DANRA_EINVAL, "lsm_cond is required by this model");
    if (h->cfg.has_topo && !topo) return fail(DANRA_EINVAL, "topo_cond is required by this model");
    if (h->cfg.has_cond && !cond) return fail(DANRA_EINVAL, "cond_img is required by this model");
    RC_TRY(ensure_workspace(h, B, H, W));
    return forward_impl(h, x, t, nullptr, y, cond, lsm, topo, eps, B, H, W, (hipStream_t)stream);
}

int danra_get_fmaps(danra_handle_t h, float* f1, float* f2, float* f3, float* f4, float* f5, void* stream) {
    if (!h || !h->ws.base) return fail(DANRA_ESTATE, "no forward has run");
    float* dst[5] = {f1, f2, f3, f4, f5};
    int hw = (h->ws.H / 2) * (h->ws.W / 2);
    for (int i = 0; i < 5; ++i) {
        if (dst[i]) HIP_TRY(launch_nhwc_to_nchw(h->ws.fmap[i], dst[i], h->ws.B, kFmap[i], hw, (hipStream_t)stream));
        hw /= 4;
    }
    return DANRA_OK;
}

int danra_ddpm_update(float* x, const float* eps, const float* noise, float c1, float c2, float c3, size_t n,
                      void* stream) {
    if (!x || !eps) return fail(DANRA_EINVAL, "null argument");
    HIP_TRY(launch_ddpm_update(x, eps, noise, c1, c2, c3, n, 0, 0, 0, (hipStream_t)stream));
    return DANRA_OK;
}

int danra_ddpm_update_philox(float* x, const float* eps, float c1, float c2, float c3, size_t n, uint64_t seed,
                             uint64_t step, void* stream) {
    if (!x || !eps) return fail(DANRA_EINVAL, "null argument");
    if (n % 4) return fail(DANRA_EINVAL, "n must be a multiple of 4");
    HIP_TRY(launch_ddpm_update(x, eps, nullptr, c1, c2, c3, n, 1, seed, step, (hipStream_t)stream));
    return DANRA_OK;
}

int danra_noise_image(const float* x, const int64_t* t, const float* alpha_hat, int T, float* noisy, float* noise, int B,
                      size_t per_image, int noise_mode, uint64_t seed, uint64_t offset, void* stream) {
    if (!x || !t || !alpha_hat || !noisy || !noise) return fail(DANRA_EINVAL, "null argument");
    if (B <= 0 || T <= 0 || per_image == 0 || per_image % 4)
        return fail(DANRA_EINVAL, "B=%d T=%d per_image=%zu: per_image must be a positive multiple of 4", B, T, per_image);
    if (noise_mode != DANRA_NOISE_HOST && noise_mode != DANRA_NOISE_PHILOX) return fail(DANRA_EINVAL, "unknown noise mode");
    HIP_TRY(launch_noise_image(x, t, alpha_hat, T, noisy, noise, B, per_image, noise_mode == DANRA_NOISE_PHILOX ? 1 : 0, seed,
                               offset, (hipStream_t)stream));
    return DANRA_OK;
}

static int check_core_shape(int B, int heads, int L, int head_dim) {
    if (head_dim != 16)
        return fail(DANRA_EINVAL, "head_dim %d: the differentiable attention core exists for head_dim 16 only", head_dim);
    if (B <= 0 || heads <= 0 || L <= 0 || L % 128)
        return fail(DANRA_EINVAL, "B=%d heads=%d L=%d: L must be a positive multiple of 128", B, heads, L);
    if ((long long)B * heads * L >= (1LL << 27)) return fail(DANRA_EINVAL, "B*heads*L = %lld is too large", (long long)B * heads * L);
    return DANRA_OK;
}

int danra_attention_core_fwd(const float* q, const float* k, const float* v, float* out, float* lse2, int B, int heads,
                             int L, int head_dim, void* stream) {
    if (!q || !k || !v || !out) return fail(DANRA_EINVAL, "null argument");
    if (int rc = check_core_shape(B, heads, L, head_dim)) return rc;
    HIP_TRY(launch_attention_hd16(q, k, v, out, lse2, B, heads, L, (hipStream_t)stream));
    return DANRA_OK;
}

int danra_attention_core_presplit_fwd(const float* q, const float* k, const float* v, float* out, int B, int heads, int L,
                                      int head_dim, void* stream) {
    if (!q || !k || !v || !out) return fail(DANRA_EINVAL, "null argument");
    if (int rc = check_core_shape(B, heads, L, head_dim)) return rc;
    hipStream_t st = (hipStream_t)stream;
    const size_t plane = (size_t)B * heads * L * 16;
    float* tmp = nullptr;
    if (hipMalloc((void**)&tmp, (2 * plane + (size_t)B * heads * (L / 16) * 2) * sizeof(float)) != hipSuccess)
        return fail(DANRA_ENOMEM, "temporary buffer for the split K and V");
    int* flags = reinterpret_cast<int*>(tmp + 2 * plane);
    hipError_t e = launch_presplit_kv(k, v, tmp, tmp + plane, flags, B * heads, L, st);
    if (e == hipSuccess) e = launch_attention_hd16_pre(q, tmp, tmp + plane, k, v, flags, out, B, heads, L, st);
    const hipError_t se = hipStreamSynchronize(st);
    (void)hipFree(tmp);
    if (e != hipSuccess || se != hipSuccess) return fail(DANRA_EHIP, "pre-split attention core: %s", hipGetErrorString(e != hipSuccess ? e : se));
    return DANRA_OK;
}

int danra_set_fp32_products(int fp32_products) {
    set_attention_hd16_fp32_products(fp32_products);
    set_igemm_fp32_products(fp32_products);
    return DANRA_OK;
}

int danra_attention_core_bwd(const float* q, const float* k, const float* v, const float* out, const float* dout,
                             const float* lse2, float* dq, float* dk, float* dv, float* delta, int B, int heads, int L,
                             int head_dim, void* stream) {
    if (!q || !k || !v || !out || !dout || !lse2 || !dq || !dk || !dv || !delta) return fail(DANRA_EINVAL, "null argument");
    if (int rc = check_core_shape(B, heads, L, head_dim)) return rc;
    HIP_TRY(launch_attention_hd16_bwd(q, k, v, out, dout, lse2, dq, dk, dv, delta, B, heads, L, (hipStream_t)stream));
    return DANRA_OK;
}

int danra_attention_core_qkv_fwd(const float* qkv, float* out, float* lse2, int B, int heads, int L, int head_dim, void* stream) {
    if (!qkv || !out) return fail(DANRA_EINVAL, "null argument");
    if (int rc = check_core_shape(B, heads, L, head_dim)) return rc;
    HIP_TRY(launch_attention_hd16_tokens(qkv, out, lse2, B, heads, L, (hipStream_t)stream));
    return DANRA_OK;
}

int danra_attention_core_qkv_bwd(const float* qkv, const float* out, const float* dout, const float* lse2, float* dqkv,
                                 float* delta, int B, int heads, int L, int head_dim, void* stream) {
    if (!qkv || !out || !dout || !lse2 || !dqkv || !delta) return fail(DANRA_EINVAL, "null argument");
    if (int rc = check_core_shape(B, heads, L, head_dim)) return rc;
    HIP_TRY(launch_attention_hd16_bwd_tokens(qkv, out, dout, lse2, dqkv, delta, B, heads, L, (hipStream_t)stream));
    return DANRA_OK;
}

int danra_attention_small_supported(int B, int heads, int L, int head_dim) {
    return attention_small_train_supported(B, heads, L, head_dim) ? 1 : 0;
}

int danra_attention_small_fwd(const float* q, const float* k, const float* v, float* out, int B, int heads, int L, int head_dim,
                              void* stream) {
    if (!q || !k || !v || !out) return fail(DANRA_EINVAL, "null argument");
    if (!attention_small_train_supported(B, heads, L, head_dim))
        return fail(DANRA_EINVAL, "B=%d heads=%d L=%d head_dim=%d: the short-sequence core takes 1..64 tokens, head_dim a multiple of 4 "
                    "up to 128 and at most 2048 values per head", B, heads, L, head_dim);
    HIP_TRY(launch_attention_small_fwd(q, k, v, out, B, heads, L, head_dim, (hipStream_t)stream));
    return DANRA_OK;
}

int danra_attention_small_bwd(const float* q, const float* k, const float* v, const float* dout, float* dq, float* dk, float* dv,
                              int B, int heads, int L, int head_dim, void* stream) {
    if (!q || !k || !v || !dout || !dq || !dk || !dv) return fail(DANRA_EINVAL, "null argument");
    if (!attention_small_train_supported(B, heads, L, head_dim))
        return fail(DANRA_EINVAL, "B=%d heads=%d L=%d head_dim=%d: not a shape of the short-sequence core (danra_attention_small_supported)",
                    B, heads, L, head_dim);
    HIP_TRY(launch_attention_small_bwd(q, k, v, dout, dq, dk, dv, B, heads, L, head_dim, (hipStream_t)stream));
    return DANRA_OK;
}

int danra_attention_small_qkv_fwd(const float* qkv, float* out, int B, int heads, int L, int head_dim, void* stream) {
    if (!qkv || !out) return fail(DANRA_EINVAL, "null argument");
    if (!attention_small_train_supported(B, heads, L, head_dim))
        return fail(DANRA_EINVAL, "B=%d heads=%d L=%d head_dim=%d: not a shape of the short-sequence core (danra_attention_small_supported)",
                    B, heads, L, head_dim);
    HIP_TRY(launch_attention_small_qkv_fwd(qkv, out, B, heads, L, head_dim, (hipStream_t)stream));
    return DANRA_OK;
}

int danra_attention_small_qkv_bwd(const float* qkv, const float* dout, float* dqkv, int B, int heads, int L, int head_dim,
                                  void* stream) {
    if (!qkv || !dout || !dqkv) return fail(DANRA_EINVAL, "null argument");
    if (!attention_small_train_supported(B, heads, L, head_dim))
        return fail(DANRA_EINVAL, "B=%d heads=%d L=%d head_dim=%d: not a shape of the short-sequence core (danra_attention_small_supported)",
                    B, heads, L, head_dim);
    HIP_TRY(launch_attention_small_qkv_bwd(qkv, dout, dqkv, B, heads, L, head_dim, (hipStream_t)stream));
    return DANRA_OK;
}

int danra_batchnorm_train_fwd(const float* x, const float* gamma, const float* beta, float* y, float* save_mean,
                              float* save_rstd, float* running_mean, float* running_var, int B, int C, int HW, float eps,
                              float momentum, int relu, long long* num_batches_tracked, void* stream) {
    if (!x || !gamma || !beta || !y || !save_mean || !save_rstd) return fail(DANRA_EINVAL, "null argument");
    if ((running_mean == nullptr) != (running_var == nullptr)) return fail(DANRA_EINVAL, "running_mean and running_var go together");
    if (B <= 0 || C <= 0 || HW <= 0 || HW % 4 || (long long)B * HW < 2 || (long long)B * C * HW >= (1LL << 31))
        return fail(DANRA_EINVAL, "B=%d C=%d HW=%d: HW must be a positive multiple of 4, at least two values per channel, "
                    "fewer than 2^31 elements", B, C, HW);
    HIP_TRY(launch_bn_train_fwd(x, gamma, beta, y, save_mean, save_rstd, running_mean, running_var, B, C, HW, eps, momentum,
                                relu ? 1 : 0, (hipStream_t)stream, nullptr, num_batches_tracked));
    return DANRA_OK;
}

int danra_batchnorm_train_bwd(const float* x, const float* dy, const float* y, const float* gamma, const float* save_mean,
                              const float* save_rstd, float* dx, float* dgamma, float* dbeta, int B, int C, int HW, int relu,
                              void* stream) {
    if (!x || !dy || !gamma || !save_mean || !save_rstd || !dx || !dgamma || !dbeta || (relu && !y))
        return fail(DANRA_EINVAL, "null argument");
    if (B <= 0 || C <= 0 || HW <= 0 || HW % 4 || (long long)B * C * HW >= (1LL << 31))
        return fail(DANRA_EINVAL, "B=%d C=%d HW=%d: HW must be a positive multiple of 4, fewer than 2^31 elements", B, C, HW);
    HIP_TRY(launch_bn_train_bwd(x, dy, y, gamma, save_mean, save_rstd, dx, dgamma, dbeta, B, C, HW, relu ? 1 : 0,
                                (hipStream_t)stream));
    return DANRA_OK;
}

int danra_batchnorm_train_add_relu_fwd(const float* x, const float* res, const float* gamma, const float* beta, float* y,
                                       float* save_mean, float* save_rstd, float* running_mean, float* running_var, int B, int C,
                                       int HW, float eps, float momentum, long long* num_batches_tracked, void* stream) {
    if (!x || !res || !gamma || !beta || !y || !save_mean || !save_rstd) return fail(DANRA_EINVAL, "null argument");
    if ((running_mean == nullptr) != (running_var == nullptr)) return fail(DANRA_EINVAL, "running_mean and running_var go together");
    if (B <= 0 || C <= 0 || HW <= 0 || HW % 4 || (long long)B * HW < 2 || (long long)B * C * HW >= (1LL << 31))
        return fail(DANRA_EINVAL, "B=%d C=%d HW=%d: HW must be a positive multiple of 4, at least two values per channel, "
                    "fewer than 2^31 elements", B, C, HW);
    HIP_TRY(launch_bn_train_fwd(x, gamma, beta, y, save_mean, save_rstd, running_mean, running_var, B, C, HW, eps, momentum, 1,
                                (hipStream_t)stream, res, num_batches_tracked));
    return DANRA_OK;
}

int danra_batchnorm_train_add_relu_bwd(const float* x, const float* dy, const float* y, const float* gamma,
                                       const float* save_mean, const float* save_rstd, float* dx, float* dres, float* dgamma,
                                       float* dbeta, int B, int C, int HW, void* stream) {
    if (!x || !dy || !y || !gamma || !save_mean || !save_rstd || !dx || !dres || !dgamma || !dbeta)
        return fail(DANRA_EINVAL, "null argument");
    if (B <= 0 || C <= 0 || HW <= 0 || HW % 4 || (long long)B * C * HW >= (1LL << 31))
        return fail(DANRA_EINVAL, "B=%d C=%d HW=%d: HW must be a positive multiple of 4, fewer than 2^31 elements", B, C, HW);
    HIP_TRY(launch_bn_train_bwd(x, dy, y, gamma, save_mean, save_rstd, dx, dgamma, dbeta, B, C, HW, 1, (hipStream_t)stream, dres));
    return DANRA_OK;
}

int danra_instancenorm_fwd(const float* x, float* y, float* save_mean, float* save_rstd, int planes, int HW, float eps,
                           void* stream) {
    if (!x || !y || !save_mean || !save_rstd) return fail(DANRA_EINVAL, "null argument");
    if (planes <= 0 || HW <= 0 || HW % 4 || (long long)planes * HW >= (1LL << 31))
        return fail(DANRA_EINVAL, "planes=%d HW=%d: HW must be a positive multiple of 4, fewer than 2^31 elements", planes, HW);
    HIP_TRY(launch_in_train_fwd(x, y, save_mean, save_rstd, planes, HW, eps, (hipStream_t)stream));
    return DANRA_OK;
}

int danra_instancenorm_bwd(const float* x, const float* dy, const float* save_mean, const float* save_rstd, float* dx,
                           int planes, int HW, void* stream) {
    if (!x || !dy || !save_mean || !save_rstd || !dx) return fail(DANRA_EINVAL, "null argument");
    if (planes <= 0 || HW <= 0 || HW % 4 || (long long)planes * HW >= (1LL << 31))
        return fail(DANRA_EINVAL, "planes=%d HW=%d: HW must be a positive multiple of 4, fewer than 2^31 elements", planes, HW);
    HIP_TRY(launch_in_train_bwd(x, dy, save_mean, save_rstd, dx, planes, HW, (hipStream_t)stream));
    return DANRA_OK;
}

int danra_linear_wgrad_supported(int M, int N, int K) { return linear_wgrad_supported(M, N, K) ? 1 : 0; }

size_t danra_linear_wgrad_workspace(int M, int N, int K) {
    return linear_wgrad_supported(M, N, K) ? linear_wgrad_workspace_floats(M, N, K) * sizeof(float) : 0;
}

int danra_linear_wgrad(const float* dy, const float* x, float* dw, float* db, int M, int N, int K, void* workspace,
                       void* stream) {
    if (!dy || !x || !dw || !workspace) return fail(DANRA_EINVAL, "null argument");
    if (!linear_wgrad_supported(M, N, K))
        return fail(DANRA_EINVAL, "M=%d N=%d K=%d: supported are K = 64 with N = 64, 128 or 192 and M > 0", M, N, K);
    HIP_TRY(launch_linear_wgrad(dy, x, dw, db, M, N, K, static_cast<float*>(workspace), (hipStream_t)stream));
    return DANRA_OK;
}

int danra_linear_supported(int M, int N, int K) {
    return (M > 0 && N > 0 && K > 0 && N % 64 == 0 && K % 32 == 0 && (long long)M * N < (1LL << 31) &&
            (long long)M * K < (1LL << 31)) ? 1 : 0;
}

int danra_linear(const float* x, const float* w, const float* bias, float* y, int M, int N, int K, void* stream) {
    if (!x || !w || !y) return fail(DANRA_EINVAL, "null argument");
    if (!danra_linear_supported(M, N, K))
        return fail(DANRA_EINVAL, "M=%d N=%d K=%d: N must be a multiple of 64, K of 32, fewer than 2^31 elements", M, N, K);
    IGemmArgs a = blank_args();
    a.A = x; a.lda = K; a.a_mode = A_GEMM;
    a.W = w; a.bias = bias; a.out = y;
    a.M = M; a.N = N; a.K = K;
    a.st_mode = ST_ROW; a.ldc = N;
    HIP_TRY(launch_igemm(a, (hipStream_t)stream));
    return DANRA_OK;
}

int danra_conv3x3_train_supported(int B, int Cin, int Cout, int H, int W) {
    return conv3x3_train_supported(B, Cin, Cout, H, W) ? 1 : 0;
}

size_t danra_conv3x3_train_workspace(int B, int Cin, int Cout, int H, int W) {
    return conv3x3_train_supported(B, Cin, Cout, H, W) ? conv3x3_train_workspace_floats(B, Cin, Cout, H, W) * sizeof(float) : 0;
}

static int conv3x3_train_check(const void* a, const void* b, const void* c, const void* ws, int B, int Cin, int Cout, int H,
                               int W) {
    if (!a || !b || !c || !ws) return fail(DANRA_EINVAL, "null argument");
    if (!conv3x3_train_supported(B, Cin, Cout, H, W))
        return fail(DANRA_EINVAL, "B=%d Cin=%d Cout=%d H=%d W=%d: not a shape the training convolution kernels take "
                    "(danra_conv3x3_train_supported)", B, Cin, Cout, H, W);
    return DANRA_OK;
}

int danra_conv3x3_train_fwd(const float* x, const float* w, float* y, int B, int Cin, int Cout, int H, int W, void* workspace,
                            void* stream) {
    RC_TRY(conv3x3_train_check(x, w, y, workspace, B, Cin, Cout, H, W));
    HIP_TRY(launch_conv3x3_train(false, x, w, y, B, Cin, Cout, H, W, static_cast<float*>(workspace), (hipStream_t)stream));
    return DANRA_OK;
}

int danra_conv3x3_train_dgrad(const float* dy, const float* w, float* dx, int B, int Cin, int Cout, int H, int W,
                              void* workspace, void* stream) {
    RC_TRY(conv3x3_train_check(dy, w, dx, workspace, B, Cin, Cout, H, W));
    HIP_TRY(launch_conv3x3_train(true, dy, w, dx, B, Cin, Cout, H, W, static_cast<float*>(workspace), (hipStream_t)stream));
    return DANRA_OK;
}

int danra_conv3x3_train_wgrad(const float* x, const float* dy, float* dw, int B, int Cin, int Cout, int H, int W,
                              void* workspace, void* stream) {
    RC_TRY(conv3x3_train_check(x, dy, dw, workspace, B, Cin, Cout, H, W));
    HIP_TRY(launch_conv3x3_wgrad_train(x, dy, dw, B, Cin, Cout, H, W, static_cast<float*>(workspace), (hipStream_t)stream));
    return DANRA_OK;
}

int danra_conv3x3_c1_train_supported(int B, int Cin, int H, int W) { return conv_c1_train_supported(B, Cin, H, W) ? 1 : 0; }

size_t danra_conv3x3_c1_train_workspace(int B, int Cin, int H, int W) { return conv_c1_train_workspace_floats(B, Cin, H, W) * sizeof(float); }

static int conv_c1_check(const void* a, const void* b, const void* c, int B, int Cin, int H, int W) {
    if (!a || !b || !c) return fail(DANRA_EINVAL, "null argument");
    if (!conv_c1_train_supported(B, Cin, H, W))
        return fail(DANRA_EINVAL, "B=%d Cin=%d H=%d W=%d: not a shape of the one-output-channel convolution kernels "
                    "(danra_conv3x3_c1_train_supported)", B, Cin, H, W);
    return DANRA_OK;
}

int danra_conv3x3_c1_train_fwd(const float* x, const float* w, float* y, int B, int Cin, int H, int W, void* stream) {
    RC_TRY(conv_c1_check(x, w, y, B, Cin, H, W));
    HIP_TRY(launch_conv_c1_train_fwd(x, w, y, B, Cin, H, W, (hipStream_t)stream));
    return DANRA_OK;
}

int danra_conv3x3_c1_train_dgrad(const float* dy, const float* w, float* dx, int B, int Cin, int H, int W, void* stream) {
    RC_TRY(conv_c1_check(dy, w, dx, B, Cin, H, W));
    HIP_TRY(launch_conv_c1_train_dgrad(dy, w, dx, B, Cin, H, W, (hipStream_t)stream));
    return DANRA_OK;
}

int danra_conv3x3_c1_train_wgrad(const float* x, const float* dy, float* dw, int B, int Cin, int H, int W, void* workspace,
                                 void* stream) {
    RC_TRY(conv_c1_check(x, dy, dw, B, Cin, H, W));
    if (!workspace) return fail(DANRA_EINVAL, "null workspace");
    HIP_TRY(launch_conv_c1_train_wgrad(x, dy, dw, B, Cin, H, W, static_cast<float*>(workspace), (hipStream_t)stream));
    return DANRA_OK;
}

int danra_conv3x3s2_train_supported(int B, int Cin, int Cout, int H, int W) {
    return conv3x3s2_train_supported(B, Cin, Cout, H, W) ? 1 : 0;
}

size_t danra_conv3x3s2_train_workspace(int B, int Cin, int Cout, int H, int W) {
    return conv3x3s2_train_supported(B, Cin, Cout, H, W) ? conv3x3s2_train_workspace_floats(B, Cin, Cout, H, W) * sizeof(float) : 0;
}

static int conv3s2_train_check(const void* a, const void* b, const void* c, const void* ws, int B, int Cin, int Cout, int H,
                               int W) {
    if (!a || !b || !c || !ws) return fail(DANRA_EINVAL, "null argument");
    if (!conv3x3s2_train_supported(B, Cin, Cout, H, W))
        return fail(DANRA_EINVAL, "B=%d Cin=%d Cout=%d H=%d W=%d: not a shape the 3x3 stride-2 training kernels take "
                    "(danra_conv3x3s2_train_supported)", B, Cin, Cout, H, W);
    return DANRA_OK;
}

int danra_conv3x3s2_train_fwd(const float* x, const float* w, float* y, int B, int Cin, int Cout, int H, int W, void* workspace,
                              void* stream) {
    RC_TRY(conv3s2_train_check(x, w, y, workspace, B, Cin, Cout, H, W));
    HIP_TRY(launch_conv3x3s2_train_fwd(x, w, y, B, Cin, Cout, H, W, static_cast<float*>(workspace), (hipStream_t)stream));
    return DANRA_OK;
}

int danra_conv3x3s2_train_dgrad(const float* dy, const float* w, float* dx, int B, int Cin, int Cout, int H, int W,
                                void* workspace, void* stream) {
    RC_TRY(conv3s2_train_check(dy, w, dx, workspace, B, Cin, Cout, H, W));
    HIP_TRY(launch_conv3x3s2_train_dgrad(dy, w, dx, B, Cin, Cout, H, W, static_cast<float*>(workspace), (hipStream_t)stream));
    return DANRA_OK;
}

int danra_conv3x3s2_train_wgrad(const float* x, const float* dy, float* dw, int B, int Cin, int Cout, int H, int W,
                                void* workspace, void* stream) {
    RC_TRY(conv3s2_train_check(x, dy, dw, workspace, B, Cin, Cout, H, W));
    HIP_TRY(launch_conv3x3s2_train_wgrad(x, dy, dw, B, Cin, Cout, H, W, static_cast<float*>(workspace), (hipStream_t)stream));
    return DANRA_OK;
}

int danra_conv8x8s2_train_supported(int B, int Cin, int Cout, int H, int W) {
    return conv8s2_train_supported(B, Cin, Cout, H, W) ? 1 : 0;
}

int danra_conv8x8s2_train_dgrad_supported(int B, int Cin, int Cout, int H, int W) {
    return conv8s2_train_dgrad_supported(B, Cin, Cout, H, W) ? 1 : 0;
}

size_t danra_conv8x8s2_train_workspace(int B, int Cin, int Cout, int H, int W) {
    return conv8s2_train_supported(B, Cin, Cout, H, W) ? conv8s2_train_workspace_floats(B, Cin, Cout, H, W) * sizeof(float) : 0;
}

static int conv8_train_check(const void* a, const void* b, const void* c, int B, int Cin, int Cout, int H, int W, bool dgrad) {
    if (!a || !b || !c) return fail(DANRA_EINVAL, "null argument");
    if (!(dgrad ? conv8s2_train_dgrad_supported(B, Cin, Cout, H, W) : conv8s2_train_supported(B, Cin, Cout, H, W)))
        return fail(DANRA_EINVAL, "B=%d Cin=%d Cout=%d H=%d W=%d: not a shape the 8x8 stride-2 training kernels take "
                    "(danra_conv8x8s2_train_%ssupported)", B, Cin, Cout, H, W, dgrad ? "dgrad_" : "");
    return DANRA_OK;
}

int danra_conv8x8s2_train_fwd(const float* x, const float* w, float* y, int B, int Cin, int Cout, int H, int W, void* stream) {
    RC_TRY(conv8_train_check(x, w, y, B, Cin, Cout, H, W, false));
    HIP_TRY(launch_conv8s2_train_fwd(x, w, y, B, Cin, Cout, H, W, (hipStream_t)stream));
    return DANRA_OK;
}

int danra_conv8x8s2_train_dgrad(const float* dy, const float* w, float* dx, int B, int Cin, int Cout, int H, int W, void* stream) {
    RC_TRY(conv8_train_check(dy, w, dx, B, Cin, Cout, H, W, true));
    HIP_TRY(launch_conv8s2_train_dgrad(dy, w, dx, B, Cin, Cout, H, W, (hipStream_t)stream));
    return DANRA_OK;
}

int danra_conv8x8s2_train_wgrad(const float* x, const float* dy, float* dw, int B, int Cin, int Cout, int H, int W,
                                void* workspace, void* stream) {
    RC_TRY(conv8_train_check(x, dy, dw, B, Cin, Cout, H, W, false));
    if (!workspace) return fail(DANRA_EINVAL, "null workspace");
    HIP_TRY(launch_conv8s2_train_wgrad(x, dy, dw, B, Cin, Cout, H, W, static_cast<float*>(workspace), (hipStream_t)stream));
    return DANRA_OK;
}

size_t danra_channel_sum_workspace(int B, int C, int HW) {
    if (B <= 0 || C <= 0 || HW <= 0 || HW % 4) return 0;
    const int S = channel_sum_slices(B, C, HW);
    return S > 1 ? (size_t)S * C * sizeof(float) : 0;
}

int danra_channel_sum(const float* x, float* out, int B, int C, int HW, void* workspace, void* stream) {
    if (!x || !out) return fail(DANRA_EINVAL, "null argument");
    if (B <= 0 || C <= 0 || HW <= 0 || HW % 4 || (long long)B * C * HW >= (1LL << 31))
        return fail(DANRA_EINVAL, "B=%d C=%d HW=%d: HW must be a positive multiple of 4, fewer than 2^31 elements", B, C, HW);
    HIP_TRY(launch_channel_sum(x, out, B, C, HW, static_cast<float*>(workspace), (hipStream_t)stream));
    return DANRA_OK;
}

int danra_ln_tokens_supported(int B, int C, int L) { return ln_tokens_supported(B, C, L) ? 1 : 0; }

size_t danra_ln_tokens_workspace(int B, int C, int L) { return ln_tokens_workspace_floats(B, C, L) * sizeof(float); }

int danra_ln_tokens_fwd(const float* x, const float* gamma, const float* beta, float* normed, float* mean, float* rstd, int B,
                        int C, int L, float eps, void* stream) {
    if (!x || !gamma || !beta || !normed || !mean || !rstd) return fail(DANRA_EINVAL, "null argument");
    if (!ln_tokens_supported(B, C, L)) return fail(DANRA_EINVAL, "B=%d C=%d L=%d: not supported (danra_ln_tokens_supported)", B, C, L);
    HIP_TRY(launch_ln_tokens_fwd(x, gamma, beta, normed, mean, rstd, B, C, L, eps, (hipStream_t)stream));
    return DANRA_OK;
}

int danra_ln_tokens_bwd(const float* dn, const float* x, const float* mean, const float* rstd, const float* gamma, float* dx,
                        float* dgamma_dbeta, int B, int C, int L, void* workspace, void* stream) {
    if (!dn || !x || !mean || !rstd || !gamma || !dx || !dgamma_dbeta || !workspace) return fail(DANRA_EINVAL, "null argument");
    if (!ln_tokens_supported(B, C, L)) return fail(DANRA_EINVAL, "B=%d C=%d L=%d: not supported (danra_ln_tokens_supported)", B, C, L);
    HIP_TRY(launch_ln_tokens_bwd(dn, x, mean, rstd, gamma, dx, dgamma_dbeta, B, C, L, static_cast<float*>(workspace),
                                 (hipStream_t)stream));
    return DANRA_OK;
}

int danra_tokens_to_nchw_add(const float* tokens, const float* add, float* out, int B, int C, int L, void* stream) {
    if (!tokens || !out) return fail(DANRA_EINVAL, "null argument");
    if (B <= 0 || C <= 0 || L <= 0 || B > 65535 || (long long)B * C * L >= (1LL << 31))
        return fail(DANRA_EINVAL, "B=%d C=%d L=%d: positive sizes, B <= 65535, fewer than 2^31 elements", B, C, L);
    HIP_TRY(launch_tokens_to_nchw_add(tokens, add, out, B, C, L, (hipStream_t)stream));
    return DANRA_OK;
}

int danra_pointwise_train_supported(int kind, int B, int Cin, int Cout, int H, int W) {
    return pointwise_train_supported(kind, B, Cin, Cout, H, W) ? 1 : 0;
}

size_t danra_pointwise_train_workspace(int kind, int B, int Cin, int Cout, int H, int W) {
    return pointwise_train_supported(kind, B, Cin, Cout, H, W)
               ? pointwise_train_workspace_floats(kind, B, Cin, Cout, H, W) * sizeof(float) : 0;
}

static int pointwise_train_check(const void* a, const void* b, const void* c, int kind, int B, int Cin, int Cout, int H, int W) {
    if (!a || !b || !c) return fail(DANRA_EINVAL, "null argument");
    if (!pointwise_train_supported(kind, B, Cin, Cout, H, W))
        return fail(DANRA_EINVAL, "kind=%d B=%d Cin=%d Cout=%d H=%d W=%d: not a layer the pointwise training kernels take "
                    "(danra_pointwise_train_supported)", kind, B, Cin, Cout, H, W);
    return DANRA_OK;
}

int danra_pointwise_train_fwd(int kind, const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout,
                              int H, int W, void* stream) {
    RC_TRY(pointwise_train_check(x, w, y, kind, B, Cin, Cout, H, W));
    HIP_TRY(launch_pointwise_train_fwd(kind, x, w, bias, y, B, Cin, Cout, H, W, (hipStream_t)stream));
    return DANRA_OK;
}

int danra_pointwise_train_dgrad(int kind, const float* dy, const float* w, float* dx, int B, int Cin, int Cout, int H, int W,
                                void* workspace, void* stream) {
    RC_TRY(pointwise_train_check(dy, w, dx, kind, B, Cin, Cout, H, W));
    if (!workspace) return fail(DANRA_EINVAL, "null workspace");
    HIP_TRY(launch_pointwise_train_dgrad(kind, dy, w, dx, B, Cin, Cout, H, W, static_cast<float*>(workspace), (hipStream_t)stream));
    return DANRA_OK;
}

int danra_pointwise_train_wgrad(int kind, const float* x, const float* dy, float* dw, int B, int Cin, int Cout, int H, int W,
                                void* workspace, void* stream) {
    RC_TRY(pointwise_train_check(x, dy, dw, kind, B, Cin, Cout, H, W));
    if (!workspace) return fail(DANRA_EINVAL, "null workspace");
    HIP_TRY(launch_pointwise_train_wgrad(kind, x, dy, dw, B, Cin, Cout, H, W, static_cast<float*>(workspace), (hipStream_t)stream));
    return DANRA_OK;
}

int danra_sample(danra_handle_t h, float* x, const int64_t* y, const float* cond, const float* lsm, const float* topo,
                 int B, int H, int W, int T, const float* betas, const float* alphas, const float* alpha_hat,
                 int noise_mode, const float* noise, uint64_t seed, const int* trace_steps, int n_trace, float* trace,
                 void* stream) {
    return danra_sample_guided(h, x, y, cond, lsm, topo, B, H, W, T, betas, alphas, alpha_hat, noise_mode, noise, seed,
                               0.0f, 1.0f, trace_steps, n_trace, trace, stream);
}

int danra_sample_guided(danra_handle_t h, float* x, const int64_t* y, const float* cond, const float* lsm,
                        const float* topo, int B, int H, int W, int T, const float* betas, const float* alphas,
                        const float* alpha_hat, int noise_mode, const float* noise, uint64_t seed, float cfg_scale,
                        float noise_scale, const int* trace_steps, int n_trace, float* trace, void* stream) {
    RC_TRY(check_ready(h, B, H, W));
    // The unconditional evaluation of a guided step passes no label and no conditioning plane
    // (DDPM_clean_application/src/diffusion_modules.py:158), so the network must not expect any plane.
    const bool guided = cfg_scale > 0.0f;
    if (guided && h->in_channels != 1)
        return fail(DANRA_EINVAL, "classifier-free guidance evaluates the network without conditioning planes, "
                                  "but conv1 of this network expects %d input channels", h->in_channels);
    if (!x || !betas || !alphas || !alpha_hat || T < 2) return fail(DANRA_EINVAL, "bad sampling arguments");
    if (h->cfg.out_channels != 1) return fail(DANRA_EINVAL, "sampling needs out_channels == 1 (eps has x's shape)");
    if (noise_mode == DANRA_NOISE_HOST && T > 2 && !noise) return fail(DANRA_EINVAL, "host-noise mode needs noise_dev");
    if (h->cfg.has_lsm && !lsm) return fail(DANRA_EINVAL, "lsm_cond is required by this model");
    if (h->cfg.has_topo && !topo) return fail(DANRA_EINVAL, "topo_cond is required by this model");
    if (h->cfg.has_cond && !cond) return fail(DANRA_EINVAL, "cond_img is required by this model");
    if (T - 1 >= kMaxTimesteps) return fail(DANRA_EINVAL, "T=%d exceeds the %d-step coefficient table", T, kMaxTimesteps);
    RC_TRY(ensure_workspace(h, B, H, W));
    hipStream_t st = (hipStream_t)stream;
    Workspace& ws = h->ws;
    const size_t n = (size_t)B * H * W;
    // Per-timestep coefficients: fp32 scalar arithmetic in the order of diffusion_DANRA_conditional.py:135-157
    h->h_coef.assign((size_t)T, make_float4(0.f, 0.f, 0.f, 0.f));
    for (int i = 1; i < T; ++i) {
        const float alpha = alphas[i], beta = betas[i], ah = alpha_hat[i];
        const float one_minus_alpha = 1.0f - alpha, one_minus_ah = 1.0f - ah;
        h->h_coef[i] = make_float4(1.0f / sqrtf(alpha), one_minus_alpha / sqrtf(one_minus_ah), sqrtf(beta), 0.f);
    }
    HIP_TRY(hipMemcpyAsync(ws.coef, h->h_coef.data(), (size_t)T * sizeof(float4), hipMemcpyHostToDevice, st));
    // private copies of the inputs: the step (eager or captured) only ever touches workspace memory
    const bool has_y = (h->cfg.num_classes > 0) && y;
    HIP_TRY(hipMemcpyAsync(ws.xs, x, n * 4, hipMemcpyDeviceToDevice, st));
    if (has_y) HIP_TRY(hipMemcpyAsync(ws.ys, y, (size_t)B * 8, hipMemcpyDeviceToDevice, st));
    if (h->cfg.has_cond) HIP_TRY(hipMemcpyAsync(ws.conds, cond, n * 4, hipMemcpyDeviceToDevice, st));
    if (h->cfg.has_lsm) HIP_TRY(hipMemcpyAsync(ws.lsms, lsm, n * 4, hipMemcpyDeviceToDevice, st));
    if (h->cfg.has_topo) HIP_TRY(hipMemcpyAsync(ws.topos, topo, n * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(launch_init_state(ws.state, T - 1, noise_mode == DANRA_NOISE_PHILOX ? 1 : 0, seed,
                              noise_mode == DANRA_NOISE_HOST ? noise : nullptr, cfg_scale, noise_scale,
                              (unsigned long long)h->first_image * ((unsigned long long)H * W / 4), st));
    RC_TRY(upload_freq(h, st));

    // the time projections of every (timestep, class) pair, tabulated once: [T][classes][width]; a guided
    // loop with labels also tabulates the label-free rows [T][width] for its unconditional evaluations
    // (a separate allocation, so that no pointer a captured step holds depends on T)
    const int classes = has_y ? h->cfg.num_classes : 1;
    const bool two_tables = guided && has_y;
    const size_t rows_cond = (size_t)T * classes;
    if (two_tables && (size_t)T > h->tb_u_cap) {
        HIP_TRY(hipStreamSynchronize(st));
        if (h->tb_table_u) (void)hipFree(h->tb_table_u);
        h->tb_table_u = nullptr; h->tb_u_cap = 0;
        drop_step_graph(ws);
        if (hipMalloc((void**)&h->tb_table_u, (size_t)T * h->tb_width * sizeof(float)) != hipSuccess)
            return fail(DANRA_ENOMEM, "label-free time-projection table of %d rows: allocation failed", T);
        h->tb_u_cap = (size_t)T;
    }
    {
        const size_t rows = rows_cond + (two_tables ? (size_t)T : 0);
        if (rows > h->tb_rows_cap) {
            HIP_TRY(hipStreamSynchronize(st));
            if (h->tb_table) (void)hipFree(h->tb_table);
            if (h->tb_index) (void)hipFree(h->tb_index);
            h->tb_table = nullptr; h->tb_index = nullptr; h->tb_rows_cap = 0;
            drop_step_graph(ws);  // the captured step refers to the table
            if (hipMalloc((void**)&h->tb_table, rows * h->tb_width * sizeof(float)) != hipSuccess ||
                hipMalloc((void**)&h->tb_index, 2 * rows * sizeof(int64_t)) != hipSuccess)
                return fail(DANRA_ENOMEM, "time-projection table of %zu rows: allocation failed", rows);
            h->tb_rows_cap = rows;
        }
        h->h_tb_index.resize(2 * rows);
        for (size_t r = 0; r < rows_cond; ++r) {
            h->h_tb_index[r] = (int64_t)(r / classes);
            h->h_tb_index[rows + r] = (int64_t)(r % classes);
        }
        for (size_t r = rows_cond; r < rows; ++r) {
            h->h_tb_index[r] = (int64_t)(r - rows_cond);
            h->h_tb_index[rows + r] = 0;
        }
        HIP_TRY(hipMemcpyAsync(h->tb_index, h->h_tb_index.data(), 2 * rows * sizeof(int64_t), hipMemcpyHostToDevice, st));
        RC_TRY(run_time_mlp(h, h->tb_index, nullptr, has_y ? h->tb_index + rows : nullptr, (int)rows_cond, nullptr, nullptr,
                            h->tb_table, st));
        if (two_tables)
            RC_TRY(run_time_mlp(h, h->tb_index + rows_cond, nullptr, nullptr, T, nullptr, nullptr, h->tb_table_u, st));
    }
    const float* tb_uncond = two_tables ? h->tb_table_u : h->tb_table;
    // the conditioning planes' share of the stem convolution is the same at every step
    if (h->stem.Cin > 1) {
        const float* planes[4] = {nullptr, nullptr, nullptr, nullptr};
        int np = 0;
        planes[np++] = ws.xs;
        if (h->cfg.has_lsm) planes[np++] = ws.lsms;
        if (h->cfg.has_topo) planes[np++] = ws.topos;
        if (h->cfg.has_cond) planes[np++] = ws.conds;
        RC_TRY(run_stem_planes(h, planes, 1, np - 1, ws.stem_static, B, H, W, nullptr, nullptr, st));
    }
    // One evaluation per step: the update kernel gathers the next step's time projections (the first step's are
    // gathered here).  A guided step gathers twice, from different tables, at the head of each evaluation.
    const NextGather next{h->tb_table, has_y ? ws.ys : nullptr, ws.tbias, classes, h->tb_width, B};
    if (!guided) HIP_TRY(launch_tbias_gather(ws.state, h->tb_table, next.y, ws.tbias, classes, h->tb_width, B, st));
    auto enqueue_step = [&](hipStream_t s) -> int {
        RC_TRY(forward_impl(h, ws.xs, nullptr, ws.state, has_y ? ws.ys : nullptr, ws.conds, ws.lsms, ws.topos, ws.eps, B,
                            H, W, s, true, h->tb_table, !guided));
        if (guided)
            RC_TRY(forward_impl(h, ws.xs, nullptr, ws.state, nullptr, nullptr, nullptr, nullptr, ws.eps_uncond, B, H, W, s,
                                true, tb_uncond));
        HIP_TRY(launch_ddpm_update_state(ws.xs, ws.eps, guided ? ws.eps_uncond : nullptr, ws.coef, ws.state, n, s,
                                         guided ? nullptr : &next));
        return DANRA_OK;
    };

    // While a kernel family is being timed, every kProfStride-th step is enqueued kernel by kernel with events
    // attached to that family's launches; all other steps run as usual (a step enqueued that way takes 85 us
    // longer than its graph replay: at every 8th step that cost 3 % of the loop that is being measured, at every
    // 64th 0.1 %).
    constexpr int kProfStride = 64;
    const bool use_graph = h->graph_mode != 0;
    const int step_mode = (has_y ? 1 : 0) | (guided ? 2 : 0) | (igemm_fp32_products_forced() ? 4 : 0);
    if (use_graph && (!ws.step_exec || ws.step_exec_mode != step_mode)) {
        h->prof_now = false;
        drop_step_graph(ws);
        if (!h->cap_stream) HIP_TRY(hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking));
        HIP_TRY(hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeRelaxed));
        const int rc = enqueue_step(h->cap_stream);
        const hipError_t ce = hipStreamEndCapture(h->cap_stream, &ws.step_graph);
        if (rc != DANRA_OK) { drop_step_graph(ws); return rc; }
        if (ce != hipSuccess) {
            drop_step_graph(ws);
            return fail(DANRA_EHIP, "capturing the denoise step failed: %s", hipGetErrorString(ce));
        }
        const hipError_t ie = hipGraphInstantiate(&ws.step_exec, ws.step_graph, nullptr, nullptr, 0);
        if (ie != hipSuccess) {
            drop_step_graph(ws);
            return fail(DANRA_EHIP, "instantiating the denoise-step graph failed: %s", hipGetErrorString(ie));
        }
        ws.step_exec_mode = step_mode;
    }
    for (int i = T - 1; i >= 1; --i) {
        h->prof_now = h->prof_kind != 0 && (i % kProfStride == 0);
        if (use_graph && !h->prof_now)
            HIP_TRY(hipGraphLaunch(ws.step_exec, st));
        else
            RC_TRY(enqueue_step(st));
        for (int k = 0; k < n_trace; ++k)
            if (trace && trace_steps && trace_steps[k] == i)
                HIP_TRY(hipMemcpyAsync(trace + (size_t)k * n, ws.xs, n * 4, hipMemcpyDeviceToDevice, st));
    }
    h->prof_now = true;
    HIP_TRY(hipMemcpyAsync(x, ws.xs, n * 4, hipMemcpyDeviceToDevice, st));
    return DANRA_OK;
}

int danra_set_first_image(danra_handle_t h, uint64_t first_image) {
    if (!h) return fail(DANRA_EINVAL, "null handle");
    if (first_image >= (1ull << 40)) return fail(DANRA_EINVAL, "first_image must stay below 2^40");
    h->first_image = first_image;
    return DANRA_OK;
}

int danra_set_graph_mode(danra_handle_t h, int enable) {
    if (!h) return fail(DANRA_EINVAL, "null handle");
    h->graph_mode = enable ? 1 : 0;
    return DANRA_OK;
}

int danra_profile_begin(danra_handle_t h, int kind) {
    if (!h) return fail(DANRA_EINVAL, "null handle");
    if (kind != DANRA_PROF_FLASH_ATTENTION && kind != DANRA_PROF_IGEMM) return fail(DANRA_EINVAL, "unknown kernel family %d", kind);
    h->prof_kind = kind;
    h->prof_used = 0;
    h->prof_flops = 0.0;
    h->prof_exec_flops = 0.0;
    h->prof_split_flops = 0.0;
    h->prof_bytes = 0.0;
    return DANRA_OK;
}

int danra_profile_end(danra_handle_t h, double* total_ms, long long* launches, double* total_flops) {
    if (!h) return fail(DANRA_EINVAL, "null handle");
    double ms = 0.0;
    const size_t pairs = h->prof_used / 2;
    if (pairs) HIP_TRY(hipEventSynchronize(h->prof_events[2 * pairs - 1]));
    for (size_t i = 0; i < pairs; ++i) {
        float one = 0.f;
        HIP_TRY(hipEventElapsedTime(&one, h->prof_events[2 * i], h->prof_events[2 * i + 1]));
        ms += one;
    }
    if (total_ms) *total_ms = ms;
    if (launches) *launches = (long long)pairs;
    if (total_flops) *total_flops = h->prof_flops;
    h->prof_kind = 0;
    h->prof_used = 0;
    return DANRA_OK;
}

int danra_profile_executed(danra_handle_t h, double* executed_flops, double* split_flops) {
    if (!h) return fail(DANRA_EINVAL, "null handle");
    if (executed_flops) *executed_flops = h->prof_exec_flops;
    if (split_flops) *split_flops = h->prof_split_flops;
    return DANRA_OK;
}

int danra_profile_bytes(danra_handle_t h, double* algorithmic_bytes) {
    if (!h) return fail(DANRA_EINVAL, "null handle");
    if (algorithmic_bytes) *algorithmic_bytes = h->prof_bytes;
    return DANRA_OK;
}

// ---------------------------------------------------------------- per-op entry points
int danra_tbias_width(danra_handle_t h) { return h ? h->tb_width : 0; }

int danra_time_embed(danra_handle_t h, const int64_t* t, const int64_t* y, int B, float* enc, float* dec, float* tbias,
                     void* stream) {
    if (!h || !h->packed) return fail(DANRA_ESTATE, "weights not packed");
    return run_time_mlp(h, t, nullptr, y, B, enc, dec, tbias, (hipStream_t)stream);
}

// The per-op entry points size the workspace for the smallest image whose buffers hold the op.
static int op_workspace(danra_handle_t h, int B, long tokens64) {
    // tokens64 = number of 64-channel NHWC pixels per image the op needs at most
    int side = 32;
    while ((long)(side / 2) * (side / 2) < tokens64) side *= 2;
    if (h->ws.base && h->ws.B == B && (long)(h->ws.H / 2) * (h->ws.W / 2) >= tokens64) return DANRA_OK;
    return ensure_workspace(h, B, side, side);
}

// argument checks shared by the per-op entry points: a bad call is refused, never launched
static int op_args(const void* in, const void* out, int B, long extent) {
    if (!in || !out) return fail(DANRA_EINVAL, "null tensor pointer");
    if (B <= 0 || extent <= 0) return fail(DANRA_EINVAL, "B=%d and the spatial extent %ld must be positive", B, extent);
    if ((unsigned long long)B * (unsigned long long)extent >= (1ull << 24))
        return fail(DANRA_EINVAL, "B=%d x %ld pixels: must stay below 2^24 per call", B, extent);
    return DANRA_OK;
}

int danra_attention(danra_handle_t h, int index, const float* x, float* out, int B, int HW, int relu, void* stream) {
    if (!h || !h->packed) return fail(DANRA_ESTATE, "weights not packed");
    if (index < 0 || index > 8) return fail(DANRA_EINVAL, "attention index out of range");
    RC_TRY(op_args(x, out, B, HW));
    const int C = h->attn[index].C;
    RC_TRY(op_workspace(h, B, ((long)HW * C + 63) / 64));
    return run_attention(h, index, x, out, B, HW, relu, (hipStream_t)stream);
}

int danra_conv1(danra_handle_t h, const float* x, const float* lsm, const float* topo, const float* cond, float* out,
                int B, int H, int W, void* stream) {
    if (!h || !h->packed) return fail(DANRA_ESTATE, "weights not packed");
    const float* planes[4] = {nullptr, nullptr, nullptr, nullptr};
    int np = 0;
    planes[np++] = x;
    if (h->cfg.has_lsm) planes[np++] = lsm;
    if (h->cfg.has_topo) planes[np++] = topo;
    if (h->cfg.has_cond) planes[np++] = cond;
    for (int i = 0; i < np; ++i)
        if (!planes[i]) return fail(DANRA_EINVAL, "missing input plane %d", i);
    RC_TRY(op_args(x, out, B, (long)H * W));
    if (H % 16 || W % 16) return fail(DANRA_EINVAL, "H=%d W=%d must be multiples of 16", H, W);
    RC_TRY(op_workspace(h, B, (long)(H / 2) * (W / 2)));
    return run_stem(h, planes, out, B, H, W, nullptr, false, (hipStream_t)stream);
}

int danra_conv2_bn_relu(danra_handle_t h, const float* x, float* out, int B, int H, int W, void* stream) {
    if (!h || !h->packed) return fail(DANRA_ESTATE, "weights not packed");
    RC_TRY(op_args(x, out, B, (long)H * W));
    RC_TRY(op_workspace(h, B, 1));
    return run_conv(h, h->conv2, x, out, B, H, W, nullptr, 1, nullptr, nullptr, nullptr, (hipStream_t)stream);
}

int danra_resnet_layer(danra_handle_t h, int level, const float* x, float* out, int B, int H, int W, void* stream) {
    if (!h || !h->packed) return fail(DANRA_ESTATE, "weights not packed");
    if (level < 1 || level > 4) return fail(DANRA_EINVAL, "level must be 1..4");
    RC_TRY(op_args(x, out, B, (long)H * W));
    const int cin = level == 1 ? 64 : (64 << (level - 2));
    RC_TRY(op_workspace(h, B, 4 * (((long)H * W * cin + 63) / 64)));
    return run_layer(h, level, x, out, B, H, W, nullptr, (hipStream_t)stream);
}

int danra_decoder_block(danra_handle_t h, int block, const float* fmap, const float* skip, const int64_t* t, float* out,
                        int B, int H, int W, void* stream) {
    if (!h || !h->packed) return fail(DANRA_ESTATE, "weights not packed");
    if (block < 0 || block > 4) return fail(DANRA_EINVAL, "block must be 0..4");
    RC_TRY(op_args(fmap, out, B, 4L * H * W));
    if (block < 4 && !skip) return fail(DANRA_EINVAL, "skip_nhwc is required by blocks 0..3");
    hipStream_t st = (hipStream_t)stream;
    if (block == 4) {
        RC_TRY(op_workspace(h, B, (long)H * W));
        if ((long)h->ws.H * h->ws.W < 4L * H * W) RC_TRY(ensure_workspace(h, B, 2 * H, 2 * W));
        return run_final_block(h, fmap, out, B, H, W, st);
    }
    const int cin = h->dec[block].Cin;
    RC_TRY(op_workspace(h, B, 4 * (((long)H * W * cin + 63) / 64)));
    if (!t) return fail(DANRA_EINVAL, "t is required");
    RC_TRY(run_time_mlp(h, t, nullptr, nullptr, B, nullptr, nullptr, h->ws.tbias, st));
    return run_decoder_block(h, block, fmap, skip, h->ws.tbias, out, B, H, W, st);
}

// Scratch of the InstanceNorm pair (statistics kernel + apply kernel): mean[B*C], rstd[B*C], then the sliced partial
// statistics.  The statistics launcher slices an image over S0 = smallest power of two with HW/S0 <= 1024 workgroups
// (B*S0*2*C floats) and may slice small problems further, which never needs more than 16384 floats.
static size_t in_norm_scratch_floats(int B, int HW, int C) {
    size_t S0 = 1;
    while ((size_t)HW / S0 > 1024 || (size_t)HW % S0) {
        S0 *= 2;
        if (S0 > 4096) return 0;  // no equal slices of at most 1024 pixels: the launcher refuses such a map
    }
    return (size_t)2 * B * C + 16384 + (S0 > 1 ? (size_t)2 * B * C * S0 : 0);
}

static hipError_t instance_norm_launches(const float* x, float* out, int B, int HW, int C, float* scratch_dev,
                                         size_t scratch_floats, hipStream_t st) {
    float* mean = scratch_dev;
    float* rstd = scratch_dev + (size_t)B * C;
    float* part = scratch_dev + (size_t)2 * B * C;
    hipError_t e = launch_in_stats(x, mean, rstd, part, scratch_floats - (size_t)2 * B * C, B, HW, C, st);
    if (e == hipSuccess) e = launch_in_apply(x, mean, rstd, nullptr, nullptr, 0, out, B, HW, C, st);
    return e;
}

int danra_instance_norm(const float* x, float* out, int B, int HW, int C, void* stream) {
    if (C <= 0 || C % 32) return fail(DANRA_EINVAL, "C must be a positive multiple of 32");
    RC_TRY(op_args(x, out, B, HW));
    const size_t need = in_norm_scratch_floats(B, HW, C);
    if (!need) return fail(DANRA_EINVAL, "HW=%d cannot be cut into equal slices of at most 1024 pixels", HW);
    float* stats = nullptr;
    HIP_TRY(hipMalloc((void**)&stats, need * sizeof(float)));
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = instance_norm_launches(x, out, B, HW, C, stats, need, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(stats);
    HIP_TRY(e);
    return DANRA_OK;
}

// ---- the same operator for callers that cannot allocate: its scratch lives in the code object itself.
// `scratch` below is the HOST-side handle of a device array: every GPU the library runs on holds its own copy at its
// own address, which hipGetSymbolAddress resolves once per device; host code uses that address, never the symbol.
constexpr size_t kNoAllocScratchFloats = (size_t)1 << 20;  // 4 MiB per GPU, held until the process ends
static __device__ __align__(256) float scratch[kNoAllocScratchFloats];
constexpr int kMaxDevices = 64;
static float* g_scratch_dev[kMaxDevices] = {};
static std::mutex g_scratch_mutex;

// The address of `scratch` on the current device (resolved on the first call per device).
static int resolve_scratch(float** out) {
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return fail(DANRA_EINVAL, "device ordinal %d is outside 0..%d", dev, kMaxDevices - 1);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_dev[dev]) {
        float* p = nullptr;
        HIP_TRY(hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch)));
        if (!p) return fail(DANRA_EHIP, "hipGetSymbolAddress returned a null address for the scratch array");
        g_scratch_dev[dev] = p;
    }
    *out = g_scratch_dev[dev];
    return DANRA_OK;
}

int danra_instance_norm_noalloc(const float* x, float* out, int B, int HW, int C, void* stream) {
    if (C <= 0 || C % 32) return fail(DANRA_EINVAL, "C must be a positive multiple of 32");
    RC_TRY(op_args(x, out, B, HW));
    const size_t need = in_norm_scratch_floats(B, HW, C);
    if (!need) return fail(DANRA_EINVAL, "HW=%d cannot be cut into equal slices of at most 1024 pixels", HW);
    if (need > kNoAllocScratchFloats)
        return fail(DANRA_EINVAL, "B=%d HW=%d C=%d needs %zu floats of scratch, the built-in array holds %zu: use "
                    "danra_instance_norm or a smaller batch", B, HW, C, need, (size_t)kNoAllocScratchFloats);
    float* dev_scratch = nullptr;
    RC_TRY(resolve_scratch(&dev_scratch));
    // every word of the scratch that the apply kernel or the merge reads is written by the statistics kernel of this
    // very call first, so nothing needs clearing between calls
    HIP_TRY(instance_norm_launches(x, out, B, HW, C, dev_scratch, kNoAllocScratchFloats, (hipStream_t)stream));
    return DANRA_OK;
}

// A 3x3 stride-1 convolution through the window-staged kernel, for unit tests of that kernel on shapes and value
// ranges of the caller's choosing: packs the weights (scale from their largest magnitude, fragment order) into a
// temporary buffer, runs the kernel and waits for it.
// The per-op entries of the window kernels: scale and pack the weights into a temporary buffer, run, wait.
static int conv_window_op(int taps, const float* x, const float* w, const float* bias, const float* res, const float* nmean,
                          const float* nrstd, float* out, int B, int H, int W, int Cin, int Cout, int relu, hipStream_t st,
                          int stride = 1, float* out2 = nullptr, int n_split = 0) {
    const size_t n = (size_t)Cout * taps * Cin;
    float* tmp = nullptr;
    if (hipMalloc((void**)&tmp, (n + 64) * sizeof(float)) != hipSuccess) return fail(DANRA_ENOMEM, "temporary weight buffer");
    auto done = [&](int rc) { (void)hipFree(tmp); return rc; };
    float mx = 0.f;
    if (launch_maxabs(w, n, tmp + n, st) != hipSuccess || hipMemcpyAsync(&mx, tmp + n, sizeof(float), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        return done(fail(DANRA_EHIP, "weight maximum: %s", hipGetErrorString(hipGetLastError())));
    if (!(mx > 0.f) || !(mx <= 3.0e38f)) return done(fail(DANRA_EINVAL, "the weights are all zero or not finite: no split form"));
    int e = 0;
    (void)frexpf(mx, &e);
    const int sft = 13 - e;   // 2^sft * mx in [2^12, 2^13), as the packed model's matrices are scaled
    if (launch_pack_conv_frag(w, tmp, Cout, Cin, ldexpf(1.0f, sft), st, taps) != hipSuccess) return done(fail(DANRA_EHIP, "weight packing failed"));
    ConvWinArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.wf = tmp; a.w32 = w; a.wscale = ldexpf(1.0f, -sft); a.bias = bias; a.res = res; a.nmean = nmean; a.nrstd = nrstd; a.out = out;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.act = relu ? ACT_RELU : ACT_NONE;
    a.stride = stride; a.out2 = out2; a.n_split = n_split;
    const hipError_t le = taps == 9 ? launch_conv_win(a, st) : launch_conv8s2(a, st);
    const hipError_t se = hipStreamSynchronize(st);
    if (le != hipSuccess || se != hipSuccess) return done(fail(DANRA_EHIP, "window convolution: %s", hipGetErrorString(le != hipSuccess ? le : se)));
    return done(DANRA_OK);
}

int danra_conv3x3(const float* x, const float* w, const float* bias, const float* res, const float* nmean, const float* nrstd,
                  float* out, int B, int H, int W, int Cin, int Cout, int relu, void* stream) {
    if (!x || !w || !out) return fail(DANRA_EINVAL, "null argument");
    if ((nmean == nullptr) != (nrstd == nullptr)) return fail(DANRA_EINVAL, "nmean and nrstd come together");
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return fail(DANRA_EINVAL, "sizes must be positive");
    if (!conv_win_supported(B, H, W, Cin, Cout, nmean != nullptr))
        return fail(DANRA_EINVAL, "B=%d %dx%d %d->%d%s: not a shape of the window-staged kernel (sides powers of two up to 64 "
                    "wide, Cin %% 64 == 0, Cout %% 32 == 0)", B, H, W, Cin, Cout, nmean ? " with norm on load" : "");
    return conv_window_op(9, x, w, bias, res, nmean, nrstd, out, B, H, W, Cin, Cout, relu, (hipStream_t)stream);
}

int danra_conv3x3s2(const float* x, const float* w, const float* bias, float* out, float* out2, int B, int H, int W, int Cin,
                    int Cout, int n_split, int relu, void* stream) {
    if (!x || !w || !out) return fail(DANRA_EINVAL, "null argument");
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return fail(DANRA_EINVAL, "sizes must be positive");
    if (out2 && (n_split <= 0 || n_split % 32 || n_split >= Cout)) return fail(DANRA_EINVAL, "n_split=%d: a multiple of 32 inside 0..Cout", n_split);
    if (!conv_win_supported(B, H, W, Cin, Cout, false, 2))
        return fail(DANRA_EINVAL, "B=%d %dx%d %d->%d: not a shape of the window-staged kernel at stride 2 (sides powers of two, an "
                    "output map of at most 64 pixels, Cin %% 64 == 0, Cout %% 32 == 0)", B, H, W, Cin, Cout);
    return conv_window_op(9, x, w, bias, nullptr, nullptr, nullptr, out, B, H, W, Cin, Cout, relu, (hipStream_t)stream, 2, out2, n_split);
}

int danra_conv8x8s2(const float* x, const float* w, const float* bias, const float* res, float* out, int B, int H, int W, int Cin,
                    int Cout, int relu, void* stream) {
    if (!x || !w || !out) return fail(DANRA_EINVAL, "null argument");
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return fail(DANRA_EINVAL, "sizes must be positive");
    if (!conv8s2_supported(B, H, W, Cin, Cout))
        return fail(DANRA_EINVAL, "B=%d %dx%d %d->%d: not a shape of the strided window kernel (sides powers of two from 16 on, "
                    "Cin %% 64 == 0, Cout %% 64 == 0)", B, H, W, Cin, Cout);
    return conv_window_op(64, x, w, bias, res, nullptr, nullptr, out, B, H, W, Cin, Cout, relu, (hipStream_t)stream);
}

int danra_nchw_to_nhwc(const float* src, float* dst, int B, int C, int HW, void* stream) {
    RC_TRY(op_args(src, dst, B, HW));
    if (C <= 0) return fail(DANRA_EINVAL, "C must be positive");
    HIP_TRY(launch_nchw_to_nhwc(src, dst, B, C, HW, (hipStream_t)stream));
    return DANRA_OK;
}
int danra_nhwc_to_nchw(const float* src, float* dst, int B, int C, int HW, void* stream) {
    RC_TRY(op_args(src, dst, B, HW));
    if (C <= 0) return fail(DANRA_EINVAL, "C must be positive");
    HIP_TRY(launch_nhwc_to_nchw(src, dst, B, C, HW, (hipStream_t)stream));
    return DANRA_OK;
}

// ---- elucidated (EDM) sampler steps
static int edm_noise_args(const float* noise, int noise_mode) {
    if (noise_mode != DANRA_NOISE_HOST && noise_mode != DANRA_NOISE_PHILOX) return fail(DANRA_EINVAL, "unknown noise mode");
    if (noise_mode == DANRA_NOISE_HOST && !noise) return fail(DANRA_EINVAL, "noise_dev is null in DANRA_NOISE_HOST mode");
    return DANRA_OK;
}
static int edm_image_args(int B, size_t per_image) {
    if (B <= 0 || per_image == 0) return fail(DANRA_EINVAL, "B=%d per_image=%zu: both must be positive", B, per_image);
    if (per_image >= ((size_t)1 << 32)) return fail(DANRA_EINVAL, "per_image=%zu: at most 2^32 - 1 values per image", per_image);
    return DANRA_OK;
}

int danra_edm_perturb(const float* images, const float* noise, const float* inpaint, const unsigned char* mask,
                      float* images_hat, float* net_in, int B, int C, int HW, float s_noise, float noise_scale, float c_in,
                      int noise_mode, uint64_t seed, uint64_t offset, void* stream) {
    if (!images || !images_hat || !net_in) return fail(DANRA_EINVAL, "null argument");
    if (C <= 0 || HW <= 0) return fail(DANRA_EINVAL, "C=%d HW=%d: both must be positive", C, HW);
    RC_TRY(edm_image_args(B, (size_t)C * HW));
    RC_TRY(edm_noise_args(noise, noise_mode));
    if ((inpaint == nullptr) != (mask == nullptr)) return fail(DANRA_EINVAL, "inpaint images and masks come together");
    HIP_TRY(launch_edm_perturb(images, noise, inpaint, mask, images_hat, net_in, B, C, HW, s_noise, noise_scale, c_in,
                               noise_mode == DANRA_NOISE_PHILOX ? 1 : 0, seed, offset, (hipStream_t)stream));
    return DANRA_OK;
}

int danra_edm_precondition(const float* x, const float* net_out, float* model_out, size_t n, float c_skip, float c_out,
                           int clamp, void* stream) {
    if (!x || !net_out || !model_out) return fail(DANRA_EINVAL, "null argument");
    if (n == 0) return fail(DANRA_EINVAL, "n must be positive");
    HIP_TRY(launch_edm_precondition(x, net_out, model_out, n, c_skip, c_out, clamp ? 1 : 0, (hipStream_t)stream));
    return DANRA_OK;
}

size_t danra_edm_dynamic_threshold_workspace(int B, size_t per_image) {
    if (B <= 0 || per_image == 0) return 0;
    return edm_quantile_workspace_bytes(B, per_image);
}

int danra_edm_dynamic_threshold(const float* x, float* s_out, int B, size_t per_image, float percentile, void* workspace,
                                void* stream) {
    if (!x || !s_out) return fail(DANRA_EINVAL, "null argument");
    RC_TRY(edm_image_args(B, per_image));
    if (!(percentile >= 0.f && percentile <= 1.f)) return fail(DANRA_EINVAL, "percentile %g is outside [0, 1]", (double)percentile);
    if (!workspace && edm_quantile_workspace_bytes(B, per_image))
        return fail(DANRA_EINVAL, "images of %zu values need a workspace of danra_edm_dynamic_threshold_workspace() bytes", per_image);
    HIP_TRY(launch_edm_quantile_abs(x, s_out, B, per_image, percentile, 1.0f, workspace, (hipStream_t)stream));
    return DANRA_OK;
}

int danra_edm_euler(const float* images_hat, const float* net_out, float* model_out, const float* thresholds, float* d_out,
                    float* images_next, float* net_in_next, int B, size_t per_image, float c_skip, float c_out, int clamp,
                    float sigma_hat, float dt, float c_in_next, void* stream) {
    if (!images_hat || !model_out || !images_next) return fail(DANRA_EINVAL, "null argument");
    RC_TRY(edm_image_args(B, per_image));
    if (sigma_hat == 0.f) return fail(DANRA_EINVAL, "sigma_hat is zero");
    HIP_TRY(launch_edm_euler(images_hat, net_out, model_out, thresholds, d_out, images_next, net_in_next, B, per_image, c_skip,
                             c_out, clamp ? 1 : 0, sigma_hat, dt, c_in_next, (hipStream_t)stream));
    return DANRA_OK;
}

int danra_edm_heun(const float* images_hat, const float* images_next, const float* net_out, float* model_out,
                   const float* thresholds, const float* d, float* images_out, int B, size_t per_image, float c_skip,
                   float c_out, int clamp, float sigma_next, float half_dt, void* stream) {
    if (!images_hat || !images_next || !model_out || !d || !images_out) return fail(DANRA_EINVAL, "null argument");
    RC_TRY(edm_image_args(B, per_image));
    if (sigma_next == 0.f) return fail(DANRA_EINVAL, "sigma_next is zero: the last step has no second-order correction");
    HIP_TRY(launch_edm_heun(images_hat, images_next, net_out, model_out, thresholds, d, images_out, B, per_image, c_skip,
                            c_out, clamp ? 1 : 0, sigma_next, half_dt, (hipStream_t)stream));
    return DANRA_OK;
}

int danra_edm_renoise(float* images, const float* noise, size_t n, float scale, int noise_mode, uint64_t seed,
                      uint64_t offset, void* stream) {
    if (!images) return fail(DANRA_EINVAL, "null argument");
    if (n == 0) return fail(DANRA_EINVAL, "n must be positive");
    RC_TRY(edm_noise_args(noise, noise_mode));
    HIP_TRY(launch_edm_renoise(images, noise, n, scale, noise_mode == DANRA_NOISE_PHILOX ? 1 : 0, seed, offset,
                               (hipStream_t)stream));
    return DANRA_OK;
}

int danra_edm_finish(float* images, const float* inpaint, const unsigned char* mask, int B, int C, int HW, int unnormalize,
                     void* stream) {
    if (!images) return fail(DANRA_EINVAL, "null argument");
    if (C <= 0 || HW <= 0) return fail(DANRA_EINVAL, "C=%d HW=%d: both must be positive", C, HW);
    RC_TRY(edm_image_args(B, (size_t)C * HW));
    if ((inpaint == nullptr) != (mask == nullptr)) return fail(DANRA_EINVAL, "inpaint images and masks come together");
    HIP_TRY(launch_edm_finish(images, inpaint, mask, B, C, HW, unnormalize ? 1 : 0, (hipStream_t)stream));
    return DANRA_OK;
}

}  // extern "C"
