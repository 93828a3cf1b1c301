// The per-step arithmetic of the elucidated (Karras et al., EDM) sampler of the cascaded model, around the network
// evaluations: stochastic churn, input/output preconditioning, static or dynamic thresholding of the predicted image,
// the Euler step and its second-order (Heun) correction, repaint noise, and the final clamp / inpaint merge.
// Reference: CascadedDDPM/elucidated_imagen.py:309-369 (thresholding, preconditioning) and :393-545 (one_unet_sample).
//
// Every kernel is a streaming pass (bound by HBM bandwidth: a few reads and writes of the image tensor per element)
// and evaluates the reference's fp32 statements one rounding at a time, in the reference's order (contraction into
// FMAs is switched off), so that with caller-supplied noise the results are those of the reference's CPU path bit
// for bit.  The per-batch scalars (c_in, c_skip, c_out, step sizes) come from the host, which forms them exactly as
// the reference does.  The exception is the dynamic threshold, a per-image order statistic: one workgroup per image
// finds it with an exact radix select over the bit patterns of |x| (no sort, no approximation).
#include <initializer_list>
#include <type_traits>

#include "danra_kernels.h"
#include "philox_noise.h"

namespace danra {
namespace {

template <int V>
struct Pack {
    float v[V];
};

template <int V>
__device__ __forceinline__ Pack<V> ld(const float* __restrict__ p, size_t u) {
    Pack<V> r;
    if constexpr (V == 4) {
        const float4 t = reinterpret_cast<const float4*>(p)[u];
        r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
    } else {
        r.v[0] = p[u];
    }
    return r;
}

template <int V>
__device__ __forceinline__ void st(float* __restrict__ p, size_t u, const Pack<V>& r) {
    if constexpr (V == 4) {
        reinterpret_cast<float4*>(p)[u] = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
    } else {
        p[u] = r.v[0];
    }
}

// Standard normals of unit u (V consecutive elements).  Element e always takes component e % 4 of the Philox quad
// e / 4, whatever the vector width of the launch.
template <int V>
__device__ __forceinline__ Pack<V> draw(size_t u, uint64_t seed, uint64_t offset) {
    const size_t quad = V == 4 ? u : (u >> 2);
    uint32_t r[4];
    philox4((uint32_t)quad, (uint32_t)(quad >> 32), (uint32_t)offset, (uint32_t)(offset >> 32), (uint32_t)seed,
            (uint32_t)(seed >> 32), r);
    float z[4];
    box_muller(r[0], r[1], z[0], z[1]);
    box_muller(r[2], r[3], z[2], z[3]);
    Pack<V> out;
    if constexpr (V == 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) out.v[j] = z[j];
    } else {
        const int c = (int)(u & 3);
        out.v[0] = c == 0 ? z[0] : c == 1 ? z[1] : c == 2 ? z[2] : z[3];
    }
    return out;
}

// torch.clamp: NaN stays NaN
__device__ __forceinline__ float clamp_keep_nan(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

// c_skip * x + c_out * net, then the static clamp if asked for (elucidated_imagen.py:364, :311)
__device__ __forceinline__ float precondition_one(float x, float net, float c_skip, float c_out, int clamp) {
#pragma clang fp contract(off)
    const float a = c_skip * x;
    const float b = c_out * net;
    const float o = a + b;
    return clamp ? clamp_keep_nan(o, -1.0f, 1.0f) : o;
}

// x.clamp(-s, s) / s (elucidated_imagen.py:321)
__device__ __forceinline__ float threshold_one(float v, float s) { return clamp_keep_nan(v, -s, s) / s; }

struct Precond {
    float c_skip, c_out;
    int clamp;  // static clamp to [-1, 1] inside the kernel (dynamic thresholds come as a per-image array instead)
};

// ---- churn: images_hat = images + k * (S_noise * z), inpainted region replaced, and the network's input c_in * images_hat
template <int V>
__global__ __launch_bounds__(256) void k_edm_perturb(const float* __restrict__ images, const float* __restrict__ noise,
                                                     const float* __restrict__ inpaint,
                                                     const unsigned char* __restrict__ mask, float* __restrict__ images_hat,
                                                     float* __restrict__ net_in, size_t units, unsigned C, unsigned HW,
                                                     float s_noise, float k, float c_in, int philox, uint64_t seed,
                                                     uint64_t offset) {
#pragma clang fp contract(off)
    const size_t per_image = (size_t)C * HW;
    for (size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x; u < units; u += (size_t)gridDim.x * blockDim.x) {
        const Pack<V> x = ld<V>(images, u);
        Pack<V> z;
        if (philox) z = draw<V>(u, seed, offset);
        else z = ld<V>(noise, u);
        Pack<V> h, a;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const float e = s_noise * z.v[j];
            a.v[j] = k * e;
            h.v[j] = x.v[j] + a.v[j];
        }
        if (mask) {
            const size_t e0 = u * V;
            const size_t b = e0 / per_image;
            const size_t pix = (e0 - b * per_image) % HW;
            const Pack<V> inp = ld<V>(inpaint, u);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float m = mask[b * HW + pix + j] ? 1.0f : 0.0f;
                const float keep = h.v[j] * (1.0f - m);
                const float fill = (inp.v[j] + a.v[j]) * m;
                h.v[j] = keep + fill;
            }
        }
        st<V>(images_hat, u, h);
        Pack<V> ni;
#pragma unroll
        for (int j = 0; j < V; ++j) ni.v[j] = c_in * h.v[j];
        st<V>(net_in, u, ni);
    }
}

template <int V>
__global__ __launch_bounds__(256) void k_edm_precondition(const float* __restrict__ x, const float* __restrict__ net,
                                                          float* __restrict__ out, size_t units, Precond pc) {
    for (size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x; u < units; u += (size_t)gridDim.x * blockDim.x) {
        const Pack<V> xv = ld<V>(x, u), nv = ld<V>(net, u);
        Pack<V> o;
#pragma unroll
        for (int j = 0; j < V; ++j) o.v[j] = precondition_one(xv.v[j], nv.v[j], pc.c_skip, pc.c_out, pc.clamp);
        st<V>(out, u, o);
    }
}

// ---- dynamic threshold: s[b] = max(quantile(|x[b]|, q), floor_value) with torch.quantile's linear interpolation
// (elucidated_imagen.py:313-319).  The order statistics sorted[floor(rank)] and sorted[ceil(rank)], rank = q (n - 1)
// in fp32, are found exactly, without sorting: for x >= 0 the bit pattern of a float orders like the float, so three
// passes over the image - each fixing the next 11, 10, 10 bits of the wanted 31-bit key from a histogram of the
// candidates that still match the bits fixed so far - select the element of rank floor(rank).  Its upper neighbour
// is the same key (inside a run of equal values), else the next occupied bin of the last histogram, else the
// smallest key with larger high bits, which the last pass collects on the way.
// Small images: one workgroup per image does the three passes in one launch (k_edm_quantile_abs).  Large images: the
// image is cut in slices over many workgroups and every pass is a launch of its own (k_edm_quantile_pass): a workgroup
// counts its slice in LDS, adds its non-empty bins to the image's histogram in the workspace, and the workgroups of
// the NEXT launch each repeat the (cheap) bin selection from it, so no workgroup waits for another inside a launch.
constexpr int Q_WORDS = 3 * 2048 + 8;  // workspace words per image: three histograms, then {smallest larger key, NaN seen}
constexpr int Q_MIN = 3 * 2048, Q_NAN = 3 * 2048 + 1;
__host__ __device__ constexpr int q_bins(int pass) { return pass == 0 ? 2048 : 1024; }
__host__ __device__ constexpr int q_shift(int pass) { return pass == 0 ? 20 : pass == 1 ? 10 : 0; }
__host__ __device__ constexpr unsigned q_fixed(int pass) { return pass == 0 ? 0u : pass == 1 ? 0x7ff00000u : 0x7ffffc00u; }

struct QRank {
    float rank, rank_lo;
    unsigned lower, upper;
};
__device__ __forceinline__ QRank q_rank(float q, size_t n) {
#pragma clang fp contract(off)
    QRank r;
    r.rank = q * (float)(n - 1);
    r.rank_lo = floorf(r.rank);
    r.lower = (unsigned)r.rank_lo;
    r.upper = (unsigned)ceilf(r.rank);
    return r;
}

// Lanes 0-63 of a workgroup together: the bin of `hist` (NBINS counts, LDS or global, 16-byte aligned) that holds
// rank r -> sel = {bin, rank inside the bin, count of the bin}.  A lane fetches its NBINS / 64 consecutive counts with
// 16-byte loads, all in flight at once, and keeps them in registers for the search inside its range.
template <int NBINS>
__device__ __forceinline__ void q_select(const unsigned* hist, unsigned r, unsigned* sel) {
    constexpr int PER = NBINS / 64;
    const int lane = threadIdx.x;
    uint4 c4[PER / 4];
#pragma unroll
    for (int j = 0; j < PER / 4; ++j) c4[j] = reinterpret_cast<const uint4*>(hist + lane * PER)[j];
    unsigned c[PER];
#pragma unroll
    for (int j = 0; j < PER / 4; ++j) {
        c[4 * j] = c4[j].x; c[4 * j + 1] = c4[j].y; c[4 * j + 2] = c4[j].z; c[4 * j + 3] = c4[j].w;
    }
    unsigned s = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) s += c[j];
    unsigned incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    unsigned cum = incl - s;
    if (cum <= r && r < incl) {
        bool found = false;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            if (!found && r < cum + c[j]) {
                sel[0] = lane * PER + j;
                sel[1] = r - cum;
                sel[2] = c[j];
                found = true;
            }
            cum += c[j];
        }
    }
}

// Lanes 0-63: smallest occupied bin above `bin` in the last pass's histogram, or 0xffffffff.
__device__ __forceinline__ unsigned q_next_bin(const unsigned* hist, unsigned bin) {
    const int lane = threadIdx.x;
    constexpr int per = q_bins(2) / 64;
    unsigned best = 0xffffffffu;
    for (int j = per - 1; j >= 0; --j) {
        const unsigned t = lane * per + j;
        if (t > bin && hist[t]) best = t;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned t = __shfl_xor(best, o);
        best = t < best ? t : best;
    }
    return best;
}

// Counts units [u_begin, u_end) of an image into the LDS histogram of pass PASS.
template <int V, int PASS>
__device__ __forceinline__ void q_accumulate(const float* __restrict__ img, size_t u_begin, size_t u_end, unsigned prefix,
                                             unsigned* hist, unsigned* lds_min, int* lds_nan) {
    constexpr unsigned fixed = q_fixed(PASS), binmask = q_bins(PASS) - 1;
    constexpr int shift = q_shift(PASS);
    unsigned best = 0xffffffffu;
    bool nan = false;
    for (size_t u = u_begin + threadIdx.x; u < u_end; u += blockDim.x) {
        const Pack<V> v = ld<V>(img, u);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const unsigned key = __float_as_uint(v.v[j]) & 0x7fffffffu;
            if (PASS == 0 && key > 0x7f800000u) nan = true;
            if ((key & fixed) == prefix) atomicAdd(&hist[(key >> shift) & binmask], 1u);
            // the low bits of `prefix` are still zero: a key above it that differs in the fixed bits has larger high bits
            else if (PASS == 2 && key > prefix && key < best) best = key;
        }
    }
    if (PASS == 0 && nan) *lds_nan = 1;
    if (PASS == 2) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned t = __shfl_xor(best, o);
            best = t < best ? t : best;
        }
        if ((threadIdx.x & 63) == 0 && best != 0xffffffffu) atomicMin(lds_min, best);
    }
}

// key of rank `lower` (with r = its position inside the run of `run` equal keys), the key above it, NaN flag -> s
__device__ __forceinline__ float q_finish(const QRank& rk, unsigned key_lo, unsigned key_above, bool nan, float floor_value) {
#pragma clang fp contract(off)
    const float v_lo = __uint_as_float(key_lo);
    const float v_hi = rk.upper != rk.lower ? __uint_as_float(key_above) : v_lo;
    const float w = rk.rank - rk.rank_lo;
    const float diff = v_hi - v_lo;
    float s = w < 0.5f ? v_lo + w * diff : v_hi - diff * (1.0f - w);  // torch.lerp
    if (nan) s = __uint_as_float(0x7fc00000u);
    return s < floor_value ? floor_value : s;  // clamp_(min): NaN stays
}

template <int V>
__global__ __launch_bounds__(1024) void k_edm_quantile_abs(const float* __restrict__ x, float* __restrict__ out, size_t n,
                                                           float q, float floor_value) {
    __shared__ __attribute__((aligned(16))) unsigned hist[2048];
    __shared__ unsigned sel[3];
    __shared__ unsigned lds_min;
    __shared__ int lds_nan;
    const float* img = x + (size_t)blockIdx.x * n;
    const size_t units = n / V;
    const int tid = threadIdx.x;
    const QRank rk = q_rank(q, n);
    if (tid == 0) {
        lds_nan = 0;
        lds_min = 0xffffffffu;
    }
    unsigned prefix = 0, r = rk.lower;
    auto pass = [&](auto tag) {
        constexpr int PASS = decltype(tag)::value;
        for (int t = tid; t < q_bins(PASS); t += 1024) hist[t] = 0;
        __syncthreads();
        q_accumulate<V, PASS>(img, 0, units, prefix, hist, &lds_min, &lds_nan);
        __syncthreads();
        if (tid < 64) q_select<q_bins(PASS)>(hist, r, sel);
        __syncthreads();
        prefix |= sel[0] << q_shift(PASS);
        r = sel[1];
        if (PASS < 2) __syncthreads();  // hist and sel are rewritten by the next pass
    };
    pass(std::integral_constant<int, 0>{});
    pass(std::integral_constant<int, 1>{});
    pass(std::integral_constant<int, 2>{});
    if (tid < 64) {
        unsigned above = prefix;
        if (r + 1 >= sel[2]) {  // the upper rank lies beyond the run of equal keys
            const unsigned nb = q_next_bin(hist, sel[0]);
            above = nb != 0xffffffffu ? ((prefix & q_fixed(2)) | nb) : lds_min;
        }
        if (tid == 0) out[blockIdx.x] = q_finish(rk, prefix, above, lds_nan != 0, floor_value);
    }
}

__global__ __launch_bounds__(256) void k_edm_quantile_init(unsigned* __restrict__ ws, size_t words) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x)
        ws[i] = (i % Q_WORDS) == Q_MIN ? 0xffffffffu : 0u;
}

// The selections of the passes before PASS, from the image's histograms in the workspace (lanes 0-63 select, all
// threads of the workgroup receive): prefix = the key bits fixed so far, r = the rank among the remaining candidates.
template <int PASS>
__device__ __forceinline__ void q_replay(const unsigned* ws_img, unsigned lower, unsigned* sel, unsigned& prefix, unsigned& r) {
    prefix = 0;
    r = lower;
#pragma unroll
    for (int p = 0; p < PASS; ++p) {
        if (threadIdx.x < 64) {
            if (p == 0) q_select<q_bins(0)>(ws_img, r, sel);
            else q_select<q_bins(1)>(ws_img + p * 2048, r, sel);
        }
        __syncthreads();
        prefix |= sel[0] << q_shift(p);
        r = sel[1];
        __syncthreads();
    }
}

template <int V, int PASS>
__global__ __launch_bounds__(512) void k_edm_quantile_pass(const float* __restrict__ x, unsigned* __restrict__ ws, size_t n,
                                                           float q) {
    __shared__ __attribute__((aligned(16))) unsigned hist[2048];
    __shared__ unsigned sel[3];
    __shared__ unsigned lds_min;
    __shared__ int lds_nan;
    const int tid = threadIdx.x;
    const float* img = x + (size_t)blockIdx.y * n;
    unsigned* ws_img = ws + (size_t)blockIdx.y * Q_WORDS;
    const QRank rk = q_rank(q, n);
    unsigned prefix, r;
    q_replay<PASS>(ws_img, rk.lower, sel, prefix, r);
    for (int t = tid; t < q_bins(PASS); t += 512) hist[t] = 0;
    if (tid == 0) {
        lds_nan = 0;
        lds_min = 0xffffffffu;
    }
    __syncthreads();
    const size_t units = n / V;
    const size_t per = (units + gridDim.x - 1) / gridDim.x;
    const size_t begin = (size_t)blockIdx.x * per;
    const size_t end = begin + per < units ? begin + per : units;
    q_accumulate<V, PASS>(img, begin, end, prefix, hist, &lds_min, &lds_nan);
    __syncthreads();
    for (int t = tid; t < q_bins(PASS); t += 512) {
        const unsigned c = hist[t];
        if (c) atomicAdd(&ws_img[PASS * 2048 + t], c);
    }
    if (tid == 0) {
        if (PASS == 0 && lds_nan) atomicOr(&ws_img[Q_NAN], 1u);
        if (PASS == 2 && lds_min != 0xffffffffu) atomicMin(&ws_img[Q_MIN], lds_min);
    }
}

__global__ __launch_bounds__(64) void k_edm_quantile_result(const unsigned* __restrict__ ws, float* __restrict__ out, size_t n,
                                                            float q, float floor_value) {
    __shared__ unsigned sel[3];
    const unsigned* ws_img = ws + (size_t)blockIdx.x * Q_WORDS;
    const QRank rk = q_rank(q, n);
    unsigned prefix, r;
    q_replay<3>(ws_img, rk.lower, sel, prefix, r);
    unsigned above = prefix;
    if (r + 1 >= sel[2]) {
        const unsigned nb = q_next_bin(ws_img + 2 * 2048, sel[0]);
        above = nb != 0xffffffffu ? ((prefix & q_fixed(2)) | nb) : ws_img[Q_MIN];
    }
    if (threadIdx.x == 0) out[blockIdx.x] = q_finish(rk, prefix, above, ws_img[Q_NAN] != 0, floor_value);
}

// ---- Euler step.  model_out: with `net` the preconditioned network output is formed here and written; without, the
// array already holds it (unthresholded) and is read.  `thr` (per image) applies the dynamic threshold to it in place.
// Then d = (images_hat - model_out) / sigma_hat, images_next = images_hat + dt * d, and the second evaluation's input
// c_in_next * images_next when net_in_next is given (elucidated_imagen.py:509-511, :359).
template <int V>
__global__ __launch_bounds__(256) void k_edm_euler(const float* __restrict__ images_hat, const float* __restrict__ net,
                                                   float* __restrict__ model_out, const float* __restrict__ thr,
                                                   float* __restrict__ d_out, float* __restrict__ images_next,
                                                   float* __restrict__ net_in_next, size_t units, size_t units_per_image,
                                                   Precond pc, float sigma_hat, float dt, float c_in_next) {
#pragma clang fp contract(off)
    for (size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x; u < units; u += (size_t)gridDim.x * blockDim.x) {
        const Pack<V> h = ld<V>(images_hat, u);
        Pack<V> o;
        if (net) {
            const Pack<V> nv = ld<V>(net, u);
#pragma unroll
            for (int j = 0; j < V; ++j) o.v[j] = precondition_one(h.v[j], nv.v[j], pc.c_skip, pc.c_out, pc.clamp);
        } else {
            o = ld<V>(model_out, u);
        }
        if (thr) {
            const float s = thr[u / units_per_image];
#pragma unroll
            for (int j = 0; j < V; ++j) o.v[j] = threshold_one(o.v[j], s);
        }
        if (net || thr) st<V>(model_out, u, o);
        Pack<V> d, nx, ni;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const float diff = h.v[j] - o.v[j];
            d.v[j] = diff / sigma_hat;
            const float step = dt * d.v[j];
            nx.v[j] = h.v[j] + step;
            ni.v[j] = c_in_next * nx.v[j];
        }
        if (d_out) st<V>(d_out, u, d);
        st<V>(images_next, u, nx);
        if (net_in_next) st<V>(net_in_next, u, ni);
    }
}

// ---- second-order correction: model_out (of the evaluation at images_next, sigma_next) formed / thresholded as in
// the Euler kernel, d' = (images_next - model_out) / sigma_next, images = images_hat + half_dt * (d + d')
// (elucidated_imagen.py:528-529).
template <int V>
__global__ __launch_bounds__(256) void k_edm_heun(const float* __restrict__ images_hat, const float* __restrict__ images_next,
                                                  const float* __restrict__ net, float* __restrict__ model_out,
                                                  const float* __restrict__ thr, const float* __restrict__ d_in,
                                                  float* __restrict__ images_out, size_t units, size_t units_per_image,
                                                  Precond pc, float sigma_next, float half_dt) {
#pragma clang fp contract(off)
    for (size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x; u < units; u += (size_t)gridDim.x * blockDim.x) {
        const Pack<V> h = ld<V>(images_hat, u), nx = ld<V>(images_next, u), d = ld<V>(d_in, u);
        Pack<V> o;
        if (net) {
            const Pack<V> nv = ld<V>(net, u);
#pragma unroll
            for (int j = 0; j < V; ++j) o.v[j] = precondition_one(nx.v[j], nv.v[j], pc.c_skip, pc.c_out, pc.clamp);
        } else {
            o = ld<V>(model_out, u);
        }
        if (thr) {
            const float s = thr[u / units_per_image];
#pragma unroll
            for (int j = 0; j < V; ++j) o.v[j] = threshold_one(o.v[j], s);
        }
        if (net || thr) st<V>(model_out, u, o);
        Pack<V> res;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const float diff = nx.v[j] - o.v[j];
            const float dp = diff / sigma_next;
            const float sum = d.v[j] + dp;
            const float step = half_dt * sum;
            res.v[j] = h.v[j] + step;
        }
        st<V>(images_out, u, res);
    }
}

// ---- repaint: images += scale * z (elucidated_imagen.py:535-536)
template <int V>
__global__ __launch_bounds__(256) void k_edm_renoise(float* __restrict__ images, const float* __restrict__ noise, size_t units,
                                                     float scale, int philox, uint64_t seed, uint64_t offset) {
#pragma clang fp contract(off)
    for (size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x; u < units; u += (size_t)gridDim.x * blockDim.x) {
        Pack<V> x = ld<V>(images, u);
        Pack<V> z;
        if (philox) z = draw<V>(u, seed, offset);
        else z = ld<V>(noise, u);
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const float t = scale * z.v[j];
            x.v[j] = x.v[j] + t;
        }
        st<V>(images, u, x);
    }
}

// ---- end of a stage: clamp to [-1, 1], inpainted region restored, (x + 1) / 2 when images are kept in [0, 1]
// outside the sampler (elucidated_imagen.py:540-545)
template <int V>
__global__ __launch_bounds__(256) void k_edm_finish(float* __restrict__ images, const float* __restrict__ inpaint,
                                                    const unsigned char* __restrict__ mask, size_t units, unsigned C,
                                                    unsigned HW, int unnormalize) {
#pragma clang fp contract(off)
    const size_t per_image = (size_t)C * HW;
    for (size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x; u < units; u += (size_t)gridDim.x * blockDim.x) {
        Pack<V> x = ld<V>(images, u);
#pragma unroll
        for (int j = 0; j < V; ++j) x.v[j] = clamp_keep_nan(x.v[j], -1.0f, 1.0f);
        if (mask) {
            const size_t e0 = u * V;
            const size_t b = e0 / per_image;
            const size_t pix = (e0 - b * per_image) % HW;
            const Pack<V> inp = ld<V>(inpaint, u);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float m = mask[b * HW + pix + j] ? 1.0f : 0.0f;
                const float keep = x.v[j] * (1.0f - m);
                const float fill = inp.v[j] * m;
                x.v[j] = keep + fill;
            }
        }
        if (unnormalize) {
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float t = x.v[j] + 1.0f;
                x.v[j] = t * 0.5f;
            }
        }
        st<V>(images, u, x);
    }
}

int grid_for(size_t units) {
    const size_t blocks = (units + 255) / 256;
    return (int)(blocks > 2048 ? 2048 : blocks);
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

// Vector width of a launch: float4 when no unit straddles an image row group and every array allows it.
static int edm_width(size_t per_image, size_t HW, std::initializer_list<const void*> ptrs) {
    if (per_image % 4 || HW % 4) return 1;
    for (const void* p : ptrs)
        if (p && !aligned16(p)) return 1;
    return 4;
}

hipError_t launch_edm_perturb(const float* images, const float* noise, const float* inpaint, const unsigned char* mask,
                              float* images_hat, float* net_in, int B, int C, int HW, float s_noise, float k, float c_in,
                              int philox, uint64_t seed, uint64_t offset, hipStream_t stream) {
    const size_t per_image = (size_t)C * HW, n = (size_t)B * per_image;
    const int V = edm_width(per_image, HW, {images, noise, inpaint, images_hat, net_in});
    const size_t units = n / V;
    if (V == 4)
        hipLaunchKernelGGL(k_edm_perturb<4>, dim3(grid_for(units)), dim3(256), 0, stream, images, noise, inpaint, mask,
                           images_hat, net_in, units, (unsigned)C, (unsigned)HW, s_noise, k, c_in, philox, seed, offset);
    else
        hipLaunchKernelGGL(k_edm_perturb<1>, dim3(grid_for(units)), dim3(256), 0, stream, images, noise, inpaint, mask,
                           images_hat, net_in, units, (unsigned)C, (unsigned)HW, s_noise, k, c_in, philox, seed, offset);
    return hipGetLastError();
}

hipError_t launch_edm_precondition(const float* x, const float* net, float* out, size_t n, float c_skip, float c_out,
                                   int clamp, hipStream_t stream) {
    const int V = edm_width(n, 4, {x, net, out});
    const size_t units = n / V;
    const Precond pc{c_skip, c_out, clamp};
    if (V == 4)
        hipLaunchKernelGGL(k_edm_precondition<4>, dim3(grid_for(units)), dim3(256), 0, stream, x, net, out, units, pc);
    else
        hipLaunchKernelGGL(k_edm_precondition<1>, dim3(grid_for(units)), dim3(256), 0, stream, x, net, out, units, pc);
    return hipGetLastError();
}

// Images above this many values are cut in slices over several workgroups (one launch per pass, workspace needed).
static constexpr size_t Q_SINGLE_MAX = 32768;
static int q_slices(int B, size_t per_image, int V) {
    const size_t units = per_image / V;
    size_t s = (size_t)(1024 + B - 1) / B;               // about a thousand workgroups in all
    const size_t most = (units + 511) / 512;              // at least one unit per thread
    if (s > most) s = most;
    return (int)(s < 1 ? 1 : s);
}

size_t edm_quantile_workspace_bytes(int B, size_t per_image) {
    return per_image <= Q_SINGLE_MAX ? 0 : (size_t)B * Q_WORDS * sizeof(unsigned);
}

hipError_t launch_edm_quantile_abs(const float* x, float* out, int B, size_t per_image, float q, float floor_value,
                                   void* workspace, hipStream_t stream) {
    const int V = edm_width(per_image, 4, {x});
    if (per_image <= Q_SINGLE_MAX) {
        if (V == 4)
            hipLaunchKernelGGL(k_edm_quantile_abs<4>, dim3(B), dim3(1024), 0, stream, x, out, per_image, q, floor_value);
        else
            hipLaunchKernelGGL(k_edm_quantile_abs<1>, dim3(B), dim3(1024), 0, stream, x, out, per_image, q, floor_value);
        return hipGetLastError();
    }
    if (!workspace) return hipErrorInvalidValue;
    unsigned* ws = static_cast<unsigned*>(workspace);
    const size_t words = (size_t)B * Q_WORDS;
    hipLaunchKernelGGL(k_edm_quantile_init, dim3(grid_for(words)), dim3(256), 0, stream, ws, words);
    const dim3 grid(q_slices(B, per_image, V), B);
    if (V == 4) {
        hipLaunchKernelGGL((k_edm_quantile_pass<4, 0>), grid, dim3(512), 0, stream, x, ws, per_image, q);
        hipLaunchKernelGGL((k_edm_quantile_pass<4, 1>), grid, dim3(512), 0, stream, x, ws, per_image, q);
        hipLaunchKernelGGL((k_edm_quantile_pass<4, 2>), grid, dim3(512), 0, stream, x, ws, per_image, q);
    } else {
        hipLaunchKernelGGL((k_edm_quantile_pass<1, 0>), grid, dim3(512), 0, stream, x, ws, per_image, q);
        hipLaunchKernelGGL((k_edm_quantile_pass<1, 1>), grid, dim3(512), 0, stream, x, ws, per_image, q);
        hipLaunchKernelGGL((k_edm_quantile_pass<1, 2>), grid, dim3(512), 0, stream, x, ws, per_image, q);
    }
    hipLaunchKernelGGL(k_edm_quantile_result, dim3(B), dim3(64), 0, stream, ws, out, per_image, q, floor_value);
    return hipGetLastError();
}

hipError_t launch_edm_euler(const float* images_hat, const float* net, float* model_out, const float* thr, float* d_out,
                            float* images_next, float* net_in_next, int B, size_t per_image, float c_skip, float c_out,
                            int clamp, float sigma_hat, float dt, float c_in_next, hipStream_t stream) {
    const int V = edm_width(per_image, 4, {images_hat, net, model_out, d_out, images_next, net_in_next});
    const size_t units = (size_t)B * per_image / V;
    const Precond pc{c_skip, c_out, clamp};
    if (V == 4)
        hipLaunchKernelGGL(k_edm_euler<4>, dim3(grid_for(units)), dim3(256), 0, stream, images_hat, net, model_out, thr, d_out,
                           images_next, net_in_next, units, per_image / 4, pc, sigma_hat, dt, c_in_next);
    else
        hipLaunchKernelGGL(k_edm_euler<1>, dim3(grid_for(units)), dim3(256), 0, stream, images_hat, net, model_out, thr, d_out,
                           images_next, net_in_next, units, per_image, pc, sigma_hat, dt, c_in_next);
    return hipGetLastError();
}

hipError_t launch_edm_heun(const float* images_hat, const float* images_next, const float* net, float* model_out,
                           const float* thr, const float* d_in, float* images_out, int B, size_t per_image, float c_skip,
                           float c_out, int clamp, float sigma_next, float half_dt, hipStream_t stream) {
    const int V = edm_width(per_image, 4, {images_hat, images_next, net, model_out, d_in, images_out});
    const size_t units = (size_t)B * per_image / V;
    const Precond pc{c_skip, c_out, clamp};
    if (V == 4)
        hipLaunchKernelGGL(k_edm_heun<4>, dim3(grid_for(units)), dim3(256), 0, stream, images_hat, images_next, net, model_out,
                           thr, d_in, images_out, units, per_image / 4, pc, sigma_next, half_dt);
    else
        hipLaunchKernelGGL(k_edm_heun<1>, dim3(grid_for(units)), dim3(256), 0, stream, images_hat, images_next, net, model_out,
                           thr, d_in, images_out, units, per_image, pc, sigma_next, half_dt);
    return hipGetLastError();
}

hipError_t launch_edm_renoise(float* images, const float* noise, size_t n, float scale, int philox, uint64_t seed,
                              uint64_t offset, hipStream_t stream) {
    const int V = edm_width(n, 4, {images, noise});
    const size_t units = n / V;
    if (V == 4)
        hipLaunchKernelGGL(k_edm_renoise<4>, dim3(grid_for(units)), dim3(256), 0, stream, images, noise, units, scale, philox,
                           seed, offset);
    else
        hipLaunchKernelGGL(k_edm_renoise<1>, dim3(grid_for(units)), dim3(256), 0, stream, images, noise, units, scale, philox,
                           seed, offset);
    return hipGetLastError();
}

hipError_t launch_edm_finish(float* images, const float* inpaint, const unsigned char* mask, int B, int C, int HW,
                             int unnormalize, hipStream_t stream) {
    const size_t per_image = (size_t)C * HW, n = (size_t)B * per_image;
    const int V = edm_width(per_image, HW, {images, inpaint});
    const size_t units = n / V;
    if (V == 4)
        hipLaunchKernelGGL(k_edm_finish<4>, dim3(grid_for(units)), dim3(256), 0, stream, images, inpaint, mask, units,
                           (unsigned)C, (unsigned)HW, unnormalize);
    else
        hipLaunchKernelGGL(k_edm_finish<1>, dim3(grid_for(units)), dim3(256), 0, stream, images, inpaint, mask, units,
                           (unsigned)C, (unsigned)HW, unnormalize);
    return hipGetLastError();
}

}  // namespace danra
