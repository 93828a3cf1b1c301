// Developer tool (not part of libdanra_hip.so): the window-staged 3x3 convolution kernel (kernels_conv_win.hip) on the
// network's layer shapes: checked against a float64 evaluation on the host (sampled outputs) and timed beside the
// implicit-GEMM kernel it replaces.
//   ./tune_conv_win [batch] [only_shape] [iters] [range_case]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "danra_kernels.h"

using namespace danra;
#if DANRA_WIN_DEBUG
namespace danra { extern __device__ unsigned long long g_win_stamps[8 * 16]; }
#endif

#define CK(x)                                                                           \
    do {                                                                                \
        hipError_t e = (x);                                                             \
        if (e != hipSuccess) {                                                          \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__);  \
            exit(1);                                                                    \
        }                                                                               \
    } while (0)

struct Shape { const char* name; int H, W, Cin, Cout; int norm; int bdiv; };

int main(int argc, char** argv) {
    const int B0 = argc > 1 ? atoi(argv[1]) : 64;
    const int only = argc > 2 ? atoi(argv[2]) : -1;
    const int iters = argc > 3 ? atoi(argv[3]) : 20;
    const int range_case = argc > 4 ? atoi(argv[4]) : 0;
    std::vector<Shape> shapes = {
        {"layer1 64->64 @16", 16, 16, 64, 64, 0, 1},   {"layer2 128->128 @8", 8, 8, 128, 128, 0, 1},
        {"layer3 256->256 @4", 4, 4, 256, 256, 0, 1},  {"layer4 512->512 @2", 2, 2, 512, 512, 0, 1},
        {"D0 512->256 @4", 4, 4, 512, 256, 0, 1},      {"D1 256->128 @8", 8, 8, 256, 128, 0, 1},
        {"D2 128->64 @16 (IN)", 16, 16, 128, 64, 1, 1}, {"D3 64->64 @32 (IN)", 32, 32, 64, 64, 1, 1},
        {"D3 64->64 @64 (IN)", 64, 64, 64, 64, 1, 4},   {"layer1 64->64 @32", 32, 32, 64, 64, 0, 2},
        {"ragged 256->256 @4 B=3", 4, 4, 256, 256, 0, -3}, {"ragged 512->512 @2 B=5", 2, 2, 512, 512, 0, -5},
    };
    const size_t big = (size_t)64 * 64 * 64 * 64 + (1 << 20);
    float *x, *out, *out_ref, *res, *bias, *tb, *w, *wh, *wf, *mean, *rstd, *zeros, *partial;
    CK(hipMalloc(&x, big * 4)); CK(hipMalloc(&out, big * 4)); CK(hipMalloc(&out_ref, big * 4)); CK(hipMalloc(&res, big * 4));
    CK(hipMalloc(&bias, 4096)); CK(hipMalloc(&tb, 64 * 1024 * 4)); CK(hipMalloc(&w, (size_t)512 * 512 * 9 * 4));
    CK(hipMalloc(&wh, (size_t)512 * 512 * 9 * 4)); CK(hipMalloc(&wf, (size_t)512 * 512 * 9 * 4));
    CK(hipMalloc(&mean, 64 * 512 * 4)); CK(hipMalloc(&rstd, 64 * 512 * 4)); CK(hipMalloc(&zeros, 256)); CK(hipMemset(zeros, 0, 256));
    CK(hipMalloc(&partial, (size_t)64 << 20));
    std::vector<float> hx(big), hw((size_t)512 * 512 * 9), hres(big), hb(1024), htb(64 * 1024), hm(64 * 512), hr(64 * 512);
    srand(7);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (auto& v : hx) v = rnd();
    for (auto& v : hw) v = rnd() * 0.05f;
    for (auto& v : hres) v = rnd();
    for (auto& v : hb) v = rnd();
    for (auto& v : htb) v = rnd();
    for (auto& v : hm) v = rnd() * 0.3f;
    for (auto& v : hr) v = 0.5f + fabsf(rnd());
    if (range_case == 1) for (size_t i = 12345; i < big; i += 1000003) hx[i] = 3.0e6f;
    if (range_case == 2) for (auto& v : hx) v *= 1.0e-4f;
    if (range_case == 3) for (auto& v : hx) v *= 3.0e-3f;
    CK(hipMemcpy(x, hx.data(), big * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(res, hres.data(), big * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, hb.data(), 4096, hipMemcpyHostToDevice));
    CK(hipMemcpy(tb, htb.data(), htb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(mean, hm.data(), hm.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(rstd, hr.data(), hr.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const float wscale = 65536.f;  // |w| < 0.05 -> below 2^12
    int failures = 0;
    for (size_t si = 0; si < shapes.size(); ++si) {
        if (only >= 0 && (int)si != only) continue;
        const Shape& s = shapes[si];
        const int B = s.bdiv < 0 ? -s.bdiv : std::max(1, B0 / s.bdiv);
        const int K = 9 * s.Cin, M = B * s.H * s.W;
        if (!conv_win_supported(B, s.H, s.W, s.Cin, s.Cout, s.norm != 0)) { printf("%-26s unsupported\n", s.name); continue; }
        CK(launch_pack_conv_frag(w, wf, s.Cout, s.Cin, wscale, 0));
        CK(launch_split_weights(w, wh, (size_t)s.Cout * K, wscale, 0));
        ConvWinArgs a;
        memset(&a, 0, sizeof(a));
        a.x = x; a.wf = wf; a.w32 = w; a.wscale = 1.0f / wscale; a.bias = bias; a.res = res; a.tbias = tb; a.tb_stride = 1024;
        a.nmean = s.norm ? mean : nullptr; a.nrstd = s.norm ? rstd : nullptr; a.out = out;
        a.B = B; a.H = s.H; a.W = s.W; a.Cin = s.Cin; a.Cout = s.Cout; a.act = ACT_RELU;
        CK(hipMemset(out, 0xff, (size_t)M * s.Cout * 4));
        CK(launch_conv_win(a, 0));
        CK(hipDeviceSynchronize());
        std::vector<float> got((size_t)M * s.Cout);
        CK(hipMemcpy(got.data(), out, got.size() * 4, hipMemcpyDeviceToHost));
        // float64 reference on sampled outputs (all rows of a few channels + a few full rows)
        double err = 0, mx = 0;
        size_t checked = 0;
        auto check = [&](int m, int n) {
            const int b = m / (s.H * s.W), r = m % (s.H * s.W), oy = r / s.W, ox = r % s.W;
            double acc = 0;
            for (int kh = 0; kh < 3; ++kh)
                for (int kw = 0; kw < 3; ++kw) {
                    const int iy = oy + kh - 1, ix = ox + kw - 1;
                    if (iy < 0 || iy >= s.H || ix < 0 || ix >= s.W) continue;
                    const float* xr = &hx[((size_t)(b * s.H + iy) * s.W + ix) * s.Cin];
                    const float* wr = &hw[(size_t)n * K + (kh * 3 + kw) * s.Cin];
                    for (int c = 0; c < s.Cin; ++c) {
                        double xv = xr[c];
                        if (s.norm) xv = (double)(float)((xr[c] - hm[b * s.Cin + c]) * hr[b * s.Cin + c]);
                        acc += xv * wr[c];
                    }
                }
            acc += hb[n];
            acc += hres[(size_t)m * s.Cout + n];
            acc = acc > 0 ? acc : 0;
            acc += htb[b * 1024 + n];
            const double g = got[(size_t)m * s.Cout + n];
            if (!(fabs(g - acc) <= 1e30)) err = 1e30;
            err = std::max(err, fabs(g - acc));
            mx = std::max(mx, fabs(acc));
            ++checked;
        };
        for (int m = 0; m < M; m += (M > 4096 ? 7 : 1))
            for (int n = (m % 5); n < s.Cout; n += 13) check(m, n);
        for (int n = 0; n < s.Cout; ++n) { check(0, n); check(M - 1, n); check(M / 2 + 3 < M ? M / 2 + 3 : 0, n); }
        const bool ok = err <= 1e-5 * mx;
        if (!ok) ++failures;
        // timing, and the implicit-GEMM kernel on the same problem
        for (int i = 0; i < 3; ++i) CK(launch_conv_win(a, 0));
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) CK(launch_conv_win(a, 0));
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us_new = ms * 1e3 / iters;
#if DANRA_WIN_DEBUG
        {
            unsigned long long st[8 * 16];
            CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(danra::g_win_stamps), sizeof(st)));
            for (int wv = 0; wv < 8; wv += 3) {
                printf("      wave %d:", wv);
                const char* names[] = {"geom", "wprefetch", "win-load-issue", "win-store", "barrier", "loop", "vote+partials", "epilogue"};
                for (int k = 0; k < 8; ++k) printf("  %s %llu", names[k], st[wv * 16 + k + 1] - st[wv * 16 + k]);
                printf("  total %llu | args+wloads %llu\n", st[wv * 16 + 8] - st[wv * 16], st[wv * 16 + 9] - st[wv * 16]);
            }
        }
#endif
        IGemmArgs g;
        memset(&g, 0, sizeof(g));
        g.A = x; g.W = w; g.Wh = wh; g.wscale = 1.0f / wscale; g.half = 1; g.bias = bias; g.out = out_ref; g.partial = partial; g.zeros = zeros;
        g.M = M; g.N = s.Cout; g.K = K; g.a_mode = A_CONV; g.Hi = s.H; g.Wi = s.W; g.Cin = s.Cin; g.Ho = s.H; g.Wo = s.W; g.KH = 3; g.KW = 3;
        g.stride = 1; g.pad = 1; g.res = res; g.ldr = s.Cout; g.act = ACT_RELU; g.rows_per_img = s.H * s.W; g.st_mode = ST_ROW; g.ldc = s.Cout;
        g.tbias = tb; g.tb_stride = 1024; g.nmean = a.nmean; g.nrstd = a.nrstd;
        g.pix_major = (s.H * s.W <= 64) ? 1 : 0; g.batch = B;
        g.ksplit = 1; g.chunks_per_split = K / 32;
        {
            const long tiles = (long)((M + 63) / 64) * (s.Cout / 64);
            const int chunks = K / 32;
            if (tiles < 256 && chunks >= 16) {
                int want = (int)((256 + tiles - 1) / tiles);
                if (want > chunks / 8) want = chunks / 8;
                if (want > 1) {
                    int cps = (chunks + want - 1) / want;
                    cps += cps & 1;
                    g.chunks_per_split = cps;
                    g.ksplit = (chunks + cps - 1) / cps;
                }
            }
        }
        double us_old = 0;
        if (s.Cout % 64 == 0) {
            for (int i = 0; i < 3; ++i) CK(launch_igemm(g, 0));
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < iters; ++i) CK(launch_igemm(g, 0));
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            us_old = ms * 1e3 / iters;
            // and the two kernels against each other on every output
            std::vector<float> ref((size_t)M * s.Cout);
            CK(hipMemcpy(ref.data(), out_ref, ref.size() * 4, hipMemcpyDeviceToHost));
            double d = 0;
            for (size_t i = 0; i < ref.size(); ++i) d = std::max(d, (double)fabsf(ref[i] - got[i]));
            printf("%-26s B=%-3d new %7.1f us  igemm(ks=%d) %7.1f us   err vs f64 %.2e of %.2e (%.1e) over %zu   max|new - igemm| %.2e  %s\n",
                   s.name, B, us_new, g.ksplit, us_old, err, mx, err / mx, checked, d, ok ? "ok" : "FAIL");
        } else {
            printf("%-26s B=%-3d new %7.1f us   err vs f64 %.2e of %.2e (%.1e) over %zu  %s\n", s.name, B, us_new, err, mx, err / mx,
                   checked, ok ? "ok" : "FAIL");
        }
        fflush(stdout);
    }
    printf(failures ? "FAILURES: %d\n" : "all ok\n", failures);
    return failures ? 1 : 0;
}
