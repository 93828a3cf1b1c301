// Developer tool (not part of libdanra_hip.so): the strided window kernel (kernels_conv_s2.hip) on the encoder's 8x8
// stride-2 convolution: timed beside the implicit-GEMM kernel it replaces, with the phase clock of workgroup 0.
//   ./tune_conv_s2 [batch] [iters]
#include <hip/hip_runtime.h>

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

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64;
    const int iters = argc > 2 ? atoi(argv[2]) : 20;
    const int H = 32, W = 32, Cin = 64, Cout = 64, K = 64 * Cin, M = B * (H / 2) * (W / 2);
    const size_t nx = (size_t)B * H * W * Cin, nw = (size_t)Cout * K, no = (size_t)M * Cout;
    float *x, *out, *out_ref, *bias, *w, *wh, *wf, *zeros, *partial;
    CK(hipMalloc(&x, nx * 4)); CK(hipMalloc(&out, no * 4)); CK(hipMalloc(&out_ref, no * 4)); CK(hipMalloc(&bias, 4096));
    CK(hipMalloc(&w, nw * 4)); CK(hipMalloc(&wh, nw * 4)); CK(hipMalloc(&wf, nw * 4)); CK(hipMalloc(&zeros, 256)); CK(hipMemset(zeros, 0, 256));
    CK(hipMalloc(&partial, (size_t)64 << 20));
    std::vector<float> hx(nx), hw(nw), hb(1024);
    srand(7);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (auto& v : hx) v = rnd();
    for (auto& v : hw) v = rnd() * 0.02f;
    for (auto& v : hb) v = rnd();
    CK(hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, hb.data(), 4096, hipMemcpyHostToDevice));
    const float wscale = 131072.f;  // |w| < 0.02 -> below 2^12
    CK(launch_pack_conv_frag(w, wf, Cout, Cin, wscale, 0, 64));
    CK(launch_split_weights(w, wh, nw, wscale, 0));
    ConvWinArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.wf = wf; a.w32 = w; a.wscale = 1.0f / wscale; a.bias = bias; a.out = out;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.act = ACT_RELU;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) CK(launch_conv8s2(a, 0));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) CK(launch_conv8s2(a, 0));
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
            printf("      wave %d: geom+wloads %llu  win-load-issue %llu  geometry+store %llu  wloads+barrier %llu  loop %llu (16 steps %llu, next 16 %llu)  barrier+vote %llu  epilogue %llu  total %llu\n",
                   wv, st[wv * 16 + 1] - st[wv * 16], st[wv * 16 + 2] - st[wv * 16 + 1], st[wv * 16 + 3] - st[wv * 16 + 2],
                   st[wv * 16 + 4] - st[wv * 16 + 3], st[wv * 16 + 5] - st[wv * 16 + 4], st[wv * 16 + 9] - st[wv * 16 + 4],
                   st[wv * 16 + 10] - st[wv * 16 + 9], st[wv * 16 + 6] - st[wv * 16 + 5], st[wv * 16 + 7] - st[wv * 16 + 6],
                   st[wv * 16 + 7] - st[wv * 16]);
        }
    }
#endif
    IGemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = x; g.W = w; g.Wh = wh; g.wscale = 1.0f / wscale; g.half = 1; g.bias = bias; g.out = out_ref; g.partial = partial; g.zeros = zeros;
    g.M = M; g.N = Cout; g.K = K; g.a_mode = A_CONV; g.Hi = H; g.Wi = W; g.Cin = Cin; g.Ho = H / 2; g.Wo = W / 2; g.KH = 8; g.KW = 8;
    g.stride = 2; g.pad = 3; g.ldr = Cout; g.act = ACT_RELU; g.rows_per_img = (H / 2) * (W / 2); g.st_mode = ST_ROW; g.ldc = Cout;
    g.batch = B; g.ksplit = 1; g.chunks_per_split = K / 32;
    for (int i = 0; i < 3; ++i) CK(launch_igemm(g, 0));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) CK(launch_igemm(g, 0));
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<float> got(no), ref(no);
    CK(hipMemcpy(got.data(), out, no * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ref.data(), out_ref, no * 4, hipMemcpyDeviceToHost));
    double d = 0, mx = 0;
    for (size_t i = 0; i < no; ++i) { d = fmax(d, fabs((double)ref[i] - got[i])); mx = fmax(mx, fabs((double)ref[i])); }
    printf("8x8 s2 %d->%d @%dx%d B=%d: new %.1f us  igemm %.1f us   max|new - igemm| %.2e of %.2e\n", Cin, Cout, H, W, B, us_new,
           ms * 1e3 / iters, d, mx);
    return 0;
}
