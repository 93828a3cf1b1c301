// Developer tool (not part of libdanra_hip.so): times the implicit-GEMM kernel on the network's
// shapes with forced tile configurations, with HIP events, on random data.
//   ./tune_igemm            all shapes x all configurations
// Output: one line per (shape, config): microseconds and TFLOP/s.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "danra_kernels.h"

using namespace danra;

#define CK(x)                                                                           \
    do {                                                                                \
        hipError_t e = (x);                                                             \
        if (e != hipSuccess) {                                                          \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                      \
            exit(1);                                                                    \
        }                                                                               \
    } while (0)

struct Shape {
    const char* name;
    int mode;  // A_GEMM / A_CONV
    int B, Hi, Wi, Cin, Cout, k, stride, pad;  // conv; for GEMM: M = B*Hi*Wi, K = Cin, N = Cout
    int ln, st_mode;
};

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64;
    const int only_shape = argc > 2 ? atoi(argv[2]) : -1;  // index into `shapes`
    const int only_cfg = argc > 3 ? atoi(argv[3]) : -1;    // index into `cfgs`
    const int iters = argc > 4 ? atoi(argv[4]) : 20;
    const int debug = argc > 5 ? atoi(argv[5]) : 0;
    std::vector<Shape> shapes = {
        {"conv2 8x8s2 64->64 @32", A_CONV, B, 32, 32, 64, 64, 8, 2, 3, 0, ST_ROW},
        {"layer1 3x3 64->64 @16", A_CONV, B, 16, 16, 64, 64, 3, 1, 1, 0, ST_ROW},
        {"layer2 3x3 128->128 @8", A_CONV, B, 8, 8, 128, 128, 3, 1, 1, 0, ST_ROW},
        {"layer3 3x3 256->256 @4", A_CONV, B, 4, 4, 256, 256, 3, 1, 1, 0, ST_ROW},
        {"layer4 3x3 512->512 @2", A_CONV, B, 2, 2, 512, 512, 3, 1, 1, 0, ST_ROW},
        {"D0 3x3 512->256 @4", A_CONV, B, 4, 4, 512, 256, 3, 1, 1, 0, ST_ROW},
        {"D1 3x3 256->128 @8", A_CONV, B, 8, 8, 256, 128, 3, 1, 1, 0, ST_ROW},
        {"D2 3x3 128->64 @16", A_CONV, B, 16, 16, 128, 64, 3, 1, 1, 0, ST_ROW},
        {"D3 3x3 64->64 @32", A_CONV, B, 32, 32, 64, 64, 3, 1, 1, 0, ST_ROW},
        {"qkv L1024 C64 (ln)", A_GEMM, B, 32, 32, 64, 192, 1, 1, 0, 1, ST_QKV},
        {"outproj L1024 C64", A_GEMM, B, 32, 32, 64, 64, 1, 1, 0, 0, ST_ROW},
        {"convT DF 64->4x64 @32", A_GEMM, B, 32, 32, 64, 256, 1, 1, 0, 0, ST_CONVT},
        {"qkv L4 C512 (ln)", A_GEMM, B, 2, 2, 512, 1536, 1, 1, 0, 1, ST_QKV},
    };
    struct Cfg { int bm, kw, ks; };
    std::vector<Cfg> cfgs = {{0, 0, 1}, {128, 1, 1}, {64, 1, 1}, {64, 2, 1}, {64, 4, 1}, {64, 4, 2}, {64, 4, 4}, {64, 4, 8},
                             {64, 2, 2}, {64, 2, 4}, {64, 1, 4}, {128, 1, 2}, {128, 1, 4}, {64, 12, 1}, {64, 12, 2}, {64, 12, 4}};
    const size_t big = (size_t)B * 64 * 64 * 64 + (1 << 20);
    float *A, *W, *out, *res, *bias, *partial;
    CK(hipMalloc(&A, big * 4));
    CK(hipMalloc(&out, big * 4 * 4));
    CK(hipMalloc(&res, big * 4));
    CK(hipMalloc(&W, (size_t)16 << 20));
    CK(hipMalloc(&bias, 1 << 16));
    CK(hipMalloc(&partial, (size_t)64 << 20));
    float* zeros;
    CK(hipMalloc(&zeros, 256));
    CK(hipMemset(zeros, 0, 256));
    std::vector<float> h(big);
    srand(1);
    for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    CK(hipMemcpy(res, h.data(), big * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, h.data(), (size_t)16 << 20, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, h.data(), 1 << 16, hipMemcpyHostToDevice));
    if (const char* e = getenv("DANRA_TUNE_RANGE")) {  // exercise the range check of the split-operand form
        if (atoi(e) == 1) for (size_t i = 12345; i < big; i += 1000003) h[i] = 3.0e6f;     // values beyond the fp16 range in some tiles
        if (atoi(e) == 2) for (auto& v : h) v *= 1.0e-4f;                                  // a uniformly tiny operand
        if (atoi(e) == 3) for (size_t i = 0; i < big; ++i) if ((i / 4096) % 3 == 0) h[i] = 0.f;  // zero stretches
    }
    CK(hipMemcpy(A, h.data(), big * 4, hipMemcpyHostToDevice));
    float* Wh;
    CK(hipMalloc(&Wh, (size_t)16 << 20));
    CK(launch_split_weights(W, Wh, (size_t)4 << 20, 4096.f, 0));  // |w| < 1 here
    float* Wh64;  // the same weights in the fragment order of the K = 64 row-stationary GEMM
    CK(hipMalloc(&Wh64, (size_t)1 << 20));
    CK(launch_split_weights(W, Wh64, (size_t)1 << 18, 4096.f, 0, true));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (size_t si = 0; si < shapes.size(); ++si) {
        if (only_shape >= 0 && (int)si != only_shape) continue;
        const Shape& s = shapes[si];
        IGemmArgs a;
        memset(&a, 0, sizeof(a));
        a.A = A; a.W = W; a.Wh = Wh; a.wscale = 1.0f / 4096.f; a.bias = bias; a.out = out; a.partial = partial; a.zeros = zeros; a.rows_per_img = 1;
        if (s.mode == A_CONV) {
            const int Ho = (s.Hi + 2 * s.pad - s.k) / s.stride + 1, Wo = (s.Wi + 2 * s.pad - s.k) / s.stride + 1;
            a.M = s.B * Ho * Wo; a.N = s.Cout; a.K = s.k * s.k * s.Cin; a.a_mode = A_CONV;
            a.Hi = s.Hi; a.Wi = s.Wi; a.Cin = s.Cin; a.Ho = Ho; a.Wo = Wo; a.KH = s.k; a.KW = s.k; a.stride = s.stride; a.pad = s.pad;
            a.res = res; a.ldr = s.Cout; a.act = 1; a.rows_per_img = Ho * Wo; a.st_mode = ST_ROW; a.ldc = s.Cout;
        } else {
            a.M = s.B * s.Hi * s.Wi; a.N = s.Cout; a.K = s.Cin; a.a_mode = A_GEMM; a.lda = s.Cin; a.ln = s.ln;
            a.st_mode = s.st_mode; a.ldc = s.Cout;
            a.qkv_C = s.Cin; a.qkv_hd = s.Cin / 4; a.qkv_L = s.Hi * s.Wi; a.qkv_heads = 4;
            a.qkv_plane = (long long)a.M * s.Cin;
            a.ct_C = s.Cout / 4; a.ct_H = s.Hi; a.ct_W = s.Wi;
        }
        if (a.K == 64) a.Wh = Wh64;
        const double flops = 2.0 * a.M * a.N * a.K;
        printf("%-26s M=%d N=%d K=%d  (%.2f GFLOP, %.1f us at 157 TF)\n", s.name, a.M, a.N, a.K, flops / 1e9, flops / 157.3e6);
        for (size_t ci = 0; ci < cfgs.size(); ++ci) {
            if (only_cfg >= 0 && (int)ci != only_cfg) continue;
            const Cfg& c = cfgs[ci];
            const int chunks = a.K / 32;
            if (c.ks > 1 && (chunks / c.ks < 4 || (size_t)c.ks * a.M * a.N > ((size_t)16 << 20))) continue;
            if (c.kw >= 4 && a.K % 64) continue;
            if (c.kw == 12 && !getenv("DANRA_TUNE_HALF")) continue;
            a.force_bm = c.bm; a.force_kw = c.kw; a.debug = debug;
            a.half = getenv("DANRA_TUNE_HALF") ? atoi(getenv("DANRA_TUNE_HALF")) : 0;
            a.ksplit = c.ks;
            int cps = (chunks + c.ks - 1) / c.ks;
            if (c.ks > 1) cps += cps & 1;
            a.chunks_per_split = cps;
            if (c.ks > 1) a.ksplit = (chunks + cps - 1) / cps;
            for (int i = 0; i < 3; ++i) CK(launch_igemm(a, 0));
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < iters; ++i) CK(launch_igemm(a, 0));
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / iters;
            printf("    bm=%3d kw=%d ks=%d  %8.1f us  %6.1f TF/s", c.bm, c.kw, a.ksplit, us, flops / us / 1e6);
            if (a.half && getenv("DANRA_TUNE_CHECK")) {  // the split-operand result against the fp32 instructions' result
                const size_t n_out = (size_t)a.M * a.N;
                std::vector<float> got(n_out), want(n_out);
                CK(hipMemcpy(got.data(), out, n_out * 4, hipMemcpyDeviceToHost));
                a.half = 0;
                CK(launch_igemm(a, 0));
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(want.data(), out, n_out * 4, hipMemcpyDeviceToHost));
                double err = 0, mx = 0;
                for (size_t i = 0; i < n_out; ++i) {
                    err = std::max(err, (double)fabsf(got[i] - want[i]));
                    mx = std::max(mx, (double)fabsf(want[i]));
                }
                printf("   max |half - fp32| = %.3e of max |out| = %.3e (%.2e)", err, mx, err / mx);
            }
            printf("\n");
            if (debug & 64) {  // clock readings of workgroup 0: per wave, mean cycles between the four stamps of an iteration
                const int waves = c.kw == 12 ? 12 : 4 * c.kw;
                std::vector<unsigned long long> ts((size_t)waves * 64 * 4);
                CK(hipMemcpy(ts.data(), partial, ts.size() * 8, hipMemcpyDeviceToHost));
                {
                    std::vector<unsigned long long> ph(16 * 8);
                    CK(hipMemcpy(ph.data(), (char*)partial + (size_t)16 * 64 * 4 * 8, ph.size() * 8, hipMemcpyDeviceToHost));
                    for (int w = 0; w < waves; w += (waves > 4 ? 4 : 1))
                        printf("      wave %2d phases: entry->pipeline %6llu  fill %6llu  loop %7llu  vote+reduce %6llu  epilogue %6llu ticks\n", w,
                               ph[w * 8 + 1] - ph[w * 8], ph[w * 8 + 2] - ph[w * 8 + 1], ph[w * 8 + 3] - ph[w * 8 + 2], ph[w * 8 + 4] - ph[w * 8 + 3],
                               ph[w * 8 + 5] - ph[w * 8 + 4]);
                }
                const int n_it = std::min(64, a.K / 64);
                for (int w = 0; w < waves; ++w) {
                    double d01 = 0, d12 = 0, d23 = 0, d30 = 0;
                    for (int it = 2; it + 1 < n_it; ++it) {
                        const unsigned long long* q = &ts[((size_t)w * 64 + it) * 4];
                        d01 += (double)(q[1] - q[0]); d12 += (double)(q[2] - q[1]); d23 += (double)(q[3] - q[2]); d30 += (double)(q[4] - q[3]);
                    }
                    const int n = n_it - 3;
                    printf("      wave %2d: s0->s1 %7.1f  s1->s2 %7.1f  s2->s3 %7.1f  s3->next s0 %7.1f   (iteration %7.1f ticks)\n", w, d01 / n, d12 / n,
                           d23 / n, d30 / n, (d01 + d12 + d23 + d30) / n);
                }
            }
        }
    }
    return 0;
}
