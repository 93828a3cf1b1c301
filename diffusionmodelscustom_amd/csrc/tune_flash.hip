// Developer tool (not part of libdanra_hip.so): times the head_dim-16 flash attention kernel and its
// timing ablations (see the ABL template parameter) with HIP events on random data.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "kernels_attn.hip"

using namespace danra;

#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"

template <int ABL, bool PVH = true, bool QKH = PVH, int QT = 2, int KT64 = 1>
static void run(const char* name, const float* q, const float* k, const float* v, float* o, int B, int L) {
    const int heads = 4;
    const float sc = 0.25f * 1.4426950408889634f;
    dim3 grid((L / (64 * QT)) * B * heads);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_flash_hd16<QT, ABL, PVH, QKH, KT64>), grid, dim3(256), 0, 0, q, k, v, o, nullptr, L, heads, B * heads, sc, FlashPre{nullptr, nullptr, nullptr});
    hipEventRecord(e0, 0);
    const int iters = 10;
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k_flash_hd16<QT, ABL, PVH, QKH, KT64>), grid, dim3(256), 0, 0, q, k, v, o, nullptr, L, heads, B * heads, sc, FlashPre{nullptr, nullptr, nullptr});
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / iters;
    const double fl = 4.0 * B * heads * (double)L * L * 16;
    printf("%-34s %8.1f us  %6.1f TF/s (nominal)\n", name, us, fl / us / 1e6);
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64, L = argc > 2 ? atoi(argv[2]) : 1024;
    const float data_scale = argc > 3 ? (float)atof(argv[3]) : 1.f;  // > 1: strongly peaked rows, frequent reference moves
    const size_t n = (size_t)B * L * 64;
    float *q, *k, *v, *o;
    hipMalloc(&q, n * 4); hipMalloc(&k, n * 4); hipMalloc(&v, n * 4); hipMalloc(&o, n * 4);
    std::vector<float> h(n);
    srand(2);
    for (auto& x : h) x = data_scale * (float)(rand() % 2001 - 1000) / 500.f;
    hipMemcpy(q, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(k, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(v, h.data(), n * 4, hipMemcpyHostToDevice);
    run<0>("full", q, k, v, o, B, L);
    run<0, true, true, 2, 2>("full, 128 keys per barrier", q, k, v, o, B, L);
    run<0>("full", q, k, v, o, B, L);
    run<0, true, true, 2, 2>("full, 128 keys per barrier", q, k, v, o, B, L);
    run<0, true, true, 4>("full, 64 queries per wave", q, k, v, o, B, L);
    run<0, true, true, 1>("full, 16 queries per wave", q, k, v, o, B, L);
    {   // the fp16-halves product against the all-fp32 product on the same data
        float* o2;
        hipMalloc(&o2, n * 4);
        run<0>("full again", q, k, v, o2, B, L);
        {
            std::vector<float> a(n), b(n);
            hipMemcpy(a.data(), o, n * 4, hipMemcpyDeviceToHost);
            hipMemcpy(b.data(), o2, n * 4, hipMemcpyDeviceToHost);
            size_t diff = 0;
            for (size_t i = 0; i < n; ++i) diff += (a[i] != b[i]);
            printf("two runs of the same kernel differ in %zu values\n", diff);
        }
        run<0, false>("full, PV in fp32", q, k, v, o2, B, L);
        std::vector<float> a(n), b(n);
        hipMemcpy(a.data(), o, n * 4, hipMemcpyDeviceToHost);
        hipMemcpy(b.data(), o2, n * 4, hipMemcpyDeviceToHost);
        double md = 0, mx = 0;
        size_t bad = 0, where = 0;
        for (size_t i = 0; i < n; ++i) {
            const double d = fabs((double)a[i] - b[i]);
            if (!(d <= md)) { md = d; where = i; }
            if (d > 1e-4) ++bad;
            if (fabs(b[i]) > mx) mx = fabs(b[i]);
        }
        printf("largest difference at image %zu token %zu channel %zu: %.6f against %.6f; %zu of %zu differ by more than 1e-4\n",
               where / ((size_t)L * 64), (where / 64) % L, where % 64, a[where], b[where], bad, n);
        printf("max |fp16-halves - fp32| = %.3e  (max |out| %.3e, ratio %.2e)\n", md, mx, md / mx);
        hipFree(o2);
    }
    {   // K and V pre-split by their producer: the copy kernel, then the core on its output
        float *ks, *vs, *o3;
        int* fl;
        hipMalloc(&ks, n * 4); hipMalloc(&vs, n * 4); hipMalloc(&o3, n * 4); hipMalloc(&fl, (size_t)B * 4 * (L / 16) * 2 * 4);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            launch_presplit_kv(k, v, ks, vs, fl, B * 4, L, 0);
            hipEventRecord(e0, 0);
            for (int i = 0; i < 10; ++i) launch_attention_hd16_pre(q, ks, vs, k, v, fl, o3, B, 4, L, 0);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("%-34s %8.1f us\n", "full, K and V pre-split", ms * 100.0);
        }
        run<0>("full again", q, k, v, o, B, L);
        std::vector<float> a(n), b(n);
        hipMemcpy(a.data(), o, n * 4, hipMemcpyDeviceToHost);
        hipMemcpy(b.data(), o3, n * 4, hipMemcpyDeviceToHost);
        double md = 0, mx = 0;
        size_t diff = 0;
        for (size_t i = 0; i < n; ++i) { md = fmax(md, fabs((double)a[i] - b[i])); mx = fmax(mx, fabs((double)a[i])); diff += a[i] != b[i]; }
        printf("pre-split against in-kernel split: max difference %.3e of %.3e, %zu of %zu values differ\n", md, mx, diff, n);
    }
    run<1>("no exp2", q, k, v, o, B, L);
    run<2>("no V LDS reads", q, k, v, o, B, L);
    run<4>("no staging/barrier", q, k, v, o, B, L);
    run<8>("no QK^T MFMA", q, k, v, o, B, L);
    run<16>("no PV MFMA", q, k, v, o, B, L);
    run<24>("no MFMA at all", q, k, v, o, B, L);
    run<7>("MFMA + max/sum only", q, k, v, o, B, L);
    return 0;
}
