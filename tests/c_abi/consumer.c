/* A plain-C consumer of libdanra_hip.so: no Python, no torch, only include/danra_hip.h and the HIP runtime for
 * device memory.  It builds the conditional network with synthetic weights, runs one UNet evaluation twice
 * (bit-identical), then checks that danra_sample with caller-supplied noise (T = 4) equals the same three steps
 * composed by hand from danra_unet_forward + danra_ddpm_update, bit for bit.  Prints "OK" and exits 0 on success.
 * Built by tests/c_abi/Makefile with gcc; run on the GPU by tests/test_gpu_c_consumer.py. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "danra_hip.h"

#define CHECK_HIP(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { \
    fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); return 2; } } while (0)
#define CHECK_RC(e) do { int r_ = (e); if (r_ != DANRA_OK) { \
    fprintf(stderr, "%s -> %d: %s\n", #e, r_, danra_last_error()); return 3; } } while (0)

static uint32_t rng_state = 12345u;
static float frand(void) { /* xorshift, uniform in (-1, 1) */
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 17; rng_state ^= rng_state << 5;
    return (float)(rng_state >> 8) * (2.0f / 16777216.0f) - 1.0f;
}

static float* device_fill(size_t n, float scale, float offset) {
    float* h = (float*)malloc(n * sizeof(float));
    float* d = NULL;
    for (size_t i = 0; i < n; ++i) h[i] = offset + scale * frand();
    if (hipMalloc((void**)&d, n * sizeof(float)) != hipSuccess) { free(h); return NULL; }
    if (hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { free(h); return NULL; }
    free(h);
    return d;
}

static int ends_with(const char* s, const char* suffix) {
    size_t a = strlen(s), b = strlen(suffix);
    return a >= b && strcmp(s + a - b, suffix) == 0;
}

int main(void) {
    enum { B = 3, H = 32, W = 64, T = 4 };
    const size_t n = (size_t)B * H * W;
    danra_config_t cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.has_lsm = cfg.has_topo = cfg.has_cond = 1;
    cfg.num_classes = 4; cfg.time_embedding = 256; cfg.out_channels = 1; cfg.n_heads = 4;
    for (int i = 0; i < 4; ++i) cfg.block_layers[i] = 2;
    danra_handle_t h = NULL;
    CHECK_RC(danra_create(&cfg, &h));

    /* parameters in their native PyTorch layout: scale ~ 1/sqrt(fan_in); norm scales / variances around 1 */
    const int np = danra_num_params(h);
    float** params = (float**)calloc((size_t)np, sizeof(float*));
    for (int i = 0; i < np; ++i) {
        const char* name = danra_param_name(h, i);
        const size_t numel = danra_param_numel(h, i);
        float scale = 0.05f, offset = 0.0f;
        if (ends_with(name, "running_var")) { scale = 0.25f; offset = 1.0f; }
        else if (strstr(name, "bn") && ends_with(name, ".weight")) { scale = 0.1f; offset = 1.0f; }
        else if (strstr(name, "downsample.1.weight") || strstr(name, "layernorm.weight")) { scale = 0.1f; offset = 1.0f; }
        else if (numel >= 4096) scale = 1.0f / sqrtf((float)(numel / 64));
        params[i] = device_fill(numel, scale, offset);
        if (!params[i]) { fprintf(stderr, "allocation of %s failed\n", name); return 2; }
        CHECK_RC(danra_set_param(h, name, params[i], numel));
    }
    CHECK_RC(danra_pack_weights(h, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    for (int i = 0; i < np; ++i) CHECK_HIP(hipFree(params[i]));  /* borrowed for packing only */
    free(params);

    float* x = device_fill(n, 1.0f, 0.0f);
    float* cond = device_fill(n, 6.0f, 8.0f);
    float* lsm = device_fill(n, 0.5f, 0.5f);
    float* topo = device_fill(n, 170.0f, 160.0f);
    float* noise = device_fill((size_t)(T - 2) * n, 1.0f, 0.0f);
    float *eps1, *eps2, *xa, *xb;
    int64_t *t_dev, *y_dev;
    CHECK_HIP(hipMalloc((void**)&eps1, n * 4)); CHECK_HIP(hipMalloc((void**)&eps2, n * 4));
    CHECK_HIP(hipMalloc((void**)&xa, n * 4)); CHECK_HIP(hipMalloc((void**)&xb, n * 4));
    CHECK_HIP(hipMalloc((void**)&t_dev, B * 8)); CHECK_HIP(hipMalloc((void**)&y_dev, B * 8));
    int64_t y_host[B] = {0, 3, 1}, t_host[B];
    CHECK_HIP(hipMemcpy(y_dev, y_host, sizeof(y_host), hipMemcpyHostToDevice));

    /* one evaluation, twice */
    for (int b = 0; b < B; ++b) t_host[b] = 2;
    CHECK_HIP(hipMemcpy(t_dev, t_host, sizeof(t_host), hipMemcpyHostToDevice));
    CHECK_RC(danra_unet_forward(h, x, t_dev, y_dev, cond, lsm, topo, eps1, B, H, W, NULL));
    CHECK_RC(danra_unet_forward(h, x, t_dev, y_dev, cond, lsm, topo, eps2, B, H, W, NULL));
    float* ha = (float*)malloc(n * 4);
    float* hb = (float*)malloc(n * 4);
    CHECK_HIP(hipMemcpy(ha, eps1, n * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(hb, eps2, n * 4, hipMemcpyDeviceToHost));
    double sum = 0.0;
    for (size_t i = 0; i < n; ++i) {
        if (!isfinite(ha[i])) { fprintf(stderr, "non-finite eps at %zu\n", i); return 4; }
        sum += fabs(ha[i]);
    }
    if (memcmp(ha, hb, n * 4) != 0 || sum == 0.0) { fprintf(stderr, "forward not reproducible / all zero\n"); return 4; }

    /* the loop, and the same loop by hand: linear schedule, coefficients in fp32 as the sampler forms them */
    float betas[T], alphas[T], alpha_hat[T];
    for (int i = 0; i < T; ++i) {
        betas[i] = 1e-4f + (0.02f - 1e-4f) * (float)i / (float)(T - 1);
        alphas[i] = 1.0f - betas[i];
        alpha_hat[i] = (i ? alpha_hat[i - 1] : 1.0f) * alphas[i];
    }
    CHECK_HIP(hipMemcpy(xa, x, n * 4, hipMemcpyDeviceToDevice));
    CHECK_HIP(hipMemcpy(xb, x, n * 4, hipMemcpyDeviceToDevice));
    CHECK_RC(danra_sample(h, xa, y_dev, cond, lsm, topo, B, H, W, T, betas, alphas, alpha_hat, DANRA_NOISE_HOST, noise, 0,
                          NULL, 0, NULL, NULL));
    for (int i = T - 1; i >= 1; --i) {
        for (int b = 0; b < B; ++b) t_host[b] = i;
        CHECK_HIP(hipMemcpy(t_dev, t_host, sizeof(t_host), hipMemcpyHostToDevice));
        CHECK_RC(danra_unet_forward(h, xb, t_dev, y_dev, cond, lsm, topo, eps1, B, H, W, NULL));
        const float c1 = 1.0f / sqrtf(alphas[i]);
        const float c2 = (1.0f - alphas[i]) / sqrtf(1.0f - alpha_hat[i]);
        const float c3 = sqrtf(betas[i]);
        CHECK_RC(danra_ddpm_update(xb, eps1, i > 1 ? noise + (size_t)(T - 1 - i) * n : NULL, c1, c2, c3, n, NULL));
    }
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipMemcpy(ha, xa, n * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(hb, xb, n * 4, hipMemcpyDeviceToHost));
    if (memcmp(ha, hb, n * 4) != 0) { fprintf(stderr, "danra_sample differs from the hand-composed loop\n"); return 5; }

    /* error convention: a bad shape is refused with a message, nothing is thrown */
    if (danra_unet_forward(h, x, t_dev, y_dev, cond, lsm, topo, eps1, B, 30, W, NULL) != DANRA_EINVAL ||
        strstr(danra_last_error(), "multiples of 32") == NULL) { fprintf(stderr, "bad-shape call not refused\n"); return 6; }
    CHECK_RC(danra_destroy(h));
    printf("OK params=%d mean|eps|=%.6f\n", np, sum / (double)n);
    return 0;
}
