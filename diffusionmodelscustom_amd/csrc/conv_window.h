// What the window-staged convolution kernels share (device code only): vector types, the LDS pixel format, the split of
// four staged values, the epilogue activation and the LDS-only barrier.
#pragma once
#include "danra_kernels.h"
#include "split_f16.h"

namespace danra {
namespace convwin {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

// bytes of a window pixel in LDS when CBS 64-channel blocks are staged together: per block 64 hi halves then 64 lo
// halves, and 16 bytes at the end so that consecutive pixels rotate through the banks
constexpr int pixel_bytes(int cbs) { return cbs * 256 + 16; }

__device__ __forceinline__ void split4w(const v4f v, f16x4& hi, f16x4& lo) {
    unsigned h[2], l[2];
    split_pair_f16(v.x, v.y, h[0], l[0]);
    split_pair_f16(v.z, v.w, h[1], l[1]);
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    hi = __builtin_bit_cast(f16x4, (u32x2){h[0], h[1]});
    lo = __builtin_bit_cast(f16x4, (u32x2){l[0], l[1]});
}

__device__ __forceinline__ float act_apply(int act, float v) {
    if (act == ACT_RELU) return fmaxf(v, 0.f);
    if (act == ACT_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
    return v;
}

// Workgroup barrier for LDS traffic only: waits for this wave's LDS operations, not for its global loads.
// (__syncthreads() also waits for every outstanding global load - vmcnt(0) - which would drain the weight ring and the
// window loads that are meant to stay in flight across the barrier.)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

}  // namespace convwin

// Timing build of the developer harnesses (tune_conv_win, tune_conv_s2: -DDANRA_WIN_DEBUG=1): workgroup 0 leaves clock
// readings of its phases, per wave, in g_win_stamps (defined in kernels_conv_win.hip).  Compiled out of the library.
#ifndef DANRA_WIN_DEBUG
#define DANRA_WIN_DEBUG 0
#endif
#if DANRA_WIN_DEBUG
extern __device__ unsigned long long g_win_stamps[8 * 16];
#define WIN_STAMP(k)                                                                        \
    do {                                                                                    \
        if (blockIdx.x == 0 && blockIdx.y == 0) {                                           \
            __builtin_amdgcn_sched_barrier(0);                                              \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                     \
            if (lane == 0) g_win_stamps[wave * 16 + (k)] = t_;                              \
            __builtin_amdgcn_sched_barrier(0);                                              \
        }                                                                                   \
    } while (0)
#else
#define WIN_STAMP(k) do {} while (0)
#endif

}  // namespace danra
