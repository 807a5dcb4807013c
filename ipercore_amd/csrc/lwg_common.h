// Shared device helpers for the LWG (Liquid Warping GAN) per-frame synthesis kernels - gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

#define LWG_WAVE 64

// Activation codes shared by the C ABI (include/lwg_hip.h) and the kernels.
enum { LWG_ACT_NONE = 0, LWG_ACT_RELU = 1, LWG_ACT_TANH = 2, LWG_ACT_SIGMOID = 3 };

__device__ __forceinline__ float lwg_act(float v, int act) {
    if (act == LWG_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == LWG_ACT_TANH) return tanhf(v);
    if (act == LWG_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
    return v;
}

// XCD-aware block remap (8 XCDs, blocks are dealt round-robin): logical ids that are adjacent share
// operand panels, so give every XCD one contiguous chunk of the logical id space.  Bijective for any nwg.
__device__ __forceinline__ int lwg_xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + idx;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// More than 64 KB of dynamic LDS is an opt-in per kernel AND per device: `done` holds one bit per device ordinal (one process may
// drive several GPUs even though the runners use one process per GPU).
static inline hipError_t lwg_allow_dynamic_lds(const void* kern, size_t bytes, unsigned long long& done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
    if (dev >= 0 && dev < 64 && ((done >> dev) & 1ull)) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess && dev >= 0 && dev < 64) done |= 1ull << dev;
    return e;
}
