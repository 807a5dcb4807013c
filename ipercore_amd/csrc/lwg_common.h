// Shared device helpers for the LWG (Liquid Warping GAN) per-frame synthesis kernels - gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

#define LWG_WAVE 64

// Tuning constants.  The product library reads NO environment: every knob is compiled to its measured-best value (DESIGN.md
// records what the other settings gave); tools/labbuild.sh SRC.hip NAME -DLWG_xxx=v builds a variant library for tools/convlab.py /
// tools/bf16lab.py A/B runs.
#ifndef LWG_CONV_SPLIT_MAX_TILES
#define LWG_CONV_SPLIT_MAX_TILES 512   // conv_igemm.hip: launches with at least this many 64x64 tiles are never split over K
#endif
#ifndef LWG_CONV_SPLITK
#define LWG_CONV_SPLITK 1          // conv_igemm.hip: split-K of the small-M fp32 launches that hand in a workspace (0 = never)
#endif
#ifndef LWG_WINO_SPLITK
#define LWG_WINO_SPLITK 1          // conv_winograd.hip: split-K of the training launches that cover half the CUs or less and hand in a workspace (0 = never)
#endif
#ifndef LWG_CONV_SMALL_TILES
#define LWG_CONV_SMALL_TILES 300   // conv_igemm.hip: launches with fewer 128 x 128 tiles than this use 64 x 64 tiles
#endif
#ifndef LWG_WINO_PERSIST
#define LWG_WINO_PERSIST 1         // conv_winograd.hip / convt_winograd.hip: persistent workgroups (one per CU walks the blocks, the next block's first loads under the epilogue); 0 = one block per workgroup
#endif
#ifndef LWG_CONV_DEEP
#define LWG_CONV_DEEP 1            // conv_igemm.hip: the small-tile launches load two K-steps ahead (0 = one, as the large tiles)
#endif
#ifndef LWG_CONV_SPADE_SMALL
#define LWG_CONV_SPADE_SMALL 1     // conv_igemm.hip: small SPADE launches on 128 x 64 tiles (0 = 128 x 128 always)
#endif
#ifndef LWG_BF16_PW_TM
#define LWG_BF16_PW_TM 2           // conv_igemm_bf16.hip, pointwise kernel: 2 = two 8-wave workgroups per CU with 64-row wave tiles, 4 = one with 128-row tiles
#endif
#ifndef LWG_BF16_DMA_A
#define LWG_BF16_DMA_A 1           // conv_igemm_bf16.hip, linear kernel: the A operand by LDS-DMA (0 = through registers)
#endif
#ifndef LWG_BF16_TILE64
#define LWG_BF16_TILE64 0          // conv_igemm_bf16.hip, linear kernel: 0 = 128 x 64 tiles for launches under two 128 x 128 tiles per CU, 1 = always, -1 = never
#endif
#ifndef LWG_BF16_BIG
#define LWG_BF16_BIG 1             // conv_igemm_bf16.hip, linear kernel: the 8-wave 256 x 256 tile for N % 256 == 0 launches with >= one tile per CU
#endif
#ifndef LWG_SPLIT_PP
#define LWG_SPLIT_PP 2             // conv_igemm_split.hip: the 8-wave ping-pong kernel - 0 never, 1 whenever legal, 2 heuristic
#endif
#ifndef LWG_ATTN_OCC
#define LWG_ATTN_OCC 5             // lwb_attn.hip: waves per SIMD the fp32 attention kernel's registers are held to (5, 6 or 8)
#endif
#ifndef LWG_ATTNX_OCC
#define LWG_ATTNX_OCC 4            // lwb_attn_x.hip: waves per SIMD the x-form attention kernel's registers are held to
#endif
#ifndef LWG_ATTNX_OCC16
#define LWG_ATTNX_OCC16 4          // the same for its bf16 instantiations
#endif
#ifndef LWG_ATTNX_BVREG16
#define LWG_ATTNX_BVREG16 0        // bf16 instantiations: 1 = hold bv in registers (8 more VGPRs: spills at 4 waves / SIMD)
#endif
#ifndef LWG_ATTNX_NSU2
#define LWG_ATTNX_NSU2 1           // lwb_attn_x.hip: ns == 2 launches on the form with prefetched flows and an unrolled source loop (0: the generic form)
#endif
#ifndef LWG_ATTNX_PIPE
#define LWG_ATTNX_PIPE 0           // lwb_attn_x.hip: 1 = fp32 C = 256 frame batches on the form with two passes of gathers in flight + a tile-level background
                                   // shortcut (round 6 lab: bit-identical, 125 vs 114 us per 32 x 64^2 x 256 launch - SLOWER: a body tile is not a per-wave latency
                                   // chain; profiles/r06_o_attnlab_pipe.txt)
#endif
#ifndef LWG_ATTNX_WIDE
#define LWG_ATTNX_WIDE 1            // lwb_attn_x.hip: launches of a few frames run sixteen waves (four passes at a time) per tile (0: always four)
#endif
#ifndef LWG_ATTNX_INTERLEAVE
#define LWG_ATTNX_INTERLEAVE 1     // lwb_attn_x.hip: tiles dealt round-robin over the XCDs (0: a contiguous band of tile rows per XCD)
#endif
#ifndef LWG_ATTN16_PAIR
#define LWG_ATTN16_PAIR 0          // bf16_ops.hip: 1 = two sources in flight per wave (159 VGPRs, 3 waves / SIMD): measured SLOWER (111 vs 97 us at C = 256)
#endif
#ifndef LWG_ATTN16_OCC
#define LWG_ATTN16_OCC 5           // bf16_ops.hip: waves per SIMD the bf16 attention kernel's registers are held to (4, 5 or 6)
#endif
#ifndef LWG_HEAD16_NCB
#define LWG_HEAD16_NCB 2           // bf16_ops.hip: 16-pixel column blocks per tile row of the bf16 output head (2, 3 or 4)
#endif

// Activation codes shared by the C ABI (include/lwg_hip.h) and the kernels.
enum { LWG_ACT_NONE = 0, LWG_ACT_RELU = 1, LWG_ACT_TANH = 2, LWG_ACT_SIGMOID = 3 };
// With LWG_EPI_RESIDUAL only: y = res > 0 ? acc + bias : 0 - the ReLU backward of the layer that PRODUCED the conv's forward input,
// applied where a data gradient is written (training step: dX of a conv whose input was relu(...); res = that forward input).
#define LWG_ACT_RELU_MASK 5

__device__ __forceinline__ float lwg_act(float v, int act) {
    if (act == LWG_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == LWG_ACT_TANH) return tanhf(v);
    if (act == LWG_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
    return v;
}

// The activation of an epilogue resolved ONCE per workgroup instead of once per value (round 6): with the runtime code inside the innermost loops the
// compiler emits a uniform branch ladder around every group of outputs (the tanh / sigmoid bodies in between) - measured on the fused Winograd kernel:
// 4.5 k cycles per block for the output pass against 2.1 k with the activation a constant (profiles/r06_i_*).  An epilogue body is a generic lambda
// taking LwgActC<A>; A >= 0: compile-time activation, -1: a.act at run time (tanh / sigmoid / the ReLU mask of a data gradient).  Same expressions per
// value in every copy: same bits.
template <int A> struct LwgActC { static constexpr int value = A; };
template <int A> __device__ __forceinline__ float lwg_act_c(float v, int act_runtime) { return lwg_act(v, A >= 0 ? A : act_runtime); }
template <int A> __device__ __forceinline__ bool lwg_act_is_mask(int act_runtime) { return A == LWG_ACT_RELU_MASK || (A < 0 && act_runtime == LWG_ACT_RELU_MASK); }
template <typename F> __device__ __forceinline__ void lwg_act_dispatch(int act, F&& body) {
    if (act == LWG_ACT_RELU) body(LwgActC<LWG_ACT_RELU>());
    else if (act == LWG_ACT_NONE) body(LwgActC<LWG_ACT_NONE>());
    else if (act == LWG_ACT_RELU_MASK) body(LwgActC<LWG_ACT_RELU_MASK>());      // (the data gradients of the training step)
    else body(LwgActC<-1>());
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
// Work order of the per-pixel gather kernels (attention blocks): 8 x 8-pixel tiles, the B frames of a tile back to back, and - through
// lwg_xcd_remap on the block id - every XCD a contiguous range of that order.  The source texels (K / V) a tile's flows point at are then
// fetched into ONE XCD's L2 once for all frames of the batch; in frame-major order every frame streamed the whole K / V set (134 MB per
// site at 1024^2: larger than the L2s) from the Infinity Cache again.  L: position in that order; returns false for padding positions.
__device__ __forceinline__ bool lwg_tile_frame_pixel(long L, int B, int h, int w, int& b, int& y, int& x) {
    const int tiles_x = (w + 7) >> 3, tiles_y = (h + 7) >> 3;
    const long per_tile = (long)B * 64;
    const long tile = L / per_tile;
    const int r = (int)(L - tile * per_tile);
    b = r >> 6;
    const int p = r & 63;
    const int ty = (int)(tile / tiles_x), tx = (int)(tile - (long)ty * tiles_x);
    y = ty * 8 + (p >> 3);
    x = tx * 8 + (p & 7);
    return ty < tiles_y && y < h && x < w;
}
static inline long lwg_tile_frame_positions(int B, int h, int w) { return (long)((w + 7) >> 3) * ((h + 7) >> 3) * B * 64; }

// conv_igemm.hip: y = act(sum of the (M, N) slabs in slice order + bias) (| ReLU mask) - the finish of a split-K launch of either conv engine
struct LwgConvArgs;
hipError_t lwg_splitk_finish_launch(const LwgConvArgs& a, const float* ws, int slices, hipStream_t stream);

// compute units of the current device, cached per device id (a benign race: every thread stores the same value)
static inline int lwg_device_cus() {
    static int cus[64];
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (cus[dev] == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev] = n;
    }
    return cus[dev];
}

static inline hipError_t lwg_allow_dynamic_lds(const void* kern, size_t bytes, unsigned long long& done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = -1;
    if (dev >= 0 && dev < 64 && ((done >> dev) & 1ull)) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess && dev >= 0 && dev < 64) done |= 1ull << dev;
    return e;
}
