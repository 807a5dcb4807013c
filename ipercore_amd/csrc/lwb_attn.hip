// Liquid Warping Block, attention form: flow resize + feature warp + per-pixel softmax over the sources.
// Replaces, per AttLWB site (9 per frame), the chain of reference
//   generators/attlwb_spade_resunet.py:175-182  LWB.resize_trans  (F.interpolate bilinear, align_corners=True)
//   :190                                        LWB.transform     (F.grid_sample bilinear/zeros/align_corners=False)
//   :226-227                                    fk / fv 1x1 convs on the warped source features
//   :121-139, :116-117                          SelfAttentionBlock.query + weighted sum
// A 1x1 conv commutes with the (linear, zero-padded) warp:  fk(warp(x)) = warp(Wk x) + bk  at every pixel,
// including out-of-range ones where the warp is 0.  Ks = Wk x and Vs = Wv x are therefore computed ONCE per
// source (they only depend on the cached source features) and this kernel gathers them: an HBM/L2-bound
// gather with 16-byte loads, one pixel per LPP = C/4 lanes, online softmax over the ns sources.

#include "lwg_common.h"
#include "lwg_conv_args.h"

// BUF: the K / V taps through raw buffer loads - an out-of-image tap gets an out-of-range offset and the hardware returns zeros, so the
// eight gathers of a source are unconditional and in flight together (with per-tap branches each load waited for its own exec mask
// and its own s_waitcnt); needs the K / V tensors below 3 GB (the launcher picks the pointer form otherwise).
// KVS: Ks / Vs rows are ``kvs`` floats apart instead of C (the two halves of ONE (nsrc,h,w,2C) tensor: the training step projects K | V
// with a single stacked 1x1 convolution, lwg_lwb_attention_kv_f32); false: dense tensors, the stride a compile-time constant.
template <int LPP, bool BUF, int OCC, bool KVS = false>       // OCC: resident waves per SIMD the register allocation is held to (see csrc/bf16_ops.hip)
__global__ __launch_bounds__(256, OCC) void lwg_lwb_attn_kernel(const float* __restrict__ q, const float* __restrict__ Ks,
                                                          const float* __restrict__ Vs, const float* __restrict__ bk,
                                                          const float* __restrict__ bv, const float* __restrict__ T,
                                                          float* __restrict__ out, int B, int ns, int h, int w, int S, int src_batched,
                                                          int kvs) {
    constexpr int C = 4 * LPP;
    const int PS = KVS ? kvs : C;                // floats between consecutive pixels of Ks / Vs
    constexpr int PPW = 64 / LPP;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int cl = lane % LPP;
    // tile-major, frame-minor work order (lwg_tile_frame_pixel): the frames of a batch share the L2-resident source texels
    const long L = ((long)lwg_xcd_remap(blockIdx.x, gridDim.x) * 4 + wid) * PPW + lane / LPP;
    int b, y, x;
    const bool live = lwg_tile_frame_pixel(L, B, h, w, b, y, x);
    if (!live) { b = 0; y = 0; x = 0; }  // keep all lanes in the shuffles
    const int hw = h * w;
    const long gp = ((long)b * h + y) * w + x;

    const floatx4 q4 = *reinterpret_cast<const floatx4*>(q + gp * C + 4 * cl);
    const floatx4 bk4 = *reinterpret_cast<const floatx4*>(bk + 4 * cl);
    const floatx4 bv4 = *reinterpret_cast<const floatx4*>(bv + 4 * cl);

    // flow resize S x S -> h x w, bilinear, align_corners=True (ATen area_pixel_compute_source_index)
    const float sc_y = h > 1 ? (float)(S - 1) / (float)(h - 1) : 0.f;
    const float sc_x = w > 1 ? (float)(S - 1) / (float)(w - 1) : 0.f;
    const float sy = sc_y * (float)y, sx = sc_x * (float)x;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < S - 1 ? 1 : 0), x1 = x0 + (x0 < S - 1 ? 1 : 0);
    const float ly1 = sy - (float)y0, lx1 = sx - (float)x0;
    const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
    const bool same = (h == S) && (w == S);

    const float inv_sqrt_c = 1.0f / sqrtf((float)C);
    float mrun = -INFINITY, lrun = 0.f;
    floatx4 o = {0.f, 0.f, 0.f, 0.f};

    for (int s = 0; s < ns; ++s) {
        const float2* Tp = reinterpret_cast<const float2*>(T) + ((size_t)b * ns + s) * S * S;
        float gx, gy;
        if (same) {
            const float2 t = Tp[(size_t)y * S + x];
            gx = t.x; gy = t.y;
        } else {
            const float2 t00 = Tp[(size_t)y0 * S + x0], t01 = Tp[(size_t)y0 * S + x1];
            const float2 t10 = Tp[(size_t)y1 * S + x0], t11 = Tp[(size_t)y1 * S + x1];
            gx = ly0 * (lx0 * t00.x + lx1 * t01.x) + ly1 * (lx0 * t10.x + lx1 * t11.x);
            gy = ly0 * (lx0 * t00.y + lx1 * t01.y) + ly1 * (lx0 * t10.y + lx1 * t11.y);
        }
        // grid_sample, align_corners=False: pixel = ((g + 1) * size - 1) / 2
        const float ix = ((gx + 1.f) * (float)w - 1.f) * 0.5f, iy = ((gy + 1.f) * (float)h - 1.f) * 0.5f;
        const float fx0 = floorf(ix), fy0 = floorf(iy);
        const float wx1 = ix - fx0, wy1 = iy - fy0, wx0 = (fx0 + 1.f) - ix, wy0 = (fy0 + 1.f) - iy;
        // clamp before the int conversion so wild flows cannot overflow; out-of-range taps are skipped anyway
        const int tx0 = (int)fminf(fmaxf(fx0, -2.f), (float)w + 1.f), ty0 = (int)fminf(fmaxf(fy0, -2.f), (float)h + 1.f);
        const size_t sidx = src_batched ? (size_t)b * ns + s : (size_t)s;
        const float* Kb = Ks + sidx * hw * PS + 4 * cl;
        const float* Vb = Vs + sidx * hw * PS + 4 * cl;
        floatx4 ka = {0.f, 0.f, 0.f, 0.f}, va = {0.f, 0.f, 0.f, 0.f};
        if (BUF) {
            const unsigned nsrc = (unsigned)(src_batched ? B * ns : ns);
            const int nbytes = (int)(((nsrc * (unsigned)hw - 1u) * (unsigned)PS + (unsigned)C) * 4u);     // up to the last pixel's C channels
            __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Ks), 0, nbytes, 0x00020000);
            __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Vs), 0, nbytes, 0x00020000);
            const unsigned sbase = ((unsigned)sidx * (unsigned)hw * (unsigned)PS + 4u * (unsigned)cl) * 4u;
            floatx4 k4[4], v4[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int ty = ty0 + (t >> 1), tx = tx0 + (t & 1);
                const bool ok = ty >= 0 && ty < h && tx >= 0 && tx < w;
                const unsigned voff = ok ? sbase + (unsigned)(ty * w + tx) * (unsigned)PS * 4u : 0xC0000000u;
                k4[t] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rK, (int)voff, 0, 0));
                v4[t] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rV, (int)voff, 0, 0));
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float wt = ((t >> 1) ? wy1 : wy0) * ((t & 1) ? wx1 : wx0);
#pragma unroll
                for (int k = 0; k < 4; ++k) { ka[k] += k4[t][k] * wt; va[k] += v4[t][k] * wt; }
            }
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int ty = ty0 + (t >> 1), tx = tx0 + (t & 1);
                const float wt = ((t >> 1) ? wy1 : wy0) * ((t & 1) ? wx1 : wx0);
                if (ty >= 0 && ty < h && tx >= 0 && tx < w) {
                    const size_t off = ((size_t)ty * w + tx) * PS;
                    const floatx4 k4 = *reinterpret_cast<const floatx4*>(Kb + off);
                    const floatx4 v4 = *reinterpret_cast<const floatx4*>(Vb + off);
#pragma unroll
                    for (int k = 0; k < 4; ++k) { ka[k] += k4[k] * wt; va[k] += v4[k] * wt; }
                }
            }
        }
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) dot += (ka[k] + bk4[k]) * q4[k];
#pragma unroll
        for (int off = LPP >> 1; off > 0; off >>= 1) dot += __shfl_xor(dot, off, 64);
        const float logit = dot * inv_sqrt_c;
        const float mnew = fmaxf(mrun, logit);
        const float corr = expf(mrun - mnew);  // exp(-inf) = 0 on the first source
        const float p = expf(logit - mnew);
        lrun = lrun * corr + p;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = o[k] * corr + p * (va[k] + bv4[k]);
        mrun = mnew;
    }
    if (live) {
        const float invl = 1.f / lrun;
        floatx4 r;
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = o[k] * invl;
        *reinterpret_cast<floatx4*>(out + gp * C + 4 * cl) = r;
    }
}

// src_batched = 0: Ks/Vs (ns,h,w,C) shared by all B frames; 1: (B*ns,h,w,C), frame b uses rows b*ns+s.
// q (B,h,w,C) = fq(tsf_x) incl. bias; Ks/Vs (ns,h,w,C) = Wk x_src / Wv x_src WITHOUT bias; bk/bv (C);
// The flow resize of LWB.resize_trans (attlwb_spade_resunet.py:175-181: F.interpolate(T, size=(h, w), mode="bilinear", align_corners=True))
// as its own pass: n = B * ns flow fields (S,S,2) -> (h,w,2).  The attention / fusion kernels do this resize per pixel when handed the
// full-resolution flows - four 8-byte loads per pixel and source at a stride of S / h pixels, each a trip to HBM at the head of the
// pixel's dependent chain (flow -> tap addresses -> K / V gathers -> softmax), and the same resize again at every site of the same
// resolution (seven of the nine sites share one).  Resized once per frame batch and resolution, the block kernels take the (h,w) field
// with S = h: one coalesced load.  Same formula as the in-kernel resize (results agree to ~1e-6: the compiler contracts the two copies
// differently); within a run every site uses the same resized field, so a frame stays independent of its batch.
__global__ void lwg_flow_resize_kernel(const float* __restrict__ T, int n, int S, int h, int w, float* __restrict__ out) {
    const long total = (long)n * h * w;
    const float sc_y = h > 1 ? (float)(S - 1) / (float)(h - 1) : 0.f;
    const float sc_x = w > 1 ? (float)(S - 1) / (float)(w - 1) : 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % w), y = (int)((i / w) % h);
        const long f = i / ((long)w * h);
        const float sy = sc_y * (float)y, sx = sc_x * (float)x;
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + (y0 < S - 1 ? 1 : 0), x1 = x0 + (x0 < S - 1 ? 1 : 0);
        const float ly1 = sy - (float)y0, lx1 = sx - (float)x0;
        const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
        const float2* Tp = reinterpret_cast<const float2*>(T) + (size_t)f * S * S;
        const float2 t00 = Tp[(size_t)y0 * S + x0], t01 = Tp[(size_t)y0 * S + x1];
        const float2 t10 = Tp[(size_t)y1 * S + x0], t11 = Tp[(size_t)y1 * S + x1];
        float2 r;
        r.x = ly0 * (lx0 * t00.x + lx1 * t01.x) + ly1 * (lx0 * t10.x + lx1 * t11.x);
        r.y = ly0 * (lx0 * t00.y + lx1 * t01.y) + ly1 * (lx0 * t10.y + lx1 * t11.y);
        reinterpret_cast<float2*>(out)[i] = r;
    }
}

extern "C" int lwg_flow_resize_f32(const float* T, int n, int S, int h, int w, float* out, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!T || !out || n <= 0 || S <= 0 || h <= 0 || w <= 0) return (int)hipErrorInvalidValue;
    const long total = (long)n * h * w;
    const long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(lwg_flow_resize_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, stream, T, n, S, h, w, out);
    return (int)hipGetLastError();
}

// T (B,ns,S,S,2) flows in grid_sample coordinates (-2 = background); out (B,h,w,C).
extern "C" int lwg_lwb_attention_f32(const float* q, const float* Ks, const float* Vs, const float* bk, const float* bv,
                                     const float* T, float* out, int B, int ns, int h, int w, int C, int S,
                                     int src_batched, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!q || !Ks || !Vs || !bk || !bv || !T || !out || B <= 0 || ns <= 0 || h <= 0 || w <= 0 || S <= 0)
        return (int)hipErrorInvalidValue;
    const long total = lwg_tile_frame_positions(B, h, w);
    const bool buf_ok = (unsigned long long)(src_batched ? B * ns : ns) * (unsigned long long)h * w * (unsigned long long)C * 4ull < 0xC0000000ull;
#define LWG_ATTN_LAUNCH(LPP)                                                                                      \
    {                                                                                                             \
        const long per_block = 4 * (64 / LPP);                                                                    \
        const dim3 grid((unsigned)((total + per_block - 1) / per_block));                                         \
        if (!buf_ok)                                                                                              \
            hipLaunchKernelGGL((lwg_lwb_attn_kernel<LPP, false, 5>), grid, dim3(256), 0, stream, q, Ks, Vs, bk, bv, T, out, B, ns, h, w, S, src_batched, 0); \
        else                                                                                                      \
            hipLaunchKernelGGL((lwg_lwb_attn_kernel<LPP, true, LWG_ATTN_OCC>), grid, dim3(256), 0, stream, q, Ks, Vs, bk, bv, T, out, B, ns, h, w, S, src_batched, 0);  \
    }
    switch (C) {
        case 32: LWG_ATTN_LAUNCH(8) break;
        case 64: LWG_ATTN_LAUNCH(16) break;
        case 128: LWG_ATTN_LAUNCH(32) break;
        case 256: LWG_ATTN_LAUNCH(64) break;
        default: return (int)hipErrorInvalidValue;
    }
#undef LWG_ATTN_LAUNCH
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// The same block with K | V as ONE tensor kv (nsrc,h,w,2C): K = kv[..., :C], V = kv[..., C:] (the training step's stacked fk | fv
// projection, networks/training.py attlwb): no slicing copies in front of the kernel.
extern "C" int lwg_lwb_attention_kv_f32(const float* q, const float* kv, const float* bk, const float* bv, const float* T, float* out,
                                        int B, int ns, int h, int w, int C, int S, int src_batched, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!q || !kv || !bk || !bv || !T || !out || B <= 0 || ns <= 0 || h <= 0 || w <= 0 || S <= 0) return (int)hipErrorInvalidValue;
    const long total = lwg_tile_frame_positions(B, h, w);
    const bool buf_ok = (unsigned long long)(src_batched ? B * ns : ns) * (unsigned long long)h * w * (unsigned long long)C * 8ull < 0xC0000000ull;
#define LWG_ATTN_KV_LAUNCH(LPP)                                                                                   \
    {                                                                                                             \
        const long per_block = 4 * (64 / LPP);                                                                    \
        const dim3 grid((unsigned)((total + per_block - 1) / per_block));                                         \
        if (!buf_ok)                                                                                              \
            hipLaunchKernelGGL((lwg_lwb_attn_kernel<LPP, false, 5, true>), grid, dim3(256), 0, stream, q, kv, kv + C, bk, bv, T, out, B, ns, h, w, S, src_batched, 2 * C); \
        else                                                                                                      \
            hipLaunchKernelGGL((lwg_lwb_attn_kernel<LPP, true, LWG_ATTN_OCC, true>), grid, dim3(256), 0, stream, q, kv, kv + C, bk, bv, T, out, B, ns, h, w, S, src_batched, 2 * C);  \
    }
    switch (C) {
        case 32: LWG_ATTN_KV_LAUNCH(8) break;
        case 64: LWG_ATTN_KV_LAUNCH(16) break;
        case 128: LWG_ATTN_KV_LAUNCH(32) break;
        case 256: LWG_ATTN_KV_LAUNCH(64) break;
        default: return (int)hipErrorInvalidValue;
    }
#undef LWG_ATTN_KV_LAUNCH
    return (int)hipGetLastError();
}

// Backward of the attention-form Liquid Warping Block (personalization step).  Per pixel, with K_s = warp_s(Ks) + bk,
// V_s = warp_s(Vs) + bv, l_s = K_s.q / sqrt(C), a = softmax_s(l), out = sum_s a_s V_s and upstream gradient g = dout:
//   dV_s = a_s g                      da_s = g . V_s            dl_s = a_s (da_s - sum_j a_j da_j)
//   dK_s = dl_s q / sqrt(C)           dq   = sum_s dl_s K_s / sqrt(C)
//   dKs[tap] += w_tap dK_s,  dVs[tap] += w_tap dV_s  (bilinear scatter: fp32 atomics - same as grid_sample's backward)
//   dbv = sum_pixels g  (sum_s a_s = 1);   dbk = 0 exactly (a bias on every K shifts all logits of a pixel equally)
// The flows are not differentiated (the reference computes them under no_grad, lwg_trainer.py:649-697).
// One pixel per LPP = C/4 lanes, the gathers are recomputed instead of stored.  dKs / dVs must be zero on entry.
template <int LPP, bool KVS = false>
__global__ __launch_bounds__(256) void lwg_lwb_attn_bwd_kernel(const float* __restrict__ q, const float* __restrict__ Ks,
                                                              const float* __restrict__ Vs, const float* __restrict__ bk,
                                                              const float* __restrict__ bv, const float* __restrict__ T,
                                                              const float* __restrict__ dout, float* __restrict__ dq,
                                                              float* __restrict__ dKs, float* __restrict__ dVs, int B, int ns, int h,
                                                              int w, int S, int src_batched, int kvs) {
    constexpr int C = 4 * LPP;
    const int PS = KVS ? kvs : C;                // floats between consecutive pixels of Ks / Vs / dKs / dVs
    constexpr int PPW = 64 / LPP;
    constexpr int MAXS = 8;   // sources + temporal frames per pixel kept in registers
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int cl = lane % LPP;
    const long total = (long)B * h * w;
    long gp = ((long)blockIdx.x * 4 + wid) * PPW + lane / LPP;
    const bool live = gp < total;
    if (!live) gp = total - 1;
    const int hw = h * w;
    const int b = (int)(gp / hw), rem = (int)(gp - (long)b * hw);
    const int y = rem / w, x = rem - y * w;

    const floatx4 q4 = *reinterpret_cast<const floatx4*>(q + gp * C + 4 * cl);
    const floatx4 g4 = *reinterpret_cast<const floatx4*>(dout + gp * C + 4 * cl);
    const floatx4 bk4 = *reinterpret_cast<const floatx4*>(bk + 4 * cl);
    const floatx4 bv4 = *reinterpret_cast<const floatx4*>(bv + 4 * cl);

    const float sc_y = h > 1 ? (float)(S - 1) / (float)(h - 1) : 0.f;
    const float sc_x = w > 1 ? (float)(S - 1) / (float)(w - 1) : 0.f;
    const float sy = sc_y * (float)y, sx = sc_x * (float)x;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < S - 1 ? 1 : 0), x1 = x0 + (x0 < S - 1 ? 1 : 0);
    const float ly1 = sy - (float)y0, lx1 = sx - (float)x0;
    const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
    const bool same = (h == S) && (w == S);
    const float inv_sqrt_c = 1.0f / sqrtf((float)C);

    float logit[MAXS], da[MAXS];
    floatx4 kf[MAXS];                 // K_s (this lane's 4 channels) for dq
    int tx0s[MAXS], ty0s[MAXS];
    float wx1s[MAXS], wy1s[MAXS], wx0s[MAXS], wy0s[MAXS];
    float mx = -INFINITY;
    for (int s = 0; s < ns && s < MAXS; ++s) {
        const float2* Tp = reinterpret_cast<const float2*>(T) + ((size_t)b * ns + s) * S * S;
        float gx, gy;
        if (same) {
            const float2 t = Tp[(size_t)y * S + x];
            gx = t.x; gy = t.y;
        } else {
            const float2 t00 = Tp[(size_t)y0 * S + x0], t01 = Tp[(size_t)y0 * S + x1];
            const float2 t10 = Tp[(size_t)y1 * S + x0], t11 = Tp[(size_t)y1 * S + x1];
            gx = ly0 * (lx0 * t00.x + lx1 * t01.x) + ly1 * (lx0 * t10.x + lx1 * t11.x);
            gy = ly0 * (lx0 * t00.y + lx1 * t01.y) + ly1 * (lx0 * t10.y + lx1 * t11.y);
        }
        const float ix = ((gx + 1.f) * (float)w - 1.f) * 0.5f, iy = ((gy + 1.f) * (float)h - 1.f) * 0.5f;
        const float fx0 = floorf(ix), fy0 = floorf(iy);
        wx1s[s] = ix - fx0; wy1s[s] = iy - fy0; wx0s[s] = (fx0 + 1.f) - ix; wy0s[s] = (fy0 + 1.f) - iy;
        tx0s[s] = (int)fminf(fmaxf(fx0, -2.f), (float)w + 1.f);
        ty0s[s] = (int)fminf(fmaxf(fy0, -2.f), (float)h + 1.f);
        const size_t sidx = src_batched ? (size_t)b * ns + s : (size_t)s;
        const float* Kb = Ks + sidx * hw * PS + 4 * cl;
        const float* Vb = Vs + sidx * hw * PS + 4 * cl;
        floatx4 ka = {0.f, 0.f, 0.f, 0.f}, va = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ty = ty0s[s] + (t >> 1), tx = tx0s[s] + (t & 1);
            const float wt = ((t >> 1) ? wy1s[s] : wy0s[s]) * ((t & 1) ? wx1s[s] : wx0s[s]);
            if (ty >= 0 && ty < h && tx >= 0 && tx < w) {
                const size_t off = ((size_t)ty * w + tx) * PS;
                const floatx4 k4 = *reinterpret_cast<const floatx4*>(Kb + off);
                const floatx4 v4 = *reinterpret_cast<const floatx4*>(Vb + off);
#pragma unroll
                for (int k = 0; k < 4; ++k) { ka[k] += k4[k] * wt; va[k] += v4[k] * wt; }
            }
        }
        float dot = 0.f, dav = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            kf[s][k] = ka[k] + bk4[k];
            dot += kf[s][k] * q4[k];
            dav += (va[k] + bv4[k]) * g4[k];
        }
#pragma unroll
        for (int off = LPP >> 1; off > 0; off >>= 1) {
            dot += __shfl_xor(dot, off, 64);
            dav += __shfl_xor(dav, off, 64);
        }
        logit[s] = dot * inv_sqrt_c;
        da[s] = dav;
        mx = fmaxf(mx, logit[s]);
    }
    float den = 0.f;
    for (int s = 0; s < ns && s < MAXS; ++s) { logit[s] = expf(logit[s] - mx); den += logit[s]; }
    float adot = 0.f;
    for (int s = 0; s < ns && s < MAXS; ++s) { logit[s] /= den; adot += logit[s] * da[s]; }   // logit[] now holds a_s
    floatx4 dq4 = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < ns && s < MAXS; ++s) {
        const float a = logit[s];
        const float dl = a * (da[s] - adot) * inv_sqrt_c;
#pragma unroll
        for (int k = 0; k < 4; ++k) dq4[k] += dl * kf[s][k];
        if (!live) continue;
        const size_t sidx = src_batched ? (size_t)b * ns + s : (size_t)s;
        float* dKb = dKs + sidx * hw * PS + 4 * cl;
        float* dVb = dVs + sidx * hw * PS + 4 * cl;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ty = ty0s[s] + (t >> 1), tx = tx0s[s] + (t & 1);
            const float wt = ((t >> 1) ? wy1s[s] : wy0s[s]) * ((t & 1) ? wx1s[s] : wx0s[s]);
            if (ty >= 0 && ty < h && tx >= 0 && tx < w) {
                const size_t off = ((size_t)ty * w + tx) * PS;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    atomicAdd(dKb + off + k, wt * dl * q4[k]);
                    atomicAdd(dVb + off + k, wt * a * g4[k]);
                }
            }
        }
    }
    if (live) *reinterpret_cast<floatx4*>(dq + gp * C + 4 * cl) = dq4;
}

// dq (B,h,w,C) is written; dKs / dVs (shaped like Ks / Vs) are ACCUMULATED into (zero them first); ns <= 8.
extern "C" int lwg_lwb_attention_bwd_f32(const float* q, const float* Ks, const float* Vs, const float* bk, const float* bv,
                                         const float* T, const float* dout, float* dq, float* dKs, float* dVs, int B, int ns, int h,
                                         int w, int C, int S, int src_batched, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!q || !Ks || !Vs || !bk || !bv || !T || !dout || !dq || !dKs || !dVs || B <= 0 || ns <= 0 || ns > 8 || h <= 0 || w <= 0 || S <= 0)
        return (int)hipErrorInvalidValue;
    const long total = (long)B * h * w;
#define LWG_ATTN_BWD_LAUNCH(LPP)                                                                                      \
    {                                                                                                                 \
        const long per_block = 4 * (64 / LPP);                                                                        \
        hipLaunchKernelGGL(lwg_lwb_attn_bwd_kernel<LPP>, dim3((unsigned)((total + per_block - 1) / per_block)), dim3(256), 0, \
                           stream, q, Ks, Vs, bk, bv, T, dout, dq, dKs, dVs, B, ns, h, w, S, src_batched, 0);                    \
    }
    switch (C) {
        case 32: LWG_ATTN_BWD_LAUNCH(8) break;
        case 64: LWG_ATTN_BWD_LAUNCH(16) break;
        case 128: LWG_ATTN_BWD_LAUNCH(32) break;
        case 256: LWG_ATTN_BWD_LAUNCH(64) break;
        default: return (int)hipErrorInvalidValue;
    }
#undef LWG_ATTN_BWD_LAUNCH
    return (int)hipGetLastError();
}

// Backward with K | V as one tensor kv (nsrc,h,w,2C): d(kv) is ACCUMULATED into dkv of the same shape (zero it first).
extern "C" int lwg_lwb_attention_kv_bwd_f32(const float* q, const float* kv, const float* bk, const float* bv, const float* T,
                                            const float* dout, float* dq, float* dkv, int B, int ns, int h, int w, int C, int S,
                                            int src_batched, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!q || !kv || !bk || !bv || !T || !dout || !dq || !dkv || B <= 0 || ns <= 0 || ns > 8 || h <= 0 || w <= 0 || S <= 0)
        return (int)hipErrorInvalidValue;
    const long total = (long)B * h * w;
#define LWG_ATTN_KV_BWD_LAUNCH(LPP)                                                                                   \
    {                                                                                                                 \
        const long per_block = 4 * (64 / LPP);                                                                        \
        hipLaunchKernelGGL((lwg_lwb_attn_bwd_kernel<LPP, true>), dim3((unsigned)((total + per_block - 1) / per_block)), dim3(256), 0, \
                           stream, q, kv, kv + C, bk, bv, T, dout, dq, dkv, dkv + C, B, ns, h, w, S, src_batched, 2 * C);             \
    }
    switch (C) {
        case 32: LWG_ATTN_KV_BWD_LAUNCH(8) break;
        case 64: LWG_ATTN_KV_BWD_LAUNCH(16) break;
        case 128: LWG_ATTN_KV_BWD_LAUNCH(32) break;
        case 256: LWG_ATTN_KV_BWD_LAUNCH(64) break;
        default: return (int)hipErrorInvalidValue;
    }
#undef LWG_ATTN_KV_BWD_LAUNCH
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// The non-attention Liquid Warping Blocks of the reference's other generators:
//   AddLWB / AvgLWB      generators/lwb_resunet.py:77-152            out = sum | mean over [tsf_x, warp_1(src), .., warp_ns(src)]
//   SoftGateLWB          generators/lwb_softgate_resunet.py:77-123   out = tsf_x + gate * (sum | mean over the warped sources)
// as one gather kernel:   out = (tsf_x + gate * scale_w * sum_s warp_s(src_x)) * scale_o      (gate == nullptr: 1)
// Same flow resize / grid_sample conventions, lane mapping and 16-byte gathers as lwg_lwb_attn_kernel.
template <int LPP>
__global__ __launch_bounds__(256) void lwg_lwb_fuse_kernel(const float* __restrict__ tsf, const float* __restrict__ src,
                                                          const float* __restrict__ gate, const float* __restrict__ T,
                                                          float* __restrict__ out, int B, int ns, int h, int w, int S, int src_batched,
                                                          float scale_w, float scale_o) {
    constexpr int C = 4 * LPP;
    constexpr int PPW = 64 / LPP;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int cl = lane % LPP;
    const long total = (long)B * h * w;
    const long gp = ((long)blockIdx.x * 4 + wid) * PPW + lane / LPP;
    if (gp >= total) return;
    const int hw = h * w;
    const int b = (int)(gp / hw), rem = (int)(gp - (long)b * hw);
    const int y = rem / w, x = rem - y * w;

    const float sc_y = h > 1 ? (float)(S - 1) / (float)(h - 1) : 0.f;
    const float sc_x = w > 1 ? (float)(S - 1) / (float)(w - 1) : 0.f;
    const float sy = sc_y * (float)y, sx = sc_x * (float)x;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < S - 1 ? 1 : 0), x1 = x0 + (x0 < S - 1 ? 1 : 0);
    const float ly1 = sy - (float)y0, lx1 = sx - (float)x0;
    const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
    const bool same = (h == S) && (w == S);

    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < ns; ++s) {
        const float2* Tp = reinterpret_cast<const float2*>(T) + ((size_t)b * ns + s) * S * S;
        float gx, gy;
        if (same) {
            const float2 t = Tp[(size_t)y * S + x];
            gx = t.x; gy = t.y;
        } else {
            const float2 t00 = Tp[(size_t)y0 * S + x0], t01 = Tp[(size_t)y0 * S + x1];
            const float2 t10 = Tp[(size_t)y1 * S + x0], t11 = Tp[(size_t)y1 * S + x1];
            gx = ly0 * (lx0 * t00.x + lx1 * t01.x) + ly1 * (lx0 * t10.x + lx1 * t11.x);
            gy = ly0 * (lx0 * t00.y + lx1 * t01.y) + ly1 * (lx0 * t10.y + lx1 * t11.y);
        }
        const float ix = ((gx + 1.f) * (float)w - 1.f) * 0.5f, iy = ((gy + 1.f) * (float)h - 1.f) * 0.5f;
        const float fx0 = floorf(ix), fy0 = floorf(iy);
        const float wx1 = ix - fx0, wy1 = iy - fy0, wx0 = (fx0 + 1.f) - ix, wy0 = (fy0 + 1.f) - iy;
        const int tx0 = (int)fminf(fmaxf(fx0, -2.f), (float)w + 1.f), ty0 = (int)fminf(fmaxf(fy0, -2.f), (float)h + 1.f);
        const size_t sidx = src_batched ? (size_t)b * ns + s : (size_t)s;
        const float* Sb = src + sidx * hw * C + 4 * cl;
        floatx4 wv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ty = ty0 + (t >> 1), tx = tx0 + (t & 1);
            const float wt = ((t >> 1) ? wy1 : wy0) * ((t & 1) ? wx1 : wx0);
            if (ty >= 0 && ty < h && tx >= 0 && tx < w) {
                const floatx4 v4 = *reinterpret_cast<const floatx4*>(Sb + ((size_t)ty * w + tx) * C);
#pragma unroll
                for (int k = 0; k < 4; ++k) wv[k] += v4[k] * wt;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] += wv[k];       // the reference sums the warped sources one after the other
    }
    const floatx4 t4 = *reinterpret_cast<const floatx4*>(tsf + gp * C + 4 * cl);
    floatx4 g4 = {1.f, 1.f, 1.f, 1.f};
    if (gate) g4 = *reinterpret_cast<const floatx4*>(gate + gp * C + 4 * cl);
    floatx4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = (t4[k] + g4[k] * (acc[k] * scale_w)) * scale_o;
    *reinterpret_cast<floatx4*>(out + gp * C + 4 * cl) = r;
}

// tsf_x / gate / out (B,h,w,C); src_x (ns,h,w,C) or (B*ns,h,w,C); T (B,ns,S,S,2); gate may be NULL; C in {32,64,128,256}.
extern "C" int lwg_lwb_fuse_f32(const float* tsf_x, const float* src_x, const float* gate, const float* T, float* out, int B, int ns,
                                int h, int w, int C, int S, int src_batched, float scale_w, float scale_o, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!tsf_x || !src_x || !T || !out || B <= 0 || ns <= 0 || h <= 0 || w <= 0 || S <= 0) return (int)hipErrorInvalidValue;
    const long total = (long)B * h * w;
#define LWG_FUSE_LAUNCH(LPP)                                                                                          \
    {                                                                                                                 \
        const long per_block = 4 * (64 / LPP);                                                                        \
        hipLaunchKernelGGL(lwg_lwb_fuse_kernel<LPP>, dim3((unsigned)((total + per_block - 1) / per_block)), dim3(256), 0, stream, \
                           tsf_x, src_x, gate, T, out, B, ns, h, w, S, src_batched, scale_w, scale_o);                            \
    }
    switch (C) {
        case 32: LWG_FUSE_LAUNCH(8) break;
        case 64: LWG_FUSE_LAUNCH(16) break;
        case 128: LWG_FUSE_LAUNCH(32) break;
        case 256: LWG_FUSE_LAUNCH(64) break;
        default: return (int)hipErrorInvalidValue;
    }
#undef LWG_FUSE_LAUNCH
    return (int)hipGetLastError();
}
