// The HBM-bound kernels of the per-frame path on bf16 activations (BASELINE configs[3]: bf16 activation storage end to end):
// the attention-form Liquid Warping Block and the output head + compositing (the InstanceNorm statistics live in norm.hip).  Same algorithms and
// reference citations as their fp32 twins (csrc/norm.hip, csrc/lwb_attn.hip, csrc/head.hip); arithmetic stays fp32, only the
// tensors in HBM are bf16: half the bytes per launch, 16-byte accesses carry 8 channels.

#include "lwg_common.h"
#include "lwg_conv_args.h"

typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float lwg_b2f_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float lwg_b2f_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
__device__ __forceinline__ void lwg_unpack8(const uintx4 v, float (&f)[8]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f[2 * k] = lwg_b2f_lo(v[k]);
        f[2 * k + 1] = lwg_b2f_hi(v[k]);
    }
}
__device__ __forceinline__ unsigned lwg_pack2(float lo, float hi) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned, v);
}

// ---------------------------------------------------------------------------------------------- Liquid Warping Block (attention)
// csrc/lwb_attn.hip lwg_lwb_attn_kernel on bf16 q / Ks / Vs / out: one pixel per LPP = C/8 lanes, 16-byte gathers of 8 channels,
// fp32 flows, fp32 online softmax.
// BUF: as in lwg_lwb_attn_kernel - zero-filling buffer loads for the taps; PAIR: two sources in flight; OCC: resident waves per SIMD the
// register allocation is held to (the kernel runs on the number of pixel chains in flight)
template <int LPP, bool BUF, bool PAIR, int OCC>
__global__ __launch_bounds__(256, OCC) void lwg_lwb_attn_bf16_kernel(const __bf16* __restrict__ q, const __bf16* __restrict__ Ks,
                                                               const __bf16* __restrict__ Vs, const float* __restrict__ bk,
                                                               const float* __restrict__ bv, const float* __restrict__ T,
                                                               __bf16* __restrict__ out, int B, int ns, int h, int w, int S, int src_batched) {
    constexpr int C = 8 * LPP;
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

    float q8[8];
    lwg_unpack8(*reinterpret_cast<const uintx4*>(q + gp * C + 8 * cl), q8);
    // the biases leave the loop (registers = resident waves = pixel chains in flight, which is what this kernel runs on):
    // sum_k (K_k + bk_k) q_k = sum_k K_k q_k + qbk, and sum_s a_s (V_s + bv) = sum_s a_s V_s + bv because the a_s sum to one
    float qbk = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) qbk += bk[8 * cl + k] * q8[k];

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
    float o[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = 0.f;

    // one source: the flow of this pixel -> sampling position (grid_sample, align_corners=False) -> corner tap + bilinear weights
    auto position = [&](int s, int& tx0, int& ty0, float& wx0, float& wx1, float& wy0, float& wy1) {
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
        wx1 = ix - fx0; wy1 = iy - fy0; wx0 = (fx0 + 1.f) - ix; wy0 = (fy0 + 1.f) - iy;
        tx0 = (int)fminf(fmaxf(fx0, -2.f), (float)w + 1.f); ty0 = (int)fminf(fmaxf(fy0, -2.f), (float)h + 1.f);
    };
    // logit of one source from its warped K, online-softmax update with its warped V
    auto fold = [&](const float (&ka)[8], const float (&va)[8]) {
        float dot = qbk;
#pragma unroll
        for (int k = 0; k < 8; ++k) dot += ka[k] * q8[k];
#pragma unroll
        for (int off = LPP >> 1; off > 0; off >>= 1) dot += __shfl_xor(dot, off, 64);
        const float logit = dot * inv_sqrt_c;
        const float mnew = fmaxf(mrun, logit);
        const float corr = expf(mrun - mnew);  // exp(-inf) = 0 on the first source
        const float pr = expf(logit - mnew);
        lrun = lrun * corr + pr;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = o[k] * corr + pr * va[k];
        mrun = mnew;
    };
    if (BUF) {
        // sources two at a time: the flow loads and the sixteen K / V gathers of a PAIR are in flight together - the kernel is a latency
        // chain (flow -> addresses -> gathers -> softmax) with two pixels per wave, so the second source's chain rides under the first's
        const unsigned nsrc = (unsigned)(src_batched ? B * ns : ns);
        const int nbytes = (int)(nsrc * (unsigned)hw * (unsigned)C * 2u);
        __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(Ks), 0, nbytes, 0x00020000);
        __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(Vs), 0, nbytes, 0x00020000);
        struct Taps { uintx4 kr[4], vr[4]; float wx0, wx1, wy0, wy1; };
        auto issue = [&](int s, Taps& tp) {
            int tx0, ty0;
            position(s, tx0, ty0, tp.wx0, tp.wx1, tp.wy0, tp.wy1);
            const unsigned sidx = (unsigned)(src_batched ? b * ns + s : s);
            const unsigned sbase = (sidx * (unsigned)hw * (unsigned)C + 8u * (unsigned)cl) * 2u;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int ty = ty0 + (t >> 1), tx = tx0 + (t & 1);
                const bool ok = ty >= 0 && ty < h && tx >= 0 && tx < w;
                const unsigned voff = ok ? sbase + (unsigned)(ty * w + tx) * (unsigned)C * 2u : 0xC0000000u;
                tp.kr[t] = __builtin_bit_cast(uintx4, __builtin_amdgcn_raw_buffer_load_b128(rK, (int)voff, 0, 0));
                tp.vr[t] = __builtin_bit_cast(uintx4, __builtin_amdgcn_raw_buffer_load_b128(rV, (int)voff, 0, 0));
            }
        };
        auto consume = [&](const Taps& tp) {
            float ka[8], va[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) ka[k] = va[k] = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float wt = ((t >> 1) ? tp.wy1 : tp.wy0) * ((t & 1) ? tp.wx1 : tp.wx0);
                float k8[8], v8[8];
                lwg_unpack8(tp.kr[t], k8);
                lwg_unpack8(tp.vr[t], v8);
#pragma unroll
                for (int k = 0; k < 8; ++k) { ka[k] += k8[k] * wt; va[k] += v8[k] * wt; }
            }
            fold(ka, va);
        };
        for (int s = 0; s < ns; s += PAIR ? 2 : 1) {
            Taps t0, t1;
            const bool two = PAIR && s + 1 < ns;   // wave-uniform
            issue(s, t0);
            if (two) issue(s + 1, t1);
            consume(t0);
            if (two) consume(t1);
        }
    } else {
        for (int s = 0; s < ns; ++s) {
            int tx0, ty0;
            float wx0, wx1, wy0, wy1;
            position(s, tx0, ty0, wx0, wx1, wy0, wy1);
            const size_t sidx = src_batched ? (size_t)b * ns + s : (size_t)s;
            const __bf16* Kb = Ks + sidx * hw * C + 8 * cl;
            const __bf16* Vb = Vs + sidx * hw * C + 8 * cl;
            float ka[8], va[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) ka[k] = va[k] = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int ty = ty0 + (t >> 1), tx = tx0 + (t & 1);
                const float wt = ((t >> 1) ? wy1 : wy0) * ((t & 1) ? wx1 : wx0);
                if (ty >= 0 && ty < h && tx >= 0 && tx < w) {
                    const size_t off = ((size_t)ty * w + tx) * C;
                    float k8[8], v8[8];
                    lwg_unpack8(*reinterpret_cast<const uintx4*>(Kb + off), k8);
                    lwg_unpack8(*reinterpret_cast<const uintx4*>(Vb + off), v8);
#pragma unroll
                    for (int k = 0; k < 8; ++k) { ka[k] += k8[k] * wt; va[k] += v8[k] * wt; }
                }
            }
            fold(ka, va);
        }
    }
    if (live) {
        const float invl = 1.f / lrun;
        uintx4 r;
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = lwg_pack2(o[2 * k] * invl + bv[8 * cl + 2 * k], o[2 * k + 1] * invl + bv[8 * cl + 2 * k + 1]);
        *reinterpret_cast<uintx4*>(out + gp * C + 8 * cl) = r;
    }
}

// q / Ks / Vs / out: bf16 tensors shaped as in lwg_lwb_attention_f32; bk / bv / T fp32.
extern "C" int lwg_lwb_attention_bf16(const void* q, const void* Ks, const void* Vs, const float* bk, const float* bv, const float* T,
                                      void* out, int B, int ns, int h, int w, int C, int S, int src_batched, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!q || !Ks || !Vs || !bk || !bv || !T || !out || B <= 0 || ns <= 0 || h <= 0 || w <= 0 || S <= 0) return (int)hipErrorInvalidValue;
    const long total = lwg_tile_frame_positions(B, h, w);
    const bool buf_ok = (unsigned long long)(src_batched ? B * ns : ns) * (unsigned long long)h * w * (unsigned long long)C * 2ull < 0xC0000000ull;
    const __bf16* qb = reinterpret_cast<const __bf16*>(q);
    const __bf16* Kb = reinterpret_cast<const __bf16*>(Ks);
    const __bf16* Vb = reinterpret_cast<const __bf16*>(Vs);
    __bf16* ob = reinterpret_cast<__bf16*>(out);
#define LWG_ATTN16_GO(LPP, BUF, PAIR, OCC)                                                                                              \
    hipLaunchKernelGGL((lwg_lwb_attn_bf16_kernel<LPP, BUF, PAIR, OCC>), dim3((unsigned)((total + 4 * (64 / LPP) - 1) / (4 * (64 / LPP)))), \
                       dim3(256), 0, stream, qb, Kb, Vb, bk, bv, T, ob, B, ns, h, w, S, src_batched)
#define LWG_ATTN16_LAUNCH(LPP)                                                          \
    {                                                                                   \
        if (!buf_ok) LWG_ATTN16_GO(LPP, false, false, 4);                               \
        else LWG_ATTN16_GO(LPP, true, (LWG_ATTN16_PAIR != 0), (LWG_ATTN16_PAIR ? 3 : LWG_ATTN16_OCC)); \
    }
    switch (C) {
        case 64: LWG_ATTN16_LAUNCH(8) break;
        case 128: LWG_ATTN16_LAUNCH(16) break;
        case 256: LWG_ATTN16_LAUNCH(32) break;
        default: return (int)hipErrorInvalidValue;
    }
#undef LWG_ATTN16_LAUNCH
#undef LWG_ATTN16_GO
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------- output head + compositing
// The two bias-free 5x5 regressors (tsf_img_reg 64 -> 3 + tanh, tsf_att_reg 64 -> 1 + sigmoid, attlwb_spade_resunet.py:605-613)
// and pred = mask * bg + (1 - mask) * img (models/imitator.py:393) on a bf16 NHWC input, on the matrix cores.
// v_mfma_f32_16x16x32_bf16 with the weight block as the row operand: the 16 rows are (tap kx = 0..3) x (output 0..3) of one
// kernel row ky, the 16 columns are 16 consecutive INPUT pixels of an image row, K = 32 channels.  One MFMA therefore forms, for
// 16 input pixels, their contribution through four horizontal taps to the 4 outputs; the fifth tap (kx = 4) is a second MFMA on
// the same activation fragment (rows 4..15 of its weight block are zero).  The contribution of input pixel x through tap kx
// belongs to output pixel x - kx + 2: the per-tap partials are written to LDS at the shifted position and summed there.
// Executed MFMA work = 2 / 5 * (16 / 4) = 1.6 x the algorithmic flops (13.4 GFLOP per 1024^2 frame -> 21 executed) on a pipe that
// delivers > 1 PFLOP/s, instead of a VALU convolution at 60 TFLOP/s (csrc/head.hip).
// A workgroup (4 waves, one output row each) owns a 4-row x (16 NCB - 4)-pixel output tile: the 8 x 16 NCB-pixel x 64-channel halo tile (32 / 64 KB)
// goes global -> LDS by LDS-DMA (16 B per lane, 8 pixels per wave-instruction, out-of-image pixels = out-of-range offsets = zeros),
// k-octets swizzled by the pixel column (source-side permutation, lane-linear destination) so the ds_read_b128 operand reads are
// conflict-free; four (NCB = 2) or two (NCB = 4) workgroups per CU overlap one's staging with the others' MFMAs.
#define H16_TH 4                  // output rows per tile = waves per workgroup
#define H16_HROWS (H16_TH + 4)

// NCB: 16-pixel MFMA column blocks per tile row: input span 16 NCB, output 16 NCB - 4.  NCB = 4: 64 KB of LDS, two workgroups per CU;
// NCB = 2: 32 KB, four per CU (the tile is a short latency chain - DMA, 40 MFMAs, two LDS passes, scattered plane stores - so more
// resident workgroups stream better; the extra halo columns are L2 hits).
template <int NCB>
__global__ __launch_bounds__(256, NCB == 4 ? 2 : (NCB == 3 ? 3 : 4)) void lwg_head_bf16_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ wb,
                                                              const float* __restrict__ bg, size_t bg_bstride, int S, unsigned xbytes,
                                                              float* __restrict__ pred, float* __restrict__ mask_out,
                                                              float* __restrict__ img_out) {
    // NCB = 3: 32 output pixels of the 44 the 48-pixel span could give - every plane store / background load of a tile row is then one
    // aligned 128-byte line (28- or 60-pixel rows straddle lines: the fp32 NCHW planes are written 4 bytes per pixel)
    constexpr int H16_HWID = 16 * NCB, H16_TW = NCB == 3 ? 32 : H16_HWID - 4;
    constexpr int H16_PW = H16_HWID + 4;                            // partial-sum row length: output column c is stored at c + 3 (c in [-3, HWID))
    extern __shared__ __attribute__((aligned(16))) char sm[];      // [H16_HROWS][HWID px][128 B]; later the partial sums
    typedef float floatx4v __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // 1-D grid, XCD-aware: a contiguous band of tile rows per XCD - the halo rows of vertically adjacent tiles meet in one L2
    const int tiles_x = (S + H16_TW - 1) / H16_TW, tiles_y = (S + H16_TH - 1) / H16_TH;
    const int lid = lwg_xcd_remap(blockIdx.x, gridDim.x);
    const int b = lid / (tiles_x * tiles_y), trem = lid - b * tiles_x * tiles_y;
    const int x0 = (trem % tiles_x) * H16_TW, y0 = (trem / tiles_x) * H16_TH;
    // ---- stage the halo tile: halo pixel (py, px) = image (y0 + py - 2, x0 + px - 2); piece = 8 consecutive px of one row
    {
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(x), 0, (int)xbytes, 0x00020000);
        constexpr int PPR = H16_HWID / 8;                          // 8-pixel DMA pieces per halo row
        constexpr int PIECES = H16_HROWS * PPR;
#pragma unroll
        for (int q = 0; q < PIECES / 4; ++q) {
            const int piece = wid * (PIECES / 4) + q;
            const int py = piece / PPR, px = (piece % PPR) * 8 + (lane >> 3);
            const int gy = y0 + py - 2, gx = x0 + px - 2;
            const bool ok = gy >= 0 && gy < S && gx >= 0 && gx < S;
            const unsigned oct = (unsigned)((lane & 7) ^ (px & 7));
            const unsigned voff = ok ? (unsigned)((((size_t)b * S + gy) * S + gx) * 128u) + oct * 16u : 0xC0000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(sm + piece * 1024), 16, (int)voff, 0, 0, 0);
        }
    }
    // ---- weight fragments (20 x 16 B per lane), while the DMA is in flight
    bf16x8 wf[5][2][2];
#pragma unroll
    for (int ky = 0; ky < 5; ++ky)
#pragma unroll
        for (int ps = 0; ps < 2; ++ps)
#pragma unroll
            for (int ch = 0; ch < 2; ++ch)
                wf[ky][ps][ch] = *reinterpret_cast<const bf16x8*>(wb + ((((size_t)ky * 2 + ps) * 2 + ch) * 64 + lane) * 8);
    floatx4v acc[NCB][2];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) acc[cb][0] = acc[cb][1] = floatx4v{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_s_waitcnt(0x0f70);       // vmcnt(0): this wave's pieces (and its weight loads) have landed
    __syncthreads();
    const int pxl = lane & 15, koct = lane >> 4;
#pragma unroll
    for (int ky = 0; ky < 5; ++ky) {
        const int py = wid + ky;                                   // halo row feeding output row wid through kernel row ky
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            const int px = cb * 16 + pxl;
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                const bf16x8 av = *reinterpret_cast<const bf16x8*>(sm + ((size_t)py * H16_HWID + px) * 128 + (((ch * 4 + koct) ^ (px & 7)) << 4));
                acc[cb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ky][0][ch], av, acc[cb][0], 0, 0, 0);
                acc[cb][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ky][1][ch], av, acc[cb][1], 0, 0, 0);
            }
        }
    }
    __syncthreads();           // every wave is done with the halo tile: reuse the LDS for the shifted partial sums
    // D layout: lane (pixel j = lane & 15, tap t = lane >> 4) holds the 4 outputs of tap t for INPUT pixel x0 - 2 + cb*16 + j, which
    // belongs to output column c = cb*16 + j - t (pass 0) / c = cb*16 + j - 4 for the fifth tap (pass 1, held by the t = 0 lanes).
    float* part = reinterpret_cast<float*>(sm);                    // [H16_TH][5][H16_PW][4]
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        const int j = cb * 16 + pxl;
        *reinterpret_cast<floatx4v*>(part + (((size_t)wid * 5 + koct) * H16_PW + (j - koct + 3)) * 4) = acc[cb][0];
        if (koct == 0 && j >= 1) *reinterpret_cast<floatx4v*>(part + (((size_t)wid * 5 + 4) * H16_PW + (j - 4 + 3)) * 4) = acc[cb][1];
    }
    __syncthreads();
    const size_t plane = (size_t)S * S;
    for (int i = tid; i < H16_TH * H16_TW; i += 256) {
        const int r = i / H16_TW, c = i - r * H16_TW;
        const int gy = y0 + r, gx = x0 + c;
        if (gy >= S || gx >= S) continue;
        floatx4v s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 5; ++t) s += *reinterpret_cast<const floatx4v*>(part + (((size_t)r * 5 + t) * H16_PW + c + 3) * 4);
        const size_t pix = (size_t)gy * S + gx;
        const float m = 1.f / (1.f + expf(-s[3]));
        if (mask_out) mask_out[(size_t)b * plane + pix] = m;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float im = tanhf(s[ch]);
            if (img_out) img_out[((size_t)b * 3 + ch) * plane + pix] = im;
            if (pred) {
                const float bgv = bg[(size_t)b * bg_bstride + ch * plane + pix];
                pred[((size_t)b * 3 + ch) * plane + pix] = m * bgv + (1.f - m) * im;
            }
        }
    }
}

// x (B,S,S,64) bf16 NHWC (any batch: beyond 3 GiB the launch runs in slices of frames); wb: the bf16 operand panel of ipercore_amd.networks.packing.pack_head_bf16 ([5][2][2][64][8]);
// bg / pred / mask / img as in lwg_head_compose_f32 (fp32 NCHW).
extern "C" int lwg_head_compose_bf16(const void* x, const void* wb, const float* bg, size_t bg_bstride, int B, int S, int C, float* pred,
                                     float* mask, float* img, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!x || !wb || (pred && !bg) || (!pred && !mask && !img) || B <= 0 || S <= 0 || C != 64 || B > 65535) return (int)hipErrorInvalidValue;
    const unsigned long long per = (unsigned long long)S * S * 128ull;
    if (per >= 0xC0000000ull) return (int)hipErrorInvalidValue;
    if ((unsigned long long)B * per >= 0xC0000000ull) {     // the halo goes through 32-bit buffer offsets: larger batches run in slices of frames
        const int nbs = (int)((0xC0000000ull - 1ull) / per);
        const size_t plane = (size_t)S * S;
        for (int b0 = 0; b0 < B; b0 += nbs) {
            const int nb = B - b0 < nbs ? B - b0 : nbs;
            const int e = lwg_head_compose_bf16(static_cast<const char*>(x) + (size_t)b0 * per, wb, bg ? bg + (size_t)b0 * bg_bstride : bg, bg_bstride, nb, S,
                                                C, pred ? pred + (size_t)b0 * 3 * plane : pred, mask ? mask + (size_t)b0 * plane : mask,
                                                img ? img + (size_t)b0 * 3 * plane : img, stream_);
            if (e != 0) return e;
        }
        return 0;
    }
    const unsigned long long xbytes = (unsigned long long)B * per;
    constexpr int ncb = LWG_HEAD16_NCB;             // 16-pixel column blocks per tile row (compile-time, lwg_common.h)
    static_assert(ncb == 2 || ncb == 3 || ncb == 4, "LWG_HEAD16_NCB");
    const int tw = ncb == 3 ? 32 : 16 * ncb - 4;
    const size_t lds = (size_t)H16_HROWS * 16 * ncb * 128;       // >= the partial sums: TH * 5 * (16 NCB + 4) * 16 B
    const unsigned grid = (unsigned)(((S + tw - 1) / tw) * ((S + H16_TH - 1) / H16_TH) * B);
    hipLaunchKernelGGL(lwg_head_bf16_kernel<ncb>, dim3(grid), dim3(256), lds, stream, reinterpret_cast<const __bf16*>(x),
                       reinterpret_cast<const __bf16*>(wb), bg, bg_bstride, S, (unsigned)xbytes, pred, mask, img);
    return (int)hipGetLastError();
}
