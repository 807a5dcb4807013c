// Liquid Warping Block, attention form, with the QUERY PROJECTION FOLDED INTO THE SOURCE SIDE ("x form").
//
// Reference semantics (generators/attlwb_spade_resunet.py:106-139 SelfAttentionBlock, :175-191 LWB, :208-252 SelfAttentionLWB):
//     q      = Wq x + bq                                   (fq, 1x1 conv on the transfer feature x)
//     K_s    = warp_s(Wk f_s) + bk,   V_s = warp_s(Wv f_s) + bv          (fk / fv after the bilinear warp; csrc/lwb_attn.hip hoists them)
//     l_s    = K_s . q / sqrt(C),     a = softmax_s(l),     out = sum_s a_s V_s
// The warp is linear (zero padding), so
//     K_s . q = warp_s(Wq^T Wk f_s) . x  +  warp_s(bq . Wk f_s)  +  bk . (Wq x + bq)
// and the last term does not depend on s: it cancels in the softmax.  With  Kq_s = (Wq^T Wk) f_s  (C channels) and
// kappa_s = (Wk^T bq) . f_s  (one scalar per source texel) - both functions of the cached source features only, computed ONCE per source -
//     l_s = [ warp_s(Kq_s) . x + warp_s(kappa_s) ] / sqrt(C)      out = sum_s a_s warp_s(Vs_s) + bv
// The per-frame fq convolution (9 launches and 8 GFLOP per 512x512 frame), the q tensor (one write + one read per site) and bk disappear.
//
// One workgroup = one 8 x 8-pixel tile of one frame (tile-major, frame-minor order: lwg_common.h), 4 waves; a pixel = LPP lanes of 16 bytes
// (4 fp32 / 8 bf16 channels); the tile is walked in passes of 256 / LPP pixels, the x row of the next pass in flight during the gathers of
// the current one.  A wave whose pixels ALL have no in-image tap for a source (background: flow = -2) issues no gathers for it.
// STATS: the kernel reads every element of x exactly once, so it also leaves the InstanceNorm partial statistics of x (SPADE's
// parameter-free norm, attlwb_spade_resunet.py:62,83) - one (count, mean, M2) record per (frame, tile, channel), merged by
// lwg_in_stats_final (norm.hip) - and the separate statistics pass over x (9 launches, 9 full reads per frame) disappears.
#include "lwg_common.h"
#include "lwg_conv_args.h"

typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));

template <typename ST> struct AttnXTraits;
template <> struct AttnXTraits<float> {
    static constexpr int CPL = 4;
    __device__ static __forceinline__ void unpack(const uintx4 v, float (&f)[4]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned u = v[k];           // (a bit_cast applied to the vector-element lvalue itself reads element 0 for every k)
            f[k] = __builtin_bit_cast(float, u);
        }
    }
    __device__ static __forceinline__ uintx4 pack(const float (&f)[4]) {
        uintx4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = __builtin_bit_cast(unsigned, f[k]);
        return v;
    }
};
template <> struct AttnXTraits<__bf16> {
    static constexpr int CPL = 8;
    __device__ static __forceinline__ void unpack(const uintx4 v, float (&f)[8]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f[2 * k] = __builtin_bit_cast(float, v[k] << 16);
            f[2 * k + 1] = __builtin_bit_cast(float, v[k] & 0xffff0000u);
        }
    }
    __device__ static __forceinline__ uintx4 pack(const float (&f)[8]) {
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        uintx4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bf16x2 p;
            p[0] = (__bf16)f[2 * k];
            p[1] = (__bf16)f[2 * k + 1];
            v[k] = __builtin_bit_cast(unsigned, p);
        }
        return v;
    }
};

template <typename ST, int LPP, bool STATS, int OCC>
__global__ __launch_bounds__(256, OCC) void lwg_lwb_attnx_kernel(const ST* __restrict__ x, const ST* __restrict__ Kq,
                                                                 const float* __restrict__ kappa, const ST* __restrict__ Vs,
                                                                 const float* __restrict__ bv, const float* __restrict__ T,
                                                                 ST* __restrict__ out, float* __restrict__ stats, int B, int ns, int h, int w,
                                                                 int src_batched) {
    using TR = AttnXTraits<ST>;
    constexpr int CPL = TR::CPL;
    constexpr int C = CPL * LPP;
    constexpr int PPW = 64 / LPP;          // pixels per wave and pass
    constexpr int PPP = 4 * PPW;           // pixels per pass of the workgroup
    constexpr int NPASS = 64 / PPP;
    constexpr unsigned ROWB = (unsigned)C * sizeof(ST);      // bytes of one pixel row
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int cl = lane % LPP, pg = lane / LPP;
    const int tiles_x = (w + 7) >> 3;
    const long L = lwg_xcd_remap(blockIdx.x, gridDim.x);
    const int tile = (int)(L / B), b = (int)(L - (long)tile * B);
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int hw = h * w;
    const unsigned nsrc = (unsigned)(src_batched ? B * ns : ns);
    __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc(const_cast<ST*>(Kq), 0, (int)(nsrc * (unsigned)hw * ROWB), 0x00020000);
    __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc(const_cast<ST*>(Vs), 0, (int)(nsrc * (unsigned)hw * ROWB), 0x00020000);
    __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(kappa), 0, (int)(nsrc * (unsigned)hw * 4u), 0x00020000);
    const float inv_sqrt_c = 1.0f / sqrtf((float)C);
    float bvr[CPL];
#pragma unroll
    for (int k = 0; k < CPL; ++k) bvr[k] = bv[CPL * cl + k];

    auto pixel_of = [&](int pass, int& y, int& xx) -> bool {
        const int p = pass * PPP + wid * PPW + pg;
        y = ty * 8 + (p >> 3);
        xx = tx * 8 + (p & 7);
        return y < h && xx < w;
    };
    auto load_x = [&](int pass) -> uintx4 {
        int y, xx;
        const bool live = pixel_of(pass, y, xx);
        const long gp = ((long)b * h + (live ? y : ty * 8)) * w + (live ? xx : tx * 8);
        return *reinterpret_cast<const uintx4*>(x + gp * C + CPL * cl);
    };

    // InstanceNorm partial statistics of x over this tile: shifted sums (shift = the tile's first pixel), fixed combination order
    float shift[CPL], s1[CPL], s2[CPL];
    if (STATS) {
        const long g0 = ((long)b * h + ty * 8) * w + tx * 8;
        TR::unpack(*reinterpret_cast<const uintx4*>(x + g0 * C + CPL * cl), shift);
#pragma unroll
        for (int k = 0; k < CPL; ++k) s1[k] = s2[k] = 0.f;
    }

    uintx4 xraw = load_x(0);
#pragma unroll 1
    for (int pass = 0; pass < NPASS; ++pass) {
        int y, xx;
        const bool live = pixel_of(pass, y, xx);
        const int yc = live ? y : ty * 8, xc = live ? xx : tx * 8;
        const long gp = ((long)b * h + yc) * w + xc;
        float xv[CPL];
        TR::unpack(xraw, xv);
        if (pass + 1 < NPASS) xraw = load_x(pass + 1);               // in flight during this pass's gathers
        if (STATS && live) {
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                const float d = xv[k] - shift[k];
                s1[k] += d;
                s2[k] = __builtin_fmaf(d, d, s2[k]);
            }
        }
        float mrun = -INFINITY, lrun = 0.f;
        float o[CPL];
#pragma unroll
        for (int k = 0; k < CPL; ++k) o[k] = 0.f;
#pragma unroll 1
        for (int s = 0; s < ns; ++s) {
            const float2 t = reinterpret_cast<const float2*>(T)[((size_t)b * ns + s) * hw + (size_t)yc * w + xc];
            // grid_sample, align_corners=False: pixel = ((g + 1) * size - 1) / 2
            const float ix = ((t.x + 1.f) * (float)w - 1.f) * 0.5f, iy = ((t.y + 1.f) * (float)h - 1.f) * 0.5f;
            const float fx0 = floorf(ix), fy0 = floorf(iy);
            const float wx1 = ix - fx0, wy1 = iy - fy0, wx0 = (fx0 + 1.f) - ix, wy0 = (fy0 + 1.f) - iy;
            // clamp before the int conversion so wild flows cannot overflow; out-of-range taps read zeros
            const int tx0 = (int)fminf(fmaxf(fx0, -2.f), (float)w + 1.f), ty0 = (int)fminf(fmaxf(fy0, -2.f), (float)h + 1.f);
            const bool any_tap = live && tx0 >= -1 && tx0 < w && ty0 >= -1 && ty0 < h;
            float ka[CPL], va[CPL], kap = 0.f;
#pragma unroll
            for (int k = 0; k < CPL; ++k) ka[k] = va[k] = 0.f;
            if (__builtin_amdgcn_ballot_w64(any_tap) != 0ull) {      // wave-uniform: background waves issue no gathers
                const unsigned sidx = (unsigned)(src_batched ? b * ns + s : s);
                const unsigned pbase = sidx * (unsigned)hw;
                uintx4 kr[4], vr[4];
                float ar[4];
#pragma unroll
                for (int tp = 0; tp < 4; ++tp) {
                    const int tyy = ty0 + (tp >> 1), txx = tx0 + (tp & 1);
                    const bool ok = live && tyy >= 0 && tyy < h && txx >= 0 && txx < w;
                    const unsigned pix = pbase + (unsigned)(tyy * w + txx);
                    const unsigned voff = ok ? pix * ROWB + (unsigned)(CPL * cl) * (unsigned)sizeof(ST) : 0xC0000000u;
                    kr[tp] = __builtin_bit_cast(uintx4, __builtin_amdgcn_raw_buffer_load_b128(rK, (int)voff, 0, 0));
                    vr[tp] = __builtin_bit_cast(uintx4, __builtin_amdgcn_raw_buffer_load_b128(rV, (int)voff, 0, 0));
                    ar[tp] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rA, ok ? (int)(pix * 4u) : (int)0xC0000000u, 0, 0));
                }
#pragma unroll
                for (int tp = 0; tp < 4; ++tp) {
                    const float wt = ((tp >> 1) ? wy1 : wy0) * ((tp & 1) ? wx1 : wx0);
                    float k8[CPL], v8[CPL];
                    TR::unpack(kr[tp], k8);
                    TR::unpack(vr[tp], v8);
#pragma unroll
                    for (int k = 0; k < CPL; ++k) {
                        ka[k] = __builtin_fmaf(k8[k], wt, ka[k]);
                        va[k] = __builtin_fmaf(v8[k], wt, va[k]);
                    }
                    kap = __builtin_fmaf(ar[tp], wt, kap);
                }
            }
            float dot = 0.f;
#pragma unroll
            for (int k = 0; k < CPL; ++k) dot = __builtin_fmaf(ka[k], xv[k], dot);
#pragma unroll
            for (int off = LPP >> 1; off > 0; off >>= 1) dot += __shfl_xor(dot, off, 64);
            const float logit = (dot + kap) * inv_sqrt_c;
            const float mnew = fmaxf(mrun, logit);
            const float corr = expf(mrun - mnew);  // exp(-inf) = 0 on the first source
            const float pr = expf(logit - mnew);
            lrun = __builtin_fmaf(lrun, corr, pr);
#pragma unroll
            for (int k = 0; k < CPL; ++k) o[k] = __builtin_fmaf(o[k], corr, pr * va[k]);
            mrun = mnew;
        }
        if (live) {
            const float invl = 1.f / lrun;
            float r[CPL];
#pragma unroll
            for (int k = 0; k < CPL; ++k) r[k] = __builtin_fmaf(o[k], invl, bvr[k]);
            *reinterpret_cast<uintx4*>(out + gp * C + CPL * cl) = TR::pack(r);
        }
    }

    if (STATS) {
        // lanes holding the same channels: pixel groups of a wave (xor shuffles), then the four waves through LDS - a fixed order
#pragma unroll
        for (int off = LPP; off < 64; off <<= 1) {
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                s1[k] += __shfl_xor(s1[k], off, 64);
                s2[k] += __shfl_xor(s2[k], off, 64);
            }
        }
        __shared__ float sh[4][2][C];
        if (pg == 0) {
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                sh[wid][0][CPL * cl + k] = s1[k];
                sh[wid][1][CPL * cl + k] = s2[k];
            }
        }
        __syncthreads();
        if (wid == 0 && pg == 0) {
            const int ny = min(8, h - ty * 8), nx = min(8, w - tx * 8);
            const float tn = (float)(ny * nx);
            const int ntiles = tiles_x * ((h + 7) >> 3);
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                const int c = CPL * cl + k;
                const float t1 = ((sh[0][0][c] + sh[1][0][c]) + sh[2][0][c]) + sh[3][0][c];
                const float t2 = ((sh[0][1][c] + sh[1][1][c]) + sh[2][1][c]) + sh[3][1][c];
                const float mean = shift[k] + t1 / tn;
                const float m2 = t2 - t1 * t1 / tn;
                float* rec = stats + (((size_t)b * ntiles + tile) * C + c) * 3;
                rec[0] = tn;
                rec[1] = mean;
                rec[2] = m2 > 0.f ? m2 : 0.f;
            }
        }
    }
}

template <typename ST>
static int lwg_attnx_launch(const ST* x, const ST* Kq, const float* kappa, const ST* Vs, const float* bv, const float* T, ST* out, float* stats,
                            int B, int ns, int h, int w, int C, int src_batched, hipStream_t stream) {
    if (!x || !Kq || !kappa || !Vs || !bv || !T || !out || B <= 0 || ns <= 0 || h <= 0 || w <= 0) return (int)hipErrorInvalidValue;
    const unsigned long long nsrc = (unsigned long long)(src_batched ? B * ns : ns);
    // the taps go through 32-bit buffer offsets with 0xC0000000 as the out-of-range marker (zero fill): K / V below 3 GiB each
    if (nsrc * (unsigned long long)h * w * (unsigned long long)C * sizeof(ST) >= 0xC0000000ull) return (int)hipErrorInvalidValue;
    const long tiles = (long)((w + 7) >> 3) * ((h + 7) >> 3);
    const dim3 grid((unsigned)(tiles * B));
    constexpr int CPL = AttnXTraits<ST>::CPL;
#define LWG_ATTNX_LAUNCH(LPP)                                                                                                      \
    if (stats)                                                                                                                    \
        hipLaunchKernelGGL((lwg_lwb_attnx_kernel<ST, LPP, true, 4>), grid, dim3(256), 0, stream, x, Kq, kappa, Vs, bv, T, out, stats, B, ns, h, w, src_batched); \
    else                                                                                                                          \
        hipLaunchKernelGGL((lwg_lwb_attnx_kernel<ST, LPP, false, 4>), grid, dim3(256), 0, stream, x, Kq, kappa, Vs, bv, T, out, stats, B, ns, h, w, src_batched);
    switch (C / CPL) {
        case 8: LWG_ATTNX_LAUNCH(8) break;
        case 16: LWG_ATTNX_LAUNCH(16) break;
        case 32: LWG_ATTNX_LAUNCH(32) break;
        case 64: if (CPL == 4) { LWG_ATTNX_LAUNCH(64) break; }
                 return (int)hipErrorInvalidValue;
        default: return (int)hipErrorInvalidValue;
    }
#undef LWG_ATTNX_LAUNCH
    return (int)hipGetLastError();
}

// x (B,h,w,C) transfer feature; Kq (nsrc,h,w,C) = (Wq^T Wk) f_src; kappa (nsrc,h,w) = (Wk^T bq) . f_src; Vs (nsrc,h,w,C) = Wv f_src (no bias);
// bv (C); T (B,ns,h,w,2) flows ALREADY RESIZED to (h,w) (lwg_flow_resize_f32), grid_sample coordinates, -2 = background; out (B,h,w,C).
// stats: nullptr, or B * ntiles * C * 3 floats (ntiles = ceil(h/8) * ceil(w/8)) receiving the per-tile InstanceNorm records of x
// (count, mean, M2) - finish with lwg_instnorm_finalize_f32(stats, B, C, ntiles, ...).  C in {32, 64, 128, 256} (bf16: {64, 128, 256}).
extern "C" int lwg_lwb_attention_x_f32(const float* x, const float* Kq, const float* kappa, const float* Vs, const float* bv, const float* T,
                                       float* out, float* stats, int B, int ns, int h, int w, int C, int src_batched, lwg_stream_t stream_) {
    if (C != 32 && C != 64 && C != 128 && C != 256) return (int)hipErrorInvalidValue;
    return lwg_attnx_launch<float>(x, Kq, kappa, Vs, bv, T, out, stats, B, ns, h, w, C, src_batched, reinterpret_cast<hipStream_t>(stream_));
}

extern "C" int lwg_lwb_attention_x_bf16(const void* x, const void* Kq, const float* kappa, const void* Vs, const float* bv, const float* T,
                                        void* out, float* stats, int B, int ns, int h, int w, int C, int src_batched, lwg_stream_t stream_) {
    if (C != 64 && C != 128 && C != 256) return (int)hipErrorInvalidValue;
    return lwg_attnx_launch<__bf16>(static_cast<const __bf16*>(x), static_cast<const __bf16*>(Kq), kappa, static_cast<const __bf16*>(Vs), bv, T,
                                    static_cast<__bf16*>(out), stats, B, ns, h, w, C, src_batched, reinterpret_cast<hipStream_t>(stream_));
}

// Merge of the per-tile records the STATS form leaves: ws (B, nrec, C, 3) = (count, mean, M2) -> mean / rstd (B, C).  One workgroup per
// (frame, 64 channels): lane = channel (a wave reads 768 contiguous bytes per record), wave k takes records k, k + 16, ...  Two passes of
// plain sums instead of a chain of pairwise (Chan) updates with a division each: mu = sum n_i mean_i / sum n_i, then
// M2 = sum [M2_i + n_i (mean_i - mu)^2]; the 16 per-wave partials are added in wave order.  The order depends on nrec only: a frame's
// statistics do not depend on its batch.
__global__ __launch_bounds__(1024) void lwg_in_stats_merge_tiles(const float* __restrict__ ws, int C, int nrec, float eps,
                                                                float* __restrict__ mean, float* __restrict__ rstd) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane, b = blockIdx.y;
    const bool cok = c < C;
    const float* base = ws + ((size_t)b * nrec * C + (cok ? c : 0)) * 3;
    __shared__ float sh[16][2][64];
    float sn = 0.f, sm = 0.f;
    if (cok) {
        for (int r = wid; r < nrec; r += 16) {
            const float* o = base + (size_t)r * C * 3;
            sn += o[0];
            sm = __builtin_fmaf(o[0], o[1], sm);
        }
    }
    sh[wid][0][lane] = sn;
    sh[wid][1][lane] = sm;
    __syncthreads();
    float tn = 0.f, tm = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) { tn += sh[k][0][lane]; tm += sh[k][1][lane]; }
    const float mu = tm / tn;
    __syncthreads();
    float sq = 0.f;
    if (cok) {
        for (int r = wid; r < nrec; r += 16) {
            const float* o = base + (size_t)r * C * 3;
            const float d = o[1] - mu;
            sq += __builtin_fmaf(o[0] * d, d, o[2]);
        }
    }
    sh[wid][0][lane] = sq;
    __syncthreads();
    if (wid == 0 && cok) {
        float m2 = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) m2 += sh[k][0][lane];
        mean[(size_t)b * C + c] = mu;
        rstd[(size_t)b * C + c] = 1.0f / sqrtf(m2 / tn + eps);
    }
}

// ws (B, nrec, C, 3) records (count, mean, M2) -> mean, rstd (B, C): rstd = 1 / sqrt(M2 / n + eps), biased variance (nn.InstanceNorm2d).
extern "C" int lwg_instnorm_finalize_f32(const float* ws, int B, int C, int nrec, float eps, float* mean, float* rstd, lwg_stream_t stream_) {
    if (!ws || !mean || !rstd || B <= 0 || C <= 0 || nrec <= 0 || B > 65535) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(lwg_in_stats_merge_tiles, dim3((C + 63) / 64, B), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream_), ws, C, nrec, eps,
                       mean, rstd);
    return (int)hipGetLastError();
}
