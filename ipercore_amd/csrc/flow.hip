// Fused per-pixel consumer of the face-index / weight maps: everything the reference derives from (fim, wim)
// for one target frame, in ONE pass over the maps (16 B/pixel read once instead of 2 + ns times):
//   cond   = map_fn[fim]                      renders/nmr.py:390-401 (encode_fim; fim == -1 -> last row)
//   Tuv2t  = sum_k wim_k * f_uvs2img[fim,k]   renders/nmr.py:713-757 (cal_bc_transform), flowcomposition.py:240
//   syn    = grid_sample(uv_img, Tuv2t)       models/flowcomposition.py:242 (bilinear, zeros, align_corners=False)
//   tsf_inputs = cat[syn, cond]               models/flowcomposition.py:244   -> NHWC, 8 channels (6 used)
//   Tst[s] = sum_k wim_k * src_f2pts[s,fim,k] models/flowcomposition.py:551-567 (make_trans_flow), s < ns
// No dynamic shapes, no host sync (the reference's boolean-mask indexing forces both).
// HBM-bound: per pixel 16 B in, 32 + 8*ns B out (+ optional NCHW cond / Tuv2t for API parity); tables are
// L2-resident (0.33 MB each).
#include "lwg_common.h"
#include "lwg_conv_args.h"

__global__ __launch_bounds__(256) void lwg_flow_compose_kernel(
    const int* __restrict__ fim, const float* __restrict__ wim, int B, int S, const float* __restrict__ map_fn, int nf,
    const float* __restrict__ f_uvs2img, const float* __restrict__ uv_img4, int Hu, int Wu,
    const float* __restrict__ src_f2pts, int ns, float* __restrict__ tsf_inputs, float* __restrict__ Tst,
    float* __restrict__ cond_nchw, float* __restrict__ Tuv) {
    const size_t total = (size_t)B * S * S;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int f = fim[i];
        const bool fg = f >= 0;
        const float w0 = wim[3 * i + 0], w1 = wim[3 * i + 1], w2 = wim[3 * i + 2];
        const int b = (int)(i / ((size_t)S * S));
        const size_t pix = i - (size_t)b * S * S;
        // cond: background indexes the LAST row (Python negative indexing in the reference)
        const float* mrow = map_fn + (size_t)(fg ? f : nf) * 3;
        const float c0 = mrow[0], c1 = mrow[1], c2 = mrow[2];
        // UV flow
        float ux = -2.f, uy = -2.f;
        if (fg) {
            const float* t = f_uvs2img + (size_t)f * 6;
            ux = (t[0] * w0 + t[2] * w1) + t[4] * w2;
            uy = (t[1] * w0 + t[3] * w1) + t[5] * w2;
        }
        // syn = bilinear sample of uv_img (stored NHWC with 4 channels, 4th unused)
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
        {
            const float ix = ((ux + 1.f) * (float)Wu - 1.f) * 0.5f, iy = ((uy + 1.f) * (float)Hu - 1.f) * 0.5f;
            const float fx0 = floorf(ix), fy0 = floorf(iy);
            const float wx1 = ix - fx0, wy1 = iy - fy0, wx0 = (fx0 + 1.f) - ix, wy0 = (fy0 + 1.f) - iy;
            const int tx0 = (int)fminf(fmaxf(fx0, -2.f), (float)Wu + 1.f), ty0 = (int)fminf(fmaxf(fy0, -2.f), (float)Hu + 1.f);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int ty = ty0 + (t >> 1), tx = tx0 + (t & 1);
                if (ty >= 0 && ty < Hu && tx >= 0 && tx < Wu) {
                    const float wt = ((t >> 1) ? wy1 : wy0) * ((t & 1) ? wx1 : wx0);
                    const floatx4 v = *reinterpret_cast<const floatx4*>(uv_img4 + ((size_t)ty * Wu + tx) * 4);
                    s0 += v[0] * wt; s1 += v[1] * wt; s2 += v[2] * wt;
                }
            }
        }
        floatx4* o = reinterpret_cast<floatx4*>(tsf_inputs + i * 8);
        o[0] = floatx4{s0, s1, s2, c0};
        o[1] = floatx4{c1, c2, 0.f, 0.f};
        if (cond_nchw) {
            float* cb = cond_nchw + (size_t)b * 3 * S * S + pix;
            cb[0] = c0; cb[(size_t)S * S] = c1; cb[2 * (size_t)S * S] = c2;
        }
        if (Tuv) { Tuv[2 * i] = ux; Tuv[2 * i + 1] = uy; }
        for (int s = 0; s < ns; ++s) {
            float tx = -2.f, ty = -2.f;
            if (fg) {
                const float* t = src_f2pts + ((size_t)s * nf + f) * 6;
                tx = (t[0] * w0 + t[2] * w1) + t[4] * w2;
                ty = (t[1] * w0 + t[3] * w1) + t[5] * w2;
            }
            float2* dst = reinterpret_cast<float2*>(Tst) + ((size_t)b * ns + s) * S * S + pix;
            *dst = make_float2(tx, ty);
        }
    }
}

// Generic barycentric flow (renders/nmr.py:713-757) for API parity: T[b] = sum_k wim * f2pts[b, fim].
__global__ void lwg_bc_transform_kernel(const float* __restrict__ f2pts, const int* __restrict__ fim,
                                        const float* __restrict__ wim, int B, int S, int nf, float* __restrict__ T) {
    const size_t total = (size_t)B * S * S;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int f = fim[i];
        float tx = -2.f, ty = -2.f;
        if (f >= 0) {
            const int b = (int)(i / ((size_t)S * S));
            const float* t = f2pts + ((size_t)b * nf + f) * 6;
            const float w0 = wim[3 * i + 0], w1 = wim[3 * i + 1], w2 = wim[3 * i + 2];
            tx = (t[0] * w0 + t[2] * w1) + t[4] * w2;
            ty = (t[1] * w0 + t[3] * w1) + t[5] * w2;
        }
        T[2 * i] = tx; T[2 * i + 1] = ty;
    }
}

// table[fim] gather (encode_fim with any map_fn): out NCHW (B,D,S,S)
__global__ void lwg_encode_fim_kernel(const int* __restrict__ fim, const float* __restrict__ map_fn, int B, int S, int nf,
                                      int D, float* __restrict__ out) {
    const size_t total = (size_t)B * S * S;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int f = fim[i];
        const int b = (int)(i / ((size_t)S * S));
        const size_t pix = i - (size_t)b * S * S;
        const float* row = map_fn + (size_t)(f >= 0 ? f : nf) * D;
        for (int d = 0; d < D; ++d) out[((size_t)b * D + d) * S * S + pix] = row[d];
    }
}

static inline int lwg_grid_for(size_t total) {
    const size_t blocks = (total + 255) / 256;
    return (int)(blocks < 8192 ? blocks : 8192);
}

extern "C" int lwg_flow_compose_f32(const int32_t* fim, const float* wim, int B, int S, const float* map_fn, int nf,
                                    const float* f_uvs2img, const float* uv_img4, int Hu, int Wu, const float* src_f2pts,
                                    int ns, float* tsf_inputs, float* Tst, float* cond_nchw, float* Tuv,
                                    lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!fim || !wim || !map_fn || !f_uvs2img || !uv_img4 || !tsf_inputs || (ns > 0 && (!src_f2pts || !Tst)) || B <= 0 || S <= 0)
        return (int)hipErrorInvalidValue;
    const size_t total = (size_t)B * S * S;
    hipLaunchKernelGGL(lwg_flow_compose_kernel, dim3(lwg_grid_for(total)), dim3(256), 0, stream, fim, wim, B, S, map_fn, nf,
                       f_uvs2img, uv_img4, Hu, Wu, src_f2pts, ns, tsf_inputs, Tst, cond_nchw, Tuv);
    return (int)hipGetLastError();
}

extern "C" int lwg_bc_transform_f32(const float* f2pts, const int32_t* fim, const float* wim, int B, int S, int nf, float* T,
                                    lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!f2pts || !fim || !wim || !T || B <= 0 || S <= 0) return (int)hipErrorInvalidValue;
    const size_t total = (size_t)B * S * S;
    hipLaunchKernelGGL(lwg_bc_transform_kernel, dim3(lwg_grid_for(total)), dim3(256), 0, stream, f2pts, fim, wim, B, S, nf, T);
    return (int)hipGetLastError();
}

extern "C" int lwg_encode_fim_f32(const int32_t* fim, const float* map_fn, int B, int S, int nf, int D, float* out,
                                  lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!fim || !map_fn || !out || B <= 0 || S <= 0 || D <= 0) return (int)hipErrorInvalidValue;
    const size_t total = (size_t)B * S * S;
    hipLaunchKernelGGL(lwg_encode_fim_kernel, dim3(lwg_grid_for(total)), dim3(256), 0, stream, fim, map_fn, B, S, nf, D, out);
    return (int)hipGetLastError();
}
