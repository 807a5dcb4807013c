// Elementwise / normalisation kernels of the personalization step (forward pieces that the inference path fuses into conv
// epilogues, and their backward).  Reference ops: nn.InstanceNorm2d + ReLU / LeakyReLU (bg_inpaintor.py:31-57,
// discriminators/patch_dis.py:33-47), SPADE's normalized * (1 + gamma) + beta (attlwb_spade_resunet.py:80-93), ReLU after
// the convs, as differentiated by loss.backward() in tools/trainers/lwg_trainer.py:345,351.
// All tensors NHWC fp32; HBM-bound streaming kernels with 16-byte accesses; reductions are two-pass and deterministic.
#include "lwg_common.h"
#include "lwg_conv_args.h"

#define LWG_ACT_LRELU 4   // LeakyReLU(0.2) (patch discriminator); extends the activation codes of lwg_common.h for these kernels

__device__ __forceinline__ float lwg_act_t(float v, int act) {
    if (act == LWG_ACT_LRELU) return v > 0.f ? v : 0.2f * v;
    return lwg_act(v, act);
}
// derivative of the activation expressed with its OUTPUT y (ReLU / LeakyReLU keep the sign; tanh' = 1 - y^2; sigmoid' = y(1-y))
__device__ __forceinline__ float lwg_dact_from_y(float y, int act) {
    if (act == LWG_ACT_RELU) return y > 0.f ? 1.f : 0.f;
    if (act == LWG_ACT_LRELU) return y > 0.f ? 1.f : 0.2f;
    if (act == LWG_ACT_TANH) return 1.f - y * y;
    if (act == LWG_ACT_SIGMOID) return y * (1.f - y);
    return 1.f;
}

// ---------------------------------------------------------------------------------------------- activation backward
// out = dy * act'(y)     (ReLU mask of ConvFn.backward, tanh / sigmoid of the regressors)
__global__ void lwg_act_bwd_kernel(const floatx4* __restrict__ dy, const floatx4* __restrict__ y, size_t n4, int act,
                                   floatx4* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const floatx4 g = dy[i], v = y[i];
        floatx4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = g[k] * lwg_dact_from_y(v[k], act);
        out[i] = o;
    }
}

extern "C" int lwg_act_bwd_f32(const float* dy, const float* y, size_t n, int act, float* out, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!dy || !y || !out || n == 0 || (n & 3)) return (int)hipErrorInvalidValue;
    const size_t n4 = n / 4;
    const int blocks = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    hipLaunchKernelGGL(lwg_act_bwd_kernel, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const floatx4*>(dy),
                       reinterpret_cast<const floatx4*>(y), n4, act, reinterpret_cast<floatx4*>(out));
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------- per-channel PReLU (+ residual)
// The frozen Sphere20a of the face loss (criterions/faceloss.py:203-285): y = res + (x >= 0 ? x : slope[c] x) on NHWC rows, res optional;
// backward dx = dy * (x >= 0 ? 1 : slope[c]) (the slopes are frozen: no gradient for them; the residual's gradient is dy itself).
template <bool BWD>
__global__ void lwg_prelu_kernel(const floatx4* __restrict__ x, const floatx4* __restrict__ slope, const floatx4* __restrict__ other,
                                 int C4, size_t n4, floatx4* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const floatx4 v = x[i], a = slope[i % C4];
        floatx4 o;
        if (BWD) {
            const floatx4 g = other[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = v[k] >= 0.f ? g[k] : g[k] * a[k];
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = v[k] >= 0.f ? v[k] : v[k] * a[k];
            if (other) {
                const floatx4 r = other[i];
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] += r[k];
            }
        }
        out[i] = o;
    }
}

template <bool BWD>
static int lwg_prelu_launch(const float* x, const float* slope, const float* other, size_t rows, int C, float* out, lwg_stream_t stream_) {
    if (!x || !slope || !out || (BWD && !other) || rows == 0 || C <= 0 || (C & 3)) return (int)hipErrorInvalidValue;
    const size_t n4 = rows * (size_t)(C / 4);
    const int blocks = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    hipLaunchKernelGGL((lwg_prelu_kernel<BWD>), dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_),
                       reinterpret_cast<const floatx4*>(x), reinterpret_cast<const floatx4*>(slope), reinterpret_cast<const floatx4*>(other), C / 4, n4,
                       reinterpret_cast<floatx4*>(out));
    return (int)hipGetLastError();
}

extern "C" int lwg_prelu_f32(const float* x, const float* slope, const float* res, size_t rows, int C, float* y, lwg_stream_t stream) {
    return lwg_prelu_launch<false>(x, slope, res, rows, C, y, stream);
}

extern "C" int lwg_prelu_bwd_f32(const float* x, const float* slope, const float* dy, size_t rows, int C, float* dx, lwg_stream_t stream) {
    return lwg_prelu_launch<true>(x, slope, dy, rows, C, dx, stream);
}

// ---------------------------------------------------------------------------------------------- normalise (+ modulate) forward
// y = act( (x - mean) * rstd * (1 + gamma) + beta )   gamma / beta optional (NULL: plain InstanceNorm + activation)
// gs4: float4s per pixel row of gamma / beta (C4 for dense tensors; 2 * C4 when both are halves of ONE (B,HW,2C) convolution output)
__global__ void lwg_norm_fwd_kernel(const floatx4* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                    const floatx4* __restrict__ gamma, const floatx4* __restrict__ beta, int gs4, int HW, int C4,
                                    size_t total4, int act, floatx4* __restrict__ y) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const size_t pix = i / C4;
        const int b = (int)(pix / (size_t)HW);
        const floatx4 v = x[i];
        const floatx4 mu = *reinterpret_cast<const floatx4*>(mean + ((size_t)b * C4 + c4) * 4);
        const floatx4 rs = *reinterpret_cast<const floatx4*>(rstd + ((size_t)b * C4 + c4) * 4);
        floatx4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (v[k] - mu[k]) * rs[k];
        if (gamma) {
            const floatx4 g = gamma[pix * gs4 + c4], bt = beta[pix * gs4 + c4];
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = o[k] * (1.f + g[k]) + bt[k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = lwg_act_t(o[k], act);
        y[i] = o;
    }
}

// gstride: floats per pixel row of gamma / beta (0 or C: dense (B,HW,C) tensors; 2C with beta = gamma + C: the two halves of one
// (B,HW,2C) tensor - SPADE's gamma | beta convolutions run as ONE launch in the training step)
extern "C" int lwg_norm_fwd_nhwc_f32(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta, int gstride,
                                     int B, int HW, int C, int act, float* y, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!x || !mean || !rstd || !y || (C & 3) || B <= 0 || HW <= 0 || ((gamma == nullptr) != (beta == nullptr))) return (int)hipErrorInvalidValue;
    if (gstride == 0) gstride = C;
    if (gstride < C || (gstride & 3)) return (int)hipErrorInvalidValue;
    const size_t total4 = (size_t)B * HW * (C / 4);
    const int blocks = (int)((total4 + 255) / 256 < 8192 ? (total4 + 255) / 256 : 8192);
    hipLaunchKernelGGL(lwg_norm_fwd_kernel, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const floatx4*>(x), mean, rstd,
                       reinterpret_cast<const floatx4*>(gamma), reinterpret_cast<const floatx4*>(beta), gstride / 4, HW, C / 4, total4, act,
                       reinterpret_cast<floatx4*>(y));
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------- normalise (+ modulate) backward
// With xhat = (x - mean) rstd, z = xhat (1 + gamma) + beta, y = act(z), g = dy act'(y):
//   dgamma = g xhat, dbeta = g, dxhat = g (1 + gamma)            (gamma absent: dxhat = g)
//   dx = rstd ( dxhat - mean_hw(dxhat) - xhat mean_hw(dxhat xhat) )
// Pass 1 (partial): per (b, split, c) the sums of dxhat and dxhat*xhat over the split's pixels, and dgamma / dbeta.
// Pass 2 (apply): folds the split sums in split order and writes dx.
__global__ __launch_bounds__(256) void lwg_norm_bwd_partial_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                                  const float* __restrict__ x, const float* __restrict__ mean,
                                                                  const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                  int gstride, int HW, int C, int nsplit, int act, float* __restrict__ dgamma,
                                                                  float* __restrict__ dbeta, float* __restrict__ ws) {
    const int C4 = C >> 2;
    const int lpp = C4 < 64 ? C4 : 64;                 // lanes per pixel (channel quads handled by this block)
    const int pgs = 256 / lpp;                         // pixel groups in flight
    const int cql = threadIdx.x % lpp, pg = threadIdx.x / lpp;
    const int cq = blockIdx.x * 64 + cql;              // blockIdx.x > 0 only when C > 256
    const int split = blockIdx.y, b = blockIdx.z;
    const bool cok = cq < C4 && pg < pgs;
    const int per = (HW + nsplit - 1) / nsplit;
    const int p0 = split * per, p1 = min(HW, p0 + per);
    floatx4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
    if (cok) {
        const floatx4 mu = *reinterpret_cast<const floatx4*>(mean + (size_t)b * C + cq * 4);
        const floatx4 rs = *reinterpret_cast<const floatx4*>(rstd + (size_t)b * C + cq * 4);
        for (int p = p0 + pg; p < p1; p += pgs) {
            const size_t o = ((size_t)b * HW + p) * C + cq * 4;
            const floatx4 g0 = *reinterpret_cast<const floatx4*>(dy + o), yv = *reinterpret_cast<const floatx4*>(y + o);
            const floatx4 xv = *reinterpret_cast<const floatx4*>(x + o);
            floatx4 gm = {0.f, 0.f, 0.f, 0.f};
            const size_t og = ((size_t)b * HW + p) * gstride + cq * 4;     // gamma / dgamma / dbeta rows (gstride floats apart)
            if (gamma) gm = *reinterpret_cast<const floatx4*>(gamma + og);
            floatx4 g, xh, dxh;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                g[k] = g0[k] * lwg_dact_from_y(yv[k], act);
                xh[k] = (xv[k] - mu[k]) * rs[k];
                dxh[k] = g[k] * (1.f + gm[k]);
                s1[k] += dxh[k];
                s2[k] += dxh[k] * xh[k];
            }
            if (gamma) {
                floatx4 dg;
#pragma unroll
                for (int k = 0; k < 4; ++k) dg[k] = g[k] * xh[k];
                *reinterpret_cast<floatx4*>(dgamma + og) = dg;
                *reinterpret_cast<floatx4*>(dbeta + og) = g;
            }
        }
    }
    __shared__ floatx4 sh1[256], sh2[256];
    sh1[threadIdx.x] = s1;
    sh2[threadIdx.x] = s2;
    __syncthreads();
    if (pg == 0 && cq < C4) {
        floatx4 t1 = {0.f, 0.f, 0.f, 0.f}, t2 = t1;
        for (int g = 0; g < pgs; ++g) { t1 += sh1[g * lpp + cql]; t2 += sh2[g * lpp + cql]; }
        float* o = ws + (((size_t)b * nsplit + split) * C + cq * 4) * 2;
#pragma unroll
        for (int k = 0; k < 4; ++k) { o[2 * k] = t1[k]; o[2 * k + 1] = t2[k]; }
    }
}

// Pass 1b (fold): the nsplit records of a (b, c) added up -> one pair per (b, c).  (The first version left this to the apply kernel:
// every thread of it walked all nsplit records - 2 KB of L2 reads per 48 bytes of tensor data at nsplit = 64.)  A block serves 32
// channels of one image with 8 lanes per channel: lane g adds the records g, g + 8, ... (a 32-channel record row is 256 contiguous
// bytes), then a fixed-order LDS pass adds the 8 partial sums - the association depends only on nsplit: deterministic.  (One thread per
// (b, c) walking 512 records took 52 us per launch, 1.9 ms per personalization step.)
__global__ __launch_bounds__(256) void lwg_norm_bwd_fold_kernel(const float* __restrict__ ws, int nsplit, int C, float* __restrict__ fold) {
    __shared__ float2 sh[8][32];
    const int cl = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl, b = blockIdx.y;
    float m1 = 0.f, m2 = 0.f;
    if (c < C) {
        for (int s = g; s < nsplit; s += 8) {
            const float2 v = *reinterpret_cast<const float2*>(ws + (((size_t)b * nsplit + s) * C + c) * 2);
            m1 += v.x;
            m2 += v.y;
        }
    }
    sh[g][cl] = make_float2(m1, m2);
    __syncthreads();
    if (g == 0 && c < C) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { t1 += sh[k][cl].x; t2 += sh[k][cl].y; }
        *reinterpret_cast<float2*>(fold + 2 * ((size_t)b * C + c)) = make_float2(t1, t2);
    }
}

__global__ void lwg_norm_bwd_apply_kernel(const floatx4* __restrict__ dy, const floatx4* __restrict__ y, const floatx4* __restrict__ x,
                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                          const floatx4* __restrict__ gamma, int gs4, const float* __restrict__ fold, int HW, int C4,
                                          size_t total4, int act, floatx4* __restrict__ dx) {
    const float inv_hw = 1.f / (float)HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const size_t pix = i / C4;
        const int b = (int)(pix / (size_t)HW);
        const float* o = fold + ((size_t)b * C4 + c4) * 8;         // [m1, m2] of four channels
        const floatx4 f0 = *reinterpret_cast<const floatx4*>(o), f1 = *reinterpret_cast<const floatx4*>(o + 4);
        const floatx4 m1 = {f0[0], f0[2], f1[0], f1[2]}, m2 = {f0[1], f0[3], f1[1], f1[3]};
        const floatx4 mu = *reinterpret_cast<const floatx4*>(mean + ((size_t)b * C4 + c4) * 4);
        const floatx4 rs = *reinterpret_cast<const floatx4*>(rstd + ((size_t)b * C4 + c4) * 4);
        const floatx4 g0 = dy[i], yv = y[i], xv = x[i];
        floatx4 gm = {0.f, 0.f, 0.f, 0.f};
        if (gamma) gm = gamma[pix * gs4 + c4];
        floatx4 out;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xh = (xv[k] - mu[k]) * rs[k];
            const float dxh = g0[k] * lwg_dact_from_y(yv[k], act) * (1.f + gm[k]);
            out[k] = rs[k] * (dxh - m1[k] * inv_hw - xh * (m2[k] * inv_hw));
        }
        dx[i] = out;
    }
}

// dy, y, x (B,HW,C); mean, rstd (B,C); gamma (B,HW,C) or NULL.  Outputs dx, and dgamma / dbeta when gamma is given.
// gstride: floats per pixel row of gamma, dgamma and dbeta (0 or C: dense; 2C: halves of (B,HW,2C) tensors, dbeta = dgamma + C - the
//   gradient of a fused gamma | beta convolution output is written in place, no concatenation).
// ws: B * (nsplit + 1) * C * 2 floats (the split records, then their fold); nsplit: enough splits to fill the chip (ops._nsplit_bwd).
extern "C" int lwg_norm_bwd_nhwc_f32(const float* dy, const float* y, const float* x, const float* mean, const float* rstd,
                                     const float* gamma, int gstride, int B, int HW, int C, int act, int nsplit, float* dx, float* dgamma,
                                     float* dbeta, float* ws, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!dy || !y || !x || !mean || !rstd || !dx || !ws || (C & 3) || B <= 0 || HW <= 0 || nsplit < 1 || nsplit > 65535 ||
        (gamma && (!dgamma || !dbeta)))
        return (int)hipErrorInvalidValue;
    if (gstride == 0) gstride = C;
    if (gstride < C || (gstride & 3)) return (int)hipErrorInvalidValue;
    const int C4 = C / 4;
    hipLaunchKernelGGL(lwg_norm_bwd_partial_kernel, dim3((C4 + 63) / 64, nsplit, B), dim3(256), 0, stream, dy, y, x, mean, rstd, gamma,
                       gstride, HW, C, nsplit, act, dgamma, dbeta, ws);
    float* fold = ws + (size_t)B * nsplit * C * 2;
    hipLaunchKernelGGL(lwg_norm_bwd_fold_kernel, dim3((C + 31) / 32, B), dim3(256), 0, stream, ws, nsplit, C, fold);
    const size_t total4 = (size_t)B * HW * C4;
    const int blocks = (int)((total4 + 255) / 256 < 8192 ? (total4 + 255) / 256 : 8192);
    hipLaunchKernelGGL(lwg_norm_bwd_apply_kernel, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const floatx4*>(dy),
                       reinterpret_cast<const floatx4*>(y), reinterpret_cast<const floatx4*>(x), mean, rstd,
                       reinterpret_cast<const floatx4*>(gamma), gstride / 4, fold, HW, C4, total4, act, reinterpret_cast<floatx4*>(dx));
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------- Adam
// torch.optim.Adam (lwg_trainer.py:140-146; no weight decay, no amsgrad) over one flat fp32 parameter buffer:
//   m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= lr / (1 - b1^t) * m / (sqrt(v / (1 - b2^t)) + eps)
__global__ void lwg_adam_kernel(floatx4* __restrict__ p, const floatx4* __restrict__ g, floatx4* __restrict__ m, floatx4* __restrict__ v,
                                size_t n4, float lr, float b1, float b2, float eps, float bc1, float bc2) {
    const float step = lr / bc1, isq = 1.f / sqrtf(bc2);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        floatx4 pp = p[i], mm = m[i], vv = v[i];
        const floatx4 gg = g[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mm[k] = b1 * mm[k] + (1.f - b1) * gg[k];
            vv[k] = b2 * vv[k] + (1.f - b2) * gg[k] * gg[k];
            pp[k] -= step * mm[k] / (sqrtf(vv[k]) * isq + eps);
        }
        p[i] = pp; m[i] = mm; v[i] = vv;
    }
}

extern "C" int lwg_adam_step_f32(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                                 int t, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!p || !g || !m || !v || n == 0 || (n & 3) || t < 1) return (int)hipErrorInvalidValue;
    const float bc1 = 1.f - powf(beta1, (float)t), bc2 = 1.f - powf(beta2, (float)t);
    const size_t n4 = n / 4;
    const int blocks = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    hipLaunchKernelGGL(lwg_adam_kernel, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<floatx4*>(p), reinterpret_cast<const floatx4*>(g),
                       reinterpret_cast<floatx4*>(m), reinterpret_cast<floatx4*>(v), n4, lr, beta1, beta2, eps, bc1, bc2);
    return (int)hipGetLastError();
}

// The same update with the step count on the DEVICE: *t_dev is incremented by a one-thread launch, then the update reads it and
// forms the bias corrections itself - so a captured (hipGraph) training step replays with the right t (a host-side t would be
// frozen into the graph at capture time).
__global__ void lwg_adam_tick_kernel(int* t_dev) { *t_dev += 1; }

__global__ void lwg_adam_dev_kernel(floatx4* __restrict__ p, const floatx4* __restrict__ g, floatx4* __restrict__ m, floatx4* __restrict__ v,
                                    size_t n4, float lr, float b1, float b2, float eps, const int* __restrict__ t_dev) {
    const float t = (float)*t_dev;
    const float bc1 = 1.f - powf(b1, t), bc2 = 1.f - powf(b2, t);
    const float step = lr / bc1, isq = 1.f / sqrtf(bc2);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        floatx4 pp = p[i], mm = m[i], vv = v[i];
        const floatx4 gg = g[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mm[k] = b1 * mm[k] + (1.f - b1) * gg[k];
            vv[k] = b2 * vv[k] + (1.f - b2) * gg[k] * gg[k];
            pp[k] -= step * mm[k] / (sqrtf(vv[k]) * isq + eps);
        }
        p[i] = pp; m[i] = mm; v[i] = vv;
    }
}

extern "C" int lwg_adam_step_dev_f32(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                                     int* t_dev, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!p || !g || !m || !v || !t_dev || n == 0 || (n & 3)) return (int)hipErrorInvalidValue;
    const size_t n4 = n / 4;
    const int blocks = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    hipLaunchKernelGGL(lwg_adam_tick_kernel, dim3(1), dim3(1), 0, stream, t_dev);
    hipLaunchKernelGGL(lwg_adam_dev_kernel, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<floatx4*>(p), reinterpret_cast<const floatx4*>(g),
                       reinterpret_cast<floatx4*>(m), reinterpret_cast<floatx4*>(v), n4, lr, beta1, beta2, eps, t_dev);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Weight panels straight from the parameter tensors (personalization step: every step re-packs every weight - forward panel,
// data-gradient panel - and un-packs every weight gradient; as chains of torch view / permute / cat / copy kernels that was
// ~1600 launches and 15 % of the step).  One launch per panel:
//   panel[(k / 4), n, k % 4]  with  k = ((c / 32) * ntaps + tap) * 32 + c % 32   (cin_pad % 32 == 0)   or   tap * cin_pad + c
//   value = W[n][c][kidx[tap]] (transposed = 0: Conv2d forward, ConvTranspose2d data gradient)
//         = W[c][n][kidx[tap]] (transposed = 1: ConvTranspose2d forward, Conv2d data gradient),   0 beyond (cin, nout)
// W is (D0, D1, KH, KW) contiguous, kidx[tap] = ky * KW + kx of the weight slice a GEMM tap reads.
struct LwgTapIdx {
    int kidx[LWG_MAX_TAPS];
};

__device__ __forceinline__ void lwg_korder_decode(int k, int ntaps, int cin_pad, int& tap, int& c) {
    if ((cin_pad & 31) == 0) {
        const int chunk = k / (ntaps * 32), rem = k - chunk * ntaps * 32;
        tap = rem >> 5;
        c = chunk * 32 + (rem & 31);
    } else {
        tap = k / cin_pad;
        c = k - tap * cin_pad;
    }
}

__global__ void lwg_pack_panel_kernel(const float* __restrict__ w, int D1, int KHW, int transposed, LwgTapIdx taps, int ntaps, int cin,
                                      int cin_pad, int nout, int n_pad, int Kp, float* __restrict__ out) {
    const int total = (Kp >> 2) * n_pad;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int k4 = i / n_pad, n = i - k4 * n_pad;
        floatx4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int k = k4 * 4 + kk;
            if (k < ntaps * cin_pad && n < nout) {
                int tap, c;
                lwg_korder_decode(k, ntaps, cin_pad, tap, c);
                if (c < cin) v[kk] = w[((size_t)(transposed ? c : n) * D1 + (transposed ? n : c)) * KHW + taps.kidx[tap]];
            }
        }
        *reinterpret_cast<floatx4*>(out + (size_t)i * 4) = v;
    }
}

extern "C" int lwg_pack_panel_f32(const float* w, int D0, int D1, int KH, int KW, int transposed, const int* kidx, int ntaps, int cin,
                                  int cin_pad, int nout, int n_pad, float* out, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!w || !kidx || !out || ntaps < 1 || ntaps > LWG_MAX_TAPS || cin < 1 || cin > cin_pad || nout < 1 || nout > n_pad)
        return (int)hipErrorInvalidValue;
    if ((transposed ? cin : nout) > D0 || (transposed ? nout : cin) > D1) return (int)hipErrorInvalidValue;
    LwgTapIdx t;
    for (int i = 0; i < ntaps; ++i) {
        if (kidx[i] < 0 || kidx[i] >= KH * KW) return (int)hipErrorInvalidValue;
        t.kidx[i] = kidx[i];
    }
    const int Kp = (ntaps * cin_pad + 31) / 32 * 32;
    const int total = (Kp / 4) * n_pad;
    hipLaunchKernelGGL(lwg_pack_panel_kernel, dim3((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096), dim3(256), 0, stream, w, D1,
                       KH * KW, transposed, t, ntaps, cin, cin_pad, nout, n_pad, Kp, out);
    return (int)hipGetLastError();
}

// Every panel of a training step in ONE launch.  A step re-packs each weight twice (forward panel, data-gradient panel): 337
// launches of ~5 us for the generator + discriminator - 5 % of the step as single launches.  descs (device memory, built once per
// network: the parameters live in flat buffers, so every pointer is stable) lists the panels; workgroup b serves descriptor
// d = the last one with first_block <= b (binary search), one float4 of the panel per thread.
__global__ void lwg_pack_panels_kernel(const LwgPackDesc* __restrict__ descs, int ndesc) {
    int lo = 0, hi = ndesc - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const LwgPackDesc* d = descs + lo;
    const int n_pad = d->n_pad, ntaps = d->ntaps, cin_pad = d->cin_pad, cin = d->cin, nout = d->nout, D1 = d->D1, KHW = d->KHW, tr = d->transposed;
    const int total = (d->Kp >> 2) * n_pad;
    const int i = ((int)blockIdx.x - d->first_block) * 256 + (int)threadIdx.x;
    if (i >= total) return;
    const float* __restrict__ w = d->w;
    if ((cin_pad & 31) == 0) {
        // chunked K order (Kp = ntaps cin_pad): a thread owns one output column x FOUR input channels and walks the taps - per channel it reads
        // (a subset of) KHW consecutive floats, a whole cache sector's worth, instead of 4 bytes of a sector per tap (the launch was bound by
        // the L2 -> L1 traffic of 32-byte sectors used once: 0.32 ms for 350 MB, r05_g); the descriptor's other workgroups have nothing to do
        if (i >= (cin_pad >> 2) * n_pad) return;
        const int cq = i / n_pad, n = i - cq * n_pad, c0 = cq * 4;
        const size_t ob = ((size_t)(c0 >> 5) * ntaps * 8 + ((c0 & 31) >> 2)) * n_pad + n;      // float4 index of tap 0
        size_t src[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) src[kk] = ((size_t)(tr ? c0 + kk : n) * D1 + (tr ? n : c0 + kk)) * KHW;
        for (int tap = 0; tap < ntaps; ++tap) {
            floatx4 v = {0.f, 0.f, 0.f, 0.f};
            if (n < nout) {
                const int kp = d->kidx[tap];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    if (c0 + kk < cin) v[kk] = w[src[kk] + kp];
            }
            *reinterpret_cast<floatx4*>(d->out + (ob + (size_t)tap * 8 * n_pad) * 4) = v;
        }
        return;
    }
    const int k4 = i / n_pad, n = i - k4 * n_pad;
    floatx4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int k = k4 * 4 + kk;
        if (k < ntaps * cin_pad && n < nout) {
            int tap, c;
            lwg_korder_decode(k, ntaps, cin_pad, tap, c);
            if (c < cin) v[kk] = w[((size_t)(tr ? c : n) * D1 + (tr ? n : c)) * KHW + d->kidx[tap]];
        }
    }
    *reinterpret_cast<floatx4*>(d->out + (size_t)i * 4) = v;
}

extern "C" int lwg_pack_panels_f32(const LwgPackDesc* descs_dev, int ndesc, int total_blocks, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!descs_dev || ndesc < 1 || total_blocks < 1) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(lwg_pack_panels_kernel, dim3((unsigned)total_blocks), dim3(256), 0, stream, descs_dev, ndesc);
    return (int)hipGetLastError();
}

// The inverse for weight gradients: dwk (ntaps * cin_pad, n_pad) in the kernel's K order (lwg_conv2d_wgrad_nhwc_f32) ->
// dW[n][c][kidx[tap]] (transposed = 0) or dW[c][n][kidx[tap]] (transposed = 1); weight positions no tap maps to are untouched
// (the four parity launches of a transposed convolution fill disjoint positions of one dW).
__global__ void lwg_unpack_wgrad_kernel(const float* __restrict__ dwk, int D1, int KHW, int transposed, LwgTapIdx taps, int ntaps, int cin,
                                        int cin_pad, int nout, int n_pad, float* __restrict__ dw) {
    const int total = ntaps * cin * nout;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int n = i % nout, r = i / nout;
        const int c = r % cin, tap = r / cin;
        const int k = (cin_pad & 31) == 0 ? ((c >> 5) * ntaps + tap) * 32 + (c & 31) : tap * cin_pad + c;
        dw[((size_t)(transposed ? c : n) * D1 + (transposed ? n : c)) * KHW + taps.kidx[tap]] = dwk[(size_t)k * n_pad + n];
    }
}

extern "C" int lwg_unpack_wgrad_f32(const float* dwk, int D0, int D1, int KH, int KW, int transposed, const int* kidx, int ntaps, int cin,
                                    int cin_pad, int nout, int n_pad, float* dw, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!dwk || !kidx || !dw || ntaps < 1 || ntaps > LWG_MAX_TAPS || cin < 1 || cin > cin_pad || nout < 1 || nout > n_pad)
        return (int)hipErrorInvalidValue;
    if ((transposed ? cin : nout) > D0 || (transposed ? nout : cin) > D1) return (int)hipErrorInvalidValue;
    LwgTapIdx t;
    for (int i = 0; i < ntaps; ++i) {
        if (kidx[i] < 0 || kidx[i] >= KH * KW) return (int)hipErrorInvalidValue;
        t.kidx[i] = kidx[i];
    }
    const int total = ntaps * cin * nout;
    hipLaunchKernelGGL(lwg_unpack_wgrad_kernel, dim3((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096), dim3(256), 0, stream, dwk,
                       D1, KH * KW, transposed, t, ntaps, cin, cin_pad, nout, n_pad, dw);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// MaxPool2d(kernel 2, stride 2) on NHWC, forward and backward (VGG19 perceptual loss of the personalization step,
// criterions/vggloss.py:6-96).  The backward routes each gradient to the FIRST maximum of its window in scan order
// (0,0) (0,1) (1,0) (1,1), as ATen's max_pool2d_with_indices does; H and W even.
__global__ void lwg_maxpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int OH, int OW, int C4) {
    const size_t total = (size_t)B * OH * OW * C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        const size_t p = i / C4;
        const int ox = (int)(p % OW);
        const size_t q = p / OW;
        const int oy = (int)(q % OH), b = (int)(q / OH);
        const floatx4* xb = reinterpret_cast<const floatx4*>(x) + (((size_t)b * 2 * OH + 2 * oy) * 2 * OW + 2 * ox) * C4 + c;
        const floatx4 a = xb[0], bb = xb[C4], cc = xb[(size_t)2 * OW * C4], d = xb[(size_t)2 * OW * C4 + C4];
        floatx4 r;
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = fmaxf(fmaxf(a[k], bb[k]), fmaxf(cc[k], d[k]));
        reinterpret_cast<floatx4*>(y)[i] = r;
    }
}

__global__ void lwg_maxpool2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int B, int OH,
                                        int OW, int C4) {
    const size_t total = (size_t)B * OH * OW * C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        const size_t p = i / C4;
        const int ox = (int)(p % OW);
        const size_t q = p / OW;
        const int oy = (int)(q % OH), b = (int)(q / OH);
        const size_t base = (((size_t)b * 2 * OH + 2 * oy) * 2 * OW + 2 * ox) * C4 + c;
        const size_t off[4] = {0, (size_t)C4, (size_t)2 * OW * C4, (size_t)2 * OW * C4 + C4};
        const floatx4* xb = reinterpret_cast<const floatx4*>(x) + base;
        floatx4 v[4], g = reinterpret_cast<const floatx4*>(dy)[i], o[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { v[t] = xb[off[t]]; o[t] = floatx4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int best = 0;
            float m = v[0][k];
#pragma unroll
            for (int t = 1; t < 4; ++t)
                if (v[t][k] > m) { m = v[t][k]; best = t; }
#pragma unroll
            for (int t = 0; t < 4; ++t) o[t][k] = t == best ? g[k] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) (reinterpret_cast<floatx4*>(dx) + base)[off[t]] = o[t];
    }
}

extern "C" int lwg_maxpool2_fwd_nhwc_f32(const float* x, float* y, int B, int H, int W, int C, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!x || !y || B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C <= 0 || (C & 3)) return (int)hipErrorInvalidValue;
    const size_t total = (size_t)B * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(lwg_maxpool2_fwd_kernel, dim3((unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384)), dim3(256), 0,
                       stream, x, y, B, H / 2, W / 2, C / 4);
    return (int)hipGetLastError();
}

extern "C" int lwg_maxpool2_bwd_nhwc_f32(const float* x, const float* dy, float* dx, int B, int H, int W, int C, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!x || !dy || !dx || B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C <= 0 || (C & 3)) return (int)hipErrorInvalidValue;
    const size_t total = (size_t)B * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(lwg_maxpool2_bwd_kernel, dim3((unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384)), dim3(256), 0,
                       stream, x, dy, dx, B, H / 2, W / 2, C / 4);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Head crop + bilinear resize with the box read ON THE DEVICE (FaceLoss, criterions/faceloss.py:316-341,384-406: imgs[i, :, y0:y1, x0:x1] ->
// F.interpolate(size = (112, 96), bilinear, align_corners = True); the reference reads the box on the host and drops samples whose box is
// empty).  Here the box (min_x, max_x, min_y, max_y as int64, what cal_head_bbox_by_kps leaves on the device) never leaves the GPU: every
// sample gets a crop - zeros and valid[i] = 0 for an empty box - so the step has static shapes and can be captured in a hipGraph.
// Interpolation exactly as torch's upsample_bilinear2d (align_corners): scale = (in - 1) / (out - 1), src = scale * dst, i0 = (int)src,
// i1 = i0 + (i0 < in - 1), l1 = src - i0, value = l0y (l0x v00 + l1x v01) + l1y (l0x v10 + l1x v11).
struct LwgCropGeom { int x0, y0, cw, ch; float sx, sy; };

__device__ __forceinline__ bool lwg_crop_geom(const long long* __restrict__ box, int i, int H, int W, int OH, int OW, LwgCropGeom& g) {
    const long long bx0 = box[4 * i], by0 = box[4 * i + 2];
    long long bx1 = box[4 * i + 1], by1 = box[4 * i + 3];
    if (bx0 == bx1 || by0 == by1 || bx0 < 0 || by0 < 0) return false;       // the reference's own test (faceloss.py:398); negative (wrapping) starts: refused
    // Python slice semantics of imgs[i, :, y0:y1, x0:x1]: a stop beyond the image is clamped to it (the sample is KEPT); a slice that is then
    // empty would make the reference's F.interpolate raise - here the sample is dropped (valid = 0)
    bx1 = bx1 > W ? W : bx1;
    by1 = by1 > H ? H : by1;
    if (bx1 <= bx0 || by1 <= by0) return false;
    g.x0 = (int)bx0; g.y0 = (int)by0; g.cw = (int)(bx1 - bx0); g.ch = (int)(by1 - by0);
    g.sx = OW > 1 ? (float)(g.cw - 1) / (float)(OW - 1) : 0.f;
    g.sy = OH > 1 ? (float)(g.ch - 1) / (float)(OH - 1) : 0.f;
    return true;
}

__global__ __launch_bounds__(256) void lwg_crop_resize_kernel(const float* __restrict__ x, const long long* __restrict__ box, float* __restrict__ y,
                                                              float* __restrict__ valid, int N, int C, int H, int W, int OH, int OW) {
    const int i = blockIdx.y, c = blockIdx.z;
    LwgCropGeom g;
    const bool ok = lwg_crop_geom(box, i, H, W, OH, OW, g);
    if (valid && c == 0 && blockIdx.x == 0 && threadIdx.x == 0) valid[i] = ok ? 1.f : 0.f;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= OH * OW) return;
    float v = 0.f;
    if (ok) {
        const int oy = p / OW, ox = p - oy * OW;
        const float fy = g.sy * (float)oy, fx = g.sx * (float)ox;
        const int iy0 = (int)fy, ix0 = (int)fx;
        const int iy1 = iy0 + (iy0 < g.ch - 1), ix1 = ix0 + (ix0 < g.cw - 1);
        const float ly1 = fy - (float)iy0, lx1 = fx - (float)ix0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
        const float* s = x + ((size_t)i * C + c) * H * W;
        const float v00 = s[(size_t)(g.y0 + iy0) * W + g.x0 + ix0], v01 = s[(size_t)(g.y0 + iy0) * W + g.x0 + ix1];
        const float v10 = s[(size_t)(g.y0 + iy1) * W + g.x0 + ix0], v11 = s[(size_t)(g.y0 + iy1) * W + g.x0 + ix1];
        v = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
    }
    y[((size_t)i * C + c) * OH * OW + p] = v;
}

// dx (N,C,H,W), zero-filled by the caller, += the transposed interpolation of dy (N,C,OH,OW) (atomic adds: up to four output pixels share a source)
__global__ __launch_bounds__(256) void lwg_crop_resize_bwd_kernel(const float* __restrict__ dy, const long long* __restrict__ box, float* __restrict__ dx,
                                                                  int N, int C, int H, int W, int OH, int OW) {
    const int i = blockIdx.y, c = blockIdx.z;
    LwgCropGeom g;
    if (!lwg_crop_geom(box, i, H, W, OH, OW, g)) return;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= OH * OW) return;
    const int oy = p / OW, ox = p - oy * OW;
    const float fy = g.sy * (float)oy, fx = g.sx * (float)ox;
    const int iy0 = (int)fy, ix0 = (int)fx;
    const int iy1 = iy0 + (iy0 < g.ch - 1), ix1 = ix0 + (ix0 < g.cw - 1);
    const float ly1 = fy - (float)iy0, lx1 = fx - (float)ix0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
    const float d = dy[((size_t)i * C + c) * OH * OW + p];
    float* s = dx + ((size_t)i * C + c) * H * W;
    atomicAdd(s + (size_t)(g.y0 + iy0) * W + g.x0 + ix0, ly0 * lx0 * d);
    atomicAdd(s + (size_t)(g.y0 + iy0) * W + g.x0 + ix1, ly0 * lx1 * d);
    atomicAdd(s + (size_t)(g.y0 + iy1) * W + g.x0 + ix0, ly1 * lx0 * d);
    atomicAdd(s + (size_t)(g.y0 + iy1) * W + g.x0 + ix1, ly1 * lx1 * d);
}

extern "C" int lwg_crop_resize_bilinear_f32(const float* x, const long long* box, float* y, float* valid, int N, int C, int H, int W, int OH, int OW,
                                            lwg_stream_t stream_) {
    if (!x || !box || !y || N <= 0 || C <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0 || C > 65535 || N > 65535) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(lwg_crop_resize_kernel, dim3((unsigned)((OH * OW + 255) / 256), (unsigned)N, (unsigned)C), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream_), x, box, y, valid, N, C, H, W, OH, OW);
    return (int)hipGetLastError();
}

extern "C" int lwg_crop_resize_bilinear_bwd_f32(const float* dy, const long long* box, float* dx, int N, int C, int H, int W, int OH, int OW,
                                                lwg_stream_t stream_) {
    if (!dy || !box || !dx || N <= 0 || C <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0 || C > 65535 || N > 65535) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(lwg_crop_resize_bwd_kernel, dim3((unsigned)((OH * OW + 255) / 256), (unsigned)N, (unsigned)C), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream_), dy, box, dx, N, C, H, W, OH, OW);
    return (int)hipGetLastError();
}
