// InstanceNorm2d(affine=False, eps) statistics and application on NHWC fp32 - HBM-bound streaming kernels.
// Replaces nn.InstanceNorm2d at reference generators/attlwb_spade_resunet.py:62,:83 (SPADE's parameter-free
// norm; the normalisation itself is fused into the gamma/beta conv epilogue) and bg_inpaintor.py:14-51.
//
// Layout: x is (B, HW, C); a wave reads 64 consecutive channels of one pixel (256 contiguous bytes).
// Pass 1: grid (C/64, nsplit, B), 4 waves stride over the pixels of one split, shifted sums (shift = first
//         pixel of the split) to avoid E[x^2]-E[x]^2 cancellation; per-split (n, mean, M2) to scratch.
// Pass 2: Chan combination of the nsplit partials -> mean, rstd = 1/sqrt(M2/n + eps) (biased variance).
#include "lwg_common.h"
#include "lwg_conv_args.h"

__global__ __launch_bounds__(256) void lwg_in_stats_partial(const float* __restrict__ x, int HW, int C, int nsplit,
                                                           float* __restrict__ ws) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane, split = blockIdx.y, b = blockIdx.z;
    const bool cok = c < C;
    const int per = (HW + nsplit - 1) / nsplit;
    const int p0 = split * per, p1 = min(HW, p0 + per);
    const float* xb = x + (size_t)b * HW * C;
    float shift = 0.f, s1 = 0.f, s2 = 0.f;
    int n = 0;
    if (cok && p0 < p1) shift = xb[(size_t)p0 * C + c];
    if (cok) {
        for (int p = p0 + wid; p < p1; p += 4) {
            const float d = xb[(size_t)p * C + c] - shift;
            s1 += d;
            s2 += d * d;
            ++n;
        }
    }
    __shared__ float sh[3][4][64];
    sh[0][wid][lane] = s1;
    sh[1][wid][lane] = s2;
    sh[2][wid][lane] = (float)n;
    __syncthreads();
    if (wid == 0 && cok) {
        float t1 = 0.f, t2 = 0.f, tn = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { t1 += sh[0][w][lane]; t2 += sh[1][w][lane]; tn += sh[2][w][lane]; }
        float mean = shift, m2 = 0.f;
        if (tn > 0.f) { mean = shift + t1 / tn; m2 = t2 - t1 * t1 / tn; }
        float* o = ws + (((size_t)b * nsplit + split) * C + c) * 3;
        o[0] = tn; o[1] = mean; o[2] = m2 > 0.f ? m2 : 0.f;
    }
}

// Vectorised pass 1 for C in {64, 128, 256} (every site of the generator): a lane owns 4 channels (16-byte loads), LPP =
// C/4 lanes cover one pixel, the 256 threads cover PG = 256/LPP pixels per iteration, four iterations in flight.
// grid (nsplit, B).  Same shifted-sum algebra; the PG partials are combined in a fixed order (deterministic).
template <int LPP>
__global__ __launch_bounds__(256) void lwg_in_stats_partial4(const float* __restrict__ x, int HW, int nsplit,
                                                            float* __restrict__ ws) {
    constexpr int C = LPP * 4, PG = 256 / LPP;
    const int cq = threadIdx.x % LPP, pg = threadIdx.x / LPP;
    const int split = blockIdx.x, b = blockIdx.y;
    const int per = (HW + nsplit - 1) / nsplit;
    const int p0 = split * per, p1 = min(HW, p0 + per);
    const floatx4* xb = reinterpret_cast<const floatx4*>(x + (size_t)b * HW * C) + cq;
    floatx4 shift = {0.f, 0.f, 0.f, 0.f}, s1 = shift, s2 = shift;
    int n = 0;
    if (p0 < p1) shift = xb[(size_t)p0 * LPP];
    int p = p0 + pg;
    for (; p + 3 * PG < p1; p += 4 * PG) {
        floatx4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = xb[(size_t)(p + u * PG) * LPP];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const floatx4 d = v[u] - shift;
            s1 += d;
            s2 += d * d;
        }
        n += 4;
    }
    for (; p < p1; p += PG) {
        const floatx4 d = xb[(size_t)p * LPP] - shift;
        s1 += d;
        s2 += d * d;
        ++n;
    }
    __shared__ floatx4 sh1[PG][LPP], sh2[PG][LPP];
    __shared__ int shn[PG];
    sh1[pg][cq] = s1;
    sh2[pg][cq] = s2;
    if (cq == 0) shn[pg] = n;
    __syncthreads();
    if (pg == 0) {
        floatx4 t1 = {0.f, 0.f, 0.f, 0.f}, t2 = t1;
        float tn = 0.f;
#pragma unroll
        for (int g = 0; g < PG; ++g) { t1 += sh1[g][cq]; t2 += sh2[g][cq]; tn += (float)shn[g]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float mean = shift[k], m2 = 0.f;
            if (tn > 0.f) { mean = shift[k] + t1[k] / tn; m2 = t2[k] - t1[k] * t1[k] / tn; }
            float* o = ws + (((size_t)b * nsplit + split) * C + cq * 4 + k) * 3;
            o[0] = tn; o[1] = mean; o[2] = m2 > 0.f ? m2 : 0.f;
        }
    }
}

// Pass 2: one wave per (image, channel), lane = split.  The nsplit <= 64 partial records (count, mean, M2) are merged with Chan's
// pairwise update as a fixed-shape shuffle tree (6 rounds; the first version folded them sequentially in one lane: 64 dependent
// divisions, 15 us per launch and 54 launches per frame batch).  Deterministic: the tree depends on nsplit only.
__global__ __launch_bounds__(256) void lwg_in_stats_final(const float* __restrict__ ws, int BC, int C, int nsplit, float eps,
                                                         float* __restrict__ mean, float* __restrict__ rstd) {
    const int ch = threadIdx.x >> 6, sp = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + ch;
    if (i >= BC) return;                                         // wave-uniform
    const int b = i / C, c = i - b * C;
    float n = 0.f, mu = 0.f, m2 = 0.f;
    for (int s = sp; s < nsplit; s += 64) {                      // one record per lane on every call site (nsplit <= 64)
        const float* o = ws + (((size_t)b * nsplit + s) * C + c) * 3;
        const float nb = o[0];
        if (nb <= 0.f) continue;
        const float tot = n + nb, delta = o[1] - mu, r = nb / tot;
        mu += delta * r;
        m2 += o[2] + delta * delta * (n * r);
        n = tot;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float nb = __shfl_down(n, off, 64), mub = __shfl_down(mu, off, 64), m2b = __shfl_down(m2, off, 64);
        if (nb > 0.f) {
            const float tot = n + nb, delta = mub - mu, r = nb / tot;
            mu += delta * r;
            m2 += m2b + delta * delta * (n * r);
            n = tot;
        }
    }
    if (sp == 0) {
        mean[i] = mu;
        rstd[i] = 1.0f / sqrtf(m2 / n + eps);
    }
}

typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lwg_unpack8(const uintx4 v, float (&f)[8]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f[2 * k] = __builtin_bit_cast(float, v[k] << 16);
        f[2 * k + 1] = __builtin_bit_cast(float, v[k] & 0xffff0000u);
    }
}

// ---------------------------------------------------------------------------------------------- InstanceNorm statistics
// Pass 1 of csrc/norm.hip on a bf16 (B, HW, C) tensor, C in {64, 128, 256}: a lane owns 8 channels (one 16-byte load), LPP = C/8
// lanes cover a pixel, PG = 256/LPP pixels per iteration, four iterations in flight; shifted sums; records (n, mean, M2) per
// (image, split, channel) in the layout lwg_in_stats_final (norm.hip) merges.  grid (nsplit, B).
template <int LPP>
__global__ __launch_bounds__(256) void lwg_in_stats_partial8_bf16(const __bf16* __restrict__ x, int HW, int nsplit, float* __restrict__ ws) {
    constexpr int C = LPP * 8, PG = 256 / LPP;
    const int cq = threadIdx.x % LPP, pg = threadIdx.x / LPP;
    const int split = blockIdx.x, b = blockIdx.y;
    const int per = (HW + nsplit - 1) / nsplit;
    const int p0 = split * per, p1 = min(HW, p0 + per);
    const uintx4* xb = reinterpret_cast<const uintx4*>(x + (size_t)b * HW * C) + cq;
    float shift[8], s1[8], s2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) shift[k] = s1[k] = s2[k] = 0.f;
    int n = 0;
    if (p0 < p1) lwg_unpack8(xb[(size_t)p0 * LPP], shift);
    int p = p0 + pg;
    for (; p + 3 * PG < p1; p += 4 * PG) {
        uintx4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = xb[(size_t)(p + u * PG) * LPP];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float f[8];
            lwg_unpack8(v[u], f);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float d = f[k] - shift[k];
                s1[k] += d;
                s2[k] += d * d;
            }
        }
        n += 4;
    }
    for (; p < p1; p += PG) {
        float f[8];
        lwg_unpack8(xb[(size_t)p * LPP], f);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float d = f[k] - shift[k];
            s1[k] += d;
            s2[k] += d * d;
        }
        ++n;
    }
    __shared__ float sh1[PG][LPP][8], sh2[PG][LPP][8];
    __shared__ int shn[PG];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        sh1[pg][cq][k] = s1[k];
        sh2[pg][cq][k] = s2[k];
    }
    if (cq == 0) shn[pg] = n;
    __syncthreads();
    if (pg == 0) {
        float tn = 0.f;
#pragma unroll
        for (int g = 0; g < PG; ++g) tn += (float)shn[g];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int g = 0; g < PG; ++g) { t1 += sh1[g][cq][k]; t2 += sh2[g][cq][k]; }
            float mean = shift[k], m2 = 0.f;
            if (tn > 0.f) { mean = shift[k] + t1 / tn; m2 = t2 - t1 * t1 / tn; }
            float* o = ws + (((size_t)b * nsplit + split) * C + cq * 8 + k) * 3;
            o[0] = tn; o[1] = mean; o[2] = m2 > 0.f ? m2 : 0.f;
        }
    }
}

extern "C" int lwg_instnorm_stats_nhwc_bf16(const void* x, int B, int HW, int C, float eps, float* mean, float* rstd, float* ws,
                                            int nsplit, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!x || !mean || !rstd || !ws || B <= 0 || HW <= 0 || nsplit <= 0 || nsplit > 64 || B > 65535) return (int)hipErrorInvalidValue;
    const __bf16* xb = reinterpret_cast<const __bf16*>(x);
    if (C == 64)
        hipLaunchKernelGGL(lwg_in_stats_partial8_bf16<8>, dim3(nsplit, B), dim3(256), 0, stream, xb, HW, nsplit, ws);
    else if (C == 128)
        hipLaunchKernelGGL(lwg_in_stats_partial8_bf16<16>, dim3(nsplit, B), dim3(256), 0, stream, xb, HW, nsplit, ws);
    else if (C == 256)
        hipLaunchKernelGGL(lwg_in_stats_partial8_bf16<32>, dim3(nsplit, B), dim3(256), 0, stream, xb, HW, nsplit, ws);
    else
        return (int)hipErrorInvalidValue;
    const int BC = B * C;
    hipLaunchKernelGGL(lwg_in_stats_final, dim3((BC + 3) / 4), dim3(256), 0, stream, ws, BC, C, nsplit, eps, mean, rstd);
    return (int)hipGetLastError();
}


__global__ __launch_bounds__(256) void lwg_in_apply(const floatx4* __restrict__ x, const float* __restrict__ mean,
                                                   const float* __restrict__ rstd, const floatx4* __restrict__ res,
                                                   floatx4* __restrict__ y, int HW, int C4, size_t total4, int act) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const int b = (int)(i / ((size_t)HW * C4));
        const floatx4 v = x[i];
        const floatx4 mu = *reinterpret_cast<const floatx4*>(mean + (size_t)b * C4 * 4 + c4 * 4);
        const floatx4 rs = *reinterpret_cast<const floatx4*>(rstd + (size_t)b * C4 * 4 + c4 * 4);
        floatx4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = lwg_act((v[k] - mu[k]) * rs[k], act);
        if (res) {
            const floatx4 r = res[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] += r[k];
        }
        y[i] = o;
    }
}

extern "C" int lwg_instnorm_stats_nhwc_f32(const float* x, int B, int HW, int C, float eps, float* mean, float* rstd,
                                           float* ws, int nsplit, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!x || !mean || !rstd || !ws || B <= 0 || HW <= 0 || C <= 0 || nsplit <= 0 || nsplit > 65535)
        return (int)hipErrorInvalidValue;
    if (C == 64)
        hipLaunchKernelGGL(lwg_in_stats_partial4<16>, dim3(nsplit, B), dim3(256), 0, stream, x, HW, nsplit, ws);
    else if (C == 128)
        hipLaunchKernelGGL(lwg_in_stats_partial4<32>, dim3(nsplit, B), dim3(256), 0, stream, x, HW, nsplit, ws);
    else if (C == 256)
        hipLaunchKernelGGL(lwg_in_stats_partial4<64>, dim3(nsplit, B), dim3(256), 0, stream, x, HW, nsplit, ws);
    else
        hipLaunchKernelGGL(lwg_in_stats_partial, dim3((C + 63) / 64, nsplit, B), dim3(256), 0, stream, x, HW, C, nsplit, ws);
    const int BC = B * C;
    hipLaunchKernelGGL(lwg_in_stats_final, dim3((BC + 3) / 4), dim3(256), 0, stream, ws, BC, C, nsplit, eps, mean, rstd);
    return (int)hipGetLastError();
}

extern "C" int lwg_instnorm_apply_nhwc_f32(const float* x, const float* mean, const float* rstd, const float* res,
                                           float* y, int B, int HW, int C, int act, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!x || !mean || !rstd || !y || (C & 3) || B <= 0) return (int)hipErrorInvalidValue;
    const size_t total4 = (size_t)B * HW * (C / 4);
    const int blocks = (int)((total4 + 255) / 256 < 4096 ? (total4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(lwg_in_apply, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const floatx4*>(x), mean, rstd,
                       reinterpret_cast<const floatx4*>(res), reinterpret_cast<floatx4*>(y), HW, C / 4, total4, act);
    return (int)hipGetLastError();
}
