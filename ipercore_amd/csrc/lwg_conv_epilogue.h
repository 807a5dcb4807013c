// Shared epilogue of the implicit-GEMM convolution kernels (fp32 and bf16-operand MFMA variants).
#pragma once
#include "lwg_common.h"
#include "lwg_conv_args.h"

// acc[i][j] are 32x32 D^T tiles: rows = output channels, columns = output pixels (see conv_igemm.hip).
// EA: the activation as a compile-time constant (>= 0) or a.act at run time (-1): lwg_conv_epilogue below resolves it once (lwg_common.h, LwgActC).
template <int TM, int TN, int EPI, int EA>
__device__ __forceinline__ void lwg_conv_epilogue_a(const LwgConvArgs& a, floatx16 (&acc)[TM][TN], int m_base, int n_base,
                                                    int wm, int wn, int lane, int ooy_add, int oox_add) {
    const int khalf = lane >> 5;
    const int HW = a.OH * a.OW;
    // ---- epilogue.  The MFMAs computed D^T (weights as the row operand), so a lane owns ONE output pixel
    // m = lane&31 of each 32x32 tile and 16 channels n = 8*(r>>2) + 4*(lane>>5) + (r&3): four float4 per tile,
    // channel-contiguous in NHWC -> 16-byte stores, pixel index math once per row tile.
    const bool direct = (a.omul == 1) && (a.YH == a.OH) && (a.YW == a.OW);
    const int ncol0 = n_base + wn * TN * 32 + 4 * khalf;  // + 32*j + 8*g
    floatx4 bias4[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            bias4[j][g] = floatx4{0.f, 0.f, 0.f, 0.f};
            if (a.bias) bias4[j][g] = *reinterpret_cast<const floatx4*>(a.bias + ncol0 + 32 * j + 8 * g);
        }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m_base + wm * TM * 32 + i * 32 + (lane & 31);
        if (m >= a.M) continue;
        size_t opix = (size_t)m;
        int bimg = 0;
        if (!direct || EPI == LWG_EPI_SPADE) {
            const int b = m / HW;
            bimg = b;
            if (!direct) {
                const int rem = m - b * HW;
                const int oy = rem / a.OW, ox = rem - oy * a.OW;
                opix = ((size_t)b * a.YH + (oy * a.omul + a.ooy + ooy_add)) * a.YW + (ox * a.omul + a.oox + oox_add);
            }
        }
        if (EPI == LWG_EPI_SPADE) {
            // wave columns [0,32) = gamma, [32,64) = beta of the same 32 channels (host packs them so)
            static_assert(EPI != LWG_EPI_SPADE || TN == 2, "SPADE epilogue needs gamma|beta in one wave");
            const int ch0 = ((n_base + wn * TN * 32) >> 1) + 4 * khalf;
            const float* xr = a.xn + opix * a.YC + ch0;
            const float* mr = a.mean + (size_t)bimg * a.YC + ch0;
            const float* rr = a.rstd + (size_t)bimg * a.YC + ch0;
            float* yr = a.y + opix * a.YC + ch0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const floatx4 xv = *reinterpret_cast<const floatx4*>(xr + 8 * g);
                const floatx4 mu = *reinterpret_cast<const floatx4*>(mr + 8 * g);
                const floatx4 rs = *reinterpret_cast<const floatx4*>(rr + 8 * g);
                floatx4 o;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float gm = acc[i][0][4 * g + c] + bias4[0][g][c];
                    const float bt = acc[i][TN - 1][4 * g + c] + bias4[TN - 1][g][c];
                    o[c] = lwg_act_c<EA>((xv[c] - mu[c]) * rs[c] * (1.f + gm) + bt, a.act);
                }
                *reinterpret_cast<floatx4*>(yr + 8 * g) = o;
            }
        } else if (EPI == LWG_EPI_NONE && a.ydt == LWG_DT_F32_Q4) {
            // channel-quad planes (B, YC/4, YH, YW, 4): this lane's quads of its pixel; 32 consecutive lanes = 32 output pixels of a row
            // (every second one in a transposed convolution's parity launch) -> 16-byte stores side by side instead of 256 bytes apart
            const int b = m / HW;
            const int rem = m - b * HW;
            const int oy = rem / a.OW, ox = rem - oy * a.OW;
            const size_t plane = (size_t)a.YH * a.YW;
            const size_t pix = (size_t)(oy * a.omul + a.ooy + ooy_add) * a.YW + (ox * a.omul + a.oox + oox_add);
            float* yq = a.y + (((size_t)b * (a.YC >> 2) + ((a.ycoff + ncol0) >> 2)) * plane + pix) * 4;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    floatx4 o;
#pragma unroll
                    for (int c = 0; c < 4; ++c) o[c] = lwg_act_c<EA>(acc[i][j][4 * g + c] + bias4[j][g][c], a.act);
                    *reinterpret_cast<floatx4*>(yq + (size_t)(8 * j + 2 * g) * plane * 4) = o;
                }
        } else {
            float* yr = a.y + opix * a.YC + a.ycoff + ncol0;
            const float* rr = EPI == LWG_EPI_RESIDUAL ? a.res + opix * a.YC + a.ycoff + ncol0 : nullptr;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    floatx4 o;
                    floatx4 rv = floatx4{0.f, 0.f, 0.f, 0.f};
                    if (EPI == LWG_EPI_RESIDUAL) rv = *reinterpret_cast<const floatx4*>(rr + 32 * j + 8 * g);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float v = acc[i][j][4 * g + c] + bias4[j][g][c];
                        o[c] = (EPI == LWG_EPI_RESIDUAL && lwg_act_is_mask<EA>(a.act)) ? (rv[c] > 0.f ? v : 0.f) : lwg_act_c<EA>(v + rv[c], a.act);
                    }
                    if (EPI == LWG_EPI_NONE && a.ydt == LWG_DT_BF16) {      // first layer of the bf16 mode: fp32 in, bf16 NHWC out
                        typedef __bf16 lwg_bf16x4 __attribute__((ext_vector_type(4)));
                        lwg_bf16x4 ob;
#pragma unroll
                        for (int c = 0; c < 4; ++c) ob[c] = (__bf16)o[c];
                        *reinterpret_cast<lwg_bf16x4*>(reinterpret_cast<__bf16*>(a.y) + opix * a.YC + a.ycoff + ncol0 + 32 * j + 8 * g) = ob;
                    } else {
                        *reinterpret_cast<floatx4*>(yr + 32 * j + 8 * g) = o;
                    }
                }
        }
    }
}

template <int TM, int TN, int EPI>
__device__ __forceinline__ void lwg_conv_epilogue(const LwgConvArgs& a, floatx16 (&acc)[TM][TN], int m_base, int n_base,
                                                  int wm, int wn, int lane, int ooy_add = 0, int oox_add = 0) {
    if (a.act == LWG_ACT_RELU) lwg_conv_epilogue_a<TM, TN, EPI, LWG_ACT_RELU>(a, acc, m_base, n_base, wm, wn, lane, ooy_add, oox_add);
    else if (a.act == LWG_ACT_NONE) lwg_conv_epilogue_a<TM, TN, EPI, LWG_ACT_NONE>(a, acc, m_base, n_base, wm, wn, lane, ooy_add, oox_add);
    else if (EPI == LWG_EPI_RESIDUAL && a.act == LWG_ACT_RELU_MASK)
        lwg_conv_epilogue_a<TM, TN, EPI, LWG_ACT_RELU_MASK>(a, acc, m_base, n_base, wm, wn, lane, ooy_add, oox_add);
    else lwg_conv_epilogue_a<TM, TN, EPI, -1>(a, acc, m_base, n_base, wm, wn, lane, ooy_add, oox_add);
}

// Split-K launches: the raw partial sums of one K slice as a dense (M, N) slab (no bias / activation / output geometry; the
// finishing kernel applies those once the slices are added up).
template <int TM, int TN>
__device__ __forceinline__ void lwg_conv_epilogue_slab(float* __restrict__ slab, int M, int N, floatx16 (&acc)[TM][TN], int m_base,
                                                       int n_base, int wm, int wn, int lane) {
    const int ncol0 = n_base + wn * TN * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m_base + wm * TM * 32 + i * 32 + (lane & 31);
        if (m >= M) continue;
        float* yr = slab + (size_t)m * N + ncol0;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                floatx4 o;
#pragma unroll
                for (int c = 0; c < 4; ++c) o[c] = acc[i][j][4 * g + c];
                *reinterpret_cast<floatx4*>(yr + 32 * j + 8 * g) = o;
            }
    }
}
