// NHWC implicit-GEMM convolution, bf16 end to end: bf16 activations in HBM, bf16 MFMA operands
// (v_mfma_f32_32x32x16_bf16), fp32 accumulation and epilogue arithmetic, bf16 output - BASELINE configs[3]
// ("novel_view 1024x1024 bf16, MFMA bf16 conv tiles").  Replaces the same torch.nn.Conv2d / ConvTranspose2d calls as
// csrc/conv_igemm.hip (reference generators/attlwb_spade_resunet.py:14-25, :73-93, :202-204, :268-271, :331-340).
//
// The kernels of this file, in the order they were built (DESIGN.md 3.11 has the measurements behind each step):
//   lwg_conv_bf16_kernel        general launches (strided convs, 1x1 with N != Cin): both operands global -> LDS by LDS-DMA
//                               (buffer_load_dwordx4 ... lds, 16 B per lane, no VGPR round trip); a K-step = 64 channels of one tap = one
//                               128-byte line per pixel; 128 x 128 / 128 x 64 tile per 4-wave workgroup or 256 x 256 per 8 waves;
//   lwg_conv_bf16_hr2_kernel    the 3x3 and 2x2-tap stride-1 launches: the 8 x 16-pixel block's halo staged once per channel chunk,
//                               weights streamed global -> register ring (one barrier per chunk), ROW RENAMING - a wave's row tile
//                               is the image-row pair (i, i + TM), so one fragment read feeds all vertical taps;
//   lwg_conv_bf16_up4_kernel    ConvTranspose2d(4, 2, 1) with Cin <= 128 as ONE launch (four parities, input block staged once);
//   lwg_conv_bf16_pw_kernel     1x1 C -> C (query projections): weights resident in registers, persistent workgroups;
//   lwg_conv_c8_bf16_kernel     first layer: fp32 NHWC-8 input converted in registers, no LDS.
// Shared by all of them:
//   * the LDS image of an activation / weight row is [row][128 B] with the eight 16-byte k-octets at slot octet ^ key - key =
//     ((row >> 1) & 7) for linear tiles, ((halo column >> 1) & 7) for halo tiles - applied to the per-lane GLOBAL address (or baked
//     into the packed panel), the LDS destination stays lane-linear as the DMA requires: every ds_read_b128 is conflict-free;
//   * padding pixels use an out-of-range buffer offset: loads / LDS-DMA return zeros (probed: tools/probes/dma_probe.hip);
//   * D^T accumulators (weights as the row operand): a lane owns one pixel and 4 consecutive channels per 8-channel group; the
//     two half-waves exchange halves (v_permlane32_swap) so a lane stores 8 consecutive channels = one 16-byte bf16 store.
// Epilogues: bias, ReLU / tanh / sigmoid, residual add, SPADE's IN(x) * (1 + gamma) + beta - all read / written as bf16.
#include "lwg_common.h"
#include "lwg_conv_args.h"
#ifndef LWG_BF16_NT_ST
#define LWG_BF16_NT_ST 0     // lab: the epilogue's 16-byte output stores non-temporal (worth 0.5 % in conv_winograd4.hip)
#endif
#include "lwg_conv_slices.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));
typedef unsigned int uintx2 __attribute__((ext_vector_type(2)));

#define LWG_OOB_OFFSET 0xC0000000u
#define LWG_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ unsigned lwg_pack_bf16x2(float lo, float hi) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float lwg_bf16_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float lwg_bf16_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

// D^T accumulator tiles -> bf16 NHWC with the fused epilogue (shared by the kernel variants below).
// SPATIAL = 0: GEMM row m_base + r is output position (b, oy, ox) in row-major order.  SPATIAL = 1 (the halo-tile kernels):
// the workgroup's 128 rows are the 8 x 16 pixel block whose corner (image tb, row ty0, column tx0) the caller passes; row r is
// pixel (ty0 + r / 16, tx0 + r % 16) and rows outside the image are dead.
// EA: the activation as a compile-time constant (>= 0) or a.act at run time (-1) - lwg_bf16_epilogue below resolves it once per workgroup
// (lwg_common.h, LwgActC: a runtime code inside the innermost loops costs a uniform branch ladder per group of outputs).
// pre (halo-tile kernels only; nullptr: read from global memory here): the workgroup's epilogue operands parked in LDS by the kernel's prologue -
// [0, BN): bias of columns n_base ..; SPADE: [BN, BN + BN/2) mean, [BN + BN/2, 2 BN) rstd of image tb, channels n_base / 2 ..
template <int TM, int TN, int EPI, int SPATIAL, int EA>
__device__ __forceinline__ void lwg_bf16_epilogue_a(const LwgConvArgs& a, floatx16 (&acc)[TM][TN], int m_base, int n_base, int wm, int wn,
                                                    int lane, int tb, int ty0, int tx0, const float* pre = nullptr, int pre_bn = 0) {
    const int khalf = lane >> 5;
    const int HW = a.OH * a.OW;
    // ---- epilogue: D^T tiles -> bf16 NHWC.  acc[i][j][4*g + c] = pixel (lane & 31) of row tile i, channel 8*g + 4*khalf + c of
    // column tile j: a lane owns 4 of every 8 channels.  The two half-waves exchange halves (v_permlane32_swap on packed bf16
    // pairs) so that lanes 0..31 hold channels [0,16) and lanes 32..63 channels [16,32) of the 32-channel tile, 8 consecutive
    // channels per register quad: every access to y / res / xn is a 16-byte access.  swap(x, y) trades x's upper half-wave with
    // y's lower half-wave and is its own inverse, so tensors READ in the store layout are brought to the D^T layout the same way.
    const bool direct = (a.omul == 1) && (a.YH == a.OH) && (a.YW == a.OW);
    __bf16* const yb = reinterpret_cast<__bf16*>(a.y);
    const __bf16* const resb = reinterpret_cast<const __bf16*>(a.res);
    const __bf16* const xnb = reinterpret_cast<const __bf16*>(a.xn);
    // 32 channels of one pixel in the store layout (this lane: 16 of them) -> fp32 in the D^T layout f[g][c]
    auto load_dt = [&](const __bf16* p16, bool live, float (&f)[4][4]) {
        uintx4 L[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
        if (live) {
            L[0] = *reinterpret_cast<const uintx4*>(p16 + 16 * khalf);
            L[1] = *reinterpret_cast<const uintx4*>(p16 + 16 * khalf + 8);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                auto sres = __builtin_amdgcn_permlane32_swap(L[h][d], L[h][2 + d], false, false);
                f[h][2 * d] = lwg_bf16_lo(sres[0]); f[h][2 * d + 1] = lwg_bf16_hi(sres[0]);
                f[2 + h][2 * d] = lwg_bf16_lo(sres[1]); f[2 + h][2 * d + 1] = lwg_bf16_hi(sres[1]);
            }
    };
    auto store_dt = [&](__bf16* p16, bool live, const float (&o)[4][4]) {
        uintx4 st[2];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                auto sres = __builtin_amdgcn_permlane32_swap(lwg_pack_bf16x2(o[h][2 * d], o[h][2 * d + 1]),
                                                             lwg_pack_bf16x2(o[2 + h][2 * d], o[2 + h][2 * d + 1]), false, false);
                st[h][d] = sres[0];
                st[h][2 + d] = sres[1];
            }
        if (live) {
#if LWG_BF16_NT_ST
            __builtin_nontemporal_store(st[0], reinterpret_cast<uintx4*>(p16 + 16 * khalf));
            __builtin_nontemporal_store(st[1], reinterpret_cast<uintx4*>(p16 + 16 * khalf + 8));
#else
            *reinterpret_cast<uintx4*>(p16 + 16 * khalf) = st[0];
            *reinterpret_cast<uintx4*>(p16 + 16 * khalf + 8) = st[1];
#endif
        }
    };
    if constexpr (TN == 1 && SPATIAL != 0) {
        // The halo-tile kernels (one 32-column tile per wave, a workgroup inside ONE image): EVERY load of the epilogue - the bias quads (once, not once
        // per row tile), the residual / xn rows of all TM row tiles, SPADE's statistics - is issued BEFORE the first store.  The memory counter retires
        // in order: with the loads of row tile i + 1 behind the stores of row tile i (rounds 2-5) every row tile waited for the previous one's stores
        // to be acknowledged AND for an L2 round trip of its own - the tile time that did not scale with K.
        bool live[TM];
        size_t opix[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int r = wm * TM * 32 + i * 32 + (lane & 31);
            const int oy = ty0 + (SPATIAL == 2 ? wm * 2 * TM + i + TM * ((lane >> 4) & 1) : (r >> 4)), ox = tx0 + (r & 15);
            live[i] = oy < a.OH && ox < a.OW;
            opix[i] = live[i] ? ((size_t)tb * a.YH + (oy * a.omul + a.ooy)) * a.YW + (ox * a.omul + a.oox) : 0;
        }
        const int ncol = n_base + wn * 32;
        if constexpr (EPI == LWG_EPI_SPADE) {
            const int chb = ncol >> 1;                              // first of the wave's 16 output channels (gamma | beta interleaved in blocks of 16)
            uintx2 xv[TM][2];
            floatx4 mu[2], rs[2], bg4[2], bb4[2];
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int ch = chb + 8 * g + 4 * khalf;
                if (pre) {
                    mu[g] = *reinterpret_cast<const floatx4*>(pre + pre_bn + wn * 16 + 8 * g + 4 * khalf);
                    rs[g] = *reinterpret_cast<const floatx4*>(pre + pre_bn + (pre_bn >> 1) + wn * 16 + 8 * g + 4 * khalf);
                    bg4[g] = *reinterpret_cast<const floatx4*>(pre + wn * 32 + 8 * g + 4 * khalf);
                    bb4[g] = *reinterpret_cast<const floatx4*>(pre + wn * 32 + 16 + 8 * g + 4 * khalf);
                } else {
                    mu[g] = *reinterpret_cast<const floatx4*>(a.mean + (size_t)tb * a.YC + ch);
                    rs[g] = *reinterpret_cast<const floatx4*>(a.rstd + (size_t)tb * a.YC + ch);
                    bg4[g] = *reinterpret_cast<const floatx4*>(a.bias + ncol + 8 * g + 4 * khalf);
                    bb4[g] = *reinterpret_cast<const floatx4*>(a.bias + ncol + 16 + 8 * g + 4 * khalf);
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                xv[i][0] = xv[i][1] = uintx2{0u, 0u};
                if (live[i]) {
                    xv[i][0] = *reinterpret_cast<const uintx2*>(xnb + opix[i] * a.YC + chb + 4 * khalf);
                    xv[i][1] = *reinterpret_cast<const uintx2*>(xnb + opix[i] * a.YC + chb + 8 + 4 * khalf);
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const float xf[4] = {lwg_bf16_lo(xv[i][g][0]), lwg_bf16_hi(xv[i][g][0]), lwg_bf16_lo(xv[i][g][1]), lwg_bf16_hi(xv[i][g][1])};
                    float o[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float gm = acc[i][0][4 * g + c] + bg4[g][c];
                        const float bt = acc[i][0][4 * (g + 2) + c] + bb4[g][c];
                        o[c] = lwg_act_c<EA>((xf[c] - mu[g][c]) * rs[g][c] * (1.f + gm) + bt, a.act);
                    }
                    if (live[i]) {
                        uintx2 pk;
                        pk[0] = lwg_pack_bf16x2(o[0], o[1]);
                        pk[1] = lwg_pack_bf16x2(o[2], o[3]);
                        *reinterpret_cast<uintx2*>(yb + opix[i] * a.YC + chb + 8 * g + 4 * khalf) = pk;
                    }
                }
        } else {
            floatx4 b4[4];
            if (pre) {
#pragma unroll
                for (int g = 0; g < 4; ++g) b4[g] = *reinterpret_cast<const floatx4*>(pre + wn * 32 + 8 * g + 4 * khalf);
            } else if (a.bias) {                                    // (ONE branch around the four loads: four quads in flight, one wait)
#pragma unroll
                for (int g = 0; g < 4; ++g) b4[g] = *reinterpret_cast<const floatx4*>(a.bias + ncol + 8 * g + 4 * khalf);
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) b4[g] = floatx4{0.f, 0.f, 0.f, 0.f};
            }
            uintx4 L[TM][2];
            if constexpr (EPI == LWG_EPI_RESIDUAL) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    L[i][0] = L[i][1] = uintx4{0u, 0u, 0u, 0u};
                    if (live[i]) {
                        const __bf16* p16 = resb + opix[i] * a.YC + a.ycoff + ncol;
                        L[i][0] = *reinterpret_cast<const uintx4*>(p16 + 16 * khalf);
                        L[i][1] = *reinterpret_cast<const uintx4*>(p16 + 16 * khalf + 8);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                float r4[4][4], o[4][4];
                if constexpr (EPI == LWG_EPI_RESIDUAL) {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int d = 0; d < 2; ++d) {
                            auto sres = __builtin_amdgcn_permlane32_swap(L[i][h][d], L[i][h][2 + d], false, false);
                            r4[h][2 * d] = lwg_bf16_lo(sres[0]); r4[h][2 * d + 1] = lwg_bf16_hi(sres[0]);
                            r4[2 + h][2 * d] = lwg_bf16_lo(sres[1]); r4[2 + h][2 * d + 1] = lwg_bf16_hi(sres[1]);
                        }
                }
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        o[g][c] = lwg_act_c<EA>(acc[i][0][4 * g + c] + b4[g][c] + (EPI == LWG_EPI_RESIDUAL ? r4[g][c] : 0.f), a.act);
                store_dt(yb + opix[i] * a.YC + a.ycoff + ncol, live[i], o);
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        bool live;
        size_t opix;
        int bimg = 0;
        if (SPATIAL) {
            const int r = wm * TM * 32 + i * 32 + (lane & 31);
            // SPATIAL == 2 (the row-renaming kernel): row tile i of a wave = image rows (i, i + TM) of its 2*TM rows
            const int oy = ty0 + (SPATIAL == 2 ? wm * 2 * TM + i + TM * ((lane >> 4) & 1) : (r >> 4)), ox = tx0 + (r & 15);
            live = oy < a.OH && ox < a.OW;
            bimg = tb;
            opix = live ? ((size_t)tb * a.YH + (oy * a.omul + a.ooy)) * a.YW + (ox * a.omul + a.oox) : 0;
        } else {
            const int m = m_base + wm * TM * 32 + i * 32 + (lane & 31);
            live = m < a.M;
            const int mm = live ? m : 0;
            opix = (size_t)mm;
            if (!direct || EPI == LWG_EPI_SPADE) {
                const int b = mm / HW;
                bimg = b;
                if (!direct) {
                    const int rem = mm - b * HW;
                    const int oy = rem / a.OW, ox = rem - oy * a.OW;
                    opix = ((size_t)b * a.YH + (oy * a.omul + a.ooy)) * a.YW + (ox * a.omul + a.oox);
                }
            }
        }
        if (EPI == LWG_EPI_SPADE && TN == 1) {
            // register-streamed-weights kernel: a wave owns ONE 32-column tile = gamma | beta of 16 output channels (the host
            // interleaves the SPADE panel in blocks of 16 for it): D^T groups g = 0, 1 are gamma of channels 8g + 4*khalf + c,
            // groups 2, 3 beta of the same channels
            const int chb = (n_base + wn * 32) >> 1;                // first of the wave's 16 output channels
            uintx2 xv[2] = {{0u, 0u}, {0u, 0u}};
            if (live) {
                xv[0] = *reinterpret_cast<const uintx2*>(xnb + opix * a.YC + chb + 4 * khalf);
                xv[1] = *reinterpret_cast<const uintx2*>(xnb + opix * a.YC + chb + 8 + 4 * khalf);
            }
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int ch = chb + 8 * g + 4 * khalf;
                const floatx4 mu = *reinterpret_cast<const floatx4*>(a.mean + (size_t)bimg * a.YC + ch);
                const floatx4 rs = *reinterpret_cast<const floatx4*>(a.rstd + (size_t)bimg * a.YC + ch);
                const floatx4 bg4 = *reinterpret_cast<const floatx4*>(a.bias + n_base + wn * 32 + 8 * g + 4 * khalf);
                const floatx4 bb4 = *reinterpret_cast<const floatx4*>(a.bias + n_base + wn * 32 + 16 + 8 * g + 4 * khalf);
                const float xf[4] = {lwg_bf16_lo(xv[g][0]), lwg_bf16_hi(xv[g][0]), lwg_bf16_lo(xv[g][1]), lwg_bf16_hi(xv[g][1])};
                float o[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float gm = acc[i][0][4 * g + c] + bg4[c];
                    const float bt = acc[i][0][4 * (g + 2) + c] + bb4[c];
                    o[c] = lwg_act_c<EA>((xf[c] - mu[c]) * rs[c] * (1.f + gm) + bt, a.act);
                }
                if (live) {
                    uintx2 pk;
                    pk[0] = lwg_pack_bf16x2(o[0], o[1]);
                    pk[1] = lwg_pack_bf16x2(o[2], o[3]);
                    *reinterpret_cast<uintx2*>(yb + opix * a.YC + ch) = pk;
                }
            }
        } else if (EPI == LWG_EPI_SPADE) {
            // wave columns [0,32) = gamma, [32,64) = beta of the same 32 output channels
            const int chb = (n_base + wn * TN * 32) >> 1;           // first of the wave's 32 output channels
            float xf[4][4], o[4][4];
            load_dt(xnb + opix * a.YC + chb, live, xf);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = chb + 8 * g + 4 * khalf;
                const floatx4 mu = *reinterpret_cast<const floatx4*>(a.mean + (size_t)bimg * a.YC + ch);
                const floatx4 rs = *reinterpret_cast<const floatx4*>(a.rstd + (size_t)bimg * a.YC + ch);
                const floatx4 bg4 = *reinterpret_cast<const floatx4*>(a.bias + n_base + wn * 64 + 8 * g + 4 * khalf);
                const floatx4 bb4 = *reinterpret_cast<const floatx4*>(a.bias + n_base + wn * 64 + 32 + 8 * g + 4 * khalf);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float gm = acc[i][0][4 * g + c] + bg4[c];
                    const float bt = acc[i][TN - 1][4 * g + c] + bb4[c];
                    o[g][c] = lwg_act_c<EA>((xf[g][c] - mu[c]) * rs[c] * (1.f + gm) + bt, a.act);
                }
            }
            store_dt(yb + opix * a.YC + chb, live, o);
        } else {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int ncol = n_base + wn * TN * 32 + 32 * j;
                float r4[4][4], o[4][4];
                if (EPI == LWG_EPI_RESIDUAL) load_dt(resb + opix * a.YC + a.ycoff + ncol, live, r4);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    floatx4 b4 = {0.f, 0.f, 0.f, 0.f};
                    if (a.bias) b4 = *reinterpret_cast<const floatx4*>(a.bias + ncol + 8 * g + 4 * khalf);
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        o[g][c] = lwg_act_c<EA>(acc[i][j][4 * g + c] + b4[c] + (EPI == LWG_EPI_RESIDUAL ? r4[g][c] : 0.f), a.act);
                }
                store_dt(yb + opix * a.YC + a.ycoff + ncol, live, o);
            }
        }
    }
}

template <int TM, int TN, int EPI, int SPATIAL = 0>
__device__ __forceinline__ void lwg_bf16_epilogue(const LwgConvArgs& a, floatx16 (&acc)[TM][TN], int m_base, int n_base, int wm, int wn,
                                                  int lane, int tb = 0, int ty0 = 0, int tx0 = 0, const float* pre = nullptr, int pre_bn = 0) {
    if (a.act == LWG_ACT_RELU) lwg_bf16_epilogue_a<TM, TN, EPI, SPATIAL, LWG_ACT_RELU>(a, acc, m_base, n_base, wm, wn, lane, tb, ty0, tx0, pre, pre_bn);
    else if (a.act == LWG_ACT_NONE) lwg_bf16_epilogue_a<TM, TN, EPI, SPATIAL, LWG_ACT_NONE>(a, acc, m_base, n_base, wm, wn, lane, tb, ty0, tx0, pre, pre_bn);
    else lwg_bf16_epilogue_a<TM, TN, EPI, SPATIAL, -1>(a, acc, m_base, n_base, wm, wn, lane, tb, ty0, tx0, pre, pre_bn);
}

template <int WAVES_M, int WAVES_N, int TM, int TN, int EPI, bool DMA_A>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64, WAVES_M * WAVES_N == 4 ? 2 : 1) void lwg_conv_bf16_kernel(const LwgConvArgs a) {
    constexpr int NW = WAVES_M * WAVES_N;                    // waves per workgroup: 4 (128 x 128 / 128 x 64 tiles) or 8 (256 x 256)
    constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
    constexpr int A_STAGE = BM * 128, B_STAGE = BN * 128;    // bytes: [row][64 bf16]
    constexpr int PA = BM / (8 * NW);                        // 8-row DMA pieces per wave per K-step (A)
    constexpr int PB = BN / (8 * NW);
    static_assert(NW == 4 || NW == 8, "4 or 8 waves per workgroup");

    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    char* As = smem_c;
    char* Bs = smem_c + 2 * A_STAGE;
    int* taptab = reinterpret_cast<int*>(smem_c + 2 * A_STAGE + 2 * B_STAGE);  // [3][LWG_MAX_TAPS]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WAVES_N, wn = wid % WAVES_N;
    const int tiles_n = a.N / BN;
    const int lid = lwg_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = lid % tiles_n, tile_m = lid / tiles_n;
    const int m_base = tile_m * BM, n_base = tile_n * BN;

    // ---- gather coordinates of this lane's PA rows (fixed over the K loop) ----
    const int HW = a.OH * a.OW;
    const int Cin = a.C0 + a.C1;
    int pixlin[PA], piy[PA], pix[PA];
    unsigned chunk16[PA];
    unsigned long long vmask[PA];
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        const int r = (wid * PA + p) * 8 + (lane >> 3);      // row inside the tile
        const int m = m_base + r;
        const bool ok = m < a.M;
        const int mm = ok ? m : 0;
        const int b = mm / HW, rem = mm - b * HW;
        const int oy = rem / a.OW, ox = rem - oy * a.OW;
        piy[p] = ok ? oy * a.stride : -100000;
        pix[p] = ox * a.stride;
        pixlin[p] = (b * a.H + oy * a.stride) * a.W + ox * a.stride;
        chunk16[p] = (unsigned)((lane & 7) ^ ((r >> 1) & 7)) * 16u;   // which 16-byte k-octet this lane fetches for LDS slot lane & 7
        vmask[p] = 0ull;
    }
    if (tid < a.ntaps) {
        const int dy = a.dy[tid], dx = a.dx[tid];
        taptab[tid] = (dy * a.W + dx) * a.C0 * 2;
        taptab[LWG_MAX_TAPS + tid] = (dy * a.W + dx) * a.C1 * 2;
        taptab[2 * LWG_MAX_TAPS + tid] = (dy & 0xffff) | (dx << 16);
    }
    __syncthreads();
    for (int tp = 0; tp < a.ntaps; ++tp) {
        const int packed = taptab[2 * LWG_MAX_TAPS + tp];
        const int dy = (int)(short)(packed & 0xffff), dx = packed >> 16;
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const int iy = piy[p] + dy, ix = pix[p] + dx;
            const bool ok = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            vmask[p] |= (unsigned long long)ok << tp;
        }
    }

    const unsigned bytes0 = (unsigned)a.B * a.H * a.W * a.C0 * 2u;
    const unsigned bytes1 = (unsigned)a.B * a.H * a.W * a.C1 * 2u;
    const int nsteps = a.ntaps * (Cin >> 6);
    const unsigned wbytes = (unsigned)nsteps * (unsigned)a.N * 128u;

    // ---- loader state: the K-step whose loads are issued next (channel-chunk major, tap minor) ----
    int ld_tap = 0, ld_cc = 0, ld_use1 = 0;
    unsigned ld_soffA = 0, ld_soffB = 0;
    const void* ld_src = a.x0;
    unsigned ld_bytes = bytes0;
    unsigned pixb[PA], vbase[PA], wvoff[PB];
#pragma unroll
    for (int p = 0; p < PB; ++p) wvoff[p] = (unsigned)(n_base * 128 + (wid * PB + p) * 1024 + lane * 16);
    auto source = [&]() {
        ld_use1 = ld_cc >= a.C0;
        const int cs = ld_use1 ? a.C1 : a.C0;
        ld_src = ld_use1 ? (const void*)a.x1 : (const void*)a.x0;
        ld_bytes = ld_use1 ? bytes1 : bytes0;
        ld_soffA = (unsigned)(ld_cc - (ld_use1 ? a.C0 : 0)) * 2u;
#pragma unroll
        for (int p = 0; p < PA; ++p) pixb[p] = (unsigned)pixlin[p] * (unsigned)cs * 2u + chunk16[p];
    };
    auto tap_rows = [&]() {
        const int toff = taptab[ld_use1 * LWG_MAX_TAPS + ld_tap];
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const bool ok = (vmask[p] >> ld_tap) & 1ull;
            vbase[p] = ok ? pixb[p] + (unsigned)toff : LWG_OOB_OFFSET;
        }
    };
    auto advance = [&]() {
        ld_soffB += (unsigned)a.N * 128u;
        if (++ld_tap == a.ntaps) {
            ld_tap = 0;
            ld_cc += 64;
            ld_soffA += 128u;
            if (ld_cc == a.C0 && a.C1 > 0) source();
        }
        tap_rows();
    };

    uintx4 ra[PA];
    auto issue = [&](int buf) {
        __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ld_src), 0, (int)ld_bytes, 0x00020000);
        __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, (int)wbytes, 0x00020000);
        if (DMA_A) {
#pragma unroll
            for (int p = 0; p < PA; ++p)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, LWG_LDS_PTR(As + buf * A_STAGE + (wid * PA + p) * 1024), 16, (int)vbase[p],
                                                         (int)ld_soffA, 0, 0);
        } else {
#pragma unroll
            for (int p = 0; p < PA; ++p)
                ra[p] = __builtin_bit_cast(uintx4, __builtin_amdgcn_raw_buffer_load_b128(rA, (int)vbase[p], (int)ld_soffA, 0));
        }
#pragma unroll
        for (int p = 0; p < PB; ++p)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, LWG_LDS_PTR(Bs + buf * B_STAGE + (wid * PB + p) * 1024), 16, (int)wvoff[p],
                                                     (int)ld_soffB, 0, 0);
    };
    auto store_a = [&](int buf) {      // register-staged variant only: the same lane-linear destination the DMA would write
#pragma unroll
        for (int p = 0; p < PA; ++p)
            *reinterpret_cast<uintx4*>(As + buf * A_STAGE + (wid * PA + p) * 1024 + lane * 16) = ra[p];
    };

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int khalf = lane >> 5;
    const int sw = (lane >> 1) & 7;
    const char* fr_a = As + (wm * TM * 32 + (lane & 31)) * 128;
    const char* fr_b = Bs + (wn * TN * 32 + (lane & 31)) * 128;
    int koff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = ((2 * ks + khalf) ^ sw) << 4;

    source();
    tap_rows();
    issue(0);
    if (nsteps > 1) advance();
    if (!DMA_A) store_a(0);
    __builtin_amdgcn_s_waitcnt(0x0f70);        // vmcnt(0): this wave's DMA pieces of stage 0 have landed
    __syncthreads();

    for (int t = 0; t < nsteps; ++t) {
        const int cur = t & 1;
        const bool next = t + 1 < nsteps;
        if (next) {
            issue(cur ^ 1);                    // K-step t+1 -> the other stage (all waves left it at the barrier of step t-1)
            if (t + 2 < nsteps) advance();
        }
        bf16x8 fa[2][TM], fb[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[0][i] = *reinterpret_cast<const bf16x8*>(fr_a + cur * A_STAGE + i * 4096 + koff[0]);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[0][j] = *reinterpret_cast<const bf16x8*>(fr_b + cur * B_STAGE + j * 4096 + koff[0]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks < 3) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    fa[(ks + 1) & 1][i] = *reinterpret_cast<const bf16x8*>(fr_a + cur * A_STAGE + i * 4096 + koff[ks + 1 < 4 ? ks + 1 : 3]);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    fb[(ks + 1) & 1][j] = *reinterpret_cast<const bf16x8*>(fr_b + cur * B_STAGE + j * 4096 + koff[ks + 1 < 4 ? ks + 1 : 3]);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks & 1][j], fa[ks & 1][i], acc[i][j], 0, 0, 0);
        }
        if (next && !DMA_A) store_a(cur ^ 1);
        __builtin_amdgcn_s_waitcnt(0x0f70);    // vmcnt(0) before the barrier that publishes the next stage
        __syncthreads();
    }

    lwg_bf16_epilogue<TM, TN, EPI>(a, acc, m_base, n_base, wm, wn, lane);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Halo tile + register-streamed weights + row renaming: the 3x3 and 2x2-tap stride-1 launches (every 3x3 conv of the generator and
// the parity launches of its transposed convs: > 90 % of the flops).
// The linear kernel above puts a barrier and a "wait for ALL my outstanding loads" into every K-step, and each step's operands are
// requested one step ahead: the step time cannot go below one L2 round trip (~1 us under load, measured: res-block launch = 36
// steps x 1.1 us whatever the bytes per step) while its 16 MFMAs need 0.25 us.  It also re-fetches an activation chunk once per tap.
//   * activations: a workgroup's 128 GEMM rows are an 8 x 16 PIXEL BLOCK of one image; its (8+2) x (16+2) halo, 64 channels deep
//     (23 KB, out-of-image pixels = out-of-range offsets = zeros), is staged ONCE per channel chunk THROUGH REGISTERS (global -> VGPR
//     at the top of a chunk, ds_write at its end: nine steps of latency slack), double-buffered - one barrier per chunk, none inside
//     it; all taps read their operand fragments from it at shifted pixel positions.  LDS image: [pixel hp = hy*18 + hx][128 B],
//     k-octet o at slot o ^ ((halo column >> 1) & 7);
//   * weights: never in LDS.  Each wave loads ITS fragments straight from the packed panel (lane-contiguous 1 KB per load) into a
//     register ring D steps ahead; the compiler's counted vmcnt waits for exactly the oldest ring slot.  Panel layout:
//     [step = chunk*ntaps + tap][ks 4][N][16] (the two k-octets of MFMA k-step ks); SPADE panels interleave gamma | beta in blocks
//     of 16 columns so one 32-column tile carries both for 16 channels;
//   * row renaming: with one 1 KB activation fragment read from LDS per MFMA and eight waves per CU the LDS port is exactly as busy
//     as the matrix pipe would be at 100 % (PMC on that first form: matrix pipe 52 %).  Here a wave's row tile i is the image-row
//     PAIR (i, i + TM) of its 2*TM rows instead of (2i, 2i + 1).  A tap's vertical shift then maps row tile i onto row tile i + dy:
//     the fragments E_e = rows (e + dymin, e + dymin + TM), e = 0 .. TM + NDY - 2, are read once per (tap column, k-step) and feed ALL
//     NDY vertical taps by register renaming - TM + NDY - 1 reads for NDY * TM MFMAs (6 for 12 on the 3x3 layers, 5 for 8 on the
//     2x2-tap up-sampling launches), no shuffles.  Their addresses differ by a constant (one halo row), so a (tap column, k-step)
//     costs one address computation.  The weight ring holds 4 * D fragments refilled one fragment at a time; taps must be an
//     ascending NDY x NDX grid (the host sorts them).
// The two earlier forms (halo tile with LDS-DMA weights; one fragment per MFMA) are in the history of this file, DESIGN.md 3.11 has
// their measurements.
// lab instrumentation (compiled out of the product): tools/hr2ts.py on a -DLWG_HR2_TS variant library - wave 0 of every workgroup stamps kernel entry, the
// first barrier passed, K-loop exit and the end of its epilogue into args->res (LWG_EPI_NONE launches; eight 64-bit slots per workgroup)
#ifdef LWG_HR2_TS
#define HTS(i) do { if (threadIdx.x == 0) reinterpret_cast<unsigned long long*>(const_cast<void*>(static_cast<const void*>(a.res)))[(size_t)blockIdx.x * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define HTS(i) do { } while (0)
#endif
#define LWG_HALO_W 18
#define LWG_HALO_PIX 180
#define LWG_HALO_PIECES 23                 // 8 pixels per DMA piece; the 23rd is half used
#define LWG_HALO_BYTES (LWG_HALO_PIECES * 1024)

template <int NDY, int NDX, int EPI, int WAVES_M>
__global__ __launch_bounds__(256, (WAVES_M == 1 || NDY * NDX == 9) ? 2 : 3) void lwg_conv_bf16_hr2_kernel(const LwgConvArgs a) {
    constexpr int NTAPS = NDY * NDX;
    constexpr int WAVES_N = 4 / WAVES_M, TM = 4 / WAVES_M, BN = WAVES_N * 32;
    constexpr int NE = TM + NDY - 1;                         // fragments per (tap column, k-step)
    constexpr int FPC = NDX * 4 * NDY;                       // weight fragments per 64-channel chunk, in consumption order (dx, ks, dy)
    constexpr int R = NTAPS == 9 ? 12 : 8;                   // ring: the same look-ahead as the kernel above (3 / 2 K-steps)
    static_assert(FPC % R == 0, "the weight ring must close at a chunk boundary");
    constexpr int PAH = (LWG_HALO_PIECES + 3) / 4;
    constexpr int ROWB = LWG_HALO_W * 128;                   // bytes between two halo rows

    extern __shared__ __attribute__((aligned(16))) char smem_r2[];
    char* Ah = smem_r2;                                      // [2][LWG_HALO_BYTES]
    HTS(0);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WAVES_N, wn = wid % WAVES_N;
    const int tiles_n = a.N / BN;
    const int tiles_x = (a.OW + 15) >> 4, tiles_y = (a.OH + 7) >> 3;
    const int lid = lwg_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = lid % tiles_n;
    int rest = lid / tiles_n;
    const int tix = rest % tiles_x;
    rest /= tiles_x;
    const int tiy = rest % tiles_y, tb = rest / tiles_y;
    const int x0 = tix * 16, y0 = tiy * 8;
    const int n_base = tile_n * BN;

    const int Cin = a.C0 + a.C1;
    const int nchunks = Cin >> 6;
    const int nfrags = nchunks * FPC;
    const unsigned bytes0 = (unsigned)a.B * a.H * a.W * a.C0 * 2u;
    const unsigned bytes1 = (unsigned)a.B * a.H * a.W * a.C1 * 2u;
    const unsigned wbytes = (unsigned)nchunks * NTAPS * (unsigned)a.N * 128u;

    int hpixlin[PAH];
    unsigned hoct[PAH];
#pragma unroll
    for (int p = 0; p < PAH; ++p) {
        const int hp = (wid * PAH + p) * 8 + (lane >> 3);
        const int hy = hp / LWG_HALO_W, hx = hp - hy * LWG_HALO_W;
        const int gy = y0 + hy - 1, gx = x0 + hx - 1;
        const bool ok = hp < LWG_HALO_PIX && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        hpixlin[p] = ok ? (tb * a.H + gy) * a.W + gx : -1;
        hoct[p] = (unsigned)((lane & 7) ^ ((hx >> 1) & 7)) * 16u;
    }
    uintx4 hreg[PAH];
    auto load_halo = [&](int chunk) {
        const int cc = chunk << 6;
        const bool use1 = cc >= a.C0;
        const void* src = use1 ? (const void*)a.x1 : (const void*)a.x0;
        const unsigned cs = (unsigned)(use1 ? a.C1 : a.C0);
        const unsigned soff = (unsigned)(cc - (use1 ? a.C0 : 0)) * 2u;
        __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(src), 0, (int)(use1 ? bytes1 : bytes0), 0x00020000);
#pragma unroll
        for (int p = 0; p < PAH; ++p) {
            const unsigned voff = hpixlin[p] >= 0 ? (unsigned)hpixlin[p] * cs * 2u + hoct[p] : LWG_OOB_OFFSET;
            hreg[p] = __builtin_bit_cast(uintx4, __builtin_amdgcn_raw_buffer_load_b128(rA, (int)voff, (int)soff, 0));
        }
    };
    auto store_halo = [&](int buf) {
#pragma unroll
        for (int p = 0; p < PAH; ++p) {
            const int q = wid * PAH + p;
            if (q < LWG_HALO_PIECES) *reinterpret_cast<uintx4*>(Ah + buf * LWG_HALO_BYTES + q * 1024 + lane * 16) = hreg[p];
        }
    };

    const int khalf = lane >> 5;
    const unsigned wv = (unsigned)((n_base + wn * 32 + (lane & 31)) * 32 + khalf * 16);
    __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, (int)wbytes, 0x00020000);
    bf16x8 bq[R];
    // weight fragment number ql of a chunk (consumption order: tap column, k-step, tap row) -> its place in the panel
    auto load_w = [&](int chunk, int ql, int slot) {
        const int dxi = ql / (4 * NDY), ks = (ql / NDY) & 3, dyi = ql % NDY;
        const int step = chunk * NTAPS + dyi * NDX + dxi;
        bq[slot] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rW, (int)wv, (int)((unsigned)(step * 4 + ks) * (unsigned)a.N * 32u), 0));
    };

    floatx16 acc[TM][1];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;

    const int dymin = (int)a.dy[0], dxmin = (int)a.dx[0];
    const int hx0 = (lane & 15) + 1;                       // halo column of this lane's pixel for a centred tap
    // halo pixel of fragment E_0 for the first tap column: image row wm*2*TM + dymin + TM * (second half of the 32 lanes)
    const int prow0 = (wm * 2 * TM + dymin + TM * ((lane >> 4) & 1) + 1) * LWG_HALO_W + hx0 + dxmin;

    load_halo(0);
#pragma unroll
    for (int q = 0; q < R; ++q)
        if (q < nfrags) load_w(0, q, q);
    // the epilogue's operands (bias; SPADE: mean / rstd of this image) parked in LDS behind the halo buffers: read from global memory in the epilogue
    // they were an L2 round trip between the last MFMA and the first store of every tile
    float* const pre = reinterpret_cast<float*>(smem_r2 + 2 * LWG_HALO_BYTES);
    if (tid < BN) pre[tid] = a.bias ? a.bias[n_base + tid] : 0.f;
    if (EPI == LWG_EPI_SPADE && tid < BN) {
        const int c = (n_base >> 1) + (tid & (BN / 2 - 1));
        pre[BN + tid] = tid < BN / 2 ? a.mean[(size_t)tb * a.YC + c] : a.rstd[(size_t)tb * a.YC + c];
    }
    store_halo(0);
    __syncthreads();
    HTS(1);

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const char* Acur = Ah + (chunk & 1) * LWG_HALO_BYTES;
        const bool more = chunk + 1 < nchunks;
        const int cnext = more ? chunk + 1 : chunk;
        if (more) load_halo(chunk + 1);                    // lands during this chunk's MFMAs
        bf16x8 E[2][NE];
        auto read_e = [&](int g, int buf) {                // fragments of group g = (tap column, k-step)
            const int dxi = g >> 2, ks = g & 3;
            int px = prow0 + dxi;
            asm volatile("" : "+v"(px));                   // opaque: keeps the per-group addresses out of the chunk-invariant code motion
            const int sw = ((hx0 + dxmin + dxi) >> 1) & 7;
            const char* p = Acur + (px << 7) + (((2 * ks + khalf) ^ sw) << 4);
#pragma unroll
            for (int e = 0; e < NE; ++e) E[buf][e] = *reinterpret_cast<const bf16x8*>(p + e * ROWB);
        };
        read_e(0, 0);
#pragma unroll
        for (int g = 0; g < NDX * 4; ++g) {
            if (g + 1 < NDX * 4) read_e(g + 1, (g + 1) & 1);
            __builtin_amdgcn_sched_group_barrier(0x100, NE, 0);                    // the next group's DS reads first ...
#pragma unroll
            for (int dyi = 0; dyi < NDY; ++dyi) {
                const int ql = g * NDY + dyi;
                const int slot = ql % R;
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[slot], E[g & 1][i + dyi], acc[i][0], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, TM, 0);                // ... then the MFMAs of this tap row
                const int nq = ql + R;                                             // refill the slot just consumed
                if (nq < FPC) load_w(chunk, nq, slot);
                else load_w(cnext, nq - FPC, slot);        // (last chunk: a harmless re-load of its own first fragments - unconditional, so that no
                                                           //  uniform branch sits between the MFMAs: twelve per chunk in the 3 x 3 form until round 6)
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) {
            store_halo((chunk + 1) & 1);                   // the other buffer: every wave left it at the previous chunk's barrier
            __syncthreads();
        }
    }
    HTS(2);
    lwg_bf16_epilogue<TM, 1, EPI, 2>(a, acc, 0, n_base, wm, wn, lane, tb, y0, x0, pre, BN);
    HTS(3);
}

template <int NDY, int NDX, int EPI, int WAVES_M>
static hipError_t launch_cfg_bf16_hr2(const LwgConvArgs& a, hipStream_t stream) {
    constexpr size_t lds = (size_t)2 * LWG_HALO_BYTES + 2 * 128 * sizeof(float);   // two halo buffers + the parked epilogue operands (2 BN floats)
    auto kern = lwg_conv_bf16_hr2_kernel<NDY, NDX, EPI, WAVES_M>;
    const long tiles = (long)a.B * ((a.OH + 7) / 8) * ((a.OW + 15) / 16) * (a.N / (128 / WAVES_M));
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), lds, stream, a);
    return hipGetLastError();
}

// taps are an ascending row-major NDY x NDX grid of consecutive offsets?
static bool lwg_bf16_tap_grid(const LwgConvArgs& a, int ndy, int ndx) {
    if (a.ntaps != ndy * ndx) return false;
    for (int t = 0; t < a.ntaps; ++t)
        if (a.dy[t] != a.dy[0] + t / ndx || a.dx[t] != a.dx[0] + t % ndx) return false;
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------------
// ConvTranspose2d(4, 2, 1) as ONE launch: the four output parities (2x2 taps each) of a transposed convolution read the same
// input block.  As four launches the last up-sampling layer (128 -> 64 channels at full resolution) reads its input four times and is
// HBM-governed (259 us per parity launch at 1024^2 x 8 frames against a 160 us byte floor).  Here a workgroup stages the halo of its
// 8 x 16 input pixels ONCE - every 64-channel chunk resident (Cin <= 128: 46 KB) - and walks the parities: per parity the row-renaming
// K loop above with NDY = NDX = 2 (dy in {py - 1, py}, dx in {px - 1, px}), then the epilogue into the (2y + py, 2x + px) pixels.  No
// barrier after the staging; the weight ring runs through the four panels without a bubble.
template <int NCH, int WAVES_M>
__global__ __launch_bounds__(256, WAVES_M == 1 ? 2 : 3) void lwg_conv_bf16_up4_kernel(const LwgConvArgs a) {
    constexpr int WAVES_N = 4 / WAVES_M, TM = 4 / WAVES_M, BN = WAVES_N * 32;
    constexpr int NDY = 2, NDX = 2, NTAPS = 4, NE = TM + NDY - 1, FPC = NDX * 4 * NDY, R = 8, FPP = NCH * FPC;   // FPP: fragments per parity
    constexpr int PAH = (LWG_HALO_PIECES + 3) / 4;
    constexpr int ROWB = LWG_HALO_W * 128;

    extern __shared__ __attribute__((aligned(16))) char smem_u[];
    char* Ah = smem_u;                                       // [NCH][LWG_HALO_BYTES]

#ifdef LAB_TS       // tools/up4ts.py: wave 0 stamps s_memtime at the phase boundaries of its tile and leaves the stamps in args->res (12 per tile)
    unsigned long long ts[12];
    int nts = 0;
#define LAB_STAMP() do { ts[nts++] = __builtin_amdgcn_s_memtime(); } while (0)
    LAB_STAMP();
#else
#define LAB_STAMP() do { } while (0)
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WAVES_N, wn = wid % WAVES_N;
    const int tiles_n = a.N / BN;
    const int tiles_x = (a.OW + 15) >> 4, tiles_y = (a.OH + 7) >> 3;
    const int lid = lwg_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = lid % tiles_n;
    int rest = lid / tiles_n;
    const int tix = rest % tiles_x;
    rest /= tiles_x;
    const int tiy = rest % tiles_y, tb = rest / tiles_y;
    const int x0 = tix * 16, y0 = tiy * 8;
    const int n_base = tile_n * BN;

    const unsigned bytes0 = (unsigned)a.B * a.H * a.W * a.C0 * 2u;
    const unsigned ppanel = (unsigned)NCH * NTAPS * (unsigned)a.N * 128u;           // bytes of one parity's panel
    {
        __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x0), 0, (int)bytes0, 0x00020000);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            uintx4 hreg[PAH];
#pragma unroll
            for (int p = 0; p < PAH; ++p) {
                const int hp = (wid * PAH + p) * 8 + (lane >> 3);
                const int hy = hp / LWG_HALO_W, hx = hp - hy * LWG_HALO_W;
                const int gy = y0 + hy - 1, gx = x0 + hx - 1;
                const bool ok = hp < LWG_HALO_PIX && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                const unsigned voff = ok ? (unsigned)((tb * a.H + gy) * a.W + gx) * (unsigned)a.C0 * 2u + (unsigned)((lane & 7) ^ ((hx >> 1) & 7)) * 16u
                                         : LWG_OOB_OFFSET;
                hreg[p] = __builtin_bit_cast(uintx4, __builtin_amdgcn_raw_buffer_load_b128(rA, (int)voff, c * 128, 0));
            }
#pragma unroll
            for (int p = 0; p < PAH; ++p) {
                const int q = wid * PAH + p;
                if (q < LWG_HALO_PIECES) *reinterpret_cast<uintx4*>(Ah + c * LWG_HALO_BYTES + q * 1024 + lane * 16) = hreg[p];
            }
        }
    }

    const int khalf = lane >> 5;
    const unsigned wv = (unsigned)((n_base + wn * 32 + (lane & 31)) * 32 + khalf * 16);
    __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, (int)(4u * ppanel), 0x00020000);
    bf16x8 bq[R];
    auto load_w = [&](int parity, int qp, int slot) {      // fragment qp of a parity: (chunk, tap column, k-step, tap row)
        const int chunk = qp / FPC, ql = qp % FPC;
        const int dxi = ql / (4 * NDY), ks = (ql / NDY) & 3, dyi = ql % NDY;
        const int step = chunk * NTAPS + dyi * NDX + dxi;
        bq[slot] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
            rW, (int)wv, (int)((unsigned)parity * ppanel + (unsigned)(step * 4 + ks) * (unsigned)a.N * 32u), 0));
    };
#pragma unroll
    for (int q = 0; q < R; ++q) load_w(0, q, q);
    const int hx0 = (lane & 15) + 1;
    __syncthreads();
    LAB_STAMP();             // halo in LDS

#pragma unroll 1
    for (int parity = 0; parity < 4; ++parity) {
        const int py = parity >> 1, px = parity & 1;
        floatx16 acc[TM][1];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
        const int prow0 = (wm * 2 * TM + (py - 1) + TM * ((lane >> 4) & 1) + 1) * LWG_HALO_W + hx0 + (px - 1);
#pragma unroll
        for (int chunk = 0; chunk < NCH; ++chunk) {
            const char* Acur = Ah + chunk * LWG_HALO_BYTES;
            bf16x8 E[2][NE];
            auto read_e = [&](int g, int buf) {
                const int dxi = g >> 2, ks = g & 3;
                int pxl = prow0 + dxi;
                asm volatile("" : "+v"(pxl));
                const int sw = ((hx0 + (px - 1) + dxi) >> 1) & 7;
                const char* p = Acur + (pxl << 7) + (((2 * ks + khalf) ^ sw) << 4);
#pragma unroll
                for (int e = 0; e < NE; ++e) E[buf][e] = *reinterpret_cast<const bf16x8*>(p + e * ROWB);
            };
            read_e(0, 0);
#pragma unroll
            for (int g = 0; g < NDX * 4; ++g) {
                if (g + 1 < NDX * 4) read_e(g + 1, (g + 1) & 1);
                __builtin_amdgcn_sched_group_barrier(0x100, NE, 0);
#pragma unroll
                for (int dyi = 0; dyi < NDY; ++dyi) {
                    const int qp = chunk * FPC + g * NDY + dyi;
                    const int slot = qp % R;
#pragma unroll
                    for (int i = 0; i < TM; ++i) acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[slot], E[g & 1][i + dyi], acc[i][0], 0, 0, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, TM, 0);
                    const int nq = qp + R;                 // refill: this parity's later fragments, then the next parity's first ones
                    if (nq < FPP) load_w(parity, nq, slot);
                    else if (parity < 3) load_w(parity + 1, nq - FPP, slot);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        LAB_STAMP();         // this parity's K loop issued
        LwgConvArgs ap = a;
        ap.ooy = py;
        ap.oox = px;
        lwg_bf16_epilogue<TM, 1, LWG_EPI_NONE, 2>(ap, acc, 0, n_base, wm, wn, lane, tb, y0, x0);
        LAB_STAMP();         // its stores issued
    }
#ifdef LAB_TS
    __builtin_amdgcn_s_waitcnt(0);
    LAB_STAMP();             // everything of this wave acknowledged
    if (a.res && lane == 0 && wid == 0) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(const_cast<float*>(a.res)) + (size_t)lid * 12;
        for (int q = 0; q < 12; ++q) o[q] = q < nts ? ts[q] : 0ull;
    }
#endif
}

template <int NCH, int WAVES_M>
static hipError_t launch_cfg_bf16_up4(const LwgConvArgs& a, hipStream_t stream) {
    constexpr size_t lds = (size_t)NCH * LWG_HALO_BYTES;
    auto kern = lwg_conv_bf16_up4_kernel<NCH, WAVES_M>;
    const long tiles = (long)a.B * ((a.OH + 7) / 8) * ((a.OW + 15) / 16) * (a.N / (128 / WAVES_M));
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), lds, stream, a);
    return hipGetLastError();
}

// nn.ConvTranspose2d(kernel 4, stride 2, padding 1) on bf16 NHWC in one launch.  args: the launch description of the parity-(0,0)
// launch (ntaps = 4, stride = 1, omul = 2, OH = H, OW = W, YH = 2H, YW = 2W); args->dy / dx / ooy / oox are ignored (parity (py, px)
// uses dy in {py-1, py}, dx in {px-1, px} and writes pixels (2y + py, 2x + px)); args->w = the four register-streamed panels
// [parity = 2 py + px][Cin/64 * 4][4][N][16], each with its taps ascending in (dy, dx).  Cin = 64 or 128, no second input,
// N % 64 == 0, LWG_EPI_NONE (bias + activation).
extern "C" int lwg_conv_transpose4_nhwc_bf16(const LwgConvArgs* pa, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!pa) return (int)hipErrorInvalidValue;
    const LwgConvArgs& a = *pa;
    {   // inputs beyond the 32-bit buffer-offset range: the same launch in batch slices (lwg_conv_slices.h)
        int sliced_err = 0;
        if (lwg_conv_run_sliced(a, [&](const LwgConvArgs& s) { return lwg_conv_transpose4_nhwc_bf16(&s, stream_); }, &sliced_err)) return sliced_err;
    }
    if (!a.x0 || !a.w || !a.y || a.M <= 0 || a.ntaps != 4 || a.C1 != 0 || (a.C0 != 64 && a.C0 != 128)) return (int)hipErrorInvalidValue;
    if (a.xdt != LWG_DT_BF16 || a.ydt != LWG_DT_BF16 || a.stride != 1 || a.H != a.OH || a.W != a.OW || a.omul != 2) return (int)hipErrorInvalidValue;
    if (a.YH != 2 * a.OH || a.YW != 2 * a.OW || a.N % 64 != 0 || (a.YC & 7) != 0 || (a.ycoff & 7) != 0 || a.epi != LWG_EPI_NONE) return (int)hipErrorInvalidValue;
    if ((unsigned long long)a.B * a.H * a.W * (unsigned long long)a.C0 * 2ull >= 0xC0000000ull) return (int)hipErrorInvalidValue;
    if (a.N % 128 == 0) return (int)(a.C0 == 64 ? launch_cfg_bf16_up4<1, 1>(a, stream) : launch_cfg_bf16_up4<2, 1>(a, stream));
    return (int)(a.C0 == 64 ? launch_cfg_bf16_up4<1, 2>(a, stream) : launch_cfg_bf16_up4<2, 2>(a, stream));
}

// ---------------------------------------------------------------------------------------------------------------------------
// Pointwise (1 x 1, stride 1, C -> C) kernel: the query projections fq of the attention blocks.  K is one to four 64-channel
// chunks - the tiled kernels above spend such a launch in prologue / epilogue and one memory round trip per K-step (0.18 of the
// HBM rate).  Here the whole weight matrix lives in registers (a wave owns 32 columns: K/16 fragments = 16..64 VGPRs), a
// persistent 8-wave workgroup per CU walks the row tiles, and a tile's activations (rows x K, one contiguous block of the NHWC
// tensor) come in by LDS-DMA, double-buffered by TILE: the DMA of tile t+1 is in flight for the whole of tile t, one barrier
// per tile.  HBM-bound: reads M x K, writes M x N bf16 once.  Two workgroups per CU (TM = 2: 64-row wave tiles, 2 x 32 KB stages each)
// stream better than one with 128-row tiles: while one waits for its DMA the other computes and stores (C = 256: 51 -> 39 us).
template <int NCH, int WAVES_N, int TM>        // TM = 4: 128 rows per wave, one workgroup per CU; TM = 2: 64 rows, two workgroups per CU
__global__ __launch_bounds__(512, TM == 4 ? 1 : 2) void lwg_conv_bf16_pw_kernel(const LwgConvArgs a) {
    constexpr int WAVES_M = 8 / WAVES_N, BM = WAVES_M * TM * 32;
    constexpr int CH_BYTES = BM * 128;                       // one 64-channel chunk of a stage: [row][128 B]
    constexpr int STAGE = NCH * CH_BYTES;
    constexpr int PIECES = STAGE / 1024 / 8;                 // 1 KB DMA pieces per wave per stage
    constexpr int RB = BM / 8;                               // 8-row pieces per chunk
    constexpr unsigned K2 = NCH * 128u;                      // bytes per pixel

    extern __shared__ __attribute__((aligned(16))) char smem_p[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WAVES_N, wn = wid % WAVES_N;
    const int khalf = lane >> 5;
    const int ntiles = (a.M + BM - 1) / BM;

    __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x0), 0, (int)((unsigned)a.M * K2), 0x00020000);
    __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, (int)(NCH * 4u * (unsigned)a.N * 32u), 0x00020000);

    // byte offset (inside a tile's block of the tensor) this lane fetches for piece p: the LDS image is [chunk][row][8 slots of
    // 16 B], slot s of row r holds k-octet s ^ ((r >> 1) & 7); rows past M are out of the buffer's range -> the DMA writes zeros
    unsigned pv[PIECES];
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
        const int q = wid * PIECES + p;
        const int chunk = q / RB, row = (q % RB) * 8 + (lane >> 3);
        pv[p] = (unsigned)row * K2 + (unsigned)chunk * 128u + (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) << 4);
    }
    auto issue = [&](int tile, int buf) {
        const unsigned tb = (unsigned)tile * (unsigned)BM * K2;
#pragma unroll
        for (int p = 0; p < PIECES; ++p)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, LWG_LDS_PTR(smem_p + buf * STAGE + (wid * PIECES + p) * 1024), 16, (int)(tb + pv[p]), 0, 0, 0);
    };

    int tile = blockIdx.x;
    if (tile < ntiles) issue(tile, 0);

    // the wave's 32 columns of the whole weight matrix: [chunk][ks][N][16] (the register-streamed panel with one tap)
    bf16x8 bq[NCH][4];
    const unsigned wv = (unsigned)((wn * 32 + (lane & 31)) * 32 + khalf * 16);
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            bq[c][ks] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rW, (int)wv, (int)((unsigned)(c * 4 + ks) * (unsigned)a.N * 32u), 0));

    const int sw = (lane >> 1) & 7;
    int koff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = ((2 * ks + khalf) ^ sw) << 4;
    const int frow = (wm * TM * 32 + (lane & 31)) * 128;

    for (int buf = 0; tile < ntiles; tile += gridDim.x, buf ^= 1) {
        __builtin_amdgcn_s_waitcnt(0x0f70);        // vmcnt(0): this wave's pieces of the current stage (and its last stores) are done
        __syncthreads();                           // ... everyone's; and every wave has finished reading the other stage
        const int nxt = tile + gridDim.x;
        if (nxt < ntiles) issue(nxt, buf ^ 1);
        floatx16 acc[TM][1];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
        const char* S = smem_p + buf * STAGE + frow;
        bf16x8 fa[2][TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[0][i] = *reinterpret_cast<const bf16x8*>(S + i * 4096 + koff[0]);
#pragma unroll
        for (int s = 0; s < NCH * 4; ++s) {
            if (s + 1 < NCH * 4) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    fa[(s + 1) & 1][i] = *reinterpret_cast<const bf16x8*>(S + ((s + 1) >> 2) * CH_BYTES + i * 4096 + koff[(s + 1) & 3]);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[s >> 2][s & 3], fa[s & 1][i], acc[i][0], 0, 0, 0);
        }
        lwg_bf16_epilogue<TM, 1, LWG_EPI_NONE>(a, acc, tile * BM, 0, wm, wn, lane);
    }
}

template <int NCH, int WAVES_N, int TM>
static hipError_t launch_cfg_bf16_pw_tm(const LwgConvArgs& a, hipStream_t stream) {
    constexpr int BM = (8 / WAVES_N) * TM * 32;
    constexpr size_t lds = (size_t)2 * NCH * BM * 128;
    auto kern = lwg_conv_bf16_pw_kernel<NCH, WAVES_N, TM>;
    static unsigned long long attr_done = 0ull;
    if (hipError_t e = lwg_allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds, attr_done); e != hipSuccess) return e;
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    }
    const int ntiles = (a.M + BM - 1) / BM;
    const int slots = cus * (TM == 4 ? 1 : 2);
    hipLaunchKernelGGL(kern, dim3((unsigned)(ntiles < slots ? ntiles : slots)), dim3(512), lds, stream, a);
    return hipGetLastError();
}

template <int NCH, int WAVES_N>
static hipError_t launch_cfg_bf16_pw(const LwgConvArgs& a, hipStream_t stream) {
    // LWG_BF16_PW_TM (compile-time, tools/labbuild.sh): 4 = one 8-wave workgroup per CU with 128-row wave tiles, 2 = two with 64-row
    // tiles - measured in the frame loop: C = 256 51 -> 39 us, C = 128 79 -> 65, C = 64 153 -> 123
    return launch_cfg_bf16_pw_tm<NCH, WAVES_N, LWG_BF16_PW_TM>(a, stream);
}

static bool lwg_bf16_pw_ok(const LwgConvArgs& a) {
    return a.ntaps == 1 && a.dy[0] == 0 && a.dx[0] == 0 && a.stride == 1 && a.C1 == 0 && a.N == a.C0 && a.epi == LWG_EPI_NONE &&
           (a.N == 64 || a.N == 128 || a.N == 256) && a.H == a.OH && a.W == a.OW;
}

// ---------------------------------------------------------------------------------------------------------------------------
// First layer of a stream in bf16 mode: fp32 NHWC-8 input (the 6-channel network input, zero-extended), <= 10 taps, any stride,
// 64 output columns, bf16 output.  8 channels x 4 B = the two 16-byte loads of one lane; converted to bf16 they are exactly one
// k-octet of the MFMA operand, so a k-step of 16 is two taps (tap = 2 ks + lane / 32) and the A operand needs no LDS at all.
// The 80 x 64 weight panel sits in 40 VGPRs; a wave produces 64 pixels x 64 channels per pass.  HBM-bound (reads the input
// once - the shifted re-reads hit L1 / L2 - and writes M x 64 bf16).  Panel: [ceil(ntaps / 2)][64][16] bf16, k = tap * 8 + c.
__global__ __launch_bounds__(256, 2) void lwg_conv_c8_bf16_kernel(const LwgConvArgs a, const int nks) {
    constexpr int TM = 2, TN = 2, MAXKS = 5;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int khalf = lane >> 5;
    const __bf16* wp = reinterpret_cast<const __bf16*>(a.w);
    bf16x8 bq[MAXKS][TN];
    int tdy[MAXKS], tdx[MAXKS];
#pragma unroll
    for (int ks = 0; ks < MAXKS; ++ks) {
        const int tap = 2 * ks + khalf;
        const bool ok = ks < nks && tap < a.ntaps;
        tdy[ks] = ok ? (int)a.dy[tap] : -100000;               // dead taps read out of range: zeros (their weights are zero too)
        tdx[ks] = ok ? (int)a.dx[tap] : 0;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            bf16x8 z;
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] = (__bf16)0.f;
            bq[ks][j] = ks < nks ? *reinterpret_cast<const bf16x8*>(wp + ((size_t)(ks * 64 + j * 32 + (lane & 31)) * 16 + khalf * 8)) : z;
        }
    }
    __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x0), 0, (int)((unsigned)a.B * a.H * a.W * 32u), 0x00020000);
    const int HW = a.OH * a.OW;
    const int ngroups = (a.M + 63) >> 6;
    for (int g = blockIdx.x * 4 + wid; g < ngroups; g += gridDim.x * 4) {
        floatx16 acc[TM][TN];
        floatx4 raw[TM][MAXKS][2];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = g * 64 + i * 32 + (lane & 31);
            const bool live = m < a.M;
            const int mm = live ? m : 0;
            const int b = mm / HW, rem = mm - b * HW;
            const int oy = rem / a.OW, ox = rem - oy * a.OW;
            const int iy0 = live ? oy * a.stride : -100000, ix0 = ox * a.stride;
#pragma unroll
            for (int ks = 0; ks < MAXKS; ++ks) {
                const int iy = iy0 + tdy[ks], ix = ix0 + tdx[ks];
                const bool ok = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                const unsigned voff = ok ? (unsigned)((b * a.H + iy) * a.W + ix) * 32u : LWG_OOB_OFFSET;
                raw[i][ks][0] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rA, (int)voff, 0, 0));
                raw[i][ks][1] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rA, (int)voff, 16, 0));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
#pragma unroll
        for (int ks = 0; ks < MAXKS; ++ks)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                bf16x8 fa;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    fa[e] = (__bf16)raw[i][ks][0][e];
                    fa[4 + e] = (__bf16)raw[i][ks][1][e];
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[ks][j], fa, acc[i][j], 0, 0, 0);
            }
        lwg_bf16_epilogue<TM, TN, LWG_EPI_NONE>(a, acc, g * 64, 0, 0, 0, lane);
    }
}

// args->x0 fp32 (B,H,W,8), args->w = bf16 panel [ceil(ntaps/2)][64][16] (k = tap*8 + c, zero rows past ntaps*8), N = 64,
// bf16 output; bias + activation epilogue.
extern "C" int lwg_conv2d_nhwc_c8_bf16(const LwgConvArgs* pa, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!pa) return (int)hipErrorInvalidValue;
    const LwgConvArgs& a = *pa;
    {   // inputs beyond the 32-bit buffer-offset range: the same launch in batch slices (lwg_conv_slices.h)
        int sliced_err = 0;
        if (lwg_conv_run_sliced(a, [&](const LwgConvArgs& s) { return lwg_conv2d_nhwc_c8_bf16(&s, stream_); }, &sliced_err)) return sliced_err;
    }
    if (!a.x0 || !a.w || !a.y || a.M <= 0 || a.ntaps < 1 || a.ntaps > 10) return (int)hipErrorInvalidValue;
    if (a.xdt != LWG_DT_F32 || a.ydt != LWG_DT_BF16 || a.C0 != 8 || a.C1 != 0 || a.N != 64 || a.epi != LWG_EPI_NONE) return (int)hipErrorInvalidValue;
    if ((a.YC & 7) != 0 || (a.ycoff & 7) != 0 || a.stride < 1) return (int)hipErrorInvalidValue;
    if ((unsigned long long)a.B * a.H * a.W * 32ull >= 0xC0000000ull) return (int)hipErrorInvalidValue;
    const int ngroups4 = (a.M + 255) / 256;
    hipLaunchKernelGGL(lwg_conv_c8_bf16_kernel, dim3((unsigned)(ngroups4 < 2048 ? ngroups4 : 2048)), dim3(256), 0, stream, a, (a.ntaps + 1) / 2);
    return hipGetLastError();
}

template <int EPI>
static hipError_t launch_epi_bf16_hr(const LwgConvArgs& a, hipStream_t stream) {
    if (lwg_bf16_tap_grid(a, 3, 3)) return a.N % 128 == 0 ? launch_cfg_bf16_hr2<3, 3, EPI, 1>(a, stream) : launch_cfg_bf16_hr2<3, 3, EPI, 2>(a, stream);
    if (lwg_bf16_tap_grid(a, 2, 2)) return a.N % 128 == 0 ? launch_cfg_bf16_hr2<2, 2, EPI, 1>(a, stream) : launch_cfg_bf16_hr2<2, 2, EPI, 2>(a, stream);
    return hipErrorInvalidValue;             // taps must form an ascending 3 x 3 or 2 x 2 grid (packing sorts them)
}

// args->w = the register-streamed panel [ntaps*Cin/64][4][N][16] (bias / SPADE columns in the matching order, see above);
// 3x3 (9 taps) or 2x2 (4 taps) within [-1,1]^2, stride 1, same input / output grid, N % 64 == 0, Cin % 64 == 0.
extern "C" int lwg_conv2d_nhwc_bf16_hr(const LwgConvArgs* pa, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!pa) return (int)hipErrorInvalidValue;
    const LwgConvArgs& a = *pa;
    {   // inputs beyond the 32-bit buffer-offset range: the same launch in batch slices (lwg_conv_slices.h)
        int sliced_err = 0;
        if (lwg_conv_run_sliced(a, [&](const LwgConvArgs& s) { return lwg_conv2d_nhwc_bf16_hr(&s, stream_); }, &sliced_err)) return sliced_err;
    }
    const int Cin = a.C0 + a.C1;
    if (!a.x0 || !a.w || !a.y || a.M <= 0 || (a.ntaps != 9 && a.ntaps != 4 && a.ntaps != 1)) return (int)hipErrorInvalidValue;
    if (a.xdt != LWG_DT_BF16 || a.ydt != LWG_DT_BF16 || a.stride != 1 || a.H != a.OH || a.W != a.OW) return (int)hipErrorInvalidValue;
    if (a.ntaps == 1) {
        // pointwise C -> C launches: weights resident in registers, persistent workgroups (lwg_conv_bf16_pw_kernel)
        if (!lwg_bf16_pw_ok(a) || (a.YC & 7) != 0 || (a.ycoff & 7) != 0) return (int)hipErrorInvalidValue;
        if ((unsigned long long)a.M * (unsigned long long)a.C0 * 2ull >= 0xC0000000ull) return (int)hipErrorInvalidValue;
        if (a.N == 64) return (int)launch_cfg_bf16_pw<1, 2>(a, stream);
        if (a.N == 128) return (int)launch_cfg_bf16_pw<2, 4>(a, stream);
        return (int)launch_cfg_bf16_pw<4, 8>(a, stream);
    }
    if (a.N % 64 != 0 || Cin % 64 != 0 || (a.YC & 7) != 0 || (a.ycoff & 7) != 0) return (int)hipErrorInvalidValue;
    if (a.C1 != 0 && (a.C0 % 64 != 0 || !a.x1)) return (int)hipErrorInvalidValue;
    for (int t = 0; t < a.ntaps; ++t)
        if (a.dy[t] < -1 || a.dy[t] > 1 || a.dx[t] < -1 || a.dx[t] > 1) return (int)hipErrorInvalidValue;
    const unsigned long long pix = (unsigned long long)a.B * a.H * a.W;
    if (pix * (unsigned long long)(a.C0 > a.C1 ? a.C0 : a.C1) * 2ull >= 0xC0000000ull) return (int)hipErrorInvalidValue;
    if ((unsigned long long)a.ntaps * (Cin / 64) * (unsigned long long)a.N * 128ull >= 0xC0000000ull) return (int)hipErrorInvalidValue;
    if (a.epi == LWG_EPI_SPADE) {
        if (!a.xn || !a.mean || !a.rstd || !a.bias || a.YC * 2 != a.N || a.ycoff != 0) return (int)hipErrorInvalidValue;
        return (int)launch_epi_bf16_hr<LWG_EPI_SPADE>(a, stream);
    }
    if (a.epi == LWG_EPI_RESIDUAL) {
        if (!a.res) return (int)hipErrorInvalidValue;
        return (int)launch_epi_bf16_hr<LWG_EPI_RESIDUAL>(a, stream);
    }
    if (a.epi != LWG_EPI_NONE) return (int)hipErrorInvalidValue;
    return (int)launch_epi_bf16_hr<LWG_EPI_NONE>(a, stream);
}

template <int WAVES_M, int WAVES_N, int TM, int TN, int EPI, bool DMA_A>
static hipError_t launch_cfg_bf16(const LwgConvArgs& a, hipStream_t stream) {
    constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
    constexpr size_t lds = (size_t)2 * (BM + BN) * 128 + 3 * LWG_MAX_TAPS * sizeof(int);
    auto kern = lwg_conv_bf16_kernel<WAVES_M, WAVES_N, TM, TN, EPI, DMA_A>;
    static unsigned long long attr_done = 0ull;
    if (hipError_t e = lwg_allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds, attr_done); e != hipSuccess) return e;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = a.N / BN;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(WAVES_M * WAVES_N * 64), lds, stream, a);
    return hipGetLastError();
}

template <int EPI>
static hipError_t launch_epi_bf16(const LwgConvArgs& a, hipStream_t stream) {
    // compile-time tuning constants (tools/labbuild.sh -D...): LWG_BF16_DMA_A 1 = the A operand by LDS-DMA (0: global -> VGPR ->
    // ds_write); LWG_BF16_TILE64 0 = heuristic below, 1 = 128 x 64 tiles wherever the epilogue allows, -1 = never; LWG_BF16_BIG 1 =
    // the 8-wave 256 x 256 tile where it applies
    constexpr bool dma_a = LWG_BF16_DMA_A != 0;
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    }
    // N % 256 == 0 and at least one 256 x 256 tile per CU: eight waves, wave tile 128 x 64 - six fragment reads per eight MFMAs
    // instead of four per four (the 4-wave kernel's LDS port is as busy as its matrix pipe)
    if (LWG_BF16_BIG && dma_a && a.N % 256 == 0 && (long)((a.M + 255) / 256) * (a.N / 256) >= (long)cus) return launch_cfg_bf16<2, 4, 4, 2, EPI, true>(a, stream);
    // a launch with fewer 128 x 128 tiles than two per CU leaves every CU with ONE resident workgroup - nothing to run while it
    // waits for its DMA / barrier.  Halving the tile (128 x 64: 48 KB of LDS, three per CU) doubles the workgroups; measured on the
    // 64^2 x 256 -> 128 SPADE convs (256 tiles): 45 -> 3x us.  Larger launches keep 128 x 128 (more flops per staged byte).
    const long tiles128 = (long)((a.M + 127) / 128) * (a.N / 128);
    const bool small = a.N % 128 == 0 && tiles128 < 2L * cus;
    if (EPI == LWG_EPI_SPADE || (a.N % 128 == 0 && LWG_BF16_TILE64 != 1 && !(small && LWG_BF16_TILE64 == 0)))
        return launch_cfg_bf16<2, 2, 2, 2, EPI, dma_a>(a, stream);
    return launch_cfg_bf16<4, 1, 1, 2, EPI, dma_a>(a, stream);
}

extern "C" int lwg_conv2d_nhwc_bf16(const LwgConvArgs* pa, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!pa) return (int)hipErrorInvalidValue;
    const LwgConvArgs& a = *pa;
    {   // inputs beyond the 32-bit buffer-offset range: the same launch in batch slices (lwg_conv_slices.h)
        int sliced_err = 0;
        if (lwg_conv_run_sliced(a, [&](const LwgConvArgs& s) { return lwg_conv2d_nhwc_bf16(&s, stream_); }, &sliced_err)) return sliced_err;
    }
    const int Cin = a.C0 + a.C1;
    if (!a.x0 || !a.w || !a.y || a.ntaps < 1 || a.ntaps > LWG_MAX_TAPS || a.M <= 0) return (int)hipErrorInvalidValue;
    if (a.xdt != LWG_DT_BF16 || a.ydt != LWG_DT_BF16) return (int)hipErrorInvalidValue;
    if (a.N % 64 != 0 || Cin % 64 != 0 || (a.YC & 7) != 0 || (a.ycoff & 7) != 0) return (int)hipErrorInvalidValue;
    if (a.C1 != 0 && (a.C0 % 64 != 0 || !a.x1)) return (int)hipErrorInvalidValue;
    const unsigned long long pix = (unsigned long long)a.B * a.H * a.W;
    if (pix * (unsigned long long)(a.C0 > a.C1 ? a.C0 : a.C1) * 2ull >= 0xC0000000ull) return (int)hipErrorInvalidValue;
    if ((unsigned long long)a.ntaps * (Cin / 64) * (unsigned long long)a.N * 128ull >= 0xC0000000ull) return (int)hipErrorInvalidValue;
    if (a.epi == LWG_EPI_SPADE) {
        if (!a.xn || !a.mean || !a.rstd || !a.bias || a.N % 128 != 0 || a.YC * 2 != a.N || a.ycoff != 0) return (int)hipErrorInvalidValue;
        return (int)launch_epi_bf16<LWG_EPI_SPADE>(a, stream);
    }
    if (a.epi == LWG_EPI_RESIDUAL) {
        if (!a.res) return (int)hipErrorInvalidValue;
        return (int)launch_epi_bf16<LWG_EPI_RESIDUAL>(a, stream);
    }
    if (a.epi != LWG_EPI_NONE) return (int)hipErrorInvalidValue;
    return (int)launch_epi_bf16<LWG_EPI_NONE>(a, stream);
}
