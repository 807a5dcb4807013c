// Output head of TSFNet fused with the final compositing, plus NCHW <-> NHWC layout changes for the API edge.
// Replaces reference generators/attlwb_spade_resunet.py:605-613 (tsf_img_reg: conv5x5 64->3 + tanh,
// tsf_att_reg: conv5x5 64->1 + sigmoid; both bias-free) as called at :533, and models/imitator.py:393
// (pred = mask * bg + (1 - mask) * img).
//
// Cout = 4 is too narrow for the 32-wide MFMA tiles (12.5 % utilisation), and the fp32 MFMA rate equals the
// fp32 VALU rate on gfx950, so this is a register-blocked VALU direct convolution: a workgroup owns a 32x32
// pixel tile, each thread 4 pixels x 4 outputs; the input halo tile is staged through LDS 8 channels at a
// time as channel-quads ([cq][row][col] float4: a wave reads 512 contiguous bytes per ds_read_b128), weights
// are LDS broadcasts.  Outputs are written NCHW (what Imitator.inference hands to the host), coalesced in x.
#include "lwg_common.h"
#include "lwg_conv_args.h"

#define HT 32          // tile edge
#define HH (HT + 4)    // halo edge (5x5, pad 2)
#define HCH 8          // channels per stage

// J: pixels per thread = tile height / 8.  J = 4: 32 x 32 tiles (frame batches); J = 2: 32 x 16 tiles for launches that would otherwise
// leave CUs without a workgroup (one frame at 512x512 is 256 tiles of 32 x 32: one 4-wave workgroup per CU - 150 us per image against
// 54 us per image inside an 8-frame batch); the arithmetic per output pixel is the same in both forms.
template <int J>
__global__ __launch_bounds__(256) void lwg_head_compose_kernel(const float* __restrict__ x, const float* __restrict__ wpk,
                                                              const float* __restrict__ bg, size_t bg_bstride, int S, int C,
                                                              float* __restrict__ pred, float* __restrict__ mask_out,
                                                              float* __restrict__ img_out) {
    constexpr int HTY = 8 * J, HHY = HTY + 4;
    __shared__ __attribute__((aligned(16))) float sx[HCH / 4][HHY][HH][4];
    __shared__ __attribute__((aligned(16))) float sw[25][HCH][4];
    const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
    // 1-D grid, XCD-aware: every XCD walks a contiguous band of tile rows, so the halo rows two vertically adjacent tiles share are
    // fetched into ONE L2 (with the 3-D grid the hardware dealt neighbouring tiles to different XCDs: PMC showed 2.4 GB fetched per
    // launch for a 0.54 GB input)
    const int tiles1 = (S + HT - 1) / HT, tilesy = (S + HTY - 1) / HTY;
    const int lid = lwg_xcd_remap(blockIdx.x, gridDim.x);
    const int b = lid / (tiles1 * tilesy), trem = lid - b * tiles1 * tilesy;
    const int x0 = (trem % tiles1) * HT, y0 = (trem / tiles1) * HTY;
    const float* xb = x + (size_t)b * S * S * C;
    float acc[J][4];
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[j][o] = 0.f;

    for (int c0 = 0; c0 < C; c0 += HCH) {
        // stage input halo: HH*HH pixels x 2 channel quads
        for (int i = tid; i < HHY * HH * (HCH / 4); i += 256) {
            const int cq = i & 1, p = i >> 1;
            const int py = p / HH, px = p - py * HH;
            const int gy = y0 + py - 2, gx = x0 + px - 2;
            floatx4 v = {0.f, 0.f, 0.f, 0.f};
            if (gy >= 0 && gy < S && gx >= 0 && gx < S)
                v = *reinterpret_cast<const floatx4*>(xb + ((size_t)gy * S + gx) * C + c0 + cq * 4);
            *reinterpret_cast<floatx4*>(&sx[cq][py][px][0]) = v;
        }
        // stage weights: wpk is [25][C][4]
        for (int i = tid; i < 25 * HCH; i += 256) {
            const int tap = i / HCH, c = i - tap * HCH;
            *reinterpret_cast<floatx4*>(&sw[tap][c][0]) =
                *reinterpret_cast<const floatx4*>(wpk + ((size_t)tap * C + c0 + c) * 4);
        }
        __syncthreads();
#pragma unroll 1
        for (int ky = 0; ky < 5; ++ky) {
#pragma unroll
            for (int kx = 0; kx < 5; ++kx) {
#pragma unroll
                for (int cq = 0; cq < HCH / 4; ++cq) {
                    floatx4 w4[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) w4[c] = *reinterpret_cast<const floatx4*>(&sw[ky * 5 + kx][cq * 4 + c][0]);
#pragma unroll
                    for (int j = 0; j < J; ++j) {
                        const floatx4 xv = *reinterpret_cast<const floatx4*>(&sx[cq][ty + 8 * j + ky][tx + kx][0]);
#pragma unroll
                        for (int c = 0; c < 4; ++c)
#pragma unroll
                            for (int o = 0; o < 4; ++o) acc[j][o] = __builtin_fmaf(xv[c], w4[c][o], acc[j][o]);   // explicit: every tile form contracts alike
                    }
                }
            }
        }
        __syncthreads();
    }
    const int gx = x0 + tx;
    if (gx >= S) return;
    const size_t plane = (size_t)S * S;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int gy = y0 + ty + 8 * j;
        if (gy >= S) continue;
        const size_t pix = (size_t)gy * S + gx;
        const float m = 1.f / (1.f + expf(-acc[j][3]));
        if (mask_out) mask_out[(size_t)b * plane + pix] = m;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float im = tanhf(acc[j][c]);
            if (img_out) img_out[((size_t)b * 3 + c) * plane + pix] = im;
            if (pred) {
                const float bgv = bg[(size_t)b * bg_bstride + c * plane + pix];
                pred[((size_t)b * 3 + c) * plane + pix] = m * bgv + (1.f - m) * im;
            }
        }
    }
}

// The head on an input stored as channel-quad planes, x (B, C/4, S, S, 4) (LWG_DT_F32_Q4: what the last decoder layer's epilogue
// writes for it).  Why a second form: lwg_head_compose_kernel sits on two bounds at once - the LDS array (four broadcast weight quads
// + four activation quads per 32 packed FMAs: twice the FMA time) and the re-fetch of its NHWC input (a stage takes 32 B of every
// pixel's 256-B row, the co-resident tiles' lines do not fit L2, so each 128-B line comes from the memory side four times: 16 GB
// per 48-frame launch at the ~7.5 TB/s that pattern sustains).  Fewer LDS reads need more pixels per thread, i.e. a larger tile,
// i.e. fewer channels per stage - which on NHWC multiplies the re-fetch (measured: 4.2 ms against 2.15).  On quad planes a stage
// of ONE channel quad reads whole lines (consecutive pixels are consecutive 16-byte quads), so:
//   a thread owns RY CONSECUTIVE rows of NX columns 32 apart (lanes stay on consecutive columns: every ds_read_b128 conflict-free);
//   for a tap column kx it reads the RY + 4 rows it needs once and feeds the five vertical taps from registers, and a tap's four weight
//   quads serve NX * RY pixels: (NX (RY + 4) + 20) LDS reads per 80 NX RY FMAs - 36 per 640 for NX = 2, RY = 4 (64 x 32-pixel tiles,
//   frame batches) against 80; the input is read 1.2 times (the halo) instead of 5.
// NX = 1, RY = 2 (32 x 16 tiles) serves launches that would leave CUs without a workgroup (one 512 x 512 frame is 128 tiles of 64 x 32).
// Both forms add an output's terms in the same order (channel quad, kx, ky, channel), products as explicit fmaf: a frame is bitwise
// the same in either.
template <int NX, int RY>
#ifndef LWG_HEADQ4_OCC
#define LWG_HEADQ4_OCC 2
#endif
__global__ __launch_bounds__(256, LWG_HEADQ4_OCC) void lwg_head_q4_kernel(const float* __restrict__ x, const float* __restrict__ wpk,
                                                         const float* __restrict__ bg, size_t bg_bstride, int S, int C,
                                                         float* __restrict__ pred, float* __restrict__ mask_out,
                                                         float* __restrict__ img_out) {
    constexpr int TW = 32 * NX, TH = 8 * RY, HX = TW + 4, HY = TH + 4, NR = RY + 4;
    __shared__ __attribute__((aligned(16))) float sx[HY][HX][4];
    __shared__ __attribute__((aligned(16))) float sw[25][4][4];
    const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
    // 1-D grid, XCD-aware: every XCD walks a contiguous band of tile rows, so the halo rows two vertically adjacent tiles share are
    // fetched into ONE L2
    const int tilesx = (S + TW - 1) / TW, tilesy = (S + TH - 1) / TH;
    const int lid = lwg_xcd_remap(blockIdx.x, gridDim.x);
    const int b = lid / (tilesx * tilesy), trem = lid - b * tilesx * tilesy;
    const int x0 = (trem % tilesx) * TW, y0 = (trem / tilesx) * TH;
    const size_t plane = (size_t)S * S;
    const float* xb = x + (size_t)b * (C >> 2) * plane * 4;
    float acc[NX][RY][4];
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int r = 0; r < RY; ++r)
#pragma unroll
            for (int o = 0; o < 4; ++o) acc[i][r][o] = 0.f;

    // The halo tile of a channel quad is NLD 16-byte loads per thread (rows of consecutive quads: whole 128-byte lines).  Quad cq + 1's
    // loads are issued before quad cq's FMAs and sit in registers meanwhile (a stage is ~6400 FMA cycles per wave, a load round trip
    // under load is of that order: issued one at a time in front of the LDS store they cost more than the FMAs).
    constexpr int NLD = (HY * HX + 255) / 256;
    int goff[NLD];                                              // element offset of this thread's halo quads inside a plane; -1: padding
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int i = tid + 256 * k;
        const int py = i / HX, px = i - py * HX;
        const int gy = y0 + py - 2, gx = x0 + px - 2;
        goff[k] = (i < HY * HX && gy >= 0 && gy < S && gx >= 0 && gx < S) ? (gy * S + gx) * 4 : -1;
    }
    floatx4 pre[NLD], wpre = {0.f, 0.f, 0.f, 0.f};
    auto fetch = [&](int cq) {
        const float* xq = xb + (size_t)cq * plane * 4;
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            pre[k] = floatx4{0.f, 0.f, 0.f, 0.f};
            if (goff[k] >= 0) pre[k] = *reinterpret_cast<const floatx4*>(xq + goff[k]);
        }
        if (tid < 100) wpre = *reinterpret_cast<const floatx4*>(wpk + ((size_t)(tid >> 2) * C + 4 * cq + (tid & 3)) * 4);   // wpk is [25][C][4]
    };
    fetch(0);
    for (int cq = 0; cq < (C >> 2); ++cq) {
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = tid + 256 * k;
            if (i < HY * HX) *reinterpret_cast<floatx4*>(&sx[0][0][0] + (size_t)i * 4) = pre[k];
        }
        if (tid < 100) *reinterpret_cast<floatx4*>(&sw[tid >> 2][tid & 3][0]) = wpre;
        __syncthreads();
        if (cq + 1 < (C >> 2)) fetch(cq + 1);
#pragma unroll 1
        for (int kx = 0; kx < 5; ++kx) {
            floatx4 av[NX][NR];
#pragma unroll
            for (int i = 0; i < NX; ++i)
#pragma unroll
                for (int rr = 0; rr < NR; ++rr) av[i][rr] = *reinterpret_cast<const floatx4*>(&sx[ty * RY + rr][tx + 32 * i + kx][0]);
#pragma unroll
            for (int ky = 0; ky < 5; ++ky) {
                floatx4 w4[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) w4[c] = *reinterpret_cast<const floatx4*>(&sw[ky * 5 + kx][c][0]);
#pragma unroll
                for (int i = 0; i < NX; ++i)
#pragma unroll
                    for (int r = 0; r < RY; ++r)
#pragma unroll
                        for (int c = 0; c < 4; ++c)
#pragma unroll
                            for (int o = 0; o < 4; ++o) acc[i][r][o] = __builtin_fmaf(av[i][r + ky][c], w4[c][o], acc[i][r][o]);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int gx = x0 + tx + 32 * i;
        if (gx >= S) continue;
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            const int gy = y0 + ty * RY + r;
            if (gy >= S) continue;
            const size_t pix = (size_t)gy * S + gx;
            const float m = 1.f / (1.f + expf(-acc[i][r][3]));
            if (mask_out) mask_out[(size_t)b * plane + pix] = m;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float im = tanhf(acc[i][r][c]);
                if (img_out) img_out[((size_t)b * 3 + c) * plane + pix] = im;
                if (pred) {
                    const float bgv = bg[(size_t)b * bg_bstride + c * plane + pix];
                    pred[((size_t)b * 3 + c) * plane + pix] = m * bgv + (1.f - m) * im;
                }
            }
        }
    }
}

// x (B, C/4, S, S, 4) channel-quad planes (LWG_DT_F32_Q4), C % 4 == 0; everything else as lwg_head_compose_f32.
extern "C" int lwg_head_compose_q4_f32(const float* x, const float* wpk, const float* bg, size_t bg_bstride, int B, int S, int C,
                                       float* pred, float* mask, float* img, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!x || !wpk || (pred && !bg) || (!pred && !mask && !img) || B <= 0 || S <= 0 || C <= 0 || (C & 3) != 0 || B > 65535)
        return (int)hipErrorInvalidValue;
    const long big = (long)((S + 63) / 64) * ((S + 31) / 32) * B;       // 64 x 32-pixel tiles
    if (big < 512) {                                                     // fewer than two per CU: 32 x 16 tiles
        const long small = (long)((S + 31) / 32) * ((S + 15) / 16) * B;
        hipLaunchKernelGGL((lwg_head_q4_kernel<1, 2>), dim3((unsigned)small), dim3(256), 0, stream, x, wpk, bg, bg_bstride, S, C, pred, mask, img);
    } else {
        hipLaunchKernelGGL((lwg_head_q4_kernel<2, 4>), dim3((unsigned)big), dim3(256), 0, stream, x, wpk, bg, bg_bstride, S, C, pred, mask, img);
    }
    return (int)hipGetLastError();
}

// Thin regressor forward: a stride-1 KS x KS convolution (pad KS / 2, no bias) with <= 4 output channels at full resolution - the
// 7x7 image head of the background network (bg_inpaintor.py:53: Conv2d(64, 3, 7, 1, 3, bias=False) before the Tanh).  As an MFMA
// launch its 3 outputs are zero-extended to 64 GEMM columns: 21x the useful flops (0.78 ms per personalization step at 512x512).
// Same register-blocked VALU form as the compose kernel above (32 x 32 pixel tile, 4 pixels x 4 outputs per thread, halo tile staged
// 8 channels at a time as channel quads); writes the PRE-activation NHWC-4 tensor (the layout the thin backward consumes), channels
// beyond the real outputs are zero because their weight columns are.
template <int KS, int J>
__global__ __launch_bounds__(256) void lwg_thin_conv_kernel(const float* __restrict__ x, const float* __restrict__ wpk, int S, int C,
                                                           float* __restrict__ y) {
    constexpr int PAD = KS / 2, HHK = HT + KS - 1, HTY = 8 * J, HHKY = HTY + KS - 1;
    __shared__ __attribute__((aligned(16))) float sx[HCH / 4][HHKY][HHK][4];
    __shared__ __attribute__((aligned(16))) float sw[KS * KS][HCH][4];
    const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
    const int tiles1 = (S + HT - 1) / HT, tilesy = (S + HTY - 1) / HTY;
    const int lid = lwg_xcd_remap(blockIdx.x, gridDim.x);
    const int b = lid / (tiles1 * tilesy), trem = lid - b * tiles1 * tilesy;
    const int x0 = (trem % tiles1) * HT, y0 = (trem / tiles1) * HTY;
    const float* xb = x + (size_t)b * S * S * C;
    float acc[J][4];
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int o = 0; o < 4; ++o) acc[j][o] = 0.f;
    for (int c0 = 0; c0 < C; c0 += HCH) {
        for (int i = tid; i < HHKY * HHK * (HCH / 4); i += 256) {
            const int cq = i & 1, p = i >> 1;
            const int py = p / HHK, px = p - py * HHK;
            const int gy = y0 + py - PAD, gx = x0 + px - PAD;
            floatx4 v = {0.f, 0.f, 0.f, 0.f};
            if (gy >= 0 && gy < S && gx >= 0 && gx < S)
                v = *reinterpret_cast<const floatx4*>(xb + ((size_t)gy * S + gx) * C + c0 + cq * 4);
            *reinterpret_cast<floatx4*>(&sx[cq][py][px][0]) = v;
        }
        for (int i = tid; i < KS * KS * HCH; i += 256) {
            const int tap = i / HCH, c = i - tap * HCH;
            *reinterpret_cast<floatx4*>(&sw[tap][c][0]) = *reinterpret_cast<const floatx4*>(wpk + ((size_t)tap * C + c0 + c) * 4);
        }
        __syncthreads();
#pragma unroll 1
        for (int ky = 0; ky < KS; ++ky) {
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
#pragma unroll
                for (int cq = 0; cq < HCH / 4; ++cq) {
                    floatx4 w4[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) w4[c] = *reinterpret_cast<const floatx4*>(&sw[ky * KS + kx][cq * 4 + c][0]);
#pragma unroll
                    for (int j = 0; j < J; ++j) {
                        const floatx4 xv = *reinterpret_cast<const floatx4*>(&sx[cq][ty + 8 * j + ky][tx + kx][0]);
#pragma unroll
                        for (int c = 0; c < 4; ++c)
#pragma unroll
                            for (int o = 0; o < 4; ++o) acc[j][o] = __builtin_fmaf(xv[c], w4[c][o], acc[j][o]);   // explicit: every tile form contracts alike
                    }
                }
            }
        }
        __syncthreads();
    }
    const int gx = x0 + tx;
    if (gx >= S) return;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int gy = y0 + ty + 8 * j;
        if (gy >= S) continue;
        floatx4 o = {acc[j][0], acc[j][1], acc[j][2], acc[j][3]};
        *reinterpret_cast<floatx4*>(y + (((size_t)b * S + gy) * S + gx) * 4) = o;
    }
}

// (B,C,P) -> (B,P,Cp), channels >= C zero-filled.  32x32 LDS tile transpose.
__global__ __launch_bounds__(256) void lwg_nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int C,
                                                              int Cp, int P) {
    __shared__ float t[32][33];
    const int b = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    for (int r = ly; r < 32; r += 8) {
        const int c = c0 + r, p = p0 + lx;
        t[r][lx] = (c < C && p < P) ? src[((size_t)b * C + c) * P + p] : 0.f;
    }
    __syncthreads();
    for (int r = ly; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + lx;
        if (p < P && c < Cp) dst[((size_t)b * P + p) * Cp + c] = t[lx][r];
    }
}

// (B,P,Cs) -> (B,C,P) taking the first C channels.
__global__ __launch_bounds__(256) void lwg_nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst, int C,
                                                              int Cs, int P) {
    __shared__ float t[32][33];
    const int b = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    for (int r = ly; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + lx;
        t[r][lx] = (p < P && c < C) ? src[((size_t)b * P + p) * Cs + c] : 0.f;
    }
    __syncthreads();
    for (int r = ly; r < 32; r += 8) {
        const int c = c0 + r, p = p0 + lx;
        if (c < C && p < P) dst[((size_t)b * C + c) * P + p] = t[lx][r];
    }
}

// x (B,S,S,C) NHWC, C % 8 == 0; wpk [25][C][4] (out 0..2 = img_reg, out 3 = att_reg; tap = ky*5+kx);
// bg (Bbg,3,S,S) NCHW with batch stride bg_bstride floats (0 = shared); pred (B,3,S,S), mask (B,1,S,S),
// img (B,3,S,S): any of the three outputs may be NULL (bg may be NULL when pred is NULL).
extern "C" int lwg_head_compose_f32(const float* x, const float* wpk, const float* bg, size_t bg_bstride, int B, int S, int C,
                                    float* pred, float* mask, float* img, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!x || !wpk || (pred && !bg) || (!pred && !mask && !img) || B <= 0 || S <= 0 || C <= 0 || (C % HCH) != 0 || B > 65535)
        return (int)hipErrorInvalidValue;
    const int tiles = (S + HT - 1) / HT;
    if ((long)tiles * tiles * B < 512) {          // fewer 32 x 32 tiles than two per CU: 32 x 16 tiles
        const int ty2 = (S + 15) / 16;
        hipLaunchKernelGGL(lwg_head_compose_kernel<2>, dim3(tiles * ty2 * B), dim3(256), 0, stream, x, wpk, bg, bg_bstride, S, C, pred, mask, img);
    } else {
        hipLaunchKernelGGL(lwg_head_compose_kernel<4>, dim3(tiles * tiles * B), dim3(256), 0, stream, x, wpk, bg, bg_bstride, S, C, pred, mask, img);
    }
    return (int)hipGetLastError();
}

// x (B,S,S,C) NHWC, C % 8 == 0; wpk [ks*ks][C][4] (tap = ky*ks + kx; unused output columns zero); ks = 5 or 7, stride 1, pad ks/2,
// no bias, no activation -> y (B,S,S,4) NHWC.
extern "C" int lwg_thin_conv_f32(const float* x, const float* wpk, int B, int S, int C, int ks, float* y, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!x || !wpk || !y || B <= 0 || S <= 0 || C <= 0 || (C % HCH) != 0 || B > 65535 || (ks != 5 && ks != 7)) return (int)hipErrorInvalidValue;
    const int tiles = (S + HT - 1) / HT, ty2 = (S + 15) / 16;
    const bool small = (long)tiles * tiles * B < 512;      // as lwg_head_compose_f32: 32 x 16 tiles when 32 x 32 ones leave CUs idle
    const dim3 grid(small ? tiles * ty2 * B : tiles * tiles * B);
    if (ks == 7) {
        if (small) hipLaunchKernelGGL((lwg_thin_conv_kernel<7, 2>), grid, dim3(256), 0, stream, x, wpk, S, C, y);
        else hipLaunchKernelGGL((lwg_thin_conv_kernel<7, 4>), grid, dim3(256), 0, stream, x, wpk, S, C, y);
    } else {
        if (small) hipLaunchKernelGGL((lwg_thin_conv_kernel<5, 2>), grid, dim3(256), 0, stream, x, wpk, S, C, y);
        else hipLaunchKernelGGL((lwg_thin_conv_kernel<5, 4>), grid, dim3(256), 0, stream, x, wpk, S, C, y);
    }
    return (int)hipGetLastError();
}

extern "C" int lwg_nchw_to_nhwc_f32(const float* src, float* dst, int B, int C, int Cp, int P, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!src || !dst || B <= 0 || C <= 0 || Cp < C || P <= 0 || B > 65535) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(lwg_nchw_to_nhwc_kernel, dim3((P + 31) / 32, (Cp + 31) / 32, B), dim3(256), 0, stream, src, dst, C, Cp, P);
    return (int)hipGetLastError();
}

extern "C" int lwg_nhwc_to_nchw_f32(const float* src, float* dst, int B, int C, int Cs, int P, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!src || !dst || B <= 0 || C <= 0 || Cs < C || P <= 0 || B > 65535) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(lwg_nhwc_to_nchw_kernel, dim3((P + 31) / 32, (C + 31) / 32, B), dim3(256), 0, stream, src, dst, C, Cs, P);
    return (int)hipGetLastError();
}

// Output conversion of Imitator.inference (models/imitator.py:368-372 -> cv_utils.save_cv2_img(normalize=True),
// tools/utils/filesio/cv_utils.py:100-116): (B,3,S,S) fp32 in [-1,1] -> (B,S,S,3) uint8 HWC, value
// uint8((x + 1) / 2.0 * 255) with numpy's fp32 arithmetic (each operation rounded, conversion truncates toward zero).
// bgr = 1 writes the channel order the reference hands to cv2.imwrite.  HBM-bound: 12 B in, 3 B out per pixel; done on
// the device so the D2H copy is 0.75 MB/frame instead of 3 MB and the host threads only encode.
__global__ void lwg_frames_to_u8_kernel(const float* __restrict__ pred, int B, int P, int bgr, unsigned char* __restrict__ out) {
    const size_t total = (size_t)B * P;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / P, pix = i - b * P;
        const float* src = pred + b * 3 * (size_t)P + pix;
        unsigned char v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float x = src[(size_t)c * P];
            const float y = __fmul_rn(__fmul_rn(__fadd_rn(x, 1.f), 0.5f), 255.f);
            v[c] = (unsigned char)(int)y;
        }
        unsigned char* o = out + i * 3;
        o[0] = bgr ? v[2] : v[0];
        o[1] = v[1];
        o[2] = bgr ? v[0] : v[2];
    }
}

extern "C" int lwg_frames_to_u8(const float* pred, int B, int S, int bgr, uint8_t* out, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!pred || !out || B <= 0 || S <= 0) return (int)hipErrorInvalidValue;
    const size_t total = (size_t)B * S * S;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(lwg_frames_to_u8_kernel, dim3(blocks), dim3(256), 0, stream, pred, B, S * S, bgr, out);
    return (int)hipGetLastError();
}
