// BASELINE configs[3] (bf16 activations), the LAST stage of forward_tsf as ONE kernel: the decoder's last up-sampling layer
// ConvTranspose2d(128 -> 64, 4, 2, 1) + ReLU (attlwb_spade_resunet.py:331-340), the two bias-free 5x5 regressors tsf_img_reg (tanh) / tsf_att_reg (sigmoid)
// (:605-613) and the compositing pred = mask * bg + (1 - mask) * img (models/imitator.py:393) - the (B, 2H, 2W, 64) tensor between them is never
// written to HBM (2.7 GB written by lwg_conv_bf16_up4_kernel and 2.9 GB read back by lwg_head_bf16_kernel per 20 frames at 1024 x 1024: DESIGN.md 3.11
// measured the up-sampling launch as HBM time + MFMA time ADDED UP - the waves that issue MFMAs are the waves that wait in the store queue).
//
// Workgroup = 8 waves = one 12 x 28-pixel tile of the final frame.  The head's 5x5 window needs the 16 x 32-pixel block of the 64-channel tensor around it
// (2-pixel halo), i.e. the transposed convolution's outputs of an 8 x 16 block of input pixels ("centres"), i.e. a 10 x 18 input halo (128 channels: two
// 64-channel chunks, 46 KB, LDS-DMA).  The halo of neighbouring tiles is recomputed: (16 x 32) / (12 x 28) = 1.52 x the layer's MFMAs, which a launch
// that was ~30 % matrix-bound can afford.
//   phase 1  lwg_conv_bf16_up4_kernel's K loop (halo tile + register-streamed weights + row renaming: one fragment read feeds both vertical taps); a wave
//            = one output parity x one 32-channel tile x all eight centre rows (a weight fragment feeds four MFMAs); bias + ReLU, pixels outside
//            the image forced to ZERO (the head pads with zeros), bf16, into the LDS tile T[16][32][128 B] in lwg_head_bf16_kernel's layout (k-octet ^ (column & 7));
//   phase 2  lwg_head_bf16_kernel's MFMA form on T (v_mfma_f32_16x16x32_bf16: 16 rows = 4 taps x 4 outputs; a fifth-tap pass), three (row, 16-column
//            block) units per wave, the per-tap partial sums shifted through LDS (the input halo's space), tanh / sigmoid / compositing, fp32 NCHW planes.
// Same arithmetic in the same order as the two kernels it replaces (bf16 rounding of the intermediate included): check_bf16_up4_head compares them.
#include "lwg_common.h"
#include "lwg_conv_args.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));
typedef float floatx4v __attribute__((ext_vector_type(4)));

#define UH_OOB 0xC0000000u
#define UH_TH 12                          // final-frame rows per tile
#define UH_TW 28                          // ... and columns
#define UH_IC 32                          // columns of the intermediate tile (UH_TW + 4); rows: UH_TH + 4 = 16
#define UH_HALO_W 18
#define UH_HALO_PIX 180
#define UH_PIECES 23                      // 8 pixels per 1 KB DMA piece; the 23rd is half used
#define UH_HALO_BYTES (UH_PIECES * 1024)
#define UH_T_OFF (4 * UH_HALO_BYTES)      // the intermediate tile behind the two halo BUFFERS (two 64-channel chunks each)
#define UH_T_BYTES (16 * UH_IC * 128)
#define UH_PW (UH_IC + 4)                 // partial-sum row length: output column c is stored at c + 3
#define UH_BIAS_OFF (UH_T_OFF + UH_T_BYTES)   // the layer's 64 bias values (256 B)
#define UH_LDS (UH_BIAS_OFF + 256)
// LDS byte address of a pointer into the kernel's dynamic shared memory (what M0 takes for an LDS-DMA)
#define UH_LDS_BASE_OF(p) ((unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)(p))

// lab instrumentation (compiled out of the product): tools/up4headlab.py on a -DUH_LAB_TS variant library - waves 0 and 7 of every workgroup stamp the
// phase boundaries of the workgroup's SECOND tile into mask_out (reinterpreted: 2 waves x 8 stamps of 64 bits per workgroup)
#ifdef UH_LAB_TS
#define UTS(i) do { if (it == 1 && lane == 0 && (wid == 0 || wid == 7)) reinterpret_cast<unsigned long long*>(mask_out)[((size_t)blockIdx.x * 2 + (wid == 7)) * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define UTS(i) do { } while (0)
#endif

__device__ __forceinline__ unsigned uh_pack_bf16x2(float lo, float hi) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned, v);
}

__global__ __launch_bounds__(512, 1) void lwg_up4_head_bf16_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ wup, const float* __restrict__ bias_up,
                                                                  const __bf16* __restrict__ whead, const float* __restrict__ bg, size_t bg_bstride, int H, int W, int nimg,
                                                                  unsigned xbytes, float* __restrict__ pred, float* __restrict__ mask_out,
                                                                  float* __restrict__ img_out) {
    constexpr int NCH = 2, N = 64, TM = 4, NDY = 2, NDX = 2, NTAPS = 4, NE = TM + NDY - 1, FPC = NDX * 4 * NDY, R = 8, FPP = NCH * FPC;
    constexpr int ROWB = UH_HALO_W * 128;
    extern __shared__ __attribute__((aligned(16))) char sm[];
    char* const T = sm + UH_T_OFF;                            // [16][32][128 B]: ReLU(convT) of the tile + 2-pixel halo, bf16, zero outside the image
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int OH = 2 * H, OW = 2 * W;
    const int tiles_x = (OW + UH_TW - 1) / UH_TW, tiles_y = (OH + UH_TH - 1) / UH_TH;
    // PERSISTENT workgroups (one per CU: LDS): workgroup g walks the tiles of ITS contiguous range of the (frame, tile row, tile column) order - so the
    // halo rows of vertically adjacent tiles meet in one XCD's L2 - and requests tile k + 1's input halo (LDS-DMA into the OTHER halo buffer) and first
    // weight fragments as soon as tile k's K loops are issued: they land under tile k's epilogue and head phase instead of in front of tile k + 1
    const int total = tiles_x * tiles_y * nimg;
    const int g0 = lwg_xcd_remap(blockIdx.x, gridDim.x);
    const int per = total / (int)gridDim.x, extra = total % (int)gridDim.x;
    const int t_first = g0 * per + (g0 < extra ? g0 : extra), t_count = per + (g0 < extra ? 1 : 0);
    __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(x), 0, (int)xbytes, 0x00020000);
    // the 10 x 18 x 128-channel input halo of tile `lid` by LDS-DMA into halo buffer `buf`: piece = 8 halo pixels x 128 B of one 64-channel chunk, k-octets at
    // slot octet ^ ((halo column >> 1) & 7); out-of-image pixels: out-of-range offsets (zeros)
    auto stage = [&](int lid, int buf) {
        const int b = lid / (tiles_x * tiles_y), trem = lid - b * tiles_x * tiles_y;
        const int cy0 = (((trem / tiles_x) * UH_TH) >> 1) - 1, cx0 = (((trem % tiles_x) * UH_TW) >> 1) - 1;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int piece = wid + 8 * q;                     // 0 .. 45: chunk = piece / 23
            if (piece < 2 * UH_PIECES) {
                const int c = piece >= UH_PIECES ? 1 : 0, pp = piece - c * UH_PIECES;
                const int hp = pp * 8 + (lane >> 3);
                const int hy = hp / UH_HALO_W, hx = hp - hy * UH_HALO_W;
                const int gy = cy0 + hy - 1, gx = cx0 + hx - 1;
                const bool ok = hp < UH_HALO_PIX && gy >= 0 && gy < H && gx >= 0 && gx < W;
                const unsigned voff = ok ? (unsigned)((b * H + gy) * W + gx) * 256u + (unsigned)((lane & 7) ^ ((hx >> 1) & 7)) * 16u : UH_OOB;
                // (inline asm, not __builtin_amdgcn_raw_ptr_buffer_load_lds: the compiler treats every LDS-DMA as an access to ALL of LDS and puts an
                // s_waitcnt vmcnt(0) in front of each one - six serialized round trips per tile, behind every load issued before them: 12.8 k cycles in the
                // first persistent version, tools/up4headlab.py --ts.  The explicit vmcnt(0) waits in front of the barriers cover these requests.)
                const unsigned ldsaddr = (unsigned)(UH_LDS_BASE_OF(sm) + buf * 2 * UH_HALO_BYTES + piece * 1024);
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                             :: "s"(ldsaddr), "v"(voff), "s"(rA), "s"(c * 128) : "memory");
            }
        }
    };
    // ---- phase 1 constants: wave = (output parity wid / 2, 32-channel tile wid % 2), ALL eight centre rows (TM = 4 row tiles: image rows (i, i + 4)): a
    // weight fragment (1 KB from L2) feeds FOUR MFMAs.  (First version: two parities per wave, four rows - one fragment per TWO MFMAs = 64 B / clk / CU at
    // matrix-pipe saturation, the L2 -> CU fill rate: with the stores gone THAT bound the K loops - 15.5 k cycles per tile for 8.2 k of matrix work,
    // tools/up4headlab.py --ts.)
    const int par = wid >> 1, py = par >> 1, px = par & 1, wn = wid & 1;
    const int khalf = lane >> 5;
    const unsigned wv = (unsigned)((wn * 32 + (lane & 31)) * 32 + khalf * 16);
    const unsigned ppanel = (unsigned)NCH * NTAPS * (unsigned)N * 128u;            // bytes of one parity's panel
    __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(wup), 0, (int)(4u * ppanel), 0x00020000);
    bf16x8 bq[R];
    auto load_w = [&](int qp, int slot) {                   // fragment qp of this wave's parity: (chunk, tap column, k-step, tap row)
        const int chunk = qp / FPC, ql = qp % FPC;
        const int dxi = ql / (4 * NDY), ks = (ql / NDY) & 3, dyi = ql % NDY;
        const int step = chunk * NTAPS + dyi * NDX + dxi;
        bq[slot] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
            rW, (int)wv, (int)((unsigned)par * ppanel + (unsigned)(step * 4 + ks) * (unsigned)N * 32u), 0));
    };
    const int hx0 = (lane & 15) + 1;
    // fragment E_0's halo pixel for the first tap column: halo row (py - 1) + 1 + TM * (second half of the 32 lanes)
    const int prow0 = ((py - 1) + TM * ((lane >> 4) & 1) + 1) * UH_HALO_W + hx0 + (px - 1);
    if (t_count > 0) stage(t_first, 0);
#pragma unroll
    for (int q = 0; q < R; ++q) load_w(q, q);
    if (tid < 64) reinterpret_cast<float*>(sm + UH_BIAS_OFF)[tid] = bias_up[tid];      // the layer's bias: read from LDS in the epilogue (as four global loads
                                                                                        // per use each one was a waited-for L2 round trip: 2-3 k cycles per tile)
    __builtin_amdgcn_s_waitcnt(0x0f70);                       // vmcnt(0): this wave's DMA pieces of the first tile (and its first weights) have landed
#pragma unroll 1
    for (int it = 0; it < t_count; ++it) {
    const int lid = t_first + it, hb = it & 1;
    char* const Ah = sm + hb * 2 * UH_HALO_BYTES;             // this tile's input halo [2 chunks][UH_HALO_BYTES], later the head's partial sums
    const int b = lid / (tiles_x * tiles_y), trem = lid - b * tiles_x * tiles_y;
    const int ox0 = (trem % tiles_x) * UH_TW, oy0 = (trem / tiles_x) * UH_TH;
    UTS(0);
    __syncthreads();                                          // everybody's DMA pieces of this tile have landed (each wave waited for its own: before the loop /
                                                              // before the previous tile's last barrier); the previous tile's partial sums and T are consumed
    UTS(1);
    // this thread's pixel of the final pass (threads 0 .. 335) and its background values: requested now, used after the head phase
    const size_t plane = (size_t)OH * OW;
    const int fr = tid / UH_TW, fc = tid - fr * UH_TW;
    const bool fin = tid < UH_TH * UH_TW && oy0 + fr < OH && ox0 + fc < OW;
    const size_t fpix = fin ? (size_t)(oy0 + fr) * OW + (ox0 + fc) : 0;
    float bgv[3] = {0.f, 0.f, 0.f};
    if (pred && fin) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) bgv[ch] = bg[(size_t)b * bg_bstride + ch * plane + fpix];
    }
    {
        floatx16 acc[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
        for (int chunk = 0; chunk < NCH; ++chunk) {
            // the NEXT tile's input halo into the other halo buffer (free: its partial sums were consumed before this tile's first barrier): requested
            // between the two channel chunks, so that the six requests per wave do not queue up behind everybody's burst at the end of the phase
            if (chunk == 1 && it + 1 < t_count) stage(lid + 1, hb ^ 1);
            const char* Acur = Ah + chunk * UH_HALO_BYTES;
            bf16x8 E[2][NE];
            auto read_e = [&](int g, int buf) {
                const int dxi = g >> 2, ks = g & 3;
                int pxl_ = prow0 + dxi;
                asm volatile("" : "+v"(pxl_));
                const int sw = ((hx0 + (px - 1) + dxi) >> 1) & 7;
                const char* p = Acur + (pxl_ << 7) + (((2 * ks + khalf) ^ sw) << 4);
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
                    for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq[slot], E[g & 1][i + dyi], acc[i], 0, 0, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, TM, 0);
                    const int nq = qp + R;                 // refill the slot just consumed - the ring WRAPS into the next tile (same parity, same weights):
                    load_w(nq < FPP ? nq : nq - FPP, slot);  // it never drains, a tile's first fragments are in flight since the previous tile's last groups
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        UTS(2);
        // epilogue into T: acc[i][4 g + c] = centre (row i + 4 * ((lane >> 4) & 1), column lane & 15), channel wn * 32 + 8 g + 4 khalf + c
        // (the lane id through an empty asm per tile: the address arithmetic of the epilogue and of the head phase must not be hoisted out of the tile
        // loop - it would occupy ~60 registers through the K loop)
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int khalf_e = lane_e >> 5;
        // (the epilogue is vector-ALU work between two barriers - 64 outputs per lane: the bias quads are read once, ReLU is one v_max per value, the
        // out-of-image test is applied to the PACKED words - 8 selects per row tile instead of 16 compares + 16 selects + 8 merges; r06_d: 8-11 k cycles
        // of a 27 k-cycle tile were spent here)
        floatx4 b4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) b4[g] = *reinterpret_cast<const floatx4*>(sm + UH_BIAS_OFF + (wn * 32 + 8 * g + 4 * khalf_e) * 4);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int cr = i + TM * ((lane_e >> 4) & 1), cc = lane_e & 15;
            const int iy = 2 * cr + py, ix = 2 * cc + px;      // pixel inside the 16 x 32 intermediate tile
            const int gy = oy0 - 2 + iy, gx = ox0 - 2 + ix;    // ... and inside the frame
            const bool inside = gy >= 0 && gy < OH && gx >= 0 && gx < OW;
            unsigned pk[4][2];
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const float v0 = __builtin_fmaxf(acc[i][4 * g + 2 * d] + b4[g][2 * d], 0.f), v1 = __builtin_fmaxf(acc[i][4 * g + 2 * d + 1] + b4[g][2 * d + 1], 0.f);
                    const unsigned w2 = uh_pack_bf16x2(v0, v1);
                    pk[g][d] = inside ? w2 : 0u;
                }
            // the two half-waves exchange halves (as lwg_bf16_epilogue's store): lanes 0..31 end up with channels [0, 16), lanes 32..63 with [16, 32) of the tile
            uintx4 st[2];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    auto sres = __builtin_amdgcn_permlane32_swap(pk[h][d], pk[2 + h][d], false, false);
                    st[h][d] = sres[0];
                    st[h][2 + d] = sres[1];
                }
            char* prow = T + (iy * UH_IC + ix) * 128;
            const int oct = wn * 4 + 2 * khalf_e;
            *reinterpret_cast<uintx4*>(prow + (((oct) ^ (ix & 7)) << 4)) = st[0];
            *reinterpret_cast<uintx4*>(prow + (((oct + 1) ^ (ix & 7)) << 4)) = st[1];
        }
    }
    // ---- the head's weight fragments (per kernel row ky: 4 x 16 B per lane, L2), one kernel row ahead of the MFMAs that use them (two register sets: all
    // twenty at once cost 80 registers - the weight ring of phase 1 had to shrink for them)
    bf16x8 wf[2][2][2];
    int lane_w = lane;
    asm volatile("" : "+v"(lane_w));
    auto load_wf = [&](int ky, int set) {
#pragma unroll
        for (int ps = 0; ps < 2; ++ps)
#pragma unroll
            for (int ch = 0; ch < 2; ++ch)
                wf[set][ps][ch] = *reinterpret_cast<const bf16x8*>(whead + ((((size_t)ky * 2 + ps) * 2 + ch) * 64 + lane_w) * 8);
    };
    load_wf(0, 0);
    // ---- phase 2: the 5x5 regressors on T; a unit = (output row, 16-column block): 24 units, three per wave
    UTS(3);
    __syncthreads();                                          // T is complete; nobody reads the input halo any more
    UTS(4);
    float* const part = reinterpret_cast<float*>(Ah);          // [UH_TH][5][UH_PW][4]
    {
        int lane_h = lane;
        asm volatile("" : "+v"(lane_h));
        const int pxl = lane_h & 15, koct = lane_h >> 4;
        floatx4v acc2[3][2];
#pragma unroll
        for (int u = 0; u < 3; ++u) acc2[u][0] = acc2[u][1] = floatx4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 5; ++ky) {
            if (ky + 1 < 5) load_wf(ky + 1, (ky + 1) & 1);
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int unit = wid + 8 * u, r = unit >> 1, cb = unit & 1;
                    const int iy = r + ky, ix = cb * 16 + pxl;
                    const bf16x8 av = *reinterpret_cast<const bf16x8*>(T + ((size_t)iy * UH_IC + ix) * 128 + (((ch * 4 + koct) ^ (ix & 7)) << 4));
                    acc2[u][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ky & 1][0][ch], av, acc2[u][0], 0, 0, 0);
                    acc2[u][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ky & 1][1][ch], av, acc2[u][1], 0, 0, 0);
                }
            }
        }
        // lane (pixel j, tap t): the 4 outputs of tap t for intermediate column cb * 16 + j -> output column c = cb * 16 + j - t (fifth tap: c = .. - 4)
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int unit = wid + 8 * u, r = unit >> 1, cb = unit & 1;
            const int j = cb * 16 + pxl;
            *reinterpret_cast<floatx4v*>(part + (((size_t)r * 5 + koct) * UH_PW + (j - koct + 3)) * 4) = acc2[u][0];
            if (koct == 0 && j >= 1) *reinterpret_cast<floatx4v*>(part + (((size_t)r * 5 + 4) * UH_PW + (j - 4 + 3)) * 4) = acc2[u][1];
        }
    }
    UTS(5);
    // this wave's DMA pieces of the NEXT tile, its first weights and the background values: waited for HERE (they had the head phase to land), in front of
    // the final pass - so that nothing ever waits for that pass's global stores (the counter retires in order)
    __builtin_amdgcn_s_waitcnt(0x0f70);
    UTS(6);
    __syncthreads();
    if (fin) {
        floatx4v s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 5; ++t) s += *reinterpret_cast<const floatx4v*>(part + (((size_t)fr * 5 + t) * UH_PW + fc + 3) * 4);
        const float m = 1.f / (1.f + expf(-s[3]));
#ifndef UH_LAB_TS
        if (mask_out) mask_out[(size_t)b * plane + fpix] = m;
#endif
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float im = tanhf(s[ch]);
            if (img_out) img_out[((size_t)b * 3 + ch) * plane + fpix] = im;
            if (pred) pred[((size_t)b * 3 + ch) * plane + fpix] = m * bgv[ch] + (1.f - m) * im;
        }
    }
    UTS(7);
    }
}

// args: the launch description of lwg_conv_transpose4_nhwc_bf16 for the layer (parity-(0, 0) description: ntaps = 4, stride = 1, omul = 2, OH = H, OW = W, one
// bf16 input with C0 = 128, N = 64, LWG_EPI_NONE + ReLU, args->w = the four register-streamed parity panels [4][Cin/64 * 4][4][N][16], args->bias (64)) - its
// y / YH / YW / YC are ignored: the layer's output is never written.  whead: packing.pack_head_bf16's panel ([5][2][2][64][8] bf16).  bg (B or 1, 3, 2H, 2W)
// fp32 with batch stride bg_bstride (0: one background for all frames); pred / mask / img: fp32 NCHW planes (B, 3 | 1 | 3, 2H, 2W), any of them NULL.
// Any batch size: the input goes through 32-bit buffer offsets, larger batches run in slices of frames.
extern "C" int lwg_up4_head_compose_bf16(const LwgConvArgs* pa, const void* whead, const float* bg, size_t bg_bstride, float* pred, float* mask, float* img,
                                         lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!pa) return (int)hipErrorInvalidValue;
    const LwgConvArgs& a = *pa;
    if (!a.x0 || !a.w || !a.bias || !whead || (pred && !bg) || (!pred && !mask && !img) || a.B <= 0 || a.H <= 0 || a.W <= 0 || a.C0 != 128 || a.C1 != 0 ||
        a.N != 64 || a.ntaps != 4 || a.stride != 1 || a.omul != 2 || a.OH != a.H || a.OW != a.W || a.xdt != LWG_DT_BF16 || a.epi != LWG_EPI_NONE ||
        a.act != LWG_ACT_RELU)
        return (int)hipErrorInvalidValue;
    const unsigned long long per = (unsigned long long)a.H * a.W * 256ull;
    if (per >= 0xC0000000ull) return (int)hipErrorInvalidValue;
    const int nbs = (int)((0xC0000000ull - 1ull) / per);                        // frames per launch (32-bit buffer offsets)
    const size_t plane = (size_t)4 * a.H * a.W;
    const int tiles = ((2 * a.W + UH_TW - 1) / UH_TW) * ((2 * a.H + UH_TH - 1) / UH_TH);
    const int cus = lwg_device_cus();                                          // persistent workgroups: one per CU (156 KB of LDS each)
    static unsigned long long done = 0;
    if (hipError_t e = lwg_allow_dynamic_lds(reinterpret_cast<const void*>(lwg_up4_head_bf16_kernel), (size_t)UH_LDS, done); e != hipSuccess) return (int)e;
    for (int b0 = 0; b0 < a.B; b0 += nbs) {
        const int nb = a.B - b0 < nbs ? a.B - b0 : nbs;
        const __bf16* xs = reinterpret_cast<const __bf16*>(a.x0) + (size_t)b0 * a.H * a.W * 128;
        const long total = (long)tiles * nb;
        hipLaunchKernelGGL(lwg_up4_head_bf16_kernel, dim3((unsigned)(total < cus ? total : cus)), dim3(512), (size_t)UH_LDS, stream, xs,
                           reinterpret_cast<const __bf16*>(a.w), a.bias, reinterpret_cast<const __bf16*>(whead), bg ? bg + (size_t)b0 * bg_bstride : bg, bg_bstride,
                           a.H, a.W, nb, (unsigned)((unsigned long long)nb * per), pred ? pred + (size_t)b0 * 3 * plane : pred, mask ? mask + (size_t)b0 * plane : mask,
                           img ? img + (size_t)b0 * 3 * plane : img);
    }
    return (int)hipGetLastError();
}
