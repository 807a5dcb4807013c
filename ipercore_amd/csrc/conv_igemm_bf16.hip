// NHWC implicit-GEMM convolution, bf16 end to end: bf16 activations in HBM, bf16 MFMA operands
// (v_mfma_f32_32x32x16_bf16), fp32 accumulation and epilogue arithmetic, bf16 output - BASELINE configs[3]
// ("novel_view 1024x1024 bf16, MFMA bf16 conv tiles").  Replaces the same torch.nn.Conv2d / ConvTranspose2d calls as
// csrc/conv_igemm.hip (reference generators/attlwb_spade_resunet.py:14-25, :73-93, :202-204, :268-271, :331-340).
//
// Same GEMM view as the fp32 kernel (D[M,N] = A[M,K] Wp[K,N], M = B*OH*OW pixels, K = taps*Cin ordered channel-chunk major /
// tap minor so the shifted re-reads of an activation chunk are L2 hits) with everything that made that kernel the wrong shape
// for a 16x faster matrix pipe removed:
//   * activations are STORED as bf16 (half the HBM / L2 bytes of the round-1 kernel, no conversion in the loop); a K-step is
//     64 channels of one tap = one full 128-byte line per pixel;
//   * both operands go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, 16 B per lane, no VGPR round trip, no ds_write):
//     a wave-instruction moves 8 rows x 128 B.  The LDS image is the plain [row][128 B] tile; the eight 16-byte k-octets of a row
//     sit at slot  octet ^ ((row >> 1) & 7)  - the permutation is applied to the per-lane GLOBAL address of the A gather and
//     baked into the packed weight panel, the LDS destination stays lane-linear as the DMA requires - which makes every
//     ds_read_b128 fragment read conflict-free (16 lanes of a read group hit 16 distinct 4-bank groups);
//   * padding pixels use an out-of-range buffer offset: the DMA writes zeros for them (probed: tools/probes/dma_probe.hip);
//   * 128 x 128 (or 128 x 64) output tile per 4-wave workgroup, BK = 64: 16 MFMAs per wave between barriers, two LDS stages
//     (64 KB), two workgroups per CU - one computes while the other waits for its DMA / barrier;
//   * D^T accumulators (weights as the row operand): a lane owns one pixel and 4 consecutive channels per 8-channel group; the
//     two half-waves exchange halves (v_permlane32_swap) so a lane stores 8 consecutive channels = one 16-byte bf16 store.
// Epilogues: bias, ReLU / tanh / sigmoid, residual add, SPADE's IN(x) * (1 + gamma) + beta - all read / written as bf16.
#include <stdlib.h>

#include "lwg_common.h"
#include "lwg_conv_args.h"

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
template <int TM, int TN, int EPI>
__device__ __forceinline__ void lwg_bf16_epilogue(const LwgConvArgs& a, floatx16 (&acc)[TM][TN], int m_base, int n_base, int wm, int wn,
                                                  int lane) {
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
            *reinterpret_cast<uintx4*>(p16 + 16 * khalf) = st[0];
            *reinterpret_cast<uintx4*>(p16 + 16 * khalf + 8) = st[1];
        }
    };
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m_base + wm * TM * 32 + i * 32 + (lane & 31);
        const bool live = m < a.M;
        const int mm = live ? m : 0;
        size_t opix = (size_t)mm;
        int bimg = 0;
        if (!direct || EPI == LWG_EPI_SPADE) {
            const int b = mm / HW;
            bimg = b;
            if (!direct) {
                const int rem = mm - b * HW;
                const int oy = rem / a.OW, ox = rem - oy * a.OW;
                opix = ((size_t)b * a.YH + (oy * a.omul + a.ooy)) * a.YW + (ox * a.omul + a.oox);
            }
        }
        if (EPI == LWG_EPI_SPADE) {
            static_assert(EPI != LWG_EPI_SPADE || TN == 2, "SPADE epilogue needs gamma|beta in one wave");
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
                    const float bt = acc[i][1][4 * g + c] + bb4[c];
                    o[g][c] = lwg_act((xf[g][c] - mu[c]) * rs[c] * (1.f + gm) + bt, a.act);
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
                        o[g][c] = lwg_act(acc[i][j][4 * g + c] + b4[c] + (EPI == LWG_EPI_RESIDUAL ? r4[g][c] : 0.f), a.act);
                }
                store_dt(yb + opix * a.YC + a.ycoff + ncol, live, o);
            }
        }
    }
}

template <int WAVES_M, int WAVES_N, int TM, int TN, int EPI, bool DMA_A>
__global__ __launch_bounds__(256, 2) void lwg_conv_bf16_kernel(const LwgConvArgs a) {
    constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
    constexpr int A_STAGE = BM * 128, B_STAGE = BN * 128;    // bytes: [row][64 bf16]
    constexpr int PA = BM / 32;                              // 8-row DMA pieces per wave per K-step (A)
    constexpr int PB = BN / 32;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");

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
// Deep-pipeline variant: BK = 32 per stage, FOUR LDS stages (the same 64 KB per 128 x 128 tile, two workgroups per CU), the
// DMA of K-step t+3 issued while step t computes.  One barrier per step does both jobs: "everyone's stage-t pieces have landed"
// (each wave first waits for its OWN pieces with a counted s_waitcnt vmcnt(N) that leaves the younger stages' DMAs in flight -
// never vmcnt(0) inside the loop) and "everyone is done reading stage t-1", which is the stage the new DMAs overwrite.
// LDS image: [row][32 bf16] = 64-byte rows; k-octet o of row r sits at slot o ^ ((r >> 2) & 3) (conflict-free ds_read_b128 for
// 64-byte rows); a DMA piece = 16 rows.  Weight panel: [ntaps*Cin/32][N][32], K order of the fp32 panel (32-channel chunk major,
// tap minor), slots permuted the same way (ops._w16v3).
#define LWG_WAIT_VM_LGKM0(n) asm volatile("s_waitcnt vmcnt(" #n ") lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <int WAVES_M, int WAVES_N, int TM, int TN, int EPI>
__global__ __launch_bounds__(256, 2) void lwg_conv_bf16_kernel4(const LwgConvArgs a) {
    constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
    constexpr int NST = 4;
    constexpr int A_STAGE = BM * 64, B_STAGE = BN * 64;      // bytes: [row][32 bf16]
    constexpr int PA = BM / 64;                              // 16-row DMA pieces per wave per K-step
    constexpr int PB = BN / 64;
    static_assert(WAVES_M * WAVES_N == 4 && PA + PB == 4, "4 waves; the vmcnt thresholds below assume 4 DMA instructions per wave and step");

    extern __shared__ __attribute__((aligned(16))) char smem_d[];
    char* As = smem_d;
    char* Bs = smem_d + NST * A_STAGE;
    int* taptab = reinterpret_cast<int*>(smem_d + NST * A_STAGE + NST * B_STAGE);  // [3][LWG_MAX_TAPS]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WAVES_N, wn = wid % WAVES_N;
    const int tiles_n = a.N / BN;
    const int lid = lwg_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = lid % tiles_n, tile_m = lid / tiles_n;
    const int m_base = tile_m * BM, n_base = tile_n * BN;

    const int HW = a.OH * a.OW;
    const int Cin = a.C0 + a.C1;
    int pixlin[PA], piy[PA], pix[PA];
    unsigned long long vmask[PA];
    const unsigned chunk16 = (unsigned)((lane & 3) ^ (lane >> 4)) * 16u;     // row = piece*16 + lane/4: (row >> 2) & 3 = lane >> 4
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        const int r = (wid * PA + p) * 16 + (lane >> 2);
        const int m = m_base + r;
        const bool ok = m < a.M;
        const int mm = ok ? m : 0;
        const int b = mm / HW, rem = mm - b * HW;
        const int oy = rem / a.OW, ox = rem - oy * a.OW;
        piy[p] = ok ? oy * a.stride : -100000;
        pix[p] = ox * a.stride;
        pixlin[p] = (b * a.H + oy * a.stride) * a.W + ox * a.stride;
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
    const int nsteps = a.ntaps * (Cin >> 5);
    const unsigned wbytes = (unsigned)nsteps * (unsigned)a.N * 64u;

    int ld_tap = 0, ld_cc = 0, ld_use1 = 0;
    unsigned ld_soffA = 0, ld_soffB = 0;
    const void* ld_src = a.x0;
    unsigned ld_bytes = bytes0;
    unsigned pixb[PA], vbase[PA], wvoff[PB];
#pragma unroll
    for (int p = 0; p < PB; ++p) wvoff[p] = (unsigned)(n_base * 64 + (wid * PB + p) * 1024 + lane * 16);
    auto source = [&]() {
        ld_use1 = ld_cc >= a.C0;
        const int cs = ld_use1 ? a.C1 : a.C0;
        ld_src = ld_use1 ? (const void*)a.x1 : (const void*)a.x0;
        ld_bytes = ld_use1 ? bytes1 : bytes0;
        ld_soffA = (unsigned)(ld_cc - (ld_use1 ? a.C0 : 0)) * 2u;
#pragma unroll
        for (int p = 0; p < PA; ++p) pixb[p] = (unsigned)pixlin[p] * (unsigned)cs * 2u + chunk16;
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
        ld_soffB += (unsigned)a.N * 64u;
        if (++ld_tap == a.ntaps) {
            ld_tap = 0;
            ld_cc += 32;
            ld_soffA += 64u;
            if (ld_cc == a.C0 && a.C1 > 0) source();
        }
        tap_rows();
    };
    auto issue = [&](int buf) {
        __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ld_src), 0, (int)ld_bytes, 0x00020000);
        __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, (int)wbytes, 0x00020000);
#pragma unroll
        for (int p = 0; p < PA; ++p)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, LWG_LDS_PTR(As + buf * A_STAGE + (wid * PA + p) * 1024), 16, (int)vbase[p],
                                                     (int)ld_soffA, 0, 0);
#pragma unroll
        for (int p = 0; p < PB; ++p)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, LWG_LDS_PTR(Bs + buf * B_STAGE + (wid * PB + p) * 1024), 16, (int)wvoff[p],
                                                     (int)ld_soffB, 0, 0);
    };

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int khalf = lane >> 5;
    const int sw = (lane >> 2) & 3;
    const char* fr_a = As + (wm * TM * 32 + (lane & 31)) * 64;
    const char* fr_b = Bs + (wn * TN * 32 + (lane & 31)) * 64;
    int koff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) koff[ks] = ((2 * ks + khalf) ^ sw) << 4;

    // prologue: the DMAs of steps 0, 1, 2 (the loader state always describes the next step to issue)
    source();
    tap_rows();
#pragma unroll
    for (int pre = 0; pre < NST - 1; ++pre) {
        if (pre < nsteps) {
            issue(pre);
            if (pre + 1 < nsteps) advance();
        }
    }
    for (int t = 0; t < nsteps; ++t) {
        // my pieces of stage t have landed when at most the pieces of the (up to two) younger issued steps are outstanding
        const int younger = nsteps - 1 - t;
        if (younger >= 2) LWG_WAIT_VM_LGKM0(8);
        else if (younger == 1) LWG_WAIT_VM_LGKM0(4);
        else LWG_WAIT_VM_LGKM0(0);
        // past the barrier: stage t is complete for every wave, and every wave has finished reading stage t-1 = (t+3) % 4
        if (t + NST - 1 < nsteps) {
            issue((t + NST - 1) & (NST - 1));
            if (t + NST < nsteps) advance();
        }
        const int cur = t & (NST - 1);
        bf16x8 fa[2][TM], fb[2][TN];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[ks][i] = *reinterpret_cast<const bf16x8*>(fr_a + cur * A_STAGE + i * 2048 + koff[ks]);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[ks][j] = *reinterpret_cast<const bf16x8*>(fr_b + cur * B_STAGE + j * 2048 + koff[ks]);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[ks][j], fa[ks][i], acc[i][j], 0, 0, 0);
    }
    lwg_bf16_epilogue<TM, TN, EPI>(a, acc, m_base, n_base, wm, wn, lane);
}

template <int WAVES_M, int WAVES_N, int TM, int TN, int EPI>
static hipError_t launch_cfg_bf16_4(const LwgConvArgs& a, hipStream_t stream) {
    constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
    constexpr size_t lds = (size_t)4 * (BM + BN) * 64 + 3 * LWG_MAX_TAPS * sizeof(int);
    auto kern = lwg_conv_bf16_kernel4<WAVES_M, WAVES_N, TM, TN, EPI>;
    static unsigned long long attr_done = 0ull;
    if (hipError_t e = lwg_allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds, attr_done); e != hipSuccess) return e;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = a.N / BN;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), lds, stream, a);
    return hipGetLastError();
}

template <int WAVES_M, int WAVES_N, int TM, int TN, int EPI, bool DMA_A>
static hipError_t launch_cfg_bf16(const LwgConvArgs& a, hipStream_t stream) {
    constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
    constexpr size_t lds = (size_t)2 * (BM + BN) * 128 + 3 * LWG_MAX_TAPS * sizeof(int);
    auto kern = lwg_conv_bf16_kernel<WAVES_M, WAVES_N, TM, TN, EPI, DMA_A>;
    static unsigned long long attr_done = 0ull;
    if (hipError_t e = lwg_allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds, attr_done); e != hipSuccess) return e;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = a.N / BN;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), lds, stream, a);
    return hipGetLastError();
}

template <int EPI>
static hipError_t launch_epi_bf16(const LwgConvArgs& a, hipStream_t stream) {
    static int dma_a = -1;
    if (dma_a < 0) {
        const char* ev = getenv("LWG_BF16_DMA_A");   // lab knob: 0 = A through registers (global -> VGPR -> ds_write), 1 = LDS-DMA
        dma_a = ev ? atoi(ev) : 1;
    }
    static int force64 = -1;
    if (force64 < 0) {
        const char* ev = getenv("LWG_BF16_TILE64");  // lab knob: 1 = 128 x 64 tiles (48 KB LDS: 3 workgroups per CU) wherever the epilogue allows
        force64 = ev ? atoi(ev) : 0;
    }
    if (EPI == LWG_EPI_SPADE || (a.N % 128 == 0 && !force64))
        return dma_a ? launch_cfg_bf16<2, 2, 2, 2, EPI, true>(a, stream) : launch_cfg_bf16<2, 2, 2, 2, EPI, false>(a, stream);
    return dma_a ? launch_cfg_bf16<4, 1, 1, 2, EPI, true>(a, stream) : launch_cfg_bf16<4, 1, 1, 2, EPI, false>(a, stream);
}

// args->w = the [ntaps*Cin/32][N][32] panel of the deep-pipeline variant (see lwg_conv_bf16_kernel4); N % 128 == 0 only.
extern "C" int lwg_conv2d_nhwc_bf16_p4(const LwgConvArgs* pa, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!pa) return (int)hipErrorInvalidValue;
    const LwgConvArgs& a = *pa;
    const int Cin = a.C0 + a.C1;
    if (!a.x0 || !a.w || !a.y || a.ntaps < 1 || a.ntaps > LWG_MAX_TAPS || a.M <= 0) return (int)hipErrorInvalidValue;
    if (a.xdt != LWG_DT_BF16 || a.ydt != LWG_DT_BF16) return (int)hipErrorInvalidValue;
    if (a.N % 128 != 0 || Cin % 32 != 0 || (a.YC & 7) != 0 || (a.ycoff & 7) != 0) return (int)hipErrorInvalidValue;
    if (a.C1 != 0 && (a.C0 % 32 != 0 || !a.x1)) return (int)hipErrorInvalidValue;
    const unsigned long long pix = (unsigned long long)a.B * a.H * a.W;
    if (pix * (unsigned long long)(a.C0 > a.C1 ? a.C0 : a.C1) * 2ull >= 0xC0000000ull) return (int)hipErrorInvalidValue;
    if ((unsigned long long)a.ntaps * Cin * (unsigned long long)a.N * 2ull >= 0xC0000000ull) return (int)hipErrorInvalidValue;
    if (a.epi == LWG_EPI_SPADE) {
        if (!a.xn || !a.mean || !a.rstd || !a.bias || a.YC * 2 != a.N || a.ycoff != 0) return (int)hipErrorInvalidValue;
        return (int)launch_cfg_bf16_4<2, 2, 2, 2, LWG_EPI_SPADE>(a, stream);
    }
    if (a.epi == LWG_EPI_RESIDUAL) {
        if (!a.res) return (int)hipErrorInvalidValue;
        return (int)launch_cfg_bf16_4<2, 2, 2, 2, LWG_EPI_RESIDUAL>(a, stream);
    }
    if (a.epi != LWG_EPI_NONE) return (int)hipErrorInvalidValue;
    return (int)launch_cfg_bf16_4<2, 2, 2, 2, LWG_EPI_NONE>(a, stream);
}

extern "C" int lwg_conv2d_nhwc_bf16(const LwgConvArgs* pa, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!pa) return (int)hipErrorInvalidValue;
    const LwgConvArgs& a = *pa;
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
