// NHWC fp32 implicit-GEMM convolution on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32).
//
// Replaces every torch.nn.Conv2d / ConvTranspose2d call of the AttLWB-SPADE generator on the per-frame
// path (reference iPERCore/models/networks/generators/attlwb_spade_resunet.py:14-25 ResidualBlock,
// :73-93 SPADE convs, :202-204 fq, :268-271 encoder convs, :331-340 SkipDecoder, and bg_inpaintor.py).
//
// GEMM view:  D[M, N] = A[M, K] * Wp[K, N],  M = B*OH*OW output positions, N = Cout, K = ntaps*Cin.
//   * A is never materialised: every K-step (one tap, 32 input channels) gathers a BM x 32 slab straight
//     from the NHWC activation tensor, 128 contiguous bytes per pixel.  Loads are raw buffer loads: a
//     padding pixel gets an out-of-range offset and the hardware returns zeros (no branches, no exec masks).
//   * Wp is the weight panel pre-packed on the host as [K/4][N][4] so a lane's 16-byte load is
//     already the LDS image (k-quads are what one lane feeds to four consecutive MFMAs).  K is ordered
//     channel-chunk major, tap minor (k = ((c/32)*ntaps + tap)*32 + c%32; plain tap*Cin + c when Cin < 32): the
//     taps of one 32-channel chunk are consecutive K-steps, so an activation chunk is fetched from HBM /
//     Infinity Cache once and its shifted re-reads hit the XCD's L2.
//   * a transposed conv (k4 s2 p1) is four such GEMMs (one per output parity) with 2x2 taps each.
//
// Pipeline (one wave must keep its SIMD's matrix pipe busy on its own - the two co-resident workgroups of a
// CU run in lock step, so "the other wave covers my bubble" does not happen):
//   K-step t = 4 phases of 16 MFMAs.  Fragments are double-buffered in registers: the ds_reads of phase g+1
//   are issued before the MFMAs of phase g.  The global loads of step t+1 are issued between the first MFMAs
//   of phase 0 with ZERO vector ALU work (per-pixel byte offsets are recomputed only when the tap or the
//   concat source changes; the channel chunk rides in the scalar offset), their ds_writes go to the other LDS
//   stage in the shadow of the first MFMAs of phase 3, then the single barrier of the step, then the fragment
//   reads of step t+1 phase 0 - issued BEFORE the last 4 MFMAs of step t so their latency is covered.  The K
//   loop is unrolled by two so every LDS address is a base register + immediate.
// The MFMAs compute D^T (weight fragment as the row operand): a lane then owns one output pixel and 4x4
// consecutive channels of each 32x32 tile, so the epilogue issues 16-byte NHWC stores and does the pixel
// index math once per row tile.
// Epilogues fuse bias, ReLU/tanh/sigmoid, the residual add, or SPADE's IN(x)*(1+gamma)+beta.

#include <type_traits>

#include "lwg_common.h"
#include "lwg_conv_args.h"
#include "lwg_conv_slices.h"
#include "lwg_conv_epilogue.h"

typedef int intx4 __attribute__((ext_vector_type(4)));

#define LWG_OOB_OFFSET 0xC0000000u  // >= any tensor's byte size (host enforces < 3 GiB): the buffer load returns 0
#define LWG_SB() __builtin_amdgcn_sched_barrier(0)
#ifndef LWG_CONV_DMA_B
#define LWG_CONV_DMA_B 0     // 1: the weight panel goes global -> LDS directly (buffer_load_dwordx4 ... lds), no VGPR round trip.
                             // Bit-identical results; measured A/B/A/B with tools/convlab.py: within 0.3 % of the register path (kept off)
#endif

__device__ __forceinline__ floatx4 lwg_buf_load(const float* base, unsigned bytes, unsigned voff, unsigned soff) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
    return __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}

// DEEP: the global loads run TWO K-steps ahead of the MFMAs (two register sets) instead of one.  A wave tile of one or two 32 x 32
// MFMA tiles spends 1024 / 2048 cycles per K-step; with the loads of step t + 1 issued at the top of step t and stored to LDS at its
// end they have ~700 cycles to return - less than an L2 round trip under load, and the small-tile launches (one frame, one training
// sample) have ONE workgroup per CU, nobody to cover the stall (measured: 57 us for a 4096 x 256 x 2304 launch whose MFMAs need 33).
// Same MFMA order, same results.
template <int WAVES_M, int WAVES_N, int TM, int TN, int EPI, bool SMALLC, bool DEEP = false>
__global__ __launch_bounds__(256, 2) void lwg_conv_igemm_kernel(const LwgConvArgs a, const int split_chunks, float* __restrict__ slabs,
                                                               const unsigned parity_panel_floats) {
    static_assert(!DEEP || !LWG_CONV_DMA_B, "the two-steps-ahead loader stages both operands through registers");
    // parity_panel_floats > 0: the FOUR output parities of a ConvTranspose2d(4, 2, 1) in one grid (lwg_conv_transpose4_nhwc_f32):
    // a describes parity (0, 0); workgroups with blockIdx.z = 2 py + px read the panel a.w + z * parity_panel_floats, shift every tap
    // by (py, px) (packing._CT_TAPS: parity p's input offsets are parity 0's + p) and write the pixels (2y + py, 2x + px)
    const int ppy = parity_panel_floats ? (int)(blockIdx.z >> 1) : 0, ppx = parity_panel_floats ? (int)(blockIdx.z & 1) : 0;
    const float* const wbase = a.w + (size_t)blockIdx.z * parity_panel_floats;
    constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
    constexpr int A_ROW = (BM + 1) * 4;  // floats per k-quad row; +1 float4 pad => conflict-free b128 stores
    constexpr int B_ROW = BN * 4;
    constexpr int A_STAGE = 8 * A_ROW, B_STAGE = 8 * B_ROW;
    constexpr int PA = BM / 32;  // A float4 loads per thread per K-step
    constexpr int PB = BN / 32;  // B float4 loads per thread per K-step
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + 2 * A_STAGE;
    int* taptab = reinterpret_cast<int*>(smem + 2 * A_STAGE + 2 * B_STAGE);  // [3][LWG_MAX_TAPS]: tap byte offset in x0, in x1, packed dy|dx

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WAVES_N, wn = wid % WAVES_N;
    const int tiles_n = a.N / BN;
    const int lid = lwg_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = lid % tiles_n, tile_m = lid / tiles_n;
    const int m_base = tile_m * BM, n_base = tile_n * BN;

    // ---- per-thread gather coordinates (fixed over the K loop) ----
    const int kq = tid & 7, mrow = tid >> 3;
    const int HW = a.OH * a.OW;
    const int Cin = a.C0 + a.C1;
    int pixlin[PA];              // linear pixel index (b*H + oy*stride)*W + ox*stride of GEMM row p
    unsigned long long vmask[PA];  // bit tap = the tap's sample position is inside the image (and the row is < M)
    int piy[PA], pix[PA];
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        const int m = m_base + mrow + 32 * p;
        const bool ok = m < a.M;
        const int mm = ok ? m : 0;
        const int b = mm / HW, rem = mm - b * HW;
        const int oy = rem / a.OW, ox = rem - oy * a.OW;
        piy[p] = ok ? oy * a.stride : -1000;
        pix[p] = ox * a.stride;
        pixlin[p] = (b * a.H + oy * a.stride) * a.W + ox * a.stride;
        vmask[p] = 0ull;
    }
    if (tid < a.ntaps) {
        const int dy = a.dy[tid] + ppy, dx = a.dx[tid] + ppx;
        taptab[tid] = (dy * a.W + dx) * a.C0 * 4;
        taptab[LWG_MAX_TAPS + tid] = (dy * a.W + dx) * a.C1 * 4;
        taptab[2 * LWG_MAX_TAPS + tid] = (dy & 0xffff) | (dx << 16);
    }
    __syncthreads();  // taptab visible
    if (!SMALLC) {
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
    }

    const unsigned bytes0 = (unsigned)a.B * a.H * a.W * a.C0 * 4u;
    const unsigned bytes1 = (unsigned)a.B * a.H * a.W * a.C1 * 4u;
    const int K4 = a.ntaps * (Cin >> 2);
    const int nsteps_all = (K4 + 7) >> 3;
    const unsigned wbytes = (unsigned)nsteps_all * 8u * a.N * 16u;
    // split-K (blockIdx.y = slice): this workgroup reduces the channel chunks [cc_first, cc_first + split_chunks) over all taps
    // and leaves a dense slab; lwg_splitk_finish_kernel adds the slices.  split_chunks == 0: the whole K range, normal epilogue.
    const int cc_first = (!SMALLC && split_chunks > 0) ? (int)blockIdx.y * split_chunks * 32 : 0;
    const int nsteps = (!SMALLC && split_chunks > 0) ? (min(Cin - cc_first, split_chunks * 32) >> 5) * a.ntaps : nsteps_all;

    // ---- loader state: describes the K-step whose loads are issued next.  K runs channel-chunk major, tap minor
    // (k = ((c / 32) * ntaps + tap) * 32 + c % 32): consecutive steps gather the SAME 32 channels at neighbouring
    // pixels, so the 9 (or 4, 49) shifted reads of an activation chunk are L2 hits instead of one trip to the
    // Infinity Cache / HBM per tap.
    int ld_tap = 0, ld_cc = cc_first;  // tap and channel offset (concat space) of that step
    int ld_use1 = 0;               // the chunk comes from x1 (skip concat)
    unsigned ld_soffA = 0;         // scalar byte offset of the channel chunk inside the current source
    unsigned ld_soffB = (unsigned)((cc_first >> 5) * a.ntaps) * (unsigned)a.N * 128u;   // scalar byte offset of the weight-panel step
    const float* ld_src = a.x0;
    unsigned ld_bytes = bytes0;
    unsigned pixb[PA];             // byte offset of (row p's pixel, channel quad kq) in the current source
    unsigned vbase[PA];            // pixb + tap offset, or LWG_OOB_OFFSET where the tap leaves the image
    unsigned wvoff[PB];
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        const int idx = tid + 256 * p;
        const int n = idx % BN, kqb = idx / BN;
        wvoff[p] = ((unsigned)kqb * a.N + n_base + n) * 16u;
    }

    auto source = [&]() {  // (re)bind the loader to the source tensor of channel ld_cc; runs at most twice per tile
        ld_use1 = ld_cc >= a.C0;
        const int cs = ld_use1 ? a.C1 : a.C0;
        ld_src = ld_use1 ? a.x1 : a.x0;
        ld_bytes = ld_use1 ? bytes1 : bytes0;
        ld_soffA = (unsigned)(ld_cc - (ld_use1 ? a.C0 : 0)) * 4u;
#pragma unroll
        for (int p = 0; p < PA; ++p) pixb[p] = ((unsigned)pixlin[p] * (unsigned)cs + (unsigned)kq * 4u) * 4u;
    };
    auto tap_row = [&](int p, int toff) {  // 5 VALU: issued one row at a time in the shadow of the MFMAs
        const bool ok = (vmask[p] >> ld_tap) & 1ull;
        vbase[p] = ok ? pixb[p] + (unsigned)toff : LWG_OOB_OFFSET;
    };
    auto advance_scalar = [&]() {  // move the loader state one K-step forward (uniform); returns the tap byte offset
        ld_soffB += (unsigned)a.N * 128u;
        if (!SMALLC) {
            if (++ld_tap == a.ntaps) {
                ld_tap = 0;
                ld_cc += 32;
                ld_soffA += 128u;
                if (ld_cc == a.C0 && a.C1 > 0) source();
            }
        }
        return SMALLC ? 0 : taptab[ld_use1 * LWG_MAX_TAPS + ld_tap];
    };

    constexpr int NSET = DEEP ? 2 : 1;
    floatx4 ra[NSET][PA], rb[NSET][PB];
    auto load_a = [&](int tstep, int set = 0) {
        if (!SMALLC) {
#pragma unroll
            for (int p = 0; p < PA; ++p) ra[set][p] = lwg_buf_load(ld_src, ld_bytes, vbase[p], ld_soffA);
        } else {
            const int k4 = tstep * 8 + kq;
            const int tap = k4 >> a.cshift;
            const int c = (k4 & ((1 << a.cshift) - 1)) * 4;
            const bool tap_ok = tap < a.ntaps;
            const int packed = taptab[2 * LWG_MAX_TAPS + (tap_ok ? tap : 0)];
            const int dy = (int)(short)(packed & 0xffff), dx = packed >> 16;
#pragma unroll
            for (int p = 0; p < PA; ++p) {
                const int iy = piy[p] + dy, ix = pix[p] + dx;
                const bool ok = tap_ok && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                const unsigned off = ((unsigned)(pixlin[p] + dy * a.W + dx) * (unsigned)Cin + (unsigned)c) * 4u;
                ra[set][p] = lwg_buf_load(a.x0, bytes0, ok ? off : LWG_OOB_OFFSET, 0u);
            }
        }
    };
    auto load_b = [&](int set = 0) {
#pragma unroll
        for (int p = 0; p < PB; ++p) rb[set][p] = lwg_buf_load(wbase, wbytes, wvoff[p], ld_soffB);
    };
    // LDS-DMA form: lane i of a wave lands at (M0 base) + 16 * i, i.e. the wave's 1 KB slice of the lane-linear B stage
    const int wave_b = __builtin_amdgcn_readfirstlane(tid >> 6) * 256;          // floats
    auto dma_b = [&](int buf) {
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wbase), 0, (int)wbytes, 0x00020000);
#pragma unroll
        for (int p = 0; p < PB; ++p)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(Bs + buf * B_STAGE + wave_b + 1024 * p), 16,
                                                     (int)wvoff[p], (int)ld_soffB, 0, 0);
    };

    const int st_a = kq * A_ROW + mrow * 4;  // + 128 * p
    const int st_b = tid * 4;                // + 1024 * p   ([kq][n] is linear in idx)
    auto lstore_a = [&](int buf, int set = 0) {
        float* Ab = As + buf * A_STAGE + st_a;
#pragma unroll
        for (int p = 0; p < PA; ++p) *reinterpret_cast<floatx4*>(Ab + 128 * p) = ra[set][p];
    };
    auto lstore_b = [&](int buf, int set = 0) {
        float* Bb = Bs + buf * B_STAGE + st_b;
#pragma unroll
        for (int p = 0; p < PB; ++p) *reinterpret_cast<floatx4*>(Bb + 1024 * p) = rb[set][p];
    };
    auto lstore = [&](int buf, int set = 0) {
        lstore_a(buf, set);
        lstore_b(buf, set);
    };

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int khalf = lane >> 5;
    const float* fr_a = As + (wm * TM * 32 + (lane & 31)) * 4 + khalf * A_ROW;
    const float* fr_b = Bs + (wn * TN * 32 + (lane & 31)) * 4 + khalf * B_ROW;
    floatx4 fa[2][TM], fb[2][TN];
    auto read_frags = [&](int buf, int g, int set) {  // all three are compile-time constants after unrolling
#pragma unroll
        for (int i = 0; i < TM; ++i)
            fa[set][i] = *reinterpret_cast<const floatx4*>(fr_a + buf * A_STAGE + 2 * g * A_ROW + i * 128);
#pragma unroll
        for (int j = 0; j < TN; ++j)
            fb[set][j] = *reinterpret_cast<const floatx4*>(fr_b + buf * B_STAGE + 2 * g * B_ROW + j * 128);
    };
    auto mfma_e = [&](int set, int e) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[set][j][e], fa[set][i][e], acc[i][j], 0, 0, 0);
    };

    // ---- prologue: stage 0 (DEEP: the loads of step 1 are in flight as well, the loader state describes step 2) ----
    if (!SMALLC) {
        source();
        const int toff0 = taptab[ld_use1 * LWG_MAX_TAPS];      // a split-K slice may start inside x1
#pragma unroll
        for (int p = 0; p < PA; ++p) tap_row(p, toff0);
    }
    load_a(0);
    if (LWG_CONV_DMA_B) dma_b(0); else load_b();
    {   // loader state -> step 1
        const int toff1 = advance_scalar();
        if (!SMALLC) {
#pragma unroll
            for (int p = 0; p < PA; ++p) tap_row(p, toff1);
        }
    }
    if (DEEP) {
        if (nsteps > 1) {
            load_a(1, NSET - 1);
            load_b(NSET - 1);
        }
        const int toff2 = advance_scalar();                    // loader state -> step 2
        if (!SMALLC) {
#pragma unroll
            for (int p = 0; p < PA; ++p) tap_row(p, toff2);
        }
    }
    if (LWG_CONV_DMA_B) {
        lstore_a(0);
        __builtin_amdgcn_s_waitcnt(0x0f70);        // vmcnt(0): the DMA of stage 0 has landed
    } else {
        lstore(0);
    }
    __syncthreads();
    read_frags(0, 0, 0);

    // NEXT: step t + 1 exists (its stage is stored + published here).  LOAD: the loads this step issues exist - step t + 1's
    // (= NEXT) in the one-step-ahead form, step t + 2's in the DEEP form (register set CUR: it held step t's operands, stored to LDS
    // one step ago; step t + 1's operands, loaded one step ago, sit in set CUR ^ 1 until this step's phase 3 stores them).
    auto step = [&](auto cur_c, auto next_c, auto load_c, int t) {
        constexpr int CUR = decltype(cur_c)::value;
        constexpr bool NEXT = decltype(next_c)::value;
        constexpr bool LOAD = decltype(load_c)::value;
        constexpr int LSET = DEEP ? CUR : 0;            // set the loads of this step go to
        constexpr int SSET = DEEP ? (CUR ^ 1) : 0;      // set phase 3 stores to LDS
        // phase 0: fragments of g=1 in flight, loads issued between the MFMAs
        read_frags(CUR, 1, 1);
        LWG_SB();
        mfma_e(0, 0);
        LWG_SB();
        if (LOAD) load_a(t + (DEEP ? 2 : 1), LSET);
        LWG_SB();
        mfma_e(0, 1);
        LWG_SB();
        if (LOAD) {
            if (LWG_CONV_DMA_B) dma_b(CUR ^ 1); else load_b(LSET);
        }
        LWG_SB();
        mfma_e(0, 2);
        mfma_e(0, 3);
        LWG_SB();
        // phase 1: the loader state moves on by one step (the offsets of its tap: one row per MFMA group)
        read_frags(CUR, 2, 0);
        LWG_SB();
        int toff = 0;
        if (LOAD) toff = advance_scalar();
        LWG_SB();
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            mfma_e(1, e);
            LWG_SB();
            if (LOAD && !SMALLC) {
#pragma unroll
                for (int p = e * PA / 4; p < (e + 1) * PA / 4; ++p) tap_row(p, toff);
            }
            LWG_SB();
        }
        // phase 2
        read_frags(CUR, 3, 1);
        LWG_SB();
        mfma_e(0, 0);
        mfma_e(0, 1);
        mfma_e(0, 2);
        mfma_e(0, 3);
        LWG_SB();
        // phase 3: the stores of step t+1 ride in the shadow of MFMAs 1..8, the barrier sits before the last 4
        mfma_e(1, 0);
        LWG_SB();
        if (NEXT) lstore_a(CUR ^ 1, SSET);
        LWG_SB();
        mfma_e(1, 1);
        LWG_SB();
        if (NEXT && !LWG_CONV_DMA_B) lstore_b(CUR ^ 1, SSET);
        LWG_SB();
        mfma_e(1, 2);
        LWG_SB();
        if (NEXT) {
            if (LWG_CONV_DMA_B) __builtin_amdgcn_s_waitcnt(0x0f70);     // vmcnt(0) before the barrier that publishes the stage
            __syncthreads();
            read_frags(CUR ^ 1, 0, 0);
        }
        LWG_SB();
        mfma_e(1, 3);
        LWG_SB();
    };
    using c0 = std::integral_constant<int, 0>;
    using c1 = std::integral_constant<int, 1>;
    using T_ = std::true_type;
    using F_ = std::false_type;
    int t = 0;
    if (DEEP) {
        for (; t + 3 < nsteps; t += 2) {                   // steps t + 2 and t + 3 exist: both steps load
            step(c0{}, T_{}, T_{}, t);
            step(c1{}, T_{}, T_{}, t + 1);
        }
        const int left = nsteps - t;                       // 1, 2 or 3
        if (left == 3) {
            step(c0{}, T_{}, T_{}, t);
            step(c1{}, T_{}, F_{}, t + 1);
            step(c0{}, F_{}, F_{}, t + 2);
        } else if (left == 2) {
            step(c0{}, T_{}, F_{}, t);
            step(c1{}, F_{}, F_{}, t + 1);
        } else {
            step(c0{}, F_{}, F_{}, t);
        }
    } else {
        for (; t + 2 < nsteps; t += 2) {
            step(c0{}, T_{}, T_{}, t);
            step(c1{}, T_{}, T_{}, t + 1);
        }
        if (nsteps - t == 2) {
            step(c0{}, T_{}, T_{}, t);
            step(c1{}, F_{}, F_{}, t + 1);
        } else {
            step(c0{}, F_{}, F_{}, t);
        }
    }

    if (EPI == LWG_EPI_NONE && !SMALLC && split_chunks > 0) {
        lwg_conv_epilogue_slab<TM, TN>(slabs + (size_t)blockIdx.y * a.M * a.N, a.M, a.N, acc, m_base, n_base, wm, wn, lane);
        return;
    }
    lwg_conv_epilogue<TM, TN, EPI>(a, acc, m_base, n_base, wm, wn, lane, ppy, ppx);
}

// y[row m -> output pixel][ycoff + n] = act(sum_s slab[s][m][n] + bias[n]): the epilogue of a split-K launch (slices added in
// slice order: deterministic).
__global__ __launch_bounds__(256) void lwg_splitk_finish_kernel(const LwgConvArgs a, const float* __restrict__ slabs, int nslices) {
    const int N4 = a.N >> 2;
    const size_t total = (size_t)a.M * N4, MN = (size_t)a.M * a.N;
    const int HW = a.OH * a.OW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / N4), n = (int)(i - (size_t)m * N4) * 4;
        floatx4 s = *reinterpret_cast<const floatx4*>(slabs + (size_t)m * a.N + n);
        for (int k = 1; k < nslices; ++k) {
            const floatx4 v = *reinterpret_cast<const floatx4*>(slabs + k * MN + (size_t)m * a.N + n);
#pragma unroll
            for (int c = 0; c < 4; ++c) s[c] += v[c];
        }
        if (a.bias) {
            const floatx4 b4 = *reinterpret_cast<const floatx4*>(a.bias + n);
#pragma unroll
            for (int c = 0; c < 4; ++c) s[c] += b4[c];
        }
        const int b = m / HW, rem = m - b * HW;
        const int oy = rem / a.OW, ox = rem - oy * a.OW;
        const size_t opix = ((size_t)b * a.YH + (oy * a.omul + a.ooy)) * a.YW + (ox * a.omul + a.oox);
        if (a.act == LWG_ACT_RELU_MASK) {            // ReLU backward of the producer of the forward input (see lwg_common.h)
            const floatx4 rv = *reinterpret_cast<const floatx4*>(a.res + opix * a.YC + a.ycoff + n);
#pragma unroll
            for (int c = 0; c < 4; ++c) s[c] = rv[c] > 0.f ? s[c] : 0.f;
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) s[c] = lwg_act(s[c], a.act);
        }
        *reinterpret_cast<floatx4*>(a.y + opix * a.YC + a.ycoff + n) = s;
    }
}

// (also the finish of the Winograd kernel's split launches: conv_winograd.hip)
hipError_t lwg_splitk_finish_launch(const LwgConvArgs& a, const float* ws, int slices, hipStream_t stream) {
    const size_t total4 = (size_t)a.M * (a.N / 4);
    const int blocks = (int)((total4 + 255) / 256 < 2048 ? (total4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(lwg_splitk_finish_kernel, dim3(blocks), dim3(256), 0, stream, a, ws, slices);
    return hipGetLastError();
}

// Split-K plan of a launch (0 slices = run it whole).  Only the 64x64-tile regime (one training sample, the discriminator's
// deep layers: M of a few thousand rows against K of a few thousand) is split: the tile grid alone leaves most CUs with one
// workgroup or none, while 3-4 fit (33 KB LDS, ~80 VGPRs).  Slices are whole 32-channel chunks (all taps of a chunk stay
// together, so the L2-friendly tap-minor K order is kept) with at least 8 K-steps each.
static int lwg_conv_split_plan(const LwgConvArgs& a, int* chunks_per_slice) {
    const int Cin = a.C0 + a.C1;
    const bool mask = a.epi == LWG_EPI_RESIDUAL && a.act == LWG_ACT_RELU_MASK;     // finished by lwg_splitk_finish_kernel like LWG_EPI_NONE
    if (!LWG_CONV_SPLITK || (a.epi != LWG_EPI_NONE && !mask) || (Cin % 32) != 0) return 0;
    const long tiles128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
    if (tiles128 >= 300) return 0;                              // not the 64x64 regime (launch_epi)
    const long tiles = (long)((a.M + 63) / 64) * (a.N / 64);
    const int chunks = Cin / 32;
    if (tiles >= LWG_CONV_SPLIT_MAX_TILES || chunks < 2) return 0;   // (512-tile launches split in two measured slower: 35.0 vs 34.0 ms / step in round 1, 24.4 vs 24.1 in round 3)
    int want = (int)((1024 + tiles - 1) / tiles);               // aim at ~1024 workgroups = 4 per CU
    if (want > 8) want = 8;
    int cps = (chunks + want - 1) / want;
    while (cps * a.ntaps < 8 && cps < chunks) ++cps;
    const int slices = (chunks + cps - 1) / cps;
    if (slices < 2) return 0;
    *chunks_per_slice = cps;
    return slices;
}

extern "C" size_t lwg_conv2d_ws_floats(const LwgConvArgs* pa) {
    if (!pa || pa->M <= 0 || pa->N <= 0) return 0;
    int cps = 0;
    return (size_t)lwg_conv_split_plan(*pa, &cps) * (size_t)pa->M * (size_t)pa->N;
}

template <int WAVES_M, int WAVES_N, int TM, int TN, int EPI, bool SMALLC, bool DEEP = false>
static hipError_t launch_cfg(const LwgConvArgs& a, hipStream_t stream, float* ws = nullptr) {
    constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
    constexpr size_t lds = (size_t)2 * 8 * ((BM + 1) * 4 + BN * 4) * sizeof(float) + 3 * LWG_MAX_TAPS * sizeof(int);
    auto kern = lwg_conv_igemm_kernel<WAVES_M, WAVES_N, TM, TN, EPI, SMALLC, DEEP>;
    static unsigned long long attr_done = 0ull;
    if (hipError_t e = lwg_allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds, attr_done); e != hipSuccess) return e;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = a.N / BN;
    if constexpr (EPI == LWG_EPI_NONE && !SMALLC && BM == 64 && BN == 64) {
        int cps = 0;
        const int slices = ws ? lwg_conv_split_plan(a, &cps) : 0;
        if (slices > 1) {
            hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n, slices), dim3(256), lds, stream, a, cps, ws, 0u);
            return lwg_splitk_finish_launch(a, ws, slices, stream);
        }
    }
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), lds, stream, a, 0, (float*)nullptr, 0u);
    return hipGetLastError();
}

// The four parity launches of a transposed convolution as ONE grid (blockIdx.z = parity) on the small-launch tile configuration.
template <int EPI>
static hipError_t launch_parity4(const LwgConvArgs& a, hipStream_t stream, unsigned panel_floats) {
    constexpr size_t lds = (size_t)2 * 8 * ((64 + 1) * 4 + 64 * 4) * sizeof(float) + 3 * LWG_MAX_TAPS * sizeof(int);
    auto kern = lwg_conv_igemm_kernel<2, 2, 1, 1, EPI, false, LWG_CONV_DEEP != 0>;
    static unsigned long long attr_done = 0ull;
    if (hipError_t e = lwg_allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds, attr_done); e != hipSuccess) return e;
    const int tiles_m = (a.M + 63) / 64, tiles_n = a.N / 64;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n, 1, 4), dim3(256), lds, stream, a, 0, (float*)nullptr, panel_floats);
    return hipGetLastError();
}

template <int EPI, bool SMALLC>
static hipError_t launch_epi(const LwgConvArgs& a, hipStream_t stream, float* ws = nullptr) {
    // small launches (one training sample, short clips): 128x128 tiles would leave most of the 256 CUs idle - 64x64 tiles
    // quadruple the workgroup count (each wave then owns one 32x32 MFMA tile: fewer flops per staged byte, but it runs)
    const long tiles128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
    if constexpr (EPI != LWG_EPI_SPADE && !SMALLC) {
        if (tiles128 < (long)LWG_CONV_SMALL_TILES) return launch_cfg<2, 2, 1, 1, EPI, SMALLC, LWG_CONV_DEEP != 0>(a, stream, ws);  // 64 x 64 (split-K when ws is given)
    }
    if constexpr (EPI == LWG_EPI_SPADE && !SMALLC) {
        // SPADE needs gamma | beta of the same channels in one wave (TN = 2): a small launch takes 128 x 64 tiles - four waves stacked
        // in M, each 32 rows x (32 gamma | 32 beta) - instead of 128 x 128: twice the workgroups (a 4096-row frame at N = 512: 256
        // instead of 128, one per CU)
        if (LWG_CONV_SPADE_SMALL && tiles128 < (long)LWG_CONV_SMALL_TILES) return launch_cfg<4, 1, 1, 2, EPI, SMALLC, LWG_CONV_DEEP != 0>(a, stream);
    }
    if (EPI == LWG_EPI_SPADE || a.N % 128 == 0) return launch_cfg<2, 2, 2, 2, EPI, SMALLC>(a, stream);  // 128 x 128
    return launch_cfg<4, 1, 1, 2, EPI, SMALLC>(a, stream);                                                // 128 x 64
}

// Host-side validation + dispatch; returns hipError_t as int (hipErrorInvalidValue for contract violations).
extern "C" int lwg_conv2d_nhwc_f32_ws(const LwgConvArgs* pa, float* ws, lwg_stream_t stream_);
extern "C" int lwg_conv2d_nhwc_f32(const LwgConvArgs* pa, lwg_stream_t stream_) { return lwg_conv2d_nhwc_f32_ws(pa, nullptr, stream_); }

// ws: NULL, or lwg_conv2d_ws_floats(args) floats - small-M / large-K launches then run split-K through it.
extern "C" int lwg_conv2d_nhwc_f32_ws(const LwgConvArgs* pa, float* ws, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!pa) return (int)hipErrorInvalidValue;
    const LwgConvArgs& a = *pa;
    {   // inputs beyond the 32-bit buffer-offset range: the same launch in batch slices (lwg_conv_slices.h)
        int sliced_err = 0;
        if (lwg_conv_run_sliced(a, [&](const LwgConvArgs& s) { return ws ? 1 : lwg_conv2d_nhwc_f32_ws(&s, nullptr, stream_); }, &sliced_err)) return sliced_err;
    }
    const int Cin = a.C0 + a.C1;
    if (!a.x0 || !a.w || !a.y || a.ntaps < 1 || a.ntaps > LWG_MAX_TAPS || a.M <= 0) return (int)hipErrorInvalidValue;
    if (a.N % 64 != 0 || (Cin & 3) != 0 || (a.YC & 3) != 0 || (a.ycoff & 3) != 0) return (int)hipErrorInvalidValue;
    if (a.xdt != LWG_DT_F32 || (a.ydt != LWG_DT_F32 && ((a.ydt != LWG_DT_BF16 && a.ydt != LWG_DT_F32_Q4) || a.epi != LWG_EPI_NONE || ws))) return (int)hipErrorInvalidValue;
    if (a.ydt == LWG_DT_F32_Q4 && a.act == LWG_ACT_RELU_MASK) return (int)hipErrorInvalidValue;
    // buffer-load addressing: every tensor the kernel gathers from must be smaller than LWG_OOB_OFFSET bytes
    const unsigned long long pix = (unsigned long long)a.B * a.H * a.W;
    if (pix * (unsigned long long)(a.C0 > a.C1 ? a.C0 : a.C1) * 4ull >= 0xC0000000ull) return (int)hipErrorInvalidValue;
    const bool smallc = (Cin % 32) != 0;
    if (smallc) {
        if (a.C1 != 0 || Cin > 16 || (Cin & (Cin - 1)) != 0 || (1 << a.cshift) != (Cin >> 2)) return (int)hipErrorInvalidValue;
    } else if (a.C1 != 0 && (a.C0 % 32 != 0 || !a.x1)) {
        return (int)hipErrorInvalidValue;
    }
    if (a.epi == LWG_EPI_SPADE) {
        if (!a.xn || !a.mean || !a.rstd || !a.bias || a.N % 128 != 0 || smallc || a.YC * 2 != a.N) return (int)hipErrorInvalidValue;
        return (int)launch_epi<LWG_EPI_SPADE, false>(a, stream);
    }
    if (a.epi == LWG_EPI_RESIDUAL) {
        if (!a.res || smallc) return (int)hipErrorInvalidValue;
        if (a.act == LWG_ACT_RELU_MASK && ws) {
            int cps = 0;
            if (lwg_conv_split_plan(a, &cps) > 1) return (int)launch_epi<LWG_EPI_NONE, false>(a, stream, ws);   // slabs + the masking finish kernel
        }
        return (int)launch_epi<LWG_EPI_RESIDUAL, false>(a, stream);
    }
    if (a.act == LWG_ACT_RELU_MASK) return (int)hipErrorInvalidValue;
    if (a.epi != LWG_EPI_NONE) return (int)hipErrorInvalidValue;
    return smallc ? (int)launch_epi<LWG_EPI_NONE, true>(a, stream) : (int)launch_epi<LWG_EPI_NONE, false>(a, stream, ws);
}

// nn.ConvTranspose2d(kernel 4, stride 2, padding 1) on fp32 NHWC as ONE call (decoder up-sampling, attlwb_spade_resunet.py:331-340;
// bg_inpaintor.py:49-50).  args: the launch description of the parity-(0, 0) launch (ntaps = 4 with dy, dx in {-1, 0}, stride = 1,
// omul = 2, ooy = oox = 0, OH = H, OW = W, YH = 2H, YW = 2W, LWG_EPI_NONE); args->w = the four parity panels stacked [2 py + px][4 Cin][N]
// (each as lwg_conv2d_nhwc_f32 takes it); parity (py, px) reads the taps shifted by (py, px) and writes the pixels (2y + py, 2x + px).
// Small launches (one frame: a parity is 64..256 tiles of 64 x 64, a workgroup per CU or less) run as ONE grid of four times the
// workgroups; large ones as the four launches of lwg_conv2d_nhwc_f32.  Either way every output element is computed exactly as by four
// separate lwg_conv2d_nhwc_f32 calls (same tile, same K order).
// 1 if lwg_conv_transpose4_nhwc_f32 runs this description as ONE grid (a parity has fewer 128 x 128 tiles than the small-launch
// threshold), 0 if as four launches - so that a caller bracketing launches with events (bench.py) brackets what really runs.
extern "C" int lwg_conv_transpose4_is_one_grid(const LwgConvArgs* pa) {
    if (!pa || pa->M <= 0 || pa->N <= 0) return 0;
    const long tiles128 = (long)((pa->M + 127) / 128) * ((pa->N + 127) / 128);
    return tiles128 < (long)LWG_CONV_SMALL_TILES ? 1 : 0;
}

extern "C" int lwg_conv_transpose4_nhwc_f32(const LwgConvArgs* pa, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!pa) return (int)hipErrorInvalidValue;
    const LwgConvArgs& a = *pa;
    {   // inputs beyond the 32-bit buffer-offset range: the same launch in batch slices (lwg_conv_slices.h)
        int sliced_err = 0;
        if (lwg_conv_run_sliced(a, [&](const LwgConvArgs& s) { return lwg_conv_transpose4_nhwc_f32(&s, stream_); }, &sliced_err)) return sliced_err;
    }
    if (!a.x0 || !a.w || !a.y || a.M <= 0 || a.ntaps != 4 || a.stride != 1 || a.omul != 2 || a.ooy != 0 || a.oox != 0 || a.C1 != 0 ||
        a.epi != LWG_EPI_NONE || a.xdt != LWG_DT_F32 || (a.ydt != LWG_DT_F32 && a.ydt != LWG_DT_F32_Q4) || a.C0 % 32 != 0 || a.N % 64 != 0 || (a.YC & 3) != 0 ||
        (a.ycoff & 3) != 0 || a.OH != a.H || a.OW != a.W || a.YH != 2 * a.H || a.YW != 2 * a.W)
        return (int)hipErrorInvalidValue;
    for (int t = 0; t < 4; ++t)
        if (a.dy[t] < -1 || a.dy[t] > 0 || a.dx[t] < -1 || a.dx[t] > 0) return (int)hipErrorInvalidValue;
    if ((unsigned long long)a.B * a.H * a.W * (unsigned long long)a.C0 * 4ull >= 0xC0000000ull) return (int)hipErrorInvalidValue;
    const unsigned panel_floats = 4u * (unsigned)a.C0 * (unsigned)a.N;
    if (lwg_conv_transpose4_is_one_grid(pa)) return (int)launch_parity4<LWG_EPI_NONE>(a, stream, panel_floats);
    for (int p = 0; p < 4; ++p) {
        LwgConvArgs ap = a;
        ap.w = a.w + (size_t)p * panel_floats;
        ap.ooy = p >> 1;
        ap.oox = p & 1;
        for (int t = 0; t < 4; ++t) {
            ap.dy[t] = (signed char)(a.dy[t] + (p >> 1));
            ap.dx[t] = (signed char)(a.dx[t] + (p & 1));
        }
        const int e = lwg_conv2d_nhwc_f32_ws(&ap, nullptr, stream_);
        if (e != 0) return e;
    }
    return 0;
}
