// NHWC implicit-GEMM convolution, fp32 in / fp32 out, with every fp32 product formed on the bf16 matrix pipe from an
// EXACT three-way split of both operands ("bf16x6"): x = x_hi + x_mid + x_lo with each part a bf16 (8 significant bits,
// fp32's exponent range), computed by round-to-nearest residuals
//     hi = bf16(x),  mid = bf16(x - hi),  lo = bf16(x - hi - mid)            (x - hi - mid has <= 8 significant bits: exact)
// and   a*b  ~=  a_lo*b_hi + a_hi*b_lo + a_mid*b_mid + a_mid*b_hi + a_hi*b_mid + a_hi*b_hi       (6 MFMAs, fp32 accumulate).
// Every partial product of two bf16 is exact in the fp32 accumulator; the three dropped terms (mid*lo, lo*mid, lo*lo) are
// bounded by 2^-23 |a*b| and zero-mean, i.e. below the rounding the fp32 accumulation itself commits per term.  tests /
// tools/convlab.py --split measure it against an fp64 convolution next to the native fp32 MFMA kernel.
//
// Why: v_mfma_f32_32x32x16_bf16 retires 16x the flops per cycle of v_mfma_f32_32x32x2_f32, so six of them cost 0.375 of
// the native fp32 MFMA time for the same contraction (2.5 PFLOP/s / 6 = 417 "fp32-equivalent" TFLOP/s vs 157).
//
// Structure (same GEMM view, K order, gather machinery and D^T epilogue as csrc/conv_igemm.hip / conv_igemm_bf16.hip):
//   * 128x128 (or 128x64) tile, 4 waves, 64x64 (32x64) per wave, BK = 16 per stage, two LDS stages of
//     [plane 0..2][k-octet 0..1][row][8 bf16] for A and B: one ds_read_b128 = one MFMA operand;
//   * activations arrive as fp32 (raw buffer loads, hardware zero fill for padding), are split in registers
//     (v_cvt_pk_bf16_f32 + shifts + v_sub: 11 VALU per pair of elements) and stored as three bf16 planes;
//   * weights come pre-split from the host as three bf16 panels [3][K/8][N][8] (networks/packing via ops._w16x3);
//   * per stage and wave: 12 ds_read_b128 feed 24 MFMAs (768 matrix-pipe cycles) - the bound is the matrix pipe, then LDS;
//   * global loads run two stages ahead of the MFMAs (two register stages), two workgroups per CU.
// Measured (tools/convlab.py --split --ref64, profiles/r01_convlab_split.txt): 190-210 algorithmic TFLOP/s on the 512^2 layer
// shapes = 1.45-1.55x the native fp32 kernel, i.e. ~47 % of the bf16 pipe with six MFMAs per product; error against an fp64
// convolution BELOW the native fp32 MFMA kernel's (rms 7.4e-7 vs 8.5e-7 of the output rms on the 3x3 256->256 layer).  What
// is left is the barrier-per-stage structure (two co-resident workgroups fall into phase and idle together around the
// barrier).  Launches with N % 128 == 0 and >= 200 256x128 tiles use the 8-wave ping-pong kernel further down (+7-9 %).

#include "lwg_common.h"
#include "lwg_conv_args.h"
#include "lwg_conv_slices.h"
#include "lwg_conv_epilogue.h"

typedef __bf16 sbf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int uintx2 __attribute__((ext_vector_type(2)));

#define LWG_OOB_OFFSET 0xC0000000u
#ifndef LWG_SPLIT_SCHED
#define LWG_SPLIT_SCHED 1    // 0: compiler schedules freely; 1: loads pinned at the top of the stage; 2: + split/stores after the MFMAs
#endif
#ifndef LWG_SPLIT_ABL
#define LWG_SPLIT_ABL 0      // lab only (wrong results): 1 no global loads in the loop, 2 no split arithmetic, 4 no MFMAs, 8 no LDS stores
#endif
#ifndef LWG_SPLIT_PRIO
#define LWG_SPLIT_PRIO 0
#endif
#ifndef LWG_SPLIT_OCC
#define LWG_SPLIT_OCC 2
#endif

__device__ __forceinline__ floatx4 lwg_sbuf_load(const void* base, unsigned bytes, unsigned voff, unsigned soff) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
    return __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}

// two fp32 -> packed bf16 pair (round to nearest even), low half = a
__device__ __forceinline__ unsigned lwg_pk_bf16(float a, float b) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    bf16x2_t v;
    v[0] = (__bf16)a;
    v[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, v);
}

// exact 3-way split of a pair: returns the packed hi / mid / lo words
__device__ __forceinline__ void lwg_split_pair(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = lwg_pk_bf16(a, b);
    const float ra = a - __builtin_bit_cast(float, hi << 16);
    const float rb = b - __builtin_bit_cast(float, hi & 0xffff0000u);
    mid = lwg_pk_bf16(ra, rb);
    const float sa = ra - __builtin_bit_cast(float, mid << 16);
    const float sb = rb - __builtin_bit_cast(float, mid & 0xffff0000u);
    lo = lwg_pk_bf16(sa, sb);
}

template <int WAVES_M, int WAVES_N, int TM, int TN, int EPI>
__global__ __launch_bounds__(256, LWG_SPLIT_OCC) void lwg_conv_igemm_split_kernel(const LwgConvArgs a) {
    constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
    constexpr int A_ROW = (BM + 4) * 16;     // bytes per k-octet row; +4 slots: LDS stores bank on (addr/4) % 32, the two octets of a
                                             // 16-lane ds_write_b64 group land 16 banks apart (PMC: +8 gave 24 conflict cycles per wave-stage)
    constexpr int B_ROW = BN * 16;
    constexpr int A_PLANE = 2 * A_ROW, B_PLANE = 2 * B_ROW;        // BK = 16 = 2 octets
    constexpr int A_STAGE = 3 * A_PLANE, B_STAGE = 3 * B_PLANE;
    constexpr int PA = BM / 64;              // fp32 float4 loads per thread per stage (A: BM rows x 4 float4)
    constexpr int B_ITEMS = 6 * BN;          // 16-byte items per stage (3 planes x 2 octets x BN)
    constexpr int PB = (B_ITEMS + 255) / 256;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");

    extern __shared__ __attribute__((aligned(16))) char smem_s[];
    char* As = smem_s;
    char* Bs = smem_s + 2 * A_STAGE;
    int* taptab = reinterpret_cast<int*>(smem_s + 2 * A_STAGE + 2 * B_STAGE);  // [3][LWG_MAX_TAPS]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WAVES_N, wn = wid % WAVES_N;
    const int tiles_n = a.N / BN;
    const int lid = lwg_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = lid % tiles_n, tile_m = lid / tiles_n;
    const int m_base = tile_m * BM, n_base = tile_n * BN;

    const int kq = tid & 3, mrow = tid >> 2;
    const int HW = a.OH * a.OW;
    const int Cin = a.C0 + a.C1;
    int pixlin[PA];
    unsigned long long vmask[PA];
    {
        int piy[PA], pix[PA];
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const int m = m_base + mrow + 64 * p;
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
            const int dy = a.dy[tid], dx = a.dx[tid];
            taptab[tid] = (dy * a.W + dx) * a.C0 * 4;
            taptab[LWG_MAX_TAPS + tid] = (dy * a.W + dx) * a.C1 * 4;
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
    }

    const unsigned bytes0 = (unsigned)a.B * a.H * a.W * a.C0 * 4u;
    const unsigned bytes1 = (unsigned)a.B * a.H * a.W * a.C1 * 4u;
    const int nsteps = a.ntaps * (Cin >> 4);                      // 16 k per stage
    const unsigned plane_bytes = (unsigned)nsteps * 2u * a.N * 16u;
    const unsigned wbytes = 3u * plane_bytes;

    // loader state: K order = 32-channel chunk major, tap minor, the two 16-channel halves of a chunk innermost
    int ld_tap = 0, ld_cc = 0, ld_use1 = 0, ld_half = 0;
    unsigned ld_soffA = 0, ld_soffB = 0;
    const float* ld_src = a.x0;
    unsigned ld_bytes = bytes0;
    unsigned pixb[PA], vbase[PA], wvoff[PB];
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        const int idx = tid + 256 * p;
        const int plane = idx / (2 * BN), rem = idx - plane * 2 * BN;
        const int oct = rem / BN, n = rem - oct * BN;
        wvoff[p] = idx < B_ITEMS ? (unsigned)plane * plane_bytes + ((unsigned)oct * a.N + n_base + n) * 16u : LWG_OOB_OFFSET;
    }
    auto source = [&]() {
        ld_use1 = ld_cc >= a.C0;
        const int cs = ld_use1 ? a.C1 : a.C0;
        ld_src = ld_use1 ? a.x1 : a.x0;
        ld_bytes = ld_use1 ? bytes1 : bytes0;
        ld_soffA = (unsigned)(ld_cc - (ld_use1 ? a.C0 : 0)) * 4u;
#pragma unroll
        for (int p = 0; p < PA; ++p) pixb[p] = ((unsigned)pixlin[p] * (unsigned)cs + (unsigned)kq * 4u) * 4u;
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
        ld_soffB += (unsigned)a.N * 32u;       // 2 octets * N * 16 B per plane
        ld_half ^= 1;
        if (ld_half) {
            ld_soffA += 64u;
            return;
        }
        ld_soffA -= 64u;
        if (++ld_tap == a.ntaps) {
            ld_tap = 0;
            ld_cc += 32;
            ld_soffA += 128u;
            if (ld_cc == a.C0 && a.C1 > 0) source();
        }
        tap_rows();
    };

    // two register stages: the loads of stage t+2 are issued while stage t computes, and are split / stored into LDS during
    // stage t+1 (prefetch distance 2: a stage is only 24 MFMAs = 768 matrix-pipe cycles, less than one trip to L2 / HBM)
    floatx4 ra0[PA], rb0[PB], ra1[PA], rb1[PB];
    auto gload = [&](floatx4 (&ra)[PA], floatx4 (&rb)[PB]) {
#pragma unroll
        for (int p = 0; p < PA; ++p) ra[p] = lwg_sbuf_load(ld_src, ld_bytes, vbase[p], ld_soffA);
#pragma unroll
        for (int p = 0; p < PB; ++p) rb[p] = lwg_sbuf_load(a.w, wbytes, wvoff[p], ld_soffB);
    };
    // A: this lane's 4 channels are half of octet kq>>1 -> 8-byte stores at [plane][octet][row][(kq&1)*8]
    const int st_a = (kq >> 1) * A_ROW + mrow * 16 + (kq & 1) * 8;
    const int st_b = tid * 16;               // B items are already (plane, octet, n)-linear
    auto lstore = [&](int buf, const floatx4 (&ra)[PA], const floatx4 (&rb)[PB]) {
        char* Ab = As + buf * A_STAGE + st_a;
        char* Bb = Bs + buf * B_STAGE + st_b;
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            unsigned h0, m0, l0, h1, m1, l1;
            if (LWG_SPLIT_ABL & 2) {
                h0 = m0 = l0 = __builtin_bit_cast(unsigned, ra[p][0]);
                h1 = m1 = l1 = __builtin_bit_cast(unsigned, ra[p][2]);
            } else {
                lwg_split_pair(ra[p][0], ra[p][1], h0, m0, l0);
                lwg_split_pair(ra[p][2], ra[p][3], h1, m1, l1);
            }
            const uintx2 h = {h0, h1}, m = {m0, m1}, l = {l0, l1};
            *reinterpret_cast<uintx2*>(Ab + 64 * p * 16) = h;
            *reinterpret_cast<uintx2*>(Ab + 64 * p * 16 + A_PLANE) = m;
            *reinterpret_cast<uintx2*>(Ab + 64 * p * 16 + 2 * A_PLANE) = l;
        }
#pragma unroll
        for (int p = 0; p < PB; ++p)
            if (PB * 256 == B_ITEMS || tid + 256 * p < B_ITEMS) *reinterpret_cast<floatx4*>(Bb + 4096 * p) = rb[p];
    };

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int khalf = lane >> 5;
    const char* fr_a = As + khalf * A_ROW + (wm * TM * 32 + (lane & 31)) * 16;
    const char* fr_b = Bs + khalf * B_ROW + (wn * TN * 32 + (lane & 31)) * 16;

    sbf16x8 fa[3][TM], fb[3][TN];
    // fragment reads in the order the MFMAs consume them: hi planes first
    auto read_frags = [&](int cur) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[pl][i] = *reinterpret_cast<const sbf16x8*>(fr_a + cur * A_STAGE + pl * A_PLANE + i * 512);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[pl][j] = *reinterpret_cast<const sbf16x8*>(fr_b + cur * B_STAGE + pl * B_PLANE + j * 512);
        }
    };
    auto mfmas = [&]() {
        // (plane_a, plane_b): (hi,hi) (mid,hi) (hi,mid) (mid,mid) (lo,hi) (hi,lo): planes are consumed in read order
        constexpr int PAIRS[6][2] = {{0, 0}, {1, 0}, {0, 1}, {1, 1}, {2, 0}, {0, 2}};
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    if (LWG_SPLIT_ABL & 4) {
                        if (q == 0) acc[i][j][0] += (float)fb[0][j][0] + (float)fa[0][i][0] + (float)fb[1][j][0] + (float)fa[1][i][0] + (float)fb[2][j][0] + (float)fa[2][i][0];
                    } else {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[PAIRS[q][1]][j], fa[PAIRS[q][0]][i], acc[i][j], 0, 0, 0);
                    }
    };
    // one stage: issue the loads of stage t+2, read the fragments of stage t, then MFMAs with the split + LDS stores of stage
    // t+1 scheduled among them by the compiler
    auto stage = [&](int cur, bool load, floatx4 (&ral)[PA], floatx4 (&rbl)[PB], bool store, const floatx4 (&ras)[PA], const floatx4 (&rbs)[PB]) {
        if (load) {
            advance();
            if (!(LWG_SPLIT_ABL & 1)) gload(ral, rbl);
        }
        read_frags(cur);
        if (LWG_SPLIT_SCHED >= 1) __builtin_amdgcn_sched_barrier(0);
        if (LWG_SPLIT_PRIO) __builtin_amdgcn_s_setprio(1);
        mfmas();
        if (LWG_SPLIT_SCHED >= 2) __builtin_amdgcn_sched_barrier(0);
        if (store && !(LWG_SPLIT_ABL & 8)) lstore(cur ^ 1, ras, rbs);
        if (LWG_SPLIT_PRIO) __builtin_amdgcn_s_setprio(0);
        __syncthreads();
    };

    source();
    tap_rows();
    gload(ra0, rb0);
    advance();
    gload(ra1, rb1);
    lstore(0, ra0, rb0);
    __syncthreads();
    // nsteps is even (Cin % 32 == 0): stages come in (even, odd) pairs so the register stage indices are static
    for (int t = 0; t + 2 < nsteps; t += 2) {
        stage(0, true, ra0, rb0, true, ra1, rb1);
        stage(1, true, ra1, rb1, true, ra0, rb0);
    }
    stage(0, false, ra0, rb0, true, ra1, rb1);
    stage(1, false, ra1, rb1, false, ra0, rb0);
    lwg_conv_epilogue<TM, TN, EPI>(a, acc, m_base, n_base, wm, wn, lane);
}

template <int WAVES_M, int WAVES_N, int TM, int TN, int EPI>
static hipError_t launch_cfg_split(const LwgConvArgs& a, hipStream_t stream) {
    constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
    constexpr size_t lds = (size_t)2 * 3 * 2 * ((BM + 4) * 16 + BN * 16) + 3 * LWG_MAX_TAPS * sizeof(int);
    auto kern = lwg_conv_igemm_split_kernel<WAVES_M, WAVES_N, TM, TN, EPI>;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = a.N / BN;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), lds, stream, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// 8-wave "ping-pong" schedule of the same computation: one 512-thread workgroup per CU owns a 256x128 tile; its two wave
// groups (waves 0-3 / 4-7, one wave of each per SIMD) own the upper / lower 128 rows and ALTERNATE between an MFMA phase
// (24 back-to-back MFMAs on fragments already in registers, raised priority) and a staging phase (LDS stores of a stage
// loaded two phases earlier, global loads of a later stage, fragment reads for its next MFMA phase).  While one group's waves
// occupy the matrix pipes the other group's waves use the VALU / LDS / VMEM pipes of the same SIMDs; one s_barrier per phase
// keeps the groups exactly out of phase (two independent workgroups drift INTO phase and idle together, see the header).
// Stage bookkeeping (g = group 0, h = group 1; phase 2t: g computes stage t, phase 2t+1: h computes stage t):
//   g writes stage s (its A rows + its half of B) in phase 2s-3, h in phase 2s-2; g reads fragments of s at the end of phase
//   2s-1, h at the end of phase 2s; buffer s&1 is therefore overwritten (with s+2) only after both groups read s, and every
//   read of s comes at least one barrier after the last write of s.  Global loads are issued one staging phase (= two
//   phases) before their LDS store.
// Measured (lab build -DLWG_PP_TS, tools/ppts.py): MFMA phase 930 cycles (768 of pipe time), staging phase 790, barrier waits
// 130-190 per phase: 75 % of the matrix pipe in cycles; the shader clock drops to ~1.76 GHz under this kernel (2.24 GHz under the
// fp32 kernel), which is what separates it from the 2.4 GHz roofline.
#ifndef LWG_PP_ABL
#define LWG_PP_ABL 0      // lab only (wrong results): 1 no global loads, 2 no LDS stores, 4 no MFMAs, 8 no fragment reads
#endif
#ifndef LWG_PP_TS
#define LWG_PP_TS 0       // lab only: per-wave cycle totals of the MFMA phases, staging phases and barrier waits of workgroup 0
#endif
#if LWG_PP_TS
__device__ unsigned long long lwg_pp_ts[8 * 4];
extern "C" int lwg_lab_read_pp_ts(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(lwg_pp_ts), sizeof(lwg_pp_ts)); }
#define PP_LGKM0() __builtin_amdgcn_s_waitcnt(0xc07f);   /* lgkmcnt(0): count the LDS round trips into the staging phase */
#define PP_ACC(k) { const unsigned long long t1_ = __builtin_readcyclecounter(); ts_[k] += t1_ - tl_; tl_ = t1_; }
#else
#define PP_ACC(k)
#define PP_LGKM0()
#endif
#ifndef LWG_PP_SGB
#define LWG_PP_SGB 0
#endif
#ifndef LWG_SPLIT_PP_PRIO
#define LWG_SPLIT_PP_PRIO 1
#endif
template <int EPI>
__global__ __launch_bounds__(512, 1) void lwg_conv_igemm_split_pp_kernel(const LwgConvArgs a) {
    constexpr int TM = 2, TN = 2, BMG = 128, BN = 128;       // per wave group: 128 x 128, waves 2 x 2, 64 x 64 per wave
    constexpr int A_ROW = (BMG + 4) * 16, B_ROW = BN * 16;
    constexpr int A_PLANE = 2 * A_ROW, B_PLANE = 2 * B_ROW;
    constexpr int A_GRP = 3 * A_PLANE, A_STAGE = 2 * A_GRP, B_STAGE = 3 * B_PLANE;
    constexpr int PA = 2;                                     // float4 A loads per thread and stage
    constexpr int B_HALF = 3 * BN;                            // 16-byte B items per group and stage (768 / 2)
    constexpr int PB = 2;

    extern __shared__ __attribute__((aligned(16))) char smem_p[];
    char* As = smem_p;
    char* Bs = smem_p + 2 * A_STAGE;
    int* taptab = reinterpret_cast<int*>(smem_p + 2 * A_STAGE + 2 * B_STAGE);   // prologue only (validity masks)

    const int tid = threadIdx.x;
    const int grp = __builtin_amdgcn_readfirstlane(tid >> 8);
    const int gt = tid & 255, lane = tid & 63, wid = gt >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int tiles_n = a.N / BN;
    const int lid = lwg_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = lid % tiles_n, tile_m = lid / tiles_n;
    const int m_base = tile_m * 256 + grp * BMG, n_base = tile_n * BN;

    const int kq = gt & 3, mrow = gt >> 2;
    const int HW = a.OH * a.OW;
    const int Cin = a.C0 + a.C1;
    int pixlin[PA];
    unsigned long long vmask[PA];
    {
        int piy[PA], pix[PA];
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const int m = m_base + mrow + 64 * p;
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
            const int dy = a.dy[tid], dx = a.dx[tid];
            taptab[tid] = (dy * a.W + dx) * a.C0 * 4;
            taptab[LWG_MAX_TAPS + tid] = (dy * a.W + dx) * a.C1 * 4;
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
    }

    const unsigned bytes0 = (unsigned)a.B * a.H * a.W * a.C0 * 4u;
    const unsigned bytes1 = (unsigned)a.B * a.H * a.W * a.C1 * 4u;
    const int nsteps = a.ntaps * (Cin >> 4);
    const unsigned plane_bytes = (unsigned)nsteps * 2u * a.N * 16u;
    const unsigned wbytes = 3u * plane_bytes;

    // per-tap byte offsets live in two VGPRs (lane i = tap i) and are fetched with v_readlane: no LDS round trip in the loop
    const int tl = lane < a.ntaps ? lane : 0;
    const int vtab0 = (a.dy[tl] * a.W + a.dx[tl]) * a.C0 * 4, vtab1 = (a.dy[tl] * a.W + a.dx[tl]) * a.C1 * 4;
    int ld_tap = 0, ld_cc = 0, ld_use1 = 0, ld_half = 0;
    unsigned ld_soffA = 0, ld_soffB = 0;
    const float* ld_src = a.x0;
    unsigned ld_bytes = bytes0;
    unsigned pixb[PA], vbase[PA], wvoff[PB];
    int st_b[PB];
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        const int loc = gt + 256 * p;                         // this group's items [grp * 384, grp * 384 + 384)
        const int idx = grp * B_HALF + loc;
        const int plane = idx / (2 * BN), rem = idx - plane * 2 * BN;
        const int oct = rem / BN, n = rem - oct * BN;
        const bool ok = loc < B_HALF;
        wvoff[p] = ok ? (unsigned)plane * plane_bytes + ((unsigned)oct * a.N + n_base + n) * 16u : LWG_OOB_OFFSET;
        st_b[p] = ok ? idx * 16 : -1;
    }
    auto source = [&]() {
        ld_use1 = ld_cc >= a.C0;
        const int cs = ld_use1 ? a.C1 : a.C0;
        ld_src = ld_use1 ? a.x1 : a.x0;
        ld_bytes = ld_use1 ? bytes1 : bytes0;
        ld_soffA = (unsigned)(ld_cc - (ld_use1 ? a.C0 : 0)) * 4u;
#pragma unroll
        for (int p = 0; p < PA; ++p) pixb[p] = ((unsigned)pixlin[p] * (unsigned)cs + (unsigned)kq * 4u) * 4u;
    };
    auto tap_rows = [&]() {
        const int toff = __builtin_amdgcn_readlane(ld_use1 ? vtab1 : vtab0, ld_tap);
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const bool ok = (vmask[p] >> ld_tap) & 1ull;
            vbase[p] = ok ? pixb[p] + (unsigned)toff : LWG_OOB_OFFSET;
        }
    };
    auto advance = [&]() {
        ld_soffB += (unsigned)a.N * 32u;
        ld_half ^= 1;
        if (ld_half) {
            ld_soffA += 64u;
            return;
        }
        ld_soffA -= 64u;
        if (++ld_tap == a.ntaps) {
            ld_tap = 0;
            ld_cc += 32;
            ld_soffA += 128u;
            if (ld_cc == a.C0 && a.C1 > 0) source();
        }
        tap_rows();
    };

    floatx4 ra[PA], rb[PB];
    auto gload = [&]() {
#pragma unroll
        for (int p = 0; p < PA; ++p) ra[p] = lwg_sbuf_load(ld_src, ld_bytes, vbase[p], ld_soffA);
#pragma unroll
        for (int p = 0; p < PB; ++p) rb[p] = lwg_sbuf_load(a.w, wbytes, wvoff[p], ld_soffB);
    };
    const int st_a = grp * A_GRP + (kq >> 1) * A_ROW + mrow * 16 + (kq & 1) * 8;
    uintx2 pkh[PA], pkm[PA], pkl[PA];                          // the split of ra: packed bf16 pairs per plane
    auto split_a = [&]() {
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            unsigned h0, m0, l0, h1, m1, l1;
            lwg_split_pair(ra[p][0], ra[p][1], h0, m0, l0);
            lwg_split_pair(ra[p][2], ra[p][3], h1, m1, l1);
            pkh[p] = uintx2{h0, h1}; pkm[p] = uintx2{m0, m1}; pkl[p] = uintx2{l0, l1};
        }
    };
    auto lstore = [&](int buf) {
        char* Ab = As + buf * A_STAGE + st_a;
        char* Bb = Bs + buf * B_STAGE;
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            *reinterpret_cast<uintx2*>(Ab + 64 * p * 16) = pkh[p];
            *reinterpret_cast<uintx2*>(Ab + 64 * p * 16 + A_PLANE) = pkm[p];
            *reinterpret_cast<uintx2*>(Ab + 64 * p * 16 + 2 * A_PLANE) = pkl[p];
        }
#pragma unroll
        for (int p = 0; p < PB; ++p)
            if (st_b[p] >= 0) *reinterpret_cast<floatx4*>(Bb + st_b[p]) = rb[p];
    };

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int khalf = lane >> 5;
    const char* fr_a = As + grp * A_GRP + khalf * A_ROW + (wm * TM * 32 + (lane & 31)) * 16;
    const char* fr_b = Bs + khalf * B_ROW + (wn * TN * 32 + (lane & 31)) * 16;
    sbf16x8 fa[3][TM], fb[3][TN];
    auto read_frags = [&](int cur) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[pl][i] = *reinterpret_cast<const sbf16x8*>(fr_a + cur * A_STAGE + pl * A_PLANE + i * 512);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[pl][j] = *reinterpret_cast<const sbf16x8*>(fr_b + cur * B_STAGE + pl * B_PLANE + j * 512);
        }
    };
    // the MFMA phase also splits this wave's next A stage (loaded one staging phase earlier): ~45 VALU among 24 MFMAs - the
    // staging phase of the OTHER group then issues no VALU beside address arithmetic and does not compete for the VALU port
    auto mfmas_split = [&]() {
        constexpr int PAIRS[6][2] = {{0, 0}, {1, 0}, {0, 1}, {1, 1}, {2, 0}, {0, 2}};
        if (LWG_SPLIT_PP_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    if (!(LWG_PP_ABL & 4)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[PAIRS[q][1]][j], fa[PAIRS[q][0]][i], acc[i][j], 0, 0, 0);
        split_a();
#if LWG_PP_SGB
#pragma unroll
        for (int k = 0; k < 24; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);     // 2 VALU
        }
#endif
        if (LWG_SPLIT_PP_PRIO) __builtin_amdgcn_s_setprio(0);
    };

    // prologue: stage 0 by everyone; group 0 also stores stage 1 and holds stage 2's loads, group 1 holds stage 1 (A already split)
    source();
    tap_rows();
    gload();
    split_a();
    lstore(0);
    advance();
    gload();                                                  // stage 1 (nsteps is even, >= 2)
    split_a();
    if (grp == 0) {
        lstore(1);
        if (nsteps > 2) {
            advance();
            gload();                                          // stage 2: split during the first MFMA phase
        }
    }
    __syncthreads();
    if (grp == 0) read_frags(0);

    // Each group runs its own loop (two barriers per iteration in both, so the groups meet at every phase boundary): straight-line
    // bodies keep the loop-carried fragments / load registers in place (a per-phase branch on the group made the compiler copy
    // them at every merge, behind an s_waitcnt vmcnt(0) that exposed the whole global-load latency in every phase).
#if LWG_PP_TS
    unsigned long long ts_[4] = {0, 0, 0, 0};
    unsigned long long tl_ = __builtin_readcyclecounter();
#endif
    if (grp == 0) {
        int t = 0;
        for (; t + 3 < nsteps; ++t) {          // phase 2t: compute t, split t+2 | phase 2t+1: store t+2, load t+3, read fragments of t+1
            mfmas_split();
            PP_ACC(0)
            __syncthreads();
            PP_ACC(2)
            if (!(LWG_PP_ABL & 2)) lstore(t & 1);
            advance();
            if (!(LWG_PP_ABL & 1)) gload();
            if (!(LWG_PP_ABL & 8)) read_frags((t + 1) & 1);
            PP_LGKM0()
            PP_ACC(1)
            __syncthreads();
            PP_ACC(3)
        }
        for (; t < nsteps; ++t) {
            mfmas_split();
            __syncthreads();
            if (t + 2 < nsteps) lstore(t & 1);
            if (t + 1 < nsteps) read_frags((t + 1) & 1);
            __syncthreads();
        }
    } else {
        int t = 0;
        for (; t + 2 < nsteps; ++t) {          // phase 2t: store t+1, load t+2, read fragments of t | phase 2t+1: compute t, split t+2
            if (!(LWG_PP_ABL & 2)) lstore((t + 1) & 1);
            advance();
            if (!(LWG_PP_ABL & 1)) gload();
            if (!(LWG_PP_ABL & 8)) read_frags(t & 1);
            PP_LGKM0()
            PP_ACC(1)
            __syncthreads();
            PP_ACC(3)
            mfmas_split();
            PP_ACC(0)
            __syncthreads();
            PP_ACC(2)
        }
        for (; t < nsteps; ++t) {
            if (t + 1 < nsteps) lstore((t + 1) & 1);
            read_frags(t & 1);
            __syncthreads();
            mfmas_split();
            __syncthreads();
        }
    }
#if LWG_PP_TS
    if (blockIdx.x == 0 && lane == 0)
        for (int k = 0; k < 4; ++k) lwg_pp_ts[(tid >> 6) * 4 + k] = ts_[k];
#endif
    lwg_conv_epilogue<TM, TN, EPI>(a, acc, m_base, n_base, wm, wn, lane);
}

template <int EPI>
static hipError_t launch_split_pp(const LwgConvArgs& a, hipStream_t stream) {
    constexpr size_t lds = (size_t)2 * (2 * 3 * 2 * (128 + 4) * 16 + 3 * 2 * 128 * 16) + 3 * LWG_MAX_TAPS * sizeof(int);
    auto kern = lwg_conv_igemm_split_pp_kernel<EPI>;
    static unsigned long long attr_done = 0ull;
    if (hipError_t e = lwg_allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds, attr_done); e != hipSuccess) return e;
    const int tiles_m = (a.M + 255) / 256, tiles_n = a.N / 128;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(512), lds, stream, a);
    return hipGetLastError();
}

// the 256x128 ping-pong kernel needs N % 128 == 0 and enough tiles that one workgroup per CU keeps the chip busy
static bool lwg_split_use_pp(const LwgConvArgs& a) {
    constexpr int mode = LWG_SPLIT_PP;                        // compile-time (lwg_common.h): 0 never, 1 whenever legal, 2 heuristic
    if (mode == 0 || a.N % 128 != 0) return false;
    if (mode == 1) return true;
    const long tiles = (long)((a.M + 255) / 256) * (a.N / 128);
    const long waves = (tiles + 255) / 256;
    return tiles >= 200 && (double)tiles / (double)(waves * 256) >= 0.75;
}

template <int EPI>
static hipError_t launch_epi_split(const LwgConvArgs& a, hipStream_t stream) {
    if (lwg_split_use_pp(a)) return launch_split_pp<EPI>(a, stream);
    if (EPI == LWG_EPI_SPADE || a.N % 128 == 0) return launch_cfg_split<2, 2, 2, 2, EPI>(a, stream);
    return launch_cfg_split<4, 1, 1, 2, EPI>(a, stream);
}

// Same contract as lwg_conv2d_nhwc_f32 except: args->w is the three-plane bf16 panel [3][ntaps*Cin/8][N][8] (exact split of
// the fp32 weights, K order as the fp32 panel) and Cin % 32 == 0 is required (the small-Cin first layers stay on the fp32 kernel).
extern "C" int lwg_conv2d_nhwc_f32_split(const LwgConvArgs* pa, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!pa) return (int)hipErrorInvalidValue;
    const LwgConvArgs& a = *pa;
    {   // inputs beyond the 32-bit buffer-offset range: the same launch in batch slices (lwg_conv_slices.h)
        int sliced_err = 0;
        if (lwg_conv_run_sliced(a, [&](const LwgConvArgs& s) { return lwg_conv2d_nhwc_f32_split(&s, stream_); }, &sliced_err)) return sliced_err;
    }
    const int Cin = a.C0 + a.C1;
    if (!a.x0 || !a.w || !a.y || a.ntaps < 1 || a.ntaps > LWG_MAX_TAPS || a.M <= 0) return (int)hipErrorInvalidValue;
    if (a.N % 64 != 0 || Cin % 32 != 0 || (a.YC & 3) != 0 || (a.ycoff & 3) != 0) return (int)hipErrorInvalidValue;
    if (a.C1 != 0 && (a.C0 % 32 != 0 || !a.x1)) return (int)hipErrorInvalidValue;
    const unsigned long long pix = (unsigned long long)a.B * a.H * a.W;
    if (pix * (unsigned long long)(a.C0 > a.C1 ? a.C0 : a.C1) * 4ull >= 0xC0000000ull) return (int)hipErrorInvalidValue;
    if ((unsigned long long)a.ntaps * Cin * a.N * 6ull >= 0xC0000000ull) return (int)hipErrorInvalidValue;
    if (a.xdt != LWG_DT_F32 || a.ydt != LWG_DT_F32) return (int)hipErrorInvalidValue;
    if (a.epi == LWG_EPI_SPADE) {
        if (!a.xn || !a.mean || !a.rstd || !a.bias || a.N % 128 != 0 || a.YC * 2 != a.N) return (int)hipErrorInvalidValue;
        return (int)launch_epi_split<LWG_EPI_SPADE>(a, stream);
    }
    if (a.epi == LWG_EPI_RESIDUAL) {
        if (!a.res) return (int)hipErrorInvalidValue;
        return (int)launch_epi_split<LWG_EPI_RESIDUAL>(a, stream);
    }
    if (a.epi != LWG_EPI_NONE) return (int)hipErrorInvalidValue;
    return (int)launch_epi_split<LWG_EPI_NONE>(a, stream);
}
