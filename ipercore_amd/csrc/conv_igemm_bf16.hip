// NHWC implicit-GEMM convolution with bf16 MFMA operands (v_mfma_f32_32x32x16_bf16), fp32 activations in HBM and
// fp32 accumulation / epilogue - BASELINE configs[3] ("novel_view 1024x1024 bf16, MFMA bf16 conv tiles").
//
// Same GEMM view, K order, gather machinery (raw buffer loads with hardware zero fill, per-tap byte offsets, scalar
// channel-chunk offset) and D^T epilogue as csrc/conv_igemm.hip; what changes is the operand path:
//   * activations are read as fp32 (the layout every other kernel of the path produces/consumes) and rounded to bf16
//     (v_cvt_pk_bf16_f32, round-to-nearest-even) while they are staged into LDS as [k-octet][m][8 bf16]: one
//     ds_read_b128 is exactly one MFMA operand (8 consecutive k of one row);
//   * weights come pre-packed as bf16 panels [K/8][N][8] (networks/packing.py);
//   * at 16x the fp32 MFMA rate the matrix pipe is no longer the bound: per 32-k step a workgroup moves 16 KB of fp32
//     activations + 8 KB of weights from L2 for 8 MFMAs (256 cycles) per wave - the kernel is L2 / HBM bound, so the
//     structure is a plain double-buffered loop at 3 workgroups per CU (loads of step t+1 in flight during step t, other
//     workgroups' MFMAs covering this one's waits) instead of the fp32 kernel's instruction-level interleaving.
// Numerics: products of bf16-rounded operands accumulated in fp32; the generator's output stays within PSNR >= 40 dB of
// the fp32 path (tests/gpu_checks.py check_bf16_generator).
#include "lwg_common.h"
#include "lwg_conv_args.h"
#include "lwg_conv_epilogue.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define LWG_OOB_OFFSET 0xC0000000u

__device__ __forceinline__ floatx4 lwg_bbuf_load(const void* base, unsigned bytes, unsigned voff, unsigned soff) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
    return __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}

template <int WAVES_M, int WAVES_N, int TM, int TN, int EPI>
__global__ __launch_bounds__(256, 3) void lwg_conv_igemm_bf16_kernel(const LwgConvArgs a) {
    constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
    constexpr int A_ROW = (BM + 1) * 16;   // bytes per k-octet row of the A stage (+1 slot pad)
    constexpr int B_ROW = BN * 16;
    constexpr int A_STAGE = 4 * A_ROW, B_STAGE = 4 * B_ROW;   // BK = 32 = 4 octets
    constexpr int PA = BM / 32;            // fp32 float4 loads per thread per step (A)
    constexpr int PB = BN / 64;            // 16-byte loads per thread per step (B: 4 octets * BN * 16 B / 256 threads)
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");

    extern __shared__ __attribute__((aligned(16))) char smem_b[];
    char* As = smem_b;
    char* Bs = smem_b + 2 * A_STAGE;
    int* taptab = reinterpret_cast<int*>(smem_b + 2 * A_STAGE + 2 * B_STAGE);  // [3][LWG_MAX_TAPS]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WAVES_N, wn = wid % WAVES_N;
    const int tiles_n = a.N / BN;
    const int lid = lwg_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = lid % tiles_n, tile_m = lid / tiles_n;
    const int m_base = tile_m * BM, n_base = tile_n * BN;

    const int kq = tid & 7, mrow = tid >> 3;
    const int HW = a.OH * a.OW;
    const int Cin = a.C0 + a.C1;
    int pixlin[PA];
    unsigned long long vmask[PA];
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

    const unsigned bytes0 = (unsigned)a.B * a.H * a.W * a.C0 * 4u;
    const unsigned bytes1 = (unsigned)a.B * a.H * a.W * a.C1 * 4u;
    const int nsteps = a.ntaps * (Cin >> 5);
    const unsigned wbytes = (unsigned)nsteps * 4u * a.N * 16u;

    // loader state (same K order as the fp32 kernel: channel-chunk major, tap minor)
    int ld_tap = 0, ld_cc = 0, ld_use1 = 0;
    unsigned ld_soffA = 0, ld_soffB = 0;
    const float* ld_src = a.x0;
    unsigned ld_bytes = bytes0;
    unsigned pixb[PA], vbase[PA], wvoff[PB];
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        const int idx = tid + 256 * p;
        const int n = idx % BN, oct = idx / BN;
        wvoff[p] = ((unsigned)oct * a.N + n_base + n) * 16u;
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
        ld_soffB += (unsigned)a.N * 64u;       // 4 octets * N * 16 B
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
        for (int p = 0; p < PA; ++p) ra[p] = lwg_bbuf_load(ld_src, ld_bytes, vbase[p], ld_soffA);
#pragma unroll
        for (int p = 0; p < PB; ++p) rb[p] = lwg_bbuf_load(a.w, wbytes, wvoff[p], ld_soffB);
    };
    // A: this lane's 4 channels are half of octet kq>>1 -> 8-byte store at [octet][m][(kq&1)*4]
    const int st_a = (kq >> 1) * A_ROW + mrow * 16 + (kq & 1) * 8;
    const int st_b = tid * 16;
    auto lstore = [&](int buf) {
        char* Ab = As + buf * A_STAGE + st_a;
        char* Bb = Bs + buf * B_STAGE + st_b;
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            bf16x4 v;
            v[0] = (__bf16)ra[p][0]; v[1] = (__bf16)ra[p][1]; v[2] = (__bf16)ra[p][2]; v[3] = (__bf16)ra[p][3];
            *reinterpret_cast<bf16x4*>(Ab + 32 * p * 16) = v;
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) *reinterpret_cast<floatx4*>(Bb + 4096 * p) = rb[p];
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

    source();
    tap_rows();
    gload();
    lstore(0);
    __syncthreads();

    for (int t = 0; t < nsteps; ++t) {
        const int cur = t & 1;
        const bool next = t + 1 < nsteps;
        if (next) {
            advance();
            gload();
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(fr_a + cur * A_STAGE + 2 * s * A_ROW + i * 512);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(fr_b + cur * B_STAGE + 2 * s * B_ROW + j * 512);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        if (next) lstore(cur ^ 1);
        __syncthreads();
    }
    lwg_conv_epilogue<TM, TN, EPI>(a, acc, m_base, n_base, wm, wn, lane);
}

template <int WAVES_M, int WAVES_N, int TM, int TN, int EPI>
static hipError_t launch_cfg_bf16(const LwgConvArgs& a, hipStream_t stream) {
    constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
    constexpr size_t lds = (size_t)2 * 4 * ((BM + 1) * 16 + BN * 16) + 3 * LWG_MAX_TAPS * sizeof(int);
    auto kern = lwg_conv_igemm_bf16_kernel<WAVES_M, WAVES_N, TM, TN, EPI>;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = a.N / BN;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), lds, stream, a);
    return hipGetLastError();
}

template <int EPI>
static hipError_t launch_epi_bf16(const LwgConvArgs& a, hipStream_t stream) {
    if (EPI == LWG_EPI_SPADE || a.N % 128 == 0) return launch_cfg_bf16<2, 2, 2, 2, EPI>(a, stream);
    return launch_cfg_bf16<4, 1, 1, 2, EPI>(a, stream);
}

// Same contract as lwg_conv2d_nhwc_f32 except: args->w is the bf16 panel [ntaps*Cin/8][N][8] (K order as the fp32 panel) and
// Cin % 32 == 0 is required (the small-Cin first layers stay on the fp32 kernel).
extern "C" int lwg_conv2d_nhwc_bf16mma(const LwgConvArgs* pa, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!pa) return (int)hipErrorInvalidValue;
    const LwgConvArgs& a = *pa;
    const int Cin = a.C0 + a.C1;
    if (!a.x0 || !a.w || !a.y || a.ntaps < 1 || a.ntaps > LWG_MAX_TAPS || a.M <= 0) return (int)hipErrorInvalidValue;
    if (a.N % 64 != 0 || Cin % 32 != 0 || (a.YC & 3) != 0 || (a.ycoff & 3) != 0) return (int)hipErrorInvalidValue;
    if (a.C1 != 0 && (a.C0 % 32 != 0 || !a.x1)) return (int)hipErrorInvalidValue;
    const unsigned long long pix = (unsigned long long)a.B * a.H * a.W;
    if (pix * (unsigned long long)(a.C0 > a.C1 ? a.C0 : a.C1) * 4ull >= 0xC0000000ull) return (int)hipErrorInvalidValue;
    if (a.epi == LWG_EPI_SPADE) {
        if (!a.xn || !a.mean || !a.rstd || !a.bias || a.N % 128 != 0 || a.YC * 2 != a.N) return (int)hipErrorInvalidValue;
        return (int)launch_epi_bf16<LWG_EPI_SPADE>(a, stream);
    }
    if (a.epi == LWG_EPI_RESIDUAL) {
        if (!a.res) return (int)hipErrorInvalidValue;
        return (int)launch_epi_bf16<LWG_EPI_RESIDUAL>(a, stream);
    }
    if (a.epi != LWG_EPI_NONE) return (int)hipErrorInvalidValue;
    return (int)launch_epi_bf16<LWG_EPI_NONE>(a, stream);
}
