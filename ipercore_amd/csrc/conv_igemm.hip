// NHWC fp32 implicit-GEMM convolution on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32).
//
// Replaces every torch.nn.Conv2d / ConvTranspose2d call of the AttLWB-SPADE generator on the per-frame
// path (reference iPERCore/models/networks/generators/attlwb_spade_resunet.py:14-25 ResidualBlock,
// :73-93 SPADE convs, :202-204 fq, :268-271 encoder convs, :331-340 SkipDecoder, and bg_inpaintor.py).
//
// GEMM view:  D[M, N] = A[M, K] * Wp[K, N],  M = B*OH*OW output positions, N = Cout, K = ntaps*Cin.
//   * A is never materialised: every K-step (one tap, 32 input channels) gathers a BM x 32 slab straight
//     from the NHWC activation tensor (zero fill for padding), 128 contiguous bytes per pixel.
//   * Wp is the weight panel pre-packed on the host as [K/4][N][4] so a lane's 16-byte global load is
//     already the LDS image (k-quads are what one lane feeds to four consecutive MFMAs).
//   * a transposed conv (k4 s2 p1) is four such GEMMs (one per output parity) with 2x2 taps each.
// Workgroup = 4 waves; wave tile = (TM*32) x (TN*32) accumulators in VGPRs; LDS double buffered,
// one barrier per K-step; global->register prefetch of step t+1 is issued before the MFMAs of step t.
// Epilogues fuse bias, ReLU/tanh/sigmoid, the residual add, or SPADE's IN(x)*(1+gamma)+beta.
#include "lwg_common.h"
#include "lwg_conv_args.h"

template <int WAVES_M, int WAVES_N, int TM, int TN, int EPI, bool SMALLC>
__global__ __launch_bounds__(256, 2) void lwg_conv_igemm_kernel(const LwgConvArgs a) {
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

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WAVES_N, wn = wid % WAVES_N;
    const int tiles_n = a.N / BN;
    const int lid = lwg_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = lid % tiles_n, tile_m = lid / tiles_n;
    const int m_base = tile_m * BM, n_base = tile_n * BN;

    // ---- per-thread gather coordinates (fixed over the K loop) ----
    const int kq = tid & 7, mrow = tid >> 3;
    const int HW = a.OH * a.OW;
    int piy[PA], pix[PA], pbase[PA];
    bool pok[PA];
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        const int m = m_base + mrow + 32 * p;
        pok[p] = m < a.M;
        const int mm = pok[p] ? m : 0;
        const int b = mm / HW, rem = mm - b * HW;
        const int oy = rem / a.OW, ox = rem - oy * a.OW;
        piy[p] = oy * a.stride;
        pix[p] = ox * a.stride;
        pbase[p] = b * a.H * a.W;
    }
    const int Cin = a.C0 + a.C1;
    const int K4 = a.ntaps * (Cin >> 2);
    const int nsteps = (K4 + 7) >> 3;

    floatx4 ra[PA], rb[PB];

    auto gload = [&](int t) {
        // B panel: contiguous float4 per lane, always in range (host pads K to a multiple of 32)
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const int idx = tid + 256 * p;
            const int n = idx % BN, kqb = idx / BN;
            rb[p] = *reinterpret_cast<const floatx4*>(a.w + ((size_t)(t * 8 + kqb) * a.N + n_base + n) * 4);
        }
        int tap, c;
        if (SMALLC) {
            const int k4 = t * 8 + kq;
            tap = k4 >> a.cshift;
            c = (k4 & ((1 << a.cshift) - 1)) * 4;
        } else {
            const int cpt = Cin >> 5;  // 32-channel chunks per tap
            tap = t / cpt;
            c = (t - tap * cpt) * 32 + kq * 4;
        }
        const bool tap_ok = tap < a.ntaps;
        const int tt = tap_ok ? tap : 0;
        const int dy = a.dy[tt], dx = a.dx[tt];
        const float* src = a.x0;
        int cs = a.C0;
        if (c >= a.C0) { src = a.x1; c -= a.C0; cs = a.C1; }
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const int iy = piy[p] + dy, ix = pix[p] + dx;
            const bool ok = pok[p] && tap_ok && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            floatx4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *reinterpret_cast<const floatx4*>(src + (size_t)(pbase[p] + iy * a.W + ix) * cs + c);
            ra[p] = v;
        }
    };
    auto lstore = [&](int buf) {
        float* Ab = As + buf * A_STAGE;
        float* Bb = Bs + buf * B_STAGE;
#pragma unroll
        for (int p = 0; p < PA; ++p)
            *reinterpret_cast<floatx4*>(Ab + kq * A_ROW + (mrow + 32 * p) * 4) = ra[p];
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const int idx = tid + 256 * p;
            *reinterpret_cast<floatx4*>(Bb + idx * 4) = rb[p];  // [kq][n] is linear in idx
        }
    };

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    gload(0);
    lstore(0);
    __syncthreads();

    const int a_off = (wm * TM * 32 + (lane & 31)) * 4;
    const int b_off = (wn * TN * 32 + (lane & 31)) * 4;
    const int khalf = lane >> 5;

    for (int t = 0; t < nsteps; ++t) {
        const int cur = t & 1;
        if (t + 1 < nsteps) gload(t + 1);
        const float* Ab = As + cur * A_STAGE + a_off;
        const float* Bb = Bs + cur * B_STAGE + b_off;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int q = 2 * g + khalf;
            floatx4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const floatx4*>(Ab + q * A_ROW + i * 128);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const floatx4*>(Bb + q * B_ROW + j * 128);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < nsteps) lstore(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane owns column n = lane&31 of each 32x32 tile, rows (r&3) + 8*(r>>2) + 4*(lane>>5) ----
    const bool direct = (a.omul == 1) && (a.YH == a.OH) && (a.YW == a.OW);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m_base + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
            if (m >= a.M) continue;
            size_t opix;
            int bimg = 0;
            if (direct) {
                opix = (size_t)m;
                if (EPI == LWG_EPI_SPADE) bimg = m / HW;
            } else {
                const int b = m / HW, rem = m - b * HW;
                const int oy = rem / a.OW, ox = rem - oy * a.OW;
                opix = ((size_t)b * a.YH + (oy * a.omul + a.ooy)) * a.YW + (ox * a.omul + a.oox);
                bimg = b;
            }
            if (EPI == LWG_EPI_SPADE) {
                // wave columns [0,32) = gamma, [32,64) = beta of the same 32 channels (host packs them so)
                static_assert(EPI != LWG_EPI_SPADE || TN == 2, "SPADE epilogue needs gamma|beta in one wave");
                const int ch = ((n_base + wn * TN * 32) >> 1) + (lane & 31);
                const int ng = n_base + wn * TN * 32 + (lane & 31);
                const float g = acc[i][0][r] + a.bias[ng];
                const float bt = acc[i][TN - 1][r] + a.bias[ng + 32];
                const float xv = a.xn[opix * a.YC + ch];
                const float mu = a.mean[bimg * a.YC + ch], rs = a.rstd[bimg * a.YC + ch];
                a.y[opix * a.YC + ch] = lwg_act((xv - mu) * rs * (1.f + g) + bt, a.act);
            } else {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = n_base + wn * TN * 32 + j * 32 + (lane & 31);
                    float v = acc[i][j][r];
                    if (a.bias) v += a.bias[n];
                    if (EPI == LWG_EPI_RESIDUAL) v += a.res[opix * a.YC + a.ycoff + n];
                    a.y[opix * a.YC + a.ycoff + n] = lwg_act(v, a.act);
                }
            }
        }
    }
}

template <int WAVES_M, int WAVES_N, int TM, int TN, int EPI, bool SMALLC>
static hipError_t launch_cfg(const LwgConvArgs& a, hipStream_t stream) {
    constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
    constexpr size_t lds = (size_t)2 * 8 * ((BM + 1) * 4 + BN * 4) * sizeof(float);
    auto kern = lwg_conv_igemm_kernel<WAVES_M, WAVES_N, TM, TN, EPI, SMALLC>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = a.N / BN;
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), lds, stream, a);
    return hipGetLastError();
}

template <int EPI, bool SMALLC>
static hipError_t launch_epi(const LwgConvArgs& a, hipStream_t stream) {
    if (EPI == LWG_EPI_SPADE || a.N % 128 == 0) return launch_cfg<2, 2, 2, 2, EPI, SMALLC>(a, stream);  // 128 x 128
    return launch_cfg<4, 1, 1, 2, EPI, SMALLC>(a, stream);                                                // 128 x 64
}

// Host-side validation + dispatch; returns hipError_t as int (hipErrorInvalidValue for contract violations).
extern "C" int lwg_conv2d_nhwc_f32(const LwgConvArgs* pa, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!pa) return (int)hipErrorInvalidValue;
    const LwgConvArgs& a = *pa;
    const int Cin = a.C0 + a.C1;
    if (!a.x0 || !a.w || !a.y || a.ntaps < 1 || a.ntaps > LWG_MAX_TAPS || a.M <= 0) return (int)hipErrorInvalidValue;
    if (a.N % 64 != 0 || (Cin & 3) != 0) return (int)hipErrorInvalidValue;
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
        return (int)launch_epi<LWG_EPI_RESIDUAL, false>(a, stream);
    }
    if (a.epi != LWG_EPI_NONE) return (int)hipErrorInvalidValue;
    return smallc ? (int)launch_epi<LWG_EPI_NONE, true>(a, stream) : (int)launch_epi<LWG_EPI_NONE, false>(a, stream);
}
