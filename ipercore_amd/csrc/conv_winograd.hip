// Fused F(2x2, 3x3) Winograd form of the 3x3 / stride 1 / pad 1 fp32 convolution on v_mfma_f32_32x32x2_f32 (opt-in: ops.conv_precision("winograd"),
// round 4; developed as tools/probes/winograd_f23.hip, numerics in tools/winograd_study.py and DESIGN.md 7).
//   y = act(bias + sum x * w [+ res]) computed as U = G w G^T (host: the fragment panel Upk[16][Cin/8][2][N][4]), V = B^T d B per 4x4 input patch d
//   (patches overlap by two pixels), M_{xi,nu} = V_{xi,nu} U_{xi,nu} - sixteen GEMMs over Cin -, Y = A^T M A: 16 multiplies per 2x2 outputs instead of 36.
// Workgroup: 512 threads = 8 waves; block = 8 x 8 patches (16 x 16 output pixels) x 64 output channels; wave w owns the two products (xi,nu) = 2w, 2w+1
// for all 64 patches x 64 channels (2 x 2 x 2 accumulator tiles of 32 x 32 = 128 VGPRs).  A K stage is 8 input channels:
//   * U fragments: a 32x32x2 fp32 MFMA operand is ONE register, so a lane loads its four k-pairs of a (product, channel tile) as one 16-byte load
//     from the panel, one stage ahead, into the other of two register sets - the weights never touch LDS;
//   * the raw 18 x 18 x 8 halo patch goes global -> registers one stage ahead -> raw[s % 2] (channel-major planes);
//   * iteration s: eight groups of four MFMAs of stage s (fragments of Vs[s % 2], all read up front) with the next stage's input transform
//     (one (patch, channel) 4x4 -> V[16] per thread: raw[(s + 1) % 2] -> Vs[(s + 1) % 2]) cut into pieces behind them; ONE barrier.
// Epilogue: the 16 products of a (patch, channel) live in 8 different waves -> through LDS one 32-channel half at a time (rows padded to 65 floats),
// inverse transform + bias (+ residual, or the SPADE modulation of a gamma | beta launch) + activation per thread, NHWC stores with lanes on
// consecutive channels.  A second input (skip concatenation) is read stage by stage: a stage's 8 channels lie in one of the two tensors.
// Measured as a probe on 32 x 64x64 x 256 -> 256: 0.756 ms against 1.128 ms for lwg_conv_igemm_kernel (205 algorithmic TFLOP/s; the fp32 MFMA roof is 157).
// Rounding: relative L2 error against fp64 1.6x that of the direct fp32 convolution over the generator's layers (tools/winograd_study.py); NOT bitwise
// the direct kernel's result - which is why it is a precision mode of its own and not the default.
#include <hip/hip_runtime.h>
#include "lwg_common.h"
#include "lwg_conv_args.h"

#define WG_THREADS 512
#define TPB 8            // patches per block edge: 8 x 8 patches = 16 x 16 output pixels
#define NPATCH 64
#define NB 64            // output channels per block
#define KS 8             // input channels per stage
#define HALO 18
#define RAW_FLOATS (KS * HALO * HALO)        // [c][py][px]
// lab knobs (defaults = the measured kernel; variant libraries: tools/labvariant.sh NAME conv_winograd.hip -D...; tools/winolab.py times them)
#ifndef LWG_WINO_VSTRIDE
#define LWG_WINO_VSTRIDE 64      // floats between two k rows of Vs: 64 = dense (the two half-waves of a fragment read meet in the same banks), 96 = staggered
#endif
#ifndef LWG_WINO_PRIO
#define LWG_WINO_PRIO 0          // 1: s_setprio 1 around every MFMA group
#endif
#ifndef LWG_WINO_FRAG128
#define LWG_WINO_FRAG128 0       // 1: Vs as [xinu][kh][patch][4 k-pairs] - a lane's four fragments of a (product, patch tile) are ONE ds_read_b128 (VSTRIDE ignored)
#endif
#ifndef LWG_WINO_FINE
#define LWG_WINO_FINE 0          // 1: ONE MFMA per scheduling slot, the stage's other work cut into ~24 pieces of a few instructions behind them; barrier eight MFMAs early
#endif
#ifndef LWG_WINO_EPI4
#define LWG_WINO_EPI4 0          // 1: epilogue threads own (patch, four channels): 16-byte NHWC stores / residual loads
#endif
#define VIDX128(p, k, patch) (((((p) * 2 + ((k) & 1)) * NPATCH + (patch)) << 2) + ((k) >> 1))
#define VSTR LWG_WINO_VSTRIDE
#define VS_FLOATS (16 * KS * VSTR)           // [xinu][k][patch (+ pad)]
#define MS_STRIDE 65
#define MS_FLOATS (16 * 32 * MS_STRIDE)      // [xinu][n (32)][patch (64) + 1]

#ifdef LWG_WINO_TS           // lab: wave 0 of every workgroup stamps s_memtime into args->res (64 stamps per workgroup; LWG_EPI_NONE launches only; tools/winots.py)
#define WTS(i) do { if (tid == 0) reinterpret_cast<unsigned long long*>(const_cast<float*>(a.res))[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 64 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define WTS(i) do { } while (0)
#endif

template <int V> struct IntC { static constexpr int value = V; };

template <int EPI>
__global__ __launch_bounds__(WG_THREADS, 1) void lwg_conv_winograd_kernel(const LwgConvArgs a) {
    const float* __restrict__ x = a.x0;
    const float* __restrict__ U = a.w;
    const float* __restrict__ bias = a.bias;
    float* __restrict__ y = a.y;
    const int H = a.H, W = a.W, Cin = a.C0 + a.C1, N = a.N;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const raw0 = smem;                                // [2][RAW]
    float* const Vs0 = smem + 2 * RAW_FLOATS;                // [2][VS]
    float* Ms = smem;                                        // the epilogue's exchange buffer (after the K loop)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bx = (W + 2 * TPB - 1) / (2 * TPB), by = (H + 2 * TPB - 1) / (2 * TPB);
    int blk = blockIdx.x;
    const int b = blk / (bx * by);
    blk -= b * bx * by;
    const int x0 = (blk % bx) * 2 * TPB, y0 = (blk / bx) * 2 * TPB;
    const int n0 = blockIdx.y * NB;
    const float* xb = x + (size_t)b * H * W * a.C0;
    const float* xb1 = a.C1 ? a.x1 + (size_t)b * H * W * a.C1 : nullptr;      // second input, concatenated along C (skip connection)
    const int nst = Cin / KS;
    WTS(0);

    floatx16 acc[2][2][2];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[e][nb][tb][r] = 0.f;

    int roff[2];                                             // pixel index of this thread's raw float4s inside the image; -1: padding / none
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = tid + WG_THREADS * q;
        const int pix = i >> 1, py = pix / HALO, px = pix - py * HALO;
        const int gy = y0 - 1 + py, gx = x0 - 1 + px;
        roff[q] = (i < HALO * HALO * 2 && gy >= 0 && gy < H && gx >= 0 && gx < W) ? gy * W + gx : -1;
    }
    floatx4 rreg[2];
    auto rload = [&](int st) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            rreg[q] = floatx4{0.f, 0.f, 0.f, 0.f};
            const int c = st * KS + 4 * ((tid + WG_THREADS * q) & 1);          // a stage lies in ONE input (C0 % 8 == 0)
            if (roff[q] >= 0)
                rreg[q] = c < a.C0 ? *reinterpret_cast<const floatx4*>(xb + (size_t)roff[q] * a.C0 + c)
                                   : *reinterpret_cast<const floatx4*>(xb1 + (size_t)roff[q] * a.C1 + (c - a.C0));
        }
    };
    auto rstore = [&](int buf) {
        float* raw = raw0 + buf * RAW_FLOATS;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = tid + WG_THREADS * q;
            if (i < HALO * HALO * 2) {
                const int pix = i >> 1, half = i & 1;
#pragma unroll
                for (int k = 0; k < 4; ++k) raw[(4 * half + k) * (HALO * HALO) + pix] = rreg[q][k];
            }
        }
    };
    floatx4 ufr[2][2][2];                                    // [register set][product e][channel tile nb]: the four k-pairs of this lane's row
    const float* ubase = U + ((size_t)(lane >> 5) * N + n0 + (lane & 31)) * 4;
    auto uload = [&](int st, auto SET) {
        constexpr int set = decltype(SET)::value;
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
                ufr[set][e][nb] = *reinterpret_cast<const floatx4*>(ubase + (((size_t)(2 * wid + e) * nst + st) * 2 * N + nb * 32) * 4);
    };
    const int patch = tid & 63, tc = tid >> 6;
    const int pty = patch >> 3, ptx = patch & 7;
    auto transform = [&](int buf) {
        const float* d = raw0 + buf * RAW_FLOATS + tc * (HALO * HALO) + (2 * pty) * HALO + 2 * ptx;
        float* Vs = Vs0 + buf * VS_FLOATS;
        float t[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d0 = d[j], d1 = d[HALO + j], d2 = d[2 * HALO + j], d3 = d[3 * HALO + j];
            t[0][j] = d0 - d2;
            t[1][j] = d1 + d2;
            t[2][j] = d2 - d1;
            t[3][j] = d1 - d3;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#if LWG_WINO_FRAG128
            Vs[VIDX128(i * 4 + 0, tc, patch)] = t[i][0] - t[i][2];
            Vs[VIDX128(i * 4 + 1, tc, patch)] = t[i][1] + t[i][2];
            Vs[VIDX128(i * 4 + 2, tc, patch)] = t[i][2] - t[i][1];
            Vs[VIDX128(i * 4 + 3, tc, patch)] = t[i][1] - t[i][3];
#else
            float* v = Vs + ((i * 4) * KS + tc) * VSTR + patch;
            v[0 * KS * VSTR] = t[i][0] - t[i][2];
            v[1 * KS * VSTR] = t[i][1] + t[i][2];
            v[2 * KS * VSTR] = t[i][2] - t[i][1];
            v[3 * KS * VSTR] = t[i][1] - t[i][3];
#endif
        }
    };
#if LWG_WINO_FINE
    // FINE: one iteration = 32 slots of ONE MFMA (m = kk * 8 + e * 4 + nb * 2 + tb) each followed by at most a handful of the stage's other
    // instructions (sched_barrier keeps the order), so that neither wave of a SIMD ever leaves the matrix pipe waiting for a block of
    // vector / LDS / memory instructions: halo store (slots 0-1), next halo + weight loads (2-7), the next stage's input transform (reads 8-11,
    // first pass 16-17, second pass + V stores 18-21).  The stage's barrier sits after slot 23: the next stage's fragments for k-pairs 0..2 are
    // read right behind it into the registers slots 0-23 have finished with, the k-pair 3 fragments at the top of the next iteration.
    float fb[4][2][2];
    auto fragread = [&](int set, int kk) {
        const float* Vs = Vs0 + set * VS_FLOATS;
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) fb[kk][e][tb] = Vs[((2 * wid + e) * KS + 2 * kk + (lane >> 5)) * VSTR + tb * 32 + (lane & 31)];
    };
    auto iteration = [&](int s, auto SET, auto NXT) {
        constexpr int set = decltype(SET)::value;
        constexpr bool nxt = decltype(NXT)::value != 0;      // the last stage has no next one to prepare (peeled: no branches in the loop)
        const float* d = raw0 + (set ^ 1) * RAW_FLOATS + tc * (HALO * HALO) + (2 * pty) * HALO + 2 * ptx;
        float* Vn = Vs0 + (set ^ 1) * VS_FLOATS + tc * VSTR + patch;
        float* rawst = raw0 + set * RAW_FLOATS;
        const int s3 = s + 3 < nst ? s + 3 : nst - 1;        // past the end: a harmless re-load of the last stage (its halo store lands in a dead buffer)
        float dd[4][4];
        auto mf = [&](int m) {
            const int kk = m >> 3, e = (m >> 2) & 1, nb = (m >> 1) & 1, tb = m & 1;
            acc[e][nb][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(ufr[set][e][nb][kk], fb[kk][e][tb], acc[e][nb][tb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto rst = [&](int q) {                              // stage s + 2's halo (loaded during iteration s - 1) -> raw[s % 2]
            const int i = tid + WG_THREADS * q;
            if (nxt && i < HALO * HALO * 2) {
                const int pix = i >> 1, half = i & 1;
#pragma unroll
                for (int k = 0; k < 4; ++k) rawst[(4 * half + k) * (HALO * HALO) + pix] = rreg[q][k];
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto rld = [&](int q) {
            if (nxt) {
                const int c = s3 * KS + 4 * ((tid + WG_THREADS * q) & 1);
                const float* src = c < a.C0 ? xb + (size_t)roff[q] * a.C0 + c : xb1 + (size_t)roff[q] * a.C1 + (c - a.C0);
                floatx4 v = floatx4{0.f, 0.f, 0.f, 0.f};
                if (roff[q] >= 0) v = *reinterpret_cast<const floatx4*>(src);
                rreg[q] = v;
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto uld = [&](int e, int nb) {
            if (nxt) ufr[set ^ 1][e][nb] = *reinterpret_cast<const floatx4*>(ubase + (((size_t)(2 * wid + e) * nst + s + 1) * 2 * N + nb * 32) * 4);
            __builtin_amdgcn_sched_barrier(0);
        };
        auto ddr = [&](int i) {
            if (nxt) {
#pragma unroll
                for (int j = 0; j < 4; ++j) dd[i][j] = d[i * HALO + j];
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto t2 = [&](int i) {                               // row i of B^T d B: first pass over the four columns, second pass, V stores
            if (nxt) {
                float t[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    t[j] = i == 0 ? dd[0][j] - dd[2][j] : i == 1 ? dd[1][j] + dd[2][j] : i == 2 ? dd[2][j] - dd[1][j] : dd[1][j] - dd[3][j];
                float* v = Vn + (size_t)(i * 4) * KS * VSTR;
                v[0 * KS * VSTR] = t[0] - t[2];
                v[1 * KS * VSTR] = t[1] + t[2];
                v[2 * KS * VSTR] = t[2] - t[1];
                v[3 * KS * VSTR] = t[1] - t[3];
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        fragread(set, 3);                                    // k-pair 3 of THIS stage (its registers were busy until the previous slot 31)
        __builtin_amdgcn_sched_barrier(0);
        mf(0); rst(0);
        mf(1); rst(1);
        mf(2); rld(0);
        mf(3); rld(1);
        mf(4); uld(0, 0);
        mf(5); uld(0, 1);
        mf(6); uld(1, 0);
        mf(7); uld(1, 1);
        mf(8); ddr(0);
        mf(9); ddr(1);
        mf(10); ddr(2);
        mf(11); ddr(3);
        mf(12); mf(13); mf(14); mf(15);
        mf(16); t2(0);
        mf(17); t2(1);
        mf(18); t2(2);
        mf(19); t2(3);
        mf(20); mf(21); mf(22); mf(23);
        __syncthreads();
        if (nxt) {
            fragread(set ^ 1, 0);
            fragread(set ^ 1, 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        mf(24);
        if (nxt) fragread(set ^ 1, 2);
        __builtin_amdgcn_sched_barrier(0);
        mf(25); mf(26); mf(27); mf(28); mf(29); mf(30); mf(31);
    };
#else
    // one iteration = eight groups of four MFMAs (k-pair kk = g / 2, product e = g % 2) with the rest of the stage's work cut into pieces that
    // ride behind them (sched_barrier keeps the order): fragments of the whole stage read up front, then halo store / next loads / the next
    // stage's input transform in four pieces
    auto iteration = [&](int s, auto SET) {
        constexpr int set = decltype(SET)::value;
        const float* Vs = Vs0 + set * VS_FLOATS;
        float fb[4][2][2];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
#if LWG_WINO_FRAG128
                for (int tb = 0; tb < 2; ++tb) fb[kk][e][tb] = Vs[VIDX128(2 * wid + e, 2 * kk + (lane >> 5), tb * 32 + (lane & 31))];
#else
                for (int tb = 0; tb < 2; ++tb) fb[kk][e][tb] = Vs[((2 * wid + e) * KS + 2 * kk + (lane >> 5)) * VSTR + tb * 32 + (lane & 31)];
#endif
        const bool nxt = s + 1 < nst;
        const float* d = raw0 + (set ^ 1) * RAW_FLOATS + tc * (HALO * HALO) + (2 * pty) * HALO + 2 * ptx;
        float* Vn = Vs0 + (set ^ 1) * VS_FLOATS + tc * VSTR + patch;
        float dd[4][4], t[4][4];
        auto group = [&](int g) {
            const int kk = g >> 1, e = g & 1;
            if (LWG_WINO_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int tb = 0; tb < 2; ++tb)
                    acc[e][nb][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(ufr[set][e][nb][kk], fb[kk][e][tb], acc[e][nb][tb], 0, 0, 0);
            if (LWG_WINO_PRIO) __builtin_amdgcn_s_setprio(0);
        };
        group(0);
        if (s + 2 < nst) rstore(set);                        // stage s + 2's halo (loaded during iteration s - 1) -> raw[s % 2]
        __builtin_amdgcn_sched_barrier(0);
        group(1);
        if (s + 3 < nst) rload(s + 3);
        if (nxt) uload(s + 1, IntC<set ^ 1>());
        __builtin_amdgcn_sched_barrier(0);
        group(2);
        if (nxt) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) dd[i][j] = d[i * HALO + j];
        }
        __builtin_amdgcn_sched_barrier(0);
        group(3);
        __builtin_amdgcn_sched_barrier(0);
        group(4);
        if (nxt) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t[0][j] = dd[0][j] - dd[2][j];
                t[1][j] = dd[1][j] + dd[2][j];
                t[2][j] = dd[2][j] - dd[1][j];
                t[3][j] = dd[1][j] - dd[3][j];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        group(5);
        if (nxt) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#if LWG_WINO_FRAG128
                float* vb = Vs0 + (set ^ 1) * VS_FLOATS;
                vb[VIDX128(i * 4 + 0, tc, patch)] = t[i][0] - t[i][2];
                vb[VIDX128(i * 4 + 1, tc, patch)] = t[i][1] + t[i][2];
                vb[VIDX128(i * 4 + 2, tc, patch)] = t[i][2] - t[i][1];
                vb[VIDX128(i * 4 + 3, tc, patch)] = t[i][1] - t[i][3];
#else
                float* v = Vn + (size_t)(i * 4) * KS * VSTR;
                v[0 * KS * VSTR] = t[i][0] - t[i][2];
                v[1 * KS * VSTR] = t[i][1] + t[i][2];
                v[2 * KS * VSTR] = t[i][2] - t[i][1];
                v[3 * KS * VSTR] = t[i][1] - t[i][3];
#endif
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        group(6);
        if (nxt) {
#pragma unroll
            for (int i = 2; i < 4; ++i) {
#if LWG_WINO_FRAG128
                float* vb = Vs0 + (set ^ 1) * VS_FLOATS;
                vb[VIDX128(i * 4 + 0, tc, patch)] = t[i][0] - t[i][2];
                vb[VIDX128(i * 4 + 1, tc, patch)] = t[i][1] + t[i][2];
                vb[VIDX128(i * 4 + 2, tc, patch)] = t[i][2] - t[i][1];
                vb[VIDX128(i * 4 + 3, tc, patch)] = t[i][1] - t[i][3];
#else
                float* v = Vn + (size_t)(i * 4) * KS * VSTR;
                v[0 * KS * VSTR] = t[i][0] - t[i][2];
                v[1 * KS * VSTR] = t[i][1] + t[i][2];
                v[2 * KS * VSTR] = t[i][2] - t[i][1];
                v[3 * KS * VSTR] = t[i][1] - t[i][3];
#endif
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        group(7);
        __syncthreads();
    };
#endif

    // prologue: stage 0 transformed, stage 1 in raw[1], stage 2's halo in registers, U(0) in set 0
    rload(0);
    uload(0, IntC<0>());
    rstore(0);
    if (nst > 1) rload(1);
    __syncthreads();
    transform(0);
    if (nst > 1) rstore(1);
    if (nst > 2) rload(2);
    __syncthreads();
#if LWG_WINO_FINE
    fragread(0, 0);
    fragread(0, 1);
    fragread(0, 2);
#endif
    WTS(1);
#if LWG_WINO_FINE
    {
        int s = 0;
        for (; s + 2 < nst; s += 2) {
            iteration(s, IntC<0>(), IntC<1>());
            iteration(s + 1, IntC<1>(), IntC<1>());
            WTS(2 + (s >> 1));
        }
        iteration(s, IntC<0>(), IntC<1>());                  // nst is even (host: Cin % 16 == 0)
        iteration(s + 1, IntC<1>(), IntC<0>());
    }
#else
    for (int s = 0; s < nst; s += 2) {
        iteration(s, IntC<0>());
        if (s + 1 < nst) iteration(s + 1, IntC<1>());
        WTS(2 + (s >> 1));
    }
#endif
    WTS(40);
    // epilogue, one 32-channel half at a time: products -> LDS [xinu][n][patch], inverse transform, bias, (residual | SPADE modulation), activation,
    // NHWC stores.  LWG_EPI_SPADE: the block's 64 columns are gamma | beta of the SAME 32 channels (the host interleaves the stacked panel in blocks
    // of 32, as for lwg_conv_igemm_kernel): the first half leaves gamma in registers, the second forms (xn - mean) rstd (1 + gamma) + beta.
#if LWG_WINO_EPI4
    floatx4 gam4[2][2];
#else
    float gam[4][2][2];
#endif
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int p = 2 * wid + e;
#pragma unroll
            for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);           // D layout: row of this accumulator register
                    Ms[(p * 32 + n) * MS_STRIDE + tb * 32 + (lane & 31)] = acc[e][nb][tb][r];
                }
        }
        __syncthreads();
#if LWG_WINO_EPI4
        {   // a thread owns one patch x four consecutive channels: 64 conflict-free LDS reads, 16-byte NHWC stores / residual / xn loads
            const int n4 = (tid & 7) * 4, ep = tid >> 3;
            const int ty = ep >> 3, tx = ep & 7;
            floatx4 o[2][2];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float m[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) m[i][j] = Ms[((i * 4 + j) * 32 + n4 + c) * MS_STRIDE + ep];
                float sr[2][4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    sr[0][j] = m[0][j] + m[1][j] + m[2][j];
                    sr[1][j] = m[1][j] - m[2][j] - m[3][j];
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    o[i][0][c] = sr[i][0] + sr[i][1] + sr[i][2];
                    o[i][1][c] = sr[i][1] - sr[i][2] - sr[i][3];
                }
            }
            const floatx4 bv = bias ? *reinterpret_cast<const floatx4*>(bias + n0 + nb * 32 + n4) : floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int px = 0; px < 2; ++px) {
                    const int oy = y0 + 2 * ty + i, ox = x0 + 2 * tx + px;
                    floatx4 v = o[i][px] + bv;
                    if (EPI == LWG_EPI_SPADE) {
                        if (nb == 0) {
                            gam4[i][px] = v;
                        } else if (oy < H && ox < W) {
                            const int ch = (n0 >> 1) + n4;                             // the modulated channels
                            const size_t off = (((size_t)b * H + oy) * W + ox) * a.YC + ch;
                            const floatx4 mu = *reinterpret_cast<const floatx4*>(a.mean + (size_t)b * a.YC + ch);
                            const floatx4 rs = *reinterpret_cast<const floatx4*>(a.rstd + (size_t)b * a.YC + ch);
                            const floatx4 xv = *reinterpret_cast<const floatx4*>(a.xn + off);
                            floatx4 r;
#pragma unroll
                            for (int c = 0; c < 4; ++c) r[c] = lwg_act((xv[c] - mu[c]) * rs[c] * (1.f + gam4[i][px][c]) + v[c], a.act);
                            *reinterpret_cast<floatx4*>(y + off) = r;
                        }
                    } else if (oy < H && ox < W) {
                        const size_t off = (((size_t)b * H + oy) * W + ox) * a.YC + a.ycoff + n0 + nb * 32 + n4;
                        if (EPI == LWG_EPI_RESIDUAL) v += *reinterpret_cast<const floatx4*>(a.res + off);
                        floatx4 r;
#pragma unroll
                        for (int c = 0; c < 4; ++c) r[c] = lwg_act(v[c], a.act);
                        *reinterpret_cast<floatx4*>(y + off) = r;
                    }
                }
        }
#else
#pragma unroll
        for (int it = 0; it < NPATCH * 32 / WG_THREADS; ++it) {
            const int q = tid + WG_THREADS * it;
            const int n = q & 31, patch = q >> 5;
            const int ty = patch >> 3, tx = patch & 7;
            float m[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) m[i][j] = Ms[((i * 4 + j) * 32 + n) * MS_STRIDE + patch];
            float s[2][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s[0][j] = m[0][j] + m[1][j] + m[2][j];
                s[1][j] = m[1][j] - m[2][j] - m[3][j];
            }
            const float bv = bias ? bias[n0 + nb * 32 + n] : 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int oy = y0 + 2 * ty + i;
                const float o0 = s[i][0] + s[i][1] + s[i][2] + bv, o1 = s[i][1] - s[i][2] - s[i][3] + bv;
                const int ox = x0 + 2 * tx;
                if (EPI == LWG_EPI_SPADE) {
                    if (nb == 0) {
                        gam[it][i][0] = o0;
                        gam[it][i][1] = o1;
                    } else {
                        const int ch = (n0 >> 1) + n;                                  // the modulated channel
                        const size_t o = (((size_t)b * H + oy) * W + ox) * a.YC + ch;
                        const float mu = a.mean[(size_t)b * a.YC + ch], rs = a.rstd[(size_t)b * a.YC + ch];
                        if (oy < H && ox < W) y[o] = lwg_act((a.xn[o] - mu) * rs * (1.f + gam[it][i][0]) + o0, a.act);
                        if (oy < H && ox + 1 < W) y[o + a.YC] = lwg_act((a.xn[o + a.YC] - mu) * rs * (1.f + gam[it][i][1]) + o1, a.act);
                    }
                } else {
                    const size_t o = (((size_t)b * H + oy) * W + ox) * a.YC + a.ycoff + n0 + nb * 32 + n;
                    if (oy < H && ox < W) y[o] = lwg_act(o0 + (EPI == LWG_EPI_RESIDUAL ? a.res[o] : 0.f), a.act);
                    if (oy < H && ox + 1 < W) y[o + a.YC] = lwg_act(o1 + (EPI == LWG_EPI_RESIDUAL ? a.res[o + a.YC] : 0.f), a.act);
                }
            }
        }
#endif
        __syncthreads();
        WTS(41 + nb);
    }
}


// args: the launch description of the 3 x 3 / stride 1 / pad 1 convolution as lwg_conv2d_nhwc_f32 takes it (nine taps, omul = 1, OH = H, OW = W,
// one or two inputs with C0 % 8 == 0 and C1 % 8 == 0, N % 64 == 0; LWG_EPI_NONE, LWG_EPI_RESIDUAL or LWG_EPI_SPADE (N = 2 YC, columns gamma | beta
// interleaved in blocks of 32, ycoff = 0); any activation of lwg_act) EXCEPT args->w = the Winograd fragment panel Upk[16][Cin/8][2][N][4]:
// element (p, s, kh, n, kk) = (G w G^T)[xi = p / 4][nu = p % 4] of input channel 8 s + 2 kk + kh (concatenated order) and output column n.
extern "C" int lwg_conv2d_winograd_f32(const LwgConvArgs* pa, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!pa) return (int)hipErrorInvalidValue;
    const LwgConvArgs& a = *pa;
    if (!a.x0 || !a.w || !a.y || a.M <= 0 || a.ntaps != 9 || a.stride != 1 || a.omul != 1 || a.C0 <= 0 || (a.C0 % KS) != 0 || a.C1 < 0 ||
        (a.C1 % KS) != 0 || (a.C1 > 0 && !a.x1) || a.N <= 0 || (a.N % NB) != 0 || a.OH != a.H || a.OW != a.W || a.YH != a.H || a.YW != a.W ||
        a.xdt != LWG_DT_F32 || a.ydt != LWG_DT_F32 || a.M != a.B * a.H * a.W || a.ycoff < 0 || a.act == LWG_ACT_RELU_MASK)
        return (int)hipErrorInvalidValue;
    if (a.epi == LWG_EPI_SPADE) {
        if (!a.xn || !a.mean || !a.rstd || !a.bias || a.YC * 2 != a.N || a.ycoff != 0) return (int)hipErrorInvalidValue;
    } else {
        if (a.ycoff + a.N > a.YC) return (int)hipErrorInvalidValue;
        if (a.epi != LWG_EPI_NONE && (a.epi != LWG_EPI_RESIDUAL || !a.res)) return (int)hipErrorInvalidValue;
    }
    if ((unsigned long long)a.H * a.W >= 0x7fffffffull) return (int)hipErrorInvalidValue;
    const size_t loop = (size_t)(RAW_FLOATS + VS_FLOATS) * 8, epi = (size_t)MS_FLOATS * 4;
    const size_t lds = loop > epi ? loop : epi;
    const int bx = (a.W + 2 * TPB - 1) / (2 * TPB), by = (a.H + 2 * TPB - 1) / (2 * TPB);
    const dim3 grid((unsigned)(bx * by * a.B), (unsigned)(a.N / NB));
    static unsigned long long done[3] = {0, 0, 0};
#define LWG_WINO_GO(E, SLOT)                                                                                                          \
    {                                                                                                                                 \
        if (hipError_t e = lwg_allow_dynamic_lds(reinterpret_cast<const void*>(lwg_conv_winograd_kernel<E>), lds, done[SLOT]); e != hipSuccess) \
            return (int)e;                                                                                                            \
        hipLaunchKernelGGL(lwg_conv_winograd_kernel<E>, grid, dim3(WG_THREADS), lds, stream, a);                                      \
    }
    if (a.epi == LWG_EPI_RESIDUAL) LWG_WINO_GO(LWG_EPI_RESIDUAL, 1)
    else if (a.epi == LWG_EPI_SPADE) LWG_WINO_GO(LWG_EPI_SPADE, 2)
    else LWG_WINO_GO(LWG_EPI_NONE, 0)
#undef LWG_WINO_GO
    return (int)hipGetLastError();
}
