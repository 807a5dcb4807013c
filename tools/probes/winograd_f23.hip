// Probe (lab tool, NOT part of the library): a fused F(2x2, 3x3) Winograd form of the 3x3 / stride 1 / pad 1 fp32 convolution on
// v_mfma_f32_32x32x2_f32 (the round-5 project).  Version 0 (-DWINO_V=0): correct first, synchronous stages, no overlap;
// version 2 (default): pipelined, one barrier per stage, weights streamed through registers.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC tools/probes/winograd_f23.hip -o tools/lab/libwino.so   (tools/winolab.py drives it)
//
// y[b][oy][ox][n] = act(bias[n] + sum_{c,ky,kx} x[b][oy+ky-1][ox+kx-1][c] w[n][c][ky][kx]) computed as
//   U = G w G^T (host, panel [16][Cin][N]),  V = B^T d B per 4x4 input patch d (stride 2),  M_{xi,nu} = V_{xi,nu} U_{xi,nu} (16 GEMMs over Cin),
//   Y = A^T M A (2x2 outputs per patch): 16 multiplies per 4 outputs instead of 36.
// Workgroup: 512 threads = 8 waves; block = 8 x 8 patches (16 x 16 output pixels) x 64 output channels; wave w owns the two products
// (xi,nu) = 2w, 2w+1 for all 64 patches x 64 channels (2 x 2 x 2 accumulator tiles of 32 x 32 = 128 VGPRs).  K stage = 8 input channels:
//   (1) raw 18 x 18 x 8 halo patch + the U slice [16][8][64] global -> LDS; (2) every thread transforms ONE (patch, channel) 4x4 -> V[16];
//   (3) four k-pairs of MFMAs from LDS fragments (one ds_read_b32 each: a 32x32x2 fp32 operand is one register).
// Epilogue: the 16 products of a (patch, channel) live in 8 different waves -> through LDS (rows padded to 65 floats), inverse
// transform + bias + activation per thread, NHWC stores with lanes on consecutive channels.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

#define WG_THREADS 512
#define TPB 8            // patches per block edge: 8 x 8 patches = 16 x 16 output pixels
#define NPATCH 64
#define NB 64            // output channels per block
#define KS 8             // input channels per stage
#define HALO 18
#define RAW_FLOATS (KS * HALO * HALO)        // [c][py][px]
#define US_FLOATS (16 * KS * NB)             // [xinu][k][n]
#define VS_FLOATS (16 * KS * NPATCH)         // [xinu][k][patch]
#define MS_STRIDE 65
#define MS_FLOATS (16 * 32 * MS_STRIDE)      // [xinu][n (32)][patch (64) + 1]

__device__ __forceinline__ float act_of(float v, int act) { return act == 1 ? (v > 0.f ? v : 0.f) : v; }

// Version 2 (default): software-pipelined over the 8-channel stages, ONE barrier per stage, the weights never touch LDS.
//   * U fragments: a 32x32x2 fp32 MFMA operand is one register and every wave owns its own two products, so a lane loads ITS four k-pairs of a
//     (product, channel tile) as one 16-byte load from the panel Upk[16][Cin/8][2][N][4] (element (p, s, kh, n, kk) = U[p][8 s + 2 kk + kh][n]),
//     one stage ahead, into the other of two register sets.
//   * raw halo patch: global -> registers one stage ahead -> raw[s % 2] at the top of iteration s (free since iteration s - 1's barrier).
//   * iteration s: MFMAs of stage s (Vs[s % 2]) with the input transform of stage s + 1 (raw[(s + 1) % 2] -> Vs[(s + 1) % 2]) in their shadow; barrier.
// LDS: 2 x (10.4 + 32.8) KB for the loop, 133 KB for the epilogue's exchange.
#ifndef WINO_V
#define WINO_V 2
#endif
#if WINO_V == 2
template <int V> struct IntC { static constexpr int value = V; };
extern "C" __global__ __launch_bounds__(WG_THREADS, 1) void wino_f23_kernel(const float* __restrict__ x, const float* __restrict__ U,
                                                                         const float* __restrict__ bias, float* __restrict__ y, int B, int H,
                                                                         int W, int Cin, int N, int act) {
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
    const float* xb = x + (size_t)b * H * W * Cin;
    const int nst = Cin / KS;

    floatx16 acc[2][2][2];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[e][nb][tb][r] = 0.f;

    int roff[2];                                             // element offset of this thread's raw float4s inside the image (without c0); -1: padding / none
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = tid + WG_THREADS * q;
        const int pix = i >> 1, py = pix / HALO, px = pix - py * HALO;
        const int gy = y0 - 1 + py, gx = x0 - 1 + px;
        roff[q] = (i < HALO * HALO * 2 && gy >= 0 && gy < H && gx >= 0 && gx < W) ? (gy * W + gx) * Cin + 4 * (i & 1) : -1;
    }
    floatx4 rreg[2];
    auto rload = [&](int st) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            rreg[q] = floatx4{0.f, 0.f, 0.f, 0.f};
            if (roff[q] >= 0) rreg[q] = *reinterpret_cast<const floatx4*>(xb + roff[q] + st * KS);
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
            float* v = Vs + ((i * 4) * KS + tc) * NPATCH + patch;
            v[0 * KS * NPATCH] = t[i][0] - t[i][2];
            v[1 * KS * NPATCH] = t[i][1] + t[i][2];
            v[2 * KS * NPATCH] = t[i][2] - t[i][1];
            v[3 * KS * NPATCH] = t[i][1] - t[i][3];
        }
    };
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
                for (int tb = 0; tb < 2; ++tb) fb[kk][e][tb] = Vs[((2 * wid + e) * KS + 2 * kk + (lane >> 5)) * NPATCH + tb * 32 + (lane & 31)];
        const bool nxt = s + 1 < nst;
        const float* d = raw0 + (set ^ 1) * RAW_FLOATS + tc * (HALO * HALO) + (2 * pty) * HALO + 2 * ptx;
        float* Vn = Vs0 + (set ^ 1) * VS_FLOATS + tc * NPATCH + patch;
        float dd[4][4], t[4][4];
        auto group = [&](int g) {
            const int kk = g >> 1, e = g & 1;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int tb = 0; tb < 2; ++tb)
                    acc[e][nb][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(ufr[set][e][nb][kk], fb[kk][e][tb], acc[e][nb][tb], 0, 0, 0);
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
                float* v = Vn + (size_t)(i * 4) * KS * NPATCH;
                v[0 * KS * NPATCH] = t[i][0] - t[i][2];
                v[1 * KS * NPATCH] = t[i][1] + t[i][2];
                v[2 * KS * NPATCH] = t[i][2] - t[i][1];
                v[3 * KS * NPATCH] = t[i][1] - t[i][3];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        group(6);
        if (nxt) {
#pragma unroll
            for (int i = 2; i < 4; ++i) {
                float* v = Vn + (size_t)(i * 4) * KS * NPATCH;
                v[0 * KS * NPATCH] = t[i][0] - t[i][2];
                v[1 * KS * NPATCH] = t[i][1] + t[i][2];
                v[2 * KS * NPATCH] = t[i][2] - t[i][1];
                v[3 * KS * NPATCH] = t[i][1] - t[i][3];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        group(7);
        __syncthreads();
    };

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
    for (int s = 0; s < nst; s += 2) {
        iteration(s, IntC<0>());
        if (s + 1 < nst) iteration(s + 1, IntC<1>());
    }
    // epilogue, one 32-channel half at a time: products -> LDS [xinu][n][patch], inverse transform, bias, activation, NHWC stores
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
        for (int q = tid; q < NPATCH * 32; q += WG_THREADS) {
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
            const int ch = n0 + nb * 32 + n;
            const float bv = bias ? bias[ch] : 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int oy = y0 + 2 * ty + i;
                const float o0 = s[i][0] + s[i][1] + s[i][2], o1 = s[i][1] - s[i][2] - s[i][3];
                const int ox = x0 + 2 * tx;
                if (oy < H && ox < W) y[(((size_t)b * H + oy) * W + ox) * N + ch] = act_of(o0 + bv, act);
                if (oy < H && ox + 1 < W) y[(((size_t)b * H + oy) * W + ox + 1) * N + ch] = act_of(o1 + bv, act);
            }
        }
        __syncthreads();
    }
}

#else
extern "C" __global__ __launch_bounds__(WG_THREADS, 1) void wino_f23_kernel(const float* __restrict__ x, const float* __restrict__ U,
                                                                         const float* __restrict__ bias, float* __restrict__ y, int B, int H,
                                                                         int W, int Cin, int N, int act) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* raw = smem;                         // stage buffers ...
    float* Us = raw + RAW_FLOATS;
    float* Vs = Us + US_FLOATS;
    float* Ms = smem;                          // ... reused by the epilogue
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bx = (W + 2 * TPB - 1) / (2 * TPB), by = (H + 2 * TPB - 1) / (2 * TPB);
    int blk = blockIdx.x;
    const int b = blk / (bx * by);
    blk -= b * bx * by;
    const int x0 = (blk % bx) * 2 * TPB, y0 = (blk / bx) * 2 * TPB;      // first output pixel of the block
    const int n0 = blockIdx.y * NB;
    const float* xb = x + (size_t)b * H * W * Cin;

    floatx16 acc[2][2][2];                     // [product e][channel tile nb][patch tile tb]
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[e][nb][tb][r] = 0.f;

    for (int c0 = 0; c0 < Cin; c0 += KS) {
        // (1) raw halo patch: pixel (py, px) = image (y0 - 1 + py, x0 - 1 + px); two float4 (8 channels) per pixel; channel-major planes in LDS
        for (int i = tid; i < HALO * HALO * 2; i += WG_THREADS) {
            const int pix = i >> 1, half = i & 1;
            const int py = pix / HALO, px = pix - py * HALO;
            const int gy = y0 - 1 + py, gx = x0 - 1 + px;
            floatx4 v = {0.f, 0.f, 0.f, 0.f};
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = *reinterpret_cast<const floatx4*>(xb + ((size_t)gy * W + gx) * Cin + c0 + 4 * half);
#pragma unroll
            for (int k = 0; k < 4; ++k) raw[(4 * half + k) * (HALO * HALO) + pix] = v[k];
        }
        // the U slice [16][8][64] of this stage: U is [16][Cin][N]
        for (int i = tid; i < US_FLOATS / 4; i += WG_THREADS) {
            const int n4 = i % (NB / 4), k = (i / (NB / 4)) % KS, p = i / (NB / 4 * KS);
            *reinterpret_cast<floatx4*>(Us + (p * KS + k) * NB + 4 * n4) =
                *reinterpret_cast<const floatx4*>(U + ((size_t)p * Cin + c0 + k) * N + n0 + 4 * n4);
        }
        __syncthreads();
        // (2) input transform V = B^T d B of one (patch, channel) per thread
        {
            const int patch = tid & 63, c = tid >> 6;
            const int ty = patch >> 3, tx = patch & 7;
            const float* d = raw + c * (HALO * HALO) + (2 * ty) * HALO + 2 * tx;
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
                float* v = Vs + ((i * 4) * KS + c) * NPATCH + patch;
                v[0 * KS * NPATCH] = t[i][0] - t[i][2];
                v[1 * KS * NPATCH] = t[i][1] + t[i][2];
                v[2 * KS * NPATCH] = t[i][2] - t[i][1];
                v[3 * KS * NPATCH] = t[i][1] - t[i][3];
            }
        }
        __syncthreads();
        // (3) the products of this wave: D^T tiles (rows = output channels, columns = patches)
#pragma unroll
        for (int kk = 0; kk < KS / 2; ++kk) {
            const int k = 2 * kk + (lane >> 5);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int p = 2 * wid + e;
                float fa[2], fb[2];
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) fa[nb] = Us[(p * KS + k) * NB + nb * 32 + (lane & 31)];
#pragma unroll
                for (int tb = 0; tb < 2; ++tb) fb[tb] = Vs[(p * KS + k) * NPATCH + tb * 32 + (lane & 31)];
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int tb = 0; tb < 2; ++tb) acc[e][nb][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[nb], fb[tb], acc[e][nb][tb], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // epilogue, one 32-channel half at a time: products -> LDS [xinu][n][patch], inverse transform, bias, activation, NHWC stores
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
        for (int q = tid; q < NPATCH * 32; q += WG_THREADS) {
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
            const int ch = n0 + nb * 32 + n;
            const float bv = bias ? bias[ch] : 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int oy = y0 + 2 * ty + i;
                const float o0 = s[i][0] + s[i][1] + s[i][2], o1 = s[i][1] - s[i][2] - s[i][3];
                const int ox = x0 + 2 * tx;
                if (oy < H && ox < W) y[(((size_t)b * H + oy) * W + ox) * N + ch] = act_of(o0 + bv, act);
                if (oy < H && ox + 1 < W) y[(((size_t)b * H + oy) * W + ox + 1) * N + ch] = act_of(o1 + bv, act);
            }
        }
        __syncthreads();
    }
}

#endif

// x (B,H,W,Cin) NHWC fp32, Cin % 8 == 0; U = G w G^T (index xi * 4 + nu): version 2 takes the fragment panel Upk[16][Cin/8][2][N][4] (see above),
// version 0 the plain [16][Cin][N]; bias (N) or NULL; y (B,H,W,N); N % 64 == 0; act: 0 none, 1 ReLU
extern "C" int wino_conv3x3_f32(const float* x, const float* U, const float* bias, float* y, int B, int H, int W, int Cin, int N, int act,
                                void* stream) {
    if (!x || !U || !y || B <= 0 || H <= 0 || W <= 0 || Cin % KS != 0 || N % NB != 0) return 1;
    const size_t stage = WINO_V == 2 ? (size_t)(RAW_FLOATS + VS_FLOATS) * 8 : (size_t)(RAW_FLOATS + US_FLOATS + VS_FLOATS) * 4, epi = (size_t)MS_FLOATS * 4;
    const size_t lds = stage > epi ? stage : epi;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(wino_f23_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 2;
        attr = true;
    }
    const int bx = (W + 2 * TPB - 1) / (2 * TPB), by = (H + 2 * TPB - 1) / (2 * TPB);
    hipLaunchKernelGGL(wino_f23_kernel, dim3((unsigned)(bx * by * B), (unsigned)(N / NB)), dim3(WG_THREADS), lds, reinterpret_cast<hipStream_t>(stream), x, U, bias,
                       y, B, H, W, Cin, N, act);
    return (int)hipGetLastError();
}
