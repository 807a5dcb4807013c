// Weight gradient of the NHWC fp32 implicit-GEMM convolution on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Backward of every torch.nn.Conv2d / ConvTranspose2d of the generator and discriminator in the personalization step
// (reference iPERCore/tools/trainers/lwg_trainer.py:326-352 loss.backward(); layers as in
// models/networks/generators/attlwb_spade_resunet.py and discriminators/patch_dis.py).  The data gradient needs no
// kernel of its own: it is the forward kernel (csrc/conv_igemm.hip) run on dY with a transposed panel
// (networks/packing.py: pack_dgrad).
//
// GEMM view:  dW[K, N] = A[M, K]^T * dY[M, N],  the reduction runs over the M = B*OH*OW output positions.
//   * A is the same implicit operand as in the forward pass (row m = output position, k = (tap, channel) sampled at
//     (oy*stride + dy, ox*stride + dx), zero outside the image), gathered with raw buffer loads (hardware zero fill);
//     K in the forward panel's order (32-channel chunk major, tap minor; tap*Cin + c for the small-Cin first layers).
//   * Workgroup = 4 waves, output tile 128 (k) x 128 (n), wave tile 64 x 64 = 2x2 MFMA tiles; the M range of the
//     workgroup is walked in chunks of 32 rows: A chunk [32][128] and dY chunk [32][128] are staged in LDS
//     (double buffered, loads of chunk i+1 in flight during the MFMAs of chunk i).  v_mfma_f32_32x32x2: lanes 0-31
//     supply reduction index 2s, lanes 32-63 index 2s+1, so a fragment is one ds_read_b32 of a [m][k] row.
//   * The reduction is split over gridDim.y workgroups; each writes its partial tile to a workspace slab and
//     lwg_wgrad_reduce adds the slabs in slab order (deterministic - no float atomics).
#include <type_traits>

#include "lwg_common.h"
#include "lwg_conv_args.h"

#define LWG_OOB_OFFSET 0xC0000000u

__device__ __forceinline__ floatx4 lwg_wg_buf_load(const float* base, unsigned bytes, unsigned voff) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
    return __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0));
}

// a: the FORWARD geometry (x0/x1, taps, stride, OH/OW/M, N, output mapping YH/YW/YC/ycoff/omul/ooy/oox); a.y is unused.
// dy_: gradient of the forward output, laid out like the forward y.  part: [gridDim.y][Ktot][N] partial sums.
// BN: columns of the output tile.  128: wave tile 64 (k) x 64 (n) = 2 x 2 MFMA tiles.  64: wave tile 64 (k) x 32 (n) = 2 x 1 - for
// the N = 64 layers (last up-sampling layer, first discriminator / encoder layers), where a 128-column tile spends half of its MFMAs on
// columns that do not exist (33 - 53 TFLOP/s on those launches).
template <bool SMALLC, int BN>
__global__ __launch_bounds__(256, 2) void lwg_conv_wgrad_kernel(const LwgConvArgs a, const float* __restrict__ dy_, int Ktot,
                                                               int chunks_per_split, float* __restrict__ part, int want_bias) {
    constexpr int BK = 128, BR = 32;                     // output tile rows and reduction chunk
    constexpr int NT = BN / 64;                          // 32-column MFMA tiles per wave
    constexpr int NYP = BN / 32;                         // 16-byte dY pieces per thread and chunk
    constexpr int ROW = 128;                             // floats per staged row (no pad: see header)
    constexpr int STAGE = BR * ROW;                      // one operand, one stage
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                                    // [2][BR][ROW]
    float* Ys = smem + 2 * STAGE;                        // [2][BR][ROW]
    int* taptab = reinterpret_cast<int*>(smem + 4 * STAGE);  // packed dy | dx << 16

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wk = wid >> 1, wn = wid & 1;
    const int tiles_n = (a.N + BN - 1) / BN;
    const int tile_k = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
    const int k_base = tile_k * BK, n_base = tile_n * BN;
    if (tid < a.ntaps) taptab[tid] = (a.dy[tid] & 0xffff) | (a.dx[tid] << 16);
    __syncthreads();

    // ---- this thread's slice of the gather: row mrow of the chunk, channel quad kq of each of the 4 groups of 32 k ----
    const int mrow = tid >> 3, kq = tid & 7;
    const unsigned bytes0 = (unsigned)a.B * a.H * a.W * a.C0 * 4u;
    const unsigned bytes1 = (unsigned)a.B * a.H * a.W * a.C1 * 4u;
    const unsigned ybytes = (unsigned)a.B * a.YH * a.YW * a.YC * 4u;
    // per group: source pointer, bytes, per-pixel channel count, channel offset, tap (dy, dx); k beyond Ktot -> invalid
    const float* gsrc[4];
    unsigned gbytes[4];
    int gcs[4], gcoff[4], gdy[4], gdx[4];
    bool gok[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        int tap, c;
        const int k0 = k_base + g * 32;                  // first k of the group
        gok[g] = k0 < Ktot;
        bool use1 = false;                               // workgroup-uniform (a 32-channel chunk never straddles x0 | x1):
        if (SMALLC) {                                    // keeps the buffer descriptor in SGPRs - no waterfall loop
            const int k4 = (k0 >> 2) + kq;               // this lane's k-quad: tap and channel depend on the lane
            tap = k4 >> a.cshift;
            c = (k4 & ((1 << a.cshift) - 1)) * 4;
            gok[g] = gok[g] && tap < a.ntaps;
        } else {
            const int G = k0 >> 5;                       // group id = cchunk * ntaps + tap
            const int cchunk = G / a.ntaps;
            tap = G - cchunk * a.ntaps;
            c = cchunk * 32 + kq * 4;
            use1 = cchunk * 32 >= a.C0;
        }
        const int tp = gok[g] ? tap : 0;
        const int packed = taptab[tp];
        gdy[g] = (int)(short)(packed & 0xffff);
        gdx[g] = packed >> 16;
        gsrc[g] = use1 ? a.x1 : a.x0;
        gbytes[g] = use1 ? bytes1 : bytes0;
        gcs[g] = use1 ? a.C1 : a.C0;
        gcoff[g] = use1 ? c - a.C0 : c;
    }
    const int HW = a.OH * a.OW;
    const bool direct = (a.omul == 1) && (a.YH == a.OH) && (a.YW == a.OW);

    const int nchunks_total = (a.M + BR - 1) / BR;
    const int c_begin = blockIdx.y * chunks_per_split;
    const int c_end = min(nchunks_total, c_begin + chunks_per_split);

    // this thread's GEMM row of the chunk being loaded: (image, oy, ox), advanced by 32 rows per chunk without divisions
    int rm = c_begin * BR + mrow;
    int rb, roy, rox;
    {
        const int mm = rm < a.M ? rm : 0;
        rb = mm / HW;
        const int rem = mm - rb * HW;
        roy = rem / a.OW;
        rox = rem - roy * a.OW;
    }
    floatx4 rx[4], ry[NYP];
    // Bias gradient (column sums of dY) as row Ktot of the slab: the workgroups of k-tile 0 add up the dY quads they stage anyway
    // (NYP vector adds per chunk) - no separate pass over dY (lwg_colsum_nhwc_f32: 102 launch pairs per training step).
    const bool do_bias = want_bias != 0 && tile_k == 0;
    floatx4 cs[NYP];
#pragma unroll
    for (int j = 0; j < NYP; ++j) cs[j] = floatx4{0.f, 0.f, 0.f, 0.f};
    // The eight loads of a chunk are issued as eight separate pieces between the MFMAs of the current chunk.
    // row_setup(): per-chunk base offsets of this thread's row; load_piece(i): i < 4 activation group i, else dY quad i-4.
    const unsigned tapb[4] = {   // (tap offset in pixels) * channels of the group's source * 4 + channel offset * 4  (constants)
        (unsigned)((gdy[0] * a.W + gdx[0]) * gcs[0] + gcoff[0]) * 4u, (unsigned)((gdy[1] * a.W + gdx[1]) * gcs[1] + gcoff[1]) * 4u,
        (unsigned)((gdy[2] * a.W + gdx[2]) * gcs[2] + gcoff[2]) * 4u, (unsigned)((gdy[3] * a.W + gdx[3]) * gcs[3] + gcoff[3]) * 4u};
    bool r_ok = false;
    int r_iy0 = 0, r_ix0 = 0;
    unsigned r_pix = 0, r_yoff = 0;
    auto row_setup = [&]() {
        r_ok = rm < a.M;
        r_iy0 = roy * a.stride;
        r_ix0 = rox * a.stride;
        r_pix = (unsigned)((rb * a.H + r_iy0) * a.W + r_ix0);
        const size_t opix = direct ? (size_t)rm : ((size_t)rb * a.YH + (roy * a.omul + a.ooy)) * a.YW + (rox * a.omul + a.oox);
        r_yoff = (unsigned)((opix * a.YC + a.ycoff + n_base + kq * 4) * 4u);
        rm += BR;                                       // advance to the next chunk's row
        rox += BR;
        while (rox >= a.OW) {
            rox -= a.OW;
            if (++roy == a.OH) { roy = 0; ++rb; }
        }
    };
    auto load_piece = [&](int i) {
        if (i < 4) {
            const int iy = r_iy0 + gdy[i], ix = r_ix0 + gdx[i];
            const bool ok = r_ok && gok[i] && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            const unsigned off = r_pix * (unsigned)(gcs[i] * 4) + tapb[i];
            rx[i] = lwg_wg_buf_load(gsrc[i], gbytes[i], ok ? off : LWG_OOB_OFFSET);
        } else if (i - 4 < NYP) {
            const int j = i - 4;
            const bool ok = r_ok && n_base + (kq + 8 * j) * 4 < a.N;
            ry[j] = lwg_wg_buf_load(dy_, ybytes, ok ? r_yoff + 128u * j : LWG_OOB_OFFSET);
        }
    };
    auto gload = [&]() {
        row_setup();
#pragma unroll
        for (int i = 0; i < 8; ++i) load_piece(i);
    };
    auto store_piece = [&](int buf, int i) {
        if (i < 4) *reinterpret_cast<floatx4*>(Xs + buf * STAGE + mrow * ROW + kq * 4 + i * 32) = rx[i];
        else if (i - 4 < NYP) {
            const int j = i - 4 < NYP ? i - 4 : 0;
            *reinterpret_cast<floatx4*>(Ys + buf * STAGE + mrow * ROW + kq * 4 + j * 32) = ry[j];
            if (do_bias) cs[j] += ry[j];
        }
    };
    auto lstore = [&](int buf) {
        float* xb = Xs + buf * STAGE + mrow * ROW + kq * 4;
        float* yb = Ys + buf * STAGE + mrow * ROW + kq * 4;
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<floatx4*>(xb + g * 32) = rx[g];
#pragma unroll
        for (int j = 0; j < NYP; ++j) {
            *reinterpret_cast<floatx4*>(yb + j * 32) = ry[j];
            if (do_bias) cs[j] += ry[j];
        }
    };

    floatx16 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int khalf = lane >> 5;
    const float* fx = Xs + khalf * ROW + wk * 64 + (lane & 31);
    const float* fy = Ys + khalf * ROW + wn * (BN / 2) + (lane & 31);
    // fragments of reduction step s: one scalar per operand tile (lanes 0-31: row 2s, lanes 32-63: row 2s+1)
    float fa[2][2], fb[2][NT];
    auto read_frags = [&](int buf, int s_, int set) {
        fa[set][0] = fx[buf * STAGE + 2 * s_ * ROW];
        fa[set][1] = fx[buf * STAGE + 2 * s_ * ROW + 32];
#pragma unroll
        for (int j = 0; j < NT; ++j) fb[set][j] = fy[buf * STAGE + 2 * s_ * ROW + 32 * j];
    };
    auto mfma4 = [&](int set) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][i], fb[set][j], acc[i][j], 0, 0, 0);
    };

    if (c_begin < c_end) {
        gload();
        lstore(0);
        __syncthreads();
        read_frags(0, 0, 0);
        // the chunk loop is unrolled by two so the LDS stage is a compile-time constant (base register + immediate)
        auto chunk = [&](auto cur_c, auto next_c) {
            constexpr int CUR = decltype(cur_c)::value;
            constexpr bool NEXT = decltype(next_c)::value;
            if (NEXT) row_setup();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s_ = 0; s_ < BR / 2 - 1; ++s_) {    // fragments of step s+1 in flight during the MFMAs of step s
                read_frags(CUR, s_ + 1, (s_ + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
                mfma4(s_ & 1);
                __builtin_amdgcn_sched_barrier(0);
                if (NEXT && s_ < 4) {                    // loads of the next chunk: two pieces after each of the first steps
                    load_piece(2 * s_);
                    load_piece(2 * s_ + 1);
                }
                if (NEXT && s_ >= 11) {                  // ... and their LDS stores in the shadow of the last steps
                    store_piece(CUR ^ 1, 2 * (s_ - 11));
                    store_piece(CUR ^ 1, 2 * (s_ - 11) + 1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (NEXT) {
                __syncthreads();
                read_frags(CUR ^ 1, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma4(1);                                    // reduction step 15 (fragment set 1), covering the reads above
            __builtin_amdgcn_sched_barrier(0);
        };
        using k0 = std::integral_constant<int, 0>;
        using k1 = std::integral_constant<int, 1>;
        int c = c_begin;
        for (; c + 2 < c_end; c += 2) {
            chunk(k0{}, std::true_type{});
            chunk(k1{}, std::true_type{});
        }
        if (c_end - c == 2) {
            chunk(k0{}, std::true_type{});
            chunk(k1{}, std::false_type{});
        } else {
            chunk(k0{}, std::false_type{});
        }
    }
    // ---- partial tile -> workspace slab blockIdx.y.  Lane owns column n = lane&31, rows k = (r&3) + 8*(r>>2) + 4*khalf ----
    float* slab = part + (size_t)blockIdx.y * (Ktot + 1) * a.N;     // a slab = Ktot weight rows + the bias-gradient row
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = k_base + wk * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
            if (k >= Ktot) continue;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int n = n_base + wn * (BN / 2) + j * 32 + (lane & 31);
                if (n < a.N) slab[(size_t)k * a.N + n] = acc[i][j][r];
            }
        }
    if (do_bias) {   // the 32 rows' partial column sums through LDS, added in row order (deterministic)
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NYP; ++j) *reinterpret_cast<floatx4*>(Ys + mrow * ROW + kq * 4 + j * 32) = cs[j];
        __syncthreads();
        if (tid < BN && n_base + tid < a.N) {
            float t = Ys[tid];
#pragma unroll 8
            for (int r = 1; r < BR; ++r) t += Ys[r * ROW + tid];
            slab[(size_t)Ktot * a.N + n_base + tid] = t;
        }
    }
}

// dW[i] = sum_s part[s][i] in slab order; optionally also the bias gradient db[n] = sum_m dY[m, n] is NOT computed here
// (it is a plain column sum, done by lwg_colsum_nhwc_f32).
__global__ void lwg_wgrad_reduce_kernel(const float* __restrict__ part, int nsplit, size_t total, size_t stride, float* __restrict__ dw) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < nsplit; ++k) s += part[(size_t)k * stride + i];
        dw[i] = s;
    }
}

// Few elements, many slabs (bias column sums: C <= 1024 elements over up to 512 row blocks; the weight gradient of a 1x1 conv:
// 4096 elements over 512 splits): with one thread per element each thread walks all slabs through dependent loads (120 us for
// 512 slabs).  G lanes share an element - lane g adds slabs g, g + G, ... - and a fixed-order LDS pass adds the G partial sums
// (deterministic: the association depends only on (nsplit, G)).
template <int G>
__global__ __launch_bounds__(256) void lwg_slab_reduce_g_kernel(const float* __restrict__ part, int nsplit, size_t total, size_t stride,
                                                                float* __restrict__ out) {
    constexpr int EPB = 256 / G;                                   // elements per block
    __shared__ float sh[256];
    const int e = threadIdx.x % EPB, g = threadIdx.x / EPB;
    const size_t i = (size_t)blockIdx.x * EPB + e;
    float s = 0.f;
    if (i < total)
        for (int k = g; k < nsplit; k += G) s += part[(size_t)k * stride + i];
    sh[threadIdx.x] = s;
    __syncthreads();
    if (g == 0 && i < total) {
        float t = sh[e];
#pragma unroll
        for (int j = 1; j < G; ++j) t += sh[j * EPB + e];
        out[i] = t;
    }
}

// out[i] = sum_k part[k * stride + i], i < total (stride = total for dense slab sets; a weight-gradient slab carries one more row)
static void lwg_launch_slab_reduce(const float* part, int nsplit, size_t total, size_t stride, float* out, hipStream_t stream) {
    if (total < 65536 && nsplit >= 64) {
        hipLaunchKernelGGL(lwg_slab_reduce_g_kernel<16>, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, stream, part, nsplit, total, stride, out);
    } else if (total < 65536 && nsplit >= 16) {
        hipLaunchKernelGGL(lwg_slab_reduce_g_kernel<4>, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, stream, part, nsplit, total, stride, out);
    } else {
        const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        hipLaunchKernelGGL(lwg_wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, stream, part, nsplit, total, stride, out);
    }
}

// The slab reduction of a weight gradient fused with its un-packing: element i of the OUTPUT (tap, c, n; n fastest) is the sum
// over the slices of slab[k(tap, c)][n] and lands at the parameter tensor's own position (nn.Conv2d (N, Cin, KH, KW), or
// (Cin, N, KH, KW) for transposed = 1) - what lwg_wgrad_reduce_kernel + lwg_unpack_wgrad_f32 did in two launches and one extra
// pass over dW (~150 launch pairs per training step).
struct LwgUnpackMap {
    int D1, KHW, transposed, ntaps, cin, cin_pad, nout, n_pad;
    int kidx[LWG_MAX_TAPS];
};

// element i of the output -> its slab offset (src) and parameter-tensor offset (dst).  Elements [total_w, total_w + nout) are the bias
// gradient (slab row ntaps * cin_pad, returned with bias = true and dst = n) when the launch carries one.
__device__ __forceinline__ void lwg_unpack_map(const LwgUnpackMap& u, size_t i, size_t& src, size_t& dst, bool& bias) {
    const int n = (int)(i % u.nout), r = (int)(i / u.nout);
    bias = r >= u.ntaps * u.cin;
    if (bias) {
        src = (size_t)u.ntaps * u.cin_pad * u.n_pad + n;
        dst = (size_t)n;
        return;
    }
    const int c = r % u.cin, tap = r / u.cin;
    const int k = (u.cin_pad & 31) == 0 ? ((c >> 5) * u.ntaps + tap) * 32 + (c & 31) : tap * u.cin_pad + c;
    src = (size_t)k * u.n_pad + n;
    dst = ((size_t)(u.transposed ? c : n) * u.D1 + (u.transposed ? n : c)) * u.KHW + u.kidx[tap];
}

template <int G>
__global__ __launch_bounds__(256) void lwg_slab_reduce_unpack_kernel(const float* __restrict__ part, int nsplit, size_t slab, size_t total,
                                                                     const LwgUnpackMap u, float* __restrict__ out, float* __restrict__ db) {
    constexpr int EPB = 256 / G;
    __shared__ float sh[256];
    const int e = threadIdx.x % EPB, g = threadIdx.x / EPB;
    for (size_t i0 = (size_t)blockIdx.x * EPB; i0 < total; i0 += (size_t)gridDim.x * EPB) {
        const size_t i = i0 + e;
        size_t src = 0, dst = 0;
        bool bias = false;
        float s = 0.f;
        if (i < total) {
            lwg_unpack_map(u, i, src, dst, bias);
            for (int k = g; k < nsplit; k += G) s += part[(size_t)k * slab + src];
        }
        float* o = bias ? db : out;
        if (G == 1) {
            if (i < total) o[dst] = s;
        } else {
            __syncthreads();
            sh[threadIdx.x] = s;
            __syncthreads();
            if (g == 0 && i < total) {
                float t = sh[e];
#pragma unroll
                for (int j = 1; j < G; ++j) t += sh[j * EPB + e];
                o[dst] = t;
            }
        }
    }
}

// The large-tensor form (G = 1 above keeps ONE 4-byte load in flight per thread while it walks the slabs: 17 us for a 33 MB slab set that
// HBM streams in 7).  Here a thread owns FOUR consecutive n of one (tap, c) - their slab entries are 16 contiguous bytes - and the slab
// loop is unrolled by four (independent loads in flight; the sums are still added in slab order: deterministic).
__global__ __launch_bounds__(256) void lwg_slab_reduce_unpack4_kernel(const float* __restrict__ part, int nsplit, size_t slab, size_t total4,
                                                                      const LwgUnpackMap u, float* __restrict__ out, float* __restrict__ db) {
    const int nq = u.nout >> 2;
    for (size_t i4 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i4 < total4; i4 += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i4 % nq) * 4, r = (int)(i4 / nq);
        const bool bias = r >= u.ntaps * u.cin;              // the quads past the weight elements: the bias-gradient row of the slabs
        const int c = r % u.cin, tap = bias ? 0 : r / u.cin;
        const int k = bias ? u.ntaps * u.cin_pad : ((u.cin_pad & 31) == 0 ? ((c >> 5) * u.ntaps + tap) * 32 + (c & 31) : tap * u.cin_pad + c);
        const float* src = part + (size_t)k * u.n_pad + n;
        floatx4 s = {0.f, 0.f, 0.f, 0.f};
        int kk = 0;
        for (; kk + 4 <= nsplit; kk += 4) {
            const floatx4 v0 = *reinterpret_cast<const floatx4*>(src + (size_t)kk * slab);
            const floatx4 v1 = *reinterpret_cast<const floatx4*>(src + (size_t)(kk + 1) * slab);
            const floatx4 v2 = *reinterpret_cast<const floatx4*>(src + (size_t)(kk + 2) * slab);
            const floatx4 v3 = *reinterpret_cast<const floatx4*>(src + (size_t)(kk + 3) * slab);
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; kk < nsplit; ++kk) s += *reinterpret_cast<const floatx4*>(src + (size_t)kk * slab);
        if (bias) {
            *reinterpret_cast<floatx4*>(db + n) = s;
            continue;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const size_t dst = ((size_t)(u.transposed ? c : n + j) * u.D1 + (u.transposed ? n + j : c)) * u.KHW + u.kidx[tap];
            out[dst] = s[j];
        }
    }
}

// The same with G lanes per quad (lane g adds slabs g, g + G, ...; fixed-order LDS pass over the G partial sums): medium-sized gradients
// reduced over many slabs (a transposed convolution's parity launch: 32 K quads x 64 - 128 slabs) gave the one-lane form 64 - 128
// workgroups, each thread walking every slab (8 us); G = 4 / 8 puts 4 - 8x the loads in flight.
template <int G>
__global__ __launch_bounds__(256) void lwg_slab_reduce_unpack4g_kernel(const float* __restrict__ part, int nsplit, size_t slab, size_t total4,
                                                                       const LwgUnpackMap u, float* __restrict__ out, float* __restrict__ db) {
    constexpr int EPB = 256 / G;
    __shared__ floatx4 sh[256];
    const int e = threadIdx.x % EPB, g = threadIdx.x / EPB;
    const int nq = u.nout >> 2;
    const size_t i4 = (size_t)blockIdx.x * EPB + e;
    const bool live = i4 < total4;
    const int n = live ? (int)(i4 % nq) * 4 : 0, r = live ? (int)(i4 / nq) : 0;
    const bool bias = r >= u.ntaps * u.cin;
    const int c = r % u.cin, tap = bias ? 0 : r / u.cin;
    const int k = bias ? u.ntaps * u.cin_pad : ((u.cin_pad & 31) == 0 ? ((c >> 5) * u.ntaps + tap) * 32 + (c & 31) : tap * u.cin_pad + c);
    const float* src = part + (size_t)k * u.n_pad + n;
    floatx4 s = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        int kk = g;
        for (; kk + G < nsplit; kk += 2 * G) {
            const floatx4 v0 = *reinterpret_cast<const floatx4*>(src + (size_t)kk * slab);
            const floatx4 v1 = *reinterpret_cast<const floatx4*>(src + (size_t)(kk + G) * slab);
            s += v0; s += v1;
        }
        if (kk < nsplit) s += *reinterpret_cast<const floatx4*>(src + (size_t)kk * slab);
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    if (g != 0 || !live) return;
    floatx4 t = sh[e];
#pragma unroll
    for (int j = 1; j < G; ++j) t += sh[j * EPB + e];
    if (bias) {
        *reinterpret_cast<floatx4*>(db + n) = t;
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const size_t dst = ((size_t)(u.transposed ? c : n + j) * u.D1 + (u.transposed ? n + j : c)) * u.KHW + u.kidx[tap];
        out[dst] = t[j];
    }
}

// total: weight elements (ntaps * cin * nout); db != NULL appends the nout bias-gradient elements (slab row Ktot) to the same launch
static void lwg_launch_slab_reduce_unpack(const float* part, int nsplit, size_t slab, size_t total, const LwgUnpackMap& u, float* out, float* db,
                                          hipStream_t stream) {
    const size_t weights = total;
    if (db) total += (size_t)u.nout;
    if (weights >= 65536 && (u.nout & 3) == 0 && (u.n_pad & 3) == 0 && (slab & 3) == 0) {
        const size_t total4 = total / 4;
        if (total4 < 65536 && nsplit >= 32) {        // few quads, many slabs: several lanes per quad
            if (nsplit >= 64)
                hipLaunchKernelGGL(lwg_slab_reduce_unpack4g_kernel<8>, dim3((unsigned)((total4 + 31) / 32)), dim3(256), 0, stream, part, nsplit, slab, total4, u, out, db);
            else
                hipLaunchKernelGGL(lwg_slab_reduce_unpack4g_kernel<4>, dim3((unsigned)((total4 + 63) / 64)), dim3(256), 0, stream, part, nsplit, slab, total4, u, out, db);
            return;
        }
        const unsigned blocks = (unsigned)((total4 + 255) / 256 < 4096 ? (total4 + 255) / 256 : 4096);
        hipLaunchKernelGGL(lwg_slab_reduce_unpack4_kernel, dim3(blocks), dim3(256), 0, stream, part, nsplit, slab, total4, u, out, db);
        return;
    }
    if (weights < 65536 && nsplit >= 64) {
        hipLaunchKernelGGL(lwg_slab_reduce_unpack_kernel<16>, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, stream, part, nsplit, slab, total, u, out, db);
    } else if (weights < 65536 && nsplit >= 16) {
        hipLaunchKernelGGL(lwg_slab_reduce_unpack_kernel<4>, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, stream, part, nsplit, slab, total, u, out, db);
    } else {
        const unsigned blocks = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        hipLaunchKernelGGL(lwg_slab_reduce_unpack_kernel<1>, dim3(blocks), dim3(256), 0, stream, part, nsplit, slab, total, u, out, db);
    }
}

// Column sums of an NHWC tensor viewed as (rows, C): out[c] = sum_r x[r, c] (bias gradients).  Two deterministic passes.
// C % 4 == 0: a lane owns four consecutive channels (16-byte loads), C/4 lanes cover a row, 256 / (C/4) rows per pass;
// up to 512 row blocks keep every CU streaming (the first version used 64 blocks of 4-byte loads: 2.4 TB/s).
__global__ __launch_bounds__(256) void lwg_colsum_partial4_kernel(const float* __restrict__ x, size_t rows, int C, int rows_per_block,
                                                                 float* __restrict__ ws) {
    const int C4 = C >> 2;
    const int lanes_per_row = C4 < 256 ? C4 : 256;          // C <= 1024
    const int rpp = 256 / lanes_per_row;                     // rows per pass of the block
    const int cq = threadIdx.x % lanes_per_row, rr = threadIdx.x / lanes_per_row;
    const size_t r0 = (size_t)blockIdx.x * rows_per_block, r1 = min(rows, r0 + (size_t)rows_per_block);
    floatx4 s = {0.f, 0.f, 0.f, 0.f};
    if (rr < rpp)
        for (size_t r = r0 + rr; r < r1; r += rpp) {
            const floatx4 v = *reinterpret_cast<const floatx4*>(x + r * C + 4 * cq);
#pragma unroll
            for (int k = 0; k < 4; ++k) s[k] += v[k];
        }
    __shared__ floatx4 sh[256];
    sh[threadIdx.x] = s;
    __syncthreads();
    if (rr == 0) {
        for (int j = 1; j < rpp; ++j) {
            const floatx4 v = sh[j * lanes_per_row + cq];
#pragma unroll
            for (int k = 0; k < 4; ++k) s[k] += v[k];
        }
        *reinterpret_cast<floatx4*>(ws + (size_t)blockIdx.x * C + 4 * cq) = s;
    }
}

__global__ __launch_bounds__(256) void lwg_colsum_partial_kernel(const float* __restrict__ x, size_t rows, int C, int rows_per_block,
                                                                float* __restrict__ ws) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
    const size_t r0 = (size_t)blockIdx.y * rows_per_block, r1 = min(rows, r0 + (size_t)rows_per_block);
    float s = 0.f;
    if (c < C)
        for (size_t r = r0 + w; r < r1; r += 4) s += x[r * C + c];
    __shared__ float sh[4][64];
    sh[w][threadIdx.x & 63] = s;
    __syncthreads();
    if (w == 0 && c < C) ws[(size_t)blockIdx.y * C + c] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

// Reduction splits: enough workgroups to fill the 256 CUs twice (2 workgroups/CU), at most 48 MB of slabs (every slab is
// written and read once more by the reduction), at least 4 chunks of 32 rows per workgroup.
static int lwg_wgrad_bn(int N) { return (N % 128 != 0 && N % 128 <= 64) ? 64 : 128; }    // 64-column tiles when the last 128 would be half empty

static int lwg_wgrad_splits(int Ktot, int N, int M) {
    const int bn = lwg_wgrad_bn(N);
    const int tiles = ((Ktot + 127) / 128) * ((N + bn - 1) / bn);
    const int nchunks = (M + 31) / 32;
    int splits = 512 / tiles;          // tiles * splits <= 512 = one full wave of workgroups at 2 per CU (no ragged second wave)
    const long slab = (long)(Ktot + 1) * N * 4;
    if ((long)splits * slab > (48l << 20)) splits = (int)((48l << 20) / slab);     // <= 48 MB of slabs per launch
    if (splits > nchunks / 4) splits = nchunks / 4;
    if (splits < 1) splits = 1;
    const int cps = (nchunks + splits - 1) / splits;          // chunks per workgroup; drop the splits that would get none
    return (nchunks + cps - 1) / cps;                          // (128 chunks over 14 splits: 10 each -> 13 workgroups have work)
}

extern "C" size_t lwg_conv2d_wgrad_ws_floats(int Ktot, int N, int M) {
    return (size_t)lwg_wgrad_splits(Ktot, N, M) * (size_t)(Ktot + 1) * N;     // a slab = Ktot weight rows + the bias-gradient row
}

// Validation + the slab launch shared by the two entry points; *splits_out slices of (Ktot, N) land in ws.
static int lwg_wgrad_launch(const LwgConvArgs* pa, const float* dy, float* ws, hipStream_t stream, int* splits_out, int want_bias = 0) {
    if (!pa || !dy || !ws) return (int)hipErrorInvalidValue;
    const LwgConvArgs& a = *pa;
    const int Cin = a.C0 + a.C1;
    if (!a.x0 || a.ntaps < 1 || a.ntaps > LWG_MAX_TAPS || a.M <= 0 || a.N <= 0 || (a.N & 3) || (Cin & 3) || (a.YC & 3) || (a.ycoff & 3))
        return (int)hipErrorInvalidValue;
    const unsigned long long pix = (unsigned long long)a.B * a.H * a.W;
    if (pix * (unsigned long long)(a.C0 > a.C1 ? a.C0 : a.C1) * 4ull >= 0xC0000000ull) return (int)hipErrorInvalidValue;
    if ((unsigned long long)a.B * a.YH * a.YW * a.YC * 4ull >= 0xC0000000ull) return (int)hipErrorInvalidValue;
    const bool smallc = (Cin % 32) != 0;
    if (smallc && (a.C1 != 0 || Cin > 16 || (Cin & (Cin - 1)) != 0 || (1 << a.cshift) != (Cin >> 2))) return (int)hipErrorInvalidValue;
    if (!smallc && a.C1 != 0 && (a.C0 % 32 != 0 || !a.x1)) return (int)hipErrorInvalidValue;
    const int Ktot = a.ntaps * Cin;
    const int bn = lwg_wgrad_bn(a.N);
    const int tiles = ((Ktot + 127) / 128) * ((a.N + bn - 1) / bn);
    const int nchunks = (a.M + 31) / 32;
    const int splits = lwg_wgrad_splits(Ktot, a.N, a.M);
    const int cps = (nchunks + splits - 1) / splits;
    const size_t lds = (size_t)4 * 32 * 128 * sizeof(float) + LWG_MAX_TAPS * sizeof(int);
    auto kern = bn == 64 ? (smallc ? lwg_conv_wgrad_kernel<true, 64> : lwg_conv_wgrad_kernel<false, 64>)
                         : (smallc ? lwg_conv_wgrad_kernel<true, 128> : lwg_conv_wgrad_kernel<false, 128>);
    static unsigned long long attr_done[4] = {0ull, 0ull, 0ull, 0ull};
    if (hipError_t e = lwg_allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds, attr_done[(bn == 64 ? 2 : 0) + (smallc ? 1 : 0)]); e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(kern, dim3(tiles, splits), dim3(256), lds, stream, a, dy, Ktot, cps, ws, want_bias);
    *splits_out = splits;
    return 0;
}

// args: forward geometry; dy: gradient of the forward output (same layout as y); dw: (ntaps*Cin, N) row-major in the
// forward panel's K order.  ws: lwg_conv2d_wgrad_ws_floats(ntaps*Cin, N, M) floats.
extern "C" int lwg_conv2d_wgrad_nhwc_f32(const LwgConvArgs* pa, const float* dy, float* dw, float* ws, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!dw) return (int)hipErrorInvalidValue;
    int splits = 0;
    if (int e = lwg_wgrad_launch(pa, dy, ws, stream, &splits); e != 0) return e;
    const size_t total = (size_t)pa->ntaps * (pa->C0 + pa->C1) * pa->N;
    lwg_launch_slab_reduce(ws, splits, total, total + (size_t)pa->N, dw, stream);
    return (int)hipGetLastError();
}

// The same weight gradient delivered in the parameter's own layout: dw is the (D0, D1, KH, KW) gradient tensor of an
// nn.Conv2d (transposed = 0: (N, Cin, KH, KW)) or of the GEMM-transposed forms (transposed = 1: (Cin-of-the-GEMM, N, KH, KW)),
// tap t of the launch goes to kernel position kidx[t]; cin <= C0 + C1 and nout <= N drop the zero-extended channels.  Positions no
// tap maps to are left untouched (the four parity launches of a transposed convolution fill one tensor).
// db: NULL, or nout floats that receive the bias gradient sum_m dY[m, n] of THIS launch's rows (the column sums ride along in the
// workgroups of k-tile 0 and the same reduction launch; nout % 4 == 0 is not required).
extern "C" int lwg_conv2d_wgrad_unpacked_f32(const LwgConvArgs* pa, const float* dy, float* ws, float* dw, int D0, int D1, int KH, int KW,
                                             int transposed, const int* kidx, int cin, int nout, float* db, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!pa || !dw || !kidx || cin < 1 || nout < 1 || cin > pa->C0 + pa->C1 || nout > pa->N) return (int)hipErrorInvalidValue;
    if ((transposed ? cin : nout) > D0 || (transposed ? nout : cin) > D1 || pa->ntaps < 1 || pa->ntaps > LWG_MAX_TAPS) return (int)hipErrorInvalidValue;
    LwgUnpackMap u;
    u.D1 = D1; u.KHW = KH * KW; u.transposed = transposed; u.ntaps = pa->ntaps; u.cin = cin; u.cin_pad = pa->C0 + pa->C1;
    u.nout = nout; u.n_pad = pa->N;
    for (int i = 0; i < pa->ntaps; ++i) {
        if (kidx[i] < 0 || kidx[i] >= KH * KW) return (int)hipErrorInvalidValue;
        u.kidx[i] = kidx[i];
    }
    int splits = 0;
    if (int e = lwg_wgrad_launch(pa, dy, ws, stream, &splits, db ? 1 : 0); e != 0) return e;
    lwg_launch_slab_reduce_unpack(ws, splits, ((size_t)u.ntaps * u.cin_pad + 1) * u.n_pad, (size_t)u.ntaps * cin * nout, u, dw, db, stream);
    return (int)hipGetLastError();
}

// out[c] = sum over rows of x (rows, C); ws: 64 * C floats.
extern "C" int lwg_colsum_nhwc_f32(const float* x, size_t rows, int C, float* out, float* ws, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!x || !out || !ws || rows == 0 || C <= 0) return (int)hipErrorInvalidValue;
    if ((C & 3) == 0 && C <= 1024 && (256 % (C / 4 < 256 ? C / 4 : 256)) == 0) {
        const int nblk = rows >= 512 * 64 ? 512 : (int)((rows + 63) / 64);      // ws holds nblk * C floats (callers size it 512 * C)
        const int rpb = (int)((rows + nblk - 1) / nblk);
        hipLaunchKernelGGL(lwg_colsum_partial4_kernel, dim3(nblk), dim3(256), 0, stream, x, rows, C, rpb, ws);
        lwg_launch_slab_reduce(ws, nblk, (size_t)C, (size_t)C, out, stream);
        return (int)hipGetLastError();
    }
    const int nblk = rows >= 64 * 64 ? 64 : (int)((rows + 63) / 64);
    const int rpb = (int)((rows + nblk - 1) / nblk);
    hipLaunchKernelGGL(lwg_colsum_partial_kernel, dim3((C + 63) / 64, nblk), dim3(256), 0, stream, x, rows, C, rpb, ws);
    lwg_launch_slab_reduce(ws, nblk, (size_t)C, (size_t)C, out, stream);
    return (int)hipGetLastError();
}
