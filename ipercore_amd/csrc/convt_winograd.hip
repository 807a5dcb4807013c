// nn.ConvTranspose2d(kernel 4, stride 2, padding 1) in fp32 as a fused F(2x2, 2x2) Winograd convolution on v_mfma_f32_32x32x2_f32 - the sibling of
// conv_winograd.hip for the decoders' up-sampling layers (attlwb_spade_resunet.py:331-340).
//   Output parity (py, px) of the layer is a 2 x 2-tap convolution of the input: y[2i + py][2j + px] = sum_{r,q in {0,1}} x[i + py - 1 + r][j + px - 1 + q]
//   g_p[r][q] with g_p[r][q] = w[3 - py - 2 r][3 - px - 2 q].  Two adjacent outputs of one parity per dimension from three inputs with three
//   multiplies instead of four: m0 = (d0 - d1) g0, m1 = d1 (g0 + g1), m2 = (d2 - d1) g1, y0 = m0 + m1, y1 = m1 + m2 - in two dimensions 9 products
//   per 2 x 2 outputs of a parity instead of 16, 36 per 4 x 4 input patch instead of 64.  The patch of parity (py, px) is rows py .. py + 2, columns
//   px .. px + 2 of the SAME 4 x 4 input patch (rows i - 1 .. i + 2) the 3 x 3 kernel stages, so the halo staging is that kernel's.  Row forms of the
//   patch: R0 = r0 - r1, R1 = r1, R2 = r2 - r1, R3 = r2, R4 = r3 - r2 (parity 0 uses R0 R1 R2, parity 1 uses -R2 R3 R4: the sign lives in the
//   weight panel), the same over columns: 25 transformed values per (patch, input channel) serve all 36 products.
// Workgroup: 512 threads = 8 waves; block = 8 x 8 patches (16 x 16 input pixels -> 32 x 32 output pixels) x 32 output channels; wave w owns ALL NINE
// products of parity w % 4 for the 32 patches of tile w / 4 (9 accumulator tiles of 32 x 32 = 144 VGPRs, rows = output channels): the output
// transform A^T M A is register-local; the block's outputs then cross LDS once so that the global stores are whole 128-byte (NHWC) / 512-byte
// (channel-quad planes) runs.
// A K stage is 8 input channels = four k-pairs; per k-pair a lane loads its nine weights as two 16-byte and one 4-byte buffer loads from the
// panel Upk[4][Cin/8][4][2][9 N] (contiguous per load instruction; two k-pairs ahead, four register sets) and reads nine V fragments (4 bytes each) from LDS.  The raw 18 x 18 x 8
// halo goes global -> registers (three stages ahead) -> raw[s % 2]; every thread transforms ONE (patch, channel) 4 x 4 -> 25 values (27 subtractions)
// from raw[(s + 1) % 2] into Vs[(s + 1) % 2] beside the MFMAs of k-pairs 0 and 1; one barrier per stage, in the middle of k-pair 3.
// Rounding: the transforms only add / subtract (no 1/2 factors as in F(2x2, 3x3)); panel entries are sums of up to four weights formed in fp64
// and rounded once.  fp32-grade, NOT the direct kernel's bits: part of the "winograd" precision mode.
#include <hip/hip_runtime.h>
#include "lwg_common.h"
#include "lwg_conv_args.h"

#ifndef CTW_NT_ST
#define CTW_NT_ST 0          // cache policy of the output stores: 0 = default; 2 = non-temporal - worth 0.5 % in conv_winograd4.hip, measured neutral here (profiles/r06_az_*)
#endif
#ifndef LWG_CTW_XCD
#define LWG_CTW_XCD 1        // XCD-aware block order (see the kernel): 0 = column-block-major (lab)
#endif
#define WG_THREADS 512
#define TPB 8            // patches per block edge: 8 x 8 patches = 16 x 16 input pixels
#define NPATCH 64
#define NBT 32           // output channels per block
#define KS 8             // input channels per stage
#define HALO 18
#define PLANE (HALO * HALO)
#define RAW_FLOATS (KS * PLANE)              // [c][py][px]
#define VSTR 64
#define NFORM 25
#define VS_FLOATS (NFORM * KS * VSTR)        // [form][k][patch]
#define DUMP_OFF (2 * RAW_FLOATS + 2 * VS_FLOATS)                 // where the threads without a halo element store their zeros (dead LDS)
#define DUMP_FLOATS (WG_THREADS + 3 * PLANE + RAW_FLOATS)
#define LOOP_FLOATS (DUMP_OFF + DUMP_FLOATS)
#define OROW 36                              // floats per pixel row of the epilogue's exchange buffer [32 x 32 output pixels][32 channels + 4]
#define OUT_FLOATS (32 * 32 * OROW)
#define BIAS_OFF (LOOP_FLOATS > OUT_FLOATS ? LOOP_FLOATS : OUT_FLOATS)    // the block's 32 bias values, behind both uses of the LDS
#define WINO_OOB 0xC0000000u                 // >= any image's byte size (host: H * W * C * 4 < 3 GiB): the buffer load returns 0
// slot plan of the K loop (profiles/r05_h_convt_winograd_lab.txt, 64 frames, all three layers): first plan (patch reads and transform in one slot each,
// barrier behind k-pair 2) 0.622 of the pipe; barrier in front of k-pair 2 0.644; barrier in the middle of k-pair 3 0.646; that + the patch reads and the
// transform a few instructions per slot 0.650 = the product's plan.  -DCTW_LAB_BASE builds the first plan.
#ifndef CTW_LAB_BASE
#define CTW_BAR_LATE
#define CTW_SPREAD
#endif
#ifdef CTW_NOWSB                             // lab: leave the order of the loop's instructions to the compiler
#define WSB() do { } while (0)
#else
#define WSB() __builtin_amdgcn_sched_barrier(0)
#endif


template <int V> struct IntT { static constexpr int value = V; };

// lab instrumentation (compiled out of the product): tools/up4lab.py --ts on a -DLWG_CTW_TS variant library - every wave stamps kernel entry, K-loop
// entry, K-loop exit and its end into args->res (four 64-bit stamps per wave)
#ifdef LWG_CTW_TS
#define CTS(i) do { if (lane == 0) reinterpret_cast<unsigned long long*>(const_cast<float*>(a.res))[(((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + wid) * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#define CTSB(k, i) do { if (lab_bi == (k)) CTS(i); } while (0)     // the workgroup's k-th block only (persistent form: block 1 = steady state)
#define CTS_COUNT() ++lab_bi
#else
#define CTS(i) do { } while (0)
#define CTSB(k, i) do { } while (0)
#define CTS_COUNT() do { } while (0)
#endif

// Where pixel column lx of a row of the epilogue's exchange buffer lies (rows of 32 pixel slots x 36 floats).  A write instruction's eight consecutive
// lanes are the patches etx = 0..7 of one patch row - pixels 4 etx + r, r = 2 ib + px fixed: in column order their 16-byte pieces lie 144 floats apart,
// i.e. on TWO of the eight bank quads of a ds_write_b128 (four-way conflicts on every write; PMC: 22 % of the kernel's LDS-array cycles).  Slot
// 8 r + (etx + 2 r) % 8 puts the eight lanes on eight bank quads, and the sixteen lanes of either ds_read_b128 lane group of the channel-quad-plane reader
// (32 consecutive pixels: lanes {0-3, 12-15, 20-27} | {4-11, 16-19, 28-31}) on sixteen distinct slots mod 16 (36 floats = 9 quads per slot: the bank quad
// of a slot is 9 slot % 16, a bijection) - checked by enumeration in tests/test_bf16_panels.py::test_convt_exchange_slots.
__device__ __forceinline__ int ctw_slot(int lx) {
#ifdef CTW_LAB_LINEAR_SLOTS
    return lx;
#else
    const int r = lx & 3;
    return 8 * r + (((lx >> 2) + 2 * r) & 7);
#endif
}

__device__ __forceinline__ floatx4 ctw_buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}

__global__ __launch_bounds__(WG_THREADS, 1) void lwg_convt_winograd_kernel(const LwgConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int H = a.H, W = a.W, Cin = a.C0, N = a.N;
    float* const raw0 = smem;                                // [2][RAW], then [2][VS] (25 planes of [k][patch])
    const int tid = threadIdx.x, lane = tid & 63;
    const int bx = (W + 2 * TPB - 1) / (2 * TPB), by = (H + 2 * TPB - 1) / (2 * TPB);
    // persistent workgroups (round 6, as conv_winograd.hip): min(blocks, CUs) workgroups walk the block ids blockIdx.x + k gridDim.x (id = column block *
    // tiles + tile); the next block's first halo stages and weights are requested inside this block's epilogue.  Bitwise the one-block-per-workgroup results.
    const int tiles = bx * by * a.B;
    const int total = tiles * (N / NBT);
    int blk = blockIdx.x;
    const int nst = Cin / KS;                                // even (host: Cin % 16 == 0)
    CTS(0);
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, (int)(144u * (unsigned)Cin * (unsigned)N), 0x00020000);
    const int par = wid & 3, py = par >> 1, px = par & 1;    // this wave's output parity
    const int pt = wid >> 2;                                  // ... and its 32-patch tile
    floatx16 acc[9];                                         // [product 3 xi + nu]
    // ---- per-block state (workgroup-uniform): image b, tile corner (x0, y0), first output column n0, this image as a buffer; this thread's two halo
    // elements (pixel, channel quad): byte offset of the pixel inside the image (out of range: padding / none); this lane's column of the panel
    int b, x0, y0, n0;
    __amdgpu_buffer_rsrc_t rx0;
    unsigned voff0[2];
    unsigned uvoff, uvoffc;                                  // this lane's column of the panel: the 16-byte parts, the ninth product
    // XCD-aware block order (as conv_winograd4.hip; persistent grids of a multiple of 8 workgroups, N / 32 = 2, 4 or 8 column blocks): workgroup w runs on
    // XCD w % 8 and keeps ONE column block, (w % 8) % ncb, for the whole launch (an XCD's L2 holds that column block's panel only), while the ncb
    // workgroups (w % 8) / ncb, w / 8 of adjacent XCDs walk the same tile sequence in step: a tile's halo comes from HBM once instead of ncb times
    const int ncb = N / NBT;
    const bool xcd = LWG_CTW_XCD && (gridDim.x & 7u) == 0 && (ncb == 2 || ncb == 4 || ncb == 8) && (int)gridDim.x < total && tiles >= (int)gridDim.x / ncb;
    const int xg = (int)gridDim.x / ncb;                     // workgroups per column block = tiles per round
    const int xr = (int)(((blockIdx.x & 7u) / (unsigned)ncb) * (gridDim.x >> 3) + (blockIdx.x >> 3));      // this workgroup's place among them
    auto has_block = [&](int id) -> bool {                   // (id = blockIdx.x + k gridDim.x)
        return xcd ? (id / (int)gridDim.x) * xg + xr < tiles : id < total;
    };
    auto setup = [&](int id) {
        int cb, t;
        if (xcd) {
            cb = (int)(blockIdx.x & 7u) & (ncb - 1);
            t = __builtin_amdgcn_readfirstlane((id / (int)gridDim.x) * xg + xr);
        } else {
            cb = __builtin_amdgcn_readfirstlane(id / tiles);
            t = __builtin_amdgcn_readfirstlane(id - cb * tiles);
        }
        b = __builtin_amdgcn_readfirstlane(t / (bx * by));
        t -= b * bx * by;
        x0 = (t % bx) * 2 * TPB;
        y0 = (t / bx) * 2 * TPB;
        n0 = cb * NBT;
        rx0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x0 + (size_t)b * H * W * Cin), 0, (int)((unsigned)(H * W) * (unsigned)Cin * 4u), 0x00020000);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = tid + WG_THREADS * q;
            const int pix = i >> 1, half = i & 1, hy = pix / HALO, hx = pix - hy * HALO;
            const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
            const bool in = i < PLANE * 2 && gy >= 0 && gy < H && gx >= 0 && gx < W;
            voff0[q] = in ? (unsigned)((gy * W + gx) * Cin + 4 * half) * 4u : WINO_OOB;
        }
        uvoff = (unsigned)(((lane >> 5) * 9 * N + 4 * (n0 + (lane & 31))) * 4);
        uvoffc = (unsigned)(((lane >> 5) * 9 * N + 8 * N + n0 + (lane & 31)) * 4);
    };
    setup(blk);
    int wst[2];                                              // the halo elements' LDS slot (threads without one store their zeros into dead LDS)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = tid + WG_THREADS * q;
        wst[q] = i < PLANE * 2 ? 4 * (i & 1) * PLANE + (i >> 1) : DUMP_OFF + tid;
    }
    floatx4 rreg[2];
    auto rld1 = [&](int st, int q) -> floatx4 { return ctw_buf_load(rx0, voff0[q], (unsigned)(st * KS) * 4u); };
    auto rst1 = [&](int buf, int q, floatx4 v) {
        float* dst = raw0 + buf * RAW_FLOATS + wst[q];
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[k * PLANE] = v[k];
    };
    // weights: lane = (k-half lane / 32, channel lane % 32); element (parity, stage, k-pair, k-half) = 9 N floats: [N][4] products 0-3, [N][4] products 4-7,
    // [N] product 8 - every load instruction reads contiguous memory (as conv_winograd4.hip; the first layout, [N][12] with three of padding, made a
    // half-wave's 16-byte load span 1.5 KB for 512 useful bytes)
    floatx4 ufr[4][3];                                       // [register set = k-pair][products 0-3 | 4-7 | 8 (element 0)]: loaded TWO k-pairs ahead
    const unsigned ukk = (unsigned)N * 72u;                  // bytes between two k-pairs: [2][9 N] floats
    const unsigned upar = (unsigned)par * (unsigned)nst * 4u * ukk;
    const unsigned ubo = (unsigned)N * 16u;                  // bytes from part A to part B
    auto uld1 = [&](int st, int kk, int j) -> floatx4 {
        const unsigned so = upar + (unsigned)(st * 4 + kk) * ukk;
        if (j < 2) return ctw_buf_load(ru, uvoff, so + (j ? ubo : 0u));
        floatx4 r;
        r[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ru, (int)uvoffc, (int)so, 0));
        r[1] = r[2] = r[3] = 0.f;
        return r;
    };
    const int patch = tid & 63, tc = tid >> 6;
    const int pty = patch >> 3, ptx = patch & 7;
    unsigned dbs[2];                                         // this thread's 4 x 4 input patch inside raw[u] (float index into smem)
    unsigned vbs[2];                                         // ... its 25 transformed values inside Vs[u]
    unsigned fbs[2];                                         // this lane's fragments inside Vs[u]: form (2 py + xi, 2 px + nu), k-half, patch
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        dbs[u] = (unsigned)(u * RAW_FLOATS + tc * PLANE + (2 * pty) * HALO + 2 * ptx) >> 1;
        asm volatile("" : "+v"(dbs[u]));
        dbs[u] <<= 1;
        vbs[u] = (unsigned)(2 * RAW_FLOATS + u * VS_FLOATS + tc * VSTR + patch);
        asm volatile("" : "+v"(vbs[u]));
        fbs[u] = (unsigned)(2 * RAW_FLOATS + u * VS_FLOATS + (10 * py + 2 * px) * KS * VSTR + (lane >> 5) * VSTR + pt * 32 + (lane & 31));
        asm volatile("" : "+v"(fbs[u]));
    }
    float fb[2][9];                                          // [register set = k-pair % 2][product]
    auto fragread = [&](int buf, int kk) {                    // the nine fragments of k-pair kk of the stage in Vs[buf] -> set kk % 2
#pragma unroll
        for (int xi = 0; xi < 3; ++xi)
#pragma unroll
            for (int nu = 0; nu < 3; ++nu) fb[kk & 1][3 * xi + nu] = smem[fbs[buf] + ((5 * xi + nu) * KS + 2 * kk) * VSTR];
    };
    auto transform_store = [&](int buf, const float (&dd)[4][4]) {
        float t[5][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t[0][j] = dd[0][j] - dd[1][j];
            t[1][j] = dd[1][j];
            t[2][j] = dd[2][j] - dd[1][j];
            t[3][j] = dd[2][j];
            t[4][j] = dd[3][j] - dd[2][j];
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            float* v = smem + vbs[buf] + (5 * i) * KS * VSTR;
            v[0 * KS * VSTR] = t[i][0] - t[i][1];
            v[1 * KS * VSTR] = t[i][1];
            v[2 * KS * VSTR] = t[i][2] - t[i][1];
            v[3 * KS * VSTR] = t[i][2];
            v[4 * KS * VSTR] = t[i][3] - t[i][2];
        }
    };
    auto iteration = [&](int s, auto SET, auto NXT) {
        constexpr int set = decltype(SET)::value;            // s % 2
        constexpr bool nxt = decltype(NXT)::value != 0;      // the last stage has no next one to prepare (peeled: no branches in the loop)
        const int s3 = s + 3 < nst ? s + 3 : nst - 1;        // past the end: a harmless re-load of the last stage (its halo store lands in a dead buffer)
        float dd[4][4];
        auto mf = [&](int kk, int q) {
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(ufr[kk][q >> 2][q & 3], fb[kk & 1][q], acc[q], 0, 0, 0);
            WSB();
        };
        auto uldn = [&](int kk) {                            // the weights of the k-pair after the next (kk + 2 of this stage, or kk - 2 of the next)
            if (kk < 2) {
#pragma unroll
                for (int j = 0; j < 3; ++j) ufr[kk + 2][j] = uld1(s, kk + 2, j);
            } else if (nxt) {
#pragma unroll
                for (int j = 0; j < 3; ++j) ufr[kk - 2][j] = uld1(s + 1, kk - 2, j);
            }
            WSB();
        };
        // k-pair 0: next weights, next fragments; the halo of stage s + 2 -> raw[s % 2], the loads of stage s + 3
        uldn(0);
        fragread(set, 1);
        WSB();
        mf(0, 0);
        if (nxt) { rst1(set, 0, rreg[0]); rreg[0] = rld1(s3, 0); }
        WSB();
        mf(0, 1);
        if (nxt) { rst1(set, 1, rreg[1]); rreg[1] = rld1(s3, 1); }
        WSB();
#ifdef CTW_SPREAD                            // the patch reads and the transform a few instructions per MFMA slot
        float t[5][4];
        auto ddr = [&](int i) {
            if (nxt) {
#pragma unroll
                for (int j = 0; j < 4; ++j) dd[i][j] = smem[dbs[set ^ 1] + i * HALO + j];
            }
            WSB();
        };
        auto tcol = [&](int i) {
            if (nxt) {
                float* v = smem + vbs[set ^ 1] + (5 * i) * KS * VSTR;
                v[0 * KS * VSTR] = t[i][0] - t[i][1];
                v[1 * KS * VSTR] = t[i][1];
                v[2 * KS * VSTR] = t[i][2] - t[i][1];
                v[3 * KS * VSTR] = t[i][2];
                v[4 * KS * VSTR] = t[i][3] - t[i][2];
            }
            WSB();
        };
        mf(0, 2); ddr(0);
        mf(0, 3); ddr(1);
        mf(0, 4); ddr(2);
        mf(0, 5); ddr(3);
        mf(0, 6);
        mf(0, 7);
        if (nxt) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t[0][j] = dd[0][j] - dd[1][j];
                t[1][j] = dd[1][j];
                t[2][j] = dd[2][j] - dd[1][j];
                t[3][j] = dd[2][j];
                t[4][j] = dd[3][j] - dd[2][j];
            }
        }
        WSB();
        mf(0, 8);
        uldn(1);
        fragread(set, 2);
        WSB();
        mf(1, 0); tcol(0);
        mf(1, 1); tcol(1);
        mf(1, 2); tcol(2);
        mf(1, 3); tcol(3);
        mf(1, 4); tcol(4);
        mf(1, 5); mf(1, 6); mf(1, 7); mf(1, 8);
#else
        mf(0, 2);
        if (nxt) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) dd[i][j] = smem[dbs[set ^ 1] + i * HALO + j];
        }
        WSB();
        mf(0, 3); mf(0, 4); mf(0, 5); mf(0, 6); mf(0, 7); mf(0, 8);
        // k-pair 1: the transform of stage s + 1 (its 25 values into Vs[(s + 1) % 2])
        uldn(1);
        fragread(set, 2);
        WSB();
        mf(1, 0);
        if (nxt) transform_store(set ^ 1, dd);
        WSB();
        mf(1, 1); mf(1, 2); mf(1, 3); mf(1, 4); mf(1, 5); mf(1, 6); mf(1, 7); mf(1, 8);
#endif
        // k-pair 2
        uldn(2);
        fragread(set, 3);
        WSB();
#if defined(CTW_BAR_EARLY)                   // lab: the stage's barrier in front of k-pair 2 instead of behind it
        __syncthreads();
        WSB();
#endif
        mf(2, 0); mf(2, 1); mf(2, 2); mf(2, 3); mf(2, 4); mf(2, 5); mf(2, 6); mf(2, 7); mf(2, 8);
#if !defined(CTW_BAR_EARLY) && !defined(CTW_BAR_LATE)
        __syncthreads();
#endif
        // k-pair 3: the next stage's first fragments
        uldn(3);
#if defined(CTW_BAR_LATE)                    // ... or in the middle of k-pair 3 (the product's plan)
        mf(3, 0); mf(3, 1); mf(3, 2); mf(3, 3);
        __syncthreads();
        if (nxt) fragread(set ^ 1, 0);
        WSB();
        mf(3, 4); mf(3, 5); mf(3, 6); mf(3, 7); mf(3, 8);
#else
        if (nxt) fragread(set ^ 1, 0);
        WSB();
        mf(3, 0); mf(3, 1); mf(3, 2); mf(3, 3); mf(3, 4); mf(3, 5); mf(3, 6); mf(3, 7); mf(3, 8);
#endif
    };

    // prologue loads of a block: stages 0 and 1 (-> raw[0], raw[1]), stage 2's halo (kept in registers), the first two k-pairs' weights - requested
    // here for the workgroup's first block, for every later one from inside the previous block's epilogue
    floatx4 r0[2], r1[2];
    float bq;                                                // the block's bias, one value per lane: requested with the first loads, parked in LDS by the
                                                             // prologue (read straight from global memory in the epilogue it was an exposed L2 round trip)
    auto issue_loads = [&]() {
        bq = a.bias ? a.bias[n0 + (tid & 31)] : 0.f;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            r0[q] = rld1(0, q);
            r1[q] = rld1(1, q);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            ufr[0][j] = uld1(0, 0, j);
            ufr[1][j] = uld1(0, 1, j);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) rreg[q] = rld1(nst > 2 ? 2 : 1, q);
    };
    issue_loads();
#ifdef LWG_CTW_TS
    int lab_bi = 0;
#endif
    for (;;) {
    CTSB(2, 10);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        rst1(0, q, r0[q]);
        rst1(1, q, r1[q]);
    }
    if (tid < 32) smem[BIAS_OFF + tid] = bq;
    __syncthreads();
    CTSB(2, 11);
    {
        float dd[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) dd[i][j] = smem[dbs[0] + i * HALO + j];
        transform_store(0, dd);
    }
    __syncthreads();
    fragread(0, 0);
#pragma unroll
    for (int q = 0; q < 9; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    CTS(1);
    CTSB(1, 4);
    CTSB(2, 12);
    {
        int s = 0;
        for (; s + 2 < nst; s += 2) {
            iteration(s, IntT<0>(), IntT<1>());
            iteration(s + 1, IntT<1>(), IntT<1>());
        }
        iteration(s, IntT<0>(), IntT<1>());
        iteration(s + 1, IntT<1>(), IntT<0>());
    }
    CTS(2);
    CTSB(1, 5);
    const int eb = b, ex0 = x0, ey0 = y0, en0 = n0;          // this block's coordinates (the state moves on to the next block below)
    // epilogue: Y[a][b] = sum over xi in {a, a + 1}, nu in {b, b + 1} of M[xi][nu] is register-local (this lane holds patch pt * 32 + lane % 32 and,
    // per register group g, the four channels 8 g + 4 (lane / 32) ..); + bias, activation; then ONE exchange through LDS - the block's 32 x 32 output
    // pixels x 32 channels, rows of 36 floats - so that the global stores are whole lines: written straight from the accumulator layout a lane's
    // 16-byte pieces lie a pixel (YC floats) apart and the stores of a 256-channel layer took 55 k cycles per block (9 K stages' worth; r05_h)
    int tide = tid;                                          // (through an empty asm per block: the epilogue's address arithmetic must not be hoisted out
    asm volatile("" : "+v"(tide));                           //  of the block loop - it would sit in registers through the K loop)
    const int lanee = tide & 63;
    const int p = pt * 32 + (lanee & 31);
    const int ety = p >> 3, etx = p & 7;
    const int chl = 4 * (lanee >> 5);
    __syncthreads();                                         // every wave has read its last fragments: the loop's LDS is free
    CTSB(1, 6);
    lwg_act_dispatch(a.act, [&](auto ACTC) {                 // (the activation resolved once per block: lwg_common.h)
    constexpr int EA = decltype(ACTC)::value;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const floatx4 bv = *reinterpret_cast<const floatx4*>(smem + BIAS_OFF + 8 * g + chl);
#pragma unroll
        for (int ia = 0; ia < 2; ++ia)
#pragma unroll
            for (int ib = 0; ib < 2; ++ib) {
                floatx4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int r = 4 * g + k;
                    const float v = (acc[3 * ia + ib][r] + acc[3 * ia + ib + 1][r]) + (acc[3 * ia + 3 + ib][r] + acc[3 * ia + 4 + ib][r]);
                    o[k] = lwg_act_c<EA>(v + bv[k], a.act);
                }
                const int ly = 2 * (2 * ety + ia) + py, lx = 2 * (2 * etx + ib) + px;      // the pixel inside the block's 32 x 32 outputs
                *reinterpret_cast<floatx4*>(smem + (ly * 32 + ctw_slot(lx)) * OROW + 8 * g + chl) = o;
            }
    }
    });
    // the next block of this workgroup: its first loads go out here - the accumulators are dead - and land under the stores below (unconditional:
    // the last block re-requests its own first stages, nobody waits for them; see conv_winograd.hip)
    CTSB(1, 7);
    const int nblk = blk + (int)gridDim.x;
    const bool more = has_block(nblk);
    setup(more ? nblk : blk);
    issue_loads();
    CTSB(1, 8);
    __syncthreads();
    CTSB(1, 9);
    const int oy0 = 2 * ey0, ox0 = 2 * ex0;
    const size_t plane = (size_t)a.YH * a.YW;
    // The block's 32 x 32 x 32 outputs leave as BUFFER stores (round 6): image eb is one buffer, a thread's offset inside it is computed once, the
    // sixteen passes differ by one 32-bit add - no per-pass 64-bit address arithmetic, no per-pass bounds branch (profiles/r06_m_*: the store phase was
    // 4.2-5.4 k cycles of instruction issue per block).  Pixels right of the image: an out-of-range thread offset (the store is dropped); rows below it:
    // beyond the buffer's end in the NHWC layout (rows are its slowest dimension), an out-of-range scalar offset for the pass in the plane layout.
    typedef unsigned int ctw_u4 __attribute__((ext_vector_type(4)));
    if (a.ydt == LWG_DT_F32_Q4) {
        // channel-quad planes (B, YC/4, YH, YW, 4): 32 lanes = one output row of the block in one plane, 512 contiguous bytes
        const int lx = tide & 31, cq = (tide >> 5) & 7, lyh = tide >> 8;
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y + (size_t)eb * (size_t)(a.YC >> 2) * plane * 4, 0,
                                                                            (int)((unsigned)(a.YC >> 2) * (unsigned)plane * 16u), 0x00020000);
        const unsigned yv = ox0 + lx < a.YW ? (unsigned)(((((a.ycoff + en0) >> 2) + cq) * (int)plane + (oy0 + lyh) * a.YW + ox0 + lx) * 16) : WINO_OOB;
        const float* src = smem + (lyh * 32 + ctw_slot(lx)) * OROW + 4 * cq;
        const unsigned rowpair = (unsigned)a.YW * 32u;           // bytes between the rows of two passes (two rows of 16-byte pixels)
#pragma unroll
        for (int pass = 0; pass < 16; ++pass) {
            const ctw_u4 v = *reinterpret_cast<const ctw_u4*>(src + pass * 64 * OROW);
            // (the pass offset goes into the VECTOR offset, the scalar offset stays the constant 0: with a register in the scalar-offset field the compiler
            //  plans no wait state between a 16-byte store and a VALU write of its data registers - and the next pass's address add landed in the first
            //  data register right behind the store: intermittently corrupted first channels, found by tools/determinism_stress.py; r06_ar)
            __builtin_amdgcn_raw_buffer_store_b128(v, ry, (int)(oy0 + 2 * pass < a.YH ? yv + (unsigned)pass * rowpair : WINO_OOB), 0, CTW_NT_ST);
        }
    } else {
        // NHWC: 8 lanes = the block's 32 channels of one pixel, 128 contiguous bytes
        const int cq = tide & 7, lx = (tide >> 3) & 31, lyh = tide >> 8;
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y + (size_t)eb * plane * a.YC, 0, (int)((unsigned)plane * (unsigned)a.YC * 4u), 0x00020000);
        const unsigned yv = ox0 + lx < a.YW ? (unsigned)((((oy0 + lyh) * a.YW + ox0 + lx) * a.YC + a.ycoff + en0 + 4 * cq) * 4) : WINO_OOB;
        const float* src = smem + (lyh * 32 + ctw_slot(lx)) * OROW + 4 * cq;
        const unsigned rowpair = (unsigned)a.YW * (unsigned)a.YC * 8u;
#pragma unroll
        for (int pass = 0; pass < 16; ++pass) {
            const ctw_u4 v = *reinterpret_cast<const ctw_u4*>(src + pass * 64 * OROW);
            __builtin_amdgcn_raw_buffer_store_b128(v, ry, (int)(yv + (unsigned)pass * rowpair), 0, CTW_NT_ST);        // (vector offset: see above)
        }
    }
    CTSB(1, 13);
    if (!more) break;
    blk = nblk;
    __syncthreads();                                         // every thread has read its outputs from the exchange buffer: raw[0] / raw[1] may be written
    CTS_COUNT();
    }
    CTS(3);
}

// args: the parity-(0, 0) launch description of lwg_conv_transpose4_nhwc_f32 (ntaps = 4, stride = 1, omul = 2, OH = H, OW = W, YH = 2 H, YW = 2 W,
// LWG_EPI_NONE, one input) with Cin % 16 == 0, N % 32 == 0, ydt LWG_DT_F32 or LWG_DT_F32_Q4, EXCEPT args->w = the Winograd panel
// Upk[4][Cin/8][4][2][9 N] ([N][4] products 0-3, [N][4] products 4-7, [N] product 8 per (parity 2 py + px, stage s, k-pair kk, k-half kh)): product 3 xi + nu of column n =
// sgn * (G g G^T)[xi][nu] with g[r][q] = w[c][n][3 - py - 2 r][3 - px - 2 q] the parity's 2 x 2 sub-kernel of input channel c = 8 s + 2 kk + kh,
// G = [[1,0],[1,1],[0,1]] and sgn = (py == 1 && xi == 0 ? -1 : 1) * (px == 1 && nu == 0 ? -1 : 1).
extern "C" int lwg_conv_transpose4_winograd_f32(const LwgConvArgs* pa, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!pa) return (int)hipErrorInvalidValue;
    const LwgConvArgs& a = *pa;
    if (!a.x0 || !a.w || !a.y || a.M <= 0 || a.ntaps != 4 || a.stride != 1 || a.omul != 2 || a.C0 <= 0 || (a.C0 % (2 * KS)) != 0 || a.C1 != 0 ||
        a.N <= 0 || (a.N % NBT) != 0 || a.OH != a.H || a.OW != a.W || a.YH != 2 * a.H || a.YW != 2 * a.W || a.xdt != LWG_DT_F32 ||
        (a.ydt != LWG_DT_F32 && a.ydt != LWG_DT_F32_Q4) || a.M != a.B * a.H * a.W || a.epi != LWG_EPI_NONE || a.act == LWG_ACT_RELU_MASK ||
        a.ycoff < 0 || (a.ycoff % 4) != 0 || (a.YC % 4) != 0 || a.ycoff + a.N > a.YC)
        return (int)hipErrorInvalidValue;
    if ((unsigned long long)a.H * a.W * a.C0 * 4ull >= (unsigned long long)WINO_OOB || 144ull * a.C0 * a.N >= 0xffffffffull) return (int)hipErrorInvalidValue;
    // (an output image is one buffer of the store path: byte offsets + the sixteen row-pair offsets of a block stay below the out-of-range marker)
    if ((unsigned long long)a.YH * a.YW * a.YC * 4ull + 32ull * a.YW * a.YC * 4ull >= (unsigned long long)WINO_OOB) return (int)hipErrorInvalidValue;
    const size_t lds = (size_t)(BIAS_OFF + 32) * 4;
    static unsigned long long done = 0;
    if (hipError_t e = lwg_allow_dynamic_lds(reinterpret_cast<const void*>(lwg_convt_winograd_kernel), lds, done); e != hipSuccess) return (int)e;
    const int bx = (a.W + 2 * TPB - 1) / (2 * TPB), by = (a.H + 2 * TPB - 1) / (2 * TPB);
    const long total = (long)bx * by * a.B * (a.N / NBT);
    const int cus = lwg_device_cus();                        // persistent workgroups: one per CU (LDS) at most
    hipLaunchKernelGGL(lwg_convt_winograd_kernel, dim3((unsigned)(LWG_WINO_PERSIST && total > cus ? cus : total)), dim3(WG_THREADS), lds, stream, a);
    return (int)hipGetLastError();
}
