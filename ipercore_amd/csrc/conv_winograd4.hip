// Fused F(4x4, 3x3) Winograd form of the 3x3 / stride 1 / pad 1 fp32 convolution on v_mfma_f32_32x32x2_f32 (round 6; the sibling of conv_winograd.hip's
// F(2x2, 3x3) kernel for the layers whose K loop is long enough to pay for the larger transforms - DESIGN.md 3.12d; attlwb_spade_resunet.py:14-25,62-93,316-357).
//   y = act(bias + sum x * w [+ res]) computed as U = G w G^T (6 x 6 per channel pair; the fragment panel built by lwg_winograd4_panel_f32 below), V = B^T d B per
//   6 x 6 input patch d (patches overlap by two pixels), M_{xi,nu} = V_{xi,nu} U_{xi,nu} - thirty-six GEMMs over Cin -, Y = A^T M A (4 x 4 outputs per patch):
//   36 multiplies per 16 outputs = 2.25 per output instead of 9 (F(2x2, 3x3): 4).  Points 0, +-1, +-2, inf (Lavin & Gray):
//     B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//     G   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
//     A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
// Workgroup: 512 threads = 8 waves; block = 8 x 4 patches (32 x 16 output pixels) x 64 output channels.  Wave w = (q = w % 4, channel tile ct = w / 4) owns NINE
// products for the block's 32 patches x 32 channels (9 accumulator tiles of 32 x 32 = 144 VGPRs, rows = output channels - the transposed kernel's shape):
// the whole row xi = q of the transformed patch (nu = 0..5) and three products of row 4 + q / 2 (nu = 3 (q % 2) .. + 2).  The nu half of A^T M A is then
// register-local for rows 0..3 and half-local for rows 4, 5: 28 planes of (patch, channel) values cross LDS in the epilogue.
// A K stage is 8 input channels = four k-pairs; per k-pair a lane loads its nine weights as two 16-byte and one 4-byte buffer loads from the panel
// Upk[4][Cin/8][4][2][9 N] (every instruction reads contiguous memory; two k-pairs ahead, four register sets) and reads nine V fragments (4 bytes each) from LDS, each one right behind the MFMA that
// used its register for the previous k-pair.  The raw 34 x 18 x 8 halo goes global -> registers -> raw[s % 2] (channel-major planes, rows of 40 floats);
// the 256 (patch, channel) transforms of the next stage are shared by the 512 threads: waves 0-3 form rows 0..2 of B^T d B, waves 4-7 rows 3..5 (72 vector
// instructions per thread and stage), one barrier per stage in front of k-pair 3.
// Persistent workgroups in an XCD-aware block order (one column block per XCD for the whole launch; see the kernel), a 4-wave form for small launches.
// Rounding: relative L2 error against fp64 ~14-20x the direct fp32 convolution's (3e-6; tools/winograd_study.py --f43): fp32-grade, NOT the direct kernel's
// and not the F(2x2, 3x3) kernel's bits.  A frame's result does not depend on the batch it is launched in (per-image work in a fixed order).
#include <hip/hip_runtime.h>
#include "lwg_common.h"
#include "lwg_conv_args.h"

#define W4_THREADS 512
#define W4_PBX 8             // patches per block row: 32 output pixels
#define W4_PBY 4             // patch rows per block: 16 output pixels
#define W4_NB 64             // output channels per block
#define W4_KS 8              // input channels per stage
#define W4_HW 34             // halo pixels per row
#define W4_HH 18             // halo rows
#ifndef W4_RS
#define W4_RS 40             // floats per halo row in LDS (16-byte aligned rows).  40: a patch row (four halo rows) is 160 = 32 mod 64 floats, so the 16 lanes of a
                             // ds_read_b128 lane group - patches (pty, ptx 0-3 | 4-7) of four patch rows - read sixteen distinct bank quads in the transform
                             // (36: 144 = 16 mod 64, two-way conflicts on every patch-row read; PMC r06_ah / r06_ak)
#endif
#define W4_PLANE (18 * W4_RS + 4)   // floats per channel plane: 18 rows + 4 (4 PLANE = 16 mod 32: the two channel quads of a pixel fall into different banks)
                             // (lab: consecutive lanes = consecutive pixels of one quad + planes of 704 floats, i.e. ds_write2st64_b32 halo stores: 6-12 % SLOWER -
                             //  a lane pair no longer reads 32 contiguous bytes; profiles/r06_y3_*)
#define W4_RAW (W4_KS * W4_PLANE)
#define W4_VS (36 * W4_KS * 32)                  // [product 6 xi + nu][k][patch]
#define W4_NEL (W4_HW * W4_HH * 2)               // (pixel, channel quad) elements of a stage's halo
#define W4_NQ 3                                  // ... per thread (the third round: threads < W4_NEL - 1024 only)
#define W4_DUMP_OFF (2 * W4_RAW + 2 * W4_VS)     // where the threads without a halo element store their zeros (dead LDS)
#define W4_DUMP (W4_THREADS + 3 * W4_PLANE)
#define W4_LOOP (W4_DUMP_OFF + W4_DUMP)
#define W4_MSR 68                                // floats per (plane, patch) row of the epilogue's exchange buffer: 64 channels + 4
#define W4_NPL 28                                // planes: rows 0..3 x 4 output columns, rows 4 / 5: 2 halves x 3 partial sums
#define W4_MS (W4_NPL * 16 * W4_MSR)             // one pass = 16 patches
#define W4_BIAS_OFF (W4_LOOP > W4_MS ? W4_LOOP : W4_MS)          // the block's 64 bias values, behind both uses of the LDS
#define W4_OOB 0xC0000000u
#ifndef LWG_W4_XCD
#define LWG_W4_XCD 1         // XCD-aware block order (see the kernel): 0 = off (lab)
#endif
#ifndef LWG_W4_CHUNK
#define LWG_W4_CHUNK 1       // block order: 1 = chunks of gridDim.x tiles through every column block where the panel fits L2, 2 = always, 0 = never (lab)
#endif
#ifndef LWG_W4_RDMAP
#define LWG_W4_RDMAP 1       // epilogue reader lanes mapped to the ds_read_b128 lane groups (lab: 0 = lane order)
#endif
#ifndef LWG_W4_SMALL
#define LWG_W4_SMALL 1       // the 4-wave form for small launches (lab: 0 = off)
#endif
#define W4SB() __builtin_amdgcn_sched_barrier(0)

// lab instrumentation (compiled out of the product): tools/wino4lab.py --ts on a -DLWG_W4_TS variant library; wave 0 stamps into args->res (LWG_EPI_NONE)
#ifdef LWG_W4_TS
#define W4TS(i) do { if (tid == 0 && lab_bi == 1) reinterpret_cast<unsigned long long*>(const_cast<float*>(a.res))[(size_t)blockIdx.x * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#define W4TSA(i) do { if (tid == 0) reinterpret_cast<unsigned long long*>(const_cast<float*>(a.res))[(size_t)blockIdx.x * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define W4TS(i) do { } while (0)
#define W4TSA(i) do { } while (0)
#endif

template <int V> struct W4Int { static constexpr int value = V; };

__device__ __forceinline__ floatx4 w4_buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
#ifndef W4_NT
#define W4_NT 0              // cache policy of the activation traffic (halo loads, output stores): 0 = default; 2 = non-temporal: measured 20 % SLOWER (profiles/r06_w_*)
#endif
#ifndef W4_NT_LD
#define W4_NT_LD W4_NT       // ... of the halo loads alone
#endif
#ifndef W4_NT_RES
#define W4_NT_RES 0          // ... of the epilogue's residual / xn loads (read once per launch)
#endif
#ifndef W4_NT_ST
#define W4_NT_ST 2           // ... of the output stores alone: NON-TEMPORAL - the block's 128 KB of outputs are not read again by this launch and otherwise push
                             // panel and halo lines out of the XCD's L2: +0.4-0.6 % in the step (A/B/A/B and five policies, profiles/r06_ay_*; the halo LOADS
                             // non-temporal were the 20 % loss of r06_w)
#endif
__device__ __forceinline__ floatx4 w4_buf_load_nt(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, W4_NT_LD));
}

// the 1-D input transform B^T (.) of six values
__device__ __forceinline__ void w4_bt6(const float (&r)[6], float (&v)[6]) {
    const float a = __builtin_fmaf(-4.f, r[2], r[4]), b = __builtin_fmaf(-4.f, r[1], r[3]);
    const float c = r[4] - r[2], e = r[3] - r[1];
    v[0] = __builtin_fmaf(-5.f, r[2], __builtin_fmaf(4.f, r[0], r[4]));
    v[1] = a + b;
    v[2] = a - b;
    v[3] = __builtin_fmaf(2.f, e, c);
    v[4] = __builtin_fmaf(-2.f, e, c);
    v[5] = __builtin_fmaf(-5.f, r[3], __builtin_fmaf(4.f, r[1], r[5]));
}

// SM (small launches - a frame or two -, every epilogue): 256 threads = the four waves q = 0..3 of ONE 32-channel tile - block = the same 32 patches x
// 32 channels: twice the workgroups (one wave per SIMD: the 36 MFMAs of a wave and stage run unshared), every thread forms BOTH halves of its (patch, channel)
// transform and stages five halo pieces.  Every output element is accumulated over the stages and k-pairs in the same order and finished by the same
// expressions in both forms (the transform, the folds and the reader are explicit fma / add sequences): bitwise the same result - a frame does not depend on
// its batch (check_winograd4: frame n of a 40-frame launch = the frame alone, across the forms).
template <int EPI, bool TWO, bool SM = false>
__global__ __launch_bounds__(SM ? 256 : W4_THREADS, 1) void lwg_conv_winograd4_kernel(const LwgConvArgs a) {
    // (SM with the SPADE epilogue: the block's 32 accumulator rows are gamma | beta of the SAME 16 channels - columns n0 .. + 15 and n0 + 32 .. + 47 of the
    // stacked panel, n0 = 64 (block / 2) + 16 (block % 2) - so the modulation still finds both in one block)
    constexpr bool SMS = SM && EPI == LWG_EPI_SPADE;
    constexpr int NVP = SM && !SMS ? 2 : 1;                  // patches per reader thread and pass
    constexpr int NTH = SM ? 256 : W4_THREADS;               // threads per workgroup
    constexpr int NQ = SM ? 5 : W4_NQ;                       // halo pieces per thread and stage
    constexpr int NBV = SM ? 32 : W4_NB;                     // output channels per block
    constexpr int MSR = SM ? 36 : W4_MSR;                    // floats per (plane, patch) row of the exchange buffer
    constexpr int NHF = SM ? 2 : 1;                          // halves of the input transform per thread
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float* __restrict__ bias = a.bias;
    const int H = a.H, W = a.W, Cin = a.C0 + a.C1, N = a.N;
    float* __restrict__ y = a.y;
    float* const raw0 = smem;                                // [2][RAW], then [2][VS]
    float* const Ms = smem;                                  // the epilogue's exchange buffer (after the K loop)
    const int tid = threadIdx.x, lane = tid & 63;
    const int bx = (W + 4 * W4_PBX - 1) / (4 * W4_PBX), by = (H + 4 * W4_PBY - 1) / (4 * W4_PBY);
    // persistent workgroups (as conv_winograd.hip): min(blocks, CUs) workgroups walk the block ids blockIdx.x + k gridDim.x (id = column block * tiles + tile)
    const int tiles = bx * by * a.B;
    const int total = tiles * (N / NBV);
    int blk = blockIdx.x;
    const int nst = Cin / W4_KS;                             // even (host: Cin % 16 == 0)
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, (int)(144u * (unsigned)Cin * (unsigned)N), 0x00020000);
    const int q = wid & 3, ct = SM ? 0 : wid >> 2;           // this wave's product set and 32-channel tile
    const int half = SM ? 0 : wid >> 2;                      // ... and its half of the input transform (rows 3 half .. 3 half + 2 of B^T d B; SM: both)
    floatx16 acc[9];                                         // products 0..5: (xi = q, nu); 6..8: (xi = 4 + q / 2, nu = 3 (q % 2) + 0..2)
    int b, x0, y0, n0;
    __amdgpu_buffer_rsrc_t rx0, rx1;
    unsigned voff0[NQ], voff1[NQ];                     // this thread's halo elements (pixel, channel quad): byte offsets inside either input
    unsigned uvoff, uvoffc;                                  // this lane's column of the fragment panel: the 16-byte parts, the ninth product
    // XCD-aware block order (8-wave form, persistent grids of a multiple of 8 workgroups, N / 64 = 2, 4 or 8 column blocks): workgroup w runs on XCD w % 8
    // (round-robin dispatch) and keeps ONE column block, (w % 8) % ncb, for the whole launch - an XCD's 4 MB L2 holds that column block's panel only and
    // never turns it over -, while the ncb workgroups (w % 8) / ncb, w / 8 of adjacent XCDs walk the SAME tile sequence in step: a tile's halo is fetched
    // by ncb XCDs at about the same time - once from HBM, the rest out of the memory-side cache - instead of ncb times a whole pass over the batch apart
    const int ncb = N / NBV;
    // Measured inside the 300-frame step (profiles/r06_am_*): 2-7 % per launch for N >= 256 and for N = 128 with Cin >= 192; the N = 128 layers with
    // Cin <= 128 keep the chunked order below (1-4 % faster there)
    const bool xcd = LWG_W4_XCD && !SM && (gridDim.x & 7u) == 0 && (ncb == 4 || ncb == 8 || (ncb == 2 && (Cin >= 192 || LWG_W4_XCD == 2))) &&
                     (int)gridDim.x < total && tiles >= (int)gridDim.x / ncb;
    const int xg = (int)gridDim.x / ncb;                     // workgroups per column block = tiles per round
    const int xr = (int)(((blockIdx.x & 7u) / (unsigned)ncb) * (gridDim.x >> 3) + (blockIdx.x >> 3));      // this workgroup's place among them
    auto has_block = [&](int id) -> bool {                   // (id = blockIdx.x + k gridDim.x)
        return xcd ? (id / (int)gridDim.x) * xg + xr < tiles : id < total;
    };
    auto setup = [&](int id) {
        // block id -> (column block, tile).  Chunked order (layers whose WHOLE fragment panel stays in an XCD's 4 MB L2 - in the generator N = 128): the
        // grid's G persistent workgroups walk a chunk of G tiles through ALL column blocks before the next chunk (workgroup w: tile ch G + w in N / 64
        // consecutive blocks) - a tile's halo is re-read one round after its first read instead of a whole pass over the batch apart.  Measured inside the
        // 300-frame step (profiles/r06_ag_*): 2-3.4 % faster per launch for N = 128, 2-5 % SLOWER for N >= 256 (every round then pulls another column
        // block's panel through L2): those keep the column-block-major order
        int cb, t;
        if (xcd) {
            cb = (int)(blockIdx.x & 7u) & (ncb - 1);
            t = __builtin_amdgcn_readfirstlane((id / (int)gridDim.x) * xg + xr);
        } else if (LWG_W4_CHUNK == 2 || (LWG_W4_CHUNK == 1 && 144u * (unsigned)Cin * (unsigned)N <= (5u << 20))) {
            const int G = (int)gridDim.x, per = G * (N / NBV);
            const int ch = __builtin_amdgcn_readfirstlane(id / per);
            const int r = id - ch * per, base = ch * G;
            const int nt = tiles - base < G ? tiles - base : G;
            cb = __builtin_amdgcn_readfirstlane(r / nt);
            t = __builtin_amdgcn_readfirstlane(base + r - cb * nt);
        } else {
            cb = __builtin_amdgcn_readfirstlane(id / tiles);
            t = __builtin_amdgcn_readfirstlane(id - cb * tiles);
        }
        b = __builtin_amdgcn_readfirstlane(t / (bx * by));
        t -= b * bx * by;
        x0 = (t % bx) * 4 * W4_PBX;
        y0 = (t / bx) * 4 * W4_PBY;
        n0 = SMS ? (cb >> 1) * 64 + (cb & 1) * 16 : cb * NBV;
        rx0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x0 + (size_t)b * H * W * a.C0), 0, (int)((unsigned)(H * W) * (unsigned)a.C0 * 4u), 0x00020000);
        if constexpr (TWO) rx1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x1 + (size_t)b * H * W * a.C1), 0, (int)((unsigned)(H * W) * (unsigned)a.C1 * 4u), 0x00020000);
        int tids = tid;                                      // (through an empty asm: the halo geometry is recomputed per block - hoisted out of the block
        asm volatile("" : "+v"(tids));                       //  loop it would sit in registers through the K loop)
#pragma unroll
        for (int k = 0; k < NQ; ++k) {                       // padding pixels / threads without an element: an out-of-range offset (the hardware returns zeros)
            const int i = tids + NTH * k;
            const int pix = i >> 1, hq = i & 1, hy = pix / W4_HW, hx = pix - hy * W4_HW;
            const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
            const bool in = i < W4_NEL && gy >= 0 && gy < H && gx >= 0 && gx < W;
            voff0[k] = in ? (unsigned)((gy * W + gx) * a.C0 + 4 * hq) * 4u : W4_OOB;
            if constexpr (TWO) voff1[k] = in ? (unsigned)((gy * W + gx) * a.C1 + 4 * hq) * 4u : W4_OOB;
        }
        const int col = SMS ? n0 + (((lane & 31) >> 4) << 5) + (lane & 15) : n0 + ct * 32 + (lane & 31);      // this lane's accumulator row = panel column
        uvoff = (unsigned)(((lane >> 5) * 9 * N + 4 * col) * 4);
        uvoffc = (unsigned)(((lane >> 5) * 9 * N + 8 * N + col) * 4);
    };
    setup(blk);
    int wst[NQ];                                             // the halo elements' LDS slot
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
        const int i = tid + NTH * k;
        const int pix = i >> 1, hq = i & 1, hy = pix / W4_HW, hx = pix - hy * W4_HW;
        wst[k] = i < W4_NEL ? 4 * hq * W4_PLANE + hy * W4_RS + hx : W4_DUMP_OFF + tid;
    }
    floatx4 rreg[NQ];
    auto rld1 = [&](int st, int k) -> floatx4 {              // a stage's 8 channels lie in ONE input (C0 % 8 == 0)
#ifdef W4_LAB_HFIX            // lab: every halo load reads stage 0
        const int c = 0 * st;
#else
        const int c = st * W4_KS;
#endif
        if constexpr (!TWO) {
            return w4_buf_load_nt(rx0, voff0[k], (unsigned)c * 4u);
        } else {
            const bool first = c < a.C0;
            const __amdgpu_buffer_rsrc_t r = first ? rx0 : rx1;
            const unsigned v = first ? voff0[k] : voff1[k];
            return w4_buf_load_nt(r, v, (unsigned)(first ? c : c - a.C0) * 4u);
        }
    };
    auto rst1 = [&](int buf, int k, floatx4 v) {
        float* dst = raw0 + buf * W4_RAW + wst[k];
#pragma unroll
        for (int c = 0; c < 4; ++c) dst[c * W4_PLANE] = v[c];
    };
    // weights: lane = (k-half lane / 32, channel lane % 32); element (q, stage, k-pair, k-half, n) = twelve floats (nine products + padding)
    // weights: lane = (k-half lane / 32, channel lane % 32); element (q, stage, k-pair, k-half) = 9 N floats: [N][4] products 0-3, [N][4] products 4-7, [N] product 8 -
    // every load instruction reads contiguous memory (32 lanes x 16 bytes), nothing is padding
    floatx4 ufa[4], ufb[4];                                  // [register set = k-pair]: loaded TWO k-pairs ahead
    float ufc[4];
    const unsigned ukk = (unsigned)N * 72u;                  // bytes between two k-pairs: [2][9 N] floats
    const unsigned uq = (unsigned)q * (unsigned)nst * 4u * ukk;
    const unsigned ubo = (unsigned)N * 16u;                  // bytes from part A to part B
    auto uldpart = [&](int set, int st, int kk, int part) {  // part 0 | 1 | 2: products 0-3 | 4-7 | 8
#ifdef W4_LAB_UFIX
        const unsigned so = uq + 0u * (unsigned)(st + kk);
#else
        const unsigned so = uq + (unsigned)(st * 4 + kk) * ukk;
#endif
        if (part == 0) ufa[set] = w4_buf_load(ru, uvoff, so);
        else if (part == 1) ufb[set] = w4_buf_load(ru, uvoff, so + ubo);
        else ufc[set] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ru, (int)uvoffc, (int)so, 0));
    };
    auto uldset = [&](int set, int st, int kk) {
#ifdef W4_LAB_UFIX            // lab: every weight load reads k-pair 0 of stage 0 (cache-hot: the loads' issue / wait mechanics without the memory system behind them)
        const unsigned so = uq + 0u * (unsigned)(st + kk);
#else
        const unsigned so = uq + (unsigned)(st * 4 + kk) * ukk;
#endif
        ufa[set] = w4_buf_load(ru, uvoff, so);
        ufb[set] = w4_buf_load(ru, uvoff, so + ubo);
        ufc[set] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ru, (int)uvoffc, (int)so, 0));
    };
    // the input transform's thread: patch tid % 32, channel (tid / 32) % 8, rows 3 half .. 3 half + 2
    const int patch = tid & 31, tc = (tid >> 5) & 7;
    const int pty = patch >> 3, ptx = patch & 7;
    unsigned dbs = (unsigned)(tc * W4_PLANE + (4 * pty) * W4_RS + 4 * ptx) >> 2;      // its 6 x 6 input patch inside raw[0] (float index; 16-byte aligned)
    asm volatile("" : "+v"(dbs));
    dbs <<= 2;
    // per half hf of the transform (this thread's one - index 0 - or, SM, both): the rows of X start at row hf; its 18 transformed values inside Vs[0]:
    // row X (xi = 0 | 5) and rows T+, T- (xi = 1, 2 | 3, 4); alpha, beta (wave-uniform: scalar operands)
    unsigned dbx[NHF], vbx[NHF], vbt[NHF];
    float alpha[NHF], beta[NHF], nbeta[NHF];
#pragma unroll
    for (int hi = 0; hi < NHF; ++hi) {
        const int hf = SM ? hi : half;
        dbx[hi] = (unsigned)(tc * W4_PLANE + (4 * pty + hf) * W4_RS + 4 * ptx) >> 2;
        asm volatile("" : "+v"(dbx[hi]));
        dbx[hi] <<= 2;
        vbx[hi] = (unsigned)(2 * W4_RAW + (hf ? 30 : 0) * W4_KS * 32 + tc * 32 + patch);
        vbt[hi] = (unsigned)(2 * W4_RAW + (hf ? 18 : 6) * W4_KS * 32 + tc * 32 + patch);
        asm volatile("" : "+v"(vbx[hi]));
        asm volatile("" : "+v"(vbt[hi]));
        alpha[hi] = hf ? -1.f : -4.f;
        beta[hi] = hf ? 2.f : 1.f;
        nbeta[hi] = -beta[hi];
    }
    unsigned fbs[2];                                         // this lane's fragments: products 0..5 | 6..8, inside Vs[0]
    fbs[0] = (unsigned)(2 * W4_RAW + (6 * q) * W4_KS * 32 + lane);
    fbs[1] = (unsigned)(2 * W4_RAW + (6 * (4 + (q >> 1)) + 3 * (q & 1)) * W4_KS * 32 + lane);
    asm volatile("" : "+v"(fbs[0]));
    asm volatile("" : "+v"(fbs[1]));
    float fb[9];                                             // the current k-pair's nine fragments (each refilled right behind its MFMA)
    auto frag1 = [&](int buf, int kk, int j) -> float {
        return smem[fbs[j < 6 ? 0 : 1] + buf * W4_VS + ((j < 6 ? j : j - 6) * W4_KS + 2 * kk) * 32];
    };
    auto rd6 = [&](unsigned base, int buf, int r, float (&d)[6]) {     // row r (from the base's row) of this thread's patch in raw[buf]
        const floatx4 v4 = *reinterpret_cast<const floatx4*>(smem + base + buf * W4_RAW + r * W4_RS);
        const float2 v2 = *reinterpret_cast<const float2*>(smem + base + buf * W4_RAW + r * W4_RS + 4);
        d[0] = v4[0]; d[1] = v4[1]; d[2] = v4[2]; d[3] = v4[3]; d[4] = v2.x; d[5] = v2.y;
    };
    auto vst6 = [&](unsigned base, int buf, int i, const float (&v)[6]) {   // six values of a row of V -> Vs[buf], row i behind the base's
        float* dst = smem + base + buf * W4_VS + (6 * i) * W4_KS * 32;
#pragma unroll
        for (int nu = 0; nu < 6; ++nu) dst[nu * W4_KS * 32] = v[nu];
    };
    // column pass of the transform, the SAME instructions in both halves (no branches, one copy of the K loop):
    //   X  = 4 dx0 - 5 dx1 + dx2 over the rows (half, half + 2, half + 4)           = t0 | t5
    //   P  = d4 + alpha d2, Q = d3 + alpha d1 (alpha = -4 | -1); T+ = P + beta Q, T- = P - beta Q (beta = 1 | 2)   = t1, t2 | t3, t4
    auto transform_full = [&](int buf) {                     // prologue only
#pragma unroll
        for (int hi = 0; hi < NHF; ++hi) {
            float X[6], P[6], Q[6], v[6];
            {
                float dR[3][6];
                rd6(dbx[hi], buf, 0, dR[0]); rd6(dbx[hi], buf, 2, dR[1]); rd6(dbx[hi], buf, 4, dR[2]);
#pragma unroll
                for (int j = 0; j < 6; ++j) X[j] = __builtin_fmaf(-5.f, dR[1][j], __builtin_fmaf(4.f, dR[0][j], dR[2][j]));
            }
            {
                float dR[2][6];
                rd6(dbs, buf, 2, dR[0]); rd6(dbs, buf, 4, dR[1]);
#pragma unroll
                for (int j = 0; j < 6; ++j) P[j] = __builtin_fmaf(alpha[hi], dR[0][j], dR[1][j]);
                rd6(dbs, buf, 1, dR[0]); rd6(dbs, buf, 3, dR[1]);
#pragma unroll
                for (int j = 0; j < 6; ++j) Q[j] = __builtin_fmaf(alpha[hi], dR[0][j], dR[1][j]);
            }
            w4_bt6(X, v); vst6(vbx[hi], buf, 0, v);
            float T[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) T[j] = __builtin_fmaf(beta[hi], Q[j], P[j]);
            w4_bt6(T, v); vst6(vbt[hi], buf, 0, v);
#pragma unroll
            for (int j = 0; j < 6; ++j) T[j] = __builtin_fmaf(nbeta[hi], Q[j], P[j]);
            w4_bt6(T, v); vst6(vbt[hi], buf, 1, v);
        }
    };
    auto iteration = [&](int s, auto SET, auto NXT) {
        constexpr int set = decltype(SET)::value;            // s % 2
        constexpr bool nxt = decltype(NXT)::value != 0;      // the last stage has no next one to prepare (peeled: no branches in the loop)
        const int s3 = s + 3 < nst ? s + 3 : nst - 1;        // past the end: a harmless re-load of the last stage (its halo store lands in a dead buffer)
        float X[6], P[6], Q[6], dR[3][6];
        auto mf = [&](int kk, int j, bool refill) {          // product j of k-pair kk; then its register takes the fragment of the next k-pair
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(j < 4 ? ufa[kk][j & 3] : j < 8 ? ufb[kk][j & 3] : ufc[kk], fb[j], acc[j], 0, 0, 0);
#ifndef W4_KO_FRAG
            // (two fragments per LDS instruction: products j - 1 and j lie 1 KB apart - ds_read2st64_b32 -, refilled behind the second one's MFMA)
            if (refill && (j & 1)) {
                fb[j - 1] = kk < 3 ? frag1(set, kk + 1, j - 1) : frag1(set ^ 1, 0, j - 1);
                fb[j] = kk < 3 ? frag1(set, kk + 1, j) : frag1(set ^ 1, 0, j);
            } else if (refill && j == 8) {
                fb[8] = kk < 3 ? frag1(set, kk + 1, 8) : frag1(set ^ 1, 0, 8);
            }
#endif
            W4SB();
        };
        auto ul = [&](int kk, int part) {                    // ONE load of them
#ifndef W4_KO_ULD
            if (kk < 2) uldpart(kk + 2, s, kk + 2, part);
            else if (nxt) uldpart(kk - 2, s + 1, kk - 2, part);
#endif
            W4SB();
        };
        auto halo = [&](int k) {                             // the halo of stage s + 2 -> raw[s % 2], the loads of stage s + 3
#ifdef W4_KO_HALO
            return;
#endif
            if (nxt) { rst1(set, k, rreg[k]); rreg[k] = rld1(s3, k); }
            W4SB();
        };
#ifdef W4_KO_TR
        constexpr bool trn = false;
#else
        constexpr bool trn = nxt;
#endif
        // the next stage's transform in fourteen steps per half: 0-2 the rows of X, 3-4 X, 5-6 rows 2 / 4, 7 X's row of V, 8 P, 9-10 rows 1 / 3, 11 Q, 12-13 the rows T+ / T-
        auto tr = [&](int hi, int step) {
            if (trn) {
                if (step < 3) rd6(dbx[hi], set ^ 1, 2 * step, dR[step]);
                else if (step < 5) {
#pragma unroll
                    for (int j = 3 * (step - 3); j < 3 * (step - 3) + 3; ++j) X[j] = __builtin_fmaf(-5.f, dR[1][j], __builtin_fmaf(4.f, dR[0][j], dR[2][j]));
                } else if (step < 7) rd6(dbs, set ^ 1, 2 * (step - 5) + 2, dR[step - 5]);
                else if (step == 7) {
                    float v[6];
                    w4_bt6(X, v);
                    vst6(vbx[hi], set ^ 1, 0, v);
                } else if (step == 8) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) P[j] = __builtin_fmaf(alpha[hi], dR[0][j], dR[1][j]);
                } else if (step < 11) rd6(dbs, set ^ 1, 2 * (step - 9) + 1, dR[step - 9]);
                else if (step == 11) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) Q[j] = __builtin_fmaf(alpha[hi], dR[0][j], dR[1][j]);
                } else {
                    float T[6], v[6];
#pragma unroll
                    for (int j = 0; j < 6; ++j) T[j] = __builtin_fmaf(step == 12 ? beta[hi] : nbeta[hi], Q[j], P[j]);
                    w4_bt6(T, v);
                    vst6(vbt[hi], set ^ 1, step - 12, v);
                }
            }
            W4SB();
        };
        // Slot plan: the stage's fifteen loads (twelve weight parts two k-pairs ahead, three halo pieces a stage ahead) ONE per MFMA slot - requested in
        // bursts (six weight loads at the top of a k-pair, by eight waves at once) the memory pipeline's queue fills and the wave stalls in front of its
        // next MFMA (profiles/r06_z_*: with every load cache-hot the K loop still lost 20 % to its fillers; the transform and the fragment reads cost 1 % each)
        if constexpr (!SM) {
            // k-pair 0: the halo of stage s + 2 -> raw[s % 2], the loads of stage s + 3; the rows of X of the next stage's patch
            mf(0, 0, true); ul(0, 0);
            mf(0, 1, true); halo(0);
            mf(0, 2, true); ul(0, 1); tr(0, 0);
            mf(0, 3, true); halo(1); tr(0, 1);
            mf(0, 4, true); ul(0, 2); tr(0, 2);
            mf(0, 5, true); halo(2);
            mf(0, 6, true); tr(0, 3);
            mf(0, 7, true); tr(0, 4);
            mf(0, 8, true); tr(0, 5);
            // k-pair 1: the rest of the column pass, the row pass
            mf(1, 0, true); ul(1, 0); tr(0, 6);
            mf(1, 1, true); tr(0, 7);
            mf(1, 2, true); ul(1, 1); tr(0, 8); tr(0, 9);
            mf(1, 3, true); tr(0, 10);
            mf(1, 4, true); ul(1, 2);
            mf(1, 5, true); tr(0, 11);
            mf(1, 6, true); tr(0, 12);
            mf(1, 7, true);
            mf(1, 8, true); tr(0, 13);
            // k-pair 2
            mf(2, 0, true); ul(2, 0);
            mf(2, 1, true);
            mf(2, 2, true); ul(2, 1);
            mf(2, 3, true);
            mf(2, 4, true); ul(2, 2);
            mf(2, 5, true); mf(2, 6, true); mf(2, 7, true); mf(2, 8, true);
        } else {
            // the small form: five halo pieces and both halves of the transform per thread in front of the barrier
            mf(0, 0, true); ul(0, 0); halo(0);
            mf(0, 1, true); tr(0, 0);
            mf(0, 2, true); ul(0, 1); tr(0, 1);
            mf(0, 3, true); halo(1); tr(0, 2);
            mf(0, 4, true); ul(0, 2); tr(0, 3);
            mf(0, 5, true); halo(2); tr(0, 4);
            mf(0, 6, true); tr(0, 5);
            mf(0, 7, true); halo(3); tr(0, 6);
            mf(0, 8, true); tr(0, 7);
            mf(1, 0, true); ul(1, 0); tr(0, 8); tr(0, 9);
            mf(1, 1, true); halo(4); tr(0, 10);
            mf(1, 2, true); ul(1, 1); tr(0, 11);
            mf(1, 3, true); tr(0, 12);
            mf(1, 4, true); ul(1, 2); tr(0, 13);
            mf(1, 5, true); tr(1, 0);
            mf(1, 6, true); tr(1, 1);
            mf(1, 7, true); tr(1, 2);
            mf(1, 8, true); tr(1, 3);
            mf(2, 0, true); ul(2, 0); tr(1, 4);
            mf(2, 1, true); tr(1, 5);
            mf(2, 2, true); ul(2, 1); tr(1, 6);
            mf(2, 3, true); tr(1, 7);
            mf(2, 4, true); ul(2, 2); tr(1, 8); tr(1, 9);
            mf(2, 5, true); tr(1, 10);
            mf(2, 6, true); tr(1, 11);
            mf(2, 7, true); tr(1, 12);
            mf(2, 8, true); tr(1, 13);
        }
        // k-pair 3: behind the stage's barrier (the refills read the NEXT stage's fragments)
        __syncthreads();
        W4SB();
        mf(3, 0, nxt); ul(3, 0);
        mf(3, 1, nxt);
        mf(3, 2, nxt); ul(3, 1);
        mf(3, 3, nxt);
        mf(3, 4, nxt); ul(3, 2);
        mf(3, 5, nxt); mf(3, 6, nxt); mf(3, 7, nxt); mf(3, 8, nxt);
    };

    // prologue loads of a block: stages 0 and 1 (-> raw[0], raw[1]), stage 2's halo (kept in registers), the first two k-pairs' weights - requested here for
    // the workgroup's first block, for every later one from inside the previous block's epilogue
    floatx4 r0[NQ], r1[NQ];
    float bq;                                                // the block's bias, one value per lane of wave 0: requested with the first loads, parked in LDS by the prologue
    auto issue_loads = [&]() {
        bq = bias ? bias[SMS ? n0 + (((tid & 31) >> 4) << 5) + (tid & 15) : n0 + (tid & (NBV - 1))] : 0.f;
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            r0[k] = rld1(0, k);
            r1[k] = rld1(1, k);
        }
        uldset(0, 0, 0);
        uldset(1, 0, 1);
#pragma unroll
        for (int k = 0; k < NQ; ++k) rreg[k] = rld1(nst > 2 ? 2 : 1, k);
    };
    issue_loads();
#ifdef LWG_W4_TS
    int lab_bi = 0;
#endif
    W4TSA(11);
    for (;;) {
    W4TS(0);
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
        rst1(0, k, r0[k]);
        rst1(1, k, r1[k]);
    }
    if (tid < NBV) smem[W4_BIAS_OFF + tid] = bq;
    __syncthreads();
    transform_full(0);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 9; ++j) fb[j] = frag1(0, 0, j);
#pragma unroll
    for (int j = 0; j < 9; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // the bias for free: A^T m = (1, 1, 1, 1) for m = (0, 1, 0, 0, 0, 0), so a constant c added to product (xi, nu) = (1, 1) of every patch adds c to all
    // sixteen outputs - the accumulator of that product starts at the bias of its channel (rows = channels) instead of zero
    if (q == 1) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const floatx4 b4 = *reinterpret_cast<const floatx4*>(smem + W4_BIAS_OFF + ct * 32 + 8 * g + 4 * (lane >> 5));
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[1][4 * g + k] = b4[k];
        }
    }
    W4TS(1);
    {
        int s = 0;
        for (; s + 2 < nst; s += 2) {
            iteration(s, W4Int<0>(), W4Int<1>());
            iteration(s + 1, W4Int<1>(), W4Int<1>());
        }
        iteration(s, W4Int<0>(), W4Int<1>());
        iteration(s + 1, W4Int<1>(), W4Int<0>());
    }
    W4TS(2);
    const int eb = b, ex0 = x0, ey0 = y0, en0 = n0;          // this block's coordinates (the state moves on to the next block below)
    // epilogue.  (1) M A in registers: row q -> F[b] (b = 0..3, into acc[0..3]); the half row -> three partial sums (into acc[6..8]):
    //   nu 0..2: a0 = m0 + m1 + m2, a1 = m1 - m2, a2 = m1 + m2;  nu 3..5: b0 = m3 + m4, b1 = m3 - m4, b2 = m5
    //   (F[0] = a0 + b0, F[1] = a1 + 2 b1, F[2] = a2 + 4 b0, F[3] = a1 + 8 b1 + b2: finished by the reader)
    // (2) two passes of 16 patches: the 28 planes cross LDS ([plane][patch][64 channels + 4]: 16-byte stores and loads, conflict-free both ways); a reader
    // thread owns one output column b of one patch x the channel quads n4.. and 32 + n4..: A^T (.) over xi, bias, (residual | SPADE modulation), activation,
    // 16-byte NHWC stores.  LWG_EPI_SPADE: the block's 64 columns are gamma | beta of the SAME 32 channels (as conv_winograd.hip).
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float m0 = acc[0][r], m1 = acc[1][r], m2 = acc[2][r], m3 = acc[3][r], m4 = acc[4][r], m5 = acc[5][r];
        const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
        acc[0][r] = (m0 + s12) + s34;
        acc[1][r] = __builtin_fmaf(2.f, d34, d12);
        acc[2][r] = __builtin_fmaf(4.f, s34, s12);
        acc[3][r] = __builtin_fmaf(8.f, d34, d12) + m5;
    }
    if (q & 1) {                                             // (ONE wave-uniform branch around the whole loop)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float h0 = acc[6][r], h1 = acc[7][r];
            acc[6][r] = h0 + h1;
            acc[7][r] = h0 - h1;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float h0 = acc[6][r], h1 = acc[7][r], h2 = acc[8][r];
            acc[6][r] = (h0 + h1) + h2;
            acc[7][r] = h1 - h2;
            acc[8][r] = h1 + h2;
        }
    }
    W4TS(4);
    int tide = tid;                                          // (through an empty asm per block: the epilogue's address arithmetic must not be hoisted out
    asm volatile("" : "+v"(tide));                           //  of the block loop - it would sit in registers through the K loop)
    const int lanee = tide & 63;
    const int rb = wid & 3;                                  // reader: output column inside a patch (wave-uniform)
    // ... patch inside the pass and channel quad (and 32 + n4).  8-wave form: a ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27} and
    // {4-11, 16-19, 28-31} of either half wave (MI355X_MICROARCH: LDS), and a 16-byte slot of the exchange buffer lies in bank quad (patch + quad) % 16
    // (rows of 68 floats): a group reads the eight quads of patches a and a + 8 - sixteen distinct bank quads (eight lanes per patch in lane order
    // cost 2-3 LDS cycles per group; PMC: 24 % of the kernel's LDS cycles were bank conflicts, profiles/r06_ah_pmc_lds_f32_512.md).
    // SM: patches lanee >> 3 and 8 + lanee >> 3, the block's only 32 channels; SM + SPADE: gamma | beta rows n4 and 16 + n4
    const unsigned l5 = (unsigned)lanee & 31u, gsel = (0x0FF0F00Fu >> l5) & 1u;                    // in the first group of its half wave?
    const int gidx = __builtin_popcount((gsel ? 0x0FF0F00Fu : 0xF00F0FF0u) & ((1u << l5) - 1u));   // its place in the group: 0..15
    const int p16 = LWG_W4_RDMAP && !SM ? (wid >> 2) * 4 + 2 * (lanee >> 5) + (gsel ? 0 : 1) + 8 * (gidx >> 3) : (wid >> 2) * 8 + (lanee >> 3);
    const int n4 = SMS ? (lanee & 3) * 4 : LWG_W4_RDMAP && !SM ? (gidx & 7) * 4 : (lanee & 7) * 4;
    const int cbase = SMS ? (en0 >> 6) * 32 + ((en0 >> 4) & 1) * 16 : en0 >> 1;       // SPADE: the block's first output channel
    floatx4 mu, rs;
    if (EPI == LWG_EPI_SPADE) {
        mu = *reinterpret_cast<const floatx4*>(a.mean + (size_t)eb * a.YC + cbase + n4);
        rs = *reinterpret_cast<const floatx4*>(a.rstd + (size_t)eb * a.YC + cbase + n4);
    }
    // image eb of the output (and of res / xn: the output's layout) as ONE buffer: a reader thread's four pixels are 32-bit offsets inside it (out of
    // range: right of / below the image - the hardware drops the store and returns zeros for the load), the second channel group an immediate
    typedef unsigned int w4_u4 __attribute__((ext_vector_type(4)));
    const unsigned img_bytes = (unsigned)(H * W) * (unsigned)a.YC * 4u;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(y + (size_t)eb * H * W * a.YC, 0, (int)img_bytes, 0x00020000);
    const float* const esrc = EPI == LWG_EPI_SPADE ? a.xn : EPI == LWG_EPI_RESIDUAL ? a.res : a.y;
    const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(esrc) + (size_t)eb * H * W * a.YC, 0, (int)img_bytes, 0x00020000);
    const int chan = EPI == LWG_EPI_SPADE ? cbase + n4 : a.ycoff + en0 + n4;
    bool more = false;
    int nblk = blk;
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
        // (pass 0 needs no barrier in front of its writes: the last LDS reads of the K loop - the fragments of its last k-pair - were issued in front of
        // the last stage's barrier; pass 1 waits for the readers of pass 0)
        if (ph == 1) __syncthreads();
        if (((lanee >> 4) & 1) == ph) {
            float* dst = Ms + (lanee & 15) * MSR + ct * 32 + 4 * (lanee >> 5);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int bb = 0; bb < 4; ++bb)
                    *reinterpret_cast<floatx4*>(dst + (4 * q + bb) * 16 * MSR + 8 * g) = floatx4{acc[bb][4 * g], acc[bb][4 * g + 1], acc[bb][4 * g + 2], acc[bb][4 * g + 3]};
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    *reinterpret_cast<floatx4*>(dst + (16 + 3 * q + i) * 16 * MSR + 8 * g) = floatx4{acc[6 + i][4 * g], acc[6 + i][4 * g + 1], acc[6 + i][4 * g + 2], acc[6 + i][4 * g + 3]};
            }
        }
        // this pass's pixels: patch p, output column rb, rows 0..3 (SM: two patches, one channel group)
        unsigned vo[NVP][4];
#pragma unroll
        for (int hp = 0; hp < NVP; ++hp) {
            const int p = ph * 16 + (SMS ? lanee >> 2 : SM ? 8 * hp + (lanee >> 3) : p16);
            const int ox = ex0 + 4 * (p & 7) + rb, oyb = ey0 + 4 * (p >> 3);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                vo[hp][i] = ox < W && oyb + i < H ? (unsigned)(((oyb + i) * W + ox) * a.YC + chan) * 4u : W4_OOB;
        }
        floatx4 ext[2][4];                                   // residual (both channel groups | SM: both patches) | xn (group 0)
        if (EPI != LWG_EPI_NONE) {
#pragma unroll
            for (int h = 0; h < (EPI == LWG_EPI_SPADE ? 1 : 2); ++h)
#pragma unroll
                for (int i = 0; i < 4; ++i) ext[h][i] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(re, (int)(vo[NVP == 2 ? h : 0][i] + (SM ? 0u : 128u * h)), 0, W4_NT_RES));
        }
        if (ph == 1) {
            // the next block of this workgroup: its first loads go out here - the accumulators are dead - and land under the second pass's output
            // (unconditional: the last block re-requests its own first stages, nobody waits for them; see conv_winograd.hip)
            W4TS(8);
            nblk = blk + (int)gridDim.x;
            more = has_block(nblk);
            setup(more ? nblk : blk);
            W4TS(9);
            issue_loads();
            W4TS(10);
        }
        __syncthreads();
        if (ph == 0) W4TS(5); else W4TS(7);
        floatx4 gam[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float* base = Ms + (SMS ? lanee >> 2 : SM ? 8 * h + (lanee >> 3) : p16) * MSR + (SMS ? h * 16 : SM ? 0 : h * 32) + n4;
            floatx4 F[6];
#pragma unroll
            for (int xi = 0; xi < 4; ++xi) F[xi] = *reinterpret_cast<const floatx4*>(base + (4 * xi + rb) * 16 * MSR);
            const int ia = rb == 0 ? 0 : rb == 2 ? 2 : 1, ib = (rb & 1) ? 4 : 3;
            const float cb = (float)(1 << rb);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const float* hb = base + (16 + 6 * r) * 16 * MSR;
                const floatx4 pa = *reinterpret_cast<const floatx4*>(hb + ia * 16 * MSR);
                const floatx4 pb = *reinterpret_cast<const floatx4*>(hb + ib * 16 * MSR);
                floatx4 f;
#pragma unroll
                for (int c = 0; c < 4; ++c) f[c] = __builtin_fmaf(cb, pb[c], pa[c]);
                if (rb == 3) f += *reinterpret_cast<const floatx4*>(hb + 5 * 16 * MSR);
                F[4 + r] = f;
            }
            floatx4 o[4];                                    // the four rows of this thread's output column (bias included: see the accumulators' start)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float s12 = F[1][c] + F[2][c], d12 = F[1][c] - F[2][c], s34 = F[3][c] + F[4][c], d34 = F[3][c] - F[4][c];
                o[0][c] = (F[0][c] + s12) + s34;
                o[1][c] = __builtin_fmaf(2.f, d34, d12);
                o[2][c] = __builtin_fmaf(4.f, s34, s12);
                o[3][c] = __builtin_fmaf(8.f, d34, d12) + F[5][c];
            }
            if (EPI == LWG_EPI_SPADE) {
                if (h == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) gam[i] = o[i];   // gamma
                    continue;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int c = 0; c < 4; ++c) o[i][c] = (ext[0][i][c] - mu[c]) * rs[c] * (1.f + gam[i][c]) + o[i][c];
            } else if (EPI == LWG_EPI_RESIDUAL) {
                if (a.act == LWG_ACT_RELU_MASK) {            // data gradient behind a ReLU: res = the forward input, the mask source
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int c = 0; c < 4; ++c) o[i][c] = ext[h][i][c] > 0.f ? o[i][c] : 0.f;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] += ext[h][i];
                }
            }
            // (the activation resolved once per sixteen values - lwg_common.h; around the whole output pass the four copies of the pass cost registers:
            // 40-90 spilled, their reloads between the stores each waiting for every store before them)
            lwg_act_dispatch(a.act, [&](auto ACTC) {
                constexpr int EA = decltype(ACTC)::value;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int c = 0; c < 4; ++c) o[i][c] = lwg_act_c<EA>(o[i][c], a.act);
            });
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(w4_u4, o[i]), ry, (int)(vo[NVP == 2 ? h : 0][i] + (EPI == LWG_EPI_SPADE || SM ? 0u : 128u * h)), 0, W4_NT_ST);
        }
        if (ph == 0) W4TS(6);
    }
    W4TS(3);
#ifdef LWG_W4_TS
    ++lab_bi;
#endif
    if (!more) break;
    blk = nblk;
    __syncthreads();                                         // every reader is done with the exchange buffer: raw[0] / raw[1] (the same LDS) may be written
    }
    W4TSA(12);
#ifdef LWG_W4_TS
    if (tid == 0) reinterpret_cast<unsigned long long*>(const_cast<float*>(a.res))[(size_t)blockIdx.x * 16 + 13] = (unsigned long long)lab_bi;
#endif
}


// The fragment panel from the fp32 GEMM panel of the same convolution (lwg_conv2d_nhwc_f32's w: [9 Cin / 4][N][4], k = ((c / 32) 9 + tap) 32 + c % 32):
// U = G w G^T (6 x 6) per (input channel, output column) in fp64, rounded once, written as Upk[4][Cin/8][4][2][9 N]: block (q, s, kk, kh) of input channel
// c = 8 s + 2 kk + kh holds product j of column n at [n][j] (j = 0..3), 4 N + [n][j - 4] (j = 4..7), 8 N + [n] (j = 8); j < 6: U[q][j]; j = 6..8: U[4 + q / 2][3 (q % 2) + j - 6].  tap9[3 r + s] = the tap index of kernel
// position (dy, dx) = (r - 1, s - 1) in the GEMM panel.  One thread per (c, n).
struct LwgWino4Taps { int t[9]; };

__global__ __launch_bounds__(256) void lwg_winograd4_panel_kernel(const float* __restrict__ wp, float* __restrict__ U, int Cin, int N, LwgWino4Taps taps) {
    const int n = blockIdx.x * 64 + (threadIdx.x & 63), c = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (n >= N || c >= Cin) return;
    double g[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int k = ((c >> 5) * 9 + taps.t[3 * r + s]) * 32 + (c & 31);
            g[r][s] = (double)wp[((size_t)(k >> 2) * N + n) * 4 + (k & 3)];
        }
    const double G[6][3] = {{1.0 / 4, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6}, {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    double t[6][3];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int s = 0; s < 3; ++s) t[i][s] = G[i][0] * g[0][s] + G[i][1] * g[1][s] + G[i][2] * g[2][s];
    const int s8 = c >> 3, kk = (c & 7) >> 1, kh = c & 1;
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
        float* dst = U + ((((size_t)qq * (Cin >> 3) + s8) * 4 + kk) * 2 + kh) * 9 * (size_t)N;
        float o[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int xi = j < 6 ? qq : 4 + (qq >> 1), nu = j < 6 ? j : 3 * (qq & 1) + (j - 6);
            o[j] = (float)(t[xi][0] * G[nu][0] + t[xi][1] * G[nu][1] + t[xi][2] * G[nu][2]);
        }
        *reinterpret_cast<floatx4*>(dst + 4 * n) = floatx4{o[0], o[1], o[2], o[3]};
        *reinterpret_cast<floatx4*>(dst + 4 * (size_t)N + 4 * n) = floatx4{o[4], o[5], o[6], o[7]};
        dst[8 * (size_t)N + n] = o[8];
    }
}

extern "C" int lwg_winograd4_panel_f32(const float* wpanel, float* upk, int Cin, int N, const int* tap9, lwg_stream_t stream_) {
    if (!wpanel || !upk || !tap9 || Cin <= 0 || (Cin % 32) != 0 || N <= 0) return (int)hipErrorInvalidValue;
    LwgWino4Taps taps;
    for (int i = 0; i < 9; ++i) {
        if (tap9[i] < 0 || tap9[i] > 8) return (int)hipErrorInvalidValue;
        taps.t[i] = tap9[i];
    }
    hipLaunchKernelGGL(lwg_winograd4_panel_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)((Cin + 3) / 4)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream_), wpanel, upk, Cin, N, taps);
    return (int)hipGetLastError();
}

// args: lwg_conv2d_winograd_f32's launch description (3 x 3 / stride 1 / pad 1, one or two inputs with C0 % 8 == 0, C1 % 8 == 0, (C0 + C1) % 16 == 0, N % 64 == 0,
// YC % 4 == 0; LWG_EPI_NONE, LWG_EPI_RESIDUAL or LWG_EPI_SPADE; any activation of lwg_act) EXCEPT args->w = the F(4x4, 3x3) fragment panel of
// lwg_winograd4_panel_f32, 144 Cin N bytes.
static bool lwg_wino4_contract(const LwgConvArgs& a) {
    if (!a.x0 || !a.w || !a.y || a.M <= 0 || a.ntaps != 9 || a.stride != 1 || a.omul != 1 || a.C0 <= 0 || (a.C0 % W4_KS) != 0 || a.C1 < 0 ||
        (a.C1 % W4_KS) != 0 || ((a.C0 + a.C1) % (2 * W4_KS)) != 0 || (a.C1 > 0 && !a.x1) || a.N <= 0 || (a.N % W4_NB) != 0 || a.OH != a.H || a.OW != a.W ||
        a.YH != a.H || a.YW != a.W || a.xdt != LWG_DT_F32 || a.ydt != LWG_DT_F32 || a.M != a.B * a.H * a.W || a.ycoff < 0 || (a.ycoff % 4) != 0 ||
        (a.YC % 4) != 0 || (a.act == LWG_ACT_RELU_MASK && a.epi != LWG_EPI_RESIDUAL))
        return false;
    if (a.epi == LWG_EPI_SPADE) {
        if (!a.xn || !a.mean || !a.rstd || !a.bias || a.YC * 2 != a.N || a.ycoff != 0) return false;
    } else {
        if (a.ycoff + a.N > a.YC) return false;
        if (a.epi != LWG_EPI_NONE && (a.epi != LWG_EPI_RESIDUAL || !a.res)) return false;
    }
    const unsigned long long cmax = (unsigned long long)(a.C0 > a.C1 ? a.C0 : a.C1);
    if ((unsigned long long)a.H * a.W * cmax * 4ull >= (unsigned long long)W4_OOB || 144ull * (a.C0 + a.C1) * a.N >= 0xffffffffull) return false;
    if ((unsigned long long)a.H * a.W * a.YC * 4ull + 256ull >= (unsigned long long)W4_OOB) return false;      // (an output image is one buffer of the store path)
    return true;
}

extern "C" int lwg_conv2d_winograd4_f32(const LwgConvArgs* pa, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!pa || !lwg_wino4_contract(*pa)) return (int)hipErrorInvalidValue;
    const LwgConvArgs& a = *pa;
    const size_t lds = (size_t)(W4_BIAS_OFF + 64) * 4;
    const int bx = (a.W + 4 * W4_PBX - 1) / (4 * W4_PBX), by = (a.H + 4 * W4_PBY - 1) / (4 * W4_PBY);
    const int cus = lwg_device_cus();
    long total = (long)bx * by * a.B * (a.N / W4_NB);
    // small launches (the 8-wave blocks would leave half the chip or more without a workgroup): the 4-wave form, 32 channels per block (SPADE: gamma | beta
    // of 16 channels) - bitwise the same result (see the kernel), so the choice may depend on the batch
    const bool sm = LWG_W4_SMALL && 2 * total <= cus;
    if (sm) total *= 2;
    const dim3 grid((unsigned)(LWG_WINO_PERSIST && total > cus ? cus : total));
    static unsigned long long done[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const bool two = a.C1 > 0;
#define LWG_W4_GO2(E, T, S, SLOT)                                                                                                       \
    {                                                                                                                                   \
        if (hipError_t e = lwg_allow_dynamic_lds(reinterpret_cast<const void*>(lwg_conv_winograd4_kernel<E, T, S>), lds, done[SLOT]); e != hipSuccess) \
            return (int)e;                                                                                                              \
        hipLaunchKernelGGL((lwg_conv_winograd4_kernel<E, T, S>), grid, dim3(S ? 256 : W4_THREADS), lds, stream, a);                     \
    }
#define LWG_W4_GO(E, SLOT)                                                                                                              \
    {                                                                                                                                   \
        if (two) LWG_W4_GO2(E, true, false, SLOT + 3) else LWG_W4_GO2(E, false, false, SLOT)                                            \
    }
#define LWG_W4_GOS(E, SLOT)                                                                                                             \
    {                                                                                                                                   \
        if (two) LWG_W4_GO2(E, true, true, SLOT + 2) else LWG_W4_GO2(E, false, true, SLOT)                                              \
    }
    if (sm) {
        if (a.epi == LWG_EPI_SPADE) {
            if (two) LWG_W4_GO2(LWG_EPI_SPADE, true, true, 11) else LWG_W4_GO2(LWG_EPI_SPADE, false, true, 10)
        } else if (a.epi == LWG_EPI_RESIDUAL) LWG_W4_GOS(LWG_EPI_RESIDUAL, 7)
        else LWG_W4_GOS(LWG_EPI_NONE, 6)
    } else if (a.epi == LWG_EPI_SPADE) LWG_W4_GO(LWG_EPI_SPADE, 2)
    else if (a.epi == LWG_EPI_RESIDUAL) LWG_W4_GO(LWG_EPI_RESIDUAL, 1)
    else LWG_W4_GO(LWG_EPI_NONE, 0)
#undef LWG_W4_GOS
#undef LWG_W4_GO
#undef LWG_W4_GO2
    return (int)hipGetLastError();
}
