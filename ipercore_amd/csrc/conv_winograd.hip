// Fused F(2x2, 3x3) Winograd form of the 3x3 / stride 1 / pad 1 fp32 convolution on v_mfma_f32_32x32x2_f32 (ops.conv_precision("winograd");
// round 4, re-designed in round 5; first developed as tools/probes/winograd_f23.hip, numerics in tools/winograd_study.py).
//   y = act(bias + sum x * w [+ res]) computed as U = G w G^T (the fragment panel Upk[16][Cin/8][2][N][4], built by the panel kernels below), V = B^T d B per 4x4 input patch d
//   (patches overlap by two pixels), M_{xi,nu} = V_{xi,nu} U_{xi,nu} - sixteen GEMMs over Cin -, Y = A^T M A: 16 multiplies per 2x2 outputs instead of 36.
// Workgroup: 512 threads = 8 waves; block = 8 x 8 patches (16 x 16 output pixels) x 64 output channels; wave w owns the FOUR products of one row of
// the transformed patch, (xi, nu = 0..3) with xi = w % 4, for all 64 patches x the 32 channels of tile w / 4 (4 x 2 accumulator tiles of 32 x 32 =
// 128 VGPRs) - so the nu half of the output transform A^T M A is register-local and only HALF of the products' volume crosses waves in the epilogue.
// A K stage is 8 input channels:
//   * U fragments: a 32x32x2 fp32 MFMA operand is ONE register, so a lane loads its four k-pairs of a (product, channel) as one 16-byte buffer
//     load from the panel, one stage ahead, into the other of two register sets - the weights never touch LDS;
//   * the raw 18 x 18 x 8 halo patch goes global -> registers (three stages ahead) -> raw[s % 2] (channel-major planes, two stages ahead); padding
//     pixels and threads without a halo element carry an out-of-range buffer offset (the hardware returns zeros: no branches, no exec masks);
//   * the next stage's input transform (one (patch, channel) 4x4 -> V[16] per thread: raw[(s + 1) % 2] -> Vs[(s + 1) % 2]) runs beside the MFMAs.
// Schedule (round 5, measured with tools/winoshapes.py --ts / --ts2 on knock-out and calibration builds, profiles/r05_b_*): an iteration is 32 slots
// of ONE MFMA (m = kk * 8 + nu * 2 + tb) with the stage's other work behind the first dozen of them; the barrier sits after slot 23 and the next
// stage's fragments for k-pairs 0..2 are read right behind it into the registers slots 0-23 have finished with (k-pair 3 at the next top).  What
// the filler instructions cost in matrix-pipe time (the fp32 MFMA runs at the vector rate and shares the SIMD's issue with the vector ALU): a
// vector ALU instruction 2 cycles + 6 per slot that has any, an LDS instruction 1-2.5, a buffer load nothing to issue - so every LDS address is a
// base register + immediate, stage offsets are scalar, the transform's 32 adds sit in two slots, and the loads go first (only their latency matters).
// Epilogue: every wave folds its four products over nu in registers (M A: two values per (patch, channel)), the four rows xi of a (patch, channel)
// live in four different waves -> ONE exchange through LDS ([xi][column][patch][n], rows padded to 68 floats: 16-byte stores and loads, conflict-free
// both ways); a thread then owns one patch x two channel quads (n4.., 32 + n4..): A^T (.) over xi + bias (+ residual, or the SPADE modulation of a
// gamma | beta launch) + activation, 16-byte NHWC stores.  A second input (skip concatenation) is read stage by stage: a stage's 8 channels lie in one of the two tensors.
// Training step (trainers.TrainOpts.conv_precision = "winograd"): the same kernel runs the data gradients (dY with the flipped, transposed kernel; the
// LWG_ACTIVATION_RELU_MASK epilogue = the producing layer's ReLU backward); the fragment panels of weights that change every step are built on the
// device from the GEMM panels (lwg_winograd_panel_f32 / lwg_winograd_panels_f32 below); one-sample launches that would leave half the chip idle run
// their K loop in slices (template flag SPLIT, lwg_conv2d_winograd_f32_ws: slabs + the direct engine's finishing kernel).
// Rounding: relative L2 error against fp64 1.6x that of the direct fp32 convolution over the generator's layers (tools/winograd_study.py); NOT bitwise
// the direct kernel's result - which is why it is a precision mode of its own.
#include <hip/hip_runtime.h>
#include "lwg_common.h"
#include "lwg_conv_args.h"

#define WG_THREADS 512
#define TPB 8            // patches per block edge: 8 x 8 patches = 16 x 16 output pixels
#define NPATCH 64
#define NB 64            // output channels per block
#define KS 8             // input channels per stage
#define HALO 18
#define PLANE (HALO * HALO)
#define RAW_FLOATS (KS * PLANE)              // [c][py][px]
#define VSTR 64
#define VS_FLOATS (16 * KS * VSTR)           // [xinu][k][patch]
#define DUMP_OFF (2 * RAW_FLOATS + 2 * VS_FLOATS)                 // where the threads without a halo element store their zeros (dead LDS)
#define DUMP_FLOATS (WG_THREADS + 3 * PLANE + RAW_FLOATS)
#define LOOP_FLOATS (DUMP_OFF + DUMP_FLOATS)
#define MS_FLOATS (8 * NPATCH * (NB + 4))    // the epilogue's exchange buffer [xi (4)][output column (2)][patch (64)][n (NBV) + 4]: rows of NBV + 4
                                             // floats - 16-byte accesses of eight consecutive patches then cover all banks once (68 j, 36 j = 4 j mod 32)
#define WINO_OOB 0xC0000000u                 // >= any image's byte size (host: H * W * C * 4 < 3 GiB): the buffer load returns 0
#define WSB() __builtin_amdgcn_sched_barrier(0)

// lab instrumentation (compiled out of the product): tools/winoshapes.py --ts / --ts2 on -DLWG_WINO_TS / -DLWG_WINO_TS2 variant libraries
// (tools/labvariant.sh); the stamps go to args->res (LWG_EPI_NONE launches only)
#ifdef LWG_WINO_TS           // wave 0 of every workgroup: entry, K-loop entry, every second stage, K-loop exit, end (64 stamps per workgroup)
#define WTS(i) do { if (tid == 0) reinterpret_cast<unsigned long long*>(const_cast<float*>(a.res))[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 64 + (i)] = __builtin_readcyclecounter(); } while (0)
#define WTSB(k, i) do { if (lab_bi == (k)) WTS(i); } while (0)     // the workgroup's k-th block only (persistent form: block 1 = steady state)
#define WTS_COUNT() ++lab_bi
#else
#define WTS(i) do { } while (0)
#define WTSB(k, i) do { } while (0)
#define WTS_COUNT() do { } while (0)
#endif
#ifdef LWG_WINO_TS2          // every wave: eight points of iterations 8 and 9
#define WTS2(k) do { if (s == 8 || s == 9) ts2[(s - 8) * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define WTS2(k) do { } while (0)
#endif

template <int V> struct IntC { static constexpr int value = V; };

__device__ __forceinline__ floatx4 wino_buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}

// NBV = output channels per workgroup.  64: the form described above.  32 (small launches - a frame or two -, plain / residual epilogues): the
// same 64 patches x 32 channels, wave w = (xi = w % 4, patch tile w / 4) with ONE accumulator tile per product: twice the workgroups of half the
// matrix work each, so that a launch that would leave most of the chip idle fills it.  Every output element is accumulated over the stages and
// k-pairs in the same order and finished by the same expressions in both forms: bitwise the same result (a frame does not depend on its batch).
// SPLIT (training launches that leave half the chip or more idle; NBV = 32, plain epilogue): blockIdx.z owns the stages [z a.cshift, (z + 1) a.cshift)
// of the K loop (a.cshift - unused by convolutions with Cin >= 32 - carries the stages per slice here) and leaves raw sums in its dense (M, N) slab
// a.y + z M N (the host passes YC = N, ycoff = 0, no bias, no activation); lwg_splitk_finish_kernel adds the slabs in slice order and finishes.
// TWO (round 6): the launch has a second input (skip concatenation).  One-input launches - four fifths of the engine's time - carry no per-load choice
// of the source tensor at all: as a uniform branch pair around every halo load it cost 1.9 % of the K loop and 700 cycles of every block's set-up
// (profiles/r06_k_*); the two-input form selects the descriptor / offset (scalar selects + one v_cndmask per load) instead of branching.
template <int EPI, int NBV, bool SPLIT = false, bool TWO = false>
__global__ __launch_bounds__(WG_THREADS, 1) void lwg_conv_winograd_kernel(const LwgConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NT = NBV / 32;                             // accumulator (patch) tiles per product and wave
    constexpr int MSR = NBV + 4;                             // floats per (plane, patch) row of the epilogue's exchange buffer
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float* __restrict__ bias = a.bias;
    const int H = a.H, W = a.W, Cin = a.C0 + a.C1, N = a.N;
    float* __restrict__ y = SPLIT ? a.y + (size_t)blockIdx.z * (size_t)a.M * (size_t)N : a.y;
    float* const raw0 = smem;                                // [2][RAW]
    float* const Vs0 = smem + 2 * RAW_FLOATS;                // [2][VS]
    float* Ms = smem;                                        // the epilogue's exchange buffer (after the K loop)
    const int tid = threadIdx.x, lane = tid & 63;
    const int bx = (W + 2 * TPB - 1) / (2 * TPB), by = (H + 2 * TPB - 1) / (2 * TPB);
    // PERSISTENT WORKGROUPS (round 6): the grid is min(blocks, CUs) workgroups (the chip holds one per CU: LDS) and a workgroup walks the block ids
    // blockIdx.x, blockIdx.x + gridDim.x, ... (id = column block * tiles + tile: the order the 2-D grid of round 5 was dispatched in).  What that
    // buys: the first halo stages and weight fragments of block k + 1 are REQUESTED before block k's epilogue exchange and land under it (the
    // prologue's exposed global round trip, ~1/2 of its 6.3 k cycles, is gone), and the per-workgroup dispatch / descriptor set-up is paid once.
    // A block's arithmetic does not depend on which workgroup runs it: results are bitwise those of the one-block-per-workgroup form.
    const int tiles = bx * by * a.B;
    const int total = tiles * (N / NBV);
    int blk = blockIdx.x;
    const int nst_all = Cin / KS;                            // even (host: Cin % 16 == 0)
    const int sbeg = SPLIT ? (int)blockIdx.z * a.cshift : 0; // this workgroup's first stage and its number of stages (even as well)
    const int nst = SPLIT ? (a.cshift < nst_all - sbeg ? a.cshift : nst_all - sbeg) : nst_all;
    WTS(0);
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, (int)(64u * (unsigned)Cin * (unsigned)N), 0x00020000);
    // ---- per-block state (workgroup-uniform): image b, tile corner (x0, y0), first output column n0; this image of each input as a buffer
    int b, x0, y0, n0;
    __amdgpu_buffer_rsrc_t rx0, rx1;
    unsigned voff0[2], voff1[2];                             // this thread's two halo elements (pixel, channel quad): byte offsets inside either input
    unsigned uvoff;                                          // this lane's column of the fragment panel
    const int nbw_ = (NBV / 32) == 2 ? wid >> 2 : 0;
    auto setup = [&](int id) {
        const int cb = __builtin_amdgcn_readfirstlane(id / tiles);
        int t = __builtin_amdgcn_readfirstlane(id - cb * tiles);
        b = __builtin_amdgcn_readfirstlane(t / (bx * by));
        t -= b * bx * by;
        x0 = (t % bx) * 2 * TPB;
        y0 = (t / bx) * 2 * TPB;
        n0 = cb * NBV;
        rx0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x0 + (size_t)b * H * W * a.C0), 0, (int)((unsigned)(H * W) * (unsigned)a.C0 * 4u), 0x00020000);
        if constexpr (TWO) rx1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x1 + (size_t)b * H * W * a.C1), 0, (int)((unsigned)(H * W) * (unsigned)a.C1 * 4u), 0x00020000);
#pragma unroll
        for (int q = 0; q < 2; ++q) {                        // padding pixels / threads without a halo element: an out-of-range offset (zeros)
            const int i = tid + WG_THREADS * q;
            const int pix = i >> 1, half = i & 1, py = pix / HALO, px = pix - py * HALO;
            const int gy = y0 - 1 + py, gx = x0 - 1 + px;
            const bool in = i < PLANE * 2 && gy >= 0 && gy < H && gx >= 0 && gx < W;
            voff0[q] = in ? (unsigned)((gy * W + gx) * a.C0 + 4 * half) * 4u : WINO_OOB;
            if constexpr (TWO) voff1[q] = in ? (unsigned)((gy * W + gx) * a.C1 + 4 * half) * 4u : WINO_OOB;
        }
        uvoff = (unsigned)(((lane >> 5) * N + n0 + nbw_ * 32 + (lane & 31)) * 16);
    };
    setup(blk);
    const int xi = wid & 3;                                   // this wave's row of the transformed patch
    const int nbw = NT == 2 ? wid >> 2 : 0;                   // ... its 32-channel tile (NBV = 64)
    const int ptw = NT == 2 ? 0 : wid >> 2;                   // ... or its 32-patch tile (NBV = 32)
    floatx16 acc[4][NT];                                     // [product nu][patch tile] (cleared in the prologue, behind the first loads)

    int wst[2];                                              // ... and their LDS slot (threads without a halo element store their zeros into dead LDS)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = tid + WG_THREADS * q;
        const int pix = i >> 1, half = i & 1;
        wst[q] = i < PLANE * 2 ? 4 * half * PLANE + pix : DUMP_OFF + tid;
    }
    floatx4 rreg[2];
    auto rld1 = [&](int st, int q) -> floatx4 {              // a stage's 8 channels lie in ONE input (C0 % 8 == 0)
        const int c = (st + sbeg) * KS;
        if constexpr (!TWO) {
            return wino_buf_load(rx0, voff0[q], (unsigned)c * 4u);
        } else {
            const bool first = c < a.C0;
            const __amdgpu_buffer_rsrc_t r = first ? rx0 : rx1;
            const unsigned v = first ? voff0[q] : voff1[q];
            return wino_buf_load(r, v, (unsigned)(first ? c : c - a.C0) * 4u);
        }
    };
    auto rst1 = [&](int buf, int q, floatx4 v) {
        float* dst = raw0 + buf * RAW_FLOATS + wst[q];
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[k * PLANE] = v[k];
    };
    floatx4 ufr[2][4];                                       // [register set][product nu]: the four k-pairs of this lane's row (channel)
    const unsigned ustage = (unsigned)N * 32u;               // bytes between two stages of a product: [2][N][4] floats
    unsigned usoff[4];
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) usoff[nu] = ((unsigned)(4 * xi + nu) * (unsigned)nst_all + (unsigned)sbeg) * ustage;
    auto uld1 = [&](int st, int nu) -> floatx4 { return wino_buf_load(ru, uvoff, usoff[nu] + (unsigned)st * ustage); };
    const int patch = tid & 63, tc = tid >> 6;
    const int pty = patch >> 3, ptx = patch & 7;
    const float* const dbase = raw0 + tc * PLANE + (2 * pty) * HALO + 2 * ptx;      // this thread's 4 x 4 input patch inside raw[0]
    // Vs rows hold patch p at position (p % 32) * 2 + p / 32: a lane's fragments of the two patch tiles are ONE 8-byte read
    float* const vbase = Vs0 + tc * VSTR + (patch & 31) * 2 + (patch >> 5);        // its 16 transformed values inside Vs[0]
    // base registers with immediate offsets behind them (a vector add between two MFMAs costs matrix-pipe time: the fp32 MFMA shares the SIMD's
    // issue with the vector ALU - measured 2 cycles per instruction + 6 per slot that has any, profiles/r05_b_winograd_calibration.txt);
    // the empty asm keeps the compiler from re-deriving one base from another inside the loop
    // (float indices into smem, not pointers: the LDS address space has to survive the asm)
    unsigned dbs[2];
    unsigned fbs[2][4];                                      // [buffer][product nu]: this lane's fragments
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        dbs[u] = (unsigned)(u * RAW_FLOATS + tc * PLANE + (2 * pty) * HALO + 2 * ptx) >> 1;     // (even: the asm hides the value, the shift
        asm volatile("" : "+v"(dbs[u]));                                                          //  keeps the 8-byte alignment visible)
        dbs[u] <<= 1;
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
            fbs[u][nu] = (unsigned)(2 * RAW_FLOATS + u * VS_FLOATS + ((4 * xi + nu) * KS + (lane >> 5)) * VSTR + (lane & 31) * 2) >> 1;
            asm volatile("" : "+v"(fbs[u][nu]));
            fbs[u][nu] <<= 1;
            if (NT == 1) fbs[u][nu] += ptw;                  // one patch tile: the 4-byte half of the pair
        }
    }
    auto transform = [&](int buf) {                          // prologue only
        const float* d = dbase + buf * RAW_FLOATS;
        float* v = vbase + buf * VS_FLOATS;
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
            v[(i * 4 + 0) * KS * VSTR] = t[i][0] - t[i][2];
            v[(i * 4 + 1) * KS * VSTR] = t[i][1] + t[i][2];
            v[(i * 4 + 2) * KS * VSTR] = t[i][2] - t[i][1];
            v[(i * 4 + 3) * KS * VSTR] = t[i][1] - t[i][3];
        }
    };
#ifdef LWG_WINO_TS2
    unsigned long long ts2[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) ts2[i] = 0;
#endif
    typedef float fragx __attribute__((ext_vector_type(NT)));
    fragx fb[4][4];                                          // [k-pair][product nu]: patch tiles 0 | 1 (NBV = 64) or this wave's one
    auto fragread = [&](int set, int kk) {
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) fb[kk][nu] = *reinterpret_cast<const fragx*>(smem + fbs[set][nu] + 2 * kk * VSTR);
    };
    auto iteration = [&](int s, auto SET, auto NXT) {
        constexpr int set = decltype(SET)::value;
        constexpr bool nxt = decltype(NXT)::value != 0;      // the last stage has no next one to prepare (peeled: no branches in the loop)
        const int s3 = s + 3 < nst ? s + 3 : nst - 1;        // past the end: a harmless re-load of the last stage (its halo store lands in a dead buffer)
        float dd[4][4], t[4];
        auto mf = [&](int m) {                               // slot m of 16 NT: k-pair m / (4 NT), product (m / NT) % 4, patch tile m % NT
            const int kk = m / (4 * NT), nu = (m / NT) & 3, tb = m % NT;
            acc[nu][tb] = __builtin_amdgcn_mfma_f32_32x32x2f32(ufr[set][nu][kk], fb[kk][nu][tb], acc[nu][tb], 0, 0, 0);
            WSB();
        };
        auto rst = [&](int q) {                              // stage s + 2's halo (loaded during iteration s - 1) -> raw[s % 2]
            if (nxt) rst1(set, q, rreg[q]);
            WSB();
        };
        auto rld = [&](int q) {
            if (nxt) rreg[q] = rld1(s3, q);
            WSB();
        };
        auto uld = [&](int nu) {
            if (nxt) ufr[set ^ 1][nu] = uld1(s + 1, nu);
            WSB();
        };
        auto ddr = [&](int i) {
            if (nxt) {
#pragma unroll
                for (int j = 0; j < 4; ++j) dd[i][j] = smem[dbs[set ^ 1] + i * HALO + j];
            }
            WSB();
        };
        auto t2a = [&](int i) {                              // row i of B^T d: over the four columns
            if (nxt) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    t[j] = i == 0 ? dd[0][j] - dd[2][j] : i == 1 ? dd[1][j] + dd[2][j] : i == 2 ? dd[2][j] - dd[1][j] : dd[1][j] - dd[3][j];
            }
            WSB();
        };
        auto t2b = [&](int i) {                              // ... times B, V stores
            if (nxt) {
                float* v = vbase + (set ^ 1) * VS_FLOATS + (i * 4) * KS * VSTR;
                v[0 * KS * VSTR] = t[0] - t[2];
                v[1 * KS * VSTR] = t[1] + t[2];
                v[2 * KS * VSTR] = t[2] - t[1];
                v[3 * KS * VSTR] = t[1] - t[3];
            }
            WSB();
        };
        WTS2(0);
        fragread(set, 3);                                    // k-pair 3 of THIS stage (its registers were busy until the previous slot 31)
        WSB();
        // slot plan: loads first (their issue is free, their latency is what has to be covered), then the halo store, the transform reads, and the
        // transform's 32 vector adds in TWO slots (a slot that has any vector instruction costs ~6 cycles of matrix pipe on top of 2 per instruction)
        if (NT == 2) {
            mf(0); uld(0);
            mf(1); uld(1);
            mf(2); uld(2);
            mf(3); uld(3);
            WTS2(1);
            mf(4); rst(0); rld(0);
            mf(5); rst(1); rld(1);
            mf(6); ddr(0); ddr(1); ddr(2); ddr(3);
            mf(7);
            WTS2(2);
            mf(8);
            mf(9); t2a(0); t2b(0); t2a(1); t2b(1);
            mf(10);
            mf(11); t2a(2); t2b(2); t2a(3); t2b(3);
            mf(12); mf(13); mf(14); mf(15);
            WTS2(3);
            mf(16); mf(17); mf(18); mf(19); mf(20);
            mf(21); mf(22); mf(23);
        } else {
            mf(0); uld(0); uld(1);
            mf(1); uld(2); uld(3);
            mf(2); rst(0); rld(0);
            mf(3); rst(1); rld(1);
            mf(4); ddr(0); ddr(1); ddr(2); ddr(3);
            mf(5);
            mf(6); t2a(0); t2b(0); t2a(1); t2b(1);
            mf(7);
            mf(8); t2a(2); t2b(2); t2a(3); t2b(3);
            mf(9); mf(10); mf(11);
        }
        WTS2(4);
        __syncthreads();
        WTS2(5);
        if (nxt) {
            fragread(set ^ 1, 0);
            fragread(set ^ 1, 1);
        }
        WSB();
        mf(12 * NT);
        if (nxt) fragread(set ^ 1, 2);
        WSB();
        if (NT == 2) {
            mf(25); mf(26); mf(27); mf(28);
            WTS2(6);
            mf(29); mf(30); mf(31);
        } else {
            mf(13); mf(14); mf(15);
        }
        WTS2(7);
    };

    // prologue loads of a block: stages 0 and 1 (-> raw[0], raw[1]), stage 2's halo (kept in registers), U(0) in set 0 - requested here for the
    // workgroup's first block, and for every later one from inside the previous block's epilogue
    floatx4 r0[2], r1[2];
    auto issue_loads = [&]() {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            r0[q] = rld1(0, q);
            r1[q] = rld1(1, q);
        }
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) ufr[0][nu] = uld1(0, nu);
#pragma unroll
        for (int q = 0; q < 2; ++q) rreg[q] = rld1(nst > 2 ? 2 : 1, q);
    };
    issue_loads();
#ifdef LWG_WINO_TS
    int lab_bi = 0;
#endif
    for (;;) {
    WTSB(2, 50);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        rst1(0, q, r0[q]);
        rst1(1, q, r1[q]);
    }
    __syncthreads();
    WTSB(2, 51);
    transform(0);
    __syncthreads();
    WTSB(2, 52);
    fragread(0, 0);
    fragread(0, 1);
    fragread(0, 2);
#pragma unroll
    for (int nu = 0; nu < 4; ++nu)                           // (cleared here, not at the top: the first stages' halo registers are free by now)
#pragma unroll
        for (int tb = 0; tb < NT; ++tb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nu][tb][r] = 0.f;
    WTS(1);
    WTSB(1, 44);
    WTSB(2, 53);
    {
        int s = 0;
        for (; s + 2 < nst; s += 2) {
            iteration(s, IntC<0>(), IntC<1>());
            iteration(s + 1, IntC<1>(), IntC<1>());
            WTS(2 + (s >> 1));
        }
        iteration(s, IntC<0>(), IntC<1>());
        iteration(s + 1, IntC<1>(), IntC<0>());
    }
    WTS(40);
    WTSB(1, 45);
#ifdef LWG_WINO_TS2
    if (lane == 0)
        for (int i = 0; i < 16; ++i)
            reinterpret_cast<unsigned long long*>(const_cast<float*>(a.res))[(((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + wid) * 16 + i] = ts2[i];
#endif
    const int eb = b, ex0 = x0, ey0 = y0, en0 = n0;          // this block's coordinates (the state moves on to the next block below)
    // epilogue: M A in registers (two values per product row, patch, channel), ONE exchange through LDS [xi][column][patch][n], then a thread owns
    // one patch x the channel quads n4.. and 32 + n4..: 16 conflict-free 16-byte LDS reads, A^T (.) over xi, bias, (residual | SPADE modulation),
    // activation, 16-byte NHWC stores (the residual / xn values are fetched before the exchange).  LWG_EPI_SPADE: the block's 64 columns are
    // gamma | beta of the SAME 32 channels (the host interleaves the stacked panel in blocks of 32, as for lwg_conv_igemm_kernel): quad n4.. is
    // gamma, quad 32 + n4.. beta of channels (n0 / 2) + n4..: y = (xn - mean) rstd (1 + gamma) + beta.
    // reader threads: channel quad n4 = 4 (lane % 8), patch = (lane / 8) * 8 + wave: the sixteen lanes one ds_read_b128 serves together hold
    // quads and patches whose 16-byte rows fall into different banks (68 p + n4 over p, p + 8, p + 16, p + 24)
    // (thread ids taken through an empty asm per block: the epilogue's address arithmetic must not be hoisted out of the block loop - it would sit in
    // registers through the K loop, which has none to spare)
    int tide = tid;
    asm volatile("" : "+v"(tide));
    const int lanee = tide & 63;
    const int n4 = (tide & 7) * 4, ep = (lanee & 56) + wid;
    const int ety = ep >> 3, etx = ep & 7;
    constexpr int NH = NT;                                   // channel quads per reader thread: n4.. and (NBV = 64) 32 + n4..
    floatx4 ext[2][2][2];                                    // [channel group h][row][column]: residual (LWG_EPI_RESIDUAL) / xn (LWG_EPI_SPADE: h = 0 only)
    floatx4 bv[2], mu, rs;
#pragma unroll
    for (int h = 0; h < NH; ++h) bv[h] = bias ? *reinterpret_cast<const floatx4*>(bias + en0 + h * 32 + n4) : floatx4{0.f, 0.f, 0.f, 0.f};
    if (EPI == LWG_EPI_SPADE) {
        mu = *reinterpret_cast<const floatx4*>(a.mean + (size_t)eb * a.YC + (en0 >> 1) + n4);
        rs = *reinterpret_cast<const floatx4*>(a.rstd + (size_t)eb * a.YC + (en0 >> 1) + n4);
    }
    if (EPI != LWG_EPI_NONE) {
#pragma unroll
        for (int h = 0; h < (EPI == LWG_EPI_SPADE ? 1 : NH); ++h) {
            const int ch = EPI == LWG_EPI_SPADE ? (en0 >> 1) + n4 : a.ycoff + en0 + h * 32 + n4;
            const float* src = EPI == LWG_EPI_SPADE ? a.xn : a.res;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int px = 0; px < 2; ++px) {
                    const int oy = ey0 + 2 * ety + i, ox = ex0 + 2 * etx + px;
                    ext[h][i][px] = floatx4{0.f, 0.f, 0.f, 0.f};
                    if (oy < H && ox < W) ext[h][i][px] = *reinterpret_cast<const floatx4*>(src + (((size_t)eb * H + oy) * W + ox) * a.YC + ch);
                }
        }
    }
#pragma unroll
    for (int tb = 0; tb < NT; ++tb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {                        // D layout: registers 4 g .. 4 g + 3 are four consecutive rows (channels): one 16-byte store
            floatx4 c0, c1;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = 4 * g + k;
                c0[k] = acc[0][tb][r] + acc[1][tb][r] + acc[2][tb][r];
                c1[k] = acc[1][tb][r] - acc[2][tb][r] - acc[3][tb][r];
            }
            float* dst = Ms + ((xi * 2) * NPATCH + (NT == 2 ? tb : ptw) * 32 + (lanee & 31)) * MSR + nbw * 32 + 8 * g + 4 * (lanee >> 5);
            *reinterpret_cast<floatx4*>(dst) = c0;
            *reinterpret_cast<floatx4*>(dst + NPATCH * MSR) = c1;
        }
    // the next block of this workgroup: its first loads go out HERE - the accumulators are dead (folded into the exchange buffer), the residual / xn
    // fetches are ahead of them in the queue - and land while the products cross LDS and the outputs are stored
    // (UNCONDITIONAL for the last block too - it re-requests its own first stages, ten loads per thread nobody waits for: under `if (more)` the
    // halo registers would be conditionally defined, i.e. merged with their previous values, i.e. live through the whole K loop - 24 registers the
    // 64-channel form does not have)
    WTSB(1, 54);
    const int nblk = blk + (int)gridDim.x;
    const bool more = !SPLIT && nblk < total;
    if constexpr (!SPLIT) {
        setup(more ? nblk : blk);
        issue_loads();
    }
    WTSB(1, 46);
    __syncthreads();
    WTSB(1, 47);
    // (the activation resolved once per block: lwg_act_dispatch, lwg_common.h)
    lwg_act_dispatch(a.act, [&](auto ACTC) {
    constexpr int EA = decltype(ACTC)::value;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        floatx4 o[2][2], sx[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int px = 0; px < 2; ++px) sx[i][px] = *reinterpret_cast<const floatx4*>(Ms + ((i * 2 + px) * NPATCH + ep) * MSR + h * 32 + n4);
#pragma unroll
        for (int px = 0; px < 2; ++px) {
            o[0][px] = sx[0][px] + sx[1][px] + sx[2][px];
            o[1][px] = sx[1][px] - sx[2][px] - sx[3][px];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int px = 0; px < 2; ++px) {
                const int oy = ey0 + 2 * ety + i, ox = ex0 + 2 * etx + px;
                floatx4 v = o[i][px] + bv[h];
                if (EPI == LWG_EPI_SPADE) {
                    if (h == 0) {
                        ext[1][i][px] = v;                   // gamma
                    } else if (oy < H && ox < W) {
                        floatx4 r;
#pragma unroll
                        for (int c = 0; c < 4; ++c) r[c] = lwg_act_c<EA>((ext[0][i][px][c] - mu[c]) * rs[c] * (1.f + ext[1][i][px][c]) + v[c], a.act);
                        *reinterpret_cast<floatx4*>(y + (((size_t)eb * H + oy) * W + ox) * a.YC + (en0 >> 1) + n4) = r;
                    }
                } else if (oy < H && ox < W) {
                    floatx4 r;
                    if (EPI == LWG_EPI_RESIDUAL && lwg_act_is_mask<EA>(a.act)) {       // data gradient behind a ReLU: res = the forward input, the mask source
#pragma unroll
                        for (int c = 0; c < 4; ++c) r[c] = ext[h][i][px][c] > 0.f ? v[c] : 0.f;
                    } else {
                        if (EPI == LWG_EPI_RESIDUAL) v += ext[h][i][px];
#pragma unroll
                        for (int c = 0; c < 4; ++c) r[c] = lwg_act_c<EA>(v[c], a.act);
                    }
                    *reinterpret_cast<floatx4*>(y + (((size_t)eb * H + oy) * W + ox) * a.YC + a.ycoff + en0 + h * 32 + n4) = r;
                }
            }
    }
    });
    WTS(41);
    WTSB(1, 48);
    if (!more) break;
    blk = nblk;
    __syncthreads();
    WTSB(1, 49);
    WTS_COUNT();                                         // every reader is done with the exchange buffer: raw[0] / raw[1] (the same LDS) may be written
    }
    WTS(42);
}


// The fragment panel from the fp32 GEMM panel of the same convolution (lwg_conv2d_nhwc_f32's w: [9 Cin / 4][N][4], k = ((c / 32) 9 + tap) 32 + c % 32):
// U = G w G^T per (input channel, output column) in fp64, rounded once, written as Upk[16][Cin/8][2][N][4] - one thread per (c, n).  tap9[3 r + s] =
// the tap index of kernel position (dy, dx) = (r - 1, s - 1) in that panel.  Inference builds it once per weight version; the personalization step
// once per convolution call (the weights change every step).
struct LwgWinoTaps { int t[9]; };

// one thread = one output column n x the FOUR input channels 8 s + kh + {0, 2, 4, 6} that share a 16-byte element of the panel: sixteen coalesced
// 16-byte stores per thread (a thread per channel wrote 4 bytes of each: 2.2 TB/s, r05_g)
__device__ __forceinline__ void lwg_winograd_panel_quad(const float* __restrict__ wp, float* __restrict__ U, int Cin, int N, const int* t9, int cq, int n) {
    const int s8 = cq >> 1, kh = cq & 1;
    floatx4 u4[16];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int c = 8 * s8 + 2 * kk + kh;
        double g[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int k = ((c >> 5) * 9 + t9[3 * r + q]) * 32 + (c & 31);
                g[r][q] = (double)wp[((size_t)(k >> 2) * N + n) * 4 + (k & 3)];
            }
        double t[4][3];                                      // G g: rows (g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            t[0][q] = g[0][q];
            t[1][q] = 0.5 * (g[0][q] + g[1][q] + g[2][q]);
            t[2][q] = 0.5 * (g[0][q] - g[1][q] + g[2][q]);
            t[3][q] = g[2][q];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double u[4] = {t[i][0], 0.5 * (t[i][0] + t[i][1] + t[i][2]), 0.5 * (t[i][0] - t[i][1] + t[i][2]), t[i][2]};
#pragma unroll
            for (int j = 0; j < 4; ++j) u4[i * 4 + j][kk] = (float)u[j];
        }
    }
#pragma unroll
    for (int p = 0; p < 16; ++p) *reinterpret_cast<floatx4*>(U + ((((size_t)p * (Cin >> 3) + s8) * 2 + kh) * N + n) * 4) = u4[p];
}

__global__ __launch_bounds__(256) void lwg_winograd_panel_kernel(const float* __restrict__ wp, float* __restrict__ U, int Cin, int N, LwgWinoTaps taps) {
    const int n = blockIdx.x * 64 + (threadIdx.x & 63), cq = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (n < N && cq < (Cin >> 2)) lwg_winograd_panel_quad(wp, U, Cin, N, taps.t, cq, n);
}

// every registered panel of a training step in one launch (the weights change every step): workgroup b serves descriptor d = the last one with
// first_block <= b (binary search), as lwg_pack_panels_f32 does for the GEMM panels
__global__ __launch_bounds__(256) void lwg_winograd_panels_kernel(const LwgWinoDesc* __restrict__ descs, int ndesc) {
    int lo = 0, hi = ndesc - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const LwgWinoDesc* d = descs + lo;
    const int N = d->N, Cin = d->Cin, nbx = (N + 63) >> 6;
    const int lb = (int)blockIdx.x - d->first_block;
    const int n = (lb % nbx) * 64 + (threadIdx.x & 63), cq = (lb / nbx) * 4 + (threadIdx.x >> 6);
    if (n < N && cq < (Cin >> 2)) lwg_winograd_panel_quad(d->wpanel, d->upk, Cin, N, d->tap9, cq, n);
}

extern "C" int lwg_winograd_panel_f32(const float* wpanel, float* upk, int Cin, int N, const int* tap9, lwg_stream_t stream_) {
    if (!wpanel || !upk || !tap9 || Cin <= 0 || (Cin % 32) != 0 || N <= 0) return (int)hipErrorInvalidValue;
    LwgWinoTaps taps;
    for (int i = 0; i < 9; ++i) {
        if (tap9[i] < 0 || tap9[i] > 8) return (int)hipErrorInvalidValue;
        taps.t[i] = tap9[i];
    }
    hipLaunchKernelGGL(lwg_winograd_panel_kernel, dim3((unsigned)((N + 63) / 64), (unsigned)((Cin + 15) / 16)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream_), wpanel, upk, Cin, N, taps);
    return (int)hipGetLastError();
}

extern "C" int lwg_winograd_panels_f32(const LwgWinoDesc* descs_dev, int ndesc, int total_blocks, lwg_stream_t stream_) {
    if (!descs_dev || ndesc < 1 || total_blocks < 1) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(lwg_winograd_panels_kernel, dim3((unsigned)total_blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_), descs_dev, ndesc);
    return (int)hipGetLastError();
}


// args: the launch description of the 3 x 3 / stride 1 / pad 1 convolution as lwg_conv2d_nhwc_f32 takes it (nine taps, omul = 1, OH = H, OW = W,
// one or two inputs with C0 % 8 == 0, C1 % 8 == 0 and (C0 + C1) % 16 == 0, N % 64 == 0, YC % 4 == 0; LWG_EPI_NONE, LWG_EPI_RESIDUAL (ycoff % 4 == 0)
// or LWG_EPI_SPADE (N = 2 YC, columns gamma | beta interleaved in blocks of 32, ycoff = 0); any activation of lwg_act; every image of an input
// < 3 GiB) EXCEPT args->w = the Winograd fragment panel Upk[16][Cin/8][2][N][4]: element (p, s, kh, n, kk) = (G w G^T)[xi = p / 4][nu = p % 4] of
// input channel 8 s + 2 kk + kh (concatenated order) and output column n.
static bool lwg_wino_contract(const LwgConvArgs& a);

// Split plan of a training launch (0 = run it whole): a launch whose 64-patch x 32-channel workgroups cover half the CUs or less runs its K loop
// in 2 .. 8 slices of >= 8 stages (64 input channels; each slice pays the prologue and the epilogue again) - twice to eight times the workgroups.
// Plain epilogues and the ReLU-mask data gradient only (what lwg_splitk_finish_kernel finishes).
static int lwg_wino_split_plan(const LwgConvArgs& a, int* stages_per_slice) {
    const bool mask = a.epi == LWG_EPI_RESIDUAL && a.act == LWG_ACT_RELU_MASK;
    if (!LWG_WINO_SPLITK || (a.epi != LWG_EPI_NONE && !mask)) return 0;
    const int bx = (a.W + 2 * TPB - 1) / (2 * TPB), by = (a.H + 2 * TPB - 1) / (2 * TPB);
    const long blocks32 = (long)bx * by * a.B * (a.N / 32);
    if (blocks32 > 128) return 0;
    const int nst = (a.C0 + a.C1) / KS;
    int want = (int)(256 / blocks32);
    if (want > 8) want = 8;
    int sps = (nst + want - 1) / want;
    if (sps < 8) sps = 8;
    sps += sps & 1;
    const int slices = (nst + sps - 1) / sps;
    if (slices < 2) return 0;
    *stages_per_slice = sps;
    return slices;
}

extern "C" size_t lwg_conv2d_winograd_ws_floats(const LwgConvArgs* pa) {
    if (!pa || !lwg_wino_contract(*pa)) return 0;
    int sps = 0;
    return (size_t)lwg_wino_split_plan(*pa, &sps) * (size_t)pa->M * (size_t)pa->N;
}

extern "C" int lwg_conv2d_winograd_f32_ws(const LwgConvArgs* pa, float* ws, lwg_stream_t stream_);
extern "C" int lwg_conv2d_winograd_f32(const LwgConvArgs* pa, lwg_stream_t stream_) { return lwg_conv2d_winograd_f32_ws(pa, nullptr, stream_); }

static bool lwg_wino_contract(const LwgConvArgs& a) {
    if (!a.x0 || !a.w || !a.y || a.M <= 0 || a.ntaps != 9 || a.stride != 1 || a.omul != 1 || a.C0 <= 0 || (a.C0 % KS) != 0 || a.C1 < 0 ||
        (a.C1 % KS) != 0 || ((a.C0 + a.C1) % (2 * KS)) != 0 || (a.C1 > 0 && !a.x1) || a.N <= 0 || (a.N % NB) != 0 || a.OH != a.H || a.OW != a.W ||
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
    if ((unsigned long long)a.H * a.W * cmax * 4ull >= (unsigned long long)WINO_OOB || 64ull * (a.C0 + a.C1) * a.N >= 0xffffffffull)
        return false;
    return true;
}

// small launches (a frame or two): blocks of 64 patches x 32 channels when that shortens the launch - the chip holds one workgroup per CU (LDS), a
// launch is ceil(blocks / CUs) rounds, and a half-size block costs ~0.55 of a full one (same per-stage overheads on half the MFMAs).  Same bits.
static bool lwg_wino_small(const LwgConvArgs& a, int cus) {
    const int bx = (a.W + 2 * TPB - 1) / (2 * TPB), by = (a.H + 2 * TPB - 1) / (2 * TPB);
    const long blocks64 = (long)bx * by * a.B * (a.N / NB);
    const long rounds64 = (blocks64 + cus - 1) / cus, rounds32 = (2 * blocks64 + cus - 1) / cus;
    return a.epi != LWG_EPI_SPADE && (double)rounds32 * 0.55 < (double)rounds64;
}

// How the library runs a launch (the host-side mirror of nothing: ops._wino_plan asks instead of recomputing the tile geometry - ADVICE r05):
// *blocks = 64-patch blocks of the launch (x K slices when with_ws and the split plan applies), *slices = K slices (0 = whole), *nbv = output
// channels per block (64 | 32), *workgroups = persistent workgroups launched.  Returns 0, or 1 when the launch does not meet the contract.
extern "C" int lwg_conv2d_winograd_plan(const LwgConvArgs* pa, int with_ws, long long* blocks, int* slices, int* nbv, int* workgroups) {
    if (!pa || !lwg_wino_contract(*pa)) return (int)hipErrorInvalidValue;
    const LwgConvArgs& a = *pa;
    const int bx = (a.W + 2 * TPB - 1) / (2 * TPB), by = (a.H + 2 * TPB - 1) / (2 * TPB);
    const long tiles = (long)bx * by * a.B;
    int sps = 0;
    const int sl = with_ws ? lwg_wino_split_plan(a, &sps) : 0;
    const int cus = lwg_device_cus();
    const int v = sl > 1 ? 32 : (lwg_wino_small(a, cus) ? 32 : NB);
    const long long nb = (long long)tiles * (a.N / v) * (sl > 1 ? sl : 1);
    if (blocks) *blocks = nb;
    if (slices) *slices = sl > 1 ? sl : 0;
    if (nbv) *nbv = v;
    if (workgroups) *workgroups = (int)(sl > 1 ? nb : (nb < cus ? nb : cus));
    return 0;
}

// ws: NULL, or lwg_conv2d_winograd_ws_floats(args) floats - a launch that would leave half the chip idle then runs split over K through it.
extern "C" int lwg_conv2d_winograd_f32_ws(const LwgConvArgs* pa, float* ws, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!pa || !lwg_wino_contract(*pa)) return (int)hipErrorInvalidValue;
    const LwgConvArgs& a = *pa;
    const size_t loop = (size_t)LOOP_FLOATS * 4, epi = (size_t)MS_FLOATS * 4;
    const size_t lds = loop > epi ? loop : epi;
    const int bx = (a.W + 2 * TPB - 1) / (2 * TPB), by = (a.H + 2 * TPB - 1) / (2 * TPB);
    const int cus = lwg_device_cus();
    const bool small = lwg_wino_small(a, cus);
    static unsigned long long done[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const bool two = a.C1 > 0;
    int sps = 0;
    const int slices = ws ? lwg_wino_split_plan(a, &sps) : 0;
    if (slices > 1) {
        auto kern = two ? lwg_conv_winograd_kernel<LWG_EPI_NONE, 32, true, true> : lwg_conv_winograd_kernel<LWG_EPI_NONE, 32, true, false>;
        if (hipError_t e = lwg_allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds, done[two ? 11 : 5]); e != hipSuccess) return (int)e;
        LwgConvArgs part = a;                                // raw sums into the slabs: dense (M, N) rows, no bias / residual / activation
        part.y = ws;
        part.YC = a.N;
        part.ycoff = 0;
        part.bias = nullptr;
        part.epi = LWG_EPI_NONE;
        part.act = LWG_ACT_NONE;
        part.res = nullptr;
        part.cshift = sps;
        hipLaunchKernelGGL(kern, dim3((unsigned)(bx * by * a.B * (a.N / 32)), 1u, (unsigned)slices), dim3(WG_THREADS), lds, stream, part);
        return (int)lwg_splitk_finish_launch(a, ws, slices, stream);
    }
    // persistent workgroups: one per CU (LDS) at most, each walking block ids blockIdx.x + k gridDim.x (LWG_WINO_PERSIST = 0: one block per workgroup)
    const long total = (long)bx * by * a.B * (a.N / (small ? 32 : NB));
    const dim3 grid((unsigned)(LWG_WINO_PERSIST && total > cus ? cus : total));
#define LWG_WINO_GO2(E, V, T, SLOT)                                                                                                   \
    {                                                                                                                                 \
        if (hipError_t e = lwg_allow_dynamic_lds(reinterpret_cast<const void*>(lwg_conv_winograd_kernel<E, V, false, T>), lds, done[SLOT]); e != hipSuccess) \
            return (int)e;                                                                                                            \
        hipLaunchKernelGGL((lwg_conv_winograd_kernel<E, V, false, T>), grid, dim3(WG_THREADS), lds, stream, a);                       \
    }
#define LWG_WINO_GO(E, V, SLOT)                                                                                                       \
    {                                                                                                                                 \
        if (two) LWG_WINO_GO2(E, V, true, SLOT + 6) else LWG_WINO_GO2(E, V, false, SLOT)                                              \
    }
    if (a.epi == LWG_EPI_SPADE) LWG_WINO_GO(LWG_EPI_SPADE, 64, 2)
    else if (a.epi == LWG_EPI_RESIDUAL) {
        if (small) LWG_WINO_GO(LWG_EPI_RESIDUAL, 32, 3)
        else LWG_WINO_GO(LWG_EPI_RESIDUAL, 64, 1)
    } else {
        if (small) LWG_WINO_GO(LWG_EPI_NONE, 32, 4)
        else LWG_WINO_GO(LWG_EPI_NONE, 64, 0)
    }
#undef LWG_WINO_GO
#undef LWG_WINO_GO2
    return (int)hipGetLastError();
}
