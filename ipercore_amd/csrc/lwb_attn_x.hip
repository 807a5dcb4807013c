// Liquid Warping Block, attention form, with the QUERY PROJECTION FOLDED INTO THE SOURCE SIDE ("x form").
//
// Reference semantics (generators/attlwb_spade_resunet.py:106-139 SelfAttentionBlock, :175-191 LWB, :208-252 SelfAttentionLWB):
//     q      = Wq x + bq                                   (fq, 1x1 conv on the transfer feature x)
//     K_s    = warp_s(Wk f_s) + bk,   V_s = warp_s(Wv f_s) + bv          (fk / fv after the bilinear warp; csrc/lwb_attn.hip hoists them)
//     l_s    = K_s . q / sqrt(C),     a = softmax_s(l),     out = sum_s a_s V_s
// The warp is linear (zero padding), so
//     K_s . q = warp_s(Wq^T Wk f_s) . x  +  warp_s(bq . Wk f_s)  +  bk . (Wq x + bq)
// and the last term does not depend on s: it cancels in the softmax.  With  Kq_s = (Wq^T Wk) f_s  (C channels) and
// kappa_s = (Wk^T bq) . f_s  (one scalar per source texel) - both functions of the cached source features only, computed ONCE per source -
//     l_s = [ warp_s(Kq_s) . x + warp_s(kappa_s) ] / sqrt(C)      out = sum_s a_s warp_s(Vs_s) + bv
// The per-frame fq convolution (9 launches and 8 GFLOP per 512x512 frame), the q tensor (one write + one read per site) and bk disappear.
//
// One workgroup = one 8 x 8-pixel tile of one frame (tile-major, frame-minor order: lwg_common.h), 4 waves; a pixel = LPP lanes of 16 bytes
// (4 fp32 / 8 bf16 channels); the tile is walked in passes of 256 / LPP pixels, the x row of the next pass in flight during the gathers of
// the current one.  A wave whose pixels ALL have no in-image tap for a source (background: flow = -2) issues no gathers for it.
// STATS: the kernel reads every element of x exactly once, so it also leaves the InstanceNorm partial statistics of x (SPADE's
// parameter-free norm, attlwb_spade_resunet.py:62,83) - one (count, mean, M2) record per (frame, tile, channel), merged by
// lwg_in_stats_final (norm.hip) - and the separate statistics pass over x (9 launches, 9 full reads per frame) disappears.
#include "lwg_common.h"
#include "lwg_conv_args.h"

typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));

template <typename ST> struct AttnXTraits;
template <> struct AttnXTraits<float> {
    static constexpr int CPL = 4;
    __device__ static __forceinline__ void unpack(const uintx4 v, float (&f)[4]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned u = v[k];           // (a bit_cast applied to the vector-element lvalue itself reads element 0 for every k)
            f[k] = __builtin_bit_cast(float, u);
        }
    }
    __device__ static __forceinline__ uintx4 pack(const float (&f)[4]) {
        uintx4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = __builtin_bit_cast(unsigned, f[k]);
        return v;
    }
};
template <> struct AttnXTraits<__bf16> {
    static constexpr int CPL = 8;
    __device__ static __forceinline__ void unpack(const uintx4 v, float (&f)[8]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            f[2 * k] = __builtin_bit_cast(float, v[k] << 16);
            f[2 * k + 1] = __builtin_bit_cast(float, v[k] & 0xffff0000u);
        }
    }
    __device__ static __forceinline__ uintx4 pack(const float (&f)[8]) {
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        uintx4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bf16x2 p;
            p[0] = (__bf16)f[2 * k];
            p[1] = (__bf16)f[2 * k + 1];
            v[k] = __builtin_bit_cast(unsigned, p);
        }
        return v;
    }
};

// NSU: sources handled with their flows prefetched a pass ahead and the source loop unrolled (2 = the runner's default num_source,
// deploy.toml:7); 0: any ns, flows loaded inside the loop.
// PW: passes in flight per workgroup.  1: four waves walk the tile's passes one after the other (frame batches: thousands of
// workgroups).  4: sixteen waves, four passes at a time - the form for launches of a few frames, where a 64 x 64 feature map is 64
// workgroups per frame and a workgroup's chain of passes is what the launch waits for.  Both forms leave the SAME statistics record,
// bit for bit: the per-lane sums run over the tile's passes in pass order in either (PW = 4 replays that order through LDS), so a frame
// does not depend on which form its batch size selected.
template <typename ST, int LPP, bool STATS, int NSU, int PW, int OCC, bool PIPE = false>
__global__ __launch_bounds__(256 * PW, PW == 1 ? OCC : 1) void lwg_lwb_attnx_kernel(const ST* __restrict__ x, const ST* __restrict__ Kq,
                                                                 const float* __restrict__ kappa, const ST* __restrict__ Vs,
                                                                 const float* __restrict__ bv, const float* __restrict__ T,
                                                                 ST* __restrict__ out, float* __restrict__ stats, int B, int ns, int h, int w,
                                                                 int src_batched) {
    using TR = AttnXTraits<ST>;
    constexpr int CPL = TR::CPL;
    constexpr int C = CPL * LPP;
    constexpr int PPW = 64 / LPP;          // pixels per wave and pass
    constexpr int PPP = 4 * PPW;           // pixels per pass (four waves)
    constexpr int NPASS = 64 / PPP;
    constexpr int NIT = NPASS / PW;        // passes per wave
    static_assert(NPASS % PW == 0, "PW must divide the tile's pass count");
    constexpr int NT = NSU > 0 ? NSU : 1;
    constexpr unsigned ROWB = (unsigned)C * sizeof(ST);      // bytes of one pixel row
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int wq = wid & 3, pl = wid >> 2;                   // position inside a pass, pass slot
    const int cl = lane % LPP, pg = lane / LPP;
    const int tiles_x = (w + 7) >> 3;
    // Work order.  The frames of a tile run back to back on ONE XCD (they gather the same source texels: fetched into that L2 once), and
    // the TILES are dealt round-robin over the eight XCDs (hardware: workgroup id % 8): the body covers ~12 % of the pixels, in the middle
    // of the image - with a contiguous band of tile rows per XCD (the order of the first version) the XCDs holding the middle bands did
    // nearly all the gathers while the others ran out of work.
    int tile, b;
    if (LWG_ATTNX_INTERLEAVE) {
        const int xcd = blockIdx.x & 7;
        const long idx = blockIdx.x >> 3;
        const long grp = idx / B;
        tile = (int)(grp * 8 + xcd);
        b = (int)(idx - grp * B);
        if (tile >= tiles_x * ((h + 7) >> 3)) return;             // padding of the last group of eight tiles (workgroup-uniform)
    } else {
        const long L = lwg_xcd_remap(blockIdx.x, gridDim.x);       // (tile, frame), tile-major / frame-minor (lwg_common.h)
        tile = (int)(L / B);
        b = (int)(L - (long)tile * B);
    }
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int hw = h * w;
    const unsigned nsrc = (unsigned)(src_batched ? B * ns : ns);
    __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc(const_cast<ST*>(Kq), 0, (int)(nsrc * (unsigned)hw * ROWB), 0x00020000);
    __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc(const_cast<ST*>(Vs), 0, (int)(nsrc * (unsigned)hw * ROWB), 0x00020000);
    __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(kappa), 0, (int)(nsrc * (unsigned)hw * 4u), 0x00020000);
    const float inv_sqrt_c = 1.0f / sqrtf((float)C);
    constexpr bool BV_REG = CPL == 4 || LWG_ATTNX_BVREG16;       // bf16 (8 channels per lane): bv is re-read per pass instead of held (registers)
    float bvr[CPL];
    if (BV_REG) {
#pragma unroll
        for (int k = 0; k < CPL; ++k) bvr[k] = bv[CPL * cl + k];
    }
    const float2* Tb = reinterpret_cast<const float2*>(T) + (size_t)b * ns * hw;

    auto pixel_of = [&](int pass, int& yc, int& xc) -> bool {     // clamped to the tile's first pixel when outside the image
        const int p = pass * PPP + wq * PPW + pg;
        const int y = ty * 8 + (p >> 3), xx = tx * 8 + (p & 7);
        const bool live = y < h && xx < w;
        yc = live ? y : ty * 8;
        xc = live ? xx : tx * 8;
        return live;
    };
    // The tile's x rows and flows go through LDS.  x: every wave requests the rows of ALL its passes at once with LDS-DMA
    // (buffer_load ... lds: no registers held while in flight; a wave-instruction = 64 lanes x 16 B = the 1 KB it will read back itself, lane
    // for lane) - one round trip for the tile's 64 x C values instead of one per pass; with a register prefetch a wave kept ONE row in
    // flight and the background tiles (88 % of them: no gathers at all) streamed at a fraction of the HBM rate.
    extern __shared__ __attribute__((aligned(16))) char smem_ax[];
    char* xs = smem_ax;                                                              // [NPASS][4 waves][64 lanes][16 B]
    float2* Tl = reinterpret_cast<float2*>(smem_ax + NPASS * 4096);                   // [ns][64 pixels]
    {
        __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<ST*>(x) + (size_t)b * hw * C, 0, (int)((unsigned)hw * ROWB), 0x00020000);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int pass = it * PW + pl;
            int yc, xc;
            pixel_of(pass, yc, xc);
            const unsigned voff = (unsigned)(yc * w + xc) * ROWB + (unsigned)(CPL * cl) * (unsigned)sizeof(ST);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, (__attribute__((address_space(3))) void*)(xs + (pass * 4 + wq) * 1024), 16, (int)voff, 0, 0, 0);
        }
        for (int i = threadIdx.x; i < ns * 64; i += 256 * PW) {
            const int s = i >> 6, p = i & 63;
            const int y = ty * 8 + (p >> 3), xx = tx * 8 + (p & 7);
            const bool in = y < h && xx < w;
            Tl[i] = Tb[(size_t)s * hw + (size_t)(in ? y : ty * 8) * w + (in ? xx : tx * 8)];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                            // the DMA writes are tracked by vmcnt only
        __syncthreads();
    }
    auto x_row = [&](int pass) -> uintx4 { return *reinterpret_cast<const uintx4*>(xs + ((pass * 4 + wq) * 64 + lane) * 16); };

    // InstanceNorm partial statistics of x over the tile: shifted sums (shift = the tile's first pixel); per lane over the passes in
    // pass order, then the pixel groups of a wave, then the four waves - a fixed order
    float shift[CPL], s1[CPL], s2[CPL];
    if (STATS) {
        TR::unpack(*reinterpret_cast<const uintx4*>(xs + cl * 16), shift);          // row of (pass 0, wave 0, pixel group 0) = the tile's first pixel
#pragma unroll
        for (int k = 0; k < CPL; ++k) s1[k] = s2[k] = 0.f;
    }
    int nlive = 0;
    auto stats_add = [&](const float (&xv)[CPL]) {
        ++nlive;
        if (PW == 1) {
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                const float d = xv[k] - shift[k];
                s1[k] += d;
                s2[k] = __builtin_fmaf(d, d, s2[k]);
            }
        }
    };
    // grid_sample, align_corners=False: pixel = ((g + 1) * size - 1) / 2; the four bilinear weights and the top-left tap (clamped before the int
    // conversion so wild flows cannot overflow; out-of-range taps read zeros)
    auto taps_of = [&](const float2 t, float (&wt)[4], int& tx0, int& ty0) {
        const float ix = ((t.x + 1.f) * (float)w - 1.f) * 0.5f, iy = ((t.y + 1.f) * (float)h - 1.f) * 0.5f;
        const float fx0 = floorf(ix), fy0 = floorf(iy);
        const float wx1 = ix - fx0, wy1 = iy - fy0, wx0 = (fx0 + 1.f) - ix, wy0 = (fy0 + 1.f) - iy;
        tx0 = (int)fminf(fmaxf(fx0, -2.f), (float)w + 1.f);
        ty0 = (int)fminf(fmaxf(fy0, -2.f), (float)h + 1.f);
        wt[0] = wy0 * wx0; wt[1] = wy0 * wx1; wt[2] = wy1 * wx0; wt[3] = wy1 * wx1;
    };
    // one source's logit / value folded into the running softmax of a pixel (online form, the sources in order)
    auto fold = [&](const float (&ka)[CPL], const float (&va)[CPL], float kap, const float (&xv)[CPL], float& mrun, float& lrun, float (&o)[CPL]) {
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < CPL; ++k) dot = __builtin_fmaf(ka[k], xv[k], dot);
#pragma unroll
        for (int off = LPP >> 1; off > 0; off >>= 1) dot += __shfl_xor(dot, off, 64);
        const float logit = (dot + kap) * inv_sqrt_c;
        const float mnew = fmaxf(mrun, logit);
        const float corr = expf(mrun - mnew);  // exp(-inf) = 0 on the first source
        const float pr = expf(logit - mnew);
        lrun = __builtin_fmaf(lrun, corr, pr);
#pragma unroll
        for (int k = 0; k < CPL; ++k) o[k] = __builtin_fmaf(o[k], corr, pr * va[k]);
        mrun = mnew;
    };
    auto finish = [&](long gp, float lrun, const float (&o)[CPL]) {
        const float invl = 1.f / lrun;
        float r[CPL];
        if (!BV_REG) {
#pragma unroll
            for (int k = 0; k < CPL; k += 4) {
                const floatx4 b4 = *reinterpret_cast<const floatx4*>(bv + CPL * cl + k);
#pragma unroll
                for (int j = 0; j < 4; ++j) bvr[k + j] = b4[j];
            }
        }
#pragma unroll
        for (int k = 0; k < CPL; ++k) r[k] = __builtin_fmaf(o[k], invl, bvr[k]);
        *reinterpret_cast<uintx4*>(out + gp * C + CPL * cl) = TR::pack(r);
    };

    if constexpr (PIPE) {
        // Body tiles in TWO passes at a time (round 6; fp32 C = 256, where the tile's 64 KB of x rows hold the CU to two workgroups = two waves per SIMD
        // and the registers of the other six are free): the sixteen gathers of pass p + 1 are REQUESTED before pass p is folded, so a wave's chain is
        // no longer (flow -> gathers -> softmax) x 16 one after the other.  The loads of a pass are unconditional (taps outside the image / pixels
        // outside the tile: out-of-range offsets, zero fill) so that the compiler's vmcnt counts stay exact; a tile WITHOUT any in-image tap (88 % of
        // them) takes the branch below instead: x in, bv out, statistics - the values the general path computes for it (every logit 0, every value 0).
        static_assert(NSU == 2 && PW == 1 && (NIT % 2) == 0, "the pipelined form is the ns = 2 frame-batch form");
        bool mine = false;
#pragma unroll 1
        for (int it = 0; it < NIT; ++it) {
            int yc, xc;
            const bool live = pixel_of(it, yc, xc);
            const int ptile = it * PPP + wq * PPW + pg;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const float2 t = Tl[s * 64 + ptile];
                float wt[4];
                int tx0, ty0;
                taps_of(t, wt, tx0, ty0);
                mine |= live && tx0 >= -1 && tx0 < w && ty0 >= -1 && ty0 < h;
            }
        }
        if (__syncthreads_or(mine ? 1 : 0) == 0) {
#pragma unroll 1
            for (int it = 0; it < NIT; ++it) {
                int yc, xc;
                const bool live = pixel_of(it, yc, xc);
                float xv[CPL];
                TR::unpack(x_row(it), xv);
                if (live) {
                    if (STATS) stats_add(xv);
                    float o[CPL];
#pragma unroll
                    for (int k = 0; k < CPL; ++k) o[k] = 0.f;
                    finish(((long)b * h + yc) * w + xc, 2.f, o);
                }
            }
        } else {
            struct Taps { uintx4 kr[2][4], vr[2][4]; float ar[2][4], wt[2][4]; };
            auto issue = [&](int pass, Taps& g) {
                int yc, xc;
                const bool live = pixel_of(pass, yc, xc);
                const int ptile = pass * PPP + wq * PPW + pg;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    int tx0, ty0;
                    taps_of(Tl[s * 64 + ptile], g.wt[s], tx0, ty0);
                    const unsigned pbase = (unsigned)(src_batched ? b * ns + s : s) * (unsigned)hw;
#pragma unroll
                    for (int tp = 0; tp < 4; ++tp) {
                        const int tyy = ty0 + (tp >> 1), txx = tx0 + (tp & 1);
                        const bool ok = live && tyy >= 0 && tyy < h && txx >= 0 && txx < w;
                        const unsigned pix = pbase + (unsigned)(tyy * w + txx);
                        const unsigned voff = ok ? pix * ROWB + (unsigned)(CPL * cl) * (unsigned)sizeof(ST) : 0xC0000000u;
                        g.kr[s][tp] = __builtin_bit_cast(uintx4, __builtin_amdgcn_raw_buffer_load_b128(rK, (int)voff, 0, 0));
                        g.vr[s][tp] = __builtin_bit_cast(uintx4, __builtin_amdgcn_raw_buffer_load_b128(rV, (int)voff, 0, 0));
                        g.ar[s][tp] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rA, ok ? (int)(pix * 4u) : (int)0xC0000000u, 0, 0));
                    }
                }
            };
            auto consume = [&](int pass, const Taps& g) {
                int yc, xc;
                const bool live = pixel_of(pass, yc, xc);
                float xv[CPL];
                TR::unpack(x_row(pass), xv);
                if (STATS && live) stats_add(xv);
                float mrun = -INFINITY, lrun = 0.f;
                float o[CPL];
#pragma unroll
                for (int k = 0; k < CPL; ++k) o[k] = 0.f;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    float ka[CPL], va[CPL], kap = 0.f;
#pragma unroll
                    for (int k = 0; k < CPL; ++k) ka[k] = va[k] = 0.f;
#pragma unroll
                    for (int tp = 0; tp < 4; ++tp) {
                        float k8[CPL], v8[CPL];
                        TR::unpack(g.kr[s][tp], k8);
                        TR::unpack(g.vr[s][tp], v8);
#pragma unroll
                        for (int k = 0; k < CPL; ++k) {
                            ka[k] = __builtin_fmaf(k8[k], g.wt[s][tp], ka[k]);
                            va[k] = __builtin_fmaf(v8[k], g.wt[s][tp], va[k]);
                        }
                        kap = __builtin_fmaf(g.ar[s][tp], g.wt[s][tp], kap);
                    }
                    fold(ka, va, kap, xv, mrun, lrun, o);
                }
                if (live) finish(((long)b * h + yc) * w + xc, lrun, o);
            };
            Taps g0, g1;
            issue(0, g0);
#pragma unroll 1
            for (int it = 0; it < NIT; it += 2) {
                issue(it + 1, g1);
                consume(it, g0);
                issue(it + 2 < NIT ? it + 2 : it + 1, g0);        // (past the end: a harmless re-request of the last pass - no branch, exact vmcnt counts)
                consume(it + 1, g1);
            }
        }
    } else {
#pragma unroll 1
    for (int it = 0; it < NIT; ++it) {
        const int pass = it * PW + pl;
        int yc, xc;
        const bool live = pixel_of(pass, yc, xc);
        const long gp = ((long)b * h + yc) * w + xc;
        const int ptile = pass * PPP + wq * PPW + pg;                             // this lane's pixel of the tile
        float xv[CPL];
        TR::unpack(x_row(pass), xv);
        if (STATS && live) stats_add(xv);
        float mrun = -INFINITY, lrun = 0.f;
        float o[CPL];
#pragma unroll
        for (int k = 0; k < CPL; ++k) o[k] = 0.f;

        auto source = [&](int s, const float2 t) {
            float wt[4];
            int tx0, ty0;
            taps_of(t, wt, tx0, ty0);
            const bool any_tap = live && tx0 >= -1 && tx0 < w && ty0 >= -1 && ty0 < h;
            float ka[CPL], va[CPL], kap = 0.f;
#pragma unroll
            for (int k = 0; k < CPL; ++k) ka[k] = va[k] = 0.f;
            if (__builtin_amdgcn_ballot_w64(any_tap) != 0ull) {      // wave-uniform: background waves issue no gathers
                const unsigned sidx = (unsigned)(src_batched ? b * ns + s : s);
                const unsigned pbase = sidx * (unsigned)hw;
                uintx4 kr[4], vr[4];
                float ar[4];
#pragma unroll
                for (int tp = 0; tp < 4; ++tp) {
                    const int tyy = ty0 + (tp >> 1), txx = tx0 + (tp & 1);
                    const bool ok = live && tyy >= 0 && tyy < h && txx >= 0 && txx < w;
                    const unsigned pix = pbase + (unsigned)(tyy * w + txx);
                    const unsigned voff = ok ? pix * ROWB + (unsigned)(CPL * cl) * (unsigned)sizeof(ST) : 0xC0000000u;
                    kr[tp] = __builtin_bit_cast(uintx4, __builtin_amdgcn_raw_buffer_load_b128(rK, (int)voff, 0, 0));
                    vr[tp] = __builtin_bit_cast(uintx4, __builtin_amdgcn_raw_buffer_load_b128(rV, (int)voff, 0, 0));
                    ar[tp] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rA, ok ? (int)(pix * 4u) : (int)0xC0000000u, 0, 0));
                }
#pragma unroll
                for (int tp = 0; tp < 4; ++tp) {
                    float k8[CPL], v8[CPL];
                    TR::unpack(kr[tp], k8);
                    TR::unpack(vr[tp], v8);
#pragma unroll
                    for (int k = 0; k < CPL; ++k) {
                        ka[k] = __builtin_fmaf(k8[k], wt[tp], ka[k]);
                        va[k] = __builtin_fmaf(v8[k], wt[tp], va[k]);
                    }
                    kap = __builtin_fmaf(ar[tp], wt[tp], kap);
                }
            }
            fold(ka, va, kap, xv, mrun, lrun, o);
        };
        if (NSU > 0) {
#pragma unroll
            for (int s = 0; s < NT; ++s) source(s, Tl[s * 64 + ptile]);
        } else {
#pragma unroll 1
            for (int s = 0; s < ns; ++s) source(s, Tl[s * 64 + ptile]);
        }
        if (live) finish(gp, lrun, o);
    }
    }

    if (STATS) {
        if (PW > 1) {
            // replay the per-lane sums in pass order: pass p belongs to the waves of slot p % PW; they add it to the lane's running sums
            // in LDS - exactly the additions (and the order) the PW = 1 form performs in registers
            __shared__ float acc[4][64][2 * CPL];
            if (pl == 0) {
#pragma unroll
                for (int k = 0; k < 2 * CPL; ++k) acc[wq][lane][k] = 0.f;
            }
            __syncthreads();
#pragma unroll 1
            for (int p = 0; p < NPASS; ++p) {
                if (pl == p % PW) {
                    int yc, xc;
                    if (pixel_of(p, yc, xc)) {
                        float xv[CPL];
                        TR::unpack(x_row(p), xv);
#pragma unroll
                        for (int k = 0; k < CPL; ++k) {
                            const float d = xv[k] - shift[k];
                            acc[wq][lane][k] += d;
                            acc[wq][lane][CPL + k] = __builtin_fmaf(d, d, acc[wq][lane][CPL + k]);
                        }
                    }
                }
                __syncthreads();
            }
            if (pl == 0) {
#pragma unroll
                for (int k = 0; k < CPL; ++k) {
                    s1[k] = acc[wq][lane][k];
                    s2[k] = acc[wq][lane][CPL + k];
                }
            }
        }
        __syncthreads();                                                  // every wave is done with the x rows: their LDS is reused below
        float (*sh)[2][C] = reinterpret_cast<float (*)[2][C]>(smem_ax);    // [4][2][C]
        float* shn = reinterpret_cast<float*>(smem_ax + 4 * 2 * C * sizeof(float));   // [4 PW]
        // lanes holding the same channels: pixel groups of a wave (xor shuffles), then the four waves through LDS
        float fn = (float)nlive;                                      // small integers: exact in any order
#pragma unroll
        for (int off = LPP; off < 64; off <<= 1) {
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                s1[k] += __shfl_xor(s1[k], off, 64);
                s2[k] += __shfl_xor(s2[k], off, 64);
            }
            fn += __shfl_xor(fn, off, 64);
        }
        if (pg == 0) {
            if (pl == 0) {
#pragma unroll
                for (int k = 0; k < CPL; ++k) {
                    sh[wq][0][CPL * cl + k] = s1[k];
                    sh[wq][1][CPL * cl + k] = s2[k];
                }
            }
            if (cl == 0) shn[wid] = fn;
        }
        __syncthreads();
        if (wid == 0 && pg == 0) {
            float tn = 0.f;
#pragma unroll
            for (int k = 0; k < 4 * PW; ++k) tn += shn[k];               // live pixels of the tile
            const int nrec = tiles_x * ((h + 7) >> 3);
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                const int c = CPL * cl + k;
                const float t1 = ((sh[0][0][c] + sh[1][0][c]) + sh[2][0][c]) + sh[3][0][c];
                const float t2 = ((sh[0][1][c] + sh[1][1][c]) + sh[2][1][c]) + sh[3][1][c];
                const float mean = shift[k] + t1 / tn;                  // tn >= 1: a tile's first pixel is inside the image
                const float m2 = t2 - t1 * t1 / tn;
                float* rec = stats + (((size_t)b * nrec + tile) * C + c) * 3;
                rec[0] = tn;
                rec[1] = mean;
                rec[2] = m2 > 0.f ? m2 : 0.f;
            }
        }
    }
}

// records per image the STATS form writes for an (h, w, C) launch of storage width esz (4: fp32, 2: bf16): one per 8 x 8 tile
extern "C" int lwg_lwb_attention_x_records(int h, int w, int C, int esz) {
    if (h <= 0 || w <= 0 || C <= 0 || (esz != 2 && esz != 4)) return 0;
    const int lpp = C / (16 / esz);
    if (lpp < 8 || lpp > 64 || (lpp & (lpp - 1)) != 0) return 0;
    return ((w + 7) >> 3) * ((h + 7) >> 3);
}

template <typename ST>
static int lwg_attnx_launch(const ST* x, const ST* Kq, const float* kappa, const ST* Vs, const float* bv, const float* T, ST* out, float* stats,
                            int B, int ns, int h, int w, int C, int src_batched, hipStream_t stream) {
    if (!x || !Kq || !kappa || !Vs || !bv || !T || !out || B <= 0 || ns <= 0 || ns > 64 || h <= 0 || w <= 0) return (int)hipErrorInvalidValue;
    const unsigned long long nsrc = (unsigned long long)(src_batched ? B * ns : ns);
    // the taps go through 32-bit buffer offsets with 0xC0000000 as the out-of-range marker (zero fill): K / V below 3 GiB each
    if (nsrc * (unsigned long long)h * w * (unsigned long long)C * sizeof(ST) >= 0xC0000000ull) return (int)hipErrorInvalidValue;
    const long tiles = (long)((w + 7) >> 3) * ((h + 7) >> 3);
    constexpr int CPL = AttnXTraits<ST>::CPL;
    const int npass = (C / CPL) / 4;
    const dim3 grid((unsigned)((LWG_ATTNX_INTERLEAVE ? (tiles + 7) / 8 * 8 : tiles) * B));
    // a few frames: fewer workgroups than two per CU, each with a chain of >= 4 passes -> sixteen waves per tile (same values, see the kernel)
    const bool wide = LWG_ATTNX_WIDE && tiles * B < 512 && npass >= 4;
#define LWG_ATTNX_GO(LPP, ST_, NSU, PW)                                                                                              \
    {                                                                                                                                \
        /* fp32 C = 256, ns = 2, frame batches: the pipelined body form (two workgroups per CU by LDS: 256 registers per wave) */     \
        constexpr bool pipe = LWG_ATTNX_PIPE && CPL == 4 && (LPP) == 64 && (NSU) == 2 && (PW) == 1;                                  \
        auto kern = lwg_lwb_attnx_kernel<ST, LPP, ST_, NSU, PW, (pipe ? 2 : CPL == 4 ? LWG_ATTNX_OCC : LWG_ATTNX_OCC16), pipe>;        \
        const size_t lds = (size_t)((LPP) / 4) * 4096 + (size_t)ns * 512;        /* x rows of the tile + its flows */                 \
        static unsigned long long lds_ok = 0;                                                                                        \
        if (lds > 65536) {                                                                                                           \
            const hipError_t e = lwg_allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds, lds_ok);                            \
            if (e != hipSuccess) return (int)e;                                                                                      \
        }                                                                                                                            \
        hipLaunchKernelGGL(kern, grid, dim3(256 * PW), lds, stream, x, Kq, kappa, Vs, bv, T, out, stats, B, ns, h, w, src_batched);   \
    }
#define LWG_ATTNX_PW(LPP, ST_, NSU)                                                                                                  \
    {                                                                                                                                \
        if constexpr ((LPP) / 4 >= 4) {                                                                                              \
            if (wide) LWG_ATTNX_GO(LPP, ST_, NSU, 4) else LWG_ATTNX_GO(LPP, ST_, NSU, 1)                                             \
        } else {                                                                                                                     \
            LWG_ATTNX_GO(LPP, ST_, NSU, 1)                                                                                           \
        }                                                                                                                            \
    }
#define LWG_ATTNX_LAUNCH(LPP)                                                                                                        \
    if (stats) { if (LWG_ATTNX_NSU2 && ns == 2) LWG_ATTNX_PW(LPP, true, 2) else LWG_ATTNX_PW(LPP, true, 0) }                         \
    else { if (LWG_ATTNX_NSU2 && ns == 2) LWG_ATTNX_PW(LPP, false, 2) else LWG_ATTNX_PW(LPP, false, 0) }
    switch (C / CPL) {
        case 8: LWG_ATTNX_LAUNCH(8) break;
        case 16: LWG_ATTNX_LAUNCH(16) break;
        case 32: LWG_ATTNX_LAUNCH(32) break;
        case 64: if constexpr (CPL == 4) { LWG_ATTNX_LAUNCH(64) break; }
                 return (int)hipErrorInvalidValue;
        default: return (int)hipErrorInvalidValue;
    }
#undef LWG_ATTNX_LAUNCH
#undef LWG_ATTNX_PW
#undef LWG_ATTNX_GO
    return (int)hipGetLastError();
}

// x (B,h,w,C) transfer feature; Kq (nsrc,h,w,C) = (Wq^T Wk) f_src; kappa (nsrc,h,w) = (Wk^T bq) . f_src; Vs (nsrc,h,w,C) = Wv f_src (no bias);
// bv (C); T (B,ns,h,w,2) flows ALREADY RESIZED to (h,w) (lwg_flow_resize_f32), grid_sample coordinates, -2 = background; out (B,h,w,C).
// stats: nullptr, or the record buffer of lwg_instnorm_finalize_f32 (B * nrec * C * 3 floats of records, nrec =
// lwg_lwb_attention_x_records(h, w, C, element size), + its scratch) receiving the per-tile InstanceNorm records of x (count, mean, M2).
// C in {32, 64, 128, 256} (bf16: {64, 128, 256}).
extern "C" int lwg_lwb_attention_x_f32(const float* x, const float* Kq, const float* kappa, const float* Vs, const float* bv, const float* T,
                                       float* out, float* stats, int B, int ns, int h, int w, int C, int src_batched, lwg_stream_t stream_) {
    if (C != 32 && C != 64 && C != 128 && C != 256) return (int)hipErrorInvalidValue;
    return lwg_attnx_launch<float>(x, Kq, kappa, Vs, bv, T, out, stats, B, ns, h, w, C, src_batched, reinterpret_cast<hipStream_t>(stream_));
}

extern "C" int lwg_lwb_attention_x_bf16(const void* x, const void* Kq, const float* kappa, const void* Vs, const float* bv, const float* T,
                                        void* out, float* stats, int B, int ns, int h, int w, int C, int src_batched, lwg_stream_t stream_) {
    if (C != 64 && C != 128 && C != 256) return (int)hipErrorInvalidValue;
    return lwg_attnx_launch<__bf16>(static_cast<const __bf16*>(x), static_cast<const __bf16*>(Kq), kappa, static_cast<const __bf16*>(Vs), bv, T,
                                    static_cast<__bf16*>(out), stats, B, ns, h, w, C, src_batched, reinterpret_cast<hipStream_t>(stream_));
}

// Merge of the records the STATS form leaves: ws (B, nrec, C, 3) = (count, mean, M2) -> mean / rstd (B, C).
// One pass of plain sums about a reference r = the mean of the image's FIRST record (close to the image mean, so the moments below do
// not cancel):  A = sum n_i,  S = sum n_i (mean_i - r),  Q = sum [M2_i + n_i (mean_i - r)^2]  ->  mean = r + S / A,  M2 = Q - S^2 / A.
// No chain of pairwise (Chan) updates with a division each.  Workgroup = (64 channels, frame, segment of the records): lane = channel
// (a wave reads 768 contiguous bytes per record), wave k takes the segment's records k, k + 16, ...; the 16 per-wave partials are added in
// wave order, the segments in segment order (second kernel; one segment: written directly).  The order depends on nrec only: a
// frame's statistics do not depend on its batch.
__global__ __launch_bounds__(1024) void lwg_in_stats_merge_seg(const float* __restrict__ ws, int C, int nrec, int nseg, float eps,
                                                              float* __restrict__ part, float* __restrict__ mean, float* __restrict__ rstd) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane, b = blockIdx.y, seg = blockIdx.z;
    const bool cok = c < C;
    const float* base = ws + ((size_t)b * nrec * C + (cok ? c : 0)) * 3;
    const float r = base[1];
    const int per = (nrec + nseg - 1) / nseg;
    const int r0 = seg * per, r1 = min(nrec, r0 + per);
    float sa = 0.f, ss = 0.f, sq = 0.f;
    if (cok) {
        for (int i = r0 + wid; i < r1; i += 16) {
            const float* o = base + (size_t)i * C * 3;
            const float n = o[0], d = o[1] - r;
            sa += n;
            ss = __builtin_fmaf(n, d, ss);
            sq += __builtin_fmaf(n * d, d, o[2]);
        }
    }
    __shared__ float sh[16][3][64];
    sh[wid][0][lane] = sa;
    sh[wid][1][lane] = ss;
    sh[wid][2][lane] = sq;
    __syncthreads();
    if (wid == 0 && cok) {
        float ta = 0.f, ts = 0.f, tq = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) { ta += sh[k][0][lane]; ts += sh[k][1][lane]; tq += sh[k][2][lane]; }
        if (nseg == 1) {
            const float m2 = tq - ts * ts / ta;
            mean[(size_t)b * C + c] = r + ts / ta;
            rstd[(size_t)b * C + c] = 1.0f / sqrtf((m2 > 0.f ? m2 : 0.f) / ta + eps);
        } else {
            float* o = part + (((size_t)b * nseg + seg) * C + c) * 3;
            o[0] = ta; o[1] = ts; o[2] = tq;
        }
    }
}

__global__ __launch_bounds__(256) void lwg_in_stats_merge_fin(const float* __restrict__ ws, const float* __restrict__ part, int BC, int C, int nrec,
                                                             int nseg, float eps, float* __restrict__ mean, float* __restrict__ rstd) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= BC) return;
    const int b = i / C, c = i - b * C;
    const float r = ws[((size_t)b * nrec * C + c) * 3 + 1];
    float ta = 0.f, ts = 0.f, tq = 0.f;
    for (int s = 0; s < nseg; ++s) {
        const float* o = part + (((size_t)b * nseg + s) * C + c) * 3;
        ta += o[0]; ts += o[1]; tq += o[2];
    }
    const float m2 = tq - ts * ts / ta;
    mean[i] = r + ts / ta;
    rstd[i] = 1.0f / sqrtf((m2 > 0.f ? m2 : 0.f) / ta + eps);
}

// ws (B, nrec, C, 3) records (count, mean, M2) -> mean, rstd (B, C): rstd = 1 / sqrt(M2 / n + eps), biased variance (nn.InstanceNorm2d).
// Record 0 of every image must be non-empty (count > 0): it supplies the reference about which the moments are summed.
// More than 512 records: the tail of ws - B * ceil(nrec / 256) * C * 3 floats BEHIND the records - is scratch for the segment partials.
extern "C" size_t lwg_instnorm_finalize_ws_floats(int B, int C, int nrec) {
    if (B <= 0 || C <= 0 || nrec <= 0) return 0;
    const size_t nseg = nrec > 512 ? (size_t)(nrec + 255) / 256 : 0;
    return (size_t)B * nrec * C * 3 + (size_t)B * nseg * C * 3;
}
extern "C" int lwg_instnorm_finalize_f32(float* ws, int B, int C, int nrec, float eps, float* mean, float* rstd, lwg_stream_t stream_) {
    if (!ws || !mean || !rstd || B <= 0 || C <= 0 || nrec <= 0 || B > 65535) return (int)hipErrorInvalidValue;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    const int nseg = nrec > 512 ? (nrec + 255) / 256 : 1;
    if (nseg > 65535) return (int)hipErrorInvalidValue;
    float* part = ws + (size_t)B * nrec * C * 3;
    hipLaunchKernelGGL(lwg_in_stats_merge_seg, dim3((C + 63) / 64, B, nseg), dim3(1024), 0, stream, ws, C, nrec, nseg, eps, part, mean, rstd);
    if (nseg > 1)
        hipLaunchKernelGGL(lwg_in_stats_merge_fin, dim3((B * C + 255) / 256), dim3(256), 0, stream, ws, part, B * C, C, nrec, nseg, eps, mean, rstd);
    return (int)hipGetLastError();
}
