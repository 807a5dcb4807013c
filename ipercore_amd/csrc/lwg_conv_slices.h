// Batch slicing of a convolution launch whose gathered inputs exceed the 32-bit buffer-offset range.
// The kernels address x0 / x1 with raw buffer loads (32-bit byte offsets, 0xC0000000 = the out-of-range marker that makes the hardware
// return zeros for padding taps), so ONE launch can gather from tensors below 3 GiB.  A frame batch whose input is larger is not the
// caller's problem: the entry points cut the launch along the batch dimension into slices that fit and re-base every per-frame pointer
// (frames are independent rows of the implicit GEMM, so the values are those of the single launch bit for bit).
#pragma once
#include "lwg_conv_args.h"

// frames per slice; 0 = the launch fits as it is; -1 = a single frame does not fit (contract violation)
static inline int lwg_conv_slice_frames(const LwgConvArgs& a) {
    const unsigned long long esz = a.xdt == LWG_DT_BF16 ? 2ull : 4ull;
    const unsigned long long per = (unsigned long long)a.H * a.W * (unsigned long long)(a.C0 > a.C1 ? a.C0 : a.C1) * esz;
    if (per == 0ull || (unsigned long long)a.B * per < 0xC0000000ull) return 0;
    const unsigned long long n = (0xC0000000ull - 1ull) / per;
    return n >= 1ull ? (int)n : -1;
}

static inline LwgConvArgs lwg_conv_slice(const LwgConvArgs& a, int b0, int nb) {
    LwgConvArgs s = a;
    const size_t xs = a.xdt == LWG_DT_BF16 ? 2 : 4, ys = a.ydt == LWG_DT_BF16 ? 2 : 4;
    const size_t xpix = (size_t)a.H * a.W, ypix = (size_t)a.YH * a.YW;
    auto adv = [](const float* p, size_t bytes) { return p ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(p) + bytes) : p; };
    s.x0 = adv(a.x0, (size_t)b0 * xpix * a.C0 * xs);
    s.x1 = adv(a.x1, (size_t)b0 * xpix * a.C1 * xs);
    s.y = const_cast<float*>(adv(a.y, (size_t)b0 * ypix * a.YC * ys));
    s.res = adv(a.res, (size_t)b0 * ypix * a.YC * ys);
    s.xn = adv(a.xn, (size_t)b0 * ypix * a.YC * ys);
    s.mean = adv(a.mean, (size_t)b0 * a.YC * 4);
    s.rstd = adv(a.rstd, (size_t)b0 * a.YC * 4);
    s.B = nb;
    s.M = nb * a.OH * a.OW;
    return s;
}

// Runs ``one(slice)`` over the batch slices of ``a`` when it needs slicing: returns 1 and sets *err (0 or the first failure);
// returns 0 when the launch fits and the caller should proceed as usual.
template <class F>
static inline int lwg_conv_run_sliced(const LwgConvArgs& a, F&& one, int* err) {
    const int nbs = lwg_conv_slice_frames(a);
    if (nbs == 0) return 0;
    *err = 1;                                      // hipErrorInvalidValue
    if (nbs < 0 || a.M != a.B * a.OH * a.OW) return 1;
    for (int b0 = 0; b0 < a.B; b0 += nbs) {
        const LwgConvArgs s = lwg_conv_slice(a, b0, a.B - b0 < nbs ? a.B - b0 : nbs);
        const int e = one(s);
        if (e != 0) { *err = e; return 1; }
    }
    *err = 0;
    return 1;
}
