/*
 * lwg_hip.h - C ABI of liblwg_hip.so: hand-written HIP (gfx950 / MI355X) kernels for the per-frame
 * synthesis path of iPERCore's Liquid Warping GAN (Imitator.inference, models/imitator.py:327-395).
 *
 * The reference has no FFI: its boundary for this path is Python calling (a) torch ATen ops and (b) the
 * third-party CUDA extension `neural_renderer` (requirements/build.txt:3).  Each entry point below names the
 * reference call it replaces (file:line relative to the iPERCore checkout).  INTEGRATION.md shows the
 * ctypes binding a maintainer would add on the reference side.
 *
 * Conventions (every function):
 *   - all pointers are DEVICE pointers into buffers owned by the caller (PyTorch-ROCm allocations);
 *     kernels never allocate, never synchronise, and are enqueued on `stream` (a hipStream_t passed as void*);
 *   - tensors are contiguous; activations are NHWC fp32 (bf16 in the *_bf16 entry points); face/vertex/pixel indices are int32;
 *   - the return value is a hipError_t cast to int (0 = hipSuccess; 1 = hipErrorInvalidValue is also used
 *     for contract violations detected on the host side before any launch).
 */
#ifndef LWG_HIP_H
#define LWG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* lwg_stream_t; /* hipStream_t */

/* 10: the panel of lwg_conv_transpose4_winograd_f32 is [4][Cin/8][4][2][9 N] (contiguous per load; round 6, end); 9: lwg_conv2d_winograd4_f32, lwg_winograd4_panel_f32 (F(4x4, 3x3); round 6); 8: lwg_conv2d_winograd_plan, lwg_up4_head_compose_bf16; the Winograd kernels run as persistent workgroups (round 6); 7: lwg_conv_slice_count, lwg_winograd_panel(s)_f32, lwg_crop_resize_bilinear(_bwd)_f32, lwg_conv2d_winograd_f32 contract (Cin % 16, 16-byte output alignment) (round 5); 6: lwg_conv2d_winograd_f32 (round 4); 5: LWG_DT_F32_Q4 output storage of the fp32 convolutions + lwg_head_compose_q4_f32 (round 4); 4: lwg_lwb_attention_x_*, lwg_instnorm_finalize_*;
 * 3: lwg_conv2d_wgrad_unpacked_f32 gained db, lwg_norm_fwd / lwg_norm_bwd gained gstride (round 3); 2: LwgConvArgs.xdt / ydt */
#define LWG_ABI_VERSION 10
int lwg_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * Dense convolution on the matrix cores (fp32 MFMA implicit GEMM), NHWC.
 * Replaces F.conv2d / F.conv_transpose2d as called by
 *   generators/attlwb_spade_resunet.py:18-22 (ResidualBlock), :73-78,:87-92 (SPADE), :202-204 (fq/fk/fv),
 *   :268-271 (Encoder), :331-340,:353-355 (SkipDecoder), generators/bg_inpaintor.py:31-57.
 * ------------------------------------------------------------------------------------------------ */
#define LWG_MAX_TAPS 52
enum { LWG_EPI_NONE = 0, LWG_EPI_RESIDUAL = 1, LWG_EPI_SPADE = 2 };
enum { LWG_ACTIVATION_NONE = 0, LWG_ACTIVATION_RELU = 1, LWG_ACTIVATION_TANH = 2, LWG_ACTIVATION_SIGMOID = 3 };
/* fp32 launches with LWG_EPI_RESIDUAL only: y = res > 0 ? acc + bias : 0.  The data gradient of a convolution whose forward input was
 * relu(...) (res = that input): the ReLU backward of the producing layer rides in the epilogue that writes its output gradient
 * (torch autograd runs it as a separate threshold_backward pass, lwg_trainer.py:326-352 loss.backward()). */
#define LWG_ACTIVATION_RELU_MASK 5
enum { LWG_DT_F32 = 0, LWG_DT_BF16 = 1,     /* activation storage type (BASELINE configs[3]: bf16 activations end to end) */
       /* ydt only, fp32 launches with LWG_EPI_NONE and no split-K workspace: y is written as channel-quad PLANES (B, YC/4, YH, YW, 4) instead of
        * NHWC - the layout lwg_head_compose_q4_f32 stages whole 128-byte lines from (the last decoder layer -> the output head) */
       LWG_DT_F32_Q4 = 2 };

typedef struct LwgConvArgs {
    const float* x0;   /* input, NHWC (B,H,W,C0); each input tensor must be < 3 GiB (32-bit buffer offsets) */
    const float* x1;   /* optional second input concatenated along C (skip connection), (B,H,W,C1), or NULL */
    int C0, C1;        /* Cin = C0 + C1; Cin % 32 == 0 (C0 % 32 == 0 when C1 > 0) or Cin in {4,8,16} with C1 == 0 */
    int B, H, W;       /* input dims */
    int OH, OW;        /* GEMM row grid: rows are (b, oy, ox), oy < OH, ox < OW */
    int M;             /* B*OH*OW */
    int stride;        /* input sample position = (oy*stride + dy[tap], ox*stride + dx[tap]); zero outside */
    int ntaps;         /* number of kernel taps (<= LWG_MAX_TAPS) */
    int cshift;        /* log2(Cin/4) when Cin < 32, else unused */
    const float* w;    /* packed weights [ceil(ntaps*Cin/32)*8][N][4]: w[k/4][n][k%4] with
                          k = ((c/32)*ntaps + tap)*32 + c%32 when Cin % 32 == 0, k = tap*Cin + c when Cin < 32 */
    int N;             /* GEMM columns (Cout, or 2*Cout gamma|beta interleaved by 32 for LWG_EPI_SPADE) */
    const float* bias; /* [N] or NULL */
    float* y;          /* output NHWC (B,YH,YW,YC); row (b,oy,ox) -> pixel (oy*omul+ooy, ox*omul+oox) */
    int YH, YW, YC, ycoff; /* channel n is written at ycoff + n; YC % 4 == 0 and ycoff % 4 == 0 (16-byte stores) */
    int omul, ooy, oox;
    int epi;           /* LWG_EPI_* */
    int act;           /* LWG_ACTIVATION_* applied last */
    const float* res;  /* LWG_EPI_RESIDUAL: tensor shaped like y, added before the activation (LWG_ACTIVATION_RELU_MASK: the mask source) */
    const float* xn;   /* LWG_EPI_SPADE: tensor to normalise (B,YH,YW,YC) */
    const float* mean; /* LWG_EPI_SPADE: (B,YC) instance mean   */
    const float* rstd; /* LWG_EPI_SPADE: (B,YC) 1/sqrt(var+eps) */
    int xdt;           /* LWG_DT_*: storage type of x0 / x1 (the float* fields then point at bf16 data) */
    int ydt;           /* LWG_DT_*: storage type of y, res and xn */
    signed char dy[LWG_MAX_TAPS];
    signed char dx[LWG_MAX_TAPS];
} LwgConvArgs;

int lwg_conv2d_nhwc_f32(const LwgConvArgs* args, lwg_stream_t stream);
/* The same launch with an optional workspace: small-M / large-K launches (one training sample, the discriminator's deep layers:
 * fewer 64x64 output tiles than the chip has room for) are split over K into lwg_conv2d_ws_floats(args) / (M*N) slices whose
 * dense (M,N) slabs a finishing kernel adds in slice order (deterministic) before bias / activation / output geometry.
 * lwg_conv2d_ws_floats returns 0 when the launch would not be split; ws = NULL runs it whole. */
size_t lwg_conv2d_ws_floats(const LwgConvArgs* args);
int lwg_conv2d_nhwc_f32_ws(const LwgConvArgs* args, float* ws, lwg_stream_t stream);
/* Kernel launches behind one call of the convolution entry points above / below for this description: 1, or the number of batch slices when a
 * gathered input exceeds the 32-bit buffer range (the entry points cut such a launch along the batch dimension); 0: one frame does not fit. */
int lwg_conv_slice_count(const LwgConvArgs* args);

/* BASELINE configs[3] ("MFMA bf16 conv tiles"): bf16 activations in HBM end to end, bf16 MFMA operands, fp32 accumulation.
 * Same launch description as lwg_conv2d_nhwc_f32 with xdt = ydt = LWG_DT_BF16: x0 / x1 / y / res / xn are bf16 NHWC tensors
 * (bias, mean, rstd stay fp32), Cin % 64 == 0 (C0 % 64 == 0 when C1 > 0), N % 64 == 0, YC % 8 == 0, ycoff % 8 == 0, and
 * args->w is the bf16 panel [ntaps*Cin/64][N][64] with k = ((c/64)*ntaps + tap)*64 + c%64, whose eight 16-byte k-octets of a
 * row n are stored at slot  octet ^ ((n >> 1) & 7)  (the LDS image the kernel's fragment reads expect: conflict-free
 * ds_read_b128 on 128-byte rows).  The first layer of a network (fp32 image-like input, Cin < 32) runs lwg_conv2d_nhwc_f32 with
 * ydt = LWG_DT_BF16 (LWG_EPI_NONE only): fp32 in, bf16 out. */
int lwg_conv2d_nhwc_bf16(const LwgConvArgs* args, lwg_stream_t stream);
/* The 3 x 3 / stride 1 / pad 1 fp32 convolution (attlwb_spade_resunet.py:18-22 ResidualBlock, :73-78,:87-92 SPADE, :337-340,:353-355 skip
 * convolutions) as a fused F(2x2, 3x3) Winograd convolution on the fp32 matrix pipe: 16 multiplies per 2 x 2 outputs instead of 36
 * (csrc/conv_winograd.hip).  Launch description as for lwg_conv2d_nhwc_f32 with: nine taps, stride = omul = 1, OH = YH = H, OW = YW = W,
 * xdt = ydt = LWG_DT_F32; one or two inputs (skip concatenation) with C0 % 8 == 0, C1 % 8 == 0 and (C0 + C1) % 16 == 0, every image of an input
 * < 3 GiB (any batch size: no slicing); N % 64 == 0, YC % 4 == 0; LWG_EPI_NONE or LWG_EPI_RESIDUAL (any channel slice ycoff % 4 == 0 of a wider
 * output) or LWG_EPI_SPADE (N = 2 YC, columns and bias gamma | beta interleaved in blocks of 32, ycoff = 0); activation none / ReLU / tanh /
 * sigmoid.  args->w = the transformed-weight fragment panel Upk[16][Cin/8][2][N][4]: element (p, s, kh, n, kk) = (G w G^T)[p / 4][p % 4] for
 * input channel 8 s + 2 kk + kh (concatenated order) and output column n, with G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]] and w the 3 x 3
 * kernel of that channel pair (ipercore_amd.ops builds it from the fp32 panel in fp64, rounded once).
 * fp32-grade results (relative error against fp64 1.6x the direct kernel's), not bitwise those of lwg_conv2d_nhwc_f32: a precision mode
 * of its own (ops.conv_precision("winograd")); a frame's result does not depend on the batch it is launched in. */
int lwg_conv2d_winograd_f32(const LwgConvArgs* args, lwg_stream_t stream);
/* The same launch with an optional workspace (the training step's one-sample launches): when the launch's 64-patch x 32-channel workgroups would
 * cover half the CUs or less (plain epilogue, or LWG_EPI_RESIDUAL with LWG_ACTIVATION_RELU_MASK) the K loop runs in lwg_conv2d_winograd_ws_floats(args)
 * / (M N) slices of >= 64 input channels whose dense (M, N) slabs the finishing kernel of lwg_conv2d_nhwc_f32_ws adds in slice order (deterministic).
 * lwg_conv2d_winograd_ws_floats returns 0 when the launch would not be split (or does not meet the contract above); ws = NULL runs it whole.
 * The synthesis path never passes one: a frame must not depend on its batch. */
size_t lwg_conv2d_winograd_ws_floats(const LwgConvArgs* args);
int lwg_conv2d_winograd_f32_ws(const LwgConvArgs* args, float* ws, lwg_stream_t stream);
/* How the library runs such a launch (so that callers that route launches by their size - ops._wino_plan: a one-sample training launch whose grid would leave
 * the chip idle stays on the direct split-K kernel - ask the kernel's own tile geometry instead of mirroring it): *blocks = 64-patch blocks of the launch
 * (times the K slices when with_ws != 0 and the split plan applies), *slices = K slices (0 = run whole), *nbv = output channels per block (64, or 32 for
 * small launches and slices), *workgroups = workgroups launched (persistent: min(blocks, compute units) for whole launches).  Any pointer may be NULL.
 * Returns 0, or 1 (hipErrorInvalidValue) when args does not meet lwg_conv2d_winograd_f32's contract. */
int lwg_conv2d_winograd_plan(const LwgConvArgs* args, int with_ws, long long* blocks, int* slices, int* nbv, int* workgroups);
/* The fragment panel Upk[16][Cin/8][2][N][4] of the call above from the fp32 GEMM panel of the same convolution (lwg_conv2d_nhwc_f32's w, nine taps,
 * Cin % 32 == 0): U = G w G^T per (input channel, output column) in fp64, rounded once.  tap9[3 r + s] = index of the tap (dy, dx) = (r - 1, s - 1)
 * in the GEMM panel's tap order.  With LWG_EPI_RESIDUAL the Winograd call also takes LWG_ACTIVATION_RELU_MASK (the data gradient behind a ReLU). */
int lwg_winograd_panel_f32(const float* wpanel, float* upk, int Cin, int N, const int* tap9, lwg_stream_t stream);
/* The same for every registered panel of a training step in ONE launch (the weights change every step): descs_dev = ndesc records in DEVICE memory,
 * each the argument list of one lwg_winograd_panel_f32 call plus first_block = the sum of ceil(N / 64) * ceil(Cin / 16) over the records before it;
 * total_blocks = that sum over all records.  Same values as ndesc single launches. */
typedef struct LwgWinoDesc {
    const float* wpanel;
    float* upk;
    int Cin, N, first_block;
    int tap9[9];
} LwgWinoDesc;
int lwg_winograd_panels_f32(const LwgWinoDesc* descs_dev, int ndesc, int total_blocks, lwg_stream_t stream);
/* The same layers as a fused F(4x4, 3x3) Winograd convolution (csrc/conv_winograd4.hip; round 6): 36 multiplies per 4 x 4 outputs = 2.25 per output
 * (F(2x2, 3x3): 4, direct: 9).  lwg_conv2d_winograd_f32's launch description and contract (one or two inputs, LWG_EPI_NONE / _RESIDUAL / _SPADE, any
 * activation incl. LWG_ACTIVATION_RELU_MASK with LWG_EPI_RESIDUAL, any batch size in one launch) EXCEPT args->w = the fragment panel
 * Upk[4][Cin/8][4][2][9 N] (144 Cin N bytes): block (q, s, kk, kh) of input channel c = 8 s + 2 kk + kh (concatenated order) holds, for output column n and
 * product j, (G w G^T)[xi][nu] at [n][j] (j = 0..3), 4 N + [n][j - 4] (j = 4..7), 8 N + [n] (j = 8), with (xi, nu) = (q, j) for j < 6 and
 * (4 + q / 2, 3 (q % 2) + j - 6) for j = 6..8; G = [[1/4,0,0],[-1/6,-1/6,-1/6],
 * [-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]] (points 0, +-1, +-2, inf).  fp32-grade results (relative L2 error against fp64 1-4e-6:
 * 0.4-4.3x the direct kernel's on the adversarial operands of tests/gpu_checks.py), neither the direct nor the F(2x2, 3x3) kernel's bits; a frame's result does not depend on the batch it is launched in
 * NOR on the form the entry point chooses for the launch (an 8-wave block of 64 output channels in persistent workgroups - block order by column block
 * per XCD, see the kernel - or, for launches that would leave half the chip without a workgroup, a 4-wave block of 32: bitwise the same values). */
int lwg_conv2d_winograd4_f32(const LwgConvArgs* args, lwg_stream_t stream);
/* That panel from the fp32 GEMM panel of the same convolution (arguments as lwg_winograd_panel_f32): U = G w G^T in fp64, rounded once. */
int lwg_winograd4_panel_f32(const float* wpanel, float* upk, int Cin, int N, const int* tap9, lwg_stream_t stream);
/* The same convolution for the 3x3 (9 taps) and 2x2 (4 taps, transposed-conv parity) stride-1 launches, as the
 * halo-tile kernel with register-streamed weights: args->w = the bf16 panel [ntaps*Cin/64][4][N][16] - element
 * [step][ks][n][e] = weight of GEMM column n at k = step*64 + ks*16 + e (k order as above) - and, for LWG_EPI_SPADE, columns
 * (and args->bias) interleaved gamma | beta in blocks of 16: column 32q + r is gamma of channel 16q + r (r < 16) or beta of
 * channel 16q + r - 16.  ntaps == 1 (the attention blocks' query projection, attlwb_spade_resunet.py:202-204: 1x1, stride 1,
 * C -> C with C in {64, 128, 256}, no second input, LWG_EPI_NONE): the weights stay in registers and persistent workgroups
 * stream the rows - same panel layout with one step per 64-channel chunk. */
int lwg_conv2d_nhwc_bf16_hr(const LwgConvArgs* args, lwg_stream_t stream);
/* First layer of a stream in bf16 mode (attlwb_spade_resunet.py:268-271 Encoder.0 on the 6-channel network input): x0 is the
 * fp32 NHWC-8 input (xdt = LWG_DT_F32, C0 = 8, C1 = 0), up to 10 taps, any stride, N = 64, bf16 output (ydt = LWG_DT_BF16),
 * bias + activation epilogue (LWG_EPI_NONE).  The input is rounded to bf16 in registers (one pixel's 8 channels = one MFMA
 * k-octet, no LDS); args->w = bf16 panel [ceil(ntaps/2)][64][16], element [ks][n][e] = weight of column n at k = 16 ks + e,
 * k = tap*8 + c, zero past ntaps*8. */
int lwg_conv2d_nhwc_c8_bf16(const LwgConvArgs* args, lwg_stream_t stream);
/* nn.ConvTranspose2d(kernel 4, stride 2, padding 1) (attlwb_spade_resunet.py:331-340 SkipDecoder upconvs) on bf16 NHWC in ONE
 * launch: the four output parities read the same input block, staged once.  args = the launch description of the parity-(0,0)
 * launch (ntaps = 4, stride = 1, omul = 2, OH = H, OW = W, YH = 2H, YW = 2W; dy / dx / ooy / oox ignored: parity (py, px) uses
 * dy in {py-1, py}, dx in {px-1, px} and writes pixels (2y + py, 2x + px)); args->w = the four register-streamed panels
 * [parity = 2 py + px][Cin/64 * 4][4][N][16], taps ascending in (dy, dx); C0 = 64 or 128, C1 = 0, N % 64 == 0, LWG_EPI_NONE. */
int lwg_conv_transpose4_nhwc_bf16(const LwgConvArgs* args, lwg_stream_t stream);
/* The same call on fp32 NHWC (nn.ConvTranspose2d(4, 2, 1) of the decoders, attlwb_spade_resunet.py:331-340, bg_inpaintor.py:49-50): args =
 * the parity-(0,0) launch description of lwg_conv2d_nhwc_f32 (ntaps = 4 with dy, dx in {-1, 0}, stride = 1, omul = 2, ooy = oox = 0, OH = H,
 * OW = W, YH = 2H, YW = 2W, LWG_EPI_NONE, one input, Cin % 32 == 0); args->w = the four parity panels stacked [2 py + px][4 Cin][N].  Small
 * launches (a frame or two) run as ONE grid of four times the workgroups, large ones as the four lwg_conv2d_nhwc_f32 launches; every
 * output element is computed exactly as by four separate calls. */
int lwg_conv_transpose4_nhwc_f32(const LwgConvArgs* args, lwg_stream_t stream);
/* 1 if the call above runs this description as one grid, 0 if as four launches (for callers that bracket launches with events). */
int lwg_conv_transpose4_is_one_grid(const LwgConvArgs* args);
/* The same layer as a fused F(2x2, 2x2) Winograd convolution on the fp32 matrix pipe (csrc/convt_winograd.hip): an output parity of
 * ConvTranspose2d(4, 2, 1) is a 2 x 2-tap convolution - 9 multiplies per 2 x 2 outputs of a parity instead of 16.  args = the parity-(0,0) launch
 * description as above (ntaps = 4, stride = 1, omul = 2, OH = H, OW = W, YH = 2H, YW = 2W, LWG_EPI_NONE, one input) with Cin % 16 == 0, N % 32 == 0,
 * the image < 3 GiB, ydt = LWG_DT_F32 or LWG_DT_F32_Q4, any activation of the forward path; args->w = the transformed-weight panel
 * Upk[4][Cin/8][4][2][9 N] (144 Cin N bytes; ABI 10) - per (parity 2 py + px, s, kk, kh) [N][4] products 0-3, [N][4] products 4-7, [N] product 8: every load of
 * the kernel reads contiguous memory -, product 3 xi + nu of column n = sgn (G g G^T)[xi][nu] for input channel 8 s + 2 kk + kh and output
 * column n, g[r][q] = w[c][n][3 - py - 2 r][3 - px - 2 q] (the parity's 2 x 2 sub-kernel), G = [[1,0],[1,1],[0,1]], sgn = (py == 1 && xi == 0 ? -1 : 1)
 * (px == 1 && nu == 0 ? -1 : 1).  fp32-grade results, not the bits of the call above: part of the "winograd" precision
 * mode of ipercore_amd.ops; a frame's result does not depend on the batch it is launched in. */
int lwg_conv_transpose4_winograd_f32(const LwgConvArgs* args, lwg_stream_t stream);

/* fp32 convolution on the bf16 matrix pipe ("bf16x6"): both operands are split exactly into three bf16 parts
 * (activations in the kernel, weights on the host: args->w = [3][ntaps*Cin/8][N][8] bf16 planes hi / mid / lo), six bf16 MFMAs
 * per fp32 product, fp32 accumulation; dropped terms < 2^-23 |a b|.  Same contract and restrictions as the bf16 entry point. */
int lwg_conv2d_nhwc_f32_split(const LwgConvArgs* args, lwg_stream_t stream);

/* Backward of the same convolutions (personalization step, tools/trainers/lwg_trainer.py:326-352: loss.backward()
 * through torch.nn.Conv2d / ConvTranspose2d).
 * lwg_conv2d_wgrad_nhwc_f32: dW[K,N] = A[M,K]^T dY[M,N] on the matrix cores.  `args` is the FORWARD launch description
 *   (x0/x1, taps, stride, OH/OW/M, N and the output mapping; args->y/w/bias/epi are ignored); dy has the layout of the
 *   forward output; dw is (ntaps*Cin, N) row-major with K in the forward panel's order; ws: lwg_conv2d_wgrad_ws_floats().
 * The data gradient is lwg_conv2d_nhwc_f32 itself on dy with a transposed panel (ipercore_amd/networks/packing.py
 *   pack_dgrad_*); lwg_colsum_nhwc_f32 gives bias gradients (out[c] = sum over rows of x (rows, C); ws: 512*C floats). */
size_t lwg_conv2d_wgrad_ws_floats(int Ktot, int N, int M);
int lwg_conv2d_wgrad_nhwc_f32(const LwgConvArgs* args, const float* dy, float* dw, float* ws, lwg_stream_t stream);
/* The same weight gradient written straight into the parameter's gradient tensor dw (D0,D1,KH,KW) - the slab reduction and
 * lwg_unpack_wgrad_f32 (below; same transposed / kidx / cin / nout convention, cin_pad = C0 + C1, n_pad = N) in one launch.
 * db: NULL, or nout floats receiving the bias gradient db[n] = sum over the launch's M rows of dy[m, n] (torch's
 *   grad_bias of nn.Conv2d): the column sums are accumulated by the workgroups that stage dy anyway and reduced by the same
 *   launch - no separate pass over dy (lwg_colsum_nhwc_f32 remains for tensors that have no weight gradient next to them). */
int lwg_conv2d_wgrad_unpacked_f32(const LwgConvArgs* args, const float* dy, float* ws, float* dw, int D0, int D1, int KH, int KW,
                                  int transposed, const int* kidx, int cin, int nout, float* db, lwg_stream_t stream);
int lwg_colsum_nhwc_f32(const float* x, size_t rows, int C, float* out, float* ws, lwg_stream_t stream);

/* Elementwise / normalisation pieces of the personalization step and their backward (csrc/train_ops.hip); NHWC fp32.
 *   act codes: LWG_ACTIVATION_* plus 4 = LeakyReLU(0.2) (discriminators/patch_dis.py:33-47).
 * lwg_act_bwd_f32:        out = dy * act'(y)  (ReLU after the convs, tanh / sigmoid of the regressors), n % 4 == 0.
 * lwg_norm_fwd_nhwc_f32:  y = act((x - mean) rstd (1 + gamma) + beta); gamma = beta = NULL: InstanceNorm + activation
 *                         (bg_inpaintor.py:31-57); with gamma / beta (B,HW,C): SPADE (attlwb_spade_resunet.py:92).
 * lwg_norm_bwd_nhwc_f32:  the backward of the above: dx (and dgamma, dbeta); ws: B*(nsplit+1)*C*2 floats (split records + their fold).
 *   gstride (both): floats per pixel row of gamma / beta (and dgamma / dbeta): 0 or C for dense (B,HW,C) tensors; 2C with
 *   beta = gamma + C (dbeta = dgamma + C) when the two are the halves of ONE (B,HW,2C) tensor - the training step runs SPADE's
 *   mlp_gamma | mlp_beta convolutions (attlwb_spade_resunet.py:66-67, 88-89) as a single launch of 2C columns.
 * lwg_adam_step_f32:      torch.optim.Adam update (lwg_trainer.py:140-146) of a flat parameter buffer, step count t >= 1. */
int lwg_act_bwd_f32(const float* dy, const float* y, size_t n, int act, float* out, lwg_stream_t stream);
int lwg_norm_fwd_nhwc_f32(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta, int gstride,
                          int B, int HW, int C, int act, float* y, lwg_stream_t stream);
int lwg_norm_bwd_nhwc_f32(const float* dy, const float* y, const float* x, const float* mean, const float* rstd,
                          const float* gamma, int gstride, int B, int HW, int C, int act, int nsplit, float* dx, float* dgamma,
                          float* dbeta, float* ws, lwg_stream_t stream);
int lwg_adam_step_f32(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps, int t,
                      lwg_stream_t stream);
/* The same update with the step count kept on the device (*t_dev is incremented, then read): what a captured (hipGraph) training step
 * must use, because a host-side step count would be frozen into the graph. */
int lwg_adam_step_dev_f32(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                          int* t_dev, lwg_stream_t stream);
/* Weight panels straight from the parameter tensors, one launch each (the personalization step re-packs every weight and
 * un-packs every weight gradient every step).  w (D0, D1, KH, KW) contiguous; kidx[tap] = ky*KW + kx of the weight slice a GEMM
 * tap reads; transposed = 0: value = w[n][c][kidx] (Conv2d forward, ConvTranspose2d data gradient), 1: w[c][n][kidx]
 * (ConvTranspose2d forward, Conv2d data gradient); (cin, nout) zero-extended to (cin_pad, n_pad); out = the
 * [ceil32(ntaps*cin_pad)/4][n_pad][4] panel lwg_conv2d_nhwc_f32 reads.  lwg_unpack_wgrad_f32 is the inverse for the
 * (ntaps*cin_pad, n_pad) output of lwg_conv2d_wgrad_nhwc_f32; weight positions no tap maps to are left untouched. */
int lwg_pack_panel_f32(const float* w, int D0, int D1, int KH, int KW, int transposed, const int* kidx, int ntaps, int cin,
                       int cin_pad, int nout, int n_pad, float* out, lwg_stream_t stream);
int lwg_unpack_wgrad_f32(const float* dwk, int D0, int D1, int KH, int KW, int transposed, const int* kidx, int ntaps, int cin,
                         int cin_pad, int nout, int n_pad, float* dw, lwg_stream_t stream);
/* Every panel of a training step in one launch: descs_dev = ndesc LwgPackDesc records in DEVICE memory (built once per network -
 * the parameters live in flat buffers, their addresses are stable), each the argument list of one lwg_pack_panel_f32 call plus
 * first_block = the sum of ceil((Kp/4)*n_pad / 256) over the records before it (Kp = ceil32(ntaps*cin_pad)); total_blocks = that
 * sum over all records.  Same values as ndesc single launches. */
typedef struct LwgPackDesc {
    const float* w;    /* (D0, D1, KH, KW) contiguous */
    float* out;        /* [Kp/4][n_pad][4] */
    int D1, KHW, transposed, ntaps, cin, cin_pad, nout, n_pad, Kp, first_block;
    int kidx[LWG_MAX_TAPS];
} LwgPackDesc;
int lwg_pack_panels_f32(const LwgPackDesc* descs_dev, int ndesc, int total_blocks, lwg_stream_t stream);
/* MaxPool2d(2, 2) on NHWC (VGG19 perceptual loss, criterions/vggloss.py): y (B,H/2,W/2,C); the backward writes all of dx
 * (B,H,W,C), each gradient going to the first maximum of its window in scan order.  H, W even; C % 4 == 0. */
int lwg_maxpool2_fwd_nhwc_f32(const float* x, float* y, int B, int H, int W, int C, lwg_stream_t stream);
int lwg_maxpool2_bwd_nhwc_f32(const float* x, const float* dy, float* dx, int B, int H, int W, int C, lwg_stream_t stream);
/* FaceLoss head crops (criterions/faceloss.py:316-341,384-406: imgs[i, :, y0:y1, x0:x1] resized to (OH, OW) = (112, 96) with F.interpolate(bilinear,
 * align_corners = True)) with the boxes read ON THE DEVICE: x (N,C,H,W) NCHW fp32; box (N,4) int64 = (min_x, max_x, min_y, max_y) as
 * tools/trainers/base.py:205-246 computes them; y (N,C,OH,OW).  The reference reads the boxes on the host and drops samples whose box is empty; here
 * every sample gets a crop - zeros, and valid[i] = 0 (valid: (N) fp32 or NULL), for an empty or out-of-image box - so the personalization step keeps
 * static shapes and is captured as a hipGraph.  _bwd: dx (N,C,H,W), ZERO-FILLED by the caller, += the transposed interpolation of dy (atomic adds). */
int lwg_crop_resize_bilinear_f32(const float* x, const long long* box, float* y, float* valid, int N, int C, int H, int W, int OH, int OW,
                                 lwg_stream_t stream);
int lwg_crop_resize_bilinear_bwd_f32(const float* dy, const long long* box, float* dx, int N, int C, int H, int W, int OH, int OW, lwg_stream_t stream);
/* nn.PReLU(C) of the frozen Sphere20a (criterions/faceloss.py:209-283, relu{b}_{i}) on NHWC rows, with the block's residual add (:262-281, x + relu(conv(..)))
 * folded in: y = (res ? res : 0) + (x >= 0 ? x : slope[c] x); x, res, y (rows, C), slope (C), C % 4 == 0.  _bwd: dx = dy * (x >= 0 ? 1 : slope[c]) - the slopes
 * are frozen (no gradient for them) and the residual's gradient is dy itself. */
int lwg_prelu_f32(const float* x, const float* slope, const float* res, size_t rows, int C, float* y, lwg_stream_t stream);
int lwg_prelu_bwd_f32(const float* x, const float* slope, const float* dy, size_t rows, int C, float* dx, lwg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * InstanceNorm2d(affine=False) statistics (biased variance), NHWC.
 * Replaces nn.InstanceNorm2d at attlwb_spade_resunet.py:62,:83 and bg_inpaintor.py:14,17,33,40,51.
 *   x (B,HW,C) -> mean (B,C), rstd (B,C) = 1/sqrt(var + eps).   ws: >= B*C*nsplit*3 floats of scratch.
 * lwg_instnorm_apply_nhwc_f32: y = act((x - mean) * rstd) (+ res), used by the background network.
 * ------------------------------------------------------------------------------------------------ */
int lwg_instnorm_stats_nhwc_f32(const float* x, int B, int HW, int C, float eps, float* mean, float* rstd,
                                float* ws, int nsplit, lwg_stream_t stream);
/* The statistics of a bf16 (B,HW,C) tensor (fp32 arithmetic and outputs), C in {64,128,256}, nsplit <= 64. */
int lwg_instnorm_stats_nhwc_bf16(const void* x, int B, int HW, int C, float eps, float* mean, float* rstd, float* ws, int nsplit,
                                 lwg_stream_t stream);
int lwg_instnorm_apply_nhwc_f32(const float* x, const float* mean, const float* rstd, const float* res,
                                float* y, int B, int HW, int C, int act, lwg_stream_t stream);

/* LWB.resize_trans (attlwb_spade_resunet.py:175-181): n flow fields (S,S,2) -> (h,w,2), bilinear, align_corners = True, as its own pass.
 * The block kernels below resize per pixel when handed full-resolution flows; handed a field already at their resolution (S == h == w)
 * they read it directly - the engine resizes once per frame batch and resolution. */
int lwg_flow_resize_f32(const float* T, int n, int S, int h, int w, float* out, lwg_stream_t stream);
/* ------------------------------------------------------------------------------------------------
 * Liquid Warping Block, attention form (one of 9 sites per frame).
 * Replaces LWB.resize_trans + LWB.transform (attlwb_spade_resunet.py:175-191), the fk/fv 1x1 convs on the
 * warped features (:226-227) and SelfAttentionBlock (:106-139).  Ks/Vs are Wk*x_src / Wv*x_src (no bias),
 * computed once per source with lwg_conv2d_nhwc_f32; the biases are added after the warp, as the reference
 * does.  q (B,h,w,C), Ks/Vs (ns,h,w,C) shared by all frames (src_batched = 0) or (B*ns,h,w,C) (= 1),
 * T (B,ns,S,S,2), out (B,h,w,C); C in {32,64,128,256}.
 * ------------------------------------------------------------------------------------------------ */
int lwg_lwb_attention_f32(const float* q, const float* Ks, const float* Vs, const float* bk, const float* bv,
                          const float* T, float* out, int B, int ns, int h, int w, int C, int S,
                          int src_batched, lwg_stream_t stream);
/* The same block on bf16 q / Ks / Vs / out (BASELINE configs[3]: bf16 activation storage); bk / bv / T stay fp32; C in {64,128,256}. */
int lwg_lwb_attention_bf16(const void* q, const void* Ks, const void* Vs, const float* bk, const float* bv, const float* T, void* out,
                           int B, int ns, int h, int w, int C, int S, int src_batched, lwg_stream_t stream);
/* The same block with the QUERY PROJECTION FOLDED INTO THE SOURCE SIDE (the form the per-frame engine runs; csrc/lwb_attn_x.hip).
 * With q = Wq x + bq (fq, attlwb_spade_resunet.py:121-131) and the linear zero-padded warp,
 *   K_s . q = warp_s(Wq^T Wk f_s) . x + warp_s(bq . Wk f_s) + bk . q,  the last term the same for every source: it cancels in softmax_s.
 * Kq (nsrc,h,w,C) = (Wq^T Wk) f_src, kappa (nsrc,h,w) = (Wk^T bq) . f_src and Vs (nsrc,h,w,C) = Wv f_src are computed ONCE per source;
 * per frame:  logit_s = (warp_s(Kq) . x + warp_s(kappa)) / sqrt(C),  out = sum_s softmax_s(logit) warp_s(Vs) + bv.
 * Replaces, per site and frame, the fq convolution + LWB.transform + SelfAttentionBlock (:106-139, :175-191, :226-227).
 * x, out (B,h,w,C); T (B,ns,h,w,2): flows ALREADY RESIZED to (h,w) (lwg_flow_resize_f32); K / V tensors < 3 GiB each.
 * stats: NULL, or B * nrec * C * 3 floats with nrec = lwg_lwb_attention_x_records(h, w, C, element size): the kernel reads every element
 * of x once and leaves per workgroup (an 8 x 8 tile, or a part of one on small feature maps - a function of (h, w, C) only, so that a
 * frame never depends on its batch) the InstanceNorm partial record (count, mean, M2) of x (SPADE's parameter-free norm, :62,:83);
 * finish with lwg_instnorm_finalize_f32. */
int lwg_lwb_attention_x_records(int h, int w, int C, int element_size);
int lwg_lwb_attention_x_f32(const float* x, const float* Kq, const float* kappa, const float* Vs, const float* bv, const float* T,
                            float* out, float* stats, int B, int ns, int h, int w, int C, int src_batched, lwg_stream_t stream);
int lwg_lwb_attention_x_bf16(const void* x, const void* Kq, const float* kappa, const void* Vs, const float* bv, const float* T,
                             void* out, float* stats, int B, int ns, int h, int w, int C, int src_batched, lwg_stream_t stream);
/* ws (B,nrec,C,3) records (count, mean, M2) -> mean, rstd (B,C); rstd = 1 / sqrt(M2 / n + eps) (nn.InstanceNorm2d: biased variance).
 * ws holds lwg_instnorm_finalize_ws_floats(B, C, nrec) floats: the records, then scratch for the segment partials of long record lists
 * (nrec > 512); record 0 of every image must be non-empty (its mean is the reference the moments are summed about). */
size_t lwg_instnorm_finalize_ws_floats(int B, int C, int nrec);
int lwg_instnorm_finalize_f32(float* ws, int B, int C, int nrec, float eps, float* mean, float* rstd, lwg_stream_t stream);
/* Backward of the above for the personalization step (lwg_trainer.py:649-697 runs the same block under autograd; the
 * flows are constants there).  dq is written; dKs / dVs are accumulated with fp32 atomics and must be zero on entry;
 * ns <= 8.  The bias gradients need no kernel: dbv = column sum of dout, dbk = 0. */
int lwg_lwb_attention_bwd_f32(const float* q, const float* Ks, const float* Vs, const float* bk, const float* bv,
                              const float* T, const float* dout, float* dq, float* dKs, float* dVs, int B, int ns,
                              int h, int w, int C, int S, int src_batched, lwg_stream_t stream);
/* The fp32 block and its backward with K | V as ONE tensor kv (nsrc,h,w,2C), K = kv[..., :C], V = kv[..., C:]: the training step
 * projects the source features with a single stacked fk | fv 1x1 convolution (attlwb_spade_resunet.py:226-227 are two) and these
 * entry points read / accumulate the halves in place.  dkv (same shape) must be zero on entry. */
int lwg_lwb_attention_kv_f32(const float* q, const float* kv, const float* bk, const float* bv, const float* T, float* out,
                             int B, int ns, int h, int w, int C, int S, int src_batched, lwg_stream_t stream);
int lwg_lwb_attention_kv_bwd_f32(const float* q, const float* kv, const float* bk, const float* bv, const float* T,
                                 const float* dout, float* dq, float* dkv, int B, int ns, int h, int w, int C, int S,
                                 int src_batched, lwg_stream_t stream);
/* The non-attention Liquid Warping Blocks (AddLWB / AvgLWB: generators/lwb_resunet.py:77-152; SoftGateLWB:
 * generators/lwb_softgate_resunet.py:77-123) as one gather kernel:
 *   out = (tsf_x + gate * scale_w * sum_s warp_s(src_x)) * scale_o        gate == NULL: 1
 * Add: (1, 1); Avg: scale_o = 1/(ns+1); SoftGateAdd: gate, (1, 1); SoftGateAvg: gate, scale_w = 1/ns.  Layouts as above. */
int lwg_lwb_fuse_f32(const float* tsf_x, const float* src_x, const float* gate, const float* T, float* out, int B, int ns,
                     int h, int w, int C, int S, int src_batched, float scale_w, float scale_o, lwg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Renderer (replaces the `neural_renderer` CUDA package as used by renders/nmr.py).
 * lwg_project_faces_f32:   nmr.py:34-52 orthographic_proj_withz_idrot + :331 y flip + :333 nr.look_at
 *                          (eye (0,0,-eye_dist): identity rotation) + :336 nr.vertices_to_faces, and the
 *                          f2pts of :339-340.  verts (B,nv,3), cam (B,3), faces (nf,3) int32 ->
 *                          faces_v (B,nf,3,3) and/or f2pts (B,nf,3,2).
 * lwg_rasterize_fim_wim_f32: nmr.py:337,356 nr.rasterize_face_index_map_and_weight_map(faces, S, False):
 *                          faces_v (B,nf,3,3) -> fim (B,S,S) int32 (-1 = background), wim (B,S,S,3).
 *                          ws: lwg_rasterize_ws_bytes(B,nf,S) bytes of scratch; S <= 2048.
 * ------------------------------------------------------------------------------------------------ */
size_t lwg_rasterize_ws_bytes(int B, int nf, int S);
int lwg_project_faces_f32(const float* verts, const float* cam, const int32_t* faces, int B, int nv, int nf,
                          float eye_dist, float* faces_v, float* f2pts, lwg_stream_t stream);
int lwg_rasterize_fim_wim_f32(const float* faces_v, int B, int nf, int S, float near, float far, int32_t* fim,
                              float* wim, void* ws, lwg_stream_t stream);
/* Textured rendering from the maps above (SMPLRenderer.render -> nr.rasterize, renders/nmr.py:271-290): perspective-correct
 * trilinear sampling of per-face T x T x T textures.  PARITY UNPINNED: neural_renderer is not vendored with the reference
 * and the reference holds no output of this function; the published forward_texture_sampling algorithm is restated.
 * textures (B | 1, nf, T, T, T, 3); bg_color: 3 host floats; rgb (B, S, S, 3) on the pixel grid of (fim, wim). */
int lwg_texture_sample_f32(const int32_t* fim, const float* wim, const float* faces_v, const float* textures, int B, int nf,
                           int S, int T, int tex_batched, float eps, const float* bg_color3_host, float* rgb, lwg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Flow composition from (fim, wim).
 * lwg_flow_compose_f32 fuses, for B target frames: encode_fim (nmr.py:390-401), cal_bc_transform with the UV
 *   table (nmr.py:713-757 as called at flowcomposition.py:240), the UV grid_sample (:242), the tsf_inputs
 *   concat (:244) and make_trans_flow (:551-567).  map_fn (nf+1,3); f_uvs2img (nf,3,2); uv_img4 (Hu,Wu,4)
 *   NHWC (4th channel ignored); src_f2pts (ns,nf,3,2).  Outputs: tsf_inputs (B,S,S,8) NHWC [syn3|cond3|0|0],
 *   Tst (B,ns,S,S,2); optional cond_nchw (B,3,S,S) and Tuv (B,S,S,2) (NULL to skip).
 * lwg_bc_transform_f32: cal_bc_transform for arbitrary per-batch tables f2pts (B,nf,3,2) -> T (B,S,S,2).
 * lwg_encode_fim_f32:   map_fn[fim] for any (nf+1,D) table -> (B,D,S,S).
 * ------------------------------------------------------------------------------------------------ */
int lwg_flow_compose_f32(const int32_t* fim, const float* wim, int B, int S, const float* map_fn, int nf,
                         const float* f_uvs2img, const float* uv_img4, int Hu, int Wu, const float* src_f2pts,
                         int ns, float* tsf_inputs, float* Tst, float* cond_nchw, float* Tuv, lwg_stream_t stream);
int lwg_bc_transform_f32(const float* f2pts, const int32_t* fim, const float* wim, int B, int S, int nf, float* T,
                         lwg_stream_t stream);
int lwg_encode_fim_f32(const int32_t* fim, const float* map_fn, int B, int S, int nf, int D, float* out,
                       lwg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * SMPL / SMPL-H linear blend skinning for B frames.
 * Replaces smplx/lbs.py:137-227 (lbs) incl. rotations.py:318-375, lbs.py:321-375 and base_smpl.py:28-50
 * (link, 2-D ids) and :7-18 (j2d).  pose rows are 3*nj axis-angle floats (row stride pose_stride);
 * shapedirs (nv,3,nbeta); posedirs ((nj-1)*9, nv*3); J_regressor (nj,nv); parents (nj) int32;
 * lbs_weights (nv,nj); offsets NULL | (nv,3) | (B,nv,3); links NULL | (nlinks,2) int32 (to, from).
 * ws: lwg_smpl_lbs_ws_floats(B,nv,nj) floats.
 * ------------------------------------------------------------------------------------------------ */
size_t lwg_smpl_lbs_ws_floats(int B, int nv, int nj);
int lwg_smpl_lbs_f32(const float* pose, int pose_stride, const float* beta, int beta_stride, int nbeta,
                     const float* cam, int cam_stride, const float* v_template, const float* offsets, int off_batched,
                     const float* shapedirs, const float* posedirs, const float* J_regressor, const int32_t* parents,
                     const float* lbs_weights, const int32_t* links, int nlinks, int B, int nv, int nj, float* verts,
                     float* j3d, float* j2d, float* ws, lwg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Output head + compositing: tsf_img_reg / tsf_att_reg (attlwb_spade_resunet.py:605-613, called :533) and
 * Imitator.forward's pred = mask*bg + (1-mask)*img (models/imitator.py:393).
 * x (B,S,S,C) NHWC; wpk [25][C][4] (outputs 0..2 image, 3 mask); bg (.,3,S,S) NCHW, batch stride bg_bstride
 * floats (0 = one shared background).  pred (B,3,S,S), mask (B,1,S,S), img (B,3,S,S), each optional.
 * lwg_nchw_to_nhwc_f32 / lwg_nhwc_to_nchw_f32: layout changes at the API edge ((B,C,P) <-> (B,P,Cp)).
 * ------------------------------------------------------------------------------------------------ */
int lwg_head_compose_f32(const float* x, const float* wpk, const float* bg, size_t bg_bstride, int B, int S, int C,
                         float* pred, float* mask, float* img, lwg_stream_t stream);
/* The same head on an input stored as channel-quad planes, x (B, C/4, S, S, 4) = what a convolution with ydt = LWG_DT_F32_Q4 writes: a
 * stage of the halo tile then reads whole 128-byte lines whatever its channel count (NHWC: 32 B of every pixel's 256-B row per stage,
 * every line re-fetched four times), which lets a thread keep 8 pixels in registers (64 x 32-pixel tiles).  Same outputs; the sums
 * are formed in a different order than lwg_head_compose_f32's (fp32 rounding differs), identically for every batch size. */
int lwg_head_compose_q4_f32(const float* x, const float* wpk, const float* bg, size_t bg_bstride, int B, int S, int C,
                            float* pred, float* mask, float* img, lwg_stream_t stream);
/* Thin regressor forward: a stride-1 ks x ks convolution (ks = 5 or 7, pad ks/2, no bias, no activation) with <= 4 output channels at
 * full resolution on the vector ALUs - the 7x7 image head of the background network, bg_inpaintor.py:53 (Conv2d(64, 3, 7, 1, 3,
 * bias=False), the Tanh that follows is the caller's).  x (B,S,S,C) NHWC, C % 8 == 0; wpk [ks*ks][C][4] (tap = ky*ks + kx, unused
 * output columns zero) -> y (B,S,S,4) NHWC, pre-activation. */
int lwg_thin_conv_f32(const float* x, const float* wpk, int B, int S, int C, int ks, float* y, lwg_stream_t stream);
/* The same head on a bf16 NHWC input (B,S,S,64) with the regressors on the matrix cores (v_mfma_f32_16x16x32_bf16): wb is the bf16
 * operand panel [ky 5][pass 2][channel half 2][lane 64][8]: lane l of a block holds row (l % 16) = 4 * tap + output and the
 * k-octet (l / 16) of the 32 channels; pass 0 carries the taps kx = 0..3, pass 1 the tap kx = 4 in its rows 0..3 (rows 4..15 zero).
 * Outputs fp32 NCHW as lwg_head_compose_f32. */
int lwg_head_compose_bf16(const void* x, const void* wb, const float* bg, size_t bg_bstride, int B, int S, int C, float* pred,
                          float* mask, float* img, lwg_stream_t stream);
/* BASELINE configs[3], the last stage of forward_tsf as ONE launch (csrc/up4_head_bf16.hip): nn.ConvTranspose2d(128 -> 64, 4, 2, 1) + ReLU
 * (attlwb_spade_resunet.py:331-340), the two 5x5 regressors with tanh / sigmoid (:605-613) and the compositing (models/imitator.py:393) - the
 * (B, 2H, 2W, 64) tensor between them is never written.  args: the launch description lwg_conv_transpose4_nhwc_bf16 takes for that layer (C0 = 128, N = 64,
 * LWG_EPI_NONE + LWG_ACT_RELU, args->w = the four parity panels, args->bias), its y / YH / YW / YC ignored; whead: lwg_head_compose_bf16's panel; bg / pred /
 * mask / img as there, at (2H, 2W).  Same values as the two calls it replaces (the intermediate is rounded to bf16 as they round it). */
int lwg_up4_head_compose_bf16(const LwgConvArgs* args, const void* whead, const float* bg, size_t bg_bstride, float* pred, float* mask, float* img,
                              lwg_stream_t stream);
/* lwg_frames_to_u8: the output conversion of Imitator.inference (models/imitator.py:368-372 ->
 * cv_utils.save_cv2_img(normalize=True), tools/utils/filesio/cv_utils.py:100-116): pred (B,3,S,S) fp32 ->
 * (B,S,S,3) uint8 = uint8((x+1)/2.0*255) in numpy fp32 arithmetic (truncation); bgr = 1: cv2's channel order. */
int lwg_frames_to_u8(const float* pred, int B, int S, int bgr, uint8_t* out, lwg_stream_t stream);
int lwg_nchw_to_nhwc_f32(const float* src, float* dst, int B, int C, int Cp, int P, lwg_stream_t stream);
int lwg_nhwc_to_nchw_f32(const float* src, float* dst, int B, int C, int Cs, int P, lwg_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Once-per-source image stage (Imitator.source_setup, models/imitator.py:177-246 -> FlowComposition.process_source,
 * models/flowcomposition.py:452-512).  All tensors NCHW fp32 as in the reference API at this stage.
 * lwg_morph_f32:          tools/utils/morphology/morph_ops.py:7-37 (mode 0 erode: pad 1, sum == ks^2; 1 dilate: pad 0,
 *                         sum >= 1) and :40-63 (mode 2 soft_dilate: sum >= ks^2/2).  in/out (n,1,H,W); ws n*H*W floats.
 * lwg_canny_f32:          tools/utils/morphology/canny_ops.py:137-212 CannyFilter.forward(img, low, high, True) for a
 *                         1-channel image; gauss9 / sobelx9 are the reference's 3x3 kernels (row-major, HOST pointers;
 *                         sobel_y is the transpose); edges (n,1,H,W) in {0,1}; ws 3*n*H*W floats.
 * lwg_boundary_fill_f32:  flowcomposition.py:268-386 (cal_top_k_ids + morph_image + make_morph_image body): pixels with
 *                         outpad*(1-confidant) != 0 take sum_k w_k * src[nn_k] over their 3 nearest edge pixels,
 *                         w_k = d_k^2 / sum d^2; the rest src * confidant.  Distance ties -> lowest row-major edge
 *                         index.  top3 (n,3,H,W) int32 optional (squared distances, -1 elsewhere).
 *                         ws: n*(H*W + 1) int32; the edge counts land at ws[n*H*W + i] (>= 3 required per image
 *                         that has uncertain pixels - the reference's topk raises otherwise).
 * lwg_grid_sample_nchw_f32: F.grid_sample(img, grid) bilinear / zeros / align_corners=False as called at
 *                         flowcomposition.py:117-118 (img_bstride = 0 broadcasts one image).
 * lwg_uv_merge_f32:       flowcomposition.py:123-130: src_warp (ns,3,H,W), dilated vis (ns,1,H,W) -> merge_uv (3,H,W).
 * lwg_pack_inputs_f32:    cat[a * mask, b] NCHW planes -> NHWC Cp channels (zero padded): make_bg_inputs
 *                         (flowcomposition.py:250-260, a = src image, b = mask = eroded mask) and make_src_inputs
 *                         (:262-265, a = morphed image, b = cond, mask = NULL).
 * ------------------------------------------------------------------------------------------------ */
int lwg_morph_f32(const float* in, float* out, int n, int H, int W, int ks, int mode, float* ws, lwg_stream_t stream);
int lwg_canny_f32(const float* sil, int n, int H, int W, const float* gauss9, const float* sobelx9, float low, float high,
                  float* edges, float* ws, lwg_stream_t stream);
int lwg_boundary_fill_f32(const float* src, const float* confidant, const float* outpad, const float* edges, int n, int H,
                          int W, float* out, int32_t* top3, int32_t* ws, lwg_stream_t stream);
int lwg_grid_sample_nchw_f32(const float* img, size_t img_bstride, const float* grid, int n, int C, int H, int W, int Ho,
                             int Wo, float* out, lwg_stream_t stream);
int lwg_uv_merge_f32(const float* src_warp, const float* vis, int ns, int H, int W, float* out, lwg_stream_t stream);
/* Swapper (FlowCompositionForSwapper.merge_uv_img, flowcomposition.py:816-856): n people's UV images (n,3,H,W) merged with
 * their selected-part visibility maps (n,1,H,W):  out = sum_i uv_i * vis_i / (sum_j vis_j + 1e-7)  -> (3,H,W). */
int lwg_uv_merge_parts_f32(const float* uv_imgs, const float* vis, int n, int H, int W, float* out, lwg_stream_t stream);
int lwg_pack_inputs_f32(const float* a, int Ca, const float* b, int Cb, const float* mask, int n, int H, int W, int Cp,
                        float* out, lwg_stream_t stream);

int lwg_device_cu_count(void);

#ifdef __cplusplus
}
#endif
#endif /* LWG_HIP_H */
