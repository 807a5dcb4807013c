// Once-per-source image stage of Imitator.source_setup (reference iPERCore/models/imitator.py:177-246 ->
// models/flowcomposition.py:452-512 process_source):
//   morph            tools/utils/morphology/morph_ops.py:7-37      box sum + threshold (erode / dilate / soft dilate)
//   CannyFilter      tools/utils/morphology/canny_ops.py:71-212    blur, Sobel, orientation, NMS, thresholds, hysteresis
//   make_morph_image models/flowcomposition.py:268-386             3 nearest boundary pixels fill of the uncertain band
//   grid_sample      F.grid_sample(img, T) bilinear / zeros / align_corners=False (flowcomposition.py:117-118)
//   make_uv_img      models/flowcomposition.py:87-137              multi-source UV merge
// All HBM/latency-bound image kernels on (n,C,H,W) fp32 NCHW tensors (the layout of the reference API at this
// stage); they run once per source set, so they are written for exactness and zero host synchronisation (the
// reference builds an (n1, n2) distance matrix and calls nonzero(): quadratic memory + host syncs), not for peak
// bandwidth.
#include "lwg_common.h"
#include "lwg_conv_args.h"

// ---------------------------------------------------------------------------------------------- morph
// Two separable passes: row sums (with the pad value outside), then column sums + threshold.  Sums of 0/1 masks are
// exact integers in fp32; for fractional inputs the order is left-to-right, top-to-bottom.
__global__ void lwg_morph_rows_kernel(const float* __restrict__ in, float* __restrict__ ws, int n, int H, int W, int ks,
                                      float padv) {
    const size_t total = (size_t)n * H * W;
    const int r = ks / 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const float* row = in + (i - x);
        float s = 0.f;
        for (int d = -r; d < ks - r; ++d) {
            const int xx = x + d;
            s += (xx >= 0 && xx < W) ? row[xx] : padv;
        }
        ws[i] = s;
    }
}

__global__ void lwg_morph_cols_kernel(const float* __restrict__ ws, float* __restrict__ out, int n, int H, int W, int ks,
                                      float padv, int mode) {
    const size_t total = (size_t)n * H * W;
    const int r = ks / 2;
    const float nks = (float)(ks * ks);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const int y = (int)((i / W) % H);
        const float* img = ws + (i - (size_t)y * W - x);
        float s = 0.f;
        for (int d = -r; d < ks - r; ++d) {
            const int yy = y + d;
            s += (yy >= 0 && yy < H) ? img[(size_t)yy * W + x] : padv * (float)ks;
        }
        float o;
        if (mode == 0) o = (s == nks) ? 1.f : 0.f;           // erode: every tap is 1
        else if (mode == 1) o = (s >= 1.f) ? 1.f : 0.f;      // dilate
        else o = (s >= nks / 2.f) ? 1.f : 0.f;               // soft dilate
        out[i] = o;
    }
}

extern "C" int lwg_morph_f32(const float* in, float* out, int n, int H, int W, int ks, int mode, float* ws,
                             lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!in || !out || !ws || n <= 0 || H <= 0 || W <= 0 || ks < 1 || (ks & 1) == 0 || mode < 0 || mode > 2)
        return (int)hipErrorInvalidValue;
    const float padv = mode == 0 ? 1.f : 0.f;
    const size_t total = (size_t)n * H * W;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(lwg_morph_rows_kernel, dim3(blocks), dim3(256), 0, stream, in, ws, n, H, W, ks, padv);
    hipLaunchKernelGGL(lwg_morph_cols_kernel, dim3(blocks), dim3(256), 0, stream, ws, out, n, H, W, ks, padv, mode);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------- Canny
// 3x3 cross-correlation with zero padding, taps accumulated in row-major order (what a direct conv does).
__device__ __forceinline__ float lwg_tap(const float* img, int H, int W, int y, int x) {
    return (y >= 0 && y < H && x >= 0 && x < W) ? img[(size_t)y * W + x] : 0.f;
}

struct LwgK3 { float w[9]; };

__global__ void lwg_conv3_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int H, int W, LwgK3 k) {
    const size_t total = (size_t)n * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        const float* img = in + (i - (size_t)y * W - x);
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) s = __fadd_rn(s, __fmul_rn(k.w[t], lwg_tap(img, H, W, y + t / 3 - 1, x + t % 3 - 1)));
        out[i] = s;
    }
}

// Sobel x / y on the blurred map, magnitude and the quantised orientation index (canny_ops.py:158-163,171).
// idx = (round((atan(gy/gx) * 360/pi + 180) / 45) * 45 / 45) % 8 ; NaN (0/0) stays NaN -> -1 (never "oriented").
__global__ void lwg_canny_grad_kernel(const float* __restrict__ blur, float* __restrict__ mag, int* __restrict__ oidx, int n,
                                      int H, int W, LwgK3 kx, LwgK3 ky) {
    const size_t total = (size_t)n * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        const float* img = blur + (i - (size_t)y * W - x);
        float gx = 0.f, gy = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float v = lwg_tap(img, H, W, y + t / 3 - 1, x + t % 3 - 1);
            gx = __fadd_rn(gx, __fmul_rn(kx.w[t], v));
            gy = __fadd_rn(gy, __fmul_rn(ky.w[t], v));
        }
        // every operation individually rounded (torch evaluates pow, add, sqrt, div, mul, add as separate fp32 ops)
        // sqrt and divide go through fp64 and round once to fp32: the fp32 v_sqrt / v_rcp expansions of this toolchain are
        // not correctly rounded (measured: 20 % of random inputs differ from IEEE sqrtf by 1 ulp), and the equality tests
        // of the non-maximum suppression need the IEEE value (2p+2 <= 53: the double rounding is innocuous)
        mag[i] = (float)sqrt((double)__fadd_rn(__fmul_rn(gx, gx), __fmul_rn(gy, gy)));
        const float ratio = (float)((double)gy / (double)gx);
        float o = __fadd_rn(__fmul_rn(atanf(ratio), (float)(360.0 / 3.14159265358979323846)), 180.f);
        o = __fmul_rn(rintf((float)((double)o / 45.0)), 45.f);
        const float q = fmodf((float)((double)o / 45.0), 8.f);
        oidx[i] = (q == q) ? (int)q : -1;
    }
}

// Non-maximum suppression along the orientation + double threshold -> 0, 0.5 (weak) or 1 (strong)
// (canny_ops.py:165-199).  Direction i in 0..7 = neighbour E, NE, N, NW, W, SW, S, SE (image y down).
__global__ void lwg_canny_nms_kernel(const float* __restrict__ mag, const int* __restrict__ oidx, float* __restrict__ thin,
                                     int n, int H, int W, float low, float high) {
    const int DX[8] = {1, 1, 0, -1, -1, -1, 0, 1};
    const int DY[8] = {0, -1, -1, -1, 0, 1, 1, 1};
    const size_t total = (size_t)n * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        const float* img = mag + (i - (size_t)y * W - x);
        float m = img[(size_t)y * W + x];
        const int o = oidx[i];
        if (o >= 0) {
            const int p = o & 3;  // the pair (p, p + 4)
            const float dpos = __fsub_rn(m, lwg_tap(img, H, W, y + DY[p], x + DX[p]));
            const float dneg = __fsub_rn(m, lwg_tap(img, H, W, y + DY[p + 4], x + DX[p + 4]));
            if (!(fminf(dpos, dneg) > 0.f)) m = 0.f;
        }
        thin[i] = (m > low ? 0.5f : 0.f) + (m > high ? 0.5f : 0.f);
    }
}

// Hysteresis (canny_ops.py:201-206): weak pixels with conv(thin, 1.25 * ones3x3) > 1 become edges.
__global__ void lwg_canny_hyst_kernel(const float* __restrict__ thin, float* __restrict__ edges, int n, int H, int W) {
    const size_t total = (size_t)n * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        const float* img = thin + (i - (size_t)y * W - x);
        const float c = img[(size_t)y * W + x];
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) s += 1.25f * lwg_tap(img, H, W, y + t / 3 - 1, x + t % 3 - 1);
        const float strong = c > 0.75f ? 1.f : 0.f;   // thin == 1.0
        const float weak = c == 0.5f ? 1.f : 0.f;
        edges[i] = strong + ((s > 1.f) ? weak : 0.f);
    }
}

// sil (n,1,H,W) -> thin edges (n,1,H,W) in {0,1}.  gauss/sobel: the reference's 3x3 kernels (row-major), passed by
// the host so the float64 -> float32 rounding of the weights is the reference's.  ws: 3*n*H*W floats.
extern "C" int lwg_canny_f32(const float* sil, int n, int H, int W, const float* gauss9, const float* sobelx9,
                             float low, float high, float* edges, float* ws, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!sil || !edges || !ws || !gauss9 || !sobelx9 || n <= 0 || H <= 0 || W <= 0) return (int)hipErrorInvalidValue;
    LwgK3 g, kx, ky;
    for (int t = 0; t < 9; ++t) {
        g.w[t] = gauss9[t];
        kx.w[t] = sobelx9[t];
        ky.w[t] = sobelx9[(t % 3) * 3 + t / 3];  // transpose
    }
    const size_t total = (size_t)n * H * W;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    float* blur = ws;
    float* mag = ws + total;
    int* oidx = reinterpret_cast<int*>(ws + 2 * total);
    hipLaunchKernelGGL(lwg_conv3_kernel, dim3(blocks), dim3(256), 0, stream, sil, blur, n, H, W, g);
    hipLaunchKernelGGL(lwg_canny_grad_kernel, dim3(blocks), dim3(256), 0, stream, blur, mag, oidx, n, H, W, kx, ky);
    float* thin = blur;  // blur is dead after the gradient pass
    hipLaunchKernelGGL(lwg_canny_nms_kernel, dim3(blocks), dim3(256), 0, stream, mag, oidx, thin, n, H, W, low, high);
    hipLaunchKernelGGL(lwg_canny_hyst_kernel, dim3(blocks), dim3(256), 0, stream, thin, edges, n, H, W);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------- boundary fill
// Ordered compaction of the boundary pixels of one image per workgroup (row-major = torch.nonzero order).
__global__ __launch_bounds__(1024) void lwg_compact_kernel(const float* __restrict__ edges, int HW, int* __restrict__ list,
                                                            int* __restrict__ count) {
    __shared__ int part[1024];
    const int img = blockIdx.x, tid = threadIdx.x;
    const float* e = edges + (size_t)img * HW;
    const int per = (HW + 1023) / 1024;
    const int lo = tid * per, hi = min(HW, lo + per);
    int c = 0;
    for (int i = lo; i < hi; ++i) c += e[i] != 0.f;
    part[tid] = c;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {  // inclusive Hillis-Steele scan
        const int v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int pos = part[tid] - c;
    int* dst = list + (size_t)img * HW;
    for (int i = lo; i < hi; ++i)
        if (e[i] != 0.f) dst[pos++] = i;
    if (tid == 1023) count[img] = part[1023];
}

// Every uncertain pixel (outpad * (1 - confidant) != 0) takes the distance-weighted colour of its 3 nearest boundary
// pixels, weights = d_k^2 / sum d^2 (the reference's formula, flowcomposition.py:288-291); other pixels =
// src * confidant.  Ties in distance resolve to the lowest boundary index.  top3 (optional): the three squared
// distances per pixel (n,3,H,W) int32, -1 where the pixel is not uncertain.
__global__ __launch_bounds__(256) void lwg_boundary_fill_kernel(const float* __restrict__ src, const float* __restrict__ conf,
                                                                 const float* __restrict__ outpad, const int* __restrict__ list,
                                                                 const int* __restrict__ count, int H, int W,
                                                                 float* __restrict__ out, int* __restrict__ top3) {
    __shared__ int pts[1024];
    const int img = blockIdx.y;
    const int HW = H * W;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool inb = i < HW;
    const float cf = inb ? conf[(size_t)img * HW + i] : 1.f;
    const float unc = inb ? outpad[(size_t)img * HW + i] * (1.f - cf) : 0.f;
    const bool active = unc != 0.f;
    const int py = inb ? i / W : 0, px = inb ? i - (i / W) * W : 0;
    const int n2 = count[img];
    const int* lst = list + (size_t)img * HW;
    long long d0 = 0x7fffffffffffffffLL, d1 = d0, d2 = d0;  // (dist << 32 | index), ascending
    const bool any = __syncthreads_or(active);
    if (any) {
        for (int base = 0; base < n2; base += 1024) {
            const int m = min(1024, n2 - base);
            __syncthreads();
            for (int j = threadIdx.x; j < m; j += 256) pts[j] = lst[base + j];
            __syncthreads();
            if (active) {
                for (int j = 0; j < m; ++j) {
                    const int q = pts[j];
                    const int qy = q / W, qx = q - qy * W;
                    const int dy = py - qy, dx = px - qx;
                    const long long key = ((long long)(dy * dy + dx * dx) << 32) | (unsigned)(base + j);
                    if (key < d2) {
                        if (key < d1) {
                            d2 = d1;
                            if (key < d0) { d1 = d0; d0 = key; } else d1 = key;
                        } else d2 = key;
                    }
                }
            }
        }
    }
    if (!inb) return;
    const float* s = src + (size_t)img * 3 * HW;
    float* o = out + (size_t)img * 3 * HW;
    if (active && n2 >= 3) {
        const float v0 = (float)(d0 >> 32), v1 = (float)(d1 >> 32), v2 = (float)(d2 >> 32);
        const float sum = (v0 + v1) + v2;
        const float w0 = v0 / sum, w1 = v1 / sum, w2 = v2 / sum;
        const int q0 = lst[(int)(d0 & 0xffffffff)], q1 = lst[(int)(d1 & 0xffffffff)], q2 = lst[(int)(d2 & 0xffffffff)];
#pragma unroll
        for (int c = 0; c < 3; ++c)
            o[(size_t)c * HW + i] = (s[(size_t)c * HW + q0] * w0 + s[(size_t)c * HW + q1] * w1) + s[(size_t)c * HW + q2] * w2;
        if (top3) {
            int* t = top3 + (size_t)img * 3 * HW;
            t[i] = (int)(d0 >> 32); t[HW + i] = (int)(d1 >> 32); t[2 * HW + i] = (int)(d2 >> 32);
        }
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) o[(size_t)c * HW + i] = s[(size_t)c * HW + i] * cf;
        if (top3) {
            int* t = top3 + (size_t)img * 3 * HW;
            t[i] = -1; t[HW + i] = -1; t[2 * HW + i] = -1;
        }
    }
}

// src (n,3,H,W), confidant / outpad / edges (n,1,H,W) -> morph image (n,3,H,W).  ws: n*(H*W + 1) ints
// (boundary lists + counts; counts at ws[n*H*W + img], readable by the host after the stream drains).
extern "C" int lwg_boundary_fill_f32(const float* src, const float* confidant, const float* outpad, const float* edges,
                                     int n, int H, int W, float* out, int32_t* top3, int32_t* ws, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!src || !confidant || !outpad || !edges || !out || !ws || n <= 0 || H <= 0 || W <= 0) return (int)hipErrorInvalidValue;
    const int HW = H * W;
    int* list = ws;
    int* count = ws + (size_t)n * HW;
    hipLaunchKernelGGL(lwg_compact_kernel, dim3(n), dim3(1024), 0, stream, edges, HW, list, count);
    hipLaunchKernelGGL(lwg_boundary_fill_kernel, dim3((HW + 255) / 256, n), dim3(256), 0, stream, src, confidant, outpad, list,
                       count, H, W, out, top3);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------- grid_sample
// F.grid_sample(img (n,C,H,W), grid (n,Ho,Wo,2)) bilinear, zeros padding, align_corners=False -> (n,C,Ho,Wo).
// img_bstride = 0 broadcasts one image over the batch.  Corner order nw, ne, sw, se as ATen's CPU kernel.
__global__ void lwg_grid_sample_kernel(const float* __restrict__ img, size_t img_bstride, const float* __restrict__ grid, int n,
                                       int C, int H, int W, int Ho, int Wo, float* __restrict__ out) {
    const size_t total = (size_t)n * Ho * Wo;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / ((size_t)Ho * Wo));
        const size_t pix = i - (size_t)b * Ho * Wo;
        const float gx = grid[2 * i], gy = grid[2 * i + 1];
        const float ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f, iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
        const float fx0 = floorf(ix), fy0 = floorf(iy);
        const float wx1 = ix - fx0, wy1 = iy - fy0, wx0 = (fx0 + 1.f) - ix, wy0 = (fy0 + 1.f) - iy;
        const int x0 = (int)fminf(fmaxf(fx0, -2.f), (float)W + 1.f), y0 = (int)fminf(fmaxf(fy0, -2.f), (float)H + 1.f);
        const float* src = img + (size_t)b * img_bstride;
        for (int c = 0; c < C; ++c) {
            const float* ch = src + (size_t)c * H * W;
            float s = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int yy = y0 + (t >> 1), xx = x0 + (t & 1);
                if (yy >= 0 && yy < H && xx >= 0 && xx < W)
                    s += ch[(size_t)yy * W + xx] * (((t >> 1) ? wy1 : wy0) * ((t & 1) ? wx1 : wx0));
            }
            out[((size_t)b * C + c) * Ho * Wo + pix] = s;
        }
    }
}

extern "C" int lwg_grid_sample_nchw_f32(const float* img, size_t img_bstride, const float* grid, int n, int C, int H, int W,
                                        int Ho, int Wo, float* out, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!img || !grid || !out || n <= 0 || C <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return (int)hipErrorInvalidValue;
    const size_t total = (size_t)n * Ho * Wo;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(lwg_grid_sample_kernel, dim3(blocks), dim3(256), 0, stream, img, img_bstride, grid, n, C, H, W, Ho, Wo, out);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------- UV merge
// flowcomposition.py:123-130: src_warp (ns,3,H,W), vis (ns,1,H,W) (already dilated) -> merge (3,H,W).
__global__ void lwg_uv_merge_kernel(const float* __restrict__ warp, const float* __restrict__ vis, int ns, int HW,
                                    float* __restrict__ out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
        float vis_sum = 0.f, t0 = 0.f, t1 = 0.f, t2 = 0.f;
        for (int s = 1; s < ns; ++s) {
            const float v = vis[(size_t)s * HW + i];
            vis_sum += v;
            const float* w = warp + (size_t)s * 3 * HW;
            t0 += w[i] * v; t1 += w[(size_t)HW + i] * v; t2 += w[2 * (size_t)HW + i] * v;
        }
        const float den = vis_sum + 1e-5f;
        const float fi = (1.f - vis[i]) * (vis_sum >= 1.f ? 1.f : 0.f);
        out[i] = warp[i] * (1.f - fi) + (t0 / den) * fi;
        out[(size_t)HW + i] = warp[(size_t)HW + i] * (1.f - fi) + (t1 / den) * fi;
        out[2 * (size_t)HW + i] = warp[2 * (size_t)HW + i] * (1.f - fi) + (t2 / den) * fi;
    }
}

extern "C" int lwg_uv_merge_f32(const float* src_warp, const float* vis, int ns, int H, int W, float* out, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!src_warp || !vis || !out || ns <= 0 || H <= 0 || W <= 0) return (int)hipErrorInvalidValue;
    const int HW = H * W;
    hipLaunchKernelGGL(lwg_uv_merge_kernel, dim3((HW + 255) / 256), dim3(256), 0, stream, src_warp, vis, ns, HW, out);
    return (int)hipGetLastError();
}

// Swapper: UV images of several people merged by their selected-part visibility (flowcomposition.py:816-856, merge_uv_img):
// norm_i = vis_i / (sum_j vis_j + 1e-7);  out = sum_i uv_i * norm_i, in the reference's order of operations.
__global__ void lwg_uv_merge_parts_kernel(const float* __restrict__ uv, const float* __restrict__ vis, int n, int HW,
                                          float* __restrict__ out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
        float vs = 0.f;
        for (int s = 0; s < n; ++s) vs += vis[(size_t)s * HW + i];
        const float den = vs + 1e-7f;
        float o0 = 0.f, o1 = 0.f, o2 = 0.f;
        for (int s = 0; s < n; ++s) {
            const float nv = vis[(size_t)s * HW + i] / den;
            const float* u = uv + (size_t)s * 3 * HW;
            o0 += u[i] * nv; o1 += u[(size_t)HW + i] * nv; o2 += u[2 * (size_t)HW + i] * nv;
        }
        out[i] = o0; out[(size_t)HW + i] = o1; out[2 * (size_t)HW + i] = o2;
    }
}

extern "C" int lwg_uv_merge_parts_f32(const float* uv_imgs, const float* vis, int n, int H, int W, float* out, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!uv_imgs || !vis || !out || n <= 0 || H <= 0 || W <= 0) return (int)hipErrorInvalidValue;
    const int HW = H * W;
    hipLaunchKernelGGL(lwg_uv_merge_parts_kernel, dim3((HW + 255) / 256), dim3(256), 0, stream, uv_imgs, vis, n, HW, out);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------- network inputs
// cat[a * mask?, b] (NCHW planes) -> NHWC with Cp channels (zero padded): the bg net input [img * m, m] (NHWC-4) and the
// SIDNet input [morph_img, cond] (NHWC-8) (flowcomposition.py:250-266) written straight in the engine's layout.
__global__ void lwg_pack_inputs_kernel(const float* __restrict__ a, int Ca, const float* __restrict__ b, int Cb,
                                       const float* __restrict__ mask, int n, int HW, int Cp, float* __restrict__ out) {
    const size_t total = (size_t)n * HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int img = (int)(i / HW);
        const size_t pix = i - (size_t)img * HW;
        const float m = mask ? mask[i] : 1.f;
        float* o = out + i * Cp;
        int c = 0;
        for (int k = 0; k < Ca; ++k) o[c++] = a[((size_t)img * Ca + k) * HW + pix] * m;
        for (int k = 0; k < Cb; ++k) o[c++] = b[((size_t)img * Cb + k) * HW + pix];
        for (; c < Cp; ++c) o[c] = 0.f;
    }
}

extern "C" int lwg_pack_inputs_f32(const float* a, int Ca, const float* b, int Cb, const float* mask, int n, int H, int W,
                                   int Cp, float* out, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!a || !out || Ca <= 0 || Cb < 0 || (Cb > 0 && !b) || Ca + Cb > Cp || n <= 0) return (int)hipErrorInvalidValue;
    const size_t total = (size_t)n * H * W;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(lwg_pack_inputs_kernel, dim3(blocks), dim3(256), 0, stream, a, Ca, b, Cb, mask, n, H * W, Cp, out);
    return (int)hipGetLastError();
}
