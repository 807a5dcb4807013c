// Triangle rasterizer: face-index map + barycentric weight map (the `neural_renderer` CUDA dependency of the
// reference, call sites renders/nmr.py:337,356 `nr.rasterize_face_index_map_and_weight_map`), plus the
// projection / look_at / vertices_to_faces prologue (renders/nmr.py:34-52,:326-336).
//
// Same per-pixel arithmetic, in the same order, as oracle/raster_oracle.c (compile with -ffp-contract=off):
// back-face cull, three edge functions, clamped + renormalised weights, depth 1/sum(w/z), nearest wins, the
// lowest face id wins exact depth ties (order-independent formulation of "first face wins").
//
// MI355X design: instead of the upstream per-pixel loop over all 13776 faces (3.6 G tests at 512^2), one
// workgroup owns a 16x16 pixel tile; faces are streamed in chunks of 256, each lane tests one face's pixel
// bounding box against the tile, survivors are compacted into LDS together with their 80-byte setup record,
// and only those (a handful per chunk) are evaluated per pixel from LDS broadcasts.
#include "lwg_common.h"
#include "lwg_conv_args.h"

#define LWG_REC_FLOATS 20  // f[9], inv[9], 2 pad -> five 16-byte loads

// ---- projection: verts (B,nv,3), cam (B,3) = (s,tx,ty), faces (nf,3) -> faces_v (B,nf,3,3), f2pts (B,nf,3,2) ----
__global__ void lwg_project_faces_kernel(const float* __restrict__ verts, const float* __restrict__ cam,
                                         const int* __restrict__ faces, int B, int nv, int nf, float eye_dist,
                                         float* __restrict__ faces_v, float* __restrict__ f2pts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * nf * 3) return;
    const int b = i / (nf * 3), fv = i - b * nf * 3;
    const int vid = faces[fv];
    const float s = cam[b * 3 + 0], tx = cam[b * 3 + 1], ty = cam[b * 3 + 2];
    const float* v = verts + ((size_t)b * nv + vid) * 3;
    const float px = s * (v[0] + tx);
    const float py = s * (v[1] + ty);
    if (faces_v) {
        float* o = faces_v + (size_t)i * 3;
        o[0] = px;
        o[1] = -py;               // y flipped into the rasterizer's y-up space (nmr.py:331)
        o[2] = v[2] + eye_dist;   // look_at with eye (0,0,-d), at 0, up +y: identity rotation, z + d
    }
    if (f2pts) {
        f2pts[(size_t)i * 2 + 0] = px;
        f2pts[(size_t)i * 2 + 1] = py;  // (-1) * (-py): the wrapper un-flips y (nmr.py:339-340)
    }
}

// ---- per-face setup: inverse matrix + conservative pixel bounding box ----
__global__ void lwg_raster_setup_kernel(const float* __restrict__ faces_v, int total, int S,
                                        float* __restrict__ rec, short4* __restrict__ bbox) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const float* f = faces_v + (size_t)i * 9;
    float v[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) v[k] = f[k];
    short4 bb = make_short4(32767, -32768, 32767, -32768);  // empty: overlaps no tile
    float* r = rec + (size_t)i * LWG_REC_FLOATS;
    const bool front = !((v[7] - v[1]) * (v[3] - v[0]) < (v[4] - v[1]) * (v[6] - v[0]));
    float inv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) inv[k] = 0.f;
    if (front) {
        float p[3][2];
        const float fs = (float)S;
#pragma unroll
        for (int n = 0; n < 3; ++n)
#pragma unroll
            for (int d = 0; d < 2; ++d) p[n][d] = 0.5f * (v[3 * n + d] * fs + fs - 1.0f);
        const float den = p[2][0] * (p[0][1] - p[1][1]) + p[0][0] * (p[1][1] - p[2][1]) + p[1][0] * (p[2][1] - p[0][1]);
        const float m[9] = {
            p[1][1] - p[2][1], p[2][0] - p[1][0], p[1][0] * p[2][1] - p[2][0] * p[1][1],
            p[2][1] - p[0][1], p[0][0] - p[2][0], p[2][0] * p[0][1] - p[0][0] * p[2][1],
            p[0][1] - p[1][1], p[1][0] - p[0][0], p[0][0] * p[1][1] - p[1][0] * p[0][1]};
#pragma unroll
        for (int k = 0; k < 9; ++k) inv[k] = m[k] / den;
        const float xmin = fminf(p[0][0], fminf(p[1][0], p[2][0])), xmax = fmaxf(p[0][0], fmaxf(p[1][0], p[2][0]));
        const float ymin = fminf(p[0][1], fminf(p[1][1], p[2][1])), ymax = fmaxf(p[0][1], fmaxf(p[1][1], p[2][1]));
        const float lim = (float)S + 1.f;
        // one pixel of margin on every side: the inside test is evaluated in fp32 on normalised coordinates
        if (xmin == xmin && xmax == xmax && ymin == ymin && ymax == ymax) {
            const int x0 = (int)fmaxf(floorf(xmin) - 1.f, -1.f), x1 = (int)fminf(ceilf(xmax) + 1.f, lim);
            const int y0 = (int)fmaxf(floorf(ymin) - 1.f, -1.f), y1 = (int)fminf(ceilf(ymax) + 1.f, lim);
            if (xmax >= -2.f && ymax >= -2.f && xmin <= lim && ymin <= lim)
                bb = make_short4((short)x0, (short)x1, (short)y0, (short)y1);
        }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) { r[k] = v[k]; r[9 + k] = inv[k]; }
    r[18] = 0.f; r[19] = 0.f;
    bbox[i] = bb;
}

__device__ __forceinline__ float lwg_clamp01(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }

#define LWG_RCHUNK 256
__global__ __launch_bounds__(256) void lwg_raster_tiles_kernel(const float* __restrict__ rec, const short4* __restrict__ bbox,
                                                              int nf, int S, float near, float far,
                                                              int* __restrict__ fim, float* __restrict__ wim) {
    __shared__ __attribute__((aligned(16))) float srec[LWG_RCHUNK][LWG_REC_FLOATS];
    __shared__ int sid[LWG_RCHUNK];
    __shared__ int scount;
    const int tid = threadIdx.x;
    const int b = blockIdx.z;
    const int xi = blockIdx.x * 16 + (tid & 15);
    const int r = blockIdx.y * 16 + (tid >> 4);
    const int yi = S - 1 - r;  // row 0 is the top of the image = largest y (vertical flip of the upstream maps)
    const int tx0 = blockIdx.x * 16, tx1 = tx0 + 15;
    const int ty1 = S - 1 - blockIdx.y * 16, ty0 = ty1 - 15;
    const float xp = (float)((2.0 * xi + 1 - S) / S);
    const float yp = (float)((2.0 * yi + 1 - S) / S);
    const float fxi = (float)xi, fyi = (float)yi;
    const float* recb = rec + (size_t)b * nf * LWG_REC_FLOATS;
    const short4* bbb = bbox + (size_t)b * nf;

    float zmin = far;
    int best = -1;
    float wb0 = 0.f, wb1 = 0.f, wb2 = 0.f;

    for (int base = 0; base < nf; base += LWG_RCHUNK) {
        if (tid == 0) scount = 0;
        __syncthreads();
        const int fi = base + tid;
        if (fi < nf) {
            const short4 bb = bbb[fi];
            if (bb.x <= tx1 && bb.y >= tx0 && bb.z <= ty1 && bb.w >= ty0) {
                const int slot = atomicAdd(&scount, 1);
                const floatx4* src = reinterpret_cast<const floatx4*>(recb + (size_t)fi * LWG_REC_FLOATS);
                floatx4* dst = reinterpret_cast<floatx4*>(&srec[slot][0]);
#pragma unroll
                for (int k = 0; k < 5; ++k) dst[k] = src[k];
                sid[slot] = fi;
            }
        }
        __syncthreads();
        const int n = scount;
        for (int e = 0; e < n; ++e) {
            const float* f = &srec[e][0];
            if (((yp - f[1]) * (f[3] - f[0]) < (xp - f[0]) * (f[4] - f[1])) ||
                ((yp - f[4]) * (f[6] - f[3]) < (xp - f[3]) * (f[7] - f[4])) ||
                ((yp - f[7]) * (f[0] - f[6]) < (xp - f[6]) * (f[1] - f[7])))
                continue;
            const float* m = f + 9;
            float w0 = m[0] * fxi + m[1] * fyi + m[2];
            float w1 = m[3] * fxi + m[4] * fyi + m[5];
            float w2 = m[6] * fxi + m[7] * fyi + m[8];
            w0 = lwg_clamp01(w0); w1 = lwg_clamp01(w1); w2 = lwg_clamp01(w2);
            const float ws = w0 + w1 + w2;
            w0 = w0 / ws; w1 = w1 / ws; w2 = w2 / ws;
            const float zp = 1.0f / (w0 / f[2] + w1 / f[5] + w2 / f[8]);
            if (zp <= near || far <= zp) continue;
            const int fid = sid[e];
            if (zp < zmin || (zp == zmin && best >= 0 && fid < best)) {
                zmin = zp; best = fid; wb0 = w0; wb1 = w1; wb2 = w2;
            }
        }
        __syncthreads();
    }
    if (xi < S && r < S) {
        const size_t o = ((size_t)b * S + r) * S + xi;
        fim[o] = best;
        wim[3 * o + 0] = wb0; wim[3 * o + 1] = wb1; wim[3 * o + 2] = wb2;
    }
}

extern "C" size_t lwg_rasterize_ws_bytes(int B, int nf) {
    return (size_t)B * nf * (LWG_REC_FLOATS * sizeof(float) + sizeof(short4));
}

extern "C" int lwg_project_faces_f32(const float* verts, const float* cam, const int32_t* faces, int B, int nv, int nf,
                                     float eye_dist, float* faces_v, float* f2pts, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!verts || !cam || !faces || B <= 0 || nv <= 0 || nf <= 0 || (!faces_v && !f2pts)) return (int)hipErrorInvalidValue;
    const int total = B * nf * 3;
    hipLaunchKernelGGL(lwg_project_faces_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, verts, cam, faces, B, nv,
                       nf, eye_dist, faces_v, f2pts);
    return (int)hipGetLastError();
}

// faces_v (B,nf,3,3) -> fim (B,S,S) int32 (-1 background), wim (B,S,S,3).  ws: lwg_rasterize_ws_bytes(B,nf) bytes.
extern "C" int lwg_rasterize_fim_wim_f32(const float* faces_v, int B, int nf, int S, float near, float far, int32_t* fim,
                                         float* wim, void* ws, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!faces_v || !fim || !wim || !ws || B <= 0 || nf <= 0 || S <= 0 || S > 16384 || B > 65535) return (int)hipErrorInvalidValue;
    float* rec = reinterpret_cast<float*>(ws);
    short4* bbox = reinterpret_cast<short4*>(rec + (size_t)B * nf * LWG_REC_FLOATS);
    const int total = B * nf;
    hipLaunchKernelGGL(lwg_raster_setup_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, faces_v, total, S, rec, bbox);
    const int tiles = (S + 15) / 16;
    hipLaunchKernelGGL(lwg_raster_tiles_kernel, dim3(tiles, tiles, B), dim3(256), 0, stream, rec, bbox, nf, S, near, far,
                       fim, wim);
    return (int)hipGetLastError();
}
