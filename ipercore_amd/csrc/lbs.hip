// SMPL / SMPL-H linear blend skinning for a batch of frames.
// Replaces reference tools/human_digitalizer/smplx/lbs.py:137-227 (lbs), :230-271 (vertices2joints,
// blend_shapes), :321-375 (batch_rigid_transform), tools/utils/geometry/rotations.py:318-375
// (rotvec -> quaternion -> rotation matrix), bodynets/base_smpl.py:28-50 (link), :7-18 (j2d projection).
//
// The reference runs this per frame with B = 1: a 38 MB posedirs GEMV plus 52 sequential 4x4 matmul launches.
// Here the whole frame batch goes through five small kernels; the pose-blend tensor is streamed ONCE per
// group of up to 8 frames by 323 workgroups (HBM-bound: 38 MB / group), everything else is L2-resident.
#include "lwg_common.h"
#include "lwg_conv_args.h"

#define LWG_MAX_JOINTS 64
#define LWG_LBS_FB 8  // frames per thread in the skinning kernel

// v_shaped[b,v,k] = v_template[v,k] + offsets[(b),v,k] + sum_l beta[b,l] * shapedirs[v,k,l]
__global__ void lwg_lbs_shape_kernel(const float* __restrict__ v_template, const float* __restrict__ offsets, int off_batched,
                                     const float* __restrict__ shapedirs, const float* __restrict__ beta, int nbeta, int B,
                                     int nv3, float* __restrict__ v_shaped) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * nv3) return;
    const int b = i / nv3, e = i - b * nv3;
    float acc = 0.f;
    for (int l = 0; l < nbeta; ++l) acc += beta[b * nbeta + l] * shapedirs[(size_t)e * nbeta + l];
    float base = v_template[e];
    if (offsets) base += offsets[off_batched ? (size_t)i : (size_t)e];
    v_shaped[i] = base + acc;
}

// J[b,j,k] = sum_v J_regressor[j,v] * v_shaped[b,v,k]     grid (nj, B), 256 threads
__global__ __launch_bounds__(256) void lwg_lbs_joints_kernel(const float* __restrict__ Jreg, const float* __restrict__ v_shaped,
                                                            int nv, int nj, float* __restrict__ J) {
    const int j = blockIdx.x, b = blockIdx.y;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int v = threadIdx.x; v < nv; v += 256) {
        const float w = Jreg[(size_t)j * nv + v];
        const float* p = v_shaped + ((size_t)b * nv + v) * 3;
        a0 += w * p[0]; a1 += w * p[1]; a2 += w * p[2];
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
    __shared__ float sh[4][3];
    const int wid = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[wid][0] = a0; sh[wid][1] = a1; sh[wid][2] = a2; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        J[((size_t)b * nj + j) * 3 + k] = (sh[0][k] + sh[1][k]) + (sh[2][k] + sh[3][k]);
    }
}

// One block per frame: rotation matrices, pose feature, kinematic chain, relative transforms A (3x4 per joint),
// posed joints j3d and their weak-perspective projection j2d.
__global__ __launch_bounds__(64) void lwg_lbs_pose_kernel(const float* __restrict__ pose, int pose_stride,
                                                         const float* __restrict__ cam, int cam_stride,
                                                         const float* __restrict__ J, const int* __restrict__ parents, int nj,
                                                         float* __restrict__ pose_feature, float* __restrict__ A,
                                                         float* __restrict__ j3d, float* __restrict__ j2d) {
    __shared__ float R[LWG_MAX_JOINTS][9];
    __shared__ float G[LWG_MAX_JOINTS][12];  // world transforms, rows of [R | t]
    __shared__ float Jl[LWG_MAX_JOINTS][3];
    const int b = blockIdx.x, j = threadIdx.x;
    if (j < nj) {
        const float* rv = pose + (size_t)b * pose_stride + 3 * j;
        const float x = rv[0], y = rv[1], z = rv[2];
        const float ex = x + 1e-8f, ey = y + 1e-8f, ez = z + 1e-8f;
        const float ang = sqrtf(ex * ex + ey * ey + ez * ez);
        const float nx = x / ang, ny = y / ang, nz = z / ang;
        const float half = ang * 0.5f;
        const float c = cosf(half), s = sinf(half);
        float qw = c, qx = s * nx, qy = s * ny, qz = s * nz;
        const float qn = sqrtf(qw * qw + qx * qx + qy * qy + qz * qz);
        qw /= qn; qx /= qn; qy /= qn; qz /= qn;
        const float w2 = qw * qw, x2 = qx * qx, y2 = qy * qy, z2 = qz * qz;
        const float wx = qw * qx, wy = qw * qy, wz = qw * qz, xy = qx * qy, xz = qx * qz, yz = qy * qz;
        float* r = R[j];
        r[0] = w2 + x2 - y2 - z2; r[1] = 2 * xy - 2 * wz;     r[2] = 2 * wy + 2 * xz;
        r[3] = 2 * wz + 2 * xy;     r[4] = w2 - x2 + y2 - z2; r[5] = 2 * yz - 2 * wx;
        r[6] = 2 * xz - 2 * wy;     r[7] = 2 * wx + 2 * yz;     r[8] = w2 - x2 - y2 + z2;
        for (int k = 0; k < 3; ++k) Jl[j][k] = J[((size_t)b * nj + j) * 3 + k];
        if (j >= 1) {
            float* pf = pose_feature + (size_t)b * (nj - 1) * 9 + (size_t)(j - 1) * 9;
            for (int k = 0; k < 9; ++k) pf[k] = r[k] - ((k == 0 || k == 4 || k == 8) ? 1.f : 0.f);
        }
    }
    __syncthreads();
    if (j == 0) {
        // kinematic chain in index order (parents[i] < i): G_i = G_parent * [R_i | J_i - J_parent]
        for (int i = 0; i < nj; ++i) {
            float t[3];
            const int p = i == 0 ? -1 : parents[i];
            for (int k = 0; k < 3; ++k) t[k] = Jl[i][k] - (p >= 0 ? Jl[p][k] : 0.f);
            if (p < 0) {
                for (int rr = 0; rr < 3; ++rr) {
                    for (int cc = 0; cc < 3; ++cc) G[i][rr * 4 + cc] = R[i][rr * 3 + cc];
                    G[i][rr * 4 + 3] = t[rr];
                }
            } else {
                for (int rr = 0; rr < 3; ++rr) {
                    const float g0 = G[p][rr * 4 + 0], g1 = G[p][rr * 4 + 1], g2 = G[p][rr * 4 + 2], g3 = G[p][rr * 4 + 3];
                    for (int cc = 0; cc < 3; ++cc)
                        G[i][rr * 4 + cc] = g0 * R[i][cc] + g1 * R[i][3 + cc] + g2 * R[i][6 + cc];
                    G[i][rr * 4 + 3] = g0 * t[0] + g1 * t[1] + g2 * t[2] + g3;
                }
            }
        }
    }
    __syncthreads();
    if (j < nj) {
        // A = G - [0 | G_rot * J]  (lbs.py:370-373): translation column minus the rotated rest joint
        float* a = A + ((size_t)b * nj + j) * 12;
        for (int rr = 0; rr < 3; ++rr) {
            const float g0 = G[j][rr * 4 + 0], g1 = G[j][rr * 4 + 1], g2 = G[j][rr * 4 + 2];
            a[rr * 4 + 0] = g0; a[rr * 4 + 1] = g1; a[rr * 4 + 2] = g2;
            a[rr * 4 + 3] = G[j][rr * 4 + 3] - (g0 * Jl[j][0] + g1 * Jl[j][1] + g2 * Jl[j][2]);
        }
        const float px = G[j][3], py = G[j][7], pz = G[j][11];
        float* o3 = j3d + ((size_t)b * nj + j) * 3;
        o3[0] = px; o3[1] = py; o3[2] = pz;
        if (j2d && cam) {
            const float* cm = cam + (size_t)b * cam_stride;
            j2d[((size_t)b * nj + j) * 2 + 0] = cm[0] * (px + cm[1]);
            j2d[((size_t)b * nj + j) * 2 + 1] = cm[0] * (py + cm[2]);
        }
    }
}

// v_posed[b,col] = v_shaped[b,col] + sum_p pose_feature[b,p] * posedirs[p,col]      (col = 3*v + k)
// grid (ceil(3nv/64), ceil(B/FB)), 4 waves: a lane owns one column for up to FB frames (posedirs is streamed once per
// frame group, 256 contiguous bytes per wave load), wave w takes the rows p = w, w+4, ...; the four partial sums are
// combined in wave order (fixed order: a frame's result does not depend on the batch it is in).  The multiply-adds are EXPLICIT
// fused fmaf: left to -ffp-contract the compiler packed some of the FB frame slots into v_pk_mul / v_pk_add pairs and fused the
// others, so frames 6 and 7 of an 8-frame batch differed in the last bit from the same frames skinned alone (slot 0).
__global__ __launch_bounds__(256) void lwg_lbs_posed_kernel(const float* __restrict__ v_shaped, const float* __restrict__ posedirs,
                                                           const float* __restrict__ pose_feature, int npf, int nv3, int B,
                                                           float* __restrict__ v_posed) {
    extern __shared__ float sm[];
    float* spf = sm;                                  // [FB][npf]
    float* part = sm + LWG_LBS_FB * npf;              // [4][FB][64]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int b0 = blockIdx.y * LWG_LBS_FB;
    const int nb = min(LWG_LBS_FB, B - b0);
    for (int i = threadIdx.x; i < LWG_LBS_FB * npf; i += 256) {
        const int fb = i / npf;
        spf[i] = fb < nb ? pose_feature[(size_t)(b0 + fb) * npf + (i - fb * npf)] : 0.f;
    }
    __syncthreads();
    const int col = blockIdx.x * 64 + lane;
    const bool ok = col < nv3;
    float po[LWG_LBS_FB];
#pragma unroll
    for (int fb = 0; fb < LWG_LBS_FB; ++fb) po[fb] = 0.f;
    if (ok) {
        int p = wid;
        for (; p + 12 < npf; p += 16) {               // four rows in flight per wave
            float d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) d[u] = posedirs[(size_t)(p + 4 * u) * nv3 + col];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int fb = 0; fb < LWG_LBS_FB; ++fb) po[fb] = __builtin_fmaf(spf[fb * npf + p + 4 * u], d[u], po[fb]);
        }
        for (; p < npf; p += 4) {
            const float d = posedirs[(size_t)p * nv3 + col];
#pragma unroll
            for (int fb = 0; fb < LWG_LBS_FB; ++fb) po[fb] = __builtin_fmaf(spf[fb * npf + p], d, po[fb]);
        }
    }
#pragma unroll
    for (int fb = 0; fb < LWG_LBS_FB; ++fb) part[(wid * LWG_LBS_FB + fb) * 64 + lane] = po[fb];
    __syncthreads();
    if (wid == 0 && ok) {
        for (int fb = 0; fb < nb; ++fb) {
            const float t = ((part[(0 * LWG_LBS_FB + fb) * 64 + lane] + part[(1 * LWG_LBS_FB + fb) * 64 + lane]) +
                             part[(2 * LWG_LBS_FB + fb) * 64 + lane]) + part[(3 * LWG_LBS_FB + fb) * 64 + lane];
            v_posed[(size_t)(b0 + fb) * nv3 + col] = t + v_shaped[(size_t)(b0 + fb) * nv3 + col];
        }
    }
}

// verts[b,v] = (sum_j W[v,j] A[b,j]) * [v_posed[b,v]; 1]        grid (ceil(nv/256), B); A[b] staged in LDS
__global__ __launch_bounds__(256) void lwg_lbs_blend_kernel(const float* __restrict__ v_posed, const float* __restrict__ W,
                                                           const float* __restrict__ A, int nj, int nv,
                                                           float* __restrict__ verts) {
    extern __shared__ float sA[];                     // [nj][12]
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < nj * 12; i += 256) sA[i] = A[(size_t)b * nj * 12 + i];
    __syncthreads();
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= nv) return;
    const float* vp = v_posed + ((size_t)b * nv + v) * 3;
    const float x = vp[0], y = vp[1], z = vp[2];
    float T[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) T[k] = 0.f;
    for (int j = 0; j < nj; ++j) {
        const float w = W[(size_t)v * nj + j];
        const float* a = sA + j * 12;
#pragma unroll
        for (int k = 0; k < 12; ++k) T[k] += w * a[k];
    }
    float* o = verts + ((size_t)b * nv + v) * 3;
    o[0] = T[0] * x + T[1] * y + T[2] * z + T[3];
    o[1] = T[4] * x + T[5] * y + T[6] * z + T[7];
    o[2] = T[8] * x + T[9] * y + T[10] * z + T[11];
}

__global__ void lwg_lbs_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// linked[b, ids[i,0]] = verts[b, ids[i,1]]  (dst must already hold a copy of verts)
__global__ void lwg_lbs_link_kernel(const float* __restrict__ src, float* __restrict__ dst, const int* __restrict__ ids,
                                    int nlinks, int nv, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * nlinks) return;
    const int b = i / nlinks, l = i - b * nlinks;
    const int to = ids[2 * l], from = ids[2 * l + 1];
    const float* s = src + ((size_t)b * nv + from) * 3;
    float* d = dst + ((size_t)b * nv + to) * 3;
    d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
}

extern "C" size_t lwg_smpl_lbs_ws_floats(int B, int nv, int nj) {
    return (size_t)B * nv * 3 * 2 + (size_t)B * nj * 3 + (size_t)B * (nj - 1) * 9 + (size_t)B * nj * 12 + 64;
}

// pose: (B, 3*nj) axis-angle rows with row stride pose_stride floats; beta (B,nbeta); cam (B,>=3) row stride cam_stride
// or NULL; offsets: NULL, (nv,3) or (B,nv,3) (off_batched); links: NULL or (nlinks,2) int32 (to, from).
// Outputs: verts (B,nv,3), j3d (B,nj,3), j2d (B,nj,2) or NULL.
extern "C" int lwg_smpl_lbs_f32(const float* pose, int pose_stride, const float* beta, int beta_stride, int nbeta,
                                const float* cam, int cam_stride, const float* v_template, const float* offsets,
                                int off_batched, const float* shapedirs, const float* posedirs, const float* J_regressor,
                                const int32_t* parents, const float* lbs_weights, const int32_t* links, int nlinks, int B,
                                int nv, int nj, float* verts, float* j3d, float* j2d, float* ws, lwg_stream_t stream_) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    if (!pose || !beta || !v_template || !shapedirs || !posedirs || !J_regressor || !parents || !lbs_weights || !verts || !j3d ||
        !ws || B <= 0 || nv <= 0 || nj <= 1 || nj > LWG_MAX_JOINTS || beta_stride != nbeta)
        return (int)hipErrorInvalidValue;
    const int nv3 = nv * 3, npf = (nj - 1) * 9;
    float* v_shaped = ws;
    float* vraw = v_shaped + (size_t)B * nv3;
    float* J = vraw + (size_t)B * nv3;
    float* pf = J + (size_t)B * nj * 3;
    float* A = pf + (size_t)B * npf;
    hipLaunchKernelGGL(lwg_lbs_shape_kernel, dim3((B * nv3 + 255) / 256), dim3(256), 0, stream, v_template, offsets, off_batched,
                       shapedirs, beta, nbeta, B, nv3, v_shaped);
    hipLaunchKernelGGL(lwg_lbs_joints_kernel, dim3(nj, B), dim3(256), 0, stream, J_regressor, v_shaped, nv, nj, J);
    hipLaunchKernelGGL(lwg_lbs_pose_kernel, dim3(B), dim3(64), 0, stream, pose, pose_stride, cam, cam_stride, J, parents, nj, pf, A,
                       j3d, j2d);
    const size_t lds = (size_t)LWG_LBS_FB * (npf + 4 * 64) * sizeof(float);
    float* skin_out = (links && nlinks > 0) ? vraw : verts;
    float* v_posed = (links && nlinks > 0) ? verts : vraw;      // scratch for the pose-blended rest vertices
    hipLaunchKernelGGL(lwg_lbs_posed_kernel, dim3((nv3 + 63) / 64, (B + LWG_LBS_FB - 1) / LWG_LBS_FB), dim3(256), lds, stream,
                       v_shaped, posedirs, pf, npf, nv3, B, v_posed);
    hipLaunchKernelGGL(lwg_lbs_blend_kernel, dim3((nv + 255) / 256, B), dim3(256), (size_t)nj * 12 * sizeof(float), stream,
                       v_posed, lbs_weights, A, nj, nv, skin_out);
    if (links && nlinks > 0) {
        // a copy kernel, not hipMemcpyAsync: the per-frame path is replayed as a hipGraph of kernel nodes only
        const size_t ncopy = (size_t)B * nv3;
        hipLaunchKernelGGL(lwg_lbs_copy_kernel, dim3((unsigned)((ncopy + 255) / 256)), dim3(256), 0, stream, vraw, verts, ncopy);
        hipLaunchKernelGGL(lwg_lbs_link_kernel, dim3((B * nlinks + 255) / 256), dim3(256), 0, stream, vraw, verts, links, nlinks, nv, B);
    }
    return (int)hipGetLastError();
}
