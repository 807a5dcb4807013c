/*
 * TEST INFRASTRUCTURE ONLY - CPU oracle for the triangle rasterizer (face-index map + barycentric
 * weight map).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * PARITY UNPINNED: the reference calls `nr.rasterize_face_index_map_and_weight_map` of the third-party
 * package `neural_renderer`, pinned at iPERDance/neural_renderer@e5f54f71a8941acf372514eb92e289872f272653
 * (reference requirements/build.txt:3).  Its source is not in /root/reference and cannot be fetched, and
 * the reference's own tests hold no numeric fixture for it (tests/test_human_digitalizer/test_renders.py
 * only pushes images to Visdom).  This file therefore RESTATES the published algorithm of the upstream
 * neural_renderer "forward_face_index_map" kernels (per-face inverse matrix; per-pixel loop over faces in
 * index order; back-face cull; three edge-function inside test; weights clamped to [0,1] and renormalised;
 * perspective-correct depth 1/sum(w_k/z_k); keep nearest with near < z < far; first face wins ties) and is
 * anchored on the reference's call sites:
 *   - input conventions: renders/nmr.py:319-342 (y flipped before the call, un-flipped on f2pts after),
 *     renders/nmr.py:344-358 (UV-atlas call), fim == -1 is background (nmr.py:742), wim is (B,S,S,3);
 *   - consumer: renders/nmr.py:713-757 (cal_bc_transform uses wim as barycentric weights of the face's
 *     three vertices in face order) and models/flowcomposition.py:242 (result used as a grid_sample grid,
 *     align_corners=False) => "identity-warp" property: pixel (r,c) has centre x=(2c+1-S)/S and, in the
 *     rasterizer's y-up input space, y=(S-1-2r)/S (row 0 is the top of the image);
 *   - back-face rule pinned by in-repo data: every UV-atlas triangle of mapper_fim_enc.txt is kept.
 * All arithmetic is IEEE fp32 in the order written (build with -ffp-contract=off); the HIP kernel
 * evaluates the same expressions in the same order so face ids can be compared bit-exactly.
 */
#include <stdint.h>
#include <string.h>

static inline float clamp01(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }

/* per-face setup: returns 0 when the face is culled (back-facing or degenerate) */
static int face_setup(const float* f, int S, float inv[9]) {
    /* back-face: keep iff (y2-y0)(x1-x0) >= (y1-y0)(x2-x0) */
    if ((f[7] - f[1]) * (f[3] - f[0]) < (f[4] - f[1]) * (f[6] - f[0])) return 0;
    float p[3][2];
    for (int n = 0; n < 3; ++n)
        for (int d = 0; d < 2; ++d) p[n][d] = 0.5f * (f[3 * n + d] * (float)S + (float)S - 1.0f);
    float den = p[2][0] * (p[0][1] - p[1][1]) + p[0][0] * (p[1][1] - p[2][1]) + p[1][0] * (p[2][1] - p[0][1]);
    float m[9] = {
        p[1][1] - p[2][1], p[2][0] - p[1][0], p[1][0] * p[2][1] - p[2][0] * p[1][1],
        p[2][1] - p[0][1], p[0][0] - p[2][0], p[2][0] * p[0][1] - p[0][0] * p[2][1],
        p[0][1] - p[1][1], p[1][0] - p[0][0], p[0][0] * p[1][1] - p[1][0] * p[0][1]};
    for (int k = 0; k < 9; ++k) inv[k] = m[k] / den;
    return 1;
}

/*
 * faces: (B, nf, 3, 3) fp32, per vertex (x, y, z) in the rasterizer's input space (y up, as the reference
 *        passes them after `proj_verts[:, :, 1] *= -1` and look_at).
 * fim:   (B, S, S) int32, -1 = background.   wim: (B, S, S, 3) fp32 (zeros on background).
 */
void lwg_oracle_rasterize_fim_wim(const float* faces, int B, int nf, int S, float near, float far,
                                  int32_t* fim, float* wim) {
    float* inv = (float*)__builtin_malloc((size_t)nf * 9 * sizeof(float));
    unsigned char* keep = (unsigned char*)__builtin_malloc((size_t)nf);
    for (int b = 0; b < B; ++b) {
        const float* fb = faces + (size_t)b * nf * 9;
        for (int i = 0; i < nf; ++i) keep[i] = (unsigned char)face_setup(fb + (size_t)i * 9, S, inv + (size_t)i * 9);
#pragma omp parallel for schedule(dynamic, 4)
        for (int r = 0; r < S; ++r) {
            const int yi = S - 1 - r; /* vertical flip: row 0 = top = largest y */
            const float yp = (float)((2.0 * yi + 1 - S) / S);
            for (int xi = 0; xi < S; ++xi) {
                const float xp = (float)((2.0 * xi + 1 - S) / S);
                float zmin = far;
                int best = -1;
                float wb[3] = {0.f, 0.f, 0.f};
                for (int i = 0; i < nf; ++i) {
                    if (!keep[i]) continue;
                    const float* f = fb + (size_t)i * 9;
                    if (((yp - f[1]) * (f[3] - f[0]) < (xp - f[0]) * (f[4] - f[1])) ||
                        ((yp - f[4]) * (f[6] - f[3]) < (xp - f[3]) * (f[7] - f[4])) ||
                        ((yp - f[7]) * (f[0] - f[6]) < (xp - f[6]) * (f[1] - f[7])))
                        continue;
                    const float* m = inv + (size_t)i * 9;
                    float w0 = m[0] * (float)xi + m[1] * (float)yi + m[2];
                    float w1 = m[3] * (float)xi + m[4] * (float)yi + m[5];
                    float w2 = m[6] * (float)xi + m[7] * (float)yi + m[8];
                    w0 = clamp01(w0); w1 = clamp01(w1); w2 = clamp01(w2);
                    const float ws = w0 + w1 + w2;
                    w0 = w0 / ws; w1 = w1 / ws; w2 = w2 / ws;
                    const float zp = 1.0f / (w0 / f[2] + w1 / f[5] + w2 / f[8]);
                    if (zp <= near || far <= zp) continue;
                    if (zp < zmin) { zmin = zp; best = i; wb[0] = w0; wb[1] = w1; wb[2] = w2; }
                }
                const size_t o = ((size_t)b * S + r) * S + xi;
                fim[o] = best;
                wim[3 * o + 0] = wb[0]; wim[3 * o + 1] = wb[1]; wim[3 * o + 2] = wb[2];
            }
        }
    }
    __builtin_free(inv);
    __builtin_free(keep);
}
