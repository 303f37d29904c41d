// project.hip -- per-Gaussian 3D->2D projection (+ RaDe-GS ray-distance plane and normal),
// forward and backward, and the spherical-harmonics colour kernels.  gfx950 only.
//
// Replaces gsplat-rade's fully_fused_projection / spherical_harmonics as called at
// /root/reference/collab_splats/models/rade_gs_model.py:373-394 and rade_features_model.py:430-434
// (SURVEY.md section 8 rows a2.1, a2.2).  Math: SURVEY.md Appendix B.
//
// This file is compiled with -ffp-contract=off and the forward is written in one fixed IEEE
// operation order, so radii, tile rectangles and depth bits -- everything that feeds the
// integer binning stage -- are reproducible bit for bit by an independent fp32 implementation
// that uses the same order.  These kernels are streaming and HBM-bound (about 44 B read + 60 B
// written per Gaussian forward); arithmetic is irrelevant to their run time.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "misplat.h"
#include "sh_eval.h"
#include "internal.h"

namespace {

__device__ __forceinline__ float det_log(float x) {
    // natural log from plain mul/add/div only (deterministic): x = m * 2^e, m in (0.707, 1.414]
    uint32_t u = __float_as_uint(x);
    int e = (int)((u >> 23) & 0xffu) - 127;
    float m = __uint_as_float((u & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
    float f = m - 1.0f;
    float s = f / (2.0f + f);
    float s2 = s * s;
    float p = 0.11111111f;
    p = p * s2 + 0.14285715f;
    p = p * s2 + 0.2f;
    p = p * s2 + 0.33333334f;
    p = p * s2 + 1.0f;
    return (float)e * 0.69314718f + 2.0f * s * p;
}

struct ProjState {
    float Rc[9], qn[4], qnorm;
    float mu[3], u, v, tx, ty, limx, limy;
    int clampx, clampy;
    float cov[9];
    float J00, J02, J11, J12;
    float a0, b0, c0, det0, a, b, c, det, comp;
    float w[3], p[3], m[3], mnorm, nhat[3], ell, nh;
    int plane_ok, kmin;
};

struct Cam {
    float Rw[9], t[3], fx, fy, cx, cy;
};

__device__ __forceinline__ Cam load_cam(const float* __restrict__ V, const float* __restrict__ K) {
    Cam c;
    c.Rw[0] = V[0]; c.Rw[1] = V[1]; c.Rw[2] = V[2];
    c.Rw[3] = V[4]; c.Rw[4] = V[5]; c.Rw[5] = V[6];
    c.Rw[6] = V[8]; c.Rw[7] = V[9]; c.Rw[8] = V[10];
    c.t[0] = V[3]; c.t[1] = V[7]; c.t[2] = V[11];
    c.fx = K[0]; c.fy = K[4]; c.cx = K[2]; c.cy = K[5];
    return c;
}

__device__ __forceinline__ bool project_one(const float* mean, const float* quat, const float* scale,
                                            const Cam& cam, const misplat_params& P, ProjState& S) {
    float n = sqrtf(quat[0] * quat[0] + quat[1] * quat[1] + quat[2] * quat[2] + quat[3] * quat[3]);
    float r = quat[0] / n, x = quat[1] / n, y = quat[2] / n, z = quat[3] / n;
    S.qn[0] = r; S.qn[1] = x; S.qn[2] = y; S.qn[3] = z; S.qnorm = n;
    float Rg[9];
    Rg[0] = 1.0f - 2.0f * (y * y + z * z); Rg[1] = 2.0f * (x * y - r * z); Rg[2] = 2.0f * (x * z + r * y);
    Rg[3] = 2.0f * (x * y + r * z); Rg[4] = 1.0f - 2.0f * (x * x + z * z); Rg[5] = 2.0f * (y * z - r * x);
    Rg[6] = 2.0f * (x * z - r * y); Rg[7] = 2.0f * (y * z + r * x); Rg[8] = 1.0f - 2.0f * (x * x + y * y);
    const float* Rw = cam.Rw;
#pragma unroll
    for (int i = 0; i < 3; i++)
        S.mu[i] = Rw[i * 3 + 0] * mean[0] + Rw[i * 3 + 1] * mean[1] + Rw[i * 3 + 2] * mean[2] + cam.t[i];
    float zc = S.mu[2];
    if (!(zc >= P.near_plane) || !(zc <= P.far_plane)) return false;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            S.Rc[i * 3 + j] = Rw[i * 3 + 0] * Rg[0 * 3 + j] + Rw[i * 3 + 1] * Rg[1 * 3 + j] + Rw[i * 3 + 2] * Rg[2 * 3 + j];
    float M[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) M[i * 3 + k] = S.Rc[i * 3 + k] * scale[k];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            S.cov[i * 3 + j] = M[i * 3 + 0] * M[j * 3 + 0] + M[i * 3 + 1] * M[j * 3 + 1] + M[i * 3 + 2] * M[j * 3 + 2];
    float rz = 1.0f / zc;
    S.u = S.mu[0] * rz; S.v = S.mu[1] * rz;
    float tanx = 0.5f * (float)P.width / cam.fx, tany = 0.5f * (float)P.height / cam.fy;
    float lxp = ((float)P.width - cam.cx) / cam.fx + P.jacobian_margin * tanx;
    float lxn = cam.cx / cam.fx + P.jacobian_margin * tanx;
    float lyp = ((float)P.height - cam.cy) / cam.fy + P.jacobian_margin * tany;
    float lyn = cam.cy / cam.fy + P.jacobian_margin * tany;
    float uc = S.u, vc = S.v;
    S.clampx = 0; S.clampy = 0;
    if (uc > lxp) { uc = lxp; S.clampx = 1; } else if (uc < -lxn) { uc = -lxn; S.clampx = 1; }
    if (vc > lyp) { vc = lyp; S.clampy = 1; } else if (vc < -lyn) { vc = -lyn; S.clampy = 1; }
    S.limx = uc; S.limy = vc;
    S.tx = zc * uc; S.ty = zc * vc;
    float rz2 = rz * rz;
    S.J00 = cam.fx * rz; S.J11 = cam.fy * rz;
    S.J02 = -cam.fx * S.tx * rz2; S.J12 = -cam.fy * S.ty * rz2;
    const float* c3 = S.cov;
    float t00 = S.J00 * c3[0] + S.J02 * c3[6], t01 = S.J00 * c3[1] + S.J02 * c3[7], t02 = S.J00 * c3[2] + S.J02 * c3[8];
    float t11 = S.J11 * c3[4] + S.J12 * c3[7], t12 = S.J11 * c3[5] + S.J12 * c3[8];
    S.a0 = t00 * S.J00 + t02 * S.J02;
    S.b0 = t01 * S.J11 + t02 * S.J12;
    S.c0 = t11 * S.J11 + t12 * S.J12;
    S.det0 = S.a0 * S.c0 - S.b0 * S.b0;
    S.a = S.a0 + P.eps2d; S.c = S.c0 + P.eps2d; S.b = S.b0;
    S.det = S.a * S.c - S.b * S.b;
    if (!(S.det > 0.0f)) return false;
    float ratio = S.det0 / S.det;
    S.comp = sqrtf(ratio > 0.0f ? ratio : 0.0f);
    return true;
}

__device__ __forceinline__ void rade_extras(const float* scale, const Cam& cam, const misplat_params& P,
                                            ProjState& S, float& ray_t, float* ray_plane, float* normal) {
    int kmin = 0;
    if (scale[1] < scale[kmin]) kmin = 1;
    if (scale[2] < scale[kmin]) kmin = 2;
    S.kmin = kmin;
    float smin = kmin == 0 ? scale[0] : (kmin == 1 ? scale[1] : scale[2]);
#pragma unroll
    for (int k = 0; k < 3; k++) { float q = smin / scale[k]; S.w[k] = q * q; }
#pragma unroll
    for (int k = 0; k < 3; k++)
        S.p[k] = S.Rc[0 * 3 + k] * S.mu[0] + S.Rc[1 * 3 + k] * S.mu[1] + S.Rc[2 * 3 + k] * S.mu[2];
    float r0 = S.w[0] * S.p[0], r1 = S.w[1] * S.p[1], r2 = S.w[2] * S.p[2];
#pragma unroll
    for (int i = 0; i < 3; i++) S.m[i] = S.Rc[i * 3 + 0] * r0 + S.Rc[i * 3 + 1] * r1 + S.Rc[i * 3 + 2] * r2;
    S.mnorm = sqrtf(S.m[0] * S.m[0] + S.m[1] * S.m[1] + S.m[2] * S.m[2]);
    float zc = S.mu[2];
    S.ell = sqrtf(S.u * S.u + S.v * S.v + 1.0f);
    ray_t = zc * S.ell;
    S.plane_ok = 0;
    ray_plane[0] = ray_plane[1] = 0.0f;
    normal[0] = normal[1] = normal[2] = 0.0f;
    if (!(S.mnorm > 0.0f)) return;
#pragma unroll
    for (int i = 0; i < 3; i++) S.nhat[i] = S.m[i] / S.mnorm;
    S.nh = S.nhat[0] * S.u + S.nhat[1] * S.v + S.nhat[2];
    if (!(fabsf(S.nh) >= P.plane_eps) || !isfinite(S.nh)) return;
    S.plane_ok = 1;
    float A = zc * S.ell / S.nh;
    float dtdu = -A * S.nhat[0] + zc * S.u / S.ell;
    float dtdv = -A * S.nhat[1] + zc * S.v / S.ell;
    ray_plane[0] = dtdu / cam.fx; ray_plane[1] = dtdv / cam.fy;
    normal[0] = -S.nhat[0]; normal[1] = -S.nhat[1]; normal[2] = -S.nhat[2];
}

__global__ __launch_bounds__(256) void project_fwd_kernel(
    misplat_params P, const float* __restrict__ means, const float* __restrict__ quats,
    const float* __restrict__ scales, const float* __restrict__ opacities,
    const float* __restrict__ viewmats, const float* __restrict__ Ks, int32_t* __restrict__ radii,
    float* __restrict__ means2d, float* __restrict__ depths, float* __restrict__ conics,
    float* __restrict__ comps, float* __restrict__ ray_ts, float* __restrict__ ray_planes,
    float* __restrict__ normals) {
    const int64_t total = (int64_t)P.n_cams * P.n_gauss;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int cam_i = (int)(idx / P.n_gauss);
        const int g = (int)(idx - (int64_t)cam_i * P.n_gauss);
        const Cam cam = load_cam(viewmats + 16 * cam_i, Ks + 9 * cam_i);
        float mean[3] = {means[3 * g], means[3 * g + 1], means[3 * g + 2]};
        float quat[4] = {quats[4 * g], quats[4 * g + 1], quats[4 * g + 2], quats[4 * g + 3]};
        float scale[3] = {scales[3 * g], scales[3 * g + 1], scales[3 * g + 2]};
        int32_t rxi = 0, ryi = 0;
        float mx = 0.f, my = 0.f, dep = 0.f, cn0 = 0.f, cn1 = 0.f, cn2 = 0.f, comp = 0.f, rt = 0.f;
        float rp[2] = {0.f, 0.f}, nr[3] = {0.f, 0.f, 0.f};
        ProjState S;
        bool ok = project_one(mean, quat, scale, cam, P, S);
        if (ok) {
            float extend = P.radius_sigma;
            if (opacities != nullptr && P.opacity_aware_radius) {
                float o = opacities[g];
                if (P.antialiased) o = o * S.comp;
                if (o < P.alpha_min) ok = false;
                else {
                    float e2 = sqrtf(2.0f * det_log(o / P.alpha_min));
                    extend = extend < e2 ? extend : e2;
                }
            }
            if (ok) {
                float mid = 0.5f * (S.a + S.c);
                float disc = mid * mid - S.det;
                float v1 = mid + sqrtf(0.01f > disc ? 0.01f : disc);
                float sv1 = extend * sqrtf(v1);
                float ex = extend * sqrtf(S.a), ey = extend * sqrtf(S.c);
                float rx = ceilf(ex < sv1 ? ex : sv1);
                float ry = ceilf(ey < sv1 ? ey : sv1);
                float mxx = cam.fx * S.u + cam.cx, myy = cam.fy * S.v + cam.cy;
                if (rx <= P.radius_clip && ry <= P.radius_clip) ok = false;
                else if (mxx + rx <= 0.0f || mxx - rx >= (float)P.width || myy + ry <= 0.0f ||
                         myy - ry >= (float)P.height) ok = false;
                if (ok) {
                    rxi = (int32_t)rx; ryi = (int32_t)ry;
                    mx = mxx; my = myy; dep = S.mu[2];
                    cn0 = S.c / S.det; cn1 = -S.b / S.det; cn2 = S.a / S.det;
                    comp = S.comp;
                    rade_extras(scale, cam, P, S, rt, rp, nr);
                }
            }
        }
        radii[2 * idx] = rxi; radii[2 * idx + 1] = ryi;
        means2d[2 * idx] = mx; means2d[2 * idx + 1] = my;
        depths[idx] = dep;
        conics[3 * idx] = cn0; conics[3 * idx + 1] = cn1; conics[3 * idx + 2] = cn2;
        comps[idx] = comp; ray_ts[idx] = rt;
        ray_planes[2 * idx] = rp[0]; ray_planes[2 * idx + 1] = rp[1];
        normals[3 * idx] = nr[0]; normals[3 * idx + 1] = nr[1]; normals[3 * idx + 2] = nr[2];
    }
}

struct ProjGrads {
    float v_m2d[2], v_depth, v_conic[3], v_comp, v_rt, v_rp[2], v_nr[3];
    // m2d_sums: v_m2d holds (sum dx dL/dsigma, sum dy dL/dsigma) of the compositing backward's MSUM form (blend.hip): the mean2d
    // gradient is formed here from the recomputed conic and ray plane -- and left in v_m2d for the caller
    bool m2d_sums = false;
};

// Backward of project_one + rade_extras for ONE (camera, Gaussian); ACCUMULATES into o_m/o_q/o_s.
__device__ __forceinline__ void project_bwd_one(const float* mean, const float* quat, const float* sc,
                                                const Cam& cam, const misplat_params& P, ProjGrads& G,
                                                float* o_m, float* o_q, float* o_s) {
    ProjState S;
    if (!project_one(mean, quat, sc, cam, P, S)) {
        if (G.m2d_sums) { G.v_m2d[0] = 0.f; G.v_m2d[1] = 0.f; }
        return;
    }
    float rt, rp[2], nr[3];
    rade_extras(sc, cam, P, S, rt, rp, nr);
    if (G.m2d_sums) {
        // sigma = (a dx^2 + c dy^2) / 2 + b dx dy with the conic the forward packed (project_pack: c/det, -b/det, a/det), and the
        // pixel's depth  (rt - rp . d) / |ray|  falls with d: v_mean2d = conic (s0, s1) - rp v_rt
        const float ca = S.c / S.det, cb = -S.b / S.det, cc = S.a / S.det;
        const float s0 = G.v_m2d[0], s1 = G.v_m2d[1];
        G.v_m2d[0] = ca * s0 + cb * s1 - rp[0] * G.v_rt;
        G.v_m2d[1] = cb * s0 + cc * s1 - rp[1] * G.v_rt;
    }
    const float* Rw = cam.Rw;
    float z = S.mu[2], rz = 1.0f / z, rz2 = rz * rz;
    float v_mu[3] = {0.f, 0.f, 0.f}, v_u = 0.f, v_v = 0.f, v_s[3] = {0.f, 0.f, 0.f};
    float v_Rc[9];
#pragma unroll
    for (int i = 0; i < 9; i++) v_Rc[i] = 0.f;
    // 1. RaDe extras
    float v_rt = G.v_rt;
    float v_ell = v_rt * z;
    v_mu[2] += v_rt * S.ell;
    if (S.plane_ok) {
        float v_dtdu = G.v_rp[0] / cam.fx, v_dtdv = G.v_rp[1] / cam.fy;
        float A = z * S.ell / S.nh;
        float v_n[3] = {-G.v_nr[0], -G.v_nr[1], -G.v_nr[2]};
        v_n[0] += -A * v_dtdu; v_n[1] += -A * v_dtdv;
        float v_A = -(S.nhat[0] * v_dtdu + S.nhat[1] * v_dtdv);
        v_mu[2] += v_A * S.ell / S.nh;
        v_ell += v_A * z / S.nh;
        float v_nh = -v_A * A / S.nh;
        float uv = S.u * v_dtdu + S.v * v_dtdv;
        v_mu[2] += uv / S.ell;
        v_u += z * v_dtdu / S.ell; v_v += z * v_dtdv / S.ell;
        v_ell += -z * uv / (S.ell * S.ell);
        v_n[0] += v_nh * S.u; v_n[1] += v_nh * S.v; v_n[2] += v_nh;
        v_u += v_nh * S.nhat[0]; v_v += v_nh * S.nhat[1];
        float dotn = S.nhat[0] * v_n[0] + S.nhat[1] * v_n[1] + S.nhat[2] * v_n[2];
        float v_m[3];
#pragma unroll
        for (int i = 0; i < 3; i++) v_m[i] = (v_n[i] - S.nhat[i] * dotn) / S.mnorm;
        float r[3] = {S.w[0] * S.p[0], S.w[1] * S.p[1], S.w[2] * S.p[2]};
        float v_w[3], v_p[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float v_r = S.Rc[0 * 3 + k] * v_m[0] + S.Rc[1 * 3 + k] * v_m[1] + S.Rc[2 * 3 + k] * v_m[2];
            v_w[k] = v_r * S.p[k]; v_p[k] = v_r * S.w[k];
        }
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int k = 0; k < 3; k++) v_Rc[i * 3 + k] += v_m[i] * r[k] + S.mu[i] * v_p[k];
#pragma unroll
        for (int i = 0; i < 3; i++)
            v_mu[i] += S.Rc[i * 3 + 0] * v_p[0] + S.Rc[i * 3 + 1] * v_p[1] + S.Rc[i * 3 + 2] * v_p[2];
        float smin = S.kmin == 0 ? sc[0] : (S.kmin == 1 ? sc[1] : sc[2]);
        float v_smin = 0.f;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            v_s[k] += v_w[k] * (-2.0f * smin * smin / (sc[k] * sc[k] * sc[k]));
            v_smin += v_w[k] * 2.0f * smin / (sc[k] * sc[k]);
        }
#pragma unroll
        for (int k = 0; k < 3; k++) if (k == S.kmin) v_s[k] += v_smin;
    }
    v_u += v_ell * S.u / S.ell; v_v += v_ell * S.v / S.ell;
    // 2. conic / compensation -> cov2d
    float v0 = G.v_conic[0], v1 = G.v_conic[1], v2 = G.v_conic[2];
    float det = S.det, v_det = -(S.c * v0 - S.b * v1 + S.a * v2) / (det * det);
    float v_det0 = 0.f;
    if (S.det0 / det > 0.f && S.comp > 0.f) {
        float v_ratio = G.v_comp / (2.0f * S.comp);
        v_det0 = v_ratio / det;
        v_det += -v_ratio * S.det0 / (det * det);
    }
    float v_a = v2 / det + v_det * S.c, v_c = v0 / det + v_det * S.a, v_b = -v1 / det - 2.0f * S.b * v_det;
    float G00 = v_a + v_det0 * S.c0, G11 = v_c + v_det0 * S.a0, G01 = 0.5f * (v_b - 2.0f * S.b0 * v_det0);
    // 3. cov2d = J cov J^T
    float Jm[6] = {S.J00, 0.f, S.J02, 0.f, S.J11, S.J12};
    float GJ[6];
#pragma unroll
    for (int k = 0; k < 3; k++) { GJ[k] = G00 * Jm[k] + G01 * Jm[3 + k]; GJ[3 + k] = G01 * Jm[k] + G11 * Jm[3 + k]; }
    float v_cov[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) v_cov[i * 3 + j] = Jm[i] * GJ[j] + Jm[3 + i] * GJ[3 + j];
    float v_J[6];
#pragma unroll
    for (int r_ = 0; r_ < 2; r_++)
#pragma unroll
        for (int k = 0; k < 3; k++)
            v_J[r_ * 3 + k] = 2.0f * (GJ[r_ * 3 + 0] * S.cov[0 * 3 + k] + GJ[r_ * 3 + 1] * S.cov[1 * 3 + k] + GJ[r_ * 3 + 2] * S.cov[2 * 3 + k]);
    v_mu[2] += -v_J[0] * cam.fx * rz2 - v_J[4] * cam.fy * rz2;
    float v_tx = -v_J[2] * cam.fx * rz2, v_ty = -v_J[5] * cam.fy * rz2;
    v_mu[2] += 2.0f * v_J[2] * cam.fx * S.tx * rz2 * rz + 2.0f * v_J[5] * cam.fy * S.ty * rz2 * rz;
    if (S.clampx) v_mu[2] += v_tx * S.limx; else v_mu[0] += v_tx;
    if (S.clampy) v_mu[2] += v_ty * S.limy; else v_mu[1] += v_ty;
    // 4. mean2d, depth
    v_u += cam.fx * G.v_m2d[0]; v_v += cam.fy * G.v_m2d[1];
    v_mu[0] += v_u * rz; v_mu[1] += v_v * rz;
    v_mu[2] += -(v_u * S.u + v_v * S.v) * rz;
    v_mu[2] += G.v_depth;
    // 5. cov = M M^T, M = Rc diag(s)
    float M[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) M[i * 3 + k] = S.Rc[i * 3 + k] * sc[k];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < 3; j++) acc += (v_cov[i * 3 + j] + v_cov[j * 3 + i]) * M[j * 3 + k];
            v_Rc[i * 3 + k] += acc * sc[k];
            v_s[k] += acc * S.Rc[i * 3 + k];
        }
    float w_[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int k = 0; k < 3; k++)
            w_[i * 3 + k] = Rw[0 * 3 + i] * v_Rc[0 * 3 + k] + Rw[1 * 3 + i] * v_Rc[1 * 3 + k] + Rw[2 * 3 + i] * v_Rc[2 * 3 + k];
    float qr = S.qn[0], qx = S.qn[1], qy = S.qn[2], qz = S.qn[3];
    float vq[4];
    vq[0] = 2.0f * (-qz * w_[1] + qy * w_[2] + qz * w_[3] - qx * w_[5] - qy * w_[6] + qx * w_[7]);
    vq[1] = 2.0f * (qy * w_[1] + qz * w_[2] + qy * w_[3] - 2.0f * qx * w_[4] - qr * w_[5] + qz * w_[6] + qr * w_[7] - 2.0f * qx * w_[8]);
    vq[2] = 2.0f * (-2.0f * qy * w_[0] + qx * w_[1] + qr * w_[2] + qx * w_[3] + qz * w_[5] - qr * w_[6] + qz * w_[7] - 2.0f * qy * w_[8]);
    vq[3] = 2.0f * (-2.0f * qz * w_[0] - qr * w_[1] + qx * w_[2] + qr * w_[3] - 2.0f * qz * w_[4] + qy * w_[5] + qx * w_[6] + qy * w_[7]);
    float dq = S.qn[0] * vq[0] + S.qn[1] * vq[1] + S.qn[2] * vq[2] + S.qn[3] * vq[3];
#pragma unroll
    for (int k = 0; k < 4; k++) o_q[k] += (vq[k] - S.qn[k] * dq) / S.qnorm;
#pragma unroll
    for (int i = 0; i < 3; i++) o_m[i] += Rw[0 * 3 + i] * v_mu[0] + Rw[1 * 3 + i] * v_mu[1] + Rw[2 * 3 + i] * v_mu[2];
#pragma unroll
    for (int k = 0; k < 3; k++) o_s[k] += v_s[k];
}

// One thread per Gaussian, loop over cameras: gradients summed over cameras in a fixed order.
__global__ __launch_bounds__(256) void project_bwd_kernel(
    misplat_params P, const float* __restrict__ means, const float* __restrict__ quats,
    const float* __restrict__ scales, const float* __restrict__ viewmats,
    const float* __restrict__ Ks, const int32_t* __restrict__ radii,
    const float* __restrict__ v_means2d, const float* __restrict__ v_depths,
    const float* __restrict__ v_conics, const float* __restrict__ v_comps,
    const float* __restrict__ v_ray_ts, const float* __restrict__ v_ray_planes,
    const float* __restrict__ v_normals, float* __restrict__ v_means, float* __restrict__ v_quats,
    float* __restrict__ v_scales) {
    for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < P.n_gauss; g += gridDim.x * blockDim.x) {
        float mean[3] = {means[3 * g], means[3 * g + 1], means[3 * g + 2]};
        float quat[4] = {quats[4 * g], quats[4 * g + 1], quats[4 * g + 2], quats[4 * g + 3]};
        float sc[3] = {scales[3 * g], scales[3 * g + 1], scales[3 * g + 2]};
        float o_m[3] = {0.f, 0.f, 0.f}, o_q[4] = {0.f, 0.f, 0.f, 0.f}, o_s[3] = {0.f, 0.f, 0.f};
        for (int ci = 0; ci < P.n_cams; ci++) {
            const int64_t idx = (int64_t)ci * P.n_gauss + g;
            if (radii[2 * idx] <= 0 && radii[2 * idx + 1] <= 0) continue;
            const Cam cam = load_cam(viewmats + 16 * ci, Ks + 9 * ci);
            ProjGrads G;
            G.v_m2d[0] = v_means2d[2 * idx]; G.v_m2d[1] = v_means2d[2 * idx + 1];
            G.v_depth = v_depths[idx];
            G.v_conic[0] = v_conics[3 * idx]; G.v_conic[1] = v_conics[3 * idx + 1]; G.v_conic[2] = v_conics[3 * idx + 2];
            G.v_comp = v_comps[idx]; G.v_rt = v_ray_ts[idx];
            G.v_rp[0] = v_ray_planes[2 * idx]; G.v_rp[1] = v_ray_planes[2 * idx + 1];
            G.v_nr[0] = v_normals[3 * idx]; G.v_nr[1] = v_normals[3 * idx + 1]; G.v_nr[2] = v_normals[3 * idx + 2];
            project_bwd_one(mean, quat, sc, cam, P, G, o_m, o_q, o_s);
        }
#pragma unroll
        for (int k = 0; k < 3; k++) { v_means[3 * g + k] = o_m[k]; v_scales[3 * g + k] = o_s[k]; }
#pragma unroll
        for (int k = 0; k < 4; k++) v_quats[4 * g + k] = o_q[k];
    }
}

// ------------------------------------------------------------------ spherical harmonics

using misplat_sh::C0; using misplat_sh::C1; using misplat_sh::C2; using misplat_sh::C3; using misplat_sh::sh_eval;

template <bool GRAD>
__device__ __forceinline__ void sh_basis(int deg, float x, float y, float z, float* b, float* bx, float* by, float* bz) {
#pragma unroll
    for (int k = 0; k < 16; k++) { b[k] = 0.f; if (GRAD) { bx[k] = 0.f; by[k] = 0.f; bz[k] = 0.f; } }
    b[0] = C0;
    if (deg > 0) {
        b[1] = -C1 * y; b[2] = C1 * z; b[3] = -C1 * x;
        if (GRAD) { by[1] = -C1; bz[2] = C1; bx[3] = -C1; }
    }
    if (deg > 1) {
        float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        b[4] = C2[0] * xy; b[5] = C2[1] * yz; b[6] = C2[2] * (2.f * zz - xx - yy); b[7] = C2[3] * xz; b[8] = C2[4] * (xx - yy);
        if (GRAD) {
            bx[4] = C2[0] * y; by[4] = C2[0] * x;
            by[5] = C2[1] * z; bz[5] = C2[1] * y;
            bx[6] = -2.f * C2[2] * x; by[6] = -2.f * C2[2] * y; bz[6] = 4.f * C2[2] * z;
            bx[7] = C2[3] * z; bz[7] = C2[3] * x;
            bx[8] = 2.f * C2[4] * x; by[8] = -2.f * C2[4] * y;
        }
        if (deg > 2) {
            b[9] = C3[0] * y * (3.f * xx - yy); b[10] = C3[1] * xy * z; b[11] = C3[2] * y * (4.f * zz - xx - yy);
            b[12] = C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy); b[13] = C3[4] * x * (4.f * zz - xx - yy);
            b[14] = C3[5] * z * (xx - yy); b[15] = C3[6] * x * (xx - 3.f * yy);
            if (GRAD) {
                bx[9] = 6.f * C3[0] * xy; by[9] = C3[0] * (3.f * xx - 3.f * yy);
                bx[10] = C3[1] * yz; by[10] = C3[1] * xz; bz[10] = C3[1] * xy;
                bx[11] = -2.f * C3[2] * xy; by[11] = C3[2] * (4.f * zz - xx - 3.f * yy); bz[11] = 8.f * C3[2] * yz;
                bx[12] = -6.f * C3[3] * xz; by[12] = -6.f * C3[3] * yz; bz[12] = C3[3] * (6.f * zz - 3.f * xx - 3.f * yy);
                bx[13] = C3[4] * (4.f * zz - 3.f * xx - yy); by[13] = -2.f * C3[4] * xy; bz[13] = 8.f * C3[4] * xz;
                bx[14] = 2.f * C3[5] * xz; by[14] = -2.f * C3[5] * yz; bz[14] = C3[5] * (xx - yy);
                bx[15] = C3[6] * (3.f * xx - 3.f * yy); by[15] = -6.f * C3[6] * xy;
            }
        }
    }
}


__global__ __launch_bounds__(256) void sh_fwd_kernel(int n_gauss, int n_cams, int K, int deg,
                                                     const float* __restrict__ dirs,
                                                     const float* __restrict__ coeffs,
                                                     const int32_t* __restrict__ radii,
                                                     float* __restrict__ colors) {
    const int64_t total = (int64_t)n_gauss * n_cams;
    const int nb = (deg + 1) * (deg + 1);
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        float c0 = 0.f, c1 = 0.f, c2 = 0.f;
        if (radii == nullptr || radii[2 * idx] > 0 || radii[2 * idx + 1] > 0) {
            const int g = (int)(idx % n_gauss);
            float dx = dirs[3 * idx], dy = dirs[3 * idx + 1], dz = dirs[3 * idx + 2];
            float n = sqrtf(dx * dx + dy * dy + dz * dz);
            float inv = n > 0.f ? 1.0f / n : 0.f;
            float b[16];
            sh_basis<false>(deg, dx * inv, dy * inv, dz * inv, b, nullptr, nullptr, nullptr);
            const float* cf = coeffs + (size_t)g * K * 3;
#pragma unroll
            for (int k = 0; k < 16; k++)
                if (k < nb) { c0 += b[k] * cf[3 * k]; c1 += b[k] * cf[3 * k + 1]; c2 += b[k] * cf[3 * k + 2]; }
        }
        colors[3 * idx] = c0; colors[3 * idx + 1] = c1; colors[3 * idx + 2] = c2;
    }
}

__global__ __launch_bounds__(256) void sh_bwd_kernel(int n_gauss, int n_cams, int K, int deg,
                                                     const float* __restrict__ dirs,
                                                     const float* __restrict__ coeffs,
                                                     const int32_t* __restrict__ radii,
                                                     const float* __restrict__ v_colors,
                                                     float* __restrict__ v_coeffs,
                                                     float* __restrict__ v_dirs) {
    const int nb = (deg + 1) * (deg + 1);
    for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < n_gauss; g += gridDim.x * blockDim.x) {
        float acc[48];
#pragma unroll
        for (int k = 0; k < 48; k++) acc[k] = 0.f;
        const float* cf = coeffs + (size_t)g * K * 3;
        for (int ci = 0; ci < n_cams; ci++) {
            const int64_t idx = (int64_t)ci * n_gauss + g;
            float vd0 = 0.f, vd1 = 0.f, vd2 = 0.f, x = 0.f, y = 0.f, z = 0.f, inv = 0.f;
            if (radii == nullptr || radii[2 * idx] > 0 || radii[2 * idx + 1] > 0) {
                float dx = dirs[3 * idx], dy = dirs[3 * idx + 1], dz = dirs[3 * idx + 2];
                float n = sqrtf(dx * dx + dy * dy + dz * dz);
                inv = n > 0.f ? 1.0f / n : 0.f;
                x = dx * inv; y = dy * inv; z = dz * inv;
                float b[16], bx[16], by[16], bz[16];
                sh_basis<true>(deg, x, y, z, b, bx, by, bz);
                float vc0 = v_colors[3 * idx], vc1 = v_colors[3 * idx + 1], vc2 = v_colors[3 * idx + 2];
#pragma unroll
                for (int k = 0; k < 16; k++)
                    if (k < nb) {
                        acc[3 * k] += b[k] * vc0; acc[3 * k + 1] += b[k] * vc1; acc[3 * k + 2] += b[k] * vc2;
                        float s = cf[3 * k] * vc0 + cf[3 * k + 1] * vc1 + cf[3 * k + 2] * vc2;
                        vd0 += bx[k] * s; vd1 += by[k] * s; vd2 += bz[k] * s;
                    }
            }
            float dot = x * vd0 + y * vd1 + z * vd2;
            v_dirs[3 * idx] = (vd0 - x * dot) * inv;
            v_dirs[3 * idx + 1] = (vd1 - y * dot) * inv;
            v_dirs[3 * idx + 2] = (vd2 - z * dot) * inv;
        }
        float* o = v_coeffs + (size_t)g * K * 3;
        for (int k = 0; k < K; k++) {
            if (k < 16 && k < nb) { o[3 * k] = acc[3 * k]; o[3 * k + 1] = acc[3 * k + 1]; o[3 * k + 2] = acc[3 * k + 2]; }
            else { o[3 * k] = 0.f; o[3 * k + 1] = 0.f; o[3 * k + 2] = 0.f; }
        }
    }
}


// misplat_params.activations: the caller's parameters are log-scales (bit 0) / opacity logits (bit 1) and the kernels
// apply exp / sigmoid themselves (rade_gs_model.py:443-444 does it with two torch launches per direction).
__device__ __forceinline__ void apply_activations(const misplat_params& P, float (&scale)[3], float& opac) {
    // (expf / the sigmoid expression of torch's own device kernels: the same bits as torch.exp / torch.sigmoid in front of
    // the call, so the extension and the reference's call are interchangeable down to the integer stages)
    if (P.activations & 1) { scale[0] = expf(scale[0]); scale[1] = expf(scale[1]); scale[2] = expf(scale[2]); }
    if (P.activations & 2) opac = 1.0f / (1.0f + expf(-opac));
}

// ================================================================================================
// Fused per-Gaussian stages (the path rasterization() takes): projection writes the packed blend
// record directly, the colour kernel (SH + 0.5 clamp, or pass-through) fills its colour slots, and
// the two backward kernels consume the packed gradient rows -- no intermediate SoA round trips.
// Record layout (MISPLAT_REC floats): [0:2] mean2d [2:5] conic [5] opacity_eff [6] ray_t
// [7:9] ray_plane [9:12] normal [12:16] colour channels.
// ================================================================================================
constexpr int kRecPlane = 256 + 4;
__global__ __launch_bounds__(256) void project_pack_fwd_kernel(
    misplat_params P, const float* __restrict__ means, const float* __restrict__ quats,
    const float* __restrict__ scales, const float* __restrict__ opacities,
    const float* __restrict__ viewmats, const float* __restrict__ Ks, int32_t* __restrict__ radii,
    float* __restrict__ means2d, float* __restrict__ depths, float* __restrict__ comps,
    float4* __restrict__ grec, uint32_t* __restrict__ zero_words, int n_zero, float4* __restrict__ lazy_rows,
    float2* __restrict__ abs_rows, int clear_lazy_rows, const int32_t* __restrict__ order_table,
    int32_t* __restrict__ order_sel, int order_slots, int order_stride) {
    // scratch the NEXT kernels of the stream accumulate into (bucketing counters): cleared here, no memset launch
    // (a few thousand words -- cell counts, cursors, tile counts: spread over the first workgroups of the grid)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_zero; i += (int64_t)gridDim.x * blockDim.x)
        zero_words[i] = 0u;
    // View-keyed launch orders (misplat_params.unit_sel): one thread hashes the cameras of the call (FNV-1a over the
    // bit patterns of viewmats and Ks) and looks for a record with that tag in four consecutive slots; a miss takes the
    // first empty one of them, or evicts the home slot.  {slot, found, tag} go to order_sel for the compositing launches.
    if (order_sel && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        unsigned long long h = 1469598103934665603ull;
        for (int i = 0; i < 16 * P.n_cams; i++) h = (h ^ (unsigned long long)__float_as_uint(viewmats[i])) * 1099511628211ull;
        for (int i = 0; i < 9 * P.n_cams; i++) h = (h ^ (unsigned long long)__float_as_uint(Ks[i])) * 1099511628211ull;
        // (FNV leaves differences of high input bits -- a sign flip -- out of its low bits: cameras that mirror each other
        // would share a home slot; murmur3's finaliser spreads them)
        h ^= h >> 33; h *= 0xff51afd7ed558ccdull; h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ull; h ^= h >> 33;
        if (h == 0ull) h = 1ull;
        const int32_t lo = (int32_t)(uint32_t)h, hi = (int32_t)(uint32_t)(h >> 32);
        int found = -1, empty = -1;
        for (int i = 0; i < 4; i++) {
            const int sidx = (int)((h + (unsigned long long)i) % (unsigned long long)order_slots);
            const int32_t* hd = order_table + (size_t)sidx * order_stride;
            if (hd[2] != 0 && hd[0] == lo && hd[1] == hi) { found = sidx; break; }
            if (hd[2] == 0 && empty < 0) empty = sidx;
        }
        const int slot = found >= 0 ? found : (empty >= 0 ? empty : (int)(h % (unsigned long long)order_slots));
        order_sel[0] = slot; order_sel[1] = found >= 0 ? 1 : 0; order_sel[2] = lo; order_sel[3] = hi;
    }
    const int64_t total = (int64_t)P.n_cams * P.n_gauss;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total + blockDim.x - 1 - (total + blockDim.x - 1) % blockDim.x;
         idx += (int64_t)gridDim.x * blockDim.x) {
        if (lazy_rows) {
            // on-demand colours (misplat_blend_fwd_lazy): the colour slots start UNSET, and the gradient rows the
            // backward adds into are cleared here, as whole lines (the block's 256 rows are contiguous)
            if (clear_lazy_rows) {               // (0: the compositing forward clears them inside its own grid)
                const int64_t base = idx - threadIdx.x;
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int64_t e4 = 4 * base + threadIdx.x + u * (int64_t)blockDim.x;
                    if (e4 < 4 * total) lazy_rows[e4] = z;
                }
            }
        }
        // the 64-byte records leave through LDS as whole lines (one 16-byte slot per lane, consecutive lanes on consecutive
        // slots) instead of four 16-byte stores per lane at a stride of 64 bytes; planes of 256 + 4 slots: conflict-free
        // both ways.  Slot 3 (colours) is written only with on-demand colours, as UNSET.
        __shared__ float4 rec_sm[4 * kRecPlane];
        if (idx < total) {
        if (P.touched) P.touched[idx] = 0;
        if (abs_rows) abs_rows[idx] = make_float2(0.f, 0.f);      // the |mean2d gradient| rows the backward adds into
        const int cam_i = (int)(idx / P.n_gauss);
        const int g = (int)(idx - (int64_t)cam_i * P.n_gauss);
        const Cam cam = load_cam(viewmats + 16 * cam_i, Ks + 9 * cam_i);
        float mean[3] = {means[3 * g], means[3 * g + 1], means[3 * g + 2]};
        float quat[4] = {quats[4 * g], quats[4 * g + 1], quats[4 * g + 2], quats[4 * g + 3]};
        float scale[3] = {scales[3 * g], scales[3 * g + 1], scales[3 * g + 2]};
        float opac = opacities[g];
        apply_activations(P, scale, opac);
        int32_t rxi = 0, ryi = 0;
        float mx = 0.f, my = 0.f, dep = 0.f, cn0 = 0.f, cn1 = 0.f, cn2 = 0.f, comp = 0.f, rt = 0.f, oeff = 0.f;
        float rp[2] = {0.f, 0.f}, nr[3] = {0.f, 0.f, 0.f};
        ProjState S;
        bool ok = project_one(mean, quat, scale, cam, P, S);
        if (ok) {
            float extend = P.radius_sigma;
            float o = opac;
            if (P.antialiased) o = o * S.comp;
            if (P.opacity_aware_radius) {
                if (o < P.alpha_min) ok = false;
                else {
                    float e2 = sqrtf(2.0f * det_log(o / P.alpha_min));
                    extend = extend < e2 ? extend : e2;
                }
            }
            if (ok) {
                float mid = 0.5f * (S.a + S.c);
                float disc = mid * mid - S.det;
                float v1 = mid + sqrtf(0.01f > disc ? 0.01f : disc);
                float sv1 = extend * sqrtf(v1);
                float ex = extend * sqrtf(S.a), ey = extend * sqrtf(S.c);
                float rx = ceilf(ex < sv1 ? ex : sv1);
                float ry = ceilf(ey < sv1 ? ey : sv1);
                float mxx = cam.fx * S.u + cam.cx, myy = cam.fy * S.v + cam.cy;
                if (rx <= P.radius_clip && ry <= P.radius_clip) ok = false;
                else if (mxx + rx <= 0.0f || mxx - rx >= (float)P.width || myy + ry <= 0.0f ||
                         myy - ry >= (float)P.height) ok = false;
                if (ok) {
                    rxi = (int32_t)rx; ryi = (int32_t)ry;
                    mx = mxx; my = myy; dep = S.mu[2];
                    cn0 = S.c / S.det; cn1 = -S.b / S.det; cn2 = S.a / S.det;
                    comp = S.comp; oeff = o;
                    rade_extras(scale, cam, P, S, rt, rp, nr);
                }
            }
        }
        radii[2 * idx] = rxi; radii[2 * idx + 1] = ryi;
        means2d[2 * idx] = mx; means2d[2 * idx + 1] = my;
        depths[idx] = dep; comps[idx] = comp;
        const float u_ = __uint_as_float(0x7fc0deadu);
        rec_sm[threadIdx.x] = make_float4(mx, my, cn0, cn1);
        rec_sm[kRecPlane + threadIdx.x] = make_float4(cn2, oeff, rt, rp[0]);
        rec_sm[2 * kRecPlane + threadIdx.x] = make_float4(rp[1], nr[0], nr[1], nr[2]);
        rec_sm[3 * kRecPlane + threadIdx.x] = make_float4(u_, u_, u_, u_);
        }
        __syncthreads();
        {
            const int64_t base4 = 4 * (idx - threadIdx.x);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int j = threadIdx.x + u * 256;                   // slot j of the block = (row j / 4, plane j % 4)
                if (base4 + j < 4 * total && ((j & 3) != 3 || lazy_rows)) grec[base4 + j] = rec_sm[(j & 3) * kRecPlane + (j >> 2)];
            }
        }
        __syncthreads();
    }
}

// Colour slots of the record.  sh_degree >= 0: colour = max(SH(dir) + 0.5, 0) with
// dir = mean - camera_centre (rade_features_model.py:428-438); sh_degree < 0: the first
// min(D, 4) entries of colors[(cam,) g, D] are copied.  depth_channel: channel n_color = depth.
// SH coefficients ([N, K, 3], 12 K bytes per Gaussian) are staged through LDS in whole
// coalesced lines and read back at a stride of 3K+1 floats (conflict-free).
// AUX: the forward also leaves, per (camera, Gaussian), the 3x3 Jacobian d rgb / d dir of the clamped colour
// (rows of clamped channels zeroed) and the three clamp flags in sh_aux[idx][12]; the backward then takes
// v_dir = J^T v_rgb from those 48 bytes and never reads the 12 K bytes of coefficients again.
// Measured (1 M Gaussians, one camera, K = 16, forward with AUX; 99.7 us before this list): term-by-term evaluation
// (sh_eval: 245 -> 151 VGPRs, two -> three waves per SIMD) +-0; gradient rows and Jacobian rows written as whole lines
// instead of one strided row per lane -6 us; the lane's own radii / mean / depth requested before the staging barrier
// (one HBM round trip per block instead of two) -3 us: 90.4 us.  What the streams cost, by leaving one out at a time:
// the 192 MB of coefficients 55 us (3.5 TB/s), the 64 MB of cleared gradient rows 9 us, the 48 MB of Jacobians 11 us, and
// the 16 MB of colours 11 us -- 16 bytes into each 64-byte record that the projection kernel started, the one
// partial-line stream left (closing it means fusing this kernel into the projection).
// KC: compile-time K (16 = degree-3 storage: the row length 48 and the LDS stride 49 become constants, the
// coefficients move as 16-byte vectors, and rows of Gaussians culled in every camera are never read) or 0 (any K).
// SPLIT: 0 = [N,K,3] coefficients, 1 = features_dc + features_rest, -1 = decided at run time (the staging variants
// differ a lot in registers: compiled together the kernel is allocated for the hungriest of them).
template <bool BWD, int BLOCK, bool MULTI, bool AUX = false, int KC = 0, int SPLIT = -1>
#ifndef MISPLAT_SH_FWD_WAVES
#define MISPLAT_SH_FWD_WAVES 1         /* waves per SIMD the K = 16 forward is compiled for (1: no constraint) */
#endif
__global__ __launch_bounds__(BLOCK, (!BWD && KC == 16) ? MISPLAT_SH_FWD_WAVES : ((BWD && !AUX && !MULTI && KC == 16) ? 4 : 1)) void color_sh_kernel(
    misplat_params P, int K, int deg, int depth_channel, const float* __restrict__ means,
    const float* __restrict__ viewmats, const float* __restrict__ coeffs, const float* __restrict__ coeffs_rest,
    const int32_t* __restrict__ radii, const float* __restrict__ depths, float* __restrict__ grec,
    const float* __restrict__ v_grec, float* __restrict__ v_coeffs, float* __restrict__ v_coeffs_rest,
    float* __restrict__ v_means_dir, float* __restrict__ sh_aux = nullptr, float4* __restrict__ zero_rows = nullptr,
    const float* __restrict__ slot3 = nullptr, int slot3_stride = 0) {
    // slot3 (forward, or NULL): the record's fourth colour slot takes slot3[g * slot3_stride] -- the first of the feature
    // channels the features model composites behind its SH colours (rade_features_model.py:441) -- instead of the depth / 0
    // coeffs_rest == NULL: coeffs is [N,K,3]; else coeffs is features_dc [N,3] and coeffs_rest is
    // features_rest [N,K-1,3] (the two parameter tensors of rade_gs_model.py:119-120, read in place
    // instead of through the per-step torch.cat of :128-130)
    extern __shared__ float lds[];
    __shared__ uint8_t s_vis[BLOCK];
    if (SPLIT == 0) { coeffs_rest = nullptr; v_coeffs_rest = nullptr; }
    if (KC) K = KC;
    const int row = 3 * K, stride = row + 1;
    const int nb = (deg + 1) * (deg + 1);
    const int n_blocks = (P.n_gauss + BLOCK - 1) / BLOCK;
    for (int blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const int g0 = blk * BLOCK;
        const int cnt = min(BLOCK, P.n_gauss - g0);
        __syncthreads();
        // forward: what the lane needs about its own row for camera 0 is requested NOW, together with the coefficient
        // lines below -- behind the staging barrier it would be a second dependent HBM round trip per block
        int pr0 = 0, pr1 = 0;
        float pm0 = 0.f, pm1 = 0.f, pm2 = 0.f, pdep = 0.f;
        if (!BWD && (int)threadIdx.x < cnt) {
            const int gp = g0 + threadIdx.x;
            pr0 = radii[2 * (int64_t)gp]; pr1 = radii[2 * (int64_t)gp + 1];
            pm0 = means[3 * gp]; pm1 = means[3 * gp + 1]; pm2 = means[3 * gp + 2];
            if (slot3) pdep = slot3[(size_t)gp * slot3_stride];
            else if (depth_channel) pdep = depths[gp];
        }
        constexpr bool USE_VIS = KC && BWD && !AUX;
        float pv0 = 0.f, pv1 = 0.f, pv2 = 0.f;      // (USE_VIS) camera 0: the row's colour gradient, fetched with the radii
        if (USE_VIS && (int)threadIdx.x < cnt) {
            const int gp = g0 + threadIdx.x;
            pr0 = radii[2 * (int64_t)gp]; pr1 = radii[2 * (int64_t)gp + 1];
            pm0 = means[3 * gp]; pm1 = means[3 * gp + 1]; pm2 = means[3 * gp + 2];
            const float* vg = v_grec + (size_t)gp * MISPLAT_REC + 12;
            pv0 = vg[0]; pv1 = vg[1]; pv2 = vg[2];
        }
        if (USE_VIS) {
            // visible in any camera?  (rows that are not are not staged)  Only where the staged rows are consumed late
            // enough: in the forward, waiting for the radii before the first coefficient load costs more (a dependent
            // round trip per block) than the culled rows' 192 bytes save.
            bool v = false;
            if ((int)threadIdx.x < cnt)
                {   // ... and with a colour gradient: a row the compositing never reached needs no coefficients
                    v = (pr0 > 0 || pr1 > 0) && (pv0 != 0.f || pv1 != 0.f || pv2 != 0.f);
                    for (int ci = 1; ci < P.n_cams; ci++) {
                        const int64_t idx = (int64_t)ci * P.n_gauss + g0 + threadIdx.x;
                        const float* vg = v_grec + (size_t)idx * MISPLAT_REC + 12;
                        v |= (radii[2 * idx] > 0 || radii[2 * idx + 1] > 0) && (vg[0] != 0.f || vg[1] != 0.f || vg[2] != 0.f);
                    }
                }
            s_vis[threadIdx.x] = v ? 1 : 0;
            __syncthreads();
        }
        if (BWD && AUX) {
            // nothing to stage: the LDS rows only carry the gradient back out
        } else if (KC && SPLIT != 1 && coeffs_rest == nullptr) {
            // [N, 16, 3]: a row is 12 aligned float4s; the whole block (64 rows = 12 vectors per lane) is requested
            // in ONE round of loads -- three dependent rounds of four cost two more HBM round trips per block
            const float4* src4 = reinterpret_cast<const float4*>(coeffs + (size_t)g0 * row);
            constexpr int V4 = KC ? 3 * KC / 4 : 1;
            // forward: all 12 vectors of a lane in flight at once; backward (few rows are staged at all: only those with
            // a colour gradient): 4 at a time, the kernel's registers set its occupancy for the dense stores that follow
            constexpr int CH = BWD ? 4 : V4;
#pragma unroll
            for (int u0 = 0; u0 < V4; u0 += CH) {
                float4 v[CH];
#pragma unroll
                for (int u = 0; u < CH; u++) {
                    const int e4 = threadIdx.x + (u0 + u) * BLOCK;
                    const int tt = e4 / V4;
                    if (e4 < cnt * V4 && (!USE_VIS || s_vis[tt])) v[u] = src4[e4];
                }
#pragma unroll
                for (int u = 0; u < CH; u++) {
                    const int e4 = threadIdx.x + (u0 + u) * BLOCK;
                    const int tt = e4 / V4, kk = 4 * (e4 - tt * V4);
                    if (e4 < cnt * V4 && (!USE_VIS || s_vis[tt])) {
                        float* d = lds + tt * stride + kk;
                        d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
                    }
                }
            }
        } else if (KC && SPLIT != 0) {
            // features_dc [N, 3] + features_rest [N, 15, 3]: rows of 3 and 45 floats; vectors may straddle two rows
            const float* src_dc = coeffs + (size_t)g0 * 3;
            for (int e = threadIdx.x; e < cnt * 3; e += BLOCK) lds[(e / 3) * stride + (e % 3)] = src_dc[e];
            constexpr int RR = KC ? 3 * KC - 3 : 1;
            const float4* src4 = reinterpret_cast<const float4*>(coeffs_rest + (size_t)g0 * RR);
            const int n4 = (cnt * RR) / 4;                        // cnt * 45 is a multiple of 4 for full blocks
            for (int e0 = threadIdx.x; e0 < n4; e0 += 4 * BLOCK) {
                float4 v[4];
                bool on[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int e4 = e0 + u * BLOCK;
                    on[u] = false;
                    if (e4 < n4) {
                        const int ta = (4 * e4) / RR, tb = (4 * e4 + 3) / RR;
                        on[u] = !USE_VIS || s_vis[ta] || s_vis[tb];
                        if (on[u]) v[u] = src4[e4];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (on[u]) {
                        const int e = 4 * (e0 + u * BLOCK);
                        const float vv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const int t = (e + j) / RR, k = (e + j) - t * RR;
                            lds[t * stride + 3 + k] = vv[j];
                        }
                    }
            }
            for (int e = 4 * n4 + threadIdx.x; e < cnt * RR; e += BLOCK) {      // tail of a partial last block
                const int t = e / RR, k = e - t * RR;
                lds[t * stride + 3 + k] = coeffs_rest[(size_t)g0 * RR + e];
            }
        } else if (coeffs_rest == nullptr) {
            const float* src = coeffs + (size_t)g0 * row;
            for (int e = threadIdx.x; e < cnt * row; e += BLOCK) {
                const int t = e / row, k = e - t * row;
                lds[t * stride + k] = src[e];
            }
        } else {
            const float* src_dc = coeffs + (size_t)g0 * 3;
            const float* src_rest = coeffs_rest + (size_t)g0 * (row - 3);
            for (int e = threadIdx.x; e < cnt * 3; e += BLOCK) lds[(e / 3) * stride + (e % 3)] = src_dc[e];
            const int rrow = row - 3;
            for (int e = threadIdx.x; e < cnt * rrow; e += BLOCK) {
                const int t = e / rrow, k = e - t * rrow;
                lds[t * stride + 3 + k] = src_rest[e];
            }
        }
        __syncthreads();
        const int t = threadIdx.x;
        const int g = g0 + t;
        float acc[MULTI ? 48 : 1];           // cross-camera accumulators (one camera: written in place)
        float vmd[3] = {0.f, 0.f, 0.f};
        if (BWD && MULTI) {
#pragma unroll
            for (int k = 0; k < 48; k++) acc[k] = 0.f;
        }
        bool wrote = false;
        if (t < cnt) {
            float* cf = lds + t * stride;
            for (int ci = 0; ci < P.n_cams; ci++) {
                const int64_t idx = (int64_t)ci * P.n_gauss + g;
                const bool first = (!BWD || USE_VIS) && ci == 0;
                bool vis = first ? (pr0 > 0 || pr1 > 0) : (radii[2 * idx] > 0 || radii[2 * idx + 1] > 0);
                float vgc0 = pv0, vgc1 = pv1, vgc2 = pv2;             // (USE_VIS) this camera's colour gradient
                if (USE_VIS) {                       // (its coefficients were staged only if some camera has a gradient for it)
                    if (ci > 0) { const float* vg = v_grec + (size_t)idx * MISPLAT_REC + 12; vgc0 = vg[0]; vgc1 = vg[1]; vgc2 = vg[2]; }
                    vis = vis && s_vis[t] && (vgc0 != 0.f || vgc1 != 0.f || vgc2 != 0.f);
                }
                float c0 = 0.f, c1 = 0.f, c2 = 0.f;
                if (vis) {
                    const float* V = viewmats + 16 * ci;
                    // camera centre = -R^T t
                    const float ccx = -(V[0] * V[3] + V[4] * V[7] + V[8] * V[11]);
                    const float ccy = -(V[1] * V[3] + V[5] * V[7] + V[9] * V[11]);
                    const float ccz = -(V[2] * V[3] + V[6] * V[7] + V[10] * V[11]);
                    const bool pre = !BWD || USE_VIS;
                    const float mx_ = pre ? pm0 : means[3 * g], my_ = pre ? pm1 : means[3 * g + 1], mz_ = pre ? pm2 : means[3 * g + 2];
                    const float dx = mx_ - ccx, dy = my_ - ccy, dz = mz_ - ccz;
                    const float n = sqrtf(dx * dx + dy * dy + dz * dz);
                    const float inv = n > 0.f ? 1.0f / n : 0.f;
                    const float x = dx * inv, y = dy * inv, z = dz * inv;
                    float b[16], bx[16], by[16], bz[16];
                    float J[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};         // J[ch][axis]
                    if (!BWD || !AUX) sh_eval<(!BWD && AUX)>(deg, x, y, z, cf, c0, c1, c2, J);
                    else sh_basis<false>(deg, x, y, z, b, bx, by, bz);
                    if (!BWD && AUX) {
                        const float m0 = (c0 + 0.5f > 0.f) ? 1.f : 0.f, m1 = (c1 + 0.5f > 0.f) ? 1.f : 0.f;
                        const float m2 = (c2 + 0.5f > 0.f) ? 1.f : 0.f;
                        if (KC && P.n_cams == 1) {
                            // the coefficients of this row are dead: its first 12 LDS slots carry the Jacobian out, so that
                            // the block's 64 x 48 bytes leave as whole contiguous lines below
                            cf[0] = J[0] * m0; cf[1] = J[1] * m0; cf[2] = J[2] * m0; cf[3] = J[3] * m1;
                            cf[4] = J[4] * m1; cf[5] = J[5] * m1; cf[6] = J[6] * m2; cf[7] = J[7] * m2;
                            cf[8] = J[8] * m2; cf[9] = m0; cf[10] = m1; cf[11] = m2;
                        } else {
                            float4* ax = reinterpret_cast<float4*>(sh_aux + (size_t)idx * 12);
                            ax[0] = make_float4(J[0] * m0, J[1] * m0, J[2] * m0, J[3] * m1);
                            ax[1] = make_float4(J[4] * m1, J[5] * m1, J[6] * m2, J[7] * m2);
                            ax[2] = make_float4(J[8] * m2, m0, m1, m2);
                        }
                    }
                    if (BWD && AUX) {
                        const float* vg = v_grec + (size_t)idx * MISPLAT_REC + 12;
                        const float g0v = vg[0], g1v = vg[1], g2v = vg[2];
                        // a Gaussian the compositing never reached has a zero colour gradient: its gradient row is
                        // zeros (written below) and its 48 bytes of Jacobian are not fetched
                        if (g0v == 0.f && g1v == 0.f && g2v == 0.f) continue;
                        const float4* ax = reinterpret_cast<const float4*>(sh_aux + (size_t)idx * 12);
                        const float4 a0 = ax[0], a1 = ax[1], a2 = ax[2];
                        const float vc0 = a2.y * g0v, vc1 = a2.z * g1v, vc2 = a2.w * g2v;     // clamp flags are 0 / 1
                        const float vd0 = a0.x * g0v + a0.w * g1v + a1.z * g2v;
                        const float vd1 = a0.y * g0v + a1.x * g1v + a1.w * g2v;
                        const float vd2 = a0.z * g0v + a1.y * g1v + a2.x * g2v;
#pragma unroll
                        for (int k = 0; k < 16; k++)
                            if (k < nb) {
                                if (MULTI) {
                                    acc[3 * k] += b[k] * vc0; acc[3 * k + 1] += b[k] * vc1; acc[3 * k + 2] += b[k] * vc2;
                                } else {
                                    cf[3 * k] = b[k] * vc0; cf[3 * k + 1] = b[k] * vc1; cf[3 * k + 2] = b[k] * vc2;
                                }
                            }
                        wrote = true;
                        const float dot = x * vd0 + y * vd1 + z * vd2;
                        vmd[0] += (vd0 - x * dot) * inv; vmd[1] += (vd1 - y * dot) * inv; vmd[2] += (vd2 - z * dot) * inv;
                    }
                    if (BWD && !AUX) {
                        const float* vg = v_grec + (size_t)idx * MISPLAT_REC + 12;
                        const float vc0 = (c0 + 0.5f > 0.f) ? (USE_VIS ? vgc0 : vg[0]) : 0.f;
                        const float vc1 = (c1 + 0.5f > 0.f) ? (USE_VIS ? vgc1 : vg[1]) : 0.f;
                        const float vc2 = (c2 + 0.5f > 0.f) ? (USE_VIS ? vgc2 : vg[2]) : 0.f;
                        float vd0 = 0.f, vd1 = 0.f, vd2 = 0.f;
                        // (one camera: the coefficient is dead once its term is done, its slot takes the gradient)
                        misplat_sh::sh_grad<MULTI>(deg, x, y, z, cf, vc0, vc1, vc2, vd0, vd1, vd2, acc);
                        wrote = true;
                        const float dot = x * vd0 + y * vd1 + z * vd2;
                        vmd[0] += (vd0 - x * dot) * inv; vmd[1] += (vd1 - y * dot) * inv; vmd[2] += (vd2 - z * dot) * inv;
                    }
                }
                if (!BWD) {
                    float* o = grec + (size_t)idx * MISPLAT_REC + 12;
                    float4 c = make_float4(fmaxf(c0 + 0.5f, 0.f), fmaxf(c1 + 0.5f, 0.f), fmaxf(c2 + 0.5f, 0.f),
                                           slot3 ? pdep : (depth_channel ? (first ? pdep : depths[idx]) : 0.f));
                    if (!vis) c = make_float4(0.f, 0.f, 0.f, 0.f);
                    *reinterpret_cast<float4*>(o) = c;
                    if (zero_rows && !KC) {          // the gradient row the backward's atomics will add into: no memset later
                        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                        zero_rows[4 * idx] = z; zero_rows[4 * idx + 1] = z; zero_rows[4 * idx + 2] = z; zero_rows[4 * idx + 3] = z;
                    }
                }
            }
        }
        if (!BWD && KC) {
            // K = 16 forward: the block's gradient rows (64 x 64 B) and Jacobian rows (64 x 48 B) are contiguous in
            // memory: written as whole lines, 16 bytes per lane and instruction, instead of one strided row per lane
            if (zero_rows) {
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int ci = 0; ci < P.n_cams; ci++) {
                    float4* zr = zero_rows + 4 * ((size_t)ci * P.n_gauss + g0);
                    for (int e4 = threadIdx.x; e4 < cnt * 4; e4 += BLOCK) zr[e4] = z;
                }
            }
            if (AUX && P.n_cams == 1) {
                __syncthreads();
                float4* ax4 = reinterpret_cast<float4*>(sh_aux + (size_t)g0 * 12);
                for (int e4 = threadIdx.x; e4 < cnt * 3; e4 += BLOCK) {
                    const int tt = e4 / 3, k = 4 * (e4 - 3 * tt);
                    const float* sp = lds + tt * stride + k;
                    ax4[e4] = make_float4(sp[0], sp[1], sp[2], sp[3]);
                }
            }
        }
        if (BWD) {
            // gradient rows back through LDS so the global stores are whole coalesced lines
            if (t < cnt) {
                float* cf = lds + t * stride;     // own row only: no barrier needed before writing it
                if (MULTI) {
#pragma unroll
                    for (int k = 0; k < 16; k++) {
                        if (k < K) {
                            const bool on = k < nb;
                            cf[3 * k] = on ? acc[3 * k] : 0.f; cf[3 * k + 1] = on ? acc[3 * k + 1] : 0.f; cf[3 * k + 2] = on ? acc[3 * k + 2] : 0.f;
                        }
                    }
                    for (int k = 16; k < K; k++) { cf[3 * k] = 0.f; cf[3 * k + 1] = 0.f; cf[3 * k + 2] = 0.f; }
                } else {
                    // rows of culled Gaussians and the coefficients above the active degree get zeros
                    for (int k = wrote ? nb : 0; k < K; k++) { cf[3 * k] = 0.f; cf[3 * k + 1] = 0.f; cf[3 * k + 2] = 0.f; }
                }
                v_means_dir[3 * g] = vmd[0]; v_means_dir[3 * g + 1] = vmd[1]; v_means_dir[3 * g + 2] = vmd[2];
            }
            __syncthreads();
            if (KC && SPLIT != 1 && v_coeffs_rest == nullptr) {
                float4* dst4 = reinterpret_cast<float4*>(v_coeffs + (size_t)g0 * row);
                constexpr int V4 = KC ? 3 * KC / 4 : 1;
                for (int e4 = threadIdx.x; e4 < cnt * V4; e4 += BLOCK) {
                    const int tt = e4 / V4, k = 4 * (e4 - tt * V4);
                    const float* sp = lds + tt * stride + k;
                    dst4[e4] = make_float4(sp[0], sp[1], sp[2], sp[3]);
                }
            } else if (KC && SPLIT != 0) {
                float* dst_dc = v_coeffs + (size_t)g0 * 3;
                for (int e = threadIdx.x; e < cnt * 3; e += BLOCK) dst_dc[e] = lds[(e / 3) * stride + (e % 3)];
                constexpr int RR = KC ? 3 * KC - 3 : 1;
                float4* dst4 = reinterpret_cast<float4*>(v_coeffs_rest + (size_t)g0 * RR);
                const int n4 = (cnt * RR) / 4;
                for (int e4 = threadIdx.x; e4 < n4; e4 += BLOCK) {
                    float vv[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int e = 4 * e4 + j, t = e / RR, k = e - t * RR;
                        vv[j] = lds[t * stride + 3 + k];
                    }
                    dst4[e4] = make_float4(vv[0], vv[1], vv[2], vv[3]);
                }
                for (int e = 4 * n4 + threadIdx.x; e < cnt * RR; e += BLOCK) {
                    const int t = e / RR, k = e - t * RR;
                    v_coeffs_rest[(size_t)g0 * RR + e] = lds[t * stride + 3 + k];
                }
            } else if (v_coeffs_rest == nullptr) {
                float* dst = v_coeffs + (size_t)g0 * row;
                for (int e = threadIdx.x; e < cnt * row; e += BLOCK) {
                    const int tt = e / row, k = e - tt * row;
                    dst[e] = lds[tt * stride + k];
                }
            } else {
                float* dst_dc = v_coeffs + (size_t)g0 * 3;
                float* dst_rest = v_coeffs_rest + (size_t)g0 * (row - 3);
                for (int e = threadIdx.x; e < cnt * 3; e += BLOCK) dst_dc[e] = lds[(e / 3) * stride + (e % 3)];
                const int rrow = row - 3;
                for (int e = threadIdx.x; e < cnt * rrow; e += BLOCK) {
                    const int tt = e / rrow, k = e - tt * rrow;
                    dst_rest[e] = lds[tt * stride + 3 + k];
                }
            }
        }
    }
}

// pass-through colours: colors [N, D] (per_cam = 0) or [C, N, D] (per_cam = 1), D <= 4 channels used
__global__ __launch_bounds__(256) void color_copy_kernel(misplat_params P, int D, int n_color, int per_cam,
                                                         int depth_channel, const float* __restrict__ colors,
                                                         const int32_t* __restrict__ radii,
                                                         const float* __restrict__ depths, float* __restrict__ grec,
                                                         float4* __restrict__ zero_rows) {
    const int64_t total = (int64_t)P.n_cams * P.n_gauss;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        if (zero_rows) {
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            zero_rows[4 * idx] = z; zero_rows[4 * idx + 1] = z; zero_rows[4 * idx + 2] = z; zero_rows[4 * idx + 3] = z;
        }
        const int64_t src = per_cam ? idx : idx % P.n_gauss;
        float c[4] = {0.f, 0.f, 0.f, 0.f};
        if (radii[2 * idx] > 0 || radii[2 * idx + 1] > 0) {
            for (int k = 0; k < n_color; k++) c[k] = colors[(size_t)src * D + k];
            if (depth_channel) c[n_color] = depths[idx];
        }
        *reinterpret_cast<float4*>(grec + (size_t)idx * MISPLAT_REC + 12) = make_float4(c[0], c[1], c[2], c[3]);
    }
}

__global__ __launch_bounds__(256) void color_copy_bwd_kernel(misplat_params P, int D, int n_color, int per_cam,
                                                             const int32_t* __restrict__ radii,
                                                             const float* __restrict__ v_grec,
                                                             float* __restrict__ v_colors) {
    const int64_t rows = per_cam ? (int64_t)P.n_cams * P.n_gauss : P.n_gauss;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x) {
        float c[4] = {0.f, 0.f, 0.f, 0.f};
        const int c_lo = per_cam ? (int)(r / P.n_gauss) : 0, c_hi = per_cam ? c_lo + 1 : P.n_cams;
        const int g = (int)(r % P.n_gauss);
        for (int ci = c_lo; ci < c_hi; ci++) {
            const int64_t idx = (int64_t)ci * P.n_gauss + g;
            if (radii[2 * idx] > 0 || radii[2 * idx + 1] > 0)
                for (int k = 0; k < n_color; k++) c[k] += v_grec[(size_t)idx * MISPLAT_REC + 12 + k];
        }
        for (int k = 0; k < D; k++) v_colors[(size_t)r * D + k] = k < n_color ? c[k] : 0.f;
    }
}

// N-D pass-through colours: channel c of [user colours ..., depth] goes to record slot 12+c for c < 4 and
// to featx[row][c-4] beyond (zero padded up to 4*nxq).
// N-D colours (a8): the first four of the D (+ depth) channels go to the record's colour slots, the rest to featx[row][4 nxq].
// One thread per 16-byte GROUP of a row's destination (group 0 = the record slots, groups 1.. = featx): consecutive threads
// read consecutive 16-byte pieces of the colour rows (a whole 64-byte row of 16 channels per four threads) instead of one
// thread walking its row channel by channel through 64-byte strides -- the first version took 230 us for 144 MB at 1 M
// Gaussians, 20x the streaming time.
// n_pre: the fused row's first n_pre channels come from elsewhere (3: the SH colours the colour kernel writes, together with
// channel 3 = colors[., 0], into the record): source channel j is fused channel n_pre + j, and group 0 is not written here.
// zero_grec / zero_featx (or NULL): the gradient rows [C*N,16] / [C*N,4 nxq] the compositing backward adds into are cleared
// in passing (no fill launch in the backward).
__global__ __launch_bounds__(256) void color_copy_x_kernel(misplat_params P, int D, int n_pre, int per_cam, int depth_channel,
                                                           int nxq, const float* __restrict__ colors,
                                                           const int32_t* __restrict__ radii,
                                                           const float* __restrict__ depths, float* __restrict__ grec,
                                                           float* __restrict__ featx, float4* __restrict__ zero_grec,
                                                           float4* __restrict__ zero_featx) {
    const int64_t total = (int64_t)P.n_cams * P.n_gauss;
    const int groups = 1 + nxq;
    const bool vec = n_pre == 0 && (D & 3) == 0 && ((uintptr_t)colors & 15) == 0;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total * groups; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t idx = t / groups;
        const int q = (int)(t - idx * groups);
        if (q == 0) {
            if (zero_grec) { float4* zr = zero_grec + 4 * idx; zr[0] = z4; zr[1] = z4; zr[2] = z4; zr[3] = z4; }
            if (n_pre > 0) continue;                              // (the record's colour slots are the colour kernel's)
        } else if (zero_featx) zero_featx[idx * nxq + (q - 1)] = z4;
        const int64_t src = per_cam ? idx : idx % P.n_gauss;
        const int2 rd = reinterpret_cast<const int2*>(radii)[idx];
        const bool vis = rd.x > 0 || rd.y > 0;
        float4 v = z4;
        if (vis) {
            const int c0 = 4 * q;
            if (vec && c0 + 4 <= D) {
                v = reinterpret_cast<const float4*>(colors + (size_t)src * D)[q];
            } else {
                float e[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int j = c0 + k - n_pre;
                    e[k] = j < D ? colors[(size_t)src * D + j] : ((j == D && depth_channel) ? depths[idx] : 0.f);
                }
                v = make_float4(e[0], e[1], e[2], e[3]);
            }
        }
        if (q == 0) reinterpret_cast<float4*>(grec + (size_t)idx * MISPLAT_REC + 12)[0] = v;
        else reinterpret_cast<float4*>(featx + (size_t)idx * (4 * nxq))[q - 1] = v;
    }
}

__global__ __launch_bounds__(256) void color_copy_x_bwd_kernel(misplat_params P, int D, int n_pre, int per_cam, int nxq,
                                                               const int32_t* __restrict__ radii,
                                                               const float* __restrict__ v_grec,
                                                               const float* __restrict__ v_featx,
                                                               float* __restrict__ v_colors) {
    const int64_t rows = per_cam ? (int64_t)P.n_cams * P.n_gauss : P.n_gauss;
    const int nx = 4 * nxq;
    const int groups = (D + 3) >> 2;                             // 16-byte groups of an output row
    const bool vec = n_pre == 0 && (D & 3) == 0 && ((uintptr_t)v_colors & 15) == 0;
    if (n_pre > 0) {
        // source channel j is fused channel n_pre + j: the groups of the output row do not line up with those of the fused
        // row -- four scalar picks per thread (slot 12 + c of the record row for c < 4, featx beyond)
        for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < rows * groups; t += (int64_t)gridDim.x * blockDim.x) {
            const int64_t r = t / groups;
            const int q = (int)(t - r * groups);
            const int c_lo = per_cam ? (int)(r / P.n_gauss) : 0, c_hi = per_cam ? c_lo + 1 : P.n_cams;
            const int g = (int)(r % P.n_gauss);
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            for (int ci = c_lo; ci < c_hi; ci++) {
                const int64_t idx = (int64_t)ci * P.n_gauss + g;
                const int2 rd = reinterpret_cast<const int2*>(radii)[idx];
                if (rd.x > 0 || rd.y > 0) {
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int c = n_pre + 4 * q + k;
                        if (4 * q + k < D) acc[k] += c < 4 ? v_grec[(size_t)idx * MISPLAT_REC + 12 + c] : v_featx[(size_t)idx * nx + c - 4];
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (4 * q + k < D) v_colors[(size_t)r * D + 4 * q + k] = acc[k];
        }
        return;
    }
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < rows * groups; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / groups;
        const int q = (int)(t - r * groups);
        const int c_lo = per_cam ? (int)(r / P.n_gauss) : 0, c_hi = per_cam ? c_lo + 1 : P.n_cams;
        const int g = (int)(r % P.n_gauss);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int ci = c_lo; ci < c_hi; ci++) {
            const int64_t idx = (int64_t)ci * P.n_gauss + g;
            const int2 rd = reinterpret_cast<const int2*>(radii)[idx];
            if (rd.x > 0 || rd.y > 0) {
                const float4 v = q == 0 ? reinterpret_cast<const float4*>(v_grec + (size_t)idx * MISPLAT_REC + 12)[0]
                                        : reinterpret_cast<const float4*>(v_featx + (size_t)idx * nx)[q - 1];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
        if (vec) {
            reinterpret_cast<float4*>(v_colors + (size_t)r * D)[q] = acc;
        } else {
            const float e[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (4 * q + k < D) v_colors[(size_t)r * D + 4 * q + k] = e[k];
        }
    }
}

// Backward of project_pack_fwd from packed gradient rows.  depth_slot: index (12..15) of the
// colour channel that carries the depth (RGB+ED / ED), or -1.  v_means_dir (or NULL) is the
// gradient that reached the means through the SH view direction; it is added here.
__global__ __launch_bounds__(256) void project_pack_bwd_kernel(
    misplat_params P, int depth_slot, const float* __restrict__ means, const float* __restrict__ quats,
    const float* __restrict__ scales, const float* __restrict__ opacities,
    const float* __restrict__ viewmats, const float* __restrict__ Ks, const int32_t* __restrict__ radii,
    const float* __restrict__ comps, const float* __restrict__ v_means2d, const float* __restrict__ v_grec,
    const float* __restrict__ v_means_dir, float* __restrict__ v_means, float* __restrict__ v_quats,
    float* __restrict__ v_scales, float* __restrict__ v_opacities, const float* __restrict__ v_depth_rows, int v_depth_stride) {
    for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < P.n_gauss; g += gridDim.x * blockDim.x) {
        float mean[3] = {means[3 * g], means[3 * g + 1], means[3 * g + 2]};
        float quat[4] = {quats[4 * g], quats[4 * g + 1], quats[4 * g + 2], quats[4 * g + 3]};
        float sc[3] = {scales[3 * g], scales[3 * g + 1], scales[3 * g + 2]};
        const float opac = opacities[g];
        float o_m[3] = {0.f, 0.f, 0.f}, o_q[4] = {0.f, 0.f, 0.f, 0.f}, o_s[3] = {0.f, 0.f, 0.f};
        float o_op = 0.f;
        if (v_means_dir) { o_m[0] = v_means_dir[3 * g]; o_m[1] = v_means_dir[3 * g + 1]; o_m[2] = v_means_dir[3 * g + 2]; }
        for (int ci = 0; ci < P.n_cams; ci++) {
            const int64_t idx = (int64_t)ci * P.n_gauss + g;
            if (radii[2 * idx] <= 0 && radii[2 * idx + 1] <= 0) continue;
            const Cam cam = load_cam(viewmats + 16 * ci, Ks + 9 * ci);
            const float4* vg = reinterpret_cast<const float4*>(v_grec + (size_t)idx * MISPLAT_REC);
            const float4 g0 = vg[0], g1 = vg[1], g2 = vg[2], g3 = vg[3];
            ProjGrads G;
            // v_means2d == NULL: the 2-D mean gradient is columns 0:2 of the packed row (nothing else
            // was accumulated into means2d by autograd; ops.py checks the aliasing)
            G.v_m2d[0] = v_means2d ? v_means2d[2 * idx] : g0.x;
            G.v_m2d[1] = v_means2d ? v_means2d[2 * idx + 1] : g0.y;
            G.v_conic[0] = g0.z; G.v_conic[1] = g0.w; G.v_conic[2] = g1.x;
            const float v_oeff = g1.y;
            G.v_rt = g1.z; G.v_rp[0] = g1.w; G.v_rp[1] = g2.x;
            G.v_nr[0] = g2.y; G.v_nr[1] = g2.z; G.v_nr[2] = g2.w;
            G.v_depth = depth_slot == 12 ? g3.x : (depth_slot == 13 ? g3.y : (depth_slot == 14 ? g3.z : (depth_slot == 15 ? g3.w : 0.f)));
            if (v_depth_rows) G.v_depth = v_depth_rows[(size_t)idx * v_depth_stride];      // (the depth channel lives outside the record)
            if (P.antialiased) { o_op += v_oeff * comps[idx]; G.v_comp = v_oeff * opac; }
            else { o_op += v_oeff; G.v_comp = 0.f; }
            project_bwd_one(mean, quat, sc, cam, P, G, o_m, o_q, o_s);
        }
#pragma unroll
        for (int k = 0; k < 3; k++) { v_means[3 * g + k] = o_m[k]; v_scales[3 * g + k] = o_s[k]; }
#pragma unroll
        for (int k = 0; k < 4; k++) v_quats[4 * g + k] = o_q[k];
        v_opacities[g] = o_op;
    }
}

// SH backward for scenes where few rows have a colour gradient (one camera, 16 coefficients, no Jacobian cache -- the
// companion of the on-demand forward).  The LDS-staged kernel above exists to turn per-lane gradient rows into whole
// coalesced lines; when nine rows in ten are zeros that is the wrong shape.  Two launches instead: zero_fill_kernel
// streams zeros over the whole gradient tensor at full occupancy (7 TB/s), then this kernel scans the rows -- a row whose
// `touched` flag was never set costs one byte --, queues the live ones per wave (as project_pack_bwd_sparse_kernel
// does) and, in batches of 64, fetches their 16 x 3 coefficients, walks the terms once for the clamp and once for the
// gradients, and stores the rows over the zeros.  (One kernel doing
// both was latency-bound at three waves per SIMD: 60 us, 74 with the flag's extra round trip; LDS-staged: 69.)
__global__ __launch_bounds__(256) void zero_fill_kernel(float4* __restrict__ dst, int64_t n4, float* __restrict__ tail, int n_tail) {
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) dst[i] = z;
    if (blockIdx.x == 0 && (int)threadIdx.x < n_tail) tail[threadIdx.x] = 0.f;
}

constexpr int kFlagStep = 512;          // rows a wave of the sparse backward kernels scans per step (8 flag bytes per lane)

// SH backward of one row whose colour gradient is non-zero (no Jacobian cache): coefficients in, the clamp from a
// re-evaluation, the coefficient gradient row out; dir[3] = the gradient that reaches the mean through the view direction.
template <bool SPLIT>
__device__ __forceinline__ void sh_bwd_row(int deg, int g, float ccx, float ccy, float ccz, const float* __restrict__ means,
                                           const float* __restrict__ coeffs, const float* __restrict__ coeffs_rest,
                                           const float* __restrict__ v_grec, float* __restrict__ v_coeffs,
                                           float* __restrict__ v_coeffs_rest, float (&dir)[3]) {
    const float* vg = v_grec + (size_t)g * MISPLAT_REC + 12;
    const float vg0 = vg[0], vg1 = vg[1], vg2 = vg[2];
    const float dx = means[3 * g] - ccx, dy = means[3 * g + 1] - ccy, dz = means[3 * g + 2] - ccz;
    const float n = sqrtf(dx * dx + dy * dy + dz * dz);
    const float inv = n > 0.f ? 1.0f / n : 0.f;
    const float x = dx * inv, y = dy * inv, z = dz * inv;
    float cf[48];
    if (!SPLIT) {
        const float4* s4 = reinterpret_cast<const float4*>(coeffs + (size_t)g * 48);
#pragma unroll
        for (int u = 0; u < 12; u++) { const float4 v = s4[u]; cf[4 * u] = v.x; cf[4 * u + 1] = v.y; cf[4 * u + 2] = v.z; cf[4 * u + 3] = v.w; }
    } else {
        cf[0] = coeffs[3 * (size_t)g]; cf[1] = coeffs[3 * (size_t)g + 1]; cf[2] = coeffs[3 * (size_t)g + 2];
        const float* sr = coeffs_rest + (size_t)g * 45;
#pragma unroll
        for (int u = 0; u < 45; u++) cf[3 + u] = sr[u];
    }
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
    float J[9];
    sh_eval<false>(deg, x, y, z, cf, c0, c1, c2, J);
    const float vc0 = (c0 + 0.5f > 0.f) ? vg0 : 0.f, vc1 = (c1 + 0.5f > 0.f) ? vg1 : 0.f, vc2 = (c2 + 0.5f > 0.f) ? vg2 : 0.f;
    float vd0 = 0.f, vd1 = 0.f, vd2 = 0.f;
    misplat_sh::sh_grad<false>(deg, x, y, z, cf, vc0, vc1, vc2, vd0, vd1, vd2, nullptr);
    const int nb = (deg + 1) * (deg + 1);
#pragma unroll
    for (int k = 0; k < 16; k++)
        if (k >= nb) { cf[3 * k] = 0.f; cf[3 * k + 1] = 0.f; cf[3 * k + 2] = 0.f; }       // above the active degree
    if (!SPLIT) {
        float4* o4 = reinterpret_cast<float4*>(v_coeffs + (size_t)g * 48);
#pragma unroll
        for (int u = 0; u < 12; u++) o4[u] = make_float4(cf[4 * u], cf[4 * u + 1], cf[4 * u + 2], cf[4 * u + 3]);
    } else {
        v_coeffs[3 * (size_t)g] = cf[0]; v_coeffs[3 * (size_t)g + 1] = cf[1]; v_coeffs[3 * (size_t)g + 2] = cf[2];
        float* orr = v_coeffs_rest + (size_t)g * 45;
#pragma unroll
        for (int u = 0; u < 45; u++) orr[u] = cf[3 + u];
    }
    const float dot = x * vd0 + y * vd1 + z * vd2;
    dir[0] = (vd0 - x * dot) * inv; dir[1] = (vd1 - y * dot) * inv; dir[2] = (vd2 - z * dot) * inv;
}

template <bool SPLIT>
__global__ __launch_bounds__(64) void color_sh_bwd_sparse_kernel(
    misplat_params P, int deg, const float* __restrict__ means, const float* __restrict__ viewmats,
    const float* __restrict__ coeffs, const float* __restrict__ coeffs_rest, const int32_t* __restrict__ radii,
    const float* __restrict__ v_grec, float* __restrict__ v_coeffs, float* __restrict__ v_coeffs_rest,
    float* __restrict__ v_means_dir) {
    __shared__ int queue[128];
    const int lane = threadIdx.x;
    const int live_block = (int)blockIdx.x;
    const int live_grid = (int)gridDim.x;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const float* V = viewmats;
    const float ccx = -(V[0] * V[3] + V[4] * V[7] + V[8] * V[11]);
    const float ccy = -(V[1] * V[3] + V[5] * V[7] + V[9] * V[11]);
    const float ccz = -(V[2] * V[3] + V[6] * V[7] + V[10] * V[11]);
    auto heavy = [&](int g) {
        float dir[3];
        sh_bwd_row<SPLIT>(deg, g, ccx, ccy, ccz, means, coeffs, coeffs_rest, v_grec, v_coeffs, v_coeffs_rest, dir);
        v_means_dir[3 * g] = dir[0]; v_means_dir[3 * g + 1] = dir[1]; v_means_dir[3 * g + 2] = dir[2];
    };
    int qn = 0;
    auto push = [&](bool live, int g) {
        const unsigned long long mask = __ballot(live);
        if (live) queue[qn + __popcll(mask & lt)] = g;
        qn += __popcll(mask);
        __builtin_amdgcn_wave_barrier();
        if (qn >= 64) {
            const int gq = queue[lane];
            const int keep = lane + 64 < qn ? queue[lane + 64] : 0;
            __builtin_amdgcn_wave_barrier();
            queue[lane] = keep;
            qn -= 64;
            heavy(gq);
        }
    };
    // (v_means_dir was cleared with the gradient tensor: the dead rows are done)
    if (P.touched && (((uintptr_t)P.touched) & 15) == 0) {
        // flags 8 at a time per lane: one load covers kFlagStep = 512 rows of the wave, the ballots run from registers (a
        // dependent one-byte load per 64 rows made the scan, not the gradients, the cost of this kernel; 1 024 rows per
        // step left too few waves: a wave's two batches ran one behind the other at one wave per SIMD)
        for (int64_t base = (int64_t)live_block * kFlagStep; base < P.n_gauss; base += (int64_t)live_grid * kFlagStep) {
            const int64_t r0 = base + 8 * lane;
            unsigned long long fl = 0ull;
            if (r0 + 8 <= P.n_gauss) fl = *reinterpret_cast<const unsigned long long*>(P.touched + r0);
            else
                for (int b = 0; b < 8; b++)
                    if (r0 + b < P.n_gauss) fl |= (unsigned long long)P.touched[r0 + b] << (8 * b);
            if (__ballot(fl != 0ull) == 0ull) continue;
#pragma unroll 1
            for (int b = 0; b < 8; b++) {                    // (rolled: `push` holds the whole gradient evaluation)
                push((fl & 0xffull) != 0ull, (int)(r0 + b));
                fl >>= 8;
            }
        }
    } else {
        for (int base = live_block * 64; base < P.n_gauss; base += live_grid * 64) {
            const int g = base + lane;
            bool live = false;
            if (g < P.n_gauss) {
                if (P.touched) live = P.touched[g] != 0;
                else {
                    const float* vg = v_grec + (size_t)g * MISPLAT_REC + 12;
                    live = (radii[2 * (int64_t)g] > 0 || radii[2 * (int64_t)g + 1] > 0) && (vg[0] != 0.f || vg[1] != 0.f || vg[2] != 0.f);
                }
            }
            push(live, g);
        }
    }
    if (lane < qn) heavy(queue[lane]);
}

// Projection backward of one row from its packed gradient row; dir[3]: what already reached the mean (the SH direction
// gradient), the start value of the mean's gradient.
__device__ __forceinline__ void pp_bwd_row(const misplat_params& P, const Cam& cam, int depth_slot, int g,
                                           const float* __restrict__ means, const float* __restrict__ quats,
                                           const float* __restrict__ scales, const float* __restrict__ opacities,
                                           const float* __restrict__ comps, const float* __restrict__ v_means2d,
                                           const float* __restrict__ v_grec, const float (&dir)[3], float* __restrict__ v_means,
                                           float* __restrict__ v_quats, float* __restrict__ v_scales,
                                           float* __restrict__ v_opacities, const float* __restrict__ v_depth_rows = nullptr,
                                           int v_depth_stride = 0, bool mean_sums = false, float2* __restrict__ m2d_out = nullptr) {
    float mean[3] = {means[3 * g], means[3 * g + 1], means[3 * g + 2]};
    float quat[4] = {quats[4 * g], quats[4 * g + 1], quats[4 * g + 2], quats[4 * g + 3]};
    float sc[3] = {scales[3 * g], scales[3 * g + 1], scales[3 * g + 2]};
    float opac = opacities[g];
    apply_activations(P, sc, opac);
    float o_m[3] = {dir[0], dir[1], dir[2]}, o_q[4] = {0.f, 0.f, 0.f, 0.f}, o_s[3] = {0.f, 0.f, 0.f};
    float o_op = 0.f;
    const float4* vg = reinterpret_cast<const float4*>(v_grec + (size_t)g * MISPLAT_REC);
    const float4 g0 = vg[0], g1 = vg[1], g2 = vg[2], g3 = vg[3];
    ProjGrads G;
    G.v_m2d[0] = v_means2d ? v_means2d[2 * g] : g0.x;
    G.v_m2d[1] = v_means2d ? v_means2d[2 * g + 1] : g0.y;
    G.m2d_sums = mean_sums;
    G.v_conic[0] = g0.z; G.v_conic[1] = g0.w; G.v_conic[2] = g1.x;
    float v_oeff = g1.y;
    if (mean_sums) {                       // (slot 5 = o_eff times the gradient: blend.hip, MSUM; a flagged row has o_eff >= 1/255)
        const float oe = P.antialiased ? opac * comps[g] : opac;
        v_oeff = oe > 0.f ? v_oeff / oe : 0.f;
    }
    G.v_rt = g1.z; G.v_rp[0] = g1.w; G.v_rp[1] = g2.x;
    G.v_nr[0] = g2.y; G.v_nr[1] = g2.z; G.v_nr[2] = g2.w;
    G.v_depth = depth_slot == 12 ? g3.x : (depth_slot == 13 ? g3.y : (depth_slot == 14 ? g3.z : (depth_slot == 15 ? g3.w : 0.f)));
    if (v_depth_rows) G.v_depth = v_depth_rows[(size_t)g * v_depth_stride];
    if (P.antialiased) { o_op += v_oeff * comps[g]; G.v_comp = v_oeff * opac; }
    else { o_op += v_oeff; G.v_comp = 0.f; }
    project_bwd_one(mean, quat, sc, cam, P, G, o_m, o_q, o_s);
    if (m2d_out) m2d_out[g] = make_float2(G.v_m2d[0], G.v_m2d[1]);       // (meta["means2d"].grad of the row)
    if (P.activations & 1) { o_s[0] *= sc[0]; o_s[1] *= sc[1]; o_s[2] *= sc[2]; }       // d exp(x) = exp(x) dx
    if (P.activations & 2) o_op *= opac * (1.0f - opac);                                 // d sigmoid(x) = s (1 - s) dx
#pragma unroll
    for (int k = 0; k < 3; k++) { v_means[3 * g + k] = o_m[k]; v_scales[3 * g + k] = o_s[k]; }
#pragma unroll
    for (int k = 0; k < 4; k++) v_quats[4 * g + k] = o_q[k];
    v_opacities[g] = o_op;
}

// One camera: most rows of a dense scene never receive a gradient -- the compositing stops at the first opaque layers,
// and the packed gradient row of a Gaussian behind them is still the zeros the forward left (1 M random Gaussians at
// 1080p: 11 % of the visible rows get one, at 5 M 2 %; scripts/touched_fraction.py).  A zero row in gives a zero row out,
// but a thread-per-row kernel gains nothing from skipping it: a wave runs as long as its one live lane.  So every wave
// SCANS its rows 64 at a time (visibility, the 64-byte row, is it non-zero?), writes the outputs of the dead rows at
// once, queues the live ones in LDS, and runs the ~1 400-instruction backward only on full batches of 64 live rows.
__global__ __launch_bounds__(64) void project_pack_bwd_sparse_kernel(
    misplat_params P, int depth_slot, const float* __restrict__ means, const float* __restrict__ quats,
    const float* __restrict__ scales, const float* __restrict__ opacities,
    const float* __restrict__ viewmats, const float* __restrict__ Ks, const int32_t* __restrict__ radii,
    const float* __restrict__ comps, const float* __restrict__ v_means2d, const float* __restrict__ v_grec,
    const float* __restrict__ v_means_dir, float* __restrict__ v_means, float* __restrict__ v_quats,
    float* __restrict__ v_scales, float* __restrict__ v_opacities, const float* __restrict__ v_depth_rows, int v_depth_stride) {
    __shared__ int queue[128];
    const int lane = threadIdx.x;
    const unsigned long long lt = (1ull << lane) - 1ull;
    int qn = 0;
    const Cam cam = load_cam(viewmats, Ks);
    auto heavy = [&](int g) {
        float dir[3] = {0.f, 0.f, 0.f};
        if (v_means_dir) { dir[0] = v_means_dir[3 * g]; dir[1] = v_means_dir[3 * g + 1]; dir[2] = v_means_dir[3 * g + 2]; }
        pp_bwd_row(P, cam, depth_slot, g, means, quats, scales, opacities, comps, v_means2d, v_grec, dir, v_means, v_quats, v_scales,
                   v_opacities, v_depth_rows, v_depth_stride);
    };
    for (int base = blockIdx.x * 64; base < P.n_gauss; base += gridDim.x * 64) {
        const int g = base + lane;
        bool live = false;
        if (g < P.n_gauss) {
            if (P.touched && !v_means2d) live = P.touched[g] != 0;         // (set by the compositing backward: no row fetch)
            else if (radii[2 * g] > 0 || radii[2 * g + 1] > 0) {
                const float4* vg = reinterpret_cast<const float4*>(v_grec + (size_t)g * MISPLAT_REC);
                const float4 g0 = vg[0], g1 = vg[1], g2 = vg[2], g3 = vg[3];
                live = g0.x != 0.f || g0.y != 0.f || g0.z != 0.f || g0.w != 0.f || g1.x != 0.f || g1.y != 0.f || g1.z != 0.f ||
                       g1.w != 0.f || g2.x != 0.f || g2.y != 0.f || g2.z != 0.f || g2.w != 0.f || g3.x != 0.f || g3.y != 0.f ||
                       g3.z != 0.f || g3.w != 0.f;
                if (v_means2d) live = live || v_means2d[2 * g] != 0.f || v_means2d[2 * g + 1] != 0.f;
                if (v_depth_rows) live = live || v_depth_rows[(size_t)g * v_depth_stride] != 0.f;
            }
            if (!live) {
                float m0 = 0.f, m1 = 0.f, m2 = 0.f;
                if (v_means_dir) { m0 = v_means_dir[3 * g]; m1 = v_means_dir[3 * g + 1]; m2 = v_means_dir[3 * g + 2]; }
                v_means[3 * g] = m0; v_means[3 * g + 1] = m1; v_means[3 * g + 2] = m2;
                v_scales[3 * g] = 0.f; v_scales[3 * g + 1] = 0.f; v_scales[3 * g + 2] = 0.f;
                *reinterpret_cast<float4*>(v_quats + 4 * (size_t)g) = make_float4(0.f, 0.f, 0.f, 0.f);
                v_opacities[g] = 0.f;
            }
        }
        const unsigned long long mask = __ballot(live);
        if (live) queue[qn + __popcll(mask & lt)] = g;
        qn += __popcll(mask);
        __builtin_amdgcn_wave_barrier();
        if (qn >= 64) {
            const int gq = queue[lane];
            const int keep = lane + 64 < qn ? queue[lane + 64] : 0;
            __builtin_amdgcn_wave_barrier();
            queue[lane] = keep;
            qn -= 64;
            heavy(gq);
        }
    }
    if (lane < qn) heavy(queue[lane]);
}

// sh_bwd_row for a whole wave of queued rows, the coefficient rows staged through LDS.  One thread per row reading its
// own 192 bytes puts 64 different lines into every load and store instruction (and 48 coefficients + 48 gradients into
// registers); here the wave moves the rows cooperatively -- 5 rows of 12 x 16 B per instruction ([N,16,3]), or one row of
// 45 + 3 floats (features_rest + features_dc) -- and sh_eval / sh_grad read and overwrite the row in LDS term by term.
// Row stride 49 words: a thread's word k lands in bank (49 lane + k) mod 64, all different.
constexpr int kShStageStride = 49;
template <bool SPLIT>
__device__ __forceinline__ void sh_bwd_wave(int deg, int g, bool active, int nrows, float ccx, float ccy, float ccz,
                                            const float* __restrict__ means, const float* __restrict__ coeffs,
                                            const float* __restrict__ coeffs_rest, const float* __restrict__ v_grec,
                                            float* __restrict__ v_coeffs, float* __restrict__ v_coeffs_rest,
                                            float* stage, int* rows, float (&dir)[3]) {
    const int lane = threadIdx.x;
    rows[lane] = active ? g : 0;
    __syncthreads();
    if (!SPLIT) {
        const int sub = lane / 12, q = lane - 12 * sub;
        float4 v[13];
#pragma unroll
        for (int it = 0; it < 13; it++) {
            const int r = it * 5 + sub;
            v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (lane < 60 && r < nrows) v[it] = reinterpret_cast<const float4*>(coeffs + (size_t)rows[r] * 48)[q];
        }
#pragma unroll
        for (int it = 0; it < 13; it++) {
            const int r = it * 5 + sub;
            if (lane < 60 && r < nrows) {
                float* d = stage + r * kShStageStride + 4 * q;
                d[0] = v[it].x; d[1] = v[it].y; d[2] = v[it].z; d[3] = v[it].w;
            }
        }
    } else {
        const int word = lane < 45 ? lane + 3 : lane - 45;                 // lanes 45..47: the three DC coefficients
#pragma unroll 1
        for (int r0 = 0; r0 < nrows; r0 += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                v[u] = 0.f;
                if (r0 + u < nrows && lane < 48) {
                    const size_t gr = (size_t)rows[r0 + u];
                    v[u] = lane < 45 ? coeffs_rest[gr * 45 + lane] : coeffs[gr * 3 + (lane - 45)];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (r0 + u < nrows && lane < 48) stage[(r0 + u) * kShStageStride + word] = v[u];
        }
    }
    __syncthreads();
    float x = 0.f, y = 0.f, z = 0.f, inv = 0.f, vd0 = 0.f, vd1 = 0.f, vd2 = 0.f;
    if (active) {
        float* cf = stage + lane * kShStageStride;
        const float* vg = v_grec + (size_t)g * MISPLAT_REC + 12;
        const float vg0 = vg[0], vg1 = vg[1], vg2 = vg[2];
        const float dx = means[3 * g] - ccx, dy = means[3 * g + 1] - ccy, dz = means[3 * g + 2] - ccz;
        const float n = sqrtf(dx * dx + dy * dy + dz * dz);
        inv = n > 0.f ? 1.0f / n : 0.f;
        x = dx * inv; y = dy * inv; z = dz * inv;
        float c0 = 0.f, c1 = 0.f, c2 = 0.f;
        float J[9];
        sh_eval<false>(deg, x, y, z, cf, c0, c1, c2, J);
        const float vc0 = (c0 + 0.5f > 0.f) ? vg0 : 0.f, vc1 = (c1 + 0.5f > 0.f) ? vg1 : 0.f, vc2 = (c2 + 0.5f > 0.f) ? vg2 : 0.f;
        misplat_sh::sh_grad<false>(deg, x, y, z, cf, vc0, vc1, vc2, vd0, vd1, vd2, nullptr);
        const int nb = (deg + 1) * (deg + 1);
        for (int k = nb; k < 16; k++) { cf[3 * k] = 0.f; cf[3 * k + 1] = 0.f; cf[3 * k + 2] = 0.f; }   // above the active degree
    }
    __syncthreads();
    if (!SPLIT) {
        const int sub = lane / 12, q = lane - 12 * sub;
#pragma unroll
        for (int it = 0; it < 13; it++) {
            const int r = it * 5 + sub;
            if (lane < 60 && r < nrows) {
                const float* d = stage + r * kShStageStride + 4 * q;
                reinterpret_cast<float4*>(v_coeffs + (size_t)rows[r] * 48)[q] = make_float4(d[0], d[1], d[2], d[3]);
            }
        }
    } else {
        const int word = lane < 45 ? lane + 3 : lane - 45;
#pragma unroll 4
        for (int r = 0; r < nrows; r++) {
            if (lane < 48) {
                const size_t gr = (size_t)rows[r];
                const float v = stage[r * kShStageStride + word];
                if (lane < 45) v_coeffs_rest[gr * 45 + lane] = v;
                else v_coeffs[gr * 3 + (lane - 45)] = v;
            }
        }
    }
    const float dot = x * vd0 + y * vd1 + z * vd2;
    dir[0] = (vd0 - x * dot) * inv; dir[1] = (vd1 - y * dot) * inv; dir[2] = (vd2 - z * dot) * inv;
}

// Both per-Gaussian backward stages of the flagged rows in ONE launch (outputs cleared beforehand, one camera, SH colours
// without Jacobian cache): one scan of the flags, one queue, and a row's SH direction gradient goes from the SH stage to
// the projection stage in registers (no v_means_dir round trip, one launch and one dependent scan less).
// FPL: flag bytes a lane scans per step (a wave covers 64 * FPL rows per step).
template <bool SPLIT, int FPL>
__global__ __launch_bounds__(64) void gauss_bwd_sparse_kernel(
    misplat_params P, int deg, int depth_slot, const float* __restrict__ means, const float* __restrict__ quats,
    const float* __restrict__ scales, const float* __restrict__ opacities, const float* __restrict__ viewmats,
    const float* __restrict__ Ks, const float* __restrict__ coeffs, const float* __restrict__ coeffs_rest,
    const float* __restrict__ comps, const float* __restrict__ v_grec, float* __restrict__ v_coeffs,
    float* __restrict__ v_coeffs_rest, float* __restrict__ v_means, float* __restrict__ v_quats, float* __restrict__ v_scales,
    float* __restrict__ v_opacities, float2* __restrict__ v_m2d, const float* __restrict__ v_featx, int nx,
    float* __restrict__ v_features, int n_feat, int depth_in_featx, int mean_sums) {
    // (N-D records, v_features != NULL: the flagged rows' feature gradients are picked from the record row's slot 15 and the
    // featx row -- color_copy_x_bwd's job for these rows --, the depth gradient rides behind the last feature in featx)
    __shared__ int queue[128];
    __shared__ int rows[64];
    __shared__ float stage[64 * kShStageStride];
    const int lane = threadIdx.x;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const float* V = viewmats;
    const float ccx = -(V[0] * V[3] + V[4] * V[7] + V[8] * V[11]);
    const float ccy = -(V[1] * V[3] + V[5] * V[7] + V[9] * V[11]);
    const float ccz = -(V[2] * V[3] + V[6] * V[7] + V[10] * V[11]);
    const Cam cam = load_cam(viewmats, Ks);
    auto heavy = [&](int g, int nrows) {           // called by the whole wave: lanes < nrows hold a row
        float dir[3];
        const bool active = lane < nrows;
        sh_bwd_wave<SPLIT>(deg, g, active, nrows, ccx, ccy, ccz, means, coeffs, coeffs_rest, v_grec, v_coeffs, v_coeffs_rest,
                           stage, rows, dir);
        if (active) {
            pp_bwd_row(P, cam, depth_slot, g, means, quats, scales, opacities, comps, nullptr, v_grec, dir, v_means, v_quats,
                       v_scales, v_opacities, depth_in_featx ? v_featx + (n_feat - 1) : nullptr, depth_in_featx ? nx : 0,
                       mean_sums != 0, v_m2d);
            if (v_features) {
                const float* gx = v_featx + (size_t)g * nx;
                float* o = v_features + (size_t)g * n_feat;
                o[0] = v_grec[(size_t)g * MISPLAT_REC + 15];
                for (int j = 1; j < n_feat; j++) o[j] = gx[j - 1];
            }
        }
    };
    int qn = 0;
    for (int64_t base = (int64_t)blockIdx.x * (64 * FPL); base < P.n_gauss; base += (int64_t)gridDim.x * (64 * FPL)) {
        const int64_t r0 = base + FPL * lane;
        unsigned long long fl = 0ull;
        if (r0 + FPL <= P.n_gauss) {
            if (FPL == 8) fl = *reinterpret_cast<const unsigned long long*>(P.touched + r0);
            else fl = *reinterpret_cast<const uint32_t*>(P.touched + r0);
        } else
            for (int b = 0; b < FPL; b++)
                if (r0 + b < P.n_gauss) fl |= (unsigned long long)P.touched[r0 + b] << (8 * b);
        if (__ballot(fl != 0ull) == 0ull) continue;
#pragma unroll 1
        for (int b = 0; b < FPL; b++) {
            const bool live = (fl & 0xffull) != 0ull;
            const unsigned long long mask = __ballot(live);
            if (live) queue[qn + __popcll(mask & lt)] = (int)(r0 + b);
            qn += __popcll(mask);
            __builtin_amdgcn_wave_barrier();
            if (qn >= 64) {
                const int gq = queue[lane];
                const int keep = lane + 64 < qn ? queue[lane + 64] : 0;
                __builtin_amdgcn_wave_barrier();
                queue[lane] = keep;
                qn -= 64;
                heavy(gq, 64);
            }
            fl >>= 8;
        }
    }
    if (qn > 0) heavy(lane < qn ? queue[lane] : 0, qn);
}

inline int grid_for(int64_t n, int block) {
    int64_t b = (n + block - 1) / block;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (int)b;
}
inline int check_launch() { return hipGetLastError() == hipSuccess ? MISPLAT_OK : MISPLAT_ELAUNCH; }

}  // namespace

int misplat_internal::zero_fill(float* dst, int64_t n_floats, int max_blocks, hipStream_t s) {
    if (n_floats <= 0) return MISPLAT_OK;
    if (((uintptr_t)dst) & 15) return MISPLAT_EINVAL;
    const int64_t n4 = n_floats / 4;
    int g = grid_for(n4, 256);
    if (max_blocks > 0 && g > max_blocks) g = max_blocks;
    hipLaunchKernelGGL(zero_fill_kernel, dim3(g), dim3(256), 0, s, (float4*)dst, n4, dst + 4 * n4, (int)(n_floats - 4 * n4));
    return check_launch();
}

extern "C" int misplat_project_fwd(const misplat_params* p, const float* means, const float* quats,
                                   const float* scales, const float* opacities, const float* viewmats,
                                   const float* Ks, int32_t* radii, float* means2d, float* depths,
                                   float* conics, float* compensations, float* ray_ts,
                                   float* ray_planes, float* normals, misplat_stream_t stream) {
    if (!p || p->n_gauss < 0 || p->n_cams < 1 || p->width < 1 || p->height < 1 || p->activations) return MISPLAT_EINVAL;
    int64_t total = (int64_t)p->n_gauss * p->n_cams;
    if (total == 0) return MISPLAT_OK;
    hipLaunchKernelGGL(project_fwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, *p,
                       means, quats, scales, opacities, viewmats, Ks, radii, means2d, depths, conics,
                       compensations, ray_ts, ray_planes, normals);
    return check_launch();
}

extern "C" int misplat_project_bwd(const misplat_params* p, const float* means, const float* quats,
                                   const float* scales, const float* viewmats, const float* Ks,
                                   const int32_t* radii, const float* v_means2d, const float* v_depths,
                                   const float* v_conics, const float* v_compensations,
                                   const float* v_ray_ts, const float* v_ray_planes,
                                   const float* v_normals, float* v_means, float* v_quats,
                                   float* v_scales, misplat_stream_t stream) {
    if (!p || p->n_gauss < 0 || p->n_cams < 1 || p->activations) return MISPLAT_EINVAL;
    if (p->n_gauss == 0) return MISPLAT_OK;
    hipLaunchKernelGGL(project_bwd_kernel, dim3(grid_for(p->n_gauss, 256)), dim3(256), 0, (hipStream_t)stream,
                       *p, means, quats, scales, viewmats, Ks, radii, v_means2d, v_depths, v_conics,
                       v_compensations, v_ray_ts, v_ray_planes, v_normals, v_means, v_quats, v_scales);
    return check_launch();
}

extern "C" int misplat_sh_fwd(int32_t n_gauss, int32_t n_cams, int32_t K, int32_t degree,
                              const float* dirs, const float* coeffs, const int32_t* radii,
                              float* colors, misplat_stream_t stream) {
    if (n_gauss < 0 || n_cams < 1 || degree < 0 || degree > 3 || K < (degree + 1) * (degree + 1)) return MISPLAT_EINVAL;
    int64_t total = (int64_t)n_gauss * n_cams;
    if (total == 0) return MISPLAT_OK;
    hipLaunchKernelGGL(sh_fwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, n_gauss,
                       n_cams, K, degree, dirs, coeffs, radii, colors);
    return check_launch();
}

extern "C" int misplat_sh_bwd(int32_t n_gauss, int32_t n_cams, int32_t K, int32_t degree,
                              const float* dirs, const float* coeffs, const int32_t* radii,
                              const float* v_colors, float* v_coeffs, float* v_dirs,
                              misplat_stream_t stream) {
    if (n_gauss < 0 || n_cams < 1 || degree < 0 || degree > 3 || K < (degree + 1) * (degree + 1)) return MISPLAT_EINVAL;
    if (n_gauss == 0) return MISPLAT_OK;
    hipLaunchKernelGGL(sh_bwd_kernel, dim3(grid_for(n_gauss, 256)), dim3(256), 0, (hipStream_t)stream, n_gauss,
                       n_cams, K, degree, dirs, coeffs, radii, v_colors, v_coeffs, v_dirs);
    return check_launch();
}

// ------------------------------------------------------------------ fused path entry points
extern "C" int misplat_project_pack_fwd(const misplat_params* p, const float* means, const float* quats,
                                        const float* scales, const float* opacities, const float* viewmats,
                                        const float* Ks, int32_t* radii, float* means2d, float* depths,
                                        float* compensations, float* grec, uint32_t* zero_words, int32_t n_zero,
                                        float* lazy_rows, float* abs_rows, misplat_stream_t stream) {
    return misplat_internal::project_pack_fwd(p, means, quats, scales, opacities, viewmats, Ks, radii, means2d, depths,
                                              compensations, grec, zero_words, n_zero, lazy_rows, abs_rows, 1, nullptr,
                                              nullptr, 0, 0, (hipStream_t)stream);
}

int misplat_internal::project_pack_fwd(const misplat_params* p, const float* means, const float* quats,
                                       const float* scales, const float* opacities, const float* viewmats,
                                       const float* Ks, int32_t* radii, float* means2d, float* depths,
                                       float* compensations, float* grec, uint32_t* zero_words, int32_t n_zero,
                                       float* lazy_rows, float* abs_rows, int32_t clear_lazy_rows,
                                       const int32_t* order_table, int32_t* order_sel, int32_t order_slots,
                                       int32_t order_stride, hipStream_t stream) {
    if (order_sel && (!order_table || order_slots < 1 || order_stride < MISPLAT_ORDER_HEADER)) return MISPLAT_EINVAL;
    if (!p || p->n_gauss < 0 || p->n_cams < 1 || p->width < 1 || p->height < 1) return MISPLAT_EINVAL;
    int64_t total = (int64_t)p->n_gauss * p->n_cams;
    if (n_zero < 0 || (n_zero > 0 && !zero_words)) return MISPLAT_EINVAL;
    if (total == 0 && n_zero == 0) return MISPLAT_OK;
    if (total > 0 && !opacities) return MISPLAT_EINVAL;
    hipLaunchKernelGGL(project_pack_fwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, *p,
                       means, quats, scales, opacities, viewmats, Ks, radii, means2d, depths, compensations,
                       (float4*)grec, zero_words, n_zero, (float4*)lazy_rows, (float2*)abs_rows, (int)clear_lazy_rows,
                       order_table, order_sel, (int)order_slots, (int)order_stride);
    return check_launch();
}

extern "C" int misplat_color_fwd(const misplat_params* p, int32_t sh_degree, int32_t K_or_D, int32_t n_color,
                                 int32_t per_cam, int32_t depth_channel, const float* means,
                                 const float* viewmats, const float* coeffs_or_colors, const float* coeffs_rest,
                                 const int32_t* radii, const float* depths, float* grec, float* sh_aux,
                                 float* zero_rows, misplat_stream_t stream) {
    return misplat_internal::color_fwd(p, sh_degree, K_or_D, n_color, per_cam, depth_channel, means, viewmats, coeffs_or_colors,
                                       coeffs_rest, radii, depths, grec, sh_aux, zero_rows, nullptr, 0, (hipStream_t)stream);
}

int misplat_internal::color_fwd(const misplat_params* p, int32_t sh_degree, int32_t K_or_D, int32_t n_color, int32_t per_cam,
                                int32_t depth_channel, const float* means, const float* viewmats, const float* coeffs_or_colors,
                                const float* coeffs_rest, const int32_t* radii, const float* depths, float* grec, float* sh_aux,
                                float* zero_rows, const float* slot3, int32_t slot3_stride, hipStream_t stream) {
    if (!p || p->n_gauss < 0 || p->n_cams < 1 || (slot3 && (sh_degree < 0 || depth_channel))) return MISPLAT_EINVAL;
    if (n_color < 0 || n_color + (depth_channel ? 1 : 0) > 4) return MISPLAT_EINVAL;
    if (p->n_gauss == 0) return MISPLAT_OK;
    hipStream_t s = (hipStream_t)stream;
    if (sh_degree >= 0) {
        if (sh_degree > 3 || K_or_D < (sh_degree + 1) * (sh_degree + 1) || K_or_D > 16 || n_color != 3) return MISPLAT_EINVAL;
        constexpr int BLK = 64;
        const int n_blocks = (p->n_gauss + BLK - 1) / BLK;
        const size_t lds = (size_t)BLK * (3 * K_or_D + 1) * sizeof(float);
        // K = 16 with 16-byte aligned coefficient arrays: the vectorised variant (compile-time row length)
        const bool k16 = K_or_D == 16 && ((uintptr_t)coeffs_or_colors & 15) == 0 && ((uintptr_t)coeffs_rest & 15) == 0;
#define LAUNCH_SH_FWD(AUX_, KC_, SPLIT_)                                                                           \
    hipLaunchKernelGGL((color_sh_kernel<false, BLK, false, AUX_, KC_, SPLIT_>),                                     \
                       dim3(n_blocks < 16384 ? n_blocks : 16384),                                                   \
                       dim3(BLK), lds, s, *p, K_or_D, sh_degree, depth_channel, means, viewmats, coeffs_or_colors, \
                       coeffs_rest, radii, depths, grec, (const float*)nullptr, (float*)nullptr, (float*)nullptr,  \
                       (float*)nullptr, sh_aux, (float4*)zero_rows, slot3, (int)slot3_stride)
#define LAUNCH_SH_FWD_K(AUX_)                                                                                      \
    do {                                                                                                           \
        if (!k16) LAUNCH_SH_FWD(AUX_, 0, -1);                                                                      \
        else if (coeffs_rest) LAUNCH_SH_FWD(AUX_, 16, 1);                                                          \
        else LAUNCH_SH_FWD(AUX_, 16, 0);                                                                           \
    } while (0)
        if (sh_aux) LAUNCH_SH_FWD_K(true); else LAUNCH_SH_FWD_K(false);
#undef LAUNCH_SH_FWD_K
#undef LAUNCH_SH_FWD
    } else {
        if (K_or_D < n_color) return MISPLAT_EINVAL;
        int64_t total = (int64_t)p->n_gauss * p->n_cams;
        hipLaunchKernelGGL(color_copy_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, *p, K_or_D, n_color,
                           per_cam, depth_channel, coeffs_or_colors, radii, depths, grec, (float4*)zero_rows);
    }
    return check_launch();
}

extern "C" int misplat_color_bwd(const misplat_params* p, int32_t sh_degree, int32_t K_or_D, int32_t n_color,
                                 int32_t per_cam, const float* means, const float* viewmats,
                                 const float* coeffs_or_colors, const float* coeffs_rest, const int32_t* radii,
                                 const float* v_grec, float* v_coeffs_or_colors, float* v_coeffs_rest,
                                 float* v_means_dir, const float* sh_aux, misplat_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!p || p->n_gauss < 0 || p->n_cams < 1 || n_color < 0 || n_color > 4) return MISPLAT_EINVAL;
    if (p->n_gauss == 0) return MISPLAT_OK;
    if (sh_degree >= 0) {
        if (sh_degree > 3 || K_or_D < (sh_degree + 1) * (sh_degree + 1) || K_or_D > 16 || !v_means_dir) return MISPLAT_EINVAL;
        constexpr int BLK = 64;
        const int n_blocks = (p->n_gauss + BLK - 1) / BLK;
        const size_t lds = (size_t)BLK * (3 * K_or_D + 1) * sizeof(float);
        float* ax = const_cast<float*>(sh_aux);
        const bool k16 = K_or_D == 16 && ((uintptr_t)coeffs_or_colors & 15) == 0 && ((uintptr_t)coeffs_rest & 15) == 0 &&
                         ((uintptr_t)v_coeffs_or_colors & 15) == 0 && ((uintptr_t)v_coeffs_rest & 15) == 0;
        if (k16 && !sh_aux && p->n_cams == 1 && (coeffs_rest != nullptr) == (v_coeffs_rest != nullptr)) {
            // no Jacobian cache: the on-demand forward ran, i.e. a scene where few rows have a gradient
            const int grid = n_blocks < 2048 ? n_blocks : 2048;          // one wave each: every wave scans enough rows to fill batches
            auto zero = [&](float* ptr_, int64_t n_floats) {            // (16-byte aligned by the k16 test above)
                const int64_t n4 = n_floats / 4;
                hipLaunchKernelGGL(zero_fill_kernel, dim3(grid_for(n4, 256)), dim3(256), 0, s, (float4*)ptr_, n4, ptr_ + 4 * n4,
                                   (int)(n_floats - 4 * n4));
            };
            if (((uintptr_t)v_means_dir & 15) != 0) return MISPLAT_EINVAL;
            if (coeffs_rest) { zero(v_coeffs_or_colors, (int64_t)p->n_gauss * 3); zero(v_coeffs_rest, (int64_t)p->n_gauss * 45); }
            else zero(v_coeffs_or_colors, (int64_t)p->n_gauss * 48);
            zero(v_means_dir, (int64_t)p->n_gauss * 3);
            if (coeffs_rest)
                hipLaunchKernelGGL(color_sh_bwd_sparse_kernel<true>, dim3(grid), dim3(64), 0, s, *p, sh_degree, means, viewmats,
                                   coeffs_or_colors, coeffs_rest, radii, v_grec, v_coeffs_or_colors, v_coeffs_rest, v_means_dir);
            else
                hipLaunchKernelGGL(color_sh_bwd_sparse_kernel<false>, dim3(grid), dim3(64), 0, s, *p, sh_degree, means, viewmats,
                                   coeffs_or_colors, coeffs_rest, radii, v_grec, v_coeffs_or_colors, v_coeffs_rest, v_means_dir);
            return check_launch();
        }
#define LAUNCH_SH_BWD(MULTI_, AUX_, KC_, SPLIT_)                                                                     \
    hipLaunchKernelGGL((color_sh_kernel<true, BLK, MULTI_, AUX_, KC_, SPLIT_>),                                       \
                       dim3(n_blocks < 16384 ? n_blocks : 16384),                                                     \
                       dim3(BLK), lds, s, *p, K_or_D, sh_degree, 0, means, viewmats, coeffs_or_colors, coeffs_rest,   \
                       radii, (const float*)nullptr, (float*)nullptr, v_grec, v_coeffs_or_colors, v_coeffs_rest,     \
                       v_means_dir, ax)
#define LAUNCH_SH_BWD_K(MULTI_, AUX_)                                                    \
    do {                                                                                 \
        if (!k16) LAUNCH_SH_BWD(MULTI_, AUX_, 0, -1);                                    \
        else if (coeffs_rest || v_coeffs_rest) LAUNCH_SH_BWD(MULTI_, AUX_, 16, 1);       \
        else LAUNCH_SH_BWD(MULTI_, AUX_, 16, 0);                                         \
    } while (0)
#define DISPATCH_SH_BWD(MULTI_)                                                          \
    do {                                                                                 \
        if (sh_aux) LAUNCH_SH_BWD_K(MULTI_, true); else LAUNCH_SH_BWD_K(MULTI_, false);  \
    } while (0)
        if (p->n_cams > 1) DISPATCH_SH_BWD(true);
        else DISPATCH_SH_BWD(false);
#undef DISPATCH_SH_BWD
#undef LAUNCH_SH_BWD_K
#undef LAUNCH_SH_BWD
    } else {
        int64_t rows = per_cam ? (int64_t)p->n_gauss * p->n_cams : p->n_gauss;
        hipLaunchKernelGGL(color_copy_bwd_kernel, dim3(grid_for(rows, 256)), dim3(256), 0, s, *p, K_or_D, n_color,
                           per_cam, radii, v_grec, v_coeffs_or_colors);
    }
    return check_launch();
}

int misplat_internal::gauss_bwd_sparse(const misplat_params* p, int32_t sh_degree, int32_t depth_slot, const float* means,
                                       const float* quats, const float* scales, const float* opacities, const float* viewmats,
                                       const float* Ks, const float* coeffs, const float* coeffs_rest, const float* compensations,
                                       const float* v_grec, float* v_coeffs, float* v_coeffs_rest, float* v_means, float* v_quats,
                                       float* v_scales, float* v_opacities, float* v_means2d_out, hipStream_t s,
                                       const float* v_featx, int32_t nxq, float* v_features, int32_t n_feat,
                                       int32_t depth_in_featx, bool mean_sums) {
    if (!p || p->n_gauss < 1 || p->n_cams != 1 || !p->touched || sh_degree < 0 || sh_degree > 3) return MISPLAT_EINVAL;
    if (v_features && (!v_featx || nxq < 1 || nxq > 4 || n_feat < 1 || n_feat - 1 + (depth_in_featx ? 1 : 0) > 4 * nxq ||
                       (depth_in_featx && depth_slot != -1)))
        return MISPLAT_EINVAL;
    if (!v_features && depth_in_featx) return MISPLAT_EINVAL;
    if (((uintptr_t)v_means2d_out) & 7) return MISPLAT_EINVAL;
    if (depth_slot != -1 && (depth_slot < 12 || depth_slot > 15)) return MISPLAT_EINVAL;
    if ((coeffs_rest != nullptr) != (v_coeffs_rest != nullptr)) return MISPLAT_EINVAL;
    if ((((uintptr_t)p->touched) & 7) || (((uintptr_t)coeffs | (uintptr_t)v_coeffs | (uintptr_t)v_grec | (uintptr_t)v_quats) & 15))
        return MISPLAT_EINVAL;
    // One wave per 256 rows (4 flag bytes per lane) up to 2.5 M rows, per 512 rows (8 bytes) above, at most 4 096 / 2 048
    // waves: measured on one box, 1 M: 88 -> 83 us with the finer split (twice the waves for the same three per SIMD the
    // LDS stage allows), 5 M: 66 -> 73 us with it (its rows are sparse: the wider step skips more per load).
    const bool fine = p->n_gauss < 2500000;
    const int step = fine ? 256 : 512;
    int64_t waves = ((int64_t)p->n_gauss + step - 1) / step;
    const int64_t cap = fine ? 4096 : 2048;
    if (waves > cap) waves = cap;
#define LAUNCH_SPARSE(SPLIT_, FPL_)                                                                                          \
    hipLaunchKernelGGL((gauss_bwd_sparse_kernel<SPLIT_, FPL_>), dim3((unsigned)waves), dim3(64), 0, s, *p, sh_degree, depth_slot, \
                       means, quats, scales, opacities, viewmats, Ks, coeffs, coeffs_rest, compensations, v_grec, v_coeffs,  \
                       v_coeffs_rest, v_means, v_quats, v_scales, v_opacities, (float2*)v_means2d_out, v_featx, 4 * (int)nxq, \
                       v_features, (int)n_feat, (int)depth_in_featx, mean_sums ? 1 : 0)
    if (coeffs_rest) { if (fine) LAUNCH_SPARSE(true, 4); else LAUNCH_SPARSE(true, 8); }
    else { if (fine) LAUNCH_SPARSE(false, 4); else LAUNCH_SPARSE(false, 8); }
#undef LAUNCH_SPARSE
    return check_launch();
}

extern "C" int misplat_project_pack_bwd(const misplat_params* p, int32_t depth_slot, const float* means,
                                        const float* quats, const float* scales, const float* opacities,
                                        const float* viewmats, const float* Ks, const int32_t* radii,
                                        const float* compensations, const float* v_means2d, const float* v_grec,
                                        const float* v_means_dir, float* v_means, float* v_quats,
                                        float* v_scales, float* v_opacities, const float* v_depth_rows,
                                        int32_t v_depth_stride, misplat_stream_t stream) {
    if (!p || p->n_gauss < 0 || p->n_cams < 1 || (p->activations && p->n_cams != 1)) return MISPLAT_EINVAL;   // (activations: one camera)
    if (depth_slot != -1 && (depth_slot < 12 || depth_slot > 15)) return MISPLAT_EINVAL;
    if (v_depth_rows && (v_depth_stride < 1 || depth_slot != -1)) return MISPLAT_EINVAL;
    if (p->n_gauss == 0) return MISPLAT_OK;
    if (p->n_cams == 1) {
        // one wave per workgroup, at most 2 048 of them (the kernel's 2 waves per SIMD): every wave scans enough rows to
        // fill batches of live ones
        int64_t waves = ((int64_t)p->n_gauss + 63) / 64;
        if (waves > 2048) waves = 2048;
        hipLaunchKernelGGL(project_pack_bwd_sparse_kernel, dim3((unsigned)waves), dim3(64), 0, (hipStream_t)stream, *p, depth_slot,
                           means, quats, scales, opacities, viewmats, Ks, radii, compensations, v_means2d, v_grec, v_means_dir,
                           v_means, v_quats, v_scales, v_opacities, v_depth_rows, (int)v_depth_stride);
        return check_launch();
    }
    hipLaunchKernelGGL(project_pack_bwd_kernel, dim3(grid_for(p->n_gauss, 256)), dim3(256), 0, (hipStream_t)stream,
                       *p, depth_slot, means, quats, scales, opacities, viewmats, Ks, radii, compensations, v_means2d,
                       v_grec, v_means_dir, v_means, v_quats, v_scales, v_opacities, v_depth_rows, (int)v_depth_stride);
    return check_launch();
}

extern "C" int misplat_color_fwd_x(const misplat_params* p, int32_t D, int32_t per_cam, int32_t depth_channel,
                                   int32_t nxq, const float* colors, const int32_t* radii, const float* depths,
                                   float* grec, float* featx, misplat_stream_t stream) {
    return misplat_internal::color_fwd_x(p, D, 0, per_cam, depth_channel, nxq, colors, radii, depths, grec, featx, nullptr, nullptr,
                                         (hipStream_t)stream);
}

int misplat_internal::color_fwd_x(const misplat_params* p, int32_t D, int32_t n_pre, int32_t per_cam, int32_t depth_channel,
                                  int32_t nxq, const float* colors, const int32_t* radii, const float* depths, float* grec,
                                  float* featx, float* zero_grec, float* zero_featx, hipStream_t stream) {
    if (!p || p->n_gauss < 0 || p->n_cams < 1 || nxq < 1 || nxq > 4 || D < 1 || n_pre < 0 || n_pre > 3 ||
        n_pre + D + (depth_channel ? 1 : 0) > 4 + 4 * nxq || (((uintptr_t)zero_grec | (uintptr_t)zero_featx | (uintptr_t)featx) & 15))
        return MISPLAT_EINVAL;
    int64_t total = (int64_t)p->n_gauss * p->n_cams;
    if (total == 0) return MISPLAT_OK;
    hipLaunchKernelGGL(color_copy_x_kernel, dim3(grid_for(total * (1 + nxq), 256)), dim3(256), 0, stream, *p, D, (int)n_pre,
                       per_cam, depth_channel, nxq, colors, radii, depths, grec, featx, (float4*)zero_grec, (float4*)zero_featx);
    return check_launch();
}

extern "C" int misplat_color_bwd_x(const misplat_params* p, int32_t D, int32_t per_cam, int32_t nxq,
                                   const int32_t* radii, const float* v_grec, const float* v_featx,
                                   float* v_colors, misplat_stream_t stream) {
    return misplat_internal::color_bwd_x(p, D, 0, per_cam, nxq, radii, v_grec, v_featx, v_colors, (hipStream_t)stream);
}

int misplat_internal::color_bwd_x(const misplat_params* p, int32_t D, int32_t n_pre, int32_t per_cam, int32_t nxq,
                                  const int32_t* radii, const float* v_grec, const float* v_featx, float* v_colors,
                                  hipStream_t stream) {
    if (!p || p->n_gauss < 0 || p->n_cams < 1 || nxq < 1 || nxq > 4 || D < 1 || n_pre < 0 || n_pre > 3) return MISPLAT_EINVAL;
    int64_t rows = per_cam ? (int64_t)p->n_gauss * p->n_cams : p->n_gauss;
    if (rows == 0) return MISPLAT_OK;
    hipLaunchKernelGGL(color_copy_x_bwd_kernel, dim3(grid_for(rows * ((D + 3) / 4), 256)), dim3(256), 0, stream, *p,
                       D, (int)n_pre, per_cam, nxq, radii, v_grec, v_featx, v_colors);
    return check_launch();
}
