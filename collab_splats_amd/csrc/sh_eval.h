// sh_eval.h -- real spherical harmonics up to degree 3 (the 3DGS convention) evaluated term by term; shared by the colour
// kernels (project.hip) and the compositing forward's on-demand colours (blend.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace misplat_sh {

constexpr float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
__device__ constexpr float C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f};
__device__ constexpr float C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

// Colour (and, JAC, its 3x3 Jacobian d colour / d direction, J[ch][axis]) of one Gaussian from its staged
// coefficient row cf[3k + ch], one basis function at a time: a term needs its basis value, its three derivatives and
// three coefficients, nothing else stays live (12 accumulators + the shared monomials).  The array form (sh_basis into
// b / bx / by / bz[16], then the sums) made the forward colour kernel a 245-VGPR kernel with two waves per SIMD, which
// is what a streaming kernel waiting on HBM can least afford; the scheduling barriers keep the compiler from hoisting
// all 48 LDS reads and 64 basis values back to the top.  The accumulations are explicit fused multiply-adds (the same
// bits whatever the translation unit's contraction setting: the colour kernel and the on-demand evaluation in blend.hip
// must agree bit for bit); the basis expressions are evaluated without contraction in both.
// The 16 basis functions as a list: MISPLAT_SH_WALK(deg, x, y, z, T) expands T(k, B, BX, BY, BZ) -- basis value and
// its derivatives by x, y, z -- for every k of the active degree, degree band by degree band (monomials xx ... xz are in
// scope for degree >= 2), with a scheduling barrier between groups.
#define MISPLAT_SH_WALK(deg, x, y, z, T)                                                                              \
    T(0, C0, 0.f, 0.f, 0.f)                                                                                           \
    if ((deg) > 0) {                                                                                                  \
        T(1, -C1 * y, 0.f, -C1, 0.f) T(2, C1 * z, 0.f, 0.f, C1) T(3, -C1 * x, -C1, 0.f, 0.f)                          \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
    }                                                                                                                 \
    if ((deg) > 1) {                                                                                                  \
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;                           \
        T(4, C2[0] * xy, C2[0] * y, C2[0] * x, 0.f)                                                                   \
        T(5, C2[1] * yz, 0.f, C2[1] * z, C2[1] * y)                                                                   \
        T(6, C2[2] * (2.f * zz - xx - yy), -2.f * C2[2] * x, -2.f * C2[2] * y, 4.f * C2[2] * z)                       \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
        T(7, C2[3] * xz, C2[3] * z, 0.f, C2[3] * x)                                                                   \
        T(8, C2[4] * (xx - yy), 2.f * C2[4] * x, -2.f * C2[4] * y, 0.f)                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
        if ((deg) > 2) {                                                                                              \
            T(9, C3[0] * y * (3.f * xx - yy), 6.f * C3[0] * xy, C3[0] * (3.f * xx - 3.f * yy), 0.f)                   \
            T(10, C3[1] * xy * z, C3[1] * yz, C3[1] * xz, C3[1] * xy)                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            T(11, C3[2] * y * (4.f * zz - xx - yy), -2.f * C3[2] * xy, C3[2] * (4.f * zz - xx - 3.f * yy), 8.f * C3[2] * yz) \
            T(12, C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy), -6.f * C3[3] * xz, -6.f * C3[3] * yz,                 \
              C3[3] * (6.f * zz - 3.f * xx - 3.f * yy))                                                               \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            T(13, C3[4] * x * (4.f * zz - xx - yy), C3[4] * (4.f * zz - 3.f * xx - yy), -2.f * C3[4] * xy, 8.f * C3[4] * xz) \
            T(14, C3[5] * z * (xx - yy), 2.f * C3[5] * xz, -2.f * C3[5] * yz, C3[5] * (xx - yy))                      \
            __builtin_amdgcn_sched_barrier(0);                                                                        \
            T(15, C3[6] * x * (xx - 3.f * yy), C3[6] * (3.f * xx - 3.f * yy), -6.f * C3[6] * xy, 0.f)                 \
        }                                                                                                             \
    }

template <bool JAC>
__device__ __forceinline__ void sh_eval(int deg, float x, float y, float z, const float* cf, float& c0, float& c1, float& c2,
                                        float (&J)[9]) {
#define SH_TERM(k, B, BX, BY, BZ)                                                              \
    {                                                                                          \
        const float f0 = cf[3 * (k)], f1 = cf[3 * (k) + 1], f2 = cf[3 * (k) + 2];              \
        const float b_ = (B);                                                                  \
        c0 = fmaf(b_, f0, c0); c1 = fmaf(b_, f1, c1); c2 = fmaf(b_, f2, c2);                   \
        if (JAC) {                                                                             \
            const float bx_ = (BX), by_ = (BY), bz_ = (BZ);                                    \
            J[0] = fmaf(bx_, f0, J[0]); J[1] = fmaf(by_, f0, J[1]); J[2] = fmaf(bz_, f0, J[2]); \
            J[3] = fmaf(bx_, f1, J[3]); J[4] = fmaf(by_, f1, J[4]); J[5] = fmaf(bz_, f1, J[5]); \
            J[6] = fmaf(bx_, f2, J[6]); J[7] = fmaf(by_, f2, J[7]); J[8] = fmaf(bz_, f2, J[8]); \
        }                                                                                      \
    }
    MISPLAT_SH_WALK(deg, x, y, z, SH_TERM)
#undef SH_TERM
}

// Backward of one evaluation from the coefficients themselves (no cached Jacobian): vc = the gradient of the three
// colour channels after the clamp.  Term k: s = coefficient_k . vc feeds the direction gradient through the basis
// derivatives, and the coefficient's own gradient is b_k * vc -- written over the coefficient in `cf` (ACC: added into
// acc[3k + ch] instead, several cameras).  One term at a time, as sh_eval.
template <bool ACC>
__device__ __forceinline__ void sh_grad(int deg, float x, float y, float z, float* cf, float vc0, float vc1, float vc2,
                                        float& vd0, float& vd1, float& vd2, float* acc) {
#define SH_TERM(k, B, BX, BY, BZ)                                                              \
    {                                                                                          \
        const float s_ = cf[3 * (k)] * vc0 + cf[3 * (k) + 1] * vc1 + cf[3 * (k) + 2] * vc2;    \
        const float b_ = (B);                                                                  \
        vd0 += (BX) * s_; vd1 += (BY) * s_; vd2 += (BZ) * s_;                                  \
        if (ACC) { acc[3 * (k)] += b_ * vc0; acc[3 * (k) + 1] += b_ * vc1; acc[3 * (k) + 2] += b_ * vc2; } \
        else { cf[3 * (k)] = b_ * vc0; cf[3 * (k) + 1] = b_ * vc1; cf[3 * (k) + 2] = b_ * vc2; } \
    }
    MISPLAT_SH_WALK(deg, x, y, z, SH_TERM)
#undef SH_TERM
}

}  // namespace misplat_sh
