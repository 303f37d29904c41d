// epilogue.hip -- RadegsModel.get_outputs post-processing (SURVEY.md section 8 row a3,
// /root/reference/collab_splats/models/rade_gs_model.py:221-254) as three streaming kernels instead
// of ~25 elementwise torch kernels and 4 global max reductions each way.  gfx950 only; HBM-bound
// (about 52 B read + 44 B written per pixel forward).
//
//   rgb          = clamp(render[..., :3] + (1 - alpha) * background, 0, 1)            (:227-229)
//   normals      = where(alpha > 0, (expected_normals + 1) / 2, max of that tensor)      (:221, 254)
//   depth        = where(alpha > 0, expected_depths, max(expected_depths))              (:248-250)
//   median_depth = where(alpha > 0, median_depths,  max(median_depths))                 (:251-253)
//   depth_im     = where(alpha > 0, render[..., 3:4], max(render[..., 3:4]))            (:236-240, RGB+ED)
// The maxima are over the WHOLE un-masked tensor and carry no gradient (.detach().max()).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "misplat.h"

namespace {

__device__ __forceinline__ void atomic_max_float(float* addr, float v) {
    // order-preserving integer view: works for any finite float
    if (v >= 0.f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}

// maxes[0] = max expected depth, [1] = max median depth, [2] = max (n+1)/2, [3] = max render[..., 3]
__global__ __launch_bounds__(256) void outputs_max_kernel(int64_t n_pix, int cd, const float* __restrict__ render,
                                                          const float* __restrict__ ed, const float* __restrict__ md,
                                                          const float* __restrict__ nr, float* __restrict__ maxes) {
    float m0 = -3.0e38f, m1 = -3.0e38f, m2 = -3.0e38f, m3 = -3.0e38f;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_pix; p += (int64_t)gridDim.x * blockDim.x) {
        m0 = fmaxf(m0, ed[p]);
        m1 = fmaxf(m1, md[p]);
        m2 = fmaxf(m2, fmaxf(fmaxf(nr[3 * p], nr[3 * p + 1]), nr[3 * p + 2]));
        if (cd > 3) m3 = fmaxf(m3, render[p * cd + 3]);
    }
    // wave -> block (LDS) -> ONE atomic per block and value: same-address atomics serialise at ~11 ns each
    __shared__ float red[4][4];
    m0 = wave_max_f(m0); m1 = wave_max_f(m1); m2 = wave_max_f(m2); m3 = wave_max_f(m3);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wv][0] = m0; red[wv][1] = m1; red[wv][2] = m2; red[wv][3] = m3; }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int v = threadIdx.x;
        float m = fmaxf(fmaxf(red[0][v], red[1][v]), fmaxf(red[2][v], red[3][v]));
        if (v == 2) m = (m + 1.0f) * 0.5f;
        if (v < 3 || cd > 3) atomic_max_float(maxes + v, m);
    }
}

__global__ __launch_bounds__(256) void outputs_fwd_kernel(int64_t n_pix, int cd, float bg0, float bg1, float bg2,
                                                          const float* __restrict__ render,
                                                          const float* __restrict__ alpha, const float* __restrict__ ed,
                                                          const float* __restrict__ md, const float* __restrict__ nr,
                                                          const float* __restrict__ maxes, float* __restrict__ rgb,
                                                          float* __restrict__ depth, float* __restrict__ median,
                                                          float* __restrict__ normals, float* __restrict__ depth_im) {
    const float mx0 = maxes[0], mx1 = maxes[1], mx2 = maxes[2], mx3 = maxes[3];
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_pix; p += (int64_t)gridDim.x * blockDim.x) {
        const float a = alpha[p];
        const float ia = 1.0f - a;
        const bool hit = a > 0.f;
        rgb[3 * p + 0] = fminf(fmaxf(render[p * cd + 0] + ia * bg0, 0.f), 1.f);
        rgb[3 * p + 1] = fminf(fmaxf(render[p * cd + 1] + ia * bg1, 0.f), 1.f);
        rgb[3 * p + 2] = fminf(fmaxf(render[p * cd + 2] + ia * bg2, 0.f), 1.f);
        depth[p] = hit ? ed[p] : mx0;
        median[p] = hit ? md[p] : mx1;
        normals[3 * p + 0] = hit ? (nr[3 * p + 0] + 1.0f) * 0.5f : mx2;
        normals[3 * p + 1] = hit ? (nr[3 * p + 1] + 1.0f) * 0.5f : mx2;
        normals[3 * p + 2] = hit ? (nr[3 * p + 2] + 1.0f) * 0.5f : mx2;
        if (depth_im) depth_im[p] = hit ? render[p * cd + 3] : mx3;
    }
}

__global__ __launch_bounds__(256) void outputs_bwd_kernel(int64_t n_pix, int cd, float bg0, float bg1, float bg2,
                                                          const float* __restrict__ render,
                                                          const float* __restrict__ alpha,
                                                          const float* __restrict__ v_rgb, const float* __restrict__ v_depth,
                                                          const float* __restrict__ v_median,
                                                          const float* __restrict__ v_normals,
                                                          const float* __restrict__ v_depth_im,
                                                          float* __restrict__ v_render, float* __restrict__ v_alpha,
                                                          float* __restrict__ v_ed, float* __restrict__ v_md,
                                                          float* __restrict__ v_nr) {
    const float bg[3] = {bg0, bg1, bg2};
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_pix; p += (int64_t)gridDim.x * blockDim.x) {
        const float a = alpha[p];
        const float ia = 1.0f - a;
        const bool hit = a > 0.f;
        float va = 0.f;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const float pre = render[p * cd + ch] + ia * bg[ch];
            const float g = (v_rgb && pre >= 0.f && pre <= 1.f) ? v_rgb[3 * p + ch] : 0.f;   // torch.clamp passes the edges
            v_render[p * cd + ch] = g;
            va -= g * bg[ch];
        }
        if (cd > 3) v_render[p * cd + 3] = (v_depth_im && hit) ? v_depth_im[p] : 0.f;
        v_alpha[p] = va;
        v_ed[p] = (hit && v_depth) ? v_depth[p] : 0.f;          // an absent upstream gradient (NULL) is zero
        v_md[p] = (hit && v_median) ? v_median[p] : 0.f;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) v_nr[3 * p + ch] = (hit && v_normals) ? 0.5f * v_normals[3 * p + ch] : 0.f;
    }
}

inline int grid_for(int64_t n, int block) {
    int64_t b = (n + block - 1) / block;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}
inline int check_launch() { return hipGetLastError() == hipSuccess ? MISPLAT_OK : MISPLAT_ELAUNCH; }

}  // namespace

extern "C" int misplat_outputs_fwd(int64_t n_pix, int32_t color_dim, const float* background3_host,
                                   const float* render, const float* alpha, const float* exp_depth,
                                   const float* med_depth, const float* exp_normal, float* maxes4, float* rgb,
                                   float* depth, float* median_depth, float* normals, float* depth_im,
                                   misplat_stream_t stream) {
    if (n_pix < 0 || color_dim < 3 || color_dim > 4 || !background3_host || !maxes4) return MISPLAT_EINVAL;
    if (depth_im && color_dim < 4) return MISPLAT_EINVAL;
    if (n_pix == 0) return MISPLAT_OK;
    hipStream_t s = (hipStream_t)stream;
    // seed the four maxima with -FLT_MAX on the device (a memset node: no pageable host buffer, no hidden sync)
    if (hipMemsetD32Async((hipDeviceptr_t)maxes4, 0xff7fffff, 4, s) != hipSuccess) return MISPLAT_ELAUNCH;
    const int max_grid = grid_for(n_pix, 256) < 512 ? grid_for(n_pix, 256) : 512;
    hipLaunchKernelGGL(outputs_max_kernel, dim3(max_grid), dim3(256), 0, s, n_pix, color_dim, render,
                       exp_depth, med_depth, exp_normal, maxes4);
    hipLaunchKernelGGL(outputs_fwd_kernel, dim3(grid_for(n_pix, 256)), dim3(256), 0, s, n_pix, color_dim,
                       background3_host[0], background3_host[1], background3_host[2], render, alpha, exp_depth,
                       med_depth, exp_normal, maxes4, rgb, depth, median_depth, normals, depth_im);
    return check_launch();
}

extern "C" int misplat_outputs_bwd(int64_t n_pix, int32_t color_dim, const float* background3_host,
                                   const float* render, const float* alpha, const float* v_rgb,
                                   const float* v_depth, const float* v_median_depth, const float* v_normals,
                                   const float* v_depth_im, float* v_render, float* v_alpha, float* v_exp_depth,
                                   float* v_med_depth, float* v_exp_normal, misplat_stream_t stream) {
    if (n_pix < 0 || color_dim < 3 || color_dim > 4 || !background3_host) return MISPLAT_EINVAL;
    if (n_pix == 0) return MISPLAT_OK;
    hipLaunchKernelGGL(outputs_bwd_kernel, dim3(grid_for(n_pix, 256)), dim3(256), 0, (hipStream_t)stream, n_pix,
                       color_dim, background3_host[0], background3_host[1], background3_host[2], render, alpha, v_rgb,
                       v_depth, v_median_depth, v_normals, v_depth_im, v_render, v_alpha, v_exp_depth, v_med_depth,
                       v_exp_normal);
    return check_launch();
}
