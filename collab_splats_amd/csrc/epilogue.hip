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
#include "internal.h"

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
// ---- a5 get_loss_dict (rade_gs_model.py:289-307 + the base model's L1 term): three image means and their backward.
// Forward: kLossBlocks workgroups leave (sum |gt - rgb|, sum err_exp, sum err_med) per workgroup -- fixed grid, fixed
// tree, so the values are reproducible bit for bit --, one workgroup adds the partial sums in fp64 and writes the two
// loss values.  Backward: the three constant-magnitude gradient images in one pass.
constexpr int kLossBlocks = 512;

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

__global__ __launch_bounds__(256) void loss_partial_kernel(int64_t n_pix, const float* __restrict__ rgb,
                                                           const float* __restrict__ gt, const float* __restrict__ err_exp,
                                                           const float* __restrict__ err_med, float* __restrict__ partials) {
    __shared__ float sm[3][4];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
    if (gt) {
        const int64_t n3 = 3 * n_pix, n4 = n3 >> 2;                       // (both images 16-byte aligned: launcher)
        const float4* a4 = reinterpret_cast<const float4*>(rgb);
        const float4* b4 = reinterpret_cast<const float4*>(gt);
        for (int64_t i = tid; i < n4; i += stride) {
            const float4 a = a4[i], b = b4[i];
            s0 += (fabsf(b.x - a.x) + fabsf(b.y - a.y)) + (fabsf(b.z - a.z) + fabsf(b.w - a.w));
        }
        if (tid < (n3 & 3)) s0 += fabsf(gt[4 * n4 + tid] - rgb[4 * n4 + tid]);
    }
    if (err_exp)
        for (int64_t i = tid; i < n_pix; i += stride) { s1 += err_exp[i]; s2 += err_med[i]; }
    s0 = wave_sum_f(s0); s1 = wave_sum_f(s1); s2 = wave_sum_f(s2);
    if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.x >> 6] = s0; sm[1][threadIdx.x >> 6] = s1; sm[2][threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x < 3)
        partials[3 * blockIdx.x + threadIdx.x] = (sm[threadIdx.x][0] + sm[threadIdx.x][1]) + (sm[threadIdx.x][2] + sm[threadIdx.x][3]);
}

__global__ __launch_bounds__(256) void loss_final_kernel(int n_blocks, const float* __restrict__ partials, double inv_n3, double inv_n,
                                                         float depth_ratio, float lambda, float* __restrict__ rgb_loss,
                                                         float* __restrict__ dn_loss) {
    __shared__ double sm[3][256];
    double s[3] = {0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < n_blocks; b += 256)
        for (int k = 0; k < 3; k++) s[k] += (double)partials[3 * b + k];
    for (int k = 0; k < 3; k++) sm[k][threadIdx.x] = s[k];
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w)
            for (int k = 0; k < 3; k++) sm[k][threadIdx.x] += sm[k][threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (rgb_loss) *rgb_loss = (float)(sm[0][0] * inv_n3);
        if (dn_loss) {
            const float m_exp = (float)(sm[1][0] * inv_n), m_med = (float)(sm[2][0] * inv_n);
            *dn_loss = lambda * ((1.0f - depth_ratio) * m_exp + depth_ratio * m_med);
        }
    }
}

// v_rgb = -g_rgb / (3 n) * sign(gt - rgb) (torch's abs backward: sign(0) = 0); v_err_exp = g_dn lambda (1 - r) / n,
// v_err_med = g_dn lambda r / n.  g_*: device scalars (the upstream gradients of the two loss values), or NULL = 0.
__global__ __launch_bounds__(256) void loss_bwd_kernel(int64_t n_pix, const float* __restrict__ rgb, const float* __restrict__ gt,
                                                       const float* __restrict__ g_rgb, const float* __restrict__ g_dn,
                                                       float inv_n3, float w_exp, float w_med, float* __restrict__ v_rgb,
                                                       float* __restrict__ v_err_exp, float* __restrict__ v_err_med) {
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
    if (v_rgb) {
        const float g = g_rgb ? -(*g_rgb) * inv_n3 : 0.f;
        const int64_t n3 = 3 * n_pix;
        for (int64_t i = tid; i < n3; i += stride) {
            const float d = gt[i] - rgb[i];
            v_rgb[i] = d > 0.f ? g : (d < 0.f ? -g : 0.f);
        }
    }
    if (v_err_exp) {
        const float gd = g_dn ? *g_dn : 0.f;
        const float a = gd * w_exp, b = gd * w_med;
        for (int64_t i = tid; i < n_pix; i += stride) { v_err_exp[i] = a; v_err_med[i] = b; }
    }
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
    // seed the four maxima with -FLT_MAX on the device (a fill KERNEL: no pageable host buffer, no hidden sync, and no
    // memset node in a graph that captures this call)
    if (misplat_internal::fill_bytes(maxes4, 4 * sizeof(float), 0xff7fffffu, s) != MISPLAT_OK) return MISPLAT_ELAUNCH;
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

extern "C" int misplat_loss_fwd(int64_t n_pix, const float* rgb, const float* gt, const float* err_exp, const float* err_med,
                                float depth_ratio, float depth_normal_lambda, float* partials, float* rgb_loss, float* dn_loss,
                                misplat_stream_t stream) {
    if (n_pix < 1 || !partials || (gt && (!rgb || !rgb_loss)) || (err_exp && (!err_med || !dn_loss)) || (!gt && !err_exp))
        return MISPLAT_EINVAL;
    if (gt && ((((uintptr_t)rgb) | ((uintptr_t)gt)) & 15)) return MISPLAT_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(loss_partial_kernel, dim3(kLossBlocks), dim3(256), 0, s, n_pix, rgb, gt, err_exp, err_med, partials);
    hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, s, kLossBlocks, partials, 1.0 / (3.0 * (double)n_pix),
                       1.0 / (double)n_pix, depth_ratio, depth_normal_lambda, gt ? rgb_loss : nullptr, err_exp ? dn_loss : nullptr);
    return check_launch();
}

extern "C" int misplat_loss_bwd(int64_t n_pix, const float* rgb, const float* gt, const float* g_rgb_loss, const float* g_dn_loss,
                                float depth_ratio, float depth_normal_lambda, float* v_rgb, float* v_err_exp, float* v_err_med,
                                misplat_stream_t stream) {
    if (n_pix < 1 || (v_rgb && (!rgb || !gt)) || ((v_err_exp != nullptr) != (v_err_med != nullptr)) || (!v_rgb && !v_err_exp))
        return MISPLAT_EINVAL;
    int64_t blocks = (3 * n_pix + 1023) / 1024;
    if (blocks > 4096) blocks = 4096;
    const float inv_n = (float)(1.0 / (double)n_pix);
    hipLaunchKernelGGL(loss_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n_pix, rgb, gt, g_rgb_loss,
                       g_dn_loss, (float)(1.0 / (3.0 * (double)n_pix)), depth_normal_lambda * (1.0f - depth_ratio) * inv_n,
                       depth_normal_lambda * depth_ratio * inv_n, v_rgb, v_err_exp, v_err_med);
    return check_launch();
}
