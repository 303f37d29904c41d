// optim.hip -- fused multi-tensor Adam for the six Gaussian parameter groups (SURVEY.md section 8(f) rank 3:
// the step on the other side of the rasterizer; the reference configures one torch Adam per group,
// /root/reference/collab_splats/configs/rade_gs_method.py:44-71, eps = 1e-15, no weight decay, no amsgrad).
// gfx950 only.  Pure HBM streaming: per element 16 B read (param, grad, exp_avg, exp_avg_sq) + 12 B written;
// one launch covers up to MISPLAT_ADAM_MAX_TENSORS tensors, each with its own learning rate and step count.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "misplat.h"

namespace {

struct AdamTable {
    float* p[MISPLAT_ADAM_MAX_TENSORS];
    const float* g[MISPLAT_ADAM_MAX_TENSORS];
    float* m[MISPLAT_ADAM_MAX_TENSORS];
    float* v[MISPLAT_ADAM_MAX_TENSORS];
    int64_t n[MISPLAT_ADAM_MAX_TENSORS];
    int64_t first_block[MISPLAT_ADAM_MAX_TENSORS + 1];   // block range of every tensor
    float step_size[MISPLAT_ADAM_MAX_TENSORS];           // lr / (1 - beta1^t)
    float sqrt_bias2[MISPLAT_ADAM_MAX_TENSORS];          // sqrt(1 - beta2^t)
    int count;
};

constexpr int kBlock = 256;
constexpr int kPerThread = 8;                             // 2 x float4
constexpr int64_t kChunk = (int64_t)kBlock * kPerThread;

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float omb1, float b2, float omb2, float eps,
                                         float step_size, float sb2) {
    // torch.optim.Adam (single-tensor path): exp_avg.lerp_(grad, 1 - beta1); exp_avg_sq = beta2 * v + (1 - beta2) g^2;
    // denom = sqrt(exp_avg_sq) / sqrt(bias2) + eps; p -= step_size * exp_avg / denom
    // 1 - beta is rounded from double on the host, as torch does (python floats), not computed in fp32
    m = m + (g - m) * omb1;
    v = b2 * v + omb2 * g * g;
    const float denom = sqrtf(v) / sb2 + eps;
    p = p - step_size * (m / denom);
}

__global__ __launch_bounds__(kBlock) void adam_kernel(AdamTable T, float omb1, float b2, float omb2, float eps) {
    // which tensor does this block belong to (<= 8 entries: linear search in SGPRs)
    int t = 0;
#pragma unroll
    for (int k = 1; k < MISPLAT_ADAM_MAX_TENSORS; k++)
        if (k < T.count && (int64_t)blockIdx.x >= T.first_block[k]) t = k;
    const int64_t base = ((int64_t)blockIdx.x - T.first_block[t]) * kChunk;
    const int64_t n = T.n[t];
    float* __restrict__ p = T.p[t];
    const float* __restrict__ g = T.g[t];
    float* __restrict__ m = T.m[t];
    float* __restrict__ v = T.v[t];
    const float ss = T.step_size[t], isb2 = T.sqrt_bias2[t];
#pragma unroll
    for (int u = 0; u < kPerThread / 4; u++) {
        const int64_t i = base + ((int64_t)u * kBlock + threadIdx.x) * 4;
        if (i + 3 < n) {
            float4 pp = *reinterpret_cast<float4*>(p + i);
            const float4 gg = *reinterpret_cast<const float4*>(g + i);
            float4 mm = *reinterpret_cast<float4*>(m + i);
            float4 vv = *reinterpret_cast<float4*>(v + i);
            adam_one(pp.x, gg.x, mm.x, vv.x, omb1, b2, omb2, eps, ss, isb2);
            adam_one(pp.y, gg.y, mm.y, vv.y, omb1, b2, omb2, eps, ss, isb2);
            adam_one(pp.z, gg.z, mm.z, vv.z, omb1, b2, omb2, eps, ss, isb2);
            adam_one(pp.w, gg.w, mm.w, vv.w, omb1, b2, omb2, eps, ss, isb2);
            *reinterpret_cast<float4*>(p + i) = pp;
            *reinterpret_cast<float4*>(m + i) = mm;
            *reinterpret_cast<float4*>(v + i) = vv;
        } else {
            for (int64_t j = i; j < n && j < i + 4; j++) {
                float pp = p[j], mm = m[j], vv = v[j];
                adam_one(pp, g[j], mm, vv, omb1, b2, omb2, eps, ss, isb2);
                p[j] = pp; m[j] = mm; v[j] = vv;
            }
        }
    }
}

}  // namespace

extern "C" int misplat_adam_step(int32_t n_tensors, float* const* params, const float* const* grads,
                                 float* const* exp_avg, float* const* exp_avg_sq, const int64_t* numel,
                                 const float* lr, const int64_t* step, double beta1, double beta2, double eps,
                                 misplat_stream_t stream) {
    if (n_tensors < 0 || n_tensors > MISPLAT_ADAM_MAX_TENSORS) return MISPLAT_EINVAL;
    if (n_tensors == 0) return MISPLAT_OK;
    if (!params || !grads || !exp_avg || !exp_avg_sq || !numel || !lr || !step) return MISPLAT_EINVAL;
    AdamTable T;
    int64_t blocks = 0;
    int k = 0;
    for (int i = 0; i < n_tensors; i++) {
        if (numel[i] < 0 || step[i] < 1) return MISPLAT_EINVAL;
        if (numel[i] == 0) continue;
        if (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i]) return MISPLAT_EINVAL;
        // the float4 path needs 16-byte alignment (torch allocations are 512-byte aligned)
        if ((((uintptr_t)params[i]) | ((uintptr_t)grads[i]) | ((uintptr_t)exp_avg[i]) | ((uintptr_t)exp_avg_sq[i])) & 15)
            return MISPLAT_EINVAL;
        T.p[k] = params[i]; T.g[k] = grads[i]; T.m[k] = exp_avg[i]; T.v[k] = exp_avg_sq[i];
        T.n[k] = numel[i];
        T.first_block[k] = blocks;
        const double bias1 = 1.0 - pow(beta1, (double)step[i]);
        const double bias2 = 1.0 - pow(beta2, (double)step[i]);
        T.step_size[k] = (float)((double)lr[i] / bias1);
        T.sqrt_bias2[k] = (float)sqrt(bias2);
        blocks += (numel[i] + kChunk - 1) / kChunk;
        k++;
    }
    if (k == 0) return MISPLAT_OK;
    for (int i = k; i <= MISPLAT_ADAM_MAX_TENSORS; i++) T.first_block[i] = blocks;
    T.count = k;
    if (blocks > 0x7fffffffLL) return MISPLAT_EINVAL;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, T,
                       (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps);
    return hipGetLastError() == hipSuccess ? MISPLAT_OK : MISPLAT_ELAUNCH;
}
