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

// ---- shared-Gaussian gradient reduce that moves only the rows that HAVE a gradient (parallel.GradientBuckets; SURVEY.md
// section 8(e) "sparse alternative") -------------------------------------------------------------------------------------
// Data-parallel training sums 236 B per Gaussian over the ranks (the gradient set of
// /root/reference/collab_splats/configs/rade_gs_method.py:44-71), but a view's backward reaches few of them: 11 % of the
// rows of the 1 M scene, 2 % at 5 M (the compositing stops at the first opaque layers), and the compositing backward already
// flags them (misplat_params.touched).  So: every rank turns its flags into a bitmap (N / 8 bytes), the bitmaps are
// all-gathered and OR-ed, the union's row ids are listed once (the same list on every rank), the rows are packed into
// [|union|, W] floats, reduced, and scattered back.  All streaming; the host reads one number (|union|).
namespace {

constexpr int kBitsBlock = 256;                           // bitmap bytes per workgroup = 2 048 rows

__global__ __launch_bounds__(256) void touched_bits_kernel(const uint8_t* __restrict__ flags, int64_t n, uint8_t* __restrict__ bits,
                                                           int64_t nbytes) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= nbytes) return;
    const int64_t r0 = b * 8;
    uint32_t v = 0u;
    if (r0 + 8 <= n) {
        const uint2 w = *reinterpret_cast<const uint2*>(flags + r0);       // 8 flags (the flag array is 8-byte aligned)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            v |= ((w.x >> (8 * k)) & 0xffu) ? (1u << k) : 0u;
            v |= ((w.y >> (8 * k)) & 0xffu) ? (1u << (4 + k)) : 0u;
        }
    } else {
        for (int k = 0; k < 8; k++)
            if (r0 + k < n && flags[r0 + k]) v |= 1u << k;
    }
    bits[b] = (uint8_t)v;
}

__device__ __forceinline__ uint32_t union_byte(const uint8_t* __restrict__ gathered, int world, int64_t nbytes, int64_t b) {
    uint32_t v = 0u;
    if (b < nbytes)
        for (int w = 0; w < world; w++) v |= gathered[(int64_t)w * nbytes + b];
    return v;
}

__device__ __forceinline__ uint32_t block_scan_excl(uint32_t x, uint32_t* wsum, uint32_t& total) {   // 256 threads
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = x;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t before = 0u;
    total = 0u;
#pragma unroll
    for (int w = 0; w < 4; w++) { before += (w < wave) ? wsum[w] : 0u; total += wsum[w]; }
    return before + incl - x;
}

__global__ __launch_bounds__(256) void union_count_kernel(const uint8_t* __restrict__ gathered, int world, int64_t nbytes,
                                                          int32_t* __restrict__ block_counts) {
    __shared__ uint32_t wsum[4];
    const uint32_t v = union_byte(gathered, world, nbytes, (int64_t)blockIdx.x * kBitsBlock + threadIdx.x);
    uint32_t total;
    (void)block_scan_excl((uint32_t)__popc(v), wsum, total);
    if (threadIdx.x == 0) block_counts[blockIdx.x] = (int32_t)total;
}

__global__ __launch_bounds__(256) void union_ids_kernel(const uint8_t* __restrict__ gathered, int world, int64_t nbytes,
                                                        const int64_t* __restrict__ block_offsets, int32_t* __restrict__ ids,
                                                        int64_t ids_cap) {
    __shared__ uint32_t wsum[4];
    const int64_t b = (int64_t)blockIdx.x * kBitsBlock + threadIdx.x;
    uint32_t v = union_byte(gathered, world, nbytes, b);
    uint32_t total;
    int64_t pos = block_offsets[blockIdx.x] + (int64_t)block_scan_excl((uint32_t)__popc(v), wsum, total);
    while (v) {
        const int k = __ffs((int)v) - 1;
        if (pos < ids_cap) ids[pos] = (int32_t)(b * 8 + k);     // (a speculative capacity may be short: the caller checks the count)
        pos++;
        v &= v - 1u;
    }
}

// Exclusive scan of the per-block counts (int64 offsets) and their total, by ONE workgroup: a few thousand counts (one per 2 048
// rows) -- a library scan call here costs the host more than the whole reduce's kernels (measured: the first torch op behind
// the backward's launch blocked the autograd thread for 0.9 ms per step at 5 M Gaussians).
__global__ __launch_bounds__(1024) void union_scan_kernel(const int32_t* __restrict__ counts, int64_t n_blocks,
                                                          int64_t* __restrict__ offsets, int64_t* __restrict__ total) {
    __shared__ unsigned long long wsum[16];
    __shared__ unsigned long long carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0ull;
    __syncthreads();
    for (int64_t base = 0; base < n_blocks; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const unsigned long long x = i < n_blocks ? (unsigned long long)(uint32_t)counts[i] : 0ull;
        unsigned long long incl = x;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long o = __shfl_up(incl, off);
            if (lane >= off) incl += o;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        unsigned long long before = carry_s, all = 0ull;
#pragma unroll
        for (int w = 0; w < 16; w++) { before += (w < wave) ? wsum[w] : 0ull; all += wsum[w]; }
        if (i < n_blocks) offsets[i] = (int64_t)(before + incl - x);
        __syncthreads();
        if (threadIdx.x == 0) carry_s += all;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = (int64_t)carry_s;
}

struct RowTable {
    float* p[MISPLAT_ROWS_MAX_TENSORS];
    int32_t width[MISPLAT_ROWS_MAX_TENSORS];
    int32_t first[MISPLAT_ROWS_MAX_TENSORS + 1];          // column range of every tensor inside a packed row
    int32_t count, W;
};

template <bool PACK>
__global__ __launch_bounds__(256) void rows_move_kernel(RowTable T, const int32_t* __restrict__ ids, int64_t n_ids,
                                                        const int64_t* __restrict__ count_dev, float* __restrict__ packed) {
    // count_dev (or NULL = n_ids): how many of the n_ids slots hold a row -- known on the device only (the union's size; the
    // host sized the buffers from the previous step's).  PACK: the slots behind it are zeros (they are reduced with the
    // rest).  UNPACK: nothing at all when the count exceeds n_ids (the caller then reduces the dense buffer, which must be
    // as the backward left it).
    const int64_t have = count_dev ? *count_dev : n_ids;
    if (!PACK && have > n_ids) return;
    const int64_t total = n_ids * T.W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t u = i / T.W;
        if (u >= have) {
            if (PACK) packed[i] = 0.f;
            continue;
        }
        const int c = (int)(i - u * T.W);
        int t = 0;
#pragma unroll
        for (int k = 1; k < MISPLAT_ROWS_MAX_TENSORS; k++)
            if (k < T.count && c >= T.first[k]) t = k;
        float* q = T.p[t] + (int64_t)ids[u] * T.width[t] + (c - T.first[t]);
        if (PACK) packed[i] = *q;
        else *q = packed[i];
    }
}

int rows_move(bool pack, int32_t n_tensors, float* const* tensors, const int32_t* widths, const int32_t* ids, int64_t n_ids,
              const int64_t* count_dev, float* packed, hipStream_t s) {
    if (n_tensors < 1 || n_tensors > MISPLAT_ROWS_MAX_TENSORS || !tensors || !widths || n_ids < 0) return MISPLAT_EINVAL;
    if (n_ids == 0) return MISPLAT_OK;
    if (!ids || !packed) return MISPLAT_EINVAL;
    RowTable T;
    int W = 0;
    for (int i = 0; i < n_tensors; i++) {
        if (!tensors[i] || widths[i] < 1) return MISPLAT_EINVAL;
        T.p[i] = tensors[i]; T.width[i] = widths[i]; T.first[i] = W;
        W += widths[i];
    }
    for (int i = n_tensors; i <= MISPLAT_ROWS_MAX_TENSORS; i++) T.first[i] = W;
    for (int i = n_tensors; i < MISPLAT_ROWS_MAX_TENSORS; i++) { T.p[i] = nullptr; T.width[i] = 0; }
    T.count = n_tensors; T.W = W;
    int64_t blocks = (n_ids * W + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (pack) hipLaunchKernelGGL(rows_move_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, T, ids, n_ids, count_dev, packed);
    else hipLaunchKernelGGL(rows_move_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, T, ids, n_ids, count_dev, packed);
    return hipGetLastError() == hipSuccess ? MISPLAT_OK : MISPLAT_ELAUNCH;
}

}  // namespace

extern "C" int misplat_touched_bits(const uint8_t* flags, int64_t n_rows, uint8_t* bits, misplat_stream_t stream) {
    if (n_rows < 0 || (n_rows > 0 && (!flags || !bits)) || (((uintptr_t)flags) & 7)) return MISPLAT_EINVAL;
    const int64_t nbytes = (n_rows + 7) / 8;
    if (nbytes == 0) return MISPLAT_OK;
    hipLaunchKernelGGL(touched_bits_kernel, dim3((unsigned)((nbytes + 255) / 256)), dim3(256), 0, (hipStream_t)stream, flags, n_rows,
                       bits, nbytes);
    return hipGetLastError() == hipSuccess ? MISPLAT_OK : MISPLAT_ELAUNCH;
}

extern "C" int misplat_union_count(const uint8_t* gathered, int32_t world, int64_t nbytes, int32_t* block_counts,
                                   misplat_stream_t stream) {
    if (world < 1 || nbytes < 0 || (nbytes > 0 && (!gathered || !block_counts))) return MISPLAT_EINVAL;
    if (nbytes == 0) return MISPLAT_OK;
    hipLaunchKernelGGL(union_count_kernel, dim3((unsigned)((nbytes + kBitsBlock - 1) / kBitsBlock)), dim3(256), 0,
                       (hipStream_t)stream, gathered, (int)world, nbytes, block_counts);
    return hipGetLastError() == hipSuccess ? MISPLAT_OK : MISPLAT_ELAUNCH;
}

extern "C" int misplat_union_scan(const int32_t* block_counts, int64_t n_blocks, int64_t* block_offsets, int64_t* total,
                                  misplat_stream_t stream) {
    if (n_blocks < 0 || !total || (n_blocks > 0 && (!block_counts || !block_offsets))) return MISPLAT_EINVAL;
    hipLaunchKernelGGL(union_scan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, block_counts, n_blocks, block_offsets, total);
    return hipGetLastError() == hipSuccess ? MISPLAT_OK : MISPLAT_ELAUNCH;
}

extern "C" int misplat_union_ids(const uint8_t* gathered, int32_t world, int64_t nbytes, const int64_t* block_offsets,
                                 int32_t* ids, int64_t ids_cap, misplat_stream_t stream) {
    if (world < 1 || nbytes < 0 || ids_cap < 0 || (nbytes > 0 && (!gathered || !block_offsets || !ids))) return MISPLAT_EINVAL;
    if (nbytes == 0) return MISPLAT_OK;
    hipLaunchKernelGGL(union_ids_kernel, dim3((unsigned)((nbytes + kBitsBlock - 1) / kBitsBlock)), dim3(256), 0,
                       (hipStream_t)stream, gathered, (int)world, nbytes, block_offsets, ids, ids_cap);
    return hipGetLastError() == hipSuccess ? MISPLAT_OK : MISPLAT_ELAUNCH;
}

extern "C" int misplat_rows_pack(int32_t n_tensors, const float* const* srcs, const int32_t* widths, const int32_t* ids,
                                 int64_t n_ids, const int64_t* count_dev, float* packed, misplat_stream_t stream) {
    return rows_move(true, n_tensors, (float* const*)srcs, widths, ids, n_ids, count_dev, packed, (hipStream_t)stream);
}

extern "C" int misplat_rows_unpack(int32_t n_tensors, float* const* dsts, const int32_t* widths, const int32_t* ids,
                                   int64_t n_ids, const int64_t* count_dev, const float* packed, misplat_stream_t stream) {
    return rows_move(false, n_tensors, dsts, widths, ids, n_ids, count_dev, (float*)packed, (hipStream_t)stream);
}
