// sort.hip -- hand-written stable LSD radix sort of (uint32 key, int32 value) pairs for gfx950.
//
// Used by the two ordering stages of the binning step: Gaussian rows
// by depth bits (32 bits), intersections by tile id (13 bits at 1080p).  HBM-bound integer work: no
// MFMA; what matters is coalesced reads, per-wave LDS counters and no cross-workgroup spinning.
//
// One pass over `bits` key bits = three launches, none of which waits on another workgroup
// (dispatch order and XCD placement are irrelevant to correctness):
//   radix_hist     every WAVE owns a contiguous chunk of CH elements and counts its digits in its
//                  own LDS table -> hist[chunk][bin]
//   radix_scan     one workgroup per bin: exclusive prefix of that bin's counts over the chunks
//                  (-> prefix[chunk][bin]) and the bin total
//   radix_scatter  every wave rebuilds base[bin] = (exclusive scan of totals)[bin] + prefix[chunk][bin]
//                  in LDS, then walks its chunk 64 elements at a time IN ORDER: lanes holding the
//                  same digit find each other with one __ballot per key bit, rank = popcount of the
//                  lower peers, position = base[digit] + rank, and the first peer advances base.
// Stability: chunks are contiguous and ordered, the prefix is over ascending chunk index, rounds are
// processed in order and the in-round rank is by lane index -- equal digits keep their input order.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "misplat.h"

namespace {

constexpr int kMaxBits = 11;             // <= 2048 bins: 8 KiB of LDS counters per wave
constexpr int kWavesPerBlock = 4;

__device__ __forceinline__ uint32_t digit_of(uint32_t key, int shift, uint32_t mask) { return (key >> shift) & mask; }

__global__ __launch_bounds__(64 * kWavesPerBlock) void radix_hist_kernel(const uint32_t* __restrict__ keys, int64_t n,
                                                                         int chunk, int n_chunks, int shift, int bits,
                                                                         uint32_t* __restrict__ hist) {
    extern __shared__ uint32_t lds[];
    const int nbins = 1 << bits;
    const uint32_t mask = (uint32_t)nbins - 1u;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = blockIdx.x * kWavesPerBlock + wave;
    uint32_t* tab = lds + wave * nbins;
    for (int b = lane; b < nbins; b += 64) tab[b] = 0u;
    __syncthreads();
    if (c < n_chunks) {
        const int64_t beg = (int64_t)c * chunk;
        const int64_t end = beg + chunk < n ? beg + chunk : n;
        for (int64_t i = beg + lane; i < end; i += 64) atomicAdd(&tab[digit_of(keys[i], shift, mask)], 1u);
    }
    __syncthreads();
    if (c < n_chunks)
        for (int b = lane; b < nbins; b += 64) hist[(size_t)c * nbins + b] = tab[b];
}

// grid = nbins / 4 workgroups; workgroup q scans the four adjacent columns 4q..4q+3 of hist over the
// chunks with 16-byte accesses (a quarter of a 64-byte sector instead of a sixteenth).
__global__ __launch_bounds__(256) void radix_scan_kernel(const uint32_t* __restrict__ hist, int n_chunks, int nbins,
                                                         uint32_t* __restrict__ prefix, uint32_t* __restrict__ totals) {
    __shared__ uint4 part[256];
    const int q = blockIdx.x;
    const int t = threadIdx.x;
    const int per = (n_chunks + 255) / 256;                 // consecutive chunks per thread
    const int c0 = t * per, c1 = min(c0 + per, n_chunks);
    uint4 sum = make_uint4(0, 0, 0, 0);
    for (int c = c0; c < c1; c++) {
        const uint4 h = *reinterpret_cast<const uint4*>(hist + (size_t)c * nbins + 4 * q);
        sum.x += h.x; sum.y += h.y; sum.z += h.z; sum.w += h.w;
    }
    part[t] = sum;
    __syncthreads();
    uint4 v = sum;                                          // inclusive Hillis-Steele scan over the 256 threads
    for (int off = 1; off < 256; off <<= 1) {
        uint4 add = make_uint4(0, 0, 0, 0);
        if (t >= off) add = part[t - off];
        __syncthreads();
        v.x += add.x; v.y += add.y; v.z += add.z; v.w += add.w;
        part[t] = v;
        __syncthreads();
    }
    uint4 run = make_uint4(v.x - sum.x, v.y - sum.y, v.z - sum.z, v.w - sum.w);
    for (int c = c0; c < c1; c++) {
        const uint4 h = *reinterpret_cast<const uint4*>(hist + (size_t)c * nbins + 4 * q);
        *reinterpret_cast<uint4*>(prefix + (size_t)c * nbins + 4 * q) = run;
        run.x += h.x; run.y += h.y; run.z += h.z; run.w += h.w;
    }
    if (t == 255) *reinterpret_cast<uint4*>(totals + 4 * q) = v;
}

__global__ __launch_bounds__(64 * kWavesPerBlock) void radix_scatter_kernel(
    const uint32_t* __restrict__ keys_in, const int32_t* __restrict__ vals_in, int64_t n, int chunk, int n_chunks,
    int shift, int bits, const uint32_t* __restrict__ prefix, const uint32_t* __restrict__ totals,
    uint32_t* __restrict__ keys_out, int32_t* __restrict__ vals_out) {
    extern __shared__ uint32_t lds[];
    const int nbins = 1 << bits;
    const uint32_t mask = (uint32_t)nbins - 1u;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = blockIdx.x * kWavesPerBlock + wave;
    uint32_t* base = lds + wave * nbins;
    if (c >= n_chunks) return;                               // no barrier below: waves are independent
    // base[b] = exclusive scan of totals + this chunk's prefix.  Lane l owns bins [l*per, (l+1)*per).
    {
        const int per = (nbins + 63) / 64;
        const int b0 = lane * per, b1 = min(b0 + per, nbins);
        uint32_t s = 0;
        for (int b = b0; b < b1; b++) s += totals[b];
        uint32_t incl = s;                                    // inclusive wave scan of the lane sums
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off);
            if (lane >= off) incl += o;
        }
        uint32_t run = incl - s;
        for (int b = b0; b < b1; b++) {
            base[b] = run + prefix[(size_t)c * nbins + b];
            run += totals[b];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    const int64_t beg = (int64_t)c * chunk;
    const int64_t end = beg + chunk < n ? beg + chunk : n;
    const unsigned long long lt = (1ull << lane) - 1ull;
    constexpr int U = 4;                                      // rounds whose loads are issued together
    for (int64_t r0 = beg; r0 < end; r0 += 64 * U) {
        uint32_t kk[U];
        int32_t vv[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t i = r0 + u * 64 + lane;
            kk[u] = i < end ? keys_in[i] : 0u;
            vv[u] = i < end ? vals_in[i] : 0;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t i = r0 + u * 64 + lane;
            const bool valid = i < end;
            const uint32_t dg = digit_of(kk[u], shift, mask);
            unsigned long long peers = __ballot(valid);
            if (peers == 0ull) break;
            for (int bit = 0; bit < bits; bit++) {
                const bool on = (dg >> bit) & 1u;
                const unsigned long long b = __ballot(on);
                peers &= on ? b : ~b;
            }
            const int rank = __popcll(peers & lt);
            uint32_t pos = 0;
            if (valid) pos = base[dg] + (uint32_t)rank;
            __builtin_amdgcn_wave_barrier();                  // all lanes have read base before it moves
            if (valid && rank == 0) base[dg] += (uint32_t)__popcll(peers);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            if (valid) { keys_out[pos] = kk[u]; vals_out[pos] = vv[u]; }
        }
    }
}

inline int check_launch() { return hipGetLastError() == hipSuccess ? MISPLAT_OK : MISPLAT_ELAUNCH; }
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
inline int chunk_for(int64_t n) { return n <= (int64_t)2 << 20 ? 256 : 512; }

struct Plan {
    int chunk, n_chunks, n_pass, max_bins;
    size_t off_tk, off_tv, off_hist, off_prefix, off_totals, total;
};

inline Plan make_plan(int64_t n, int begin_bit, int end_bit, int bits_per_pass) {
    Plan p;
    p.chunk = chunk_for(n);
    p.n_chunks = (int)((n + p.chunk - 1) / p.chunk);
    const int span = end_bit - begin_bit;
    p.n_pass = (span + bits_per_pass - 1) / bits_per_pass;
    p.max_bins = 1 << bits_per_pass;
    size_t o = 0;
    p.off_tk = o; o = align_up(o + (size_t)n * 4, 256);
    p.off_tv = o; o = align_up(o + (size_t)n * 4, 256);
    p.off_hist = o; o = align_up(o + (size_t)p.n_chunks * p.max_bins * 4, 256);
    p.off_prefix = o; o = align_up(o + (size_t)p.n_chunks * p.max_bins * 4, 256);
    p.off_totals = o; o = align_up(o + (size_t)p.max_bins * 4, 256);
    p.total = o;
    return p;
}

}  // namespace

extern "C" size_t misplat_radix_workspace_bytes(int64_t n, int32_t begin_bit, int32_t end_bit, int32_t bits_per_pass) {
    if (n < 0 || begin_bit < 0 || end_bit > 32 || end_bit <= begin_bit || bits_per_pass < 1 || bits_per_pass > kMaxBits)
        return 0;
    if (n == 0) return 256;
    return make_plan(n, begin_bit, end_bit, bits_per_pass).total;
}

extern "C" int misplat_radix_sort_pairs(void* workspace, size_t workspace_bytes, const uint32_t* keys_in,
                                        uint32_t* keys_out, const int32_t* vals_in, int32_t* vals_out, int64_t n,
                                        int32_t begin_bit, int32_t end_bit, int32_t bits_per_pass,
                                        misplat_stream_t stream) {
    if (n < 0 || n > 0x7fffffffLL || begin_bit < 0 || end_bit > 32 || end_bit <= begin_bit || bits_per_pass < 2 ||
        bits_per_pass > kMaxBits)
        return MISPLAT_EINVAL;
    if (n == 0) return MISPLAT_OK;
    const Plan p = make_plan(n, begin_bit, end_bit, bits_per_pass);
    if (!workspace || workspace_bytes < p.total) return MISPLAT_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    uint32_t* tk = (uint32_t*)(ws + p.off_tk);
    int32_t* tv = (int32_t*)(ws + p.off_tv);
    uint32_t* hist = (uint32_t*)(ws + p.off_hist);
    uint32_t* prefix = (uint32_t*)(ws + p.off_prefix);
    uint32_t* totals = (uint32_t*)(ws + p.off_totals);
    const int grid = (p.n_chunks + kWavesPerBlock - 1) / kWavesPerBlock;
    const uint32_t* src_k = keys_in;
    const int32_t* src_v = vals_in;
    for (int pass = 0; pass < p.n_pass; pass++) {
        const int shift = begin_bit + pass * bits_per_pass;
        int bits = (end_bit - shift) < bits_per_pass ? (end_bit - shift) : bits_per_pass;
        if (bits < 2) bits = 2;                                // (sorting one extra, higher key bit is harmless)
        const int nbins = 1 << bits;
        // destinations alternate so that the LAST pass writes keys_out / vals_out
        const bool to_out = ((p.n_pass - 1 - pass) % 2) == 0;
        uint32_t* dst_k = to_out ? keys_out : tk;
        int32_t* dst_v = to_out ? vals_out : tv;
        const size_t lds = (size_t)kWavesPerBlock * nbins * sizeof(uint32_t);
        hipLaunchKernelGGL(radix_hist_kernel, dim3(grid), dim3(64 * kWavesPerBlock), lds, s, src_k, n, p.chunk,
                           p.n_chunks, shift, bits, hist);
        hipLaunchKernelGGL(radix_scan_kernel, dim3(nbins / 4), dim3(256), 0, s, hist, p.n_chunks, nbins, prefix, totals);
        hipLaunchKernelGGL(radix_scatter_kernel, dim3(grid), dim3(64 * kWavesPerBlock), lds, s, src_k, src_v, n,
                           p.chunk, p.n_chunks, shift, bits, prefix, totals, dst_k, dst_v);
        src_k = dst_k;
        src_v = dst_v;
    }
    return check_launch();
}
