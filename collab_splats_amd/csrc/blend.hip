// blend.hip -- per-tile front-to-back alpha compositing (colour, alpha, expected depth, median
// depth, normal), its backward, the per-Gaussian gradient-row reduction, and the fused
// depth->normal stencil.  gfx950 (CDNA4) only: 64-lane wavefronts.
//
// Replaces the rasterize-to-pixels stage inside gsplat-rade's rasterization(...,
// return_depth_normal=True) as called at /root/reference/collab_splats/models/rade_gs_model.py:439-465
//  and camera_utils.py:176-279 + rade_gs_model.py:212-214 (row a4).
//
// Design:
//   * ONE wavefront owns a band of 16 x (4*PPL) pixels of a tile, PPL pixels per lane; a block is a
//     single wave, so no inter-wave barrier exists anywhere and bands of very different depth
//     complexity retire independently (the hardware scheduler does the load balancing).
//   * Gaussians of the tile are staged 64 at a time: one 64-byte record per lane, gathered through
//     the sorted id list, tested against the band's pixel box (exact minimum of the quadratic),
//     survivors compacted with a ballot prefix into LDS (quad-major: conflict-free stores) and then
//     broadcast-read (same address in every lane: conflict-free) once per Gaussian.
//   * Early termination is a 64-bit __ballot over "all my pixels are done".
//   * Backward: per (band, Gaussian) the 16 gradient components are summed over the lane's pixels
//     in registers, then over the 64 lanes with a halving butterfly (17 adds instead of 96), and
//     leave as ONE 64-byte row: a contiguous no-return fp32 atomic into the per-Gaussian gradient
//     (default; hidden under the VALU-bound kernel) or a store into a per-intersection slab that a
//     second kernel sums in a fixed order (bitwise reproducible).
//   * No MFMA: there is no dense contraction here; the loop is v_exp_f32 + FMA bound.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "misplat.h"
#include "internal.h"
#include "sh_eval.h"

namespace {

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

// One wavefront owns a band of 16 x (4*PPL) pixels of a tile: lane -> x = lane & 15,
// y = band_y0 + (lane >> 4) + 4k, k < PPL.  PPL = 4: one wave per tile; PPL = 2 / 1: two / four
// independent waves per tile (more waves in flight, finer early termination and culling, at the
// price of re-staging the tile's list per wave).
struct BandCtx {
    int tile, cam, band, tx, ty, y0, beg, end, unit;
    float fx, fy, cx, cy;
    bool trunc;                  // front-only ordering: the list ends at its sorted head, more entries follow unsorted
};

// XCD-aware block -> work map: blocks are dealt round-robin over the 8 XCDs, so every XCD gets a
// contiguous run of (tile, band) units (row-major neighbours share Gaussian records in its L2).
template <int PPL>
__device__ __forceinline__ bool band_ctx(const misplat_params& P, const float* __restrict__ Ks,
                                         const int32_t* __restrict__ offsets, int64_t n_isects,
                                         BandCtx& c, int first_block = 0) {
    constexpr int WPT = 4 / PPL;
    const int tiles_per_cam = P.tile_w * P.tile_h;
    const int total_tiles = tiles_per_cam * P.n_cams;
    const int total = total_tiles * WPT;
    const int per_xcd = (total + 7) >> 3;
    const int b = (int)blockIdx.x - first_block;         // (first_block: a multiple of 8, the XCD of a unit stays)
    int unit = (b & 7) * per_xcd + (b >> 3);
    // caller-supplied launch order (longest first); the grid has exactly 8 * ceil(total / 8) workgroups = entries of a
    // permutation (padding: total).  unit_sel: unit_perm is a table of records, {slot, valid} chosen on the device from
    // the call's cameras (misplat_params.unit_sel); no valid record = the default map.
    if (P.unit_perm) {
        if (!P.unit_sel) unit = P.unit_perm[b];
        else if (P.unit_sel[1] && (unsigned)P.unit_sel[0] < (unsigned)P.unit_slots)
            unit = P.unit_perm[(size_t)P.unit_sel[0] * P.unit_stride + MISPLAT_ORDER_HEADER + b];
    }
    if ((unsigned)unit >= (unsigned)total) return false;
    c.unit = unit;
    c.tile = unit / WPT;
    c.band = unit - c.tile * WPT;
    c.cam = c.tile / tiles_per_cam;
    const int t = c.tile - c.cam * tiles_per_cam;
    c.ty = t / P.tile_w;
    c.tx = t - c.ty * P.tile_w;
    c.y0 = c.ty * MISPLAT_TILE + c.band * 4 * PPL;
    if (c.y0 >= P.height) {                      // band below the image: nothing to do (and it costs nothing)
        if (P.unit_work && threadIdx.x == 0) P.unit_work[unit] = 0;
        if (P.unit_reach && threadIdx.x == 0) P.unit_reach[unit] = 0.f;
        return false;
    }
    // offsets has C*tiles + 1 entries (the last one = number of intersections); n_isects is the capacity of
    // flatten_ids: a speculative launch whose capacity turned out too small must stay inside its buffers
    c.end = min(offsets[c.tile + 1], (int)n_isects);
    c.beg = min(offsets[c.tile], c.end);
    // Front-only ordering (the forward of misplat_raster_fwd; csrc/binning.hip): front_n[tile] >= 0 -- only that many
    // entries at the head of the tile's list are there (sorted); the second pass composites the flagged tiles again, in full.
    c.trunc = false;
    if (P.front_n) {
        if (P.front_pass) {
            if (P.tile_flag[c.tile] == 0) return false;
        } else {
            const int fn = P.front_n[c.tile];
            if (fn >= 0 && fn < c.end - c.beg) { c.end = c.beg + fn; c.trunc = true; }
        }
    }
    c.fx = Ks[9 * c.cam]; c.fy = Ks[9 * c.cam + 4]; c.cx = Ks[9 * c.cam + 2]; c.cy = Ks[9 * c.cam + 5];
    return true;
}

// Exact minimum of sigma(d) = 0.5 (a dx^2 + c dy^2) + b dx dy over the box d in [dxl,dxh] x [dyl,dyh]
// (d = mean2d - pixel centre): 0 if the mean is inside, else the minimum lies on one of the two box
// faces nearest to the mean (convexity).
// nb_c = -b / c, nb_a = -b / a: the slopes of the two lines of stationary points (hardware reciprocals will do: the slopes
// only place the candidate points, and the minimum is second-order flat around them; the test has 0.2 % of slack).
__device__ __forceinline__ float sigma_min_box(float a, float b, float c, float nb_c, float nb_a, float dxl, float dxh, float dyl,
                                               float dyh) {
    const float dxc = fminf(fmaxf(0.f, dxl), dxh), dyc = fminf(fmaxf(0.f, dyl), dyh);
    const float dys = fminf(fmaxf(nb_c * dxc, dyl), dyh);
    const float dxs = fminf(fmaxf(nb_a * dyc, dxl), dxh);
    const float s1 = 0.5f * (a * dxc * dxc + c * dys * dys) + b * dxc * dys;
    const float s2 = 0.5f * (a * dxs * dxs + c * dyc * dyc) + b * dxs * dyc;
    return fminf(s1, s2);
}

// Stage up to 64 records: each lane fetches one Gaussian of the tile list, tests it against the
// wave's pixel band (conservatively: kept unless max alpha over the band is provably < alpha_min),
// and the survivors are compacted with a ballot prefix into LDS, conic pre-multiplied so that
// vis = exp2(e), e = cA' dx^2 + cC' dy^2 + cB' dx dy (= -sigma log2 e).  Returns the survivor count.
// On-demand colours (LAZY): the projection kernel leaves the colour slots of every record UNSET, and the first wave
// that stages a record past its cull evaluates the SH colour (and its Jacobian for the backward) and stores it.  A dense
// scene is mostly a hidden scene -- 1 M random Gaussians at 1080p: a third of the visible ones is ever staged, 11 % are
// composited -- so most colours are never computed.  Two waves that meet at a record compute the same bits twice.
constexpr uint32_t kColourUnset = 0x7fc0dead;      // a NaN no colour can be (colours are max(c + 0.5, 0))
struct LazyColour {
    const float* means; const float* coeffs; const float* coeffs_rest; const float* depths; const float* viewmats;
    float4* grec_rw; float* sh_aux;
    float4* v_rows;                                  // or NULL: gradient rows [C*N,16], a row is cleared when its colour is set
    float ccx, ccy, ccz;                             // camera centre of the workgroup's tile (set in the kernel)
    int deg, depth_channel, n_gauss;
    // N-D records (the features model, nxq > 0): channel 3 = features[g][0]; channels 4.. are read from the feature rows
    // themselves by whoever stages the record (FeatSrc: no featx array is written -- what one wave lays out during a launch
    // another wave, on another XCD's L2, could not rely on); v_rows_x (or NULL): the featx gradient rows, cleared with v_rows
    const float* features; float4* v_rows_x;
    int n_feat, nxq;
};
// Channels 4.. of an N-D record straight from the caller's tensors (color_copy_x_kernel's layout without the copy: fused
// channel 3 + j = features[g][j], the depth behind the last feature, zero padded); features == NULL: read featx rows.
struct FeatSrc {
    const float* features; const float* depths;
    int n_feat, depth_channel, n_gauss;
};

// Colour of row g from its 16 x 3 coefficients, term by term (MISPLAT_SH_WALK: the same expressions in the same order as
// the colour kernel, and like it without FMA contraction, so both produce the same bits -- which of the two runs is a
// speed decision the results must not show).  The coefficients are read straight from global memory as the walk
// reaches them; its scheduling barriers keep the compiler from hoisting all 48 loads to the top: the evaluation sits
// inside the compositing kernel and must not cost it its occupancy (80 VGPRs without it, 111 with).
template <int NXQ>
__device__ __forceinline__ float4 lazy_colour(const LazyColour& lz, int g) {
#pragma clang fp contract(off)
    using namespace misplat_sh;
    const int gg = g % lz.n_gauss;
    const float dx = lz.means[3 * gg] - lz.ccx, dy = lz.means[3 * gg + 1] - lz.ccy, dz = lz.means[3 * gg + 2] - lz.ccz;
    const float nn = sqrtf(dx * dx + dy * dy + dz * dz);
    const float inv = nn > 0.f ? 1.0f / nn : 0.f;
    const float x = dx * inv, y = dy * inv, z = dz * inv;
    const bool split = lz.coeffs_rest != nullptr;
    // coefficient (k, ch): float 3 k + ch of the row ([N,16,3]), or dc[ch] for k = 0 and rest[3 (k - 1) + ch] (split)
    const float* row = split ? lz.coeffs_rest + (size_t)gg * 45 - 3 : lz.coeffs + (size_t)gg * 48;
    const float* dc = split ? lz.coeffs + (size_t)gg * 3 : row;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
    // The coefficients arrive in FOUR round trips (terms 0 - 3, 4 - 8, 9 - 11, 12 - 15: a block per degree, the third degree in two),
    // each requested as a block before the first of its terms is used: the wave that evaluates sits in the compositing loop's
    // staging step and every round trip is a stall of all its pixels (term by term, as the walk's scheduling groups would have
    // it, there were eight).  At most 15 registers at a time: two round trips (27 registers) spilled inside this branch.
    float cf[48];
#define LZ_FETCH(k0, k1)                                                                \
    {                                                                                   \
        _Pragma("unroll") for (int k2 = (k0); k2 < (k1); k2++) {                        \
            const float* f_ = k2 == 0 ? dc : row + 3 * k2;                              \
            cf[3 * k2] = f_[0]; cf[3 * k2 + 1] = f_[1]; cf[3 * k2 + 2] = f_[2];         \
        }                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                              \
    }
    if (lz.deg > 0) LZ_FETCH(0, 4) else LZ_FETCH(0, 1)
#define LZ_TERM(k, B, BX, BY, BZ)                                                       \
    {                                                                                   \
        if ((k) == 4) LZ_FETCH(4, 9)                                                    \
        if ((k) == 9) LZ_FETCH(9, 12)                                                   \
        if ((k) == 12) LZ_FETCH(12, 16)                                                 \
        const float b_ = (B);                                                           \
        c0 = fmaf(b_, cf[3 * (k)], c0); c1 = fmaf(b_, cf[3 * (k) + 1], c1); c2 = fmaf(b_, cf[3 * (k) + 2], c2); \
    }
    MISPLAT_SH_WALK(lz.deg, x, y, z, LZ_TERM)
#undef LZ_FETCH
#undef LZ_TERM
    float w3 = 0.f;
    if (NXQ > 0) {                                             // (compile time: the 4-channel kernel must not pay for it)
        w3 = lz.features[(size_t)gg * lz.n_feat];              // channel 3 = feature 0 (the other channels: FeatSrc)
        if (lz.v_rows_x) {
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int q = 0; q < NXQ; q++) lz.v_rows_x[(size_t)g * NXQ + q] = z;
        }
    } else if (lz.depth_channel)
        w3 = lz.depths[g];
    const float4 q3 = make_float4(fmaxf(c0 + 0.5f, 0.f), fmaxf(c1 + 0.5f, 0.f), fmaxf(c2 + 0.5f, 0.f), w3);
    lz.grec_rw[4 * (size_t)g + 3] = q3;
    if (lz.v_rows) {
        // The backward adds into the gradient row of a record only where a pixel of the band takes it, and such a band
        // has staged the record past this same cull in this launch: whoever sets the colour clears the row (64 bytes),
        // and nothing has to clear the rows of the records no band ever reaches (misplat_raster_args.lazy_colour = 2).
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        float4* r = lz.v_rows + 4 * (size_t)g;
        r[0] = z; r[1] = z; r[2] = z; r[3] = z;
    }
    return q3;
}

// Sub-blocks (round 5).  The band's 16 x 8 pixels are two 8 x 8 HALVES (columns 0..7 | 8..15 of the tile): lane l owns
// pixel (l & 7, l >> 3) of each half.  Every staged record carries a 2-bit mask of the halves whose pixel box it can reach with
// alpha >= alpha_min (the same exact-minimum test, per half): the staged Gaussian is wave-uniform in the trip loops, so "run the
// body of half h" is a real scalar branch, and a half that the Gaussian cannot reach -- or whose pixels have all terminated
// (forward) / all ended in front of this entry (backward) -- costs nothing.  Measured on the headline's eight views with the C
// port (scripts/half_band_stats.py, cr_subblock_stats): 34 - 47 % of the contributing (band, Gaussian) trips need ONE half;
// halves of 16 x 4 (the lane's two pixels four rows apart, the layout of rounds 1 - 4): 20 - 30 %.  The union of the two
// boxes of pixel centres has a gap of one pixel pitch that the band box covered: the cull got tighter, not looser.
template <int NXQ = 0, bool LAZY = false, bool HALVES = false>
__device__ __forceinline__ int stage_records(float4* sm, int* sm_idx, int* sm_slot, int* sm_sub, int lane, int i, bool valid,
                                             const float4* grec,
                                             const int32_t* __restrict__ flatten_ids,
                                             const int32_t* __restrict__ slots, float xlo, float xhi,
                                             float ylo, float yhi, float alpha_min, float alpha_max,
                                             float4* smx = nullptr, const float4* __restrict__ featx = nullptr,
                                             const LazyColour* lz = nullptr, const FeatSrc* fs = nullptr, int* n_both = nullptr) {
    float4 q0, q1, q2, q3;
    bool keep = false;
    int slot = 0;
    int g = 0;
    int sub = 0;
    if (valid) {
        g = flatten_ids[i];
        q0 = grec[4 * (size_t)g + 0]; q1 = grec[4 * (size_t)g + 1];
        q2 = grec[4 * (size_t)g + 2]; q3 = grec[4 * (size_t)g + 3];
        if (slots) slot = slots[i]; else slot = g;
        if (HALVES) {
            const float xmid = xlo + 7.0f;                     // (xhi = xlo + 15: left half [xlo, xlo+7], right half [xlo+8, xhi])
            const float nb_c = -q0.w * __builtin_amdgcn_rcpf(q1.x), nb_a = -q0.w * __builtin_amdgcn_rcpf(q0.z);
            const float s0 = sigma_min_box(q0.z, q0.w, q1.x, nb_c, nb_a, q0.x - xmid, q0.x - xlo, q0.y - yhi, q0.y - ylo);
            const float s1 = sigma_min_box(q0.z, q0.w, q1.x, nb_c, nb_a, q0.x - xhi, q0.x - (xmid + 1.0f), q0.y - yhi, q0.y - ylo);
            const float bound = q1.y * 1.002f;
            sub = (bound * __builtin_amdgcn_exp2f(-s0 * kLog2e) >= alpha_min ? 1 : 0) |
                  (bound * __builtin_amdgcn_exp2f(-s1 * kLog2e) >= alpha_min ? 2 : 0);
            keep = sub != 0;
            // bit 2: the opacity is above alpha_max, so alpha = min(alpha_max, o vis) CAN clamp for this Gaussian; for every other
            // one (vis <= 1 where sigma >= 0) the backward's trips skip the clamp and its test
            if (q1.y > alpha_max) sub |= 4;
        } else {
            const float smin = sigma_min_box(q0.z, q0.w, q1.x, -q0.w * __builtin_amdgcn_rcpf(q1.x), -q0.w * __builtin_amdgcn_rcpf(q0.z),
                                             q0.x - xhi, q0.x - xlo, q0.y - yhi, q0.y - ylo);
            keep = q1.y * __builtin_amdgcn_exp2f(-smin * kLog2e) * 1.002f >= alpha_min;
        }
    }
    if (LAZY) {
        // (a real call, not inlined: the evaluation needs ~100 registers that the compositing loop must not pay for)
        // (all four slots start unset and any unset slot means "not there yet": correct whatever the granularity at which a
        // concurrent 16-byte store of another wave becomes visible)
        if (keep && (__float_as_uint(q3.x) == kColourUnset || __float_as_uint(q3.y) == kColourUnset ||
                     __float_as_uint(q3.z) == kColourUnset || __float_as_uint(q3.w) == kColourUnset))
            q3 = lazy_colour<NXQ>(*lz, g);
    }
    const unsigned long long mask = __ballot(keep);
    if (HALVES && n_both) *n_both = __popcll(__ballot((sub & 3) == 3));
    if (keep) {
        const int pos = __popcll(mask & ((1ull << lane) - 1ull));
        q0.z *= -0.5f * kLog2e; q0.w *= -kLog2e; q1.x *= -0.5f * kLog2e;
        sm[pos] = q0; sm[64 + pos] = q1; sm[128 + pos] = q2; sm[192 + pos] = q3;
        sm_idx[pos] = i;
        if (HALVES && sm_sub) sm_sub[pos] = sub;
        if (sm_slot) sm_slot[pos] = slot;      // emission slot (slab mode) or Gaussian row (atomic mode)
        if (NXQ > 0 && fs && fs->features) {
            const float* fr = fs->features + (size_t)g * fs->n_feat;       // (one camera: row = Gaussian; the launchers check)
            const float dz = fs->depth_channel ? fs->depths[g] : 0.f;
#pragma unroll
            for (int q = 0; q < NXQ; q++) {
                float e[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int j = 4 * q + k + 1;
                    e[k] = j < fs->n_feat ? fr[j] : (j == fs->n_feat ? dz : 0.f);
                }
                smx[q * 64 + pos] = make_float4(e[0], e[1], e[2], e[3]);
            }
        } else {
#pragma unroll
            for (int q = 0; q < NXQ; q++) smx[q * 64 + pos] = featx[(size_t)g * NXQ + q];   // channels 4 .. 4+4*NXQ
        }
    }
    return __popcll(mask);
}

// The exponent of a pair, e = -sigma log2(e) = A' dx^2 + B' dx dy + C' dy^2 with the staged (pre-multiplied) conic, from the
// half-independent terms of the trip (bdy = B' dy, ecc = C' dy^2): written with explicit fused operations so that the forward
// and the backward -- which must take the same side of every threshold for the same pair -- evaluate the same bits.
__device__ __forceinline__ float pair_exponent(float a_, float dx, float bdy, float ecc) {
    return __builtin_fmaf(__builtin_fmaf(a_, dx, bdy), dx, ecc);
}
template <int K> struct KC { static constexpr int value = K; };
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f mk2(float a, float b) { v2f r; r.x = a; r.y = b; return r; }
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f fma2(float a, v2f b, v2f c) { return __builtin_elementwise_fma(mk2(a, a), b, c); }
__device__ __forceinline__ v2f fma2(float a, v2f b, float c) { return __builtin_elementwise_fma(mk2(a, a), b, mk2(c, c)); }
__device__ __forceinline__ v2f fma2(v2f a, v2f b, float c) { return __builtin_elementwise_fma(a, b, mk2(c, c)); }
__device__ __forceinline__ v2f fma2(v2f a, float b, v2f c) { return __builtin_elementwise_fma(a, mk2(b, b), c); }
__device__ __forceinline__ float dot2(v2f a, v2f b) { return __builtin_fmaf(a.y, b.y, a.x * b.x); }

// Forward.  The per-pixel loop is branch-free: a pixel that has terminated carries T = 0 (its
// transmittance at termination is parked in Tfin), so every later weight w = a*T vanishes by itself;
// "skip" is a = 0.  The next record is prefetched from LDS while the current one is consumed.
// NXQ > 0: N-D colours (rade_features_model.py:441-476, D = 16 / 17): channels 0..3 ride in the record,
// channels 4.. in featx[row][NXQ] (float4s, zero padded); n_channels = D' is the render width.
constexpr int kBandsPerTile = MISPLAT_BANDS;
constexpr int kFillBlocks = 512;
constexpr int64_t kFillHeadRows = 2500000;
// The background role of a compositing launch (F.blocks > 0): its last workgroups -- dispatched when the machine starts
// to drain -- or (at_head: fills too large for the tail) its first ones clear the tensors of F: memory-bound waves
// beside the kernel's issue-bound ones, no launch, no graph branch.  True for a workgroup that had this role.
__device__ __forceinline__ bool background_fill(const misplat_internal::FillList& F) {
    const int fill_b = F.at_head ? (int)blockIdx.x : (int)blockIdx.x - ((int)gridDim.x - F.blocks);
    if (!(F.blocks > 0 && fill_b >= 0 && fill_b < F.blocks)) return false;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (k >= F.count) break;
        float4* __restrict__ d = (float4*)F.p[k];
        const int64_t n4 = F.n[k] >> 2;
        for (int64_t i = (int64_t)fill_b * 64 + threadIdx.x; i < n4; i += (int64_t)F.blocks * 64) d[i] = z;
        if (fill_b == 0 && (int)threadIdx.x < (int)(F.n[k] & 3)) F.p[k][4 * n4 + threadIdx.x] = 0.f;
    }
    return true;
}

template <int CD, int PPL, int NXQ = 0, bool LAZY = false>
#ifndef MISPLAT_FWD_WAVES
#define MISPLAT_FWD_WAVES 5            /* waves per SIMD the PPL-2 forward is compiled for (5: <= 102 VGPRs, 6: 80).  The plain
                                          kernel needs 80 anyway; the on-demand-colour variant (98 unbounded) bounded to six
                                          waves spills around the per-BATCH staging / evaluation -- 7 registers when this was
                                          first measured (0.222 -> 0.212 ms fixed view, 0.370 -> 0.362 cycling, 1 M / 1080p),
                                          10 since the evaluation also clears the record's gradient row.  Re-measured at the
                                          end of round 3 (view-keyed orders on): five waves, no spill, 0.339 -> 0.323 ms on the
                                          cycling views, 0.217 -> 0.213 fixed, 5 M and 100 k unchanged; four waves 0.344 */
#endif
// (N-D records on demand: 129 VGPRs unbounded at NXQ = 4 -- one over the four-wave limit)
__global__ __launch_bounds__(64, (MISPLAT_FWD_WAVES > 0 && PPL == 2 && NXQ == 0) ? MISPLAT_FWD_WAVES : ((NXQ > 0 && LAZY) ? 4 : 1)) void blend_fwd_kernel(
    misplat_params P, const float* __restrict__ Ks, const float4* grec,
    const int32_t* __restrict__ flatten_ids, const int32_t* __restrict__ offsets, int64_t n_isects,
    float* __restrict__ render, float* __restrict__ alpha, float* __restrict__ exp_depth,
    float* __restrict__ med_depth, float* __restrict__ normal, int32_t* __restrict__ last_ids,
    int32_t* __restrict__ median_ids, const float4* __restrict__ featx = nullptr, int n_channels = CD,
    LazyColour lz = LazyColour()) {
    __shared__ float4 sm[4 * 64 + 4];
    __shared__ int sm_idx[64 + 4];
    __shared__ float4 smx[NXQ > 0 ? NXQ * 64 + 4 : 1];
    float colx[PPL][NXQ > 0 ? 4 * NXQ : 1];
#pragma unroll
    for (int k = 0; k < PPL; k++)
#pragma unroll
        for (int ch = 0; ch < (NXQ > 0 ? 4 * NXQ : 1); ch++) colx[k][ch] = 0.f;
    BandCtx c;
    if (!band_ctx<PPL>(P, Ks, offsets, n_isects, c)) return;
    if (LAZY) {                                  // camera centre = -R^T t of the tile's camera (as the colour kernel has it)
#pragma clang fp contract(off)
        const float* V = lz.viewmats + 16 * c.cam;
        lz.ccx = -(V[0] * V[3] + V[4] * V[7] + V[8] * V[11]);
        lz.ccy = -(V[1] * V[3] + V[5] * V[7] + V[9] * V[11]);
        lz.ccz = -(V[2] * V[3] + V[6] * V[7] + V[10] * V[11]);
    }
    FeatSrc fsrc;
    fsrc.features = lz.features; fsrc.depths = lz.depths; fsrc.n_feat = lz.n_feat; fsrc.depth_channel = lz.depth_channel;
    fsrc.n_gauss = lz.n_gauss;
    const int lane = threadIdx.x;
    // lane -> pixel (lane & 7, lane >> 3) of each 8 x 8 half of the band: pixel k of the lane sits at column 8 k + (lane & 7).
    // The halves are what the BACKWARD's trips skip (see stage_records); here every trip runs the lane's two pixels in packed
    // instructions -- a forward that branches on the halves was built and measured (round 5): its accumulators are pixel
    // pairs either way, the three-way control flow costs the compiler ~20 register copies per trip, 0.33 -> 0.40 ms.
    // Both kernels evaluate a pair's exponent with the same fused operations (pair_exponent / its packed form), so that a
    // pair is on the same side of every threshold in the forward and in the backward.
    const int y = c.y0 + (lane >> 3);
    const float py = (float)y + 0.5f;
    float px[PPL], T[PPL], Tfin[PPL], dep[PPL], med[PPL], col[PPL][CD], nrm[PPL][3];
    int xk[PPL], last[PPL], medi[PPL];
#pragma unroll
    for (int k = 0; k < PPL; k++) {
        xk[k] = c.tx * MISPLAT_TILE + 8 * k + (lane & 7);
        px[k] = (float)xk[k] + 0.5f;
        const bool inside = xk[k] < P.width && y < P.height;
        T[k] = inside ? 1.0f : 0.0f;           // T == 0  <=>  pixel finished
        Tfin[k] = 1.0f; dep[k] = 0.f; med[k] = 0.f; last[k] = -1; medi[k] = -1;
#pragma unroll
        for (int ch = 0; ch < CD; ch++) col[k][ch] = 0.f;
        nrm[k][0] = nrm[k][1] = nrm[k][2] = 0.f;
    }
    const float amax = P.alpha_max, amin = P.alpha_min, tstop = P.t_stop, tmed = P.median_t;
    const uint32_t amax_bits = __float_as_uint(amax);
    const float xlo = (float)(c.tx * MISPLAT_TILE) + 0.5f, xhi = xlo + 15.0f;
    const float ylo = (float)c.y0 + 0.5f, yhi = ylo + (float)(4 * PPL - 1);

    // pixel pairs (k = 2 kp, 2 kp + 1) as 2-vectors for the packed-math loop (even PPL)
    static_assert(PPL % 2 == 0, "the compositing loops work on pixel pairs");
    constexpr int NP = PPL / 2;
    constexpr int NXF = NXQ > 0 ? 4 * NXQ : 1;
    v2f px2[NP], T2[NP], dep2[NP], med2[NP], col2[NP][CD], nrm2[NP][3], colx2[NP][NXF];
    {
#pragma unroll
        for (int kp = 0; kp < NP; kp++) {
            const int k0 = 2 * kp, k1 = 2 * kp + 1;
            px2[kp] = mk2(px[k0], px[k1]);
            T2[kp] = mk2(T[k0], T[k1]);
            dep2[kp] = mk2(0.f, 0.f); med2[kp] = mk2(0.f, 0.f);
#pragma unroll
            for (int ch = 0; ch < CD; ch++) col2[kp][ch] = mk2(0.f, 0.f);
#pragma unroll
            for (int ch = 0; ch < 3; ch++) nrm2[kp][ch] = mk2(0.f, 0.f);
#pragma unroll
            for (int ch = 0; ch < NXF; ch++) colx2[kp][ch] = mk2(0.f, 0.f);
        }
    }

    // Packed path (even PPL): the per-pixel control state lives in LANE MASKS (scalar registers), not in vector compares
    // of the transmittance -- `alive` (not terminated), `above` (T still above the median threshold) -- so that the
    // trip's decisions are scalar ANDs of three vector compares per pixel (sigma >= 0, alpha >= alpha_min, T' <= t_stop)
    // instead of six, a finished pixel needs no parking register (its T is kept NEGATED: magnitude = transmittance at
    // termination, and alpha = 0 for it, so nothing accumulates), and the median bookkeeping is skipped by a wave-uniform
    // branch once no pixel of the band is above the threshold any more (most trips: T falls below 0.5 early, the
    // traversal runs on to 1e-4).
    // (Masks are carried explicitly as 64-bit scalars and turned back into lane predicates with inverse_ballot: a `bool`
    // carried around the loop would be legalised into a 0/1 vector register, three vector instructions per trip each.)
    typedef unsigned long long lmask;
    lmask alive[PPL], above[PPL];
#pragma unroll
    for (int k = 0; k < PPL; k++) { alive[k] = __ballot(T[k] > 0.f); above[k] = alive[k]; }
    bool med_live = true;                       // (wave-uniform) some pixel may still take its median from a later Gaussian
#define MISPLAT_LANE(m) __builtin_amdgcn_inverse_ballot_w64(m)

    int work = 0;                               // staged Gaussians composited by this unit (its measured cost)
    int reach_end = c.beg;                      // end of the last batch of list entries the band looked at
    for (int bs = c.beg; bs < c.end; bs += 64) {
        {
            lmask any = alive[0];
#pragma unroll
            for (int k = 1; k < PPL; k++) any |= alive[k];
            if (any == 0ull) break;
        }
        __syncthreads();
        int n_both = 0;
        const int n = stage_records<NXQ, LAZY, true>(sm, sm_idx, nullptr, nullptr, lane, bs + lane, bs + lane < c.end, grec, flatten_ids,
                                                     nullptr, xlo, xhi, ylo, yhi, amin, amax, smx, featx, &lz, (LAZY && NXQ > 0) ? &fsrc : nullptr, &n_both);
        __syncthreads();
        reach_end = min(bs + 64, c.end);
        if (n == 0) continue;
        work += 2 * n + n_both;           // (what the backward will pay for these entries: 3 per two-half trip, 2 per one-half trip)
        // LDS latency is hidden without a second register set: the half of the record consumed late (q2, q3 =
        // ray plane / normal / colour, and the index) is read at the top of its own iteration, the half consumed
        // early (q0, q1 = mean, conic, opacity) is read for the NEXT Gaussian as soon as this one's alpha is known.
        // The scheduling fences keep the reads where they are written (the compiler otherwise sinks them to
        // their first use and every iteration starts with an exposed ds_read).
        float4 q0 = sm[0], q1 = sm[64];
        for (int j = 0; j < n; j++) {
            const float4 q2 = sm[128 + j], q3 = sm[192 + j];
            const int i = sm_idx[j];
            float4 xq[NXQ > 0 ? NXQ : 1];
#pragma unroll
            for (int q = 0; q < NXQ; q++) xq[q] = smx[q * 64 + j];
            __builtin_amdgcn_sched_barrier(0);
            const float dy = q0.y - py;
            const float ecc = (q1.x * dy) * dy, bdy = q0.w * dy;
            {
                // two pixels of the lane per packed instruction (see blend_bwd_kernel)
                v2f dx_[NP], a_[NP];
                lmask ok_[PPL];
#pragma unroll
                for (int kp = 0; kp < NP; kp++) {
                    const v2f dx = q0.x - px2[kp];
                    const v2f e = fma2(fma2(q0.z, dx, bdy), dx, ecc);
                    v2f vis;
                    vis.x = __builtin_amdgcn_exp2f(e.x); vis.y = __builtin_amdgcn_exp2f(e.y);
                    const v2f ov = q1.y * vis;
                    // min(alpha_max, o vis) on the bit patterns: both are non-negative floats (a NaN sorts above every
                    // number and fails the `e <= 0` test anyway), so the unsigned integer minimum is the float minimum
                    // -- without the canonicalisation a float min needs
                    const float am0 = __uint_as_float(min(amax_bits, __float_as_uint(ov.x)));
                    const float am1 = __uint_as_float(min(amax_bits, __float_as_uint(ov.y)));
                    // (one ballot per compare: the compare then writes its lane mask straight into scalar registers)
                    const lmask ok0 = alive[2 * kp] & __ballot(e.x <= 0.f) & __ballot(am0 >= amin);
                    const lmask ok1 = alive[2 * kp + 1] & __ballot(e.y <= 0.f) & __ballot(am1 >= amin);
                    v2f a;
                    a.x = MISPLAT_LANE(ok0) ? am0 : 0.f; a.y = MISPLAT_LANE(ok1) ? am1 : 0.f;
                    dx_[kp] = dx; a_[kp] = a; ok_[2 * kp] = ok0; ok_[2 * kp + 1] = ok1;
                }
                const float q1z = q1.z, q1w = q1.w;
                q0 = sm[j + 1]; q1 = sm[64 + j + 1];                   // next Gaussian (array padded)
                __builtin_amdgcn_sched_barrier(0);
                lmask any_alive = 0ull;
#pragma unroll
                for (int kp = 0; kp < NP; kp++) {
                    const int k0 = 2 * kp, k1 = 2 * kp + 1;
                    const v2f dx = dx_[kp], a = a_[kp];
                    const v2f Tk = T2[kp];
                    v2f w = a * Tk;                                      // (a = 0 for a finished pixel: its T stays as it is)
                    const v2f Tn = Tk - w;
                    const lmask stop0 = ok_[k0] & __ballot(Tn.x <= tstop), stop1 = ok_[k1] & __ballot(Tn.y <= tstop);   // this Gaussian is excluded
                    const lmask use0 = ok_[k0] & ~stop0, use1 = ok_[k1] & ~stop1;
                    T2[kp].x = MISPLAT_LANE(stop0) ? -Tk.x : Tn.x; T2[kp].y = MISPLAT_LANE(stop1) ? -Tk.y : Tn.y;
                    w.x = MISPLAT_LANE(stop0) ? 0.f : w.x; w.y = MISPLAT_LANE(stop1) ? 0.f : w.y;
                    alive[k0] &= ~stop0; alive[k1] &= ~stop1;
                    any_alive |= alive[k0] | alive[k1];
                    // (depth times |ray| = rt - rp . d: the 1 / |ray| of the pixel multiplies the finished sums, below)
                    const v2f zp = (q1z - q2.x * dy) - q1w * dx;                 // (q2 is consumed late: its LDS read hides behind alpha)
                    col2[kp][0] += w * q3.x;
                    if (CD > 1) col2[kp][CD > 1 ? 1 : 0] += w * q3.y;
                    if (CD > 2) col2[kp][CD > 2 ? 2 : 0] += w * q3.z;
                    if (CD > 3) col2[kp][CD > 3 ? 3 : 0] += w * q3.w;
#pragma unroll
                    for (int q = 0; q < NXQ; q++) {
                        colx2[kp][4 * q + 0] += w * xq[q].x; colx2[kp][4 * q + 1] += w * xq[q].y;
                        colx2[kp][4 * q + 2] += w * xq[q].z; colx2[kp][4 * q + 3] += w * xq[q].w;
                    }
                    dep2[kp] += w * zp;
                    nrm2[kp][0] += w * q2.y; nrm2[kp][1] += w * q2.z; nrm2[kp][2] += w * q2.w;
                    last[k0] = MISPLAT_LANE(use0) ? i : last[k0]; last[k1] = MISPLAT_LANE(use1) ? i : last[k1];
                    if (med_live) {                                      // wave-uniform
                        asm volatile("; median bookkeeping" ::);         // (a real branch: not worth speculating on every trip)
                        const lmask med0 = use0 & above[k0], med1 = use1 & above[k1];        // T BEFORE this Gaussian > median_t
                        med2[kp].x = MISPLAT_LANE(med0) ? zp.x : med2[kp].x; med2[kp].y = MISPLAT_LANE(med1) ? zp.y : med2[kp].y;
                        medi[k0] = MISPLAT_LANE(med0) ? i : medi[k0]; medi[k1] = MISPLAT_LANE(med1) ? i : medi[k1];
                        above[k0] = __ballot(T2[kp].x > tmed); above[k1] = __ballot(T2[kp].y > tmed);   // (a finished pixel's T is negative)
                    }
                }
                if (med_live) {
                    lmask any_above = above[0];
#pragma unroll
                    for (int k = 1; k < PPL; k++) any_above |= above[k];
                    med_live = any_above != 0ull;
                }
                if (any_alive == 0ull) break;
            }
        }
    }
    if (P.unit_work && lane == 0) P.unit_work[c.unit] = work;
    if (P.unit_reach) {
        // How deep this visit went -- the pivot of the view's next visit (front-only ordering): the depth of the last list
        // entry of the last batch the band staged; +inf when pixels are still alive at the end of the list (the whole list
        // was needed) -- and if that list was only the sorted head of a longer one, the tile is flagged: it is sorted in
        // full and composited again by the second pass.
        lmask any = alive[0];
#pragma unroll
        for (int k = 1; k < PPL; k++) any |= alive[k];
        float reach = 0.f;
        if (any != 0ull) {
            reach = __builtin_inff();
            if (c.trunc && lane == 0) P.tile_flag[c.tile] = 1;
        } else if (reach_end > c.beg)
            reach = P.front_depths[flatten_ids[reach_end - 1]];
        if (lane == 0) P.unit_reach[c.unit] = reach;
    }
    {
#pragma unroll
        for (int kp = 0; kp < NP; kp++) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int k = 2 * kp + h;
                const float Ts = h ? T2[kp].y : T2[kp].x;               // negative: terminated, magnitude = T at termination
                T[k] = Ts > 0.f ? Ts : 0.f; Tfin[k] = fabsf(Ts);
                // 1 / |ray| of the pixel, formed here (nothing of it lives across the loop); the backward uses the same expression
                const float rxn = ((h ? px2[kp].y : px2[kp].x) - c.cx) / c.fx, ryn = (py - c.cy) / c.fy;
                const float ilk = 1.0f / sqrtf(rxn * rxn + ryn * ryn + 1.0f);
                dep[k] = (h ? dep2[kp].y : dep2[kp].x) * ilk; med[k] = (h ? med2[kp].y : med2[kp].x) * ilk;
#pragma unroll
                for (int ch = 0; ch < CD; ch++) col[k][ch] = h ? col2[kp][ch].y : col2[kp][ch].x;
#pragma unroll
                for (int ch = 0; ch < 3; ch++) nrm[k][ch] = h ? nrm2[kp][ch].y : nrm2[kp][ch].x;
#pragma unroll
                for (int ch = 0; ch < (NXQ > 0 ? 4 * NXQ : 0); ch++) colx[k][ch] = h ? colx2[kp][ch].y : colx2[kp][ch].x;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < PPL; k++) {
        if (xk[k] < P.width && y < P.height) {
            const size_t pid = ((size_t)c.cam * P.height + y) * P.width + xk[k];
            const float al = 1.0f - (T[k] > 0.f ? T[k] : Tfin[k]);
            const float inv_al = 1.0f / fmaxf(al, 1e-10f);
            if (NXQ == 0) {
#pragma unroll
                for (int ch = 0; ch < CD; ch++) render[pid * CD + ch] = (ch == P.ed_slot) ? col[k][ch] * inv_al : col[k][ch];
            } else {
                float* o = render + pid * (size_t)n_channels;
#pragma unroll
                for (int ch = 0; ch < CD; ch++) o[ch] = (ch == P.ed_slot) ? col[k][ch] * inv_al : col[k][ch];
#pragma unroll
                for (int ch = 0; ch < 4 * NXQ; ch++)
                    if (CD + ch < n_channels) o[CD + ch] = (CD + ch == P.ed_slot) ? colx[k][ch] * inv_al : colx[k][ch];
            }
            alpha[pid] = al;
            exp_depth[pid] = dep[k];
            med_depth[pid] = med[k];
            normal[pid * 3 + 0] = nrm[k][0]; normal[pid * 3 + 1] = nrm[k][1]; normal[pid * 3 + 2] = nrm[k][2];
            last_ids[pid] = last[k];
            median_ids[pid] = medi[k];
        }
    }
}

// ---- wave reductions (DPP / permlane: no LDS traffic) ------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
constexpr int kDppHalfMirror = 0x141;  // lane l <-> l ^ 7   (within 8)
constexpr int kDppXor1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int kDppRor8 = 0x128;        // row_ror:8 == l ^ 8 (within 16)

__device__ __forceinline__ float cross_row_sum(float r) {      // + lanes l^16, l^32 (v_permlane*_swap)
    typedef int v2i __attribute__((ext_vector_type(2)));
    int ri = __float_as_int(r);
    v2i s16 = __builtin_amdgcn_permlane16_swap(ri, ri, false, false);
    r = __int_as_float(s16.x) + __int_as_float(s16.y);
    ri = __float_as_int(r);
    v2i s32 = __builtin_amdgcn_permlane32_swap(ri, ri, false, false);
    return __int_as_float(s32.x) + __int_as_float(s32.y);
}

// Halving butterfly: 16 values per lane -> 1 per lane, summed over all 64 lanes; lane l ends with
// component butterfly_comp(l) = (l >> 2) & 15 (replicated over the 4 lanes of its quad).
// An exchange unit takes two values (a, b) and a lane bit s and leaves a_self + a_partner in the lanes
// with s = 0, b_self + b_partner in those with s = 1.  Measured on gfx950 (scripts/ubench/valu_rates.hip):
// the select-select-add form (2 v_cndmask + v_add_dpp) costs ~22 cycles per unit in a dependent chain,
// two BANK-MASKED DPP adds ~7 (bank_mask enables groups of 4 lanes of a row, i.e. lane bits 2 and 3), a
// v_permlane{16,32}_swap + add ~12 (lane bits 4 and 5).  So the two big stages run on the bank bits, the
// two small ones on the row bits, and the quad bits (0, 1) are a plain 2-step sum of the single survivor:
//   bit 2: 8 units, row_half_mirror, banks 1|3 vs 0|2      bit 3: 4 units, row_ror:8, banks 2|3 vs 0|1
//   bit 4: 2 units, v_permlane16_swap                      bit 5: 1 unit,  v_permlane32_swap
// The compiler cannot see inside the asm, so each block opens with the 2 wait states a DPP / permlane
// read of a freshly written VGPR needs on gfx9.
#define MISPLAT_XU(k, n, ctrl, m_hi, m_lo)                                                            \
    "v_add_f32_dpp %" #k ", %" #k ", %" #k " " ctrl " row_mask:0xf bank_mask:" m_hi "\n\t"            \
    "v_add_f32_dpp %" #k ", %" #n ", %" #n " " ctrl " row_mask:0xf bank_mask:" m_lo "\n\t"
__device__ __forceinline__ float wave_reduce16(float (&v)[16], int lane) {
    (void)lane;
    // bit 2: (v[2i], v[2i+1]) -> v[2i+1]
    asm("s_nop 1\n\t"
        MISPLAT_XU(0, 8, "row_half_mirror", "0xa", "0x5") MISPLAT_XU(1, 9, "row_half_mirror", "0xa", "0x5")
        MISPLAT_XU(2, 10, "row_half_mirror", "0xa", "0x5") MISPLAT_XU(3, 11, "row_half_mirror", "0xa", "0x5")
        MISPLAT_XU(4, 12, "row_half_mirror", "0xa", "0x5") MISPLAT_XU(5, 13, "row_half_mirror", "0xa", "0x5")
        MISPLAT_XU(6, 14, "row_half_mirror", "0xa", "0x5") MISPLAT_XU(7, 15, "row_half_mirror", "0xa", "0x5")
        : "+v"(v[1]), "+v"(v[3]), "+v"(v[5]), "+v"(v[7]), "+v"(v[9]), "+v"(v[11]), "+v"(v[13]), "+v"(v[15])
        : "v"(v[0]), "v"(v[2]), "v"(v[4]), "v"(v[6]), "v"(v[8]), "v"(v[10]), "v"(v[12]), "v"(v[14]));
    // bit 3: (v[1], v[3]) -> v[3], (v[5], v[7]) -> v[7], (v[9], v[11]) -> v[11], (v[13], v[15]) -> v[15]
    asm("s_nop 1\n\t"
        MISPLAT_XU(0, 4, "row_ror:8", "0xc", "0x3") MISPLAT_XU(1, 5, "row_ror:8", "0xc", "0x3")
        MISPLAT_XU(2, 6, "row_ror:8", "0xc", "0x3") MISPLAT_XU(3, 7, "row_ror:8", "0xc", "0x3")
        : "+v"(v[3]), "+v"(v[7]), "+v"(v[11]), "+v"(v[15])
        : "v"(v[1]), "v"(v[5]), "v"(v[9]), "v"(v[13]));
    // bit 4: rows 0|2 keep the first value, rows 1|3 the second
    asm("s_nop 1\n\t"
        "v_permlane16_swap_b32 %0, %1\n\t"
        "v_permlane16_swap_b32 %2, %3"
        : "+v"(v[3]), "+v"(v[7]), "+v"(v[11]), "+v"(v[15]));
    float lo = v[3] + v[7], hi = v[11] + v[15];
    // bit 5: lanes 0..31 keep lo, lanes 32..63 keep hi
    asm("s_nop 1\n\t"
        "v_permlane32_swap_b32 %0, %1"
        : "+v"(lo), "+v"(hi));
    float r = lo + hi;
    r += dpp_mov<kDppXor1>(r);
    r += dpp_mov<kDppXor2>(r);
    return r;
}
#undef MISPLAT_XU
__device__ __forceinline__ int butterfly_comp(int l) { return (l >> 2) & 15; }

__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_mov<kDppHalfMirror>(v);
    v += dpp_mov<kDppXor1>(v);
    v += dpp_mov<kDppXor2>(v);
    v += dpp_mov<kDppRor8>(v);
    return cross_row_sum(v);
}

__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v = max(v, __shfl_xor(v, m));
    return v;
}

// Backward.  Branch-free per-pixel body (a == 0 makes a pair a no-op), next record prefetched from
// LDS.  ATOMIC: the reduced 64-byte row is added into v_grec[row]; otherwise it is stored to
// slab[band][slot] and marked in valid[band][slot] (zeroed by the launcher).
// Measured (1 M / 1080p, PPL 2): 4 waves/SIMD (111 VGPRs) 0.663 ms, 5 waves (96 VGPRs + 48 B
// scratch) 0.616 ms, 6 waves (80 VGPRs + 88 B scratch) 0.711 ms.
// PPL 4: 2 waves (172 VGPRs) 0.73 ms, 3 waves 0.655 ms, 4 waves (spills) 1.28 ms.
// MSUM (the one-call backward's flagged-row form, where the per-Gaussian kernel of the same call is the only reader of the rows):
// row slots 0 - 1 carry  sum dx dL/dsigma,  sum dy dL/dsigma  instead of the mean2d gradient -- which is linear in them,
//   v_mean2d = (a s0 + b s1 - rp_x v_rt,  b s0 + c s1 - rp_y v_rt)     (conic a b c, ray plane rp, v_rt = slot 6),
// and slot 5 carries  sum o vis dL/dalpha = o times the opacity gradient (the sum of -dL/dsigma, which the trip has anyway).
// Both are finished once per ROW by the per-Gaussian kernel (project.hip: project_bwd_one, pp_bwd_row) where this kernel
// finished them once per PIXEL: seven packed + eight plain instructions of a two-pixel trip, ten of a one-pixel trip.
template <int CD, int PPL, bool ABS, bool ATOMIC, int NXQ = 0, bool MSUM = false>
#ifndef MISPLAT_BWD_WAVES
#define MISPLAT_BWD_WAVES 5
#endif
__global__ __launch_bounds__(64, (PPL == 2 && NXQ == 0) ? MISPLAT_BWD_WAVES : (NXQ > 0 ? 4 : 1)) void blend_bwd_kernel(
    misplat_params P, const float* __restrict__ Ks, const float4* __restrict__ grec,
    const int32_t* __restrict__ flatten_ids, const int32_t* __restrict__ slots,
    const int32_t* __restrict__ offsets, int64_t n_isects, const float* __restrict__ alpha,
    const int32_t* __restrict__ last_ids, const int32_t* __restrict__ median_ids,
    const float* __restrict__ render, const float* __restrict__ v_render, const float* __restrict__ v_alpha,
    const float* __restrict__ v_exp_depth, const float* __restrict__ v_med_depth,
    const float* __restrict__ v_normal, float* __restrict__ slab, float* __restrict__ slab_abs,
    uint8_t* __restrict__ valid, const float4* __restrict__ featx = nullptr, float* __restrict__ v_featx = nullptr,
    int n_channels = CD, misplat_internal::FillList F = {}, FeatSrc fsrc = FeatSrc()) {
    static_assert(NXQ == 0 || ATOMIC, "N-D colours: atomic gradient mode only");
    static_assert(!MSUM || (ATOMIC && !ABS), "sums for the mean2d gradient: the flagged-row backward without absgrad");
    static_assert(PPL == 2, "a band is two 8 x 8 halves, one pixel of each per lane");
    // Background role (F.blocks > 0): the last workgroups of the grid -- dispatched when the machine starts to drain --
    // or (at_head: fills too large for the tail) the first ones clear the tensors the per-Gaussian backward kernels write
    // sparsely afterwards: memory-bound waves beside this kernel's issue-bound ones, no launch, no graph branch.
    if (background_fill(F)) return;
    __shared__ float4 sm[4 * 64 + 4];
    __shared__ int sm_idx[64 + 4];
    __shared__ int sm_slot[64 + 4];
    __shared__ int sm_sub[64 + 4];
    __shared__ float4 smx[NXQ > 0 ? NXQ * 64 + 4 : 1];
    constexpr int NX = NXQ > 0 ? 4 * NXQ : 1;
    const size_t rstride = NXQ > 0 ? (size_t)n_channels : (size_t)CD;
    BandCtx c;
    if (!band_ctx<PPL>(P, Ks, offsets, n_isects, c, F.at_head ? F.blocks : 0)) return;
    if (c.end <= c.beg) return;
    const int lane = threadIdx.x;
    // lane -> pixel (lane & 7, lane >> 3) of each 8 x 8 half (see stage_records); the lane's two pixels as 2-vectors, a trip
    // that reaches both halves in packed instructions, a trip that reaches one on component k (see blend_fwd_kernel)
    const int y = c.y0 + (lane >> 3);
    const float py = (float)y + 0.5f;
    const float ryn = (py - c.cy) / c.fy;
    v2f px2, il2, T2, D2, vd2, vm2, vcol2[CD], vn2[3], vcolx2[NX];   // D2 = T_final v_alpha - (what the pixels behind have composited) . v
    int last[PPL], medi[PPL], sublast[PPL];
#pragma unroll
    for (int ch = 0; ch < CD; ch++) vcol2[ch] = mk2(0.f, 0.f);
#pragma unroll
    for (int ch = 0; ch < NX; ch++) vcolx2[ch] = mk2(0.f, 0.f);
#pragma unroll
    for (int ch = 0; ch < 3; ch++) vn2[ch] = mk2(0.f, 0.f);
#pragma unroll
    for (int k = 0; k < PPL; k++) {
        const int x = c.tx * MISPLAT_TILE + 8 * k + (lane & 7);
        px2[k] = (float)x + 0.5f;
        const float rxn = (px2[k] - c.cx) / c.fx;
        il2[k] = 1.0f / sqrtf(rxn * rxn + ryn * ryn + 1.0f);
        last[k] = -1; medi[k] = -1; T2[k] = 1.f; D2[k] = 0.f; vd2[k] = 0.f; vm2[k] = 0.f;
        if (x < P.width && y < P.height) {
            const size_t pid = ((size_t)c.cam * P.height + y) * P.width + x;
            last[k] = last_ids[pid];
            medi[k] = median_ids[pid];
            const float al = alpha[pid];
            const float Tf = 1.0f - al;
            T2[k] = Tf;
            float va = v_alpha[pid];
#pragma unroll
            for (int ch = 0; ch < CD; ch++) {
                float g = v_render[pid * rstride + ch];
                if (ch == P.ed_slot) {         // out = raw / max(alpha, 1e-10)
                    const float inv_al = 1.0f / fmaxf(al, 1e-10f);
                    g *= inv_al;
                    if (al > 1e-10f) va -= g * render[pid * rstride + ch];
                }
                vcol2[ch][k] = g;
            }
#pragma unroll
            for (int ch = 0; ch < (NXQ > 0 ? NX : 0); ch++) {
                if (CD + ch < n_channels) {
                    float g = v_render[pid * rstride + CD + ch];
                    if (CD + ch == P.ed_slot) {
                        const float inv_al = 1.0f / fmaxf(al, 1e-10f);
                        g *= inv_al;
                        if (al > 1e-10f) va -= g * render[pid * rstride + CD + ch];
                    }
                    vcolx2[ch][k] = g;
                }
            }
            D2[k] = Tf * va;
            vn2[0][k] = v_normal[pid * 3]; vn2[1][k] = v_normal[pid * 3 + 1]; vn2[2][k] = v_normal[pid * 3 + 2];
            vd2[k] = v_exp_depth[pid] * il2[k];         // (the depth gradients ride pre-multiplied by 1 / |ray|: the pixel's
                                                          //  depth is (rt - rp . d) / |ray|, and only these two see the factor)
            vm2[k] = v_med_depth[pid] * il2[k];
        }
        // (wave-uniform, and in a SCALAR register: the deepest entry any pixel of half k composited)
        sublast[k] = __builtin_amdgcn_readfirstlane(wave_max(last[k]));
    }
    const int maxlast = max(sublast[0], sublast[1]);
    if (maxlast < c.beg) return;
    const int comp = butterfly_comp(lane);
    const bool writer = (lane & 3) == 0;          // one lane per quad holds (and writes) component `comp`
    // per-lane scale undoing the conic pre-multiplication (component = record layout index)
    const float out_scale = MSUM ? ((comp == 2 || comp == 4) ? -0.5f : ((comp < 2 || comp == 3) ? -1.0f : 1.0f))
                                 : ((comp == 2 || comp == 4) ? -0.5f * kLog2e : (comp == 3 ? -kLog2e : 1.0f));
    const float amax = P.alpha_max, amin = P.alpha_min;
    const uint32_t amax_bits = __float_as_uint(amax);
    const uint32_t comp_off = 4u * (uint32_t)comp;           // byte offset of this lane's component inside a 64-byte row
    const float xlo = (float)(c.tx * MISPLAT_TILE) + 0.5f, xhi = xlo + 15.0f;
    const float ylo = (float)c.y0 + 0.5f, yhi = ylo + (float)(4 * PPL - 1);
    // slab mode: one plane per band; atomic mode: slab IS v_grec[C*N,16] (zeroed by the launcher)
    float* slab_b = ATOMIC ? slab : slab + (size_t)c.band * (size_t)n_isects * MISPLAT_REC;
    float* abs_b = ABS ? (ATOMIC ? slab_abs : slab_abs + (size_t)c.band * (size_t)n_isects * 2) : nullptr;
    uint8_t* valid_b = ATOMIC ? nullptr : valid + (size_t)c.band * (size_t)n_isects;
    typedef unsigned long long lmask;

    for (int b = (maxlast - c.beg) >> 6; b >= 0; b--) {
        const int bs = c.beg + (b << 6);
        __syncthreads();
        const int n = stage_records<NXQ, false, true>(sm, sm_idx, sm_slot, sm_sub, lane, bs + lane, bs + lane <= maxlast, grec, flatten_ids,
                                         ATOMIC ? nullptr : slots, xlo, xhi, ylo, yhi, amin, amax, smx, featx, nullptr,
                                         NXQ > 0 ? &fsrc : nullptr);
        __syncthreads();
        if (n == 0) continue;
        // does any pixel of the band take its median depth from a Gaussian of this batch?
        const bool batch_has_median = __ballot(((unsigned)(medi[0] - bs) < 64u) | ((unsigned)(medi[1] - bs) < 64u)) != 0ull;
        // The record of the next Gaussian is fetched from LDS into the SAME registers right after the last
        // use of the current one (before the butterfly, which hides the latency): no second register set
        // and no copies.
        float4 q0 = sm[n - 1], q1 = sm[64 + n - 1], q2 = sm[128 + n - 1], q3 = sm[192 + n - 1];
        int iv = sm_idx[n - 1], islot = sm_slot[n - 1], subv = sm_sub[n - 1];
        unsigned long long touched_j = 0ull;       // staged entries of this batch that received a gradient row
        for (int j = n - 1; j >= 0; j--) {
            // The halves this trip has to run: those the staged box test admits AND whose pixels reach this deep
            // (i <= the half's deepest last_id) -- all wave-uniform, so a real scalar branch per half.
            const int i = __builtin_amdgcn_readfirstlane(iv);
            const int subr = __builtin_amdgcn_readfirstlane(subv);
            const int sub = subr & ((i <= sublast[0] ? 1 : 0) | (i <= sublast[1] ? 2 : 0));
            const bool can_clamp = (subr & 4) != 0;        // (wave-uniform, a scalar: see stage_records)
            float acc[16];
            float accx[16];
            float ab0 = 0.f, ab1 = 0.f;
            lmask any_ok = 0ull;                   // lanes with a contributing pixel
            if (sub != 0) {
                float4 xq[NXQ > 0 ? NXQ : 1];
#pragma unroll
                for (int q = 0; q < NXQ; q++) xq[q] = smx[q * 64 + j];
                // terms that do not depend on the half (the lane's two pixels share their row)
                const float dy = q0.y - py;
                const float bdy = q0.w * dy, ecc = (q1.x * dy) * dy;
                const float tpy = q1.z - q2.x * dy;
                // (N-D records, NXQ > 0: every trip takes the two-pixel body -- 17 more accumulations and a second butterfly per
                //  trip leave no registers for a second body: with one, the NXQ = 4 kernel spilled 160 bytes inside the loop and
                //  the features model's backward went from 0.83 to 1.73 ms)
                if (NXQ > 0 || sub == 3) {
                    // ---- both halves: the lane's two pixels in packed instructions (v_pk_{fma,mul,add}_f32); the per-pixel
                    // decisions are lane masks in scalar registers (one ballot per vector compare, combined with scalar ANDs,
                    // turned back into select conditions with inverse_ballot): see blend_fwd_kernel
                    const float ndy = -dy;
                    const v2f dx = q0.x - px2;
                    const v2f e = fma2(fma2(q0.z, dx, bdy), dx, ecc);
                    v2f vis;
                    vis.x = __builtin_amdgcn_exp2f(e.x); vis.y = __builtin_amdgcn_exp2f(e.y);
                    const v2f ov = q1.y * vis;
                    // min(alpha_max, o vis) as an unsigned minimum of the bit patterns (non-negative floats; a NaN fails e <= 0)
                    float am0 = ov.x, am1 = ov.y;
                    lmask below0 = ~0ull, below1 = ~0ull;       // pixels whose alpha is NOT clamped (d alpha / d (o vis) = 1)
                    if (can_clamp) {
                        asm volatile("; alpha_max clamp" ::);   // (a real branch: few Gaussians have o > alpha_max)
                        am0 = __uint_as_float(min(amax_bits, __float_as_uint(ov.x)));
                        am1 = __uint_as_float(min(amax_bits, __float_as_uint(ov.y)));
                        below0 = __ballot(ov.x <= amax); below1 = __ballot(ov.y <= amax);
                    }
                    const lmask okm0 = __ballot(i <= last[0]) & __ballot(e.x <= 0.f) & __ballot(am0 >= amin);
                    const lmask okm1 = __ballot(i <= last[1]) & __ballot(e.y <= 0.f) & __ballot(am1 >= amin);
                    any_ok = okm0 | okm1;
                    const bool ok0 = __builtin_amdgcn_inverse_ballot_w64(okm0), ok1 = __builtin_amdgcn_inverse_ballot_w64(okm1);
                    v2f a;
                    a.x = ok0 ? am0 : 0.f; a.y = ok1 ? am1 : 0.f;
                    const v2f om = 1.0f - a;
                    v2f ra;
                    ra.x = __builtin_amdgcn_rcpf(om.x); ra.y = __builtin_amdgcn_rcpf(om.y);
                    T2 *= ra;
                    const v2f Tk = T2;
                    const v2f w = a * Tk;
                    const v2f zp = tpy - q1.w * dx;                       // (depth times |ray|: vd2 carries the 1 / |ray|)
                    v2f dot = q3.x * vcol2[0];
                    if (CD > 1) dot += q3.y * vcol2[CD > 1 ? 1 : 0];
                    if (CD > 2) dot += q3.z * vcol2[CD > 2 ? 2 : 0];
                    if (CD > 3) dot += q3.w * vcol2[CD > 3 ? 3 : 0];
#pragma unroll
                    for (int q = 0; q < NXQ; q++)
                        dot += xq[q].x * vcolx2[4 * q] + xq[q].y * vcolx2[4 * q + 1] + xq[q].z * vcolx2[4 * q + 2] + xq[q].w * vcolx2[4 * q + 3];
                    dot += q2.y * vn2[0] + q2.z * vn2[1] + q2.w * vn2[2] + zp * vd2;
                    const v2f v_a = D2 * ra + Tk * dot;                   // (only used through vam below: masked there)
                    D2 -= w * dot;
                    v2f vzl = w * vd2;                                    // gradient of (rt - rp . d): depth gradient / |ray|
                    if (batch_has_median) {          // wave-uniform: most batches hold no pixel's median Gaussian
                        asm volatile("; median gradient" ::);               // (a real branch, not a speculated select)
                        v2f vmed;
                        vmed.x = (ok0 && i == medi[0]) ? vm2.x : 0.f; vmed.y = (ok1 && i == medi[1]) ? vm2.y : 0.f;
                        vzl += vmed;
                    }
                    // d alpha / d (o vis) is 1 below the clamp, 0 at it; and nothing flows through a pixel that skipped
                    const bool un0 = __builtin_amdgcn_inverse_ballot_w64(okm0 & below0);
                    const bool un1 = __builtin_amdgcn_inverse_ballot_w64(okm1 & below1);
                    v2f vam;
                    vam.x = un0 ? v_a.x : 0.f; vam.y = un1 ? v_a.y : 0.f;
                    const v2f v_e = MSUM ? ov * vam : (kLn2 * ov) * vam;    // (MSUM: -dL/dsigma, the row scales carry no log2 e)
                    const v2f dxve = dx * v_e;
                    // Per-Gaussian sums over the lane's two pixels.  A packed multiply costs two plain issue slots on gfx950
                    // (scripts/ubench/valu_rates.hip), so forming 16 packed products and then adding their halves (48 slots)
                    // loses against scalar mul + fma on the halves (<= 2 per component) with shared factors pulled out: 28.
                    const float s_ve = v_e.x + v_e.y, s_dxve = dxve.x + dxve.y, s_vzl = vzl.x + vzl.y;
                    v2f vmx, vmy;
                    if (MSUM) {
                        acc[0] = s_dxve; acc[1] = dy * s_ve;
                        acc[4] = dy * acc[1];
                    } else {
                        const float c1y = 2.0f * q1.x * dy;
                        vmx = (2.0f * q0.z * dx + bdy) * v_e - vzl * q1.w;
                        vmy = (c1y + q0.w * dx) * v_e - vzl * q2.x;
                        acc[0] = vmx.x + vmx.y; acc[1] = vmy.x + vmy.y;
                        acc[4] = (dy * dy) * s_ve;
                    }
                    acc[2] = dot2(dx, dxve); acc[3] = dy * s_dxve;
                    acc[5] = MSUM ? s_ve : dot2(vis, vam);                  // (MSUM: o times the opacity gradient)
                    acc[6] = s_vzl; acc[7] = dot2(-vzl, dx); acc[8] = ndy * s_vzl;
                    acc[9] = dot2(w, vn2[0]); acc[10] = dot2(w, vn2[1]); acc[11] = dot2(w, vn2[2]);
                    acc[12] = dot2(w, vcol2[0]);
                    acc[13] = CD > 1 ? dot2(w, vcol2[CD > 1 ? 1 : 0]) : 0.f;
                    acc[14] = CD > 2 ? dot2(w, vcol2[CD > 2 ? 2 : 0]) : 0.f;
                    acc[15] = CD > 3 ? dot2(w, vcol2[CD > 3 ? 3 : 0]) : 0.f;
#pragma unroll
                    for (int ch = 0; ch < 16; ch++) accx[ch] = (NXQ > 0 && ch < NX) ? dot2(w, vcolx2[ch < NX ? ch : 0]) : 0.f;
                    if (ABS) { ab0 = fabsf(vmx.x) + fabsf(vmx.y); ab1 = fabsf(vmy.x) + fabsf(vmy.y); }
                } else if (NXQ == 0) {
                    // ---- one half: the same operations on component k alone
                    auto half = [&](auto kc) {
                        constexpr int k = decltype(kc)::value;
                        const float dx = q0.x - px2[k];
                        const float e = pair_exponent(q0.z, dx, bdy, ecc);
                        const float vis = __builtin_amdgcn_exp2f(e);
                        const float ov = q1.y * vis;
                        float am = ov;
                        lmask below = ~0ull;
                        if (can_clamp) {
                            asm volatile("; alpha_max clamp" ::);
                            am = __uint_as_float(min(amax_bits, __float_as_uint(ov)));
                            below = __ballot(ov <= amax);
                        }
                        const lmask okm = __ballot(i <= last[k]) & __ballot(e <= 0.f) & __ballot(am >= amin);
                        any_ok = okm;
                        const bool ok = __builtin_amdgcn_inverse_ballot_w64(okm);
                        const float a = ok ? am : 0.f;
                        const float ra = __builtin_amdgcn_rcpf(1.0f - a);
                        // (the two state vectors are updated with PACKED operations here too, the other half's component by a
                        // neutral operand: a write to one component of a register pair on one path and to the pair on another
                        // makes the compiler copy the whole state around the loop)
                        T2 *= k == 0 ? mk2(ra, 1.0f) : mk2(1.0f, ra);
                        const float Tk = T2[k];
                        const float w = a * Tk;
                        const float zp = tpy - q1.w * dx;
                        float dot = q3.x * vcol2[0][k];
                        if (CD > 1) dot += q3.y * vcol2[CD > 1 ? 1 : 0][k];
                        if (CD > 2) dot += q3.z * vcol2[CD > 2 ? 2 : 0][k];
                        if (CD > 3) dot += q3.w * vcol2[CD > 3 ? 3 : 0][k];
#pragma unroll
                        for (int q = 0; q < NXQ; q++)
                            dot += xq[q].x * vcolx2[4 * q][k] + xq[q].y * vcolx2[4 * q + 1][k] + xq[q].z * vcolx2[4 * q + 2][k] +
                                   xq[q].w * vcolx2[4 * q + 3][k];
                        dot += q2.y * vn2[0][k] + q2.z * vn2[1][k] + q2.w * vn2[2][k] + zp * vd2[k];
                        const float v_a = D2[k] * ra + Tk * dot;
                        D2 -= (k == 0 ? mk2(w, 0.f) : mk2(0.f, w)) * dot;
                        float vzl = w * vd2[k];
                        if (batch_has_median) {
                            asm volatile("; median gradient" ::);
                            vzl += (ok && i == medi[k]) ? vm2[k] : 0.f;
                        }
                        const bool un = __builtin_amdgcn_inverse_ballot_w64(okm & below);
                        const float vam = un ? v_a : 0.f;
                        const float v_e = MSUM ? ov * vam : (kLn2 * ov) * vam;
                        const float dxve = dx * v_e;
                        if (MSUM) {
                            acc[0] = dxve; acc[1] = dy * v_e;
                            acc[4] = dy * acc[1];
                        } else {
                            acc[0] = (2.0f * q0.z * dx + bdy) * v_e - vzl * q1.w;
                            acc[1] = (2.0f * q1.x * dy + q0.w * dx) * v_e - vzl * q2.x;
                            acc[4] = (dy * dy) * v_e;
                        }
                        acc[2] = dx * dxve; acc[3] = dy * dxve;
                        acc[5] = MSUM ? v_e : vis * vam;
                        acc[6] = vzl; acc[7] = -vzl * dx; acc[8] = -dy * vzl;
                        acc[9] = w * vn2[0][k]; acc[10] = w * vn2[1][k]; acc[11] = w * vn2[2][k];
                        acc[12] = w * vcol2[0][k];
                        acc[13] = CD > 1 ? w * vcol2[CD > 1 ? 1 : 0][k] : 0.f;
                        acc[14] = CD > 2 ? w * vcol2[CD > 2 ? 2 : 0][k] : 0.f;
                        acc[15] = CD > 3 ? w * vcol2[CD > 3 ? 3 : 0][k] : 0.f;
#pragma unroll
                        for (int ch = 0; ch < 16; ch++) accx[ch] = (NXQ > 0 && ch < NX) ? w * vcolx2[ch < NX ? ch : 0][k] : 0.f;
                        if (ABS) { ab0 = fabsf(acc[0]); ab1 = fabsf(acc[1]); }
                    };
                    if (sub == 1) half(KC<0>()); else half(KC<1>());
                }
            }
            const size_t slot = (size_t)islot;
            const uint32_t slot_off = (uint32_t)islot << 6;   // (atomic mode: rows * 64 B < 4 GiB, checked by the launcher)
            {
                const int jn = j > 0 ? j - 1 : 0;
                q0 = sm[jn]; q1 = sm[64 + jn]; q2 = sm[128 + jn]; q3 = sm[192 + jn];
                iv = sm_idx[jn]; islot = sm_slot[jn]; subv = sm_sub[jn];
            }
            if (any_ok != 0ull) {                  // (implies sub != 0: acc is set)
                const float r = wave_reduce16(acc, lane);
                if (ATOMIC) {
                    // one 64-byte contiguous no-return fp32 atomic per (band, Gaussian): 32-bit byte offset from a uniform base
#if defined(MISPLAT_DIAG_NO_ROW_ATOMICS)
                    // diagnostic build only (scripts/build_variant.sh): the reduced row is kept alive but leaves the wave nowhere --
                    // what the kernel costs without its memory-side atomics (verdict round 4, item 7)
                    asm volatile("" ::"v"(r * out_scale), "v"(slot_off + comp_off));
#else
                    if (writer) atomicAdd(reinterpret_cast<float*>(reinterpret_cast<char*>(slab_b) + (slot_off + comp_off)), r * out_scale);
#endif
                    touched_j |= 1ull << j;        // (flagged once per batch, below)
                    if (NXQ > 0) {
                        const float rx = wave_reduce16(accx, lane);
                        if (writer && comp < NX) atomicAdd(&v_featx[slot * NX + comp], rx);
                    }
                } else {
                    if (writer) slab_b[slot * MISPLAT_REC + comp] = r * out_scale;
                    if (lane == 1) valid_b[slot] = 1;
                }
                if (ABS) {
                    ab0 = wave_sum(ab0); ab1 = wave_sum(ab1);
                    if (ATOMIC) {
                        if (lane < 2) atomicAdd(&abs_b[slot * 2 + lane], lane == 0 ? ab0 : ab1);
                    } else if (lane == 0) { abs_b[slot * 2] = ab0; abs_b[slot * 2 + 1] = ab1; }
                }
            }
        }
        // row flags for the per-Gaussian backward kernels: one store instruction per batch (lane p flags the row of staged
        // entry p) instead of a predicated store per trip
        if (ATOMIC && P.touched && ((touched_j >> lane) & 1ull)) P.touched[sm_slot[lane]] = 1;
    }
}

// 16 lanes per Gaussian row r: v_grec[r][c] = sum over its intersections (ascending slot) and over
// the bands (ascending) of the VALID slab rows: a fixed order, so the result is bitwise
// reproducible.  Rows are fetched four slots at a time with unconditional loads (invalid rows are
// redirected to a zero row) so that the flag and row loads of a chunk are all in flight together.
__device__ const float kZeroRow[MISPLAT_REC] = {0.f};

template <int PLANES>
__global__ __launch_bounds__(256) void slab_reduce_kernel(int64_t n_rows, int64_t n_isects,
                                                          const int64_t* __restrict__ cum,
                                                          const int32_t* __restrict__ tiles_per_gauss,
                                                          const float* __restrict__ slab,
                                                          const float* __restrict__ slab_abs,
                                                          const uint8_t* __restrict__ valid,
                                                          float* __restrict__ v_grec,
                                                          float* __restrict__ v_abs) {
    const int c = threadIdx.x & 15;
    constexpr int U = 4;
    for (int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4; r < n_rows;
         r += ((int64_t)gridDim.x * blockDim.x) >> 4) {
        const int n = tiles_per_gauss[r];
        const int64_t base = cum[r];
        float acc = 0.f, acc_abs = 0.f;
        for (int j0 = 0; j0 < n; j0 += U) {
            bool f[U][PLANES];
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int w = 0; w < PLANES; w++)
                    f[u][w] = (j0 + u < n) && valid[(size_t)w * (size_t)n_isects + (size_t)(base + j0 + u)] != 0;
            float v[U][PLANES], va[U][PLANES];
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int w = 0; w < PLANES; w++) {
                    const size_t row = (size_t)w * (size_t)n_isects + (size_t)(base + j0 + u);
                    const float* src = f[u][w] ? slab + row * MISPLAT_REC : kZeroRow;
                    v[u][w] = src[c];
                    if (slab_abs != nullptr) {
                        const float* sa = f[u][w] ? slab_abs + row * 2 : kZeroRow;
                        va[u][w] = sa[c & 1];
                    }
                }
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int w = 0; w < PLANES; w++) {
                    acc += v[u][w];
                    if (slab_abs != nullptr) acc_abs += va[u][w];
                }
        }
        v_grec[r * MISPLAT_REC + c] = acc;
        if (slab_abs != nullptr && c < 2) v_abs[r * 2 + c] = acc_abs;
    }
}

// ---- a4: depth -> normal (camera_utils.py:191-279) + error map (rade_gs_model.py:212-214) ------
struct DN {
    int W, H;
    float fx, fy;
};
__device__ __forceinline__ void dn_point(const DN& d, const float* __restrict__ depth, int y, int x, float* p) {
    // rays_d = K^-1 [x+.5, y+.5, 1] with cx = W/2, cy = H/2 (camera_utils.py:211-241)
    const float z = depth[(size_t)y * d.W + x];
    p[0] = z * (((float)x + 0.5f) / d.fx - (float)d.W / (2.0f * d.fx));
    p[1] = z * (((float)y + 0.5f) / d.fy - (float)d.H / (2.0f * d.fy));
    p[2] = z;
}
// normal at interior pixel; returns cross-product vectors for the backward
__device__ __forceinline__ void dn_normal(const DN& d, const float* __restrict__ depth, int y, int x,
                                          float* a, float* b, float* cr, float& len, float* n) {
    float p0[3], p1[3], p2[3], p3[3];
    dn_point(d, depth, y + 1, x, p0); dn_point(d, depth, y - 1, x, p1);  // "dx": along rows (camera_utils.py:269)
    dn_point(d, depth, y, x + 1, p2); dn_point(d, depth, y, x - 1, p3);  // "dy": along columns (:270)
#pragma unroll
    for (int k = 0; k < 3; k++) { a[k] = p0[k] - p1[k]; b[k] = p2[k] - p3[k]; }
    cr[0] = a[1] * b[2] - a[2] * b[1];
    cr[1] = a[2] * b[0] - a[0] * b[2];
    cr[2] = a[0] * b[1] - a[1] * b[0];
    len = sqrtf(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]);
    const float inv = 1.0f / fmaxf(len, 1e-12f);  // F.normalize eps
    n[0] = cr[0] * inv; n[1] = cr[1] * inv; n[2] = cr[2] * inv;
}

__global__ __launch_bounds__(256) void depth_normal_fwd_kernel(DN d, const float* __restrict__ ed,
                                                               const float* __restrict__ md,
                                                               const float* __restrict__ nr,
                                                               float* __restrict__ normals2,
                                                               float* __restrict__ err) {
    const size_t P = (size_t)d.W * d.H;
    for (size_t pid = (size_t)blockIdx.x * blockDim.x + threadIdx.x; pid < P; pid += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(pid / d.W), x = (int)(pid - (size_t)y * d.W);
        const bool interior = x >= 1 && y >= 1 && x < d.W - 1 && y < d.H - 1;
        const float r0 = nr[pid * 3], r1 = nr[pid * 3 + 1], r2 = nr[pid * 3 + 2];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            float n[3] = {0.f, 0.f, 0.f};
            if (interior) {
                float a[3], b[3], cr[3], len;
                dn_normal(d, k == 0 ? ed : md, y, x, a, b, cr, len, n);
            }
            float* o = normals2 + ((size_t)k * P + pid) * 3;
            o[0] = n[0]; o[1] = n[1]; o[2] = n[2];
            err[(size_t)k * P + pid] = 1.0f - (r0 * n[0] + r1 * n[1] + r2 * n[2]);
        }
    }
}

// Backward of a4.  A pixel's depth feeds the normals of its four neighbours; v_a / v_b are what a normal's gradient
// sends back along its row / column difference (c = a x b: v_a = b x v_c, v_b = v_c x a).  The first version was
// pixel-parallel -- every pixel re-evaluated the normals of its four neighbours and its own from global memory, ten
// evaluations per pixel for the two depth maps: 94 us at 1080p against ~21 us of algorithmic traffic.  Here a 32 x 8
// tile stages both depth maps with a halo of 2, every pixel of the tile + halo 1 gets its adjoint vectors ONCE into
// LDS, and the owners gather the four they need: 1.33 evaluations per pixel and map.  Every element of the outputs is
// owned by one thread: no atomics.
constexpr int kDnTX = 32, kDnTY = 8;
__global__ __launch_bounds__(kDnTX * kDnTY) void depth_normal_bwd_tiled_kernel(
    DN d, const float* __restrict__ ed, const float* __restrict__ md, const float* __restrict__ nr,
    const float* __restrict__ v_n2, const float* __restrict__ v_err, float* __restrict__ v_ed, float* __restrict__ v_md,
    float* __restrict__ v_nr, const int accumulate) {
    constexpr int DW = kDnTX + 4, DH = kDnTY + 4;           // depth tile, halo 2
    constexpr int RW = kDnTX + 2, RH = kDnTY + 2;           // adjoint tile, halo 1
    __shared__ float dep[2][DH][DW];
    __shared__ float adj[2][9][RH][RW];                     // per map: v_a[3], v_b[3], v_err * n[3]
    const int x0 = blockIdx.x * kDnTX, y0 = blockIdx.y * kDnTY;
    const size_t P = (size_t)d.W * d.H;
    const int tid = threadIdx.x;
    for (int e = tid; e < 2 * DH * DW; e += kDnTX * kDnTY) {
        const int k = e / (DH * DW), r = e - k * DH * DW, ly = r / DW, lx = r - ly * DW;
        const int y = y0 + ly - 2, x = x0 + lx - 2;
        float z = 0.f;
        if (x >= 0 && y >= 0 && x < d.W && y < d.H) z = (k == 0 ? ed : md)[(size_t)y * d.W + x];
        dep[k][ly][lx] = z;
    }
    __syncthreads();
    // The backward multiplies by reciprocals where the forward divides (IEEE divisions are ~10 instructions each and an
    // adjoint evaluation had a dozen of them: the kernel was 77 % vector-ALU busy at 3.5 x its memory time; gradients are
    // compared at 1e-4, the forward's normals stay bit for bit what dn_normal computes).
    const float ifx = 1.0f / d.fx, ify = 1.0f / d.fy;
    const float cxr = (float)d.W * 0.5f * ifx, cyr = (float)d.H * 0.5f * ify;
    // (one adjoint evaluation per (map, pixel) item: 2 x 340 items on 256 threads are 2.7 rounds of one evaluation; both maps
    // of a pixel in one thread were 2 rounds of two, the second round a third full)
    for (int e2 = tid; e2 < 2 * RH * RW; e2 += kDnTX * kDnTY) {
        const int k = e2 / (RH * RW), e = e2 - k * (RH * RW);
        const int ly = e / RW, lx = e - ly * RW;
        const int yn = y0 + ly - 1, xn = x0 + lx - 1;
        const bool interior = xn >= 1 && yn >= 1 && xn < d.W - 1 && yn < d.H - 1;
        const size_t pid = interior ? (size_t)yn * d.W + xn : 0;
        float r0 = 0.f, r1 = 0.f, r2 = 0.f;
        if (interior && v_err) { r0 = nr[pid * 3]; r1 = nr[pid * 3 + 1]; r2 = nr[pid * 3 + 2]; }
        {
            float va[3] = {0.f, 0.f, 0.f}, vb[3] = {0.f, 0.f, 0.f}, ven[3] = {0.f, 0.f, 0.f};
            if (interior) {
                // points of the four neighbours (dn_point), from the staged tile: (ly, lx) of the adjoint tile is
                // (ly + 1, lx + 1) of the depth tile
                float pt[4][3];
                const int ny[4] = {yn + 1, yn - 1, yn, yn}, nx[4] = {xn, xn, xn + 1, xn - 1};
                const int dy[4] = {ly + 2, ly, ly + 1, ly + 1}, dx[4] = {lx + 1, lx + 1, lx + 2, lx};
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float z = dep[k][dy[q]][dx[q]];
                    pt[q][0] = z * (((float)nx[q] + 0.5f) * ifx - cxr);
                    pt[q][1] = z * (((float)ny[q] + 0.5f) * ify - cyr);
                    pt[q][2] = z;
                }
                float a[3], b[3], cr[3], n[3];
#pragma unroll
                for (int q = 0; q < 3; q++) { a[q] = pt[0][q] - pt[1][q]; b[q] = pt[2][q] - pt[3][q]; }
                cr[0] = a[1] * b[2] - a[2] * b[1];
                cr[1] = a[2] * b[0] - a[0] * b[2];
                cr[2] = a[0] * b[1] - a[1] * b[0];
                const float len = sqrtf(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]);
                const float inv = 1.0f / fmaxf(len, 1e-12f);
                n[0] = cr[0] * inv; n[1] = cr[1] * inv; n[2] = cr[2] * inv;
                float vn[3] = {0.f, 0.f, 0.f};
                if (v_n2) { const float* sp = v_n2 + ((size_t)k * P + pid) * 3; vn[0] = sp[0]; vn[1] = sp[1]; vn[2] = sp[2]; }
                if (v_err) {
                    const float ve = v_err[(size_t)k * P + pid];
                    vn[0] -= ve * r0; vn[1] -= ve * r1; vn[2] -= ve * r2;
                    ven[0] = ve * n[0]; ven[1] = ve * n[1]; ven[2] = ve * n[2];
                }
                float vc[3];
                if (len > 1e-12f) {
                    const float dot = n[0] * vn[0] + n[1] * vn[1] + n[2] * vn[2];
#pragma unroll
                    for (int q = 0; q < 3; q++) vc[q] = (vn[q] - n[q] * dot) * inv;          // (inv = 1 / len here)
                } else {
#pragma unroll
                    for (int q = 0; q < 3; q++) vc[q] = vn[q] * 1e12f;
                }
                va[0] = b[1] * vc[2] - b[2] * vc[1]; va[1] = b[2] * vc[0] - b[0] * vc[2]; va[2] = b[0] * vc[1] - b[1] * vc[0];
                vb[0] = vc[1] * a[2] - vc[2] * a[1]; vb[1] = vc[2] * a[0] - vc[0] * a[2]; vb[2] = vc[0] * a[1] - vc[1] * a[0];
            }
#pragma unroll
            for (int q = 0; q < 3; q++) { adj[k][q][ly][lx] = va[q]; adj[k][3 + q][ly][lx] = vb[q]; adj[k][6 + q][ly][lx] = ven[q]; }
        }
    }
    __syncthreads();
    const int lx = tid % kDnTX, ly = tid / kDnTX;
    const int x = x0 + lx, y = y0 + ly;
    if (x >= d.W || y >= d.H) return;
    const size_t pid = (size_t)y * d.W + x;
    const float rx = ((float)x + 0.5f) * ifx - cxr;
    const float ry = ((float)y + 0.5f) * ify - cyr;
    float vnr[3] = {0.f, 0.f, 0.f};
    const int cy = ly + 1, cx = lx + 1;                      // own position in the adjoint tile
#pragma unroll
    for (int k = 0; k < 2; k++) {
        float g[3] = {0.f, 0.f, 0.f};
        // P(y,x) is the +row neighbour of (y-1,x), the -row neighbour of (y+1,x), the +col neighbour of (y,x-1) and
        // the -col neighbour of (y,x+1); adjoints of pixels outside the image are zero
#pragma unroll
        for (int q = 0; q < 3; q++) g[q] += adj[k][q][cy - 1][cx];
#pragma unroll
        for (int q = 0; q < 3; q++) g[q] -= adj[k][q][cy + 1][cx];
#pragma unroll
        for (int q = 0; q < 3; q++) g[q] += adj[k][3 + q][cy][cx - 1];
#pragma unroll
        for (int q = 0; q < 3; q++) g[q] -= adj[k][3 + q][cy][cx + 1];
        float* dst = k == 0 ? v_ed : v_md;
        const float gd = g[0] * rx + g[1] * ry + g[2];
        dst[pid] = accumulate ? dst[pid] + gd : gd;
        vnr[0] -= adj[k][6][cy][cx]; vnr[1] -= adj[k][7][cy][cx]; vnr[2] -= adj[k][8][cy][cx];
    }
    if (accumulate) { vnr[0] += v_nr[pid * 3]; vnr[1] += v_nr[pid * 3 + 1]; vnr[2] += v_nr[pid * 3 + 2]; }
    v_nr[pid * 3] = vnr[0]; v_nr[pid * 3 + 1] = vnr[1]; v_nr[pid * 3 + 2] = vnr[2];
}

// ---- launch order of the compositing kernels: longest units first, per XCD strip ---------------------------
// A compositing launch is ~3 "rounds" of one-wave workgroups whose durations differ by 2x; dealt in tile order,
// the last round runs at low occupancy (a 40 % / 10 % tail of the forward / backward).  Dealt longest-first the
// tail disappears.  The cost of a unit is what the forward measured (unit_work = staged Gaussians composited);
// workgroup b runs on XCD b & 7 (round-robin dispatch), so strip x = units [x * per, (x + 1) * per) keeps its XCD
// (its Gaussian records stay in that L2) and is ordered by descending work inside: perm[rank * 8 + x] = unit.
// Counting sort on 256 work classes -- an approximate order is all a greedy scheduler needs.
// sel (or NULL): perm is a table of records of `stride` words; the permutation goes into record sel[0], whose header takes
// the tag sel[2..3] and becomes valid -- for the launches after this one (the backward of the same call) and for the next
// call with the same cameras.
__global__ __launch_bounds__(1024) void unit_order_kernel(int units, int per, const int32_t* __restrict__ work,
                                                          int32_t* __restrict__ perm, int32_t* __restrict__ sel, int stride,
                                                          int slots, const float* __restrict__ reach = nullptr,
                                                          int n_tiles = 0, int pivot_off = 0) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t wmax[16];
    const int x = blockIdx.x;
    if (sel) {
        if ((unsigned)sel[0] >= (unsigned)slots) return;       // (a selector nobody wrote: leave the table alone)
        int32_t* rec = perm + (size_t)sel[0] * stride;
        perm = rec + MISPLAT_ORDER_HEADER;
        if (x == 0 && threadIdx.x == 0) { rec[0] = sel[2]; rec[1] = sel[3]; rec[2] = 1; rec[3] = reach ? 1 : 0; }
        // per-tile depth pivots for the view's next visit (front-only ordering, csrc/binning.hip): the deeper of the two
        // bands' reach
        if (reach)
            for (int t = x * 1024 + (int)threadIdx.x; t < n_tiles; t += 8 * 1024) {
                float pv = 0.f;
#pragma unroll
                for (int b = 0; b < kBandsPerTile; b++) pv = fmaxf(pv, reach[t * kBandsPerTile + b]);
                rec[pivot_off + t] = __float_as_int(pv);
            }
    }
    const int lo = x * per, hi = min(lo + per, units);
    // (the work counts are a hint from another launch: whatever they hold -- negative, huge, changing while this kernel
    // reads them -- the result is a permutation of the strip's units and every access stays inside hist[] / perm[])
    int mx = 0;
    for (int u = lo + threadIdx.x; u < hi; u += 1024) mx = max(mx, max(work[u], 0));
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) mx = max(mx, __shfl_xor(mx, m));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = (uint32_t)mx;
    if (threadIdx.x < 256) hist[threadIdx.x] = 0u;
    __syncthreads();
    uint32_t top = 1u;
#pragma unroll
    for (int w = 0; w < 16; w++) top = max(top, wmax[w]);
    const float scale = 255.0f / (float)top;
    auto cls_of = [&](int w) { return 255 - min(255, (int)((float)min(max(w, 0), (int)top) * scale)); };
    for (int u = lo + threadIdx.x; u < hi; u += 1024) atomicAdd(&hist[cls_of(work[u])], 1u);
    __syncthreads();
    // exclusive scan of the 256 classes (class 0 = heaviest): waves 0..3, one class per lane
    uint32_t v = 0u, incl = 0u;
    if (threadIdx.x < 256) {
        v = hist[threadIdx.x];
        incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off);
            if ((int)(threadIdx.x & 63) >= off) incl += o;
        }
        if ((threadIdx.x & 63) == 63) wmax[threadIdx.x >> 6] = incl;
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        uint32_t carry = 0u;
        for (int w = 0; w < (int)(threadIdx.x >> 6); w++) carry += wmax[w];
        hist[threadIdx.x] = carry + incl - v;
    }
    __syncthreads();
    for (int u = lo + threadIdx.x; u < hi; u += 1024) {
        const uint32_t rank = atomicAdd(&hist[cls_of(work[u])], 1u);
        if (rank < (uint32_t)per) perm[rank * 8 + x] = u;
    }
    for (int r = (hi > lo ? hi - lo : 0) + threadIdx.x; r < per; r += 1024) perm[r * 8 + x] = units;   // padding: no unit
    if (sel && x == 0) {                        // (the record is complete when every strip is: the flag is only read by
        __syncthreads();                        //  LATER launches, which see all of this kernel's stores)
        if (threadIdx.x == 0) sel[1] = 1;
    }
}

inline int check_launch() { return hipGetLastError() == hipSuccess ? MISPLAT_OK : MISPLAT_ELAUNCH; }
inline int grid_for(int64_t n, int block) {
    int64_t b = (n + block - 1) / block;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (int)b;
}
inline bool params_ok(const misplat_params* p) {
    return p && p->tile_size == MISPLAT_TILE && p->n_cams >= 1 && p->width >= 1 && p->height >= 1 &&
           p->ed_slot >= -1 && p->ed_slot <= 19 &&
           p->tile_w == (p->width + MISPLAT_TILE - 1) / MISPLAT_TILE &&
           p->tile_h == (p->height + MISPLAT_TILE - 1) / MISPLAT_TILE;
}

}  // namespace

// A tile is covered by two independent wavefronts ("bands" of 16 x 8 pixels, two pixels per lane): measured against one
// (16 x 16, four pixels per lane: 0.513 against 0.481 ms backward) and four (0.605) bands per tile, DESIGN.md section 6.
constexpr int kPpl = 2, kBands = 4 / kPpl;

extern "C" int misplat_blend_planes(const misplat_params* p) { return p ? kBands : 0; }

extern "C" int misplat_unit_order(const misplat_params* p, const int32_t* unit_work, int32_t* unit_perm,
                                  misplat_stream_t stream) {
    if (!params_ok(p) || !unit_work || !unit_perm) return MISPLAT_EINVAL;
    const int units = p->tile_w * p->tile_h * p->n_cams * kBands;
    const int per = (units + 7) >> 3;
    hipLaunchKernelGGL(unit_order_kernel, dim3(8), dim3(1024), 0, (hipStream_t)stream, units, per, unit_work, unit_perm,
                       (int32_t*)nullptr, 0, 0);
    return check_launch();
}

int misplat_internal::unit_order_table(const misplat_params* p, const int32_t* unit_work, int32_t* table,
                                       int32_t* sel, int32_t stride, int32_t slots, const float* unit_reach, hipStream_t s) {
    if (!params_ok(p) || !unit_work || !table || !sel || slots < 1) return MISPLAT_EINVAL;
    const int n_tiles = p->tile_w * p->tile_h * p->n_cams;
    const int units = n_tiles * kBands;
    const int per = (units + 7) >> 3;
    const int pivot_off = MISPLAT_ORDER_HEADER + 8 * per;
    if (stride < pivot_off + (unit_reach ? n_tiles : 0)) return MISPLAT_EINVAL;
    hipLaunchKernelGGL(unit_order_kernel, dim3(8), dim3(1024), 0, s, units, per, unit_work, table, sel, (int)stride, (int)slots,
                       unit_reach, n_tiles, pivot_off);
    return check_launch();
}

extern "C" int misplat_blend_fwd(const misplat_params* p, int32_t color_dim, const float* Ks,
                                 const float* grec, const int32_t* flatten_ids, const int32_t* offsets,
                                 int64_t n_isects, float* render, float* alpha, float* exp_depth,
                                 float* med_depth, float* normal, int32_t* last_ids,
                                 int32_t* median_ids, misplat_stream_t stream) {
    if (!params_ok(p) || n_isects < 0 || n_isects > 0x7fffffffLL) return MISPLAT_EINVAL;
    const int total = p->tile_w * p->tile_h * p->n_cams * kBands;
    const int grid = ((total + 7) / 8) * 8;
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH_FWD(CD_)                                                                                    \
    hipLaunchKernelGGL((blend_fwd_kernel<CD_, kPpl>), dim3(grid), dim3(64), 0, s, *p, Ks, (const float4*)grec, \
                       flatten_ids, offsets, n_isects, render, alpha, exp_depth, med_depth, normal, last_ids,  \
                       median_ids)
    if (color_dim == 1) LAUNCH_FWD(1);
    else if (color_dim == 2) LAUNCH_FWD(2);
    else if (color_dim == 3) LAUNCH_FWD(3);
    else if (color_dim == 4) LAUNCH_FWD(4);
    else return MISPLAT_EINVAL;
#undef LAUNCH_FWD
    return check_launch();
}

// The same forward with ON-DEMAND SH colours (see LazyColour): the colour slots of grec must hold the "unset" pattern
// (misplat_project_pack_fwd with lazy_rows) and are filled here for the records that are staged past their cull; sh_aux
// (or NULL) receives the Jacobians of those records.  K = 16 coefficients per Gaussian ([N,16,3], or [N,3] + [N,15,3]).
extern "C" int misplat_blend_fwd_lazy(const misplat_params* p, int32_t color_dim, const float* Ks, float* grec,
                                      const int32_t* flatten_ids, const int32_t* offsets, int64_t n_isects,
                                      float* render, float* alpha, float* exp_depth, float* med_depth, float* normal,
                                      int32_t* last_ids, int32_t* median_ids, const float* means, const float* viewmats,
                                      const float* coeffs, const float* coeffs_rest, int32_t sh_degree,
                                      int32_t depth_channel, const float* depths, float* sh_aux,
                                      misplat_stream_t stream) {
    return misplat_internal::blend_fwd_lazy(p, color_dim, Ks, grec, flatten_ids, offsets, n_isects, render, alpha, exp_depth,
                                            med_depth, normal, last_ids, median_ids, means, viewmats, coeffs, coeffs_rest,
                                            sh_degree, depth_channel, depths, sh_aux, nullptr, (hipStream_t)stream);
}

int misplat_internal::blend_fwd_lazy(const misplat_params* p, int32_t color_dim, const float* Ks, float* grec,
                                     const int32_t* flatten_ids, const int32_t* offsets, int64_t n_isects, float* render,
                                     float* alpha, float* exp_depth, float* med_depth, float* normal, int32_t* last_ids,
                                     int32_t* median_ids, const float* means, const float* viewmats, const float* coeffs,
                                     const float* coeffs_rest, int32_t sh_degree, int32_t depth_channel,
                                     const float* depths, float* sh_aux, float* rows_on_touch, hipStream_t s) {
    if (!params_ok(p) || n_isects < 0 || n_isects > 0x7fffffffLL || !means || !viewmats || !coeffs || sh_degree < 0 ||
        sh_degree > 3 || (depth_channel && !depths) || color_dim < 3 || color_dim > 4)
        return MISPLAT_EINVAL;
    const int total = p->tile_w * p->tile_h * p->n_cams * kBands;
    if (((uintptr_t)rows_on_touch) & 15) return MISPLAT_EINVAL;
    const int grid = ((total + 7) / 8) * 8;
    LazyColour lz;
    lz.means = means; lz.coeffs = coeffs; lz.coeffs_rest = coeffs_rest; lz.depths = depths; lz.viewmats = viewmats;
    lz.grec_rw = (float4*)grec; lz.sh_aux = sh_aux; lz.v_rows = (float4*)rows_on_touch; lz.ccx = lz.ccy = lz.ccz = 0.f;
    lz.deg = sh_degree; lz.depth_channel = depth_channel; lz.n_gauss = p->n_gauss;
    lz.features = nullptr; lz.v_rows_x = nullptr; lz.n_feat = 0; lz.nxq = 0;
    if (color_dim == 3)
        hipLaunchKernelGGL((blend_fwd_kernel<3, 2, 0, true>), dim3(grid), dim3(64), 0, s, *p, Ks, (const float4*)grec, flatten_ids,
                           offsets, n_isects, render, alpha, exp_depth, med_depth, normal, last_ids, median_ids,
                           (const float4*)nullptr, 3, lz);
    else
        hipLaunchKernelGGL((blend_fwd_kernel<4, 2, 0, true>), dim3(grid), dim3(64), 0, s, *p, Ks, (const float4*)grec, flatten_ids,
                           offsets, n_isects, render, alpha, exp_depth, med_depth, normal, last_ids, median_ids,
                           (const float4*)nullptr, 4, lz);
    return check_launch();
}

extern "C" int misplat_blend_bwd(const misplat_params* p, int32_t color_dim, const float* Ks,
                                 const float* grec, const int32_t* flatten_ids,
                                 const int32_t* slots_sorted, const int32_t* offsets, int64_t n_isects,
                                 const float* alpha, const int32_t* last_ids, const int32_t* median_ids,
                                 const float* render, const float* v_render, const float* v_alpha,
                                 const float* v_exp_depth,
                                 const float* v_med_depth, const float* v_normal, float* slab,
                                 float* slab_abs, uint8_t* slab_valid, misplat_stream_t stream) {
    if (!params_ok(p) || n_isects < 0 || n_isects > 0x7fffffffLL || !slab_valid) return MISPLAT_EINVAL;
    if (n_isects == 0) return MISPLAT_OK;
    const int planes = kBands;
    const int total = p->tile_w * p->tile_h * p->n_cams * planes;
    const int grid = ((total + 7) / 8) * 8;
    hipStream_t s = (hipStream_t)stream;
    if (misplat_internal::fill_bytes(slab_valid, (size_t)n_isects * planes, 0u, s) != MISPLAT_OK) return MISPLAT_ELAUNCH;
#define LAUNCH_BWD(CD_, ABS_)                                                                                \
    hipLaunchKernelGGL((blend_bwd_kernel<CD_, kPpl, ABS_, false>), dim3(grid), dim3(64), 0, s, *p, Ks,         \
                       (const float4*)grec, flatten_ids, slots_sorted, offsets, n_isects, alpha, last_ids,     \
                       median_ids, render, v_render, v_alpha, v_exp_depth, v_med_depth, v_normal, slab,        \
                       slab_abs, slab_valid)
#define DISPATCH_BWD(CD_) do { if (slab_abs) LAUNCH_BWD(CD_, true); else LAUNCH_BWD(CD_, false); } while (0)
    if (color_dim == 1) DISPATCH_BWD(1);
    else if (color_dim == 2) DISPATCH_BWD(2);
    else if (color_dim == 3) DISPATCH_BWD(3);
    else if (color_dim == 4) DISPATCH_BWD(4);
    else return MISPLAT_EINVAL;
#undef DISPATCH_BWD
#undef LAUNCH_BWD
    return check_launch();
}

// misplat_blend_bwd_atomic + background fills (internal.h): 512 one-wave workgroups (a multiple of 8: the unit -> XCD map
// stays) behind the last unit of the grid, or in front of the first one when the fill is too large to hide in the
// kernel's tail (placement and count measured: see enqueue_backward in raster.hip).
int misplat_internal::blend_bwd_atomic(const misplat_params* p, int32_t color_dim, const float* Ks, const float* grec,
                                       const int32_t* flatten_ids, const int32_t* offsets, int64_t n_isects,
                                       const float* alpha, const int32_t* last_ids, const int32_t* median_ids,
                                       const float* render, const float* v_render, const float* v_alpha,
                                       const float* v_exp_depth, const float* v_med_depth, const float* v_normal,
                                       float* v_grec, float* v_abs, int32_t v_grec_is_zero, const FillList* fills,
                                       hipStream_t s, bool mean_sums) {
    if (!params_ok(p) || n_isects < 0 || n_isects > 0x7fffffffLL || !v_grec || (mean_sums && v_abs)) return MISPLAT_EINVAL;
    const size_t rows = (size_t)p->n_gauss * p->n_cams;
    if (rows == 0) return MISPLAT_OK;
    if (rows >= ((size_t)1 << 26)) return MISPLAT_EINVAL;   // the kernels address gradient rows with 32-bit byte offsets (64 B each)
    // v_grec_is_zero: bit 0 = v_grec, bit 1 = v_abs have been cleared by the caller
    if (!(v_grec_is_zero & 1) && fill_bytes(v_grec, rows * MISPLAT_REC * sizeof(float), 0u, s) != MISPLAT_OK) return MISPLAT_ELAUNCH;
    if (v_abs && !(v_grec_is_zero & 2) && fill_bytes(v_abs, rows * 2 * sizeof(float), 0u, s) != MISPLAT_OK) return MISPLAT_ELAUNCH;
    FillList F = {};
    if (fills) {
        if (fills->count < 0 || fills->count > 8) return MISPLAT_EINVAL;
        for (int k = 0; k < fills->count; k++)
            if (!fills->p[k] || (((uintptr_t)fills->p[k]) & 15) || fills->n[k] < 0) return MISPLAT_EINVAL;
        const bool in_kernel = n_isects > 0 && color_dim >= 1 && color_dim <= 4;
        if (in_kernel) {
            F = *fills;
            F.blocks = kFillBlocks;
            F.at_head = (int64_t)rows >= kFillHeadRows;
        } else {
            for (int k = 0; k < fills->count; k++) {
                const int rf = zero_fill(fills->p[k], fills->n[k], 0, s);
                if (rf != MISPLAT_OK) return rf;
            }
        }
    }
    if (n_isects == 0) return MISPLAT_OK;
    const int total = p->tile_w * p->tile_h * p->n_cams * kBands;
    const int grid = ((total + 7) / 8) * 8 + F.blocks;
#define LAUNCH_BWDA(CD_, ABS_)                                                                               \
    hipLaunchKernelGGL((blend_bwd_kernel<CD_, kPpl, ABS_, true>), dim3(grid), dim3(64), 0, s, *p, Ks,          \
                       (const float4*)grec, flatten_ids, (const int32_t*)nullptr, offsets, n_isects, alpha,    \
                       last_ids, median_ids, render, v_render, v_alpha, v_exp_depth, v_med_depth, v_normal,    \
                       v_grec, v_abs, (uint8_t*)nullptr, (const float4*)nullptr, (float*)nullptr, CD_, F)
#define LAUNCH_BWDA_MSUM(CD_)                                                                                \
    hipLaunchKernelGGL((blend_bwd_kernel<CD_, kPpl, false, true, 0, true>), dim3(grid), dim3(64), 0, s, *p, Ks, \
                       (const float4*)grec, flatten_ids, (const int32_t*)nullptr, offsets, n_isects, alpha,    \
                       last_ids, median_ids, render, v_render, v_alpha, v_exp_depth, v_med_depth, v_normal,    \
                       v_grec, v_abs, (uint8_t*)nullptr, (const float4*)nullptr, (float*)nullptr, CD_, F)
#define DISPATCH_BWDA(CD_) do { if (v_abs) LAUNCH_BWDA(CD_, true); else if (mean_sums) LAUNCH_BWDA_MSUM(CD_); else LAUNCH_BWDA(CD_, false); } while (0)
    if (color_dim == 1) DISPATCH_BWDA(1);
    else if (color_dim == 2) DISPATCH_BWDA(2);
    else if (color_dim == 3) DISPATCH_BWDA(3);
    else if (color_dim == 4) DISPATCH_BWDA(4);
    else return MISPLAT_EINVAL;
#undef DISPATCH_BWDA
#undef LAUNCH_BWDA_MSUM
#undef LAUNCH_BWDA
    return check_launch();
}

// Same backward, but every (band, Gaussian) row is added straight into v_grec[C*N,16] (and
// v_abs[C*N,2]) with no-return fp32 atomics -- no slab, no second kernel; the sums then depend on
// arrival order (not bitwise reproducible).  v_grec / v_abs are zeroed here on `stream`.
extern "C" int misplat_blend_bwd_atomic(const misplat_params* p, int32_t color_dim, const float* Ks,
                                        const float* grec, const int32_t* flatten_ids,
                                        const int32_t* offsets, int64_t n_isects, const float* alpha,
                                        const int32_t* last_ids, const int32_t* median_ids,
                                        const float* render, const float* v_render, const float* v_alpha,
                                        const float* v_exp_depth, const float* v_med_depth,
                                        const float* v_normal, float* v_grec, float* v_abs,
                                        int32_t v_grec_is_zero, misplat_stream_t stream) {
    return misplat_internal::blend_bwd_atomic(p, color_dim, Ks, grec, flatten_ids, offsets, n_isects, alpha, last_ids,
                                              median_ids, render, v_render, v_alpha, v_exp_depth, v_med_depth, v_normal,
                                              v_grec, v_abs, v_grec_is_zero, nullptr, (hipStream_t)stream, false);
}

// ---- N-D colours: D' = n_channels in 5..20 composited in ONE pass.
// Channels 0..3 ride in the record's colour slots, channels 4.. in featx[C*N, 4*nxq] (zero padded).
extern "C" int misplat_blend_fwd_x(const misplat_params* p, int32_t n_channels, int32_t nxq, const float* Ks,
                                   const float* grec, const float* featx, const int32_t* flatten_ids,
                                   const int32_t* offsets, int64_t n_isects, float* render, float* alpha,
                                   float* exp_depth, float* med_depth, float* normal, int32_t* last_ids,
                                   int32_t* median_ids, misplat_stream_t stream) {
    if (!params_ok(p) || n_isects < 0 || n_isects > 0x7fffffffLL || nxq < 1 || nxq > 4 || n_channels < 5 ||
        n_channels > 4 + 4 * nxq || !featx)
        return MISPLAT_EINVAL;
    const int total = p->tile_w * p->tile_h * p->n_cams * kBands;
    const int grid = ((total + 7) / 8) * 8;
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH_FWDX(NXQ_)                                                                                   \
    hipLaunchKernelGGL((blend_fwd_kernel<4, 2, NXQ_>), dim3(grid), dim3(64), 0, s, *p, Ks, (const float4*)grec, \
                       flatten_ids, offsets, n_isects, render, alpha, exp_depth, med_depth, normal, last_ids,  \
                       median_ids, (const float4*)featx, n_channels)
    if (nxq == 1) LAUNCH_FWDX(1);
    else if (nxq == 2) LAUNCH_FWDX(2);
    else if (nxq == 3) LAUNCH_FWDX(3);
    else LAUNCH_FWDX(4);
#undef LAUNCH_FWDX
    return check_launch();
}

// misplat_blend_fwd_x with on-demand records (the features model's call in training): the colour slots of grec start UNSET,
// the first wave that stages a record past its cull evaluates SH -> max(. + 0.5, 0), takes channel 3 from features[g][0] and
// clears the record's gradient rows (rows_on_touch / rows_on_touch_x or NULL); channels 4.. are read from features[g][1..]
// (+ depths[g]) by every wave that stages the record -- there is no featx array in this mode.
int misplat_internal::blend_fwd_x_lazy(const misplat_params* p, int32_t n_channels, int32_t nxq, const float* Ks, float* grec,
                                       const int32_t* flatten_ids, const int32_t* offsets, int64_t n_isects,
                                       float* render, float* alpha, float* exp_depth, float* med_depth, float* normal,
                                       int32_t* last_ids, int32_t* median_ids, const float* means, const float* viewmats,
                                       const float* coeffs, const float* coeffs_rest, int32_t sh_degree, int32_t depth_channel,
                                       const float* depths, const float* features, int32_t n_feat, float* rows_on_touch,
                                       float* rows_on_touch_x, hipStream_t s) {
    if (!params_ok(p) || n_isects < 0 || n_isects > 0x7fffffffLL || nxq < 3 || nxq > 4 || n_channels < 5 ||
        n_channels > 4 + 4 * nxq || !means || !viewmats || !coeffs || sh_degree < 0 || sh_degree > 3 || !features ||
        n_feat < 1 || 3 + n_feat + (depth_channel ? 1 : 0) != n_channels || (depth_channel && !depths) || p->n_cams != 1 ||
        ((((uintptr_t)rows_on_touch) | ((uintptr_t)rows_on_touch_x)) & 15))
        return MISPLAT_EINVAL;
    const int total = p->tile_w * p->tile_h * p->n_cams * kBands;
    const int grid = ((total + 7) / 8) * 8;
    LazyColour lz;
    lz.means = means; lz.coeffs = coeffs; lz.coeffs_rest = coeffs_rest; lz.depths = depths; lz.viewmats = viewmats;
    lz.grec_rw = (float4*)grec; lz.sh_aux = nullptr; lz.v_rows = (float4*)rows_on_touch; lz.ccx = lz.ccy = lz.ccz = 0.f;
    lz.deg = sh_degree; lz.depth_channel = depth_channel; lz.n_gauss = p->n_gauss;
    lz.features = features; lz.v_rows_x = (float4*)rows_on_touch_x; lz.n_feat = n_feat; lz.nxq = nxq;
#define LAUNCH_FWDXL(NXQ_)                                                                                            \
    hipLaunchKernelGGL((blend_fwd_kernel<4, 2, NXQ_, true>), dim3(grid), dim3(64), 0, s, *p, Ks, (const float4*)grec,  \
                       flatten_ids, offsets, n_isects, render, alpha, exp_depth, med_depth, normal, last_ids,          \
                       median_ids, (const float4*)nullptr, n_channels, lz)
    if (nxq == 3) LAUNCH_FWDXL(3);                 // (D' = 16 and 17: the features model without / with the depth channel)
    else LAUNCH_FWDXL(4);
#undef LAUNCH_FWDXL
    return check_launch();
}

extern "C" int misplat_blend_bwd_x_atomic(const misplat_params* p, int32_t n_channels, int32_t nxq, const float* Ks,
                                          const float* grec, const float* featx, const int32_t* flatten_ids,
                                          const int32_t* offsets, int64_t n_isects, const float* alpha,
                                          const int32_t* last_ids, const int32_t* median_ids, const float* render,
                                          const float* v_render, const float* v_alpha, const float* v_exp_depth,
                                          const float* v_med_depth, const float* v_normal, float* v_grec,
                                          float* v_featx, float* v_abs, misplat_stream_t stream) {
    return misplat_internal::blend_bwd_x_atomic(p, n_channels, nxq, Ks, grec, featx, flatten_ids, offsets, n_isects, alpha, last_ids,
                                                median_ids, render, v_render, v_alpha, v_exp_depth, v_med_depth, v_normal, v_grec,
                                                v_featx, v_abs, 0, nullptr, (hipStream_t)stream, nullptr, 0, 0, nullptr, false);
}

int misplat_internal::blend_bwd_x_atomic(const misplat_params* p, int32_t n_channels, int32_t nxq, const float* Ks,
                                         const float* grec, const float* featx, const int32_t* flatten_ids,
                                         const int32_t* offsets, int64_t n_isects, const float* alpha,
                                         const int32_t* last_ids, const int32_t* median_ids, const float* render,
                                         const float* v_render, const float* v_alpha, const float* v_exp_depth,
                                         const float* v_med_depth, const float* v_normal, float* v_grec,
                                         float* v_featx, float* v_abs, int32_t zero_flags, const FillList* fills,
                                         hipStream_t stream, const float* features, int32_t n_feat, int32_t depth_channel,
                                         const float* depths, bool mean_sums) {
    // featx == NULL: channels 4.. straight from features [N, n_feat] (+ depths): the forward ran with on-demand records
    if (!params_ok(p) || n_isects < 0 || n_isects > 0x7fffffffLL || nxq < 1 || nxq > 4 || n_channels < 5 ||
        n_channels > 4 + 4 * nxq || !v_grec || !v_featx || (mean_sums && v_abs))
        return MISPLAT_EINVAL;
    if (!featx && (!features || n_feat < 1 || 3 + n_feat + (depth_channel ? 1 : 0) != n_channels || (depth_channel && !depths) ||
                   p->n_cams != 1))
        return MISPLAT_EINVAL;
    FeatSrc fsrc;
    fsrc.features = featx ? nullptr : features; fsrc.depths = depths; fsrc.n_feat = n_feat; fsrc.depth_channel = depth_channel;
    fsrc.n_gauss = p->n_gauss;
    hipStream_t s = (hipStream_t)stream;
    const size_t rows = (size_t)p->n_gauss * p->n_cams;
    if (rows == 0) return MISPLAT_OK;
    if (rows >= ((size_t)1 << 26)) return MISPLAT_EINVAL;   // (32-bit byte offsets into the gradient rows)
    if (!(zero_flags & 1) && misplat_internal::fill_bytes(v_grec, rows * MISPLAT_REC * sizeof(float), 0u, s) != MISPLAT_OK) return MISPLAT_ELAUNCH;
    if (!(zero_flags & 4) && misplat_internal::fill_bytes(v_featx, rows * 4 * nxq * sizeof(float), 0u, s) != MISPLAT_OK) return MISPLAT_ELAUNCH;
    if (v_abs && !(zero_flags & 2) && misplat_internal::fill_bytes(v_abs, rows * 2 * sizeof(float), 0u, s) != MISPLAT_OK) return MISPLAT_ELAUNCH;
    FillList F = {};
    if (fills) {                                    // (as blend_bwd_atomic: zeros for the next kernel, written by extra workgroups)
        if (fills->count < 0 || fills->count > 8) return MISPLAT_EINVAL;
        for (int k = 0; k < fills->count; k++)
            if (!fills->p[k] || (((uintptr_t)fills->p[k]) & 15) || fills->n[k] < 0) return MISPLAT_EINVAL;
        if (n_isects > 0) {
            F = *fills;
            F.blocks = kFillBlocks;
            F.at_head = (int64_t)rows >= kFillHeadRows;
        } else {
            for (int k = 0; k < fills->count; k++) {
                const int rf = zero_fill(fills->p[k], fills->n[k], 0, s);
                if (rf != MISPLAT_OK) return rf;
            }
        }
    }
    if (n_isects == 0) return MISPLAT_OK;
    const int total = p->tile_w * p->tile_h * p->n_cams * kBands;
    const int grid = ((total + 7) / 8) * 8 + F.blocks;
#define LAUNCH_BWDX(NXQ_, ABS_, MSUM_)                                                                       \
    hipLaunchKernelGGL((blend_bwd_kernel<4, 2, ABS_, true, NXQ_, MSUM_>), dim3(grid), dim3(64), 0, s, *p, Ks,   \
                       (const float4*)grec, flatten_ids, (const int32_t*)nullptr, offsets, n_isects, alpha,    \
                       last_ids, median_ids, render, v_render, v_alpha, v_exp_depth, v_med_depth, v_normal,    \
                       v_grec, v_abs, (uint8_t*)nullptr, (const float4*)featx, v_featx, n_channels, F, fsrc)
#define DISPATCH_BWDX(NXQ_)                     \
    do {                                        \
        if (v_abs) LAUNCH_BWDX(NXQ_, true, false);     \
        else if (mean_sums) LAUNCH_BWDX(NXQ_, false, true); \
        else LAUNCH_BWDX(NXQ_, false, false);          \
    } while (0)
    if (nxq == 1) DISPATCH_BWDX(1);
    else if (nxq == 2) DISPATCH_BWDX(2);
    else if (nxq == 3) DISPATCH_BWDX(3);
    else DISPATCH_BWDX(4);
#undef DISPATCH_BWDX
#undef LAUNCH_BWDX
    return check_launch();
}

extern "C" int misplat_slab_reduce(const misplat_params* p, int64_t n_rows, int64_t n_isects,
                                   const int64_t* cum, const int32_t* tiles_per_gauss, const float* slab,
                                   const float* slab_abs, const uint8_t* slab_valid, float* v_grec,
                                   float* v_abs, misplat_stream_t stream) {
    if (!p || n_rows < 0 || n_isects < 0 || (slab_abs != nullptr) != (v_abs != nullptr)) return MISPLAT_EINVAL;
    if (n_rows == 0) return MISPLAT_OK;
    const dim3 grid(grid_for(n_rows * 16, 256)), block(256);
    hipLaunchKernelGGL(slab_reduce_kernel<kBands>, grid, block, 0, (hipStream_t)stream, n_rows, n_isects, cum, tiles_per_gauss,
                       slab, slab_abs, slab_valid, v_grec, v_abs);
    return check_launch();
}

extern "C" int misplat_depth_normal_fwd(int32_t width, int32_t height, float fx, float fy,
                                        const float* exp_depth, const float* med_depth,
                                        const float* n_render, float* normals2, float* err,
                                        misplat_stream_t stream) {
    if (width < 1 || height < 1 || !(fx > 0.f) || !(fy > 0.f)) return MISPLAT_EINVAL;
    DN d{width, height, fx, fy};
    hipLaunchKernelGGL(depth_normal_fwd_kernel, dim3(grid_for((int64_t)width * height, 256)), dim3(256), 0,
                       (hipStream_t)stream, d, exp_depth, med_depth, n_render, normals2, err);
    return check_launch();
}

extern "C" int misplat_depth_normal_bwd(int32_t width, int32_t height, float fx, float fy,
                                        const float* exp_depth, const float* med_depth,
                                        const float* n_render, const float* v_normals2,
                                        const float* v_err, float* v_exp_depth, float* v_med_depth,
                                        float* v_n_render, int32_t accumulate, misplat_stream_t stream) {
    if (width < 1 || height < 1 || !(fx > 0.f) || !(fy > 0.f)) return MISPLAT_EINVAL;
    DN d{width, height, fx, fy};
    hipLaunchKernelGGL(depth_normal_bwd_tiled_kernel, dim3((width + kDnTX - 1) / kDnTX, (height + kDnTY - 1) / kDnTY),
                       dim3(kDnTX * kDnTY), 0, (hipStream_t)stream, d, exp_depth, med_depth, n_render, v_normals2, v_err,
                       v_exp_depth, v_med_depth, v_n_render, (int)(accumulate != 0));
    return check_launch();
}

extern "C" const char* misplat_version(void) { return "misplat 0.1.0 gfx950"; }
