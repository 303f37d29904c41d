/* misplat.h -- C ABI of libmisplat.so: the MI355X (gfx950) RaDe-GS splat rasterizer.
 *
 * This is the drop-in boundary for the one hot path BASELINE.json's north_star names.
 * The reference (BasisResearch/collab-splats) has NO native code and no FFI of its own
 * (SURVEY.md section 2.1): the path leaves the repo through three Python entry points of the
 * third-party CUDA package gsplat-rade:
 *     gsplat.rendering.rasterization(...)             collab_splats/models/rade_gs_model.py:439-465
 *                                                     collab_splats/models/rade_features_model.py:450-476
 *     gsplat.cuda._wrapper.fully_fused_projection(...) collab_splats/models/rade_gs_model.py:373-394
 *     gsplat.cuda._wrapper.spherical_harmonics(...)    collab_splats/models/rade_features_model.py:430-434
 * plus the pure-PyTorch depth->normal stage    collab_splats/utils/camera_utils.py:176-279.
 * The functions below are what those Python entry points bind in this build (ctypes stub in
 * collab_splats_amd/_lib.py; the reference-side binding is shown in INTEGRATION.md).  Each entry
 * names the stage of the replaced call it implements.
 *
 * Rules (all entry points):
 *   - plain device pointers and sizes only; the caller (PyTorch) owns and allocates every
 *     buffer including scratch, so the caching allocator and stream order are respected;
 *   - asynchronous on `stream`; no hidden synchronisation, no allocation, no host read-back;
 *   - return 0 on success, a negative MISPLAT_E* code otherwise; never throws;
 *   - re-entrant: no mutable global state;
 *   - all floating point is fp32 (reference: mixed_precision=False,
 *     collab_splats/configs/rade_gs_method.py:31); indices int32; sort keys uint64.
 */
#ifndef MISPLAT_H
#define MISPLAT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* misplat_stream_t; /* == hipStream_t */

#define MISPLAT_OK 0
#define MISPLAT_EINVAL (-1)      /* bad size / unsupported option (tile_size != 16, colour dim) */
#define MISPLAT_ELAUNCH (-2)     /* hipGetLastError() after a launch was not hipSuccess */
#define MISPLAT_EWORKSPACE (-3)  /* scratch buffer too small */

#define MISPLAT_TILE 16          /* pixels per tile side (gsplat default tile_size)          */
#define MISPLAT_REC 16           /* floats per packed Gaussian record / per gradient row      */
#define MISPLAT_BANDS 2          /* wavefronts (bands of 16 x 8 pixels) that composite one tile */

/* Scalar configuration of one call (host memory, passed by pointer, copied at launch). */
typedef struct misplat_params {
    int32_t n_gauss;   /* N */
    int32_t n_cams;    /* C (the reference always uses 1: rade_gs_model.py:94-95, 446-447) */
    int32_t width, height;
    int32_t tile_size; /* must be MISPLAT_TILE */
    int32_t tile_w, tile_h;
    int32_t antialiased;          /* rasterize_mode == "antialiased" (rade_gs_model.py:459) */
    int32_t opacity_aware_radius; /* tighten the extent to the alpha_min level set */
    float eps2d;          /* 0.3   rade_gs_model.py:382 */
    float near_plane;     /* 0.01  rade_gs_model.py:451 */
    float far_plane;      /* 1e10  rade_gs_model.py:452 */
    float radius_clip;    /* 0.0   rade_gs_model.py:386 */
    float radius_sigma;   /* 3.33 */
    float alpha_max;      /* 0.999 */
    float alpha_min;      /* 1/255 */
    float t_stop;         /* 1e-4 */
    float median_t;       /* 0.5 */
    float jacobian_margin;/* 0.3 */
    float plane_eps;      /* 1e-6 */
    int32_t ed_slot;      /* colour channel (0..3) the compositing kernels divide by max(alpha,1e-10)
                             (the "ED" of render_mode RGB+ED / ED, rade_gs_model.py:237), or -1     */
    int32_t reserved_q;
    /* Launch order of the compositing kernels (speed only; results never depend on it).  A tile is covered by
     * MISPLAT_BANDS independent wavefronts ("bands" of 16 x 8 pixels, two pixels per lane); a "unit" is one band of
     * one tile: unit = tile * MISPLAT_BANDS + band.  unit_work (or NULL): the forward writes the number of staged
     * Gaussians each unit composited -- its measured cost; unit_perm (or NULL): workgroup b of a compositing launch
     * processes unit unit_perm[b] instead of the default XCD-strip map (misplat_unit_order builds it, longest first). */
    const int32_t* unit_perm;
    int32_t* unit_work;
    /* View-keyed orders (unit_sel != NULL): unit_perm is then the base of a TABLE of records of unit_stride int32 each --
     * MISPLAT_ORDER_HEADER header words {tag lo, tag hi, valid, ...} followed by a permutation -- and unit_sel points to
     * four device words {slot, valid, tag lo, tag hi} that misplat_raster_fwd's projection kernel writes from a hash
     * of the call's cameras (viewmats, Ks): a compositing launch uses the permutation of record `slot` if `valid`, the
     * default map otherwise; the forward's order kernel stores the new permutation (and the tag) into that record.  A
     * training loop revisits its cameras every epoch: each view finds the order its own last visit measured. */
#define MISPLAT_ORDER_HEADER 16
    /* (or NULL) one byte per (camera, Gaussian) row, a speed hint that never changes a result: cleared by
     * misplat_project_pack_fwd, set by the ATOMIC compositing backward for every row it adds a gradient to, read by the
     * per-Gaussian backward kernels -- a row that was never set has an all-zero gradient row, which then is not fetched
     * (in a dense scene nine rows in ten; their 64-byte rows were the largest read of both kernels). */
    uint8_t* touched;
    /* Extension (not in gsplat's interface): bit 0 -- `scales` are log-scales, bit 1 -- `opacities` are logits; the fused
     * projection kernels (misplat_project_pack_fwd / _bwd, one camera; misplat_raster_fwd / _bwd) apply exp / sigmoid
     * themselves and return the gradients of the raw parameters -- the four activation launches of
     * rade_gs_model.py:443-444 and their backward disappear.  0 = activated values, as gsplat takes them. */
    int32_t activations;
    int32_t unit_stride;     /* see unit_sel */
    const int32_t* unit_sel; /* or NULL */
    int32_t unit_slots;      /* records in the table: a selector outside [0, unit_slots) counts as "no record" */
    int32_t front_pass;      /* see front_n */
    /* Front-only ordering (set by misplat_raster_fwd for its compositing forward; NULL / 0 everywhere else -- the
     * backward never reads behind the entries the forward used).  unit_reach (or NULL) [units]: the forward writes the
     * DEPTH of the deepest list entry each unit looked at (front_depths[row]; +inf: pixels were still alive at the end of
     * its list, 0: nothing looked at) -- the pivot of the view's next visit.  front_n (or NULL) [C*tiles]: >= 0 -- only
     * that many entries at the head of the tile's list exist (the sorted front part, misplat_raster_args.front_n); a unit
     * that reaches their end with pixels alive sets tile_flag[tile].  front_pass 1: the launch composites the flagged
     * tiles only, over their whole (by then fully sorted) lists. */
    const int32_t* front_n;
    int32_t* tile_flag;
    float* unit_reach;
    const float* front_depths;
} misplat_params;

/* ---- a2.1 projection: fully_fused_projection(means, None, quats, scales, viewmats, Ks, W, H, ...)
 * (rade_gs_model.py:373-394).  Inputs: means[N,3] quats[N,4] (wxyz, un-normalised) scales[N,3]
 * opacities[N] or NULL, viewmats[C,4,4] row-major world->camera, Ks[C,3,3].
 * Outputs, all [C,N,...]: radii int32[.,2], means2d[.,2], depths[.], conics[.,3],
 * compensations[.], ray_ts[.], ray_planes[.,2], normals[.,3]; rows with radii==0 are zero. */
int misplat_project_fwd(const misplat_params* p, const float* means, const float* quats,
                        const float* scales, const float* opacities, const float* viewmats,
                        const float* Ks, int32_t* radii, float* means2d, float* depths,
                        float* conics, float* compensations, float* ray_ts, float* ray_planes,
                        float* normals, misplat_stream_t stream);

/* Backward of the above.  v_* inputs are [C,N,...] gradients of the eight outputs (radii has
 * none); outputs v_means[N,3], v_quats[N,4], v_scales[N,3] are summed over cameras and fully
 * overwritten. */
int misplat_project_bwd(const misplat_params* p, const float* means, const float* quats,
                        const float* scales, const float* viewmats, const float* Ks,
                        const int32_t* radii, const float* v_means2d, const float* v_depths,
                        const float* v_conics, const float* v_compensations,
                        const float* v_ray_ts, const float* v_ray_planes, const float* v_normals,
                        float* v_means, float* v_quats, float* v_scales, misplat_stream_t stream);

/* ---- a2.2 SH: spherical_harmonics(degrees_to_use, dirs, coeffs) (rade_features_model.py:430-434).
 * dirs[C*N,3] (normalised inside), coeffs[N,K,3] shared by the C cameras, radii[C*N,2] or NULL
 * (rows with radii==0 are skipped and get colour 0), colors[C*N,3] = raw SH (no +0.5). */
int misplat_sh_fwd(int32_t n_gauss, int32_t n_cams, int32_t K, int32_t degree, const float* dirs,
                   const float* coeffs, const int32_t* radii, float* colors,
                   misplat_stream_t stream);
int misplat_sh_bwd(int32_t n_gauss, int32_t n_cams, int32_t K, int32_t degree, const float* dirs,
                   const float* coeffs, const int32_t* radii, const float* v_colors,
                   float* v_coeffs /*[N,K,3]*/, float* v_dirs /*[C*N,3]*/,
                   misplat_stream_t stream);

/* ---- fused per-Gaussian stages (what rasterization() runs; same math as the entry points above,
 * without the SoA round trips).  project_pack_fwd writes the geometry part of the packed blend
 * record (layout: see misplat_pack) straight away; color_fwd fills the record's colour slots:
 *   sh_degree >= 0: colour = max(SH_deg(mean - camera_centre) . coeffs[N,K,3] + 0.5, 0), K <= 16
 *                   (rade_features_model.py:428-438; coefficients staged through LDS, coalesced);
 *   sh_degree <  0: the first n_color (<= 4) of the D colour channels are copied
 *                   (colors [N,D], or [C,N,D] when per_cam).
 * depth_channel != 0 puts the z-depth into colour slot n_color (RGB+ED / ED render modes).
 * The backward kernels consume the packed gradient rows v_grec[C*N,16] produced by
 * misplat_slab_reduce; v_means2d is passed separately (it is a retained autograd node). */
int misplat_project_pack_fwd(const misplat_params* p, const float* means, const float* quats,
                             const float* scales, const float* opacities, const float* viewmats,
                             const float* Ks, int32_t* radii, float* means2d, float* depths,
                             float* compensations, float* grec,
                             uint32_t* zero_words /* or NULL */, int32_t n_zero /* words the kernel clears for its
                             successors on the stream (the bucketing counters): saves a memset launch */,
                             float* lazy_rows /* or NULL.  Given (v_grec-shaped, [C*N,16]): on-demand colours -- the colour
                             slots of grec are left UNSET for misplat_blend_fwd_lazy, and these gradient rows are cleared */,
                             float* abs_rows /* or NULL: v_abs-shaped rows [C*N,2] cleared for the compositing backward
                             (misplat_blend_bwd_atomic, v_grec_is_zero bit 1): saves that memset launch */,
                             misplat_stream_t stream);
/* coeffs_rest (SH only, may be NULL): when given, coeffs_or_colors is features_dc[N,3] and
 * coeffs_rest is features_rest[N,K-1,3] -- the reference's two parameter tensors
 * (rade_gs_model.py:119-120) read in place instead of through its per-step torch.cat (:128-130);
 * v_coeffs_rest is the matching gradient output.
 * sh_aux[C*N,12] (SH only, may be NULL): written by color_fwd -- the 3x3 Jacobian d rgb / d dir of the clamped
 * colour (rows of clamped channels zeroed) and the three clamp flags -- and, when passed to color_bwd, used
 * instead of re-reading the 12 K bytes of coefficients per Gaussian (48 B instead of 192 B at degree 3). */
int misplat_color_fwd(const misplat_params* p, int32_t sh_degree, int32_t K_or_D, int32_t n_color,
                      int32_t per_cam, int32_t depth_channel, const float* means,
                      const float* viewmats, const float* coeffs_or_colors, const float* coeffs_rest,
                      const int32_t* radii, const float* depths, float* grec,
                      float* sh_aux /* or NULL */,
                      float* zero_rows /* or NULL: [C*N,16] cleared by the kernel -- the gradient rows the backward's
                      atomics add into (misplat_blend_bwd_atomic, v_grec_is_zero): saves the 64 B/row memset launch */,
                      misplat_stream_t stream);
int misplat_color_bwd(const misplat_params* p, int32_t sh_degree, int32_t K_or_D, int32_t n_color,
                      int32_t per_cam, const float* means, const float* viewmats,
                      const float* coeffs_or_colors, const float* coeffs_rest, const int32_t* radii,
                      const float* v_grec, float* v_coeffs_or_colors, float* v_coeffs_rest,
                      float* v_means_dir /*[N,3], SH only*/, const float* sh_aux /* or NULL */,
                      misplat_stream_t stream);
/* depth_slot: 12..15 = record slot carrying the depth channel, -1 = none.  v_means_dir may be
 * NULL.  v_depth_rows (or NULL; then depth_slot must be -1): the depth channel's gradient lives OUTSIDE the record -- row
 * g's value is v_depth_rows[g * v_depth_stride] (the N-D colour path keeps channels >= 4 in featx: pass a pointer into
 * v_featx).  Outputs v_means[N,3] v_quats[N,4] v_scales[N,3] v_opacities[N], summed over cameras. */
int misplat_project_pack_bwd(const misplat_params* p, int32_t depth_slot, const float* means,
                             const float* quats, const float* scales, const float* opacities,
                             const float* viewmats, const float* Ks, const int32_t* radii,
                             const float* compensations, const float* v_means2d, const float* v_grec,
                             const float* v_means_dir, float* v_means, float* v_quats,
                             float* v_scales, float* v_opacities, const float* v_depth_rows,
                             int32_t v_depth_stride, misplat_stream_t stream);

/* ---- a2.3 binning: for every tile the Gaussian rows whose rectangle mean2d +- radii touches it, in
 * (depth, row) order -- the order of gsplat's 64-bit (tile | depth) key sort, without the keys: the cell-ordered
 * bucketing (bucket_*, below) fills every tile's bucket in arbitrary order, then misplat_tile_sort orders every bucket by
 * (depth, row).  offsets arrays have C*tiles + 1 entries: the last one is the number of intersections. */
/* gsplat's meta["isect_ids"] on demand: (tile << 32) | bits(depth[flatten_ids[i]]). */
int misplat_isect_ids(const uint32_t* tiles_sorted, const int32_t* flatten_ids, const float* depths,
                      int64_t n_isects, uint64_t* isect_ids, misplat_stream_t stream);

/* Per-tile ordering: once the intersections have been bucketed by tile, one workgroup per tile sorts its bucket --
 * entries in registers, exchange through LDS: one counting pass on the bucket's own depth range + ranking inside the
 * bins, or stable LSD radix passes -- into the (depth, row) order, equal depths included: exactly the (tile, depth, id)
 * order of a global 64-bit key sort, without one.  Buckets arrive in arbitrary order (misplat_bucket_tiles).
 * offsets: n_tiles_total + 1 entries; n_isects: the number of intersections or an upper bound of it (only used to
 * size the grids of the size classes); payload (in/out): rows, or emission slots when isect_gid != NULL
 * (row = isect_gid[slot]); flatten_ids (out): rows in final order; scratch[2 * n_isects] backs the rare tiles
 * longer than 8192 entries. */
int misplat_tile_sort(const int32_t* offsets, int32_t n_tiles_total, int64_t n_isects,
                      const float* depths, const int32_t* isect_gid, int32_t* payload,
                      int32_t* flatten_ids, uint32_t* scratch, int32_t flags /* bit 1:
                      offsets[n_tiles_total + 1] holds the longest bucket (misplat_bucket_tiles writes it), so the
                      launches of unused size classes return at once; bit 2: the payload holds positions in the cell-ordered
                      row list (misplat_raster_args.depth_sorted): isect_gid = that list (row = isect_gid[entry], equal depths
                      come out in ROW order), depths is indexed by the entry */, misplat_stream_t stream);

/* ---- Cell-ordered bucketing (csrc/bucket.hip).  Replaces gsplat's isect_tiles +
 * radix sort + isect_offset_encode: every intersection is written once (its row) and no tile-id array exists.
 *   bucket_plan   (host only) number of screen cells and of counting workgroups for this configuration:
 *                 cellhist holds n_blocks * n_cells uint32, cell_count n_cells, cell_offs n_cells + 1;
 *   bucket_count  tiles_per_gauss[C*N], rect2[C*N] (x0 | y0 << 16, w | h << 16 of the tile rectangle), the
 *                 per-workgroup cell histograms, the global cell counts; counters[0] = number of intersections
 *                 (device int64; cell_count and counters[2] are zeroed here on `stream` unless already_zero);
 *   bucket_rows   cell_offs = scan of the cell counts, counters[1] = visible rows, order[0 .. n_vis) = the visible
 *                 rows in cell order and rect_sorted[0 .. n_vis) their rectangles (layout of rect2, capacity C*N);
 *                 cell_cursor[n_cells] (scratch) and tile_count[n_tiles + 1] must be zero on entry: cleared here unless
 *                 already_zero says an earlier kernel of the stream did (bit 0: cell_cursor, bit 1: tile_count);
 *                 n_isects_host (or NULL): 8 bytes of PINNED, device-mapped host memory that receives counters[0] as a
 *                 system-scope store from the kernel -- the caller polls it (misplat_wait_count), no copy is enqueued;
 *   bucket_tiles  (tile_count must be all zero on entry, as bucket_rows leaves it; it is NOT zero afterwards)
 *                 offsets[0 .. n_tiles + 1] (offsets[n_tiles] = number of intersections, offsets[n_tiles + 1] = the
 *                 longest bucket: n_tiles + 2 entries) and
 *                 payload[offsets[t] .. offsets[t + 1]) = the rows touching tile t, in arbitrary order
 *                 (misplat_tile_sort follows).  cum != NULL (deterministic backward): the payload
 *                 is the emission slot cum[row] + k and isect_gid[slot] = row.  Writes beyond cap_isects entries
 *                 are dropped: the caller compares counters[0] with cap_isects afterwards.
 * Nothing here needs a host read-back: all sizes live in `counters` on the device. */
#define MISPLAT_BUCKET_MAX_CELLS 2048
#define MISPLAT_BUCKET_MAX_BLOCKS 256
int misplat_bucket_plan(const misplat_params* p, int32_t* n_cells, int32_t* n_blocks);
int misplat_bucket_count(const misplat_params* p, const float* means2d, const int32_t* radii,
                         int32_t* tiles_per_gauss, uint32_t* rect2, uint32_t* cellhist, uint32_t* cell_count,
                         int64_t* counters, int32_t already_zero /* cell_count and counters were cleared by an
                         earlier kernel of the stream (misplat_project_pack_fwd's zero_words): no memset here */,
                         misplat_stream_t stream);
int misplat_bucket_rows(const misplat_params* p, const int32_t* tiles_per_gauss, const uint32_t* rect2,
                        const uint32_t* cellhist, const uint32_t* cell_count, uint32_t* cell_cursor, uint32_t* cell_offs,
                        int32_t* order, uint32_t* rect_sorted, int64_t* counters, int32_t* tile_count,
                        int64_t* n_isects_host /* or NULL */, int32_t already_zero, misplat_stream_t stream);
int misplat_bucket_tiles(const misplat_params* p, const int32_t* order, const uint32_t* rect_sorted,
                         const int64_t* counters, int32_t* tile_count, int32_t* offsets, const int64_t* cum,
                         int64_t cap_isects, int32_t* payload, int32_t* isect_gid, misplat_stream_t stream);

/* ---- a2.4 / a2.5 compositing.  Packed record per (camera, Gaussian), MISPLAT_REC floats:
 *   [0:2] mean2d  [2:5] conic  [5] opacity_eff  [6] ray_t  [7:9] ray_plane  [9:12] normal
 *   [12:16] colour channels 0..3 (unused channels zero).                         */
int misplat_pack(int64_t n_rows, int32_t color_dim, const float* means2d, const float* conics,
                 const float* opacities_eff, const float* ray_ts, const float* ray_planes,
                 const float* normals, const float* colors, float* grec, misplat_stream_t stream);

/* Forward: one wavefront per band of 16 x 8 pixels of a tile (two pixels per lane).
 * offsets: C*tiles + 1 entries (tile t owns flatten_ids[offsets[t] .. offsets[t+1])); n_isects: the number of
 * intersections or an upper bound (the backward's slab stride).
 * Outputs [C,H,W,...]: render[.,color_dim], alpha[.], exp_depth[.] (sum w*z, un-normalised),
 * med_depth[.], normal[.,3], last_ids[.], median_ids[.] (sorted positions; -1 = none).
 * color_dim in 1..4. */
int misplat_blend_fwd(const misplat_params* p, int32_t color_dim, const float* Ks,
                      const float* grec, const int32_t* flatten_ids, const int32_t* offsets,
                      int64_t n_isects, float* render, float* alpha, float* exp_depth,
                      float* med_depth, float* normal, int32_t* last_ids, int32_t* median_ids,
                      misplat_stream_t stream);
/* The same forward with ON-DEMAND SH colours: the colour slots of grec hold the "unset" pattern left by
 * misplat_project_pack_fwd(lazy_rows != NULL) and are filled here (grec is read AND written) for the records that are
 * staged past their cull -- in a dense scene most visible Gaussians never are.  16 coefficients per Gaussian: coeffs
 * [N,16,3], or features_dc [N,3] + coeffs_rest [N,15,3]. */
int misplat_blend_fwd_lazy(const misplat_params* p, int32_t color_dim, const float* Ks, float* grec,
                           const int32_t* flatten_ids, const int32_t* offsets, int64_t n_isects, float* render,
                           float* alpha, float* exp_depth, float* med_depth, float* normal, int32_t* last_ids,
                           int32_t* median_ids, const float* means, const float* viewmats, const float* coeffs,
                           const float* coeffs_rest, int32_t sh_degree, int32_t depth_channel, const float* depths,
                           float* sh_aux /* NULL */, misplat_stream_t stream);


/* Launch order for the compositing kernels (speed only): unit_perm[8 * ceil(units / 8)] from the per-unit cost
 * unit_work[units] the forward measured (misplat_params.unit_work), longest first inside every XCD strip;
 * units = C * tiles * MISPLAT_BANDS.  Padding entries hold `units` (no unit).  Pass the result as
 * misplat_params.unit_perm to any later compositing launch -- the backward of the same step, or the next forward of the
 * same view. */
int misplat_unit_order(const misplat_params* p, const int32_t* unit_work, int32_t* unit_perm,
                       misplat_stream_t stream);

/* Number of gradient planes (= bands per tile) the backward of this configuration writes. */
int misplat_blend_planes(const misplat_params* p);

/* Backward: every band re-traverses its tile list back to front and writes ONE gradient row
 * (record layout) per Gaussian that contributed to the band, to slab[band][slot] (slot = position in
 * emission order, so the rows of one Gaussian are contiguous), and sets slab_valid[band][slot] = 1.
 * slab: [planes, n_isects, 16] floats; slab_abs: [planes, n_isects, 2] or NULL (sum |dL/dmean2d|);
 * slab_valid: [planes, n_isects] bytes, zeroed here on `stream`.  No float atomics anywhere. */
int misplat_blend_bwd(const misplat_params* p, int32_t color_dim, const float* Ks,
                      const float* grec, const int32_t* flatten_ids, const int32_t* slots_sorted,
                      const int32_t* offsets, int64_t n_isects, const float* alpha,
                      const int32_t* last_ids, const int32_t* median_ids,
                      const float* render /* forward output, read only if ed_slot >= 0 */,
                      const float* v_render, const float* v_alpha, const float* v_exp_depth,
                      const float* v_med_depth, const float* v_normal, float* slab, float* slab_abs,
                      uint8_t* slab_valid,
                      misplat_stream_t stream);
/* Same backward with the rows added straight into v_grec[C*N,16] / v_abs[C*N,2] (or NULL) by
 * no-return fp32 atomics of 64 contiguous bytes: no slab and no reduce pass, but the sums depend on
 * arrival order.  v_grec / v_abs are zeroed here on `stream`. */
int misplat_blend_bwd_atomic(const misplat_params* p, int32_t color_dim, const float* Ks,
                             const float* grec, const int32_t* flatten_ids, const int32_t* offsets,
                             int64_t n_isects, const float* alpha, const int32_t* last_ids,
                             const int32_t* median_ids, const float* render, const float* v_render,
                             const float* v_alpha, const float* v_exp_depth, const float* v_med_depth,
                             const float* v_normal, float* v_grec, float* v_abs,
                             int32_t v_grec_is_zero /* bit 0: v_grec was cleared by misplat_color_fwd (zero_rows) and has
                             not been used since; bit 1: v_abs was cleared by the caller: skip that memset */,
                             misplat_stream_t stream);
/* ---- a8: N-D colours in one pass (rade_features_model.py:441-476: D = 16 fused channels, 17 with
 * RGB+ED).  n_channels = D' in 5..20; channels 0..3 live in the record's colour slots, channels 4..
 * in featx[C*N, 4*nxq] (nxq = ceil((D'-4)/4) float4s per row, zero padded); color_fwd_x writes both
 * from colors[(C,)N,D] (+ the depth channel); render is [C,H,W,n_channels].  Gradients: atomic mode
 * only (v_grec, v_featx, v_abs zeroed here on `stream`). */
int misplat_color_fwd_x(const misplat_params* p, int32_t D, int32_t per_cam, int32_t depth_channel,
                        int32_t nxq, const float* colors, const int32_t* radii, const float* depths,
                        float* grec, float* featx, misplat_stream_t stream);
int misplat_color_bwd_x(const misplat_params* p, int32_t D, int32_t per_cam, int32_t nxq,
                        const int32_t* radii, const float* v_grec, const float* v_featx,
                        float* v_colors, misplat_stream_t stream);
int misplat_blend_fwd_x(const misplat_params* p, int32_t n_channels, int32_t nxq, const float* Ks,
                        const float* grec, const float* featx, const int32_t* flatten_ids,
                        const int32_t* offsets, int64_t n_isects, float* render, float* alpha,
                        float* exp_depth, float* med_depth, float* normal, int32_t* last_ids,
                        int32_t* median_ids, misplat_stream_t stream);
int misplat_blend_bwd_x_atomic(const misplat_params* p, int32_t n_channels, int32_t nxq, const float* Ks,
                               const float* grec, const float* featx, const int32_t* flatten_ids,
                               const int32_t* offsets, int64_t n_isects, const float* alpha,
                               const int32_t* last_ids, const int32_t* median_ids, const float* render,
                               const float* v_render, const float* v_alpha, const float* v_exp_depth,
                               const float* v_med_depth, const float* v_normal, float* v_grec,
                               float* v_featx, float* v_abs, misplat_stream_t stream);

/* v_grec[r] = sum of the VALID slab rows of Gaussian row r (slots cum[r] .. cum[r]+tiles_per_gauss[r],
 * all planes) in a fixed order => bitwise reproducible.  v_abs[n_rows,2] likewise from slab_abs
 * (both may be NULL together). */
int misplat_slab_reduce(const misplat_params* p, int64_t n_rows, int64_t n_isects, const int64_t* cum,
                        const int32_t* tiles_per_gauss, const float* slab, const float* slab_abs,
                        const uint8_t* slab_valid, float* v_grec, float* v_abs,
                        misplat_stream_t stream);

/* ---- a4 depth->normal (camera_utils.py:176-279) fused with the error map of
 * rade_gs_model.py:212-214: n_d = normalize(cross(dP/drow, dP/dcol)) of the back-projected
 * expected / median z-depth maps (interior pixels; border 0), err[k] = 1 - <n_render, n_d[k]>.
 * depths: exp_depth[H,W], med_depth[H,W]; n_render[H,W,3]; outputs normals2[2,H,W,3], err[2,H,W]. 
 * accumulate != 0 (backward): the three gradients are ADDED to the output buffers (which then already hold the
 * gradients of the a3 epilogue, misplat_outputs_bwd) instead of stored. */
int misplat_depth_normal_fwd(int32_t width, int32_t height, float fx, float fy,
                             const float* exp_depth, const float* med_depth,
                             const float* n_render, float* normals2, float* err,
                             misplat_stream_t stream);
int misplat_depth_normal_bwd(int32_t width, int32_t height, float fx, float fy,
                             const float* exp_depth, const float* med_depth,
                             const float* n_render, const float* v_normals2 /*or NULL*/,
                             const float* v_err /*or NULL*/, float* v_exp_depth,
                             float* v_med_depth, float* v_n_render, int32_t accumulate, misplat_stream_t stream);

/* ---- a3 get_outputs post-processing (rade_gs_model.py:221-254), one pixel = one image element
 * (C = 1).  background3_host: 3 floats in HOST memory.  maxes4: 4-float device scratch receiving
 * the un-masked maxima of expected depth, median depth, (normal+1)/2 and render[...,3].
 *   rgb = clamp(render[:3] + (1-alpha)*bg, 0, 1); depth/median_depth/normals/depth_im =
 *   where(alpha > 0, x, max(x)) with normals = (expected_normals+1)/2; depth_im (or NULL) needs
 *   color_dim == 4.  The maxima carry no gradient (the reference detaches them). */
int misplat_outputs_fwd(int64_t n_pix, int32_t color_dim, const float* background3_host,
                        const float* render, const float* alpha, const float* exp_depth,
                        const float* med_depth, const float* exp_normal, float* maxes4, float* rgb,
                        float* depth, float* median_depth, float* normals, float* depth_im,
                        misplat_stream_t stream);
int misplat_outputs_bwd(int64_t n_pix, int32_t color_dim, const float* background3_host,
                        const float* render, const float* alpha, const float* v_rgb,
                        const float* v_depth, const float* v_median_depth, const float* v_normals,
                        const float* v_depth_im /* or NULL */, float* v_render, float* v_alpha,
                        float* v_exp_depth, float* v_med_depth, float* v_exp_normal,
                        misplat_stream_t stream);

/* ---- a5 get_loss_dict (rade_gs_model.py:289-307: depth_normal_lambda * ((1 - depth_ratio) * mean(depth_normal_error_map)
 * + depth_ratio * mean(middepth_normal_error_map)); the base model's L1 term mean |gt - rgb|) in two launches, and
 * its backward in one.  rgb / gt [n_pix,3] (16-byte aligned), err_exp / err_med [n_pix]; either pair may be NULL (that
 * loss is then not computed).  partials: 3 * 512 floats of device scratch.  rgb_loss / dn_loss: device scalars.
 * Reproducible bit for bit (fixed grid and summation tree, fp64 final sums). */
#define MISPLAT_LOSS_PARTIALS (3 * 512)
int misplat_loss_fwd(int64_t n_pix, const float* rgb, const float* gt, const float* err_exp, const float* err_med,
                     float depth_ratio, float depth_normal_lambda, float* partials, float* rgb_loss, float* dn_loss,
                     misplat_stream_t stream);
/* g_rgb_loss / g_dn_loss: the upstream gradients of the two loss values as DEVICE scalars (or NULL = 0);
 * v_rgb[n_pix,3] = -g / (3 n) sign(gt - rgb), v_err_exp / v_err_med[n_pix] = the constant images g lambda (1 - r) / n and
 * g lambda r / n.  v_rgb or the v_err pair may be NULL. */
int misplat_loss_bwd(int64_t n_pix, const float* rgb, const float* gt, const float* g_rgb_loss, const float* g_dn_loss,
                     float depth_ratio, float depth_normal_lambda, float* v_rgb, float* v_err_exp, float* v_err_med,
                     misplat_stream_t stream);

/* ---- the image term of the loss the model inherits (rade_gs_model.py:289 `super().get_loss_dict`: nerfstudio Splatfacto,
 * third-party and absent from the reference tree) [UNVERIFIED-UPSTREAM]:
 *   main_loss = (1 - ssim_lambda) * mean |gt - rgb| + ssim_lambda * (1 - SSIM(gt, rgb)),
 * SSIM as pytorch_msssim.SSIM(data_range=1.0, size_average=True, channel=3): 11-tap gaussian window (sigma 1.5), valid
 * region (H - 10) x (W - 10), K = (0.01, 0.03), mean over positions and channels.  rgb / gt: [H,W,3]; H, W >= 11.
 * scratch: misplat_ssim_scratch_floats(H, W) floats, written by the forward (three derivative maps + tile sums) and read
 * by the backward.  l1_loss: DEVICE scalar mean |gt - rgb| computed elsewhere (misplat_loss_fwd's rgb_loss), or NULL =
 * the forward sums it itself (its tiles hold both images anyway: fixed tile sums, fp64 final sum); ssim / main_loss:
 * device scalars (either may be NULL).  Two launches; reproducible bit for bit. */
int64_t misplat_ssim_scratch_floats(int32_t height, int32_t width);
int misplat_ssim_fwd(int32_t height, int32_t width, const float* rgb, const float* gt, float* scratch,
                     const float* l1_loss /* or NULL */, float ssim_lambda, float* ssim /* or NULL */,
                     float* main_loss /* or NULL */, misplat_stream_t stream);
/* v_rgb[H,W,3] = g_main * d main_loss / d rgb (both terms: the L1 sign gradient and the transposed window filter of the
 * forward's maps); g_main: DEVICE scalar (or NULL = 0).  One launch, no atomics. */
int misplat_ssim_bwd(int32_t height, int32_t width, const float* rgb, const float* gt, const float* scratch,
                     const float* g_main, float ssim_lambda, float* v_rgb, misplat_stream_t stream);

/* ---- the whole forward of rasterization() (rade_gs_model.py:439-465) as ONE host entry: csrc/raster.hip.
 * Every pointer is a caller-allocated device buffer of the size the per-stage entry points above document
 * (n_isects_host: 8 bytes of PINNED host memory).
 *   phases & 1 (A): project_pack_fwd, bucket_count, bucket_rows (which stores counters[0] into *n_isects_host), color_fwd
 *   phases & 2 (B): bucket_tiles, tile_sort; blend_fwd (+ unit_work), unit_order
 * B only reads device-side sizes, so it may be enqueued with a speculative cap_isects before the host knows the count:
 * the result is exact iff the count (misplat_wait_count) is <= cap_isects (else: clear tile_count and call B again
 * with the exact size).  Atomic gradient mode only (the deterministic slab needs the emission-slot scan between A and B). */
typedef struct misplat_raster_args {
    /* inputs (rasterization() arguments) */
    const float *means, *quats, *scales, *opacities, *colors, *colors_rest, *viewmats, *Ks;
    int32_t sh_degree;      /* -1: pass-through colours */
    int32_t K_or_D, n_color, per_cam, depth_channel;
    int32_t color_dim;      /* channels the compositing kernels carry (1..4) */
    int32_t reserved_c;
    int32_t lazy_colour;    /* != 0 (SH colours with K = 16, one pass): no colour kernel -- the compositing
                               forward evaluates the colour of a record when it first stages it (misplat_blend_fwd_lazy);
                               sh_aux is not written.  2: in addition v_grec_zero is NOT cleared as a whole -- only the
                               rows of the records whose colour gets set, which are the only rows the compositing backward
                               adds into: for a backward that reads flagged rows only (misplat_raster_bwd_plan bit 0);
                               any other backward must clear v_grec first (zero_flags bit 0 unset) */
    /* per (camera, Gaussian) outputs */
    int32_t* radii;
    float *means2d, *depths, *compensations, *grec, *sh_aux /* or NULL */;
    float* v_grec_zero; /* or NULL: gradient rows [C*N,16] cleared by the colour kernel (see misplat_color_fwd) */
    /* bucketing workspace + results */
    int32_t* tiles_per_gauss;
    uint32_t *rect2, *cellhist, *cell_count, *cell_offs;
    uint32_t* cell_cursor;  /* n_cells words of scratch; cell_count, cell_cursor, counters (4 words) and tile_count must be
                               slices of ONE allocation in this order: the projection kernel clears the whole range */
    int32_t* order;
    uint32_t* rect_sorted;
    int64_t* counters;
    int32_t *tile_count, *offsets, *payload, *flatten_ids;
    uint32_t* scratch;
    int64_t cap_isects;
    int64_t* n_isects_host; /* pinned host memory, or NULL */
    float* v_abs_zero; /* or NULL: |mean2d gradient| rows [C*N,2] cleared by the projection kernel */
    /* images */
    float *render, *alpha, *exp_depth, *med_depth, *normal;
    int32_t *last_ids, *median_ids;
    /* launch order (speed only): see misplat_params.unit_perm / unit_work */
    const int32_t* unit_perm_in;
    int32_t *unit_work, *unit_perm_out;
    /* measurement (or NULL): two hipEvent_t recorded on `stream` directly before and after the compositing forward; a
     * call that carries them is launched plainly (no graph) */
    void *ev_blend_begin, *ev_blend_end;
    /* view-keyed launch orders (misplat_params.unit_sel), or NULL: order_table[order_slots][order_stride] int32 --
     * zero-initialised by the caller once and kept between calls --, order_sel[4].  Phase A selects the record of the
     * call's cameras, phase B composites in that record's order (if it holds one) and stores the order it measured.
     * unit_perm_in / unit_perm_out are ignored then; unit_work is still needed. */
    int32_t *order_table, *order_sel;
    int32_t order_slots, order_stride;
    /* Front-only ordering for dense scenes (needs the view-keyed table; csrc/binning.hip).  unit_reach (or NULL) [units]:
     * the compositing forward's reach per unit; the order kernel turns it into per-tile depth pivots stored behind the
     * permutation of the view's record (order_stride >= MISPLAT_ORDER_HEADER + 8 * ceil(units / 8) + C * tiles).
     * front_n / tile_flag (both or neither) [C*tiles]: with them, phase B sorts only the part of every bucket at or in
     * front of front_margin x the pivot of the view's last visit (buckets of >= front_min_bucket entries; everything else,
     * and every tile of a view without a record, is sorted in full), composites, sorts the tiles whose pixels were still
     * alive at the end of their truncated list in full and composites those again: exact images either way.  flatten_ids
     * is then complete only for the tiles with front_n < 0 or tile_flag != 0; misplat_tile_sort over `payload` (which stays
     * a permutation of every bucket) completes it. */
    float* unit_reach;
    int32_t *front_n, *tile_flag;
    float front_margin;
    int32_t front_min_bucket;
    /* (or NULL) [C*N] floats: the bucket entries (payload) are then each row's POSITION in `order` (the cell-ordered row
     * list) instead of the row, and depth_sorted[position] = depths[row] -- the per-tile sort gathers its keys from a few
     * hundred KB around the tile instead of from all over depths[] (at 5 M Gaussians that gather was most of the sort's time).
     * flatten_ids holds rows either way.  Sorting such a payload by hand: misplat_tile_sort(flags | 4, depths = depth_sorted,
     * isect_gid = order).  Required by front_n. */
    float* depth_sorted;
    /* N-D colours in the one-entry path (a8: rade_features_model.py:427-476), nxq > 0: D' = color_dim composited channels
     * in 5..20, channels 0..3 in the record's colour slots, channels 4.. in featx[C*N, 4*nxq] (nxq = ceil((D' - 4) / 4)).
     *   sh_degree < 0:  colors [(C,)N,D] (D = K_or_D) are the channels (+ depth behind them if depth_channel);
     *   sh_degree >= 0: the model's call as ONE entry -- channels 0..2 = max(SH(colors | colors, colors_rest) + 0.5, 0),
     *                   channels 3.. = features[N, n_feat] (+ depth): no [N,16] concatenation exists anywhere.
     * v_featx_zero (or NULL): gradient rows [C*N, 4*nxq] cleared by the forward, like v_grec_zero.  lazy_colour must be 0. */
    const float* features;
    float *featx, *v_featx_zero;
    int32_t n_feat, nxq;
    int64_t est_isects; /* > 0: the caller's estimate of the intersection count (cap_isects may be a generous, long-lived
                           capacity): picks the per-tile sort's size classes; 0: 4/5 of cap_isects */
} misplat_raster_args;
/* Graph cache (optional, caller-owned, thread-safe; the library itself keeps no state): with a cache, the launch
 * sequence of a call is captured into a hipGraph the first time a given (params, args, phases, stream) block is seen
 * and replayed with one hipGraphLaunch afterwards -- small scenes are launch-bound (~16 launches per forward).  The
 * least recently used of max_entries graphs is evicted.  A cache must not outlive the device context. */
typedef struct misplat_graph_cache misplat_graph_cache;
misplat_graph_cache* misplat_graph_cache_create(int32_t max_entries);
void misplat_graph_cache_destroy(misplat_graph_cache* cache);
int misplat_graph_cache_stats(misplat_graph_cache* cache, int64_t* hits, int64_t* captures);
int misplat_raster_fwd(const misplat_params* p, const misplat_raster_args* a, int32_t phases, misplat_stream_t stream,
                       misplat_graph_cache* cache /* or NULL: plain launches */);
/* The whole backward of rasterization() as ONE host entry: misplat_blend_bwd_atomic, misplat_color_bwd,
 * misplat_project_pack_bwd back to back (arguments as documented there).  zero_flags: see misplat_blend_bwd_atomic;
 * with a cache the sequence is replayed as a graph when it contains no memset (zero_flags covers v_grec and v_abs).
 * With misplat_params.touched, one camera, 16 SH coefficients without sh_aux and >= 262 144 Gaussians the same results
 * come from two launches: the compositing backward, whose grid also clears the seven output tensors, and ONE kernel for
 * both per-Gaussian stages that visits only the flagged rows (v_means_dir is then not written). */
typedef struct misplat_raster_bwd_args {
    /* compositing backward: saved by the forward */
    const float *Ks, *grec;
    const int32_t *flatten_ids, *offsets;
    int64_t n_isects;   /* the number of list entries, or any upper bound inside flatten_ids' buffer (the forward's capacity):
                           every range comes from `offsets`, this only clamps them -- a caller that replays graphs passes the
                           capacity, because the exact count of a scene in training changes with every step */
    const float* alpha;
    const int32_t *last_ids, *median_ids;
    const float* render;
    /* upstream gradients of the five images (all given) */
    const float *v_render, *v_alpha, *v_exp_depth, *v_med_depth, *v_normal;
    float *v_grec, *v_abs /* or NULL */;   /* v_grec is WORKSPACE of this call: in the two-launch form without v_abs its rows hold sums
                                            * the per-Gaussian kernel finishes (slots 0 - 1, 5: csrc/blend.hip MSUM), not the layout of
                                            * misplat_blend_bwd_atomic -- take the mean2d gradient from v_means2d_out */
    const int32_t* unit_perm /* or NULL */;
    int32_t color_dim, zero_flags;
    /* per-Gaussian backward */
    int32_t sh_degree, K_or_D, n_color, per_cam, depth_slot, reserved;
    const float *means, *quats, *scales, *opacities, *colors, *colors_rest, *viewmats;
    const int32_t* radii;
    const float *compensations, *sh_aux /* or NULL */;
    const float* v_means2d /* or NULL: the mean2d gradient is columns 0:2 of v_grec */;
    float *v_colors, *v_colors_rest /* or NULL */, *v_means_dir /* or NULL */, *v_means, *v_quats, *v_scales, *v_opacities;
    /* measurement (or NULL): two hipEvent_t recorded on `stream` directly before and after the compositing backward; a
     * call that carries them is launched plainly (no graph) */
    void *ev_blend_begin, *ev_blend_end;
    /* or NULL: [N,2], the mean2d gradient as a tensor of its own (zeros where no gradient arrived) -- written by the
     * two-launch form only (misplat_raster_bwd_plan bit 0), where v_grec may hold defined values in flagged rows only
     * (misplat_raster_args.lazy_colour = 2) and cannot be sliced for it */
    float* v_means2d_out;
    /* view-keyed launch order: unit_perm is the table base, see misplat_params.unit_sel (or NULL / 0) */
    const int32_t* unit_sel;
    int32_t unit_stride, unit_slots;
    /* N-D colours (nxq > 0; see misplat_raster_args.featx): featx as the forward wrote it, v_featx [C*N, 4*nxq] (zero_flags
     * bit 2: cleared by the forward), depth_channel != 0 if a depth channel rides behind the user channels (its gradient is
     * read from where it sits: record slot or featx); sh_degree >= 0: features [N,n_feat] were the channels 3.. and
     * v_features [N,n_feat] receives their gradient, v_colors (/ v_colors_rest) the SH coefficients'; sh_degree < 0:
     * v_colors [(C,)N,D] receives all of them.  depth_slot is ignored. */
    const float *featx, *features;
    float *v_featx, *v_features;
    int32_t n_feat, nxq, depth_channel, reserved_x;
    /* featx == NULL (the forward ran with on-demand N-D records, misplat_raster_args.lazy_colour with nxq > 0: it wrote no
     * featx): channels 4.. are read from features [N,n_feat] and the depth channel from depths [C*N] */
    const float* depths;
} misplat_raster_bwd_args;
int misplat_raster_bwd(const misplat_params* p, const misplat_raster_bwd_args* b, misplat_stream_t stream,
                       misplat_graph_cache* cache /* or NULL */);
/* Host-only query: what misplat_raster_bwd does with these arguments.  Bit 0: the two-launch form (zeros written in the
 * background of the compositing backward, one kernel for the flagged rows); bit 1: replayable as a graph.  < 0: error. */
int misplat_raster_bwd_plan(const misplat_params* p, const misplat_raster_bwd_args* b);
/* Host-side wait for phase A's count: the caller stores -1 in *n_isects_host before the call; returns the count as
 * soon as the device-to-host copy has landed, or -1 after timeout_us. */
int64_t misplat_wait_count(const volatile int64_t* n_isects_host, int64_t timeout_us);
/* hipMemsetAsync(dst, 0, bytes) on `stream` (used to clear tile_count before phase B is enqueued a second time). */
int misplat_zero_bytes(void* dst, size_t bytes, misplat_stream_t stream);

/* ---- optimiser step --------------------------------------------
 * Fused multi-tensor Adam with torch.optim.Adam semantics (no weight decay, no amsgrad, maximize = False),
 * replacing the six per-group optimisers the reference configures at
 * collab_splats/configs/rade_gs_method.py:44-71 (lr per group, eps = 1e-15):
 *   m = m + (g - m)(1 - beta1);  v = beta2 v + (1 - beta2) g^2;
 *   p -= lr / (1 - beta1^t) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
 * params / grads / exp_avg / exp_avg_sq: HOST arrays of n_tensors device pointers (16-byte aligned, fp32,
 * contiguous); numel, lr, step (t >= 1, already incremented): host arrays.  One launch, in place.
 * beta1 / beta2 / eps are doubles because torch forms 1 - beta and the bias corrections in double. */
#define MISPLAT_ADAM_MAX_TENSORS 8
int misplat_adam_step(int32_t n_tensors, float* const* params, const float* const* grads,
                      float* const* exp_avg, float* const* exp_avg_sq, const int64_t* numel,
                      const float* lr, const int64_t* step, double beta1, double beta2, double eps,
                      misplat_stream_t stream);

/* ---- shared-Gaussian gradient reduce that moves only the rows that have a gradient (the gradient
 * set of collab_splats/configs/rade_gs_method.py:44-71; csrc/optim.hip, driven by collab_splats_amd/parallel.py).
 *   touched_bits  bits[ceil(n_rows / 8)]: bit (r & 7) of byte r >> 3 = flags[r] != 0 (flags: misplat_params.touched, 8-byte
 *                 aligned) -- the bitmap a rank contributes to the all-gather;
 *   union_count   gathered[world][nbytes] = the ranks' bitmaps; block_counts[ceil(nbytes / 256)] = rows set in the OR of
 *                 them, per block of 2 048 rows;
 *   union_scan    block_offsets[b] = sum of block_counts[0 .. b) (int64), *total = the union's size (one workgroup: the counts
 *                 are a few thousand; the caller reads the total when -- and if -- it needs it on the host);
 *   union_ids     ids[block_offsets[b] ...] = the set rows of block b, ascending: the same list on every rank;
 *   rows_pack     packed[u][W] = the rows ids[u] of n_tensors (<= 8) row-major tensors of `widths` floats per row, side by
 *                 side (W = sum of the widths); rows_unpack: the inverse (rows outside `ids` are not touched). */
#define MISPLAT_ROWS_MAX_TENSORS 8
int misplat_touched_bits(const uint8_t* flags, int64_t n_rows, uint8_t* bits, misplat_stream_t stream);
int misplat_union_count(const uint8_t* gathered, int32_t world, int64_t nbytes, int32_t* block_counts,
                        misplat_stream_t stream);
int misplat_union_scan(const int32_t* block_counts, int64_t n_blocks, int64_t* block_offsets, int64_t* total,
                       misplat_stream_t stream);
int misplat_union_ids(const uint8_t* gathered, int32_t world, int64_t nbytes, const int64_t* block_offsets, int32_t* ids,
                      int64_t ids_cap, misplat_stream_t stream);
/* count_dev (device, or NULL = n_ids): how many of the n_ids slots hold a row, for a caller that sized ids / packed from the
 * previous step's union and reads this step's size only after everything is enqueued: pack writes zeros into the slots behind
 * the count; unpack does NOTHING when the count exceeds n_ids (the caller then reduces the dense buffer instead). */
int misplat_rows_pack(int32_t n_tensors, const float* const* srcs, const int32_t* widths, const int32_t* ids, int64_t n_ids,
                      const int64_t* count_dev, float* packed, misplat_stream_t stream);
int misplat_rows_unpack(int32_t n_tensors, float* const* dsts, const int32_t* widths, const int32_t* ids, int64_t n_ids,
                        const int64_t* count_dev, const float* packed, misplat_stream_t stream);

/* Diagnosis (tests only; csrc/raster.hip): a 16-byte memset node + a kernel captured the way the graph cache captures, replayed
 * n_replays (<= 64) times with the 16 bytes overwritten with garbage in between; out_host[k] = the first counter after replay k
 * (= 1000 + k iff the memset node was applied).  counters16 / add8: device memory (16 / 8 bytes), out_host: host memory. */
int misplat_debug_memset_replay(void* counters16, void* add8, int32_t n_replays, int64_t* out_host, misplat_stream_t stream);

/* Measurement helper: dst[i] = src[i] over n_float4 16-byte elements (a plain streaming copy; bench.py times it to
 * report the HBM roof of the box it runs on).  variant: 0 plain, 1 non-temporal loads / stores, 2 four loads in flight
 * per lane + non-temporal stores, 3 the same with one contiguous piece per workgroup. */
int misplat_stream_copy(const void* src, void* dst, int64_t n_float4, int32_t variant, misplat_stream_t stream);

/* Library identification ("misplat <version> gfx950"). */
const char* misplat_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MISPLAT_H */
