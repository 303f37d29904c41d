// raster.hip -- the whole rasterization() forward as ONE host entry (two phases) instead of ~15 separate calls,
// optionally replayed as a hipGraph.  gfx950 only.  Replaces, on the host side, the kernel sequence inside
// gsplat-rade's rasterization(..., return_depth_normal=True) as called at
// /root/reference/collab_splats/models/rade_gs_model.py:439-465 (one call per training step on the caller's main
// thread): every launch below is one of the stages of include/misplat.h, enqueued back to back on the caller's
// stream with NO host read-back inside:
//   phase A  projection -> cell-ordered row bucketing (tile counts on the device) -> asynchronous copy of the
//            intersection count to pinned host memory -> colours (SH or pass-through)
//   phase B  per-tile buckets (capacity cap_isects) -> per-tile depth sort -> compositing -> launch order
// The caller may enqueue B right behind A with a SPECULATIVE capacity (e.g. 1.25 x the previous call's count) and
// only then wait for the count (misplat_wait_count): if it is <= cap_isects the results are exact; otherwise the
// caller allocates the exact size and enqueues B again.  The step's one host wait is thereby hidden behind the
// whole chain instead of stalling it.
// Small scenes are launch-bound (~16 launches of a few microseconds of GPU work each, ~6 us of host time per
// launch): with a misplat_graph_cache the sequence is captured once per distinct argument block and replayed with
// one hipGraphLaunch (~10-15 us of host time).  The cache is an explicit, caller-owned object: the library keeps
// no global state.  HBM-bound integer work + the VALU-bound compositing: no MFMA anywhere on this path.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <sched.h>
#include <time.h>
#include <mutex>
#include <vector>
#include "misplat.h"
#include "internal.h"

namespace {

int enqueue_colour(const misplat_params* p, const misplat_raster_args* a, misplat_stream_t stream) {
    return misplat_color_fwd(p, a->sh_degree, a->K_or_D, a->n_color, a->per_cam, a->depth_channel, a->means, a->viewmats,
                             a->colors, a->colors_rest, a->radii, a->depths, a->grec, a->sh_aux, a->v_grec_zero, stream);
}

int enqueue_forward(const misplat_params* p, const misplat_raster_args* a, int32_t phases, hipStream_t s) {
    misplat_stream_t stream = (misplat_stream_t)s;
    int rc;
    if (phases & 1) {
        // the projection kernel also clears everything the bucketing accumulates into -- cell counts, cell cursors, the
        // two counters, the tile counters: ONE contiguous range (cell_count ... tile_count) --, so phase A contains no
        // memset and no clearing kernel at all
        const int64_t n_tiles1 = (int64_t)p->tile_w * p->tile_h * p->n_cams + 1;
        if (!a->cell_count || !a->cell_cursor || !a->counters || !a->tile_count) return MISPLAT_EINVAL;
        const uint32_t* zend = (const uint32_t*)a->tile_count + n_tiles1;
        const int64_t n_zero = zend - a->cell_count;
        if (!(a->cell_count < a->cell_cursor && a->cell_cursor < (const uint32_t*)a->counters &&
              (const uint32_t*)a->counters + 4 <= (const uint32_t*)a->tile_count) || n_zero < 4 || n_zero > (1 << 22))
            return MISPLAT_EINVAL;
        // (on-demand colours: the colour slots start UNSET.  lazy_colour = 2: the compositing forward clears the gradient
        // row of every record whose colour it sets, and no other row is ever read -- the caller promises the backward
        // that reads flagged rows only (misplat_raster_bwd_plan bit 0) or clears v_grec itself first)
        rc = misplat_internal::project_pack_fwd(p, a->means, a->quats, a->scales, a->opacities, a->viewmats, a->Ks, a->radii,
                                                a->means2d, a->depths, a->compensations, a->grec, a->cell_count,
                                                (int32_t)n_zero, a->lazy_colour ? a->v_grec_zero : nullptr, a->v_abs_zero,
                                                a->lazy_colour == 2 ? 0 : 1, a->order_table, a->order_table ? a->order_sel : nullptr,
                                                a->order_slots, a->order_stride, s);
        if (rc != MISPLAT_OK) return rc;
        rc = misplat_bucket_count(p, a->means2d, a->radii, a->tiles_per_gauss, a->rect2, a->cellhist, a->cell_count,
                                  a->counters, 1, stream);
        if (rc != MISPLAT_OK) return rc;
        // (the intersection count reaches the caller's pinned slot as a store from bucket_rows: no copy node)
        rc = misplat_internal::bucket_rows(p, a->tiles_per_gauss, a->rect2, a->cellhist, a->cell_count, a->cell_cursor,
                                           a->cell_offs, a->order, a->rect_sorted, a->counters, a->tile_count, a->n_isects_host,
                                           3, a->depth_sorted ? a->depths : nullptr, a->depth_sorted, s);
        if (rc != MISPLAT_OK) return rc;
        if (a->nxq > 0) {
            // N-D channels (the features model): SH colours + feature channels written side by side, or pass-through
            if ((!a->featx && !a->lazy_colour) || a->nxq > 4) return MISPLAT_EINVAL;
            if (a->lazy_colour) {
                // on-demand N-D records: nothing to launch here -- the compositing forward evaluates SH, picks the features and
                // lays out featx for the records it stages (blend_fwd_x_lazy); SH colours + a feature tensor only
                if (a->sh_degree < 0 || !a->features || a->n_feat < 1 || a->nxq < 3 || p->n_cams != 1) return MISPLAT_EINVAL;
            } else if (a->sh_degree >= 0) {
                if (!a->features || a->n_feat < 1) return MISPLAT_EINVAL;
                rc = misplat_internal::color_fwd(p, a->sh_degree, a->K_or_D, 3, 0, 0, a->means, a->viewmats, a->colors,
                                                 a->colors_rest, a->radii, a->depths, a->grec, a->sh_aux, a->v_grec_zero,
                                                 a->features, a->n_feat, s);
                if (rc != MISPLAT_OK) return rc;
                rc = misplat_internal::color_fwd_x(p, a->n_feat, 3, 0, a->depth_channel, a->nxq, a->features, a->radii, a->depths,
                                                   a->grec, a->featx, nullptr, a->v_featx_zero, s);
            } else
                rc = misplat_internal::color_fwd_x(p, a->K_or_D, 0, a->per_cam, a->depth_channel, a->nxq, a->colors, a->radii,
                                                   a->depths, a->grec, a->featx, a->v_grec_zero, a->v_featx_zero, s);
            if (rc != MISPLAT_OK) return rc;
        } else if (!a->lazy_colour) {
            rc = enqueue_colour(p, a, stream);
            if (rc != MISPLAT_OK) return rc;
        }
    }
    if (phases & 2) {
        if (a->cap_isects < 0 || a->cap_isects > 0x7fffffffLL) return MISPLAT_EINVAL;
        // (depth_sorted given: bucket entries are positions in the cell-ordered row list, the sort's depth gather stays local)
        rc = misplat_internal::bucket_tiles(p, a->order, a->rect_sorted, a->counters, a->tile_count, a->offsets, nullptr,
                                            a->cap_isects, a->payload, nullptr, a->depth_sorted, s);
        if (rc != MISPLAT_OK) return rc;
        const float* sort_depths = a->depth_sorted ? a->depth_sorted : a->depths;
        const int32_t* sort_map = a->depth_sorted ? a->order : nullptr;
        const int sort_flags = a->depth_sorted ? 7 : 3;
        const int n_tiles = p->tile_w * p->tile_h * p->n_cams;
        const int64_t units = (int64_t)n_tiles * MISPLAT_BANDS;
        const int pivot_off = (int)(MISPLAT_ORDER_HEADER + 8 * ((units + 7) / 8));
        const bool by_view = a->order_table && a->order_sel && a->unit_work;
        if (by_view) {          // (a record must hold a whole permutation: the compositing launch indexes it by workgroup)
            if (a->order_slots < 1 || a->order_stride < pivot_off + (a->unit_reach ? n_tiles : 0)) return MISPLAT_EINVAL;
        }
        // front-only ordering: only with the view-keyed records (they carry the pivots) and the reach output that feeds them
        const bool front = by_view && a->unit_reach && a->front_n && a->tile_flag && a->depth_sorted;
        if ((a->front_n != nullptr) != (a->tile_flag != nullptr) || (a->front_n && !front)) return MISPLAT_EINVAL;
        if (front) {
            misplat_internal::FrontSort F;
            F.order_table = a->order_table; F.order_sel = a->order_sel; F.order_slots = a->order_slots;
            F.order_stride = a->order_stride; F.pivot_off = pivot_off;
            F.margin = a->front_margin > 0.f ? a->front_margin : 1.0f;      // (< 1: a pivot that is too shallow -- tests force the flag path with it)
            F.min_bucket = a->front_min_bucket > 0 ? a->front_min_bucket : 1;
            F.front_n = a->front_n; F.tile_flag = a->tile_flag;
            rc = misplat_internal::tile_sort_front(a->offsets, n_tiles, a->cap_isects, sort_depths, sort_map, a->payload,
                                                   a->flatten_ids, a->scratch, F, a->est_isects > 0 ? a->est_isects : -1, s);
            if (rc != MISPLAT_OK) return rc;
        } else if (a->cap_isects > 0) {
            rc = misplat_internal::tile_sort(a->offsets, n_tiles, a->cap_isects, a->est_isects > 0 ? a->est_isects : -1, sort_depths,
                                             sort_map, a->payload, a->flatten_ids, a->scratch, sort_flags, s);
            if (rc != MISPLAT_OK) return rc;
        }
        misplat_params q = *p;
        q.unit_perm = by_view ? a->order_table : a->unit_perm_in;
        q.unit_sel = by_view ? a->order_sel : nullptr;
        q.unit_stride = by_view ? a->order_stride : 0;
        q.unit_slots = by_view ? a->order_slots : 0;
        q.unit_work = a->unit_work;
        q.unit_reach = by_view ? a->unit_reach : nullptr;
        q.front_depths = a->depths;
        q.front_n = front ? a->front_n : nullptr;
        q.tile_flag = front ? a->tile_flag : nullptr;
        q.front_pass = 0;
        if (a->ev_blend_begin && hipEventRecord((hipEvent_t)a->ev_blend_begin, s) != hipSuccess) return MISPLAT_ELAUNCH;
        if (a->lazy_colour == 2 && (!a->v_grec_zero || (a->nxq > 0 && !a->v_featx_zero))) return MISPLAT_EINVAL;
        auto composite = [&]() -> int {
        if (a->nxq > 0 && a->lazy_colour)
            return misplat_internal::blend_fwd_x_lazy(&q, a->color_dim, a->nxq, a->Ks, a->grec, a->flatten_ids, a->offsets,
                                                      a->cap_isects, a->render, a->alpha, a->exp_depth, a->med_depth, a->normal,
                                                      a->last_ids, a->median_ids, a->means, a->viewmats, a->colors, a->colors_rest,
                                                      a->sh_degree, a->depth_channel, a->depths, a->features, a->n_feat,
                                                      a->lazy_colour == 2 ? a->v_grec_zero : nullptr,
                                                      a->lazy_colour == 2 ? a->v_featx_zero : nullptr, s);
        if (a->nxq > 0)
            return misplat_blend_fwd_x(&q, a->color_dim, a->nxq, a->Ks, a->grec, a->featx, a->flatten_ids, a->offsets,
                                       a->cap_isects, a->render, a->alpha, a->exp_depth, a->med_depth, a->normal, a->last_ids,
                                       a->median_ids, stream);
        if (a->lazy_colour) {
            return misplat_internal::blend_fwd_lazy(&q, a->color_dim, a->Ks, a->grec, a->flatten_ids, a->offsets, a->cap_isects,
                                                  a->render, a->alpha, a->exp_depth, a->med_depth, a->normal, a->last_ids,
                                                  a->median_ids, a->means, a->viewmats, a->colors, a->colors_rest,
                                                  a->sh_degree, a->depth_channel, a->depths, nullptr,
                                                  a->lazy_colour == 2 ? a->v_grec_zero : nullptr, s);
        }
        return misplat_blend_fwd(&q, a->color_dim, a->Ks, a->grec, a->flatten_ids, a->offsets, a->cap_isects, a->render,
                                 a->alpha, a->exp_depth, a->med_depth, a->normal, a->last_ids, a->median_ids, stream);
        };
        rc = composite();
        if (rc != MISPLAT_OK) return rc;
        if (front) {
            // the tiles whose pixels were still alive at the end of their truncated list: sorted in full, composited again
            rc = misplat_internal::tile_sort_flagged(a->offsets, n_tiles, a->cap_isects, sort_depths, sort_map, a->payload,
                                                     a->flatten_ids, a->scratch, a->tile_flag, s);
            if (rc != MISPLAT_OK) return rc;
            q.front_pass = 1;
            rc = composite();
            if (rc != MISPLAT_OK) return rc;
        }
        if (a->ev_blend_end && hipEventRecord((hipEvent_t)a->ev_blend_end, s) != hipSuccess) return MISPLAT_ELAUNCH;
        if (by_view) {
            rc = misplat_internal::unit_order_table(p, a->unit_work, a->order_table, a->order_sel, a->order_stride,
                                                    a->order_slots, a->unit_reach, s);
            if (rc != MISPLAT_OK) return rc;
        } else if (a->unit_work && a->unit_perm_out) {
            rc = misplat_unit_order(p, a->unit_work, a->unit_perm_out, stream);
            if (rc != MISPLAT_OK) return rc;
        }
    }
    return MISPLAT_OK;
}

struct GraphEntry {
    std::vector<uint8_t> key;
    uint64_t hash;                  // of key: the lookup compares 8 bytes per entry, the ~1 KB key only on a match
    hipGraphExec_t exec;
    hipGraph_t graph;               // the captured template stays alive as long as its executable does
    uint64_t stamp;
    hipStream_t last_stream;        // where it was launched last
    uint32_t hits = 0;              // replays of this graph
};
struct Retired {                    // an evicted graph may still be executing: destroyed once `done` has fired
    hipGraphExec_t exec;
    hipGraph_t graph;
    hipEvent_t done;
};

}  // namespace

struct misplat_graph_cache {
    std::mutex mu;
    std::vector<GraphEntry> entries;
    std::vector<Retired> retired;
    uint64_t clock = 0, hits = 0, captures = 0;
    std::vector<uint64_t> seen;                 // hashes of the last 1 024 argument blocks that missed (capture on second sighting)
    uint64_t seen_next = 0, window_start = 0, window_captures = 0;
    // A working set larger than the cache (more resident camera tensors than max_entries / 2) evicts every graph before its
    // block comes round again: each capture is then wasted.  Graphs evicted without a single replay are counted; 32 of them
    // and the cache stops capturing for the next 4 096 lookups (plain launches, which is what such a caller gets anyway).
    uint64_t wasted = 0, quiet_until = 0;
    int max_entries = 16;
    // Sequences are captured on this private stream (the caller's may be the legacy default stream, which cannot be
    // captured) and the resulting graph is launched on the caller's stream.
    hipStream_t capture_stream = nullptr;
};

extern "C" misplat_graph_cache* misplat_graph_cache_create(int32_t max_entries) {
    misplat_graph_cache* c = new (std::nothrow) misplat_graph_cache();
    if (!c) return nullptr;
    if (max_entries > 0) c->max_entries = max_entries;
    if (hipStreamCreateWithFlags(&c->capture_stream, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        misplat_graph_cache_destroy(c);
        return nullptr;
    }
    return c;
}

extern "C" void misplat_graph_cache_destroy(misplat_graph_cache* c) {
    if (!c) return;
    (void)hipDeviceSynchronize();                                   // nothing of ours is in flight any more
    for (auto& e : c->entries) { (void)hipGraphExecDestroy(e.exec); (void)hipGraphDestroy(e.graph); }
    for (auto& r : c->retired) { (void)hipGraphExecDestroy(r.exec); (void)hipGraphDestroy(r.graph); (void)hipEventDestroy(r.done); }
    if (c->capture_stream) (void)hipStreamDestroy(c->capture_stream);
    delete c;
}

extern "C" int misplat_graph_cache_stats(misplat_graph_cache* c, int64_t* hits, int64_t* captures) {
    if (!c || !hits || !captures) return MISPLAT_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    *hits = (int64_t)c->hits;
    *captures = (int64_t)c->captures;
    return MISPLAT_OK;
}

// Run `enqueue(stream)` through the cache: replay the graph captured for `key`, or capture it now.
template <class Enqueue>
static int run_cached(misplat_graph_cache* cache, std::vector<uint8_t>&& key, hipStream_t s, Enqueue enqueue) {
    std::lock_guard<std::mutex> g(cache->mu);
    cache->clock++;
    // retired graphs whose last launch has completed can go now
    for (size_t i = 0; i < cache->retired.size();) {
        if (hipEventQuery(cache->retired[i].done) == hipSuccess) {
            (void)hipGraphExecDestroy(cache->retired[i].exec);
            (void)hipGraphDestroy(cache->retired[i].graph);
            (void)hipEventDestroy(cache->retired[i].done);
            cache->retired.erase(cache->retired.begin() + i);
        } else {
            (void)hipGetLastError();                                // hipErrorNotReady is not an error
            i++;
        }
    }
    uint64_t h = 1469598103934665603ull;
    for (uint8_t bt : key) { h ^= bt; h *= 1099511628211ull; }
    for (auto& e : cache->entries)
        if (e.hash == h && e.key == key) {
            e.stamp = cache->clock;
            e.last_stream = s;
            e.hits++;
            cache->hits++;
            return hipGraphLaunch(e.exec, s) == hipSuccess ? MISPLAT_OK : MISPLAT_ELAUNCH;
        }
    // A capture + instantiation costs host time (of the order of a millisecond), a replay saves some tens of
    // microseconds: a caller whose argument blocks do not repeat (fresh camera tensors at addresses that never come back,
    // intersection capacities that drift) must not capture at all.  So a block is captured when it is seen for the
    // SECOND time: its first sighting only leaves a 64-bit hash in a ring of the last 1 024 misses and launches plainly.
    // A trainer that cycles through a few resident camera tensors pays one plain round, one capturing round and
    // replays from then on; a caller whose blocks never recur pays nothing.  Safety net for blocks that recur exactly
    // once (period-2 address patterns that then move on): at most max_entries captures per 256 calls.
    bool seen = false;
    for (uint64_t v : cache->seen) seen |= (v == h);
    if (cache->clock - cache->window_start >= 256) { cache->window_start = cache->clock; cache->window_captures = 0; }
    if (!seen || cache->window_captures >= (uint64_t)cache->max_entries || cache->clock < cache->quiet_until) {
        if (!seen) {
            if (cache->seen.size() < 1024) cache->seen.push_back(h);
            else cache->seen[cache->seen_next++ & 1023] = h;
        }
        return enqueue(s);
    }
    cache->window_captures++;
    // This call's work goes out plainly FIRST: the GPU runs it while the host records and instantiates the graph for the
    // next visit (capturing executes nothing).  Captured-then-launched, the capturing round of the bench's eight views ran
    // 0.06 ms per step behind the replaying rounds (1.338 vs 1.277 ms: the device waited for the host).
    const int rc_plain = enqueue(s);
    if (rc_plain != MISPLAT_OK) return rc_plain;
    // capture on the private stream (thread-local mode: other host threads keep using the runtime normally)
    hipStream_t cs = cache->capture_stream;
    if (hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal) != hipSuccess) {
        (void)hipGetLastError();
        return MISPLAT_OK;
    }
    const int rc = enqueue(cs);
    hipGraph_t graph = nullptr;
    const hipError_t ec = hipStreamEndCapture(cs, &graph);
    if (rc != MISPLAT_OK || ec != hipSuccess || !graph) {
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        return MISPLAT_OK;                                          // (the work itself is already out)
    }
    hipGraphExec_t exec = nullptr;
    const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (ei != hipSuccess || !exec) {
        (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
        return MISPLAT_OK;
    }
    if ((int)cache->entries.size() >= cache->max_entries) {       // evict the least recently used graph
        size_t victim = 0;
        for (size_t i = 1; i < cache->entries.size(); i++)
            if (cache->entries[i].stamp < cache->entries[victim].stamp) victim = i;
        if (cache->entries[victim].hits == 0 && ++cache->wasted >= 32) {
            cache->wasted = 0;
            cache->quiet_until = cache->clock + 4096;
        }
        Retired r{cache->entries[victim].exec, cache->entries[victim].graph, nullptr};
        // it may still be running: keep it until an event recorded behind its last launch has fired
        if (hipEventCreateWithFlags(&r.done, hipEventDisableTiming) == hipSuccess &&
            hipEventRecord(r.done, cache->entries[victim].last_stream) == hipSuccess) {
            cache->retired.push_back(r);
        } else {
            (void)hipGetLastError();
            (void)hipStreamSynchronize(cache->entries[victim].last_stream);
            (void)hipGraphExecDestroy(r.exec);
            (void)hipGraphDestroy(r.graph);
            if (r.done) (void)hipEventDestroy(r.done);
        }
        cache->entries.erase(cache->entries.begin() + victim);
    }
    cache->entries.push_back(GraphEntry{std::move(key), h, exec, graph, cache->clock, s, 0u});
    cache->captures++;
    return MISPLAT_OK;
}

template <class A>
static std::vector<uint8_t> make_key(int32_t tag, hipStream_t s, const misplat_params* p, const A* a) {
    // everything the enqueued work depends on: the two argument blocks, the entry / phases tag and the stream
    std::vector<uint8_t> key(sizeof(int32_t) + sizeof(void*) + sizeof(*p) + sizeof(*a));
    uint8_t* k = key.data();
    memcpy(k, &tag, sizeof(int32_t)); k += sizeof(int32_t);
    memcpy(k, &s, sizeof(void*)); k += sizeof(void*);
    memcpy(k, p, sizeof(*p)); k += sizeof(*p);
    memcpy(k, a, sizeof(*a));
    return key;
}

extern "C" int misplat_raster_fwd(const misplat_params* p, const misplat_raster_args* a, int32_t phases,
                                  misplat_stream_t stream, misplat_graph_cache* cache) {
    if (!p || !a || (phases & ~3) != 0 || phases == 0) return MISPLAT_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (!cache) return enqueue_forward(p, a, phases, s);
    return run_cached(cache, make_key(phases, s, p, a), s, [&](hipStream_t st) { return enqueue_forward(p, a, phases, st); });
}

// ---- the whole backward of rasterization(): compositing backward (atomic gradient rows), colour backward,
// projection backward -- three launches, no memset when the forward left cleared gradient rows behind
// In the background of the compositing backward (VALU-bound for ~0.5 ms, the memory system idle): the dense output
// gradients of the per-Gaussian kernels -- 236 bytes per Gaussian, nine tenths of them zeros in a dense scene -- are
// cleared by 512 extra one-wave workgroups at the END of that kernel's own grid (misplat_internal::FillList: they are
// dispatched when the machine starts to drain), and the two kernels then write only the rows that have a gradient.
// Measured on one box, 1 M / 1080p, ms per step, three alternating rounds: a small fill grid on a parallel graph branch
// 1.092 (the branch's fork and join edges cost ~10 us each on the main branch: kernel timeline), workgroups at the
// head of the grid 1.100 (512) / 1.085 (128) / 1.106 (2048), at the end 1.075 (512) / 1.077 (128).  At 5 M (1.18 GB
// of zeros, model step) the tail is too short for them: end 2.478 (512) / 2.492 (2048), head 2.448 (512) -- so the
// launcher puts them in front from 2.5 M rows.  Needs the row flags (misplat_params.touched), one camera, 16 SH
// coefficients without Jacobian cache.
static bool background_fill_ok(const misplat_params* p, const misplat_raster_bwd_args* b) {
    if (!(p->touched && p->n_cams == 1 && p->n_gauss >= 262144 && b->sh_degree >= 0 && !b->sh_aux && b->K_or_D == 16 &&
          !b->v_means2d && (b->colors_rest != nullptr) == (b->v_colors_rest != nullptr)))
        return false;
    // N-D records: SH colours + a feature tensor (the features model's call), whose gradient is one more tensor to clear
    if (b->nxq > 0 && !(b->features && b->v_features && b->v_featx && b->n_feat >= 1 &&
                        (((uintptr_t)b->v_features | (uintptr_t)b->v_featx) & 15) == 0))
        return false;
    const uintptr_t a16 = (uintptr_t)b->colors | (uintptr_t)b->v_colors | (uintptr_t)b->v_colors_rest | (uintptr_t)b->v_grec |
                          (uintptr_t)b->v_means | (uintptr_t)b->v_quats | (uintptr_t)b->v_scales | (uintptr_t)b->v_opacities |
                          (uintptr_t)b->v_means2d_out;
    return (a16 & 15) == 0 && (((uintptr_t)p->touched) & 7) == 0;
}

static int enqueue_backward(const misplat_params* p, const misplat_raster_bwd_args* b, hipStream_t s) {
    misplat_params q = *p;
    q.unit_perm = b->unit_perm;
    q.unit_sel = b->unit_perm ? b->unit_sel : nullptr;
    q.unit_stride = b->unit_stride;
    q.unit_slots = b->unit_slots;
    q.unit_work = nullptr;
    // Dense scenes: the per-Gaussian backward kernels only write the rows that received a gradient (`touched`); the
    // zeros of all the others are written in the background of the compositing backward (issue-bound, memory idle).
    const bool background = background_fill_ok(p, b);
    misplat_internal::FillList F = {};
    if (background) {
        const int64_t n = p->n_gauss;
        auto add = [&](float* ptr, int64_t count) { F.p[F.count] = ptr; F.n[F.count] = count; F.count++; };
        add(b->v_colors, n * (b->colors_rest ? 3 : 48));
        if (b->colors_rest) add(b->v_colors_rest, n * 45);
        add(b->v_means, n * 3);
        add(b->v_quats, n * 4);
        add(b->v_scales, n * 3);
        add(b->v_opacities, n);
        if (b->v_means2d_out) add(b->v_means2d_out, n * 2);
        if (b->nxq > 0) add(b->v_features, n * b->n_feat);
    }
    // the flagged-row form without absgrad: the rows are read by gauss_bwd_sparse alone, and their mean2d slots carry the two sums
    // that gradient is linear in (blend.hip, MSUM) -- formed per row there instead of per pixel here
#if defined(MISPLAT_DIAG_NO_MEAN_SUMS)          // (diagnostic build, scripts/build_variant.sh: the per-pixel form everywhere)
    const bool mean_sums = false;
#else
    const bool mean_sums = background && !b->v_abs;
#endif
    if (b->nxq > 0) {
        // N-D channels: compositing backward over record + featx rows, then the colour stage's backward, then the projection's
        if (!b->v_featx || b->nxq > 4 || (!b->featx && !(b->features && b->sh_degree >= 0))) return MISPLAT_EINVAL;
        if (b->ev_blend_begin && hipEventRecord((hipEvent_t)b->ev_blend_begin, s) != hipSuccess) return MISPLAT_ELAUNCH;
        int rx = misplat_internal::blend_bwd_x_atomic(&q, b->color_dim, b->nxq, b->Ks, b->grec, b->featx, b->flatten_ids, b->offsets,
                                                      b->n_isects, b->alpha, b->last_ids, b->median_ids, b->render, b->v_render,
                                                      b->v_alpha, b->v_exp_depth, b->v_med_depth, b->v_normal, b->v_grec,
                                                      b->v_featx, b->v_abs, b->zero_flags, background ? &F : nullptr, s,
                                                      b->features, b->n_feat, b->depth_channel, b->depths, mean_sums);
        if (rx != MISPLAT_OK) return rx;
        if (b->ev_blend_end && hipEventRecord((hipEvent_t)b->ev_blend_end, s) != hipSuccess) return MISPLAT_ELAUNCH;
        if (background)    // flagged rows only: SH backward, feature gradients and the projection backward in one launch
            return misplat_internal::gauss_bwd_sparse(p, b->sh_degree, -1, b->means, b->quats, b->scales, b->opacities, b->viewmats,
                                                      b->Ks, b->colors, b->colors_rest, b->compensations, b->v_grec, b->v_colors,
                                                      b->v_colors_rest, b->v_means, b->v_quats, b->v_scales, b->v_opacities,
                                                      b->v_means2d_out, s, b->v_featx, b->nxq, b->v_features, b->n_feat,
                                                      b->depth_channel ? 1 : 0, mean_sums);
        const int n_pre = b->sh_degree >= 0 ? 3 : 0, d_src = b->sh_degree >= 0 ? b->n_feat : b->K_or_D;
        if (b->sh_degree >= 0) {
            if (!b->v_features || !b->v_means_dir) return MISPLAT_EINVAL;
            rx = misplat_color_bwd(p, b->sh_degree, b->K_or_D, 3, 0, b->means, b->viewmats, b->colors, b->colors_rest, b->radii,
                                   b->v_grec, b->v_colors, b->v_colors_rest, b->v_means_dir, b->sh_aux, (misplat_stream_t)s);
            if (rx != MISPLAT_OK) return rx;
            rx = misplat_internal::color_bwd_x(p, d_src, 3, 0, b->nxq, b->radii, b->v_grec, b->v_featx, b->v_features, s);
        } else
            rx = misplat_internal::color_bwd_x(p, d_src, 0, b->per_cam, b->nxq, b->radii, b->v_grec, b->v_featx, b->v_colors, s);
        if (rx != MISPLAT_OK) return rx;
        int depth_slot = -1, stride = 0;
        const float* v_depth_rows = nullptr;
        if (b->depth_channel) {
            const int c = n_pre + d_src;                               // the depth rides behind the user channels
            if (c < 4) depth_slot = 12 + c;
            else { v_depth_rows = b->v_featx + (c - 4); stride = 4 * b->nxq; }
        }
        return misplat_project_pack_bwd(p, depth_slot, b->means, b->quats, b->scales, b->opacities, b->viewmats, b->Ks, b->radii,
                                        b->compensations, b->v_means2d, b->v_grec, b->sh_degree >= 0 ? b->v_means_dir : nullptr,
                                        b->v_means, b->v_quats, b->v_scales, b->v_opacities, v_depth_rows, stride, (misplat_stream_t)s);
    }
    if (b->ev_blend_begin && hipEventRecord((hipEvent_t)b->ev_blend_begin, s) != hipSuccess) return MISPLAT_ELAUNCH;
    int rc = misplat_internal::blend_bwd_atomic(&q, b->color_dim, b->Ks, b->grec, b->flatten_ids, b->offsets, b->n_isects,
                                                b->alpha, b->last_ids, b->median_ids, b->render, b->v_render, b->v_alpha,
                                                b->v_exp_depth, b->v_med_depth, b->v_normal, b->v_grec, b->v_abs, b->zero_flags,
                                                background ? &F : nullptr, s, mean_sums);
    if (rc != MISPLAT_OK) return rc;
    if (b->ev_blend_end && hipEventRecord((hipEvent_t)b->ev_blend_end, s) != hipSuccess) return MISPLAT_ELAUNCH;
    if (background)        // flagged rows only, both per-Gaussian stages in one launch
        return misplat_internal::gauss_bwd_sparse(p, b->sh_degree, b->depth_slot, b->means, b->quats, b->scales, b->opacities,
                                                  b->viewmats, b->Ks, b->colors, b->colors_rest, b->compensations, b->v_grec,
                                                  b->v_colors, b->v_colors_rest, b->v_means, b->v_quats, b->v_scales,
                                                  b->v_opacities, b->v_means2d_out, s, nullptr, 0, nullptr, 0, 0, mean_sums);
    rc = misplat_color_bwd(p, b->sh_degree, b->K_or_D, b->n_color, b->per_cam, b->means, b->viewmats, b->colors, b->colors_rest,
                           b->radii, b->v_grec, b->v_colors, b->v_colors_rest, b->v_means_dir, b->sh_aux, (misplat_stream_t)s);
    if (rc != MISPLAT_OK) return rc;
    return misplat_project_pack_bwd(p, b->depth_slot, b->means, b->quats, b->scales, b->opacities, b->viewmats, b->Ks, b->radii,
                                    b->compensations, b->v_means2d, b->v_grec, b->v_means_dir, b->v_means, b->v_quats, b->v_scales,
                                    b->v_opacities, nullptr, 0, (misplat_stream_t)s);
}

// What misplat_raster_bwd will do with these arguments (host only, nothing is enqueued): bit 0 -- the two-launch form
// (zeros in the background of the compositing backward + one kernel for the flagged rows); bit 1 -- replayable as a
// graph (memset-free, no measurement events).
extern "C" int misplat_raster_bwd_plan(const misplat_params* p, const misplat_raster_bwd_args* b) {
    if (!p || !b) return MISPLAT_EINVAL;
    const bool memset_free = (b->zero_flags & 1) && (!b->v_abs || (b->zero_flags & 2)) && (b->nxq == 0 || (b->zero_flags & 4));
    return (background_fill_ok(p, b) ? 1 : 0) | ((memset_free && !b->ev_blend_begin && !b->ev_blend_end) ? 2 : 0);
}

extern "C" int misplat_raster_bwd(const misplat_params* p, const misplat_raster_bwd_args* b, misplat_stream_t stream,
                                  misplat_graph_cache* cache) {
    if (!p || !b) return MISPLAT_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    // memset nodes are kept out of graphs (see the note on phase A): only the memset-free form is captured
    const bool memset_free = (b->zero_flags & 1) && (!b->v_abs || (b->zero_flags & 2)) && (b->nxq == 0 || (b->zero_flags & 4));
    if (!cache || !memset_free || b->ev_blend_begin || b->ev_blend_end) return enqueue_backward(p, b, s);
    return run_cached(cache, make_key(0x100, s, p, b), s, [&](hipStream_t st) { return enqueue_backward(p, b, st); });
}

// float4 streaming copy: the measured HBM roof of the box the benchmark runs on (bench.py reports fractions of it
// next to the 8 TB/s specification and the 6.29 TB/s the hardware guide measured).  Variants (bench.py takes the best):
//   0  one 16-byte load + store per lane and iteration, 2 048 workgroups, grid-stride
//   1  the same with non-temporal loads and stores (no reuse: keep the lines out of the caches' way)
//   2  four independent 16-byte loads per lane and iteration, then four non-temporal stores (more bytes in flight per CU)
typedef float copy_v4 __attribute__((ext_vector_type(4)));       // (the non-temporal builtins take native vector types)
template <int VARIANT>
__global__ __launch_bounds__(256) void stream_copy_kernel(const copy_v4* __restrict__ src, copy_v4* __restrict__ dst, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (VARIANT == 3) {
        // every workgroup streams ONE contiguous piece (consecutive 4 KB pages per wave front instead of lines 8 MB apart),
        // four 16-byte loads in flight per lane
        const int64_t per = (n + gridDim.x - 1) / gridDim.x;
        const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
        int64_t j = lo + threadIdx.x;
        for (; j + 3 * 256 < hi; j += 4 * 256) {
            const copy_v4 a = __builtin_nontemporal_load(src + j), b = __builtin_nontemporal_load(src + j + 256);
            const copy_v4 c = __builtin_nontemporal_load(src + j + 512), d = __builtin_nontemporal_load(src + j + 768);
            __builtin_nontemporal_store(a, dst + j); __builtin_nontemporal_store(b, dst + j + 256);
            __builtin_nontemporal_store(c, dst + j + 512); __builtin_nontemporal_store(d, dst + j + 768);
        }
        for (; j < hi; j += 256) dst[j] = src[j];
    } else if (VARIANT == 0) {
        for (; i < n; i += stride) dst[i] = src[i];
    } else if (VARIANT == 1) {
        for (; i < n; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
    } else {
        for (; i + 3 * stride < n; i += 4 * stride) {
            const copy_v4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride);
            const copy_v4 c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
            __builtin_nontemporal_store(a, dst + i); __builtin_nontemporal_store(b, dst + i + stride);
            __builtin_nontemporal_store(c, dst + i + 2 * stride); __builtin_nontemporal_store(d, dst + i + 3 * stride);
        }
        for (; i < n; i += stride) dst[i] = src[i];
    }
}

extern "C" int misplat_stream_copy(const void* src, void* dst, int64_t n_float4, int32_t variant, misplat_stream_t stream) {
    if (n_float4 < 0 || (n_float4 > 0 && (!src || !dst)) || variant < 0 || variant > 3) return MISPLAT_EINVAL;
    if (n_float4 == 0) return MISPLAT_OK;
    hipStream_t s = (hipStream_t)stream;
    if (variant == 0)
        hipLaunchKernelGGL(stream_copy_kernel<0>, dim3(256 * 8), dim3(256), 0, s, (const copy_v4*)src, (copy_v4*)dst, n_float4);
    else if (variant == 1)
        hipLaunchKernelGGL(stream_copy_kernel<1>, dim3(256 * 8), dim3(256), 0, s, (const copy_v4*)src, (copy_v4*)dst, n_float4);
    else if (variant == 2)
        hipLaunchKernelGGL(stream_copy_kernel<2>, dim3(256 * 8), dim3(256), 0, s, (const copy_v4*)src, (copy_v4*)dst, n_float4);
    else
        hipLaunchKernelGGL(stream_copy_kernel<3>, dim3(256 * 8), dim3(256), 0, s, (const copy_v4*)src, (copy_v4*)dst, n_float4);
    return hipGetLastError() == hipSuccess ? MISPLAT_OK : MISPLAT_ELAUNCH;
}

// Kernel-side fill (see internal.h: no memset node is ever enqueued by this library): bytes before the first 16-byte
// boundary and behind the last one go one at a time, the body as 16-byte stores.
__global__ __launch_bounds__(256) void fill_bytes_kernel(uint8_t* __restrict__ dst, size_t bytes, uint32_t word) {
    const uintptr_t a = (uintptr_t)dst;
    size_t head = (size_t)((16 - (a & 15)) & 15);
    if (head > bytes) head = bytes;
    const size_t n16 = (bytes - head) >> 4;
    const size_t tail0 = head + (n16 << 4);
    uint4* body = reinterpret_cast<uint4*>(dst + head);
    // the pattern is anchored at dst: byte i holds byte (i & 3) of `word`, so the body's words are `word` rotated by head
    const uint32_t sh = (uint32_t)(head & 3) * 8u;
    const uint32_t w = sh ? ((word >> sh) | (word << (32u - sh))) : word;
    const uint4 v = make_uint4(w, w, w, w);
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = tid; i < n16; i += stride) body[i] = v;
    if (blockIdx.x == 0) {
        for (size_t i = threadIdx.x; i < head; i += blockDim.x) dst[i] = (uint8_t)(word >> (8u * (uint32_t)(i & 3)));
        for (size_t i = tail0 + threadIdx.x; i < bytes; i += blockDim.x) dst[i] = (uint8_t)(word >> (8u * (uint32_t)(i & 3)));
    }
}

int misplat_internal::fill_bytes(void* dst, size_t bytes, uint32_t word, hipStream_t s) {
    if (!dst && bytes) return MISPLAT_EINVAL;
    if (bytes == 0) return MISPLAT_OK;
    size_t blocks = ((bytes >> 4) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(fill_bytes_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (uint8_t*)dst, bytes, word);
    return hipGetLastError() == hipSuccess ? MISPLAT_OK : MISPLAT_ELAUNCH;
}

// Clears a device buffer on `stream` (the retry of phase B clears tile_count with it: the buffers of one forward are
// slices of one allocation, so a framework-level in-place clear would invalidate tensors saved for the backward).
extern "C" int misplat_zero_bytes(void* dst, size_t bytes, misplat_stream_t stream) {
    return misplat_internal::fill_bytes(dst, bytes, 0u, (hipStream_t)stream);
}

// ---- diagnosis (tests only): does a MEMSET NODE inside a sequence captured the way run_cached captures (private
// non-blocking stream, thread-local mode) take effect on every replay?  Round 2 saw the bucketing counters come back as
// "garbage + n" from a replayed forward whose sequence began with hipMemsetAsync(counters, 0, 16) and designed every memset
// out without establishing the cause.  The sequence here is that one in miniature: memset 16 bytes ->
// a kernel adds `add[k]` into the two 64-bit counters; between replays a plain kernel overwrites the counters with garbage
// (what recycled allocator memory holds).  out[k] (k < n_replays) = counters[0] after replay k: equal to add[k] iff the
// memset node ran.  Returns MISPLAT_OK, or MISPLAT_ELAUNCH when the runtime refuses any step.
__global__ void debug_add_kernel(unsigned long long* c, const unsigned long long* add) { c[0] += add[0]; c[1] += add[0]; }
__global__ void debug_garbage_kernel(unsigned long long* c) { c[0] = 0xdeadbeefcafeull; c[1] = 0x123456789abcull; }
extern "C" int misplat_debug_memset_replay(void* counters16, void* add8, int32_t n_replays, int64_t* out_host, misplat_stream_t stream) {
    if (!counters16 || !add8 || !out_host || n_replays < 1 || n_replays > 64) return MISPLAT_EINVAL;
    hipStream_t s = (hipStream_t)stream, cs = nullptr;
    if (hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess) return MISPLAT_ELAUNCH;
    int rc = MISPLAT_ELAUNCH;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    do {
        if (hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal) != hipSuccess) break;
        const hipError_t em = hipMemsetAsync(counters16, 0, 16, cs);
        hipLaunchKernelGGL(debug_add_kernel, dim3(1), dim3(1), 0, cs, (unsigned long long*)counters16, (const unsigned long long*)add8);
        const hipError_t ek = hipGetLastError();
        if (hipStreamEndCapture(cs, &graph) != hipSuccess || em != hipSuccess || ek != hipSuccess || !graph) break;
        if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) break;
        rc = MISPLAT_OK;
        for (int k = 0; k < n_replays && rc == MISPLAT_OK; k++) {
            const unsigned long long addv = 1000ull + (unsigned long long)k;
            hipLaunchKernelGGL(debug_garbage_kernel, dim3(1), dim3(1), 0, s, (unsigned long long*)counters16);
            if (hipMemcpyAsync(add8, &addv, 8, hipMemcpyHostToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess ||
                hipGraphLaunch(exec, s) != hipSuccess ||
                hipMemcpyAsync(&out_host[k], counters16, 8, hipMemcpyDeviceToHost, s) != hipSuccess ||
                hipStreamSynchronize(s) != hipSuccess)
                rc = MISPLAT_ELAUNCH;
        }
    } while (0);
    (void)hipGetLastError();
    if (exec) (void)hipGraphExecDestroy(exec);
    if (graph) (void)hipGraphDestroy(graph);
    (void)hipStreamDestroy(cs);
    return rc;
}

// Host-side wait for the intersection count of phase A: *slot was set to -1 by the caller before the launch and is
// overwritten (>= 0) by the asynchronous device-to-host copy.  Returns the count, or -1 after timeout_us.
extern "C" int64_t misplat_wait_count(const volatile int64_t* slot, int64_t timeout_us) {
    if (!slot) return -1;
    struct timespec t0, t;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (uint32_t spin = 0;; spin++) {
        const int64_t v = *slot;
        if (v >= 0) return v;
        if ((spin & 63) == 63) {
            clock_gettime(CLOCK_MONOTONIC, &t);
            const int64_t us = (int64_t)(t.tv_sec - t0.tv_sec) * 1000000 + (t.tv_nsec - t0.tv_nsec) / 1000;
            if (us > timeout_us) return -1;
            if (us > 200) sched_yield();
        }
    }
}
