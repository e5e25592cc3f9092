// gswt_api.hip -- C ABI of libgswt_hip.so (include/gswt_hip.h): context, HBM buffers, frame slots (up to five frames in flight,
// each on its own stream with its own per-frame buffers), draw sets (one per frame in flight + 1, filled by sort events on a stream
// of their own) and the per-frame launch sequence
// cull -> project -> totals -> emit -> sort -> ranges -> items -> composite -> combine, launched kernel by kernel or replayed as one
// hipGraphLaunch (GSWT_OPT_GRAPH); background passes; sharding; the framebuffer gather (RCCL or peer copies).
//
// HBM layout (all resident, sized for a 288 GB part; nothing is re-uploaded per frame):
//   tex          U x 32 B      packed splat records, exactly Scene.tex_data (scene.rs:306-411)
//   static_list  4 B / entry   every base list [lod][tile][view], entry = gs_index | lod_id << 28,
//                              stored twice: interleaved (as the reference binds it) and
//                              LOD-filtered (what survives A1 for a plain tile)
//   merged_*     4+4 B / entry per-sort-event merged-group lists (gs_index|lod, map_id)
//   draws        DrawDev[]     one per reference draw call; chunk_tab maps a 256-entry workgroup
//                              to (draw, first entry)
//   rects/recs   8 + 32 B / slot  per-frame projection output, slot = composite order (+ 4 B depth side array when depth-tested)
//   keys/vals    2 x (4+4) B / pair  ping-pong for the tile sort
#include "../../include/gswt_hip.h"
#include "gswt_device.h"

#include <dlfcn.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

namespace gswt {
void launch_cull(hipStream_t, const Frame&, const DrawDev*, uint32_t, uint32_t*, uint32_t*, uint32_t, uint32_t*, uint32_t, uint32_t*, uint32_t, uint32_t*, uint32_t,
                 uint32_t*, uint4*, uint32_t*, uint32_t, const uint2*, uint32_t, const float*, bool, uint32_t*);
void launch_merge_copy(hipStream_t, const MergeCopy*, const uint2*, uint32_t, const uint2*, const MergeSources&, uint32_t*, uint32_t*);
void launch_chunk_tabs(hipStream_t, const DrawDev*, const uint32_t*, uint32_t, uint2*, uint2*, const uint64_t*, uint64_t);
size_t radix_ws_words(uint32_t, int);
size_t radix_ws_zero_words(uint32_t, int);
void launch_merge_build(hipStream_t, const MergeSeg*, uint32_t, const uint2*, uint32_t, MergeGroup*, uint32_t, const int32_t*, uint32_t, const unsigned long long*,
                        uint32_t*, uint32_t*, uint32_t*, uint32_t*, uint32_t*, int, uint32_t*, uint32_t*);
void launch_project(hipStream_t, bool, const Frame&, const DrawDev*, const uint2*, uint32_t, uint32_t, const uint32_t*, const uint32_t*,
                    const uint32_t*, const uint4*, const float*, const uint32_t*, const uint32_t*, uint32_t*, const uint4*, uint2*, Rec*, float*, uint32_t*, uint32_t*,
                    unsigned long long*, Varyings*, float4*, uint32_t, bool);
void launch_totals(hipStream_t, uint32_t*, uint32_t, unsigned long long*, uint32_t);
void launch_emit(hipStream_t, const Frame&, uint32_t, const uint2*, const uint32_t*, const uint32_t*, uint32_t, unsigned long long*,
                 uint32_t*, uint32_t*, const float*, uint32_t*, uint32_t*, const uint32_t*, const uint32_t*, uint32_t);
int launch_sort(hipStream_t, uint32_t*, uint32_t*, uint32_t*, uint32_t*, uint32_t, const unsigned long long*, int, uint32_t*, uint2* = nullptr,
                const uint32_t* = nullptr, uint32_t* = nullptr, uint32_t* = nullptr);
void launch_composite(hipStream_t, const Frame&, const uint2*, const uint32_t*, const Rec*, const float*, const float4*, const float4*, const float*, float4*, int, int,
                      uint32_t, uint32_t, uint32_t*, uint4*, float4*, hipEvent_t, hipEvent_t, unsigned long long*, unsigned long long*, int, const uint32_t*, uint32_t, uint32_t*, bool, bool);
void launch_tile_depth_sort(hipStream_t, const uint2*, uint32_t*, uint32_t*, uint32_t*, uint32_t*, int, uint32_t*, unsigned long long*);
uint32_t tile_depth_sort_cap();
void launch_unshard(hipStream_t, const float4*, float4*, int, int, int, int, int);
void launch_skybox(hipStream_t, const float*, float, float, int, int, int, int, const float4*, float4*);
void launch_proxy(hipStream_t, const ProxyArgs&, const float*, const float4*, float4*, float*);
void launch_fill_f32(hipStream_t, float*, size_t, float);
}  // namespace gswt

using namespace gswt;

namespace {

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;  // elements
    hipError_t ensure(size_t n, bool keep = false, hipStream_t s = nullptr)
    {
        if (n <= cap) return hipSuccess;
        size_t ncap = n + n / 4 + 1024;
        T* np = nullptr;
        // (GSWT_LOG_ALLOC=1: every device allocation of the library on stderr -- a growth inside a frame loop frees the old buffer,
        // which waits for the device)
        static const bool log_alloc = getenv("GSWT_LOG_ALLOC") != nullptr;
        if (log_alloc) fprintf(stderr, "gswt alloc: %zu -> %zu bytes%s\n", cap * sizeof(T), ncap * sizeof(T), p ? " (grow: frees the old buffer)" : "");
        hipError_t e = hipMalloc(&np, ncap * sizeof(T));
        if (e != hipSuccess) return e;
        if (keep && p && cap) hipMemcpy(np, p, cap * sizeof(T), hipMemcpyDeviceToDevice);
        (void)s;
        if (p) hipFree(p);
        p = np; cap = ncap;
        return hipSuccess;
    }
    // Buffers whose size follows the frame's pair count or a sort event's list sizes: when one has to grow it grows to TWICE the
    // request.  Growing frees the old buffer, which waits for the device -- with four frames in flight most of a millisecond, and
    // each frame slot / draw set repeats it when its turn comes (a fly path whose pair count crosses the old capacity stalled ~1 ms
    // per slot); 288 GB of HBM make the headroom cheap.
    hipError_t ensure_roomy(size_t n) { return n <= cap ? hipSuccess : ensure(2 * n); }
    void release() { if (p) hipFree(p); p = nullptr; cap = 0; }
};

// (pair_box / self_box: first chunk box of the list in static_boxes; chunk k = the k-th 256 entries from the END of the list, as k_project walks it)
struct ListRef { uint32_t pair_base, pair_count, self_base, self_count, pair_box, self_box; };

// Behind synchronous copies whose data the frames read: the frame slots' streams are non-blocking, i.e. not ordered behind the
// null stream, and a synchronous copy from pageable memory may return once the data is staged.  Setup paths only.
static inline hipError_t null_stream_done() { return hipStreamSynchronize(nullptr); }

// pinned host staging (asynchronous uploads read it after the call has returned)
template <typename T>
struct HostBuf {
    T* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t n)
    {
        if (n <= cap) return hipSuccess;
        const size_t ncap = n + n / 4 + 64;
        T* np = nullptr;
        hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&np), ncap * sizeof(T), hipHostMallocDefault);
        if (e != hipSuccess) return e;
        if (p) hipHostFree(p);
        p = np; cap = ncap;
        return hipSuccess;
    }
    void release() { if (p) hipHostFree(p); p = nullptr; cap = 0; }
};

}  // namespace

// Frame slots.  Round 1: c3 4 220 (two in flight) -> 4 600 frames/s (three); a fourth was SLOWER on a static camera (the frames then
// rotate over four sets of per-frame buffers and the working set outgrows the Infinity Cache).  Round 2: with sort events in the
// frame stream a fourth frame in flight covers the bubble a swap-in leaves (fly path 3 570 -> 3 770 frames/s) while it still costs
// a static camera 6 % (4 476 -> 4 214), so the library offers four and the caller decides how many it keeps in flight:
// gswt_render_async takes the lowest free slot, unused slots cost nothing.
// Frame slots = frames that can be in flight.  c3 fly path (worker thread + swap-ins), frames/s on one box with every slot in flight:
// 2: 3 770, 3: 4 095, 4: 4 445-4 500, 5: 4 770-4 790, 6: 4 700-4 765.  A static camera peaks at three in flight (bench.py keeps three
// there): more frames rotate over more sets of per-frame buffers and the working set outgrows the Infinity Cache.
#ifndef GSWT_FRAME_SLOTS
#define GSWT_FRAME_SLOTS 5
#endif
constexpr int kFrameSlots = GSWT_FRAME_SLOTS;
// Stream creation order (see gswt_create).  Measured on one box, c3 (tools: GSWT_STREAM_LAYOUT sweeps, gpurun_out/stream_layouts.txt):
//   layout        fly path, 4 / 5 in flight   static, 3 in flight   rank 0 of 8 (fake world): fly / static
//   c012p3s       4 492 / 4 658                4 924                 8 240 / 11 099      (fifth slot stream created at first use)
//   c012p34s      4 490 / 4 821                4 857                 8 055 /  8 849
//   c012p3ps      4 510 / 4 321                4 901                 7 790 /  8 767
//   c0123s        4 506 / 4 621                4 899                 7 317 / 10 241
// Four frames in flight do not care; the fifth frame pays only with its stream created in front of the build stream (which then
// shares its hardware queue with slot 2), and band frames of a sharded run want the older layout (bench.py sets it for --gpus N > 1).
constexpr const char* kStreamLayout = "c012p34s";

// The per-sort-event state (GSWTRenderer's swap-in of a SortData, state.rs:361-376): draw descriptors, chunk tables, merged
// lists, band-cull bounds.  Double-buffered: gswt_set_draws* fills the set that is NOT current while the frames in flight
// keep reading the one they were submitted with, so a sort event does not drain the frame pipeline.
// frames in flight + 1 (the set being refilled is never one a frame in flight still reads) + 4 more, so that with the sets refilled
// round robin the merged lists of the last kDrawSets - 1 = 9 sort events stay addressable for gswt_set_draws_merge_groups
constexpr int kDrawSets = kFrameSlots + 5;
static_assert(kDrawSets <= kMergeSources, "MergeSources holds one pointer pair per draw set");
template <typename T>
struct Ref { T* p = nullptr; };

struct DrawSet {
    // Everything a sort event uploads lives in ONE pinned host block mirrored by one device block of the same layout (a
    // single asynchronous copy on the ctx stream per event): draw records, per-draw XCD positions, and the tables of the
    // device-side merged-list step (groups to sort, their segments and block table; groups to copy, their block table and
    // map-id remap pairs; the sort's item count).  The views below point into the device block.
    HostBuf<uint8_t> h_blob;
    DevBuf<uint8_t> d_blob;
    size_t blob_bytes = 0;
    size_t off_draws = 0, off_xcd = 0, off_groups = 0, off_jobs = 0, off_remap = 0, off_segs = 0, off_blocks = 0, off_cblocks = 0, off_n64 = 0;
    template <typename T> T* hp(size_t off) { return reinterpret_cast<T*>(h_blob.p + off); }
    template <typename T> T* dp(size_t off) { return reinterpret_cast<T*>(d_blob.p + off); }
    hipError_t plan(size_t n_draws, size_t n_groups, size_t n_members, size_t total_entries)
    {
        size_t o = 0;
        auto take = [&o](size_t bytes) { const size_t at = o; o = (o + bytes + 255) & ~(size_t)255; return at; };
        const size_t n_blk = total_entries / 1024 + 2 * n_members + 2;          // upper bound of either block table
        off_draws = take((n_draws + 1) * sizeof(DrawDev)); off_xcd = take((n_draws + 1) * 4);
        off_groups = take((n_groups + 1) * sizeof(MergeGroup)); off_jobs = take((n_groups + 1) * sizeof(MergeCopy));
        off_remap = take((n_members + 1) * sizeof(uint2)); off_segs = take((2 * n_members + 1) * sizeof(MergeSeg));
        off_blocks = take(n_blk * sizeof(uint2)); off_cblocks = take(n_blk * sizeof(uint2)); off_n64 = take(64);
        blob_bytes = o;
        hipError_t e = o <= h_blob.cap ? hipSuccess : h_blob.ensure(2 * o);      // (grows to twice the request, like DevBuf::ensure_roomy)
        if (e != hipSuccess) return e;
        e = d_blob.ensure_roomy(o);
        if (e != hipSuccess) return e;
        draws.p = dp<DrawDev>(off_draws); xcd_first.p = dp<uint32_t>(off_xcd);
        return hipSuccess;
    }
    uint64_t per_xcd[8] = {};              // chunks per XCD launch list, and the longest of them
    uint64_t longest = 0;
    Ref<DrawDev> draws;
    DevBuf<uint2> chunk_tab;
    DevBuf<uint2> chunk_tab_xcd;           // chunk_tab in k_project's launch order: all chunks of a draw on one XCD (DrawDev::xcd)
    DevBuf<uint32_t> merged_list, merged_map;
    Ref<uint32_t> xcd_first;               // per draw: position of its first chunk in its XCD's launch list
    // what the merged arrays of this set hold, for the next sort event's reuse test (device-built sets only)
    struct GroupDesc { uint32_t view, base, len, first, n; uint64_t hash; };
    std::vector<GroupDesc> g_desc;
    std::vector<gswt_merge_member> g_members;
    bool g_valid = false;
    uint32_t src_mask = 0;                 // draw sets the device-side build of THIS set copies merged lists from (bit per set)
    hipEvent_t ev_up = nullptr;            // behind the upload: the pinned block may be refilled once it has fired
    bool ev_up_pending = false;
    bool built = false;                    // ev_up has been seen complete: frames on this set need not wait for it any more
    size_t n_merged = 0;
    uint32_t n_launch = 0;                 // length of chunk_tab_xcd (>= n_chunks: short per-XCD lists are padded)
    uint32_t n_draws = 0, n_chunks = 0;
    uint64_t n_entries = 0;
    void release()
    {
        chunk_tab.release(); chunk_tab_xcd.release(); merged_list.release(); merged_map.release(); h_blob.release(); d_blob.release();
        draws.p = nullptr; xcd_first.p = nullptr;
        if (ev_up) hipEventDestroy(ev_up);
        ev_up = nullptr;
    }
};

struct FrameArgs {
    gswt_camera_uniforms cam;
    gswt_scene_uniforms su;
    gswt_render_config cfg;
    int width = 0, height = 0;
    const float4* d_bg = nullptr;
    const float* d_bgd = nullptr;
    float4* d_out = nullptr;
};

// One frame in flight.  Each slot owns a stream and every per-frame buffer, so two frames overlap on the GPU:
// the latency-bound kernels of one (sort passes, single-workgroup scans, tails) fill the gaps of the other.
struct FrameSlot {
    hipStream_t stream = nullptr;
    hipEvent_t ev[10] = {};
    hipEvent_t ev_in = nullptr;            // recorded on the ctx stream at enqueue: the frame starts after it
    hipEvent_t ev_gather = nullptr;        // recorded on the ctx stream behind the frame's gather + re-assembly (gswt_render_gather / gswt_group_render_gather)
    bool gather_recorded = false;
    unsigned long long seq = 0;            // submission order of the frame in this slot
    unsigned long long* hc = nullptr;      // pinned host: [0] visible [1] pairs [2] scratch [3] overflow ... [7] staging
    unsigned long long* hc_dev = nullptr;  // the same words as the device sees them (k_combine writes [0..3] at the end of a frame)
    bool pending = false;                  // submitted through gswt_render_async, ticket not yet handed back by gswt_render_wait
    bool collected = false;                // finish_frame already ran for the pending frame (fence / gswt_set_draws*): its status and
    int collected_rc = 0;                  // timings wait here for gswt_render_wait
    gswt_timings collected_timings = {};
    FrameArgs args;
    int set = 0;                           // draw set the frame was submitted with (a re-run after overflow uses the same one)
    uint32_t cap = 0;
    int n_tiles = 0;
    int timing_level = 0;
    // per-frame HBM buffers
    DevBuf<uint2> rects;
    DevBuf<Rec> recs;
    DevBuf<uint4> live_tab;                // this frame's launch table of k_project: the chunks of the draws that survive k_cull
    DevBuf<uint32_t> live_cnt;             // entries per XCD list of live_tab (8 words, a cache line apart; zero between frames) + k_totals' copy for k_emit
    DevBuf<uint32_t> live_cid;             // live_tab's chunks as chunk numbers in slot order (k_emit walks the same table)
    DevBuf<uint32_t> cell_culled;          // column-band shards: per map cell, 1 = no splat of that tile instance can reach the band
    DevBuf<uint32_t> block_sums, draw_culled, scan_ws, keys_a, keys_b, vals_a, vals_b, ghist;
    DevBuf<uint2> ranges;
    DevBuf<uint32_t> item_base;
    DevBuf<uint4> item_tab;
    DevBuf<uint32_t> aux_a, aux_b;         // GSWT_ORDER_DEPTH: the pairs' tile ids, carried through the depth passes as the sort's payload
    bool strict_vs = false;                // GSWT_OPT_STRICT_VS as it stood when the frame was submitted (a re-run keeps it)
    uint32_t depth_passes = 0;             // GSWT_ORDER_DEPTH: radix passes this frame's depth sort was launched with
    bool depth_local = false;              // ... or the tile-local depth sort (k_tile_depth_sort)
    bool full_grid = false;                // this (re-run) frame launches k_project / k_emit over the whole launch table, whatever the hint says
    uint32_t n_launch_eff = 0;             // positions of the launch table this frame's grids cover
    DevBuf<float4> partials;
    DevBuf<float4> col_f;                  // debug draw modes: float colours per slot
    DevBuf<float> depths;                  // per-slot depth: frames with a proxy depth buffer or GSWT_ORDER_DEPTH only
    // hipGraph replay (GSWT_OPT_GRAPH): the chain of kernel nodes of this slot's frames and the argument records they were last set to
    GraphRec grec;
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    hipGraphNode_t graph_nodes[kGraphMaxNodes] = {};
    GraphNodeRec graph_last[kGraphMaxNodes];
    uint32_t graph_n = 0;
    void release_graph()
    {
        if (graph_exec) hipGraphExecDestroy(graph_exec);
        if (graph) hipGraphDestroy(graph);
        graph_exec = nullptr; graph = nullptr; graph_n = 0;
    }
    void release_buffers()
    {
        rects.release(); recs.release(); cell_culled.release(); live_tab.release(); live_cid.release(); live_cnt.release(); block_sums.release(); draw_culled.release(); scan_ws.release(); keys_a.release();
        keys_b.release(); vals_a.release(); vals_b.release(); ghist.release(); ranges.release(); item_base.release();
        aux_a.release(); aux_b.release(); partials.release(); item_tab.release(); col_f.release(); depths.release();
    }
};

struct gswt_ctx {
    int device = 0;
    FrameSlot slots[kFrameSlots];
    unsigned long long frame_seq = 0;      // frames submitted through gswt_render_async
    hipStream_t stream = nullptr;
    bool own_stream = true;
    std::string err;
    // scene
    DevBuf<uint4> tex;
    size_t n_splats = 0;
    float loc_lo[3] = {}, loc_hi[3] = {}, loc_max_trace = 0.0f;      // tile-local bounds of the splat centres, largest covariance trace
    DevBuf<uint32_t> static_list;
    DevBuf<float> static_boxes;            // tile-local bounding box (lo.xyz, hi.xyz) of every 256-entry chunk of every static list (k_live's chunk cull)
    std::vector<ListRef> lists;
    int n_lod = 0, n_tile = 0, n_view = 0;
    bool scene_ready = false;
    DevBuf<float> hmap;
    int hm_w = 0, hm_h = 0;
    // bounds of the height map for the column-band cull on the HeightMap surface: texel min / max, and the largest texel-to-texel
    // step along u and along v (repeat addressing) times the map's width / height = the largest slope of the bilinear surface
    // per unit of u / v (a bilinear sample lies between its texels, a difference quotient of it below the largest texel slope)
    float hm_min = 0.f, hm_max = 0.f, hm_du = 0.f, hm_dv = 0.f;
    // background passes
    DevBuf<float4> sky_faces;
    int sky_size = 0, sky_equi = 0;
    DevBuf<float4> proxy_tex;
    int proxy_size = 0, proxy_mips = 0, proxy_grid_dim = 2048;
    uint32_t proxy_mip_off[16] = {};
    // draws
    DrawSet sets[kDrawSets];
    int cur_set = 0;                       // the set frames submitted from now on read
    int latest_set = 0;                    // the set filled last (== cur_set unless a deferred swap-in is pending)
    int pending_set = -1;                  // GSWT_OPT_DEFER_SWAP: filled, still being built on set_stream, not yet read by frames
    hipStream_t set_stream = nullptr;      // uploads and device-side builds of a sort event: beside the frames, not in front of them
    std::vector<hipStream_t> pad_streams;  // never used: they steer the hardware-queue assignment (gswt_create)
    int opt_defer_swap = 0;
    int opt_graph = 0;
    // GSWT_OPT_STRICT_VS (default ON since round 4: k_project<.,.,STRICT> costs +1 us of 71 at c3 and nothing in frames/s): vs_main is
    // evaluated operator by operator as gswt.wgsl:152-258 writes it; 0 selects the fma-chain / single-reciprocal sequence v2
    int opt_strict_vs = 1;
    // GSWT_OPT_COMPOSITE: 0 = k_composite + k_combine, 1 = k_composite_dw (decoupled waves) + k_combine, 2 = k_composite<FOLD>: the last
    // segment of a tile to finish folds the partials, empty tiles are work items, no k_combine launch
    int opt_composite = 0;
    int opt_no_chunk_cull = 0;             // GSWT_OPT_NO_CHUNK_CULL: k_cull keeps every chunk of a surviving draw (A/B and tests: same image)
    unsigned long long stat_graph_launches = 0, stat_graph_rebuilds = 0, stat_graph_node_updates = 0;
    int pending_frames = 0;                // GSWT_OPT_DEFER_SWAP >= 2: frames still to be submitted on the old set
    int merge_target = 0;                  // gswt_set_draws_merge_groups -> set_draws_impl: the set being filled
    // on-device merged lists
    DevBuf<int32_t> raw_depth;
    std::vector<uint32_t> raw_off;          // [(lod*n_tile + tile)*n_view + view] -> offset in raw_depth
    std::vector<uint32_t> raw_cnt, raw_merge_offset;   // [lod*n_tile + tile]

    int opt_no_merge_reuse = 0;            // GSWT_OPT_NO_MERGE_REUSE: every merged group is re-sorted at every sort event
    unsigned long long stat_groups_built = 0, stat_groups_reused = 0, stat_groups_reused_deep = 0;
    DevBuf<uint32_t> mg_ws;
    bool draws_ready = false;
    // frame (the per-frame buffers live in the slots)
    uint32_t pair_cap = 0;                 // capacity the pair buffers / grids are sized for (grows on overflow)
    // GSWT_ORDER_DEPTH: the number of 8-bit passes of the depth sort: as many as the key ranges of the recent frames needed (the depths of one c3 frame span ~2^21
    // ulps: three).  A frame that needs more is flagged on the device and re-run; 32 frames in a row that need fewer give one back.
    uint32_t depth_passes = 3;
    uint32_t depth_passes_low_run = 0, depth_passes_low_max = 0;
    // ... or the tile-local path: tile passes first (depth bits as payload), then one kernel that depth-sorts each tile's slice in LDS.
    // GSWT_OPT_DEPTH_SORT: 0 / 2 = tile-local (lists of any length: the ones beyond the LDS buffer go through global memory,
    // k_tile_depth_sort_xl), 1 = the global passes.
    int opt_depth_sort = 0;
    int opt_item_order = 0;               // GSWT_OPT_ITEM_ORDER: 1 = the compositor's work items heaviest first (k_items)
    // Launch grids of k_project / k_emit: the launch table has a position for every chunk of the draw list, the frame's live chunks fill its
    // head (k_cull), and every position past an XCD's live count is a workgroup that starts, reads the count and leaves -- 300 k of them at c5.
    // The grids cover the longest live list of the last finished frame (k_totals reports it) + 50 % + 256; a frame whose own lists turn out
    // longer is flagged by k_totals and re-run with the full grid, like a pair overflow.  (+ 25 % + 64 was too tight on c3's fly path: a sort
    // event re-balances the lists, frames were re-run, 5 250-5 310 against 5 440-5 470 frames/s; with + 50 % c3's grid is the whole table again
    // -- 17.8 k positions for 10.7 k live chunks -- and c5's is 80 k of 366 k: 736-742 against 723-728 frames/s.  The cut is only taken where it
    // removes at least half of the grid.)
    uint32_t live_hint = 0;                // longest live list (per XCD) of the last finished frame; 0: none yet
    int opt_no_grid_hint = 0;              // GSWT_NO_GRID_HINT=1 (environment): always the full grid
    uint32_t depth_max_tile_len = 0;       // longest tile list of the last finished depth-ordered frame (0: none yet -- try tile-local)
    unsigned long long stat_depth_local = 0, stat_depth_global = 0;     // depth-ordered frames enqueued on either path (re-runs included)
    int last_slot = 0;
    DevBuf<float4> bg_rgba, out_img;
    DevBuf<float> bg_depth;
    DevBuf<Varyings> dbg;
    // options
    int opt_no_prefilter = 0;
    int opt_debug_varyings = 0;
    int opt_dbg_flags = 0;
    int opt_timing = 2;      // 0: no events, 1: frame + k_composite, 2: every stage
    // pairs per compositor work item (multiple of 256).  A tile's list is cut into segments that are composited in parallel and folded
    // by k_combine; a segment cannot know that the segments in front of it already saturated its pixels, so with the early-out on
    // (transmittance_eps > 0) short segments redo work that a longer one would have skipped.  k_composite alone, us (stage events):
    //   segment   512    768   1024   1536   2048   4096
    //   c3        96.3   96.9   96.0   95.7   97.9  141.5     (horizon tiles of 5-7 k pairs serialise at 4096)
    //   c3d (P = 8.2 M)  430    385    342    298    268    236
    //   c5        551     -     542     -     540     -
    // 1536 is the default; a host that knows its scene is dense raises it (bench.py: from the first frame's pairs per screen tile).
    int opt_segment = 1536;
    int opt_fixed_pair_cap = 0;   // test hook (GSWT_OPT_PAIR_CAP): the pair capacity is pinned until a frame overflows it
    uint32_t last_n_tiles = 0;
    gswt_timings timings = {};
    // multi-GPU gather: RCCL communicator (one process per GPU) or a local group of contexts (one process, peer copies)
    void* comm = nullptr;                  // ncclComm_t
    int comm_rank = 0, comm_world = 0;
    std::vector<gswt_ctx*> group;          // non-empty: hipMemcpyPeerAsync transport; group[r] is rank r
    int group_rank = 0;
    DevBuf<float4> gather_buf;             // world x shard image, as an all-gather delivers them
    hipEvent_t ev_push = nullptr;          // local group: this rank's shard has been pushed to every peer
    hipEvent_t ev_unshard = nullptr;       // local group: this rank's re-assembly of the PREVIOUS gather has read its gather buffer
    bool unshard_pending = false;          // ... and has been recorded at least once
};

namespace {

int fail(gswt_ctx* c, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return code;
}

 // waits for the ctx stream and every frame slot's stream
hipError_t sync_all(gswt_ctx* c)
{
    hipError_t e = hipStreamSynchronize(c->stream);
    if (c->set_stream) { hipError_t e2 = hipStreamSynchronize(c->set_stream); if (e == hipSuccess) e = e2; }
    for (auto& sl : c->slots)
        if (sl.stream) { hipError_t e2 = hipStreamSynchronize(sl.stream); if (e == hipSuccess) e = e2; }
    return e;
}

// No C++ exception may unwind through the C ABI: every extern "C" body that returns a status is a function-try-block
// closed by this handler (std::vector / std::string allocations of the draw-list code are the throwing sites).
#define GSWT_CATCH(NAME)                                                                   \
    catch (const std::bad_alloc&) { return GSWT_ERR_CAPACITY; }                             \
    catch (...) { return GSWT_ERR_HIP; }

#define HIP_TRY(c, expr)                                                                              \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) return fail((c), GSWT_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

// cgmath Matrix4 * Matrix4 (camera.rs:86-88): out[c][r] = sum_k a[k][r] * b[c][k], left to right
void mat4_mul(const float* a, const float* b, float* out)
{
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++) {
            float acc = a[r] * b[4 * c];
            for (int k = 1; k < 4; k++) acc = acc + a[4 * k + r] * b[4 * c + k];
            out[4 * c + r] = acc;
        }
}

}  // namespace

// ---- multi-GPU gather: transport state (the entry points are at the end of the file) ---------------------------------
namespace {

// RCCL through dlopen: a single-GPU host never needs the library, and a process that already holds one (PyTorch-ROCm bundles
// its own librccl.so) gets that copy instead of a second one.
struct Id128 { char b[GSWT_COMM_ID_BYTES]; };
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, /* ncclUniqueId by value: 128 bytes */ Id128, int) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool tried = false;
};
RcclApi g_rccl;

const char* rccl_load()
{
    if (g_rccl.lib) return nullptr;
    if (g_rccl.tried) return "librccl.so could not be loaded";
    g_rccl.tried = true;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.lib) break;
    }
    if (!g_rccl.lib) return "librccl.so could not be loaded";
    g_rccl.GetUniqueId = reinterpret_cast<int (*)(void*)>(dlsym(g_rccl.lib, "ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<int (*)(void**, int, Id128, int)>(dlsym(g_rccl.lib, "ncclCommInitRank"));
    g_rccl.AllGather = reinterpret_cast<int (*)(const void*, void*, size_t, int, void*, hipStream_t)>(dlsym(g_rccl.lib, "ncclAllGather"));
    g_rccl.CommDestroy = reinterpret_cast<int (*)(void*)>(dlsym(g_rccl.lib, "ncclCommDestroy"));
    g_rccl.GetErrorString = reinterpret_cast<const char* (*)(int)>(dlsym(g_rccl.lib, "ncclGetErrorString"));
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllGather || !g_rccl.CommDestroy) {
        dlclose(g_rccl.lib); g_rccl.lib = nullptr;
        return "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy";
    }
    return nullptr;
}

constexpr int kNcclFloat = 7;      // ncclFloat32 (rccl.h ncclDataType_t)

// geometry of a slot's shard image (what its frame wrote to args.d_out)
struct ShardGeom { int world, mode, out_rows, out_w; size_t px; };
ShardGeom shard_geom(const FrameSlot& sl)
{
    ShardGeom g;
    const gswt_render_config& cfg = sl.args.cfg;
    g.world = cfg.shard_count <= 1 ? 1 : cfg.shard_count;
    g.mode = cfg.shard_mode;
    const bool cols = g.world > 1 && g.mode == GSWT_SHARD_COLUMNS;
    g.out_rows = g.world > 1 && !cols ? gswt_shard_rows_padded(sl.args.height, g.world) : sl.args.height;
    g.out_w = cols ? gswt_shard_cols_padded(sl.args.width, g.world) : sl.args.width;
    g.px = (size_t)g.out_rows * g.out_w;
    return g;
}

}  // namespace

static int finish_frame(gswt_ctx* c, FrameSlot& sl);

// Runs every frame still in flight to completion (including the re-run of a frame whose pair buffers overflowed) while the
// state it was submitted with -- scene, draw list, capacities -- is still in place; the status and timings wait in the slot
// for the ticket's gswt_render_wait.  Called before anything that changes that state.
static hipError_t collect_pending(gswt_ctx* c)
{
    for (auto& sl : c->slots)
        if (sl.pending && !sl.collected) {
            sl.collected_rc = finish_frame(c, sl);
            sl.collected_timings = c->timings;
            sl.collected = true;
        }
    return sync_all(c);
}

// The merged lists the draw sets retain as copy sources belong to ONE scene and ONE set of raw depths: a re-uploaded scene of the same
// shape (same counts, other depths or gs_index values) would otherwise match the old lists by (view, member tile ids, length) from the
// second sort event on and copy the old scene's order.
static void invalidate_merge_sources(gswt_ctx* c)
{
    for (auto& ds : c->sets) { ds.g_valid = false; ds.g_desc.clear(); ds.g_members.clear(); ds.src_mask = 0; }
}

// Frames still in flight on draw set `set` are run to completion before that set is refilled.
// host_upload: the refill writes the set's merged arrays with copies that are NOT ordered on the build stream (gswt_set_draws with host
// lists), so a device-side build of another set that is still copying from them has to be waited for; a refill on the build stream
// (gswt_set_draws_merge_groups) is ordered behind such a copy anyway.
static void collect_set(gswt_ctx* c, int set, bool host_upload)
{
    for (auto& sl : c->slots)
        if (sl.pending && !sl.collected && sl.set == set) {
            sl.collected_rc = finish_frame(c, sl);
            sl.collected_timings = c->timings;
            sl.collected = true;
        }
    // ... and no device-side list build still in flight may be COPYING from this set's merged arrays (the lists of the last sort events
    // stay addressable as copy sources): everything that refills the set on the build stream is ordered behind such a copy anyway, a
    // host-side upload (gswt_set_draws with host lists) is not
    if (!host_upload) return;
    for (int k = 0; k < kDrawSets; k++) {
        DrawSet& o = c->sets[k];
        if (k != set && o.ev_up_pending && ((o.src_mask >> set) & 1u)) { (void)hipEventSynchronize(o.ev_up); o.ev_up_pending = false; }
    }
}

namespace gswt {
int ctx_device(const gswt_ctx* c) { return c->device; }     // gswt_worker.hip: the worker lives on its ctx's device
}

extern "C" {

int gswt_create(int device_id, gswt_ctx** out)
try {
    if (!out) return GSWT_ERR_BAD_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return GSWT_ERR_HIP;
    if (device_id < 0 || device_id >= ndev) return GSWT_ERR_BAD_ARG;
    if (hipSetDevice(device_id) != hipSuccess) return GSWT_ERR_HIP;
    gswt_ctx* c = new (std::nothrow) gswt_ctx();
    if (!c) return GSWT_ERR_CAPACITY;
    c->device = device_id;
    // Stream creation order matters: the runtime spreads streams over 4 hardware queues (GPU_MAX_HW_QUEUES) in creation order, and
    // streams that share a queue run one behind the other.  The layout string names the order: c = ctx stream, 0-4 = frame slot
    // streams, s = the stream of the sort-event builds, p = a placeholder that is never used; a slot stream the string leaves out is
    // created when that slot is first used.  (GSWT_STREAM_LAYOUT overrides it: tuning only.)
    c->opt_no_grid_hint = getenv("GSWT_NO_GRID_HINT") && atoi(getenv("GSWT_NO_GRID_HINT")) != 0;
    const char* layout = getenv("GSWT_STREAM_LAYOUT");
    if (!layout || !*layout) layout = kStreamLayout;
    for (const char* q = layout; *q; q++) {
        hipStream_t* dst = nullptr;
        hipStream_t pad = nullptr;
        if (*q == 'c') dst = &c->stream;
        else if (*q == 's') dst = &c->set_stream;
        else if (*q >= '0' && *q < '0' + kFrameSlots) dst = &c->slots[*q - '0'].stream;
        else if (*q == 'p') dst = &pad;
        else continue;
        if (*dst) continue;
        if ((q[1] == '+' || q[1] == '-')) {                   // "s+" / "s-": highest / lowest stream priority (tuning experiments)
            int lo = 0, hi = 0;
            hipDeviceGetStreamPriorityRange(&lo, &hi);          // lo = least, hi = greatest priority (numerically lower)
            if (hipStreamCreateWithPriority(dst, hipStreamNonBlocking, q[1] == '+' ? hi : lo) != hipSuccess) { delete c; return GSWT_ERR_HIP; }
        } else
        if (hipStreamCreateWithFlags(dst, hipStreamNonBlocking) != hipSuccess) { delete c; return GSWT_ERR_HIP; }
        if (dst == &pad) c->pad_streams.push_back(pad);
    }
    if (!c->stream && hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return GSWT_ERR_HIP; }
    if (!c->set_stream && hipStreamCreateWithFlags(&c->set_stream, hipStreamNonBlocking) != hipSuccess) { delete c; return GSWT_ERR_HIP; }
    for (auto& sl : c->slots) {
        if (hipEventCreateWithFlags(&sl.ev_in, hipEventDisableTiming) != hipSuccess) { delete c; return GSWT_ERR_HIP; }
        if (hipEventCreate(&sl.ev_gather) != hipSuccess) { delete c; return GSWT_ERR_HIP; }
        for (auto& e : sl.ev)
            if (hipEventCreate(&e) != hipSuccess) { delete c; return GSWT_ERR_HIP; }
        if (hipHostMalloc(reinterpret_cast<void**>(&sl.hc), 8 * sizeof(unsigned long long), hipHostMallocMapped) != hipSuccess) { delete c; return GSWT_ERR_HIP; }
        memset(sl.hc, 0, 8 * sizeof(unsigned long long));
        if (hipHostGetDevicePointer(reinterpret_cast<void**>(&sl.hc_dev), sl.hc, 0) != hipSuccess) sl.hc_dev = nullptr;   // then the copy stays
    }
    *out = c;
    return GSWT_OK;
} GSWT_CATCH("gswt_create")

void gswt_destroy(gswt_ctx* c)
{
    if (!c) return;
    hipSetDevice(c->device);
    sync_all(c);
    for (gswt_ctx* m : std::vector<gswt_ctx*>(c->group))       // leave a peer-copy group before the memory goes away
        if (m && m != c) m->group.clear();
    c->group.clear();
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    if (c->ev_push) hipEventDestroy(c->ev_push);
    if (c->ev_unshard) hipEventDestroy(c->ev_unshard);
    c->gather_buf.release();
    c->tex.release(); c->static_list.release(); c->static_boxes.release(); c->hmap.release(); for (auto& ds : c->sets) ds.release();
    c->raw_depth.release(); 
    c->mg_ws.release(); c->sky_faces.release(); c->proxy_tex.release(); c->bg_rgba.release(); c->out_img.release(); c->bg_depth.release(); c->dbg.release();
    for (auto& sl : c->slots) {
        sl.release_graph();
        sl.release_buffers();
        for (auto& e : sl.ev) if (e) hipEventDestroy(e);
        if (sl.ev_in) hipEventDestroy(sl.ev_in);
        if (sl.ev_gather) hipEventDestroy(sl.ev_gather);
        if (sl.hc) hipHostFree(sl.hc);
        if (sl.stream) hipStreamDestroy(sl.stream);
    }
    if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
    if (c->set_stream) hipStreamDestroy(c->set_stream);
    for (hipStream_t ps : c->pad_streams) hipStreamDestroy(ps);
    delete c;
}

const char* gswt_last_error(const gswt_ctx* c) { return c ? c->err.c_str() : "null ctx"; }

int gswt_set_stream(gswt_ctx* c, void* hip_stream)
try {
    if (!c) return GSWT_ERR_BAD_ARG;
    collect_pending(c);
    if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
    if (hip_stream) { c->stream = (hipStream_t)hip_stream; c->own_stream = false; return GSWT_OK; }
    c->stream = nullptr; c->own_stream = true;            // NULL: back to a stream of the ctx's own
    hipSetDevice(c->device);
    HIP_TRY(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    return GSWT_OK;
} GSWT_CATCH("gswt_set_stream")

int gswt_set_option(gswt_ctx* c, int key, int value)
try {
    if (!c) return GSWT_ERR_BAD_ARG;
    switch (key) {
    case GSWT_OPT_NO_LOD_PREFILTER: c->opt_no_prefilter = value; c->draws_ready = false; return GSWT_OK;
    case GSWT_OPT_DEBUG_VARYINGS: c->opt_debug_varyings = value; return GSWT_OK;
    case GSWT_OPT_DEBUG_FLAGS:
#ifndef GSWT_EXPERIMENTS
        // the ablation branches (wrong images by design) exist only in the measurement build: the product cannot be switched into them
        if (value != 0) return fail(c, GSWT_ERR_BAD_ARG, "GSWT_OPT_DEBUG_FLAGS: this library was built without -DGSWT_EXPERIMENTS (make variants)");
#endif
        c->opt_dbg_flags = value; return GSWT_OK;
    case GSWT_OPT_TIMING: c->opt_timing = value; return GSWT_OK;
    case GSWT_OPT_NO_MERGE_REUSE: c->opt_no_merge_reuse = value; return GSWT_OK;
    case GSWT_OPT_DEFER_SWAP: c->opt_defer_swap = value; return GSWT_OK;
    case GSWT_OPT_GRAPH: c->opt_graph = value; return GSWT_OK;
    case GSWT_OPT_STRICT_VS: c->opt_strict_vs = value != 0; return GSWT_OK;
    case GSWT_OPT_COMPOSITE:
        if (value < 0 || value > 2) return fail(c, GSWT_ERR_BAD_ARG, "unknown compositor variant %d", value);
        c->opt_composite = value; return GSWT_OK;
    case GSWT_OPT_NO_CHUNK_CULL: c->opt_no_chunk_cull = value != 0; return GSWT_OK;
    case GSWT_OPT_ITEM_ORDER: c->opt_item_order = value != 0; return GSWT_OK;
    case GSWT_OPT_DEPTH_SORT:
        if (value < 0 || value > 2) return fail(c, GSWT_ERR_BAD_ARG, "GSWT_OPT_DEPTH_SORT: 0 (auto), 1 (global passes) or 2 (tile-local)");
        c->opt_depth_sort = value; c->depth_max_tile_len = 0;
        return GSWT_OK;
    case GSWT_OPT_DEPTH_PASSES:
        if (value < 1 || value > 4) return fail(c, GSWT_ERR_BAD_ARG, "depth-sort passes must be 1..4");
        c->depth_passes = (uint32_t)value; c->depth_passes_low_run = 0;
        return GSWT_OK;
    case GSWT_OPT_PAIR_CAP:
        if (value < 0) return fail(c, GSWT_ERR_BAD_ARG, "pair capacity must be >= 0");
        c->opt_fixed_pair_cap = value > 0; if (value > 0) c->pair_cap = (uint32_t)value;
        return GSWT_OK;
    case GSWT_OPT_SEGMENT:
        if (value < 256 || value % 256) return fail(c, GSWT_ERR_BAD_ARG, "segment must be a positive multiple of 256");
        c->opt_segment = value; return GSWT_OK;
    default: return fail(c, GSWT_ERR_BAD_ARG, "unknown option %d", key);
    }
} GSWT_CATCH("gswt_set_option")

int gswt_upload_scene(gswt_ctx* c, const uint32_t* tex_data, size_t n_splats, const gswt_base_list* lists, int n_lod,
                      int n_tile, int n_view)
try {
    if (!c) return GSWT_ERR_BAD_ARG;
    if (!tex_data || n_splats == 0 || !lists || n_lod <= 0 || n_tile <= 0 || n_view <= 0)
        return fail(c, GSWT_ERR_BAD_ARG, "gswt_upload_scene: empty scene");
    if (n_lod > 16) return fail(c, GSWT_ERR_BAD_ARG, "gswt_upload_scene: n_lod %d > 16 (transition_dist_vec holds 16)", n_lod);
    if (n_splats > (size_t)kIdxMask) return fail(c, GSWT_ERR_CAPACITY, "gswt_upload_scene: %zu splats exceed 2^28", n_splats);
    hipSetDevice(c->device);
    HIP_TRY(c, collect_pending(c));
    c->scene_ready = false; c->draws_ready = false;
    invalidate_merge_sources(c);
    HIP_TRY(c, c->tex.ensure(2 * n_splats));
    HIP_TRY(c, hipMemcpy(c->tex.p, tex_data, n_splats * 32, hipMemcpyHostToDevice));
    c->n_splats = n_splats;
    {   // tile-local bounds of every splat centre and the largest covariance trace: what the band cull of column-sharded frames
        // places at a map cell's origin (every Wang-tile instance is the same tile-local content)
        float lo[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, hi[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
        float tr_max = 0.0f;
        bool odd = false;
        auto half_val = [](uint32_t h) -> float {           // halfToFloat of the shader (gswt.wgsl:478-494), upper bound is enough
            const uint32_t e = (h >> 10) & 0x1Fu, fr = h & 0x3FFu;
            if (e == 31u) return 0.0f;
            const float m = e == 0u ? (float)fr * 2.98023223876953125e-08f : ldexpf(1.0f + (float)fr / 1024.0f, (int)e - 15);
            return (h & 0x8000u) ? -m : m;
        };
        for (size_t i = 0; i < n_splats; i++) {
            const uint32_t* r = tex_data + 8 * i;
            float p[3];
            memcpy(p, r, 12);
            for (int k = 0; k < 3; k++) {
                if (!(p[k] == p[k]) || p[k] > 3e38f || p[k] < -3e38f) { odd = true; continue; }
                lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]);
            }
            const float tr = half_val(r[4] & 0xFFFFu) + half_val(r[5] >> 16) + half_val(r[6] >> 16);      // xx + yy + zz
            if (tr == tr) tr_max = std::max(tr_max, tr);
        }
        if (odd || lo[0] > hi[0]) { for (int k = 0; k < 3; k++) { lo[k] = -3.402823466e+38f; hi[k] = 3.402823466e+38f; } }    // never cull
        for (int k = 0; k < 3; k++) { c->loc_lo[k] = lo[k]; c->loc_hi[k] = hi[k]; }
        c->loc_max_trace = tr_max;
    }
    const size_t nl = (size_t)n_lod * n_tile * n_view;
    c->lists.assign(nl, ListRef{});
    std::vector<uint32_t> arena;
    std::vector<float> boxes;               // six floats per chunk
    // bounding box of every chunk of list [base, base + count): chunk k = entries count - 256 (k + 1) .. count - 256 k - 1 (k_project's order);
    // a chunk that holds a non-finite position gets the infinite box (never culled)
    auto add_boxes = [&](uint32_t base, uint32_t count) -> uint32_t {
        const uint32_t first = (uint32_t)(boxes.size() / 6);
        for (uint32_t k = 0; (size_t)k * kChunk < count; k++) {
            const uint32_t hi_i = count - k * (uint32_t)kChunk, lo_i = hi_i > (uint32_t)kChunk ? hi_i - (uint32_t)kChunk : 0u;
            float lo[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, hi[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
            bool odd = false;
            for (uint32_t j = lo_i; j < hi_i; j++) {
                float pq[3];
                memcpy(pq, tex_data + 8 * (size_t)(arena[base + j] & kIdxMask), 12);
                for (int a = 0; a < 3; a++) {
                    if (!(pq[a] == pq[a]) || pq[a] > 3e38f || pq[a] < -3e38f) odd = true;
                    lo[a] = std::min(lo[a], pq[a]); hi[a] = std::max(hi[a], pq[a]);
                }
            }
            for (int a = 0; a < 3; a++) boxes.push_back(odd ? -__builtin_inff() : lo[a]);
            for (int a = 0; a < 3; a++) boxes.push_back(odd ? __builtin_inff() : hi[a]);
        }
        return first;
    };
    size_t total = 0;
    for (size_t i = 0; i < nl; i++) total += lists[i].splat_count;
    arena.reserve(2 * total);
    for (size_t i = 0; i < nl; i++) {
        const gswt_base_list& L = lists[i];
        const uint32_t lod = (uint32_t)(i / ((size_t)n_tile * n_view));
        if (L.splat_count && (!L.gs_index || !L.gs_lod_id)) return fail(c, GSWT_ERR_BAD_ARG, "gswt_upload_scene: list %zu has null arrays", i);
        ListRef ref;
        ref.pair_base = (uint32_t)arena.size();
        ref.pair_count = L.splat_count;
        for (uint32_t j = 0; j < L.splat_count; j++) {
            if (L.gs_index[j] >= n_splats) return fail(c, GSWT_ERR_BAD_ARG, "gswt_upload_scene: list %zu entry %u out of range", i, j);
            if (L.gs_lod_id[j] > 15u) return fail(c, GSWT_ERR_BAD_ARG, "gswt_upload_scene: list %zu lod id out of range", i);
            arena.push_back(L.gs_index[j] | (L.gs_lod_id[j] << kLodShift));
        }
        ref.self_base = (uint32_t)arena.size();
        for (uint32_t j = 0; j < L.splat_count; j++)
            if (L.gs_lod_id[j] == lod) arena.push_back(L.gs_index[j] | (lod << kLodShift));
        ref.self_count = (uint32_t)arena.size() - ref.self_base;
        ref.pair_box = add_boxes(ref.pair_base, ref.pair_count);
        ref.self_box = add_boxes(ref.self_base, ref.self_count);
        c->lists[i] = ref;
    }
    if (arena.size() >= 0xFFFFFFFFull) return fail(c, GSWT_ERR_CAPACITY, "gswt_upload_scene: static lists exceed 2^32 entries");
    HIP_TRY(c, c->static_list.ensure(arena.size() + 1));
    HIP_TRY(c, hipMemcpy(c->static_list.p, arena.data(), arena.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(c, c->static_boxes.ensure(boxes.size() + 6));
    if (!boxes.empty()) HIP_TRY(c, hipMemcpy(c->static_boxes.p, boxes.data(), boxes.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(c, null_stream_done());
    c->n_lod = n_lod; c->n_tile = n_tile; c->n_view = n_view;
    c->scene_ready = true;
    return GSWT_OK;
} GSWT_CATCH("gswt_upload_scene")

int gswt_configure(gswt_ctx* c, const float* height_map, int hm_w, int hm_h)
try {
    if (!c) return GSWT_ERR_BAD_ARG;
    hipSetDevice(c->device);
    HIP_TRY(c, collect_pending(c));
    if (!height_map || hm_w <= 0 || hm_h <= 0) { c->hm_w = c->hm_h = 0; return GSWT_OK; }
    HIP_TRY(c, c->hmap.ensure((size_t)hm_w * hm_h + 2));         // + padding: k_project reads a cell's two texels of a row as one 8-byte load
    HIP_TRY(c, hipMemcpy(c->hmap.p, height_map, (size_t)hm_w * hm_h * 4, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemset(c->hmap.p + (size_t)hm_w * hm_h, 0, 8));
    {
        float mn = height_map[0], mx = height_map[0], du = 0.f, dv = 0.f;
        bool finite = true;
        for (int y = 0; y < hm_h; y++)
            for (int x = 0; x < hm_w; x++) {
                const float h = height_map[(size_t)y * hm_w + x];
                const float hr = height_map[(size_t)y * hm_w + (x + 1 == hm_w ? 0 : x + 1)], hd = height_map[(size_t)(y + 1 == hm_h ? 0 : y + 1) * hm_w + x];
                if (!(h == h) || h > 3e38f || h < -3e38f) finite = false;
                mn = std::min(mn, h); mx = std::max(mx, h);
                du = std::max(du, std::fabs(hr - h)); dv = std::max(dv, std::fabs(hd - h));
            }
        c->hm_min = mn; c->hm_max = mx; c->hm_du = du * (float)hm_w; c->hm_dv = dv * (float)hm_h;
        if (!finite) { c->hm_min = -3e38f; c->hm_max = 3e38f; c->hm_du = c->hm_dv = 3e38f; }      // the band cull then keeps everything
    }
    HIP_TRY(c, null_stream_done());
    c->hm_w = hm_w; c->hm_h = hm_h;
    return GSWT_OK;
} GSWT_CATCH("gswt_configure")

// Shared by gswt_set_draws (merged arrays from the host) and gswt_set_draws_merge_groups (built on the device:
// merged_gs_index == nullptr && device_merge).
// `device_merge`: called by gswt_set_draws_merge_groups, which has already planned the target set's upload block (it holds the
// merge tables too) and issues the one copy + the chunk-table kernel itself once its own tables are in place.
// Which draw set the next sort event fills, and when frames start reading it.  Sets rotate; the one after the set filled last is
// free once the frames still reading it have been collected.  By default a new set is current at once (the next frame waits for its
// build on the device).  With GSWT_OPT_DEFER_SWAP it becomes current with the first frame submitted AFTER its build has finished
// on set_stream: frames submitted meanwhile keep the previous list and nothing waits -- the reference's swap-in likewise takes
// effect with the frame after the worker's message (state.rs:361-376).  With a value n >= 2 it becomes current with the n-th frame
// submitted after the call whatever the device is doing (that frame waits if the build is late): ranks that render the shards of
// one frame stream then all switch at the same frame.  At most one set is pending: the next event makes it current.
static void activate_pending(gswt_ctx* c, bool force)
{
    if (c->pending_set < 0) return;
    DrawSet& P = c->sets[c->pending_set];
    bool now = force;
    if (!now && c->opt_defer_swap >= 2) now = c->pending_frames-- <= 0;                       // a fixed number of frames later: the same on every rank
    else if (!now) now = !P.ev_up || hipEventQuery(P.ev_up) == hipSuccess;                    // as soon as it has been built
    if (now) { c->cur_set = c->pending_set; c->pending_set = -1; }
}
static int next_target(gswt_ctx* c)
{
    activate_pending(c, true);
    return c->draws_ready ? (c->latest_set + 1) % kDrawSets : c->cur_set;
}
static void publish_set(gswt_ctx* c, int target)
{
    const bool first = !c->draws_ready || target == c->cur_set;
    c->latest_set = target;
    if (c->opt_defer_swap && !first) { c->pending_set = target; c->pending_frames = c->opt_defer_swap - 1; }
    else { c->cur_set = target; c->pending_set = -1; }
}

static int set_draws_impl(gswt_ctx* c, const gswt_draw* draws, int n_draws, const uint32_t* merged_gs_index,
                          const uint32_t* merged_map_id, const uint32_t* merged_lod_id, size_t n_merged, bool device_merge)
{
    if (!c) return GSWT_ERR_BAD_ARG;
    if (!c->scene_ready) return fail(c, GSWT_ERR_STATE, "gswt_set_draws before gswt_upload_scene");
    if (n_draws < 0 || (n_draws > 0 && !draws)) return fail(c, GSWT_ERR_BAD_ARG, "gswt_set_draws: bad draw list");
    if (!device_merge && n_merged && (!merged_gs_index || !merged_map_id)) return fail(c, GSWT_ERR_BAD_ARG, "gswt_set_draws: merged arrays missing");
    if (n_merged >= 0xFFFFFFFFull) return fail(c, GSWT_ERR_CAPACITY, "gswt_set_draws: merged lists exceed 2^32 entries");
    hipSetDevice(c->device);
    // The new list goes into the draw set that is not current; the frames in flight keep the set they were submitted with
    // (only a frame still running on the set being refilled -- two sort events old -- is waited for), so a sort event
    // does not drain the frame pipeline.  The upload is asynchronous on the ctx stream from the set's pinned staging; every
    // frame submitted afterwards starts behind an event recorded on that stream (enqueue_frame).
    const int target = device_merge ? c->merge_target : next_target(c);
    collect_set(c, target, !device_merge);
    DrawSet& D = c->sets[target];
    // the set's pinned staging is free again once its previous upload has been consumed
    if (D.ev_up_pending) { HIP_TRY(c, hipEventSynchronize(D.ev_up)); D.ev_up_pending = false; }
    if (!D.ev_up) HIP_TRY(c, hipEventCreateWithFlags(&D.ev_up, hipEventDisableTiming));
    if (!device_merge) HIP_TRY(c, D.plan((size_t)n_draws, 0, 0, 0));
    DrawDev* const dd = D.hp<DrawDev>(D.off_draws);
    uint32_t* const h_xcd_first = D.hp<uint32_t>(D.off_xcd);
    uint64_t entries = 0;
    for (int i = 0; i < n_draws; i++) {
        const gswt_draw& g = draws[i];
        DrawDev& d = dd[i];
        memset(&d, 0, sizeof(d));
        d.single_draw = g.tile.single_draw;
        d.valid_lod_id = g.tile.valid_lod_id;
        d.changing = g.tile.changing;
        d.changing_to_lower = g.tile.changing_to_lower;
        d.tile_lod = g.tile.tile_id[0];
        d.tile_idx = g.tile.tile_id[1]; d.tile_view = g.tile.tile_id[2];
        d.single_lod_id = g.tile.single_lod_id;
        d.map_coord[0] = g.tile.map_coord[0]; d.map_coord[1] = g.tile.map_coord[1];
        d.off[0] = g.tile.offset[0]; d.off[1] = g.tile.offset[1]; d.off[2] = g.tile.offset[2];
        d.cull_enable = g.cull_enable;
        d.lod = g.lod;
        memcpy(d.corners, g.corners, sizeof(d.corners));
        if (g.merged) {
            if ((size_t)g.merged_offset + g.merged_count > n_merged)
                return fail(c, GSWT_ERR_BAD_ARG, "gswt_set_draws: draw %d merged range out of bounds", i);
            if (g.merged_has_lod && !merged_lod_id && !device_merge)
                return fail(c, GSWT_ERR_BAD_ARG, "gswt_set_draws: draw %d needs merged_lod_id", i);
            if (g.tile.single_draw != 1u) return fail(c, GSWT_ERR_BAD_ARG, "gswt_set_draws: merged draw %d without single_draw", i);
            d.merged = 1; d.list_base = g.merged_offset; d.count = g.merged_count; d.box_base = 0xFFFFFFFFu;
        } else {
            if (g.tile.single_draw == 1u) return fail(c, GSWT_ERR_BAD_ARG, "gswt_set_draws: single_draw on static draw %d", i);
            if ((int)g.base_lod >= c->n_lod || (int)g.base_tile >= c->n_tile || (int)g.base_view >= c->n_view)
                return fail(c, GSWT_ERR_BAD_ARG, "gswt_set_draws: draw %d base list (%u,%u,%u) out of range", i, g.base_lod,
                            g.base_tile, g.base_view);
            const ListRef& L = c->lists[((size_t)g.base_lod * c->n_tile + g.base_tile) * c->n_view + g.base_view];
            const bool prefilter = !c->opt_no_prefilter && g.tile.valid_lod_id >= 0 && (uint32_t)g.tile.valid_lod_id == g.base_lod;
            d.merged = 0;
            d.list_base = prefilter ? L.self_base : L.pair_base;
            d.count = prefilter ? L.self_count : L.pair_count;
            d.box_base = prefilter ? L.self_box : L.pair_box;
        }
        d.entry_base = (uint32_t)entries;
        entries += d.count;
    }
    if (entries >= 0xFFFFFF00ull) return fail(c, GSWT_ERR_CAPACITY, "gswt_set_draws: %llu list entries exceed 2^32", (unsigned long long)entries);
    // composite-order slots: the LAST draw is nearest (drawn last = on top), so it gets the lowest slots.  A chunk = 256 list
    // entries of one draw; chunk c of the frame = slot c * 256.  The two chunk tables (slot order; k_project's launch order:
    // per-XCD lists, a draw's XCD = DrawDev::xcd, interleaved so that position p runs on XCD p % 8) are written on the DEVICE from the
    // draw records (k_chunk_tabs): the host only sums the O(#draws) counts, and a sort event uploads O(#draws) bytes.
    uint64_t slot = 0, per_xcd[8] = {};
    for (int i = n_draws - 1; i >= 0; i--) {
        DrawDev& d = dd[i];
        d.slot_base = (uint32_t)slot;
        const uint32_t nch = (d.count + kChunk - 1) / kChunk;
        // all chunks of a draw on one XCD (its gathers stay in that XCD's L2): the least loaded one.  (draw % 8 until round 4: a few merged
        // groups of ~190 chunks each made the longest of the eight lists 2.6 x the mean at c3 -- 42.8 k launch positions for 16.3 k chunks,
        // and every position past an XCD's live count is a workgroup that starts, reads the count and leaves.)
        int x = 0;
        for (int q = 1; q < 8; q++) if (per_xcd[q] < per_xcd[x]) x = q;
        d.xcd = (uint32_t)x;
        h_xcd_first[i] = (uint32_t)per_xcd[x];
        per_xcd[x] += nch;
        slot += (uint64_t)nch * kChunk;
    }
    if (slot >= 0xFFFFFFFFull) return fail(c, GSWT_ERR_CAPACITY, "gswt_set_draws: slot space exceeds 2^32");
    const size_t n_chunks = (size_t)(slot / kChunk);
    size_t longest = 0;
    for (int x = 0; x < 8; x++) longest = std::max<size_t>(longest, (size_t)per_xcd[x]);
    if (longest * 8 >= 0xFFFFFFFFull) return fail(c, GSWT_ERR_CAPACITY, "gswt_set_draws: chunk table too large");
    HIP_TRY(c, D.chunk_tab.ensure_roomy(n_chunks + 1));
    HIP_TRY(c, D.chunk_tab_xcd.ensure_roomy(longest * 8 + 1));
    D.n_launch = (uint32_t)(longest * 8);
    for (int x = 0; x < 8; x++) D.per_xcd[x] = per_xcd[x];
    D.longest = longest;
    if (!device_merge) {
        hipStream_t s = c->set_stream;
        HIP_TRY(c, hipMemcpyAsync(D.d_blob.p, D.h_blob.p, D.blob_bytes, hipMemcpyHostToDevice, s));      // the one upload of the event
        launch_chunk_tabs(s, D.draws.p, D.xcd_first.p, (uint32_t)n_draws, D.chunk_tab.p, D.chunk_tab_xcd.p, D.per_xcd, D.longest);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipEventRecord(D.ev_up, s));
        D.ev_up_pending = true; D.built = false;
    }
    // merged arrays: pack gs_index | lod << 28
    HIP_TRY(c, D.merged_list.ensure_roomy(n_merged + 1));
    HIP_TRY(c, D.merged_map.ensure_roomy(n_merged + 1));
    D.n_merged = n_merged;
    if (n_merged && !device_merge) {
        std::vector<uint32_t> packed(n_merged);
        for (size_t k = 0; k < n_merged; k++) {
            if (merged_gs_index[k] >= c->n_splats) return fail(c, GSWT_ERR_BAD_ARG, "gswt_set_draws: merged gs_index[%zu] out of range", k);
            packed[k] = merged_gs_index[k];
        }
        for (int i = 0; i < n_draws; i++) {
            const gswt_draw& g = draws[i];
            if (g.merged && g.merged_has_lod)
                for (uint32_t k = 0; k < g.merged_count; k++) {
                    uint32_t l = merged_lod_id[g.merged_offset + k];
                    if (l > 15u) return fail(c, GSWT_ERR_BAD_ARG, "gswt_set_draws: merged lod id out of range");
                    packed[g.merged_offset + k] |= l << kLodShift;
                }
        }
        HIP_TRY(c, hipMemcpy(D.merged_list.p, packed.data(), n_merged * 4, hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(D.merged_map.p, merged_map_id, n_merged * 4, hipMemcpyHostToDevice));
        HIP_TRY(c, null_stream_done());
    }
    D.n_draws = (uint32_t)n_draws;
    D.n_chunks = (uint32_t)n_chunks;
    D.g_valid = false;
    D.src_mask = 0;
    D.n_entries = entries;
    // The first draw list of a scene sizes the other draw sets too: every set's first fill used to allocate its own buffers (a
    // pinned upload block, its device mirror, chunk tables, merged arrays) inside the sort event that reached it -- the first five
    // swap-ins of a run each stalled the render thread for most of a millisecond.
    // A set that already has buffers grows with the largest one seen only while nothing reads it (no frame in flight, not the
    // current / pending / latest set: the next event's group copies read the latest one).
    for (int k = 0; k < kDrawSets; k++) {
        DrawSet& o = c->sets[k];
        if (&o == &D) continue;
        const bool empty = !o.d_blob.p && !o.chunk_tab.p && !o.merged_list.p;
        bool used = k == c->cur_set || k == c->pending_set || k == c->latest_set || o.ev_up_pending;
        for (const FrameSlot& fs : c->slots) used = used || (fs.pending && fs.set == k);
        if (used && !empty) continue;
        if (o.g_valid) continue;      // it holds the group lists of an earlier sort event that later events may still copy from
        if (o.h_blob.cap >= D.h_blob.cap && o.d_blob.cap >= D.d_blob.cap && o.chunk_tab.cap >= D.chunk_tab.cap &&
            o.chunk_tab_xcd.cap >= D.chunk_tab_xcd.cap && o.merged_list.cap >= D.merged_list.cap && o.merged_map.cap >= D.merged_map.cap) continue;
        HIP_TRY(c, o.h_blob.ensure(D.h_blob.cap)); HIP_TRY(c, o.d_blob.ensure(D.d_blob.cap));
        HIP_TRY(c, o.chunk_tab.ensure(D.chunk_tab.cap)); HIP_TRY(c, o.chunk_tab_xcd.ensure(D.chunk_tab_xcd.cap));
        {
            const uint32_t* const l0 = o.merged_list.p; const uint32_t* const m0 = o.merged_map.p;
            HIP_TRY(c, o.merged_list.ensure(D.merged_list.cap)); HIP_TRY(c, o.merged_map.ensure(D.merged_map.cap));
            if (o.merged_list.p != l0 || o.merged_map.p != m0) o.g_valid = false;      // its lists are gone: no later event may copy from them
        }
    }
    if (!device_merge) { publish_set(c, target); c->draws_ready = true; }      // (gswt_set_draws_merge_groups publishes behind its builds)
    return GSWT_OK;
}

int gswt_set_draws(gswt_ctx* c, const gswt_draw* draws, int n_draws, const uint32_t* merged_gs_index,
                   const uint32_t* merged_map_id, const uint32_t* merged_lod_id, size_t n_merged)
try {
    return set_draws_impl(c, draws, n_draws, merged_gs_index, merged_map_id, merged_lod_id, n_merged, false);
} GSWT_CATCH("gswt_set_draws")

int gswt_upload_raw_depth(gswt_ctx* c, const int32_t* const* raw_depth, const uint32_t* counts, const uint32_t* merge_offset)
try {
    if (!c || !raw_depth || !counts || !merge_offset) return GSWT_ERR_BAD_ARG;
    if (!c->scene_ready) return fail(c, GSWT_ERR_STATE, "gswt_upload_raw_depth before gswt_upload_scene");
    hipSetDevice(c->device);
    HIP_TRY(c, sync_all(c));
    invalidate_merge_sources(c);
    const size_t nlt = (size_t)c->n_lod * c->n_tile, nv = (size_t)c->n_view;
    c->raw_cnt.assign(counts, counts + nlt);
    c->raw_merge_offset.assign(merge_offset, merge_offset + nlt);
    c->raw_off.assign(nlt * nv, 0);
    size_t total = 0;
    for (size_t i = 0; i < nlt; i++) for (size_t v = 0; v < nv; v++) { c->raw_off[i * nv + v] = (uint32_t)total; total += counts[i]; }
    if (total >= 0xFFFFFFFFull) return fail(c, GSWT_ERR_CAPACITY, "gswt_upload_raw_depth: raw depth arena exceeds 2^32");
    std::vector<int32_t> arena(total);
    for (size_t i = 0; i < nlt; i++)
        for (size_t v = 0; v < nv; v++) {
            if (counts[i] && !raw_depth[i * nv + v]) return fail(c, GSWT_ERR_BAD_ARG, "gswt_upload_raw_depth: null array");
            if (counts[i]) memcpy(arena.data() + c->raw_off[i * nv + v], raw_depth[i * nv + v], (size_t)counts[i] * 4);
        }
    HIP_TRY(c, c->raw_depth.ensure(total + 1));
    if (total) HIP_TRY(c, hipMemcpy(c->raw_depth.p, arena.data(), total * 4, hipMemcpyHostToDevice));
    HIP_TRY(c, null_stream_done());
    return GSWT_OK;
} GSWT_CATCH("gswt_upload_raw_depth")

int gswt_set_draws_merge_groups(gswt_ctx* c, const gswt_draw* draws, int n_draws, const gswt_merge_group* groups, int n_groups,
                                const gswt_merge_member* members, int n_members)
try {
    if (!c) return GSWT_ERR_BAD_ARG;
    if (n_groups < 0 || n_members < 0 || (n_groups && (!groups || !members))) return fail(c, GSWT_ERR_BAD_ARG, "gswt_set_draws_merge_groups: bad groups");
    if (n_groups && c->raw_cnt.empty()) return fail(c, GSWT_ERR_STATE, "gswt_set_draws_merge_groups before gswt_upload_raw_depth");
    if (n_groups > 32768) return fail(c, GSWT_ERR_CAPACITY, "gswt_set_draws_merge_groups: more than 32768 merged groups");
    // Group ranges in the merged arrays (the concatenation order IS the merged-arena order).  A group whose view and ordered
    // member tids equal a group of the PREVIOUS sort event keeps that event's list: it is copied on the device from the
    // previous draw set with the members' map ids rewritten -- the reference's LRU hit (wangtile.rs:575-593) -- and only
    // the other groups go through the segmented sort.  Everything is staged in the target set's pinned memory.
    hipSetDevice(c->device);
    const int target = next_target(c);
    c->merge_target = target;
    collect_set(c, target, false);
    DrawSet& D = c->sets[target];
    // every other set that still holds the group lists of an earlier sort event, newest first (the sets are refilled round robin)
    std::vector<int> sources;
    if (c->draws_ready && !c->opt_no_merge_reuse)
        for (int j = 0; j < kDrawSets - 1; j++) {
            const int k = ((c->latest_set - j) % kDrawSets + kDrawSets) % kDrawSets;
            if (k != target && c->sets[k].g_valid && c->sets[k].merged_list.p) sources.push_back(k);
        }
    if (D.ev_up_pending) { HIP_TRY(c, hipEventSynchronize(D.ev_up)); D.ev_up_pending = false; }
    {   // upper bound of the merged entries (sizes the block tables of the upload block)
        size_t total_upper = 0;
        for (int q = 0; q < n_members; q++) {
            const gswt_merge_member& M = members[q];
            if ((int)M.lod < c->n_lod && (int)M.tile < c->n_tile) total_upper += c->raw_cnt[(size_t)M.lod * c->n_tile + M.tile];
            if (M.other_lod >= 0 && M.other_lod < c->n_lod && (int)M.tile < c->n_tile) total_upper += c->raw_cnt[(size_t)M.other_lod * c->n_tile + M.tile];
        }
        HIP_TRY(c, D.plan((size_t)std::max(n_draws, 0), (size_t)n_groups, (size_t)n_members, total_upper));
    }
    MergeSeg* const segs = D.hp<MergeSeg>(D.off_segs);
    MergeGroup* const grp = D.hp<MergeGroup>(D.off_groups);       // build space: only the groups that are sorted
    MergeCopy* const jobs = D.hp<MergeCopy>(D.off_jobs);
    uint2* const h_remap = D.hp<uint2>(D.off_remap);
    // hash -> (set, group) of every retained event; a set listed earlier (newer) wins on equal keys
    std::unordered_multimap<uint64_t, std::pair<int, uint32_t>> prev_by_hash;
    for (int k : sources) for (uint32_t q = 0; q < c->sets[k].g_desc.size(); q++) prev_by_hash.emplace(c->sets[k].g_desc[q].hash, std::make_pair(k, q));
    std::vector<DrawSet::GroupDesc> desc((size_t)n_groups);
    size_t n_segs = 0, n_build = 0, n_jobs = 0, n_remap = 0;
    uint64_t total = 0, build_total = 0;
    const size_t nv = (size_t)c->n_view;
    for (int g = 0; g < n_groups; g++) {
        const gswt_merge_group& G = groups[g];
        if ((uint64_t)G.first_member + G.n_members > (uint64_t)n_members || (int)G.view_id >= c->n_view)
            return fail(c, GSWT_ERR_BAD_ARG, "gswt_set_draws_merge_groups: group %d out of range", g);
        uint64_t h = 1469598103934665603ull ^ G.view_id, len = 0;
        for (uint32_t m = 0; m < G.n_members; m++) {
            const gswt_merge_member& M = members[G.first_member + m];
            const int lods[2] = {(int)M.lod, M.other_lod};
            for (int k = 0; k < 2; k++) {
                if (lods[k] < 0) continue;
                if (lods[k] >= c->n_lod || (int)M.tile >= c->n_tile) return fail(c, GSWT_ERR_BAD_ARG, "gswt_set_draws_merge_groups: member tid out of range");
                len += c->raw_cnt[(size_t)lods[k] * c->n_tile + M.tile];
            }
            h = (h ^ (((uint64_t)M.lod << 40) | ((uint64_t)M.tile << 8) | (uint64_t)(uint8_t)(M.other_lod + 1))) * 1099511628211ull;
        }
        DrawSet::GroupDesc& d = desc[g];
        d.view = G.view_id; d.base = (uint32_t)total; d.len = (uint32_t)len; d.first = G.first_member; d.n = G.n_members; d.hash = h;
        int match = -1, match_set = -1, match_rank = 1 << 30;
        if (!sources.empty() && len) {
            auto range = prev_by_hash.equal_range(h);
            for (auto it = range.first; it != range.second; ++it) {
                const DrawSet& ps = c->sets[it->second.first];
                const DrawSet::GroupDesc& p = ps.g_desc[it->second.second];
                if (p.view != G.view_id || p.n != G.n_members || p.len != (uint32_t)len || G.n_members > 256u) continue;
                bool same = true;
                for (uint32_t m = 0; m < G.n_members && same; m++) {
                    const gswt_merge_member& a = members[G.first_member + m];
                    const gswt_merge_member& b = ps.g_members[p.first + m];
                    same = a.lod == b.lod && a.tile == b.tile && a.other_lod == b.other_lod;
                }
                if (!same) continue;
                // the newest holder (fewest map ids to rewrite, and its buffers are the warmest)
                int rank = 0;
                while (rank < (int)sources.size() && sources[rank] != it->second.first) rank++;
                if (rank < match_rank) { match_rank = rank; match = (int)it->second.second; match_set = it->second.first; }
            }
        }
        if (match >= 0) {
            const DrawSet& ps = c->sets[match_set];
            const DrawSet::GroupDesc& p = ps.g_desc[match];
            MergeCopy jb;
            jb.src = p.base; jb.dst = (uint32_t)total; jb.len = (uint32_t)len; jb.first_pair = (uint32_t)n_remap; jb.n_pairs = 0; jb.src_set = (uint32_t)match_set; jb._pad[0] = jb._pad[1] = 0;
            bool moved = false;
            for (uint32_t m = 0; m < G.n_members; m++) moved = moved || members[G.first_member + m].map_index != ps.g_members[p.first + m].map_index;
            if (moved) {
                for (uint32_t m = 0; m < G.n_members; m++)
                    h_remap[n_remap++] = make_uint2(ps.g_members[p.first + m].map_index, members[G.first_member + m].map_index);
                jb.n_pairs = G.n_members;
            }
            jobs[n_jobs++] = jb;
            if (match_rank > 0) c->stat_groups_reused_deep++;
        } else if (len) {
            MergeGroup& B = grp[n_build];
            B.base = (uint32_t)build_total; B.len = (uint32_t)len; B.mn = 2147483647; B.mx = -2147483647 - 1; B.out_base = (uint32_t)total; B._pad[0] = B._pad[1] = B._pad[2] = 0;
            for (uint32_t m = 0; m < G.n_members; m++) {
                const gswt_merge_member& M = members[G.first_member + m];
                const int lods[2] = {(int)M.lod, M.other_lod};
                for (int k = 0; k < 2; k++) {
                    if (lods[k] < 0) continue;
                    const size_t lt = (size_t)lods[k] * c->n_tile + M.tile;
                    MergeSeg sg;
                    sg.group = (uint32_t)n_build; sg.src = c->raw_off[lt * nv + G.view_id]; sg.len = c->raw_cnt[lt]; sg.start = (uint32_t)build_total;
                    sg.gs_offset = c->raw_merge_offset[lt]; sg.map_index = M.map_index; sg.lod = (uint32_t)lods[k]; sg._pad = 0;
                    if (sg.len) segs[n_segs++] = sg;
                    build_total += sg.len;
                }
            }
            n_build++;
        }
        total += len;
    }
    if (total >= 0xFFFFFFFFull) return fail(c, GSWT_ERR_CAPACITY, "gswt_set_draws_merge_groups: merged lists exceed 2^32 entries");
    for (int i = 0; i < n_draws; i++)
        if (draws[i].merged) {
            const uint32_t g = draws[i].merged_group;
            if ((int)g >= n_groups || draws[i].merged_offset != desc[g].base || draws[i].merged_count != desc[g].len)
                return fail(c, GSWT_ERR_BAD_ARG, "gswt_set_draws_merge_groups: draw %d does not match group %u (offset %u/%u count %u/%u)", i, g,
                            draws[i].merged_offset, (int)g < n_groups ? desc[g].base : 0u, draws[i].merged_count, (int)g < n_groups ? desc[g].len : 0u);
        }
    int rc = set_draws_impl(c, draws, n_draws, nullptr, nullptr, nullptr, (size_t)total, true);       // fills the same target set
    if (rc != GSWT_OK) return rc;
    // (set_draws_impl grows the target's arrays and those of sets nothing reads; a source whose arrays it re-allocated lost g_valid:
    // that cannot be one of `sources` -- they are kept from growing below -- but it is checked all the same)
    MergeSources msrc;
    for (int k = 0; k < kMergeSources; k++) { msrc.list[k] = nullptr; msrc.map[k] = nullptr; }
    D.src_mask = 0;
    for (int k : sources) {
        if (!c->sets[k].g_valid) return fail(c, GSWT_ERR_STATE, "gswt_set_draws_merge_groups: a source set lost its lists");
        msrc.list[k] = c->sets[k].merged_list.p; msrc.map[k] = c->sets[k].merged_map.p;
    }
    for (size_t q = 0; q < n_jobs; q++) D.src_mask |= 1u << jobs[q].src_set;
    D.g_desc.swap(desc);
    D.g_members.assign(members, members + n_members);
    D.g_valid = false;                 // (a copy source only once everything below has been enqueued: a failed HIP call leaves unbuilt lists)
    c->stat_groups_built += n_build; c->stat_groups_reused += n_jobs;
    hipStream_t s = c->set_stream;
    // block tables: every copy job / segment cut into runs of <= 1024 entries (what one workgroup handles)
    uint2* const h_cblocks = D.hp<uint2>(D.off_cblocks);
    uint2* const h_blocks = D.hp<uint2>(D.off_blocks);
    size_t n_cb = 0, n_blocks = 0;
    for (size_t q = 0; q < n_jobs; q++)
        for (uint32_t off = 0; off < jobs[q].len; off += 1024u) h_cblocks[n_cb++] = make_uint2((uint32_t)q, off);
    for (size_t q = 0; q < n_segs; q++)
        for (uint32_t off = 0; off < segs[q].len; off += 1024u) h_blocks[n_blocks++] = make_uint2((uint32_t)q, off);
    const uint32_t n_total = (uint32_t)build_total;
    memset(D.hp<uint8_t>(D.off_n64), 0, 64);              // the sort reads its item count and an overflow word 16 bytes behind it
    *D.hp<unsigned long long>(D.off_n64) = n_total;
    // ONE upload for the whole event, then the chunk tables, the copies and (for the groups that changed) the segmented sort, all
    // on the ctx stream.  Nothing is waited for: every frame submitted from now on starts behind an event recorded on that
    // stream (enqueue_frame), i.e. behind the finished lists.  (A grown buffer is the exception: hipFree waits for the device.)
    HIP_TRY(c, hipMemcpyAsync(D.d_blob.p, D.h_blob.p, D.blob_bytes, hipMemcpyHostToDevice, s));
    launch_chunk_tabs(s, D.draws.p, D.xcd_first.p, D.n_draws, D.chunk_tab.p, D.chunk_tab_xcd.p, D.per_xcd, D.longest);
    if (n_jobs)
        launch_merge_copy(s, D.dp<MergeCopy>(D.off_jobs), D.dp<uint2>(D.off_cblocks), (uint32_t)n_cb, D.dp<uint2>(D.off_remap), msrc,
                          D.merged_list.p, D.merged_map.p);
    if (n_build) {
        int gbits = 1;
        while ((1u << gbits) < n_build) gbits++;
        const size_t radix_words = radix_ws_words(n_total, 16 + gbits);
        HIP_TRY(c, c->mg_ws.ensure_roomy(4 * (size_t)n_total + radix_words + 16));        // sort workspace, shared by all events (stream-ordered)
        uint32_t* w = c->mg_ws.p;
        uint32_t* radix = w + 4 * (size_t)n_total;
        HIP_TRY(c, hipMemsetAsync(radix, 0, (radix_words + 16) * 4, s));
        launch_merge_build(s, D.dp<MergeSeg>(D.off_segs), (uint32_t)n_segs, D.dp<uint2>(D.off_blocks), (uint32_t)n_blocks, D.dp<MergeGroup>(D.off_groups),
                           (uint32_t)n_build, c->raw_depth.p, n_total, D.dp<unsigned long long>(D.off_n64), w, w + n_total, w + 2 * (size_t)n_total,
                           w + 3 * (size_t)n_total, radix, gbits, D.merged_list.p, D.merged_map.p);
    }
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(D.ev_up, s));
    D.ev_up_pending = true; D.built = false;
    D.g_valid = true;
    publish_set(c, target);
    c->draws_ready = true;
    return GSWT_OK;
} GSWT_CATCH("gswt_set_draws_merge_groups")

int gswt_debug_read_merged(gswt_ctx* c, uint32_t* packed_list, uint32_t* map_id, size_t capacity, size_t* n)
try {
    if (!c || !n) return GSWT_ERR_BAD_ARG;
    const DrawSet& D = c->sets[c->cur_set];
    *n = D.n_merged;
    if (!packed_list || !map_id) return GSWT_OK;
    if (capacity < D.n_merged) return fail(c, GSWT_ERR_CAPACITY, "buffer holds %zu entries, need %zu", capacity, D.n_merged);
    hipSetDevice(c->device);
    HIP_TRY(c, sync_all(c));
    if (D.n_merged) {
        HIP_TRY(c, hipMemcpy(packed_list, D.merged_list.p, D.n_merged * 4, hipMemcpyDeviceToHost));
        HIP_TRY(c, hipMemcpy(map_id, D.merged_map.p, D.n_merged * 4, hipMemcpyDeviceToHost));
    }
    return GSWT_OK;
} GSWT_CATCH("gswt_debug_read_merged")

int gswt_shard_rows_padded(int height, int shard_count)
try {
    int tiles_y = (height + kTile - 1) / kTile;
    int sc = shard_count <= 1 ? 1 : shard_count;
    return ((tiles_y + sc - 1) / sc) * kTile;
} GSWT_CATCH("gswt_shard_rows_padded")

int gswt_shard_cols_padded(int width, int shard_count)
try {
    int tiles_x = (width + kTile - 1) / kTile;
    int sc = shard_count <= 1 ? 1 : shard_count;
    return ((tiles_x + sc - 1) / sc) * kTile;
} GSWT_CATCH("gswt_shard_cols_padded")

int gswt_shard_rows(int height, int shard_index, int shard_count)
try {
    int sc = shard_count <= 1 ? 1 : shard_count;
    if (shard_index < 0 || shard_index >= sc) return 0;
    int rows = 0;
    for (int y = 0; y < height; y++) if (((y / kTile) % sc) == shard_index) rows++;
    return rows;
} GSWT_CATCH("gswt_shard_rows")

// ---- frame machinery ------------------------------------------------------------------------------
// A frame is enqueued without any host round trip (enqueue_frame) and collected later (finish_frame).
// gswt_render = enqueue + finish; gswt_render_async / gswt_render_wait expose the two halves so a caller
// can queue frame i+1 before collecting frame i (two slots, same stream, executed in order).
static int validate_frame(gswt_ctx* c, const gswt_camera_uniforms* cam, const gswt_scene_uniforms* su, const gswt_render_config* cfg,
                          int width, int height, const void* out_rgba)
{
    if (!cam || !su || !cfg || !out_rgba) return fail(c, GSWT_ERR_BAD_ARG, "gswt_render: null argument");
    if (!c->draws_ready) return fail(c, GSWT_ERR_STATE, "gswt_render before gswt_set_draws");
    if (width <= 0 || height <= 0 || width > 65535 * kTile || height > 65535 * kTile)
        return fail(c, GSWT_ERR_BAD_ARG, "gswt_render: bad target size %dx%d", width, height);
    if ((float)width != cam->viewport[0] || (float)height != cam->viewport[1])
        return fail(c, GSWT_ERR_BAD_ARG, "gswt_render: camera viewport (%g,%g) != target %dx%d", cam->viewport[0], cam->viewport[1], width, height);
    if (su->surface_type == 1u && (c->hm_w == 0 || c->hm_h == 0))
        return fail(c, GSWT_ERR_STATE, "gswt_render: surface_type HeightMap without gswt_configure height map");
    if (su->surface_type > 2u) return fail(c, GSWT_ERR_BAD_ARG, "gswt_render: unknown surface_type %u", su->surface_type);
    if (su->surface_type == 2u && (su->map_half_wh[0] == 0u || su->map_half_wh[1] == 0u))
        return fail(c, GSWT_ERR_BAD_ARG, "gswt_render: Sphere surface needs a non-empty map");
    if (su->draw_mode > 4u) return fail(c, GSWT_ERR_BAD_ARG, "gswt_render: unknown draw_mode %u", su->draw_mode);
    if (cfg->order_mode != GSWT_ORDER_REFERENCE && cfg->order_mode != GSWT_ORDER_DEPTH)
        return fail(c, GSWT_ERR_BAD_ARG, "gswt_render: unknown order mode %d", cfg->order_mode);
    const int sc = cfg->shard_count <= 1 ? 1 : cfg->shard_count;
    if (sc > 1 && (cfg->shard_index < 0 || cfg->shard_index >= sc)) return fail(c, GSWT_ERR_BAD_ARG, "gswt_render: bad shard index");
    if (cfg->shard_mode != GSWT_SHARD_ROWS && cfg->shard_mode != GSWT_SHARD_COLUMNS) return fail(c, GSWT_ERR_BAD_ARG, "gswt_render: unknown shard mode %d", cfg->shard_mode);
    return GSWT_OK;
}

// Enqueues every kernel of one frame on the ctx stream.  All pointers in `a` are device pointers.
// GSWT_OPT_GRAPH: the frame's launch sequence was recorded, not launched (gswt_device.h); replay it as one hipGraphLaunch on the
// slot's stream.  The graph is a chain of kernel nodes; it is rebuilt when the sequence of kernels changes (another template
// variant, a k_radix_supscan more or less) and otherwise only the nodes whose grid or arguments differ from the previous frame of
// this slot are updated in the executable graph: with a moving camera the five kernels that take the frame constants, after a
// sort event also those that take draw-set pointers or sizes.
static int replay_graph(gswt_ctx* c, FrameSlot& sl)
{
    GraphRec& R = sl.grec;
    if (R.overflow || R.n == 0) return fail(c, GSWT_ERR_STATE, "gswt_render: frame graph record overflow");
    bool rebuild = !sl.graph_exec || sl.graph_n != R.n;
    for (uint32_t i = 0; !rebuild && i < R.n; i++) rebuild = sl.graph_last[i].fn != R.nodes[i].fn;
    void* ptrs[40];
    auto fill = [&](GraphNodeRec& nd, hipKernelNodeParams& kp) {
        memset(&kp, 0, sizeof(kp));
        for (uint32_t a = 0; a < nd.n_args; a++) ptrs[a] = nd.args + nd.offs[a];
        kp.func = const_cast<void*>(nd.fn);
        kp.gridDim = nd.grid; kp.blockDim = nd.block; kp.sharedMemBytes = 0;
        kp.kernelParams = ptrs; kp.extra = nullptr;
    };
    if (rebuild) {
        sl.release_graph();
        HIP_TRY(c, hipGraphCreate(&sl.graph, 0));
        for (uint32_t i = 0; i < R.n; i++) {
            hipKernelNodeParams kp;
            fill(R.nodes[i], kp);
            HIP_TRY(c, hipGraphAddKernelNode(&sl.graph_nodes[i], sl.graph, i ? &sl.graph_nodes[i - 1] : nullptr, i ? 1 : 0, &kp));
            sl.graph_last[i] = R.nodes[i];
        }
        HIP_TRY(c, hipGraphInstantiate(&sl.graph_exec, sl.graph, nullptr, nullptr, 0));
        sl.graph_n = R.n;
        c->stat_graph_rebuilds++;
    } else {
        for (uint32_t i = 0; i < R.n; i++) {
            if (sl.graph_last[i].same(R.nodes[i])) continue;
            hipKernelNodeParams kp;
            fill(R.nodes[i], kp);
            HIP_TRY(c, hipGraphExecKernelNodeSetParams(sl.graph_exec, sl.graph_nodes[i], &kp));
            sl.graph_last[i] = R.nodes[i];
            c->stat_graph_node_updates++;
        }
    }
    HIP_TRY(c, hipGraphLaunch(sl.graph_exec, sl.stream));
    c->stat_graph_launches++;
    return GSWT_OK;
}

static int enqueue_frame(gswt_ctx* c, FrameSlot& sl)
{
    const FrameArgs& a = sl.args;
    DrawSet& D = c->sets[sl.set];
    const gswt_camera_uniforms* cam = &a.cam;
    const gswt_scene_uniforms* su = &a.su;
    const gswt_render_config* cfg = &a.cfg;
    const int width = a.width, height = a.height;
    hipStream_t s = sl.stream;
    const int sc = cfg->shard_count <= 1 ? 1 : cfg->shard_count;
    {
        const size_t n_slots_all = (size_t)D.n_chunks * kChunk;
        HIP_TRY(c, sl.rects.ensure(n_slots_all + 1));
        HIP_TRY(c, sl.recs.ensure(n_slots_all + 1));
        HIP_TRY(c, sl.block_sums.ensure_roomy((size_t)D.n_chunks + 1));
        HIP_TRY(c, sl.live_tab.ensure_roomy((size_t)D.n_launch + 8));
        HIP_TRY(c, sl.live_cid.ensure_roomy((size_t)D.n_launch + 8));
        // (cleared ON THE SLOT'S STREAM: the slot streams are non-blocking, so a null-stream hipMemset -- asynchronous to the host for
        // device memory -- could land after this frame's k_cull had filled the counts: the slot's FIRST frame then projected nothing
        // and came back as background.  Seen once five slots made test_async_all_slots_in_flight_and_slot_reuse the first user of slot 4.)
        if (!sl.live_cnt.p) { HIP_TRY(c, sl.live_cnt.ensure(16 * kSuperStride)); HIP_TRY(c, hipMemsetAsync(sl.live_cnt.p, 0, sl.live_cnt.cap * 4, s)); }
        HIP_TRY(c, sl.draw_culled.ensure_roomy((size_t)D.n_draws + 1));
        if (su->draw_mode != 0u) HIP_TRY(c, sl.col_f.ensure(n_slots_all + 1));
        if (a.d_bgd || cfg->order_mode == GSWT_ORDER_DEPTH) HIP_TRY(c, sl.depths.ensure(n_slots_all + 1));
    }

    Frame f;
    memset(&f, 0, sizeof(f));
    memcpy(f.V, cam->view, 64);
    for (int cc = 0; cc < 4; cc++) {          // opengl_to_wgpu * projection, gswt.wgsl:152-160
        const float* P = cam->projection;
        f.GP[4 * cc + 0] = P[4 * cc + 0];
        f.GP[4 * cc + 1] = P[4 * cc + 1];
        f.GP[4 * cc + 2] = 0.5f * P[4 * cc + 2] + 0.5f * P[4 * cc + 3];
        f.GP[4 * cc + 3] = P[4 * cc + 3];
    }
    mat4_mul(cam->projection, cam->view, f.VP);
    f.focal[0] = cam->focal[0]; f.focal[1] = cam->focal[1];
    f.htan[0] = cam->htan_fov[0]; f.htan[1] = cam->htan_fov[1];
    f.cam_pos[0] = cam->cam_pos[0]; f.cam_pos[1] = cam->cam_pos[1]; f.cam_pos[2] = cam->cam_pos[2];
    f.W = (float)width; f.H = (float)height;
    f.splat_scale = su->splat_scale; f.tile_width = su->tile_width; f.clip_height = su->clip_height;
    f.point_cloud_radius = su->point_cloud_radius; f.transition_width_ratio = su->transition_width_ratio;
    f.sphere_radius = su->sphere_radius;
    f.use_clip = su->use_clip; f.surface_type = su->surface_type; f.num_lod = su->num_lod; f.draw_mode = su->draw_mode;
    f.map_half_wh[0] = su->map_half_wh[0]; f.map_half_wh[1] = su->map_half_wh[1];
    f.center_coord[0] = su->center_coord[0]; f.center_coord[1] = su->center_coord[1];
    memcpy(f.transition_dist, su->transition_dist_vec, 64);
    for (int k = 0; k < 3; k++) { f.height_map_scale[k] = su->height_map_scale[k]; f.scene_scale[k] = su->scene_scale[k]; }
    f.culling_dist = cfg->culling_dist; f.lod_enable_mask = cfg->lod_enable_mask; f.t_eps = cfg->transmittance_eps;
    f.has_depth = a.d_bgd ? 1 : 0;
    f.width = width; f.height = height;
    const int tiles_x_full = (width + kTile - 1) / kTile;
    const bool cols = sc > 1 && cfg->shard_mode == GSWT_SHARD_COLUMNS;
    f.tiles_x = tiles_x_full; f.tiles_y = (height + kTile - 1) / kTile;
    f.col0 = 0; f.col1 = tiles_x_full; f.out_w = width; f.out_x0 = 0; f.band_cull = 0;
    f.shard_index = sc > 1 && !cols ? cfg->shard_index : 0; f.shard_count = cols ? 1 : sc;
    if (cols) {                                   // contiguous band of tile columns, equal width on every rank
        const int band_tiles = (tiles_x_full + sc - 1) / sc;
        f.col0 = std::min(cfg->shard_index * band_tiles, tiles_x_full);
        f.col1 = std::min(f.col0 + band_tiles, tiles_x_full);
        f.tiles_x = f.col1 - f.col0;
        f.out_w = band_tiles * kTile; f.out_x0 = cfg->shard_index * band_tiles * kTile;
        // band culling bounds where a cell's splats can land: the plain surface (positions are the list positions) and the
        // HeightMap surface (the reference's default, structure.rs:75: the mapped centre is (x, y, h(x, y) hz) + n z with |n| = 1 and h
        // between the map's extremes; the covariance becomes F Vrk F^T with |F|_F^2 <= 3 + slope_x^2 + slope_y^2) and the
        // Sphere surface (cells whose footprint stays inside one block of the strip parametrisation: sphere_cell_box in gswt_kernels.hip;
        // the columns of F are finite differences of lz R, |.| <= 2.5 R / block_w, and lz itself).  Not the point-cloud covariance.
        f.band_cull = (su->surface_type <= 2u && !(su->point_cloud_radius > 0.0f)) ? 1 : 0;
    }
    for (int k = 0; k < 3; k++) { f.loc_lo[k] = c->loc_lo[k]; f.loc_hi[k] = c->loc_hi[k]; }
    f.loc_max_trace = c->loc_max_trace;
    f.surf_zlo = f.surf_zhi = 0.0f; f.surf_f2 = 1.0f;
    if (su->surface_type == 1u) {
        const float hz = su->height_map_scale[2];
        f.surf_zlo = std::min(c->hm_min * hz, c->hm_max * hz); f.surf_zhi = std::max(c->hm_min * hz, c->hm_max * hz);
        // slopes of the mapped surface per world unit: |dh/du| hz / x_range, |dh/dv| hz / y_range (gswt.wgsl:565-599)
        const float xr = (2.0f * (float)su->map_half_wh[0] + 1.0f) * su->tile_width * su->height_map_scale[0];
        const float yr = (2.0f * (float)su->map_half_wh[1] + 1.0f) * su->tile_width * su->height_map_scale[1];
        const float sx = c->hm_du * std::fabs(hz) / std::fabs(xr), sy = c->hm_dv * std::fabs(hz) / std::fabs(yr);
        f.surf_f2 = (3.0f + sx * sx + sy * sy) * 1.01f;
        // a non-finite height map (gswt_configure leaves +-3e38 bounds), an infinite or NaN range or slope bound: no band culling at all
        // (fminf / fmaxf drop NaN operands, so a box with NaN corners used to read as "misses the band")
        const float fmax_ = 3.0e38f;
        if (!(f.surf_f2 == f.surf_f2) || !(f.surf_zlo == f.surf_zlo) || !(f.surf_zhi == f.surf_zhi) || !(std::fabs(f.surf_f2) < fmax_) ||
            !(std::fabs(f.surf_zlo) < fmax_) || !(std::fabs(f.surf_zhi) < fmax_) || !(c->hm_du < fmax_) || !(c->hm_dv < fmax_)) f.band_cull = 0;
    }
    if (su->surface_type == 2u) {
        const float block_w = ((float)su->map_half_wh[0] * 2.0f) * su->tile_width / 5.0f;
        const float g = 2.5f * std::fabs(su->sphere_radius) / block_w;
        f.surf_f2 = (2.0f * g * g + 1.0f) * 1.01f;
        if (!(block_w > 0.0f) || !(f.surf_f2 == f.surf_f2) || su->map_half_wh[0] == 0u || su->map_half_wh[1] == 0u) f.band_cull = 0;
    }
    f.hm_w = c->hm_w; f.hm_h = c->hm_h;
    f.map_wh_y = 2u * su->map_half_wh[1] + (su->surface_type != 2u ? 1u : 0u);
    // (0 = "divide": a one-row map, map_wh_y == 1, would wrap the magic to 0 and the multiply-high quotient to 0 -- ADVICE r3)
    f.map_wh_y_magic = f.map_wh_y >= 2u && f.map_wh_y < 65536u ? 0xFFFFFFFFu / f.map_wh_y + 1u : 0u;
    f.tiles_x_magic = f.tiles_x > 0 && f.tiles_x < 65536 ? 0xFFFFFFFFu / (uint32_t)f.tiles_x + 1u : 0u;
    f.dbg_flags = c->opt_dbg_flags;

    const int rsc = f.shard_count;                                  // row-shard count (1 in column mode)
    const int tiles_y_local = rsc > 1 ? (f.tiles_y - f.shard_index + rsc - 1) / rsc : f.tiles_y;
    const int n_tiles = f.tiles_x * (tiles_y_local > 0 ? tiles_y_local : 0);
    const int out_rows = rsc > 1 ? gswt_shard_rows_padded(height, rsc) : height;
    const size_t out_px = (size_t)out_rows * f.out_w;
    sl.n_tiles = n_tiles;
    float4* const d_out = a.d_out;
    // (behind the per-tile ranges: one ticket word per tile for GSWT_OPT_COMPOSITE = 2, cleared with them by k_cull)
    // (and behind those: the tile-local depth sort's two lists of long tiles, each [0] = count + n_tiles entries, cleared likewise)
    HIP_TRY(c, sl.ranges.ensure((size_t)n_tiles + 1 + ((size_t)n_tiles + 1) / 2 + 1 + 2 * (((size_t)n_tiles + 2) / 2 + 1)));
    uint32_t* const d_tile_tick = reinterpret_cast<uint32_t*>(sl.ranges.p + (size_t)n_tiles + 1);
    uint32_t* const d_long_tiles = d_tile_tick + n_tiles;
    const bool dbg = c->opt_debug_varyings != 0;
    const bool need_depths = a.d_bgd != nullptr || cfg->order_mode == GSWT_ORDER_DEPTH;
    if (dbg) HIP_TRY(c, c->dbg.ensure((size_t)D.n_entries + 1));

    // The pair count P is only known on the device.  Everything downstream of k_project is launched
    // for a capacity `pair_cap` (blocks past the real P do nothing), so a frame needs no host round
    // trip; the count and an overflow flag travel back with the frame.  If P exceeded the capacity
    // the buffers grow and the frame is re-run by finish_frame (first frame / sudden scene change only).
    int key_bits = 1;
    while ((1 << key_bits) < n_tiles) key_bits++;
    if (c->pair_cap == 0) c->pair_cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(D.n_entries / 4, 1u << 20), 0xFFFFFF00ull);
    const uint32_t cap = c->pair_cap;
    sl.cap = cap;
    HIP_TRY(c, sl.keys_a.ensure_roomy((size_t)cap + 1)); HIP_TRY(c, sl.keys_b.ensure_roomy((size_t)cap + 1));
    HIP_TRY(c, sl.vals_a.ensure_roomy((size_t)cap + 1)); HIP_TRY(c, sl.vals_b.ensure_roomy((size_t)cap + 1));
    const bool depth_order = cfg->order_mode == GSWT_ORDER_DEPTH;
    const size_t n_super = (size_t)D.n_chunks / 256 + 1;
    const size_t n_super2 = (2 * (size_t)kSuperStride + 1) * n_super;     // pair sums, visible sums (a cache line per word), exclusive pair prefix (k_totals)
    // GSWT_ORDER_DEPTH: the pair list is sorted on the depth bits first, with as many 8-bit passes as the key ranges of the recent frames
    // needed (k_items flags a frame that needs more: finish_frame re-runs it); the tile ids travel as the payload of those passes
    sl.depth_passes = depth_order ? std::min<uint32_t>(std::max<uint32_t>(c->depth_passes, 1u), 4u) : 0u;
    // a re-run after the tile-local path flagged a list that does not fit takes the global passes (finish_frame raised depth_max_tile_len)
    // (GSWT_OPT_DEPTH_SORT: 0 = tile-local while the longest list fits its LDS buffer, 1 = always the global passes, 2 = as 0)
    sl.depth_local = depth_order && c->opt_depth_sort != 1;      // (no list is too long for it: k_tile_depth_sort_xl)
    const int depth_bits = sl.depth_local ? 0 : 8 * (int)sl.depth_passes;
    if (depth_order) (sl.depth_local ? c->stat_depth_local : c->stat_depth_global)++;
    if (depth_order) { HIP_TRY(c, sl.aux_a.ensure_roomy((size_t)cap + 1)); HIP_TRY(c, sl.aux_b.ensure_roomy((size_t)cap + 1)); }
    // radix workspaces: per sort the zeroed part (group rows, digit totals) in front of its per-workgroup rows (written in full)
    const size_t rz_pair = radix_ws_zero_words(cap, key_bits), rz_depth = depth_bits ? radix_ws_zero_words(cap, depth_bits) : 0;
    const size_t rw_pair = radix_ws_words(cap, key_bits), rw_depth = depth_bits ? radix_ws_words(cap, depth_bits) : 0;
    // one contiguous u32 region whose head k_cull clears: [counters: 16][super_sums: n_super2][pair sort: zeroed part .. rows][depth sort likewise]
    HIP_TRY(c, sl.ghist.ensure_roomy(16 + n_super2 + rw_pair + rw_depth + 16));
    uint32_t* const zero_a = sl.ghist.p;
    unsigned long long* const d_counters = reinterpret_cast<unsigned long long*>(zero_a);
    uint32_t* const d_super = zero_a + 16;
    uint32_t* const d_radix = d_super + n_super2;
    uint32_t* const d_radix_depth = d_radix + rw_pair;
    uint32_t* const d_krange = reinterpret_cast<uint32_t*>(d_counters + 5);     // counters[5]: (~smallest, largest) depth key of the frame (k_emit<DEPTH>)
    const uint32_t seg = (uint32_t)c->opt_segment;
    HIP_TRY(c, sl.item_base.ensure_roomy((size_t)n_tiles + 2));
    HIP_TRY(c, sl.partials.ensure_roomy(((size_t)n_tiles + cap / seg + 1) * 256));
    HIP_TRY(c, sl.item_tab.ensure_roomy((size_t)n_tiles + cap / seg + 2));
    unsigned long long* const d_P = d_counters + 1;
    hipEvent_t* ev = sl.ev;
    // ---- cull (+ clears the frame's accumulators) + project
    // the frame starts after everything submitted to the ctx stream so far (inputs produced there, earlier readers of
    // the output buffer, the device-side merged-list build and the draw bounds of its draw set), on the slot's own stream
    // (an idle ctx stream has nothing to wait for; a cross-stream event wait per frame is not free: c3 static camera 4 700 -> 4 900 frames/s)
    if (hipStreamQuery(c->stream) != hipSuccess) {
        HIP_TRY(c, hipEventRecord(sl.ev_in, c->stream));
        HIP_TRY(c, hipStreamWaitEvent(s, sl.ev_in, 0));
    }
    // ... and behind its draw set's upload / merged-list build on set_stream, unless that has long finished (the usual case: a
    // wait on another stream's event costs a barrier packet per frame)
    if (D.ev_up && D.ev_up_pending && !D.built) {
        if (hipEventQuery(D.ev_up) == hipSuccess) D.built = true;
        else HIP_TRY(c, hipStreamWaitEvent(s, D.ev_up, 0));
    }
    if (sc > 1 && out_px > 0) HIP_TRY(c, hipMemsetAsync(d_out, 0, out_px * 16, s));
    // GSWT_OPT_GRAPH: from here to the end of the frame the launch sites record instead of launching (frames that carry timing
    // events, debug varyings and shards without tiles launch as before)
    const bool use_graph = c->opt_graph != 0 && c->opt_timing == 0 && !dbg && n_tiles > 0 && sl.hc_dev != nullptr;
    struct RecorderScope {
        explicit RecorderScope(GraphRec* r) { if (r) { r->n = 0; r->overflow = false; } graph_recorder() = r; }
        ~RecorderScope() { graph_recorder() = nullptr; }
    } recorder_scope(use_graph ? &sl.grec : nullptr);
    if (c->opt_timing >= 1) HIP_TRY(c, hipEventRecord(ev[0], s));
    const uint32_t n_cells = f.band_cull ? (2u * su->map_half_wh[0] + 1u) * (2u * su->map_half_wh[1] + 1u) : 0u;
    HIP_TRY(c, sl.cell_culled.ensure((size_t)n_cells + 1));
    // cleared per frame: counters, super-group sums and the atomically accumulated part of the sort's tables.  The pair sort's zeroed
    // part is contiguous with the head; the depth sort's (behind the pair sort's rows) is the kernel's second clear range.
    const size_t n_zero_a = 16 + n_super2 + rz_pair;
    launch_cull(s, f, D.draws.p, D.n_draws, sl.draw_culled.p, sl.cell_culled.p, n_cells, zero_a, (uint32_t)n_zero_a,
                reinterpret_cast<uint32_t*>(sl.ranges.p), ((uint32_t)n_tiles + 1u) * 2u + (uint32_t)n_tiles + ((uint32_t)n_tiles + 2u), sl.block_sums.p, D.n_chunks, sl.live_cnt.p, sl.live_tab.p,
                d_radix_depth, (uint32_t)rz_depth, D.chunk_tab.p, D.n_chunks, c->static_boxes.p, c->opt_no_chunk_cull == 0 && !dbg, sl.live_cid.p);
    {
        uint64_t eff = D.n_launch;
        if (!dbg && !sl.full_grid && !c->opt_no_grid_hint && c->live_hint) {
            const uint64_t cut = 8ull * ((uint64_t)c->live_hint + c->live_hint / 2 + 256);
            if (cut * 2 <= eff) eff = cut;                     // (only where it removes most of the grid: a re-run costs a frame)
        }
        sl.n_launch_eff = (uint32_t)eff;
    }
    launch_project(s, dbg, f, D.draws.p, D.chunk_tab_xcd.p, sl.n_launch_eff, D.n_chunks, c->static_list.p, D.merged_list.p, D.merged_map.p,
                   c->tex.p, c->hmap.p, sl.draw_culled.p, sl.cell_culled.p, sl.live_cnt.p, sl.live_tab.p, sl.rects.p, sl.recs.p, need_depths ? sl.depths.p : nullptr, sl.block_sums.p, d_super,
                   d_counters, c->dbg.p, sl.col_f.p, cap, sl.strict_vs);
    if (c->opt_timing >= 2) HIP_TRY(c, hipEventRecord(ev[1], s));
    // ---- emit + sort
    // reference order: pairs in composite (slot) order, stably sorted on the tile bits.  Depth order: the same pairs with their depth bits,
    // stably sorted on the depth bits in use first (payload: the tile id), then on the tile bits: inside a tile true depth order, equal
    // depths in composite order.  The last tile pass also leaves every screen tile's [start, end) of the sorted list in sl.ranges (zeroed by k_cull).
    const uint32_t* vals_sorted = nullptr;
    if (!depth_order) {
        launch_emit(s, f, D.n_chunks, sl.rects.p, sl.block_sums.p, d_super, cap, d_counters, sl.keys_a.p, sl.vals_a.p, nullptr, nullptr, nullptr, dbg ? nullptr : sl.live_cnt.p, sl.live_cid.p, sl.n_launch_eff);
        if (c->opt_timing >= 2) HIP_TRY(c, hipEventRecord(ev[3], s));
        const int where = launch_sort(s, sl.keys_a.p, sl.vals_a.p, sl.keys_b.p, sl.vals_b.p, cap, d_P, key_bits, d_radix, sl.ranges.p);
        vals_sorted = where ? sl.vals_b.p : sl.vals_a.p;
    } else if (sl.depth_local) {
        // tile-local path: tile ids are the sort key, the depth bits its payload; then every tile's slice is depth-sorted in LDS
        launch_emit(s, f, D.n_chunks, sl.rects.p, sl.block_sums.p, d_super, cap, d_counters, sl.keys_a.p, sl.vals_a.p, sl.depths.p, sl.aux_a.p, nullptr, dbg ? nullptr : sl.live_cnt.p, sl.live_cid.p, sl.n_launch_eff);
        if (c->opt_timing >= 2) HIP_TRY(c, hipEventRecord(ev[3], s));
        const int where = launch_sort(s, sl.keys_a.p, sl.vals_a.p, sl.keys_b.p, sl.vals_b.p, cap, d_P, key_bits, d_radix, sl.ranges.p, nullptr, sl.aux_a.p, sl.aux_b.p);
        uint32_t* const vals_t = where ? sl.vals_b.p : sl.vals_a.p;
        launch_tile_depth_sort(s, sl.ranges.p, vals_t, where ? sl.aux_b.p : sl.aux_a.p, where ? sl.vals_a.p : sl.vals_b.p, where ? sl.aux_a.p : sl.aux_b.p, n_tiles,
                               d_long_tiles, d_counters);
        vals_sorted = vals_t;
    } else {
        launch_emit(s, f, D.n_chunks, sl.rects.p, sl.block_sums.p, d_super, cap, d_counters, sl.aux_a.p, sl.vals_a.p, sl.depths.p, sl.keys_a.p, d_krange, dbg ? nullptr : sl.live_cnt.p, sl.live_cid.p, sl.n_launch_eff);
        if (c->opt_timing >= 2) HIP_TRY(c, hipEventRecord(ev[3], s));
        const int wd = launch_sort(s, sl.keys_a.p, sl.vals_a.p, sl.keys_b.p, sl.vals_b.p, cap, d_P, depth_bits, d_radix_depth, nullptr, d_krange, sl.aux_a.p, sl.aux_b.p);
        // (the depth keys are dead now: keys_a serves as the other half of the tile-key ping-pong)
        uint32_t* const tiles_in = wd ? sl.aux_b.p : sl.aux_a.p;
        uint32_t* const vals_in = wd ? sl.vals_b.p : sl.vals_a.p;
        uint32_t* const vals_other = wd ? sl.vals_a.p : sl.vals_b.p;
        const int where = launch_sort(s, tiles_in, vals_in, sl.keys_a.p, vals_other, cap, d_P, key_bits, d_radix, sl.ranges.p);
        vals_sorted = where ? vals_other : vals_in;
    }
    if (c->opt_timing >= 2) HIP_TRY(c, hipEventRecord(ev[4], s));
    // ---- ranges
    if (c->opt_timing >= 2) HIP_TRY(c, hipEventRecord(ev[5], s));
    // ---- composite
    launch_composite(s, f, sl.ranges.p, vals_sorted, sl.recs.p, sl.depths.p, sl.col_f.p, a.d_bg, a.d_bgd, d_out, n_tiles, out_rows, seg, cap,
                     sl.item_base.p, sl.item_tab.p, sl.partials.p, c->opt_timing >= 1 ? ev[7] : nullptr, c->opt_timing >= 1 ? ev[8] : nullptr,
                     d_counters, sl.hc_dev, c->opt_composite, depth_order && !sl.depth_local ? d_krange : nullptr, sl.depth_passes, d_tile_tick, depth_order,
                     c->opt_item_order != 0);
    c->last_n_tiles = (uint32_t)n_tiles;
    c->last_slot = (int)(&sl - c->slots);
    if (c->opt_timing >= 1) HIP_TRY(c, hipEventRecord(ev[6], s));
    if (use_graph) {
        graph_recorder() = nullptr;
        const int grc = replay_graph(c, sl);
        if (grc != GSWT_OK) return grc;
    }
    HIP_TRY(c, hipGetLastError());
    // k_combine (the frame's last kernel) stores the result counters into the pinned host words itself; only a frame
    // without screen tiles has no such launch
    if (n_tiles == 0 || !sl.hc_dev) HIP_TRY(c, hipMemcpyAsync(sl.hc, d_counters, 5 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipEventRecord(ev[9], s));
    sl.timing_level = c->opt_timing;
    return GSWT_OK;
}

// Waits for the slot's frame; re-runs it with larger pair buffers if it overflowed; fills c->timings.
static int finish_frame(gswt_ctx* c, FrameSlot& sl)
{
    for (int attempt = 0;; attempt++) {
        HIP_TRY(c, hipEventSynchronize(sl.ev[9]));
        const unsigned long long P64 = sl.hc[1];
        if (P64 >= 0xFFFFFF00ull) return fail(c, GSWT_ERR_CAPACITY, "gswt_render: %llu pairs exceed 2^32", P64);
        if (sl.hc[3] == 0 && P64 <= sl.cap) break;
        if (attempt >= 3) return fail(c, GSWT_ERR_CAPACITY, "gswt_render: pair capacity did not converge");      // (a short launch grid, a pair overflow and a depth-pass shortfall can each cost one re-run)
        if (P64 > sl.cap || sl.args.cfg.order_mode != GSWT_ORDER_DEPTH)
            c->pair_cap = std::max<uint32_t>(c->pair_cap, (uint32_t)std::min<uint64_t>(P64 + P64 / 2 + 4096, 0xFFFFFF00ull));
        if (sl.args.cfg.order_mode == GSWT_ORDER_DEPTH) {
            const uint32_t need = (uint32_t)(sl.hc[2] & 0xFFFFFFFFull), max_len = (uint32_t)(sl.hc[2] >> 32);
            if (!sl.depth_local && need > sl.depth_passes) { c->depth_passes = std::min<uint32_t>(need, 4u); c->depth_passes_low_run = 0; }
            // the tile-local depth sort met a list longer than its LDS buffer: the re-run takes the global passes (k_items may not have
            // seen the whole frame's lengths yet when the flag was raised, so at least cap + 1)
            if (sl.depth_local) c->depth_max_tile_len = std::max<uint32_t>(std::max<uint32_t>(max_len, c->depth_max_tile_len), tile_depth_sort_cap() + 1u);
        }
        sl.full_grid = true;                                   // (whatever flagged it: the re-run covers the whole launch table)
        c->live_hint = std::max<uint32_t>(c->live_hint, (uint32_t)std::min<unsigned long long>(sl.hc[4], 0xFFFFFFFFull));
        int rc = enqueue_frame(c, sl);
        sl.full_grid = false;
        if (rc != GSWT_OK) return rc;
    }
    c->live_hint = (uint32_t)std::min<unsigned long long>(sl.hc[4], 0xFFFFFFFFull);
    const uint32_t P = (uint32_t)sl.hc[1];
    // keep 25-50 % headroom over the running pair count without shrinking on every small dip
    if (!c->opt_fixed_pair_cap && (uint64_t)P + P / 4 > c->pair_cap) c->pair_cap = (uint32_t)std::min<uint64_t>((uint64_t)P + P / 2 + 4096, 0xFFFFFF00ull);
    if (sl.args.cfg.order_mode == GSWT_ORDER_DEPTH) {
        const uint32_t need = std::min<uint32_t>(std::max<uint32_t>((uint32_t)(sl.hc[2] & 0xFFFFFFFFull), 1u), 4u);
        c->depth_max_tile_len = (uint32_t)(sl.hc[2] >> 32);        // (every depth-ordered frame reports it: the next one picks its path by it)
        if (sl.depth_local) { /* the pass count is the kernel's own business there */ }
        else if (need < c->depth_passes && sl.depth_passes == c->depth_passes) {
            c->depth_passes_low_max = c->depth_passes_low_run ? std::max(c->depth_passes_low_max, need) : need;
            if (++c->depth_passes_low_run >= 32u) { c->depth_passes = c->depth_passes_low_max; c->depth_passes_low_run = 0; }
        } else if (need >= c->depth_passes) c->depth_passes_low_run = 0;
    }
    gswt_timings& t = c->timings;
    memset(&t, 0, sizeof(t));
    hipEvent_t* ev = sl.ev;
    if (sl.timing_level >= 2) {
        HIP_TRY(c, hipEventElapsedTime(&t.ms_project, ev[0], ev[1]));
        HIP_TRY(c, hipEventElapsedTime(&t.ms_emit, ev[1], ev[3]));
        HIP_TRY(c, hipEventElapsedTime(&t.ms_sort, ev[3], ev[4]));
        HIP_TRY(c, hipEventElapsedTime(&t.ms_ranges, ev[4], ev[5]));
        HIP_TRY(c, hipEventElapsedTime(&t.ms_composite, ev[5], ev[6]));
    }
    if (sl.timing_level >= 1) {
        HIP_TRY(c, hipEventElapsedTime(&t.ms_total, ev[0], ev[6]));
        HIP_TRY(c, hipEventElapsedTime(&t.ms_composite_kernel, ev[7], ev[8]));
    }
    t.n_draws = c->sets[sl.set].n_draws; t.n_instanced = c->sets[sl.set].n_entries; t.n_visible = sl.hc[0]; t.n_pairs = P; t.n_tiles = (uint32_t)sl.n_tiles;
    return GSWT_OK;
}

static void fill_args(FrameArgs& a, const gswt_camera_uniforms* cam, const gswt_scene_uniforms* su, const gswt_render_config* cfg,
                      int width, int height, const float4* d_bg, const float* d_bgd, float4* d_out)
{
    a.cam = *cam; a.su = *su; a.cfg = *cfg; a.width = width; a.height = height; a.d_bg = d_bg; a.d_bgd = d_bgd; a.d_out = d_out;
}

int gswt_render(gswt_ctx* c, const gswt_camera_uniforms* cam, const gswt_scene_uniforms* su, const gswt_render_config* cfg,
                int width, int height, const float* bg_rgba, const float* bg_depth, int bg_on_device, float* out_rgba,
                int out_on_device)
try {
    if (!c) return GSWT_ERR_BAD_ARG;
    int rc = validate_frame(c, cam, su, cfg, width, height, out_rgba);
    if (rc != GSWT_OK) return rc;
    hipSetDevice(c->device);
    hipStream_t s = c->stream;
    HIP_TRY(c, collect_pending(c));                // frames still in flight from the async API keep their tickets
    const int sc = cfg->shard_count <= 1 ? 1 : cfg->shard_count;
    const bool cols = sc > 1 && cfg->shard_mode == GSWT_SHARD_COLUMNS;
    const int out_rows = sc > 1 && !cols ? gswt_shard_rows_padded(height, sc) : height;
    const size_t out_px = (size_t)out_rows * (cols ? gswt_shard_cols_padded(width, sc) : width), npx = (size_t)width * height;
    const float4* d_bg = nullptr; const float* d_bgd = nullptr; float4* d_out = nullptr;
    if (bg_rgba) {
        if (bg_on_device) d_bg = reinterpret_cast<const float4*>(bg_rgba);
        else { HIP_TRY(c, c->bg_rgba.ensure(npx)); HIP_TRY(c, hipMemcpyAsync(c->bg_rgba.p, bg_rgba, npx * 16, hipMemcpyHostToDevice, s)); d_bg = c->bg_rgba.p; }
    }
    if (bg_depth) {
        if (bg_on_device) d_bgd = bg_depth;
        else { HIP_TRY(c, c->bg_depth.ensure(npx)); HIP_TRY(c, hipMemcpyAsync(c->bg_depth.p, bg_depth, npx * 4, hipMemcpyHostToDevice, s)); d_bgd = c->bg_depth.p; }
    }
    if (out_on_device) d_out = reinterpret_cast<float4*>(out_rgba);
    else { HIP_TRY(c, c->out_img.ensure(out_px)); d_out = c->out_img.p; }
    int si0 = 0;
    for (int k = 0; k < kFrameSlots; k++) if (!c->slots[k].pending) { si0 = k; break; }
    FrameSlot& sl = c->slots[si0];
    if (sl.pending) return fail(c, GSWT_ERR_STATE, "gswt_render: every frame slot holds an uncollected gswt_render_async ticket");
    if (!sl.stream) HIP_TRY(c, hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
    activate_pending(c, false);
    sl.set = c->cur_set;
    fill_args(sl.args, cam, su, cfg, width, height, d_bg, d_bgd, d_out);
    sl.strict_vs = c->opt_strict_vs != 0;
    rc = enqueue_frame(c, sl);
    if (rc != GSWT_OK) return rc;
    rc = finish_frame(c, sl);
    if (rc != GSWT_OK) return rc;
    if (!out_on_device) {
        HIP_TRY(c, hipMemcpyAsync(out_rgba, d_out, out_px * 16, hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
    }
    return GSWT_OK;
} GSWT_CATCH("gswt_render")

int gswt_render_async(gswt_ctx* c, const gswt_camera_uniforms* cam, const gswt_scene_uniforms* su, const gswt_render_config* cfg,
                      int width, int height, const float* bg_rgba_dev, const float* bg_depth_dev, float* out_rgba_dev, int* ticket)
try {
    if (!c || !ticket) return GSWT_ERR_BAD_ARG;
    int rc = validate_frame(c, cam, su, cfg, width, height, out_rgba_dev);
    if (rc != GSWT_OK) return rc;
    hipSetDevice(c->device);
    // lowest free slot (a caller that keeps fewer frames in flight than there are slots then cycles over fewer buffer sets:
    // the per-frame buffers of a c5-sized frame are ~5 GB per slot); all busy: the oldest frame is collected first
    int si = -1;
    for (int k = 0; k < kFrameSlots; k++) if (!c->slots[k].pending) { si = k; break; }
    if (si < 0) {
        si = 0;
        for (int k = 1; k < kFrameSlots; k++) if (c->slots[k].seq < c->slots[si].seq) si = k;
    }
    FrameSlot& sl = c->slots[si];
    if (!sl.stream) HIP_TRY(c, hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));      // a slot past the fourth: first use
    if (sl.pending) {                   // every slot in flight: the oldest frame is collected here and its slot reused
        sl.pending = false;
        if (sl.collected) { sl.collected = false; rc = sl.collected_rc; } else rc = finish_frame(c, sl);
        if (rc != GSWT_OK) return rc;
    }
    sl.seq = ++c->frame_seq;
    activate_pending(c, false);
    sl.set = c->cur_set;
    sl.strict_vs = c->opt_strict_vs != 0;
    sl.gather_recorded = false;
    fill_args(sl.args, cam, su, cfg, width, height, reinterpret_cast<const float4*>(bg_rgba_dev), bg_depth_dev,
              reinterpret_cast<float4*>(out_rgba_dev));
    rc = enqueue_frame(c, sl);
    if (rc != GSWT_OK) return rc;
    sl.pending = true; sl.collected = false;
    *ticket = si;
    return GSWT_OK;
} GSWT_CATCH("gswt_render_async")

int gswt_frame_slots(void) { return kFrameSlots; }

int gswt_render_wait(gswt_ctx* c, int ticket)
try {
    if (!c || ticket < 0 || ticket >= kFrameSlots) return GSWT_ERR_BAD_ARG;
    FrameSlot& sl = c->slots[ticket];
    if (!sl.pending) return fail(c, GSWT_ERR_STATE, "gswt_render_wait: ticket %d is not in flight", ticket);
    hipSetDevice(c->device);
    sl.pending = false;
    if (sl.collected) {                 // already finished by gswt_render_fence / a state change: hand back what it left
        sl.collected = false;
        c->timings = sl.collected_timings;
        return sl.collected_rc;
    }
    return finish_frame(c, sl);
} GSWT_CATCH("gswt_render_wait")

// All-or-nothing, like GSWTRenderer::render (renderer.rs:407-414): work ordered behind the fence never reads a frame whose pair
// buffers overflowed.  Overflow is only known once the frame's counters are back on the host, so the fence first waits
// (host side) for THIS frame -- the younger frames in flight keep the GPU busy meanwhile --, lets finish_frame re-run it
// with grown buffers if it has to, and only then orders the ctx stream behind the frame's final completion event.
int gswt_render_fence(gswt_ctx* c, int ticket)
try {
    if (!c || ticket < 0 || ticket >= kFrameSlots) return GSWT_ERR_BAD_ARG;
    FrameSlot& sl = c->slots[ticket];
    if (!sl.pending) return fail(c, GSWT_ERR_STATE, "gswt_render_fence: ticket %d is not in flight", ticket);
    hipSetDevice(c->device);
    if (!sl.collected) {
        sl.collected_rc = finish_frame(c, sl);
        sl.collected_timings = c->timings;
        sl.collected = true;
    }
    if (sl.collected_rc != GSWT_OK) return sl.collected_rc;
    HIP_TRY(c, hipStreamWaitEvent(c->stream, sl.ev[9], 0));
    return GSWT_OK;
} GSWT_CATCH("gswt_render_fence")

int gswt_skybox_configure(gswt_ctx* c, const float* faces_rgba, int face_size, int equirectangular)
try {
    if (!c) return GSWT_ERR_BAD_ARG;
    if (!faces_rgba || face_size <= 0 || face_size > 16384) return fail(c, GSWT_ERR_BAD_ARG, "gswt_skybox_configure: bad cube map");
    hipSetDevice(c->device);
    HIP_TRY(c, sync_all(c));
    const size_t n = (size_t)6 * face_size * face_size;
    HIP_TRY(c, c->sky_faces.ensure(n));
    HIP_TRY(c, hipMemcpy(c->sky_faces.p, faces_rgba, n * 16, hipMemcpyHostToDevice));
    HIP_TRY(c, null_stream_done());
    c->sky_size = face_size; c->sky_equi = equirectangular ? 1 : 0;
    return GSWT_OK;
} GSWT_CATCH("gswt_skybox_configure")

int gswt_skybox_render(gswt_ctx* c, const gswt_camera_uniforms* cam, int width, int height, float* out_rgba_dev)
try {
    if (!c) return GSWT_ERR_BAD_ARG;
    if (!cam || !out_rgba_dev || width <= 0 || height <= 0) return fail(c, GSWT_ERR_BAD_ARG, "gswt_skybox_render: bad argument");
    if (c->sky_size == 0) return fail(c, GSWT_ERR_STATE, "gswt_skybox_render before gswt_skybox_configure");
    if (cam->projection[0] == 0.0f || cam->projection[5] == 0.0f) return fail(c, GSWT_ERR_BAD_ARG, "gswt_skybox_render: singular projection");
    hipSetDevice(c->device);
    launch_skybox(c->stream, cam->view, cam->projection[0], cam->projection[5], width, height, c->sky_size, c->sky_equi, c->sky_faces.p,
                  reinterpret_cast<float4*>(out_rgba_dev));
    HIP_TRY(c, hipGetLastError());
    return GSWT_OK;
} GSWT_CATCH("gswt_skybox_render")

int gswt_proxy_configure(gswt_ctx* c, const float* const* mips, int tex_size, int n_mips, int grid_dim)
try {
    if (!c) return GSWT_ERR_BAD_ARG;
    if (!mips || tex_size <= 0 || n_mips <= 0 || n_mips > 16 || (tex_size >> (n_mips - 1)) < 1 || grid_dim <= 0 || grid_dim > 32768)
        return fail(c, GSWT_ERR_BAD_ARG, "gswt_proxy_configure: bad mip chain / grid");
    hipSetDevice(c->device);
    HIP_TRY(c, sync_all(c));
    size_t total = 0;
    for (int l = 0; l < n_mips; l++) {
        if (!mips[l]) return fail(c, GSWT_ERR_BAD_ARG, "gswt_proxy_configure: mip %d is null", l);
        c->proxy_mip_off[l] = (uint32_t)total;
        total += (size_t)(tex_size >> l) * (tex_size >> l);
    }
    if (total >= 0xFFFFFFFFull) return fail(c, GSWT_ERR_CAPACITY, "gswt_proxy_configure: texture too large");
    HIP_TRY(c, c->proxy_tex.ensure(total));
    for (int l = 0; l < n_mips; l++)
        HIP_TRY(c, hipMemcpy(c->proxy_tex.p + c->proxy_mip_off[l], mips[l], (size_t)(tex_size >> l) * (tex_size >> l) * 16, hipMemcpyHostToDevice));
    HIP_TRY(c, null_stream_done());
    c->proxy_size = tex_size; c->proxy_mips = n_mips; c->proxy_grid_dim = grid_dim;
    return GSWT_OK;
} GSWT_CATCH("gswt_proxy_configure")

int gswt_proxy_render(gswt_ctx* c, const gswt_proxy_uniforms* u, int width, int height, float* rgba_dev, float* depth_dev, int clear_depth)
try {
    if (!c) return GSWT_ERR_BAD_ARG;
    if (!u || !rgba_dev || !depth_dev || width <= 0 || height <= 0) return fail(c, GSWT_ERR_BAD_ARG, "gswt_proxy_render: bad argument");
    if (c->proxy_size == 0 && !u->black_background) return fail(c, GSWT_ERR_STATE, "gswt_proxy_render before gswt_proxy_configure");
    if (u->surface_type == 1u && (c->hm_w == 0 || c->hm_h == 0)) return fail(c, GSWT_ERR_STATE, "gswt_proxy_render: HeightMap surface without gswt_configure height map");
    if (!(u->tile_width > 0.0f) || (!u->map_proxy && !(u->width_scale > 0.0f))) return fail(c, GSWT_ERR_BAD_ARG, "gswt_proxy_render: cell size must be positive");
    if (u->projection[0] == 0.0f || u->projection[5] == 0.0f) return fail(c, GSWT_ERR_BAD_ARG, "gswt_proxy_render: singular projection");
    hipSetDevice(c->device);
    ProxyArgs a;
    memset(&a, 0, sizeof(a));
    a.height_offset = u->height_offset; a.tile_width = u->tile_width; a.width_scale = u->width_scale; a.clip_height = u->clip_height;
    a.brightness = u->brightness; a.surface_type = u->surface_type; a.use_clip = u->use_clip; a.black_background = u->black_background;
    memcpy(a.V, u->view, 64);
    for (int cc = 0; cc < 4; cc++) {          // opengl_to_wgpu * projection, proxy.wgsl:84-91
        const float* P = u->projection;
        a.GP[4 * cc + 0] = P[4 * cc + 0]; a.GP[4 * cc + 1] = P[4 * cc + 1];
        a.GP[4 * cc + 2] = 0.5f * P[4 * cc + 2] + 0.5f * P[4 * cc + 3]; a.GP[4 * cc + 3] = P[4 * cc + 3];
    }
    a.p00 = u->projection[0]; a.p11 = u->projection[5];
    for (int k = 0; k < 3; k++) { a.cam[k] = u->cam_pos[k]; a.height_map_scale[k] = u->height_map_scale[k]; }
    a.map_half_wh[0] = u->map_half_wh[0]; a.map_half_wh[1] = u->map_half_wh[1];
    const float tw = u->tile_width;
    if (u->map_proxy == 1u) {                 // proxy.rs:219-251 + proxy.wgsl:51
        a.nx = 2 * (int)u->map_half_wh[0] + 1; a.ny = 2 * (int)u->map_half_wh[1] + 1; a.cs = tw;
        a.gx0 = (float)(-(int)u->map_half_wh[0]) * tw + (float)u->center_coord[0] * tw;
        a.gy0 = (float)(-(int)u->map_half_wh[1]) * tw + (float)u->center_coord[1] * tw;
    } else {                                  // proxy.rs:136-163 + proxy.wgsl:66-68
        const int g = c->proxy_grid_dim;
        a.nx = a.ny = g; a.cs = u->width_scale;
        a.gx0 = (float)(-(g / 2)) * u->width_scale + floorf((float)u->center_coord[0] * tw / u->width_scale) * u->width_scale;
        a.gy0 = (float)(-(g / 2)) * u->width_scale + floorf((float)u->center_coord[1] * tw / u->width_scale) * u->width_scale;
    }
    a.hm_w = c->hm_w; a.hm_h = c->hm_h; a.tex_size = c->proxy_size > 0 ? c->proxy_size : 1; a.n_mips = c->proxy_mips > 0 ? c->proxy_mips : 1;
    memcpy(a.mip_off, c->proxy_mip_off, sizeof(a.mip_off));
    a.width = width; a.height = height;
    if (clear_depth) launch_fill_f32(c->stream, depth_dev, (size_t)width * height, 1.0f);
    launch_proxy(c->stream, a, c->hmap.p, c->proxy_tex.p, reinterpret_cast<float4*>(rgba_dev), depth_dev);
    HIP_TRY(c, hipGetLastError());
    return GSWT_OK;
} GSWT_CATCH("gswt_proxy_render")

int gswt_unshard(gswt_ctx* c, const float* gathered, int width, int height, int shard_count, float* out_rgba)
try {
    return gswt_unshard_mode(c, gathered, width, height, shard_count, GSWT_SHARD_ROWS, out_rgba);
} GSWT_CATCH("gswt_unshard")

int gswt_unshard_mode(gswt_ctx* c, const float* gathered, int width, int height, int shard_count, int shard_mode, float* out_rgba)
try {
    if (!c || !gathered || !out_rgba || width <= 0 || height <= 0 || shard_count < 1) return GSWT_ERR_BAD_ARG;
    if (shard_mode != GSWT_SHARD_ROWS && shard_mode != GSWT_SHARD_COLUMNS) return GSWT_ERR_BAD_ARG;
    hipSetDevice(c->device);
    launch_unshard(c->stream, reinterpret_cast<const float4*>(gathered), reinterpret_cast<float4*>(out_rgba), width, height,
                   shard_count, gswt_shard_rows_padded(height, shard_count),
                   shard_mode == GSWT_SHARD_COLUMNS ? gswt_shard_cols_padded(width, shard_count) : 0);
    HIP_TRY(c, hipGetLastError());
    return GSWT_OK;
} GSWT_CATCH("gswt_unshard_mode")

// ---- multi-GPU gather ------------------------------------------------------------------------------------------------
int gswt_comm_unique_id(void* id_out)
try {
    if (!id_out) return GSWT_ERR_BAD_ARG;
    if (rccl_load()) return GSWT_ERR_RCCL;
    return g_rccl.GetUniqueId(id_out) == 0 ? GSWT_OK : GSWT_ERR_RCCL;
} catch (...) { return GSWT_ERR_RCCL; }

int gswt_comm_init(gswt_ctx* c, const void* unique_id, int rank, int world)
try {
    if (!c || !unique_id || world < 1 || rank < 0 || rank >= world) return fail(c, GSWT_ERR_BAD_ARG, "gswt_comm_init: bad rank / world");
    if (c->comm || !c->group.empty()) return fail(c, GSWT_ERR_STATE, "gswt_comm_init: the ctx already has a communicator (gswt_comm_destroy first)");
    if (const char* e = rccl_load()) return fail(c, GSWT_ERR_RCCL, "gswt_comm_init: %s", e);
    hipSetDevice(c->device);
    Id128 id;
    memcpy(id.b, unique_id, GSWT_COMM_ID_BYTES);
    void* comm = nullptr;
    const int rc = g_rccl.CommInitRank(&comm, world, id, rank);
    if (rc != 0) return fail(c, GSWT_ERR_RCCL, "ncclCommInitRank(rank %d of %d): %s", rank, world, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error");
    c->comm = comm; c->comm_rank = rank; c->comm_world = world;
    return GSWT_OK;
} GSWT_CATCH("gswt_comm_init")

int gswt_comm_destroy(gswt_ctx* c)
try {
    if (!c) return GSWT_ERR_BAD_ARG;
    hipSetDevice(c->device);
    HIP_TRY(c, collect_pending(c));
    if (c->comm) { g_rccl.CommDestroy(c->comm); c->comm = nullptr; c->comm_world = 0; }
    // a peer-copy group is dissolved as a whole: no member keeps a pointer to a context that may be destroyed next
    const std::vector<gswt_ctx*> members = c->group;
    for (gswt_ctx* m : members)
        if (m && m != c) { hipSetDevice(m->device); collect_pending(m); m->group.clear(); }
    c->group.clear();
    hipSetDevice(c->device);
    return GSWT_OK;
} GSWT_CATCH("gswt_comm_destroy")

int gswt_group_init(gswt_ctx* const* ctxs, int n)
try {
    if (!ctxs || n < 1) return GSWT_ERR_BAD_ARG;
    for (int r = 0; r < n; r++) {
        if (!ctxs[r]) return GSWT_ERR_BAD_ARG;
        if (ctxs[r]->comm || !ctxs[r]->group.empty()) return fail(ctxs[r], GSWT_ERR_STATE, "gswt_group_init: rank %d already has a communicator", r);
    }
    for (int r = 0; r < n; r++) {
        gswt_ctx* c = ctxs[r];
        hipSetDevice(c->device);
        for (int p = 0; p < n; p++)
            if (ctxs[p]->device != c->device) {
                int can = 0;
                HIP_TRY(c, hipDeviceCanAccessPeer(&can, c->device, ctxs[p]->device));
                if (can) { hipError_t e = hipDeviceEnablePeerAccess(ctxs[p]->device, 0); if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) HIP_TRY(c, e); (void)hipGetLastError(); }
            }
        if (!c->ev_push) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_push, hipEventDisableTiming));
        if (!c->ev_unshard) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_unshard, hipEventDisableTiming));
        c->group.assign(ctxs, ctxs + n);
        c->group_rank = r;
    }
    return GSWT_OK;
} catch (...) { return GSWT_ERR_HIP; }

int gswt_render_gather(gswt_ctx* c, int ticket, float* frame_out_dev)
try {
    if (!c || ticket < 0 || ticket >= kFrameSlots || !frame_out_dev) return GSWT_ERR_BAD_ARG;
    if (!c->comm) return fail(c, GSWT_ERR_STATE, "gswt_render_gather before gswt_comm_init");
    int rc = gswt_render_fence(c, ticket);               // overflow-safe: the gathered shard is complete
    if (rc != GSWT_OK) return rc;
    FrameSlot& sl = c->slots[ticket];
    const ShardGeom g = shard_geom(sl);
    if (g.world != c->comm_world || sl.args.cfg.shard_index != c->comm_rank)
        return fail(c, GSWT_ERR_BAD_ARG, "gswt_render_gather: the frame was rendered as shard %d of %d, the communicator is rank %d of %d",
                    sl.args.cfg.shard_index, g.world, c->comm_rank, c->comm_world);
    hipSetDevice(c->device);
    if (g.world == 1) {                                   // nothing to gather: the shard is the frame
        if (reinterpret_cast<float4*>(frame_out_dev) != sl.args.d_out)
            HIP_TRY(c, hipMemcpyAsync(frame_out_dev, sl.args.d_out, g.px * 16, hipMemcpyDeviceToDevice, c->stream));
        return GSWT_OK;
    }
    HIP_TRY(c, c->gather_buf.ensure((size_t)g.world * g.px));
    const int nrc = g_rccl.AllGather(sl.args.d_out, c->gather_buf.p, g.px * 4, kNcclFloat, c->comm, c->stream);
    if (nrc != 0) return fail(c, GSWT_ERR_RCCL, "ncclAllGather: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(nrc) : "error");
    launch_unshard(c->stream, c->gather_buf.p, reinterpret_cast<float4*>(frame_out_dev), sl.args.width, sl.args.height, g.world,
                   gswt_shard_rows_padded(sl.args.height, g.world), g.mode == GSWT_SHARD_COLUMNS ? g.out_w : 0);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(sl.ev_gather, c->stream));
    sl.gather_recorded = true;
    return GSWT_OK;
} GSWT_CATCH("gswt_render_gather")

int gswt_group_render_gather(gswt_ctx* const* ctxs, const int* tickets, float* const* frames_out_dev, int n)
try {
    if (!ctxs || !tickets || !frames_out_dev || n < 1) return GSWT_ERR_BAD_ARG;
    for (int r = 0; r < n; r++)
        if (!ctxs[r] || (int)ctxs[r]->group.size() != n || ctxs[r]->group[r] != ctxs[r] || !frames_out_dev[r] || tickets[r] < 0 || tickets[r] >= kFrameSlots)
            return GSWT_ERR_BAD_ARG;
    // 1. every rank's frame is complete (overflow-safe fence) and the ctx streams are ordered behind them
    for (int r = 0; r < n; r++) { int rc = gswt_render_fence(ctxs[r], tickets[r]); if (rc != GSWT_OK) return rc; }
    ShardGeom g0 = shard_geom(ctxs[0]->slots[tickets[0]]);
    for (int r = 0; r < n; r++) {
        gswt_ctx* c = ctxs[r];
        const FrameSlot& sl = c->slots[tickets[r]];
        const ShardGeom g = shard_geom(sl);
        if (g.world != n || sl.args.cfg.shard_index != r || g.mode != g0.mode || g.px != g0.px || sl.args.width != ctxs[0]->slots[tickets[0]].args.width)
            return fail(c, GSWT_ERR_BAD_ARG, "gswt_group_render_gather: rank %d rendered shard %d of %d", r, sl.args.cfg.shard_index, g.world);
        hipSetDevice(c->device);
        HIP_TRY(c, c->gather_buf.ensure((size_t)n * g.px));
    }
    // 2. push: rank r copies its shard into slot r of every peer's gather buffer (xGMI peer copies; a plain copy on one device).
    // A peer's gather buffer may still be read by the re-assembly of the PREVIOUS gather on the peer's own stream (gathers are
    // issued back to back with frames in flight): the pushing stream first waits for that re-assembly (write-after-read).
    for (int r = 0; r < n; r++) {
        gswt_ctx* c = ctxs[r];
        hipSetDevice(c->device);
        const FrameSlot& sl = c->slots[tickets[r]];
        for (int p = 0; p < n; p++)
            if (p != r && ctxs[p]->unshard_pending) HIP_TRY(c, hipStreamWaitEvent(c->stream, ctxs[p]->ev_unshard, 0));
        for (int p = 0; p < n; p++)
            HIP_TRY(c, hipMemcpyPeerAsync(ctxs[p]->gather_buf.p + (size_t)r * g0.px, ctxs[p]->device, sl.args.d_out, c->device, g0.px * 16, c->stream));
        HIP_TRY(c, hipEventRecord(c->ev_push, c->stream));
    }
    // 3. every rank waits (on the device) for all pushes, then re-assembles the frame
    for (int p = 0; p < n; p++) {
        gswt_ctx* c = ctxs[p];
        hipSetDevice(c->device);
        const FrameSlot& sl = c->slots[tickets[p]];
        for (int r = 0; r < n; r++) HIP_TRY(c, hipStreamWaitEvent(c->stream, ctxs[r]->ev_push, 0));
        if (n == 1) {
            if (reinterpret_cast<float4*>(frames_out_dev[p]) != sl.args.d_out)
                HIP_TRY(c, hipMemcpyAsync(frames_out_dev[p], c->gather_buf.p, g0.px * 16, hipMemcpyDeviceToDevice, c->stream));
        } else {
            launch_unshard(c->stream, c->gather_buf.p, reinterpret_cast<float4*>(frames_out_dev[p]), sl.args.width, sl.args.height, n,
                           gswt_shard_rows_padded(sl.args.height, n), g0.mode == GSWT_SHARD_COLUMNS ? g0.out_w : 0);
            HIP_TRY(c, hipGetLastError());
        }
        HIP_TRY(c, hipEventRecord(c->ev_unshard, c->stream));
        c->unshard_pending = true;
        FrameSlot& slw = c->slots[tickets[p]];
        HIP_TRY(c, hipEventRecord(slw.ev_gather, c->stream));
        slw.gather_recorded = true;
    }
    return GSWT_OK;
} catch (...) { return GSWT_ERR_HIP; }

int gswt_synchronize(gswt_ctx* c)
try {
    if (!c) return GSWT_ERR_BAD_ARG;
    hipSetDevice(c->device);
    HIP_TRY(c, sync_all(c));
    return GSWT_OK;
} GSWT_CATCH("gswt_synchronize")

int gswt_last_timings(const gswt_ctx* c, gswt_timings* out)
try {
    if (!c || !out) return GSWT_ERR_BAD_ARG;
    *out = c->timings;
    return GSWT_OK;
} GSWT_CATCH("gswt_last_timings")

int gswt_debug_totals(gswt_ctx* c, const uint32_t* pair_sums, const uint32_t* visible_sums, uint32_t n_super, uint32_t pair_cap,
                      unsigned long long counters_out[4], uint32_t* super_excl_out)
try {
    if (!c || !pair_sums || !visible_sums || n_super == 0 || !counters_out) return GSWT_ERR_BAD_ARG;
    hipSetDevice(c->device);
    DevBuf<uint32_t> buf;
    const size_t S = kSuperStride, words = (2 * S + 1) * (size_t)n_super;
    HIP_TRY(c, buf.ensure(words + 16));
    unsigned long long* d_cnt = reinterpret_cast<unsigned long long*>(buf.p + words + (words & 1));
    HIP_TRY(c, hipMemset(buf.p, 0, (words + 16) * 4));
    HIP_TRY(c, hipMemcpy2D(buf.p, S * 4, pair_sums, 4, 4, n_super, hipMemcpyHostToDevice));                 // one word per cache line
    HIP_TRY(c, hipMemcpy2D(buf.p + S * n_super, S * 4, visible_sums, 4, 4, n_super, hipMemcpyHostToDevice));
    launch_totals(c->stream, buf.p, n_super, d_cnt, pair_cap);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(counters_out, d_cnt, 32, hipMemcpyDeviceToHost));
    if (super_excl_out) HIP_TRY(c, hipMemcpy(super_excl_out, buf.p + 2 * S * (size_t)n_super, (size_t)n_super * 4, hipMemcpyDeviceToHost));
    buf.release();
    return GSWT_OK;
} GSWT_CATCH("gswt_debug_totals")

int gswt_debug_sort(gswt_ctx* c, uint32_t* keys, uint32_t* vals, size_t n, int key_bits)
try {
    if (!c || !keys || !vals || n == 0 || n >= 0xFFFFFF00ull || key_bits < 1 || key_bits > 32) return GSWT_ERR_BAD_ARG;
    hipSetDevice(c->device);
    const uint32_t cap = (uint32_t)n;
    DevBuf<uint32_t> ka, kb, va, vb, ws;
    HIP_TRY(c, ka.ensure(n + 4)); HIP_TRY(c, kb.ensure(n + 4)); HIP_TRY(c, va.ensure(n + 1)); HIP_TRY(c, vb.ensure(n + 1));
    const size_t words = radix_ws_words(cap, key_bits) + 16;
    HIP_TRY(c, ws.ensure(words));
    HIP_TRY(c, hipMemset(ws.p, 0, words * 4));
    // the item count lives in device memory, as in a frame: {n, -, overflow flag = 0} in front of the histograms
    unsigned long long hn[4] = {(unsigned long long)n, 0ull, 0ull, 0ull};
    HIP_TRY(c, hipMemcpy(ws.p, hn, 32, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(ka.p, keys, n * 4, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(va.p, vals, n * 4, hipMemcpyHostToDevice));
    const int where = launch_sort(c->stream, ka.p, va.p, kb.p, vb.p, cap, reinterpret_cast<const unsigned long long*>(ws.p), key_bits, ws.p + 16);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(keys, where ? kb.p : ka.p, n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(vals, where ? vb.p : va.p, n * 4, hipMemcpyDeviceToHost));
    ka.release(); kb.release(); va.release(); vb.release(); ws.release();
    return GSWT_OK;
} GSWT_CATCH("gswt_debug_sort")

// Test hook: k_tile_depth_sort alone.  lens[t] = length of screen tile t's slice of the (tile-sorted) pair list, the slices back to back;
// vals / dkeys: the list's values and depth keys (n = sum of lens).  Sorts every slice's vals by its dkeys, stably, in place.  flagged_out:
// 1 when a slice was longer than the LDS buffer holds (nothing is then guaranteed about that slice; a frame would be re-run).
int gswt_debug_tile_depth_sort(gswt_ctx* c, const uint32_t* lens, size_t n_tiles, uint32_t* vals, const uint32_t* dkeys, size_t n, int* flagged_out)
try {
    if (!c || !lens || !vals || !dkeys || n_tiles == 0 || n_tiles > 0x7FFFFFFFull || n == 0 || n >= 0xFFFFFF00ull) return GSWT_ERR_BAD_ARG;
    hipSetDevice(c->device);
    std::vector<uint2> rg(n_tiles);
    size_t at = 0;
    for (size_t t = 0; t < n_tiles; t++) {                 // the device's encoding: (~start, end), (0, 0) for a tile without pairs
        rg[t] = lens[t] ? make_uint2(~(uint32_t)at, (uint32_t)(at + lens[t])) : make_uint2(0u, 0u);
        at += lens[t];
    }
    if (at != n) return fail(c, GSWT_ERR_BAD_ARG, "gswt_debug_tile_depth_sort: the lengths sum to %zu, not %zu", at, n);
    DevBuf<uint2> d_rg; DevBuf<uint32_t> d_vals, d_keys, d_vals2, d_keys2, d_long; DevBuf<unsigned long long> d_cnt;
    HIP_TRY(c, d_rg.ensure(n_tiles)); HIP_TRY(c, d_vals.ensure(n + 1)); HIP_TRY(c, d_keys.ensure(n + 1)); HIP_TRY(c, d_vals2.ensure(n + 1)); HIP_TRY(c, d_keys2.ensure(n + 1));
    HIP_TRY(c, d_long.ensure(2 * n_tiles + 4)); HIP_TRY(c, d_cnt.ensure(8));
    HIP_TRY(c, hipMemcpy(d_rg.p, rg.data(), n_tiles * 8, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(d_vals.p, vals, n * 4, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(d_keys.p, dkeys, n * 4, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemset(d_long.p, 0, (2 * n_tiles + 4) * 4));
    HIP_TRY(c, hipMemset(d_cnt.p, 0, 64));
    launch_tile_depth_sort(c->stream, d_rg.p, d_vals.p, d_keys.p, d_vals2.p, d_keys2.p, (int)n_tiles, d_long.p, d_cnt.p);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    unsigned long long cnt[4];
    HIP_TRY(c, hipMemcpy(cnt, d_cnt.p, 32, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(vals, d_vals.p, n * 4, hipMemcpyDeviceToHost));
    if (flagged_out) *flagged_out = cnt[3] != 0ull;
    d_rg.release(); d_vals.release(); d_keys.release(); d_vals2.release(); d_keys2.release(); d_long.release(); d_cnt.release();
    return GSWT_OK;
} GSWT_CATCH("gswt_debug_tile_depth_sort")

int gswt_debug_graph_stats(const gswt_ctx* c, unsigned long long out[3])
{
    if (!c || !out) return GSWT_ERR_BAD_ARG;
    out[0] = c->stat_graph_launches; out[1] = c->stat_graph_rebuilds; out[2] = c->stat_graph_node_updates;
    return GSWT_OK;
}

int gswt_debug_depth_stats(const gswt_ctx* c, unsigned long long out[3])
try {
    if (!c || !out) return GSWT_ERR_BAD_ARG;
    out[0] = c->stat_depth_local; out[1] = c->stat_depth_global; out[2] = c->depth_max_tile_len;
    return GSWT_OK;
} GSWT_CATCH("gswt_debug_depth_stats")

int gswt_debug_merge_stats(const gswt_ctx* c, unsigned long long out[2])
{
    if (!c || !out) return GSWT_ERR_BAD_ARG;
    out[0] = c->stat_groups_built; out[1] = c->stat_groups_reused;
    return GSWT_OK;
}

int gswt_debug_merge_stats_deep(const gswt_ctx* c, unsigned long long* out)
{
    if (!c || !out) return GSWT_ERR_BAD_ARG;
    *out = c->stat_groups_reused_deep;
    return GSWT_OK;
}

// Device timeline of frame slots (tests of the overlap claims): out_ms[0] = start of slot `ticket`'s frame kernels, [1] = their end,
// [2] = end of its gather + re-assembly on the ctx stream (NaN when the frame was not gathered), all in milliseconds after the START of
// slot `ticket_ref`'s frame.  Both frames must have been submitted with GSWT_OPT_TIMING >= 1 and have completed.
int gswt_debug_frame_times(gswt_ctx* c, int ticket_ref, int ticket, float out_ms[3])
try {
    if (!c || !out_ms || ticket_ref < 0 || ticket_ref >= kFrameSlots || ticket < 0 || ticket >= kFrameSlots) return GSWT_ERR_BAD_ARG;
    hipSetDevice(c->device);
    HIP_TRY(c, sync_all(c));
    FrameSlot& a = c->slots[ticket_ref];
    FrameSlot& b = c->slots[ticket];
    if (a.timing_level < 1 || b.timing_level < 1) return fail(c, GSWT_ERR_STATE, "gswt_debug_frame_times: frames without timing events");
    HIP_TRY(c, hipEventElapsedTime(&out_ms[0], a.ev[0], b.ev[0]));
    HIP_TRY(c, hipEventElapsedTime(&out_ms[1], a.ev[0], b.ev[6]));
    out_ms[2] = __builtin_nanf("");
    if (b.gather_recorded) HIP_TRY(c, hipEventElapsedTime(&out_ms[2], a.ev[0], b.ev_gather));
    return GSWT_OK;
} GSWT_CATCH("gswt_debug_frame_times")

int gswt_debug_read_ranges(gswt_ctx* c, uint32_t* out, size_t capacity_tiles, size_t* n_tiles)
try {
    if (!c || !n_tiles) return GSWT_ERR_BAD_ARG;
    *n_tiles = c->last_n_tiles;
    if (!out) return GSWT_OK;
    if (capacity_tiles < c->last_n_tiles) return fail(c, GSWT_ERR_CAPACITY, "buffer holds %zu tiles, need %u", capacity_tiles, c->last_n_tiles);
    hipSetDevice(c->device);
    HIP_TRY(c, sync_all(c));
    HIP_TRY(c, hipMemcpy(out, c->slots[c->last_slot].ranges.p, (size_t)c->last_n_tiles * 8, hipMemcpyDeviceToHost));
    for (uint32_t t = 0; t < c->last_n_tiles; t++)         // the device keeps (~start, end), (0, 0) for a tile without pairs
        out[2 * (size_t)t] = out[2 * (size_t)t + 1] ? ~out[2 * (size_t)t] : 0u;
    return GSWT_OK;
} GSWT_CATCH("gswt_debug_read_ranges")

int gswt_debug_read_projected(gswt_ctx* c, void* out, size_t capacity_entries, size_t* n_entries)
try {
    if (!c || !n_entries) return GSWT_ERR_BAD_ARG;
    *n_entries = (size_t)c->sets[c->cur_set].n_entries;
    if (!out) return GSWT_OK;
    if (!c->opt_debug_varyings || !c->dbg.p) return fail(c, GSWT_ERR_STATE, "enable GSWT_OPT_DEBUG_VARYINGS and render first");
    if (capacity_entries < c->sets[c->cur_set].n_entries) return fail(c, GSWT_ERR_CAPACITY, "buffer holds %zu entries, need %llu", capacity_entries, (unsigned long long)c->sets[c->cur_set].n_entries);
    hipSetDevice(c->device);
    HIP_TRY(c, sync_all(c));
    HIP_TRY(c, hipMemcpy(out, c->dbg.p, (size_t)c->sets[c->cur_set].n_entries * sizeof(Varyings), hipMemcpyDeviceToHost));
    return GSWT_OK;
} GSWT_CATCH("gswt_debug_read_projected")

}  // extern "C"
