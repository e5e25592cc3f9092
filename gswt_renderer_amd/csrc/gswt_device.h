// gswt_device.h -- device-side data layout shared by the kernels and the C-ABI host code.
// Internal to libgswt_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <string.h>
#include <type_traits>

namespace gswt {

constexpr int kTile = 16;            // screen tile edge, pixels (BASELINE north_star: 16x16 binning)
constexpr int kChunk = 256;          // list entries per projection workgroup
constexpr uint32_t kLodShift = 28;   // packed list entry = gs_index | lod_id << 28
constexpr uint32_t kIdxMask = (1u << kLodShift) - 1u;

// Device-side draw descriptor: what one reference draw call binds (renderer.rs:499-590).
struct DrawDev {
    uint32_t single_draw;
    int32_t valid_lod_id;
    uint32_t changing;
    int32_t changing_to_lower;
    uint32_t tile_lod;        // tile_id.x
    float off[3];             // tile offset
    uint32_t list_base;       // first packed entry of this draw's list in its arena
    uint32_t count;           // list length
    uint32_t merged;          // 0: static arena, 1: merged arena (has map ids)
    uint32_t slot_base;       // first composite-order slot of this draw (multiple of kChunk)
    uint32_t cull_enable;
    uint32_t lod;             // tid.0 for lod_enable
    float corners[12];
    uint32_t entry_base;      // index of the first entry in draw order (debug output)
    int32_t single_lod_id;    // TileUniforms.single_lod_id (debug draw mode 3)
    uint32_t tile_idx;        // tile_id.y (debug draw mode 1)
    uint32_t tile_view;       // tile_id.z (debug draw mode 4)
    uint32_t map_coord[2];    // TileUniforms.map_coord (sphere surface)
    uint32_t box_base;        // static draws: first chunk box of the draw's list (chunk k = the k-th 256 entries from the END of the list); ~0: none
    uint32_t xcd;             // the XCD (0..7) all chunks of this draw are projected on (gswt_set_draws: the least loaded one when the draw is planned)
};

// Per-frame constants (kernel argument, by value).
struct Frame {
    float V[16];       // view
    float GP[16];      // opengl_to_wgpu * projection   (gswt.wgsl:152-160)
    float VP[16];      // projection * view             (camera.rs:86-88) for the tile cull
    float focal[2];
    float htan[2];
    float cam_pos[3];
    float W, H;
    // scene uniforms
    float splat_scale, tile_width, clip_height, point_cloud_radius, transition_width_ratio, sphere_radius;
    uint32_t use_clip, surface_type, num_lod, draw_mode;
    uint32_t map_half_wh[2];
    int32_t center_coord[2];
    float transition_dist[16];
    float height_map_scale[3];
    float scene_scale[3];
    // render config
    float culling_dist;
    uint32_t lod_enable_mask;
    float t_eps;
    int32_t has_depth;       // proxy depth buffer bound
    // screen tiling / sharding
    int32_t width, height, tiles_x, tiles_y;
    int32_t shard_index, shard_count;      // interleaved tile-row sharding (count 1 = off)
    // contiguous tile-column band (off: col0 = 0, col1 = all columns).  tiles_x is the number of columns THIS ctx composites.
    int32_t col0, col1;
    int32_t out_w, out_x0;                 // width of the output image in pixels and the frame pixel column of its column 0
    int32_t band_cull;                     // 1: k_cull drops the draws / merged-group members whose splats cannot reach the band
    float loc_lo[3], loc_hi[3];            // tile-local bounds of every splat centre of the scene (gswt_upload_scene)
    float loc_max_trace;                   // largest trace of a stored covariance (>= its largest eigenvalue)
    float surf_zlo, surf_zhi;              // HeightMap surface: range of the mapped height h(x, y) * height_map_scale.z (0, 0 on the plain surface)
    float surf_f2;                         // bound of |F|^2 of the surface frame F (1 on the plain surface): Vrk -> F Vrk F^T
    int32_t hm_w, hm_h;
    uint32_t tiles_x_magic;                // floor(2^32 / tiles_x) + 1: tile -> (column, row) by multiply-high for tile ids below 2^16 (tile_xy)
    uint32_t map_wh_y, map_wh_y_magic;     // height of the tile map in cells (gswt.wgsl:53-56) and floor(2^32 / it) + 1 (quotients of 16-bit map ids by multiply-high)
    int32_t dbg_flags;       // profiling ablations (GSWT_OPT_DEBUG_FLAGS); 0 in normal operation
};

// Projected splat record consumed by the compositor (32 B, two 16-B words; one 32-B-aligned sector per gather).
//   q0 = (iux, iuy, ivx, ivy)   q1 = (ndc.x, ndc.y, alpha, rgba8 bits)
// iu / iv: rows of the inverse affine map pixel -> quad space (F2); the centre stays in NDC (round 4): the compositor derives its offset
// from a tile origin with one rounding (F3), instead of subtracting the origin from a pixel-space centre that carries two roundings at ~W.
// The conservative pixel half extents of |p| <= 2 are re-derived from iu / iv by the compositor's staging lane; the depth
// lives in a side array (4 B per slot) that only depth-tested (proxy depth bound) and depth-ordered frames touch.
struct __attribute__((aligned(32))) Rec {
    float iux, iuy, ivx, ivy;
    float ndcx, ndcy, alpha, rgba8;
};

// Device-side merged-list building (see k_mg_* in gswt_kernels.hip)
struct MergeSeg {
    uint32_t group, src, len, start, gs_offset, map_index, lod, _pad;
};
struct MergeGroup {
    uint32_t base, len;       // range in the BUILD space (the concatenation of the groups that are sorted this event)
    int32_t mn, mx;
    uint32_t out_base;        // first entry of the group in the merged arrays
    uint32_t _pad[3];
};
// A group whose (view, ordered member tids) equal a group of the previous sort event: its list is copied from the previous
// draw set, map ids rewritten member by member (the reference's LRU hit, wangtile.rs:575-593).
struct MergeCopy {
    uint32_t src, dst, len;   // ranges in the source / new merged arrays
    uint32_t first_pair, n_pairs;   // (old map index, new map index) pairs of its members in the remap table; n_pairs = 0: map ids unchanged
    uint32_t src_set;         // which retained draw set holds the source list (MergeSources)
    uint32_t _pad[2];
};
// The merged arrays of the retained draw sets a copy job may read from: the lists of the last kMergeSources - 1 sort events stay
// addressable, keyed by (view, ordered member tile ids, transition states) -- the reference's LRU of merged lists (wangtile.rs:427,575-593).
constexpr int kMergeSources = 12;
struct MergeSources {
    const uint32_t* list[kMergeSources];
    const uint32_t* map[kMergeSources];
};

// vs_main varyings for the debug/parity hook (48 B, same layout as the oracle's orc_splat)
struct Varyings {
    int32_t visible;
    float ndc[2];
    float depth;
    float major[2];
    float minor[2];
    float rgba[4];
};

// Kernel arguments of the background passes (gswt_passes.hip)
struct SkyArgs {
    float V[16];          // view (only its rotation is used: skybox.wgsl:41-47 removes the translation)
    float p00, p11;       // projection[0][0], projection[1][1] (symmetric perspective, camera.rs:94)
    int width, height, face_size, equirectangular;
};

struct ProxyArgs {
    // proxy.wgsl Uniforms
    float height_offset, tile_width, width_scale, clip_height, brightness;
    uint32_t surface_type, use_clip, black_background;
    float V[16], GP[16];
    float p00, p11;
    float cam[3];
    uint32_t map_half_wh[2];
    float height_map_scale[3];
    // grid
    int nx, ny;
    float cs, gx0, gy0;
    // resources
    int hm_w, hm_h, tex_size, n_mips;
    uint32_t mip_off[16];          // float4 offsets of the mip levels
    int width, height;
};

// ---- hipGraph replay of a frame's launch sequence (GSWT_OPT_GRAPH) -------------------------------------------------------------
// The frame is a fixed chain of kernel launches whose grids follow capacities; what changes from frame to frame is a handful of
// kernel arguments (the camera block, the output pointer).  With a recorder set, the launch sites of the frame path do not
// launch: they leave (function, grid, block, packed arguments) per kernel, and the caller replays the chain as ONE
// hipGraphLaunch, after updating only the kernel nodes whose record differs from the previous frame's.
struct GraphNodeRec {
    const void* fn = nullptr;
    dim3 grid, block;
    uint32_t n_args = 0, n_bytes = 0;
    uint16_t offs[40];
    alignas(16) unsigned char args[1280];
    template <typename T>
    void push(const T& v)
    {
        static_assert(alignof(T) <= 16, "kernel argument alignment");
        n_bytes = (n_bytes + (uint32_t)alignof(T) - 1u) & ~((uint32_t)alignof(T) - 1u);
        if (n_args >= 40u || n_bytes + sizeof(T) > sizeof(args)) { fn = nullptr; return; }      // (checked by the caller: a null fn fails the frame)
        memcpy(args + n_bytes, &v, sizeof(T));
        offs[n_args++] = (uint16_t)n_bytes;
        n_bytes += (uint32_t)sizeof(T);
    }
    bool same(const GraphNodeRec& o) const
    {
        return fn == o.fn && grid.x == o.grid.x && grid.y == o.grid.y && grid.z == o.grid.z && block.x == o.block.x && n_args == o.n_args &&
               n_bytes == o.n_bytes && memcmp(args, o.args, n_bytes) == 0;
    }
};
// k_project's two-level sums: one word per super-group (256 chunks) for the pairs and one for the visible splats, each on a cache line of its
// own (16 words apart).  Device-scope atomics on one 64-byte line retire at ~10 ns each whichever XCD they come from (measured: 43 k atomics
// on 4 lines = +125 us), and side by side the 64 super-group words of c3 were 4 + 4 lines taking 2.7 k atomics each.
constexpr uint32_t kSuperStride = 16;
constexpr uint32_t kGraphMaxNodes = 32;          // reference order: 10-12 kernels per frame; GSWT_ORDER_DEPTH: 19-23
struct GraphRec {
    GraphNodeRec nodes[kGraphMaxNodes];
    uint32_t n = 0;
    bool overflow = false;
};
// the recorder of the calling thread (null: the launch sites launch); set around the frame's launch sequence by gswt_api.hip
GraphRec*& graph_recorder();

template <typename... KA, typename... A>
inline void graph_record(GraphRec* rec, void (*k)(KA...), dim3 g, dim3 b, A&&... a)
{
    static_assert(sizeof...(KA) == sizeof...(A), "kernel argument count");
    if (rec->n >= kGraphMaxNodes) { rec->overflow = true; return; }
    GraphNodeRec& nd = rec->nodes[rec->n++];
    nd.fn = reinterpret_cast<const void*>(k);
    nd.grid = g; nd.block = b; nd.n_args = 0; nd.n_bytes = 0;
    memset(nd.args, 0, sizeof(nd.args));            // padding bytes take part in the comparison
    (nd.push(static_cast<typename std::decay<KA>::type>(a)), ...);
    if (!nd.fn) rec->overflow = true;
}
#define GSWT_LAUNCH(K, G, B, S, ...)                                                                     \
    do {                                                                                                 \
        if (GraphRec* rec_ = graph_recorder()) graph_record(rec_, K, G, B, __VA_ARGS__);                 \
        else hipLaunchKernelGGL(K, G, B, 0, S, __VA_ARGS__);                                             \
    } while (0)

// A launch that carries its own pair of timing events (E0, E1 non-null, no graph recorder): hipExtLaunchKernelGGL binds both to the DISPATCH,
// so hipEventElapsedTime(E0, E1) is the kernel's own begin -> end on the device -- the interval rocprofv3 --kernel-trace reports -- rather than
// "previous command of the stream done -> this kernel done" of two hipEventRecord calls (GSWT_KERNEL_EVENTS=0).  Measured at the end of round 4
// (profiles/r04_kernel_events.txt): the two read alike (c3 fly path, k_composite 0.104-0.106 against 0.100-0.114 ms, frame rate unchanged), and
// under rocprofv3 either agrees with the profiler's own average of the same run (0.0907 / 0.0919 against 0.0909 / 0.0969 ms): what separates the
// bench line's kernel time (0.105 ms) from a rocprofv3 summary (0.089-0.097) is the run -- under the profiler fewer frames overlap (4 090-4 750
// against 5 450 frames/s) and the kernel shares the chip with less -- not the events.
bool kernel_events_enabled();
#define GSWT_LAUNCH_TIMED(K, G, B, S, E0, E1, ...)                                                       \
    do {                                                                                                 \
        if ((E0) && (E1) && !graph_recorder() && kernel_events_enabled())                                \
            hipExtLaunchKernelGGL(K, G, B, 0, S, E0, E1, 0, __VA_ARGS__);                                \
        else {                                                                                           \
            if (E0) hipEventRecord(E0, S);                                                               \
            GSWT_LAUNCH(K, G, B, S, __VA_ARGS__);                                                        \
            if (E1) hipEventRecord(E1, S);                                                               \
        }                                                                                                \
    } while (0)

}  // namespace gswt
