/*
 * gswt_hip.h -- C ABI of libgswt_hip.so, the MI355X-native drop-in for the GPU half
 * of the GSWT hot path: per-frame Wang-tile instancing, Gaussian projection,
 * ordering, 16x16 screen-tile binning and alpha compositing.
 *
 * Each entry point names the reference interface it replaces (file:line are into
 * zengyf131/gswt_renderer).  Plain pointers and sizes only; no C++/torch types.
 * INTEGRATION.md shows the Rust `extern "C"` block a maintainer would add.
 *
 * Threading (mirrors renderer.rs: render runs on the main thread only): a ctx is
 * single-owner, one caller thread at a time.  A ctx owns one HIP stream; all work
 * of a call is enqueued on it.  Every host pointer is borrowed for the duration of
 * the call only.  No call aborts or throws across the ABI: all return an int status
 * and gswt_last_error() holds the text of the last failure on that ctx.
 */
#ifndef GSWT_HIP_H
#define GSWT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSWT_API __attribute__((visibility("default")))

typedef struct gswt_ctx gswt_ctx;

enum {
    GSWT_OK = 0,
    GSWT_ERR_BAD_ARG = -1,   /* reference: assert!/unwrap panic on malformed input      */
    GSWT_ERR_CAPACITY = -2,  /* reference: silent wgpu validation error past 20 000 draws
                                / 10 M merged splats (renderer.rs:253,273); here buffers
                                grow on demand, this code only reports a failed grow     */
    GSWT_ERR_HIP = -3,       /* any HIP runtime error                                    */
    GSWT_ERR_STATE = -4,     /* call order violated (e.g. render before upload_scene)    */
    /* -5 is GSWT_ERR_IO of gswt_host.h */
    GSWT_ERR_RCCL = -6       /* RCCL missing / a collective failed (text in gswt_last_error) */
};

/* Composite order.  REFERENCE reproduces the reference's order exactly: draw rank
 * (back-to-front tile list, wangtile.rs:489-499) x position in that draw's presorted
 * list (scene.rs:685-695).  DEPTH is a true per-splat global depth sort. */
enum { GSWT_ORDER_REFERENCE = 0, GSWT_ORDER_DEPTH = 1 };

/* ---- byte-exact mirrors of the reference's uniform blocks ------------------------ */

/* CameraUniforms, camera.rs:158-167 / gswt.wgsl:437-444 (176 bytes) */
typedef struct {
    float projection[16]; /* column-major */
    float view[16];
    float focal[2];
    float viewport[2];
    float htan_fov[4];
    float cam_pos[4];
} gswt_camera_uniforms;

/* SceneUniforms, renderer.rs:602-622 / gswt.wgsl:446-463 (160 bytes) */
typedef struct {
    float splat_scale;
    float tile_width;
    uint32_t use_clip;
    float clip_height;
    uint32_t surface_type; /* 0 None, 1 HeightMap, 2 Sphere (structure.rs:435-440) */
    float sphere_radius;
    float point_cloud_radius;
    float transition_width_ratio;
    uint32_t num_lod;
    uint32_t draw_mode;
    uint32_t map_half_wh[2];
    int32_t center_coord[2];
    uint32_t _pad0[2];
    float transition_dist_vec[16];
    float height_map_scale[4];
    float scene_scale[4];
} gswt_scene_uniforms;

/* TileUniforms, renderer.rs:675-689 / gswt.wgsl:465-476 (80 bytes) */
typedef struct {
    uint32_t single_draw;
    uint32_t map_index;
    int32_t single_lod_id;
    int32_t valid_lod_id;
    uint32_t changing;
    int32_t changing_to_lower;
    uint32_t _pad0[2];
    uint32_t tile_id[4]; /* lod, tile, view, 0 */
    float offset[4];
    uint32_t map_coord[4];
} gswt_tile_uniforms;

/* One static presorted list of PreloadData.tile_base_data[lod][tile][view]
 * (structure.rs:546-554: gs_index + gs_lod_id, splat_count entries each). */
typedef struct {
    const uint32_t *gs_index;
    const uint32_t *gs_lod_id;
    uint32_t splat_count;
    uint32_t _pad;
} gswt_base_list;

/* One iteration of the draw loop in GSWTRenderer::render (renderer.rs:466-591):
 * the tile uniforms written at :499-515 plus which instance buffers get bound. */
typedef struct {
    gswt_tile_uniforms tile;
    /* list selection, renderer.rs:517-579 */
    uint32_t merged;        /* 1: merged group (Some(render_data_value)) -> range of the
                               merged arrays passed to gswt_set_draws; 0: static base list */
    uint32_t base_lod;      /* static list [base_lod][base_tile][base_view]; the caller   */
    uint32_t base_tile;     /* applies renderer.rs:564-572 (Changing(false) -> lod-1)     */
    uint32_t base_view;
    uint32_t merged_offset; /* first element of this draw in the merged arrays            */
    uint32_t merged_count;  /* render_data_value.splat_count                              */
    uint32_t merged_has_lod;/* 1 iff single_lod_id == -1 (gs_lod_id uploaded, :544-556)   */
    /* CPU viewport culling inputs, renderer.rs:472-494 (non-merged draws only) */
    uint32_t cull_enable;   /* 1 iff render_data_key.tid.len() == 1                        */
    float corners[12];      /* tile_instance.corner_data[ci].0, ci = 0..3                  */
    uint32_t lod;           /* tid.0, index into lod_enable (renderer.rs:495)              */
    uint32_t merged_group;  /* gswt_set_draws_merge_groups only: index of this draw's group        */
    uint32_t _pad[2];
} gswt_draw;

/* The parts of RenderConfig (structure.rs:346-388) read on the hot path. */
typedef struct {
    float culling_dist;       /* render_config.culling_dist, renderer.rs:490 */
    uint32_t lod_enable_mask; /* bit l = render_config.lod_enable[l], renderer.rs:495 */
    int32_t order_mode;       /* GSWT_ORDER_* */
    float transmittance_eps;  /* front-to-back early-out threshold; 0 = never stop early */
    /* screen-tile sharding for multi-GPU; shard_count <= 1 renders the whole frame.
       GSWT_SHARD_ROWS: this ctx composites only the 16-px tile rows with (row % shard_count) == shard_index and writes
         them compacted, in row order (gswt_shard_rows_padded rows x W).  Every rank projects every splat.
       GSWT_SHARD_COLUMNS: this ctx composites the contiguous band of ceil(tiles_x / shard_count) tile columns number
         shard_index (H rows x gswt_shard_cols_padded pixels) and, on the plain surface, skips the draws none of whose
         splats can reach the band -- projection is sharded too.  Pair counts are even across column bands for a
         horizon-dominated view, across contiguous row bands they are not. */
    int32_t shard_index;
    int32_t shard_count;
    int32_t shard_mode;       /* GSWT_SHARD_* */
    uint32_t _pad;
} gswt_render_config;

enum { GSWT_SHARD_ROWS = 0, GSWT_SHARD_COLUMNS = 1 };

/* Per-stage device times of the last gswt_render (hipEvent, ms) and workload sizes.  ms_ranges is ~0 since the per-tile
 * [start, end) table is left by the last pass of the pair sort (its time is inside ms_sort); ms_scan is unused (no scan launch). */
typedef struct {
    float ms_project, ms_scan, ms_emit, ms_sort, ms_ranges, ms_composite, ms_total;
    uint32_t n_draws;
    uint64_t n_instanced; /* N: list entries iterated             */
    uint64_t n_visible;   /* survivors of the vertex stage        */
    uint64_t n_pairs;     /* P: (splat, 16x16 screen tile) pairs  */
    uint32_t n_tiles;     /* screen tiles composited by this ctx  */
    uint32_t _pad;
    float ms_composite_kernel; /* k_composite alone (ms_composite also covers work-item setup + k_combine) */
    float _pad2;
} gswt_timings;

/* ---- lifecycle ------------------------------------------------------------------- */

/* wgpu adapter/device acquisition in State::new (state.rs:56-92). */
GSWT_API int gswt_create(int device_id, gswt_ctx **out);
GSWT_API void gswt_destroy(gswt_ctx *ctx);
GSWT_API const char *gswt_last_error(const gswt_ctx *ctx);
/* Runs on a user-provided HIP stream (hipStream_t as void*) instead of the ctx's own; NULL returns to a stream of the ctx's own. */
GSWT_API int gswt_set_stream(gswt_ctx *ctx, void *hip_stream);

/* Options (no reference counterpart; test / profiling switches).
 * NO_LOD_PREFILTER: a plain tile iterates the interleaved LOD l / l+1 base list exactly as the
 *   reference binds it (renderer.rs:571-578) instead of the pre-filtered own-LOD list; results
 *   are identical (gswt.wgsl:38-42 discards the other LOD), only the entry count differs.
 * DEBUG_VARYINGS: keep vs_main's per-entry outputs for gswt_debug_read_projected. */
enum { GSWT_OPT_NO_LOD_PREFILTER = 1, GSWT_OPT_DEBUG_VARYINGS = 2,
       GSWT_OPT_SEGMENT = 3 /* pairs per compositor work item, multiple of 256 (default 1536; dense scenes with the early-out on
                               gain from up to 4096: a segment cannot skip what the segments in front of it already saturated) */,
       GSWT_OPT_DEBUG_FLAGS = 4 /* ablation bits for profiling (the image is wrong when nonzero): only the measurement build
                                   (-DGSWT_EXPERIMENTS, `make variants`) has them; the product library rejects a nonzero value */,
       GSWT_OPT_TIMING = 5 /* hipEvent timing: 0 none, 1 frame + k_composite, 2 every stage (default) */,
       GSWT_OPT_PAIR_CAP = 6 /* test hook: pin the pair-buffer capacity to `value` pairs until a frame overflows it (0: automatic) */,
       GSWT_OPT_NO_MERGE_REUSE = 7 /* gswt_set_draws_merge_groups re-sorts every merged group at every sort event instead of copying
                                      the groups that did not change since the previous one (results are identical) */,
       GSWT_OPT_DEFER_SWAP = 8 /* 1: a gswt_set_draws* call takes effect with the first frame submitted AFTER its uploads and device-side
                                  list builds have finished (frames submitted meanwhile keep the previous draw list and nothing waits);
                                  n >= 2: with the n-th frame submitted after the call, finished or not (deterministic: for ranks that
                                  render shards of the same frames); 0 (default): with the next frame, which then waits for them on
                                  the device */,
       GSWT_OPT_GRAPH = 9 /* 1: a frame's kernel launches are replayed as ONE hipGraphLaunch per frame slot (a chain of kernel nodes;
                             only the nodes whose grid or arguments changed since the slot's previous frame are updated in the
                             executable graph) instead of ~13 separate launches: less submitting-thread time per frame, same
                             kernels, same results.  Frames with GSWT_OPT_TIMING > 0 or debug varyings
                             launch as before.  0 (default): separate launches */,
       GSWT_OPT_STRICT_VS = 10 /* 1 (default): the vertex stage evaluates gswt.wgsl:152-258,260-265,402-419 operator by operator -- one correctly
                                  rounded binary32 operation per written `*` `+` `-` `/`, full matrix products, no fused multiply-add; per
                                  splat bit-identical to the CPU checker's restatement of the shader text.  0: the rounding sequence "v2"
                                  (fma chains, one reciprocal per quotient: also legal WGSL, ~25 % fewer instructions, the default until
                                  round 3); the two differ by up to 5e-4 in the image on thin ellipses */,
       GSWT_OPT_DEPTH_PASSES = 12 /* test hook: 8-bit radix passes on the depth bits the next GSWT_ORDER_DEPTH frame launches (1..4; default 3,
                                     then as many as the depth ranges of the recent frames needed); a frame whose visible depths span more
                                     bits is flagged on the device and re-run with more, like a pair-buffer overflow */,
       GSWT_OPT_COMPOSITE = 13 /* compositor kernel: 0 (default) k_composite -- 256-pair batches staged by the whole workgroup, two barriers per
                                  batch; 1 k_composite_dw -- the four waves of a work item decoupled (128-pair batches through a ring of three LDS
                                  buffers, ready / consumed counters instead of barriers); 2 k_composite<FOLD> -- the segments of a long tile list
                                  are folded by whichever of their workgroups finishes last (agent-scope stores + a ticket per tile), tiles
                                  without pairs are work items: no k_combine launch behind the compositor.  Same image bit for bit */,
       GSWT_OPT_DEPTH_SORT = 14 /* how GSWT_ORDER_DEPTH orders the pairs: 0 (default) / 2 = tile passes first (depth bits as payload), then every
                                   screen tile's slice is depth-sorted by one wave / workgroup (k_tile_depth_sort: inside LDS up to 16 384 pairs,
                                   through global memory beyond); 1 = global radix passes on the depth bits in front of the tile passes.
                                   Same image bit for bit */,
       GSWT_OPT_NO_CHUNK_CULL = 15 /* 1: the per-chunk frustum cull in front of the projection is off (every 256-entry chunk of a draw that
                                      survives the reference's tile cull is projected, as until round 3).  Same image bit for bit: the cull
                                      only leaves out chunks none of whose splats vs_main's own frustum test (gswt.wgsl:163-167) would keep */,
       GSWT_OPT_ITEM_ORDER = 16 /* order in which the compositor's work items (screen tile, segment of its pair list) are handed out: 0 = tile
                                  order, 1 = heaviest first (full segments, then the remainders by falling length).  Same image bit for bit */ };
GSWT_API int gswt_set_option(gswt_ctx *ctx, int key, int value);

/* GSWTRenderer::new (renderer.rs:31-349): uploads PreloadData.tile_splats_merged.tex_data
 * (8 u32 per splat, renderer.rs:236-248) and all static base lists (renderer.rs:290-327).
 * lists is [n_lod][n_tile][n_view] row-major. */
GSWT_API int gswt_upload_scene(gswt_ctx *ctx, const uint32_t *tex_data, size_t n_splats,
                               const gswt_base_list *lists, int n_lod, int n_tile, int n_view);

/* GSWTRenderer::configure (renderer.rs:351-405): height map, R32Float / linear / repeat.
 * height_map may be NULL (surface_type None). */
GSWT_API int gswt_configure(gswt_ctx *ctx, const float *height_map, int hm_w, int hm_h);

/* Swap-in of a SortData (state.rs:361-376) = the per-sort-event part of render():
 * the ordered draw list (back-to-front, renderer.rs:466) and the concatenated
 * gs_index / gs_map_id / gs_lod_id arrays of all merged groups (renderer.rs:517-561).
 * merged_lod_id may be NULL when no draw has merged_has_lod. */
GSWT_API int gswt_set_draws(gswt_ctx *ctx, const gswt_draw *draws, int n_draws,
                            const uint32_t *merged_gs_index, const uint32_t *merged_map_id,
                            const uint32_t *merged_lod_id, size_t n_merged);

/* On-device merged lists (replaces the CPU hot loop of sort_tiles, wangtile.rs:595-670, and the per-sort-event
 * upload of 12 B per merged splat, renderer.rs:517-561).
 * gswt_upload_raw_depth: TileBaseData.raw_depth of every [lod][tile][view] (structure.rs:551; one i32 per splat
 *   of that tile scene) and splats_merge_offset[lod][tile] (wangtile.rs:34), once after gswt_upload_scene.
 * gswt_set_draws_merge_groups: like gswt_set_draws, but the merged arrays are built on the device from the
 *   group descriptions: group g = one MergedFrom tile (view_id, members in from_vec order); member = (map index,
 *   tid, and the other LOD it is Changing to, or -1).  A merged draw's merged_offset / merged_count must be
 *   the group's range in the concatenation of all groups (sum of the members' raw-depth lengths, in order);
 *   draws[i].merged_group names its group. */
typedef struct {
    uint32_t view_id;
    uint32_t first_member;
    uint32_t n_members;
    uint32_t _pad;
} gswt_merge_group;
typedef struct {
    uint32_t map_index;
    uint32_t lod, tile;
    int32_t other_lod;   /* lod + 1 for Changing(true), lod - 1 for Changing(false), -1 otherwise */
} gswt_merge_member;
GSWT_API int gswt_upload_raw_depth(gswt_ctx *ctx, const int32_t *const *raw_depth /* [n_lod*n_tile*n_view] */,
                                   const uint32_t *counts /* [n_lod*n_tile] */,
                                   const uint32_t *merge_offset /* [n_lod*n_tile] */);
GSWT_API int gswt_set_draws_merge_groups(gswt_ctx *ctx, const gswt_draw *draws, int n_draws,
                                         const gswt_merge_group *groups, int n_groups,
                                         const gswt_merge_member *members, int n_members);
/* Test hook: the device-resident merged arrays (gs_index | lod << 28, map id), n entries each. Host pointers. */
GSWT_API int gswt_debug_read_merged(gswt_ctx *ctx, uint32_t *packed_list, uint32_t *map_id, size_t capacity, size_t *n);

/* GSWTRenderer::render (renderer.rs:407-592), per frame.  bg_rgba (W*H*4 f32) is the
 * colour attachment content the pass loads (LoadOp::Load, :425; skybox/proxy output) or
 * NULL for transparent black; bg_depth (W*H f32) is the proxy depth buffer (:433-434) or
 * NULL for the 1.0 clear (:436).  out_rgba receives rows_out*W*4 f32 where rows_out = H,
 * or the shard's rows when cfg->shard_count > 1 (see gswt_shard_rows).  Pointers are
 * device pointers when *_on_device is nonzero, host pointers otherwise. */
GSWT_API int gswt_render(gswt_ctx *ctx, const gswt_camera_uniforms *camera,
                         const gswt_scene_uniforms *scene, const gswt_render_config *cfg,
                         int width, int height,
                         const float *bg_rgba, const float *bg_depth, int bg_on_device,
                         float *out_rgba, int out_on_device);

/* Asynchronous form of gswt_render for pipelining consecutive frames (State::render is called once per
 * display refresh; with the device library the next frame can be queued while the previous one is still
 * executing).  All pointers are DEVICE pointers.  gswt_render_async enqueues the frame and returns a
 * ticket; gswt_render_wait(ticket) blocks the host until it finished, re-runs it if the pair buffers had
 * to grow, and makes gswt_last_timings refer to it.  At most gswt_frame_slots() (= 4) frames may be in
 * flight (enqueuing one more waits for the oldest).  Each frame in flight runs on its own internal stream
 * with its own per-frame buffers, so the frames OVERLAP on the GPU; frames in flight together must write
 * different output buffers.  Ordering against the ctx stream (gswt_set_stream):
 *   - a frame starts after all work submitted to the ctx stream before its gswt_render_async call
 *     (producers of bg_*, earlier consumers of out_rgba_dev);
 *   - gswt_render_fence(ticket) orders the ctx stream behind that frame, so work submitted to the ctx stream afterwards sees
 *     its output (e.g. the RCCL all-gather of the shards).  All-or-nothing like GSWTRenderer::render (renderer.rs:407-414): a
 *     frame whose pair buffers overflowed is re-run with larger ones BEFORE the fence releases the ctx stream, so a fenced
 *     consumer never reads an incomplete image.  Overflow is only known on the host, hence the call blocks the host until
 *     THIS frame has finished (not the younger frames in flight: submit them before fencing frame i and the GPU stays
 *     busy).  The ticket stays valid; gswt_render_wait(ticket) then returns at once with the frame's status and timings.
 *   - gswt_set_draws*, gswt_upload_scene, gswt_configure first run every frame in flight to completion against the state it
 *     was submitted with (tickets stay valid). */
GSWT_API int gswt_render_async(gswt_ctx *ctx, const gswt_camera_uniforms *camera,
                               const gswt_scene_uniforms *scene, const gswt_render_config *cfg,
                               int width, int height, const float *bg_rgba_dev, const float *bg_depth_dev,
                               float *out_rgba_dev, int *ticket);
GSWT_API int gswt_frame_slots(void);   /* frames that may be in flight at once */
GSWT_API int gswt_render_wait(gswt_ctx *ctx, int ticket);
GSWT_API int gswt_render_fence(gswt_ctx *ctx, int ticket);

/* ---- background passes (state.rs:384-392) -------------------------------------------------------
 * State::render runs Skybox::render and Proxy::render before GSWTRenderer::render; their colour /
 * depth targets are what gswt_render takes as bg_rgba / bg_depth.  Here they are per-pixel compute
 * passes on the ctx stream (a following gswt_render / gswt_render_async is ordered after them). */

/* proxy.wgsl `Uniforms` / proxy.rs:470-511, byte-exact (224 B) */
typedef struct gswt_proxy_uniforms {
    float height_offset;            /*   0 */
    float tile_width;               /*   4 */
    uint32_t surface_type;          /*   8 */
    float width_scale;              /*  12 */
    uint32_t map_proxy;             /*  16 */
    uint32_t use_clip;              /*  20 */
    float clip_height;              /*  24 */
    float brightness;               /*  28 */
    uint32_t black_background;      /*  32 */
    uint32_t _pad0[3];              /*  36 */
    float view[16];                 /*  48 */
    float projection[16];           /* 112 */
    uint32_t map_half_wh[2];        /* 176 */
    int32_t center_coord[2];        /* 184 */
    float height_map_scale[4];      /* 192 */
    float cam_pos[4];               /* 208 */
} gswt_proxy_uniforms;              /* 224 */

/* Skybox::configure (skybox.rs:341-455): the cube map, 6 faces (+X -X +Y -Y +Z -Z) of face_size^2 RGBA f32 texels
 * (host pointer).  `equirectangular` is Skybox.is_equi (skybox.wgsl:35-38). */
GSWT_API int gswt_skybox_configure(gswt_ctx *ctx, const float *faces_rgba, int face_size, int equirectangular);
/* Skybox::render (skybox.rs:457-488): LoadOp::Clear + the cube drawn at depth 1 = every pixel of out_rgba_dev
 * (W*H*4 f32, device) is overwritten with (cube rgb, 1). */
GSWT_API int gswt_skybox_render(gswt_ctx *ctx, const gswt_camera_uniforms *camera, int width, int height, float *out_rgba_dev);
/* Proxy::configure (proxy.rs:208-364): the mip chain of the proxy texture (square, tex_size >> level, RGBA f32, host
 * pointers) and the grid dimension of the "full" proxy (Proxy::GRID_DIM = 2048, proxy.rs:29).  The height map is the one
 * given to gswt_configure. */
GSWT_API int gswt_proxy_configure(gswt_ctx *ctx, const float *const *mips, int tex_size, int n_mips, int grid_dim);
/* ONE draw of Proxy::render (proxy.rs:366-447): u->map_proxy selects the GRID_DIM grid (0, `proxy_full`) or the tile-map
 * grid (1, `proxy_map`).  rgba_dev (W*H*4) and depth_dev (W*H) are device buffers updated in place: colour LoadOp::Load,
 * depth test Less with depth write.  clear_depth != 0 first fills depth_dev with 1.0 (the pass's LoadOp::Clear(1.0)). */
GSWT_API int gswt_proxy_render(gswt_ctx *ctx, const gswt_proxy_uniforms *u, int width, int height, float *rgba_dev, float *depth_dev,
                               int clear_depth);


/* Number of pixel rows the shard (index, count) owns for a frame of `height` rows. */
GSWT_API int gswt_shard_rows(int height, int shard_index, int shard_count);
/* Scatter `shard_count` gathered shard images (concatenated in shard order, as an
 * all-gather delivers them; each padded to gswt_shard_rows_padded rows) back into a full
 * H x W frame.  Device pointers; runs on the ctx stream. */
GSWT_API int gswt_shard_rows_padded(int height, int shard_count);
/* Width in pixels of every rank's shard image in GSWT_SHARD_COLUMNS mode (ceil(tiles_x / shard_count) * 16). */
GSWT_API int gswt_shard_cols_padded(int width, int shard_count);
GSWT_API int gswt_unshard(gswt_ctx *ctx, const float *gathered, int width, int height,
                          int shard_count, float *out_rgba);
/* The same for either shard mode (GSWT_SHARD_COLUMNS: the gathered shards are H x gswt_shard_cols_padded images). */
GSWT_API int gswt_unshard_mode(gswt_ctx *ctx, const float *gathered, int width, int height,
                               int shard_count, int shard_mode, float *out_rgba);

/* ---- multi-GPU: the framebuffer all-gather behind the ABI (new; the reference is single-GPU) ---------------------
 * One ctx per GPU.  Every rank renders its shard (cfg->shard_index = rank, shard_count = world) with gswt_render_async,
 * then gswt_render_gather(ticket, frame) = overflow-safe fence + all-gather of the equal-sized shard images + the index
 * permutation of gswt_unshard_mode, all on the ctx stream: `frame` (W*H*4 f32, device) then holds the whole image on
 * every rank.  Two transports:
 *   RCCL (one process per GPU, xGMI): rank 0 calls gswt_comm_unique_id, ships the 128 bytes to the other ranks by any
 *     means (MPI, a file, torch.distributed ...), every rank calls gswt_comm_init(ctx, id, rank, world) ->
 *     ncclCommInitRank; the gather is one ncclAllGather.  librccl is loaded on first use (dlopen), so a single-GPU host
 *     does not need it; GSWT_ERR_RCCL reports a missing library or a failed collective.
 *   hipMemcpyPeerAsync (all GPUs in ONE process, no RCCL): gswt_group_init binds n contexts into a group;
 *     gswt_group_render_gather pushes every rank's shard into every peer's gather buffer (n x n peer copies), orders the
 *     peers' streams behind them with events and runs the permutation on each.  Also what the single-GPU tests use
 *     (several contexts on one device). */
#define GSWT_COMM_ID_BYTES 128
GSWT_API int gswt_comm_unique_id(void *id_out /* GSWT_COMM_ID_BYTES */);
GSWT_API int gswt_comm_init(gswt_ctx *ctx, const void *unique_id, int rank, int world);
GSWT_API int gswt_comm_destroy(gswt_ctx *ctx);
GSWT_API int gswt_render_gather(gswt_ctx *ctx, int ticket, float *frame_out_dev);
GSWT_API int gswt_group_init(gswt_ctx *const *ctxs, int n);          /* rank r = ctxs[r]; undone by gswt_comm_destroy on each */
GSWT_API int gswt_group_render_gather(gswt_ctx *const *ctxs, const int *tickets, float *const *frames_out_dev, int n);


/* ---- SortData: what one sort event hands from the worker to the renderer --------------- */
/* One element of SortData.tile_instance_vec + render_data_vec (structure.rs:488-509,670-694). */
typedef struct {
    uint32_t lod, tile, view_id;      /* tid.0, tid.1, view_id                                   */
    float tile_offset[3];
    uint32_t map_index;
    uint32_t map_coord[2];
    float tile_center[3];
    int32_t transition;               /* 0 None, 1 Spawning, 2 Changing(false), 3 Changing(true)   */
    float spawning_factor;
    uint32_t has_corners;
    float corners[12];                /* corner_data[ci].0                                        */
    uint32_t key_len;                 /* render_data_key.tid.len()                                */
    uint32_t merged;                  /* Some(render_data_value)                                  */
    uint32_t merged_offset;           /* into the concatenated merged arrays of gswt_sort_data    */
    uint32_t merged_count;
    int32_t single_lod_id;
    uint32_t cache_hit;               /* value came from the LRU cache (wangtile.rs:575-593)      */
    uint32_t merged_group;            /* index into gswt_sort_data.groups when merged               */
} gswt_sorted_tile;

typedef struct {
    uint32_t scene_id;
    uint32_t n_tiles;
    const gswt_sorted_tile *tiles;    /* back-to-front */
    size_t n_merged;
    const uint32_t *merged_gs_index, *merged_map_id, *merged_lod_id;   /* NULL in device-merge mode */
    /* group descriptions for gswt_set_draws_merge_groups (always filled) */
    uint32_t n_groups, n_members;
    const gswt_merge_group *groups;
    const gswt_merge_member *members;
} gswt_sort_data;


/* ---- Device-side worker stages: update_lod, selective merging, the four tile orders and the presort-view
 * choice of WangTile (wangtile.rs:476-690,720-1218,1496-1607) as HIP kernels.  The tile MAP itself (update_tile_map,
 * :1671-1781: map shift, tile ids from the RNG, centres / corners / edges through surface_mapping) stays in libgswt_host and
 * is handed over once per build event as gswt_cell[]; every per-sort-event stage then runs on the device and leaves a
 * gswt_sort_data (device-merge form: group descriptions, no CPU lists) that is byte-identical to gswt_wang_sort_tiles'. */

/* One map cell, TileInstance as update_tile_map leaves it (structure.rs:495-509); index = x * map_h + y. */
typedef struct {
    uint32_t tile;             /* tid.1 */
    uint32_t has_corner;
    float tile_offset[3];
    float tile_center[3];
    float to_local[9];         /* column-major Matrix3 */
    float corner_pos[12];      /* corner_data[ci].0 */
    float corner_up[12];       /* corner_data[ci].1 column 2 (surface normal at the corner) */
    float edge_pos[12];        /* edge_data[ei].0 */
    float edge_normal[12];     /* edge_data[ei].1 */
} gswt_cell;                   /* 260 bytes */

/* Per-cell results of update_lod and of selective merging (the TileInstance fields those stages write). */
typedef struct {
    uint32_t lod;
    int32_t transition;        /* 0 None, 1 Spawning, 2 Changing(false), 3 Changing(true) */
    float spawning_factor;
    uint32_t merge;            /* 0 None, 1 MergedFrom (group head), 2 MergedTo */
    uint32_t merged_to;        /* map index of the head when merge == 2 */
} gswt_cell_state;

/* UserData fields the stages read plus the tables preprocess / configure leave behind.  Filled by
 * gswt_wang_worker_config (pointers borrowed from the gswt_wang); gswt_worker_create copies everything to the device. */
typedef struct {
    uint32_t map_w, map_h, half_w, half_h;
    uint32_t n_lod, n_tile, n_view;
    float tile_width;
    uint32_t surface_type, tile_sort_type, merge_type;
    float height_map_scale[3];
    float sphere_radius;
    uint32_t lod_blending, lod_bbox_check;
    float lod_transition_width_ratio, lod_dist_tolerance;
    int32_t merge_tile_dist[2];
    float merge_dot_threshold;
    uint32_t merge_topk;
    uint32_t hm_w, hm_h;
    const float *height_map;          /* hm_w * hm_h, or NULL */
    const float *lod_transition_dist; /* n_lod */
    const float *tile_center;         /* n_tile * 3 */
    const float *tile_aabb;           /* n_tile * 6: lo.xyz, hi.xyz */
    const uint32_t *splat_count;      /* n_lod * n_tile: raw_depth lengths */
    const float *presort_dirs;        /* n_view * 3 */
    const int32_t *neighbors;         /* map_w * map_h * 4 slots (W, N, E, S): neighbour map index << 2 | its slot for us, or -1 */
} gswt_worker_config;

typedef struct gswt_worker gswt_worker;
/* Lives on ctx's device with a stream of its own (the reference's worker is a thread of its own, state.rs:478-561).
 * GSWT_ERR_CAPACITY: maps beyond 32 767 cells (u16 ids), or beyond ~17 700 cells with Edge merging (its LDS tables). */
GSWT_API int gswt_worker_create(gswt_ctx *ctx, const gswt_worker_config *cfg, gswt_worker **out);
GSWT_API void gswt_worker_destroy(gswt_worker *w);
GSWT_API const char *gswt_worker_last_error(const gswt_worker *w);
/* After every build event on the host (update_tile_map): the whole map and its centre coordinate. */
GSWT_API int gswt_worker_set_cells(gswt_worker *w, const gswt_cell *cells, size_t n_cells, const int32_t center_coord[2]);
/* update_lod (wangtile.rs:1496-1607): LOD, transition status and spawning factor of every cell. */
GSWT_API int gswt_worker_update_lod(gswt_worker *w, const float cam_pos[3]);
/* sort_tiles (wangtile.rs:476-690): merge, order, views, records.  Enqueued on the worker's stream. */
GSWT_API int gswt_worker_sort_tiles(gswt_worker *w, const float cam_pos[3], const float view_proj16[16]);
/* Threads: every gswt_worker_* call belongs to ONE thread (the reference's worker thread); gswt_set_draws_from_worker belongs to
 * the ctx's thread.  The two meet in three host-side result sets: gswt_worker_fetch waits for the worker's stream, copies the
 * records, group descriptions and draws of the last sort event into a free set and publishes it; gswt_set_draws_from_worker
 * takes the set published last (GSWT_ERR_STATE if there is none yet).  Neither waits for the other. */
GSWT_API int gswt_worker_fetch(gswt_worker *w);
/* Test hooks (worker thread, blocking).  read_sort fetches if needed; its pointers stay valid for two further fetches. */
GSWT_API int gswt_worker_read_cell_state(gswt_worker *w, gswt_cell_state *out, size_t capacity);
GSWT_API int gswt_worker_read_sort(gswt_worker *w, gswt_sort_data *out);
/* Swap-in (state.rs:521-540 + renderer.rs:466-591 host half): the draws were built on the device (TileUniforms, list
 * selection, cull inputs per record); this runs gswt_set_draws_merge_groups' planning on the published set.  ctx must be the
 * worker's ctx; a failure's text is in gswt_last_error(ctx). */
GSWT_API int gswt_set_draws_from_worker(gswt_ctx *ctx, gswt_worker *w);

GSWT_API int gswt_synchronize(gswt_ctx *ctx);
GSWT_API int gswt_last_timings(const gswt_ctx *ctx, gswt_timings *out);

/* Test / profiling hook: per-splat vertex-stage output of the last gswt_render, one
 * 48-byte record per list entry in draw order (visible, ndc.xy, depth, major.xy,
 * minor.xy, rgba) -- the varyings of vs_main (gswt.wgsl:4-8,412-419). Host pointer. */
GSWT_API int gswt_debug_read_projected(gswt_ctx *ctx, void *out, size_t capacity_entries,
                                       size_t *n_entries);

/* Test hook: k_totals alone -- folds n_super (pair, visible) super-group sums into the frame counters {visible, pairs, -,
 * overflow flag} (64-bit) and the exclusive pair prefix per super-group (u32).  Host pointers. */
GSWT_API int gswt_debug_totals(gswt_ctx *ctx, const uint32_t *pair_sums, const uint32_t *visible_sums, uint32_t n_super,
                               uint32_t pair_cap, unsigned long long counters_out[4], uint32_t *super_excl_out);

/* Test hook: the frame's stable LSD radix sort alone (k_radix_hist / k_radix_supscan / k_radix_scatter) on the low `key_bits` bits
 * of n (key, value) pairs, in place.  Equal keys keep their input order (the order contract of scene.rs:685-695 rests on it).
 * Host pointers. */
GSWT_API int gswt_debug_sort(gswt_ctx *ctx, uint32_t *keys, uint32_t *vals, size_t n, int key_bits);

/* Test hook: the tile-local depth sort of GSWT_ORDER_DEPTH alone (k_tile_depth_sort).  lens[t] = length of screen tile t's slice of the pair
 * list (the slices lie back to back, n = their sum); every slice's vals are sorted by its dkeys, stably, in place (any length: slices of
 * more than 16 384 pairs go through k_tile_depth_sort_xl).  *flagged_out: the frame-level error flag of the kernels (0).  Host pointers. */
GSWT_API int gswt_debug_tile_depth_sort(gswt_ctx *ctx, const uint32_t *lens, size_t n_tiles, uint32_t *vals, const uint32_t *dkeys, size_t n,
                                        int *flagged_out);

/* GSWT_OPT_GRAPH bookkeeping since gswt_create: {frames replayed through hipGraphLaunch, graphs (re)built, kernel nodes updated}. */
GSWT_API int gswt_debug_graph_stats(const gswt_ctx *ctx, unsigned long long out[3]);

/* Merged groups sorted / copied from a retained earlier sort event by gswt_set_draws_merge_groups since gswt_create. */
GSWT_API int gswt_debug_merge_stats(const gswt_ctx *ctx, unsigned long long out[2]);
/* ... and how many of the copied ones came from an event OLDER than the previous one (the lists of the last 9 events stay addressable by
 * (view, ordered member tile ids, transition states): the reference's 1 024-entry LRU of merged lists, wangtile.rs:427,575-593). */
GSWT_API int gswt_debug_merge_stats_deep(const gswt_ctx *ctx, unsigned long long *out);

/* Test / profiling hook: [start, end) of every screen tile in the sorted pair list of the last
 * gswt_render (2 u32 per tile, shard-local tile order). Host pointer. */
GSWT_API int gswt_debug_read_ranges(gswt_ctx *ctx, uint32_t *out, size_t capacity_tiles, size_t *n_tiles);
/* GSWT_ORDER_DEPTH bookkeeping: out[0] = frames enqueued on the tile-local path (k_tile_depth_sort), [1] = on the global depth passes
 * (re-runs count), [2] = the longest screen-tile pair list of the last finished depth-ordered frame. */
GSWT_API int gswt_debug_depth_stats(const gswt_ctx *ctx, unsigned long long out[3]);
/* Device timeline of two frame slots (tests of the frame / gather overlap): out_ms[0] = start of slot `ticket`'s frame kernels, [1] = their
 * end, [2] = end of its gather + re-assembly (NaN if it was not gathered), in ms after the START of slot `ticket_ref`'s frame.  Both frames
 * submitted with GSWT_OPT_TIMING >= 1 and complete (the call synchronises). */
GSWT_API int gswt_debug_frame_times(gswt_ctx *ctx, int ticket_ref, int ticket, float out_ms[3]);

#ifdef __cplusplus
}
#endif
#endif /* GSWT_HIP_H */
