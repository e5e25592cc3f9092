/*
 * gswt_host.h -- C ABI of libgswt_host.so: the host (CPU) side of the GSWT hot path,
 * i.e. the C++ mirror of the reference's `scene::Scene` loader and `wangtile::WangTile`
 * worker (same names, argument meaning and error behaviour), plus the host half of
 * `GSWTRenderer::render` that turns a SortData into the draw list libgswt_hip.so consumes.
 *
 * The reference is Rust; this container has no Rust toolchain, so the host side above the
 * device ABI is C++17 (see INTEGRATION.md for the `extern "C"` block a Rust caller adds).
 * file:line citations are into zengyf131/gswt_renderer.  Reference panics
 * (unwrap/expect/assert!) become negative status codes + gswt_host_last_error().
 *
 * Threading mirrors the reference: a gswt_wang is owned by exactly one (worker) thread.
 * Pointers returned by accessor calls stay valid until the next mutating call on the same
 * object (configure / build_tiles / sort_tiles) or its destruction.
 */
#ifndef GSWT_HOST_H
#define GSWT_HOST_H

#include <stddef.h>
#include <stdint.h>

#include "gswt_hip.h" /* gswt_draw, gswt_base_list, uniform block layouts, status codes */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gswt_tileset gswt_tileset; /* Vec<Vec<Scene>> [n_lod][n_tile]  (scene.rs:1091)   */
typedef struct gswt_wang gswt_wang;       /* wangtile::WangTile               (wangtile.rs:18)  */

#define GSWT_ERR_IO (-5)    /* unreadable / malformed file (reference: panic in load_scene_zip) */

GSWT_API const char *gswt_host_last_error(void); /* thread-local text of the last failure */

/* ---- scene::Scene / tile-zip loader ------------------------------------------------- */
GSWT_API int gswt_tileset_create(int n_lod, int n_tile, gswt_tileset **out);
GSWT_API void gswt_tileset_destroy(gswt_tileset *ts);
/* Scene::parse_file_header + Scene::load (scene.rs:72-212): binary-LE 3DGS PLY bytes. */
GSWT_API int gswt_tileset_set_ply(gswt_tileset *ts, int lod, int tile, const uint8_t *bytes, size_t len);
/* 62-float vertex records without a PLY header (same packing, scene.rs:115-212). */
GSWT_API int gswt_tileset_set_vertices(gswt_tileset *ts, int lod, int tile, const float *verts62, size_t n);
/* Already packed 32 B/splat rows (the `.splat` layout, scene.rs:920-922). */
GSWT_API int gswt_tileset_set_rows(gswt_tileset *ts, int lod, int tile, const uint8_t *rows32, size_t n);
/* load_scene_zip (scene.rs:1030-1141): GSWT tile-zip, entries matched by lod(\d+)_tile_(\d+). */
GSWT_API int gswt_load_scene_zip(const char *path, gswt_tileset **out);
GSWT_API int gswt_load_scene_zip_mem(const uint8_t *bytes, size_t len, gswt_tileset **out);
GSWT_API int gswt_tileset_dims(const gswt_tileset *ts, int *n_lod, int *n_tile);
GSWT_API size_t gswt_tileset_splat_count(const gswt_tileset *ts, int lod, int tile);
GSWT_API const uint8_t *gswt_tileset_rows(const gswt_tileset *ts, int lod, int tile); /* Scene.buffer */
/* Scene::generate_texture (scene.rs:306-411) on arbitrary rows: tex = 8 u32 per splat. */
GSWT_API int gswt_generate_texture(const uint8_t *rows32, size_t n, uint32_t *tex_out);
/* Scene::sort_raw_depth_vec (scene.rs:655-698) on one concatenated depth array. */
GSWT_API int gswt_sort_raw_depth(const int32_t *depths, size_t n, uint32_t *order_out);
/* utils::pack_half_2x16 (utils.rs:66-73) */
GSWT_API uint32_t gswt_pack_half_2x16(float x, float y);

/* ---- camera::Camera (camera.rs:90-131,169-188) ---------------------------------------- */
/* Camera::new_perspective + CameraUniforms::from_camera; also returns view_proj (16 f32). */
GSWT_API int gswt_camera_uniforms_from_camera(const float pos[3], const float target[3], const float up[3], float fovy_deg,
                                  float z_near, float z_far, int width, int height,
                                  gswt_camera_uniforms *out, float *view_proj16);

/* ---- wangtile::WangTile --------------------------------------------------------------- */
/* UserData (structure.rs:15-65), the fields the worker reads. */
typedef struct {
    uint32_t tile_map_half_wh[2];
    uint32_t center_option;
    float update_distance2;
    float tile_width;
    uint32_t tile_sort_type;   /* 0 Distance 1 Viewport 2 Object 3 Graph (structure.rs:451-457) */
    uint32_t surface_type;     /* 0 None 1 HeightMap 2 Sphere            (structure.rs:435-440) */
    uint32_t height_map_wh[2];
    uint32_t height_map_type;  /* 0 Texture 1 Random 2 SlopeX 3 SlopeY 4 DualSlope (:442-449) */
    float height_map_scale[3];
    const float *height_tex;   /* optional, height_tex_wh[0] * height_tex_wh[1] floats */
    uint32_t height_tex_wh[2];
    float sphere_radius;
    float lod_max_dist;
    uint32_t lod_blending;
    float lod_transition_width_ratio;
    uint32_t lod_bbox_check;
    float lod_dist_tolerance;
    uint32_t merge_type;       /* 0 None 1 Axis 2 Edge (structure.rs:459-464) */
    int32_t merge_tile_dist[2];
    float merge_dot_threshold;
    uint32_t merge_topk;
    uint32_t use_cache;
    uint32_t cache_size;
    uint32_t reset_rng;
    uint32_t always_sort;
} gswt_user_data;

/* What WangTile::configure adds to UserData (wangtile.rs:349-432). */
typedef struct {
    uint32_t tile_map_wh[2];
    uint32_t height_map_wh[2];
    const float *height_map;   /* height_map_wh[0] * height_map_wh[1] */
    float lod_transition_dist[16];
    uint32_t n_lod, n_tile, n_view;
} gswt_configured;

/* SceneData (structure.rs:466-474) */
typedef struct {
    uint32_t scene_id;
    uint64_t splat_count;
    uint64_t blending_splat_count;
    int32_t center_coord[2];
    uint64_t lod_splat_count[16];
    uint64_t lod_instance_count[16];
} gswt_scene_data;

/* gswt_sorted_tile / gswt_sort_data (SortData, structure.rs:488-509,670-694) are declared in gswt_hip.h: the device-side
 * worker stages of libgswt_hip produce the same records. */

/* PreloadData (structure.rs:731-736) */
typedef struct {
    const uint32_t *tex_data;         /* tile_splats_merged.tex_data, 8 u32 per splat */
    size_t n_splats;
    int n_lod, n_tile, n_view;
    const gswt_base_list *lists;      /* [n_lod][n_tile][n_view] */
} gswt_preload;

/* WangTile::new (wangtile.rs:41-69): takes ownership of the tile set (also on failure: the tile
 * set is destroyed either way), runs preprocess. */
GSWT_API int gswt_wang_new(gswt_tileset *ts, gswt_wang **out);
GSWT_API void gswt_wang_destroy(gswt_wang *w);
/* WangTile::preload (wangtile.rs:340-347) */
GSWT_API int gswt_wang_preload(gswt_wang *w, gswt_preload *out);
/* preprocess outputs used by tests: per-tile centre / aabb, per-LOD avg scale, raw depths */
GSWT_API int gswt_wang_tile_base(const gswt_wang *w, int tile, float center[3], float aabb[6]);
GSWT_API int gswt_wang_lod_avg_scale(const gswt_wang *w, float *out, int cap);
GSWT_API const int32_t *gswt_wang_raw_depth(const gswt_wang *w, int lod, int tile, int view, size_t *n);
GSWT_API int gswt_wang_merge_offset(const gswt_wang *w, int lod, int tile, uint32_t *out);
/* WangTile::configure (wangtile.rs:349-432) */
GSWT_API int gswt_wang_configure(gswt_wang *w, const gswt_user_data *user, gswt_configured *out);
/* WangTile::check_update (wangtile.rs:692-699) */
GSWT_API int gswt_wang_check_update(const gswt_wang *w, const float cam_pos[3]);
/* WangTile::build_tiles (wangtile.rs:434-474) */
GSWT_API int gswt_wang_build_tiles(gswt_wang *w, const float cam_pos[3], gswt_scene_data *out);
/* WangTile::sort_tiles (wangtile.rs:476-690) */
GSWT_API int gswt_wang_sort_tiles(gswt_wang *w, const float cam_pos[3], const float view_proj16[16],
                                  gswt_sort_data *out);
/* Device-merge mode: sort_tiles only describes the merged groups (members, view) and leaves the
 * counting sort of their splats to libgswt_hip (gswt_set_draws_merge_groups); the CPU lists and the LRU
 * cache (wangtile.rs:575-593,642-675) are skipped. */
GSWT_API int gswt_wang_set_device_merge(gswt_wang *w, int enable);
/* Raw-depth tables for gswt_upload_raw_depth: ptrs[(lod*n_tile+tile)*n_view+view], counts / merge_offset [lod*n_tile+tile]. */
GSWT_API int gswt_wang_raw_depth_tables(gswt_wang *w, const int32_t *const **ptrs, const uint32_t **counts,
                                        const uint32_t **merge_offset);

/* Hand-over to the device-side worker stages (gswt_worker_* in gswt_hip.h).  worker_config: UserData fields + tables,
 * pointers valid until the next gswt_wang_configure / destroy.  export_cells: the tile map after build_tiles;
 * export_cell_state: what update_lod and the last sort_tiles' merging left in every cell (for parity tests). */
GSWT_API int gswt_wang_worker_config(gswt_wang *w, gswt_worker_config *out);
GSWT_API int gswt_wang_export_cells(const gswt_wang *w, gswt_cell *out, size_t capacity, int32_t center_coord[2]);
GSWT_API int gswt_wang_export_cell_state(const gswt_wang *w, gswt_cell_state *out, size_t capacity);

/* Inspect / override the tile-id map (tile ids come from an unpinned RNG in the reference; parity
 * fixtures pass them explicitly).  ids: tile_map_wh[0] * tile_map_wh[1], index = x * h + y. */
GSWT_API int gswt_wang_get_tile_ids(const gswt_wang *w, uint32_t *ids, size_t cap);
GSWT_API int gswt_wang_set_tile_ids(gswt_wang *w, const uint32_t *ids, size_t n);

/* Host half of GSWTRenderer::render (renderer.rs:466-591): SortData -> gswt_draw[].  The CPU
 * viewport cull (:472-494) and lod_enable skip (:495) are carried as per-draw inputs and applied
 * on the device.  draws_out must hold sort->n_tiles elements. */
GSWT_API int gswt_renderer_build_draws(const gswt_sort_data *sort, gswt_draw *draws_out);
/* SceneUniforms::from_data (renderer.rs:631-672) */
GSWT_API int gswt_scene_uniforms_from_data(const gswt_user_data *user, const gswt_configured *conf,
                                 const gswt_scene_data *scene, float splat_scale, const float scene_scale[3],
                                 float height_map_scale_v, gswt_scene_uniforms *out);

#ifdef __cplusplus
}
#endif
#endif /* GSWT_HOST_H */
