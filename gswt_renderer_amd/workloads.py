"""BASELINE.json workload configurations (synthetic, seeded; SURVEY.md 8d / BASELINE.md 3).

Each entry fixes the Wang-tile map, the per-tile splat budget, the render target and the
UserData the reference's GUI would produce.  `c3` is the configuration BASELINE.json's metric
is quoted on (32x32 grid -> nearest reference-valid map 33x33, ~10 M instanced Gaussians,
1920x1080, LOD blending + selective merging on).
"""
from __future__ import annotations

from . import host

# lod_max_dist: the GUI default is 96 * tile_width on a 97x97 map (structure.rs:121-137,198-199).
# For c3 it is 64 * tile_width so that the instanced count stays ~10 M (as BASELINE.json states)
# while the LOD1 / LOD2 rings and their blending bands are present inside the 33x33 map.
WORKLOADS = {
    "c1": dict(desc="1x1 map, ~50k Gaussians, 640x480 (reference CPU-runnable plumbing case)",
               half=(0, 0), lod0=50000, n_lod=3, width=640, height=480,
               # a 1x1 map sits under the camera's own tile: look down at it (the reference default camera looks along +y at z = 5
               # and sees nothing of a single tile)
               camera=dict(pos=(2.0, 0.5, 7.0), target=(2.0, 2.0, 0.0), up=(0.0, 0.0, 1.0), fovy=45.0, near=0.1, far=2400.0),
               user=dict(surface_type=host.SURFACE_NONE, tile_sort_type=host.SORT_DISTANCE, merge_type=host.MERGE_EDGE,
                         lod_max_dist=96.0 * 4.0)),
    "c2": dict(desc="5x5 map (4x4 grid nearest valid), ~1M instanced Gaussians, 1280x720, LOD off",
               half=(2, 2), lod0=62500, n_lod=1, width=1280, height=720,
               camera=dict(pos=(2.0, 2.0, 14.0), target=(2.0, 6.0, 0.0), up=(0.0, 0.0, 1.0), fovy=45.0, near=0.1, far=2400.0),
               user=dict(surface_type=host.SURFACE_NONE, tile_sort_type=host.SORT_GRAPH, merge_type=host.MERGE_EDGE,
                         lod_blending=False, lod_max_dist=96.0 * 4.0)),
    "c3": dict(desc="33x33 map (32x32 grid nearest valid), ~10M instanced Gaussians, 1920x1080, LOD blending + Edge merging",
               half=(16, 16), lod0=9800, n_lod=3, width=1920, height=1080,
               user=dict(surface_type=host.SURFACE_NONE, tile_sort_type=host.SORT_GRAPH, merge_type=host.MERGE_EDGE,
                         lod_blending=True, lod_transition_width_ratio=0.05, merge_topk=100, merge_dot_threshold=0.2,
                         lod_max_dist=64.0 * 4.0)),
    # c3 on the GUI's default surface (HeightMap, random 10x10 map resized to 1024^2, structure.rs:121-137): five bilinear
    # height samples + frame transform per splat in k_project
    "c3h": dict(desc="c3 on the HeightMap surface (GUI default): 33x33 map, ~10M instanced Gaussians, 1920x1080",
                half=(16, 16), lod0=9800, n_lod=3, width=1920, height=1080,
                user=dict(surface_type=host.SURFACE_HEIGHTMAP, height_map_scale=(1.0, 1.0, 0.25), tile_sort_type=host.SORT_GRAPH,
                          merge_type=host.MERGE_EDGE, lod_blending=True, lod_transition_width_ratio=0.05, merge_topk=100,
                          merge_dot_threshold=0.2, lod_max_dist=64.0 * 4.0)),
    # c3 with 4 x larger splats (same counts, same cameras): ~16 x the footprint area, so that the pair count is what
    # SURVEY 8(d) assumed for this configuration (P ~ 10 - 15 M); shows how the compositor scales with overdraw
    "c3d": dict(desc="c3 with 4x larger Gaussians (dense overdraw): 33x33 map, ~10M instanced, 1920x1080",
                half=(16, 16), lod0=9800, n_lod=3, width=1920, height=1080, base_scale=0.08,
                user=dict(surface_type=host.SURFACE_NONE, tile_sort_type=host.SORT_GRAPH, merge_type=host.MERGE_EDGE,
                          lod_blending=True, lod_transition_width_ratio=0.05, merge_topk=100, merge_dot_threshold=0.2,
                          lod_max_dist=64.0 * 4.0)),
    # c3-sized frame on the Sphere surface (gswt.wgsl:600-623; maps are 5 : 2 there): 60 x 24 cells wrapped onto a sphere whose
    # circumference is the map's width, seen whole from outside (sphere mapping per splat = k_project<FULL>; band cull through sphere_cell_box)
    "c3s": dict(desc="60x24 map on the Sphere surface (R = 38.2), ~13M instanced Gaussians, 1920x1080, LOD blending + Edge merging",
                half=(30, 12), lod0=9800, n_lod=3, width=1920, height=1080,
                camera=dict(pos=(20.0, -95.0, 35.0), target=(0.0, 0.0, 0.0), up=(0.0, 0.0, 1.0), fovy=45.0, near=0.1, far=2400.0),
                user=dict(surface_type=host.SURFACE_SPHERE, sphere_radius=38.2, tile_sort_type=host.SORT_GRAPH, merge_type=host.MERGE_EDGE,
                          lod_blending=True, lod_transition_width_ratio=0.05, merge_topk=100, merge_dot_threshold=0.2,
                          lod_max_dist=64.0 * 4.0)),
    "c5": dict(desc="129x129 map (128x128 grid nearest valid), ~100M instanced Gaussians, 3840x2160, full LOD",
               half=(64, 64), lod0=6100, n_lod=3, width=3840, height=2160,
               user=dict(surface_type=host.SURFACE_NONE, tile_sort_type=host.SORT_GRAPH, merge_type=host.MERGE_EDGE,
                         lod_blending=True, lod_transition_width_ratio=0.05, merge_topk=100, merge_dot_threshold=0.2,
                         lod_max_dist=256.0 * 4.0)),
    # small variant for smoke tests
    "tiny": dict(desc="7x7 map, 3 LODs, 320x240", half=(3, 3), lod0=600, n_lod=3, width=320, height=240,
                 user=dict(surface_type=host.SURFACE_NONE, tile_sort_type=host.SORT_GRAPH, merge_type=host.MERGE_EDGE,
                           lod_max_dist=20.0)),
}

DEFAULT_CAMERA = dict(pos=(0.0, 0.0, 5.0), target=(0.0, 1.0, 5.0), up=(0.0, 0.0, 1.0), fovy=45.0, near=0.1, far=2400.0)  # state.rs:114-122


def camera_for(name: str) -> dict:
    """The workload's camera: its own where the reference default would see nothing of the map, else state.rs:114-122."""
    return WORKLOADS[name].get("camera", DEFAULT_CAMERA)


def user_data_for(name: str) -> host.UserData:
    w = WORKLOADS[name]
    return host.user_data(tile_map_half_wh=w["half"], **w["user"])
