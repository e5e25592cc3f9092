"""Device-side worker stages: update_lod, selective merging, the four tile orders, view choice and SortData records
built by gswt_worker_* must equal libgswt_host's (which tests/test_host_parity.py pins bit-exactly to oracle/wangtile_oracle.py)
byte for byte, and the oracle's sort_tiles directly on the golden-style cases."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pipe(half, user_kw, n_lod=3, lod0=600, height_tex=None, seed=3):
    from gswt_renderer_amd import host, synth
    from gswt_renderer_amd.pipeline import GSWTPipeline
    verts = synth.make_tileset(n_lod=n_lod, n_tile=16, lod0_count=lod0, seed_offset=seed)
    user = host.user_data(tile_map_half_wh=half, **user_kw)
    pipe = GSWTPipeline(verts, user, device_merge=True)
    if height_tex is not None:
        pipe.configure(user, height_tex)
    return pipe, verts


def _host_event(pipe, pos, vp, rebuild):
    from gswt_renderer_amd import _lib as L
    w = pipe.wang
    if rebuild:
        w.build_tiles(pos)
    lod_state = w.export_cell_state()
    vpa = np.ascontiguousarray(vp, dtype=np.float32)
    from gswt_renderer_amd.host import SortDataC, _check, _f3, _ptr
    sd = SortDataC()
    _check(w._lib.gswt_wang_sort_tiles(w._h, _f3(pos), _ptr(vpa), C.byref(sd)))
    tiles = C.string_at(C.cast(sd.tiles, C.c_void_p), sd.n_tiles * C.sizeof(L.SortedTile)) if sd.n_tiles else b""
    groups = C.string_at(sd.groups, sd.n_groups * C.sizeof(L.MergeGroup)) if sd.n_groups else b""
    members = C.string_at(sd.members, sd.n_members * C.sizeof(L.MergeMember)) if sd.n_members else b""
    draws = (L.Draw * max(1, sd.n_tiles))()
    _check(w._lib.gswt_renderer_build_draws(C.byref(sd), draws))
    return dict(lod_state=lod_state, state=w.export_cell_state(), tiles=tiles, groups=groups, members=members,
                n=(int(sd.n_tiles), int(sd.n_groups), int(sd.n_members), int(sd.n_merged)), draws=bytes(draws)[:sd.n_tiles * C.sizeof(L.Draw)])


def _tile_fields(buf):
    from gswt_renderer_amd import _lib as L
    n = len(buf) // C.sizeof(L.SortedTile)
    return np.frombuffer(buf, dtype=np.uint32).reshape(n, C.sizeof(L.SortedTile) // 4)


def _compare(pipe, dw, pos, vp, rebuild, tag):
    ref = _host_event(pipe, pos, vp, rebuild)
    if rebuild:
        dw.build_tiles(pos)
    st_lod = dw.cell_state()
    assert np.array_equal(st_lod[:, :3], ref["lod_state"][:, :3]), f"{tag}: update_lod differs in {np.argwhere(st_lod[:, :3] != ref['lod_state'][:, :3])[:5]}"
    dw.sort_tiles(pos, vp)
    st = dw.cell_state()
    bad = np.argwhere(st != ref["state"])
    assert bad.size == 0, f"{tag}: cell state (lod, transition, spawning, merge, merged_to) differs at {bad[:8].tolist()}"
    tiles, groups, members, nt, ng, nm, nmerged = dw.read_sort()
    assert (nt, ng, nm, nmerged) == ref["n"], (tag, (nt, ng, nm, nmerged), ref["n"])
    a, b = _tile_fields(tiles), _tile_fields(ref["tiles"])
    assert np.array_equal(a[:, 6], b[:, 6]), f"{tag}: tile order differs, first at {int(np.argmax(a[:, 6] != b[:, 6]))}"
    assert np.array_equal(a[:, 2], b[:, 2]), f"{tag}: view ids differ"
    assert tiles == ref["tiles"], f"{tag}: records differ in columns {sorted(set(np.argwhere(a != b)[:, 1].tolist()))}"
    assert groups == ref["groups"] and members == ref["members"], tag
    return ref


CAMS = [((4.2, 1.0, 3.0), (5.0, 3.0, 2.5)), ((0.3, -0.4, 2.0), (3.0, 2.0, 0.0)), ((-2.5, 6.1, 4.0), (0.0, 0.0, 0.0)), ((1.0, 1.0, 9.0), (1.2, 1.1, 0.0))]


def _cam(pos, tgt, W=640, H=360):
    from gswt_renderer_amd import host
    return host.camera_uniforms(pos, tgt, (0, 0, 1), 45.0, 0.1, 2400.0, W, H)


@pytest.mark.parametrize("sort_type", [0, 1, 2, 3])
@pytest.mark.parametrize("merge_type", [0, 1, 2])
def test_orders_and_merges_match_host(sort_type, merge_type):
    from gswt_renderer_amd import host
    from gswt_renderer_amd.worker import DeviceWorker
    if sort_type != host.SORT_GRAPH and merge_type != host.MERGE_EDGE:
        pytest.skip("the reference panics without corner data (renderer.rs:476): no draw list to compare")
    pipe, _ = _pipe((5, 4), dict(surface_type=host.SURFACE_NONE, tile_sort_type=sort_type, merge_type=merge_type, lod_blending=True,
                                 lod_transition_width_ratio=0.1, merge_topk=12, merge_dot_threshold=0.6, merge_tile_dist=(1, 4), lod_max_dist=6.0))
    dw = DeviceWorker(pipe.renderer, pipe.wang)
    for k, (pos, tgt) in enumerate(CAMS):
        cu, vp = _cam(pos, tgt)
        _compare(pipe, dw, pos, vp, rebuild=(k % 2 == 0), tag=f"sort {sort_type} merge {merge_type} cam {k}")
    dw.close()


def test_heightmap_and_sphere_surfaces_match_host():
    from gswt_renderer_amd import host
    from gswt_renderer_amd.worker import DeviceWorker
    rng = np.random.default_rng(5)
    tex = rng.random((16, 16), dtype=np.float32)
    pipe, _ = _pipe((4, 4), dict(surface_type=host.SURFACE_HEIGHTMAP, height_map_type=host.HMAP_TEXTURE if hasattr(host, "HMAP_TEXTURE") else 0,
                                 height_map_wh=(32, 32), height_map_scale=(1.0, 1.0, 0.6), tile_sort_type=host.SORT_GRAPH, merge_type=host.MERGE_EDGE,
                                 lod_blending=True, lod_transition_width_ratio=0.1, merge_topk=20, merge_dot_threshold=0.5, lod_max_dist=7.0), height_tex=tex)
    dw = DeviceWorker(pipe.renderer, pipe.wang)
    for k, (pos, tgt) in enumerate(CAMS):
        cu, vp = _cam(pos, tgt)
        _compare(pipe, dw, pos, vp, rebuild=True, tag=f"heightmap cam {k}")
    dw.close()
    pipe, _ = _pipe((5, 2), dict(surface_type=host.SURFACE_SPHERE, sphere_radius=3.0, tile_sort_type=host.SORT_GRAPH, merge_type=host.MERGE_EDGE,
                                 lod_blending=True, lod_transition_width_ratio=0.1, merge_topk=10, merge_dot_threshold=0.5, lod_max_dist=8.0))
    dw = DeviceWorker(pipe.renderer, pipe.wang)
    for k, (pos, tgt) in enumerate([((6.0, 1.0, 2.0), (0, 0, 0)), ((-1.0, 5.5, -3.0), (0, 0, 0)), ((0.5, 0.2, 7.0), (0, 0, 0))]):
        cu, vp = _cam(pos, tgt)
        _compare(pipe, dw, pos, vp, rebuild=(k == 0), tag=f"sphere cam {k}")
    dw.close()


def test_graph_cycles_are_removed_like_petgraph():
    """A steep random height map makes the edge orientations cyclic (the oracle takes 2-10 nodes out per sort event on these
    cameras: remove_node's swap_remove renumbering and the re-run toposort are on the path)."""
    from gswt_renderer_amd import host
    from gswt_renderer_amd.worker import DeviceWorker
    rng = np.random.default_rng(1)
    for half in [(4, 4), (5, 2)]:
        pipe, _ = _pipe(half, dict(surface_type=host.SURFACE_HEIGHTMAP, height_map_scale=(1.0, 1.0, 3.0), height_map_wh=(8, 8), tile_sort_type=host.SORT_GRAPH,
                                   merge_type=host.MERGE_EDGE, merge_topk=30, merge_dot_threshold=0.8, lod_max_dist=8.0), lod0=100)
        dw = DeviceWorker(pipe.renderer, pipe.wang)
        short = 0
        for k in range(12):
            pos = tuple(float(x) for x in rng.uniform((-8, -8, 0.2), (8, 8, 6.0)))
            tgt = tuple(float(x) for x in rng.uniform((-4, -4, 0), (4, 4, 1)))
            cu, vp = _cam(pos, tgt)
            rebuild = pipe.wang.check_update(pos)
            ref = _compare(pipe, dw, pos, vp, rebuild=rebuild, tag=f"steep {half} cam {k}")
            short += ref["n"][0]
        assert short > 0
        dw.close()


def test_large_map_graph_tables_in_global_memory():
    """57x57 cells: the graph tables (50 B per cell) no longer fit the LDS budget and k_w_order_seq runs from global scratch;
    same bytes as the host, flat and on a steep height map (cycles)."""
    from gswt_renderer_amd import host
    from gswt_renderer_amd.worker import DeviceWorker
    for surface, extra in [(host.SURFACE_NONE, {}), (host.SURFACE_HEIGHTMAP, dict(height_map_scale=(1.0, 1.0, 2.0), height_map_wh=(16, 16)))]:
        pipe, _ = _pipe((28, 28), dict(surface_type=surface, tile_sort_type=host.SORT_GRAPH, merge_type=host.MERGE_EDGE, merge_topk=60,
                                       merge_dot_threshold=0.5, lod_max_dist=40.0, **extra), lod0=60)
        dw = DeviceWorker(pipe.renderer, pipe.wang)
        for k, (pos, tgt) in enumerate([((3.0, -2.0, 6.0), (20.0, 30.0, 0.0)), ((-30.0, 12.0, 9.0), (0.0, 0.0, 0.0))]):
            cu, vp = _cam(pos, tgt)
            ref = _compare(pipe, dw, pos, vp, rebuild=True, tag=f"57x57 surface {surface} cam {k}")
            assert ref["n"][0] > 3000
        dw.close()


def test_random_sweep_matches_host():
    """Random maps, parameters and cameras (Graph order + Edge merge, the reference's defaults, and the other modes)."""
    from gswt_renderer_amd import host
    from gswt_renderer_amd.worker import DeviceWorker
    rng = np.random.default_rng(11)
    for case in range(10):
        half = (int(rng.integers(2, 8)), int(rng.integers(2, 8)))
        sort_type = int(rng.choice([3, 3, 0, 1, 2]))
        user = dict(surface_type=host.SURFACE_NONE, tile_sort_type=sort_type, merge_type=host.MERGE_EDGE, lod_blending=bool(rng.integers(0, 2)),
                    lod_bbox_check=bool(rng.integers(0, 2)), lod_transition_width_ratio=float(rng.uniform(0.02, 0.2)),
                    merge_topk=int(rng.integers(1, 60)), merge_dot_threshold=float(rng.uniform(0.1, 0.9)), lod_max_dist=float(rng.uniform(3.0, 12.0)))
        pipe, _ = _pipe(half, user, seed=case)
        dw = DeviceWorker(pipe.renderer, pipe.wang)
        for k in range(4):
            pos = tuple(float(x) for x in rng.uniform((-6, -6, 0.5), (6, 6, 8.0)))
            tgt = tuple(float(x) for x in rng.uniform((-4, -4, 0.0), (4, 4, 1.0)))
            cu, vp = _cam(pos, tgt)
            _compare(pipe, dw, pos, vp, rebuild=(k != 2), tag=f"case {case} {half} sort {sort_type} cam {k}")
        dw.close()


def test_c3_sort_event_matches_host_and_oracle_image():
    """BASELINE config c3 (33x33 map, Graph order, Edge merge, LOD blending): the device-built sort event equals the host's,
    and a frame rendered from it (gswt_set_draws_from_worker) equals the frame rendered from the host's draw list bit for bit."""
    from gswt_renderer_amd import host, synth, workloads
    from gswt_renderer_amd.pipeline import GSWTPipeline
    from gswt_renderer_amd.worker import DeviceWorker
    w = workloads.WORKLOADS["c3"]
    verts = synth.make_tileset(n_lod=w["n_lod"], n_tile=16, lod0_count=w["lod0"])
    pipe = GSWTPipeline(verts, host.user_data(tile_map_half_wh=w["half"], **w["user"]), device_merge=True)
    dw = DeviceWorker(pipe.renderer, pipe.wang)
    W, H = w["width"], w["height"]
    cam = workloads.camera_for("c3")
    pos = cam["pos"]
    cu, vp = host.camera_uniforms(cam["pos"], cam["target"], cam["up"], cam["fovy"], cam["near"], cam["far"], W, H)
    ref = _compare(pipe, dw, pos, vp, rebuild=True, tag="c3")
    assert ref["n"][0] > 900 and ref["n"][1] > 0
    pipe.update(pos, vp, force_sort=True)
    img_host = pipe.render(cu, W, H)
    dw.swap_in()
    img_dev = pipe.render(cu, W, H)
    assert np.array_equal(img_host, img_dev)
    dw.close()


def test_device_sort_event_matches_oracle_directly():
    """The device-built draw list against oracle/wangtile_oracle.py itself (order, views, classes, merged groups)."""
    from gswt_renderer_amd import _lib as L
    from gswt_renderer_amd import host, synth
    from gswt_renderer_amd.pipeline import GSWTPipeline
    from gswt_renderer_amd.worker import DeviceWorker
    from oracle import gswt_oracle as orc
    from oracle import wangtile_oracle as wo
    for sort_type, surface in [(3, 0), (0, 0), (2, 0), (3, 1)]:
        cfg = dict(tile_map_half_wh=(4, 3), surface_type=surface, tile_sort_type=sort_type, merge_type=2, lod_blending=True, lod_transition_width_ratio=0.1,
                   merge_topk=15, merge_dot_threshold=0.6, lod_max_dist=6.0, height_map_scale=(1.0, 1.0, 0.5))
        verts = synth.make_tileset(n_lod=3, n_tile=16, lod0_count=300)
        pipe = GSWTPipeline(verts, host.user_data(**cfg), device_merge=True)
        dw = DeviceWorker(pipe.renderer, pipe.wang)
        pp = orc.preprocess([[orc.scene_load(v) for v in lod] for lod in verts])
        ow = wo.WangTile(pp)
        ow.configure(wo.UserData(**cfg))
        with np.errstate(all="ignore"):
            for pos, tgt in CAMS[:3]:
                cu, vp = _cam(pos, tgt)
                if pipe.wang.check_update(pos):
                    assert ow.check_update(pos)
                    pipe.wang.build_tiles(pos)
                    ow.build_tiles(pos)
                    dw.build_tiles(pos)
                os_ = ow.sort_tiles(pos, vp)
                dw.sort_tiles(pos, vp)
                tiles, groups, members, nt, ng, nm, nmerged = dw.read_sort()
                assert nt == len(os_["tile_instance_vec"])
                recs = [L.SortedTile.from_buffer_copy(tiles[i * C.sizeof(L.SortedTile):(i + 1) * C.sizeof(L.SortedTile)]) for i in range(nt)]
                mem = np.frombuffer(members, dtype=np.uint32).reshape(nm, 4) if nm else np.zeros((0, 4), np.uint32)
                grp = np.frombuffer(groups, dtype=np.uint32).reshape(ng, 4) if ng else np.zeros((0, 4), np.uint32)
                for tc, to, (key, val) in zip(recs, os_["tile_instance_vec"], os_["render_data_vec"]):
                    assert tc.map_index == to.map_index
                    assert (tc.lod, tc.tile, tc.view_id) == (to.tid[0], to.tid[1], to.view_id)
                    st = to.transition_status
                    code = 0 if st[0] == "none" else (1 if st[0] == "spawning" else (3 if st[1] else 2))
                    assert tc.transition == code, (tc.transition, st)
                    assert tc.key_len == len(key[1]) and bool(tc.merged) == (val is not None)
                    if val is not None:
                        assert tc.merged_count == val["splat_count"] and tc.single_lod_id == val["single_lod_id"]
                        g = grp[tc.merged_group]
                        assert g[0] == tc.view_id and g[2] == len(val["merge_from_vec"])
                        assert mem[g[1]:g[1] + g[2], 0].tolist() == [int(x) for x in val["merge_from_vec"]]
        dw.close()


def test_worker_argument_errors():
    from gswt_renderer_amd import host
    from gswt_renderer_amd.worker import DeviceWorker
    pipe, _ = _pipe((2, 2), dict(surface_type=host.SURFACE_NONE, tile_sort_type=host.SORT_GRAPH, merge_type=host.MERGE_EDGE))
    dw = DeviceWorker(pipe.renderer, pipe.wang)
    cu, vp = _cam(*CAMS[0])
    with pytest.raises(RuntimeError, match="before build_tiles"):
        dw.sort_tiles(CAMS[0][0], vp)
    with pytest.raises(RuntimeError, match="before gswt_worker_set_cells"):
        dw.update_lod(CAMS[0][0])
    with pytest.raises(RuntimeError):
        dw.read_sort()
    dw.close()
