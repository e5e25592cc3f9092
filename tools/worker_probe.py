"""Sort-event cost of the device-side worker stages beside libgswt_host's, per stage where the device allows it.
Usage: python tools/worker_probe.py [workload] [events]   (run under rocprofv3 --kernel-trace --stats for per-kernel times)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gswt_renderer_amd import flypath, host, synth, workloads  # noqa: E402
from gswt_renderer_amd.pipeline import GSWTPipeline  # noqa: E402
from gswt_renderer_amd.worker import DeviceWorker  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c3"
    n_ev = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    w = workloads.WORKLOADS[name]
    verts = synth.make_tileset(n_lod=w["n_lod"], n_tile=16, lod0_count=w["lod0"])
    pipe = GSWTPipeline(verts, workloads.user_data_for(name), device_merge=True)
    dw = DeviceWorker(pipe.renderer, pipe.wang)
    W, H = w["width"], w["height"]
    cam = workloads.camera_for(name)
    cams = flypath.sample(flypath.load("c3"), n_ev)
    t_host_build, t_host_sort, t_dev_build, t_dev_sort, t_dev_read, t_swap = [], [], [], [], [], []
    for k in range(n_ev):
        pos, tgt = (tuple(float(x) for x in v) for v in cams[k])
        cu, vp = host.camera_uniforms(pos, tgt, cam["up"], cam["fovy"], cam["near"], cam["far"], W, H)
        rebuild = pipe.wang.check_update(pos)
        if rebuild:
            t0 = time.perf_counter(); pipe.wang.build_tiles(pos); t_host_build.append(time.perf_counter() - t0)
            t0 = time.perf_counter(); dw.build_tiles(pos); dw.cell_state(); t_dev_build.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); s = pipe.wang.sort_tiles_raw(pos, vp); t_host_sort.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); dw.sort_tiles(pos, vp)
        tiles, groups, members, nt, ng, nm, nmerged = dw.read_sort(); t_dev_sort.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); dw.swap_in(); t_swap.append(time.perf_counter() - t0)
        assert nt == s[1] and ng == s[3] and nm == s[5], (nt, s[1], ng, s[3])
    ms = lambda a: 1e3 * float(np.mean(a[1:] if len(a) > 1 else a)) if a else float("nan")
    print(f"{name}: {n_ev} sort events, {len(t_host_build)} build events, {nt} tiles, {ng} groups")
    print(f"  host  build_tiles {ms(t_host_build):.3f} ms   sort_tiles {ms(t_host_sort):.3f} ms")
    print(f"  device map upload + update_lod + readback {ms(t_dev_build):.3f} ms   sort_tiles + record readback {ms(t_dev_sort):.3f} ms   swap-in from the worker {ms(t_swap):.3f} ms")


if __name__ == "__main__":
    main()
