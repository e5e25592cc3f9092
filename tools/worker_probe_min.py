import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gswt_renderer_amd import host, synth, workloads, flypath
from gswt_renderer_amd.pipeline import GSWTPipeline
from gswt_renderer_amd.worker import DeviceWorker
name="c3"; w=workloads.WORKLOADS[name]
verts=synth.make_tileset(n_lod=w["n_lod"], n_tile=16, lod0_count=w["lod0"])
pipe=GSWTPipeline(verts, workloads.user_data_for(name), device_merge=True)
dw=DeviceWorker(pipe.renderer, pipe.wang)
cam=workloads.camera_for(name)
for pos,tgt in flypath.sample(flypath.load("c3"), 20):
    pos=tuple(float(x) for x in pos)
    cu,vp=host.camera_uniforms(pos,tgt,cam["up"],cam["fovy"],cam["near"],cam["far"],w["width"],w["height"])
    if pipe.wang.check_update(pos):
        pipe.wang.build_tiles(pos); dw.build_tiles(pos)
    dw.sort_tiles(pos,vp); dw.cell_state()
print("done")
