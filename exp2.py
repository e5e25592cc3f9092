import sys; sys.path.insert(0,'.')
import numpy as np, torch, bench
from gswt_renderer_amd.renderer import GSWTRenderer
from gswt_renderer_amd import _lib as L
w,wang,cu,vp,sort=bench.build_workload('c3')
W,H=w['width'],w['height']
r=GSWTRenderer(0); wang.upload_to(r); r.configure(None)
r.set_draws(sort.draws, sort.merged_gs_index, sort.merged_map_id, sort.merged_lod_id)
su=wang.scene_uniforms()
out=torch.empty((H,W,4),dtype=torch.float32,device='cuda')
for seg in (1024, 512):
  for eps in (1e-5, 0.0):
    r.set_option(L.GSWT_OPT_SEGMENT, seg)
    ts=[]
    for i in range(12):
        r.render(cu,su,W,H,transmittance_eps=eps,out_device_ptr=out.data_ptr()); ts.append(r.timings())
    print('seg',seg,'eps',eps,{k:round(float(np.mean([t[k] for t in ts[2:]])),4) for k in ('ms_composite','ms_total')})
