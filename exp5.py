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
culled=sum(1 for d in sort.draws if d.cull_enable)
print('draws',len(sort.draws))
for flags,name in ((0,'full'),(8,'no stores'),(16,'stop after frustum cull')):
    r.set_option(L.GSWT_OPT_DEBUG_FLAGS, flags)
    ts=[]
    for i in range(12):
        r.render(cu,su,W,H,transmittance_eps=1e-5,out_device_ptr=out.data_ptr()); ts.append(r.timings())
    print('%-26s'%name,{k:round(float(np.mean([t[k] for t in ts[2:]])),4) for k in ('ms_project','ms_total')}, ts[-1]['n_visible'], ts[-1]['n_pairs'])
