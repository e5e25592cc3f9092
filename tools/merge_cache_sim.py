#!/usr/bin/env python3
"""How many merged groups of the fly path's sort events would a cache of group lists save?  CPU only (libgswt_host): replays the fly
path (two laps of 240 cameras, every `every`-th one a sort event), keys every merged group as gswt_set_draws_merge_groups does -- the
reference's cache key, wangtile.rs:575-593: (view, ordered member (lod, tile, other lod)) -- and counts the groups NOT found in (a) the
last K events (what the retained draw sets give: K <= 9), (b) an LRU of N groups (the reference keeps 1 024).
Usage: python tools/merge_cache_sim.py [workload] [every]      numbers quoted in DESIGN.md section 7"""
import sys, time; sys.path.insert(0,'/root/repo')
import numpy as np
import bench
from collections import OrderedDict
from gswt_renderer_amd import flypath, host, workloads
name = sys.argv[1] if len(sys.argv)>1 else "c3"
every = int(sys.argv[2]) if len(sys.argv)>2 else 1
w, wang, cu0, vp0, sort0 = bench.build_workload(name)
W,H=w["width"],w["height"]; cam0=workloads.camera_for(name)
wang.set_device_merge(True)
cams=[]
for pos,tgt in flypath.sample(flypath.load("c3" if name!="c5" else ("c5" if __import__('os').path.exists('/root/repo/gswt_renderer_amd/flypaths/c5.json') else "c3")), 240):
    cams.append((tuple(float(x) for x in pos),)+host.camera_uniforms(pos,tgt,cam0["up"],cam0["fovy"],cam0["near"],cam0["far"],W,H))
events=[]
prev_vp=None
for lap in range(2):
  for k,(pos,cu,vp) in enumerate(cams):
    if k % every: continue
    vpn=np.asarray(vp,dtype=np.float32)
    rebuilt=False
    if wang.check_update(pos):
        wang.build_tiles(pos); rebuilt=True
    moved = prev_vp is None or float(np.abs(vpn-prev_vp).sum())>=0.01
    if not (rebuilt or moved): continue
    prev_vp=vpn.copy()
    draws,nd,groups,ng,members,nm = wang.sort_tiles_raw(pos,vp)
    keys=[]
    for g in range(ng):
        G=groups[g]
        key=(int(G.view_id),)+tuple((int(members[G.first_member+m].lod),int(members[G.first_member+m].tile),int(members[G.first_member+m].other_lod)) for m in range(G.n_members))
        keys.append(key)
    events.append(keys)
print("events", len(events), "groups per event", np.mean([len(e) for e in events]))
def sim_hist(K):
    miss=tot=0
    hist=[]
    for e in events:
        have=set().union(*hist[-K:]) if hist else set()
        for k in e:
            tot+=1; miss+= k not in have
        hist.append(set(e))
    return miss/tot
def sim_lru(cap):
    miss=tot=0
    lru=OrderedDict()
    for e in events:
        for k in e:
            tot+=1
            if k in lru: lru.move_to_end(k)
            else:
                miss+=1; lru[k]=1
                if len(lru)>cap: lru.popitem(last=False)
    return miss/tot, len(lru)
for K in (1,2,4,9,16,32,64): print("history", K, "events: sorted share", round(sim_hist(K),4))
for cap in (256,1024,4096,16384): print("LRU", cap, "groups:", sim_lru(cap))
