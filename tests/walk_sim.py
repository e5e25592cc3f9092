#!/usr/bin/env python3
"""CPU simulation of the compositor's walk on a frame's real pixel boxes (analysis helper of round 3; lives under tests/ because it
drives the CPU oracle's vertex stage -- checker-side only, never the product).  For a workload's default camera it rebuilds every
(splat, 16x16 tile) pair in composite order and counts, per lane geometry (sub-block size, lists walked in lock step per wave pass, pairs
per batch, pairs per work item), the wave-steps of the walk; `decouple` prices how much of a workgroup's per-batch critical path is
statistical (waves allowed to run `lead` batches apart) and how much is spatial.  Usage: python tests/walk_sim.py [workload]
Numbers quoted in DESIGN.md section 6 / 10 and profiles/r03_composite_variants.txt come from `python tests/walk_sim.py c3`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from oracle import gswt_oracle as orc
name = sys.argv[1] if len(sys.argv)>1 else "c3"
w, wang, cu, vp, sort = bench.build_workload(name)
W,H = w["width"], w["height"]
su = wang.scene_uniforms()
hm = wang.height_map() if int(wang.user.surface_type)==1 else None
tex, draws = bench.oracle_draws(wang, sort, vp)
ocu = orc.Camera176.from_buffer_copy(bytes(cu)); osu = orc.Scene160.from_buffer_copy(bytes(su))
v = orc.project_draws(ocu, osu, tex, draws, height_map=hm)
# composite order = reverse of draw order (front-to-back): reverse whole array
v = v[::-1]
vis = v["visible"]==1
v = v[vis]
f32=np.float32
ndc=v["ndc"].astype(f32); mj=v["major"].astype(f32); mn=v["minor"].astype(f32)
cx=(f32(0.5)*ndc[:,0]+f32(0.5))*f32(W); cy=(f32(-0.5)*ndc[:,1]+f32(0.5))*f32(H)
hs=f32(0.5*su.splat_scale)
ux=hs*mj[:,0]; uy=-(hs*mj[:,1]); wx=hs*mn[:,0]; wy=-(hs*mn[:,1])
uu=ux*ux+uy*uy; ww=wx*wx+wy*wy
ok=(uu>0)&(ww>0)&np.isfinite(uu)&np.isfinite(ww)&(v["depth"]<1.0)
cx,cy,ux,uy,wx,wy=[a[ok] for a in (cx,cy,ux,uy,wx,wy)]
hx=2*np.sqrt(wx*wx+ux*ux)*1.00001+0.001; hy=2*np.sqrt(wy*wy+uy*uy)*1.00001+0.001
x0=np.ceil(cx-hx-0.5); x1=np.floor(cx+hx-0.5); y0=np.ceil(cy-hy-0.5); y1=np.floor(cy+hy-0.5)
m=(x1>=x0)&(y1>=y0)&(x1>=0)&(y1>=0)&(x0<=W-1)&(y0<=H-1)
x0,x1,y0,y1=[np.clip(a[m],0,lim).astype(np.int64) for a,lim in ((x0,W-1),(x1,W-1),(y0,H-1),(y1,H-1))]
n=len(x0); print("visible with pixels", n)
tx0,tx1,ty0,ty1=x0>>4,x1>>4,y0>>4,y1>>4
cnt=(tx1-tx0+1)*(ty1-ty0+1)
P=int(cnt.sum()); print("pairs", P)
# expand pairs
sid=np.repeat(np.arange(n),cnt)
off=np.arange(P)-np.repeat(np.cumsum(cnt)-cnt,cnt)
wdt=np.repeat(tx1-tx0+1,cnt)
ptx=np.repeat(tx0,cnt)+off%wdt; pty=np.repeat(ty0,cnt)+off//wdt
tiles_x=(W+15)//16
tile=pty*tiles_x+ptx
# tile-local pixel box of the pair (clamped 0..15)
lx0=np.clip(x0[sid]-ptx*16,0,15); lx1=np.clip(x1[sid]-ptx*16,0,15); ly0=np.clip(y0[sid]-pty*16,0,15); ly1=np.clip(y1[sid]-pty*16,0,15)
# stable sort by tile keeps composite order
o=np.argsort(tile,kind="stable")
tile=tile[o]; lx0,lx1,ly0,ly1=lx0[o],lx1[o],ly0[o],ly1[o]
# position within tile
first=np.r_[0,np.flatnonzero(np.diff(tile))+1]
tstart=np.repeat(first,np.diff(np.r_[first,P]))
pos=np.arange(P)-tstart
def simulate(bw,bh,groups_per_wave,batch,seg, label, row_major_assign=True):
    # sub-blocks bw x bh; nbx=16/bw, nby=16/bh; hit matrix P x (nbx*nby)
    nbx,nby=16//bw,16//bh
    bx=np.arange(nbx); by=np.arange(nby)
    hxm=(lx0[:,None]<=bx[None,:]*bw+bw-1)&(lx1[:,None]>=bx[None,:]*bw)   # P x nbx
    hym=(ly0[:,None]<=by[None,:]*bh+bh-1)&(ly1[:,None]>=by[None,:]*bh)   # P x nby
    # batch id: (tile, segment, batch within segment)
    bid_local=(pos%seg)//batch + (pos//seg)*((seg+batch-1)//batch)
    key=tile*100000+bid_local
    bfirst=np.r_[0,np.flatnonzero(np.diff(key))+1]
    # counts per batch per sub-block: loop over sub-blocks
    nb=len(bfirst)
    counts=np.zeros((nb,nby,nbx),dtype=np.int32)
    for j in range(nby):
        for i in range(nbx):
            h=(hxm[:,i]&hym[:,j]).astype(np.int32)
            counts[:,j,i]=np.add.reduceat(h,bfirst)
    entries=int(counts.sum())
    # waves: groups_per_wave sub-blocks walk in parallel; a wave pass takes groups in row-major order
    flat=counts.reshape(nb,-1)
    npass=flat.shape[1]//groups_per_wave
    steps=0
    for p in range(npass):
        steps+=int(flat[:,p*groups_per_wave:(p+1)*groups_per_wave].max(axis=1).sum())
    # even padding
    steps_even=0
    for p in range(npass):
        mx=flat[:,p*groups_per_wave:(p+1)*groups_per_wave].max(axis=1)
        steps_even+=int(((mx+1)&~1).sum())
    print(f"{label}: sub-block {bw}x{bh}, {groups_per_wave} lists/wave-pass, batch {batch}, seg {seg}: entries {entries}, wave-pass-steps {steps} (even-padded {steps_even}), batches {nb}, ideal {entries/groups_per_wave:.0f}")
    return steps
t0=time.time()
simulate(4,4,4,256,1536,"current (16-lane groups, 1px)")
simulate(4,4,4,128,1536,"current, batch 128")
simulate(4,4,8,256,1536,"2px: 8 lists (16x8 half tile per pass)")
simulate(4,4,8,128,1536,"2px: 8 lists, batch 128")
simulate(4,4,8,128,512,"2px: 8 lists, batch 128, seg 512")
simulate(4,4,16,256,1536,"4px: 16 lists (whole tile)")
simulate(4,4,16,128,1536,"4px: 16 lists, batch 128")
simulate(8,4,4,256,1536,"2px in 16-lane groups: 8x4 sub-blocks, 4 lists (16x8... 2 passes)")
simulate(8,4,8,256,1536,"4px-ish: 8x4 sub-blocks, 8 lists one pass")
simulate(8,8,4,256,1536,"4px in 16-lane groups: 8x8 quadrants, 4 lists one pass")
simulate(4,2,8,256,1536,"1px 8-lane groups 4x2 (r2 sim)")
simulate(8,2,4,256,1536,"1px 16-lane 8x2 sub-blocks")
simulate(8,2,8,256,1536,"2px 16-lane 8x2... 8 lists")
print("time", time.time()-t0)

# ---- decoupling potential: per tile-segment, critical path of the workgroup's walk
def decouple(batch, seg, lead=None):
    bw=bh=4
    nbx=nby=4
    bx=np.arange(nbx); by=np.arange(nby)
    hxm=(lx0[:,None]<=bx[None,:]*bw+bw-1)&(lx1[:,None]>=bx[None,:]*bw)
    hym=(ly0[:,None]<=by[None,:]*bh+bh-1)&(ly1[:,None]>=by[None,:]*bh)
    nbat=(seg+batch-1)//batch
    bid_local=(pos%seg)//batch + (pos//seg)*nbat
    key=tile*100000+bid_local
    bfirst=np.r_[0,np.flatnonzero(np.diff(key))+1]
    nb=len(bfirst)
    counts=np.zeros((nb,nby,nbx),dtype=np.int64)
    for j in range(nby):
        for i in range(nbx):
            counts[:,j,i]=np.add.reduceat((hxm[:,i]&hym[:,j]).astype(np.int64),bfirst)
    wave=counts.max(axis=2)               # nb x 4 strips: steps of each wave per batch
    item=key[bfirst]//nbat               # item id (tile, segment): tile*100000//nbat ... use (tile, seg index)
    item=tile[bfirst]*1000+ (bid_local[bfirst]//nbat)
    A=int(wave.max(axis=1).sum())
    # fully decoupled waves: per item, max over waves of the sum over batches
    order=np.argsort(item,kind="stable")
    it=item[order]; wv=wave[order]
    f=np.r_[0,np.flatnonzero(np.diff(it))+1]
    sums=np.add.reduceat(wv,f,axis=0)
    B=int(sums.max(axis=1).sum())
    C=int(wave.sum())//4
    # limited lead: simulate per item with a window of `lead` batches
    Ld=None
    if lead is not None:
        tot=0
        ends=np.r_[f[1:],len(it)]
        for a,b in zip(f,ends):
            w=wv[a:b]          # batches x 4
            n=b-a
            fin=np.zeros((n,4))   # finish time of batch k for wave w
            for k in range(n):
                for q in range(4):
                    start=fin[k-1,q] if k else 0.0
                    # batch k can be walked by wave q only when all waves have STAGED it: every wave stages batch k before walking k-lead
                    if k-lead>=1:
                        start=max(start, fin[k-lead-1,:].max())
                    fin[k,q]=start+w[k,q]
            tot+=fin[n-1,:].max()
        Ld=int(tot)
    print(f"batch {batch} seg {seg}: barrier per batch (now) {A}, waves fully decoupled {B}, mean per wave {C}" + (f", lead {lead}: {Ld}" if Ld is not None else ""))
decouple(256,1536)
decouple(128,1536,lead=2)
decouple(128,1536,lead=4)
decouple(256,1536,lead=1)

# ---- how much shorter would the lists be if the staging lane tested the ELLIPSE (r2 <= 4) against each 4x4 sub-block instead of its box?
def exact_cover(batch=256, seg=1536, chunk=200000):
    sidp = sid[o]                                   # splat of each pair, composite order within tile
    cxm, cym, uxm, uym, wxm, wym = [a_[m].astype(np.float64) for a_ in (cx, cy, ux, uy, wx, wy)]
    det = uxm*wym - uym*wxm
    ia, ib, ic, idd = wym/det, -wxm/det, -uym/det, uxm/det   # [s t]^T = [[ia ib],[ic idd]] (pixel - c)
    px = np.arange(16, dtype=np.float64) + 0.5
    box_mask = np.zeros(P, dtype=np.uint16); ell_mask = np.zeros(P, dtype=np.uint16); covered = np.zeros(P, dtype=np.int32)
    tx = (tile % tiles_x) * 16; ty = (tile // tiles_x) * 16
    for a in range(0, P, chunk):
        b = min(P, a + chunk); s_ = sidp[a:b]
        dx = (tx[a:b, None] + px[None, :]) - cxm[s_, None]      # n x 16
        dy = (ty[a:b, None] + px[None, :]) - cym[s_, None]
        s = ia[s_, None, None] * dx[:, None, :] + ib[s_, None, None] * dy[:, :, None]   # n x 16(y) x 16(x)
        t = ic[s_, None, None] * dx[:, None, :] + idd[s_, None, None] * dy[:, :, None]
        cov = (s * s + t * t) <= 4.0
        inbox = ((np.arange(16)[None, None, :] >= lx0[a:b, None, None]) & (np.arange(16)[None, None, :] <= lx1[a:b, None, None]) &
                 (np.arange(16)[None, :, None] >= ly0[a:b, None, None]) & (np.arange(16)[None, :, None] <= ly1[a:b, None, None]))
        cov &= inbox
        covered[a:b] = cov.sum(axis=(1, 2))
        sb = cov.reshape(-1, 4, 4, 4, 4).any(axis=(2, 4))                         # n x 4(by) x 4(bx)
        bb = inbox.reshape(-1, 4, 4, 4, 4).any(axis=(2, 4))
        wts = (1 << np.arange(16)).reshape(4, 4)
        ell_mask[a:b] = (sb * wts).sum(axis=(1, 2)); box_mask[a:b] = (bb * wts).sum(axis=(1, 2))
    nbat = (seg + batch - 1) // batch
    bid_local = (pos % seg) // batch + (pos // seg) * nbat
    key = tile * 100000 + bid_local
    bfirst = np.r_[0, np.flatnonzero(np.diff(key)) + 1]
    def steps(mask):
        counts = np.stack([np.add.reduceat(((mask >> k) & 1).astype(np.int64), bfirst) for k in range(16)], axis=1).reshape(-1, 4, 4)
        return int(counts.sum()), int(counts.max(axis=2).sum()), int(counts.max(axis=2).max(axis=1).sum())
    eb, sb_, cb = steps(box_mask); ee, se, ce = steps(ell_mask)
    print(f"box lists: entries {eb}, wave-steps {sb_}, workgroup critical path {cb}")
    print(f"ellipse-exact lists: entries {ee} ({ee/eb:.3f}), wave-steps {se} ({se/sb_:.3f}), critical path {ce} ({ce/cb:.3f})")
    print(f"pairs with no covered pixel at all: {(covered == 0).mean():.3f}; covered pixels per pair {covered.mean():.1f}; lane utilisation box {covered.sum()/(sb_*64):.3f} -> exact {covered.sum()/(se*64):.3f}")
if os.environ.get("WALK_SIM_EXACT", "1") != "0":
    exact_cover()
