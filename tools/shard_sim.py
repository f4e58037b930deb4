#!/usr/bin/env python3
"""Shard balance of an N-rank frame under different STATIC tile -> rank maps, from the CPU oracle's per-sample counters (no GPU):
per 16x16 tile a cost proxy (Aabb tests + 5 x primitive tests + 30 per ray, a few spp) and, per map, the slowest rank's load
against the mean / the lightest rank's.  Maps: `tile % N` (what rtg_params.rank / nranks does), a golden-ratio permutation of
the tile indices, two 2-D lattices, a random permutation, and longest-processing-time-first on the TRUE cost (the bound of any
cost-aware static map).  TEST INFRASTRUCTURE (imports the oracle).  usage: python tools/shard_sim.py"""
import sys, os, numpy as np, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import __graft_entry__ as g
from scene_cases import build_case
from concurrent.futures import ThreadPoolExecutor
pkg=g.load_package(); ora=g.load_oracle()
def tile_cost(name,nx,ny,spp):
    sc,cam,_,_,_=build_case(pkg,ora,name,nx,ny)
    ys,xs=np.mgrid[0:ny,0:nx]
    xs=xs.ravel().astype(np.uint32); ys=ys.ravel().astype(np.uint32)
    cost=np.zeros(nx*ny)
    def run(s):
        rgb,info=sc.debug_samples(cam,nx,ny,spp,xs,ys,np.full(xs.size,s,np.uint32))
        return info[:,2]+5.0*info[:,3]+30.0*(info[:,0]+1)   # box tests + 5 x prim tests + 30 per ray: a wave-instruction-ish proxy
    with ThreadPoolExecutor(8) as ex:
        for c in ex.map(run, range(spp)): cost+=c
    # debug_samples' y is the reference's y (bottom-up)? either way tiles are symmetric for this purpose: map y -> row = ny-1-y
    img=cost.reshape(ny,nx)[::-1]
    ty,tx=(ny+15)//16,(nx+15)//16
    t=np.zeros((ty,tx))
    for j in range(ty):
        for i in range(tx):
            t[j,i]=img[16*j:16*j+16,16*i:16*i+16].sum()
    return t
def spread(owner,t,N):
    loads=np.bincount(owner.ravel(),weights=t.ravel(),minlength=N)
    return loads.max()/loads.mean()-1, loads.max()/loads.min()-1
for name,nx,ny,spp in (("book1",1200,800,4),("book2",800,800,3)):
    t0=time.time(); t=tile_cost(name,nx,ny,spp); T=t.size; ty,tx=t.shape
    print(name,t.shape,"%.0fs"%(time.time()-t0), "cost max/mean tile %.1f"%(t.max()/t.mean()))
    idx=np.arange(T)
    for N in (2,4,8):
        res={}
        res['tile%N']=spread((idx%N).reshape(ty,tx),t,N)
        # golden-ratio permutation
        A=int(T*0.6180339887)|1
        import math
        while math.gcd(A,T)!=1: A+=2
        inv=pow(A,-1,T)
        res['golden']=spread((((idx*inv)%T)%N).reshape(ty,tx),t,N)
        # 2D lattice (tx+3ty)%N style
        for a in (3,5):
            yy,xx=np.mgrid[0:ty,0:tx]
            res['(x+%dy)%%N'%a]=spread(((xx+a*yy)%N),t,N)
        # random perm
        rng=np.random.default_rng(1); perm=rng.permutation(T)
        res['random']=spread((perm%N).reshape(ty,tx),t,N)
        # LPT greedy on the true cost (upper bound of what cost-aware can do) with equal tile counts not enforced
        order=np.argsort(-t.ravel()); loads=np.zeros(N); own=np.zeros(T,int)
        for k in order:
            r=loads.argmin(); own[k]=r; loads[r]+=t.ravel()[k]
        res['lpt(true cost)']=spread(own.reshape(ty,tx),t,N)
        # LPT on a noisy estimate: cost from a different sample subset ~ emulate by adding sampling noise
        print(" N=%d "%N+"  ".join("%s: max/mean %+.1f%% max/min %+.1f%%"%(k,100*v[0],100*v[1]) for k,v in res.items()))
