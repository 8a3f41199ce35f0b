#!/usr/bin/env python3
"""The dependency structure of the covariance stage on the bench frames (CPU; oracle heat maps): lone walks by a literal
Python BFS (sp_extractor.cpp:281-314), which lower keypoints each lone region shares a pixel with, components and the
longest dependency path — what a replay along the true overlap DAG could gain over the per-component replay of
csrc/cov.hip (nothing, the components are paths)."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from collections import deque
from oracle import oracle
from sp_orb_slam_amd import synth, weights
H,W,nf=480,752,1000
for det in ("dense","sparse"):
  blob=weights.synthetic(7,det)
  for seed in (200,201):
    img=synth.make_image(seed,H,W)
    ref=oracle.extract(blob,img,nf)
    hinv=ref["heat_inv"]; kp=ref["kp_xy"].astype(int); K=ref["K"]
    def lone(j):
        x0,y0=kp[j]; vis=set(); q=deque([(x0,y0)]); pops=[]
        while q:
            x,y=q.popleft(); pops.append((x,y)); vis.add((x,y)); here=hinv[y,x]
            for dx,dy in ((-1,0),(0,-1),(1,0),(0,1)):
                xx,yy=x+dx,y+dy
                if dx<0 and not xx>0: continue
                if dy<0 and not yy>0: continue
                if dx>0 and not xx<W: continue
                if dy>0 and not yy<H: continue
                v=hinv[yy,xx]
                if (xx,yy) not in vis and v>0 and v<here: q.append((xx,yy))
            if len(pops)>20000: break
        return set(pops), len(pops)
    iso=[]; npops=[]
    for j in range(K):
        s_,n_=lone(j); iso.append(s_); npops.append(n_)
    owner={}
    for j in range(K):
        for p in iso[j]: owner.setdefault(p,[]).append(j)
    preds=[set() for _ in range(K)]
    for p,l in owner.items():
        if len(l)>1:
            l=sorted(l)
            for a in range(1,len(l)):
                for b in range(a): preds[l[a]].add(l[b])
    dirty=[j for j in range(K) if preds[j]]
    depth=[0]*K
    for j in range(K):
        if preds[j]: depth[j]=1+max(depth[i] if preds[i] else 0 for i in preds[j])
    # components
    parent=list(range(K))
    def find(x):
        while parent[x]!=x:
            parent[x]=parent[parent[x]]; x=parent[x]
        return x
    for j in range(K):
        for i in preds[j]:
            a,b=find(i),find(j)
            if a!=b: parent[max(a,b)]=min(a,b)
    comp={}
    for j in dirty: comp.setdefault(find(j),[]).append(j)
    sizes=sorted((len(v) for v in comp.values()),reverse=True)
    # cost model: old = max over components sum of pops of dirty members; dag = longest path weighted
    print(det,seed,"K",K,"pops mean",np.mean(npops),"max",max(npops),"dirty",len(dirty),"components",len(comp),"largest",sizes[:5],"max depth",max(depth),"depth hist",np.bincount(depth)[:12], "max preds", max(len(p) for p in preds))
