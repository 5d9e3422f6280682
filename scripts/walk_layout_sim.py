import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from suchtree_amd import synth
t0=time.time()
parent, dist = synth.skewed_tree(np.random.default_rng(5), 1_000_000, 0.9)
n=len(parent); print("tree", n, time.time()-t0)
# depth (edges to root) and height (nodes down to deepest leaf)
order=np.argsort(parent, kind="stable")  # not topological; compute depth iteratively
depth=np.zeros(n,np.int32)
# in-order ids: compute depth by repeated parent jumps (pointer doubling)
anc=parent.copy().astype(np.int64); d=np.where(parent>=0,1,0).astype(np.int32)
root=int(np.flatnonzero(parent<0)[0])
anc[root]=root
for _ in range(12):
    d=d+d[anc]; anc=anc[anc]
depth=d
print("depth max", depth.max(), "mean leaf depth", depth[::2].mean())
# height: process nodes by decreasing depth
height=np.ones(n,np.int32)
idx=np.argsort(-depth, kind="stable")
for x in idx:
    p=parent[x]
    if p>=0 and height[p]<height[x]+1: height[p]=height[x]+1
print("height root", height[root], time.time()-t0)
for H in (16,24,32,48,64,96,128,255):
    crown=height>H
    print("H",H,"crown nodes",int(crown.sum()))
# crown at H=115ish: choose H with crown <= 8192
H=128
while (height>H-1).sum() <= 8192: H-=1
crown=height>H
print("chosen H",H,"crown",int(crown.sum()))
# nb(x) = nodes on lineage below the portal (x itself included if not crown)
nbv=np.zeros(n,np.int32)
# process by increasing depth: nb[x] = 0 if crown else nb[parent]+1
for x in idx[::-1]:
    if not crown[x]:
        nbv[x]=nbv[parent[x]]+1
leaves=np.arange(0,n,2)
print("nb leaves: mean %.1f median %d p90 %d max %d"%(nbv[leaves].mean(), np.median(nbv[leaves]), np.percentile(nbv[leaves],90), nbv[leaves].max()))
# random leaf pairs: MRCA depth via in-order property: shallowest node between a and b
rng=np.random.default_rng(3)
P=rng.integers(0,1_000_000,(200_000,2))*2
lo=np.minimum(P[:,0],P[:,1]); hi=np.maximum(P[:,0],P[:,1])
# sparse table on depth
K=int(np.log2(n))+1
tab=[depth.astype(np.int32)]
for k in range(1,K):
    prev=tab[-1]; step=1<<(k-1)
    cur=prev.copy(); cur[:n-step]=np.minimum(prev[:n-step],prev[step:])
    tab.append(cur)
ln=hi-lo+1; kk=np.floor(np.log2(ln)).astype(int)
dm=np.array([min(tab[k][l], tab[k][h+1-(1<<k)]) for k,l,h in zip(kk,lo,hi)])
print("dm: mean %.1f median %d p90 %d; P(dm<12)=%.3f P(dm<28)=%.3f"%(dm.mean(), np.median(dm), np.percentile(dm,90), (dm<12).mean(), (dm<28).mean()))
da=depth[P[:,0]]; db=depth[P[:,1]]
kb=db-dm; nb_b=nbv[P[:,1]]
stream=np.minimum(kb,nb_b)
print("b stream below portal: mean %.1f floats; sectors now (64B-aligned block) mean %.2f; with 16B header in front mean %.2f"%(stream.mean(), np.ceil(stream/16).mean(), np.ceil((stream+4)/16).mean()))
print("fraction kb<=nb (meeting below the crown):", (kb<=nb_b).mean())
