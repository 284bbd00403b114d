#!/usr/bin/env python3
"""How many launches would a multi-pick farthest-point initialisation over the CELLS need?  (round 6, CPU only, before
k_init_cells_multi was built)  numpy over every second colour of the RGB cube (cells of 8 x 8 x 8 colours), k = 256,
approximate binary32 CIE94 -- an acceptance-rate estimate, not a parity check.  A launch takes the M largest cell records; record i
becomes the next pick when every earlier candidate's cell is certified below it by the box bound and no pick so far lowers it; a
candidate that a pick lowers is skipped.
    python tools/init_cells_accept_sim.py M MAXPICKS      e.g. 8 4 -> 90 launches instead of 255"""
import numpy as np, time
# farthest-point init over the colours of the full cube (sub-lattice step 2), cells 8x8x8 RGB; how many launches would a multi-pick need?
step=2
v=np.arange(0,256,step,dtype=np.float32)
def srgb(u):
    u=u/255.0
    return np.where(u<=0.04045,u/12.92,((u+0.055)/1.055)**2.4)*100.0
lin=srgb(v)
R,G,B=np.meshgrid(lin,lin,lin,indexing='ij')
X=(R*0.4124+G*0.3576+B*0.1805)/95.047; Y=(R*0.2126+G*0.7152+B*0.0722)/100.0; Z=(R*0.0193+G*0.1192+B*0.9505)/108.883
def f(t): return np.where(t>0.008856,np.cbrt(t),7.787*t+16/116)
fx,fy,fz=f(X),f(Y),f(Z)
L=(116*fy-16).astype(np.float32).ravel(); a=(500*(fx-fy)).astype(np.float32).ravel(); b=(200*(fy-fz)).astype(np.float32).ravel()
n=L.size; m=256//step
idx=np.arange(n); ri=idx//(m*m); gi=(idx//m)%m; bi=idx%m
cs=8//step
cell=((ri//cs)*32+(gi//cs))*32+(bi//cs)
order=np.argsort(cell,kind='stable'); L,a,b,cell=L[order],a[order],b[order],cell[order]
per=cs**3
Lc=L.reshape(-1,per);ac=a.reshape(-1,per);bc=b.reshape(-1,per)
L0,L1,a0,a1,b0,b1=Lc.min(1),Lc.max(1),ac.min(1),ac.max(1),bc.min(1),bc.max(1)
C=np.sqrt(a*a+b*b)
def cie94(cL,ca,cb):
    # cie94(pixel, centroid): delta_e.wgsl -- reference colour = first argument? use pixel's chroma as C1
    dL=L-cL; c2=np.sqrt(ca*ca+cb*cb); dC=C-c2; da=a-ca; db=b-cb
    dH2=np.maximum(da*da+db*db-dC*dC,0)
    sc=1+0.045*C; sh=1+0.015*C
    return np.sqrt(dL*dL+(dC/sc)**2+dH2/(sh*sh))
def ub(cellid,cL,ca,cb):
    dl=max(abs(L1[cellid]-cL),abs(cL-L0[cellid])); dA=max(abs(a1[cellid]-ca),abs(ca-a0[cellid])); dB=max(abs(b1[cellid]-cb),abs(cb-b0[cellid]))
    return np.sqrt(dl*dl+dA*dA+dB*dB)
k=256
import sys
MULTI=int(sys.argv[1]); MAXP=int(sys.argv[2])
rng=np.random.default_rng(1)
first=rng.integers(n)
dist=cie94(L[first],a[first],b[first])
have=1; launches=1
hist=[]
t0=time.time()
while have<k:
    cm=dist.reshape(-1,per)
    am=cm.argmax(1); cmax=cm.max(1)
    top=np.argsort(-cmax)[:MULTI]
    picks=[]; earlier=[]
    for rank,cid in enumerate(top):
        ci=cid*per+am[cid]; d=cmax[cid]
        if rank==0:
            picks.append((cid,ci)); earlier.append(cid); continue
        if d<=0: break
        # every earlier cell certified below d?
        okb=True
        for xc in earlier:
            if not min(ub(xc,L[pi],a[pi],b[pi]) for (pc,pi) in picks)*1.001<d: okb=False
        if not okb: break
        oka=True
        for (pc,pi) in picks:
            dL=L[ci]-L[pi]; c2=C[pi]; dC=C[ci]-c2; da=a[ci]-a[pi]; db=b[ci]-b[pi]
            dH2=max(da*da+db*db-dC*dC,0); sc=1+0.045*C[ci]; sh=1+0.015*C[ci]
            de=np.sqrt(dL*dL+(dC/sc)**2+dH2/(sh*sh))
            if not de>=d: oka=False
        earlier.append(cid)
        if oka:
            picks.append((cid,ci))
            if len(picks)>=MAXP or have+len(picks)>=k: break
    for (pc,pi) in picks:
        dist=np.minimum(dist,cie94(L[pi],a[pi],b[pi]))
    have+=len(picks); launches+=1; hist.append(len(picks))
print("launches",launches,"for k",k, "picks histogram",np.bincount(hist))
# late-phase acceptance
h=np.array(hist); print("first half mean picks",h[:len(h)//2].mean(),"second half",h[len(h)//2:].mean())
