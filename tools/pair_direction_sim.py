# (CPU only: numpy + the oracle; writes / reads /tmp/kmg_sim -- mkdir it first; run from the repository root)
import sys, numpy as np
sys.path.insert(0,__import__('os').path.join(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))), 'tests')); sys.path.insert(0,__import__('os').path.join(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))), 'kmeans-gpu_amd', 'python'))
import oracle_lib as O
lab=np.load('/tmp/kmg_sim/labels.npy'); cent=np.load('/tmp/kmg_sim/cent.npy')
idx=np.arange(1<<24,dtype=np.uint32)
r=idx&255; g=(idx>>8)&255; b=(idx>>16)&255
cell=((r>>3)<<10)|((g>>3)<<5)|(b>>3)
within=((r&7)<<6)|((g&7)<<3)|(b&7)
order=np.argsort(cell*512+within,kind='stable')
L=lab[order].reshape(32768,512)            # labels per cell, colour order x=(r&7) major
cube=np.zeros((1<<24,4),np.uint8); cube[:,0]=r; cube[:,1]=g; cube[:,2]=b; cube[:,3]=255
lab3=O.rgb_to_lab(cube)[order].reshape(32768,512,3)
xyz=np.stack(np.meshgrid(np.arange(8),np.arange(8),np.arange(8),indexing='ij'),-1).reshape(512,3).astype(np.float64)
# features
Lc=lab3[...,0].astype(np.float64); a=lab3[...,1].astype(np.float64); bb=lab3[...,2].astype(np.float64)
C=np.sqrt(a*a+bb*bb); wC=1/(1+0.045*C)**2; wH=1/(1+0.015*C)**2
F=np.stack([Lc,wC,wC*C,wH*a,wH*bb,wH*C,wH],-1)       # [cells,512,7]
xc=xyz-3.5
beta=np.einsum('cpm,pd->cmd',F,xc)/ (xc**2).sum(0)      # [cells,7,3]
# per cell label stats
nlab=np.array([len(np.unique(row)) for row in L])
multi=np.where(nlab>1)[0]
print('cells by labels',np.bincount(np.minimum(nlab,4)))
def top2(row):
    v,c=np.unique(row,return_counts=True); o=np.argsort(-c,kind='stable'); return v[o[0]],v[o[1]]
AB=np.array([top2(L[c]) for c in multi])
A=AB[:,0]; B=AB[:,1]
Lm=L[multi]
def gather_count(dirs):     # dirs [M,3] ints per cell -> colours needing the gather under the entry encoding
    p=(xyz[None,:,:]*dirs[:,None,:]).sum(-1)            # [M,512]
    notA=Lm!=A[:,None]; notB=Lm!=B[:,None]
    tlo=np.where(notA,p,1e9).min(1); thi=np.where(notB,p,-1e9).max(1)
    w=np.maximum(thi+1-tlo,0)
    fine_narrow=((p>=tlo[:,None])&(p<=thi[:,None])).sum(1)
    low=(p<tlo[:,None]).sum(1); high=(p>thi[:,None]).sum(1)
    fine_wide=512-np.maximum(low,high)
    return np.where(w<=6,fine_narrow,fine_wide)
def quant(gv,m=2):
    mx=np.abs(gv).max(1,keepdims=True); mx[mx==0]=1
    q=np.rint(m*gv/mx).astype(int)
    z=(q==0).all(1); q[z]=[m,0,0]
    return q
# (a) today's heuristic: centre of mass of A vs rest at half-cell resolution
half=(xyz>=4).astype(np.float64)
isA=(Lm==A[:,None])
nA=isA.sum(1,keepdims=True); nR=512-nA
hA=(isA[:,:,None]*half[None]).sum(1); hR=((~isA)[:,:,None]*half[None]).sum(1)
gh=hR*nA-hA*nR
d_heur=quant(gh)
# exact centre of mass at full resolution
cA=(isA[:,:,None]*xyz[None]).sum(1)/np.maximum(nA,1); cR=((~isA)[:,:,None]*xyz[None]).sum(1)/np.maximum(nR,1)
d_com=quant(cR-cA)
# (b) analytic normal from the two centroids A, B
def wvec(cj,ci):
    Lj,aj,bj=cj[:,0],cj[:,1],cj[:,2]; Li,ai,bi=ci[:,0],ci[:,1],ci[:,2]
    Cj=np.sqrt(aj*aj+bj*bj); Ci=np.sqrt(ai*ai+bi*bi)
    Nj=aj*aj+bj*bj-Cj*Cj; Ni=ai*ai+bi*bi-Ci*Ci
    return np.stack([-2*(Lj-Li),Cj*Cj-Ci*Ci,-2*(Cj-Ci),-2*(aj-ai),-2*(bj-bi),2*(Cj-Ci),Nj-Ni],-1)
c64=cent.astype(np.float64)
w=wvec(c64[B],c64[A])                   # K_B - K_A = c0 + w.F : positive on the A side
grad=np.einsum('cm,cmd->cd',w,beta[multi])
d_ana=quant(-grad)                      # p grows from A towards B
d_ana3=quant(-grad,3)
# (c) best of 125 / of 343
def best_of(m):
    r_=np.arange(-m,m+1); D=np.stack(np.meshgrid(r_,r_,r_,indexing='ij'),-1).reshape(-1,3); D=D[(D!=0).any(1)]
    best=np.full(len(multi),10**9)
    for d in D:
        best=np.minimum(best,gather_count(np.broadcast_to(d,(len(multi),3))))
    return best
tot=32768*512
for name,cnt in (('heuristic (half-cell CoM)',gather_count(d_heur)),('full CoM',gather_count(d_com)),('analytic normal -2..2',gather_count(d_ana)),('best of 125',best_of(2))):
    two=nlab[multi]==2
    print(f'{name:28s} gather share of all colours: {cnt.sum()/tot:.4f}   2-label cells {cnt[two].sum()/tot:.4f}   3+ {cnt[~two].sum()/tot:.4f}')
np.save('/tmp/kmg_sim/d_ana.npy',d_ana)
