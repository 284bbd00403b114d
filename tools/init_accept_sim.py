# (CPU only: numpy + the oracle; writes / reads /tmp/kmg_sim -- mkdir it first; run from the repository root)
# acceptance rate of "runner-up is exactly the next centroid" in the farthest-point initialisation (numpy float64 approximation
# of the literal CIE94: only the STATISTICS matter here)
import sys, numpy as np
sys.path.insert(0,__import__('os').path.join(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))), 'tests')); sys.path.insert(0,__import__('os').path.join(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))), 'kmeans-gpu_amd', 'python'))
import oracle_lib as O
from PIL import Image
def cie94(x, c):   # x [n,3] pixel first, c [3]
    dL=x[:,0]-c[0]; da=x[:,1]-c[1]; db=x[:,2]-c[2]
    C1=np.sqrt(x[:,1]**2+x[:,2]**2); C2=np.sqrt(c[1]**2+c[2]**2)
    dC=C1-C2; dH=np.sqrt(np.maximum(da*da+db*db-dC*dC,0))
    return np.sqrt(dL**2+(dC/(1+0.045*C1))**2+(dH/(1+0.015*C1))**2)
def run(lab, w, h, k, m):
    n=lab.shape[0]
    i0=int(h*0.93359375)*w+int(w*0.5625)
    cents=[i0]; D=cie94(lab,lab[i0])
    launches=0; accepted_hist=[]
    while len(cents)<k:
        launches+=1
        # top-m by D (ties ignored in the statistic)
        top=np.argpartition(-D,min(m,n-1))[:m]; top=top[np.argsort(-D[top])]
        new=[top[0]]
        for p in top[1:]:
            if len(cents)+len(new)>=k: break
            if D[p]<=0: break
            if all(cie94(lab[p:p+1],lab[q])[0]>=D[p] for q in new): new.append(p)
            else: break
        for q in new:
            D=np.minimum(D,cie94(lab,lab[q]))
        cents+=new; accepted_hist.append(len(new))
    return launches, np.bincount(accepted_hist,minlength=m+1)
tokyo=np.array(Image.open('tests/golden/tokyo.png').convert('RGBA'))
small=O.resize(tokyo,256,171).reshape(-1,4)
lab_t=O.rgb_to_lab(small).astype(np.float64)
rng=np.random.default_rng(1)
noise=rng.integers(0,256,(256*256,4),dtype=np.uint8); lab_n=O.rgb_to_lab(noise).astype(np.float64)
# cfg3-like: colours of a noise image = nearly all colours; subsample 2^18 colours
for name,lab,w,h in (('tokyo 256x171',lab_t,256,171),('noise 256x256',lab_n,256,256)):
    for k in (8,64,256):
        for m in (2,4,8):
            l,hist=run(lab,w,h,k,m)
            print(f'{name} k={k} m={m}: {l} launches instead of {k-1}  picks per launch {hist.tolist()}')
