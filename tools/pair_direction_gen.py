# (CPU only: numpy + the oracle; writes / reads /tmp/kmg_sim -- mkdir it first; run from the repository root)
# realistic k=256 centroids (12 Lloyd iterations of the bench workload on a 2M-pixel sample) + labels of all 2^24 colours (CPU oracle)
import sys, time, numpy as np
sys.path.insert(0,__import__('os').path.join(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))), 'tests')); sys.path.insert(0,__import__('os').path.join(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))), 'kmeans-gpu_amd', 'python'))
import oracle_lib as O
from kmeans_gpu_amd import synth
k=256; n=1<<21
px=synth.uniform_rgba_numpy(synth.SEED_CFG3,n)
N=8192*8192
sel=synth.uniform_rgba_at(synth.SEED_CFG3,np.arange(k,dtype=np.uint64)*np.uint64(N//k))
cent=O.centroids4(O.rgb_to_lab(sel))
t=time.time()
for it in range(12):
    lab,acc=O.assign_accumulate_rgba(px,cent)
    cent,_=O.finalize(acc,cent)
print('lloyd',time.time()-t)
idx=np.arange(1<<24,dtype=np.uint32)
cube=np.zeros((1<<24,4),np.uint8); cube[:,0]=idx&255; cube[:,1]=(idx>>8)&255; cube[:,2]=(idx>>16)&255; cube[:,3]=255
t=time.time()
lab,_=O.assign_accumulate_rgba(cube,cent)
print('cube',time.time()-t)
np.save('/tmp/kmg_sim/labels.npy',lab.astype(np.uint8)); np.save('/tmp/kmg_sim/cent.npy',cent)
