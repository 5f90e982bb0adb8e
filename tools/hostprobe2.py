import sys, time, os
R=os.environ.get('GRAFT_REPO_ROOT','/root/repo'); sys.path[:0]=[R, os.path.join(R,'oracle')]
import numpy as np, torch
import deepspeaker_oracle as O
from deepspeaker_pytorch_amd.model import DeepSpeakerModel, TripletMarginLoss, get_engine
from deepspeaker_pytorch_amd.mining import mine_semihard_negatives, select_triplets
dev=torch.device('cuda',0)
sd=O.make_state_dict(seed=0,num_classes=1211)
g=torch.Generator(device='cpu').manual_seed(1234)
x=torch.randn(768,1,160,64,generator=g).to(dev)
c1=torch.randint(0,64,(256,),generator=g).to(dev); lab=torch.cat([c1,c1,(c1+1)%64])
lf=TripletMarginLoss(0.1); eng=get_engine()
m=DeepSpeakerModel(512,1211,precision='bf16x3'); m.load_state_dict({k:torch.from_numpy(np.array(v)) for k,v in sd.items()}); m=m.to(dev).eval()
def step():
    with torch.no_grad():
        embs=list(m(x).split(256)); l=lf.forward(*embs); s=select_triplets(*embs,margin=0.1)
        mi=mine_semihard_negatives(embs[0],embs[1],c1,torch.cat(embs),lab)
    return l,s,mi
for _ in range(5): step()
torch.cuda.synchronize()
eng.profile=[]
ts=[]
t0=time.perf_counter()
for i in range(12):
    a=time.perf_counter(); step(); b=time.perf_counter(); ts.append((b-a)*1e3)
t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
print('per-step host ms', [round(v,2) for v in ts]); print('enqueue total %.1f sync tail %.1f'%((t1-t0)*1e3,(t2-t1)*1e3))
# pure host overhead: tiny batch (GPU work negligible)
xs=torch.randn(24,1,160,64,device=dev); c1s=c1[:8]; labs=torch.cat([c1s,c1s,(c1s+1)%64])
def small():
    with torch.no_grad():
        embs=list(m(xs).split(8)); l=lf.forward(*embs); s=select_triplets(*embs,margin=0.1)
        mi=mine_semihard_negatives(embs[0],embs[1],c1s,torch.cat(embs),labs)
eng.profile=None
for _ in range(5): small()
torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(50): small()
t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
print('small batch: host ms/step %.3f total %.3f'%((t1-t0)/50*1e3,(t2-t0)/50*1e3))
import cProfile, pstats
pr=cProfile.Profile(); pr.enable()
for _ in range(50): small()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
