import sys, time, os
sys.path[:0]=[os.environ.get('GRAFT_REPO_ROOT','/root/repo'), os.path.join(os.environ.get('GRAFT_REPO_ROOT','/root/repo'),'oracle')]
import numpy as np, torch
import deepspeaker_oracle as O
from deepspeaker_pytorch_amd.model import DeepSpeakerModel, TripletMarginLoss, get_engine
from deepspeaker_pytorch_amd.mining import mine_semihard_negatives, select_triplets
dev=torch.device('cuda',0)
sd=O.make_state_dict(seed=0,num_classes=1211)
m=DeepSpeakerModel(512,1211,precision='bf16x3'); m.load_state_dict({k:torch.from_numpy(np.array(v)) for k,v in sd.items()}); m=m.to(dev).eval()
x=torch.randn(768,1,160,64,device=dev)
c1=torch.randint(0,64,(256,),device=dev); lab=torch.cat([c1,c1,(c1+1)%64])
lf=TripletMarginLoss(0.1)
def fwd_only():
    with torch.no_grad(): return m(x)
def full():
    with torch.no_grad():
        embs=list(m(x).split(256)); l=lf.forward(*embs); s=select_triplets(*embs,margin=0.1)
        mi=mine_semihard_negatives(embs[0],embs[1],c1,torch.cat(embs),lab)
for name,fn in (('fwd_only',fwd_only),('full',full)):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    for n in (10,40):
        t0=time.perf_counter()
        for _ in range(n): fn()
        t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
        print(name,n,'host enqueue ms/step %.3f'%((t1-t0)/n*1e3),'total ms/step %.3f'%((t2-t0)/n*1e3))
eng=get_engine(); eng.profile=[]
for n in (20,):
    t0=time.perf_counter()
    for _ in range(n): full()
    t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    print('full+events',n,'host enqueue ms/step %.3f'%((t1-t0)/n*1e3),'total ms/step %.3f'%((t2-t0)/n*1e3))
