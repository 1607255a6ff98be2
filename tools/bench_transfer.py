"""Transfer-operator mode (lag_tau > 0) train step at the config-3 shape: extra measurement."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "colvars-finder_amd")): sys.path.insert(0, p)
import numpy as np, torch
from colvarsfinder import core, nn, pp
from tests.synth import Traj, make_molecule_traj
dev=torch.device("cuda")
N=22; n=100000; B=20000
traj,w,ref=make_molecule_traj(N,n,seed=1,scale=2.0,sigma=0.3)
layer=pp.AlignFeatureLayer(N,list(range(N)),ref,[("position",tuple(range(N)))])
model=nn.EigenFunctions([66,20,20,20,1],3)
task=core.EigenFunctionTask(Traj(traj,w,0.1),layer,model,"/tmp/cvf_b",20.0,[1.0,0.75,0.5],lag_tau=1.0,learning_rate=1e-3,k=3,batch_size=B,device=dev,verbose=False,save_model_every_step=0)
X,W=task._traj,task._weights
lag=task.lag_idx
log=torch.zeros(4,9,device=dev,dtype=torch.float64)
def step(i):
    b=i%4; s=b*B
    task._graph_step(("b",b),lambda: task.train_step(X[s:s+B],W[s:s+B],X[s+lag:s+lag+B],W[s+lag:s+lag+B]),log[b])
for i in range(10): step(i)
torch.cuda.synchronize(); t0=time.perf_counter()
for i in range(100): step(10+i)
torch.cuda.synchronize(); el=time.perf_counter()-t0
print(json.dumps({"transfer_mode_ms_per_step": el/100*1e3, "frames_per_s": B*100/el, "loss": float(log[0,0])}))
task._use_graphs, task._events = False, {}
for i in range(30):
    torch.cuda._sleep(2_000_000)
    step(200 + i)
torch.cuda.synchronize()
print(json.dumps({"call_avg_us": {n_: float(np.mean([a_.elapsed_time(b_) for a_, b_ in ev[5:]])) * 1e3 for n_, ev in task._events.items()}}))
