import sys, os, time, json, tempfile
ROOT="/root/repo"
for p in (ROOT, os.path.join(ROOT, "colvars-finder_amd")): sys.path.insert(0, p)
import numpy as np, torch
import bench
from colvarsfinder import core, nn, pp
from tests.synth import Traj, diag_coeff_for
epochs=int(os.environ.get('EPOCHS','2001'))
dev=torch.device("cuda:0")
x,w,ref=bench.make_shard(100_000,0)
a=torch.tensor(diag_coeff_for(bench.N_ATOMS,bench.SEED),dtype=torch.float32)
torch.manual_seed(bench.SEED); np.random.seed(bench.SEED)
model=nn.EigenFunctions(bench.LAYERS,bench.K_NETS)
layer=pp.AlignFeatureLayer(bench.N_ATOMS,list(range(bench.N_ATOMS)),ref,[("position",tuple(range(bench.N_ATOMS)))])
T={"graph":0.0,"push":0.0,"n":0}
orig_gc=core.EigenFunctionTask._graph_call
def gc(self,key,body):
    t=time.perf_counter(); r=orig_gc(self,key,body); T["graph"]+=time.perf_counter()-t; T["n"]+=1; return r
core.EigenFunctionTask._graph_call=gc
orig_push=core._AsyncEpochLog.push
def push(self,ep):
    t=time.perf_counter(); r=orig_push(self,ep); T["push"]+=time.perf_counter()-t; return r
core._AsyncEpochLog.push=push
with tempfile.TemporaryDirectory() as tmp:
    task=core.EigenFunctionTask(Traj(x,w,1.0),layer,model,tmp,bench.ALPHA,bench.EIG_W,diag_coeff=a,beta=bench.BETA,lag_tau=0,learning_rate=bench.LR,k=bench.K_NETS,batch_size=20000,num_epochs=epochs,test_ratio=0.2,device=dev,verbose=False,save_model_every_step=0)
    torch.cuda.synchronize(); t0=time.perf_counter(); task.train(); torch.cuda.synchronize(); wall=time.perf_counter()-t0
import hashlib
h=hashlib.sha1(b"".join(np.ascontiguousarray(torch.cat([e[0],e[1]]).double().numpy()).tobytes() for e in task.loss_list)).hexdigest()
print(json.dumps(dict(n_loss_rows=len(task.loss_list),loss_sha1=h,wall_us_per_epoch=wall/epochs*1e6, graph_call_us=T["graph"]/T["n"]*1e6, push_us=T["push"]/epochs*1e6)))
