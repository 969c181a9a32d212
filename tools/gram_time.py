import time, torch, sys
sys.path.insert(0, "/root/repo")
from cassierl_amd import trpo as T
n=524288
bk=T.BaselineKernels(torch.device("cuda:0"), 26)
obs=torch.randn(n,26,device="cuda"); t=torch.randint(0,1000,(n,),device="cuda"); y=torch.randn(n,dtype=torch.float64,device="cuda")
for _ in range(3): bk.gram(obs,t,y)
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(20): bk.gram(obs,t,y)
torch.cuda.synchronize(); print("gram ms", (time.perf_counter()-t0)/20*1e3)
