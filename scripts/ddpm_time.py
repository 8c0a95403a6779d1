import sys, torch
sys.path.insert(0, '/root/repo')
import bench as B
from stdiff_amd import ops, synth as O
from stdiff_amd.engine import step_coefficients
from stdiff_amd.model.stDiff_scheduler import NoiseScheduler
dev='cuda:0'; R,G,T=131072,1024,1000
mask=torch.from_numpy(O.kfold_mask(G,0)).to(dev)
x=torch.randn(R,G,device=dev); eps=torch.randn(R,G,device=dev); gt=torch.randn(R,G,device=dev)
xp=torch.zeros(R,G,device=dev,dtype=torch.float16)
coef=step_coefficients(NoiseScheduler(T,beta_schedule='cosine')).to(dev)
t_dev=torch.full((1,),500,device=dev,dtype=torch.int32)
for _ in range(3): ops.ddpm_step(x,eps,gt,mask,coef,t_dev,500,seed=1,xp=xp)
torch.cuda.synchronize()
a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): ops.ddpm_step(x,eps,gt,mask,coef,t_dev,500,seed=1,xp=xp)
b.record(); torch.cuda.synchronize()
us=a.elapsed_time(b)*1e3/20
print(f"ddpm_step {R}x{G}: {us:.1f} us; nominal 18 B/elt -> {R*G*18/us/1e3:.0f} GB/s")
