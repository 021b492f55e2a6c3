"""time one train step (4096 rays, 64+64, planes 200^2) forward + backward + Adam.
usage: train_step_time.py [planes|decoder|planes+decoder]   (nerf.train.what; default planes = Feature_Planes_Only.yml)"""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nvsr_amd
from bench import make_synthetic_scene, render_options
dev=torch.device('cuda',0)
mc,mf,sid,pose=make_synthetic_scene(dev,200,32,seed=0)
what=sys.argv[1] if len(sys.argv)>1 else "planes"
for m in (mc,mf):
    for n,p in m.named_parameters(): p.requires_grad_(("planes" in what) if "planes_" in n else ("decoder" in what and "rot_mats" not in n))
    m.train()
H=W=100; focal=0.5*W/np.tan(0.5*0.6911112)
ro,rd=nvsr_amd.nerf_helpers.get_ray_bundle(H,W,focal,pose)
opts,scfg=render_options(64,64,perturb=True,noise=0.2)
sel=torch.randperm(H*W,device=dev)[:4096]
batch=torch.stack([ro.reshape(-1,3)[sel],rd.reshape(-1,3)[sel]],0)
target=torch.rand(4096,3,device=dev)
train_params=[p for m in (mc,mf) for p in m.parameters() if p.requires_grad]
train_params=list({id(p):p for p in train_params}.values())       # the planes are shared by the two models
opt=torch.optim.Adam(train_params,lr=4e-3 if what=="planes" else 5e-4)
N=4096
rnd=dict(t_rand=torch.rand(N,64,device=dev),u=torch.rand(N,64,device=dev),noise_coarse=0.2*torch.randn(N,64,device=dev),noise_fine=0.2*torch.randn(N,128,device=dev))
def step():
    opt.zero_grad(set_to_none=True)
    out=nvsr_amd.train_utils.run_one_iter_of_nerf(H,W,focal,mc,mf,batch,opts,sid,mode="train",scene_config=scfg,randoms=rnd)
    loss=torch.nn.functional.mse_loss(out[0],target)+torch.nn.functional.mse_loss(out[3],target)
    loss.backward(); opt.step()
    return loss
for i in range(3): l=step()
torch.cuda.synchronize(); t0=time.perf_counter()
K=20
for i in range(K): l=step()
torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/K
print("train step (4096 rays, 64+64, planes 200^2, Adam on "+what+"): %.2f ms/iter, loss %.4f  (reference CPU: 7.9 s fwd + 4.9 s bwd)"%(dt*1e3,float(l)))
with torch.no_grad():
    t0=time.perf_counter()
    for i in range(K): out=nvsr_amd.train_utils.run_one_iter_of_nerf(H,W,focal,mc,mf,batch,opts,sid,mode="train",scene_config=scfg,randoms=rnd)
    torch.cuda.synchronize(); print("forward only: %.2f ms"%((time.perf_counter()-t0)/K*1e3))
