"""per-step time of the fine pass at 1 and 2 workgroups per CU"""
import sys, os, time, ctypes as C; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nvsr_amd
from bench import make_synthetic_scene
dev=torch.device('cuda',0)
mc,mf,sid,pose=make_synthetic_scene(dev,800,32,seed=0)
H=W=800; focal=0.5*W/np.tan(0.5*0.6911112)
ro,rd=nvsr_amd.nerf_helpers.get_ray_bundle(H,W,focal,pose)
rays_all=nvsr_amd.train_utils.pack_rays(ro,rd,2.0,6.0)
capi=nvsr_amd.capi
sc,keep=mc.native_scene(); packed=mf.packed_decoder()
S=192
for N in [int(a) for a in sys.argv[1:]] or [32768, 65536, 131072, 640000]:
    rays=rays_all[:N].contiguous()
    z=torch.linspace(2,6,S,device=dev).expand(N,S).contiguous()
    o3,o1,o2=torch.empty((N,3),device=dev),torch.empty(N,device=dev),torch.empty(N,device=dev)
    ts=[]
    for i in range(3):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record()
        capi.call("nvsr_render_pass",C.byref(sc),capi.ptr(packed),N,S,capi.ptr(rays),capi.ptr(z),None,0,capi.ptr(o3),capi.ptr(o1),capi.ptr(o2),None,None,capi.stream())
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    t=min(ts)
    print("N=%7d: %.3f ms  -> %.1f us/step  %.1f TFLOP/s"%(N,t,1e3*t/S/max(1,np.ceil(N/65536)), 259072*N*S/t/1e9))
