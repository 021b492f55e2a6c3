"""times the fine render pass kernel only (C2 shapes): python scratch/kbench.py [reps]"""
import sys, os, time, ctypes as C; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nvsr_amd
from bench import make_synthetic_scene, render_options
dev=torch.device('cuda',0)
reps=int(sys.argv[1]) if len(sys.argv)>1 else 3
mc,mf,sid,pose=make_synthetic_scene(dev,800,32,seed=0)
H=W=800; focal=0.5*W/np.tan(0.5*0.6911112)
ro,rd=nvsr_amd.nerf_helpers.get_ray_bundle(H,W,focal,pose)
rays=nvsr_amd.train_utils.pack_rays(ro,rd,2.0,6.0); N=rays.shape[0]
if not os.environ.get("KBENCH_ROW_ORDER"):      # the bench renders a frame in patch order (train_utils.patch_order)
    rays=rays[nvsr_amd.train_utils.patch_order(N,W,dev)[0]].contiguous()
capi=nvsr_amd.capi
ws=torch.empty(capi.lib().nvsr_render_workspace_floats(N,64,128),device=dev)
bufs=[torch.empty((N,3),device=dev),torch.empty(N,device=dev),torch.empty(N,device=dev),torch.empty((N,3),device=dev),torch.empty(N,device=dev),torch.empty(N,device=dev)]
sc,keep=mc.native_scene()
capi.call("nvsr_render_rays",C.byref(sc),capi.ptr(mc.packed_decoder()),capi.ptr(mf.packed_decoder()),N,64,128,capi.ptr(rays),0,0,None,None,None,None,*[capi.ptr(b) for b in bufs],capi.ptr(ws),capi.stream())
zf=ws[2*N*64:].view(N,192)
torch.cuda.synchronize()
ref=bufs[3].clone()
packed=mf.packed_decoder()
ts=[]
for i in range(reps):
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    capi.call("nvsr_render_pass",C.byref(sc),capi.ptr(packed),N,192,capi.ptr(rays),capi.ptr(zf),None,0,capi.ptr(bufs[3]),capi.ptr(bufs[4]),capi.ptr(bufs[5]),None,None,capi.stream())
    b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
fl=259072*N*192
print(capi.get_decoder_arithmetic(), os.environ.get("NVSR_HIP_LIB","product"), end=" ")
print("fine pass: min %.2f ms  med %.2f ms -> %.1f TFLOP/s (%.1f%% of 157.3)  same=%s  mean rgb %.4f"%(min(ts),np.median(ts),fl/min(ts)/1e9,100*fl/min(ts)/1e9/157.3, torch.equal(ref,bufs[3]), float(bufs[3].mean())))
