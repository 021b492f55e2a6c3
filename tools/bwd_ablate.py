"""gates backward kernel: time with subsets of the planes receiving gradient (fine pass, 4096 rays x 128)"""
import sys, os, ctypes as C; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import nvsr_amd
from bench import make_synthetic_scene, render_options
dev=torch.device('cuda',0); capi=nvsr_amd.capi
mc,mf,sid,pose=make_synthetic_scene(dev,200,32,seed=0)
N,S=4096,128
H=W=800; focal=0.5*W/np.tan(0.5*0.6911112)
sel=torch.randint(0,H,(N,2),device=dev)
ro,rd=nvsr_amd.training.get_ray_bundle_at(H,W,focal,pose,sel)
rays=nvsr_amd.train_utils.pack_rays(ro,rd,2.0,6.0)
z=torch.sort(torch.rand(N,S,device=dev)*4+2,-1)[0].contiguous()
raw=torch.empty(N,S,4,device=dev); gates=torch.empty(N,S,32,dtype=torch.int32,device=dev)
sc,keep=mf.native_scene()
capi.call("nvsr_decode_rays_ex",C.byref(sc),capi.ptr(mf.packed_decoder()),N,S,capi.ptr(rays),capi.ptr(z),capi.ptr(raw),capi.ptr(gates),None,capi.stream())
g_raw=torch.randn(N,S,4,device=dev)*1e-3
gpl=[torch.zeros_like(k) for k in keep]
vws=torch.empty(N*S*48,device=dev)
def run(mask,label,use_ws=False):
    gptrs=(C.c_void_p*4)(*[gpl[d].data_ptr() if mask[d] else None for d in range(4)])
    ts=[]
    for i in range(4):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record()
        capi.call("nvsr_render_pass_backward_gates",C.byref(sc),capi.ptr(mf.packed_decoder()),capi.ptr(mf.packed_decoder_bwd()),N,S,capi.ptr(rays),capi.ptr(z),capi.ptr(g_raw),capi.ptr(gates),gptrs,capi.ptr(vws) if use_ws else None,None,capi.stream())
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    print("%-28s %.3f ms"%(label,min(ts)))
run([0,0,0,0],"no scatter")
run([0,0,0,1],"view plane only (atomics)")
run([1,0,0,0],"one position plane")
run([1,1,1,0],"3 position planes")
run([1,1,1,1],"all (view by atomics)")
run([0,0,0,1],"view plane only, row ws",True)
run([1,1,1,1],"all, row ws",True)
