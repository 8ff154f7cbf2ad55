import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from selfpose3d_amd import _lib, synthetic as syn
from selfpose3d_amd.camera_pack import pack_cameras, CAM_W0, CAM_H0
dev=torch.device("cuda:0")
def timed(fn, iters=100):
    for _ in range(5): fn()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/iters*1e3
img,(w,h),J=(960,512),(240,128),15
for name,B,V,cube,gs in (("coarse_b4",4,5,syn.INITIAL_CUBE_SIZE,syn.SPACE_SIZE),("fine_b10",10,5,syn.FINE_CUBE_SIZE,syn.FINE_GRID_SIZE)):
    meta=syn.make_meta(B,V,img); tab=pack_cameras(meta,B,img)
    tab_oob=tab.copy(); tab_oob[:,:,CAM_W0]=0; tab_oob[:,:,CAM_H0]=0
    if name.startswith("fine"):
        rng=np.random.default_rng(0); c=np.stack([rng.uniform(-1500,1500,B),rng.uniform(-2000,1000,B),rng.uniform(700,1100,B)],1).astype(np.float32)
        centers=torch.from_numpy(c).to(dev)
    else: centers=torch.tensor([syn.SPACE_CENTER]*B,dtype=torch.float32,device=dev)
    valid=torch.ones(B,dtype=torch.uint8,device=dev)
    hms=[x.to(dev) for x in syn.random_heatmaps(B,V,J,h,w,seed=7)]
    packed=_lib.pack_heatmaps(hms,jp=16); views=[packed[c] for c in range(V)]
    res={}
    for tag,t in (("normal",tab),("all_oob",tab_oob)):
        cam=torch.from_numpy(t).to(dev)
        res[tag]=round(timed(lambda: _lib.unproject_fwd(views,_lib.LAYOUT_NHWC,16,cam,centers,valid,B,J,h,w,cube,gs,img,False)),2)
    # invalid samples: only zero-fill
    inval=torch.zeros(B,dtype=torch.uint8,device=dev); cam=torch.from_numpy(tab).to(dev)
    res["all_invalid_zero_fill"]=round(timed(lambda: _lib.unproject_fwd(views,_lib.LAYOUT_NHWC,16,cam,centers,inval,B,J,h,w,cube,gs,img,False)),2)
    print(name,res)
