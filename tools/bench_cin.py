import sys, time, json, torch, torch.nn as nn
dev=torch.device("cuda:0"); torch.backends.cudnn.benchmark=True
def timeit(fn,n=10,w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
res={}
with torch.no_grad():
    for cin in (1,2,4,8,12,15,16,17,20,24):
        conv=nn.Conv3d(cin,16,7,1,3).to(dev); x=torch.rand((4,cin,80,80,20),device=dev)
        res[f"conv7_cin{cin}"]=round(timeit(lambda: conv(x)),3)
    for cout in (1,4,15,16):
        conv=nn.Conv3d(32,cout,1,1,0).to(dev); x=torch.rand((4,32,80,80,20),device=dev)
        res[f"conv1_cout{cout}"]=round(timeit(lambda: conv(x)),3)
    for cin in (15,16):
        conv=nn.Conv3d(cin,16,7,1,3).to(dev); x=torch.rand((10,cin,64,64,64),device=dev)
        res[f"fine_conv7_cin{cin}_b10"]=round(timeit(lambda: conv(x),n=5,w=2),3)
print(json.dumps(res))
