import sys, os, time, torch
sys.path.insert(0, os.getcwd())
from gaustudio_amd import scenes, _C
from gaustudio_diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
P,W,H,D=1_000_000,1920,1080,3
cam=scenes.make_camera(W,H); sc=scenes.make_scene(P,cam,seed=0); dev="cuda"
params={k:getattr(sc,k).to(dev).requires_grad_(True) for k in ("means3D","shs","opacities","scales","rotations")}
m2=torch.zeros_like(params["means3D"],requires_grad=True)
grads=[g.to(dev) for g in scenes.make_output_grads(cam,seed=1)]
rs=GaussianRasterizationSettings(H,W,cam.tanfovx,cam.tanfovy,torch.zeros(3,device=dev),1.0,cam.viewmatrix.to(dev),cam.projmatrix.to(dev),D,cam.campos.to(dev),False,False)
r=GaussianRasterizer(rs)
pool=torch.empty(8<<30,dtype=torch.uint8,device=dev); del pool
torch.cuda.synchronize()
ts=[]
for i in range(16):
    t0=time.perf_counter()
    for p in params.values(): p.grad=None
    c,ra,d,m,o=r(means3D=params["means3D"],means2D=m2,opacities=params["opacities"],shs=params["shs"],scales=params["scales"],rotations=params["rotations"])
    t1=time.perf_counter()
    torch.autograd.backward([c,d,m,o],grads)
    t2=time.perf_counter()
    torch.cuda.synchronize()
    t3=time.perf_counter()
    ts.append((round((t1-t0)*1e3,2),round((t2-t1)*1e3,2),round((t3-t2)*1e3,2)))
print(ts)
print(torch.cuda.memory_stats()["num_alloc_retries"], torch.cuda.memory_reserved()/1e9)
