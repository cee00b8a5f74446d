import sys, os, time, torch
sys.path.insert(0, os.getcwd())
from gaustudio_amd import scenes, runtime
from gaustudio_diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
P,W,H,D=1_000_000,1920,1080,3
dev=torch.device("cuda:0")
runtime.warm_start(dev)
cam=scenes.make_camera(W,H); sc=scenes.make_scene(P,cam,seed=0)
params={k:getattr(sc,k).to(dev).requires_grad_(True) for k in ("means3D","shs","opacities","scales","rotations")}
m2=torch.zeros_like(params["means3D"],requires_grad=True)
grads=[g.to(dev) for g in scenes.make_output_grads(cam,seed=1)]
rs=GaussianRasterizationSettings(H,W,cam.tanfovx,cam.tanfovy,torch.zeros(3,device=dev),1.0,cam.viewmatrix.to(dev),cam.projmatrix.to(dev),D,cam.campos.to(dev),False,False)
r=GaussianRasterizer(rs)
torch.cuda.synchronize()
ev=[torch.cuda.Event(enable_timing=True) for _ in range(41)]
ev[0].record()
for i in range(40):
    for p in params.values(): p.grad=None
    c,ra,d,m,o=r(means3D=params["means3D"],means2D=m2,opacities=params["opacities"],shs=params["shs"],scales=params["scales"],rotations=params["rotations"])
    torch.autograd.backward([c,d,m,o],grads)
    ev[i+1].record()
torch.cuda.synchronize()
print([round(ev[i].elapsed_time(ev[i+1]),3) for i in range(40)])
