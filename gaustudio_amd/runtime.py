"""Process-level initialisation helpers for the MI355X rasterizer.

`warm_start()` does the one-time work a serving / training process wants out of its first iterations:
  * loads libgsrast.so and its gfx950 code objects (first launch of every kernel);
  * reserves a pool in torch's caching allocator.  The operator allocates a few large buffers per call
    (records 64 B/Gaussian, keys 12 B/instance, backward rows 48 B/instance, gradients 304 B/Gaussian); until the
    allocator's cache has seen them, each costs a hipMalloc of several milliseconds.  An MI355X has 288 GB of
    HBM3E: reserving a few GiB up front is free and makes the first iteration as fast as the thousandth.
"""
import torch

from . import _C, scenes


def warm_start(device=None, pool_bytes=8 << 30):
    if not torch.cuda.is_available():
        raise RuntimeError("warm_start needs a ROCm device")
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    _C.lib()
    with torch.cuda.device(dev):
        free, _total = torch.cuda.mem_get_info(dev)
        n = int(min(pool_bytes, free // 2))
        if n > 0:
            pool = torch.empty(n, dtype=torch.uint8, device=dev)
            del pool                                   # stays cached: later allocations are carved from it
        # one tiny forward + backward: loads every kernel, creates the pinned read-back word and the stage events
        from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer
        cam = scenes.make_camera(64, 48)
        sc = scenes.make_scene(512, cam, seed=0, sigma_px_median=3.0)
        leaves = [getattr(sc, k).to(dev).requires_grad_(True) for k in ("means3D", "opacities", "shs", "scales", "rotations")]
        for D in (0, 1, 2, 3):
            rs = GaussianRasterizationSettings(cam.height, cam.width, cam.tanfovx, cam.tanfovy, torch.zeros(3), 1.0,
                                               cam.viewmatrix.to(dev), cam.projmatrix.to(dev), D, cam.campos.to(dev),
                                               False, False)
            out = GaussianRasterizer(rs)(means3D=leaves[0], means2D=torch.zeros_like(leaves[0]), opacities=leaves[1],
                                         shs=leaves[2], scales=leaves[3], rotations=leaves[4])
            (out[0].sum() + out[2].sum() + out[3][0].sum() + out[4].sum()).backward()
        torch.cuda.synchronize(dev)
