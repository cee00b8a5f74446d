"""Run as a script by tests/test_gpu_forward.py::test_trailing_chunks_do_not_read_past_the_geometry_buffer, in a process whose
torch allocator does not cache (PYTORCH_NO_CUDA_MEMORY_CACHING=1: every opaque buffer is its own hipMalloc, nothing of this
process sits behind it), so that a read megabytes past the geometry buffer meets unmapped memory instead of a pooled block."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GSR_FAST_EXP", "0")


def main():
    import torch
    from gaustudio_amd import scenes
    from oracle import pyoracle as po   # checker
    from util import compare_forward_exact, hip_forward, oracle_forward, scene_kwargs
    po.build()
    for arg in sys.argv[1:]:
        P, W, H = (int(x) for x in arg.split("x"))
        cam = scenes.make_camera(W, H)
        sc = scenes.make_scene(P, cam, seed=P % 97, sigma_px_median=0.6)
        kw = scene_kwargs(sc, True, False)
        hs = hip_forward(sc, cam, 1, kw)
        torch.cuda.synchronize()
        compare_forward_exact(hs, oracle_forward(po, sc, cam, 1, kw))
        print(f"ok {P} {W}x{H} R={hs['num_rendered']}", flush=True)


if __name__ == "__main__":
    main()
