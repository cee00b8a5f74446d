"""CPU (-m "not gpu"): the N > 1 path -- one camera per rank, replicated Gaussians, ONE flat all-reduce of the
per-Gaussian gradients (gaustudio_amd/parallel.py) -- with the gloo backend and world_size 2.
Per-view gradients come from the CPU oracle here (there is no GPU); on the MI355X node the same code runs
over RCCL with gradients from the HIP kernels (bench.py --gpus N).  The invariant checked is the one the
north star states: all-reduced gradients == gradients accumulated over the views on one device."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gaustudio_amd import parallel, scenes

KEYS = ("means3D", "shs", "opacities", "scales", "rotations")
ORC = dict(means3D="dL_dmeans3D", shs="dL_dsh", opacities="dL_dopacity", scales="dL_dscales", rotations="dL_drotations")


def _view_grads(sc, cam):
    from oracle import pyoracle as po
    st = po.forward(sc.means3D.numpy(), sc.opacities.numpy(), cam.viewmatrix.numpy(), cam.projmatrix.numpy(),
                    cam.campos.numpy(), cam.width, cam.height, cam.tanfovx, cam.tanfovy, sh_degree=3,
                    shs=sc.shs.numpy(), scales=sc.scales.numpy(), rotations=sc.rotations.numpy())
    g = scenes.make_output_grads(cam, seed=5)
    bw = po.backward(st, *[t.numpy() for t in g], want_abs=False)
    return {k: torch.from_numpy(bw[ORC[k]].reshape(getattr(sc, k).shape).copy()) for k in KEYS}


def _scene_and_cams(n):
    sc = scenes.make_ball_scene(1500, radius=3.0, seed=4, sigma=0.06)
    return sc, scenes.ring_cameras(n, 96, 64, radius=8.0)


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc, cams = _scene_and_cams(world)
        params = [getattr(sc, k).clone().requires_grad_(True) for k in KEYS]
        bucket = parallel.FlatGradBucket(params)
        mine = parallel.shard_views(cams)
        assert len(mine) == 1

        def render(cam):
            for p, k in zip(params, KEYS):
                gk = _view_grads(sc, cam)[k]
                p.grad = gk if p.grad is None else p.grad + gk

        parallel.render_views_and_reduce(render, mine, bucket)
        assert bucket.nbytes == sum(p.numel() for p in params) * 4 == 1500 * 59 * 4
        for p, v in zip(params, bucket.views()):
            assert p.grad.data_ptr() == v.data_ptr()          # grads live in the flat buffer after the reduce
        torch.save([p.grad.clone() for p in params], os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_allreduced_grads_equal_single_device_accumulation(tmp_path, oracle):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    sc, cams = _scene_and_cams(world)
    per_view = [_view_grads(sc, c) for c in cams]
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"))
    for i, k in enumerate(KEYS):
        want = per_view[0][k] + per_view[1][k]                # accumulating both views on one device
        assert torch.equal(r0[i], r1[i]), k                   # every rank holds the same reduced gradient
        assert torch.allclose(r0[i], want, rtol=0, atol=1e-6 * float(want.abs().max())), k
        assert float(want.abs().max()) > 0


def _worker_chunked(rank, world, port, out_dir):
    """The overlapped reduction without a GPU: the hook the backward calls per SH chunk is driven by hand."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P, M = 1000, 16
        g = torch.Generator().manual_seed(100 + rank)
        shapes = dict(means3D=(P, 3), shs=(P, M, 3), opacities=(P, 1), scales=(P, 3), rotations=(P, 4))
        params = {k: torch.zeros(shp, requires_grad=True) for k, shp in shapes.items()}
        grads = {k: torch.randn(shp, generator=g) for k, shp in shapes.items()}
        bucket = parallel.FlatGradBucket(list(params.values()), roles=params)
        assert bucket.offsets[1] == 0                      # the SH gradients lead the buffer
        # what the armed backward does: MAIN part writes the small gradients, then SH ranges land chunk by chunk
        views = dict(zip(params, bucket.views()))
        for k in ("means3D", "opacities", "scales", "rotations"):
            views[k].copy_(grads[k])
        for c, (g0, g1) in enumerate(((0, 256), (256, 512), (512, 768), (768, 1000))):
            views["shs"][g0:g1].copy_(grads["shs"][g0:g1])
            bucket._on_sh_chunk(c, g0, g1)
        for k, p in params.items():
            p.grad = views[k]                              # autograd adopts the arena slices
        assert bucket.stats["chunks"] == 4 and bucket._reduced_upto == P * M * 3
        parallel.allreduce_gaussian_grads(bucket)
        assert bucket.stats["chunk_bytes"] + bucket.stats["tail_bytes"] == bucket.nbytes   # every element reduced exactly once
        torch.save({k: p.grad.clone() for k, p in params.items()}, os.path.join(out_dir, f"chunk_rank{rank}.pt"))
        torch.save(grads, os.path.join(out_dir, f"chunk_in{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_chunked_overlapped_reduction_equals_one_allreduce(tmp_path):
    """FlatGradBucket.arm(overlap_chunks=K): SH chunks reduced from inside the backward + one tail call == the sum over
    ranks of every gradient, bit for bit, and every element of the flat buffer is reduced exactly once."""
    world = 2
    mp.spawn(_worker_chunked, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    ins = [torch.load(os.path.join(tmp_path, f"chunk_in{r}.pt")) for r in range(world)]
    outs = [torch.load(os.path.join(tmp_path, f"chunk_rank{r}.pt")) for r in range(world)]
    for k in ins[0]:
        want = ins[0][k] + ins[1][k]
        assert torch.equal(outs[0][k], want) and torch.equal(outs[1][k], want), k


# ------------------------------------------------------------------------------------------------ factored exchange
def _sh_basis_torch(d, D):
    """SH basis of unit directions d[...,3] up to degree D (forward.cu:20-71 constants), float32 torch: the test-side
    stand-in for gsr_sh_grad_from_colors on CPU tensors."""
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    C0, C1 = 0.28209479177387814, 0.4886025119029199
    C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
    C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
          1.445305721320277, -0.5900435899266435)
    b = [torch.full_like(x, C0)]
    if D > 0:
        b += [-C1 * y, C1 * z, -C1 * x]
    if D > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        b += [C2[0] * xy, C2[1] * yz, C2[2] * (2 * zz - xx - yy), C2[3] * xz, C2[4] * (xx - yy)]
    if D > 2:
        b += [C3[0] * y * (3 * xx - yy), C3[1] * xy * z, C3[2] * y * (4 * zz - xx - yy), C3[3] * z * (2 * zz - 3 * xx - 3 * yy),
              C3[4] * x * (4 * zz - xx - yy), C3[5] * z * (xx - yy), C3[6] * x * (xx - 3 * yy)]
    return torch.stack(b, dim=-1)


def _sh_from_colors_torch(means3D, campos, colors, D, out):
    """out[P,M,3] = sum over views (ascending) of basis(dir) (x) colors[r] -- what gsr_sh_grad_from_colors computes."""
    out.zero_()
    nc = (D + 1) ** 2
    for r in range(colors.shape[0]):
        d = means3D - campos[r]
        d = d / d.norm(dim=1, keepdim=True)
        out[:, :nc] += _sh_basis_torch(d, D)[:, :, None] * colors[r][:, None, :]
    return out


def _factored_inputs(world, V, P=700, M=16):
    """Per-view colour gradients and geometry gradients (about a third of the Gaussians culled per view: zero rows)."""
    views = []
    for gview in range(world * V):
        g = torch.Generator().manual_seed(900 + gview)
        vis = (torch.rand(P, generator=g) > 0.35).float()[:, None]
        views.append(dict(colors=torch.randn(P, 3, generator=g) * vis, means3D=torch.randn(P, 3, generator=g) * vis,
                          opacities=torch.randn(P, 1, generator=g) * vis, scales=torch.randn(P, 3, generator=g) * vis,
                          rotations=torch.randn(P, 4, generator=g) * vis))
    g = torch.Generator().manual_seed(7)
    means = torch.randn(P, 3, generator=g)
    campos = torch.randn(world * V, 3, generator=g) * 5 + 20.0
    return views, means, campos, (P, M)


def _worker_factored(rank, world, port, out_dir, V, compact):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if isinstance(compact, str):
            return _worker_factored_by_view(rank, world, out_dir, V, compact)
        views, means, campos, (P, M) = _factored_inputs(world, V)
        shapes = dict(means3D=(P, 3), shs=(P, M, 3), opacities=(P, 1), scales=(P, 3), rotations=(P, 4))
        params = {k: (means.clone() if k == "means3D" else torch.zeros(shp)).requires_grad_(True) for k, shp in shapes.items()}
        fx = parallel.FactoredGradExchange(params, views_per_rank=V, compact=compact, sh_from_colors=_sh_from_colors_torch)
        assert fx.world == world and fx.colors.shape == (V, world, P, 3)
        assert fx.view_order() == [r * V + v for v in range(V) for r in range(world)]
        # what the armed backwards of this rank's V views leave behind: colour gradients in their slots, geometry
        # gradients accumulated in p.grad; the early hook (called by the real backward right after its geometry stage,
        # csrc/torch_binding.cpp) is driven by hand here: view v's all-gather starts before view v + 1 is "rendered"
        for v in range(V):
            gv = views[rank * V + v]
            fx.colors[v, rank].copy_(gv["colors"])
            if not compact:
                fx._on_colors_ready(v)
            for k in parallel.GEOMETRY_ROLES:
                params[k].grad = gv[k].clone() if params[k].grad is None else params[k].grad + gv[k]
        if not compact:
            assert fx.stats["early_allgathers"] == V and sorted(fx._works) == list(range(V))
        if rank == 0:          # a step armed with one degree and exchanged with another is refused
            fx._step_degrees = [2]
            with pytest.raises(ValueError, match="armed with degree 2"):
                fx._step_degree(3)
            fx._step_degrees = [1, 3]
            with pytest.raises(ValueError, match="different SH degrees"):
                fx._step_degree(None)
            with pytest.raises(ValueError, match="unknown"):
                fx._step_degree(None)
        fx.exchange(campos, sh_degree=3)
        pay = fx.payload()
        assert pay["dense_payload_bytes_per_rank"] == P * 59 * 4
        if not compact:
            assert pay["payload_bytes_per_rank"] == P * (44 + 12 * V)
        else:
            assert pay["rows_per_step"] < P                     # some Gaussians are culled in every view
        torch.save({k: p.grad.clone() for k, p in params.items()}, os.path.join(out_dir, f"fx_rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def _culled_everywhere(views):
    seen = torch.zeros(views[0]["colors"].shape[0], dtype=torch.bool)
    for gv in views:
        seen |= gv["colors"].abs().amax(1) > 0
    rows = torch.nonzero(~seen).flatten()
    assert rows.numel() > 0
    return int(rows[0])


def _worker_factored_by_view(rank, world, out_dir, V, compact):
    """compact="view" / "view+geometry" on CPU tensors: the packed-message primitives come from tests/packed_ref.py (the HIP
    kernels are held against the same restatement on the GPU: test_packed_messages_hip_vs_torch)."""
    from packed_ref import TorchPacked
    TorchPacked.sh_from_colors = staticmethod(_sh_from_colors_torch)
    views, means, campos, (P, M) = _factored_inputs(world, V)
    shapes = dict(means3D=(P, 3), shs=(P, M, 3), opacities=(P, 1), scales=(P, 3), rotations=(P, 4))
    params = {k: (means.clone() if k == "means3D" else torch.zeros(shp)).requires_grad_(True) for k, shp in shapes.items()}
    bands = 2 if compact.endswith("+bands2") else 1
    compact = compact.replace("+bands2", "")
    fx = parallel.FactoredGradExchange(params, views_per_rank=V, compact=compact.replace("+reg", ""), packed=TorchPacked, bands=bands,
                                       band_split=3 if bands == 2 else None)
    assert fx.by_view and fx.colors.shape == (V, P, 3) and fx.hdr.shape[0] == V * bands
    with pytest.raises(RuntimeError, match="call visible" if bands == 1 else "classes of local view 0 have not arrived"):
        fx._send_view(0)                                   # the radii (the classes) of the view have not arrived
    for v in range(V):
        gv = views[rank * V + v]
        vis = gv["colors"].abs().amax(1) > 0                         # culled rows of the inputs are all-zero rows
        fx.visible(v, vis.int() * 3)                                 # right after the "forward" (bands = 2: a no-op)
        if bands == 2:
            # the banded backward of the real binding, driven by hand: the two classes of the cut (here: an arbitrary, per-rank and
            # per-view different split of the visible rows), the first band's callback, then the second's
            first = vis & (((torch.arange(P) * 7 + rank + v) % 3) != 0)
            fx._on_classes(v, first.int(), (vis & ~first).int())
            fx.colors[v].zero_()
            fx.colors[v][first] = gv["colors"][first]                # only class 1 is final when the first band reports
            fx._on_band_ready(v)
        fx.colors[v].copy_(gv["colors"])
        fx._on_colors_ready(v)                                       # the backward's callback: pack + all-gather start
        for k in parallel.GEOMETRY_ROLES:
            params[k].grad = gv[k].clone() if params[k].grad is None else params[k].grad + gv[k]
    assert fx.stats["early_allgathers"] == V and fx.stats.get("early_band_allgathers", 0) == (V if bands == 2 else 0)
    assert fx.payload()["bands"] == bands
    regularised = compact == "view+geometry+reg"
    if regularised and rank == 1:
        # another loss term (a scale regulariser, say) leaves a gradient on a Gaussian NO view of the step sees -- on one rank only
        params["scales"].grad[_culled_everywhere(views), 1] += 0.25
    fx.exchange(campos, sh_degree=3)
    pay = fx.payload()
    assert pay["color_rows_per_view"] < 0.8 * P and pay["allgather_bytes_sent"] < 12 * P * V      # a third of the rows are culled per view
    if fx.union_geometry:
        assert 0 < pay["geometry_rows"] <= P
        # detected on every rank (the flag is agreed), and that step summed the whole dense block
        assert pay["geometry_fallbacks"] == (1 if regularised else 0) and (pay["geometry_rows"] == P) == regularised
    torch.save({k: p.grad.clone() for k, p in params.items()}, os.path.join(out_dir, f"fx_rank{rank}.pt"))


@pytest.mark.parametrize("V,compact", [(1, False), (2, False), (1, True), (1, "view"), (2, "view"), (2, "view+geometry"),
                                       (1, "view+geometry+reg"), (1, "view+bands2"), (2, "view+geometry+bands2")])
def test_factored_exchange_equals_dense_accumulation(tmp_path, V, compact):
    """FactoredGradExchange (all-gather of per-view colour gradients + all-reduce of the geometry block + local rebuild of the
    SH gradient) == the sum over all world * V views of the dense per-view gradients, on every rank; with and without
    visible-row compaction."""
    world = 2
    mp.spawn(_worker_factored, args=(world, _free_port(), str(tmp_path), V, compact), nprocs=world, join=True)
    views, means, campos, (P, M) = _factored_inputs(world, V)
    want = {k: sum(v[k] for v in views) for k in parallel.GEOMETRY_ROLES}
    # "+bands2" (round 6): every view leaves as TWO messages (the banded backward's classes) -- the very same sums, bit for bit
    if compact == "view+geometry+reg":     # "+reg": rank 1 adds a regulariser gradient on a row culled in every view:
        want["scales"][_culled_everywhere(views), 1] += 0.25   # it must arrive in the sum (the step falls back to the dense block)
    order = [r * V + v for v in range(V) for r in range(world)]          # FactoredGradExchange.view_order(): local view major, rank minor
    want["shs"] = _sh_from_colors_torch(means, campos[order], torch.stack([views[g]["colors"] for g in order]), 3, torch.zeros(P, M, 3))
    got = [torch.load(os.path.join(tmp_path, f"fx_rank{r}.pt")) for r in range(world)]
    for k in want:
        assert torch.equal(got[0][k], got[1][k]), k                           # replicated result
        tol = 0.0 if k == "shs" else 1e-6 * float(want[k].abs().max())         # geometry: (a + b) + c vs the ring's order; dL_dsh BIT-equal
        assert float((got[0][k] - want[k]).abs().max()) <= tol, k
    assert float(want["shs"].abs().max()) > 0


def test_bucket_roundtrip_and_view_sharding_without_process_group():
    ps = [torch.randn(5, 3, requires_grad=True), torch.randn(5, 16, 3, requires_grad=True), torch.randn(5, 1, requires_grad=True)]
    b = parallel.FlatGradBucket(ps)
    ps[0].grad = torch.ones(5, 3)
    ps[2].grad = torch.full((5, 1), 2.0)
    assert parallel.allreduce_gaussian_grads(b) is None        # world_size 1: nothing to exchange, nothing copied
    assert float(b.flat.abs().sum()) == 0.0 and ps[1].grad is None
    b.pack()
    b.unpack()
    assert b.flat.numel() == 15 + 240 + 5
    assert float(b.flat[:15].sum()) == 15.0 and float(b.flat[15:255].abs().sum()) == 0.0 and float(b.flat[255:].sum()) == 10.0
    assert ps[1].grad is not None and float(ps[1].grad.abs().sum()) == 0.0
    assert parallel.shard_views(list(range(8)), rank=3, world_size=8) == [3]
    assert parallel.shard_views(list(range(8)), rank=1, world_size=4) == [1, 5]
    with pytest.raises(ValueError):
        parallel.FlatGradBucket([])


# ------------------------------------------------------------------------------------------------ GPU, 2 ranks on 1 device
def _gpu_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)     # RCCL refuses two ranks on one device; gloo stages through the host
    try:
        from gaustudio_amd import GaussianRasterizationSettings, GaussianRasterizer
        dev = torch.device("cuda", 0)
        sc, cams = _scene_and_cams(world)
        params = {k: getattr(sc, k).to(dev).requires_grad_(True) for k in KEYS}
        bucket = parallel.FlatGradBucket(list(params.values()), roles=params)
        m2 = torch.zeros_like(params["means3D"])

        def render(cam):
            rs = GaussianRasterizationSettings(cam.height, cam.width, cam.tanfovx, cam.tanfovy, torch.zeros(3), 1.0,
                                               cam.viewmatrix.to(dev), cam.projmatrix.to(dev), 3, cam.campos.to(dev), False, False)
            out = GaussianRasterizer(rs)(means3D=params["means3D"], means2D=m2, opacities=params["opacities"], shs=params["shs"],
                                         scales=params["scales"], rotations=params["rotations"])
            g = [t.to(dev) for t in scenes.make_output_grads(cam, seed=5)]
            torch.autograd.backward([out[0], out[2], out[3], out[4]], g)

        # overlap_chunks=3: the SH stage of the backward runs in Gaussian ranges and each range's slice of the flat
        # buffer is all-reduced from inside the backward (csrc/torch_binding.cpp -> FlatGradBucket._on_sh_chunk)
        parallel.render_views_and_reduce(render, parallel.shard_views(cams), bucket, overlap_chunks=3)
        assert bucket.stats["chunks"] == 3, bucket.stats
        assert bucket.stats["chunk_bytes"] + bucket.stats["tail_bytes"] == bucket.nbytes
        for p, v in zip(bucket.params, bucket.views()):
            assert p.grad.data_ptr() == v.data_ptr()
        torch.save([p.grad.cpu() for p in bucket.params], os.path.join(out_dir, f"gpu_rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_real_kernels_gradients_born_in_the_bucket(tmp_path):
    """The N > 1 step with the HIP kernels: each rank renders its camera, the backward writes into the armed flat bucket
    and reduces its SH gradients chunk by chunk while it runs, one tail call finishes the reduction; result == both
    views accumulated in one process (fp32 sum of two terms: exact up to commutation)."""
    from gaustudio_amd import GaussianRasterizationSettings, GaussianRasterizer
    world = 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_gpu_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = [torch.load(os.path.join(tmp_path, f"gpu_rank{r}.pt")) for r in range(world)]
    for a, b in zip(got[0], got[1]):
        assert torch.equal(a, b)                                   # both ranks hold the same reduced gradients
    dev = torch.device("cuda", 0)
    sc, cams = _scene_and_cams(world)
    params = {k: getattr(sc, k).to(dev).requires_grad_(True) for k in KEYS}
    for cam in cams:                                              # single process: autograd accumulates the two views
        rs = GaussianRasterizationSettings(cam.height, cam.width, cam.tanfovx, cam.tanfovy, torch.zeros(3), 1.0,
                                           cam.viewmatrix.to(dev), cam.projmatrix.to(dev), 3, cam.campos.to(dev), False, False)
        out = GaussianRasterizer(rs)(means3D=params["means3D"], means2D=torch.zeros_like(params["means3D"]),
                                     opacities=params["opacities"], shs=params["shs"], scales=params["scales"],
                                     rotations=params["rotations"])
        g = [t.to(dev) for t in scenes.make_output_grads(cam, seed=5)]
        torch.autograd.backward([out[0], out[2], out[3], out[4]], g)
    for k, a in zip(KEYS, got[0]):
        assert torch.equal(a, params[k].grad.cpu()), k


# ------------------------------------------------------------------------------------------------ GPU: factored exchange with the real kernels
def _accumulate_views_dense(sc, cams, dev, D=3):
    """Both views one after the other in one process, autograd accumulating: the north star's reference for the exchange."""
    from gaustudio_amd import GaussianRasterizationSettings, GaussianRasterizer
    params = {k: getattr(sc, k).to(dev).requires_grad_(True) for k in KEYS}
    for cam in cams:
        rs = GaussianRasterizationSettings(cam.height, cam.width, cam.tanfovx, cam.tanfovy, torch.zeros(3), 1.0,
                                           cam.viewmatrix.to(dev), cam.projmatrix.to(dev), D, cam.campos.to(dev), False, False)
        out = GaussianRasterizer(rs)(means3D=params["means3D"], means2D=torch.zeros_like(params["means3D"]),
                                     opacities=params["opacities"], shs=params["shs"], scales=params["scales"],
                                     rotations=params["rotations"])
        g = [t.to(dev) for t in scenes.make_output_grads(cam, seed=5)]
        torch.autograd.backward([out[0], out[2], out[3], out[4]], g)
    return {k: p.grad.cpu() for k, p in params.items()}


def _factored_step(sc, my_cams, all_cams, dev, V, D=3, compact=False, bands=1):
    from gaustudio_amd import GaussianRasterizationSettings, GaussianRasterizer
    params = {k: getattr(sc, k).to(dev).requires_grad_(True) for k in KEYS}
    # bands = 2: every backward runs banded, cut in the middle of the image (tile rows)
    fx = parallel.FactoredGradExchange(params, views_per_rank=V, compact=compact, bands=bands,
                                       band_split=((my_cams[0].height + 15) // 16) // 2 if bands == 2 else None)
    m2 = torch.zeros_like(params["means3D"])
    for v, cam in enumerate(my_cams):
        rs = GaussianRasterizationSettings(cam.height, cam.width, cam.tanfovx, cam.tanfovy, torch.zeros(3), 1.0,
                                           cam.viewmatrix.to(dev), cam.projmatrix.to(dev), D, cam.campos.to(dev), False, False)
        fx.arm(v, sh_degree=D)
        out = GaussianRasterizer(rs)(means3D=params["means3D"], means2D=m2, opacities=params["opacities"], shs=params["shs"],
                                     scales=params["scales"], rotations=params["rotations"])
        fx.visible(v, out[1])                                  # compact="view": the view's radii -> header + count gather (no-op otherwise)
        g = [t.to(dev) for t in scenes.make_output_grads(cam, seed=5)]
        torch.autograd.backward([out[0], out[2], out[3], out[4]], g)
        assert params["shs"].grad is None                      # the factored backward produces no dL_dsh
    if V >= 1 and fx.geo.is_cuda:
        gv = fx.geo_views()
        assert all(params[r].grad.data_ptr() == gv[r].data_ptr() for r in parallel.GEOMETRY_ROLES)   # born in the all-reduce buffer
    if (parallel._multi(None) and not compact) or isinstance(compact, str):
        assert fx.stats["early_allgathers"] == V       # every view's all-gather started from inside its backward
    if bands == 2:
        assert fx.stats["early_band_allgathers"] == V  # ... and the first band's message before the second band was enqueued
    fx.exchange(torch.stack([c.campos for c in all_cams]).to(dev), sh_degree=D)
    return {k: p.grad.cpu() for k, p in params.items()}, fx


@pytest.mark.gpu
def test_banded_factored_exchange_single_process_bit_equal():
    """bands = 2 (round 6): the banded backward + two packed messages per view give, bit for bit, what plain autograd accumulation of
    the dense backward gives -- world_size 1, two local views, both packed forms."""
    dev = torch.device("cuda", 0)
    sc, cams = _scene_and_cams(2)
    want = _accumulate_views_dense(sc, cams, dev, 3)
    for compact in ("view", "view+geometry"):
        got, fx = _factored_step(sc, cams, cams, dev, V=2, D=3, compact=compact, bands=2)
        for k in KEYS:
            assert torch.equal(got[k], want[k]), (compact, k)
        pay = fx.payload()
        assert pay["bands"] == 2 and 0 < pay["color_rows_per_view"] <= 1500
    with pytest.raises(ValueError, match="bands=2 needs the packed"):
        parallel.FactoredGradExchange({k: getattr(sc, k).to(dev) for k in KEYS}, bands=2)


@pytest.mark.gpu
@pytest.mark.parametrize("D", [0, 3])
def test_factored_exchange_single_process_bit_equal(D):
    """world_size 1, two local views: the factored path (colour gradients -> gsr_sh_grad_from_colors, geometry gradients
    accumulated in the exchange buffer) gives bit for bit what plain autograd accumulation of the dense backward gives."""
    dev = torch.device("cuda", 0)
    sc, cams = _scene_and_cams(2)
    want = _accumulate_views_dense(sc, cams, dev, D)
    got, fx = _factored_step(sc, cams, cams, dev, V=2, D=D)
    for k in KEYS:
        assert torch.equal(got[k], want[k]), k
    assert float(want["shs"].abs().max()) > 0
    assert fx.payload()["payload_bytes_per_rank"] == 1500 * (44 + 24) and fx.payload()["dense_payload_bytes_per_rank"] == 1500 * 236


def _gpu_worker_factored(rank, world, port, out_dir, compact, bands=1):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)     # RCCL refuses two ranks on one device
    try:
        dev = torch.device("cuda", 0)
        sc, cams = _scene_and_cams(world)
        got, fx = _factored_step(sc, parallel.shard_views(cams), cams, dev, V=1, compact=compact, bands=bands)
        if compact is True:
            assert fx.payload()["rows_per_step"] <= 1500
        elif compact:
            # (bands = 2: each of a view's two messages is padded to ITS largest row count over the ranks)
            assert fx.payload()["color_rows_per_view"] <= 1500 * bands and fx.payload()["compacted"] == compact
        torch.save(got, os.path.join(out_dir, f"fxgpu_rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("compact", ["view", "view+geometry"])
def test_two_ranks_banded_factored_exchange_real_kernels(tmp_path, compact):
    """bands = 2 over two ranks (gloo, one GPU), one camera each, the HIP kernels: every rank ends with the gradients of both
    views, bit-identical to single-process accumulation and to the unbanded exchange."""
    world = 2
    mp.spawn(_gpu_worker_factored, args=(world, _free_port(), str(tmp_path), compact, 2), nprocs=world, join=True)
    got = [torch.load(os.path.join(tmp_path, f"fxgpu_rank{r}.pt")) for r in range(world)]
    sc, cams = _scene_and_cams(world)
    want = _accumulate_views_dense(sc, cams, torch.device("cuda", 0))
    for k in KEYS:
        assert torch.equal(got[0][k], got[1][k]), k
        assert torch.equal(got[0][k], want[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("compact", [False, True, "view", "view+geometry"])
def test_two_ranks_factored_exchange_real_kernels(tmp_path, compact):
    """Two ranks (gloo, one GPU), one camera each, the HIP kernels: after FactoredGradExchange every rank holds the gradients
    of both views -- dL_dsh bit-identical to single-process accumulation (same arithmetic, same view order), the geometry
    gradients equal (a sum of two terms commutes)."""
    world = 2
    mp.spawn(_gpu_worker_factored, args=(world, _free_port(), str(tmp_path), compact), nprocs=world, join=True)
    got = [torch.load(os.path.join(tmp_path, f"fxgpu_rank{r}.pt")) for r in range(world)]
    sc, cams = _scene_and_cams(world)
    want = _accumulate_views_dense(sc, cams, torch.device("cuda", 0))
    for k in KEYS:
        assert torch.equal(got[0][k], got[1][k]), k
        assert torch.equal(got[0][k], want[k]), k


# ------------------------------------------------------------------------------------------------ GPU: the RCCL calls themselves, world_size 1
def _nccl_world1_worker(rank, world, port, out_dir):
    """Every collective call of the N > 1 step on backend "nccl" (= RCCL) with a process group of ONE rank on the leased GPU:
    init_process_group(device_id=...), the armed flat bucket with SH chunks reduced from inside the backward (async work
    handles on the communicator's stream), the in-place all_gather_into_tensor, the geometry all-reduce, the mask all-reduce
    of the compacted form.  A sum / gather over one rank is the identity, so every result must equal the no-collective
    run BIT FOR BIT."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        from gaustudio_amd import GaussianRasterizationSettings, GaussianRasterizer
        parallel.FORCE_COLLECTIVES = True
        sc, cams = _scene_and_cams(2)
        report = {}
        # --- dense: gradients born in the flat bucket, SH stage in 4 ranges, each all-reduced asynchronously from the hook
        params = {k: getattr(sc, k).to(dev).requires_grad_(True) for k in KEYS}
        bucket = parallel.FlatGradBucket(list(params.values()), roles=params)
        m2 = torch.zeros_like(params["means3D"])

        def render(cam):
            rs = GaussianRasterizationSettings(cam.height, cam.width, cam.tanfovx, cam.tanfovy, torch.zeros(3), 1.0,
                                               cam.viewmatrix.to(dev), cam.projmatrix.to(dev), 3, cam.campos.to(dev), False, False)
            out = GaussianRasterizer(rs)(means3D=params["means3D"], means2D=m2, opacities=params["opacities"], shs=params["shs"],
                                         scales=params["scales"], rotations=params["rotations"])
            g = [t.to(dev) for t in scenes.make_output_grads(cam, seed=5)]
            torch.autograd.backward([out[0], out[2], out[3], out[4]], g)

        parallel.render_views_and_reduce(render, cams[:1], bucket, overlap_chunks=4)
        torch.cuda.synchronize()
        # 1500 Gaussians in 4 requested ranges of multiples of 256 -> 512 + 512 + 476: three chunk collectives + the tail
        assert bucket.stats["chunks"] == 3 and bucket.stats["chunk_bytes"] + bucket.stats["tail_bytes"] == bucket.nbytes, bucket.stats
        report["dense"] = {k: p.grad.cpu() for k, p in params.items()}
        # two local views: no chunking (a later local accumulation would follow the partial reduction), one all-reduce
        parallel.render_views_and_reduce(render, cams, bucket, overlap_chunks=4)
        torch.cuda.synchronize()
        report["dense2"] = {k: p.grad.cpu() for k, p in params.items()}
        # --- factored: colour all-gather (in place) + geometry all-reduce, with and without visible-row compaction
        for compact, tag in ((False, "0"), (True, "1"), ("view", "view"), ("view+geometry", "viewgeo")):
            got, fx = _factored_step(sc, cams[:1], cams[:1], dev, V=1, compact=compact)
            assert fx.world == 1 and parallel._multi(None)
            report[f"factored_compact{tag}"] = got
            got2, _ = _factored_step(sc, cams, cams, dev, V=2, compact=compact)
            report[f"factored2_compact{tag}"] = got2
            if isinstance(compact, str):          # bands = 2: the banded backward, two messages per view
                report[f"banded_compact{tag}"], _ = _factored_step(sc, cams[:1], cams[:1], dev, V=1, compact=compact, bands=2)
                report[f"banded2_compact{tag}"], _ = _factored_step(sc, cams, cams, dev, V=2, compact=compact, bands=2)
        torch.save(report, os.path.join(out_dir, "nccl1.pt"))
    finally:
        dist.destroy_process_group()


def _spawn_with_timeout(fn, args, nprocs, timeout_s):
    ctx = mp.spawn(fn, args=args, nprocs=nprocs, join=False)
    import time
    t0 = time.time()
    while not ctx.join(timeout=5):
        if time.time() - t0 > timeout_s:
            for p in ctx.processes:
                if p.is_alive():
                    p.kill()
            pytest.fail(f"worker did not finish within {timeout_s} s (RCCL initialisation or a collective hangs)")


@pytest.mark.gpu
def test_nccl_world1_every_collective_call_of_the_step(tmp_path):
    """backend "nccl" (RCCL), world_size 1, on the leased GPU: the dense step (arm(overlap_chunks=4): chunk hooks + tail) and
    the factored step (FactoredGradExchange.exchange, with and without `compact`, V = 1 and 2) give gradients bit-equal to
    the run without any collective.  The N > 1 path had only ever met gloo (missing #2)."""
    _spawn_with_timeout(_nccl_world1_worker, (1, _free_port(), str(tmp_path)), 1, 240)
    got = torch.load(os.path.join(tmp_path, "nccl1.pt"))
    dev = torch.device("cuda", 0)
    sc, cams = _scene_and_cams(2)
    want1 = _accumulate_views_dense(sc, cams[:1], dev)
    want2 = _accumulate_views_dense(sc, cams, dev)
    for name, want in (("dense", want1), ("dense2", want2), ("factored_compact0", want1), ("factored_compact1", want1),
                       ("factored2_compact0", want2), ("factored2_compact1", want2), ("factored_compactview", want1),
                       ("factored2_compactview", want2), ("factored_compactviewgeo", want1), ("factored2_compactviewgeo", want2),
                       ("banded_compactview", want1), ("banded2_compactview", want2), ("banded_compactviewgeo", want1),
                       ("banded2_compactviewgeo", want2)):
        for k in KEYS:
            assert torch.equal(got[name][k], want[k]), (name, k)
    assert float(want1["shs"].abs().max()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("exchange,compact", [("dense", "none"), ("factored", "none"), ("factored", "view"), ("factored", "view+geometry"),
                                              ("factored", "view/bands2")])
def test_nccl_world1_through_bench_py(tmp_path, exchange, compact):
    """`bench.py --gpus 1` with GSR_BENCH_FORCE_PG=1: the exact code path of the driver's N > 1 runs (init_process_group("nccl",
    device_id=...), armed bucket / FactoredGradExchange, barrier + MAX all-reduce of the clock, comm block) on one GPU.
    The line must carry a `comm` block of the requested exchange WITHOUT a fallback."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GSR_BENCH_FORCE_PG="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    bands = 2 if compact.endswith("/bands2") else 1          # round 6: `--bands 2`, the banded exchange (comm.bands on the line)
    compact = compact.split("/")[0]
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--workload", "C2", "--steps", "4", "--warmup", "2",
                        "--exchange", exchange, "--compact", compact, "--bands", str(bands), "--no-cpu-baseline", "--no-ref-ab"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["comm"]["exchange"] == exchange and line["comm"]["exchange_fallback"] is None, line["comm"]
    assert line["n_gpus"] == 1 and line["value"] > 0
    if exchange == "dense":
        assert line["comm"]["chunks_per_step"] == 4 and line["comm"]["tail_bytes_per_step"] > 0
    elif compact != "none":
        assert line["comm"]["compacted"] == compact and line["comm"]["color_rows_per_view"] < line["comm"]["rows_total"]
        assert line["comm"]["early_allgathers_per_step"] == 1 and line["comm"]["bands"] == bands
        if bands == 2:
            assert 0 < line["comm"]["first_band_rows_per_view"] < line["comm"]["color_rows_per_view"]


@pytest.mark.gpu
def test_bench_py_gpus2_launches_its_own_ranks():
    """`python bench.py --gpus 2 ...` as a PLAIN subprocess -- no torchrun, no WORLD_SIZE: bench.py re-runs itself under
    torch.distributed.run (the driver starts N = 1 that way, and a SCALE run that did the same for N > 1 used to
    die with a usage message).  GSR_BENCH_BACKEND=gloo: both ranks share the leased GPU (RCCL refuses two ranks on one device).
    One JSON line: n_gpus 2, a `comm` block for the factored headline and the dense exchange beside it (`variants.dense`)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(GSR_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "C2", "--steps", "3", "--warmup", "1",
                        "--settle", "0", "--no-cpu-baseline", "--no-ref-ab"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["value"] > 0
    assert line["config"]["views_per_step"] == 2
    assert line["comm"]["exchange"] == "factored" and line["comm"]["exchange_fallback"] is None, line["comm"]
    dense = line["variants"]["dense"]
    assert "error" not in dense and dense["value"] > 0 and dense["payload_bytes_per_rank"] == 300_000 * 59 * 4, dense


def test_bench_py_self_launch_command(monkeypatch):
    """The launcher half of the test above without a GPU: `--gpus 4` with no WORLD_SIZE in the environment becomes
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node=4 --master-addr 127.0.0.1 --master-port <free> bench.py <same argv>`
    and bench.py exits with the launcher's return code; under a launcher (WORLD_SIZE set) it never re-launches."""
    import importlib
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.syspath_prepend(root)
    bench = importlib.import_module("bench")
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("MASTER_PORT", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.join(root, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    # under a launcher: no re-launch (it goes on and, on this CPU box, stops at the GPU check)
    if not torch.cuda.is_available():
        seen.clear()
        monkeypatch.setenv("WORLD_SIZE", "4")
        with pytest.raises(SystemExit) as e:
            bench.main()
        assert not seen and "ROCm GPU" in str(e.value.code)


@pytest.mark.gpu
@pytest.mark.parametrize("P", [1, 255, 256, 257, 5000, 100003])
def test_packed_messages_hip_vs_torch(P):
    """The packed row messages of compact="view" (csrc/gsr_comm.hip): header (count, block bases, mask), row packing /
    unpacking, the union header and the SH-gradient rebuild from packed messages -- HIP kernels against the torch restatement
    (tests/packed_ref.py) word for word, and the packed rebuild against the dense one bit for bit."""
    from gaustudio_amd import _C
    from packed_ref import TorchPacked as T
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(P)
    Hw = _C.msg_header_words(P)
    assert Hw == T.header_words(P)
    N = 3
    radii = [((torch.rand(P, generator=g) > f).int() * 7).to(dev) for f in (0.8, 0.3, 0.55)]
    scratch = torch.zeros((P + 255) // 256, dtype=torch.int32, device=dev)
    Lmax = Hw + 3 * P
    msgs = torch.zeros(N * Lmax, dtype=torch.int32, device=dev)
    offsets, cols = [], []
    for r in range(N):
        hdr, ref = torch.zeros(Hw, dtype=torch.int32, device=dev), torch.zeros(Hw, dtype=torch.int32, device=dev)
        _C.visible_index(radii[r], hdr, scratch)
        T.visible_index(radii[r], ref, None)
        assert torch.equal(hdr, ref), r
        K = int(hdr[0])
        assert K == int((radii[r] > 0).sum())
        col = (torch.randn(P, 3, generator=g).to(dev)) * (radii[r] > 0)[:, None]
        cols.append(col)
        off = r * Lmax
        offsets.append(off)
        msgs[off:off + Hw] = hdr
        want_rows = torch.zeros(3 * K, device=dev)
        T.pack_rows(hdr, col, want_rows, 3, 0)
        _C.pack_rows(hdr, col, msgs[off + Hw:off + Hw + 3 * P].view(torch.float32), 3, 0)
        assert torch.equal(msgs[off + Hw:off + Hw + 3 * K].view(torch.float32), want_rows)
        back = torch.full((P, 3), 7.0, device=dev)
        _C.unpack_rows(hdr, msgs[off + Hw:].view(torch.float32), 3, 0, back)
        assert torch.equal(back[radii[r] > 0], col[radii[r] > 0]) and bool((back[radii[r] == 0] == 7.0).all())
    off_t = torch.tensor(offsets, dtype=torch.int64, device=dev)
    hu, hu_ref = torch.zeros(Hw, dtype=torch.int32, device=dev), torch.zeros(Hw, dtype=torch.int32, device=dev)
    _C.union_index(P, msgs, off_t, hu, scratch)
    T.union_index(P, msgs, off_t, hu_ref, None)
    assert torch.equal(hu, hu_ref) and int(hu[0]) == int(((radii[0] > 0) | (radii[1] > 0) | (radii[2] > 0)).sum())
    # geometry-style packing with a stride and a column offset
    geo = torch.randn(P, 4, generator=g).to(dev)
    K = int(hu[0])
    if K > 0:
        rows = torch.zeros((K, 11), device=dev)
        _C.pack_rows(hu, geo, rows.view(-1), 11, 7)
        assert torch.equal(rows[:, 7:11], geo[T._mask(hu, P)]) and float(rows[:, :7].abs().sum()) == 0.0
    # the geometry block in one pass each way (round 5): HIP == torch restatement, flag only for non-zero rows OUTSIDE the header
    gv = dict(means3D=torch.randn(P, 3, generator=g).to(dev), opacities=torch.randn(P, 1, generator=g).to(dev),
              scales=torch.randn(P, 3, generator=g).to(dev), rotations=torch.randn(P, 4, generator=g).to(dev))
    um = T._mask(hu, P)
    for k in gv:
        gv[k] = (gv[k] * um[:, None]).contiguous()                      # rasterizer-like: zero rows outside the union
    rows_h, rows_t = torch.full((P, 11), 7.0, device=dev), torch.full((P, 11), 7.0, device=dev)
    fl_h, fl_t = torch.zeros(1, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
    _C.pack_geometry(hu, gv["means3D"], gv["opacities"], gv["scales"], gv["rotations"], rows_h, fl_h)
    T.pack_geometry(hu, gv, rows_t, fl_t)
    assert torch.equal(rows_h, rows_t) and int(fl_h) == 0 == int(fl_t)
    back = {k: torch.full_like(v, 3.0) for k, v in gv.items()}
    _C.unpack_geometry(hu, rows_h, back["means3D"], back["opacities"], back["scales"], back["rotations"])
    for k in gv:
        assert torch.equal(back[k][um], gv[k][um]) and bool((back[k][~um] == 3.0).all()), k
    if int((~um).sum()) > 0:                                            # a regulariser's gradient on a Gaussian no view sees
        gv["scales"][int(torch.nonzero(~um)[0]), 2] = 0.5
        _C.pack_geometry(hu, gv["means3D"], gv["opacities"], gv["scales"], gv["rotations"], rows_h, fl_h)
        assert int(fl_h) == 1
    # rebuild: packed == dense, bit for bit, at every degree
    means = torch.randn(P, 3, generator=g).to(dev)
    campos = (torch.randn(N, 3, generator=g) * 4 + 15).to(dev)
    for D in (0, 1, 2, 3):
        for M in ((D + 1) ** 2, 16):
            a = torch.full((P, M, 3), float("nan"), device=dev)
            b = torch.full((P, M, 3), float("nan"), device=dev)
            _C.sh_grad_from_colors(means, campos, torch.stack(cols).contiguous(), D, a)
            _C.sh_grad_from_packed(means, campos, msgs, off_t, D, b)
            assert torch.equal(a, b), (D, M)
