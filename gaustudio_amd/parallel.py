"""Multi-GPU data parallelism for the rasterizer: one camera per GPU, replicated Gaussians, ONE logical
exchange of the per-Gaussian gradients after backward (BASELINE.json north_star): either the dense
all-reduce of the flat 236 B/Gaussian buffer (FlatGradBucket / allreduce_gaussian_grads) or its factored form
(FactoredGradExchange: 44 B/Gaussian all-reduced + 12 B/Gaussian/view all-gathered, same sums).

The reference has no distributed code at all ("Collective / NCCL call sites: none");
this is new design for an 8 x MI355X node: one process per GPU, torch.distributed backend "nccl"
(= RCCL over xGMI on ROCm).  Views are independent given replicated parameters, so forward and
backward need no exchange; the only collective is a SUM over ranks of a single flat fp32 buffer
[sh 3M | means3D 3 | opacity 1 | scales 3 | rotations 4] = 59 floats (236 B) per Gaussian at M = 16,
which equals accumulating the N views on one GPU.

Volume and overlap (DESIGN.md s7 has the alpha-beta model).  xGMI is point-to-point (7 links x ~153 GB/s per GPU):
a ring all-reduce moves 2 (N-1)/N x 236 B per Gaussian over every link, so the collective costs about as much as
the whole forward + backward and must not simply be appended to it.  81 % of the payload (dL_dsh) is produced by the
LAST kernel of the backward (the SH stage), which is independent per Gaussian: `arm(overlap_chunks=K)` makes that
backward run the SH stage in K Gaussian ranges and reduce each range's slice of the flat buffer as soon as its kernel
is enqueued (async, on the communicator's stream), so all but the last chunk's transfer and the 19 % tail (means /
opacity / scale / rotation gradients) hide behind compute.  It stays one logical reduction of one flat buffer: the
chunks partition it, every element is reduced exactly once, and the result is bit-identical to the single call.

The same code runs on CPU tensors with the "gloo" backend (tests/test_distributed.py, world_size 2).
"""
import os
from typing import Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist


ARENA_ROLES = ("means3D", "shs", "opacities", "scales", "rotations")
_KEY_ROLES = ("means3D", "shs", "scales", "rotations")      # the operator inputs the backward sees (arena key)


# True: take the collective code paths even in a process group of ONE rank (a sum / gather over one rank is the identity).
# For exercising every RCCL call of the N > 1 step on a single GPU (tests, bench.py with GSR_BENCH_FORCE_PG=1).
FORCE_COLLECTIVES = os.environ.get("GSR_FORCE_COLLECTIVES") == "1"


def _multi(group=None):
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or FORCE_COLLECTIVES)


class FlatGradBucket:
    """A persistent flat fp32 buffer holding the gradients of `params` back to back.

    `roles` (optional) names which parameter plays which operator input (keys of ARENA_ROLES).  With roles,
    `arm()` makes the NEXT rasterizer backward OF THESE PARAMETERS write its gradients straight into this buffer,
    so that autograd adopts slices of it as `p.grad` and the all-reduce needs no pack copy (236 MB read + written
    per step at 1 M Gaussians otherwise).  The `shs` parameter is laid out FIRST, so that the chunks reduced while
    the backward is still running are a contiguous prefix and what remains for the final call is one contiguous tail.
    Anything that does not end up aliased (another op in the graph, autograd deciding to copy) is still handled by
    pack()."""

    def __init__(self, params: Sequence[torch.Tensor], roles: Optional[dict] = None):
        self.params: List[torch.Tensor] = list(params)
        self.roles = None
        if not self.params:
            raise ValueError("FlatGradBucket needs at least one parameter")
        idx = {id(p): i for i, p in enumerate(self.params)}
        if roles is not None:
            if set(roles) != set(ARENA_ROLES):
                raise ValueError(f"roles must name exactly {ARENA_ROLES}")
            if any(id(roles[r]) not in idx for r in ARENA_ROLES):
                raise ValueError("every role parameter must be one of the bucket's parameters")
            self.roles = [idx[id(roles[r])] for r in ARENA_ROLES]
            self._key_params = [roles[r] for r in _KEY_ROLES]
        dev = self.params[0].device
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("all parameters must be float32 on one device")
        self.sizes = [p.numel() for p in self.params]
        # storage order: the SH parameter first (see class docstring), the others as given
        order = list(range(len(self.params)))
        self._sh = self.roles[1] if self.roles is not None else None
        if self._sh is not None:
            order.remove(self._sh)
            order.insert(0, self._sh)
        self.offsets = [0] * len(self.params)
        o = 0
        for i in order:
            self.offsets[i] = o
            o += self.sizes[i]
        self.flat = torch.zeros(o, dtype=torch.float32, device=dev)
        self._works = []          # async chunk reductions issued from inside the backward
        self._reduced_upto = 0    # elements [0, _reduced_upto) of flat are already (being) reduced
        self._group = None
        self.stats = {"chunks": 0, "chunk_bytes": 0, "tail_bytes": 0}

    def views(self):
        return [self.flat[o:o + n].view_as(p) for o, n, p in zip(self.offsets, self.sizes, self.params)]

    # ---- arming: gradients born in the flat buffer -----------------------------------------------------------
    def arm(self, overlap_chunks: int = 0, group: Optional[dist.ProcessGroup] = None):
        """Directs the next rasterizer backward of these parameters into this buffer.  Only when every p.grad is
        None (an existing p.grad could BE a slice of this buffer from the previous step: writing the new gradient
        over it and then accumulating would be wrong), on a ROCm device, and with roles given; otherwise a no-op
        (pack() copies).

        overlap_chunks > 1 (and a process group with more than one rank): that backward runs its SH stage in this
        many Gaussian ranges and the slice of each finished range is all-reduced asynchronously while the next range
        computes.  Only valid when the armed backward is the LAST one contributing to these gradients before the
        reduction (one view per rank): a later local accumulation would be added after the sum over ranks."""
        self._works = []
        self._reduced_upto = 0
        if self.roles is None or not self.flat.is_cuda or any(p.grad is not None for p in self.params):
            return False
        from . import _C
        views = self.views()
        keys = [int(p.data_ptr()) for p in self._key_params]
        hook = None
        chunks = 1
        if overlap_chunks > 1 and _multi(group):
            self._group = group
            hook = self._on_sh_chunk
            chunks = int(overlap_chunks)
        _C.set_grad_arena([views[i] for i in self.roles], keys, chunks, hook)
        return True

    def disarm(self):
        if self.roles is not None and self.flat.is_cuda:
            from . import _C
            _C.set_grad_arena([])

    def _on_sh_chunk(self, c: int, g0: int, g1: int):
        """Called by the backward (csrc/torch_binding.cpp) right after the SH kernel for the Gaussians [g0, g1) has
        been enqueued: their rows of dL_dsh -- flat[g0 * 3M, g1 * 3M), the buffer starts with the SH gradients -- are
        final.  The collective is stream-ordered behind that kernel and runs on the communicator's own stream."""
        P = self.params[self._sh].shape[0]
        per = self.sizes[self._sh] // max(P, 1)
        lo, hi = g0 * per, g1 * per
        if lo != self._reduced_upto:       # ranges must arrive in order and without gaps; otherwise leave it to the tail
            return
        w = dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self._group, async_op=True)
        self._works.append(w)
        self._reduced_upto = hi
        self.stats["chunks"] += 1
        self.stats["chunk_bytes"] += (hi - lo) * 4

    # ---- pack / unpack ----------------------------------------------------------------------------------------
    def pack(self):
        """Brings every p.grad into the flat buffer: zeros where a parameter received none, nothing to do where
        p.grad already IS the slice (arm()), a copy otherwise."""
        for v, p in zip(self.views(), self.params):
            if p.grad is None:
                v.zero_()
            elif not (p.grad.data_ptr() == v.data_ptr() and p.grad.shape == v.shape and p.grad.is_contiguous()):
                v.copy_(p.grad)
        return self.flat

    def unpack(self):
        """Points every p.grad at its (reduced) slice of the flat buffer."""
        for v, p in zip(self.views(), self.params):
            p.grad = v
        return self.params

    @property
    def nbytes(self):
        return self.flat.numel() * 4


def allreduce_gaussian_grads(bucket: FlatGradBucket, group: Optional[dist.ProcessGroup] = None,
                             async_op: bool = False):
    """SUM-all-reduces the packed gradients across ranks and re-attaches them: ONE collective over the whole flat
    buffer, or -- when the backward already reduced a prefix chunk by chunk (FlatGradBucket.arm(overlap_chunks=K))
    -- one collective over the remaining tail, after which the chunk handles are waited for.  Every element of the
    buffer is reduced exactly once either way.  With world_size 1 (or no process group) there is nothing to
    exchange: gradients stay where autograd put them and nothing is copied."""
    if not _multi(group):
        bucket._works = []
        bucket._reduced_upto = 0
        return None
    done = bucket._reduced_upto
    works = bucket._works
    if done > 0:
        # the prefix was reduced in place while the backward ran; it is only valid if those gradients really live in
        # the buffer (they do when the arena was consumed, which is the only way the hook gets called)
        sh = bucket.params[bucket._sh]
        v = bucket.views()[bucket._sh]
        if sh.grad is None or sh.grad.data_ptr() != v.data_ptr():
            raise RuntimeError("chunked all-reduce ran but the SH gradient does not live in the flat buffer")
    # pack everything that is not yet reduced (copies only what is not already aliased)
    for i, (v, p) in enumerate(zip(bucket.views(), bucket.params)):
        if done > 0 and i == bucket._sh:
            continue
        if p.grad is None:
            v.zero_()
        elif not (p.grad.data_ptr() == v.data_ptr() and p.grad.shape == v.shape and p.grad.is_contiguous()):
            v.copy_(p.grad)
    tail = bucket.flat[done:]
    bucket.stats["tail_bytes"] += tail.numel() * 4
    work = dist.all_reduce(tail, op=dist.ReduceOp.SUM, group=group, async_op=async_op or bool(works))
    if works:
        for w in works:
            w.wait()
        if work is not None and not async_op:
            work.wait()
            work = None
    bucket._works = []
    bucket._reduced_upto = 0
    if done > 0:
        bucket.disarm()          # the chunk-reduced backward is complete now: its parameters may be differentiated again
    if work is None or not async_op:
        bucket.unpack()
    return work


# ---- the factored exchange: 12 B per Gaussian and view + 44 B per Gaussian instead of 236 B per Gaussian -----------------
GEOMETRY_ROLES = ("means3D", "opacities", "scales", "rotations")     # 3 + 1 + 3 + 4 = 11 floats per Gaussian


class _HipPacked:
    """The packed-message primitives of csrc/gsr_comm.hip (include/gsrast.h) on device tensors; tests substitute a torch
    restatement on CPU tensors (tests/packed_ref.py) through FactoredGradExchange(packed=...)."""

    @staticmethod
    def header_words(P):
        from . import _C
        return _C.msg_header_words(P)

    @staticmethod
    def visible_index(radii, hdr, scratch):
        from . import _C
        _C.visible_index(radii, hdr, scratch)

    @staticmethod
    def union_index(P, msgs, offsets, out_hdr, scratch):
        from . import _C
        _C.union_index(P, msgs, offsets, out_hdr, scratch)

    @staticmethod
    def pack_rows(hdr, src, dst, dst_stride, col0=0):
        from . import _C
        _C.pack_rows(hdr, src, dst, dst_stride, col0)

    @staticmethod
    def unpack_rows(hdr, src, src_stride, col0, dst):
        from . import _C
        _C.unpack_rows(hdr, src, src_stride, col0, dst)

    @staticmethod
    def pack_geometry(hdr, views, rows, flag):
        from . import _C
        _C.pack_geometry(hdr, views["means3D"], views["opacities"], views["scales"], views["rotations"], rows, flag)

    @staticmethod
    def unpack_geometry(hdr, rows, views):
        from . import _C
        _C.unpack_geometry(hdr, rows, views["means3D"], views["opacities"], views["scales"], views["rotations"])

    @staticmethod
    def sh_from_packed(means3D, campos, msgs, offsets, D, out):
        from . import _C
        _C.sh_grad_from_packed(means3D, campos, msgs, offsets, D, out)


class FactoredGradExchange:
    """The gradient exchange of a multi-GPU step in FACTORED form (new design).

    81 % of the dense payload is dL_dsh, and dL_dsh of ONE view is an outer product: basis(view direction) (x) dRGB, where
    dRGB is the Gaussian's clamp-masked colour gradient (3 floats) and the direction follows from the Gaussian's mean and
    the view's camera centre -- both known on every rank.  So instead of all-reducing (D+1)^2 x 12 B per Gaussian, every
    view's dRGB[P,3] is ALL-GATHERED (12 B per Gaussian and view) and each rank rebuilds
    sum_views basis (x) dRGB itself (`gsr_sh_grad_from_colors`, same arithmetic as the per-view SH backward, views added
    in a fixed order: bit-identical to accumulating the views on one device in that order); only the geometry block
    [means3D 3 | opacity 1 | scales 3 | rotations 4] = 44 B per Gaussian is all-reduced.  At 8 ranks x 1 view a rank
    moves 2 x 7/8 x 44 + 7 x 12 = 161 B per Gaussian instead of 2 x 7/8 x 236 = 413 B, independent of the SH degree.

        fx = FactoredGradExchange(params_by_role, views_per_rank=V)
        for v, cam in enumerate(my_views):
            fx.arm(v, sh_degree=D); out = rasterizer_of(cam)(...); loss(out).backward()   # gradients accumulate as usual, no dL_dsh
        fx.exchange(campos_of_all_views, sh_degree=D)      # [world * V, 3], global view order: rank-major
        # now p.grad of all five parameters holds the sum over all world * V views

    EARLY, PER-VIEW all-gather (round 4).  The colour slots are laid out view-major, colors[V, world, P, 3], so that local
    view v of all ranks is one contiguous all-gather.  The backward armed for view v calls back as soon as its geometry
    stage is enqueued -- that stage already leaves dRGB in the slot (GSR_BWD_PART_COLORS_EARLY) -- and the all-gather of
    view v starts THERE: before the SH-direction stage of that backward, and, at V > 1, while the later views of the step
    are still being rendered ((V - 1) / V of the colour traffic is hidden behind them).  The order in which the views are
    added is the memory order (local view index major, rank minor): `view_order()`.

    `sh_degree` is the ACTIVE degree of the step's renders (it changes during training: the usual schedule raises it every
    1000 iterations) -- given per step to arm() / exchange(); all views of a step must use the same one, and a mismatch
    raises instead of silently rebuilding gradients for coefficients the renders did not use.

    `compact="view"` (round 4): PER-VIEW compaction of the colour slots.  A view of a real capture sees a fraction of the
    scene (cameras inside a 360-degree scene: 14-19 % of the Gaussians per view, tools/comm_model.py), and dRGB of a Gaussian
    culled in a view is exactly zero.  After the forward of view v the caller hands over its `radii`
    (`fx.visible(v, radii)`): the visibility header (count, block bases, bit mask: P / 8 bytes) is built and the counts
    of all ranks gathered right there, off the critical path; when the backward calls back, the view's K visible rows are
    packed behind the header and ONE all-gather moves `header + 12 B x max-over-ranks K` per rank instead of 12 B x P; the
    SH gradient is rebuilt straight from the packed messages (gsr_sh_grad_from_packed: same arithmetic, same order, same
    bits).  `compact="view+geometry"` also cuts the geometry all-reduce down to the UNION of the step's views (the OR of the
    gathered masks -- every rank already holds them, no mask all-reduce; one host synchronisation for the row count).
    Precondition of "view+geometry": geometry gradients outside the union come from the rasterizer only (they are zero there);
    another loss term that touches culled Gaussians (a scale / opacity regulariser) is DETECTED per step and that step falls
    back to the dense 44 B/Gaussian all-reduce (`payload()["geometry_fallbacks"]`), so the sum is always complete.
    `compact=True` (round 3; kept): exchanges only the Gaussians that are visible (radii > 0 <=> a non-zero colour OR
    geometry gradient row) in at least one view of the step: one bit-mask all-reduce, then compacted buffers (it costs a
    host synchronisation for the row count, and the mask must be agreed first: no early all-gather in this form).
    Runs on CPU tensors with the "gloo" backend when `sh_from_colors` is given (tests; the product kernel is HIP only)."""

    def __init__(self, params_by_role: dict, views_per_rank: int = 1, sh_degree: Optional[int] = None, group=None, compact=False,
                 sh_from_colors=None, early: bool = True, packed=None, bands: int = 1, band_split: Optional[int] = None):
        if set(params_by_role) != set(ARENA_ROLES):
            raise ValueError(f"params_by_role must name exactly {ARENA_ROLES}")
        self.p = dict(params_by_role)
        self.group = group
        self.world = dist.get_world_size(group) if _multi(group) else 1
        self.rank = dist.get_rank(group) if _multi(group) else 0
        self.V = int(views_per_rank)
        self.D = None if sh_degree is None else int(sh_degree)      # default for steps that do not name their degree
        if compact not in (False, True, "view", "view+geometry"):
            raise ValueError('compact must be False, True, "view" or "view+geometry"')
        self.by_view = compact in ("view", "view+geometry")      # per-view packed colour messages
        self.union_geometry = compact == "view+geometry"
        # bands = 2 (round 6): every view's backward runs BANDED (include/gsrast.h GSR_BWD_PART_BAND_*): the image is cut at tile row
        # `band_split`, and the colour rows of the Gaussians that END above the cut (class 1: final after the first band) leave in a
        # message of their own while the second band is still being composited; the rest (class 2) follows as before.  Two packed
        # messages per view instead of one, the same rows, the same order of views per Gaussian: bit-identical gradients.
        if bands not in (1, 2):
            raise ValueError("bands must be 1 or 2")
        if bands == 2 and not self.by_view:
            raise ValueError('bands=2 needs the packed per-view messages: compact="view" or "view+geometry"')
        self.bands = int(bands)
        self.band_split = None if band_split is None else int(band_split)
        self.compact = compact is True                            # round-3 form: union mask agreed by an all-reduce
        self.early = bool(early) and not self.compact
        self._pk = packed if packed is not None else _HipPacked
        self._sh_from_colors = sh_from_colors
        means = self.p["means3D"]
        dev, self.P = means.device, means.shape[0]
        for t in self.p.values():
            if t.dtype != torch.float32 or t.device != dev:
                raise ValueError("all parameters must be float32 on one device")
        self.M = self.p["shs"].shape[1]
        self._geo_sizes = [self.p[r].numel() for r in GEOMETRY_ROLES]
        self.geo = torch.zeros(sum(self._geo_sizes), dtype=torch.float32, device=dev)
        if self.by_view:
            self.Hw = int(self._pk.header_words(self.P))
            self.Lmax = self.Hw + 3 * self.P                           # words of a message whose view sees everything
            self.colors = torch.zeros((self.V, self.P, 3), dtype=torch.float32, device=dev)        # own views only, dense
            S = self.V * self.bands                                    # message slots: slot = local view * bands + part
            self.hdr = torch.zeros((S, self.Hw), dtype=torch.int32, device=dev)
            self.msgs = torch.zeros((S, self.world * self.Lmax), dtype=torch.int32, device=dev)
            self.counts = torch.zeros((S, self.world), dtype=torch.int32, device=dev)
            self._scratch = torch.zeros((self.P + 255) // 256, dtype=torch.int32, device=dev)
            self.hdr_union = torch.zeros(self.Hw, dtype=torch.int32, device=dev)
            # the visible counts reach the HOST without touching the compute stream: header kernels, count all-gather and a
            # non-blocking copy into pinned memory run on a side stream; the backward's callback waits for that stream's event
            # (long signalled by then), never for the compute stream it is being enqueued on
            self._cuda = dev.type == "cuda"
            self._side = torch.cuda.Stream(device=dev) if self._cuda else None
            self._counts_host = torch.zeros((S, self.world), dtype=torch.int32, pin_memory=self._cuda)
            self._off_host = torch.zeros(S * self.world, dtype=torch.int64, pin_memory=self._cuda)
            self._off_dev = torch.zeros(S * self.world, dtype=torch.int64, device=dev)
            self._count_events = {}
        else:
            self.colors = torch.zeros((self.V, self.world, self.P, 3), dtype=torch.float32, device=dev)     # view-major
        self.sh_grad = torch.empty((self.P, self.M, 3), dtype=torch.float32, device=dev)
        self._geo_rows = None     # compact="view+geometry": persistent [P, 11] buffer of packed geometry rows (allocated on first use)
        self._kf = None
        self._kf_host = None
        self.stats = {"steps": 0, "rows_exchanged": 0, "early_allgathers": 0, "color_rows_sent": 0, "geometry_rows": 0}
        self._count_works = {}    # local view -> work handle of the all-gather of its visible count
        self._seen = set()        # local views whose radii arrived (visible())
        self._L = {}              # local view -> message length (words) agreed for this step
        self._works = {}          # local view -> work handle of its early all-gather
        self._step_degrees = []   # degrees the step's armed backwards were rendered with
        self._order = torch.tensor(self.view_order(), dtype=torch.long, device=dev)
        # one camera centre per MESSAGE, in the order the rebuilt SH gradient adds them: local view major, rank, then part
        self._msg_order = self._order.repeat_interleave(self.bands)

    def geo_views(self):
        out, o = {}, 0
        for r, n in zip(GEOMETRY_ROLES, self._geo_sizes):
            out[r] = self.geo[o:o + n].view_as(self.p[r])
            o += n
        return out

    def view_order(self):
        """Global view ids (rank * V + local view) in the order the rebuilt SH gradient adds them = memory order of `colors`."""
        return [r * self.V + v for v in range(self.V) for r in range(self.world)]

    # wire sizes of one step (per rank): what goes out and what comes in
    @property
    def geometry_bytes(self):
        return self.geo.numel() * 4

    @property
    def color_bytes_per_rank(self):
        return self.V * self.P * 12

    def payload(self):
        n = max(1, self.stats["steps"])
        if self.by_view:
            crow = self.stats["color_rows_sent"] / n               # sum over this rank's views of the agreed row capacity
            grow = self.stats["geometry_rows"] / n if self.union_geometry else self.P
            hdrb = self.Hw * 4 * self.V * self.bands
            return {"bands": self.bands, "payload_bytes_per_rank": int(crow * 12 + hdrb + grow * 44), "allreduce_bytes": int(grow * 44),
                    "allgather_bytes_sent": int(crow * 12 + hdrb), "allgather_bytes_received": int((crow * 12 + hdrb) * (self.world - 1)),
                    "dense_payload_bytes_per_rank": self.P * (self.M * 3 + 11) * 4, "color_rows_per_view": crow / self.V,
                    "geometry_rows": grow, "rows_total": self.P, "compacted": "view+geometry" if self.union_geometry else "view",
                    "geometry_fallbacks": self.stats.get("geometry_fallbacks", 0),
                    "first_band_rows_per_view": self.stats.get("color_rows_sent_first_band", 0) / n / self.V if self.bands == 2 else None,
                    "early_allgathers_per_step": self.stats["early_allgathers"] / n}
        rows = self.stats["rows_exchanged"] / n if self.compact else self.P
        sent = rows * (44 + 12 * self.V)
        return {"payload_bytes_per_rank": int(sent), "allreduce_bytes": int(rows * 44), "allgather_bytes_sent": int(rows * 12 * self.V),
                "allgather_bytes_received": int(rows * 12 * self.V * (self.world - 1)),
                "dense_payload_bytes_per_rank": self.P * (self.M * 3 + 11) * 4, "rows_per_step": rows, "rows_total": self.P,
                "compacted": self.compact, "early_allgathers_per_step": self.stats["early_allgathers"] / n}

    def arm(self, v: int, sh_degree: Optional[int] = None, band_split: Optional[int] = None):
        """Before the forward + backward of this rank's local view v: its colour gradients go to slot [v, rank] of the
        all-gather buffer.  The first view's geometry gradients are born in the all-reduce buffer when no p.grad exists
        yet; later views accumulate into them through autograd as usual.  sh_degree: the settings' sh_degree of this view's
        render (checked against the step's other views and against exchange()).  band_split (bands = 2): the tile row this
        view's backward is cut at (default: the constructor's; typically half of ceil(height / 16))."""
        from . import _C
        if sh_degree is not None:
            self._step_degrees.append(int(sh_degree))
        slot = self.colors[v] if self.by_view else self.colors[v, self.rank]
        fresh = all(self.p[r].grad is None for r in ARENA_ROLES)
        outs = []
        if v == 0 and fresh and self.geo.is_cuda:
            g = self.geo_views()
            outs = [g["means3D"], self.sh_grad, g["opacities"], g["scales"], g["rotations"]]     # slot 1 (dL_dsh) is ignored
        keys = [int(self.p[r].data_ptr()) for r in _KEY_ROLES]
        hook = None
        if self.early and (_multi(self.group) or self.by_view):
            hook = lambda v=v: self._on_colors_ready(v)
        if self.bands == 2:
            S = self.band_split if band_split is None else int(band_split)
            if S is None or S <= 0:
                raise ValueError("FactoredGradExchange(bands=2): give the split tile row (band_split) to the constructor or to arm()")
            _C.set_grad_arena(outs, keys, 1, hook, colors_out=slot, band_split=S, band_hook=lambda v=v: self._on_band_ready(v),
                              class_hook=lambda first, second, v=v: self._on_classes(v, first, second))
            return
        _C.set_grad_arena(outs, keys, 1, hook, colors_out=slot)

    def visible(self, v: int, radii: torch.Tensor):
        """compact="view": right after the FORWARD of local view v, with the `radii` it returned -- builds the view's
        visibility header and gathers the visible counts of all ranks (asynchronously: nothing here waits).  With bands = 2 the
        headers come from the two Gaussian classes the banded backward reports when it starts (_on_classes): a no-op here."""
        if not self.by_view or self.bands == 2:
            return
        self._index_slots(v, [radii.detach().contiguous()])

    def _on_classes(self, v: int, first: torch.Tensor, second: torch.Tensor):
        """bands = 2: called by the FORWARD of local view v (csrc/torch_binding.cpp, on the caller's thread, right behind the forward's
        kernels) with the two classes of the cut its armed banded backward will use (int32[P], 1 = member; include/gsrast.h
        gsr_band_classes): the headers of the view's two messages and the gathers of their row counts leave on the side stream, as
        visible() does for the unbanded view.  (A first version computed the classes at the start of the BACKWARD, on the autograd
        thread: the host then waited on an event recorded microseconds earlier on the same thread, and the step stalled for 9-90 ms at
        a time on the MI355X -- tools/band_exchange_timing.py.)"""
        self._index_slots(2 * v, [first, second])

    def _index_slots(self, slot0: int, members):
        """Headers of the consecutive message slots slot0, slot0 + 1, ... from members[i][P] (> 0 = the Gaussian has a row in that
        message) + the all-gathers of their row counts -- ONE excursion to the side stream, ONE copy of the counts to pinned host
        memory and ONE event for all of them (two excursions per view, each with its own wait on the compute stream, its own 4-byte
        copy and its own event, stalled the compute stream for ~9 ms per step on the MI355X: tools/band_exchange_timing.py)."""
        n = len(members)

        def build():
            for i, member in enumerate(members):
                slot = slot0 + i
                self._pk.visible_index(member, self.hdr[slot], self._scratch)
                if _multi(self.group):
                    w = dist.all_gather_into_tensor(self.counts[slot], self.hdr[slot][0:1], group=self.group, async_op=True)
                    if self._cuda:
                        w.wait()                           # the side stream waits for the communicator's stream; the host does not
                    else:
                        self._count_works[slot] = w
                else:
                    self.counts[slot, 0:1].copy_(self.hdr[slot][0:1])
        if self._cuda:
            cur = torch.cuda.current_stream(self.geo.device)
            with torch.cuda.stream(self._side):
                self._side.wait_stream(cur)                # radii / the classes come from kernels just enqueued
                build()
                self._counts_host[slot0:slot0 + n].copy_(self.counts[slot0:slot0 + n], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self._side)
            for member in members:
                member.record_stream(self._side)
            for i in range(n):
                self._count_events[slot0 + i] = ev
        else:
            build()
        for i in range(n):
            self._seen.add(slot0 + i)

    def _send_view(self, v: int):
        """(bands = 1) the one message of local view v."""
        self._send(v)

    def _send(self, slot: int):
        """compact="view": packs the rows of message slot `slot` (= local view * bands + part) behind its header and starts the
        all-gather of the messages.  The message length is the same on every rank: header + 3 floats x the LARGEST row count of
        the slot over the ranks (known from the counts gathered earlier; reading it is the only host synchronisation, and at that
        point the device still has the rest of the backward queued)."""
        v = slot // self.bands
        if slot not in self._seen:
            if self.bands == 2:
                raise RuntimeError(f"FactoredGradExchange(bands=2): the classes of local view {v} have not arrived (its backward must run banded: arm())")
            raise RuntimeError(f'FactoredGradExchange(compact="view"): call visible({v}, radii) after the forward of local view {v}')
        if self._cuda:
            ev = self._count_events.pop(slot)
            ev.synchronize()                                                 # the side stream only: signalled long ago
            torch.cuda.current_stream(self.geo.device).wait_event(ev)        # the header is read by the kernels enqueued below
            kcap = max(self._counts_host[slot].tolist())          # (plain Python: see _exchange_by_view on host-side torch reductions)
        else:
            w = self._count_works.pop(slot, None)
            if w is not None:
                w.wait()
            kcap = int(self.counts[slot].max().item())
            self._counts_host[slot].copy_(self.counts[slot])
        L = self.Hw + 3 * kcap
        buf = self.msgs[slot][: self.world * L].view(self.world, L)
        mine = buf[self.rank]
        mine[: self.Hw].copy_(self.hdr[slot])
        if kcap > 0:
            self._pk.pack_rows(self.hdr[slot], self.colors[v], mine[self.Hw:].view(torch.float32), 3, 0)
        self._L[slot] = L
        self.stats["color_rows_sent"] += kcap
        if self.bands == 2 and slot % 2 == 0:
            self.stats["color_rows_sent_first_band"] = self.stats.get("color_rows_sent_first_band", 0) + kcap
        if _multi(self.group):
            self._works[slot] = _all_gather_in_place(buf, self.rank, 1, self.group)
        else:
            self._works[slot] = None

    def _on_band_ready(self, v: int):
        """bands = 2: called by the banded backward of local view v right after its FIRST band has been enqueued: the rows of
        slot [v] of the Gaussians that end above the cut hold their final dRGB -- their message leaves now, stream-ordered behind
        that band, while the second band is composited."""
        slot = 2 * v
        if slot in self._works:
            return
        self._send(slot)
        self.stats["early_band_allgathers"] = self.stats.get("early_band_allgathers", 0) + 1

    def _on_colors_ready(self, v: int):
        """Called by the backward armed for local view v (csrc/torch_binding.cpp) right after its geometry stage has been
        enqueued: slot [v, rank] holds the view's final dRGB.  The all-gather of view v over all ranks is stream-ordered
        behind that kernel and runs on the communicator's stream while the backward's SH-direction stage -- and the
        step's remaining views -- compute."""
        slot = v * self.bands + (self.bands - 1) if self.by_view else v      # bands = 2: the second message (class 2)
        if slot in self._works:
            return
        if self.by_view:
            self._send(slot)
        else:
            self._works[v] = _all_gather_in_place(self.colors[v], self.rank, 1, self.group)
        self.stats["early_allgathers"] += 1

    def _step_degree(self, sh_degree):
        degs = set(self._step_degrees)
        self._step_degrees = []
        if len(degs) > 1:
            raise ValueError(f"FactoredGradExchange: the views of one step were rendered with different SH degrees {sorted(degs)}")
        if sh_degree is not None and degs and int(sh_degree) not in degs:
            raise ValueError(f"FactoredGradExchange.exchange(sh_degree={sh_degree}) but the step's backwards were armed with degree {sorted(degs)[0]}")
        D = int(sh_degree) if sh_degree is not None else (degs.pop() if degs else self.D)
        if D is None:
            raise ValueError("FactoredGradExchange: the active SH degree of the step is unknown -- pass sh_degree to arm() / exchange() "
                             "(or a default to the constructor)")
        if (D + 1) ** 2 > self.M:
            raise ValueError(f"SH degree {D} does not fit the {self.M} stored coefficients")
        return D

    def exchange(self, campos_all: torch.Tensor, sh_degree: Optional[int] = None):
        """campos_all [world * V, 3]: the camera centres of ALL views of this step in global order (rank-major: view v of
        rank r is row r * V + v) -- every rank knows the step's camera list.  Afterwards p.grad of all five parameters is
        the sum over all views.  sh_degree: the active degree the step's views were rendered with (see the class docstring)."""
        P, V, W = self.P, self.V, self.world
        D = self._step_degree(sh_degree)
        views = self.geo_views()
        for r in GEOMETRY_ROLES:                              # pack what autograd did not put there itself
            p, v = self.p[r], views[r]
            if p.grad is None:
                v.zero_()
            elif not (p.grad.data_ptr() == v.data_ptr() and p.grad.shape == v.shape and p.grad.is_contiguous()):
                v.copy_(p.grad)
        multi = _multi(self.group)
        rows = None
        w2 = None
        if self.by_view:
            return self._exchange_by_view(campos_all, D, views, multi)
        if multi and self.compact:
            mine = self.colors[:, self.rank]                  # [V, P, 3]
            # rows with a non-zero gradient in ANY view of ANY rank (a culled Gaussian's rows are exactly zero): every one of
            # the 11 geometry floats and the colour slots counts, so that no partial row is left out of the sum
            live = mine.abs().amax(dim=(0, 2)) > 0
            for r in GEOMETRY_ROLES:
                live |= views[r].reshape(P, -1).abs().amax(dim=1) > 0
            mask = live.to(torch.int32)
            dist.all_reduce(mask, op=dist.ReduceOp.MAX, group=self.group)
            rows = torch.nonzero(mask, as_tuple=False).flatten()
            K = rows.numel()                                  # host synchronisation: the buffers below are sized by it
            geo_rows = torch.cat([views[r].reshape(P, -1)[rows] for r in GEOMETRY_ROLES], dim=1).contiguous()       # [K,11]
            col = torch.zeros((V, W, K, 3), dtype=torch.float32, device=self.geo.device)
            col[:, self.rank] = mine[:, rows]
            w1s = [_all_gather_in_place(col[v], self.rank, 1, self.group) for v in range(V)]
            w2 = dist.all_reduce(geo_rows, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            for w in w1s:
                w.wait()
            self.colors.zero_()
            self.colors[:, :, rows] = col
            w2.wait()
            w2 = None
            o = 0
            for r in GEOMETRY_ROLES:
                n = views[r].reshape(P, -1).shape[1]
                views[r].reshape(P, -1)[rows] = geo_rows[:, o:o + n]
                o += n
            self.stats["rows_exchanged"] += K
        elif multi:
            # views whose all-gather did not start from inside their backward (early=False, a CPU run, an arena that autograd
            # did not consume) start now; then the geometry all-reduce; the colour works are waited for in view order
            for v in range(V):
                if v not in self._works:
                    self._works[v] = _all_gather_in_place(self.colors[v], self.rank, 1, self.group)
            w2 = dist.all_reduce(self.geo, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            for v in range(V):
                self._works[v].wait()
        self._works = {}
        # the SH gradient of the whole step from every view's colour gradient (runs while the geometry all-reduce is in flight)
        fn = self._sh_from_colors
        if fn is None:
            from . import _C
            fn = _C.sh_grad_from_colors
        campos = campos_all.detach().to(self.geo.device, torch.float32)[self._order].contiguous()      # (no host round trip)
        fn(self.p["means3D"].detach(), campos, self.colors.view(V * W, P, 3), D, self.sh_grad)
        if w2 is not None:
            w2.wait()
        for r in GEOMETRY_ROLES:
            self.p[r].grad = views[r]
        self.p["shs"].grad = self.sh_grad
        self.stats["steps"] += 1


def _exchange_by_view_impl(self, campos_all, D, views, multi):
    """compact="view" / "view+geometry" (FactoredGradExchange._exchange_by_view)."""
    P, V, W, B, dev = self.P, self.V, self.world, self.bands, self.geo.device
    S = V * B
    for slot in range(S):                            # messages that did not leave from inside their backward
        if slot not in self._works:
            self._send(slot)
    w2 = None
    if multi and not self.union_geometry:
        w2 = dist.all_reduce(self.geo, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
    for slot in range(S):
        if self._works[slot] is not None:
            self._works[slot].wait()
    # word offset of every message, in the order the SH gradient adds them: local view major, then rank, then part (a Gaussian has a
    # row in exactly one part of a view, so per Gaussian this is the unbanded order of views: bit-identical sums) -- from the counts
    # the host already holds, handed to the device by a non-blocking copy out of pinned memory.  Message (v, r, part) lies at word
    # slot * (W * Lmax) + r * L_slot of `msgs`, slot = v * B + part.
    # (plain Python integers: a torch reduction over this [S, W] host tensor enters an OpenMP region -- on the 128-core host of an
    # MI355X box `counts.max(dim=1)` of a 2 x 1 tensor took 26 ms per step, tools/band_exchange_timing.py)
    cnt = self._counts_host.tolist()                                                                        # [S][W], host
    Ls = [self.Hw + 3 * max(row) for row in cnt]
    self._off_host.copy_(torch.tensor([(v * B + part) * (W * self.Lmax) + r * Ls[v * B + part]
                                       for v in range(V) for r in range(W) for part in range(B)], dtype=torch.int64))
    self._off_dev.copy_(self._off_host, non_blocking=True)
    offsets = self._off_dev
    msgs = self.msgs.view(-1)
    campos = campos_all.detach().to(dev, torch.float32)[self._msg_order].contiguous()
    kf_ev = None
    if self.union_geometry:
        # rows with a non-zero RASTERIZER geometry gradient anywhere in the step = the OR of the gathered masks (identical on
        # every rank).  PRECONDITION of summing only those rows: no other loss term put a gradient on a row outside the union
        # (a scale / opacity regulariser does: ADVICE r4) -- checked: rows of this rank's geometry block that are non-zero outside the
        # union raise a flag, the ranks agree on it (4-byte MAX all-reduce), and a flagged step sums the whole dense block instead
        # (`geometry_fallbacks` in payload()).
        # ONE pass packs the four tensors' rows of the union into the persistent [P, 11] buffer (sized for the worst case: the host
        # does not know K yet) and raises the flag; K and the agreed flag travel to pinned host memory behind an EVENT, and all of
        # this is enqueued IN FRONT of the SH-gradient rebuild below: the host then waits for that event only -- while the device
        # still rebuilds -- instead of draining the stream (round 5: a full synchronisation here cost ~0.6 ms of idle device per step)
        self._pk.union_index(P, msgs, offsets, self.hdr_union, self._scratch)
        if self._geo_rows is None:
            self._geo_rows = torch.empty((P, 11), dtype=torch.float32, device=dev)
            self._kf = torch.zeros(2, dtype=torch.int32, device=dev)
            self._kf_host = torch.zeros(2, dtype=torch.int32, pin_memory=self._cuda)
        self._kf.zero_()
        self._pk.pack_geometry(self.hdr_union, views, self._geo_rows, self._kf[1:2])
        self._kf[0:1].copy_(self.hdr_union[0:1])
        if multi:
            dist.all_reduce(self._kf[1:2], op=dist.ReduceOp.MAX, group=self.group)
        self._kf_host.copy_(self._kf, non_blocking=True)
        if self._cuda:
            kf_ev = torch.cuda.Event()
            kf_ev.record()
    self._pk.sh_from_packed(self.p["means3D"].detach(), campos, msgs, offsets, D, self.sh_grad)
    if self.union_geometry:
        if kf_ev is not None:
            kf_ev.synchronize()                               # the 8-byte copy only; the rebuild kernel is still running
        K, outside = (int(x) for x in self._kf_host.tolist())
        if outside:
            self.stats["geometry_fallbacks"] = self.stats.get("geometry_fallbacks", 0) + 1
            self.stats["geometry_rows"] += P
            if multi:
                dist.all_reduce(self.geo, op=dist.ReduceOp.SUM, group=self.group)
        else:
            self.stats["geometry_rows"] += K
            if K > 0:
                rows = self._geo_rows[:K]
                if multi:
                    dist.all_reduce(rows, op=dist.ReduceOp.SUM, group=self.group)
                self._pk.unpack_geometry(self.hdr_union, rows, views)
    if w2 is not None:
        w2.wait()
    for r in GEOMETRY_ROLES:
        self.p[r].grad = views[r]
    self.p["shs"].grad = self.sh_grad
    self._works, self._L = {}, {}
    self._seen = set()
    self.stats["steps"] += 1


FactoredGradExchange._exchange_by_view = _exchange_by_view_impl


def _all_gather_in_place(buf: torch.Tensor, rank: int, V: int, group):
    """All-gathers buf[world * V, ...] whose rows [rank * V, (rank + 1) * V) hold this rank's contribution; returns a work
    handle.  RCCL gathers in place (the input is the rank's slice of the output)."""
    flat = buf.view(-1)
    n = flat.numel() // (dist.get_world_size(group))
    try:
        return dist.all_gather_into_tensor(flat, flat[rank * n:(rank + 1) * n], group=group, async_op=True)
    except (RuntimeError, NotImplementedError):              # a backend without the tensor form
        chunks = [flat[i * n:(i + 1) * n] for i in range(dist.get_world_size(group))]
        return dist.all_gather(chunks, flat[rank * n:(rank + 1) * n].clone(), group=group, async_op=True)


def shard_views(views: Sequence, rank: Optional[int] = None, world_size: Optional[int] = None):
    """The cameras rank `rank` renders this iteration: views[rank::world_size] (one camera per GPU when
    len(views) == world_size, the C4 configuration)."""
    if rank is None:
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    return list(views[rank::world_size])


def render_views_and_reduce(render_fn, views: Iterable, bucket: FlatGradBucket,
                            group: Optional[dist.ProcessGroup] = None, overlap_chunks: int = 0):
    """One data-parallel iteration: `render_fn(view)` must run forward+backward for one camera and
    accumulate into p.grad; afterwards gradients are summed over ranks.  Returns what render_fn
    returned for each local view.  overlap_chunks > 1 overlaps the reduction with the tail of the backward
    (FlatGradBucket.arm) when this rank renders exactly one view."""
    views = list(views)
    for p in bucket.params:
        p.grad = None
    multi = _multi(group)
    if multi:
        # the first view's gradients are born in the flat buffer; with a single local view its backward is also the
        # last one, so its SH chunks may be reduced while it is still running.  Every rank must then issue the same chunk
        # collectives: the ranks agree (one 4-byte MIN all-reduce) that all of them could arm, else nobody chunks.
        chunks = overlap_chunks if len(views) == 1 else 0
        armed = bucket.arm(chunks, group)
        if chunks > 1:
            ok = torch.tensor([1 if armed else 0], dtype=torch.int32, device=bucket.flat.device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
            if int(ok.item()) == 0:
                bucket.disarm()
                bucket.arm(0, group)
    try:
        outs = [render_fn(v) for v in views]
    finally:
        if multi:
            bucket.disarm()     # nothing consumed it (no backward ran): do not leak into a later step
    allreduce_gaussian_grads(bucket, group)
    return outs


# ---- tile-grid sharding of ONE view ("additionally possible" for a single 4K view) ------------------
def tile_row_band(height: int, rank: Optional[int] = None, world_size: Optional[int] = None):
    """The 16-pixel tile rows [lo, hi) of an image of `height` pixels that rank `rank` renders when one view is
    split across `world_size` processes (contiguous bands of equal height, the last one possibly shorter)."""
    if rank is None:
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rows = (height + 15) // 16
    per = (rows + world_size - 1) // world_size
    return min(rows, rank * per), min(rows, (rank + 1) * per)


class tile_band:
    """Context manager: inside it the rasterizer calls of THIS THREAD bin, composite and differentiate only the tile rows
    [lo, hi) of every view (per-call option `tile_row_lo` / `tile_row_hi`, gaustudio_amd/options.py -- nothing
    process-wide is changed, so two bands can be rendered concurrently from two threads of one process, and the backward
    of a call keeps the band of its forward).  Preprocessing stays replicated (every rank projects every Gaussian: 0.06 ms
    per million), there is no exchange in the forward; pixels outside the band come back as an empty scene's, and the
    per-Gaussian gradients are the band's partial sums -- the same flat all-reduce as for view sharding
    (allreduce_gaussian_grads) completes them.  A loss must only use the band's rows (band_rows())."""

    def __init__(self, lo: int, hi: int):
        self.lo, self.hi = int(lo), int(hi)
        from .options import options
        self._ctx = options(tile_band=(self.lo, self.hi))

    def band_rows(self, height: int):
        return slice(min(height, 16 * self.lo), min(height, 16 * self.hi))

    def __enter__(self):
        self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        return self._ctx.__exit__(*exc)
