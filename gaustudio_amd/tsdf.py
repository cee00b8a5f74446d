"""TSDF fusion and mesh extraction on the GPU.

Mirrors the part of `vdbfusion.VDBVolume` gs-extract-mesh uses (gaustudio/scripts/extract_mesh.py:86,115,145):

    vol = TSDFVolume(voxel_size=0.01, sdf_trunc=0.04, space_carving=False)
    vol.integrate(points_world, origin)                       # per rendered view, points stay on the GPU
    vertices, faces = vol.extract_triangle_mesh(min_weight=5)

The volume is a block-sparse grid in HBM owned by this object as torch tensors (hash keys + 4 KiB of voxels per
hash slot); the kernels are stateless (include/gsrast.h, csrc/gsr_tsdf.hip).  ROCm devices only, no CPU fallback.
"""
import ctypes

import numpy as np
import torch

from . import _C

_EMPTY = -1          # all bits set, as int64


class TSDFVolume:
    def __init__(self, voxel_size, sdf_trunc, space_carving=False, device="cuda", capacity_blocks=1 << 18):
        """capacity_blocks: hash slots (power of two).  Every slot reserves 512 voxels x 8 B = 4 KiB of HBM, so the
        default 2^18 slots = 1 GiB holds scenes of ~130 k occupied 8^3 blocks at a load factor of 0.5."""
        self.voxel_size = float(voxel_size)
        self.sdf_trunc = float(sdf_trunc)
        self.space_carving = bool(space_carving)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError(f"TSDFVolume lives on a ROCm device, got '{self.device}' (no CPU fallback)")
        if capacity_blocks & (capacity_blocks - 1) or capacity_blocks <= 0:
            raise ValueError("capacity_blocks must be a power of two")
        self.capacity = int(capacity_blocks)
        self.keys = torch.full((self.capacity,), _EMPTY, dtype=torch.int64, device=self.device)
        self.voxels = torch.zeros((self.capacity, 512), dtype=torch.int64, device=self.device)
        self.status = torch.zeros(1, dtype=torch.int32, device=self.device)

    # ------------------------------------------------------------------ integrate
    def integrate(self, points, extrinsic=None, origin=None):
        """VDBVolume.integrate(points, extrinsic): `extrinsic` is the sensor origin [3] (what the reference passes,
        extract_mesh.py:115) or a 4x4 camera-to-world pose whose translation is the origin.
        `points`: [N,3], or an image-shaped point map [H,W,3] (what depth_to_points returns): the kernel then works in
        32 x 32 pixel patches, whose rays share voxels in both image directions -- same volume to the bit, fewer atomics."""
        o = origin if origin is not None else extrinsic
        if o is None:
            raise ValueError("integrate needs the sensor origin")
        o = np.asarray(o.detach().cpu().numpy() if torch.is_tensor(o) else o, dtype=np.float64)
        if o.shape == (4, 4):
            o = o[:3, 3]
        if o.shape != (3,):
            raise ValueError("origin must have shape [3] or [4,4]")
        pts = torch.as_tensor(points)
        row_w = 0
        if pts.dim() == 3 and pts.shape[2] == 3:
            row_w = int(pts.shape[1])
            pts = pts.reshape(-1, 3)
        if pts.dim() != 2 or pts.shape[1] != 3:
            raise ValueError("points must have shape [N,3] or [H,W,3]")
        pts = pts.to(device=self.device, dtype=torch.float32).contiguous()
        if pts.shape[0] == 0:
            return
        origin_c = (ctypes.c_float * 3)(*[float(v) for v in o])
        L = _C.lib()
        with torch.cuda.device(self.device):
            rc = L.gsr_tsdf_integrate_map(_C._ptr(pts), ctypes.c_int(pts.shape[0]), ctypes.c_int(row_w), origin_c,
                                          ctypes.c_float(self.voxel_size), ctypes.c_float(self.sdf_trunc),
                                          ctypes.c_int(int(self.space_carving)), _C._ptr(self.keys),
                                          ctypes.c_uint64(self.capacity), _C._ptr(self.voxels), _C._ptr(self.status),
                                          _C._stream(self.device))
        if rc < 0:
            raise RuntimeError(f"gsr_tsdf_integrate_map failed (rc={rc})")

    def _check_overflow(self):
        st = int(self.status.item())
        if st & 1:
            raise RuntimeError(f"TSDFVolume: the block hash ({self.capacity} slots) overflowed; "
                               "create the volume with a larger capacity_blocks")
        if st & 2:
            raise RuntimeError("TSDFVolume: a voxel received more than 2^24 - 2^20 observations (the count field of the packed "
                               "voxel word is full); later observations of it were dropped")

    # ------------------------------------------------------------------ inspection
    def occupied_blocks(self):
        """(slots [n] int32, block coordinates [n,3] int32), ordered by block key (deterministic)."""
        self._check_overflow()
        slots = torch.nonzero(self.keys != _EMPTY).flatten()
        k = self.keys[slots]
        k, order = torch.sort(k)
        slots = slots[order]
        B = 1 << 20
        coords = torch.stack([((k >> 42) & 0x1fffff) - B, ((k >> 21) & 0x1fffff) - B, (k & 0x1fffff) - B], dim=1)
        return slots.to(torch.int32).contiguous(), coords.to(torch.int32)

    def export_voxels(self):
        """All observed voxels as (coords [m,3] int32, tsdf [m] f32, weight [m] int32, sum_q [m] int64), sorted by
        (z, y, x).  Test / inspection helper."""
        slots, bcoords = self.occupied_blocks()
        n = slots.shape[0]
        counts = torch.empty((n, 512), dtype=torch.int32, device=self.device)
        tsdf = torch.empty((n, 512), dtype=torch.float32, device=self.device)
        sums = torch.empty((n, 512), dtype=torch.int64, device=self.device)
        if n:
            L = _C.lib()
            with torch.cuda.device(self.device):
                rc = L.gsr_tsdf_export_blocks(_C._ptr(self.voxels), _C._ptr(slots), ctypes.c_int(n), ctypes.c_float(self.sdf_trunc),
                                              _C._ptr(counts), _C._ptr(tsdf), _C._ptr(sums), _C._stream(self.device))
            if rc < 0:
                raise RuntimeError(f"gsr_tsdf_export_blocks failed (rc={rc})")
        local = torch.arange(512, device=self.device)
        lx, ly, lz = local & 7, (local >> 3) & 7, local >> 6
        coords = bcoords[:, None, :] * 8 + torch.stack([lx, ly, lz], dim=1)[None].to(torch.int32)
        m = counts > 0
        coords, tsdf, counts, sums = coords[m], tsdf[m], counts[m], sums[m]
        key = (coords[:, 2].long() << 42) + (coords[:, 1].long() << 21) + coords[:, 0].long()
        order = torch.argsort(key)
        return coords[order], tsdf[order], counts[order], sums[order]

    # ------------------------------------------------------------------ mesh
    def extract_triangle_mesh(self, fill_holes=True, min_weight=0.5):
        """VDBVolume.extract_triangle_mesh(fill_holes, min_weight) -> (vertices [nv,3] float64, triangles [nt,3] int32)
        as numpy arrays (what trimesh.Trimesh(vertices, faces) at extract_mesh.py:146 takes)."""
        v, t = self.extract_triangle_mesh_device(fill_holes, min_weight)
        return v.double().cpu().numpy(), t.cpu().numpy()

    def extract_triangle_mesh_device(self, fill_holes=True, min_weight=0.5):
        slots, _ = self.occupied_blocks()
        n = slots.shape[0]
        dev = self.device
        if n == 0:
            return torch.zeros((0, 3), dtype=torch.float32, device=dev), torch.zeros((0, 3), dtype=torch.int32, device=dev)
        slot_to_block = torch.zeros(self.capacity, dtype=torch.int32, device=dev)
        slot_to_block[slots.long()] = torch.arange(n, dtype=torch.int32, device=dev)
        cases = torch.empty((n, 512), dtype=torch.uint8, device=dev)
        flags = torch.empty((n, 512), dtype=torch.int32, device=dev)
        bnv = torch.empty(n, dtype=torch.int32, device=dev)
        bnt = torch.empty(n, dtype=torch.int32, device=dev)
        L = _C.lib()
        cap = ctypes.c_uint64(self.capacity)
        with torch.cuda.device(dev):
            st = _C._stream(dev)
            rc = L.gsr_tsdf_mc_classify(_C._ptr(self.keys), cap, _C._ptr(self.voxels), _C._ptr(slots), ctypes.c_int(n),
                                        _C._ptr(slot_to_block), ctypes.c_float(self.sdf_trunc), ctypes.c_float(float(min_weight)),
                                        ctypes.c_int(int(bool(fill_holes))), _C._ptr(cases), _C._ptr(flags), _C._ptr(bnv),
                                        _C._ptr(bnt), st)
            if rc < 0:
                raise RuntimeError(f"gsr_tsdf_mc_classify failed (rc={rc})")
            voff = torch.cumsum(bnv.long(), 0)
            toff = torch.cumsum(bnt.long(), 0)
            nv, nt = int(voff[-1].item()), int(toff[-1].item())
            if nv >= 2 ** 31 or nt >= 2 ** 31:
                raise RuntimeError("mesh too large for 32-bit indices")
            voff = (voff - bnv).to(torch.int32).contiguous()
            toff = (toff - bnt).to(torch.int32).contiguous()
            vertices = torch.empty((nv, 3), dtype=torch.float32, device=dev)
            triangles = torch.empty((nt, 3), dtype=torch.int32, device=dev)
            vbase = torch.empty((n, 512), dtype=torch.int32, device=dev)
            if nt:
                rc = L.gsr_tsdf_mc_emit(_C._ptr(self.keys), cap, _C._ptr(self.voxels), _C._ptr(slots), ctypes.c_int(n),
                                        _C._ptr(slot_to_block), ctypes.c_float(self.voxel_size), ctypes.c_float(self.sdf_trunc),
                                        _C._ptr(cases), _C._ptr(flags), _C._ptr(voff), _C._ptr(toff), _C._ptr(vbase),
                                        _C._ptr(vertices), _C._ptr(triangles), st)
                if rc < 0:
                    raise RuntimeError(f"gsr_tsdf_mc_emit failed (rc={rc})")
        return vertices, triangles
