"""On-disk formats either side of the rasterizer: the Gaussian PLY and cameras.json.

Host-side only (numpy + torch tensors, no kernels): these feed `GaussianRasterizer` /
`FusedGaussianRasterizer` with exactly the tensors the reference's loaders would have produced.

Reference behaviour followed here
  * PLY load   /root/reference/gaustudio/models/base.py:73-105  (per attribute: every vertex property whose
               name starts with the attribute, ordered by its trailing integer; xyz / opacity special-cased)
  * PLY export /root/reference/gaustudio/models/vanilla_sg.py:144-181 (x,y,z,nx,ny,nz,f_dc_*,f_rest_*,opacity,
               scale_*,rot_*; all float32; f_dc / f_rest written channel-major via reshape(P,-1,3).transpose(1,2))
  * features   /root/reference/gaustudio/models/vanilla_sg.py:102-106 get_features reshapes f_rest to [P,15,3]
               WITHOUT undoing the channel-major order export applies.  This asymmetry is a quirk of the
               reference; `GaussianCloud.features()` reproduces it by default and offers
               `channel_major=True` for PLYs written by the original 3DGS trainer.
  * cameras    /root/reference/gaustudio/utils/cameras_utils.py:8-38 (JSON_to_camera),
               /root/reference/gaustudio/datasets/utils.py:58-80 (camera_to_JSON),
               /root/reference/gaustudio/datasets/__init__.py:52-104,148-183 (Camera matrices)

The reference reads and writes PLY through the third-party `plyfile` package (not vendored, not installed
here).  The reader/writer below restate the PLY 1.0 container format itself (header grammar + packed
little/big-endian or ascii scalar records); parity for the container is pinned by byte-level known-answer
tests (tests/test_formats.py), parity for the camera math by a fixture generated from the reference's own
functions (tests/golden/py_cameras.npz).
"""
import json
import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import scenes

_PLY_SCALARS = {
    "char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
    "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
    "double": "f8", "float64": "f8",
}
_NP_TO_PLY = {"i1": "char", "u1": "uchar", "i2": "short", "u2": "ushort", "i4": "int", "u4": "uint", "f4": "float",
              "f8": "double"}


class PlyFormatError(ValueError):
    pass


def _read_header(f):
    """Returns (format, [(element_name, count, [(prop_name, numpy_code | None)])]) and leaves `f` at the data."""
    magic = f.readline()
    if magic.strip() != b"ply":
        raise PlyFormatError("not a PLY file (missing 'ply' magic)")
    fmt = None
    elements = []
    while True:
        line = f.readline()
        if not line:
            raise PlyFormatError("unterminated PLY header")
        tok = line.decode("ascii", errors="replace").split()
        if not tok or tok[0] in ("comment", "obj_info"):
            continue
        if tok[0] == "end_header":
            break
        if tok[0] == "format":
            if len(tok) != 3 or tok[1] not in ("ascii", "binary_little_endian", "binary_big_endian"):
                raise PlyFormatError(f"unsupported PLY format line: {line!r}")
            fmt = tok[1]
        elif tok[0] == "element":
            elements.append((tok[1], int(tok[2]), []))
        elif tok[0] == "property":
            if not elements:
                raise PlyFormatError("property before any element")
            if tok[1] == "list":
                elements[-1][2].append((tok[4], None))
            else:
                if tok[1] not in _PLY_SCALARS:
                    raise PlyFormatError(f"unknown PLY scalar type {tok[1]!r}")
                elements[-1][2].append((tok[2], _PLY_SCALARS[tok[1]]))
        else:
            raise PlyFormatError(f"unexpected header line: {line!r}")
    if fmt is None:
        raise PlyFormatError("PLY header has no format line")
    return fmt, elements


def read_ply_vertices(path) -> np.ndarray:
    """The `vertex` element of a PLY file as a numpy structured array (native byte order).
    Elements before `vertex` must be fixed-size (no list properties); anything after it is ignored."""
    with open(path, "rb") as f:
        fmt, elements = _read_header(f)
        order = {"ascii": "=", "binary_little_endian": "<", "binary_big_endian": ">"}[fmt]
        for name, count, props in elements:
            has_list = any(code is None for _, code in props)
            if name != "vertex":
                if has_list:
                    raise PlyFormatError(f"variable-size element {name!r} precedes 'vertex'")
                if fmt == "ascii":
                    for _ in range(count):
                        f.readline()
                else:
                    f.seek(count * sum(np.dtype(code).itemsize for _, code in props), os.SEEK_CUR)
                continue
            if has_list:
                raise PlyFormatError("list properties on the vertex element are not supported")
            if len({n for n, _ in props}) != len(props):
                raise PlyFormatError("duplicate vertex property names")
            native = np.dtype([(n, code) for n, code in props])
            if fmt == "ascii":
                out = np.empty(count, dtype=native)
                for i in range(count):
                    tok = f.readline().split()
                    if len(tok) < len(props):
                        raise PlyFormatError(f"vertex {i}: expected {len(props)} values, got {len(tok)}")
                    out[i] = tuple(np.dtype(code).type(float(t) if code[0] == "f" else int(t))
                                   for t, (_, code) in zip(tok, props))
                return out
            disk = np.dtype([(n, order + code) for n, code in props])
            raw = f.read(count * disk.itemsize)
            if len(raw) != count * disk.itemsize:
                raise PlyFormatError(f"truncated vertex data: {len(raw)} of {count * disk.itemsize} bytes")
            return np.frombuffer(raw, dtype=disk, count=count).astype(native)
    raise PlyFormatError("PLY file has no vertex element")


def write_ply_vertices(path, vertices: np.ndarray, comments: Sequence[str] = ()):
    """Writes one `vertex` element as binary_little_endian PLY (the layout plyfile's PlyData([el]).write emits:
    header lines `property <type> <name>` in field order, then packed records)."""
    if vertices.dtype.names is None:
        raise PlyFormatError("write_ply_vertices needs a structured array")
    lines = ["ply", "format binary_little_endian 1.0"] + [f"comment {c}" for c in comments]
    lines.append(f"element vertex {len(vertices)}")
    disk = []
    for n in vertices.dtype.names:
        dt = vertices.dtype[n]
        code = dt.kind + str(dt.itemsize)
        if code not in _NP_TO_PLY:
            raise PlyFormatError(f"field {n!r}: dtype {dt} has no PLY scalar type")
        lines.append(f"property {_NP_TO_PLY[code]} {n}")
        disk.append((n, "<" + code))
    lines.append("end_header")
    with open(path, "wb") as f:
        f.write(("\n".join(lines) + "\n").encode("ascii"))
        f.write(np.ascontiguousarray(vertices.astype(np.dtype(disk))).tobytes())


DEFAULT_ATTRIBUTES = ("xyz", "f_dc", "f_rest", "opacity", "scale", "rot")


@dataclass
class GaussianCloud:
    """Raw (pre-activation) per-Gaussian parameters in the reference's storage convention
    (models/vanilla_sg.py: `_xyz [P,3]`, `_f_dc`, `_f_rest`, `_opacity [P,1]`, `_scale [P,3]`, `_rot [P,4]`)."""
    xyz: torch.Tensor
    f_dc: torch.Tensor
    f_rest: torch.Tensor
    opacity: torch.Tensor
    scale: torch.Tensor
    rot: torch.Tensor
    extra: Dict[str, torch.Tensor] = field(default_factory=dict)

    @property
    def num_points(self):
        return self.xyz.shape[0]

    @property
    def max_sh_degree(self):
        return int(round(math.sqrt((self.f_dc.numel() + self.f_rest.numel()) // max(self.num_points, 1) // 3))) - 1

    def to(self, device):
        return GaussianCloud(*(t.to(device) for t in (self.xyz, self.f_dc, self.f_rest, self.opacity, self.scale,
                                                      self.rot)), {k: v.to(device) for k, v in self.extra.items()})

    def split_features(self, channel_major=False):
        """(f_dc [P,1,3], f_rest [P,M-1,3]) as `FusedGaussianRasterizer` takes them.
        channel_major=False: the reference's get_features reshape (vanilla_sg.py:104-105), quirk included.
        channel_major=True: interpret the stored f_rest_* as [P,3,M-1] (how export / the 3DGS trainer wrote it)."""
        P = self.num_points
        if channel_major:
            return (self.f_dc.reshape(P, 3, -1).transpose(1, 2).contiguous(),
                    self.f_rest.reshape(P, 3, -1).transpose(1, 2).contiguous())
        return self.f_dc.reshape(P, -1, 3), self.f_rest.reshape(P, -1, 3)

    def features(self, channel_major=False):
        """[P,M,3] SH coefficients = get_features (vanilla_sg.py:102-106)."""
        return torch.cat(self.split_features(channel_major), dim=1)

    def activated(self):
        """Post-activation tensors as VanillaRenderer.get_gaussians_properties hands them to the op
        (renderers/vanilla_renderer.py:28-51 with the default activations sigmoid / exp / normalize)."""
        return dict(means3D=self.xyz, opacities=torch.sigmoid(self.opacity), scales=torch.exp(self.scale),
                    rotations=torch.nn.functional.normalize(self.rot), shs=self.features())


def load_gaussian_ply(path, device="cpu", attributes: Sequence[str] = DEFAULT_ATTRIBUTES) -> GaussianCloud:
    """models/base.py:73-105 for the vanilla attribute set: float32 tensors, `[P, n_matching_properties]` each."""
    v = read_ply_vertices(path)
    names = list(v.dtype.names)
    got: Dict[str, torch.Tensor] = {}
    needed = {"xyz": ("x", "y", "z"), "opacity": ("opacity",), "rgb": ("red", "green", "blue")}
    absent = [n for a in attributes for n in needed.get(a, ()) if n not in names]
    if absent:
        raise PlyFormatError(f"{path}: missing Gaussian attributes (vertex properties {absent})")
    for elem in attributes:
        if elem == "xyz":
            arr = np.stack((v["x"], v["y"], v["z"]), axis=1)
        elif elem == "opacity":
            arr = v["opacity"][..., np.newaxis]
        elif elem == "rgb":
            arr = np.stack((v["red"], v["green"], v["blue"]), axis=1).astype(np.float32) / 255
        else:
            cols = [n for n in names if n.startswith(elem)]
            try:
                cols = sorted(cols, key=lambda n: int(n.split("_")[-1]))
            except ValueError as e:
                raise PlyFormatError(f"attribute {elem!r}: property without a trailing index ({e})") from None
            if not cols:
                continue
            arr = np.zeros((len(v), len(cols)))
            for i, n in enumerate(cols):
                arr[:, i] = v[n]
        got[elem] = torch.from_numpy(np.ascontiguousarray(arr)).float().to(device)
    missing = [a for a in DEFAULT_ATTRIBUTES if a not in got and a != "f_rest"]
    if missing:
        raise PlyFormatError(f"{path}: missing Gaussian attributes {missing}")
    P = got["xyz"].shape[0]
    f_rest = got.pop("f_rest", torch.zeros(P, 0, dtype=torch.float32, device=device))
    core = {k: got.pop(k) for k in ("xyz", "f_dc", "opacity", "scale", "rot")}
    return GaussianCloud(core["xyz"], core["f_dc"], f_rest, core["opacity"], core["scale"], core["rot"], got)


def gaussian_ply_fields(cloud: GaussianCloud) -> List[str]:
    """construct_list_of_attributes (models/vanilla_sg.py:161-181)."""
    P = max(cloud.num_points, 1)
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(cloud.f_dc.numel() // P)]
    names += [f"f_rest_{i}" for i in range(cloud.f_rest.numel() // P)]
    names.append("opacity")
    names += [f"scale_{i}" for i in range(cloud.scale.shape[1])]
    names += [f"rot_{i}" for i in range(cloud.rot.shape[1])]
    return names


def export_gaussian_ply(path, cloud: GaussianCloud):
    """models/vanilla_sg.py:144-159: one float32 record per Gaussian, normals zero, SH written channel-major."""
    P = cloud.num_points
    cm = lambda t: t.detach().reshape(P, -1, 3).transpose(1, 2).flatten(start_dim=1).cpu().numpy()
    xyz = cloud.xyz.detach().cpu().numpy()
    cols = np.concatenate((xyz, np.zeros_like(xyz), cm(cloud.f_dc), cm(cloud.f_rest),
                           cloud.opacity.detach().cpu().numpy().reshape(P, -1), cloud.scale.detach().cpu().numpy(),
                           cloud.rot.detach().cpu().numpy()), axis=1).astype(np.float32)
    names = gaussian_ply_fields(cloud)
    if cols.shape[1] != len(names):
        raise PlyFormatError(f"attribute count mismatch: {cols.shape[1]} columns vs {len(names)} names")
    rec = np.empty(P, dtype=[(n, "f4") for n in names])
    for i, n in enumerate(names):
        rec[n] = cols[:, i]
    write_ply_vertices(path, rec)


# ------------------------------------------------------------------ cameras.json

@dataclass
class CameraRecord:
    """What the rasterizer needs from gaustudio's Camera (datasets/__init__.py:113-183)."""
    id: int
    image_name: str
    image_width: int
    image_height: int
    R: np.ndarray            # [3,3] camera-to-world rotation (3DGS convention: W2C[:3,:3] = R^T)
    T: np.ndarray            # [3]   world-to-camera translation
    FoVx: float
    FoVy: float
    znear: float = 0.1
    zfar: float = 100.0

    @property
    def cam(self) -> scenes.Cam:
        """Matrices as Camera._setup builds them (:154-159,182-183) and the tangents renderers/base.py:20-21 uses."""
        Rt = np.zeros((4, 4))
        Rt[:3, :3] = self.R.transpose()
        Rt[:3, 3] = self.T
        Rt[3, 3] = 1.0
        c2w = np.linalg.inv(Rt)                       # getWorld2View2 with translate 0, scale 1 (:52-64)
        c2w[:3, 3] = (c2w[:3, 3] + np.array([0.0, 0.0, 0.0])) * 1.0
        view = torch.tensor(np.float32(np.linalg.inv(c2w))).transpose(0, 1)
        proj = scenes.projection_matrix(self.znear, self.zfar, self.FoVx, self.FoVy).transpose(0, 1)
        full = (view.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
        campos = torch.inverse(view)[3][:3]
        return scenes.Cam(int(self.image_width), int(self.image_height), math.tan(self.FoVx * 0.5),
                          math.tan(self.FoVy * 0.5), view.contiguous(), full.contiguous(), campos.contiguous())


def focal2fov(focal, pixels):
    return 2 * math.atan(pixels / (2 * focal))


def fov2focal(fov, pixels):
    return pixels / (2 * math.tan(fov / 2))


def camera_from_json(entry: dict) -> CameraRecord:
    """utils/cameras_utils.py:8-38: position/rotation are camera-to-world; cx/cy are not read."""
    c2w = np.eye(4)
    c2w[:3, :3] = np.array(entry["rotation"])
    c2w[:3, 3] = np.array(entry["position"])
    w2c = np.linalg.inv(c2w)
    return CameraRecord(id=entry["id"], image_name=entry["img_name"], image_width=entry["width"],
                        image_height=entry["height"], R=w2c[:3, :3].transpose(), T=w2c[:3, 3],
                        FoVx=focal2fov(entry["fx"], entry["width"]), FoVy=focal2fov(entry["fy"], entry["height"]))


def camera_to_json(id: int, rec: CameraRecord) -> dict:
    """datasets/utils.py:58-80 (principal point at the image centre, as every camera read from JSON has)."""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = rec.R.transpose()
    Rt[:3, 3] = rec.T
    Rt[3, 3] = 1.0
    c2w = np.linalg.inv(Rt)
    return {"id": id, "img_name": rec.image_name, "width": rec.image_width, "height": rec.image_height,
            "position": c2w[:3, 3].tolist(), "rotation": [row.tolist() for row in c2w[:3, :3]],
            "fy": fov2focal(rec.FoVy, rec.image_height), "fx": fov2focal(rec.FoVx, rec.image_width),
            "cy": rec.image_height * 0.5, "cx": rec.image_width * 0.5}


def load_cameras_json(path) -> List[CameraRecord]:
    """datasets/vanilla.py:20-31: every entry through JSON_to_camera, sorted by image name."""
    with open(path, "r") as f:
        data = json.load(f)
    return sorted((camera_from_json(e) for e in data), key=lambda c: c.image_name)


def save_cameras_json(path, cameras: Sequence[CameraRecord]):
    """datasets/vanilla.py:34-41."""
    with open(path, "w") as f:
        json.dump([camera_to_json(i, c) for i, c in enumerate(cameras)], f)
