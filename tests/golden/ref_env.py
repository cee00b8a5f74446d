"""TEST INFRASTRUCTURE (dev container only): an import environment in which the UNMODIFIED Python classes of the reference
(/root/reference/gaustudio: renderers/base.py, vanilla_renderer.py, pcd_renderer.py, models/base.py, models/vanilla_sg.py)
can be executed on a box without a GPU and without the reference's un-vendored third-party packages.

  * `plyfile` is absent from the image.  `PlyStandIn` below is a ~60-line stand-in for the four things the reference's
    loader / exporter use (models/base.py:73-105, models/vanilla_sg.py:144-159): PlyData.read(path), plydata['vertex'].count,
    plydata.elements[0][name] / .properties[i].name, PlyElement.describe(structured_array, 'vertex'), PlyData([el]).write(path).
    It stores ONE `vertex` element as binary_little_endian PLY 1.0.  What the fixtures pin is the REFERENCE'S OWN LOGIC on
    top of that container (attribute naming and ordering, the f_dc / f_rest channel-major export, the get_features reshape
    quirk) -- not the third-party container code, which stays "parity unpinned" (DESIGN.md s8).
  * trimesh / open3d / skimage / omegaconf ... are attribute sinks (never touched by the classes executed here).
  * renderers/base.py:13 creates its screen-space carrier with device="cuda"; on this GPU-less box the module's global
    `torch` is replaced by a proxy that records the requested device and creates the tensor on the CPU.  No source line
    of the reference is edited or copied.
"""
import contextlib
import sys
import types

import numpy as np
import torch

REF_ROOT = "/root/reference"


# ---------------------------------------------------------------------------------------------- plyfile stand-in
_PLY = {"f4": "float", "f8": "double", "u1": "uchar", "i4": "int", "u4": "uint", "i2": "short", "u2": "ushort", "i1": "char"}
_NP = {v: k for k, v in _PLY.items()}


class _Prop:
    def __init__(self, name):
        self.name = name


class PlyElement:
    def __init__(self, name, data):
        self.name, self.data = name, data
        self.count = len(data)
        self.properties = tuple(_Prop(n) for n in data.dtype.names)

    @staticmethod
    def describe(data, name):
        assert data.dtype.names is not None, "PlyElement.describe needs a structured array"
        return PlyElement(name, data)

    def __getitem__(self, key):
        return self.data[key]


class PlyData:
    def __init__(self, elements):
        self.elements = list(elements)

    def __getitem__(self, name):
        return next(e for e in self.elements if e.name == name)

    def write(self, path):
        (el,) = self.elements
        lines = ["ply", "format binary_little_endian 1.0", f"element {el.name} {el.count}"]
        disk = []
        for n in el.data.dtype.names:
            dt = el.data.dtype[n]
            code = dt.kind + str(dt.itemsize)
            lines.append(f"property {_PLY[code]} {n}")
            disk.append((n, "<" + code))
        lines.append("end_header")
        with open(path, "wb") as f:
            f.write(("\n".join(lines) + "\n").encode("ascii"))
            f.write(np.ascontiguousarray(el.data.astype(np.dtype(disk))).tobytes())

    @staticmethod
    def read(path):
        with open(path, "rb") as f:
            assert f.readline().strip() == b"ply"
            assert f.readline().split() == [b"format", b"binary_little_endian", b"1.0"]
            name, count, props = None, 0, []
            while True:
                tok = f.readline().decode("ascii").split()
                if tok[0] == "end_header":
                    break
                if tok[0] == "element":
                    assert name is None, "stand-in: one element only"
                    name, count = tok[1], int(tok[2])
                elif tok[0] == "property":
                    props.append((tok[2], "<" + _NP[tok[1]]))
            data = np.frombuffer(f.read(), dtype=np.dtype(props), count=count)
        return PlyData([PlyElement(name, data.astype(np.dtype([(n, c[1:]) for n, c in props])))])


def plyfile_module():
    m = types.ModuleType("plyfile")
    m.PlyData, m.PlyElement = PlyData, PlyElement
    return m


# ---------------------------------------------------------------------------------------------- attribute sinks
class _Any:
    def __getattr__(self, k):
        return _Any()

    def __call__(self, *a, **k):
        return _Any()


class _Stub(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _Any()


class CudaToCpuTorch(types.ModuleType):
    """`torch` as seen by ONE reference module on a GPU-less box: factory calls with device="cuda" create on the CPU and the
    request is recorded in `.requested`."""

    def __init__(self):
        super().__init__("torch")
        self.requested = []

    def __getattr__(self, k):
        real = getattr(torch, k)
        if k in ("zeros_like", "ones_like", "zeros", "ones", "empty", "tensor"):
            def f(*a, **kw):
                if str(kw.get("device")) == "cuda":
                    self.requested.append((k, "cuda"))
                    kw = dict(kw, device="cpu")
                return real(*a, **kw)
            return f
        return real


@contextlib.contextmanager
def reference_modules(rasterizer_module=None):
    """Inside: `import gaustudio.renderers`, `gaustudio.models` ... resolve to the UNMODIFIED reference sources.
    rasterizer_module: what `gaustudio_diff_gaussian_rasterization` resolves to (default: this repository's drop-in)."""
    stubs = {"plyfile": plyfile_module()}
    for name in ("trimesh", "open3d", "skimage", "skimage.measure", "omegaconf", "cv2", "vdbfusion", "mcubes", "kornia", "kiui",
                 "torch_scatter", "pytorch3d", "pytorch3d.ops", "simple_knn", "simple_knn._C", "einops", "tqdm"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:       # noqa: BLE001
                stubs[name] = _Stub(name)
    saved = {k: sys.modules.get(k) for k in stubs}
    if rasterizer_module is not None:
        saved["gaustudio_diff_gaussian_rasterization"] = sys.modules.get("gaustudio_diff_gaussian_rasterization")
        stubs["gaustudio_diff_gaussian_rasterization"] = rasterizer_module
    sys.modules.update(stubs)
    sys.path.insert(0, REF_ROOT)
    try:
        yield
    finally:
        sys.path.remove(REF_ROOT)
        for k in list(sys.modules):
            if k == "gaustudio" or k.startswith("gaustudio."):
                del sys.modules[k]
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
