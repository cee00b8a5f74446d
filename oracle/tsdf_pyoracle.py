"""ctypes front-end of oracle/tsdf_oracle.c + a dict-based marching cubes.  TEST INFRASTRUCTURE ONLY (tests/ and
measurement scripts); the product never imports it.  PARITY UNPINNED against vdbfusion (absent here), see the header
of tsdf_oracle.c.

`extract_mesh` restates VDBVolume::ExtractTriangleMesh of vdbfusion (itself the Open3D-style marching cubes the paper
cites): cubes anchored at observed voxels, skipped when a corner has weight < min_weight (or zero weight without
fill_holes), one shared vertex per crossing edge at |f0| / (|f0| + |f1|) along the edge from the voxel centre, triangle
tables from gaustudio_amd/csrc/gen_mc_tables.py (derived, not vdbfusion's transcribed table: triangulation of a cube's
polygon may differ, the surface does not)."""
import ctypes
import importlib.util
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libtsdforacle.so")
        src = os.path.join(_HERE, "tsdf_oracle.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-B", "libtsdforacle.so"], stdout=subprocess.DEVNULL)
        _LIB = ctypes.CDLL(so)
        _LIB.tso_create.restype = ctypes.c_void_p
        _LIB.tso_create.argtypes = [ctypes.c_float, ctypes.c_float, ctypes.c_int]
        _LIB.tso_destroy.argtypes = [ctypes.c_void_p]
        _LIB.tso_integrate.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
        _LIB.tso_num_voxels.restype = ctypes.c_int64
        _LIB.tso_num_voxels.argtypes = [ctypes.c_void_p]
        _LIB.tso_export.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 4
    return _LIB


class Volume:
    def __init__(self, voxel_size, sdf_trunc, space_carving=False):
        self.voxel_size, self.sdf_trunc = np.float32(voxel_size), np.float32(sdf_trunc)
        self._h = ctypes.c_void_p(lib().tso_create(float(voxel_size), float(sdf_trunc), int(space_carving)))

    def close(self):
        if getattr(self, "_h", None) and _LIB is not None:
            _LIB.tso_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:          # interpreter shutdown
            pass

    def integrate(self, points, origin):
        p = np.ascontiguousarray(points, dtype=np.float32)
        o = np.ascontiguousarray(origin, dtype=np.float32)
        lib().tso_integrate(self._h, p.ctypes.data, p.shape[0], o.ctypes.data)

    def export(self):
        n = lib().tso_num_voxels(self._h)
        coords = np.zeros((n, 3), np.int32); tsdf = np.zeros(n, np.float32)
        weight = np.zeros(n, np.int32); sum_q = np.zeros(n, np.int64)
        if n:
            lib().tso_export(self._h, coords.ctypes.data, tsdf.ctypes.data, weight.ctypes.data, sum_q.ctypes.data)
        return coords, tsdf, weight, sum_q


def _tables():
    path = os.path.join(os.path.dirname(_HERE), "gaustudio_amd", "csrc", "gen_mc_tables.py")
    spec = importlib.util.spec_from_file_location("gen_mc_tables", path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def extract_mesh(coords, weight, sum_q, voxel_size, sdf_trunc, min_weight=0.5, fill_holes=True, blocks=None):
    """Marching cubes over the observed voxels given as the fixed-point sums the GPU keeps (mean tsdf = sum_q / weight *
    sdf_trunc / 2^15, evaluated in float32 like the kernel).  `blocks`: set of allocated 8^3 block coordinates (a
    corner in a block that was never allocated makes the cube non-extractable, as in the kernel); default: the
    blocks of the observed voxels.  Returns (vertices [nv,3] f32, triangles [nt,3] i32)."""
    g = _tables()
    table, _ = g.build()
    vs, tr = np.float32(voxel_size), np.float32(sdf_trunc)
    scale = np.float32(tr / np.float32(32768.0))
    vox = {}
    for c, w, s in zip(map(tuple, coords.tolist()), weight.tolist(), sum_q.tolist()):
        vox[c] = (w, np.float32(np.float32(s) / np.float32(w)) * scale)
    if blocks is None:
        blocks = {(x >> 3, y >> 3, z >> 3) for (x, y, z) in vox}
    min_count = 0 if min_weight <= 0 else int(np.ceil(np.float32(min_weight)))
    OWNER = [0, 1, 3, 0, 4, 5, 7, 4, 0, 1, 2, 3]
    AXIS = [0, 1, 0, 1, 0, 1, 0, 1, 2, 2, 2, 2]
    half = np.float32(vs * np.float32(0.5))
    verts, vidx, tris = [], {}, []

    def get(c):
        if (c[0] >> 3, c[1] >> 3, c[2] >> 3) not in blocks:
            return None
        return vox.get(c, (0, tr))

    # every cube with at least one observed corner, in (z, y, x) order of its minimum corner (cubes without an observed
    # corner are uniformly +sdf_trunc and produce nothing)
    anchors = sorted({(x - i, y - j, z - k) for (x, y, z) in vox for i in (0, 1) for j in (0, 1) for k in (0, 1)},
                     key=lambda c: (c[2], c[1], c[0]))
    for v in anchors:
        f = []
        ok = True
        for off in g.CORNERS:
            r = get((v[0] + off[0], v[1] + off[1], v[2] + off[2]))
            if r is None or (not fill_holes and r[0] == 0) or r[0] < min_count:
                ok = False
                break
            f.append(r[1])
        if not ok:
            continue
        case = sum(1 << i for i in range(8) if f[i] < 0)
        if case in (0, 255):
            continue
        for t in table[case]:
            tri = []
            for e in t:
                o = OWNER[e]
                oc = (v[0] + g.CORNERS[o][0], v[1] + g.CORNERS[o][1], v[2] + g.CORNERS[o][2])
                key = (oc, AXIS[e])
                if key not in vidx:
                    a, b = g.EDGES[e]
                    if a != o:
                        a, b = b, a
                    f0, f1 = np.abs(f[a]), np.abs(f[b])
                    p = [np.float32(half + vs * np.float32(oc[d])) for d in range(3)]
                    p[AXIS[e]] = np.float32(p[AXIS[e]] + np.float32(np.float32(f0 * vs) / np.float32(f0 + f1)))
                    vidx[key] = len(verts)
                    verts.append(p)
                tri.append(vidx[key])
            tris.append(tri)
    return np.array(verts, np.float32).reshape(-1, 3), np.array(tris, np.int32).reshape(-1, 3)
