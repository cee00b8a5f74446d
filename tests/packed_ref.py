"""TEST INFRASTRUCTURE: a torch restatement of the packed row messages of the multi-GPU exchange (csrc/gsr_comm.hip, layout in
include/gsrast.h) on tensors of any device.  Two uses: (1) the gloo / CPU tests of gaustudio_amd.parallel.FactoredGradExchange
(compact="view") run without the HIP library (`packed=TorchPacked`), (2) the GPU tests hold the HIP kernels against it word for
word."""
import torch


def _i32(x):
    """int64 values in [0, 2^32) -> the int32 tensor with the same bits."""
    return torch.where(x >= 2 ** 31, x - 2 ** 32, x).to(torch.int32)


def _u32(x):
    return x.to(torch.int64) & 0xFFFFFFFF


class TorchPacked:
    sh_from_colors = None          # dense reference: fn(means3D, campos, colors[N,P,3], D, out)

    @staticmethod
    def header_words(P):
        w = 4 + (P + 255) // 256 + (P + 31) // 32
        return (w + 3) // 4 * 4

    @staticmethod
    def _fill_header(vis, hdr):
        P = vis.numel()
        nb, nw = (P + 255) // 256, (P + 31) // 32
        v = torch.zeros(nw * 32, dtype=torch.int64, device=vis.device)
        v[:P] = vis.to(torch.int64)
        words = (v.view(nw, 32) << torch.arange(32, device=vis.device, dtype=torch.int64)[None, :]).sum(1)
        vb = torch.zeros(nb * 256, dtype=torch.int64, device=vis.device)
        vb[:P] = vis.to(torch.int64)
        cnt = vb.view(nb, 256).sum(1)
        base = torch.cumsum(cnt, 0) - cnt
        hdr.zero_()
        hdr[0] = int(cnt.sum())
        hdr[1] = P
        hdr[4:4 + nb] = _i32(base)
        hdr[4 + nb:4 + nb + nw] = _i32(words)

    @staticmethod
    def _mask(hdr, P):
        nb, nw = (P + 255) // 256, (P + 31) // 32
        words = _u32(hdr[4 + nb:4 + nb + nw])
        bits = (words[:, None] >> torch.arange(32, device=hdr.device, dtype=torch.int64)[None, :]) & 1
        return bits.reshape(-1)[:P].bool()

    @classmethod
    def visible_index(cls, radii, hdr, scratch):
        cls._fill_header(radii > 0, hdr)

    @classmethod
    def union_index(cls, P, msgs, offsets, out_hdr, scratch):
        Hw = cls.header_words(P)
        vis = torch.zeros(P, dtype=torch.bool, device=msgs.device)
        for o in offsets.tolist():
            vis |= cls._mask(msgs[o:o + Hw], P)
        cls._fill_header(vis, out_hdr)

    @classmethod
    def pack_rows(cls, hdr, src, dst, dst_stride, col0=0):
        P = src.shape[0]
        C = src.numel() // max(P, 1)
        vis = cls._mask(hdr, P)
        K = int(vis.sum())
        d = dst[:K * dst_stride].view(K, dst_stride)
        d[:, col0:col0 + C] = src.reshape(P, C)[vis]

    @classmethod
    def unpack_rows(cls, hdr, src, src_stride, col0, dst):
        P = dst.shape[0]
        C = dst.numel() // max(P, 1)
        vis = cls._mask(hdr, P)
        K = int(vis.sum())
        dst.reshape(P, C)[vis] = src[:K * src_stride].view(K, src_stride)[:, col0:col0 + C]

    @classmethod
    def pack_geometry(cls, hdr, views, rows, flag):
        """csrc/gsr_comm.hip pack_geometry_kernel: rows[r] = [means3D 3 | opacity 1 | scales 3 | rotations 4] of the r-th Gaussian of the
        header; flag |= 1 when a Gaussian outside the header has a non-zero value."""
        P = views["means3D"].shape[0]
        vis = cls._mask(hdr, P)
        allv = torch.cat([views[r].reshape(P, -1) for r in ("means3D", "opacities", "scales", "rotations")], dim=1)
        K = int(vis.sum())
        rows[:K] = allv[vis]
        if bool((allv[~vis] != 0).any()):
            flag |= 1

    @classmethod
    def unpack_geometry(cls, hdr, rows, views):
        P = views["means3D"].shape[0]
        vis = cls._mask(hdr, P)
        K = int(vis.sum())
        col = 0
        for r in ("means3D", "opacities", "scales", "rotations"):
            v = views[r].reshape(P, -1)
            v[vis] = rows[:K, col:col + v.shape[1]]
            col += v.shape[1]

    @classmethod
    def sh_from_packed(cls, means3D, campos, msgs, offsets, D, out):
        P = means3D.shape[0]
        Hw = cls.header_words(P)
        dense = torch.zeros((offsets.numel(), P, 3), dtype=torch.float32, device=msgs.device)
        for r, o in enumerate(offsets.tolist()):
            hdr = msgs[o:o + Hw]
            vis = cls._mask(hdr, P)
            K = int(vis.sum())
            dense[r][vis] = msgs[o + Hw:o + Hw + 3 * K].view(torch.float32).view(K, 3)
        cls.sh_from_colors(means3D, campos, dense, D, out)
