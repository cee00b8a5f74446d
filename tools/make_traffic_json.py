#!/usr/bin/env python
"""Builds profiles/rNN_traffic.json (read by bench.py) from the PMC summaries of one tools/profile_session.sh run:
per kernel, HBM traffic per launch (FETCH_SIZE / WRITE_SIZE in KiB, separate passes; calibration factors from
profiles/r02_fetch_write_calibration.txt: the compositing kernels' record gathers read 1.0x, their scattered 48-B
row writes are already counted at their 64-B cost) and the VALU / SALU / LDS instruction counts.
The compositing kernels read two kinds of streams: 64-B record gathers (FETCH_SIZE counts them 1.0x) and COALESCED
streams -- composite_bwd the per-pixel state and upstream gradients (H*W*36 B) and the ids + block masks (R*6 B),
composite_fwd the ids (R*4 B) -- which FETCH_SIZE counts at HALF their size (the guide's 32-B-unit case, reproduced by
tools/fetch_calib.hip).  traffic_bytes therefore adds the uncounted half of those streams.
usage: python tools/make_traffic_json.py gpurun_out/<session> C3 profiles/r03_traffic.json [gpurun_out/<rotating session>]"""
import json
import os
import re
import sys

sess, wl, out = sys.argv[1], sys.argv[2], sys.argv[3]
sess_rot = sys.argv[4] if len(sys.argv) > 4 else None
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (workload dimensions)
_P, _W, _H, _D, _ = bench.WORKLOADS[wl]


def parse(path):
    """{(kernel short name, grid): {counter: mean}}"""
    res, cur = {}, None
    for line in open(path):
        m = re.match(r"^(?:void )?gsr::(\w+).*grid=(\d+)", line)
        if m:
            cur = res.setdefault((m.group(1), int(m.group(2))), {})
            continue
        m = re.match(r"^\s+(\w+)\s+mean\s+([\d.]+)", line)
        if m and cur is not None:
            cur[m.group(1)] = float(m.group(2))
    return res


sq = parse(f"{sess}/{wl}_pmc_sq.txt")
fe = parse(f"{sess}/{wl}_pmc_fetch.txt")
wr = parse(f"{sess}/{wl}_pmc_write.txt")
# instances staged per launch, for the coalesced-stream correction: from the bench line of the stats pass when present
R_binned = None
for cand in (f"{sess}/bench_line.json", f"{sess}/{wl}_bench_line.json"):
    if os.path.exists(cand):
        try:
            R_binned = json.loads(open(cand).read().strip().splitlines()[-1])["config"]["instances_binned"]
        except Exception:
            pass


def coalesced_read_bytes(name, fetch_kb=0.0):
    R = R_binned or 0
    if name.startswith("composite_bwd"):
        return _H * _W * 36 + R * 6
    if name.startswith("composite_fwd"):
        return R * 4
    if name.startswith("preprocess_"):
        return int(2 * fetch_kb * 1024)     # pure streaming kernels: EVERY read is a coalesced stream -> 2 x FETCH_SIZE
    return 0
doc = {"_comment": "per-launch counters of the benchmark-sized dispatches (largest grid of each kernel) from rocprofv3 PMC passes "
                   f"of `bench.py --workload {wl}` ({sess}); FETCH_SIZE / WRITE_SIZE are KiB; traffic_bytes = FETCH + WRITE with the "
                   "factors calibrated in profiles/r02_fetch_write_calibration.txt for these kernels' access patterns (64-B record "
                   "gather: 1.0), traffic_upper_bytes = 2*FETCH + WRITE (the coalesced-stream factor applied to every read)",
       "workload": wl}
# (the per-quarter compositing kernels are the default ones; the per-wave variants only show up in A/B runs)
for name in ("composite_fwd_quarter_kernel", "composite_bwd_quarter_kernel", "composite_fwd_kernel", "composite_bwd_kernel",
             "preprocess_fwd_kernel", "preprocess_bwd_kernel", "preprocess_bwd_sh_coop_kernel", "bin_chunk_kernel",
             "tile_sort_kernel"):
    grids = [g for (n, g) in sq if n == name]
    if not grids:
        continue
    g = max(grids)
    s, f, w = sq.get((name, g), {}), fe.get((name, g), {}), wr.get((name, g), {})
    key = name.replace("_quarter_kernel", "").replace("_kernel", "")
    if key in doc:
        continue
    co = coalesced_read_bytes(name, f.get("FETCH_SIZE", 0))
    doc[key] = {"grid": g, "fetch_size_kb": f.get("FETCH_SIZE"), "write_size_kb": w.get("WRITE_SIZE"),
                "coalesced_read_bytes_counted_at_half": co,
                "traffic_bytes": int((f.get("FETCH_SIZE", 0) + w.get("WRITE_SIZE", 0)) * 1024 + co // 2),
                "traffic_upper_bytes": int((2 * f.get("FETCH_SIZE", 0) + w.get("WRITE_SIZE", 0)) * 1024),
                "valu_insts": int(s.get("SQ_INSTS_VALU", 0)), "salu_insts": int(s.get("SQ_INSTS_SALU", 0)),
                "lds_insts": int(s.get("SQ_INSTS_LDS", 0)), "valu_cycles_per_inst_model": 2.7, "kernel": name}
# the same counters with a different camera every step and an optimizer update between the steps (bench.py --rotate-cameras 8)
if sess_rot:
    fe2, wr2 = parse(f"{sess_rot}/{wl}_pmc_fetch.txt"), parse(f"{sess_rot}/{wl}_pmc_write.txt")
    by_mode = {}
    for name in ("composite_fwd_quarter_kernel", "composite_bwd_quarter_kernel", "preprocess_fwd_kernel", "preprocess_bwd_kernel",
                 "preprocess_bwd_sh_coop_kernel"):
        key = name.replace("_quarter_kernel", "").replace("_kernel", "")
        grids = [g for (n, g) in fe2 if n == name]
        if not grids or key not in doc:
            continue
        g = max(grids)
        co = coalesced_read_bytes(name, fe2[(name, g)].get("FETCH_SIZE", 0))
        rot = int((fe2[(name, g)].get("FETCH_SIZE", 0) + wr2.get((name, g), {}).get("WRITE_SIZE", 0)) * 1024 + co // 2)
        by_mode[key] = {"static_bytes": doc[key]["traffic_bytes"], "rotating_bytes": rot}
    doc["by_mode"] = by_mode
# fingerprint of the kernel sources these counters were captured with: bench.py reports whether it still matches
import hashlib
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
h = hashlib.sha1()
for f in ("gsr_kernels_fwd.hip", "gsr_kernels_bwd.hip", "gsr_common.h", "Makefile"):
    h.update(open(os.path.join(root, "gaustudio_amd", "csrc", f), "rb").read())
doc["_kernel_sources_sha1"] = h.hexdigest()
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps(doc, indent=1)[:1500])
