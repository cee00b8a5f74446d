#!/usr/bin/env python
"""Builds profiles/rNN_traffic.json (read by bench.py) from the PMC summaries of one tools/profile_session.sh run:
per kernel, HBM traffic per launch (FETCH_SIZE / WRITE_SIZE in KiB, separate passes; calibration factors from
profiles/r02_fetch_write_calibration.txt: the compositing kernels' record gathers read 1.0x, their scattered 48-B
row writes are already counted at their 64-B cost) and the VALU / SALU / LDS instruction counts.
usage: python tools/make_traffic_json.py gpurun_out/<session> C3 profiles/r02_traffic.json"""
import json
import re
import sys

sess, wl, out = sys.argv[1], sys.argv[2], sys.argv[3]


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
    doc[key] = {"grid": g, "fetch_size_kb": f.get("FETCH_SIZE"), "write_size_kb": w.get("WRITE_SIZE"),
                "traffic_bytes": int((f.get("FETCH_SIZE", 0) + w.get("WRITE_SIZE", 0)) * 1024),
                "traffic_upper_bytes": int((2 * f.get("FETCH_SIZE", 0) + w.get("WRITE_SIZE", 0)) * 1024),
                "valu_insts": int(s.get("SQ_INSTS_VALU", 0)), "salu_insts": int(s.get("SQ_INSTS_SALU", 0)),
                "lds_insts": int(s.get("SQ_INSTS_LDS", 0)), "valu_cycles_per_inst_model": 2.7, "kernel": name}
# fingerprint of the kernel sources these counters were captured with: bench.py reports whether it still matches
import hashlib, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
h = hashlib.sha1()
for f in ("gsr_kernels_fwd.hip", "gsr_kernels_bwd.hip", "gsr_common.h", "Makefile"):
    h.update(open(os.path.join(root, "gaustudio_amd", "csrc", f), "rb").read())
doc["_kernel_sources_sha1"] = h.hexdigest()
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps(doc, indent=1)[:1500])
