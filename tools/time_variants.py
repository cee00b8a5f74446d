#!/usr/bin/env python
"""A/B timing of prebuilt library variants on the GPU box (tools/build_variant.sh builds them in the dev container):

    python tools/time_variants.py --variants base,coop --workloads C3,C4 --reps 2 --keys forward.preprocess,backward.composite_bwd

For every variant: gpurun_variants/libgsrast_<name>.so is copied over gaustudio_amd/libgsrast.so (the original is restored at the
end), then `python bench.py --workload W --steps S --warmup 10 --no-extras --no-cpu-baseline --no-ref-ab` runs `reps` times per
workload, interleaved across the variants so that clock drift hits them alike.  Prints one table row per (workload, variant):
ms_per_step and the requested stage_ms entries, every repetition."""
import argparse
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "gaustudio_amd", "libgsrast.so")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", required=True)
    ap.add_argument("--workloads", default="C3")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--keys", default="forward.preprocess")
    ap.add_argument("--extra", default="", help="extra bench.py arguments, space separated")
    a = ap.parse_args()
    variants = a.variants.split(",")
    keys = a.keys.split(",")
    backup = LIB + ".orig"
    shutil.copy2(LIB, backup)
    res = {}
    try:
        for rep in range(a.reps):
            for w in a.workloads.split(","):
                for v in variants:
                    shutil.copy2(os.path.join(ROOT, "gpurun_variants", f"libgsrast_{v}.so"), LIB)
                    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", w, "--steps", str(a.steps), "--warmup", "10",
                           "--no-extras", "--no-cpu-baseline", "--no-ref-ab"] + a.extra.split()
                    r = subprocess.run(cmd, capture_output=True, text=True)
                    rows = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                    if r.returncode != 0 or not rows:
                        res.setdefault((w, v), []).append({"error": (r.stderr or r.stdout)[-200:]})
                        continue
                    j = json.loads(rows[-1])
                    row = {"ms_per_step": j["ms_per_step"]}
                    for k in keys:
                        d = j["stage_ms"]
                        for part in k.split("."):
                            d = (d or {}).get(part)
                        row[k] = d
                    res.setdefault((w, v), []).append(row)
    finally:
        shutil.copy2(backup, LIB)
        os.remove(backup)
    for (w, v), rows in res.items():
        cols = ["ms_per_step"] + keys
        txt = "  ".join(f"{c}: " + " / ".join("err" if "error" in r else f"{r[c]:.4f}" for r in rows) for c in cols)
        print(f"{w:10s} {v:16s} {txt}")
        for r in rows:
            if "error" in r:
                print("    ", r["error"])


if __name__ == "__main__":
    main()
