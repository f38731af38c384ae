#!/usr/bin/env python
"""Assemble profiles/counters_at_head.json (what bench.py's roofline reads) and the per-config profile summaries from the
outputs of scripts/profile_head.sh that gpurun merged back under gpurun_out/prof_<cfg>/.

    python scripts/make_counters_at_head.py cfg2 [cfg3 ...]

Refuses counters whose kernel digest differs from the library in the tree (they would be stale by construction)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ahmc_amd as A  # noqa: E402
from ahmc_amd.build import kernel_digest  # noqa: E402

kd = kernel_digest()
head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip()
path = os.path.join(ROOT, "profiles", "counters_at_head.json")
try:
    out = json.load(open(path))
    if out.get("sources_digest") != kd:
        out = None
except Exception:
    out = None
if out is None:
    out = {"sources_digest": kd, "configs": {}}
out["source"] = ("rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_INSTS_* | SQ_ACTIVE_* | GRBM_*; --kernel-trace only) over "
                 "`python bench.py --config <cfg> --warmup 0 --repeats 1 --ess 0 --no-cpu-baseline` (scripts/profile_head.sh), "
                 "distilled by scripts/profile_counters.py; valid for the device code with this digest only")
out["taken_at_commit_after"] = head
for cfg in sys.argv[1:]:
    s = json.load(open(os.path.join(ROOT, "gpurun_out", f"prof_{cfg}", "summary.json")))
    if s.get("kernel_digest") != kd:
        print(f"{cfg}: counters were taken on kernel digest {s.get('kernel_digest')}, the tree has {kd}: NOT used")
        continue
    c = {}
    for mode, r in s["counters"].items():
        c[mode] = {k: r.get(k) for k in ("valu_per_leapfrog", "salu_per_leapfrog", "lds_per_leapfrog", "vmem_per_leapfrog", "mfma_f64_per_leapfrog",
                                          "hbm_bytes_per_leapfrog", "valu_busy", "mean_waves_per_simd")}
    out["configs"][cfg] = c
    keep = {k: s.get(k) for k in ("config", "command", "kernel_digest", "kernel_stats", "mode0_launches", "mode3_launches", "counters", "leapfrogs_by_pass", "per_kernel_counters")}
    keep["bench_plain"] = s.get("bench_plain")
    with open(os.path.join(ROOT, "profiles", f"r2_{cfg}_profile_summary.json"), "w") as f:
        json.dump(keep, f, indent=1)
    with open(os.path.join(ROOT, "profiles", f"r2_{cfg}_top_kernels.txt"), "w") as f:
        f.write(f"# {s.get('command')}\n# rocprofv3 --kernel-trace --stats, device code {kd[:16]}, after commit {head}\n")
        for k in s.get("kernel_stats", []):
            f.write("%10.1f ms %7d calls %10.1f us avg %5.1f %%  %s\n" % (k["total_us"] / 1e3, k["calls"], k["average_us"], k["percent"], k["name"]))
    print(cfg, json.dumps(c)[:400])
with open(path, "w") as f:
    json.dump(out, f, indent=1)
print("wrote", path)
