"""Distil the rocprofv3 (rocpd SQLite) outputs of scripts/profile_head.sh for ONE config into a JSON summary:
per-kernel time statistics (kernel trace) and, for each instantiation of the dominant kernel, the PMC counters summed
over ALL its launches in the profiled run divided by the leapfrogs that run reports for the phase the kernel serves
(mode 0 = the draws, mode 3 = the warm-up with adapt! inside).  HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB units;
gfx950 tallies 128-B read requests at 64 B: MI355X_MICROARCH.md, HBM section).

    python scripts/profile_counters.py gpurun_out/prof_cfg2 cfg2 > summary.json
"""
import glob
import json
import os
import re
import sqlite3
import sys
from collections import defaultdict

O, CFG = sys.argv[1], sys.argv[2]
out = {"config": CFG}
try:  # the device-code digest of the library these counters were taken on (advancedhmc.jl_amd/build.py: kernel_digest)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out["kernel_digest"] = open(os.path.join(root, "advancedhmc.jl_amd", "csrc", "libahmc_hip.so.kdigest")).read().strip()
    out["unit_digests"] = json.load(open(os.path.join(root, "advancedhmc.jl_amd", "csrc", "libahmc_hip.so.kdigests")))
    out["command"] = open(os.path.join(O, "cmd.txt")).read().strip()
except OSError:
    pass


def db(name):
    f = glob.glob(os.path.join(O, name, "**", "*_results.db"), recursive=True)
    return sqlite3.connect(f[0]).cursor() if f else None


def bench_json(name):
    try:
        return json.loads(open(os.path.join(O, name + ".json")).read().strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)}


def leapfrogs(b):
    c = b["config"]
    n = c["chains_per_gpu"]
    return {"mode0": c["post_adaptation"]["mean_leapfrogs_per_transition"] * c["n_draws"] * n,
            "mode3": c["warmup_phase"]["mean_leapfrogs_per_transition"] * c["n_adapts"] * n}


MODE_RE = {m: re.compile(r"k_nuts<(double|float), \d+, \d+, %d, \d+>" % m) for m in (0, 3)}
out["bench_plain"] = bench_json("bench_plain")
cur = db("kt")
if cur:
    out["kernel_stats"] = [dict(zip(("name", "calls", "total_us", "average_us", "percent"), r))
                           for r in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 14")]
    out["bench_under_kernel_trace"] = bench_json("kt")
    for m, rx in MODE_RE.items():
        names = [k["name"] for k in out["kernel_stats"] if rx.search(k["name"])]
        if names:
            d = cur.execute("select duration, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, scratch_size from kernels "
                            "where name = ? order by start", (names[0],)).fetchall()
            out[f"mode{m}_launches"] = {"kernel": names[0], "n": len(d), "total_ms": sum(x[0] for x in d) / 1e6,
                                        "shape": dict(zip(("grid", "workgroup", "lds_bytes", "vgpr", "agpr", "sgpr", "scratch"), d[-1][1:]))}

per_mode = {"mode0": defaultdict(float), "mode3": defaultdict(float)}
lf_by_pass = {}
for name in ("fetch", "write", "sq1", "sq2", "sq3", "grbm"):
    cur = db(name)
    b = bench_json(name)
    if not cur or "config" not in b:
        out.setdefault("missing_passes", []).append(name)
        continue
    lf = leapfrogs(b)
    lf_by_pass[name] = lf
    knames = [r[0] for r in cur.execute("select distinct kernel_name from counters_collection")]
    for m, rx in MODE_RE.items():
        for kn in knames:
            if not rx.search(kn):
                continue
            for cn, v in cur.execute("select counter_name, sum(value) from counters_collection where kernel_name = ? group by counter_name", (kn,)):
                if name == "sq3" and cn == "SQ_INSTS_VALU":
                    cn = "SQ_INSTS_VALU_MIXPASS"   # (also taken by sq1: keep the two passes' totals apart)
                per_mode[f"mode{m}"][cn] += v / lf[f"mode{m}"]   # per leapfrog, with THIS pass's leapfrog count
out["leapfrogs_by_pass"] = lf_by_pass
# every kernel of the run: counters summed over all its dispatches (what the dense engine's kernels — k_dgemm, k_d_tree —
# are read from: bytes moved, MFMA instructions, time from the kernel trace)
per_kernel = defaultdict(lambda: defaultdict(float))
for name in ("fetch", "write", "sq1", "sq2", "sq3", "grbm"):
    cur = db(name)
    if not cur:
        continue
    for kn, cn, v in cur.execute("select kernel_name, counter_name, sum(value) from counters_collection group by kernel_name, counter_name"):
        if name == "sq3" and cn == "SQ_INSTS_VALU":
            cn = "SQ_INSTS_VALU_MIXPASS"
        per_kernel[kn][cn] += v
tk = {k["name"]: k for k in out.get("kernel_stats", [])}
rows = []
for kn, c in per_kernel.items():
    if kn not in tk:
        continue
    r = {"kernel": kn, "calls": tk[kn]["calls"], "total_ms": tk[kn]["total_us"] / 1e3, "percent": tk[kn]["percent"]}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        r["hbm_gbytes"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / 1e9
        r["hbm_tb_per_s_by_trace_time"] = r["hbm_gbytes"] / 1e3 / (r["total_ms"] / 1e3) if r["total_ms"] else None
    for k2 in ("SQ_INSTS_VALU", "SQ_INSTS_VALU_MFMA_F64", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES"):
        if k2 in c:
            r[k2] = c[k2]
    if "SQ_ACTIVE_INST_VALU" in c and "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"]:
        r["valu_busy"] = c["SQ_ACTIVE_INST_VALU"] * 4 / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)
    rows.append(r)
out["per_kernel_counters"] = sorted(rows, key=lambda r: -r["total_ms"])
res = {}
for mode, c in per_mode.items():
    if not c:
        continue
    r = {"per_leapfrog": dict(c)}
    if "SQ_INSTS_VALU" in c:
        r["valu_per_leapfrog"] = c["SQ_INSTS_VALU"]
        r["salu_per_leapfrog"] = c.get("SQ_INSTS_SALU")
        r["lds_per_leapfrog"] = c.get("SQ_INSTS_LDS")
        r["vmem_per_leapfrog"] = (c.get("SQ_INSTS_VMEM_RD", 0) + c.get("SQ_INSTS_VMEM_WR", 0))
        r["mfma_f64_per_leapfrog"] = c.get("SQ_INSTS_VALU_MFMA_F64")
    if "SQ_INSTS_VALU_FMA_F64" in c:
        # dynamic class mix (its own pass: SQ_INSTS_VALU of THAT pass is the denominator of the shares)
        r["valu_mix_pass_total_per_leapfrog"] = c.get("SQ_INSTS_VALU_MIXPASS")
        r["valu_mix_per_leapfrog"] = {k[len("SQ_INSTS_VALU_"):].lower(): c.get(k) for k in
                                      ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64",
                                       "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_CVT")}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        r["hbm_bytes_per_leapfrog"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
    if "SQ_ACTIVE_INST_VALU" in c and "GRBM_GUI_ACTIVE" in c:
        # SQ_ACTIVE_INST_* count quad-cycles summed over SIMDs; GRBM_GUI_ACTIVE (summed over 8 XCD instances) / 8 = GPU cycles
        gpu_cycles = c["GRBM_GUI_ACTIVE"] / 8
        r["valu_busy"] = c["SQ_ACTIVE_INST_VALU"] * 4 / (gpu_cycles * 1024)
        if "SQ_WAVE_CYCLES" in c:
            r["mean_waves_per_simd"] = c["SQ_WAVE_CYCLES"] * 4 / (gpu_cycles * 1024)
    res[mode] = r
out["counters"] = res
print(json.dumps(out, indent=1))
