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
from ahmc_amd.build import kernel_digest, config_digest, unit_digests, OBJ  # noqa: E402
from ahmc_amd import isa_check  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import valu_mix  # noqa: E402

ROUND = os.environ.get("AHMC_ROUND", "r6")
RATES = os.path.join(ROOT, "profiles", "r3_valu_rate.json")   # scripts/probe/valu_rate.hip on the MI355X


def kernel_text(kernel_name):
    """disassembly of ONE kernel (by its demangled name in the kernel trace) from the object cache of the build"""
    import re
    import tempfile
    m = re.search(r"k_nuts<(double|float), (\d+), (\d+), (\d+), (\d+)>", kernel_name)
    if not m:
        return None
    # (round 4: the warm-up instantiations and the multi-wave geometries are built in the unit's part B, ahmc_inst.hpp: nuts_in_part_b)
    part_b = int(m.group(4)) >= 3 or int(m.group(2)) > 64
    unit = os.path.join(OBJ, f"inst_{'f64' if m.group(1) == 'double' else 'f32'}_t{m.group(5)}{'b' if part_b else ''}.o")
    if not os.path.exists(unit):
        return None
    with tempfile.TemporaryDirectory(prefix="ahmc_isa_") as tmp:
        text = isa_check.disassemble(unit, tmp)
    names = subprocess.run(["c++filt"], input=text, capture_output=True, text=True).stdout
    want = f"k_nuts<{m.group(1)}, {m.group(2)}, {m.group(3)}, {m.group(4)}, {m.group(5)}>"
    out, on = [], False
    for raw, dem in zip(text.splitlines(), names.splitlines()):
        if re.match(r"^[0-9a-f]+ <.*>:$", raw.strip()):
            on = want in dem
        if on:
            out.append(raw)
    return "\n".join(out) if out else None

kd = kernel_digest()
head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip()
path = os.path.join(ROOT, "profiles", "counters_at_head.json")
try:
    out = json.load(open(path))
except Exception:
    out = {"configs": {}}
out["sources_digest"] = kd   # (informational: validity is per config, `unit_digest` below)
FAMILY = {"cfg2": 0, "cfg3": 2, "cfg5": 3}
out["source"] = ("rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_INSTS_* | SQ_ACTIVE_* | GRBM_*; --kernel-trace only) over "
                 "`python bench.py --config <cfg> --warmup 0 --repeats 1 --ess 0 --no-cpu-baseline` (scripts/profile_head.sh), "
                 "distilled by scripts/profile_counters.py; valid for the device code with this digest only")
out["taken_at_commit_after"] = head
for cfg in sys.argv[1:]:
    s = json.load(open(os.path.join(ROOT, "gpurun_out", f"prof_{cfg}", "summary.json")))
    # valid if the two units that hold this config's kernels are the ones the counters were taken on
    ud_then = config_digest(FAMILY[cfg], "f64", s.get("unit_digests") or {}) if cfg in FAMILY else ""
    ud_now = config_digest(FAMILY[cfg]) if cfg in FAMILY else ""
    if cfg in FAMILY and (not ud_then or ud_then != ud_now):
        print(f"{cfg}: counters were taken on other device code (units of family {FAMILY.get(cfg)}: {ud_then[:12]} then, {ud_now[:12]} now): NOT used")
        continue
    c = {"unit_digest": ud_now}
    for mode, r in (s["counters"].items() if cfg in FAMILY else []):  # (cfg4: the dense engine — kernel table and summary only)
        c[mode] = {k: r.get(k) for k in ("valu_per_leapfrog", "salu_per_leapfrog", "lds_per_leapfrog", "vmem_per_leapfrog", "mfma_f64_per_leapfrog",
                                          "hbm_bytes_per_leapfrog", "valu_busy", "mean_waves_per_simd", "valu_mix_per_leapfrog")}
        # the mix-weighted VALU-issue roof of this kernel: measured per-class issue rates (the probe) weighted by its dynamic
        # instruction mix (SQ_INSTS_VALU_* per leapfrog; the classes the hardware does not count are split by the static
        # census of the kernel's hot loop) — scripts/valu_mix.py
        kn = (s.get(f"{mode}_launches") or {}).get("kernel")
        if kn and r.get("valu_mix_per_leapfrog") and os.path.exists(RATES):
            text = kernel_text(kn)
            if text:
                try:
                    # a roof is an upper bound: every class priced at its HIGHEST sustained rate over the probed occupancies
                    a = valu_mix.analyse(text, valu_mix.load_rates(RATES, "max"), r["valu_mix_per_leapfrog"], r["valu_per_leapfrog"], 1, "max")
                    a0 = valu_mix.analyse(text, valu_mix.load_rates(RATES, "max"), r["valu_mix_per_leapfrog"], r["valu_per_leapfrog"], 0, "max")
                    a4 = valu_mix.analyse(text, valu_mix.load_rates(RATES, "W4"), r["valu_mix_per_leapfrog"], r["valu_per_leapfrog"], 1, "W4")
                    c[mode]["valu_peak_mix_gwave_instr_per_s"] = a["peak_mix_nominal_gwave_instr_per_s"]
                    c[mode]["valu_peak_mix"] = {"rates": os.path.relpath(RATES, ROOT),
                                                "priced_at": "nominal issue cycles (2 / 4 / 8 / 16 per wave64 instruction; the class of every instruction type "
                                                             "from its MEASURED rate) at 2.4 GHz x 1024 SIMDs",
                                                "nominal_cycles_by_probe_entry": a["nominal_cycles_by_probe_entry"],
                                                "peak_from_measured_single_class_rates": a["peak_mix_gwave_instr_per_s"],
                                                "peak_if_priced_at_4_waves_per_simd": a4["peak_mix_gwave_instr_per_s"], "kernel": kn,
                                                "class_counts_per_leapfrog": a["dynamic"]["class_counts_per_leapfrog"],
                                                "issue_time_share": a["dynamic"]["issue_time_share"],
                                                "hot_loop_static_valu": a["hot_loop"]["static_valu"],
                                                "sensitivity_innermost_loop_peak": a0["peak_mix_gwave_instr_per_s"]}
                except Exception as ex:  # noqa: BLE001
                    c[mode]["valu_peak_mix_error"] = repr(ex)
    if cfg in FAMILY:
        out["configs"][cfg] = c
    elif cfg == "cfg4":
        # the dense engine: bytes beyond L2 per USEFUL chain-leapfrog over all its kernels (2 x FETCH_SIZE + WRITE_SIZE summed over every
        # dispatch of the fetch / write passes), keyed on the digest of the unit that holds those kernels (`api`)
        api_then, api_now = (s.get("unit_digests") or {}).get("api"), unit_digests().get("api")
        if not api_then or api_then != api_now:
            print(f"cfg4: counters were taken on other device code (unit api: {str(api_then)[:12]} then, {str(api_now)[:12]} now): NOT used")
        else:
            pk = [k for k in s.get("per_kernel_counters", []) if "hbm_gbytes" in k]
            lfp = s.get("leapfrogs_by_pass", {})
            lf = [sum(v.values()) if isinstance(v, dict) else v for v in (lfp.get("fetch"), lfp.get("write")) if v]
            if pk and lf:
                useful = sum(lf) / len(lf)
                total = sum(k["hbm_gbytes"] for k in pk) * 1e9
                ep = [k for k in pk if "k_dense_epoch" in k["kernel"]]
                out["configs"]["cfg4"] = {
                    "unit_digest": api_now, "d_vector_bytes": 4096.0, "useful_chain_leapfrogs": useful,
                    "hbm_d_vectors_per_chain_leapfrog_all_kernels": total / useful / 4096.0,
                    "k_dense_epoch_share_of_hbm_bytes": sum(k["hbm_gbytes"] for k in ep) * 1e9 / total if ep else None,
                    "k_dense_epoch_mfma_f64_per_useful_leapfrog": sum(k.get("SQ_INSTS_VALU_MFMA_F64", 0) for k in ep) / useful if ep else None,
                    "command": s.get("command"),
                    "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over the cfg4 run named in `command` (scripts/profile_head.sh cfg4; round 6: the "
                              "bench's own size, PROFILE_STEPS=6 — rounds 3-5 used a 10 + 10-transition run)"}
                c = out["configs"]["cfg4"]
    keep = {k: s.get(k) for k in ("config", "command", "kernel_digest", "kernel_stats", "mode0_launches", "mode3_launches", "counters", "leapfrogs_by_pass", "per_kernel_counters")}
    keep["bench_plain"] = s.get("bench_plain")
    with open(os.path.join(ROOT, "profiles", f"{ROUND}_{cfg}_profile_summary.json"), "w") as f:
        json.dump(keep, f, indent=1)
    with open(os.path.join(ROOT, "profiles", f"{ROUND}_{cfg}_top_kernels.txt"), "w") as f:
        f.write(f"# {s.get('command')}\n# rocprofv3 --kernel-trace --stats, device code {kd[:16]}, after commit {head}\n")
        for k in s.get("kernel_stats", []):
            f.write("%10.1f ms %7d calls %10.1f us avg %5.1f %%  %s\n" % (k["total_us"] / 1e3, k["calls"], k["average_us"], k["percent"], k["name"]))
    print(cfg, json.dumps(c)[:400])
with open(path, "w") as f:
    json.dump(out, f, indent=1)
print("wrote", path)
