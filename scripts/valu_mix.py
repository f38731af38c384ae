#!/usr/bin/env python
"""Mix-weighted VALU-issue roof of one kernel (no GPU needed once the two inputs exist):

    python scripts/valu_mix.py KERNEL.s profiles/r3_valu_rate.json [counters.json mode0] [--waves 4]

Inputs
  KERNEL.s         the kernel's slice of `llvm-objdump -d --no-show-raw-insn` (see scripts/README.md)
  valu_rate.json   measured issue rate of every instruction class on THIS chip (scripts/probe/valu_rate.hip)
  counters.json    optional: scripts/profile_counters.py summary — its `valu_mix_per_leapfrog` (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64,
                   _INT32, _INT64, _CVT per leapfrog) are DYNAMIC counts; without it everything is static

Method.  The chip does not issue "a VALU instruction" at one rate (probe: 32-bit add / xor / mov ≈ 1 000 G wave-instr/s, f64
arithmetic, DPP moves, 64-bit moves, multiplies ≈ 450–590, permlane swaps ≈ 300), so the roof of a kernel is the harmonic mean
of the class rates weighted by its instruction mix:   peak_mix = N / Σ_c n_c / rate_c.
The hot region is the innermost natural loop that holds the NUTS leaf (signature: v_ldexp_f64 — the leaf weight) AND a
Philox draw (v_mul_hi_u32 — the merges): the `for leaf` loop of k_nuts; every instruction in it is classified by mnemonic.
With counters, the dynamic per-leapfrog counts replace the static ones for the classes the hardware counts (f64 add / mul /
fma / transcendental, int32, int64, cvt); the static shares split (a) INT32 into 2-cycle (add/xor/shift/and/or) and
multiply classes and (b) the uncounted remainder (moves, DPP, permlane, selects, compares, readlanes) — the remainder's
dynamic total is SQ_INSTS_VALU minus the counted classes.
"""
import json
import re
import sys
from collections import defaultdict


# mnemonic -> (class, probe entry that prices it)
RULES = [
    (r"v_fma_f64|v_fmac_f64", "fma_f64", "v_fma_f64"),
    (r"v_add_f64", "add_f64", "v_add_f64"),
    (r"v_mul_f64", "mul_f64", "v_mul_f64"),
    (r"v_(max|min)_f64", "other_f64", "v_max_f64"),
    (r"v_ldexp_f64|v_frexp|v_fract_f64|v_trunc_f64|v_floor_f64|v_ceil_f64", "other_f64", "v_ldexp_f64"),
    (r"v_rndne_f64", "other_f64", "v_rndne_f64"),
    (r"v_(rcp|rsq|sqrt)_f64", "trans_f64", "v_rcp_f64"),
    (r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_f32", "trans_f32", "v_rcp_f64"),
    (r"v_cmp\w*_f64|v_cmp_class_f64", "cmp_f64", "v_cmp_lt_f64"),
    (r"v_cmp", "cmp_32", "v_add_co_u32"),
    (r"v_cvt_\w*f64|v_cvt_f64", "cvt", "v_cvt_i32_f64"),
    (r"v_cvt", "cvt", "v_cvt_i32_f64"),
    (r"v_permlane16_swap", "permlane", "v_permlane16_swap_b32"),
    (r"v_permlane32_swap", "permlane", "v_permlane32_swap_b32"),
    (r"v_readfirstlane|v_readlane|v_writelane", "readlane", "v_readfirstlane_b32"),
    (r"v_mul_hi_u32|v_mul_hi_i32", "int32_mul", "v_mul_hi_u32"),
    (r"v_mul_lo_u32|v_mul_u32_u24|v_mul_i32_i24|v_mad_u32_u24|v_mad_i32_i24", "int32_mul", "v_mul_lo_u32"),
    (r"v_mad_u64_u32|v_mad_i64_i32", "int64", "v_mad_u64_u32"),
    (r"v_lshl_add_u64", "int64", "v_lshl_add_u64"),
    (r"v_lshlrev_b64|v_lshrrev_b64|v_ashrrev_i64", "int64", "v_lshlrev_b64"),
    (r"v_div_(scale|fmas|fixup)_f64", "other_f64", "v_fma_f64"),
    (r"v_(add|sub|subrev)_co_u32|v_addc_co_u32|v_subb_co_u32|v_subbrev_co_u32", "int32_carry", "v_add_co_u32"),
    (r"v_lshlrev_b32|v_lshrrev_b32|v_ashrrev_i32|v_lshl_add_u32|v_lshl_or_b32|v_bfe|v_bfi|v_alignbit|v_perm_b32|v_add3|v_and_or|v_or3|v_xad|v_xor3|v_add_lshl",
     "int32_shift", "v_lshlrev_b32"),
    (r"v_(add|sub|subrev)_u32|v_(add|sub)_i32|v_(xor|and|or|not)_b32|v_bitop3_b32|v_(min|max)_[ui]32", "int32_simple", "v_add_u32"),
    (r"v_(add|sub|mul|fma|fmac|max|min|mac)_f32", "f32", "v_fma_f32"),
    (r"v_pk_", "pk_f32", "v_pk_fma_f32"),
    (r"v_cndmask_b32", "cndmask", "v_cndmask_b32"),
    (r"v_mov_b64|v_accvgpr", "mov_b64", "v_mov_b64"),
    (r"v_mov_b32|v_swap_b32|v_nop", "mov_b32", "v_mov_b32"),
]


def classify(mn, ops):
    if mn.startswith("v_mov_b32") and ("dpp" in mn or "quad_perm" in ops or "row_" in ops or "bank_mask" in ops):
        if "quad_perm" in ops:
            return "dpp", "v_mov_b32_dpp quad_perm"
        if "row_mirror" in ops or "row_half_mirror" in ops:
            return "dpp", "v_mov_b32_dpp row_mirror"
        if "row_newbcast" in ops or "row_bcast" in ops:
            return "dpp", "v_mov_b32_dpp row_newbcast:0"
        return "dpp", "v_mov_b32_dpp row_ror:4"
    if ("quad_perm" in ops or "row_" in ops) and mn.startswith("v_"):   # arithmetic with a DPP operand
        return "dpp", "v_mov_b32_dpp row_ror:4"
    for rx, cls, probe in RULES:
        if re.match(rx, mn):
            return cls, probe
    return "unclassified", "v_mov_b64"   # priced at the 4-cycle class



def analyse(text, rates, mix=None, total=None, LEVEL=0, W="W4"):
    """text: the kernel's disassembly; rates: {probe entry: G wave-instr/s at W waves per SIMD}; mix / total: the dynamic
    per-leapfrog class counts (SQ_INSTS_VALU_* of scripts/profile_counters.py) and SQ_INSTS_VALU per leapfrog;
    LEVEL: 0 = the innermost loop holding leaf + draw, 1 = the loop around it, ..."""
    # ---- parse, basic blocks, natural loops (as scripts/isa_blocks.py) ----
    ins, base = [], None
    for line in text.splitlines():
        m = re.match(r"^([0-9a-f]+) <.*>:$", line.strip())
        if m:
            base = int(m.group(1), 16)
            continue
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if not m:
            continue
        mn, ops, addr = m.group(1), m.group(2), int(m.group(3), 16)
        tgt = None
        if mn.startswith(("s_branch", "s_cbranch")):
            t = re.search(r"\+0x([0-9a-f]+)>", line)
            tgt = base + int(t.group(1), 16) if t else None
        ins.append((addr, mn, ops, tgt))
    addrs = [a for a, *_ in ins]
    idx = {a: i for i, a in enumerate(addrs)}
    leaders = {addrs[0]}
    for i, (a, mn, ops, tgt) in enumerate(ins):
        if tgt is not None:
            leaders.add(tgt)
            if i + 1 < len(ins):
                leaders.add(addrs[i + 1])
    leaders = sorted(x for x in leaders if x in idx)
    blocks = [ins[idx[s]:(idx[leaders[k + 1]] if k + 1 < len(leaders) else len(ins))] for k, s in enumerate(leaders)]
    bidx = {s: k for k, s in enumerate(leaders)}
    succ = defaultdict(list)
    for k, body in enumerate(blocks):
        last = body[-1]
        if last[3] is not None:
            succ[k].append(bidx.get(last[3], -1))
            if not last[1].startswith("s_branch") and k + 1 < len(blocks):
                succ[k].append(k + 1)
        elif last[1] not in ("s_endpgm", "s_setpc_b64") and k + 1 < len(blocks):
            succ[k].append(k + 1)
    pred = defaultdict(list)
    for a in succ:
        for b in succ[a]:
            pred[b].append(a)
    loops = []
    for k in list(succ):
        for t in succ[k]:
            if 0 <= t <= k:
                body, work = {t, k}, [k]
                while work:
                    x = work.pop()
                    if x == t:
                        continue
                    for p_ in pred[x]:
                        if p_ not in body and t <= p_ <= max(k, x):
                            body.add(p_)
                            work.append(p_)
                loops.append(body)


    def has(body, pat):
        return any(pat in i[1] for b in body for i in blocks[b])


    # the leaf loop holds the leaf weight's exp (v_ldexp_f64), the tree draws (Philox: v_mul_hi_u32) AND the cross-lane reductions (DPP moves);
    # round 5: without the last condition the smallest such "loop" of the restructured kernels was a fragment with no reduction in it
    cands = sorted((l for l in loops if has(l, "v_ldexp_f64") and has(l, "v_mul_hi_u32") and has(l, "_dpp")), key=len)
    if not cands:
        cands = sorted((l for l in loops if has(l, "v_ldexp_f64") and has(l, "v_mul_hi_u32")), key=len)
    if not cands:
        cands = sorted((l for l in loops if has(l, "v_ldexp_f64") or has(l, "v_exp")), key=len) or [set(range(len(blocks)))]
    hot = cands[min(LEVEL, len(cands) - 1)]
    static = defaultdict(float)
    probe_of = {}
    per_probe_static = defaultdict(float)
    for b in hot:
        for a, mn, ops, tgt in blocks[b]:
            if not mn.startswith("v_"):
                continue
            cls, probe = classify(mn, ops)
            static[cls] += 1
            per_probe_static[(cls, probe)] += 1
    n_static = sum(static.values())
    out = {"waves_per_simd_priced_at": W, "hot_loop": {"blocks": len(hot), "static_valu": n_static, "nesting_level_above_innermost": LEVEL,
                                                                            "candidates_static_valu": [sum(1 for b in l for i in blocks[b] if i[1].startswith("v_")) for l in cands]},
           "static_class_counts": dict(sorted(static.items(), key=lambda kv: -kv[1]))}


    def harmonic(weights):
        """weights: {(cls, probe): n} -> N / Σ n / rate"""
        N = sum(weights.values())
        t = sum(n / rates.get(probe, rates["v_mov_b64"]) for (cls, probe), n in weights.items())  # (a class the probe file lacks: the 4-cycle rate)
        return N / t if t else None


    out["peak_mix_static_gwave_instr_per_s"] = harmonic(per_probe_static)
    if mix is not None:
        dyn = defaultdict(float)

        def spread(classes, n_dyn):
            """n_dyn dynamic instructions over the (cls, probe) entries of `classes` in their static proportions"""
            keys = [k for k in per_probe_static if k[0] in classes]
            tot = sum(per_probe_static[k] for k in keys)
            if tot == 0:
                if keys or n_dyn <= 0:
                    return
                dyn[(classes[0], {"int32_simple": "v_add_u32"}.get(classes[0], "v_mov_b64"))] += n_dyn
                return
            for k in keys:
                dyn[k] += n_dyn * per_probe_static[k] / tot

        spread(["add_f64"], mix["add_f64"])
        spread(["mul_f64"], mix["mul_f64"])
        spread(["fma_f64"], mix["fma_f64"])
        spread(["trans_f64"], mix["trans_f64"]) if static.get("trans_f64") else dyn.__setitem__(("trans_f64", "v_rcp_f64"), mix["trans_f64"])
        spread(["int32_simple", "int32_shift", "int32_mul", "int32_carry", "cmp_32"], mix["int32"])
        spread(["int64"], mix["int64"])
        spread(["cvt"], mix["cvt"])
        counted = mix["add_f64"] + mix["mul_f64"] + mix["fma_f64"] + mix["trans_f64"] + mix["int32"] + mix["int64"] + mix["cvt"]
        rest_classes = [cl for cl in static if cl not in ("add_f64", "mul_f64", "fma_f64", "trans_f64", "int32_simple", "int32_shift", "int32_mul",
                                                          "int32_carry", "cmp_32", "int64", "cvt")]
        spread(rest_classes, total - counted)
        by_cls = defaultdict(float)
        for (cls, probe), n in dyn.items():
            by_cls[cls] += n
        out["dynamic"] = {"valu_per_leapfrog": total, "counted_by_hardware_classes": counted, "remainder_split_by_static_shares": total - counted,
                          "class_counts_per_leapfrog": dict(sorted(by_cls.items(), key=lambda kv: -kv[1])),
                          "issue_time_share": {cls: sum(n / rates.get(p, rates["v_mov_b64"]) for (c2, p), n in dyn.items() if c2 == cls) / sum(n / rates.get(p, rates["v_mov_b64"]) for (c2, p), n in dyn.items())
                                               for cls in by_cls}}
        out["peak_mix_gwave_instr_per_s"] = harmonic(dyn)
        out["uniform_4_cycle_peak_for_comparison"] = 1024 * 2.4 / 4
        # The same mix priced at NOMINAL issue cycles: the probe tells which class an instruction type belongs to — its measured
        # rate snapped to the nearest of 2 / 4 / 8 / 16 cycles per wave64 instruction at 2.4 GHz — and the roof is those cycles at
        # the nominal clock.  Single-class microbenchmarks run a few per cent under nominal (a pure v_fma_f64 stream 27 %: power),
        # so a kernel with a mixed stream can EXCEED a roof made from the measured single-class rates; it cannot exceed this one.
        import math

        def nominal_rate(r):
            return 1024 * 2.4 / (2 ** max(1, min(4, round(math.log2(1024 * 2.4 / r)))))

        cyc = {k: nominal_rate(v) for k, v in rates.items()}
        n_all = sum(dyn.values())
        t_all = sum(n / cyc.get(pr, 1024 * 2.4 / 4) for (c2, pr), n in dyn.items())
        out["peak_mix_nominal_gwave_instr_per_s"] = n_all / t_all
        out["nominal_cycles_by_probe_entry"] = {k: round(1024 * 2.4 / v) for k, v in cyc.items()}
    return out


def load_rates(path, W="W4"):
    """per-class issue rates; W = "max": the highest sustained rate of the class over the probed occupancies — what a ROOF is
    (an upper bound: the kernel runs at 3.9 waves per SIMD, and several classes still gain a few per cent from 4 to 8 waves)"""
    cl = json.load(open(path))["classes"]
    if W == "max":
        return {k: max(x["gwave_instr_per_s_chip"] for x in v.values()) for k, v in cl.items()}
    return {k: v[W]["gwave_instr_per_s_chip"] for k, v in cl.items()}


if __name__ == "__main__":
    _argv = sys.argv[1:]
    args = [a for i, a in enumerate(_argv) if not a.startswith("--") and not (i > 0 and _argv[i - 1] in ("--waves", "--loop"))]
    W = ("max" if _argv[_argv.index("--waves") + 1] == "max" else "W" + _argv[_argv.index("--waves") + 1]) if "--waves" in _argv else "max"
    LEVEL = int(_argv[_argv.index("--loop") + 1]) if "--loop" in _argv else 0
    mix = total = None
    if len(args) >= 4:
        cj = json.load(open(args[2]))
        c = cj["counters"][args[3]]
        mix, total = c["valu_mix_per_leapfrog"], c["valu_per_leapfrog"]
    print(json.dumps(analyse(open(args[0]).read(), load_rates(args[1], W), mix, total, LEVEL, W), indent=1))
