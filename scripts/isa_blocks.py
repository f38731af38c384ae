#!/usr/bin/env python
"""Static instruction census of ONE kernel's gfx950 code, basic block by basic block (no GPU needed):

    python scripts/isa_blocks.py build_tmp/isa/k_nuts_64_2_m0.s [--loops]

Input: the kernel's slice of `llvm-objdump -d --no-show-raw-insn` (see scripts/README.md).  Per block: address range,
VALU / SALU / LDS / VMEM / SMEM / branch counts, a signature of tell-tale instructions (permlane swaps and DPP moves =
cross-lane reductions, v_ldexp / v_rndne = the leaf weight, v_mul_hi_u32 = Philox ...), successors; then the natural
loops (back edges), innermost first, with their per-iteration static counts.  Used to see where the measured
193 VALU per leapfrog of k_nuts<double,64,2,0,0> go without a profiler."""
import re
import sys
from collections import defaultdict

path = sys.argv[1]
ins = []
base = None
for line in open(path):
    m = re.match(r"^([0-9a-f]+) <.*>:$", line.strip())
    if m:
        base = int(m.group(1), 16)
        continue
    m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
    if not m:
        continue
    mn, ops, addr = m.group(1), m.group(2), int(m.group(3), 16)
    tgt = None
    if mn.startswith("s_branch") or mn.startswith("s_cbranch"):
        t = re.search(r"\+0x([0-9a-f]+)>", line)
        tgt = base + int(t.group(1), 16) if t else None
    ins.append((addr, mn, ops, tgt))
if base is None:
    base = ins[0][0]
addrs = [a for a, *_ in ins]
idx = {a: i for i, a in enumerate(addrs)}
leaders = {addrs[0]}
for i, (a, mn, ops, tgt) in enumerate(ins):
    if tgt is not None:
        leaders.add(tgt)
        if i + 1 < len(ins):
            leaders.add(addrs[i + 1])
    if mn in ("s_endpgm", "s_setpc_b64") and i + 1 < len(ins):
        leaders.add(addrs[i + 1])
leaders = sorted(x for x in leaders if x in idx)
blocks = []
for k, s in enumerate(leaders):
    e = idx[leaders[k + 1]] if k + 1 < len(leaders) else len(ins)
    blocks.append((s, ins[idx[s]:e]))
bidx = {s: k for k, (s, _) in enumerate(blocks)}


def klass(mn):
    if mn.startswith(("s_branch", "s_cbranch")):
        return "BR"
    if mn.startswith(("s_waitcnt", "s_nop", "s_endpgm", "s_barrier", "s_sleep", "s_setprio", "s_clause")):
        return "MISC"
    if mn.startswith(("s_load", "s_buffer_load", "s_store", "s_dcache")):
        return "SMEM"
    if mn.startswith("s_"):
        return "SALU"
    if mn.startswith("ds_"):
        return "LDS"
    if mn.startswith(("global_", "scratch_", "buffer_", "flat_")):
        return "VMEM"
    if mn.startswith("v_"):
        return "VALU"
    return "MISC"


SIG = [("permlane", "xlane"), ("_dpp", "dpp"), ("v_ldexp", "ldexp"), ("v_rndne", "rndne"), ("v_mul_hi_u32", "philox"), ("v_readfirstlane", "rfl"),
       ("v_cvt", "cvt"), ("v_rcp", "rcp"), ("v_log", "log"), ("v_exp", "exp"), ("v_div", "div"), ("scratch_", "scratch"), ("v_fma_f64", "fma64"),
       ("v_add_f64", "add64"), ("v_mul_f64", "mul64"), ("v_cndmask", "cnd"), ("v_cmp", "cmp")]
succ = defaultdict(list)
stats = []
for k, (s, body) in enumerate(blocks):
    c = defaultdict(int)
    sig = defaultdict(int)
    for a, mn, ops, tgt in body:
        c[klass(mn)] += 1
        for pat, name in SIG:
            if pat in mn or (pat == "_dpp" and ("dpp" in ops or "quad_perm" in ops or "row_" in ops)):
                sig[name] += 1
    last = body[-1]
    if last[3] is not None:
        succ[k].append(bidx.get(last[3], -1))
        if not last[1].startswith("s_branch") and k + 1 < len(blocks):
            succ[k].append(k + 1)
    elif last[1] not in ("s_endpgm", "s_setpc_b64") and k + 1 < len(blocks):
        succ[k].append(k + 1)
    stats.append((c, sig))
tot = defaultdict(int)
for c, _ in stats:
    for kk, v in c.items():
        tot[kk] += v
print("# %s: %d instructions in %d blocks; static totals %s" % (path, len(ins), len(blocks), dict(tot)))
back = [(k, t) for k in succ for t in succ[k] if 0 <= t <= k]
# natural loop of a back edge k -> t (assuming reducible, blocks laid out in order): blocks t..k that can reach k
loops = []
for k, t in back:
    body = {t, k}
    work = [k]
    pred = defaultdict(list)
    for a in succ:
        for b in succ[a]:
            pred[b].append(a)
    while work:
        x = work.pop()
        if x == t:
            continue
        for p_ in pred[x]:
            if p_ not in body and t <= p_ <= max(k, x):
                body.add(p_)
                work.append(p_)
    loops.append((t, k, body))
loops.sort(key=lambda l: len(l[2]))
show_blocks = "--loops" not in sys.argv
if show_blocks:
    for k, (s, body) in enumerate(blocks):
        c, sig = stats[k]
        print("B%-4d +0x%05x n=%-4d V=%-3d S=%-3d L=%-2d M=%-2d sm=%-2d  -> %-12s %s" % (k, s - base, len(body), c["VALU"], c["SALU"], c["LDS"], c["VMEM"], c["SMEM"],
              ",".join("B%d" % x for x in succ[k]), " ".join("%s:%d" % kv for kv in sorted(sig.items()) if kv[0] not in ("cnd", "cmp"))))
print("# loops (head..tail: blocks, static per-iteration counts if every block ran once)")
for t, k, body in loops:
    c = defaultdict(int)
    for b in body:
        for kk, v in stats[b][0].items():
            c[kk] += v
    print("loop B%d..B%d: %d blocks  V=%d S=%d L=%d M=%d sm=%d BR=%d" % (t, k, len(body), c["VALU"], c["SALU"], c["LDS"], c["VMEM"], c["SMEM"], c["BR"]))
