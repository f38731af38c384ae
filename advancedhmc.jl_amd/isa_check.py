#!/usr/bin/env python
"""Find VGPR spill stores that the compiler placed under a NARROWED exec mask (between an `s_and_saveexec` and the
`s_or_b64 exec, exec, …` that restores it) whose slot is reloaded elsewhere: only the active lanes' copies reach scratch,
a later full-wave reload hands the other lanes stale memory.  This is what made the (128,8) multi-wave instantiation of
k_nuts return wrong candidates / fault when its leaf used the single-value reduction: the lane-0-only block
of the cross-wave exchange (`if ((threadIdx.x & 63) == 0) b[w] = v;`) received the spill of a wave-uniform 64-bit index.

`build.py` runs `check_object` on every translation unit it compiles and FAILS THE BUILD when one has the pattern (the
disassembly goes through a temporary directory outside the repository and is deleted).  By hand:

    python -m ahmc_amd.isa_check                 # every unit in the object cache of the build
    python advancedhmc.jl_amd/isa_check.py file.s ...  # disassembly files (llvm-objdump -d --no-show-raw-insn)

Exit status 1 if any kernel has such a store."""
import os
import re
import subprocess
import sys

import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def available() -> bool:
    return os.path.exists(f"{LLVM}/llvm-objdump")


def check_object(obj, label=None, quiet=False) -> int:
    """number of masked-spill findings in one object file (disassembled in a throw-away directory under $TMPDIR)"""
    return analyse_object(obj, label, quiet)[0]


def code_digest(text) -> str:
    """sha256 over the gfx950 INSTRUCTIONS of a disassembly (addresses, encodings and comments stripped): equal digests =
    the same device code, whatever was edited in comments, host code or unrelated templates, and whichever compiler made it"""
    import hashlib

    h = hashlib.sha256()
    for line in text.splitlines():
        t = line.strip()
        if not t or t == "..." or "file format" in t or t.startswith("Disassembly of section"):
            continue  # (the "file format" line carries the path of the temporary copy that was disassembled)
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", t)
        if m:
            t = m.group(1)            # a kernel's header line: its name, not its address
        else:
            t = re.sub(r"\s*//.*$", "", t)            # encodings / addresses
            t = re.sub(r"\s*<[^>]*\+0x[0-9a-f]+>", "", t)  # a branch target spelled as symbol + offset (the offset operand itself stays)
        h.update(t.encode())
        h.update(b"\n")
    return h.hexdigest()


def analyse_object(obj, label=None, quiet=False):
    """(number of masked-spill findings, digest of the device code) of one object file — one disassembly for both"""
    with tempfile.TemporaryDirectory(prefix="ahmc_isa_") as tmp:
        text = disassemble(obj, tmp)
    return scan(text, label or os.path.basename(obj), quiet=quiet), code_digest(text)


def disassemble(obj, tmp):
    import shutil

    cp = os.path.join(tmp, os.path.basename(obj))
    shutil.copyfile(obj, cp)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", cp], capture_output=True, check=True)
    co = cp + ".0.hipv4-amdgcn-amd-amdhsa--gfx950"
    return subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout


def scan(text, label, quiet=False):
    bad = []
    kernel, lines, addrs, targets, kbase = None, [], [], set(), 0

    def flush():
        if not kernel:
            return
        tset = {kbase + t for t in targets}
        # stores under a narrowed exec: a region opens at s_and(n2)_saveexec and closes at the next instruction that writes exec
        # (linear order; nested regions close their parents too — conservative in the direction of fewer reports).  A store in
        # a region is suspicious when the LAST definition of the stored registers lies BEFORE the region opened, i.e. the value
        # was made under the wider mask and only the active lanes' copies reach the slot, and the slot is reloaded outside.
        masked = []
        region = None
        loads = {}
        for i, t in enumerate(lines):
            m = re.search(r"scratch_load_\w+ .*offset:(\d+)", t)
            if m:
                loads.setdefault(int(m.group(1)), []).append((i, region is not None))
            if re.match(r"s_and(n2)?_saveexec_b64", t):
                region = i
                continue
            if region is not None and re.match(r"(s_or_b64|s_mov_b64|s_xor_b64|s_andn2_b64|s_and_b64|s_or_saveexec_b64) exec", t.replace("s_or_saveexec_b64 ", "s_or_saveexec_b64 exec ")):
                region = None
                continue
            m = re.search(r"scratch_store_\w+ off, v\[?(\d+)(?::(\d+))?\]?, .*offset:(\d+)", t)
            if m and region is not None:
                lo, hi = int(m.group(1)), int(m.group(2) or m.group(1))
                masked.append((i, region, lo, hi, int(m.group(3)), t.split("//")[0].strip()))

        def writes(t, lo, hi):
            mm = re.match(r"(v_\w+|scratch_load_\w+|global_load_\w+|flat_load_\w+|ds_read\w*|buffer_load_\w+)\s+v\[?(\d+)(?::(\d+))?\]?", t)
            if not mm or mm.group(1).startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
                return False
            a, b = int(mm.group(2)), int(mm.group(3) or mm.group(2))
            if a <= hi and b >= lo:
                return True
            if "permlane" in mm.group(1):  # the swaps write their second operand too
                m2 = re.search(r",\s*v(\d+)", t)
                return bool(m2) and lo <= int(m2.group(1)) <= hi
            return False

        # every store per slot: a slot that ALSO receives a store under the full mask is the compiler's way of merging two
        # paths through memory (store the bypass value for all lanes, overwrite the active lanes inside the region) — legal;
        # the dangerous case is a slot whose ONLY stores are masked ones of values made under the wider mask
        all_stores = {}
        reg = None
        for i, t in enumerate(lines):
            if re.match(r"s_and(n2)?_saveexec_b64", t):
                reg = i
            elif reg is not None and re.match(r"(s_or_b64|s_mov_b64|s_xor_b64|s_andn2_b64|s_and_b64) exec", t):
                reg = None
            m = re.search(r"scratch_store_\w+ .*offset:(\d+)", t)
            if m:
                all_stores.setdefault(int(m.group(1)), []).append(reg is not None)
        for i, r, lo, hi, off, t in masked:
            d = next((j for j in range(i - 1, -1, -1) if writes(lines[j], lo, hi)), -1)
            outside = [j for j, inside in loads.get(off, []) if not inside]
            # only straight-line regions are judged (linear order says nothing across loops): no branch between the saveexec
            # and the store other than the skip branch right behind the saveexec, and no branch target inside
            straight = all(not re.match(r"s_c?branch", lines[j]) for j in range(r + 2, i)) and not any(addrs[j] in tset for j in range(r + 1, i + 1))
            if d < r and outside and all(all_stores.get(off, [True])) and straight:
                bad.append((kernel, i + 1, t + f"   [value defined {i - d} instructions earlier, region opened {i - r} earlier]", len(outside)))

    def flush_nested():
        """Second pass (round 4), for what the straight-line pass cannot see: regions NESTED inside the narrowed one (a store
        behind an inner `s_and_saveexec … s_or_b64 exec` pair is still under the outer mask) and regions with branches in them.
        Regions are taken from their skip branches — `s_and(n2)_saveexec_b64 D, S ; s_cbranch_execz L` narrows exec for the
        ADDRESS RANGE up to L, whatever is nested or branches inside it — so no stack has to survive the control flow.  A spill
        SLOT is reported when every store to it in the whole kernel lies inside such a range, one of those stores saves a value
        that was made BEFORE its range opened (all lanes hold it, only the active ones reach the slot), and a reload of the slot
        lies outside the ranges the stores are in.  Found on the MI355X as a memory fault of k_nuts<double,8,2,3,1>: the spills
        of the chain index and of the lane's first dimension landed at the end of the `if (on && kt > 0)` block of the prologue —
        at kt = 0 no lane enters it and the epilogue reloads uninitialised scratch."""
        if not kernel:
            return
        regions = []          # (index of the saveexec, first index after the range)
        idx_of = {a: i for i, a in enumerate(addrs) if a >= 0}
        for i, t in enumerate(lines[:-1]):
            if re.match(r"s_and(n2)?_saveexec_b64", t) and re.match(r"s_cbranch_execz", lines[i + 1]):
                mt = re.search(r"<[^>]*\+0x([0-9a-f]+)>", lines[i + 1])
                if mt and (kbase + int(mt.group(1), 16)) in idx_of:
                    j = idx_of[kbase + int(mt.group(1), 16)]
                    # (the branch lands on or before the `s_or_b64 exec, exec, D` that restores the mask: spills the allocator
                    # puts between the two still run under the narrowed — or empty — mask)
                    dreg = re.match(r"s_and(?:n2)?_saveexec_b64 (\S+),", t).group(1)
                    k = next((q for q in range(j, min(j + 40, len(lines))) if re.match(r"s_or_b64 exec, exec, " + re.escape(dreg), lines[q])), j)
                    if k > i:
                        regions.append((i, k))

        def inside(i):
            return tuple(r for r in regions if r[0] < i < r[1])

        def writes(t, lo, hi):
            mm = re.match(r"(v_\w+|scratch_load_\w+|global_load_\w+|flat_load_\w+|ds_read\w*|buffer_load_\w+)\s+v\[?(\d+)(?::(\d+))?\]?", t)
            if not mm or mm.group(1).startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
                return False
            a, b = int(mm.group(2)), int(mm.group(3) or mm.group(2))
            return a <= hi and b >= lo

        stores, loads = {}, {}
        for i, t in enumerate(lines):
            op = t.split("//")[0].strip()
            m = re.match(r"scratch_store_\w+ off, v\[?(\d+)(?::(\d+))?\]?, off(?: offset:(\d+))?$", op)
            if m:
                stores.setdefault(int(m.group(3) or 0), []).append((i, inside(i), int(m.group(1)), int(m.group(2) or m.group(1))))
                continue
            m = re.match(r"scratch_load_\w+ v\S+, off, off(?: offset:(\d+))?$", op)
            if m:
                loads.setdefault(int(m.group(1) or 0), []).append((i, inside(i)))
        def innermost(ins):
            return min(ins, key=lambda r: r[1] - r[0]) if ins else None

        for off, st in stores.items():
            # a reload is in danger when NO store of the slot runs under a mask that covers the reload's lanes: every store's
            # innermost narrowed range must exclude the reload (a store outside all ranges covers everything)
            inner = [innermost(ins) for _, ins, _, _ in st]
            if any(r is None for r in inner):
                continue
            out = [i for i, _ in loads.get(off, []) if all(not (r[0] < i < r[1]) for r in inner)]
            if not out:
                continue
            live_in = []
            for (i, ins, lo, hi), r in zip(st, inner):
                d = next((j for j in range(i - 1, -1, -1) if writes(lines[j], lo, hi)), -1)
                if d < r[0]:
                    live_in.append(i)
            if not live_in and not os.environ.get("AHMC_ISA_STRICT"):
                continue
            i0 = (live_in or [st[0][0]])[0]
            bad.append((kernel, i0 + 1, lines[i0].split("//")[0].strip() + f"   [slot {off}: none of its {len(st)} store(s) runs under a mask that covers the reload(s); nested regions followed]", len(out)))

    for raw in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", raw.strip())
        if m:
            flush()
            flush_nested()
            kernel, lines, addrs, targets = m.group(1), [], [], set()
            kbase = int(raw.strip().split()[0], 16)
            continue
        t = raw.strip()
        if t:
            lines.append(t)
            ma = re.search(r"//\s*([0-9A-Fa-f]+):", t)
            addrs.append(int(ma.group(1), 16) if ma else -1)
            mt = re.search(r"^s_c?branch\S*\s.*<[^>]*\+0x([0-9a-f]+)>", t)
            if mt:
                targets.add(int(mt.group(1), 16))
    flush()
    flush_nested()
    names = subprocess.run(["c++filt"], input="\n".join(b[0] for b in bad), capture_output=True, text=True).stdout.splitlines() if bad else []
    for (k, ln, t, n), nm in zip(bad, names):
        if quiet:
            continue
        print(f"{label}: {nm[:70]}: line {ln}: {t}  (slot reloaded {n}x outside a masked region)")
    return len(bad)


def main():
    n = 0
    if len(sys.argv) > 1:
        for f in sys.argv[1:]:
            n += scan(open(f).read(), os.path.basename(f))
    else:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import build as B  # the object cache lives outside the repository

        obj = B.OBJ
        for f in sorted(os.listdir(obj)):
            if f.endswith(".o"):
                k = check_object(os.path.join(obj, f), f)
                print(f"{f}: {k} masked spill store(s) with outside reloads")
                n += k
    sys.exit(1 if n else 0)


if __name__ == "__main__":
    main()
