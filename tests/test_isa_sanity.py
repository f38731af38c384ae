"""Static check of the shipped gfx950 code objects (no GPU): no kernel may hold a VGPR spill whose ONLY store sits under a
narrowed exec mask while the slot is reloaded under the full one.

That pattern is a register-allocation artefact, not something the source says: the allocator may place the spill of a
long-lived value inside a short `if (lane == 0) …` block; only the active lanes' copies reach scratch and the later reload
hands every other lane stale memory.  It is what made the (128,8) multi-wave instantiation of k_nuts return wrong candidates
and fault on the MI355X when its leaf used the single-value reduction (the spilled value was the chain index, reloaded to
address the chain's vectors): DESIGN.md §7.3, scripts/isa_masked_spills.py.  The scan takes the disassembly of every unit
of the build that `__graft_entry__.build()` made."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "advancedhmc.jl_amd", "csrc", "build")


@pytest.mark.skipif(not os.path.isdir(OBJ) or not any(f.endswith(".o") for f in os.listdir(OBJ)) or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"),
                    reason="needs the object files of the HIP build and llvm-objdump")
def test_no_spill_is_stored_only_under_a_narrowed_exec_mask():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "isa_masked_spills.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert "inst_f64_t0.o: 0 masked" in r.stdout


def test_scanner_flags_the_known_bad_pattern(tmp_path):
    """a hand-made kernel with the faulty shape (value defined under the full mask, its only spill store inside a
    lane-0 block, reload outside) is reported; the same with a full-mask store of the slot is not"""
    bad = """0000000000001000 <_Z5k_badv>:
\tv_mov_b64_e32 v[170:171], s[14:15]                         // 000000001000: 00000000
\tv_add_f64 v[20:21], v[20:21], v[22:23]                     // 000000001008: 00000000
\ts_and_saveexec_b64 s[28:29], s[4:5]                        // 000000001010: 00000000
\tds_write2_b64 v0, v[20:21], v[22:23] offset1:1             // 000000001014: 00000000
\tscratch_store_dwordx2 off, v[170:171], off offset:252      // 00000000101C: 00000000
\ts_or_b64 exec, exec, s[28:29]                              // 000000001024: 00000000
\ts_barrier                                                  // 000000001028: 00000000
\tscratch_load_dwordx2 v[170:171], off, off offset:252       // 00000000102C: 00000000
\ts_endpgm                                                   // 000000001034: 00000000
"""
    good = bad.replace("\ts_and_saveexec_b64 s[28:29], s[4:5] ", "\tscratch_store_dwordx2 off, v[170:171], off offset:252      // 00000000100C: 00000000\n\ts_and_saveexec_b64 s[28:29], s[4:5] ")
    fb, fg = tmp_path / "bad.s", tmp_path / "good.s"
    fb.write_text(bad)
    fg.write_text(good)
    script = os.path.join(ROOT, "scripts", "isa_masked_spills.py")
    rb = subprocess.run([sys.executable, script, str(fb)], capture_output=True, text=True)
    rg = subprocess.run([sys.executable, script, str(fg)], capture_output=True, text=True)
    assert rb.returncode == 1 and "offset:252" in rb.stdout, rb.stdout + rb.stderr
    assert rg.returncode == 0, rg.stdout + rg.stderr
