"""Static check of the Julia `ccall` layer against the C header (no Julia in this image: the shim cannot be executed here, so
what CAN be checked is checked on every CPU run): every `ccall((:name, LIB), Cint, (argtypes…), args…)` in
julia/AdvancedHMCMI355XExt.jl names a function include/ahmc_hip.h declares, passes as many argument types as the C prototype has
parameters and as many values as types, and each Julia type is one the C parameter's type admits (Cint ↔ int32_t, Int64 ↔
int64_t, Cdouble ↔ double, UInt64 ↔ uint64_t, Ptr{…} / Ref / Cstring ↔ pointers).  The reference's hook this shim mirrors:
ext/AdvancedHMCCUDAExt.jl:6-34."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def c_prototypes():
    src = open(os.path.join(ROOT, "include", "ahmc_hip.h"), encoding="utf-8").read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int32_t|void\*|const char\*)\s+(ahmc_[a-z_0-9]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, params = m.group(1), m.group(2), " ".join(m.group(3).split())
        plist = [] if params in ("", "void") else [p.strip() for p in params.split(",")]
        protos[name] = (ret, plist)
    return protos


def split_top(s):
    """split on commas that are not inside brackets"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def julia_ccalls():
    src = open(os.path.join(ROOT, "julia", "AdvancedHMCMI355XExt.jl"), encoding="utf-8").read()
    src = re.sub(r"#[^\n]*", "", src)
    calls = []
    for m in re.finditer(r"ccall\(\(:(ahmc_[a-z_0-9]+), LIB\),\s*(\w+),\s*\(", src):
        name, ret = m.group(1), m.group(2)
        i, depth = m.end(), 1                                  # the argument-type tuple
        while depth:
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        types = split_top(src[m.end():i - 1])
        j, depth = i, 1                                       # the rest of the call: `, args…)`
        while depth:
            depth += {"(": 1, ")": -1}.get(src[j], 0)
            j += 1
        args = split_top(src[i:j - 1].lstrip(","))
        calls.append((name, ret, types, args))
    return calls


def admits(ctype, jtype):
    c = ctype.replace("const ", "").strip()
    base = re.sub(r"\s+\w+$", "", c) if not c.endswith("*") else c   # drop the parameter name
    base = base.strip()
    ptr = "*" in c
    if ptr:
        return jtype.startswith(("Ptr{", "Ref{")) or jtype in ("Cstring",)
    table = {"int32_t": {"Cint", "Int32"}, "int64_t": {"Int64", "Clonglong"}, "uint64_t": {"UInt64"}, "double": {"Cdouble", "Float64"}}
    return jtype in table.get(base, set())


def test_every_ccall_matches_the_header():
    protos = c_prototypes()
    calls = julia_ccalls()
    assert len(protos) >= 56 and len(calls) >= 50, (len(protos), len(calls))
    for name, ret, types, args in calls:
        assert name in protos, f"ccall of {name}: not declared in include/ahmc_hip.h"
        cret, params = protos[name]
        assert len(types) == len(params), f"{name}: {len(types)} Julia argument types, {len(params)} C parameters"
        assert len(args) == len(types), f"{name}: {len(args)} values for {len(types)} argument types"
        assert ret == {"int32_t": "Cint", "const char*": "Cstring", "void*": "Ptr"}[cret] or (cret == "void*" and ret.startswith("Ptr")), (name, ret, cret)
        for k, (jt, cp) in enumerate(zip(types, params)):
            assert admits(cp, jt), f"{name}: argument {k + 1}: Julia {jt} for C `{cp}`"


def test_struct_layouts_match():
    """AdaptorState in the shim has the fields of ahmc_adaptor_state, in order, with matching widths"""
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "ahmc_hip.h"), encoding="utf-8").read(), flags=re.S)
    body = re.search(r"typedef struct \{(.*?)\} ahmc_adaptor_state;", hdr, flags=re.S).group(1)
    cfields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        ty, names = decl.split(None, 1)
        cfields += [(n.strip(), ty) for n in names.split(",")]
    jl = re.sub(r"#[^\n]*", "", open(os.path.join(ROOT, "julia", "AdvancedHMCMI355XExt.jl"), encoding="utf-8").read())
    jbody = re.search(r"struct AdaptorState\n(.*?)\nend", jl, flags=re.S).group(1)
    jfields = [tuple(f.strip().split("::")) for line in jbody.splitlines() for f in line.split(";") if "::" in f]
    width = {"int32_t": "Cint", "double": "Cdouble", "int64_t": "Int64"}
    assert [(n, width[t]) for n, t in cfields] == jfields, (cfields, jfields)
