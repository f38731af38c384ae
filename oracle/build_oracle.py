"""Build oracle/libahmc_oracle.so (CPU checker — test infrastructure only; see ahmc_oracle.cpp)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libahmc_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "ahmc_oracle.cpp")
    hdr = os.path.join(HERE, "..", "include", "ahmc_hip.h")
    stale = (not os.path.exists(SO)) or any(os.path.getmtime(SO) < os.path.getmtime(f) for f in (src, hdr))
    if force or stale:
        res = subprocess.run(["make", "-B", "-C", HERE], capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("oracle build failed:\n" + res.stdout + res.stderr)
    return SO


if __name__ == "__main__":
    print(build(force=True))
