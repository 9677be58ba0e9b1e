"""Build libranenv_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
SRC = os.path.join(HERE, "ranenv.hip")
HDR = os.path.join(REPO, "include", "ranenv.h")
OUT = os.path.join(HERE, "libranenv_hip.so")


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def is_stale() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.exists(p) and os.path.getmtime(p) > t for p in (SRC, HDR))


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return OUT
    cmd = [hipcc_path(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-Wno-pass-failed",
           "-ffp-contract=off",     # numpy rounds a*b and +c separately; fma() is written out where it is exact
           "-mllvm", "-amdgpu-atomic-optimizer-strategy=None",   # the kernel's few atomic adds come from one lane each
           # no hoisting at machine level: around the step kernel's per-TTI loop it parks a dozen constants in VGPRs for the
           # whole launch and spills to make room (96 VGPRs are the budget of 5 waves per SIMD); without it: no spills
           "-mllvm", "-disable-machine-licm",
           "-I", os.path.join(REPO, "include"), SRC, "-o", OUT + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(OUT + ".tmp", OUT)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
