"""Build libranenv_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).

Five objects, compiled in parallel, then linked:
  ranenv_step.hip x 3   the builds of the step kernel for row widths NP = 8, 10, 16 (-DRANENV_NP=...)
  ranenv_aux.hip        the small kernels (class sort, sidecars, re-tiling, ingest, heads, episode advance, traffic examination)
  ranenv_host.cpp       the host side of the C ABI

    python -m intent_radio_sched_multi_slice_amd.csrc.build [--force] [-o other.so] [-DFLAG ...]    # extra -D flags: diagnostic variants
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
HDR = os.path.join(REPO, "include", "ranenv.h")
OUT = os.path.join(HERE, "libranenv_hip.so")
SOURCES = ("ranenv_step.hip", "ranenv_aux.hip", "ranenv_host.cpp", "ranenv_step_body.hpp", "ranenv_numeric.hpp", "ranenv_internal.h")
# (object name, source, extra flags)
UNITS = (("step_np10", "ranenv_step.hip", ("-DRANENV_NP=10",)), ("step_np8", "ranenv_step.hip", ("-DRANENV_NP=8",)),
         ("step_np16", "ranenv_step.hip", ("-DRANENV_NP=16",)), ("aux", "ranenv_aux.hip", ()), ("host", "ranenv_host.cpp", ()))
CFLAGS = ("-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wno-pass-failed",
          "-ffp-contract=off",     # numpy rounds a*b and +c separately; fma() is written out where it is exact
          "-mllvm", "-amdgpu-atomic-optimizer-strategy=None",   # the kernel's few atomic adds come from one lane each
          # no hoisting at machine level: around the step kernel's per-TTI loop it parks a dozen constants in VGPRs for the
          # whole launch and spills to make room (96 VGPRs are the budget of 5 waves per SIMD); without it: no spills
          "-mllvm", "-disable-machine-licm")


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def is_stale(out: str = OUT) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    paths = [os.path.join(HERE, s) for s in SOURCES] + [HDR, os.path.abspath(__file__)]
    return any(os.path.exists(p) and os.path.getmtime(p) > t for p in paths)


def build(force: bool = False, verbose: bool = False, out: str = OUT, defines=()) -> str:
    if not force and not is_stale(out):
        return out
    hipcc = hipcc_path()
    # (objects of the shipped library stay in-tree beside it; a variant's objects are scratch)
    objdir = os.path.join(HERE, "_obj") if out == OUT else tempfile.mkdtemp(prefix="ranenv_obj_")
    os.makedirs(objdir, exist_ok=True)

    def compile_unit(unit):
        name, src, extra = unit
        obj = os.path.join(objdir, name + ".o")
        cmd = [hipcc, *CFLAGS, *extra, *defines, "-I", os.path.join(REPO, "include"), "-x", "hip", "-c", os.path.join(HERE, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(UNITS), os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_unit, UNITS))
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", *objs, "-o", out + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(out + ".tmp", out)
    if out != OUT:
        shutil.rmtree(objdir, ignore_errors=True)
    return out


if __name__ == "__main__":
    args = sys.argv[1:]
    out = OUT
    if "-o" in args:
        out = os.path.abspath(args[args.index("-o") + 1])
    extra = []
    for a in args:                 # -DNAME=value, and for compiler experiments: -X<flag> passes <flag> on (-X-mllvm -X-some-option)
        if a.startswith("-D"):
            extra.append(a)
        elif a.startswith("-X"):
            extra.append(a[2:])
    print(build(force="--force" in args or out != OUT, verbose=True, out=out, defines=tuple(extra)))
