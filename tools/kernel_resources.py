"""Per-kernel resources of the SHIPPED library (csrc/libranenv_hip.so): registers, spills, scratch, LDS.

    python tools/kernel_resources.py            # table
    python tools/kernel_resources.py --json     # list of dicts

Reads the gfx950 code object out of the library's offload bundle and its AMDGPU metadata note
(llvm-readelf --notes): what the loader will really run, not what some other compile would give.
tests/test_kernel_resources.py asserts on it (no kernel with scratch; the step kernels within their wave budgets).
"""
from __future__ import annotations

import json
import os
import re
import struct
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(REPO, "intent_radio_sched_multi_slice_amd", "csrc", "libranenv_hip.so")
LLVM_BIN = "/opt/rocm/lib/llvm/bin"
FIELDS = ("private_segment_fixed_size", "group_segment_fixed_size", "sgpr_count", "sgpr_spill_count", "vgpr_count",
          "vgpr_spill_count", "agpr_count", "max_flat_workgroup_size")


def code_objects(so_path: str = SO):
    """The gfx950 ELFs inside the library's __CLANG_OFFLOAD_BUNDLE__s (one bundle per linked object)."""
    so = open(so_path, "rb").read()
    out, i = [], so.find(b"__CLANG_OFFLOAD_BUNDLE__")
    if i < 0:
        raise RuntimeError("no offload bundle in " + so_path)
    while i >= 0:
        (n,) = struct.unpack_from("<Q", so, i + 24)
        p = i + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", so, p)
            p += 24
            triple = so[p:p + tl].decode()
            p += tl
            if "gfx950" in triple and size > 0:
                out.append(so[i + off:i + off + size])
        i = so.find(b"__CLANG_OFFLOAD_BUNDLE__", i + 24)
    if not out:
        raise RuntimeError("no gfx950 image in " + so_path)
    return out


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
        return [re.sub(r"^void ", "", re.sub(r"\(anonymous namespace\)::", "", x)).replace("(ranenv_dev::KP)", "").replace("(KP)", "") for x in out.split("\n")[:len(names)]]
    except Exception:
        return list(names)


def kernel_resources(so_path: str = SO):
    kernels = []
    for co in code_objects(so_path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            notes = subprocess.run([os.path.join(LLVM_BIN, "llvm-readelf"), "--notes", f.name], capture_output=True, text=True, check=True).stdout
        kernels += _reparse(notes)
    for k, d in zip(kernels, demangle([k["mangled"] for k in kernels])):
        k["name"] = d
    return kernels


def _reparse(notes: str):
    """One dict per entry of amdhsa.kernels: an entry starts with '  - ', its own keys sit at four spaces (the fields of its
    .args list are indented deeper)."""
    out = []
    body = notes.split("amdhsa.kernels:", 1)[1].split("amdhsa.target:", 1)[0]
    for entry in re.split(r"\n  - ", "\n" + body)[1:]:
        d = {}
        for i, line in enumerate(entry.split("\n")):
            m = re.match(r"(\s*)\.(\w+):\s*(.*)$", line)
            if not m or (i > 0 and len(m.group(1)) != 4):
                continue
            k, v = m.group(2), m.group(3).strip().strip("'")
            if k == "name":
                d["mangled"] = v
            elif k in FIELDS:
                d[k] = int(v)
        if "mangled" in d:
            out.append(d)
    return out


def main():
    paths = [a for a in sys.argv[1:] if not a.startswith("--")]          # another build of the library: tools/kernel_resources.py <lib.so>
    ks = kernel_resources(paths[0]) if paths else kernel_resources()
    if "--json" in sys.argv:
        print(json.dumps(ks, indent=1))
        return
    print(f"{'kernel':78s} vgpr agpr sgpr spillV spillS scratch   lds")
    for k in sorted(ks, key=lambda k: k["name"]):
        print(f"{k['name'][:78]:78s} {k.get('vgpr_count', 0):4d} {k.get('agpr_count', 0):4d} {k.get('sgpr_count', 0):4d} {k.get('vgpr_spill_count', 0):6d} "
              f"{k.get('sgpr_spill_count', 0):6d} {k.get('private_segment_fixed_size', 0):7d} {k.get('group_segment_fixed_size', 0):5d}")
    bad = [k["name"] for k in ks if k.get("private_segment_fixed_size", 0) or k.get("vgpr_spill_count", 0)]
    print(f"{len(ks)} kernels; with scratch or VGPR spills: {len(bad)}")
    for b in bad:
        print("  ", b)


if __name__ == "__main__":
    main()
