"""Are the kernels of two builds of the library instruction-identical?

    python tools/isa_compare.py old.so new.so

Disassembles every gfx950 code object inside each library's offload bundles (a library linked from several objects carries one
bundle per object), keys the kernels by demangled name (the namespace of the argument block is ignored) and compares the
instruction streams with addresses, encodings and comments stripped.  Prints the kernels that differ or exist on one side only.
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_resources as kr


def kernels_isa(so_path: str):
    out = {}
    for co in kr.code_objects(so_path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            dis = subprocess.run([os.path.join(kr.LLVM_BIN, "llvm-objdump"), "-d", "--no-show-raw-insn", "--no-leading-addr", f.name],
                                 capture_output=True, text=True, check=True).stdout
        cur, body = None, []
        for line in dis.split("\n"):
            m = re.match(r"^(?:[0-9a-f]+ )?<([^>]+)>:$", line)
            if m:
                if cur:
                    out[cur] = body
                cur, body = m.group(1), []
                continue
            if cur is None:
                continue
            ins = re.sub(r"\s*//.*$", "", line).strip()
            if ins:
                body.append(ins)
        if cur:
            out[cur] = body
    for body in out.values():          # (alignment padding behind a kernel depends on what follows it in the object)
        while body and re.match(r"^(s_nop\b|s_code_end\b|\.\.\.)", body[-1]):
            body.pop()
    names = list(out)
    return {d.replace("ranenv_dev::", ""): out[n] for n, d in zip(names, kr.demangle(names))}


def main():
    a, b = kernels_isa(sys.argv[1]), kernels_isa(sys.argv[2])
    same = diff = 0
    for n in sorted(set(a) | set(b)):
        if n not in a or n not in b:
            print(("only in old: " if n in a else "only in new: ") + n)
            diff += 1
        elif a[n] != b[n]:
            k = next((i for i, (x, y) in enumerate(zip(a[n], b[n])) if x != y), min(len(a[n]), len(b[n])))
            print(f"DIFFERS: {n}: {len(a[n])} vs {len(b[n])} instructions, first difference at {k}")
            diff += 1
        else:
            same += 1
    print(f"{same} kernels instruction-identical, {diff} differ")
    return 1 if diff else 0


if __name__ == "__main__":
    sys.exit(main())
