"""Kernels of the shipped library with spilled vector registers or scratch: python tools/spills.py [lib.so]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_resources as kr
ks = kr.kernel_resources(*sys.argv[1:2])
bad = [k for k in ks if k.get("vgpr_spill_count", 0) or k.get("private_segment_fixed_size", 0)]
for k in bad:
    print(f"{k['name']:60s} vgpr {k['vgpr_count']:4d} spillV {k['vgpr_spill_count']:3d} scratch {k['private_segment_fixed_size']:4d} spillS {k['sgpr_spill_count']:3d}")
print(f"{len(ks)} kernels, {len(bad)} with spills / scratch")
