"""The SHIPPED code object's own metadata (tools/kernel_resources.py: the gfx950 image inside csrc/libranenv_hip.so): no kernel of
the library has scratch or a spilled vector register, and the step kernels stay inside the register budget of the waves per SIMD
they are launched for.  VERDICT r4: DESIGN.md claimed "no kernel with scratch" while three packed builds spilled; this keeps
the claim from rotting."""
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))


@pytest.fixture(scope="module")
def kernels():
    from intent_radio_sched_multi_slice_amd.csrc import build as hip_build
    import kernel_resources
    hip_build.build()
    ks = kernel_resources.kernel_resources()
    assert len(ks) >= 60, len(ks)                       # every build of the step kernel + the small kernels
    return ks


def test_no_kernel_has_scratch_or_spilled_vector_registers(kernels):
    bad = {k["name"]: (k.get("private_segment_fixed_size"), k.get("vgpr_spill_count")) for k in kernels
           if k.get("private_segment_fixed_size", 0) != 0 or k.get("vgpr_spill_count", 0) != 0}
    assert not bad, bad


def test_step_kernels_fit_the_register_budget_of_their_waves_per_simd(kernels):
    """512 VGPRs per SIMD lane: 5 waves -> 96 (allocation granule 8), 4 -> 128, 2 -> 256.  No AGPRs anywhere (no MFMA: the RB x SE
    accumulation is a masked row reduction)."""
    def budget(name):
        if "_tiny" in name:           # the whole-row builds (persistent and one-TTI): 2 waves per SIMD
            return 256
        if "packed" in name or "_small" in name:
            return 128
        if "<" in name and (", 16" in name or "<16" in name or "16>" in name):          # the 16-wide row builds: 4 waves per SIMD
            return 128
        return 96
    for k in kernels:
        n = k["name"]
        assert k.get("agpr_count", 0) == 0, n
        if any(t in n for t in ("ranenv_core_kernel", "ranenv_persist_kernel")):
            assert k["vgpr_count"] <= budget(n), (n, k["vgpr_count"], budget(n))


def test_headline_kernels_keep_five_waves_per_simd(kernels):
    by = {k["name"]: k for k in kernels}
    for n in ("ranenv_core_kernel<0, 10, true>", "ranenv_core_kernel<0, 10, false>", "ranenv_core_kernel_gather<0, 10, true>",
              "ranenv_persist_kernel<true, 10>", "ranenv_persist_kernel<false, 10>", "ranenv_core_kernel_mixed<10, false, false>"):
        assert n in by, (n, sorted(by)[:5])
        assert by[n]["vgpr_count"] <= 96 and by[n]["private_segment_fixed_size"] == 0, (n, by[n])
