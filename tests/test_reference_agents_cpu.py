"""tests/golden/agents_on_facade.npz (the reference's real agents on the facade, tests/golden/gen_golden_agents.py) replayed
on the CPU: the facade's host logic and this build's plugins with the CPU oracle's env core standing in for the HIP device,
exactly as when the fixture was made -- so what is checked here is everything but the kernel: plugin draw order on the
shared rng, the facade's call order, the oracle's agent side against the real IBSched / MARR / MAPF.  The GPU run of the
same replay is tests/test_gpu_reference_agents.py."""
import importlib.util
import os

import pytest

torch = pytest.importorskip("torch")

from intent_radio_sched_multi_slice_amd import comm_env
from tests.common import GOLDEN
from tests.test_gpu_reference_agents import replay_fixture


def _oracle_device():
    spec = importlib.util.spec_from_file_location("gen_golden_agents", os.path.join(GOLDEN, "gen_golden_agents.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)          # defines OracleDevice; its main() (which needs the reference) is not run
    return mod.OracleDevice


@pytest.mark.parametrize("name", ["ib_sched", "marr", "mapf"])
def test_fixture_of_the_real_reference_agents_replays_with_the_cpu_stand_in(name, tmp_path, monkeypatch):
    monkeypatch.setattr(comm_env, "BatchedRanEnv", _oracle_device())
    replay_fixture(name, tmp_path)
