"""bench.py's rank logic without a GPU: world-size-2 gloo run of the shard ranges, per-rank seeds, the barrier + sync
bracket, the MAX-over-ranks clock, the metrics all_gather and the JSON line, with a stub env in place of the HIP one."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import json, os, sys, time, types
sys.path.insert(0, os.environ["REPO"])
import torch, torch.distributed as dist
import bench
from intent_radio_sched_multi_slice_amd.dist import gather_metrics, local_metrics, shard_range, summarize
world, rank, local_rank = bench.rank_env()
dist.init_process_group("gloo", rank=rank, world_size=world)
B, U, S, K = 8, 5, 3, 7
lo, hi = shard_range(B * world, rank, world)
assert hi - lo == B
class StubEnv:
    def __init__(self):
        g = torch.Generator().manual_seed(10 + 31 * (rank + 1))
        self.reward = torch.rand((B, S + 1), generator=g, dtype=torch.float64) - 0.5
        self.done = torch.zeros(B, dtype=torch.uint8)
        self.n = 0
        z = lambda: torch.ones((B, U), dtype=torch.int32) * (rank + 1)
        self._v = {"pkt_effective_thr": z(), "dropped_pkts": z(), "pkt_incoming": z() * 2, "queue_pkts": z()}
    def step(self):
        self.n += 1
        time.sleep(0.002 * (rank + 1))          # rank 1 is slower: the MAX must pick it up
    def views(self): return self._v
env = StubEnv()
def max_over_ranks(x):
    t = torch.tensor([x], dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX); return float(t.item())
elapsed = bench.timed_steps(env.step, K, lambda: None, dist.barrier, max_over_ranks)
assert env.n == K
gathered = gather_metrics(local_metrics(env.reward, env.views(), env.done, K))
assert gathered.shape == (world, 8)
args = types.SimpleNamespace(steps=K, warmup=2, config=2, traces=4, trace_len=5)
# blocks of exactly K steps, repeated until the sample is long enough: every rank must derive the same repeat count
n0 = env.n
times = bench.timed_blocks(lambda: [env.step() for _ in range(K)], lambda: None, dist.barrier, max_over_ranks, sample_s=0.05)
assert len(times) >= 2 and (env.n - n0) == K * (len(times) + 1)
line = bench.build_line(args, world, B, "stub", (S, U, 9), 1000, [elapsed], {"step": 0.5, "n_launches": 2 * K, "n_ttis": 2 * K}, 2, None, summarize(gathered),
                        extras={"pipelined_step": bench.block_stats(times, B * world * K, K)})
# one file per rank: two ranks printing at once can interleave on the launcher's pipe
with open(os.path.join(os.environ["OUT_DIR"], f"rank{rank}.txt"), "w") as f:
    if rank == 0:
        f.write("LINE " + json.dumps(line) + "\n")
    f.write(f"RANK {rank} elapsed {elapsed!r} lo {lo} hi {hi} repeats {len(times)}\n")
dist.barrier(); dist.destroy_process_group()
'''


def test_bench_rank_logic_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, REPO=REPO, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", OUT_DIR=str(tmp_path))
    port = 29500 + (os.getpid() % 2000)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         env=env, capture_output=True, text=True, timeout=180)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    text = "".join((tmp_path / f"rank{r}.txt").read_text() for r in range(2))
    lines = [l for l in text.splitlines() if l.startswith("LINE ")]
    ranks = sorted(l for l in text.splitlines() if l.startswith("RANK "))
    assert len(lines) == 1 and len(ranks) == 2
    assert ranks[0].split()[-1] == ranks[1].split()[-1]         # the same repeat count on both ranks
    line = json.loads(lines[0][5:])
    K, B, world = 7, 8, 2
    e0, e1 = (float(r.split()[3]) for r in ranks)
    assert e0 == e1                                             # both ranks report the MAX over ranks ...
    assert e0 >= K * 0.004                                      # ... which is the slow rank's time
    assert "lo 0 hi 8" in ranks[0] and "lo 8 hi 16" in ranks[1]
    assert line["repeats"] == 1 and line["pipelined_step"]["repeats"] >= 2
    assert line["pipelined_step"]["value_min"] <= line["pipelined_step"]["value"] <= line["pipelined_step"]["value_max"]
    assert line["n_gpus"] == world and line["steps"] == K and line["scaling"] == "weak"
    assert line["value"] == pytest.approx(B * world * K / e0, rel=1e-9)          # whole-job aggregate
    assert line["ms_per_step"] == pytest.approx(e0 / K * 1e3, rel=1e-9)
    rf = line["roofline"]
    assert rf["frac"] == pytest.approx(line["value"] / world * 1000 / 8e12, rel=1e-9)   # same clock as value, per GPU
    assert rf["dominant_kernel"]["ms"] == 0.5 and rf["traffic"] is None and rf["dominant_kernel"]["concurrent_launches"] == 2
    assert rf["dominant_kernel"]["algorithmic_bytes"] == 1000 * B / 2 and "partitions" in line["config"]["launch"]
    m = line["metrics"]
    assert m["env_steps"] == B * K * world and m["pkts_sent"] == B * 5 * (1 + 2) and m["pkts_incoming"] == 2 * m["pkts_sent"]
    assert line["config"]["global_batch"] == B * world


def test_plain_gpus_n_command_starts_its_own_ranks_and_relays_line_and_rc(monkeypatch, capsys):
    """`python bench.py --gpus 2 ...` without a launcher environment: the child command is torch.distributed.run with one
    rank per GPU on 127.0.0.1 and the same arguments; rank 0's JSON line goes to stdout, other output to stderr, the
    child's exit code is returned.  The child is a stub: nothing here starts a rank or touches a GPU."""
    import io
    sys.path.insert(0, REPO)
    import bench
    seen = {}

    class StubChild:
        def __init__(self, cmd, env=None, stdout=None, text=None):
            seen["cmd"], seen["env"] = cmd, env
            self.stdout = io.StringIO('some rank noise\n{"not": "the line"}\n{"metric": "env-steps/s", "value": 1.0, "n_gpus": 2}\n')

        def wait(self):
            return 7

    argv = ["--gpus", "2", "--steps", "20", "--warmup", "5", "--rehearse-on-one-gpu"]
    rc = bench.self_launch(2, argv, popen=StubChild)
    out = capsys.readouterr()
    assert rc == 7
    assert out.out.strip() == '{"metric": "env-steps/s", "value": 1.0, "n_gpus": 2}'
    assert "some rank noise" in out.err and '{"not": "the line"}' in out.err
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "2"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    k = cmd.index(os.path.join(REPO, "bench.py"))
    assert cmd[k + 1:] == argv
    assert seen["env"]["MASTER_ADDR"] == "127.0.0.1" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"

    # main() takes that path only for N > 1 with no WORLD_SIZE, and before importing anything that could touch a GPU
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py"] + argv)
    called = {}
    monkeypatch.setattr(bench, "self_launch", lambda n, a: called.setdefault("args", (n, a)) and 0 or 3)
    with pytest.raises(SystemExit) as ei:
        bench.main()
    assert ei.value.code == 3 and called["args"] == (2, argv)
