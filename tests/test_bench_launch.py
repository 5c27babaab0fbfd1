"""bench.py's launch decision: `--gpus N` can never silently run as one rank."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_launch_decision_table():
    import bench
    d = bench.launch_decision
    assert d(1, {}) == ("run", None)
    assert d(1, {"RANK": "0", "WORLD_SIZE": "1"}) == ("run", None)
    assert d(8, {}) == ("spawn", None)                       # no launcher: bench.py starts its own ranks
    assert d(8, {"RANK": "3", "WORLD_SIZE": "8", "LOCAL_RANK": "3"}) == ("run", None)
    assert d(8, {"RANK": "0", "WORLD_SIZE": "1"})[0] == "error"   # a world that is not N: refuse
    assert d(2, {"RANK": "0", "WORLD_SIZE": "4"})[0] == "error"
    assert d(1, {"RANK": "0", "WORLD_SIZE": "2"})[0] == "error"
    assert d(4, {"RANK": "0"})[0] == "error"                      # RANK without WORLD_SIZE = world 1
    assert d(4, {"WORLD_SIZE": "x"})[0] == "error"


def test_spawn_is_a_child_process_with_the_contract_arguments(monkeypatch):
    import bench
    seen = {}

    class R:
        returncode = 7

    def fake_run(cmd, **kw):
        seen["cmd"] = cmd
        return R()

    monkeypatch.setattr(subprocess, "run", fake_run)
    assert bench.spawn_ranks(4, ["--gpus", "4", "--steps", "3", "--warmup", "1"]) == 7  # the child's exit code
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 <= int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]


def test_wrong_world_exits_non_zero_before_torch_is_imported():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, "-X", "importtime", os.path.join(ROOT, "bench.py"), "--gpus", "8"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 2
    assert p.stdout == ""                       # no JSON line that could be mistaken for a result
    assert "WORLD_SIZE=1" in p.stderr
    assert "| torch" not in p.stderr            # -X importtime lists every import: torch is not among them
