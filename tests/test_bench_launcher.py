"""CPU: `python bench.py --gpus N` with WORLD_SIZE unset must start the N ranks itself (the driver's 1-GPU
style invocation with --gpus 8 died on an assert before touching a GPU). The launch path is rehearsed without a GPU: the
child ranks join a gloo group on 127.0.0.1, count themselves with one all-reduce, and rank 0's JSON line is relayed."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launcher_command_is_the_drivers_own():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launcher_command(8, 29511, ["--gpus", "8", "--steps", "5", "--warmup", "2"])
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]         # the script's own arguments follow it unchanged


def test_self_launch_two_ranks_gloo():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher-selftest"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["launcher_selftest"] and d["n_gpus"] == 2 and d["ranks_in_collective"] == 2
    assert d["master_addr"] == "127.0.0.1" and d["ipc_mode_legacy"] == "0"


def test_launched_rank_does_not_relaunch():
    """A rank started by the launcher (WORLD_SIZE set) runs the bench itself: with a mismatching --gpus it must stop on the
    explicit assertion, not fork another launcher."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher-selftest"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr
