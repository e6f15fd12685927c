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


_EXCHANGE_WORKER = r"""
import json, os, sys, time
sys.path.insert(0, sys.argv[1])
import torch
import torch.distributed as dist
import bench

class FakeEngine:
    # the C-ABI's communicator calls, as far as choose_exchange uses them; rank `fail_rank` cannot create its communicator
    # (mode "raise") or never returns from the rendezvous (mode "hang")
    def __init__(self, rank, fail_rank, mode="raise"):
        self.rank, self.fail_rank, self.mode, self.inited, self.destroyed = rank, fail_rank, mode, False, False
        self.n_ids = 0
    def comm_unique_id(self):
        self.n_ids += 1
        return bytes([self.n_ids]) * 128
    def comm_init(self, rank, world, uid):
        assert len(uid) == 128
        if rank == self.fail_rank:
            if self.mode == "hang":
                time.sleep(3600)
            raise RuntimeError("ncclCommInitRank failed")
        self.inited = True
    def comm_destroy(self):
        self.destroyed = True

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
out = {}
for name, fail, mode in (("one_rank_fails", 1, "raise"), ("one_rank_hangs", 1, "hang"), ("all_ok", -1, "raise")):
    e = FakeEngine(rank, -1)                     # the measured engine itself never fails: the PROBE decides
    probes = []
    def make_probe():
        probes.append(FakeEngine(rank, fail, mode))
        return probes[-1]
    # backend label "nccl": the decision logic of the real runs; the collectives themselves run on this gloo group
    t0 = time.time()
    label = bench.choose_exchange(e, rank, world, torch.device("cpu"), "nccl", make_probe=make_probe, probe_timeout=3.0)
    out[name] = dict(label=label, inited=e.inited, destroyed=e.destroyed, probe_inited=probes[0].inited, probe_destroyed=probes[0].destroyed,
                     seconds=time.time() - t0)
out["torch_comm"] = bench.choose_exchange(FakeEngine(rank, -1), rank, world, torch.device("cpu"), "nccl", torch_comm=True)
print("RANK%d %s" % (rank, json.dumps(out)), flush=True)
dist.destroy_process_group()
os._exit(0)                                      # a probe thread may still sleep inside its fake rendezvous
"""


def test_one_rank_failing_or_hanging_comm_init_sends_every_rank_to_the_torch_path(tmp_path):
    """the exchange is chosen collectively, and ncclCommInitRank (never run here with more than one rank) is first tried
    on a throw-away probe handle under a deadline. Two gloo ranks drive bench.choose_exchange with stand-ins for the engine's
    communicator calls: when rank 1's probe raises OR never returns, BOTH ranks report the torch path within the deadline, the
    measured engine is never initialised and the probe that did succeed is destroyed; when every probe succeeds both ranks report
    RCCL behind the C-ABI with the measured engine initialised; --torch-comm never tries."""
    import socket
    script = tmp_path / "worker.py"
    script.write_text(_EXCHANGE_WORKER)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(script), ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    res = {}
    for ln in p.stdout.splitlines():
        if ln.startswith("RANK"):
            res[int(ln[4])] = json.loads(ln[6:])
    assert set(res) == {0, 1}
    for r in (0, 1):
        for case in ("one_rank_fails", "one_rank_hangs"):
            assert res[r][case]["label"] == "torch.distributed (nccl)", (r, case)
            assert not res[r][case]["inited"] and not res[r][case]["destroyed"]             # the measured engine was never touched
            assert res[r][case]["seconds"] < 30
        assert res[r]["all_ok"]["label"].startswith("rag_comm_allgather_dev") and res[r]["all_ok"]["inited"] and not res[r]["all_ok"]["destroyed"]
        assert res[r]["all_ok"]["probe_inited"] and res[r]["all_ok"]["probe_destroyed"]
        assert res[r]["torch_comm"] == "torch.distributed (nccl)"
    for case in ("one_rank_fails", "one_rank_hangs"):
        assert res[0][case]["probe_inited"] and res[0][case]["probe_destroyed"]            # the rank whose probe succeeded backs out
        assert not res[1][case]["probe_inited"]
