"""CPU-only: `python bench.py --gpus N` (N > 1) without WORLD_SIZE is a LAUNCHER — it starts the ranks as a child process
(python -m torch.distributed.run ...), relays rank 0's single JSON line and the child's exit code, and never touches the GPU itself
(a GPU-initialised process must not be what forks/execs the ranks)."""
import io
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _FakeChild:
    def __init__(self, lines, rc):
        self.stdout = io.StringIO("".join(l + "\n" for l in lines))
        self._rc = rc

    def wait(self):
        return self._rc


def _forbid_gpu(monkeypatch):
    import torch

    def boom(*a, **k):
        raise AssertionError("the launcher touched a GPU API before/while spawning the ranks")
    for name in ("is_available", "set_device", "init", "current_device", "synchronize", "device_count", "get_device_properties"):
        monkeypatch.setattr(torch.cuda, name, boom)


def test_launcher_spawns_torchrun_relays_one_line_and_touches_no_gpu(monkeypatch, capsys):
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    _forbid_gpu(monkeypatch)
    seen = {}
    line = json.dumps({"metric": "CRDT field-merges/s", "value": 1.0, "n_gpus": 2})

    def fake_popen(cmd, **kw):
        seen["cmd"] = cmd
        seen["env"] = kw.get("env")
        return _FakeChild(["NCCL version banner that does not belong on stdout", line], 0)
    monkeypatch.setattr(subprocess, "Popen", fake_popen)
    with pytest.raises(SystemExit) as ei:
        bench.main(["--gpus", "2", "--steps", "6", "--warmup", "2", "--config", "5"])
    assert ei.value.code == 0
    cmd = seen["cmd"]
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "2" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "2", "--steps", "6", "--warmup", "2", "--config", "5"]        # the ranks get the very same arguments
    assert seen["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
    out = capsys.readouterr()
    assert out.out.strip() == line                                   # exactly one line on stdout
    assert "NCCL version banner" in out.err


def test_launcher_fails_when_the_ranks_fail_or_say_nothing(monkeypatch, capsys):
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    _forbid_gpu(monkeypatch)
    monkeypatch.setattr(subprocess, "Popen", lambda cmd, **kw: _FakeChild([json.dumps({"metric": "x", "value": 1})], 3))
    with pytest.raises(SystemExit) as ei:
        bench.main(["--gpus", "4"])
    assert ei.value.code == 3 and capsys.readouterr().out == ""       # a failed run reports nothing, even if a line was printed
    monkeypatch.setattr(subprocess, "Popen", lambda cmd, **kw: _FakeChild(["no json here"], 0))
    with pytest.raises(SystemExit) as ei:
        bench.main(["--gpus", "4"])
    assert ei.value.code == 1 and capsys.readouterr().out == ""


def test_a_rank_is_not_a_launcher(monkeypatch):
    """with WORLD_SIZE set (a rank started by torch.distributed.run) main() goes on to the GPU check instead of spawning again"""
    sys.path.insert(0, ROOT)
    import bench
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setattr(subprocess, "Popen", lambda *a, **k: (_ for _ in ()).throw(AssertionError("a rank must not spawn ranks")))
    fd1 = os.dup(1)
    try:
        with pytest.raises(SystemExit) as ei:
            bench.main(["--gpus", "2"])
    finally:
        os.dup2(fd1, 1); os.close(fd1)         # main() points fd 1 at stderr for the ranks' native libraries
    assert "needs a GPU" in str(ei.value.code)


def test_real_launch_without_a_gpu_exits_nonzero_and_prints_no_json():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: the real multi-rank run belongs to the gpu tests / the driver")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "launching 2 ranks" in r.stderr and "needs a GPU" in r.stderr


def test_numa_node_lookup_for_the_js_children(tmp_path):
    """bench.py keeps its node children on the cores of ONE NUMA node (taskset): the node that holds the CPU the bench runs on, read from sysfs cpulists"""
    import bench
    for name, cl in (("node0", "0-63,128-191"), ("node1", "64-127,192-255")):
        d = tmp_path / name; d.mkdir(); (d / "cpulist").write_text(cl + "\n")
    assert bench.numa_node_of_cpu(5, str(tmp_path), 256) == ("node0", "0-63,128-191")
    assert bench.numa_node_of_cpu(130, str(tmp_path), 256) == ("node0", "0-63,128-191")
    assert bench.numa_node_of_cpu(200, str(tmp_path), 256) == ("node1", "64-127,192-255")
    assert bench.numa_node_of_cpu(300, str(tmp_path), 256) is None
    one = tmp_path / "single"; (one / "node0").mkdir(parents=True); (one / "node0" / "cpulist").write_text("0-7\n")
    assert bench.numa_node_of_cpu(3, str(one), 8) is None          # one node = the whole machine: nothing to pin
