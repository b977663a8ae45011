"""GPU, two and four ranks: the N>1 product path (bmx/sharded.py with EngineOps) run by processes that share cuda:0, launched with
torch.distributed.run exactly as the driver launches bench.py. The union of the shards must equal ONE oracle fed the same batches in global
order, bit for bit. Both exchanges: the direct one (every rank's receive slabs are IPC-mapped by the others; the owner partition stores straight
into them and sets arrival words; merges wait for those and free the slabs through words in the origins' memory — the same code that crosses xGMI
between GPUs crosses process boundaries here) and the slab all-to-all (RCCL cannot run two ranks on one device: its transport here is gloo)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize("mode,world", [("exact", 2), ("pipelined_exchange_rccl", 2), ("pipelined_merge_rccl", 2), ("pipelined_merge_rccl", 4),
                                        ("pipelined_merge", 2), ("pipelined_merge", 4)])   # without _rccl: the direct exchange (IPC-mapped receive slabs, peer stores, arrival words)
def test_ranks_sharing_one_gpu_equal_single_merge(tmp_path, mode, world):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(HERE, "sharded_gpu_worker.py"), str(tmp_path), mode]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    from bmx import synth
    from oracle.oracle import Oracle, rows_digest, owner_of
    R, D, NB = 40000, 6000, 5
    o = Oracle()
    o.load_rows(*synth.big_resident(R, seed=1, T0=1000, DT=1000))
    for b in range(NB):
        for rank in range(world):
            o.merge_batch(*synth.big_deltas(D, R, seed=5 + 100 * rank, T0=1000, DT=1000, insert_pct=15, hot_pct=30, hot_keys=40, unique=False, batch=b))
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    assert sum(int(p["nloaded"]) for p in parts) == R
    for r, p in enumerate(parts):
        assert (owner_of(p["id"], world) == r).all()
        assert int(p["recv"]) > 0 and int(p["sent"]) > 0 and int(p["winners"].sum()) > 0
    ids = np.concatenate([p["id"] for p in parts]); f = np.concatenate([p["f"] for p in parts])
    ts = np.concatenate([p["ts"] for p in parts]); val = np.concatenate([p["val"] for p in parts])
    assert len(ids) == len(o)
    assert rows_digest(ids, f, ts, val) == o.digest()


@pytest.mark.parametrize("hook", ["BMX_BENCH_SELFTEST_RAISE", "BMX_SHARDED_FAIL_SETUP"])
def test_a_rank_that_fails_in_the_direct_exchange_sends_every_rank_to_the_fallback(hook):
    """bench.py launched by ITSELF (`--gpus 2`: launcher -> torch.distributed.run -> two ranks sharing this GPU over gloo); rank 1 raises in the middle
    of the direct exchange's self-test, or fails its mapping step in the set-up. Every rank must fall back to the all-to-all INSIDE the same processes: one short
    line, exchange kind "rccl", the refused path and the reason named, every shard verified against the oracle."""
    import json
    root = os.path.dirname(HERE)
    env = dict(os.environ, BMX_BENCH_ONE_GPU_REHEARSAL="1")
    env[hook] = "1"
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--rank-timeout", "500"],
                       capture_output=True, text=True, timeout=560, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    assert len(r.stdout.strip().splitlines()) == 1 and len(r.stdout) < 4096
    line = json.loads(r.stdout)
    assert line["n_gpus"] == 2 and line["verified"]["ok"] and line["verified"]["ranks_ok"] == 2
    assert line["exchange"]["kind"] == "rccl" and line["exchange"]["refused"] == "direct" and line["exchange"]["why"]
    assert line["roofline"]["frac"] > 0 and line["cpu_baseline"]["value"] > 0


@pytest.mark.skipif(__import__("torch").cuda.device_count() < 2, reason="needs two physical GPUs: the direct exchange across xGMI (peer access + IPC mappings between devices) runs here first")
@pytest.mark.parametrize("world", [2, 4, 8])
def test_one_rank_per_physical_gpu_bench_verifies_itself(world):
    """bench.py exactly as the driver launches it (one process per GPU, RCCL process group, direct exchange with RCCL as the agreed fallback): a short
    run whose every shard is compared with the oracle replay inside bench.py itself (`verified`)."""
    import json
    import torch
    if torch.cuda.device_count() < world:
        pytest.skip("needs %d GPUs" % world)
    root = os.path.dirname(HERE)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "6", "--warmup", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == world and line["verified"]["ok"]
