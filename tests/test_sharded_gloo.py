"""N>1 routing on CPU: world_size-2 gloo. The exchange logic (stable owner partition -> all_to_all(v) ->
per-shard merge) is the product's bmx/sharded.py; the per-shard engine is replaced by the CPU oracle here
because this container has no GPU. The sharded result must equal ONE oracle fed the concatenated batch."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleOps:
    """CPU stand-in for EngineOps (tests only)."""

    def __init__(self):
        from oracle.oracle import Oracle
        self.o = Oracle()
        self.winners = None

    def empty_records(self, n):
        return torch.empty((max(int(n), 1), 4), dtype=torch.int64)

    def zeros_i64(self, n):
        return torch.zeros(int(n), dtype=torch.int64)

    def partition(self, n, id, field, ts, val, nshards, recs_out, counts_out):
        import bmx
        from oracle.oracle import owner_of
        idn = id.numpy().view(np.uint64)[:n]
        own = owner_of(idn, nshards)
        order = np.argsort(own, kind="stable")
        r = np.zeros(n, dtype=bmx.DELTA_REC_DTYPE)
        r["id"] = idn[order]; r["field"] = field.numpy().view(np.uint32)[:n][order]; r["aux"] = order.astype(np.uint32)
        r["ts"] = ts.numpy()[:n][order]; r["val"] = val.numpy()[:n][order]
        recs_out[:n] = torch.from_numpy(r.view(np.int64).reshape(n, 4))
        counts_out[:] = torch.from_numpy(np.bincount(own, minlength=nshards).astype(np.int64))

    def zeros_i32(self, n):
        return torch.zeros(int(n), dtype=torch.int32)

    def partition_slabs(self, n, id, field, ts, val, nshards, slab, recs_out, counts_out):
        tmp = torch.empty((n, 4), dtype=torch.int64)
        self.partition(n, id, field, ts, val, nshards, tmp, counts_out)
        recs_out.view(torch.uint8).fill_(0xFF)
        off = 0
        for g, c in enumerate(counts_out.tolist()):
            k = min(c, slab)
            recs_out[g * slab:g * slab + k] = tmp[off:off + k]
            off += c

    def comm_ctx(self):
        import contextlib
        return contextlib.nullcontext()

    # CPU stand-in: everything is synchronous, so the sequence words need no device
    def new_seq(self):
        return [0]

    def signal(self, seq, value, on_comm):
        seq[0] = value

    def wait_seq(self, seq, at_least, on_comm):
        assert seq[0] >= at_least, "the signal a wait refers to must already have been issued"

    def partition_slabs_on_comm(self, *a):
        self.partition_slabs(*a)

    def merge_records(self, n, recs, insert_mode, applied, n_applied):
        import bmx
        r = recs[:n].numpy().view(bmx.DELTA_REC_DTYPE).reshape(-1)
        r = r[r["id"] != np.uint64(2**64 - 1)]            # padding records are skipped, as k_probe_apply does
        _, w = self.o.merge_batch(r["id"], r["field"], r["ts"], r["val"], insert_mode)
        applied[:len(w)] = torch.from_numpy(w.astype(np.int32))
        n_applied[0] = len(w)

    def load_rows(self, id, field, ts, val):
        self.o.load_rows(id, field, ts, val)

    def sync(self):
        pass


class _FakeIpcEngine:
    """the bmx_ipc_* surface of an engine whose peer mapping FAILS on one rank (what a machine without peer access between two GPUs, or a
    refused hipIpcOpenMemHandle, looks like from bmx/sharded.py)"""

    def __init__(self, rank, fail_open_on):
        self.rank, self.fail_open_on, self.live, self.opened = rank, fail_open_on, set(), 0

    def ipc_alloc(self, nbytes, uncached=False):
        ptr = 0x1000000 * (len(self.live) + 1)
        self.live.add(ptr)
        return ptr, b"handle-%d-%d" % (self.rank, ptr)

    def ipc_open(self, handle, device):
        if self.rank == self.fail_open_on:
            raise RuntimeError("hipIpcOpenMemHandle: invalid argument (injected)")
        self.opened += 1
        return 0x7000000 + self.opened

    def ipc_close(self, ptr):
        self.opened -= 1

    def ipc_free(self, ptr):
        self.live.discard(ptr)


class DirectRefusingOps(OracleOps):
    """an ops object that SAYS it can do the direct exchange and whose rank 1 cannot map its peer's slabs: every rank must end on the all-to-all"""
    direct_capable = True

    def __init__(self, rank):
        super().__init__()
        self.e = _FakeIpcEngine(rank, fail_open_on=1)
        self.device = torch.device("cpu")


def _worker(rank, world, port, tmp, pipelined=False):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bullet-js_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bmx import synth
    from bmx.sharded import ShardedGraph
    refuse = pipelined == "direct_refused"
    if refuse:
        pipelined = "merge"
    ops = DirectRefusingOps(rank) if refuse else OracleOps()
    sg = ShardedGraph(ops, dist, rank, world)
    R = 20000
    nloaded = sg.load_owned_resident(R // world * world // world, T0=1000, DT=1000)   # R/world per rank
    digests = []
    for b in range(3):
        d = synth.big_deltas(3000, R, seed=5 + 100 * rank, T0=1000, DT=1000, insert_pct=15, hot_pct=30, hot_keys=40, unique=False, batch=b)
        t = [torch.from_numpy(np.ascontiguousarray(x).view(np.int64 if x.dtype.itemsize == 8 else np.int32)) for x in d]
        if pipelined:
            if b == 0:
                sg.setup_pipeline(3000, slack=1.2, partition_on=pipelined)
                if refuse:      # a set-up step failed on ONE rank: BOTH ranks are on the all-to-all, know why, and hold no mapping or allocation of the attempt
                    assert sg.exchange == "rccl" and sg._direct is None and "map a peer" in sg.direct_refused
                    assert not ops.e.live and ops.e.opened == 0
            p = sg.merge(sg.route(3000, *t))
            assert not sg.overflowed()
            digests.append(int(p["n_applied"][0]))
        else:
            sg.merge_step(3000, *t)
            applied, recv = sg.last_applied()
            digests.append(len(applied))
    id, f, ts, val = ops.o.dump_rows()
    np.savez(os.path.join(tmp, "rank%d.npz" % rank), id=id, f=f, ts=ts, val=val, nloaded=nloaded, winners=np.array(digests),
             sent=sg.sent_remote, recv=sg.received)
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize("pipelined", [False, "exchange", "merge", "direct_refused"])
def test_two_rank_routing_equals_single_merge(tmp_path, pipelined):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), pipelined), nprocs=world, join=True)
    from bmx import synth
    from oracle.oracle import Oracle, rows_digest, owner_of
    R = 20000
    # single-oracle truth: the whole graph, batches applied in global order (rank-major inside each step)
    o = Oracle()
    o.load_rows(*synth.big_resident(R, seed=1, T0=1000, DT=1000))
    for b in range(3):
        for rank in range(world):
            o.merge_batch(*synth.big_deltas(3000, R, seed=5 + 100 * rank, T0=1000, DT=1000, insert_pct=15, hot_pct=30, hot_keys=40, unique=False, batch=b))
    parts = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    assert sum(int(p["nloaded"]) for p in parts) == R
    for r, p in enumerate(parts):
        assert (owner_of(p["id"], world) == r).all()          # every row sits on its owner
        assert int(p["recv"]) > 0 and int(p["sent"]) > 0
    ids = np.concatenate([p["id"] for p in parts]); f = np.concatenate([p["f"] for p in parts])
    ts = np.concatenate([p["ts"] for p in parts]); val = np.concatenate([p["val"] for p in parts])
    assert len(ids) == len(o)
    assert rows_digest(ids, f, ts, val) == o.digest()          # union of shards == single merge, bit for bit
    assert sum(int(p["sent"]) for p in parts) + sum(int(p["recv"]) for p in parts) > 0
