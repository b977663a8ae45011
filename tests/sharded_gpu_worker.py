"""Worker of tests/test_gpu_sharded_ranks.py: one rank of a sharded graph with the REAL engine (HIP kernels, device slabs,
streams, events) on cuda:0. Launched by torch.distributed.run like bench.py; backend gloo because two ranks share the one
GPU of the test box (RCCL refuses duplicate devices) — everything except the transport is the N>1 product path."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bullet-js_amd"))


def main():
    out_dir, mode = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bmx
    from bmx import synth
    from bmx.sharded import ShardedGraph, EngineOps
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    eng = bmx.Engine(200_000, device=0)
    ops = EngineOps(eng, dev)
    sg = ShardedGraph(ops, dist, rank, world)
    R, D, NB = 40000, 6000, 5
    nloaded = sg.load_owned_resident(R // world, T0=1000, DT=1000)
    counts = []
    batches = []
    for b in range(NB):
        d = synth.big_deltas(D, R, seed=5 + 100 * rank, T0=1000, DT=1000, insert_pct=15, hot_pct=30, hot_keys=40, unique=False, batch=b)
        batches.append([torch.from_numpy(np.ascontiguousarray(x).view(np.int64 if x.dtype.itemsize == 8 else np.int32)).to(dev) for x in d])
    if mode.startswith("pipelined"):
        parts = mode.split("_")            # pipelined_<where the partition runs>[_rccl]: the direct (IPC) exchange unless "rccl" is asked for
        sg.setup_pipeline(D, slack=1.2, partition_on=parts[1], exchange="rccl" if parts[-1] == "rccl" else "direct")
        assert sg.exchange == ("rccl" if parts[-1] == "rccl" else "direct")
        tk = sg.route(D, *batches[0])
        for b in range(NB):
            nxt = sg.route(D, *batches[b + 1]) if b + 1 < NB else None     # route(b+1) before merge(b), as bench.py does
            p = sg.merge(tk)
            tk = nxt
        assert not sg.overflowed()
        ops.sync()
        counts = [int(p["n_applied"].cpu()[0]) for p in sg._pipe]
    else:
        for b in range(NB):
            sg.merge_step(D, *batches[b])
            applied, _ = sg.last_applied()
            counts.append(int(applied.shape[0]))
    ops.sync()
    id, f, ts, val = eng.dump_rows()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), id=id, f=f, ts=ts, val=val, nloaded=nloaded, winners=np.array(counts),
             sent=sg.sent_remote, recv=sg.received)
    dist.barrier()
    sg.close()
    ops.close()
    eng.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
