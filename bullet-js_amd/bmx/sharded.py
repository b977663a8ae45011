"""sharded.py — one graph sharded by node-id hash over the GPUs of a node.

The reference replicates every delta to every peer over WebSocket gossip and lets every peer merge it
(src/bullet-network.js:378-418). Inside one node this module routes instead: rows live on exactly one shard
(owner = bmx_owner_of(id, N)), each rank partitions the deltas it originates by owner (K7, stable), the
32-byte records are exchanged with ONE all-to-all(v) (RCCL over xGMI when the backend is nccl), and each
shard merges what it received with the normal merge kernels. No other collective is on the data path.

Order: a shard receives the runs of rank 0, 1, ..., N-1 concatenated, each run in its origin's index order, so
the global batch order (rank-major, then index) restricted to the shard is preserved and the sequential
semantics of the merge (smallest index wins ties / creates absent rows) stay well defined.

Two exchanges for the pipelined mode (setup_pipeline(exchange=...), env BMX_SHARDED_EXCHANGE, default "auto"):
  direct   every rank owns receive slabs that the other ranks have MAPPED (hipIpc*): the owner partition of a batch writes each shard's
           slab straight into that shard's memory over xGMI (bmx_partition_scatter) and sets an arrival word there when its last record is
           stored; the owner's merge waits for the arrival words of all its origins (bmx_seq_wait_all), and its last workgroup tells every
           origin that the slab set is free again (bmx_merge_notify). One stream, no copy kernel beside the probe kernel, no collective.
  rccl     fixed slabs through ONE all-to-all on a second, high-priority stream (the round 1-2 path; what "auto" falls back to, on every
           rank together, when a rank cannot set up or verify the mappings).

Two ways to run a step:
  merge_step()        exact split sizes: the counts go through the host (two small syncs per step). Always safe.
  route() + merge()   pipelined: fixed-size slabs padded with reserved-id records (skipped by the merge), equal
                      splits, no host round trip; call route(b+1) BEFORE merge(b): the partition of b+1 runs on the
                      merge stream ahead of merge(b), its all-to-all on a second stream under merge(b), so the exchange
                      leaves the critical path. A slab overflow (one
                      shard receiving > slab records from one origin) is reported by overflowed(); the caller then
                      re-sends that batch with merge_step() — merging is idempotent.

`ops` hides where the bytes live: EngineOps = the HIP engine + torch CUDA tensors (product);
tests inject a CPU implementation to exercise the routing with the gloo backend.
"""
import contextlib
import os
import sys
import numpy as np
import torch

from . import INSERT_REFERENCE
from . import synth


class EngineOps:
    """GPU side: bmx.Engine does the work; torch only owns buffers, streams and events shared with RCCL."""

    def __init__(self, engine, device):
        self.e = engine
        self.device = device
        # merge stream = a dedicated stream that also becomes torch's CURRENT stream, so that in merge_step() partition ->
        # counts.cpu() -> all_to_all -> merge are ordered without extra events. (Not the default stream: its handle is 0,
        # which bmx_set_stream reads as "use the engine's own stream", and that one is not ordered with torch's work.)
        self._prev_stream = torch.cuda.current_stream(device)
        self.main = torch.cuda.Stream(device=device)
        self.main.wait_stream(self._prev_stream)
        torch.cuda.set_stream(self.main)
        engine.set_stream(self.main.cuda_stream)
        # the probe kernel leaves three of a SIMD's eight wave slots to the exchange stream's partition kernels, which are meant to run under it (they were
        # starved beside the full-occupancy kernel and finished when it did: profiles/r04_sharded_timeline.log); BMX_K1_WAVES in the environment wins
        if not os.environ.get("BMX_K1_WAVES"):
            engine.set_probe_waves(5)
        # The sharded pipeline keeps every batch's compaction on the merge stream: beside the exchange kernels the deferred form measured SLOWER (104-106 against
        # 93-99 us per step, profiles/r04_sharded_rehearsal_ab.log). Decided HERE, not by the bench, so that ShardedGraph users run what was measured;
        # BMX_SHARDED_DEFER=1 switches the deferral on for A/Bs. Queue-sharing note: the deferral's side stream and the exchange stream are both
        # high-priority streams; with the deferral off only ONE high-priority stream exists per process, so it cannot share a hardware queue with a waiting kernel.
        # BMX_SHARDED_DEFER=2: deferred, and the compactions run on the EXCHANGE stream (bmx_set_side_stream) instead of a third stream.
        self.defer_mode = os.environ.get("BMX_SHARDED_DEFER", "0")
        self.deferred = self.defer_mode in ("1", "2")
        if hasattr(engine, "set_deferred"):
            engine.set_deferred(self.deferred)
        self.comm = None
        self.pe = None

    def _pipeline(self):
        if self.comm is None:
            from . import Engine
            # a HIGH-PRIORITY stream: HIP maps streams onto a few hardware queues round-robin, and two streams that land on the same
            # queue run their kernels one after the other (seen in a kernel trace: merge and exchange streams both on queue 1, the
            # exchange kernel waiting for the probe kernel to finish, 170 instead of 125 us per step). Priority streams have queues of
            # their own, so the exchange always runs beside the merge.
            self.comm = torch.cuda.Stream(device=self.device, priority=-1)
            self.pe = Engine(capacity_rows=1024, device=self.device.index or 0)   # owns only the partition scratch
            self.pe.set_stream(self.comm.cuda_stream)
            if getattr(self, "defer_mode", "0") == "2":
                self.e.set_side_stream(self.comm.cuda_stream)

    direct_capable = True          # bmx_ipc_* / bmx_partition_scatter are available behind this ops object

    def gpu_identity(self, index=None):
        """(PCI domain, bus, device) of a GPU this process can see — how two processes find out whether they name the same GPU, whatever each calls it"""
        try:
            pr = torch.cuda.get_device_properties(self.device.index or 0 if index is None else index)
            return (int(pr.pci_domain_id), int(pr.pci_bus_id), int(pr.pci_device_id))
        except Exception:
            return None

    def visible_gpus(self):
        return {self.gpu_identity(i): i for i in range(torch.cuda.device_count())}

    def empty_records(self, n):
        return torch.empty((max(int(n), 1), 4), dtype=torch.int64, device=self.device)

    def zeros_i64(self, n):
        return torch.zeros(int(n), dtype=torch.int64, device=self.device)

    def zeros_i32(self, n):
        return torch.zeros(int(n), dtype=torch.int32, device=self.device)

    def partition(self, n, id, field, ts, val, nshards, recs_out, counts_out):
        self.e.partition_by_owner_dev(n, id, field, ts, val, nshards, recs_out, counts_out)

    def partition_slabs(self, n, id, field, ts, val, nshards, slab, recs_out, counts_out):
        # on the MERGE stream: the partition is short (~20 us alone) but slows to 60 us when it competes with the probe
        # kernel for memory bandwidth, so it runs between two merges and only the all-to-all overlaps the merge
        self.e.partition_by_owner_slabs_dev(n, id, field, ts, val, nshards, slab, recs_out, counts_out)

    def merge_records(self, n, recs, insert_mode, applied, n_applied):
        self.e.merge_records_dev(n, recs, insert_mode, applied=applied, n_applied=n_applied)

    def load_rows(self, id, field, ts, val):
        self.e.load_rows(id, field, ts, val)

    def sync(self):
        self.e.sync()
        if self.comm is not None:
            self.comm.synchronize()
        if self.pe is not None:
            self.pe.sync()                             # the partition engine's sticky status (slab overflow)
        self.main.synchronize()

    # stream plumbing for the pipelined mode: the communication stream becomes torch's CURRENT stream once
    # (RCCL collectives are enqueued on the current stream), the merge engine keeps its own stream; no per-step
    # context managers on the host path.
    def enter_pipeline(self):
        self._pipeline()
        if not getattr(self, "_entered", False):
            torch.cuda.set_stream(self.comm)
            self._entered = True

    def leave_pipeline(self):
        if getattr(self, "_entered", False):
            self.sync()
            torch.cuda.set_stream(self.main)
            self._entered = False

    def comm_ctx(self):
        self.enter_pipeline()
        return contextlib.nullcontext()

    # cross-stream ordering: 64-bit sequence words in device memory + one-wave kernels (bmx_seq_signal / bmx_seq_wait). Events are
    # not used on the data path: a record, or a wait on a not-yet-complete event, holds the stream it is enqueued on for ~10 us.
    def new_seq(self):
        return torch.zeros(1, dtype=torch.int64, device=self.device)

    def signal(self, seq, value, on_comm):
        self.e.seq_signal((self.comm if on_comm else self.main).cuda_stream, seq, value)

    def wait_seq(self, seq, at_least, on_comm):
        self.e.seq_wait((self.comm if on_comm else self.main).cuda_stream, seq, at_least)

    def partition_slabs_on_comm(self, n, id, field, ts, val, nshards, slab, recs_out, counts_out):
        # exchange stream, in front of the all-to-all that sends the slabs: runs underneath the merge of the previous batch
        self.pe.partition_by_owner_slabs_dev(n, id, field, ts, val, nshards, slab, recs_out, counts_out)

    def close(self):
        self.leave_pipeline()
        self.sync()
        if getattr(self, "defer_mode", "0") == "2" and self.comm is not None:
            self.e.set_side_stream(0)
        self.e.set_stream(0)                       # back to the engine's own stream
        torch.cuda.set_stream(self._prev_stream)
        if self.pe is not None:
            self.pe.close()
            self.pe = None


class ShardedGraph:
    def __init__(self, ops, dist, rank, world, insert_mode=INSERT_REFERENCE):
        self.ops, self.dist, self.rank, self.world = ops, dist, rank, world
        self.insert_mode = insert_mode
        self._recs = None
        self._recv = None
        self._applied = None
        self._n_applied = ops.zeros_i64(1)
        self._counts = ops.zeros_i64(world)
        self._rcounts = ops.zeros_i64(world)
        self.n_steps = 0
        self.sent_remote = 0
        self.received = 0
        self._direct = None
        self._tail_wait = os.environ.get("BMX_SHARDED_TAIL_WAIT", "1") == "1"
        self.exchange = "exact"
        self.direct_refused = None      # why the direct exchange was not taken (the first reason any rank gave), for the bench line

    def owned_rows(self, R_global, chunk=4_000_000):
        """row ordinals in [0, R_global) whose node this rank owns."""
        out = []
        for r0 in range(0, R_global, chunk):
            rows = np.arange(r0, min(R_global, r0 + chunk), dtype=np.int64)
            ids, _ = synth.rows_to_keys(rows)
            out.append(rows[synth.owner_of_np(ids, self.world) == self.rank])
        return np.concatenate(out) if out else np.zeros(0, np.int64)

    def owned_resident_host(self, R_per_gpu, T0=1_000_000, DT=1_000_000, seed=1):
        """This rank's part of a global graph of R_per_gpu * world rows (same rows as synth.big_resident), as host columns."""
        R_global = R_per_gpu * self.world
        rows = self.owned_rows(R_global)
        ids, fld = synth.rows_to_keys(rows)
        # same per-row values as big_resident(R_global): draws are indexed by the row ordinal
        with np.errstate(over="ignore"):
            i = rows.astype(np.uint64)
            base1 = np.uint64((seed * 0x632BE59BD9B4E019 + 1 * 0xD1342543DE82EF95) & synth.M64)
            base2 = np.uint64((seed * 0x632BE59BD9B4E019 + 2 * 0xD1342543DE82EF95) & synth.M64)
            u1 = synth.splitmix64_np(i * np.uint64(0x9E3779B97F4A7C15) + base1)
            u2 = synth.splitmix64_np(i * np.uint64(0x9E3779B97F4A7C15) + base2)
        ts = (T0 + (u1 % np.uint64(DT))).astype(np.int64)
        val = (u2 % np.uint64(1 << 32)).astype(np.int64) - (1 << 31)
        return ids, fld, ts, val

    def load_owned_resident(self, R_per_gpu, T0=1_000_000, DT=1_000_000, seed=1):
        """Load this rank's part of the global graph."""
        cols = self.owned_resident_host(R_per_gpu, T0=T0, DT=DT, seed=seed)
        self.ops.load_rows(*cols)
        return len(cols[0])

    def merge_step(self, n, id, field, ts, val):
        """Route this rank's n deltas to their owners and merge what arrives here. Returns the number received."""
        ops, dist, W = self.ops, self.dist, self.world
        # (safe beside the direct exchange: only bmx_merge_records_after — the merges that read a receive slab set — count up the peers' free words)
        if self._recs is None or self._recs.shape[0] < n:
            self._recs = ops.empty_records(n)
        ops.partition(n, id, field, ts, val, W, self._recs, self._counts)
        send = self._counts.cpu().tolist()               # host needs the split sizes (one sync)
        dist.all_to_all_single(self._rcounts, self._counts)
        recv = self._rcounts.cpu().tolist()
        nrecv = int(sum(recv))
        if self._recv is None or self._recv.shape[0] < nrecv:
            self._recv = ops.empty_records(int(nrecv * 1.25) + 1024)
            self._applied = torch.zeros(self._recv.shape[0], dtype=torch.int32, device=self._recv.device)
        dist.all_to_all_single(self._recv[:nrecv], self._recs[:n], output_split_sizes=recv, input_split_sizes=send)
        ops.merge_records(nrecv, self._recv, self.insert_mode, self._applied, self._n_applied)
        self.n_steps += 1
        self.sent_remote += n - send[self.rank]
        self.received += nrecv
        return nrecv

    # ---- pipelined mode ----------------------------------------------------------------------
    def setup_pipeline(self, max_deltas, slack=1.03, depth=2, partition_on="merge", exchange=None):
        """Allocate `depth` (>= 2) send/receive slab sets for batches of up to max_deltas deltas per rank. Protocol: call
        route(b+1) before merge(b) and merge in route order; at most `depth` routed-but-unmerged batches may exist.
        partition_on: "exchange" = the owner partition of batch b+1 runs on the exchange stream in front of its all-to-all, i.e.
        underneath merge(b); "merge" = it runs on the merge stream between two merges (default: 121 vs 125 us per step in the
        one-rank rehearsal — under the probe kernel the partition and the exchange kernels are dispatched late and run 2x slower)."""
        assert depth >= 2 and partition_on in ("exchange", "merge")
        W = self.world
        self.slab = int(max_deltas / W * slack) + 64
        self.partition_on = partition_on
        exchange = exchange or os.environ.get("BMX_SHARDED_EXCHANGE", "auto")
        assert exchange in ("auto", "direct", "rccl")
        self.exchange = "rccl"
        self._direct = None
        if exchange != "rccl" and getattr(self.ops, "direct_capable", False):
            if self._setup_direct(depth):
                self.exchange = "direct"
                return
            if exchange == "direct":
                raise RuntimeError("direct exchange requested but a rank could not set up or verify the IPC mappings")
        self._pipe = []
        for _ in range(depth):
            self._pipe.append(dict(send=self.ops.empty_records(W * self.slab), recv=self.ops.empty_records(W * self.slab),
                                   counts=self.ops.zeros_i64(W), applied=self.ops.zeros_i32(W * self.slab), n_applied=self.ops.zeros_i64(1),
                                   used=False))
        # one throw-away exchange now: the first all-to-all of a process group sets up its peer connections (seconds on 8 GPUs), which
        # must not happen while a device-side wait of the pipeline is spinning on it
        with self.ops.comm_ctx():
            self._pipe[0]["send"].zero_()
            self.dist.all_to_all_single(self._pipe[0]["recv"], self._pipe[0]["send"])
        self.ops.sync()
        # sequence words (device memory): batch k has been partitioned / exchanged / merged once the word is >= k+1
        self._parted = self.ops.new_seq()
        self._ready = self.ops.new_seq()
        self._merged_seq = self.ops.new_seq()
        self._routed = 0
        self._merged = 0
        self._due = []

    def route(self, n, id, field, ts, val, exchange_now=False):
        """Enqueue the owner partition of one batch and remember that its exchange is due; returns a ticket for merge(). The
        all-to-all itself is issued by the next merge() call AFTER that merge's kernels are enqueued, so the (slow, host-side)
        collective call never delays kernels the GPU could already be running. exchange_now: issue it at once instead — for the
        first batch of a pipeline, which has no merge to hide behind (its exchange then overlaps the next batch's partition)."""
        depth = len(self._pipe)
        assert self._routed - self._merged < depth, "merge() the oldest routed batch before routing another"
        k = self._routed
        p = self._pipe[k % depth]
        self._routed += 1
        p["k"] = k
        if self._direct is not None:
            self._route_direct(p, n, id, field, ts, val)
            self.sent_remote += n - n // self.world
            return p
        with self.ops.comm_ctx():                             # GPU: a no-op after the first call (streams are set once)
            if self.partition_on == "merge":
                # merge stream, between merge(k-2) and merge(k-1): the all-to-all that last read these send slabs finished before
                # merge(k-depth) started, and merge(k-depth) was the last reader of the receive slabs exchange k will overwrite
                self.ops.partition_slabs(n, id, field, ts, val, self.world, self.slab, p["send"], p["counts"])
                self.ops.signal(self._parted, k + 1, on_comm=False)
            else:
                p["args"] = (n, id, field, ts, val)           # partitioned on the exchange stream when the exchange is issued
        p["used"] = True
        p["exchanged"] = False
        if exchange_now:
            self._exchange(p)
        else:
            self._due.append(p)
        self.sent_remote += n - n // self.world
        return p

    def _exchange(self, p):
        k, depth = p["k"], len(self._pipe)
        with self.ops.comm_ctx():
            if self.partition_on == "merge":
                self.ops.wait_seq(self._parted, k + 1, on_comm=True)
            else:
                # the receive slabs were last read by merge(k-depth); the send slabs by exchange k-depth (this stream, earlier)
                if k >= depth:
                    self.ops.wait_seq(self._merged_seq, k - depth + 1, on_comm=True)
                n, id, field, ts, val = p.pop("args")
                self.ops.partition_slabs_on_comm(n, id, field, ts, val, self.world, self.slab, p["send"], p["counts"])
            self.dist.all_to_all_single(p["recv"], p["send"])  # equal splits: world slabs of `slab` records
            self.ops.signal(self._ready, k + 1, on_comm=True)
        p["exchanged"] = True

    def merge(self, ticket):
        """Merge a routed batch on the merge stream (waits for its exchange on the device, not on the host), then issue
        the exchanges of the batches routed meanwhile."""
        p = ticket
        if self._direct is not None:
            return self._merge_direct(p)
        if not p["exchanged"]:                                 # first batch of a pipeline: nothing to hide it behind
            self._due.remove(p)
            self._exchange(p)
        self.ops.wait_seq(self._ready, p["k"] + 1, on_comm=False)
        nrecv = self.world * self.slab                         # padding records are skipped by the kernel
        self.ops.merge_records(nrecv, p["recv"], self.insert_mode, p["applied"], p["n_applied"])
        if self.partition_on != "merge":
            self.ops.signal(self._merged_seq, p["k"] + 1, on_comm=False)
        while self._due:
            self._exchange(self._due.pop(0))
        self._merged += 1
        self.n_steps += 1
        self.received += nrecv
        return p

    # ---- direct exchange (IPC-mapped receive slabs) ------------------------------------------------
    def _all_agree(self, ok):
        flags = [None] * self.world
        self.dist.all_gather_object(flags, bool(ok))
        return all(flags)

    def _setup_direct(self, depth):
        """Allocate this rank's receive slabs + arrival / free words, exchange the IPC handles, map the peers', run one empty batch through the
        whole hand-off (partition -> peer stores -> arrival words -> merge -> free words) and check it. Every step is agreed on by ALL ranks:
        either everybody switches to the direct exchange or nobody does. Returns True when it is in use."""
        from . import BmxError
        e, W, r, dist = self.ops.e, self.world, self.rank, self.dist
        rec_bytes = 32
        own = {}                 # every allocation is recorded as it succeeds: whatever fails later, _free_direct gives all of them back
        try:
            # uncached: peers store into all three while kernels here poll or read them
            own["recv"], h_recv = e.ipc_alloc(depth * W * self.slab * rec_bytes, uncached=True)
            own["arrived"], h_arr = e.ipc_alloc(max(W * 8, 256), uncached=True)
            own["freed"], h_free = e.ipc_alloc(max(W * 8, 256), uncached=True)
            mine = (self.ops.device.index or 0, os.getpid(), h_recv, h_arr, h_free, self.ops.gpu_identity() if hasattr(self.ops, "gpu_identity") else None)
        except Exception as err:      # anything at all: a rank that left here without an answer would leave the others hanging in the all_gather below
            print("bmx sharded: direct exchange unavailable on rank %d (%s)" % (r, err), file=sys.stderr)
            mine = None
        infos = [None] * W
        dist.all_gather_object(infos, mine)
        if any(x is None for x in infos):
            self.direct_refused = "rank %d could not allocate or export its receive slabs" % [i for i, x in enumerate(infos) if x is None][0]
            self._free_direct(own, [])
            return False
        peers, opened, ok = [None] * W, [], True
        try:
            for g in range(W):
                if g == r:
                    peers[g] = own
                    continue
                dev_g, pci_g, pci_me = infos[g][0], infos[g][5], infos[r][5]
                if pci_g is not None and pci_me is not None:
                    # the peer's GPU as THIS process numbers it: ranks need not number the GPUs alike (per-rank visibility masks)
                    if pci_g == pci_me:
                        pd = -1
                    else:
                        pd = self.ops.visible_gpus().get(pci_g)
                        if pd is None:
                            raise BmxError(-1, "rank %d's GPU %s is not visible to this process: no peer access" % (g, pci_g))
                else:
                    pd = -1 if dev_g == (self.ops.device.index or 0) else dev_g
                ptrs = dict(recv=e.ipc_open(infos[g][2], pd), arrived=e.ipc_open(infos[g][3], pd), freed=e.ipc_open(infos[g][4], pd))
                opened.extend(ptrs.values())
                peers[g] = ptrs
        except Exception as err:      # not only BmxError: every rank must reach the agreement below
            print("bmx sharded: rank %d cannot map a peer's receive slabs (%s)" % (r, err), file=sys.stderr)
            ok = False
        if os.environ.get("BMX_SHARDED_FAIL_SETUP") == str(r):     # test hook: this rank's mapping step fails
            ok = False
        if not self._all_agree(ok):
            self.direct_refused = "a rank could not map a peer's receive slabs (hipIpcOpenMemHandle / peer access)"
            self._free_direct(own, opened)
            return False
        self.ops._pipeline()                                                   # the exchange stream and its partition context
        pe = self.ops.pe
        sets = []
        for q in range(depth):                                                 # pointer arrays per slab set, built once
            off = (q * W + r) * self.slab * rec_bytes
            sets.append(dict(dst=pe.ptr_array([peers[g]["recv"] + off for g in range(W)]), arrive=pe.ptr_array([peers[g]["arrived"] + 8 * r for g in range(W)])))
        self._direct = dict(own=own, peers=peers, opened=opened, depth=depth, sets=sets)
        self._pipe = [dict(counts=self.ops.zeros_i64(W), applied=self.ops.zeros_i32(W * self.slab), n_applied=self.ops.zeros_i64(1), used=False) for _ in range(depth)]
        self._routed = self._merged = 0
        self._due = []
        e.merge_notify([peers[g]["freed"] + 8 * r for g in range(W)])      # my merges tell origin g: word [me] of ITS free words
        dist.barrier()                                                        # nobody stores into a peer before every peer has mapped everything
        # pre-flight: batch 0 is empty (padding only); it travels the whole hand-off on every rank
        try:
            p = self.route(0, None, None, None, None)
            self.merge(p)
            self.ops.sync()
            ok = int(p["n_applied"].cpu()[0]) == 0
        except Exception as err:
            print("bmx sharded: rank %d: the direct exchange did not complete its pre-flight batch (%s)" % (r, err), file=sys.stderr)
            ok = False
        if not self._all_agree(ok):
            self.direct_refused = "the pre-flight batch did not complete its hand-off on some rank (a device-side wait expired or raised)"
            self._teardown_direct()
            return False
        self.n_steps = 0; self.received = 0; self.sent_remote = 0        # the pre-flight batch is not a step
        return True

    def _route_direct(self, p, n, id, field, ts, val):
        # on the EXCHANGE stream (a high-priority stream of its own, like the all-to-all it replaces): the scatter of batch k + 1 is bound by the
        # xGMI links (7/8 of the records leave the GPU) and runs underneath the merge of batch k; the two streams meet only through words in memory
        d, pe, W, k = self._direct, self.ops.pe, self.world, p["k"]
        depth = d["depth"]
        st = d["sets"][k % depth]
        # the slab set is free once EVERY owner has merged batch k - depth (their merges count up my free words): waited for on the device, same call
        pe.partition_scatter_raw(n, id, field, ts, val, W, self.slab, st["dst"], p["counts"], st["arrive"], k + 1,
                                 d["own"]["freed"] if k >= depth else 0, W, k - depth + 1)
        p["used"] = True

    def _merge_direct(self, p):
        d, e, W, k = self._direct, self.ops.e, self.world, p["k"]
        nrecv = W * self.slab
        # batch k + 1 is routed already (route(b + 1) comes before merge(b)): THIS merge's resolve kernel waits for its slabs' arrival words as its
        # last act, and merge(k + 1) then needs no wait launch in front of its probe kernel (bmx_merge_tail_wait)
        if self._tail_wait and self._routed > k + 1:
            e.merge_tail_wait(d["own"]["arrived"], W, k + 2)
        # every origin's slab of batch k has arrived (waited for on the device), then the merge: one host call
        e.merge_records_after(d["own"]["arrived"], W, k + 1, nrecv, d["own"]["recv"] + (k % d["depth"]) * nrecv * 32, self.insert_mode, p["applied"], p["n_applied"])
        self._merged += 1
        self.n_steps += 1
        self.received += nrecv
        return p

    def _free_direct(self, own, opened):
        e = self.ops.e
        for ptr in opened:
            try:
                e.ipc_close(ptr)
            except Exception:
                pass
        self.dist.barrier()                 # peers have unmapped before the owner frees
        if own:
            for ptr in own.values():
                try:
                    e.ipc_free(ptr)
                except Exception:
                    pass

    def _teardown_direct(self):
        d = self._direct
        if d is None:
            return
        try:
            self.ops.sync()
        except Exception:
            pass
        self.ops.e.merge_notify([])
        self.dist.barrier()                 # nobody is storing into a peer any more
        self._direct = None
        self._free_direct(d["own"], d["opened"])

    def close(self):
        """Give the direct exchange's mappings back (collective: every rank calls it)."""
        self._teardown_direct()

    def overflowed(self):
        """True if a slab of ANY batch routed so far was too small for what its origin had to send (host sync). The partition kernel
        raises a sticky error on its context when it drops records, so an overflow of an earlier batch whose counts have been
        overwritten since is still reported."""
        from . import BmxError, ERR_OVERFLOW
        try:
            self.ops.sync()
        except BmxError as e:
            if e.code != ERR_OVERFLOW:
                raise
            self._overflowed = True
        return getattr(self, "_overflowed", False) or any(int(p["counts"].max().item()) > self.slab for p in self._pipe if p.get("used"))

    def last_applied(self):
        """(indices into the received batch, received records) of the last step's winners."""
        self.ops.sync()
        na = int(self._n_applied.cpu().item())
        return self._applied[:na], self._recv

    def stats(self):
        return {"steps": self.n_steps, "records_sent_to_other_shards": self.sent_remote, "records_received": self.received,
                "bytes_per_record": 32}
