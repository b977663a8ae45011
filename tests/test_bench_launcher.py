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
    pid = 2 ** 22 + 12345          # no such process group: a kill aimed at it can only fail

    def __init__(self, lines, rc):
        self.stdout = io.StringIO("".join(l + "\n" for l in lines))
        self._rc = rc

    def wait(self, timeout=None):
        return self._rc

    def kill(self):
        pass


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
    seen, kw_seen = {}, {}
    line = json.dumps({"metric": "CRDT field-merges/s", "value": 1.0, "n_gpus": 2})

    def fake_popen(cmd, **kw):
        seen["cmd"] = cmd
        seen["env"] = kw.get("env")
        kw_seen.update(kw)
        return _FakeChild(["NCCL version banner that does not belong on stdout", line], 0)
    monkeypatch.setattr(subprocess, "Popen", fake_popen)
    with pytest.raises(SystemExit) as ei:
        bench.main(["--gpus", "2", "--steps", "6", "--warmup", "2", "--config", "5"])
    assert ei.value.code == 0
    cmd = seen["cmd"]
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "2" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[cmd.index("--tee") + 1] == "2" and "--log-dir" in cmd            # every rank's stderr is kept: what a killed run shows
    assert kw_seen.get("start_new_session") is True                               # a process group of its own: the only thing a time-out kills
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


def test_launcher_kills_a_hung_child_within_its_limit_and_reports_nothing(tmp_path):
    """a rank stuck in a rendezvous / IPC open / barrier must cost the launcher's limit, not the driver's slot. A REAL child
    (a stand-in for torch.distributed.run that sleeps, with a grandchild that sleeps too) is started through the launcher's own code path:
    the launcher returns 124 inside the limit, prints no JSON, and the whole process group is gone."""
    import time
    import bench
    marker = tmp_path / "pids"
    fake = tmp_path / "fake_torchrun.py"
    fake.write_text(
        "import os, subprocess, sys, time\n"
        "g = subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(600)'])\n"
        "open(%r, 'w').write('%%d %%d' %% (os.getpid(), g.pid))\n"
        "sys.stderr.write('rank 0: stuck in init_process_group\\n'); sys.stderr.flush()\n"
        "time.sleep(600)\n" % str(marker))
    real_popen = subprocess.Popen

    def popen(cmd, **kw):
        return real_popen([sys.executable, str(fake)], **kw)
    old = subprocess.Popen
    subprocess.Popen = popen
    try:
        t0 = time.monotonic()
        rc = bench.launch_ranks(2, ["--gpus", "2"], timeout_s=3.0)
        dt = time.monotonic() - t0
    finally:
        subprocess.Popen = old
    assert rc == 124 and dt < 25.0
    pids = [int(x) for x in marker.read_text().split()]
    time.sleep(0.5)
    for pid in pids:                                   # child and grandchild: both terminated with the group
        alive = True
        try:
            os.kill(pid, 0)
            alive = open("/proc/%d/stat" % pid).read().rsplit(")", 1)[1].split()[0] != "Z"
        except (ProcessLookupError, FileNotFoundError):
            alive = False
        assert not alive, "process %d of the hung run survived the launcher's limit" % pid


def test_a_result_that_arrived_before_a_hung_shutdown_is_still_delivered(tmp_path, capsys):
    import bench
    line = json.dumps({"metric": "CRDT field-merges/s", "value": 2.0, "n_gpus": 2})
    fake = tmp_path / "fake_torchrun.py"
    fake.write_text("import sys, time\nprint(%r); sys.stdout.flush()\ntime.sleep(600)\n" % line)
    real_popen = subprocess.Popen
    old = subprocess.Popen
    subprocess.Popen = lambda cmd, **kw: real_popen([sys.executable, str(fake)], **kw)
    try:
        rc = bench.launch_ranks(2, ["--gpus", "2"], timeout_s=3.0)
    finally:
        subprocess.Popen = old
    out = capsys.readouterr()
    assert rc == 0 and out.out.strip() == line and "did not finish within" in out.err


def _full_out():
    """a synthetic record with EVERY optional section present, shaped like (and larger than) round 4's 21 KB line"""
    q = lambda us: {"matches": 9994936, "us": us, "achieved_GBs": 2126.3, "frac_of_8TBs": 0.2658, "rows_per_s": 443011035710,
                    "position_output": {"us": 84.94, "achieved_GBs": 5179.7, "frac_of_8TBs": 0.6475, "algorithmic_bytes": "w*R + 4*M"},
                    "roofline_mask_kernel": {"bound": "hbm", "kernel": "k_scan_mask", "achieved": 6123.1, "peak": 8000.0, "unit": "GB/s", "frac": 0.7654, "timed_as": "x" * 170,
                                             "count_only_scan_us": 65.3, "traffic": 400090000, "traffic_emit": 748000000, "kernel_us_between_events": {"scan_mask": 70.1, "offsets_and_emit": 150.2}}}
    ovq = {"matches": 100070, "us": 7.36, "bytes_moved": 1601120, "moved_GBs": 217.4, "position_output_us": 8.65, "count_only_us": 5.3, "speedup_over_the_column_scan": 10.8}
    size = lambda: {"rows": 100000000, "column": "int32", "index_first_build_ms": 3.9, "index_build_ms": 3.9, "equals_0.1pct": q(79.3), "range_1pct": q(91.3), "range_10pct": q(225.7),
                    "range_50pct": q(337.7), "verified": {"against": "numpy " * 30, "scans_checked": 16, "ok": True},
                    "ordered_view": {"sort_ms": 4.6, "equals_0.1pct": ovq, "range_1pct": ovq, "range_10pct": ovq, "range_50pct": ovq, "note": "n" * 300,
                                     "first_equals_after_a_1M_delta_merge_us": 812.5, "next_equals_us": 7.1, "view_kept_current_by": "patch", "patch_us": 700.1, "patch_keys": 2100000},
                    "first_scan_after_a_1M_delta_merge": {"us": 293.9, "index_brought_up_to_date_by": "change log", "matches": 100096, "index_rows": 100100000}}
    return {"metric": "CRDT field-merges/s", "value": 12872332136.121153, "unit": "merges/s", "n_gpus": 8, "steps": 20, "warmup": 3, "ms_per_step": 0.07768600044073537,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int64", "data": "synthetic",
            "config": {"workload": "config 4 shape: 80M-row graph id-hash sharded over 8 MI355X, 8M mixed-shard deltas per step routed to their owners by direct peer stores (fallback: RCCL all-to-all)",
                       "resident_rows_per_gpu": 10000000, "deltas_per_step_per_gpu": 1000000, "insert_mode": "reference", "sharding": "owner = hash(node id) mod N"},
            "roofline": {"bound": "hbm", "kernel": "k_probe_apply", "achieved": 977.8, "peak": 8000.0, "unit": "GB/s", "frac": 0.1222, "traffic": 228329984,
                         "requests_per_launch": {"total": 3145863, "reads": 1288329, "writes_incl_atomics": 1857534, "atomics": 886108}, "algorithmic_bytes_per_launch": 69402625.6,
                         "kernel_ms": {"probe_apply": 0.07098, "resolve_lists": 0.0066, "compact": 0.00947}, "launches_averaged": 12, "whole_merge_achieved_GBs": 936.5,
                         "traffic_source": "t" * 160, "note": "n" * 140},
            "verified": {"against": "oracle " * 20, "batches": 184, "winner_indices_compared": 19159209, "rows": 12298638, "table_digest": "a55ef666384fd8b3", "ok": True, "seconds": 3.48,
                         "per_rank": [{"rank": r, "ok": True, "rows": 10799352, "table_digest": "ac6ba45200371b52"} for r in range(8)]},
            "unique_keys_mode": {"note": "u" * 100, "ms_per_step_with_event_brackets": 0.09249, "kernel_ms": {"probe_apply": 0.06273, "resolve_lists": 0.00543, "compact": 0.00918}},
            "event_ms_per_step": 0.07661, "winners_per_step": 837664.1, "host_enqueue_ms_per_step": 0.0433,
            "table_placement": {"candidates": 5, "probe_us_chosen": 65.92, "probe_us_slowest": 75.16, "note": "p" * 210},
            "deferred_compaction": {"merges_deferred": 23, "compactions_on_side_stream": 21},
            "exchange": {"steps": 31, "records_sent_to_other_shards": 7000000, "records_received": 31931984, "bytes_per_record": 32, "kind": "rccl", "refused": "direct",
                         "why": "rank 3 cannot map a peer's receive slabs (hipIpcOpenMemHandle: invalid argument) " * 4, "mode": "m" * 300},
            "scan_config3": {"10M": size(), "10M_int64": size(), "100M": size(), "100M_int64": size()},
            "js_host": {"applied_path": {"batchSync_apply_entries_per_s": 698754.1, "per_entry_host_loop_entries_per_s": 262277.4, "nodes": 1099723},
                        "vector": {"mergeEntries_per_s": 3624708.3, "concurrent_merges": 123642, "writers": 3, "mergeEntriesPipelined_per_s": 5097050.7, "host_only": {"hostOnlyPaths": 0}},
                        "mergeEntries_per_s": 3679731.5, "mergeEntriesPipelined_per_s": 7256021.4, "mergeBatch_typed_columns_per_s": 927796122.3, "applied_path_sample": "a" * 444,
                        "vector_sample": "v" * 169, "sample": "s" * 358, "cpus": "node1: CPUs 64-127,192-255"},
            "cpu_baseline": {"value": 11962425.662124598, "unit": "merges/s", "cores": 1, "kind": "port", "sample": "c" * 400,
                             "all_cores": {"value": 31381467.6, "unit": "merges/s", "cores": 64, "kind": "port", "sample": "x" * 96},
                             "js_twin": {"value": 630342.7, "unit": "merges/s", "cores": 1, "node": "v12.22.9", "sample": "y" * 102}}}


def test_the_stdout_line_stays_short_whatever_sections_a_run_adds(tmp_path):
    """VERDICT r4 item 1 / ADVICE r4: round 4's single line grew to 21 KB, the driver kept an 8 KB tail and parsed nothing. The line the driver
    reads is bounded: < 4096 bytes with every optional section present, contract fields first; everything else goes to bench_detail.json."""
    import bench
    out = _full_out()
    assert len(json.dumps(out)) > 20000                       # the shape that broke round 4
    line = bench.compact_line(out, str(tmp_path / "bench_detail.json"))
    assert len(line) < 4096 and "\n" not in line
    j = json.loads(line)
    keys = list(j)
    assert keys[:12] == ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"]
    assert keys[12:16] == ["config", "roofline", "cpu_baseline", "verified"]
    assert j["roofline"]["frac"] == 0.1222 and j["roofline"]["kernel"] == "k_probe_apply" and j["roofline"]["traffic"] == 228329984
    assert j["roofline"]["kernel_ms"]["probe_apply"] == 0.07098 and j["roofline"]["algorithmic_bytes_per_launch"] == 69402625.6
    assert j["cpu_baseline"]["value"] == pytest.approx(11962425.7) and j["cpu_baseline"]["cores"] == 1 and j["cpu_baseline"]["kind"] == "port"
    assert j["verified"] == {"ok": True, "batches": 184, "winner_indices_compared": 19159209, "rows": 12298638, "table_digest": "a55ef666384fd8b3", "ranks_ok": 8}
    assert j["config"]["workload"].startswith("config 4 shape") and j["value"] == pytest.approx(12872332136.1) and j["ms_per_step"] == pytest.approx(0.077686)
    assert j["exchange"]["kind"] == "rccl" and j["exchange"]["refused"] == "direct" and "cannot map" in j["exchange"]["why"]
    assert j["scan_config3"]["100M"]["mask_frac"] == 0.7654 and j["scan_config3"]["100M"]["range10_ids_us"] == 225.7
    assert j["js_host"]["store_kept_entries_per_s"] == 698754.0 and j["detail"] == "bench_detail.json"
    assert j["scan_config3"]["100M"]["view_first_equals_after_merge_us_worst_of_4"] == 812.5 and j["scan_config3"]["100M"]["view_kept_current_by"] == "patch"
    # sections are given up from the end, never the contract: a record that cannot fit still yields a parseable short line
    huge = _full_out()
    huge["scan_config3"] = {"%dM" % i: huge["scan_config3"]["100M"] for i in range(60)}
    line2 = bench.compact_line(huge)
    j2 = json.loads(line2)
    assert len(line2) < 4096 and "scan_config3" not in j2 and j2["roofline"]["frac"] == 0.1222 and j2["cpu_baseline"]["value"] > 0 and j2["verified"]["ok"]


def test_emit_writes_the_full_record_beside_the_short_line(tmp_path, capsys):
    import bench
    out = _full_out()
    buf = io.StringIO()
    bench.emit(out, buf, str(tmp_path / "d.json"))
    assert buf.getvalue().count("\n") == 1 and len(buf.getvalue()) < 4097
    assert json.load(open(tmp_path / "d.json")) == out
    assert "bench detail: " in capsys.readouterr().err
