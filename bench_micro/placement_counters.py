"""One context whose table is allocated BMX_TABLE_PLACEMENT_TRIES times (run with 8): every candidate gets four launches of k_placement_probe. Under
rocprofv3 --pmc ... --kernel-trace the per-dispatch counters of those launches, in dispatch order, are candidates 0..7 x repetitions 0..3; the engine's own
probe times (BMX_PLACEMENT_DEBUG=1) go to stderr. profiles/make_placement_counters.py joins the passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bullet-js_amd"))
import bmx
e = bmx.Engine(capacity_rows=int(sys.argv[1]) if len(sys.argv) > 1 else 22_000_000, device=0)
print("placement:", e.placement(), flush=True)
e.close()
