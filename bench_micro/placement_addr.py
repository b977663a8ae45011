import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bullet-js_amd"))
os.environ["BMX_PLACEMENT_DEBUG"] = "1"; os.environ["BMX_TABLE_PLACEMENT_TRIES"] = "10"; os.environ["BMX_TABLE_CONTIGUOUS"] = sys.argv[1] if len(sys.argv) > 1 else "5"
import bmx
for cap in (22_000_000 + 4_000_000, 22_000_000 + 4_000_000):
    print("capacity", cap, flush=True)
    e = bmx.Engine(cap); print(e.placement(), flush=True); e.close()
