import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bullet-js_amd"))
os.environ["BMX_PLACEMENT_DEBUG"] = "1"; os.environ["BMX_TABLE_PLACEMENT_TRIES"] = "12"
import bmx
for cap in (22_000_000 + 4_000_000, 11_000_000, 44_000_000):
    print("capacity", cap, flush=True)
    e = bmx.Engine(cap); print(e.placement(), flush=True); e.close()
