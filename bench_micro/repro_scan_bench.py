"""runs bench.py's scan_bench alone (one index size), to look at a verification failure outside the full bench"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bullet-js_amd"))
import torch, bmx, bench, json
R = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
out = bench.scan_bench(bmx, torch.device("cuda", 0), R=R, reps=int(sys.argv[2]) if len(sys.argv) > 2 else 20)
print(json.dumps({k: (v if not isinstance(v, dict) else {kk: v[kk] for kk in ("us", "matches") if kk in v}) for k, v in out.items()}))
