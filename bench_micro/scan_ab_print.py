import json,sys
for tag in ("two_pass","fused"):
    j=json.loads(open("gpurun_out/r03/scan_%s.json"%tag).read().strip().splitlines()[-1])
    sc=j["scan_config3"]
    print("==",tag, "verified", j.get("verified",{}).get("ok"), j.get("verified",{}).get("scans"))
    def walk(d,pre=""):
        for k,v in d.items():
            if isinstance(v,dict): walk(v,pre+k+".")
            elif isinstance(v,(int,float)) and ("us" in k or "frac" in k): print("   ",pre+k,v)
    walk(sc)
