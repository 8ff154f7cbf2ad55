import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import subprocess
res = {}
for wl in ("coarse_b4_v5", "stress_b1_v10", "coarse_b1_v5"):
    for lg in (0, 2, 4, 5, 6, 8):
        v = 24 | (lg << 17)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ab_variants.py"), "--rounds", "2", "--iters", "40",
                              "--variants", str(v)], capture_output=True, text=True).stdout
        d = json.loads(out)[wl]
        res[f"{wl}_K{1 << lg}"] = d[f"nhwc_v{v}"]["median_us"]
print(json.dumps(res, indent=0))
