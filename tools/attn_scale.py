import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import attn_bench as ab
for (B, H) in ((1, 4), (1, 8), (1, 16), (2, 16), (4, 16), (8, 16)):
    ab.case(f"self B{B} H{H}", B, H, 1349, 1349)
