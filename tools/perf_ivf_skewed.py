#!/usr/bin/env python3
"""The bench's `ivf_skewed` leg on its own (Zipf list sizes, overlapping clusters): recall and search ms vs nprobe."""
import sys, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import bench
r = bench.ivf_skewed_leg(torch.device("cuda", 0), 10)
print({k: v for k, v in r.items() if k != "points"})
for p in r["points"]:
    print("  ", p)
