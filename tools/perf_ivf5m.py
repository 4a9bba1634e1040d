#!/usr/bin/env python3
"""C5 at full size on one GPU (5 M x 768, nlist 4096, nprobe 32, 10 k queries): search timing; run under
rocprofv3 --kernel-trace --stats for the per-kernel split.   python tools/perf_ivf5m.py [n_rows]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
r = bench.ivf_leg(torch.device("cuda", 0), 10, n=n)
print({k: v for k, v in r.items() if k not in ("note", "kernel")})
