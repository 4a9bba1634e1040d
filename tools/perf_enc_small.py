import sys, time; sys.path.insert(0,".")
import numpy as np
from mrag_amd.encoder import HipSentenceEncoder
for arch in ("minilm-l6","bge-base"):
    enc = HipSentenceEncoder.from_seed(arch, seed=0)
    rng = np.random.default_rng(0)
    for B,S in ((1,16),(1,64),(8,32),(50,32),(32,64),(128,64),(256,128),(2048,128)):
        ids = rng.integers(1000, 30000, size=(B, S)).astype(np.int32); mask = np.ones((B,S),dtype=np.int32)
        enc.forward(ids, mask)
        ts=[]
        for _ in range(10):
            t0=time.perf_counter(); enc.forward(ids, mask); ts.append(time.perf_counter()-t0)
        print(arch, B, S, "host call %.3f ms" % (np.median(ts)*1e3), "device %.3f" % enc.last_timing_ms(), flush=True)
