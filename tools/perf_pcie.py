import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from mrag_amd.index import DenseIndex
d=768; ix=DenseIndex(d)
g=torch.Generator(device="cuda").manual_seed(1)
for i in range(4): ix.add(torch.randn(250000,d,device="cuda",generator=g))
q=torch.randn(10000,d,device="cuda",generator=g); qh=q.cpu().numpy()
for name,qq in (("device",q),("host",qh)):
    ix.search(qq,10); torch.cuda.synchronize()
    t=time.perf_counter()
    for _ in range(5):
        r=ix.search(qq,10)
        if name=="device": torch.cuda.synchronize()
    print(name, (time.perf_counter()-t)/5*1e3, "ms per 10k-query search")
