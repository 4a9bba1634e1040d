#!/usr/bin/env python3
"""Encoder forward at a FIXED token count cut into passage chunks: does keeping a chunk's
intermediates (chunk x 3072 x 2 B for FFN-up) inside the 256 MiB Infinity Cache beat one big batch?
Device ms per chunk are hipEvent times inside the library (mrag_encoder_last_timing), summed over the chunks;
ids live on the device, so nothing but the kernels is in the figure.
    python tools/enc_chunk_sweep.py [arch B S]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from mrag_amd.encoder import HipSentenceEncoder, ARCHS

arch = sys.argv[1] if len(sys.argv) > 1 else "bge-base"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
S = int(sys.argv[3]) if len(sys.argv) > 3 else 128
enc = HipSentenceEncoder.from_seed(arch, seed=0)
rng = np.random.default_rng(0)
ids = torch.from_numpy(rng.integers(1000, 30000, size=(B, S)).astype(np.int32)).cuda()
mask = torch.ones((B, S), dtype=torch.int32, device="cuda")
out = torch.empty((B, enc.spec.hidden), dtype=torch.float32, device="cuda")
a = ARCHS[arch]
flop_tok = 2 * a["layers"] * (4 * a["hidden"] ** 2 + 2 * a["hidden"] * a["intermediate"]) + 4 * S * a["hidden"] * a["layers"]
ref = None
for rnd in range(2):
    for chunk in (B, B // 2, B // 4, B // 8, B // 16, B // 32):
        if chunk < 16:
            continue
        tot = []
        for it in range(4):
            ms = 0.0
            for lo in range(0, B, chunk):
                enc.forward_device(ids[lo:lo + chunk], mask[lo:lo + chunk], out=out[lo:lo + chunk])
                torch.cuda.synchronize()
                ms += enc.last_timing_ms()
            tot.append(ms)
        t = float(np.median(tot[1:]))
        o = out.cpu().numpy()
        if ref is None:
            ref = o
        print(f"{arch} {B}x{S} tokens={B*S} chunk={chunk} passages ({chunk*S} tokens, FFN-up {chunk*S*a['intermediate']*2/2**20:.0f} MiB): "
              f"{t:.2f} ms = {B*S*flop_tok/t/1e9:.0f} TFLOP/s, max |d| vs one batch {np.abs(o-ref).max():.2e}", flush=True)
