#!/usr/bin/env python3
"""Headline benchmark: queries/sec of brute-force cosine top-10 over the 1M x 768 fp16 corpus
(BASELINE.json metric; workload = config C4: a 10 000-query batch, corpus row-sharded over the
N GPUs of one node, ONE RCCL all-gather of the per-shard packed (score, id) partial top-k + merge on
the device).

    python bench.py [--gpus N] [--steps K] [--warmup W]

``--gpus N`` with N > 1 launches its own ranks (``python -m torch.distributed.run``, one per GPU, as a
child process started before anything touches the GPU) unless it already runs under such a launcher
(WORLD_SIZE set), so both ``python bench.py --gpus 8`` and the driver's explicit torchrun line work.

A "step" = one pass of the hot path over one query batch: K1 (normalise + fp16 round of the
batch) -> K2 (fused MFMA similarity + top-k over this rank's rows) -> K4 (candidate merge) ->
[N>1: all-gather + device merge] -> results on the host.  Corpus and query batch are resident in HBM
before the timed region.  Total work is fixed as N grows ("scaling": "strong").

Rank 0 prints ONE JSON line.  Besides the contract keys:
  median_ms_per_step, value_incl_h2d (fp32 queries handed over in pinned HOST memory every step),
  value_incl_h2d_pipelined (the same with the copy of batch i+1 on a second stream under the search of batch i)
  roofline             dominant kernel (K2) at the bench workload: algorithmic FLOP = 2*Q*N_local*D per
                       launch over the kernel's hipEvent time measured inside the library on the launch
                       stream; held_clock_ghz = in-kernel s_memtime / s_memrealtime reading
  north_star_roofline  the same kernel at BASELINE's north-star GEMM shape 1 000 x 1 000 000 x 768 (N=1)
  c2_roofline          ... and at config C2, 1 000 x 100 000 x 768 (N=1)
  online_roofline      1 query per call (the reference's own usage): streaming kernel, HBM-bound
  ivf_roofline         config C5 on one GPU's share (625 000 x 768, nlist 4096, nprobe 32, 10 k queries):
                       list scan (score-segment scan + per-query select kernels), bytes = rows streamed x 768 x 2 (N=1);
                       its cpu_baseline = the numpy IVF restatement on a bounded query sample
  ivf_roofline_5m      the same over the WHOLE 5 M x 768 corpus on one GPU (7.7 GB; C5's N = 1 point)
  ivf_skewed           recall@10 and search ms vs nprobe on Zipf-sized, overlapping clusters (not the best case)
  ingest_from_text     C3's 64 k passages, text -> tokens -> bge-base forward -> index, passages/s
  step_phases_ms       per-phase device ms of a step (local search, pack, all-gather, merge, D2H); N > 1: two_half_pipeline
  encoder_roofline     config C3 shape (bge-base, 2048 passages x 128 tokens): device ms of the forward (N=1)
  cpu_baseline         N=1: the numpy restatement (oracle.dense_search.brute_force_topk_f32: sgemm + top-k on
                       the same fp16-rounded rows) timed on this host's cores on a bounded query sample
  recall_at_10         N=1: GPU ids vs the fp64 oracle on 1 000 of the queries
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

N_CORPUS = 1_000_000
DIM = 768
N_QUERIES = 10_000
TOP_K = 10
CHUNK = 65536            # data is generated per 65536-row chunk, seed 1234 + chunk: same corpus for every N
PEAK_TFLOPS_F16 = 2500.0  # dense fp16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0     # HBM3E spec, same guide


def gen_chunk(c: int, dim: int, device):
    import torch
    g = torch.Generator(device=device).manual_seed(1234 + c)
    return torch.randn(CHUNK, dim, device=device, generator=g, dtype=torch.float32)


def build_shard(ix, lo: int, hi: int, dim: int, device):
    c = lo // CHUNK
    while c * CHUNK < hi:
        rows = gen_chunk(c, dim, device)
        a, b = max(lo, c * CHUNK), min(hi, (c + 1) * CHUNK)
        ix.add(rows[a - c * CHUNK:b - c * CHUNK])
        c += 1


def self_launch(args) -> int:
    """Start one rank per GPU as a CHILD process (no exec: nothing here has touched the GPU, and a process
    that has must never be replaced) and relay its output; rank 0 of the child job prints the JSON line."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    return subprocess.run(cmd, env=env).returncode


def k2_roofline(ix, queries, k, n_rows, d, iters=20, warm=3, label=""):
    """K2 time at another shape of the same kernel (hipEvents inside the library), median of ``iters``."""
    for _ in range(warm):
        ix.search(queries, k)
    ms = []
    for _ in range(iters):
        ix.search(queries, k)
        ms.append(ix.last_timing_ms())
    k2 = float(np.median([m[0] for m in ms]))
    tot = float(np.median([m[1] for m in ms]))
    flop = 2.0 * queries.shape[0] * n_rows * d
    ach = flop / (k2 * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": "bf_gemm_topk_kernel", "workload": label, "kernel_ms": k2, "search_ms": tot,
            "achieved": ach, "peak": PEAK_TFLOPS_F16, "unit": "TFLOP/s", "frac": ach / PEAK_TFLOPS_F16,
            "flop_per_launch": flop, "timing": f"median of {iters} launches"}


def ivf_leg(device, k, n=625_000, with_cpu=False):
    """C5: n x 768 clustered rows (625 000 = one GPU's share of the 5 M corpus; 5 000 000 = the whole corpus on one
    GPU, the N = 1 point of C5's scaling curve), nlist 4096 (device k-means), nprobe 32, 10 k queries."""
    import torch
    from mrag_amd.index import IVFFlatIndex, DenseIndex
    d, nlist, nprobe, nq = DIM, 4096, 32, N_QUERIES
    g = torch.Generator(device=device).manual_seed(1)
    cent = torch.randn(4096, d, device=device, generator=g)
    ix, bf = IVFFlatIndex(d, nlist, device=device.index), DenseIndex(d, device=device.index)
    train = cent[torch.randint(0, 4096, (100_000,), device=device, generator=g)] + 0.3 * torch.randn(100_000, d, device=device, generator=g)
    ix.train(train, iters=5, seed=1)
    for lo in range(0, n, 125_000):
        rows = cent[torch.randint(0, 4096, (125_000,), device=device, generator=g)] + 0.3 * torch.randn(125_000, d, device=device, generator=g)
        ix.add(rows)
        bf.add(rows)
    pick = torch.randint(0, 4096, (nq,), device=device, generator=g)
    q = cent[pick] + 0.3 * torch.randn(nq, d, device=device, generator=g)
    ix.search(q, k, nprobe)
    scan, tot = [], []
    for _ in range(5):
        sc, ids = ix.search(q, k, nprobe)
        t = ix.last_timing()
        scan.append(t["scan_ms"]); tot.append(t["total_ms"])
    bs, bi = bf.search(q, k)
    torch.cuda.synchronize()
    bf_ms = bf.last_timing_ms()[1]
    ids, bi = ids.cpu().numpy(), bi.cpu().numpy()
    rec = float(np.mean([len(set(a) & set(b)) / k for a, b in zip(ids, bi)]))
    scan_ms, tot_ms = float(np.median(scan)), float(np.median(tot))
    nbytes = t["scanned_rows"] * d * 2.0
    ach = nbytes / (scan_ms * 1e-3) / 1e9
    cpu = None
    if with_cpu:
        # this leg's CPU baseline: the numpy restatement of IVF-flat (oracle.dense_search.ivf_search: fp64 probe
        # selection + exact scan of the probed lists) on the GPU's own centroids / assignments, bounded query sample
        from oracle import dense_search as ods
        c16 = np.empty((n, d), dtype=np.float16)
        for lo in range(0, n, 125_000):
            c16[lo:lo + 125_000] = bf.rows(lo, min(125_000, n - lo))
        cen, asg = ix.centroids().astype(np.float16), ix.assignments().astype(np.int64)
        q16 = ods.normalize_round(q[:256].cpu().numpy())
        t1 = time.perf_counter(); ods.ivf_search(q16[:8], c16, cen, asg, nprobe, k); probe = time.perf_counter() - t1
        ns = int(min(256, max(8, 10.0 / max(probe / 8, 1e-6))))
        t1 = time.perf_counter(); ods.ivf_search(q16[:ns], c16, cen, asg, nprobe, k); cpu_s = time.perf_counter() - t1
        cpu = {"value": ns / cpu_s, "unit": "queries/s", "cores": 1, "kind": "port",
               "sample": f"{ns} of the {nq} queries, numpy fp64 IVF-flat (probe selection + exact scan of {nprobe} lists) over the "
                         f"same {n} rows / centroids / assignments, {cpu_s:.1f} s (per-query loop: one core + BLAS gemv)"}
        del c16
    ix.close(); bf.close()
    return {"bound": "hbm", "cpu_baseline": cpu, "brute_force_search_ms": bf_ms, "kernel": "ivfs_scan_kernel + ivfs_select_lists_kernel (IVF list scan: fp32 score segments, then the k best per query)",
            "workload": f"C5 {'per-GPU share' if n < 5_000_000 else 'whole corpus on one GPU'}: {n} x {d} fp16, nlist {nlist}, nprobe {nprobe}, {nq} queries, k={k}, clustered rows",
            "kernel_ms": scan_ms, "search_ms": tot_ms, "queries_per_s": nq / (tot_ms * 1e-3),
            "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS,
            "bytes_per_launch": nbytes, "rows_streamed": t["scanned_rows"], "workgroups": t["n_wg"],
            "recall_at_10_vs_brute_force": rec,
            "note": "algorithmic bytes = list rows streamed once per (list, <=128 probing queries) group (a list is cut into 128-row scan "
                    "descriptors); kernel_ms covers the scan AND the selection. The scan also moves the gathered query rows (nq x nprobe x "
                    "1.5 KB from the fabric) and 4 B per (query, row) score out and back: about twice the algorithmic bytes "
                    "(counter: profiles/r03*_counters.json, ivfs_scan_kernel traffic per launch)"}


def ivf_skewed_leg(device, k):
    """the clustered C5 data above (centroid + 0.3 N(0,I) in 768 dimensions) is trivially separable and its
    lists are balanced -- the best case.  This leg: Zipf-like cluster popularity (list sizes spread over ~100x) and
    overlapping clusters (sigma = 1.0: noise norm = centroid norm), queries = a corpus row + 0.7 N(0,I); recall@10
    against exact brute force and the search time as nprobe grows."""
    import torch
    from mrag_amd.index import IVFFlatIndex, DenseIndex
    n, d, nlist, nq = 625_000, DIM, 4096, N_QUERIES
    g = torch.Generator(device=device).manual_seed(7)
    cent = torch.randn(nlist, d, device=device, generator=g)
    pop = 1.0 / torch.arange(1, nlist + 1, device=device, dtype=torch.float32) ** 0.8
    ix, bf = IVFFlatIndex(d, nlist, device=device.index), DenseIndex(d, device=device.index)

    def draw(m):
        c = torch.multinomial(pop, m, replacement=True, generator=g)
        return cent[c] + 1.0 * torch.randn(m, d, device=device, generator=g)
    ix.train(draw(100_000), iters=5, seed=1)
    keep = []
    for lo in range(0, n, 125_000):
        rows = draw(125_000)
        ix.add(rows); bf.add(rows)
        keep.append(rows[:2000].clone())
    pool = torch.cat(keep)
    q = pool[torch.randint(0, pool.shape[0], (nq,), device=device, generator=g)] + 0.7 * torch.randn(nq, d, device=device, generator=g)
    counts = np.bincount(ix.assignments(), minlength=nlist)
    bs, bi = bf.search(q, k)
    torch.cuda.synchronize()
    bi = bi.cpu().numpy()
    pts = []
    for nprobe in (1, 4, 8, 32, 128):
        ix.search(q, k, nprobe)
        ms = []
        for _ in range(3):
            sc, ids = ix.search(q, k, nprobe)
            ms.append(ix.last_timing()["total_ms"])
        ids = ids.cpu().numpy()
        rec = float(np.mean([len(set(a) & set(b)) / k for a, b in zip(ids, bi)]))
        pts.append({"nprobe": nprobe, "recall_at_10": rec, "search_ms": float(np.median(ms)), "rows_streamed": ix.last_timing()["scanned_rows"]})
    ix.close(); bf.close()
    return {"workload": f"{n} x {d} fp16, nlist {nlist}, Zipf(0.8) cluster popularity, sigma 1.0 (overlapping clusters), {nq} queries, k={k}",
            "list_rows_max": int(counts.max()), "list_rows_median": float(np.median(counts)), "list_rows_min": int(counts.min()),
            "points": pts}


def encoder_leg(device, with_cpu=False):
    """C3 shape: bge-base (12 x 768, 12 heads, FFN 3072, 30 522-row vocabulary), 2048 passages x 128 tokens."""
    from mrag_amd.encoder import HipSentenceEncoder, ARCHS
    arch, B, S = "bge-base", 2048, 128
    enc = HipSentenceEncoder.from_seed(arch, seed=0, device=device.index)
    rng = np.random.default_rng(0)
    ids = rng.integers(1000, 30000, size=(B, S)).astype(np.int32)
    mask = np.ones((B, S), dtype=np.int32)
    enc.forward(ids, mask)
    dev_ms, wall_ms = [], []
    for _ in range(5):
        t0 = time.perf_counter()
        enc.forward(ids, mask)
        wall_ms.append((time.perf_counter() - t0) * 1e3)
        dev_ms.append(enc.last_timing_ms())
    a = ARCHS[arch]
    flop_tok = 2.0 * a["layers"] * (4 * a["hidden"] ** 2 + 2 * a["hidden"] * a["intermediate"]) + 4.0 * S * a["hidden"] * a["layers"]
    ms = float(np.median(dev_ms))
    ach = B * S * flop_tok / (ms * 1e-3) / 1e12
    cpu = None
    if with_cpu:
        # this leg's CPU baseline: the numpy fp64 restatement of the BERT forward (oracle.encoder.forward, pinned to HF
        # BertModel by F6 / F9) on a few of the same passages with the same seeded weights
        from oracle import encoder as oenc
        from mrag_amd.encoder import seeded_weights, EncoderSpec
        w = seeded_weights(EncoderSpec(**a), 0)
        spec = {k_: a[k_] for k_ in ("vocab_size", "hidden", "layers", "heads", "intermediate", "max_position", "type_vocab_size", "layer_norm_eps")}
        nb = 4
        t1 = time.perf_counter(); ref = oenc.forward(spec, w, ids[:nb].astype(np.int64), mask[:nb].astype(np.int64), pool=a["pool"]); cpu_s = time.perf_counter() - t1
        got = enc.forward(ids[:nb], mask[:nb])
        cpu = {"value": nb / cpu_s, "unit": "passages/s", "cores": os.cpu_count(), "kind": "port",
               "sample": f"{nb} of the {B} passages x {S} tokens, numpy fp64 BERT forward (BLAS threads), {cpu_s:.1f} s",
               "max_abs_diff_gpu_vs_cpu": float(np.abs(got - ref).max())}
        # HF BertModel on this host's cores via torch, same seeded weights and ids (fp32; what
        # sentence-transformers would run on the CPU) -- only if transformers is importable on this box
        try:
            import torch
            import transformers as tr
            cfg = tr.BertConfig(vocab_size=spec["vocab_size"], hidden_size=spec["hidden"], num_hidden_layers=spec["layers"],
                                num_attention_heads=spec["heads"], intermediate_size=spec["intermediate"],
                                max_position_embeddings=spec["max_position"], type_vocab_size=spec["type_vocab_size"],
                                layer_norm_eps=spec["layer_norm_eps"], hidden_act="gelu", hidden_dropout_prob=0.0,
                                attention_probs_dropout_prob=0.0)
            m = tr.BertModel(cfg, add_pooling_layer=False).eval()
            sd = m.state_dict()
            for k_, v_ in w.items():
                if k_ in sd and tuple(sd[k_].shape) == v_.shape:
                    sd[k_] = torch.from_numpy(np.ascontiguousarray(v_, dtype=np.float32))
            m.load_state_dict(sd, strict=False)
            nh = 64
            ti, tm = torch.from_numpy(ids[:nh]).long(), torch.from_numpy(mask[:nh]).long()
            with torch.no_grad():
                m(input_ids=ti[:8], attention_mask=tm[:8])
                t1 = time.perf_counter()
                h = m(input_ids=ti, attention_mask=tm).last_hidden_state
                hf_s = time.perf_counter() - t1
            e = h[:, 0] if a["pool"] == "cls" else (h * tm[:, :, None]).sum(1) / tm.sum(1, keepdim=True).clamp(min=1e-9)
            e = torch.nn.functional.normalize(e, dim=1).numpy()
            cpu["hf_torch"] = {"value": nh / hf_s, "unit": "passages/s", "torch_threads": torch.get_num_threads(),
                               "sample": f"{nh} of the {B} passages x {S} tokens, transformers.BertModel fp32 on the CPU, {hf_s:.1f} s",
                               "max_abs_diff_gpu_vs_hf": float(np.abs(enc.forward(ids[:nh], mask[:nh]) - e).max())}
        except Exception as ex:      # transformers absent on this box: the numpy port above stays the baseline
            cpu["hf_torch"] = {"skipped": repr(ex)}
    enc.close()
    return {"bound": "mfma", "cpu_baseline": cpu, "kernel": "enc_gemm256_kernel (LayerNorm folded into its epilogues) + enc_attention_s128_kernel (whole forward)",
            "workload": f"C3 shape: {arch}, {B} passages x {S} tokens, seeded weights",
            "device_ms": ms, "host_call_ms": float(np.median(wall_ms)), "passages_per_s": B / (ms * 1e-3),
            "achieved": ach, "peak": PEAK_TFLOPS_F16, "unit": "TFLOP/s", "frac": ach / PEAK_TFLOPS_F16,
            "flop_per_token": flop_tok}


def synthetic_sentences(n: int, seed: int = 0):
    """HotpotQA-shaped passages (one sentence per corpus row, my_code/ingest_hotpotqa.py:73-81): 3 .. 80 words, mean ~ 25."""
    rng = np.random.default_rng(seed)
    vocab = np.array([f"w{i}" for i in range(20000)])
    lens = np.clip(rng.normal(25, 10, size=n).astype(np.int64), 3, 80)
    words = vocab[rng.integers(0, len(vocab), size=int(lens.sum()))]
    out, at = [], 0
    for L in lens:
        out.append(" ".join(words[at:at + L]))
        at += L
    return out


def ingest_leg(device, n=65_536):
    """texts -> index: C3's 64 k passages through the provider's bulk device form (tokenise on the host,
    overlapped; bge-base forward in large length-sorted batches; rows device -> device into DenseIndex.add / K1)."""
    import torch
    from mrag_amd.provider import HipEmbeddingProvider
    from mrag_amd.index import DenseIndex
    texts = synthetic_sentences(n)
    prov = HipEmbeddingProvider(arch="bge-base", seed=0, device=device.index)
    enc = prov.encoder
    t0 = time.perf_counter()
    ids, mask = enc.tokenize(texts[:8192])
    tok_s = (time.perf_counter() - t0) * n / 8192
    prov.embed_device(texts[:4096])                      # warm-up (kernel load, buffers)
    ix = DenseIndex(enc.spec.hidden, device=device.index)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    emb = prov.embed_device(texts, batch_size=4096)
    ix.add(emb)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # the router-shaped path on a bounded sample: embed() 256 texts at a time -> python float lists -> numpy -> host add
    m = 4096
    t0 = time.perf_counter()
    for lo in range(0, m, 256):
        v = np.asarray(prov.embed(model="x", texts=texts[lo:lo + 256])["vectors"], dtype=np.float32)
    rt = time.perf_counter() - t0
    ix.close()
    return {"workload": f"{n} synthetic HotpotQA-shaped sentences (mean {int(mask.sum() / 8192)} tokens) -> bge-base (seeded) -> DenseIndex",
            "passages_per_s_from_text": n / dt, "seconds": dt, "tokenise_seconds_if_serial": tok_s,
            "router_path_passages_per_s": m / rt,
            "note": "bulk: provider.embed_device (4096-passage length-sorted batches, tokeniser thread overlapped) + DenseIndex.add "
                    "device to device; router path: provider.embed 256 texts per call (lists of Python floats), timed on 4096 passages"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--n-corpus", type=int, default=N_CORPUS)
    ap.add_argument("--n-queries", type=int, default=N_QUERIES)
    ap.add_argument("--dim", type=int, default=DIM)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline / recall legs")
    ap.add_argument("--no-extras", action="store_true", help="skip the north-star / C2 / IVF / encoder sub-objects")
    ap.add_argument("--skip-legs", default="", help="comma list of sub-objects to skip (e.g. ivf_roofline_5m,ingest_from_text): counter passes")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py --gpus {args.gpus} is running under a launcher with WORLD_SIZE={world}")
    n_dev = torch.cuda.device_count()
    # REHEARSAL switch (never what the driver runs): MRAG_BENCH_REHEARSE=1 puts every rank on cuda:0 and moves the exchange over
    # gloo (which carries CUDA tensors too), so that the N > 1 code path of this file can be executed end to end on a one-GPU box;
    # the line it prints says so in `config.rehearsal` and is not a scaling measurement
    rehearse = os.environ.get("MRAG_BENCH_REHEARSE", "") not in ("", "0")
    if rehearse:
        local_rank = 0
    if n_dev < (1 if rehearse else world) or local_rank >= n_dev:
        raise SystemExit(f"bench.py --gpus {world} needs {world} visible MI355X devices, this box has {n_dev}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no CUDA/HIP device visible")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    from mrag_amd.sharded import ShardedDenseIndex

    n, nq, d, k = args.n_corpus, args.n_queries, args.dim, TOP_K
    sh = ShardedDenseIndex(d, n, rank, world, device=local_rank)
    build_shard(sh.index, sh.lo, sh.hi, d, device)
    gq = torch.Generator(device=device).manual_seed(5678)
    queries = torch.randn(nq, d, device=device, generator=gq, dtype=torch.float32)
    q_host = queries.cpu().pin_memory()                 # the boundary's host-buffer form (value_incl_h2d)
    q_stage = torch.empty_like(queries)
    torch.cuda.synchronize()

    def step():
        return sh.search(queries, k)

    def step_h2d():
        q_stage.copy_(q_host, non_blocking=True)         # 30.7 MB of fp32 queries over PCIe, same stream as the search
        return sh.search(q_stage, k)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    gemm_ms, step_ms, phase_ms = [], [], []
    fence()
    t0 = time.perf_counter()
    t_prev = t0
    for _ in range(args.steps):
        sc, ids = step()                                 # returns with the merged results on the host
        t_now = time.perf_counter()
        step_ms.append((t_now - t_prev) * 1e3)
        t_prev = t_now
        gemm_ms.append(sh.index.last_timing_ms()[0])
        phase_ms.append(dict(sh.last_phases))            # (CUDA events recorded during the step, read after its one sync)
    fence()
    elapsed = time.perf_counter() - t0
    sc, ids = sc.copy(), ids.copy()
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # the same steps with the queries handed over in HOST memory (PCIe-inclusive rate; never `value`)
    n_h2d = max(5, min(args.steps, 20))
    for _ in range(2):
        step_h2d()
    fence()
    t1 = time.perf_counter()
    for _ in range(n_h2d):
        step_h2d()
    fence()
    elapsed_h2d = time.perf_counter() - t1
    if world > 1:
        tmax = torch.tensor([elapsed_h2d], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed_h2d = float(tmax.item())

    # ... and pipelined, as a server would: the H2D copy of batch i + 1 runs on a second stream under the search of batch i
    # (two staging buffers; every batch still crosses PCIe once)
    copy_stream = torch.cuda.Stream(device=device)
    q_stages = [torch.empty_like(queries), torch.empty_like(queries)]
    ready = [torch.cuda.Event(), torch.cuda.Event()]

    def enqueue_copy(j):
        copy_stream.wait_stream(torch.cuda.current_stream(device))      # the search that last read this buffer is queued ahead
        with torch.cuda.stream(copy_stream):
            q_stages[j].copy_(q_host, non_blocking=True)
            ready[j].record(copy_stream)

    enqueue_copy(0)
    for i in range(2):
        enqueue_copy((i + 1) & 1)
        torch.cuda.current_stream(device).wait_event(ready[i & 1])
        sh.search(q_stages[i & 1], k)
    fence()
    t2 = time.perf_counter()
    enqueue_copy(0)
    for i in range(n_h2d):
        if i + 1 < n_h2d:
            enqueue_copy((i + 1) & 1)
        torch.cuda.current_stream(device).wait_event(ready[i & 1])
        sh.search(q_stages[i & 1], k)
    fence()
    elapsed_h2d_pipe = time.perf_counter() - t2
    if world > 1:
        tmax = torch.tensor([elapsed_h2d_pipe], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed_h2d_pipe = float(tmax.item())

    # N > 1: the same steps with the batch cut in two halves, the all-gather of half A in flight under the local search of half B
    elapsed_pipe, pipe_phase_ms = None, []
    if world > 1:
        for _ in range(2):
            sh.search(queries, k, pipeline=True)
        fence()
        t3 = time.perf_counter()
        for _ in range(n_h2d):
            sh.search(queries, k, pipeline=True)
            pipe_phase_ms.append(dict(sh.last_phases))
        fence()
        elapsed_pipe = time.perf_counter() - t3
        tmax = torch.tensor([elapsed_pipe], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed_pipe = float(tmax.item())

    # held clock under K2: one more step with the in-kernel stamps on
    clock_ghz = None
    try:
        sh.index.measure_clock(True)
        step()
        clock_ghz = sh.index.last_clock_ghz()
    except Exception:
        clock_ghz = None
    sh.index.measure_clock(False)

    # the online regime (one query per call = the reference's own usage): K2s, HBM-bound
    online_ms, online_search_ms, online_pool_ms = [], [], []
    for i in range(8):
        sh.index.search(queries[:1], k)
        if i > 1:
            g_ms, t_ms = sh.index.last_timing_ms()
            online_ms.append(g_ms)
            online_search_ms.append(t_ms)
    for i in range(6):                      # the reference's dense pool: 200 candidates per question (settings.yaml:101-102)
        sh.index.search(queries[:1], 200)
        if i > 1:
            online_pool_ms.append(sh.index.last_timing_ms()[1])
    fence()

    out = None
    if rank == 0:
        n_local = sh.hi - sh.lo
        k2_ms = float(np.mean(gemm_ms))
        flop = 2.0 * nq * n_local * d
        achieved = flop / (k2_ms * 1e-3) / 1e12
        out = {
            "metric": "queries/sec (brute-force cosine top-10 over the 1M x 768 fp16 corpus)",
            "value": nq * args.steps / elapsed,
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f16",
            "data": "synthetic",
            "config": {"workload": "C4: 10k-query batch x 1M x 768 fp16 corpus, k=10, corpus row-sharded across GPUs, "
                                   "one all-gather of packed (score, id) partial top-k + device merge",
                       "n_queries": nq, "n_corpus": n, "dim": d, "k": k,
                       "rows_per_gpu": n_local, "parallelism": f"row-shard x{world}",
                       **({"rehearsal": "all ranks on ONE GPU, exchange over gloo: exercises the N > 1 code path, not a scaling number"} if rehearse else {})},
            "median_ms_per_step": float(np.median(step_ms)),
            "value_incl_h2d": nq * n_h2d / elapsed_h2d,
            "value_incl_h2d_pipelined": nq * n_h2d / elapsed_h2d_pipe,
            "value_incl_h2d_note": f"{n_h2d} steps with the fp32 query batch in pinned host memory (H2D {nq * d * 4 / 1e6:.1f} MB "
                                   "per step on the search stream) and results on the host; `value` has the batch resident in HBM",
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_TFLOPS_F16, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_TFLOPS_F16, "traffic": None,
                         "kernel": "bf_gemm_topk_kernel", "kernel_ms": k2_ms, "kernel_ms_median": float(np.median(gemm_ms)),
                         "kernel_ms_min": float(np.min(gemm_ms)), "flop_per_launch": flop, "held_clock_ghz": clock_ghz,
                         "timing": f"hipEvents around the launch, mean over the {args.steps} timed steps"},
            "cpu_baseline": None,
            "recall_at_10": None,
            # rank 0's per-phase device ms of a step (median over the timed steps): local search (K1 + K2 + K4), pack,
            # all-gather, unpack + merge, result D2H -- the Amdahl terms of SURVEY 8e next to the one number above
            "step_phases_ms": {kk_: float(np.median([p_.get(kk_, 0.0) for p_ in phase_ms])) for kk_ in
                               ("local_ms", "pack_ms", "gather_ms", "merge_ms", "d2h_ms")},
        }
        if elapsed_pipe is not None:
            out["two_half_pipeline"] = {"ms_per_step": elapsed_pipe / n_h2d * 1e3, "value": nq * n_h2d / elapsed_pipe,
                                        "phases_ms": {kk_: float(np.median([p_.get(kk_, 0.0) for p_ in pipe_phase_ms])) for kk_ in
                                                      ("local_ms", "pack_ms", "gather_ms", "merge_ms", "d2h_ms")},
                                        "note": "batch cut in two halves, all-gather of half A (async_op) under the local search of half B; "
                                                "gather_ms = the part of the collectives the local search did not cover; `value` above is the one-shot step"}
        on_ms = float(np.median(online_ms))
        on_bytes = (sh.hi - sh.lo) * d * 2.0 + d * 2.0
        out["online_roofline"] = {"bound": "hbm", "kernel": "bf_stream_topk_kernel", "workload": "1 query x this rank's rows, k=10",
                                  "kernel_ms": on_ms, "achieved": on_bytes / (on_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                  "frac": on_bytes / (on_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "bytes_per_launch": on_bytes,
                                  "search_ms": float(np.median(online_search_ms)),
                                  "search_ms_k200": float(np.median(online_pool_ms)),
                                  "note": "search_ms = K1 + streaming kernel + merge on the device, one question; k200 = the reference's 200-candidate pool"}
        # HBM-side traffic of K2 comes from separate rocprofv3 --pmc passes of this same command
        # (FETCH_SIZE / WRITE_SIZE, gfx950 correction applied; profiles/r*_pmc_traffic.json, tools/profile_round.sh)
        pmcs = sorted((ROOT / "profiles").glob("r*_pmc_traffic.json"))
        if world == 1 and (n, nq, d) == (N_CORPUS, N_QUERIES, DIM) and pmcs:
            try:
                pmc = pmcs[-1]   # the latest round's counters (tools/profile_round.sh)
                out["roofline"]["traffic"] = json.loads(pmc.read_text())["kernels"]["bf_gemm_topk"]["traffic_bytes_per_launch"]
                out["roofline"]["traffic_note"] = f"bytes/launch, L2<->fabric requests (Infinity-Cache hits included); profiles/{pmc.name}"
            except Exception:
                pass
        if world == 1 and not args.no_extras:
            # BASELINE's north-star GEMM shape and config C2 on the same kernel
            out["north_star_roofline"] = k2_roofline(sh.index, queries[:1000], k, n_local, d,
                                                     label=f"north star: 1 000 x {n_local} x {d} fp16, k={k}")
            from mrag_amd.index import DenseIndex
            c2 = DenseIndex(d, device=local_rank)
            build_shard(c2, 0, 100_000, d, device)
            out["c2_roofline"] = k2_roofline(c2, queries[:1000], k, 100_000, d, label=f"C2: 1 000 x 100 000 x {d} fp16, k={k}")
            c2.close()
        if world == 1 and not args.no_cpu:
            from oracle import dense_search as ods
            c32 = sh.index.rows()                                  # the stored fp16 bits, as fp32
            q16 = ods.normalize_round(queries.cpu().numpy())       # same arithmetic as K1
            # recall / parity vs the fp64 oracle on 1 000 of the queries (blocked fp64 GEMM over the full corpus)
            sub = np.arange(0, nq, max(1, nq // 1000))[:1000]
            rv, ri = ods.brute_force_topk(q16[sub], c32, k, block=65536)
            out["recall_at_10"] = ods.recall_at_k(ids[sub], ri)
            out["recall_queries"] = int(len(sub))
            out["max_abs_score_err"] = float(np.max(np.abs(sc[sub] - rv)))
            strict, bad = ods.gap_aware_id_match(ids[sub], sc[sub], ri, rv, tol=1e-5)
            out["id_mismatches_outside_near_ties"] = int(bad)
            # CPU baseline: bounded sample, ~15 s of sgemm + top-k over the FULL corpus
            threads = torch.get_num_threads()
            q32 = q16.astype(np.float32)
            ods.brute_force_topk_f32(q32[:64], c32, k)                       # warm the BLAS threads / page in the corpus
            t1 = time.perf_counter(); ods.brute_force_topk_f32(q32[:256], c32, k); probe = time.perf_counter() - t1
            ns = int(min(nq, max(256, 15.0 / max(probe / 256, 1e-6))))     # ~15 s of CPU work
            t1 = time.perf_counter(); ods.brute_force_topk_f32(q32[:ns], c32, k); cpu_s = time.perf_counter() - t1
            # the reference's own arithmetic for this path, restated: DenseReranker._cosine per candidate in
            # pure Python (retrieval_backend.py:192-197,245), one query over a bounded candidate sample,
            # extrapolated to the full corpus (context only: the numpy port above is the stronger baseline)
            from oracle import ref_semantics as ors
            n_c = 4000
            ql = q32[0].tolist()
            cl = c32[:n_c].tolist()
            t1 = time.perf_counter()
            for row in cl:
                ors.cosine(ql, row)
            py_s = time.perf_counter() - t1
            ref_style = {"value": 1.0 / (py_s / n_c * n), "unit": "queries/s", "cores": 1,
                         "sample": f"1 query x {n_c} candidates in {py_s:.2f} s, scaled to {n} candidates "
                                   "(per-candidate pure-Python cosine, the reference's loop)"}
            out["cpu_baseline"] = {"value": ns / cpu_s, "unit": "queries/s", "cores": os.cpu_count(),
                                   "kind": "port", "blas_threads": threads,
                                   "sample": f"{ns} of the {nq} queries against the full {n} x {d} corpus "
                                             f"(numpy sgemm + argpartition top-{k}, {cpu_s:.1f} s)",
                                   "reference_style_python": ref_style}
            del c32
        if world == 1 and not args.no_extras:
            sh.index.close()                                       # free the 1.5 GB corpus before the other configs
            for name, leg in (("ivf_roofline", lambda: ivf_leg(device, k, with_cpu=not args.no_cpu)),
                              ("ivf_roofline_5m", lambda: ivf_leg(device, k, n=5_000_000)),
                              ("ivf_skewed", lambda: ivf_skewed_leg(device, k)),
                              ("encoder_roofline", lambda: encoder_leg(device, with_cpu=not args.no_cpu)),
                              ("ingest_from_text", lambda: ingest_leg(device))):
                if name in args.skip_legs.split(","):
                    continue
                try:
                    out[name] = leg()
                except Exception as e:                             # a sub-object must never cost the headline line
                    out[name] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
