import os, torch, torch.distributed as dist
dist.init_process_group("gloo")
r = dist.get_rank(); w = dist.get_world_size()
x = torch.full((4, 3), r, dtype=torch.int64, device="cuda")
out = torch.empty((w * 4, 3), dtype=torch.int64, device="cuda")
try:
    dist.all_gather_into_tensor(out, x)
    print(r, "cuda all_gather_into_tensor on gloo OK", out[:, 0].tolist())
except Exception as e:
    print(r, "FAILED:", repr(e)[:200])
dist.destroy_process_group()
