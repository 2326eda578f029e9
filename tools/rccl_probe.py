import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
x = torch.ones(1 << 20, dtype=torch.bfloat16, device="cuda")
w = dist.all_reduce(x, async_op=True); w.wait(); torch.cuda.synchronize()
dist.barrier(); print("rccl bf16 all_reduce ok", float(x.sum()))
dist.destroy_process_group()
