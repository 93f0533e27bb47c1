import os, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
t = torch.ones(4, device="cuda") * 3
dist.all_reduce(t); dist.barrier()
x = torch.empty(8, device="cuda"); dist.all_gather_into_tensor(x, torch.arange(8., device="cuda"))
torch.cuda.synchronize()
print("nccl ok", t.tolist(), x.sum().item(), torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else "")
dist.destroy_process_group()
