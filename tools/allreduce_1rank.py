import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29551"); os.environ.setdefault("RANK","0"); os.environ.setdefault("WORLD_SIZE","1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
x = torch.randn(86_000_000, device="cuda")
for bucket_mb in (64, 512):
    step = bucket_mb*1024*1024//4
    for _ in range(3):
        works=[dist.all_reduce(x[i:i+step], async_op=True) for i in range(0, x.numel(), step)]
        for w in works: w.wait()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(10):
        works=[dist.all_reduce(x[i:i+step], async_op=True) for i in range(0, x.numel(), step)]
        for w in works: w.wait()
    torch.cuda.synchronize(); print(f"bucket {bucket_mb} MB: all_reduce of 344 MB, 1 rank: {(time.perf_counter()-t0)/10*1e3:.2f} ms")
dist.destroy_process_group()
