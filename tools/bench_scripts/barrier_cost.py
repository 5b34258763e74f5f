import os, sys, time, torch, torch.distributed as dist
torch.cuda.set_device(0); dev=torch.device("cuda",0)
os.environ.setdefault("MASTER_ADDR","127.0.0.1")
dist.init_process_group("nccl", device_id=dev)
dist.barrier(); torch.cuda.synchronize()
def bar():
    dist.barrier(); torch.cuda.synchronize()
for _ in range(3): bar()
t0=time.perf_counter()
for _ in range(20): bar()
print("barrier+sync: %.3f ms"%((time.perf_counter()-t0)/20*1e3), file=sys.stderr)
t=torch.zeros(1,device=dev)
def bar2():
    dist.all_reduce(t); torch.cuda.synchronize()
for _ in range(3): bar2()
t0=time.perf_counter()
for _ in range(20): bar2()
print("all_reduce(1 elem)+sync: %.3f ms"%((time.perf_counter()-t0)/20*1e3), file=sys.stderr)
dist.destroy_process_group()
