import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29511")
dev=torch.device("cuda",0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x=torch.ones(4096,device=dev,dtype=torch.bfloat16)
dist.all_reduce(x); torch.cuda.synchronize()
g=torch.cuda.CUDAGraph()
s=torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    dist.all_reduce(x)
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
with torch.cuda.graph(g):
    y = x*2
    dist.all_reduce(y)
g.replay(); torch.cuda.synchronize(); print("nccl in graph ok", y[:2])
dist.destroy_process_group()
