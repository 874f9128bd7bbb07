import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29531")
torch.cuda.set_device(0); dev=torch.device("cuda",0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x=torch.arange(10,dtype=torch.int64,device=dev)
bufs=[torch.empty(10,dtype=torch.int64,device=dev)]
dist.gather(x,bufs,dst=0); print("gather ok", bufs[0][:3].tolist())
t=torch.tensor([1.5],dtype=torch.float64,device=dev); dist.all_reduce(t,op=dist.ReduceOp.MAX); print("allreduce f64 ok", t.item())
dist.barrier(); print("barrier ok")
dist.destroy_process_group(); print("done")
