"""Does a synchronous dist.all_reduce on the RCCL backend block the HOST until the stream reaches it?  One rank.
usage: MASTER_ADDR=127.0.0.1 MASTER_PORT=29531 RANK=0 WORLD_SIZE=1 python profiles/debug/allreduce_host_block.py"""
import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
g = torch.zeros(563_000, device="cuda")
a = torch.randn(8192, 8192, device="cuda")
dist.all_reduce(g); torch.cuda.synchronize()
for mode in ("sync", "async+wait"):
    for busy in (0, 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if busy:
            for _ in range(4):
                b = a @ a                                  # a few ms of queued device work
        t1 = time.perf_counter()
        if mode == "sync":
            dist.all_reduce(g)
        else:
            w = dist.all_reduce(g, async_op=True); w.wait()
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        print(f"{mode:10s} queued work {busy}: enqueue matmuls {1e6*(t1-t0):8.1f} us, all_reduce call {1e6*(t2-t1):8.1f} us, drain {1e6*(t3-t2):8.1f} us")
dist.destroy_process_group()
