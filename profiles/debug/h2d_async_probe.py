import torch, time
dev = torch.device("cuda", 0)
src = torch.empty((900_000, 831), dtype=torch.float32).pin_memory()
dst = torch.empty((600_000, 831), dtype=torch.float32, device=dev)
side = torch.cuda.Stream()
x = torch.randn(4096, 4096, device=dev)
for mb in (50, 100, 200, 300, 400, 800):
    n = int(mb * 1e6 / (831 * 4))
    for busy in (False, True):
        torch.cuda.synchronize()
        if busy:
            for _ in range(20):
                y = x @ x
        t0 = time.perf_counter()
        with torch.cuda.stream(side):
            dst[:n].copy_(src[1000:1000 + n], non_blocking=True)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{mb} MB busy={busy}: host {1e3 * (t1 - t0):.2f} ms, total {1e3 * (t2 - t0):.2f} ms  pinned={src[1000:1000+n].is_pinned()}")
