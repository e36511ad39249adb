"""Does torch.cuda.MemPool / use_mem_pool route a thread's allocations on this build (ROCm)?  Prints segment counts."""
import threading
import time
import torch

dev = torch.device("cuda:0")
print(torch.__version__, hasattr(torch.cuda, "MemPool"), hasattr(torch.cuda, "use_mem_pool"))
pool = torch.cuda.MemPool()
st = torch.cuda.memory_stats()
print("segments before", st["segment.all.current"])
with torch.cuda.use_mem_pool(pool):
    warm = torch.empty(2 << 30, dtype=torch.uint8, device=dev)
    del warm
print("segments after warm", torch.cuda.memory_stats()["segment.all.current"], "pool segments", len(pool.snapshot()))
def work():
    torch.cuda.set_device(0)
    with torch.cuda.use_mem_pool(pool):
        t0 = time.time()
        xs = [torch.empty(100 << 20, dtype=torch.uint8, device=dev) for _ in range(8)]
        print("thread alloc 8 x 100 MB in pool: %.2f ms, pool segments %d" % ((time.time() - t0) * 1e3, len(pool.snapshot())))
th = threading.Thread(target=work); th.start(); th.join()
t0 = time.time()
y = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
print("plain 1 GiB alloc outside the pool: %.2f ms; segments %d" % ((time.time() - t0) * 1e3, torch.cuda.memory_stats()["segment.all.current"]))
for gb in (4, 12, 24):
    t0 = time.time()
    z = torch.empty(gb << 30, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    print("fresh %d GiB hipMalloc through the allocator: %.1f ms" % (gb, (time.time() - t0) * 1e3))
    del z
