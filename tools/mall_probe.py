"""Is a tensor that was just written served from the 256 MB Infinity Cache when the next kernel reads it?
Times a read pass (sum) over y right after a write pass (copy into y) for several sizes."""
import torch
dev = "cuda:0"
for mb in (32, 64, 128, 192, 256, 384, 512, 1024):
    n = mb * (1 << 20) // 4
    x = torch.randn(n, device=dev); y = torch.empty_like(x)
    for _ in range(3):
        y.copy_(x); y.sum()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tw = tr = 0.0
    for _ in range(10):
        e[0].record(); y.copy_(x); e[1].record(); s = y.sum(); e[2].record(); torch.cuda.synchronize()
        tw += e[0].elapsed_time(e[1]); tr += e[1].elapsed_time(e[2])
    print(f"{mb:5d} MB  write pass {2 * mb / 1024 / (tw / 10 / 1e3) / 1e3:6.2f} TB/s (r+w)   read-after-write {mb / 1024 / (tr / 10 / 1e3) / 1e3:6.2f} TB/s", flush=True)
