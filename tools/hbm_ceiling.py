# Measures what this box's HBM delivers to plain streaming kernels (torch ops), as a yardstick for the stash traffic.
import torch, time
dev = "cuda:0"
n = 1 << 29            # 2 GiB of fp32
x = torch.empty(n, dtype=torch.float32, device=dev).normal_()
y = torch.empty_like(x)
def t(f, reps=10):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
b = n * 4
print("copy  (r+w) TB/s", 2 * b / t(lambda: y.copy_(x)) / 1e12)
print("fill  (w)   TB/s", b / t(lambda: y.fill_(1.0)) / 1e12)
print("sum   (r)   TB/s", b / t(lambda: x.sum()) / 1e12)
print("add   (2r+w)TB/s", 3 * b / t(lambda: torch.add(x, y, out=y)) / 1e12)
