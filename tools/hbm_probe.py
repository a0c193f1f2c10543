"""Achievable HBM bandwidth of the box (SURVEY 8d asks for a measured figure next to the 8 TB/s of the data sheet):
device-to-device copy and a read-only reduction over buffers far larger than the caches."""
import torch
assert torch.cuda.is_available()
dev = torch.device("cuda", 0)
n = 1 << 30                                          # 4 GiB of fp32 per buffer
a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
b = torch.empty_like(a)

def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3

t = timed(lambda: b.copy_(a))
print("copy  : %.2f TB/s (read + write, %d GiB each way)" % (2 * a.numel() * 4 / t / 1e12, a.numel() * 4 >> 30))
t = timed(lambda: a.sum())
print("reduce: %.2f TB/s (read only)" % (a.numel() * 4 / t / 1e12))
t = timed(lambda: b.fill_(1.0))
print("fill  : %.2f TB/s (write only)" % (a.numel() * 4 / t / 1e12))
