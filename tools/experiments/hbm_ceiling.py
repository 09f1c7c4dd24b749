"""What this box's HBM delivers to plain streaming kernels (the yardstick for the HBM-bound kernels' TB/s): a read-only reduction, a copy,
a fill over 4 GiB, best of 5, HIP events."""
import torch

dev = torch.device("cuda:0")
n = 1 << 30
a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
b = torch.empty_like(a)


def best(f, reps=5):
    t = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        t.append(e0.elapsed_time(e1))
    return min(t)


for name, f, nbytes in (("read  (a.sum())", lambda: a.sum(), 4 * n), ("read  (a.max())", lambda: a.max(), 4 * n), ("copy  (b.copy_(a))", lambda: b.copy_(a), 8 * n),
                        ("write (b.fill_(1))", lambda: b.fill_(1.0), 4 * n), ("a.mul_(2) (read + write)", lambda: a.mul_(1.0001), 8 * n)):
    f()
    ms = best(f)
    print("%-28s %.3f ms  %.2f TB/s" % (name, ms, nbytes / ms / 1e9))
