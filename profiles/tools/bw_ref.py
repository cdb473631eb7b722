import torch, time
n = 8 * 1024**3
a = torch.empty(n, dtype=torch.uint8, device="cuda"); b = torch.empty(n, dtype=torch.uint8, device="cuda")
def t(f, reps=5):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    ms = []
    for _ in range(reps):
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ms.append(e0.elapsed_time(e1))
    return min(ms)
m = t(lambda: a.fill_(7)); print("fill 8 GiB: %.3f ms -> %.2f TB/s" % (m, n / m / 1e9))
m = t(lambda: a.zero_()); print("zero 8 GiB: %.3f ms -> %.2f TB/s" % (m, n / m / 1e9))
m = t(lambda: b.copy_(a)); print("copy 8 GiB: %.3f ms -> %.2f TB/s (read+write)" % (m, 2 * n / m / 1e9))
a32 = a.view(torch.int32)
m = t(lambda: a32.sum()); print("read 8 GiB (sum): %.3f ms -> %.2f TB/s" % (m, n / m / 1e9))
