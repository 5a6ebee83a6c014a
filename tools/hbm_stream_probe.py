import torch, time
n = 276_480_000 // 2
x = torch.empty(n, dtype=torch.float16, device="cuda")
y = torch.empty(n, dtype=torch.float16, device="cuda")
def t(fn, reps=50):
    fn(); torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
ms = t(lambda: x.zero_()); print("fill 276 MB: %.4f ms = %.2f TB/s" % (ms, 276.48e6 / ms / 1e9))
ms = t(lambda: y.copy_(x)); print("copy 276 MB: %.4f ms = %.2f TB/s (read+write)" % (ms, 2 * 276.48e6 / ms / 1e9))
ms = t(lambda: x.sum()); print("read 276 MB: %.4f ms = %.2f TB/s" % (ms, 276.48e6 / ms / 1e9))
