"""Plain fill / copy / reduce of a 411 MB bf16 tensor (the 64-channel activation at 224 x 224, batch 64) with torch's
own kernels: what the card gives a streaming kernel, for judging the HBM-bound kernels (not a pytest file)."""
import torch, time
n = 64*224*224*64
a = torch.empty(n, dtype=torch.bfloat16, device="cuda"); b = torch.empty_like(a)
def t(fn, it=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
print("fill  %.1f us  %.2f TB/s" % ((u := t(lambda: a.fill_(1.0))), n*2/u/1e6))
print("copy  %.1f us  %.2f TB/s (r+w)" % ((u := t(lambda: b.copy_(a))), n*4/u/1e6))
print("sum   %.1f us  %.2f TB/s" % ((u := t(lambda: a.view(torch.int16).max())), n*2/u/1e6))
