"""Probe: cost of drawing the Potes dropout masks with torch's generator, in different forms."""
import torch, time
dev = torch.device("cuda:0")
B, K = 256, 19968
x = torch.randn(B, K, device=dev)
def t(name, fn, n=200):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:44s} {e0.elapsed_time(e1) / n * 1e3:6.1f} us")
mb = torch.empty(B, K, dtype=torch.bool, device=dev)
mu = torch.empty(B, K, dtype=torch.uint8, device=dev)
mall = torch.empty(B * (K + 20), dtype=torch.bool, device=dev)
ms = torch.empty(B, 20, dtype=torch.bool, device=dev)
t("native_dropout(x, .25) -> (out, mask)", lambda: torch.ops.aten.native_dropout(x, 0.25, True))
t("bool (B,K).bernoulli_(.75)", lambda: mb.bernoulli_(0.75))
t("uint8 (B,K).bernoulli_(.75)", lambda: mu.bernoulli_(0.75))
t("bool (B,K+20) flat .bernoulli_", lambda: mall.bernoulli_(0.75))
t("bool (B,20).bernoulli_(.5)", lambda: ms.bernoulli_(0.5))
r = torch.empty(B * K // 4, dtype=torch.int32, device=dev)
t("int32 (B*K/4).random_()", lambda: r.random_())
