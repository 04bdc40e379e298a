import sys, subprocess
CODE = r'''
import sys, torch, torch.nn.functional as F
sys.path.insert(0, ".")
import pcgmix_amd
from pcgmix_amd import models
variant = sys.argv[1]
device = torch.device("cuda", 0)
torch.manual_seed(8)
B, K, C = (100, 19968, 2) if "b100" in variant else (32, 9968, 2)
w1 = (torch.randn(20, K, device=device) * 0.01).requires_grad_(True)
b1 = torch.randn(20, device=device).requires_grad_(True)
w2 = torch.randn(C, 20, device=device).requires_grad_(True)
b2 = torch.randn(C, device=device).requires_grad_(True)
feat = torch.randn(B, K, device=device).requires_grad_(True)
t = F.one_hot(torch.randint(0, C, (B,), device=device), C).float()
gs = torch.tensor(0.5, device=device)
params = (feat, w1, b1, w2, b2)
def run():
    loss, logits = models.PotesHeadLossFunction.apply(feat, w1, b1, w2, b2, t, 0.0, 0.0, True, None)
    if "grad" in variant:
        grads = torch.autograd.grad(loss, params, gs)
    else:
        for p in params: p.grad = None
        loss.backward(gs)
        grads = [p.grad for p in params]
    return loss, logits, grads
run()
side = torch.cuda.Stream(device)
side.wait_stream(torch.cuda.current_stream(device))
with torch.cuda.stream(side):
    run()
torch.cuda.current_stream(device).wait_stream(side)
torch.cuda.synchronize()
print("warm ok", flush=True)
graph = torch.cuda.CUDAGraph()
mode = "thread_local" if "tl" in variant else "global"
with torch.cuda.graph(graph, capture_error_mode=mode):
    out = run()
print("captured", flush=True)
graph.replay(); torch.cuda.synchronize()
print("replayed", float(out[0]), flush=True)
'''
for v in ("backward_tl", "grad_tl", "backward_tl_b100", "grad_global"):
    r = subprocess.run([sys.executable, "-c", CODE, v], capture_output=True, text=True, timeout=300)
    print(v, "rc", r.returncode, r.stdout.strip().replace("\n", " | "), r.stderr.strip()[-300:].replace("\n", " | "))
