"""Does the strict-signature step depend on which NUMA node the calling thread runs on?
    python profiles/probes/numa_affinity_ab.py <local|remote|none>"""
import glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
mode = sys.argv[1] if len(sys.argv) > 1 else "none"


def cpulist(s):
    out = []
    for part in s.strip().split(","):
        if "-" in part:
            a, b = part.split("-"); out += list(range(int(a), int(b) + 1))
        elif part:
            out.append(int(part))
    return out


nodes = {}
for p in glob.glob("/sys/devices/system/node/node*/cpulist"):
    nodes[int(p.split("node")[-1].split("/")[0])] = cpulist(open(p).read())
gpu_node = None
for p in sorted(glob.glob("/sys/class/drm/card*/device/numa_node")):
    try:
        v = int(open(p).read())
        if v >= 0:
            gpu_node = v; break
    except Exception:
        pass
allowed = sorted(os.sched_getaffinity(0))
info = "nodes %s, gpu numa node %s, allowed cpus %d (%d..%d)" % (sorted(nodes), gpu_node, len(allowed), allowed[0], allowed[-1])
if mode != "none" and gpu_node is not None and len(nodes) > 1:
    want = nodes[gpu_node] if mode == "local" else [c for n, cs in nodes.items() if n != gpu_node for c in cs]
    want = sorted(set(want) & set(allowed))
    if want:
        os.sched_setaffinity(0, want)
        info += "; pinned to %d cpus of the %s node(s)" % (len(want), mode)
import torch
import bench
dev = torch.device("cuda:0")
_, data, tgt, frames, labels, wav = bench.make_device_batch(256, 4, 5000, 2000, 0, dev)
bench.settle_clocks(dev)
dt, _ = bench.run_augment_steps("durratiomixup", data, tgt, frames, wav, dev, 3000, 50, lambda: None)
print("%-6s strict augment() step %.2f us   [%s]" % (mode, dt / 3000 * 1e6, info), flush=True)
