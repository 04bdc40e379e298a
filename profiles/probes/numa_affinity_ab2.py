import glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
mode = sys.argv[1]
def cpulist(s):
    out = []
    for part in s.strip().split(","):
        if "-" in part:
            a, b = part.split("-"); out += list(range(int(a), int(b) + 1))
        elif part: out.append(int(part))
    return out
import torch
pr = torch.cuda.get_device_properties(0)
bus = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
node = None
try: node = int(open("/sys/bus/pci/devices/%s/numa_node" % bus).read())
except Exception as e: node = "err %s" % e
nodes = {int(p.split("node")[-1].split("/")[0]): cpulist(open(p).read()) for p in glob.glob("/sys/devices/system/node/node*/cpulist")}
info = "bus %s numa %s" % (bus, node)
if mode.startswith("node"):
    os.sched_setaffinity(0, nodes[int(mode[4:])])
elif mode == "auto" and isinstance(node, int) and node in nodes:
    os.sched_setaffinity(0, nodes[node])
elif mode == "core" and isinstance(node, int) and node in nodes:
    os.sched_setaffinity(0, nodes[node][8:12])
import bench
dev = torch.device("cuda:0")
_, data, tgt, frames, labels, wav = bench.make_device_batch(256, 4, 5000, 2000, 0, dev)
bench.settle_clocks(dev)
dt, _ = bench.run_augment_steps("durratiomixup", data, tgt, frames, wav, dev, 3000, 50, lambda: None)
print("%-6s strict %.2f us  [%s; cpus now %d]" % (mode, dt / 3000 * 1e6, info, len(os.sched_getaffinity(0))), flush=True)
