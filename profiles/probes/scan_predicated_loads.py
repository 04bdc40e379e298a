#!/usr/bin/env python3
"""Round 4: scan the gfx950 ISA of every kernel in csrc/ for loads that sit inside an exec-masked
branch with an `s_waitcnt` right behind them — what `cond ? p[i] : 0` compiles to: the request is
not issued with the lane's other loads and the wave waits for it on the spot.  Found the partner
load of the splice kernels, 31 loads in saliency_post's staging loop and the filter weights of the
log-mel setup this way (profiles/r4_mix_unpredicated_loads.txt).  Runs on the build host (no GPU):
    python profiles/probes/scan_predicated_loads.py"""
import glob, os, re, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = glob.glob(os.path.join(ROOT, "pcgmix-*_amd"))[0]
tmp = tempfile.mkdtemp()
for src in sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip"))):
    out = os.path.join(tmp, os.path.basename(src)[:-4] + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off",
                    "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", out, src],
                   check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    cur, hits = None, {}
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur = m.group(1)
        if cur and "s_cbranch_execz" in l:
            for j in range(i + 1, min(i + 9, len(lines))):
                kind = "global" if ("global_load" in lines[j] or "buffer_load" in lines[j]) else \
                       ("lds" if "ds_read" in lines[j] else None)
                if kind:
                    cnt = "vmcnt(0)" if kind == "global" else "lgkmcnt(0)"
                    if any("s_waitcnt" in lines[k] and cnt in lines[k] for k in range(j + 1, min(j + 5, len(lines)))):
                        hits.setdefault((cur, kind), 0)
                        hits[(cur, kind)] += 1
                    break
    for (k, kind), v in sorted(hits.items(), key=lambda kv: -kv[1]):
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0]
        print(f"{os.path.basename(src):22s} {kind:6s} {v:3d}  {name}")
