"""Round 4: the displacement search's launch as a timeline of its blocks.  Builds a PROBE copy of the
library with -DPCGMIX_PHASE_CLOCK (every block of salopt_disp_kernel leaves wall_clock64, 100 MHz, at
entry / lengths known / staged / wave 0 done / all waves done / exit, plus HW_ID and XCC_ID) into
build_probe/ and runs the bs-256 x 5000 bench batch and the single-candidate batch through it.
    [PCGMIX_PROBE_HOSTED=1] python profiles/probes/disp_phase_clock.py     (on the GPU box, from the repo root)
PCGMIX_PROBE_HOSTED=1: the planned launch (pcgmix_salopt_disp_hosted_f32: blocks with candidates only,
longest chain first)."""
import ctypes, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
PKG = glob.glob(os.path.join(ROOT, "pcgmix-*_amd"))[0]
out = os.path.join(ROOT, "build_probe", "libpcgmix_phase_clock.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
srcs = sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip")))
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-shared", "-std=c++17", "--offload-arch=gfx950",
                "-ffp-contract=off", "-DPCGMIX_PHASE_CLOCK", "-I" + os.path.join(ROOT, "include"), "-o", out]
               + srcs, check=True)
import numpy as np
import torch
import pcgmix_amd  # noqa: F401
from pcgmix_amd import _lib
_lib.LIB_PATH = out                      # the probe build instead of the product library
from pcgmix_amd import saliency, synthetic
dev = torch.device("cuda:0")
B, T = 256, 5000
lib = _lib.load()
raw = ctypes.CDLL(out)
sal = torch.rand(B, T, device=dev)
mix_np = np.random.RandomState(0).permutation(B).astype(np.int32)
mix = torch.from_numpy(mix_np).to(dev)
NB = B * 4 * 4
HOSTED = os.environ.get("PCGMIX_PROBE_HOSTED") == "1"


def run(frames, label):
    fr = torch.from_numpy(frames.astype(np.int32)).to(dev)
    L = np.diff(frames, axis=1)
    ml = int(L.max())
    for it in range(4):
        saliency.optimal_displacements(sal, fr.data_ptr(), mix.data_ptr(), 0.37, 0, B, T, max_len=ml,
                                       frames_host=frames if HOSTED else None, mix_host=mix_np if HOSTED else None)
        torch.cuda.synchronize()
    buf = (ctypes.c_longlong * (NB * 8))()
    assert raw.pcgmix_disp_phase_clock(buf, NB) == 0
    if HOSTED:
        return run_hosted(np.frombuffer(buf, dtype=np.int64).reshape(NB, 8), label)
    t = np.frombuffer(buf, dtype=np.int64).reshape(4, 4, B, 8).transpose(1, 0, 2, 3)   # launch order [state slot y][slice z][x] -> [z][y][x][stamp]
    t0 = t[..., 0].min()
    us = (t[..., :6] - t0) / 100.0
    end = us[..., 5]
    worked = t[..., 1] >= t[..., 0]
    worked &= (t[..., 2] >= t[..., 1]) & (t[..., 2] - t[..., 0] < 10**7)
    print(f"## {label}: launch span {end.max():.1f} us; {int(worked.sum())} of {NB} blocks had candidates")
    print("   entry times of all blocks: min %.1f  median %.1f  p90 %.1f  max %.1f us" %
          tuple(np.percentile(us[..., 0], [0, 50, 90, 100])))
    w = us[worked]
    for name, a, b_ in (("lengths known (3 dependent loads)", 0, 1), ("staging", 1, 2), ("wave 0 scan", 2, 3),
                        ("slowest wave after wave 0", 3, 4), ("arg-max + store", 4, 5), ("block total", 0, 5)):
        dt = w[:, b_] - w[:, a]
        print("   %-36s median %6.2f  p90 %6.2f  max %6.2f us" % (name, np.median(dt), np.percentile(dt, 90), dt.max()))
    empty = us[~worked]
    if len(empty):
        dt = empty[:, 5] - empty[:, 0]
        print("   %-36s median %6.2f  p90 %6.2f  max %6.2f us" % ("empty block total", np.median(dt), np.percentile(dt, 90), dt.max()))
    # the ten blocks that end last
    flat = np.argsort(end.ravel())[::-1][:10]
    print("   last blocks: (z, state-slot y, sample x) entry -> exit, own len, partner len")
    order = [3, 1, 0, 2]
    for f in flat:
        z, y, x = np.unravel_index(f, end.shape)
        b = (x + z * (B // 4 + 3)) % B
        k = order[y]
        print("     z=%d y=%d b=%3d  %.1f -> %.1f us   n1=%d n2=%d  hw=%x xcc=%d" %
              (z, y, b, us[z, y, x, 0], us[z, y, x, 5], L[b, k], L[mix_np[b], k], t[z, y, x, 6] & 0xffffffff, t[z, y, x, 7] & 0xf))
    # blocks per CU (xcc, se, cu from HW_ID) among those with candidates
    hw = t[..., 6][worked] & 0xffffffff
    xcc = t[..., 7][worked] & 0xf
    cu = (xcc << 12) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf)
    cnt = np.bincount(np.unique(cu, return_inverse=True)[1])
    print("   working blocks per CU: CUs used %d, min %d, median %d, max %d" % (len(cnt), cnt.min(), np.median(cnt), cnt.max()))
    busy = np.zeros(len(cnt))
    inv = np.unique(cu, return_inverse=True)[1]
    np.add.at(busy, inv, w[:, 5] - w[:, 0])
    print("   sum of block times per CU: min %.1f median %.1f max %.1f us" % (busy.min(), np.median(busy), busy.max()))


def run_hosted(t, label):
    """Planned launch: block i of the grid is entry i of the plan; stamps of the LAST launch (earlier
    launches of this process had the same plan)."""
    t = t[t[:, 0] > 0]
    t = t[t[:, 0] >= t[:, 0].max() - 10**6]          # this launch only (within 10 ms)
    t0 = t[:, 0].min()
    us = (t[:, :6] - t0) / 100.0
    worked = (t[:, 1] >= t[:, 0]) & (t[:, 2] >= t[:, 1]) & (t[:, 2] - t[:, 0] < 10**7)
    print(f"## {label} [planned]: launch span {us[:, 5].max():.1f} us; {len(t)} blocks, {int(worked.sum())} with candidates")
    print("   entry times of all blocks: min %.1f  median %.1f  p90 %.1f  max %.1f us" %
          tuple(np.percentile(us[:, 0], [0, 50, 90, 100])))
    w = us[worked]
    sh = (t[:, 7][worked] >> 8).astype(np.float64)
    scan = w[:, 3] - w[:, 2]
    ok = scan > 1.0
    print("   shader clocks per us of wall clock over the scan (s_memtime / wall_clock64): median %.0f MHz" %
          np.median(sh[ok] / scan[ok]))
    for name, a, b_ in (("lengths known (3 dependent loads)", 0, 1), ("staging", 1, 2), ("wave 0 scan", 2, 3),
                        ("slowest wave after wave 0", 3, 4), ("arg-max + store", 4, 5), ("block total", 0, 5)):
        dt = w[:, b_] - w[:, a]
        print("   %-36s median %6.2f  p90 %6.2f  max %6.2f us" % (name, np.median(dt), np.percentile(dt, 90), dt.max()))
    order = np.argsort(us[:, 5])[::-1][:10]
    print("   last blocks: plan position, entry -> exit")
    for i in order:
        print("     #%4d  %.1f -> %.1f us" % (i, us[i, 0], us[i, 5]))
    hw = t[:, 6][worked] & 0xffffffff
    xcc = t[:, 7][worked] & 0xf
    cu = (xcc << 12) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf)
    inv = np.unique(cu, return_inverse=True)[1]
    cnt = np.bincount(inv)
    busy = np.zeros(len(cnt))
    np.add.at(busy, inv, w[:, 5] - w[:, 0])
    print("   working blocks per CU: CUs used %d, min %d, median %d, max %d" % (len(cnt), cnt.min(), np.median(cnt), cnt.max()))
    print("   sum of block times per CU: min %.1f median %.1f max %.1f us" % (busy.min(), np.median(busy), busy.max()))


same = np.tile(np.array([0, 300, 1100, 1400, 2800]), (B, 1))
one = same.copy(); one[::2, 4] += 1
run(one, "diastole gap 1 (two candidates, mid 1400), other states equal")
frames, labels, wav = synthetic.make_index_data(B, T, sample_rate=2000, seed=0)
run(frames, "the bench batch")
