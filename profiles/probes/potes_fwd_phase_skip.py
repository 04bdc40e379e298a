"""Timing-only experiment (wrong results): the conv stack forward with whole phases left out.
Needs a TEMPORARY patch of csrc/pcgmix_potes.hip that is not in the tree (an `int skip` kernel
argument read from PCGMIX_SKIP by the launchers; bit 1 around the layer1_t calls, bit 2 around
the second layer's channel loop).  Result: profiles/r2_potes_fwd_phases.txt."""
import os, subprocess, sys
for skip in (0, 1, 2, 3):
    env = dict(os.environ, PCGMIX_SKIP=str(skip))
    r = subprocess.run([sys.executable, "profiles/probes/potes_variants_time.py"], env=env, capture_output=True, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith("fwd")]
    print("skip", skip, " | ".join(lines) if lines else r.stderr[-300:])
