"""Timing-only experiment behind DESIGN.md §3.5 "where the backward's time is": the weight-gradient
kernel with whole phases left out (wrong results).  Needs a TEMPORARY patch of
csrc/pcgmix_potes.hip that is not in the tree: an `int skip` kernel argument read from the
environment variable PCGMIX_SKIP by the launcher, and `if (!(skip & 1))` around layer1,
`(skip & 2)` around the dgrad + gw1 block, `(skip & 4)` around the gw2 block.  Result of the run
recorded in profiles/r2_potes_bwd_phases.txt."""
import os, subprocess, sys
for skip in (0, 1, 2, 4, 3, 6, 7):
    env = dict(os.environ, PCGMIX_SKIP=str(skip))
    r = subprocess.run([sys.executable, "profiles/probes/potes_variants_time.py"], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if "bwd masks,     768" in l]
    print("skip", skip, line[0] if line else r.stderr[-300:])
