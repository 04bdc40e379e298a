"""Timing-only experiment (wrong results; needs a temporary `getenv_skip` patch of
potes_head_bwd_kernel: bit 1 = plain stores instead of the two-addend atomicAdd on dW1, bit 2 = no
dW1 reduction/store at all): cost of the feature pass's epilogue inside the captured train step.
Result (MI355X): 153.5 us per step as is, 153.0 with plain stores, 152.4 without the epilogue —
the two-addend atomics are not worth replacing."""
import os, subprocess, sys
for skip in (0, 1, 2):
    env = dict(os.environ, PCGMIX_SKIP=str(skip))
    r = subprocess.run([sys.executable, "profiles/probes/train_step_host.py"], env=env, capture_output=True, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith("rep 2")]
    print("skip", skip, lines[0] if lines else r.stderr[-300:])
