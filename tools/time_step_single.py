"""Latency of the drop-in single-environment step (per_state_rejection.py:85-95: one Python call per simulated step): the resident step
server against one launch + stream synchronise per call (OFFSIM_STEP_SERVER=0).  usage: time_step_single.py [N] [calls]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rl_offline_simulation_amd import synth
from rl_offline_simulation_amd.evaluators import PSRS
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000
e = synth.synth_iid(N, 162, 2, seed=3)
res = {}
for mode in ("1", "0"):
    os.environ["OFFSIM_STEP_SERVER"] = mode
    env = PSRS.from_arrays(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], e["steps"] == 0)
    env.reset_sampler(7)
    env.reset()
    p = np.array([0.5, 0.5])
    rows = []
    for _ in range(200):
        env.step(p)
    t0 = time.perf_counter()
    n = 0
    for _ in range(calls):
        s, r, d, info = env.step(p)
        if s is None:
            break
        n += 1
        if d and env.reset() is None:
            break
    dt = time.perf_counter() - t0
    res[mode] = dt / max(n, 1) * 1e6
    print(f"OFFSIM_STEP_SERVER={mode}: {n} steps, {res[mode]:.2f} us per PSRS.step (incl. env.reset() at episode ends)")
    del env
