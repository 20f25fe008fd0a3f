"""Time reset_sampler alone on an iid synthetic log: usage  time_reset.py N nS R [keyed]  (permutation form unless `keyed`).  Few states -> segments
above 65536 rows -> the global-memory variant of the shuffle."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl_offline_simulation_amd import synth
from rl_offline_simulation_amd.table import TransitionTable
from rl_offline_simulation_amd.evaluators import BatchedPSRS
N, nS, R = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
keyed = len(sys.argv) > 4 and sys.argv[4] == "keyed"
e = synth.synth_iid(N, nS, 2, seed=1)
t = TransitionTable(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], e["steps"] == 0)
env = BatchedPSRS(t, R)
pi = t.policy_slots(synth.dirichlet_policy(nS, 2)) if keyed else None
seeds = np.arange(R, dtype=np.uint64)
for k in range(4):
    torch.cuda.synchronize(); t0 = time.time()
    env.reset_sampler(seeds, policy=pi)
    torch.cuda.synchronize(); dt = time.time() - t0
print(f"N={N} nS={nS} R={R}{' keyed' if keyed else ''}: reset_sampler {dt:.4f} s  ({N*R/dt:.3e} swaps/s)", flush=True)
