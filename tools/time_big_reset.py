"""Sampler reset (keyed) of tables whose states do not fit LDS: seconds per pass and ns per queue position, chunked kernel against the
in-place variant.  usage: time_big_reset.py N nS R [chunked=1]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl_offline_simulation_amd import synth
from rl_offline_simulation_amd.table import TransitionTable
from rl_offline_simulation_amd.evaluators import BatchedPSRS
N, nS, R = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
e = synth.synth_iid(N, nS, 2, seed=7)
table = TransitionTable(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], e["steps"] == 0)
pi = table.policy_slots(synth.dirichlet_policy(nS, 2))
env = BatchedPSRS(table, R)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    env.reset_sampler(list(range(R)), policy=pi)
    torch.cuda.synchronize(); dt = time.time() - t0
print(f"N {N} states {nS} (max {table.max_seg} rows) x {R} rollouts, chunk {os.environ.get('OFFSIM_SHUFFLE_CHUNK', 'default')} chunked {os.environ.get('OFFSIM_SHUFFLE_CHUNKED', '1')}: "
      f"{dt:.3f} s per pass, {dt / (N * R) * 1e12:.1f} ps per position")
