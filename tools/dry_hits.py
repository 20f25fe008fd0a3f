"""Debug: share of the dry rows that the loop served from the request areas (library built with -DROWS_DRY_COUNT_HITS)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl_offline_simulation_amd import synth
from rl_offline_simulation_amd.table import TransitionTable
from rl_offline_simulation_amd.evaluators import BatchedPSRS
N, R = int(sys.argv[1]), int(sys.argv[2])
e = synth.synth_iid(N, 162, 2, seed=20221107)
table = TransitionTable(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], e["steps"] == 0)
pi = table.policy_slots(synth.dirichlet_policy(162, 2))
env = BatchedPSRS(table, R)
env.reset_sampler(list(range(R)), policy=pi)
o = env.eval_mc(pi, 0.99, dbg=True)
torch.cuda.synchronize()
raw = o["dbg"].cpu().numpy()
v = raw[:, 0] & 0xffffffff
print("dry rows (asm + C++):", (v & 0xffff).mean(), "(mod 65536)  served from the request areas:", (v >> 16).mean())
