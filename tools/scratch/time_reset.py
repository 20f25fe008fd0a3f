import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rl_offline_simulation_amd import synth
from rl_offline_simulation_amd.table import TransitionTable
from rl_offline_simulation_amd.evaluators import BatchedPSRS
N, R = int(sys.argv[1]), int(sys.argv[2])
e = synth.synth_iid(N, 162, 2, seed=20221107)
t = TransitionTable(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], e["steps"] == 0)
pi = t.policy_slots(synth.dirichlet_policy(162, 2))
env = BatchedPSRS(t, R)
seeds = np.arange(R, dtype=np.uint64)
for k in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    env.reset_sampler(seeds, policy=pi)
    torch.cuda.synchronize(); dt = time.time() - t0
print(f"reset_sampler {dt:.4f} s", flush=True)
