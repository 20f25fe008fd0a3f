import os, sys
import numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
from rl_offline_simulation_amd import synth
from rl_offline_simulation_amd.table import TransitionTable
from rl_offline_simulation_amd.evaluators import BatchedPSRS
class A: pass
a = A(); a.workload = "grid"; a.transitions = 10_000_000; a.n_states = 162; a.n_actions = 2
e, _ = bench.make_log(a, 20221107, torch.device("cuda", 0))
table = TransitionTable(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], e["steps"] == 0)
pi = table.policy_slots(synth.dirichlet_policy(a.n_states, a.n_actions))
R = 1024
env = BatchedPSRS(table, R)
env.reset_sampler(list(range(R)), policy=pi)
o = env.eval_mc(pi, 0.99, dbg=True)
torch.cuda.synchronize()
raw = o["dbg"].cpu().numpy()
n_dry = (raw[:, 0] & 0xffffffff).astype(float); n_tie = (raw[:, 1] & 0xffff).astype(float)
it = o["steps"].cpu().numpy().astype(float)
cyc = raw[::4, 2].astype(float)
print("steps per rollout %.0f, candidates per step %.2f, dry events per row %.0f (%.2f %% of its steps), ties %.0f, cycles per iteration %.0f" % (
    it.mean(), o["cand"].cpu().numpy().astype(float).sum() / it.sum(), n_dry.mean(), 100 * n_dry.mean() / it.mean(), n_tie.mean(), cyc.mean() / it.max()))
