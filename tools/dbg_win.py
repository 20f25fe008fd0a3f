import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from common import load
from rl_offline_simulation_amd.table import TransitionTable
from rl_offline_simulation_amd.evaluators import BatchedPSRS
name = sys.argv[1]
d = load(name)
dev = torch.device("cuda", 0)
table = TransitionTable(d["in_z"], d["in_a"], d["in_r"], d["in_z_next"], d["in_done"], d["in_p_log"], d["in_t0"], device=dev)
seeds = [int(s) for s in d["seeds"]]
env = BatchedPSRS(table, len(seeds))
env.reset_sampler(seeds)
torch.cuda.synchronize(); print("sampler ok", table.n_slots, table.N, table.N0, flush=True)
trace = int(sys.argv[2]) if len(sys.argv) > 2 else 0
o = env.eval_mc(table.policy_slots(d["pi"]), float(d["gamma"]), ep_cap=(table.N0 + 1) if trace else 0, trace_cap=(table.N + 1) if trace else 0, fast=True)
torch.cuda.synchronize()
print("steps", o["steps"].tolist(), "cand", o["cand"].tolist(), "status", o["status"].tolist())
