"""The clock the row-packed scan runs at and the spread between its workgroups, from the product build (eval_mc(dbg=True):
shader cycles and 100 MHz ticks of every chain wavefront).  usage: clock_rows.py N R"""
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
for rep in range(2):
    env.reset_sampler(list(range(R)), policy=pi)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); o = env.eval_mc(pi, 0.99, dbg=True); t1.record(); torch.cuda.synchronize()
raw = o["dbg"].cpu().numpy()
n_dry, n_req = (raw[:, 0] & 0xffffffff).astype(float), (raw[:, 0] >> 32).astype(float)
n_tie, n_late, n_miss = (raw[:, 1] & 0xffff).astype(float), ((raw[:, 1] >> 16) & 0xffffff).astype(float), (raw[:, 1] >> 40).astype(float)
d = raw[::4].astype(float)  # one row per chain wavefront
it = o["steps"].cpu().numpy().astype(float).max()
cyc, rt = d[:, 2], d[:, 3]
print(f"kernel {t0.elapsed_time(t1):.1f} ms; chain wavefronts: {cyc.mean() / it:.1f} cycles per iteration (mean), clock {(cyc / rt).mean() * 0.1:.3f} GHz, "
      f"wall of the mean / slowest wavefront {rt.mean() * 1e-5:.1f} / {rt.max() * 1e-5:.1f} ms; dry events per row {n_dry.mean():.0f}, ties {n_tie.mean():.1f}")
dry = n_dry.reshape(-1, 16)
print("by workgroup % 8, per row: top-ups", np.round([n_req.reshape(-1, 16)[b::8].mean() for b in range(8)], 0), " not arrived in time", np.round([n_late.reshape(-1, 16)[b::8].mean() for b in range(8)], 0),
      " window end had moved", np.round([n_miss.reshape(-1, 16)[b::8].mean() for b in range(8)], 0))
print("by workgroup % 8: dry events per row", np.round([dry[b::8].mean() for b in range(8)], 0), " cycles per iteration", np.round([cyc.reshape(-1, 4)[b::8].mean() / it for b in range(8)], 1))
print("by workgroup % 8: wall ms", np.round([rt.reshape(-1, 4)[b::8].mean() * 1e-5 for b in range(8)], 1), " clock GHz", np.round([(cyc / rt).reshape(-1, 4)[b::8].mean() * 0.1 for b in range(8)], 3))
