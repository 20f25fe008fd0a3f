"""Debug aid: first step at which the row-packed scan and the window kernel disagree (run on the GPU box)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl_offline_simulation_amd import synth
from rl_offline_simulation_amd.table import TransitionTable
from rl_offline_simulation_amd.evaluators import BatchedPSRS

N, nS, nA, R = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000, 162, 2, int(sys.argv[2]) if len(sys.argv) > 2 else 64
e = synth.synth_iid(N, nS, nA, seed=11)
t0 = e["steps"] == 0
table = TransitionTable(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], t0)
pi = table.policy_slots(synth.dirichlet_policy(nS, nA))
seeds = list(range(100, 100 + R))
a = BatchedPSRS(table, R)
a.reset_sampler(seeds, policy=pi)
oa = a.eval_mc(pi, 0.99, trace_cap=N, dbg=True)
b = BatchedPSRS(table, R)
b.reset_sampler(seeds)
os.environ["OFFSIM_SCAN_ROWS"] = "0"
ob = b.eval_mc(pi, 0.99, trace_cap=N)
torch.cuda.synchronize()
za, zn = e["z"], e["z_next"]
for i in range(R):
    ra, rb = oa["trace_row"][i].cpu().numpy(), ob["trace_row"][i].cpu().numpy()
    pa, pb = oa["trace_pop"][i].cpu().numpy(), ob["trace_pop"][i].cpu().numpy()
    na, nb = int(oa["steps"][i]), int(ob["steps"][i])
    d = np.nonzero((ra[:min(na, nb)] != rb[:min(na, nb)]) | (pa[:min(na, nb)] != pb[:min(na, nb)]))[0]
    same_g = float(oa["sum_g"][i]) == float(ob["sum_g"][i])
    if len(d) == 0 and na == nb and same_g:
        continue
    print(f"rollout {i}: steps {na} vs {nb}, sum_g equal {same_g}, n_ep {int(oa['n_ep'][i])} vs {int(ob['n_ep'][i])}, dbg {oa['dbg'][i].cpu().tolist()}")
    if len(d):
        k = int(d[0])
        lo = max(0, k - 3)
        print("  first diff at step", k, "tick pos", k % 16)
        print("  rows kernel rows", ra[lo:k + 3], "pop", pa[lo:k + 3], "z of rows", za[ra[lo:k + 3]], "done", e["terminals"][ra[lo:k + 3]])
        print("  win  kernel rows", rb[lo:k + 3], "pop", pb[lo:k + 3], "z of rows", za[rb[lo:k + 3]])
print("done")
