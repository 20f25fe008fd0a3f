import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl_offline_simulation_amd import synth, _lib as L
from rl_offline_simulation_amd.table import TransitionTable
from rl_offline_simulation_amd.evaluators import BatchedPSRS
n, nS, R = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
allinit = len(sys.argv) > 4
e = synth.synth_iid(n, nS, 2, seed=n)
t0 = np.ones(n, bool) if allinit else e["steps"] == 0
table = TransitionTable(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], t0)
pi = table.policy_slots(synth.dirichlet_policy(nS, 2))
seeds = [int(x) for x in np.random.default_rng(n).integers(0, 1 << 62, R)]
plain = BatchedPSRS(table, R); plain.reset_sampler(seeds)
envc = BatchedPSRS(table, R); envc.reset_sampler(seeds, policy=pi)
torch.cuda.synchronize()
print("faults", L.load().offsim_async_faults(), "format", envc._streams["format"], "N0", table.N0, "max_seg", table.max_seg)
a = (envc.perm.to(torch.int64) & 0xFFFFFFFF).cpu().numpy(); b = (plain.state.perm.to(torch.int64) & 0xFFFFFFFF).cpu().numpy()
bad = np.argwhere(a != b)
print("perm mismatches", len(bad), bad[:10].tolist(), [(int(a[i, j]), int(b[i, j])) for i, j in bad[:10]])
ia = envc.state.init_perm.cpu().numpy(); ib = plain.state.init_perm.cpu().numpy()
bad = np.argwhere(ia != ib)
print("init mismatches", len(bad), bad[:10].tolist())
