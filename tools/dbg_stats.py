"""Window-kernel counters at bench scale: dry-window events, digest ties, refill phases (tools only)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rl_offline_simulation_amd import synth
from rl_offline_simulation_amd.table import TransitionTable
from rl_offline_simulation_amd.evaluators import BatchedPSRS
N, R, nS, nA = int(float(sys.argv[1])), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
shuffle = sys.argv[5] if len(sys.argv) > 5 else "per_rollout"
e = synth.synth_iid(N, nS, nA)
t0 = e["steps"] == 0
dev = torch.device("cuda", 0)
table = TransitionTable(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], t0, device=dev)
pi = table.policy_slots(synth.dirichlet_policy(nS, nA))
env = BatchedPSRS(table, R)
for it in range(2):
    env.reset_sampler(np.arange(R), shuffle, shuffle_seed=1234)
    torch.cuda.synchronize(); t = time.perf_counter()
    o = env.eval_mc(pi, 0.99, dbg=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
d = o["dbg"].sum(0).tolist(); st = int(o["steps"].sum()); ca = int(o["cand"].sum())
print(f"N={N} R={R} nS={nS} nA={nA} {shuffle}: scan {dt:.3f}s steps={st:.3e} cand={ca:.3e} ns/step/rollout={dt/(st/R)*1e9:.1f} "
      f"dry={d[0]} ({d[0]/st:.4f}/step) tie={d[1]} flush={d[2]} genblocks={d[3]}")
