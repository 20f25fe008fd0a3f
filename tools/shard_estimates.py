"""How far the episode-sharded value estimate (the log split into N episode-disjoint shards, every seed on every shard, per-seed
sum of returns and episode counts added up) sits from the unsharded one, per seed, on the headline table.  One GPU; usage:
shard_estimates.py [transitions] [rollouts]"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl_offline_simulation_amd import synth
from rl_offline_simulation_amd.table import TransitionTable
from rl_offline_simulation_amd.evaluators import evalmc_rollouts
from rl_offline_simulation_amd.distributed import shard_episodes
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
e = synth.synth_iid(N, 162, 2, seed=20221107)
pi = synth.dirichlet_policy(162, 2)
seeds = np.arange(R)


def run(rows):
    ex = {k: (v[rows] if isinstance(v, np.ndarray) and v.shape[:1] == rows.shape else v) for k, v in e.items()}
    t = TransitionTable(ex["z"], ex["actions"], ex["rewards"], ex["z_next"], ex["terminals"], ex["action_distributions"], ex["steps"] == 0)
    o = evalmc_rollouts(t, seeds, pi, 0.99)
    del t
    torch.cuda.empty_cache()
    return o["sum_g"], o["n_ep"].astype(np.float64), o["steps"]


g0, n0, s0 = run(np.ones(N, bool))
v0 = g0 / n0
out = {"transitions": N, "rollouts": R, "unsharded": {"value_mean": float(v0.mean()), "value_std_over_seeds": float(v0.std()), "episodes_per_seed": float(n0.mean()),
                                                         "steps_per_seed": float(s0.mean())}, "sharded": []}
for n_sh in (2, 4, 8):
    g, n, st = np.zeros(R), np.zeros(R), np.zeros(R)
    for k in range(n_sh):
        gk, nk, sk = run(shard_episodes(e["episode_ids"], k, n_sh))
        g, n, st = g + gk, n + nk, st + sk
    v = g / n
    out["sharded"].append({"shards": n_sh, "value_mean": float(v.mean()), "mean_abs_diff_per_seed": float(np.abs(v - v0).mean()), "max_abs_diff_per_seed": float(np.abs(v - v0).max()),
                           "diff_of_means": float(v.mean() - v0.mean()), "episodes_per_seed": float(n.mean()), "steps_per_seed": float(st.mean())})
print(json.dumps(out))
