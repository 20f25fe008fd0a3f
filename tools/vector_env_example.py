"""A PPO-shaped driver on the batched evaluator: a small policy network reveals an action distribution per environment and
per step, VectorPSRS serves all environments with one launch per step (the single-environment loop of the reference's
examples/cartpole/psrs_from_expert_heuristic.py:59-80, vectorised).  Prints simulated steps/s; run on the GPU box.

usage: python tools/vector_env_example.py [n_envs] [log_transitions]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl_offline_simulation_amd import OfflineDataset, ProbDistribution, spaces, synth
from rl_offline_simulation_amd.encoders import CartpoleBoxEncoder
from rl_offline_simulation_amd.evaluators import VectorPSRS

R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
e = synth.cartpole_log(N, seed=0)
ds = OfflineDataset(spaces.Box(-np.inf, np.inf, (4,), np.float32), spaces.Discrete(2), ProbDistribution.Discrete,
                    **{k: e[k] for k in ("observations", "actions", "action_distributions", "rewards", "next_observations", "terminals", "steps", "episode_ids")})
env = VectorPSRS(ds, num_envs=R, num_states=162, encoder=CartpoleBoxEncoder())
dev = env.table.device
torch.manual_seed(0)
policy = torch.nn.Sequential(torch.nn.Linear(4, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64), torch.nn.Tanh(), torch.nn.Linear(64, 2)).to(dev)
env.reset_sampler(np.arange(R))
obs, alive = env.reset()
use_graph = os.environ.get("OFFSIM_EXAMPLE_GRAPH", "1") != "0"
dist = lambda o: torch.softmax(policy(o), dim=1).to(torch.float64)
if os.environ.get("OFFSIM_EXAMPLE_POLICY") == "fixed":  # the environments' share of an iteration: no network, one fixed distribution per environment
    fixed_probs = torch.full((R, 2), 0.5, dtype=torch.float64, device=dev)
    dist = lambda o: fixed_probs
n_calls = 3000
torch.cuda.synchronize()
if use_graph:  # one driver iteration captured in a HIP graph: the eager loop is bound by its ~15 launches per iteration
    g, (a, r, done) = env.graph_iteration(dist)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_calls):
        g.replay()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    calls = n_calls
else:
    t0 = time.perf_counter()
    calls = 0
    with torch.no_grad():
        while calls < n_calls:
            a, obs, r, done, alive = env.step_dist_batch(dist(env.obs))
            calls += 1
            if calls % 50 == 0 and not bool(alive.any()):  # (a host sync every 50 calls only)
                break
            env.reset(mask=done)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
live_frac = float(env.alive.float().mean())
import json
print(json.dumps({"tool": "vector_env_example", "mode": "hip graph replay" if use_graph else "eager", "policy": os.environ.get("OFFSIM_EXAMPLE_POLICY", "4-64-64-2 MLP + softmax (8 torch kernels)"), "environments": R, "log_transitions": N,
                  "calls": calls, "us_per_call": el / calls * 1e6, "simulated_steps_per_s_lower_bound": R * calls * live_frac / el,
                  "alive_fraction_at_end": live_frac,
                  "reference": "single-environment Python loop: ~1e5 steps/s at N = 5e4, ~2e4 at N = 1e6 (BASELINE.md)"}))
