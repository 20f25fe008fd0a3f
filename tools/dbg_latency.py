"""Single-environment drop-in latency: PerStateRejectionSampling.step_dist calls per second (H8 of the survey)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rl_offline_simulation_amd import OfflineDataset, ProbDistribution, spaces, synth
from rl_offline_simulation_amd.evaluators import PerStateRejectionSampling
from rl_offline_simulation_amd.encoders import CartpoleBoxEncoder
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50000
e = synth.cartpole_log(N, seed=1)
ds = OfflineDataset(spaces.Box(-5, 5, (4,)), spaces.Discrete(2), ProbDistribution.Discrete,
                    **{k: e[k] for k in ("observations", "actions", "rewards", "next_observations", "terminals", "steps", "episode_ids", "action_distributions")})
t = time.perf_counter()
env = PerStateRejectionSampling(ds, num_states=162, encoder=CartpoleBoxEncoder(), new_step_api=True)
print(f"construction {time.perf_counter()-t:.3f}s")
t = time.perf_counter(); env.reset_sampler(0); torch.cuda.synchronize(); print(f"reset_sampler {time.perf_counter()-t:.4f}s")
p = np.array([0.5, 0.5])
obs = env.reset(); n = 0
t = time.perf_counter()
while n < 3000:
    try:
        a, obs, r, term, trunc, info = env.step_dist(p)
    except KeyError:
        obs = env.reset(); continue
    if a is None: break
    n += 1
    if term: obs = env.reset()
dt = time.perf_counter() - t
print(f"N={N}: {n} step_dist calls in {dt:.3f}s = {n/dt:.0f} steps/s ({dt/n*1e6:.1f} us per call)")
