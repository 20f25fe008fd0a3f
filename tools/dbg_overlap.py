"""Experiment: does reset_sampler (shuffle) of one rollout tile overlap with the evalMC scan of another?
Two BatchedPSRS tiles on two HIP streams; prints solo and concurrent durations."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl_offline_simulation_amd import synth
from rl_offline_simulation_amd.table import TransitionTable
from rl_offline_simulation_amd.evaluators import BatchedPSRS

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
e = synth.synth_iid(N, 162, 2, seed=20221107)
pi = synth.dirichlet_policy(162, 2)
table = TransitionTable(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], e["steps"] == 0)
ps = table.policy_slots(pi)
A, B = BatchedPSRS(table, R), BatchedPSRS(table, R)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
seedsA, seedsB = np.arange(R, dtype=np.uint64), np.arange(R, 2 * R, dtype=np.uint64)


def timed(fn, stream):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        e0.record()
        r = fn()
        e1.record()
    return e0, e1, r


# warm / solo
for _ in range(2):
    x = timed(lambda: A.reset_sampler(seedsA), sa); torch.cuda.synchronize(); t_shufA = x[0].elapsed_time(x[1])
    x = timed(lambda: A.eval_mc(ps, 0.99), sa); torch.cuda.synchronize(); t_scanA = x[0].elapsed_time(x[1]); stepsA = int(x[2]["steps"].sum())
print(f"solo: shuffle {t_shufA:.0f} ms, scan {t_scanA:.0f} ms, steps {stepsA}")
# concurrent: scan A (needs shuffled A) with shuffle B
A.reset_sampler(seedsA); torch.cuda.synchronize()
for order in ("scan_first", "shuffle_first"):
    A.state.rewind(); A.set_rejection_seeds(seedsA); torch.cuda.synchronize()
    t0 = time.perf_counter()
    if order == "scan_first":
        xs = timed(lambda: A.eval_mc(ps, 0.99), sa)
        xb = timed(lambda: B.reset_sampler(seedsB), sb)
    else:
        xb = timed(lambda: B.reset_sampler(seedsB), sb)
        xs = timed(lambda: A.eval_mc(ps, 0.99), sa)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    print(f"{order}: scan {xs[0].elapsed_time(xs[1]):.0f} ms, shuffle {xb[0].elapsed_time(xb[1]):.0f} ms, wall {wall:.0f} ms, steps {int(xs[2]['steps'].sum())}")
