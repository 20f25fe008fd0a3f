import os, sys, time
import numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
from rl_offline_simulation_amd import synth, _lib as L
from rl_offline_simulation_amd.table import TransitionTable
from rl_offline_simulation_amd.evaluators import BatchedPSRS
class A: pass
for wl, N in (("grid", 10_000_000), ("cartpole", 10_000_000)):
    a = A(); a.workload = wl; a.transitions = N; a.n_states = 162; a.n_actions = 2
    e, _ = bench.make_log(a, 20221107, torch.device("cuda", 0))
    t0 = e["steps"] == 0
    table = TransitionTable(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], t0)
    pi = table.policy_slots(synth.dirichlet_policy(a.n_states, a.n_actions))
    R = 48
    seeds = [int(x) for x in np.random.default_rng(3).integers(0, 1 << 62, R)]
    for form in ("streams", "permutations"):  # (both forms of the orders: OFFSIM_SCAN_ROWS forces the one or the other)
        os.environ["OFFSIM_SCAN_ROWS"] = "1" if form == "streams" else "0"
        out = {}
        for mode in ("1", "0"):
            os.environ["OFFSIM_SHUFFLE_CHUNKED"] = mode
            env = BatchedPSRS(table, R)
            torch.cuda.synchronize(); t1 = time.time()
            env.reset_sampler(seeds, policy=pi)
            torch.cuda.synchronize(); dt = time.time() - t1
            bufs = (env._dig_buf, env._loc_buf, env._init_perm_buf) if form == "streams" else (env.state.perm, env._init_perm_buf)
            out[mode] = (bufs, dt, L.load().offsim_async_faults())
        same = all(torch.equal(x, y) for x, y in zip(out["1"][0], out["0"][0]))
        print(wl, form, "max_seg", table.max_seg, "N0", table.N0, "equal", same, "faults", out["1"][2], out["0"][2], "t %.3f / %.3f" % (out["1"][1], out["0"][1]), flush=True)
        del out, env
        torch.cuda.empty_cache()
