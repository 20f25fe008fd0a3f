import os, sys, time
import numpy as np, torch
sys.path.insert(0, '/root/repo')
from rl_offline_simulation_amd import synth, _lib as L
from rl_offline_simulation_amd.table import TransitionTable
from rl_offline_simulation_amd.evaluators import BatchedPSRS
from oracle import oracle as O
for n, nS, R in [(70001, 1, 3), (131073, 1, 3), (300000, 2, 4), (1000000, 3, 4), (3000000, 1, 2)]:
    e = synth.synth_iid(n, nS, 2, seed=n)
    table = TransitionTable(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], e["steps"] == 0)
    pi = table.policy_slots(synth.dirichlet_policy(nS, 2))
    seeds = [5, 2**40 + 1, 77, 123456][:R]
    res = {}
    for mode in ("1", "0"):
        os.environ["OFFSIM_SHUFFLE_CHUNKED"] = mode
        env = BatchedPSRS(table, R)
        env.reset_sampler(seeds, policy=pi)
        torch.cuda.synchronize()
        t0 = time.time(); env.reset_sampler(seeds, policy=pi); torch.cuda.synchronize(); dt = time.time() - t0
        res[mode] = (env._dig_buf.clone(), env._loc_buf.clone(), getattr(env, "_ws", None) is not None, dt, L.load().offsim_async_faults())
    same = torch.equal(res["1"][0], res["0"][0]) and torch.equal(res["1"][1], res["0"][1])
    print(n, nS, "max_seg", table.max_seg, "chunked used", res["1"][2], "equal", same, "t chunked %.4f in-place %.4f" % (res["1"][3], res["0"][3]), "faults", res["1"][4], res["0"][4], flush=True)
    if os.environ.get("SHC_PROF"):
        env = BatchedPSRS(table, 1); os.environ["OFFSIM_SHUFFLE_CHUNKED"] = "1"
        env.reset_sampler([5], policy=pi); torch.cuda.synchronize()
        pf = env._ws[64:64 + 40].view(torch.int64).cpu().numpy()
        print("   100MHz ticks: phase I %d, phase II %d, identity+store %d, waiting for j %d, final pass %d" % tuple(pf), flush=True)
