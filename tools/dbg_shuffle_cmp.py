"""Checks offsim_shuffle_queues against numpy (default_rng(seed).shuffle per queue, psrs.py:22-30) on several shapes,
then times it.  Usage: dbg_shuffle_cmp.py [time_N time_R time_nS]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rl_offline_simulation_amd import synth
from rl_offline_simulation_amd.table import TransitionTable, seeds_tensor, shuffle_queues

dev = torch.device("cuda", 0)


def expect(table, seed):
    so = table.seg_off.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    out = np.arange(table.N, dtype=np.int64)
    for s in range(table.n_slots):
        q = list(range(so[s], so[s + 1]))
        np.random.default_rng(seed).shuffle(q)
        out[so[s]:so[s + 1]] = q
    iq = list(range(table.N0))
    np.random.default_rng(seed).shuffle(iq)
    return out, np.asarray(iq, dtype=np.int64)


def check(N, nS, seeds, p_t0=None, skew=False):
    e = synth.synth_iid(N, nS, 2, seed=N + nS)
    if skew:  # a few huge states and many tiny ones
        rng = np.random.default_rng(5)
        e["z"] = np.where(rng.random(N) < 0.7, 0, rng.integers(0, nS, N)).astype(e["z"].dtype)
    t0 = e["steps"] == 0 if p_t0 is None else np.random.default_rng(3).random(N) < p_t0
    table = TransitionTable(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], t0, device=dev)
    sd = seeds_tensor(np.asarray(seeds, dtype=np.uint64), dev)
    perm, ip = shuffle_queues(table, sd)
    torch.cuda.synchronize()
    perm, ip = perm.cpu().numpy().astype(np.int64) & 0xFFFFFFFF, ip.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    bad = 0
    for k, seed in enumerate(seeds):
        ep, ei = expect(table, int(seed))
        if not np.array_equal(perm[k][:table.N], ep):
            bad += 1
            w = np.nonzero(perm[k][:table.N] != ep)[0]
            print(f"  MISMATCH perm N={N} nS={nS} seed={seed}: {len(w)} positions, first {w[:5]}")
        if table.N0 and not np.array_equal(ip[k][:table.N0], ei):
            bad += 1
            w = np.nonzero(ip[k][:table.N0] != ei)[0]
            print(f"  MISMATCH init N={N} nS={nS} seed={seed}: {len(w)} positions, first {w[:5]}")
    print(f"N={N} nS={nS} max_seg={table.max_seg} N0={table.N0} seeds={len(seeds)}: {'OK' if not bad else 'FAILED'}")
    return bad


bad = 0
bad += check(2, 1, [0, 1])
bad += check(50, 3, [0, 1, 2, 3])
bad += check(1000, 7, [0, 5, 2**40 + 3])
bad += check(20000, 2, [1, 2, 3])
bad += check(70000, 1, [0, 9], p_t0=1.0)       # one state of 70000 rows and an init queue of 70000: the global-memory path
bad += check(200000, 3, [4, 5])                 # states ~66k: around the LDS capacity
bad += check(300000, 40, [7], skew=True)
bad += check(131072, 2, [11, 12], p_t0=0.5)
print("ALL OK" if not bad else f"{bad} FAILURES")
if len(sys.argv) > 3:
    N, R, nS = int(float(sys.argv[1])), int(sys.argv[2]), int(sys.argv[3])
    e = synth.synth_iid(N, nS, 2)
    table = TransitionTable(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], e["steps"] == 0, device=dev)
    sd = seeds_tensor(np.arange(R), dev)
    perm = torch.empty((R, N), dtype=torch.int32, device=dev); ip = torch.empty((R, table.N0), dtype=torch.int32, device=dev)
    for it in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        shuffle_queues(table, sd, perm, ip)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        print(f"shuffle N={N} R={R} nS={nS}: {dt:.3f}s  {N*R/dt/1e9:.2f} G swaps/s")
