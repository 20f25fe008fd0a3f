"""In-kernel cycle stamps of the row-packed scan (library built with -DOFFSIM_ROWS_PROF): where an iteration's cycles go."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl_offline_simulation_amd import synth
from rl_offline_simulation_amd.table import TransitionTable
from rl_offline_simulation_amd.evaluators import BatchedPSRS
N, R = int(sys.argv[1]), int(sys.argv[2])
e = synth.synth_iid(N, 162, 2, seed=20221107)
table = TransitionTable(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], e["steps"] == 0)
pi = table.policy_slots(synth.dirichlet_policy(162, 2))
env = BatchedPSRS(table, R)
env.reset_sampler(list(range(R)), policy=pi)
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record(); o = env.eval_mc(pi, 0.99, dbg=True, ep_cap=24); t1.record(); torch.cuda.synchronize()
d = o["dbg"].cpu().numpy()[::16]  # one lane-0 row per wavefront is enough (all rows of a wave stamp the same clock)
steps = o["steps"].cpu().numpy().astype(float)
it = steps.max()
fast, slow, tick = d[:, 0].mean(), d[:, 1].mean(), d[:, 2].mean()
nslow = (d[:, 3] & 0xffffffff).mean(); ndry = (d[:, 3] >> 32).mean()
print(f"kernel {t0.elapsed_time(t1):.1f} ms, iterations/wave ~{it:.0f}")
print(f"memtime ticks per iteration: fast {fast / it:.1f} slow {slow / it:.1f} tick {tick / it:.1f}  (total {(fast + slow + tick) / it:.1f})")
print(f"slow iterations {nslow / it * 100:.2f} % of all, {slow / max(nslow, 1):.0f} ticks each; tick() {tick / (it / 16):.0f} ticks each; dry events/row {ndry:.0f}")
ph = o["ep_g"].cpu().numpy()[::16].mean(axis=0)
names = ["batch reads + flag", "land reads + hand-off", "landing", "-", "-", "-", "-", "counters, init prefetch, draws check"]
print(f"chain wavefront: {ph[10] / it:.1f} cycles per iteration in all, clock {ph[10] / ph[11] * 0.1:.3f} GHz; slowest / mean wavefront {o['ep_g'].cpu().numpy()[::16, 10].max() / ph[10]:.4f}")
print("chain tick phases (cycles per tick):", {n: int(v / (it / 16)) for n, v in zip(names, ph[:8]) if n != "-"})
sn = {3: "rows with a clear accept commit", 4: "held + land read", 5: "segment, draws check, address", 6: "candidate load returns", 8: "look, commit, reset", 9: "draws check"}
print("chain slow iteration phases (cycles per slow iteration; stamps inside divergent code count once per wavefront):", {v: int(ph[k] / max(nslow, 1)) for k, v in sn.items()})
hn = {8: "poll (idle)", 9: "log read + requests", 3: "slot reads", 4: "R3 sums", 5: "R2 rewards", 6: "R1 loc/discount", 10: "draws"}
print("helper phases (cycles per tick):", {v: int(ph[12 + k] / (it / 16)) for k, v in hn.items()})
# spread between workgroups (one sampled wavefront each): the kernel ends with its slowest one
g = o["ep_g"].cpu().numpy()[::16]
tot, rt = g[:, 10], g[:, 11]
q = np.percentile(tot, [0, 10, 50, 90, 99, 100]) / tot.mean()
print("workgroup total cycles / mean at percentiles 0 10 50 90 99 100:", np.round(q, 4))
for m in (2, 8, 32):
    print(f"by block % {m}: cycles/mean", np.round([tot[b::m].mean() / tot.mean() for b in range(m)], 3))
print("by block % 8: clock GHz", np.round([(tot[b::8] / rt[b::8]).mean() * 0.1 for b in range(8)], 3), " wall ms", np.round([rt[b::8].mean() * 1e-5 for b in range(8)], 1))
for par in (0, 1):
    sel = slice(par, None, 2)
    print(f"blocks of parity {par}: fast {d[sel, 0].mean() / it:.1f} slow {d[sel, 1].mean() / it:.1f} tick {d[sel, 2].mean() / it:.1f} nslow {(d[sel, 3] & 0xffffffff).mean():.0f} per-slow {d[sel, 1].mean() / (d[sel, 3] & 0xffffffff).mean():.0f}")
order = np.argsort(tot)
st16 = steps.reshape(-1, 16)
for w in list(order[-5:]) + list(order[:3]):
    print(f"  block {w:3d}: total {tot[w] / tot.mean():.4f}  fast {d[w, 0] / it:.1f} slow {d[w, 1] / it:.1f} tick {d[w, 2] / it:.1f}  nslow {(d[w, 3] & 0xffffffff)}  steps of wave 0 {st16[w, :4].astype(int).tolist()}")
ga = o["ep_g"].cpu().numpy()[::16]
for par in (0, 1):
    phs = ga[par::2].mean(axis=0)
    print(f"parity {par}: chain tick phases", {n: int(v / (it / 16)) for n, v in zip(names, phs[:8]) if n != "-"}, " helper", {v: int(phs[12 + k] / (it / 16)) for k, v in hn.items()})
