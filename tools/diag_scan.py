"""Where a scan's time goes on one of bench.py's workloads: kernel ms, the spread of steps per rollout (the kernel ends with its longest
rollout), dry windows / exact looks per rollout (eval_mc(dbg=True)), for the kernel the table would get or a forced one.
usage: diag_scan.py WORKLOAD TRANSITIONS ROLLOUTS [rows|win]   (WORKLOAD: iid | cartpole | grid | obs128)"""
import os, sys, types
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 4:
    os.environ["OFFSIM_SCAN_ROWS"] = "1" if sys.argv[4] == "rows" else "0"
import bench
from rl_offline_simulation_amd import synth
from rl_offline_simulation_amd.table import TransitionTable
from rl_offline_simulation_amd.evaluators import BatchedPSRS
a = types.SimpleNamespace(workload=sys.argv[1], transitions=int(sys.argv[2]), n_states=162, n_actions=2)
R = int(sys.argv[3])
dev = torch.device("cuda", 0)
e, _ = bench.make_log(a, 20221107, dev)
table = TransitionTable(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], e["steps"] == 0, device=dev)
pi = table.policy_slots(synth.dirichlet_policy(a.n_states, a.n_actions))
env = BatchedPSRS(table, R)
so = (table.seg_off.to(torch.int64) & 0xFFFFFFFF).cpu().numpy()
lens = np.diff(so)
print(f"{a.workload}: N={table.N} states={table.n_slots} nA={table.nA} largest states (share of rows): {np.round(np.sort(lens)[::-1][:6] / table.N, 3)}  N0={table.N0}")
for rep in range(2):
    t0, t1, t2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    t0.record(); env.reset_sampler(list(range(R)), policy=pi); t1.record(); o = env.eval_mc(pi, 0.99, dbg=True); t2.record(); torch.cuda.synchronize()
st, cd = o["steps"].cpu().numpy().astype(float), o["cand"].cpu().numpy().astype(float)
raw = o["dbg"].cpu().numpy()
print(f"kernel {env.scan_variant()}: reset {t0.elapsed_time(t1):.1f} ms, scan {t1.elapsed_time(t2):.1f} ms; steps per rollout min/mean/max {st.min():.0f}/{st.mean():.0f}/{st.max():.0f}; "
      f"candidates per step {cd.sum() / st.sum():.2f}; ns per step of the LONGEST rollout {t1.elapsed_time(t2) * 1e6 / st.max():.0f}")
if env.scan_variant() == "k_eval_mc_rows":
    cyc = raw[:, 2].astype(float)
    print(f"  rows kernel: {cyc.mean() / st.max():.0f} cycles per iteration, rows without a clear accept per rollout {(raw[:, 0] & 0xffffffff).mean():.0f}, exact looks {(raw[:, 1] & 0xffff).mean():.0f}")
else:
    print(f"  window kernel: dry windows per rollout {raw[:, 0].mean():.0f} ({raw[:, 0].mean() / st.mean() * 100:.2f} % of the steps), ties {raw[:, 1].mean():.1f}, flushes {raw[:, 2].mean():.0f}")
