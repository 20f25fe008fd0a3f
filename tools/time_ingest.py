"""One-time ingest cost (host arrays -> grouped SoA table in HBM), not part of the timed bench region."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rl_offline_simulation_amd import synth
from rl_offline_simulation_amd.table import TransitionTable
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
e = synth.synth_iid(N, 162, 2, seed=20221107)
cols = (e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], e["steps"] == 0)
nbytes = sum(np.asarray(c).nbytes for c in cols)
for it in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    table = TransitionTable(*cols)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
print(f"ingest N={N}: {dt*1e3:.1f} ms for {nbytes/1e6:.0f} MB of host columns = {nbytes/dt/1e9:.2f} GB/s (H2D copies + group-by + gathers)")
