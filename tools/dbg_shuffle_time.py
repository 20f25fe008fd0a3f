import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rl_offline_simulation_amd import synth
from rl_offline_simulation_amd.table import TransitionTable, seeds_tensor, shuffle_queues
dev = torch.device("cuda", 0)
N, R, nS = int(float(sys.argv[1])), int(sys.argv[2]), int(sys.argv[3])
e = synth.synth_iid(N, nS, 2)
table = TransitionTable(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], e["steps"] == 0, device=dev)
sd = seeds_tensor(np.arange(R), dev)
perm = torch.empty((R, N), dtype=torch.int32, device=dev); ip = torch.empty((R, table.N0), dtype=torch.int32, device=dev)
for it in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    shuffle_queues(table, sd, perm, ip)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
print(f"mode={os.environ.get('OFFSIM_SHUFFLE_DBG','0')} shuffle N={N} R={R} nS={nS}: {dt:.3f}s  {N*R/dt/1e9:.2f} G swaps/s")
