"""In-kernel clock stamps of the keyed shuffle's roles (library built with -DSHUF_PROF: the stamps overwrite the first
digests of every chain's stream, so such a build is for this script only).  usage: prof_shuffle.py [N] [R]"""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl_offline_simulation_amd import synth
from rl_offline_simulation_amd.table import TransitionTable
from rl_offline_simulation_amd.evaluators import BatchedPSRS
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
e = synth.synth_iid(N, 162, 2, seed=20221107)
t = TransitionTable(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], e["steps"] == 0)
pi = t.policy_slots(synth.dirichlet_policy(162, 2))
env = BatchedPSRS(t, R)
seeds = np.arange(R, dtype=np.uint64)
for k in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    env.reset_sampler(seeds, policy=pi)
    torch.cuda.synchronize(); dt = time.time() - t0
print(f"reset_sampler {dt:.4f} s")
seg = t.seg_off.cpu().numpy().astype(np.int64)
dig = env._streams["dig"]
rows = torch.arange(0, R, 37, device=dig.device)
acc = [dig[rows, seg[s]:seg[s] + 13].cpu().numpy().astype(np.int64) & 0xffffffff for s in range(0, len(seg) - 1, 9) if seg[s + 1] - seg[s] >= 64]
m = (np.concatenate(acc).astype(float) * 64).mean(axis=0)
print("clocks per chain (workgroup):", int(m[12]))
for w, (name, extra) in enumerate([("G0", "writing chunks out"), ("C", "waiting for room in the j ring"), ("A", "groups with a conflict"), ("G1", "writing chunks out")]):
    print(f"{name}: loop {int(m[3 * w])}, waiting for its neighbour {int(m[3 * w + 1])}, {extra} {int(m[3 * w + 2])}")
