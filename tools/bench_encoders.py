"""Encoder throughput on the device (rows a10, a11): achieved GB/s against the HBM roofline.
C3 shape: 10 M x 2 f32 observations, 2 -> 64 -> 25;  C5 shape: 128-d fp16 observations, 128 -> 64 -> 50 (N bounded by memory)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rl_offline_simulation_amd.encoders import CartpoleBoxEncoder, HOMEREncoder
dev = torch.device("cuda", 0)
g = np.random.default_rng(0)

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps

def mlp(N, dO, H, nZ, dtype):
    sd = {"obs_encoder.0.weight": g.standard_normal((H, dO)).astype(np.float32), "obs_encoder.0.bias": g.standard_normal(H).astype(np.float32),
          "obs_encoder.2.weight": g.standard_normal((nZ, H)).astype(np.float32), "obs_encoder.2.bias": g.standard_normal(nZ).astype(np.float32)}
    enc = HOMEREncoder(dO, 5, nZ, H, state_dict=sd)
    x = torch.randn((N, dO), device=dev, dtype=torch.float32).to(dtype)
    dt = timeit(lambda: enc.encode_device(x))
    byt = N * (dO * x.element_size() + 4)
    flop = 2.0 * N * (dO * H + H * nZ)
    print(f"mlp {dO}->{H}->{nZ} {str(dtype)[6:]} N={N}: {dt*1e3:.2f} ms  {byt/dt/1e9:.0f} GB/s ({byt/dt/8e12*100:.1f}% of 8 TB/s)  {flop/dt/1e12:.2f} TFLOP/s f32-MFMA")

N = 10_000_000
obs = torch.randn((N, 4), device=dev) * torch.tensor([1.5, 1.0, 0.15, 1.0], device=dev)
box = CartpoleBoxEncoder()
dt = timeit(lambda: box.encode_device(obs))
print(f"box encoder N={N}: {dt*1e3:.2f} ms  {N*20/dt/1e9:.0f} GB/s ({N*20/dt/8e12*100:.1f}% of 8 TB/s)")
mlp(10_000_000, 2, 64, 25, torch.float32)
mlp(20_000_000, 128, 64, 50, torch.float16)
mlp(10_000_000, 128, 64, 50, torch.float32)
