"""HBM-roofline characterisation of the encoder forward (SURVEY 8 a10 / BASELINE configs C3 and C5): rows/s and algorithmic
bytes/s of offsim_encode_mlp (csrc/encode_mfma.hpp) with observations resident in HBM.  Prints one JSON line per shape.

usage: python tools/bench_encoder.py [rows]"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rl_offline_simulation_amd.encoders import HOMEREncoder

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
dev = torch.device("cuda", 0)
g = np.random.default_rng(0)
for name, dO, H, nZ, dt in (("C3: 2-64-25, f32 observations", 2, 64, 25, torch.float32), ("C5: 128-64-50, fp16 observations", 128, 64, 50, torch.float16)):
    n = N if dO <= 8 else N // 4
    W1, b1 = g.standard_normal((H, dO)).astype(np.float32) / np.sqrt(dO), g.standard_normal(H).astype(np.float32) * 0.1
    W2, b2 = g.standard_normal((nZ, H)).astype(np.float32) / np.sqrt(H), g.standard_normal(nZ).astype(np.float32) * 0.1
    enc = HOMEREncoder(dO, 4, nZ, H, state_dict={"obs_encoder.0.weight": W1, "obs_encoder.0.bias": b1, "obs_encoder.2.weight": W2, "obs_encoder.2.bias": b2})
    x = torch.randn((n, dO), device=dev, dtype=torch.float32).to(dt)
    enc.encode_device(x)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    t0.record()
    for _ in range(reps):
        z = enc.encode_device(x)
    t1.record()
    torch.cuda.synchronize()
    s = t0.elapsed_time(t1) / 1e3 / reps
    nbytes = n * (dO * x.element_size() + 4)
    flops = 2.0 * n * (dO * H + H * nZ)
    print(json.dumps({"tool": "bench_encoder", "shape": name, "rows": n, "s_per_call": s, "rows_per_s": n / s,
                      "algorithmic_bytes_per_row": dO * x.element_size() + 4, "achieved_GBps": nbytes / s / 1e9, "frac_of_8TBps": nbytes / s / 8e12,
                      "TFLOPs": flops / s / 1e12}))
