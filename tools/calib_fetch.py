"""Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE on this GPU for the two access shapes the PSRS kernels use:
(a) wide coalesced streaming, (b) one 4-byte word per random 64-byte sector / per random 128-byte line.  Run under
rocprofv3 --pmc FETCH_SIZE (and WRITE_SIZE); compare the counters of the marked kernels with the byte counts printed."""
import torch
dev = torch.device("cuda", 0)
n = 1 << 28                       # 1 GiB of float32: well past the 256 MiB Infinity Cache
x = torch.arange(n, dtype=torch.float32, device=dev)
torch.cuda.synchronize()
y = x.clone()                     # (a) streaming copy: reads 4n bytes, writes 4n bytes
torch.cuda.synchronize()
m = 1 << 22
g = torch.Generator(device=dev); g.manual_seed(0)
sec = torch.randperm(n // 16, device=dev, generator=g)[:m]            # distinct 64-byte sectors
idx64 = sec * 16
z = torch.index_select(x, 0, idx64)                                    # (b1) 4 B from each of m random 64-B sectors
torch.cuda.synchronize()
line = torch.randperm(n // 32, device=dev, generator=g)[:m]
idx128 = line * 32
w = torch.index_select(x, 0, idx128)                                   # (b2) 4 B from each of m random 128-B lines
torch.cuda.synchronize()
print(f"stream copy: {4*n} B read, {4*n} B written; gather: {m} words = {64*m} B of 64-B sectors / {128*m} B of 128-B lines; index bytes {8*m}")
