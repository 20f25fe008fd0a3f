"""CartpoleBoxEncoder (offsim4rl/encoders/heuristic.py:10-71) on the device: Sutton's 162 boxes, -1 out of bounds."""
import numpy as np
import torch

from .. import _lib as L


class CartpoleBoxEncoder:
    N_BOXES = 162

    def encode_device(self, observations):
        """observations: [N,4] float32 tensor on the GPU -> int32 tensor [N]."""
        obs = observations.to(torch.float32).contiguous()
        out = torch.empty(obs.shape[0], dtype=torch.int32, device=obs.device)
        L.check(L.load().offsim_encode_box(L.ptr(obs), obs.shape[0], L.ptr(out), L.stream_ptr()))
        return out

    def encode(self, observations):
        dev = L.require_device()
        obs = torch.from_numpy(np.ascontiguousarray(np.asarray(observations), np.float32)).to(dev)
        return self.encode_device(obs).cpu().numpy().astype(np.int64)
