"""HOMEREncoder.encode (offsim4rl/encoders/homer.py:159-168): forward of EncoderModel.obs_encoder
(offsim4rl/encoders/models.py:15-19) + argmax, on the device.  Training is out of scope; weights come
from a reference state_dict (keys obs_encoder.0.weight/.0.bias/.2.weight/.2.bias)."""
import numpy as np
import torch

from .. import _lib as L


class HOMEREncoder:
    def __init__(self, obs_dim, action_dim, latent_size, hidden_size, model_path=None, state_dict=None, device=None):
        self.obs_dim, self.latent_size, self.hidden_size = obs_dim, latent_size, hidden_size
        self.device = device
        self._w = None
        if model_path:
            state_dict = torch.load(model_path, map_location="cpu")
        if state_dict is not None:
            self.load_state_dict(state_dict)

    def load_state_dict(self, sd):
        dev = self.device or L.require_device()
        get = lambda k: torch.as_tensor(np.asarray(sd[k]) if not isinstance(sd[k], torch.Tensor) else sd[k]).to(dev, torch.float32).contiguous()
        W1, b1, W2, b2 = get("obs_encoder.0.weight"), get("obs_encoder.0.bias"), get("obs_encoder.2.weight"), get("obs_encoder.2.bias")
        assert W1.shape == (self.hidden_size, self.obs_dim) and W2.shape == (self.latent_size, self.hidden_size)
        self._w = (W1, b1, W2, b2)

    def encode_device(self, x, return_logits=False):
        if self._w is None:  # homer.py:160-161
            raise ValueError("Model not initialized. Either train a new model for the encoder or load an existing one.")
        W1, b1, W2, b2 = self._w
        if x.dtype not in (torch.float32, torch.float16):
            x = x.to(torch.float32)
        x = x.contiguous()
        N = x.shape[0]
        z = torch.empty(N, dtype=torch.int32, device=x.device)
        logits = torch.empty((N, self.latent_size), dtype=torch.float32, device=x.device) if return_logits else None
        L.check(L.load().offsim_encode_mlp(L.ptr(x), L.F16 if x.dtype == torch.float16 else L.F32, N, self.obs_dim, L.ptr(W1), L.ptr(b1),
                                           self.hidden_size, L.ptr(W2), L.ptr(b2), self.latent_size, L.ptr(z), L.ptr(logits), L.stream_ptr()))
        return (z, logits) if return_logits else z

    def encode(self, observations):
        """homer.py:159-168.  float16 observations (config C5) cross the PCIe bus as float16 and are widened by the kernel -- the
        reference's `.float()` (homer.py:163) is a cast on the device side too, and exact; every other dtype is cast to float32 on the
        host as before.  `last_input_dtype` records which instance of the kernel ran."""
        dev = self.device or L.require_device()
        obs = observations if isinstance(observations, torch.Tensor) else np.asarray(observations)
        half = obs.dtype in (np.float16, torch.float16)
        if isinstance(obs, torch.Tensor):
            x = obs.to(device=dev, dtype=torch.float16 if half else torch.float32)
        else:
            x = torch.from_numpy(np.ascontiguousarray(obs, dtype=np.float16 if half else np.float32)).to(dev)
        self.last_input_dtype = x.dtype
        return self.encode_device(x.reshape(x.shape[0], -1)).cpu().numpy().astype(np.int64)
