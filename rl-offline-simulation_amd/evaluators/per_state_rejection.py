"""PerStateRejectionSampling: the drop-in evaluator of offsim4rl/evaluators/per_state_rejection.py:7-98,
backed by the device table and HIP kernels (R = 1 rollout).  Same constructor, same errors, same
return tuples; the latent states come from `encoder.encode` (any object with that method, including
this package's device encoders)."""
import numpy as np
import torch

from .. import _lib as L
from ..core import RevealedRandomnessEnv
from ..spaces import is_discrete
from .psrs import PSRS

try:
    from torch.distributions import Distribution
except Exception:  # pragma: no cover
    class Distribution:  # type: ignore
        pass


class PerStateRejectionSampling(RevealedRandomnessEnv):
    def __init__(self, dataset, num_states=None, encoder=None, new_step_api=False):
        # per_state_rejection.py:16-25
        if not is_discrete(dataset.observation_space) and num_states is None and encoder is None:
            raise ValueError("PerStateRejectionSampling only supports discrete observation spaces")
        if (num_states is None or encoder is None) and (num_states != encoder):
            raise ValueError("num_states and encoder either both need to be None, or both need to be specified")
        if not is_discrete(dataset.action_space):
            raise ValueError("PerStateRejectionSampling currently only supports discrete action spaces")
        self._dataset = dataset
        e = dataset.experience
        if encoder is not None:  # :29-31
            zs = np.asarray(encoder.encode(e["observations"]))
            next_zs = np.asarray(encoder.encode(e["next_observations"]))
        else:  # :33-35 discrete observations are the states
            zs, next_zs = np.asarray(e["observations"]), np.asarray(e["next_observations"])
        n = len(zs)
        nA = dataset.action_space.n
        if "action_distributions" not in e:
            raise ValueError("PerStateRejectionSampling needs action_distributions (the logging policy's probabilities)")
        t0 = (np.asarray(e["steps"]) == 0) if "steps" in e else None  # data.py:72: no steps => every row initial
        # subclasses may override _reject (trivial_baselines.py); the built-in rules run on the device
        rule = getattr(type(self), "_device_reject_mode", None)
        overridden = type(self)._reject is not PerStateRejectionSampling._reject
        self._impl = PSRS.from_arrays(
            zs, e["actions"], e["rewards"], next_zs, e["terminals"], np.asarray(e["action_distributions"]).reshape(n, -1),
            t0=t0, nS=num_states if num_states is not None else 25, nA=nA,
            reject_func=(self._reject if (overridden and rule is None) else None),
            obs=e["observations"], next_obs=e["next_observations"], reject_mode=rule)
        self.new_step_api = new_step_api

    @property
    def observation_space(self):
        return self._dataset.observation_space

    @property
    def action_space(self):
        return self._dataset.action_space

    def reset_sampler(self, seed=None):
        return self._impl.reset_sampler(seed=seed)

    def reset(self, seed=None):
        return self._impl.reset(seed=seed)

    def step(self, action):
        raise NotImplementedError(
            f"{self.__class__.__name__} does not support step(). To implement Per-State Rejection "
            "Sampling efficiently, your agent needs to reveal its action distribution via the step_dist() method instead.")

    def step_dist(self, action_dist):
        """per_state_rejection.py:85-95"""
        if isinstance(action_dist, Distribution):
            action_dist = action_dist.probs
        next_obs, r, done, info = self._impl.step(action_dist)
        dones = [done, False] if self.new_step_api else [done]
        if next_obs is None:
            return (None,) * (6 if self.new_step_api else 5)
        return (info["a"], next_obs, r, *dones, info)

    def _reject(self, p_new, p_log, a) -> bool:
        return self._impl._default_reject(p_new, p_log, a)
