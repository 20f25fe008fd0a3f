"""VectorPSRS: many PerStateRejectionSampling environments behind one call per step.

The reference evaluator (offsim4rl/evaluators/per_state_rejection.py:7-98) is one environment driven by one Python call
per simulated step (examples/cartpole/psrs_from_expert_heuristic.py:59-80); a neural learner that reveals an action
distribution per step cannot be moved into a kernel.  What can be done for it is to step thousands of independent
environments (sampler seeds) with ONE launch: this class is `PerStateRejectionSampling` vectorised over `num_envs`,
every environment identical to the reference evaluator constructed from the same dataset and reset_sampler(seed_k).

    env = VectorPSRS(dataset, num_envs=4096, num_states=162, encoder=CartpoleBoxEncoder())
    env.reset_sampler(seeds)                  # PSRS.reset_sampler(seed_k) for every environment k
    obs, alive = env.reset()                  # [R, ...] device tensor, alive[k] False where reset() returned None
    a, next_obs, r, done, alive = env.step_dist_batch(probs)   # probs [R, nA] (tensor or torch Distribution)

Rows of environments that are exhausted (step_dist would return the all-None tuple) have alive == False and keep their
previous observation.  Everything stays on the device; no per-environment Python work and (with strict=False) no host synchronisation per call.
"""
import numpy as np
import torch

from .. import _lib as L
from ..spaces import is_discrete
from ..table import TransitionTable
from .psrs import BatchedPSRS

try:
    from torch.distributions import Distribution
except Exception:  # pragma: no cover
    class Distribution:  # type: ignore
        pass


class VectorPSRS:
    def __init__(self, dataset, num_envs, num_states=None, encoder=None, device=None, strict=False):
        # the validation of per_state_rejection.py:16-25
        if not is_discrete(dataset.observation_space) and num_states is None and encoder is None:
            raise ValueError("PerStateRejectionSampling only supports discrete observation spaces")
        if (num_states is None or encoder is None) and (num_states != encoder):
            raise ValueError("num_states and encoder either both need to be None, or both need to be specified")
        if not is_discrete(dataset.action_space):
            raise ValueError("PerStateRejectionSampling currently only supports discrete action spaces")
        e = dataset.experience
        if encoder is not None:
            zs, next_zs = np.asarray(encoder.encode(e["observations"])), np.asarray(encoder.encode(e["next_observations"]))
        else:
            zs, next_zs = np.asarray(e["observations"]), np.asarray(e["next_observations"])
        n = len(zs)
        t0 = (np.asarray(e["steps"]) == 0) if "steps" in e else None
        self.num_envs = int(num_envs)
        self.strict = bool(strict)  # raise KeyError like the reference (costs a host sync per call); otherwise such environments just stop
        self.table = TransitionTable(zs, e["actions"], e["rewards"], next_zs, e["terminals"],
                                     np.asarray(e["action_distributions"]).reshape(n, -1), t0, device=device)
        dev = self.table.device
        self.env = BatchedPSRS(self.table, self.num_envs)
        # payload columns in the caller's row order (the kernels report rows of the caller's buffer)
        self._obs = torch.as_tensor(np.asarray(e["observations"])).to(dev)
        self._next_obs = torch.as_tensor(np.asarray(e["next_observations"])).to(dev)
        self._a = torch.as_tensor(np.asarray(e["actions"])).to(dev)
        self._r = torch.as_tensor(np.asarray(e["rewards"])).to(dev)
        self._done = torch.as_tensor(np.asarray(e["terminals"]) != 0).to(dev)
        self.observation_space, self.action_space = dataset.observation_space, dataset.action_space
        self.obs = torch.zeros((self.num_envs,) + tuple(self._obs.shape[1:]), dtype=self._obs.dtype, device=dev)
        self.alive = torch.zeros(self.num_envs, dtype=torch.bool, device=dev)

    def reset_sampler(self, seeds):
        self.env.reset_sampler(seeds)
        self.alive.zero_()

    def reset(self, mask=None):
        """PSRS.reset (psrs.py:32-37) for the environments in `mask` (all if None).  Returns (obs [R, ...], alive [R])."""
        row = self.env.reset(mask).to(torch.int64)
        m = torch.ones_like(self.alive) if mask is None else mask.to(torch.bool)
        ok = m & (row >= 0)
        self.obs = torch.where(ok.reshape((-1,) + (1,) * (self.obs.dim() - 1)), self._obs[row.clamp(min=0)], self.obs)
        self.alive = torch.where(m, row >= 0, self.alive)
        return self.obs, self.alive

    def step_dist_batch(self, action_dists):
        """per_state_rejection.py:85-95 for every environment at once.  action_dists: [R, nA] probabilities (tensor / array /
        torch Distribution with .probs).  Returns device tensors (action, next_obs, reward, done, alive); entries of
        environments with alive == False are meaningless (the reference returns the all-None tuple there)."""
        if isinstance(action_dists, Distribution):
            action_dists = action_dists.probs
        row, status, _ = self.env.step(action_dists)
        if self.strict and bool((status == L.ST_KEYERROR).any()):  # psrs.py:44: the state has no queue
            k = int(torch.nonzero(status == L.ST_KEYERROR)[0])
            raise KeyError(self.table.z_of(int(self.env.state.cur_slot[k])))
        ok = status == L.ST_OK
        rr = row.to(torch.int64).clamp_(min=0)
        self.obs = torch.where(ok.reshape((-1,) + (1,) * (self.obs.dim() - 1)), self._next_obs[rr], self.obs)
        self.alive = self.alive & ok
        return self._a[rr], self.obs, self._r[rr], self._done[rr] & ok, self.alive
