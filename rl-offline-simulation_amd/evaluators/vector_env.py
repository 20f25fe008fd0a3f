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
previous observation.  Everything stays on the device; no per-environment Python work and (with strict=False) no host
synchronisation per call.  `obs` and `alive` are updated in place, and a call is nothing but kernel launches on the current
stream, so a whole driver iteration (policy forward -> step_dist_batch -> reset of the finished environments) can be
captured once in a HIP graph and replayed (`torch.cuda.graph`, see `graph_iteration`): the loop is launch-bound otherwise.
"""
import ctypes as C

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
        # outputs of step_dist_batch (overwritten by every call) and the column descriptors of offsim_vector_gather
        self._obs, self._next_obs, self._a, self._r, self._done = (x.contiguous() for x in (self._obs, self._next_obs, self._a, self._r, self._done))
        self.action = torch.zeros(self.num_envs, dtype=self._a.dtype, device=dev)
        self.reward = torch.zeros(self.num_envs, dtype=self._r.dtype, device=dev)
        self.done = torch.zeros(self.num_envs, dtype=torch.bool, device=dev)

        def cols(*pairs):
            arr = (L.Column * len(pairs))()
            for c, (src, dst, zero) in zip(arr, pairs):
                nb = src.element_size() * int(np.prod(src.shape[1:], dtype=np.int64))
                assert nb == dst.element_size() * int(np.prod(dst.shape[1:], dtype=np.int64))
                c.src, c.dst, c.row_bytes, c.zero_if_not_ok = src.data_ptr(), dst.data_ptr(), nb, int(zero)
            return arr

        self._cols_step = cols((self._next_obs, self.obs, False), (self._a, self.action, False), (self._r, self.reward, False), (self._done, self.done, True))
        self._cols_reset = cols((self._obs, self.obs, False))

    def reset_sampler(self, seeds):
        self.env.reset_sampler(seeds)
        self.alive.zero_()
        if self.strict:  # (strict mode synchronises with the host anyway: a sampler reset that gave up a bounded wait raises here)
            self.check_faults()

    def check_faults(self):
        """Synchronise and raise OffsimError if the sampler reset gave up a bounded wait (include/offsim.h: offsim_async_faults).  With
        strict=False nothing synchronises with the host, so the caller does this once after reset_sampler (or whenever it reads results)."""
        self.env.check_faults()

    def reset(self, mask=None):
        """PSRS.reset (psrs.py:32-37) for the environments in `mask` (all if None).  Returns (obs [R, ...], alive [R])."""
        m = None if mask is None else mask.to(torch.uint8).contiguous()
        row = self.env.reset(m)
        L.check(L.load().offsim_vector_gather(L.ptr(row), None, L.ptr(m), self.num_envs, self._cols_reset, len(self._cols_reset),
                                              L.ptr(self.alive), L.stream_ptr()))
        return self.obs, self.alive

    def step_dist_batch(self, action_dists):
        """per_state_rejection.py:85-95 for every environment at once.  action_dists: [R, nA] probabilities (tensor / array /
        torch Distribution with .probs).  Returns device tensors (action, next_obs, reward, done, alive) -- the environment's own
        buffers, overwritten by the next call (clone what has to outlive it); entries of environments with alive == False are
        meaningless (the reference returns the all-None tuple there; action / reward keep their last value, done is False)."""
        if isinstance(action_dists, Distribution):
            action_dists = action_dists.probs
        row, status, _ = self.env.step(action_dists)
        if self.strict and bool((status == L.ST_KEYERROR).any()):  # psrs.py:44: the state has no queue
            k = int(torch.nonzero(status == L.ST_KEYERROR)[0])
            raise KeyError(self.table.z_of(int(self.env.state.cur_slot[k])))
        L.check(L.load().offsim_vector_gather(L.ptr(row), L.ptr(status), None, self.num_envs, self._cols_step, len(self._cols_step),
                                              L.ptr(self.alive), L.stream_ptr()))
        return self.action, self.obs, self.reward, self.done, self.alive

    def step_and_reset(self, action_dists):
        """step_dist_batch followed by reset(mask=done) -- a whole driver iteration of the environments -- as ONE launch
        (offsim_vector_step).  Returns (action, obs, reward, done, alive): obs is the next observation, or the initial observation of
        the next episode where `done`; the same buffers as step_dist_batch's."""
        if isinstance(action_dists, Distribution):
            action_dists = action_dists.probs
        env, t = self.env, self.table
        env._orders_for_generic()
        p = action_dists if isinstance(action_dists, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(action_dists))
        f32 = p.dtype == torch.float32 and t.p_log.dtype == torch.float32
        p = p.to(device=t.device, dtype=torch.float32 if f32 else torch.float64).reshape(self.num_envs, t.nA).contiguous()
        L.check(L.load().offsim_vector_step(C.byref(t.c), C.byref(env.state.c), L.ptr(p), L.PROB_F32 if f32 else L.PROB_F64, env.reject_mode,
                                            self._cols_step, len(self._cols_step), self._cols_reset, len(self._cols_reset), L.ptr(self.alive),
                                            L.ptr(env._row), L.ptr(env._status), L.stream_ptr()))
        if self.strict and bool((env._status == L.ST_KEYERROR).any()):
            k = int(torch.nonzero(env._status == L.ST_KEYERROR)[0])
            raise KeyError(self.table.z_of(int(env.state.cur_slot[k])))
        self._keep = p
        return self.action, self.obs, self.reward, self.done, self.alive

    def graph_iteration(self, dist_fn, warmup=3, fused=True):
        """Capture one driver iteration in a HIP graph: probs = dist_fn(self.obs); step_and_reset(probs) (fused=False: the four
        launches of step_dist_batch(probs); reset(mask=done)).
        Returns (graph, (action, reward, done)): every graph.replay() advances all environments by one step and leaves the
        step's outputs in those three tensors (and in self.obs / self.alive).  `warmup` eager iterations run first on a side
        stream (PyTorch's capture recipe; they are real steps of the environments).  strict must be False."""
        if self.strict:
            raise ValueError("graph capture needs strict=False (the KeyError check synchronises with the host)")

        def iteration():
            if fused:  # one launch for the environments' whole iteration
                a, _, r, done, _ = self.step_and_reset(dist_fn(self.obs))
                return a, r, done
            a, _, r, done, _ = self.step_dist_batch(dist_fn(self.obs))
            self.reset(mask=done)
            return a, r, done

        side = torch.cuda.Stream(device=self.table.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):
                iteration()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g), torch.no_grad():
            outs = iteration()
        return g, outs
