"""QueueEvaluator (offsim4rl/evaluators/queue_evaluator.py:8-131) on the device tables.

Queues are keyed by (z, a); the agent samples its own action and the simulator pops the head of queue (z, a) -- no
rejection.  That is the PSRS machinery with the composite key z*nA + a as the grouping state and the "always accept,
pop one" rule: offsim_group_by_state / offsim_shuffle_queues / offsim_env_set_state / offsim_step_batch are reused as is.
"""
import os

import numpy as np
import torch

from .. import _lib as L
from ..spaces import is_discrete
from ..table import TransitionTable, gather_rows
from .psrs import BatchedPSRS, SHUFFLE_PER_ROLLOUT


class BatchedQueueEvaluator:
    """R independent QueueEvaluator_impl environments (queue_evaluator.py:90-131) over one table."""

    def __init__(self, z, a, r, z_next, done, p_log, t0=None, R=1, device=None):
        z = np.asarray(z, np.int64)
        a = np.asarray(a, np.int64)
        z_next = np.asarray(z_next, np.int64)
        self.nA = int(np.asarray(p_log).reshape(len(z), -1).shape[1]) if len(z) else 1
        self.z_lo = int(min(z.min(), z_next.min(), 0)) if len(z) else 0
        nA = self.nA
        comp = (z - self.z_lo) * nA + a                      # sorted(buffer, key=(z, a)), queue_evaluator.py:105
        comp_next = (z_next - self.z_lo) * nA                # base slot of the next state; the action is added per step
        self.table = TransitionTable(comp, a, r, comp_next, done, p_log, t0, device=device)
        t = self.table
        # the init queue holds (z, s): its slot is the base slot of z, not the composite of the logged action
        t.init_slot = (torch.div(t.init_slot, nA, rounding_mode="floor") * nA).to(torch.int32).contiguous()
        t.c.init_slot = L.ptr(t.init_slot)
        self.R = int(R)
        self.env = BatchedPSRS(t, R, L.REJECT_NEVER)
        self._dummy_p = torch.zeros((R, nA), dtype=torch.float64, device=t.device)

    def reset_sampler(self, seeds):
        self.env.reset_sampler(seeds, SHUFFLE_PER_ROLLOUT)

    def reset(self, mask=None):
        return self.env.reset(mask)

    def step(self, actions):
        """actions [R] int: pops the head of queue (z, a) for every rollout; returns device tensors (row, status).
        status: 0 ok, 1 queue empty (None), 3 KeyError (no such (z, a) queue), 4 no current state."""
        st = self.env.state
        act = torch.as_tensor(np.asarray(actions), dtype=torch.int32).to(self.table.device)
        base = st.cur_slot
        comp = torch.where(base >= 0, base + act, base).to(torch.int32).contiguous()
        self.env.set_state(comp)
        row, status, _ = self.env.step(self._dummy_p, advance=True, reject_mode=L.REJECT_NEVER)
        # a rollout whose pop failed keeps its state (queue_evaluator.py:122-123): undo the composite slot
        failed = status != L.ST_OK
        if bool(failed.any()):
            self.env.set_state(base.clone(), mask=failed)
        return row, status


class QueueEvaluator:
    """Drop-in for offsim4rl.evaluators.queue_evaluator.QueueEvaluator (queue_evaluator.py:8-88)."""

    def __init__(self, dataset, num_states=None, encoder=None):
        if not is_discrete(dataset.observation_space) and num_states is None and encoder is None:
            raise ValueError("QueueEvaluator only supports discrete observation spaces")
        if (num_states is None or encoder is None) and (num_states != encoder):
            raise ValueError("num_states and encoder either both need to be None, or both need to be specified")
        if not is_discrete(dataset.action_space):
            raise ValueError("QueueEvaluator currently only supports discrete action spaces")
        self._dataset = dataset
        e = dataset.experience
        if encoder is not None:
            zs, next_zs = np.asarray(encoder.encode(e["observations"])), np.asarray(encoder.encode(e["next_observations"]))
        else:
            zs, next_zs = np.asarray(e["observations"]), np.asarray(e["next_observations"])
        n = len(zs)
        t0 = (np.asarray(e["steps"]) == 0) if "steps" in e else None
        p_log = np.asarray(e["action_distributions"]).reshape(n, -1) if "action_distributions" in e else np.zeros((n, dataset.action_space.n))
        self._impl = BatchedQueueEvaluator(zs, e["actions"], e["rewards"], next_zs, e["terminals"], p_log, t0, R=1)
        self._z, self._zn = zs, next_zs
        self.nS = num_states if num_states is not None else 25
        self.nA = dataset.action_space.n
        self.s, self.z = None, None
        self.reset_sampler()
        self.reset()

    @property
    def observation_space(self):
        return self._dataset.observation_space

    @property
    def action_space(self):
        return self._dataset.action_space

    def reset_sampler(self, seed=None):
        if seed is None:
            seed = int.from_bytes(os.urandom(8), "little")
        self._impl.reset_sampler([seed])

    def reset(self, seed=None):
        row = int(self._impl.reset().cpu()[0])
        if row < 0:
            self.s = None
            return None
        self.z = int(self._z[row])
        self.s = self._dataset.experience["observations"][row]
        return self.s

    def step(self, action):
        z = self.z
        a = int(action)
        if a < 0 or a >= self._impl.nA:
            raise KeyError((z, a))
        row, status = self._impl.step([a])
        row, status = int(row.cpu()[0]), int(status.cpu()[0])
        if status == L.ST_KEYERROR:
            raise KeyError((z, a))
        if status != L.ST_OK:
            return None, None, None, None
        e = self._dataset.experience
        self.s, self.z = e["next_observations"][row], int(self._zn[row])
        p = e["action_distributions"][row] if "action_distributions" in e else None
        return self.s, e["rewards"][row], bool(e["terminals"][row]), {"z": z, "next_z": self.z, "a": int(e["actions"][row]), "p": p}

    def step_dist(self, action_dist):
        a = action_dist.sample()
        next_obs, r, done, info = self.step(a)
        if next_obs is None:
            return None, None, None, None, None
        return info["a"], next_obs, r, done, info
