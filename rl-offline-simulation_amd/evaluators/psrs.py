"""Per-State Rejection Sampling on the MI355X: host-side mirror of offsim4rl/evaluators/psrs.py.

  PSRS          the reference's single-environment class (psrs.py:5-57), as BatchedPSRS (batched.py) with R = 1
  evalMC_psrs   psrs.py:241-271; one kernel launch when given a device-backed env
  qlearn_psrs / expSARSA_psrs   psrs.py:119-239, whole on the device (offsim_eval_td)

The batched engine itself -- BatchedPSRS, evalmc_rollouts, the stream layouts -- lives in batched.py and is re-exported here (callers
and tests import it from either).  No CPU fallback.
"""
import os

import numpy as np
import torch

from .. import _lib as L
from ..table import TransitionTable
from .batched import (BatchedPSRS, SHUFFLE_NONE, SHUFFLE_PER_ROLLOUT, SHUFFLE_SHARED, ROWS_MAX_ALL_REJECTED, ROWS_MAX_WINDOW_LOAD,  # noqa: F401
                      ROWS_TICK_STEPS, _gamma_pow, _prob_mode, evalmc_rollouts, resident_rollouts, rollout_resident_bytes, stream_format)


# ---------------------------------------------------------------------------------------------------
# The reference's single-environment class
# ---------------------------------------------------------------------------------------------------
class PSRS:
    """Rejection sampler that acts as an environment (psrs.py:5-57), device-backed.

    PSRS(buffer, nS=25, nA=5, reject_func=None): `buffer` is the reference's iterable of legacy tuples
    (s, a, r, s', done, p, info) with info['z'], info['z_next'|'next_z'], info['t'].  PSRS.from_arrays
    builds the same object from columns without the per-row Python loop.
    """

    def __init__(self, buffer, nS=25, nA=5, reject_func=None):
        rows = list(buffer)
        self.raw_buffer = rows
        n = len(rows)
        z = np.fromiter((r[6]["z"] for r in rows), np.int64, n)
        zn = np.fromiter((r[6]["z_next"] if "z_next" in r[6] else r[6]["next_z"] for r in rows), np.int64, n)
        t0 = np.fromiter((r[6]["t"] == 0 for r in rows), bool, n)
        a = np.fromiter((int(r[1]) for r in rows), np.int64, n)
        rew = np.array([r[2] for r in rows]) if n else np.zeros(0)
        done = np.fromiter((bool(r[4]) for r in rows), bool, n)
        p_log = np.stack([np.asarray(r[5]) for r in rows]) if n else np.zeros((0, nA))
        self._setup(z, a, rew, zn, done, p_log, t0, nS, nA, reject_func,
                    obs=[r[0] for r in rows], next_obs=[r[3] for r in rows], infos=[r[6] for r in rows], p_objs=[r[5] for r in rows])

    @classmethod
    def from_arrays(cls, z, a, r, z_next, done, p_log, t0=None, nS=None, nA=None, reject_func=None, obs=None, next_obs=None,
                    reject_mode=None):
        self = cls.__new__(cls)
        self.raw_buffer = None
        p_log = np.asarray(p_log)
        self._setup(np.asarray(z, np.int64), np.asarray(a, np.int64), np.asarray(r), np.asarray(z_next, np.int64),
                    np.asarray(done, bool), p_log, None if t0 is None else np.asarray(t0, bool),
                    nS if nS is not None else (int(max(np.max(z), np.max(z_next))) + 1 if len(z) else 1),
                    nA if nA is not None else p_log.shape[1], reject_func, obs=obs, next_obs=next_obs, reject_mode=reject_mode)
        return self

    def _setup(self, z, a, r, zn, done, p_log, t0, nS, nA, reject_func, obs=None, next_obs=None, infos=None, p_objs=None,
               reject_mode=None):
        self.nS, self.nA = nS, nA
        self._z, self._a, self._r, self._zn, self._done, self._p_log = z, a, r, zn, done, p_log
        self._t0 = np.ones(len(z), bool) if t0 is None else t0
        self._obs = z if obs is None else obs          # observation == latent state unless given
        self._next_obs = zn if next_obs is None else next_obs
        self._obs_is_state = obs is None or (len(z) > 0 and _all_equal(obs, z) and _all_equal(next_obs, zn))
        self._infos, self._p_objs = infos, p_objs
        self._reject_func = reject_func
        self.table = TransitionTable(z, a, r, zn, done, p_log, t0)
        self._env = BatchedPSRS(self.table, 1, L.REJECT_DEFAULT if reject_mode is None else reject_mode)
        self._fault_check = False
        self.s = None
        self.z = None
        self.reset_sampler()   # psrs.py:13 (unseeded on construction)
        self.reset()           # psrs.py:14

    # -- psrs.py:19-30 --
    def reset_sampler(self, seed=None):
        if seed is None:
            seed = int.from_bytes(os.urandom(8), "little")  # default_rng(None): fresh OS entropy
        self._sampler_seed = int(seed)
        self._env.reset_sampler([seed], SHUFFLE_PER_ROLLOUT)
        self._fault_check = True  # the shuffle bounds its waits: looked at where the host next synchronises (reset / step)

    def _check_faults(self):
        """After a host synchronisation: did the sampler reset give up a bounded wait (include/offsim.h: offsim_async_faults)?"""
        if getattr(self, "_fault_check", False):
            self._fault_check = False
            L.check_async_faults()

    # -- psrs.py:32-37 (the seed argument is ignored there too) --
    def reset(self, seed=None):
        row = self._env.reset_single()
        self._check_faults()
        if row < 0:
            self.s = None
            return None
        self.z = int(self._z[row])
        self.s = self._obs[row]
        return self.s

    # -- psrs.py:39-51 --
    def step(self, p_new):
        if isinstance(p_new, torch.Tensor):
            p_new = p_new.detach().cpu().numpy()
        elif not isinstance(p_new, np.ndarray):
            p_new = np.asarray(p_new)
        z = self.z
        if self._reject_func is None:
            row, status, popped = self._env.step_single(p_new)
        else:  # Python-side _reject hook: pop candidates one at a time and ask the callable (psrs.py:48)
            while True:
                row, status, _ = self._env.step_single(p_new, advance=False, reject_mode=L.REJECT_NEVER)
                if status != L.ST_OK:
                    break
                if not self._reject_func(p_new, self._p_of(row), self._a_of(row)):
                    self._env.set_state(torch.tensor([self.table.slot_of(self._zn[row])]))
                    break
        if self._fault_check:
            self._check_faults()
        if status != L.ST_OK:
            if status == L.ST_KEYERROR:
                raise KeyError(z)
            return None, None, None, None  # (ST_EXHAUSTED, ST_INACTIVE)
        self.s = self._next_obs[row]
        self.z = int(self._zn[row])
        return self.s, self._r[row], bool(self._done[row]), {"z": z, "a": self._a_of(row), "p": self._p_of(row)}

    def _p_of(self, row):
        return self._p_objs[row] if self._p_objs is not None else self._p_log[row]

    def _a_of(self, row):
        return self.raw_buffer[row][1] if self.raw_buffer is not None else self._a[row]

    def _default_reject(self, p_new, p_log, a) -> bool:
        """psrs.py:53-57.  Only reachable through user hooks that call it explicitly; the draw comes from the
        device stream so that the sequence stays the reference's."""
        u = self._draw_uniform()
        a = int(a)
        M = (p_new / p_log).max()
        return u > p_new[a] / p_log[a] / M

    # -- public attributes of the reference object, materialised on demand --
    @property
    def rejection_sampling_rng(self):
        self._env._quiesce()
        st = self._env.state.rng.cpu().numpy().view(np.uint64)[0]
        g = np.random.Generator(np.random.PCG64())
        g.bit_generator.state = {"bit_generator": "PCG64", "state": {"state": (int(st[0]) << 64) | int(st[1]),
                                 "inc": (int(st[2]) << 64) | int(st[3])}, "has_uint32": 0, "uinteger": 0}
        return g

    @rejection_sampling_rng.setter
    def rejection_sampling_rng(self, gen):
        s = gen.bit_generator.state
        if s["bit_generator"] != "PCG64":
            raise ValueError("only PCG64 generators (np.random.default_rng) can drive the device stream")
        self._env._quiesce()
        st, inc = s["state"]["state"], s["state"]["inc"]
        m = (1 << 64) - 1
        w = np.array([st >> 64, st & m, inc >> 64, inc & m], dtype=np.uint64).view(np.int64)
        self._env.state.rng.copy_(torch.from_numpy(w.copy()).reshape(1, 4))

    def _draw_uniform(self):
        g = self.rejection_sampling_rng
        u = g.random()
        self.rejection_sampling_rng = g
        return u

    def _orders(self):
        self._env._quiesce()
        t, st = self.table, self._env.state
        seg = t.seg_off.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        order = t.order.cpu().numpy()
        perm = st.perm.cpu().numpy().reshape(-1)[: t.N].astype(np.int64) if st.perm is not None else np.arange(t.N)
        cur = st.cursor.cpu().numpy()[0]
        return seg, order, perm, cur

    @property
    def queues(self):
        """{z: remaining rows of the queue, head first} as caller-buffer row indices (psrs.py:26-30)."""
        seg, order, perm, cur = self._orders()
        out = {}
        for s in range(self.table.n_slots):
            if seg[s + 1] > seg[s]:
                out[self.table.z_of(s)] = [int(order[g]) for g in perm[seg[s] + cur[s]: seg[s + 1]]]
        return out

    @property
    def init_queue(self):
        self._env._quiesce()
        st, t = self._env.state, self.table
        ip = st.init_perm.cpu().numpy().reshape(-1)[: t.N0] if st.init_perm is not None else np.arange(t.N0)
        rows = t.init_orig.cpu().numpy()[ip]
        ic = int(st.init_cursor.cpu()[0])
        return [(int(self._z[r]), self._obs[r]) for r in rows[ic:]]


def _all_equal(xs, arr):
    try:
        return bool(np.array_equal(np.asarray(xs), arr))
    except Exception:
        return False


# ---------------------------------------------------------------------------------------------------
def _require_device_env(env, what):
    """The drivers run on the device or not at all: a PSRS of this package whose observations are its latent states (Q / pi are
    indexed by the observation, psrs.py:158,255) and whose reject rule is built in."""
    if not isinstance(env, PSRS):
        raise TypeError(f"{what}: `env` must be the device-backed PSRS of this package (PSRS(...) / PSRS.from_arrays(...)), got {type(env).__name__}")
    if env._reject_func is not None:
        raise NotImplementedError(f"{what}: a Python reject hook decides every candidate on the host; drive env.step yourself, or use the built-in "
                                  "rule (reject_func=None) / the trivial baselines' reject modes")
    if not env._obs_is_state:
        raise NotImplementedError(f"{what}: the observations of this PSRS differ from its latent states, but the tabular drivers index their "
                                  "tables by the observation (psrs.py:158, :255)")


def evalMC_psrs(env, n_episodes, pi, gamma):
    """psrs.py:241-271: the whole loop -- env.reset(), env.step(pi[S]) until exhaustion, discounted returns -- in one kernel launch."""
    _require_device_env(env, "evalMC_psrs")
    pi = np.asarray(pi)
    if pi.ndim != 2:
        raise ValueError("evalMC_psrs: pi must be a [nS, nA] table")
    t = env.table
    if t.N and pi.shape[0] <= int(t.slot_z.max()):
        raise IndexError("pi has no row for some latent state")
    n_ep = int(min(n_episodes, t.N0 + 1))
    o = env._env.eval_mc(t.policy_slots(pi), gamma, n_ep, ep_cap=max(n_ep, 1))
    status = int(o["status"].cpu()[0])
    env._fault_check = False
    L.check_async_faults()  # (the copy above synchronised the stream)
    if status == L.ST_KEYERROR:
        raise KeyError(t.z_of(env._env.state.cur_slot.cpu()[0]))
    ne, nl = int(o["n_ep"].cpu()[0]), int(o["n_len"].cpu()[0])
    cs = int(env._env.state.cur_slot.cpu()[0])
    env.z = t.z_of(cs) if cs >= 0 else env.z
    if cs < 0:
        env.s = None
    return o["ep_g"].cpu().numpy()[0, :ne].copy(), o["ep_len"].cpu().numpy()[0, :nl].astype(np.int64)


# ---------------------------------------------------------------------------------------------------
# Learner-in-the-loop drivers (psrs.py:119-239)
# ---------------------------------------------------------------------------------------------------
def _q_to_slots(table, Q):
    """Rows of a [nS,nA] table in slot order (NumPy indexing for z = -1), zeros where Q has no row."""
    return np.nan_to_num(table.policy_slots(np.asarray(Q, dtype=np.float64)), nan=0.0)


def _slots_to_q(table, q_slots, Q):
    """q_slots [..., n_slots, nA] back into copies of the [nS,nA] table Q (rows without a slot keep Q's values)."""
    q_slots = np.asarray(q_slots)
    out = np.broadcast_to(np.asarray(Q, dtype=np.float64), q_slots.shape[:-2] + np.shape(Q)).copy()
    for s in range(table.n_slots):
        z = table.z_of(s)
        if -out.shape[-2] <= z < out.shape[-2]:
            out[..., z, :] = q_slots[..., s, :]
    return out


def _behaviour_kind(behavior_policy, nA):
    """Which of the reference's tabular behaviour policies (offsim4rl/agents/tabular.py:7-32) `behavior_policy(Q, {'epsilon': e})` is,
    found by probing it on a few Q rows: ('fixed', dist) for one that ignores Q (uniformly_random_policy), 'eps_greedy', 'greedy',
    'soft_greedy'.  Anything else cannot run on the device.  (The global NumPy stream, which the greedy policies draw from on ties,
    is put back as it was.)"""
    g = np.random.default_rng(1)
    state = np.random.get_state()
    try:
        call = lambda q, e: np.asarray(behavior_policy(np.asarray(q, dtype=np.float64)[None, :], dict(epsilon=e)), dtype=np.float64)[0]
        rows = [g.permutation(nA).astype(float) + g.random(nA) * 0.5 for _ in range(3)]
        outs = {e: [call(q, e) for q in rows] for e in (0.3, 0.7)}
        if all(np.array_equal(outs[e][0], o) for e in outs for o in outs[e][1:]):
            if np.array_equal(outs[0.3][0], outs[0.7][0]):
                return "fixed", outs[0.3][0]
            raise NotImplementedError("behaviour policy ignores Q but follows epsilon: not one of the reference's tabular policies")
        if nA < 2:
            raise NotImplementedError("behaviour policy on a single action")

        def pattern(q, lo, hi):
            w = np.full(nA, lo)
            w[int(np.argmax(q))] = hi
            return w
        if all(np.array_equal(o, pattern(q, e / nA, 1 - e + e / nA)) for e in outs for q, o in zip(rows, outs[e])):
            return "eps_greedy", None
        if all(np.array_equal(o, pattern(q, 0.0, 1.0)) for e in outs for q, o in zip(rows, outs[e])):
            near = rows[0].copy()
            near[int(np.argsort(near)[-2])] = near.max() - 1e-9  # two values np.isclose to each other, not equal
            w = call(near, 0.3)
            if np.count_nonzero(w) == 2 and np.allclose(w[w > 0], 0.5):
                return "soft_greedy", None
            if np.array_equal(w, pattern(near, 0.0, 1.0)):
                return "greedy", None
    except NotImplementedError:
        raise
    except Exception as ex:
        raise NotImplementedError(f"behaviour policy could not be probed ({ex!r})") from ex
    finally:
        np.random.set_state(state)
    raise NotImplementedError("behaviour policy is none of the reference's tabular policies (uniformly_random_policy, greedy_policy, soft_greedy_policy, "
                              "epsilon_greedy_policy: offsim4rl/agents/tabular.py:7-32); only those run on the device")


def _schedule(x, n_ep):
    """psrs.py:128-135: a callable alpha / epsilon is evaluated per episode; tabulated for the episodes the log can hold."""
    return np.array([float(x(e)) for e in range(max(n_ep, 1))], dtype=np.float64) if callable(x) else None


class _StepMemory:
    """The `memory` list of the drivers: (S, A, R, S', done, p, info) per accepted step, p = the behaviour distribution the learner
    revealed for that step.  Built on first use (for the Q-dependent policies p is a function of the step's Q row, which is
    replayed from the TD errors the device returned: Q[S,A] += alpha * td is the device's own update)."""

    def __init__(self, env, rows, p_of_step):
        self._env, self._rows, self._p_of_step, self._items = env, rows, p_of_step, None

    def _build(self):
        if self._items is None:
            env, ps = self._env, self._p_of_step()
            self._items = [(env._obs[r], env._a_of(r), env._r[r], env._next_obs[r], bool(env._done[r]), ps[i],
                            {"z": int(env._z[r]), "a": env._a_of(r), "p": env._p_of(r)}) for i, r in enumerate(self._rows)]
        return self._items

    def __len__(self):
        return len(self._rows)

    def __iter__(self):
        return iter(self._build())

    def __getitem__(self, i):
        return self._build()[i]


def _td_run(env, what, n_episodes, gamma, alpha, Q_init, save_Q, mode, kind, pi_rows, epsilon):
    """One launch of offsim_eval_td for a driver call; returns (Q, info) as the reference does."""
    _require_device_env(env, what)
    t = env.table
    nS, nA = env.nS, env.nA
    Q0 = np.zeros((nS, nA)) if Q_init is None else np.asarray(Q_init).copy().astype(float)
    n_ep = int(min(n_episodes, t.N0 + 1))
    a_tab, e_tab = _schedule(alpha, n_ep), _schedule(epsilon, n_ep)
    if kind == "greedy":
        e_tab, epsilon = None, 0.0
    behaviour = {"fixed": L.BEHAVIOUR_FIXED, "eps_greedy": L.BEHAVIOUR_EPS_GREEDY, "greedy": L.BEHAVIOUR_EPS_GREEDY,
                 "soft_greedy": L.BEHAVIOUR_SOFT_GREEDY}[kind]
    ties = behaviour == L.BEHAVIOUR_EPS_GREEDY
    if a_tab is not None and e_tab is None:
        e_tab = np.full(len(a_tab), float(epsilon) if not callable(epsilon) else 0.0)
    if e_tab is not None and a_tab is None:
        a_tab = np.full(len(e_tab), float(alpha))
    mt = None
    if ties:  # _random_argmax draws from NumPy's global stream (agents/tabular.py:4-5): the device continues it, and hands it back
        st = np.random.get_state()
        if st[0] != "MT19937":
            raise NotImplementedError("np.random's global bit generator is not MT19937")
        mt = np.concatenate([np.asarray(st[1], dtype=np.uint32), np.array([st[2]], dtype=np.uint32)])[None, :]
    cap = t.N + 1
    o = env._env.eval_td(t.policy_slots(pi_rows), gamma, mode, 0.0 if callable(alpha) else alpha, q_slots=_q_to_slots(t, Q0)[None],
                         n_episodes=n_ep, ep_cap=n_ep + 1, trace_cap=cap, behaviour=behaviour, epsilon=0.0 if callable(epsilon) else epsilon,
                         alpha_ep=a_tab, epsilon_ep=e_tab if behaviour == L.BEHAVIOUR_EPS_GREEDY else None,
                         snap_cap=cap if save_Q else 0, snap_stride=1, tie_mt=mt)
    status = int(o["status"].cpu()[0])
    env._fault_check = False
    L.check_async_faults()  # (the copy above synchronised the stream)
    if ties:
        w = o["tie_mt"].cpu().numpy().view(np.uint32)[0]
        np.random.set_state(("MT19937", w[:624].copy(), int(w[624]), st[3], st[4]))
    if status == L.ST_KEYERROR:
        raise KeyError(t.z_of(env._env.state.cur_slot.cpu()[0]))
    ne, steps = int(o["n_ep"].cpu()[0]), int(o["steps"].cpu()[0])
    n_g = ne + (1 if status == L.ST_EXHAUSTED else 0)  # the cut-short episode's return is appended too (psrs.py:177, :232)
    Q = _slots_to_q(t, o["q"].cpu().numpy()[0], Q0)
    rows = o["trace_row"].cpu().numpy()[0, :steps]
    td = o["td_err"].cpu().numpy()[0, :steps].copy()
    Qs = np.array([Q0])
    if save_Q:  # psrs.py:145,172-173: Q before the first step, then after every step
        Qs = np.concatenate([Q0[None], _slots_to_q(t, o["q_snap"].cpu().numpy()[0, :steps], Q0)], axis=0)
    done = env._done[rows] if steps else np.zeros(0, bool)
    episode = np.concatenate([[0], np.cumsum(done[:-1])]).astype(np.int64) if steps else np.zeros(0, np.int64)  # psrs.py:180: +1 per finished episode
    eps_of = (lambda i: e_tab[min(int(episode[i]), len(e_tab) - 1)]) if e_tab is not None else (lambda i: float(epsilon))
    alpha_of = (lambda i: a_tab[min(int(episode[i]), len(a_tab) - 1)]) if a_tab is not None else (lambda i: float(alpha))

    def p_of_step():
        if kind == "fixed":
            return [pi_rows[int(env._z[r])] for r in rows]
        if behaviour == L.BEHAVIOUR_EPS_GREEDY:
            best = o["beh_arg"].cpu().numpy()[0, :steps]
            ps = []
            for i in range(steps):
                e = eps_of(i)
                w = np.ones(nA) * e / nA
                w[int(best[i])] = 1 - e + e / nA
                ps.append(w)
            return ps
        Qr, ps = Q0.copy(), []  # soft greedy: replay Q from the TD errors
        for i, r in enumerate(rows):
            S, A = int(env._z[r]), int(env._a[r])
            w = np.zeros(nA)
            w[np.where(np.isclose(Qr[S], np.max(Qr[S])))[0]] = 1
            ps.append(w / w.sum())
            Qr[S, A] = Qr[S, A] + alpha_of(i) * td[i]
        return ps

    info = {"Gs": o["ep_g"].cpu().numpy()[0, :n_g].copy(), "Qs": Qs, "memory": _StepMemory(env, rows, p_of_step)}
    if mode == L.TD_QLEARN:
        info["TD_errors"] = td
    return Q, info


def qlearn_psrs(env, n_episodes, behavior_policy, gamma, alpha=0.1, epsilon=1.0, Q_init=None, save_Q=0):
    """psrs.py:119-185: Q-learning inside the simulator, the whole loop -- PSRS steps, the behaviour policy on the learner's own Q, the
    updates -- in one kernel launch.  `behavior_policy` must be one of the reference's tabular policies (agents/tabular.py:7-32; it is
    recognised by probing, see _behaviour_kind); alpha and epsilon may be callables of the episode (psrs.py:128-135); ties between
    maximal Q values are broken with NumPy's global stream exactly as the reference's _random_argmax does, and the stream is left
    where the reference would leave it; save_Q returns Q after every step (psrs.py:172-173)."""
    _require_device_env(env, "qlearn_psrs")
    kind, fixed = _behaviour_kind(behavior_policy, env.nA)
    t = env.table
    n_rows = max(env.nS, int(t.slot_z.max()) + 1 if t.N else 1)
    pi_rows = np.tile(np.asarray(fixed, dtype=np.float64), (n_rows, 1)) if kind == "fixed" else np.full((n_rows, env.nA), 1.0 / env.nA)
    return _td_run(env, "qlearn_psrs", n_episodes, gamma, alpha, Q_init, save_Q, L.TD_QLEARN, kind, pi_rows, epsilon)


def expSARSA_psrs(env, n_episodes, pi, gamma, alpha=0.1, Q_init=None, save_Q=0):
    """psrs.py:187-239: expected SARSA under a fixed target / behaviour policy pi, one kernel launch; alpha may be a callable of the
    episode, save_Q returns Q after every step."""
    _require_device_env(env, "expSARSA_psrs")
    pi = np.asarray(pi, dtype=np.float64)
    if pi.ndim != 2:
        raise ValueError("expSARSA_psrs: pi must be a [nS, nA] table")
    return _td_run(env, "expSARSA_psrs", n_episodes, gamma, alpha, Q_init, save_Q, L.TD_EXPSARSA, "fixed", pi, 0.0)
