"""Per-State Rejection Sampling on the MI355X: host-side mirror of offsim4rl/evaluators/psrs.py.

  BatchedPSRS   R independent PSRS environments over one device table, stepped by HIP kernels
  PSRS          the reference's single-environment class (psrs.py:5-57), as BatchedPSRS with R = 1
  evalMC_psrs   psrs.py:241-271; one kernel launch when given a device-backed env
  evalmc_rollouts  the batched driver behind the headline metric (thousands of seeds per launch)

Every decision (queue order, accept/reject, state walk, discounted return) is computed on the GPU
through the C ABI of include/offsim.h; the host only moves arguments and payload.  No CPU fallback.
"""
import ctypes as C
import os

import numpy as np
import torch

from .. import _lib as L
from ..table import RolloutState, TransitionTable, seed_streams, seeds_tensor, shuffle_queues

SHUFFLE_PER_ROLLOUT = "per_rollout"  # reset_sampler(seed_r) for every rollout r: the reference's meaning
SHUFFLE_SHARED = "shared"            # one queue order (shuffle_seed) shared by all rollouts, per-rollout rejection streams
SHUFFLE_NONE = "table_order"         # queues in buffer order (no shuffle); per-rollout rejection streams


_GP_CACHE = {}


def _gamma_pow(gamma, n, device, cap=None):
    """gamma**t exactly as the host computes it for the reference (Python float ** int == libm pow, psrs.py:262), for every
    t an episode can reach: at least `n` entries, then on until the factor is stationary (0, inf or 1: the device clamps t to
    the last entry in that case, csrc/discount.hpp) or `cap` entries (an episode has at most N steps) are there."""
    g = float(gamma)
    cap = max(int(n), 2) if cap is None else max(int(cap), int(n), 2)
    key = (g, int(n), cap, str(device))
    if key not in _GP_CACHE:
        vals = [g ** t for t in range(max(int(n), 2))]
        while len(vals) < cap and not (vals[-1] == vals[-2] and (vals[-1] in (0.0, 1.0) or vals[-1] in (float("inf"), float("-inf")))):
            t0 = len(vals)
            vals.extend(g ** t for t in range(t0, min(cap, t0 + 65536)))
        while len(vals) > max(int(n), 2) and vals[-1] == vals[-2] == vals[-3] and vals[-1] in (0.0, 1.0, float("inf"), float("-inf")):
            vals.pop()  # (extended in blocks: keep exactly two stationary entries)
        if len(_GP_CACHE) > 64:
            _GP_CACHE.clear()
        _GP_CACHE[key] = torch.tensor(vals, dtype=torch.float64, device=device)
    return _GP_CACHE[key]


def _prob_mode(table, p_dtype):
    f32 = (p_dtype in (np.float32, torch.float32, np.dtype(np.float32))) and table.p_log.dtype == torch.float32
    return L.PROB_F32 if f32 else L.PROB_F64


class BatchedPSRS:
    """R PSRS environments sharing one logged-transition table."""

    def __init__(self, table: TransitionTable, R: int, reject_mode=L.REJECT_DEFAULT):
        self.table, self.R = table, int(R)
        self.reject_mode = reject_mode
        self.state = RolloutState(table, R)
        dev = table.device
        self._row = torch.empty(R, dtype=torch.int32, device=dev)
        self._status = torch.empty(R, dtype=torch.int32, device=dev)
        self._popped = torch.empty(R, dtype=torch.int32, device=dev)
        self._perm_buf = None
        self._init_perm_buf = None
        self._dig_buf = self._loc_buf = None
        self._streams = None
        self._perm_lazy = None
        self._pk_cache = None
        self._dig32 = None

    # -- PSRS.reset_sampler (psrs.py:19-30) for all rollouts --
    def reset_sampler(self, seeds, shuffle=SHUFFLE_PER_ROLLOUT, shuffle_seed=None, policy=None):
        """`policy` (optional, [n_slots,nA] f64): the tabular policy the following eval_mc calls will evaluate.  It changes no
        result; it lets the sampler reset write the queue orders as the candidate streams the row-packed scan reads
        sequentially (offsim_shuffle_queues_keys: digest + 16-bit local row per queue position) instead of as permutations."""
        t, dev = self.table, self.table.device
        sd = seeds_tensor(seeds, dev)
        assert sd.numel() == self.R, "one seed per rollout"
        seed_streams(sd, self.state.rng)
        self.state.rewind()
        self._streams = None
        self._perm_lazy = None
        keyed = policy is not None and self._streams_apply(policy)
        if shuffle == SHUFFLE_PER_ROLLOUT:
            if self._init_perm_buf is None or self._init_perm_buf.shape[0] != self.R:
                self._init_perm_buf = torch.empty((self.R, max(t.N0, 1)), dtype=torch.int32, device=dev)
            if keyed:
                self._perm_buf = None  # (the two forms of the orders are not kept side by side: 4 + 6 bytes per entry and rollout)
                if self._dig_buf is None or self._dig_buf.shape[0] != self.R:
                    self._dig_buf = torch.empty((self.R, max(t.N, 1)), dtype=torch.int32, device=dev)
                    self._loc_buf = torch.empty((self.R, max(t.N, 1)), dtype=torch.int16, device=dev)
                keys, dig32 = self._policy_keys(policy)
                L.check(L.load().offsim_shuffle_queues_keys(C.byref(t.c), L.ptr(sd), self.R, L.ptr(dig32), L.ptr(self._dig_buf),
                                                            L.ptr(self._loc_buf), L.ptr(self._init_perm_buf), L.stream_ptr()))
                self._streams = dict(dig=self._dig_buf, dig_stride=t.N, loc=self._loc_buf, loc_stride=t.N, key=self._policy_key(policy))
                self.state.set_orders(None, 0, self._init_perm_buf, t.N0)
                self._perm_lazy = "streams"
            else:
                self._dig_buf = self._loc_buf = None
                if self._perm_buf is None or self._perm_buf.shape[0] != self.R:
                    self._perm_buf = torch.empty((self.R, max(t.N, 1)), dtype=torch.int32, device=dev)
                shuffle_queues(t, sd, self._perm_buf, self._init_perm_buf)
                self.state.set_orders(self._perm_buf, t.N, self._init_perm_buf, t.N0)
        elif shuffle == SHUFFLE_SHARED:
            assert shuffle_seed is not None
            perm, init_perm = shuffle_queues(t, seeds_tensor([shuffle_seed], dev))
            self._perm_buf, self._init_perm_buf = perm, init_perm
            self.state.set_orders(perm, 0, init_perm, 0)
            if keyed:  # one shared order: the streams are one row, built from the permutation
                keys, dig32 = self._policy_keys(policy)
                p = perm[0, :t.N].to(torch.int64) & 0xFFFFFFFF
                self._streams = dict(dig=dig32[p].contiguous(), dig_stride=0, loc=(p - self._seg_base()).to(torch.int16).contiguous(),
                                     loc_stride=0, key=self._policy_key(policy))
        elif shuffle == SHUFFLE_NONE:
            self.state.set_orders(None, 0, None, 0)
            if keyed:
                keys, dig32 = self._policy_keys(policy)
                self._streams = dict(dig=dig32, dig_stride=0, loc=None, loc_stride=0, key=self._policy_key(policy))
        else:
            raise ValueError(shuffle)

    # ---- candidate streams for the row-packed scan ----
    def _streams_apply(self, policy):
        """The streams exist for what offsim_eval_mc_streams covers: f64 probabilities, the default reject rule, <= 256 states,
        segments <= 65536 rows (16-bit local rows)."""
        t = self.table
        p = policy if isinstance(policy, torch.Tensor) else np.asarray(policy)
        f64 = p.dtype in (torch.float64, np.float64, np.dtype(np.float64))
        return (f64 and self.reject_mode == L.REJECT_DEFAULT and t.n_slots <= 256 and 0 < t.max_seg <= 65536 and t.N < 2 ** 32 - 1
                and os.environ.get("OFFSIM_SCAN_ROWS", "1") != "0")

    @staticmethod
    def _policy_key(policy):
        p = policy.detach().cpu().numpy() if isinstance(policy, torch.Tensor) else np.asarray(policy)
        return (p.shape, p.dtype.str, hash(np.ascontiguousarray(p).tobytes()))

    def _policy_keys(self, policy):
        """(compiled 64-bit keys, their 32-bit digests) of `policy` on the device, cached per policy."""
        k = self._policy_key(policy)
        if getattr(self, "_pk_cache", None) is None or self._pk_cache[0] != k:
            t = self.table
            pi_d = torch.as_tensor(np.ascontiguousarray(policy) if not isinstance(policy, torch.Tensor) else policy,
                                   dtype=torch.float64).to(t.device).reshape(t.n_slots, t.nA).contiguous()
            keys = self.compile_policy(pi_d)
            if getattr(self, "_dig32", None) is None:
                self._dig32 = torch.empty(max(t.N, 1), dtype=torch.int32, device=t.device)
            L.check(L.load().offsim_compile_digests(C.byref(t.c), L.ptr(keys), L.ptr(self._dig32), L.stream_ptr()))
            self._pk_cache = (k, keys, self._dig32)
        return self._pk_cache[1], self._pk_cache[2]

    def _derive_streams(self, policy, max_entries=1 << 26):
        """Candidate streams from queue orders that exist as permutations (reset_sampler without `policy`): one gather, done
        for jobs of up to `max_entries` queue positions; bigger jobs pass `policy` to reset_sampler or run the window kernels."""
        t, st = self.table, self.state
        if not self._streams_apply(policy) or (st.perm is None and self._perm_lazy == "streams"):
            return
        n_rows = 1 if (st.perm is None or st.perm_stride == 0) else self.R
        if n_rows * t.N > max_entries:
            return
        keys, dig32 = self._policy_keys(policy)
        key = self._policy_key(policy)
        if st.perm is None:  # table order
            self._streams = dict(dig=dig32, dig_stride=0, loc=None, loc_stride=0, key=key)
            return
        p = st.perm.reshape(n_rows, -1)[:, :t.N].to(torch.int64) & 0xFFFFFFFF
        self._streams = dict(dig=dig32[p].contiguous(), dig_stride=t.N if n_rows > 1 else 0,
                             loc=(p - self._seg_base()[None, :]).to(torch.int16).contiguous(), loc_stride=t.N if n_rows > 1 else 0, key=key)

    def _seg_base(self):
        """seg_off of the state every grouped position belongs to ([N] int64)."""
        t = self.table
        so = (t.seg_off.to(torch.int64) & 0xFFFFFFFF)
        return torch.repeat_interleave(so[:-1], so[1:] - so[:-1])

    @property
    def perm(self):
        """Queue orders as permutations of grouped rows [R or 1, N] (built from the streams when the reset wrote those)."""
        if self.state.perm is not None or self._perm_lazy != "streams":
            return self.state.perm
        return ((self._loc_buf.to(torch.int64) & 0xFFFF) + self._seg_base()[None, :]).to(torch.int32)

    def set_rejection_seeds(self, seeds):
        """Replace only the rejection streams (env.rejection_sampling_rng = default_rng(seed), psrs.py:20)."""
        seed_streams(seeds_tensor(seeds, self.table.device), self.state.rng)

    # -- PSRS.reset (psrs.py:32-37) --
    def reset(self, mask=None):
        m = None if mask is None else mask.to(torch.uint8).contiguous()
        L.check(L.load().offsim_env_reset(C.byref(self.table.c), C.byref(self.state.c), L.ptr(m), L.ptr(self._row), L.stream_ptr()))
        return self._row

    # -- PSRS.step (psrs.py:39-51) --
    def step(self, p_new, advance=True, reject_mode=None):
        """p_new: [R,nA] tensor/array.  Returns device tensors (row, status, popped)."""
        t = self.table
        if not isinstance(p_new, torch.Tensor):
            p_new = torch.from_numpy(np.ascontiguousarray(p_new))
        mode = _prob_mode(t, p_new.dtype)
        p = p_new.to(device=t.device, dtype=torch.float32 if mode == L.PROB_F32 else torch.float64).reshape(self.R, t.nA).contiguous()
        rm = self.reject_mode if reject_mode is None else reject_mode
        L.check(L.load().offsim_step_batch(C.byref(t.c), C.byref(self.state.c), L.ptr(p), mode, rm, 1 if advance else 0,
                                           L.ptr(self._row), L.ptr(self._status), L.ptr(self._popped), L.stream_ptr()))
        return self._row, self._status, self._popped

    def step_single(self, p_new, advance=True, reject_mode=None):
        """R = 1 convenience for the drop-in classes: one launch per call.  The kernel reads p_new from, and writes
        (row, status, popped) to, pinned host memory mapped into the device's address space, so that no copy is enqueued:
        the call is launch + stream synchronise.  Returns host ints (row, status, popped)."""
        t = self.table
        p_new = np.asarray(p_new)
        mode = _prob_mode(t, p_new.dtype)
        key = (mode, t.nA)
        if getattr(self, "_single_key", None) != key:
            dt = torch.float32 if mode == L.PROB_F32 else torch.float64
            self._p_host = torch.empty((1, t.nA), dtype=dt).pin_memory()
            self._o_host = torch.empty(3, dtype=torch.int32).pin_memory()
            self._p_np, self._o_np = self._p_host.numpy(), self._o_host.numpy()
            self._single_key = key
        self._p_np[0, :] = p_new.reshape(-1)
        rm = self.reject_mode if reject_mode is None else reject_mode
        base = self._o_host.data_ptr()
        L.check(L.load().offsim_step_batch(C.byref(t.c), C.byref(self.state.c), self._p_host.data_ptr(), mode, rm, 1 if advance else 0,
                                           base, base + 4, base + 8, L.stream_ptr()))
        torch.cuda.current_stream().synchronize()
        row, status, popped = int(self._o_np[0]), int(self._o_np[1]), int(self._o_np[2])
        self.last_row = row
        return row, status, popped

    def set_state(self, slots, mask=None):
        s = slots.to(device=self.table.device, dtype=torch.int32).contiguous()
        m = None if mask is None else mask.to(torch.uint8).contiguous()
        L.check(L.load().offsim_env_set_state(C.byref(self.state.c), L.ptr(s), L.ptr(m), L.stream_ptr()))

    # -- evalMC_psrs (psrs.py:241-271) for all rollouts in one launch --
    def eval_mc(self, pi_slots, gamma, n_episodes=None, ep_cap=0, trace_cap=0, n_gamma_pow=4096, out=None, fast=None, dbg=False):
        """pi_slots: [n_slots,nA] policy per state slot (TransitionTable.policy_slots).  Returns a dict of device
        tensors: sum_g, n_ep, steps, cand, n_len, status (+ ep_g, ep_len, trace_row, trace_pop when asked).
        fast=None picks the compiled-policy / LDS-window kernel (offsim_eval_mc_keys) whenever it applies
        (f64 probabilities, default reject rule, <= 256 states); fast=False forces the generic kernel."""
        t, dev, R = self.table, self.table.device, self.R
        if not isinstance(pi_slots, torch.Tensor):
            pi_slots = torch.from_numpy(np.ascontiguousarray(pi_slots))
        mode = _prob_mode(t, pi_slots.dtype)
        pi_d = pi_slots.to(device=dev, dtype=torch.float32 if mode == L.PROB_F32 else torch.float64).reshape(t.n_slots, t.nA).contiguous()
        if n_episodes is None:
            n_episodes = 1 << 62
        o = out or {}
        for k, dt in (("sum_g", torch.float64), ("n_ep", torch.int64), ("steps", torch.int64), ("cand", torch.int64),
                      ("n_len", torch.int64), ("status", torch.int32)):
            if k not in o:
                o[k] = torch.empty(R, dtype=dt, device=dev)
        if ep_cap:
            o["ep_g"] = torch.zeros((R, ep_cap), dtype=torch.float64, device=dev)
            o["ep_len"] = torch.zeros((R, ep_cap + 1), dtype=torch.int32, device=dev)
        if trace_cap:
            o["trace_row"] = torch.full((R, trace_cap), -1, dtype=torch.int32, device=dev)
            o["trace_pop"] = torch.zeros((R, trace_cap), dtype=torch.int32, device=dev)
        if dbg:
            o["dbg"] = torch.zeros((R, 4), dtype=torch.int64, device=dev)
        gp = _gamma_pow(gamma, n_gamma_pow, dev, cap=t.N + 2)
        oc = L.EvalMCOut(dbg=L.ptr(o.get("dbg")), sum_g=L.ptr(o["sum_g"]), n_ep=L.ptr(o["n_ep"]), steps=L.ptr(o["steps"]), cand=L.ptr(o["cand"]),
                         n_len=L.ptr(o["n_len"]), status=L.ptr(o["status"]), ep_g=L.ptr(o.get("ep_g")),
                         ep_len=L.ptr(o.get("ep_len")), ep_cap=ep_cap, trace_row=L.ptr(o.get("trace_row")),
                         trace_pop=L.ptr(o.get("trace_pop")), trace_cap=trace_cap)
        can_fast = mode == L.PROB_F64 and self.reject_mode == L.REJECT_DEFAULT and t.n_slots <= 256
        if fast is None:
            fast = can_fast
        if fast and not can_fast:
            raise L.OffsimError("the compiled-policy scan needs f64 probabilities, the default reject rule and <= 256 states")
        if fast and not (self._streams is not None and self._streams["key"] == self._policy_key(pi_slots)):
            self._derive_streams(pi_slots)  # small jobs: the streams are gathered from the permutations on the spot
        rows = bool(fast) and self._streams is not None and self._streams["key"] == self._policy_key(pi_slots)
        if rows:  # the sampler reset laid the orders out as candidate streams for this policy: row-packed scan
            keys, _ = self._policy_keys(pi_slots)
            sm = self._streams
            smc = L.Streams(dig=L.ptr(sm["dig"]), dig_stride=sm["dig_stride"], loc=L.ptr(sm["loc"]), loc_stride=sm["loc_stride"])
            L.check(L.load().offsim_eval_mc_streams(C.byref(t.c), C.byref(self.state.c), C.byref(smc), L.ptr(keys), float(gamma), L.ptr(gp),
                                                    gp.numel(), int(n_episodes), C.byref(oc), L.stream_ptr()))
            o["_keepalive"] = (pi_d, gp, keys, sm)
        elif fast and self.state.perm is None and self._perm_lazy == "streams":
            raise L.OffsimError("the queue orders were laid out for another policy (reset_sampler(policy=...)): call reset_sampler again")
        elif fast:
            keys = self.compile_policy(pi_d)
            L.check(L.load().offsim_eval_mc_keys(C.byref(t.c), C.byref(self.state.c), L.ptr(keys), float(gamma), L.ptr(gp),
                                                 gp.numel(), int(n_episodes), C.byref(oc), L.stream_ptr()))
            o["_keepalive"] = (pi_d, gp, keys)
        else:
            if self.state.perm is None and self._perm_lazy == "streams":
                self.state.set_orders(self.perm, t.N, self.state.init_perm, self.state.init_stride)  # (materialise the permutations)
            L.check(L.load().offsim_eval_mc(C.byref(t.c), C.byref(self.state.c), L.ptr(pi_d), mode, self.reject_mode, float(gamma),
                                            L.ptr(gp), gp.numel(), int(n_episodes), C.byref(oc), L.stream_ptr()))
            o["_keepalive"] = (pi_d, gp)
        return o

    # -- qlearn_psrs / expSARSA_psrs (psrs.py:119-239) with a Q-independent behaviour policy, all rollouts in one launch --
    def eval_td(self, pi_slots, gamma, mode, alpha, q_slots=None, n_episodes=None, ep_cap=0, trace_cap=0, n_gamma_pow=4096,
                behaviour=L.BEHAVIOUR_FIXED, epsilon=0.0):
        """mode: _lib.TD_QLEARN | _lib.TD_EXPSARSA.  q_slots [R,n_slots,nA] f64 (Q_init; zeros if None) is updated in place
        and returned as out["q"]; out["td_err"] [R,trace_cap] holds the TD errors in step order.
        behaviour = _lib.BEHAVIOUR_EPS_GREEDY: every rollout acts epsilon-greedily on its own Q table (the learner-in-the-loop
        case of psrs.py:158); pi_slots is then only the target policy of expected SARSA."""
        t, dev, R = self.table, self.table.device, self.R
        pi_d = torch.as_tensor(np.ascontiguousarray(pi_slots), dtype=torch.float64).to(dev).reshape(t.n_slots, t.nA).contiguous()
        if n_episodes is None:
            n_episodes = 1 << 62
        q = torch.zeros((R, t.n_slots, t.nA), dtype=torch.float64, device=dev) if q_slots is None else \
            torch.as_tensor(q_slots, dtype=torch.float64).to(dev).reshape(R, t.n_slots, t.nA).contiguous().clone()
        o = {k: torch.empty(R, dtype=dt, device=dev) for k, dt in (("sum_g", torch.float64), ("n_ep", torch.int64), ("steps", torch.int64),
                                                                   ("cand", torch.int64), ("n_len", torch.int64), ("status", torch.int32))}
        if ep_cap:
            o["ep_g"] = torch.zeros((R, ep_cap), dtype=torch.float64, device=dev)
            o["ep_len"] = torch.zeros((R, ep_cap + 1), dtype=torch.int32, device=dev)
        if trace_cap:
            o["trace_row"] = torch.full((R, trace_cap), -1, dtype=torch.int32, device=dev)
            o["trace_pop"] = torch.zeros((R, trace_cap), dtype=torch.int32, device=dev)
            o["td_err"] = torch.zeros((R, trace_cap), dtype=torch.float64, device=dev)
        gp = _gamma_pow(gamma, n_gamma_pow, dev, cap=t.N + 2)
        oc = L.EvalMCOut(sum_g=L.ptr(o["sum_g"]), n_ep=L.ptr(o["n_ep"]), steps=L.ptr(o["steps"]), cand=L.ptr(o["cand"]),
                         n_len=L.ptr(o["n_len"]), status=L.ptr(o["status"]), ep_g=L.ptr(o.get("ep_g")), ep_len=L.ptr(o.get("ep_len")),
                         ep_cap=ep_cap, trace_row=L.ptr(o.get("trace_row")), trace_pop=L.ptr(o.get("trace_pop")), trace_cap=trace_cap)
        tdc = L.TD(mode=mode, alpha=float(alpha), q=L.ptr(q), td_err=L.ptr(o.get("td_err")), td_cap=trace_cap, behaviour=int(behaviour),
                   epsilon=float(epsilon))
        L.check(L.load().offsim_eval_td(C.byref(t.c), C.byref(self.state.c), L.ptr(pi_d), self.reject_mode, float(gamma), L.ptr(gp),
                                        gp.numel(), int(n_episodes), C.byref(oc), C.byref(tdc), L.stream_ptr()))
        o["q"] = q
        o["_keepalive"] = (pi_d, gp)
        return o

    def scan_variant(self):
        """Name of the kernel eval_mc's fast path launches for this table and batch size (measurement label)."""
        if self._streams is not None:
            return "k_eval_mc_rows"
        return L.load().offsim_eval_mc_keys_kernel(self.table.n_slots, self.R).decode() or "k_eval_mc"

    def compile_policy(self, pi_d):
        """offsim_compile_policy: one 64-bit key per grouped row for the tabular policy pi_d [n_slots,nA] f64 (device)."""
        t = self.table
        if getattr(self, "_keys", None) is None:
            self._keys = torch.empty(max(t.N, 1), dtype=torch.int64, device=t.device)
        self._pk_cache = None  # (the key buffer is shared with _policy_keys)
        L.check(L.load().offsim_compile_policy(C.byref(t.c), L.ptr(pi_d), L.ptr(self._keys), L.stream_ptr()))
        return self._keys


def evalmc_rollouts(table, seeds, pi, gamma, shuffle=SHUFFLE_PER_ROLLOUT, shuffle_seed=None, tile=None,
                    reject_mode=L.REJECT_DEFAULT, n_episodes=None):
    """evalMC_psrs for many sampler seeds.  Rollouts are processed in tiles of `tile` seeds so that the per-rollout
    queue permutations (4*N bytes each) fit the HBM budget.  Returns host arrays: sum_g, n_ep, steps, cand, status
    and value = sum_g / n_ep (the per-seed value estimate, Gs.mean())."""
    seeds = np.asarray(seeds, dtype=np.uint64)
    R = len(seeds)
    if tile is None:
        budget = 48 << 30  # bytes of permutation indices kept resident at once
        tile = R if shuffle != SHUFFLE_PER_ROLLOUT else int(max(1, min(R, budget // max(4 * table.N, 1))))
    pi_slots = table.policy_slots(pi)
    outs = {k: [] for k in ("sum_g", "n_ep", "steps", "cand", "status")}
    env = None
    for b in range(0, R, tile):
        sd = seeds[b:b + tile]
        if env is None or env.R != len(sd):
            env = BatchedPSRS(table, len(sd), reject_mode)
        env.reset_sampler(sd, shuffle, shuffle_seed, policy=pi_slots)
        o = env.eval_mc(pi_slots, gamma, n_episodes)
        for k in outs:
            outs[k].append(o[k].cpu().numpy())
    res = {k: np.concatenate(v) for k, v in outs.items()}
    with np.errstate(invalid="ignore", divide="ignore"):
        res["value"] = res["sum_g"] / res["n_ep"]
    return res


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

    # -- psrs.py:32-37 (the seed argument is ignored there too) --
    def reset(self, seed=None):
        row = int(self._env.reset().cpu()[0])
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
        if status == L.ST_KEYERROR:
            raise KeyError(z)
        if status in (L.ST_EXHAUSTED, L.ST_INACTIVE):
            return None, None, None, None
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
def evalMC_psrs(env, n_episodes, pi, gamma):
    """psrs.py:241-271.  For a device-backed PSRS whose observations are its latent states and whose reject rule is
    built in, the whole loop runs in one kernel launch; otherwise the reference's host loop drives env.step."""
    if isinstance(env, PSRS) and env._reject_func is None and env._obs_is_state and isinstance(pi, np.ndarray) and pi.ndim == 2:
        t = env.table
        if t.N and pi.shape[0] <= int(t.slot_z.max()):
            raise IndexError("pi has no row for some latent state")
        n_ep = int(min(n_episodes, t.N0 + 1))
        o = env._env.eval_mc(t.policy_slots(pi), gamma, n_ep, ep_cap=max(n_ep, 1))
        status = int(o["status"].cpu()[0])
        if status == L.ST_KEYERROR:
            raise KeyError(t.z_of(env._env.state.cur_slot.cpu()[0]))
        ne, nl = int(o["n_ep"].cpu()[0]), int(o["n_len"].cpu()[0])
        cs = int(env._env.state.cur_slot.cpu()[0])
        env.z = t.z_of(cs) if cs >= 0 else env.z
        if cs < 0:
            env.s = None
        return o["ep_g"].cpu().numpy()[0, :ne].copy(), o["ep_len"].cpu().numpy()[0, :nl].astype(np.int64)
    Gs, lengths = [], []
    episode, terminate = 0, False
    while episode < n_episodes and not terminate:
        G, t = 0, 0
        S = env.reset(seed=episode)
        if S is None:
            break
        done = False
        while not done:
            S_, R, done, info = env.step(pi[S])
            if S_ is None:
                terminate = True
                break
            S = S_
            G = G + (gamma ** t) * R
            t = t + 1
        lengths.append(t)
        if done:
            Gs.append(G)
            episode += 1
    return np.array(Gs), np.array(lengths)


# ---------------------------------------------------------------------------------------------------
# Learner-in-the-loop drivers (psrs.py:119-239)
# ---------------------------------------------------------------------------------------------------
def _q_to_slots(table, Q):
    """Rows of a [nS,nA] table in slot order (NumPy indexing for z = -1), zeros where Q has no row."""
    out = np.nan_to_num(table.policy_slots(np.asarray(Q, dtype=np.float64)), nan=0.0)
    return out


def _slots_to_q(table, q_slots, Q):
    Q = np.array(Q, dtype=np.float64, copy=True)
    for s in range(table.n_slots):
        z = table.z_of(s)
        if -Q.shape[0] <= z < Q.shape[0]:
            Q[z] = q_slots[s]
    return Q


def _fixed_behaviour(behavior_policy, nA, epsilon):
    """The behaviour distribution if it does not depend on Q (e.g. epsilon = 1 or a uniform policy), else None."""
    g = np.random.default_rng(0)
    outs = [np.asarray(behavior_policy(q, dict(epsilon=epsilon)))[0] for q in (np.zeros((1, nA)), g.standard_normal((1, nA)), -g.random((1, nA)))]
    return outs[0] if all(np.array_equal(outs[0], o) for o in outs[1:]) else None


def _eps_greedy_behaviour(behavior_policy, nA, epsilon):
    """True if behavior_policy(Q, {'epsilon': e}) is epsilon_greedy_policy of offsim4rl/agents/tabular.py:24-32: e / nA everywhere
    and 1 - e + e / nA at the maximum of each row (probed on rows without ties)."""
    if nA < 2:
        return False
    g = np.random.default_rng(1)
    lo, hi = np.ones(1)[0] * epsilon / nA, 1 - epsilon + epsilon / nA
    state = np.random.get_state()  # the reference's _random_argmax draws from the global stream even without a tie
    try:
        for _ in range(4):
            q = g.permutation(nA).astype(float)[None, :] + g.random((1, nA)) * 0.5
            want = np.full((1, nA), lo)
            want[0, int(np.argmax(q[0]))] = hi
            if not np.array_equal(np.asarray(behavior_policy(q, dict(epsilon=epsilon))), want):
                return False
    except Exception:
        return False
    finally:
        np.random.set_state(state)
    return True


class _ReplayedMemory:
    """The `memory` list of qlearn_psrs for the device path with an epsilon-greedy learner: the behaviour distribution of
    every step depends on Q at that step, so the tuples are rebuilt by replaying the TD updates over the accepted rows on first use."""

    def __init__(self, env, rows, Q0, gamma, alpha, epsilon):
        self._args, self._items = (env, rows, Q0, gamma, alpha, epsilon), None

    def _build(self):
        if self._items is None:
            env, rows, Q0, gamma, alpha, epsilon = self._args
            Q = Q0.copy()
            nA = Q.shape[1]
            lo, hi = epsilon / nA, 1 - epsilon + epsilon / nA
            items = []
            for r in rows:
                S, A, R, S_ = int(env._z[r]), env._a_of(r), env._r[r], int(env._zn[r])
                p = np.full(nA, lo)
                p[int(np.argmax(Q[S]))] = hi
                items.append((env._obs[r], A, R, env._next_obs[r], bool(env._done[r]), p, {"z": S, "a": A, "p": env._p_of(r)}))
                Q[S, A] = Q[S, A] + alpha * (R + gamma * Q[S_].max() - Q[S, A])
            self._items = items
        return self._items

    def __len__(self):
        return len(self._args[1])

    def __iter__(self):
        return iter(self._build())

    def __getitem__(self, i):
        return self._build()[i]


def _td_device(env, n_episodes, p_rows, gamma, alpha, mode, Q_init, behaviour=L.BEHAVIOUR_FIXED, epsilon=0.0):
    t = env.table
    nS, nA = env.nS, env.nA
    Q0 = np.zeros((nS, nA)) if Q_init is None else np.asarray(Q_init).copy().astype(float)
    n_ep = int(min(n_episodes, t.N0 + 1))
    o = env._env.eval_td(t.policy_slots(p_rows), gamma, mode, alpha, q_slots=_q_to_slots(t, Q0)[None], n_episodes=n_ep,
                         ep_cap=n_ep + 1, trace_cap=t.N + 1, behaviour=behaviour, epsilon=epsilon)
    status = int(o["status"].cpu()[0])
    if status == L.ST_KEYERROR:
        raise KeyError(t.z_of(env._env.state.cur_slot.cpu()[0]))
    ne, steps = int(o["n_ep"].cpu()[0]), int(o["steps"].cpu()[0])
    n_g = ne + (1 if status == L.ST_EXHAUSTED else 0)  # the cut-short episode's return is appended too (psrs.py:177, :232)
    Q = _slots_to_q(t, o["q"].cpu().numpy()[0], Q0)
    rows = o["trace_row"].cpu().numpy()[0, :steps]
    return Q, o["ep_g"].cpu().numpy()[0, :n_g].copy(), o["td_err"].cpu().numpy()[0, :steps].copy(), rows, Q0


def _memory(env, rows, p_of_state):
    return [(env._obs[r], env._a_of(r), env._r[r], env._next_obs[r], bool(env._done[r]), p_of_state(int(env._z[r])),
             {"z": int(env._z[r]), "a": env._a_of(r), "p": env._p_of(r)}) for r in rows]


def qlearn_psrs(env, n_episodes, behavior_policy, gamma, alpha=0.1, epsilon=1.0, Q_init=None, save_Q=0):
    """psrs.py:119-185.  On a device-backed PSRS the whole loop -- PSRS steps and Q-learning updates -- is one kernel launch
    when the behaviour policy is Q-independent (epsilon = 1, uniform, ...) or the reference's epsilon_greedy_policy with a
    constant epsilon; otherwise the reference's host loop drives env.step."""
    fixed = None
    on_device = isinstance(env, PSRS) and env._reject_func is None and env._obs_is_state and not callable(alpha) and not callable(epsilon) and not save_Q
    if on_device:
        fixed = _fixed_behaviour(behavior_policy, env.nA, epsilon)
    if on_device and fixed is None and _eps_greedy_behaviour(behavior_policy, env.nA, epsilon):
        # the learner acts epsilon-greedily on the Q table it is learning: p_new is rebuilt from the rollout's Q row (in LDS) before
        # every step.  Ties between maximal Q values go to the first action (the reference draws among them from np.random).
        t = env.table
        p_rows = np.full((max(env.nS, int(t.slot_z.max()) + 1 if t.N else 1), env.nA), 1.0 / env.nA)
        Q, Gs, td, rows, Q0 = _td_device(env, n_episodes, p_rows, gamma, alpha, L.TD_QLEARN, Q_init, L.BEHAVIOUR_EPS_GREEDY, epsilon)
        return Q, {"Gs": Gs, "Qs": np.array([Q0]), "TD_errors": td, "memory": _ReplayedMemory(env, rows, Q0, gamma, alpha, epsilon)}
    if fixed is not None:
        t = env.table
        p_rows = np.tile(np.asarray(fixed, dtype=np.float64), (max(env.nS, int(t.slot_z.max()) + 1 if t.N else 1), 1))
        Q, Gs, td, rows, Q0 = _td_device(env, n_episodes, p_rows, gamma, alpha, L.TD_QLEARN, Q_init)
        return Q, {"Gs": Gs, "Qs": np.array([Q0]), "TD_errors": td, "memory": _memory(env, rows, lambda z: fixed)}
    epsilon_func = epsilon if callable(epsilon) else (lambda episode: epsilon)
    alpha_func = alpha if callable(alpha) else (lambda episode: alpha)
    Q = np.zeros((env.nS, env.nA)) if Q_init is None else Q_init.copy().astype(float)
    Gs, Qs, TD_errors, memory_buffer = [], [Q.copy()], [], []
    episode, terminate = 0, False
    while episode < n_episodes and not terminate:
        G, t = 0, 0
        S = env.reset(seed=episode)
        if S is None:
            break
        done = False
        while not done:
            p = behavior_policy(Q[[S], :], dict(epsilon=epsilon_func(episode)))[0]
            S_, R, done, info = env.step(p)
            if S_ is None:
                terminate = True
                break
            A = info["a"]
            memory_buffer.append((S, A, R, S_, done, p, info))
            TD_errors.append(R + gamma * Q[S_].max() - Q[S, A])
            Q[S, A] = Q[S, A] + alpha_func(episode) * (R + gamma * Q[S_].max() - Q[S, A])
            S = S_
            G = G + (gamma ** t) * R
            t = t + 1
            if save_Q:
                Qs.append(Q.copy())
        Gs.append(G)
        episode += 1
    return Q, {"Gs": np.array(Gs), "Qs": np.array(Qs), "TD_errors": np.array(TD_errors), "memory": memory_buffer}


def expSARSA_psrs(env, n_episodes, pi, gamma, alpha=0.1, Q_init=None, save_Q=0):
    """psrs.py:187-239: expected SARSA under a fixed target/behaviour policy pi; one launch on a device-backed PSRS."""
    if isinstance(env, PSRS) and env._reject_func is None and env._obs_is_state and not callable(alpha) and not save_Q and \
            isinstance(pi, np.ndarray) and pi.ndim == 2:
        Q, Gs, td, rows, Q0 = _td_device(env, n_episodes, pi, gamma, alpha, L.TD_EXPSARSA, Q_init)
        return Q, {"Gs": Gs, "Qs": np.array([Q0]), "memory": _memory(env, rows, lambda z: pi[z])}
    alpha_func = alpha if callable(alpha) else (lambda episode: alpha)
    Q = np.zeros((env.nS, env.nA)) if Q_init is None else Q_init.copy().astype(float)
    Gs, Qs, memory_buffer = [], [Q.copy()], []
    episode, terminate = 0, False
    while episode < n_episodes and not terminate:
        G, t = 0, 0
        S = env.reset(seed=episode)
        if S is None:
            break
        done = False
        while not done:
            p = pi[S]
            S_, R, done, info = env.step(p)
            if S_ is None:
                terminate = True
                break
            A = info["a"]
            memory_buffer.append((S, A, R, S_, done, p, info))
            Q[S, A] = Q[S, A] + alpha_func(episode) * (R + gamma * (Q[S_] @ pi[S_]) - Q[S, A])
            S = S_
            G = G + (gamma ** t) * R
            t = t + 1
            if save_Q:
                Qs.append(Q.copy())
        Gs.append(G)
        episode += 1
    return Q, {"Gs": np.array(Gs), "Qs": np.array(Qs), "memory": memory_buffer}
