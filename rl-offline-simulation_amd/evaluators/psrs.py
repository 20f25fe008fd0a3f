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
import time

import numpy as np
import torch

from .. import _lib as L
from ..table import RolloutState, TransitionTable, seed_streams, seeds_tensor, shuffle_queues

SHUFFLE_PER_ROLLOUT = "per_rollout"  # reset_sampler(seed_r) for every rollout r: the reference's meaning
SHUFFLE_SHARED = "shared"            # one queue order (shuffle_seed) shared by all rollouts, per-rollout rejection streams
SHUFFLE_NONE = "table_order"         # queues in buffer order (no shuffle); per-rollout rejection streams


_GP_CACHE = {}
_GP_CACHE_BYTES = 256 << 20


def _gamma_pow(gamma, n, device, cap=None):
    """gamma**t exactly as the host computes it for the reference (Python float ** int == libm pow, psrs.py:262), for every
    t an episode can reach: at least `n` entries, then on until the factor is stationary (0, inf or 1: the device clamps t to
    the last entry in that case, csrc/discount.hpp) or `cap` entries (an episode has at most N steps) are there.  Built in
    blocks of 65536 with Python's own `float ** int` (NumPy's array pow is vectorised differently and differs in the last bit for
    some t), so a gamma below 1 stops after the block its factor underflows in; the cache is bounded by bytes (_GP_CACHE_BYTES)."""
    g = float(gamma)
    cap = max(int(n), 2) if cap is None else max(int(cap), int(n), 2)
    key = (g, int(n), cap, str(device))
    if key not in _GP_CACHE:
        stationary = lambda v: len(v) >= 2 and v[-1] == v[-2] and (v[-1] in (0.0, 1.0) or np.isinf(v[-1]))
        blocks, total = [], 0
        while total < max(int(n), 2) or (total < cap and not stationary(blocks[-1])):
            m = min(65536, (max(int(n), 2) if total < max(int(n), 2) else cap) - total)
            blocks.append(np.array([g ** t for t in range(total, total + m)], dtype=np.float64))  # (Python's own pow: NumPy's array pow is not bit-identical)
            total += m
        vals = np.concatenate(blocks)
        keep = len(vals)
        while keep > max(int(n), 2) and vals[keep - 1] == vals[keep - 2] == vals[keep - 3] and (vals[keep - 1] in (0.0, 1.0) or np.isinf(vals[keep - 1])):
            keep -= 1  # (extended in blocks: keep exactly two stationary entries)
        if sum(v.numel() * 8 for v in _GP_CACHE.values()) + keep * 8 > _GP_CACHE_BYTES:
            _GP_CACHE.clear()
        _GP_CACHE[key] = torch.from_numpy(vals[:keep].copy()).to(device)
    return _GP_CACHE[key]


def stream_format(table):
    """Layout of the candidate streams of `table` (include/offsim.h): A while every state has at most 65536 rows (21-bit thresholds, 16-bit
    local rows); C for states of up to 2^17 rows (14-bit thresholds, the local row's bits 8.. inside the digest, ONE byte beside it: 5
    bytes per queue position instead of 6 -- what lets a 12.5 M-row shard keep 4096 rollouts resident at once; written by the chunked
    shuffle only, so not with OFFSIM_SHUFFLE_CHUNKED=0); B beyond (16-bit thresholds, bits 16.. inside the digest).
    OFFSIM_STREAMS_FORMAT=B keeps B where C would apply (A/B runs)."""
    if table.max_seg <= 65536:
        return L.STREAMS_A
    if (table.max_seg <= (1 << 17) and table.n_slots <= 255 and os.environ.get("OFFSIM_SHUFFLE_CHUNKED", "1") != "0"
            and os.environ.get("OFFSIM_STREAMS_FORMAT", "") != "B"):
        return L.STREAMS_C
    return L.STREAMS_B


ROWS_TICK_STEPS = 16  # steps between two top-up rounds of the row-packed scan (csrc/scan_rows.hpp: ROWS_TICK)
ROWS_MAX_WINDOW_LOAD = 1.2  # candidates a tick takes out of the busiest 8-entry window, above which the window kernel is the faster scan
ROWS_MAX_ALL_REJECTED = 0.03  # probability that a full 8-entry window holds no accept, from which the 32-entry window kernel (<= 64 states) is the faster scan


def _prob_mode(table, p_dtype):
    f32 = (p_dtype in (np.float32, torch.float32, np.dtype(np.float32))) and table.p_log.dtype == torch.float32
    return L.PROB_F32 if f32 else L.PROB_F64


class BatchedPSRS:
    """R PSRS environments sharing one logged-transition table.

    Every method only enqueues kernels on the current stream.  The sampler reset and the row-packed scan bound their
    inter-wavefront waits and raise a device-wide fault word instead of hanging (include/offsim.h: offsim_async_faults): whoever
    drives this class directly calls `check_faults()` once the results have been copied back (the host-facing drivers of this
    module -- PSRS, evalMC_psrs, qlearn_psrs, expSARSA_psrs, evalmc_rollouts, VectorPSRS(strict=True) -- do)."""

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

    @staticmethod
    def check_faults():
        """Synchronise the current stream and raise OffsimError if a kernel gave up a bounded wait since the last check."""
        torch.cuda.current_stream().synchronize()
        L.check_async_faults()

    # -- PSRS.reset_sampler (psrs.py:19-30) for all rollouts --
    def reset_sampler(self, seeds, shuffle=SHUFFLE_PER_ROLLOUT, shuffle_seed=None, policy=None):
        """`policy` (optional, [n_slots,nA] f64): the tabular policy the following eval_mc calls will evaluate.  It changes no
        result; it lets the sampler reset write the queue orders as the candidate streams the row-packed scan reads
        sequentially (offsim_shuffle_queues_keys: digest + 16-bit local row per queue position) instead of as permutations.
        (step / step_single / eval_td / the generic eval_mc need permutations and rebuild them from the streams on first use.)"""
        self._quiesce()
        t, dev = self.table, self.table.device
        sd = seeds_tensor(seeds, dev)
        assert sd.numel() == self.R, "one seed per rollout"
        seed_streams(sd, self.state.rng)
        self.state.rng_kind = L.STREAM_PCG64
        self.state._refresh()
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
                    self._loc_buf = torch.empty((self.R, max(t.N, 1)), dtype=torch.uint8 if self._loc_bits() == 8 else torch.int16, device=dev)
                keys, dig32 = self._policy_keys(policy)
                ws = self._shuffle_workspace()
                L.check(L.load().offsim_shuffle_queues_keys_ws(C.byref(t.c), L.ptr(sd), self.R, L.ptr(dig32), self._stream_format(), L.ptr(self._dig_buf),
                                                               L.ptr(self._loc_buf), L.ptr(self._init_perm_buf), L.ptr(ws), 0 if ws is None else ws.numel(),
                                                               L.stream_ptr()))
                self._streams = dict(dig=self._dig_buf, dig_stride=t.N, loc=self._loc_buf, loc_stride=t.N, key=self._policy_key(policy),
                                     format=self._stream_format())
                self.state.set_orders(None, 0, self._init_perm_buf, t.N0)
                self._perm_lazy = "streams"
            else:
                self._dig_buf = self._loc_buf = None
                if self._perm_buf is None or self._perm_buf.shape[0] != self.R:
                    self._perm_buf = torch.empty((self.R, max(t.N, 1)), dtype=torch.int32, device=dev)
                shuffle_queues(t, sd, self._perm_buf, self._init_perm_buf, workspace=self._shuffle_workspace())
                self.state.set_orders(self._perm_buf, t.N, self._init_perm_buf, t.N0)
        elif shuffle == SHUFFLE_SHARED:
            assert shuffle_seed is not None
            perm, init_perm = shuffle_queues(t, seeds_tensor([shuffle_seed], dev), workspace=self._shuffle_workspace(1))
            self._perm_buf, self._init_perm_buf = perm, init_perm
            self.state.set_orders(perm, 0, init_perm, 0)
            if keyed:  # one shared order: the streams are one row, built from the permutation
                keys, dig32 = self._policy_keys(policy)
                p = perm[0, :t.N].to(torch.int64) & 0xFFFFFFFF
                dg, lc = self._pack_streams(dig32, p)
                self._streams = dict(dig=dg, dig_stride=0, loc=lc, loc_stride=0, key=self._policy_key(policy), format=self._stream_format())
        elif shuffle == SHUFFLE_NONE:
            self.state.set_orders(None, 0, None, 0)
            if keyed:
                keys, dig32 = self._policy_keys(policy)
                self._streams = self._table_order_streams(dig32, self._policy_key(policy))
        else:
            raise ValueError(shuffle)

    def _shuffle_workspace(self, n_orders=None):
        """Workspace of the chunked shuffle (states, or an init queue, of more than 65536 rows, csrc/shuffle_chunk.hpp): pools for up to
        four persistent workgroups per compute unit -- no more than there are chains (`n_orders` queue orders x (states + 1)), nor than
        keeps each busy with about four of the longest -- within a budget of the free HBM (`ws_budget_frac`, default 92 %, and never the
        last `ws_keep_free` bytes, default 2 GiB: the policy's key buffer, rebuilt permutations, snapshots and outputs of the same
        job are allocated later; a workgroup's pools are ~22 bytes per row of the longest chain).  None (the in-place shuffle) when the
        table has no such chain, when OFFSIM_SHUFFLE_CHUNKED=0, or when not even one workgroup's pools fit the budget."""
        t = self.table
        if max(t.max_seg, t.N0) <= 65536 or os.environ.get("OFFSIM_SHUFFLE_CHUNKED", "1") == "0":
            return None
        if not L.lds_order_ok(t.device):  # (the chunked kernel's one-exchange-per-lane apply needs the property: in-place shuffle)
            return None
        lib = L.load()
        want = torch.cuda.get_device_properties(t.device).multi_processor_count * 4
        n_orders = self.R if n_orders is None else n_orders
        want = min(want, max(1, n_orders * (t.n_slots + 1)))
        # (no more workgroups than keeps each busy with about four chains of the longest kind: a small job does not wait for gigabytes
        # of pools to be allocated).  Format C sends EVERY chain of the table through the chunked kernel, not only the long ones.
        long_rows = (t.N + t.N0) if stream_format(t) == L.STREAMS_C else getattr(t, "long_rows", t.N)
        want = min(want, max(8, n_orders * long_rows // (4 * max(t.max_seg, t.N0, 1))))
        if getattr(self, "_ws", None) is None or getattr(self, "_ws_wg", 0) < want:
            one = int(lib.offsim_shuffle_workspace_bytes(C.byref(t.c), 1))
            head = 2 * one - int(lib.offsim_shuffle_workspace_bytes(C.byref(t.c), 2))  # header bytes
            if one <= 0:
                return None
            have_bytes = 0 if getattr(self, "_ws", None) is None else self._ws.numel()
            self._ws = None  # (released first: what it held counts as free)
            torch.cuda.empty_cache()
            free = torch.cuda.mem_get_info(t.device)[0]
            budget = min(int(free * getattr(self, "ws_budget_frac", 0.92)), free - int(getattr(self, "ws_keep_free", 2 << 30)))
            n = min(want, max(0, (budget - head) // max(one - head, 1)))
            if n < 1 and have_bytes:  # nothing bigger fits: what there was is put back
                n = max(0, (have_bytes - head) // max(one - head, 1))
            if n < 1:
                self._ws_wg = want
                return None
            self._ws = torch.empty(int(lib.offsim_shuffle_workspace_bytes(C.byref(t.c), int(n))), dtype=torch.uint8, device=t.device)
            self._ws_wg = want  # (what was asked for: a smaller grant is not asked for again)
        return self._ws

    # ---- candidate streams for the row-packed scan ----
    def _streams_apply(self, policy):
        """Whether the candidate streams and the row-packed scan serve `policy` on this table: what offsim_eval_mc_streams covers (f64
        probabilities, the default reject rule, <= 256 states, states of up to 2^23 rows), and -- unless OFFSIM_SCAN_ROWS forces it --
        where that kernel is the faster one (below)."""
        t = self.table
        p = policy if isinstance(policy, torch.Tensor) else np.asarray(policy)
        f64 = p.dtype in (torch.float64, np.float64, np.dtype(np.float64))
        # Which scan: the row-packed kernel tops a state's 8-entry window up once per tick of 16 steps, the window kernel (one rollout
        # per wavefront, csrc/scan_win.hpp) refills on the spot.  A row whose window gives no clear accept -- dry, or every entry rejected --
        # costs the whole wavefront a trip to memory, and how often that happens is a matter of how many candidates a tick takes out of
        # the busiest window: 16 steps x (share of the steps that visit the state = its share of the rows) / acceptance.  Measured at the
        # end of round 4 (tools/sweep_kernel_choice.sh, profiles/r04_kernel_choice_sweep.txt; 10 M rows, equal states, 1024 rollouts,
        # scan + reset seconds per pass, row-packed / window kernel), with that load L in brackets:
        #   162 states, acceptance 0.54 [0.18]: 0.98 / 1.40    0.38 [0.26]: 1.03 / 1.33    0.29 [0.34]: 1.15 / 1.34    0.24 [0.40]: 1.21 / 1.33
        #   acceptance 0.54, 50 states [0.59]: 1.21 / 1.52    35 [0.85]: 1.32 / 1.52    25 [1.19]: 1.50 / 1.53    12 [2.5]: 2.08 / 1.57
        #   50 states at 0.24 [1.31]: 1.45 / 0.78    25 states at 0.38 [1.71]: 1.69 / 1.13
        # So: the row-packed kernel while L < 1.2.  (Round 3's rule -- no state above 3 % of the rows AND acceptance >= 0.4 -- was fitted
        # to a kernel whose dry rows went through the C++ path; with the in-loop handler a low acceptance alone no longer decides.)
        # OFFSIM_SCAN_ROWS = 1 / 0 forces the one or the other.
        mode = os.environ.get("OFFSIM_SCAN_ROWS", "auto")
        ok = (f64 and self.reject_mode == L.REJECT_DEFAULT and t.n_slots <= 256 and 0 < t.max_seg <= (1 << 23) and t.N < 2 ** 32 - 1
              and mode != "0")
        ok = ok and L.lds_order_ok(t.device)  # (runtime guard of the tick's lane-ordered LDS atomic; never forced past)
        if not ok or mode == "1":
            return ok
        # Round 5 (tools/diag_scan.py, profiles/r05_diag_scan_c2_c3_c5.txt): L says nothing about a window that is FULL and still gives no
        # clear accept -- eight entries are all rejected with probability (1 - acceptance)^8: 0.2 % of the looks at 0.54, 6.5 % at
        # 0.29 (C5's shard: 50 states, 4 actions, L = 1.1 -- the row-packed kernel ran at 1210 cycles per iteration, 13 % of its
        # row-steps without a clear accept: 0.878 + 0.433 s per pass against 0.755 + 0.257 s for the window kernel).  Up to 64 states
        # the window kernel keeps 32 entries per state, where this cannot happen; beyond 64 it has 8 as well and the row-packed
        # kernel stays ahead at any acceptance (162 states at 0.24: 1.08 against 1.17 s).
        acc = max(self._acceptance(policy), 1e-9)
        if t.n_slots <= 64 and (1.0 - min(acc, 1.0)) ** 8 >= ROWS_MAX_ALL_REJECTED:
            return False
        return ROWS_TICK_STEPS * (t.max_seg / max(t.N, 1)) / acc < ROWS_MAX_WINDOW_LOAD

    def _acceptance(self, policy):
        """Acceptance probability of a candidate under `policy`, averaged over the table's rows: the mean of the compiled thresholds'
        top 21 bits (one reduction and one host read per policy; cached with the compiled keys)."""
        key = self._policy_key(policy)
        if getattr(self, "_acc_cache", None) is None or self._acc_cache[0] != key:
            keys, _ = self._policy_keys(policy, key=key)
            a = float((((keys >> 43) & 0x1FFFFF).to(torch.float64)).mean().item()) / 2 ** 21 if self.table.N else 1.0
            self._acc_cache = (key, a)
        return self._acc_cache[1]

    def _stream_format(self):
        """Layout of the candidate streams (module function stream_format; settled at first use: buffers written in one layout are
        read in that layout whatever the environment says later)."""
        f = self.__dict__.get("_fmt")
        if f is None:
            f = stream_format(self.table)
            # format C is written by the chunked shuffle only (offsim_shuffle_queues_keys_ws refuses it without a workspace that holds
            # at least one workgroup's pools, and with an init queue beyond 2^23 rows): decided once the workspace is known, BEFORE the
            # loc stream is allocated in either width -- under memory pressure the table takes format B and the in-place shuffle
            if f == L.STREAMS_C and (self.table.N0 > (1 << 23) or not self._workspace_holds_a_workgroup(self._shuffle_workspace())):
                f = L.STREAMS_B
            self._fmt = f
        return f

    def _workspace_holds_a_workgroup(self, ws):
        return ws is not None and ws.numel() >= int(L.load().offsim_shuffle_workspace_bytes(C.byref(self.table.c), 1)) > 0

    def _loc_bits(self):
        """Bits of the local row the loc stream holds (the others travel inside the digest: formats B, C)."""
        return 8 if self._stream_format() == L.STREAMS_C else 16

    def _local_rows(self, loc, dig):
        """Local rows (int64) out of slices of the two streams."""
        lb = self._loc_bits()
        local = loc.to(torch.int64) & ((1 << lb) - 1)
        if self._stream_format() != L.STREAMS_A:
            dg = dig.to(torch.int64)
            local |= (((dg >> 8) & 3) | (((dg >> 11) & (0x7F if lb == 8 else 0x1F)) << 2)) << lb
        return local

    @staticmethod
    def _policy_key(policy):
        """What identifies the tabular policy the streams / compiled keys were made for: shape, dtype and the bytes themselves (a
        few KB; compared for equality, not by hash)."""
        p = policy.detach().cpu().numpy() if isinstance(policy, torch.Tensor) else np.asarray(policy)
        return (p.shape, p.dtype.str, np.ascontiguousarray(p).tobytes())

    def _policy_keys(self, policy, key=None):
        """(compiled 64-bit keys, their 32-bit digests) of `policy` on the device, cached per policy."""
        k = self._policy_key(policy) if key is None else key
        if getattr(self, "_pk_cache", None) is None or self._pk_cache[0] != k:
            t = self.table
            pi_d = torch.as_tensor(np.ascontiguousarray(policy) if not isinstance(policy, torch.Tensor) else policy,
                                   dtype=torch.float64).to(t.device).reshape(t.n_slots, t.nA).contiguous()
            keys = self.compile_policy(pi_d)
            if getattr(self, "_dig32", None) is None:
                self._dig32 = torch.empty(max(t.N, 1), dtype=torch.int32, device=t.device)
            L.check(L.load().offsim_compile_digests(C.byref(t.c), L.ptr(keys), self._stream_format(), L.ptr(self._dig32), L.stream_ptr()))
            self._pk_cache = (k, keys, self._dig32)
        return self._pk_cache[1], self._pk_cache[2]

    def _derive_streams(self, policy, max_entries=1 << 26, key=None):
        """Candidate streams from queue orders that exist as permutations (reset_sampler without `policy`): one gather, done
        for jobs of up to `max_entries` queue positions; bigger jobs pass `policy` to reset_sampler or run the window kernels."""
        t, st = self.table, self.state
        if not self._streams_apply(policy) or (st.perm is None and self._perm_lazy == "streams"):
            return
        n_rows = 1 if (st.perm is None or st.perm_stride == 0) else self.R
        if n_rows * t.N > max_entries:
            return
        key = self._policy_key(policy) if key is None else key
        keys, dig32 = self._policy_keys(policy, key=key)
        if st.perm is None:  # table order
            self._streams = self._table_order_streams(dig32, key)
            return
        p = st.perm.reshape(n_rows, -1)[:, :t.N].to(torch.int64) & 0xFFFFFFFF
        dg, lc = self._pack_streams(dig32, p)
        self._streams = dict(dig=dg, dig_stride=t.N if n_rows > 1 else 0, loc=lc, loc_stride=t.N if n_rows > 1 else 0, key=key,
                             format=self._stream_format())

    def _rekey_streams(self, policy, key):
        """reset_sampler(policy=A) laid the queue orders out as A's candidate streams and another policy is evaluated on the same
        sampler state (the reference allows it: the queues just go on, psrs.py:241-271 takes any pi): the orders -- the local rows --
        stay, the digest of every queue position is replaced IN PLACE by the new policy's, a few rollouts at a time (no second set
        of resident buffers)."""
        t = self.table
        _, dig32 = self._policy_keys(policy, key=key)
        base = self._seg_base()
        step = max(1, (128 << 20) // max(t.N * 8, 1))
        for b in range(0, self.R, step):
            local = self._local_rows(self._loc_buf[b:b + step, :t.N], self._dig_buf[b:b + step, :t.N])
            self._dig_buf[b:b + step, :t.N] = self._pack_streams(dig32, local + base[None, :])[0]
        self._streams = dict(dig=self._dig_buf, dig_stride=t.N, loc=self._loc_buf, loc_stride=t.N, key=key, format=self._stream_format())

    def _pack_streams(self, dig32, p):
        """(dig, loc) streams of queue orders given as grouped rows p [..., N] (int64): the digest of the row at every position -- in
        formats B / C with bits 16.. / 8.. of its local row in the digest's bits 8, 9, 11.. -- and the local row's low 16 / 8 bits."""
        local = p - (self._seg_base() if p.dim() == 1 else self._seg_base()[None, :])
        dg = dig32[p].to(torch.int64) & 0xFFFFFFFF
        lb = self._loc_bits()
        if self._stream_format() != L.STREAMS_A:
            h = local >> lb
            dg = dg | ((h & 3) << 8) | ((h >> 2) << 11)
        dg = torch.where(dg >= 2 ** 31, dg - 2 ** 32, dg).to(torch.int32)
        lo = local & ((1 << lb) - 1)
        return dg.contiguous(), (lo.to(torch.uint8) if lb == 8 else torch.where(lo >= 2 ** 15, lo - 2 ** 16, lo).to(torch.int16)).contiguous()

    def _table_order_streams(self, dig32, key):
        """Streams of queues in table order (no shuffle): format A needs no loc stream (the local row is the queue position)."""
        if self._stream_format() == L.STREAMS_A:
            return dict(dig=dig32, dig_stride=0, loc=None, loc_stride=0, key=key, format=L.STREAMS_A)
        dg, lc = self._pack_streams(dig32, torch.arange(self.table.N, device=self.table.device, dtype=torch.int64))
        return dict(dig=dg, dig_stride=0, loc=lc, loc_stride=0, key=key, format=self._stream_format())

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
        return (self._local_rows(self._loc_buf, self._dig_buf) + self._seg_base()[None, :]).to(torch.int32)

    def set_rejection_seeds(self, seeds, provider="pcg64"):
        """Replace only the rejection streams (env.rejection_sampling_rng = ..., psrs.py:20 is a plain attribute).
        provider = "pcg64": default_rng(seed) -- the reference's numbers.  provider = "philox": rocRAND's Philox4x32-10 through its
        device API (include/offsim.h OFFSIM_STREAM_PHILOX): another, equally valid sample path, taken by step / step_single / eval_td
        and the generic eval_mc (the compiled-policy scans draw from PCG64 only); reset_sampler puts PCG64 back."""
        self._quiesce()
        sd = seeds_tensor(seeds, self.table.device)
        assert sd.numel() == self.R, "one seed per rollout"
        if provider == "pcg64":
            seed_streams(sd, self.state.rng)
            self.state.rng_kind = L.STREAM_PCG64
        elif provider == "philox":
            self.state.rng.zero_()
            self.state.rng[:, 0] = sd  # seed, draws consumed, 0, 0
            self.state.rng_kind = L.STREAM_PHILOX
        else:
            raise ValueError(provider)
        self.state._refresh()

    def _orders_for_generic(self):
        """The kernels that take any p_new per step (step, step_single, eval_td, the generic eval_mc) walk the queues through
        permutations.  After reset_sampler(policy=...) the orders exist only as candidate streams: the permutations are rebuilt
        from the streams' local rows here, once (4 * R * N bytes) -- never left as table order by default."""
        if self.state.perm is None and self._perm_lazy == "streams":
            self._quiesce()
            self.state.set_orders(self.perm.contiguous(), self.table.N, self.state.init_perm, self.state.init_stride)
            self._perm_buf = self.state.perm

    # -- PSRS.reset (psrs.py:32-37) --
    def reset(self, mask=None):
        self._quiesce()
        m = None if mask is None else mask.to(torch.uint8).contiguous()
        L.check(L.load().offsim_env_reset(C.byref(self.table.c), C.byref(self.state.c), L.ptr(m), L.ptr(self._row), L.stream_ptr()))
        return self._row

    # -- PSRS.step (psrs.py:39-51) --
    def step(self, p_new, advance=True, reject_mode=None):
        """p_new: [R,nA] tensor/array.  Returns device tensors (row, status, popped)."""
        self._quiesce()
        self._orders_for_generic()
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
        """R = 1 convenience for the drop-in classes (per_state_rejection.py:85-95 is one Python call per simulated step): the step is
        served by a RESIDENT wavefront (offsim_step_server_start: no kernel launch, no stream synchronise per call -- ~27 us before)
        through a mailbox in host-coherent pinned memory; the server is started on first use, ends by itself when idle, and is stopped
        before anything else touches this environment's state (`_quiesce`).  OFFSIM_STEP_SERVER=0, more than 24 actions, or R != 1: one
        launch per call, p_new and the results in pinned mapped memory.  Returns host ints (row, status, popped).
        PITFALL: while the server is up (until ~20-40 ms after the last step) any DEVICE-WIDE synchronisation in the caller's own code
        between two steps -- torch.cuda.synchronize(), empty_cache / hipFree, hipHostFree -- waits for that idle timeout; a loop that
        must synchronise the device every step sets OFFSIM_STEP_SERVER=0 (one launch per call, ~27 us) or synchronises its own stream."""
        t = self.table
        p_new = np.asarray(p_new)
        mode = _prob_mode(t, p_new.dtype)
        rm = self.reject_mode if reject_mode is None else reject_mode
        srv = self.__dict__.get("_srv_enabled")
        if srv is None:  # (looked up once per environment)
            srv = self._srv_enabled = self.R == 1 and t.nA <= L.MAILBOX_MAX_ACTIONS and os.environ.get("OFFSIM_STEP_SERVER", "1") != "0"
        if srv and (advance or rm == L.REJECT_NEVER):
            return self._server_step(p_new, mode, L.SERVER_CMD_STEP if advance else L.SERVER_CMD_POP_ONE, rm)
        self._quiesce()
        self._orders_for_generic()
        key = (mode, t.nA)
        if getattr(self, "_single_key", None) != key:
            dt = torch.float32 if mode == L.PROB_F32 else torch.float64
            self._p_host = torch.empty((1, t.nA), dtype=dt).pin_memory()
            self._o_host = torch.empty(3, dtype=torch.int32).pin_memory()
            self._p_np, self._o_np = self._p_host.numpy(), self._o_host.numpy()
            self._single_key = key
        self._p_np[0, :] = p_new.reshape(-1)
        base = self._o_host.data_ptr()
        L.check(L.load().offsim_step_batch(C.byref(t.c), C.byref(self.state.c), self._p_host.data_ptr(), mode, rm, 1 if advance else 0,
                                           base, base + 4, base + 8, L.stream_ptr()))
        torch.cuda.current_stream().synchronize()
        row, status, popped = int(self._o_np[0]), int(self._o_np[1]), int(self._o_np[2])
        self.last_row = row
        return row, status, popped

    # ---- the resident step server (include/offsim.h: offsim_step_server_start) ----
    _SERVER_IDLE_POLLS = 20000  # polls of ~1-2 us without a request before the server ends by itself

    def _server_start(self, mode):
        lib = L.load()
        if getattr(self, "_mb", None) is None:
            ptr = C.c_void_p()
            L.check(lib.offsim_host_alloc(C.sizeof(L.StepMailbox), C.byref(ptr)))
            self._mb_ptr = ptr.value
            self._mb = L.StepMailbox.from_address(ptr.value)
            view = lambda f, n: np.ctypeslib.as_array((C.c_double * n).from_address(ptr.value + getattr(L.StepMailbox, f).offset))
            self._mb_head64, self._mb_tail64 = view("p_head", 5), view("p_tail", L.MAILBOX_MAX_ACTIONS - 5)
            self._mb_head32, self._mb_tail32 = self._mb_head64.view(np.float32), self._mb_tail64.view(np.float32)
            self._srv_p = np.zeros(L.MAILBOX_MAX_ACTIONS, np.float64)  # staging of p_new at a fixed address (f32: the same bytes, packed)
            self._srv_p32, self._srv_p_addr = self._srv_p.view(np.float32), self._srv_p.ctypes.data
            self._srv_out = (C.c_int32 * 3)()
            self._srv_out_addr = C.addressof(self._srv_out)
            self._srv_call = lib.offsim_step_server_call
            self._srv_stream = torch.cuda.Stream(device=self.table.device)
        self._orders_for_generic()
        self._srv_stream.wait_stream(torch.cuda.current_stream())  # (everything enqueued so far: sampler reset, env.reset, ...)
        self._mb.seq_in2 = self._mb.seq_out
        self._mb.seq_in = self._mb.seq_out  # (nothing pending)
        L.check(lib.offsim_step_server_start(C.byref(self.table.c), C.byref(self.state.c), self._mb_ptr, mode, self._SERVER_IDLE_POLLS,
                                             self._srv_stream.cuda_stream))
        self._srv_mode = mode
        self._srv_last = (None, None)

    def reset_single(self):
        """PSRS.reset (psrs.py:32-37) for the R = 1 environment, as a host int (the initial row or -1): served by the resident step
        server when it is up (no stop / start around an episode end), otherwise offsim_env_reset."""
        mb = getattr(self, "_mb", None)
        if self.R == 1 and mb is not None and mb.state in (L.SERVER_STARTING, L.SERVER_RUNNING):
            return self._server_step(None, self._srv_mode, L.SERVER_CMD_RESET, self.reject_mode)[0]
        return int(self.reset().cpu()[0])

    def _server_step(self, p_new, mode, cmd, rm):
        mb = self.__dict__.get("_mb")
        if mb is None or mb.state not in (L.SERVER_STARTING, L.SERVER_RUNNING) or self._srv_mode != mode:
            self._quiesce()
            self._server_start(mode)
            mb = self._mb
            self._srv_last = (None, None)
        # the request itself -- payload, command, the two sequence words, the spin on the answer -- is ONE foreign call
        n = 0
        if p_new is not None:
            n = p_new.size
            (self._srv_p32 if mode == L.PROB_F32 else self._srv_p)[:n] = p_new.reshape(-1)
        rc = self._srv_call(self._mb_ptr, self._srv_p_addr if n else None, n, mode, cmd, rm, 2_000_000_000, self._srv_out_addr)
        if rc == 0:
            out = self._srv_out
            self.last_row = out[0]
            return out[0], out[1], out[2]
        if rc != L.SERVER_GONE:
            L.check(rc)
        # the server ended (idle) between our look at its state and the request, which stays posted: start it again; it serves it
        seq = mb.seq_in
        self._srv_stream.synchronize()
        self._server_start(mode)
        mb.seq_in2 = seq
        mb.seq_in = seq
        spins, t_end = 0, None
        while mb.seq_out != seq:  # (bounded by wall-clock time, like the C side: OFFSIM_SERVER_ANSWER_SECONDS)
            spins += 1
            if (spins & 0xFFF) == 0:
                now = time.monotonic()
                t_end = now + L.SERVER_ANSWER_SECONDS if t_end is None else t_end
                if now > t_end:
                    raise L.OffsimError("the resident step server does not answer")
        row = mb.row
        self.last_row = row
        return row, mb.status, mb.popped

    def _quiesce(self):
        """Stop the resident step server (if it runs) before anything else reads or writes this environment's state: it owns the
        rollout's cursor / stream / state rows while it is up."""
        mb = getattr(self, "_mb", None)
        if mb is None or mb.state not in (L.SERVER_STARTING, L.SERVER_RUNNING):
            return
        mb.cmd = L.SERVER_CMD_EXIT
        self._srv_last = (None, None)
        mb.seq_in2 = (mb.seq_in + 1) & 0xFFFFFFFF
        mb.seq_in = mb.seq_in2
        self._srv_stream.synchronize()  # (it ends on the command, or has ended by itself)
        mb.seq_out = mb.seq_in
        torch.cuda.current_stream().wait_stream(self._srv_stream)

    def __del__(self):
        try:
            if getattr(self, "_mb", None) is not None:
                self._quiesce()
                L.load().offsim_host_free(self._mb_ptr)
                self._mb = None
        except Exception:
            pass

    def set_state(self, slots, mask=None):
        self._quiesce()
        s = slots.to(device=self.table.device, dtype=torch.int32).contiguous()
        m = None if mask is None else mask.to(torch.uint8).contiguous()
        L.check(L.load().offsim_env_set_state(C.byref(self.state.c), L.ptr(s), L.ptr(m), L.stream_ptr()))

    # -- evalMC_psrs (psrs.py:241-271) for all rollouts in one launch --
    def eval_mc(self, pi_slots, gamma, n_episodes=None, ep_cap=0, trace_cap=0, n_gamma_pow=4096, out=None, fast=None, dbg=False):
        """pi_slots: [n_slots,nA] policy per state slot (TransitionTable.policy_slots).  Returns a dict of device
        tensors: sum_g, n_ep, steps, cand, n_len, status (+ ep_g, ep_len, trace_row, trace_pop when asked).
        fast=None picks the compiled-policy / LDS-window kernel (offsim_eval_mc_keys) whenever it applies
        (f64 probabilities, default reject rule, <= 256 states); fast=False forces the generic kernel."""
        self._quiesce()
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
        can_fast = mode == L.PROB_F64 and self.reject_mode == L.REJECT_DEFAULT and t.n_slots <= 256 and self.state.rng_kind == L.STREAM_PCG64
        if fast is None:
            fast = can_fast
        if fast and not can_fast:
            raise L.OffsimError("the compiled-policy scan needs f64 probabilities, the default reject rule, <= 256 states and the PCG64 stream")
        pkey = self._policy_key(pi_slots) if fast else None  # (once per call: a device tensor is copied to the host for it)
        if fast and not (self._streams is not None and self._streams["key"] == pkey):
            if self.state.perm is None and self._perm_lazy == "streams":
                self._rekey_streams(pi_slots, pkey)  # the orders exist only as another policy's streams: same orders, this policy's digests
            else:
                self._derive_streams(pi_slots, key=pkey)  # small jobs: the streams are gathered from the permutations on the spot
        rows = bool(fast) and self._streams is not None and self._streams["key"] == pkey
        if rows:  # the sampler reset laid the orders out as candidate streams for this policy: row-packed scan
            keys, _ = self._policy_keys(pi_slots, key=pkey)
            sm = self._streams
            smc = L.Streams(dig=L.ptr(sm["dig"]), dig_stride=sm["dig_stride"], loc=L.ptr(sm["loc"]), loc_stride=sm["loc_stride"],
                            format=sm.get("format", L.STREAMS_A))
            L.check(L.load().offsim_eval_mc_streams(C.byref(t.c), C.byref(self.state.c), C.byref(smc), L.ptr(keys), float(gamma), L.ptr(gp),
                                                    gp.numel(), int(n_episodes), C.byref(oc), L.stream_ptr()))
            o["_keepalive"] = (pi_d, gp, keys, sm)
        elif fast:
            keys = self.compile_policy(pi_d)
            L.check(L.load().offsim_eval_mc_keys(C.byref(t.c), C.byref(self.state.c), L.ptr(keys), float(gamma), L.ptr(gp),
                                                 gp.numel(), int(n_episodes), C.byref(oc), L.stream_ptr()))
            o["_keepalive"] = (pi_d, gp, keys)
        else:
            self._orders_for_generic()  # (materialise the permutations)
            L.check(L.load().offsim_eval_mc(C.byref(t.c), C.byref(self.state.c), L.ptr(pi_d), mode, self.reject_mode, float(gamma),
                                            L.ptr(gp), gp.numel(), int(n_episodes), C.byref(oc), L.stream_ptr()))
            o["_keepalive"] = (pi_d, gp)
        return o

    # -- qlearn_psrs / expSARSA_psrs (psrs.py:119-239) with a Q-independent behaviour policy, all rollouts in one launch --
    def eval_td(self, pi_slots, gamma, mode, alpha, q_slots=None, n_episodes=None, ep_cap=0, trace_cap=0, n_gamma_pow=4096,
                behaviour=L.BEHAVIOUR_FIXED, epsilon=0.0, alpha_ep=None, epsilon_ep=None, snap_cap=0, snap_stride=1, tie_mt=None):
        """mode: _lib.TD_QLEARN | _lib.TD_EXPSARSA.  q_slots [R,n_slots,nA] f64 (Q_init; zeros if None) is updated in place
        and returned as out["q"]; out["td_err"] [R,trace_cap] holds the TD errors in step order.
        behaviour = _lib.BEHAVIOUR_EPS_GREEDY / BEHAVIOUR_SOFT_GREEDY: every rollout acts on its own Q table (the learner-in-the-loop
        case of psrs.py:158); pi_slots is then only the target policy of expected SARSA.  alpha_ep / epsilon_ep: per-episode
        schedules (psrs.py:128-135); snap_cap > 0: out["q_snap"] [R,snap_cap,n_slots,nA] = Q after every snap_stride-th step
        (save_Q); tie_mt [R,625] int32/uint32: NumPy's MT19937 state per rollout for ties between maxima (advanced in place)."""
        self._quiesce()
        self._orders_for_generic()
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
            if behaviour == L.BEHAVIOUR_EPS_GREEDY:
                o["beh_arg"] = torch.zeros((R, trace_cap), dtype=torch.int32, device=dev)
        a_ep = None if alpha_ep is None else torch.as_tensor(np.ascontiguousarray(alpha_ep, dtype=np.float64)).to(dev)
        e_ep = None if epsilon_ep is None else torch.as_tensor(np.ascontiguousarray(epsilon_ep, dtype=np.float64)).to(dev)
        n_sched = max(a_ep.numel() if a_ep is not None else 0, e_ep.numel() if e_ep is not None else 0)
        assert all(x is None or x.numel() == n_sched for x in (a_ep, e_ep)), "alpha_ep and epsilon_ep cover the same episodes"
        if snap_cap:
            o["q_snap"] = torch.zeros((R, snap_cap, t.n_slots, t.nA), dtype=torch.float64, device=dev)
        mt = None
        if tie_mt is not None:
            mt = torch.as_tensor(np.ascontiguousarray(tie_mt).view(np.int32) if not isinstance(tie_mt, torch.Tensor) else tie_mt)
            mt = mt.to(device=dev, dtype=torch.int32).reshape(R, 625).contiguous()
            o["tie_mt"] = mt
        gp = _gamma_pow(gamma, n_gamma_pow, dev, cap=t.N + 2)
        oc = L.EvalMCOut(sum_g=L.ptr(o["sum_g"]), n_ep=L.ptr(o["n_ep"]), steps=L.ptr(o["steps"]), cand=L.ptr(o["cand"]),
                         n_len=L.ptr(o["n_len"]), status=L.ptr(o["status"]), ep_g=L.ptr(o.get("ep_g")), ep_len=L.ptr(o.get("ep_len")),
                         ep_cap=ep_cap, trace_row=L.ptr(o.get("trace_row")), trace_pop=L.ptr(o.get("trace_pop")), trace_cap=trace_cap)
        tdc = L.TD(mode=mode, alpha=float(alpha), q=L.ptr(q), td_err=L.ptr(o.get("td_err")), td_cap=trace_cap, behaviour=int(behaviour),
                   epsilon=float(epsilon), alpha_ep=L.ptr(a_ep), epsilon_ep=L.ptr(e_ep), n_sched=n_sched, q_snap=L.ptr(o.get("q_snap")),
                   snap_cap=snap_cap, snap_stride=max(int(snap_stride), 1), tie_mt=L.ptr(mt), beh_arg=L.ptr(o.get("beh_arg")))
        L.check(L.load().offsim_eval_td(C.byref(t.c), C.byref(self.state.c), L.ptr(pi_d), self.reject_mode, float(gamma), L.ptr(gp),
                                        gp.numel(), int(n_episodes), C.byref(oc), C.byref(tdc), L.stream_ptr()))
        o["q"] = q
        o["_keepalive"] = (pi_d, gp, a_ep, e_ep)
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


def rollout_resident_bytes(table, keyed=True):
    """HBM one rollout keeps resident between reset_sampler and the scan: its queue orders (candidate streams: 4 + 2 bytes per queue
    position; as permutations: 4), its init order, cursors and random-stream state."""
    per_pos = 4 if not keyed else 5 if stream_format(table) == L.STREAMS_C else 6
    return int(table.N) * per_pos + int(table.N0) * 4 + int(table.n_slots) * 4 + 64


def resident_rollouts(table, keyed=True, free_bytes=None):
    """How many rollouts' queue orders the free HBM of the table's device holds at once (shared by bench.py and evalmc_rollouts):
    rollout_resident_bytes each, after 2 GiB for everything else of the job (policy keys, outputs, rebuilt permutations) and the
    chunked shuffle's workspace -- pools for up to 1024 persistent workgroups, at most a tenth of what is free -- which
    `_shuffle_workspace` allocates AFTER the stream buffers and which a table with chains above 65536 rows (or in stream format C,
    where a missing workspace would cost the format) must still find room for.  Returns (rollouts, bytes per rollout, free, total)."""
    free_b, total_b = torch.cuda.mem_get_info(table.device)
    if free_bytes is not None:
        free_b = int(free_bytes)
    per = max(rollout_resident_bytes(table, keyed=keyed), 1)
    ws_b = min(int(L.load().offsim_shuffle_workspace_bytes(C.byref(table.c), 1024)), free_b // 10)
    return int(max(0, free_b - (2 << 30) - ws_b) // per), per, int(free_b), int(total_b)


def evalmc_rollouts(table, seeds, pi, gamma, shuffle=SHUFFLE_PER_ROLLOUT, shuffle_seed=None, tile=None,
                    reject_mode=L.REJECT_DEFAULT, n_episodes=None):
    """evalMC_psrs for many sampler seeds.  Rollouts are processed in tiles of `tile` seeds so that the per-rollout queue orders
    (rollout_resident_bytes each) fit the device's free memory.  Returns host arrays: sum_g, n_ep, steps, cand, status
    and value = sum_g / n_ep (the per-seed value estimate, Gs.mean())."""
    seeds = np.asarray(seeds, dtype=np.uint64)
    R = len(seeds)
    if tile is None:
        tile = R if shuffle != SHUFFLE_PER_ROLLOUT else int(max(1, min(R, resident_rollouts(table, keyed=True)[0])))
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
        L.check_async_faults()  # (the copies above synchronised the stream)
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
