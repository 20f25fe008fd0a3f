"""The batched PSRS engine on the MI355X: R independent PSRS environments over one device table, stepped by HIP kernels.

  BatchedPSRS       reset_sampler / reset / step / eval_mc / eval_td for R rollouts at once (offsim4rl/evaluators/psrs.py:5-57,
                    119-271 for every rollout), candidate streams and their layouts, the resident step server
  evalmc_rollouts   the batched driver behind the headline metric (thousands of seeds per launch, tiled to the free HBM)
  resident_rollouts / rollout_resident_bytes   the one tile-size rule (shared with bench.py)

Every decision (queue order, accept / reject, state walk, discounted return) is computed on the GPU through the C ABI of
include/offsim.h; the host only moves arguments and payload.  No CPU fallback.  The reference's single-environment class and its
drivers (PSRS, evalMC_psrs, qlearn_psrs, expSARSA_psrs) are in psrs.py, on top of this module.
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
    def reset_sampler(self, seeds, shuffle=SHUFFLE_PER_ROLLOUT, shuffle_seed=None, policy=None, rejection="pcg64"):
        """`policy` (optional, [n_slots,nA] f64): the tabular policy the following eval_mc calls will evaluate.  It changes no
        result; it lets the sampler reset write the queue orders as the candidate streams the row-packed scan reads
        sequentially (offsim_shuffle_queues_keys: digest + 16-bit local row per queue position) instead of as permutations.
        (step / step_single / eval_td / the generic eval_mc need permutations and rebuild them from the streams on first use.)
        `rejection`: the provider of the rejection stream the rollouts start with -- "pcg64" = default_rng(seed) (psrs.py:20, the
        reference's numbers) or "philox" = rocRAND's Philox4x32-10 of the same seeds (set_rejection_seeds; the queue orders stay
        NumPy's).  Every scan takes either provider; which scan a table gets does not depend on it."""
        if rejection not in ("pcg64", "philox"):
            raise ValueError(rejection)
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
        if rejection == "philox":
            self.set_rejection_seeds(sd, provider="philox")

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
        ok = (f64 and self.reject_mode == L.REJECT_DEFAULT and t.n_slots <= (256 if t.max_seg <= 65536 else 255) and 0 < t.max_seg <= (1 << 23)
              and t.N < 2 ** 32 - 1 and mode != "0")  # (formats B and C: 255 states)
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
        device API (include/offsim.h OFFSIM_STREAM_PHILOX): another, equally valid sample path, taken by every kernel (step /
        step_single / eval_td, the generic eval_mc, the row-packed scan and the window kernel); reset_sampler puts PCG64 back unless it
        is told rejection="philox"."""
        self._quiesce()
        sd = seeds.to(self.table.device) if isinstance(seeds, torch.Tensor) else seeds_tensor(seeds, self.table.device)
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
        can_fast = mode == L.PROB_F64 and self.reject_mode == L.REJECT_DEFAULT and t.n_slots <= 256  # (either stream provider)
        if fast is None:
            fast = can_fast
        if fast and not can_fast:
            raise L.OffsimError("the compiled-policy scan needs f64 probabilities, the default reject rule and <= 256 states")
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
            o["_kernel"] = "k_eval_mc"
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


def _workspace_reservation(table, free_b):
    """(bytes the chunked shuffle's workspace will take out of `free_b`, whether at least one workgroup's pools fit): the rule of
    BatchedPSRS._shuffle_workspace -- pools for up to 1024 persistent workgroups within 92 % of what is free and never its last 2 GiB."""
    lib = L.load()
    one = int(lib.offsim_shuffle_workspace_bytes(C.byref(table.c), 1))
    if one <= 0 or max(table.max_seg, table.N0) <= 65536 or os.environ.get("OFFSIM_SHUFFLE_CHUNKED", "1") == "0":
        return 0, False
    budget = max(0, min(int(free_b * 0.92), free_b - (2 << 30)))
    full = int(lib.offsim_shuffle_workspace_bytes(C.byref(table.c), 1024))
    # A format-C table allocates the workspace FIRST (its format depends on it) and gets what the rule gives; any other table allocates
    # its stream buffers first and the workspace adapts to what is left (fewer workgroups' pools): a tenth of the free memory is kept
    # for it, as before round 6 (the full 1024-workgroup reservation cost C3's 2.83 M-row state half its resident rollouts).
    return (min(full, budget) if stream_format(table) == L.STREAMS_C else min(full, free_b // 10)), one <= budget


def rollout_resident_bytes(table, keyed=True, fmt=None):
    """HBM one rollout keeps resident between reset_sampler and the scan: its queue orders (candidate streams: 4 + 2 bytes per queue
    position, 4 + 1 in format C; as permutations: 4), its init order, cursors and random-stream state.  `fmt`: the stream format the
    environment will settle on (BatchedPSRS._stream_format: a format-C table falls back to B, 6 bytes, when the chunked shuffle's
    workspace does not fit or the init queue is beyond 2^23 rows); default: decided here by the same rule on the free memory as it is."""
    if fmt is None:
        fmt = stream_format(table)
        if keyed and fmt == L.STREAMS_C and table.device.type == "cuda":
            fits = _workspace_reservation(table, torch.cuda.mem_get_info(table.device)[0])[1]
            if table.N0 > (1 << 23) or not fits:
                fmt = L.STREAMS_B
    per_pos = 4 if not keyed else 5 if fmt == L.STREAMS_C else 6
    return int(table.N) * per_pos + int(table.N0) * 4 + int(table.n_slots) * 4 + 64


def resident_rollouts(table, keyed=True, free_bytes=None):
    """How many rollouts' queue orders the free HBM of the table's device holds at once (shared by bench.py and evalmc_rollouts):
    rollout_resident_bytes each -- in the stream format the environment will really use -- after 2 GiB for everything else of the job
    (policy keys, outputs, rebuilt permutations) and the chunked shuffle's workspace, reserved IN FULL by the rule that allocates it
    (_workspace_reservation).  Order of the real allocations: a format-C table decides its format through `_shuffle_workspace`, i.e.
    the workspace comes FIRST and the stream buffers after it; other tables allocate the stream buffers first and the workspace at
    the first reset.  Either way both are accounted for here.  Returns (rollouts, bytes per rollout, free, total)."""
    free_b, total_b = torch.cuda.mem_get_info(table.device)
    if free_bytes is not None:
        free_b = int(free_bytes)
    ws_b, fits = _workspace_reservation(table, free_b)
    fmt = stream_format(table)
    if fmt == L.STREAMS_C and (table.N0 > (1 << 23) or not fits):
        fmt = L.STREAMS_B
    per = max(rollout_resident_bytes(table, keyed=keyed, fmt=fmt), 1)
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
