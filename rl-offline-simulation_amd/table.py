"""Device-resident logged-transition table and rollout state (the `offsim4rl/data` rewrite).

Layout: structure-of-arrays in HBM, rows physically grouped by from-state so that
each state's queue is one contiguous CSR segment.  The hot candidate stream (p_log[nA], a) is
separate from the accept-only stream (r, z_next, done); `orig_idx` maps a grouped row back to the
caller's buffer row for accepted-index reports.

Replaces PSRS._calculate_latent_state and the sorted()/groupby() of PSRS.reset_sampler
(offsim4rl/evaluators/psrs.py:16-17,26) and the tuple assembly of
PerStateRejectionSampling.__init__ (offsim4rl/evaluators/per_state_rejection.py:38-50).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L

_TORCH_OF = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64, np.dtype(np.float16): torch.float16}
_TAG_OF = {torch.float32: L.F32, torch.float64: L.F64, torch.float16: L.F16}


def _dev(x, device, dtype=None):
    if isinstance(x, torch.Tensor):
        t = x.to(device)
    else:
        t = torch.from_numpy(np.ascontiguousarray(x)).to(device)
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


def group_by_state(slot, n_slots):
    """Stable group-by on the device: returns (seg_off[n_slots+1] uint32-as-int32 tensor, order[N] int32)."""
    lib = L.load()
    N = slot.numel()
    dev = slot.device
    seg_off = torch.empty(n_slots + 1, dtype=torch.int32, device=dev)
    order = torch.empty(max(N, 1), dtype=torch.int32, device=dev)[:N]
    scratch = torch.empty(int(lib.offsim_group_scratch_bytes(N, n_slots)), dtype=torch.uint8, device=dev)
    L.check(lib.offsim_group_by_state(L.ptr(slot), N, n_slots, L.ptr(seg_off), L.ptr(order) if N else None,
                                      L.ptr(scratch), L.stream_ptr()))
    return seg_off, order


def gather_rows(src, order):
    """dst[g] = src[order[g]] (rows of any width) on the device."""
    lib = L.load()
    n = order.numel()
    dst = torch.empty((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    if n:
        row_bytes = src.element_size() * int(np.prod(src.shape[1:], dtype=np.int64))
        L.check(lib.offsim_gather_rows(L.ptr(src), L.ptr(order), n, row_bytes, L.ptr(dst), L.stream_ptr()))
    return dst


class TransitionTable:
    """SoA table of N logged transitions, grouped by latent from-state, resident in HBM."""

    def __init__(self, z, a, r, z_next, done, p_log, t0=None, device=None, plog_dtype=None):
        L.load()
        device = device or L.require_device()
        self.device = device
        if torch.device(device).type == "cuda":
            L.lds_order_ok(device)  # (the one-time LDS lane-order self-test of this device runs here, outside any launch path or capture)
        z = _dev(z, device, torch.int64)
        z_next = _dev(z_next, device, torch.int64)
        N = int(z.numel())
        self.N = N
        p_log = _dev(p_log, device)
        if plog_dtype is not None:
            p_log = p_log.to(_TORCH_OF[np.dtype(plog_dtype)])
        if p_log.dtype not in _TAG_OF:
            p_log = p_log.to(torch.float64)
        p_log = (p_log.reshape(N, -1) if N else p_log.reshape(0, p_log.shape[-1] if p_log.dim() > 1 else 1)).contiguous()
        self.nA = int(p_log.shape[1])
        r = _dev(r, device)
        if r.dtype not in (torch.float32, torch.float64):
            r = r.to(torch.float64)
        # states -> non-negative slots; z = -1 is a legal key (heuristic.py:23-24).  Dense ids map to slot = z - z_base; sparse
        # or hashed ids (the reference keys a dict by z, psrs.py:26, so any integers work there) are compacted to their rank
        # among the distinct ids, so that the per-rollout cursor arrays and the LDS carve-outs stay proportional to the
        # number of states and not to the id range.  slot_z[slot] is the id either way.
        if N:
            lo = int(min(z.min().item(), z_next.min().item(), 0))
            hi = int(max(z.max().item(), z_next.max().item()))
            uniq = torch.unique(torch.cat([z, z_next]))
        else:
            lo, hi, uniq = 0, 0, z
        if N and (hi - lo + 1) > max(1024, 4 * int(uniq.numel())):
            self.z_base = None
            self.slot_z = uniq.cpu().numpy().astype(np.int64)
            self.n_slots = int(uniq.numel())
            slot = torch.searchsorted(uniq, z).to(torch.int32).contiguous()
            slot_next = torch.searchsorted(uniq, z_next).to(torch.int32).contiguous()
        else:
            self.z_base = lo
            self.n_slots = hi - lo + 1
            self.slot_z = np.arange(lo, hi + 1, dtype=np.int64)
            slot = (z - lo).to(torch.int32).contiguous()
            slot_next = (z_next - lo).to(torch.int32).contiguous()
        # actions index p_log[row] and p_new on the device: NumPy would raise IndexError (or wrap a negative index) where
        # the kernels would read out of bounds, so the range is checked here (psrs.py:54-57)
        a_dev = _dev(a, device, torch.int64)
        if N and (int(a_dev.min().item()) < -self.nA or int(a_dev.max().item()) >= self.nA):
            raise IndexError(f"logged action outside [-{self.nA}, {self.nA}) for action_distributions with {self.nA} columns")
        a = torch.where(a_dev < 0, a_dev + self.nA, a_dev).to(torch.int32)  # NumPy's negative-index wrap
        self.seg_off, self.order = group_by_state(slot, self.n_slots)
        self.p_log = gather_rows(p_log, self.order)
        self.a = gather_rows(_dev(a, device, torch.int32), self.order)
        self.r = gather_rows(r, self.order)
        self.z_next = gather_rows(slot_next, self.order)
        self.done = gather_rows(_dev(np.asarray(done) != 0 if not isinstance(done, torch.Tensor) else done != 0, device,
                                     torch.uint8), self.order)
        # initial rows in buffer order (psrs.py:22); no `steps` => every row (data.py:72)
        if t0 is None:
            init_rows = torch.arange(N, dtype=torch.int32, device=device)
        else:
            key = (_dev(np.asarray(t0) != 0 if not isinstance(t0, torch.Tensor) else t0 != 0, device, torch.uint8) == 0).to(torch.int32)
            off2, order2 = group_by_state(key.contiguous(), 2)
            n0 = int(off2[1].item())
            init_rows = order2[:n0].contiguous()
        self.N0 = int(init_rows.numel())
        self.init_orig = init_rows
        self.init_slot = gather_rows(slot, init_rows)
        self._slot = slot
        so = self.seg_off.to(torch.int64) & 0xFFFFFFFF
        lens = so[1:] - so[:-1]
        self.max_seg = int(lens.max().item()) if N else 0
        self.min_seg = int(lens[lens > 0].min().item()) if N else 0
        # rows of the chains that do not fit the shuffle's LDS (states and the init queue): sizes the chunked shuffle's workspace
        self.long_rows = (int(lens[lens > 65536].sum().item()) if N else 0) + (self.N0 if self.N0 > 65536 else 0)
        self.c = L.Table(N=N, n_slots=self.n_slots, nA=self.nA, plog_dtype=_TAG_OF[self.p_log.dtype],
                         r_dtype=_TAG_OF[self.r.dtype], seg_off=L.ptr(self.seg_off), p_log=L.ptr(self.p_log),
                         a=L.ptr(self.a), r=L.ptr(self.r), z_next=L.ptr(self.z_next), done=L.ptr(self.done),
                         orig_idx=L.ptr(self.order), N0=self.N0, init_slot=L.ptr(self.init_slot),
                         init_orig=L.ptr(self.init_orig), max_seg=self.max_seg, min_seg=self.min_seg)

    # ---- bookkeeping used by the measurement code ----
    @property
    def bytes_per_candidate(self):
        """B_c = nA*sizeof(p_log) + sizeof(a)"""
        return self.nA * self.p_log.element_size() + 4

    @property
    def bytes_per_step(self):
        """B_s = sizeof(r) + sizeof(z_next) + sizeof(done)"""
        return self.r.element_size() + 4 + 1

    def slot_of(self, z):
        """Slot of latent state z; -1 if z occurs nowhere in the log (its queue is missing: KeyError at the next step, psrs.py:44)."""
        z = int(z)
        if self.z_base is not None:
            s = z - self.z_base
            return s if 0 <= s < self.n_slots else -1
        i = int(np.searchsorted(self.slot_z, z))
        return i if i < self.n_slots and int(self.slot_z[i]) == z else -1

    def z_of(self, slot):
        return int(self.slot_z[int(slot)])

    def segment_lengths(self):
        so = self.seg_off.to(torch.int64).cpu().numpy() & 0xFFFFFFFF
        return np.diff(so)

    def policy_slots(self, pi):
        """Rows of a tabular policy indexed as the reference does, pi[S] with NumPy semantics
        (S = -1 selects the last row; psrs.py:255): returns pi_slots[n_slots, nA]."""
        pi = np.asarray(pi)
        zs = self.slot_z
        idx = np.where(zs < 0, zs + pi.shape[0], zs)
        ok = (idx >= 0) & (idx < pi.shape[0])
        out = np.zeros((self.n_slots, pi.shape[1]), pi.dtype)
        out[ok] = pi[idx[ok]]
        # states without a policy row can only be reached through a missing queue; flag them with NaN so that
        # nothing is silently accepted there (the reference would raise IndexError)
        out[~ok] = np.nan
        return out


class RolloutState:
    """State of R simulated rollouts (struct offsim_rollouts): RNG streams, queue cursors, current states."""

    def __init__(self, table, R):
        dev = table.device
        self.table, self.R = table, int(R)
        self.rng = torch.zeros((R, 4), dtype=torch.int64, device=dev)
        self.cursor = torch.zeros((R, table.n_slots), dtype=torch.int32, device=dev)
        self.init_cursor = torch.zeros(R, dtype=torch.int32, device=dev)
        self.cur_slot = torch.full((R,), -1, dtype=torch.int32, device=dev)
        self.perm = None
        self.perm_stride = 0
        self.init_perm = None
        self.init_stride = 0
        self.rng_kind = L.STREAM_PCG64
        self._refresh()

    def _refresh(self):
        self.c = L.Rollouts(R=self.R, rng=L.ptr(self.rng), cursor=L.ptr(self.cursor), init_cursor=L.ptr(self.init_cursor),
                            cur_slot=L.ptr(self.cur_slot), perm=L.ptr(self.perm), perm_stride=self.perm_stride,
                            init_perm=L.ptr(self.init_perm), init_stride=self.init_stride, rng_kind=self.rng_kind)

    def set_orders(self, perm, perm_stride, init_perm, init_stride):
        self.perm, self.perm_stride, self.init_perm, self.init_stride = perm, int(perm_stride), init_perm, int(init_stride)
        self._refresh()

    def rewind(self):
        self.cursor.zero_()
        self.init_cursor.zero_()
        self.cur_slot.fill_(-1)


def seeds_tensor(seeds, device):
    s = np.asarray(seeds, dtype=np.uint64).reshape(-1)
    return torch.from_numpy(s.view(np.int64).copy()).to(device)


def seed_streams(seeds_dev, out=None):
    """default_rng(seed) for every seed: [R,4] PCG64 words on the device (psrs.py:20)."""
    R = seeds_dev.numel()
    if out is None:
        out = torch.empty((R, 4), dtype=torch.int64, device=seeds_dev.device)
    L.check(L.load().offsim_seed_streams(L.ptr(seeds_dev), R, L.ptr(out), L.stream_ptr()))
    return out


def shuffle_queues(table, seeds_dev, perm=None, init_perm=None, workspace=None):
    """PSRS.reset_sampler's shuffles for every seed (psrs.py:22-23,29-30): perm [n,N], init_perm [n,N0] (uint32).  `workspace` (a
    uint8 device tensor of offsim_shuffle_workspace_bytes): chains of more than 65536 rows run on the chunked kernel."""
    n = seeds_dev.numel()
    dev = table.device
    if perm is None:
        perm = torch.empty((n, max(table.N, 1)), dtype=torch.int32, device=dev)
    if init_perm is None:
        init_perm = torch.empty((n, max(table.N0, 1)), dtype=torch.int32, device=dev)
    L.check(L.load().offsim_shuffle_queues_ws(C.byref(table.c), L.ptr(seeds_dev), n, L.ptr(perm), L.ptr(init_perm), L.ptr(workspace),
                                              0 if workspace is None else workspace.numel(), L.stream_ptr()))
    return perm, init_perm
