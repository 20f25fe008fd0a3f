"""PSRS_Exo (offsim4rl/evaluators/psrs.py:59-117): PSRS with an exogenous state component.

An observation o splits into (s, x).  Transitions are queued twice: by s (the part the agent controls: action, reward,
next s, done, logging probabilities) and by x (the exogenous part: next x).  Every candidate pops the head of both
queues; the rejection test uses the s-element; the next observation recombines the accepted s' with the popped x'.
Both queue families reuse the ordinary device tables, shuffles and resets; the step is offsim_step_exo.
"""
import ctypes as C
import os

import numpy as np
import torch

from .. import _lib as L
from ..table import RolloutState, TransitionTable, seed_streams, seeds_tensor, shuffle_queues


class PSRS_Exo:
    def __init__(self, buffer, nO=25, nA=5, o_split_func=lambda o: (o, 0), o_combine_func=lambda s, x: s):
        rows = list(buffer)  # (o, a, r, o', done, p, info)
        self.raw_buffer = rows
        self.nS = self.nO = nO
        self.nA = nA
        self.o_split_func, self.o_combine_func = o_split_func, o_combine_func
        n = len(rows)
        sx = [o_split_func(r[0]) for r in rows]
        sx_ = [o_split_func(r[3]) for r in rows]
        s = np.fromiter((int(v[0]) for v in sx), np.int64, n)
        x = np.fromiter((int(v[1]) for v in sx), np.int64, n)
        s_ = np.fromiter((int(v[0]) for v in sx_), np.int64, n)
        x_ = np.fromiter((int(v[1]) for v in sx_), np.int64, n)
        a = np.fromiter((int(r[1]) for r in rows), np.int64, n)
        rew = np.array([r[2] for r in rows], dtype=np.float64) if n else np.zeros(0)
        done = np.fromiter((bool(r[4]) for r in rows), bool, n)
        p_log = np.stack([np.asarray(r[5]) for r in rows]) if n else np.zeros((0, nA))
        t0 = np.fromiter((r[6]["t"] == 0 for r in rows), bool, n)
        self._s, self._x, self._s_next, self._x_next = s, x, s_, x_
        self.ts = TransitionTable(s, a, rew, s_, done, p_log, t0)
        self.tx = TransitionTable(x, np.zeros(n, np.int64), np.zeros(n), x_, np.zeros(n, bool), np.ones((n, 1), np.float32), t0)
        self.rs, self.rx = RolloutState(self.ts, 1), RolloutState(self.tx, 1)
        dev = self.ts.device
        self._o = torch.empty(4, dtype=torch.int32, device=dev)
        self._row = torch.empty(1, dtype=torch.int32, device=dev)
        self.o = None
        self.s = None
        self.reset_sampler()
        self.reset()

    def reset_sampler(self, seed=None):
        """psrs.py:77-89: init queue and every s- and x-queue shuffled by a fresh default_rng(seed)."""
        if seed is None:
            seed = int.from_bytes(os.urandom(8), "little")
        sd = seeds_tensor([seed], self.ts.device)
        for t, ro in ((self.ts, self.rs), (self.tx, self.rx)):
            perm, init_perm = shuffle_queues(t, sd)
            ro.rewind()
            ro.set_orders(perm, t.N, init_perm, t.N0)

    def reset(self, seed=None):
        """psrs.py:91-97: the rejection stream restarts from `seed` at EVERY reset."""
        if seed is None:
            seed = int.from_bytes(os.urandom(8), "little")
        seed_streams(seeds_tensor([seed], self.ts.device), self.rs.rng)
        lib = L.load()
        L.check(lib.offsim_env_reset(C.byref(self.ts.c), C.byref(self.rs.c), None, L.ptr(self._row), L.stream_ptr()))
        L.check(lib.offsim_env_reset(C.byref(self.tx.c), C.byref(self.rx.c), None, None, L.stream_ptr()))
        row = int(self._row.cpu()[0])
        if row < 0:
            self.s = None
            return None
        self.o = self.raw_buffer[row][0]
        return self.o

    def step(self, p_new):
        """psrs.py:99-117"""
        if isinstance(p_new, torch.Tensor):
            p_new = p_new.detach().cpu().numpy()
        p_new = np.asarray(p_new)
        f32 = p_new.dtype == np.float32 and self.ts.p_log.dtype == torch.float32
        p = torch.from_numpy(np.ascontiguousarray(p_new.reshape(1, -1))).to(self.ts.device, torch.float32 if f32 else torch.float64)
        s_cur, x_cur = self.o_split_func(self.o)
        base = self._o.data_ptr()
        L.check(L.load().offsim_step_exo(C.byref(self.ts.c), C.byref(self.tx.c), C.byref(self.rs.c), C.byref(self.rx.c), L.ptr(p),
                                         L.PROB_F32 if f32 else L.PROB_F64, base, base + 4, base + 8, base + 12, L.stream_ptr()))
        row_s, row_x, status, _ = self._o.cpu().tolist()
        if status == L.ST_KEYERROR:
            raise KeyError((s_cur, x_cur))
        if status != L.ST_OK:
            return None, None, None, None
        rs_ = self.raw_buffer[row_s]
        self.o = self.o_combine_func(int(self._s_next[row_s]), int(self._x_next[row_x]))
        return self.o, rs_[2], bool(rs_[4]), {"s": s_cur, "a": rs_[1], "p": rs_[5]}
