"""ctypes front-end of the CPU oracle (oracle/psrs_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by the product package.  Each method names the reference
function it restates (paths relative to /root/reference/).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpsrs_oracle.so")

REJECT_DEFAULT, REJECT_NEVER = 0, 1
PROB_F64, PROB_F32 = 0, 1


def build(force=False):
    src = os.path.join(_HERE, "psrs_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpsrs_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        p64, pd, pu8 = C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_uint8)
        L.oracle_psrs_new.restype = C.c_void_p
        L.oracle_psrs_new.argtypes = [C.c_int64, C.c_int64, p64, p64, pd, p64, pu8, pu8, pd]
        L.oracle_psrs_free.argtypes = [C.c_void_p]
        L.oracle_psrs_clone.restype = C.c_void_p
        L.oracle_psrs_clone.argtypes = [C.c_void_p]
        L.oracle_psrs_reset_sampler.argtypes = [C.c_void_p, C.c_uint64]
        L.oracle_psrs_set_rejection_seed.argtypes = [C.c_void_p, C.c_uint64]
        L.oracle_psrs_set_rejection_philox.argtypes = [C.c_void_p, C.c_uint64]
        L.oracle_philox_doubles.argtypes = [C.c_uint64, C.c_uint64, C.c_int64, pd]
        L.oracle_psrs_get_orders.argtypes = [C.c_void_p, p64, p64, p64, p64]
        L.oracle_psrs_get_heads.argtypes = [C.c_void_p, p64, p64]
        L.oracle_psrs_n_keys.restype = C.c_int64
        L.oracle_psrs_n_keys.argtypes = [C.c_void_p]
        L.oracle_psrs_n_init.restype = C.c_int64
        L.oracle_psrs_n_init.argtypes = [C.c_void_p]
        L.oracle_psrs_reset.restype = C.c_int64
        L.oracle_psrs_reset.argtypes = [C.c_void_p]
        L.oracle_psrs_cur_z.restype = C.c_int64
        L.oracle_psrs_cur_z.argtypes = [C.c_void_p]
        L.oracle_psrs_step.restype = C.c_int64
        L.oracle_psrs_step.argtypes = [C.c_void_p, pd, C.c_int, C.c_int, p64]
        L.oracle_evalmc.restype = C.c_int
        L.oracle_evalmc.argtypes = [C.c_void_p, C.c_int64, pd, C.c_int64, C.c_double, C.c_int, C.c_int,
                                    pd, p64, p64, p64, p64, p64, C.c_int64, p64, p64]
        L.oracle_rng_doubles.argtypes = [C.c_uint64, C.c_int64, pd]
        L.oracle_rng_raw64.argtypes = [C.c_uint64, C.c_int64, C.POINTER(C.c_uint64)]
        L.oracle_permutation.argtypes = [C.c_uint64, C.c_int64, p64]
        L.oracle_seedseq_words.argtypes = [C.c_uint64, C.POINTER(C.c_uint64)]
        L.oracle_cartpole_encode.argtypes = [C.POINTER(C.c_float), C.c_int64, p64]
        L.oracle_mlp_encode.argtypes = [C.POINTER(C.c_float), C.c_int64, C.c_int64, C.POINTER(C.c_float),
                                        C.POINTER(C.c_float), C.c_int64, C.POINTER(C.c_float),
                                        C.POINTER(C.c_float), C.c_int64, p64, C.POINTER(C.c_float)]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def rng_doubles(seed, n):
    """First n values of default_rng(seed).random()  (psrs.py:20,56)."""
    out = np.empty(n, np.float64)
    lib().oracle_rng_doubles(seed, n, _p(out, C.c_double))
    return out


def philox_doubles(seed, n, first=0):
    """Draws first .. first + n - 1 of the Philox4x32-10 rejection stream of `seed` (u in (0, 1])."""
    out = np.empty(n, dtype=np.float64)
    lib().oracle_philox_doubles(int(seed), int(first), n, _p(out, C.c_double))
    return out


def rng_raw64(seed, n):
    out = np.empty(n, np.uint64)
    lib().oracle_rng_raw64(seed, n, _p(out, C.c_uint64))
    return out


def seedseq_words(seed):
    out = np.empty(4, np.uint64)
    lib().oracle_seedseq_words(seed, _p(out, C.c_uint64))
    return out


def permutation(seed, n):
    """default_rng(seed).shuffle(list(range(n)))  (psrs.py:23,30)."""
    out = np.empty(n, np.int64)
    lib().oracle_permutation(seed, n, _p(out, C.c_int64))
    return out


def cartpole_encode(obs):
    """CartpoleBoxEncoder.encode  (offsim4rl/encoders/heuristic.py:65-71)."""
    obs = np.ascontiguousarray(obs, np.float32)
    out = np.empty(obs.shape[0], np.int64)
    lib().oracle_cartpole_encode(_p(obs, C.c_float), obs.shape[0], _p(out, C.c_int64))
    return out


def mlp_encode(x, W1, b1, W2, b2):
    """HOMEREncoder.encode  (offsim4rl/encoders/homer.py:159-168); returns (z, logits)."""
    x = np.ascontiguousarray(x, np.float32)
    W1, b1, W2, b2 = (np.ascontiguousarray(w, np.float32) for w in (W1, b1, W2, b2))
    N, dO = x.shape
    H, nZ = W1.shape[0], W2.shape[0]
    z = np.empty(N, np.int64)
    logits = np.empty((N, nZ), np.float32)
    f = C.c_float
    lib().oracle_mlp_encode(_p(x, f), N, dO, _p(W1, f), _p(b1, f), H, _p(W2, f), _p(b2, f), nZ,
                            _p(z, C.c_int64), _p(logits, f))
    return z, logits


class OraclePSRS:
    """PSRS  (offsim4rl/evaluators/psrs.py:5-57) over array inputs; rows are identified by index."""

    def __init__(self, z, a, r, z_next, done, p_log, t0=None):
        self.z = np.ascontiguousarray(z, np.int64)
        N = self.z.shape[0]
        self.a = np.ascontiguousarray(a, np.int64)
        self.r = np.ascontiguousarray(r, np.float64)
        self.z_next = np.ascontiguousarray(z_next, np.int64)
        self.done = np.ascontiguousarray(np.asarray(done) != 0, np.uint8)
        self.t0 = np.ones(N, np.uint8) if t0 is None else np.ascontiguousarray(np.asarray(t0) != 0, np.uint8)
        self.p_log = np.ascontiguousarray(p_log, np.float64).reshape(N, -1)
        self.N, self.nA = N, self.p_log.shape[1]
        self._h = lib().oracle_psrs_new(N, self.nA, _p(self.z, C.c_int64), _p(self.a, C.c_int64),
                                        _p(self.r, C.c_double), _p(self.z_next, C.c_int64),
                                        _p(self.done, C.c_uint8), _p(self.t0, C.c_uint8), _p(self.p_log, C.c_double))

    def clone(self):
        """A second sampler over the same buffer (shares the arrays and the grouping; own queues, cursors, streams)."""
        o = object.__new__(OraclePSRS)
        for k in ("z", "a", "r", "z_next", "done", "t0", "p_log", "N", "nA"):
            setattr(o, k, getattr(self, k))
        o._parent = self  # keeps the shared grouping alive
        o._h = lib().oracle_psrs_clone(self._h)
        return o

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:  # (module globals are gone at interpreter shutdown)
            lib().oracle_psrs_free(self._h)
            self._h = None

    def reset_sampler(self, seed):
        lib().oracle_psrs_reset_sampler(self._h, int(seed))

    def set_rejection_seed(self, seed):
        lib().oracle_psrs_set_rejection_seed(self._h, int(seed))

    def set_rejection_philox(self, seed):
        """env.rejection_sampling_rng = a replay of rocRAND's Philox4x32-10 stream of `seed` (include/offsim.h OFFSIM_STREAM_PHILOX)."""
        lib().oracle_psrs_set_rejection_philox(self._h, int(seed))

    def orders(self):
        """(keys, key_off, queue rows CSR by key, init rows) after reset_sampler."""
        nk, n0 = lib().oracle_psrs_n_keys(self._h), lib().oracle_psrs_n_init(self._h)
        keys, off = np.empty(nk, np.int64), np.empty(nk + 1, np.int64)
        q, iq = np.empty(self.N, np.int64), np.empty(n0, np.int64)
        lib().oracle_psrs_get_orders(self._h, _p(keys, C.c_int64), _p(off, C.c_int64), _p(q, C.c_int64), _p(iq, C.c_int64))
        return keys, off, q, iq

    def heads(self):
        nk = lib().oracle_psrs_n_keys(self._h)
        h, ih = np.empty(nk, np.int64), np.empty(1, np.int64)
        lib().oracle_psrs_get_heads(self._h, _p(h, C.c_int64), _p(ih, C.c_int64))
        return h, int(ih[0])

    def reset(self):
        """PSRS.reset  (psrs.py:32-37): initial row index or None."""
        row = lib().oracle_psrs_reset(self._h)
        return None if row < 0 else int(row)

    @property
    def cur_z(self):
        return int(lib().oracle_psrs_cur_z(self._h))

    def step(self, p_new, prob_dtype=PROB_F64, reject_mode=REJECT_DEFAULT):
        """PSRS.step  (psrs.py:39-51): (accepted row | None, candidates popped); KeyError like the reference."""
        p = np.ascontiguousarray(p_new, np.float64)
        n = C.c_int64(0)
        row = lib().oracle_psrs_step(self._h, _p(p, C.c_double), prob_dtype, reject_mode, C.byref(n))
        if row == -2:
            raise KeyError(self.cur_z)
        return (None if row < 0 else int(row)), int(n.value)

    def evalmc(self, n_episodes, pi, gamma, prob_dtype=PROB_F64, reject_mode=REJECT_DEFAULT, trace_cap=0):
        """evalMC_psrs  (psrs.py:241-271)."""
        pi = np.ascontiguousarray(pi, np.float64)
        cap = int(min(n_episodes, self.N + 1))
        Gs, lengths = np.empty(cap, np.float64), np.empty(cap + 1, np.int64)
        tr = np.empty(max(trace_cap, 1), np.int64)
        tp = np.empty(max(trace_cap, 1), np.int64)
        nG, nL, nS, nC = (C.c_int64(0) for _ in range(4))
        rc = lib().oracle_evalmc(self._h, cap, _p(pi, C.c_double), pi.shape[0], float(gamma), prob_dtype, reject_mode,
                                 _p(Gs, C.c_double), C.byref(nG), _p(lengths, C.c_int64), C.byref(nL),
                                 _p(tr, C.c_int64) if trace_cap else None, _p(tp, C.c_int64) if trace_cap else None,
                                 trace_cap, C.byref(nS), C.byref(nC))
        if rc == -2:
            raise KeyError(self.cur_z)
        if rc == -3:
            raise IndexError("pi index out of range")
        k = min(nS.value, trace_cap)
        return dict(Gs=Gs[:nG.value].copy(), lengths=lengths[:nL.value].copy(), steps=nS.value, candidates=nC.value,
                    trace_rows=tr[:k].copy(), trace_popped=tp[:k].copy())
