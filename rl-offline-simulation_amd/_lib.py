"""ctypes binding of the C ABI in include/offsim.h (csrc/liboffsim_hip.so).

There is no CPU fallback: if the shared library is missing or no HIP device is visible, the product
path raises.  torch is used only for device memory, streams and torch.distributed.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# OFFSIM_LIB: another build of the same sources (A/B timing of kernel variants, the -DOFFSIM_ROWS_PROF / -DSHUF_FAULT_INJECT builds)
LIB_PATH = os.environ.get("OFFSIM_LIB") or os.path.join(_HERE, "csrc", "liboffsim_hip.so")

OK = 0
EINVAL, EHIP, EUNSUPPORTED = -1, -2, -3
F32, F64, F16 = 0, 1, 2
REJECT_DEFAULT, REJECT_NEVER = 0, 1
STREAM_PCG64, STREAM_PHILOX = 0, 1
STREAMS_A, STREAMS_B, STREAMS_C = 0, 1, 2
PROB_F64, PROB_F32 = 0, 1
ST_OK, ST_EXHAUSTED, ST_NO_INIT, ST_KEYERROR, ST_INACTIVE, ST_PROTOCOL = 0, 1, 2, 3, 4, 5

_vp, _i32, _i64, _u8 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint8


class Table(C.Structure):
    """struct offsim_table"""
    _fields_ = [("N", _i64), ("n_slots", _i32), ("nA", _i32), ("plog_dtype", _i32), ("r_dtype", _i32),
                ("seg_off", _vp), ("p_log", _vp), ("a", _vp), ("r", _vp), ("z_next", _vp), ("done", _vp),
                ("orig_idx", _vp), ("N0", _i64), ("init_slot", _vp), ("init_orig", _vp), ("max_seg", _i64), ("min_seg", _i64)]


class Rollouts(C.Structure):
    """struct offsim_rollouts"""
    _fields_ = [("R", _i32), ("rng", _vp), ("cursor", _vp), ("init_cursor", _vp), ("cur_slot", _vp),
                ("perm", _vp), ("perm_stride", _i64), ("init_perm", _vp), ("init_stride", _i64), ("rng_kind", _i32)]


class EvalMCOut(C.Structure):
    """struct offsim_evalmc_out"""
    _fields_ = [("sum_g", _vp), ("n_ep", _vp), ("steps", _vp), ("cand", _vp), ("n_len", _vp), ("status", _vp),
                ("ep_g", _vp), ("ep_len", _vp), ("ep_cap", _i64), ("trace_row", _vp), ("trace_pop", _vp),
                ("trace_cap", _i64), ("dbg", _vp)]


class Streams(C.Structure):
    """struct offsim_streams"""
    _fields_ = [("dig", _vp), ("dig_stride", _i64), ("loc", _vp), ("loc_stride", _i64), ("format", _i32)]


class Column(C.Structure):
    """struct offsim_column"""
    _fields_ = [("src", _vp), ("dst", _vp), ("row_bytes", _i64), ("zero_if_not_ok", _i32), ("reserved", _i32)]


class TD(C.Structure):
    """struct offsim_td"""
    _fields_ = [("mode", _i32), ("alpha", C.c_double), ("q", _vp), ("td_err", _vp), ("td_cap", _i64), ("behaviour", _i32), ("epsilon", C.c_double),
                ("alpha_ep", _vp), ("epsilon_ep", _vp), ("n_sched", _i64), ("q_snap", _vp), ("snap_cap", _i64), ("snap_stride", _i64),
                ("tie_mt", _vp), ("beh_arg", _vp)]


MAILBOX_MAX_ACTIONS = 24
SERVER_CMD_STEP, SERVER_CMD_POP_ONE, SERVER_CMD_EXIT, SERVER_CMD_RESET = 1, 2, 3, 4
SERVER_STARTING, SERVER_RUNNING, SERVER_EXITED = 1, 2, 3
SERVER_GONE = 1  # offsim_step_server_call: the server ended before it saw the request
SERVER_ANSWER_SECONDS = 10.0  # include/offsim.h: OFFSIM_SERVER_ANSWER_SECONDS


class StepMailbox(C.Structure):
    """struct offsim_step_mailbox (host-coherent pinned memory shared with the resident step server)"""
    _fields_ = [("seq_in", C.c_uint32), ("cmd", C.c_uint32), ("reject_mode", _i32), ("reserved0", C.c_uint32), ("p_head", C.c_double * 5),
                ("reserved1", C.c_uint32), ("seq_in2", C.c_uint32), ("p_tail", C.c_double * (MAILBOX_MAX_ACTIONS - 5)),
                ("reserved2", C.c_uint32 * 2), ("seq_out", C.c_uint32), ("row", _i32), ("status", _i32), ("popped", C.c_uint32),
                ("state", C.c_uint32), ("reserved3", C.c_uint32 * 3)]


TD_QLEARN, TD_EXPSARSA = 1, 2
BEHAVIOUR_FIXED, BEHAVIOUR_EPS_GREEDY, BEHAVIOUR_SOFT_GREEDY = 0, 1, 2

# name -> (restype, argtypes): exactly the entry points include/offsim.h declares
SIGNATURES = {
    "offsim_last_error": (C.c_char_p, []),
    "offsim_version": (C.c_int, []),
    "offsim_device_count": (C.c_int, []),
    "offsim_group_scratch_bytes": (_i64, [_i64, _i32]),
    "offsim_group_by_state": (C.c_int, [_vp, _i64, _i32, _vp, _vp, _vp, _vp]),
    "offsim_gather_rows": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp]),
    "offsim_seed_streams": (C.c_int, [_vp, _i32, _vp, _vp]),
    "offsim_shuffle_queues": (C.c_int, [C.POINTER(Table), _vp, _i32, _vp, _vp, _vp]),
    "offsim_env_reset": (C.c_int, [C.POINTER(Table), C.POINTER(Rollouts), _vp, _vp, _vp]),
    "offsim_env_set_state": (C.c_int, [C.POINTER(Rollouts), _vp, _vp, _vp]),
    "offsim_vector_gather": (C.c_int, [_vp, _vp, _vp, C.c_int32, _vp, C.c_int32, _vp, _vp]),
    "offsim_vector_step": (C.c_int, [C.POINTER(Table), C.POINTER(Rollouts), _vp, _i32, _i32, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp]),
    "offsim_step_batch": (C.c_int, [C.POINTER(Table), C.POINTER(Rollouts), _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "offsim_eval_mc": (C.c_int, [C.POINTER(Table), C.POINTER(Rollouts), _vp, _i32, _i32, C.c_double, _vp, _i64, _i64,
                                 C.POINTER(EvalMCOut), _vp]),
    "offsim_step_exo": (C.c_int, [C.POINTER(Table), C.POINTER(Table), C.POINTER(Rollouts), C.POINTER(Rollouts), _vp, _i32,
                                  _vp, _vp, _vp, _vp, _vp]),
    "offsim_eval_td": (C.c_int, [C.POINTER(Table), C.POINTER(Rollouts), _vp, _i32, C.c_double, _vp, _i64, _i64,
                                 C.POINTER(EvalMCOut), C.POINTER(TD), _vp]),
    "offsim_compile_policy": (C.c_int, [C.POINTER(Table), _vp, _vp, _vp]),
    "offsim_eval_mc_keys_kernel": (C.c_char_p, [_i32, _i32]),
    "offsim_compile_digests": (C.c_int, [C.POINTER(Table), _vp, _i32, _vp, _vp]),
    "offsim_shuffle_queues_keys": (C.c_int, [C.POINTER(Table), _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp]),
    "offsim_shuffle_queues_keys_ws": (C.c_int, [C.POINTER(Table), _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _i64, _vp]),
    "offsim_shuffle_workspace_bytes": (_i64, [C.POINTER(Table), _i32]),
    "offsim_shuffle_queues_ws": (C.c_int, [C.POINTER(Table), _vp, _i32, _vp, _vp, _vp, _i64, _vp]),
    "offsim_eval_mc_streams": (C.c_int, [C.POINTER(Table), C.POINTER(Rollouts), C.POINTER(Streams), _vp, C.c_double, _vp, _i64, _i64,
                                         C.POINTER(EvalMCOut), _vp]),
    "offsim_eval_mc_keys": (C.c_int, [C.POINTER(Table), C.POINTER(Rollouts), _vp, C.c_double, _vp, _i64, _i64,
                                      C.POINTER(EvalMCOut), _vp]),
    "offsim_selftest_lds_atomic_order": (C.c_int, [_vp, _vp]),
    "offsim_lds_order_ok": (C.c_int, []),
    "offsim_host_alloc": (C.c_int, [_i64, C.POINTER(_vp)]),
    "offsim_host_free": (C.c_int, [_vp]),
    "offsim_step_server_start": (C.c_int, [C.POINTER(Table), C.POINTER(Rollouts), _vp, _i32, C.c_uint32, _vp]),
    "offsim_step_server_call": (C.c_int, [_vp, _vp, _i32, _i32, C.c_uint32, _i32, C.c_uint64, _vp]),
    "offsim_async_faults": (C.c_int, []),
    "offsim_encode_box": (C.c_int, [_vp, _i64, _vp, _vp]),
    "offsim_encode_mlp": (C.c_int, [_vp, _i32, _i64, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp]),
}

_lib = None


class OffsimError(RuntimeError):
    pass


def load():
    """dlopen the HIP library and bind every symbol; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OffsimError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(or rl-offline-simulation_amd/csrc/build.sh).  There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(rc):
    if rc != OK:
        raise OffsimError(f"offsim error {rc}: {load().offsim_last_error().decode()}")


FAULT_SHUFFLE, FAULT_SCAN = 1, 2


def check_async_faults():
    """Raise if a kernel gave up a bounded inter-wavefront wait since the last check (include/offsim.h: offsim_async_faults).
    Call with the stream synchronised -- the host-facing drivers do, after they have copied their results back."""
    v = load().offsim_async_faults()
    if v < 0:
        check(v)
    if v:
        what = [n for b, n in ((FAULT_SHUFFLE, "sampler reset (shuffle ring protocol)"), (FAULT_SCAN, "scan (chain / helper hand-off)")) if v & b]
        raise OffsimError("a kernel gave up a bounded wait instead of hanging: " + ", ".join(what) + "; the results of that call are invalid")


_LDS_ORDER = {}


def lds_order_ok(device=None):
    """include/offsim.h: offsim_lds_order_ok for `device` (default: the current one) -- whether the LDS of this part serves the
    same-address lanes of one ds_add_rtn_u32 / ds_wrxchg_rtn_b32 in lane order, which the row-packed scan and the chunked shuffle rely
    on.  One short self-test per device and process, cached (here and in the library); a device without the property gets one warning
    and the host mirror routes it to the window kernel on permutations and the in-place shuffle (same results, slower)."""
    import torch
    idx = torch.cuda.current_device() if device is None else torch.device(device).index
    idx = torch.cuda.current_device() if idx is None else idx
    if idx not in _LDS_ORDER:
        with torch.cuda.device(idx):
            v = load().offsim_lds_order_ok()
        if v < 0:
            check(v)
        _LDS_ORDER[idx] = bool(v)
        if not v:
            import warnings
            warnings.warn(f"offsim: cuda:{idx} does not apply same-address LDS lanes in lane order (offsim_lds_order_ok = 0): the row-packed "
                          "scan and the chunked shuffle are off for this device -- window kernel on permutations, in-place shuffle",
                          RuntimeWarning, stacklevel=2)
    return _LDS_ORDER[idx]


def require_device():
    """Fail loudly when the HIP path cannot run (no GPU visible to torch)."""
    import torch
    if not torch.cuda.is_available():
        raise OffsimError("no HIP device visible: the PSRS engine runs only on the GPU (no CPU fallback)")
    dev = torch.device("cuda", torch.cuda.current_device())
    # the LDS lane-order guard runs its one-time self-test here, at table construction -- never inside a launch path, whose stream may
    # be capturing (include/offsim.h, offsim_lds_order_ok)
    lds_order_ok(dev)
    return dev


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    """device pointer of a contiguous torch tensor (None -> NULL)"""
    if t is None:
        return None
    assert t.is_contiguous(), "offsim needs contiguous tensors"
    return t.data_ptr()
