"""Round-5 GPU tests: the LDS lane-order property as a runtime guard (a device without it is routed to the window kernel and the
in-place shuffle, never to wrong numbers), world-size-8 plumbing of bench.py on one device, the extra bench workloads."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gpu():
    from rl_offline_simulation_amd import _lib
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    _lib.load()
    return torch.device("cuda", 0)


def test_lds_order_guard_says_yes_on_this_device(gpu):
    """offsim_lds_order_ok: the short per-device self-test behind offsim_eval_mc_streams and the chunked shuffle; 1 on gfx950, cached."""
    from rl_offline_simulation_amd import _lib as L
    assert L.load().offsim_lds_order_ok() == 1 and L.load().offsim_lds_order_ok() == 1
    assert L.lds_order_ok(gpu) is True


def test_lds_order_mismatch_routes_to_the_window_kernel_and_the_in_place_shuffle():
    """OFFSIM_FORCE_LDS_ORDER_MISMATCH=1 (child process: the verdict is cached per process) stands for a part whose LDS does not serve
    same-address lanes in lane order.  The C entry points that rely on it refuse (OFFSIM_EUNSUPPORTED) instead of computing; the host
    mirror warns once and takes the window kernel on permutations and the in-place shuffle -- with the oracle's results -- even where
    OFFSIM_SCAN_ROWS=1 asks for the row-packed kernel and a state has more than 65536 rows."""
    code = r'''
import ctypes as C, os, sys, warnings
import numpy as np, torch
sys.path.insert(0, %r)
from oracle import oracle as O
from rl_offline_simulation_amd import synth, _lib as L
from rl_offline_simulation_amd.table import TransitionTable
from rl_offline_simulation_amd.evaluators import BatchedPSRS
assert os.environ["OFFSIM_SCAN_ROWS"] == "1"
assert L.load().offsim_lds_order_ok() == 0
for N, nS in ((30000, 12), (90000, 1)):   # the second: one state of 90 k rows -- chunked shuffle / stream format C territory
    e = synth.synth_iid(N, nS, 2, seed=N)
    t0 = e["steps"] == 0
    pi = synth.dirichlet_policy(nS, 2)
    seeds = [3, 4, 5]
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        # (the guard's one self-test per device runs when the first table is built -- never inside a launch path: round 6)
        table = TransitionTable(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], t0)
        env = BatchedPSRS(table, len(seeds))
        env.reset_sampler(seeds, policy=table.policy_slots(pi))
        o = env.eval_mc(table.policy_slots(pi), 0.98, ep_cap=table.N0 + 1)
        torch.cuda.synchronize()
    if N == 30000:
        assert any("lane order" in str(x.message) for x in w), [str(x.message) for x in w]
    assert env._streams is None and env.state.perm is not None and getattr(env, "_ws", None) is None
    assert env.scan_variant().startswith("k_eval_mc_win"), env.scan_variant()
    L.check_async_faults()
    ora = O.OraclePSRS(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], t0)
    for i, sd in enumerate(seeds):
        ora.reset_sampler(sd)
        ref = ora.evalmc(10 ** 9, pi, 0.98)
        ne = int(o["n_ep"][i])
        assert int(o["steps"][i]) == ref["steps"] and int(o["cand"][i]) == ref["candidates"] and ne == len(ref["Gs"])
        assert np.array_equal(o["ep_g"][i, :ne].cpu().numpy(), ref["Gs"])
# the C ABI itself refuses: streams built by hand from the permutations, then offsim_eval_mc_streams
keys, dig32 = env._policy_keys(table.policy_slots(pi))
p = env.perm[:, :table.N].to(torch.int64) & 0xFFFFFFFF
env._fmt = L.STREAMS_B
dg, lc = env._pack_streams(dig32, p)
smc = L.Streams(dig=L.ptr(dg), dig_stride=table.N, loc=L.ptr(lc), loc_stride=table.N, format=L.STREAMS_B)
out = {k: torch.empty(3, dtype=dt, device="cuda") for k, dt in (("sum_g", torch.float64), ("n_ep", torch.int64), ("steps", torch.int64),
                                                                  ("cand", torch.int64), ("n_len", torch.int64), ("status", torch.int32))}
oc = L.EvalMCOut(sum_g=L.ptr(out["sum_g"]), n_ep=L.ptr(out["n_ep"]), steps=L.ptr(out["steps"]), cand=L.ptr(out["cand"]), n_len=L.ptr(out["n_len"]),
                 status=L.ptr(out["status"]))
from rl_offline_simulation_amd.evaluators.psrs import _gamma_pow
gp = _gamma_pow(0.98, 4096, "cuda", cap=table.N + 2)
env.state.rewind()
rc = L.load().offsim_eval_mc_streams(C.byref(table.c), C.byref(env.state.c), C.byref(smc), L.ptr(keys), 0.98, L.ptr(gp), gp.numel(), 1 << 62,
                                     C.byref(oc), L.stream_ptr())
assert rc == L.EUNSUPPORTED and b"lane order" in L.load().offsim_last_error(), (rc, L.load().offsim_last_error())
ws = torch.empty(int(L.load().offsim_shuffle_workspace_bytes(C.byref(table.c), 4)), dtype=torch.uint8, device="cuda")
sd = torch.tensor(seeds, dtype=torch.int64, device="cuda")
lc8 = torch.empty((3, table.N), dtype=torch.uint8, device="cuda")
rc = L.load().offsim_shuffle_queues_keys_ws(C.byref(table.c), L.ptr(sd), 3, L.ptr(dig32), L.STREAMS_C, L.ptr(dg), L.ptr(lc8), L.ptr(env._init_perm_buf),
                                            L.ptr(ws), ws.numel(), L.stream_ptr())
assert rc == L.EUNSUPPORTED and b"lane order" in L.load().offsim_last_error(), (rc, L.load().offsim_last_error())
torch.cuda.synchronize()
print("ok")
''' % ROOT
    env = dict(os.environ, OFFSIM_FORCE_LDS_ORDER_MISMATCH="1", OFFSIM_SCAN_ROWS="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-3000:]


def _bench(args, env_extra=None, timeout=900):
    env = dict(os.environ, **(env_extra or {}))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.timeout(1200)
def test_bench_plumbing_at_world_size_8_on_one_device(gpu):
    """What the driver's 8-GPU run executes, with the eight ranks sharing this box's one GPU and gloo carrying the collective:
    `bench.py --gpus 8` starts its eight ranks itself (children of a process that never touched the GPU), the log is split into eight
    episode-disjoint shards with all 4096 seeds on each, the extra rollout-sharded measurement gives every rank 512 seeds of the whole
    log, rank 0 alone prints the line and checks four seeds of ITS shard against the oracle while the others wait in the barrier.
    No scaling number is expected from this -- the plumbing is: shard arithmetic, tile sizing, the [R,2] all-reduce at world 8."""
    out = _bench(["--gpus", "8", "--all-ranks-on-device0", "--dist-backend", "gloo", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
                 env_extra={"OFFSIM_BENCH_TEST_SCALE": "200"}, timeout=1200)
    assert out["n_gpus"] == 8 and out["scaling"] == "strong" and out["test_scale"] == 200
    c = out["collective"]
    assert c["world"] == 8 and c["backend"] == "gloo" and c["bytes"] == 4096 * 2 * 8 and c["allreduce_us"] > 0
    assert out["parity_check"]["ok"] and out["parity_check"]["table_rows"] == out["config"]["transitions_on_rank0"]
    assert 0 < out["config"]["transitions_on_rank0"] < out["config"]["transitions"] == 50_000
    rs = out["rollout_sharded"]
    assert rs["rollouts_per_gpu"] == 512 and rs["transitions_per_gpu"] == 50_000 and rs["value"] > 0
    # every seed's estimate went through the all-reduce: the mean over seeds of sum(G)/n is finite, and the steps of all eight shards are in
    assert np.isfinite(out["value_estimate_mean"]) and out["value"] > 0


@pytest.mark.timeout(900)
def test_bench_obs128_workload_alone(gpu):
    """`bench.py --workload obs128` (config C5's shape as its own command): device-generated 128-d fp16 observations, the encoder
    forward on MFMA, an fp16 p_log table, reset + scan with its oracle parity check and the encoder's own roofline object."""
    out = _bench(["--workload", "obs128", "--transitions", "200000", "--rollouts", "256", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"])
    assert out["parity_check"]["ok"] and out["config"]["p_log"] == "float16"
    assert out["roofline"]["bytes_per_candidate"] == 16 and out["encoder"]["argmax_equals_oracle_mlp"] is True


@pytest.mark.timeout(900)
def test_bench_under_torch_distributed_run(gpu):
    """The driver's multi-GPU launch line -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` -- with N = 2 ranks sharing this box's one GPU (gloo): bench.py is then ONE OF the ranks (RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment, no ranks of its own), rank 0 alone prints the line, `--gpus` must equal
    WORLD_SIZE.  (The launcher is a child process of a process that never initialised the GPU.)"""
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    env = dict(os.environ, OFFSIM_BENCH_TEST_SCALE="200")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--all-ranks-on-device0", "--dist-backend", "gloo"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["collective"]["world"] == 2 and out["parity_check"]["ok"] and out["rollout_sharded"]["rollouts_per_gpu"] == 2048
    # --gpus that disagrees with WORLD_SIZE is refused
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True, timeout=120,
                         env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)))
    assert bad.returncode == 2 and "WORLD_SIZE" in bad.stderr
