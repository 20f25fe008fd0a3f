"""Round-4 GPU tests: a second policy on the same sampler state (streams re-keyed in place), the RCCL leg on one rank, the fault
word reaching the drop-in drivers, the bench line's extra configurations."""
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


@pytest.mark.parametrize("nS,N,fmt", [(20, 40_000, "A"), (2, 180_000, "B"), (2, 180_000, "C")])  # states of more than 65536 rows: B, or the 5-byte C
def test_second_policy_on_the_same_sampler_state_rekeys_the_streams(nS, N, fmt, gpu, monkeypatch):
    """reset_sampler(policy=A) writes the queue orders as A's candidate streams; evaluating policy B afterwards on the same sampler
    state is what the reference allows (evalMC_psrs takes any pi, the queues just go on: psrs.py:241-271).  The streams' digests
    are replaced in place (BatchedPSRS._rekey_streams) and the row-packed kernel goes on -- against the oracle doing the same."""
    from oracle import oracle as O
    from rl_offline_simulation_amd import synth, _lib as L
    from rl_offline_simulation_amd.table import TransitionTable
    from rl_offline_simulation_amd.evaluators import BatchedPSRS
    if fmt == "B":
        monkeypatch.setenv("OFFSIM_STREAMS_FORMAT", "B")
    e = synth.synth_iid(N, nS, 3, seed=nS + 5)
    t0 = e["steps"] == 0
    table = TransitionTable(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], t0, device=gpu)
    pa, pb = synth.dirichlet_policy(nS, 3, seed=1), synth.dirichlet_policy(nS, 3, seed=2)
    seeds = [3, 4, 5, 6, 7]
    env = BatchedPSRS(table, len(seeds))
    env.reset_sampler(seeds, policy=table.policy_slots(pa))
    assert env._streams is not None and env.state.perm is None
    assert env._streams["format"] == {"A": L.STREAMS_A, "B": L.STREAMS_B, "C": L.STREAMS_C}[fmt]
    o1 = env.eval_mc(table.policy_slots(pa), 0.97, n_episodes=25, ep_cap=table.N0 + 1)
    o1 = {k: v.clone() for k, v in o1.items() if isinstance(v, torch.Tensor)}
    o2 = env.eval_mc(table.policy_slots(pb), 0.97, ep_cap=table.N0 + 1)  # (automatic mode: used to raise "laid out for another policy")
    torch.cuda.synchronize()
    L.check_async_faults()
    assert env.scan_variant() == "k_eval_mc_rows" and env.state.perm is None
    ora = O.OraclePSRS(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], t0)
    for i, sd in enumerate(seeds):
        ora.reset_sampler(sd)
        for o, ref in ((o1, ora.evalmc(25, pa, 0.97)), (o2, ora.evalmc(10 ** 9, pb, 0.97))):
            ne = int(o["n_ep"][i])
            assert int(o["steps"][i]) == ref["steps"] and int(o["cand"][i]) == ref["candidates"] and ne == len(ref["Gs"]), (i, sd)
            assert np.array_equal(o["ep_g"][i, :ne].cpu().numpy(), ref["Gs"])


def _bench(args, env_extra=None, timeout=900):
    env = dict(os.environ, **(env_extra or {}))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.timeout(900)
def test_rccl_executes_the_allreduce_of_the_estimates_on_one_rank(gpu):
    """The multi-GPU leg's collective on the hardware this box has: `bench.py --force-dist` creates the process group with backend
    "nccl" (= RCCL) for its ONE rank (device_id = cuda:0) and runs measure()'s all-reduce of the [R,2] f64 device tensor in every
    pass; a one-rank SUM leaves the table as it was, librccl is mapped into the process, and the line still passes its parity check
    against the oracle.  (Child process: nothing here re-execs after the GPU is initialised.)"""
    out = _bench(["--transitions", "300000", "--rollouts", "64", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--force-dist"])
    c = out["collective"]
    assert c["backend"] == "nccl" and c["world"] == 1 and c["device_tensor"] and c["librccl_mapped"], c
    assert c["unchanged_at_one_rank"] is True and c["bytes"] == 64 * 2 * 8 and c["allreduce_us"] > 0
    assert out["parity_check"]["ok"] and out["n_gpus"] == 1


def test_drop_in_drivers_raise_when_the_sampler_reset_gave_up_a_wait(gpu):
    """a shuffle role that gives up a bounded wait voids the orders of that call and raises only a device-wide fault
    word.  The host-facing drivers look at it where they synchronise anyway: with the fault-injection build (role A of the shuffle
    never starts, variants/lib_fault.so) PSRS.from_arrays -- reset_sampler() then reset(), psrs.py:13-14 -- raises OffsimError instead
    of serving a void order, and so does VectorPSRS(strict=True).reset_sampler; the word is read and cleared by ONE atomic exchange."""
    from _variants import fault_lib
    lib = fault_lib()
    code = r'''
import sys
sys.path.insert(0, %r)
import numpy as np, torch
from rl_offline_simulation_amd import synth, _lib as L
from rl_offline_simulation_amd.evaluators import PSRS, evalMC_psrs
e = synth.synth_iid(5000, 5, 2, seed=1)
try:
    PSRS.from_arrays(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], e["steps"] == 0)
    print("NO ERROR")
except L.OffsimError as ex:
    print("RAISED", "bounded wait" in str(ex))
assert L.load().offsim_async_faults() == 0  # read and cleared
print("ok")
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, OFFSIM_LIB=lib), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RAISED True" in r.stdout and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.timeout(900)
def test_bench_line_carries_the_extra_configurations(gpu):
    """bench.py appends C2 (CartPole, box encoder) and C3 (continuous_grid, MLP encoder on MFMA) to the headline line, each with its
    own oracle parity check.  Exercised here at a reduced headline size through the same code (the default sizes run in the driver's
    own bench call): OFFSIM_BENCH_CONFIG_SCALE shrinks the two logs."""
    out = _bench(["--steps", "1", "--warmup", "0", "--no-cpu-baseline"], env_extra={"OFFSIM_BENCH_TEST_SCALE": "50"}, timeout=900)
    cfg = out["configs"]
    assert set(cfg) == {"C2", "C3", "C4_shard", "C5_shard"}  # (round 5: one GPU's shard of C4 and of C5 are driver-timed lines too)
    for k in cfg:
        assert cfg[k]["parity_ok"] is True and cfg[k]["value"] > 0 and cfg[k]["scan_s"] > 0 and cfg[k]["kernel"].startswith("k_eval_mc")
        assert 0 < cfg[k]["roofline"]["frac"] < 1 and 0 < cfg[k]["roofline_reset"]["frac"] < 1
    c5 = cfg["C5_shard"]
    assert c5["p_log"] == "float16" and c5["roofline"]["bytes_per_candidate"] == 4 * 2 + 4 + 4
    enc = c5["encoder"]
    assert enc["argmax_equals_oracle_mlp"] is True and enc["argmax_rows_checked"] > 1000 and enc["bytes_per_row"] == 260 and 0 < enc["roofline"]["frac"] < 1
    assert enc["latent_equals_logged_state_frac"] > 0.99
    assert out["parity_check"]["ok"]
    # the line explains its own fraction: the kernel's clock read-out, the reset's fraction beside the scan's
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and rf["cycles_per_iteration"] > rf["step_floor_cycles"] > 0 and "chain" in rf["bound_measured"]
    assert rf["chain"]["iterations"] > 0 and 1.0 < rf["chain"]["shader_clock_GHz"] < 3.0
    assert 0 < out["roofline_reset"]["frac"] < 1 and out["roofline_reset"]["bytes_written_per_pass"] > 0


def test_step_server_serves_the_same_steps_as_one_launch_per_call(gpu):
    """PSRS.step through the resident step server (offsim_step_server_start: one wavefront, mailbox in host-coherent pinned memory)
    against one offsim_step_batch launch per call: same served rows, rewards, done flags, states and stream positions over whole
    episodes -- incl. env.reset() at episode ends (the server is stopped and started again around it), a float32 p_new (the other
    instance of the kernel), the Python-side reject hook (POP_ONE + set_state) and reset_sampler between runs; and against the oracle."""
    from oracle import oracle as O
    from rl_offline_simulation_amd import synth, _lib as L
    from rl_offline_simulation_amd.evaluators import PSRS
    e = synth.synth_iid(6000, 12, 3, seed=11)
    t0 = e["steps"] == 0
    args = (e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], t0)
    ora = O.OraclePSRS(*args)
    g = np.random.default_rng(5)
    ps = g.dirichlet(np.ones(3), size=4000)

    def run(server, p_dtype=np.float64, hook=False, seed=21):
        prev = os.environ.get("OFFSIM_STEP_SERVER")
        os.environ["OFFSIM_STEP_SERVER"] = "1" if server else "0"
        try:
            kw = {}
            if hook:
                kw["reject_func"] = lambda p_new, p_log, a: False  # accept whatever comes (FollowObservationOnly's rule, on the host)
            env = PSRS.from_arrays(*args, **kw)
            env.reset_sampler(seed)
            out, s = [], env.reset()
            for k in range(len(ps)):
                r = env.step(ps[k].astype(p_dtype))
                if r[0] is None:
                    break
                out.append((int(env._env.last_row), float(r[1]), bool(r[2]), int(env.z)))
                if r[2] and env.reset() is None:
                    break
            if server and not hook:
                assert env._env._mb is not None and env._env._mb.state in (L.SERVER_STARTING, L.SERVER_RUNNING, L.SERVER_EXITED)
            st = env.rejection_sampling_rng.bit_generator.state["state"]["state"]
            return out, st
        finally:
            if prev is None:
                os.environ.pop("OFFSIM_STEP_SERVER", None)
            else:
                os.environ["OFFSIM_STEP_SERVER"] = prev

    a, sa = run(True)
    b, sb = run(False)
    assert len(a) > 500 and a == b and sa == sb
    # the oracle on the same seed and distributions
    ora.reset_sampler(21)
    ora.reset()
    for k, (row, r, d, z) in enumerate(a):
        got, _ = ora.step(ps[k])
        assert got == row, k
        if d:
            ora.reset()
    for kw in (dict(p_dtype=np.float32), dict(hook=True), dict(seed=22)):
        x, sx = run(True, **kw)
        y, sy = run(False, **kw)
        assert len(x) > 100 and x == y and sx == sy, kw


def test_five_byte_streams_hold_the_same_orders_and_give_the_same_rollouts(gpu, monkeypatch):
    """Stream format C (include/offsim.h: 14-bit thresholds, bits 8..16 of the local row inside the digest, ONE byte beside it; every
    chain on the chunked shuffle) against the permutation form of the same reset, against format B on the same table and against the
    oracle: a table whose states hold 1, 2, 63 .. 65, 4095 .. 4097, 70000 and exactly 2^17 rows (the most the format takes; chains of
    one chunk and of a single row run through the chunked kernel here), 255 states; beyond either limit format B serves."""
    from oracle import oracle as O
    from rl_offline_simulation_amd import synth, _lib as L
    from rl_offline_simulation_amd.table import TransitionTable
    from rl_offline_simulation_amd.evaluators import BatchedPSRS
    from rl_offline_simulation_amd.evaluators.psrs import stream_format, rollout_resident_bytes
    lengths = [1, 2, 63, 64, 65, 4095, 4096, 4097, 70000, 1 << 17]
    N, nS, nA = sum(lengths), 255, 3
    e = synth.synth_iid(N, nS, nA, seed=77, p_done=0.05, p_init=0.01)
    z = np.repeat(np.array([0, 3, 9, 17, 50, 100, 101, 200, 253, 254]), lengths)
    g = np.random.default_rng(3)
    g.shuffle(z)
    e["z"] = z.astype(e["z"].dtype)
    # next states: mostly the two big states (long rollouts), sometimes the small ones
    zn = g.choice(np.array([253, 254, 200, 101, 100, 50, 17, 9, 3, 0]), N, p=[0.46, 0.46, 0.04, 0.02, 0.0195, 0.0001, 0.0001, 0.0001, 0.0001, 0.0001])
    e["z_next"] = zn.astype(e["z_next"].dtype)
    t0 = e["steps"] == 0
    args = (e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], t0)
    table = TransitionTable(*args, device=gpu)
    assert table.max_seg == 1 << 17 and stream_format(table) == L.STREAMS_C and rollout_resident_bytes(table) < 5 * N + 4 * table.N0 + 2048
    pi = synth.dirichlet_policy(nS, nA, seed=4)
    seeds = [11, 12, 13, 14, 15, 16]
    plain = BatchedPSRS(table, len(seeds))
    plain.reset_sampler(seeds)
    envc = BatchedPSRS(table, len(seeds))
    envc.reset_sampler(seeds, policy=table.policy_slots(pi))
    torch.cuda.synchronize()
    L.check_async_faults()
    assert envc._streams["format"] == L.STREAMS_C and envc._loc_buf.dtype == torch.uint8
    assert torch.equal(envc.perm.to(torch.int64) & 0xFFFFFFFF, plain.state.perm.to(torch.int64) & 0xFFFFFFFF)
    assert torch.equal(envc.state.init_perm, plain.state.init_perm)
    oc = envc.eval_mc(table.policy_slots(pi), 0.98, ep_cap=table.N0 + 1)
    assert envc.scan_variant() == "k_eval_mc_rows"
    monkeypatch.setenv("OFFSIM_STREAMS_FORMAT", "B")
    envb = BatchedPSRS(table, len(seeds))
    envb.reset_sampler(seeds, policy=table.policy_slots(pi))
    assert envb._streams["format"] == L.STREAMS_B and envb._loc_buf.dtype == torch.int16
    ob = envb.eval_mc(table.policy_slots(pi), 0.98, ep_cap=table.N0 + 1)
    torch.cuda.synchronize()
    L.check_async_faults()
    for k in ("steps", "cand", "n_ep", "sum_g", "status", "n_len"):
        assert torch.equal(oc[k], ob[k]), k
    assert torch.equal(oc["ep_g"], ob["ep_g"]) and torch.equal(oc["ep_len"], ob["ep_len"])
    monkeypatch.delenv("OFFSIM_STREAMS_FORMAT")
    ora = O.OraclePSRS(*args)
    for i, sd in enumerate(seeds[:3]):
        ora.reset_sampler(sd)
        ref = ora.evalmc(10 ** 9, pi, 0.98)
        ne = int(oc["n_ep"][i])
        assert int(oc["steps"][i]) == ref["steps"] and int(oc["cand"][i]) == ref["candidates"] and ne == len(ref["Gs"]), (i, sd)
        assert np.array_equal(oc["ep_g"][i, :ne].cpu().numpy(), ref["Gs"])
    # derived streams (reset without a policy, streams gathered from the permutations) take the same format
    envd = BatchedPSRS(table, 2)
    envd.reset_sampler(seeds[:2])
    od = envd.eval_mc(table.policy_slots(pi), 0.98, ep_cap=table.N0 + 1)
    assert envd.scan_variant() == "k_eval_mc_rows" and envd._streams["format"] == L.STREAMS_C
    for k in ("steps", "cand", "n_ep", "sum_g"):
        assert torch.equal(od[k], oc[k][:2]), k
    # one row more per state, or one state more: format B
    for lens, n_states in (([1, (1 << 17) + 1], 255), ([70000, 5], 256)):
        n = sum(lens)
        e2 = synth.synth_iid(n, n_states, 2, seed=5)
        e2["z"] = np.repeat(np.array([0, n_states - 1]), lens).astype(e2["z"].dtype)
        t2 = TransitionTable(e2["z"], e2["actions"], e2["rewards"], e2["z_next"], e2["terminals"], e2["action_distributions"], e2["steps"] == 0, device=gpu)
        assert stream_format(t2) == L.STREAMS_B, (lens, n_states)


_SHAPE_JOB = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from rl_offline_simulation_amd import _lib, synth
from rl_offline_simulation_amd.table import TransitionTable
from rl_offline_simulation_amd.evaluators import BatchedPSRS
_lib.load()
N, nS, nA = 2_000_000, 162, 2
e = synth.synth_iid(N, nS, nA, seed=21)
table = TransitionTable(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], e["steps"] == 0)
pi = table.policy_slots(synth.dirichlet_policy(nS, nA))
out = {}
for R in (1, 3, 250, 600, 1100, 2100):
    env = BatchedPSRS(table, R)
    env.reset_sampler(list(range(7, 7 + R)), policy=pi)
    o = env.eval_mc(pi, 0.99)
    torch.cuda.synchronize()
    assert env.scan_variant() == "k_eval_mc_rows"
    _lib.check_async_faults()
    for k in ("sum_g", "n_ep", "steps", "cand", "n_len", "status"):
        out[f"{R}_{k}"] = o[k].cpu().numpy()
    out[f"{R}_cursor"] = env.state.cursor.cpu().numpy()
    out[f"{R}_rng"] = env.state.rng.cpu().numpy()
np.savez(sys.argv[2], **out)
"""


@pytest.mark.timeout(900)
def test_spread_launches_of_the_row_packed_scan_give_the_packed_launch_results(gpu, tmp_path):
    """The launcher deals a launch of few rollouts out over the CUs (fewer chain wavefronts per workgroup, and below four rollouts per CU
    two or one rollouts to a chain wavefront, rows left empty: offsim_eval_mc_streams).  2 M transitions x 1, 3, 250, 600, 1100 and 2100
    rollouts -- one rollout per wavefront, two, two pairs of two, two and three pairs of four on a 256-CU device -- under the launcher's
    own choice, against the same jobs packed sixteen rollouts to a CU (four chain wavefronts of four: OFFSIM_ROWS_WAVES=4, every launch
    before this change): every per-rollout output, the cursors and the stream states, bit for bit.  (The shape is read once per process:
    two child processes.)"""
    script = tmp_path / "job.py"
    script.write_text(_SHAPE_JOB)
    res = {}
    for name, extra in (("auto", {}), ("packed", {"OFFSIM_ROWS_WAVES": "4", "OFFSIM_ROWS_PER_WAVE": "4"})):
        env = {k: v for k, v in os.environ.items() if k not in ("OFFSIM_ROWS_WAVES", "OFFSIM_ROWS_PER_WAVE")}
        env.update(extra, OFFSIM_SCAN_ROWS="1")
        r = subprocess.run([sys.executable, str(script), ROOT, str(tmp_path / (name + ".npz"))], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        res[name] = np.load(tmp_path / (name + ".npz"))
    assert sorted(res["auto"].files) == sorted(res["packed"].files)
    for k in res["auto"].files:
        assert np.array_equal(res["auto"][k], res["packed"][k]), k
    assert int(res["auto"]["2100_steps"].sum()) > 2100 * 500_000  # (whole rollouts: ~1 M steps each)
