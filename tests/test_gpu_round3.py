"""Round-3 GPU tests: the learner drivers' remaining arguments against the reference's outputs (schedules, ties broken with NumPy's
global stream, save_Q, greedy / soft-greedy behaviour), generic kernels after a keyed sampler reset, the hardware self-test."""
import numpy as np
import pytest

from common import load

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda", 0)


# the reference's tabular policies (offsim4rl/agents/tabular.py:4-32), restated: what a caller would pass
def _random_argmax(x):
    return np.random.choice(np.where(x == np.max(x))[0])


def greedy_policy(Q, args):
    pi = np.zeros_like(Q)
    for s, a in enumerate([_random_argmax(Q[s]) for s in range(len(Q))]):
        pi[s, a] = 1
    return pi


def soft_greedy_policy(Q, args):
    pi = np.zeros_like(Q)
    for s in range(len(Q)):
        pi[s, np.where(np.isclose(Q[s], np.max(Q[s])))[0]] = 1
    return pi / pi.sum(axis=1, keepdims=True)


def epsilon_greedy_policy(Q, args):
    epsilon = args["epsilon"]
    pi = np.ones_like(Q) * epsilon / (Q.shape[1])
    for s, a in enumerate([_random_argmax(Q[s]) for s in range(len(Q))]):
        pi[s, a] = 1 - epsilon + epsilon / (Q.shape[1])
    return pi


sched_alpha = lambda ep: 0.5 / (1.0 + 0.1 * ep)   # (tests/golden/make_golden.py, section 11b)
sched_eps = lambda ep: max(0.05, 0.9 ** ep)


def test_learner_drivers_schedules_ties_snapshots_against_the_reference(gpu):
    """qlearn_psrs / expSARSA_psrs (psrs.py:119-239) with callable alpha / epsilon, Q_init = None (ties at every first visit, broken
    by np.random.choice on the global MT19937 stream: agents/tabular.py:4-5), save_Q, greedy_policy and soft_greedy_policy: Q, Gs, TD
    errors, accepted rows, every step's behaviour distribution and the position of the global stream afterwards equal what the
    reference produced (tests/golden/td2_iid_2k.npz)."""
    from rl_offline_simulation_amd.evaluators import PSRS, qlearn_psrs, expSARSA_psrs
    d = load("td2_iid_2k")
    env = PSRS.from_arrays(d["in_z"], d["in_a"], d["in_r"], d["in_z_next"], d["in_done"], d["in_p_log"], d["in_t0"], nS=d["Q_init"].shape[0], nA=5)
    gam, Qi, pi = float(d["gamma"]), d["Q_init"], d["pi"]
    runs = {
        "sched": lambda: qlearn_psrs(env, 10 ** 9, epsilon_greedy_policy, gam, alpha=sched_alpha, epsilon=sched_eps, Q_init=Qi),
        "ties": lambda: qlearn_psrs(env, 10 ** 9, epsilon_greedy_policy, gam, alpha=0.1, epsilon=0.3, Q_init=None),
        "ties_sched": lambda: qlearn_psrs(env, 10 ** 9, epsilon_greedy_policy, gam, alpha=sched_alpha, epsilon=sched_eps, Q_init=None),
        "greedy": lambda: qlearn_psrs(env, 10 ** 9, greedy_policy, gam, alpha=0.1, Q_init=None),
        "soft": lambda: qlearn_psrs(env, 10 ** 9, soft_greedy_policy, gam, alpha=0.1, Q_init=None),
        "saveq": lambda: qlearn_psrs(env, 12, epsilon_greedy_policy, gam, alpha=0.1, epsilon=0.2, Q_init=None, save_Q=1),
        "es_sched": lambda: expSARSA_psrs(env, 12, pi, gam, alpha=sched_alpha, save_Q=1),
    }
    for s in d["seeds"]:
        s = int(s)
        for tag, fn in runs.items():
            k = f"s{s}_{tag}"
            np.random.seed(int(d[k + "_np_seed"]))
            env.reset_sampler(s)
            Q, info = fn()
            after = np.random.random(3)
            rows = d[k + "_rows"]
            assert [int(m[6]["a"]) for m in info["memory"]] == [int(d["in_a"][r]) for r in rows], (tag, "accepted rows")
            if tag.startswith("es_"):  # expected SARSA: Q[S_] @ pi[S_] is BLAS in the reference (rounding only)
                assert np.abs(Q - d[k + "_Q"]).max() <= 1e-12 and np.abs(info["Qs"] - d[k + "_Qs"]).max() <= 1e-12
            else:
                assert np.array_equal(Q, d[k + "_Q"]), tag
                assert np.array_equal(info["TD_errors"], d[k + "_td"]), tag
                if tag == "saveq":
                    assert np.array_equal(info["Qs"], d[k + "_Qs"])
            assert np.array_equal(info["Gs"], d[k + "_Gs"]), tag
            assert np.array_equal(np.array([m[5] for m in info["memory"]]), d[k + "_p"]), (tag, "behaviour distributions")
            assert np.array_equal(after, d[k + "_after"]), (tag, "the global NumPy stream stands where the reference leaves it")


def test_generic_kernels_after_a_keyed_sampler_reset(gpu):
    """reset_sampler(seeds, policy=...) lays the queue orders out as candidate streams only; step(), step_single() and eval_td() walk
    permutations and must see the SAME orders (rebuilt from the streams), not table order: compared with the oracle."""
    from oracle import oracle as O
    from rl_offline_simulation_amd import synth, _lib as L
    from rl_offline_simulation_amd.table import TransitionTable
    from rl_offline_simulation_amd.evaluators import BatchedPSRS
    e = synth.synth_iid(3000, 25, 5, seed=5)
    t0 = e["steps"] == 0
    table = TransitionTable(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], t0, device=gpu)
    pi = synth.dirichlet_policy(25, 5)
    seeds = [3, 4, 5]
    env = BatchedPSRS(table, len(seeds))
    env.reset_sampler(seeds, policy=table.policy_slots(pi))
    assert env.state.perm is None  # (keyed: streams only)
    first = env.reset().cpu().numpy()
    g = np.random.default_rng(0)
    oras = []
    for i, sd in enumerate(seeds):
        o = O.OraclePSRS(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], t0)
        o.reset_sampler(sd)
        assert o.reset() == first[i]
        oras.append(o)
    alive = np.ones(len(seeds), bool)
    for it in range(200):
        p_new = g.dirichlet(np.ones(5), size=len(seeds))
        row, status, popped = (x.cpu().numpy() for x in env.step(p_new))
        again = np.zeros(len(seeds), bool)
        for i, o in enumerate(oras):
            if not alive[i]:
                continue
            ref_row, ref_pop = o.step(p_new[i])
            assert popped[i] == ref_pop, (it, i)
            if ref_row is None:
                assert status[i] == L.ST_EXHAUSTED
                alive[i] = False
                continue
            assert status[i] == L.ST_OK and int(row[i]) == ref_row, (it, i)
            if e["terminals"][ref_row]:
                again[i] = True
        if again.any():
            nxt = env.reset(mask=torch.from_numpy(again).to(gpu)).cpu().numpy()
            for i in np.nonzero(again)[0]:
                ref = oras[i].reset()
                assert (ref is None and nxt[i] < 0) or ref == nxt[i]
                if ref is None:
                    alive[i] = False
    # eval_td after a keyed reset walks the same orders too: its accepted rows equal a plain (permutation) reset's
    from rl_offline_simulation_amd import _lib
    env.reset_sampler(seeds, policy=table.policy_slots(pi))
    a = env.eval_td(table.policy_slots(pi), 0.9, _lib.TD_EXPSARSA, 0.1, trace_cap=3001)
    env.reset_sampler(seeds)
    b = env.eval_td(table.policy_slots(pi), 0.9, _lib.TD_EXPSARSA, 0.1, trace_cap=3001)
    assert torch.equal(a["trace_row"], b["trace_row"]) and torch.equal(a["q"], b["q"])


def test_lds_atomic_order_selftest(gpu):
    """The hardware property the scan's queue positions rely on (one ds_add_rtn_u32, same address: ascending lane order)."""
    from rl_offline_simulation_amd import _lib as L
    out = torch.full((1,), -1, dtype=torch.int64, device=gpu)
    L.check(L.load().offsim_selftest_lds_atomic_order(out.data_ptr(), L.stream_ptr()))
    torch.cuda.synchronize()
    assert int(out[0]) == 0


def test_shuffle_protocol_slip_ends_with_an_error_not_a_hang(gpu):
    """A build whose applier role never starts (-DSHUF_FAULT_INJECT, rl-offline-simulation_amd/csrc/variants/lib_fault.so): the
    classifier runs into the bound of its wait for room in the j ring, every role leaves, the call returns within seconds and
    offsim_async_faults() / check_async_faults() report it.  Run in a child process (the library is chosen per process)."""
    import os, subprocess, sys, time
    here = os.path.dirname(os.path.abspath(__file__))
    from _variants import fault_lib
    lib = fault_lib()
    code = """
import sys, time
sys.path.insert(0, %r)
import numpy as np, torch
from rl_offline_simulation_amd import synth, _lib as L
from rl_offline_simulation_amd.table import TransitionTable
from rl_offline_simulation_amd.evaluators import BatchedPSRS
e = synth.synth_iid(20000, 25, 5, seed=1)
t = TransitionTable(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], e["steps"] == 0)
env = BatchedPSRS(t, 8)
assert L.load().offsim_async_faults() == 0
t0 = time.time()
env.reset_sampler(list(range(8)))
torch.cuda.synchronize()
dt = time.time() - t0
v = L.load().offsim_async_faults()
print("FAULTS", v, "SECONDS", round(dt, 2))
assert v & L.FAULT_SHUFFLE and dt < 5.0
env.reset_sampler(list(range(8)))
torch.cuda.synchronize()
try:
    L.check_async_faults()
except L.OffsimError as ex:
    print("RAISED", ex)
else:
    raise SystemExit("check_async_faults did not raise")
""" % os.path.dirname(here)
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, OFFSIM_LIB=lib), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "RAISED" in r.stdout and "FAULTS" in r.stdout
    # and the product build raises nothing on the same job
    from rl_offline_simulation_amd import synth, _lib as L
    from rl_offline_simulation_amd.table import TransitionTable
    from rl_offline_simulation_amd.evaluators import BatchedPSRS
    e = synth.synth_iid(20000, 25, 5, seed=1)
    t = TransitionTable(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], e["steps"] == 0, device=gpu)
    env = BatchedPSRS(t, 8)
    env.reset_sampler(list(range(8)))
    torch.cuda.synchronize()
    L.check_async_faults()


def test_bench_two_ranks_on_one_device(gpu):
    """The whole multi-rank bench path -- launcher, episode shards, per-rank tile from the free HBM, all-reduce of the per-seed
    estimates, parity check against the oracle -- with both ranks on this box's one GPU (gloo; RCCL needs two)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--all-ranks-on-device0", "--dist-backend", "gloo",
                        "--transitions", "2000000", "--rollouts", "256", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["parity_check"]["ok"]
    assert line["config"]["rollout_tile"] == 256 and line["config"]["queue_orders_resident_bytes_rank0"] > 256 * 6 * 900000
    assert line["rollout_sharded"]["rollouts_per_gpu"] == 128


def test_philox_stream_provider_against_the_reference_with_a_replayed_stream(gpu):
    """OFFSIM_STREAM_PHILOX in the GENERIC kernels (rocRAND's Philox4x32-10 through its device API): the rows the reference's PSRS serves when its
    rejection_sampling_rng replays the same stream (tests/golden/philox_iid_2k.npz) -- step protocol through
    offsim_step_batch, evalMC through offsim_eval_mc; accepted rows, candidates popped per step, Gs and lengths equal."""
    from rl_offline_simulation_amd import _lib as L
    from rl_offline_simulation_amd.table import TransitionTable
    from rl_offline_simulation_amd.evaluators import BatchedPSRS
    d = load("philox_iid_2k")
    table = TransitionTable(d["in_z"], d["in_a"], d["in_r"], d["in_z_next"], d["in_done"], d["in_p_log"], d["in_t0"], device=gpu)
    seeds = [int(s) for s in d["seeds"]]
    R = len(seeds)
    env = BatchedPSRS(table, R)
    # evalMC
    env.reset_sampler(seeds)
    env.set_rejection_seeds(seeds, provider="philox")
    cap = table.N + 1
    # (fast=False: the generic kernel, the literal rocrand_init / rocrand API; since round 6 the default would derive candidate streams
    # and run the row-packed scan on the same stream -- tests/test_gpu_round6.py)
    o = env.eval_mc(table.policy_slots(d["pi"]), float(d["gamma"]), ep_cap=table.N0 + 1, trace_cap=cap, fast=False)
    torch.cuda.synchronize()
    assert o["_kernel"] == "k_eval_mc" and env.scan_variant() != "k_eval_mc_rows"  # (the generic kernel ran; no streams were derived)
    for i, s in enumerate(seeds):
        n = int(o["steps"][i])
        assert np.array_equal(o["trace_row"][i, :n].cpu().numpy(), d[f"s{s}_mc_rows"])
        assert np.array_equal(o["trace_pop"][i, :n].cpu().numpy(), d[f"s{s}_mc_popped"][:n])
        ne, nl = int(o["n_ep"][i]), int(o["n_len"][i])
        assert np.array_equal(o["ep_g"][i, :ne].cpu().numpy(), d[f"s{s}_mc_Gs"])
        assert np.array_equal(o["ep_len"][i, :nl].cpu().numpy(), d[f"s{s}_mc_lengths"])
    # the step protocol of the reference's tests/test_psrs.py:25-31 (one fixed p_new, reset on done), all seeds side by side
    env.reset_sampler(seeds)
    env.set_rejection_seeds(seeds, provider="philox")
    first = env.reset().cpu().numpy()
    p = np.tile(d["p_new_step"], (R, 1))
    rows = [[] for _ in seeds]
    pops = [[] for _ in seeds]
    alive = first >= 0
    for it in range(4 * table.N):
        if not alive.any():
            break
        row, status, popped = (x.cpu().numpy() for x in env.step(p))
        again = np.zeros(R, bool)
        for i in range(R):
            if not alive[i]:
                continue
            pops[i].append(int(popped[i]))
            rows[i].append(int(row[i]))
            if status[i] != L.ST_OK:
                alive[i] = False
            elif d["in_done"][row[i]]:
                again[i] = True
        if again.any():
            nxt = env.reset(mask=torch.from_numpy(again).to(gpu)).cpu().numpy()
            alive &= ~(again & (nxt < 0))
        env.set_state(torch.full((R,), -1, dtype=torch.int32), mask=torch.from_numpy(~alive).to(gpu)) if (~alive).any() else None
    for i, s in enumerate(seeds):
        assert rows[i] == [int(x) for x in d[f"s{s}_step_rows"]], s
        assert pops[i] == [int(x) for x in d[f"s{s}_step_popped"]], s


@pytest.mark.gpu
@pytest.mark.parametrize("chunk", ["4096", "8192", "16384"])
def test_chunked_shuffle_of_states_that_do_not_fit_lds_equals_the_in_place_orders(chunk):
    """offsim_shuffle_queues_keys_ws (csrc/shuffle_chunk.hpp: chunks top-down in LDS, message and reply lists) against the
    in-place global-memory variant and, for one-state tables, against NumPy's shuffle (oracle restatement of psrs.py:29-30): segment
    lengths around the chunk size and its multiples, several states per table, more chains than persistent workgroups take at once."""
    import os, subprocess, sys
    code = r'''
import os, sys
import numpy as np, torch
sys.path.insert(0, %r)
from oracle import oracle as O
from rl_offline_simulation_amd import synth, _lib as L
from rl_offline_simulation_amd.table import TransitionTable
from rl_offline_simulation_amd.evaluators import BatchedPSRS
cb = int(os.environ["OFFSIM_SHUFFLE_CHUNK"])
shapes = [(65537, 1, 3), (cb * 9 - 1, 1, 2), (cb * 9, 1, 2), (cb * 9 + 1, 1, 2), (cb * 8 + 63, 1, 2), (cb * 8 + 65, 1, 2), (300000, 2, 5), (1000000, 3, 3), (700000, 5, 700), (900000, 40, 4)]
for n, nS, R in shapes:
    e = synth.synth_iid(n, nS, 2, seed=n)
    # (every third shape: each row an initial state, so the init queue is a long chain too -- with 40 states also the ONLY long one)
    t0 = np.ones(n, bool) if shapes.index((n, nS, R)) %% 3 == 0 else e["steps"] == 0
    table = TransitionTable(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], t0)
    pi = table.policy_slots(synth.dirichlet_policy(nS, 2))
    seeds = [int(x) for x in np.random.default_rng(n).integers(0, 1 << 62, R)]
    out = {}
    os.environ["OFFSIM_STREAMS_FORMAT"] = "B"  # (the same layout from both kernels; the 5-byte layout only the chunked kernel writes: below)
    for mode in ("1", "0"):
        os.environ["OFFSIM_SHUFFLE_CHUNKED"] = mode
        env = BatchedPSRS(table, R)
        env.reset_sampler(seeds, policy=pi)
        torch.cuda.synchronize()
        assert (getattr(env, "_ws", None) is not None) == (mode == "1" and max(table.max_seg, table.N0) > 65536), (n, mode)
        assert L.load().offsim_async_faults() == 0
        out[mode] = env
    assert torch.equal(out["1"]._dig_buf, out["0"]._dig_buf) and torch.equal(out["1"]._loc_buf, out["0"]._loc_buf), (n, nS)
    assert torch.equal(out["1"]._init_perm_buf, out["0"]._init_perm_buf), (n, nS)
    del os.environ["OFFSIM_STREAMS_FORMAT"]
    os.environ["OFFSIM_SHUFFLE_CHUNKED"] = "1"
    envc = BatchedPSRS(table, R)  # stream format C where it applies (states of 65537 .. 131072 rows): every chain on the chunked kernel
    envc.reset_sampler(seeds, policy=pi)
    torch.cuda.synchronize()
    assert L.load().offsim_async_faults() == 0
    assert (envc._streams["format"] == L.STREAMS_C) == (65536 < table.max_seg <= 131072 and nS <= 255), (n, nS)
    assert torch.equal(envc.perm, out["0"].perm) and torch.equal(envc._init_perm_buf, out["0"]._init_perm_buf), (n, nS)
    plain = {}
    for mode in ("1", "0"):  # the same chains as permutations (offsim_shuffle_queues_ws)
        os.environ["OFFSIM_SHUFFLE_CHUNKED"] = mode
        env = BatchedPSRS(table, R)
        env.reset_sampler(seeds)
        torch.cuda.synchronize()
        assert L.load().offsim_async_faults() == 0
        plain[mode] = env
    assert torch.equal(plain["1"].state.perm, plain["0"].state.perm) and torch.equal(plain["1"].state.init_perm, plain["0"].state.init_perm), (n, nS)
    assert torch.equal(plain["1"].state.perm.to(torch.int64) & 0xFFFFFFFF, out["1"].perm.to(torch.int64) & 0xFFFFFFFF), (n, nS)
    if nS == 1:  # (as in test_gpu_fuzz.py: a one-state table's queue order is default_rng(seed).shuffle of its rows)
        perm = (out["1"].perm.to(torch.int64) & 0xFFFFFFFF).cpu().numpy()
        for k, sd in enumerate(seeds):
            assert np.array_equal(perm[k, :n], O.permutation(sd, n)), (n, sd)
# more states than the kernel's work list holds (the permutation form takes any number): the in-place shuffle serves the long init queue
os.environ["OFFSIM_SHUFFLE_CHUNKED"] = "1"
e = synth.synth_iid(150000, 1000, 2, seed=5)
table = TransitionTable(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], np.ones(150000, bool))
env = BatchedPSRS(table, 2)
env.reset_sampler([11, 12])
torch.cuda.synchronize()
assert L.load().offsim_async_faults() == 0
ip = (env.state.init_perm.to(torch.int64) & 0xFFFFFFFF).cpu().numpy()
for k, sd in enumerate([11, 12]):
    assert np.array_equal(ip[k, :150000], O.permutation(sd, 150000))
print("ok")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OFFSIM_SHUFFLE_CHUNK=chunk)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-2000:]


def test_chunked_shuffle_list_overflow_raises_the_fault_and_a_small_workspace_falls_back(gpu):
    """csrc/shuffle_chunk.hpp: (1) a message list that overflows (-DSHC_TEST_SMALL_LISTS in variants/lib_fault.so: 16 entries per
    list) ends the chain with OFFSIM_FAULT_SHUFFLE instead of writing past its list, and the call returns; (2) a workspace that does
    not hold one workgroup's pools gives the in-place shuffle -- same orders."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    from _variants import fault_lib
    lib = fault_lib()
    code = """
import sys, time
sys.path.insert(0, %r)
import numpy as np, torch
from rl_offline_simulation_amd import synth, _lib as L
from rl_offline_simulation_amd.table import TransitionTable
from rl_offline_simulation_amd.evaluators import BatchedPSRS
n = 200000
e = synth.synth_iid(n, 1, 2, seed=1)
t = TransitionTable(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], np.ones(n, bool))  # both chains are long
env = BatchedPSRS(t, 4)
assert L.load().offsim_async_faults() == 0
t0 = time.time()
env.reset_sampler([1, 2, 3, 4], policy=t.policy_slots(synth.dirichlet_policy(1, 2)))
torch.cuda.synchronize()
dt = time.time() - t0
v = L.load().offsim_async_faults()
print("FAULTS", v, "SECONDS", round(dt, 2))
assert env._ws is not None and v & L.FAULT_SHUFFLE and dt < 10.0
""" % os.path.dirname(here)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, OFFSIM_LIB=lib), capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    # (2), in this process with the product library
    from rl_offline_simulation_amd import synth, _lib as L
    from rl_offline_simulation_amd.table import TransitionTable
    from rl_offline_simulation_amd.evaluators import BatchedPSRS
    e = synth.synth_iid(150000, 1, 2, seed=2)
    t = TransitionTable(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], e["steps"] == 0, device=gpu)
    pi = t.policy_slots(synth.dirichlet_policy(1, 2))
    a, b = BatchedPSRS(t, 3), BatchedPSRS(t, 3)
    b._ws = torch.empty(8192, dtype=torch.uint8, device=gpu)  # header + 4 KB: no pools
    b._ws_wg = 1 << 30  # (what _shuffle_workspace remembers having asked for: it keeps this workspace instead of allocating a full one)
    a.reset_sampler([7, 8, 9], policy=pi)
    b.reset_sampler([7, 8, 9], policy=pi)
    torch.cuda.synchronize()
    assert a._ws.numel() > (1 << 20) and b._ws.numel() == 8192 and L.load().offsim_async_faults() == 0  # a: chunked kernel, b: in place
    assert torch.equal(a._dig_buf, b._dig_buf) and torch.equal(a._loc_buf, b._loc_buf) and torch.equal(a._init_perm_buf, b._init_perm_buf)
    # (3) a table that would take stream format C (one state of 100 k rows: 65536 < max_seg <= 2^17), which only the chunked shuffle can
    # write: with a workspace that holds no workgroup the format is decided as B BEFORE the loc stream is allocated and the reset falls
    # back to the in-place shuffle instead of raising -- same queue orders, same evaluation
    e = synth.synth_iid(100000, 1, 2, seed=3)
    t = TransitionTable(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], e["steps"] == 0, device=gpu)
    pi = t.policy_slots(synth.dirichlet_policy(1, 2))
    a, b = BatchedPSRS(t, 3), BatchedPSRS(t, 3)
    b._ws = torch.empty(8192, dtype=torch.uint8, device=gpu)
    b._ws_wg = 1 << 30
    a.reset_sampler([7, 8, 9], policy=pi)
    b.reset_sampler([7, 8, 9], policy=pi)
    torch.cuda.synchronize()
    assert a._stream_format() == L.STREAMS_C and a._loc_buf.dtype == torch.uint8 and b._stream_format() == L.STREAMS_B and b._loc_buf.dtype == torch.int16
    assert L.load().offsim_async_faults() == 0 and torch.equal(a.perm, b.perm) and torch.equal(a._init_perm_buf, b._init_perm_buf)
    oa, ob = a.eval_mc(pi, 0.99), b.eval_mc(pi, 0.99)
    torch.cuda.synchronize()
    assert all(torch.equal(oa[k], ob[k]) for k in ("sum_g", "n_ep", "steps", "cand"))


def test_scan_kernel_is_chosen_by_the_table(gpu):
    """BatchedPSRS._streams_apply without OFFSIM_SCAN_ROWS: a table in which a tick of 16 steps takes more than 1.2 candidates out of its
    busiest window (16 x the largest state's share of the rows / the policy's acceptance) takes the window kernel on permutations, otherwise
    the row-packed kernel on streams; all give the oracle's results (child process: the suite itself forces the row-packed kernel,
    tests/conftest.py)."""
    import os, subprocess, sys
    code = r'''
import os, sys
import numpy as np, torch
sys.path.insert(0, %r)
from oracle import oracle as O
from rl_offline_simulation_amd import synth
from rl_offline_simulation_amd.table import TransitionTable
from rl_offline_simulation_amd.evaluators import BatchedPSRS
assert "OFFSIM_SCAN_ROWS" not in os.environ
# a hot state; neither; low acceptance alone (round 3's rule took the window kernel here); low acceptance AND rather few states
# ... and (round 5) 50 equal states at four actions: L = 1.1, but a FULL 8-entry window is all rejected in 6 %% of the looks at acceptance 0.29 -- the
# 32-entry window kernel; the same acceptance with 70 states (beyond 64 the window kernel has 8 entries too) stays on the row-packed kernel
for nS, nA, want in ((10, 2, "k_eval_mc_win"), (120, 2, "k_eval_mc_rows"), (120, 5, "k_eval_mc_rows"), (40, 5, "k_eval_mc_win"), (50, 4, "k_eval_mc_win"),
                     (70, 4, "k_eval_mc_rows")):
    e = synth.synth_iid(60000, nS, nA, seed=nS)
    t0 = e["steps"] == 0
    table = TransitionTable(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], t0)
    pi = synth.dirichlet_policy(nS, nA)
    seeds = [3, 4, 5, 6, 7]
    env = BatchedPSRS(table, len(seeds))
    env.reset_sampler(seeds, policy=table.policy_slots(pi))
    o = env.eval_mc(table.policy_slots(pi), 0.98, ep_cap=table.N0 + 1)
    torch.cuda.synchronize()
    assert env.scan_variant() == want, (nS, env.scan_variant(), table.max_seg, table.N)
    ora = O.OraclePSRS(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], t0)
    for i, sd in enumerate(seeds):
        ora.reset_sampler(sd)
        ref = ora.evalmc(10 ** 9, pi, 0.98)
        ne = int(o["n_ep"][i])
        assert int(o["steps"][i]) == ref["steps"] and int(o["cand"][i]) == ref["candidates"] and ne == len(ref["Gs"])
        assert np.array_equal(o["ep_g"][i, :ne].cpu().numpy(), ref["Gs"])
print("ok")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k != "OFFSIM_SCAN_ROWS"}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-2000:]
