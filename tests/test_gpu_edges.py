"""Edge cases the reference's behaviour defines: empty and ragged inputs, odd rollout counts, many states (generic
kernel), large seeds, episode caps -- every one against the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (no CPU fallback exists)")
    return torch.device("cuda", 0)


def _compare(e, pi, gamma, seeds, gpu, n_episodes=None, fast=None):
    from oracle import oracle as O
    from rl_offline_simulation_amd.table import TransitionTable
    from rl_offline_simulation_amd.evaluators import BatchedPSRS
    t0 = e["steps"] == 0
    table = TransitionTable(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], t0, device=gpu)
    env = BatchedPSRS(table, len(seeds))
    env.reset_sampler(seeds)
    N = len(e["z"])
    o = env.eval_mc(table.policy_slots(pi), gamma, n_episodes=n_episodes, ep_cap=table.N0 + 1, trace_cap=N + 1, fast=fast)
    torch.cuda.synchronize()
    ora = O.OraclePSRS(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], t0)
    for i, s in enumerate(seeds):
        ora.reset_sampler(s)
        try:
            ref = ora.evalmc(10 ** 9 if n_episodes is None else n_episodes, pi, gamma, trace_cap=N + 1)
        except KeyError:
            assert int(o["status"][i]) == 3
            continue
        n = ref["steps"]
        assert int(o["steps"][i]) == n and int(o["cand"][i]) == ref["candidates"]
        assert np.array_equal(o["trace_row"][i, :n].cpu().numpy(), ref["trace_rows"])
        ne = int(o["n_ep"][i])
        assert np.array_equal(o["ep_g"][i, :ne].cpu().numpy(), ref["Gs"])
        assert np.array_equal(o["ep_len"][i, : int(o["n_len"][i])].cpu().numpy(), ref["lengths"])
    return table, o


def test_empty_dataset(gpu):
    from rl_offline_simulation_amd.table import TransitionTable
    from rl_offline_simulation_amd.evaluators import BatchedPSRS, PSRS
    z = np.zeros(0, np.int64)
    table = TransitionTable(z, z, np.zeros(0), z, np.zeros(0, bool), np.zeros((0, 3)), np.zeros(0, bool), device=gpu)
    assert table.N == 0 and table.N0 == 0
    env = BatchedPSRS(table, 3)
    env.reset_sampler([1, 2, 3])
    o = env.eval_mc(np.full((table.n_slots, 3), 1 / 3), 0.9)
    torch.cuda.synchronize()
    assert o["status"].cpu().tolist() == [2, 2, 2] and o["steps"].cpu().tolist() == [0, 0, 0]
    single = PSRS.from_arrays(z, z, np.zeros(0), z, np.zeros(0, bool), np.zeros((0, 3)), np.zeros(0, bool), nS=1, nA=3)
    assert single.reset() is None  # psrs.py:33-35


@pytest.mark.parametrize("R", [1, 3, 5, 67])
def test_odd_rollout_counts(R, gpu):
    from rl_offline_simulation_amd import synth
    e = synth.synth_iid(3000, 12, 3, seed=R)
    _compare(e, synth.dirichlet_policy(12, 3), 0.9, list(range(100, 100 + R)), gpu)


def test_ragged_segments_and_single_rows(gpu):
    """States with 0, 1 and 2 rows next to a dominant state; self-loops; every row terminal; no terminal at all."""
    from rl_offline_simulation_amd import synth
    g = np.random.default_rng(0)
    N = 4000
    e = synth.synth_iid(N, 40, 2, seed=4)
    z = np.where(g.random(N) < 0.7, 5, e["z"])          # one dominant state
    zn = np.where(g.random(N) < 0.5, z, e["z_next"])     # half of the transitions are self-loops
    z[:3] = [37, 38, 38]                                  # 1-row and 2-row queues; state 39 never occurs as a from-state
    e = dict(e, z=z, z_next=zn, observations=z, next_observations=zn)
    pi = synth.dirichlet_policy(40, 2)
    _compare(e, pi, 0.99, [0, 1, 2, 3], gpu)
    _compare(dict(e, terminals=np.ones(N, bool)), pi, 0.99, [0, 1], gpu)
    _compare(dict(e, terminals=np.zeros(N, bool)), pi, 0.99, [0, 1], gpu)


def test_many_states_take_the_generic_kernel(gpu):
    from rl_offline_simulation_amd import synth, _lib
    e = synth.synth_iid(30_000, 700, 3, seed=9)
    table, o = _compare(e, synth.dirichlet_policy(700, 3), 0.95, [0, 1, 2], gpu)
    assert table.n_slots > 256
    from rl_offline_simulation_amd.evaluators import BatchedPSRS
    env = BatchedPSRS(table, 1)
    env.reset_sampler([0])
    with pytest.raises(_lib.OffsimError):
        env.eval_mc(table.policy_slots(synth.dirichlet_policy(700, 3)), 0.95, fast=True)


def test_large_seeds_and_episode_caps(gpu):
    from rl_offline_simulation_amd import synth
    e = synth.synth_iid(5000, 25, 5, seed=6)
    pi = synth.dirichlet_policy(25, 5)
    seeds = [2 ** 32 - 1, 2 ** 32, 2 ** 40 + 5, 2 ** 63 + 11]
    _compare(e, pi, 0.99, seeds, gpu)
    for cap in (0, 1, 7):
        for fast in (True, False):
            _compare(e, pi, 0.99, [3, 4], gpu, n_episodes=cap, fast=fast)


def test_shared_and_table_order_modes_agree_between_kernels(gpu):
    from rl_offline_simulation_amd import synth
    from rl_offline_simulation_amd.table import TransitionTable
    from rl_offline_simulation_amd.evaluators import BatchedPSRS
    e = synth.synth_iid(20_000, 60, 4, seed=12)
    table = TransitionTable(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], e["steps"] == 0, device=gpu)
    pi = table.policy_slots(synth.dirichlet_policy(60, 4))
    for mode in ("shared", "table_order"):
        outs = []
        for fast in (True, False):
            env = BatchedPSRS(table, 6)
            env.reset_sampler(list(range(6)), mode, shuffle_seed=77)
            outs.append(env.eval_mc(pi, 0.9, trace_cap=20_001, fast=fast))
        torch.cuda.synchronize()
        for k in ("sum_g", "n_ep", "steps", "cand", "n_len", "status", "trace_row", "trace_pop"):
            assert torch.equal(outs[0][k], outs[1][k]), (mode, k)


@pytest.mark.parametrize("N,nS,p_t0,skew", [
    (64, 1, 1.0, False),          # one group of exactly 64 steps is never full: the partial-group path
    (1000, 7, None, False),       # small segments, several chains per compute unit
    (70000, 1, 1.0, False),       # one state of 70000 rows and an init queue of 70000: the in-place global-memory variant
    (131072, 2, 0.5, False),      # segments around the 65536-row LDS capacity, init queue just below it
    (300000, 40, None, True),     # a 210000-row state next to tiny ones
])
def test_reset_sampler_orders_equal_the_oracle_shuffles(N, nS, p_t0, skew, gpu):
    """offsim_shuffle_queues (shuffle_wave.hpp: wave-parallel Fisher-Yates) against the oracle's restatement of
    default_rng(seed).shuffle per queue (psrs.py:22-23, 29-30), for both memory variants and the rare paths
    (mask boundaries, settled accepts, arithmetic cuts, equal partners, partial last group)."""
    from oracle import oracle as O
    from rl_offline_simulation_amd import synth
    from rl_offline_simulation_amd.table import TransitionTable, seeds_tensor, shuffle_queues
    e = synth.synth_iid(N, nS, 2, seed=N + nS)
    if skew:
        rng = np.random.default_rng(5)
        e["z"] = np.where(rng.random(N) < 0.7, 0, rng.integers(0, nS, N)).astype(e["z"].dtype)
    t0 = e["steps"] == 0 if p_t0 is None else np.random.default_rng(3).random(N) < p_t0
    table = TransitionTable(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], t0, device=gpu)
    seeds = [0, 7, 2 ** 40 + 3]
    perm, init_perm = shuffle_queues(table, seeds_tensor(np.asarray(seeds, dtype=np.uint64), gpu))
    torch.cuda.synchronize()
    perm = perm.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    init_perm = init_perm.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    so = table.seg_off.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    for k, seed in enumerate(seeds):
        for s in range(table.n_slots):
            n = int(so[s + 1] - so[s])
            if n:
                assert np.array_equal(perm[k, so[s]:so[s + 1]], so[s] + O.permutation(seed, n)), (seed, s, n)
        if table.N0:
            assert np.array_equal(init_perm[k, :table.N0], O.permutation(seed, table.N0)), (seed, "init")


@pytest.mark.parametrize("variant", ["rows_single", "win", "auto", "waves1", "waves2", "waves3", "rpw1", "rpw2"])
def test_every_scan_variant_passes_the_parity_suite(variant, gpu):
    """The fast path of eval_mc has three bit-identical forms: csrc/scan_rows.hpp with a helper wavefront per chain wavefront
    (four rollouts per wavefront, candidate streams: the default wherever it applies, so the whole in-process suite runs on
    it), the same kernel as a single wavefront (OFFSIM_ROWS_HELPER=0; also what every call with trace outputs runs), and
    csrc/scan_win.hpp on queue permutations (OFFSIM_SCAN_ROWS=0).  The switches are read per process: the golden-fixture
    parity tests, the config tests, the round-2 tests and the edge cases of this file run again in a child process for each.
    "auto" is the PRODUCT's own choice (OFFSIM_SCAN_ROWS=auto: BatchedPSRS._streams_apply picks the kernel by the table, which the
    suite otherwise overrides, tests/conftest.py): the parity, config and edge files once more, whatever kernel each table gets -- and
    whatever launch shape: the row-packed kernel runs with one to four chain wavefronts per workgroup (a launch of few rollouts is spread
    over the CUs, offsim_eval_mc_streams); the suite pins four, "auto" leaves the choice to the launcher, "wavesN" force the other shapes
    (partly filled workgroups included: 8 rollouts on a workgroup of 12), "rpwN" the number of rollouts a chain wavefront carries (one or
    two of its four rows instead of all: what a launch of at most two rollouts per CU gets)."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ)
    if variant == "win":
        env["OFFSIM_SCAN_ROWS"] = "0"
        env["OFFSIM_ENCODER_F32"] = "1"  # (and the encoder's exact-f32 products, csrc/encode_mfma.hpp: the other members run it on bf16 x 3)
    elif variant == "auto":
        env["OFFSIM_SCAN_ROWS"] = "auto"
        env.pop("OFFSIM_ROWS_WAVES", None)
    elif variant.startswith("waves"):
        env["OFFSIM_ROWS_WAVES"] = variant[5:]
    elif variant.startswith("rpw"):  # rollouts per chain wavefront (a sparse launch leaves rows of a wavefront empty on purpose)
        env["OFFSIM_ROWS_PER_WAVE"] = variant[3:]
    else:
        env["OFFSIM_ROWS_HELPER"] = "0"
    files = ["test_gpu_parity.py", "test_gpu_edges.py"] + (["test_gpu_round2.py"] if variant == "waves1" else []) if variant.startswith(("waves", "rpw")) else \
        ["test_gpu_parity.py", "test_gpu_configs.py", "test_gpu_edges.py"] + (["test_gpu_round2.py"] if variant == "rows_single" else [])
    skip = "not every_scan_variant and not headline_table_size and not two_ranks" + (" and not headline_job" if variant.startswith(("waves", "rpw")) else "")
    r = subprocess.run([sys.executable, "-m", "pytest"] + [os.path.join(here, f) for f in files] +
                       ["-m", "gpu", "-x", "-q", "-k", skip],
                       env=env, capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def _compare_untraced(e, pi, gamma, seeds, gpu, t0=None, n_episodes=None):
    """eval_mc without trace outputs (the helper-wavefront form of csrc/scan_rows.hpp wherever it applies) against the oracle:
    counts, every episode's return and length, and the in-order sum of returns."""
    from oracle import oracle as O
    from rl_offline_simulation_amd.table import TransitionTable
    from rl_offline_simulation_amd.evaluators import BatchedPSRS
    t0 = (e["steps"] == 0) if t0 is None else t0
    table = TransitionTable(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], t0, device=gpu)
    env = BatchedPSRS(table, len(seeds))
    env.reset_sampler(seeds)
    o = env.eval_mc(table.policy_slots(pi), gamma, n_episodes=n_episodes, ep_cap=table.N0 + 1)
    torch.cuda.synchronize()
    import os
    if os.environ.get("OFFSIM_SCAN_ROWS") not in ("0", "auto"):
        assert env.scan_variant() == "k_eval_mc_rows"  # (the candidate streams were derived: this ran the row-packed kernel)
    ora = O.OraclePSRS(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], t0)
    for i, s in enumerate(seeds):
        ora.reset_sampler(s)
        ref = ora.evalmc(10 ** 9 if n_episodes is None else n_episodes, pi, gamma)
        assert int(o["steps"][i]) == ref["steps"] and int(o["cand"][i]) == ref["candidates"], (i, s)
        ne = int(o["n_ep"][i])
        assert ne == len(ref["Gs"])
        assert np.array_equal(o["ep_g"][i, :ne].cpu().numpy(), ref["Gs"])
        assert np.array_equal(o["ep_len"][i, : int(o["n_len"][i])].cpu().numpy(), ref["lengths"])
        acc = 0.0
        for g in ref["Gs"]:
            acc += float(g)
        assert float(o["sum_g"][i]) == acc
    return env, o


@pytest.mark.parametrize("p_done,p_init,N", [(0.7, 0.8, 30000), (0.5, 0.5, 30000), (0.9, 0.02, 20000), (0.3, 0.0005, 20000), (0.02, 0.02, 3000)])
def test_short_episodes_and_small_init_queues(p_done, p_init, N, gpu):
    """Initial states reach the chain through a 32-entry LDS ring that is refilled one load per tick (csrc/scan_rows.hpp):
    one- and two-step episodes use it up faster than it is refilled (the fetch is then finished on the spot), init queues
    shorter than the ring or than one refill end the rollout with env.reset() returning None (psrs.py:33-35, 250-252)."""
    from rl_offline_simulation_amd import synth
    e = synth.synth_iid(N, 20, 3, seed=int(1000 * p_done) + N, p_done=p_done, p_init=p_init)
    pi = synth.dirichlet_policy(20, 3)
    _compare_untraced(e, pi, 0.9, [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10], gpu)
    _compare_untraced(e, pi, 0.9, [11, 12, 13], gpu, n_episodes=37)
