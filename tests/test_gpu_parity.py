"""Parity tests proper: the HIP path (through the C ABI) against the golden vectors the reference produced and
against the CPU oracle on seeded inputs.  Bit-exact for queue orders, accepted rows, candidate counts, per-episode
returns and lengths; value estimate within 1e-5."""
import numpy as np
import pytest

from common import PSRS_CASES, load

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (no CPU fallback exists)")
    from rl_offline_simulation_amd import _lib
    _lib.load()
    return torch.device("cuda", 0)


def build_table(d, gpu):
    from rl_offline_simulation_amd.table import TransitionTable
    return TransitionTable(d["in_z"], d["in_a"], d["in_r"], d["in_z_next"], d["in_done"], d["in_p_log"], d["in_t0"], device=gpu)


def golden_orders(table, perm_row, init_perm_row):
    """device orders -> (keys, off, queue rows, init rows) in the golden layout"""
    seg = table.seg_off.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    order = table.order.cpu().numpy().astype(np.int64)
    lens = np.diff(seg)
    keys = table.slot_z[np.nonzero(lens)[0]]
    off = np.concatenate([[0], np.cumsum(lens[lens > 0])])
    q = order[perm_row.cpu().numpy().astype(np.int64) & 0xFFFFFFFF][: table.N]
    init = table.init_orig.cpu().numpy().astype(np.int64)[init_perm_row.cpu().numpy().astype(np.int64)][: table.N0]
    return keys, off, q, init


@pytest.mark.parametrize("fast", [False, True], ids=["generic", "windows"])
@pytest.mark.parametrize("name", PSRS_CASES)
def test_golden_case(name, fast, gpu):
    from rl_offline_simulation_amd.evaluators import BatchedPSRS, SHUFFLE_PER_ROLLOUT, SHUFFLE_SHARED
    d = load(name)
    table = build_table(d, gpu)
    seeds = [int(s) for s in d["seeds"]]
    R = len(seeds)
    env = BatchedPSRS(table, R, int(d["reject_mode"]))
    shared = int(d["shared_shuffle_seed"]) if "shared_shuffle_seed" in d.files else None

    def reset_sampler():
        if shared is None:
            env.reset_sampler(seeds, SHUFFLE_PER_ROLLOUT)
        else:
            env.reset_sampler(seeds, SHUFFLE_SHARED, shuffle_seed=shared)

    reset_sampler()
    # a2: queue orders and init order, bit-exact (psrs.py:22-30)
    for i, s in enumerate(seeds):
        if f"s{s}_keys" not in d.files:
            continue
        j = 0 if shared is not None else i
        keys, off, q, init = golden_orders(table, env.state.perm[j], env.state.init_perm[j])
        assert np.array_equal(keys, d[f"s{s}_keys"])
        assert np.array_equal(off, d[f"s{s}_off"])
        assert np.array_equal(q, d[f"s{s}_queue"])
        assert np.array_equal(init, d[f"s{s}_init"])
    if "pi" in d.files:
        pi = d["pi"]
        cap = table.N + 1
        if fast and (d["pi"].dtype == np.float32 or int(d["reject_mode"]) != 0):
            pytest.skip("the compiled-policy scan covers f64 probabilities with the default rule")
        o = env.eval_mc(table.policy_slots(pi), float(d["gamma"]), ep_cap=table.N0 + 1, trace_cap=cap, fast=fast)
        torch.cuda.synchronize()
        st = o["status"].cpu().numpy()
        for i, s in enumerate(seeds):
            want = str(d[f"s{s}_mc_status"])
            if want == "keyerror":
                assert st[i] == 3
                continue
            assert st[i] in (1, 2)
            n = int(o["steps"][i])
            rows = d[f"s{s}_mc_rows"]
            assert n == len(rows)
            assert np.array_equal(o["trace_row"][i, :n].cpu().numpy(), rows)
            pop = d[f"s{s}_mc_popped"]
            assert int(o["cand"][i]) == int(pop.sum())
            assert np.array_equal(o["trace_pop"][i, :n].cpu().numpy(), pop[:n])
            ne, nl = int(o["n_ep"][i]), int(o["n_len"][i])
            Gs = o["ep_g"][i, :ne].cpu().numpy()
            assert np.array_equal(Gs, d[f"s{s}_mc_Gs"])  # bit-exact f64
            assert np.array_equal(o["ep_len"][i, :nl].cpu().numpy(), d[f"s{s}_mc_lengths"])
            if ne:
                assert abs(float(o["sum_g"][i]) / ne - float(d[f"s{s}_mc_mean"])) <= 1e-5  # north_star tolerance
        if fast:  # the same evaluation without trace outputs: the hand-scheduled chain loop and its helper wavefronts
            reset_sampler()
            o2 = env.eval_mc(table.policy_slots(pi), float(d["gamma"]), ep_cap=table.N0 + 1, fast=fast)
            torch.cuda.synchronize()
            for k in ("sum_g", "n_ep", "steps", "cand", "n_len", "status", "ep_g", "ep_len"):
                assert torch.equal(o[k], o2[k]), k


STEP_CASES = [c for c in PSRS_CASES if "p_new_step" in load(c).files and load(c)["in_z"].shape[0] <= 5000]


@pytest.mark.parametrize("name", STEP_CASES)
def test_golden_step_protocol(name, gpu):
    """The reference's tests/test_psrs.py:25-31 loop through the drop-in PSRS class (R = 1 launches)."""
    from rl_offline_simulation_amd.evaluators import PSRS
    d = load(name)
    mode = int(d["reject_mode"])
    env = PSRS.from_arrays(d["in_z"], d["in_a"], d["in_r"], d["in_z_next"], d["in_done"], d["in_p_log"], d["in_t0"],
                           reject_mode=mode)
    p_new = d["p_new_step"]
    for s in d["seeds"]:
        s = int(s)
        env.reset_sampler(s)
        rows, resets, status = [], [], "none"
        obs = env.reset()
        resets.append(-2 if obs is None else env.z)
        try:
            while obs is not None:
                obs, r, done, info = env.step(p_new)
                if obs is None:
                    rows.append(-1)
                    break
                rows.append(env._env.last_row)
                if done:
                    obs = env.reset()
                    resets.append(-2 if obs is None else env.z)
        except KeyError:
            status = "keyerror"
            rows.append(-3)
        assert status == str(d[f"s{s}_step_status"])
        assert np.array_equal(rows, d[f"s{s}_step_rows"])
        assert np.array_equal(resets, d[f"s{s}_step_reset_z"])


def test_python_reject_hook(gpu):
    """A user `reject_func` (psrs.py:11,48) runs on the host against candidates popped on the device."""
    from rl_offline_simulation_amd.evaluators import PSRS
    d = load("grid_10x10")
    calls = []

    def never(p_new, p_log, a):
        calls.append(int(a))
        return False

    env = PSRS.from_arrays(d["in_z"], d["in_a"], d["in_r"], d["in_z_next"], d["in_done"], d["in_p_log"], d["in_t0"], reject_func=never)
    ref = PSRS.from_arrays(d["in_z"], d["in_a"], d["in_r"], d["in_z_next"], d["in_done"], d["in_p_log"], d["in_t0"], reject_mode=1)
    env.reset_sampler(3)
    ref.reset_sampler(3)
    a, b = env.reset(), ref.reset()
    assert a == b
    for _ in range(8):
        x, y = env.step(d["p_new_step"]), ref.step(d["p_new_step"])
        assert x[0] == y[0] and x[2] == y[2]
        if x[0] is None or x[2]:
            break
    assert len(calls) >= 1


def test_oracle_parity_many_seeds(gpu):
    """Seeded synthetic input at a size the oracle finishes in seconds: 200k transitions x 32 rollouts."""
    from oracle import oracle as O
    from rl_offline_simulation_amd import synth
    from rl_offline_simulation_amd.table import TransitionTable
    from rl_offline_simulation_amd.evaluators import BatchedPSRS
    N, nS, nA, R = 200_000, 162, 2, 32
    e = synth.synth_iid(N, nS, nA, seed=5)
    pi = synth.dirichlet_policy(nS, nA)
    t0 = e["steps"] == 0
    table = TransitionTable(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], t0, device=gpu)
    env = BatchedPSRS(table, R)
    seeds = [1000 + 7 * i for i in range(R)]
    env.reset_sampler(seeds)
    o = env.eval_mc(table.policy_slots(pi), 0.99, trace_cap=N)
    env.reset_sampler(seeds)
    og = env.eval_mc(table.policy_slots(pi), 0.99, trace_cap=N, fast=False)
    torch.cuda.synchronize()
    for k in ("sum_g", "n_ep", "steps", "cand", "n_len", "status", "trace_row", "trace_pop"):
        assert torch.equal(o[k], og[k]), k  # window kernel == generic kernel, bit for bit
    ora = O.OraclePSRS(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], t0)
    for i, s in enumerate(seeds):
        ora.reset_sampler(s)
        ref = ora.evalmc(10 ** 9, pi, 0.99, trace_cap=N)
        n = ref["steps"]
        assert int(o["steps"][i]) == n and int(o["cand"][i]) == ref["candidates"]
        assert np.array_equal(o["trace_row"][i, :n].cpu().numpy(), ref["trace_rows"])
        assert int(o["n_ep"][i]) == len(ref["Gs"])
        assert abs(float(o["sum_g"][i]) - ref["Gs"].sum()) <= 1e-9 * max(1.0, abs(ref["Gs"].sum()))
        assert abs(float(o["sum_g"][i]) / len(ref["Gs"]) - ref["Gs"].mean()) <= 1e-5


def test_resume_across_calls(gpu):
    """The rollout state written back by eval_mc lets a second call continue exactly where the first stopped."""
    from rl_offline_simulation_amd.evaluators import BatchedPSRS
    d = load("iid_2k_s25_a5")
    table = build_table(d, gpu)
    seeds = [int(s) for s in d["seeds"]]
    pi = table.policy_slots(d["pi"])
    a = BatchedPSRS(table, len(seeds))
    a.reset_sampler(seeds)
    o1 = a.eval_mc(pi, 0.99, n_episodes=3, ep_cap=8)
    o2 = a.eval_mc(pi, 0.99, ep_cap=table.N0 + 1)
    torch.cuda.synchronize()
    for i, s in enumerate(seeds):
        g = np.concatenate([o1["ep_g"][i, : int(o1["n_ep"][i])].cpu().numpy(), o2["ep_g"][i, : int(o2["n_ep"][i])].cpu().numpy()])
        assert np.array_equal(g, d[f"s{s}_mc_Gs"])


def test_size_independent_properties(gpu):
    """At a size the oracle is not run on (2M transitions x 256 rollouts): properties the domain guarantees.
    - every queue permutation is a permutation of its own segment;
    - candidates >= steps, candidates <= N, no cursor beyond its segment;
    - the same seed twice gives identical results (determinism); different seeds differ;
    - REJECT_NEVER consumes exactly one candidate per step."""
    from rl_offline_simulation_amd import synth
    from rl_offline_simulation_amd.table import TransitionTable
    from rl_offline_simulation_amd.evaluators import BatchedPSRS
    N, nS, nA, R = 2_000_000, 162, 2, 256
    e = synth.synth_iid(N, nS, nA, seed=11)
    t0 = e["steps"] == 0
    table = TransitionTable(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], t0, device=gpu)
    pi = table.policy_slots(synth.dirichlet_policy(nS, nA))
    seeds = list(range(R - 1)) + [0]
    env = BatchedPSRS(table, R)
    env.reset_sampler(seeds)
    seg = table.seg_off.cpu().numpy().astype(np.int64)
    for r in (0, 17):
        p = env.state.perm[r].cpu().numpy().astype(np.int64)
        for s in (0, 5, nS - 1):
            assert np.array_equal(np.sort(p[seg[s]:seg[s + 1]]), np.arange(seg[s], seg[s + 1]))
    o = env.eval_mc(pi, 0.99)
    torch.cuda.synchronize()
    steps, cand = o["steps"].cpu().numpy(), o["cand"].cpu().numpy()
    assert (cand >= steps).all() and (cand <= N).all() and (steps > 0).all()
    cur = env.state.cursor.cpu().numpy().astype(np.int64)
    assert (cur <= np.diff(seg)[None, :]).all()
    assert cur.sum(axis=1).tolist() == cand.tolist()
    assert steps[0] == steps[-1] and float(o["sum_g"][0]) == float(o["sum_g"][-1])
    assert len(set(steps[:-1].tolist())) > R // 2
    env2 = BatchedPSRS(table, 4, reject_mode=1)
    env2.reset_sampler([0, 1, 2, 3])
    o2 = env2.eval_mc(pi, 0.99)
    assert o2["steps"].cpu().tolist() == o2["cand"].cpu().tolist()


def test_group_by_state_matches_stable_sort(gpu):
    from rl_offline_simulation_amd.table import group_by_state
    g = np.random.default_rng(0)
    for N, nS in ((0, 3), (1, 1), (5000, 7), (300_000, 163), (70_001, 1000)):
        slot = g.integers(0, nS, N).astype(np.int32)
        seg, order = group_by_state(torch.from_numpy(slot).to(gpu), nS)
        want = np.argsort(slot, kind="stable")
        assert np.array_equal(order.cpu().numpy(), want)
        assert np.array_equal(seg.cpu().numpy(), np.concatenate([[0], np.cumsum(np.bincount(slot, minlength=nS))]))


def test_seed_streams_and_rng_golden(gpu):
    from rl_offline_simulation_amd.table import seed_streams, seeds_tensor
    d = load("rng")
    out = seed_streams(seeds_tensor(d["seeds"], gpu)).cpu().numpy().view(np.uint64)
    for i, s in enumerate(d["seeds"]):
        bg = np.random.PCG64(int(s)).state["state"]
        assert (int(out[i, 0]) << 64) | int(out[i, 1]) == bg["state"]
        assert (int(out[i, 2]) << 64) | int(out[i, 3]) == bg["inc"]


def test_encoders_golden(gpu):
    from rl_offline_simulation_amd.encoders import CartpoleBoxEncoder, HOMEREncoder
    d = load("enc_cartpole_box")
    assert np.array_equal(CartpoleBoxEncoder().encode(d["obs"]), d["z"])
    for name in ("enc_mlp_2_64_25", "enc_mlp_128_64_50", "enc_mlp_4_16_10"):
        d = load(name)
        H, dO = d["W1"].shape
        nZ = d["W2"].shape[0]
        enc = HOMEREncoder(dO, 5, nZ, H, state_dict={"obs_encoder.0.weight": d["W1"], "obs_encoder.0.bias": d["b1"],
                                                     "obs_encoder.2.weight": d["W2"], "obs_encoder.2.bias": d["b2"]})
        z, logits = enc.encode_device(torch.from_numpy(d["x"]).to(gpu), return_logits=True)
        assert np.abs(logits.cpu().numpy() - d["logits"]).max() <= 1e-5  # SURVEY H6 tolerance
        clear = d["gap"] > 1e-4
        assert np.array_equal(z.cpu().numpy()[clear], d["z"][clear])
        assert np.array_equal(enc.encode(d["x"])[clear], d["z"][clear])


def test_drop_in_facade(gpu):
    """tests/test_per_state_rejection.py:7-21 and tests/test_trivial_baselines.py:8-31 of the reference."""
    from rl_offline_simulation_amd import OfflineDataset, ProbDistribution, spaces
    from rl_offline_simulation_amd.evaluators import (PerStateRejectionSampling, FollowObservationOnly, FollowActionOnly,
                                                      ServeRandomTransitions)
    for cls in (PerStateRejectionSampling, FollowObservationOnly, FollowActionOnly, ServeRandomTransitions):
        for new_api in (False, True):
            ds = OfflineDataset(
                observation_space=spaces.Discrete(25), action_space=spaces.Discrete(4), action_dist_type=ProbDistribution.Discrete,
                observations=np.array([0, 5], dtype=np.int64), actions=np.array([0, 1], dtype=np.int64),
                action_distributions=np.full((2, 4), fill_value=0.25, dtype=np.float32), rewards=np.array([0.0, 1.0], dtype=np.float32),
                next_observations=np.array([5, 7], dtype=np.int64), terminals=np.array([False, True], dtype=bool))
            psrs = cls(ds, new_step_api=new_api)
            obs = psrs.reset()
            assert obs.shape == tuple()
            out = psrs.step_dist(np.full(4, 0.25))
            assert len(out) == (6 if new_api else 5)
            with pytest.raises(NotImplementedError):
                psrs.step(0)
            try:  # the second step may start in state 7, which has no queue: KeyError, as in the reference (psrs.py:44)
                out = psrs.step_dist(torch.distributions.Categorical(probs=torch.ones(4) / 4))
                assert len(out) == (6 if new_api else 5)
            except KeyError as e:
                assert e.args[0] == 7


def test_mlp_encoder_shapes_vs_oracle(gpu):
    """MFMA encoder against the CPU oracle on shapes beyond the fixtures: odd input width, padded hidden/latent sizes,
    fp16 observations (config C5), N not a multiple of 32, and a shape that takes the VALU fallback (H = 160 > 128 is refused)."""
    from oracle import oracle as O
    from rl_offline_simulation_amd.encoders import HOMEREncoder
    g = np.random.default_rng(0)
    for (N, dO, H, nZ, half) in ((1000, 3, 16, 10, False), (4097, 2, 64, 25, False), (2500, 128, 64, 50, True), (333, 7, 128, 33, False),
                                 (31, 4, 96, 5, False)):
        W1, b1 = g.standard_normal((H, dO)).astype(np.float32) / np.sqrt(dO), g.standard_normal(H).astype(np.float32) * 0.1
        W2, b2 = g.standard_normal((nZ, H)).astype(np.float32) / np.sqrt(H), g.standard_normal(nZ).astype(np.float32) * 0.1
        x = g.standard_normal((N, dO)).astype(np.float32)
        if half:
            x = x.astype(np.float16)
        enc = HOMEREncoder(dO, 5, nZ, H, state_dict={"obs_encoder.0.weight": W1, "obs_encoder.0.bias": b1,
                                                     "obs_encoder.2.weight": W2, "obs_encoder.2.bias": b2})
        z, logits = enc.encode_device(torch.from_numpy(x).to(gpu), return_logits=True)
        zo, lo = O.mlp_encode(x.astype(np.float32), W1, b1, W2, b2)
        lg = logits.cpu().numpy()
        assert np.abs(lg - lo).max() <= 1e-5 * max(1.0, np.abs(lo).max())
        top2 = np.sort(lo, axis=1)[:, -2:]
        clear = (top2[:, 1] - top2[:, 0]) > 1e-4
        assert np.array_equal(z.cpu().numpy()[clear], zo[clear])
        assert np.array_equal(z.cpu().numpy(), lg.argmax(1))  # the kernel's argmax is the first maximum of its own logits
