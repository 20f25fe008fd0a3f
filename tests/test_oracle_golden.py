"""Pins the CPU oracle (oracle/psrs_oracle.c) against the golden vectors the reference produced
(tests/golden/make_golden.py).  CPU only."""
import glob
import os

import numpy as np
import pytest

from oracle import oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PSRS_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz"))
                    if not os.path.basename(p).startswith(("rng", "enc_", "td_", "td2_", "queue_", "exo_", "h5_", "philox_")))


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def make_oracle(d):
    return O.OraclePSRS(d["in_z"], d["in_a"], d["in_r"], d["in_z_next"], d["in_done"], d["in_p_log"], d["in_t0"])


def prob_dtype(d, key):
    return O.PROB_F32 if (d[key].dtype == np.float32 and d["in_p_log"].dtype == np.float32) else O.PROB_F64


def test_rng_layer():
    d = load("rng")
    for i, s in enumerate(d["seeds"]):
        s = int(s)
        assert np.array_equal(O.seedseq_words(s), d["state_words"][i])
        assert np.array_equal(O.rng_doubles(s, 64), d["doubles64"][i])
    for k in d.files:
        if k.startswith("perm_"):
            n = int(k[5:])
            seeds = d["seeds"] if n <= 1000 else np.concatenate([d["seeds"][:2], d["seeds"][-1:]])
            for i, s in enumerate(seeds):
                assert np.array_equal(O.permutation(int(s), n), d[k][i].astype(np.int64)), (k, s)


def test_fixture_inventory():
    assert len(PSRS_CASES) >= 18


@pytest.mark.parametrize("name", PSRS_CASES)
def test_psrs_case(name):
    d = load(name)
    env = make_oracle(d)
    mode = int(d["reject_mode"])
    shared = int(d["shared_shuffle_seed"]) if "shared_shuffle_seed" in d.files else None
    for s in d["seeds"]:
        s = int(s)

        def reset_sampler():
            if shared is None:
                env.reset_sampler(s)
            else:
                env.reset_sampler(shared)
                env.set_rejection_seed(s)

        reset_sampler()
        if f"s{s}_keys" in d.files:
            keys, off, q, iq = env.orders()
            assert np.array_equal(keys, d[f"s{s}_keys"])
            assert np.array_equal(off, d[f"s{s}_off"])
            assert np.array_equal(q, d[f"s{s}_queue"])
            assert np.array_equal(iq, d[f"s{s}_init"])
        if "p_new_step" in d.files:  # protocol of the reference's tests/test_psrs.py:25-31
            p_new = d["p_new_step"]
            rows, popped, resets = [], [], []
            status = "none"
            row0 = env.reset()
            resets.append(-2 if row0 is None else env.cur_z)
            alive = row0 is not None
            try:
                while alive:
                    row, n = env.step(p_new, prob_dtype(d, "p_new_step"), mode)
                    rows.append(-1 if row is None else row)
                    popped.append(n)
                    if row is None:
                        break
                    if d["in_done"][row]:
                        r0 = env.reset()
                        resets.append(-2 if r0 is None else env.cur_z)
                        alive = r0 is not None
            except KeyError:
                status = "keyerror"
                rows.append(-3)
                popped.append(0)
            assert status == str(d[f"s{s}_step_status"])
            assert np.array_equal(rows, d[f"s{s}_step_rows"])
            assert np.array_equal(popped, d[f"s{s}_step_popped"])
            assert np.array_equal(resets, d[f"s{s}_step_reset_z"])
            reset_sampler()
        if "pi" in d.files:
            want_status = str(d[f"s{s}_mc_status"])
            try:
                res = env.evalmc(10 ** 9, d["pi"], float(d["gamma"]), prob_dtype(d, "pi"), mode, trace_cap=env.N)
                status = "ok"
            except KeyError:
                status = "keyerror"
            assert status == want_status
            if status == "ok":
                assert np.array_equal(res["Gs"], d[f"s{s}_mc_Gs"])  # bit-exact f64
                assert np.array_equal(res["lengths"], d[f"s{s}_mc_lengths"])
                assert np.array_equal(res["trace_rows"], d[f"s{s}_mc_rows"])
                pop = d[f"s{s}_mc_popped"]
                assert res["candidates"] == int(pop.sum())
                assert np.array_equal(res["trace_popped"], pop[: len(res["trace_popped"])])
                if len(res["Gs"]):
                    assert abs(res["Gs"].mean() - float(d[f"s{s}_mc_mean"])) <= 1e-12


@pytest.mark.parametrize("name", ["philox_iid_2k", "philox_iid_50k"])
def test_philox_replay_stream(name):
    """The oracle's second provider of the rejection stream (Philox4x32-10 as rocRAND's device API lays it out, include/offsim.h
    OFFSIM_STREAM_PHILOX): pinned on what the reference's own PSRS served with rejection_sampling_rng replaying that stream
    (tests/golden/make_golden.py philox_fixtures)."""
    d = load(name)
    src = load(str(d["inputs_of"])) if "inputs_of" in d.files else d
    env = make_oracle(src)
    for i, s in enumerate(d["seeds"]):
        s = int(s)
        if "first_draws" in d.files:
            assert np.array_equal(O.philox_doubles(s, 8), d["first_draws"][i])
            assert np.array_equal(O.philox_doubles(s, 5, first=3), d["first_draws"][i][3:])
        env.reset_sampler(s)
        env.set_rejection_philox(s)
        res = env.evalmc(10 ** 9, d["pi"], float(d["gamma"]), O.PROB_F64, O.REJECT_DEFAULT, trace_cap=env.N)
        assert np.array_equal(res["Gs"], d[f"s{s}_mc_Gs"])
        assert np.array_equal(res["lengths"], d[f"s{s}_mc_lengths"])
        assert np.array_equal(res["trace_rows"], d[f"s{s}_mc_rows"])
        assert np.array_equal(res["trace_popped"], d[f"s{s}_mc_popped"][: len(res["trace_popped"])])
        if "p_new_step" in d.files:  # the step protocol (one fixed p_new, reset on done)
            env.reset_sampler(s)
            env.set_rejection_philox(s)
            rows, popped = [], []
            alive = env.reset() is not None
            while alive:
                row, n = env.step(d["p_new_step"])
                rows.append(-1 if row is None else row)
                popped.append(n)
                if row is None:
                    break
                if src["in_done"][row]:
                    alive = env.reset() is not None
            assert np.array_equal(rows, d[f"s{s}_step_rows"]) and np.array_equal(popped, d[f"s{s}_step_popped"])
        env.reset_sampler(s)  # ... and reset_sampler puts the PCG64 stream back (psrs.py:20)
        assert env.evalmc(10 ** 9, d["pi"], float(d["gamma"]))["candidates"] != res["candidates"] or name != "philox_iid_50k"


def test_cartpole_box_encoder():
    d = load("enc_cartpole_box")
    assert np.array_equal(O.cartpole_encode(d["obs"]), d["z"])
    assert (d["z"] == -1).sum() > 50


@pytest.mark.parametrize("name", ["enc_mlp_2_64_25", "enc_mlp_128_64_50", "enc_mlp_4_16_10"])
def test_mlp_encoder(name):
    d = load(name)
    z, logits = O.mlp_encode(d["x"], d["W1"], d["b1"], d["W2"], d["b2"])
    assert np.abs(logits - d["logits"]).max() <= 1e-5
    clear = d["gap"] > 1e-4  # argmax may flip only on near-ties
    assert np.array_equal(z[clear], d["z"][clear])
    assert clear.mean() > 0.99
