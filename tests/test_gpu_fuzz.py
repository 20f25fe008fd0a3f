"""Randomised parity sweep of the untraced fast path (sampler reset in its keyed or permutation form + the row-packed scan
with helper wavefronts) against the CPU oracle: random table shapes, skews, episode / initial-state densities, rollout
counts, discount factors and episode caps; every rollout's counts, per-episode returns and lengths must be equal.

As a test: 150 cases (OFFSIM_FUZZ_CASES / OFFSIM_FUZZ_SEED override).  As a script, for longer sweeps on the GPU box:
    python tests/test_gpu_fuzz.py [n_cases] [seed]"""
import os, sys, time
import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def sweep(n_cases, seed, verbose=True, big=False):
    import torch
    from oracle import oracle as O
    from rl_offline_simulation_amd import synth
    from rl_offline_simulation_amd.table import TransitionTable
    from rl_offline_simulation_amd.evaluators import BatchedPSRS
    g = np.random.default_rng(seed)
    bad = 0
    ran = rollouts = steps_checked = variants_rows = 0
    t_start = time.time()
    for case in range(n_cases):
        nS = int(g.choice([1, 2, 3, 7, 25, 50, 162, 200, 256]))
        nA = int(g.choice([2, 3, 5]))
        N = int(g.choice([300, 3000, 20000, 60000, 150000]))
        if big:  # states with more than 65536 rows: the second layout of the candidate streams (offsim_streams.format B)
            nS, N = int(g.choice([1, 2, 3, 5, 25])), int(g.choice([150000, 400000, 1000000]))
        p_done = float(g.choice([0.002, 0.02, 0.1, 0.5, 0.9]))
        p_init = float(g.choice([0.0005, 0.02, 0.2, 0.8]))
        R = int(g.choice([1, 3, 4, 5, 16, 33])) if not big else int(g.choice([1, 3, 4, 5]))
        skew = bool(g.random() < 0.3)
        cap = None if g.random() < 0.7 else int(g.integers(0, 40))
        gamma = float(g.choice([0.0, 0.9, 0.99, 1.0]))
        keyed = bool(g.random() < 0.6)
        e = synth.synth_iid(N, nS, nA, seed=int(g.integers(1 << 30)), p_done=p_done, p_init=p_init)
        if skew and nS > 1:
            e["z"] = np.where(g.random(N) < 0.6, 0, e["z"]).astype(e["z"].dtype)
        t0 = e["steps"] == 0
        table = TransitionTable(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], t0)
        if big and table.max_seg <= 65536:
            continue
        pi = synth.dirichlet_policy(nS, nA, seed=int(g.integers(1 << 30)))
        seeds = [int(x) for x in g.integers(0, 1 << 40, R)]
        env = BatchedPSRS(table, R)
        env.reset_sampler(seeds, policy=table.policy_slots(pi) if keyed else None)
        o = env.eval_mc(table.policy_slots(pi), gamma, n_episodes=cap, ep_cap=table.N0 + 1)
        torch.cuda.synchronize()
        variant = env.scan_variant()
        ran += 1
        variants_rows += variant == "k_eval_mc_rows"
        ora = O.OraclePSRS(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], t0)
        desc = dict(case=case, nS=nS, nA=nA, N=N, p_done=p_done, p_init=p_init, R=R, skew=skew, cap=cap, gamma=gamma, keyed=keyed, variant=variant)
        for i, sd in enumerate(seeds):
            ora.reset_sampler(sd)
            try:
                ref = ora.evalmc(10 ** 9 if cap is None else cap, pi, gamma)
            except KeyError:
                if int(o["status"][i]) != 3:
                    print("FAIL (KeyError expected)", desc, "rollout", i); bad += 1
                continue
            ne = int(o["n_ep"][i])
            rollouts += 1
            steps_checked += ref["steps"]
            ok = (int(o["steps"][i]) == ref["steps"] and int(o["cand"][i]) == ref["candidates"] and ne == len(ref["Gs"])
                  and np.array_equal(o["ep_g"][i, :ne].cpu().numpy(), ref["Gs"])
                  and np.array_equal(o["ep_len"][i, : int(o["n_len"][i])].cpu().numpy(), ref["lengths"]))
            if not ok:
                print("FAIL", desc, "rollout", i, "steps", int(o["steps"][i]), ref["steps"], "cand", int(o["cand"][i]), ref["candidates"], "n_ep", ne, len(ref["Gs"]))
                bad += 1
                break
    if big:
        assert variants_rows == ran, "tables with segments above 65536 rows must run on the row-packed kernel too"
    summary = (f"{n_cases} cases drawn, {ran} run ({variants_rows} on the row-packed kernel), {rollouts} rollouts / {steps_checked} accepted steps compared, {bad} failures, {time.time() - t_start:.0f} s")
    if verbose:
        print(summary)
    return bad, ran, steps_checked


@pytest.mark.gpu
def test_randomised_parity_sweep_of_the_untraced_fast_path():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (no CPU fallback exists)")
    bad, ran, steps = sweep(int(os.environ.get("OFFSIM_FUZZ_CASES", "150")), int(os.environ.get("OFFSIM_FUZZ_SEED", "11")))
    assert bad == 0 and ran > 100 and steps > 1_000_000


@pytest.mark.gpu
def test_randomised_parity_sweep_with_states_of_more_than_65536_rows():
    """The same sweep over tables whose largest state holds 70 k .. 1 M rows (CartPole boxes, the grid's cells): they take the
    second stream layout (format B) through the same reset and scan kernels."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (no CPU fallback exists)")
    bad, ran, steps = sweep(int(os.environ.get("OFFSIM_FUZZ_BIG_CASES", "30")), int(os.environ.get("OFFSIM_FUZZ_SEED", "11")) + 1, big=True)
    assert bad == 0 and ran >= 20 and steps > 1_000_000


@pytest.mark.gpu
def test_randomised_queue_lengths_shuffle_equals_numpy_generator_shuffle():
    """Random queue lengths up to the LDS capacity (and a few beyond it, which take the global-memory variant), random seeds:
    the orders written by both forms of the sampler reset equal default_rng(seed).shuffle (oracle restatement, psrs.py:29-30)."""
    import torch
    from oracle import oracle as O
    from rl_offline_simulation_amd import synth
    from rl_offline_simulation_amd.table import TransitionTable
    from rl_offline_simulation_amd.evaluators import BatchedPSRS
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device (no CPU fallback exists)")
    g = np.random.default_rng(int(os.environ.get("OFFSIM_FUZZ_SEED", "11")))
    lengths = [int(x) for x in g.integers(1, 65537, 24)] + [int(x) for x in g.integers(1, 700, 12)] + [65535, 65536, 65537, 70001, 131073]
    # keyed chains above 32768 rows run as three launches cut at steps 16384 and 4096: lengths that leave the first launch one
    # step, exactly one group, one group and a step, ...
    lengths += [32769, 32768 + 63, 32768 + 64, 32768 + 65, 36864, 49152, 49153, 65472, 65473]
    for n in lengths:
        e = synth.synth_iid(n, 1, 2, seed=n)
        table = TransitionTable(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], e["steps"] == 0)
        seeds = [int(x) for x in g.integers(0, 1 << 62, 3)]
        plain = BatchedPSRS(table, len(seeds))
        plain.reset_sampler(seeds)
        perm = (plain.state.perm.to(torch.int64) & 0xFFFFFFFF).cpu().numpy()
        for k, sd in enumerate(seeds):
            assert np.array_equal(perm[k, :n], O.permutation(sd, n)), (n, sd)
        keyed = BatchedPSRS(table, len(seeds))
        keyed.reset_sampler(seeds, policy=table.policy_slots(synth.dirichlet_policy(1, 2)))
        assert torch.equal(keyed.perm.to(torch.int64) & 0xFFFFFFFF, plain.state.perm.to(torch.int64) & 0xFFFFFFFF), n


if __name__ == "__main__":
    os.environ.setdefault("OFFSIM_SCAN_ROWS", "1")  # (as tests/conftest.py: the sweep is about the row-packed kernel)
    b, _, _ = sweep(int(sys.argv[1]) if len(sys.argv) > 1 else 600, int(sys.argv[2]) if len(sys.argv) > 2 else 0, big=len(sys.argv) > 3)
    sys.exit(1 if b else 0)
