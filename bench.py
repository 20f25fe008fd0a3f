#!/usr/bin/env python3
"""Headline benchmark: simulated steps/sec (node), 10 M logged transitions x 4096 rollouts.

One "step" = one full pass of the hot path over the batch of rollouts on every rank:
  PSRS.reset_sampler(seed_r) for all R seeds   (offsim_seed_streams + offsim_shuffle_queues)
  evalMC_psrs until the buffer is exhausted     (offsim_eval_mc)
with the logged-transition table already resident in HBM.  value = accepted steps of all rollouts on all
ranks / wall time (max over ranks).  Multi-GPU = weak scaling: every rank owns its own 10 M-transition shard
of the log (shards are episode-disjoint), runs all R seeds on it, and the per-seed (sum G, n episodes) pairs
are combined with one RCCL all-reduce.

Prints ONE JSON line on rank 0; see README/DESIGN.md for the fields `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--transitions", type=int, default=10_000_000, help="logged transitions per GPU")
    ap.add_argument("--rollouts", type=int, default=4096, help="sampler seeds (rollouts) per GPU")
    ap.add_argument("--workload", default="iid", choices=["iid", "cartpole", "grid"],
                    help="iid = S-iid synthetic log (headline); cartpole = CartPole dynamics + device box encoder (config C2); "
                         "grid = continuous_grid log + 2-64-25 MLP encoder forward on MFMA, random-init weights (config C3)")
    ap.add_argument("--n-states", type=int, default=162)
    ap.add_argument("--n-actions", type=int, default=2)
    ap.add_argument("--shuffle", default="per_rollout", choices=["per_rollout", "shared", "table_order"])
    ap.add_argument("--tile", type=int, default=4096, help="rollouts whose queue permutations are resident at once")
    ap.add_argument("--gamma", type=float, default=0.99)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for single-GPU plumbing tests)")
    ap.add_argument("--all-ranks-on-device0", action="store_true", help="plumbing test: several ranks share GPU 0 (use with --dist-backend gloo)")
    ap.add_argument("--cpu-sample-transitions", type=int, default=1_000_000)
    ap.add_argument("--cpu-sample-seconds", type=float, default=10.0)
    return ap.parse_args()


def cpu_baseline(e, pi, gamma, n_sample, budget_s):
    """The oracle (C port of the reference loop) on a bounded sample of the same workload, on all host cores: rollouts
    are independent, so each thread owns one oracle object and runs whole rollouts (ctypes releases the GIL)."""
    import threading
    from oracle import oracle as O
    sl = slice(0, n_sample)
    t0 = e["steps"][sl] == 0
    cols = (e["z"][sl], e["actions"][sl], e["rewards"][sl], e["z_next"][sl], e["terminals"][sl], e["action_distributions"][sl], t0)
    cores = max(1, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    O.lib()
    steps = [0] * cores
    rolls = [0] * cores
    t_start = time.perf_counter()

    def work(k, stride=None):
        ora = O.OraclePSRS(*cols)
        seed = k
        while True:
            ora.reset_sampler(seed)
            steps[k] += ora.evalmc(10 ** 9, pi, gamma)["steps"]
            rolls[k] += 1
            seed += cores if stride is None else stride
            if time.perf_counter() - t_start > budget_s or seed >= 4096:
                break

    # one core first (the reference itself is single-threaded), then all of them
    work(0, stride=1)
    el1 = time.perf_counter() - t_start
    v1, r1 = steps[0] / el1, rolls[0]
    steps[0] = rolls[0] = 0
    t_start = time.perf_counter()
    th = [threading.Thread(target=work, args=(k,)) for k in range(cores)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    el = time.perf_counter() - t_start
    return {"value": sum(steps) / el, "unit": "simulated steps/s", "cores": cores, "kind": "port", "value_one_core": v1,
            "sample": f"{sum(rolls)} rollouts (reset_sampler + evalMC to exhaustion) over the first {n_sample} transitions of the same "
                      f"synthetic log, {el:.1f} s of oracle/psrs_oracle.c on {cores} host threads (one rollout per thread); "
                      f"one thread alone: {r1} rollouts in {el1:.1f} s"}


def pmc_traffic(a):
    """Bytes per launch of the scan kernel from the committed rocprofv3 --pmc passes of this same command
    (profiles/, collected by tools/profile_bench.sh; PMC cannot be read from inside the run).  FETCH_SIZE/WRITE_SIZE
    in KB; 4-byte random-sector gathers are counted 1:1 on this GPU (tools/calib_fetch.py), and the counters sit on
    the fabric side of L2, i.e. they include Infinity-Cache hits: an upper bound of the HBM bytes."""
    default = (a.workload == "iid" and a.transitions == 10_000_000 and a.rollouts == 4096 and a.n_states == 162 and
               a.n_actions == 2 and a.shuffle == "per_rollout")
    path = os.path.join(ROOT, "profiles", "r01_rocprof_bench_10Mx4096", "summary.json")
    if not default or not os.path.exists(path):
        return None, None
    try:
        s = json.load(open(path))
        calls = [int(r["Calls"]) for r in s["kernel_stats"] if "k_eval_mc_win" in r["Name"]][0]
        kb = sum(v.get("FETCH_SIZE", 0.0) for k, v in s["pmc_fetch"].items() if "k_eval_mc_win" in k)
        kb += sum(v.get("WRITE_SIZE", 0.0) for k, v in s["pmc_write"].items() if "k_eval_mc_win" in k)
        return kb * 1024.0 / calls, "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (profiles/r01_rocprof_bench_10Mx4096), fabric-side: includes Infinity-Cache hits"
    except Exception:
        return None, None


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    from rl_offline_simulation_amd import _lib, synth
    from rl_offline_simulation_amd.table import TransitionTable
    from rl_offline_simulation_amd.evaluators import BatchedPSRS
    from rl_offline_simulation_amd.distributed import allreduce_estimates

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the PSRS engine has no CPU fallback")
    if a.all_ranks_on_device0:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if a.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(a.dist_backend)
    _lib.load()

    N, R = a.transitions, a.rollouts
    # this rank's shard of the log: shard g is generated from seed 20221107 + g (episode-disjoint by construction)
    if a.workload == "cartpole":
        from rl_offline_simulation_amd.encoders import CartpoleBoxEncoder
        e = synth.cartpole_log(N, seed=20221107 + rank)
        enc = CartpoleBoxEncoder()
        e["z"], e["z_next"] = enc.encode(e["observations"]), enc.encode(e["next_observations"])
        a.n_states, a.n_actions = 162, 2
    elif a.workload == "grid":
        from rl_offline_simulation_amd.encoders import HOMEREncoder
        e = synth.grid_coords_log_fast(N, seed=20221107 + rank)
        g = torch.Generator().manual_seed(0)  # nn.Linear's default init, fixed seed (no trained checkpoint travels)
        lin = lambda o, i: ((torch.rand((o, i), generator=g) * 2 - 1) / i ** 0.5, (torch.rand(o, generator=g) * 2 - 1) / i ** 0.5)
        (W1, b1), (W2, b2) = lin(64, 2), lin(25, 64)
        enc = HOMEREncoder(2, 5, 25, 64, state_dict={"obs_encoder.0.weight": W1, "obs_encoder.0.bias": b1,
                                                      "obs_encoder.2.weight": W2, "obs_encoder.2.bias": b2}, device=dev)
        t_e = time.perf_counter()
        e["z"], e["z_next"] = enc.encode(e["observations"]), enc.encode(e["next_observations"])
        encode_s = time.perf_counter() - t_e
        a.n_states, a.n_actions = 25, 5
    else:
        e = synth.synth_iid(N, a.n_states, a.n_actions, seed=20221107 + rank)
    pi = synth.dirichlet_policy(a.n_states, a.n_actions)
    t0 = e["steps"] == 0
    table = TransitionTable(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], e["action_distributions"], t0, device=dev)
    pi_slots = table.policy_slots(pi)
    seeds = np.arange(R, dtype=np.uint64)
    tile = min(a.tile, R)
    envs = {}

    def env_for(n):
        if n not in envs:
            envs[n] = BatchedPSRS(table, n)
        return envs[n]

    acc = {k: torch.zeros(R, dtype=dt, device=dev) for k, dt in (("sum_g", torch.float64), ("n_ep", torch.int64),
                                                                   ("steps", torch.int64), ("cand", torch.int64))}
    ev = []  # (kind, start, stop) HIP events on the launch stream

    def one_pass(record):
        for b in range(0, R, tile):
            sd = seeds[b:b + tile]
            env = env_for(len(sd))
            if record:
                e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                e0.record()
            env.reset_sampler(sd, a.shuffle, shuffle_seed=1234)
            if record:
                e1.record()
            o = env.eval_mc(pi_slots, a.gamma)
            if record:
                e2.record()
                ev.append(("reset_sampler", e0, e1))
                ev.append(("scan", e1, e2))
            for k in acc:
                acc[k][b:b + len(sd)] = o[k]
        est = torch.stack([acc["sum_g"], acc["n_ep"].to(torch.float64)], dim=1)
        if world > 1:
            allreduce_estimates(est)  # one RCCL all-reduce of [R,2] f64 (64 KiB at R = 4096)
        return est

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        one_pass(False)
    barrier()
    t_start = time.perf_counter()
    for _ in range(a.steps):
        est = one_pass(True)
    barrier()
    elapsed = time.perf_counter() - t_start
    el_t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    steps_t = torch.stack([acc["steps"].sum(), acc["cand"].sum()]).to(torch.float64)
    if world > 1:
        dist.all_reduce(el_t, op=dist.ReduceOp.MAX)
        dist.all_reduce(steps_t, op=dist.ReduceOp.SUM)
    elapsed = float(el_t[0])
    steps_pass, cand_pass = float(steps_t[0]), float(steps_t[1])  # all ranks, one pass

    if rank == 0:
        t_scan = sum(s.elapsed_time(t) for k, s, t in ev if k == "scan") * 1e-3
        t_reset = sum(s.elapsed_time(t) for k, s, t in ev if k == "reset_sampler") * 1e-3
        n_scan = sum(1 for k, _, _ in ev if k == "scan")
        my_steps, my_cand = float(acc["steps"].sum()), float(acc["cand"].sum())
        b_c = table.bytes_per_candidate + (4 if a.shuffle != "table_order" else 0)  # + permutation index
        b_s = table.bytes_per_step
        alg_bytes_pass = my_cand * b_c + my_steps * b_s
        achieved = alg_bytes_pass * a.steps / t_scan / 1e9
        value = steps_pass * a.steps / elapsed
        vest = (est[:, 0] / est[:, 1]).cpu().numpy()
        traffic, traffic_src = pmc_traffic(a)
        out = {
            "metric": "simulated steps/sec (node), 10M logged transitions x 4096 rollouts",
            "value": value, "unit": "simulated steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"CartPole-dynamics log, uniform logger, device box encoder (C2), {N} transitions" if a.workload == "cartpole" else
                                   f"continuous_grid log, uniform logger, 2-64-25 MLP encoder on MFMA with random-init weights (C3), {N} transitions, host-to-z encode {encode_s:.2f} s" if a.workload == "grid" else f"S-iid synthetic log (SURVEY 8d), {N} transitions per GPU x {R} rollouts, nS={a.n_states}, nA={a.n_actions}, "
                                   f"evalMC_psrs to exhaustion, gamma={a.gamma}"), "transitions_per_gpu": N, "rollouts": R,
                       "shuffle": a.shuffle, "rollout_tile": tile, "p_log": "f32", "sharding": f"log sharded by episode over {world} GPU(s), "
                       "all seeds on every shard, RCCL all-reduce of per-seed (sum G, n episodes)"},
            "candidates_per_s": cand_pass * a.steps / elapsed, "acceptance": steps_pass / max(cand_pass, 1.0),
            "buffer_consumed_frac": cand_pass / (world * R * N),
            "value_estimate_mean": float(np.nanmean(vest)),
            "scan_only_steps_per_s": my_steps * a.steps / t_scan, "reset_sampler_s_per_pass": t_reset / a.steps,
            "scan_s_per_pass": t_scan / a.steps,
            "roofline": {"bound": "hbm", "kernel": "k_eval_mc_win (offsim_eval_mc_keys)", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": alg_bytes_pass / max(n_scan / a.steps, 1),
                         "bytes_per_candidate": b_c, "bytes_per_step": b_s, "launches": n_scan,
                         "avg_launch_ms": t_scan / max(n_scan, 1) * 1e3},
        }
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(e, pi, a.gamma, min(a.cpu_sample_transitions, N), a.cpu_sample_seconds)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
