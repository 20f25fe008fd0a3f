#!/usr/bin/env python3
"""Headline benchmark: simulated steps/sec (node), 10 M logged transitions x 4096 rollouts.

One "step" = one full pass of the hot path over the batch of rollouts on every rank:
  PSRS.reset_sampler(seed_r) for all R seeds   (offsim_seed_streams + offsim_shuffle_queues[_keys])
  evalMC_psrs until the buffer is exhausted     (offsim_eval_mc[_keys])
  one all-reduce of the per-seed (sum G, n episodes) pairs when there is more than one rank
with the logged-transition table already resident in HBM.  value = accepted steps of all rollouts on all
ranks / wall time (max over ranks).  (The fold of the evaluated policy into per-row keys -- k_compile_policy, 86 us at
10 M rows -- is cached per policy object and therefore runs in the first pass only; every timed pass reuses it, as a
caller evaluating one policy under thousands of seeds would.)

Multi-GPU (`--gpus N`): one process per GPU.  When the ranks do not exist yet (no WORLD_SIZE in the environment)
this script starts them itself -- N child processes with RANK / LOCAL_RANK / WORLD_SIZE set, before anything in
this process has touched the GPU -- otherwise (torchrun / torch.distributed.run) it is one of the ranks.
  --scaling strong (default): the SAME 10 M-transition log at every N, split into N episode-disjoint shards, all R
      seeds on every shard, one RCCL all-reduce of [R,2] f64 (second axis).  Also measured and reported as
      the extra field `rollout_sharded`: table replicated, R/N seeds per GPU (primary axis).
  --scaling weak: every rank owns its own 10 M-transition log (N x 10 M in total), all R seeds on each.

Prints ONE JSON line on rank 0; see DESIGN.md section 5 for `roofline`, `cpu_baseline`, `parity_check`.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PARITY_SEEDS = (0, 1, 2047, 4095)
_REAL_STDOUT = sys.stdout
_T0 = time.perf_counter()


def _phase(msg):
    """OFFSIM_BENCH_TRACE=1: wall-clock stamps of the command's phases on stderr (where its host time goes)."""
    if os.environ.get("OFFSIM_BENCH_TRACE"):
        sys.stderr.write(f"[bench +{time.perf_counter() - _T0:7.2f} s] {msg}\n")
        sys.stderr.flush()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--transitions", type=int, default=10_000_000, help="logged transitions (whole job with --scaling strong, per GPU with weak)")
    ap.add_argument("--rollouts", type=int, default=4096, help="sampler seeds (rollouts)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"])
    ap.add_argument("--no-rollout-sharded", action="store_true", help="skip the extra rollout-sharded measurement of a strong multi-GPU run")
    ap.add_argument("--workload", default="iid", choices=["iid", "cartpole", "grid", "obs128"],
                    help="iid = S-iid synthetic log (headline; --transitions 12500000 = one GPU's shard of config C4); cartpole = CartPole "
                         "dynamics + device box encoder (config C2); grid = continuous_grid log + 2-64-25 MLP encoder forward on MFMA (config C3); "
                         "obs128 = 128-d fp16 observations -> 128-64-50 encoder forward on MFMA -> fp16 p_log table (config C5; one GPU's share "
                         "of its 50 M rows is --transitions 6250000)")
    ap.add_argument("--n-states", type=int, default=162)
    ap.add_argument("--n-actions", type=int, default=2)
    ap.add_argument("--shuffle", default="per_rollout", choices=["per_rollout", "shared", "table_order"])
    ap.add_argument("--tile", type=int, default=0, help="rollouts whose queue orders are resident at once (0 = as many as the free HBM of the "
                                                       "rank holds: 6 bytes per queue position as candidate streams, 4 as permutations)")
    ap.add_argument("--gamma", type=float, default=0.99)
    ap.add_argument("--rng", default="pcg64", choices=["pcg64", "philox"],
                    help="rejection stream provider: pcg64 = NumPy's default_rng (the reference's numbers, the headline); philox = rocRAND's "
                         "Philox4x32-10 device engine in the same row-packed scan (another, equally valid sample path; the oracle's parity "
                         "check replays the same stream; not the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-check", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for single-GPU plumbing tests)")
    ap.add_argument("--all-ranks-on-device0", action="store_true", help="plumbing test: several ranks share GPU 0 (use with --dist-backend gloo)")
    ap.add_argument("--cpu-sample-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-threads-cap", type=int, default=64, help="host threads of the CPU baseline (each holds its own queues: 170 MB at 10 M rows)")
    ap.add_argument("--force-dist", action="store_true", help="create the process group and run the all-reduce of the per-seed estimates even with ONE "
                                                              "rank (legal for RCCL): the only way a one-GPU box executes the collective leg")
    ap.add_argument("--no-configs", action="store_true", help="skip the extra BASELINE configurations (C2 CartPole 1 M x 4096, C3 continuous_grid 10 M x 4096 "
                                                              "with the MLP encoder, one GPU's shard of C4 12.5 M x 4096, one GPU's shard of C5 6.25 M x 4096 "
                                                              "with 128-d fp16 observations) that the default single-GPU headline run appends under \"configs\"")
    ap.add_argument("--print-csrc-digest", action="store_true", help="print the digest of the kernel sources (recorded by tools/profile_bench.sh) and exit")
    return ap.parse_args()


def csrc_digest():
    """Digest of everything that is compiled into liboffsim_hip.so: ties a committed rocprofv3 profile to the kernels it measured."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "rl-offline-simulation_amd", "csrc")
    for f in sorted(os.listdir(d)) + ["../../include/offsim.h"]:
        if f.endswith((".hip", ".hpp", ".h", ".sh")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


# ---------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` with no ranks around yet
# ---------------------------------------------------------------------------------------------------
def spawn_ranks(a):
    """Starts a.gpus copies of this command, one per GPU, and waits for them.  Nothing here touches the GPU (children are
    started with subprocess from a process that never initialised HIP), rank 0's stdout is this process's stdout."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        pending = set(range(a.gpus))
        while pending:
            for r in list(pending):
                c = procs[r].poll()
                if c is not None:
                    pending.discard(r)
                    if c != 0:
                        rc = rc or c
                        for q in pending:  # one rank failed: the others would wait in a collective forever
                            procs[q].terminate()
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


# ---------------------------------------------------------------------------------------------------
# CPU legs (the oracle is the checker and the reported baseline; never the product path)
# ---------------------------------------------------------------------------------------------------
def oracle_for(e):
    from oracle import oracle as O
    O.lib()
    p = e["action_distributions"]
    if p.dtype == np.float16:  # (C5: the oracle consumes the fp16-rounded values)
        p = p.astype(np.float64)
    return O.OraclePSRS(e["z"], e["actions"], e["rewards"], e["z_next"], e["terminals"], p, e["steps"] == 0)


class OracleRun:
    """The oracle on a few seeds of the SAME table the timed passes run on, started on host threads as soon as the log exists -- it needs
    nothing of the GPU's, so it works beside the timed passes instead of after them (round 6: the default command spent 15 s of host time
    here, one configuration after the other) -- and compared with what the GPU returned once both are done (`check`)."""

    def __init__(self, e, pi, gamma, seeds, shuffle="per_rollout", shuffle_seed=1234, rng="pcg64"):
        import threading
        self.seeds, self.res, self.base, self.err = [int(s) for s in seeds], [None] * len(seeds), None, None

        def work(k):
            o = self.base.clone()
            if shuffle == "shared":  # one queue order for all rollouts, per-rollout rejection streams (psrs.py:20 is a plain attribute)
                o.reset_sampler(int(shuffle_seed))
                o.set_rejection_seed(self.seeds[k])
            else:
                o.reset_sampler(self.seeds[k])
            if rng == "philox":  # env.rejection_sampling_rng = a replay of rocRAND's Philox4x32-10 of the same seed (tests/golden/philox_*.npz)
                o.set_rejection_philox(self.seeds[k])
            self.res[k] = o.evalmc(10 ** 9, pi, gamma)

        def lead():
            try:
                self.base = oracle_for(e)
                th = [threading.Thread(target=work, args=(k,)) for k in range(len(self.seeds))]
                for x in th:
                    x.start()
                for x in th:
                    x.join()
            except BaseException as ex:  # (re-raised by check)
                self.err = ex

        self.thread = threading.Thread(target=lead)
        self.thread.start()

    def check(self, got):
        """Accepted steps, candidates examined and completed episodes equal, value estimate within 1e-5 (BASELINE.json north_star)."""
        self.thread.join()
        if self.err is not None:
            raise self.err
        ok, worst, rows = True, 0.0, []
        for k, s in enumerate(self.seeds):
            ref = self.res[k]
            n_ep = len(ref["Gs"])
            v_ref = float(ref["Gs"].mean()) if n_ep else float("nan")
            g = got[int(s)]
            v = g["sum_g"] / g["n_ep"] if g["n_ep"] else float("nan")
            err = abs(v - v_ref) if n_ep else 0.0
            same = (g["steps"] == ref["steps"] and g["cand"] == ref["candidates"] and g["n_ep"] == n_ep and err <= 1e-5)
            ok &= bool(same)
            worst = max(worst, err)
            rows.append({"seed": int(s), "steps": ref["steps"], "candidates": ref["candidates"], "episodes": n_ep, "value": v_ref, "match": bool(same)})
        return {"seeds": self.seeds, "ok": bool(ok), "max_abs_value_err": worst, "tolerance": 1e-5, "table_rows": int(self.base.N),
                "checked": "accepted steps, candidates examined, completed episodes equal; |sum_g/n_ep - mean(Gs)| <= 1e-5; oracle/psrs_oracle.c, "
                           "one full rollout per seed on this rank's table (host threads beside the timed passes)", "oracle": rows}


def cpu_baseline(base, pi, gamma, budget_s, cap):
    """The oracle (C port of the reference loop) on the same table, bounded by time: whole rollouts (reset_sampler + evalMC
    to exhaustion), first on one thread (the reference is single-threaded), then one rollout per host thread."""
    import threading
    cores = max(1, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    cores = min(cores, cap)
    one = base.clone()
    t0 = time.perf_counter()
    s1 = r1 = 0
    while True:
        one.reset_sampler(r1)
        s1 += one.evalmc(10 ** 9, pi, gamma)["steps"]
        r1 += 1
        if time.perf_counter() - t0 > budget_s / 3:
            break
    el1 = time.perf_counter() - t0
    objs = [one] + [base.clone() for _ in range(cores - 1)]
    steps, rolls = [0] * cores, [0] * cores
    t_start = time.perf_counter()

    def work(k):
        seed = 100 + k
        while True:
            objs[k].reset_sampler(seed)
            steps[k] += objs[k].evalmc(10 ** 9, pi, gamma)["steps"]
            rolls[k] += 1
            seed += cores
            if time.perf_counter() - t_start > budget_s * 2 / 3:
                break

    th = [threading.Thread(target=work, args=(k,)) for k in range(cores)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    el = time.perf_counter() - t_start
    return {"value": sum(steps) / el, "unit": "simulated steps/s", "cores": cores, "kind": "port", "value_one_core": s1 / el1,
            "sample": f"{sum(rolls)} whole rollouts (reset_sampler + evalMC to exhaustion) over the SAME {base.N}-transition table as the "
                      f"timed passes, {el:.1f} s of oracle/psrs_oracle.c on {cores} host threads (one rollout per thread); "
                      f"one thread alone: {r1} rollouts in {el1:.1f} s"}


def pmc_traffic(a, kernel, world):
    """HBM-side bytes per launch of the dominant kernel from the rocprofv3 --pmc passes committed under profiles/ (PMC cannot
    be read from inside the run).  Used only when that profile was taken with the kernel sources of THIS build
    (summary.json records their digest) and with this command; otherwise null."""
    default = (a.workload == "iid" and a.transitions == 10_000_000 and a.rollouts == 4096 and a.n_states == 162 and
               a.n_actions == 2 and a.shuffle == "per_rollout" and world == 1 and not os.environ.get("OFFSIM_BENCH_TEST_SCALE"))
    if not default:
        return None, "no PMC profile for this command"
    dig = csrc_digest()
    pdir = os.path.join(ROOT, "profiles")
    for d in sorted(os.listdir(pdir), reverse=True) if os.path.isdir(pdir) else []:
        path = os.path.join(pdir, d, "summary.json")
        if not os.path.exists(path):
            continue
        try:
            s = json.load(open(path))
            if s.get("csrc_digest") != dig:
                continue
            calls = [int(r["Calls"]) for r in s["kernel_stats"] if kernel in r["Name"]][0]
            kb = sum(v.get("FETCH_SIZE", 0.0) for k, v in s["pmc_fetch"].items() if kernel in k)
            kb += sum(v.get("WRITE_SIZE", 0.0) for k, v in s["pmc_write"].items() if kernel in k)
            return kb * 1024.0 * s.get("fetch_scale", 1.0) / calls, (
                f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command with these kernel sources (profiles/{d}, csrc digest {dig}); "
                "fabric-side of L2: includes Infinity-Cache hits")
        except Exception:
            continue
    return None, f"no committed PMC profile matches the kernel sources of this build (csrc digest {dig})"


# ---------------------------------------------------------------------------------------------------
def make_log(a, seed, dev):
    """The logged experience (host arrays, OfflineDataset schema + z / z_next) of this workload."""
    import torch
    from rl_offline_simulation_amd import synth
    N = a.transitions
    note = ""
    t_gen = time.perf_counter()
    if a.workload == "cartpole":
        from rl_offline_simulation_amd.encoders import CartpoleBoxEncoder
        e = synth.cartpole_log(N, seed=seed)
        enc = CartpoleBoxEncoder()
        t_e = time.perf_counter()
        e["z"], e["z_next"] = enc.encode(e["observations"]), enc.encode(e["next_observations"])
        a.encode_s = time.perf_counter() - t_e
        a.n_states, a.n_actions = 162, 2
    elif a.workload == "grid":
        from rl_offline_simulation_amd.encoders import HOMEREncoder
        e = synth.grid_coords_log_fast(N, seed=seed)
        # no trained checkpoint travels: weights that send an observation to its cell (synth.grid_cell_encoder_weights), so that all
        # 25 abstract states are populated as under a trained HOMER encoder
        W1, b1, W2, b2 = (torch.from_numpy(w) for w in synth.grid_cell_encoder_weights(5, 64, seed=0))
        enc = HOMEREncoder(2, 5, 25, 64, state_dict={"obs_encoder.0.weight": W1, "obs_encoder.0.bias": b1,
                                                      "obs_encoder.2.weight": W2, "obs_encoder.2.bias": b2}, device=dev)
        t_e = time.perf_counter()
        e["z"], e["z_next"] = enc.encode(e["observations"]), enc.encode(e["next_observations"])
        a.encode_s = time.perf_counter() - t_e
        note = f", host-to-z encode {a.encode_s:.2f} s"
        a.n_states, a.n_actions = 25, 5
    elif a.workload == "obs128":
        e, note = make_obs128_log(a, N, seed, dev)
    else:
        e = synth.synth_iid(N, a.n_states, a.n_actions, seed=seed)
        a.encode_s = 0.0
    a.generate_s = time.perf_counter() - t_gen - a.encode_s
    return e, note


def make_obs128_log(a, N, seed, dev):
    """Config C5's shape: observations N(0,1)-coded [N,128] fp16, encoder 128-64-50 on the device (HOMEREncoder.encode's
    forward + arg-max, encoders/homer.py:159-168 of the reference, csrc/encode_mfma.hpp here), logging probabilities kept fp16 in HBM.
    The S-iid generator makes the log's columns (nS = 50, nA = 4); an observation is the 128-d code of its logged state plus 0.3 x N(0,1)
    noise, generated and encoded on the device (the observations never exist on the host: 2 x N x 256 B), and z / z_next are what the
    ENCODER says -- the oracle's parity run consumes the same encoded states and the fp16-rounded probabilities.  No trained checkpoint
    travels: the weights are built so that hidden unit k answers to code k and latent k to hidden unit k (a trained HOMER encoder
    populates its latent states; random-init weights collapse onto a few), stated in the line.  Sets a.encode_s (HIP events around the two
    forwards) and a.encoder (rows, bytes per row, the arg-max against the oracle's MLP on a sample)."""
    import torch
    from rl_offline_simulation_amd import synth
    from rl_offline_simulation_amd.encoders import HOMEREncoder
    dO, H, nZ, nA = 128, 64, 50, 4
    e = synth.synth_iid(N, nZ, nA, seed=seed)
    g = np.random.default_rng(seed + 77)
    code = g.standard_normal((nZ, dO)).astype(np.float32)
    W1 = np.zeros((H, dO), np.float32)
    W1[:nZ] = code / (code * code).sum(1, keepdims=True)
    W1[nZ:] = g.standard_normal((H - nZ, dO)).astype(np.float32) / np.sqrt(dO) * 0.1
    b1 = np.zeros(H, np.float32)
    W2 = np.zeros((nZ, H), np.float32)
    W2[np.arange(nZ), np.arange(nZ)] = 1.0
    W2[:, nZ:] = g.standard_normal((nZ, H - nZ)).astype(np.float32) * 0.01
    b2 = np.zeros(nZ, np.float32)
    enc = HOMEREncoder(dO, nA, nZ, H, state_dict={"obs_encoder.0.weight": torch.from_numpy(W1), "obs_encoder.0.bias": torch.from_numpy(b1),
                                                  "obs_encoder.2.weight": torch.from_numpy(W2), "obs_encoder.2.bias": torch.from_numpy(b2)}, device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed + 78)
    code_d = torch.from_numpy(code).to(dev)

    def observe(z):  # [N,128] fp16 on the device, in blocks (the f32 intermediate of 6.25 M rows would be 3.2 GB)
        zt = torch.from_numpy(np.ascontiguousarray(z)).to(dev)
        out = torch.empty((N, dO), dtype=torch.float16, device=dev)
        for b in range(0, N, 1 << 20):
            zz = zt[b:b + (1 << 20)]
            out[b:b + (1 << 20)] = (code_d[zz] + 0.3 * torch.randn((zz.numel(), dO), device=dev, generator=gen)).to(torch.float16)
        return out

    obs, nobs = observe(e["z"]), observe(e["z_next"])
    enc.encode_device(obs[:1024])  # (warm-up: module load)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0.record()
    z_d, zn_d = enc.encode_device(obs), enc.encode_device(nobs)
    t1.record()
    torch.cuda.synchronize()
    a.encode_s = t0.elapsed_time(t1) * 1e-3
    z_true = e["z"]
    e["z"], e["z_next"] = z_d.cpu().numpy().astype(np.int64), zn_d.cpu().numpy().astype(np.int64)
    e["action_distributions"] = e["action_distributions"].astype(np.float16)
    # the arg-max against the oracle's MLP (oracle/psrs_oracle.c: mlp_encode, f32 like the reference's torch forward) on a sample of
    # rows whose top-2 logits are 1e-4 apart or more
    from oracle import oracle as O
    k = min(N, 4096)
    zo, lo = O.mlp_encode(obs[:k].cpu().numpy().astype(np.float32), W1, b1, W2, b2)
    top2 = np.sort(lo, axis=1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 1e-4
    enc_ok = bool(np.array_equal(e["z"][:k][clear], zo[clear]))
    bytes_row = dO * 2 + 4
    a.encoder = {"shape": "128-64-50, fp16 observations, f32 weights (bf16 x 3 exact split on MFMA)", "rows": 2 * N, "encode_s": a.encode_s,
                 "rows_per_s": 2 * N / a.encode_s, "bytes_per_row": bytes_row,
                 "roofline": {"bound": "hbm", "achieved": 2 * N * bytes_row / a.encode_s / 1e9, "peak": 8000.0, "unit": "GB/s",
                              "frac": 2 * N * bytes_row / a.encode_s / 8e12},
                 "argmax_equals_oracle_mlp": enc_ok, "argmax_rows_checked": int(clear.sum()),
                 "latent_equals_logged_state_frac": float((e["z"] == z_true).mean()),
                 "weights": "built so that hidden unit k answers to the 128-d code of state k and latent k to hidden unit k (no trained checkpoint travels)"}
    if not enc_ok:
        raise SystemExit("bench.py: the device encoder's arg-max differs from the oracle's MLP on rows with a clear arg-max -- no number is reported")
    a.n_states, a.n_actions = nZ, nA
    del obs, nobs
    torch.cuda.empty_cache()
    return e, f", encoder forward {a.encode_s * 1e3:.1f} ms for 2 x {N} rows"


def take_rows(e, mask):
    return {k: (v[mask] if isinstance(v, np.ndarray) and v.shape[:1] == mask.shape else v) for k, v in e.items()}


def run(a):
    import torch
    import torch.distributed as dist
    from rl_offline_simulation_amd import _lib, synth
    from rl_offline_simulation_amd.table import TransitionTable
    from rl_offline_simulation_amd.evaluators import BatchedPSRS
    from rl_offline_simulation_amd.distributed import allreduce_estimates, shard_episodes, shard_rollouts

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the PSRS engine has no CPU fallback")
    if a.all_ranks_on_device0:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_pg = world > 1 or a.force_dist
    if use_pg:
        if world == 1:  # (a one-rank group needs no launcher: rendezvous with itself)
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                sk = socket.socket()
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
                sk.close()
        if a.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(a.dist_backend, rank=rank, world_size=world)
    lib = _lib.load()
    import ctypes as C

    R = a.rollouts
    strong = a.scaling == "strong"
    # the default single-GPU headline command also reports the other BASELINE configurations that fit one GPU (C2, C3)
    headline = (a.workload == "iid" and a.transitions == 10_000_000 and R == 4096 and a.n_states == 162 and a.n_actions == 2 and
                a.shuffle == "per_rollout" and a.rng == "pcg64" and world == 1)
    # OFFSIM_BENCH_TEST_SCALE = k (tests only): the same code path with every log k times shorter; the line says so (`test_scale`)
    test_scale = max(1, int(os.environ.get("OFFSIM_BENCH_TEST_SCALE", "1")))
    a.transitions //= test_scale
    # strong: every rank derives its shard from the same log; weak: shard g is its own log, generated from seed 20221107 + g
    _phase("imports done, device set")
    e_full, enc_note = make_log(a, 20221107 + (0 if strong else rank), dev)
    _phase("headline log generated")
    if strong and world > 1:
        e = take_rows(e_full, shard_episodes(e_full["episode_ids"], rank, world))
    else:
        e = e_full
    pi = synth.dirichlet_policy(a.n_states, a.n_actions)
    seeds = np.arange(R, dtype=np.uint64)
    want_parity = rank == 0 and not a.no_parity_check and a.shuffle != "table_order"  # (the reference has no unshuffled mode)
    ora_main = OracleRun(e, pi, a.gamma, [s for s in PARITY_SEEDS if s < R], a.shuffle, 1234, rng=a.rng) if want_parity else None

    def barrier():
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(e_rank, seed_lo, seed_hi, n_warm, n_timed, fill_full, pi=pi, diag=False):
        """n_warm + n_timed passes of seeds[seed_lo:seed_hi] over the table of e_rank.  Returns the timing, the per-kernel HIP-event
        times and the per-seed results of the last pass."""
        import gc
        gc.collect()  # (whatever the configuration before this one left behind -- resident queue orders of up to 250 GB -- goes back to the
        torch.cuda.empty_cache()  # driver before this one sizes its tile by the free memory)
        torch.cuda.synchronize()
        t_ing = time.perf_counter()
        table = TransitionTable(e_rank["z"], e_rank["actions"], e_rank["rewards"], e_rank["z_next"], e_rank["terminals"],
                                e_rank["action_distributions"], e_rank["steps"] == 0, device=dev)
        torch.cuda.synchronize()
        t_ing = time.perf_counter() - t_ing  # host columns -> device, group-by-state, gathers (once per log; outside the timed passes)
        pi_slots = table.policy_slots(pi)
        sd_all = seeds[seed_lo:seed_hi]
        n_loc = len(sd_all)
        # Rollouts resident at once: every rollout keeps its queue orders (candidate streams: 4 + 2 bytes per queue position; as
        # permutations 4), its init order, cursors and stream state in HBM.  All of them when they fit -- the chains of a tile run
        # concurrently, tiles run one after the other -- otherwise as many as the free memory of this rank holds.
        # (room is left for the chunked shuffle's workspace -- chains of more than 65536 rows --, at most a tenth of what is free: the
        # same rule as the library's own driver, evaluators/psrs.py: resident_rollouts)
        from rl_offline_simulation_amd.evaluators.psrs import resident_rollouts
        fit, per_rollout, free_b, total_b = resident_rollouts(table, keyed=a.shuffle == "per_rollout")
        if a.shuffle != "per_rollout":
            fit = n_loc
        tile = max(1, min(a.tile if a.tile > 0 else n_loc, n_loc, max(fit, 1)))
        # tiles of ONE size (the last one is filled up with repeats of its last seed, whose results are dropped): a second batch size
        # would be a second set of resident buffers
        n_tiles = -(-n_loc // tile)
        tile = -(-n_loc // n_tiles)
        assert a.shuffle != "per_rollout" or tile * per_rollout <= free_b, (tile, per_rollout, free_b)
        resident = tile * per_rollout if a.shuffle == "per_rollout" else per_rollout
        envs = {}

        def env_for(n):
            if n not in envs:
                envs[n] = BatchedPSRS(table, n)
            return envs[n]

        acc = {k: torch.zeros(n_loc, dtype=dt, device=dev) for k, dt in (("sum_g", torch.float64), ("n_ep", torch.int64),
                                                                           ("steps", torch.int64), ("cand", torch.int64))}
        local = {}
        ev = []  # (kind, start, stop) HIP events on the launch stream (torch's current stream is the stream handed to the C ABI)

        def one_pass(record):
            for b in range(0, n_loc, tile):
                sd = sd_all[b:b + tile]
                n_real = len(sd)
                if n_real < tile:
                    sd = np.concatenate([sd, np.repeat(sd[-1:], tile - n_real)])
                env = env_for(len(sd))
                if record:
                    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                    e0.record()
                env.reset_sampler(sd, a.shuffle, shuffle_seed=1234, policy=pi_slots, rejection=a.rng)
                if record:
                    e1.record()
                o = env.eval_mc(pi_slots, a.gamma)
                if "_kernel" in o:
                    local["kernel"] = o["_kernel"]  # (the generic kernel: a Philox job on a table the row-packed scan does not take)
                if record:
                    e2.record()
                    ev.append(("reset_sampler", e0, e1))
                    ev.append(("scan", e1, e2))
                for k in acc:
                    acc[k][b:b + n_real] = o[k][:n_real]
            if fill_full:  # rollout-sharded: every rank owns a slice of the seeds, the all-reduce assembles the [R,2] table
                est = torch.zeros((R, 2), dtype=torch.float64, device=dev)
                est[seed_lo:seed_hi, 0] = acc["sum_g"]
                est[seed_lo:seed_hi, 1] = acc["n_ep"].to(torch.float64)
            else:
                est = torch.stack([acc["sum_g"], acc["n_ep"].to(torch.float64)], dim=1)
            if use_pg:
                local["est"] = est.clone() if world == 1 else None
                allreduce_estimates(est)  # one RCCL all-reduce of [R,2] f64 (64 KiB at R = 4096), on the device tensor
            return est

        _phase(f"table ingested ({table.N} rows), buffers allocated")
        for _ in range(n_warm):
            one_pass(False)
        barrier()
        _phase("warm-up passes done")
        t_start = time.perf_counter()
        for _ in range(n_timed):
            est = one_pass(True)
        barrier()
        elapsed = time.perf_counter() - t_start
        _phase(f"{n_timed} timed passes done")
        el_t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        tot = torch.stack([acc["steps"].sum(), acc["cand"].sum()]).to(torch.float64)
        if use_pg:
            dist.all_reduce(el_t, op=dist.ReduceOp.MAX)
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        t_scan = sum(s.elapsed_time(t) for k, s, t in ev if k == "scan") * 1e-3
        t_reset = sum(s.elapsed_time(t) for k, s, t in ev if k == "reset_sampler") * 1e-3
        res = dict(elapsed=float(el_t[0]), steps_pass=float(tot[0]), cand_pass=float(tot[1]), t_scan=t_scan, t_reset=t_reset,
                   n_scan=sum(1 for k, _, _ in ev if k == "scan"), my_steps=float(acc["steps"].sum()), my_cand=float(acc["cand"].sum()),
                   est=est, acc={k: v.cpu().numpy() for k, v in acc.items()}, b_c=table.bytes_per_candidate, b_s=table.bytes_per_step,
                   rows=table.N, tile=tile, variant=local.get("kernel") or env_for(tile).scan_variant(), seg=(table.min_seg, table.max_seg),
                   resident=int(resident), hbm_free=int(free_b), hbm_total=int(total_b), ingest_s=t_ing, est_local=local.get("est"))
        _lib.check_async_faults()  # (the barrier synchronised: no kernel of these passes gave up a bounded wait)
        # bytes one pass of the sampler reset writes: every rollout's queue orders (6 / 5 bytes per position as candidate streams, 4 as
        # permutations) and its init order
        res["reset_bytes_pass"] = float(n_loc) * (per_rollout - table.n_slots * 4 - 64) if a.shuffle == "per_rollout" else float(per_rollout)
        res["plog_dtype"] = str(table.p_log.dtype).replace("torch.", "")
        # Untimed, after the timed region: one more pass of the last tile with the kernel's own clock read-out (eval_mc(dbg=True): shader
        # cycles and 100 MHz ticks of every chain wavefront, top-up and dry-row counters) -- what explains the roofline fraction of a
        # kernel that is bound by the latency of its dependent chains, not by HBM.  Row-packed kernel only.
        res["chain"] = None
        if diag and res["variant"] == "k_eval_mc_rows" and n_timed > 0:
            sd = sd_all[-tile:] if n_loc >= tile else np.concatenate([sd_all, np.repeat(sd_all[-1:], tile - n_loc)])
            env = env_for(len(sd))
            env.reset_sampler(sd, a.shuffle, shuffle_seed=1234, policy=pi_slots, rejection=a.rng)
            o = env.eval_mc(pi_slots, a.gamma, dbg=True)
            torch.cuda.synchronize()
            raw = o["dbg"].cpu().numpy()
            cyc, rt = raw[:, 2].astype(float), raw[:, 3].astype(float)
            its = float(o["steps"].max().item())
            res["chain"] = {"cycles_per_iteration": float(cyc.mean() / max(its, 1.0)), "iterations": its, "shader_clock_GHz": float((cyc / np.maximum(rt, 1.0)).mean() * 0.1),
                            "slowest_chain_ms": float(rt.max() * 1e-5), "mean_chain_ms": float(rt.mean() * 1e-5),
                            "rows_without_clear_accept_per_rollout": float((raw[:, 0] & 0xffffffff).mean()),
                            "top_ups_per_rollout": float((raw[:, 0] >> 32).mean()), "top_ups_late_per_rollout": float(((raw[:, 1] >> 16) & 0xffffff).mean()),
                            "exact_looks_per_rollout": float((raw[:, 1] & 0xffff).mean())}
            _lib.check_async_faults()
        envs.clear()
        env = o = table = None  # (locals of this frame that still reach the resident buffers)
        gc.collect()
        torch.cuda.empty_cache()
        return res

    m = measure(e, 0, R, a.warmup, a.steps, False, diag=True)
    extra = None
    if strong and world > 1 and not a.no_rollout_sharded:
        lo, hi = shard_rollouts(R, rank, world)
        x = measure(e_full, lo, hi, 1, 1, True)
        extra = {"value": x["steps_pass"] / x["elapsed"], "unit": "simulated steps/s", "ms_per_step": x["elapsed"] * 1e3,
                 "steps": 1, "warmup": 1, "rollouts_per_gpu": hi - lo, "transitions_per_gpu": int(x["rows"]),
                 "scan_s_per_pass": x["t_scan"], "reset_sampler_s_per_pass": x["t_reset"],
                 "what": "same job split the other way (SURVEY 8e primary axis): the whole log replicated on every GPU, R/N seeds per GPU, "
                         "all-reduce assembles the [R,2] table"}

    coll = None
    if use_pg:  # the collective alone: the [R,2] f64 all-reduce on the device tensor, HIP events around 20 of them
        x = m["est"].clone()
        allreduce_estimates(x)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        c0.record()
        for _ in range(20):
            allreduce_estimates(x)
        c1.record()
        torch.cuda.synchronize()
        coll = {"backend": dist.get_backend(), "world": world, "device_tensor": bool(x.is_cuda), "bytes": int(x.numel() * 8),
                "allreduce_us": c0.elapsed_time(c1) * 1e3 / 20,
                "librccl_mapped": any("librccl" in ln for ln in open("/proc/self/maps")),
                "unchanged_at_one_rank": (bool(torch.equal(m["est"], m["est_local"])) if world == 1 and m["est_local"] is not None else None)}

    def extra_config(name, workload, transitions, subshards=1):
        """One more BASELINE configuration on this GPU, same binary, same R: log generation (host), encoder forward (device), ingest,
        1 warm-up + 2 timed passes, and its own parity check (the oracle on four seeds of the same table).
        subshards = K > 1: the log is cut into K episode-disjoint parts (shard_episodes, the partition of SURVEY 8(e)) that this GPU
        evaluates one after the other, all R seeds on each; a pass is the K parts' passes, the per-seed estimate the episode-weighted mean
        over the parts -- the estimator of the multi-GPU run, one level further down.  Every part has its own oracle parity check."""
        import copy
        b = copy.copy(a)
        b.workload, b.transitions = workload, transitions
        b.encoder = None
        _phase(f"configuration {name}: start")
        e_c, _ = make_log(b, 20221107, dev)
        _phase(f"configuration {name}: log generated")
        pi_c = synth.dirichlet_policy(b.n_states, b.n_actions)
        if subshards > 1:
            parts = [take_rows(e_c, shard_episodes(e_c["episode_ids"], k, subshards)) for k in range(subshards)]
            oras = [OracleRun(p, pi_c, a.gamma, [sd for sd in PARITY_SEEDS if sd < R], a.shuffle, 1234) if not a.no_parity_check else None for p in parts]
            xs = [measure(p, 0, R, 1, 2, False, pi=pi_c, diag=(k == 0)) for k, p in enumerate(parts)]
            b_c = xs[0]["b_c"] + 4
            t_scan, t_reset, elapsed = (sum(x[k] for x in xs) for k in ("t_scan", "t_reset", "elapsed"))
            steps_pass, cand_pass = sum(x["steps_pass"] for x in xs), sum(x["cand_pass"] for x in xs)
            alg = sum(x["my_cand"] * b_c + x["my_steps"] * x["b_s"] for x in xs)
            reset_bytes = sum(x["reset_bytes_pass"] for x in xs)
            out_c = {"workload": f"{workload}, {transitions} transitions x {R} rollouts, nS={b.n_states}, nA={b.n_actions}",
                     "sharding": f"this GPU's {transitions} rows as {subshards} episode-disjoint parts of {[int(x['rows']) for x in xs]} rows, evaluated one after the "
                                 f"other, all {R} seeds on each; per-seed estimate = episode-weighted mean over the parts (SURVEY 8(e)'s estimator): states of "
                                 f"{xs[0]['seg'][1]} rows at most shuffle LDS-resident in stream format A, where the undivided shard's {transitions // b.n_states}-row "
                                 "states take the chunked shuffle and format C (profiles/r05_bench_c4_shard_12.5M_band1.json: reset 0.88 s, scan 1.17 s)",
                     "value": steps_pass * 2 / elapsed, "unit": "simulated steps/s", "ms_per_step": elapsed / 2 * 1e3, "scan_s": t_scan / 2, "reset_s": t_reset / 2,
                     "kernel": xs[0]["variant"], "rollout_tile": xs[0]["tile"], "acceptance": steps_pass / max(cand_pass, 1.0),
                     "buffer_consumed_frac": cand_pass / (R * transitions), "segment_rows_min_max": [min(x["seg"][0] for x in xs), max(x["seg"][1] for x in xs)],
                     "roofline": {"bound": "hbm", "achieved": alg * 2 / t_scan / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": alg * 2 / t_scan / 1e9 / 8000.0,
                                  "bytes_per_candidate": b_c, "bytes_per_step": xs[0]["b_s"]},
                     "roofline_reset": {"bound": "hbm", "achieved": reset_bytes * 2 / t_reset / 1e9, "peak": 8000.0, "unit": "GB/s",
                                        "frac": reset_bytes * 2 / t_reset / 8e12, "bytes_written_per_pass": reset_bytes},
                     "p_log": xs[0]["plog_dtype"], "queue_orders_resident_bytes": max(x["resident"] for x in xs),
                     "log_generate_s_host": b.generate_s, "encode_s": b.encode_s, "ingest_s": sum(x["ingest_s"] for x in xs)}
            if xs[0]["chain"]:
                out_c["chain"] = xs[0]["chain"]
            if not a.no_parity_check:
                _phase(f"configuration {name}: GPU passes done, waiting for the oracle")
                pcs = [o.check({sd: {k: (float(v[sd]) if k == "sum_g" else int(v[sd])) for k, v in x["acc"].items()} for sd in o.seeds}) for o, x in zip(oras, xs)]
                _phase(f"configuration {name}: parity checked")
                out_c["parity_ok"] = all(pc["ok"] for pc in pcs)
                out_c["parity_max_abs_value_err"] = max(pc["max_abs_value_err"] for pc in pcs)
                if not out_c["parity_ok"]:
                    sys.stderr.write(json.dumps(pcs) + "\n")
                    raise SystemExit(f"bench.py: configuration {name}: the GPU results differ from the oracle on a part of this table -- no number is reported")
            return out_c
        ora_c = OracleRun(e_c, pi_c, a.gamma, [sd for sd in PARITY_SEEDS if sd < R], a.shuffle, 1234) if not a.no_parity_check else None
        x = measure(e_c, 0, R, 1, 2, False, pi=pi_c, diag=True)
        b_c = x["b_c"] + 4
        alg = x["my_cand"] * b_c + x["my_steps"] * x["b_s"]
        out_c = {"workload": f"{workload}, {transitions} transitions x {R} rollouts, nS={b.n_states}, nA={b.n_actions}", "value": x["steps_pass"] * 2 / x["elapsed"],
                 "unit": "simulated steps/s", "ms_per_step": x["elapsed"] / 2 * 1e3, "scan_s": x["t_scan"] / 2, "reset_s": x["t_reset"] / 2,
                 "kernel": x["variant"], "rollout_tile": x["tile"], "acceptance": x["steps_pass"] / max(x["cand_pass"], 1.0),
                 "buffer_consumed_frac": x["cand_pass"] / (R * transitions), "segment_rows_min_max": list(x["seg"]),
                 "roofline": {"bound": "hbm", "achieved": alg * 2 / x["t_scan"] / 1e9, "peak": 8000.0, "unit": "GB/s",
                              "frac": alg * 2 / x["t_scan"] / 1e9 / 8000.0, "bytes_per_candidate": b_c, "bytes_per_step": x["b_s"]},
                 "roofline_reset": {"bound": "hbm", "achieved": x["reset_bytes_pass"] * 2 / x["t_reset"] / 1e9, "peak": 8000.0, "unit": "GB/s",
                                    "frac": x["reset_bytes_pass"] * 2 / x["t_reset"] / 8e12, "bytes_written_per_pass": x["reset_bytes_pass"]},
                 "p_log": x["plog_dtype"], "queue_orders_resident_bytes": x["resident"],
                 "log_generate_s_host": b.generate_s, "encode_s": b.encode_s, "ingest_s": x["ingest_s"]}
        if x["chain"]:
            out_c["chain"] = x["chain"]
        if getattr(b, "encoder", None):
            out_c["encoder"] = b.encoder
        if ora_c is not None:
            got = {sd: {k: (float(v[sd]) if k == "sum_g" else int(v[sd])) for k, v in x["acc"].items()} for sd in ora_c.seeds}
            _phase(f"configuration {name}: GPU passes done, waiting for the oracle")
            pc = ora_c.check(got)
            _phase(f"configuration {name}: parity checked")
            out_c["parity_ok"] = pc["ok"]
            out_c["parity_max_abs_value_err"] = pc["max_abs_value_err"]
            if not pc["ok"]:
                sys.stderr.write(json.dumps(pc) + "\n")
                raise SystemExit(f"bench.py: configuration {name}: the GPU results differ from the oracle on this table -- no number is reported")
        return out_c

    configs = None
    if headline and not a.no_configs:
        configs = {"C2": extra_config("C2", "cartpole", 1_000_000 // test_scale), "C3": extra_config("C3", "grid", 10_000_000 // test_scale),
                   # one GPU's shard of C4 (100 M transitions over 8 GPUs: 12.5 M rows x all 4096 seeds), evaluated as two episode-disjoint
                   # halves (round 6: their ~38 k-row states shuffle LDS-resident; `--workload iid --transitions 12500000` alone is the
                   # undivided shard: chunked shuffle, stream format C)
                   "C4_shard": extra_config("C4_shard", "iid", 12_500_000 // test_scale, subshards=2),
                   # one GPU's share of C5 (50 M rows of 128-d fp16 observations over 8 GPUs: 6.25 M rows x 4096 seeds, encoder on MFMA, fp16 p_log)
                   "C5_shard": extra_config("C5_shard", "obs128", 6_250_000 // test_scale)}

    if rank == 0:
        elapsed, t_scan, t_reset, n_scan = m["elapsed"], m["t_scan"], m["t_reset"], m["n_scan"]
        b_c = m["b_c"] + (4 if a.shuffle != "table_order" else 0)  # + queue-order index (mode A)
        b_s = m["b_s"]
        alg_bytes_pass = m["my_cand"] * b_c + m["my_steps"] * b_s
        achieved = alg_bytes_pass * a.steps / t_scan / 1e9
        value = m["steps_pass"] * a.steps / elapsed
        est = m["est"]
        vest = (est[:, 0] / est[:, 1]).cpu().numpy()
        kernel = m["variant"]
        traffic, traffic_src = pmc_traffic(a, kernel, world)
        n_rank = int(m["rows"])
        shard_txt = ("one GPU" if world == 1 else
                     f"the {a.transitions}-transition log split into {world} episode-disjoint shards ({n_rank} rows on rank 0), all {R} seeds on every shard, "
                     "one RCCL all-reduce of per-seed (sum G, n episodes)" if strong else
                     f"weak: {world} logs of {a.transitions} transitions, one per GPU, all {R} seeds on each, one RCCL all-reduce of per-seed (sum G, n episodes)")
        wl = (f"CartPole-dynamics log, uniform logger, device box encoder (C2)" if a.workload == "cartpole" else
              f"continuous_grid log, uniform logger, 2-64-25 MLP encoder on MFMA, weights built to map an observation to its cell (C3){enc_note}" if a.workload == "grid" else
              f"128-d fp16 observations, 128-64-50 MLP encoder on MFMA (weights built to decode the observation's state code), fp16 p_log (C5){enc_note}" if a.workload == "obs128" else
              f"S-iid synthetic log (SURVEY 8d), nS={a.n_states}, nA={a.n_actions}")
        out = {
            "metric": "simulated steps/sec (node), 10M logged transitions x 4096 rollouts",
            "value": value, "unit": "simulated steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{wl}, {a.transitions} transitions {'in total' if strong else 'per GPU'} x {R} rollouts, evalMC_psrs to exhaustion, gamma={a.gamma}",
                       "transitions": a.transitions, "transitions_on_rank0": n_rank, "rollouts": R, "shuffle": a.shuffle, "rollout_tile": m["tile"],
                       "queue_orders_resident_bytes_rank0": m["resident"], "hbm_free_bytes_rank0": m["hbm_free"], "hbm_total_bytes_rank0": m["hbm_total"],
                       "p_log": m["plog_dtype"], "rng": a.rng, "sharding": shard_txt, "segment_rows_min_max": list(m["seg"])},
            "candidates_per_s": m["cand_pass"] * a.steps / elapsed, "acceptance": m["steps_pass"] / max(m["cand_pass"], 1.0),
            "buffer_consumed_frac": m["cand_pass"] / (R * a.transitions * (1 if strong else world)),
            "value_estimate_mean": float(np.nanmean(vest)),
            "scan_only_steps_per_s": m["my_steps"] * a.steps / t_scan, "reset_sampler_s_per_pass": t_reset / a.steps,
            "scan_s_per_pass": t_scan / a.steps,
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": alg_bytes_pass / max(n_scan / a.steps, 1),
                         "bytes_per_candidate": b_c, "bytes_per_step": b_s, "launches": n_scan,
                         "avg_launch_ms": t_scan / max(n_scan, 1) * 1e3, "rank": 0,
                         "traffic_ratio": (traffic / (alg_bytes_pass / max(n_scan / a.steps, 1))) if traffic else None},
            # the sampler reset beside it: bytes it writes (queue orders as candidate streams / permutations + init orders) / its time
            "roofline_reset": {"bound": "hbm", "kernels": "k_shuffle_wave (+ k_shuffle_chunked)", "achieved": m["reset_bytes_pass"] * a.steps / max(t_reset, 1e-12) / 1e9,
                               "peak": 8000.0, "unit": "GB/s", "frac": m["reset_bytes_pass"] * a.steps / max(t_reset, 1e-12) / 8e12,
                               "bytes_written_per_pass": m["reset_bytes_pass"],
                               "bound_measured": "instruction issue of the single classifier / applier wavefront of each Fisher-Yates chain (LDS-resident; the only HBM traffic is the coalesced write-out)"},
        }
        if world > 1:
            out["estimator"] = ("episode-weighted mean over %d shards: per seed (sum of the shards' sum G) / (sum of the shards' completed episodes) -- SURVEY 8(e); "
                                "NOT the single-table estimate of the N = 1 line (`value_estimate_mean` differs by construction)" % world) if strong else \
                               ("per seed, the mean over %d independent logs' episodes (weak scaling: one log per GPU)" % world)
        tb = os.path.join(ROOT, "profiles", "r06_pmc_traffic_breakdown", "traffic_breakdown.json")
        if traffic and os.path.exists(tb):  # (a claim from committed PMC passes, like `traffic` itself: what the 4 x over-fetch consists of)
            out["roofline"]["traffic_breakdown"] = json.load(open(tb))
        if m["chain"]:
            # why `frac` is what it is: the scan is 4096 exact dependent chains, a chain wavefront retires one accepted step of its four
            # rollouts per iteration, and the kernel lasts (iterations) x (cycles per iteration) / clock -- measured by the kernel's own clock
            # in one extra untimed pass.  step_floor_cycles: the hand-scheduled step alone beside a busy helper wavefront
            # (tools/micro/step_loop.hip, profiles/r04_step_loop_microbench.txt); the rest is tick, dry rows, episode ends.
            out["roofline"].update({"bound_measured": "chain latency / instruction issue of one in-order wavefront per four rollouts",
                                    "cycles_per_iteration": m["chain"]["cycles_per_iteration"], "step_floor_cycles": 185, "chain": m["chain"]})
        if getattr(a, "encoder", None):
            out["encoder"] = a.encoder
        if extra:
            out["rollout_sharded"] = extra
        if coll:
            out["collective"] = coll
        if configs:
            out["configs"] = configs
        out["ingest_s"] = m["ingest_s"]
        if test_scale > 1:
            out["test_scale"] = test_scale
        base = None
        if ora_main is not None:
            got = {s: {k: (float(v[s]) if k == "sum_g" else int(v[s])) for k, v in m["acc"].items()} for s in ora_main.seeds}
            _phase("headline: waiting for the oracle")
            out["parity_check"] = ora_main.check(got)
            _phase("headline: parity checked")
            base = ora_main.base
            if not out["parity_check"]["ok"]:
                sys.stderr.write(json.dumps(out["parity_check"]) + "\n")
                raise SystemExit("bench.py: the GPU results differ from the oracle on this table -- no number is reported")
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(base or oracle_for(e), pi, a.gamma, a.cpu_sample_seconds, a.cpu_threads_cap)
        _phase("CPU baseline done")
        _REAL_STDOUT.write(json.dumps(out) + "\n")
        _REAL_STDOUT.flush()
    if use_pg:
        dist.barrier()
        dist.destroy_process_group()


def main():
    a = parse()
    if a.print_csrc_digest:
        print(csrc_digest())
        return 0
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and a.gpus > 1:
        return spawn_ranks(a)  # before any GPU call in this process
    # stdout carries exactly one JSON line: whatever libraries print there (gloo's connection banner, ...) goes to stderr
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if env_world is not None and int(env_world) != a.gpus:
        sys.stderr.write(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={env_world}; start one rank per GPU (or leave WORLD_SIZE unset and let "
                         "bench.py start them)\n")
        return 2
    run(a)
    return 0


if __name__ == "__main__":
    sys.exit(main())
