"""Multi-GPU: one process per GPU, torch.distributed over RCCL ("nccl" backend) / gloo on CPU.

The path shards naturally: rollouts are independent given the read-only table, and the log can
be split into episode-disjoint shards.  Each rank evaluates every seed on its own shard; the only exchange
is one all-reduce(SUM) of the per-seed pairs (sum of returns, number of episodes) -- 64 KiB at R = 4096,
latency-bound over xGMI.  The per-seed value estimate is the episode-weighted mean sum(G) / sum(n).
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_rollouts(n_rollouts, rank, world):
    """Contiguous slice of rollout ids for `rank` (primary partition: rollouts, table replicated)."""
    per = -(-n_rollouts // world)
    lo = min(rank * per, n_rollouts)
    return lo, min(lo + per, n_rollouts)


def shard_episodes(episode_ids, rank, world):
    """Row mask of the episode-disjoint log shard of `rank`: whole episodes, round-robin by episode id, so a rollout
    never crosses shards (second partition axis, C4)."""
    return (np.asarray(episode_ids) % world) == rank


def allreduce_estimates(est):
    """In-place SUM all-reduce of est[R,2] = (sum of returns, n episodes) per seed across ranks: on the device tensor itself
    (backend "nccl" = RCCL over xGMI; gloo stages a device tensor through the host).  A process group of one rank still runs
    the collective (legal for RCCL, and the only way a one-GPU box executes this leg); without a process group it is a no-op."""
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(est, op=dist.ReduceOp.SUM)
    return est


def combine_value(est):
    """Per-seed value estimate from the reduced pairs: sum(G) / n (NaN where a seed finished no episode)."""
    return est[:, 0] / est[:, 1]
