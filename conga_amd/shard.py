"""Chromosome -> GPU partition and the end-of-job result gather.

Chromosomes are fully independent in the reference (per-chromosome loop, bam_data.c:269-339), so
the job shards by chromosome with no data-path collective; the only exchange is one gather of the
fixed-size result records to rank 0, which writes the output files in annotation order.
torch.distributed is plumbing here: backend "nccl" is RCCL over xGMI on MI355X, "gloo" on CPU.
"""
import numpy as np


def lpt_partition(costs, n_ranks):
    """Longest-processing-time-first: returns owner[i] for every unit.  Deterministic."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0.0] * n_ranks
    owner = [0] * len(costs)
    for i in order:
        r = min(range(n_ranks), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += costs[i]
    return owner


def unit_cost(chrom_len, n_intervals, sum_interval_len=None):
    """Cost model of one chromosome: the depth pass scales with L, the interval passes with sum(l)."""
    if sum_interval_len is None:
        sum_interval_len = 8000.0 * n_intervals
    return float(chrom_len) + 2.0 * float(sum_interval_len)


def gather_records(local_bytes, counts_per_rank, rank, world_size, device=None):
    """One padded gather of result records to rank 0.

    local_bytes: uint8 torch tensor (this rank's records, packed; on `device`).
    counts_per_rank: bytes each rank contributes (known to every rank from the partition).
    Returns on rank 0 a list of uint8 tensors (one per rank, trimmed); None elsewhere.
    """
    import torch
    import torch.distributed as dist

    pad = max(max(counts_per_rank), 1)
    dev = local_bytes.device if device is None else device
    send = torch.zeros(pad, dtype=torch.uint8, device=dev)
    send[:local_bytes.numel()] = local_bytes
    if world_size == 1:
        return [send[:counts_per_rank[0]]]
    recv = [torch.empty(pad, dtype=torch.uint8, device=dev) for _ in range(world_size)] if rank == 0 else None
    dist.gather(send, recv, dst=0)
    if rank != 0:
        return None
    return [recv[r][:counts_per_rank[r]] for r in range(world_size)]


def records_from_bytes(t, dtype):
    """uint8 tensor -> numpy structured array (host copy)."""
    return np.frombuffer(t.cpu().numpy().tobytes(), dtype=dtype)
