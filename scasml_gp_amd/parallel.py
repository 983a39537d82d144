"""Multi-GPU host logic: one process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm;
"gloo" in the CPU tests).

Two shardings of the Picard solve:

* ``roots``   -- evaluation points are independent; each rank solves a contiguous slice with
                 ``root0`` = its global offset (Philox counters are keyed by the GLOBAL root index,
                 so the result does not depend on the rank count).  No data-path collective; an
                 optional all_gather returns the full result everywhere.  This is what bench.py uses.
* ``samples`` -- the Monte-Carlo units of the ROOT call (terminal samples, then per node (m, k) of every
                 level's sample paths its "+" and "-" addends with their subtrees) are dealt to ranks by cost
                 (scasml_plan_deal_units); each rank produces un-clipped partial sums of shape (B, 1+d) and ONE
                 all-reduce(sum) over xGMI combines them, followed by the clip of MLP.py:272-274.  The reference
                 has no counterpart (single device).
"""


def root_slice(total, rank, world):
    """Contiguous split of ``total`` roots: (start, count) for ``rank``; earlier ranks take the remainder."""
    base, rem = divmod(int(total), int(world))
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count


def sample_units(plan):
    """Number of shardable units of the root call: terminal samples + the addends of the NODES (l, m, k) of every level's sample paths
    ("+" with the level-l subtree; "-" with the level-(l-1) subtree where l > 0)."""
    n = plan.n
    return int(plan.mg[n]) + sum(int(plan.term[n][l].mc) * int(plan.term[n][l].q) * (2 if l else 1) for l in range(n))


def allreduce_partial_sums(partial, group=None):
    """Sum the ranks' partial (B, 1+d) estimators in place -- the single collective of the path."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(partial, op=dist.ReduceOp.SUM, group=group)
    return partial


def gather_roots(local, counts, group=None):
    """all_gather root-sharded results (ragged first dimension) into the full batch on every rank."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    mx = max(counts)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    outs = [torch.empty_like(pad) for _ in counts]
    dist.all_gather(outs, pad, group=group)
    return torch.cat([o[:c] for o, c in zip(outs, counts)], dim=0)


def solve_sharded(solver, n, par, x_t, mode="roots", group=None, gather=True):
    """Run ``solver`` (an MLP / ScaSML object of this package) on the calling rank's share.

    mode="roots":   x_t is the FULL batch on every rank; returns the full (B, 1+d) result if
                    ``gather`` else the local slice.
    mode="samples": every rank holds the full batch and 1/world of the Monte-Carlo units; returns
                    the all-reduced, clipped (B, 1+d) result on every rank.
    """
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    eng = solver._engine
    stream_id = eng.calls
    eng.calls += 1
    if mode == "roots":
        start, count = root_slice(len(x_t), rank, world)
        uz, _, _ = eng.solve(n, par, x_t[start:start + count], root0=start, stream_id=stream_id)
        if not gather:
            return uz
        return gather_roots(uz, [root_slice(len(x_t), r, world)[1] for r in range(world)], group)
    if mode == "samples":
        uz, _, _ = eng.solve(n, par, x_t, rank=rank, world=world, stream_id=stream_id)
        if world == 1:
            return uz
        return eng.finalize_partials(allreduce_partial_sums(uz, group))
    raise ValueError("mode must be 'roots' or 'samples'")
