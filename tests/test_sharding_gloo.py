"""N > 1 path on CPU: two gloo ranks.  The HIP kernels cannot run here, so each rank produces its
share with the oracle (the checker standing in for the device) and the PRODUCT's host logic --
root slicing, the single all-reduce of partial sums, the ragged all_gather -- is what is tested."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.equation import GradDependentNonlinear, sample_points
        from oracle.mlp import PicardOracle
        from scasml_gp_amd import parallel
        d, B = 8, 11
        xt = np.concatenate(sample_points(np.random.default_rng(0), d, B - 3, 3))
        ora = PicardOracle(GradDependentNonlinear(d + 1), "quad", seed=2, stream=0)
        full = ora.uz_solve(2, 2, xt)
        # (i) Monte-Carlo sample sharding + one all-reduce
        from scasml_gp_amd import tables
        from scasml_gp_amd.solvers._picard import deal_units
        owner, _ = deal_units(tables.build_plan("quad", 2, 2, 0.5, True), world)     # the product's dealing of the units
        part = torch.from_numpy(ora.uz_solve(2, 2, xt, rank=rank, world=world, owner=owner))
        summed = parallel.allreduce_partial_sums(part)
        ok_samples = np.allclose(ora.finalize(summed.numpy()), full, atol=1e-12)
        # (ii) root sharding, ragged (11 roots over 2 ranks) + all_gather
        start, count = parallel.root_slice(B, rank, world)
        local = torch.from_numpy(ora.uz_solve(2, 2, xt[start:start + count], root0=start))
        gathered = parallel.gather_roots(local, [parallel.root_slice(B, r, world)[1] for r in range(world)])
        ok_roots = gathered.shape == (B, d + 1) and np.allclose(gathered.numpy(), full, atol=1e-12)
        q.put((rank, bool(ok_samples), bool(ok_roots)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sample_and_root_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True, True), (1, True, True)]


def test_root_slice_partitions_exactly():
    from scasml_gp_amd.parallel import root_slice
    for total in (0, 1, 7, 16, 1200):
        for world in (1, 2, 3, 8):
            parts = [root_slice(total, r, world) for r in range(world)]
            assert sum(c for _, c in parts) == total
            assert all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(world - 1))


def test_sample_units_counts_root_fan_out():
    from scasml_gp_amd import tables
    from scasml_gp_amd.parallel import sample_units
    # SURVEY.md 8(e) counts whole sample paths (27 + 5 + 3 + 2 = 37 units of very unequal cost); the units dealt here are the addends of a path's
    # NODES (q = 4, 3, 3 nodes per path at levels 0, 1, 2; a "+" and a "-" addend per node above level 0), which is what balances eight ranks
    assert sample_units(tables.build_plan("quad", 3, 3, 0.5, True)) == 27 + 5 * 4 + 2 * (3 * 3 + 2 * 3)
    assert sample_units(tables.build_plan("fh", 4, 3, 0.5, True)) == 81 + 81 + 2 * (27 + 9 + 3)
