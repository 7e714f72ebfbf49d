"""Multi-rank path on CPU (gloo, world size 2): cells shard across ranks with no data-path
collective; each rank's sub-domain result must equal the reference's run on that sub-domain
(goldens `shard_*`, tests/golden/gen_golden.py:gen_shards), diagnostics are gathered per cell."""
import os
import socket
import warnings

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from pysdm_amd import Builder, Formulae, sharding
from pysdm_amd.dynamics.collisions import Coalescence, Golovin
from pysdm_amd.environments import Box, Mesh

from .trajectory import GOLDEN, compare, snapshot


def test_cell_blocks_cover_the_domain():
    for n_cell, world in ((16, 2), (1024, 8), (15, 4), (3, 8)):
        blocks = [sharding.cell_block(n_cell, r, world) for r in range(world)]
        assert blocks[0][0] == 0 and blocks[-1][1] == n_cell
        assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
        sizes = [b - a for a, b in blocks]
        assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, errors):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from oracle.backend import OracleBackend  # pylint: disable=import-outside-toplevel

        full = np.load(os.path.join(GOLDEN, "traj_multicell_golovin_4x4.npz"))
        gold = np.load(os.path.join(GOLDEN, f"shard_golovin_4x4_r{rank}of{world}.npz"))
        n_cell = int(full["cfg"][5])
        attributes = {"volume": full["init/volume"], "multiplicity": full["init/multiplicity"],
                      "cell id": full["init/cell_id"]}
        local, mine, (first, last) = sharding.shard_attributes(attributes, n_cell, rank, world)
        np.testing.assert_array_equal(mine, gold["global_indices"])
        env = Box(dt=float(full["cfg"][3]), dv=float(full["cfg"][4]))
        env.mesh = Mesh((last - first,), size=(float(last - first),))
        env.mesh.dv = float(full["cfg"][4])
        builder = Builder(n_sd=len(mine), backend=OracleBackend(Formulae(seed=int(full["cfg"][1]))),
                          environment=env)
        dynamic = Coalescence(collision_kernel=Golovin(b=1.5e3), adaptive=bool(full["cfg"][2]))
        builder.add_dynamic(dynamic)
        particulator = builder.build(local)
        dynamic = particulator.dynamics["Collision"]
        for step in (1, 3, 10):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                particulator.run(step - particulator.n_steps)
            compare(snapshot(particulator, dynamic), gold, step)
        # per-cell diagnostics of the whole domain, assembled on every rank
        rates = sharding.gather_per_cell(dynamic.coalescence_rate.to_ndarray(), n_cell, world)
        expected = np.concatenate([
            np.load(os.path.join(GOLDEN, f"shard_golovin_4x4_r{r}of{world}.npz"))[
                "step10/coalescence_rate"] for r in range(world)])
        np.testing.assert_array_equal(rates, expected)
        total = sharding.global_sum(particulator.attributes.super_droplet_count)
        assert int(total) == sum(
            int(np.load(os.path.join(GOLDEN, f"shard_golovin_4x4_r{r}of{world}.npz"))[
                "step10/length"]) for r in range(world))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as exc:  # pylint: disable=broad-except
        errors.put(f"rank {rank}: {exc!r}")
        raise


@pytest.mark.timeout(300)
def test_two_rank_sharded_run_matches_reference_subdomains():
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    ctx = mp.get_context("spawn")
    errors = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, errors)) for r in range(2)]
    for proc in procs:
        proc.start()
    for proc in procs:
        proc.join(240)
    failed = [p.exitcode for p in procs if p.exitcode != 0]
    messages = []
    while not errors.empty():
        messages.append(errors.get())
    assert not failed and not messages, f"{failed} {messages}"


def test_bench_cpu_baseline_leg_runs():
    """bench.py's `cpu_baseline` (the oracle timed on the host) on a tiny box"""
    import importlib.util  # pylint: disable=import-outside-toplevel

    spec = importlib.util.spec_from_file_location(
        "bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                              "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for workload, adaptive, n_sd in (("shima", False, 2**12), ("kinematic2d", True, 2**17)):
        result = bench.cpu_baseline(workload, n_sd, adaptive, seconds_budget=0.3)
        assert result["kind"] == "port" and result["cores"] == 1 and result["value"] > 0
