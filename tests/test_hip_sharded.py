"""The sharded mode on HIP: two processes (one card - the GPU box has one -, rank-to-rank traffic
over gloo; on a node the same code runs one process per GPU over RCCL), each computing half of
the cells.  The state put together from the owners equals (i) the UNSHARDED reference goldens of
the 4 x 4 grids, (ii) the reference's digest of the 32 x 32 grid, (iii) the one-process HIP run
where super-droplets die and the permutation has to be put together across the processes, (iv) the
reference's golden of displacement + collisions, where super-droplets change owner every step."""
import os
import socket
import warnings

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from pysdm_amd import sharding

from . import digests
from .test_sharding import (BEYOND_THE_CELL_KERNELS, CASES, sharded_box_equals_single,
                            sharded_run_equals_golden)

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, errors):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from pysdm_amd import cases  # pylint: disable=import-outside-toplevel
        from pysdm_amd.engine import HipEngine  # pylint: disable=import-outside-toplevel

        engine = HipEngine.get(0)

        def stage(text):  # (visible with pytest -s: which stage a stuck run was in)
            print(f"[sharded worker {rank}] {text}", flush=True)

        for name in CASES:
            stage(name)
            sharded_run_equals_golden(name, engine, rank, world)
        for box in BEYOND_THE_CELL_KERNELS:  # (cells of 8192: the generic kernels; global croupier)
            stage(f"beyond the per-cell kernels: {box}")
            sharded_box_equals_single(engine, rank, world, **box)
        stage("32 x 32 digest")
        # 32 x 32 cells against the reference's digest
        digests.check("kinematic2d_64percell", engine,
                      prepare=lambda runner: sharding.attach(runner, rank, world),
                      snapshot=sharding.gather)
        # deaths (compaction + re-sort run replicated after the exchange of the dead positions)
        for adaptive in (True, False):
            stage(f"deaths, adaptive={adaptive}")
            single = cases.make_box(engine, "shima", n_sd=2**13, adaptive=adaptive, dt=200.0,
                                    thin=0.02, grid=(4, 4))
            shard = cases.make_box(engine, "shima", n_sd=2**13, adaptive=adaptive, dt=200.0,
                                   thin=0.02, grid=(4, 4))
            sharding.attach(shard, rank, world)
            for steps in (1, 4, 3):
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    single.run(steps)
                    shard.run(steps)
                got, ref = sharding.gather(shard), single.snapshot()
                length = int(ref["length"])
                assert int(got["length"]) == length < 2**13
                for key, value in ref.items():
                    if key == "stats_dt_min" and not adaptive:
                        continue
                    mine = got[key]
                    if key == "idx":
                        value, mine = value[:length], mine[:length]
                    np.testing.assert_array_equal(mine, value, err_msg=f"{adaptive} {key}")
            assert shard.shard.calls[2] > 0
        # the same with other grids / cell sizes / seeds, several steps per call (launch-ahead with
        # the gate closing on a death, the working copy under sharding, exchanges of dead positions)
        for k, (grid, n_sd, chunks) in enumerate((((8, 5), 40000, (5,)), ((3, 8), 2**16, (2, 2, 1)),
                                                  ((5, 4), 20000, (5, 8, 2)), ((4, 4), 2**13, (3, 4)),
                                                  ((8, 5), 40000, (4, 4)), ((5, 4), 20000, (7,)))):
            stage(f"deaths, grid {grid}, {n_sd} super-droplets, steps {chunks}")
            options = dict(n_sd=n_sd, adaptive=True, dt=200.0, thin=0.02, grid=grid,
                           seed=3000 + k, optimized_random=bool(k % 2))
            single, shard = cases.make_box(engine, "shima", **options), cases.make_box(
                engine, "shima", **options)
            sharding.attach(shard, rank, world)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                for steps in chunks:
                    single.run(steps)
                    shard.run(steps)
            got, ref = sharding.gather(shard), single.snapshot()
            length = int(ref["length"])
            assert int(got["length"]) == length < n_sd
            for key, value in ref.items():
                mine = got[key]
                if key == "idx":
                    value, mine = value[:length], mine[:length]
                np.testing.assert_array_equal(mine, value, err_msg=f"{grid} {key}")
            assert shard.sub_steps_done == single.sub_steps_done and shard.offset == single.offset
        # replicated displacement on the completed state + sharded collisions (migration between
        # the processes' cells every step) against the reference's golden
        from . import displacement_cases  # pylint: disable=import-outside-toplevel

        stage("displacement + collisions")
        displacement_cases.run_case("disp2d_collide", engine, shard=(rank, world))
        # the displacement step sharded as well (sdm_displacement_step_sharded): every process
        # moves its own super-droplets, rows and positions of those that change owner cross the
        # processes, nothing the size of a column does
        for name in displacement_cases.CASES:
            stage(f"sharded displacement: {name}")
            moved = displacement_cases.run_case(name, engine, shard=(rank, world),
                                                owner_moves=True)
            stats = moved.shard_stats
            assert stats["moved"] > 0 and stats["calls"] > 0, stats
            if name != "disp1d_implicit_sed":
                assert stats["removed"] > 0 and stats["left"] + stats["arrived"] > 0, stats
        stage("a population that dies out")
        stats = displacement_cases.random_flow_pair_equal(engine, rank, world,
                                                          **displacement_cases.DIES_OUT)
        assert stats["live"] == 0, stats
        stage("sharded flow at 2^18 beside the one-process run")
        stats = displacement_cases.sharded_flow_equals_single(engine, rank, world, n_sd=2**18,
                                                              grid=(16, 16), steps=5)
        assert stats["left"] > 0 and stats["arrived"] > 0 and stats["removed"] > 0, stats
        stage(f"done {stats}")
        dist.barrier()
        dist.destroy_process_group()
    except Exception as exc:  # pylint: disable=broad-except
        errors.put(f"rank {rank}: {exc!r}")
        raise


@pytest.mark.timeout(900)
def test_two_processes_sharded_on_hip_equal_the_unsharded_reference(hip_engine):  # pylint: disable=unused-argument
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    ctx = mp.get_context("spawn")
    errors = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, errors)) for r in range(2)]
    for proc in procs:
        proc.start()
    for proc in procs:
        proc.join(800)
    failed = [p.exitcode for p in procs if p.exitcode != 0]
    messages = []
    while not errors.empty():
        messages.append(errors.get())
    assert not failed and not messages, f"{failed} {messages}"


def test_the_library_issues_the_collectives_itself_over_rccl(hip_engine):
    """sdm_comm_unique_id / sdm_comm_init: with a communicator the exchanges of a sharded step are
    ncclAllReduce calls on the library's stream - the host's callback is not called (here it would
    fail the step if it were).  World size 1 (a box of this pool has one GPU): no peer, but the
    whole device path; the run equals the unsharded one, deaths included."""
    import ctypes

    from pysdm_amd import abi, cases

    def box():
        return cases.make_box(hip_engine, "shima", n_sd=2**13, adaptive=True, dt=200.0, thin=0.02,
                              grid=(4, 4))

    ident = np.zeros(abi.COMM_ID_BYTES, dtype=np.uint8)
    hip_engine.call("sdm_comm_unique_id", ident)
    assert ident.any()
    hip_engine.call("sdm_comm_init", ident, 0, 1)
    stats = (ctypes.c_int64 * 8)()
    hip_engine.call("sdm_ctx_read_stats", stats, 1)
    try:
        plain, sharded = box(), box()
        sharded.shard = sharding.RecordingShard(hip_engine, 2**13, 16)  # (no process group needed)
        sharded.read_back = True
        sharded.counts_global_pairs = True

        def refuse(*_):
            return 1

        sharded.shard.callback = abi.ExchangeFn(refuse)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for steps in (1, 4, 3):
                plain.run(steps)
                sharded.run(steps)
        hip_engine.call("sdm_ctx_read_stats", stats, 0)
        assert stats[6] > 0 and stats[7] >= 8 * stats[6]  # collectives issued by the library
        want, got = plain.snapshot(), sharded.snapshot()
        length = int(want["length"])
        assert length < 2**13
        for key, value in want.items():
            mine = got[key]
            if key == "idx":
                value, mine = value[:length], mine[:length]
            np.testing.assert_array_equal(mine, value, err_msg=key)
    finally:
        hip_engine.call("sdm_comm_destroy")
    # without a communicator and with a callback that refuses, the step fails loudly
    again = box()
    again.shard = sharding.RecordingShard(hip_engine, 2**13, 16)
    again.read_back = True
    again.shard.callback = abi.ExchangeFn(refuse)
    with pytest.raises(RuntimeError):
        again.run(1)
