"""Multi-process path on CPU (gloo, world size 2; the oracle library, which implements the sharded
mode of include/sdm_hip.h like the product): the cells of a 4 x 4 grid divided over two
processes.  The state put together from the owners must equal the UNSHARDED run of the reference
(tests/golden/traj_multicell_*.npz) bit for bit: same seed, one random stream."""
import os
import socket
import warnings

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from pysdm_amd import sharding

from .trajectory import compare, setup_from_golden


def test_cell_blocks_cover_the_domain():
    for n_cell, world in ((16, 2), (1024, 8), (15, 4), (3, 8)):
        blocks = [sharding.cell_block(n_cell, r, world) for r in range(world)]
        assert blocks[0][0] == 0 and blocks[-1][1] == n_cell
        assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
        sizes = [b - a for a, b in blocks]
        assert max(sizes) - min(sizes) <= 1


CASES = ("traj_multicell_golovin_4x4", "traj_multicell_geometric_4x4",
         "traj_multicell_golovin_4x4_na", "traj_multicell_golovin_8x8_sparse",
         "traj_multicell_geometric_3x5")


def sharded_run_equals_golden(name, engine, rank, world, float_rtol=0.0):
    runner, gold, steps = setup_from_golden(name, engine)
    sharding.attach(runner, rank, world)
    initial = engine.download(runner.population.multiplicity)
    for step in steps:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            runner.run(step - runner.steps_done)
        snap = sharding.gather(runner)
        if not runner.setup.adaptive:
            snap.pop("stats_dt_min")
        else:  # NaN-sticky minima: NaN + 0 = NaN on the owner's side, as in the golden
            pass
        compare(snap, gold, step, float_rtol=float_rtol, idx_tail=False)
    # the work really is divided: this process never touched the other processes' droplets
    local = engine.download(runner.population.multiplicity)
    mine = sharding.owned_droplets(runner)
    np.testing.assert_array_equal(local[~mine], initial[~mine])
    if snap["collision_rate"][~runner.shard.owned_host].sum() > 0:  # the others did collide
        assert (snap["multiplicity"][~mine] != initial[~mine]).any()
    return runner


def sharded_box_equals_single(engine, rank, world, runs=(1, 4, 3), **box):
    """a box of pysdm_amd.cases run by `world` processes beside the one-process run on the same
    engine: the state gathered from the owners, bit for bit, after every call"""
    from pysdm_amd import cases  # pylint: disable=import-outside-toplevel

    single = cases.make_box(engine, **box)
    shard = cases.make_box(engine, **box)
    sharding.attach(shard, rank, world)
    for steps in runs:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            single.run(steps)
            shard.run(steps)
        got, ref = sharding.gather(shard), single.snapshot()
        length = int(ref["length"])
        assert int(got["length"]) == length
        for key, value in ref.items():
            if key == "stats_dt_min" and not box.get("adaptive", True):
                continue
            mine = got[key]
            if key == "idx":
                value, mine = value[:length], mine[:length]
            np.testing.assert_array_equal(mine, value, err_msg=f"{box} {key}")
    return shard, length


# sharded runs beyond the per-cell kernels (round 4): cells larger than their capacity (8192 per
# cell here) and the global croupier take the generic kernels, which skip other processes' pairs
BEYOND_THE_CELL_KERNELS = (
    dict(name="shima", n_sd=2**15, adaptive=True, dt=200.0, thin=0.02, grid=(2, 2)),
    dict(name="shima", n_sd=2**15, adaptive=False, dt=200.0, thin=0.02, grid=(2, 2)),
    dict(name="shima", n_sd=2**12, adaptive=False, dt=200.0, thin=0.02, grid=(4, 4),
         croupier="global"),
    dict(name="shima", n_sd=2**12, adaptive=False, dt=50.0, grid=(3, 5), croupier="global",
         substeps=2),
)


def _worker(rank, world, port, errors):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from oracle.engine import OracleEngine  # pylint: disable=import-outside-toplevel

        engine = OracleEngine.get()
        exchanged = 0
        for name in CASES:
            runner = sharded_run_equals_golden(name, engine, rank, world)
            exchanged += runner.shard.calls[1]
        assert exchanged > 0
        # deaths: the permutation is put together again across the processes (thin multiplicities
        # on a grid; no reference golden: against the one-process oracle run)
        from pysdm_amd import cases  # pylint: disable=import-outside-toplevel

        for adaptive in (True, False):
            single = cases.make_box(engine, "shima", n_sd=2**11, adaptive=adaptive, dt=200.0,
                                    thin=0.02, grid=(4, 4))
            shard = cases.make_box(engine, "shima", n_sd=2**11, adaptive=adaptive, dt=200.0,
                                   thin=0.02, grid=(4, 4))
            sharding.attach(shard, rank, world)
            for steps in (1, 4, 3):
                single.run(steps)
                shard.run(steps)
                got, ref = sharding.gather(shard), single.snapshot()
                length = int(ref["length"])
                assert int(got["length"]) == length < 2**11
                for key, value in ref.items():
                    if key == "stats_dt_min" and not adaptive:
                        continue
                    mine = got[key]
                    if key == "idx":
                        value, mine = value[:length], mine[:length]
                    np.testing.assert_array_equal(mine, value, err_msg=f"{adaptive} {key}")
            assert shard.shard.calls[2] > 0  # the permutation did cross the processes
        for box in BEYOND_THE_CELL_KERNELS:
            shard, length = sharded_box_equals_single(engine, rank, world, **box)
            if box.get("thin"):
                assert length < box["n_sd"]
        # displacement (replicated, on the completed state) + sharded collisions: super-droplets
        # migrate between the processes' cells every step; against the reference's golden
        from . import displacement_cases  # pylint: disable=import-outside-toplevel

        displacement_cases.run_case("disp2d_collide", engine, shard=(rank, world))
        # the displacement step sharded as well (sdm_displacement_step_sharded)
        for name in displacement_cases.CASES:
            moved = displacement_cases.run_case(name, engine, shard=(rank, world),
                                                owner_moves=True)
            stats = moved.shard_stats
            assert stats["moved"] > 0 and stats["calls"] > 0, stats
            if name != "disp1d_implicit_sed":
                assert stats["removed"] > 0 and stats["left"] + stats["arrived"] > 0, stats
        stats = displacement_cases.random_flow_pair_equal(engine, rank, world,
                                                          **displacement_cases.DIES_OUT)
        assert stats["live"] == 0, stats
        # ... and beside the one-process run of the 2-D kinematic set-up (eddy + sedimentation)
        stats = displacement_cases.sharded_flow_equals_single(engine, rank, world, n_sd=2**13,
                                                              grid=(8, 8), steps=6)
        assert stats["left"] > 0 and stats["arrived"] > 0 and stats["removed"] > 0, stats
        dist.barrier()
        dist.destroy_process_group()
    except Exception as exc:  # pylint: disable=broad-except
        errors.put(f"rank {rank}: {exc!r}")
        raise


@pytest.mark.timeout(600)
def test_two_process_sharded_run_equals_the_unsharded_reference():
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    ctx = mp.get_context("spawn")
    errors = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, errors)) for r in range(2)]
    for proc in procs:
        proc.start()
    for proc in procs:
        proc.join(500)
    failed = [p.exitcode for p in procs if p.exitcode != 0]
    messages = []
    while not errors.empty():
        messages.append(errors.get())
    assert not failed and not messages, f"{failed} {messages}"


def _worker_three(rank, world, port, errors):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from oracle.engine import OracleEngine  # pylint: disable=import-outside-toplevel

        from . import displacement_cases  # pylint: disable=import-outside-toplevel

        engine = OracleEngine.get()
        for name in displacement_cases.CASES:
            displacement_cases.run_case(name, engine, shard=(rank, world), owner_moves=True)
        stats = displacement_cases.sharded_flow_equals_single(engine, rank, world, n_sd=2**12,
                                                              grid=(5, 7), steps=5)
        assert stats["left"] > 0 and stats["arrived"] > 0, stats
        dist.barrier()
        dist.destroy_process_group()
    except Exception as exc:  # pylint: disable=broad-except
        errors.put(f"rank {rank}: {exc!r}")
        raise


@pytest.mark.timeout(600)
def test_three_processes_with_uneven_blocks_of_cells():
    """both steps sharded over THREE processes (blocks of 12 / 12 / 11 cells on the 5 x 7 grid, of
    6 / 5 / 5 on the goldens' 4 x 4): the reference's displacement goldens and the kinematic flow
    beside the one-process run"""
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    ctx = mp.get_context("spawn")
    errors = ctx.Queue()
    procs = [ctx.Process(target=_worker_three, args=(r, 3, port, errors)) for r in range(3)]
    for proc in procs:
        proc.start()
    for proc in procs:
        proc.join(500)
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
        assert result["kind"] == "port" and result["value"] > 0
        assert result["value_1_thread"] > 0 and result["cores"] >= 1


def test_bench_checkpoint_brings_back_the_same_steps_after_deaths(oracle_engine):
    """bench.py restores one snapshot before every timed repetition: the K steps that follow must
    be the same K steps each time, also when super-droplets died in between (the permutation is
    then shorter and cell_start has moved: both belong to the snapshot)"""
    import importlib.util  # pylint: disable=import-outside-toplevel

    from pysdm_amd.cases import make_box  # pylint: disable=import-outside-toplevel

    spec = importlib.util.spec_from_file_location(
        "bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                              "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    runner = make_box(oracle_engine, "shima", n_sd=2**12, adaptive=True, dt=200.0, thin=0.02,
                      grid=(4, 4))
    runner.run(2)
    start = bench.Checkpoint(runner)
    seen = []
    for _ in range(3):
        start.restore(runner)
        runner.run(4)
        snap = runner.snapshot()
        seen.append((int(snap["length"]), runner.sub_steps_done, runner.offset,
                     {k: np.asarray(v).tobytes() for k, v in snap.items()}))
    assert seen[0][0] < start.live  # droplets died inside the repetition
    assert seen[1] == seen[0] and seen[2] == seen[0]
    # the flow workload (displacement + collisions): positions and cells belong to the snapshot
    flow = bench.build_workload("kinematic2d_flow", oracle_engine, 0, 1, n_sd=2**12, grid=(4, 4))
    flow.run(2)
    start = bench.Checkpoint(flow)
    assert {"cell_origin", "position_in_cell", "cell_id"} <= set(start.columns)
    seen = []
    for _ in range(2):
        start.restore(flow)
        flow.run(3)
        pop = flow.population
        seen.append((pop.live, flow.sub_steps_done, flow.offset,
                     [np.asarray(a).tobytes() for a in (
                         pop.perm[:pop.live], pop.multiplicity, pop.extensive, pop.cell_id,
                         pop.cell_origin, pop.position_in_cell)]))
    assert seen[0][0] <= start.live and seen[1] == seen[0]


def _worker_eight(rank, world, port, errors):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from oracle.engine import OracleEngine  # pylint: disable=import-outside-toplevel

        from . import digests, displacement_cases  # pylint: disable=import-outside-toplevel

        engine = OracleEngine.get()
        # BASELINE.json configs[3]'s grid at the process count its target names: 32 x 32 cells,
        # blocks of 128 per process, against the digest of the REFERENCE's own run
        runner = digests.check("kinematic2d_64percell", engine,
                               prepare=lambda r: sharding.attach(r, rank, world),
                               snapshot=sharding.gather)
        first, last = runner.shard.first, runner.shard.last
        assert last - first == 128 and runner.shard.calls[1] > 0
        # the flow - displacement and collisions both sharded - beside the one-process run after
        # every step: 32 x 32 cells again (128 per process), super-droplets change owner all along
        stats = displacement_cases.sharded_flow_equals_single(engine, rank, world, n_sd=2**15,
                                                              grid=(32, 32), steps=4)
        assert stats["left"] > 0 and stats["arrived"] > 0, stats
        dist.barrier()
        dist.destroy_process_group()
    except Exception as exc:  # pylint: disable=broad-except
        errors.put(f"rank {rank}: {exc!r}")
        raise


@pytest.mark.timeout(900)
def test_eight_processes_reproduce_the_reference_digest_of_the_32x32_grid():
    """world size 8 - the only process count BASELINE.json's target has: the 32 x 32 grid in blocks
    of 128 cells; the state gathered from the owners equals the reference's digest (three adaptive
    steps), and the sharded flow equals the one-process run step by step"""
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    ctx = mp.get_context("spawn")
    errors = ctx.Queue()
    procs = [ctx.Process(target=_worker_eight, args=(r, 8, port, errors)) for r in range(8)]
    for proc in procs:
        proc.start()
    for proc in procs:
        proc.join(800)
    failed = [p.exitcode for p in procs if p.exitcode != 0]
    messages = []
    while not errors.empty():
        messages.append(errors.get())
    for proc in procs:
        if proc.is_alive():
            proc.kill()
    assert not failed and not messages, f"{failed} {messages}"


def test_an_emulated_rank_replayed_against_a_one_process_trace_computes_its_block(oracle_engine):
    """bench.py --emulate-of N prices one rank of N on a single device: it owns its block of cells,
    and what the others would have contributed to every exchange is copied in from the trace of a
    one-process run (sharding.RecordingShard / ReplayShard).  Here on the checker: every rank of 8
    (and of 3: uneven blocks) reproduces its block of the one-process state, deaths included, and a
    replay against the trace of OTHER steps is refused"""
    from pysdm_amd import cases  # pylint: disable=import-outside-toplevel

    def box():
        return cases.make_box(oracle_engine, "shima", n_sd=2**12, adaptive=True, dt=200.0,
                              thin=0.02, grid=(8, 4))

    recorded = sharding.attach_recording(box())
    plain = box()
    for steps in (1, 3, 2):
        recorded.run(steps)
        plain.run(steps)
    whole = plain.snapshot()
    assert int(whole["length"]) < 2**12 and recorded.shard.calls[2] > 0  # super-droplets died
    got = recorded.snapshot()
    for key in ("idx", "multiplicity", "attributes", "cell_start"):
        np.testing.assert_array_equal(got[key], whole[key])
    trace = recorded.shard.trace
    for world in (8, 3):
        for rank in range(world):
            emulated = sharding.attach_replay(box(), rank, world, trace)
            for steps in (1, 3, 2):
                emulated.run(steps)
            assert emulated.shard.position == len(trace)
            assert sharding.emulated_rank_equals(emulated, whole), (rank, world)
    other = sharding.attach_replay(box(), 0, 2, trace[3:])
    with pytest.raises(RuntimeError):
        other.run(6)
