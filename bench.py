#!/usr/bin/env python3
"""bench.py -- throughput of the SDM collision hot path on MI355X.

A "step" is one collision time step (the reference's `Collision.__call__`: PCG64 draws, pair
permutation, kernel probabilities, gamma, multiplicity / attribute update, compaction) over the
whole super-droplet population, state resident in HBM, fused route (`sdm_collision_run`).
Metric (BASELINE.json): candidate super-droplet pairs per second.  With --gpus N a 0-D box runs N
independent realisations ("replicas only": one cell, no data-path collective) and rank 0 reports
the aggregate; the 32 x 32 configuration shards its cells over the ranks (pysdm_amd.sharding).

    python bench.py --gpus 1 --steps 200 --warmup 20

One JSON line on stdout (rank 0).  The timed region (exactly `--steps` steps between barriers) is
repeated `--reps` times and the median repetition is reported.  `roofline` prices the dominant
kernel of the step against the HBM peak with the algorithmic bytes of SURVEY.md 8(d) (136 B per
candidate pair, 152 B with breakup), timed live with HIP events on the library's stream;
`cpu_baseline` times the oracle (C restatement of the reference's Numba-backend algorithm, OpenMP
where Numba uses `prange`) on the host for a bounded sample, at 1 thread and at all cores.
"""
import argparse
import ctypes
import json
import os
import statistics
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_PER_PAIR = 136   # 24 u01 + 32 idx + 32 multiplicity + 16 cell id + 32*A, A=1
BYTES_PER_PAIR_BREAKUP = 152  # + proc_rand, rand_frag
# n_sd of BASELINE.json's configurations (offline counters apply to these sizes only)
CONFIG_N_SD = {"shima": 2**20, "berry_breakup": 2**20, "kinematic2d": 2**22,
               "kinematic2d_flow": 2**22, "straub": 2**22, "straub_rain": 2**22}


WORKLOADS = {
    "shima": "Shima 2009 0D box, Golovin kernel b=1500/s, n_sd=2^20 per GPU, dt=1 s "
             "(BASELINE.json configs[1]); replicas only for N>1",
    "berry_breakup": "Berry 1967 0D box, geometric kernel + Berry1967 Ec + exponential "
                     "fragmentation, n_sd=2^20 per GPU (configs[2]); replicas only for N>1",
    "straub": "Straub 2010 Ec + fragmentation on the cloud spectrum, geometric kernel, n_sd=2^22 "
              "per GPU (configs[4] as written); replicas only for N>1",
    "straub_rain": "Straub 2010 Ec + fragmentation on the Marshall-Palmer rain spectrum, dt=10 s, "
                   "n_sd=2^22 per GPU (configs[4] adaptive-substep stress variant); replicas only",
    "kinematic2d": "32x32 cells, 2^22 super-droplets, geometric kernel, adaptive, "
                   "optimized_random, dt=5 s (configs[3]); cells sharded over the ranks",
    "kinematic2d_flow": "configs[3] with the step that precedes collisions in the 2-D kinematic "
                        "set-up: displacement (single-eddy flow + sedimentation, removal of what "
                        "precipitates) then adaptive geometric coalescence, 32x32 cells, 2^22 "
                        "super-droplets, dt=5 s; both steps sharded over the ranks",
}


def build_workload(name, engine, rank, world, n_sd=None, adaptive=None, read_back=True,
                   ids_by_cell=False, grid=None, sharded=None):
    from pysdm_amd import cases, sharding

    sharded = world > 1 if sharded is None else sharded
    if name == "kinematic2d_flow":
        displacement, collisions = cases.make_kinematic_flow(
            engine, n_sd=n_sd or 2**22, grid=tuple(grid) if grid else (32, 32))
        if sharded:
            part = sharding.attach(collisions, rank, world).shard
            sharding.attach_displacement(displacement, part)
        return cases.FlowRunner(displacement, collisions)
    if name == "kinematic2d" and sharded:
        return sharding.make_sharded_box(engine, name, rank=rank, world=world, n_sd=n_sd,
                                         adaptive=adaptive)
    # 0-D boxes: every rank an independent realisation (seed 44 + rank)
    return cases.make_box(engine, name, n_sd=n_sd, adaptive=adaptive, seed=44 + rank,
                          read_back=read_back, ids_by_cell=ids_by_cell, grid=grid)


class Checkpoint:
    """the state a repetition starts from, so that every repetition times the SAME K steps (an
    adaptive or multi-cell workload evolves: later steps take other numbers of sub-steps, and
    repetitions that simply follow one another are not comparable - round 2's 12.3 / 11.3 / 7.4e9
    on the 32 x 32 grid were three different stretches of the simulation)"""

    RUNNER = ("dt_left", "stats_dt_min", "stats_n_substep", "collision_rate",
              "collision_rate_deficit", "coalescence_rate", "breakup_rate", "breakup_rate_deficit")
    SCALARS = ("offset", "offset_breakup", "sub_steps_done", "pairs_done", "steps_done")

    # (cell_start belongs to the permutation: a sorted state restored without it is a state whose
    # segments hold other cells' droplets as soon as one droplet has died since the snapshot)
    COLUMNS = ("perm", "multiplicity", "extensive", "cell_order", "cell_start")

    @staticmethod
    def _copy(array):  # torch tensors on the GPU, numpy arrays under the CPU checker
        return array.clone() if hasattr(array, "clone") else array.copy()

    @staticmethod
    def _assign(target, saved):
        if hasattr(target, "copy_"):
            target.copy_(saved)
        else:
            target[...] = saved

    # (a flow: where the super-droplets are; sharded: every id's own cell, who is whose)
    OPTIONAL = ("cell_id", "cell_origin", "position_in_cell", "cell_id_by_id")

    def __init__(self, runner):
        pop = runner.population
        self.columns = {name: self._copy(getattr(pop, name)) for name in self.COLUMNS}
        moving = hasattr(runner, "displacement")
        self.columns.update({name: self._copy(getattr(pop, name)) for name in self.OPTIONAL
                             if moving and getattr(pop, name, None) is not None})
        shard = getattr(runner, "shard", None)
        self.role = (self._copy(shard.role), shard.role_ready) if moving and shard else None
        self.diagnostics = {name: self._copy(getattr(runner, name)) for name in self.RUNNER
                            if getattr(runner, name) is not None}
        self.scalars = {name: getattr(runner, name) for name in self.SCALARS}
        self.live, self.ordered = pop.live, pop.ordered

    def restore(self, runner):
        pop = runner.population
        for name, saved in self.columns.items():
            self._assign(getattr(pop, name), saved)
        for name, saved in self.diagnostics.items():
            self._assign(getattr(runner, name), saved)
        for name, value in self.scalars.items():
            setattr(runner, name, value)
        if self.role is not None:
            self._assign(runner.shard.role, self.role[0])
            runner.shard.role_ready = self.role[1]
        pop.live = pop.working = self.live
        # (sortedness as it was: an unsorted state would be sorted again under the cell order the
        # adaptive scheme has just permuted - a different, equally valid trajectory)
        pop.ordered = self.ordered
        pop.touch_state()  # the next call starts from the host's view (control block, mirror)


def cpu_model():
    try:
        for line in subprocess.check_output(["lscpu"], text=True).splitlines():
            if line.startswith("Model name"):
                return line.split(":", 1)[1].strip()
    except (OSError, subprocess.CalledProcessError):
        pass
    return "unknown"


def cpu_baseline(workload, n_sd, adaptive, seconds_budget=8.0):
    """the oracle (kind "port": C restatement of the reference's Numba-backend algorithm; serial
    where Numba is serial, `omp parallel for` where Numba uses `prange`) on the host cores, same
    box, bounded sample: once with 1 thread, once with all cores"""
    from oracle.engine import OracleEngine

    # the host cores this process may use (the GPU box gives a 16-core share per GPU; an OpenMP team
    # wider than that only adds scheduling noise to loops this short)
    try:
        n_threads = len(os.sched_getaffinity(0))
    except AttributeError:
        n_threads = os.cpu_count() or 1
    n_threads = max(1, min(n_threads, 16))
    results = {}
    for threads in sorted({1, n_threads}):
        engine = OracleEngine.get(threads=threads)
        runner = build_workload(workload, engine, 0, 1, n_sd, adaptive)
        runner.run(1)  # warm-up (first touch)
        steps, pairs0, t0 = 0, runner.pairs_done, time.perf_counter()
        while True:
            runner.run(1)
            steps += 1
            elapsed = time.perf_counter() - t0
            if elapsed > seconds_budget or steps >= 2000:
                break
        results[threads] = ((runner.pairs_done - pairs0) / elapsed, steps, elapsed,
                            runner.population.n_sd)
    best = max(results, key=lambda k: results[k][0])
    value, steps, elapsed, n = results[best]
    return {
        "value": value,
        "unit": "candidate SD-pairs/s",
        "cores": best,
        "kind": "port",
        "cpu": cpu_model(),
        "value_1_thread": results[1][0],
        f"value_{n_threads}_threads": results[n_threads][0],
        "sample": f"{steps} time steps of the same workload ({workload}, n_sd={n}; "
                  f"{elapsed:.1f} s of CPU work per thread count, oracle/sdm_oracle*.c, "
                  f"sdm_collision_step of the oracle library)",
    }


def random_sector_ceiling(engine, n_sd, wide_records):
    """random 16-byte reads out of a table with the footprint of the pair kernel's tables (16-B
    shuffle records + 16- or 32-B {multiplicity, mass[, radius, velocity]} records per
    super-droplet), timed with HIP events inside the library: (reads/s, GB/s of 64-B sectors)"""
    table_records = n_sd * (3 if wide_records else 2)
    n_reads = 4 * n_sd
    ms, checksum = ctypes.c_double(), ctypes.c_uint64()
    engine.call("sdm_calib_random_sectors", table_records, n_reads, 10, ms, checksum)
    rate = n_reads / (ms.value * 1e-3)
    return {"table_mib": table_records * 16 / 2**20, "reads_per_launch": n_reads,
            "ms_per_launch": ms.value, "sector_misses_per_s": rate,
            "gbs": rate * 64 / 1e9}


def emulate_ranks(args, engine, adaptive, torch, dist, launch):
    """every rank of --emulate-of N in turn on this device (see the flag); returns the JSON line"""
    from pysdm_amd import cases, sharding

    n_ranks = args.emulate_of

    def fresh():
        return cases.make_box(engine, args.workload, n_sd=args.n_sd, adaptive=adaptive,
                              grid=tuple(args.grid) if args.grid else None)

    def timed(runner):
        runner.run(1)
        runner.run(args.warmup)
        runner.sync()
        torch.cuda.synchronize()
        before = (runner.pairs_done, runner.sub_steps_done)
        t0 = time.perf_counter()
        runner.run(args.steps)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        return elapsed, runner.pairs_done - before[0], runner.sub_steps_done - before[1]

    progress("one process, plain (the run the driver's line times)")
    plain = fresh()
    plain_s, pairs, substeps = timed(plain)
    del plain
    progress("one process owning every cell on the sharded code path: the trace")
    recorded = sharding.attach_recording(fresh())
    recorded_s, pairs_r, substeps_r = timed(recorded)
    assert (pairs_r, substeps_r) == (pairs, substeps)
    whole = recorded.snapshot()
    trace = recorded.shard.trace
    timed_exchanges = sum(recorded.shard.calls.values())
    n_cell, n_sd = recorded.population.n_cell, recorded.population.n_sd
    del recorded
    per_rank = {}
    for rank in (args.emulate_ranks if args.emulate_ranks else range(n_ranks)):
        runner = sharding.attach_replay(fresh(), rank, n_ranks, trace)
        elapsed, pairs_k, substeps_k = timed(runner)
        assert (pairs_k, substeps_k) == (pairs, substeps) and runner.shard.position == len(trace)
        if not sharding.emulated_rank_equals(runner, whole):
            sys.exit(f"bench.py: emulated rank {rank} of {n_ranks} does not reproduce its block")
        per_rank[rank] = elapsed / args.steps * 1e3
        progress(f"emulated rank {rank} of {n_ranks}: {per_rank[rank]:.4f} ms per step "
                 f"(cells {runner.shard.first}..{runner.shard.last - 1}; block equal to the "
                 f"one-process state)")
        del runner
    # what stands where the collective would: a device-to-device copy of the recorded words, and
    # beside it what ONE RCCL all-reduce of the per-cell exchange costs on this device (world 1:
    # no peer, but the stream-ordered device path) - both per call, Python's share included
    words = n_cell + 1 + n_ranks
    source = torch.zeros(words, dtype=torch.float64, device="cuda")
    target = torch.zeros(words, dtype=torch.float64, device="cuda")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(launch.free_port()))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))

    def per_call_us(op, n=300):
        for _ in range(20):
            op()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            op()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6

    copy_us = per_call_us(lambda: target.copy_(source))
    rccl_us = per_call_us(lambda: dist.all_reduce(target, op=dist.ReduceOp.SUM))
    dist.destroy_process_group()
    slowest = max(per_rank.values())
    exchanges_per_step = timed_exchanges / (1 + args.warmup + args.steps)
    modelled = slowest + exchanges_per_step * max(0.0, rccl_us - copy_us) * 1e-3
    return {
        "metric": f"candidate SD-pairs/s (kinematic2d, n_sd={n_sd} in total; EMULATED: each of "
                  f"{n_ranks} ranks in turn on one MI355X, the others replayed from a trace)",
        "value": pairs / (slowest * 1e-3 * args.steps), "unit": "candidate SD-pairs/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": slowest, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": WORKLOADS[args.workload] + f"; EMULATED rank of {n_ranks}",
                   "n_sd": n_sd, "seed": 44, "route": "fused sdm_collision_run"},
        "emulated": {
            "ranks": n_ranks, "cells_per_rank": n_cell // n_ranks,
            "emulated_per_rank_ms_per_step": {str(k): round(v, 5) for k, v in per_rank.items()},
            "emulated_max_ms_per_step": round(slowest, 5),
            "one_process_ms_per_step": round(plain_s / args.steps * 1e3, 5),
            "one_process_sharded_path_ms_per_step": round(recorded_s / args.steps * 1e3, 5),
            "emulated_speedup_over_one_process": round(plain_s / args.steps * 1e3 / slowest, 3),
            "substeps": substeps, "exchanges_per_step": round(exchanges_per_step, 2),
            "replay_copy_us_per_exchange": round(copy_us, 2),
            "rccl_allreduce_world1_us": round(rccl_us, 2),
            "modelled_n_ranks_ms_per_step": round(modelled, 5),
            "note": "no link was involved: the modelled step is the slowest emulated rank plus "
                    "(RCCL all-reduce - replay copy) per exchange, both measured on this device; "
                    "an N-rank ring adds per-hop xGMI latency that is not measured here",
        },
    }


def phase_timing(engine):
    n = 12
    ms = (ctypes.c_double * n)()
    count = (ctypes.c_int64 * n)()
    engine.call("sdm_ctx_read_timing", ms, count)
    name = engine.library.cdll.sdm_phase_name
    name.restype = ctypes.c_char_p
    return {name(i).decode(): (ms[i], count[i]) for i in range(n) if count[i] > 0}


_T0 = time.perf_counter()


def progress(stage):
    """one line per stage on stderr (stdout carries the JSON line alone)"""
    print(f"[bench {time.perf_counter() - _T0:7.1f} s] {stage}", file=sys.stderr, flush=True)


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--gpus", type=int, default=1)
    parser.add_argument("--steps", type=int, default=200)
    parser.add_argument("--warmup", type=int, default=20)
    parser.add_argument("--reps", type=int, default=3)
    parser.add_argument("--workload", default="shima", choices=sorted(WORKLOADS))
    parser.add_argument("--n-sd", type=int, default=None)
    parser.add_argument("--adaptive", type=int, default=None)
    parser.add_argument("--no-cpu-baseline", action="store_true")
    parser.add_argument("--roofline-steps", type=int, default=50)
    parser.add_argument("--skip-full-experiment", action="store_true",
                        help="shima: do not run the 3600-step experiment after the timed region")
    # measurement of a DIFFERENT workload (profiles/README.md): a cell's super-droplets get
    # consecutive ids; the JSON line says so in config.workload
    parser.add_argument("--ids-by-cell", action="store_true")
    # another grid for the multi-cell workload (not the configuration either): e.g. 75 75
    parser.add_argument("--grid", type=int, nargs=2, default=None)
    # measurement of the sharding protocol itself: ONE rank that owns every cell but runs the
    # sharded code path (exchange calls, ownership masks, lists), RCCL collectives on the device
    parser.add_argument("--sharded-on-one", action="store_true")
    # N = 8 without an 8-GPU node ("emulated ... state so in results"): every rank of
    # N emulated in turn on this one device - it owns its block of cells, the other ranks'
    # contributions to each exchange are replayed from the trace of a one-process run of the same
    # steps (pysdm_amd.sharding.ReplayShard); every field of the line is labelled "emulated"
    parser.add_argument("--cell-shape", type=int, default=0,
                        help="SDM_OPT_CELL_SHAPE of the context (A/B measurements; 0 = the "
                             "library's own choice, which is what the bench line is quoted on)")
    parser.add_argument("--emulate-of", type=int, default=0)
    parser.add_argument("--emulate-ranks", type=int, nargs="*", default=None,
                        help="with --emulate-of: only these ranks (default: all)")
    args = parser.parse_args()

    from pysdm_amd import launch

    if args.gpus > 1 and not launch.launched_by_torchrun():
        # `python bench.py --gpus N` without a launcher: this process becomes the launcher - it
        # starts the N ranks as children through torch.distributed.run, waits, and exits with
        # their code.  Nothing here has touched the GPU (no torch import yet), and no process that
        # has is ever re-executed; rank 0's JSON line goes to the inherited stdout.
        sys.exit(launch.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    # rehearsal knobs (not used by the driver): several ranks on one card, collectives over gloo
    dist_backend = os.environ.get("SDM_BENCH_DIST_BACKEND", "nccl")
    if os.environ.get("SDM_BENCH_ALL_ON_DEVICE0") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if args.sharded_on_one:
        if world != 1 or args.workload not in ("kinematic2d", "kinematic2d_flow"):
            sys.exit("bench.py: --sharded-on-one is for --gpus 1 and the multi-cell workloads")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(launch.free_port()))
        dist.init_process_group(dist_backend, rank=0, world_size=1,
                                **({"device_id": torch.device("cuda", local_rank)}
                                   if dist_backend == "nccl" else {}))
    if world > 1:
        if dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(dist_backend)
        # what the line reports as n_gpus is what the process group (RCCL) saw, not the flag
        world = dist.get_world_size()
    reduce_device = "cuda" if dist_backend == "nccl" else "cpu"

    from pysdm_amd.engine import HipEngine

    engine = HipEngine.get(local_rank)
    if args.cell_shape:
        engine.call("sdm_ctx_set_option", 2, args.cell_shape)
    adaptive = None if args.adaptive is None else bool(args.adaptive)
    if args.emulate_of:
        if world != 1 or args.workload != "kinematic2d":
            sys.exit("bench.py: --emulate-of is for --gpus 1 and --workload kinematic2d")
        print(json.dumps(emulate_ranks(args, engine, adaptive, torch, dist, launch)))
        return
    runner = build_workload(args.workload, engine, rank, world, args.n_sd, adaptive,
                            ids_by_cell=args.ids_by_cell, grid=args.grid,
                            sharded=world > 1 or args.sharded_on_one)
    pop, setup = runner.population, runner.setup
    n_sd = pop.n_sd
    progress(f"{args.workload}: state built (n_sd = {n_sd}), rank {rank} of {world}")
    runner.run(1)  # allocates scratch, builds the mirror
    # non-adaptive: no host read-back inside the timed loop
    runner.read_back = bool(setup.adaptive) or world > 1

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    runner.run(args.warmup)
    runner.sync()
    progress("warm-up done")
    evolving = bool(setup.adaptive) or pop.n_cell > 1
    checkpoint = Checkpoint(runner) if evolving and args.reps > 1 else None
    reps, rep_substeps, comm = [], [], None
    for rep in range(max(1, args.reps)):
        if checkpoint is not None and rep > 0:
            checkpoint.restore(runner)
        substeps_before = runner.sub_steps_done
        if runner.shard is not None:
            comm_before = runner.shard.traffic()
        barrier()
        pairs_before = int(pop.ctl[5].item())  # the library's own count (control word 5)
        host_pairs_before = runner.pairs_done
        t0 = time.perf_counter()
        runner.run(args.steps)
        barrier()
        elapsed = time.perf_counter() - t0
        runner.sync()
        if setup.adaptive or pop.n_cell > 1:
            pairs = runner.pairs_done - host_pairs_before
        else:
            # counted by the pair kernel itself: super-droplets may have coalesced away meanwhile
            pairs = int(pop.ctl[5].item()) - pairs_before
            if pop.live == n_sd:
                assert pairs == args.steps * (n_sd // 2)
        t = torch.tensor([elapsed], dtype=torch.float64, device=reduce_device)
        p = torch.tensor([float(pairs)], dtype=torch.float64, device=reduce_device)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            if not runner.counts_global_pairs:
                dist.all_reduce(p, op=dist.ReduceOp.SUM)
        reps.append((float(p.item()) / float(t.item()), float(t.item()), float(p.item())))
        rep_substeps.append(runner.sub_steps_done - substeps_before)
        progress(f"repetition {rep}: {reps[-1][0]:.4g} pairs/s")
        if runner.shard is not None:
            shard = runner.shard
            calls, payload = shard.traffic()
            comm = {"collectives": calls - comm_before[0], "bytes": payload - comm_before[1],
                    "bytes_per_step": (payload - comm_before[1]) / args.steps,
                    "backend": dist_backend,
                    # who issues them: the library (RCCL on its own stream, no host code in the
                    # sub-step loop) or a Python callback into torch.distributed
                    "issued_by": "library (ncclAllReduce)" if shard.library_comm
                                 else "python callback (torch.distributed.all_reduce)"}
    # multi-cell workload: digest of the global state (put together from the owners when sharded),
    # so that runs with different --gpus can be compared: the sharded run reproduces the
    # one-process run bit for bit
    state_digest = None
    if pop.n_cell > 1:
        import hashlib

        from pysdm_amd import sharding

        snap = sharding.gather(runner) if runner.shard is not None else runner.snapshot()
        live = snap["idx"][: int(snap["length"])]
        state_digest = hashlib.sha256(
            np.ascontiguousarray(live).tobytes()
            + np.ascontiguousarray(snap["multiplicity"]).tobytes()
            + np.ascontiguousarray(snap["attributes"]).tobytes()).hexdigest()
    rates = [r[0] for r in reps]
    median_rate = statistics.median(rates)
    _, elapsed_max, pairs_total = min(reps, key=lambda r: abs(r[0] - median_rate))

    # metric part (ii): the Shima-2009 experiment itself (settings.py:14-33: 3600 steps of 1 s
    # from the initial spectrum), run once and timed by the wall clock - not extrapolated
    shima_box_s = None
    if args.workload == "shima" and not args.skip_full_experiment:
        box = build_workload(args.workload, engine, rank, world, args.n_sd, adaptive,
                             read_back=False)
        barrier()
        t0 = time.perf_counter()
        box.run(3600)
        barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=reduce_device)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        shima_box_s = float(t.item())
        box.sync()
        shima_box_live = box.population.live
        del box

    progress("timed region done; per-kernel timing pass")
    roofline = None
    baseline = None
    sharded_run = runner.shard is not None
    if sharded_run and rank != 0:  # every process takes part in every step of a sharded run
        runner.run(args.roofline_steps)
    if rank == 0:
        # ---- per-kernel durations, HIP events on the library's own stream (separate pass)
        engine.call("sdm_ctx_set_timing", 1)
        runner.read_back = True  # events are resolved per call
        pairs_before_pass = runner.pairs_done
        t0 = time.perf_counter()
        runner.run(args.roofline_steps)
        torch.cuda.synchronize()
        timed_wall_ms = (time.perf_counter() - t0) * 1e3 / args.roofline_steps
        phases = phase_timing(engine)
        engine.call("sdm_ctx_set_timing", 0)
        per_launch = {k: v[0] / v[1] for k, v in phases.items()}
        per_step = {k: v[0] / args.roofline_steps for k, v in phases.items()}
        dominant = max(per_step, key=per_step.get)
        dom_ms = per_launch[dominant]
        # one launch of the dominant kernel covers one sub-step over the WORKING population: the
        # candidate pairs of this pass over its launches (an adaptive multi-cell step cuts the
        # working length as cells finish: n_sd // 2 would overstate the average launch)
        pass_pairs = runner.pairs_done - pairs_before_pass
        launch_pairs = (pass_pairs / phases[dominant][1] if pass_pairs > 0 else n_sd // 2)
        bytes_per_pair = BYTES_PER_PAIR_BREAKUP if setup.breakup else BYTES_PER_PAIR
        achieved = bytes_per_pair * launch_pairs / (dom_ms * 1e-3) / 1e9
        # fabric traffic and L2 counters of the dominant kernel: rocprofv3 --pmc passes of this very
        # workload, run offline and tracked in profiles/traffic.json ("r04": per workload and bench
        # phase {fetch_kb, write_kb, tcc_req, tcc_miss, bytes}; "r04_calibration": the same
        # counters for k_calib_random per read) - NOT collected in this run
        offline, counters, calibration = {}, {}, {}
        traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(traffic_file):
            with open(traffic_file, encoding="utf-8") as f:
                offline = json.load(f)
            counters = offline.get("r04", {}).get(args.workload, {}).get(dominant, {})
            calibration = offline.get("r04_calibration", {})
        full_size = n_sd == CONFIG_N_SD.get(args.workload)
        traffic = counters.get("bytes") if full_size else None
        # the access-pattern ceiling, measured now on this device with the kernel's own footprint:
        # the path is random 64-B sector misses, which this part serves far below the streaming
        # peak - both fractions are reported, the HBM one stays the contract figure
        wide = setup.breakup or args.workload in ("kinematic2d", "kinematic2d_flow",
                                                  "berry_breakup", "straub", "straub_rain")
        progress("kernel timing done; random-sector calibration")
        ceiling = random_sector_ceiling(engine, n_sd, wide)
        # like for like: the kernel's L2 misses per second against the calibration's MISSES per
        # second, its L2 requests against the calibration's REQUESTS (the calibration's reads are
        # not all misses: profiles/traffic.json has its counters per read)
        l2 = None
        if full_size and counters.get("tcc_miss") and calibration.get("tcc_miss_per_read"):
            seconds = dom_ms * 1e-3
            calib_reads_per_s = ceiling["sector_misses_per_s"]  # (reads per second, measured now)
            l2 = {
                "kernel_requests_per_launch": counters["tcc_req"],
                "kernel_misses_per_launch": counters["tcc_miss"],
                "kernel_requests_per_pair": counters["tcc_req"] / launch_pairs,
                "kernel_misses_per_pair": counters["tcc_miss"] / launch_pairs,
                "kernel_misses_per_s": counters["tcc_miss"] / seconds,
                "kernel_requests_per_s": counters["tcc_req"] / seconds,
                "calibration_misses_per_s": calib_reads_per_s * calibration["tcc_miss_per_read"],
                "calibration_requests_per_s": calib_reads_per_s * calibration["tcc_req_per_read"],
            }
            l2["frac_of_ceiling_misses"] = l2["kernel_misses_per_s"] / l2["calibration_misses_per_s"]
            l2["frac_of_ceiling_requests"] = (l2["kernel_requests_per_s"]
                                              / l2["calibration_requests_per_s"])
        roofline = {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_source": "offline PMC (profiles/traffic.json, rocprofv3 --pmc FETCH_SIZE / "
                              "WRITE_SIZE passes of this workload; not collected in this run)",
            "kernel": dominant, "kernel_ms": dom_ms,
            "pairs_per_launch": launch_pairs,
            # 64-B sectors under independent random access, same table footprint, same run
            "random_sector_ceiling_gbs": ceiling["gbs"],
            "random_sector_calibration": ceiling,
            "l2": l2,
            "algorithmic_bytes_per_launch": bytes_per_pair * launch_pairs,
            # the same bytes against the wall-clock time of a whole time step (all kernels)
            "whole_step_frac": bytes_per_pair * pairs_total / world / elapsed_max / 1e9
                               / HBM_PEAK_GBS,
            "phase_ms_per_step": {k: round(v, 5) for k, v in sorted(per_step.items())},
            "phase_sum_ms_per_step": round(sum(per_step.values()), 5),
            "timed_mode_ms_per_step": round(timed_wall_ms, 5),
        }
        if not args.no_cpu_baseline and world == 1:  # the CPU leg: rank 0 at N=1 only
            progress("CPU baseline (the oracle on the host cores, bounded sample)")
            baseline = cpu_baseline(args.workload, args.n_sd, adaptive)

    if world > 1 or args.sharded_on_one:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        value = pairs_total / elapsed_max
        sharded = runner.shard is not None
        flow = getattr(runner, "displacement", None)
        print(json.dumps({
            # BASELINE.json's metric; `value` is the aggregate over all ranks (bench contract)
            "metric": ("candidate SD-pairs/s per GPU; Shima-2009 box wall-clock at n_sd=2^20"
                       if args.workload == "shima" and n_sd == 2**20
                       else f"candidate SD-pairs/s ({args.workload}, n_sd={n_sd}"
                            f"{' in total' if sharded else ' per rank'})"),
            "value": value,
            "unit": "candidate SD-pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if sharded else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": WORKLOADS[args.workload] + "; "
                            + ("adaptive" if setup.adaptive else "non-adaptive")
                            + ("; NOT the configuration: ids ordered by cell"
                               if args.ids_by_cell else "")
                            + (f"; NOT the configuration: grid {args.grid[0]} x {args.grid[1]}"
                               if args.grid else "")
                            + ("; the sharded code path on ONE rank that owns every cell"
                               if args.sharded_on_one else ""),
                "n_sd": n_sd,
                "seed": 44,
                "route": ("fused sdm_collision_run" if flow is None else
                          "sdm_displacement_step" + ("_sharded" if sharded else "")
                          + " + fused sdm_collision_run, once each per time step"),
            },
            # the flow sharded: what the displacement steps of this rank exchanged over the run
            "displacement_exchange": (flow.shard_stats if flow is not None and sharded else None),
            "state_digest": state_digest,
            # sharded runs: what this rank handed to collectives during the LAST repetition
            # (per-sub-step sums of n_cell + 1 + world doubles; dead positions when one died)
            "comm": comm,
            "repetitions": {"n": len(reps), "reported": "median",
                            "values": [round(r, 1) for r in rates],
                            # sub-steps executed in each repetition's K steps (work per step)
                            "substeps": rep_substeps,
                            "same_steps_each_time": checkpoint is not None},
            # measured: one run of the whole experiment (3600 steps from the initial state, one
            # library call, mirror build included); null for the other workloads
            "shima_box_3600_steps_s": shima_box_s,
            "shima_box_3600_steps_live_sd": shima_box_live if shima_box_s is not None else None,
            "shima_box_3600_steps_extrapolated_s": (elapsed_max / args.steps * 3600
                                                    if args.workload == "shima" else None),
            "roofline": roofline,
            "cpu_baseline": baseline,
        }))


if __name__ == "__main__":
    main()
