#!/usr/bin/env python3
"""bench.py -- throughput of the SDM collision hot path on MI355X.

A "step" is one `Collision.__call__` (one time step of the Shima-2009 box: PCG64 draws, pair
permutation, Golovin probabilities, gamma, multiplicity/attribute update, compaction) over the
whole super-droplet population, state resident in HBM.  Metric (BASELINE.json): candidate
super-droplet pairs per second; with --gpus N every rank runs an independent realisation of the
box (the 0-D box has a single cell: "replicas only", no data-path collective) and rank 0 reports
the aggregate.

    python bench.py --gpus 1 --steps 200 --warmup 20

One JSON line on stdout (rank 0).  `roofline` prices the dominant kernel of the step against the
HBM peak with the algorithmic bytes of SURVEY.md section 8(d) (136 B per candidate pair for the
coalescence-only box), timed live with HIP events on the library's stream; `cpu_baseline` times
the oracle (serial C restatement of the reference algorithm) on the host for a bounded sample.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_PER_PAIR = 136   # 24 u01 + 32 idx + 32 multiplicity + 16 cell id + 32*A, A=1


WORKLOADS = {
    "shima": "Shima 2009 0D box, Golovin kernel b=1500/s, n_sd=2^20 per GPU, dt=1 s "
             "(BASELINE.json configs[1]); replicas only for N>1",
    "berry_breakup": "Berry 1967 0D box, geometric kernel + Berry1967 Ec + exponential "
                     "fragmentation, n_sd=2^20 per GPU (configs[2]); replicas only for N>1",
    "straub": "Straub 2010 Ec + fragmentation, geometric kernel, n_sd=2^22 per GPU (configs[4]); "
              "replicas only for N>1",
    "kinematic2d": "32x32 cells, 2^22 super-droplets, geometric kernel, adaptive, "
                   "optimized_random, dt=5 s (configs[3]); cells sharded over the ranks",
}


def build_workload(name, backend_class, rank, world, n_sd=None, adaptive=None):
    from pysdm_amd import sharding
    from pysdm_amd.examples import CONFIGS, make_box

    if name == "kinematic2d":
        n_cell = int(np.prod(CONFIGS[name]["grid"]))
        block = sharding.cell_block(n_cell, rank, world) if world > 1 else None
        return make_box(backend_class, name, n_sd=n_sd, adaptive=adaptive, cell_block=block)
    # 0-D boxes: every rank an independent realisation (seed 44 + rank)
    return make_box(backend_class, name, n_sd=n_sd, adaptive=adaptive, seed=44 + rank)


def cpu_baseline(workload, n_sd, adaptive, seconds_budget=12.0):
    """the oracle (kind "port": serial C restatement of the reference's Numba-backend algorithm,
    driven method by method like the reference) on one host core, same box, bounded sample"""
    from oracle.backend import OracleBackend

    particulator, dynamic = build_workload(workload, OracleBackend, 0, 1, n_sd, adaptive)
    particulator.run(1)  # warm-up (first-touch, lazy attribute allocation)
    steps, t0 = 0, time.perf_counter()
    substeps0 = int(dynamic.stats_n_substep.to_ndarray().sum())
    while True:
        particulator.run(1)
        steps += 1
        elapsed = time.perf_counter() - t0
        if elapsed > seconds_budget or steps >= 2000:
            break
    if dynamic.adaptive:  # candidate pairs = sum over sub-steps (all cells still full-length)
        n_cell = particulator.mesh.n_cell
        substeps = int(dynamic.stats_n_substep.to_ndarray().sum()) - substeps0
        pairs = substeps * (particulator.n_sd // n_cell // 2)
    else:
        pairs = steps * (particulator.n_sd // 2)
    return {
        "value": pairs / elapsed,
        "unit": "candidate SD-pairs/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{steps} time steps of the same workload ({workload}, n_sd={particulator.n_sd}; "
                  f"{elapsed:.1f} s of CPU work, oracle/sdm_oracle.c via oracle/backend.py)",
    }


def phase_timing(lib, handle):
    from pysdm_amd._lib import check

    n = 12
    ms = (ctypes.c_double * n)()
    count = (ctypes.c_int64 * n)()
    check(lib.sdm_ctx_read_timing(handle, ms, count))
    lib.sdm_phase_name.restype = ctypes.c_char_p
    return {lib.sdm_phase_name(i).decode(): (ms[i], count[i]) for i in range(n) if count[i] > 0}


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--gpus", type=int, default=1)
    parser.add_argument("--steps", type=int, default=200)
    parser.add_argument("--warmup", type=int, default=20)
    parser.add_argument("--workload", default="shima", choices=sorted(WORKLOADS))
    parser.add_argument("--n-sd", type=int, default=None)
    parser.add_argument("--adaptive", type=int, default=None)
    parser.add_argument("--no-cpu-baseline", action="store_true")
    parser.add_argument("--roofline-steps", type=int, default=50)
    args = parser.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # rehearsal knobs (not used by the driver): several ranks on one card, collectives over gloo
    dist_backend = os.environ.get("SDM_BENCH_DIST_BACKEND", "nccl")
    if os.environ.get("SDM_BENCH_ALL_ON_DEVICE0") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(dist_backend)
    reduce_device = "cuda" if dist_backend == "nccl" else "cpu"

    from pysdm_amd.backends import HIP
    from pysdm_amd.backends.hip import _Context

    adaptive = None if args.adaptive is None else bool(args.adaptive)
    particulator, dynamic = build_workload(args.workload, HIP, rank, world, args.n_sd, adaptive)
    n_sd = particulator.n_sd
    particulator.run(1)  # builds the fused step, allocates scratch
    fused = dynamic._fused_state  # pylint: disable=protected-access
    assert fused not in (None, False), "the fused HIP route must be the one benchmarked"
    # non-adaptive: no host read-back inside the timed loop
    fused.read_back = bool(dynamic.adaptive)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    particulator.run(args.warmup)
    fused.total_pairs = 0
    barrier()
    pairs_before = int(fused.ctl[5].item())  # the library's own count (control word 5)
    t0 = time.perf_counter()
    particulator.run(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    fused.sync()
    if dynamic.adaptive:
        pairs = fused.total_pairs
    elif particulator.mesh.n_cell == 1:
        # counted by the pair kernel itself: super-droplets may have coalesced away meanwhile
        pairs = int(fused.ctl[5].item()) - pairs_before
        if particulator.attributes.super_droplet_count == n_sd:
            assert pairs == args.steps * (n_sd // 2)
    else:  # one sub-step per time step over the whole (still complete) population
        assert particulator.attributes.super_droplet_count == n_sd, "droplets were removed"
        pairs = args.steps * (n_sd // 2)

    t = torch.tensor([elapsed], dtype=torch.float64, device=reduce_device)
    p = torch.tensor([float(pairs)], dtype=torch.float64, device=reduce_device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(p, op=dist.ReduceOp.SUM)
    elapsed_max, pairs_total = float(t.item()), float(p.item())

    roofline = None
    baseline = None
    if rank == 0:
        # ---- per-kernel durations, HIP events on the library's own stream (separate pass)
        ctx = _Context.get()
        from pysdm_amd._lib import check

        check(ctx.lib.sdm_ctx_set_timing(ctx.handle, 1))
        fused.read_back = True  # events are resolved per call
        particulator.run(args.roofline_steps)
        phases = phase_timing(ctx.lib, ctx.handle)
        check(ctx.lib.sdm_ctx_set_timing(ctx.handle, 0))
        per_launch = {k: v[0] / v[1] for k, v in phases.items()}
        per_step = {k: v[0] / args.roofline_steps for k, v in phases.items()}
        dominant = max(per_step, key=per_step.get)
        dom_ms = per_launch[dominant]
        launch_pairs = n_sd // 2
        bytes_per_pair = BYTES_PER_PAIR + (16 if dynamic.enable_breakup else 0)
        achieved = bytes_per_pair * launch_pairs / (dom_ms * 1e-3) / 1e9
        traffic = None
        traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(traffic_file):
            with open(traffic_file, encoding="utf-8") as f:
                # measured offline with rocprofv3 --pmc (see profiles/README.md); per workload
                traffic = json.load(f).get(args.workload, {}).get(dominant)
        roofline = {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "kernel": dominant, "kernel_ms": dom_ms,
            "algorithmic_bytes_per_launch": bytes_per_pair * launch_pairs,
            "phase_ms_per_step": {k: round(v, 5) for k, v in sorted(per_step.items())},
        }
        if not args.no_cpu_baseline and world == 1:  # the CPU leg: rank 0 at N=1 only
            baseline = cpu_baseline(args.workload, args.n_sd, adaptive)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        value = pairs_total / elapsed_max
        print(json.dumps({
            # BASELINE.json's metric; `value` is the aggregate over all ranks (bench contract)
            "metric": ("candidate SD-pairs/s per GPU; Shima-2009 box wall-clock at n_sd=2^20"
                       if args.workload == "shima" and n_sd == 2**20
                       else f"candidate SD-pairs/s ({args.workload}, n_sd={n_sd} per rank)"),
            "value": value,
            "unit": "candidate SD-pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if args.workload == "kinematic2d" else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": WORKLOADS[args.workload] + "; "
                            + ("adaptive" if dynamic.adaptive else "non-adaptive"),
                "n_sd": n_sd,
                "seed": 44,
                "route": "fused sdm_collision_step",
            },
            "shima_box_3600_steps_s": elapsed_max / args.steps * 3600,
            "roofline": roofline,
            "cpu_baseline": baseline,
        }))


if __name__ == "__main__":
    main()
