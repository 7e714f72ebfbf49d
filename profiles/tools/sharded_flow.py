"""what the sharded displacement step exchanges, and what it costs: the 2-D kinematic set-up
(single-eddy flow + sedimentation, then adaptive Geometric coalescence; pysdm_amd.cases.
make_kinematic_flow) at 2^22 super-droplets on 32 x 32 cells, both steps sharded over the ranks.
Rehearsal on the one-GPU box: the ranks share the card and exchange through gloo (host), so the
TIMES below say nothing about xGMI; the COUNTS (movers, rows, words per step) are what a real
N-GPU run exchanges.
    PYTHONPATH=. python profiles/tools/sharded_flow.py [--ranks 2] [--steps 10] [--n-sd 4194304]
(the parent starts the ranks itself, before anything touches the GPU)"""
import argparse
import json
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pysdm_amd import launch  # noqa: E402


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--ranks", type=int, default=2)
    parser.add_argument("--steps", type=int, default=10)
    parser.add_argument("--n-sd", type=int, default=2**22)
    parser.add_argument("--grid", type=int, default=32)
    parser.add_argument("--backend", default="gloo", choices=("gloo", "nccl"),
                        help="nccl with --ranks 1: the sharded code path on one GPU, collectives "
                             "on the device (what the protocol itself costs, no host staging)")
    args = parser.parse_args()
    if not launch.launched_by_torchrun():
        if args.ranks == 1:  # in this very process (so that a profiler sees the kernels)
            os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
                              MASTER_PORT=str(launch.free_port()))
        else:
            sys.exit(launch.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.ranks))
    import torch.distributed as dist

    from pysdm_amd import cases, sharding
    from pysdm_amd.engine import HipEngine

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if args.backend == "nccl":
        import torch

        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    engine = HipEngine.get()
    grid = (args.grid, args.grid)
    displacement, collisions = cases.make_kinematic_flow(engine, n_sd=args.n_sd, grid=grid)
    part = sharding.attach(collisions, rank, world).shard
    sharding.attach_displacement(displacement, part)
    # the one-process displacement step on the same state, for the price of the protocol
    plain_d, plain_c = cases.make_kinematic_flow(engine, n_sd=args.n_sd, grid=grid)
    plain_ms = []
    for _ in range(4):
        engine.synchronize()
        t0 = time.perf_counter()
        plain_d.run()
        engine.synchronize()
        plain_ms.append((time.perf_counter() - t0) * 1e3)
        plain_c.run(1)
    del plain_d, plain_c
    rows = []
    for step in range(1, args.steps + 1):
        before = dict(displacement.shard_stats)
        bytes_before = sum(part.bytes.values())
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            engine.synchronize()
            t0 = time.perf_counter()
            displacement.run()
            engine.synchronize()
            t1 = time.perf_counter()
            disp_bytes = sum(part.bytes.values()) - bytes_before
            collisions.run(1)
            engine.synchronize()
            t2 = time.perf_counter()
        now = displacement.shard_stats
        rows.append({"step": step, "live": collisions.population.live,
                     "moved_here": now["moved"] - before["moved"],
                     "left_here": now["left"] - before["left"],
                     "arrived_here": now["arrived"] - before["arrived"],
                     "removed_everywhere": now["removed"] - before["removed"],
                     "displacement_exchange_bytes": disp_bytes,
                     "collision_exchange_bytes": sum(part.bytes.values()) - bytes_before - disp_bytes,
                     "displacement_ms": (t1 - t0) * 1e3, "collision_ms": (t2 - t1) * 1e3})
    if rank == 0:
        n_attr = int(collisions.population.extensive.shape[0])
        print(json.dumps({
            "workload": f"kinematic flow, {args.n_sd} super-droplets, {grid[0]} x {grid[1]} cells, "
                        f"{world} ranks (one card, {args.backend})",
            "whole_column_exchange_bytes_per_step_before": args.n_sd * 8 * (2 + n_attr),
            "one_process_displacement_ms": plain_ms,
            "rank_0_per_step": rows}, indent=1))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
