"""Fuzz of the sharded displacement + collision steps on the CPU checker (gloo ranks): random
grids (1-D / 2-D / 3-D), Courant fields, schemes, sedimentation on / off, world sizes 2-4, thin
multiplicities so that super-droplets die in collisions too - beside the one-process run on the
same engine, after every step (tests/displacement_cases.py:flow_pair_equal).
    python tests/fuzz_sharded_flow.py [--cases 40] [--seed 1] [--world 0 (= random 2..4)]"""
import argparse
import os
import socket
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


def draw_case(rng):
    dims = int(rng.integers(1, 4))
    grid = tuple(int(g) for g in rng.integers(2, (9, 7, 4)[dims - 1] + 1, size=dims))
    n_cell = int(np.prod(grid))
    n_sd = int(rng.integers(max(64, 4 * n_cell), 40 * n_cell + 200))
    return {"grid": grid, "n_sd": n_sd, "seed": int(rng.integers(1, 2**31)),
            "sedimentation": bool(rng.integers(0, 2)), "explicit": bool(rng.integers(0, 2)),
            "courant": float(rng.uniform(0.05, 0.45)), "steps": int(rng.integers(2, 7)),
            "thin": bool(rng.integers(0, 2)), "adaptive_displacement": bool(rng.integers(0, 2)),
            "collisions": bool(rng.integers(0, 4))}


def worker(rank, world, port, cases, failures):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.engine import OracleEngine
    from tests import displacement_cases

    engine = OracleEngine.get()
    for number, case in enumerate(cases):
        try:
            if int(np.prod(case["grid"])) < world:
                continue
            stats = displacement_cases.random_flow_pair_equal(engine, rank, world, **case)
            if rank == 0:
                print(number, case, stats, flush=True)
        except Exception:  # pylint: disable=broad-except
            failures.put((rank, number, case, traceback.format_exc()))
            break
    dist.barrier()
    dist.destroy_process_group()


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--cases", type=int, default=40)
    parser.add_argument("--seed", type=int, default=1)
    parser.add_argument("--world", type=int, default=0)
    args = parser.parse_args()
    rng = np.random.default_rng(args.seed)
    failed = 0
    done = 0
    while done < args.cases:
        world = args.world or int(rng.integers(2, 5))
        batch = [draw_case(rng) for _ in range(min(10, args.cases - done))]
        done += len(batch)
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        ctx = mp.get_context("spawn")
        failures = ctx.Queue()
        procs = [ctx.Process(target=worker, args=(r, world, port, batch, failures))
                 for r in range(world)]
        for proc in procs:
            proc.start()
        for proc in procs:
            proc.join(600)
        for proc in procs:
            if proc.is_alive():
                proc.kill()
                failed += 1
                print("a rank did not finish (its peer failed?)", flush=True)
        while not failures.empty():
            rank, number, case, text = failures.get()
            failed += 1
            print(f"FAILED world {world} rank {rank} case {number}: {case}\n{text}", flush=True)
    print("failed:", failed)
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
