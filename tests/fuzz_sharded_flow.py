"""Fuzz of the sharded displacement + collision steps on the CPU checker (gloo ranks): random
grids (1-D / 2-D / 3-D), Courant fields, schemes, sedimentation on / off, world sizes 2-4, thin
multiplicities so that super-droplets die in collisions too - beside the one-process run on the
same engine (`--engine hip`: the product, the ranks sharing one card), after every step (tests/displacement_cases.py:flow_pair_equal).
    python tests/fuzz_sharded_flow.py [--cases 40] [--seed 1] [--world 0 (= random 2..4)]
                                      [--engine oracle|hip]"""
import argparse
import os
import socket
import sys
import time
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


def worker(rank, world, port, cases, failures, kind):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import displacement_cases

    if kind == "hip":  # (the ranks share the one card; gloo between them)
        from pysdm_amd.engine import HipEngine

        engine = HipEngine.get()
    else:
        from oracle.engine import OracleEngine

        engine = OracleEngine.get()
    for number, case in enumerate(cases):
        try:
            if int(np.prod(case["grid"])) < world:
                continue
            if rank == 0:
                print(number, case, flush=True)
            stats = displacement_cases.random_flow_pair_equal(engine, rank, world, **case)
            if rank == 0:
                print("  ", stats, flush=True)
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
    parser.add_argument("--engine", default="oracle", choices=("oracle", "hip"))
    parser.add_argument("--batch", type=int, default=-1, help="only this batch of ten cases")
    parser.add_argument("--batch-timeout", type=float, default=200.0)
    args = parser.parse_args()
    rng = np.random.default_rng(args.seed)
    failed = 0
    done = 0
    number = -1
    while done < args.cases:
        world = args.world or int(rng.integers(2, 5))
        batch = [draw_case(rng) for _ in range(min(10, args.cases - done))]
        done += len(batch)
        number += 1
        if args.batch >= 0 and number != args.batch:
            continue
        print(f"batch {number}: {len(batch)} cases on {world} ranks", flush=True)
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        ctx = mp.get_context("spawn")
        failures = ctx.Queue()
        procs = [ctx.Process(target=worker, args=(r, world, port, batch, failures, args.engine))
                 for r in range(world)]
        for proc in procs:
            proc.start()
        # a rank that failed leaves its peers waiting in a collective: report it at once, end them
        waited = 0.0
        while any(proc.is_alive() for proc in procs) and waited < args.batch_timeout:
            if not failures.empty():
                break
            time.sleep(0.5)
            waited += 0.5
        stuck = [proc for proc in procs if proc.is_alive()]
        if stuck and failures.empty():
            time.sleep(1.0)
        while not failures.empty():
            rank, index, case, text = failures.get()
            failed += 1
            print(f"FAILED batch {number} world {world} rank {rank} case {index}: {case}\n{text}",
                  flush=True)
        for proc in procs:
            proc.join(5 if failed else 30)
            if proc.is_alive():
                if not failed:
                    failed += 1
                    print(f"batch {number}: a rank did not finish within "
                          f"{args.batch_timeout} s", flush=True)
                proc.kill()
    print("failed:", failed)
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
