"""debug aid (by hand, on an MI355X): one case of tests/fuzz_sharded_flow.py run sharded beside the
one-process run, lengths compared after the displacement AND after the collision step of every
time step; at the first difference the ids involved are printed.
    python tests/helpers/dbg_sharded_flow_case.py [oracle|hip]"""
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

CASE = {'grid': (7, 7), 'n_sd': 1797, 'seed': 1119409700, 'sedimentation': True, 'explicit': False,
        'courant': 0.05103982044890074, 'steps': 6, 'thin': True, 'adaptive_displacement': True,
        'collisions': True}


def worker(rank, world, port, kind):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pysdm_amd import recipe as R, sharding
    from pysdm_amd.collisions import CollisionRunner
    from pysdm_amd.displacement import DisplacementRunner
    from pysdm_amd.population import Population
    from tests.displacement_cases import locate
    if kind == "hip":
        from pysdm_amd.engine import HipEngine
        engine = HipEngine.get()
    else:
        from oracle.engine import OracleEngine
        engine = OracleEngine.get()
    c = CASE
    grid, n_sd, seed = c["grid"], c["n_sd"], c["seed"]
    rng = np.random.default_rng(seed)
    dims = len(grid)
    size = tuple(100.0 * g for g in grid)
    positions = rng.uniform(0, 1, (dims, n_sd)) * np.asarray(grid).reshape(dims, 1)
    volume = rng.uniform(1e-13, 1e-10, n_sd)
    multiplicity = rng.integers(1, 4, n_sd).astype(np.int64)
    field = tuple(rng.uniform(-c["courant"], c["courant"],
                              tuple(g + (1 if axis == d else 0) for axis, g in enumerate(grid)))
                  for d in range(dims))

    def build():
        cell_id, cell_origin, position_in_cell = locate(positions, grid)
        population = Population(engine, multiplicity=multiplicity, volume=volume, cell_id=cell_id,
                                grid=grid, cell_origin=cell_origin,
                                position_in_cell=position_in_cell)
        displacement = DisplacementRunner(
            population, dt=10.0, size=size, enable_sedimentation=c["sedimentation"],
            adaptive=c["adaptive_displacement"], precipitation_counting_level_index=0,
            scheme="ExplicitInSpace" if c["explicit"] else "ImplicitInSpace")
        dv = float(np.prod(np.asarray(size) / np.asarray(grid))) * 1e-6
        runner = CollisionRunner(population, R.CollisionSetup.coalescence(
            R.Geometric(), adaptive=True, seed=seed % 1000), dt=10.0, dv=dv)
        displacement.set_courant(field)
        return population, displacement, runner

    pop, single_d, single_c = build()
    pop_s, shard_d, shard_c = build()
    part = sharding.attach(shard_c, rank, world).shard
    sharding.attach_displacement(shard_d, part)
    down = engine.download

    def compare(step, stage):
        pop.compact()
        pop_s.compact()
        whole = sharding.gather_population(part, pop_s)
        length = pop.live
        live = down(pop.perm)[:length]
        theirs = whole["idx"][:int(whole["length"])]
        same = int(whole["length"]) == length and np.array_equal(theirs, live)
        if rank == 0:
            print(f"step {step} {stage}: one-process {length}, sharded {int(whole['length'])}, "
                  f"{'equal' if same else 'DIFFERENT'}", flush=True)
        if not same and rank == 0:
            print("  one-process control words:", down(pop.ctl), "healthy word", down(pop.healthy),
                  "flagged entries in its live range at positions", np.nonzero(live >= pop.n_sd)[0],
                  "cell_start tail", down(pop.cell_start)[-4:])
            print("  one-process collision counters: rate", down(single_c.collision_rate).sum()
                  if hasattr(single_c, "collision_rate") else None,
                  "substeps", down(pop.stats_n_substep)[:8] if hasattr(pop, "stats_n_substep") else None)
            live = live[live < pop.n_sd]
            only_one = np.setdiff1d(live, theirs)
            only_sharded = np.setdiff1d(theirs, live)
            cells = down(pop.cell_id)
            n1, ns = down(pop.multiplicity), whole["multiplicity"]
            print("  alive in the one-process run only:", only_one, "cells", cells[only_one],
                  "n", n1[only_one], "sharded n", ns[only_one])
            print("  alive in the sharded run only:", only_sharded, "cells (one-process)",
                  cells[only_sharded], "cells (sharded)", whole["cell_id"][only_sharded],
                  "n one-process", n1[only_sharded], "sharded n", ns[only_sharded])
            first = int(np.argmax(theirs[:min(len(theirs), length)] != live[:min(len(theirs), length)])) \
                if len(theirs) and length else -1
            print("  first differing position:", first, "cell_start one-process",
                  down(pop.cell_start)[:8], "...")
            for key, column in (("multiplicity", pop.multiplicity),):
                both = np.intersect1d(live, theirs)
                bad = both[whole[key][both] != down(column)[both]]
                print("  ids alive in both with different", key, ":", bad, down(column)[bad],
                      whole[key][bad], "cells", cells[bad])
        return same

    for step in range(1, c["steps"] + 1):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            variant = os.environ.get("DBG_VARIANT", "")
            single_d.run()
            if variant == "a_first":  # A's whole time step before B's
                single_c.run(1)
                if rank == 0:
                    print(f"step {step}: A alone first: ctl {down(pop.ctl)}", flush=True)
                shard_d.run()
                shard_c.run(1)
                ok = compare(step, "after both")
                if not ok:
                    break
                continue
            if variant == "sync":
                engine.call("sdm_ctx_synchronize")
            shard_d.run()
            if variant == "sync":
                engine.call("sdm_ctx_synchronize")
            ok = compare(step, "after displacement")
            stats = np.zeros(8, dtype=np.int64)
            engine.call("sdm_ctx_read_stats", stats, 1)
            single_c.run(1)
            engine.call("sdm_ctx_read_stats", stats, 1)
            if rank == 0:
                print(f"step {step}: one-process collision step: stats {stats} (closed form, refused, "
                      f"skipped, counting sort, sub-steps, taken back, ..), ctl {down(pop.ctl)}",
                      flush=True)
            shard_c.run(1)
            ok = compare(step, "after collisions") and ok
        if not ok:
            break
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    kind = sys.argv[1] if len(sys.argv) > 1 else "hip"
    mp.spawn(worker, args=(2, 29640, kind), nprocs=2, join=True)
