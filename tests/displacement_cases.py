"""Re-runs the displacement goldens (tests/golden/traj_disp*.npz, produced by the reference's
Displacement dynamic, gen_golden.py:gen_displacement) on a given engine and route: advection by a
random Courant field in 1/2/3-D, sedimentation with precipitation removal, and displacement ahead
of the collision step in a 2-D box.  Integers (cell origin, cell id, idx, multiplicity, length)
must be identical; positions, masses and the precipitated mass within 1e-12."""
import os
import warnings

import numpy as np

from pysdm_amd import recipe as R
from pysdm_amd.collisions import CollisionRunner
from pysdm_amd.displacement import DisplacementRunner
from pysdm_amd.population import Population, locate

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ("disp1d_implicit_sed", "disp2d_implicit_sed", "disp2d_explicit", "disp3d_implicit",
         "disp2d_collide")


def run_case(name, engine, route="fused", shard=None, owner_moves=False, group=None):
    """`shard` = (rank, world): the collision step sharded over the processes; the displacement
    step either replicated on the state completed from the owners (`complete_state`), or - with
    `owner_moves` - sharded as well: every process moves the super-droplets of its own cells and
    hands over those that leave them (sdm_displacement_step_sharded), and the state compared with
    the golden is the one gathered from the owners.  Returns the displacement runner."""
    gold = np.load(os.path.join(GOLDEN, f"traj_{name}.npz"))
    n_sd, dt, explicit, sed, adaptive, collide, steps = gold["cfg"]
    steps = int(steps)
    grid = tuple(int(g) for g in gold["grid"])
    size = tuple(float(v) for v in gold["size"])
    cell_id, cell_origin, position_in_cell = locate(gold["init/positions"], grid)
    population = Population(engine, multiplicity=gold["init/multiplicity"],
                            volume=gold["init/volume"], cell_id=cell_id, grid=grid,
                            cell_origin=cell_origin, position_in_cell=position_in_cell)
    assert population.n_sd == int(n_sd)
    displacement = DisplacementRunner(
        population, dt=float(dt), size=size, enable_sedimentation=bool(sed),
        adaptive=bool(adaptive), precipitation_counting_level_index=0,
        scheme="ExplicitInSpace" if explicit else "ImplicitInSpace", route=route)
    collisions = None
    if collide:
        dv = float(np.prod(np.asarray(size) / np.asarray(grid)))
        collisions = CollisionRunner(
            population, R.CollisionSetup.coalescence(R.Geometric(), adaptive=True, seed=44),
            dt=float(dt), dv=dv, route=route)
    part = None
    if shard is not None:
        from pysdm_amd import sharding  # pylint: disable=import-outside-toplevel

        if collisions is not None:
            part = sharding.attach(collisions, *shard, group=group).shard
        else:
            part = sharding.Shard(engine, population.n_sd, population.n_cell, *shard, group=group)
        if owner_moves:
            sharding.attach_displacement(displacement, part)
    displacement.set_courant(tuple(gold[f"courant/{d}"] for d in range(len(grid))))
    assert displacement.n_substeps == int(gold["n_substeps"])
    down = engine.download
    for step in range(1, steps + 1):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            displacement.run()
            if collisions is not None:
                collisions.run(1)
                if shard is not None and not owner_moves:
                    sharding.complete_state(collisions)
        population.compact()
        tag = f"{name} step {step}"
        length = population.live
        assert length == int(gold[f"step{step}/length"]), tag
        if owner_moves:
            whole = sharding.gather_population(part, population)
            _compare(whole, length, displacement.precipitation_mass_in_last_step, gold, step, tag)
            continue
        live = down(population.perm)[:length]
        np.testing.assert_array_equal(live, gold[f"step{step}/idx"][:length], err_msg=tag)
        np.testing.assert_allclose(displacement.precipitation_mass_in_last_step,
                                   float(gold[f"step{step}/precipitation"]), rtol=1e-12,
                                   err_msg=tag)
        for column, short, exact in ((population.cell_origin, "cell_origin", True),
                                     (population.cell_id, "cell_id", True),
                                     (population.multiplicity, "multiplicity", True),
                                     (population.position_in_cell, "position", False),
                                     (population.mass, "mass", False)):
            # super-droplets that left the domain keep whatever they held: compare the live ones
            actual, expected = down(column)[..., live], gold[f"step{step}/{short}"][..., live]
            if exact:
                np.testing.assert_array_equal(actual, expected, err_msg=f"{tag} {short}")
            else:
                np.testing.assert_allclose(actual, expected, rtol=1e-12, atol=1e-13,
                                           err_msg=f"{tag} {short}")
    return displacement


def _compare(whole, length, precipitation, gold, step, tag):
    """the state gathered from the owners of a sharded run against the golden"""
    live = whole["idx"][:length]
    np.testing.assert_array_equal(live, gold[f"step{step}/idx"][:length], err_msg=tag)
    np.testing.assert_allclose(precipitation, float(gold[f"step{step}/precipitation"]),
                               rtol=1e-12, err_msg=tag)
    for key, short, exact in (("cell_origin", "cell_origin", True), ("cell_id", "cell_id", True),
                              ("multiplicity", "multiplicity", True),
                              ("position_in_cell", "position", False)):
        actual, expected = whole[key][..., live], gold[f"step{step}/{short}"][..., live]
        if exact:
            np.testing.assert_array_equal(actual, expected, err_msg=f"{tag} {short}")
        else:
            np.testing.assert_allclose(actual, expected, rtol=1e-12, atol=1e-13,
                                       err_msg=f"{tag} {short}")
    np.testing.assert_allclose(whole["attributes"][0][live], gold[f"step{step}/mass"][live],
                               rtol=1e-12, atol=1e-13, err_msg=f"{tag} mass")



def sharded_flow_equals_single(engine, rank, world, *, n_sd, grid, steps, group=None):
    """the 2-D kinematic set-up (single-eddy flow + sedimentation, then adaptive Geometric
    coalescence: pysdm_amd.cases.make_kinematic_flow) with BOTH steps sharded, beside the
    one-process run on the same engine: after every step the state gathered from the owners must
    be the one-process state - ids, positions in the permutation, multiplicities, attributes,
    cells, positions and the rainfall, to the bit.  Returns the exchange statistics."""
    from pysdm_amd import cases, sharding  # pylint: disable=import-outside-toplevel

    size = (1500.0, 1500.0)
    single_d, single_c = cases.make_kinematic_flow(engine, n_sd=n_sd, grid=grid, size=size)
    shard_d, shard_c = cases.make_kinematic_flow(engine, n_sd=n_sd, grid=grid, size=size)
    part = sharding.attach(shard_c, rank, world, group=group).shard
    sharding.attach_displacement(shard_d, part)
    down = engine.download
    for step in range(1, steps + 1):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            rain = (single_d.run(), shard_d.run())
            single_c.run(1)
            shard_c.run(1)
        assert rain[1] == rain[0], f"step {step} rain: {rain[1]!r} != {rain[0]!r}"  # to the bit
        pop = single_c.population
        whole = sharding.gather_population(part, shard_c.population)
        length = pop.live
        assert int(whole["length"]) == length, f"step {step}"
        live = down(pop.perm)[:length]
        np.testing.assert_array_equal(whole["idx"][:length], live, err_msg=f"step {step} idx")
        asked = np.zeros(pop.n_sd, dtype=bool)  # pair numbers, and whoever is alive
        asked[: (pop.n_sd + 1) // 2] = True
        asked[live] = True
        np.testing.assert_array_equal(whole["cell_id"][asked], down(pop.cell_id)[asked],
                                      err_msg=f"step {step} cells by id")
        for key, column in (("multiplicity", pop.multiplicity), ("attributes", pop.extensive),
                            ("cell_origin", pop.cell_origin),
                            ("position_in_cell", pop.position_in_cell)):
            np.testing.assert_array_equal(whole[key][..., live], down(column)[..., live],
                                          err_msg=f"step {step} {key}")
        assert shard_c.sub_steps_done == single_c.sub_steps_done, f"step {step}"
    stats = dict(shard_d.shard_stats)
    stats["collision_exchange_bytes"] = sum(part.bytes.values())
    return stats


# found by tests/fuzz_sharded_flow.py on the GPU: everything precipitates or coalesces away by the
# fifth step, and the sharded collision step refused the empty state ("largest cell unknown")
DIES_OUT = {'grid': (2,), 'n_sd': 238, 'seed': 1293093149, 'sedimentation': True, 'explicit': True, 'courant': 0.29784313388427763, 'steps': 6, 'thin': True, 'adaptive_displacement': True, 'collisions': True}


def random_flow_pair_equal(engine, rank, world, *, grid, n_sd, seed, sedimentation, explicit,
                           courant, steps, thin, adaptive_displacement, collisions, group=None):
    """a random set-up (tests/fuzz_sharded_flow.py) run sharded - displacement and collisions -
    beside the one-process run: equal after every step, as in sharded_flow_equals_single"""
    from pysdm_amd import sharding  # pylint: disable=import-outside-toplevel

    rng = np.random.default_rng(seed)
    dims = len(grid)
    size = tuple(100.0 * g for g in grid)
    positions = rng.uniform(0, 1, (dims, n_sd)) * np.asarray(grid).reshape(dims, 1)
    volume = rng.uniform(1e-15, 1e-12, n_sd) if not thin else rng.uniform(1e-13, 1e-10, n_sd)
    multiplicity = (rng.integers(1, 4, n_sd) if thin
                    else rng.integers(1000, 100000, n_sd)).astype(np.int64)
    field = tuple(rng.uniform(-courant, courant, tuple(g + (1 if axis == d else 0)
                                                       for axis, g in enumerate(grid)))
                  for d in range(dims))

    def build():
        cell_id, cell_origin, position_in_cell = locate(positions, grid)
        population = Population(engine, multiplicity=multiplicity, volume=volume, cell_id=cell_id,
                                grid=grid, cell_origin=cell_origin,
                                position_in_cell=position_in_cell)
        displacement = DisplacementRunner(
            population, dt=10.0, size=size, enable_sedimentation=sedimentation,
            adaptive=adaptive_displacement, precipitation_counting_level_index=0,
            scheme="ExplicitInSpace" if explicit else "ImplicitInSpace")
        runner = None
        if collisions:
            dv = float(np.prod(np.asarray(size) / np.asarray(grid))) * (1e-6 if thin else 1.0)
            runner = CollisionRunner(
                population, R.CollisionSetup.coalescence(R.Geometric(), adaptive=True,
                                                         seed=seed % 1000),
                dt=10.0, dv=dv)
        displacement.set_courant(field)
        return population, displacement, runner

    pop, single_d, single_c = build()
    pop_s, shard_d, shard_c = build()
    if shard_c is not None:
        part = sharding.attach(shard_c, rank, world, group=group).shard
    else:
        part = sharding.Shard(engine, pop_s.n_sd, pop_s.n_cell, rank, world, group=group)
    sharding.attach_displacement(shard_d, part)
    down = engine.download
    for step in range(1, steps + 1):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            rain = (single_d.run(), shard_d.run())
            if single_c is not None:
                single_c.run(1)
                shard_c.run(1)
        assert rain[1] == rain[0], f"step {step} rain: {rain[1]!r} != {rain[0]!r}"  # to the bit
        pop.compact()
        pop_s.compact()
        whole = sharding.gather_population(part, pop_s)
        length = pop.live
        assert int(whole["length"]) == length, f"step {step}: {int(whole['length'])} != {length}"
        live = down(pop.perm)[:length]
        np.testing.assert_array_equal(whole["idx"][:length], live, err_msg=f"step {step} idx")
        asked = np.zeros(pop.n_sd, dtype=bool)  # pair numbers, and whoever is alive
        asked[: (pop.n_sd + 1) // 2] = True
        asked[live] = True
        np.testing.assert_array_equal(whole["cell_id"][asked], down(pop.cell_id)[asked],
                                      err_msg=f"step {step} cells by id")
        for key, column in (("multiplicity", pop.multiplicity), ("attributes", pop.extensive),
                            ("cell_origin", pop.cell_origin),
                            ("position_in_cell", pop.position_in_cell)):
            np.testing.assert_array_equal(whole[key][..., live], down(column)[..., live],
                                          err_msg=f"step {step} {key}")
    stats = dict(shard_d.shard_stats)
    stats["live"] = int(pop.live)
    return stats
