"""Full-size states pinned to the REFERENCE (not only to the oracle): tests/golden/digest_*.npz
hold, for runs of the reference itself at sizes whose arrays cannot be committed, SHA-256 digests
of the permutation / multiplicity / mass columns, fp64 moments and the per-cell counters
(tests/golden/gen_golden.py: gen_digests).  The initial states are
closed-form functions of the case (pysdm_amd.cases) and are themselves pinned by digest."""
import glob
import hashlib
import os
import warnings

import numpy as np

from pysdm_amd import cases

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sha(array):
    return hashlib.sha256(np.ascontiguousarray(array).tobytes()).hexdigest()


def available():
    return sorted(os.path.basename(p)[len("digest_"):-4]
                  for p in glob.glob(os.path.join(GOLDEN, "digest_*.npz")))


def n_sd_of(name):
    if name.startswith("kinematic2d"):
        return 1024 * int(name.split("_")[1].replace("percell", ""))
    return int(name.split("_n")[1].split("_")[0])


def runner_for(name, engine, route="fused"):
    n_sd = n_sd_of(name)
    if name.startswith("shima"):
        # (the Shima box keeps dv = 1e6 m3 at every n_sd: configs[0] is the 2^14 one)
        return cases.make_box(engine, "shima", n_sd=n_sd, adaptive=name.endswith("a1"),
                              route=route, dv=1e6)
    case = name.split("_n")[0] if not name.startswith("kinematic2d") else "kinematic2d"
    return cases.make_box(engine, case, n_sd=n_sd, route=route)


def check(name, engine, route="fused", float_exact=None, steps_limit=None, prepare=None,
          snapshot=None):
    """runs case `name` and compares with the reference's digest after each recorded step.
    Integer columns by SHA-256; masses by SHA-256 where the path is free of transcendental
    functions feeding attributes (coalescence), else moments at 1e-12"""
    gold = np.load(os.path.join(GOLDEN, f"digest_{name}.npz"))
    runner = runner_for(name, engine, route)
    if prepare is not None:  # e.g. pysdm_amd.sharding.attach
        prepare(runner)
    pop, down = runner.population, engine.download
    breakup = runner.setup.breakup
    float_exact = (not breakup) if float_exact is None else float_exact
    rho_w = pop.rho_w
    # the initial state is the reference's, bit for bit
    assert sha(down(pop.multiplicity)) == str(gold["init/sha_multiplicity"]), "initial multiplicity"
    if "init/sha_cell_id" in gold.files:
        assert sha(down(pop.cell_id)) == str(gold["init/sha_cell_id"]), "initial cell ids"
    for step in (int(s) for s in gold["record_steps"]):
        if steps_limit is not None and step > steps_limit:
            break
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            runner.run(step - runner.steps_done)
        snap = runner.snapshot() if snapshot is None else snapshot(runner)
        tag = f"{name} step {step}"
        length = int(snap["length"])
        assert length == int(gold[f"step{step}/length"]), tag
        idx = snap["idx"][:length]
        n, mass = snap["multiplicity"], snap["attributes"][0]
        assert sha(idx) == str(gold[f"step{step}/sha_idx"]), tag + " idx"
        assert sha(n) == str(gold[f"step{step}/sha_multiplicity_raw"]), tag + " multiplicity"
        assert sha(snap["cell_start"]) == str(gold[f"step{step}/sha_cell_start"]), tag
        if float_exact:
            assert sha(mass) == str(gold[f"step{step}/sha_mass_raw"]), tag + " mass"
        live_n, vol = n[idx].astype(np.float64), mass[idx] / rho_w
        moments = np.asarray([np.sum(live_n * vol**k) for k in range(4)])
        np.testing.assert_allclose(moments, gold[f"step{step}/moments"], rtol=1e-12, err_msg=tag)
        np.testing.assert_allclose(np.sum(live_n * mass[idx]), gold[f"step{step}/total_mass"],
                                   rtol=1e-12, err_msg=tag)
        for key in ("collision_rate", "collision_rate_deficit", "coalescence_rate", "breakup_rate",
                    "breakup_rate_deficit", "stats_n_substep"):
            if f"step{step}/{key}" in gold.files:
                np.testing.assert_array_equal(snap[key], gold[f"step{step}/{key}"],
                                              err_msg=f"{tag} {key}")
    return runner
