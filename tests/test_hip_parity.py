"""GPU parity tests proper: the HIP path (through the C ABI) against the goldens recorded from the
reference and against the oracle on the same seeded inputs.  Integer state bit-exact; floating
point bit-exact on the coalescence-only paths and - against the checker - on the breakup paths as
well (both sides evaluate pow / exp / log / erf with csrc/sdm_math.h); 1e-12 relative against the
REFERENCE's values where transcendental functions feed attributes (its NumPy / libm are not
ours)."""
import warnings

import numpy as np
import pytest

from pysdm_amd import recipe as R
from pysdm_amd.cases import make_box
from pysdm_amd.collisions import CollisionRunner
from pysdm_amd.population import Population

from . import displacement_cases
from . import known_answers as ka
from . import micro_cases as mc
from .trajectory import golden_files, run_and_compare, setup_from_golden

pytestmark = pytest.mark.gpu

EXACT_ON_GPU = {"volume", "golovin", "frag_always_n_4"}
ROUTES = ("chain", "fused")


@pytest.fixture(scope="module", name="kit")
def kit_fixture(hip_backend_class):
    return mc.Kit(hip_backend_class, fragmentation_function="Straub2010Nf")


@pytest.mark.parametrize("check", [mc.check_pcg64, mc.check_shuffle,
                                   mc.check_shuffle_known_answers, mc.check_counting_sort,
                                   mc.check_sort_by_key_and_adaptive_end, mc.check_remove_zero, mc.check_sanitize_sorted,
                                   mc.check_pair_chain, mc.check_moments, mc.check_moments_goldens,
                                   mc.check_storage_ops])
def test_method_goldens(check, kit):
    check(kit)


def test_physics_goldens(kit):
    mc.check_physics(kit, exact=EXACT_ON_GPU, rtol=1e-13)


COALESCENCE = (golden_files("traj_golovin_*.npz") + golden_files("traj_geometric_*.npz")
               + golden_files("traj_multicell_*.npz") + golden_files("traj_kernel_*.npz"))


def assert_same(a, b, float_rtol=0.0):
    length = int(a["length"])
    for key, value in a.items():
        ref = b[key]
        if key == "idx":  # beyond `length`: dead storage (see trajectory.compare)
            value, ref = value[:length], ref[:length]
        if float_rtol and value.dtype.kind == "f":
            np.testing.assert_allclose(value, ref, rtol=float_rtol, atol=0, err_msg=key)
        else:
            np.testing.assert_array_equal(value, ref, err_msg=key)


# Several cells + global croupier + adaptive sub-stepping: once a sub-step works on a cut working
# length, the reference's counting sort writes only that range into the spare buffer and SWAPS the
# buffers (collisions_methods.py:587-631) - the range beyond the cut then holds whatever that buffer
# held before: super-droplets are duplicated and lost (ids appear twice in `idx`), and what a
# duplicated id's two pairs do to it depends on the order of execution.  The serial checker
# reproduces the reference there (tests/test_oracle_trajectories.py runs these goldens to the end);
# a parallel backend cannot, and this one keeps the range beyond the cut intact instead.  Parity is
# asserted up to the last recorded step before the first cut.
STEPS_BEFORE_THE_REFERENCE_DUPLICATES_IDS = {"traj_multicell_geometric_4x4_global": 3,
                                             "traj_multicell_geometric_8x8_global_small_cells": 3}


@pytest.mark.parametrize("route", ROUTES)
@pytest.mark.parametrize("name", COALESCENCE)
def test_coalescence_trajectories_bit_exact(name, route, hip_engine):
    run_and_compare(name, hip_engine, route=route,
                    max_step=STEPS_BEFORE_THE_REFERENCE_DUPLICATES_IDS.get(name))


@pytest.mark.parametrize("route", ROUTES)
@pytest.mark.parametrize("name", golden_files("traj_breakup_*.npz"))
def test_breakup_trajectories(name, route, hip_engine):
    run_and_compare(name, hip_engine, route=route, float_rtol=1e-12)


@pytest.mark.parametrize("route", ROUTES)
@pytest.mark.parametrize("name", golden_files("traj_breakup_*.npz"))
def test_breakup_trajectories_equal_the_oracle_to_the_bit(name, route, hip_engine, oracle_engine):
    """the reference's values are reproduced to 1e-12 (its NumPy / libm transcendentals are not
    ours); the checker's are reproduced EXACTLY - efficiencies, fragment sizes and masses come from
    csrc/sdm_math.h on both sides - and, on the fused route, over four times the golden's length
    (the stage-by-stage route checks radii against the Gunn-Kinzer table's range like the
    reference and would refuse the drops that grow beyond it)"""
    snaps = []
    for engine in (hip_engine, oracle_engine):
        runner, _, steps = setup_from_golden(name, engine, route=route)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            runner.run(steps[-1])
            if route == "fused":
                runner.run(3 * steps[-1])
        snaps.append(runner.snapshot())
    assert_same(snaps[0], snaps[1])


@pytest.mark.parametrize("name", ["traj_golovin_n4096_s44_a1", "traj_multicell_geometric_4x4"])
def test_fused_equals_oracle_beyond_goldens(name, hip_engine, oracle_engine):
    """same seeded inputs, more steps than the goldens hold"""
    snaps = []
    for engine in (hip_engine, oracle_engine):
        runner, _, _ = setup_from_golden(name, engine)
        runner.run(120)
        snaps.append(runner.snapshot())
    assert_same(snaps[0], snaps[1])


@pytest.mark.parametrize("check", ka.ALL_CHECKS)
def test_reference_known_answers(check, kit):
    check(kit)


@pytest.mark.parametrize("route", ROUTES)
@pytest.mark.parametrize("name", displacement_cases.CASES)
def test_displacement_goldens(name, route, hip_engine):
    displacement_cases.run_case(name, hip_engine, route=route)


def test_c_abi_example_equals_python_route(tmp_path, hip_engine):
    """examples/shima_box_c_abi.cpp: the Shima box driven through include/sdm_hip.h from a plain
    C++ program (no Python, no torch) gives bit for bit what the Python host code gives"""
    import os  # pylint: disable=import-outside-toplevel
    import struct  # pylint: disable=import-outside-toplevel
    import subprocess  # pylint: disable=import-outside-toplevel

    from pysdm_amd.abi import pcg64_state_inc  # pylint: disable=import-outside-toplevel

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    binary = os.path.join(root, "examples", "shima_box_c_abi")
    library = os.path.join(root, "pysdm_amd", "libsdm_hip.so")
    if not os.path.exists(binary) or os.path.getmtime(binary) < os.path.getmtime(library):
        subprocess.check_call(["bash", os.path.join(root, "examples", "build.sh")])
    n_sd, n_steps, seed = 2**14, 25, 44
    runner = make_box(hip_engine, "shima", n_sd=n_sd, seed=seed)
    pop = runner.population
    multiplicity, mass = hip_engine.download(pop.multiplicity), hip_engine.download(pop.mass)
    source, result = tmp_path / "box.in", tmp_path / "box.out"
    with open(source, "wb") as handle:
        handle.write(struct.pack("<qq", n_sd, n_steps))
        handle.write(struct.pack("<ddd", runner.dt, runner.dv, 1.5e3))
        handle.write(struct.pack("<4Q", *pcg64_state_inc(seed)))
        handle.write(multiplicity.astype("<i8").tobytes())
        handle.write(mass.astype("<f8").tobytes())
    subprocess.check_call([binary, str(source), str(result)])
    raw = np.fromfile(result, dtype=np.uint8)
    n_live = int(raw[:8].view("<i8")[0])
    idx = raw[8:8 + 8 * n_sd].view("<i8")
    mult_c = raw[8 + 8 * n_sd:8 + 16 * n_sd].view("<i8")
    mass_c = raw[8 + 16 * n_sd:8 + 24 * n_sd].view("<f8")
    runner.run(n_steps)
    snap = runner.snapshot()
    assert n_live == int(snap["length"])
    np.testing.assert_array_equal(idx[:n_live], snap["idx"][:n_live])
    np.testing.assert_array_equal(mult_c, snap["multiplicity"])
    np.testing.assert_array_equal(mass_c, snap["attributes"][0])


def _shima_like(engine, setup, route, n_sd=2**12):
    from pysdm_amd.cases import initial_state  # pylint: disable=import-outside-toplevel

    volume, multiplicity, _, dv, _ = initial_state("shima", n_sd)
    pop = Population(engine, multiplicity=multiplicity, volume=volume)
    return CollisionRunner(pop, setup, dt=1.0, dv=dv, route=route)


def test_linear_and_constant_kernels_agree_across_routes(hip_engine, oracle_engine):
    """the Linear kernel cannot run in the reference (a stub there) and ConstantK has no golden:
    fused route, chain route and the oracle agree bit for bit on a Shima-type box"""
    for kernel in (R.Linear(a=2e-9, b=1.5e3), R.ConstantK(a=2e-9)):
        snaps = []
        for engine, route in ((hip_engine, "fused"), (hip_engine, "chain"),
                              (oracle_engine, "fused")):
            runner = _shima_like(engine, R.CollisionSetup.coalescence(kernel, adaptive=True,
                                                                      seed=44), route)
            runner.run(30)
            snaps.append(runner.snapshot())
        assert snaps[2]["collision_rate"].sum() > 0
        for other in snaps[1:]:
            assert_same(snaps[0], other)


def test_overflow_warnings_step_by_step(hip_engine, oracle_engine):
    """breakups refused for multiplicity overflow (collisions_methods.py:113-118) raise the
    reference's "overflow" warning in the very step they happen - on the fused route the count
    travels through the control block - and leave the same state behind"""
    rng = np.random.default_rng(3)
    n_sd = 2048
    volume = rng.exponential(1.2e-13, n_sd)
    multiplicity = rng.integers(10**6, 10**7, n_sd)
    outcomes = []
    for engine, route in ((hip_engine, "fused"), (hip_engine, "chain"), (oracle_engine, "fused")):
        pop = Population(engine, multiplicity=multiplicity.copy(), volume=volume.copy())
        setup = R.CollisionSetup.collision(R.Golovin(b=1.5e3), R.ConstEc(Ec=0.2), R.ConstEb(1.0),
                                           R.AlwaysN(n=3e6), adaptive=True, warn_overflows=True,
                                           seed=5)
        runner = CollisionRunner(pop, setup, dt=1.0, dv=2e3, route=route)
        warned = []
        for _ in range(12):
            with warnings.catch_warnings(record=True) as caught:
                warnings.simplefilter("always")
                runner.run(1)
            warned.append(any("overflow" in str(w.message) for w in caught))
        outcomes.append((warned, runner.snapshot()))
    assert any(outcomes[2][0]) and not all(outcomes[2][0]), outcomes[2][0]
    for warned, snap in outcomes[:2]:
        assert warned == outcomes[2][0]
        assert_same(snap, outcomes[2][1])


@pytest.mark.parametrize("base,adaptive,grid,n_sd", [
    ("straub", True, (4, 4), 2**13), ("berry_breakup", True, (4, 4), 2**13),
    ("straub", False, (4, 4), 2**13), ("straub_rain", True, (4, 4), 2**13),
    ("straub", True, (2, 2), 4 * 5850),  # cells above k_cell_step2's cap: k_cell_step<.., true>
])
def test_multicell_breakup_equals_oracle(base, adaptive, grid, n_sd, hip_engine, oracle_engine):
    """breakup on a grid (the reference has no such golden): the per-cell kernels' listing of
    colliding pairs + the dense resolution, sub-steps launched ahead of the read-back, per-cell
    counters - against the oracle"""
    snaps = []
    for engine in (hip_engine, oracle_engine):
        runner = make_box(engine, base, n_sd=n_sd, adaptive=adaptive, dt=5.0, grid=grid)
        for steps in ((1, 12, 3) if n_sd == 2**13 else (1, 3, 1)):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                runner.run(steps)
        snaps.append(runner.snapshot())
    assert snaps[1]["breakup_rate"].sum() > 0
    assert_same(snaps[0], snaps[1])  # to the bit: one transcendental library on both sides


@pytest.mark.parametrize("n_sd,which", [(4 * 5850, "one workgroup per CU"),
                                        (4 * 4000, "two workgroups per CU")])
def test_both_per_cell_kernels_equal_oracle(n_sd, which, hip_engine, oracle_engine):
    """2 x 2 cells of ~5850 super-droplets take k_cell_step (cells above k_cell_step2's 5632),
    cells of ~4000 take k_cell_step2: adaptive geometric coalescence with optimized_random,
    state and counters equal the oracle's"""
    snaps = []
    for engine in (hip_engine, oracle_engine):
        runner = make_box(engine, "kinematic2d", n_sd=n_sd, grid=(2, 2))
        for steps in (1, 8, 2):
            runner.run(steps)
        snaps.append(runner.snapshot())
    sizes = np.diff(snaps[1]["cell_start"])
    assert (sizes.max() > 5632) == (which == "one workgroup per CU") and sizes.max() <= 6144
    assert snaps[1]["collision_rate"].sum() > 0
    assert_same(snaps[0], snaps[1])


@pytest.mark.parametrize("option,cases", [
    # SDM_OPT_REC_FORMAT = records: the packed records where successor words are the default
    (3, [("shima", dict(n_sd=2**15, adaptive=False), (1, 6, 2)),
         ("shima", dict(n_sd=2**15, adaptive=True, thin=0.02, dt=50.0), (1, 6, 2)),
         ("berry_breakup", dict(n_sd=2**14, adaptive=True), (1, 5, 2))]),
    # SDM_OPT_NO_PRESORT: k_bin_sort and the compaction as launches of their own in every step
    (4, [("shima", dict(n_sd=2**15, adaptive=False), (1, 6, 2)),
         ("shima", dict(n_sd=2**15, adaptive=False, thin=0.02, dt=50.0), (1, 6, 2))]),
    # SDM_OPT_NO_CELL_COPY: no cell-ordered working copy in runs of three steps or more
    (5, [("kinematic2d", dict(n_sd=2**13, grid=(4, 4)), (1, 6, 4)),
         ("shima", dict(n_sd=2**13, adaptive=True, thin=0.02, dt=200.0, grid=(4, 4)), (1, 6, 4))]),
])
def test_both_sides_of_the_ab_options_equal_oracle(option, cases, hip_engine, oracle_engine):
    """the three A/B options of a context select between two implementations of the same step
    (sdm_hip.h): both, in one process, against the oracle - also where super-droplets die"""
    def run(engine, name, kwargs, chunks):
        runner = make_box(engine, name, **kwargs)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for steps in chunks:
                runner.run(steps)
        return runner.snapshot()
    try:
        for name, kwargs, chunks in cases:
            expected = run(oracle_engine, name, kwargs, chunks)
            if "thin" in kwargs:
                assert int(expected["length"]) < kwargs["n_sd"]
            for value in (1, 0):
                hip_engine.call("sdm_ctx_set_option", option, value)
                assert_same(run(hip_engine, name, kwargs, chunks), expected)
    finally:
        hip_engine.call("sdm_ctx_set_option", option, 0)
    with pytest.raises(Exception):
        hip_engine.call("sdm_ctx_set_option", option, 2)


@pytest.mark.parametrize("grid,n_sd", [((48, 48), 2304 * 24), ((96, 96), 9216 * 10)])
def test_grids_of_thousands_of_cells_equal_oracle(grid, n_sd, hip_engine, oracle_engine):
    """k_cells_turn ranks the cells' dt_left eight cells per workgroup above 2048 cells and sixteen
    above 8192 (one below): adaptive geometric coalescence on 2304 and 9216 small cells (the packed
    shape of the cell kernel), state, counters and sub-step statistics equal the oracle's"""
    snaps = []
    for engine in (hip_engine, oracle_engine):
        runner = make_box(engine, "kinematic2d", n_sd=n_sd, grid=grid)
        for steps in (1, 5, 2):
            runner.run(steps)
        snaps.append(runner.snapshot())
    assert snaps[1]["collision_rate"].sum() > 0 and snaps[1]["stats_n_substep"].max() > 8
    assert_same(snaps[0], snaps[1])


@pytest.mark.parametrize("base", ["kinematic2d", "straub"])
def test_every_shape_of_the_cell_kernel_equals_oracle(base, hip_engine, oracle_engine):
    """SDM_OPT_CELL_SHAPE: 2 x 2 cells of ~2000 fit all three shapes of k_cell_step2 (512, 1024 and
    256 threads per cell); AUTO would take the 1024-thread one here (fewer cells than CUs), so
    the others are forced - state and counters equal the oracle's under each"""
    def run(engine):
        runner = make_box(engine, base, n_sd=4 * 2000, grid=(2, 2), **(
            {} if base == "kinematic2d" else {"adaptive": True, "dt": 5.0}))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for steps in (1, 6, 2):
                runner.run(steps)
        return runner.snapshot()
    expected = run(oracle_engine)
    assert np.diff(expected["cell_start"]).max() <= 2816
    try:
        for shape in (1, 2, 3, 0):
            hip_engine.call("sdm_ctx_set_option", 2, shape)
            assert_same(run(hip_engine), expected)
    finally:
        hip_engine.call("sdm_ctx_set_option", 2, 0)
    with pytest.raises(Exception):
        hip_engine.call("sdm_ctx_set_option", 2, 4)


@pytest.mark.parametrize("base", ["kinematic2d", "straub"])
def test_both_packed_shapes_of_the_cell_kernel_equal_oracle(base, hip_engine, oracle_engine):
    """eight small cells per workgroup: cells of at most 384 take the variant whose per-lane loops
    are sized for them (AUTO), SDM_CELL_SHAPE_512 forces the 704-position one - 6 x 6 cells of
    ~150, both against the oracle"""
    def run(engine):
        runner = make_box(engine, base, n_sd=36 * 150, grid=(6, 6), **(
            {} if base == "kinematic2d" else {"adaptive": True, "dt": 5.0}))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for steps in (1, 6, 2):
                runner.run(steps)
        return runner.snapshot()
    expected = run(oracle_engine)
    sizes = np.diff(expected["cell_start"])
    assert 64 < sizes.max() <= 384
    try:
        for shape in (1, 0):
            hip_engine.call("sdm_ctx_set_option", 2, shape)
            assert_same(run(hip_engine), expected)
    finally:
        hip_engine.call("sdm_ctx_set_option", 2, 0)


def test_degenerate_sizes_and_operand_checks(kit, hip_engine):
    """empty and tiny inputs: zero-length arrays are accepted by every entry point that takes a
    length, two super-droplets form one pair, three leave one alone, a null context is refused,
    and the binding refuses operands of the wrong element type before anything is launched"""
    eng = hip_engine
    for name, args in (
        ("sdm_identity_index", (None, 0)),
        ("sdm_elementwise_f64", (0, None, None, None, 0.0, 0)),
        ("sdm_elementwise_i64", (0, None, None, None, 0, 0)),
        ("sdm_volume_of_water_mass", (None, None, 0, 1000.0)),
        ("sdm_floor_to_i64", (None, None, 0)),
        ("sdm_subtract_i64", (None, None, 0)),
        ("sdm_ll82_coalescence_check", (None, None, 0)),
    ):
        eng.call(name, *args)
    cdll = eng.library.cdll
    assert cdll.sdm_identity_index(None, None, 0) == -1  # SDM_E_ARG, message available
    assert "bad argument" in eng.library.last_error()
    with pytest.raises(TypeError):
        eng.call("sdm_identity_index", eng.zeros(4, np.float64), 4)
    for n_sd in (2, 3):
        idx = kit.Index.identity_index(n_sd)
        u01 = kit.Storage.from_ndarray(np.full(n_sd, 0.75))
        kit.backend.shuffle_global(idx=idx.data, length=n_sd, u01=u01.data)
        assert sorted(idx.to_ndarray().tolist()) == list(range(n_sd))
        flag = kit.PairIndicator(n_sd)
        cell_start = kit.Storage.from_ndarray(np.asarray([0, n_sd]))
        cell_id = kit.IndexedStorage.from_ndarray(idx, np.zeros(n_sd, dtype=np.int64))
        kit.backend.find_pairs(cell_start, flag, cell_id, kit.Index.identity_index(1), idx)
        assert flag.indicator.to_ndarray().sum() == 1


@pytest.mark.parametrize("grid", [None, (4, 4)])
def test_dt_min_event_travels_through_the_control_block(grid, hip_engine, oracle_engine):
    """"adaptive time-step reached dt_min" (collision.py:276-277) on the fused route: the kernels
    that update stats_dt_min set the event bit of control word 7; silent with the NaN initial
    statistics, raised after a reset - one cell (k_cells_adaptive) and a grid (the per-cell
    kernels + cells_end_body) - and the state equals the checker's"""
    for reset in (False, True):
        snaps, warned = [], []
        for engine in (hip_engine, oracle_engine):
            runner = make_box(engine, "shima", n_sd=2**12, adaptive=True, dt=200.0,
                              dt_range=(100.0, 200.0), grid=grid)
            if reset:
                engine.fill(runner.stats_dt_min, 200.0)
            with warnings.catch_warnings(record=True) as caught:
                warnings.simplefilter("always")
                runner.run(1)
                runner.run(2)
            warned.append(any("dt_min" in str(w.message) for w in caught))
            snaps.append(runner.snapshot())
        assert warned == [reset, reset]
        assert_same(snaps[0], snaps[1])
        if reset:
            assert snaps[0]["stats_dt_min"].min() == 100.0


@pytest.mark.parametrize("name", ["traj_golovin_n1024_s44_a1", "traj_multicell_geometric_4x4"])
def test_fused_plugin_on_hip_storages(name, hip_backend_class):
    """`pysdm_plugin.fuse` on the GPU (PySDM does not travel to the GPU box, so PySDM's
    Particulator / ParticleAttributes are played by tests/pysdm_ducks.py, name-mangled members
    and all): FusedCollision.__call__ adopts the torch-backed Storages in place, runs ONE
    sdm_collision_step per call and hands permutation / length / sorted flag back - against the
    reference's goldens"""
    from . import pysdm_ducks  # pylint: disable=import-outside-toplevel

    pysdm_ducks.fused_plugin_run(name, hip_backend_class)


def test_random_sector_calibration_touches_the_bytes_it_claims(hip_engine):
    """the calibration kernel behind `roofline.random_sector_ceiling_gbs`: its checksum equals
    the sum over the records the header says it reads (so the timed launches did read them), and
    the rate it reports is a plausible one for this part"""
    import ctypes  # pylint: disable=import-outside-toplevel

    from .test_abi import expected_calibration_checksum  # pylint: disable=import-outside-toplevel

    table, reads, reps = 2**21, 2**22, 3
    ms, checksum = ctypes.c_double(), ctypes.c_uint64()
    hip_engine.call("sdm_calib_random_sectors", table, reads, reps, ms, checksum)
    assert checksum.value == expected_calibration_checksum(table, reads, reps)
    rate = reads / (ms.value * 1e-3)  # 16-byte reads per second
    assert 5e9 < rate < 1e12, rate


@pytest.mark.parametrize("case", ["coalescence", "deaths", "breakup", "golovin_na"])
def test_cell_ordered_working_copy_changes_nothing(case, hip_engine, oracle_engine):
    """a multi-cell run of several steps in ONE library call works on a cell-ordered copy of the
    state from its second step on (fused.hip: Relabel - call-local labels, scattered back and
    translated at the end); the same steps one call at a time never do.  Both must leave the very
    same state, counters and stream positions - and the checker's - also when super-droplets die
    (compaction + re-sort inside the copy) and with breakup (k_resolve_dense on the copy)."""
    def box(engine):
        if case == "deaths":
            return make_box(engine, "shima", n_sd=2**13, adaptive=True, dt=200.0, thin=0.02,
                            grid=(4, 4))
        if case == "breakup":
            return make_box(engine, "straub_rain", n_sd=2**13, adaptive=True, dt=5.0, grid=(4, 4))
        if case == "golovin_na":  # (non-adaptive: the copy is not used; the route must not care)
            return make_box(engine, "shima", n_sd=2**13, adaptive=False, grid=(4, 4))
        return make_box(engine, "kinematic2d", n_sd=2**15, grid=(4, 4))

    runs = {}
    for label, engine, chunks in (("one call", hip_engine, (9,)),
                                  ("step by step", hip_engine, (1,) * 9),
                                  ("mixed", hip_engine, (2, 4, 3)),
                                  ("checker", oracle_engine, (9,))):
        runner = box(engine)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for steps in chunks:
                runner.run(steps)
        runs[label] = (runner.snapshot(), runner.offset, runner.offset_breakup,
                       runner.sub_steps_done)
    reference = runs["checker"]
    if case == "deaths":
        assert int(reference[0]["length"]) < 2**13
    if case == "breakup":
        assert reference[0]["breakup_rate"].sum() > 0
    for label in ("one call", "step by step", "mixed"):
        assert_same(runs[label][0], reference[0])
        assert runs[label][1:] == reference[1:], label


def _long_breakup_case(engine, handle_all_breakups):
    """pairs whose `break_up` loop (collisions_methods.py:62-132) runs 0 .. 3e5 successive
    breakups: full blocks of the 8-at-a-time loop, exits in the middle of a block (the donor runs
    out, the multiplicity limit is hit), gamma not a multiple of 8"""
    rng = np.random.default_rng(17)
    gammas = np.array([0, 1, 7, 8, 9, 15, 16, 17, 63, 64, 65, 1000, 4099, 30011, 100003, 300007],
                      dtype=float)
    pairs = 4 * len(gammas)
    n_sd = 2 * pairs
    gamma = np.tile(gammas, 4)
    # donors j (even positions) with plenty / few droplets, receivers k
    n_j = np.where(np.arange(pairs) % 4 < 2, 10**15, 3 * 10**6).astype(np.int64)
    n_k = rng.integers(500, 1500, pairs).astype(np.int64)
    multiplicity = np.empty(n_sd, dtype=np.int64)
    multiplicity[0::2], multiplicity[1::2] = n_j, n_k
    mass = np.empty(n_sd)
    mass[0::2] = rng.uniform(0.5e-14, 2e-14, pairs)   # m_j / fragment mass ~ 1e-5
    mass[1::2] = rng.uniform(0.5e-9, 2e-9, pairs)
    fragment_mass = rng.uniform(0.8e-9, 1.2e-9, pairs)
    up = engine.upload
    state = {"multiplicity": up(multiplicity), "attributes": up(mass.reshape(1, -1).copy())}
    counters = [engine.zeros(1, np.int64) for _ in range(3)]
    healthy, overflow = engine.full(1, np.int64, 1), engine.zeros(1, np.int64)
    flag = np.zeros(n_sd, dtype=np.uint8)
    flag[0::2] = 1
    results = []
    for max_multiplicity in (2**62, 2500):  # (second round: the multiplicity limit ends loops)
        engine.call("sdm_collision_coalescence_breakup", state["multiplicity"],
                    up(np.arange(n_sd, dtype=np.int64)), n_sd, state["attributes"], 1, n_sd,
                    up(gamma), up(np.full(pairs, 0.5)), up(np.zeros(pairs)), up(np.ones(pairs)),
                    up(fragment_mass), healthy, up(np.zeros(n_sd, dtype=np.int64)), counters[0],
                    counters[1], counters[2], up(flag), int(max_multiplicity),
                    up(mass.copy()), int(handle_all_breakups), overflow)
        results.append([engine.download(a).copy() for a in
                        (state["multiplicity"], state["attributes"], *counters, overflow)])
    return results


@pytest.mark.parametrize("handle_all_breakups", [False, True])
def test_long_breakup_loops_equal_the_checker(handle_all_breakups, hip_engine, oracle_engine):
    got = _long_breakup_case(hip_engine, handle_all_breakups)
    want = _long_breakup_case(oracle_engine, handle_all_breakups)
    assert want[0][3][0] > 0  # breakups happened
    assert want[1][5][0] > 0  # ... and with the low limit, refused ones
    for round_got, round_want in zip(got, want):
        for value, ref in zip(round_got, round_want):
            np.testing.assert_array_equal(value, ref)


def test_a_stale_cell_start_is_refused_not_computed_on(hip_engine):
    """`sorted` handed over with a cell_start that belongs to another state (here: a state one
    super-droplet shorter) must end in an error, not in a run over segments that hold other cells'
    droplets (control block word 7, code 4; found the hard way: bench.py's checkpoint once forgot
    cell_start and the repetition after the first death never came back)"""
    runner = make_box(hip_engine, "kinematic2d", n_sd=2**14, grid=(4, 4))
    runner.run(2)
    pop = runner.population
    assert pop.ordered
    starts = hip_engine.download(pop.cell_start)
    starts[5:] -= 1  # what cell_start looks like after a death in cell 4
    pop.cell_start.copy_(hip_engine.upload(starts))
    pop.touch_state()
    pop.ordered = True
    with pytest.raises(RuntimeError, match="cell_start does not span"):
        runner.run(3)


def test_fused_run_takes_the_closed_form_resort_and_it_equals_the_counting_sort(hip_engine,
                                                                              oracle_engine):
    """multi-cell adaptive runs in which super-droplets die in most steps: with SDM_OPT_RESORT =
    ALWAYS_ASK the re-sort after a compaction is the closed form where it applies (the library's
    own count says it was taken), with COUNTING_SORT never - and the states are the same, and the
    checker's.  Which path a call takes does not depend on earlier calls (the back-off is reset at
    every entry: two identical calls count the same)"""
    import ctypes

    stats = (ctypes.c_int64 * 8)()
    runs = {}
    try:
        for mode in (2, 1, 0, 0):  # ALWAYS_ASK, COUNTING_SORT, AUTO, AUTO
            hip_engine.call("sdm_ctx_set_option", 0, mode)
            hip_engine.call("sdm_ctx_read_stats", stats, 1)
            runner = make_box(hip_engine, "shima", n_sd=40000, adaptive=True, dt=200.0, thin=0.02,
                              grid=(8, 5), seed=1003)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                for steps in (5, 8, 2):
                    runner.run(steps)
            hip_engine.call("sdm_ctx_read_stats", stats, 1)
            runs.setdefault(mode, []).append((runner.snapshot(), list(stats)))
    finally:
        hip_engine.call("sdm_ctx_set_option", 0, 0)
    asked, sorted_only, (auto, auto_again) = runs[2][0], runs[1][0], runs[0]
    assert asked[1][0] > 0 and asked[1][2] == 0          # closed form taken, nothing skipped
    assert asked[1][0] + asked[1][1] == asked[1][0] + asked[1][3]  # refused = counting-sorted
    assert sorted_only[1][0] == 0 and sorted_only[1][1] == 0 and sorted_only[1][3] > 0
    assert sorted_only[1][3] == asked[1][0] + asked[1][3]  # the same compactions either way
    assert auto[1] == auto_again[1]                        # no history between calls
    assert auto[1][0] > 0
    checker = make_box(oracle_engine, "shima", n_sd=40000, adaptive=True, dt=200.0, thin=0.02,
                       grid=(8, 5), seed=1003)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for steps in (5, 8, 2):
            checker.run(steps)
    want = checker.snapshot()
    for snap, _ in (asked, sorted_only, auto):
        assert_same(snap, want)


def test_ids_stay_unique_where_the_reference_duplicates_them(hip_engine):
    """several cells + global croupier + adaptive sub-stepping: the set-up warns,
    and the permutation this backend leaves behind holds every live id exactly once, however many
    working lengths were cut on the way - the behaviour chosen where the reference's is a race"""
    name = "traj_multicell_geometric_4x4_global"
    with pytest.warns(UserWarning, match="keeps every id exactly once"):
        runner, _, steps = setup_from_golden(name, hip_engine, route="fused")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        runner.run(2 * steps[-1])
    snap = runner.snapshot()
    live = snap["idx"][: int(snap["length"])]
    assert len(np.unique(live)) == len(live)
    assert (snap["multiplicity"][live] > 0).all()
    assert int(snap["stats_n_substep"].max()) > 1  # it did sub-step (working lengths were cut)
