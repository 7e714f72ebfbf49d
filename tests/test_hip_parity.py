"""GPU parity tests proper: the HIP path (through the C ABI) against the goldens recorded from the
reference and against the oracle on the same seeded inputs.  Integer state bit-exact; floating
point bit-exact on the coalescence-only paths, 1e-12 relative where device transcendentals
(OCML pow/log/exp...) feed attributes (breakup)."""
import numpy as np
import pytest

from . import displacement_cases
from . import known_answers as ka
from . import micro_cases as mc
from .trajectory import golden_files, run_and_compare, setup_from_golden, snapshot

pytestmark = pytest.mark.gpu

EXACT_ON_GPU = {"volume", "golovin", "frag_always_n_4"}


@pytest.fixture(scope="module", name="kit")
def kit_fixture(hip_backend_class):
    return mc.Kit(hip_backend_class, fragmentation_function="Straub2010Nf")


@pytest.mark.parametrize("check", [mc.check_pcg64, mc.check_shuffle,
                                   mc.check_shuffle_known_answers, mc.check_counting_sort,
                                   mc.check_sort_by_key_and_adaptive_end, mc.check_remove_zero,
                                   mc.check_pair_chain, mc.check_moments, mc.check_moments_goldens,
                                   mc.check_storage_ops])
def test_method_goldens(check, kit):
    check(kit)


def test_physics_goldens(kit):
    mc.check_physics(kit, exact=EXACT_ON_GPU, rtol=1e-13)


COALESCENCE = (golden_files("traj_golovin_*.npz") + golden_files("traj_geometric_*.npz")
               + golden_files("traj_multicell_*.npz") + golden_files("traj_kernel_*.npz"))


@pytest.mark.parametrize("fused", [False, None], ids=["methods", "fused"])
@pytest.mark.parametrize("name", COALESCENCE)
def test_coalescence_trajectories_bit_exact(name, fused, hip_backend_class):
    run_and_compare(name, hip_backend_class, fused=fused)


@pytest.mark.parametrize("fused", [False, None], ids=["methods", "fused"])
@pytest.mark.parametrize("name", golden_files("traj_breakup_*.npz"))
def test_breakup_trajectories(name, fused, hip_backend_class):
    run_and_compare(name, hip_backend_class, fused=fused, float_rtol=1e-12)


@pytest.mark.parametrize("name", ["traj_golovin_n4096_s44_a1", "traj_multicell_geometric_4x4"])
def test_fused_equals_oracle_beyond_goldens(name, hip_backend_class, oracle_backend_class):
    """same seeded inputs, more steps than the goldens hold"""
    snaps = []
    for backend_class in (hip_backend_class, oracle_backend_class):
        particulator, dynamic, _, _ = setup_from_golden(name, backend_class)
        particulator.run(120)
        snaps.append(snapshot(particulator, dynamic))
    length = int(snaps[0]["length"])
    for key, value in snaps[0].items():
        ref = snaps[1][key]
        if key == "idx":  # beyond `length`: dead storage (see trajectory.compare)
            value, ref = value[:length], ref[:length]
        np.testing.assert_array_equal(value, ref, err_msg=key)


@pytest.mark.parametrize("check", ka.ALL_CHECKS)
def test_reference_known_answers(check, kit):
    check(kit)


@pytest.mark.parametrize("fused", [False, None], ids=["methods", "fused"])
@pytest.mark.parametrize("name", displacement_cases.CASES)
def test_displacement_goldens(name, fused, hip_backend_class):
    displacement_cases.run_case(name, hip_backend_class, fused=fused)


def test_c_abi_example_equals_python_route(tmp_path, hip_backend_class):
    """examples/shima_box_c_abi.cpp: the Shima box driven through include/sdm_hip.h from a plain
    C++ program (no Python, no torch) gives bit for bit what the Python host code gives"""
    import os  # pylint: disable=import-outside-toplevel
    import struct  # pylint: disable=import-outside-toplevel
    import subprocess  # pylint: disable=import-outside-toplevel

    from pysdm_amd.backends.hip import pcg64_state_inc  # pylint: disable=import-outside-toplevel
    from pysdm_amd.examples import make_box  # pylint: disable=import-outside-toplevel

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    binary = os.path.join(root, "examples", "shima_box_c_abi")
    library = os.path.join(root, "pysdm_amd", "libsdm_hip.so")
    if not os.path.exists(binary) or os.path.getmtime(binary) < os.path.getmtime(library):
        subprocess.check_call(["bash", os.path.join(root, "examples", "build.sh")])
    n_sd, n_steps, seed = 2**14, 25, 44
    particulator, dynamic = make_box(hip_backend_class, "shima", n_sd=n_sd, seed=seed)
    attrs = particulator.attributes
    multiplicity = attrs["multiplicity"].to_ndarray(raw=True)
    mass = attrs["water mass"].to_ndarray(raw=True)
    source, result = tmp_path / "box.in", tmp_path / "box.out"
    with open(source, "wb") as handle:
        handle.write(struct.pack("<qq", n_sd, n_steps))
        handle.write(struct.pack("<ddd", particulator.dt, particulator.mesh.dv, 1.5e3))
        handle.write(struct.pack("<4Q", *pcg64_state_inc(seed)))
        handle.write(multiplicity.astype("<i8").tobytes())
        handle.write(mass.astype("<f8").tobytes())
    subprocess.check_call([binary, str(source), str(result)])
    raw = np.fromfile(result, dtype=np.uint8)
    n_live = int(raw[:8].view("<i8")[0])
    idx = raw[8:8 + 8 * n_sd].view("<i8")
    mult_c = raw[8 + 8 * n_sd:8 + 16 * n_sd].view("<i8")
    mass_c = raw[8 + 16 * n_sd:8 + 24 * n_sd].view("<f8")
    particulator.run(n_steps)
    snap = snapshot(particulator, dynamic)
    assert n_live == int(snap["length"])
    np.testing.assert_array_equal(idx[:n_live], snap["idx"][:n_live])
    np.testing.assert_array_equal(mult_c, snap["multiplicity"])
    np.testing.assert_array_equal(mass_c, snap["attributes"][0])


def test_linear_kernel_routes_agree(hip_backend_class, oracle_backend_class):
    """the Linear kernel cannot run in the reference (a stub there): fused route, method route
    and the oracle agree bit for bit on a Shima-type box"""
    from pysdm_amd.dynamics.collisions import Coalescence, Linear  # pylint: disable=import-outside-toplevel
    from pysdm_amd.examples import make_box  # pylint: disable=import-outside-toplevel

    snaps = []
    for backend_class, fused in ((hip_backend_class, None), (hip_backend_class, False),
                                 (oracle_backend_class, None)):
        particulator, _ = make_box(backend_class, "shima", n_sd=2**12, adaptive=True)
        particulator.dynamics["Collision"] = Coalescence(
            collision_kernel=Linear(a=2e-9, b=1.5e3), adaptive=True, fused=fused
        ).instantiate(builder=type("B", (), {"particulator": particulator,
                                              "formulae": particulator.formulae,
                                              "request_attribute": staticmethod(lambda *_: None)}))
        dynamic = particulator.dynamics["Collision"]
        particulator.run(30)
        snaps.append(snapshot(particulator, dynamic))
    length = int(snaps[0]["length"])
    for other in snaps[1:]:
        for key, value in snaps[0].items():
            ref = other[key]
            if key == "idx":
                value, ref = value[:length], ref[:length]
            np.testing.assert_array_equal(value, ref, err_msg=key)


def test_overflow_warnings_step_by_step(hip_backend_class, oracle_backend_class):
    """breakups refused for multiplicity overflow (collisions_methods.py:113-118) raise the
    reference's "overflow" warning in the very step they happen - on the fused route the count
    travels through the control block - and leave the same state behind"""
    import warnings  # pylint: disable=import-outside-toplevel

    from pysdm_amd import Builder, Formulae  # pylint: disable=import-outside-toplevel
    from pysdm_amd.dynamics.collisions import (  # pylint: disable=import-outside-toplevel
        AlwaysN, Collision, ConstEb, ConstEc, Golovin)
    from pysdm_amd.environments import Box  # pylint: disable=import-outside-toplevel

    rng = np.random.default_rng(3)
    n_sd = 2048
    volume = rng.exponential(1.2e-13, n_sd)
    multiplicity = rng.integers(10**6, 10**7, n_sd)
    outcomes = []
    for backend_class, fused in ((hip_backend_class, None), (hip_backend_class, False),
                                 (oracle_backend_class, None)):
        builder = Builder(n_sd=n_sd, backend=backend_class(Formulae(seed=5,
                          fragmentation_function="AlwaysN")), environment=Box(dt=1.0, dv=2e3))
        builder.add_dynamic(Collision(
            collision_kernel=Golovin(b=1.5e3), coalescence_efficiency=ConstEc(Ec=0.2),
            breakup_efficiency=ConstEb(1.0), fragmentation_function=AlwaysN(n=3e6),
            adaptive=True, warn_overflows=True, fused=fused))
        particulator = builder.build({"volume": volume.copy(),
                                      "multiplicity": multiplicity.copy()})
        dynamic = particulator.dynamics["Collision"]
        warned = []
        for _ in range(12):
            with warnings.catch_warnings(record=True) as caught:
                warnings.simplefilter("always")
                particulator.run(1)
            warned.append(any("overflow" in str(w.message) for w in caught))
        outcomes.append((warned, snapshot(particulator, dynamic)))
    assert any(outcomes[2][0]) and not all(outcomes[2][0]), outcomes[2][0]
    length = int(outcomes[0][1]["length"])
    for warned, snap in outcomes[:2]:
        assert warned == outcomes[2][0]
        for key, value in snap.items():
            ref = outcomes[2][1][key]
            if key == "idx":
                value, ref = value[:length], ref[:length]
            if value.dtype.kind == "f":
                np.testing.assert_allclose(value, ref, rtol=1e-12, atol=0, err_msg=key)
            else:
                np.testing.assert_array_equal(value, ref, err_msg=key)


@pytest.mark.parametrize("base,adaptive,grid,n_sd", [
    ("straub", True, (4, 4), 2**13), ("berry_breakup", True, (4, 4), 2**13),
    ("straub", False, (4, 4), 2**13),
    ("straub", True, (2, 2), 4 * 5850),  # cells above k_cell_step2's cap: k_cell_step<.., true>
])
def test_multicell_breakup_equals_oracle(base, adaptive, grid, n_sd, hip_backend_class,
                                         oracle_backend_class):
    """breakup on a grid (the reference has no such golden): the per-cell kernels' listing of
    colliding pairs + the dense resolution, sub-steps launched ahead of the read-back, per-cell
    counters - against the oracle's method-by-method run"""
    from pysdm_amd.examples import CONFIGS, make_box  # pylint: disable=import-outside-toplevel

    name = "_grid_" + base
    CONFIGS[name] = dict(CONFIGS[base], grid=grid)
    try:
        snaps = []
        for backend_class in (hip_backend_class, oracle_backend_class):
            particulator, dynamic = make_box(backend_class, name, n_sd=n_sd, adaptive=adaptive,
                                             dt=5.0)
            for steps in ((1, 12, 3) if n_sd == 2**13 else (1, 3, 1)):
                import warnings  # pylint: disable=import-outside-toplevel
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    particulator.run(steps)
            snaps.append(snapshot(particulator, dynamic))
    finally:
        del CONFIGS[name]
    length = int(snaps[0]["length"])
    assert snaps[1]["breakup_rate"].sum() > 0
    for key, value in snaps[0].items():
        ref = snaps[1][key]
        if key == "idx":
            value, ref = value[:length], ref[:length]
        if value.dtype.kind == "f":
            np.testing.assert_allclose(value, ref, rtol=1e-12, atol=0, err_msg=key)
        else:
            np.testing.assert_array_equal(value, ref, err_msg=key)


@pytest.mark.parametrize("n_sd,which", [(4 * 5850, "one workgroup per CU"),
                                        (4 * 4000, "two workgroups per CU")])
def test_both_per_cell_kernels_equal_oracle(n_sd, which, hip_backend_class, oracle_backend_class):
    """2 x 2 cells of ~5850 super-droplets take k_cell_step (cells above k_cell_step2's 5632),
    cells of ~4000 take k_cell_step2: adaptive geometric coalescence with optimized_random,
    state and counters equal the oracle's"""
    from pysdm_amd.examples import CONFIGS, make_box  # pylint: disable=import-outside-toplevel

    CONFIGS["_grid_2x2"] = dict(CONFIGS["kinematic2d"], grid=(2, 2))
    try:
        snaps = []
        for backend_class in (hip_backend_class, oracle_backend_class):
            particulator, dynamic = make_box(backend_class, "_grid_2x2", n_sd=n_sd)
            for steps in (1, 8, 2):
                particulator.run(steps)
            snaps.append(snapshot(particulator, dynamic))
    finally:
        del CONFIGS["_grid_2x2"]
    sizes = np.diff(snaps[1]["cell_start"])
    assert (sizes.max() > 5632) == (which == "one workgroup per CU") and sizes.max() <= 6144
    length = int(snaps[0]["length"])
    assert snaps[1]["collision_rate"].sum() > 0
    for key, value in snaps[0].items():
        ref = snaps[1][key]
        if key == "idx":
            value, ref = value[:length], ref[:length]
        np.testing.assert_array_equal(value, ref, err_msg=key)


def test_degenerate_sizes_through_the_abi(kit):
    """empty and tiny inputs: zero-length arrays are accepted by every entry point that takes a
    length, two super-droplets form one pair, three leave one alone, a null context is refused"""
    import ctypes  # pylint: disable=import-outside-toplevel

    from pysdm_amd import _lib  # pylint: disable=import-outside-toplevel
    from pysdm_amd.backends.hip import _Context  # pylint: disable=import-outside-toplevel

    lib, ctx = _lib.load(), _Context.get()
    null = ctypes.c_void_p(0)
    zero = ctypes.c_int64(0)
    for name, args in (
        ("sdm_identity_index", (null, zero)),
        ("sdm_elementwise_f64", (ctypes.c_int(0), null, null, null, ctypes.c_double(0), zero)),
        ("sdm_elementwise_i64", (ctypes.c_int(0), null, null, null, zero, zero)),
        ("sdm_volume_of_water_mass", (null, null, zero, ctypes.c_double(1000.0))),
        ("sdm_floor_to_i64", (null, null, zero)),
        ("sdm_subtract_i64", (null, null, zero)),
        ("sdm_ll82_coalescence_check", (null, null, zero)),
    ):
        assert getattr(lib, name)(ctx.handle, *args) == 0, name
    assert lib.sdm_identity_index(None, null, zero) == -1  # SDM_E_ARG, message available
    assert b"bad argument" in lib.sdm_last_error()
    for n_sd in (2, 3):
        idx = kit.Index.identity_index(n_sd)
        u01 = kit.Storage.from_ndarray(np.full(n_sd, 0.75))
        idx.shuffle(u01)
        assert sorted(idx.to_ndarray().tolist()) == list(range(n_sd))
        flag = kit.PairIndicator(n_sd)
        cell_start = kit.Storage.from_ndarray(np.asarray([0, n_sd]))
        cell_id = kit.IndexedStorage.from_ndarray(idx, np.zeros(n_sd, dtype=np.int64))
        flag.update(cell_start, kit.Index.identity_index(1), cell_id)
        assert flag.indicator.to_ndarray().sum() == 1
