"""the backend object of this package driven by the UNMODIFIED reference
front-end (PySDM's own Builder / Particulator / ParticleAttributes / Index / PairwiseStorage /
Collision / Displacement) (this package has no front-end of its own: its host layer is the Population / runner API).

Only meaningful where the reference is importable, i.e. in the build container (it does not
travel to the GPU box, and there is no GPU here): the backend plugged in is therefore the CPU
oracle - the same PySDM-shaped class `HIP` is (pysdm_amd/backends/pysdm_shaped.py), bound to the
oracle's implementation of include/sdm_hip.h - and the expected values are the committed goldens.  Skipped wherever PySDM cannot be imported."""
import os
import sys
import warnings

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = "/root/reference"
pytestmark = pytest.mark.skipif(
    not os.path.isdir(os.path.join(REFERENCE, "PySDM")), reason="reference tree not present")


@pytest.fixture(scope="module", name="ref")
def reference_modules():
    """imports PySDM in its pure-Python mode with the import-only stand-ins of tests/golden"""
    os.environ.setdefault("CI", "1")
    sys.dont_write_bytecode = True
    added = [os.path.join(HERE, "golden", "standins"), REFERENCE]
    sys.path[:0] = added
    try:
        import PySDM  # pylint: disable=import-outside-toplevel,import-error
        from PySDM import dynamics  # pylint: disable=import-outside-toplevel,import-error
        from PySDM.dynamics.collisions import (  # pylint: disable=import-outside-toplevel,import-error
            breakup_efficiencies, breakup_fragmentations, coalescence_efficiencies,
            collision_kernels,
        )
        from PySDM.environments import Box  # pylint: disable=import-outside-toplevel,import-error
        from PySDM.impl.mesh import Mesh  # pylint: disable=import-outside-toplevel,import-error
    except Exception as error:  # pylint: disable=broad-except
        pytest.skip(f"PySDM not importable here: {error}")
    finally:
        for path in added:
            sys.path.remove(path)
    return {
        "PySDM": PySDM, "dynamics": dynamics, "kernels": collision_kernels,
        "ec": coalescence_efficiencies, "eb": breakup_efficiencies,
        "frag": breakup_fragmentations, "Box": Box, "Mesh": Mesh,
    }


@pytest.fixture(scope="module", name="plugged")
def plugged_backend(ref, oracle_backend_class):  # pylint: disable=unused-argument
    from pysdm_amd.pysdm_plugin import as_pysdm_backend  # pylint: disable=import-outside-toplevel

    return as_pysdm_backend(oracle_backend_class)


def _gold(name):
    return np.load(os.path.join(HERE, "golden", name + ".npz"))


def _compare(particulator, dynamic, gold, step, float_rtol=0.0):
    attrs = particulator.attributes
    idx = attrs._ParticleAttributes__idx  # pylint: disable=protected-access
    length = len(idx)
    assert length == int(gold[f"step{step}/length"])
    np.testing.assert_array_equal(idx.to_ndarray()[:length], gold[f"step{step}/idx"][:length])
    np.testing.assert_array_equal(attrs["multiplicity"].to_ndarray(raw=True),
                                  gold[f"step{step}/multiplicity"])
    mass = attrs.get_extensive_attribute_storage().to_ndarray(raw=True)
    if float_rtol:
        np.testing.assert_allclose(mass, gold[f"step{step}/attributes"], rtol=float_rtol)
    else:
        np.testing.assert_array_equal(mass, gold[f"step{step}/attributes"])
    for key in ("collision_rate", "collision_rate_deficit", "coalescence_rate",
                "stats_n_substep"):
        np.testing.assert_array_equal(getattr(dynamic, key).to_ndarray(), gold[f"step{step}/{key}"])


def _run(particulator, gold, float_rtol=0.0):
    dynamic = particulator.dynamics["Collision"]
    steps = sorted({int(k.split("/")[0][4:]) for k in gold.files if k.startswith("step")})
    for step in steps:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            particulator.run(step - particulator.n_steps)
        _compare(particulator, dynamic, gold, step, float_rtol)


@pytest.mark.parametrize("name", ["traj_golovin_n1024_s44_a1", "traj_golovin_n256_s256_a0",
                                  "traj_golovin_global", "traj_golovin_optrand",
                                  "traj_golovin_deaths_adaptive"])
def test_reference_frontend_single_cell(name, ref, plugged):
    gold = _gold(name)
    cfg = gold["cfg"]
    kwargs = {}
    if "global" in name:
        kwargs["croupier"] = "global"
    if "optrand" in name:
        kwargs["optimized_random"] = True
    builder = ref["PySDM"].Builder(
        n_sd=int(cfg[0]), backend=plugged(ref["PySDM"].Formulae(seed=int(cfg[1]))),
        environment=ref["Box"](dt=cfg[3], dv=cfg[4]))
    builder.add_dynamic(ref["dynamics"].Coalescence(
        collision_kernel=ref["kernels"].Golovin(b=cfg[5]), adaptive=bool(cfg[2]), **kwargs))
    particulator = builder.build({"volume": gold["init/volume"],
                                  "multiplicity": gold["init/multiplicity"]})
    _run(particulator, gold)


def test_reference_frontend_multi_cell(ref, plugged):
    gold = _gold("traj_multicell_geometric_4x4")
    cfg = gold["cfg"]
    grid = tuple(int(g) for g in gold["grid"])
    env = ref["Box"](dt=cfg[3], dv=cfg[4])
    env.mesh = ref["Mesh"](grid, size=tuple(float(g) for g in grid))
    env.mesh.dv = cfg[4]
    builder = ref["PySDM"].Builder(
        n_sd=int(cfg[0]), environment=env,
        backend=plugged(ref["PySDM"].Formulae(seed=44, terminal_velocity="GunnKinzer1949")))
    builder.add_dynamic(ref["dynamics"].Coalescence(
        collision_kernel=ref["kernels"].Geometric(collection_efficiency=1),
        adaptive=bool(cfg[2]), optimized_random=bool(cfg[6])))
    particulator = builder.build({"volume": gold["init/volume"],
                                  "multiplicity": gold["init/multiplicity"],
                                  "cell id": gold["init/cell_id"]})
    _run(particulator, gold)


def test_reference_frontend_breakup(ref, plugged):
    gold = _gold("traj_breakup_straub_rain_hab1")
    cfg = gold["cfg"]
    formulae = ref["PySDM"].Formulae(
        seed=int(cfg[1]), fragmentation_function="Straub2010Nf", handle_all_breakups=bool(cfg[5]),
        terminal_velocity="GunnKinzer1949")
    builder = ref["PySDM"].Builder(n_sd=int(cfg[0]), backend=plugged(formulae),
                                   environment=ref["Box"](dt=cfg[3], dv=cfg[4]))
    builder.add_dynamic(ref["dynamics"].Collision(
        collision_kernel=ref["kernels"].Geometric(),
        coalescence_efficiency=ref["ec"].Straub2010Ec(),
        breakup_efficiency=ref["eb"].ConstEb(1.0),
        fragmentation_function=ref["frag"].Straub2010Nf(vmin=(0.01e-3) ** 3 * np.pi / 6,
                                                        nfmax=10000),
        adaptive=True, warn_overflows=False))
    particulator = builder.build({"volume": gold["init/volume"],
                                  "multiplicity": gold["init/multiplicity"]})
    _run(particulator, gold, float_rtol=1e-12)


@pytest.mark.parametrize("name", ["traj_golovin_n1024_s44_a1", "traj_golovin_deaths_adaptive",
                                  "traj_golovin_optrand", "traj_golovin_global"])
def test_fused_step_under_reference_builder(name, ref, plugged):
    """the route to the fused step in a PySDM installation: PySDM's Builder, Particulator and
    ParticleAttributes, PySDM's own Coalescence object wrapped by `fuse` - each time step is one
    `sdm_collision_step` on PySDM's arrays (here of the oracle library: no GPU in this container)"""
    from pysdm_amd.pysdm_plugin import fuse  # pylint: disable=import-outside-toplevel

    gold = _gold(name)
    cfg = gold["cfg"]
    kwargs = {}
    if "global" in name:
        kwargs["croupier"] = "global"
    if "optrand" in name:
        kwargs["optimized_random"] = True
    builder = ref["PySDM"].Builder(
        n_sd=int(cfg[0]), backend=plugged(ref["PySDM"].Formulae(seed=int(cfg[1]))),
        environment=ref["Box"](dt=cfg[3], dv=cfg[4]))
    builder.add_dynamic(fuse(ref["dynamics"].Coalescence(
        collision_kernel=ref["kernels"].Golovin(b=cfg[5]), adaptive=bool(cfg[2]), **kwargs)))
    particulator = builder.build({"volume": gold["init/volume"],
                                  "multiplicity": gold["init/multiplicity"]})
    _run(particulator, gold)


def test_fused_breakup_and_multicell_under_reference_builder(ref, plugged):
    from pysdm_amd.pysdm_plugin import fuse  # pylint: disable=import-outside-toplevel

    gold = _gold("traj_breakup_straub_rain_hab1")
    cfg = gold["cfg"]
    formulae = ref["PySDM"].Formulae(
        seed=int(cfg[1]), fragmentation_function="Straub2010Nf", handle_all_breakups=bool(cfg[5]),
        terminal_velocity="GunnKinzer1949")
    builder = ref["PySDM"].Builder(n_sd=int(cfg[0]), backend=plugged(formulae),
                                   environment=ref["Box"](dt=cfg[3], dv=cfg[4]))
    builder.add_dynamic(fuse(ref["dynamics"].Collision(
        collision_kernel=ref["kernels"].Geometric(),
        coalescence_efficiency=ref["ec"].Straub2010Ec(),
        breakup_efficiency=ref["eb"].ConstEb(1.0),
        fragmentation_function=ref["frag"].Straub2010Nf(vmin=(0.01e-3) ** 3 * np.pi / 6,
                                                        nfmax=10000),
        adaptive=True, warn_overflows=False)))
    particulator = builder.build({"volume": gold["init/volume"],
                                  "multiplicity": gold["init/multiplicity"]})
    _run(particulator, gold, float_rtol=1e-12)

    gold = _gold("traj_multicell_geometric_4x4")
    cfg = gold["cfg"]
    grid = tuple(int(g) for g in gold["grid"])
    env = ref["Box"](dt=cfg[3], dv=cfg[4])
    env.mesh = ref["Mesh"](grid, size=tuple(float(g) for g in grid))
    env.mesh.dv = cfg[4]
    builder = ref["PySDM"].Builder(
        n_sd=int(cfg[0]), environment=env,
        backend=plugged(ref["PySDM"].Formulae(seed=44, terminal_velocity="GunnKinzer1949")))
    builder.add_dynamic(fuse(ref["dynamics"].Coalescence(
        collision_kernel=ref["kernels"].Geometric(collection_efficiency=1),
        adaptive=bool(cfg[2]), optimized_random=bool(cfg[6]))))
    particulator = builder.build({"volume": gold["init/volume"],
                                  "multiplicity": gold["init/multiplicity"],
                                  "cell id": gold["init/cell_id"]})
    _run(particulator, gold)


def test_reference_displacement_dynamic(ref, plugged):
    gold = _gold("traj_disp2d_implicit_sed")
    n_sd, dt = int(gold["cfg"][0]), float(gold["cfg"][1])
    grid = tuple(int(g) for g in gold["grid"])
    env = ref["Box"](dt=dt, dv=None)
    env.mesh = ref["Mesh"](grid, tuple(float(v) for v in gold["size"]))
    formulae = ref["PySDM"].Formulae(seed=44, particle_advection="ImplicitInSpace",
                                     terminal_velocity="GunnKinzer1949")
    builder = ref["PySDM"].Builder(n_sd=n_sd, backend=plugged(formulae), environment=env)
    builder.add_dynamic(ref["dynamics"].Displacement(enable_sedimentation=True, adaptive=True))
    cell_id, cell_origin, position = env.mesh.cellular_attributes(gold["init/positions"])
    particulator = builder.build({
        "volume": gold["init/volume"], "multiplicity": gold["init/multiplicity"],
        "cell id": cell_id, "cell origin": cell_origin, "position in cell": position})
    disp = particulator.dynamics["Displacement"]
    disp.upload_courant_field(tuple(gold[f"courant/{d}"] for d in range(len(grid))))
    for step in range(1, int(gold["cfg"][-1]) + 1):
        particulator.run(1)
        attrs = particulator.attributes
        attrs.sanitize()
        length = attrs.super_droplet_count
        assert length == int(gold[f"step{step}/length"])
        live = attrs._ParticleAttributes__idx.to_ndarray()[:length]  # pylint: disable=protected-access
        np.testing.assert_array_equal(live, gold[f"step{step}/idx"][:length])
        np.testing.assert_array_equal(attrs["cell origin"].to_ndarray(raw=True)[:, live],
                                      gold[f"step{step}/cell_origin"][:, live])
        np.testing.assert_allclose(attrs["position in cell"].to_ndarray(raw=True)[:, live],
                                   gold[f"step{step}/position"][:, live], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(disp.precipitation_mass_in_last_step,
                                   float(gold[f"step{step}/precipitation"]), rtol=1e-12)


def test_fused_dynamic_honours_enable(ref, plugged):
    """collision.py:175 gates the step on `enable`; PySDM's SpinUp observer
    (examples/PySDM_examples/Arabas_et_al_2015/spin_up.py) switches collisions off and on with
    setattr(particulator.dynamics["Collision"], "enable", ...) - that object is the fused wrapper"""
    from pysdm_amd.pysdm_plugin import fuse  # pylint: disable=import-outside-toplevel

    gold = _gold("traj_golovin_n1024_s44_a1")
    cfg = gold["cfg"]
    builder = ref["PySDM"].Builder(
        n_sd=int(cfg[0]), backend=plugged(ref["PySDM"].Formulae(seed=int(cfg[1]))),
        environment=ref["Box"](dt=cfg[3], dv=cfg[4]))
    builder.add_dynamic(fuse(ref["dynamics"].Coalescence(
        collision_kernel=ref["kernels"].Golovin(b=cfg[5]), adaptive=bool(cfg[2]))))
    particulator = builder.build({"volume": gold["init/volume"],
                                  "multiplicity": gold["init/multiplicity"]})
    dynamic = particulator.dynamics["Collision"]
    setattr(dynamic, "enable", False)
    assert dynamic.enable is False and dynamic.inner.enable is False
    before = particulator.attributes["multiplicity"].to_ndarray(raw=True).copy()
    particulator.run(3)  # spin-up: nothing may collide, no random number may be drawn
    np.testing.assert_array_equal(particulator.attributes["multiplicity"].to_ndarray(raw=True),
                                  before)
    setattr(dynamic, "enable", True)
    # ... and the run that follows is the golden from its first step
    steps = sorted({int(k.split("/")[0][4:]) for k in gold.files if k.startswith("step")})
    for step in steps:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            particulator.run(3 + step - particulator.n_steps)
        _compare(particulator, dynamic, gold, step)
