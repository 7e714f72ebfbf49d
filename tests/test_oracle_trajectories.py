"""Pins the oracle (oracle/sdm_oracle*.c behind include/sdm_hip.h) against trajectories recorded
from the reference itself (tests/golden/gen_golden.py), on both routes of the host layer: the
oracle's own C restatement of the driver loop ("fused") and the stage-by-stage chain.  Also pins
it at full size through the reference's digests.  CPU only."""
import numpy as np
import pytest

from . import digests, displacement_cases
from .trajectory import golden_files, run_and_compare

ROUTES = ("fused", "chain")


@pytest.mark.parametrize("route", ROUTES)
@pytest.mark.parametrize("name", golden_files("traj_golovin_*.npz"))
def test_golovin_box_bit_exact(name, route, oracle_engine):
    run_and_compare(name, oracle_engine, route=route)


@pytest.mark.parametrize("route", ROUTES)
@pytest.mark.parametrize("name", golden_files("traj_geometric_*.npz"))
def test_geometric_box_bit_exact(name, route, oracle_engine):
    run_and_compare(name, oracle_engine, route=route)


@pytest.mark.parametrize("route", ROUTES)
@pytest.mark.parametrize("name", golden_files("traj_multicell_*.npz"))
def test_multicell_bit_exact(name, route, oracle_engine):
    run_and_compare(name, oracle_engine, route=route)


@pytest.mark.parametrize("route", ROUTES)
@pytest.mark.parametrize("name", golden_files("traj_breakup_*.npz"))
def test_breakup(name, route, oracle_engine):
    # integer state (indices, multiplicities, counters) bit-exact.  Fragment volumes go through
    # log/exp/sinh..., which numpy (SIMD loops, used by the reference run that made the goldens)
    # and glibc (the C oracle) round differently in the last bit, so attributes get 1e-12
    run_and_compare(name, oracle_engine, route=route, float_rtol=1e-12)


@pytest.mark.parametrize("route", ROUTES)
@pytest.mark.parametrize("name", displacement_cases.CASES)
def test_displacement_goldens(name, route, oracle_engine):
    displacement_cases.run_case(name, oracle_engine, route=route)


@pytest.mark.parametrize("route", ROUTES)
@pytest.mark.parametrize("name", golden_files("traj_kernel_*.npz"))
def test_other_kernels_bit_exact(name, route, oracle_engine):
    run_and_compare(name, oracle_engine, route=route)


@pytest.mark.parametrize("name", digests.available())
def test_full_size_digests_of_the_reference(name, oracle_engine):
    """the oracle equals the reference itself at BASELINE.json's sizes: 2^14 .. 2^20 boxes
    (Golovin, Berry breakup, Straub on the rain spectrum), 32 x 32 cells with 64 and 4096
    super-droplets each"""
    if digests.n_sd_of(name) > 2**22:
        pytest.skip("beyond 2^22: GPU box only")
    digests.check(name, oracle_engine)


def test_fused_route_reports_dt_min_like_the_stage_by_stage_route(oracle_engine):
    """collision.py:276-277: "adaptive time-step reached dt_min" is raised when the smallest
    `stats_dt_min` equals the lower end of dt_coal_range - never with the reference's NaN
    initial values (NaN-sticky minimum), always once a user has reset the statistics and a cell
    needs a sub-step at the limit.  Both routes must agree (the fused one learns it through the
    event bit of the control block)."""
    import warnings  # pylint: disable=import-outside-toplevel

    from pysdm_amd.cases import make_box  # pylint: disable=import-outside-toplevel

    outcomes = {}
    for route in ("fused", "chain"):
        for reset in (False, True):
            runner = make_box(oracle_engine, "shima", n_sd=2**10, adaptive=True, route=route,
                              dt=200.0, dt_range=(100.0, 200.0))
            if reset:
                oracle_engine.fill(runner.stats_dt_min, 200.0)
            with warnings.catch_warnings(record=True) as caught:
                warnings.simplefilter("always")
                runner.run(2)
            outcomes[route, reset] = any("dt_min" in str(w.message) for w in caught)
            if reset:
                assert oracle_engine.download(runner.stats_dt_min)[0] == 100.0
    assert outcomes == {("fused", False): False, ("chain", False): False,
                        ("fused", True): True, ("chain", True): True}


@pytest.mark.parametrize("name", ["traj_golovin_n1024_s44_a1", "traj_multicell_geometric_4x4"])
def test_fused_plugin_over_a_duck_particulator(name, oracle_backend_class):
    """the CPU twin of tests/test_hip_parity.py::test_fused_plugin_on_hip_storages (the same
    driver; under the real PySDM front-end: tests/test_reference_plugin.py)"""
    from . import pysdm_ducks  # pylint: disable=import-outside-toplevel

    pysdm_ducks.fused_plugin_run(name, oracle_backend_class)


def test_host_view_is_refreshed_after_steps_without_read_back(oracle_engine):
    """steps without read-back leave live / working / ordered behind the device's control block;
    a later host-side event (here: touch_state, as a displacement or an attribute edit does) makes
    the next call upload the host's view - which must have been brought up to date first, or dead
    slots would be re-admitted (collisions.py:177)"""
    from pysdm_amd.cases import make_box  # pylint: disable=import-outside-toplevel

    def box(read_back):
        return make_box(oracle_engine, "shima", n_sd=2**11, adaptive=True, dt=200.0, thin=0.02,
                        grid=(4, 4), read_back=read_back)

    lazy, eager = box(False), box(True)
    for runner in (lazy, eager):
        runner.run(4)
        runner.population.touch_state()
        runner.run(2)
        runner.population.compact()
        runner.run(1)
    lazy.sync()
    assert eager.population.live < 2**11
    a, b = lazy.snapshot(), eager.snapshot()
    for key, value in b.items():
        np.testing.assert_array_equal(a[key], value, err_msg=key)
