"""Pins the oracle (oracle/sdm_oracle*.c behind include/sdm_hip.h) against trajectories recorded
from the reference itself (tests/golden/gen_golden.py), on both routes of the host layer: the
oracle's own C restatement of the driver loop ("fused") and the stage-by-stage chain.  Also pins
it at full size through the reference's digests.  CPU only."""
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
