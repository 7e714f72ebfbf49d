"""Pins the oracle (oracle/sdm_oracle.c behind the package's host logic) against trajectories
recorded from the reference itself (tests/golden/gen_golden.py).  CPU only."""
import pytest

from . import displacement_cases
from .trajectory import golden_files, run_and_compare


@pytest.mark.parametrize("name", golden_files("traj_golovin_*.npz"))
def test_golovin_box_bit_exact(name, oracle_backend_class):
    run_and_compare(name, oracle_backend_class)


@pytest.mark.parametrize("name", golden_files("traj_geometric_*.npz"))
def test_geometric_box_bit_exact(name, oracle_backend_class):
    run_and_compare(name, oracle_backend_class)


@pytest.mark.parametrize("name", golden_files("traj_multicell_*.npz"))
def test_multicell_bit_exact(name, oracle_backend_class):
    run_and_compare(name, oracle_backend_class)


@pytest.mark.parametrize("name", golden_files("traj_breakup_*.npz"))
def test_breakup(name, oracle_backend_class):
    # integer state (indices, multiplicities, counters) bit-exact.  Fragment volumes go through
    # log/exp/sinh..., which numpy (SIMD loops, used by the reference run that made the goldens)
    # and glibc (the C oracle) round differently in the last bit, so attributes get 1e-12
    run_and_compare(name, oracle_backend_class, float_rtol=1e-12)


@pytest.mark.parametrize("name", displacement_cases.CASES)
def test_displacement_goldens(name, oracle_backend_class):
    displacement_cases.run_case(name, oracle_backend_class)


@pytest.mark.parametrize("name", golden_files("traj_kernel_*.npz"))
def test_other_kernels_bit_exact(name, oracle_backend_class):
    run_and_compare(name, oracle_backend_class)
