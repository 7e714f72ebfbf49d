"""Pins each oracle function against the per-method goldens recorded from the reference, and
against the known-answer tables of the reference's own unit tests.  CPU only."""
import numpy as np
import pytest

from . import known_answers as ka
from . import micro_cases as mc

# bit-identical entries; everything that goes through pow / exp / log is compared at 1e-14: the
# reference run that made the goldens used numpy's SIMD loops for them, the C oracle uses glibc,
# and the two round differently in the last bit for a few per cent of the arguments
EXACT_ON_CPU = {"volume", "golovin", "frag_always_n_4"}


@pytest.fixture(scope="module", name="kit")
def kit_fixture(oracle_backend_class):
    return mc.Kit(oracle_backend_class, fragmentation_function="Straub2010Nf")


@pytest.mark.parametrize("check", [mc.check_pcg64, mc.check_shuffle,
                                   mc.check_shuffle_known_answers, mc.check_counting_sort,
                                   mc.check_sort_by_key_and_adaptive_end, mc.check_remove_zero, mc.check_sanitize_sorted,
                                   mc.check_pair_chain, mc.check_moments, mc.check_moments_goldens,
                                   mc.check_storage_ops])
def test_method_goldens(check, kit):
    check(kit)


def test_physics_goldens(kit):
    mc.check_physics(kit, exact=EXACT_ON_CPU, rtol=1e-14)


@pytest.mark.parametrize("check", ka.ALL_CHECKS)
def test_reference_known_answers(check, kit):
    check(kit)


def test_lowlist82_parameter_known_answers():
    """tests/unit_tests/physics/test_fragmentation_functions.py:76-173 re-typed: the (H, mu, sigma)
    triples of the seven Low & List 1982 modes for the drop pair of their Table"""
    import ctypes  # pylint: disable=import-outside-toplevel

    from oracle.engine import OracleEngine  # pylint: disable=import-outside-toplevel

    cm = 0.01
    fun = OracleEngine.get().library.cdll.oracle_ll82_params
    fun.restype = None
    cases = (
        (0, (0.36 * cm, 0.3744 * cm, 0, 0), (105.78851401149461, 0.36, 0.003771383856549656)),
        (1, (0.18 * cm, 0, 0, 0), (31.081892267202157, 0.18, 0.01283519925273017)),
        (2, (0.0715 * cm, 0.18 * cm, 0, 0),
         (11.078017412424996, -3.4579794266811095, 0.21024917628814235)),
        (3, (0.36 * cm, 0.18 * cm, 0.3744 * cm, 0),
         (55.710586181217394, 0.36, 0.007344262785151853)),
        (4, (0.36 * cm, 0.18 * cm, 3.705e-6, 0),
         (13.120297517162507, -2.0082590717125437, 0.24857168491193957)),
        (5, (2.67, 0.36 * cm, 0.3744 * cm, 8.55e-6),
         (24.080107809942664, 0.28666015630152986, 0.016567297254868083)),
        (6, (0.18 * cm, 0.36 * cm, 8.55e-6, 0), (0.0, -4.967578, -4.967578)),
    )
    for which, args, expected in cases:
        out = (ctypes.c_double * 3)()
        fun(ctypes.c_int(which), *(ctypes.c_double(v) for v in args), ctypes.c_double(cm), out)
        np.testing.assert_array_almost_equal(list(out), expected, err_msg=str(which))
