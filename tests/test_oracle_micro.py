"""Pins each oracle function against the per-method goldens recorded from the reference, and
against the known-answer tables of the reference's own unit tests.  CPU only."""
import pytest

from . import known_answers as ka
from . import micro_cases as mc

EXACT_ON_CPU = {"volume", "radius", "velocity", "golovin", "geometric", "berry1967",
                "straub2010", "frag_always_n_4"}


@pytest.fixture(scope="module", name="kit")
def kit_fixture(oracle_backend_class):
    return mc.Kit(oracle_backend_class, fragmentation_function="Straub2010Nf")


@pytest.mark.parametrize("check", [mc.check_pcg64, mc.check_shuffle,
                                   mc.check_shuffle_known_answers, mc.check_counting_sort,
                                   mc.check_sort_by_key_and_adaptive_end, mc.check_remove_zero,
                                   mc.check_pair_chain, mc.check_moments, mc.check_storage_ops])
def test_method_goldens(check, kit):
    check(kit)


def test_physics_goldens(kit):
    # numpy evaluates log/exp through its own SIMD loops, glibc's differ in the last bit: the
    # transcendental-heavy fragmentation volumes are compared at 1e-14, the rest bit-exactly
    mc.check_physics(kit, exact=EXACT_ON_CPU, rtol=1e-14)


@pytest.mark.parametrize("check", ka.ALL_CHECKS)
def test_reference_known_answers(check, kit):
    check(kit)
