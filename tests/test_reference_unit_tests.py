"""SURVEY.md 8(f-4), second clause: the reference's OWN unit tests of the path - the files under
/root/reference/tests/unit_tests the survey names, untouched, where they lie - run against this
package's backend class (scripts/run_reference_unit_tests.py; tests/helpers/
reference_unit_plugin.py binds `PySDM.backends.CPU` to it).  Build container only: skipped where
the reference tree is absent (it never travels to the GPU box).  The tracked record of a run is
profiles/r04_reference_unit_tests.txt."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(
    not os.path.isdir("/root/reference/tests/unit_tests"), reason="reference tree not present")


def test_the_references_own_unit_tests_pass_on_the_plugged_class(tmp_path):
    report = tmp_path / "report.txt"
    done = subprocess.run(
        [sys.executable, "-B", os.path.join(ROOT, "scripts", "run_reference_unit_tests.py"),
         "--extra", "--report", str(report)],
        capture_output=True, text=True, timeout=1500, cwd=ROOT, check=False)
    text = report.read_text(encoding="utf-8") if report.exists() else ""
    assert done.returncode == 0, done.stdout[-3000:] + done.stderr[-3000:]  # no unexplained non-pass
    lines = text.splitlines()
    passed = [line for line in lines if line.startswith("passed ")]
    failed = [line for line in lines if line.startswith("failed ")]
    # every case of the four backend files and of dynamics/collisions, moments, displacement,
    # storage operations: all pass but the one that calls a private njit body of the Numba class
    assert len(passed) >= 262, text[:2000]
    assert len(failed) == 1 and "test_sum_pair_body_out_of_bounds" in failed[0], failed
    for must in ("test_sdm_breakup.py", "test_sdm_single_cell.py", "test_sdm_multi_cell.py",
                 "test_croupiers.py", "test_collisions_methods.py", "test_particle_attributes.py",
                 "test_index.py", "test_pair_methods.py"):
        assert any(must in line for line in passed), must
