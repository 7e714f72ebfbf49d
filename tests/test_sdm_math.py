"""csrc/sdm_math.h: ONE implementation of pow / exp / log / log1p / sinh / asinh / atanh / erf for
the product (hipcc) and the checker (gcc), so that both return the same bits (VERDICT r2, item 2:
breakup runs diverged in the integers after ~150 steps because device libm and glibc differ in
the last bit now and then).

CPU: accuracy of the functions against mpmath (through the checker's `sdm_math_eval`), special
values, and that the committed tables are what the generator produces.
GPU: HIP == checker, bit for bit, on a few million arguments per function.
"""
import math
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FN = {"exp": 0, "log": 1, "pow": 2, "sinh": 3, "asinh": 4, "atanh": 5, "erf": 6, "log1p": 7}


def evaluate(engine, name, a, b=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.zeros_like(a) if b is None else np.ascontiguousarray(b, dtype=np.float64)
    out = engine.empty(a.shape, np.float64)
    engine.call("sdm_math_eval", FN[name], out, engine.upload(a), engine.upload(b), a.size)
    return engine.download(out)


def arguments(name, n, rng):
    """(a, b): the ranges the path uses, the whole domain, and the neighbourhood of the awkward
    points (1 for log / pow, 0 for the odd functions, the interval ends of the tables)"""
    u = rng.uniform
    if name == "exp":
        return np.concatenate([u(-745, 709.7, n), rng.normal(0, 1, n), rng.normal(0, 1e-5, n)]), None
    if name == "log":
        return np.concatenate([np.exp(u(-700, 700, n)), u(0.5, 2, n), 1 + rng.normal(0, 1e-4, n),
                               u(0, 1, 16) * 5e-310, 181 / 128 + rng.normal(0, 1e-9, 64)]), None
    if name == "log1p":
        return np.concatenate([rng.normal(0, 1e-3, n), u(-0.99, 50, n), np.exp(u(-60, 30, n))]), None
    if name == "sinh":
        return np.concatenate([u(-3, 3, n), u(-709, 709, n), rng.normal(0, 1e-3, n)]), None
    if name == "asinh":
        return np.concatenate([u(-3, 3, n), np.exp(u(-30, 300, n)), rng.normal(0, 1e-3, n)]), None
    if name == "atanh":
        return np.concatenate([u(-1, 1, n), 1 - np.exp(u(-36, 0, n)), rng.normal(0, 1e-3, n)]), None
    if name == "erf":
        return np.concatenate([u(-6.5, 6.5, n), rng.normal(0, 0.3, n), np.exp(u(-40, 0, n)),
                               np.arange(1, 25) / 4 + rng.normal(0, 1e-12, 24)]), None
    x = np.concatenate([np.exp(u(-40, 40, n)), u(0.5, 2, n), np.exp(u(-700, 700, n))])
    y = np.concatenate([u(-5, 5, n), u(-300, 300, n), u(-1, 1, n)])
    for k, value in enumerate((1 / 3, 2 / 3, 3.0, 1.5, -1.22, 2.5, -0.718)):  # the path's exponents
        y[k::11] = value
    return x, y


MAX_ULP = {"exp": 0.7, "log": 0.51, "pow": 0.7, "log1p": 0.7, "sinh": 2.0, "asinh": 2.0,
           "atanh": 2.0, "erf": 2.0}


@pytest.mark.parametrize("name", sorted(FN))
def test_accuracy_against_mpmath(name, oracle_engine):
    mp = pytest.importorskip("mpmath")
    mp.mp.dps = 60
    exact_fn = {"exp": mp.exp, "log": mp.log, "log1p": mp.log1p, "sinh": mp.sinh,
                "asinh": mp.asinh, "atanh": mp.atanh, "erf": mp.erf, "pow": mp.power}[name]
    a, b = arguments(name, 1500, np.random.default_rng(3))
    got = evaluate(oracle_engine, name, a, b)
    worst, not_rounded = 0.0, 0
    for k, value in enumerate(got):
        exact = exact_fn(mp.mpf(float(a[k])), mp.mpf(float(b[k]))) if b is not None else exact_fn(
            mp.mpf(float(a[k])))
        if not 1e-300 < abs(exact) < 1e300:
            continue  # (subnormal results round twice; overflow is checked with the special values)
        error = float(abs(mp.mpf(float(value)) - exact) / math.ulp(float(exact)))
        worst = max(worst, error)
        not_rounded += error > 0.5
    assert worst <= MAX_ULP[name], (name, worst)
    if name in ("exp", "log", "pow"):  # correctly rounded but for a few per thousand
        assert not_rounded <= 0.004 * len(got)


def test_special_values(oracle_engine):
    inf, nan = np.inf, np.nan
    e = lambda name, a, b=None: evaluate(oracle_engine, name, np.atleast_1d(a), None if b is None else np.atleast_1d(b))  # noqa: E731
    np.testing.assert_array_equal(e("exp", [0.0, -inf, inf, 710.0, -746.0]), [1, 0, inf, inf, 0])
    assert np.isnan(e("exp", [nan])[0])
    np.testing.assert_array_equal(e("log", [1.0, 0.0, inf]), [0, -inf, inf])
    assert np.isnan(e("log", [-1.0, nan])).all()
    assert e("log", [5e-324])[0] == math.log(5e-324)
    x = np.array([2.0, 3.0, 0.0, 0.0, -8.0, -8.0, 4.0, 7.0, 1.0, inf, 0.5, -2.0, 1e-300])
    y = np.array([2.0, 0.0, 2.0, -1.0, 3.0, 2.0, 0.5, 1.0, nan, -1.0, inf, 0.5, 2.0])
    got = e("pow", x, y)
    want = np.array([4.0, 1.0, 0.0, inf, -512.0, 64.0, 2.0, 7.0, 1.0, 0.0, 0.0, nan, 0.0])
    np.testing.assert_array_equal(got, want)
    # squares are exact products, square roots correctly rounded (the reference's x**2, x**0.5)
    v = np.random.default_rng(0).uniform(1e-9, 1e3, 1000)
    np.testing.assert_array_equal(e("pow", v, np.full_like(v, 2.0)), v * v)
    np.testing.assert_array_equal(e("pow", v, np.full_like(v, 0.5)), np.sqrt(v))
    np.testing.assert_array_equal(e("erf", [0.0, 7.0, -7.0, inf]), [0, 1, -1, 1])
    np.testing.assert_array_equal(e("atanh", [1.0, -1.0, 0.0]), [inf, -inf, 0])
    for name in ("sinh", "asinh", "atanh", "erf"):  # odd functions, exactly
        a, _ = arguments(name, 200, np.random.default_rng(5))
        np.testing.assert_array_equal(e(name, -a), -e(name, a))


def test_committed_tables_are_the_generated_ones():
    pytest.importorskip("mpmath")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "gen_sdm_math_tables.py"),
                           "--check"])


def test_no_libm_transcendental_is_left_on_the_path():
    """both libraries take these functions from sdm_math.h only"""
    import re  # pylint: disable=import-outside-toplevel

    call = re.compile(r"(?<![A-Za-z0-9_])(pow|exp|log|log1p|erf|sinh|asinh|atanh|cbrt|expm1)\(")
    sources = [os.path.join(ROOT, "oracle", f) for f in ("sdm_oracle.c", "sdm_oracle_abi.c")]
    csrc = os.path.join(ROOT, "pysdm_amd", "csrc")
    sources += [os.path.join(csrc, f) for f in os.listdir(csrc)
                if f.endswith((".hip", ".h")) and not f.startswith("sdm_math")]
    for path in sources:
        with open(path, encoding="utf-8") as f:
            text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
        for number, line in enumerate(text.splitlines(), 1):
            code = line.split("//")[0]
            assert not call.search(code), f"{path}:{number}: {line.strip()}"


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(FN))
def test_hip_returns_the_checkers_bits(name, hip_engine, oracle_engine):
    a, b = arguments(name, 700_000, np.random.default_rng(11))
    special = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 5e-324, 2.2e-308, 1.7e308,
                        0.25, 6.0, 181 / 128, 1.4140625, 709.782712893384, -745.14])
    a = np.concatenate([a, special])
    if b is not None:
        b = np.concatenate([b, special[::-1]])
    got, want = evaluate(hip_engine, name, a, b), evaluate(oracle_engine, name, a, b)
    canonical = np.uint64(0x7ff8000000000000)  # (NaN payloads are not part of the contract)
    np.testing.assert_array_equal(np.where(np.isnan(got), canonical, got.view(np.uint64)),
                                  np.where(np.isnan(want), canonical, want.view(np.uint64)))
