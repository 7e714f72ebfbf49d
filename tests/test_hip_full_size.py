"""GPU parity at BASELINE.json's full sizes.  The oracle still finishes in seconds per step at
n_sd = 2^20 (serial C), so the first steps are compared directly; on top of that size-independent
properties are checked (idx is a permutation of the live droplets, mass conservation, droplet
count only decreases by coalescence, the two HIP routes agree)."""
import warnings

import numpy as np
import pytest

from pysdm_amd.examples import CONFIGS, make_box, make_kinematic_flow

from .trajectory import snapshot

pytestmark = pytest.mark.gpu


def box(backend_class, name, adaptive, fused):
    return make_box(backend_class, name, adaptive=adaptive, fused=fused)


def run(particulator, steps):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        particulator.run(steps)


def assert_same(a, b, float_rtol=0.0):
    length = int(a["length"])
    for key, value in a.items():
        ref = b[key]
        if key == "idx":
            value, ref = value[:length], ref[:length]
        if float_rtol and value.dtype.kind == "f":
            np.testing.assert_allclose(value, ref, rtol=float_rtol, atol=0, err_msg=key)
        else:
            np.testing.assert_array_equal(value, ref, err_msg=key)


def invariants(snap, n_sd, total_mass0, rtol):
    length = int(snap["length"])
    live = snap["idx"][:length]
    assert len(np.unique(live)) == length and live.min() >= 0 and live.max() < n_sd
    n, m = snap["multiplicity"], snap["attributes"][0]
    assert (n[live] > 0).all()
    np.testing.assert_allclose(np.sum(n[live].astype(float) * m[live]), total_mass0, rtol=rtol)


@pytest.mark.parametrize("name,adaptive,steps", [
    ("shima", False, 3), ("shima", True, 2), ("berry_breakup", True, 2),
    ("straub", True, 1), ("kinematic2d", True, 2),
])
def test_full_size_fused_equals_oracle(name, adaptive, steps, hip_backend_class,
                                       oracle_backend_class):
    snaps = []
    for backend_class in (hip_backend_class, oracle_backend_class):
        particulator, dynamic = box(backend_class, name, adaptive, None)
        # (taken for both backends: reading cell_start sorts by cell, which must happen at the
        # same point of both histories -- before the first sort_by_key of an adaptive step)
        first = snapshot(particulator, dynamic)
        live = first["idx"][: int(first["length"])]
        mass0 = np.sum(first["multiplicity"][live].astype(float) * first["attributes"][0][live])
        run(particulator, steps)
        snaps.append(snapshot(particulator, dynamic))
    breakup = "breakup" in name or "straub" in name
    assert_same(snaps[0], snaps[1], float_rtol=1e-12 if breakup else 0.0)
    invariants(snaps[0], CONFIGS[name]["n_sd"], mass0, rtol=1e-9 if breakup else 1e-12)


@pytest.mark.parametrize("name,adaptive,steps", [("shima", True, 20), ("kinematic2d", True, 2)])
def test_full_size_routes_agree(name, adaptive, steps, hip_backend_class):
    snaps = []
    for fused in (None, False):
        particulator, dynamic = box(hip_backend_class, name, adaptive, fused)
        run(particulator, steps)
        snaps.append(snapshot(particulator, dynamic))
    assert_same(snaps[0], snaps[1])


def test_full_size_displacement_then_collisions(hip_backend_class, oracle_backend_class):
    """configs[3] with its preceding step: 2^22 super-droplets advected by the single-eddy flow
    and sedimenting through a 32 x 32 grid (precipitation leaves through the bottom), then the
    adaptive Geometric collision step on the unsorted state; HIP (fused collision route) against
    the oracle"""
    results = []
    for backend_class in (hip_backend_class, oracle_backend_class):
        particulator, displacement, collision = make_kinematic_flow(backend_class)
        rain = []
        for _ in range(2):
            run(particulator, 1)
            rain.append(displacement.precipitation_mass_in_last_step)
        attrs = particulator.attributes
        snap = snapshot(particulator, collision)
        snap["cell_origin"] = attrs["cell origin"].to_ndarray(raw=True)
        snap["cell_id"] = attrs["cell id"].to_ndarray(raw=True)
        results.append((snap, attrs["position in cell"].to_ndarray(raw=True), rain))
    (hip, hip_pos, hip_rain), (ref, ref_pos, ref_rain) = results
    length = int(hip["length"])
    assert length == int(ref["length"]) and length < 2**22  # some rain left the domain
    live = hip["idx"][:length]
    np.testing.assert_array_equal(live, ref["idx"][:length])
    for key in ("cell_origin", "cell_id", "multiplicity", "attributes"):
        np.testing.assert_array_equal(hip[key][..., live], ref[key][..., live], err_msg=key)
    for key in ("cell_start", "collision_rate", "coalescence_rate", "stats_n_substep"):
        np.testing.assert_array_equal(hip[key], ref[key], err_msg=key)
    np.testing.assert_allclose(hip_pos[:, live], ref_pos[:, live], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(hip_rain, ref_rain, rtol=1e-12)
    assert hip_rain[0] > 0


@pytest.mark.parametrize("n_sd,steps,thin", [(2**20, 12, None), (2**16, 12, 0.02),
                                             (2**16, 40, 100.0), (2**24, 2, None)])
def test_many_steps_in_one_call_equal_oracle(n_sd, steps, thin, hip_backend_class,
                                             oracle_backend_class):
    """`Particulator.run(n)` of the single-cell non-adaptive box = one `sdm_collision_run` call: n
    time steps without the host in between.  `thin` (a cell volume): multiplicities of 1..3, so
    that super-droplets die -- in every step (0.02) or once in a few steps (100) -- and the
    device-gated compaction has to run in the middle of a call."""
    snaps = []
    for backend_class in (hip_backend_class, oracle_backend_class):
        particulator, dynamic = make_box(backend_class, "shima", n_sd=n_sd, adaptive=False,
                                         dt=200.0 if thin else None)
        if thin:
            mult = particulator.attributes["multiplicity"]
            mult.upload((1 + np.arange(n_sd) % 3).astype(np.int64))
            particulator.attributes.mark_updated("multiplicity")
            particulator.environment.mesh.dv = thin * n_sd / 2**16
        run(particulator, 1)
        run(particulator, steps)
        run(particulator, 5)
        snaps.append(snapshot(particulator, dynamic))
    assert_same(snaps[0], snaps[1])
    if thin:
        assert int(snaps[0]["length"]) < n_sd


@pytest.mark.parametrize("name,n_sd,steps,dt,thin", [
    ("shima", 2**16, 40, None, None),      # one sub-step per step
    ("shima", 2**16, 25, 400.0, None),     # several sub-steps per step
    ("shima", 2**16, 25, 200.0, 0.02),     # super-droplets die in every step
    ("shima", 2**12, 60, 200.0, 0.02),     # below the look-ahead's size threshold
    ("berry_breakup", 2**15, 40, None, None),
    ("straub", 2**14, 12, None, None),
])
def test_adaptive_steps_in_one_call_equal_oracle(name, n_sd, steps, dt, thin, hip_backend_class,
                                                 oracle_backend_class):
    """`Particulator.run(n)` of an adaptive single-cell box in one `sdm_collision_run` call: the
    head of each next sub-step (draw, shuffle build, probabilities) is launched ahead of the
    read-back that decides whether it continues the time step or opens the next one; state,
    counters and sub-step statistics equal the oracle's step-by-step run"""
    snaps = []
    for backend_class in (hip_backend_class, oracle_backend_class):
        particulator, dynamic = make_box(backend_class, name, n_sd=n_sd, adaptive=True, dt=dt)
        if thin:
            mult = particulator.attributes["multiplicity"]
            mult.upload((1 + np.arange(n_sd) % 3).astype(np.int64))
            particulator.attributes.mark_updated("multiplicity")
            particulator.environment.mesh.dv = thin * n_sd / 2**16
        run(particulator, 1)
        run(particulator, steps)
        run(particulator, 3)
        snaps.append(snapshot(particulator, dynamic))
    assert_same(snaps[0], snaps[1], float_rtol=0.0 if name == "shima" else 1e-12)
    assert snaps[0]["stats_n_substep"][0] >= steps + 4
    if dt:
        assert snaps[0]["stats_n_substep"][0] > steps + 4
    if thin:
        assert int(snaps[0]["length"]) < n_sd


@pytest.mark.parametrize("n_sd", [2**21 - 2, 2**21 - 1, 2**20 - 3])
def test_record_layouts_at_their_size_limits(n_sd, hip_backend_class, oracle_backend_class):
    """the shuffle records switch layout with the size (21-bit fields with four inline hits up to
    2^21 - 2 positions, 24-bit fields with three above): same permutation and state either side
    of the limit, and for an odd count (an unpaired last position)"""
    snaps = []
    for backend_class in (hip_backend_class, oracle_backend_class):
        particulator, dynamic = make_box(backend_class, "shima", n_sd=n_sd, adaptive=False)
        run(particulator, 3)
        snaps.append(snapshot(particulator, dynamic))
    assert_same(snaps[0], snaps[1])


def test_shima_box_3600_steps_equal_oracle(hip_backend_class, oracle_backend_class):
    """the whole Shima-2009 experiment (3600 steps of 1 s) in one library call at 2^16
    super-droplets: permutation, multiplicities, masses and counters identical to the oracle's"""
    snaps = []
    for backend_class in (hip_backend_class, oracle_backend_class):
        particulator, dynamic = make_box(backend_class, "shima", n_sd=2**16, adaptive=False)
        run(particulator, 3600)
        snaps.append(snapshot(particulator, dynamic))
    assert_same(snaps[0], snaps[1])
