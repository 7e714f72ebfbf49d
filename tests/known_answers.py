"""Known-answer tables of the reference's own unit tests for this path, re-typed (data only) and
driven through this package's backend interface; run with the oracle backend on CPU and with the
HIP backend on the GPU.

Sources (reference tree): tests/unit_tests/backends/test_collisions_methods.py:82-204 (adaptive
scaling, 5 cases) and :239-336 (the adaptivity "paper diagram" scenario);
tests/unit_tests/dynamics/collisions/test_sdm_single_cell.py:73-98 (same-multiplicity split) and
:215-258 (gamma formula grid); tests/unit_tests/dynamics/collisions/test_sdm_breakup.py:232-420
(ten single-breakup answers).
"""
import numpy as np

from pysdm_amd.dynamics.collisions import DEFAULTS

SCALE_PROB_CASES = (
    # gamma, idx, n, cell_id, dt_left, dt, dt_max, is_first_in_pair, expected_dt_left, n_substep
    ((10.0,), (0, 1), (44, 44), (0, 0), (10.0,), 10.0, 10.0, (True, False), (9.0,), (1,)),
    ((10.0,), (0, 1), (44, 44), (0, 0), (10.0,), 10.0, 0.1, (True, False), (9.9,), (1,)),
    ((0.0,), (0, 1), (44, 44), (0, 0), (10.0,), 10.0, 10.0, (False, True), (0.0,), (1,)),
    ((10.0,), (0, 1), (440, 44), (0, 0), (10.0,), 10.0, 10.0, (True, False), (0.0,), (1,)),
    ((0.5, 6), (0, 1, 2, 3, 4), (44, 44, 22, 33, 11), (0, 0, 0, 1, 1), (10.0, 10), 10.0, 10.0,
     (True, False, False, True, False), (0.0, 5.0), (1, 1)),
)


def check_scale_prob_known_answers(kit):
    for gamma, idx, n, cell_id, dt_left, dt, dt_max, first, exp_dt_left, exp_n_substep in \
            SCALE_PROB_CASES:
        _gamma = kit.Storage.from_ndarray(np.asarray(gamma, dtype=float))
        _idx = kit.Index.from_ndarray(np.asarray(idx, dtype=np.int64))
        _n = kit.IndexedStorage.from_ndarray(_idx, np.asarray(n, dtype=np.int64))
        _cell_id = kit.Storage.from_ndarray(np.asarray(cell_id, dtype=np.int64))
        _dt_left = kit.Storage.from_ndarray(np.asarray(dt_left, dtype=float))
        flag = kit.PairIndicator(len(n))
        flag.indicator.upload(np.asarray(first, dtype=bool))
        n_substep = kit.Storage.from_ndarray(np.zeros(len(dt_left), dtype=np.int64))
        dt_min_stat = kit.Storage.from_ndarray(np.zeros(len(dt_left)))
        kit.backend.scale_prob_for_adaptive_sdm_gamma(
            prob=_gamma, multiplicity=_n, cell_id=_cell_id, dt_left=_dt_left, dt=dt,
            dt_range=(np.nan, dt_max), is_first_in_pair=flag, stats_n_substep=n_substep,
            stats_dt_min=dt_min_stat)
        np.testing.assert_array_almost_equal(_dt_left.to_ndarray(), exp_dt_left)
        out = _gamma.to_ndarray()
        for i, is_first in enumerate(first):
            if is_first:
                expected = (dt - exp_dt_left[cell_id[i]]) / dt * gamma[i // 2]
                np.testing.assert_almost_equal(out[i // 2], expected)
        np.testing.assert_array_equal(n_substep.to_ndarray(), exp_n_substep)


def check_adaptivity_paper_diagram(kit):
    n_part, dt = 16, 12.0
    gamma = [2.0, 2, 2] + [3.0, 3] + [1.0, 1, 1]
    multiplicity = [1] * 2 + [100] * (n_part - 2)
    cell_id = [0] * 6 + [1] * 4 + [2] * 6
    idx = kit.Index.from_ndarray(np.arange(n_part, dtype=np.int64))
    mult = kit.IndexedStorage.from_ndarray(idx, np.asarray(multiplicity, dtype=np.int64))
    cell = kit.IndexedStorage.from_ndarray(idx, np.asarray(cell_id, dtype=np.int64))
    dt_left = kit.Storage.from_ndarray(np.full(3, dt))
    first = np.asarray([True, False] * (n_part // 2))
    flag = kit.PairIndicator(n_part)
    n_substep = kit.Storage.from_ndarray(np.zeros(3, dtype=np.int64))
    dt_min_stat = kit.Storage.from_ndarray(np.full(3, dt))
    attrs = kit.IndexedStorage.from_ndarray(idx, np.zeros((0, n_part)))
    healthy = kit.Storage.from_ndarray(np.full((1,), 1))
    coal = kit.Storage.from_ndarray(np.zeros(3, dtype=np.int64))
    for _ in range(100):
        if not (dt_left.to_ndarray() > 0).any():
            break
        flag.indicator.upload(first)
        _gamma = kit.PairwiseStorage.from_ndarray(np.asarray(gamma, dtype=float))
        kit.backend.scale_prob_for_adaptive_sdm_gamma(
            prob=_gamma, multiplicity=mult, cell_id=cell, dt_left=dt_left, dt=dt,
            dt_range=(np.nan, dt), is_first_in_pair=flag, stats_n_substep=n_substep,
            stats_dt_min=dt_min_stat)
        scaled = _gamma.to_ndarray()
        assert (scaled == scaled.astype(int)).all()
        kit.backend.collision_coalescence(
            multiplicity=mult, idx=idx, attributes=attrs, gamma=_gamma, healthy=healthy,
            cell_id=cell, coalescence_rate=coal, is_first_in_pair=flag)
        now = mult.to_ndarray(raw=True)
        for i in range(n_part // 2):  # the test's bookkeeping: emptied pairs stop colliding
            if scaled[i] != 0 and now[2 * i] == 0:
                first[2 * i] = False
                gamma[i] = 0.0
    np.testing.assert_array_equal(n_substep.to_ndarray(), (2, 3, 1))
    assert (dt_left.to_ndarray() == 0.0).all()
    np.testing.assert_array_equal(dt_min_stat.to_ndarray(), dt / np.asarray([2, 3, 1]))
    np.testing.assert_array_equal(
        mult.to_ndarray(raw=True), (0, 1, 25, 25, 25, 25, 12, 13, 12, 13, 50, 50, 50, 50, 50, 50))


def check_gamma_formula_grid(kit):
    """gamma = floor(p) + (r < p - floor(p)) over an 87 x 87 grid, all pairs in one call"""
    n = 87
    prob, rand = np.meshgrid(np.linspace(0, 3, n, endpoint=True),
                             np.linspace(0, 1, n, endpoint=False), indexing="ij")
    prob, rand = prob.ravel(), rand.ravel()
    expected = prob // 1 + (rand < prob - prob // 1)
    n_pairs = prob.size
    idx = kit.Index.identity_index(2 * n_pairs)
    mult_host = np.ones(2 * n_pairs, dtype=np.int64)
    mult_host[0::2] = np.maximum(expected, 1).astype(np.int64)  # never caps gamma
    mult = kit.IndexedStorage.from_ndarray(idx, mult_host)
    cell = kit.IndexedStorage.from_ndarray(idx, np.zeros(2 * n_pairs, dtype=np.int64))
    flag = kit.PairIndicator(2 * n_pairs)
    flag.indicator.upload(np.tile([True, False], n_pairs))
    prob_s = kit.PairwiseStorage.from_ndarray(prob)
    rates = [kit.Storage.from_ndarray(np.zeros(1, dtype=np.int64)) for _ in range(2)]
    kit.backend.compute_gamma(prob=prob_s, rand=kit.Storage.from_ndarray(rand),
                              multiplicity=mult, cell_id=cell, collision_rate=rates[0],
                              collision_rate_deficit=rates[1], is_first_in_pair=flag,
                              out=prob_s)
    np.testing.assert_array_equal(prob_s.to_ndarray(), expected)
    assert rates[1].to_ndarray()[0] == 0


def check_same_multiplicity_split(kit):
    for n_in, n_out in ((1, (1, 0)), (2, (1, 1)), (3, (2, 1))):
        idx = kit.Index.identity_index(2)
        mult = kit.IndexedStorage.from_ndarray(idx, np.full(2, n_in, dtype=np.int64))
        attrs = kit.IndexedStorage.from_ndarray(idx, np.full((1, 2), 1.0))
        flag = kit.PairIndicator(2)
        flag.indicator.upload(np.asarray([True, False]))
        healthy = kit.Storage.from_ndarray(np.full((1,), 1))
        kit.backend.collision_coalescence(
            multiplicity=mult, idx=idx, attributes=attrs,
            gamma=kit.PairwiseStorage.from_ndarray(np.asarray([1.0])), healthy=healthy,
            cell_id=kit.IndexedStorage.from_ndarray(idx, np.zeros(2, dtype=np.int64)),
            coalescence_rate=kit.Storage.from_ndarray(np.zeros(1, dtype=np.int64)),
            is_first_in_pair=flag)
        np.testing.assert_array_equal(sorted(mult.to_ndarray(raw=True)), sorted(n_out))
        assert bool(healthy.to_ndarray()[0]) == (0 not in n_out)


SINGLE_BREAKUP_CASES = (
    # gamma, n_init, v_init, n_expected, v_expected, expected_deficit, frag_volume
    (1.0, (1, 1), (1, 1), (2, 2), (0.5, 0.5), 0.0, 0.5),
    (2.0, (20, 4), (1, 2), (4, 24), (1, 1), 0.0, 1.0),
    (2.0, (1, 1), (1, 1), (2, 2), (0.5, 0.5), 1.0, 0.5),
    (2.0, (3, 1), (1, 1), (2, 4), (1.0, 0.5), 1.0, 0.5),
    (2.0, (9, 2), (1, 2), (1, 12), (1, 1), 0.0, 1.0),
    (1.0, (12, 1), (1, 1), (11, 2), (1, 1), 0.0, 1.0),
    (1.0, (15, 2), (2, 6), (13, 4), (2, 4), 0.0, 4),
    (1.0, (13, 4), (2, 4), (9, 6), (2, 4), 0.0, 4),
    (3.0, (15, 2), (2, 6), (3, 9), (2, 4), 0.0, 4),
    (0.0, (15, 2), (2, 6), (15, 2), (2, 6), 0.0, 4),
)


def check_single_breakup_known_answers(kit):
    rho_w = kit.backend.formulae.constants.rho_w
    for gamma, n_init, v_init, n_exp, v_exp, deficit, frag_volume in SINGLE_BREAKUP_CASES:
        idx = kit.Index.identity_index(2)
        mult = kit.IndexedStorage.from_ndarray(idx, np.asarray(n_init, dtype=np.int64))
        mass = rho_w * np.asarray(v_init, dtype=float)
        attrs = kit.IndexedStorage.from_ndarray(idx, mass.reshape(1, 2).copy())
        flag = kit.PairIndicator(2)
        flag.indicator.upload(np.asarray([True, False]))

        def pairwise(value):
            return kit.PairwiseStorage.from_ndarray(np.asarray([value], dtype=float))

        def counter():
            return kit.Storage.from_ndarray(np.zeros(1, dtype=np.int64))

        breakup_rate, breakup_deficit = counter(), counter()
        kit.backend.collision_coalescence_breakup(
            multiplicity=mult, idx=idx, attributes=attrs, gamma=pairwise(gamma),
            rand=pairwise(1.0), Ec=pairwise(0.0), Eb=pairwise(1.0),
            fragment_mass=pairwise(rho_w * frag_volume),
            healthy=kit.Storage.from_ndarray(np.full((1,), 1)),
            cell_id=kit.IndexedStorage.from_ndarray(idx, np.zeros(2, dtype=np.int64)),
            coalescence_rate=counter(), breakup_rate=breakup_rate,
            breakup_rate_deficit=breakup_deficit, is_first_in_pair=flag, warn_overflows=False,
            particle_mass=attrs[0, :], max_multiplicity=DEFAULTS.max_multiplicity)
        case = f"gamma={gamma} n={n_init} v={v_init}"
        np.testing.assert_array_equal(mult.to_ndarray(raw=True), n_exp, err_msg=case)
        volumes = attrs.to_ndarray(raw=True)[0] / rho_w
        np.testing.assert_array_almost_equal(volumes, v_exp, err_msg=case)
        np.testing.assert_almost_equal(np.sum(np.asarray(n_exp) * volumes),
                                       np.sum(np.asarray(n_init) * np.asarray(v_init)))
        np.testing.assert_almost_equal(breakup_deficit.to_ndarray()[0], deficit)
        assert breakup_rate.to_ndarray()[0] == gamma * min(n_init) - deficit * min(n_init) \
            or gamma == 0


ALL_CHECKS = (check_scale_prob_known_answers, check_adaptivity_paper_diagram,
              check_gamma_formula_grid, check_same_multiplicity_split,
              check_single_breakup_known_answers)
