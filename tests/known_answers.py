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


def _two_drop_setup(kit, component, volume, fragmentation_function=None, water_mass=None):
    """the arrangement shared by the reference's component tests: two droplets, one pair
    (tests/unit_tests/dynamics/collisions/test_fragmentations.py:44-71 and siblings)"""
    from pysdm_amd.environments import Box  # pylint: disable=import-outside-toplevel
    from pysdm_amd.formulae import Formulae  # pylint: disable=import-outside-toplevel
    from pysdm_amd.particulator import Builder  # pylint: disable=import-outside-toplevel

    kwargs = {} if fragmentation_function is None else {
        "fragmentation_function": fragmentation_function}
    n_sd = len(volume if volume is not None else water_mass)
    builder = Builder(n_sd, kit.backend.__class__(Formulae(**kwargs)),
                      environment=Box(dv=None, dt=None))
    component.register(builder)
    attributes = {"multiplicity": np.ones(n_sd)}
    if volume is not None:
        attributes["volume"] = np.asarray(volume)
    else:
        attributes["water mass"] = np.asarray(water_mass)
    particulator = builder.build(attributes=attributes)
    flag = particulator.PairIndicator(length=n_sd)
    flag.indicator = particulator.Storage.from_ndarray(np.asarray([True, False]))
    return particulator, flag


UM3 = 1e-18  # si.um**3
RHO_W = 1000.0


def _fragmentation_classes():
    from pysdm_amd.dynamics import collisions as C  # pylint: disable=import-outside-toplevel

    return C


def _run_fragmentation(kit, make, volume, u01=0.5, vmin=None):
    sut = make()
    if vmin is not None:
        sut.vmin = vmin
    particulator, flag = _two_drop_setup(kit, sut, volume, sut.__class__.__name__)
    nf = particulator.PairwiseStorage.from_ndarray(np.zeros(1))
    frag_mass = particulator.PairwiseStorage.from_ndarray(np.zeros(1))
    rand = particulator.PairwiseStorage.from_ndarray(np.asarray([u01], dtype=float))
    sut(nf, frag_mass, rand, flag)
    return nf.to_ndarray(), frag_mass.to_ndarray()


def check_reference_fragmentation_tests(kit):
    """test_fragmentations.py:30-262 re-typed: call, vmin / vmax / nfmax limiters, and the
    100-point sweep over u01 for two rain-size drops"""
    C = _fragmentation_classes()
    volume = np.asarray([440.0 * UM3, 6660.0 * UM3])
    total = np.sum(volume) * RHO_W
    # :30-84 call
    for make in (lambda: C.AlwaysN(n=2), lambda: C.Exponential(scale=1e6 * UM3),
                 lambda: C.Feingold1988(scale=1e6 * UM3),
                 lambda: C.Gaussian(mu=2e6 * UM3, sigma=1e6 * UM3), C.SLAMS, C.Straub2010Nf,
                 C.LowList1982Nf):
        nf, frag_mass = _run_fragmentation(kit, make, volume, vmin=1 * UM3)
        assert (nf > 0.99).all() and (frag_mass > 0).all()
        np.testing.assert_approx_equal(nf[0] * frag_mass[0], total)
    # :86-146 vmin limiter: one fragment holding all the mass
    for make in (lambda: C.Exponential(scale=1 * UM3, vmin=6660.0 * UM3),
                 lambda: C.Feingold1988(scale=1 * UM3, vmin=6660.0 * UM3),
                 lambda: C.Gaussian(mu=2 * UM3, sigma=1 * UM3, vmin=6660.0 * UM3),
                 lambda: C.SLAMS(vmin=6660.0 * UM3), lambda: C.Straub2010Nf(vmin=6660.0 * UM3)):
        nf, frag_mass = _run_fragmentation(kit, make, volume)
        np.testing.assert_array_equal([1.0], nf)
        np.testing.assert_array_equal([(6660.0 + 440.0) * UM3 * RHO_W], frag_mass)
    # :148-204 vmax limiter
    for make in (lambda: C.Exponential(scale=1.0 * 1e-6), lambda: C.Feingold1988(scale=1.0 * 1e-6),
                 lambda: C.Gaussian(mu=1.0 * 1e-6, sigma=1e6 * UM3), C.SLAMS, C.Straub2010Nf):
        nf, frag_mass = _run_fragmentation(kit, make, volume, vmin=1 * UM3)
        assert (nf > 0.999).all()
        assert (frag_mass < (6661.0 + 440.0) * UM3 * RHO_W).all()
        np.testing.assert_approx_equal(nf[0] * frag_mass[0], total)
    # :206-262 nfmax limiter
    for make in (lambda: C.Exponential(scale=1.0 * UM3, nfmax=2),
                 lambda: C.Feingold1988(scale=1.0 * UM3, nfmax=2),
                 lambda: C.Gaussian(mu=1.0 * UM3, sigma=1e6 * UM3, nfmax=2),
                 lambda: C.SLAMS(nfmax=2), lambda: C.Straub2010Nf(nfmax=2)):
        nf, frag_mass = _run_fragmentation(kit, make, volume, vmin=1 * UM3)
        assert (nf < 2.0 + 1e-6).all()
        assert (frag_mass > ((6660.0 + 440.0) / 2 - 1) * UM3).all()
        np.testing.assert_approx_equal(nf[0] * frag_mass[0], total)
    # :264-340 distribution sweep (4 mm and 2 mm drops)
    rain = np.asarray([(4 / 3) * np.pi * (0.2e-2 / 2) ** 3, (4 / 3) * np.pi * (0.4e-2 / 2) ** 3])
    for make in (lambda: C.Exponential(scale=1e6 * UM3), lambda: C.Gaussian(mu=2e6 * UM3,
                                                                            sigma=1e6 * UM3),
                 C.SLAMS, C.Straub2010Nf, C.LowList1982Nf):
        for rn in np.linspace(1e-6, 1 - 1e-6, 25):
            nf, frag_mass = _run_fragmentation(kit, make, rain, u01=rn, vmin=1 * UM3)
            assert (nf > 0.99).all() and (frag_mass > 0).all(), (make().__class__.__name__, rn)
            np.testing.assert_approx_equal(nf[0] * frag_mass[0], np.sum(rain) * RHO_W)
    # :342-400 nf and fragment mass of ConstantMass / AlwaysN
    for make in (lambda: C.ConstantMass(c=4 * UM3), lambda: C.AlwaysN(n=250)):
        sut = make()
        water_mass = np.asarray([400.0 * UM3, 600.0 * UM3])
        particulator, flag = _two_drop_setup(kit, sut, None, water_mass=water_mass)
        nf = particulator.PairwiseStorage.from_ndarray(np.zeros(1))
        frag_mass = particulator.PairwiseStorage.from_ndarray(np.zeros(1))
        rand = particulator.PairwiseStorage.from_ndarray(np.asarray([0.5]))
        sut(nf, frag_mass, rand, flag)
        np.testing.assert_array_equal(nf.to_ndarray(), [250])
        np.testing.assert_array_almost_equal(frag_mass.to_ndarray(), [np.sum(water_mass) / 250])


def check_reference_efficiency_and_kernel_tests(kit):
    """test_efficiencies.py:21-56 (values in [0, 1]) and test_kernels.py:32-89 (SimpleGeometric
    zero for C = 0 and for equal sizes, positive otherwise)"""
    C = _fragmentation_classes()
    volume = np.asarray([440.0 * UM3, 6660.0 * UM3])
    for sut in (C.Berry1967(), C.ConstEc(Ec=0.5), C.SpecifiedEff(A=0.8, B=0.6), C.Straub2010Ec(),
                C.LowList1982Ec(), C.ConstEb(Eb=0.3)):
        particulator, flag = _two_drop_setup(kit, sut, volume)
        eff = particulator.PairwiseStorage.from_ndarray(np.asarray([-1.0]))
        sut(eff, flag)
        values = eff.to_ndarray()
        assert np.min(values) >= 0 and np.max(values) <= 1, sut.__class__.__name__
    sut = C.Linear(a=2.0, b=3.0)  # no reference run exists (stub there): analytic answer
    particulator, flag = _two_drop_setup(kit, sut, np.asarray([44.0, 666.0]))
    output = particulator.PairwiseStorage.from_ndarray(np.zeros(1))
    sut(output, flag)
    np.testing.assert_allclose(output.to_ndarray(), [2.0 + 3.0 * 710.0], rtol=1e-14)
    for c_value, vol, positive in ((0.0, [44.0, 666.0], False), (1.0, [44.0, 666.0], True),
                                   (1.0, [1.0, 2.0], True), (1.0, [1.0, 1.0], False)):
        sut = C.SimpleGeometric(C=c_value)
        particulator, flag = _two_drop_setup(kit, sut, np.asarray(vol))
        output = particulator.PairwiseStorage.from_ndarray(np.zeros(1))
        sut(output, is_first_in_pair=flag)
        if positive:
            assert (output.to_ndarray() > 0).all()
        else:
            np.testing.assert_array_equal(output.to_ndarray(), [0.0])


ALL_CHECKS = (check_reference_fragmentation_tests, check_reference_efficiency_and_kernel_tests,
              check_scale_prob_known_answers, check_adaptivity_paper_diagram,
              check_gamma_formula_grid, check_same_multiplicity_split,
              check_single_breakup_known_answers)
