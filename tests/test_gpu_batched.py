"""The batched march (csrc/sk_batch_kernels.inc) against the one-panel-per-step march of the same kernels
(batch_panels = 1): grouping panels into launch triples must change neither the panel schedule nor -- beyond the rounding
of the slots' tiled end points -- the results.  Covers the paths the BASELINE configs do not reach on their own: re-runs
(a later slot of a run rejected or retiring), the output fix (a guessed run cut short by a retirement at its first panel),
sibling pairs with the left child rejected, NaN tail fits, the three criteria."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sched(cfg):
    return [(r["a"], r["b"], bool(r["accepted"]), int(r["n_active"])) for r in cfg.last_trace()]


CASES = [
    # name, sdf kind, params, config kwargs, lags
    ("c5_like", "matern", (2.14, 0.97, 0.89), dict(tol=1e-12, quadspec=(256, 16)), lambda rng: np.linspace(0.0, 5.1, 3001)),
    ("c2_like", "matern", (1.0, 1.0, 1.5), dict(quadspec=(256, 16)), lambda rng: 10.0 ** np.linspace(-6, 0, 5000)),
    ("default_rule", "matern", (1.0, 1.0, 1.5), dict(), lambda rng: 10.0 ** np.linspace(-5, 0, 4000)),
    ("slow_decay_tails", "matern", (1.0, 0.3, 0.6), dict(tol=1e-6, quadspec=(128, 8), convergence_criteria="tails"),
     lambda rng: np.sort(rng.uniform(0.0, 3.0, 700))),
    ("slow_decay_both", "matern", (1.0, 0.3, 0.6), dict(tol=1e-7, quadspec=(128, 8)), lambda rng: np.sort(rng.uniform(0.0, 3.0, 900))),
    ("panel_criterion", "matern", (2.0, 0.5, 1.2), dict(tol=1e-8, quadspec=(256, 8), convergence_criteria="panel"),
     lambda rng: np.sort(rng.uniform(0.0, 2.0, 600))),
    ("narrow_bisect", "matern", (1.0, 1e-3, 1.5), dict(quadspec=(64, 4)), lambda rng: 10.0 ** np.linspace(-4, 0, 300)),
    ("narrow_bisect_wide_rule", "matern", (1.0, 3e-3, 1.0), dict(tol=1e-9, quadspec=(128, 16)),
     lambda rng: np.r_[0.0, 10.0 ** rng.uniform(-3, 0.5, 500)]),
    ("exponential_nan_tail", "exponential", (1.0,), dict(quadspec=(256, 16)), lambda rng: np.linspace(0.0, 5.1, 800)),
    ("gaussian", "gaussian", (2.0,), dict(tol=1e-10, quadspec=(128, 8)), lambda rng: np.linspace(0.0, 1.5, 500)),
    ("derivative", "matern", (2.14, 0.97, 1.39), dict(derivative=True, tol=1e-8, quadspec=(256, 16)),
     lambda rng: np.sort(rng.uniform(0.0, 4.0, 1200))),
    ("singular", "matern", (2.14, 0.97, 0.89), dict(alpha=0.4, quadspec=(256, 16)), lambda rng: np.linspace(0.0, 1.1, 900)),
    ("user_tail", "matern", (1.0, 1.0, 0.75), dict(tail=-2.5, tol=1e-7, quadspec=(256, 8)), lambda rng: np.sort(rng.uniform(0, 2, 500))),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_batched_march_equals_the_one_panel_march(sk, oracle, case):
    name, kind, prm, kw, mk = case
    rs = mk(np.random.default_rng(3))
    base = sk.AdaptiveKernelConfig(getattr(sk.SDF, kind)(*prm), batch_panels=1, **kw)
    k0 = sk.compute_k0(base)
    k1, e1 = sk.kernel_values(base, rs, k0=k0)
    s1, st1 = _sched(base), base.last_stats()
    assert st1["n_steps"] == st1["n_panels"] and st1["n_rerun"] == 0 and st1["n_fix"] == 0
    seen = dict(rerun=0, fix=0, saved=0)
    for K in (2, 4, 8):
        cfg = sk.AdaptiveKernelConfig(getattr(sk.SDF, kind)(*prm), batch_panels=K, **kw)
        k, e = sk.kernel_values(cfg, rs, k0=k0)
        st = cfg.last_stats()
        assert _sched(cfg) == s1, (name, K)                                   # the reference's schedule, panel by panel
        assert st["n_panels"] == st1["n_panels"] and st["n_outer"] == st1["n_outer"] and st["sum_active"] == st1["sum_active"]
        assert st["n_steps"] <= st1["n_panels"]
        a0 = abs(k0)
        assert np.nanmax(np.abs(k - k1)) <= 2e-13 * a0, (name, K, np.nanmax(np.abs(k - k1)) / a0)
        assert np.array_equal(np.isnan(e), np.isnan(e1)) and np.nanmax(np.abs(e - e1)) <= 2e-13 * a0 + 1e-6 * np.nanmax(np.abs(e1))
        seen["rerun"] += st["n_rerun"]
        seen["fix"] += st["n_fix"]
        seen["saved"] += st1["n_panels"] - st["n_steps"]
    print(name, seen, "panels", st1["n_panels"])
    test_batched_march_equals_the_one_panel_march.seen = getattr(test_batched_march_equals_the_one_panel_march, "seen", {})
    test_batched_march_equals_the_one_panel_march.seen[name] = seen


def test_batching_and_the_output_fix_occurred(sk):
    # runs after the parametrised cases (file order)
    seen = getattr(test_batched_march_equals_the_one_panel_march, "seen", {})
    assert len(seen) == len(CASES)
    assert sum(s["saved"] for s in seen.values()) > 0
    assert sum(s["fix"] for s in seen.values()) > 0, seen


@pytest.mark.parametrize("case", [c for c in CASES if c[0] in ("c5_like", "slow_decay_both", "derivative", "user_tail", "gaussian",
                                                              "panel_criterion", "exponential_nan_tail")],
                         ids=lambda c: c[0])
def test_wrong_guesses_are_harmless(sk, case):
    """SK_BATCH_GUESS=always: every run is extended to the cap whatever the estimate says, so most guesses are wrong --
    a retirement in the middle of a run re-runs it capped (n_rerun), one at its first panel leaves an output fix
    (n_fix).  Schedule and results must still be the one-panel march's."""
    import os
    name, kind, prm, kw, mk = case
    rs = mk(np.random.default_rng(3))
    base = sk.AdaptiveKernelConfig(getattr(sk.SDF, kind)(*prm), batch_panels=1, **kw)
    k0 = sk.compute_k0(base)
    k1, e1 = sk.kernel_values(base, rs, k0=k0)
    s1, st1 = _sched(base), base.last_stats()
    os.environ["SK_BATCH_GUESS"] = "always"
    try:
        tot = dict(rerun=0, fix=0)
        for K in (2, 4, 8):
            cfg = sk.AdaptiveKernelConfig(getattr(sk.SDF, kind)(*prm), batch_panels=K, **kw)
            k, e = sk.kernel_values(cfg, rs, k0=k0)
            st = cfg.last_stats()
            assert _sched(cfg) == s1, (name, K)
            assert st["n_panels"] == st1["n_panels"] and st["sum_active"] == st1["sum_active"]
            assert np.nanmax(np.abs(k - k1)) <= 2e-13 * abs(k0), (name, K)
            assert np.array_equal(np.isnan(e), np.isnan(e1)) and np.nanmax(np.abs(e - e1)) <= 2e-13 * abs(k0) + 1e-6 * np.nanmax(np.abs(e1))
            tot["rerun"] += st["n_rerun"]
            tot["fix"] += st["n_fix"]
    finally:
        del os.environ["SK_BATCH_GUESS"]
    print(name, tot, "panels", st1["n_panels"])
    test_wrong_guesses_are_harmless.tot = getattr(test_wrong_guesses_are_harmless, "tot", dict(rerun=0, fix=0))
    for q in tot:
        test_wrong_guesses_are_harmless.tot[q] += tot[q]


def test_reruns_and_fixes_were_driven(sk):
    tot = getattr(test_wrong_guesses_are_harmless, "tot", None)
    assert tot and tot["rerun"] > 0 and tot["fix"] > 0, tot


def test_batch_panels_argument(sk):
    with pytest.raises(sk.SpectralKernelsError):
        sk.AdaptiveKernelConfig(sk.SDF.matern(1, 1, 1.5), batch_panels=3)
    # hosts steps (callback SDFs, dim 2, logw) ignore the setting
    cb = sk.AdaptiveKernelConfig(lambda w: (1.0 + w * w) ** -2.0, quadspec=(64, 4), sdf="callback", batch_panels=4)
    ks, _ = sk.kernel_values(cb, np.array([0.1, 0.5, 1.0]), k0=np.pi / 2)
    rs = np.array([0.1, 0.5, 1.0])
    assert np.abs(ks - np.pi / 2 * (1 + 2 * np.pi * rs) * np.exp(-2 * np.pi * rs)).max() <= 1e-7


def test_repeated_calls_stream_from_the_first_step(sk, oracle):
    """A config whose last march took several steps skips the k_advance after step 0 and streams from the start (the
    spread launches report progress, input flags and the end themselves).  Same results, schedule and statistics as the
    first call; unsorted or invalid input is still found out (one step later) and handled."""
    parms = (2.14, 0.97, 0.89)
    rs = np.linspace(0.0, 5.1, 4000)
    cfg = sk.AdaptiveKernelConfig(sk.SDF.matern(*parms), tol=1e-10, quadspec=(256, 16))
    k0 = float(oracle.matern_cov(0.0, parms))
    d_rs, d_k, d_e = sk.DeviceArray.from_host(rs), sk.DeviceArray(rs.size), sk.DeviceArray(rs.size)
    sk.kernel_values_device(cfg, d_rs.ptr, rs.size, k0, d_k.ptr, d_e.ptr)
    k_first, st_first, tr_first = d_k.to_host(), cfg.last_stats(), _sched(cfg)
    assert st_first["n_steps"] >= 2 and st_first["n_syncs"] == 2          # k_advance after step 0, then streaming
    for _ in range(3):
        sk.kernel_values_device(cfg, d_rs.ptr, rs.size, k0, d_k.ptr, d_e.ptr)
        st = cfg.last_stats()
        assert st["n_syncs"] == 1 and st["n_steps_launched"] <= st["n_steps"] + 2  # streamed from step 0
        assert np.array_equal(d_k.to_host(), k_first) and _sched(cfg) == tr_first
        assert all(st[q] == st_first[q] for q in ("n_panels", "n_outer", "n_steps", "sum_active"))
    assert np.abs(k_first - oracle.matern_cov(rs, parms)).max() / k0 <= 10 * 1e-10
    # device-resident input of unknown order on the streaming path: permuted with duplicates, then an invalid lag
    rng = np.random.default_rng(1)
    xs = np.r_[rs[rng.permutation(rs.size)], rs[:100]]
    d_x, d_k2, d_e2 = sk.DeviceArray.from_host(xs), sk.DeviceArray(xs.size), sk.DeviceArray(xs.size)
    sk.kernel_values_device(cfg, d_x.ptr, xs.size, k0, d_k2.ptr, d_e2.ptr)
    assert cfg.last_stats()["n_unique"] == rs.size
    assert np.abs(d_k2.to_host() - oracle.matern_cov(xs, parms)).max() / k0 <= 10 * 1e-10
    bad = xs.copy()
    bad[17] = -1.0
    d_b = sk.DeviceArray.from_host(bad)
    with pytest.raises(sk.SpectralKernelsError):
        sk.kernel_values_device(cfg, d_b.ptr, bad.size, k0, d_k2.ptr, d_e2.ptr)
    sk.kernel_values_device(cfg, d_rs.ptr, rs.size, k0, d_k.ptr, d_e.ptr)   # and the config is fine afterwards
    assert np.array_equal(d_k.to_host(), k_first)


def test_launch_floor_probe_is_safe_on_any_config_state(sk):
    """sk_launch_floor launches the march's kernels on a "done" state; the interpolation kernels read the lag
    array ahead of the state, so the probe brings lags of its own.  Fresh config (no call yet), after a tiny call, after
    a call on other lags, batched and callback configs: no fault, sane numbers, and the config still works."""
    rs = 10.0 ** np.linspace(-4, 0, 50)
    for mk in (lambda: sk.AdaptiveKernelConfig(sk.SDF.matern(1, 1, 1.5), quadspec=(256, 16)),
               lambda: sk.AdaptiveKernelConfig(sk.SDF.matern(1, 1, 1.5)),
               lambda: sk.AdaptiveKernelConfig(lambda w: (1.0 + w * w) ** -2.0, quadspec=(64, 4), sdf="callback")):
        cfg = mk()
        f0 = cfg.launch_floor(reps=3)            # before any kernel_values call
        ks, _ = sk.kernel_values(cfg, rs[:3], k0=np.pi / 2)   # a tiny call: the last lag array is 3 doubles long
        f1 = cfg.launch_floor(reps=3)
        ks2, _ = sk.kernel_values(cfg, rs, k0=np.pi / 2)
        f2 = cfg.launch_floor(reps=3)
        for f in (f0, f1, f2):
            assert all(0.5 < v < 200.0 for v in f.values()), f
        assert np.abs(ks2 - np.pi / 2 * (1 + 2 * np.pi * rs) * np.exp(-2 * np.pi * rs)).max() <= 1e-7
        assert np.array_equal(ks, ks2[:3]) or np.abs(ks - ks2[:3]).max() <= 1e-8


def test_sibling_pair_with_a_jacobi_left_child_is_one_step(sk, oracle):
    """a singular SDF (p != 0) on the default rule bisects its first panel [0, b] repeatedly
    (src/quadrature.jl:268-270); the left child of every such split starts at 0 -- Gauss-Jacobi first sub-panel
    (:61-78), other source positions -- and used to be evaluated in a step of its own.  Now the pair (right child
    Legendre, left child Jacobi) is one step with two tile lists: same schedule, same values, fewer steps."""
    prm = (1.0, 1e-3, 1.5)  # a narrow SDF: the first panels are bisected down to its width
    rs = 10.0 ** np.linspace(-4, 0, 600)
    kw = dict(alpha=0.4, tol=1e-8, quadspec=(64, 4))
    one = sk.AdaptiveKernelConfig(sk.SDF.matern(*prm), batch_panels=1, **kw)
    two = sk.AdaptiveKernelConfig(sk.SDF.matern(*prm), batch_panels=2, **kw)
    k0 = sk.compute_k0(one)
    k1, e1 = sk.kernel_values(one, rs, k0=k0)
    k2, e2 = sk.kernel_values(two, rs, k0=k0)
    assert _sched(one) == _sched(two)
    tr = one.last_trace()
    splits_at_zero = sum(1 for r in tr if r["a"] == 0.0 and not r["accepted"])
    assert splits_at_zero >= 3, tr[:6]
    s1, s2 = one.last_stats(), two.last_stats()
    assert s1["n_steps"] == s1["n_panels"] and s2["n_panels"] == s1["n_panels"]
    # every split of a panel that starts at 0 leaves such a pair behind: at least that many steps saved
    assert s2["n_steps"] <= s1["n_panels"] - splits_at_zero, (s1, s2, splits_at_zero)
    assert np.nanmax(np.abs(k2 - k1)) <= 2e-13 * abs(k0) and np.array_equal(np.isnan(e1), np.isnan(e2))
    # and against the oracle (direct sums, same rule): same schedule, values to a tenth of the tolerance
    ocfg = oracle.AdaptiveKernelConfig(oracle.SDF.matern(*prm), alpha=0.4, tol=1e-8, quadspec=(64, 4))
    ok, _ = oracle.kernel_values(ocfg, rs, k0=k0)
    assert np.abs(k2 - ok).max() <= 1e-9 * abs(k0)


def _full_width(kw):
    m, k = kw.get("quadspec", (4096, 16))
    return 2 if m * k > 40000 else 4  # sk_config_create: K n1 <= 2^17


AUTO_INPUTS = [
    # every panel retires lags (the C3 pattern): one-panel steps only -- the config switches to the K = 1 kernels
    ("retiring", lambda: 10.0 ** np.linspace(-8, -2, 20000)),
    # runs of outer panels (the C5 pattern): batches at the full width
    ("runs", lambda: np.linspace(0.0, 5.1, 20001)),
    ("retiring_short", lambda: 10.0 ** np.linspace(-7, -3, 5000)),
]


@pytest.mark.parametrize("kind,prm,kw", [
    ("matern", (2.14, 0.97, 0.89), dict(alpha=0.4, quadspec=(256, 16))),
    ("matern", (2.14, 0.97, 0.89), dict(tol=1e-12, quadspec=(256, 16))),
    ("matern", (1.0, 1.0, 1.5), dict()),
])
def test_automatic_batch_width_gives_results_that_depend_on_the_input_only(sk, kind, prm, kw):
    """VERDICT r3 #6 / ADVICE r3: the reference's kernel_values is a pure function of its input (src/adaptive.jl:95-108).
    With batch_panels = 0 a config may run the one-panel kernels after a march that never batched; whatever it ran
    before, every call must return, bit for bit, what the full batch width returns on that input -- including the first
    call, calls after an input of the other kind, and many repeated calls (the old rule re-probed every 64th)."""
    K = _full_width(kw)
    auto = sk.AdaptiveKernelConfig(getattr(sk.SDF, kind)(*prm), **kw)
    full = sk.AdaptiveKernelConfig(getattr(sk.SDF, kind)(*prm), batch_panels=K, **kw)
    k0 = sk.compute_k0(auto)
    ref, sched, steps = {}, {}, {}
    for name, mk in AUTO_INPUTS:
        ref[name] = sk.kernel_values(full, mk(), k0=k0)
        sched[name] = _sched(full)
        steps[name] = full.last_stats()["n_steps"]
    order = ["retiring", "retiring", "runs", "retiring", "retiring_short", "runs", "runs", "retiring"] + ["retiring"] * 70
    seen_narrow = False
    for call, name in enumerate(order):
        k, e = sk.kernel_values(auto, dict(AUTO_INPUTS)[name](), k0=k0)
        assert _sched(auto) == sched[name], (call, name)
        assert np.array_equal(k, ref[name][0]), (call, name, np.nanmax(np.abs(k - ref[name][0])) / abs(k0))
        assert np.array_equal(e, ref[name][1], equal_nan=True), (call, name)
        st = auto.last_stats()
        assert st["n_steps"] == steps[name], (call, name)
        seen_narrow |= st["n_steps"] == st["n_panels"] and st["n_panels"] >= 2
    if "alpha" in kw:  # the singular config: every panel of `retiring` retires lags -- marches the one-panel kernels can take
        assert seen_narrow




HYBRID_CASES = [
    # the C5 pattern: a few full-width steps of outer runs, then a long peel of one-panel steps
    ("c5_peel", (2.14, 0.97, 0.89), dict(tol=1e-12, quadspec=(256, 16)), lambda: np.linspace(0.0, 5.1, 20001)),
    # default rule (K = 2), log-spaced lags: a short march with a retiring tail
    ("default_rule", (2.05, 2.6, 0.67), dict(tol=1e-8), lambda: 10.0 ** np.linspace(-5, 1, 30000)),
    # many small panels, bisections in between (slow decay)
    ("small_rule_long", (2.99, 2.61, 0.68), dict(tol=1e-10, quadspec=(128, 8)), lambda: 10.0 ** np.linspace(-5, 1, 15000)),
    # singular SDF: Jacobi first panel and its bisection (mixed sibling pairs), then retiring panels
    ("singular", (1.0, 1e-3, 1.5), dict(alpha=0.4, quadspec=(64, 4)), lambda: 10.0 ** np.linspace(-6, 0, 8000)),
]


@pytest.mark.parametrize("name,prm,kw,lags", HYBRID_CASES, ids=[c[0] for c in HYBRID_CASES])
def test_one_panel_launches_inside_a_full_width_march_give_the_full_widths_bits(sk, name, prm, kw, lags, monkeypatch):
    """Automatic batch width, second half (sk_api.cpp, run_core): while the march retires lags every step is one fresh
    outer panel (src/adaptive.jl:149-199) and the host launches the NEXT step with the lighter one-panel instantiations
    (HostStatus::peel_hint); a step they cannot take -- the full width might batch there -- is skipped on the device and
    launched again at the full width (HostStatus::need_wide).  Whatever mixture of launches a call saw, schedule, values
    and error estimates are the explicit full width's, bit for bit.  SK_HYBRID_FORCE launches EVERY step with the
    one-panel kernels first: the skip / relaunch path at every step that is not a peel step."""
    K = _full_width(kw)
    auto = sk.AdaptiveKernelConfig(sk.SDF.matern(*prm), **kw)
    full = sk.AdaptiveKernelConfig(sk.SDF.matern(*prm), batch_panels=K, **kw)
    k0 = sk.compute_k0(auto)
    xs = lags()
    kf, ef = sk.kernel_values(full, xs, k0=k0)
    sched_f, st_f = _sched(full), full.last_stats()
    assert sk.step_kinds(full) == dict(one_panel_launches=0, skipped=0)
    seen = {}
    for mode in ("plain", "plain", "forced", "plain", "off"):
        if mode == "forced":
            monkeypatch.setenv("SK_HYBRID_FORCE", "1")
        else:
            monkeypatch.delenv("SK_HYBRID_FORCE", raising=False)
        cfg = sk.AdaptiveKernelConfig(sk.SDF.matern(*prm), batch_panels=K, **kw) if mode == "off" else auto
        k, e = sk.kernel_values(cfg, xs, k0=k0)
        st, kinds = cfg.last_stats(), sk.step_kinds(cfg)
        assert _sched(cfg) == sched_f, (name, mode)
        assert np.array_equal(k, kf) and np.array_equal(e, ef, equal_nan=True), (name, mode, kinds)
        assert st["n_steps"] == st_f["n_steps"] and st["n_panels"] == st_f["n_panels"], (name, mode)
        # launched in vain: the step queued behind the last one, the one whose launch reported "done", every skipped one
        assert st["n_steps_launched"] <= st["n_steps"] + st["n_rerun"] + 2 + kinds["skipped"], (name, mode, st, kinds)
        seen[mode] = kinds
    if name == "c5_peel":
        assert seen["plain"]["one_panel_launches"] >= 20 and seen["plain"]["skipped"] <= 2, seen
    assert seen["forced"]["one_panel_launches"] >= seen["plain"]["one_panel_launches"], seen
    if st_f["n_steps"] < st_f["n_panels"]:  # the march batches somewhere: forced one-panel launches must have been skipped there
        assert seen["forced"]["skipped"] >= 1, seen
