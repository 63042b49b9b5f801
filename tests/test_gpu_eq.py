"""Parametric EQ on the MI355X through the C ABI: the reference's own known-answer tests
(AirwaveTests/ParametricEqualizerProcessorTests.swift, re-expressed over aw_eq_*), batch parity with the
oracle's sequential Float64 recurrence, and size-independent properties at the cfg-4 batch size.

Tolerance: the reference's tests use 1e-6 / 1e-5 absolute on unit-scale signals; against the oracle the
bound is 1 ulp of the Float32 output (the kernel evaluates the same Float64 recurrence chunk-parallel)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
PK, LSC, HSC = 0, 1, 2
ULP = 6e-8   # 2^-24: one unit in the last place at |y| < 1


@pytest.fixture(scope="module")
def aw():
    import airwave_amd
    return airwave_amd


def flt(aw, t, f, g, q, enabled=True):
    return aw.EqualizerFilter(1, None, enabled, t, f, g, q)


def run(p, n, lv=1.0, rv=1.0):
    return p.process(np.full(n, lv, np.float32), np.full(n, rv, np.float32))


def db(x):
    return np.float32(10.0 ** (x / 20.0))


# ---- the reference's KATs ----------------------------------------------------------------------------
def test_unity_and_preamp(aw):
    unity = aw.ParametricEqualizerProcessor.prepare(None, 48000.0)                 # :87-108
    pre = aw.ParametricEqualizerProcessor.prepare(aw.EqualizerDefinition(6.0), 48000.0)
    l, r = np.array([0.25, -0.5, 1], np.float32), np.array([-0.75, 0.5, 0.125], np.float32)
    ul, ur = unity.process(l, r)
    assert np.array_equal(ul, l) and np.array_equal(ur, r)
    pl, pr = pre.process(l, r)
    assert abs(pl[0] - l[0] * db(6)) < 1e-6 and abs(pr[2] - r[2] * db(6)) < 1e-6
    assert unity.filterCount == 0 and abs(pre.preampLinear - 10 ** 0.3) < 1e-15


def test_known_impulse_response(aw):
    d = aw.EqualizerDefinition(0.0, [flt(aw, PK, 1000, 6, 0.707), flt(aw, PK, 3000, -3, 1.1)])   # :110-133
    exp = np.array([1.007962105198731, 0.026656172367575, 0.046848317472827, 0.062845911221200, 0.072328817552935,
                    0.074696369241889])
    st = aw.ParametricEqualizerProcessor.prepare(d, 48000.0)
    l, r = st.process(np.array([1, 0, 0, 0, 0, 0], np.float32), np.zeros(6, np.float32))
    assert np.max(np.abs(l - exp)) < 1e-6 and np.all(r == 0)
    # same answer on the chunk-parallel kernel (>= 32 frames)
    st = aw.ParametricEqualizerProcessor.prepare(d, 48000.0)
    x = np.zeros(64, np.float32); x[0] = 1
    l, r = st.process(x, np.zeros(64, np.float32))
    assert np.max(np.abs(l[:6] - exp)) < 1e-6 and np.all(r == 0)


def test_disabled_filters_and_subnormal_flush(aw):
    st = aw.ParametricEqualizerProcessor.prepare(aw.EqualizerDefinition(0.0, [flt(aw, PK, 1000, 12, 0.7, enabled=False)]), 48000.0)
    l, r = st.process(np.array([1, 0], np.float32), np.array([1, 0], np.float32))              # :135-152
    assert l.tolist() == [1, 0] and r.tolist() == [1, 0]
    act = aw.ParametricEqualizerProcessor.prepare(aw.EqualizerDefinition(0.0, [flt(aw, PK, 1000, 6, 0.707)]), 48000.0)
    sl, _ = act.process(np.array([1.4e-45, 0], np.float32), np.zeros(2, np.float32))
    assert sl[0] != 0 and sl[1] == 0


def test_in_place_processing_preserves_canaries(aw):
    d = aw.EqualizerDefinition(0.0, [flt(aw, HSC, 6000, -5, 0.8)])                               # :154-190
    st = aw.ParametricEqualizerProcessor.prepare(d, 48000.0)
    size, canary = 4096, np.float32(12345)
    buf = np.full((size + 2, 2), canary, np.float32)
    i = np.arange(size)
    buf[1:-1, 0] = (i % 17).astype(np.float32) / 17
    buf[1:-1, 1] = -(i % 13).astype(np.float32) / 13
    ctx = st.ctx
    dptr = ctx.alloc(buf.nbytes)
    ctx.h2d(dptr, buf)
    st.process_device(dptr + 8, dptr + 8, size)                  # in place, one frame into the buffer
    out = np.empty_like(buf)
    ctx.d2h(out, dptr)
    ctx.free(dptr)
    assert np.all(out[0] == canary) and np.all(out[-1] == canary) and np.isfinite(out).all()
    assert not np.array_equal(out[1:-1], buf[1:-1])


def test_preparation_rejections(aw):
    P = aw.ParametricEqualizerProcessor                                                         # :192-212
    with pytest.raises(aw.ParametricEqualizerPreparationError) as e:
        P.prepare(None, 0.0)
    assert e.value.kind == "invalidSampleRate"
    with pytest.raises(aw.ParametricEqualizerPreparationError) as e:
        P.prepare(aw.EqualizerDefinition(0.0, [flt(aw, PK, 24000, 1, 1)]), 48000.0)
    assert e.value.kind == "invalidFilter" and e.value.filter_error == "invalidFrequency"
    with pytest.raises(aw.ParametricEqualizerPreparationError) as e:
        P.prepare(aw.EqualizerDefinition(0.0, [flt(aw, PK, 1000, 1, 0)]), 48000.0)
    assert e.value.filter_error == "invalidQ"
    with pytest.raises(aw.ParametricEqualizerPreparationError) as e:
        P.prepare(aw.EqualizerDefinition(0.0, [flt(aw, PK, 500 + i, 1, 1) for i in range(65)]), 48000.0)
    assert e.value.kind == "tooManyFilters"
    with pytest.raises(aw.ParametricEqualizerPreparationError):
        P(48000.0, maxFramesPerCallback=4097)                                                   # :148-150
    with pytest.raises(aw.ParametricEqualizerPreparationError):
        P(float("inf"))
    p = P(48000.0, maxFramesPerCallback=256)
    with pytest.raises(aw.AirwaveError):                                                        # precondition :261
        run(p, 257)


@pytest.mark.parametrize("fs", [44100.0, 48000.0, 96000.0])
def test_crossfade_ramp_across_callbacks(aw, fs):
    p = aw.ParametricEqualizerProcessor(fs, 4096)                                               # :214-232
    g = db(6)
    p.setTarget(aw.EqualizerDefinition(6.0))
    length = max(1, int(round(fs * 0.020)))
    assert p.transitionLength == length
    first = max(1, length // 2)
    a = run(p, first)
    assert p.isTransitioning
    b = run(p, length - first)
    assert not p.isTransitioning
    assert abs(a[0][0] - (1 + (g - 1) / np.float32(length))) < 1e-5
    assert abs(b[0][-1] - g) < 1e-5 and abs(b[1][-1] - g) < 1e-5


def test_transitions_to_and_from_unity(aw):
    p = aw.ParametricEqualizerProcessor(48000.0)                                                # :234-247
    p.setTarget(aw.EqualizerDefinition(6.0)); run(p, 960)
    p.setTarget(None)
    res = run(p, 960)
    g = db(6)
    assert abs(res[0][0] - (g - (g - 1) / np.float32(960))) < 1e-5 and abs(res[0][-1] - 1) < 1e-5


def test_rapid_publication_queues_newest(aw):
    p = aw.ParametricEqualizerProcessor(48000.0)                                                # :249-266
    p.setTarget(aw.EqualizerDefinition(6.0)); run(p, 480)
    p.setTarget(aw.EqualizerDefinition(-6.0))
    assert abs(run(p, 480)[0][-1] - db(6)) < 1e-5
    assert abs(run(p, 960)[0][-1] - db(-6)) < 1e-5


def test_render_callback_keeps_prior_target_when_publication_lock_is_contended(aw):
    """ParametricEqualizerProcessorTests.swift:285-302: a control thread sits inside the publication lock; the render call
    must not wait for it and keeps its prior (unity) target although a new one has been published."""
    import threading
    p = aw.ParametricEqualizerProcessor(48000.0)
    p.setTarget(aw.EqualizerDefinition(0.0, [flt(aw, 0, 1000.0, 6.0, 0.707)]))      # published, not yet observed
    entered, release = threading.Event(), threading.Event()

    def hold():
        p.withPublicationLockForTesting(lambda: (entered.set(), release.wait(5.0)))

    th = threading.Thread(target=hold)
    th.start()
    assert entered.wait(5.0)
    try:
        left, right = p.process(np.full(128, 1.0, np.float32), np.full(128, 2.0, np.float32))
    finally:
        release.set()
        th.join()
    assert np.array_equal(left, np.full(128, 1.0, np.float32)) and np.array_equal(right, np.full(128, 2.0, np.float32))
    assert not p.isTransitioning
    # lock released: the next call observes the published target and starts the fade
    left2, _ = p.process(np.full(128, 1.0, np.float32), np.full(128, 2.0, np.float32))
    assert p.isTransitioning and not np.array_equal(left2, left)


def test_control_thread_calls_run_beside_the_render_loop(aw):
    """setTarget / reset / drainRetiredStates from a second thread while the first keeps calling process (the reference's
    threading model, ParametricEqualizerProcessor.swift:120-131): no deadlock, finite output, the last target wins."""
    import threading
    p = aw.ParametricEqualizerProcessor(48000.0)
    stop = threading.Event()
    errors = []

    def control():
        try:
            i = 0
            while not stop.is_set():
                p.setTarget(aw.EqualizerDefinition(-1.0, [flt(aw, i % 3, 200.0 + 50.0 * (i % 40), 3.0, 1.0)]) if i % 5 else None)
                p.drainRetiredStates()
                if i % 7 == 0:
                    p.reset()
                i += 1
        except Exception as e:                                   # noqa: BLE001
            errors.append(e)

    th = threading.Thread(target=control)
    th.start()
    try:
        x = np.linspace(-0.5, 0.5, 480, dtype=np.float32)
        for _ in range(300):
            l, r = p.process(x, x)
            assert np.isfinite(l).all() and np.isfinite(r).all()
    finally:
        stop.set()
        th.join()
    assert not errors
    p.setTarget(None)
    for _ in range(6):
        p.drainRetiredStates()
        l, r = p.process(x, x)
    assert np.array_equal(l, x) and np.array_equal(r, x)             # settled on unity


def test_retirement_pressure(aw):
    p = aw.ParametricEqualizerProcessor(48000.0)                                                # :268-293
    g1, g2, g3 = db(6), db(-6), db(12)
    p.setTarget(aw.EqualizerDefinition(6.0)); run(p, 960)
    p.setTarget(aw.EqualizerDefinition(-6.0)); second = run(p, 960)
    assert abs(second[0][-1] - g2) < 1e-5
    p.setTarget(aw.EqualizerDefinition(12.0)); held = run(p, 960)
    assert abs(held[0][-1] - g2) < 1e-5
    p.drainRetiredStates()
    assert abs(run(p, 960)[0][-1] - g3) < 1e-5
    assert abs(second[0][0] - (g1 + (g2 - g1) / np.float32(960))) < 1e-5


def test_reset_clears_histories_and_mono_duplication(aw):
    p = aw.ParametricEqualizerProcessor(48000.0)                                                # :318-329
    p.setTarget(aw.EqualizerDefinition(0.0, [flt(aw, PK, 1000, 6, 0.707)])); run(p, 960)
    p.reset(); p.setTarget(None); run(p, 960)
    a = run(p, 1, 0, 0)
    assert a[0].tolist() == [0] and a[1].tolist() == [0]
    l, r = p.process(np.array([0.5, 0.25], np.float32), None)                                   # inputRight == nil  :68
    assert np.array_equal(l, r) and l.tolist() == [0.5, 0.25]


def test_ten_filter_workload_stays_finite_across_callback_sizes(aw):
    """ParametricEqualizerProcessorTests.swift:317-358 as written there: ten seconds of 0.25 per callback size through the planar
    (render-callback shaped) entry of ONE prepared state.  Beyond the reference's assertions: the steady state of a constant input
    is the cascade's DC gain."""
    fl = [flt(aw, PK if i % 2 == 0 else HSC, 250 + i * 1000, (i % 3) - 1, 0.8) for i in range(10)]
    st = aw.ParametricEqualizerProcessor.prepare(aw.EqualizerDefinition(-3.0, fl), 48000.0)
    dc = 10 ** (-3 / 20) * np.prod([(c[0] + c[1] + c[2]) / (1 + c[3] + c[4]) for c in
                                    [aw.BiquadCoefficientBuilder.make(f.type, f.gainDB, f.frequencyHz, f.q, 48000.0) for f in fl]])
    for size in (128, 512, 1024):
        x = np.full(size, 0.25, np.float32)
        processed = 0
        while processed < 48000 * 10:
            left, right = st.process(x, x)
            processed += size
        assert np.isfinite(left).all() and np.isfinite(right).all()
        assert np.max(np.abs(left - 0.25 * dc)) < 1e-5 and np.max(np.abs(right - 0.25 * dc)) < 1e-5


def test_reference_fixture_curve(aw, golden_dir):
    data = open(os.path.join(golden_dir, "eq", "CCA CRA ParametricEq.txt"), "rb").read()        # :359-394
    d = aw.EqualizerAPOParser.parse(data, "CCA CRA ParametricEq.txt")
    fs, n, skip = 48000.0, 48000, 24000
    st = aw.ParametricEqualizerProcessor.prepare(d, fs, n_streams=3)
    assert st.filterCount == 10
    freqs, exp = [20, 1000, 10000], [-5.3379478445, -0.9694887656, -4.2646888095]
    x = np.stack([np.sin(2 * np.pi * f * np.arange(n) / fs).astype(np.float32) for f in freqs])
    y = st.process_batch(np.stack([x, x], axis=2))
    for i in range(3):
        got = 20 * np.log10(np.sqrt(np.mean(y[i, skip:, 0].astype(np.float64) ** 2)) / np.sqrt(np.mean(x[i, skip:].astype(np.float64) ** 2)))
        assert abs(got - exp[i]) < 0.03 and np.isfinite(y[i]).all() and np.array_equal(y[i, :, 0], y[i, :, 1])


def test_runtime_effect_policy(aw):
    fx = aw.EqualizerRuntimeEffect()                                                            # EqualizerRuntimeEffect.swift
    l, r = fx.process(np.array([1, 2], np.float32), None)                                       # :61-70 passthrough before prepare
    assert l.tolist() == [1, 2] and r.tolist() == [1, 2]
    with pytest.raises(aw.EqualizerAudioEffectError) as e:
        fx.setTarget(None)                                                                      # :37-39
    assert e.value.kind == "unavailable"
    with pytest.raises(aw.EqualizerAudioEffectError):
        fx.prepare(None, 0.0)
    fx.prepare(aw.EqualizerDefinition(6.0), 48000.0)
    assert abs(run(fx, 960)[0][-1] - db(6)) < 1e-5
    bad = aw.EqualizerDefinition(0.0, [aw.EqualizerFilter(7, None, True, PK, 30000.0, 1.0, 1.0)])
    with pytest.raises(aw.EqualizerAudioEffectError) as e:
        fx.setTarget(bad)                                                                       # :41-47 falls back to unity
    assert e.value.kind == "invalidFilter" and "Nyquist" in e.value.reason
    assert abs(run(fx, 960)[0][-1] - 1) < 1e-5


# ---- batch parity with the oracle --------------------------------------------------------------------
FILTERS = [(LSC, 105.0, -2.8, 0.7), (PK, 65.3, 1.0, 1.68), (PK, 1000.0, 6.0, 0.707), (HSC, 10000.0, -5.2, 0.7), (PK, 20.0, 3.0, 4.0)]


def odef(oracle, preamp, filters):
    return oracle.EqualizerDefinition(preamp, [oracle.EqualizerFilter(1, None, True, t, f, g, q) for t, f, g, q in filters])


def adef(aw, preamp, filters):
    return aw.EqualizerDefinition(preamp, [flt(aw, t, f, g, q) for t, f, g, q in filters])


@pytest.mark.parametrize("calls", [[1, 15, 16, 17], [31, 32, 33, 64], [4095, 4096, 4097], [8191, 8192, 8193], [10000, 3, 70000], [8192, 8192]])
def test_state_batch_matches_oracle_over_ragged_calls(aw, oracle, calls):
    S = 5
    st = aw.ParametricEqualizerState(adef(aw, -2.56, FILTERS), 48000.0, n_streams=S)
    ref = [oracle.eq_prepare(odef(oracle, -2.56, FILTERS), 48000.0) for _ in range(S)]
    rng = np.random.default_rng(sum(calls))
    for n in calls:
        x = rng.uniform(-0.5, 0.5, (S, n, 2)).astype(np.float32)
        y = st.process_batch(x)
        for s in range(S):
            el, er = ref[s].process(x[s, :, 0], x[s, :, 1])
            assert np.max(np.abs(y[s, :, 0] - el)) <= ULP and np.max(np.abs(y[s, :, 1] - er)) <= ULP
            if n < 32:
                assert np.array_equal(y[s, :, 0], el)            # sequential kernel = the recurrence itself
    st.reset()
    for r in ref:
        r.reset()
    x = rng.uniform(-0.5, 0.5, (S, 100, 2)).astype(np.float32)
    y = st.process_batch(x)
    assert np.max(np.abs(y[2, :, 0] - ref[2].process(x[2, :, 0], x[2, :, 1])[0])) <= ULP


def test_sixty_four_filters_and_other_rates(aw, oracle):
    many = [(i % 3, 100.0 + 300.0 * i, ((i % 5) - 2) * 1.5, 0.5 + 0.1 * i) for i in range(64)]
    rng = np.random.default_rng(5)
    for fs in (44100.0, 96000.0):
        x = rng.uniform(-1, 1, (2, 9000, 2)).astype(np.float32)
        y = aw.ParametricEqualizerState(adef(aw, -6.0, many), fs, n_streams=2).process_batch(x)
        for s in range(2):
            e = oracle.eq_prepare(odef(oracle, -6.0, many), fs).process(x[s, :, 0], x[s, :, 1])
            assert oracle.peak_rel_error(y[s, :, 0], e[0]) < 1e-6 and oracle.peak_rel_error(y[s, :, 1], e[1]) < 1e-6


def test_processor_batch_crossfade_matches_oracle(aw, oracle):
    """Targets published between batch calls: fades start at call boundaries, end mid-call, queue and retire
    exactly like the oracle's restatement of ParametricEqualizerProcessor."""
    S, fs = 3, 48000.0
    p = aw.ParametricEqualizerProcessor(fs, maxFramesPerCallback=0, n_streams=S)
    refs = [oracle.ParametricEqualizerProcessor(fs) for _ in range(S)]
    for r in refs:
        r.max_frames = 1 << 30
    rng = np.random.default_rng(9)
    script = [("t", (-2.56, FILTERS)), ("p", 5000), ("t", (3.0, FILTERS[:2])), ("p", 500), ("t", None), ("p", 300), ("p", 2000),
              ("d", None), ("t", (-1.0, FILTERS[2:])), ("p", 4096), ("r", None), ("p", 1000), ("d", None), ("p", 960), ("p", 17)]
    for op, arg in script:
        if op == "t":
            p.setTarget(None if arg is None else adef(aw, *arg))
            for r in refs:
                r.set_target(None if arg is None else odef(oracle, *arg))
        elif op == "d":
            p.drainRetiredStates()
            [r.drain_retired_states() for r in refs]
        elif op == "r":
            p.reset()
            [r.reset() for r in refs]
        else:
            x = rng.uniform(-0.5, 0.5, (S, arg, 2)).astype(np.float32)
            y = p.process_batch(x)
            for s in range(S):
                el, er = refs[s].process(x[s, :, 0], x[s, :, 1])
                assert np.max(np.abs(y[s, :, 0] - el)) <= 2 * ULP and np.max(np.abs(y[s, :, 1] - er)) <= 2 * ULP, (op, arg, s)


def test_full_batch_properties(aw, oracle, golden_dir):
    """cfg-4 shaped batch on one GPU (512 streams, 10-band fixture, 96 kHz): spot streams against the oracle,
    plus linearity and split-call invariance on the whole batch."""
    import torch
    d = aw.EqualizerAPOParser.parse(open(os.path.join(golden_dir, "eq", "CCA CRA ParametricEq.txt"), "rb").read(), "f.txt")
    S, F, fs = 512, 96000, 96000.0
    ctx = aw.default_context()
    x = torch.empty((S, F, 2), dtype=torch.float32, device="cuda")
    ctx.synth_fill(x.data_ptr(), S, F, 2, seed=77)
    torch.cuda.synchronize()
    st = aw.ParametricEqualizerState(d, fs, n_streams=S)
    y = torch.empty_like(x)
    st.process_device(x.data_ptr(), y.data_ptr(), F)
    ctx.synchronize()
    od = oracle.EqualizerDefinition(d.preampDB, [oracle.EqualizerFilter(f.sourceLine, f.sourceNumber, f.isEnabled, f.type, f.frequencyHz, f.gainDB, f.q) for f in d.filters])
    for s in (0, 255, 511):
        xs = x[s].cpu().numpy()
        el, er = oracle.eq_prepare(od, fs).process(xs[:, 0], xs[:, 1])
        ys = y[s].cpu().numpy()
        assert np.max(np.abs(ys[:, 0] - el)) <= ULP and np.max(np.abs(ys[:, 1] - er)) <= ULP
    # split-call invariance: two calls of ragged size continue the streams exactly
    st2 = aw.ParametricEqualizerState(d, fs, n_streams=S)
    cut = 41003
    a = x[:, :cut].contiguous(); b = x[:, cut:].contiguous()
    torch.cuda.synchronize()                       # torch's stream produced a and b; the context has its own stream
    st2.process_device(a.data_ptr(), a.data_ptr(), cut)
    st2.process_device(b.data_ptr(), b.data_ptr(), F - cut)
    ctx.synchronize()
    assert float((torch.cat([a, b], 1) - y).abs().max()) <= 2 * ULP
    # linearity: EQ(0.5 x) = 0.5 EQ(x) exactly up to rounding of the Float32 store
    st3 = aw.ParametricEqualizerState(d, fs, n_streams=S)
    h = (x * 0.5).contiguous()
    torch.cuda.synchronize()
    st3.process_device(h.data_ptr(), h.data_ptr(), F)
    ctx.synchronize()
    assert float((h - 0.5 * y).abs().max()) <= ULP


def test_low_frequency_cascade_found_by_the_fuzzer(aw, oracle, golden_dir):
    """64 random sections at 96 kHz, a dozen of them below 50 Hz (tools/fuzz_eq.py seed 12, script 2688): with the scan tables
    formed in double the carried state was 1e-7 off at the wave boundaries and the output 4.6 ulp of the peak off; the
    double-double tables (host/eq.cpp) bring it back under one."""
    import json
    case = json.load(open(os.path.join(golden_dir, "eq", "fuzz_seed12_script2688.json")))
    fl = [tuple(f) for f in case["filters"]]
    st = aw.ParametricEqualizerState(adef(aw, case["preamp_db"], fl), 96000.0, n_streams=2)
    ref = [oracle.eq_prepare(odef(oracle, case["preamp_db"], fl), 96000.0) for _ in range(2)]
    rng = np.random.default_rng(2688)
    for n in (25229, 22327):
        x = rng.uniform(-0.5, 0.5, (2, n, 2)).astype(np.float32)
        y = st.process_batch(x)
        for s in range(2):
            el, er = ref[s].process(x[s, :, 0], x[s, :, 1])
            scale = max(1.0, float(np.max(np.abs(el))), float(np.max(np.abs(er))))
            assert max(np.max(np.abs(y[s, :, 0] - el)), np.max(np.abs(y[s, :, 1] - er))) <= 2.5 * ULP * scale      # 1.25 x 2^-23; the double-precision tables gave 9 ULP
