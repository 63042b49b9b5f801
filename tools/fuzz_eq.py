"""Randomised GPU parity sweep of the parametric EQ (run on the GPU box; not part of the test suite): random filter
sets, rates, stream counts and scripts of setTarget / process / drain / reset against the oracle's processor."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import airwave_amd as aw
import airwave_oracle as orc

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
rng = np.random.default_rng(seed)
ULP = 2.0 ** -23


def rand_def():
    if rng.random() < 0.1:
        return None
    k = int(rng.choice([0, 1, 2, 5, 10, 31, 64]))
    fl = [(int(rng.integers(0, 3)), float(np.exp(rng.uniform(np.log(20.0), np.log(18000.0)))), float(rng.uniform(-12, 12)), float(rng.uniform(0.3, 6.0)))
          for _ in range(k)]
    return float(rng.uniform(-12, 6)), fl


def mk(mod, d, EF, ED):
    if d is None:
        return None
    return ED(d[0], [EF(1, None, True, t, f, g, q) for t, f, g, q in d[1]])


fails, n = [], 0
stats = {'frames': 0, 'nonzero': 0, 'worst': 0.0}
t_end = time.time() + budget
while time.time() < t_end:
    fs = float(rng.choice([44100.0, 48000.0, 96000.0]))
    S = int(rng.choice([1, 2, 3, 7, 64, 300]))
    p = aw.ParametricEqualizerProcessor(fs, maxFramesPerCallback=0, n_streams=S)
    check = sorted(set([0, S - 1, int(rng.integers(0, S))]))
    refs = {s: orc.ParametricEqualizerProcessor(fs) for s in check}
    for r in refs.values():
        r.max_frames = 1 << 30
    worst, script = 0.0, []
    try:
        for _ in range(int(rng.integers(2, 9))):
            op = rng.choice(["t", "p", "p", "p", "d", "r"])
            if op == "t":
                d = rand_def()
                script.append(("t", None if d is None else (d[0], len(d[1]))))
                p.setTarget(mk(aw, d, aw.EqualizerFilter, aw.EqualizerDefinition))
                for r in refs.values():
                    r.set_target(mk(orc, d, orc.EqualizerFilter, orc.EqualizerDefinition))
            elif op == "d":
                script.append(("d",)); p.drainRetiredStates(); [r.drain_retired_states() for r in refs.values()]
            elif op == "r":
                script.append(("r",)); p.reset(); [r.reset() for r in refs.values()]
            else:
                m = int(rng.choice([1, 2, 15, 16, 17, 31, 32, 33, 255, 256, 257, 959, 960, 961, 1920, 4095, 4096, 4097])) if rng.random() < 0.6 else int(rng.integers(1, 30000))
                script.append(("p", m))
                x = rng.uniform(-0.5, 0.5, (S, m, 2)).astype(np.float32)
                y = p.process_batch(x)
                for s, r in refs.items():
                    el, er = r.process(x[s, :, 0], x[s, :, 1])
                    scale = max(1.0, float(np.max(np.abs(el))), float(np.max(np.abs(er))))
                    dev = max(float(np.max(np.abs(y[s, :, 0] - el))), float(np.max(np.abs(y[s, :, 1] - er)))) / scale
                    if dev > 4 * ULP and os.environ.get("AW_FUZZ_TRACE"):
                        i = int(np.argmax(np.abs(y[s, :, 0] - el)))
                        print(f"TRACE call {len(script)} frames {m} stream {s}: dev {dev / ULP:.2f} ulp at frame {i}, peak {scale:.3g}, got {y[s, i, 0]!r} want {el[i]!r}", flush=True)
                    worst = max(worst, dev)
        ok = worst <= 4 * ULP
    except Exception as e:                              # noqa: BLE001
        ok, worst = False, repr(e)
    n += 1
    if ok:
        stats['worst'] = max(stats['worst'], worst); stats['nonzero'] += worst > 0; stats['frames'] += sum(a[1] for a in script if a[0] == 'p')
    if not ok:
        fails.append((fs, S, script, worst))
        print("FAIL", fails[-1], flush=True)
print(f"seed {seed}: {n} scripts, {len(fails)} failures; frames per stream {stats['frames']}, scripts with a nonzero deviation {stats['nonzero']}, worst {stats['worst'] / ULP:.2f} ulp")
