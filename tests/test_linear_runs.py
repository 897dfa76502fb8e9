"""The exact linear-run machinery of K1 (zl_plan.h): P += r evaluated as P0 + i*s must equal the sequential fp64
recurrence of SamplerSynthVoice.cpp:223 bit for bit, for every step, across binades, ties and odd mantissas."""
import ctypes as C

import numpy as np
import pytest


@pytest.fixture(scope="module")
def sim(built):
    from cpu_harness.sim import lib
    return lib()


def test_random_ratios_and_positions(sim):
    rng = np.random.default_rng(1)
    total_runs = 0
    for i in range(4000):
        kind = i % 6
        P0 = [0.0, float(rng.integers(0, 1 << 20)), float(rng.uniform(0, 1e6)), float(2.0 ** rng.integers(0, 24)) - float(rng.uniform(0, 3)),
              float(rng.uniform(0, 4)), float(rng.integers(0, 1 << 16)) + 0.5][kind]
        rk = i % 5
        if rk == 0:
            r = 2.0 ** rng.uniform(-1, 1)
        elif rk == 1:
            r = float(rng.choice([0.5, 1.0, 2.0, 0.25, 1.5, 0.75, 44100 / 48000, 48000 / 44100]))
        elif rk == 2:
            r = 2.0 ** (rng.integers(-24, 25) / 12.0) * 44100.0 / 48000.0
        elif rk == 3:
            r = float(rng.integers(1, 1 << 12)) * 2.0 ** -float(rng.integers(8, 60))
            if r > 8:
                r = 1.0 + 2.0 ** -52
        else:
            r = float(rng.uniform(1e-3, 8))
        runs = C.c_longlong(0)
        bad = sim.zlsim_check_linear_runs(max(P0, 0.0), r, 20000, C.byref(runs))
        assert bad == -1, f"P0={P0!r} r={r!r}: first mismatch at step {bad}"
        total_runs += runs.value
    assert total_runs / 4000 < 40          # a handful of runs per 20000 frames, i.e. O(1) work per block


def test_round_half_even_ties(sim):
    for e in range(1, 30):
        u = 2.0 ** (e - 52)
        for P0 in (2.0 ** e, 2.0 ** e + u, 2.0 ** e + 3 * u, 2.0 ** (e + 1) - u):
            for q in (0, 1, 2, 3, 1000, 1001):
                r = q * u + u / 2
                if r > 0:
                    assert sim.zlsim_check_linear_runs(P0, r, 4000, None) == -1, (e, P0, q)


def test_large_positions_small_ratios(sim):
    for P0 in (2.0 ** 30 + 0.25, 2.0 ** 31 - 1000.5, 1e9 + 1 / 3):
        for r in (1e-9, 2.0 ** -40, 1 / 3, 0.999999, 7.25):
            assert sim.zlsim_check_linear_runs(P0, r, 30000, None) == -1


def test_steps_to_reach_is_exact(sim):
    rng = np.random.default_rng(2)
    for _ in range(3000):
        P = float(rng.uniform(1, 1e5))
        r = float(2.0 ** rng.uniform(-2, 2))
        X = P + float(rng.uniform(0, 3000)) * r
        got = sim.zlsim_steps_to_reach(P, r, X)
        # reference answer by the sequential recurrence
        p, i = np.float64(P), 0
        while i < 100000:
            p = np.float64(p + np.float64(r)); i += 1
            if p >= X:
                break
        if got != 0x7fffffff:          # inside the current linear run: must be the exact first crossing
            assert got == i, (P, r, X, got, i)
        else:
            assert i > 1               # the run ended before the threshold


def test_envelope_ramps_are_exact_linear_runs(sim):
    """fp32 ADSR ramps (attack up to 1, decay down to the sustain level, release down to 0): zl_env_linear_run's
    arithmetic progressions reproduce the per-frame float recurrence bit for bit and stop before the state change."""
    lib = sim
    lib.zlsim_check_env_runs.restype = C.c_longlong
    lib.zlsim_check_env_runs.argtypes = [C.c_float, C.c_float, C.c_float, C.c_longlong, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    rng = np.random.default_rng(77)
    total_runs = total_lin = total_steps = 0
    cases = []
    for _ in range(300):
        sr = float(rng.choice([22050.0, 44100.0, 48000.0, 96000.0]))
        t = float(np.float32(rng.uniform(0.0005, 3.0)))
        S = float(np.float32(rng.uniform(0.05, 1.0)))
        cases.append((np.float32(0.0) + np.float32(1.0 / (t * sr)), np.float32(1.0 / (t * sr)), np.float32(1.0)))          # attack from its first value
        cases.append((np.float32(1.0), -np.float32((1.0 - S) / (t * sr)), np.float32(S)))                                   # decay
        e0 = np.float32(rng.uniform(0.01, 1.0))
        cases.append((e0, -np.float32(float(e0) / (t * sr)), np.float32(0.0)))                                              # release
    cases += [(np.float32(0.75), np.float32(2.0 ** -25), np.float32(1.0)),      # tie: half an ulp per step
              (np.float32(0.5), -np.float32(2.0 ** -25), np.float32(0.0)),      # at the bottom of a binade, going down
              (np.float32(0.3), np.float32(1e-12), np.float32(1.0))]            # below half an ulp: never moves
    for e0, d, lim in cases:
        if float(d) == 0.0:
            continue
        runs, lin = C.c_longlong(0), C.c_longlong(0)
        steps = 400000
        bad = lib.zlsim_check_env_runs(float(e0), float(d), float(lim), steps, C.byref(runs), C.byref(lin))
        assert bad == -1, (float(e0), float(d), float(lim), bad)
        total_runs += runs.value; total_lin += lin.value
    assert total_lin > 20 * total_runs                           # the runs are long: tens of frames per real step at least
