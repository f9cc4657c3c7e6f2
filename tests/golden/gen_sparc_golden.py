"""Generates tests/golden/sparc_golden.npz by IMPORTING the reference's mpinets/third_party/sparc.py (pure numpy: no
stub of anything is needed) and calling ``sparc`` the way ``Evaluator.calculate_smoothness`` does
(mpinets/metrics.py:387-409: speed profile of a trajectory, fs = 1 / dt).  Runs only in the build container
(needs /root/reference).  Only data is committed: inputs, lengths, sampling rates and the reference's results.

    python tests/golden/gen_sparc_golden.py
"""
import contextlib
import importlib.util
import io
import os
import sys

import numpy as np

sys.dont_write_bytecode = True  # importing the reference must not leave __pycache__ in /root/reference

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("ref_sparc", "/root/reference/mpinets/third_party/sparc.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)


def speed_profile(rng, n, kind):
    """A speed profile of n samples: what calculate_smoothness hands to sparc (||diff(q)|| / dt)."""
    t = np.linspace(0.0, 1.0, n + 1)
    if kind == "minimum_jerk":  # one smooth reach in joint space
        s = 10 * t ** 3 - 15 * t ** 4 + 6 * t ** 5
        q = np.outer(s, rng.uniform(-1.5, 1.5, 7))
    elif kind == "noisy":  # a reach with per-step jitter (what an unconverged policy produces)
        s = 10 * t ** 3 - 15 * t ** 4 + 6 * t ** 5
        q = np.outer(s, rng.uniform(-1.5, 1.5, 7)) + rng.normal(scale=0.01, size=(n + 1, 7))
    elif kind == "two_moves":  # stop in the middle
        s = np.where(t < 0.5, 0.5 * (1 - np.cos(2 * np.pi * t)) * 0.5, 0.5 + 0.5 * (1 - np.cos(2 * np.pi * (t - 0.5))) * 0.5)
        q = np.outer(s, rng.uniform(-1.0, 1.0, 7))
    else:  # random walk
        q = np.cumsum(rng.normal(scale=0.02, size=(n + 1, 7)), axis=0)
    return q


def main():
    rng = np.random.default_rng(2024)
    out = {}
    # the known answer of the reference's own docstring (sparc.py:86-91)
    t = np.arange(-1, 1, 0.01)
    move = np.exp(-5 * pow(t, 2))
    sal, _, _ = ref.sparc(move, fs=100.0)
    assert "%.5f" % sal == "-1.41403"
    out["doc_move"], out["doc_fs"], out["doc_sal"] = move, np.float64(100.0), np.float64(sal)
    # trajectories the way calculate_smoothness sees them: lengths 2..150 waypoints, dt = 0.12 s (run_inference's rate)
    # and two other rates; padded to 150 rows (rows past the length repeat the last configuration)
    T = 150
    kinds = ["minimum_jerk", "noisy", "two_moves", "walk"]
    lengths = np.array([2, 3, 4, 5, 8, 9, 16, 17, 31, 32, 33, 50, 64, 65, 100, 128, 129, 150] * 2, dtype=np.int32)
    dts = np.array([0.12, 0.08, 0.02], dtype=np.float64)
    traj = np.zeros((len(lengths), T, 7), dtype=np.float64)
    sal_q = np.zeros((len(dts), len(lengths)), dtype=np.float64)
    for b, n in enumerate(lengths):
        q = speed_profile(rng, int(n) - 1, kinds[b % 4])
        traj[b, :n] = q
        traj[b, n:] = q[-1]
        for d, dt in enumerate(dts):
            move = np.linalg.norm(np.diff(q, 1, axis=0) / dt, axis=1)
            with contextlib.redirect_stdout(io.StringIO()):
                s, _, _ = ref.sparc(move, 1.0 / dt)
            sal_q[d, b] = s
    out["traj"], out["lengths"], out["dts"], out["config_sparc"] = traj, lengths, dts, sal_q
    # a trajectory that does not move at all: "All movement was 0, returning 0" (sparc.py:93-95)
    with contextlib.redirect_stdout(io.StringIO()):
        z, _, _ = ref.sparc(np.zeros(20), 1.0 / 0.12)
    out["zero_sal"] = np.float64(z)
    # raw speed profiles with other parameters (padlevel / fc / amp_th) than the defaults
    raw = np.abs(rng.normal(size=(6, 90))) * np.hanning(90)[None, :]
    raw_len = np.array([90, 77, 64, 33, 12, 5], dtype=np.int32)
    params = [(4, 10.0, 0.05), (2, 5.0, 0.1), (3, 20.0, 0.02)]
    raw_sal = np.zeros((len(params), len(raw)), dtype=np.float64)
    for p, (pad, fc, th) in enumerate(params):
        for b in range(len(raw)):
            s, _, _ = ref.sparc(raw[b, :raw_len[b]], 25.0, padlevel=pad, fc=fc, amp_th=th)
            raw_sal[p, b] = s
    out["raw"], out["raw_len"], out["raw_fs"] = raw, raw_len, np.float64(25.0)
    out["raw_params"], out["raw_sal"] = np.array(params, dtype=np.float64), raw_sal
    np.savez_compressed(os.path.join(HERE, "sparc_golden.npz"), **out)
    print({k: np.asarray(v).shape for k, v in out.items()})


if __name__ == "__main__":
    main()
