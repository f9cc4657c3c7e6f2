"""SPARC smoothness (row N3, metrics.py:387-409 / third_party/sparc.py): the oracle's restatement and the engine's batched
form against golden vectors that the reference's OWN function produced (tests/golden/gen_sparc_golden.py)."""
import os

import numpy as np
import pytest
import torch

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "sparc_golden.npz"))


def test_oracle_sparc_is_pinned_to_the_reference(oracle):
    assert "%.5f" % oracle.sparc(GOLD["doc_move"], float(GOLD["doc_fs"])) == "-1.41403"  # the reference docstring's own answer
    assert abs(oracle.sparc(GOLD["doc_move"], float(GOLD["doc_fs"])) - float(GOLD["doc_sal"])) < 1e-12
    for d, dt in enumerate(GOLD["dts"]):
        for b, n in enumerate(GOLD["lengths"]):
            q = GOLD["traj"][b, :n]
            s = oracle.sparc(np.linalg.norm(np.diff(q, 1, axis=0) / dt, axis=1), 1.0 / dt)
            assert abs(s - GOLD["config_sparc"][d, b]) < 1e-12, (dt, n)
    for p, (pad, fc, th) in enumerate(GOLD["raw_params"]):
        for b in range(len(GOLD["raw"])):
            s = oracle.sparc(GOLD["raw"][b, :GOLD["raw_len"][b]], float(GOLD["raw_fs"]), int(pad), fc, th)
            assert abs(s - GOLD["raw_sal"][p, b]) < 1e-12
    assert oracle.sparc(np.zeros(20), 1.0 / 0.12) == float(GOLD["zero_sal"]) == 0.0


def test_batched_sparc_matches_the_reference_on_a_ragged_batch():
    from mpinets_amd.smoothness import fft_length, sparc_batched, speed_profile

    for n in list(range(1, 70)) + [127, 128, 129, 149, 150]:
        assert fft_length(n) == int(pow(2, np.ceil(np.log2(n)) + 4))
    got = sparc_batched(torch.from_numpy(GOLD["doc_move"])[None], None, float(GOLD["doc_fs"]))
    assert abs(got.item() - float(GOLD["doc_sal"])) < 1e-12
    traj, ln = torch.from_numpy(GOLD["traj"]), torch.from_numpy(GOLD["lengths"])
    for d, dt in enumerate(GOLD["dts"]):  # 36 trajectories of 2..150 waypoints in ONE call (six FFT lengths)
        got = sparc_batched(speed_profile(traj, dt), ln - 1, 1.0 / dt)
        np.testing.assert_allclose(got.numpy(), GOLD["config_sparc"][d], rtol=0, atol=1e-12)
    raw, rl = torch.from_numpy(GOLD["raw"]), torch.from_numpy(GOLD["raw_len"])
    for p, (pad, fc, th) in enumerate(GOLD["raw_params"]):
        got = sparc_batched(raw, rl, float(GOLD["raw_fs"]), int(pad), fc, th)
        np.testing.assert_allclose(got.numpy(), GOLD["raw_sal"][p], rtol=0, atol=1e-12)
    # float32 trajectories (what the engine holds): the score moves by the rounding of the waypoints only
    got32 = sparc_batched(speed_profile(traj.float(), float(GOLD["dts"][0])), ln - 1, 1.0 / float(GOLD["dts"][0]))
    smooth = np.arange(len(ln)) % 4 == 0  # (the minimum-jerk ones: no bin sits near the amplitude threshold by accident)
    np.testing.assert_allclose(got32.numpy()[smooth], GOLD["config_sparc"][0][smooth], rtol=0, atol=1e-4)


def test_batched_sparc_edge_cases():
    from mpinets_amd.smoothness import sparc_batched

    m = torch.zeros((3, 20), dtype=torch.float64)
    m[1, :5] = torch.tensor([0.0, 1.0, 2.0, 1.0, 0.0])
    m[2, 7:] = 5.0  # moving only past its length -> a profile that is zero where it counts
    got = sparc_batched(m, torch.tensor([20, 5, 7]), 1.0 / 0.12)
    assert got[0].item() == 0.0 and got[2].item() == 0.0 and got[1].item() < 0.0  # "All movement was 0, returning 0"
    one = sparc_batched(torch.tensor([[3.0, 9.0]], dtype=torch.float64), torch.tensor([1]), 8.0)  # a single sample: flat spectrum
    assert torch.isfinite(one).all() and one.item() <= 0.0
    with pytest.raises(AssertionError):
        sparc_batched(m, torch.tensor([0, 5, 7]), 8.0)


def test_summary_uses_the_reference_keys():
    """BatchedEvaluator.metrics (Evaluator.metrics, metrics.py:566-664) on a hand-made result."""
    from mpinets_amd.metrics import BatchedEvaluator

    t = torch.tensor
    res = {"success": t([True, False, True, False]), "collision": t([False, True, False, False]),
           "self_collision": t([False, False, False, True]), "joint_limit_violation": t([False, False, False, False]),
           "physical_violations": t([False, True, False, True]), "position_error": t([0.5, 3.0, 0.9, 20.0]),
           "orientation_error": t([2.0, 20.0, 14.0, 170.0]), "eff_position_path_length": t([1.0, 9.0, 3.0, 9.0]),
           "eff_orientation_path_length": t([10.0, 99.0, 30.0, 99.0]),
           "config_smoothness": t([-1.7, -1.5, -2.0, -1.7], dtype=torch.float64),
           "eff_smoothness": t([-1.7, -1.7, -1.5, -1.7], dtype=torch.float64)}
    m = BatchedEvaluator.metrics(res)
    assert m["success"] == 50.0 and m["total"] == 4 and m["env collision"] == 25.0 and m["self collision"] == 25.0
    assert m["joint violation"] == 0.0 and m["physical violations"] == 50.0
    assert (m["1 cm"], m["5 cm"], m["15 deg"], m["30 deg"], m["165 deg"]) == (50.0, 75.0, 50.0, 75.0, 25.0)
    assert m["is smooth"] == 50.0 and abs(m["average config sparc"] + 1.725) < 1e-12 and abs(m["average eff sparc"] + 1.65) < 1e-12
    assert m["eff position path length"] == (2.0, 1.0) and m["eff orientation path length"] == (20.0, 10.0)  # successes only
