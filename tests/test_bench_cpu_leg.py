"""CPU: the `cpu_baseline` leg of bench.py (worker processes over the oracle) on a tiny sample -- it must run without a
GPU, use the spawn start method (the GPU runtime is up in the real run) and return the fields the bench line documents."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from test_oracle_model import golden_state_dict


class _Weights:
    def __init__(self, sd):
        self.sd = sd

    def state_dict(self):
        return {k: torch.from_numpy(v) for k, v in self.sd.items()}


def test_cpu_baseline_runs_over_worker_processes(model_golden, oracle):
    import bench

    g = model_golden
    n = 3
    prob = {"xyz": torch.from_numpy(g["f_xyz"][:n].copy()), "q_norm": torch.from_numpy(g["f_q"][:n].copy())}
    for k in g:
        if k.startswith(("v_cuboid", "v_cylinder")):
            prob[k[2:]] = torch.from_numpy(g[k][:n].copy())
    out = bench.cpu_baseline(prob, _Weights(golden_state_dict(g)), n)
    assert out["kind"] == "port" and out["unit"] == "env-steps/s" and out["value"] > 0
    assert out["cores"] == out["processes"] * out["threads_per_process"] <= out["host_cores"]
    assert out["scalar_1t"]["cores"] == 1 and out["scalar_1t"]["value"] > 0
    # the torch restatement the workers time agrees with the float64 oracle (same step, two arithmetic widths)
    from mpinets_amd import franka_tables as ft

    sd = golden_state_dict(g)
    with torch.no_grad():
        dq32 = oracle.policy_forward_torch({k: torch.from_numpy(v) for k, v in sd.items()}, g["f_xyz"][:1], torch.from_numpy(g["f_q"][:1])).numpy()
    np.testing.assert_allclose(dq32, g["f_out"][:1], rtol=0, atol=1e-5)
    assert ft.JOINT_LIMITS_REAL.shape == (7, 2)
