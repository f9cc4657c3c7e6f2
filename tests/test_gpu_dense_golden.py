"""GPU: the HIP path vs the reference's model.py run on DENSE clouds (tests/golden/gen_dense_golden.py).

Two of the three environments overflow both ball queries (up to 1500 hits for 128 slots in the first module, 122+ of the 128
queries of the second): the truncation rule (first ``nsample`` hits by index, bit-exact) and the grouped-MLP kernels on rows
with NO padding to elide are compared with the reference's own grouping, shared MLPs and max-pools -- in fp32 (factored
and unfactored second module) and in ``bf16x3``, at the same 1e-5 as the sparse goldens; one sparse environment shares the
batch, so full and short rows meet in the same launch."""
import numpy as np
import pytest
import torch

from test_gpu_model_golden import T, _precision, dev
from test_oracle_model import NR, golden_state_dict

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def mdl(model_golden, dense_golden):
    import seeded_weights
    from mpinets_amd.model import TrainingMotionPolicyNetwork

    sd = golden_state_dict(model_golden)
    assert seeded_weights.digest(sd) == str(dense_golden["param_sha256"])
    m = TrainingMotionPolicyNetwork(num_robot_points=NR).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return m.to(dev())


@pytest.mark.parametrize("prec,factored", [("fp32", True), ("fp32", False), ("bf16x3", True), ("bf16x3", False)])
def test_forward_indices_and_every_module_output(dense_golden, mdl, prec, factored):
    g = dense_golden
    mdl.set_factored(factored)
    aux = {}
    with _precision(mdl, prec), torch.no_grad():
        dq = mdl(T(g["d_xyz"]), T(g["d_q"]), aux=aux)
        dq_plain = mdl(T(g["d_xyz"]), T(g["d_q"]))  # (engine form: hit-slot-only index rows, counts, padding elided)
        cnt1, cnt2 = (c.cpu().numpy() for c in mdl.point_cloud_encoder.last_counts)
    mdl.set_factored(True)
    # index path: bit-exact, overflowing rows included
    np.testing.assert_array_equal(aux["fps_idx1"].cpu().numpy(), g["d_fps1"])
    np.testing.assert_array_equal(aux["fps_idx2"].cpu().numpy(), g["d_fps2"])
    np.testing.assert_array_equal(aux["ball_idx1"].cpu().numpy(), g["d_ball1"].astype(np.int32))
    np.testing.assert_array_equal(aux["ball_idx2"].cpu().numpy(), g["d_ball2"].astype(np.int32))
    h1, h2 = g["d_hits1"].astype(np.int32), g["d_hits2"].astype(np.int32)
    np.testing.assert_array_equal(cnt1, np.minimum(h1, 128))  # the counts the fused kernels walk: full rows where the ball overflows
    np.testing.assert_array_equal(cnt2, np.minimum(h2, 128))
    assert ((cnt1 == 128).sum(1)[:2] >= 20).all() and ((cnt2 == 128).sum(1)[:2] >= 100).all()
    sa3_in = aux["sa3_in"].cpu().numpy()
    np.testing.assert_array_equal(sa3_in[:, :, :3], g["d_xyz2"])
    errs = {
        "f1": np.abs(aux["f1"].cpu().numpy() - g["d_feat1"].transpose(0, 2, 1)).max(),
        "f2": np.abs(sa3_in[:, :, 3:3 + 256] - g["d_feat2"].transpose(0, 2, 1)).max(),
        "f3": np.abs(aux["f3"].cpu().numpy() - g["d_feat3"][:, :, 0]).max(),
        "encoding": np.abs(aux["encoding"].cpu().numpy() - g["d_encoding"]).max(),
        "dq": np.abs(dq.cpu().numpy() - g["d_out"]).max(),
        "dq_plain": np.abs(dq_plain.cpu().numpy() - g["d_out"]).max(),
    }
    print("dense clouds, HIP (%s, factored=%s) vs reference-run golden:" % (prec, factored), {k: "%.2e" % v for k, v in errs.items()})
    bars = {k: TOL for k in errs}
    if prec == "bf16x3":  # (the bar is on the policy deltas; intermediate rows carry O(1) values -- as in test_gpu_model_golden)
        bars.update(f1=2e-5, f2=5e-5, f3=5e-5, encoding=5e-5)
    assert all(errs[k] <= bars[k] for k in errs), errs


def test_forward_single_c_call(dense_golden, mdl):
    g = dense_golden
    with torch.no_grad():
        dq = mdl.forward_native(T(g["d_xyz"]), T(g["d_q"]))
    np.testing.assert_allclose(dq.cpu().numpy(), g["d_out"], rtol=0, atol=TOL)


def test_elision_on_and_off_agree_bit_for_bit_on_full_rows(dense_golden, mdl):
    g = dense_golden
    with torch.no_grad():
        a = mdl(T(g["d_xyz"]), T(g["d_q"]))
        mdl.set_elide_padding(False)
        b = mdl(T(g["d_xyz"]), T(g["d_q"]))
        mdl.set_elide_padding(True)
    assert torch.equal(a, b)


@pytest.mark.parametrize("prec", ["fp32", "bf16x3"])
def test_rollout_from_the_dense_slabs(dense_golden, mdl, prec):
    from mpinets_amd.robot import FrankaSampler

    g = dense_golden
    smp, subsets, calls = FrankaSampler(dev()), T(g["d_subsets"]), []

    def sampler(q):
        out = torch.empty((q.size(0), NR, 3), device=dev())
        smp.sample_into(q, out, subsets[len(calls)])
        calls.append(1)
        return out

    slab = T(g["d_xyz"].copy())
    with _precision(mdl, prec), torch.no_grad():
        traj = mdl.rollout({"xyz": slab, "configuration": T(g["d_q"].copy())}, 5, sampler)
    err = np.abs(torch.stack(traj).cpu().numpy() - g["d_traj"]).max()
    print("5-step rollout from dense slabs (%s) vs reference-run golden: %.2e" % (prec, err))
    assert err <= 5 * TOL
    np.testing.assert_allclose(slab[:, :NR, :3].cpu().numpy(), g["d_robot"], rtol=0, atol=5 * TOL)
    np.testing.assert_array_equal(slab[:, NR:].cpu().numpy(), g["d_xyz"][:, NR:])
