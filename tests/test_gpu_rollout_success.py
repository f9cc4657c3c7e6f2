"""On-device success test and rollout_until_success (run_inference.py:137-191)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def test_success_kernel_matches_oracle(oracle):
    from mpinets_amd import _lib
    from mpinets_amd import franka_tables as ft
    from mpinets_amd.robot import franka_fk, frames_to_matrix
    from mpinets_amd.scenes import random_configurations

    B = 300
    q = random_configurations(B, 3)
    qt = q.copy()
    qt[::3] += np.random.default_rng(0).normal(0, 0.004, (len(qt[::3]), 7)).astype(np.float32)  # near misses / hits
    qt[1::3] = random_configurations(B, 9)[1::3]
    tq, tt = torch.from_numpy(q).to(dev()), torch.from_numpy(qt).to(dev())
    targets = frames_to_matrix(franka_fk(tt)[:, ft.LINK_ID["right_gripper"]]).contiguous()
    done = torch.zeros(B, dtype=torch.int32, device=dev())
    steps = torch.zeros(B, dtype=torch.int32, device=dev())
    pe, ca = torch.empty(B, device=dev()), torch.empty(B, device=dev())
    cos_tol = float(np.cos(np.radians(15.0)))
    _lib.call("mpx_franka_success", _lib.ptr(tq), _lib.ptr(targets), B, 0.025, 0.01, cos_tol, _lib.ptr(done),
              _lib.ptr(steps), _lib.ptr(pe), _lib.ptr(ca))
    eff = oracle.franka_fk(q)[:, ft.LINK_ID["right_gripper"]]
    ok, ope, oca = oracle.success(eff, targets.cpu().numpy())
    np.testing.assert_allclose(pe.cpu().numpy(), ope, atol=1e-6)
    np.testing.assert_allclose(ca.cpu().numpy(), oca, atol=1e-6)
    clear = (np.abs(ope - 0.01) > 1e-5) & (np.abs(oca - cos_tol) > 1e-5)
    np.testing.assert_array_equal(done.cpu().numpy()[clear] != 0, ok[clear])
    assert ok.sum() > 10 and (~ok).sum() > 10 and (steps == 1).all()
    # reference semantics: angle via the quaternion of R_eff R_t^T (run_inference.py:183-186)
    Re = eff[:, :9].reshape(B, 3, 3).astype(np.float64)
    Rt = targets.cpu().numpy()[:, :3, :3].astype(np.float64)
    ang = np.degrees(np.arccos(np.clip((np.einsum("bij,bij->b", Re, Rt) - 1) / 2, -1, 1)))
    np.testing.assert_array_equal(ok[clear], ((ope < 0.01) & (ang < 15))[clear])


def _frozen_policy():
    from mpinets_amd.model import MotionPolicyNetwork

    torch.manual_seed(3)
    mdl = MotionPolicyNetwork().to(dev()).eval()
    with torch.no_grad():  # zero displacement: the robot stays where it is
        mdl.decoder[6].weight.zero_()
        mdl.decoder[6].bias.zero_()
    return mdl


def test_rollout_until_success_batched_and_reference_signature():
    from mpinets_amd import franka_tables as ft
    from mpinets_amd.robot import FrankaSampler, franka_fk, frames_to_matrix
    from mpinets_amd.rollout import RolloutEngine, rollout_until_success
    from mpinets_amd.scenes import make_problem_batch

    mdl = _frozen_policy()
    prob = make_problem_batch(4, seed=5, device=dev())
    here = frames_to_matrix(franka_fk(prob["q"])[:, ft.LINK_ID["right_gripper"]])
    targets = prob["target_pose"].clone()
    targets[0] = here[0]  # env 0 and 2 are already at their target; 1 and 3 never get there
    targets[2] = here[2]
    eng = RolloutEngine(mdl, prob)
    eng.track_success(targets)
    traj, lengths = eng.rollout_until_success(max_steps=3)
    assert traj.shape == (4, 4, 7)
    np.testing.assert_array_equal(lengths.cpu().numpy(), [2, 4, 2, 4])
    np.testing.assert_allclose(traj[:, -1].cpu().numpy(), prob["q"].cpu().numpy(), atol=1e-5)  # zero policy
    assert (eng.done.cpu().numpy() != 0).tolist() == [True, False, True, False]
    # with a moving policy, a finished environment stays frozen while the others move
    torch.manual_seed(4)
    from mpinets_amd.model import MotionPolicyNetwork

    mdl2 = MotionPolicyNetwork().to(dev()).eval()
    prob2 = make_problem_batch(4, seed=5, device=dev())
    eng2 = RolloutEngine(mdl2, prob2)
    eng2.track_success(targets)
    eng2.done[0] = 1
    traj2, _ = eng2.rollout_until_success(max_steps=2)
    assert torch.equal(traj2[0, 0], traj2[0, 2]) and not torch.equal(traj2[1, 0], traj2[1, 2])
    # reference signature, one problem (run_inference.py:137-191)
    np.random.seed(0)
    smp = FrankaSampler(dev())
    one = make_problem_batch(1, seed=6, device=dev())
    t_here = frames_to_matrix(franka_fk(one["q"])[:, ft.LINK_ID["right_gripper"]])[0].cpu().numpy()
    out = rollout_until_success(mdl, one["q"][0].cpu().numpy(), t_here, one["xyz"], smp, max_rollout_length=5)
    assert out.shape == (2, 7)
    far = rollout_until_success(mdl, one["q"][0].cpu().numpy(), one["target_pose"][0].cpu().numpy(), one["xyz"], smp,
                                max_rollout_length=4)
    assert far.shape == (5, 7)
