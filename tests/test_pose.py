"""LearnPose / SE(3) exponential: closed forms against torch.linalg.matrix_exp of the twist."""
import torch

from nefes_amd.pose import LearnPose, make_c2w, se3_exp, so3_exp
from oracle import ref_cpu as O


def twist(tau, phi):
    K = torch.tensor([[0., -phi[2], phi[1]], [phi[2], 0., -phi[0]], [-phi[1], phi[0], 0.]], dtype=torch.float64)
    M = torch.zeros(4, 4, dtype=torch.float64)
    M[:3, :3], M[:3, 3] = K, tau
    return M


def test_se3_exp_matches_matrix_exponential():
    g = torch.Generator().manual_seed(0)
    for scale in (1e-6, 1e-3, 0.3, 2.5):
        x = (torch.rand(6, generator=g, dtype=torch.float64) - .5) * 2 * scale
        ref = torch.linalg.matrix_exp(twist(x[:3], x[3:]))
        assert torch.allclose(se3_exp(x), ref, rtol=1e-9, atol=1e-12)
        assert torch.allclose(so3_exp(x[3:]), ref[:3, :3], rtol=1e-9, atol=1e-12)
    assert torch.equal(se3_exp(torch.zeros(6)), torch.eye(4))


def test_make_c2w_is_the_benchmark_pose_and_batched():
    r, t = torch.tensor([0.10, -0.20, 0.05]), torch.tensor([0.10, 0.20, 0.30])
    assert torch.allclose(make_c2w(r, t)[:3], O.bench_pose(), atol=1e-7)
    both = make_c2w(torch.stack([r, -r]), torch.stack([t, t]))
    assert both.shape == (2, 4, 4) and torch.allclose(both[0], make_c2w(r, t))


def test_learnpose_parameters_and_gradients():
    init = torch.eye(4).repeat(3, 1, 1)
    init[:, :3, 3] = torch.tensor([1., 2., 3.])
    for lie in (False, True):
        m = LearnPose(3, True, True, init_c2w=init, lietorch=lie)
        assert set(dict(m.named_parameters())) == {"init_c2w", "r", "t"} and not m.init_c2w.requires_grad
        c = m(1)
        assert c.shape == (4, 4) and torch.allclose(c, init[1])           # zero delta = the initial pose
        (c[:3, :4] * torch.arange(12.).reshape(3, 4)).sum().backward()
        assert m.r.grad is not None and m.t.grad is not None and m.r.grad[0].abs().sum() == 0 and m.t.grad[1].abs().sum() > 0
        assert m(torch.tensor([0, 2])).shape == (2, 4, 4)


def test_fusion_net_matches_reference(golden):
    """FusionNet + run_fusion_net (nerfh_nff.py:356-418,578-603) against the reference module: same seed-0 init
    (checksums), same output in the never-.eval()'ed train mode the refinement loop really uses."""
    import numpy as np
    from nefes_amd.field import NeRFH_NFF
    g = golden("fusion")
    net = NeRFH_NFF('coarse', W=128, f_dim=16)
    for k, v in net.fusion_net.state_dict().items():
        np.testing.assert_allclose([v.double().sum().item(), v.double().abs().sum().item()], g["sd." + k], rtol=0, atol=0)
    r_rgb, r_feat, fused = net.run_fusion_net(torch.from_numpy(g["rgb"]).clone(), torch.from_numpy(g["feat"]).clone(), 6, 8, 1)
    np.testing.assert_allclose(fused.detach().numpy(), g["fused"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(r_feat.detach().numpy(), g["render_feat"], rtol=0, atol=0)
