"""SURVEY 8(f) rows 1-2 ON THE DEVICE: the batched GAE scan, the PPO losses and the
reference-order update run on cuda and are compared with the vectors the reference's own
classes produced on CPU (tests/golden/ppo_gae.npz, ppo_loss.npz, ppo_update.npz; generators
oracle/gen_golden_policy.py, oracle/gen_golden_ppo_update.py).  Tolerances: returns / advantages
1e-5 relative (north_star), losses 1e-5, gradients 1e-3 relative (GEMM summation order of the
device BLAS), parameters after ~40 Adam steps 1e-3."""
import os

import numpy as np
import pytest
import torch

from golden_util import GOLDEN, load
from rvo3d_amd.policy import gae_scan, multi_ppo, rnn_ac
from rvo3d_amd.policy.multi_ppo import RolloutBuffer
from test_policy_ppo import _Space, _update_case, small_ac

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_gae_scan_on_device_matches_multi_PPObuf():
    fx = load(os.path.join(GOLDEN, "ppo_gae.npz"))
    rew, val, cut = (torch.as_tensor(fx[k]).to(DEV) for k in ("rew", "val", "cuts"))
    adv, ret = gae_scan(rew, val, cut.bool(), float(fx["gamma"]), float(fx["lam"]))
    assert adv.is_cuda and adv.dtype == torch.float32
    np.testing.assert_allclose(adv.cpu().numpy(), fx["adv"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(ret.cpu().numpy(), fx["ret"], rtol=1e-5, atol=1e-6)
    # the rollout buffer's [T, E, N] layout with per-env cuts: every column is that same scan
    T, E, N = rew.shape[0], 5, 3
    buf = RolloutBuffer(T, E, N, 21, 3, DEV, float(fx["gamma"]), float(fx["lam"]))
    z = torch.zeros((E, N), device=DEV)
    for t in range(T):
        buf.store(torch.zeros((E, N, 21), device=DEV), torch.zeros((E, N), dtype=torch.int32, device=DEV),
                  torch.zeros((E, N, 3), device=DEV), z + rew[t], z + val[t], z)
        if bool(cut[t]):
            buf.finish_path(torch.ones(E, dtype=torch.bool, device=DEV))
    d = buf.get()
    assert d["shape"] == (T, E, N)
    np.testing.assert_allclose(d["adv"].view(T, E, N)[:, 4, 2].cpu().numpy(), fx["adv"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(d["ret"].view(T, E, N)[:, 0, 1].cpu().numpy(), fx["ret"], rtol=1e-5, atol=1e-6)


def test_losses_and_gradients_on_device_match_reference():
    fx = load(os.path.join(GOLDEN, "policy_rnn_ac.npz"))
    lx = load(os.path.join(GOLDEN, "ppo_loss.npz"))
    ac = small_ac(fx).to(DEV)

    class Env:
        E, N, W, device = 1, 1, 102, torch.device(DEV)
    tr = multi_ppo(Env(), ac, steps_per_epoch=2, use_gpu=True)
    dv = lambda a: torch.as_tensor(a).to(DEV)
    data = dict(obs=dv(fx["obs"]), cnt=dv(fx["count"]), act=dv(fx["act"]), adv=dv(lx["adv"]),
                ret=dv(lx["ret"]), logp=dv(lx["logp_old"]))
    ac.zero_grad()
    loss_pi, info = tr.compute_loss_pi(data)
    loss_pi.backward()
    assert loss_pi.is_cuda
    assert abs(float(loss_pi.detach()) - float(lx["loss_pi"])) < 1e-5
    for k in ("kl", "ent", "cf"):
        assert abs(info[k] - float(lx[k])) < 1e-5, k
    g = dict(ac.named_parameters())
    for name, key, rtol in (("pi.log_std", "g_pi_log_std", 1e-4), ("pi.net_out.4.weight", "g_pi_out", 1e-4),
                            ("pi.rnn_reader.rnn_net.weight_ih_l0", "g_pi_gru", 1e-3)):
        np.testing.assert_allclose(g[name].grad.cpu().numpy(), lx[key], rtol=rtol, atol=1e-6, err_msg=name)
    ac.zero_grad()
    loss_v = tr.compute_loss_v(data)
    loss_v.backward()
    assert abs(float(loss_v.detach()) - float(lx["loss_v"])) < 1e-4 * max(1.0, abs(float(lx["loss_v"])))
    np.testing.assert_allclose(g["v.v_net.4.weight"].grad.cpu().numpy(), lx["g_v_out"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(g["pi.rnn_reader.rnn_net.weight_hh_l0_reverse"].grad.cpu().numpy(),
                               lx["g_v_gru"], rtol=1e-3, atol=1e-6)


@pytest.mark.parametrize("case", ["1", "2"])
def test_reference_order_update_on_device(case):
    """multi_ppo.update of the reference (multi_ppo.py:341-376) reproduced on cuda: same agent
    order, same number of policy steps per agent (KL stop), same parameters afterwards."""
    fx = load(os.path.join(GOLDEN, "ppo_update.npz"))
    tr, st = _update_case(fx, case, device=DEV)
    assert st["order"] == fx["order" + case].tolist()
    assert st["pi_steps"] == fx["pi_steps" + case].tolist()
    for k, v in tr.ac.state_dict().items():
        assert v.is_cuda
        np.testing.assert_allclose(v.cpu().numpy(), fx[f"w{case}:" + k], rtol=1e-3, atol=1e-5, err_msg=k)


def test_split_weight_gradient_equals_nn_linear():
    """policy_rnn_ac._Linear: on large GPU batches the weight gradient is a batch of partial products + a sum;
    forward identical, gradients equal to nn.Linear's up to the order of the float32 sums (<= 2e-5 of the largest entry);
    small batches and inference take nn.Linear's own path (bitwise equal)."""
    from rvo3d_amd.policy.policy_rnn_ac import _Linear, mlp
    torch.manual_seed(3)
    net = mlp([102, 256, 256, 3], torch.nn.ReLU, torch.nn.Tanh).cuda()
    assert all(isinstance(m, torch.nn.Linear) for m in net if hasattr(m, "weight"))
    ref = torch.nn.Sequential(*[torch.nn.Linear(m.in_features, m.out_features) if isinstance(m, _Linear) else type(m)()
                                for m in net]).cuda()
    ref.load_state_dict(net.state_dict())                       # same keys: the reference's checkpoints load
    for rows in (65536, 1000):
        x = torch.randn(rows, 102, device="cuda")
        tgt = torch.randn(rows, 3, device="cuda")
        for m in (net, ref):
            m.zero_grad()
            ((m(x) - tgt) ** 2).mean().backward()
        with torch.no_grad():
            assert torch.equal(net(x), ref(x))
        for (n1, p1), (_, p2) in zip(net.named_parameters(), ref.named_parameters()):
            scale = float(p2.grad.abs().max())
            if rows < _Linear.split_rows:
                assert torch.equal(p1.grad, p2.grad), n1
            else:
                assert float((p1.grad - p2.grad).abs().max()) <= 2e-5 * scale, (n1, float((p1.grad - p2.grad).abs().max()), scale)
