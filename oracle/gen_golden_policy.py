#!/usr/bin/env python3
"""Golden vectors for the policy / PPO surface, produced by
running the PYTHON REFERENCE (train/policy/*.py) on CPU in this container.

Stand-ins (absent packages / device, no numerics): `gym.spaces.Box` record; and
`torch.cuda.synchronize` -> no-op, because policy_rnn_ac.py:39 calls it
unconditionally and this container has no GPU.

Outputs (tests/golden/):
  policy_rnn_ac.npz : reference rnn_ac (biGRU 256, MLP 256-256) with the weights of
                      train/model_save/r8_0/r8_0_check_point_0.pt (loaded weights_only)
                      -> mu, std, logp, v for ragged observations, list and single paths
  ppo_gae.npz       : multi_PPObuf.store / finish_path / get (multi_ppo.py:39-94)
  ppo_loss.npz      : multi_ppo.compute_loss_pi / compute_loss_v (multi_ppo.py:379-404)
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def install():
    gym = types.ModuleType("gym")
    spaces = types.ModuleType("gym.spaces")

    class Box:
        def __init__(self, low=None, high=None, shape=None, dtype=None):
            self.low, self.high, self.dtype = low, high, dtype
            self.shape = shape if shape is not None else np.shape(low)

    spaces.Box = Box
    gym.spaces = spaces
    gym.Env = type("Env", (), {})
    sys.modules["gym"], sys.modules["gym.spaces"] = gym, spaces
    torch.cuda.synchronize = lambda *a, **k: None  # policy_rnn_ac.py:39, multi_ppo.py:153
    sys.path.insert(0, os.path.join(REF, "train"))
    return Box


def ragged_obs(rng, B, nm=10):
    cnt = rng.integers(0, nm + 1, B)
    cnt[:4] = [0, 1, nm, 2]
    obs = []
    for c in cnt:
        k = max(int(c), 1)
        o = np.zeros(12 + 9 * k, np.float32)
        o[:12] = np.round(rng.normal(0, 3, 12), 2)
        if c > 0:
            o[12:] = np.round(rng.normal(0, 2, 9 * k), 2)
        obs.append(o)
    return obs, cnt.astype(np.int32)


def pad(obs, nm=10):
    out = np.zeros((len(obs), 12 + 9 * nm), np.float32)
    for i, o in enumerate(obs):
        out[i, :len(o)] = o
    return out


def main():
    Box = install()
    from policy.policy_rnn_ac import rnn_ac
    import policy.multi_ppo as mp

    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    obs_space = Box(-np.inf, np.inf, shape=(21,), dtype=np.float32)
    act_space = Box(low=np.array([-1, -1, -1]), high=np.array([1, 1, 1]), dtype=np.float32)
    # (B) the trained architecture: only key names + shapes travel (the weights do not)
    big = rnn_ac(obs_space, act_space, 12, 9, 256, (256, 256), (256, 256), torch.nn.ReLU,
                 torch.nn.Tanh, torch.nn.Identity, use_gpu=False, rnn_mode="biGRU")
    ck = torch.load(os.path.join(REF, "train/model_save/r8_0/r8_0_check_point_0.pt"),
                    map_location="cpu", weights_only=True)
    big.load_state_dict(ck["model_state"], strict=True)
    # (A) a small randomly initialised instance whose weights are part of the fixture
    ac = rnn_ac(obs_space, act_space, 12, 9, 32, (32, 32), (32, 32), torch.nn.ReLU,
                torch.nn.Tanh, torch.nn.Identity, use_gpu=False, rnn_mode="biGRU")
    with torch.no_grad():
        for p_ in ac.parameters():      # make every parameter (biases, log_std, LN) non-trivial
            p_.add_(0.05 * torch.randn_like(p_))
    ac.eval()
    weights = {"w:" + k: v.detach().numpy().copy() for k, v in ac.state_dict().items()}

    B = 96
    obs, cnt = ragged_obs(rng, B)
    obs_t = [torch.as_tensor(o) for o in obs]
    act = torch.as_tensor(np.round(rng.normal(0, 0.5, (B, 3)), 2).astype(np.float32))
    with torch.no_grad():
        pi, logp = ac.pi(obs_t, act)
        v = ac.v(obs_t)
        pi3, _ = ac.pi(obs_t, act, std_factor=1e-3)       # post_train.py std_factor
        single_mu = torch.stack([ac.pi._distribution(o).mean for o in obs_t[:8]])
        single_v = torch.stack([ac.v(o) for o in obs_t[:8]])
    np.savez_compressed(os.path.join(OUT, "policy_rnn_ac.npz"), obs=pad(obs), count=cnt,
                        act=act.numpy(), mu=pi.mean.numpy(), std=pi.stddev.numpy(),
                        logp=logp.numpy(), v=v.numpy(), std_small=pi3.stddev.numpy(),
                        single_mu=single_mu.numpy(), single_v=single_v.numpy(),
                        entropy=pi.entropy().numpy(), **weights)
    keys = sorted(ck["model_state"].keys())
    with open(os.path.join(OUT, "policy_rnn_ac_keys.txt"), "w") as f:
        for k in keys:
            f.write(f"{k} {tuple(ck['model_state'][k].shape)}\n")

    # ---- GAE: multi_PPObuf with path cuts at arbitrary steps, last_val = 0 (the trainer)
    T = 60
    buf = mp.multi_PPObuf((21,), (3,), T, gamma=0.99, lam=0.97)
    rew = np.round(rng.normal(0, 3, T), 3).astype(np.float32)
    val = rng.normal(0, 2, T).astype(np.float32)
    logp_b = rng.normal(-2, 1, T).astype(np.float32)
    acts = rng.normal(0, 1, (T, 3)).astype(np.float32)
    cuts = np.zeros(T, np.uint8)
    cuts[[7, 8, 30, 59]] = 1
    for t in range(T):
        buf.store(np.zeros(21, np.float32), acts[t], rew[t], val[t], logp_b[t])
        if cuts[t]:
            buf.finish_path(0)
    data = buf.get()
    np.savez_compressed(os.path.join(OUT, "ppo_gae.npz"), rew=rew, val=val, cuts=cuts,
                        adv=data["adv"].numpy(), ret=data["ret"].numpy(), gamma=0.99, lam=0.97)

    # ---- losses: compute_loss_pi / compute_loss_v on the checkpoint policy
    adv = torch.as_tensor(rng.normal(0, 1, B).astype(np.float32))
    ret = torch.as_tensor(rng.normal(0, 5, B).astype(np.float32))
    logp_old = logp + torch.as_tensor(rng.normal(0, 0.3, B).astype(np.float32))
    fake = types.SimpleNamespace(ac=ac, clip_ratio=0.2, use_gpu=False)
    d = dict(obs=obs_t, act=act, adv=adv, ret=ret, logp=logp_old)
    ac.zero_grad()
    loss_pi, info = mp.multi_ppo.compute_loss_pi(fake, d)
    loss_pi.backward()
    g_pi = {k: p.grad.clone() for k, p in ac.named_parameters() if p.grad is not None}
    ac.zero_grad()
    loss_v = mp.multi_ppo.compute_loss_v(fake, d)
    loss_v.backward()
    g_v = {k: p.grad.clone() for k, p in ac.named_parameters() if p.grad is not None}
    np.savez_compressed(os.path.join(OUT, "ppo_loss.npz"), adv=adv.numpy(), ret=ret.numpy(),
                        logp_old=logp_old.detach().numpy(), loss_pi=float(loss_pi),
                        kl=info["kl"], ent=info["ent"], cf=info["cf"], loss_v=float(loss_v),
                        g_pi_log_std=g_pi["pi.log_std"].numpy(),
                        g_pi_out=g_pi["pi.net_out.4.weight"].numpy(),
                        g_pi_gru=g_pi["pi.rnn_reader.rnn_net.weight_ih_l0"].numpy(),
                        g_v_out=g_v["v.v_net.4.weight"].numpy(),
                        g_v_gru=g_v["pi.rnn_reader.rnn_net.weight_hh_l0_reverse"].numpy())
    print("wrote policy_rnn_ac.npz, ppo_gae.npz, ppo_loss.npz;", len(keys), "state-dict keys")


if __name__ == "__main__":
    main()
