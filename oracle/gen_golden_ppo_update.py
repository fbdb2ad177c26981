#!/usr/bin/env python3
"""Golden vectors for the reference's own `multi_ppo.update(data_list)` (train/policy/
multi_ppo.py:341-376): per-agent passes in the order of a seeded np.random.shuffle, at most
max_update_num agents, each pass = <= train_pi_iters policy steps (KL check BEFORE the step,
clip_grad_norm_ over ALL ac parameters at 2.0, pi Adam) then train_v_iters value steps (vf Adam;
the reader shared by both optimizers).  Run on CPU in this container with the reference classes
(same stand-ins as oracle/gen_golden_policy.py; no numerics in them).

Output tests/golden/ppo_update.npz: initial weights (w0:*), the per-agent buffers (padded
observations + counts, act, adv, ret, logp), two cases (max_update_num = 10: every agent;
= 2: only the first two of the shuffled order) with the shuffle order, the number of policy
steps each agent took before the KL stop, and the final weights (w1:* / w2:*).
"""
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden_policy as gp  # noqa: E402

OUT = gp.OUT
PI_LR, VF_LR = 4e-4, 1e-3


def main():
    Box = gp.install()
    from policy.policy_rnn_ac import rnn_ac
    import policy.multi_ppo as mp
    from torch.optim import Adam

    obs_space = Box(-np.inf, np.inf, shape=(21,), dtype=np.float32)
    act_space = Box(low=np.array([-1, -1, -1]), high=np.array([1, 1, 1]), dtype=np.float32)
    rng = np.random.default_rng(42)
    N, T = 4, 24                       # agents, steps per agent

    def fresh():
        torch.manual_seed(3)
        ac = rnn_ac(obs_space, act_space, 12, 9, 16, (24, 24), (24, 24), torch.nn.ReLU,
                    torch.nn.Tanh, torch.nn.Identity, use_gpu=False, rnn_mode="biGRU")
        with torch.no_grad():
            for p_ in ac.parameters():
                p_.add_(0.05 * torch.randn_like(p_))
        return ac

    ac0 = fresh()
    w0 = {"w0:" + k: v.detach().numpy().copy() for k, v in ac0.state_dict().items()}

    # per-agent buffers as multi_PPObuf.get() returns them (multi_ppo.py:79-91)
    agents = []
    with torch.no_grad():
        for n in range(N):
            obs, cnt = gp.ragged_obs(rng, T)
            obs_t = [torch.as_tensor(o) for o in obs]
            act = torch.as_tensor(np.round(rng.normal(0, 0.5, (T, 3)), 2).astype(np.float32))
            _, logp = ac0.pi(obs_t, act)
            # old log-probs a little off the current policy, agent 2 far off (early KL stop)
            noise = 0.05 if n != 2 else 0.3
            logp_old = logp + torch.as_tensor(rng.normal(0, noise, T).astype(np.float32))
            if n == 2:
                logp_old = logp_old + 0.04
            if n == 0:
                logp_old = logp_old + 0.065
            agents.append(dict(obs=obs_t, act=act,
                               ret=torch.as_tensor(rng.normal(0, 3, T).astype(np.float32)),
                               adv=torch.as_tensor(rng.normal(0, 1, T).astype(np.float32)),
                               logp=logp_old.clone(), cnt=cnt, pad=gp.pad(obs)))

    out = dict(w0)
    out.update(obs=np.stack([a["pad"] for a in agents]), count=np.stack([a["cnt"] for a in agents]),
               act=np.stack([a["act"].numpy() for a in agents]),
               ret=np.stack([a["ret"].numpy() for a in agents]),
               adv=np.stack([a["adv"].numpy() for a in agents]),
               logp=np.stack([a["logp"].numpy() for a in agents]),
               pi_lr=PI_LR, vf_lr=VF_LR, train_pi_iters=6, train_v_iters=5, target_kl=0.05,
               clip_ratio=0.2, np_seed=11)

    for case, max_update_num in (("1", 10), ("2", 2)):
        ac = fresh()
        steps = []

        class Counting(Adam):  # counts policy steps per agent (observation only)
            def step(self, *a, **k):
                steps[-1] += 1
                return super().step(*a, **k)

        fake = types.SimpleNamespace(
            ac=ac, robot_num=N, max_update_num=max_update_num, train_pi_iters=6, train_v_iters=5,
            target_kl=0.05, clip_ratio=0.2, use_gpu=False,
            pi_optimizer=Counting(ac.pi.parameters(), lr=PI_LR),
            vf_optimizer=Adam(ac.v.parameters(), lr=VF_LR))
        # the policy loop of an agent starts with compute_loss_pi: open its step counter there
        seen = []

        def loss_pi(d, f=fake):
            if not seen or seen[-1] is not d:
                seen.append(d)
                steps.append(0)
            r = mp.multi_ppo.compute_loss_pi(f, d)
            kls.append(round(r[1]["kl"], 4))
            return r

        kls = []
        fake.compute_loss_pi = loss_pi
        fake.compute_loss_v = lambda d, f=fake: mp.multi_ppo.compute_loss_v(f, d)
        np.random.seed(11)
        order = np.arange(N)
        np.random.shuffle(order)       # the order update() will draw (same seed below)
        np.random.seed(11)
        data_list = [dict(obs=a["obs"], act=a["act"], ret=a["ret"], adv=a["adv"], logp=a["logp"])
                     for a in agents]
        mp.multi_ppo.update(fake, data_list)
        out["order" + case] = order.astype(np.int32)
        out["max_update_num" + case] = np.int32(max_update_num)
        out["pi_steps" + case] = np.asarray(steps, np.int32)
        out.update({f"w{case}:" + k: v.detach().numpy().copy() for k, v in ac.state_dict().items()})
        print("case", case, "order", order, "policy steps per updated agent", steps, "kl trace", kls)
    np.savez_compressed(os.path.join(OUT, "ppo_update.npz"), **out)
    print("wrote ppo_update.npz")


if __name__ == "__main__":
    main()
