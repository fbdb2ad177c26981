"""Evaluation rollouts: counterpart of `train/policy/post_train.py` (post_train.policy_test,
:38-128), batched over the E envs of a `BatchedDroneEnv` and reduced on the device.

Reference semantics kept:
  * action = acceler_vel * np.round(model.act(o, std_factor), 2) + drone.vel   (:63-74; the
    float32 product is widened and added to the float64 velocity, and - unlike the trainer's
    glue, multi_ppo.py:205 - NOT rounded again), through the plain `drone_step`;
  * per step the mean of ||vel|| over the drones of the env AFTER the step (:78-80);
  * an episode ends when any drone collided, at `max_ep_len`, or when every drone finished
    (:86: np.max(d) or ep_len == max_ep_len or np.min(finish)); the whole env is reset (:100);
  * an episode with every arrive flag set contributes its length (:89-90: np.min(info));
    it is a success when every drone finished (:104-105);
  * results: success rate over `num_episodes`, mean / std episode length of the "arrived"
    episodes and mean / std of the per-episode mean speed, np.round(., 2) (:116-128), one line
    appended to result_path + result_name.
With E envs in parallel every env contributes the same number of episodes: its first
ceil(num_episodes / E) ones (counting "the first num_episodes episodes to end" over all envs
would favour short episodes - collisions - and bias every statistic).  The statistics are over
those E * ceil(num_episodes / E) episodes; E = 1 is the reference's sequential loop exactly
(pinned by tests/golden/post_train_*.npz, produced by the reference's policy_test itself).
A "math domain error" (env_train=False: the reference's evaluator aborts when two drones
approach inside r + mr, vel_obs3D.py:13) is raised at the end of the step it happens in.
"""
from __future__ import annotations

import numpy as np
import torch


class post_train:
    def __init__(self, env, num_episodes=100, max_ep_len=150, acceler_vel=1.0, render=False,
                 save=False, neighbor_region=4, neighbor_num=5, args=None, **kwargs):
        self.env = env
        self.num_episodes = num_episodes
        self.max_ep_len = max_ep_len
        self.acceler_vel = acceler_vel
        self.render, self.save = render, save  # accepted for signature parity; plotting is out of scope
        self.drone_number = env.N
        self.inf_print = kwargs.get("inf_print", True)
        self.std_factor = kwargs.get("std_factor", 0.001)
        self.nr, self.nm = neighbor_region, neighbor_num
        self.args = args

    # -- policy ---------------------------------------------------------------------------
    def load_policy(self, policy, std_factor=1, policy_dict=False):
        """`policy`: an actor-critic module (rnn_ac / mlp_ac), or the path of a checkpoint
        written by multi_ppo.save_model (state dict under 'model_state', loaded with
        weights_only=True into `self.args.ac`; the reference's full-module pickles
        (post_train.py:143) are not loaded)."""
        if isinstance(policy, (str, bytes)) or hasattr(policy, "__fspath__"):
            ac = getattr(self.args, "ac", None)
            if ac is None:
                raise ValueError("loading a checkpoint needs args.ac (the module to load into)")
            ck = torch.load(policy, map_location=self.env.device, weights_only=True)
            ac.load_state_dict(ck["model_state"], strict=True)
            policy = ac
        policy.eval()

        def get_action(obs, cnt):  # batched model.act(x, std_factor), stays on the device
            a, _, _ = policy.step_tensors((obs, cnt), std_factor)
            return a.float()

        return get_action

    # -- evaluation -------------------------------------------------------------------------
    def policy_test(self, policy_type="drl", policy_path=None, policy_name="policy", result_path=None,
                    result_name="/result.txt", policy=None, policy_dict=False, **_unused):
        env = self.env
        E, N, dev = env.E, env.N, env.device
        act_fn = None
        if policy_type == "drl":
            act_fn = self.load_policy(policy if policy is not None else policy_path,
                                      self.std_factor, policy_dict=policy_dict)
        env.reset()
        obs, cnt = env.observe()
        ep_len = torch.zeros(E, dtype=torch.int64, device=dev)
        ep_ret = torch.zeros(E, dtype=torch.float64, device=dev)
        speed_sum = torch.zeros(E, dtype=torch.float64, device=dev)
        n = sn = 0
        ep_len_list, mean_speed_list, ep_ret_list = [], [], []
        quota = -(-self.num_episodes // E)   # episodes counted per env
        counted = [0] * E
        total = quota * E
        check_domain = not getattr(env, "env_train", True)
        while n < total:
            if act_fn is not None:
                a = act_fn(obs.view(-1, env.W), cnt.view(-1)).view(E, N, 3)
                # np.round(float32, 2) = rint(a * 100) / 100 with a TRUE division (a tensor
                # divisor: torch turns a scalar divisor into a multiplication by 1/100)
                a_inc = torch.round(a * 100.0) / torch.full_like(a, 100.0)
                vel = env.vel                                              # [E, N, 3] float64
                action = (torch.as_tensor(self.acceler_vel, dtype=torch.float32, device=dev) * a_inc).double() + vel
            else:
                action = env.des_vel()
            obs, cnt, rew, done, info, fin = env.step(action)             # plain drone_step
            speed_sum += torch.linalg.vector_norm(env.vel, dim=-1).mean(dim=1)
            ep_ret += rew[:, 0].double()                                   # r[0] (post_train.py:82)
            ep_len += 1
            ended = done.bool().any(dim=1) | (ep_len == self.max_ep_len) | fin.bool().all(dim=1)
            if check_domain:
                # ONE read of the (read-and-clear) error word per step; both bits are acted on here, so
                # nothing is lost to the clear.  With E > 1 the word is not attributed to an env: the
                # whole evaluation aborts, where the reference's sequential loop (E = 1) loses the one
                # episode it was in - evaluate with E = 1 for the reference's exact abort semantics.
                flags = env.error_flags()
                if flags & 2:
                    raise ValueError("math domain error")  # the reference's drone_step raised here
                if flags & 1:
                    raise ValueError("observation contains NaN/Inf")  # ir_gym.py:232-239
            if bool(ended.any()):
                arrived = info.bool().all(dim=1)
                success = fin.bool().all(dim=1)
                idx = torch.nonzero(ended).flatten().tolist()
                el, sp, er = ep_len.cpu().numpy(), (speed_sum / ep_len.double()).cpu().numpy(), ep_ret.cpu().numpy()
                ar, su = arrived.cpu().numpy(), success.cpu().numpy()
                for e in idx:
                    if counted[e] >= quota:
                        continue  # this env has delivered its share; it keeps stepping, uncounted
                    counted[e] += 1
                    if ar[e]:
                        ep_len_list.append(int(el[e]))
                    if self.inf_print:
                        print("%s, Episode %d \t EpRet %.3f \t EpLen %d \t EpSpeed  %.3f"
                              % ("Successful" if ar[e] else "Fail", n, er[e], el[e], sp[e]))
                    ep_ret_list.append(float(er[e]))
                    mean_speed_list.append(float(sp[e]))
                    n += 1
                    sn += int(su[e])
                env.reset(ended)
                obs, cnt = env.observe()
                z = torch.zeros_like(ep_len)
                ep_len = torch.where(ended, z, ep_len)
                ep_ret = torch.where(ended, torch.zeros_like(ep_ret), ep_ret)
                speed_sum = torch.where(ended, torch.zeros_like(speed_sum), speed_sum)
        mean_len = 0 if not ep_len_list else np.round(np.mean(ep_len_list), 2)
        std_len = 0 if not ep_len_list else np.round(np.std(ep_len_list), 2)
        average_speed = np.round(np.mean(mean_speed_list), 2)
        std_speed = np.round(np.std(mean_speed_list), 2)
        line = ("policy_name: " + policy_name + "  successful rate: {:.2%}".format(sn / total)
                + " average EpLen: %s std length %s average speed: %s std speed %s"
                % (mean_len, std_len, average_speed, std_speed))
        if result_path is not None:
            with open(result_path + result_name, "a") as f:
                print(line, file=f)
        if self.inf_print:
            print(line)
        return dict(success_rate=sn / total, mean_len=float(mean_len), std_len=float(std_len),
                    average_speed=float(average_speed), std_speed=float(std_speed),
                    episodes=n, ep_ret=ep_ret_list, ep_len=ep_len_list, speed=mean_speed_list)
