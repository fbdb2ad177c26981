"""MA-PPO trainer of the reference (train/policy/multi_ppo.py), batched over E x N.

Same constructor arguments, same method names (`training_loop`, `update`,
`compute_loss_pi`, `compute_loss_v`, `save_model`) and the same checkpoint layout
(`model_state`, `pi_optimizer`, `vf_optimizer`, multi_ppo.py:411-412).  What changes
is the shape of the data: the reference keeps one python buffer per drone and steps
one env; here a rollout is `[T, E, N, ...]` tensors on the GPU, produced by
`BatchedDroneEnv.step_policy` (the env step, the trainer's action glue and the
reset protocol in one HIP launch per step).

Path-cut rules of the reference (multi_ppo.py:226-281), per env:
  * a collision resets the collided drone and does NOT end the path;
  * when any drone finishes or exceeds max_ep_len, or the epoch ends, the path of
    EVERY drone of that env ends with bootstrap value 0 (`finish_path(0)` sits
    outside the `if`, :279);
  * advantages are not normalised.
GAE follows multi_PPObuf.finish_path (:68-77) in float64 (np.append promotes the
float32 buffers) and is stored as float32.

Update modes (`update`, multi_ppo.py:341-376):
  * reference order (`reference_order=True`, or `update(data_list)` with the reference's own
    per-agent list): the agents are visited in the order of a seeded np.random.shuffle, at
    most `max_update_num` of them; each visit is <= train_pi_iters policy steps (KL check
    BEFORE the step) followed by train_v_iters value steps on that agent's samples alone.
    With E envs an "agent" is drone n of every env.  Pinned by tests/golden/ppo_update.npz,
    produced by the reference's update() itself.
  * pooled (default, the fast path): ONE pass over all T*E*N samples (optionally in
    minibatches).  A deliberate deviation: the per-agent sequential passes are up to
    max_update_num * (train_pi_iters + train_v_iters) optimizer steps on N-times smaller
    batches.  `max_update_num` has no meaning here; passing it without reference_order warns.

Multi-GPU: envs are sharded one process per GPU with EQUAL shards (checked at
construction: unequal shards would run different numbers of minibatches and hang in the
collective); the only collectives are one all-reduce of the flattened gradient bucket per
optimizer step and the mean of the KL estimate (so every rank leaves the policy loop
together); rank 0 alone writes checkpoints.
"""
from __future__ import annotations

import os
import time
import warnings

import numpy as np
import torch
from torch.optim import Adam


def gae_scan(rew, val, cut, gamma=0.99, lam=0.97):
    """GAE over time, all columns at once.  rew, val: float [T, ...]; cut: bool [T, ...], True
    where the path ends after step t (bootstrap 0; the end of the buffer ends every path).
    Returns (adv, ret) float32.

    multi_PPObuf.finish_path (multi_ppo.py:68-77) per path:
        deltas = r_t + gamma * V_{t+1} - V_t,  V_end = 0
        adv    = discount_cumsum(deltas, gamma * lam)
        ret    = discount_cumsum(rews + [0], gamma)[:-1]
    A discounted sum to the end of the path is a difference of two reverse prefix sums:
        sum_{s = t .. end(t)} c^(s - t) x_s = (C_t - C_{end(t) + 1}) / c^t,  C_t = sum_{s >= t} c^s x_s
    in float64 (np.append promotes the reference's float32 buffers as well), so the whole scan is a
    dozen tensor ops instead of a Python loop over T.  Terms of C_t carry weights <= c^t, so the
    difference loses nothing to cancellation; c^T stays far above the float64 underflow for any
    buffer length in use (0.96^4000 = 1e-71).  `gae_scan_loop` is the step-by-step form it is
    tested against."""
    T = rew.shape[0]
    dev = rew.device
    r, v = rew.to(torch.float64), val.to(torch.float64)
    cutb = cut.to(torch.bool).clone()
    cutb[T - 1] = True
    keep = (~cutb).to(torch.float64)
    v_next = torch.cat([v[1:], torch.zeros_like(v[:1])], dim=0) * keep   # V_{t+1}, 0 behind a cut
    delta = r + gamma * v_next - v
    # end(t) + 1 = index behind the first cut at or after t
    shape1 = (T,) + (1,) * (rew.dim() - 1)
    idx = torch.arange(T, device=dev).view(shape1).expand_as(cutb)
    nxt = torch.where(cutb, idx + 1, torch.full_like(idx, T))
    end1 = torch.flip(torch.cummin(torch.flip(nxt, [0]), dim=0).values, [0])  # [T, ...] in 1..T

    def disc(x, c):
        w = torch.pow(torch.full((T,), float(c), dtype=torch.float64, device=dev),
                      torch.arange(T, dtype=torch.float64, device=dev)).view(shape1)
        C = torch.flip(torch.cumsum(torch.flip(x * w, [0]), dim=0), [0])
        C = torch.cat([C, torch.zeros_like(C[:1])], dim=0)                   # C_T = 0
        return (C[:-1] - torch.gather(C, 0, end1)) / w

    return disc(delta, gamma * lam).to(torch.float32), disc(r, gamma).to(torch.float32)


def gae_scan_loop(rew, val, cut, gamma=0.99, lam=0.97):
    """The same scan, one step at a time (reference form; tests compare gae_scan with it)."""
    T = rew.shape[0]
    r, v = rew.to(torch.float64), val.to(torch.float64)
    keep = (~cut.to(torch.bool)).to(torch.float64)
    adv, ret = torch.empty_like(r), torch.empty_like(r)
    nxt_v = torch.zeros_like(r[0])
    nxt_a = torch.zeros_like(r[0])
    nxt_r = torch.zeros_like(r[0])
    for t in range(T - 1, -1, -1):
        k = keep[t]
        delta = r[t] + gamma * (k * nxt_v) - v[t]
        nxt_a = delta + (gamma * lam) * (k * nxt_a)
        nxt_r = r[t] + gamma * (k * nxt_r)
        adv[t], ret[t] = nxt_a, nxt_r
        nxt_v = v[t]
    return adv.to(torch.float32), ret.to(torch.float32)


class RolloutBuffer:
    """[T, E, N, ...] storage on the device (the batched multi_PPObuf, multi_ppo.py:39-94)."""

    def __init__(self, T, E, N, obs_width, act_dim, device, gamma=0.99, lam=0.95):
        f32 = dict(dtype=torch.float32, device=device)
        # T + 1 observation slots: slot t is what the policy saw at step t; the env writes the
        # observation after step t straight into slot t + 1 (no copies in the rollout loop)
        self.obs = torch.zeros((T + 1, E, N, obs_width), **f32)
        self.cnt = torch.zeros((T + 1, E, N), dtype=torch.int32, device=device)
        self.act = torch.zeros((T, E, N, act_dim), **f32)
        self.rew = torch.zeros((T, E, N), **f32)
        self.val = torch.zeros((T, E, N), **f32)
        self.logp = torch.zeros((T, E, N), **f32)
        self.cut = torch.zeros((T, E), dtype=torch.bool, device=device)
        self.gamma, self.lam, self.ptr, self.T = gamma, lam, 0, T

    def store(self, obs, cnt, act, rew, val, logp):
        assert self.ptr < self.T  # multi_ppo.py:59
        t = self.ptr
        if obs is not None and obs.data_ptr() != self.obs[t].data_ptr():
            self.obs[t].copy_(obs); self.cnt[t].copy_(cnt)
        self.act[t].copy_(act)
        self.rew[t].copy_(rew); self.val[t].copy_(val); self.logp[t].copy_(logp)
        self.ptr += 1

    def finish_path(self, env_mask):
        """finish_path(0) for every drone of the masked envs at the last stored step."""
        self.cut[self.ptr - 1] |= env_mask

    def get(self):
        assert self.ptr == self.T  # buffer has to be full (multi_ppo.py:80)
        cut = self.cut.unsqueeze(-1).expand_as(self.rew)
        adv, ret = gae_scan(self.rew, self.val, cut, self.gamma, self.lam)
        self.ptr = 0
        self.cut.zero_()
        flat = lambda x: x.reshape((-1,) + x.shape[3:])
        return dict(obs=flat(self.obs[:self.T]), cnt=flat(self.cnt[:self.T]), act=flat(self.act),
                    ret=flat(ret), adv=flat(adv), logp=flat(self.logp),
                    shape=tuple(self.rew.shape))  # (T, E, N): rows are ordered t, e, n


class multi_ppo:
    def __init__(self, env, ac_policy, pi_lr=3e-4, vf_lr=1e-3, train_epoch=50,
                 steps_per_epoch=600, max_ep_len=300, gamma=0.99, lam=0.97, clip_ratio=0.2,
                 train_pi_iters=100, train_v_iters=100, target_kl=0.01, render=False,
                 render_freq=20, con_train=False, seed=7, save_freq=50, save_figure=False,
                 save_path="test/", save_name="test", load_fname=None, use_gpu=True,
                 save_result=False, counter=0, test_env=None, lr_decay_epoch=1000,
                 max_update_num=None, mpi=False, figure_save_path=None, minibatch_size=None,
                 dist=None, sanitize_rewards=True, amp=False, reference_order=False, fused_rollout=True,
                 rollout_chunk=None, tune_gemms=True, tune_update=False, fused_mlp=True, graph_rollout=False, **kwargs):
        np.random.seed(seed)
        self.env, self.ac, self.dist = env, ac_policy, dist
        # The agent order of the reference-order update comes from a generator of its own, seeded like
        # the reference's global one (multi_ppo.py:104: np.random.seed(seed); a RandomState(seed) draws
        # the very same sequence): every rank MUST visit the agents in the same order, whatever else in
        # the process uses np.random.  The action noise, on the other hand, has to differ between the
        # shards: rank r samples from torch's generator seeded seed + r (rank 0 = the reference's seed).
        self._order_rng = np.random.RandomState(seed)
        self.fused_rollout = bool(fused_rollout)
        # bf16 rollouts of the MLP(256, 256) actor-critic: the policy step as ONE matrix-core kernel
        # (rvo3d_policy_mlp_sample) instead of cast + three library GEMMs + rvo3d_policy_sample
        self.fused_mlp = bool(fused_mlp)
        # the fast paths' per-step launches replayed as HIP graphs from the second rollout on (see _collect_fused);
        # opt-in: measured at 64 x 4096, 0.158 ms per step with and 0.157-0.161 without - the gaps between the dependent
        # kernels of a graph are what they are between stream launches
        self.graph_rollout = bool(graph_rollout)
        self.rollout_chunk = rollout_chunk  # rows per policy pass of the fused rollout (None / 0: all at once)
        # the rollout's policy GEMMs ([E*N, 128] x [128, 512], [E*N, 256] x [256, 256], bf16) through PyTorch's
        # TunableOp: the first call of a shape times hipBLASLt's candidate kernels (<= 3 s per shape) and keeps
        # the fastest - at 64 x 4096 a 256x256x64 stream-K kernel, 70 us, instead of the heuristic's 84 us
        self.tune_gemms = bool(tune_gemms)
        # the same for the update's GEMMs (a dozen shapes incl. the tall-skinny weight-gradient products): measured
        # at 64 x 4096, T = 16, fp32: 0.285 -> 0.214 s per (2 + 2)-iteration update, for ~50 s of tuning in the first
        # update of the process - worth it for a training run, not for a benchmark: opt-in
        self.tune_update = bool(tune_update)
        # key of the counter-based action noise of the fused rollout (rvo3d_policy_sample): per rank
        self._sample_seed = (int(seed) * 0x9E3779B97F4A7C15 + 0x1234567 * (
            dist.get_rank() if (dist is not None and dist.is_initialized()) else 0)) & 0xFFFFFFFFFFFFFFFF
        torch.manual_seed(seed + (dist.get_rank() if (dist is not None and dist.is_initialized()) else 0))
        self.E, self.N = env.E, env.N
        self.robot_num = env.N  # env.ir_gym.drone_num (multi_ppo.py:110)
        self.device = env.device
        # the shared reader sits in both optimizers, as in the reference (multi_ppo.py:115-116)
        pi_params = self.ac.pi.parameters() if hasattr(self.ac.pi, "parameters") else \
            list(self.ac.pi_net.parameters()) + [self.ac.log_std]
        v_params = self.ac.v.parameters() if hasattr(self.ac.v, "parameters") else \
            self.ac.v_net.parameters()
        self.pi_optimizer = Adam(pi_params, lr=pi_lr)
        self.vf_optimizer = Adam(v_params, lr=vf_lr)
        if con_train and load_fname:  # resume loads the weights only (multi_ppo.py:118-121)
            ck = torch.load(load_fname, map_location=self.device, weights_only=True)
            self.ac.load_state_dict(ck["model_state"], strict=True)
            self.ac.train()
        self.epoch, self.max_ep_len, self.steps_per_epoch = train_epoch, max_ep_len, steps_per_epoch
        self.clip_ratio, self.train_pi_iters = clip_ratio, train_pi_iters
        self.train_v_iters, self.target_kl = train_v_iters, target_kl
        self.save_freq, self.save_path, self.save_name = save_freq, save_path, save_name
        self.use_gpu, self.minibatch_size = use_gpu, minibatch_size
        self.reference_order = bool(reference_order)
        if max_update_num is not None and not self.reference_order:
            warnings.warn("max_update_num only applies to the reference-order update "
                          "(reference_order=True or update(data_list)); the pooled update ignores it")
        self.max_update_num = 10 if max_update_num is None else int(max_update_num)  # multi_ppo.py:101
        self._flat = None  # multi-rank gradient bucket (see _bucket)
        self._check_equal_shards()
        # The reference's RVO reward is inf / nan while a drone sits within 0.4 m of its
        # waypoint (ir_gym.py:88, survey Q9) and would poison GAE; by default such rewards
        # enter the buffer as 0 (set sanitize_rewards=False for the literal behaviour).
        self.sanitize_rewards = sanitize_rewards
        # amp=True runs the policy GEMMs of the rollout in bf16 (MFMA); the reference is fp32
        self.amp = amp
        self.nonfinite_rewards = 0
        self.buf = RolloutBuffer(steps_per_epoch, self.E, self.N, env.W, 3, self.device, gamma, lam)
        self.ep_len = torch.zeros((self.E, self.N), dtype=torch.int32, device=self.device)
        self.ep_ret = torch.zeros((self.E, self.N), dtype=torch.float32, device=self.device)
        self.log = []

    # ---- rollout ------------------------------------------------------------------------
    def _fused_mode(self):
        """How a rollout step runs on the GPU: "mlp" - config 3's MLP(256, 256) actor-critic in reduced precision:
        the whole policy step (cast, hidden layers, heads, sampling, stores) is ONE kernel on the matrix cores,
        rvo3d_policy_mlp_sample; "heads" - other MLP actor-critics (and float32): library GEMMs up to the last
        hidden layers, everything from there on in rvo3d_policy_sample; "rnn0" - the reference's biGRU actor-critic with
        (256, 256) heads in reduced precision: rows without a velocity-obstacle row (nearly all) through the collapsed
        first layer, rvo3d_reader_zero_features + rvo3d_policy_mlp_sample, the others through the "heads" / "direct" path
        on a gathered batch; "direct" - any other actor-critic with the reference's surface
        (`ac.pi._distribution`, `ac.v`, e.g. the biGRU rnn_ac): its own forward gives mu and v, the kernel
        samples / rounds / stores; None - the module path of collect() (CPU, or fused_rollout=False)."""
        if self.device.type != "cuda" or not self.fused_rollout:
            return None
        if self.amp and self.fused_mlp and hasattr(self.ac, "mlp_blob") and self.ac.mlp_blob() is not None:
            return "mlp"
        if (self.amp and self.fused_mlp and not getattr(self, "_rnn0_dense", False) and hasattr(self.ac, "zero_vo_plan")
                and self.ac.zero_vo_plan() is not None and self.ac.zero_vo_plan()["rows_net"] is not None):
            return "rnn0"
        if hasattr(self.ac, "fused_plan"):
            plan = self.ac.fused_plan(torch.bfloat16 if self.amp else torch.float32)
            if plan is not None and plan["hidden"] in ((256, 512, 1024) if self.amp else (128, 256, 512, 1024)):
                return "heads"
        if hasattr(self.ac, "dist") or hasattr(getattr(self.ac, "pi", None), "_distribution"):
            return "direct"
        return None

    def _fused_ok(self):
        return self._fused_mode() is not None

    def _mu_v(self, obs, cnt):
        """mu (after the output activation) and v of the caller's own network, float32 [B, 3] / [B]."""
        with torch.no_grad(), torch.autocast(device_type="cuda", dtype=torch.bfloat16, enabled=self.amp):
            if hasattr(self.ac, "dist"):            # mlp_ac with a shape the heads kernel has no instantiation for
                mu, v = self.ac.dist(obs).mean, self.ac.v(obs)
            else:                                   # rnn_ac (policy_rnn_ac.py:57-69)
                pi, vf = self.ac.pi, self.ac.v
                if getattr(pi, "rnn_reader", None) is not None and pi.rnn_reader is getattr(vf, "rnn_reader", None):
                    # actor and critic share ONE reader instance (policy_rnn_ac.py:46-54): read once
                    feat = pi.rnn_reader.forward_batch(obs, torch.clamp(cnt.to(torch.int64), min=1))
                    mu, v = pi.net_out(feat), vf.v_net(feat).squeeze(-1)
                else:
                    mu, v = pi._distribution((obs, cnt)).mean, vf((obs, cnt))
        return mu.float().contiguous(), v.float().reshape(-1).contiguous()

    def _tuned_gemms(self):
        """Context: TunableOp on for the GEMMs issued inside (and back to its previous state after)."""
        import contextlib
        if not self.tune_gemms or self.device.type != "cuda":
            return contextlib.nullcontext()

        @contextlib.contextmanager
        def ctx():
            import torch.cuda.tunable as tun
            was = tun.is_enabled()
            if not getattr(self, "_tun_set", False):
                tun.set_max_tuning_duration(3000)  # ms per GEMM shape, once per process (two shapes in a rollout)
                tun.set_max_tuning_iterations(100)
                if hasattr(tun, "write_file_on_exit"):
                    tun.write_file_on_exit(False)  # results stay in the process
                else:  # this PyTorch writes its results at exit: into the temp dir, not the cwd
                    import tempfile
                    tun.set_filename(os.path.join(tempfile.gettempdir(), f"rvo3d_tunableop_{os.getpid()}.csv"))
                self._tun_set = True
            tun.enable(True)
            try:
                yield
            finally:
                tun.enable(was)
        return ctx()

    def _collect_fused(self, final_reset=True):
        """collect() with the per-step glue on the device: per step ONE cast of the observation (bf16
        rollouts), the policy GEMMs up to the last hidden layers (mlp_ac.hidden_pair), rvo3d_policy_sample
        (heads, tanh, sample, log-probability, np.round(a, 2), the three buffer stores), the env step from
        the stored action, rvo3d_rollout_account (reward slot, episode counters, path cuts, who still
        needs a reset).  Same semantics as collect(): multi_ppo.py:183-281."""
        import ctypes as C
        from .. import _lib
        env, buf, L = self.env, self.buf, _lib.lib()
        E, N, T = self.E, self.N, self.steps_per_epoch
        dt = torch.bfloat16 if self.amp else torch.float32
        cur_obs, cur_cnt = getattr(self, "_cur", (env.obs, env.vo_count))
        buf.obs[0].copy_(cur_obs); buf.cnt[0].copy_(cur_cnt)
        if getattr(self, "_acct", None) is None:
            self._acct = dict(sums=torch.zeros((E, 2), dtype=torch.float64, device=self.device),
                              any_extra=torch.zeros(1, dtype=torch.int32, device=self.device),
                              extra=torch.zeros((E, N), dtype=torch.uint8, device=self.device),
                              cut=torch.zeros((T, E), dtype=torch.uint8, device=self.device),
                              step=0)
        ac = self._acct
        if ac["cut"].shape[0] != T:
            ac["cut"] = torch.zeros((T, E), dtype=torch.uint8, device=self.device)
        ac["sums"].zero_(); ac["cut"].zero_()
        p = lambda t: C.c_void_p(t.data_ptr())
        stream = lambda: C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        since_full_reset = 0
        mode = self._fused_mode()
        log_std = self.ac.log_std
        mb = self.ac.mlp_blob() if mode == "mlp" else None  # (once per rollout: the weights do not change inside it)
        zp = self.ac.zero_vo_plan() if mode == "rnn0" else None
        if mode == "rnn0" and "vo_count" in ac:
            ac["vo_count"].zero_()  # (the kernels leave it at zero; a rollout that was interrupted half-way may not have)
        # Graph replay (graph_rollout=True; the MLP and biGRU fast paths): the launches of step t - every argument by value,
        # every buffer slot at a fixed address - are captured once per slot into a HIP graph and replayed in later
        # epochs (the loop's wall time exceeds its GPU time by the dispatch gaps between three to five dependent
        # launches; measured: a graph's kernels keep those gaps - no gain, hence opt-in).  The noise counter then lives in device memory
        # (rvo3d_rollout_set_step_counter: rvo3d_rollout_account advances it).  Only when no episode can time out
        # inside the rollout (that check reads the device) and after one eager rollout (allocations, first calls).
        use_graph = (self.graph_rollout and mode in ("mlp", "rnn0") and T <= self.max_ep_len and not self.rollout_chunk
                     and getattr(self, "_graph_warm", None) == mode and not getattr(self, "_graph_failed", False))
        self._graph_warm = mode
        if use_graph:
            if getattr(self, "_step_dev", None) is None:
                self._step_dev = torch.zeros(1, dtype=torch.int64, device=self.device)
                self._graphs = {}
            _lib.check(L.rvo3d_rollout_set_step_counter(p(self._step_dev)), "rvo3d_rollout_set_step_counter")
        step_arg = lambda: (1 << 32) if (use_graph and not getattr(self, "_graph_failed", False)) else ac["step"]

        def launches(t, epoch_ended):
            x = buf.obs[t].view(E * N, env.W)
            act_t, logp_t, val_t = buf.act[t].view(E * N, 3), buf.logp[t].view(E * N), buf.val[t].view(E * N)
            if mode == "mlp":
                # (the env's counts: column groups that are zero for all rows of a wave are skipped)
                _lib.check(L.rvo3d_policy_mlp_sample(p(mb["blob"]), env.W, p(x), x.stride(0), E * N,
                                                     p(buf.cnt[t]), 12, 9, 1 if mb["tanh"] else 0, p(log_std), 1.0, self._sample_seed,
                                                     step_arg(), p(act_t), p(logp_t), p(val_t), None, None, stream()),
                           "rvo3d_policy_mlp_sample")
                ac["step"] += 1
            if mode == "rnn0":
                # rows without a velocity-obstacle row: collapsed first layer (rvo3d_reader_zero_features builds its 20
                # inputs and lists the rows that do have VO rows) + the MFMA kernel; the listed rows: one workgroup each,
                # as the modules compute them; no host synchronisation
                if "feat0" not in ac or ac["feat0"].shape != (E * N, zp["width"]):
                    ac["feat0"] = torch.empty((E * N, zp["width"]), dtype=torch.float32, device=self.device)
                    ac["vo_list"] = torch.zeros(E * N, dtype=torch.int32, device=self.device)
                    ac["vo_count"] = torch.zeros(2, dtype=torch.int32, device=self.device)   # [count, finished workgroups]
                f0, cnt_t = ac["feat0"], buf.cnt[t]
                _lib.check(L.rvo3d_reader_zero_features(p(x), x.stride(0), E * N, zp["state_dim"], zp["feat_dim"],
                                                        p(zp["ln_w"]), p(zp["ln_b"]), zp["sum_h0"], zp["sumsq_h0"],
                                                        zp["eps"], p(f0), f0.stride(0), p(cnt_t), p(ac["vo_list"]),
                                                        p(ac["vo_count"]), stream()), "rvo3d_reader_zero_features")
                _lib.check(L.rvo3d_policy_mlp_sample(p(zp["blob"]), zp["width"], p(f0), f0.stride(0), E * N, None, 0, 0,
                                                     1 if zp["tanh"] else 0, p(log_std), 1.0, self._sample_seed,
                                                     step_arg(), p(act_t), p(logp_t), p(val_t), None, None, stream()),
                           "rvo3d_policy_mlp_sample")
                net = zp["rows_net"]
                net.slots = env.nm if hasattr(env, "nm") else (env.W - zp["state_dim"]) // 9
                _lib.check(L.rvo3d_policy_rows(C.byref(net), p(x), x.stride(0), p(cnt_t), p(ac["vo_list"]),
                                               p(ac["vo_count"]), C.c_void_p(ac["vo_count"].data_ptr() + 4),
                                               1 if zp["tanh"] else 0, p(log_std), 1.0, self._sample_seed, step_arg(),
                                               p(act_t), p(logp_t), p(val_t), stream()), "rvo3d_policy_rows")
                ac["step"] += 1
            if mode == "direct":
                mu, v = self._mu_v(x, buf.cnt[t].view(E * N))
                hd = _lib.PolicyHeads(mu.data_ptr(), v.data_ptr(), mu.stride(0), 1, _lib.RVO3D_F32, 0, 0, 0,
                                      None, None, None, None, log_std.data_ptr())
                _lib.check(L.rvo3d_policy_sample(C.byref(hd), E * N, 1.0, self._sample_seed, ac["step"],
                                                 p(act_t), p(logp_t), p(val_t), None, None, stream()),
                           "rvo3d_policy_sample")
                ac["step"] += 1
                del mu, v
            plan = self.ac.fused_plan(dt) if mode == "heads" else None
            # (rollout_chunk: the policy can run over the rows in chunks whose activations stay in the 256 MiB
            # Infinity Cache.  Measured at 64 x 4096, bf16: no gain - 422 / 420 / 427 / 555 us per step for all /
            # 131072 / 65536 / 32768 rows per pass; the [rows, 256] x [256, 256] GEMMs take 20 us per 65536 rows
            # either way: they are not bound by HBM.  Off by default.)
            B = E * N
            Cn = B if not self.rollout_chunk else min(B, int(self.rollout_chunk))
            for r0 in (range(0, B, Cn) if mode == "heads" else ()):
                n = min(Cn, B - r0)
                # the first layer's A operand: mlp_ac - the observation cast into a zero-padded buffer (ONE kernel);
                # rnn_ac - the reader's features (rvo3d_reader_first_step + the few rows with several VO rows)
                xc = self.ac.prepare_input(x[r0:r0 + n], buf.cnt[t].view(E * N)[r0:r0 + n], plan, ac)
                with torch.no_grad(), self._tuned_gemms():
                    hp, hv = self.ac.hidden_pair(xc, plan)
                hd = _lib.PolicyHeads(hp.data_ptr(), hv.data_ptr(), hp.stride(0), hv.stride(0),
                                      _lib.RVO3D_BF16 if dt == torch.bfloat16 else _lib.RVO3D_F32, plan["hidden"],
                                      1 if plan["tanh"] else 0, 0, plan["w_pi"].data_ptr(), plan["b_pi"].data_ptr(),
                                      plan["w_v"].data_ptr(), plan["b_v"].data_ptr(), self.ac.log_std.data_ptr())
                # (the generator's counter is (row of the chunk, call number): every call draws fresh noise)
                _lib.check(L.rvo3d_policy_sample(C.byref(hd), n, 1.0, self._sample_seed, ac["step"],
                                                 p(act_t[r0:]), p(logp_t[r0:]), p(val_t[r0:]), None, None, stream()),
                           "rvo3d_policy_sample")
                ac["step"] += 1
                del hp, hv
            # the env steps from the stored (rounded) action: rounding twice is rounding once
            env.step_policy(buf.act[t], autoreset=True, obs_out=buf.obs[t + 1], cnt_out=buf.cnt[t + 1])
            _lib.check(L.rvo3d_rollout_account(E, N, p(env.reward), p(env.done), p(env.finish),
                                               1 if self.sanitize_rewards else 0, int(self.max_ep_len),
                                               1 if epoch_ended else 0, p(buf.rew[t]), p(self.ep_ret), p(self.ep_len),
                                               p(ac["cut"][t]), p(ac["extra"]), p(ac["sums"]), p(ac["any_extra"]),
                                               stream()), "rvo3d_rollout_account")

        def one_step(t, epoch_ended):
            if not use_graph or getattr(self, "_graph_failed", False):
                return launches(t, epoch_ended)
            key = (mode, t, bool(epoch_ended), T, buf.obs.data_ptr(), buf.act.data_ptr(), buf.logp.data_ptr(),
                   buf.val.data_ptr(), buf.rew.data_ptr(), buf.cnt.data_ptr(), ac["cut"].data_ptr(),
                   (mb or zp)["blob"].data_ptr(), env.obs.data_ptr(), id(env))
            g = self._graphs.get(key)
            if g is None:
                if len(self._graphs) > 4 * T + 8:
                    self._graphs.clear()  # (buffers were replaced: the old slots' graphs are dead weight)
                try:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        launches(t, epoch_ended)
                    self._graphs[key] = g
                except Exception as ex:  # capture is a courtesy of the runtime: without it the loop runs as before
                    warnings.warn(f"rollout graph capture failed ({type(ex).__name__}: {ex}); continuing without graphs")
                    self._graph_failed = True
                    self._graphs.clear()
                    _lib.check(L.rvo3d_rollout_set_step_counter(None), "rvo3d_rollout_set_step_counter")
                    return launches(t, epoch_ended)
            g.replay()

        for t in range(T):
            epoch_ended = final_reset and t == T - 1
            one_step(t, epoch_ended)
            since_full_reset += 1
            buf.ptr += 1
            # only now can an episode have timed out (no episode is longer than the steps since the last
            # full reset): before that the device is not asked (no synchronisation in the loop)
            # (any_extra is sticky: the kernel only ever sets it, and it is cleared here once its drones are
            # handled - one launch per step less than clearing it before every step; a timeout cannot occur, and
            # the flag cannot be set, before since_full_reset exceeds max_ep_len or the epoch ends)
            if epoch_ended or (since_full_reset > self.max_ep_len and int(ac["any_extra"].item()) != 0):
                env.reset_drones(ac["extra"])
                env.observe(obs_out=buf.obs[t + 1], cnt_out=buf.cnt[t + 1])
                ac["any_extra"].zero_()
        if use_graph:
            _lib.check(L.rvo3d_rollout_set_step_counter(None), "rvo3d_rollout_set_step_counter")
        buf.cut[:T] |= ac["cut"].bool()
        self._cur = (buf.obs[T], buf.cnt[T])
        if mode == "rnn0":
            # the one-workgroup-per-row kernel is for short lists: a world where more than 1 row in 500 has VO rows
            # (~500 rows per step at 64 x 4096: ~0.1 ms of that kernel) goes back to the library-GEMM path
            self._rnn0_dense = float((buf.cnt[:T] > 0).float().mean()) > 2e-3
        s = ac["sums"].sum(dim=0).tolist()
        return float(s[0] / max(s[1], 1.0))

    def collect(self, final_reset=True):
        """One epoch of steps_per_epoch env steps (multi_ppo.py:183-281), on the device.  The
        env writes each observation straight into the next buffer slot; the policy GEMMs run
        under ONE autocast region (weight casts are cached across the steps).
        final_reset=False (rollout_profile only) leaves out the epoch-end full reset."""
        if self._fused_ok():
            return self._collect_fused(final_reset)
        env, buf = self.env, self.buf
        cur_obs, cur_cnt = getattr(self, "_cur", (env.obs, env.vo_count))
        buf.obs[0].copy_(cur_obs); buf.cnt[0].copy_(cur_cnt)
        ret_sum = torch.zeros((), device=self.device)
        ret_n = torch.zeros((), device=self.device)
        since_full_reset = 0  # no episode can be longer than this: timeouts need no device check before
        hundred = torch.full((self.E, self.N, 3), 100.0, dtype=torch.float32, device=self.device)
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16, enabled=self.amp):
            for t in range(self.steps_per_epoch):
                obs, cnt = buf.obs[t], buf.cnt[t]
                a, v, logp = self.ac.step_tensors((obs.view(-1, env.W), cnt.view(-1)))
                a = a.float().view(self.E, self.N, 3)
                # a_inc = round(a, 2); abs = round(acceler * a_inc + vel, 2); drone_step; resets of
                # done|finish drones + env_observation: one launch (multi_ppo.py:196-242)
                _, _, rew, done, info, fin = env.step_policy(a, autoreset=True, obs_out=buf.obs[t + 1],
                                                             cnt_out=buf.cnt[t + 1])
                # what the reference stores (multi_ppo.py:197): rint(a * 100) / 100 in float32 with a
                # true division (a scalar divisor would become a multiplication by 1/100); written
                # straight into the buffer slot, like everything else of this step
                torch.div(torch.round(a * 100.0), hundred, out=buf.act[t])
                rew_fin = torch.nan_to_num(rew, nan=0.0, posinf=0.0, neginf=0.0)
                buf.rew[t].copy_(rew_fin if self.sanitize_rewards else rew)
                buf.val[t].copy_(v.view(self.E, self.N))
                buf.logp[t].copy_(logp.view(self.E, self.N))
                buf.ptr += 1
                self.ep_ret += rew_fin
                self.ep_len += 1
                since_full_reset += 1
                epoch_ended = final_reset and t == self.steps_per_epoch - 1
                reset_by_step = (done | fin) != 0        # what the fused step already reset
                if epoch_ended:                          # full reset (multi_ppo.py:244-264)
                    ended = torch.ones_like(reset_by_step)
                    terminal = torch.ones(self.E, dtype=torch.bool, device=self.device)
                    extra = ~reset_by_step
                elif since_full_reset > self.max_ep_len:  # only now can an episode time out
                    timeout = self.ep_len > self.max_ep_len
                    ended = reset_by_step | timeout
                    terminal = ((fin != 0) | timeout).any(dim=1)  # any drone of the env (multi_ppo.py:229)
                    extra = timeout & ~reset_by_step
                else:
                    ended = reset_by_step
                    terminal = (fin != 0).any(dim=1)
                    extra = None
                ret_sum += (self.ep_ret * ended).sum()
                ret_n += ended.sum()
                if epoch_ended or (extra is not None and bool(extra.any())):
                    env.reset_drones(extra)
                    env.observe(obs_out=buf.obs[t + 1], cnt_out=buf.cnt[t + 1])
                buf.finish_path(terminal)
                self.ep_ret.masked_fill_(ended, 0.0)
                self.ep_len.masked_fill_(ended, 0)
        self._cur = (buf.obs[self.steps_per_epoch], buf.cnt[self.steps_per_epoch])
        return float(ret_sum / ret_n.clamp(min=1))

    def rollout_profile(self, steps=4):
        """What one rollout step costs on the device, for bench.py's `rollout` block: GPU kernel launches
        per step (torch.profiler over `steps` steps of collect()) and the mean time of the env kernel in
        them.  Returns {"launches_per_step", "env_kernel_us", "gpu_busy_ms_per_step"}; values are None when
        the profiler is unavailable (e.g. the process already runs under rocprofv3)."""
        out = dict(launches_per_step=None, env_kernel_us=None, gpu_busy_ms_per_step=None)
        if os.environ.get("ROCPROFILER_REGISTER_FORCE_LOAD") or "rocprof" in os.environ.get("LD_PRELOAD", ""):
            return out
        T = self.steps_per_epoch
        graphs, self.graph_rollout = self.graph_rollout, False   # (eager launches: the profiler lists the kernels)
        try:
            from torch.profiler import ProfilerActivity, profile
            self.steps_per_epoch = steps
            self.buf.ptr = 0
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                self.collect(final_reset=False)
                torch.cuda.synchronize()
            ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
            kern = [e for e in ev if "memcpy" not in e.name.lower() and "memset" not in e.name.lower()]
            envk = [e for e in kern if "env_kernel" in e.name]
            out["launches_per_step"] = round(len(ev) / steps, 1)
            out["gpu_busy_ms_per_step"] = round(sum(e.device_time for e in ev) / steps * 1e-3, 4)
            if envk:
                out["env_kernel_us"] = round(sum(e.device_time for e in envk) / len(envk), 2)
        except Exception as ex:  # profiling is a courtesy: never fail the measurement for it
            out["profile_error"] = f"{type(ex).__name__}: {ex}"
        finally:
            self.steps_per_epoch = T
            self.graph_rollout = graphs
            self.buf.ptr = 0
            self.buf.cut.zero_()
        return out

    def training_loop(self):
        self.env.reset()
        self.env.observe()
        for epoch in range(self.epoch + 1):
            t0 = time.time()
            mean_ret = self.collect()
            if (epoch % self.save_freq == 0) or (epoch == self.epoch):
                self.save_model(epoch)
            data = self.buf.get()
            stats = self.update(data)
            self.log.append(dict(epoch=epoch, mean_return=mean_ret, seconds=time.time() - t0, **stats))
        return self.log

    # ---- update -------------------------------------------------------------------------
    def _world(self):
        d = self.dist
        return d.get_world_size() if (d is not None and d.is_initialized()) else 1

    def _check_equal_shards(self):
        """Every rank must own the same number of samples per epoch: the ranks run the same
        number of optimizer steps (one collective each) and the gradients are averaged
        unweighted."""
        if self._world() == 1:
            return
        n = torch.tensor([self.E * self.N, -(self.E * self.N)], dtype=torch.int64, device=self.device)
        self.dist.all_reduce(n, op=self.dist.ReduceOp.MAX)
        if int(n[0]) != -int(n[1]):
            raise ValueError(f"ranks own different shard sizes (max {int(n[0])}, min {-int(n[1])} "
                             "drones): shard the envs equally (sharding.shard_env_range needs a "
                             "total divisible by the world size)")

    def _bucket(self):
        """Multi-rank: every gradient lives in ONE flat buffer (each p.grad is a view of it), so the
        collective of an optimizer step is a single in-place all-reduce - no gather before, no scatter
        after.  Built on first use; zero_grad keeps the views (set_to_none=False, see _zero)."""
        if self._flat is None:
            params = list(self.ac.parameters())
            # + one slot behind the gradients: the policy pass sends its KL estimate along
            flat = torch.zeros(sum(p.numel() for p in params) + 1, dtype=params[0].dtype, device=params[0].device)
            off = 0
            for p in params:
                g = flat[off:off + p.numel()].view_as(p)
                if p.grad is not None:
                    g.copy_(p.grad)
                p.grad = g
                off += p.numel()
            self._flat = flat
        return self._flat

    def _zero(self, opt):
        """optimizer.zero_grad as the reference calls it; with several ranks the gradients stay
        views of the bucket (zeroed in place)."""
        opt.zero_grad(set_to_none=self._world() == 1)

    def _allreduce_grads(self, kl=None):
        """ONE collective per optimizer step: the flat gradient bucket (0.7 - 2.7 MB: latency-bound on
        xGMI), averaged over the ranks.  A policy pass gives its KL estimate a ride in the bucket's last
        slot and gets the mean back (every rank the same value, so every rank stops at the same
        iteration) - the separate all-reduce of one scalar it replaces cost a second latency."""
        d = self.dist
        if d is None or not d.is_initialized() or d.get_world_size() == 1:
            return kl
        flat = self._bucket()
        if kl is not None:
            flat[-1] = float(kl)
        self._allreduce_bucket()
        return float(flat[-1]) if kl is not None else None

    def _allreduce_bucket(self):
        """The collective itself: the bucket averaged over the ranks, in place - ONE call on RCCL
        (ReduceOp.AVG); gloo has no AVG: sum, then divide."""
        d, flat = self.dist, self._bucket()
        if d.get_backend() == "nccl":
            d.all_reduce(flat, op=d.ReduceOp.AVG)
        else:
            d.all_reduce(flat)
            flat /= d.get_world_size()

    def _mean_over_ranks(self, x: float) -> float:
        d = self.dist
        if d is None or not d.is_initialized() or d.get_world_size() == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=self.device)
        d.all_reduce(t)
        return float(t.item()) / d.get_world_size()

    def _batches(self, n):
        mb = self.minibatch_size or n
        if mb >= n:
            yield slice(None)
        else:
            perm = torch.randperm(n, device=self.device)
            for i in range(0, n, mb):
                yield perm[i:i + mb]

    def update(self, data):
        """multi_ppo.update (multi_ppo.py:341-376).  `data`: the flattened rollout of
        RolloutBuffer.get(), or - the reference's own signature - a list with one buffer dict
        per agent (obs = ragged list or padded `obs` + `cnt`)."""
        if self.tune_update and self.device.type == "cuda" and not getattr(self, "_in_tuned_update", False):
            self._in_tuned_update = True
            try:
                keep, self.tune_gemms = self.tune_gemms, True
                with self._tuned_gemms():
                    return self.update(data)
            finally:
                self.tune_gemms = keep
                self._in_tuned_update = False
        if isinstance(data, (list, tuple)):
            return self._update_reference_order(list(data))
        if self.reference_order:
            T, E, N = data["shape"]
            per_agent = []
            for r in range(N):  # agent r = drone r of every env, all steps
                pick = lambda x: x.reshape((T, E, N) + tuple(x.shape[1:]))[:, :, r].reshape(
                    (T * E,) + tuple(x.shape[1:]))
                per_agent.append({k: pick(v) for k, v in data.items() if k != "shape"})
            return self._update_reference_order(per_agent)
        data = {k: v for k, v in data.items() if k != "shape"}
        n = data["adv"].shape[0]
        kl, pi_steps = 0.0, 0
        for i in range(self.train_pi_iters):  # multi_ppo.py:355-368
            stop = False
            for idx in self._batches(n):
                mb = {k: v[idx] for k, v in data.items()}
                self._zero(self.pi_optimizer)
                loss_pi, pi_info = self.compute_loss_pi(mb)
                kl = pi_info["kl"]
                if self._world() > 1:
                    # the mean KL comes back with the gradients (one collective); the check still
                    # precedes the step - at the stopping iteration the backward pass is discarded
                    loss_pi.backward()
                    kl = self._allreduce_grads(kl=kl)
                if kl > self.target_kl:  # KL check before the step
                    stop = True
                    break
                if self._world() == 1:
                    loss_pi.backward()
                torch.nn.utils.clip_grad_norm_(self.ac.parameters(), max_norm=2.0)
                self.pi_optimizer.step()
                pi_steps += 1
            if stop:
                break
        loss_v = torch.zeros(())
        for i in range(self.train_v_iters):  # multi_ppo.py:371-376
            for idx in self._batches(n):
                mb = {k: v[idx] for k, v in data.items()}
                self._zero(self.vf_optimizer)
                loss_v = self.compute_loss_v(mb)
                loss_v.backward()
                self._allreduce_grads()
                self.vf_optimizer.step()
        return dict(kl=kl, pi_steps=pi_steps, loss_v=float(loss_v.detach()))

    def _update_reference_order(self, data_list):
        """The reference's update, statement by statement (multi_ppo.py:341-376): shuffled
        agent order from a generator seeded in the constructor exactly as the reference seeds
        numpy's global one (the same on every rank by construction), max_update_num, per-agent policy
        loop with the KL stop before the step, then the per-agent value loop."""
        randn = np.arange(len(data_list))
        self._order_rng.shuffle(randn)
        update_num, kl, loss_v = 0, 0.0, torch.zeros(())
        pi_steps = []
        for r in randn:
            data = data_list[r]
            update_num += 1
            if update_num > self.max_update_num:
                continue
            steps = 0
            for i in range(self.train_pi_iters):
                self._zero(self.pi_optimizer)
                loss_pi, pi_info = self.compute_loss_pi(data)
                kl = pi_info["kl"]
                if self._world() > 1:  # as in the pooled update: the mean KL rides in the gradient bucket
                    loss_pi.backward()
                    kl = self._allreduce_grads(kl=kl)
                if kl > self.target_kl:
                    break
                if self._world() == 1:
                    loss_pi.backward()
                torch.nn.utils.clip_grad_norm_(self.ac.parameters(), max_norm=2.0)
                self.pi_optimizer.step()
                steps += 1
            pi_steps.append(steps)
            for i in range(self.train_v_iters):
                self._zero(self.vf_optimizer)
                loss_v = self.compute_loss_v(data)
                loss_v.backward()
                self._allreduce_grads()
                self.vf_optimizer.step()
        return dict(kl=kl, pi_steps=pi_steps, order=[int(x) for x in randn],
                    loss_v=float(loss_v.detach()))

    def _obs_arg(self, data):
        return (data["obs"], data["cnt"]) if "cnt" in data else data["obs"]

    def compute_loss_v(self, data):  # multi_ppo.py:379-383
        return ((self.ac.v(self._obs_arg(data)) - data["ret"]) ** 2).mean()

    def compute_loss_pi(self, data):  # multi_ppo.py:385-404
        act, adv, logp_old = data["act"], data["adv"], data["logp"]
        pi, logp = self.ac.pi(self._obs_arg(data), act)
        ratio = torch.exp(logp - logp_old)
        clip_adv = torch.clamp(ratio, 1 - self.clip_ratio, 1 + self.clip_ratio) * adv
        loss_pi = -(torch.min(ratio * adv, clip_adv)).mean()
        clipped = ratio.gt(1 + self.clip_ratio) | ratio.lt(1 - self.clip_ratio)
        # the three diagnostics in ONE device-to-host transfer (the reference's three .item()
        # calls are three synchronisations per optimizer step)
        approx_kl, ent, clipfrac = torch.stack([(logp_old - logp).mean().detach(), pi.entropy().mean().detach(),
                                                clipped.float().mean()]).tolist()
        return loss_pi, dict(kl=approx_kl, ent=ent, cf=clipfrac)

    def save_model(self, index=0):  # multi_ppo.py:406-420
        d = self.dist
        multi = self._world() > 1
        if multi and d.get_rank() != 0:  # replicas are identical: rank 0 alone writes
            d.barrier()
            return
        os.makedirs(self.save_path, exist_ok=True)
        state = dict(model_state=self.ac.state_dict(), pi_optimizer=self.pi_optimizer.state_dict(),
                     vf_optimizer=self.vf_optimizer.state_dict())
        torch.save(self.ac, os.path.join(self.save_path, f"{self.save_name}_{index}.pt"))
        torch.save(state, os.path.join(self.save_path, f"{self.save_name}_check_point_{index}.pt"))
        if multi:
            d.barrier()  # nobody reads the files before they are complete
