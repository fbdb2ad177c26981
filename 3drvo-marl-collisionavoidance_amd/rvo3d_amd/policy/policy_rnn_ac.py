"""Actor-critic of the reference (train/policy/policy_rnn_ac.py), batched.

Same constructor, same sub-module names - so a reference checkpoint's
`model_state` loads with strict=True - same methods (`step`, `act`, `pi(obs, act)`,
`v(obs)`), plus a batched calling convention used by the rollout engine:

    obs   : float32 [B, 12 + 9*nm]   padded rows as the env emits them
    count : int32   [B]              valid VO rows (0 = the single all-zero row)

The reference feeds a ragged list through pad_sequence / pack_padded_sequence
(policy_rnn_ac.py:129-168), which needs the lengths on the host.  Here the biGRU is
unrolled over the nm slots with a validity mask (forward direction: steps 0..len-1;
reverse direction: steps len-1..0), which is the same function of the valid rows and
never synchronises with the host; every step is a [B, 9]x[9, 3H] + [B, H]x[H, 3H] GEMM.
"""
from __future__ import annotations

import weakref

import numpy as np
import torch
import torch.nn as nn
from torch.distributions.normal import Normal


_CAST_CACHE = {}


def _cast_cached(p, dtype):
    """p.to(dtype) for inference, computed once per parameter VERSION: the rollout calls the policy
    hundreds of times between two optimizer steps, and an optimizer step bumps `p._version` (in-place
    update), which invalidates the entry.  (14 cast kernels per rollout step otherwise.)
    An entry belongs to one live Parameter object: it holds a weak reference to it and is dropped when
    the parameter dies (CPython reuses the id of a freed object - a second model built after the first
    was freed must not inherit its casts), and it is only valid for the same storage, shape and device."""
    if p.dtype == dtype:
        return p
    key = (id(p), dtype)
    hit = _CAST_CACHE.get(key)
    if (hit is not None and hit[0]() is p and hit[1] == p._version and hit[2] == p.data_ptr()
            and hit[3].shape == p.shape and hit[3].device == p.device):
        return hit[3]
    t = p.detach().to(dtype)
    ref = weakref.ref(p, lambda _r, k=key: _CAST_CACHE.pop(k, None))
    _CAST_CACHE[key] = (ref, p._version, p.data_ptr(), t)
    return t


def _mlp_pair_plan(pi_net, v_net, dtype):
    """Inference plan for an actor / critic pair of ReLU MLPs of equal depth and widths (None otherwise): the two
    first layers concatenated into ONE [K -> 2 H1] GEMM (K zero-padded to a multiple of 64 in reduced precision:
    measured, [262144, 102] x [102, 512] bf16: 130 us; K = 128: 89 us), the deeper hidden layers per net, the two heads
    as float32 rows for rvo3d_policy_sample."""
    pl = [m for m in pi_net if isinstance(m, nn.Linear)]
    vl = [m for m in v_net if isinstance(m, nn.Linear)]
    pa = [m for m in pi_net if not isinstance(m, nn.Linear)]
    va = [m for m in v_net if not isinstance(m, nn.Linear)]
    if (len(pl) != len(vl) or len(pl) < 2 or not all(isinstance(m, nn.ReLU) for m in pa[:-1] + va[:-1])
            or not isinstance(pa[-1], (nn.Tanh, nn.Identity)) or not isinstance(va[-1], nn.Identity)
            or pl[0].out_features != vl[0].out_features or pl[-1].in_features != vl[-1].in_features
            or pl[0].in_features != vl[0].in_features or pl[-1].out_features != 3 or vl[-1].out_features != 1):
        return None
    with torch.no_grad():
        W = pl[0].in_features
        Kp = W if dtype == torch.float32 else (W + 63) // 64 * 64
        w1 = torch.zeros((pl[0].out_features + vl[0].out_features, Kp), dtype=dtype, device=pl[0].weight.device)
        w1[:, :W] = torch.cat([pl[0].weight, vl[0].weight], 0).to(dtype)
        return dict(
            w1=w1, k_in=W, k_pad=Kp,                                                      # [2 H1, Kp]
            b1=torch.cat([pl[0].bias, vl[0].bias], 0).to(dtype).contiguous(),
            mid=[(p.weight.to(dtype).contiguous(), p.bias.to(dtype).contiguous(),
                  v.weight.to(dtype).contiguous(), v.bias.to(dtype).contiguous()) for p, v in zip(pl[1:-1], vl[1:-1])],
            w_pi=pl[-1].weight.detach().float().contiguous(), b_pi=pl[-1].bias.detach().float().contiguous(),
            w_v=vl[-1].weight.detach().float().reshape(-1).contiguous(), b_v=vl[-1].bias.detach().float().contiguous(),
            h1=pl[0].out_features, hidden=pl[-1].in_features, tanh=isinstance(pa[-1], nn.Tanh))


def _hidden_pair(x, plan):
    """(actor hidden, critic hidden) [B, H] for rvo3d_policy_sample: ONE GEMM for the two first layers (bias + ReLU
    in its epilogue), then one GEMM per net and further hidden layer, each reading its half of the previous output in
    place (row stride 2 H: no copies)."""
    h = torch._addmm_activation(plan["b1"], x, plan["w1"].t(), use_gelu=False)            # [B, 2 H1]
    H1 = plan["h1"]
    hp, hv = h[:, :H1], h[:, H1:]
    for wp, bp, wv, bv in plan["mid"]:
        hp = torch._addmm_activation(bp, hp, wp.t(), use_gelu=False)
        hv = torch._addmm_activation(bv, hv, wv.t(), use_gelu=False)
    return hp, hv


def _plan_cached(module, dtype, build):
    """`build()` once per (dtype, parameter versions, parameter storages) of `module`."""
    params = list(module.parameters())
    key = (dtype, tuple(p._version for p in params), tuple(p.data_ptr() for p in params))
    hit = getattr(module, "_plan", None)
    if hit is not None and hit[0] == key:
        return hit[1]
    plan = build()
    module._plan = (key, plan)
    return plan


def collapsed_first_layer(reader, lin, h0=None):
    """The first linear layer behind the reader, for rows WITHOUT a velocity-obstacle row, as a function of the row's
    state_dim proprioceptive floats p alone (float64): with h0 the GRU's hidden state for a zero input from h = 0 (the
    same for every such row; both directions summed), mean / rstd the LayerNorm statistics of concat(p, h0) and g, bt
    the LayerNorm's affine,  lin(LayerNorm(concat(p, h0))) = W_p f_p + rstd a - (mean rstd) b + c  with
    f_p = (p - mean) rstd g_p + bt_p,  a = W_h (h0 g_h),  b = W_h g_h,  c = W_h bt_h + bias.  Returns (W_p, a, b, c)."""
    sd = reader.state_dim
    with torch.no_grad():
        if h0 is None:
            z = torch.zeros((1, reader.input_dim), device=lin.weight.device)
            h0 = reader._gru_first(z, "")
            if reader.mode == "biGRU":
                h0 = h0 + reader._gru_first(z, "_reverse")
            h0 = h0.reshape(-1).double()
        g, bt = reader.ln.weight.double(), reader.ln.bias.double()
        W1, b1 = lin.weight.double(), lin.bias.double()
        Wh = W1[:, sd:]
        return W1[:, :sd], Wh @ (h0 * g[sd:]), Wh @ g[sd:], Wh @ bt[sd:] + b1


class _SplitKLinearFn(torch.autograd.Function):
    """y = x W^T + b whose weight gradient is computed as S partial products summed afterwards.  The update's weight
    gradients are [out, rows] x [rows, in] products with rows = 262 144 and 256 x 256 (or 3 x 256) results: as ONE GEMM
    the library launches as many workgroups as the tiny result has tiles (measured, float32, MI355X: 787 / 689 / 541 us
    for 256x256 / 256x102 / 3x256); as a batch of S = 64 slices of the rows + a sum: 271 / 157 / 71 us.  Same
    arithmetic up to the order of the float32 sums (relative difference 1e-5 of the largest entry)."""

    @staticmethod
    def forward(ctx, x, w, b, slices):
        ctx.save_for_backward(x, w)
        ctx.slices = slices
        return torch.nn.functional.linear(x, w, b)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        S, B = ctx.slices, x.shape[0]
        gx = gy @ w if ctx.needs_input_grad[0] else None
        gy = gy.contiguous()
        gw = torch.bmm(gy.view(S, B // S, gy.shape[1]).transpose(1, 2), x.view(S, B // S, x.shape[1])).sum(0)
        return gx, gw, gy.sum(0), None


class _Linear(nn.Linear):
    """nn.Linear (same parameters, same state-dict keys) whose backward splits the weight-gradient product over the
    rows when there are many of them (training batches on the GPU); everything else is nn.Linear's own path."""
    split_rows = 32768   # from this many rows on
    slices = 64

    def forward(self, x):
        if (x.is_cuda and x.dim() == 2 and x.shape[0] >= self.split_rows and x.shape[0] % self.slices == 0
                and torch.is_grad_enabled() and self.weight.requires_grad and self.bias is not None
                and x.dtype == torch.float32 and x.is_contiguous() and not torch.is_autocast_enabled()):
            return _SplitKLinearFn.apply(x, self.weight, self.bias, self.slices)
        return super().forward(x)


def mlp(sizes, activation, output_activation=nn.Identity):  # policy_rnn_ac.py:10-17
    layers = []
    for j in range(len(sizes) - 1):
        act = activation if j < len(sizes) - 2 else output_activation
        layers += [_Linear(sizes[j], sizes[j + 1]), act()]
    return nn.Sequential(*layers)


def _as_batch(obs, state_dim, input_dim, device=None):
    """Accept (obs[B,W], count[B]) | list of ragged 1-D tensors | one ragged 1-D tensor.
    Returns (padded [B, W], lengths [B] >= 1, was_single)."""
    if isinstance(obs, tuple):
        o, c = obs
        return o, torch.clamp(c.to(torch.int64), min=1), False
    single = not isinstance(obs, (list, tuple))
    items = [obs] if single else list(obs)
    items = [torch.as_tensor(x, dtype=torch.float32) for x in items]
    lens = torch.tensor([max((len(x) - state_dim) // input_dim, 1) for x in items])
    W = state_dim + input_dim * int(lens.max())
    out = torch.zeros((len(items), W), dtype=torch.float32)
    for i, x in enumerate(items):
        out[i, :len(x)] = x
    if device is not None:
        out, lens = out.to(device), lens.to(device)
    return out, lens, single


class rnn_Reader(nn.Module):  # policy_rnn_ac.py:75-168
    def __init__(self, state_dim, input_dim, hidden_dim, use_gpu=False, mode="GRU"):
        super().__init__()
        self.state_dim, self.input_dim, self.hidden_dim = state_dim, input_dim, hidden_dim
        self.mode, self.use_gpu = mode, use_gpu
        if mode == "GRU":
            self.rnn_net = nn.GRU(input_dim, hidden_dim, batch_first=True)
        elif mode == "LSTM":
            self.rnn_net = nn.LSTM(input_dim, hidden_dim, batch_first=True)
        elif mode == "biGRU":
            self.rnn_net = nn.GRU(input_dim, hidden_dim, batch_first=True, bidirectional=True)
        else:
            raise ValueError(mode)
        self.ln = nn.LayerNorm(state_dim + hidden_dim)
        if use_gpu:
            self.rnn_net, self.ln = self.rnn_net.cuda(), self.ln.cuda()

    def _gru_dir(self, x, lens, suffix, reverse):
        """Final hidden state of one GRU direction over the valid prefix of each row."""
        r = self.rnn_net
        w_ih, w_hh = getattr(r, "weight_ih_l0" + suffix), getattr(r, "weight_hh_l0" + suffix)
        b_ih, b_hh = getattr(r, "bias_ih_l0" + suffix), getattr(r, "bias_hh_l0" + suffix)
        B, S, _ = x.shape
        H = self.hidden_dim
        gi_all = torch.addmm(b_ih, x.reshape(B * S, -1), w_ih.t()).view(B, S, 3 * H)
        h = x.new_zeros((B, H))
        steps = range(S - 1, -1, -1) if reverse else range(S)
        for t in steps:
            gh = torch.addmm(b_hh, h, w_hh.t())
            gi = gi_all[:, t]
            i_r, i_z, i_n = gi.chunk(3, 1)
            h_r, h_z, h_n = gh.chunk(3, 1)
            rg = torch.sigmoid(i_r + h_r)
            zg = torch.sigmoid(i_z + h_z)
            ng = torch.tanh(i_n + rg * h_n)
            hn = (1 - zg) * ng + zg * h
            h = torch.where((lens > t).unsqueeze(1), hn, h)
        return h

    def _gru_first(self, xt, suffix):
        """One GRU cell step from h = 0 (the whole recurrence of a row with ONE valid VO row): with h = 0 the
        hidden-side pre-activation is its bias alone - no [B, H] x [H, 3H] GEMM - and hn = (1 - z) n.  Bit for
        bit what _gru_dir computes for such a row (b_hh + 0 W = b_hh, z * 0 = 0)."""
        r = self.rnn_net
        w_ih = getattr(r, "weight_ih_l0" + suffix)
        b_ih, b_hh = getattr(r, "bias_ih_l0" + suffix), getattr(r, "bias_hh_l0" + suffix)
        gi = torch.addmm(b_ih, xt, w_ih.t())
        i_r, i_z, i_n = gi.chunk(3, 1)
        h_r, h_z, h_n = b_hh.chunk(3, 0)
        rg = torch.sigmoid(i_r + h_r)
        zg = torch.sigmoid(i_z + h_z)
        ng = torch.tanh(i_n + rg * h_n)
        return (1 - zg) * ng

    def forward_batch(self, obs, lens):
        """obs [B, state_dim + input_dim*S] padded; lens [B] >= 1.

        In a rollout nearly every drone has zero or one VO row (lens == 1; 0.5 % of the rows have more at the
        benchmark's density), and a one-step sequence from h = 0 needs no recurrent GEMM at all.  So: the one-step
        result for every row (two [B, 9] x [9, 3H] GEMMs and a handful of elementwise passes), then the masked
        unrolled recurrence over the S slots only for the rows with lens > 1, gathered into a small batch and
        written back.  The same function of the valid rows as before (and as pack_padded_sequence in the
        reference, policy_rnn_ac.py:129-168); at 262144 rows x 10 slots it replaces twenty [B, 256] x [256, 768]
        GEMMs per forward (103 ms in fp32) by two [B, 9] x [9, 768] ones."""
        B = obs.shape[0]
        robot = obs[:, :self.state_dim]
        S = (obs.shape[1] - self.state_dim) // self.input_dim
        x = obs[:, self.state_dim:self.state_dim + S * self.input_dim].reshape(B, S, self.input_dim)
        lens = lens.to(obs.device)
        if self.mode == "LSTM":
            packed = nn.utils.rnn.pack_padded_sequence(x, lens.cpu(), batch_first=True,
                                                       enforce_sorted=False)
            _, (hn, _) = self.rnn_net(packed)
            hnv = hn[0]
        else:
            bi = self.mode == "biGRU"
            x0 = x[:, 0]
            hnv = self._gru_first(x0, "")
            if bi:  # sum of the two directions (policy_rnn_ac.py:121-122); a one-step sequence reversed is itself
                hnv = hnv + self._gru_first(x0, "_reverse")
            if S > 1:
                idx = torch.nonzero(lens > 1).flatten()   # (one host synchronisation per forward)
                if idx.numel() > 0:
                    xs, ls = x.index_select(0, idx), lens.index_select(0, idx)
                    hs = self._gru_dir(xs, ls, "", False)
                    if bi:
                        hs = hs + self._gru_dir(xs, ls, "_reverse", True)
                    hnv = hnv.index_copy(0, idx, hs.to(hnv.dtype))
        return self.ln(torch.cat((robot, hnv), 1))

    # reference names
    def obs_rnn(self, obs):
        o, lens, _ = _as_batch(obs, self.state_dim, self.input_dim, self._device())
        return self.forward_batch(o, lens)[0]

    def obs_rnn_list(self, obs_tensor_list):
        o, lens, _ = _as_batch(list(obs_tensor_list), self.state_dim, self.input_dim, self._device())
        return self.forward_batch(o, lens)

    def _device(self):
        return self.ln.weight.device


class Actor(nn.Module):  # policy_rnn_ac.py:170-188
    def forward(self, obs, act=None, std_factor=1):
        pi = self._distribution(obs, std_factor)
        logp_a = None
        if act is not None:
            logp_a = self._log_prob_from_distribution(pi, act)
        return pi, logp_a


class GaussianActor(Actor):  # policy_rnn_ac.py:191-235
    def __init__(self, obs_dim, act_dim, hidden_sizes, activation, output_activation,
                 rnn_reader=None, use_gpu=False):
        super().__init__()
        self.rnn_reader, self.use_gpu = rnn_reader, use_gpu
        self.net_out = mlp([obs_dim] + list(hidden_sizes) + [act_dim], activation, output_activation)
        log_std = -1 * np.ones(act_dim, dtype=np.float32)
        self.log_std = torch.nn.Parameter(torch.as_tensor(log_std))
        if use_gpu:
            self.net_out = self.net_out.cuda()
            self.log_std = torch.nn.Parameter(torch.as_tensor(log_std, device=torch.device("cuda")))

    def _features(self, obs):
        r = self.rnn_reader
        o, lens, single = _as_batch(obs, r.state_dim, r.input_dim, r._device())
        return r.forward_batch(o, lens), single

    def _distribution(self, obs, std_factor=1, check=False):
        feat, single = self._features(obs)
        if check and not torch.isfinite(feat).all():  # policy_rnn_ac.py:214-216 (host sync)
            raise ValueError("observation contains NaN/Inf")
        mu = self.net_out(feat)
        std = torch.clamp(std_factor * torch.exp(self.log_std) + 1e-6, min=1e-4, max=10.0)
        if single:
            mu = mu[0]
        return Normal(mu, std)

    def _log_prob_from_distribution(self, pi, act):
        return pi.log_prob(act.to(pi.mean.device)).sum(axis=-1)


class Critic(nn.Module):  # policy_rnn_ac.py:238-257
    def __init__(self, obs_dim, hidden_sizes, activation, output_activation, rnn_reader=None,
                 use_gpu=False):
        super().__init__()
        self.v_net = mlp([obs_dim] + list(hidden_sizes) + [1], activation, output_activation)
        if use_gpu:
            self.v_net = self.v_net.cuda()
        self.rnn_reader = rnn_reader

    def forward(self, obs):
        r = self.rnn_reader
        o, lens, single = _as_batch(obs, r.state_dim, r.input_dim, r._device())
        v = torch.squeeze(self.v_net(r.forward_batch(o, lens)), -1)
        return v[0] if single else v


class rnn_ac(nn.Module):  # policy_rnn_ac.py:31-72
    def __init__(self, observation_space, action_space, state_dim, rnn_input_dim=9,
                 rnn_hidden_dim=64, hidden_sizes_ac=(256, 256), hidden_sizes_v=(16, 16),
                 activation=nn.ReLU, output_activation=nn.Tanh, output_activation_v=nn.Identity,
                 use_gpu=True, rnn_mode="GRU", drop_p=0):
        super().__init__()
        self.use_gpu = use_gpu
        obs_dim = rnn_hidden_dim + state_dim
        rnn = rnn_Reader(state_dim, rnn_input_dim, rnn_hidden_dim, use_gpu=use_gpu, mode=rnn_mode)
        act_dim = action_space.shape[0] if hasattr(action_space, "shape") else int(action_space)
        self.pi = GaussianActor(obs_dim, act_dim, hidden_sizes_ac, activation, output_activation,
                                rnn_reader=rnn, use_gpu=use_gpu)
        self.v = Critic(obs_dim, hidden_sizes_v, activation, output_activation_v, rnn_reader=rnn,
                        use_gpu=use_gpu)

    # ---- rollout fast path (rvo3d_amd.policy.multi_ppo._collect_fused, "heads" mode) ----
    def fused_plan(self, dtype):
        """The reader's weights as rvo3d_reader_first_step wants them (float32) + the MLP pair plan of actor and critic
        on the reader's features; None when the architecture has no fast path (LSTM reader, separate readers, a hidden
        width or input width the kernel has no instantiation for, non-ReLU stacks)."""
        def build():
            r = self.pi.rnn_reader
            if (r is None or r is not self.v.rnn_reader or r.mode not in ("GRU", "biGRU") or r.input_dim != 9
                    or r.state_dim > 32 or r.hidden_dim not in (64, 128, 192, 256)):
                return None
            plan = _mlp_pair_plan(self.pi.net_out, self.v.v_net, dtype)
            if plan is None or plan["k_in"] != r.state_dim + r.hidden_dim:
                return None
            g = r.rnn_net
            f32 = lambda t: t.detach().float().contiguous()
            plan["reader"] = dict(
                w_ih_f=f32(g.weight_ih_l0), b_ih_f=f32(g.bias_ih_l0), b_hh_f=f32(g.bias_hh_l0),
                w_ih_r=f32(g.weight_ih_l0_reverse) if r.mode == "biGRU" else None,
                b_ih_r=f32(g.bias_ih_l0_reverse) if r.mode == "biGRU" else None,
                b_hh_r=f32(g.bias_hh_l0_reverse) if r.mode == "biGRU" else None,
                ln_w=f32(r.ln.weight), ln_b=f32(r.ln.bias), eps=float(r.ln.eps))
            return plan
        return _plan_cached(self, dtype, build)

    def zero_vo_plan(self):
        """Weights for the rollout's fastest path (multi_ppo._collect_fused, mode "rnn0"): rows WITHOUT a velocity-obstacle
        row - nearly all of a rollout - all share the GRU's hidden state h0 (zero input from h = 0), so LayerNorm(concat(p,
        h0)) depends on a row through mean and rstd only and the first MLP layer collapses to a product over
        state_dim + 3 inputs (include/rvo3d.h, rvo3d_reader_zero_features): the whole policy step of such rows is then
        rvo3d_reader_zero_features + rvo3d_policy_mlp_sample.  Returns dict(blob, width, sum_h0, sumsq_h0, ln_w, ln_b,
        eps, feat_dim, tanh), rebuilt when a parameter changed; None when the architecture does not fit (LSTM or separate
        readers, heads other than ReLU (256, 256) stacks, not on a GPU)."""
        r = self.pi.rnn_reader
        if (r is None or r is not self.v.rnn_reader or r.mode not in ("GRU", "biGRU") or r.state_dim > 16
                or r.hidden_dim > 256 or r.input_dim > 16 or next(self.parameters()).device.type != "cuda"):
            return None
        nets = (self.pi.net_out, self.v.v_net)
        lins = [[m for m in n if isinstance(m, nn.Linear)] for n in nets]
        acts = [[m for m in n if not isinstance(m, nn.Linear)] for n in nets]
        D = r.state_dim + r.hidden_dim
        if (any(len(l) != 3 for l in lins) or [m.out_features for m in lins[0]] != [256, 256, 3]
                or [m.out_features for m in lins[1]] != [256, 256, 1] or any(l[0].in_features != D for l in lins)
                or not all(isinstance(m, nn.ReLU) for a in acts for m in a[:-1])
                or not isinstance(acts[0][-1], (nn.Tanh, nn.Identity)) or not isinstance(acts[1][-1], nn.Identity)
                or any(m.bias is None for l in lins for m in l)):
            return None
        params = list(self.parameters())
        key = (tuple(p._version for p in params), tuple(p.data_ptr() for p in params))
        hit = getattr(self, "_zero_plan", None)
        if hit is not None and hit[0] == key:
            return hit[1]
        import ctypes as C
        from .. import _lib
        L = _lib.lib()
        dev = params[0].device
        sd, width = r.state_dim, r.state_dim + 8
        bf = torch.bfloat16
        with torch.no_grad():
            z = torch.zeros((1, r.input_dim), device=dev)
            h0 = r._gru_first(z, "")
            if r.mode == "biGRU":
                h0 = h0 + r._gru_first(z, "_reverse")
            h0 = h0.reshape(-1).double()
            keep = []
            for l in lins:
                Wp, a, b, c = collapsed_first_layer(r, l[0], h0)
                head = lambda x: x.float().to(bf).float()
                cols = [Wp.float()]
                for vec, sign in ((a, 1.0), (b, -1.0)):
                    v = (sign * vec).float()
                    cols += [head(v)[:, None], (v - head(v))[:, None], head(v)[:, None]]
                cf = c.float()
                cols += [head(cf)[:, None], (cf - head(cf))[:, None]]
                keep += [torch.cat(cols, 1).contiguous(), torch.zeros(256, device=dev),
                         l[1].weight.detach().float().contiguous(), l[1].bias.detach().float().contiguous(),
                         l[2].weight.detach().float().contiguous(), l[2].bias.detach().float().contiguous()]
            blob = hit[1]["blob"] if hit is not None else torch.empty(int(L.rvo3d_policy_mlp_blob_bytes(width)),
                                                                       dtype=torch.uint8, device=dev)
            wa = _lib.MlpWeights(*[t.data_ptr() for t in keep[:6]])
            wb = _lib.MlpWeights(*[t.data_ptr() for t in keep[6:]])
            with torch.cuda.device(dev):
                _lib.check(L.rvo3d_policy_mlp_pack(C.byref(wa), C.byref(wb), width, C.c_void_p(blob.data_ptr()),
                                                   C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
                           "rvo3d_policy_mlp_pack")
                torch.cuda.current_stream(dev).synchronize()  # (the temporaries above die with this scope)
            # the modules' own tensors for rvo3d_policy_rows (the rows that do have VO rows): live parameters, no copies
            gp = lambda n: getattr(r.rnn_net, n).data_ptr()
            bi = r.mode == "biGRU"
            rows_net = _lib.RnnPolicy(
                gp("weight_ih_l0"), gp("weight_hh_l0"), gp("bias_ih_l0"), gp("bias_hh_l0"),
                gp("weight_ih_l0_reverse") if bi else None, gp("weight_hh_l0_reverse") if bi else None,
                gp("bias_ih_l0_reverse") if bi else None, gp("bias_hh_l0_reverse") if bi else None,
                r.ln.weight.data_ptr(), r.ln.bias.data_ptr(), r.hidden_dim, r.input_dim, r.state_dim, 0, float(r.ln.eps), 0,
                _lib.MlpWeights(*[t.data_ptr() for l in lins[0] for t in (l.weight, l.bias)]),
                _lib.MlpWeights(*[t.data_ptr() for l in lins[1] for t in (l.weight, l.bias)]))
            ok_rows = all(t.is_contiguous() and t.dtype == torch.float32 for t in params)
            out = dict(blob=blob, width=width, sum_h0=float(h0.sum()), sumsq_h0=float((h0 * h0).sum()),
                       rows_net=rows_net if ok_rows else None,
                       ln_w=r.ln.weight.detach().float().contiguous(), ln_b=r.ln.bias.detach().float().contiguous(),
                       eps=float(r.ln.eps), feat_dim=D, state_dim=sd, tanh=isinstance(acts[0][-1], nn.Tanh))
        self._zero_plan = (key, out)
        return out

    def prepare_input(self, obs, cnt, plan, cache):
        """The reader's features [rows, Kp] as the A operand of the first MLP layer: rvo3d_reader_first_step for every
        row (one GRU cell step per direction from h = 0, direction sum, concat, LayerNorm: exact for rows with at most
        one VO row), then the rows with MORE than one VO row - 0.5 % at the benchmark's density - again through the
        module's own masked recurrence, gathered into a small batch (one host synchronisation for their indices)."""
        import ctypes as C
        from .. import _lib
        r, rd = self.pi.rnn_reader, plan["reader"]
        dt = plan["w1"].dtype
        B = obs.shape[0]
        feat = cache.get("feat")
        if feat is None or feat.shape[0] < B or feat.shape[1] != plan["k_pad"] or feat.dtype != dt:
            feat = cache["feat"] = torch.zeros((B, plan["k_pad"]), dtype=dt, device=obs.device)
        pp = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        st = _lib.GruReader(pp(rd["w_ih_f"]), pp(rd["b_ih_f"]), pp(rd["b_hh_f"]), pp(rd["w_ih_r"]), pp(rd["b_ih_r"]),
                            pp(rd["b_hh_r"]), pp(rd["ln_w"]), pp(rd["ln_b"]), r.hidden_dim, r.input_dim, r.state_dim,
                            rd["eps"])
        _lib.check(_lib.lib().rvo3d_reader_first_step(
            C.byref(st), pp(obs), obs.stride(0), B, pp(feat), _lib.RVO3D_BF16 if dt == torch.bfloat16 else _lib.RVO3D_F32,
            feat.stride(0), C.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream)), "rvo3d_reader_first_step")
        idx = torch.nonzero(cnt > 1).flatten()
        if idx.numel() > 0:
            with torch.no_grad():
                full = r.forward_batch(obs.index_select(0, idx), cnt.index_select(0, idx).to(torch.int64))
            feat[:B, :full.shape[1]].index_copy_(0, idx, full.to(dt))
        return feat[:B]

    def hidden_pair(self, x, plan):
        return _hidden_pair(x, plan)

    @property
    def log_std(self):
        return self.pi.log_std

    def step_tensors(self, obs, std_factor=1):
        """Batched, stays on the device: (a, v, logp) tensors."""
        with torch.no_grad():
            pi_dis = self.pi._distribution(obs, std_factor)
            a = pi_dis.sample()
            logp_a = self.pi._log_prob_from_distribution(pi_dis, a)
            v = self.v(obs)
        return a, v, logp_a

    def step(self, obs, std_factor=1):  # reference: numpy out (policy_rnn_ac.py:57-69)
        a, v, logp_a = self.step_tensors(obs, std_factor)
        return a.cpu().numpy(), v.cpu().numpy(), logp_a.cpu().numpy()

    def act(self, obs, std_factor=1):
        return self.step(obs, std_factor)[0]


class _NoReader(nn.Module):
    """Reader of the MLP policy: the padded observation itself (no recurrence)."""

    def __init__(self, width):
        super().__init__()
        self.state_dim, self.input_dim, self.width = width, 1, width

    def forward_batch(self, obs, lens):
        return obs

    def _device(self):
        return next(self.parameters(), torch.zeros(())).device


class mlp_ac(nn.Module):
    """BASELINE config 3's MLP(256, 256) actor-critic on the fixed-width padded
    observation (12 + 9*nm floats): same `pi` / `v` / `step` surface as rnn_ac."""

    def __init__(self, obs_width, act_dim=3, hidden_sizes=(256, 256), activation=nn.ReLU,
                 output_activation=nn.Tanh):
        super().__init__()
        self.obs_width = obs_width
        self.pi_net = mlp([obs_width] + list(hidden_sizes) + [act_dim], activation, output_activation)
        self.v_net = mlp([obs_width] + list(hidden_sizes) + [1], activation, nn.Identity)
        self.log_std = nn.Parameter(-1 * torch.ones(act_dim))

    def _obs(self, obs):
        return obs[0] if isinstance(obs, tuple) else obs

    def dist(self, obs, std_factor=1):
        mu = self.pi_net(self._obs(obs))
        std = torch.clamp(std_factor * torch.exp(self.log_std) + 1e-6, min=1e-4, max=10.0)
        return Normal(mu, std)

    def pi(self, obs, act=None, std_factor=1):
        d = self.dist(obs, std_factor)
        return d, (None if act is None else d.log_prob(act).sum(-1))

    def v(self, obs):
        return self.v_net(self._obs(obs)).squeeze(-1)

    @staticmethod
    def _fused_forward(net, x):
        """Inference-only forward of an mlp() stack whose hidden activations are ReLU: bias +
        ReLU run in the GEMM epilogue (hipBLASLt, `torch._addmm_activation`) instead of a
        separate pass over the [E*N, 256] activations; bf16 operands under autocast."""
        mods = list(net)
        lin = [m for m in mods if isinstance(m, nn.Linear)]
        acts = [m for m in mods if not isinstance(m, nn.Linear)]
        if len(acts) != len(lin) or not all(isinstance(m, nn.ReLU) for m in acts[:-1]):
            return net(x)
        dt = torch.bfloat16 if torch.is_autocast_enabled() else x.dtype
        h = x.to(dt)  # (a no-op when the caller already cast the observation: step_tensors does, once for both nets)
        for m in lin[:-1]:
            h = torch._addmm_activation(_cast_cached(m.bias, dt), h, _cast_cached(m.weight, dt).t(), use_gelu=False)
        h = torch.nn.functional.linear(h, _cast_cached(lin[-1].weight, dt), _cast_cached(lin[-1].bias, dt))
        return acts[-1](h)

    # ---- rollout fast path: everything up to the last hidden layers, merged where the two nets allow ----
    def fused_plan(self, dtype):
        """Weights of the inference plan of `hidden_pair` (see _mlp_pair_plan), cached per parameter version (an
        optimizer step rebuilds them); None when the stacks are not ReLU MLPs of equal shape (the caller then takes
        the module path)."""
        return _plan_cached(self, dtype, lambda: _mlp_pair_plan(self.pi_net, self.v_net, dtype))

    def mlp_blob(self):
        """The packed weights of rvo3d_policy_mlp_sample (the whole policy step in ONE kernel on the matrix cores:
        csrc/rvo3d_policy_mlp.hpp), repacked when a parameter changed; None when this is not the shape that kernel
        is written for - ReLU MLPs obs_width -> 256 -> 256 -> 3 (Tanh or Identity) / -> 1, obs_width <= 126, on a GPU."""
        pl = [m for m in self.pi_net if isinstance(m, nn.Linear)]
        vl = [m for m in self.v_net if isinstance(m, nn.Linear)]
        pa = [m for m in self.pi_net if not isinstance(m, nn.Linear)]
        va = [m for m in self.v_net if not isinstance(m, nn.Linear)]
        if (len(pl) != 3 or len(vl) != 3 or self.obs_width > 126 or pl[0].weight.device.type != "cuda"
                or [m.out_features for m in pl] != [256, 256, 3] or [m.out_features for m in vl] != [256, 256, 1]
                or not all(isinstance(m, nn.ReLU) for m in pa[:-1] + va[:-1]) or len(pa) != 3 or len(va) != 3
                or not isinstance(pa[-1], (nn.Tanh, nn.Identity)) or not isinstance(va[-1], nn.Identity)
                or any(m.bias is None for m in pl + vl) or pl[0].weight.dtype != torch.float32):
            return None
        params = list(self.pi_net.parameters()) + list(self.v_net.parameters())
        key = (tuple(p._version for p in params), tuple(p.data_ptr() for p in params))
        hit = getattr(self, "_blob", None)
        if hit is not None and hit[0] == key:
            return hit[1]
        import ctypes as C
        from .. import _lib
        L = _lib.lib()
        dev = pl[0].weight.device
        blob = hit[1]["blob"] if hit is not None else torch.empty(int(L.rvo3d_policy_mlp_blob_bytes(self.obs_width)),
                                                                   dtype=torch.uint8, device=dev)
        keep = [t.detach().contiguous() for lin in (pl, vl) for m in lin for t in (m.weight, m.bias)]
        a = _lib.MlpWeights(*[t.data_ptr() for t in keep[:6]])
        b = _lib.MlpWeights(*[t.data_ptr() for t in keep[6:]])
        with torch.cuda.device(dev):
            _lib.check(L.rvo3d_policy_mlp_pack(C.byref(a), C.byref(b), self.obs_width, C.c_void_p(blob.data_ptr()),
                                               C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
                       "rvo3d_policy_mlp_pack")
        out = dict(blob=blob, tanh=isinstance(pa[-1], nn.Tanh))
        self._blob = (key, out)
        return out

    def prepare_input(self, obs, cnt, plan, cache):
        """The A operand of the first-layer GEMM: the observation itself (float32) or - ONE kernel - its cast into a
        zero-padded [rows, Kp] buffer kept in `cache`."""
        if plan["w1"].dtype == torch.float32:
            return obs
        xp = cache.get("xp")
        if xp is None or xp.shape[0] < obs.shape[0] or xp.shape[1] != plan["k_pad"] or xp.dtype != plan["w1"].dtype:
            xp = cache["xp"] = torch.zeros((obs.shape[0], plan["k_pad"]), dtype=plan["w1"].dtype, device=obs.device)
        xp[:obs.shape[0], :obs.shape[1]].copy_(obs)
        return xp[:obs.shape[0]]

    def hidden_pair(self, x, plan):
        return _hidden_pair(x, plan)

    def step_tensors(self, obs, std_factor=1):
        with torch.no_grad():
            x = self._obs(obs)
            if torch.is_autocast_enabled():  # one bf16 copy of the [E*N, W] observation for both nets
                x = x.to(torch.bfloat16)
            mu = self._fused_forward(self.pi_net, x).float()
            std = torch.clamp(std_factor * torch.exp(self.log_std) + 1e-6, min=1e-4, max=10.0)
            d = Normal(mu, std)
            a = d.sample()
            v = self._fused_forward(self.v_net, x).float().squeeze(-1)
            return a, v, d.log_prob(a).sum(-1)

    def step(self, obs, std_factor=1):
        a, v, lp = self.step_tensors(obs, std_factor)
        return a.cpu().numpy(), v.cpu().numpy(), lp.cpu().numpy()
