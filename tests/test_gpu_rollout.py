"""The trainer's per-step glue on the device (train/policy/multi_ppo.py:193-281):
rvo3d_policy_sample (the two heads of the actor-critic, tanh, sampling, log-probability, np.round(a, 2)
and the buffer stores in one pass) and rvo3d_rollout_account (reward slot, episode counters, path
cuts), each against a plain PyTorch float32 statement of the same lines - these are floating-point
kernels: tolerance 1e-5 on mu / v, 1e-4 on log-probabilities (fast log / exp), stated per check -
and the fused rollout loop as a whole: replaying the stored actions through a second env must
reproduce every stored observation bit for bit."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from rvo3d_amd import BatchedDroneEnv, _lib, synthetic_world
from rvo3d_amd.policy import mlp_ac, multi_ppo

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _sample(hp, hv, wpi, bpi, wv, bv, log_std, rows, hidden, dtype, tanh=True, seed=7, step=0, std_factor=1.0):
    L = _lib.lib()
    act = torch.full((rows, 3), 9.0, device=DEV)
    logp = torch.full((rows,), 9.0, device=DEV)
    val = torch.full((rows,), 9.0, device=DEV)
    mu = torch.zeros((rows, 3), device=DEV)
    raw = torch.zeros((rows, 3), device=DEV)
    hd = _lib.PolicyHeads(hp.data_ptr(), hv.data_ptr(), hp.stride(0), hv.stride(0), dtype, hidden,
                          1 if tanh else 0, 0, wpi.data_ptr() if wpi is not None else None,
                          bpi.data_ptr() if bpi is not None else None, wv.data_ptr() if wv is not None else None,
                          bv.data_ptr() if bv is not None else None, log_std.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(L.rvo3d_policy_sample(C.byref(hd), rows, std_factor, seed, step, _p(act), _p(logp), _p(val),
                                     _p(mu), _p(raw), st), "rvo3d_policy_sample")
    torch.cuda.synchronize()
    return act, logp, val, mu, raw


@pytest.mark.parametrize("dt,hidden,rows", [(torch.float32, 256, 4096), (torch.bfloat16, 256, 65536),
                                            (torch.bfloat16, 512, 1000), (torch.float32, 128, 77),
                                            (torch.bfloat16, 1024, 130), (torch.float32, 1024, 33)])
def test_policy_sample_matches_torch(dt, hidden, rows):
    g = torch.Generator(device=DEV).manual_seed(rows)
    # the hidden activations as views of wider buffers (row stride 2 H), like the merged first-layer GEMM's halves
    H2 = torch.relu(torch.randn((rows, 2 * hidden), device=DEV, generator=g)).to(dt)
    hp, hv = H2[:, :hidden], H2[:, hidden:]
    wpi = torch.randn((3, hidden), device=DEV, generator=g) * 0.08
    bpi = torch.randn(3, device=DEV, generator=g) * 0.1
    wv = torch.randn(hidden, device=DEV, generator=g) * 0.08
    bv = torch.randn(1, device=DEV, generator=g)
    log_std = torch.tensor([-1.0, -0.5, -1.5], device=DEV)
    code = _lib.RVO3D_BF16 if dt == torch.bfloat16 else _lib.RVO3D_F32
    act, logp, val, mu, raw = _sample(hp, hv, wpi, bpi, wv, bv, log_std, rows, hidden, code)
    # float32 statement of policy_rnn_ac.py:197-235 / :242 on the same (already rounded) hidden activations
    mu_ref = torch.tanh(hp.double() @ wpi.double().t() + bpi.double())
    v_ref = hv.double() @ wv.double() + bv.double()
    assert torch.allclose(mu.double(), mu_ref, atol=2e-5, rtol=0), float((mu.double() - mu_ref).abs().max())
    assert torch.allclose(val.double(), v_ref, atol=5e-5, rtol=1e-5), float((val.double() - v_ref).abs().max())
    std = torch.clamp(torch.exp(log_std) + 1e-6, 1e-4, 10.0)
    lp_ref = torch.distributions.Normal(mu, std).log_prob(raw).sum(-1)
    assert torch.allclose(logp, lp_ref, atol=1e-4, rtol=1e-5), float((logp - lp_ref).abs().max())
    # np.round(a, 2) of the trainer (multi_ppo.py:197), in numpy's own float32 arithmetic
    assert np.array_equal(act.cpu().numpy(), np.round(raw.cpu().numpy(), 2))
    eps = ((raw - mu) / std).cpu().numpy()
    assert np.isfinite(eps).all()
    if rows >= 4096:  # the noise is standard normal, independent between the three components
        n = eps.size
        assert abs(eps.mean()) < 4 / math.sqrt(n) and abs(eps.var() - 1) < 6 * math.sqrt(2 / n)
        c = np.corrcoef(eps.T)
        assert np.abs(c - np.eye(3)).max() < 5 / math.sqrt(rows)
        assert abs((np.abs(eps) > 1.959964).mean() - 0.05) < 0.004
    # reproducible: same key and counter, same numbers; another step, other numbers
    act2, logp2, val2, _, raw2 = _sample(hp, hv, wpi, bpi, wv, bv, log_std, rows, hidden, code)
    assert torch.equal(raw, raw2) and torch.equal(logp, logp2) and torch.equal(val, val2)
    _, _, _, _, raw3 = _sample(hp, hv, wpi, bpi, wv, bv, log_std, rows, hidden, code, step=1)
    assert not torch.equal(raw, raw3)
    _, _, _, _, raw4 = _sample(hp, hv, wpi, bpi, wv, bv, log_std, rows, hidden, code, seed=8)
    assert not torch.equal(raw, raw4)


def _mlp_pack(ac, W):
    L = _lib.lib()
    blob = torch.zeros(int(L.rvo3d_policy_mlp_blob_bytes(W)), dtype=torch.uint8, device=DEV)
    keep = [t.detach().float().contiguous() for net in (ac.pi_net, ac.v_net) for m in net
            if isinstance(m, torch.nn.Linear) for t in (m.weight, m.bias)]
    a, b = _lib.MlpWeights(*[t.data_ptr() for t in keep[:6]]), _lib.MlpWeights(*[t.data_ptr() for t in keep[6:]])
    _lib.check(L.rvo3d_policy_mlp_pack(C.byref(a), C.byref(b), W, _p(blob), None), "rvo3d_policy_mlp_pack")
    torch.cuda.synchronize()
    return blob


def _mlp_sample(blob, W, x, log_std, tanh=True, seed=7, step=0, std_factor=1.0, rows=None, cnt=None):
    L = _lib.lib()
    rows = x.shape[0] if rows is None else rows
    act = torch.full((rows, 3), 9.0, device=DEV)
    logp = torch.full((rows,), 9.0, device=DEV)
    val = torch.full((rows,), 9.0, device=DEV)
    mu = torch.zeros((rows, 3), device=DEV)
    raw = torch.zeros((rows, 3), device=DEV)
    _lib.check(L.rvo3d_policy_mlp_sample(_p(blob), W, _p(x), x.stride(0), rows, _p(cnt), 12, 9, 1 if tanh else 0, _p(log_std), std_factor,
                                         seed, step, _p(act), _p(logp), _p(val), _p(mu), _p(raw),
                                         C.c_void_p(torch.cuda.current_stream().cuda_stream)), "rvo3d_policy_mlp_sample")
    torch.cuda.synchronize()
    return act, logp, val, mu, raw


def _mlp_emulation(ac, x):
    """The kernel's arithmetic in plain PyTorch: bf16 operands (observation, weights, the first layer's bias, both
    hidden activations), float32 products / sums / second bias / head bias."""
    bf = torch.bfloat16

    def net(n):
        lin = [m for m in n if isinstance(m, torch.nn.Linear)]
        h = x.to(bf).double()
        h = torch.relu(h @ lin[0].weight.to(bf).double().T + lin[0].bias.to(bf).double()).float().to(bf).double()
        h = torch.relu(h @ lin[1].weight.to(bf).double().T + lin[1].bias.double()).float().to(bf).double()
        return (h @ lin[2].weight.to(bf).double().T + lin[2].bias.double()).float()
    with torch.no_grad():
        return net(ac.pi_net), net(ac.v_net).squeeze(-1)


def test_policy_mlp_layouts_with_exact_integer_data():
    """Every fragment layout of the kernel (A / B operand maps, the permuted k order of a chained product, the
    bias table, the head rows, the two-pass row ownership) with small-integer weights and inputs: every product and
    sum is exact in bf16 x bf16 -> float32, so the result must EQUAL the integer arithmetic - a swapped row, lane
    or k index cannot hide behind a tolerance."""
    W, rows = 102, 64 * 37 + 5
    g = torch.Generator(device=DEV).manual_seed(5)
    ac = mlp_ac(W).to(DEV)
    with torch.no_grad():
        for net in (ac.pi_net, ac.v_net):
            lin = [m for m in net if isinstance(m, torch.nn.Linear)]
            lin[0].weight.copy_(torch.randint(-2, 3, lin[0].weight.shape, device=DEV, generator=g).float())
            lin[0].bias.copy_(torch.randint(-3, 4, lin[0].bias.shape, device=DEV, generator=g).float())
            # sparse second layer and head: the sums stay below 2^8 (exact in bf16 after the ReLU) / 2^24
            w2 = torch.randint(-1, 2, lin[1].weight.shape, device=DEV, generator=g).float()
            w2 *= (torch.rand(w2.shape, device=DEV, generator=g) < 0.04).float()
            lin[1].weight.copy_(w2)
            lin[1].bias.copy_(torch.randint(-2, 3, lin[1].bias.shape, device=DEV, generator=g).float())
            lin[2].weight.copy_(torch.randint(-2, 3, lin[2].weight.shape, device=DEV, generator=g).float())
            lin[2].bias.copy_(torch.randint(-5, 6, lin[2].bias.shape, device=DEV, generator=g).float())
    x = torch.randint(-1, 2, (rows, W), device=DEV, generator=g).float()
    x *= (torch.rand(x.shape, device=DEV, generator=g) < 0.15).float()
    with torch.no_grad():
        h1p, h1v = torch.relu(ac.pi_net[0](x)), torch.relu(ac.v_net[0](x))
        assert float(h1p.max()) <= 256 and float(h1v.max()) <= 256 and float(h1p.max()) > 8   # exact in bf16
        h2p, h2v = torch.relu(ac.pi_net[2](h1p)), torch.relu(ac.v_net[2](h1v))
        assert float(h2p.max()) <= 256 and float(h2v.max()) <= 256 and float(h2p.max()) > 4
        mu_want, v_want = ac.pi_net[4](h2p), ac.v_net[4](h2v).squeeze(-1)
    log_std = torch.tensor([-1.0, -0.5, -1.5], device=DEV)
    act, logp, val, mu, raw = _mlp_sample(_mlp_pack(ac, W), W, x, log_std, tanh=False)
    assert torch.equal(mu, mu_want), int((mu != mu_want).sum())
    assert torch.equal(val, v_want), int((val != v_want).sum())
    assert len(torch.unique(mu_want)) > 30 and len(torch.unique(v_want)) > 30


@pytest.mark.parametrize("W,rows", [(102, 262144), (102, 1000), (57, 5000), (39, 777), (120, 4096), (12, 130), (126, 64)])
def test_policy_mlp_sample_matches_torch(W, rows):
    """rvo3d_policy_mlp_sample against (a) a PyTorch emulation with the kernel's rounding points - the two differ by
    the summation order inside a product (float32) and by the bf16 roundings of hidden activations that such a
    difference flips (one bf16 ulp = 2^-8 of one activation): mean 1e-4, max 3e-2 on the pre-activations at these
    weights; (b) the float32 module: the price of bf16 operands, stated; (c) the sampling lines exactly as
    test_policy_sample_matches_torch does for rvo3d_policy_sample (same per-row code in the kernel)."""
    torch.manual_seed(W * 1000 + rows)
    ac = mlp_ac(W).to(DEV)
    x = torch.randn((rows + 3, W + 5), device=DEV)[:rows, :W] * 2     # a strided view: obs_ld > obs_width
    x[::7, 12:] = 0.0                                                 # rows without VO rows, as the env writes them
    log_std = torch.tensor([-1.0, -0.5, -1.5], device=DEV)
    blob = _mlp_pack(ac, W)
    act, logp, val, mu, raw = _mlp_sample(blob, W, x, log_std)
    z_emu, v_emu = _mlp_emulation(ac, x)
    mu_emu = torch.tanh(z_emu)
    d_mu, d_v = (mu - mu_emu).abs(), (val - v_emu).abs()
    assert float(d_mu.max()) < 1e-2 and float(d_mu.mean()) < 2e-4, (float(d_mu.max()), float(d_mu.mean()))
    assert float(d_v.max()) < 1e-2 and float(d_v.mean()) < 2e-4, (float(d_v.max()), float(d_v.mean()))
    with torch.no_grad():
        mu32, v32 = ac.pi_net(x), ac.v_net(x).squeeze(-1)
    assert float((mu - mu32).abs().max()) < 3e-2 and float((val - v32).abs().max()) < 3e-2
    std = torch.clamp(torch.exp(log_std) + 1e-6, 1e-4, 10.0)
    lp_ref = torch.distributions.Normal(mu, std).log_prob(raw).sum(-1)
    assert torch.allclose(logp, lp_ref, atol=1e-4, rtol=1e-5), float((logp - lp_ref).abs().max())
    assert np.array_equal(act.cpu().numpy(), np.round(raw.cpu().numpy(), 2))
    eps = ((raw - mu) / std).cpu().numpy()
    assert np.isfinite(eps).all()
    if rows >= 4096:
        n = eps.size
        assert abs(eps.mean()) < 4 / math.sqrt(n) and abs(eps.var() - 1) < 6 * math.sqrt(2 / n)
    # same key and counter: the same noise as rvo3d_policy_sample draws for the row (one generator for all paths)
    zero = torch.zeros((rows, 3), device=DEV)
    _, _, _, _, raw_d = _sample(zero, torch.zeros((rows, 1), device=DEV), None, None, None, None, log_std, rows, 0,
                                _lib.RVO3D_F32)
    assert torch.allclose(raw - mu, raw_d, atol=1e-6, rtol=0)
    # reproducible; another step / seed: other numbers; a repack after a weight change is seen
    _, logp2, val2, _, raw2 = _mlp_sample(blob, W, x, log_std)
    assert torch.equal(raw, raw2) and torch.equal(logp, logp2) and torch.equal(val, val2)
    assert not torch.equal(raw, _mlp_sample(blob, W, x, log_std, step=1)[4])
    assert not torch.equal(raw, _mlp_sample(blob, W, x, log_std, seed=8)[4])
    with torch.no_grad():
        ac.v_net[4].bias += 1.0
    assert torch.allclose(_mlp_sample(_mlp_pack(ac, W), W, x, log_std)[2], val + 1.0, atol=1e-5)


@pytest.mark.parametrize("W", [1, 14, 15, 16, 30, 31, 32, 47, 48, 63, 64, 79, 80, 95, 96, 111, 112, 125])
def test_policy_mlp_sample_at_every_step_boundary(W):
    """Observation widths on both sides of every 16-column step boundary (the bias column k = W is the last element of a
    step at W = 15, 31, ..., the first of a new one at 16, 32, ...; all eight kernel instantiations; the number of
    second-layer k-steps kept in LDS changes with the width too): against the emulation, with and without counts."""
    torch.manual_seed(W)
    rows = 64 * 3 + 37
    ac = mlp_ac(W).to(DEV)
    x = torch.randn((rows, W), device=DEV)
    log_std = torch.tensor([-1.0, -0.5, -1.5], device=DEV)
    blob = _mlp_pack(ac, W)
    act, logp, val, mu, raw = _mlp_sample(blob, W, x, log_std)
    z_emu, v_emu = _mlp_emulation(ac, x)
    assert float((mu - torch.tanh(z_emu)).abs().max()) < 5e-3 and float((val - v_emu).abs().max()) < 5e-3
    if W >= 12:   # rows as the env shapes them: 12 floats, 9 per VO row, zeros behind - with their counts: the same bits
        nmax = (W - 12) // 9
        cnt = torch.randint(0, nmax + 1, (rows,), device=DEV, dtype=torch.int32)
        xs = x * (torch.arange(W, device=DEV)[None, :] < (12 + 9 * cnt.long())[:, None]).float()
        a = _mlp_sample(blob, W, xs, log_std)
        b = _mlp_sample(blob, W, xs, log_std, cnt=cnt)
        assert all(torch.equal(p, q) for p, q in zip(a, b))


@pytest.mark.parametrize("nm", [10, 5, 3])
def test_policy_mlp_sample_skips_zero_column_groups_exactly(nm):
    """With the env's vo_count the kernel neither loads nor multiplies the 16-float column groups that are zero for all
    32 rows of a wave: a zero activation adds exactly nothing, so the results must be BIT-IDENTICAL to the dense pass -
    on rows shaped like the env's (12 floats, 9 per kept VO row, zeros behind; count 0 = one all-zero row), with every
    mixture of counts inside a 32-row group: all sparse, one dense row among sparse ones, all dense."""
    W, rows = 12 + 9 * nm, 64 * 50 + 17
    g = torch.Generator(device=DEV).manual_seed(nm)
    ac = mlp_ac(W).to(DEV)
    cnt = torch.randint(0, 3, (rows,), device=DEV, generator=g, dtype=torch.int32)
    cnt[torch.rand(rows, device=DEV, generator=g) < 0.02] = nm           # a dense row here and there
    cnt[64 * 7:64 * 9] = torch.randint(0, nm + 1, (128,), device=DEV, generator=g, dtype=torch.int32)  # two mixed groups
    cnt[64 * 20:64 * 21] = nm                                             # one all-dense group
    cnt[64 * 30:64 * 32] = 0                                              # all-empty groups
    x = torch.randn((rows, W), device=DEV, generator=g)
    used = 12 + 9 * cnt.clamp(min=0).long()
    x *= (torch.arange(W, device=DEV)[None, :] < used[:, None]).float()
    log_std = torch.tensor([-1.0, -0.5, -1.5], device=DEV)
    blob = _mlp_pack(ac, W)
    dense = _mlp_sample(blob, W, x, log_std)
    sparse = _mlp_sample(blob, W, x, log_std, cnt=cnt)
    for a, b in zip(dense, sparse):
        assert torch.equal(a, b)
    z_emu, v_emu = _mlp_emulation(ac, x)
    assert float((sparse[3] - torch.tanh(z_emu)).abs().max()) < 1e-2 and float((sparse[2] - v_emu).abs().max()) < 1e-2
    L = _lib.lib()
    assert L.rvo3d_policy_mlp_sample(_p(blob), W, _p(x), W, rows, _p(cnt), W + 1, 9, 1, _p(log_std), 1.0, 0, 0,
                                     _p(dense[0]), _p(dense[1]), _p(dense[2]), None, None, None) == -1


def test_policy_mlp_sample_reads_only_the_callers_bytes_and_rejects_bad_arguments():
    """The 16-wide k-steps run past the last row's end: the kernel reads through a buffer descriptor of exactly
    (rows - 1) ld + width floats.  Here the observation array ends flush with its allocation and is followed by NaNs in
    a second check - neither may change a result."""
    L = _lib.lib()
    W, rows = 102, 200
    ac = mlp_ac(W).to(DEV)
    blob = _mlp_pack(ac, W)
    log_std = torch.zeros(3, device=DEV)
    big = torch.full((rows * W + 64,), float("nan"), device=DEV)
    x = big[:rows * W].view(rows, W)
    x.copy_(torch.randn((rows, W), device=DEV))
    a = _mlp_sample(blob, W, x, log_std)
    b = _mlp_sample(blob, W, x.clone(), log_std)
    assert all(torch.equal(p, q) for p, q in zip(a, b)) and bool(torch.isfinite(a[3]).all())
    # a prefix of the rows gives the prefix of the results (ragged last 64-row group)
    c = _mlp_sample(blob, W, x, log_std, rows=77)
    assert torch.equal(c[3], a[3][:77]) and torch.equal(c[2], a[2][:77])
    v = torch.zeros(8, device=DEV)
    args = lambda **k: [k.get("blob", _p(blob)), k.get("W", W), _p(x), k.get("ld", W), k.get("rows", rows), None, 12, 9, 1, _p(log_std),
                        1.0, 0, 0, _p(v), _p(v), _p(v), None, None, None]
    assert L.rvo3d_policy_mlp_sample(*args(W=127)) == -1 and b"obs_width" in L.rvo3d_last_error()
    assert L.rvo3d_policy_mlp_sample(*args(ld=W - 1)) == -1
    assert L.rvo3d_policy_mlp_sample(*args(blob=None)) == -1
    assert L.rvo3d_policy_mlp_sample(*args(rows=-1)) == -1
    assert L.rvo3d_policy_mlp_sample(*args(blob=C.c_void_p(blob.data_ptr() + 4))) == -1
    assert L.rvo3d_policy_mlp_sample(*args(rows=0)) == 0
    assert L.rvo3d_policy_mlp_blob_bytes(0) == -1 and L.rvo3d_policy_mlp_blob_bytes(102) == 2 * (7 * 8192 + 131072 + 1024 + 2048 + 16)


def test_policy_sample_direct_mode_and_std_factor():
    """hidden = 0: mu / v come from the caller's own network (the biGRU actor-critic); std_factor as the
    evaluator uses it (post_train.py: std_factor 1e-3 -> std clamps at 1e-4 + ...)."""
    rows = 5000
    g = torch.Generator(device=DEV).manual_seed(1)
    mu_in = torch.tanh(torch.randn((rows, 3), device=DEV, generator=g))
    v_in = torch.randn((rows, 1), device=DEV, generator=g)
    log_std = torch.tensor([-1.0, -1.0, -1.0], device=DEV)
    for sf in (1.0, 1e-3):
        act, logp, val, mu, raw = _sample(mu_in, v_in, None, None, None, None, log_std, rows, 0, _lib.RVO3D_F32,
                                          std_factor=sf)
        assert torch.equal(mu, mu_in) and torch.equal(val, v_in[:, 0])
        std = torch.clamp(sf * torch.exp(log_std) + 1e-6, 1e-4, 10.0)
        lp_ref = torch.distributions.Normal(mu, std).log_prob(raw).sum(-1)
        assert torch.allclose(logp, lp_ref, atol=2e-3 if sf < 1 else 1e-4, rtol=1e-5)
        assert float(((raw - mu) / std).std()) == pytest.approx(1.0, abs=0.03)


def test_policy_sample_rejects_bad_arguments():
    L = _lib.lib()
    x = torch.zeros((8, 300), device=DEV)
    ls = torch.zeros(3, device=DEV)
    hd = _lib.PolicyHeads(x.data_ptr(), x.data_ptr(), 300, 300, _lib.RVO3D_F32, 300, 1, 0, x.data_ptr(),
                          x.data_ptr(), x.data_ptr(), x.data_ptr(), ls.data_ptr())
    assert L.rvo3d_policy_sample(C.byref(hd), 8, 1.0, 0, 0, _p(x), _p(x), _p(x), None, None, None) == -1
    assert b"hidden" in L.rvo3d_last_error()
    assert L.rvo3d_policy_sample(None, 8, 1.0, 0, 0, _p(x), _p(x), _p(x), None, None, None) == -1


def _account_reference(rew, done, fin, ep_ret, ep_len, sanitize, max_ep_len, epoch_end):
    """multi_ppo.collect()'s bookkeeping (the module path), statement by statement."""
    rew_fin = torch.nan_to_num(rew, nan=0.0, posinf=0.0, neginf=0.0)
    slot = rew_fin if sanitize else rew
    ep_ret = ep_ret + rew_fin
    ep_len = ep_len + 1
    by_step = (done | fin) != 0
    timeout = ep_len > max_ep_len
    if epoch_end:
        ended = torch.ones_like(by_step); terminal = torch.ones(rew.shape[0], dtype=torch.bool, device=rew.device)
        extra = ~by_step
    else:
        ended = by_step | timeout
        terminal = ((fin != 0) | timeout).any(dim=1)
        extra = timeout & ~by_step
    s = float((ep_ret.double() * ended).sum()); n = float(ended.sum())
    return slot, ep_ret.masked_fill(ended, 0.0), ep_len.masked_fill(ended, 0), terminal, extra, s, n


@pytest.mark.parametrize("E,N", [(37, 5), (16, 64), (9, 100), (3, 300)])
def test_rollout_account_matches_the_module_path(E, N):
    L = _lib.lib()
    g = torch.Generator(device=DEV).manual_seed(E * N)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ep_ret = torch.randn((E, N), device=DEV, generator=g)
    ep_len = torch.randint(0, 12, (E, N), device=DEV, generator=g, dtype=torch.int32)
    for it, (sanitize, epoch_end) in enumerate([(1, 0), (0, 0), (1, 0), (1, 1)]):
        rew = torch.randn((E, N), device=DEV, generator=g)
        rew[torch.rand((E, N), device=DEV, generator=g) < 0.05] = float("inf")
        rew[torch.rand((E, N), device=DEV, generator=g) < 0.05] = float("nan")
        rew[torch.rand((E, N), device=DEV, generator=g) < 0.02] = float("-inf")
        done = (torch.rand((E, N), device=DEV, generator=g) < 0.05).to(torch.uint8)
        fin = (torch.rand((E, N), device=DEV, generator=g) < 0.01).to(torch.uint8)
        want = _account_reference(rew, done, fin, ep_ret, ep_len, sanitize, 10, epoch_end)
        slot = torch.zeros((E, N), device=DEV)
        cut = torch.zeros(E, dtype=torch.uint8, device=DEV)
        extra = torch.full((E, N), 7, dtype=torch.uint8, device=DEV)
        sums = torch.zeros((E, 2), dtype=torch.float64, device=DEV)
        anyx = torch.zeros(1, dtype=torch.int32, device=DEV)
        _lib.check(L.rvo3d_rollout_account(E, N, _p(rew), _p(done), _p(fin), sanitize, 10, epoch_end, _p(slot),
                                           _p(ep_ret), _p(ep_len), _p(cut), _p(extra), _p(sums), _p(anyx), st),
                   "rvo3d_rollout_account")
        torch.cuda.synchronize()
        assert torch.equal(torch.nan_to_num(slot, nan=-7.0), torch.nan_to_num(want[0], nan=-7.0))
        assert torch.equal(ep_ret, want[1]) and torch.equal(ep_len, want[2])
        assert torch.equal(cut.bool(), want[3]) and torch.equal(extra.bool(), want[4])
        assert float(sums[:, 0].sum()) == pytest.approx(want[5], rel=1e-12, abs=1e-9) and float(sums[:, 1].sum()) == want[6]
        assert bool(anyx.item()) == bool(want[4].any())


@pytest.mark.parametrize("hidden,bi,dt", [(256, True, torch.float32), (256, True, torch.bfloat16), (128, False, torch.float32),
                                          (64, True, torch.float32), (192, True, torch.bfloat16)])
def test_reader_first_step_matches_the_module(hidden, bi, dt):
    """rvo3d_reader_first_step against rnn_Reader.forward_batch (policy_rnn_ac.py:75-127) on rows with one VO row:
    float32 statement of the same GRU cell / sum / concat / LayerNorm; tolerance 3e-5 absolute (fast exp / rcp in
    the kernel's sigmoid and tanh; features are O(1) after the LayerNorm), bf16 output: one bf16 ulp on top."""
    from rvo3d_amd.policy.policy_rnn_ac import rnn_Reader
    torch.manual_seed(hidden)
    rows, W = 4099, 102   # ragged: not a multiple of the eight rows per trip
    r = rnn_Reader(12, 9, hidden, use_gpu=False, mode="biGRU" if bi else "GRU").cuda()
    with torch.no_grad():   # away from the default init so that every term matters
        for p_ in r.parameters():
            p_.add_(torch.randn_like(p_) * 0.3)
    obs = torch.randn((rows, W), device=DEV)
    obs[:, 21:] = 0.0
    obs[::7, 12:21] = 0.0   # drones without any VO row: the reference's single all-zero row
    # ... which is what nearly every row of a rollout looks like: the kernel computes that row's hidden state once per
    # launch.  Whole trips of eight such rows, trips with one row that has a VO row, a long run of them:
    obs[800:1600, 12:21] = 0.0
    obs[808, 12:21] = torch.randn(9, device=DEV)
    obs[2000:4099, 12:21] = 0.0
    obs[3001, 15] = 0.25
    with torch.no_grad():
        want = r.forward_batch(obs, torch.ones(rows, dtype=torch.int64, device=DEV))
    D = 12 + hidden
    ld = (D + 63) // 64 * 64
    feat = torch.full((rows, ld), 7.0, dtype=dt, device=DEV)
    g = r.rnn_net
    f = lambda t: t.detach().float().contiguous()
    w = [f(g.weight_ih_l0), f(g.bias_ih_l0), f(g.bias_hh_l0)] + \
        ([f(g.weight_ih_l0_reverse), f(g.bias_ih_l0_reverse), f(g.bias_hh_l0_reverse)] if bi else [None] * 3) + \
        [f(r.ln.weight), f(r.ln.bias)]
    st = _lib.GruReader(*[None if t is None else t.data_ptr() for t in w], hidden, 9, 12, float(r.ln.eps))
    L = _lib.lib()
    _lib.check(L.rvo3d_reader_first_step(C.byref(st), _p(obs), W, rows, _p(feat),
                                         _lib.RVO3D_BF16 if dt == torch.bfloat16 else _lib.RVO3D_F32, ld,
                                         C.c_void_p(torch.cuda.current_stream().cuda_stream)), "rvo3d_reader_first_step")
    torch.cuda.synchronize()
    got = feat[:, :D].float()
    tol = 3e-5 if dt == torch.float32 else 3e-5 + 2 ** -8 * float(want.abs().max())
    assert float((got - want).abs().max()) < tol, float((got - want).abs().max())
    assert bool((feat[:, D:] == 7.0).all())   # the caller's padding is not touched
    # the shortcut runs the same instructions on the same (zero) inputs: a zero row's hidden part is the same bits
    # whether its trip took the shortcut (rows 2000..) or computed it beside a row with a VO row (row 7 of trip 0)
    zero_rows = (obs[:, 12:21] == 0).all(dim=1)
    hid = (feat[:, 12:D].float() - want[:, 12:D]).abs()
    assert float(hid[zero_rows].max()) < tol
    bad = _lib.GruReader(*[None if t is None else t.data_ptr() for t in w], 100, 9, 12, 1e-5)
    assert L.rvo3d_reader_first_step(C.byref(bad), _p(obs), W, rows, _p(feat), 0, ld, None) == -1


def test_collapsed_first_layer_of_rows_without_vo_rows():
    """rvo3d_reader_zero_features + rvo3d_policy_mlp_sample with rnn_ac.zero_vo_plan()'s weights against the module's own
    float32 forward (biGRU reader, LayerNorm, (256, 256) heads) on rows without a velocity-obstacle row: the GRU state of
    such rows is a constant, the LayerNorm enters through mean and rstd, the first layer is a product over 12 + 8
    inputs.  Tolerance: bf16 operands in the three products, as for the MLP policy (3e-2 on mu / v at these weights);
    the split of rstd, mean rstd and the collapsed columns into bf16 head + tail keeps the collapse itself at 1e-4."""
    from rvo3d_amd.policy import rnn_ac

    class Space:
        shape = (3,)
    torch.manual_seed(11)
    ac = rnn_ac(None, Space(), 12, 9, 256, (256, 256), (256, 256), torch.nn.ReLU, torch.nn.Tanh, torch.nn.Identity,
                use_gpu=False, rnn_mode="biGRU").cuda()
    with torch.no_grad():
        for p_ in ac.pi.rnn_reader.parameters():
            p_.add_(torch.randn_like(p_) * 0.2)
    zp = ac.zero_vo_plan()
    assert zp is not None and zp["width"] == 20
    rows, W = 5000, 102
    obs = torch.zeros((rows, W), device=DEV)
    obs[:, :12] = torch.randn((rows, 12), device=DEV) * torch.tensor([30., 30, 5, 2, 2, 2, 1, 1, 1, 1, 1, 1], device=DEV)
    cnt = torch.zeros(rows, dtype=torch.int32, device=DEV)
    L = _lib.lib()
    f0 = torch.empty((rows, 20), device=DEV)
    _lib.check(L.rvo3d_reader_zero_features(_p(obs), W, rows, 12, 268, _p(zp["ln_w"]), _p(zp["ln_b"]), zp["sum_h0"],
                                            zp["sumsq_h0"], zp["eps"], _p(f0), 20, None, None, None, None),
               "rvo3d_reader_zero_features")
    act, logp, val, mu, raw = _mlp_sample(zp["blob"], 20, f0, ac.log_std.detach())
    with torch.no_grad():
        d, _ = ac.pi((obs, cnt))
        v = ac.v((obs, cnt))
        # the collapse itself, in float64: first-layer pre-activations from the 20 inputs vs from the 268 features
        feat = ac.pi.rnn_reader.forward_batch(obs, torch.ones(rows, dtype=torch.int64, device=DEV)).double()
        lin = [m for m in ac.pi.net_out if isinstance(m, torch.nn.Linear)][0]
        z_full = feat @ lin.weight.double().T + lin.bias.double()
    assert float((mu - d.mean).abs().max()) < 3e-2 and float((val - v).abs().max()) < 3e-2
    # the 20 inputs against the module's LayerNorm: f_p, and rstd / mean rstd recombined from head + tail
    assert float((f0[:, :12].double() - feat[:, :12]).abs().max()) < 1e-4
    assert bool((f0[:, 18:] == 1).all()) and bool((f0[:, 12] == f0[:, 13]).all())
    # reconstruct the first layer from the collapsed columns exactly as the kernel's bf16 products see them
    h0 = (ac.pi.rnn_reader._gru_first(torch.zeros((1, 9), device=DEV), "")
          + ac.pi.rnn_reader._gru_first(torch.zeros((1, 9), device=DEV), "_reverse")).reshape(-1).double().detach()
    g, bt = ac.pi.rnn_reader.ln.weight.double().detach(), ac.pi.rnn_reader.ln.bias.double().detach()
    Wh = lin.weight.double()[:, 12:].detach()
    a, b, c = Wh @ (h0 * g[12:]), Wh @ g[12:], Wh @ bt[12:] + lin.bias.double().detach()
    r = (f0[:, 12] + f0[:, 14]).double(); m = (f0[:, 15] + f0[:, 17]).double()
    z_col = feat[:, :12] @ lin.weight.double()[:, :12].T.detach() + r[:, None] * a[None, :] - m[:, None] * b[None, :] + c[None, :]
    assert float((z_col - z_full.detach()).abs().max()) < 1e-4 * max(1.0, float(z_full.abs().max()))
    bad = L.rvo3d_reader_zero_features(_p(obs), W, rows, 12, 268, _p(zp["ln_w"]), _p(zp["ln_b"]), 0.0, 0.0, 1e-5, _p(f0), 19,
                                       None, None, None, None)
    assert bad == -1
    assert L.rvo3d_reader_zero_features(_p(obs), W, rows, 12, 268, _p(zp["ln_w"]), _p(zp["ln_b"]), 0.0, 0.0, 1e-5, _p(f0), 20,
                                        _p(cnt), None, None, None) == -1     # counts without a list


@pytest.mark.parametrize("bi", [True, False])
def test_policy_rows_matches_the_modules(bi):
    """rvo3d_policy_rows - the policy step of the rows that DO have velocity-obstacle rows, one workgroup per row,
    float32 - against the modules' own forward (rnn_Reader's recurrence over 1..nm VO rows, both directions, LayerNorm,
    actor / critic stacks): mu and v to 1e-4 (summation order), and the list / counter protocol: the rows are found
    by rvo3d_reader_zero_features from vo_count, every listed row is overwritten, no other row is touched, the
    counter is back at zero afterwards."""
    from rvo3d_amd.policy import rnn_ac

    class Space:
        shape = (3,)
    torch.manual_seed(5 + bi)
    ac = rnn_ac(None, Space(), 12, 9, 256, (256, 256), (256, 256), torch.nn.ReLU, torch.nn.Tanh, torch.nn.Identity,
                use_gpu=False, rnn_mode="biGRU" if bi else "GRU").cuda()
    with torch.no_grad():
        for p_ in ac.pi.rnn_reader.parameters():
            p_.add_(torch.randn_like(p_) * 0.2)
    zp = ac.zero_vo_plan()
    rows, nm = 3000, 10
    W = 12 + 9 * nm
    g = torch.Generator(device=DEV).manual_seed(3)
    cnt = torch.zeros(rows, dtype=torch.int32, device=DEV)
    pick = torch.randperm(rows, device=DEV, generator=g)[:700]
    cnt[pick] = torch.randint(1, nm + 1, (700,), device=DEV, generator=g, dtype=torch.int32)
    obs = torch.randn((rows, W), device=DEV, generator=g)
    obs *= (torch.arange(W, device=DEV)[None, :] < (12 + 9 * cnt.long())[:, None]).float()
    L = _lib.lib()
    f0 = torch.empty((rows, 20), device=DEV)
    lst = torch.zeros(rows, dtype=torch.int32, device=DEV)
    ctr = torch.zeros(2, dtype=torch.int32, device=DEV)
    log_std = ac.log_std.detach()
    act = torch.full((rows, 3), 9.0, device=DEV); logp = torch.full((rows,), 9.0, device=DEV); val = torch.full((rows,), 9.0, device=DEV)
    for rep in range(2):      # twice: the second call finds the counter reset by the first
        _lib.check(L.rvo3d_reader_zero_features(_p(obs), W, rows, 12, 268, _p(zp["ln_w"]), _p(zp["ln_b"]), zp["sum_h0"],
                                                zp["sumsq_h0"], zp["eps"], _p(f0), 20, _p(cnt), _p(lst), _p(ctr), None),
                   "rvo3d_reader_zero_features")
        torch.cuda.synchronize()
        assert int(ctr[0]) == 700 and sorted(lst[:700].tolist()) == sorted(pick.tolist())
        net = zp["rows_net"]; net.slots = nm
        _lib.check(L.rvo3d_policy_rows(C.byref(net), _p(obs), W, _p(cnt), _p(lst), _p(ctr), C.c_void_p(ctr.data_ptr() + 4), 1,
                                       _p(log_std), 1.0, 7, rep, _p(act), _p(logp), _p(val), None), "rvo3d_policy_rows")
        torch.cuda.synchronize()
        assert ctr.tolist() == [0, 0]
    untouched = torch.ones(rows, dtype=torch.bool, device=DEV); untouched[pick] = False
    assert bool((val[untouched] == 9.0).all()) and bool((act[untouched] == 9.0).all())
    with torch.no_grad():
        arg = (obs[pick], cnt[pick])
        d, _ = ac.pi(arg)
        v = ac.v(arg)
    assert float((val[pick] - v).abs().max()) < 1e-4
    std = torch.clamp(torch.exp(log_std) + 1e-6, 1e-4, 10.0)
    # the stored action is round(mu + std eps, 2) with the generator's eps for (row, call 1): recover mu from it
    zero = torch.zeros((rows, 3), device=DEV)
    _, _, _, _, eps_std = _sample(zero, torch.zeros((rows, 1), device=DEV), None, None, None, None, log_std, rows, 0,
                                  _lib.RVO3D_F32, step=1)
    mu_rec = act[pick] - eps_std[pick]
    assert float((mu_rec - d.mean).abs().max()) < 5.1e-3 + 1e-4      # np.round(a, 2): half a cent
    lp = torch.distributions.Normal(d.mean, std).log_prob(d.mean + eps_std[pick]).sum(-1)
    assert float((logp[pick] - lp).abs().max()) < 2e-3


@pytest.mark.parametrize("amp,kind", [(False, "mlp"), (True, "mlp"), (True, "mlp_gemm"), (False, "rnn"), (False, "mlp_small"),
                                      (False, "rnn256"), (True, "rnn256"), (True, "rnn256_gemm")])
def test_fused_rollout_is_a_faithful_rollout(amp, kind):
    """The fused loop (multi_ppo._collect_fused) on 16 drones x 64 envs: (a) a second env stepped with the
    STORED actions reproduces every stored observation, count and reward bit for bit - the buffer holds
    what the env really did; (b) the stored values are the critic's, the stored log-probabilities and
    actions are consistent with the actor's distribution on the stored observations; (c) cuts and
    episode statistics equal the module path's bookkeeping applied to the same flags."""
    E, N, T = 64, 16, 24
    # (the biGRU actor-critic's bf16 rollout in a tighter box: its "rnn0" mode has a kernel of its own for the rows
    # that have velocity-obstacle rows - there should be some)
    world = synthetic_world(E, N, (9, 9, 5) if (kind == "rnn256" and amp) else (20, 20, 8), n_points=3, seed=4)
    env = BatchedDroneEnv(world)
    torch.manual_seed(0)
    if kind.startswith("rnn"):
        # the reference's architecture (biGRU reader).  "rnn": widths without a kernel instantiation - its own forward,
        # then the "direct" mode; "rnn256": the trained shape (256 / (256, 256)) - reader kernel + heads kernel
        from rvo3d_amd.policy import rnn_ac

        class Space:
            shape = (3,)
        hs, mh = (32, (64, 64)) if kind == "rnn" else (256, (256, 256))   # ("rnn256_gemm": library GEMMs, see below)
        ac = rnn_ac(None, Space(), 12, 9, hs, mh, mh, torch.nn.ReLU, torch.nn.Tanh, torch.nn.Identity,
                    use_gpu=False, rnn_mode="biGRU").cuda()
    else:               # (64, 64): a hidden width the heads kernel has no instantiation for -> "direct" as well
        ac = mlp_ac(env.W, hidden_sizes=(64, 64) if kind == "mlp_small" else (256, 256)).cuda()
    # bf16 + MLP(256, 256): the whole policy step is rvo3d_policy_mlp_sample ("mlp"); "mlp_gemm" keeps the library-GEMM
    # path of the same shape alive (other widths / float32 use it)
    tr = multi_ppo(env, ac, steps_per_epoch=T, max_ep_len=9, train_pi_iters=1, train_v_iters=1, amp=amp, seed=3,
                   fused_mlp=kind not in ("mlp_gemm", "rnn256_gemm"))
    # bf16 + the biGRU actor-critic with (256, 256) heads: "rnn0" - rows without a VO row through the collapsed first
    # layer, the others (many in this dense little world) through the general path on a gathered batch
    assert tr._fused_mode() == ("mlp" if (kind == "mlp" and amp) else "rnn0" if (kind == "rnn256" and amp)
                                else "heads" if kind in ("mlp", "mlp_gemm", "rnn256", "rnn256_gemm") else "direct")
    env.reset(); env.observe()
    mean_ret = tr.collect()
    buf = tr.buf
    assert buf.ptr == T
    # (a) replay
    env2 = BatchedDroneEnv(world)
    env2.reset(); o, c = env2.observe()
    assert torch.equal(o, buf.obs[0]) and torch.equal(c, buf.cnt[0])
    ep_len = torch.zeros((E, N), dtype=torch.int32, device=DEV)
    ep_ret = torch.zeros((E, N), device=DEV)
    ret_sum = ret_n = 0.0
    for t in range(T):
        o, c, rew, done, info, fin = env2.step_policy(buf.act[t], autoreset=True)
        want = _account_reference(rew, done, fin, ep_ret, ep_len, 1, 9, t == T - 1)
        ep_ret, ep_len = want[1], want[2]
        ret_sum += want[5]; ret_n += want[6]
        assert torch.equal(torch.nan_to_num(buf.rew[t], nan=-7.0), torch.nan_to_num(want[0], nan=-7.0)), t
        assert torch.equal(buf.cut[t], want[3]), t
        if bool(want[4].any()):
            env2.reset_drones(want[4]); o, c = env2.observe()
        assert torch.equal(torch.nan_to_num(o, nan=-7.0), torch.nan_to_num(buf.obs[t + 1], nan=-7.0)), t
        assert torch.equal(c, buf.cnt[t + 1]), t
    assert mean_ret == pytest.approx(ret_sum / max(ret_n, 1.0), rel=1e-9, abs=1e-9)
    # (b) the stored numbers against the module's own float32 forward on the stored observations
    with torch.no_grad():
        x = buf.obs[:T].reshape(-1, env.W)
        arg = (x, buf.cnt[:T].reshape(-1)) if kind.startswith("rnn") else x
        d, _ = ac.pi(arg)
        v = ac.v(arg)
    tol = 3e-2 if amp else 1e-4  # bf16 GEMMs in the rollout vs the float32 module
    assert torch.allclose(buf.val.reshape(-1), v, atol=tol, rtol=tol)
    z = (buf.act.reshape(-1, 3) - d.mean) / d.stddev   # rounded action: + U(-0.005, 0.005) / std
    assert abs(float(z.mean())) < 0.02 and abs(float(z.var()) - 1) < 0.05
    lp_of_stored = d.log_prob(buf.act.reshape(-1, 3)).sum(-1)
    assert float((buf.logp.reshape(-1) - lp_of_stored).abs().mean()) < (0.12 if amp else 0.06)
    # (c) the update runs on it
    st = tr.update(buf.get())
    assert np.isfinite(st["loss_v"])
    if kind == "rnn256" and amp:
        # the one-workgroup-per-row kernel is for short lists: a rollout in which more than 1 row in 500 has VO rows
        # sends the next one back to the library-GEMM path - which works
        frac = float((buf.cnt[:T] > 0).float().mean())
        assert frac > 1e-3, frac                       # rvo3d_policy_rows had rows to do in this rollout
        assert tr._rnn0_dense == (frac > 2e-3)
        tr._rnn0_dense = True
        assert tr._fused_mode() == "heads"
        tr.buf.ptr = 0
        assert np.isfinite(tr.collect())
    env.close(); env2.close()


@pytest.mark.parametrize("kind", ["mlp", "rnn256"])
def test_graph_replayed_rollouts_are_faithful_and_draw_new_noise(kind):
    """With multi_ppo(graph_rollout=True) the fast paths replay their per-step launches as HIP graphs from the second rollout on (
    the noise counter then lives in device memory, rvo3d_rollout_set_step_counter).  Three rollouts - eager, capture +
    replay, replay: each must be a faithful rollout (a second env stepped with the STORED actions of all three
    reproduces every stored observation and reward bit for bit), the stored values
    must be the critic's, and the replays must draw new noise (other actions than the rollout before, although the
    graph's arguments are the same bytes)."""
    E, N, T = 32, 16, 12
    world = synthetic_world(E, N, (12, 12, 6), n_points=3, seed=9)
    env, env2 = BatchedDroneEnv(world), BatchedDroneEnv(world)
    torch.manual_seed(1)
    if kind == "mlp":
        ac = mlp_ac(env.W).cuda()
    else:
        from rvo3d_amd.policy import rnn_ac

        class Space:
            shape = (3,)
        ac = rnn_ac(None, Space(), 12, 9, 256, (256, 256), (256, 256), torch.nn.ReLU, torch.nn.Tanh, torch.nn.Identity,
                    use_gpu=False, rnn_mode="biGRU").cuda()
    tr = multi_ppo(env, ac, steps_per_epoch=T, max_ep_len=50, train_pi_iters=1, train_v_iters=1, amp=True, seed=3,
                   graph_rollout=True)
    env.reset(); env.observe()
    env2.reset(); o, c = env2.observe()
    acts = []
    for rollout in range(3):
        tr.buf.ptr = 0
        tr.collect(final_reset=False)
        if kind == "rnn256":
            tr._rnn0_dense = False        # (keep the mode under test whatever this little world's density)
        if not getattr(tr, "_graph_failed", False):   # (a runtime that refuses the capture: eager launches, still faithful)
            assert (len(getattr(tr, "_graphs", {})) > 0) == (rollout >= 1)
        buf = tr.buf
        # (env2 follows env through all three rollouts: its last observation is this rollout's first)
        assert torch.equal(torch.nan_to_num(o, nan=-7.0), torch.nan_to_num(buf.obs[0], nan=-7.0)) and torch.equal(c, buf.cnt[0])
        for t in range(T):
            o, c, rew, done, info, fin = env2.step_policy(buf.act[t], autoreset=True)
            assert torch.equal(torch.nan_to_num(o, nan=-7.0), torch.nan_to_num(buf.obs[t + 1], nan=-7.0)), (rollout, t)
            assert torch.equal(torch.nan_to_num(buf.rew[t], nan=-7.0),
                               torch.nan_to_num(torch.nan_to_num(rew, nan=0.0, posinf=0.0, neginf=0.0), nan=-7.0))
        with torch.no_grad():
            x = buf.obs[:T].reshape(-1, env.W)
            arg = (x, buf.cnt[:T].reshape(-1)) if kind == "rnn256" else x
            v = ac.v(arg)
            d, _ = ac.pi(arg)
        assert torch.allclose(buf.val.reshape(-1), v, atol=3e-2, rtol=3e-2)
        z = (buf.act.reshape(-1, 3) - d.mean) / d.stddev
        assert abs(float(z.mean())) < 0.03 and abs(float(z.var()) - 1) < 0.08
        acts.append(buf.act.clone())
    # new noise in every rollout and in every step of a rollout (the residuals of two steps are not the same numbers)
    assert not torch.equal(acts[1], acts[2]) and not torch.equal(acts[0], acts[1])
    with torch.no_grad():
        res = (tr.buf.act - d.mean.view(T, E, N, 3))
    assert not torch.allclose(res[0], res[1], atol=1e-3)
    L = _lib.lib()
    assert L.rvo3d_rollout_set_step_counter(None) == 0
    env.close(); env2.close()
    if getattr(tr, "_graph_failed", False):
        pytest.skip("this runtime refused the graph capture: the rollouts ran (and were checked) with stream launches")
