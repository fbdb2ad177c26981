#!/usr/bin/env python3
"""usage: FUZZ_SEED=1 FUZZ_CASES=200 python tools/fuzz_policy.py       (GPU box)
Randomised shapes for the rollout's policy kernels: rvo3d_policy_mlp_sample at random observation widths (1..126), row
counts (1..70 000, ragged), row strides, env-shaped sparsity with and without counts (must be the same bits), against
a PyTorch emulation with the kernel's rounding points; rvo3d_reader_zero_features + rvo3d_policy_rows at random row
counts / VO-row counts against the modules' float32 forward.  Not part of the test suite; fails on the first mismatch."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "3drvo-marl-collisionavoidance_amd")]
import test_gpu_rollout as T  # noqa: E402
from rvo3d_amd import _lib  # noqa: E402
from rvo3d_amd.policy import mlp_ac, rnn_ac  # noqa: E402

seed, cases = int(os.environ.get("FUZZ_SEED", "1")), int(os.environ.get("FUZZ_CASES", "200"))
rng = np.random.default_rng(seed)
DEV = "cuda"
L = _lib.lib()
worst = 0.0
for c in range(cases):
    W = int(rng.integers(1, 127))
    rows = int(rng.choice([rng.integers(1, 130), rng.integers(130, 5000), rng.integers(5000, 70000)]))
    pad = int(rng.integers(0, 9))
    torch.manual_seed(int(rng.integers(1 << 30)))
    ac = mlp_ac(W).to(DEV)
    x = (torch.randn((rows + 1, W + pad), device=DEV) * float(rng.uniform(0.3, 3.0)))[:rows, :W]
    log_std = torch.tensor(rng.uniform(-2, 0, 3), dtype=torch.float32, device=DEV)
    blob = T._mlp_pack(ac, W)
    cnt = None
    if W >= 21 and rng.random() < 0.6:
        nmax = (W - 12) // 9
        cnt = torch.as_tensor(rng.choice([0, 0, 0, 1, 2, nmax], rows), dtype=torch.int32, device=DEV).clamp(max=nmax)
        x = x * (torch.arange(W, device=DEV)[None, :] < (12 + 9 * cnt.long())[:, None]).float()
    act, logp, val, mu, raw = T._mlp_sample(blob, W, x, log_std, seed=int(rng.integers(1 << 40)), step=int(rng.integers(1 << 20)))
    z, v = T._mlp_emulation(ac, x)
    e = max(float((mu - torch.tanh(z)).abs().max()), float((val - v).abs().max()))
    worst = max(worst, e)
    assert e < 2e-2, (c, W, rows, e)
    assert np.array_equal(act.cpu().numpy(), np.round(raw.cpu().numpy(), 2)) and bool(torch.isfinite(logp).all())
    if cnt is not None:
        b = T._mlp_sample(blob, W, x, log_std, cnt=cnt)
        a = T._mlp_sample(blob, W, x, log_std)
        assert all(torch.equal(p, q) for p, q in zip(a, b)), (c, W, rows)
print(f"policy_mlp fuzz ok: {cases} cases, seed {seed}, worst |kernel - emulation| {worst:.2e}")


class Space:
    shape = (3,)


for c in range(max(cases // 20, 4)):
    bi = bool(rng.integers(2))
    nm = int(rng.integers(1, 11))
    torch.manual_seed(int(rng.integers(1 << 30)))
    ac = rnn_ac(None, Space(), 12, 9, 256, (256, 256), (256, 256), torch.nn.ReLU, torch.nn.Tanh, torch.nn.Identity,
                use_gpu=False, rnn_mode="biGRU" if bi else "GRU").cuda()
    zp = ac.zero_vo_plan()
    rows, W = int(rng.integers(1, 4000)), 12 + 9 * nm
    cnt = torch.as_tensor(rng.choice(list(range(nm + 1)) + [0] * 6, rows), dtype=torch.int32, device=DEV)
    obs = torch.randn((rows, W), device=DEV) * (torch.arange(W, device=DEV)[None, :] < (12 + 9 * cnt.long())[:, None]).float()
    f0 = torch.empty((rows, 20), device=DEV)
    lst = torch.zeros(rows, dtype=torch.int32, device=DEV); ctr = torch.zeros(2, dtype=torch.int32, device=DEV)
    p = T._p
    _lib.check(L.rvo3d_reader_zero_features(p(obs), W, rows, 12, 268, p(zp["ln_w"]), p(zp["ln_b"]), zp["sum_h0"], zp["sumsq_h0"],
                                            zp["eps"], p(f0), 20, p(cnt), p(lst), p(ctr), None), "rvo3d_reader_zero_features")
    act, logp, val, mu, raw = T._mlp_sample(zp["blob"], 20, f0, ac.log_std.detach())
    net = zp["rows_net"]; net.slots = nm
    _lib.check(L.rvo3d_policy_rows(C.byref(net), p(obs), W, p(cnt), p(lst), p(ctr), C.c_void_p(ctr.data_ptr() + 4), 1,
                                   p(ac.log_std.detach()), 1.0, 7, 0, p(act), p(logp), p(val), None), "rvo3d_policy_rows")
    torch.cuda.synchronize()
    with torch.no_grad():
        v = ac.v((obs, cnt))
    has = cnt > 0
    assert ctr.tolist() == [0, 0]
    if bool(has.any()):
        assert float((val[has] - v[has]).abs().max()) < 2e-4, (c, float((val[has] - v[has]).abs().max()))
    if bool((~has).any()):
        assert float((val[~has] - v[~has]).abs().max()) < 3e-2
print(f"reader_zero_features / policy_rows fuzz ok: {max(cases // 20, 4)} cases")
