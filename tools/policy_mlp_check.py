#!/usr/bin/env python3
"""usage: tools/policy_mlp_check.py [library.so]      (MLP_CLOCK=1 with a -DRVO3D_MLP_CLOCK=1 build: per-wave timeline)
rvo3d_policy_mlp_sample (csrc/rvo3d_policy_mlp.hpp) on the GPU box: results against a PyTorch emulation with the
kernel's rounding points and against the float32 module for several widths / row counts, then the time per call at
config 3's size (64 x 4096 rows, width 102).  With a library built with -DRVO3D_MLP_CLOCK=1 (tools/bench_variant.py
style: rvo3d_amd._lib.build_hip(out=..., extra_flags=["-DRVO3D_MLP_CLOCK=1"])) and MLP_CLOCK=1 it also prints every
wave's start / end (100 MHz ticks), the shader clock, and the cycles per pass spent in row loads + conversion, layer 1,
layer 2 and sampling - the numbers behind DESIGN.md's account of the kernel.  Not part of the product path."""

import os, sys, ctypes as C, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "3drvo-marl-collisionavoidance_amd"))
from rvo3d_amd import _lib
if len(sys.argv) > 1: _lib.use_library(os.path.abspath(sys.argv[1]))
from rvo3d_amd.policy import mlp_ac
L = _lib.lib()
torch.manual_seed(0)
dev = "cuda"
CNT = None
def pack(ac, W):
    nb = L.rvo3d_policy_mlp_blob_bytes(W)
    blob = torch.zeros(nb, dtype=torch.uint8, device=dev)
    keep = []
    def mw(net):
        lin = [m for m in net if isinstance(m, torch.nn.Linear)]
        t = [x.detach().float().contiguous() for l in lin for x in (l.weight, l.bias)]
        keep.extend(t)
        return _lib.MlpWeights(*[x.data_ptr() for x in t])
    a, b = mw(ac.pi_net), mw(ac.v_net)
    _lib.check(L.rvo3d_policy_mlp_pack(C.byref(a), C.byref(b), W, blob.data_ptr(), None), "pack")
    torch.cuda.synchronize()
    return blob
def emulate(ac, x):
    bf = torch.bfloat16
    def net(n):
        lin = [m for m in n if isinstance(m, torch.nn.Linear)]
        h = x.to(bf).float()
        h = torch.relu(h @ lin[0].weight.to(bf).float().T + lin[0].bias.to(bf).float()).to(bf).float()
        h = torch.relu(h @ lin[1].weight.to(bf).float().T + lin[1].bias.float()).to(bf).float()
        return h @ lin[2].weight.to(bf).float().T + lin[2].bias.float()
    return torch.tanh(net(ac.pi_net)), net(ac.v_net).squeeze(-1)
for (W, B) in [(102, 64), (102, 1000), (102, 262144), (57, 5000), (39, 777), (120, 4096), (12, 130)]:
    ac = mlp_ac(W).to(dev)
    with torch.no_grad():
        for p in ac.parameters(): p.mul_(3.0)      # larger weights: a sharper test
    blob = pack(ac, W)
    x = torch.randn(B, W, device=dev) * 2
    act = torch.zeros(B, 3, device=dev); logp = torch.zeros(B, device=dev); val = torch.zeros(B, device=dev)
    mu = torch.zeros(B, 3, device=dev); raw = torch.zeros(B, 3, device=dev)
    _lib.check(L.rvo3d_policy_mlp_sample(blob.data_ptr(), W, x.data_ptr(), x.stride(0), B, CNT, 12, 9, 1, ac.log_std.data_ptr(), 1.0,
                                         1234, 5, act.data_ptr(), logp.data_ptr(), val.data_ptr(), mu.data_ptr(),
                                         raw.data_ptr(), None), "sample")
    torch.cuda.synchronize()
    with torch.no_grad():
        emu, ev = emulate(ac, x)
        fmu = torch.tanh(ac.pi_net[:-1](x)) if False else ac.pi_net(x); fv = ac.v_net(x).squeeze(-1)
    print(f"W {W} B {B}: |mu - emu| {float((mu - emu).abs().max()):.2e}  |v - emu| {float((val - ev).abs().max()):.2e}"
          f"   vs fp32 module: mu {float((mu - fmu).abs().max()):.2e} v {float((val - fv).abs().max()):.2e}  (|v| max {float(fv.abs().max()):.2f})")
# timing at config 3's size
W, B = 102, 262144
ac = mlp_ac(W).to(dev); blob = pack(ac, W)
x = torch.randn(B, W, device=dev)
act = torch.zeros(B, 3, device=dev); logp = torch.zeros(B, device=dev); val = torch.zeros(B, device=dev)
def run(n):
    for i in range(n):
        L.rvo3d_policy_mlp_sample(blob.data_ptr(), W, x.data_ptr(), x.stride(0), B, CNT, 12, 9, 1, ac.log_std.data_ptr(), 1.0,
                                  1234, i, act.data_ptr(), logp.data_ptr(), val.data_ptr(), None, None, None)
def timed(tag):
    run(5); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(50); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    fl = 2 * B * 2 * (112 * 256 + 256 * 256 + 256 * 32)
    print(f"policy_mlp_sample {B} rows, {tag}: {us:.1f} us per call ({fl / us / 1e6:.0f} TFLOP/s counting every k-step)")
timed("dense rows (no counts)")
if os.environ.get("MLP_DENSE_ONLY"):      # (tools/pmc_policy_mlp.sh: counters of ONE regime)
    sys.exit(0)
# rows as a rollout of the bench world has them: the counts of a real env after 50 random steps
from rvo3d_amd import BatchedDroneEnv, synthetic_world
env = BatchedDroneEnv(synthetic_world(4096, 64, (50, 50, 10)))
env.reset(); env.observe()
for _ in range(50):
    env.step_policy(torch.rand((4096, 64, 3), device=dev) * 2 - 1, autoreset=True)
x = env.obs.view(B, W).clone(); CNT = env.vo_count.view(B).clone().data_ptr(); cnt_t = env.vo_count.view(B)
hist = torch.bincount(cnt_t.long().clamp(min=0), minlength=4)[:4].tolist()
print(f"env rows after 50 steps: vo_count 0 / 1 / 2 / 3: {hist} of {B}")
timed("the env's rows with their counts")
CNT = None
timed("the env's rows without counts")
if os.environ.get("MLP_CLOCK"):
    torch.cuda.synchronize()
    a = act[0].tolist()
    print(f"wave life {a[0]:.0f} shader cycles over {a[1] * 10:.0f} ns: {a[0] / (a[1] * 10):.3f} GHz")
if os.environ.get("MLP_CLOCK"):
    import numpy as np
    dbg = torch.zeros(B * 3, device=dev)
    L.rvo3d_policy_mlp_sample(blob.data_ptr(), W, x.data_ptr(), x.stride(0), B, CNT, 12, 9, 1, ac.log_std.data_ptr(), 1.0,
                              1234, 0, act.data_ptr(), logp.data_ptr(), val.data_ptr(), dbg.data_ptr(), None, None)
    torch.cuda.synchronize()
    NWV = int(os.environ.get("MLP_NW", "8")); d = dbg[:256 * NWV * 4].view(256, NWV, 4).cpu().numpy()
    ph = dbg[256 * NWV * 4:2 * 256 * NWV * 4].view(256, NWV, 4).cpu().numpy() / 16.0   # per pass (16 passes at NW=4, 8 at 8)
    ph = ph * (16.0 / (2 * 4096 / (128 * NWV)))
    print("cycles per pass, by wave index: rows+convert", np.round(np.median(ph[:, :, 0], axis=0)), "\n  layer 1", np.round(np.median(ph[:, :, 1], axis=0)),
          "\n  layer 2", np.round(np.median(ph[:, :, 2], axis=0)), "\n  sampling + loop", np.round(np.median(ph[:, :, 3], axis=0)))
    t0 = d[:, :, 0].min()
    st, en = (d[:, :, 0] - t0) / 100, (d[:, :, 1] - t0) / 100
    print("wave start us: min %.1f p50 %.1f p90 %.1f max %.1f | end us: min %.1f p50 %.1f max %.1f" % (
        st.min(), np.median(st), np.percentile(st, 90), st.max(), en.min(), np.median(en), en.max()))
    wg_start = st.min(axis=1)
    print("workgroups starting later than 10 us:", int((wg_start > 10).sum()), "of 256; their starts:", np.sort(wg_start[wg_start > 10])[:12])
    print("per XCC workgroups:", np.bincount(d[:, 0, 2].astype(int), minlength=8))
    life = en - st
    print("wave life us: min %.1f p50 %.1f max %.1f" % (life.min(), np.median(life), life.max()))
if os.environ.get("MLP_CLOCK"):
    b = np.arange(256)
    net = (b >> 3) & 1
    for n in (0, 1):
        print("net", n, "wave life us p50 %.1f max %.1f" % (np.median(life[net == n]), life[net == n].max()))
    for xc in range(8):
        m = d[:, 0, 2].astype(int) == xc
        print("xcc", xc, "life p50 %.1f max %.1f | net0 p50 %.1f net1 p50 %.1f" % (np.median(life[m]), life[m].max(),
              np.median(life[m & (net == 0)]), np.median(life[m & (net == 1)])))
    print("by wave index p50:", np.round(np.median(life, axis=0), 1))
    wl = life.max(axis=1)
    print("WG max life: sorted tail", np.round(np.sort(wl)[-10:], 1), " head", np.round(np.sort(wl)[:10], 1))
