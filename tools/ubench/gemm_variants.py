"""Which way of issuing the policy GEMMs of a rollout step is fastest (hipBLASLt through torch, bf16)?
[B, 102] x [102, 256|512] (first layers, separate or merged) and [B, 256] x [256, 256] with the A operand
contiguous or a half of a [B, 512] buffer (row stride 512)."""
import torch
B = 262144
dev = "cuda"
x = torch.randn(B, 102, device=dev).bfloat16()
xp = torch.zeros(B, 128, device=dev).bfloat16(); xp[:, :102] = x   # K padded to 128
w1 = torch.randn(256, 102, device=dev).bfloat16(); b1 = torch.randn(256, device=dev).bfloat16()
w1c = torch.randn(512, 102, device=dev).bfloat16(); b1c = torch.randn(512, device=dev).bfloat16()
w1p = torch.zeros(512, 128, device=dev).bfloat16(); w1p[:, :102] = w1c
w2 = torch.randn(256, 256, device=dev).bfloat16(); b2 = torch.randn(256, device=dev).bfloat16()
h512 = torch.relu(torch.randn(B, 512, device=dev)).bfloat16()
hc = h512[:, :256].contiguous()
out256 = torch.empty(B, 256, device=dev, dtype=torch.bfloat16)
out512 = torch.empty(B, 512, device=dev, dtype=torch.bfloat16)


def t(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


AA = torch._addmm_activation
print("L1 separate  [B,102]x[102,256]      %.1f us" % t(lambda: AA(b1, x, w1.t(), use_gelu=False)))
print("L1 merged    [B,102]x[102,512]      %.1f us" % t(lambda: AA(b1c, x, w1c.t(), use_gelu=False)))
print("L1 merged, K padded to 128          %.1f us" % t(lambda: AA(b1c, xp, w1p.t(), use_gelu=False)))
print("L1 merged, out=                      %.1f us" % t(lambda: AA(b1c, x, w1c.t(), use_gelu=False, out=out512)))
print("L2 contiguous A                      %.1f us" % t(lambda: AA(b2, hc, w2.t(), use_gelu=False)))
print("L2 strided A (half of [B,512])       %.1f us" % t(lambda: AA(b2, h512[:, :256], w2.t(), use_gelu=False)))
print("L2 strided A second half             %.1f us" % t(lambda: AA(b2, h512[:, 256:], w2.t(), use_gelu=False)))
print("L2 contiguous, out=                  %.1f us" % t(lambda: AA(b2, hc, w2.t(), use_gelu=False, out=out256)))
print("L2 plain addmm + relu_               %.1f us" % t(lambda: torch.addmm(b2, hc, w2.t()).relu_()))
print("cast f32->bf16 [B,102]               %.1f us" % t(lambda: torch.randn(1, device=dev) if False else x.float().bfloat16()))
xf = x.float()
print("cast only                            %.1f us" % t(lambda: xf.bfloat16()))
# fp32 input first layer (no cast kernel): tf32-less fp32 GEMM
w1f = w1c.float(); b1f = b1c.float()
print("L1 merged fp32 in/out                %.1f us" % t(lambda: AA(b1f, xf, w1f.t(), use_gelu=False)))
