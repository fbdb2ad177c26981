"""Diagnostic: where does a wave spend its life?  Attaches the stamp buffer
(rvo3d_debug_stamps) and prints per-phase mean cycles over all workgroups.
RVO3D_ABLATE=128 also counts the X1 candidates / X2 requests (global atomics inside the sweeps:
the phase times of such a run are inflated - use one run for the times, another for the counts)."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import diaglib  # noqa: F401  the -DRVO3D_DIAG build: the product library has no stamps
from rvo3d_amd import BatchedDroneEnv, synthetic_actions, synthetic_world, _lib
E, N = int(os.environ.get("STAMPS_E", 4096)), int(os.environ.get("STAMPS_N", 64))
NB = int(os.environ.get("STAMPS_NB", 0))
MAP = tuple(float(x) for x in os.environ.get("STAMPS_MAP", "50,50,10").split(","))
env = BatchedDroneEnv(synthetic_world(E, N, MAP, nb=NB), action_decimals=2)
print("launch", env.launch_info(), "E", E, "N", N, "nb", NB, "map", MAP)
acts = [torch.from_numpy(synthetic_actions(E, N, t).astype(np.float32)).cuda() for t in range(12)]
env.observe()
for t in range(10):
    env.step(acts[t], autoreset=True)
nb = env.launch_info()["blocks"]
buf = torch.zeros((nb, 32), dtype=torch.int64, device="cuda")
_lib.check(_lib.lib().rvo3d_debug_stamps(env._h, C.c_void_p(buf.data_ptr())), "stamps")
if os.environ.get("STAMPS_COLD"):  # the stamped launch finds nothing of the previous one in the Infinity Cache
    scratch = torch.zeros(int(os.environ.get("STAMPS_COLD_MB", "1024")) << 20, dtype=torch.uint8, device="cuda")
    if os.environ["STAMPS_COLD"] == "read":
        scratch.sum()   # clean lines only
    else:
        scratch.add_(1)  # dirty lines: the env step's traffic also has to push them out
    print("cold:", os.environ["STAMPS_COLD"])
env.step(acts[10], autoreset=True)
torch.cuda.synchronize()
s = buf.cpu().numpy().astype(np.int64)
names = ["load+dronestate", "stage", "sweepA", "integrate", "lite/sweepB", "reset", "final sweep", "rows + zero fill", "proprio + state stores"]
d = np.diff(s[:, :10], axis=1)
print("phase mean / p50 / p95 cycles (s_memtime ticks):")
for i, n in enumerate(names):
    print(f"  {n:16s} {d[:, i].mean():9.0f} {np.median(d[:, i]):9.0f} {np.percentile(d[:, i], 95):9.0f}")

if s[:, 16:20].any():
    print(f"  integrate: loads + rvo reward {(s[:, 18] - s[:, 3]).mean():8.0f}  barrier + state loads + kinematics + dronestate + mov reward "
          f"{(s[:, 19] - s[:, 18]).mean():8.0f}  buildings + map + stores + restage {(s[:, 4] - s[:, 19]).mean():8.0f}")
    print(f"  rows phase: kept VO rows {(s[:, 16] - s[:, 7]).mean():8.0f}  stage_row + barrier {(s[:, 17] - s[:, 16]).mean():8.0f}  "
          f"row fill {(s[:, 8] - s[:, 17]).mean():8.0f}")
if s[:, 20:24].any():
    T_ = env.launch_info()["threads"]
    print(f"  X2 requests per lane: sweep A mean {s[:, 20].mean() / T_:.4f} (max lane of a workgroup: mean {s[:, 21].mean():.2f}, "
          f"workgroups with none {100 * (s[:, 21] == 0).mean():.1f} %)  rows sweep mean {s[:, 22].mean() / T_:.4f} "
          f"(max lane mean {s[:, 23].mean():.2f}, none {100 * (s[:, 23] == 0).mean():.1f} %)")
if s[:, 24:26].any():
    T_ = env.launch_info()["threads"]
    print(f"  X1 candidates per lane in the rows sweep: mean {s[:, 24].mean() / T_:.3f}; busiest lane of a workgroup: mean {s[:, 25].mean():.2f} "
          f"p95 {np.percentile(s[:, 25], 95):.0f} (trips = half of it)")
if s[:, 26:28].any():
    print(f"  reset phase: decision + barrier + reset loads issued {(s[:, 26] - s[:, 5]).mean():8.0f}  early zero blocks {(s[:, 27] - s[:, 26]).mean():8.0f}  "
          f"reset state, stores, restage {(s[:, 6] - s[:, 27]).mean():8.0f}  re-gate + sweep entry {(s[:, 10] - s[:, 6]).mean():8.0f}")
life = s[:, 9] - s[:, 0]
print("wave life mean", life.mean(), "start spread", s[:, 0].max() - s[:, 0].min(), "end-start", s[:, 9].max() - s[:, 0].min())

if s[:, 10:14].any():  # inner stamps of the final sweep (diagnostic builds only)
    inner = np.diff(s[:, 10:14], axis=1)
    for i, n in enumerate(["G", "X1", "sync+X2+rows"]):
        print(f"  final sweep {n:14s} {inner[:, i].mean():9.0f} {np.median(inner[:, i]):9.0f} {np.percentile(inner[:, i], 95):9.0f}")
    print(f"  sweep A: G+X1 {(s[:, 14] - s[:, 2]).mean():8.0f}  X2 {(s[:, 15] - s[:, 14]).mean():8.0f}  "
          f"dronestate+reward {(s[:, 3] - s[:, 15]).mean():8.0f}")
    print("  final sweep pre-G", (s[:, 10] - s[:, 6]).mean(), "post", (s[:, 7] - s[:, 13]).mean())

# per clock-domain view: s_memtime is not synchronised across XCDs, so cluster the
# workgroups by their start stamp (gaps > 1e6 ticks separate the domains)
order = np.argsort(s[:, 0])
cuts = np.where(np.diff(s[order, 0]) > 1_000_000)[0] + 1
for grp in np.split(order, cuts):
    sx = s[grp]
    t0 = sx[:, 0].min()
    st = np.sort(sx[:, 0] - t0)
    ids = np.sort(grp)[:4]
    print(f"  domain of wgs {ids}.. n={len(grp)}: span {sx[:, 9].max() - t0:7d}  start p50 {st[len(st)//2]:6d} "
          f"p90 {st[int(len(st)*0.9)]:6d} max {st[-1]:6d}  end p50 {int(np.median(sx[:, 9] - t0)):7d}")

# timeline of one CU: the first clock domain with exactly 16 workgroups (4 waves x 4 SIMDs)
for grp in np.split(order, cuts):
    if len(grp) == 16:
        sx = s[np.sort(grp)]
        t0 = sx[:, 0].min()
        print("  one CU, stamps 0..9 relative to its first wave start (k cycles):")
        for r in sx[np.argsort(sx[:, 9])]:
            print("   ", " ".join(f"{(int(v) - int(t0)) / 1000:6.1f}" for v in r[:10]))
        break
