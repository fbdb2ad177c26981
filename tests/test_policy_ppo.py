"""Policy / PPO surface against vectors produced by the
reference's own classes on CPU (oracle/gen_golden_policy.py): rnn_ac forward,
multi_PPObuf GAE, compute_loss_pi / compute_loss_v incl. gradients.  CPU only."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from golden_util import GOLDEN, load
from rvo3d_amd.policy import gae_scan, mlp_ac, multi_ppo, rnn_ac
from rvo3d_amd.policy.multi_ppo import RolloutBuffer

TOL = dict(rtol=2e-5, atol=2e-6)


class _Space:
    shape = (3,)


def small_ac(fx):
    ac = rnn_ac(None, _Space(), 12, 9, 32, (32, 32), (32, 32), torch.nn.ReLU, torch.nn.Tanh,
                torch.nn.Identity, use_gpu=False, rnn_mode="biGRU")
    sd = {k[2:]: torch.as_tensor(v) for k, v in fx.items() if k.startswith("w:")}
    ac.load_state_dict(sd, strict=True)  # same key names as the reference module
    return ac.eval()


def ragged(fx):
    out = []
    for o, c in zip(fx["obs"], fx["count"]):
        out.append(torch.as_tensor(o[:12 + 9 * max(int(c), 1)]))
    return out


def test_checkpoint_key_names_and_shapes():
    """A reference checkpoint (`model_state`, biGRU 256 / MLP 256-256) loads strict."""
    ac = rnn_ac(None, _Space(), 12, 9, 256, (256, 256), (256, 256), torch.nn.ReLU, torch.nn.Tanh,
                torch.nn.Identity, use_gpu=False, rnn_mode="biGRU")
    want = {}
    for line in open(os.path.join(GOLDEN, "policy_rnn_ac_keys.txt")):
        k, shp = line.split(" ", 1)
        want[k] = eval(shp)
    have = {k: tuple(v.shape) for k, v in ac.state_dict().items()}
    assert have == want
    assert sum(p.numel() for p in ac.parameters()) == 680991


def test_rnn_ac_forward_matches_reference():
    fx = load(os.path.join(GOLDEN, "policy_rnn_ac.npz"))
    ac = small_ac(fx)
    obs = torch.as_tensor(fx["obs"])
    cnt = torch.as_tensor(fx["count"])
    act = torch.as_tensor(fx["act"])
    with torch.no_grad():
        pi, logp = ac.pi((obs, cnt), act)            # batched padded path (masked biGRU)
        v = ac.v((obs, cnt))
        pil, logpl = ac.pi(ragged(fx), act)          # the reference's list-of-ragged path
        pis, _ = ac.pi((obs, cnt), act, std_factor=1e-3)
    for got, key in ((pi.mean, "mu"), (pi.stddev.expand_as(pi.mean), "std"), (logp, "logp"),
                     (v, "v"), (pil.mean, "mu"), (logpl, "logp"), (pi.entropy(), "entropy"),
                     (pis.stddev.expand_as(pi.mean), "std_small")):
        np.testing.assert_allclose(got.numpy(), fx[key], **TOL, err_msg=key)
    with torch.no_grad():  # single ragged observation, as ac.step(obs) in the trainer
        for i in range(8):
            o = ragged(fx)[i]
            np.testing.assert_allclose(ac.pi._distribution(o).mean.numpy(), fx["single_mu"][i], **TOL)
            np.testing.assert_allclose(float(ac.v(o)), fx["single_v"][i], **TOL)
    a, vv, lp = ac.step(ragged(fx)[2])
    assert a.shape == (3,) and vv.shape == () and lp.shape == ()


def test_gae_matches_multi_PPObuf():
    fx = load(os.path.join(GOLDEN, "ppo_gae.npz"))
    rew, val, cut = (torch.as_tensor(fx[k]) for k in ("rew", "val", "cuts"))
    adv, ret = gae_scan(rew, val, cut.bool(), float(fx["gamma"]), float(fx["lam"]))
    np.testing.assert_allclose(adv.numpy(), fx["adv"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(ret.numpy(), fx["ret"], rtol=1e-5, atol=1e-6)
    # batched [T, E, N] layout with per-env cuts == the same scan per column
    T = rew.shape[0]
    R = rew[:, None, None].expand(T, 3, 2).contiguous()
    V = val[:, None, None].expand(T, 3, 2).contiguous()
    C = cut.bool()[:, None, None].expand(T, 3, 2)
    A2, R2 = gae_scan(R, V, C, 0.99, 0.97)
    np.testing.assert_array_equal(A2[:, 1, 1].numpy(), adv.numpy())
    np.testing.assert_array_equal(R2[:, 2, 0].numpy(), ret.numpy())


def test_vectorised_gae_equals_the_step_by_step_scan():
    """gae_scan (two reverse prefix sums) against gae_scan_loop (one step at a time) on random
    [T, E, N] data with random per-env cuts, long buffers included."""
    from rvo3d_amd.policy import gae_scan_loop
    g = torch.Generator().manual_seed(3)
    for T, E, N in ((1, 2, 3), (7, 5, 2), (300, 6, 4), (2000, 2, 2)):
        rew = torch.randn(T, E, N, generator=g) * 3
        val = torch.randn(T, E, N, generator=g) * 2
        cut = (torch.rand(T, E, generator=g) < 0.05)[:, :, None].expand(T, E, N)
        a1, r1 = gae_scan(rew, val, cut, 0.99, 0.97)
        a2, r2 = gae_scan_loop(rew, val, cut, 0.99, 0.97)
        np.testing.assert_allclose(a1.numpy(), a2.numpy(), rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(r1.numpy(), r2.numpy(), rtol=1e-6, atol=1e-6)


def test_rollout_buffer_cut_rules():
    buf = RolloutBuffer(4, 2, 3, 21, 3, "cpu", 0.99, 0.97)
    for t in range(4):
        buf.store(torch.zeros(2, 3, 21), torch.zeros(2, 3, dtype=torch.int32), torch.zeros(2, 3, 3),
                  torch.ones(2, 3), torch.zeros(2, 3), torch.zeros(2, 3))
        buf.finish_path(torch.tensor([t == 1, False]) | torch.tensor([t == 3, t == 3]))
    with pytest.raises(AssertionError):
        buf.store(*[None] * 6)
    d = buf.get()
    ret = d["ret"].view(4, 2, 3)
    np.testing.assert_allclose(ret[:, 0, 0].numpy(), [1.99, 1.0, 1.99, 1.0], rtol=1e-6)  # cut after t=1
    np.testing.assert_allclose(ret[:, 1, 2].numpy(), [1 + .99 * (1 + .99 * 1.99), 1 + .99 * 1.99, 1.99, 1.0], rtol=1e-6)
    assert d["obs"].shape == (24, 21) and buf.ptr == 0


class _FakeEnv:
    E, N, W, device = 1, 1, 102, torch.device("cpu")


def test_losses_and_gradients_match_reference():
    fx = load(os.path.join(GOLDEN, "policy_rnn_ac.npz"))
    lx = load(os.path.join(GOLDEN, "ppo_loss.npz"))
    ac = small_ac(fx)
    tr = multi_ppo(_FakeEnv(), ac, steps_per_epoch=2, use_gpu=False)
    data = dict(obs=torch.as_tensor(fx["obs"]), cnt=torch.as_tensor(fx["count"]),
                act=torch.as_tensor(fx["act"]), adv=torch.as_tensor(lx["adv"]),
                ret=torch.as_tensor(lx["ret"]), logp=torch.as_tensor(lx["logp_old"]))
    ac.zero_grad()
    loss_pi, info = tr.compute_loss_pi(data)
    loss_pi.backward()
    assert abs(float(loss_pi) - float(lx["loss_pi"])) < 1e-5
    for k in ("kl", "ent", "cf"):
        assert abs(info[k] - float(lx[k])) < 1e-5, k
    g = dict(ac.named_parameters())
    np.testing.assert_allclose(g["pi.log_std"].grad.numpy(), lx["g_pi_log_std"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(g["pi.net_out.4.weight"].grad.numpy(), lx["g_pi_out"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(g["pi.rnn_reader.rnn_net.weight_ih_l0"].grad.numpy(), lx["g_pi_gru"],
                               rtol=1e-3, atol=1e-6)
    ac.zero_grad()
    loss_v = tr.compute_loss_v(data)
    loss_v.backward()
    assert abs(float(loss_v) - float(lx["loss_v"])) < 1e-4 * max(1.0, abs(float(lx["loss_v"])))
    np.testing.assert_allclose(g["v.v_net.4.weight"].grad.numpy(), lx["g_v_out"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(g["pi.rnn_reader.rnn_net.weight_hh_l0_reverse"].grad.numpy(), lx["g_v_gru"],
                               rtol=1e-3, atol=1e-6)


def _update_case(fx, case, device="cpu", as_list=False):
    """Run our multi_ppo.update on the fixture's per-agent buffers; returns (trainer, stats)."""
    ac = rnn_ac(None, _Space(), 12, 9, 16, (24, 24), (24, 24), torch.nn.ReLU, torch.nn.Tanh,
                torch.nn.Identity, use_gpu=False, rnn_mode="biGRU")
    ac.load_state_dict({k[3:]: torch.as_tensor(v) for k, v in fx.items() if k.startswith("w0:")},
                       strict=True)
    ac = ac.to(device)
    N, T = fx["adv"].shape

    class Env:
        E, W = 1, 102
    Env.N, Env.device = N, torch.device(device)
    tr = multi_ppo(Env(), ac, pi_lr=float(fx["pi_lr"]), vf_lr=float(fx["vf_lr"]), steps_per_epoch=T,
                   train_pi_iters=int(fx["train_pi_iters"]), train_v_iters=int(fx["train_v_iters"]),
                   target_kl=float(fx["target_kl"]), clip_ratio=float(fx["clip_ratio"]),
                   max_update_num=int(fx["max_update_num" + case]), reference_order=True,
                   seed=int(fx["np_seed"]), use_gpu=False)
    dev = lambda a: torch.as_tensor(a).to(device)
    if as_list:  # the reference's own calling convention: one dict per agent, ragged obs lists
        data = []
        for n in range(N):
            obs = [dev(fx["obs"][n, t, :12 + 9 * max(int(fx["count"][n, t]), 1)]) for t in range(T)]
            data.append(dict(obs=obs, act=dev(fx["act"][n]), ret=dev(fx["ret"][n]), adv=dev(fx["adv"][n]),
                             logp=dev(fx["logp"][n])))
    else:        # the flattened [T, E, N] rollout of RolloutBuffer.get()
        tn = lambda a: dev(np.ascontiguousarray(np.swapaxes(a, 0, 1)).reshape((T * N,) + a.shape[2:]))
        data = dict(obs=tn(fx["obs"]), cnt=tn(fx["count"]), act=tn(fx["act"]), ret=tn(fx["ret"]),
                    adv=tn(fx["adv"]), logp=tn(fx["logp"]), shape=(T, 1, N))
    return tr, tr.update(data)


@pytest.mark.parametrize("case,as_list", [("1", False), ("2", False), ("1", True)])
def test_update_reproduces_the_references_update(case, as_list):
    """8(f) row 2: multi_ppo.update of the reference itself (multi_ppo.py:341-376, fixture from
    oracle/gen_golden_ppo_update.py): shuffled agent order, max_update_num (case 2: only two
    agents are visited), the KL stop before the step (agent 0 stops after 3 policy steps), Adam
    on pi and v with the shared reader: the parameters after the update match."""
    fx = load(os.path.join(GOLDEN, "ppo_update.npz"))
    tr, st = _update_case(fx, case, as_list=as_list)
    assert st["order"] == fx["order" + case].tolist()
    assert st["pi_steps"] == fx["pi_steps" + case].tolist()
    assert (fx["pi_steps1"] == [6, 6, 3, 6]).all() and len(fx["pi_steps2"]) == 2
    got = tr.ac.state_dict()
    moved = 0.0
    for k, v in got.items():
        want = fx[f"w{case}:" + k]
        np.testing.assert_allclose(v.numpy(), want, rtol=2e-4, atol=2e-6, err_msg=k)
        moved = max(moved, float(np.abs(want - fx["w0:" + k]).max()))
    assert moved > 1e-3  # the update really moved the weights


def test_pooled_update_warns_about_max_update_num():
    with pytest.warns(UserWarning, match="max_update_num"):
        multi_ppo(_FakeEnv(), mlp_ac(21, hidden_sizes=(8, 8)), steps_per_epoch=2, max_update_num=3)


def test_kl_stop_is_decided_by_the_mean_over_ranks():
    """The KL estimate travels in the last slot of the gradient bucket: both ranks see the same mean,
    stop before the same optimizer step (multi_ppo.py:362) and stay identical."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_ddp_worker, args=(r, world, port, q, 1e-4, 200)) for r in range(world)]
    for p in ps:
        p.start()
    out = dict(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    (w0, steps0, kl0), (w1, steps1, kl1) = out[0], out[1]
    assert steps0 == steps1 and 1 <= steps0 < 200, (steps0, steps1)
    assert kl0 == kl1 and kl0 > 1e-4
    np.testing.assert_array_equal(w0, w1)


def test_mlp_ac_surface():
    ac = mlp_ac(102)
    obs = torch.randn(5, 102)
    a, v, lp = ac.step_tensors((obs, torch.zeros(5, dtype=torch.int32)))
    assert a.shape == (5, 3) and v.shape == (5,) and lp.shape == (5,)
    d, logp = ac.pi(obs, a)
    assert torch.allclose(logp, lp, atol=1e-6)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _ddp_worker(rank, world, port, q, target_kl=1e9, iters=3):
    import torch.distributed as dist
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    dist.init_process_group("gloo")
    torch.manual_seed(0)
    ac = mlp_ac(21, hidden_sizes=(16, 16))
    tr = multi_ppo(_FakeEnv(), ac, steps_per_epoch=2, train_pi_iters=iters, train_v_iters=2,
                   target_kl=target_kl, use_gpu=False, dist=dist, seed=0)
    g = torch.Generator().manual_seed(100 + rank)  # every rank has its own shard of samples
    n = 64
    data = dict(obs=torch.randn(n, 21, generator=g), act=torch.randn(n, 3, generator=g) * 0.3,
                adv=torch.randn(n, generator=g), ret=torch.randn(n, generator=g),
                logp=torch.randn(n, generator=g) * 0.1 - 2.0)
    if target_kl < 1e9:  # log-probs of the policy that "collected" the samples: KL starts at 0 and grows
        with torch.no_grad():
            data["logp"] = ac.pi(data["obs"], data["act"])[1]
    st = tr.update(data)
    out = torch.cat([p.detach().reshape(-1) for p in ac.parameters()]).numpy()
    q.put((rank, out) if target_kl >= 1e9 else (rank, (out, st["pi_steps"], st["kl"])))
    dist.barrier()
    dist.destroy_process_group()


def _unequal_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    dist.init_process_group("gloo")

    class Env:
        N, W, device = 2, 21, torch.device("cpu")
    Env.E = 3 + rank  # unequal shards
    try:
        multi_ppo(Env(), mlp_ac(21, hidden_sizes=(8, 8)), steps_per_epoch=2, dist=dist)
        q.put((rank, "no error"))
    except ValueError as ex:
        q.put((rank, str(ex)))
    dist.barrier()
    dist.destroy_process_group()


def test_unequal_shards_are_rejected():
    """Ranks with different shard sizes would run different numbers of optimizer steps and hang
    in the gradient all-reduce: the constructor refuses them on every rank."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_unequal_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    out = dict(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all("different shard sizes" in v for v in out.values()), out


def test_gradient_allreduce_keeps_ranks_identical():
    """8(e): one all-reduce of the gradient bucket per optimizer step -> replicas stay equal."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_ddp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    out = dict(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    np.testing.assert_array_equal(out[0], out[1])


def test_cast_cache_is_tied_to_the_parameter_object():
    """the bf16 weight cache was keyed on id(p) + version; CPython reuses the id of a freed
    Parameter, so a second model built after the first one was freed got the first one's weights."""
    from rvo3d_amd.policy import policy_rnn_ac as P
    stale = 0
    for i in range(40):  # sequentially created and freed: ids are reused
        p = torch.nn.Parameter(torch.full((4, 4), float(i)))
        c = P._cast_cached(p, torch.bfloat16)
        stale += int(not torch.equal(c.float(), p.detach()))
        del p, c
    assert stale == 0
    assert len(P._CAST_CACHE) <= 1  # entries die with their parameter
    # same object: cached until its version (in-place update) or its storage changes
    p = torch.nn.Parameter(torch.ones(3))
    c0 = P._cast_cached(p, torch.bfloat16)
    assert P._cast_cached(p, torch.bfloat16) is c0
    with torch.no_grad():
        p.add_(1.0)
    assert torch.equal(P._cast_cached(p, torch.bfloat16).float(), torch.full((3,), 2.0))
    p.data = torch.full((3,), 5.0)  # what module.to(...) / load with assign do: new storage, same object
    assert torch.equal(P._cast_cached(p, torch.bfloat16).float(), torch.full((3,), 5.0))


def test_two_models_in_a_row_under_the_cached_inference_path(monkeypatch):
    """The cached bf16 inference path (mlp_ac.step_tensors under autocast) against the uncached module
    forward, for two models built one after the other in one process."""
    monkeypatch.setattr(torch, "is_autocast_enabled", lambda *a: True)
    for seed in (1, 2, 3):
        torch.manual_seed(seed)
        ac = mlp_ac(102)
        x = torch.randn(32, 102)
        mu = ac._fused_forward(ac.pi_net, x.to(torch.bfloat16)).float()
        ref = ac.pi_net.to(torch.bfloat16)(x.to(torch.bfloat16)).float()
        assert torch.allclose(mu, ref, atol=2e-2), seed
        del ac


def _shard_data(rank, n=64):
    g = torch.Generator().manual_seed(100 + rank)
    return dict(obs=torch.randn(n, 21, generator=g), act=torch.randn(n, 3, generator=g) * 0.3,
                adv=torch.randn(n, generator=g), ret=torch.randn(n, generator=g),
                logp=torch.randn(n, generator=g) * 0.1 - 2.0)


def _order_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    dist.init_process_group("gloo")
    torch.manual_seed(0)
    ac = mlp_ac(21, hidden_sizes=(16, 16))
    tr = multi_ppo(_FakeEnv(), ac, steps_per_epoch=2, train_pi_iters=2, train_v_iters=1, target_kl=1e9,
                   use_gpu=False, dist=dist, seed=5, reference_order=True, max_update_num=2)
    np.random.seed(1000 + rank)            # somebody else uses numpy's global generator, differently
    np.random.rand(rank + 3)               # on every rank: the agent order must not care
    noise = float(torch.randn(1))          # action noise differs between the shards (seed + rank)
    agents = [_shard_data(10 * rank + k, n=16) for k in range(4)]
    st = tr.update(agents)                 # the reference's own signature: one dict per agent
    w = torch.cat([p.detach().reshape(-1) for p in ac.parameters()]).numpy()
    q.put((rank, (st["order"], noise, w)))
    dist.barrier()
    dist.destroy_process_group()


def test_reference_order_is_the_same_on_every_rank_and_noise_is_not():
    """the shuffled agent order must not depend on the process-global numpy generator
    (ranks that drew different orders would average gradients of different agents, silently), and the
    sampling noise must differ between the shards."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_order_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    out = dict(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    (o0, n0, w0), (o1, n1, w1) = out[0], out[1]
    assert o0 == o1 and sorted(o0) == [0, 1, 2, 3]
    ref = np.arange(4)
    np.random.RandomState(5).shuffle(ref)  # = np.random.seed(5); np.random.shuffle (the reference)
    assert o0 == [int(x) for x in ref]
    assert n0 != n1
    np.testing.assert_array_equal(w0, w1)


def test_two_rank_update_equals_one_rank_on_the_concatenated_data():
    """Mean of the two shards' gradients = gradient of the mean over both shards (equal shards): the
    pooled update on 2 ranks ends where a single process ends on the concatenated samples."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_ddp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    out = dict(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(0)
    ac = mlp_ac(21, hidden_sizes=(16, 16))
    tr = multi_ppo(_FakeEnv(), ac, steps_per_epoch=2, train_pi_iters=3, train_v_iters=2, target_kl=1e9,
                   use_gpu=False, seed=0)
    both = [_shard_data(0), _shard_data(1)]
    tr.update({k: torch.cat([d[k] for d in both]) for k in both[0]})
    one = torch.cat([p.detach().reshape(-1) for p in ac.parameters()]).numpy()
    np.testing.assert_allclose(out[0], one, rtol=2e-5, atol=2e-6)


def test_fused_plan_reproduces_the_module_forward():
    """The rollout's inference plan (merged first layers, per-net hidden layers, float32 head rows for
    rvo3d_policy_sample) against the modules' own forward, float32, on the CPU."""
    from rvo3d_amd.policy.policy_rnn_ac import _hidden_pair
    torch.manual_seed(3)
    ac = mlp_ac(102)
    plan = ac.fused_plan(torch.float32)
    assert plan is not None and plan["hidden"] == 256 and plan["tanh"] and plan["k_pad"] == 102
    x = torch.randn(50, 102)
    hp, hv = _hidden_pair(x, plan)
    mu = torch.tanh(hp @ plan["w_pi"].t() + plan["b_pi"])
    v = hv @ plan["w_v"] + plan["b_v"]
    with torch.no_grad():
        assert torch.allclose(mu, ac.dist(x).mean, atol=1e-6) and torch.allclose(v, ac.v(x), atol=1e-6)
    assert ac.fused_plan(torch.float32) is plan                      # cached ...
    with torch.no_grad():
        ac.log_std.add_(0.1)                                         # ... until a parameter changes
    assert ac.fused_plan(torch.float32) is not plan
    assert ac.fused_plan(torch.bfloat16)["k_pad"] == 128             # reduced precision: K padded to a multiple of 64
    assert mlp_ac(21, hidden_sizes=(8, 8)).fused_plan(torch.float32)["hidden"] == 8  # (the trainer checks the width)
    odd = mlp_ac(21)
    odd.v_net = torch.nn.Sequential(torch.nn.Linear(21, 256), torch.nn.Tanh(), torch.nn.Linear(256, 1), torch.nn.Identity())
    assert odd.fused_plan(torch.float32) is None                     # not a ReLU stack of the actor's depth


def test_reader_one_step_fast_path_equals_the_masked_recurrence():
    """rnn_Reader.forward_batch evaluates one GRU cell step from h = 0 for every row and the unrolled masked
    recurrence only for rows with several VO rows: it must be the same function as the recurrence on all rows."""
    from rvo3d_amd.policy.policy_rnn_ac import rnn_Reader
    torch.manual_seed(5)
    for mode in ("biGRU", "GRU"):
        r = rnn_Reader(12, 9, 24, use_gpu=False, mode=mode)
        B, S = 300, 10
        obs = torch.randn(B, 12 + 9 * S)
        lens = torch.randint(1, S + 1, (B,))
        lens[: B // 2] = 1
        x = obs[:, 12:].reshape(B, S, 9)
        with torch.no_grad():
            h = r._gru_dir(x, lens, "", False)
            if mode == "biGRU":
                h = h + r._gru_dir(x, lens, "_reverse", True)
            want = r.ln(torch.cat((obs[:, :12], h), 1))
            got = r.forward_batch(obs, lens)
        assert torch.equal(got[: B // 2], want[: B // 2])            # one-step rows: bit for bit
        assert torch.allclose(got, want, atol=1e-6)
        assert torch.equal(r.forward_batch(obs[: B // 2], lens[: B // 2]), want[: B // 2])  # a batch without long rows


@pytest.mark.parametrize("mode", ["biGRU", "GRU"])
def test_collapsed_first_layer_is_the_reader_plus_layer_norm_plus_linear(mode):
    """policy_rnn_ac.collapsed_first_layer (host side of the rollout's "rnn0" mode): for rows without a velocity-obstacle
    row the GRU state is a constant and the first MLP layer is W_p f_p + rstd a - (mean rstd) b + c.  In float64 on the
    CPU against the module's own reader + LayerNorm + Linear: equal to 1e-5 of the pre-activations (the module is float32)."""
    from rvo3d_amd.policy.policy_rnn_ac import collapsed_first_layer
    torch.manual_seed(4)
    ac = rnn_ac(None, _Space(), 12, 9, 32, (64, 64), (64, 64), torch.nn.ReLU, torch.nn.Tanh, torch.nn.Identity,
                use_gpu=False, rnn_mode=mode)
    with torch.no_grad():
        for p_ in ac.pi.rnn_reader.parameters():
            p_.add_(torch.randn_like(p_) * 0.3)
    r = ac.pi.rnn_reader
    rows, W = 500, 12 + 9 * 4
    obs = torch.zeros((rows, W))
    obs[:, :12] = torch.randn((rows, 12)) * torch.tensor([30., 30, 5, 2, 2, 2, 1, 1, 1, 1, 1, 1])
    for net in (ac.pi.net_out, ac.v.v_net):
        lin = [m for m in net if isinstance(m, torch.nn.Linear)][0]
        Wp, a, b, c = collapsed_first_layer(r, lin)
        with torch.no_grad():
            feat = r.forward_batch(obs, torch.ones(rows, dtype=torch.int64))
            want = lin(feat).double()
            h0 = r._gru_first(torch.zeros((1, 9)), "")
            if mode == "biGRU":
                h0 = h0 + r._gru_first(torch.zeros((1, 9)), "_reverse")
            x = torch.cat([obs[:, :12].double(), h0.double().expand(rows, -1)], 1)
            mean = x.mean(1, keepdim=True)
            rstd = 1.0 / torch.sqrt(x.var(1, unbiased=False, keepdim=True) + r.ln.eps)
            f_p = (obs[:, :12].double() - mean) * rstd * r.ln.weight.double()[:12] + r.ln.bias.double()[:12]
            got = f_p @ Wp.T + rstd * a[None, :] - (mean * rstd) * b[None, :] + c[None, :]
        assert float((got - want).abs().max()) < 1e-5 * max(1.0, float(want.abs().max()))
    assert ac.zero_vo_plan() is None      # (the packed plan itself is a GPU matter: None on the CPU, the rollout takes another path)
