"""Classical RVO velocity selection (reciprocal_vel_obs.py).
CPU: the oracle (a) call by call against what the reference class's OWN methods return where they
run on an instance - preprocess, penalty, distance, the candidate grid of vel_candidate, the
inside branch of vel_select (tests/golden/rvo_calls.npz) - and (b) as a whole against vectors made
with the reference's helper functions (rvo_vel.npz; oracle/gen_golden_rvo.py makes both).
GPU: the HIP kernel against the oracle, with the knife-edge accounting of the step tests.
cal_vel itself, config_vo and vo_out2 raise in the reference: the loop around the pinned parts is
"parity unpinned" (DESIGN.md section 7)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "3drvo-marl-collisionavoidance_amd")]
import oracle as orc  # noqa: E402


def cases():
    z = np.load(os.path.join(HERE, "golden", "rvo_vel.npz"))
    for k in range(int(z["n_cases"])):
        yield {n[len(f"c{k}_"):]: z[n] for n in z.files if n.startswith(f"c{k}_")}


def test_oracle_matches_vectors_made_with_the_reference_helpers():
    n = 0
    for c in cases():
        env = orc.OracleEnv(c["waypoints"], c["n_points"], c["map_size"], None, nm=10)
        env.set_state(pos=c["pos"], vel=c["vel"])
        np.testing.assert_array_equal(env.des_vel(), c["des"])
        got = env.rvo_vel(tuple(c["vmax"]), float(c["acceler"]))
        np.testing.assert_array_equal(got, c["out"])
        n += got.shape[0] * got.shape[1]
    assert n == 168


def calls():
    return np.load(os.path.join(HERE, "golden", "rvo_calls.npz"))


def test_preprocess_matches_the_reference_method():
    """reciprocal_vel_obs.preprocess (:32-54): the 10 m neighbour gate (norm == 10 stays, the very
    same position stays too) and this class's own building gate (h > z - 1, <= 10 m in the plane)."""
    z = calls()
    kept = 0
    for a, d, b, kd, kb in zip(z["pre_agent"], z["pre_drones"], z["pre_blds"], z["pre_keep_d"], z["pre_keep_b"]):
        gd, gb = orc.rvo_preprocess(a, d, b)
        assert np.array_equal(gd, kd.astype(bool)) and np.array_equal(gb, kb.astype(bool))
        assert gd[3] and not gd[2]                          # the very same spot stays; 10 m + 5 um does not
        kept += int(gd.sum())
    # drones 0 / 1 sit 6-8-0 / 0-6-8 away from 2-decimal coordinates: |d| = 10 up to the rounding of the
    # sums, so the <= 10 gate falls either way - whatever the reference's own norm says is what counts
    assert 0 < int(z["pre_keep_d"][:, :2].sum()) < 120
    assert 100 < kept < 60 * 12 and 0 < int(z["pre_keep_b"].sum()) < 60 * 8


def test_penalty_distance_and_inside_selection_match_the_reference_methods():
    """reciprocal_vel_obs.penalty (:126-147: tc_min over the neighbours, inf for tc_min == 0, 0 for
    no collision course), .distance, and vel_select's inside branch (min by penalty, first of equals)."""
    z = calls()
    for k in range(len(z["pen_n"])):
        n, n_in = int(z["pen_n"][k]), int(z["pen_n_in"][k])
        ag, od = z["pen_agent"][k], z["pen_odro"][k][:n]
        got = orc.rvo_penalty(z["pen_vel"][k], z["pen_des"][k], ag[:8], od, 1.0)
        assert got == z["pen_pen"][k] or (np.isnan(got) and np.isnan(z["pen_pen"][k])), k
        assert orc.rvo_distance(z["pen_vel"][k], z["pen_des"][k]) == z["pen_dist"][k]
        assert orc.rvo_select_inside(z["pen_inside"][k][:n_in], ag, od) == int(z["pen_sel"][k]), k
    assert np.isinf(z["pen_pen"]).sum() > 5 and (z["pen_pen"] == z["pen_dist"]).sum() > 5  # both ends of 1/tc


def test_candidate_grid_matches_vel_candidate():
    """reciprocal_vel_obs.vel_candidate (:85-101) with an empty VO list returns the candidate grid
    itself: np.arange over the clipped range per axis, |v| >= 0.3, x outermost."""
    z = calls()
    sizes = set()
    for v, vm, acc, n, cand in zip(z["cand_vel"], z["cand_vmax"], z["cand_acc"], z["cand_n"], z["cand_cand"]):
        got = orc.rvo_candidates(v, vm, float(acc))
        assert len(got) == int(n)
        np.testing.assert_array_equal(got, cand[:int(n)])
        sizes.add(int(n))
    assert 0 in sizes and max(sizes) >= 27 and len(sizes) > 5


def test_what_does_not_run_in_the_reference_is_on_record():
    z = calls()
    assert str(z["raises_config_vo"]) == "ValueError"                    # :63-83, state[0:4] into get_PAA
    assert str(z["raises_vel_candidate_with_vo"]) == "AttributeError"    # :109, list.append assignment
    assert str(z["raises_cal_vel"]) != ""                                # the driver itself
    assert bool(z["vel_select_outside_returns_none"])                    # :119-124, missing return


@pytest.mark.gpu
def test_hip_rvo_vel_matches_oracle():
    import torch
    from rvo3d_amd import BatchedDroneEnv, synthetic_world
    from test_gpu_parity import Tally
    rng = np.random.default_rng(7)
    agents = 0
    for (E, N, L, acc, vmax) in [(64, 16, 8.0, 0.5, (2.0, 2.0, 2.0)), (16, 64, 12.0, 1.0, (2.0, 2.0, 2.0)),
                                 (8, 100, 14.0, 0.5, (1.0, 1.5, 1.0)), (32, 5, 4.0, 0.75, (2.0, 2.0, 1.0)),
                                 (4, 256, 30.0, 0.5, (2.0, 2.0, 2.0))]:
        w = synthetic_world(E, N, (L, L, L), seed=int(rng.integers(1 << 30)), min_sep=0.5)
        env = BatchedDroneEnv(w, neighbors_num=10)
        ref = orc.OracleEnv(w.waypoints, w.n_points, w.map_size, w.buildings, nm=10, threads=8)
        pos = np.round(w.waypoints[:, :, 0] + rng.uniform(-0.3, 0.3, (E, N, 3)), 2)
        vel = np.round(rng.uniform(-1.2, 1.2, (E, N, 3)), 2)
        vel[rng.random((E, N)) < 0.15] = 0.0
        env.set_state(pos=pos, vel=vel); ref.set_state(pos=pos, vel=vel)
        got = env.rvo_vel(vmax, acc).cpu().numpy()
        want = ref.rvo_vel(vmax, acc)
        # alpha = round(asin, 2) and beta = round(acos, 2) rest on libm results (glibc vs ocml: <= 1 ulp
        # apart): the oracle reports, per agent, how close any of its roundings came to a tie; only
        # agents within 1e-9 of one may differ, and they are counted (class Tally, as for the step)
        tl = Tally(f"rvo_vel/{N}x{E}", E)
        tl.begin(ref.margin())
        tl.check("rvo_vel", got == want)
        tl.end()
        tl.finish(agents=E * N)
        agents += E * N
        env.close()
    assert agents > 3000
