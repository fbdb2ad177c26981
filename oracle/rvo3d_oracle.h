/*
 * rvo3d_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Plain-C, scalar fp64 restatement of the reference's hot path
 * (uaisa_env/drone_envs/mdin.py:19-30 `mdin.drone_step` and the reset /
 * observation paths around it).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product path
 * (librvo3d_hip.so) never links or calls it.
 *
 * Parity status: PINNED.  oracle/gen_golden.py runs the Python reference in
 * the build container and commits per-step vectors under tests/golden/;
 * tests/test_oracle_golden.py replays them through this library.
 *
 * Layout: E independent environments x N drones, drone (e, d) lives at flat
 * index e*N + d.  World inputs are given AoS (as numpy builds them); state is
 * kept SoA inside the handle.
 */
#ifndef RVO3D_ORACLE_H
#define RVO3D_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_env orc_env;

/* E envs, N drones/env, P = max waypoints per route, nb buildings (shared by
 * all envs), nm = max VO rows in an observation (reference default 10,
 * mdin.py:7), env_train as rvo_inter.py:14. map_size = [x, y, z]. */
orc_env *orc_create(int E, int N, int P, int nb, int nm, int env_train,
                    const double *map_size);
void orc_destroy(orc_env *h);

/* waypoints [E][N][P][3]; n_points [E][N]; buildings [nb][4] = x,y,h,r
 * (env_base.py:35-39).  radius / priority [E][N] or NULL for the reference
 * constants 0.2 / 5 (drone.py:14-15).  Puts every drone in its reset state. */
void orc_load_world(orc_env *h, const double *waypoints, const int32_t *n_points,
                    const double *buildings, const double *radius,
                    const double *priority);

/* drone.reset (drone.py:270-291) for every drone of the masked envs
 * (env_mask NULL = all) / for the masked drones. */
void orc_reset(orc_env *h, const uint8_t *env_mask);
void orc_reset_drones(orc_env *h, const uint8_t *drone_mask);

/* ir_gym.env_observation / env_reset observation (ir_gym.py:334-383):
 * obs [E][N][12+9*nm] (zero padded), vo_count [E][N] (0 => one zero row). */
void orc_observe(orc_env *h, double *obs, int32_t *vo_count);

/* mdin.drone_step (mdin.py:19-30). actions [E][N][3]. */
void orc_step(orc_env *h, const double *actions, double *obs, int32_t *vo_count,
              double *reward, uint8_t *done, uint8_t *info, uint8_t *finish);

/* orc_step followed by the caller-side protocol of multi_ppo.py:230-242:
 * drones with done|finish are reset, and every env that reset at least one
 * drone has ALL its observations recomputed with action = 0.
 * reset_mask [E][N] (may be NULL) reports which drones were reset. */
void orc_step_autoreset(orc_env *h, const double *actions, double *obs,
                        int32_t *vo_count, double *reward, uint8_t *done,
                        uint8_t *info, uint8_t *finish, uint8_t *reset_mask);

/* State read-back; any pointer may be NULL. pos/vel [E][N][3]; rest [E][N]. */
void orc_get_state(const orc_env *h, double *pos, double *vel, double *yaw,
                   double *pitch, double *real_len, double *max_dev,
                   double *extra_len, int32_t *wp_idx, uint8_t *arrive,
                   uint8_t *dest);
/* State overwrite (tests: seed the oracle and the device with one state). */
void orc_set_state(orc_env *h, const double *pos, const double *vel,
                   const double *yaw, const double *pitch, const double *real_len,
                   const double *max_dev, const double *extra_len,
                   const int32_t *wp_idx, const uint8_t *arrive,
                   const uint8_t *dest);

/* ir_gym.cal_des_list (ir_gym.py:44): desired velocity [E][N][3]. */
void orc_des_vel(const orc_env *h, double *des_vel);
/* Classical RVO velocity selection of every drone on the current state
 * (reciprocal_vel_obs.py:19-166 as intended; the class itself cannot run: PARITY UNPINNED
 * for the driver loop, see the .c).  vmax[3], acceler; out [E*N][3]. */
void orc_rvo_vel(const orc_env *h, const double *vmax, double acceler, double *out);
/* The methods of reciprocal_vel_obs that run on an instance (reciprocal_vel_obs.py:32-54, :85-101
 * with an empty VO list, :119-124 with an empty vo_outside, :126-147, :149-151), call by call:
 * checked against tests/golden/rvo_calls.npz, which holds the reference's own return values. */
double orc_rvo_distance(const double *p1, const double *p2);
void orc_rvo_preprocess(const double *agent3, const double *drones, int n, const double *blds, int nb,
                        uint8_t *keep_drone, uint8_t *keep_bld);
double orc_rvo_penalty(const double *vel, const double *vel_des, const double *agent8,
                       const double *odro8, int n, double factor);
int orc_rvo_candidates(const double *vel3, const double *vmax, double acceler, double *out, int cap);
int orc_rvo_select_inside(const double *inside, int n_in, const double *agent11, const double *odro8, int n);

/* Call-level check of rvo_inter.config_vo_inf (rvo_inter.py:20-61) for drone i
 * of env e against the current state, with an arbitrary action.  rows
 * [nm][9] in the reference's order; tmin over ALL flagged pairs. */
void orc_vo_inf(orc_env *h, int e, int i, const double *action, double *rows,
                int32_t *count, int32_t *vo_flag, double *tmin, int32_t *collision);

/* Decision-margin audit: for every drone, the smallest |value - threshold|
 * over all branches and roundings evaluated for it by the LAST observe / step
 * call (absolute units; +inf if none).  Where this is below ~1e-9 the
 * reference's own outcome is decided by libm rounding noise, and a device
 * result may legitimately differ.  out [E][N]. */
void orc_get_margin(const orc_env *h, double *out, int32_t *site /* source line, may be NULL */);

/* Number of observations that contained NaN/Inf since creation (the
 * reference raises ValueError, ir_gym.py:232-239). */
int64_t orc_nan_count(const orc_env *h);

/* Number of pairs, since creation, on which the reference raises ValueError("math domain
 * error") out of get_alpha (vel_obs3D.py:13; env_train = False, r - 0.2 + mr < dis < r + mr,
 * approaching): the reference aborts the step there, the oracle treats the pair as "no VO". */
int64_t orc_domain_count(const orc_env *h);

/* Call-level check of rvo_inter.config_vo_circle2 (rvo_inter.py:116-196) for one pair:
 * self8 / other8 = [x y z vx vy vz radius priority].  obs9 is the reference's first list
 * (early returns included), exp_time its third element, min_dis its fifth. */
void orc_vo_circle2(int env_train, const double *self8, const double *other8,
                    const double *action, double *obs9, int32_t *vo_flag, double *exp_time,
                    int32_t *collision, double *min_dis, int32_t *domain_error);

/* Worker threads for the env loop (OpenMP); 1 = scalar port. */
void orc_set_threads(orc_env *h, int n);

/* Scalar helpers exported for unit tests. */
double orc_py_round2(double x);            /* Python round(x, 2)            */
double orc_np_round(double x, int decimals); /* numpy round: rint(x*10^d)/10^d */

#ifdef __cplusplus
}
#endif
#endif
