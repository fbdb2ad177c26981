/*
 * rvo3d.h -- C-ABI of the MI355X-native batched 3D-RVO drone environment
 * (librvo3d_hip.so; HIP kernels for gfx950 behind plain-C entry points).
 *
 * The reference (ZSHCRWY25/3DRVO-MARL-CollisionAvoidance) is pure Python and
 * has no FFI of its own; the boundary this library drops in under is the
 * Python surface of `uaisa_env.drone_envs.mdin.mdin`.  Each entry point names
 * the reference method it replaces (paths relative to the reference root).
 * The Python binding a maintainer adds is shown in INTEGRATION.md.
 *
 * Conventions
 *   - E environments x N drones; drone (e, d) has flat index e*N + d.
 *   - Unless marked HOST, every pointer is a DEVICE pointer owned by the
 *     caller (e.g. torch tensors) and only borrowed for the call.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  All
 *     work is enqueued on it; nothing synchronises unless stated.
 *   - Every function returns RVO3D_OK or a negative error; the message for
 *     the calling thread's last error is rvo3d_last_error().  Nothing throws.
 *   - A handle is bound to one device, is not re-entrant, and owns its state
 *     buffers; step/observe allocate nothing.
 */
#ifndef RVO3D_H
#define RVO3D_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RVO3D_VERSION 1

enum {
  RVO3D_OK = 0,
  RVO3D_ERR_INVALID = -1,  /* bad argument / unsupported configuration */
  RVO3D_ERR_HIP = -2,      /* a HIP runtime call failed                */
  RVO3D_ERR_STATE = -3     /* call order (e.g. step before load_world) */
};

enum { RVO3D_F32 = 0, RVO3D_F64 = 1, RVO3D_BF16 = 2 };

/* bits of the device error word (rvo3d_error_flags) */
enum {
  RVO3D_FLAG_NONFINITE_OBS = 1, /* an observation held NaN/Inf: the reference
                                   raises ValueError here (ir_gym.py:232-239) */
  RVO3D_FLAG_DOMAIN_ERROR = 2   /* env_train = 0 only: a pair with r - 0.2 + mr < dis < r + mr
                                   was approaching; the reference's get_alpha raises
                                   ValueError("math domain error") there (vel_obs3D.py:13,
                                   rvo_inter.py:144-165) and aborts the step; here the pair
                                   counts as "no velocity obstacle" and the step completes */
};

typedef struct rvo3d_env rvo3d_env;

typedef struct rvo3d_config {
  int32_t num_envs;       /* E >= 1                                            */
  int32_t num_drones;     /* N, 1..512   (data_1.json "drone_num")             */
  int32_t max_points;     /* P >= 2: longest route, in waypoints               */
  int32_t num_buildings;  /* nb >= 0, shared by all envs ("building_list")     */
  int32_t neighbors_num;  /* nm: VO rows kept per observation (mdin.py:7, 10)  */
  int32_t env_train;      /* rvo_inter.py:14 (1 = reference default)           */
  int32_t device;         /* HIP device ordinal                                */
  int32_t action_decimals;/* >= 0: actions are re-quantised on device to that
                             many decimals in fp64 (rint(a*10^d)/10^d), so a
                             float32 buffer of 2-decimal values is widened to
                             exactly the fp64 the reference steps with
                             (multi_ppo.py:205); -1: widen as is               */
  double map_size[3];     /* data_1.json "map_size"                            */
} rvo3d_config;

/* Device views (read-only, see rvo3d_state_ptrs) of the handle's struct-of-arrays state,
 * each [E*N]. Valid until rvo3d_destroy. (reach-through used by the trainer: drone_list[i].vel,
 * multi_ppo.py:202; indicators_*, ir_gym.py:414-420) */
typedef struct rvo3d_state_view {
  double *px, *py, *pz;       /* drone.state                                   */
  double *vx, *vy, *vz;       /* drone.vel                                     */
  double *yaw, *pitch;        /* degrees (drone.py:68-69)                      */
  double *real_len;           /* drone.real_route_len                          */
  double *max_dev;            /* drone.max_deviation                           */
  double *extra_len;          /* drone.extra_len                               */
  int32_t *wp_idx;            /* drone.i                                       */
  uint8_t *arrive, *dest;     /* arrive_flag, destination_arrive_flag          */
} rvo3d_state_view;

/* mdin.__init__ -> ir_gym.__init__ -> env_base.__init__ (mdin.py:7,
 * ir_gym.py:18, env_base.py:15): allocate state for E x N drones. */
int rvo3d_create(const rvo3d_config *cfg, rvo3d_env **out);
int rvo3d_destroy(rvo3d_env *h);

/* env_base.load_data + env_drone.__init__ (env_base.py:26-47,
 * env_drones.py:13-32).  HOST pointers: waypoints [E][N][P][3] (routes shorter
 * than P padded with anything), n_points [E][N] (each 2..P), buildings
 * [nb][4] = x,y,h,r, radius / priority [E][N] or NULL for 0.2 / 5
 * (drone.py:14-15).  Copies to the device, computes route lengths
 * (drone.py:409-429) and puts every drone in its reset state.  Synchronises. */
int rvo3d_load_world(rvo3d_env *h, const double *waypoints, const int32_t *n_points,
                     const double *buildings, const double *radius,
                     const double *priority, void *stream);

/* env_drone.drones_reset (env_drones.py:99) for the envs whose mask byte is
 * non-zero (env_mask NULL = every env). env_mask [E]. */
int rvo3d_reset(rvo3d_env *h, const uint8_t *env_mask, void *stream);
/* mdin.drone_reset_one (mdin.py:43) for every drone whose mask byte is
 * non-zero. drone_mask [E][N]. */
int rvo3d_reset_drones(rvo3d_env *h, const uint8_t *drone_mask, void *stream);

/* ir_gym.env_observation / the observation half of ir_gym.env_reset
 * (ir_gym.py:360-383): observations of every drone with action = 0.
 * obs [E][N][12+9*nm] float32, rows beyond vo_count zero; vo_count [E][N]
 * (0 = the reference's single all-zero VO row). */
int rvo3d_observe(rvo3d_env *h, float *obs, int32_t *vo_count, void *stream);

/* mdin.drone_step (mdin.py:19-30): RVO reward sweep on the pre-move state,
 * kinematic integration, observation / reward / termination sweep on the
 * post-move state.  actions [E][N][3] of action_dtype (RVO3D_F32 / RVO3D_F64).
 * reward [E][N] float32 = rvo_reward + mov_reward (may be inf/nan exactly where
 * the reference's is, ir_gym.py:88); done = collision, info = arrive_flag,
 * finish = destination_arrive_flag (ir_gym.py:248-250), all [E][N] bytes. */
int rvo3d_step(rvo3d_env *h, const void *actions, int32_t action_dtype, float *obs,
               int32_t *vo_count, float *reward, uint8_t *done, uint8_t *info,
               uint8_t *finish, void *stream);

/* rvo3d_step fused with the caller protocol of multi_ppo.py:230-242/266-281:
 * drones with done|finish are reset and every env that reset a drone has all
 * its observations recomputed with action = 0 (ir_gym.env_observation).
 * reset_mask [E][N] (nullable) reports the drones that were reset; reward /
 * done / info / finish are those of the step itself. */
int rvo3d_step_autoreset(rvo3d_env *h, const void *actions, int32_t action_dtype,
                         float *obs, int32_t *vo_count, float *reward, uint8_t *done,
                         uint8_t *info, uint8_t *finish, uint8_t *reset_mask,
                         void *stream);

/* The step driven by raw policy samples: the trainer's glue between `ac.step` and
 * `env.drone_step` (train/policy/multi_ppo.py:196-210) runs on the device, in numpy's
 * own types:  a = np.round(a_inc, 2) (float32);  action = np.round(acceler * a +
 * drone.vel, 2) (float32 product widened, float64 sum).  a_inc [E][N][3] float32;
 * acceler = ir_gym.acceler (0.5).  autoreset != 0 fuses the reset protocol as
 * rvo3d_step_autoreset does (reset_mask nullable). */
int rvo3d_step_policy(rvo3d_env *h, const float *a_inc, float acceler, float *obs,
                      int32_t *vo_count, float *reward, uint8_t *done, uint8_t *info,
                      uint8_t *finish, uint8_t *reset_mask, int32_t autoreset, void *stream);

/* ---- the trainer's per-step glue around the env step (train/policy/multi_ppo.py:193-281), no handle:
 * plain device pointers, work enqueued on `stream` of the current device ---- */

/* `a, v, logp = ac.step(obs)` (multi_ppo.py:195; policy_rnn_ac.py:57-69, 197-235) from the last hidden
 * layers on, for `rows` observations at once, plus `a = np.round(a, 2)` (:197) and the stores
 * `buf.store(.., a, .., v, logp)` (:217-221): the actor's head (hidden -> 3, tanh), the critic's head
 * (hidden -> 1), a ~ Normal(mu, clamp(std_factor * exp(log_std) + 1e-6, 1e-4, 10)) from a counter-based
 * generator (Philox4x32-10, counter = (row, step), key = seed: reproducible, no generator state), the
 * log-probability of the unrounded sample.  hidden > 0: h_pi / h_v are the hidden activations
 * [rows][ld] of dtype RVO3D_F32 or RVO3D_BF16 (hidden a multiple of 128 resp. 256, at most 1024), the
 * head weights float32 as nn.Linear stores them (w_pi [3][hidden], b_pi [3], w_v [hidden], b_v [1]).
 * hidden == 0: h_pi = mu [rows][ld_pi >= 3] and h_v = v [rows][ld_v >= 1], float32, from the caller's own
 * network (tanh_out ignored).  Outputs: act [rows][3] (rounded: what the buffer stores and
 * rvo3d_step_policy steps from), logp [rows], val [rows]; dbg_mu / dbg_raw [rows][3] nullable. */
typedef struct rvo3d_policy_heads {
  const void *h_pi, *h_v;
  int64_t ld_pi, ld_v;
  int32_t dtype, hidden, tanh_out, reserved;
  const float *w_pi, *b_pi, *w_v, *b_v, *log_std;
} rvo3d_policy_heads;
int rvo3d_policy_sample(const rvo3d_policy_heads *heads, int64_t rows, float std_factor, uint64_t seed,
                        uint64_t step, float *act, float *logp, float *val, float *dbg_mu,
                        float *dbg_raw, void *stream);

/* Config 3's policy step - MLP(256, 256) actor and critic (train/policy/policy_rnn_ac.py:197-257 with the hidden sizes
 * of BASELINE config 3; ac.step, :57-69) - as ONE kernel over the env's observation rows: cast, both hidden layers and
 * the heads on the matrix cores with the activations kept in registers (bf16 operands, float32 accumulation), then what
 * rvo3d_policy_sample does per row (tanh, sample, log-probability, np.round(a, 2), the stores).  Replaces, per rollout
 * step, the observation cast, three GEMMs and rvo3d_policy_sample.
 * rvo3d_policy_mlp_pack turns the nn.Linear tensors (float32, weight [out][in], as the modules store them; widths
 * obs_width -> 256 -> 256 -> 3 for the actor, -> 1 for the critic) into the device blob the kernel reads
 * (rvo3d_policy_mlp_blob_bytes(obs_width) bytes, 16-byte aligned; repack after every optimizer step).
 * obs [rows][obs_ld] float32, obs_width <= 126.  Noise as rvo3d_policy_sample: Philox4x32-10, counter (row, step).
 * vo_count (optional, NULL = none): the env's count output [rows] for these rows.  A row of the env holds state_dim
 * floats, then row_dim floats per velocity-obstacle row, vo_count of them, then zeros: with the counts the
 * kernel neither loads nor multiplies the 16-float column groups that are zero for all 32 rows a wave holds (exactly
 * the same sums: a zero activation adds nothing).  Pass NULL for rows that do not keep that promise. */
typedef struct rvo3d_mlp_weights {
  const float *w1, *b1; /* [256][obs_width], [256] */
  const float *w2, *b2; /* [256][256], [256] */
  const float *w3, *b3; /* [3][256], [3] (actor) or [1][256], [1] (critic) */
} rvo3d_mlp_weights;
int64_t rvo3d_policy_mlp_blob_bytes(int32_t obs_width);
int rvo3d_policy_mlp_pack(const rvo3d_mlp_weights *pi, const rvo3d_mlp_weights *v, int32_t obs_width, void *blob,
                          void *stream);
int rvo3d_policy_mlp_sample(const void *blob, int32_t obs_width, const float *obs, int64_t obs_ld, int64_t rows,
                            const int32_t *vo_count, int32_t state_dim, int32_t row_dim, int32_t tanh_out, const float *log_std, float std_factor, uint64_t seed, uint64_t step,
                            float *act, float *logp, float *val, float *dbg_mu, float *dbg_raw, void *stream);

/* The reader's features (rnn_Reader.obs_rnn + LayerNorm, train/policy/policy_rnn_ac.py:75-127) of rows WITHOUT a
 * velocity-obstacle row, in collapsed form.  The GRU of such a row sees a zero input from h = 0: its hidden state h0 is
 * the same for every row, and LayerNorm(concat(p, h0)) depends on the row only through mean and rstd, so a linear
 * layer W on the features is  W_p f_p + rstd a - (mean rstd) b + c  with a = W_h (h0 g_h), b = W_h g_h, c = W_h b_h + bias
 * (g, b: the LayerNorm's affine).  out [rows][out_ld >= state_dim + 8] receives per row f_p (state_dim floats), rstd as
 * bf16 head, head, tail, mean rstd likewise, 1, 1: the inputs of rvo3d_policy_mlp_sample with obs_width = state_dim + 8
 * and first-layer weights [W_p | a_hi a_lo a_hi | -b_hi -b_lo -b_hi | c_hi c_lo], zero bias (the caller builds them once
 * per optimizer step: rvo3d_amd.policy.policy_rnn_ac.rnn_ac.zero_vo_plan).  sum_h0 / sumsq_h0: the sums of h0 and h0^2,
 * feat_dim = state_dim + hidden.  Rows that do have VO rows get wrong values: the caller recomputes those (below). */
int rvo3d_reader_zero_features(const float *obs, int64_t obs_ld, int64_t rows, int32_t state_dim, int32_t feat_dim,
                               const float *ln_w, const float *ln_b, float sum_h0, float sumsq_h0, float ln_eps,
                               float *out, int64_t out_ld, const int32_t *vo_count, int32_t *list, int32_t *count,
                               void *stream);

/* ... and the rows that DO have velocity-obstacle rows: rvo3d_reader_zero_features, given the env's vo_count, appends
 * their indices to list [rows] (count [1], zero before the first call), and rvo3d_policy_rows computes the policy step
 * of every listed row exactly as the modules do, in float32 (rnn_Reader: the (bi)GRU over the row's vo_count rows,
 * direction sum, concat, LayerNorm - policy_rnn_ac.py:75-168; GaussianActor / Critic stacks state_dim + hidden -> 256
 * -> 256 -> 3 / 1 - :197-257; sample, log-probability, np.round, stores as rvo3d_policy_sample), two workgroups per row
 * (one per network),
 * overwriting what rvo3d_policy_mlp_sample wrote for it; it resets count for the next step (done_blocks [1]: scratch,
 * zero before the first call).  No host synchronisation.  Meant for SHORT lists (a rollout of the benchmark's world has
 * a dozen such rows among 262 144): weights are read as the modules store them, a row is a latency chain of ~35 us. */
typedef struct rvo3d_rnn_policy {
  const float *w_ih_f, *w_hh_f, *b_ih_f, *b_hh_f; /* nn.GRU: [3 hidden][in_dim], [3 hidden][hidden], [3 hidden] x 2 */
  const float *w_ih_r, *w_hh_r, *b_ih_r, *b_hh_r; /* reverse direction, all NULL for a unidirectional GRU */
  const float *ln_w, *ln_b;                       /* [state_dim + hidden] */
  int32_t hidden, in_dim, state_dim, slots;       /* slots = neighbors_num (VO rows per observation) */
  float ln_eps;
  int32_t reserved;
  rvo3d_mlp_weights pi, v;                        /* widths state_dim + hidden -> 256 -> 256 -> 3 / 1 */
} rvo3d_rnn_policy;
int rvo3d_policy_rows(const rvo3d_rnn_policy *net, const float *obs, int64_t obs_ld, const int32_t *vo_count,
                      const int32_t *list, int32_t *count, int32_t *done_blocks, int32_t tanh_out, const float *log_std,
                      float std_factor, uint64_t seed, uint64_t step, float *act, float *logp, float *val, void *stream);

/* Replaying a rollout step as a HIP graph.  The three (MLP policy) or five (biGRU policy) launches of a rollout step take
 * every argument by value, the noise counter `step` included: a captured graph would draw the same noise on every
 * replay.  With a device counter registered here (uint64 in device memory, or NULL to unregister; process-wide), every
 * sampling launch (rvo3d_policy_sample / _mlp_sample / _rows) uses step + *counter, and rvo3d_rollout_account advances
 * the counter by one per call - so a caller captures [policy, rvo3d_step_policy, rvo3d_rollout_account] once per buffer
 * slot and replays it every epoch (rvo3d_amd.policy.multi_ppo, graph_rollout=True; measured on MI355X: the same time
 * per step as the stream launches, 0.158 vs 0.157-0.161 ms). */
int rvo3d_rollout_set_step_counter(uint64_t *device_counter);

/* rnn_Reader.obs_rnn (train/policy/policy_rnn_ac.py:75-127) for observations with AT MOST ONE velocity-obstacle row
 * - nearly all of a rollout's -: the (bi)GRU over a one-step sequence from h = 0 (one cell evaluation per direction,
 * no recurrent product), the sum of the two directions, the concatenation with the proprioceptive part and the
 * LayerNorm, in one pass.  obs [rows][obs_ld] float32 (the env's rows: state_dim proprioceptive floats, then the first
 * VO row of in_dim floats); weights float32 as nn.GRU / nn.LayerNorm store them (w_ih [3 hidden][in_dim], b_ih / b_hh
 * [3 hidden], gates r, z, n; the *_r pointers NULL for a unidirectional GRU; ln_w / ln_b [state_dim + hidden]).
 * feat [rows][feat_ld] of feat_dtype RVO3D_F32 / RVO3D_BF16 receives the state_dim + hidden features (columns beyond
 * stay untouched: the caller's zero padding for the next GEMM).  hidden 64 / 128 / 192 / 256, in_dim 9, state_dim <= 32.
 * Rows whose vo_count exceeds 1 need the full recurrence: the caller recomputes those afterwards. */
typedef struct rvo3d_gru_reader {
  const float *w_ih_f, *b_ih_f, *b_hh_f;
  const float *w_ih_r, *b_ih_r, *b_hh_r;
  const float *ln_w, *ln_b;
  int32_t hidden, in_dim, state_dim;
  float ln_eps;
} rvo3d_gru_reader;
int rvo3d_reader_first_step(const rvo3d_gru_reader *reader, const float *obs, int64_t obs_ld, int64_t rows,
                            void *feat, int32_t feat_dtype, int64_t feat_ld, void *stream);

/* The bookkeeping behind `env.drone_step` in the rollout loop (multi_ppo.py:217-281), for E envs x N drones
 * (N <= 512): the reward into its buffer slot (inf / nan as 0 when sanitize), episode return / length
 * counters, which paths end behind this step (cut_slot [E]: any drone of the env finished or timed out,
 * or the epoch ends - finish_path(0) for every drone of the env, :279), which drones the trainer still has
 * to reset (extra_mask [E][N]: timeouts and, at the epoch's end, everybody the fused step did not reset;
 * *any_extra |= 1 if there is one), and per env the sum / number of finished episodes' returns
 * (sums [E][2], doubles, accumulated: the caller zeroes them and adds them up). */
int rvo3d_rollout_account(int32_t num_envs, int32_t num_drones, const float *reward, const uint8_t *done,
                          const uint8_t *finish, int32_t sanitize, int32_t max_ep_len, int32_t epoch_end,
                          float *rew_slot, float *ep_ret, int32_t *ep_len, uint8_t *cut_slot,
                          uint8_t *extra_mask, double *sums, int32_t *any_extra, void *stream);

/* mdin.drone_step returns its rewards as Python floats (mdin.py:28: rvo_reward + mov_reward in
 * float64).  Attach a device buffer reward64 [E][N] and every following step (all three step
 * entry points) also writes the float64 value next to the float32 one; NULL detaches.  The
 * buffer is borrowed until detached or the handle is destroyed. */
int rvo3d_set_reward_f64(rvo3d_env *h, double *reward64);

/* ir_gym.cal_des_list (ir_gym.py:44): desired velocity, des_vel [E][N][3] f64. */
int rvo3d_des_vel(rvo3d_env *h, double *des_vel, void *stream);

/* reciprocal_vel_obs.cal_vel (uaisa_env/vel_obs/reciprocal_vel_obs.py:19-31) for every drone
 * on the current state: candidate velocities on a 0.5 grid inside clip([v - acceler,
 * v + acceler], -vmax, vmax), |v| >= 0.3, tested against the reciprocal velocity obstacle
 * of every drone within 10 m; the one closest to the desired velocity outside all cones,
 * else the inside one with the least penalty, else 0.  The reference class cannot run as
 * written; this is the algorithm it spells out, on the helper functions it calls
 * (parity unpinned for the driver loop).  vmax: HOST [3]; 0 <= acceler
 * <= 1; out_vel [E][N][3] f64. */
int rvo3d_rvo_vel(rvo3d_env *h, const double *vmax, double acceler, double *out_vel,
                  void *stream);

/* Zero-copy views of the state arrays, READ-ONLY: the step keeps values derived from the
 * state on file between calls (des_vel, the in-range words of the pair gate, the current /
 * previous waypoint); state is changed through rvo3d_set_state / rvo3d_reset*, which bring
 * them up to date or mark them stale. */
int rvo3d_state_ptrs(rvo3d_env *h, rvo3d_state_view *out);
/* Array-of-structs copies: pos/vel [E][N][3] f64, the rest [E][N]; any
 * pointer may be NULL.  set_state is for tests and checkpoint restore. */
int rvo3d_get_state(rvo3d_env *h, double *pos, double *vel, double *yaw, double *pitch,
                    double *real_len, double *max_dev, double *extra_len,
                    int32_t *wp_idx, uint8_t *arrive, uint8_t *dest, void *stream);
int rvo3d_set_state(rvo3d_env *h, const double *pos, const double *vel,
                    const double *yaw, const double *pitch, const double *real_len,
                    const double *max_dev, const double *extra_len,
                    const int32_t *wp_idx, const uint8_t *arrive, const uint8_t *dest,
                    void *stream);

/* Reads (and clears) the device error word into *flags (HOST).  Synchronises
 * the stream: opt-in, keep it out of the rollout loop. */
int rvo3d_error_flags(rvo3d_env *h, uint32_t *flags, void *stream);

/* Launch geometry chosen for this handle (diagnostics / bench):
 * threads per block, envs per block, blocks, dynamic LDS bytes. */
int rvo3d_launch_info(rvo3d_env *h, int32_t *threads, int32_t *envs_per_block,
                      int32_t *blocks, int32_t *lds_bytes);

/* The kernel instantiation this handle's calls launch, as rocprofv3 names it
 * ("rvo3d::env_kernel<MODE, NW, NFIX, TRAIN, PAD>"; mode 0 = rvo3d_observe, 1 = rvo3d_step,
 * 2 = rvo3d_step_autoreset / rvo3d_step_policy with autoreset), written to buf (HOST, cap bytes,
 * NUL-terminated): bench.py's roofline.kernel. */
int rvo3d_kernel_name(rvo3d_env *h, int32_t mode, char *buf, int32_t cap);

/* (Diagnostics - phase stamps, phase ablation - are not part of this library: they exist
 * only in the -DRVO3D_DIAG build that tools/ makes for itself, include/rvo3d_diag.h.) */

int rvo3d_version(void);
const char *rvo3d_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
