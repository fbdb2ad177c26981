/*
 * rvo3d_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Scalar fp64 restatement of the reference step, written to follow the
 * reference's evaluation order operation by operation.  Every function cites
 * the reference file:line it restates (paths relative to the reference root).
 *
 * Arithmetic model of the reference's numpy/libm calls, measured in the build
 * container (numpy 2.2.6 + scipy-openblas 0.3.29, glibc libm):
 *   x ** 2                 -> pow(x, 2.0)            (python float and np.float64)
 *   np.dot / np.linalg.norm on 2/3-vectors
 *                          -> fma chain  fma(z,z, fma(y,y, x*x))   (OpenBLAS ddot tail)
 *   np.sin / np.cos        -> glibc sin / cos (bit-identical on 200k samples)
 *   np.arccos/arctan2/exp/log -> numpy SIMD kernels, <= 1 ulp from glibc in
 *                          0.1-9 % of calls; this file uses glibc.  Every value
 *                          they feed is rounded to 2-3 decimals before it is
 *                          observable, see DESIGN.md "ulp budget".
 * Compile with -ffp-contract=off: the only fused operations are the explicit
 * fma() calls.
 */
#define _GNU_SOURCE
#include "rvo3d_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define GOAL_THRESHOLD 0.4   /* drone.py:15  goal_threshold            */
#define NEIGHBOR_GATE 10.0   /* rvo_inter.py:96 literal                */
#define BUILDING_GATE 5.0    /* rvo_inter.py:104 literal               */
#define CTIME_THRESHOLD 2.0  /* rvo_inter.py:11 ctime_threshold        */
#define EXP_RADIUS 0.2       /* rvo_inter.py:11 exp_radius             */
#define MAX_ACC 1.0          /* drone.py:72                            */
#define MAX_ANGLE 90.0       /* drone.py:73                            */
#define DEG2RAD 0.017453292519943295 /* np.deg2rad: x * (pi/180)       */

struct orc_env {
  int E, N, P, nb, nm, env_train, threads;
  double map[3];
  /* static world */
  double *wp;        /* [E*N][P][3] */
  int32_t *n_points; /* [E*N] */
  double *route_len; /* [E*N]   drone.py:31 */
  double *radius, *prio;
  double *bld;       /* [nb][4] */
  /* mutable state, drone.py:14-82 */
  double *p, *v;     /* [E*N][3] */
  double *yaw, *pitch, *real_len, *max_dev, *extra_len;
  int32_t *wp_idx;
  uint8_t *arrive, *dest;
  int64_t nan_count;
  int64_t domain_count; /* pairs on which the reference raises "math domain error" (get_alpha) */
  double *margin;     /* [E*N] min decision margin of the last call (audit) */
  int32_t *msite;     /* [E*N] source line of the decision that set it */
  uint8_t *vnoise;    /* [E*N] velocity came out of a cancelled speed (pure libm noise) */
};

/* ---- arithmetic primitives ------------------------------------------- */
#ifndef ORC_VARIANT
static inline double sq(double x) { return pow(x, 2.0); }
static inline double dot3_blas(const double *a, const double *b) {
  return fma(a[2], b[2], fma(a[1], b[1], a[0] * b[0]));
}
static inline double norm2_blas(double x, double y) { return sqrt(fma(y, y, x * x)); }
#else
/* ulp-perturbed variant, used only by gen_golden.py's margin audit: a fixture
 * whose outputs change under this arithmetic sits on a decision boundary. */
static inline double sq(double x) { return x * x; }
static inline double dot3_blas(const double *a, const double *b) {
  return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
}
static inline double norm2_blas(double x, double y) { return sqrt(x * x + y * y); }
#endif
static inline double norm3_blas(const double *a) { return sqrt(dot3_blas(a, a)); }

/* ---- decision-margin audit ------------------------------------------------
 * Every branch / rounding whose outcome could flip under a 1-ulp change of a
 * libm result reports |value - threshold| here; the minimum per drone and call
 * is kept in h->margin.  Tests allow a device/oracle mismatch only on samples
 * whose margin is below 1e-9 (the reference's own result is libm noise there). */
static _Thread_local double *tl_margin = 0;
static _Thread_local int32_t *tl_site = 0;
static inline void mgs0(int site, double d) {
  d = fabs(d);
  if (tl_margin && d < *tl_margin) { *tl_margin = d; *tl_site = site; }
}
/* value == threshold EXACTLY is reproducible (exact input data: integer
 * coordinates, 3-4-5 distances, acos(0)); only a near tie is a knife edge.
 * The one exception is the speed cancellation in move_forward (mg0 there). */
static inline void mgs(int site, double d) { if (d != 0) mgs0(site, d); }
static inline void mgs_round(int site, double x, double f) { /* distance of x to a rounding tie */
  if (!isfinite(x)) return;
  double p = x * f;
  mgs(site, (fabs(p - floor(p) - 0.5)) / f);
}
/* Observation outputs are libm results (positions and velocities come out of sin / cos, the
 * rows out of sqrt / asin / acos): a value EXACTLY on a rounding tie - e.g. vy = 0.3 * sin 18
 * * cos 36 = 0.075 - falls to either side with the last ulp of the libm in use, so for them
 * an exact tie is a knife edge too. */
static inline void mgs_round0(int site, double x, double f) {
  if (!isfinite(x)) return;
  double p = x * f;
  mgs0(site, (fabs(p - floor(p) - 0.5)) / f);
}
#define mg_round0(x, f) mgs_round0(__LINE__, (x), (f))
#define mg(d) mgs(__LINE__, (d))
#define mg0(d) mgs0(__LINE__, (d))
#define mgx(d) mgs(__LINE__, (d))
#define mg_round(x, f) mgs_round(__LINE__, (x), (f))

double orc_np_round(double x, int decimals) {
  /* numpy round: multiply, rint, true_divide */
  double f = decimals == 2 ? 100.0 : (decimals == 3 ? 1000.0 : pow(10.0, decimals));
  return rint(x * f) / f;
}
static inline double np_round2(double x) { return rint(x * 100.0) / 100.0; }
static inline double np_round3(double x) { return rint(x * 1000.0) / 1000.0; }

double orc_py_round2(double x) {
  /* Python float round(x, 2): correctly rounded decimal of the exact binary
   * value, ties to even, then nearest double (floatobject.c double_round). */
  if (!isfinite(x)) return x;
  double p = x * 100.0;
  double e = fma(x, 100.0, -p); /* exact: x*100 = p + e */
  double c = floor(p);
  double d = (p - (c + 0.5)) + e;
  if (d > 0.0) c += 1.0;
  else if (d == 0.0 && fmod(c, 2.0) != 0.0) c += 1.0;
  return c / 100.0;
}

/* ---- drone.py helpers -------------------------------------------------- */
typedef struct {
  double s[12]; /* x y z vx vy vz r prio dvx dvy dvz deviation (drone.py:263) */
} dstate;

static inline const double *wp_at(const orc_env *h, int g, int k) {
  return h->wp + ((size_t)g * h->P + k) * 3;
}

/* drone.cal_des_vel + relative + angles_to_direction
 * (drone.py:199-210, 340-352, 319-328) */
static void cal_des_vel(const double *p, const double *cur_des, double *out) {
  double dif[3] = {cur_des[0] - p[0], cur_des[1] - p[1], cur_des[2] - p[2]};
  double dis = norm3_blas(dif);
  double az = atan2(dif[1], dif[0]);
  double el = (dis != 0.0) ? atan2(dif[2], norm2_blas(dif[0], dif[1])) : 0.0;
  mg(dis - GOAL_THRESHOLD);
  if (dis > GOAL_THRESHOLD) {
    double dir[3] = {cos(az) * cos(el), sin(az) * cos(el), sin(el)};
    for (int k = 0; k < 3; ++k) { mg_round(dir[k], 1000.0); out[k] = np_round3(1.0 * dir[k]); }
  } else {
    out[0] = out[1] = out[2] = 0.0;
  }
}

/* drone.calculate_deviation (drone.py:366-406): distance to the infinite line */
static double calc_deviation(const double *a, const double *b, const double *p) {
  double dx = b[0] - a[0], dy = b[1] - a[1], dz = b[2] - a[2];
  double mag = sqrt(sq(dx) + sq(dy) + sq(dz));
  if (mag == 0.0) return 0.0;
  double hx = dx / mag, hy = dy / mag, hz = dz / mag;
  double px = p[0] - a[0], py = p[1] - a[1], pz = p[2] - a[2];
  double t = px * hx + py * hy + pz * hz;
  double qx = a[0] + t * hx, qy = a[1] + t * hy, qz = a[2] + t * hz;
  return sqrt(sq(p[0] - qx) + sq(p[1] - qy) + sq(p[2] - qz));
}

static inline void cur_prev_des(const orc_env *h, int g, const double **cur,
                                const double **prev) {
  int i = h->wp_idx[g];
  *cur = wp_at(h, g, i);
  *prev = wp_at(h, g, i - 1);
}

/* drone.dronestate (drone.py:254-263), incl. the max_deviation side effect */
static void dronestate(orc_env *h, int g, dstate *o) {
  tl_margin = h->margin + g; tl_site = h->msite + g;
  const double *cur, *prev;
  cur_prev_des(h, g, &cur, &prev);
  const double *p = h->p + 3 * (size_t)g, *v = h->v + 3 * (size_t)g;
  o->s[0] = p[0]; o->s[1] = p[1]; o->s[2] = p[2];
  o->s[3] = v[0]; o->s[4] = v[1]; o->s[5] = v[2];
  o->s[6] = h->radius[g];
  o->s[7] = h->prio[g];
  cal_des_vel(p, cur, &o->s[8]);
  double dev = calc_deviation(prev, cur, p);
  if (dev > h->max_dev[g]) h->max_dev[g] = dev;
  o->s[11] = dev;
}

static inline int arrive(const double *p, const double *des) {
  double d[3] = {p[0] - des[0], p[1] - des[1], p[2] - des[2]};
  double dist = norm3_blas(d);
  mg(dist - GOAL_THRESHOLD);
  return dist <= GOAL_THRESHOLD; /* drone.py:172-179 */
}
/* drone.destination_arrive (drone.py:182-192): side effect on extra_len */
static inline int destination_arrive(orc_env *h, int g) {
  const double *dst = wp_at(h, g, h->n_points[g] - 1);
  if (arrive(h->p + 3 * (size_t)g, dst)) {
    h->extra_len[g] = h->real_len[g] - h->route_len[g];
    return 1;
  }
  return 0;
}

/* numpy float remainder (npy_divmod), used by `% 360` at drone.py:457 */
static inline double np_mod(double a, double b) {
  double m = fmod(a, b);
  if (m != 0.0) {
    if ((b < 0.0) != (m < 0.0)) m += b;
  } else {
    m = copysign(0.0, b);
  }
  return m;
}
static inline double clampd(double x, double lo, double hi) {
  return x < lo ? lo : (x > hi ? hi : x); /* np.clip */
}

/* env_base.drone_step -> drone.move_forward (env_base.py:135-144,
 * drone.py:96-129), with kinematicstep (drone.py:435-490).  `stop` receives
 * map_size (truthy) because of the argument shift at env_base.py:142. */
static void move_forward(orc_env *h, int g, const double *act) {
  tl_margin = h->margin + g; tl_site = h->msite + g;
  double *p = h->p + 3 * (size_t)g, *v = h->v + 3 * (size_t)g;
  double speed = norm3_blas(v);                           /* drone.py:103 */
  double acc = clampd(act[0] * MAX_ACC, -MAX_ACC, MAX_ACC);
  double dyaw = clampd(act[1] * MAX_ANGLE, -MAX_ANGLE, MAX_ANGLE);
  double dpit = clampd(act[2] * MAX_ANGLE, -MAX_ANGLE, MAX_ANGLE);
  double nv = speed + acc * 1;
  /* speed clamp at 0: when ||v|| (libm noise in its last bit) cancels against
   * the decimal acc, the sign of nv - and with it whether the new velocity is
   * 0 or 1e-17 * direction - is noise.  v == 0 exactly gives nv = acc: robust. */
  h->vnoise[g] = 0;
  if (v[0] != 0 || v[1] != 0 || v[2] != 0) {
    mg0(nv);
    if (fabs(nv) < 1e-9) h->vnoise[g] = 1;
  }
  speed = (0.0 > nv) ? 0.0 : nv;                          /* python max(nv, 0), drone.py:452 */
  h->yaw[g] = np_mod(h->yaw[g] + dyaw, 360.0);
  h->pitch[g] = clampd(h->pitch[g] + dpit, -90.0, 90.0);
  double yr = h->yaw[g] * DEG2RAD, pr = h->pitch[g] * DEG2RAD;
  double vel[3] = {speed * cos(pr) * cos(yr), speed * cos(pr) * sin(yr),
                   speed * sin(pr)};
  if (h->dest[g]) vel[0] = vel[1] = vel[2] = 0.0;         /* drone.py:107-109 */
  double prev[3] = {p[0], p[1], p[2]};
  for (int k = 0; k < 3; ++k) {
    p[k] = p[k] + vel[k] * 1;                             /* drone.py:167, dt = 1 */
    v[k] = vel[k];
  }
  double d[3] = {p[0] - prev[0], p[1] - prev[1], p[2] - prev[2]};
  h->real_len[g] = h->real_len[g] + norm3_blas(d);       /* drone.py:86-90 */
  const double *cur = wp_at(h, g, h->wp_idx[g]);
  if (arrive(p, cur) && !destination_arrive(h, g)) {      /* drone.py:116 */
    if (h->wp_idx[g] < h->n_points[g] - 1) {              /* drone.py:117,123 */
      h->wp_idx[g] += 1;
      h->arrive[g] = 0;
    }
  }
}

/* ---- vel_obs ---------------------------------------------------------- */
typedef struct {
  double obs[9]; /* PAA(3) rel(3) alpha min_dis iet */
  int flag, collision, domain;
  double t, min_dis;
} vo_inf;

/* vel_obs3D.cal_vo_exp_tim (vel_obs3D.py:145-182) */
static double cal_vo_exp_tim(double rx, double ry, double rz, double rvx,
                             double rvy, double rvz, double ra, double rb) {
  double r = ra + rb;
  double ux = -rvx, uy = -rvy, uz = -rvz;
  double a = sq(ux) + sq(uy) + sq(uz);
  double b = 2 * rx * ux + 2 * ry * uy + 2 * rz * uz;
  double c = sq(rx) + sq(ry) + sq(rz) - sq(r);
  mg(c);
  if (c <= 0) return 0.0;
  double temp = sq(b) - 4 * a * c;
  mg(temp);
  if (temp <= 0) return INFINITY;
  double t1 = (-b + sqrt(temp)) / (2 * a);
  double t2 = (-b - sqrt(temp)) / (2 * a);
  mg(t1); mg(t2);
  if (t1 < 0 && t2 < 0) return -1.0;
  double t3 = t1 >= 0 ? t1 : INFINITY;
  double t4 = t2 >= 0 ? t2 : INFINITY;
  return t3 < t4 ? t3 : t4; /* python min(t3, t4): first arg wins ties / NaN */
}

/* rvo_inter.config_vo_circle2 (rvo_inter.py:116-196) with get_alpha,
 * get_PAA, vo_out_jud_vector, get_beta (vel_obs3D.py:8-66, rvo_inter.py:212-228) */
static void config_vo_circle2(const orc_env *h, const dstate *S, const dstate *O,
                              const double *action_in, vo_inf *out) {
  double action[3] = {action_in[0], action_in[1], action_in[2]};
  if (norm3_blas(action) < 1e-5) action[0] = action[1] = action[2] = 0.0;
  const double x = S->s[0], y = S->s[1], z = S->s[2];
  const double vx = S->s[3], vy = S->s[4], vz = S->s[5], r = S->s[6];
  const double mx = O->s[0], my = O->s[1], mz = O->s[2];
  const double mvx = O->s[3], mvy = O->s[4], mvz = O->s[5], mr = O->s[6];
  double rel[3] = {mx - x, my - y, mz - z};
  double dis = sqrt(sq(rel[1]) + sq(rel[0]) + sq(rel[2]));
  double real_dis = dis;
  int collision = 0;
  mg(dis - (r + mr));
  if (h->env_train) {
    if (dis <= r + mr) { dis = r + mr; collision = 1; }
  } else {                                                 /* rvo_inter.py:144-150 */
    mg(dis - (r - EXP_RADIUS + mr));
    if (dis <= r - EXP_RADIUS + mr) collision = 1;
    if (dis <= r + mr) dis = r + mr;
  }
  out->flag = 0;
  out->collision = collision;
  out->domain = 0;
  out->t = 0.0;
  out->min_dis = dis;
  out->obs[0] = x; out->obs[1] = y; out->obs[2] = z;
  out->obs[3] = rel[0]; out->obs[4] = rel[1]; out->obs[5] = rel[2];
  out->obs[6] = out->obs[7] = out->obs[8] = 0.0;
  if (collision) return;                                   /* rvo_inter.py:152 */
  double dotp = vx * rel[0] + vy * rel[1] + vz * rel[2];
  mg(dotp); /* v == 0 exactly (reset / clamped speed) gives dotp == 0: robust */
  out->obs[7] = out->obs[8] = -1.0;
  if (dotp <= 0) return;                                   /* rvo_inter.py:159 */

  /* get_alpha */
  double ab[3] = {mx - x, my - y, mz - z};
  const double q = (r + mr) / norm3_blas(ab);
  mg(q - 1.0);
  if (q > 1.0) {
    /* math.asin raises ValueError("math domain error") (vel_obs3D.py:13): reachable only
     * with env_train = False, r - 0.2 + mr < dis < r + mr, approaching.  The reference
     * aborts the whole step; here the pair counts as "no VO" and the event is counted
     * (orc_domain_count) - the device sets RVO3D_FLAG_DOMAIN_ERROR at the same pairs. */
    __atomic_add_fetch(&((orc_env *)h)->domain_count, 1, __ATOMIC_RELAXED);
    out->domain = 1;
    return;
  }
  double alpha_raw = asin(q);
  mg_round(alpha_raw, 100.0);
  double alpha = orc_py_round2(alpha_raw);
  /* get_PAA */
  double pr = S->s[7] / (S->s[7] + O->s[7]);
  double paa[3] = {pr * (2 * x + (vx + mvx) * 1), pr * (2 * y + (vy + mvy) * 1),
                   pr * (2 * z + (vz + mvz) * 1)};
  double rvx = 2 * action[0] - mvx - vx;
  double rvy = 2 * action[1] - mvy - vy;
  double rvz = 2 * action[2] - mvz - vz;
  /* vo_out_jud_vector + get_beta */
  double w[3] = {(x + 2 * action[0] * 1) - paa[0], (y + 2 * action[1] * 1) - paa[1],
                 (z + 2 * action[2] * 1) - paa[2]};
  double dp = dot3_blas(rel, w);
  double AB = norm3_blas(rel) * norm3_blas(w);
  double cosang = (AB != 0) ? dp / AB : 0.0;
  mg(fabs(cosang) - 1.0);
  double beta_raw = acos(cosang); /* NaN when |cos| > 1, as np.arccos */
  mg_round(beta_raw, 100.0);
  double beta = np_round2(beta_raw);
  double t = INFINITY;
  int flag = 0;
  if (alpha > beta) { /* inside the cone */
    t = cal_vo_exp_tim(rel[0], rel[1], rel[2], rvx, rvy, rvz, r, mr);
    mg(t - CTIME_THRESHOLD);
    if (t < CTIME_THRESHOLD) flag = 1;
    else t = INFINITY;
  }
  double iet = 1 / (t + 0.2);
  double min_dis = real_dis - mr;
  out->obs[0] = paa[0]; out->obs[1] = paa[1]; out->obs[2] = paa[2];
  out->obs[3] = rel[0]; out->obs[4] = rel[1]; out->obs[5] = rel[2];
  out->obs[6] = alpha; out->obs[7] = min_dis; out->obs[8] = iet;
  out->flag = flag;
  out->t = t;
  out->min_dis = min_dis;
}

/* neighbour gate of rvo_inter.preprocess (rvo_inter.py:90-97) */
static inline int in_gate(const dstate *S, const dstate *O) {
  if (S->s[0] == O->s[0] && S->s[1] == O->s[1] && S->s[2] == O->s[2]) return 0;
  double dif[3] = {S->s[0] - O->s[0], S->s[1] - O->s[1], S->s[2] - O->s[2]};
  double dist = norm3_blas(dif);
  mg(dist - NEIGHBOR_GATE);
  return dist <= NEIGHBOR_GATE;
}

/* building gate (rvo_inter.py:99-105) + check_col_with_budilding (:198-209) */
static int building_collision(const orc_env *h, const dstate *S) {
  int hit = 0;
  for (int b = 0; b < h->nb; ++b) {
    const double *B = h->bld + 4 * b;
    mgx(B[2] - (S->s[2] - 2));
    if (B[2] > S->s[2] - 2) {
      double d2 = norm2_blas(S->s[0] - B[0], S->s[1] - B[1]);
      mg(d2 - BUILDING_GATE);
      if (d2 <= BUILDING_GATE) {
        mgx(S->s[2] - B[2]);
        if (S->s[2] <= B[2]) {
          double dis = sqrt(sq(S->s[0] - B[0]) + sq(S->s[1] - B[1]));
          mg(dis - (S->s[6] + B[3]));
          if (dis <= S->s[6] + B[3]) hit = 1;
        }
      }
    }
  }
  return hit;
}

/* rvo_inter.config_vo_reward (rvo_inter.py:63-83) */
static void config_vo_reward(const orc_env *h, const dstate *all, int i,
                             const double *action, int *vo_flag, double *tmin) {
  *vo_flag = 0;
  *tmin = INFINITY;
  for (int j = 0; j < h->N; ++j) {
    if (j == i || !in_gate(&all[i], &all[j])) continue;
    vo_inf v;
    config_vo_circle2(h, &all[i], &all[j], action, &v);
    if (v.flag) {
      *vo_flag = 1;
      if (v.t < *tmin) *tmin = v.t;
    }
  }
}

/* rvo_inter.config_vo_inf (rvo_inter.py:20-61).  out rows are written in the
 * reference's final order (ascending urgency, most urgent last). */
typedef struct { double iet, md; int j; double obs[9]; } vo_row;

static int row_before(const vo_row *a, const vo_row *b) {
  /* a precedes b in list.sort(reverse=True, key=(-iet, min_dis)) (stable) */
  if (a->iet != b->iet) mg((a->iet - b->iet) / (fabs(a->iet) + 1e-300));
  if (-a->iet != -b->iet) return -a->iet > -b->iet;
  if (a->md != b->md) mg(a->md - b->md);
  if (a->md != b->md) return a->md > b->md;
  return a->j < b->j;
}

static void config_vo_inf(const orc_env *h, const dstate *all, int i,
                          const double *action, vo_row *scratch, double *rows,
                          int *count, int *vo_flag, double *tmin, int *collision) {
  int k = 0;
  *vo_flag = 0;
  *tmin = INFINITY;
  *collision = building_collision(h, &all[i]);
  for (int j = 0; j < h->N; ++j) {
    if (j == i || !in_gate(&all[i], &all[j])) continue;
    vo_inf v;
    config_vo_circle2(h, &all[i], &all[j], action, &v);
    if (v.flag) {
      scratch[k].iet = v.obs[8];
      scratch[k].md = v.obs[7];
      scratch[k].j = j;
      memcpy(scratch[k].obs, v.obs, sizeof v.obs);
      ++k;
      *vo_flag = 1;
      if (v.t < *tmin) *tmin = v.t;
    }
    if (v.collision) *collision = 1;
  }
  /* stable insertion sort into the reference order */
  for (int a = 1; a < k; ++a) {
    vo_row tmp = scratch[a];
    int b = a - 1;
    while (b >= 0 && row_before(&tmp, &scratch[b])) { scratch[b + 1] = scratch[b]; --b; }
    scratch[b + 1] = tmp;
  }
  int keep = k > h->nm ? h->nm : k; /* keep the LAST nm (rvo_inter.py:53-59) */
  int first = k - keep;
  for (int a = 0; a < keep; ++a) memcpy(rows + 9 * a, scratch[first + a].obs, 9 * sizeof(double));
  *count = keep;
}

/* ---- ir_gym rewards ---------------------------------------------------- */
/* ir_gym.rvo_reward_cal (ir_gym.py:64-133) with the second
 * calculate_angle_between_vectors (ir_gym.py:447-473) */
static double rvo_reward_cal(const orc_env *h, const dstate *all, int i,
                             const double *action) {
  int vo_flag; double tmin;
  config_vo_reward(h, all, i, action, &vo_flag, &tmin);
  double des[3] = {np_round3(all[i].s[8]), np_round3(all[i].s[9]), np_round3(all[i].s[10])};
  double vel_penalty = 0.2 * norm3_blas(action) / norm3_blas(des);
  const double eps = 1e-8;
  double magA = sqrt(sq(des[0]) + sq(des[1]) + sq(des[2]) + eps);
  double magB = sqrt(sq(action[0]) + sq(action[1]) + sq(action[2]) + eps);
  double dotp = des[0] * action[0] + des[1] * action[1] + des[2] * action[2];
  double ang;
  if (magA < 1e-6 || magB < 1e-6) ang = 0.0;
  else {
    double c = dotp / (magA * magB);
    c = c < -1.0 + eps ? -1.0 + eps : (c > 1.0 - eps ? 1.0 - eps : c); /* np.clip */
    ang = acos(c);
  }
  mg(ang - M_PI / 18); mg(ang - M_PI / 6); mg(ang - M_PI / 3); mgx(ang - M_PI / 2);
  double angle_punish;
  if (-M_PI / 18 < ang && ang < M_PI / 18) angle_punish = 3;
  else if (-M_PI / 6 < ang && ang < M_PI / 6) angle_punish = 1;
  else if (-M_PI / 3 < ang && ang < M_PI / 3) angle_punish = 0.5;
  else if (-M_PI / 2 < ang && ang < M_PI / 2) angle_punish = 0;
  else angle_punish = -4;
  double safety = 0;
  if (vo_flag) {
    double urgency = 0;
    if (tmin < 2) urgency = -8.0 * exp(-tmin / 0.5);
    safety = -2.5 + urgency;
  }
  mg_round(angle_punish + vel_penalty + safety, 1000.0);
  return np_round3(angle_punish + vel_penalty + safety);
}

/* ir_gym.mov_reward + calculate_penalty_with_exp (ir_gym.py:256-311, 476-490) */
static double mov_reward(int collision, int arrive_reward, int waypoint_num,
                         int n_points, int dest_reward, double deviation,
                         int len_flag, double exlen) {
  if (collision) return -50;
  double reward = 0;
  if (arrive_reward) reward += 3.0 * pow(0.95, (double)(n_points - waypoint_num));
  if (dest_reward) reward += 20.0;
  double d = deviation * 10;
  double dev_penalty = -1.5 * (2 / (1 + exp(-(d - 5) / 0.3)));
  double exlen_penalty = 0;
  if (len_flag) {
    exlen_penalty = -0.3 * log(exlen + 1 + 1e-6);
    mg(exlen_penalty + 6);
    if (exlen_penalty < -6 || exlen_penalty != exlen_penalty) exlen_penalty = -6;
  }
  mg_round(reward + dev_penalty + exlen_penalty, 1000.0);
  return np_round3(reward + dev_penalty + exlen_penalty);
}

static int out_of_map(const orc_env *h, const double *p) { /* drone.py:213-225 */
  for (int k = 0; k < 3; ++k) { mgx(p[k]); mgx(p[k] - h->map[k]); }
  return p[0] < 0 || p[0] > h->map[0] || p[1] < 0 || p[1] > h->map[1] ||
         p[2] < 0 || p[2] > h->map[2];
}

static void write_obs(orc_env *h, const dstate *S, const double *rows, int count,
                      double *obs) {
  int W = 12 + 9 * h->nm, bad = 0;
  for (int k = 0; k < 12; ++k) {
    /* radius, priority and des_vel (already a 3-decimal value: its own rounding
     * is audited in cal_des_vel) are decimals, not libm results: a tie is robust */
    if (k < 6 || k == 11) mg_round0(S->s[k], 100.0);
    obs[k] = np_round2(S->s[k]);
  }
  for (int k = 0; k < 9 * count; ++k) { mg_round0(rows[k], 100.0); obs[12 + k] = np_round2(rows[k]); }
  for (int k = 12 + 9 * count; k < W; ++k) obs[k] = 0.0;
  for (int k = 0; k < W; ++k) if (!isfinite(obs[k])) bad = 1;
  if (bad) {
#ifdef _OPENMP
#pragma omp atomic
#endif
    h->nan_count += 1;
  }
}

/* ---- per-env drivers ---------------------------------------------------- */
typedef struct { dstate *st; vo_row *scratch; double *rows; } work;

static void env_observe(orc_env *h, int e, work *w, double *obs, int32_t *vo_count) {
  /* ir_gym.observation / env_observation (ir_gym.py:334-383): action = 0 */
  const int N = h->N, W = 12 + 9 * h->nm;
  const double zero[3] = {0, 0, 0};
  for (int i = 0; i < N; ++i) dronestate(h, e * N + i, &w->st[i]);
  for (int i = 0; i < N; ++i) {
    int cnt, vf, col; double tmin;
    tl_margin = h->margin + (e * N + i); tl_site = h->msite + (e * N + i);
    /* as in env_step's sweep A: a velocity that is the noise a cancelled speed left behind (1e-17 * direction
     * here, exactly 0 there) decides the sign of v . rel in this observation too - e.g. the re-observation
     * after OTHER drones of the env were reset (found by tools/fuzz_shapes.py seeds 42 / 43, round 3) */
    if (h->vnoise[e * N + i]) mg0(0.0);
    config_vo_inf(h, w->st, i, zero, w->scratch, w->rows, &cnt, &vf, &tmin, &col);
    write_obs(h, &w->st[i], w->rows, cnt, obs + (size_t)(e * N + i) * W);
    vo_count[e * N + i] = cnt;
  }
}

static void env_step(orc_env *h, int e, work *w, const double *actions, double *obs,
                     int32_t *vo_count, double *reward, uint8_t *done,
                     uint8_t *info, uint8_t *finish) {
  const int N = h->N, W = 12 + 9 * h->nm;
  /* sweep A: ir_gym.rvo_reward_list_cal (ir_gym.py:50-62) on pre-move states */
  for (int i = 0; i < N; ++i) dronestate(h, e * N + i, &w->st[i]);
  for (int i = 0; i < N; ++i) {
    tl_margin = h->margin + (e * N + i); tl_site = h->msite + (e * N + i);
    if (h->vnoise[e * N + i]) mg0(0.0); /* pre-move velocity is the noise left by the last step */
    reward[e * N + i] = rvo_reward_cal(h, w->st, i, actions + 3 * (size_t)(e * N + i));
  }
  /* integrate: env_base.drone_step (env_base.py:135-144) */
  for (int i = 0; i < N; ++i) move_forward(h, e * N + i, actions + 3 * (size_t)(e * N + i));
  /* sweep B: ir_gym.obs_move_reward_list / observation_reward (ir_gym.py:136-254) */
  for (int i = 0; i < N; ++i) dronestate(h, e * N + i, &w->st[i]);
  for (int i = 0; i < N; ++i) {
    const int g = e * N + i;
    tl_margin = h->margin + g; tl_site = h->msite + g;
    const double *p = h->p + 3 * (size_t)g;
    int arrive_reward = 0, dest_reward = 0;
    int waypoint_num = h->wp_idx[g];
    int n_points = h->n_points[g] - 1;
    if (arrive(p, wp_at(h, g, h->wp_idx[g])) && !h->arrive[g]) {
      h->arrive[g] = 1;
      arrive_reward = 1;
    }
    if (h->arrive[g]) {
      if (destination_arrive(h, g) && !h->dest[g]) {
        h->dest[g] = 1;
        dest_reward = 1;
      }
    }
    double deviation = w->st[i].s[11];
    double exlen = h->real_len[g] - h->route_len[g] + 4;
    mg(exlen);
    int len_flag = exlen > 0;
    int cnt, vf, col; double tmin;
    config_vo_inf(h, w->st, i, actions + 3 * (size_t)g, w->scratch, w->rows, &cnt,
                  &vf, &tmin, &col);
    if (out_of_map(h, p)) col = 1;
    write_obs(h, &w->st[i], w->rows, cnt, obs + (size_t)g * W);
    vo_count[g] = cnt;
    double mr = mov_reward(col, arrive_reward, waypoint_num, n_points, dest_reward,
                           deviation, len_flag, exlen);
    reward[g] = reward[g] + mr;                            /* mdin.py:28 */
    done[g] = (uint8_t)col;
    info[g] = h->arrive[g];
    finish[g] = h->dest[g];
  }
}

static void reset_drone(orc_env *h, int g) { /* drone.reset, drone.py:270-291 */
  const double *s = wp_at(h, g, 0);
  for (int k = 0; k < 3; ++k) { h->p[3 * (size_t)g + k] = s[k]; h->v[3 * (size_t)g + k] = 0.0; }
  h->wp_idx[g] = 1;
  h->arrive[g] = 0;
  h->dest[g] = 0;
  h->real_len[g] = 0.0;
  h->max_dev[g] = 0.0;
  h->yaw[g] = 0.0;
  h->pitch[g] = 0.0;
  h->vnoise[g] = 0;
  /* extra_len is NOT cleared by the reference's reset */
}

/* ---- public API --------------------------------------------------------- */
orc_env *orc_create(int E, int N, int P, int nb, int nm, int env_train,
                    const double *map_size) {
  if (E < 1 || N < 1 || P < 2 || nb < 0 || nm < 0) return NULL;
  orc_env *h = (orc_env *)calloc(1, sizeof *h);
  size_t G = (size_t)E * N;
  h->E = E; h->N = N; h->P = P; h->nb = nb; h->nm = nm; h->env_train = env_train;
  h->threads = 1;
  memcpy(h->map, map_size, sizeof h->map);
  h->wp = (double *)calloc(G * P * 3, sizeof(double));
  h->n_points = (int32_t *)calloc(G, sizeof(int32_t));
  h->route_len = (double *)calloc(G, sizeof(double));
  h->radius = (double *)calloc(G, sizeof(double));
  h->prio = (double *)calloc(G, sizeof(double));
  h->bld = (double *)calloc((size_t)(nb > 0 ? nb : 1) * 4, sizeof(double));
  h->p = (double *)calloc(G * 3, sizeof(double));
  h->v = (double *)calloc(G * 3, sizeof(double));
  h->yaw = (double *)calloc(G, sizeof(double));
  h->pitch = (double *)calloc(G, sizeof(double));
  h->real_len = (double *)calloc(G, sizeof(double));
  h->max_dev = (double *)calloc(G, sizeof(double));
  h->extra_len = (double *)calloc(G, sizeof(double));
  h->wp_idx = (int32_t *)calloc(G, sizeof(int32_t));
  h->arrive = (uint8_t *)calloc(G, 1);
  h->dest = (uint8_t *)calloc(G, 1);
  h->margin = (double *)calloc(G, sizeof(double));
  h->msite = (int32_t *)calloc(G, sizeof(int32_t));
  h->vnoise = (uint8_t *)calloc(G, 1);
  return h;
}

void orc_destroy(orc_env *h) {
  if (!h) return;
  free(h->wp); free(h->n_points); free(h->route_len); free(h->radius); free(h->prio);
  free(h->bld); free(h->p); free(h->v); free(h->yaw); free(h->pitch);
  free(h->real_len); free(h->max_dev); free(h->extra_len); free(h->wp_idx);
  free(h->arrive); free(h->dest); free(h->margin); free(h->msite); free(h->vnoise); free(h);
}

void orc_load_world(orc_env *h, const double *waypoints, const int32_t *n_points,
                    const double *buildings, const double *radius,
                    const double *priority) {
  size_t G = (size_t)h->E * h->N;
  memcpy(h->wp, waypoints, G * h->P * 3 * sizeof(double));
  memcpy(h->n_points, n_points, G * sizeof(int32_t));
  if (h->nb > 0) memcpy(h->bld, buildings, (size_t)h->nb * 4 * sizeof(double));
  for (size_t g = 0; g < G; ++g) {
    h->radius[g] = radius ? radius[g] : 0.2;
    h->prio[g] = priority ? priority[g] : 5.0;
    /* drone.calculate_total_length (drone.py:409-429) */
    double total = 0.0;
    for (int k = 0; k + 1 < h->n_points[g]; ++k) {
      const double *a = wp_at(h, (int)g, k), *b = wp_at(h, (int)g, k + 1);
      total += sqrt(sq(b[0] - a[0]) + sq(b[1] - a[1]) + sq(b[2] - a[2]));
    }
    h->route_len[g] = total;
    h->extra_len[g] = 0.0;
    reset_drone(h, (int)g);
  }
}

void orc_reset(orc_env *h, const uint8_t *env_mask) {
  for (int e = 0; e < h->E; ++e)
    if (!env_mask || env_mask[e])
      for (int i = 0; i < h->N; ++i) reset_drone(h, e * h->N + i);
}

void orc_reset_drones(orc_env *h, const uint8_t *mask) {
  for (int g = 0; g < h->E * h->N; ++g)
    if (mask[g]) reset_drone(h, g);
}

static work *work_alloc(const orc_env *h) {
  work *w = (work *)malloc(sizeof *w);
  w->st = (dstate *)malloc(sizeof(dstate) * h->N);
  w->scratch = (vo_row *)malloc(sizeof(vo_row) * h->N);
  w->rows = (double *)malloc(sizeof(double) * 9 * (h->nm > 0 ? h->nm : 1));
  return w;
}
static void work_free(work *w) { free(w->st); free(w->scratch); free(w->rows); free(w); }

static void margin_clear(orc_env *h) {
  for (int g = 0; g < h->E * h->N; ++g) h->margin[g] = INFINITY;
}

void orc_get_margin(const orc_env *h, double *out, int32_t *site) {
  if (out) memcpy(out, h->margin, (size_t)h->E * h->N * sizeof(double));
  if (site) memcpy(site, h->msite, (size_t)h->E * h->N * sizeof(int32_t));
}

void orc_observe(orc_env *h, double *obs, int32_t *vo_count) {
  margin_clear(h);
#ifdef _OPENMP
#pragma omp parallel num_threads(h->threads)
#endif
  {
    work *w = work_alloc(h);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (int e = 0; e < h->E; ++e) env_observe(h, e, w, obs, vo_count);
    work_free(w);
  }
}

void orc_step(orc_env *h, const double *actions, double *obs, int32_t *vo_count,
              double *reward, uint8_t *done, uint8_t *info, uint8_t *finish) {
  margin_clear(h);
#ifdef _OPENMP
#pragma omp parallel num_threads(h->threads)
#endif
  {
    work *w = work_alloc(h);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (int e = 0; e < h->E; ++e)
      env_step(h, e, w, actions, obs, vo_count, reward, done, info, finish);
    work_free(w);
  }
}

void orc_step_autoreset(orc_env *h, const double *actions, double *obs,
                        int32_t *vo_count, double *reward, uint8_t *done,
                        uint8_t *info, uint8_t *finish, uint8_t *reset_mask) {
  margin_clear(h);
#ifdef _OPENMP
#pragma omp parallel num_threads(h->threads)
#endif
  {
    work *w = work_alloc(h);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (int e = 0; e < h->E; ++e) {
      env_step(h, e, w, actions, obs, vo_count, reward, done, info, finish);
      int any = 0;
      for (int i = 0; i < h->N; ++i) {
        int g = e * h->N + i;
        int r = done[g] || finish[g];
        if (reset_mask) reset_mask[g] = (uint8_t)r;
        if (r) { reset_drone(h, g); any = 1; }
      }
      if (any) env_observe(h, e, w, obs, vo_count);
    }
    work_free(w);
  }
}

void orc_get_state(const orc_env *h, double *pos, double *vel, double *yaw,
                   double *pitch, double *real_len, double *max_dev,
                   double *extra_len, int32_t *wp_idx, uint8_t *arrive_f,
                   uint8_t *dest) {
  size_t G = (size_t)h->E * h->N;
  if (pos) memcpy(pos, h->p, G * 3 * sizeof(double));
  if (vel) memcpy(vel, h->v, G * 3 * sizeof(double));
  if (yaw) memcpy(yaw, h->yaw, G * sizeof(double));
  if (pitch) memcpy(pitch, h->pitch, G * sizeof(double));
  if (real_len) memcpy(real_len, h->real_len, G * sizeof(double));
  if (max_dev) memcpy(max_dev, h->max_dev, G * sizeof(double));
  if (extra_len) memcpy(extra_len, h->extra_len, G * sizeof(double));
  if (wp_idx) memcpy(wp_idx, h->wp_idx, G * sizeof(int32_t));
  if (arrive_f) memcpy(arrive_f, h->arrive, G);
  if (dest) memcpy(dest, h->dest, G);
}

void orc_set_state(orc_env *h, const double *pos, const double *vel,
                   const double *yaw, const double *pitch, const double *real_len,
                   const double *max_dev, const double *extra_len,
                   const int32_t *wp_idx, const uint8_t *arrive_f,
                   const uint8_t *dest) {
  size_t G = (size_t)h->E * h->N;
  if (pos) memcpy(h->p, pos, G * 3 * sizeof(double));
  if (vel) memcpy(h->v, vel, G * 3 * sizeof(double));
  if (yaw) memcpy(h->yaw, yaw, G * sizeof(double));
  if (pitch) memcpy(h->pitch, pitch, G * sizeof(double));
  if (real_len) memcpy(h->real_len, real_len, G * sizeof(double));
  if (max_dev) memcpy(h->max_dev, max_dev, G * sizeof(double));
  if (extra_len) memcpy(h->extra_len, extra_len, G * sizeof(double));
  if (wp_idx) memcpy(h->wp_idx, wp_idx, G * sizeof(int32_t));
  if (arrive_f) memcpy(h->arrive, arrive_f, G);
  if (dest) memcpy(h->dest, dest, G);
}

void orc_des_vel(const orc_env *h, double *des_vel) {
  tl_margin = 0; tl_site = 0;  /* no audit here: never a stale pointer of an earlier env */
  for (int g = 0; g < h->E * h->N; ++g)
    cal_des_vel(h->p + 3 * (size_t)g, wp_at(h, g, h->wp_idx[g]), des_vel + 3 * (size_t)g);
}

/* ---- classical RVO velocity selection ---------------------------------------
 * uaisa_env/vel_obs/reciprocal_vel_obs.py:19-166 (cal_vel -> preprocess, config_vo,
 * vel_candidate, vo_out2, vel_select, penalty).  The reference class cannot run (list
 * attribute assignment at :109, slices [0:4] / [4:6] at :63-69 / :105, a missing
 * return at :119-124, cal_vel's config_vo call at :24); this is the algorithm the code
 * spells out, built from the helpers it calls (vel_obs3D.get_alpha / get_PAA /
 * get_rvo_array / get_beta / cal_exp_tim, which do run).  PARITY UNPINNED for the
 * driver loop; the helpers are pinned by tests/golden/rvo_vel.npz (gen_golden_rvo.py).
 * Choices where the reference raises: asin domain (overlap) -> alpha = 1.57; no
 * candidate at all -> velocity 0. */
static double cal_exp_tim_classic(const double *pa, const double *pb, const double *va,
                                  const double *vb, double ra, double rb) {
  /* vel_obs3D.cal_exp_tim (vel_obs3D.py:104-143) */
  double rx = pa[0] - pb[0], ry = pa[1] - pb[1], rz = pa[2] - pb[2];
  double vx = va[0] - vb[0], vy = va[1] - vb[1], vz = va[2] - vb[2];
  double r = ra + rb;
  double a = sq(vx) + sq(vy) + sq(vz);
  double b = 2 * rx * vx + 2 * ry * vy + 2 * rz * vz;
  double c = sq(rx) + sq(ry) + sq(rz) - sq(r);
  if (c <= 0) return 0.0;
  double temp = sq(b) - 4 * a * c;
  if (temp <= 0) return INFINITY;
  double t1 = (-b + sqrt(temp)) / (2 * a);
  double t2 = (-b - sqrt(temp)) / (2 * a);
  double t3 = t1 >= 0 ? t1 : INFINITY, t4 = t2 >= 0 ? t2 : INFINITY;
  return t3 < t4 ? t3 : t4;  /* min(t3, t4) */
}

static double get_beta_classic(const double *A, const double *B) { /* vel_obs3D.py:44-66 */
  double dotp = dot3_blas(A, B);
  double AB = norm3_blas(A) * norm3_blas(B);
  double c = (AB != 0) ? dotp / AB : 0.0;
  double ang = acos(c);
  if (ang > M_PI) ang -= 2 * M_PI;  /* wraptopi */
  if (ang < -M_PI) ang += 2 * M_PI;
  return np_round2(ang);            /* round(np.float64, 2) */
}

/* np.arange(lo, hi, 0.5): value k (numpy fills start + k * ((start + step) - start)) */
static int arange_len(double lo, double hi) {
  double n = ceil((hi - lo) / 0.5);
  return n > 0 ? (int)n : 0;
}
static double arange_at(double lo, int k) {
  if (k == 0) return lo;
  double next = lo + 0.5;
  if (k == 1) return next;
  return lo + k * (next - lo);
}

/* ---- the parts of reciprocal_vel_obs that DO run on an instance, one function each; orc_rvo_vel
 * below is built from exactly these, and tests/golden/rvo_calls.npz (oracle/gen_golden_rvo.py) holds
 * what the reference's own methods return for the same arguments. ---- */

/* reciprocal_vel_obs.distance (reciprocal_vel_obs.py:149-151) */
double orc_rvo_distance(const double *p1, const double *p2) {
  return sqrt(sq(p2[0] - p1[0]) + sq(p2[1] - p1[1]) + sq(p2[2] - p1[2]));
}

/* reciprocal_vel_obs.preprocess (:32-54): neighbours with np.linalg.norm(agent - drone) <= 10 (no
 * self-exclusion: the caller passes the others), buildings [x, y, h, r] with h > z - 1 and
 * np.linalg.norm(xy - bxy) <= 10 (this class's own gate: not rvo_inter's z - 2 / 5 m).  The
 * building list is computed and then dropped by cal_vel (:22-25): it never reaches a velocity. */
void orc_rvo_preprocess(const double *agent3, const double *drones /* [n][3] */, int n,
                        const double *blds /* [nb][4] */, int nb, uint8_t *keep_drone,
                        uint8_t *keep_bld) {
  for (int j = 0; j < n; ++j) {
    const double *pb = drones + 3 * (size_t)j;
    double dif[3] = {agent3[0] - pb[0], agent3[1] - pb[1], agent3[2] - pb[2]};
    keep_drone[j] = norm3_blas(dif) <= 10;
  }
  for (int b = 0; b < nb; ++b) {
    const double *B = blds + 4 * (size_t)b;
    keep_bld[b] = (B[2] > agent3[2] - 1) && (norm2_blas(agent3[0] - B[0], agent3[1] - B[1]) <= 10);
  }
}

/* min(tc_list) of reciprocal_vel_obs.penalty (:126-143) over the records odro8 [n][8] =
 * [p(3), v(3), r, prio]; n >= 1 (the reference's min() of an empty list raises). */
static double rvo_tc_min(const double *agent8, const double *odro8, int n) {
  double tc_min = INFINITY;
  for (int j = 0; j < n; ++j) {
    const double *o = odro8 + 8 * (size_t)j;
    double tc = cal_exp_tim_classic(agent8, o, agent8 + 3, o + 3, agent8[6], o[6]);
    if (j == 0 || tc < tc_min) tc_min = tc;  /* Python min: the first of equal values */
  }
  return tc_min;
}
/* reciprocal_vel_obs.penalty (:126-147): factor * 1 / tc_min (inf when tc_min == 0) + distance(vel_des, vel) */
double orc_rvo_penalty(const double *vel, const double *vel_des, const double *agent8,
                       const double *odro8, int n, double factor) {
  double tc_min = rvo_tc_min(agent8, odro8, n);
  double tc_inv = (tc_min == 0) ? INFINITY : 1 / tc_min;
  return factor * tc_inv + orc_rvo_distance(vel_des, vel);
}

/* The candidate grid of reciprocal_vel_obs.vel_candidate (:85-101): np.arange over
 * np.clip([v - acceler, v + acceler], -vmax, vmax) per axis in steps of 0.5, |v| >= 0.3 (with an
 * empty VO list the method runs and returns exactly this list as vo_outside).  Returns the count;
 * out [cap][3] in the method's order (x outermost). */
#define ORC_RVO_MAXC 512
int orc_rvo_candidates(const double *vel3, const double *vmax, double acceler, double *out, int cap) {
  double lo[3], hi[3];
  int cnt[3], n = 0;
  for (int k = 0; k < 3; ++k) {
    lo[k] = clampd(vel3[k] - acceler, -vmax[k], vmax[k]);  /* np.clip */
    hi[k] = clampd(vel3[k] + acceler, -vmax[k], vmax[k]);
    cnt[k] = arange_len(lo[k], hi[k]);
  }
  for (int ix = 0; ix < cnt[0] && ix < ORC_RVO_MAXC; ++ix)
    for (int iy = 0; iy < cnt[1] && iy < ORC_RVO_MAXC; ++iy)
      for (int iz = 0; iz < cnt[2] && iz < ORC_RVO_MAXC; ++iz) {
        const double v[3] = {arange_at(lo[0], ix), arange_at(lo[1], iy), arange_at(lo[2], iz)};
        if (sqrt(sq(v[0]) + sq(v[1]) + sq(v[2])) < 0.3) continue;
        if (n < cap) memcpy(out + 3 * (size_t)n, v, sizeof v);
        ++n;
      }
  return n;
}

/* reciprocal_vel_obs.vel_select (:119-124) when vo_outside is empty - the one branch of it that
 * returns: min(vo_inside, key=penalty), the first of equal minima.  Returns the index, -1 if n_in == 0. */
int orc_rvo_select_inside(const double *inside /* [n_in][3] */, int n_in, const double *agent11,
                          const double *odro8, int n) {
  int best = -1;
  double bp = 0;
  for (int k = 0; k < n_in; ++k) {
    double p = orc_rvo_penalty(inside + 3 * (size_t)k, agent11 + 8, agent11, odro8, n, 1);
    if (best < 0 || p < bp) { best = k; bp = p; }
  }
  return best;
}

/* cal_vel (:19-31) for every drone.  Built from the functions above; what no running reference
 * method covers - config_vo (:63-83 raises: state[0:4] has four values for get_PAA's three),
 * vo_out2 (:103-117 raises: list.append assignment) and the vo_outside branch of vel_select
 * (returns None) - is the algorithm those lines spell out on the reference's helper functions
 * (get_alpha / get_PAA / get_rvo_array / get_beta, pinned by tests/golden/rvo_vel.npz).
 * Margin audit: the two roundings that rest on libm results (alpha = round(asin), beta =
 * round(acos)) report their distance to a tie per agent, as the step's decisions do. */
void orc_rvo_vel(const orc_env *h, const double *vmax, double acceler, double *out) {
  const int N = h->N;
  for (int g = 0; g < h->E * N; ++g) h->margin[g] = INFINITY;
#pragma omp parallel for schedule(static) num_threads(h->threads)
  for (int g = 0; g < h->E * N; ++g) {
    tl_margin = 0; tl_site = 0;
    const int e = g / N;
    const double *pa = h->p + 3 * (size_t)g, *va = h->v + 3 * (size_t)g;
    const double ra = h->radius[g], pra = h->prio[g];
    double agent[11] = {pa[0], pa[1], pa[2], va[0], va[1], va[2], ra, pra, 0, 0, 0};
    cal_des_vel(pa, wp_at(h, g, h->wp_idx[g]), agent + 8);
    const double *des = agent + 8;
    /* preprocess: the other drones of the env within 10 m, as 8-value records */
    double *others = (double *)malloc(sizeof(double) * 3 * (size_t)N);
    uint8_t *keep = (uint8_t *)malloc((size_t)N);
    double *odro = (double *)malloc(sizeof(double) * 8 * (size_t)N);
    int no = 0;
    for (int j = 0; j < N; ++j) memcpy(others + 3 * (size_t)j, h->p + 3 * (size_t)(e * N + j), 3 * sizeof(double));
    orc_rvo_preprocess(pa, others, N, 0, 0, keep, 0);
    for (int j = 0; j < N; ++j) {
      const int gj = e * N + j;
      if (gj == g || !keep[j]) continue;
      double *o = odro + 8 * (size_t)no++;
      memcpy(o, h->p + 3 * (size_t)gj, 3 * sizeof(double));
      memcpy(o + 3, h->v + 3 * (size_t)gj, 3 * sizeof(double));
      o[6] = h->radius[gj]; o[7] = h->prio[gj];
    }
    /* config_vo per neighbour (as intended): alpha, PAA, rel */
    double *vo = (double *)malloc(sizeof(double) * 7 * (size_t)(no > 0 ? no : 1));
    tl_margin = h->margin + g; tl_site = h->msite + g;
    for (int j = 0; j < no; ++j) {
      const double *o = odro + 8 * (size_t)j;
      double ab[3] = {o[0] - pa[0], o[1] - pa[1], o[2] - pa[2]};
      double q = (ra + o[6]) / norm3_blas(ab);
      double alpha = 1.57;
      if (q <= 1.0) { double as = asin(q); mg_round0(as, 100.0); alpha = orc_py_round2(as); }  /* get_alpha */
      double pr = pra / (pra + o[7]);                                                          /* get_PAA */
      double *w = vo + 7 * (size_t)j;
      w[0] = pr * (2 * pa[0] + (va[0] + o[3]) * 1);
      w[1] = pr * (2 * pa[1] + (va[1] + o[4]) * 1);
      w[2] = pr * (2 * pa[2] + (va[2] + o[5]) * 1);
      w[3] = ab[0]; w[4] = ab[1]; w[5] = ab[2]; w[6] = alpha;
    }
    double cand[3 * 125];
    int nc = orc_rvo_candidates(va, vmax, acceler, cand, 125);
    if (nc > 125) nc = 125;  /* acceler <= 1: at most 5 per axis */
    double best_out = INFINITY;
    int sel_out = -1, n_in = 0;
    double inside[3 * 125];
    for (int c = 0; c < nc; ++c) {
      const double *v = cand + 3 * (size_t)c;
      const double pn[3] = {pa[0] + v[0] * 1, pa[1] + v[1] * 1, pa[2] + v[2] * 1};
      int in = 0;
      for (int j = 0; j < no; ++j) {  /* vo_out2 over the VO list */
        const double *w = vo + 7 * (size_t)j;
        double d3[3] = {pn[0] - w[0], pn[1] - w[1], pn[2] - w[2]};
        double dotp = dot3_blas(w + 3, d3);
        double AB = norm3_blas(w + 3) * norm3_blas(d3);
        double cs = (AB != 0) ? dotp / AB : 0.0;
        double ang = acos(cs);
        mg_round0(ang, 100.0);
        if (w[6] > get_beta_classic(w + 3, d3)) in = 1;
      }
      if (!in) {  /* min(vo_outside, key=distance(v, vel_des)): the first minimum */
        const double dd = orc_rvo_distance(v, des);
        if (sel_out < 0 || dd < best_out) { best_out = dd; sel_out = c; }
      } else {
        memcpy(inside + 3 * (size_t)n_in++, v, 3 * sizeof(double));
      }
    }
    tl_margin = 0; tl_site = 0;
    double *o = out + 3 * (size_t)g;
    if (sel_out >= 0) memcpy(o, cand + 3 * (size_t)sel_out, 3 * sizeof(double));
    else if (n_in > 0) memcpy(o, inside + 3 * (size_t)orc_rvo_select_inside(inside, n_in, agent, odro, no), 3 * sizeof(double));
    else o[0] = o[1] = o[2] = 0.0;
    free(others); free(keep); free(odro); free(vo);
  }
}

void orc_vo_inf(orc_env *h, int e, int i, const double *action, double *rows,
                int32_t *count, int32_t *vo_flag, double *tmin, int32_t *collision) {
  work *w = work_alloc(h);
  margin_clear(h);
  for (int k = 0; k < h->N; ++k) dronestate(h, e * h->N + k, &w->st[k]);
  tl_margin = h->margin + (e * h->N + i); tl_site = h->msite + (e * h->N + i);
  int c, f, col;
  config_vo_inf(h, w->st, i, action, w->scratch, w->rows, &c, &f, tmin, &col);
  memcpy(rows, w->rows, sizeof(double) * 9 * c);
  *count = c; *vo_flag = f; *collision = col;
  work_free(w);
}

/* Call-level check of rvo_inter.config_vo_circle2 (rvo_inter.py:116-196): one pair, the
 * 8-value records of self / other, either env_train mode. */
void orc_vo_circle2(int env_train, const double *self8, const double *other8,
                    const double *action, double *obs9, int32_t *vo_flag, double *exp_time,
                    int32_t *collision, double *min_dis, int32_t *domain_error) {
  orc_env h;
  memset(&h, 0, sizeof h);
  h.env_train = env_train;
  dstate S, O;
  memset(&S, 0, sizeof S); memset(&O, 0, sizeof O);
  memcpy(S.s, self8, 8 * sizeof(double));
  memcpy(O.s, other8, 8 * sizeof(double));
  tl_margin = 0; tl_site = 0;
  vo_inf v;
  config_vo_circle2(&h, &S, &O, action, &v);
  memcpy(obs9, v.obs, sizeof v.obs);
  *vo_flag = v.flag; *exp_time = v.t; *collision = v.collision; *min_dis = v.min_dis;
  *domain_error = v.domain;
}

int64_t orc_nan_count(const orc_env *h) { return h->nan_count; }
int64_t orc_domain_count(const orc_env *h) { return h->domain_count; }
void orc_set_threads(orc_env *h, int n) { h->threads = n < 1 ? 1 : n; }
