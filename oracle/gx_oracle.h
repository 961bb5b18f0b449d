/*
 * gx_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, fp32) of the guardX `safe_rl_envs` Engine hot path
 * for the Goal_<Robot>_<N>Hazards family (Point, Swimmer, Ant, Walker).  It exists to CHECK the HIP path; it
 * is never shipped, never imported by guardx_amd/, and only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * PARITY UNPINNED: the reference cannot be imported in the build container
 * (gym/jax/mujoco/mjx absent) and ships no tests or golden
 * vectors for this path.  What IS pinned: the threefry2x32 block function and
 * jax.random.split ordering (published JAX known answers, see
 * tests/test_oracle_prng.py); everything else follows the reference source
 * lines cited below plus published MuJoCo/MJX semantics ([derived]).
 *
 * Citations are relative to /root/reference/safe_rl_envs/safe_rl_envs/envs/.
 */
#ifndef GX_ORACLE_H
#define GX_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Field-for-field the same as include/guardx.h:gx_config (kept separate so the
 * checker does not include product headers). */
typedef struct gxo_config {
    int32_t struct_size;        /* sizeof(gxo_config), ABI check */
    int32_t robot;              /* 0 = xmls/point.xml, 1 = xmls/swimmer.xml, 2 = xmls/ant.xml, 3 = xmls/walker.xml,
                                 * 4 = xmls/point.xml with the round-1 actuator reading (no class defaults) */
    int32_t env_num;            /* envs owned by this instance */
    int32_t env_total;          /* env_num of the whole (possibly sharded) batch */
    int32_t env_offset;         /* global index of local env 0 */
    uint32_t seed;              /* engine.py:216  PRNGKey(_seed) */
    int32_t num_steps;          /* engine.py:99 */
    int32_t hazards_num;        /* engine.py:195 */
    int32_t lidar_num_bins;     /* engine.py:148 */
    int32_t lidar_alias;        /* engine.py:153 */
    int32_t lidar_max_dist_set; /* engine.py:150  (0 == None) */
    float lidar_max_dist;
    float lidar_exp_gain;       /* engine.py:151 */
    float goal_size;            /* engine.py:167 */
    float hazards_size;         /* engine.py:199 */
    float reward_distance;      /* engine.py:174 */
    double goal_keepout;        /* engine.py:166 (python floats: kept double) */
    double hazards_keepout;     /* engine.py:198 */
    double robot_keepout;       /* engine.py:112 */
    double placements_margin;   /* engine.py:104 */
    double extents[4];          /* engine.py:103 xmin,ymin,xmax,ymax */
    int32_t observe_goal_lidar; /* engine.py:119 */
    int32_t observe_goal_comp;  /* engine.py:120 */
    int32_t observe_hazards;    /* engine.py:121 */
    int32_t observe_qpos;       /* engine.py:123 */
    int32_t observe_qvel;       /* engine.py:124 */
    int32_t observe_ctrl;       /* engine.py:128 */
    int32_t observe_vel;        /* engine.py:126 */
    int32_t observe_acc;        /* engine.py:127 */
    int32_t n_candidates;       /* engine.py:263  int(1e6) */
    int32_t physics_steps;      /* engine.py:202 */
    float robot_goal_min_dist;  /* engine.py:571  3.0 */
    int32_t reserved;
    const double* placements;   /* NULL or (H+PL+2) x 4 doubles (goal, hazards, pillars, robot), engine.py:507-531 */
    /* ---- synthetic extension, NO reference counterpart (BASELINE config 5): a second class
     * of static circles, "pillars", in the hazard style -- placed by the sampler after the hazards (own
     * keepout), seen by a lidar of their own ('pillars_lidar', sorted between 'hazards_lidar' and 'qpos'),
     * and adding sum(pillars_size - min(dist, pillars_size)) to the cost after the hazard terms.  The
     * reference only carries the colour / lidar group constants (engine.py:38,56). */
    int32_t pillars_num;        /* 0 = the reference's task */
    int32_t observe_pillars;
    float pillars_size;
    float robot_rot;   /* engine.py:114,342-345 -> world.py:117; 0 = None */
    double pillars_keepout;
} gxo_config;

typedef struct gxo_env gxo_env;

enum { GXO_OK = 0, GXO_ERR_ARG = 1, GXO_ERR_UNSUPPORTED = 2, GXO_ERR_LAYOUT = 3 };

int  gxo_create(const gxo_config* cfg, gxo_env** out);
void gxo_destroy(gxo_env* e);
int  gxo_obs_dim(const gxo_env* e);
/* robot.nq / nv / nu (world.py:435-438) and the action width */
void gxo_dims(const gxo_env* e, int32_t* nq, int32_t* nv, int32_t* nu, int32_t* na);
/* Engine.reset  engine.py:454-467 */
int  gxo_reset(gxo_env* e, float* obs);
/* Engine.step   engine.py:469-495 ; action[N*na]; qacc[N*nv] may be NULL */
int  gxo_step(gxo_env* e, const float* action, float* obs, float* reward,
              float* cost, float* done, float* qacc);
/* Engine.reset_done engine.py:497-505 */
int  gxo_reset_done(gxo_env* e, float* obs);
int  gxo_layout_size(const gxo_env* e);

/* Flat env-major state exchange used by the parity tests.
 *  qpos[N*nq] qvel[N*nv] pose0[N*4]=(x,y,cos,sin of _data.xpos/xmat) pose1[N*2]
 *  objs[N*(1+H)*2]=(goal, hazard0..) done0[N] done1[N] steps[N] key[2] hist[1] */
int  gxo_get_state(const gxo_env* e, float* qpos, float* qvel, float* pose0,
                   float* pose1, float* objs, float* done0, float* done1,
                   float* steps, uint32_t* key, int32_t* hist);
int  gxo_set_state(gxo_env* e, const float* qpos, const float* qvel,
                   const float* pose0, const float* pose1, const float* objs,
                   const float* done0, const float* done1, const float* steps,
                   const uint32_t* key, const int32_t* hist);
/* valid-layout pool of the last reset(): rows of (1+H+1)*2 floats
 * (goal, hazard0.., robot), at most `max_rows` rows are copied. */
int  gxo_get_pool(const gxo_env* e, float* pool, int32_t max_rows);

/* probes */
void  gxo_threefry2x32(uint32_t k0, uint32_t k1, uint32_t x0, uint32_t x1, uint32_t* out2);
void  gxo_split(const uint32_t* key, int32_t n, uint32_t* out_2n);
float gxo_uniform(const uint32_t* key, float minval, float maxval);
void  gxo_randint(const uint32_t* key, int32_t n, uint32_t span, int32_t* out_n);
void  gxo_math_probe(int32_t n, const float* x, const float* y, float* s, float* c,
                     float* at2, float* ex);
void  gxo_math_probe2(int32_t n, const float* x, float* lg, float* th);
/* T x (ac.step -> env.step -> reset_done) with MLPActorCritic((64,64), tanh) weights
 * (trpo_core.py:110-173) and a counter-based N(0,1) stream; time-major outputs. */
int   gxo_rollout_policy(gxo_env* e, int32_t T, int32_t hidden, const float* params, const uint32_t* seed,
                         uint32_t t0, const float* obs0, float* obs_in, float* act, float* logp, float* val,
                         float* mu, float* rew, float* cost, float* done, float* obs_last, float* val_last,
                         float* logstd);
/* one ant.xml mjx.step (test probe): dbg = 121 dense mass matrix + 11 smooth force, qpos coordinates */
void  gxo_ant_probe(const float* q, const float* v, const float* ctrl, float* q2, float* v2, float* qacc,
                    float* pose, float* dbg);
/* one walker.xml mjx.step (test probe): dbg = 169 dense mass matrix + 13 smooth force */
void  gxo_walker_probe(const float* q, const float* v, const float* ctrl, float* q2, float* v2, float* qacc,
                       float* pose, float* dbg);
void  gxo_set_threads(int32_t n);
int   gxo_get_threads(void);

#ifdef __cplusplus
}
#endif
#endif
