/*
 * gx_oracle.c -- TEST INFRASTRUCTURE ONLY (see gx_oracle.h).
 *
 * CPU restatement of guardX safe_rl_envs Engine (reset / step / reset_done)
 * for the Goal_Point_<N>Hazards family, fp32, one plain loop per env.
 * "engine.py:NNN" cites /root/reference/safe_rl_envs/safe_rl_envs/envs/engine.py.
 * "[derived]" marks MuJoCo/MJX/JAX semantics taken from their published
 * algorithms (third-party, unpinned in requirements.txt:3-4, absent here).
 *
 * PARITY UNPINNED except for the PRNG block function / split ordering.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: every fp32 operation
 * below is a single IEEE operation unless written as fmaf()).
 */
#include "gx_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ */
/* deterministic fp32 math (coefficients: tools/fit_math.py)           */
/* ------------------------------------------------------------------ */
static const float TWO_OVER_PI = 0.6366197466850281f;
static const float PIO2_HI = 1.5707963705062866f;
static const float PIO2_MID = -4.371138828673793e-08f;
static const float PIO2_LO = -1.7151245100058819e-15f;
static const float PI_F = 3.1415927410125732f;
static const float TWO_PI_F = 6.2831854820251465f; /* f32(2*pi) */
static const float LOG2E = 1.4426950216293335f;
static const float LN2_HI = 0.6931471824645996f;
static const float LN2_LO = -1.9046542121259336e-09f;

static float bits_to_f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t f_to_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* sin/cos by 3-term Cody-Waite reduction to [-pi/4,pi/4] + minimax polys. */
static void gx_sincos(float x, float* s, float* c)
{
    if (!(fabsf(x) <= 16777216.0f)) x = x * 0.0f; /* inf/nan -> nan, huge -> 0 */
    float k = rintf(x * TWO_OVER_PI);
    float r = fmaf(-k, PIO2_HI, x);
    r = fmaf(-k, PIO2_MID, r);
    r = fmaf(-k, PIO2_LO, r);
    float z = r * r;
    float ps = fmaf(z, -0.00019488747f, 0.008331924f);
    ps = fmaf(z, ps, -0.1666665f);
    float S = fmaf(r * z, ps, r);
    float pc = fmaf(z, 2.4431205e-05f, -0.0013887306f);
    pc = fmaf(z, pc, 0.041666646f);
    float C = fmaf(z * z, pc, fmaf(z, -0.5f, 1.0f));
    int q = ((int)k) & 3; /* |k| <= 2^24*0.64: exact */
    float ss = (q & 1) ? C : S;
    float cc = (q & 1) ? S : C;
    if (q == 1 || q == 2) cc = -cc;
    if (q >= 2) ss = -ss;
    if (x != x) { ss = x; cc = x; }
    *s = ss;
    *c = cc;
}

/* atan2 with one division and a degree-7 (in a^2) minimax polynomial. */
static float gx_atan2(float y, float x)
{
    if (x != x || y != y) return x + y;
    float ax = fabsf(x), ay = fabsf(y);
    float mx = ax > ay ? ax : ay;
    float mn = ax > ay ? ay : ax;
    float a;
    if (mx == 0.0f) a = 0.0f;
    else if (mx == INFINITY) a = (mn == INFINITY) ? 1.0f : 0.0f;
    else a = mn / mx;
    float z = a * a;
    float p = fmaf(z, 0.0026222442f, -0.015132535f);
    p = fmaf(z, p, 0.04112186f);
    p = fmaf(z, p, -0.07366706f);
    p = fmaf(z, p, 0.10573932f);
    p = fmaf(z, p, -0.14185975f);
    p = fmaf(z, p, 0.19990396f);
    p = fmaf(z, p, -0.33332986f);
    float t = fmaf(a * z, p, a);
    if (ay > ax) t = PIO2_HI - t;
    if (f_to_bits(x) >> 31) t = PI_F - t;
    return (f_to_bits(y) >> 31) ? -t : t;
}

/* exp for the lidar sensor.  Values below exp(-87) (< 1.7e-38) flush to 0. */
static float gx_exp(float x)
{
    if (x != x) return x;
    if (x < -87.0f) return 0.0f;
    if (x > 88.0f) return INFINITY;
    float k = rintf(x * LOG2E);
    float r = fmaf(-k, LN2_HI, x);
    r = fmaf(-k, LN2_LO, r);
    float q = fmaf(r, 0.001395172f, 0.008369599f);
    q = fmaf(r, q, 0.041666187f);
    q = fmaf(r, q, 0.16666512f);
    q = fmaf(r, q, 0.5f);
    float t = fmaf(r * r, q, r);
    float e = 1.0f + t;
    int ki = (int)k;
    return bits_to_f(f_to_bits(e) + ((uint32_t)ki << 23));
}

/* natural log for x > 0 (normal floats): exponent split + atanh series, ~1.4e-7 relative */
static float gx_log(float x)
{
    uint32_t b = f_to_bits(x);
    int e = (int)(b >> 23) - 127;
    float m = bits_to_f((b & 0x7FFFFFu) | 0x3F800000u);
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
    const float f = m - 1.0f;
    const float s = f / (2.0f + f);
    const float z = s * s;
    float P = fmaf(z, 0.22222222f, 0.2857143f);
    P = fmaf(z, P, 0.4f);
    P = fmaf(z, P, 0.6666667f);
    const float lnm = fmaf(s * z, P, 2.0f * s);
    const float fe = (float)e;
    return fmaf(fe, LN2_HI, fmaf(fe, LN2_LO, lnm));
}

/* tanh through exp: sign(x) * (1 - 2 / (exp(2|x|) + 1)), saturating at |x| > 9 */
static float gx_tanh(float x)
{
    if (x != x) return x;
    const float ax = fabsf(x);
    float t = 1.0f;
    if (ax <= 9.0f) t = 1.0f - 2.0f / (gx_exp(2.0f * ax) + 1.0f);
    return (f_to_bits(x) >> 31) ? -t : t;
}

/* jnp.maximum: NaN-propagating */
static float gx_max(float a, float b) { return (a > b || a != a) ? a : b; }

void gxo_math_probe2(int32_t n, const float* x, float* lg, float* th)
{
    for (int i = 0; i < n; ++i) { lg[i] = gx_log(x[i]); th[i] = gx_tanh(x[i]); }
}

void gxo_math_probe(int32_t n, const float* x, const float* y, float* s, float* c,
                    float* at2, float* ex)
{
    for (int i = 0; i < n; ++i) {
        gx_sincos(x[i], &s[i], &c[i]);
        at2[i] = gx_atan2(y[i], x[i]);
        ex[i] = gx_exp(x[i]);
    }
}

/* ------------------------------------------------------------------ */
/* jax.random restatement [derived: jax/_src/prng.py, random.py]        */
/* ------------------------------------------------------------------ */
static uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

/* Threefry-2x32, 20 rounds (Salmon et al. 2011; jax threefry2x32_p). */
static void threefry2x32(uint32_t k0, uint32_t k1, uint32_t x0, uint32_t x1,
                         uint32_t* o0, uint32_t* o1)
{
    static const int R[2][4] = {{13, 15, 26, 6}, {17, 29, 16, 24}};
    uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
    x0 += ks[0];
    x1 += ks[1];
    for (int i = 0; i < 5; ++i) {
        for (int j = 0; j < 4; ++j) {
            x0 += x1;
            x1 = rotl32(x1, R[i & 1][j]);
            x1 ^= x0;
        }
        x0 += ks[(i + 1) % 3];
        x1 += ks[(i + 2) % 3] + (uint32_t)(i + 1);
    }
    *o0 = x0;
    *o1 = x1;
}

void gxo_threefry2x32(uint32_t k0, uint32_t k1, uint32_t x0, uint32_t x1, uint32_t* out2)
{
    threefry2x32(k0, k1, x0, x1, &out2[0], &out2[1]);
}

/* element i of threefry_2x32(key, iota(n)): the count vector (padded to even
 * length) is cut in halves that feed the two block inputs; outputs are
 * concatenated [derived: prng.py threefry_2x32]. */
static uint32_t tf_bits_at(const uint32_t key[2], uint32_t n, uint32_t i)
{
    uint32_t half = (n + 1u) / 2u, o0, o1;
    if (i < half) {
        uint32_t hi = i + half;
        threefry2x32(key[0], key[1], i, hi < n ? hi : 0u, &o0, &o1);
        return o0;
    }
    threefry2x32(key[0], key[1], i - half, i, &o0, &o1);
    return o1;
}

/* jax.random.split(key, n)[j] = flat[2j], flat[2j+1], flat = bits over iota(2n) */
static void split_at(const uint32_t key[2], uint32_t n, uint32_t j, uint32_t out[2])
{
    out[0] = tf_bits_at(key, 2u * n, 2u * j);
    out[1] = tf_bits_at(key, 2u * n, 2u * j + 1u);
}

void gxo_split(const uint32_t* key, int32_t n, uint32_t* out_2n)
{
    for (int32_t j = 0; j < n; ++j) split_at(key, (uint32_t)n, (uint32_t)j, &out_2n[2 * j]);
}

/* jax.random.uniform(key, (), f32, minval, maxval) [derived: random.py _uniform] */
static float uniform_f32(const uint32_t key[2], float minval, float maxval)
{
    uint32_t bits = tf_bits_at(key, 1u, 0u);
    float f = bits_to_f((bits >> 9) | 0x3F800000u) - 1.0f;
    float v = f * (maxval - minval) + minval;
    return v > minval ? v : minval; /* lax.max(minval, v) */
}

float gxo_uniform(const uint32_t* key, float minval, float maxval)
{
    return uniform_f32(key, minval, maxval);
}

/* jax.random.randint(key, (n,), 0, span) element i [derived: random.py _randint] */
static uint32_t randint_at(const uint32_t k1[2], const uint32_t k2[2], uint32_t n,
                           uint32_t span, uint32_t i)
{
    uint32_t hi = tf_bits_at(k1, n, i), lo = tf_bits_at(k2, n, i);
    uint32_t mult = 65536u % span;
    mult = (mult * mult) % span;
    uint32_t off = (hi % span) * mult + (lo % span);
    return off % span;
}

void gxo_randint(const uint32_t* key, int32_t n, uint32_t span, int32_t* out_n)
{
    uint32_t k1[2], k2[2];
    split_at(key, 2, 0, k1);
    split_at(key, 2, 1, k2);
    if (span == 0) span = 1;
    for (int32_t i = 0; i < n; ++i)
        out_n[i] = (int32_t)randint_at(k1, k2, (uint32_t)n, span, (uint32_t)i);
}

/* ------------------------------------------------------------------ */
/* engine state                                                         */
/* ------------------------------------------------------------------ */
/* Point robot constants [derived from xmls/point.xml:3,5,16-20,37-39; SURVEY
 * Appendix B]: sphere r=.1 + box half .05 at (.1,0,0), density 1. */
#define PT_H 0.02f              /* point.xml:3 timestep */
#define PT_M 0.005188790204786391f
#define PT_MXC 0.0001f          /* mass * com offset along body x */
#define PT_IO 2.842182748581224e-05f /* inertia about the hinge axis */
#define PT_DXY 0.01f            /* point.xml:16-17 damping */
#define PT_DT 0.005f            /* point.xml:18 */
#define PT_GEAR 0.3f            /* point.xml:37-39 */
/* Actuator defaults the three <general gear="0.3"> elements inherit (point.xml:7-8,37-39) [derived:
 * MuJoCo XML reference, default/motor..velocity: "set the attributes of the general element using
 * Actuator shortcuts ... replacing any previous settings" -- ONE actuator default per class, written
 * by <motor> and then by <velocity> in document order]: ctrllimited, ctrlrange +-1; forcelimited,
 * forcerange +-.05; gain fixed 1; bias affine (0, 0, -kv), kv = 1 (velocity servo). */
#define PT_CTRLLIM 1.0f         /* point.xml:7-8 ctrlrange */
#define PT_FORCELIM 0.05f       /* point.xml:7-8 forcerange */
#define PT_KV 1.0f              /* <velocity> without kv: gainprm[0] stays 1, biasprm[2] = -1 */
#define GXO_ROBOT_POINT_BARE 4  /* round-1 reading kept selectable: general actuators WITHOUT the class defaults */

struct gxo_env {
    gxo_config cfg;
    int N, H, PL, NOBJ, D, bins; /* hazards, pillars (synthetic extension), goal + hazards + pillars */
    int nq, nv, nu, na; /* robot.nq/nv/nu (world.py:435-438) and action width */
    float h;            /* opt.timestep */
    int off_acc, off_ctrl, off_comp, off_glidar, off_hlidar, off_plidar, off_qpos, off_qvel, off_vel;
    float *qpos, *qvel, *pose0, *pose1, *objs; /* env-major */
    float *done0, *done1, *done2, *steps, *obs;
    float* pool; /* valid layouts, rows of (H+2)*2 */
    int layout_size;
    uint32_t key[2];
    int hist; /* number of step() calls so far, saturating at 2 (None-ness) */
};

static int g_threads = 0;
void gxo_set_threads(int32_t n) { g_threads = n; }
int gxo_get_threads(void)
{
#ifdef _OPENMP
    return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
    return 1;
#endif
}

int gxo_obs_dim(const gxo_env* e) { return e->D; }
void gxo_dims(const gxo_env* e, int32_t* nq, int32_t* nv, int32_t* nu, int32_t* na)
{
    *nq = e->nq; *nv = e->nv; *nu = e->nu; *na = e->na;
}
int gxo_layout_size(const gxo_env* e) { return e->layout_size; }

int gxo_create(const gxo_config* cfg, gxo_env** out)
{
    if (!cfg || !out || cfg->struct_size != (int32_t)sizeof(gxo_config)) return GXO_ERR_ARG;
    if (cfg->robot < 0 || cfg->robot > GXO_ROBOT_POINT_BARE) return GXO_ERR_UNSUPPORTED;
    if (cfg->env_num < 1 || cfg->hazards_num < 1 || cfg->hazards_num > 64) return GXO_ERR_ARG;
    if (cfg->pillars_num < 0 || cfg->hazards_num + cfg->pillars_num > 64) return GXO_ERR_ARG;
    if (cfg->lidar_num_bins < 3 || cfg->lidar_num_bins > 64) return GXO_ERR_ARG;
    if (cfg->env_offset < 0 || cfg->env_offset + cfg->env_num > cfg->env_total) return GXO_ERR_ARG;
    gxo_env* e = (gxo_env*)calloc(1, sizeof(gxo_env));
    e->cfg = *cfg;
    if (cfg->placements) { /* own copy: the caller's array need not outlive the call */
        const size_t n = (size_t)(cfg->hazards_num + cfg->pillars_num + 2) * 4;
        double* pc = (double*)malloc(n * sizeof(double));
        memcpy(pc, cfg->placements, n * sizeof(double));
        e->cfg.placements = pc;
    }
    e->N = cfg->env_num;
    e->H = cfg->hazards_num;
    e->PL = cfg->pillars_num;
    e->NOBJ = 1 + e->H + e->PL;
    e->bins = cfg->lidar_num_bins;
    if (cfg->robot == 0 || cfg->robot == GXO_ROBOT_POINT_BARE) { e->nq = 3; e->nv = 3; e->nu = 3; e->na = 2; e->h = PT_H; } /* point.xml */
    else if (cfg->robot == 1) { e->nq = 5; e->nv = 5; e->nu = 2; e->na = 2; e->h = 0.03f; } /* swimmer.xml */
    else if (cfg->robot == 2) { e->nq = 11; e->nv = 11; e->nu = 8; e->na = 8; e->h = 0.09f; } /* ant.xml */
    else { e->nq = 13; e->nv = 13; e->nu = 10; e->na = 10; e->h = 0.02f; }                 /* walker.xml */
    /* flat obs = concat over sorted(obs_space_dict keys)  engine.py:386-409,773-777 */
    int o = 0;
    e->off_acc = e->off_ctrl = e->off_comp = e->off_glidar = e->off_hlidar = e->off_plidar = -1;
    e->off_qpos = e->off_qvel = e->off_vel = -1;
    if (cfg->observe_acc) { e->off_acc = o; o += 2; }
    if (cfg->observe_ctrl) { e->off_ctrl = o; o += e->nu; }
    if (cfg->observe_goal_comp) { e->off_comp = o; o += 2; }
    if (cfg->observe_goal_lidar) { e->off_glidar = o; o += e->bins; }
    if (cfg->observe_hazards) { e->off_hlidar = o; o += e->bins; }
    if (cfg->observe_pillars && e->PL > 0) { e->off_plidar = o; o += e->bins; } /* 'pillars_lidar' sorts here */
    if (cfg->observe_qpos) { e->off_qpos = o; o += e->nq; }
    if (cfg->observe_qvel) { e->off_qvel = o; o += e->nv; }
    if (cfg->observe_vel) { e->off_vel = o; o += 2; }
    e->D = o;
    int N = e->N;
    e->qpos = (float*)calloc((size_t)N * e->nq, 4);
    e->qvel = (float*)calloc((size_t)N * e->nv, 4);
    e->pose0 = (float*)calloc((size_t)N * 4, 4);
    e->pose1 = (float*)calloc((size_t)N * 2, 4);
    e->objs = (float*)calloc((size_t)N * e->NOBJ * 2, 4);
    e->done0 = (float*)calloc(N, 4);
    e->done1 = (float*)calloc(N, 4);
    e->done2 = (float*)calloc(N, 4);
    e->steps = (float*)calloc(N, 4);
    e->obs = (float*)calloc((size_t)N * (e->D > 0 ? e->D : 1), 4);
    for (int i = 0; i < N; ++i) { e->pose0[4 * i + 2] = 1.0f; }
    /* PRNGKey(seed) = (seed >> 32, seed & 0xffffffff)  engine.py:216 */
    e->key[0] = 0u;
    e->key[1] = cfg->seed;
    e->hist = 0;
    *out = e;
    return GXO_OK;
}

void gxo_destroy(gxo_env* e)
{
    if (!e) return;
    free(e->qpos); free(e->qvel); free(e->pose0); free(e->pose1); free(e->objs);
    free(e->done0); free(e->done1); free(e->done2); free(e->steps); free(e->obs);
    free(e->pool);
    free((void*)e->cfg.placements);
    free(e);
}

/* ------------------------------------------------------------------ */
/* layout sampling  engine.py:433-452, 546-621                          */
/* ------------------------------------------------------------------ */
static double obj_keepout(const gxo_config* c, int obj, int nobj_total)
{
    /* placements order: goal, hazard0.., [pillar0..,] robot  engine.py:533-544 */
    if (obj == 0) return c->goal_keepout;
    if (obj == nobj_total - 1) return c->robot_keepout;
    if (obj <= c->hazards_num) return c->hazards_keepout;
    return c->pillars_keepout;
}

/* engine.py:546-572 for one candidate key; xy holds (H+PL+2)*2 floats. */
static int sample_layout(const gxo_config* c, const uint32_t key[2], float* xy)
{
    const int nobj = c->hazards_num + c->pillars_num + 2;
    uint32_t rng[2] = {key[0], key[1]};
    int success = 1;
    for (int o = 0; o < nobj; ++o) {
        double k = obj_keepout(c, o, nobj);
        /* constrain_placement  engine.py:574-577 (python floats -> f32 bounds); the rectangle is
         * placements_extents unless the object has its own placement / location (:600-612) */
        const double* rc = c->placements ? &c->placements[4 * o] : c->extents;
        float xmin = (float)(rc[0] + k), ymin = (float)(rc[1] + k);
        float xmax = (float)(rc[2] - k), ymax = (float)(rc[3] - k);
        int conflicted = 1;
        float px = -INFINITY, py = -INFINITY;
        for (int t = 0; t < 10; ++t) { /* engine.py:562 */
            uint32_t nrng[2], rng1[2], r1[2], r2[2];
            split_at(rng, 2, 0, nrng); /* rng, rng1 = split(rng, 2)  :563 */
            split_at(rng, 2, 1, rng1);
            rng[0] = nrng[0]; rng[1] = nrng[1];
            split_at(rng1, 2, 0, r1);  /* draw_placement :618 */
            split_at(rng1, 2, 1, r2);
            float cx = uniform_f32(r1, xmin, xmax); /* :619 */
            float cy = uniform_f32(r2, ymin, ymax); /* :620 */
            int flag = 1; /* placement_is_valid :549-555 */
            for (int p = 0; p < o; ++p) {
                float dx = cx - xy[2 * p], dy = cy - xy[2 * p + 1];
                float dist = sqrtf(dx * dx + dy * dy);
                float thr = (float)(obj_keepout(c, p, nobj) + c->placements_margin + k);
                if (dist < thr) flag = 0;
            }
            if (flag) { px = cx; py = cy; conflicted = 0; } /* :566-567 */
        }
        xy[2 * o] = px; xy[2 * o + 1] = py;
        if (conflicted) success = 0; /* :569 */
    }
    float dx = xy[2 * (nobj - 1)] - xy[0], dy = xy[2 * (nobj - 1) + 1] - xy[1];
    float d = sqrtf(dx * dx + dy * dy);
    if (d < c->robot_goal_min_dist) success = 0; /* :570-571 */
    return success;
}

/* reset_layout  engine.py:433-444 */
static int reset_layout(gxo_env* e)
{
    const int M = e->cfg.n_candidates, row = (e->NOBJ + 1) * 2;
    unsigned char* ok = (unsigned char*)malloc((size_t)M);
    float* all = (float*)malloc((size_t)M * row * 4);
    const uint32_t* key = e->key;
    const gxo_config* c = &e->cfg;
#pragma omp parallel for schedule(dynamic, 256) num_threads(gxo_get_threads())
    for (int j = 0; j < M; ++j) {
        uint32_t kj[2];
        split_at(key, (uint32_t)M, (uint32_t)j, kj); /* split(rng, 1e6)  :263 */
        ok[j] = (unsigned char)sample_layout(c, kj, &all[(size_t)j * row]);
    }
    int L = 0;
    for (int j = 0; j < M; ++j) L += ok[j];
    free(e->pool);
    e->pool = (float*)malloc((size_t)(L > 0 ? L : 1) * row * 4);
    int w = 0;
    for (int j = 0; j < M; ++j) /* idx = where(success > 0)[0]  :436 */
        if (ok[j]) memcpy(&e->pool[(size_t)(w++) * row], &all[(size_t)j * row], (size_t)row * 4);
    e->layout_size = L;
    free(ok);
    free(all);
    return L > e->cfg.env_total ? GXO_OK : GXO_ERR_LAYOUT; /* assert :444 */
}

/* ------------------------------------------------------------------ */
/* physics + observation                                                */
/* ------------------------------------------------------------------ */
typedef struct {
    float x, y, th, vx, vy, om;
} ptstate;

/* One mjx.step for the Point robot [derived, SURVEY Appendix B]:
 * forward(qpos,qvel,ctrl) -> pose, qacc ; Euler with implicit joint damping. */
static float pt_clip(float x, float lim) { return x < -lim ? -lim : (x > lim ? lim : x); } /* jp.clip: NaN stays */

/* actuator force on one DOF [derived: mjx fwd_actuation]: ctrl clamped to ctrlrange, force = gain*ctrl +
 * biasprm[2]*velocity with actuator velocity = gear*qvel, clamped to forcerange; qfrc = gear*force */
static float point_act(float ctrl, float vel, int bare)
{
    if (bare) return PT_GEAR * ctrl;
    const float u = pt_clip(ctrl, PT_CTRLLIM);
    const float force = pt_clip(u - PT_KV * (PT_GEAR * vel), PT_FORCELIM);
    return PT_GEAR * force;
}

static void point_substep_s(ptstate* s, const float ctrl[3], float pose[4], float qacc[3], int bare)
{
    /* kinematics: hinge quaternion (cos th/2, 0,0, sin th/2) -> xmat */
    float sh, ch;
    gx_sincos(0.5f * s->th, &sh, &ch);
    float c = ch * ch - sh * sh;
    float sn = 2.0f * (ch * sh);
    pose[0] = s->x; pose[1] = s->y; pose[2] = c; pose[3] = sn;
    /* joint-space inertia M = [[m,0,b],[0,m,d],[b,d,Io]] */
    float b = -(PT_MXC * sn), d = PT_MXC * c;
    float w2 = s->om * s->om;
    /* qfrc_smooth = passive - bias + actuator */
    float fx = (-(PT_DXY * s->vx) - (-(d * w2))) + point_act(ctrl[0], s->vx, bare);
    float fy = (-(PT_DXY * s->vy) - (b * w2)) + point_act(ctrl[1], s->vy, bare);
    float ft = (-(PT_DT * s->om) - 0.0f) + point_act(ctrl[2], s->om, bare);
    float t = b * fx + d * fy;
    /* Schur complement of the hinge row after eliminating the slides: Io - (b^2 + d^2)/m, and b^2 + d^2 =
     * (m xc)^2 (sin^2 + cos^2) = (m xc)^2 is a model constant: the solve multiplies by its reciprocal (round 3) */
    /* data.qacc = M^-1 f (no damping in M) */
    {
        const float ia = (float)(1.0 / 0.005188790204786391);
        const float id3 = (float)(1.0 / (2.842182748581224e-05 - (1.0e-4 * 1.0e-4) / 0.005188790204786391));
        float y3 = ft - t * ia;
        float q3 = y3 * id3;
        qacc[0] = (fx - b * q3) * ia;
        qacc[1] = (fy - d * q3) * ia;
        qacc[2] = q3;
    }
    /* Euler, damping implicit: (M + h D) qa = f */
    const float ia = (float)(1.0 / (0.005188790204786391 + 0.02 * 0.01));
    const float id3 = (float)(1.0 / ((2.842182748581224e-05 + 0.02 * 0.005) -
                                     (1.0e-4 * 1.0e-4) / (0.005188790204786391 + 0.02 * 0.01)));
    float y3 = ft - t * ia;
    float q3 = y3 * id3;
    float q1 = (fx - b * q3) * ia;
    float q2 = (fy - d * q3) * ia;
    s->vx = s->vx + PT_H * q1;
    s->vy = s->vy + PT_H * q2;
    s->om = s->om + PT_H * q3;
    s->x = s->x + PT_H * s->vx;
    s->y = s->y + PT_H * s->vy;
    s->th = s->th + PT_H * s->om;
}

static void point_substep(float q[3], float v[3], const float ctrl[3], float pose[4], float qacc[3], int bare)
{
    ptstate s = {q[0], q[1], q[2], v[0], v[1], v[2]};
    point_substep_s(&s, ctrl, pose, qacc, bare);
    q[0] = s.x; q[1] = s.y; q[2] = s.th; v[0] = s.vx; v[1] = s.vy; v[2] = s.om;
}

/* ------------------------------------------------------------------ */
/* Swimmer (xmls/swimmer.xml) [derived]: planar 3-link chain, qpos =      */
/* (x, y, th1, phi2, phi3); constants from tools/model_constants.py.      */
/* ------------------------------------------------------------------ */
#define SW_H 0.03f                         /* swimmer.xml:3 */
#define SW_M 0.22200588085367876f          /* capsule r=.02 l=.15 density 1000 (:18,23,27) */
#define SW_IC 0.00060383505197098225f      /* capsule inertia about a perpendicular axis */
#define SW_ARM 0.1f                        /* :6 joint armature */
#define SW_GEAR 20.0f                      /* :58-59 */
#define SW_LIM 1.7453292519943295f         /* :24,28 range +-100 deg */
#define SW_INVW2 9.3234461878793518f       /* dof_invweight0[motor1_rot] */
#define SW_INVW3 9.8671592198756102f       /* dof_invweight0[motor2_rot] */
#define SW_K 307.78701138811942f           /* 1/(dmax^2 tc^2), tc = max(.02, 2h) */
#define SW_B 35.087719298245617f           /* 2/(dmax tc) */
#define SW_A11 0.225f
#define SW_A21 0.15f
#define SW_A22 (-0.075f)
#define SW_A31 0.15f
#define SW_A32 (-0.15f)
#define SW_A33 (-0.075f)

typedef struct { float rd0, rd1, rd2, l10, l20, l21; } ldl3;

static void ldl3_factor(const float S[3][3], ldl3* f)
{
    f->rd0 = 1.0f / S[0][0];
    f->l10 = S[1][0] * f->rd0;
    f->l20 = S[2][0] * f->rd0;
    const float d1 = S[1][1] - f->l10 * S[1][0];
    f->rd1 = 1.0f / d1;
    const float t21 = S[2][1] - f->l20 * S[1][0];
    f->l21 = t21 * f->rd1;
    const float d2 = (S[2][2] - f->l20 * S[2][0]) - f->l21 * t21;
    f->rd2 = 1.0f / d2;
}

static void ldl3_solve(const ldl3* f, const float b[3], float x[3])
{
    const float y0 = b[0];
    const float y1 = b[1] - f->l10 * y0;
    const float y2 = (b[2] - f->l20 * y0) - f->l21 * y1;
    const float z2 = y2 * f->rd2;
    const float z1 = y1 * f->rd1 - f->l21 * z2;
    const float z0 = (y0 * f->rd0 - f->l10 * z1) - f->l20 * z2;
    x[0] = z0; x[1] = z1; x[2] = z2;
}

/* joint-limit row (MJX constraint._instantiate_limit_slide_hinge + _kbi):
 * present if violated; returns sign, aref and R = 1/D. */
static int limit_row(float q, float vel, float invw, float* sign, float* aref, float* R)
{
    const float dmin = q - (-SW_LIM), dmax_ = SW_LIM - q;
    const float pos = dmin < dmax_ ? dmin : dmax_;
    const float sg = dmin < dmax_ ? 1.0f : -1.0f;
    if (!(pos < 0.0f)) return 0;
    /* impedance: solimp = (.9, .95, .001, .5, 2) */
    const float ix = fabsf(pos) / 0.001f;
    float iy;
    if (ix < 0.5f) iy = 2.0f * (ix * ix);
    else iy = 1.0f - 2.0f * ((1.0f - ix) * (1.0f - ix));
    float imp = 0.9f + iy * (0.95f - 0.9f);
    if (imp < 0.9f) imp = 0.9f;
    if (imp > 0.95f) imp = 0.95f;
    if (ix > 1.0f) imp = 0.95f;
    *sign = sg;
    *aref = -(SW_B * (sg * vel)) - (SW_K * imp) * pos;
    float r = ((1.0f - imp) * invw) / imp;
    if (r < 1e-15f) r = 1e-15f;
    *R = r;
    return 1;
}

static void swimmer_substep(float q[5], float v[5], const float ctrl[2], float pose[4], float qacc[5])
{
    /* kinematics: hinge quaternions (cos, sin of half angles) composed down the chain */
    float sh1, ch1, sh2, ch2, sh3, ch3;
    gx_sincos(0.5f * q[2], &sh1, &ch1);
    gx_sincos(0.5f * q[3], &sh2, &ch2);
    gx_sincos(0.5f * q[4], &sh3, &ch3);
    const float w1 = ch1, z1 = sh1;
    const float w2 = w1 * ch2 - z1 * sh2, z2 = w1 * sh2 + z1 * ch2;
    const float w3 = w2 * ch3 - z2 * sh3, z3 = w2 * sh3 + z2 * ch3;
    const float c1 = w1 * w1 - z1 * z1, s1 = 2.0f * (w1 * z1);
    const float c2 = w2 * w2 - z2 * z2, s2 = 2.0f * (w2 * z2);
    const float c3 = w3 * w3 - z3 * z3, s3 = 2.0f * (w3 * z3);
    pose[0] = q[0]; pose[1] = q[1]; pose[2] = c1; pose[3] = s1;
    /* absolute angular rates */
    const float W1 = v[2], W2 = W1 + v[3], W3 = W2 + v[4];
    /* COM Jacobian columns g_ij = sum_{k>=j} a_ik n_k, n_k = (-s_k, c_k) */
    const float g11x = SW_A11 * -s1, g11y = SW_A11 * c1;
    const float g22x = SW_A22 * -s2, g22y = SW_A22 * c2;
    const float g21x = SW_A21 * -s1 + g22x, g21y = SW_A21 * c1 + g22y;
    const float g33x = SW_A33 * -s3, g33y = SW_A33 * c3;
    const float g32x = SW_A32 * -s2 + g33x, g32y = SW_A32 * c2 + g33y;
    const float g31x = SW_A31 * -s1 + g32x, g31y = SW_A31 * c1 + g32y;
    /* velocity-product acceleration of the COMs: -sum_k a_ik u_k W_k^2 */
    const float e1 = W1 * W1, e2 = W2 * W2, e3 = W3 * W3;
    const float q1x = -((SW_A11 * e1) * c1), q1y = -((SW_A11 * e1) * s1);
    const float q2x = -((SW_A21 * e1) * c1 + (SW_A22 * e2) * c2), q2y = -((SW_A21 * e1) * s1 + (SW_A22 * e2) * s2);
    const float q3x = -(((SW_A31 * e1) * c1 + (SW_A32 * e2) * c2) + (SW_A33 * e3) * c3);
    const float q3y = -(((SW_A31 * e1) * s1 + (SW_A32 * e2) * s2) + (SW_A33 * e3) * s3);
    /* mass matrix blocks */
    const float Mx[3] = {SW_M * ((g11x + g21x) + g31x), SW_M * (g22x + g32x), SW_M * g33x};
    const float My[3] = {SW_M * ((g11y + g21y) + g31y), SW_M * (g22y + g32y), SW_M * g33y};
    float T[3][3];
    T[0][0] = (SW_M * (((g11x * g11x + g11y * g11y) + (g21x * g21x + g21y * g21y)) + (g31x * g31x + g31y * g31y)) + 3.0f * SW_IC) + SW_ARM;
    T[1][0] = SW_M * ((g21x * g22x + g21y * g22y) + (g31x * g32x + g31y * g32y)) + 2.0f * SW_IC;
    T[2][0] = SW_M * (g31x * g33x + g31y * g33y) + SW_IC;
    T[1][1] = (SW_M * ((g22x * g22x + g22y * g22y) + (g32x * g32x + g32y * g32y)) + 2.0f * SW_IC) + SW_ARM;
    T[2][1] = SW_M * (g32x * g33x + g32y * g33y) + SW_IC;
    T[2][2] = (SW_M * (g33x * g33x + g33y * g33y) + SW_IC) + SW_ARM;
    T[0][1] = T[1][0]; T[0][2] = T[2][0]; T[1][2] = T[2][1];
    /* bias and smooth force: qfrc_smooth = (passive - bias) + actuator */
    const float bx = SW_M * ((q1x + q2x) + q3x), by = SW_M * ((q1y + q2y) + q3y);
    const float b1 = SW_M * (((g11x * q1x + g11y * q1y) + (g21x * q2x + g21y * q2y)) + (g31x * q3x + g31y * q3y));
    const float b2 = SW_M * ((g22x * q2x + g22y * q2y) + (g32x * q3x + g32y * q3y));
    const float b3 = SW_M * (g33x * q3x + g33y * q3y);
    float u0 = ctrl[0], u1 = ctrl[1]; /* ctrl clamped to ctrlrange for the force only */
    u0 = u0 < -1.0f ? -1.0f : (u0 > 1.0f ? 1.0f : u0);
    u1 = u1 < -1.0f ? -1.0f : (u1 > 1.0f ? 1.0f : u1);
    const float fx = 0.0f - bx, fy = 0.0f - by;
    const float ft[3] = {0.0f - b1, (0.0f - b2) + SW_GEAR * u0, (0.0f - b3) + SW_GEAR * u1};
    /* eliminate the (diagonal) translation block, factor the 3x3 Schur complement */
    const float imu = (float)(1.0 / (3.0 * 0.22200588085367876 + 0.1));
    float S[3][3], r[3];
    for (int j = 0; j < 3; ++j) {
        for (int k = 0; k <= j; ++k) S[j][k] = T[j][k] - (Mx[j] * Mx[k] + My[j] * My[k]) * imu;
        r[j] = ft[j] - (Mx[j] * fx + My[j] * fy) * imu;
    }
    S[0][1] = S[1][0]; S[0][2] = S[2][0]; S[1][2] = S[2][1];
    ldl3 F;
    ldl3_factor(S, &F);
    float a[3];
    ldl3_solve(&F, r, a); /* unconstrained qacc (angles) */
    /* joint limits on phi2, phi3 */
    float sg2 = 0, ar2 = 0, R2 = 0, sg3 = 0, ar3 = 0, R3 = 0;
    const int p2 = limit_row(q[3], v[3], SW_INVW2, &sg2, &ar2, &R2);
    const int p3 = limit_row(q[4], v[4], SW_INVW3, &sg3, &ar3, &R3);
    if (p2 || p3) {
        const float e2v[3] = {0.0f, 1.0f, 0.0f}, e3v[3] = {0.0f, 0.0f, 1.0f};
        float zc2[3], zc3[3];
        ldl3_solve(&F, e2v, zc2);
        ldl3_solve(&F, e3v, zc3);
        const float A22 = zc2[1], A33 = zc3[2], A23 = zc3[1] * (sg2 * sg3);
        const float E2 = sg2 * a[1] - ar2, E3 = sg3 * a[2] - ar3; /* J a0 - aref */
        float f2 = 0.0f, f3 = 0.0f;
        int done = 0;
        if (p2 && p3) {
            const float m22 = R2 + A22, m33 = R3 + A33;
            const float det = m22 * m33 - A23 * A23;
            const float g2 = ((-E2) * m33 - A23 * (-E3)) / det;
            const float g3 = (m22 * (-E3) - A23 * (-E2)) / det;
            if (g2 > 0.0f && g3 > 0.0f) { f2 = g2; f3 = g3; done = 1; }
        }
        if (!done && p2) {
            const float g2 = (-E2) / (R2 + A22);
            if (g2 > 0.0f && (!p3 || !(E3 + A23 * g2 < 0.0f))) { f2 = g2; f3 = 0.0f; done = 1; }
        }
        if (!done && p3) {
            const float g3 = (-E3) / (R3 + A33);
            if (g3 > 0.0f && (!p2 || !(E2 + A23 * g3 < 0.0f))) { f3 = g3; f2 = 0.0f; done = 1; }
        }
        const float rc[3] = {r[0], r[1] + sg2 * f2, r[2] + sg3 * f3};
        ldl3_solve(&F, rc, a);
    }
    const float ax = (fx - ((Mx[0] * a[0] + Mx[1] * a[1]) + Mx[2] * a[2])) * imu;
    const float ay = (fy - ((My[0] * a[0] + My[1] * a[1]) + My[2] * a[2])) * imu;
    qacc[0] = ax; qacc[1] = ay; qacc[2] = a[0]; qacc[3] = a[1]; qacc[4] = a[2];
    for (int k = 0; k < 5; ++k) v[k] = v[k] + SW_H * qacc[k];
    for (int k = 0; k < 5; ++k) q[k] = q[k] + SW_H * v[k];
}

#include "gx_oracle_ant.inc"

#include "gx_oracle_legs.inc"

/* probe: one walker mjx.step; dbg = nv*nv dense mass matrix + nv smooth force (nv = 13) */
void gxo_walker_probe(const float* q, const float* v, const float* ctrl, float* q2, float* v2, float* qacc,
                      float* pose, float* dbg)
{
    float qq[13], vv[13];
    for (int k = 0; k < 13; ++k) { qq[k] = q[k]; vv[k] = v[k]; }
    g_legs_dbg = dbg;
    legs_substep(&LG_WALKER, qq, vv, ctrl, pose, qacc);
    g_legs_dbg = NULL;
    for (int k = 0; k < 13; ++k) { q2[k] = qq[k]; v2[k] = vv[k]; }
}

/* probe: one ant mjx.step from (q, v, ctrl); dense M (qpos coordinates) and smooth force in dbg[132] */
void gxo_ant_probe(const float* q, const float* v, const float* ctrl, float* q2, float* v2, float* qacc,
                   float* pose, float* dbg)
{
    float qq[11], vv[11];
    for (int k = 0; k < 11; ++k) { qq[k] = q[k]; vv[k] = v[k]; }
    g_ant_dbg = dbg;
    ant_substep(qq, vv, ctrl, pose, qacc);
    g_ant_dbg = NULL;
    for (int k = 0; k < 11; ++k) { q2[k] = qq[k]; v2[k] = vv[k]; }
}

#define GX_MAXQ 13
/* 'robot_rot' (engine.py:114,342-345): world.py:117 turns the robot's ROOT body by this angle about z, and its joints
 * with it (slide axes and hinge are given in the body's frame).  The dynamics do not depend on the angle (gravity is
 * along z, the floor is z = 0), so the steps are evaluated in the root body's frame and every pose that leaves them --
 * (x, y, cos, sin) of the robot body -- is turned into the world frame: cos / sin of the angle from the root quaternion
 * rot2quat(robot_rot) = (w, 0, 0, z) as the body's x axis (w^2 - z^2, 2 w z).  qpos / qvel stay joint coordinates;
 * layout2qpos writes the layout's robot xy into the slide joints as it is (engine.py:635-638), so a rotated robot
 * starts at R(robot_rot) . xy -- the reference's behaviour, reproduced. */
static void world_pose(const gxo_env* e, float pose[4])
{
    if (e->cfg.robot_rot == 0.0f) return;
    const float w = (float)cos(0.5 * (double)e->cfg.robot_rot), z = (float)sin(0.5 * (double)e->cfg.robot_rot);
    const float rc = w * w - z * z, rs = 2.0f * (w * z);
    const float x = pose[0], y = pose[1], c = pose[2], s = pose[3];
    pose[0] = rc * x - rs * y;
    pose[1] = rs * x + rc * y;
    pose[2] = rc * c - rs * s;
    pose[3] = rs * c + rc * s;
}

/* one mjx.step of the configured robot (pose in the root body's frame: callers go through robot_step_w) */
static void robot_substep(const gxo_env* e, float* q, float* v, const float* ctrl, float pose[4], float* qacc)
{
    if (e->cfg.robot == 0) point_substep(q, v, ctrl, pose, qacc, 0);
    else if (e->cfg.robot == GXO_ROBOT_POINT_BARE) point_substep(q, v, ctrl, pose, qacc, 1);
    else if (e->cfg.robot == 1) swimmer_substep(q, v, ctrl, pose, qacc);
    else if (e->cfg.robot == 2) ant_substep(q, v, ctrl, pose, qacc);
    else legs_substep(&LG_WALKER, q, v, ctrl, pose, qacc);
}
/* physics_steps_per_control_step substeps (engine.py:689), the pose they return in the world frame */
static void robot_step_w(const gxo_env* e, float* q, float* v, const float* ctrl, float pose[4], float* qacc)
{
    for (int k = 0; k < e->cfg.physics_steps; ++k) robot_substep(e, q, v, ctrl, pose, qacc);
    world_pose(e, pose);
}

/* convert_action engine.py:672-685: Point rotates (a0,0,0) by the PRE-step xmat; others pass through */
static void convert_action(const gxo_env* e, const float pose0[4], const float* a, float* ctrl)
{
    if (e->cfg.robot == 0 || e->cfg.robot == GXO_ROBOT_POINT_BARE) { ctrl[0] = pose0[2] * a[0]; ctrl[1] = pose0[3] * a[0]; ctrl[2] = a[1]; }
    else { for (int k = 0; k < e->nu; ++k) ctrl[k] = a[k]; }
}

/* pose of the robot body from qpos (mjx.forward kinematics), qpos with zero angles */
static void pose_of_rest(const gxo_env* e, const float* q, float pose[4])
{
    if (e->cfg.robot == 2 || e->cfg.robot == 3) ant_pose(q, pose); /* ant, walker: x slide, z hinge, body-y slide */
    else {
        float sh, ch;
        gx_sincos(0.5f * q[2], &sh, &ch);
        pose[0] = q[0]; pose[1] = q[1]; pose[2] = ch * ch - sh * sh; pose[3] = 2.0f * (ch * sh);
    }
    world_pose(e, pose);
}

/* obs_lidar engine.py:846-900 over `n` objects (xy pairs), pose=(x,y,c,s). */
static void obs_lidar(const gxo_env* e, const float pose[4], const float* objs, int n, float* out)
{
    const int B = e->bins;
    const float bin_size = (float)((M_PI * 2) / B); /* :880 weak-typed python float */
    for (int b = 0; b < B; ++b) out[b] = 0.0f;
    for (int o = 0; o < n; ++o) {
        float dx = objs[2 * o] - pose[0], dy = objs[2 * o + 1] - pose[1];
        /* ego_xy :817-826: world_3vec @ R, R = [[c,-s,0],[s,c,0],[0,0,1]] */
        float zx = dx * pose[2] + dy * pose[3];
        float zy = dx * (-pose[3]) + dy * pose[2];
        float dist = sqrtf(zx * zx + zy * zy); /* :878 */
        float ang = gx_atan2(zy, zx);          /* :879 */
        if (ang != 0.0f && ang < 0.0f) ang = ang + TWO_PI_F; /* jnp.remainder */
        int bin;
        float q = ang / bin_size;
        if (!(q >= 0.0f)) bin = 0; /* NaN: defined as 0 (obs carries NaN anyway) */
        else if (q >= (float)B) bin = B;
        else bin = (int)q; /* :882 */
        float bin_angle = bin_size * (float)bin; /* :883 */
        float sensor;
        if (!e->cfg.lidar_max_dist_set)
            sensor = gx_exp((-e->cfg.lidar_exp_gain) * dist); /* :886 */
        else
            sensor = gx_max(0.0f, e->cfg.lidar_max_dist - dist) / e->cfg.lidar_max_dist; /* :888 */
        /* obs[bin] read clamps, out-of-range scatter is dropped [derived: jnp indexing] */
        if (bin < B) out[bin] = gx_max(out[bin], sensor); /* :889-890 */
        if (e->cfg.lidar_alias) {                          /* :893-899 */
            float alias = (ang - bin_angle) / bin_size;
            int bp = (bin + 1) % B, bm = (bin - 1 + B) % B;
            out[bp] = gx_max(out[bp], alias * sensor);
            out[bm] = gx_max(out[bm], (1.0f - alias) * sensor);
        }
    }
}

/* Engine.obs engine.py:738-778 -> flat row.  vel/acc passed in. */
static void build_obs(const gxo_env* e, const float pose[4], const float* objs,
                      const float* ctrl, const float* q, const float* v, const float vel[2],
                      const float acc[2], float* row)
{
    if (e->off_acc >= 0) { row[e->off_acc] = acc[0]; row[e->off_acc + 1] = acc[1]; }
    if (e->off_ctrl >= 0) { for (int i = 0; i < e->nu; ++i) row[e->off_ctrl + i] = ctrl[i]; }
    if (e->off_comp >= 0) { /* obs_compass :834-844 */
        float dx = objs[0] - pose[0], dy = objs[1] - pose[1];
        row[e->off_comp] = dx * pose[2] + dy * pose[3];
        row[e->off_comp + 1] = dx * (-pose[3]) + dy * pose[2];
    }
    if (e->off_glidar >= 0) obs_lidar(e, pose, objs, 1, &row[e->off_glidar]);
    if (e->off_hlidar >= 0) obs_lidar(e, pose, objs + 2, e->H, &row[e->off_hlidar]);
    if (e->off_plidar >= 0) obs_lidar(e, pose, objs + 2 * (1 + e->H), e->PL, &row[e->off_plidar]);
    if (e->off_qpos >= 0) { for (int i = 0; i < e->nq; ++i) row[e->off_qpos + i] = q[i]; }
    if (e->off_qvel >= 0) { for (int i = 0; i < e->nv; ++i) row[e->off_qvel + i] = v[i]; }
    if (e->off_vel >= 0) { row[e->off_vel] = vel[0]; row[e->off_vel + 1] = vel[1]; }
}

static float dist_goal(const float* objs, const float* pose_xy)
{
    float dx = objs[0] - pose_xy[0], dy = objs[1] - pose_xy[1]; /* goal_pos :780-785 */
    return sqrtf(dx * dx + dy * dy);
}

static void load_layout(gxo_env* e, int i, const float* lay)
{
    /* layout2qpos engine.py:623-639: goal, hazards -> their slide joints; robot -> qpos[0:2] */
    memcpy(&e->objs[(size_t)i * e->NOBJ * 2], lay, (size_t)e->NOBJ * 2 * 4);
    float* q = &e->qpos[(size_t)i * e->nq];
    float* v = &e->qvel[(size_t)i * e->nv];
    for (int k = 0; k < e->nq; ++k) q[k] = 0.0f;
    for (int k = 0; k < e->nv; ++k) v[k] = 0.0f;
    /* robot_x / robot_y joints by NAME (:635-638): qpos 0,1 for point and swimmer, 0,2 for the ant and the walker */
    q[0] = lay[2 * e->NOBJ];
    q[(e->cfg.robot == 2 || e->cfg.robot == 3) ? 2 : 1] = lay[2 * e->NOBJ + 1];
}

/* get_layout engine.py:446-452: idx = randint(key, (env_num,), 0, layout_size) */
static void layout_indices(const gxo_env* e, uint32_t* idx)
{
    uint32_t k1[2], k2[2];
    split_at(e->key, 2, 0, k1);
    split_at(e->key, 2, 1, k2);
    uint32_t span = e->layout_size > 0 ? (uint32_t)e->layout_size : 1u;
    for (int i = 0; i < e->N; ++i)
        idx[i] = randint_at(k1, k2, (uint32_t)e->cfg.env_total, span,
                            (uint32_t)(e->cfg.env_offset + i));
}

int gxo_reset(gxo_env* e, float* obs)
{
    int rc = reset_layout(e); /* :457 */
    if (rc != GXO_OK && e->layout_size < 1) return rc;
    uint32_t* idx = (uint32_t*)malloc((size_t)e->N * 4);
    layout_indices(e, idx); /* :458 */
    const int row = (e->NOBJ + 1) * 2;
    const float zero5[GX_MAXQ] = {0, 0, 0, 0, 0}, zero2[2] = {0, 0};
    for (int i = 0; i < e->N; ++i) {
        load_layout(e, i, &e->pool[(size_t)idx[i] * row]);
        /* mjx_reset :644-657: ctrl=0, mjx.forward -> pose(qpos), obs without history */
        float* p0 = &e->pose0[4 * i];
        pose_of_rest(e, &e->qpos[(size_t)i * e->nq], p0);
        build_obs(e, p0, &e->objs[(size_t)i * e->NOBJ * 2], zero5, &e->qpos[(size_t)i * e->nq],
                  &e->qvel[(size_t)i * e->nv], zero2, zero2, &e->obs[(size_t)i * e->D]);
        e->steps[i] = 0.0f; /* :463 */
    }
    free(idx);
    memcpy(obs, e->obs, (size_t)e->N * e->D * 4);
    return rc;
}

int gxo_step(gxo_env* e, const float* action, float* obs, float* reward, float* cost,
             float* done, float* qacc_out)
{
    const int N = e->N, D = e->D, H = e->H, PL = e->PL;
    /* update_data :426-431 */
    memcpy(e->done2, e->done1, (size_t)N * 4);
    memcpy(e->done1, e->done0, (size_t)N * 4);
    {
        uint32_t nk[2];
        split_at(e->key, 2, 0, nk);
        e->key[0] = nk[0]; e->key[1] = nk[1];
    }
    const int have_last = e->hist >= 1, have_last_last = e->hist >= 2;
    const float dt = e->h * (float)e->cfg.physics_steps; /* :235 */
    const int nq = e->nq, nv = e->nv, na = e->na;
#pragma omp parallel for schedule(static) num_threads(gxo_get_threads())
    for (int i = 0; i < N; ++i) {
        float* p0 = &e->pose0[4 * i];
        float* p1 = &e->pose1[2 * i];
        const float* objs = &e->objs[(size_t)i * e->NOBJ * 2];
        float P1[2] = {p0[0], p0[1]};   /* last_data.xpos */
        float P2[2] = {p1[0], p1[1]};   /* last_last_data.xpos */
        /* convert_action :672-685 with the PRE-step xmat */
        float ctrl[GX_MAXQ] = {0, 0, 0, 0, 0};
        convert_action(e, p0, &action[(size_t)na * i], ctrl);
        float q[GX_MAXQ], v[GX_MAXQ];
        for (int k = 0; k < nq; ++k) q[k] = e->qpos[(size_t)i * nq + k];
        for (int k = 0; k < nv; ++k) v[k] = e->qvel[(size_t)i * nv + k];
        float pose[4], qacc[GX_MAXQ] = {0, 0, 0, 0, 0};
        robot_step_w(e, q, v, ctrl, pose, qacc); /* :689 */
        /* ego_vel_acc :902-929 */
        float vel[2] = {0, 0}, acc[2] = {0, 0};
        if (e->off_vel >= 0 || e->off_acc >= 0) {
            float pl[2] = {pose[0], pose[1]}, pll[2] = {pose[0], pose[1]};
            float ld = e->done1[i], lld = e->done2[i];
            if (have_last) {
                if (!(ld > 0.0f)) { pl[0] = P1[0]; pl[1] = P1[1]; }
                if (have_last_last) {
                    if (lld + ld > 0.0f) { pll[0] = pl[0]; pll[1] = pl[1]; }
                    else { pll[0] = P2[0]; pll[1] = P2[1]; }
                }
            }
            float vw[2], lvw[2], aw[2];
            for (int k = 0; k < 2; ++k) {
                vw[k] = (pose[k] - pl[k]) / dt;
                lvw[k] = (pl[k] - pll[k]) / dt;
                aw[k] = (vw[k] - lvw[k]) / dt;
            }
            vel[0] = vw[0] * pose[2] + vw[1] * pose[3];
            vel[1] = vw[0] * (-pose[3]) + vw[1] * pose[2];
            acc[0] = aw[0] * pose[2] + aw[1] * pose[3];
            acc[1] = aw[0] * (-pose[3]) + aw[1] * pose[2];
        }
        float* row = &e->obs[(size_t)i * D];
        build_obs(e, pose, objs, ctrl, q, v, vel, acc, row); /* :690 */
        /* reward_done :787-802 */
        float dg = dist_goal(objs, pose);
        float last = dg;
        if (have_last && !(e->done1[i] > 0.0f)) last = dist_goal(objs, P1);
        float dd = last - dg;
        float r = dd * e->cfg.reward_distance;
        float dn = dg < e->cfg.goal_size ? 1.0f : 0.0f;
        if (fabsf(dd) > 1.0f) { dn = 1.0f; r = 0.0f; }
        /* cost :804-811 */
        float cs = 0.0f;
        for (int h = 0; h < H; ++h) {
            float dx = objs[2 + 2 * h] - pose[0], dy = objs[3 + 2 * h] - pose[1];
            float dh = sqrtf(dx * dx + dy * dy);
            float below = dh < e->cfg.hazards_size ? dh : e->cfg.hazards_size; /* jp.minimum */
            if (dh != dh) below = dh;
            cs = cs + (e->cfg.hazards_size - below);
        }
        for (int h = 0; h < PL; ++h) { /* synthetic pillars: the same dense form with pillars_size */
            float dx = objs[2 * (1 + H + h)] - pose[0], dy = objs[2 * (1 + H + h) + 1] - pose[1];
            float dh = sqrtf(dx * dx + dy * dy);
            float below = dh < e->cfg.pillars_size ? dh : e->cfg.pillars_size;
            if (dh != dh) below = dh;
            cs = cs + (e->cfg.pillars_size - below);
        }
        /* NaN/Inf guard :696-699 */
        int bad = 0;
        for (int k = 0; k < D; ++k) if (!(fabsf(row[k]) <= 3.4028234663852886e38f)) bad = 1;
        if (bad) { r = 0.0f; dn = 1.0f; }
        /* timeout + step counter :492-493 */
        if (e->steps[i] > (float)e->cfg.num_steps) dn = 1.0f;
        e->steps[i] = dn > 0.0f ? 0.0f : e->steps[i] + 1.0f;
        /* commit */
        for (int k = 0; k < nq; ++k) e->qpos[(size_t)i * nq + k] = q[k];
        for (int k = 0; k < nv; ++k) e->qvel[(size_t)i * nv + k] = v[k];
        p1[0] = P1[0]; p1[1] = P1[1];
        p0[0] = pose[0]; p0[1] = pose[1]; p0[2] = pose[2]; p0[3] = pose[3];
        e->done0[i] = dn;
        reward[i] = r; cost[i] = cs; done[i] = dn;
        if (qacc_out) { for (int k = 0; k < nv; ++k) qacc_out[(size_t)i * nv + k] = qacc[k]; }
    }
    if (e->hist < 2) e->hist++;
    memcpy(obs, e->obs, (size_t)N * D * 4);
    return GXO_OK;
}

int gxo_reset_done(gxo_env* e, float* obs)
{
    const int N = e->N, D = e->D, row = (e->NOBJ + 1) * 2;
    if (e->hist == 0) { /* self._done is None: :713 falls through */
        memcpy(obs, e->obs, (size_t)N * D * 4);
        return GXO_OK;
    }
    if (e->layout_size < 1) return GXO_ERR_LAYOUT;
    uint32_t* idx = (uint32_t*)malloc((size_t)N * 4);
    layout_indices(e, idx); /* :500 */
    memcpy(obs, e->obs, (size_t)N * D * 4); /* self._obs is NOT updated by reset_done (:501) */
    const float zero5[GX_MAXQ] = {0, 0, 0, 0, 0}, zero2[2] = {0, 0};
    for (int i = 0; i < N; ++i) {
        if (!(e->done0[i] > 0.0f)) continue; /* :715-717 */
        load_layout(e, i, &e->pool[(size_t)idx[i] * row]);
        /* fake step (:719-724) on a scratch copy: its pose/qpos/qvel feed the obs only;
         * the returned data keeps the stale xpos/xmat (:731). */
        float q[GX_MAXQ], v[GX_MAXQ], pose[4], qacc[GX_MAXQ];
        for (int k = 0; k < e->nq; ++k) q[k] = e->qpos[(size_t)i * e->nq + k];
        for (int k = 0; k < e->nv; ++k) v[k] = 0.0f;
        robot_step_w(e, q, v, zero5, pose, qacc);
        build_obs(e, pose, &e->objs[(size_t)i * e->NOBJ * 2], zero5, q, v, zero2, zero2,
                  &obs[(size_t)i * D]); /* :726-729 */
    }
    free(idx);
    return GXO_OK;
}

int gxo_get_state(const gxo_env* e, float* qpos, float* qvel, float* pose0, float* pose1,
                  float* objs, float* done0, float* done1, float* steps, uint32_t* key,
                  int32_t* hist)
{
    const size_t N = (size_t)e->N;
    if (qpos) memcpy(qpos, e->qpos, N * e->nq * 4);
    if (qvel) memcpy(qvel, e->qvel, N * e->nv * 4);
    if (pose0) memcpy(pose0, e->pose0, N * 16);
    if (pose1) memcpy(pose1, e->pose1, N * 8);
    if (objs) memcpy(objs, e->objs, N * e->NOBJ * 8);
    if (done0) memcpy(done0, e->done0, N * 4);
    if (done1) memcpy(done1, e->done1, N * 4);
    if (steps) memcpy(steps, e->steps, N * 4);
    if (key) { key[0] = e->key[0]; key[1] = e->key[1]; }
    if (hist) *hist = e->hist;
    return GXO_OK;
}

int gxo_set_state(gxo_env* e, const float* qpos, const float* qvel, const float* pose0,
                  const float* pose1, const float* objs, const float* done0,
                  const float* done1, const float* steps, const uint32_t* key,
                  const int32_t* hist)
{
    const size_t N = (size_t)e->N;
    if (qpos) memcpy(e->qpos, qpos, N * e->nq * 4);
    if (qvel) memcpy(e->qvel, qvel, N * e->nv * 4);
    if (pose0) memcpy(e->pose0, pose0, N * 16);
    if (pose1) memcpy(e->pose1, pose1, N * 8);
    if (objs) memcpy(e->objs, objs, N * e->NOBJ * 8);
    if (done0) memcpy(e->done0, done0, N * 4);
    if (done1) memcpy(e->done1, done1, N * 4);
    if (steps) memcpy(e->steps, steps, N * 4);
    if (key) { e->key[0] = key[0]; e->key[1] = key[1]; }
    if (hist) e->hist = *hist;
    return GXO_OK;
}

int gxo_get_pool(const gxo_env* e, float* pool, int32_t max_rows)
{
    int n = e->layout_size < max_rows ? e->layout_size : max_rows;
    if (n > 0) memcpy(pool, e->pool, (size_t)n * (e->NOBJ + 1) * 2 * 4);
    return n;
}

/* ------------------------------------------------------------------ */
/* closed-loop rollout with an on-device policy      */
/* `ac.step(o)` of safe_rl_libX/trpo/trpo_core.py:110-173 for           */
/* MLPActorCritic(hidden_sizes=(hd,hd), tanh): Gaussian actor with a    */
/* state-independent log_std, MLP critic.  The noise is our own         */
/* counter-based stream (the reference uses torch's global generator).  */
/* params: pi{W1[Hd][D] b1 W2[Hd][Hd] b2 W3[A][Hd] b3} v{.. W3[1][Hd] b3} log_std[A] */
/* ------------------------------------------------------------------ */
#define POL_HD_MAX 256

/* hd = hidden width: 64 (the reference default, trpo.py:606 --hid), 128 or 256 (any multiple of 64 up to POL_HD_MAX).
 * Hidden units: one sequential fmaf chain over the inputs each.  Output layer: 16 partial sums -- partial l takes the
 * units 64 c + 4 l + j (c = 0 .. hd/64 - 1, j = 0 .. 3) in that order -- folded by a butterfly (xor 8, 4, 2, 1): the
 * order in which the 16 lanes that own an env combine their shares in the kernels (gx_policy.h, gx_policy_step.hip). */
static void mlp_forward(const float* w, const float* x, int D, int Out, int hd, float* out)
{
    const float *W1 = w, *b1 = W1 + hd * D, *W2 = b1 + hd, *b2 = W2 + hd * hd;
    const float *W3 = b2 + hd, *b3 = W3 + Out * hd;
    float h1[POL_HD_MAX], h2[POL_HD_MAX];
    for (int j = 0; j < hd; ++j) {
        float acc = b1[j];
        for (int k = 0; k < D; ++k) acc = fmaf(x[k], W1[j * D + k], acc);
        h1[j] = gx_tanh(acc);
    }
    for (int j = 0; j < hd; ++j) {
        float acc = b2[j];
        for (int k = 0; k < hd; ++k) acc = fmaf(h1[k], W2[j * hd + k], acc);
        h2[j] = gx_tanh(acc);
    }
    for (int o = 0; o < Out; ++o) {
        float pl[16], ql[16];
        for (int l = 0; l < 16; ++l) {
            float pp = 0.0f;
            for (int c = 0; 64 * c < hd; ++c)
                for (int j = 0; j < 4; ++j) pp = fmaf(h2[64 * c + 4 * l + j], W3[o * hd + 64 * c + 4 * l + j], pp);
            pl[l] = pp;
        }
        for (int off = 8; off >= 1; off >>= 1) {
            for (int l = 0; l < 16; ++l) ql[l] = pl[l] + pl[l ^ off];
            memcpy(pl, ql, sizeof pl);
        }
        out[o] = b3[o] + pl[0];
    }
}

static int mlp_size(int D, int Out, int hd) { return hd * D + hd + hd * hd + hd + Out * hd + Out; }

/* two standard normals from one Threefry block keyed by `seed`, counter (global env, step*16+pair) */
static void normal_pair(const uint32_t seed[2], uint32_t env, uint32_t ctr, float* z0, float* z1)
{
    uint32_t b0, b1;
    threefry2x32(seed[0], seed[1], env, ctr, &b0, &b1);
    const float u1 = (float)((b0 >> 8) + 1u) * 5.9604644775390625e-08f; /* (0,1] */
    const float u2 = (float)(b1 >> 8) * 5.9604644775390625e-08f;        /* [0,1) */
    const float r = sqrtf(-2.0f * gx_log(u1));
    float sn, cs;
    gx_sincos(TWO_PI_F * u2, &sn, &cs);
    *z0 = r * cs;
    *z1 = r * sn;
}

int gxo_rollout_policy(gxo_env* e, int32_t T, int32_t hidden, const float* params, const uint32_t* seed,
                       uint32_t t0, const float* obs0, float* obs_in, float* act, float* logp, float* val,
                       float* mu_out, float* rew, float* cost, float* done, float* obs_last, float* val_last,
                       float* logstd_out)
{
    if (hidden < 64 || hidden > POL_HD_MAX || (hidden & 63)) return GXO_ERR_UNSUPPORTED;
    const int N = e->N, D = e->D, A = e->na, hd = hidden;
    const float* wpi = params;
    const float* wv = params + mlp_size(D, A, hd);
    const float* log_std = wv + mlp_size(D, 1, hd);
    float std[16], lstd[16];
    for (int d = 0; d < A; ++d) { std[d] = gx_exp(log_std[d]); lstd[d] = gx_log(std[d]); logstd_out[d] = lstd[d]; }
    float* cur = (float*)malloc((size_t)N * D * 4);
    float* nxt = (float*)malloc((size_t)N * D * 4);
    float* a_t = (float*)malloc((size_t)N * A * 4);
    float* qa = (float*)malloc((size_t)N * e->nv * 4);
    memcpy(cur, obs0, (size_t)N * D * 4);
    for (int t = 0; t < T; ++t) {
        memcpy(&obs_in[(size_t)t * N * D], cur, (size_t)N * D * 4);
        for (int i = 0; i < N; ++i) {
            float m[16], v1[1], z[16];
            mlp_forward(wpi, &cur[(size_t)i * D], D, A, hd, m);
            mlp_forward(wv, &cur[(size_t)i * D], D, 1, hd, v1);
            for (int pr = 0; 2 * pr < A; ++pr)
                normal_pair(seed, (uint32_t)(e->cfg.env_offset + i), (t0 + (uint32_t)t) * 16u + (uint32_t)pr,
                            &z[2 * pr], &z[2 * pr + 1]);
            float lp = 0.0f;
            for (int d = 0; d < A; ++d) {
                const float a = fmaf(std[d], z[d], m[d]);
                const float df = a - m[d];
                const float var = std[d] * std[d];
                lp = lp + ((-(df * df) / (2.0f * var) - lstd[d]) - 0.9189385332046727f);
                a_t[(size_t)i * A + d] = a;
                act[((size_t)t * N + i) * A + d] = a;
                mu_out[((size_t)t * N + i) * A + d] = m[d];
            }
            logp[(size_t)t * N + i] = lp;
            val[(size_t)t * N + i] = v1[0];
        }
        int rc = gxo_step(e, a_t, nxt, &rew[(size_t)t * N], &cost[(size_t)t * N], &done[(size_t)t * N], qa);
        if (rc != GXO_OK) return rc;
        rc = gxo_reset_done(e, cur); /* post-reset observation feeds the next policy step */
        if (rc != GXO_OK) return rc;
    }
    memcpy(obs_last, cur, (size_t)N * D * 4);
    for (int i = 0; i < N; ++i) {
        float v1[1];
        mlp_forward(wv, &cur[(size_t)i * D], D, 1, hd, v1);
        val_last[i] = v1[0];
    }
    free(cur); free(nxt); free(a_t); free(qa);
    return GXO_OK;
}
