// gx_device.h -- device-side building blocks of the GUARD step path (gfx950).
//
// Everything here is fp32 with one IEEE operation per written operator (the
// library is compiled with -ffp-contract=off; fused operations are spelled
// fmaf()).  Division and sqrt rely on hipcc's default correctly-rounded
// lowering.  The polynomial coefficients come from tools/fit_math.py.
//
// Reference lines: /root/reference/safe_rl_envs/safe_rl_envs/envs/engine.py.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GX_HD __host__ __device__ __forceinline__
#define GX_D __device__ __forceinline__

namespace gx {

// ---------------------------------------------------------------------------
// parameters shared by all kernels (passed by value -> SGPRs)
// ---------------------------------------------------------------------------
struct Params {
    int N;        // local envs
    int Npad;     // N rounded up to 256: stride of every SoA array
    int H;        // hazards
    int PL;       // pillars (synthetic extension, include/guardx.h): objects H+1 .. H+PL
    int nobj;     // 1 + H + PL   (goal, hazards, pillars)
    int P;        // float4 object-pair arrays = ceil(nobj / 2)
    int bins;     // lidar bins
    int D;        // flat obs width
    int off_acc, off_ctrl, off_comp, off_gl, off_hl, off_pl, off_qpos, off_qvel, off_vel;
    int lidar_alias, lidar_max_dist_set;
    float lidar_max_dist, neg_gain, bin_size;
    float goal_size, hazards_size, pillars_size, reward_distance, num_steps_f, dt;
    float goal_cut; // min{x : fl(sqrt(x)) >= goal_size}: sqrtf(d2) < goal_size  <=>  d2 < goal_cut, exactly
    int physics_steps;
    int env_total, env_offset;
    int have_last, have_last_last; // None-ness of _last_done / _last_last_done
    int hist_on;                   // observe_vel || observe_acc
    int robot;                     // 0 point, 1 swimmer
    // 'robot_rot' (engine.py:114,342-345 -> world.py:117): the robot's root body is turned by this angle about z, its
    // joints with it.  The dynamics are evaluated in the root body's frame (they do not depend on the angle: gravity
    // is along z, the floor is the plane z = 0) and every pose that leaves a step is turned into the world frame
    int rot_on;
    float rot_c, rot_s;            // cos / sin of robot_rot (from the root quaternion: w^2 - z^2, 2 w z)
};

// ---------------------------------------------------------------------------
// bit casts / NaN-propagating max (jnp.maximum)
// ---------------------------------------------------------------------------
GX_HD float u2f(uint32_t u) { union { uint32_t u; float f; } v; v.u = u; return v.f; }
GX_HD uint32_t f2u(float f) { union { uint32_t u; float f; } v; v.f = f; return v.u; }
GX_D float nmax(float a, float b) { return (a > b || a != a) ? a : b; }
GX_D bool notfinite(float v) { return !(fabsf(v) <= 3.4028234663852886e38f); }

// pose (x, y, cos, sin) of the robot body in its root body's frame -> world frame (robot_rot, see Params)
GX_D void world_pose(const Params& p, float (&pose)[4])
{
    if (p.rot_on) {
        const float x = pose[0], y = pose[1], c = pose[2], s = pose[3];
        pose[0] = p.rot_c * x - p.rot_s * y;
        pose[1] = p.rot_s * x + p.rot_c * y;
        pose[2] = p.rot_c * c - p.rot_s * s;
        pose[3] = p.rot_s * c + p.rot_c * s;
    }
}

// ---------------------------------------------------------------------------
// sin/cos: 3-term Cody-Waite to [-pi/4, pi/4], minimax polynomials
// ---------------------------------------------------------------------------
// kFinite: the caller guarantees |x| <= 2^24 (so x is not NaN either): the same bits without the two guards -- five
// instructions, two of them on the dependency chain (the dynamics pass of the two-kernel rollout, which verifies the
// guarantee after the step and redoes the step with the guarded form otherwise)
template <bool kFinite = false>
GX_D void sincos_f(float x, float& s, float& c)
{
    const float x0 = x;
    if (!kFinite && !(fabsf(x) <= 16777216.0f)) x = x * 0.0f;
    const float k = rintf(x * 0.6366197466850281f);
    float r = fmaf(-k, 1.5707963705062866f, x);
    r = fmaf(-k, -4.371138828673793e-08f, r);
    r = fmaf(-k, -1.7151245100058819e-15f, r);
    const float z = r * r;
    float ps = fmaf(z, -0.00019488747f, 0.008331924f);
    ps = fmaf(z, ps, -0.1666665f);
    const float S = fmaf(r * z, ps, r);
    float pc = fmaf(z, 2.4431205e-05f, -0.0013887306f);
    pc = fmaf(z, pc, 0.041666646f);
    const float C = fmaf(z * z, pc, fmaf(z, -0.5f, 1.0f));
    const int q = ((int)k) & 3;
    float ss = (q & 1) ? C : S;
    float cc = (q & 1) ? S : C;
    if (q == 1 || q == 2) cc = -cc;
    if (q >= 2) ss = -ss;
    if (!kFinite && x0 != x0) { ss = x0; cc = x0; }
    s = ss;
    c = cc;
}

// ---------------------------------------------------------------------------
// atan2: one division + degree-7 polynomial in a^2, a = min/max in [0,1]
// ---------------------------------------------------------------------------
GX_D float atan2_f(float y, float x)
{
    if (x != x || y != y) return x + y;
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = ax > ay ? ax : ay;
    const float mn = ax > ay ? ay : ax;
    float a;
    if (mx == 0.0f) a = 0.0f;
    else if (mx == __builtin_inff()) a = (mn == __builtin_inff()) ? 1.0f : 0.0f;
    else a = mn / mx;
    const float z = a * a;
    float p = fmaf(z, 0.0026222442f, -0.015132535f);
    p = fmaf(z, p, 0.04112186f);
    p = fmaf(z, p, -0.07366706f);
    p = fmaf(z, p, 0.10573932f);
    p = fmaf(z, p, -0.14185975f);
    p = fmaf(z, p, 0.19990396f);
    p = fmaf(z, p, -0.33332986f);
    float t = fmaf(a * z, p, a);
    if (ay > ax) t = 1.5707963705062866f - t;
    if (f2u(x) >> 31) t = 3.1415927410125732f - t;
    return (f2u(y) >> 31) ? -t : t;
}

// ---------------------------------------------------------------------------
// exp (lidar closeness).  Below exp(-87) the result is flushed to zero.
// ---------------------------------------------------------------------------
GX_D float exp_f(float x)
{
    if (x != x) return x;
    if (x < -87.0f) return 0.0f;
    if (x > 88.0f) return __builtin_inff();
    const float k = rintf(x * 1.4426950216293335f);
    float r = fmaf(-k, 0.6931471824645996f, x);
    r = fmaf(-k, -1.9046542121259336e-09f, r);
    float q = fmaf(r, 0.001395172f, 0.008369599f);
    q = fmaf(r, q, 0.041666187f);
    q = fmaf(r, q, 0.16666512f);
    q = fmaf(r, q, 0.5f);
    const float t = fmaf(r * r, q, r);
    const float e = 1.0f + t;
    const int ki = (int)k;
    return u2f(f2u(e) + ((uint32_t)ki << 23));
}

// ---------------------------------------------------------------------------
// natural log for x > 0 (normal floats): exponent split + atanh series
// ---------------------------------------------------------------------------
GX_D float log_f(float x)
{
    const uint32_t b = f2u(x);
    int e = (int)(b >> 23) - 127;
    float m = u2f((b & 0x7FFFFFu) | 0x3F800000u);
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
    const float f = m - 1.0f;
    const float s = f / (2.0f + f);
    const float z = s * s;
    float P = fmaf(z, 0.22222222f, 0.2857143f);
    P = fmaf(z, P, 0.4f);
    P = fmaf(z, P, 0.6666667f);
    const float lnm = fmaf(s * z, P, 2.0f * s);
    const float fe = (float)e;
    return fmaf(fe, 0.6931471824645996f, fmaf(fe, -1.9046542121259336e-09f, lnm));
}

// tanh through exp: sign(x) * (1 - 2 / (exp(2|x|) + 1)), saturating at |x| > 9.
// The same values as the plain form (the CPU checker's), with two things taken off the policy kernels' chains (64 tanh per
// lane and control step at hidden width 256):
//  * exp_f's guards: the argument is in [0, 18], never NaN, never beyond +-87;
//  * the IEEE division: d = exp(2|x|) + 1 lies in [2, 6.6e7], and for EVERY float d with 2^-126 <= |d| < 2^126 the short
//    sequence y0 = v_rcp_f32(d), y1 = fma(fma(-d, y0, 1), y0, y0) is the correctly rounded 1 / d (all 2^32 inputs against
//    the compiler's IEEE sequence, tools/probes/rcp_exact_probe.hip, profiles/r05_rcp_exact_probe.log); 2 / d = 2 * (1 / d)
//    exactly (a power of two commutes with rounding away from the subnormals), so fl(1 - 2 / d) = fma(-2, y1, 1).
// 3 + 1 instructions instead of the 11-instruction division and a subtraction.
GX_D float tanh_f(float x)
{
    const float ax = fabsf(x);
    const float a2 = ax > 9.0f ? 18.0f : 2.0f * ax;       // (keeps the core's argument in range on the lanes that saturate)
    const float k = rintf(a2 * 1.4426950216293335f);      // exp_f's core, verbatim
    float r = fmaf(-k, 0.6931471824645996f, a2);
    r = fmaf(-k, -1.9046542121259336e-09f, r);
    float q = fmaf(r, 0.001395172f, 0.008369599f);
    q = fmaf(r, q, 0.041666187f);
    q = fmaf(r, q, 0.16666512f);
    q = fmaf(r, q, 0.5f);
    const float e1 = 1.0f + fmaf(r * r, q, r);
    const float ex = u2f(f2u(e1) + ((uint32_t)(int)k << 23));
    const float d = ex + 1.0f;
    const float y0 = __builtin_amdgcn_rcpf(d);
    const float y1 = fmaf(fmaf(-d, y0, 1.0f), y0, y0);
    float t = fmaf(-2.0f, y1, 1.0f);
    if (!(ax <= 9.0f)) t = 1.0f;                           // saturated (and NaN, replaced below)
    t = (f2u(x) >> 31) ? -t : t;
    return x != x ? x : t;
}

// ---------------------------------------------------------------------------
// 1 / d for the pivots and 2x2 determinants of the Ant's / Walker's solves: the IEEE division WITHOUT its scaling steps.
// The compiler's sequence for 1.0f / d is v_div_scale x2, v_rcp, a Newton step on the reciprocal, a multiply and two
// residual corrections on the (scaled) quotient, v_div_fmas, v_div_fixup -- eleven dependent instructions on a chain that
// pays for every instruction it issues.  The scaling only matters when d or 1 / d leaves the normal range; without it,
//     y0 = v_rcp_f32(d);  y1 = fma(fma(-d, y0, 1), y0, y0);  v_div_fixup_f32(y1, d, 1.0f)
// has the bits of 1.0f / d for EVERY d that is +-0, +-inf, NaN (v_div_fixup decides those from d alone, as in the IEEE
// sequence) or a normal number with 2^-126 <= |d| <= 2^126 -- all 2^32 inputs checked against the compiler's sequence,
// tools/probes/rcp_exact_probe.hip variant A (profiles/r05_rcp_exact_probe.log); it differs only for denormal d and for
// |d| > 2^126 = 8.5e37.  The pivots it is used for cannot be there: each is a difference / sum of terms built from the
// model's constants (masses, inertias, armatures: 1e-4 ... 1e1), sines and cosines, and D J^2 terms with D <= 19 / invw
// -- a finite state puts them within 1e-7 ... 1e10, a cancellation leaves 0 (exact above) or a multiple of the operands'
// ulp (>= 1e-13), and a non-finite state makes them NaN (exact above).  -DGX_RCP_DOMAIN_CHECK turns any d outside the
// domain into a NaN result, which the parity soaks would report as mismatches (tests/soak_parity.py, soak_variants.py
// under such a variant build: profiles/r05_rcp_domain_soak.log: none).  Four instructions instead of eleven.
GX_D float rcp_unscaled(float d)
{
#ifdef GX_RCP_DOMAIN_CHECK
    const float ad = fabsf(d);
    if ((ad != 0.0f && ad < 1.17549435e-38f) || (ad > 8.50705917e37f && ad < __builtin_inff())) return __builtin_nanf("");
#endif
    const float y0 = __builtin_amdgcn_rcpf(d);
    const float y1 = fmaf(fmaf(-d, y0, 1.0f), y0, y0);
    return __builtin_amdgcn_div_fixupf(y1, d, 1.0f);
}

// ---------------------------------------------------------------------------
// jax.random on threefry2x32 (published algorithm: Salmon et al. 2011 /
// jax/_src/prng.py).  Used on host for the per-step key chain and on device
// for layout sampling and layout index draws.
// ---------------------------------------------------------------------------
GX_HD uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

GX_HD void threefry2x32(uint32_t k0, uint32_t k1, uint32_t x0, uint32_t x1,
                        uint32_t& o0, uint32_t& o1)
{
    const uint32_t k2 = k0 ^ k1 ^ 0x1BD11BDAu;
#define GX_R4(a, b, c, d)                                   \
    x0 += x1; x1 = rotl32(x1, a); x1 ^= x0;                 \
    x0 += x1; x1 = rotl32(x1, b); x1 ^= x0;                 \
    x0 += x1; x1 = rotl32(x1, c); x1 ^= x0;                 \
    x0 += x1; x1 = rotl32(x1, d); x1 ^= x0;
    x0 += k0; x1 += k1;
    GX_R4(13, 15, 26, 6)  x0 += k1; x1 += k2 + 1u;
    GX_R4(17, 29, 16, 24) x0 += k2; x1 += k0 + 2u;
    GX_R4(13, 15, 26, 6)  x0 += k0; x1 += k1 + 3u;
    GX_R4(17, 29, 16, 24) x0 += k1; x1 += k2 + 4u;
    GX_R4(13, 15, 26, 6)  x0 += k2; x1 += k0 + 5u;
#undef GX_R4
    o0 = x0;
    o1 = x1;
}

// element i of random_bits(key, (n,)): counts are padded to even length and cut
// into halves feeding the two block inputs.
GX_HD uint32_t tf_bits_at(uint32_t k0, uint32_t k1, uint32_t n, uint32_t i)
{
    const uint32_t half = (n + 1u) >> 1;
    uint32_t o0, o1;
    if (i < half) {
        const uint32_t hi = i + half;
        threefry2x32(k0, k1, i, hi < n ? hi : 0u, o0, o1);
        return o0;
    }
    threefry2x32(k0, k1, i - half, i, o0, o1);
    return o1;
}

// jax.random.split(key, 2): both children from two blocks.
GX_HD void split2(uint32_t k0, uint32_t k1, uint32_t& a0, uint32_t& a1, uint32_t& b0, uint32_t& b1)
{
    // flat = bits over iota(4): [B(0,2).0, B(1,3).0, B(0,2).1, B(1,3).1]
    threefry2x32(k0, k1, 0u, 2u, a0, b0);
    threefry2x32(k0, k1, 1u, 3u, a1, b1);
}

// jax.random.split(key, n)[j]
GX_HD void split_at(uint32_t k0, uint32_t k1, uint32_t n, uint32_t j, uint32_t& o0, uint32_t& o1)
{
    o0 = tf_bits_at(k0, k1, 2u * n, 2u * j);
    o1 = tf_bits_at(k0, k1, 2u * n, 2u * j + 1u);
}

// jax.random.uniform(key, (), f32, lo, hi)
GX_HD float uniform_f(uint32_t k0, uint32_t k1, float lo, float hi)
{
    uint32_t o0, o1;
    threefry2x32(k0, k1, 0u, 0u, o0, o1);
    const float f = u2f((o0 >> 9) | 0x3F800000u) - 1.0f;
    const float v = f * (hi - lo) + lo;
    return v > lo ? v : lo;
}

// jax.random.randint(key, (n,), 0, span)[i] given (k1,k2) = split(key)
GX_HD uint32_t randint_at(uint32_t k10, uint32_t k11, uint32_t k20, uint32_t k21,
                          uint32_t n, uint32_t span, uint32_t i)
{
    const uint32_t hi = tf_bits_at(k10, k11, n, i), lo = tf_bits_at(k20, k21, n, i);
    uint32_t mult = 65536u % span;
    mult = (mult * mult) % span;
    const uint32_t off = (hi % span) * mult + (lo % span);
    return off % span;
}

// ---------------------------------------------------------------------------
// pseudo-lidar (engine.py:846-900).  lidar_terms() evaluates one object:
// bin index (0..B; B = "angle rounded to 2*pi", whose own scatter is dropped),
// closeness `sensor` and the two aliased values a1 -> bin+1, a2 -> bin-1.
// ---------------------------------------------------------------------------
struct LidarTerms { int bin; float sensor, a1, a2; };

// x / bin_size for the default 16 bins WITHOUT the division (round 4).  bin_size = fl(2 pi / 16) = 0x1.921fb6p-2 is a
// constant; with inv = fl(1 / bin_size) = 0x1.45f306p+1
//     q0 = x * inv;   r = fma(-q0, bin_size, x);   q = fma(r, inv, q0)
// is the correctly rounded quotient -- bit for bit what the checker's `/` gives -- for EVERY fp32 x with
// 2^-100 <= |x| <= 2 pi (and for +0, NaN): three instructions instead of the compiler's ten-instruction IEEE division
// sequence, twice per lidar object.  That is not a heuristic: tests/test_div_bin_size.py (CPU, gcc) compares the two over
// all 1.1e9 such x, both signs.  Below 2^-100 the residual underflows and the identity fails for 0.4 % of the inputs
// (and -0 comes out +0): lidar_terms keeps the true division for any wave that holds such an angle.
constexpr int kDivFastBins = 16;
constexpr float kBinSize16 = 0x1.921fb6p-2f, kInvBinSize16 = 0x1.45f306p+1f;
constexpr uint32_t kDivFastMinBits = 0x0D800000u; // 2^-100
GX_D float div_bin16(float x)
{
    const float q0 = x * kInvBinSize16;
    return fmaf(fmaf(-q0, kBinSize16, x), kInvBinSize16, q0);
}

GX_D LidarTerms lidar_terms(const Params& p, float ox, float oy, const float (&pose)[4])
{
    const float dx = ox - pose[0], dy = oy - pose[1];
    const float zx = dx * pose[2] + dy * pose[3];
    const float zy = dx * (-pose[3]) + dy * pose[2];
    const float dist = sqrtf(zx * zx + zy * zy);
    float ang = atan2_f(zy, zx);
    if (ang < 0.0f) ang = ang + 6.2831854820251465f;
    const int B = p.bins;
    LidarTerms t;
    if (!p.lidar_max_dist_set) t.sensor = exp_f(p.neg_gain * dist);
    else t.sensor = nmax(0.0f, p.lidar_max_dist - dist) / p.lidar_max_dist;
    // every lane's angle is NaN or in [2^-100, 2 pi] (as a signed integer its pattern is then >= that of 2^-100; zeros,
    // -0 and anything smaller or negative compare below): both divisions by bin_size through div_bin16.  One
    // wave-uniform branch (the empty asm keeps the compiler from evaluating both sides and selecting).
    float q, alias;
    if (p.bins == kDivFastBins && __builtin_amdgcn_ballot_w64((int)f2u(ang) < (int)kDivFastMinBits) == 0ull) {
        q = div_bin16(ang);
        if (!(q >= 0.0f)) t.bin = 0;
        else if (q >= (float)B) t.bin = B;
        else t.bin = (int)q;
        // (ang - bin_angle is 0, or at least an ulp of an angle >= bin_size in magnitude, or the angle itself in bin 0)
        alias = div_bin16(ang - kBinSize16 * (float)t.bin);
    } else {
        asm volatile("; lidar_terms: true division");
        q = ang / p.bin_size;
        if (!(q >= 0.0f)) t.bin = 0;
        else if (q >= (float)B) t.bin = B;
        else t.bin = (int)q;
        const float bin_angle = p.bin_size * (float)t.bin;
        alias = (ang - bin_angle) / p.bin_size;
    }
    t.a1 = alias * t.sensor;
    t.a2 = (1.0f - alias) * t.sensor;
    return t;
}

// cost term of object o >= 1 (engine.py:804-811; pillars: the same dense form with pillars_size)
GX_D float cost_term(const Params& p, int o, float ox, float oy, const float (&pose)[4])
{
    const float size = (o <= p.H) ? p.hazards_size : p.pillars_size;
    const float dx = ox - pose[0], dy = oy - pose[1];
    const float dh = sqrtf(dx * dx + dy * dy);
    float below = dh < size ? dh : size;
    if (dh != dh) below = dh;
    return size - below;
}

GX_D int bin_plus(int bin, int B) { return (bin + 1 >= B) ? bin + 1 - B : bin + 1; }
GX_D int bin_minus(int bin, int B) { return (bin == 0) ? B - 1 : bin - 1; }

// values of `t` that would land in the observation are non-finite?
GX_D bool lidar_bad(const Params& p, const LidarTerms& t)
{
    bool bad = (t.bin < p.bins) && notfinite(t.sensor);
    if (p.lidar_alias) bad = bad || notfinite(t.a1) || notfinite(t.a2);
    return bad;
}

// thread-per-env form: scatter-max one object into a bin row that lives in LDS.
GX_D bool lidar_one(const Params& p, float* row, float ox, float oy, const float (&pose)[4])
{
    const LidarTerms t = lidar_terms(p, ox, oy, pose);
    const int B = p.bins;
    if (t.bin < B) row[t.bin] = nmax(row[t.bin], t.sensor);
    if (p.lidar_alias) {
        const int bp = bin_plus(t.bin, B), bm = bin_minus(t.bin, B);
        row[bp] = nmax(row[bp], t.a1);
        row[bm] = nmax(row[bm], t.a2);
    }
    return lidar_bad(p, t);
}

// lane-per-bin form: what object `t` contributes to bin b (0 when nothing).
GX_D float lidar_contrib(const Params& p, const LidarTerms& t, int b)
{
    const int B = p.bins;
    float c = 0.0f;
    if (t.bin == b) c = t.sensor; // t.bin == B never equals a valid b: dropped scatter
    if (p.lidar_alias) {
        if (bin_plus(t.bin, B) == b) c = t.a1;
        if (bin_minus(t.bin, B) == b) c = t.a2;
    }
    return c;
}

} // namespace gx
