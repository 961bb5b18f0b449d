// gx_robot_ant.h -- Ant (xmls/ant.xml, model "ant_tiny") dynamics: one mjx.step.
//
// [derived] MuJoCo/MJX semantics; constants from tools/model_constants.py.
//  * base joints in the order slide-x, hinge-z (stiffness .1), slide-y (:17-19): the y slide follows
//    the hinge, so it moves along the BODY y axis -- qpos = (x, th, y, hip1, ankle1, ..., hip4, ankle4)
//  * four legs of (hip hinge about z, +-30 deg; ankle hinge about a horizontal axis, 30..70 deg up to
//    the sign of the axis) (:20-85), joint armature 1 and damping 1 (:5), density 5 (:6)
//  * contacts: the four foot spheres (r = .02, the only geoms with contype) against the floor plane,
//    margin .01, friction .75, condim 3 -> 4 pyramid rows per foot; joint-limit rows as for the swimmer;
//    solref (.02, 1) with refsafe (timeconst = 2h = .18), solimp (.9, .95, .001, .5, 2)
//  * motors gear 70, ctrl clamped to +-1 for the force (:7,137-146); h = .09 (:2)
//
// Internally the ankle coordinate is beta = sigma * ankle (sigma = +,-,-,+): every leg pitches down for
// beta > 0 and every ankle range is [30, 70] deg.  Vectors are expressed in the torso frame.  The mass
// matrix is an arrow (3x3 base block, a 2x2 block per leg, 3x2 couplings); contact rows of foot i touch
// (x, th, y, hip_i, beta_i) only and limit rows one leg DOF, so M + J'DJ keeps the arrow and is solved
// by a Schur complement on the base.  The constraint problem
//     min 1/2 (a - a0)' M (a - a0) + sum_r 1/2 D_r min(0, J_r a - aref_r)^2
// is solved by active-set Newton iterations (at most kIters; stops when the active set repeats).
// fp32, one IEEE operation per operator, same operation order as the CPU checker.
#pragma once
#include "gx_device.h"

namespace gx {

struct AntRobot {
    static constexpr int kId = 2, NQ = 11, NV = 11, NU = 8, NA = 8, NDYN = 7;
    static constexpr float kH = 0.09f;
    // default Goal_Ant_8Hazards observation: ctrl[0:8] compass[8:10] glidar[10:26] hlidar[26:42] qpos[42:53] qvel[53:64]
    static constexpr int kD = 64, kOffCtrl = 0, kOffComp = 8, kOffGl = 10, kOffHl = 26, kOffQpos = 42, kOffQvel = 53;
    static constexpr int kIters = 12, kRows = 6;

    // dyn: q[0:11] v[0:11] pose0[0:4] done0 steps  (28 floats = 7 float4 planes)
    GX_D static void load(const float4* __restrict__ dyn, int Npad, int i, float (&q)[NQ], float (&v)[NV],
                          float (&pose0)[4], float& done0, float& steps)
    {
        float f[28];
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const float4 d = dyn[(size_t)k * Npad + i];
            f[4 * k] = d.x; f[4 * k + 1] = d.y; f[4 * k + 2] = d.z; f[4 * k + 3] = d.w;
        }
#pragma unroll
        for (int k = 0; k < 11; ++k) { q[k] = f[k]; v[k] = f[11 + k]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) pose0[k] = f[22 + k];
        done0 = f[26]; steps = f[27];
    }
    GX_D static void store(float4* __restrict__ dyn, int Npad, int i, const float (&q)[NQ], const float (&v)[NV],
                           const float (&pose0)[4], float done0, float steps)
    {
        float f[28];
#pragma unroll
        for (int k = 0; k < 11; ++k) { f[k] = q[k]; f[11 + k] = v[k]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) f[22 + k] = pose0[k];
        f[26] = done0; f[27] = steps;
#pragma unroll
        for (int k = 0; k < 7; ++k) dyn[(size_t)k * Npad + i] = make_float4(f[4 * k], f[4 * k + 1], f[4 * k + 2], f[4 * k + 3]);
    }
    GX_D static void convert_action(const float (&)[4], const float (&a)[NA], float (&ctrl)[NU])
    {
#pragma unroll
        for (int k = 0; k < NU; ++k) ctrl[k] = a[k]; // non-point robots: the action is the ctrl (:673)
    }
    // layout2qpos (:635-638): the robot_x / robot_y joints by name -> qpos[0], qpos[2]
    // at qpos0 the ankles sit outside their range, so the "fake step" of reset_done (:719-724) moves them
    static constexpr bool kRestFixed = false;
    GX_D static void place(float (&q)[NQ], float rx, float ry) { q[0] = rx; q[2] = ry; }

    // ---- constants
    static constexpr float kA = 0.070710678118654766f, kA2 = 0.035355339059327383f, kL = 0.14142135623730953f;
    static constexpr float kRf = 0.02f, kZ0 = 0.15f, kMargin = 0.01f, kMu = 0.75f;
    static constexpr float kMB = 0.0069712530291984719f, kIB = 1.1792223548992135e-05f;
    static constexpr float kMA = 0.00061183990200729241f, kITA = 5.5465437811796888e-07f;
    static constexpr float kMK = 0.0012236798040145848f, kLC = 0.080392694440424323f;
    static constexpr float kITK = 3.3619570452168863e-06f, kDIK = -3.1306252130692852e-06f;
    static constexpr float kMtot = 0.01431333185328598f, kLbb = 1.0000112705816542f;
    static constexpr float kInvwHip = 0.99997789909644164f, kInvwAnk = 0.99998872954537044f;
    static constexpr float kInvwPyr = 179.33960549370181f;
    static constexpr float kK = 34.198556820902162f, kB = 11.695906432748538f;
    static constexpr float kLim30 = 0.52359877559829882f, kLim70 = 1.2217304763960306f, kGear = 70.0f;
    static constexpr float kD7 = 0.70710678118654757f;
    static constexpr float kGK = 0.00096505793162098648f; // MK * 9.81 * LC: gravity torque of the ankle link per cos(beta)

    struct Arrow {
        float B[3][3];    // base block (x, th, y), lower triangle used
        float C[4][3][2]; // base x (hip, beta) per leg
        float Lhh[4], Lhb[4], Lbb[4];
    };
    // constraint rows of one leg, stored compactly (registers): two joint-limit rows (sg = 0: absent) and
    // the foot contact (four pyramid edges  Jn +- mu T1, Jn +- mu T2 built from the two tangent rows)
    struct Lim { float sg, aref, D; };
    struct Foot {
        int on;
        float T1[3], T2[3]; // theta, hip, beta entries of the tangent rows (x and y entries are 0/1 and c/-s)
        float jbz, D, aref[4];
    };
    struct Rows {
        Lim lim[4][2];
        Foot foot[4];
        float c, s;
    };
    struct Row {
        int present;
        float J[5]; // over (x, th, y, hip_l, beta_l)
        float aref, D;
    };
    // Opaque identity on the compact rows.  Every product D J J' of the Newton matrix is invariant across
    // the solver iterations (only the active mask changes), so the optimiser would hoist all ~500 of them
    // out of the loop and spill; rebuilding them per iteration from 72 registers is far cheaper.
    GX_D static void keep_compact(Rows& rs)
    {
#pragma unroll
        for (int l = 0; l < 4; ++l) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
                asm volatile("" : "+v"(rs.lim[l][k].sg), "+v"(rs.lim[l][k].aref), "+v"(rs.lim[l][k].D));
            Foot& ft = rs.foot[l];
            asm volatile("" : "+v"(ft.T1[0]), "+v"(ft.T1[1]), "+v"(ft.T1[2]), "+v"(ft.T2[0]), "+v"(ft.T2[1]), "+v"(ft.T2[2]));
            asm volatile("" : "+v"(ft.jbz), "+v"(ft.D), "+v"(ft.aref[0]), "+v"(ft.aref[1]), "+v"(ft.aref[2]), "+v"(ft.aref[3]));
        }
    }
    // the full row k of leg l, rebuilt with the operations that defined it
    GX_D static Row row_of(const Rows& rs, int l, int k)
    {
        Row R;
        if (k < 2) {
            const Lim& m = rs.lim[l][k];
            R.present = m.sg != 0.0f;
            R.J[0] = 0.0f; R.J[1] = 0.0f; R.J[2] = 0.0f;
            R.J[3] = (k == 0) ? m.sg : 0.0f;
            R.J[4] = (k == 1) ? m.sg : 0.0f;
            R.aref = m.aref; R.D = m.D;
        } else {
            const Foot& ft = rs.foot[l];
            const int kk = k - 2;
            const float sgn = (kk & 1) ? -kMu : kMu;
            R.present = ft.on;
            if (kk < 2) {
                R.J[0] = sgn * 0.0f; R.J[1] = sgn * ft.T1[0]; R.J[2] = sgn * rs.c; R.J[3] = sgn * ft.T1[1];
                R.J[4] = ft.jbz + sgn * ft.T1[2];
            } else {
                R.J[0] = sgn * 1.0f; R.J[1] = sgn * ft.T2[0]; R.J[2] = sgn * (-rs.s); R.J[3] = sgn * ft.T2[1];
                R.J[4] = ft.jbz + sgn * ft.T2[2];
            }
            R.aref = ft.aref[kk]; R.D = ft.D;
        }
        return R;
    }
    struct Ldl3 { float rd0, rd1, rd2, l10, l20, l21; };

    GX_D static void ldl_factor(const float (&S)[3][3], Ldl3& f)
    {
        f.rd0 = rcp_unscaled(S[0][0]);
        f.l10 = S[1][0] * f.rd0;
        f.l20 = S[2][0] * f.rd0;
        const float d1 = S[1][1] - f.l10 * S[1][0];
        f.rd1 = rcp_unscaled(d1);
        const float t21 = S[2][1] - f.l20 * S[1][0];
        f.l21 = t21 * f.rd1;
        const float d2 = (S[2][2] - f.l20 * S[2][0]) - f.l21 * t21;
        f.rd2 = rcp_unscaled(d2);
    }
    GX_D static void ldl_solve(const Ldl3& f, const float (&b)[3], float (&x)[3])
    {
        const float y0 = b[0];
        const float y1 = b[1] - f.l10 * y0;
        const float y2 = (b[2] - f.l20 * y0) - f.l21 * y1;
        const float z2 = y2 * f.rd2;
        const float z1 = y1 * f.rd1 - f.l21 * z2;
        const float z0 = (y0 * f.rd0 - f.l10 * z1) - f.l20 * z2;
        x[0] = z0; x[1] = z1; x[2] = z2;
    }

    // unknowns ordered (x, th, y, hip1, beta1, ..., hip4, beta4)
    GX_D static void arrow_solve(const Arrow& A, const float* r, float* x)
    {
        float S[3][3], g[3] = {r[0], r[1], r[2]};
        float i00[4], i01[4], i11[4];
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int c = 0; c < 3; ++c) S[b][c] = (c <= b) ? A.B[b][c] : 0.0f;
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const float det = A.Lhh[l] * A.Lbb[l] - A.Lhb[l] * A.Lhb[l];
            const float rdet = rcp_unscaled(det);
            i00[l] = A.Lbb[l] * rdet; i01[l] = -(A.Lhb[l] * rdet); i11[l] = A.Lhh[l] * rdet;
            float W[3][2];
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                W[b][0] = A.C[l][b][0] * i00[l] + A.C[l][b][1] * i01[l];
                W[b][1] = A.C[l][b][0] * i01[l] + A.C[l][b][1] * i11[l];
            }
            const float r0 = r[3 + 2 * l], r1 = r[4 + 2 * l];
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                g[b] = g[b] - (W[b][0] * r0 + W[b][1] * r1);
#pragma unroll
                for (int c = 0; c <= b; ++c) S[b][c] = S[b][c] - (W[b][0] * A.C[l][c][0] + W[b][1] * A.C[l][c][1]);
            }
        }
        Ldl3 F;
        ldl_factor(S, F);
        float xb[3];
        ldl_solve(F, g, xb);
        x[0] = xb[0]; x[1] = xb[1]; x[2] = xb[2];
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const float t0 = r[3 + 2 * l] - ((A.C[l][0][0] * xb[0] + A.C[l][1][0] * xb[1]) + A.C[l][2][0] * xb[2]);
            const float t1 = r[4 + 2 * l] - ((A.C[l][0][1] * xb[0] + A.C[l][1][1] * xb[1]) + A.C[l][2][1] * xb[2]);
            x[3 + 2 * l] = i00[l] * t0 + i01[l] * t1;
            x[4 + 2 * l] = i01[l] * t0 + i11[l] * t1;
        }
    }

    // impedance, solimp = (.9, .95, .001, .5, 2)
    GX_D static float impedance(float pos)
    {
        const float ix = fabsf(pos) / 0.001f;
        float iy;
        if (ix < 0.5f) iy = 2.0f * (ix * ix);
        else iy = 1.0f - 2.0f * ((1.0f - ix) * (1.0f - ix));
        float imp = 0.9f + iy * (0.95f - 0.9f);
        if (imp < 0.9f) imp = 0.9f;
        if (imp > 0.95f) imp = 0.95f;
        if (ix > 1.0f) imp = 0.95f;
        return imp;
    }
    GX_D static void limit_row(Lim& R, float qj, float vel, float lo, float hi, float invw)
    {
        const float dlo = qj - lo, dhi = hi - qj;
        const float pos = dlo < dhi ? dlo : dhi;
        const float sg = dlo < dhi ? 1.0f : -1.0f;
        R.sg = 0.0f; R.aref = 0.0f; R.D = 0.0f;
        if (!(pos < 0.0f)) return;
        const float imp = impedance(pos);
        R.sg = sg;
        R.aref = -(kB * (sg * vel)) - (kK * imp) * pos;
        float rr = ((1.0f - imp) * invw) / imp;
        if (rr < 1e-15f) rr = 1e-15f;
        R.D = 1.0f / rr;
    }
    GX_D static float dot5(const float* J, const float* a, int l)
    {
        return (((J[0] * a[0] + J[1] * a[1]) + J[2] * a[2]) + J[3] * a[3 + 2 * l]) + J[4] * a[4 + 2 * l];
    }
    GX_D static uint32_t active_set(const Rows& rows, const float* a)
    {
        uint32_t m = 0;
#pragma unroll
        for (int l = 0; l < 4; ++l)
#pragma unroll
            for (int k = 0; k < kRows; ++k) {
                const Row R = row_of(rows, l, k);
                if (R.present && (dot5(R.J, a, l) - R.aref < 0.0f)) m |= 1u << (l * kRows + k);
            }
        return m;
    }
    // CANONICAL ROW ORDER (round 3): whatever is summed over the six constraint rows of one leg is summed as
    //     ((t0 + t4) + t2) + ((t1 + t5) + t3),      t_k = -0 for a row that does not take part,
    // and then added to what it extends -- the order in which the four lanes that share a leg in the lane-group kernel
    // (gx_robot_ant_group.h: lane r owns rows r and r + 4) combine their partial sums with two butterfly exchanges.
    // -0 is the identity of IEEE addition for EVERY x (x + (-0) = x, signed zeros included), so a row that does not
    // take part can simply be skipped here: U = -0; U += t0, t4, t2 as they take part; V likewise; sum = U + V.
    static constexpr int kTerms = 20;
    // adds the twenty products row k of leg l contributes to the Newton system:
    // [0..5] base block (0,0) (1,0) (1,1) (2,0) (2,1) (2,2); [6..8] / [9..11] coupling columns hip / beta;
    // [12..14] base right-hand side; [15..17] Lhh, Lhb, Lbb; [18..19] leg right-hand side
    GX_D static void add_row_terms(const Rows& rows, int l, int k, uint32_t act, float (&t)[kTerms])
    {
        static_assert(kRows == 6, "canonical row order is written for six rows per leg");
        if (!((act >> (l * kRows + k)) & 1u)) return;
        const Row R = row_of(rows, l, k);
        const float da = R.D * R.aref;
        int e = 0;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const float dj = R.D * R.J[b];
#pragma unroll
            for (int c = 0; c <= b; ++c) { t[e] = t[e] + dj * R.J[c]; ++e; }
            t[6 + b] = t[6 + b] + dj * R.J[3];
            t[9 + b] = t[9 + b] + dj * R.J[4];
            t[12 + b] = t[12 + b] + da * R.J[b];
        }
        const float d3 = R.D * R.J[3], d4 = R.D * R.J[4];
        t[15] = t[15] + d3 * R.J[3]; t[16] = t[16] + d3 * R.J[4]; t[17] = t[17] + d4 * R.J[4];
        t[18] = t[18] + da * R.J[3]; t[19] = t[19] + da * R.J[4];
    }
    // minimiser of the quadratic piece selected by `act`: (M + J_A' D J_A) a = f + J_A' D aref_A
    // every leg first sums its own rows (canonical row order), then the base block and the base right-hand
    // side take the four leg sums in leg order -- the order the leg-parallel form (substep_group) reproduces
    GX_D static void newton_solve(const Arrow& M, const float* f, const Rows& rows, uint32_t act, float* a)
    {
        Arrow Hm = M;
        float r[11];
#pragma unroll
        for (int k = 0; k < 11; ++k) r[k] = f[k];
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            float U[kTerms], V[kTerms];
#pragma unroll
            for (int e = 0; e < kTerms; ++e) { U[e] = -0.0f; V[e] = -0.0f; }
            add_row_terms(rows, l, 0, act, U); add_row_terms(rows, l, 4, act, U); add_row_terms(rows, l, 2, act, U);
            add_row_terms(rows, l, 1, act, V); add_row_terms(rows, l, 5, act, V); add_row_terms(rows, l, 3, act, V);
#pragma unroll
            for (int e = 0; e < kTerms; ++e) U[e] = U[e] + V[e];
            int e = 0;
#pragma unroll
            for (int b = 0; b < 3; ++b) {
#pragma unroll
                for (int c = 0; c <= b; ++c) Hm.B[b][c] = Hm.B[b][c] + U[e++];
                Hm.C[l][b][0] = Hm.C[l][b][0] + U[6 + b];
                Hm.C[l][b][1] = Hm.C[l][b][1] + U[9 + b];
                r[b] = r[b] + U[12 + b];
            }
            Hm.Lhh[l] = Hm.Lhh[l] + U[15];
            Hm.Lhb[l] = Hm.Lhb[l] + U[16];
            Hm.Lbb[l] = Hm.Lbb[l] + U[17];
            r[3 + 2 * l] = r[3 + 2 * l] + U[18];
            r[4 + 2 * l] = r[4 + 2 * l] + U[19];
        }
        arrow_solve(Hm, r, a);
    }
    // adds the five entries of J' force of row k of leg l when the row is present and violated at `a`
    GX_D static void add_force_terms(const Rows& rows, int l, int k, const float* a, float (&t)[5])
    {
        const Row R = row_of(rows, l, k);
        if (!R.present) return;
        const float res = dot5(R.J, a, l) - R.aref;
        if (!(res < 0.0f)) return;
        const float frc = R.D * (-res);
#pragma unroll
        for (int e = 0; e < 5; ++e) t[e] = t[e] + frc * R.J[e];
    }

    // pose of the robot body (x, y, cos, sin) from qpos: the y slide acts along the rotated body axis
    GX_D static void pose_of(const float* q, float (&pose)[4])
    {
        float sh, ch;
        sincos_f(0.5f * q[1], sh, ch);
        const float c = ch * ch - sh * sh, s = 2.0f * (ch * sh);
        pose[0] = q[0] - s * q[2];
        pose[1] = c * q[2];
        pose[2] = c; pose[3] = s;
    }
    GX_D static float clip1(float u) { return u < -1.0f ? -1.0f : (u > 1.0f ? 1.0f : u); }

    // Inlined.  In round 1 it had to be a real call: inlined at TWO call sites per rollout kernel (the step and
    // reset_done's fake step, ~8k instructions each) hipcc 7.2 produced a kernel in which a value kept live across
    // the second copy came back wrong (caught by the parity tests).  The fake step now comes from Pool::fake, every
    // kernel has one call site, and the inlined step is 19 % faster than the call (thread-per-env rollout, N = 8192:
    // 49.8 -> 41.5 us per step).
    __device__ __attribute__((always_inline)) static void substep_impl(float* q, float* v, const float* ctrl, float (&pose)[4],
                                                            float* qacc)
    {
        const float kDx[4] = {kD7, -kD7, -kD7, kD7};
        const float kDy[4] = {kD7, kD7, -kD7, -kD7};
        const float kSg[4] = {1.0f, -1.0f, -1.0f, 1.0f};
        pose_of(q, pose);
        const float c = pose[2], s = pose[3];
        const float y = q[2], om = v[1], vy = v[2];
        const float wh = om * om;
        const float Ax = -(2.0f * (vy * om)), Ay = -(y * wh);
        Arrow M;
        float f[11];
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) M.B[b][cc] = 0.0f;
        M.B[0][0] = kMtot; M.B[2][2] = kMtot; M.B[2][0] = -(s * kMtot);
        float Btt = kMB * (y * y) + kIB;
        float Bxt = -(kMB * (c * y));
        float Bty = 0.0f;
        float cx = kMB * (c * Ax - s * Ay);
        float cy = kMB * Ay;
        float ct = -(kMB * (y * Ax));
        Rows rows;
        rows.c = c; rows.s = s;
        int any_row = 0;
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            const float dx = kDx[l], dy = kDy[l], sg = kSg[l];
            const float phi = q[3 + 2 * l], beta = sg * q[4 + 2 * l];
            const float dphi = v[3 + 2 * l], dbeta = sg * v[4 + 2 * l];
            float sp, cp, sb, cb;
            sincos_f(phi, sp, cp);
            sincos_f(beta, sb, cb);
            const float ex = cp * dx - sp * dy, ey = sp * dx + cp * dy;
            const float mx = -ey, my = ex;
            const float w = om + dphi, ww = w * w;
            const float hx = kA * dx, hy = kA * dy;
            const float r1x = hx + kA2 * ex, r1y = hy + kA2 * ey;
            const float a1x = (Ax - wh * hx) - ww * (kA2 * ex);
            const float a1y = (Ay - wh * hy) - ww * (kA2 * ey);
            const float lc = kA + kLC * cb;
            const float r2x = hx + lc * ex, r2y = hy + lc * ey;
            const float bw = dbeta * w, bb = dbeta * dbeta;
            const float um = -(2.0f * (sb * bw));
            const float ue = -(cb * ww + bb * cb);
            const float a2x = ((Ax - wh * hx) - ww * (kA * ex)) + kLC * (um * mx + ue * ex);
            const float a2y = ((Ay - wh * hy) - ww * (kA * ey)) + kLC * (um * my + ue * ey);
            const float a2z = kLC * (bb * sb);
            const float t1x = -(r1y + y), t1y = r1x;
            const float t2x = -(r2y + y), t2y = r2x;
            const float h1x = kA2 * mx, h1y = kA2 * my;
            const float h2x = lc * mx, h2y = lc * my;
            const float bx = -(kLC * (sb * ex)), by = -(kLC * (sb * ey)), bz = -(kLC * cb);
            const float rz = kITA + (kITK + kDIK * (sb * sb));
            Btt = Btt + ((kMA * (t1x * t1x + t1y * t1y) + kMK * (t2x * t2x + t2y * t2y)) + rz);
            Bxt = Bxt + (kMA * (c * t1x - s * t1y) + kMK * (c * t2x - s * t2y));
            Bty = Bty + (kMA * t1y + kMK * t2y);
            M.C[l][0][0] = kMA * (c * h1x - s * h1y) + kMK * (c * h2x - s * h2y);
            M.C[l][1][0] = (kMA * (t1x * h1x + t1y * h1y) + kMK * (t2x * h2x + t2y * h2y)) + rz;
            M.C[l][2][0] = kMA * h1y + kMK * h2y;
            M.C[l][0][1] = kMK * (c * bx - s * by);
            M.C[l][1][1] = kMK * (t2x * bx + t2y * by);
            M.C[l][2][1] = kMK * by;
            M.Lhh[l] = ((kMA * (h1x * h1x + h1y * h1y) + kMK * (h2x * h2x + h2y * h2y)) + rz) + 1.0f;
            M.Lhb[l] = 0.0f;
            M.Lbb[l] = kLbb;
            const float nz = (2.0f * kDIK) * (bw * (sb * cb));
            cx = cx + (kMA * (c * a1x - s * a1y) + kMK * (c * a2x - s * a2y));
            cy = cy + (kMA * a1y + kMK * a2y);
            ct = ct + ((kMA * (t1x * a1x + t1y * a1y) + kMK * (t2x * a2x + t2y * a2y)) + nz);
            const float ch_ = (kMA * (h1x * a1x + h1y * a1y) + kMK * (h2x * a2x + h2y * a2y)) + nz;
            const float cb_ = kMK * ((bx * a2x + by * a2y) + bz * a2z) - kDIK * (ww * (sb * cb));
            f[3 + 2 * l] = (-ch_ - dphi) + kGear * clip1(ctrl[2 * l]);
            // gravity (0, 0, -9.81) only has a generalized component on the ankle pitch
            f[4 + 2 * l] = ((-cb_ - dbeta) + kGK * cb) + sg * (kGear * clip1(ctrl[2 * l + 1]));
            limit_row(rows.lim[l][0], phi, dphi, -kLim30, kLim30, kInvwHip);
            limit_row(rows.lim[l][1], beta, dbeta, kLim30, kLim70, kInvwAnk);
            const float dist = (kZ0 - kL * sb) - kRf;
            const float pos = dist - kMargin;
            Foot& ft = rows.foot[l];
            ft.on = 0; ft.jbz = 0.0f; ft.D = 0.0f;
#pragma unroll
            for (int k = 0; k < 3; ++k) { ft.T1[k] = 0.0f; ft.T2[k] = 0.0f; }
#pragma unroll
            for (int k = 0; k < 4; ++k) ft.aref[k] = 0.0f;
            if (pos < 0.0f) {
                const float zc = kRf + 0.5f * dist;
                const float lf = kA + kL * cb;
                const float pcx = hx + lf * ex, pcy = hy + lf * ey;
                const float jtx = -(pcy + y), jty = pcx;
                const float jhx = lf * mx, jhy = lf * my;
                const float kb = kL * sb + zc;
                const float jbx = -(kb * ex), jby = -(kb * ey), jbz = -(kL * cb);
                const float T1[5] = {0.0f, s * jtx + c * jty, c, s * jhx + c * jhy, s * jbx + c * jby};
                const float T2[5] = {1.0f, c * jtx - s * jty, -s, c * jhx - s * jhy, c * jbx - s * jby};
                const float imp = impedance(pos);
                float rr = ((1.0f - imp) * kInvwPyr) / imp;
                if (rr < 1e-15f) rr = 1e-15f;
                const float Dc = 1.0f / rr;
                ft.on = 1; ft.jbz = jbz; ft.D = Dc;
                ft.T1[0] = T1[1]; ft.T1[1] = T1[3]; ft.T1[2] = T1[4];
                ft.T2[0] = T2[1]; ft.T2[1] = T2[3]; ft.T2[2] = T2[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const Row R = row_of(rows, l, 2 + k);
                    const float jv = (((R.J[0] * v[0] + R.J[1] * om) + R.J[2] * vy) + R.J[3] * dphi) + R.J[4] * dbeta;
                    ft.aref[k] = -(kB * jv) - (kK * imp) * pos;
                }
            }
            any_row |= (rows.lim[l][0].sg != 0.0f) | (rows.lim[l][1].sg != 0.0f) | ft.on;
        }
        M.B[1][1] = Btt; M.B[1][0] = Bxt; M.B[2][1] = Bty;
        f[0] = -cx - 0.1f * v[0];
        f[1] = (-ct - 0.01f * om) - 0.1f * q[1];
        f[2] = -cy - 0.1f * vy;
        float a[11];
        arrow_solve(M, f, a);
        float fc[11];
#pragma unroll
        for (int k = 0; k < 11; ++k) fc[k] = f[k];
        if (any_row) {
            uint32_t act = active_set(rows, a);
            for (int it = 0; it < kIters; ++it) {
                keep_compact(rows);
                newton_solve(M, f, rows, act, a);
                const uint32_t nact = active_set(rows, a);
                if (nact == act) break;
                act = nact;
            }
            keep_compact(rows);
#pragma unroll
            for (int l = 0; l < 4; ++l) { // rows in canonical order, legs in leg order
                float U[5] = {-0.0f, -0.0f, -0.0f, -0.0f, -0.0f}, V[5] = {-0.0f, -0.0f, -0.0f, -0.0f, -0.0f};
                add_force_terms(rows, l, 0, a, U); add_force_terms(rows, l, 4, a, U); add_force_terms(rows, l, 2, a, U);
                add_force_terms(rows, l, 1, a, V); add_force_terms(rows, l, 5, a, V); add_force_terms(rows, l, 3, a, V);
                fc[3 + 2 * l] = fc[3 + 2 * l] + (U[3] + V[3]);
                fc[4 + 2 * l] = fc[4 + 2 * l] + (U[4] + V[4]);
                fc[0] = fc[0] + (U[0] + V[0]); fc[1] = fc[1] + (U[1] + V[1]); fc[2] = fc[2] + (U[2] + V[2]);
            }
        }
        // Euler with implicit joint damping: (M + h diag(damping)) qacc_int = f + J' force
        Arrow Md = M;
        Md.B[0][0] = Md.B[0][0] + kH * 0.1f;
        Md.B[1][1] = Md.B[1][1] + kH * 0.01f;
        Md.B[2][2] = Md.B[2][2] + kH * 0.1f;
#pragma unroll
        for (int l = 0; l < 4; ++l) { Md.Lhh[l] = Md.Lhh[l] + kH; Md.Lbb[l] = Md.Lbb[l] + kH; }
        float ai[11];
        arrow_solve(Md, fc, ai);
#pragma unroll
        for (int k = 0; k < 3; ++k) qacc[k] = a[k];
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            qacc[3 + 2 * l] = a[3 + 2 * l];
            qacc[4 + 2 * l] = kSg[l] * a[4 + 2 * l];
            ai[4 + 2 * l] = kSg[l] * ai[4 + 2 * l];
        }
#pragma unroll
        for (int k = 0; k < 11; ++k) v[k] = v[k] + kH * ai[k];
#pragma unroll
        for (int k = 0; k < 11; ++k) q[k] = q[k] + kH * v[k];
    }

    template <bool kQacc>
    GX_D static void substep(float (&q)[NQ], float (&v)[NV], const float (&ctrl)[NU], float (&pose)[4],
                             float (&qacc)[NV])
    {
        substep_impl(q, v, ctrl, pose, qacc);
    }
};

} // namespace gx
