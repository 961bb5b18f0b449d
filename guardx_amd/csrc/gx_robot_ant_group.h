// gx_robot_ant_group.h -- the Ant step (gx_robot_ant.h) evaluated by the 16 lanes that own one environment
// in the lane-group kernel: lane l works on leg (l & 3); the four lanes of a quad hold the four legs of the
// same env (the four quads of a group are replicas).  Per-leg arithmetic is the same as in the serial form,
// and everything that couples the legs -- the base accumulators, the Schur complement of the arrow solve,
// the base block of the Newton matrix, the base entries of the constraint force -- takes the four per-leg
// terms in leg order through DPP quad broadcasts, which is exactly the order the serial form (and the CPU
// checker) sums them in: the two forms agree bit for bit.  ~3x fewer instructions per lane, and the code is
// not unrolled over the legs, so it fits the instruction cache.
//
// Round 3: the four quads of a group are no longer pure replicas.  Lane l = 4 r + L (r = its quad, L = its leg) owns
// rows r and r + 4 of leg L's six constraint rows (r >= 2: row r only): it builds those rows, the products of the
// Newton matrix / right-hand side / constraint force that belong to them, and its share of the active-set test; the
// four partial sums of a leg meet in two butterfly exchanges across the quads (DPP row_ror:8, row_ror:4), which
// evaluates ((t0 + t4) + t2) + ((t1 + t5) + t3) (t_k = -0 for a row that does not take part) on every lane -- the canonical row order of AntRobot::rowsum and of
// the CPU checker, so the forms still agree bit for bit.  Per Newton iteration a lane handles 2 rows instead of 6.
#pragma once
#include "gx_robot_ant.h"

namespace gx {

struct AntGroup {
    using A = AntRobot;
    using Lim = AntRobot::Lim;
    using Foot = AntRobot::Foot;
    using Row = AntRobot::Row;

    // value held by lane K of this lane's quad (the quad = the four legs of one env)
    template <int K>
    GX_D static float quad(float x)
    {
        return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), K * 0x55, 0xf, 0xf, true));
    }
    template <int K>
    GX_D static int quadi(int x) { return __builtin_amdgcn_mov_dpp(x, K * 0x55, 0xf, 0xf, true); }
    GX_D static float sel4(int L, float a0, float a1, float a2, float a3)
    {
        return L == 0 ? a0 : (L == 1 ? a1 : (L == 2 ? a2 : a3));
    }
    // x + t0 + t1 + t2 + t3 with the per-leg terms taken in leg order
    GX_D static float add_legs(float x, float t)
    {
        x = x + quad<0>(t); x = x + quad<1>(t); x = x + quad<2>(t); x = x + quad<3>(t);
        return x;
    }
    GX_D static float sub_legs(float x, float t)
    {
        x = x - quad<0>(t); x = x - quad<1>(t); x = x - quad<2>(t); x = x - quad<3>(t);
        return x;
    }

    struct LegBlk { float C[3][2], Lhh, Lhb, Lbb; };

    // value held by the lane N places up (cyclically) in this lane's 16-lane row = the same leg in quad (r + N / 4) % 4
    template <int N>
    GX_D static float ror(float x)
    {
        return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x120 + N, 0xf, 0xf, true));
    }
    template <int N>
    GX_D static int rori(int x) { return __builtin_amdgcn_mov_dpp(x, 0x120 + N, 0xf, 0xf, true); }
    // (p0 + p2) + (p1 + p3) over the four quads' partial sums p_r of one leg, on every lane of the leg
    // (quad 0 / 2 form p0 + p2 / p2 + p0, quads 1 / 3 p1 + p3 / p3 + p1: the same bits; likewise the second stage)
    GX_D static float quads_sum(float p)
    {
        const float s1 = p + ror<8>(p);
        return s1 + ror<4>(s1);
    }
    GX_D static uint32_t quads_or(uint32_t m)
    {
        const uint32_t m1 = m | (uint32_t)rori<8>((int)m);
        return m1 | (uint32_t)rori<4>((int)m1);
    }
    // the rows this lane owns: A = row r (a joint-limit row for r < 2, pyramid row r - 2 otherwise), B = row r + 4
    // (pyramid rows 2, 3; quads 0 and 1 only; absent -- never present, never active -- for quads 2 and 3).  Same
    // operations as AntRobot::row_of, written with selects because r differs from lane to lane.
    struct MyRows { Row A, B; };
    // pj / paref: the lane's pyramid row (row r + 4 for r < 2, row r for r >= 2) and its aref, built once per step
    GX_D static MyRows my_rows(const Lim (&lim)[2], int foot_on, float foot_D, const float (&pj)[5], float paref, int r)
    {
        MyRows M;
        const bool islim = r < 2;
        const float lsg = (r & 1) ? lim[1].sg : lim[0].sg;
        const float laref = (r & 1) ? lim[1].aref : lim[0].aref;
        const float lD = (r & 1) ? lim[1].D : lim[0].D;
        M.A.present = islim ? (lsg != 0.0f) : (foot_on != 0);
        M.A.J[0] = islim ? 0.0f : pj[0];
        M.A.J[1] = islim ? 0.0f : pj[1];
        M.A.J[2] = islim ? 0.0f : pj[2];
        M.A.J[3] = islim ? ((r == 0) ? lsg : 0.0f) : pj[3];
        M.A.J[4] = islim ? ((r == 1) ? lsg : 0.0f) : pj[4];
        M.A.aref = islim ? laref : paref;
        M.A.D = islim ? lD : foot_D;
#ifdef GX_PROTO_ONE_ROW_PER_LANE
        // COUNTING PROTOTYPE, never shipped (tools/ab/proto_one_row.sh): what would the Newton loop cost if a lane owned ONE
        // row (the 32-lanes-per-env form)?  Row B is compiled out here -- wrong physics, right instruction count for the
        // per-lane row work; the extra butterfly stage such a form needs is not included (it only adds).
        M.B.present = false;
#else
        M.B.present = islim && (foot_on != 0);
#endif
#pragma unroll
        for (int k = 0; k < 5; ++k) M.B.J[k] = pj[k];
        M.B.aref = paref;
        M.B.D = foot_D;
        return M;
    }
    // this lane's share of a row sum, then the butterfly.  A row that does not take part contributes -0, the identity
    // of IEEE addition (x + (-0) = x for every x): tA + tB is tA for the lanes that own one row
    GX_D static float rows_sum(float tA, float tB) { return quads_sum(tA + tB); }

    GX_D static float dot5(const float* J, const float (&ab)[3], float ah, float abt)
    {
        return (((J[0] * ab[0] + J[1] * ab[1]) + J[2] * ab[2]) + J[3] * ah) + J[4] * abt;
    }
    // 6-bit active mask of this leg's rows: every lane tests the rows it owns, the quads OR their bits together
    GX_D static uint32_t active_leg(const MyRows& M, int r, const float (&ab)[3], float ah, float abt)
    {
        uint32_t m = 0;
        if (M.A.present && (dot5(M.A.J, ab, ah, abt) - M.A.aref < 0.0f)) m |= 1u << r;
#ifndef GX_PROTO_ONE_ROW_PER_LANE
        if (M.B.present && (dot5(M.B.J, ab, ah, abt) - M.B.aref < 0.0f)) m |= 1u << (r + 4);
#endif
        return quads_or(m);
    }
    GX_D static uint32_t gather_mask(uint32_t own)
    {
        return (uint32_t)quadi<0>((int)own) | ((uint32_t)quadi<1>((int)own) << 6) | ((uint32_t)quadi<2>((int)own) << 12) |
               ((uint32_t)quadi<3>((int)own) << 18);
    }

    // arrow solve: B = base block (lower triangle), K = this leg's blocks, rb = base right-hand side,
    // (rh, rbt) = this leg's right-hand side -> xb (base), (xh, xbt) (this leg)
    GX_D static void arrow_solve(const float (&B)[3][3], const LegBlk& K, const float (&rb)[3], float rh, float rbt,
                                 float (&xb)[3], float& xh, float& xbt)
    {
        const float det = K.Lhh * K.Lbb - K.Lhb * K.Lhb;
        const float rdet = rcp_unscaled(det);
        const float i00 = K.Lbb * rdet, i01 = -(K.Lhb * rdet), i11 = K.Lhh * rdet;
        float W[3][2];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            W[b][0] = K.C[b][0] * i00 + K.C[b][1] * i01;
            W[b][1] = K.C[b][0] * i01 + K.C[b][1] * i11;
        }
        float S[3][3], g[3];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            g[b] = sub_legs(rb[b], W[b][0] * rh + W[b][1] * rbt);
#pragma unroll
            for (int c = 0; c < 3; ++c)
                S[b][c] = (c <= b) ? sub_legs(B[b][c], W[b][0] * K.C[c][0] + W[b][1] * K.C[c][1]) : 0.0f;
        }
        A::Ldl3 F;
        A::ldl_factor(S, F);
        A::ldl_solve(F, g, xb);
        const float t0 = rh - ((K.C[0][0] * xb[0] + K.C[1][0] * xb[1]) + K.C[2][0] * xb[2]);
        const float t1 = rbt - ((K.C[0][1] * xb[0] + K.C[1][1] * xb[1]) + K.C[2][1] * xb[2]);
        xh = i00 * t0 + i01 * t1;
        xbt = i01 * t0 + i11 * t1;
    }

    // the lane-group kernels' ONE call site of the step (reset_done's fake step comes from Pool::fake): inlined
    __device__ __attribute__((always_inline)) static void substep_call(float* q, float* v, const float* ctrl, float* pose,
                                                                  float* qacc, int l16)
    {
        substep(*reinterpret_cast<float (*)[11]>(q), *reinterpret_cast<float (*)[11]>(v),
                *reinterpret_cast<const float (*)[8]>(ctrl), *reinterpret_cast<float (*)[4]>(pose),
                *reinterpret_cast<float (*)[11]>(qacc), l16);
    }

    // q, v, ctrl, qacc: the full arrays (every lane holds a copy, as in the serial form); l16 = lane & 15:
    // leg L = l16 & 3, quad r = l16 >> 2
    GX_D static void substep(float (&q)[11], float (&v)[11], const float (&ctrl)[8], float (&pose)[4], float (&qacc)[11],
                             int l16)
    {
        const int L = l16 & 3, rq = l16 >> 2;
        A::pose_of(q, pose);
        const float c = pose[2], s = pose[3];
        const float y = q[2], om = v[1], vy = v[2];
        const float wh = om * om;
        const float Ax = -(2.0f * (vy * om)), Ay = -(y * wh);
        // ---- this lane's leg
        const float dx = sel4(L, A::kD7, -A::kD7, -A::kD7, A::kD7);
        const float dy = sel4(L, A::kD7, A::kD7, -A::kD7, -A::kD7);
        const float sg = sel4(L, 1.0f, -1.0f, -1.0f, 1.0f);
        const float phi = sel4(L, q[3], q[5], q[7], q[9]);
        const float beta = sg * sel4(L, q[4], q[6], q[8], q[10]);
        const float dphi = sel4(L, v[3], v[5], v[7], v[9]);
        const float vpsi = sel4(L, v[4], v[6], v[8], v[10]);
        const float dbeta = sg * vpsi;
        const float uh = sel4(L, ctrl[0], ctrl[2], ctrl[4], ctrl[6]);
        const float ub = sel4(L, ctrl[1], ctrl[3], ctrl[5], ctrl[7]);
        float sp, cp, sb, cb;
        sincos_f(phi, sp, cp);
        sincos_f(beta, sb, cb);
        const float ex = cp * dx - sp * dy, ey = sp * dx + cp * dy;
        const float mx = -ey, my = ex;
        const float w = om + dphi, ww = w * w;
        const float hx = A::kA * dx, hy = A::kA * dy;
        const float r1x = hx + A::kA2 * ex, r1y = hy + A::kA2 * ey;
        const float a1x = (Ax - wh * hx) - ww * (A::kA2 * ex);
        const float a1y = (Ay - wh * hy) - ww * (A::kA2 * ey);
        const float lc = A::kA + A::kLC * cb;
        const float r2x = hx + lc * ex, r2y = hy + lc * ey;
        const float bw = dbeta * w, bb = dbeta * dbeta;
        const float um = -(2.0f * (sb * bw));
        const float ue = -(cb * ww + bb * cb);
        const float a2x = ((Ax - wh * hx) - ww * (A::kA * ex)) + A::kLC * (um * mx + ue * ex);
        const float a2y = ((Ay - wh * hy) - ww * (A::kA * ey)) + A::kLC * (um * my + ue * ey);
        const float a2z = A::kLC * (bb * sb);
        const float t1x = -(r1y + y), t1y = r1x;
        const float t2x = -(r2y + y), t2y = r2x;
        const float h1x = A::kA2 * mx, h1y = A::kA2 * my;
        const float h2x = lc * mx, h2y = lc * my;
        const float bx = -(A::kLC * (sb * ex)), by = -(A::kLC * (sb * ey)), bz = -(A::kLC * cb);
        const float rz = A::kITA + (A::kITK + A::kDIK * (sb * sb));
        const float tBtt = (A::kMA * (t1x * t1x + t1y * t1y) + A::kMK * (t2x * t2x + t2y * t2y)) + rz;
        const float tBxt = A::kMA * (c * t1x - s * t1y) + A::kMK * (c * t2x - s * t2y);
        const float tBty = A::kMA * t1y + A::kMK * t2y;
        LegBlk K;
        K.C[0][0] = A::kMA * (c * h1x - s * h1y) + A::kMK * (c * h2x - s * h2y);
        K.C[1][0] = (A::kMA * (t1x * h1x + t1y * h1y) + A::kMK * (t2x * h2x + t2y * h2y)) + rz;
        K.C[2][0] = A::kMA * h1y + A::kMK * h2y;
        K.C[0][1] = A::kMK * (c * bx - s * by);
        K.C[1][1] = A::kMK * (t2x * bx + t2y * by);
        K.C[2][1] = A::kMK * by;
        K.Lhh = ((A::kMA * (h1x * h1x + h1y * h1y) + A::kMK * (h2x * h2x + h2y * h2y)) + rz) + 1.0f;
        K.Lhb = 0.0f;
        K.Lbb = A::kLbb;
        const float nz = (2.0f * A::kDIK) * (bw * (sb * cb));
        const float tcx = A::kMA * (c * a1x - s * a1y) + A::kMK * (c * a2x - s * a2y);
        const float tcy = A::kMA * a1y + A::kMK * a2y;
        const float tct = (A::kMA * (t1x * a1x + t1y * a1y) + A::kMK * (t2x * a2x + t2y * a2y)) + nz;
        const float ch_ = (A::kMA * (h1x * a1x + h1y * a1y) + A::kMK * (h2x * a2x + h2y * a2y)) + nz;
        const float cb_ = A::kMK * ((bx * a2x + by * a2y) + bz * a2z) - A::kDIK * (ww * (sb * cb));
        const float fh = (-ch_ - dphi) + A::kGear * A::clip1(uh);
        const float fb = ((-cb_ - dbeta) + A::kGK * cb) + sg * (A::kGear * A::clip1(ub));
        Lim lim[2];
        Foot ft;
        A::limit_row(lim[0], phi, dphi, -A::kLim30, A::kLim30, A::kInvwHip);
        A::limit_row(lim[1], beta, dbeta, A::kLim30, A::kLim70, A::kInvwAnk);
        const float dist = (A::kZ0 - A::kL * sb) - A::kRf;
        const float pos = dist - A::kMargin;
        ft.on = 0; ft.jbz = 0.0f; ft.D = 0.0f;
        // the ONE pyramid row this lane owns (rows 4, 5 = T2 based for quads 0, 1; rows 2, 3 = T1 based for quads 2, 3;
        // sign by the parity of the quad) and its aref -- the same operations as AntRobot::row_of / substep_impl
        float pj[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, paref = 0.0f;
        if (pos < 0.0f) {
            const float zc = A::kRf + 0.5f * dist;
            const float lf = A::kA + A::kL * cb;
            const float pcx = hx + lf * ex, pcy = hy + lf * ey;
            const float jtx = -(pcy + y), jty = pcx;
            const float jhx = lf * mx, jhy = lf * my;
            const float kb = A::kL * sb + zc;
            const float jbx = -(kb * ex), jby = -(kb * ey), jbz = -(A::kL * cb);
            const float imp = A::impedance(pos);
            float rr = ((1.0f - imp) * A::kInvwPyr) / imp;
            if (rr < 1e-15f) rr = 1e-15f;
            ft.on = 1; ft.jbz = jbz; ft.D = 1.0f / rr;
            const float sgn = (rq & 1) ? -A::kMu : A::kMu;
            if (rq >= 2) {
                const float T1t = s * jtx + c * jty, T1h = s * jhx + c * jhy, T1b = s * jbx + c * jby;
                pj[0] = sgn * 0.0f; pj[1] = sgn * T1t; pj[2] = sgn * c; pj[3] = sgn * T1h; pj[4] = jbz + sgn * T1b;
            } else {
                const float T2t = c * jtx - s * jty, T2h = c * jhx - s * jhy, T2b = c * jbx - s * jby;
                pj[0] = sgn * 1.0f; pj[1] = sgn * T2t; pj[2] = sgn * (-s); pj[3] = sgn * T2h; pj[4] = jbz + sgn * T2b;
            }
            const float jv = (((pj[0] * v[0] + pj[1] * om) + pj[2] * vy) + pj[3] * dphi) + pj[4] * dbeta;
            paref = -(A::kB * jv) - (A::kK * imp) * pos;
        }
        const int own_any = (lim[0].sg != 0.0f) | (lim[1].sg != 0.0f) | ft.on;
        const int any_row = quadi<0>(own_any) | quadi<1>(own_any) | quadi<2>(own_any) | quadi<3>(own_any);

        // ---- base block and base smooth force: per-leg terms in leg order
        float B[3][3];
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) B[b][cc] = 0.0f;
        B[0][0] = A::kMtot; B[2][2] = A::kMtot; B[2][0] = -(s * A::kMtot);
        B[1][1] = add_legs(A::kMB * (y * y) + A::kIB, tBtt);
        B[1][0] = add_legs(-(A::kMB * (c * y)), tBxt);
        B[2][1] = add_legs(0.0f, tBty);
        const float cx = add_legs(A::kMB * (c * Ax - s * Ay), tcx);
        const float cy = add_legs(A::kMB * Ay, tcy);
        const float ct = add_legs(-(A::kMB * (y * Ax)), tct);
        float fbase[3];
        fbase[0] = -cx - 0.1f * v[0];
        fbase[1] = (-ct - 0.01f * om) - 0.1f * q[1];
        fbase[2] = -cy - 0.1f * vy;

        // ---- unconstrained acceleration, then the active-set Newton iterations
        float ab[3], ah, abt;
        arrow_solve(B, K, fbase, fh, fb, ab, ah, abt);
        float fcb[3] = {fbase[0], fbase[1], fbase[2]};
        float fch = fh, fcbt = fb;
        if (any_row) {
            const MyRows MR = my_rows(lim, ft.on, ft.D, pj, paref, rq);
            const Row& RA = MR.A;
            const Row& RB = MR.B;
            uint32_t act = gather_mask(active_leg(MR, rq, ab, ah, abt));
            for (int it = 0; it < A::kIters; ++it) {
                const uint32_t own = (act >> (6 * L)) & 63u;
#ifdef GX_PROTO_ONE_ROW_PER_LANE
                const bool onA = (own >> rq) & 1u, onB = false;
#else
                const bool onA = (own >> rq) & 1u, onB = (own >> (rq + 4)) & 1u; // (bit r + 4 is never set for r >= 2)
#endif
                // this lane's rows' products (+0 for a row outside the active set), in the order of AntRobot::newton_solve
                const float DA = onA ? RA.D : 0.0f, DB = onB ? RB.D : 0.0f;
                const float daA = DA * RA.aref, daB = DB * RB.aref;
                LegBlk Hk = K;
                float PB[3][3], Pr[3];
                float djA[3], djB[3];
#pragma unroll
                for (int b = 0; b < 3; ++b) { djA[b] = DA * RA.J[b]; djB[b] = DB * RB.J[b]; }
#pragma unroll
                for (int b = 0; b < 3; ++b) {
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc)
                        PB[b][cc] = (cc <= b) ? rows_sum(onA ? djA[b] * RA.J[cc] : -0.0f, onB ? djB[b] * RB.J[cc] : -0.0f) : 0.0f;
                    Hk.C[b][0] = Hk.C[b][0] + rows_sum(onA ? djA[b] * RA.J[3] : -0.0f, onB ? djB[b] * RB.J[3] : -0.0f);
                    Hk.C[b][1] = Hk.C[b][1] + rows_sum(onA ? djA[b] * RA.J[4] : -0.0f, onB ? djB[b] * RB.J[4] : -0.0f);
                    Pr[b] = rows_sum(onA ? daA * RA.J[b] : -0.0f, onB ? daB * RB.J[b] : -0.0f);
                }
                const float d3A = DA * RA.J[3], d4A = DA * RA.J[4], d3B = DB * RB.J[3], d4B = DB * RB.J[4];
                Hk.Lhh = Hk.Lhh + rows_sum(onA ? d3A * RA.J[3] : -0.0f, onB ? d3B * RB.J[3] : -0.0f);
                Hk.Lhb = Hk.Lhb + rows_sum(onA ? d3A * RA.J[4] : -0.0f, onB ? d3B * RB.J[4] : -0.0f);
                Hk.Lbb = Hk.Lbb + rows_sum(onA ? d4A * RA.J[4] : -0.0f, onB ? d4B * RB.J[4] : -0.0f);
                const float rh = fh + rows_sum(onA ? daA * RA.J[3] : -0.0f, onB ? daB * RB.J[3] : -0.0f);
                const float rbt = fb + rows_sum(onA ? daA * RA.J[4] : -0.0f, onB ? daB * RB.J[4] : -0.0f);
                float HB[3][3], rb[3];
#pragma unroll
                for (int b = 0; b < 3; ++b) {
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) HB[b][cc] = (cc <= b) ? add_legs(B[b][cc], PB[b][cc]) : 0.0f;
                    rb[b] = add_legs(fbase[b], Pr[b]);
                }
                arrow_solve(HB, Hk, rb, rh, rbt, ab, ah, abt);
                const uint32_t nact = gather_mask(active_leg(MR, rq, ab, ah, abt));
                if (nact == act) break;
                act = nact;
            }
            // constraint force of the rows violated at the solution
            float fA = 0.0f, fB = 0.0f;
            bool vA = false, vB = false;
            if (RA.present) { const float res = dot5(RA.J, ab, ah, abt) - RA.aref; if (res < 0.0f) { vA = true; fA = RA.D * (-res); } }
#ifndef GX_PROTO_ONE_ROW_PER_LANE
            if (RB.present) { const float res = dot5(RB.J, ab, ah, abt) - RB.aref; if (res < 0.0f) { vB = true; fB = RB.D * (-res); } }
#endif
            float Pf[3];
#pragma unroll
            for (int b = 0; b < 3; ++b) Pf[b] = rows_sum(vA ? fA * RA.J[b] : -0.0f, vB ? fB * RB.J[b] : -0.0f);
            fch = fch + rows_sum(vA ? fA * RA.J[3] : -0.0f, vB ? fB * RB.J[3] : -0.0f);
            fcbt = fcbt + rows_sum(vA ? fA * RA.J[4] : -0.0f, vB ? fB * RB.J[4] : -0.0f);
#pragma unroll
            for (int b = 0; b < 3; ++b) fcb[b] = add_legs(fcb[b], Pf[b]);
        }
        // ---- Euler with implicit joint damping
        float Bd[3][3];
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) Bd[b][cc] = B[b][cc];
        Bd[0][0] = Bd[0][0] + A::kH * 0.1f;
        Bd[1][1] = Bd[1][1] + A::kH * 0.01f;
        Bd[2][2] = Bd[2][2] + A::kH * 0.1f;
        LegBlk Kd = K;
        Kd.Lhh = Kd.Lhh + A::kH; Kd.Lbb = Kd.Lbb + A::kH;
        float aib[3], aih, aibt;
        arrow_solve(Bd, Kd, fcb, fch, fcbt, aib, aih, aibt);
        // this leg's outputs, then every lane rebuilds the full arrays
        const float qa_h = ah, qa_p = sg * abt;
        const float vh2 = dphi + A::kH * aih;
        const float vp2 = vpsi + A::kH * (sg * aibt);
        const float qh2 = phi + A::kH * vh2;
        const float qp2 = sel4(L, q[4], q[6], q[8], q[10]) + A::kH * vp2;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            qacc[k] = ab[k];
            v[k] = v[k] + A::kH * aib[k];
            q[k] = q[k] + A::kH * v[k];
        }
        qacc[3] = quad<0>(qa_h); qacc[4] = quad<0>(qa_p); qacc[5] = quad<1>(qa_h); qacc[6] = quad<1>(qa_p);
        qacc[7] = quad<2>(qa_h); qacc[8] = quad<2>(qa_p); qacc[9] = quad<3>(qa_h); qacc[10] = quad<3>(qa_p);
        v[3] = quad<0>(vh2); v[4] = quad<0>(vp2); v[5] = quad<1>(vh2); v[6] = quad<1>(vp2);
        v[7] = quad<2>(vh2); v[8] = quad<2>(vp2); v[9] = quad<3>(vh2); v[10] = quad<3>(vp2);
        q[3] = quad<0>(qh2); q[4] = quad<0>(qp2); q[5] = quad<1>(qh2); q[6] = quad<1>(qp2);
        q[7] = quad<2>(qh2); q[8] = quad<2>(qp2); q[9] = quad<3>(qh2); q[10] = quad<3>(qp2);
    }
};

} // namespace gx
