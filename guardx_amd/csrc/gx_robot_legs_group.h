// gx_robot_legs_group.h -- the Walker step (gx_robot_legs.h) evaluated leg-parallel by the 16 lanes that own
// one environment in the lane-group kernel: lane l works on leg (l & 1); lanes 0/1 of every quad hold the two
// legs of the env (lanes 2/3 and the other quads are replicas).  Per-leg arithmetic is that of the serial form
// (table entries selected by the leg index instead of being compile-time constants); every sum over the legs
// -- base accumulators, the Schur complement of the arrow solve, the base block of the Newton matrix, the base
// entries of the constraint force -- takes the two per-leg terms in leg order through DPP quad broadcasts, the
// order the serial form and the CPU checker use, so the forms agree bit for bit.
//
// Round 3: the eight lanes of a leg are no longer pure replicas.  Lane l = 2 r + L (r = 0..7, L = its leg) owns term
// r & 3 of everything that is summed over four: body r & 3 of the leg in the smooth dynamics (mass-matrix blocks, bias,
// gravity: three bodies, the fourth slot is empty) and pyramid row r & 3 of the foot in the constraint solve (its row,
// its aref, its products, its share of the active-set test).  The four terms meet in two butterfly exchanges (DPP
// row_ror:4, row_ror:2), which evaluates (t0 + t2) + (t1 + t3) with -0 for an absent term on every lane -- the
// canonical order of four of WalkerRobot / the CPU checker.  A lane handles one body instead of three and one
// pyramid row instead of four.
#pragma once
#include "gx_robot_legs.h"
#include "gx_robot_ant_group.h"

namespace gx {

struct WalkerGroup {
    using W = WalkerRobot;
    using V3 = WalkerRobot::V3;
    using M3 = WalkerRobot::M3;
    using Lim = WalkerRobot::Lim;
    static constexpr int K = WalkerRobot::kK, ND = WalkerRobot::ND;
    static_assert(WalkerRobot::kLegs == 2, "two legs: lanes 0/1 of a quad");
    static_assert(W::c_blink[0][0] == W::c_blink[1][0] && W::c_blink[0][1] == W::c_blink[1][1] &&
                  W::c_blink[0][2] == W::c_blink[1][2], "same link layout on both legs");

    template <int Q> GX_D static float quad(float x) { return AntGroup::quad<Q>(x); }
    template <int Q> GX_D static int quadi(int x) { return AntGroup::quadi<Q>(x); }
    GX_D static float add_legs(float x, float t) { x = x + quad<0>(t); x = x + quad<1>(t); return x; }
    GX_D static float sub_legs(float x, float t) { x = x - quad<0>(t); x = x - quad<1>(t); return x; }
    // (t0 + t2) + (t1 + t3) over the terms the leg's lanes own (lane r: term r & 3), on every lane of the leg
    GX_D static float sum4(float p)
    {
        const float s1 = p + AntGroup::ror<4>(p);
        return s1 + AntGroup::ror<2>(s1);
    }
    GX_D static uint32_t or4(uint32_t m)
    {
        const uint32_t m1 = m | (uint32_t)AntGroup::rori<4>((int)m);
        return m1 | (uint32_t)AntGroup::rori<2>((int)m1);
    }
    // table entry of this lane's leg
    GX_D static float ts(int L, float a0, float a1) { return L ? a1 : a0; }
    GX_D static float sel3(int j, float a2, float a3, float a4) { return j == 2 ? a2 : (j == 3 ? a3 : a4); }
    GX_D static V3 selv(int j, const V3& a2, const V3& a3, const V3& a4)
    {
        return W::lv(sel3(j, a2.x, a3.x, a4.x), sel3(j, a2.y, a3.y, a4.y), sel3(j, a2.z, a3.z, a4.z));
    }
#define GX_T2(tab, ...) ts(L, W::tab[0] __VA_ARGS__, W::tab[1] __VA_ARGS__)
    GX_D static float tb6(int L, int b, float a00, float a01, float a02, float a10, float a11, float a12)
    {
        const float x0 = L ? a10 : a00, x1 = L ? a11 : a01, x2 = L ? a12 : a02;
        return b == 0 ? x0 : (b == 1 ? x1 : x2);
    }
    // table entry of this lane's leg AND body (bq)
#define GX_TB(tab, ...)                                                                                     \
    tb6(L, bq, W::tab[0][0] __VA_ARGS__, W::tab[0][1] __VA_ARGS__, W::tab[0][2] __VA_ARGS__, W::tab[1][0] __VA_ARGS__, \
        W::tab[1][1] __VA_ARGS__, W::tab[1][2] __VA_ARGS__)

    struct LegBlk { float C[3][K], L[K][K]; };
    struct FootR { int on; float J[ND], aref, D; }; // J, aref: the ONE pyramid row this lane owns (row r & 3)

    // exact broadcast from the lane(s) of the leg that own term kq: everybody else contributes -0, the identity of
    // IEEE addition, so the two exchanges of sum4 hand x itself to every lane of the leg
    GX_D static float share(bool own, float x) { return sum4(own ? x : -0.0f); }

    // eliminate this leg: Wl = L^-1 [C_x, C_th, C_y, r_leg]; contributions tg, tS to the base system.
    // Every lane factors the 5x5 block (same bits everywhere); the FOUR right-hand sides are then solved by four
    // lanes -- lane kq takes column kq and the dot products that belong to it -- and shared.  Each value is computed
    // by exactly the operations of WalkerRobot::ldl_solve / arrow_solve, by one lane: no summation order changes.
    GX_D static void eliminate(const LegBlk& B, const float (&rl)[K], int kq, float (&Wl)[4][K], float (&tg)[3],
                               float (&tS)[3][3])
    {
        float l[K][K], u[K][K], rd[K];
#pragma unroll
        for (int i = 0; i < K; ++i) {
#pragma unroll
            for (int k = 0; k < i; ++k) {
                float s = B.L[i][k];
#pragma unroll
                for (int m = 0; m < k; ++m) s = s - u[i][m] * l[k][m];
                u[i][k] = s;
                l[i][k] = s * rd[k];
            }
            float s = B.L[i][i];
#pragma unroll
            for (int m = 0; m < i; ++m) s = s - u[i][m] * l[i][m];
            rd[i] = rcp_unscaled(s);
        }
        float x[K];
#pragma unroll
        for (int i = 0; i < K; ++i) x[i] = kq == 0 ? B.C[0][i] : (kq == 1 ? B.C[1][i] : (kq == 2 ? B.C[2][i] : rl[i]));
#pragma unroll
        for (int i = 0; i < K; ++i) {
            float s = x[i];
#pragma unroll
            for (int m = 0; m < i; ++m) s = s - l[i][m] * x[m];
            x[i] = s;
        }
#pragma unroll
        for (int i = 0; i < K; ++i) x[i] = x[i] * rd[i];
#pragma unroll
        for (int i = K - 1; i >= 0; --i) {
            float s = x[i];
#pragma unroll
            for (int m = i + 1; m < K; ++m) s = s - l[m][i] * x[m];
            x[i] = s;
        }
        float pb[3]; // C_b . (own column): tg[b] on the lane of column 3, tS[b][kq] on the others
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            float t = 0.0f;
#pragma unroll
            for (int i = 0; i < K; ++i) t = t + B.C[b][i] * x[i];
            pb[b] = t;
        }
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int i = 0; i < K; ++i) Wl[n][i] = share(kq == n, x[i]);
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            tg[b] = share(kq == 3, pb[b]);
#pragma unroll
            for (int c = 0; c < 3; ++c) tS[b][c] = (c <= b) ? share(kq == c, pb[b]) : 0.0f;
        }
    }
    // arrow solve: base block Bb, this leg's blocks, base rhs rb, this leg's rhs rl -> xb, xl
    GX_D static void arrow_solve(const float (&Bb)[3][3], const LegBlk& Blk, const float (&rb)[3], const float (&rl)[K],
                                 int kq, float (&xb)[3], float (&xl)[K])
    {
        float Wl[4][K], tg[3], tS[3][3];
        eliminate(Blk, rl, kq, Wl, tg, tS);
        float S[3][3], g[3];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            g[b] = sub_legs(rb[b], tg[b]);
#pragma unroll
            for (int c = 0; c < 3; ++c) S[b][c] = (c <= b) ? sub_legs(Bb[b][c], tS[b][c]) : 0.0f;
        }
        AntRobot::Ldl3 F;
        AntRobot::ldl_factor(S, F);
        AntRobot::ldl_solve(F, g, xb);
#pragma unroll
        for (int i = 0; i < K; ++i) xl[i] = Wl[3][i] - ((Wl[0][i] * xb[0] + Wl[1][i] * xb[1]) + Wl[2][i] * xb[2]);
    }
    GX_D static float row_dot(const float* J, const float (&ab)[3], const float (&al)[K])
    {
        float s = (J[0] * ab[0] + J[1] * ab[1]) + J[2] * ab[2];
#pragma unroll
        for (int i = 0; i < K; ++i) s = s + J[3 + i] * al[i];
        return s;
    }
    // this leg's 16-bit active mask (bit i: limit row of joint i; bit 8+k: pyramid row k): every lane tests the joint
    // limits and the pyramid row it owns (kq), the leg's lanes OR their bits together
    GX_D static uint32_t active_leg(const Lim (&lim)[K], const FootR& ft, int kq, const float (&ab)[3], const float (&al)[K])
    {
        uint32_t m = 0;
#pragma unroll
        for (int i = 0; i < K; ++i)
            if (lim[i].sg != 0.0f && (lim[i].sg * al[i] - lim[i].aref < 0.0f)) m |= 1u << i;
        if (ft.on && (row_dot(ft.J, ab, al) - ft.aref < 0.0f)) m |= 1u << (8 + kq);
        return or4(m);
    }
    GX_D static uint32_t gather_mask(uint32_t own)
    {
        return (uint32_t)quadi<0>((int)own) | ((uint32_t)quadi<1>((int)own) << 16);
    }

    // Inlined, like the Ant's (round 3).  Rounds 1-2 had to keep this a real call: with the 15k-instruction step of
    // that form inlined, hipcc 7.2 produced lane-group kernels whose dynamics blew up (test_variant_configs[group-walker],
    // -O3 and -O2, with or without IPRA), and as a callee it was exposed to LLVM's inter-procedural register allocation
    // using the VGPR lanes in which the caller parks spilled exec masks -- which came back in round 3 with ONE call
    // site as soon as the callee grew (the right-hand-side sharing of `eliminate`): wrong rows after an in-kernel
    // reset_done.  The round-3 step is a third shorter; inlined it passes every parity test and the soak
    // (profiles/r03_soak_walker.log) and costs no call ABI.  tests/test_native_abi.py checks that no lane-group kernel
    // contains a call, and the compiler version is part of the library's identity.
    __device__ __attribute__((always_inline)) static void substep_call(float* q, float* v, const float* ctrl, float* pose,
                                                                  float* qacc, int l16)
    {
        substep(*reinterpret_cast<float (*)[13]>(q), *reinterpret_cast<float (*)[13]>(v),
                *reinterpret_cast<const float (*)[10]>(ctrl), *reinterpret_cast<float (*)[4]>(pose),
                *reinterpret_cast<float (*)[13]>(qacc), l16);
    }

    // l16 = lane & 15: leg L = l16 & 1; the lane owns term kq = (l16 >> 1) & 3 of every sum over four
    GX_D static void substep(float (&q)[13], float (&v)[13], const float (&ctrl)[10], float (&pose)[4], float (&qacc)[13],
                             int l16)
    {
        const int L = l16 & 1, kq = (l16 >> 1) & 3;
        AntRobot::pose_of(q, pose);
        const float c = pose[2], s = pose[3];
        const float y = q[2], om = v[1], vy = v[2];
        const float wh = om * om;
        const V3 Pacc = W::lv(-(2.0f * (vy * om)), -(y * wh), 0.0f);
        const V3 ex = W::lv(c, -s, 0.0f);
        const V3 ez = W::lv(0.0f, 0.0f, 1.0f);
        // ---- this lane's leg: joint state
        float ql[K], vl[K], ul[K];
#pragma unroll
        for (int i = 0; i < K; ++i) {
            ql[i] = L ? q[3 + K + i] : q[3 + i];
            vl[i] = L ? v[3 + K + i] : v[3 + i];
            ul[i] = L ? ctrl[K + i] : ctrl[i];
        }
        M3 R = {W::lv(1.0f, 0.0f, 0.0f), W::lv(0.0f, 1.0f, 0.0f), W::lv(0.0f, 0.0f, 1.0f)};
        M3 Rj[K];
        V3 Aj[K], uj[K], wj[K], alj[K], aAj[K];
        V3 Aprev = W::lv(0.0f, 0.0f, 0.0f), w = W::lv(0.0f, 0.0f, om), al = W::lv(0.0f, 0.0f, 0.0f), aA = Pacc;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const V3 d = W::lmul(R, W::lv(GX_T2(c_dp, [j][0]), GX_T2(c_dp, [j][1]), GX_T2(c_dp, [j][2])));
            Aj[j] = W::ladd(Aprev, d);
            uj[j] = W::lmul(R, W::lv(GX_T2(c_axis, [j][0]), GX_T2(c_axis, [j][1]), GX_T2(c_axis, [j][2])));
            aAj[j] = W::ladd(W::ladd(aA, W::lcross(al, d)), W::lcross(w, W::lcross(w, d)));
            const float qd = vl[j];
            alj[j] = W::ladd(al, W::lscale(W::lcross(w, uj[j]), qd));
            wj[j] = W::ladd(w, W::lscale(uj[j], qd));
            float sj, cj;
            sincos_f(ql[j], sj, cj);
            R.c0 = W::lrot(uj[j], sj, cj, R.c0); R.c1 = W::lrot(uj[j], sj, cj, R.c1); R.c2 = W::lrot(uj[j], sj, cj, R.c2);
            Rj[j] = R;
            Aprev = Aj[j]; w = wj[j]; al = alj[j]; aA = aAj[j];
        }
        // ---- the ONE body this lane owns (body kq of the leg; kq == 3: none, every term is -0), then the sums of four
        static_assert(W::kNb == 3 && W::c_blink[0][0] == 2 && W::c_blink[0][1] == 3 && W::c_blink[0][2] == 4,
                      "bodies on links 2, 3, 4");
        const bool hasb = kq < W::kNb;
        const int bq = hasb ? kq : W::kNb - 1;
        const int j = 2 + bq; // the body's link
        float cl[K], gl[K];
        LegBlk Mk;
        float tBtt, tBxt, tBty, tcx, tcy, tct;
        {
            const float m = GX_TB(c_bm);
            const M3 Rb = {selv(j, Rj[2].c0, Rj[3].c0, Rj[4].c0), selv(j, Rj[2].c1, Rj[3].c1, Rj[4].c1),
                           selv(j, Rj[2].c2, Rj[3].c2, Rj[4].c2)};
            const V3 Ab = selv(j, Aj[2], Aj[3], Aj[4]), aAb = selv(j, aAj[2], aAj[3], aAj[4]);
            const V3 alb = selv(j, alj[2], alj[3], alj[4]), wb = selv(j, wj[2], wj[3], wj[4]);
            const V3 rr = W::lmul(Rb, W::lv(GX_TB(c_bc, [0]), GX_TB(c_bc, [1]), GX_TB(c_bc, [2])));
            const V3 X = W::ladd(Ab, rr);
            const V3 acom = W::ladd(W::ladd(aAb, W::lcross(alb, rr)), W::lcross(wb, W::lcross(wb, rr)));
            const V3 F = W::lscale(acom, m);
            float Ib[6];
#pragma unroll
            for (int e = 0; e < 6; ++e) Ib[e] = GX_TB(c_bI, [e]);
            const V3 Iw_w = W::lmul(Rb, W::lsym(Ib, W::lmulT(Rb, wb)));
            const V3 N = W::ladd(W::lmul(Rb, W::lsym(Ib, W::lmulT(Rb, alb))), W::lcross(wb, Iw_w));
            const V3 jt = W::lv(-(X.y + y), X.x, 0.0f);
            const V3 Iz = W::lmul(Rb, W::lsym(Ib, W::lmulT(Rb, ez)));
            tBtt = sum4(hasb ? (m * W::ldot(jt, jt) + Iz.z) : -0.0f);
            tBxt = sum4(hasb ? m * W::ldot(ex, jt) : -0.0f);
            tBty = sum4(hasb ? m * jt.y : -0.0f);
            tcx = sum4(hasb ? W::ldot(ex, F) : -0.0f);
            tcy = sum4(hasb ? F.y : -0.0f);
            tct = sum4(hasb ? (W::ldot(jt, F) + N.z) : -0.0f);
            V3 jj[K], Iu[K];
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const bool on = hasb && i <= j; // joint i moves this body
                jj[i] = W::lcross(uj[i], W::lsub(X, Aj[i]));
                Iu[i] = W::lmul(Rb, W::lsym(Ib, W::lmulT(Rb, uj[i])));
                Mk.C[0][i] = sum4(on ? m * W::ldot(ex, jj[i]) : -0.0f);
                Mk.C[1][i] = sum4(on ? (m * W::ldot(jt, jj[i]) + Iu[i].z) : -0.0f);
                Mk.C[2][i] = sum4(on ? m * jj[i].y : -0.0f);
#pragma unroll
                for (int k = 0; k < K; ++k)
                    Mk.L[i][k] = (k <= i) ? sum4(on ? (m * W::ldot(jj[i], jj[k]) + W::ldot(uj[i], Iu[k])) : -0.0f) : 0.0f;
                cl[i] = sum4(on ? (W::ldot(jj[i], F) + W::ldot(uj[i], N)) : -0.0f);
                gl[i] = sum4(on ? -((m * W::kGrav) * jj[i].z) : -0.0f);
            }
        }
        float fl[K];
        Lim lim[K];
        int own_any = 0;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            Mk.L[i][i] = Mk.L[i][i] + GX_T2(c_arm, [i]);
            float u = ul[i];
            u = u < -1.0f ? -1.0f : (u > 1.0f ? 1.0f : u);
            fl[i] = ((((-cl[i]) + gl[i]) - GX_T2(c_damp, [i]) * vl[i]) - GX_T2(c_stiff, [i]) * ql[i]) + GX_T2(c_gear, [i]) * u;
            W::limit_row(lim[i], ql[i], vl[i], GX_T2(c_lo, [i]), GX_T2(c_hi, [i]), GX_T2(c_invw, [i]));
            own_any |= lim[i].sg != 0.0f;
        }
        FootR ft;
        {
            constexpr int jf = W::kFlink;
            ft.on = 0; ft.D = 0.0f; ft.aref = 0.0f;
#pragma unroll
            for (int d = 0; d < ND; ++d) ft.J[d] = 0.0f;
            const V3 Xs = W::ladd(Aj[jf], W::lmul(Rj[jf], W::lv(GX_T2(c_fs, [0]), GX_T2(c_fs, [1]), GX_T2(c_fs, [2]))));
            const float dist = (W::c_z0 + Xs.z) - W::c_fr;
            const float pos = dist - W::c_margin;
            if (pos < 0.0f) {
                const V3 Xc = W::lv(Xs.x, Xs.y, Xs.z - (W::c_fr + 0.5f * dist));
                const V3 jt = W::lv(-(Xc.y + y), Xc.x, 0.0f);
                float Jn[ND], T1[ND], T2[ND];
                Jn[0] = 0.0f; Jn[1] = 0.0f; Jn[2] = 0.0f;
                T1[0] = 0.0f; T1[1] = s * jt.x + c * jt.y; T1[2] = c;
                T2[0] = 1.0f; T2[1] = c * jt.x - s * jt.y; T2[2] = -s;
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    const V3 jc = W::lcross(uj[i], W::lsub(Xc, Aj[i]));
                    Jn[3 + i] = jc.z;
                    T1[3 + i] = s * jc.x + c * jc.y;
                    T2[3 + i] = c * jc.x - s * jc.y;
                }
                const float imp = AntRobot::impedance(pos);
                float rr = ((1.0f - imp) * W::c_invw_pyr) / imp;
                if (rr < 1e-15f) rr = 1e-15f;
                ft.on = 1; ft.D = 1.0f / rr;
                // the pyramid row this lane owns (row kq: T1 based for kq < 2, sign by the parity of kq) and its aref
                const float sgn = (kq & 1) ? -W::c_mu : W::c_mu;
#pragma unroll
                for (int d = 0; d < ND; ++d) ft.J[d] = Jn[d] + sgn * ((kq < 2) ? T1[d] : T2[d]);
                float jv = (ft.J[0] * v[0] + ft.J[1] * om) + ft.J[2] * vy;
#pragma unroll
                for (int i = 0; i < K; ++i) jv = jv + ft.J[3 + i] * vl[i];
                ft.aref = -(W::c_kB * jv) - (W::c_kK * imp) * pos;
                own_any = 1;
            }
        }
        const int any_row = quadi<0>(own_any) | quadi<1>(own_any);

        // ---- base block and base smooth force: per-leg terms in leg order
        float B[3][3];
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) B[b][cc] = 0.0f;
        B[0][0] = W::c_mtot; B[2][2] = W::c_mtot; B[2][0] = -(s * W::c_mtot);
        B[1][1] = add_legs(W::c_mB * (y * y) + W::c_IB, tBtt);
        B[1][0] = add_legs(-(W::c_mB * (c * y)), tBxt);
        B[2][1] = add_legs(0.0f, tBty);
        const float cx = add_legs(W::c_mB * (c * Pacc.x - s * Pacc.y), tcx);
        const float cy = add_legs(W::c_mB * Pacc.y, tcy);
        const float ct = add_legs(-(W::c_mB * (y * Pacc.x)), tct);
        float fbase[3];
        fbase[0] = -cx - W::c_dbx * v[0];
        fbase[1] = (-ct - W::c_dbt * om) - W::c_kt * q[1];
        fbase[2] = -cy - W::c_dby * vy;

        float ab[3], alq[K];
        arrow_solve(B, Mk, fbase, fl, kq, ab, alq);
        float fcb[3] = {fbase[0], fbase[1], fbase[2]};
        float fcl[K];
#pragma unroll
        for (int i = 0; i < K; ++i) fcl[i] = fl[i];
        if (any_row) {
            uint32_t act = gather_mask(active_leg(lim, ft, kq, ab, alq));
            for (int it = 0; it < W::kIters; ++it) {
                const uint32_t own = (act >> (16 * L)) & 0xFFFFu;
                LegBlk Hk = Mk;
                float rl[K];
#pragma unroll
                for (int i = 0; i < K; ++i) rl[i] = fl[i];
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    if (!((own >> i) & 1u)) continue;
                    Hk.L[i][i] = Hk.L[i][i] + lim[i].D;
                    rl[i] = rl[i] + (lim[i].D * lim[i].aref) * lim[i].sg;
                }
                // the pyramid row this lane owns: its products (-0 when it is not in the active set), summed over the
                // foot's four rows in the canonical order of four, then added to what they extend
                const bool on = (own >> (8 + kq)) & 1u;
                const float* J = ft.J;
                const float D = ft.D, da = D * ft.aref;
                float PB[3][3], Pr[3];
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    const float dj = D * J[b];
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) PB[b][cc] = (cc <= b) ? sum4(on ? dj * J[cc] : -0.0f) : 0.0f;
#pragma unroll
                    for (int i = 0; i < K; ++i) Hk.C[b][i] = Hk.C[b][i] + sum4(on ? dj * J[3 + i] : -0.0f);
                    Pr[b] = sum4(on ? da * J[b] : -0.0f);
                }
#pragma unroll
                for (int i = 0; i < K; ++i) {
                    const float di = D * J[3 + i];
#pragma unroll
                    for (int cc = 0; cc <= i; ++cc) Hk.L[i][cc] = Hk.L[i][cc] + sum4(on ? di * J[3 + cc] : -0.0f);
                    rl[i] = rl[i] + sum4(on ? da * J[3 + i] : -0.0f);
                }
                float HB[3][3], rb[3];
#pragma unroll
                for (int b = 0; b < 3; ++b) {
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) HB[b][cc] = (cc <= b) ? add_legs(B[b][cc], PB[b][cc]) : 0.0f;
                    rb[b] = add_legs(fbase[b], Pr[b]);
                }
                arrow_solve(HB, Hk, rb, rl, kq, ab, alq);
                const uint32_t nact = gather_mask(active_leg(lim, ft, kq, ab, alq));
                if (nact == act) break;
                act = nact;
            }
#pragma unroll
            for (int i = 0; i < K; ++i) {
                if (lim[i].sg == 0.0f) continue;
                const float res = lim[i].sg * alq[i] - lim[i].aref;
                if (!(res < 0.0f)) continue;
                fcl[i] = fcl[i] + (lim[i].D * (-res)) * lim[i].sg;
            }
            // constraint force of the pyramid rows violated at the solution: own row, canonical order of four
            bool viol = false;
            float frc = 0.0f;
            if (ft.on) {
                const float res = row_dot(ft.J, ab, alq) - ft.aref;
                if (res < 0.0f) { viol = true; frc = ft.D * (-res); }
            }
            float Pf[3];
#pragma unroll
            for (int b = 0; b < 3; ++b) Pf[b] = sum4(viol ? frc * ft.J[b] : -0.0f);
#pragma unroll
            for (int i = 0; i < K; ++i) fcl[i] = fcl[i] + sum4(viol ? frc * ft.J[3 + i] : -0.0f);
#pragma unroll
            for (int b = 0; b < 3; ++b) fcb[b] = add_legs(fcb[b], Pf[b]);
        }
        // ---- Euler with implicit joint damping
        float Bd[3][3];
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) Bd[b][cc] = B[b][cc];
        Bd[0][0] = Bd[0][0] + W::c_h * W::c_dbx;
        Bd[1][1] = Bd[1][1] + W::c_h * W::c_dbt;
        Bd[2][2] = Bd[2][2] + W::c_h * W::c_dby;
        LegBlk Kd = Mk;
#pragma unroll
        for (int i = 0; i < K; ++i) Kd.L[i][i] = Kd.L[i][i] + W::c_h * GX_T2(c_damp, [i]);
        float aib[3], ail[K];
        arrow_solve(Bd, Kd, fcb, fcl, kq, aib, ail);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            qacc[k] = ab[k];
            v[k] = v[k] + W::c_h * aib[k];
            q[k] = q[k] + W::c_h * v[k];
        }
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const float v2 = vl[i] + W::c_h * ail[i];
            const float q2 = ql[i] + W::c_h * v2;
            qacc[3 + i] = quad<0>(alq[i]); qacc[3 + K + i] = quad<1>(alq[i]);
            v[3 + i] = quad<0>(v2); v[3 + K + i] = quad<1>(v2);
            q[3 + i] = quad<0>(q2); q[3 + K + i] = quad<1>(q2);
        }
    }
#undef GX_T2
#undef GX_TB
};

// one mjx.step inside the lane-group kernel: `lane` = lane within the env's 16-lane group
template <class R, bool kQacc>
GX_D void group_substep(float (&q)[R::NQ], float (&v)[R::NV], const float (&ctrl)[R::NU], float (&pose)[4],
                        float (&qacc)[R::NV], int lane)
{
    if constexpr (R::kId == AntRobot::kId) AntGroup::substep_call(q, v, ctrl, pose, qacc, lane & 15);
    else if constexpr (R::kId == WalkerRobot::kId) WalkerGroup::substep_call(q, v, ctrl, pose, qacc, lane & 15);
    else R::template substep<kQacc>(q, v, ctrl, pose, qacc);
}

} // namespace gx
