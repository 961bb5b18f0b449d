// gx_robot.h -- per-robot dynamics (one mjx.step each) and state packing.
//
// A robot is a trait struct: widths (robot.nq/nv/nu, world.py:435-438), the
// float4 SoA packing of its dynamic state, convert_action (engine.py:672-685)
// and `substep` = forward dynamics + semi-implicit Euler [derived: MuJoCo
// computation chapter; constants from tools/model_constants.py].
// fp32, one IEEE operation per operator (see gx_device.h).
#pragma once
#include "gx_device.h"

namespace gx {

// value held by lane K of this lane's quad (DPP quad_perm broadcast: a move, no arithmetic)
template <int K>
GX_D float quad_bc(float x)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), K * 0x55, 0xf, 0xf, true));
}
template <int K>
GX_D bool quad_bcb(bool x) { return __builtin_amdgcn_mov_dpp(x ? 1 : 0, K * 0x55, 0xf, 0xf, true) != 0; }
// a per-lane choice between elements of a small register array.  Each operand goes through an empty asm first: left
// alone, the optimiser turns the select chain into a dynamically indexed load and the array moves to scratch (see
// pick() in gx_robot_kernels.inl)
GX_D float opaque(float x)
{
    asm volatile("" : "+v"(x));
    return x;
}

// ===========================================================================
// Point (xmls/point.xml): slide-x, slide-y, hinge-z; sphere r=.1 + box .05 at
// x=.1, density 1 (:5,19-20); damping .01 .01 .005 (:16-18); h=.02 (:3).
// Actuators: three <general gear=".3"> (:37-39) that set no gain/bias/limit attribute of
// their own and therefore inherit the class's ONE actuator default, which <motor> and then
// <velocity> (:7-8) wrote in document order [derived: MuJoCo XML reference, default/motor ..
// default/velocity "set the attributes of the general element using Actuator shortcuts"]:
// ctrllimited +-1, forcelimited +-.05, gain fixed 1, bias affine (0, 0, -kv), kv = 1.  Force on a
// DOF = gear * clip(clip(ctrl, +-1) - kv * gear * qvel, +-.05).  kBare = the round-1 reading
// (no class defaults: gear * ctrl), kept selectable as robot id 4.
// dyn: (x,y,th,vx) (vy,om,px,py) (pc,ps,done0,steps)
// ===========================================================================
template <bool kBare>
struct PointRobotT {
    static constexpr int kId = kBare ? 4 : 0, NQ = 3, NV = 3, NU = 3, NA = 2, NDYN = 3;
    static constexpr float kH = 0.02f;
    static constexpr float kIo = 2.842182748581224e-05f; // inertia about the hinge axis (enters substep through kInvD3*)
    static constexpr int kDynLanes = 1; // lanes per env in the dynamics pass of the two-kernel rollout
    // The observation pass of a ONE-shard rollout keeps its capped grid even beside the layout sampler: the Point's
    // dynamics chain is short (~0.2 ms in the epoch), the sampler is the epoch's critical chain, and a pass that takes
    // fewer issue slots from it wins (reset_done_heavy, same-box A/B: 746 M uncapped, 752 M capped = round 4's 753 M).
    static constexpr bool kObsCapBesideSampler = true;
    // default Goal_Point_8Hazards observation: ctrl[0:3] compass[3:5] glidar[5:21] hlidar[21:37] qpos[37:40] qvel[40:43]
    static constexpr int kD = 43, kOffCtrl = 0, kOffComp = 3, kOffGl = 5, kOffHl = 21, kOffQpos = 37, kOffQvel = 40;

    GX_D static void load(const float4* __restrict__ dyn, int Npad, int i, float (&q)[NQ], float (&v)[NV],
                          float (&pose0)[4], float& done0, float& steps)
    {
        const float4 d0 = dyn[i], d1 = dyn[Npad + i], d2 = dyn[2 * Npad + i];
        q[0] = d0.x; q[1] = d0.y; q[2] = d0.z; v[0] = d0.w; v[1] = d1.x; v[2] = d1.y;
        pose0[0] = d1.z; pose0[1] = d1.w; pose0[2] = d2.x; pose0[3] = d2.y;
        done0 = d2.z; steps = d2.w;
    }
    GX_D static void store(float4* __restrict__ dyn, int Npad, int i, const float (&q)[NQ], const float (&v)[NV],
                           const float (&pose0)[4], float done0, float steps)
    {
        dyn[i] = make_float4(q[0], q[1], q[2], v[0]);
        dyn[Npad + i] = make_float4(v[1], v[2], pose0[0], pose0[1]);
        dyn[2 * Npad + i] = make_float4(pose0[2], pose0[3], done0, steps);
    }
    // convert_action :672-685: (a0,0,0) rotated by the PRE-step xmat, a1 on the hinge
    GX_D static void convert_action(const float (&pose0)[4], const float (&a)[NA], float (&ctrl)[NU])
    {
        ctrl[0] = pose0[2] * a[0]; ctrl[1] = pose0[3] * a[0]; ctrl[2] = a[1];
    }
    // layout2qpos (:635-638): robot_x / robot_y slide joints; a step from rest with zero ctrl is a fixed point
    static constexpr bool kRestFixed = true;
    GX_D static void place(float (&q)[NQ], float rx, float ry) { q[0] = rx; q[1] = ry; }

    // xpos / xmat of the robot body for qpos `q` (mjx kinematics: hinge quaternion through the half angle): what a step
    // that STARTS from q returns as its (one step stale) pose -- substep computes exactly this
    template <bool kFinite = false> // kFinite: |q[2]| <= 2^25 guaranteed (sincos_f)
    GX_D static void pose_of(const float (&q)[NQ], float (&pose)[4])
    {
        float sh, ch;
        sincos_f<kFinite>(0.5f * q[2], sh, ch);
        pose[0] = q[0]; pose[1] = q[1]; pose[2] = ch * ch - sh * sh; pose[3] = 2.0f * (ch * sh);
    }

    // jp.clip: NaN stays.  v_med3_f32 is the exact median for ordered operands (and keeps -0); one compare puts the
    // NaN back -- 3 instructions instead of two compare / select pairs on the serial chain of the dynamics pass
    GX_D static float clip(float x, float lim)
    {
        const float m = __builtin_amdgcn_fmed3f(x, -lim, lim);
        return x == x ? m : x;
    }
    // qfrc_actuator of one DOF [derived: mjx fwd_actuation].  kNoNaN: the caller guarantees ctrl and vel are not NaN
    // (the dynamics pass checks that after the fact and redoes the step otherwise), so the median alone is the clip
    template <bool kNoNaN = false>
    GX_D static float actuate(float ctrl, float vel)
    {
        constexpr float kGear = 0.3f, kCtrlLim = 1.0f, kForceLim = 0.05f, kKv = 1.0f;
        if (kBare) return kGear * ctrl;
        if (kNoNaN) {
            const float u = __builtin_amdgcn_fmed3f(ctrl, -kCtrlLim, kCtrlLim);
            const float force = __builtin_amdgcn_fmed3f(u - kKv * (kGear * vel), -kForceLim, kForceLim);
            return kGear * force;
        }
        const float u = clip(ctrl, kCtrlLim);
        const float force = clip(u - kKv * (kGear * vel), kForceLim);
        return kGear * force;
    }

    template <bool kQacc, bool kNoNaN = false>
    GX_D static void substep(float (&q)[NQ], float (&v)[NV], const float (&ctrl)[NU], float (&pose)[4],
                             float (&qacc)[NV])
    {
        constexpr float kMxc = 0.0001f, kDxy = 0.01f, kDt = 0.005f;
        constexpr float kInvM = (float)(1.0 / 0.005188790204786391);
        constexpr float kInvA = (float)(1.0 / (0.005188790204786391 + 0.02 * 0.01));
        pose_of<kNoNaN>(q, pose); // (kNoNaN: the dynamics pass, whose state sums are below kStateLimit = 2^24)
        const float c = pose[2], sn = pose[3];
        const float b = -(kMxc * sn), d = kMxc * c;
        const float w2 = v[2] * v[2];
        const float fx = (-(kDxy * v[0]) - (-(d * w2))) + actuate<kNoNaN>(ctrl[0], v[0]);
        const float fy = (-(kDxy * v[1]) - (b * w2)) + actuate<kNoNaN>(ctrl[1], v[1]);
        const float ft = (-(kDt * v[2]) - 0.0f) + actuate<kNoNaN>(ctrl[2], v[2]);
        const float t = b * fx + d * fy;
        // Schur complement of the hinge row after eliminating the two slides: Io - (b^2 + d^2) / m with
        // b^2 + d^2 = (m xc)^2 (sin^2 + cos^2) = (m xc)^2 -- a constant of the model, so the "division" of the 3x3
        // solve is a multiplication by its reciprocal (round 3; rounds 1-2 divided by the fp32 value of
        // kIo - (b*b + d*d) * kInvM, equal to 1e-9 relative)
        constexpr float kInvD3M = (float)(1.0 / (2.842182748581224e-05 - (1.0e-4 * 1.0e-4) / 0.005188790204786391));
        constexpr float kInvD3A = (float)(1.0 / ((2.842182748581224e-05 + 0.02 * 0.005) -
                                                 (1.0e-4 * 1.0e-4) / (0.005188790204786391 + 0.02 * 0.01)));
        if (kQacc) {
            const float y3 = ft - t * kInvM;
            const float q3 = y3 * kInvD3M;
            qacc[0] = (fx - b * q3) * kInvM;
            qacc[1] = (fy - d * q3) * kInvM;
            qacc[2] = q3;
        }
        const float y3 = ft - t * kInvA;
        const float q3 = y3 * kInvD3A;
        const float q1 = (fx - b * q3) * kInvA;
        const float q2 = (fy - d * q3) * kInvA;
        v[0] = v[0] + kH * q1;
        v[1] = v[1] + kH * q2;
        v[2] = v[2] + kH * q3;
        q[0] = q[0] + kH * v[0];
        q[1] = q[1] + kH * v[1];
        q[2] = q[2] + kH * v[2];
    }
};

using PointRobot = PointRobotT<false>;     // xmls/point.xml as MuJoCo compiles it (class defaults inherited)
using PointBareRobot = PointRobotT<true>;  // round-1 reading, robot id 4

// ===========================================================================
// Swimmer (xmls/swimmer.xml): slide-x, slide-y, hinge-z at the head link, two
// limited hinges (+-100 deg, :24,28) down a 3-capsule chain (r=.02, l=.15,
// density 1000, :18,23,27); armature .1 on every DOF (:6); motors gear 20 with
// ctrlrange +-1 (:58-59); h=.03 (:3); no damping, no contacts (capsules rest at
// dist == margin).  Joint-limit rows follow MJX (_instantiate_limit_slide_hinge,
// _kbi: solref (.02,1) with refsafe -> timeconst .06, solimp (.9,.95,.001,.5,2));
// with at most two scalar rows the constraint QP is solved exactly by enumeration.
// dyn: (x,y,t1,p2) (p3,vx,vy,w1) (w2,w3,px,py) (pc,ps,done0,steps)
// ===========================================================================
struct SwimmerRobot {
    static constexpr int kId = 1, NQ = 5, NV = 5, NU = 2, NA = 2, NDYN = 4;
    static constexpr float kH = 0.03f;
    // default Goal_Swimmer_8Hazards observation: ctrl[0:2] compass[2:4] glidar[4:20] hlidar[20:36] qpos[36:41] qvel[41:46]
    static constexpr int kD = 46, kOffCtrl = 0, kOffComp = 2, kOffGl = 4, kOffHl = 20, kOffQpos = 36, kOffQvel = 41;

    GX_D static void load(const float4* __restrict__ dyn, int Npad, int i, float (&q)[NQ], float (&v)[NV],
                          float (&pose0)[4], float& done0, float& steps)
    {
        const float4 d0 = dyn[i], d1 = dyn[Npad + i], d2 = dyn[2 * Npad + i], d3 = dyn[3 * Npad + i];
        q[0] = d0.x; q[1] = d0.y; q[2] = d0.z; q[3] = d0.w; q[4] = d1.x;
        v[0] = d1.y; v[1] = d1.z; v[2] = d1.w; v[3] = d2.x; v[4] = d2.y;
        pose0[0] = d2.z; pose0[1] = d2.w; pose0[2] = d3.x; pose0[3] = d3.y;
        done0 = d3.z; steps = d3.w;
    }
    GX_D static void store(float4* __restrict__ dyn, int Npad, int i, const float (&q)[NQ], const float (&v)[NV],
                           const float (&pose0)[4], float done0, float steps)
    {
        dyn[i] = make_float4(q[0], q[1], q[2], q[3]);
        dyn[Npad + i] = make_float4(q[4], v[0], v[1], v[2]);
        dyn[2 * Npad + i] = make_float4(v[3], v[4], pose0[0], pose0[1]);
        dyn[3 * Npad + i] = make_float4(pose0[2], pose0[3], done0, steps);
    }
    GX_D static void convert_action(const float (&)[4], const float (&a)[NA], float (&ctrl)[NU])
    {
        ctrl[0] = a[0]; ctrl[1] = a[1]; // non-point robots: the action is the ctrl (:673)
    }
    static constexpr bool kRestFixed = true;
    GX_D static void place(float (&q)[NQ], float rx, float ry) { q[0] = rx; q[1] = ry; }

    // pose of the head link for qpos `q` (what substep computes for it, see PointRobotT::pose_of)
    GX_D static void pose_of(const float (&q)[NQ], float (&pose)[4])
    {
        float sh1, ch1;
        sincos_f(0.5f * q[2], sh1, ch1);
        pose[0] = q[0]; pose[1] = q[1]; pose[2] = ch1 * ch1 - sh1 * sh1; pose[3] = 2.0f * (ch1 * sh1);
    }

    struct Ldl3 { float rd0, rd1, rd2, l10, l20, l21; };
    GX_D static void ldl_solve(const Ldl3& f, float b0, float b1, float b2, float (&x)[3])
    {
        const float y0 = b0;
        const float y1 = b1 - f.l10 * y0;
        const float y2 = (b2 - f.l20 * y0) - f.l21 * y1;
        const float z2 = y2 * f.rd2;
        const float z1 = y1 * f.rd1 - f.l21 * z2;
        const float z0 = (y0 * f.rd0 - f.l10 * z1) - f.l20 * z2;
        x[0] = z0; x[1] = z1; x[2] = z2;
    }
    // joint-limit row: present when violated; sign, aref and R = 1/D
    GX_D static bool limit_row(float qj, float vel, float invw, float& sign, float& aref, float& R)
    {
        constexpr float kLim = 1.7453292519943295f, kK = 307.78701138811942f, kB = 35.087719298245617f;
        const float dlo = qj - (-kLim), dhi = kLim - qj;
        const float pos = dlo < dhi ? dlo : dhi;
        const float sg = dlo < dhi ? 1.0f : -1.0f;
        sign = sg; aref = 0.0f; R = 1.0f;
        if (!(pos < 0.0f)) return false;
        const float ix = fabsf(pos) / 0.001f;
        float iy;
        if (ix < 0.5f) iy = 2.0f * (ix * ix);
        else iy = 1.0f - 2.0f * ((1.0f - ix) * (1.0f - ix));
        float imp = 0.9f + iy * (0.95f - 0.9f);
        if (imp < 0.9f) imp = 0.9f;
        if (imp > 0.95f) imp = 0.95f;
        if (ix > 1.0f) imp = 0.95f;
        aref = -(kB * (sg * vel)) - (kK * imp) * pos;
        float r = ((1.0f - imp) * invw) / imp;
        if (r < 1e-15f) r = 1e-15f;
        R = r;
        return true;
    }

    // lanes per env in the dynamics pass of the two-kernel rollout: the four lanes of a quad share one env and split
    // what is independent in the step -- the three half-angle sincos, the two joint-limit rows, the two unit solves and
    // the three candidate active sets of the limit QP; everything else they evaluate redundantly (same operations,
    // same bits).  Every value is produced by the same expression as in the one-lane form and moved, never
    // re-associated, so the forms agree bit for bit (770 -> 600 instructions on the serial chain of a step: a 200-step
    // rollout alone 395 -> 322 us).  Used when no layout sampler runs beside the rollout: 125 waves instead of 32 each
    // take most of a SIMD's issue slots, and with the sampler saturating the vector ALUs the epoch is faster with
    // the one-lane form (0.581 against 0.595 ms), so gx_rollout picks per launch (SplitArgs::lanes).
    static constexpr int kDynLanes = 4;

    template <bool kQacc, bool kNoNaN = false> // kNoNaN: nothing to gain here (the clamp below passes NaN through by itself)
    GX_D static void substep(float (&q)[NQ], float (&v)[NV], const float (&ctrl)[NU], float (&pose)[4],
                             float (&qacc)[NV])
    {
        substep_q<false>(q, v, ctrl, pose, qacc, 0);
    }

    // kQuad: called by all four lanes of a quad that hold the same env (identical q, v, ctrl); j = lane & 3
    template <bool kQuad>
    GX_D static void substep_q(float (&q)[NQ], float (&v)[NV], const float (&ctrl)[NU], float (&pose)[4],
                               float (&qacc)[NV], int j)
    {
        constexpr float kM = 0.22200588085367876f, kIc = 0.00060383505197098225f, kArm = 0.1f, kGear = 20.0f;
        constexpr float kInvW2 = 9.3234461878793518f, kInvW3 = 9.8671592198756102f;
        constexpr float A11 = 0.225f, A21 = 0.15f, A22 = -0.075f, A31 = 0.15f, A32 = -0.15f, A33 = -0.075f;
        constexpr float kImu = (float)(1.0 / (3.0 * 0.22200588085367876 + 0.1));
        // kinematics: hinge quaternions (half angles) composed down the chain
        float sh1, ch1, sh2, ch2, sh3, ch3;
        if (kQuad) { // lane j takes joint min(j, 2)
            const float a0 = opaque(q[2]), a1 = opaque(q[3]), a2 = opaque(q[4]);
            const float ang = j == 0 ? a0 : (j == 1 ? a1 : a2);
            float sh, ch;
            sincos_f(0.5f * ang, sh, ch);
            sh1 = quad_bc<0>(sh); ch1 = quad_bc<0>(ch);
            sh2 = quad_bc<1>(sh); ch2 = quad_bc<1>(ch);
            sh3 = quad_bc<2>(sh); ch3 = quad_bc<2>(ch);
        } else {
            sincos_f(0.5f * q[2], sh1, ch1);
            sincos_f(0.5f * q[3], sh2, ch2);
            sincos_f(0.5f * q[4], sh3, ch3);
        }
        const float w1 = ch1, z1 = sh1;
        const float w2 = w1 * ch2 - z1 * sh2, z2 = w1 * sh2 + z1 * ch2;
        const float w3 = w2 * ch3 - z2 * sh3, z3 = w2 * sh3 + z2 * ch3;
        const float c1 = w1 * w1 - z1 * z1, s1 = 2.0f * (w1 * z1);
        const float c2 = w2 * w2 - z2 * z2, s2 = 2.0f * (w2 * z2);
        const float c3 = w3 * w3 - z3 * z3, s3 = 2.0f * (w3 * z3);
        pose[0] = q[0]; pose[1] = q[1]; pose[2] = c1; pose[3] = s1;
        const float W1 = v[2], W2 = W1 + v[3], W3 = W2 + v[4];
        // COM Jacobian columns g_ij = sum_{k>=j} a_ik n_k, n_k = (-s_k, c_k)
        const float g11x = A11 * -s1, g11y = A11 * c1;
        const float g22x = A22 * -s2, g22y = A22 * c2;
        const float g21x = A21 * -s1 + g22x, g21y = A21 * c1 + g22y;
        const float g33x = A33 * -s3, g33y = A33 * c3;
        const float g32x = A32 * -s2 + g33x, g32y = A32 * c2 + g33y;
        const float g31x = A31 * -s1 + g32x, g31y = A31 * c1 + g32y;
        // velocity-product acceleration of the COMs
        const float e1 = W1 * W1, e2 = W2 * W2, e3 = W3 * W3;
        const float q1x = -((A11 * e1) * c1), q1y = -((A11 * e1) * s1);
        const float q2x = -((A21 * e1) * c1 + (A22 * e2) * c2), q2y = -((A21 * e1) * s1 + (A22 * e2) * s2);
        const float q3x = -(((A31 * e1) * c1 + (A32 * e2) * c2) + (A33 * e3) * c3);
        const float q3y = -(((A31 * e1) * s1 + (A32 * e2) * s2) + (A33 * e3) * s3);
        // mass matrix blocks
        const float Mx0 = kM * ((g11x + g21x) + g31x), Mx1 = kM * (g22x + g32x), Mx2 = kM * g33x;
        const float My0 = kM * ((g11y + g21y) + g31y), My1 = kM * (g22y + g32y), My2 = kM * g33y;
        const float T00 = (kM * (((g11x * g11x + g11y * g11y) + (g21x * g21x + g21y * g21y)) + (g31x * g31x + g31y * g31y)) + 3.0f * kIc) + kArm;
        const float T10 = kM * ((g21x * g22x + g21y * g22y) + (g31x * g32x + g31y * g32y)) + 2.0f * kIc;
        const float T20 = kM * (g31x * g33x + g31y * g33y) + kIc;
        const float T11 = (kM * ((g22x * g22x + g22y * g22y) + (g32x * g32x + g32y * g32y)) + 2.0f * kIc) + kArm;
        const float T21 = kM * (g32x * g33x + g32y * g33y) + kIc;
        const float T22 = (kM * (g33x * g33x + g33y * g33y) + kIc) + kArm;
        // bias, smooth force = (passive - bias) + actuator (ctrl clamped for the force only)
        const float bx = kM * ((q1x + q2x) + q3x), by = kM * ((q1y + q2y) + q3y);
        const float b1 = kM * (((g11x * q1x + g11y * q1y) + (g21x * q2x + g21y * q2y)) + (g31x * q3x + g31y * q3y));
        const float b2 = kM * ((g22x * q2x + g22y * q2y) + (g32x * q3x + g32y * q3y));
        const float b3 = kM * (g33x * q3x + g33y * q3y);
        float u0 = ctrl[0], u1 = ctrl[1];
        u0 = u0 < -1.0f ? -1.0f : (u0 > 1.0f ? 1.0f : u0);
        u1 = u1 < -1.0f ? -1.0f : (u1 > 1.0f ? 1.0f : u1);
        const float fx = 0.0f - bx, fy = 0.0f - by;
        const float ft0 = 0.0f - b1, ft1 = (0.0f - b2) + kGear * u0, ft2 = (0.0f - b3) + kGear * u1;
        // Schur complement of the diagonal translation block, LDL^T
        const float S00 = T00 - (Mx0 * Mx0 + My0 * My0) * kImu;
        const float S10 = T10 - (Mx1 * Mx0 + My1 * My0) * kImu;
        const float S11 = T11 - (Mx1 * Mx1 + My1 * My1) * kImu;
        const float S20 = T20 - (Mx2 * Mx0 + My2 * My0) * kImu;
        const float S21 = T21 - (Mx2 * Mx1 + My2 * My1) * kImu;
        const float S22 = T22 - (Mx2 * Mx2 + My2 * My2) * kImu;
        const float r0 = ft0 - (Mx0 * fx + My0 * fy) * kImu;
        const float r1 = ft1 - (Mx1 * fx + My1 * fy) * kImu;
        const float r2 = ft2 - (Mx2 * fx + My2 * fy) * kImu;
        Ldl3 F;
        F.rd0 = 1.0f / S00;
        F.l10 = S10 * F.rd0;
        F.l20 = S20 * F.rd0;
        const float d1 = S11 - F.l10 * S10;
        F.rd1 = 1.0f / d1;
        const float t21 = S21 - F.l20 * S10;
        F.l21 = t21 * F.rd1;
        const float d2 = (S22 - F.l20 * S20) - F.l21 * t21;
        F.rd2 = 1.0f / d2;
        float a[3];
        ldl_solve(F, r0, r1, r2, a);
        // joint limits on phi2, phi3
        float sg2, ar2, R2, sg3, ar3, R3;
        bool p2, p3;
        if (kQuad) { // lanes 0, 1: phi2's row; lanes 2, 3: phi3's
            const bool two = j >= 2;
            float sg, ar, Rr;
            const float q3_ = opaque(q[3]), q4_ = opaque(q[4]), v3_ = opaque(v[3]), v4_ = opaque(v[4]);
            const bool pr = limit_row(two ? q4_ : q3_, two ? v4_ : v3_, two ? kInvW3 : kInvW2, sg, ar, Rr);
            p2 = quad_bcb<0>(pr); sg2 = quad_bc<0>(sg); ar2 = quad_bc<0>(ar); R2 = quad_bc<0>(Rr);
            p3 = quad_bcb<2>(pr); sg3 = quad_bc<2>(sg); ar3 = quad_bc<2>(ar); R3 = quad_bc<2>(Rr);
        } else {
            p2 = limit_row(q[3], v[3], kInvW2, sg2, ar2, R2);
            p3 = limit_row(q[4], v[4], kInvW3, sg3, ar3, R3);
        }
        if (p2 || p3) {
            if (!p2) sg2 = 0.0f;
            if (!p3) sg3 = 0.0f;
            float A22i, A33i, z31;
            if (kQuad) { // one unit solve per half of the quad
                const bool two = j >= 2;
                float zc[3];
                ldl_solve(F, 0.0f, two ? 0.0f : 1.0f, two ? 1.0f : 0.0f, zc);
                A22i = quad_bc<0>(zc[1]); A33i = quad_bc<2>(zc[2]); z31 = quad_bc<2>(zc[1]);
            } else {
                float zc2[3], zc3[3];
                ldl_solve(F, 0.0f, 1.0f, 0.0f, zc2);
                ldl_solve(F, 0.0f, 0.0f, 1.0f, zc3);
                A22i = zc2[1]; A33i = zc3[2]; z31 = zc3[1];
            }
            const float A23i = z31 * (sg2 * sg3);
            const float E2 = sg2 * a[1] - ar2, E3 = sg3 * a[2] - ar3;
            float f2 = 0.0f, f3 = 0.0f;
            if (kQuad) {
                // the three candidate active sets side by side: lane 0 both rows, lane 1 row 2 alone, lanes 2, 3 row 3
                // alone; then the one-lane form's order of preference (both, row 2, row 3)
                const float m22 = R2 + A22i, m33 = R3 + A33i;
                const float det = m22 * m33 - A23i * A23i;
                const float nA2 = (-E2) * m33 - A23i * (-E3), nA3 = m22 * (-E3) - A23i * (-E2);
                const float num1 = j == 0 ? nA2 : (j == 1 ? -E2 : -E3);
                const float den = j == 0 ? det : (j == 1 ? m22 : m33);
                const float g1 = num1 / den;     // lane 0: g2 of "both"; lane 1: g2 alone; lanes 2, 3: g3 alone
                const float gq = nA3 / den;      // lane 0: g3 of "both" (unused elsewhere)
                const float gA2 = quad_bc<0>(g1), gA3 = quad_bc<0>(gq), gB = quad_bc<1>(g1), gC = quad_bc<2>(g1);
                bool done = false;
                if (p2 && p3) {
                    if (gA2 > 0.0f && gA3 > 0.0f) { f2 = gA2; f3 = gA3; done = true; }
                }
                if (!done && p2) {
                    if (gB > 0.0f && (!p3 || !(E3 + A23i * gB < 0.0f))) { f2 = gB; f3 = 0.0f; done = true; }
                }
                if (!done && p3) {
                    if (gC > 0.0f && (!p2 || !(E2 + A23i * gC < 0.0f))) { f3 = gC; f2 = 0.0f; done = true; }
                }
            } else {
                bool done = false;
                if (p2 && p3) {
                    const float m22 = R2 + A22i, m33 = R3 + A33i;
                    const float det = m22 * m33 - A23i * A23i;
                    const float g2 = ((-E2) * m33 - A23i * (-E3)) / det;
                    const float g3 = (m22 * (-E3) - A23i * (-E2)) / det;
                    if (g2 > 0.0f && g3 > 0.0f) { f2 = g2; f3 = g3; done = true; }
                }
                if (!done && p2) {
                    const float g2 = (-E2) / (R2 + A22i);
                    if (g2 > 0.0f && (!p3 || !(E3 + A23i * g2 < 0.0f))) { f2 = g2; f3 = 0.0f; done = true; }
                }
                if (!done && p3) {
                    const float g3 = (-E3) / (R3 + A33i);
                    if (g3 > 0.0f && (!p2 || !(E2 + A23i * g3 < 0.0f))) { f3 = g3; f2 = 0.0f; done = true; }
                }
            }
            ldl_solve(F, r0, r1 + sg2 * f2, r2 + sg3 * f3, a);
        }
        const float ax = (fx - ((Mx0 * a[0] + Mx1 * a[1]) + Mx2 * a[2])) * kImu;
        const float ay = (fy - ((My0 * a[0] + My1 * a[1]) + My2 * a[2])) * kImu;
        qacc[0] = ax; qacc[1] = ay; qacc[2] = a[0]; qacc[3] = a[1]; qacc[4] = a[2];
#pragma unroll
        for (int k = 0; k < 5; ++k) v[k] = v[k] + kH * qacc[k];
#pragma unroll
        for (int k = 0; k < 5; ++k) q[k] = q[k] + kH * v[k];
    }
};

} // namespace gx

#include "gx_robot_ant.h"
#include "gx_robot_ant_group.h"
#include "gx_robot_legs.h"
#include "gx_robot_legs_group.h"
