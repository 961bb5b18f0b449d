"""Second, independent restatement of the same reference path in numpy.

TEST INFRASTRUCTURE ONLY.  Where gx_oracle.c is a per-env scalar loop with its
own deterministic sin/cos/atan2/exp, this file is a batched numpy float32
transcription that uses numpy's libm-backed functions and a vectorised
threefry.  The two were written separately from the reference source
(engine.py:426-505, 546-639, 644-731, 738-929); tests/test_oracle_crosscheck.py
requires them to agree (PRNG: exactly; floating point: to 1e-5, lidar compared
edge-aware), which catches transcription slips in either one.  It does not pin
parity with the reference (still PARITY UNPINNED, see gx_oracle.h).
"""
import numpy as np

f32 = np.float32
U32 = np.uint32


# --------------------------------------------------------------------------
# jax.random on threefry2x32 (vectorised)
# --------------------------------------------------------------------------
def _rotl(x, r):
    return (x << U32(r)) | (x >> U32(32 - r))


def threefry2x32(k0, k1, x0, x1):
    with np.errstate(over='ignore'):
        k0, k1 = U32(k0), U32(k1)
        x0 = np.asarray(x0, U32).copy()
        x1 = np.asarray(x1, U32).copy()
        ks = [k0, k1, U32(k0 ^ k1 ^ U32(0x1BD11BDA))]
        rot = [[13, 15, 26, 6], [17, 29, 16, 24]]
        x0 += ks[0]
        x1 += ks[1]
        for i in range(5):
            for r in rot[i % 2]:
                x0 += x1
                x1 = _rotl(x1, r)
                x1 ^= x0
            x0 += ks[(i + 1) % 3]
            x1 += ks[(i + 2) % 3] + U32(i + 1)
        return x0, x1


def random_bits(key, n):
    """threefry_2x32(key, iota(n)) as jax lays it out (prng.py)."""
    cnt = np.arange(n, dtype=U32)
    if n % 2:
        cnt = np.concatenate([cnt, np.zeros(1, U32)])
    half = cnt.size // 2
    y0, y1 = threefry2x32(key[0], key[1], cnt[:half], cnt[half:])
    return np.concatenate([y0, y1])[:n]


def split(key, n=2):
    return random_bits(key, 2 * n).reshape(n, 2)


def uniform(key, lo, hi):
    bits = random_bits(key, 1)[0]
    fl = ((bits >> U32(9)) | U32(0x3F800000)).view(f32) - f32(1.0)
    lo, hi = f32(lo), f32(hi)
    return np.maximum(lo, f32(fl * f32(hi - lo)) + lo)


def randint(key, n, span):
    k1, k2 = split(key, 2)
    hi, lo = random_bits(k1, n), random_bits(k2, n)
    span = U32(span)
    with np.errstate(over='ignore'):
        mult = U32(65536) % span
        mult = U32(mult * mult) % span
        off = (hi % span) * mult + (lo % span)
    return (off % span).astype(np.int64)


# --------------------------------------------------------------------------
# layout sampling  engine.py:546-621 (one candidate, scalar python: small cases only)
# --------------------------------------------------------------------------
def sample_layout(key, hazards_num=8, keepouts=(0.5, 0.4, 0.4), extents=(-2, -2, 2, 2), margin=0.0,
                  pillars_num=0, pillars_keepout=0.3):
    """pillars (synthetic extension, no reference counterpart) are placed after the hazards"""
    names = ['goal'] + [f'hazard{i}' for i in range(hazards_num)] + [f'pillar{i}' for i in range(pillars_num)] + ['robot']
    ko = {n: (keepouts[0] if n == 'goal' else keepouts[2] if n == 'robot' else
              pillars_keepout if n.startswith('pillar') else keepouts[1]) for n in names}
    rng = np.asarray(key, U32)
    layout, success = {}, True
    for name in names:
        k = ko[name]
        xmin, ymin, xmax, ymax = extents[0] + k, extents[1] + k, extents[2] - k, extents[3] - k
        conflicted, xy = True, np.array([-np.inf, -np.inf], f32)
        for _ in range(10):
            rng, rng1 = split(rng, 2)
            r1, r2 = split(rng1, 2)
            cur = np.array([uniform(r1, xmin, xmax), uniform(r2, ymin, ymax)], f32)
            ok = True
            for other, oxy in layout.items():
                d = np.sqrt(np.sum(np.square(cur - oxy), dtype=f32), dtype=f32)
                if d < f32(ko[other] + margin + k):
                    ok = False
            if ok:
                xy, conflicted = cur, False
        layout[name] = xy
        success = success and not conflicted
    d = np.sqrt(np.sum(np.square(layout['robot'] - layout['goal']), dtype=f32), dtype=f32)
    if d < f32(3.0):
        success = False
    return np.stack([layout[n] for n in names]), success


# --------------------------------------------------------------------------
# batched Point step + observation  engine.py:659-700, 738-929
# --------------------------------------------------------------------------
M, MXC, IO = 0.005188790204786391, 1.0e-4, 2.842182748581224e-05
H_, DXY, DTH, GEAR = f32(0.02), f32(0.01), f32(0.005), f32(0.3)


CTRL_LIM, FORCE_LIM, KV = 1.0, 0.05, 1.0      # point.xml:7-8 defaults inherited by the <general> actuators


def point_actuation(ctrl, v, bare=False):
    """qfrc_actuator [derived: one actuator default per class, <motor> then <velocity> write it;
    mjx fwd_actuation]: gear * clip(clip(ctrl, +-1) - kv * gear * qvel, +-0.05).  bare=True is the
    round-1 reading (general actuators without the class defaults): gear * ctrl."""
    g = float(GEAR)
    if bare:
        return g * ctrl
    u = np.clip(ctrl.astype(np.float64), -CTRL_LIM, CTRL_LIM)
    return g * np.clip(u - KV * (g * v.astype(np.float64)), -FORCE_LIM, FORCE_LIM)


def point_substep(q, v, ctrl, bare=False):
    """q,v (N,3) f32; returns pose (N,4), qacc (N,3), q', v'.  [derived] MuJoCo Euler step."""
    th = q[:, 2]
    c, s = np.cos(th, dtype=f32), np.sin(th, dtype=f32)
    pose = np.stack([q[:, 0], q[:, 1], c, s], 1).astype(f32)
    N = q.shape[0]
    Mm = np.zeros((N, 3, 3), np.float64)
    Mm[:, 0, 0] = Mm[:, 1, 1] = M
    Mm[:, 2, 2] = IO
    Mm[:, 0, 2] = Mm[:, 2, 0] = -MXC * s
    Mm[:, 1, 2] = Mm[:, 2, 1] = MXC * c
    w2 = v[:, 2].astype(np.float64) ** 2
    bias = np.stack([-MXC * c * w2, -MXC * s * w2, np.zeros(N)], 1)
    damp = np.array([DXY, DXY, DTH], np.float64)
    f = -damp * v - bias + point_actuation(ctrl, v, bare)
    qacc = np.linalg.solve(Mm, f[..., None])[..., 0]
    Mi = Mm + np.eye(3) * (float(H_) * damp)
    qa = np.linalg.solve(Mi, f[..., None])[..., 0]
    v2 = (v + float(H_) * qa).astype(f32)
    q2 = (q + float(H_) * v2).astype(f32)
    return pose, qacc.astype(f32), q2, v2


def ego(pose, pts):
    """pts (N,K,2) world -> robot frame (N,K,2)  engine.py:817-826"""
    d = pts - pose[:, None, :2]
    c, s = pose[:, None, 2], pose[:, None, 3]
    return np.stack([d[..., 0] * c + d[..., 1] * s, -d[..., 0] * s + d[..., 1] * c], -1).astype(f32)


def lidar(pose, pts, bins=16, gain=1.0, alias=True, max_dist=None):
    """engine.py:846-900; returns (N,bins) and the per-object (angle/bin_size) for edge checks"""
    z = ego(pose, pts)
    N, K, _ = z.shape
    dist = np.sqrt(z[..., 0] ** 2 + z[..., 1] ** 2).astype(f32)
    ang = np.arctan2(z[..., 1], z[..., 0]).astype(f32) % f32(2 * np.pi)
    bs = f32(2 * np.pi / bins)
    pos = (ang / bs).astype(f32)
    b = pos.astype(np.int64)
    sensor = np.exp(-f32(gain) * dist).astype(f32) if max_dist is None else \
        (np.maximum(0, f32(max_dist) - dist) / f32(max_dist)).astype(f32)
    al = ((ang - bs * b.astype(f32)) / bs).astype(f32)
    out = np.zeros((N, bins), f32)
    rows = np.arange(N)
    for k in range(K):
        bk = b[:, k]
        inb = bk < bins
        out[rows[inb], bk[inb]] = np.maximum(out[rows[inb], bk[inb]], sensor[inb, k])
        if alias:
            bp, bm = (bk + 1) % bins, (bk - 1) % bins
            out[rows, bp] = np.maximum(out[rows, bp], al[:, k] * sensor[:, k])
            out[rows, bm] = np.maximum(out[rows, bm], (1 - al[:, k]) * sensor[:, k])
    return out, pos


def step(state, action, cfg):
    """state: dict like OracleEngine.get_state(); returns (obs, reward, done, cost, new_state, edge_pos)."""
    q, v, p0 = state['qpos'].copy(), state['qvel'].copy(), state['pose0'].copy()
    objs, last_done, steps = state['objs'], state['done0'], state['steps']
    a = action.astype(f32)
    ctrl = np.stack([p0[:, 2] * a[:, 0], p0[:, 3] * a[:, 0], a[:, 1]], 1).astype(f32)
    for _ in range(cfg.get('physics_steps_per_control_step', 1)):
        pose, qacc, q, v = point_substep(q, v, ctrl, cfg.get('_point_bare', False))
    bins = cfg.get('lidar_num_bins', 16)
    gl, pos_g = lidar(pose, objs[:, :1], bins)
    hl, pos_h = lidar(pose, objs[:, 1:], bins)
    comp = ego(pose, objs[:, :1])[:, 0]
    obs = np.concatenate([ctrl, comp, gl, hl, q, v], 1).astype(f32)   # sorted keys, default flags
    dg = np.linalg.norm(objs[:, 0] - pose[:, :2], axis=1).astype(f32)
    dl = np.linalg.norm(objs[:, 0] - p0[:, :2], axis=1).astype(f32)
    have_last = state['hist'] >= 1
    last = np.where(last_done > 0, dg, dl) if have_last else dg
    dd = (last - dg).astype(f32)
    rew = (dd * f32(cfg.get('reward_distance', 1.0))).astype(f32)
    done = (dg < f32(cfg.get('goal_size', 0.5))).astype(f32)
    bad_sim = np.abs(dd) > 1
    done[bad_sim] = 1
    rew[bad_sim] = 0
    hs = f32(cfg.get('hazards_size', 0.3))
    dh = np.linalg.norm(objs[:, 1:] - pose[:, None, :2], axis=2).astype(f32)
    cost = np.sum(hs - np.minimum(dh, hs), axis=1, dtype=f32)
    bad = ~np.isfinite(obs).all(1)
    rew[bad] = 0
    done[bad] = 1
    done = np.where(steps > cfg.get('num_steps', 1000), f32(1), done).astype(f32)
    nsteps = np.where(done > 0, 0, steps + 1).astype(f32)
    new = dict(state, qpos=q, qvel=v, pose0=pose, done0=done, done1=last_done.copy(), steps=nsteps,
               hist=min(2, state['hist'] + 1))
    return obs, rew, done, cost, new, (pos_g, pos_h), qacc


# --------------------------------------------------------------------------
# Swimmer (xmls/swimmer.xml): independent float64 restatement of one mjx.step
# [derived].  Generic formulation on purpose (complex-number kinematics, Jacobians
# of COM positions, velocity-product term by differentiating J numerically, limit
# rows solved by an active-set loop) -- nothing shared with gx_oracle.c's closed form.
# --------------------------------------------------------------------------
SW = dict(h=0.03, r=0.02, length=0.15, rho=1000.0, arm=0.1, gear=20.0, lim=100 * np.pi / 180,
          solref=(0.02, 1.0), solimp=(0.9, 0.95, 0.001, 0.5, 2.0))


def _sw_capsule():
    r, L, rho = SW['r'], SW['length'], SW['rho']
    vc, vs = np.pi * r * r * L, 4 / 3 * np.pi * r ** 3
    mc, ms = rho * vc, rho * vs
    I = mc * (3 * r * r + L * L) / 12 + 2 * ms * r * r / 5 + ms * L * (3 * r + 2 * L) / 8
    return mc + ms, I


def _sw_coms(q):
    """COM of the three links as complex numbers; bodies per swimmer.xml:14-31"""
    p = q[0] + 1j * q[1]
    a1 = q[2]; a2 = a1 + q[3]; a3 = a2 + q[4]
    u1, u2, u3 = np.exp(1j * a1), np.exp(1j * a2), np.exp(1j * a3)
    b2 = p + 0.15 * u1            # body "mid" origin
    b3 = b2 - 0.15 * u2           # body "back" origin
    return np.array([p + 0.225 * u1, b2 - 0.075 * u2, b3 - 0.075 * u3])


def _sw_jac(q):
    """J[i] (2x5) of COM_i, analytic via the complex derivative"""
    a1 = q[2]; a2 = a1 + q[3]; a3 = a2 + q[4]
    d1, d2, d3 = 1j * np.exp(1j * a1), 1j * np.exp(1j * a2), 1j * np.exp(1j * a3)
    cols = np.zeros((3, 5), complex)
    cols[:, 0] = 1.0
    cols[:, 1] = 1j
    # link1
    cols[0, 2] = 0.225 * d1
    # link2: p + .15 u1 - .075 u2
    cols[1, 2] = 0.15 * d1 - 0.075 * d2
    cols[1, 3] = -0.075 * d2
    # link3: p + .15 u1 - .15 u2 - .075 u3
    cols[2, 2] = 0.15 * d1 - 0.15 * d2 - 0.075 * d3
    cols[2, 3] = -0.15 * d2 - 0.075 * d3
    cols[2, 4] = -0.075 * d3
    J = np.zeros((3, 2, 5))
    J[:, 0, :] = cols.real
    J[:, 1, :] = cols.imag
    return J


def swimmer_mass_bias(q, v):
    m, I = _sw_capsule()
    J = _sw_jac(q)
    Jw = np.array([[0, 0, 1, 0, 0], [0, 0, 1, 1, 0], [0, 0, 1, 1, 1]], float)   # link angular rates
    M = sum(m * J[i].T @ J[i] + I * np.outer(Jw[i], Jw[i]) for i in range(3)) + SW['arm'] * np.eye(5)
    # velocity-product force  c = sum_i m J_i^T (dJ_i/dt v): differentiate J along v numerically
    eps = 1e-6
    Jp, Jm = _sw_jac(q + eps * v), _sw_jac(q - eps * v)
    Jdot = (Jp - Jm) / (2 * eps)
    c = sum(m * J[i].T @ (Jdot[i] @ v) for i in range(3))
    return M, c


def _sw_limit_rows(q, v, M):
    rows = []
    m0, _ = swimmer_mass_bias(np.zeros(5), np.zeros(5))
    A0 = np.linalg.inv(m0)
    tc = max(SW['solref'][0], 2 * SW['h'])
    dmin, dmax, width, mid, power = SW['solimp']
    b = 2 / (dmax * tc)
    k = 1 / (dmax * dmax * tc * tc * SW['solref'][1] ** 2)
    for j in (3, 4):
        dlo, dhi = q[j] + SW['lim'], SW['lim'] - q[j]
        pos = min(dlo, dhi)
        if pos >= 0:
            continue
        sg = 1.0 if dlo < dhi else -1.0
        x = abs(pos) / width
        y = (x ** power) / mid ** (power - 1) if x < mid else 1 - ((1 - x) ** power) / (1 - mid) ** (power - 1)
        imp = min(max(dmin + y * (dmax - dmin), dmin), dmax)
        if x > 1:
            imp = dmax
        Jr = np.zeros(5); Jr[j] = sg
        aref = -b * (Jr @ v) - k * imp * pos
        R = max(1e-15, (1 - imp) / imp * A0[j, j])
        rows.append((Jr, aref, 1 / R))
    return rows


def swimmer_step(q, v, ctrl):
    """one mjx.step: returns (pose of the robot body at the START state, qacc, q', v')"""
    q = np.asarray(q, float); v = np.asarray(v, float)
    M, c = swimmer_mass_bias(q, v)
    tau = np.zeros(5)
    tau[3:] = SW['gear'] * np.clip(ctrl, -1, 1)
    f = -c + tau
    a0 = np.linalg.solve(M, f)
    rows = _sw_limit_rows(q, v, M)
    qacc = a0
    if rows:
        # minimise 1/2 (a-a0)' M (a-a0) + sum_i 1/2 D_i min(0, J_i a - aref_i)^2 : active-set iteration
        active = [True] * len(rows)
        for _ in range(20):
            H = M.copy(); g = M @ a0
            for on, (Jr, aref, D) in zip(active, rows):
                if on:
                    H += D * np.outer(Jr, Jr); g += D * Jr * aref
            a = np.linalg.solve(H, g)
            new = [(Jr @ a - aref) < 0 for (Jr, aref, D) in rows]
            if new == active:
                break
            active = new
        qacc = a
    v2 = v + SW['h'] * qacc
    q2 = q + SW['h'] * v2
    pose = np.array([q[0], q[1], np.cos(q[2]), np.sin(q[2])])
    return pose, qacc, q2, v2
