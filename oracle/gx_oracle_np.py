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
def sample_layout(key, hazards_num=8, keepouts=(0.5, 0.4, 0.4), extents=(-2, -2, 2, 2), margin=0.0):
    names = ['goal'] + [f'hazard{i}' for i in range(hazards_num)] + ['robot']
    ko = {n: (keepouts[0] if n == 'goal' else keepouts[2] if n == 'robot' else keepouts[1]) for n in names}
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


def point_substep(q, v, ctrl):
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
    f = -damp * v - bias + float(GEAR) * ctrl
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
        pose, qacc, q, v = point_substep(q, v, ctrl)
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
