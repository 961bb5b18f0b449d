"""ctypes binding of the CPU restatement (oracle/libgx_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  guardx_amd/ never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgx_oracle.so")

# Engine.DEFAULT subset that the hot path reads (engine.py:98-204)
DEFAULTS = {
    'num_steps': 1000, 'env_num': 1, '_seed': 0,
    'placements_extents': [-2, -2, 2, 2], 'placements_margin': 0.0,
    'robot_keepout': 0.4, 'robot_base': 'xmls/point.xml', 'robot_rot': None,
    'observe_goal_lidar': True, 'observe_goal_comp': True, 'observe_hazards': True,
    'observe_qpos': True, 'observe_qvel': True, 'observe_ctrl': True,
    'observe_vel': False, 'observe_acc': False,
    'lidar_num_bins': 16, 'lidar_max_dist': None, 'lidar_exp_gain': 1.0, 'lidar_alias': True,
    'goal_keepout': 0.5, 'goal_size': 0.5, 'reward_distance': 1.0,
    'hazards_num': 8, 'hazards_keepout': 0.4, 'hazards_size': 0.3,
    'physics_steps_per_control_step': 1,
    'robot_placements': None, 'robot_locations': [], 'goal_placements': None, 'goal_locations': [],
    'hazards_placements': None, 'hazards_locations': [],
    # synthetic extension (BASELINE config 5; no reference counterpart, see gx_oracle.h)
    'pillars_num': 0, 'pillars_keepout': 0.3, 'pillars_size': 0.2, 'observe_pillars': False,
    'pillars_placements': None, 'pillars_locations': [],
}


class Config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("robot", C.c_int32), ("env_num", C.c_int32),
        ("env_total", C.c_int32), ("env_offset", C.c_int32), ("seed", C.c_uint32),
        ("num_steps", C.c_int32), ("hazards_num", C.c_int32), ("lidar_num_bins", C.c_int32),
        ("lidar_alias", C.c_int32), ("lidar_max_dist_set", C.c_int32),
        ("lidar_max_dist", C.c_float), ("lidar_exp_gain", C.c_float),
        ("goal_size", C.c_float), ("hazards_size", C.c_float), ("reward_distance", C.c_float),
        ("goal_keepout", C.c_double), ("hazards_keepout", C.c_double),
        ("robot_keepout", C.c_double), ("placements_margin", C.c_double),
        ("extents", C.c_double * 4),
        ("observe_goal_lidar", C.c_int32), ("observe_goal_comp", C.c_int32),
        ("observe_hazards", C.c_int32), ("observe_qpos", C.c_int32),
        ("observe_qvel", C.c_int32), ("observe_ctrl", C.c_int32),
        ("observe_vel", C.c_int32), ("observe_acc", C.c_int32),
        ("n_candidates", C.c_int32), ("physics_steps", C.c_int32),
        ("robot_goal_min_dist", C.c_float), ("reserved", C.c_int32),
        ("placements", C.POINTER(C.c_double)),
        ("pillars_num", C.c_int32), ("observe_pillars", C.c_int32), ("pillars_size", C.c_float),
        ("robot_rot", C.c_float), ("pillars_keepout", C.c_double),
    ]


def build(force=False):
    """(Re)build libgx_oracle.so with the committed Makefile."""
    src = os.path.join(_HERE, "gx_oracle.c")
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src),
                                                 os.path.getmtime(os.path.join(_HERE, "gx_oracle.h")))):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libgx_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp, u32p, i32p = C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_int32)
        L.gxo_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
        L.gxo_destroy.argtypes = [C.c_void_p]
        L.gxo_destroy.restype = None
        L.gxo_obs_dim.argtypes = [C.c_void_p]
        L.gxo_dims.argtypes = [C.c_void_p] + [C.POINTER(C.c_int32)] * 4
        L.gxo_dims.restype = None
        L.gxo_reset.argtypes = [C.c_void_p, fp]
        L.gxo_step.argtypes = [C.c_void_p, fp, fp, fp, fp, fp, fp]
        L.gxo_reset_done.argtypes = [C.c_void_p, fp]
        L.gxo_layout_size.argtypes = [C.c_void_p]
        L.gxo_get_state.argtypes = [C.c_void_p] + [fp] * 8 + [u32p, i32p]
        L.gxo_set_state.argtypes = [C.c_void_p] + [fp] * 8 + [u32p, i32p]
        L.gxo_get_pool.argtypes = [C.c_void_p, fp, C.c_int32]
        L.gxo_threefry2x32.argtypes = [C.c_uint32] * 4 + [u32p]
        L.gxo_threefry2x32.restype = None
        L.gxo_split.argtypes = [u32p, C.c_int32, u32p]
        L.gxo_split.restype = None
        L.gxo_uniform.argtypes = [u32p, C.c_float, C.c_float]
        L.gxo_uniform.restype = C.c_float
        L.gxo_randint.argtypes = [u32p, C.c_int32, C.c_uint32, i32p]
        L.gxo_randint.restype = None
        L.gxo_math_probe2.argtypes = [C.c_int32, fp, fp, fp]
        L.gxo_math_probe2.restype = None
        L.gxo_rollout_policy.argtypes = [C.c_void_p, C.c_int32, C.c_int32, fp, u32p, C.c_uint32] + [fp] * 12
        L.gxo_math_probe.argtypes = [C.c_int32] + [fp] * 6
        L.gxo_math_probe.restype = None
        L.gxo_ant_probe.argtypes = [fp] * 8
        L.gxo_ant_probe.restype = None
        L.gxo_walker_probe.argtypes = [fp] * 8
        L.gxo_walker_probe.restype = None
        L.gxo_set_threads.argtypes = [C.c_int32]
        L.gxo_set_threads.restype = None
        L.gxo_get_threads.restype = C.c_int32
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def make_config(config, n_candidates=1_000_000, env_total=None, env_offset=0, point_actuators='mjcf'):
    cfg = dict(DEFAULTS)
    cfg.update({k: v for k, v in config.items() if k in DEFAULTS})
    c = Config()
    c.struct_size = C.sizeof(Config)
    base = cfg['robot_base']
    c.robot = {'xmls/point.xml': 0, 'xmls/swimmer.xml': 1, 'xmls/ant.xml': 2, 'xmls/walker.xml': 3}.get(base, 99)
    if c.robot == 0 and point_actuators == 'bare':
        c.robot = 4          # round-1 reading of point.xml:37-39 (no class defaults), kept selectable
    c.env_num = int(cfg['env_num'])
    c.env_total = int(env_total if env_total is not None else cfg['env_num'])
    c.env_offset = int(env_offset)
    c.seed = int(cfg['_seed']) & 0xFFFFFFFF
    c.num_steps = int(cfg['num_steps'])
    c.hazards_num = int(cfg['hazards_num'])
    c.lidar_num_bins = int(cfg['lidar_num_bins'])
    c.lidar_alias = int(bool(cfg['lidar_alias']))
    c.lidar_max_dist_set = int(cfg['lidar_max_dist'] is not None)
    c.lidar_max_dist = float(cfg['lidar_max_dist'] or 0.0)
    c.lidar_exp_gain = float(cfg['lidar_exp_gain'])
    c.goal_size = float(cfg['goal_size'])
    c.hazards_size = float(cfg['hazards_size'])
    c.reward_distance = float(cfg['reward_distance'])
    c.goal_keepout = float(cfg['goal_keepout'])
    c.hazards_keepout = float(cfg['hazards_keepout'])
    c.robot_keepout = float(cfg['robot_keepout'])
    c.placements_margin = float(cfg['placements_margin'])
    for i in range(4):
        c.extents[i] = float(cfg['placements_extents'][i])
    for k in ('observe_goal_lidar', 'observe_goal_comp', 'observe_hazards', 'observe_qpos',
              'observe_qvel', 'observe_ctrl', 'observe_vel', 'observe_acc'):
        setattr(c, k, int(bool(cfg[k])))
    c.n_candidates = int(n_candidates)
    c.physics_steps = int(cfg['physics_steps_per_control_step'])
    c.robot_goal_min_dist = 3.0
    c.pillars_num = int(cfg['pillars_num'])
    c.observe_pillars = int(bool(cfg['observe_pillars']))
    c.pillars_size = float(cfg['pillars_size'])
    c.pillars_keepout = float(cfg['pillars_keepout'])
    c.robot_rot = 0.0 if cfg.get('robot_rot') is None else float(cfg['robot_rot'])   # engine.py:342-345
    # engine.py:507-531: per-object rectangle from *_locations (a +-keepout box around the point,
    # which the keepout shrink collapses back onto the point) or a single *_placements rectangle
    rows, custom = [], False
    for kind, count in (('goal', 1), ('hazards', c.hazards_num), ('pillars', c.pillars_num), ('robot', 1)):
        locs, rects, ko = cfg[kind + '_locations'], cfg[kind + '_placements'], cfg[kind + '_keepout']
        for i in range(count):
            if i < len(locs):
                x, y = locs[i]
                k = ko + 1e-9
                rows.append((x - k, y - k, x + k, y + k)); custom = True
            elif rects is not None:
                assert len(rects) == 1
                rows.append(tuple(rects[0])); custom = True
            else:
                rows.append(tuple(cfg['placements_extents']))
    if custom:
        arr = (C.c_double * (4 * len(rows)))(*[float(v) for r in rows for v in r])
        c._keep = arr
        c.placements = C.cast(arr, C.POINTER(C.c_double))
    return c


class OracleEngine:
    """numpy-facing mirror of Engine.reset/step/reset_done backed by the C restatement."""

    def __init__(self, config, n_candidates=1_000_000, env_total=None, env_offset=0, point_actuators='mjcf'):
        self.L = lib()
        self.cfg = make_config(config, n_candidates, env_total, env_offset, point_actuators)
        h = C.c_void_p()
        rc = self.L.gxo_create(C.byref(self.cfg), C.byref(h))
        if rc != 0:
            raise RuntimeError(f"gxo_create failed rc={rc}")
        self.h = h
        self.N = self.cfg.env_num
        self.H = self.cfg.hazards_num
        self.PL = self.cfg.pillars_num
        self.D = self.L.gxo_obs_dim(self.h)
        d = [C.c_int32() for _ in range(4)]
        self.L.gxo_dims(self.h, *[C.byref(x) for x in d])
        self.nq, self.nv, self.nu, self.na = (int(x.value) for x in d)

    def __del__(self):
        if getattr(self, 'h', None):
            self.L.gxo_destroy(self.h)
            self.h = None

    def reset(self, check=True):
        obs = np.empty((self.N, self.D), np.float32)
        rc = self.L.gxo_reset(self.h, _fp(obs))
        if check and rc != 0:
            raise AssertionError(f"layout_size {self.layout_size} <= env_num (engine.py:444) rc={rc}")
        return obs

    def step(self, action):
        a = np.ascontiguousarray(action, np.float32)
        assert a.shape == (self.N, self.na)
        obs = np.empty((self.N, self.D), np.float32)
        rew = np.empty(self.N, np.float32)
        cost = np.empty(self.N, np.float32)
        done = np.empty(self.N, np.float32)
        qacc = np.empty((self.N, self.nv), np.float32)
        rc = self.L.gxo_step(self.h, _fp(a), _fp(obs), _fp(rew), _fp(cost), _fp(done), _fp(qacc))
        assert rc == 0
        return obs, rew, done, {'cost': cost, 'qacc': qacc}

    def reset_done(self):
        obs = np.empty((self.N, self.D), np.float32)
        rc = self.L.gxo_reset_done(self.h, _fp(obs))
        assert rc == 0, rc
        return obs

    def rollout_policy(self, params, T, obs0, noise_seed=(0, 0), t0=0, hidden=64):
        N, D, A = self.N, self.D, self.na
        params = np.ascontiguousarray(params, np.float32)
        obs0 = np.ascontiguousarray(obs0, np.float32)
        f = np.float32
        out = dict(obs=np.empty((T, N, D), f), act=np.empty((T, N, A), f), logp=np.empty((T, N), f),
                   val=np.empty((T, N), f), mu=np.empty((T, N, A), f), rew=np.empty((T, N), f),
                   cost=np.empty((T, N), f), done=np.empty((T, N), f), obs_last=np.empty((N, D), f),
                   val_last=np.empty(N, f), logstd=np.empty(A, f))
        seed = (C.c_uint32 * 2)(int(noise_seed[0]), int(noise_seed[1]))
        rc = self.L.gxo_rollout_policy(self.h, T, int(hidden), _fp(params), seed, t0, _fp(obs0), *[_fp(out[k]) for k in
                                       ('obs', 'act', 'logp', 'val', 'mu', 'rew', 'cost', 'done', 'obs_last',
                                        'val_last', 'logstd')])
        assert rc == 0, rc
        return out

    @property
    def layout_size(self):
        return self.L.gxo_layout_size(self.h)

    def get_state(self):
        N, H = self.N, self.H
        s = {
            'qpos': np.empty((N, self.nq), np.float32), 'qvel': np.empty((N, self.nv), np.float32),
            'pose0': np.empty((N, 4), np.float32), 'pose1': np.empty((N, 2), np.float32),
            'objs': np.empty((N, 1 + H + self.PL, 2), np.float32), 'done0': np.empty(N, np.float32),
            'done1': np.empty(N, np.float32), 'steps': np.empty(N, np.float32),
        }
        key = (C.c_uint32 * 2)()
        hist = C.c_int32()
        self.L.gxo_get_state(self.h, *[_fp(s[k]) for k in
                                       ('qpos', 'qvel', 'pose0', 'pose1', 'objs', 'done0', 'done1', 'steps')],
                             key, C.byref(hist))
        s['key'] = np.array([key[0], key[1]], np.uint32)
        s['hist'] = int(hist.value)
        return s

    def set_state(self, s):
        arrs = []
        for k in ('qpos', 'qvel', 'pose0', 'pose1', 'objs', 'done0', 'done1', 'steps'):
            v = s.get(k)
            arrs.append(None if v is None else np.ascontiguousarray(v, np.float32))
        key = None
        if s.get('key') is not None:
            key = (C.c_uint32 * 2)(int(s['key'][0]), int(s['key'][1]))
        hist = None
        if s.get('hist') is not None:
            hist = C.byref(C.c_int32(int(s['hist'])))
        self.L.gxo_set_state(self.h, *[_fp(a) for a in arrs], key, hist)

    def get_pool(self, max_rows=None):
        n = self.layout_size if max_rows is None else min(max_rows, self.layout_size)
        pool = np.empty((max(n, 1), self.H + self.PL + 2, 2), np.float32)
        got = self.L.gxo_get_pool(self.h, _fp(pool), n)
        return pool[:got]


# ---- probes ---------------------------------------------------------------
def threefry2x32(k0, k1, x0, x1):
    out = (C.c_uint32 * 2)()
    lib().gxo_threefry2x32(k0, k1, x0, x1, out)
    return int(out[0]), int(out[1])


def split(key, n=2):
    k = (C.c_uint32 * 2)(int(key[0]), int(key[1]))
    out = (C.c_uint32 * (2 * n))()
    lib().gxo_split(k, n, out)
    return np.array(list(out), np.uint32).reshape(n, 2)


def uniform(key, lo, hi):
    k = (C.c_uint32 * 2)(int(key[0]), int(key[1]))
    return float(lib().gxo_uniform(k, lo, hi))


def randint(key, n, span):
    k = (C.c_uint32 * 2)(int(key[0]), int(key[1]))
    out = np.empty(n, np.int32)
    lib().gxo_randint(k, n, span, out.ctypes.data_as(C.POINTER(C.c_int32)))
    return out


def ant_probe(q, v, ctrl):
    """one ant.xml mjx.step: (q2, v2, qacc, pose, dense mass matrix, smooth force), qpos coordinates"""
    q = np.ascontiguousarray(q, np.float32); v = np.ascontiguousarray(v, np.float32)
    ctrl = np.ascontiguousarray(ctrl, np.float32)
    q2 = np.zeros(11, np.float32); v2 = np.zeros(11, np.float32); qacc = np.zeros(11, np.float32)
    pose = np.zeros(4, np.float32); dbg = np.zeros(132, np.float32)
    lib().gxo_ant_probe(_fp(q), _fp(v), _fp(ctrl), _fp(q2), _fp(v2), _fp(qacc), _fp(pose), _fp(dbg))
    return q2, v2, qacc, pose, dbg[:121].reshape(11, 11).copy(), dbg[121:].copy()


def walker_probe(q, v, ctrl):
    """one walker.xml mjx.step: (q2, v2, qacc, pose, dense mass matrix, smooth force)"""
    q = np.ascontiguousarray(q, np.float32); v = np.ascontiguousarray(v, np.float32)
    ctrl = np.ascontiguousarray(ctrl, np.float32)
    q2 = np.zeros(13, np.float32); v2 = np.zeros(13, np.float32); qacc = np.zeros(13, np.float32)
    pose = np.zeros(4, np.float32); dbg = np.zeros(169 + 13, np.float32)
    lib().gxo_walker_probe(_fp(q), _fp(v), _fp(ctrl), _fp(q2), _fp(v2), _fp(qacc), _fp(pose), _fp(dbg))
    return q2, v2, qacc, pose, dbg[:169].reshape(13, 13).copy(), dbg[169:].copy()


def math_probe2(x):
    x = np.ascontiguousarray(x, np.float32)
    lg, th = np.empty(x.size, np.float32), np.empty(x.size, np.float32)
    lib().gxo_math_probe2(x.size, _fp(x), _fp(lg), _fp(th))
    return lg, th


def math_probe(x, y):
    x = np.ascontiguousarray(x, np.float32)
    y = np.ascontiguousarray(y, np.float32)
    n = x.size
    s, c, a, e = (np.empty(n, np.float32) for _ in range(4))
    lib().gxo_math_probe(n, _fp(x), _fp(y), _fp(s), _fp(c), _fp(a), _fp(e))
    return s, c, a, e
