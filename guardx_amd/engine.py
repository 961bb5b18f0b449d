"""Batched GUARD environment on MI355X behind the reference `Engine` interface.

Mirrors `safe_rl_envs.envs.engine.Engine` of intelligent-control-lab/guardX
(reference safe_rl_envs/safe_rl_envs/envs/engine.py): same constructor config
dict, `reset() / step(action) / reset_done()`, `observation_space`,
`action_space`, and the `(obs, reward, done, info{'cost','obs'})` float32 torch
tensor surface (engine.py:454-505), so the safe_rl_libX learners drive it
unchanged (safe_rl_libX/trpo/trpo.py:449-547).

All arithmetic runs in hand-written HIP kernels through the C ABI of
`libguardx_hip.so` (include/guardx.h).  PyTorch only provides device memory and
the current stream.  There is no CPU fallback.
"""
from collections import OrderedDict
from copy import deepcopy
import ctypes as C
import os

import numpy as np
import torch

from . import _native
from .spaces import Box

__all__ = ["Engine", "ResamplingError"]


class ResamplingError(AssertionError):
    """Raised when no valid object layout could be sampled (engine.py:79-81)."""


# robot_base -> (native robot id, nq, nv, nu, z_height, timestep, action low/high)
#   nq/nv/nu/z_height as world.py:422-438 reads them from the robot-only MJCF; action bounds =
#   actuator ctrlrange where ctrllimited, else +-inf, Point keeps its first 2 rows (engine.py:291-297)
_ROBOTS = {
    # point.xml:3,7-8,16-18,37-39: the <general> actuators inherit ctrllimited / ctrlrange +-1 from the class
    # default (written by <motor>, <velocity>), so actuator_ctrllimited == 1 and the Box is [-1, 1]^2
    'xmls/point.xml': (0, 3, 3, 3, 0.1, 0.02, (-1.0, 1.0, 2)),
    'xmls/swimmer.xml': (1, 5, 5, 2, 0.03, 0.03, (-1.0, 1.0, 2)),         # swimmer.xml:3,14,58-59
    'xmls/ant.xml': (2, 11, 11, 8, 0.15, 0.09, (-1.0, 1.0, 8)),           # ant.xml:2,7,14,137-146
    'xmls/walker.xml': (3, 13, 13, 10, 0.42, 0.02, (-1.0, 1.0, 10)),      # walker.xml:9,12,99-111
}


class _LazyObsDict(dict):
    """info['obs']: per-key views of the flat observation (engine.py:693-695), created on first
    access -- the learners never read them (only render() does, engine.py:1054)."""

    def __init__(self, obs, slices, qacc):
        super().__init__()
        self._src = (obs, slices, qacc)

    def _fill(self):
        if self._src is not None:
            obs, slices, qacc = self._src
            self._src = None
            for k, s in slices.items():
                dict.__setitem__(self, k, obs[:, s])
            if qacc is not None:
                flat, i, lo, n, nv = qacc      # the step's qacc lives in the output slab: viewed only when asked for
                dict.__setitem__(self, 'qacc', flat[i, lo:lo + n * nv].view(n, nv))

    def __getitem__(self, k):
        self._fill()
        return dict.__getitem__(self, k)

    def __iter__(self):
        self._fill()
        return dict.__iter__(self)

    def __len__(self):
        self._fill()
        return dict.__len__(self)

    def __contains__(self, k):
        self._fill()
        return dict.__contains__(self, k)

    def keys(self):
        self._fill()
        return dict.keys(self)

    def items(self):
        self._fill()
        return dict.items(self)

    def values(self):
        self._fill()
        return dict.values(self)

    def get(self, k, default=None):
        self._fill()
        return dict.get(self, k, default)

    def __repr__(self):
        self._fill()
        return dict.__repr__(self)


class _StepInfo(dict):
    """The `info` dict of step() (engine.py:693-695: {'cost': ..., 'obs': {...}}).  'cost' is there from the start -- the
    learners read it every step (trpo.py:484) --, 'obs' (the per-key views, which only render() reads) is built on first
    use: step() + reset_done() is host bound at env_num = 2000 and every Python object made per call counts."""
    __slots__ = ('_src',)

    def _fill(self):
        src = getattr(self, '_src', None)       # (an instance made any other way than by step() has no source)
        if src is not None:
            self._src = None
            dict.__setitem__(self, 'obs', _LazyObsDict(*src))

    # every entry point of dict that reads or removes entries goes through _fill() first, so the object behaves like
    # the reference's plain {'cost': ..., 'obs': {...}} (engine.py:693-695) whatever the caller does with it
    def copy(self):
        self._fill()
        return dict(self)

    def pop(self, *a):
        self._fill()
        return dict.pop(self, *a)

    def popitem(self):
        self._fill()
        return dict.popitem(self)

    def setdefault(self, k, default=None):
        self._fill()
        return dict.setdefault(self, k, default)

    def __or__(self, other):
        self._fill()
        return dict(self) | other

    def __ror__(self, other):
        self._fill()
        return other | dict(self)

    def __ior__(self, other):
        self._fill()
        dict.update(self, other)
        return self

    def __reversed__(self):
        self._fill()
        return dict.__reversed__(self)

    def __missing__(self, k):
        self._fill()
        return dict.__getitem__(self, k)       # KeyError for anything but 'obs', as a plain dict

    def __iter__(self):
        self._fill()
        return dict.__iter__(self)

    def __len__(self):
        self._fill()
        return dict.__len__(self)

    def __contains__(self, k):
        self._fill()
        return dict.__contains__(self, k)

    def keys(self):
        self._fill()
        return dict.keys(self)

    def items(self):
        self._fill()
        return dict.items(self)

    def values(self):
        self._fill()
        return dict.values(self)

    def get(self, k, default=None):
        self._fill()
        return dict.get(self, k, default)

    def __repr__(self):
        self._fill()
        return dict.__repr__(self)

    def __eq__(self, other):
        self._fill()
        return dict.__eq__(self, other)

    __hash__ = None


class Engine:
    """GUARD `Engine`: `env_num` independent Goal-task arenas stepped in lock-step.

    Extra keyword arguments (not part of the reference config dict):
      n_candidates  number of layout candidates drawn per reset (engine.py:263: 1e6)
      shard         (rank, world): this instance owns the `rank`-th contiguous slice
                    of a global batch of `env_num * world` envs and reproduces exactly
                    the rows an unsharded Engine(env_num*world) would produce
      emit_qacc     fill info['obs']['qacc'] (engine.py:763-764); costs one more output array
      out_ring      0 (default): the tensors step() returns are never written again, as in the reference
                    (engine.py:495 hands out fresh buffers; trpo.py:529 mutates the obs it keeps in place) -- they
                    are views of a slab of at most 64 MB (up to 256 calls' outputs) allocated in one piece and released
                    when the last view dies.
                    k > 0: opt-in ring of k preallocated output sets, a tensor is overwritten k step() calls
                    after it was returned (for callers that copy what they keep, trpo.py:58-64; saves the slab
                    allocations)
      point_actuators  'mjcf' (default): point.xml's <general> actuators inherit the class defaults the way
                    MuJoCo compiles them (ctrl clamp +-1, velocity-servo bias, force clamp +-.05, action
                    space Box(-1, 1)); 'bare': the round-1 reading without the defaults (force = 0.3*ctrl,
                    unbounded actions) -- see DESIGN.md section 0
    """

    # Interface restatement of Engine.DEFAULT (engine.py:98-204): the key set is
    # the constructor contract -- unknown keys assert "Bad key" (engine.py:327).
    DEFAULT = {
        'num_steps': 1000, 'device_id': 0, 'env_num': 1,
        'placements_extents': [-2, -2, 2, 2], 'placements_margin': 0.0,
        'floor_display_mode': False,
        'robot_placements': None, 'robot_locations': [], 'robot_keepout': 0.4,
        'robot_base': 'xmls/point.xml', 'robot_rot': None,
        'observation_flatten': True, 'observe_goal_lidar': True, 'observe_goal_comp': True,
        'observe_hazards': True, 'observe_qpos': True, 'observe_qvel': True,
        'observe_qacc': True, 'observe_vel': False, 'observe_acc': False,
        'observe_ctrl': True, 'observe_vision': False,
        'render_labels': False, 'render_lidar_markers': True, 'render_lidar_radius': 0.15,
        'render_lidar_size': 0.025, 'render_lidar_offset_init': 0.5,
        'render_lidar_offset_delta': 0.06,
        'sensors_obs': ['accelerometer', 'velocimeter', 'gyro', 'magnetometer'],
        'sensors_hinge_joints': True, 'sensors_ball_joints': True,
        'sensors_angle_components': True,
        'lidar_num_bins': 16, 'lidar_num_bins3D': 1, 'lidar_max_dist': None,
        'lidar_exp_gain': 1.0, 'lidar_type': 'pseudo', 'lidar_alias': True,
        'lidar_body': ['robot'],
        'task': 'goal', 'push_object': 'box', 'goal_mode': 'random', 'goal_travel': 3.0,
        'goal_velocity': 0.5,
        'goal_placements': None, 'goal_locations': [], 'goal_keepout': 0.5, 'goal_size': 0.5,
        'goal_3D': False, 'goal_z_range': [1.0, 1.0],
        'reward_distance': 1.0, 'reward_goal': 1.0, 'reward_box_dist': 1.0,
        'reward_box_goal': 1.0, 'reward_orientation': False, 'reward_orientation_scale': 0.002,
        'reward_orientation_body': 'robot', 'reward_exception': -10.0, 'reward_x': 1.0,
        'reward_z': 1.0, 'reward_circle': 1e-1, 'reward_clip': 10, 'reward_defense': 1.0,
        'reward_chase': 1.0,
        'constrain_hazards': False, 'constrain_indicator': True,
        'hazards_num': 8, 'hazards_placements': None, 'hazards_locations': [],
        'hazards_keepout': 0.4, 'hazards_size': 0.3, 'hazards_cost': 1.0,
        'physics_steps_per_control_step': 1,
        '_seed': 0,
    }

    # Synthetic extension -- NOT part of the reference (which asserts "Bad key" on every one of these):
    # BASELINE.json config 5 asks for "8 pillars" next to the hazards; the reference only has the colour /
    # lidar-group constants (engine.py:38,56).  Pillars are static circles in the hazard style: placed by the
    # layout sampler after the hazards, observed through 'pillars_lidar', costed like hazards with
    # pillars_size.  With pillars_num == 0 (default) the Engine is exactly the reference's task.
    EXTENSIONS = {
        'pillars_num': 0, 'pillars_placements': None, 'pillars_locations': [], 'pillars_keepout': 0.3,
        'pillars_size': 0.2, 'observe_pillars': False,
    }

    # step() outputs are carved out of one allocation per `k` calls, k sized by BYTES: as many output sets as fit
    # _SLAB_BYTES, at most _SLAB_STEPS, at least one (env_num = 2000: 85 sets of 0.75 MB; 2^22 Point envs: one 1.6 GB set
    # per call, as if every step allocated its own outputs).  A tensor a caller retains keeps at most one slab alive.
    _SLAB_BYTES = 64 << 20
    _SLAB_STEPS = 256

    def __init__(self, config={}, *, n_candidates=1_000_000, shard=None, emit_qacc=True, point_actuators='mjcf',
                 out_ring=0):
        self._ctor_config = deepcopy(config)
        self._ctor_kwargs = dict(n_candidates=n_candidates, shard=shard, emit_qacc=emit_qacc,
                                 point_actuators=point_actuators, out_ring=out_ring)
        if point_actuators not in ('mjcf', 'bare'):
            raise ValueError("point_actuators must be 'mjcf' or 'bare'")
        self.parse(config)
        self._h = None
        self._lib = _native.load()

        if self.task != 'goal':
            # reference: build_placements_dict only places a goal for task 'goal'
            # (engine.py:538) and sample_layout then fails on layout['goal'] (:570)
            raise KeyError('goal')
        if not self.hazards_num:
            # reference: build_world_config returns None (engine.py:370-384) -> World(None) fails
            raise TypeError("hazards_num == 0: build_world_config() returns None in the reference")
        if self.robot_base not in _ROBOTS:
            raise NotImplementedError(f"robot_base {self.robot_base!r}: only {sorted(_ROBOTS)} "
                                      "have HIP dynamics in this build")
        if not self.observation_flatten:
            raise NotImplementedError("observation_flatten=False: Engine.obs leaves flat_obs "
                                      "undefined in the reference (engine.py:773-778)")
        if self.observe_vision:
            raise NotImplementedError("observe_vision is not part of the batched path")
        robot_id, nq, nv, nu, z_height, timestep, (act_lo, act_hi, act_dim) = _ROBOTS[self.robot_base]
        if robot_id == 0 and point_actuators == 'bare':
            robot_id, act_lo, act_hi = 4, -np.inf, np.inf
        self.robot = type('Robot', (), dict(nq=nq, nv=nv, nu=nu, z_height=z_height))()

        if not torch.cuda.is_available():
            raise RuntimeError("guardx_amd.Engine needs a HIP device (no CPU fallback)")
        self.device = torch.device('cuda', int(self.device_id))
        self.emit_qacc = bool(emit_qacc)

        rank, world = (0, 1) if shard is None else (int(shard[0]), int(shard[1]))
        assert 0 <= rank < world
        self.shard = (rank, world)

        self.build_placements_dict()
        cfg = self._native_config(robot_id, int(n_candidates), rank, world)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _native.check(self._lib.gx_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self._cfg = cfg

        self.dt = timestep * self.physics_steps_per_control_step  # engine.py:235
        self.action_space = Box(np.full(act_dim, act_lo, np.float32),
                                np.full(act_dim, act_hi, np.float32), dtype=np.float32)  # :291-297
        self.build_observation_space()
        assert self.obs_flat_size == self._lib.gx_obs_dim(self._h)

        self._act_shape = torch.Size((int(self.env_num), act_dim))
        self._out_ring = max(0, int(out_ring))
        self._slab, self._slab_i = None, 0
        self._per_set = self._slab_floats()
        self._speculate = os.environ.get("GX_NO_SPECULATE", "0") != "1"
        # bit 0: write qacc; bit 1: speculate reset_done in the step launch (off: two-launch form, debugging / A-B timing)
        self._step_flags = (1 if self.emit_qacc else 0) | (2 if self._speculate else 0)
        # the per-step entry points, looked up once (step() + reset_done() is host-bound at env_num = 2000)
        self._gx_step_slab = self._lib.gx_step_slab
        self._gx_commit = self._lib.gx_reset_done_commit
        self._raw_stream = torch._C._cuda_getCurrentRawStream
        self._dev_index = self.device.index
        self._qacc_in_info = bool(self.observe_qacc)
        self._spec = C.c_int32(0)
        self._spec_ref = C.byref(self._spec)
        self._rd_obs = None          # what reset_done() returns for the step just made (speculated in-kernel)
        self._obs = None
        self._reward = None
        self._done = None
        self._info = None
        self.layout_size = None
        self.viewer = None

    # ------------------------------------------------------------------
    # configuration
    # ------------------------------------------------------------------
    def parse(self, config):
        """Constructor contract of engine.py:322-328: unknown keys fail with AssertionError('Bad key <key>'), every
        accepted key becomes an attribute and self.config the merged dict."""
        known = {**self.DEFAULT, **self.EXTENSIONS}
        for key in config:
            assert key in known, f'Bad key {key}'
        self.config = deepcopy({**known, **config})
        vars(self).update(self.config)

    # what the layout sampler places, in placement order (engine.py:533-544): (object family, is it a numbered family)
    _PLACED = (('goal', False), ('hazard', True), ('pillar', True), ('robot', False))

    def build_placements_dict(self):
        """name -> (rectangles or None, keepout) in placement order: goal, hazard0.., [pillar0..,] robot
        (engine.py:507-544).  A fixed location i < len(<family>_locations) is a degenerate rectangle of half-width
        keepout + 1e-9 around it (engine.py:524-526); otherwise the family's *_placements (None = the arena)."""
        table = OrderedDict()
        for family, numbered in self._PLACED:
            stem = family + 's' if numbered else family
            count = int(self.config[stem + '_num']) if numbered else 1
            keepout = self.config[stem + '_keepout']
            fixed = list(self.config[stem + '_locations'])[:count]
            pad = keepout + 1e-9
            for i in range(count):
                name = f'{family}{i}' if numbered else family
                if i < len(fixed):
                    cx, cy = fixed[i]
                    table[name] = ([(cx - pad, cy - pad, cx + pad, cy + pad)], keepout)
                else:
                    table[name] = (self.config[stem + '_placements'], keepout)
        self.placements = table

    def _native_config(self, robot_id, n_candidates, rank, world):
        c = _native.GxConfig()
        c.struct_size = C.sizeof(_native.GxConfig)
        c.robot = robot_id
        c.env_num = int(self.env_num)
        c.env_total = int(self.env_num) * world
        c.env_offset = int(self.env_num) * rank
        c.seed = int(self._seed) & 0xFFFFFFFF
        c.num_steps = int(self.num_steps)
        c.hazards_num = int(self.hazards_num)
        c.lidar_num_bins = int(self.lidar_num_bins)
        c.lidar_alias = int(bool(self.lidar_alias))
        c.lidar_max_dist_set = int(self.lidar_max_dist is not None)
        c.lidar_max_dist = float(self.lidar_max_dist or 0.0)
        c.lidar_exp_gain = float(self.lidar_exp_gain)
        c.goal_size = float(self.goal_size)
        c.hazards_size = float(self.hazards_size)
        c.reward_distance = float(self.reward_distance)
        c.goal_keepout = float(self.goal_keepout)
        c.hazards_keepout = float(self.hazards_keepout)
        c.robot_keepout = float(self.robot_keepout)
        c.placements_margin = float(self.placements_margin)
        for i in range(4):
            c.extents[i] = float(self.placements_extents[i])
        # explicit *_placements / *_locations (engine.py:507-531): one rectangle per object
        self._placements_arr = None
        if any(rects is not None for rects, _ in self.placements.values()):
            rows = []
            for name, (rects, _) in self.placements.items():
                if rects is None:
                    rows.append([float(v) for v in self.placements_extents])
                elif len(rects) == 1:
                    rows.append([float(v) for v in rects[0]])
                else:
                    # draw_placement's multi-rectangle branch calls self.rs.choice (engine.py:616)
                    raise AttributeError("'Engine' object has no attribute 'rs'")
            arr = (C.c_double * (4 * len(rows)))(*[v for r in rows for v in r])
            self._placements_arr = arr          # keep alive until gx_create returns
            c.placements = C.cast(arr, C.POINTER(C.c_double))
        c.observe_goal_lidar = int(bool(self.observe_goal_lidar))
        c.observe_goal_comp = int(bool(self.observe_goal_comp))
        c.observe_hazards = int(bool(self.observe_hazards))
        c.observe_qpos = int(bool(self.observe_qpos))
        c.observe_qvel = int(bool(self.observe_qvel))
        c.observe_ctrl = int(bool(self.observe_ctrl))
        c.observe_vel = int(bool(self.observe_vel))
        c.observe_acc = int(bool(self.observe_acc))
        c.n_candidates = n_candidates
        c.physics_steps = int(self.physics_steps_per_control_step)
        c.robot_goal_min_dist = 3.0  # engine.py:571
        c.pillars_num = int(self.pillars_num)
        c.observe_pillars = int(bool(self.observe_pillars))
        c.pillars_size = float(self.pillars_size)
        c.pillars_keepout = float(self.pillars_keepout)
        # engine.py:342-345: None -> random_rot(), which returns 0.0 (engine.py:330-333); else float(robot_rot)
        c.robot_rot = 0.0 if self.robot_rot is None else float(self.robot_rot)
        c.device = int(self.device_id)
        return c

    def _obs_table(self):
        """One row per observation component, in the reference's insertion order (engine.py:386-407):
        (key, enabled, width, low, high).  The flat observation concatenates the enabled rows in sorted-key order
        (engine.py:773-777)."""
        b, r, inf = int(self.lidar_num_bins), self.robot, np.inf
        return (
            ('goal_lidar', self.observe_goal_lidar, b, 0.0, 1.0),
            ('goal_compass', self.observe_goal_comp, 2, -inf, inf),
            ('hazards_lidar', self.observe_hazards, b, 0.0, 1.0),
            ('pillars_lidar', self.observe_pillars and self.pillars_num, b, 0.0, 1.0),   # synthetic extension
            ('qpos', self.observe_qpos, r.nq, -inf, inf),
            ('qvel', self.observe_qvel, r.nv, -inf, inf),
            ('ctrl', self.observe_ctrl, r.nu, -inf, inf),
            ('vel', self.observe_vel, 2, -inf, inf),
            ('acc', self.observe_acc, 2, -inf, inf),
        )

    def build_observation_space(self):
        rows = [row for row in self._obs_table() if row[1]]
        self.obs_space_dict = OrderedDict((k, Box(lo, hi, (w,), dtype=np.float32)) for k, _, w, lo, hi in rows)
        self.obs_flat_size = sum(w for _, _, w, _, _ in rows)
        self.observation_space = Box(-np.inf, np.inf, (self.obs_flat_size,), dtype=np.float32)
        self._obs_slices = OrderedDict()            # column ranges of the flat observation
        col = 0
        for k, _, w, _, _ in sorted(rows, key=lambda row: row[0]):
            self._obs_slices[k] = slice(col, col + w)
            col += w

    # ------------------------------------------------------------------
    # gym-style interface
    # ------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch._C._cuda_getCurrentRawStream(self.device.index))

    def _new(self, *shape):
        return torch.empty(shape, dtype=torch.float32, device=self.device)

    def reset(self, check=True):
        """Resample every layout and return the (env_num, obs_dim) observation (engine.py:454-467).

        check=False defers the `layout_size > env_num` assert (engine.py:444) to the next
        check_layouts() call, so that no host round trip separates consecutive epochs."""
        obs = self._new(self.env_num, self.obs_flat_size)
        _native.check(self._lib.gx_reset(self._h, obs.data_ptr(), self._stream()))
        self._rd_obs = None
        if not check:
            self._obs = obs
            return obs
        n = C.c_int32()
        st = self._lib.gx_layout_size(self._h, C.byref(n))
        self.layout_size = int(n.value)
        if st == _native.GX_ERR_LAYOUT:
            # engine.py:444  assert self.layout_size > self.env_num
            raise ResamplingError(f"number of valid layout is {self.layout_size} "
                                  f"<= env_num {self._cfg.env_total}")
        _native.check(st)
        self._obs = obs
        return obs

    def check_layouts(self):
        """The deferred engine.py:444 assert for every reset(check=False) since the last call."""
        n = C.c_int32()
        st = self._lib.gx_layout_size_min(self._h, C.byref(n))
        if st == _native.GX_ERR_LAYOUT:
            raise ResamplingError(f"number of valid layout is {int(n.value)} <= env_num {self._cfg.env_total}")
        _native.check(st)
        self.layout_size = int(n.value)
        return self.layout_size

    def _slab_floats(self):
        """floats of one set of step() outputs (obs, obs_rd, reward, cost, done, qacc), every piece 16-byte aligned:
        the layout gx_step_slab addresses (include/guardx.h)"""
        n = C.c_int64()
        _native.check(self._lib.gx_step_set_floats(self._h, C.byref(n)))
        return int(n.value)

    def _slab_steps(self):
        return max(1, min(self._SLAB_STEPS, self._SLAB_BYTES // (4 * self._slab_floats())))

    def _out_slab(self, k):
        """`k` sets of step() outputs carved out of ONE allocation: (obs, obs_rd, reward, cost, done, qacc) tuples of k
        views each -- six unbind() calls, not 6 k slicing operations -- and the base address; the kernel addresses set i
        itself (gx_step_slab), so no per-set pointer objects are made."""
        N, D, nv = self.env_num, self.obs_flat_size, self.robot.nv
        Dp = (D + 3) // 4 * 4                      # keep every piece 16-byte aligned
        Np = (N + 3) // 4 * 4
        per = self._per_set
        flat = torch.empty(k, per, dtype=torch.float32, device=self.device)
        o = 0
        obs = flat[:, o:o + N * D].view(k, N, D).unbind(0); o += N * Dp
        obs_rd = flat[:, o:o + N * D].view(k, N, D).unbind(0); o += N * Dp
        rew = flat[:, o:o + N].unbind(0); o += Np
        cost = flat[:, o:o + N].unbind(0); o += Np
        done = flat[:, o:o + N].unbind(0); o += Np
        # qacc (engine.py:763-764; not part of the flat observation) is viewed on demand: (slab, set, offset, N, nv)
        return (obs, obs_rd, rew, cost, done, (flat, o) if self.emit_qacc and self._qacc_in_info else None,
                flat.data_ptr(), k)

    def step(self, action):
        """One control step for every env (engine.py:469-495).  No auto-reset.  The same launch also
        evaluates what reset_done() would return for the envs this step finished (gx_step_rd); nothing is
        re-initialised unless reset_done() is called."""
        a = action
        if not (type(a) is torch.Tensor and a.dtype is torch.float32 and a.shape == self._act_shape
                and a.device == self.device and a.is_contiguous() and not a.requires_grad):
            a = self._as_action(action)
        i = self._slab_i
        slab = self._slab
        if slab is None or i >= slab[7]:
            # out_ring == 0 (default): a NEW slab -- tensors already handed out are never written again
            # (engine.py:495 returns fresh buffers); out_ring > 0: wrap around and reuse the ring
            if not self._out_ring or slab is None:
                slab = self._slab = self._out_slab(self._out_ring or self._slab_steps())
            i = 0
        self._slab_i = i + 1
        # (a CPython shim calling the same entry point with plain integers instead of ctypes was measured in round 4:
        # 8.1 -> 8.0 us per call -- the cost is hipLaunchKernel's own ~3.5 us, not the argument conversion; not kept)
        st = self._gx_step_slab(self._h, a.data_ptr(), slab[6], i, self._step_flags, self._spec_ref,
                                self._raw_stream(self._dev_index))
        if st:
            _native.check(st)
        obs, reward, cost, done = slab[0][i], slab[2][i], slab[3][i], slab[4][i]
        self._rd_obs = slab[1][i] if self._spec.value else None
        info = _StepInfo(cost=cost)
        q = slab[5]
        info._src = (obs, self._obs_slices, None if q is None else (q[0], i, q[1], self.env_num, self.robot.nv))
        self._obs, self._reward, self._done, self._info = obs, reward, done, info
        return obs, reward, done, info

    def reset_done(self):
        """Re-initialise the envs whose last `done` was set; other rows keep the last
        step's observation (engine.py:497-505)."""
        if self._obs is None:
            raise RuntimeError("reset_done() before reset()")
        if self._rd_obs is not None:
            # already evaluated by the step() launch: request the re-initialisation (installed by the next
            # launch on this engine) and hand out the observation -- no kernel of its own
            st = self._gx_commit(self._h)
            if st:
                _native.check(st)
            return self._rd_obs
        out = self._new(self.env_num, self.obs_flat_size)
        _native.check(self._lib.gx_reset_done(self._h, self._obs.data_ptr(), out.data_ptr(),
                                              self._stream()))
        return out

    def rollout(self, actions, packed=False):
        """Open-loop rollout: T x (step -> reset_done) driven by `actions` (T, env_num, act_dim).

        Returns time-major tensors obs (T, N, D) [post-reset_done, i.e. what the learner stores
        as the next observation], reward, cost, done (T, N).  Equivalent to the learner loop of
        safe_rl_libX/trpo/trpo.py:466-547 with a fixed action tape.

        packed=True: the kernel writes ONE (T, N, D + A + 3) tensor, rows (obs | action | reward, cost, done)
        -- the per-rank shard of the learner hand-off, ready for a single all-gather; the four results are
        views of it and it is returned as a fifth value."""
        a = actions
        if a.dtype != torch.float32 or not a.is_contiguous() or a.device != self.device:
            a = a.to(device=self.device, dtype=torch.float32).contiguous()
        T = int(a.shape[0])
        A = self.action_space.shape[0]
        assert tuple(a.shape[1:]) == (self.env_num, A)
        N, D = self.env_num, self.obs_flat_size
        self._rd_obs = None
        if packed:
            W = D + A + 3
            pk = self._new(T, N, W)
            _native.check(self._lib.gx_rollout_packed(self._h, T, a.data_ptr(), pk.data_ptr(), self._stream()))
            obs, reward, cost, done = pk[..., :D], pk[..., D + A], pk[..., D + A + 1], pk[..., D + A + 2]
        else:
            obs = self._new(T, N, D)
            reward, cost, done = self._new(T, N), self._new(T, N), self._new(T, N)
            _native.check(self._lib.gx_rollout(self._h, T, a.data_ptr(), obs.data_ptr(),
                                               reward.data_ptr(), cost.data_ptr(), done.data_ptr(),
                                               self._stream()))
        self._obs, self._reward, self._done = obs[-1], reward[-1], done[-1]
        self._info = {'cost': cost[-1]}
        return (obs, reward, cost, done, pk) if packed else (obs, reward, cost, done)

    # ------------------------------------------------------------------
    # tape hand-off (multi-GPU): step here, build the observations wherever the rollout is needed
    # ------------------------------------------------------------------
    def tape_floats(self, T):
        """(tape, layouts, entry records) float counts of one shard buffer of rollout_tape(T)."""
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        _native.check(self._lib.gx_tape_floats(self._h, int(T), C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def rollout_tape(self, actions, out=None):
        """The serial half of rollout(): T x (step -> reset_done) without building observations.  Returns
        (shard, token): `shard` a flat float32 tensor [tape | layouts at entry | entry records] (36 B per env-step
        for the Point against 192 B of packed rows) to all-gather as is, `token` naming the layout pool in effect.
        expand_tape(shard, token) -- here or on any rank's engine of the same configuration -- gives the packed
        (T, N, D + A + 3) rows of rollout(packed=True), bit for bit; call it before the second reset() after
        this rollout.  step() / reset_done() may follow, but the last observation is only known after expand_tape."""
        a = actions
        if a.dtype != torch.float32 or not a.is_contiguous() or a.device != self.device:
            a = a.to(device=self.device, dtype=torch.float32).contiguous()
        T = int(a.shape[0])
        assert tuple(a.shape[1:]) == (self.env_num, self.action_space.shape[0])
        n = sum(self.tape_floats(T))
        if out is None:
            out = torch.empty(n, dtype=torch.float32, device=self.device)
        assert out.numel() == n and out.is_contiguous() and out.dtype == torch.float32 and out.device == self.device
        tok = C.c_int64()
        self._rd_obs = None
        _native.check(self._lib.gx_rollout_tape(self._h, T, a.data_ptr(), out.data_ptr(), C.byref(tok), self._stream()))
        self._obs = None
        return out, tok.value

    def expand_tape(self, shard, token, T, out=None):
        """Observation pass over one shard of rollout_tape(): (T, N, D + A + 3) packed rows (obs | action |
        reward, cost, done).  Runs on the current stream; the engine orders its next layout sampler behind it."""
        N, W = self.env_num, self.obs_flat_size + self.action_space.shape[0] + 3
        assert shard.is_contiguous() and shard.dtype == torch.float32 and shard.device == self.device
        assert shard.numel() == sum(self.tape_floats(T))
        if out is None:
            out = torch.empty(int(T), N, W, dtype=torch.float32, device=self.device)
        assert tuple(out.shape) == (int(T), N, W) and out.is_contiguous()
        _native.check(self._lib.gx_expand_tape(self._h, int(T), shard.data_ptr(), int(token), out.data_ptr(),
                                               self._stream()))
        return out

    def expand_tapes(self, shards, stride_floats, n_shards, token, T, out):
        """expand_tape() over the shards of n_shards ranks in ONE launch: shard s at shards[s * stride_floats:] (the
        all-gathered buffer as it is), its rows into out[s]; out is (n_shards, T, N, D + A + 3)."""
        N, W, T = self.env_num, self.obs_flat_size + self.action_space.shape[0] + 3, int(T)
        assert shards.is_contiguous() and shards.dtype == torch.float32 and shards.device == self.device
        assert shards.numel() >= (int(n_shards) - 1) * int(stride_floats) + sum(self.tape_floats(T))
        assert tuple(out.shape) == (int(n_shards), T, N, W) and out.is_contiguous() and out.device == self.device
        _native.check(self._lib.gx_expand_tapes(self._h, T, shards.data_ptr(), int(stride_floats), int(n_shards),
                                                int(token), out.data_ptr(), T * N * W, self._stream()))
        return out

    # ------------------------------------------------------------------
    # sharded layout sampling (multi-GPU, optional): guardx_amd.dist.ShardedReset drives these two
    # ------------------------------------------------------------------
    @property
    def n_layout_objects(self):
        """objects of a layout row: goal, hazards, pillars, robot (engine.py:533-544)"""
        return int(self.hazards_num) + int(self.pillars_num) + 2

    def shard_capacity(self, n_shards):
        """Rows a shard export holds: an eighth of the shard's candidates (about 2 % of the candidates of the reference's
        tasks are valid layouts), at least 256."""
        return max(256, -(-int(self._cfg.n_candidates) // (8 * int(n_shards))))

    def sample_shard(self, shard, n_shards, rows=None, count=None):
        """Sample candidates [shard M / n, (shard + 1) M / n) of the reset() about to happen (M = n_candidates) and
        export the valid layouts in candidate order: (rows (cap, K, 2) float32, count (1,) int32), on the current
        stream.  The layout prefetch must be off (set_prefetch(-1))."""
        cap, K = self.shard_capacity(n_shards), self.n_layout_objects
        if rows is None:
            rows = torch.empty(cap, K, 2, dtype=torch.float32, device=self.device)
        if count is None:
            count = torch.empty(1, dtype=torch.int32, device=self.device)
        assert tuple(rows.shape) == (cap, K, 2) and rows.is_contiguous() and count.dtype == torch.int32
        _native.check(self._lib.gx_sample_shard(self._h, int(shard), int(n_shards), rows.data_ptr(), cap,
                                                count.data_ptr(), self._stream()))
        return rows, count

    def reset_from_shards(self, rows_all, counts, check=True):
        """reset() with the pool assembled from the all-gathered shard exports (n, cap, K, 2) / (n,): same pool, same
        observation, same later draws as reset()."""
        n, cap = int(rows_all.shape[0]), int(rows_all.shape[1])
        assert rows_all.is_contiguous() and rows_all.dtype == torch.float32 and counts.dtype == torch.int32
        assert counts.numel() == n and cap == self.shard_capacity(n)
        obs = self._new(self.env_num, self.obs_flat_size)
        _native.check(self._lib.gx_reset_from_shards(self._h, rows_all.data_ptr(), counts.data_ptr(), n, cap,
                                                     obs.data_ptr(), self._stream()))
        self._rd_obs = None
        self._obs = obs
        if check:
            n_ = C.c_int32()
            st = self._lib.gx_layout_size(self._h, C.byref(n_))
            self.layout_size = int(n_.value)
            if st == _native.GX_ERR_LAYOUT:
                if self.layout_size < 0:
                    raise ResamplingError(f"shard {-self.layout_size - 1} exported more than {cap} valid layouts")
                raise ResamplingError(f"number of valid layout is {self.layout_size} <= env_num {self._cfg.env_total}")
            _native.check(st)
        return obs

    # ---- the same riding on the tape hand-off (one collective per epoch): guardx_amd.dist.TapeHandoff drives these ----
    def set_layout_source(self, source):
        """'own' (default): reset() samples / prefetches all candidates itself.  'shards': the pool of the next reset()
        is installed from the ranks' export blocks (install_shards); a reset whose key has no installed pool samples
        inline, so results never depend on the source."""
        _native.check(self._lib.gx_set_layout_source(self._h, {'own': 0, 'shards': 1}[source]))

    def shard_block_floats(self, cap):
        n = C.c_int64()
        _native.check(self._lib.gx_shard_block_floats(self._h, int(cap), C.byref(n)))
        return int(n.value)

    def sample_shard_ahead(self, shard, n_shards, block, cap, resets_ahead=2):
        """Sample candidates [shard M / n, (shard + 1) M / n) of the reset() `resets_ahead` resets from now (its key is
        this key advanced by the learned number of steps between resets) on the engine's side stream and export the
        valid layouts into `block` (shard_block_floats(cap) floats: count, key, tag | rows).  Returns the ticket
        install_shards() takes for the blocks of this call (the same number on every rank)."""
        assert block.is_contiguous() and block.dtype == torch.float32 and block.device == self.device
        assert block.numel() == self.shard_block_floats(cap)
        ticket = C.c_int64()
        _native.check(self._lib.gx_sample_shard_ahead(self._h, int(shard), int(n_shards), int(resets_ahead),
                                                      block.data_ptr(), int(cap), C.byref(ticket), self._stream()))
        return int(ticket.value)

    def aux_stream(self, renew=False):
        """The engine's stream for throughput work beside the stepping -- the hand-off's installs and expansions -- as a
        torch stream: one per device and process, never destroyed, ordinary priority (gx_aux_stream).  Asking for it tells
        the engine that it works beside a hand-off: its layout sampler moves to a stream of the same priority class.
        renew=True replaces the device's stream by a new one first (gx_aux_stream_renew)."""
        ptr = C.c_void_p()
        _native.check((self._lib.gx_aux_stream_renew if renew else self._lib.gx_aux_stream)(self._h, C.byref(ptr)))
        if getattr(self, "_aux", None) is None or self._aux.cuda_stream != ptr.value:   # (another engine may have renewed it)
            self._aux = torch.cuda.ExternalStream(ptr.value, device=self.device)
        return self._aux

    def shard_join(self):
        """the current stream waits for the block of the last sample_shard_ahead()"""
        _native.check(self._lib.gx_shard_join(self._h, self._stream()))

    def install_shards(self, ticket, blocks, stride_floats, n_shards, cap):
        """Assemble the n_shards export blocks of sample_shard_ahead() call `ticket` (shard s at
        blocks[s * stride_floats:]) into the pool of the reset() they were sampled for, on the current stream; that
        reset() then takes it like a prefetched pool."""
        assert blocks.is_contiguous() and blocks.dtype == torch.float32 and blocks.device == self.device
        assert blocks.numel() >= (int(n_shards) - 1) * int(stride_floats) + self.shard_block_floats(cap)
        _native.check(self._lib.gx_install_shards(self._h, int(ticket), blocks.data_ptr(), int(stride_floats),
                                                  int(n_shards), int(cap), self._stream()))

    # ------------------------------------------------------------------
    # closed-loop fused rollout (policy evaluated inside the kernel)
    # ------------------------------------------------------------------
    POLICY_HIDDEN = (64, 128, 192, 256)

    @staticmethod
    def pack_actor_critic(ac=None, *, mu_net=None, v_net=None, log_std=None, device=None):
        """Flatten MLPActorCritic(hidden_sizes=(h, h), tanh) weights (trpo_core.py:110-164; h = 64 is the reference
        default, trpo.py:606-607 --hid / --l) into the layout gx_rollout_policy expects.  `ac` needs .pi.mu_net,
        .pi.log_std, .v.v_net (nn.Sequential of Linear/Tanh/Linear/Tanh/Linear[/Identity]); or pass the three pieces."""
        if ac is not None:
            mu_net, v_net, log_std = ac.pi.mu_net, ac.v.v_net, ac.pi.log_std
        parts, widths = [], set()
        for net in (mu_net, v_net):
            lin = [m for m in net if isinstance(m, torch.nn.Linear)]
            if len(lin) != 3 or lin[0].out_features != lin[1].out_features or lin[1].in_features != lin[0].out_features:
                raise NotImplementedError("rollout_policy supports two hidden layers of equal width (--l 2)")
            widths.add(lin[0].out_features)
            for m in lin:
                parts += [m.weight.detach().reshape(-1), m.bias.detach().reshape(-1)]
        if len(widths) != 1 or widths.pop() not in Engine.POLICY_HIDDEN:
            raise NotImplementedError(f"rollout_policy supports hidden_sizes (h, h) with h in {Engine.POLICY_HIDDEN}, "
                                      "the same for actor and critic")
        parts.append(torch.as_tensor(log_std).detach().reshape(-1))
        flat = torch.cat([t.to(torch.float32) for t in parts])
        return flat.to(device) if device is not None else flat

    @staticmethod
    def _policy_floats(D, A, h):
        return 2 * (h * D + h + h * h + h) + (A + 1) * h + (A + 1) + A

    def rollout_policy(self, params, T, obs0=None, noise_seed=(0, 0)):
        """T x (ac.step -> env.step -> reset_done) on device (trpo.py:466-547 with the actor-critic of
        trpo_core.py:110-173 evaluated there).  `params` = pack_actor_critic(ac); the hidden width is read off its
        size.  h = 64: ONE kernel launch for the whole rollout; h = 128 / 192 / 256 (the weights do not fit the fused
        kernel's LDS): two launches per control step, same results.
        Returns a dict of time-major tensors: obs (T,N,D) [what the policy saw], act, mu (T,N,A),
        logp, val, rew, cost, done (T,N), plus obs_last (N,D), val_last (N,), logstd (A,)."""
        if obs0 is None:
            obs0 = self._obs
        if obs0 is None:
            raise RuntimeError("rollout_policy() before reset()")
        N, D, A, T = self.env_num, self.obs_flat_size, self.action_space.shape[0], int(T)
        params = params.to(device=self.device, dtype=torch.float32).contiguous()
        obs0 = obs0.to(device=self.device, dtype=torch.float32).contiguous()
        assert tuple(obs0.shape) == (N, D)
        hidden = next((h for h in self.POLICY_HIDDEN if self._policy_floats(D, A, h) == params.numel()), None)
        if hidden is None:
            raise ValueError(f"params has {params.numel()} floats; expected one of "
                             f"{[self._policy_floats(D, A, h) for h in self.POLICY_HIDDEN]} (hidden {self.POLICY_HIDDEN})")
        self._rd_obs = None
        out = dict(obs=self._new(T, N, D), act=self._new(T, N, A), logp=self._new(T, N), val=self._new(T, N),
                   mu=self._new(T, N, A), rew=self._new(T, N), cost=self._new(T, N), done=self._new(T, N),
                   obs_last=self._new(N, D), val_last=self._new(N), logstd=self._new(A))
        pol = _native.GxPolicy()
        pol.struct_size = C.sizeof(_native.GxPolicy)
        pol.hidden = hidden
        pol.d_params = params.data_ptr()
        pol.seed[0], pol.seed[1] = int(noise_seed[0]) & 0xFFFFFFFF, int(noise_seed[1]) & 0xFFFFFFFF
        _native.check(self._lib.gx_rollout_policy(
            self._h, T, C.byref(pol), obs0.data_ptr(), out['obs'].data_ptr(), out['act'].data_ptr(),
            out['logp'].data_ptr(), out['val'].data_ptr(), out['mu'].data_ptr(), out['rew'].data_ptr(),
            out['cost'].data_ptr(), out['done'].data_ptr(), out['obs_last'].data_ptr(),
            out['val_last'].data_ptr(), out['logstd'].data_ptr(), self._stream()))
        self._obs, self._reward, self._done = out['obs_last'], out['rew'][-1], out['done'][-1]
        self._info = {'cost': out['cost'][-1]}
        return out

    def set_policy_impl(self, impl):
        """rollout_policy's form: 0 auto (one fused launch where there is one: every width on Point / Swimmer at the
        default observation width, width 64 everywhere), 1 VALU fmaf chains, 2 fp32 MFMA tiles (width 64: fused; wider:
        step-wise), 3 the step-wise form at every width (two launches per control step).  Same bits whichever runs."""
        _native.check(self._lib.gx_set_policy_impl(self._h, int(impl)))

    def set_prefetch(self, steps):
        """Predicted number of step() calls between reset()s for the layout-pool prefetch: >= 0 fixed,
        -1 off, -2 (default) = the interval between the last two reset() calls (num_steps before the
        second reset) -- the learners reset every max_ep_len steps.  Never changes results."""
        _native.check(self._lib.gx_set_prefetch(self._h, int(steps)))

    def prefetch_stats(self):
        """(prefetched pools used, prefetched pools discarded, current prediction in steps)"""
        h, m, z = C.c_int32(), C.c_int32(), C.c_int32()
        _native.check(self._lib.gx_prefetch_stats(self._h, C.byref(h), C.byref(m), C.byref(z)))
        return int(h.value), int(m.value), int(z.value)

    def set_path(self, mode):
        """0 auto, 1 thread-per-env kernels, 2 lane-group kernels (bit-identical results)."""
        _native.check(self._lib.gx_set_path(self._h, int(mode)))

    def _as_action(self, action):
        a = action if torch.is_tensor(action) else torch.as_tensor(action)
        if a.shape != self._act_shape:
            raise ValueError(f"action shape {tuple(a.shape)} != {tuple(self._act_shape)}")
        if a.device != self.device or a.dtype != torch.float32:
            a = a.to(device=self.device, dtype=torch.float32)
        if a.requires_grad:
            a = a.detach()
        if not a.is_contiguous():
            a = a.contiguous()
        return a

    # ------------------------------------------------------------------
    # state exchange (tests, checkpoints)
    # ------------------------------------------------------------------
    @property
    def _STATE_FIELDS(self):
        return (('qpos', self.robot.nq), ('qvel', self.robot.nv), ('pose0', 4), ('pose1', 2), ('objs', None),
                ('done0', 1), ('done1', 1), ('steps', 1))

    def get_state(self):
        N, H = self.env_num, int(self.hazards_num) + int(self.pillars_num)
        s = OrderedDict()
        for name, w in self._STATE_FIELDS:
            shape = (N, 1 + H, 2) if name == 'objs' else ((N,) if w == 1 else (N, w))
            s[name] = np.empty(shape, np.float32)
        key = (C.c_uint32 * 2)()
        hist = C.c_int32()
        fp = C.POINTER(C.c_float)
        _native.check(self._lib.gx_get_state(self._h, *[s[n].ctypes.data_as(fp) for n, _ in self._STATE_FIELDS],
                                             key, C.byref(hist)))
        s['key'] = np.array([key[0], key[1]], np.uint32)
        s['hist'] = int(hist.value)
        return s

    def set_state(self, s):
        fp = C.POINTER(C.c_float)
        keep, ptrs = [], []
        for name, _ in self._STATE_FIELDS:
            v = s.get(name)
            if v is None:
                ptrs.append(None)
            else:
                v = np.ascontiguousarray(v, np.float32)
                keep.append(v)
                ptrs.append(v.ctypes.data_as(fp))
        key = None
        if s.get('key') is not None:
            key = (C.c_uint32 * 2)(int(s['key'][0]), int(s['key'][1]))
        hist = C.byref(C.c_int32(int(s['hist']))) if s.get('hist') is not None else None
        _native.check(self._lib.gx_set_state(self._h, *ptrs, key, hist))
        self._rd_obs = None

    def get_pool(self, max_rows=4096):
        H = int(self.hazards_num) + int(self.pillars_num)
        pool = np.empty((max_rows, H + 2, 2), np.float32)
        got = C.c_int32()
        _native.check(self._lib.gx_get_pool(self._h, pool.ctypes.data_as(C.POINTER(C.c_float)),
                                            max_rows, C.byref(got)))
        return pool[:got.value]

    # ------------------------------------------------------------------
    def render(self):
        raise NotImplementedError("render() needs the MuJoCo viewer (engine.py:1036-1070); "
                                  "out of scope for the batched step path")

    def close(self):
        if getattr(self, '_h', None):
            self._lib.gx_destroy(self._h)       # (synchronises the device)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def __reduce__(self):
        # EzPickle(config=config) in the reference (engine.py:214): the pickle is the config
        return (_rebuild, (self._ctor_config, self._ctor_kwargs))


def _rebuild(config, kwargs):
    return Engine(config, **kwargs)
