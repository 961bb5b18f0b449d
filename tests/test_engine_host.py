"""Host-side logic of the Engine mirror that needs no GPU."""
import ast
import os

import numpy as np
import pytest

from guardx_amd import Engine, configuration
from guardx_amd.spaces import Box

REF_ENGINE = "/root/reference/safe_rl_envs/safe_rl_envs/envs/engine.py"


def test_default_keys_match_reference_interface():
    """The constructor contract is the DEFAULT key set (engine.py:98-204, 322-328)."""
    assert len(Engine.DEFAULT) == 74
    if not os.path.exists(REF_ENGINE):
        pytest.skip("reference not present on this machine")
    tree = ast.parse(open(REF_ENGINE).read())
    ref = None
    for node in ast.walk(tree):
        if isinstance(node, ast.ClassDef) and node.name == "Engine":
            for b in node.body:
                if isinstance(b, ast.Assign) and getattr(b.targets[0], "id", "") == "DEFAULT":
                    ref = ast.literal_eval(b.value)
    assert ref is not None
    assert list(ref.keys()) == list(Engine.DEFAULT.keys())
    assert ref == Engine.DEFAULT


def test_bad_key_asserts_like_the_reference():
    with pytest.raises(AssertionError, match="Bad key observe_box_comp"):
        Engine({'observe_box_comp': True})


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Engine({'env_num': 4})


def test_unsupported_tasks_fail_loudly():
    with pytest.raises(KeyError):
        Engine({'task': 'push'})
    with pytest.raises(TypeError):
        Engine({'hazards_num': 0})
    with pytest.raises(NotImplementedError):
        Engine({'robot_base': 'xmls/doggo.xml'})


def test_task_configs():
    c = configuration("Goal_Point_8Hazards")
    assert c['robot_base'] == 'xmls/point.xml' and c['hazards_num'] == 8 and c['lidar_num_bins'] == 16
    assert c['constrain_indicator'] is False and c['goal_size'] == 0.5 and c['hazards_size'] == 0.3
    assert set(c) <= set(Engine.DEFAULT)
    assert configuration("Goal_Swimmer_8Hazards")['robot_base'] == 'xmls/swimmer.xml'
    assert configuration("no_such_task") == {}


def test_pillars_are_an_extension_not_a_reference_key():
    """BASELINE config 5 objects: accepted by this Engine, absent from the reference's DEFAULT"""
    assert not (set(Engine.EXTENSIONS) & set(Engine.DEFAULT))
    c = configuration("Ant_8Hazards_8Pillars_synthetic")
    assert c['robot_base'] == 'xmls/ant.xml' and c['pillars_num'] == 8 and c['hazards_num'] == 8
    assert set(c) <= set(Engine.DEFAULT) | set(Engine.EXTENSIONS)
    with pytest.raises(AssertionError, match="Bad key pillars_cost"):
        Engine({'pillars_cost': 1.0})


def test_box_stand_in():
    b = Box(-np.inf, np.inf, (43,), dtype=np.float32)
    assert b.shape == (43,) and b.dtype == np.float32
    a = Box(np.full(2, -np.inf, np.float32), np.full(2, np.inf, np.float32), dtype=np.float32)
    assert a.shape == (2,) and np.isinf(a.low).all()


def _host_only(config, nq=3, nv=3, nu=3):
    """the host-side configuration logic without a device: parse + placements + observation table"""
    e = object.__new__(Engine)
    e.parse(config)
    e.robot = type('Robot', (), dict(nq=nq, nv=nv, nu=nu))()
    e.build_placements_dict()
    e.build_observation_space()
    return e


def test_placements_table_follows_the_reference_rules():
    """engine.py:507-544: order goal, hazard0.., robot; `*_locations[i]` pins object i to the degenerate rectangle
    of half-width keepout + 1e-9 around it, the others get the family's `*_placements` (None = the arena)."""
    e = _host_only({'hazards_num': 3, 'hazards_locations': [(1.0, -0.5)], 'hazards_keepout': 0.25,
                    'goal_placements': [(-1, -1, 0, 0)], 'robot_locations': [(0.5, 0.5), (9, 9)]})
    assert list(e.placements) == ['goal', 'hazard0', 'hazard1', 'hazard2', 'robot']
    assert e.placements['goal'] == ([(-1, -1, 0, 0)], 0.5)
    k = 0.25 + 1e-9
    assert e.placements['hazard0'] == ([(1.0 - k, -0.5 - k, 1.0 + k, -0.5 + k)], 0.25)
    assert e.placements['hazard1'] == (None, 0.25) and e.placements['hazard2'] == (None, 0.25)
    k = 0.4 + 1e-9
    assert e.placements['robot'] == ([(0.5 - k, 0.5 - k, 0.5 + k, 0.5 + k)], 0.4)    # only the first location is used
    p = _host_only(dict(configuration("Ant_8Hazards_8Pillars_synthetic")), 11, 11, 8).placements
    assert list(p) == ['goal'] + [f'hazard{i}' for i in range(8)] + [f'pillar{i}' for i in range(8)] + ['robot']
    assert p['pillar3'] == (None, 0.3)


def test_observation_table_orders():
    """engine.py:386-409: dict in insertion order; flat observation in sorted-key order (engine.py:773-777)"""
    e = _host_only(dict(configuration("Goal_Point_8Hazards")))
    assert list(e.obs_space_dict) == ['goal_lidar', 'goal_compass', 'hazards_lidar', 'qpos', 'qvel', 'ctrl']
    assert e.obs_flat_size == 43 and e.observation_space.shape == (43,)
    assert {k: (s.start, s.stop) for k, s in e._obs_slices.items()} == {
        'ctrl': (0, 3), 'goal_compass': (3, 5), 'goal_lidar': (5, 21), 'hazards_lidar': (21, 37), 'qpos': (37, 40),
        'qvel': (40, 43)}
    assert float(e.obs_space_dict['goal_lidar'].low[0]) == 0.0 and float(e.obs_space_dict['goal_lidar'].high[0]) == 1.0
    assert np.isinf(e.obs_space_dict['qpos'].low).all()
    e = _host_only({'observe_vel': True, 'observe_acc': True, 'observe_hazards': False, 'lidar_num_bins': 10,
                    'observe_ctrl': False}, 5, 5, 2)
    assert list(e.obs_space_dict) == ['goal_lidar', 'goal_compass', 'qpos', 'qvel', 'vel', 'acc']
    assert list(e._obs_slices) == ['acc', 'goal_compass', 'goal_lidar', 'qpos', 'qvel', 'vel']
    assert e.obs_flat_size == 2 + 2 + 10 + 5 + 5 + 2
    e = _host_only(dict(configuration("Ant_8Hazards_8Pillars_synthetic")), 11, 11, 8)
    assert list(e._obs_slices) == ['ctrl', 'goal_compass', 'goal_lidar', 'hazards_lidar', 'pillars_lidar', 'qpos', 'qvel']
    assert e.obs_flat_size == 80


def test_step_info_behaves_like_the_plain_dict_of_the_reference():
    """info = {'cost': ..., 'obs': {...}} (engine.py:693-695): 'obs' is built lazily, but every dict entry point sees it."""
    import torch
    from guardx_amd.engine import _StepInfo

    def make():
        info = _StepInfo(cost=torch.ones(2))
        info._src = (torch.arange(8.).reshape(2, 4), {'qpos': slice(0, 3), 'ctrl': slice(3, 4)}, None)
        return info
    assert set(make().copy()) == {'cost', 'obs'} and type(make().copy()) is dict
    assert set(make().pop('obs')) == {'qpos', 'ctrl'}
    assert make().setdefault('obs', None) is not None
    assert set(make() | {'x': 1}) == {'cost', 'obs', 'x'} and set({'x': 1} | make()) == {'x', 'cost', 'obs'}
    i = make(); i |= {'y': 2}
    assert set(i) == {'cost', 'obs', 'y'}
    assert make().popitem()[0] == 'obs' and list(reversed(make())) == ['obs', 'cost']
    assert len(make()) == 2 and 'obs' in make() and make()['obs']['qpos'].shape == (2, 3)
    with pytest.raises(KeyError):
        make().pop('nope')
    bare = _StepInfo(cost=1)                  # made without a source: an ordinary dict
    assert bare.copy() == {'cost': 1} and 'obs' not in bare and len(bare) == 1
