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
