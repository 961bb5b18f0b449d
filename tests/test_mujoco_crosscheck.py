"""Pins for the [derived] physics, for a machine that has MuJoCo (this container and the GPU box do not: the module
is skipped there).  It needs `mujoco` and the robot MJCF files of the reference checkout.

What it checks, per ADVICE r1 / VERDICT r1 weak #1:
  * point.xml compiles to three actuators with ctrllimited, ctrlrange +-1, forcelimited, forcerange +-.05, fixed gain
    1 and the affine bias (0, 0, -1) -- the reading DESIGN.md section 0.1 argues for, and the one the kernels carry;
  * a few Euler steps of the robot-only model under constant and random ctrl match the checker's Point step.
"""
import os

import numpy as np
import pytest

mujoco = pytest.importorskip("mujoco")
XML_DIR = "/root/reference/safe_rl_envs/safe_rl_envs/xmls"
pytestmark = pytest.mark.skipif(not os.path.isdir(XML_DIR), reason="robot MJCF files (reference checkout) not present")


def test_point_actuators_as_mujoco_compiles_them():
    m = mujoco.MjModel.from_xml_path(os.path.join(XML_DIR, "point.xml"))
    assert m.nu == 3
    np.testing.assert_array_equal(m.actuator_ctrllimited, [1, 1, 1])
    np.testing.assert_allclose(m.actuator_ctrlrange, [[-1, 1]] * 3)
    np.testing.assert_array_equal(m.actuator_forcelimited, [1, 1, 1])
    np.testing.assert_allclose(m.actuator_forcerange, [[-0.05, 0.05]] * 3)
    np.testing.assert_allclose(m.actuator_gainprm[:, 0], 1.0)
    np.testing.assert_array_equal(m.actuator_biastype, [int(mujoco.mjtBias.mjBIAS_AFFINE)] * 3)
    np.testing.assert_allclose(m.actuator_biasprm[:, :3], [[0, 0, -1]] * 3)
    np.testing.assert_allclose(m.actuator_gear[:, 0], 0.3)


def test_point_steps_match_the_checker(oracle):
    from helpers import task_config
    m = mujoco.MjModel.from_xml_path(os.path.join(XML_DIR, "point.xml"))
    d = mujoco.MjData(m)
    N = 1
    E = oracle.OracleEngine(task_config(N), n_candidates=4000)
    E.reset(check=False)
    s = E.get_state()
    s['qpos'][:] = 0; s['qvel'][:] = 0; s['pose0'][:] = [0, 0, 1, 0]; s['objs'][:] = 50.0; s['hist'] = 2
    E.set_state(s)
    rng = np.random.default_rng(0)
    mujoco.mj_forward(m, d)                 # xmat of the zero pose (engine.py:229-232)
    for t in range(60):
        a = rng.uniform(-1.5, 1.5, 2).astype(np.float32)
        # convert_action with the PRE-step heading (engine.py:672-685): the xmat the previous mj_step left behind,
        # i.e. the kinematics of the qpos before that step's integration (one step stale)
        R = d.xmat[1].reshape(3, 3)
        d.ctrl[:] = [a[0] * R[0, 0], a[0] * R[1, 0], a[1]]
        mujoco.mj_step(m, d)
        obs, *_ = E.step(a[None])
        np.testing.assert_allclose(obs[0, 37:40], d.qpos[:3], rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(obs[0, 40:43], d.qvel[:3], rtol=2e-4, atol=2e-5)
