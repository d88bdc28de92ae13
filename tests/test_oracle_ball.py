"""A10 ball scene: known answers for the ballistics and spawn logic of the oracle, env-level behaviour.  CPU only."""
import numpy as np
import pytest

from oracle import oracle as O
from safemotionsrisk_amd.scene import BALL_ENV_CONFIG, build_scene, ball_target_height_time

G = 9.81
UDT = 0.1 / 24


@pytest.fixture(scope="module")
def ball_scene(built):
    return build_scene(BALL_ENV_CONFIG, auto_reset=False)


def test_ball_scene_constants(ball_scene):
    s = ball_scene.struct
    assert ball_scene.obs_size == 27 and s.use_ball == 1 and s.n_planets == 0
    assert s.moving_link_mask == 0b11111100 and s.contact_link_mask == 0xFF      # iiwa_link_2..7 observed (scene.py:226-228)
    assert s.ball_radius == pytest.approx(0.038) and s.ball_speed == 6.0
    # observation ranges, ctlp.py:309-348: apex of a vertical 6 m/s throw from z = 1.0, impact speed on the way down
    assert s.ball_obs_pos_max[2] == pytest.approx(1.0 + 36 / (2 * G))
    t0 = (-6 + np.sqrt(36 + 2 * G * (1.0 - 0.038))) / G
    assert ball_target_height_time(1.0, -6.0, 0.038) == pytest.approx(t0)
    assert s.ball_obs_vel_max[2] == pytest.approx(6 + G * t0) and s.ball_obs_vel_min[2] == pytest.approx(-(6 + G * t0))
    assert s.ball_target_min_dist_static == pytest.approx(0.1)
    pc = ball_scene.pieces[s.ball_piece]
    v = ball_scene.verts[pc.vert_begin:pc.vert_begin + pc.vert_count]
    r = np.linalg.norm(v, axis=1)
    assert pc.vert_count == 64 and 0.034 < r.min() and r.max() < 0.042          # basketball hull scaled to 38 mm


def test_spawned_balls_hit_their_target_and_counters(ball_scene):
    s = ball_scene.struct
    env = O.OracleEnv(ball_scene, 256, seed=3, num_threads=4)
    env.reset()
    st = env.get_state()
    p0, v0, e0, om = st["ball"][:, :3], st["ball"][:, 3:6], st["ball"][:, 6:9], st["ball"][:, 9]
    k, kmax, khit, active, spawns = st["ball_i"].T
    assert (active == 1).all() and (spawns == 1).all()
    # release point on the sphere segment (ctlp.py:1943-1952), launch speed 6 m/s, always the high launch angle
    rel = p0 - np.array(s.ball_sphere_center)
    assert np.allclose(np.linalg.norm(rel, axis=1), 2.5) and (np.abs(rel[:, 2]) <= 0.5).all()
    assert np.allclose(np.linalg.norm(v0, axis=1), 6.0) and (v0[:, 2] > 6 * np.sin(np.pi / 4) - 1e-9).all()
    assert ((0 <= om) & (om <= 2 * np.pi)).all()
    # the trajectory passes through a point inside the target box (the aimed-at end-effector position)
    box = np.array([[s.ball_target_box[c][0], s.ball_target_box[c][1]] for c in range(3)])
    for e in range(0, 256, 16):
        ts = np.linspace(0, 2.0, 4001)[:, None]
        traj = p0[e] + v0[e] * ts + 0.5 * np.array([0, 0, -G]) * ts ** 2
        inside = ((traj >= box[:, 0] - 1e-6) & (traj <= box[:, 1] + 1e-6)).all(axis=1)
        assert inside.any()
    # counters: k_hit = round(t_land / udt) - 1 with t_land on the table (z = r) or the ground (z = -0.94 + r)
    for e in range(0, 256, 8):
        t_table = ball_target_height_time(p0[e, 2], v0[e, 2], 0.038)
        xy = p0[e, :2] + v0[e, :2] * t_table
        on_table = abs(xy[0]) <= 0.6 and abs(xy[1]) <= 0.8
        t_land = t_table if on_table else ball_target_height_time(p0[e, 2], v0[e, 2], -0.94 + 0.038)
        assert khit[e] == round(t_land / UDT) - 1
        assert 0 < kmax[e] <= int(np.floor(ball_target_height_time(p0[e, 2], v0[e, 2], -0.3) / UDT))
    # random initial position: a multiple of 24 updates, before the ball is retired (ctlp.py:1090-1111)
    assert (k % 24 == 0).all() and (k <= np.minimum(kmax, khit)).all() and len(np.unique(k)) > 3
    # invalid target zone (ctlp.py:1813-1848): the ball does not come down through x in [-0.4, 0], |y| <= 0.2 at z = 0.5 / 0
    for e in range(0, 256, 4):
        for z in (0.5 + 0.038, 0.0 + 0.038):
            t = ball_target_height_time(p0[e, 2], v0[e, 2], z)
            if not np.isnan(t):
                x, y = p0[e, :2] + v0[e, :2] * t
                assert not (-0.4 <= x <= 0.0 and -0.2 <= y <= 0.2)


def test_ball_observation_and_flight(ball_scene):
    s = ball_scene.struct
    env = O.OracleEnv(ball_scene, 1, seed=0)
    ball = np.array([2.0, 0.0, 1.0, -4.0, 0.0, 3.0, 0.0, 0.0, 0.0, 1.0])
    env.set_state(q=np.zeros(7), v=0, a=0, step_count=0, moving_hit=0, ball=ball, ball_i=[48, 300, 280, 1, 1])
    obs = env.observe()[0]
    t = 48 * UDT
    pos = ball[:3] + ball[3:6] * t + np.array([0, 0, -0.5 * G * t * t])
    vel = ball[3:6] + np.array([0, 0, -G * t])
    lo, hi = np.array(s.ball_obs_pos_min), np.array(s.ball_obs_pos_max)
    assert np.allclose(obs[21:24], -1 + 2 * (pos - lo) / (hi - lo), atol=1e-6)          # ctlp.py:2368-2371
    lo, hi = np.array(s.ball_obs_vel_min), np.array(s.ball_obs_vel_max)
    assert np.allclose(obs[24:27], -1 + 2 * (vel - lo) / (hi - lo), atol=1e-6)          # ctlp.py:2374-2377
    # one step: 24 more updates, still far from the arm -> distance capped at the cut-off, full reward
    _, rew, done, reason, info = env.step(np.zeros((1, 7)))
    st = env.get_state()
    assert st["ball_i"][0].tolist() == [72, 300, 280, 1, 1]
    assert info[0, 2] == pytest.approx(0.602) and not done[0]


def test_ball_retirement_and_respawn(ball_scene):
    env = O.OracleEnv(ball_scene, 3, seed=0)
    ball = np.array([2.0, 0.0, 1.0, -4.0, 0.0, 3.0, 0.0, 0.0, 0.0, 1.0])
    # env 0: passes k_max inside the step (missed), env 1: reaches k_hit (table/ground), env 2: keeps flying
    env.set_state(q=np.zeros(7), v=0, a=0, step_count=0, moving_hit=0, ball=ball,
                  ball_i=[[290, 300, 400, 1, 7], [290, 400, 300, 1, 7], [200, 400, 400, 1, 7]])
    _, rew, done, reason, info = env.step(np.zeros((3, 7)))
    st = env.get_state()
    assert np.allclose(info[:, 2], 0.602)                         # retired balls are not measured, ctlp.py:3220-3228
    assert st["ball_i"][:2, 4].tolist() == [8, 8] and (st["ball_i"][:2, 0] == 0).all() and (st["ball_i"][:2, 3] == 1).all()
    assert st["ball_i"][2].tolist() == [224, 400, 400, 1, 7]
    assert not np.allclose(st["ball"][0], ball) and np.allclose(st["ball"][2], ball)


def test_ball_hit_terminates_with_zero_distance(ball_scene):
    """A ball placed inside the wrist: contact at the first sample -> hit_robot, distance 0, reason 5, new ball."""
    env = O.OracleEnv(ball_scene, 1, seed=0)
    pos, rot = env.fk(np.zeros(7))
    ball = np.concatenate([pos[6] + [0.0, 0.0, 0.0], [0.0, 0.0, 0.0], [0, 0, 0], [0.0]])    # at rest at link 6, falls 5 cm in 0.1 s
    env.set_state(q=np.zeros(7), v=0, a=0, step_count=0, moving_hit=0, ball=ball, ball_i=[0, 1000, 1000, 1, 1])
    _, rew, done, reason, info = env.step(np.zeros((1, 7)))
    assert info[0, 2] == 0.0 and done[0] and reason[0] == 5
    assert rew[0] == pytest.approx(0.4 + 1 + 1 + 0 - 15)
    assert env.get_state()["ball_i"][0, 4] == 2                   # a new ball was spawned for the next observation


def test_ball_distance_matches_sphere_bound(ball_scene):
    """The ball hull lies between spheres of radius 35.5 and 41.2 mm: hull distance is bracketed by sphere distances."""
    s = ball_scene.struct
    env = O.OracleEnv(ball_scene, 1, seed=0)
    q = np.array([0.3, 0.8, -0.2, -1.1, 0.4, 0.9, 0.0])
    pos, rot = env.fk(q)
    rng = np.random.default_rng(0)
    checked = 0
    for _ in range(40):
        c = pos[7] + rot[7] @ np.array([0, 0, 0.1]) + rng.normal(size=3) * 0.15
        ball = np.concatenate([c, np.zeros(3), rng.uniform(-1, 1, 3), [0.0]])
        d = env.ball_distances(q, ball, 0)[2]
        # distance of the ball centre to the links (point "hull"), via a tiny ball: exact bracket
        if 0.01 < d < 0.55:
            # moving the centre by delta changes any hull distance by at most |delta|
            ball2 = ball.copy(); ball2[:3] += [0.01, 0, 0]
            assert abs(env.ball_distances(q, ball2, 0)[2] - d) <= 0.01 + 1e-9
            checked += 1
    assert checked > 5
