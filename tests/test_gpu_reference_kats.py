"""The reference's own unit tests (its `unittest.TestCase`s at the bottom of
dynamic_models.py, collision_models.py and laser_models.py), restated against the HIP
path through the C ABI."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope='module')
def car_eng():
    """DynamicsTest.setUp (dynamic_models.py:232-253): CommonRoad vehicle parameters."""
    from red_gym_amd.engine import Engine
    p = {'mu': 1.0489, 'C_Sf': 21.92 / 1.0489, 'C_Sr': 21.92 / 1.0489, 'lf': 0.3048 * 3.793293,
         'lr': 0.3048 * 4.667707, 'h': 0.3048 * 2.01355, 'm': 4.4482216152605 / 0.3048 * 74.91452,
         'I': 4.4482216152605 * 0.3048 * 1321.416, 's_min': -1.066, 's_max': 1.066, 'sv_min': -0.4, 'sv_max': 0.4,
         'v_switch': 7.319, 'a_max': 11.5, 'v_min': -13.6, 'v_max': 50.8, 'width': 0.31, 'length': 0.58}
    e = Engine(num_envs=1, num_agents=1, params=p, noise_std=0)
    yield e
    e.close()


def test_derivatives(car_eng):
    """dynamic_models.py:255-279"""
    f_ks_gt = [16.3475935934250209, 0.4819314886013121, 0.1500000000000000, 5.1464424102339752, 0.2401426578627629]
    f_st_gt = [15.7213512030862397, 0.0925527979719355, 0.1500000000000000, 5.3536773276413925, 0.0529001056654038,
               0.6435589397748606, 0.0313297971641291]
    g = 9.81
    x_ks = np.array([3.9579422297936526, 0.0391650102771405, 0.0378491427211811, 16.3546957860883566, 0.0294717351052816, 0, 0])
    x_st = np.array([2.0233348142065677, 0.0041907137716636, 0.0197545248559617, 15.7216236334290116, 0.0025857914776859,
                     0.0529001056654038, 0.0033012170610298])
    u = np.array([0.15, 0.63 * g])
    f_ks = _np(car_eng.vehicle_dynamics(x_ks, u, kinematic=True))[0, :5]
    f_st = _np(car_eng.vehicle_dynamics(x_st, u))[0]
    assert round(float(np.max(np.abs(f_ks_gt - f_ks))), 7) == 0.   # assertAlmostEqual(.., 0.)
    assert round(float(np.max(np.abs(f_st_gt - f_st))), 7) == 0.


@pytest.mark.parametrize('u,st_gt,ks_gt', [
    ([0., 0.], [0.] * 7, [0.] * 5),                                                         # test_zeroinit_roll  :281-311
    ([0., -0.7 * 9.81], [-3.4335000000000013, 0, 0, -6.8670000000000018, 0, 0, 0],          # test_zeroinit_dec   :313-348
     [-3.4335000000000013, 0, 0, -6.8670000000000018, 0]),
    ([0.15, 0.63 * 9.81], [3.0731976046859715, 0.2869835398304389, 0.15, 6.1802999999999999,  # test_zeroinit_acc   :350-386
                           0.1097747074946325, 0.3248268063223301, 0.0697547542798040],
     [3.0845676868494927, 0.1484249221523042, 0.15, 6.1803000000000017, 0.1203664469224163]),
    ([0.15, 0.], [0, 0, 0.15, 0, 0, 0, 0], [0, 0, 0.15, 0, 0]),                               # test_zeroinit_rollleft :388-423
])
def test_zeroinit_odeint(car_eng, u, st_gt, ks_gt):
    """scipy.odeint over 1 s with the GPU right-hand side, tolerance 1e-2 as upstream."""
    from scipy.integrate import odeint
    u = np.array(u)
    t = np.arange(0., 1., 1e-4)

    def f_st(x, _t):
        return _np(car_eng.vehicle_dynamics(x, u))[0]

    def f_ks(x, _t):
        return _np(car_eng.vehicle_dynamics(np.concatenate([x, [0, 0]]), u, kinematic=True))[0, :5]
    x_st = odeint(f_st, np.zeros(7), t)
    x_ks = odeint(f_ks, np.zeros(5), t)
    assert np.all(np.abs(x_st[-1] - st_gt) < 1e-2)
    assert np.all(np.abs(x_ks[-1] - ks_gt) < 1e-2)


def test_random_collision_and_multiple(car_eng):
    """collision_models.py:306-324 with np.random.seed(1234)."""
    np.random.seed(1234)
    v1 = np.asarray([[4, 11.], [5, 5], [9, 9], [10, 10]])
    a = np.stack([v1 + np.random.normal(size=v1.shape) / 100. for _ in range(1000)])
    b = np.stack([v1 + np.random.normal(size=v1.shape) / 100. for _ in range(1000)])
    assert bool(_np(car_eng.gjk_pairs(a, b)).all())
    np.random.seed(1234)
    allv = np.stack([v1 + np.random.normal(size=v1.shape) / 100. for _ in range(6)] + [v1 + 10.])
    col, idx = car_eng.collision_multiple(allv[None])
    assert np.all(_np(col)[0] == np.array([1., 1., 1., 1., 1., 1., 0.]))
    assert np.all(_np(idx)[0] == np.array([5., 5., 5., 5., 5., 4., -1.]))


def test_scan_fps_and_legacy_mse(assets, golden):
    """laser_models.py:534-552 (fps > 500 on berlin, 10 000 scans with noise) and
    unittest/scan_sim.py:321-366 (MSE < 2 against the legacy C++ simulator's scans)."""
    import time
    from f110_gym.envs.laser_models import ScanSimulator2D
    legacy = golden('legacy_scan.npz')
    for name in ('berlin', 'skirk'):
        sim = ScanSimulator2D(1080, 4.7)
        sim.set_map(os.path.join(assets, 'maps', name + '.yaml'), '.png')
        new = np.stack([sim.scan(np.array([0., 0., th]), None) for th in np.linspace(-1., 1., num=10)])
        assert np.mean((new - legacy[name]) ** 2) < 2.
    rng = np.random.default_rng(seed=12345)
    start = time.time()
    for i in range(2000):
        sim.scan(np.array([0., 0., rng.random() * np.pi / 2.]), rng)   # one-pose calls like upstream
    assert 2000 / (time.time() - start) > 500
    poses = np.zeros((10000, 3))
    poses[:, 2] = rng.random(10000) * np.pi / 2.
    start = time.time()
    sim.scan_batch(poses).cpu()
    assert 10000 / (time.time() - start) > 100000   # batched: the same 10 000 scans in one call
