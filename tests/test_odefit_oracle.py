"""CPU checks of the ODE-fit oracle (oracle/odefit_ref.py): the extension has no reference counterpart, so the oracle itself is
validated by self-consistency -- RK4 is 4th order, planted parameters are recovered -- before the GPU tests lean on it."""
import numpy as np

from oracle import odefit_ref as R

CAM = np.array([0.0, 1.0, 0.0, 0.0, 0.28, 0.0, -0.96, 0.5, 0.96, 0.0, 0.28, 6.5, 2200.0, 0.0, 959.5, 0.0, 2200.0, 539.5, 0.0, 0.0, 1.0])
P0 = np.array([-1.2, 0.2, 1.1, 4.0, -0.3, 2.0, 60.0, -120.0, 40.0])


def test_rk4_is_fourth_order():
    times = np.arange(0, 0.5, 0.05)
    exact = R.integrate(P0, times, 1e-4)
    errs = [np.abs(R.integrate(P0, times, h) - exact).max() for h in (0.05, 0.025, 0.0125)]
    assert 12.0 < errs[0] / errs[1] < 20.0 and 12.0 < errs[1] / errs[2] < 20.0, errs


def test_magnus_and_drag_signs():
    # topspin (spin about +y for motion along +x) pushes the ball down, drag slows it: against the vacuum parabola
    times = np.array([0.0, 0.3])
    vac = np.array([P0[0] + P0[3] * 0.3, P0[1] + P0[4] * 0.3, P0[2] + P0[5] * 0.3 - 0.5 * R.GRAV * 0.09])
    none = R.integrate(np.concatenate([P0[:6], [0, 0, 0]]), times, 1e-3)[1]
    top = R.integrate(np.concatenate([P0[:6], [0, 200.0, 0]]), times, 1e-3)[1]
    assert none[0] < vac[0] and abs(none[2] - vac[2]) < 0.05
    assert top[2] < none[2] - 0.01


def test_oracle_fit_recovers_planted_parameters():
    times = np.arange(40) / 60.0
    obs = R.project(CAM, R.integrate(P0, times, 2e-3))
    init = P0 + np.array([0.05, -0.04, 0.03, 0.4, -0.3, 0.2, 15.0, -10.0, 12.0])
    got = R.fit(obs, times, CAM, init, 2e-3)
    assert np.abs(got[:6] - P0[:6]).max() < 1e-6 and np.abs(got[6:] - P0[6:]).max() < 1e-4
