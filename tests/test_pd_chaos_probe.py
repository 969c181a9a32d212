"""Why the free-running PD parity bar is a shadowing bound and not 1e-5 outright (VERDICT r4, item 6c asked for a stabilising
PD sequence that stays non-chaotic for 1000 substeps): there is none in this model.  Cassie2d::StepPd (Cassie2d.cpp:96-117) puts
u = 10 (target - q) - 5 qd through gear 50 onto the toe joint, an explicit damper with h b / I ~ 20 >> 2 at h = 0.5 ms: the toe
chatters between its control limits even when the targets ARE the pose the robot stands in.  Measured here on the oracle alone
(no GPU): a twin started 1 ulp away ends O(1) away after 1000 substeps of "hold the reset pose", and even with both toe commands
pinned at their limit by far-away targets (the most benign sequence found) 1 ulp grows by > 1e9 -- so two correct implementations
whose roundings differ by ~1e-13 per substep cannot be asked to agree to 1e-5 there.  Torque mode has no such loop (10 000
free-running substeps agree to 1e-11: test_gpu_parity.py)."""
import numpy as np

QINIT = np.array([0, 0.939, 0, 0.68111815, -1.40730357, 1.62972042, -1.77611107, -0.61968407,
                  0.68111815, -1.40730357, 1.62972042, -1.77611107, -0.61968407])
TOE_HI = np.radians(-30.0)


def _twin_divergence(oracle_mod, target, substeps=1000):
    a, b = oracle_mod.Oracle(), oracle_mod.Oracle()
    a.reset(QINIT, np.zeros(13)); b.reset(QINIT, np.zeros(13))
    q, v = b.state()
    b.set_state_raw(np.nextafter(q, q + 1.0), v, b.warmstart())
    worst = 0.0
    for _ in range(substeps):
        a.step_pd(target); b.step_pd(target)
        qa, va = a.state(); qb, vb = b.state()
        worst = max(worst, np.abs(qa - qb).max(), np.abs(va - vb).max() / (1.0 + np.abs(va).max()))
    return worst


def test_holding_the_reset_pose_is_chaotic_in_pd_mode(oracle_mod):
    hold = QINIT[[3, 4, 6, 8, 9, 11]].copy()
    assert _twin_divergence(oracle_mod, hold) > 1e-2            # measured 1.6: the toe joints chatter, the 1-ulp twin is lost by substep ~300


def test_even_saturated_toe_commands_amplify_one_ulp_past_the_bar(oracle_mod):
    sat = QINIT[[3, 4, 6, 8, 9, 11]].copy()
    sat[2] = sat[5] = TOE_HI                                     # 10 (target - q) >> ctrlrange: both toe commands sit on their limit
    d = _twin_divergence(oracle_mod, sat)
    assert 1e-7 < d < 1e-2, d                                    # measured 5e-5 from a 2e-16 perturbation: amplification > 1e9
