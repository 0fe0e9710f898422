"""Shared parity metric and small utilities for the tests."""
import numpy as np

POS_TOL = 1e-4   # per vertex: |Pg - Pr|_2 <= POS_TOL * max(|Pr|_2, 1)   (north_star: 1e-4 relative fp32)
NRM_TOL = 1e-4   # per vertex: |Ng - Nr|_2 <= NRM_TOL                      (unit vectors)


def parity_errors(pos_g, nrm_g, pos_r, nrm_r):
    pos_g = np.asarray(pos_g, dtype=np.float64)
    pos_r = np.asarray(pos_r, dtype=np.float64)
    nrm_g = np.asarray(nrm_g, dtype=np.float64)
    nrm_r = np.asarray(nrm_r, dtype=np.float64)
    ep = np.linalg.norm(pos_g - pos_r, axis=1) / np.maximum(np.linalg.norm(pos_r, axis=1), 1.0)
    en = np.linalg.norm(nrm_g - nrm_r, axis=1)
    return ep, en


def assert_parity(pos_g, nrm_g, pos_r, nrm_r, what=""):
    assert np.isfinite(pos_g).all() and np.isfinite(nrm_g).all(), "NaN/Inf in GPU output " + what
    ep, en = parity_errors(pos_g, nrm_g, pos_r, nrm_r)
    assert ep.max() <= POS_TOL, "%s position error max %.3e (p99.9 %.3e)" % (what, ep.max(), np.percentile(ep, 99.9))
    assert en.max() <= NRM_TOL, "%s normal error max %.3e (p99.9 %.3e)" % (what, en.max(), np.percentile(en, 99.9))
    return float(ep.max()), float(en.max())
