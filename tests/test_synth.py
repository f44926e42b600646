import numpy as np

from graphite_amd import synth


def test_generator_is_deterministic_and_well_formed():
    a = synth.make_config("mini-50")
    b = synth.make_config("mini-50")
    assert np.array_equal(a.obs, b.obs) and np.array_equal(a.cam_idx, b.cam_idx)
    Nc, Np, No = a.shape
    assert (Nc, Np, No) == synth.CONFIGS["mini-50"][:3]
    deg = np.bincount(a.pt_idx, minlength=Np)
    assert deg.min() >= 2 and deg.sum() == No
    key = a.pt_idx.astype(np.int64) * Nc + a.cam_idx
    assert len(np.unique(key)) == No                     # no duplicate (camera, point) edge
    assert np.bincount(a.cam_idx, minlength=Nc).min() > 0
    # observations = projection of the truth + 0.5 px noise: the initial guess is close
    r = synth.project(a.cameras, a.points, a.cam_idx, a.pt_idx) - a.obs
    assert 0.5 < np.sqrt((r ** 2).mean()) < 50


def test_ladybug_shape_statistics():
    Nc, Np, No, seed, window = synth.CONFIGS["ladybug-49"]
    p = synth.make_config("ladybug-49")
    assert p.shape == (49, 7776, 31843)
    # banded: every point's cameras fit a window of 32
    order = np.lexsort((p.cam_idx, p.pt_idx))
    ci, pi = p.cam_idx[order].astype(int), p.pt_idx[order]
    first = np.r_[True, pi[1:] != pi[:-1]]
    last = np.r_[pi[1:] != pi[:-1], True]
    span = (ci[last] - ci[first]) % Nc
    assert span.max() < Nc


def test_bal_text_roundtrip(tmp_path):
    p = synth.make_config("mini-6")
    f = tmp_path / "p.txt"
    synth.write_bal(f, p)
    q = synth.read_bal(f)
    assert q.shape == p.shape
    assert np.array_equal(q.cam_idx, p.cam_idx) and np.array_equal(q.pt_idx, p.pt_idx)
    assert np.array_equal(q.obs, p.obs) and np.array_equal(q.cameras, p.cameras) and np.array_equal(q.points, p.points)


def test_schur_fixture_matches_reference_inputs():
    p = synth.schur_test_fixture()                      # tests/schur.cu:52-78
    assert p.shape == (2, 3, 6)
    assert p.cameras[0, 6] == 800.0 and p.cameras[1, 8] == 0.0009
    assert p.points[0, 0] == np.float64(np.float32(0.1))  # float literal 0.1f widened to double
    assert list(zip(p.cam_idx, p.pt_idx)) == [(0, 0), (1, 0), (0, 1), (1, 1), (0, 2), (1, 2)]
