"""Host-side pieces of the skimage facade: structuring-element masks against the
known answers the reference's tests hold (skimage/morphology/tests/test_selem.py:113-200)
and closed forms."""
import numpy as np

from cupyimg_amd.skimage.morphology import _host_ball_like, _host_octagon, _host_star

OCTAGON_5_3 = """
00011111000
00111111100
01111111110
11111111111
11111111111
11111111111
11111111111
11111111111
01111111110
00111111100
00011111000"""

STAR_4 = """
0000001000000
0000011100000
0011111111100
0011111111100
0011111111100
0111111111110
1111111111111
0111111111110
0011111111100
0011111111100
0011111111100
0000011100000
0000001000000"""


def _mask(text):
    return np.array([[int(c) for c in row] for row in text.split()], dtype=np.uint8)


def test_octagon_known_answers():
    assert np.array_equal(_host_octagon(5, 3), _mask(OCTAGON_5_3))
    assert np.array_equal(_host_octagon(1, 1), _mask("010 111 010"))
    assert _host_octagon(3, 0).all() and _host_octagon(3, 0).shape == (3, 3)


def test_star_known_answers():
    assert np.array_equal(_host_star(4), _mask(STAR_4))
    assert np.array_equal(_host_star(1), np.ones((3, 3), np.uint8))


def test_ball_like_masks():
    assert np.array_equal(_host_ball_like(1, 2, 1, np.uint8), _mask("010 111 010"))
    assert np.array_equal(_host_ball_like(2, 2, 2, np.uint8), _mask("00100 01110 11111 01110 00100"))
    d3 = _host_ball_like(3, 2, 2, np.uint8)
    assert d3.shape == (7, 7) and d3.sum() == 29
    assert _host_ball_like(1, 3, 1, np.uint8).sum() == 7
    assert _host_ball_like(2, 3, 2, np.uint8).sum() == 33
    assert _host_ball_like(2, 3, 1, np.uint8).sum() == 25
