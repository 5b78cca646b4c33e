"""The generated last stage of the 3 x 3 x 3 rank kernel (cupyimg_amd/csrc/median27_net.hpp) is what
scripts/gen_median27_network.py emits from its committed wire placement, and that network takes the median of a window on
every input the partial order of a z-, x-, y-sorted cube allows (all 980 monotone 0/1 labelings) and on random windows."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gen():
    spec = importlib.util.spec_from_file_location("gen_median27_network", os.path.join(ROOT, "scripts", "gen_median27_network.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_candidates_are_the_19_positions_that_can_hold_the_median():
    g = _gen()
    g.set_rank(13)
    assert len(g.CAND) == 19 and len(g.labelings()) == 980
    out = [c for c in g.CELLS if c not in g.CAND]
    below = [c for c in out if (3 - c[0]) * (3 - c[1]) * (3 - c[2]) > 14]      # >= 14 samples known above: rank <= 12
    above = [c for c in out if (c[0] + 1) * (c[1] + 1) * (c[2] + 1) > 14]
    assert len(below) == 4 and len(above) == 4 and not set(below) & set(above)
    for r in range(1, 26):                                  # every rank: candidates + known below + known above = the window
        cand, nb = g.candidates(r)
        assert 3 <= len(cand) <= 19 and 0 <= r - nb < len(cand)


def test_network_is_correct_and_the_header_is_the_generated_one(tmp_path):
    g = _gen()
    wires = g.load_wires()
    assert sorted(wires) == list(range(1, 26))
    code, result = g.network_for(13, wires[13])
    assert len(code) <= 62
    path = tmp_path / "median27_net.hpp"
    g.emit(str(path))                                        # verifies every rank: all labelings + 4 000 random windows each
    committed = open(os.path.join(ROOT, "cupyimg_amd", "csrc", "median27_net.hpp")).read()
    assert path.read_text() == committed
