"""The speculative sort of the FP64 pipeline (amhip_sort.hip: dsm_sort): a DSM
call of >= 2^20 points whose context saw such a call before sizes the regions of both scatter
passes from THAT call's exact (k1, k2) counts instead of counting first; the counting pipeline is
launched behind it as fixed grids that leave at once unless a region overflowed.  Whatever
happens -- a hit, a miss with the exact passes running behind it, the counting calls after a
miss -- the heights are those of a context that always counts first (tuning knob sort_no_speculation),
bit for bit: the default mode's floats do not depend on the order of the points."""
import numpy as np
import pytest

import oracle_ffi as O

pytestmark = pytest.mark.gpu


def _clouds(g, lx, ly):
    rng = np.random.default_rng(4242)
    n = 1_400_000

    def uniform(seed, m=n, x0=-0.5, x1=0.5, y0=-0.5, y1=0.5):
        r = np.random.default_rng(seed)
        return np.c_[g.pos_x + r.uniform(x0 * lx - 2, x1 * lx + 2, m), g.pos_y + r.uniform(y0 * ly - 2, y1 * ly + 2, m),
                     300.0 + r.uniform(-2.0, 2.0, m)]
    a = uniform(1)
    a2 = a[rng.permutation(n)]                               # the same set in another order
    a3 = uniform(2)                                          # another sample of the same distribution
    # 13/16 of the points on one quarter of the map: its bin rows AND column blocks hold 3 x A's counts
    b = np.r_[uniform(3, n // 4), uniform(4, 3 * n // 4, 0.0, 0.5, 0.0, 0.5)]
    return [("A", a), ("A permuted", a2), ("A'", a3), ("B", b), ("B", b), ("A", a)]


def _run(A, st, clouds, radius=1):
    out, stats = [], []
    with A.AerialGridMap(st) as m:
        m.set_dsm_sort_reuse(True)          # (opt-in since round 5)
        dsm = A.Dsm(A.DsmSettings(radius), m)
        for _, pts in clouds:
            m.reset()
            dsm.process(pts, m)
            out.append(m.get("elevation"))
            stats.append(m.dsm_sort_stats())
    return out, stats


def test_speculative_sort_gives_the_counting_sorts_heights(tuning):
    import aerial_mapper_amd as A
    res, rows, cols = 0.5, 1536, 1280
    lx, ly = rows * res, cols * res
    g = O.make_grid(lx, ly, res, 50.0, -20.0)
    st = A.GridMapSettings(g.pos_x, g.pos_y, g.length_x, g.length_y, g.resolution)
    clouds = _clouds(g, lx, ly)
    tuning(sort_no_speculation=None)
    got, stats = _run(A, st, clouds)
    tuning(sort_no_speculation=1)
    want, stats_off = _run(A, st, clouds)
    for k, (name, _) in enumerate(clouds):
        a, b = got[k], want[k]
        eq = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
        assert eq.all(), (k, name, int((~eq).sum()))
    # the same set permuted: the same floats (and both went through the speculative passes)
    assert np.array_equal(got[0].view(np.uint32), got[1].view(np.uint32))
    assert stats_off[-1]["speculative"] == 0 and stats_off[-1]["three_pass_calls"] == len(clouds)
    s = stats
    assert s[0]["speculative"] == 0                      # nothing to plan with yet
    assert s[1]["speculative"] == 1 and s[2]["speculative"] == 2   # same set, same distribution: hits
    assert s[3]["speculative"] == 3                      # B starts speculatively ...
    assert s[3]["overflowed"] == 1                       # ... and overflows (get() synchronised: the word has arrived)
    assert s[4]["speculative"] == 3 and s[4]["counting_calls_left"] > 0   # counts first after the miss
    assert s[2]["overflowed"] == 0
    # against the oracle, the call that ran the exact passes BEHIND a failed speculation
    rc, elev, _ = O.dsm_process(clouds[3][1], g, 1, 0.0, 0.0)
    assert rc == O.OK
    ok = ~np.isnan(elev)
    assert (np.isnan(got[3]) == ~ok).all()
    assert np.abs(got[3][ok].astype(np.float64) - elev[ok]).max() <= 1e-6


@pytest.mark.parametrize("knobs", [
    {"p3_min_points": "0"},
    {"p3_min_points": "0", "p3_target": "48"},
    {"p3_min_points": "0", "p3_target": "4000", "p3_cap": "64", "p3_rounds_cap": "96"},
    {"p3_min_points": "0", "p3_target": "4000", "p3_cap": "64", "p3_rounds_cap": "96",
     "p3_rounds_reread": "1"},
], ids=["three-pass", "three-pass-many-blocks", "three-pass-rounds", "three-pass-rounds-reread"])
def test_speculative_sort_on_every_placement_path(knobs):
    """The speculative passes feed every form of the placement pass (one LDS image, the big image,
    rounds from registers, rounds re-reading: forced at test size like
    test_gpu_parity.py::test_dsm_every_sort_path_matches) from regions that are NOT the final
    positions, and the exact passes behind a miss place over-full sub-partitions directly: hits on a
    uniform and on a clustered cloud, a miss from one to the other -- against the oracle and, bit for
    bit, against the counting sort."""
    import os
    import subprocess
    import sys
    import scenarios as S
    code = (
        "import os, sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np, oracle_ffi as O, scenarios as S, aerial_mapper_amd as A\n"
        "sc = S.Scene(150.0, 110.0, 0.5, 70000, seed=182)\n"
        "g = sc.grid\n"
        "rng = np.random.default_rng(15)\n"
        "U = sc.points\n"
        "U2 = np.ascontiguousarray(U[rng.permutation(U.shape[0])])\n"
        "dense = np.c_[rng.uniform(g.pos_x + 10.0, g.pos_x + 25.0, 40000), rng.uniform(g.pos_y - 20.0, g.pos_y - 5.0, 40000),\n"
        "              400.0 + rng.uniform(-0.5, 0.5, 40000)]\n"
        "C = np.ascontiguousarray(np.concatenate([U[:30000], dense]))\n"
        "st = A.GridMapSettings(g.pos_x, g.pos_y, g.length_x, g.length_y, g.resolution)\n"
        "def oracle(p):\n"
        "    rc, want, _ = O.dsm_process(p, g); assert rc == O.OK; return want\n"
        "want = {'U': oracle(U), 'U2': oracle(U2), 'C': oracle(C)}\n"
        "clouds = {'U': U, 'U2': U2, 'C': C}\n"
        "def run(seq, spec):\n"
        "    from aerial_mapper_amd import hip_lib\n"
        "    hip_lib.set_tuning('sort_no_speculation', None if spec else 1)\n"
        "    out = []\n"
        "    with A.AerialGridMap(st) as m:\n"
        "        m.set_dsm_precision(True); m.set_dsm_sort_reuse(True)\n"
        "        for name in seq:\n"
        "            m.reset(); A.Dsm(A.DsmSettings(), m).process(clouds[name], m)\n"
        "            e = m.get('elevation'); S.assert_dsm_close(e, want[name], tol=1e-6); out.append(e)\n"
        "        return out, m.dsm_sort_stats()\n"
        "for seq, hits, misses in ((('U', 'U', 'U2'), 2, 0), (('C', 'C', 'C'), 2, 0), (('U', 'C', 'C'), 1, 1)):\n"
        "    a, sa = run(seq, True); b, sb = run(seq, False)\n"
        "    assert sa['speculative'] == hits and sa['overflowed'] == misses and sb['speculative'] == 0, (seq, sa, sb)\n"
        "    for x, y in zip(a, b):\n"
        "        assert ((x.view(np.uint32) == y.view(np.uint32)) | (np.isnan(x) & np.isnan(y))).all(), seq\n"
        "print('SPEC_PATH_OK')\n" % (S.__file__.rsplit('/tests/', 1)[0], S.__file__.rsplit('/', 1)[0]))
    from conftest import tuning_env
    env = tuning_env(**knobs)
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0 and b"SPEC_PATH_OK" in r.stdout, r.stdout.decode()[-2000:]
