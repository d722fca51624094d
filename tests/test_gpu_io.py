"""GPU parity of the point-cloud text loader (amhip_io.hip: tokeniser +
Eisel-Lemire decimal->double on the device) against the reference's own
iostream loop (oracle/amo_io.cc).  Doubles and ints must match bit for bit."""
import numpy as np
import pytest

import oracle_ffi as O

pytestmark = pytest.mark.gpu


def _gpu(text):
    from aerial_mapper_amd import io as AIO
    cloud = AIO.parse_point_cloud_text(text)
    xyz, inten = cloud.to_host()
    return xyz, inten, cloud


def _same(text):
    want_xyz, want_int = O.io_load_point_cloud(text)
    xyz, inten, cloud = _gpu(text)
    assert xyz.shape == want_xyz.shape, (xyz.shape, want_xyz.shape)
    bad = np.argwhere(xyz.view(np.uint64) != want_xyz.view(np.uint64))
    assert bad.shape[0] == 0, (bad[:5], xyz[tuple(bad[0])], want_xyz[tuple(bad[0])])
    assert np.array_equal(inten, want_int)
    return cloud


def test_semantics_of_the_extraction_loop():
    for t in [b"1.5 2.25 3 7\n-4e2 .5 -100.0 9\n+1. 2 -99.999 12\n7 8 9\n",
              b"1 2 3 4abc 5 6 7 8\n", b"1 2 3 4\n5 x 7 8\n9 9 9 9\n", b"1e999 2 3 4\n",
              b"1e-400 4.9e-324 3 4\n", b"", b" \n\t \r\n", b"1 2 3 4", b"1 2 3 4\r\n5 6 7 8\r\n",
              b"\t1\v2\f3  4 \n", b"1 2 3 99999999999\n", b"1 2 3 -2147483648\n5 6 7 2147483647\n",
              b"1 2 3 4\n1e 2 3 4\n", b"- 2 3 4\n", b"1 2 3 +\n", b"-0.0 0e5 -0 1\n",
              b"1 2 3 4 5 6 7 8 9 10 11 12 13"]:
        _same(t)


def test_adversarial_tokens_follow_the_iostream_loop():
    """ the tokens a hand-written parser gets wrong.  The oracle IS the C++ library's
    own `infile >> x >> y >> z >> intensity` (oracle/amo_io.cc runs aerial-mapper-io.cc:309-347's loop
    on a stream over the buffer): nothing about number syntax is restated there.  aerial-mapper-io.cc
    itself needs GDAL, OpenCV, aslam and glog headers -- not in the image, so it is not compiled."""
    for t in [b"nan 2 3 4\n5 6 7 8\n", b"1 2 inf 4\n", b"1 NaN 3 4\n", b"-inf 2 3 4\n", b"1 2 infinity 4\n",
              b"0x10 2 3 4\n", b"1 2 3 0x1F\n", b"1 2 3 010\n",            # (no hex; a leading 0 is decimal)
              b"1 2 3 1e5\n6 7 8 9\n", b"1 2 3 4.7 5 6 7 8\n",              # (an int stops at 'e' / '.')
              b"1 2 -100 4\n5 6 -100.00000000000001 7\n8 9 -99.99999999999999 1\n",   # (z > -100, strictly)
              b"1,5 2 3 4\n", b"1..2 3 4 5\n", b"1e+ 2 3 4\n", b"1e+5e3 2 3 4\n", b"--1 2 3 4\n",
              b"+-1 2 3 4\n", b"1 2 3 4\n5 6 7", b"1 2 3 4\n5 6 7 8 ", b"1 2 3 4\n5 6 7 8\n\n\n",
              b"1 2 3 4\x00 5 6 7 8\n", b"1 2 3 4\n# comment\n5 6 7 8\n", b"1 2 3 4;\n5 6 7 8\n",
              b"1e5 +3 .5 7\n", b"1E+05 -3. 5.e-1 -7\n", b".e5 2 3 4\n", b". 2 3 4\n", b"e5 2 3 4\n",
              b"1e0000000000000000000005 2 3 4\n", b"1e-0000000000000000000005 2 3 4\n",
              b"00000000000000000000000000000000000001 2 3 4\n",
              b"1 2 3 2147483648\n5 6 7 8\n", b"1 2 3 -2147483649\n5 6 7 8\n",   # (int overflow: failbit)
              b"1 2 3 +5\n", b"1 2 3 -0\n", b"1 2 3 - 5\n",
              "١ 2 3 4\n".encode(), b"\xef\xbb\xbf1 2 3 4\n",                 # (non-ASCII digits, a BOM)
              b"1\n2\n3\n4\n5\n6\n7\n8\n", b"1 2 3 4 5 6 7 8 9 10 11\n",
              # one whitespace-delimited token, several extractions: the next one resumes where the last stopped
              b"1-2-3-4\n", b"1-2-3-4-5-6-7-8\n", b"1.5.25 3 4\n", b"1e5-3+2 7\n", b"1 2 3.5-4\n",
              b"1 2 3 4-5 6 7 8\n", b"1 2 3 4+5 6 7 8\n", b"1 2 3 4.5 6 7 8.9 1 2 3\n",
              b"0x10 2 3 4\n", b"1 2 3 4\n5.5.5.5 6\n7 8 9 1\n", b"1 2 3 4e5 6 7 8\n",
              b"-.5-.5-.5-5\n" * 3, (b"1-2-3-4" * 50) + b"\n",
              b"1 2 3 4\n" * 100 + b"5 6 7 8.5 6 7 8\n" + b"1 2 3 4\n" * 100,
              b"9" * 1000 + b" 2 3 4\n", b"1 2 3 4\n" + b"0" * 2000 + b".5 2 3 4\n5 6 7 8\n",
              b"1." + b"9" * 900 + b".5 3 4\n"]:
        _same(t)


def test_hard_decimal_to_double_cases():
    hard = ["0.1", "0.2", "0.3", "1e23", "8.5e-1", "9007199254740993", "9007199254740992",
            "9007199254740991", "4.9e-324", "2.4703282292062327e-324", "2.4703282292062328e-324",
            "2.2250738585072014e-308", "2.2250738585072011e-308", "1.7976931348623157e308",
            "1.7976931348623158e308", "123456789012345678901234567890", "0." + "0" * 30 + "12345",
            "1" + "0" * 25, "3.1415926535897932384626433832795028841971", "1.00000000000000011102230246251565404236316680908203125",
            "1.00000000000000011102230246251565404236316680908203124", "1.00000000000000011102230246251565404236316680908203126",
            "5e-324", "1e-323", "1e-310", "1.5e-310", "179769313486231580793728971405303415079934132710037826936173778980444968292764750946649017977587207096330286416692887910946555547851940402630657488671505820681908902000708383676273854845817711531764475730270069855571366959622842914819860834936475292719074168444365510704342711559699508093042880177904174497791",
            "6.02214076e23", "1e22", "1e21", "1e-22", "123456789.123456789e-20", "00000123.4500", "+.5e+3", "1E5", "1.e2"]
    lines = []
    for k, h in enumerate(hard):
        lines.append("%s -%s %d.5 %d" % (h, h.lstrip("+"), k, k))
    cloud = _same("\n".join(lines).encode())
    assert cloud.n == len(hard)
    assert cloud.strtod_tokens > 0      # the > 19-digit ties went through the host's strtod


@pytest.mark.parametrize("fmt", ["%.3f", "%.15g", "%.17g", "%.9e", "%r"])
def test_random_files_bit_exact(fmt):
    rng = np.random.default_rng(7)
    n = 200000
    xyz = np.empty((n, 3))
    xyz[:, 0] = rng.uniform(460000.0, 470000.0, n)       # UTM-like eastings
    xyz[:, 1] = rng.uniform(5.2e6, 5.3e6, n)
    xyz[:, 2] = rng.normal(400.0, 300.0, n)              # some below the -100 cut
    xyz[::97, 2] = -100.0
    xyz[::89, 2] = -99.99999999999999
    inten = rng.integers(-5, 256, n)
    if fmt == "%r":
        body = "".join("%r %r %r %d\n" % (a, b, c, i) for (a, b, c), i in zip(xyz.tolist(), inten.tolist()))
    else:
        f4 = fmt + " " + fmt + " " + fmt + " %d\n"
        body = "".join(f4 % (a, b, c, i) for (a, b, c), i in zip(xyz.tolist(), inten.tolist()))
    cloud = _same(body.encode())
    assert 0.9 * n < cloud.n < n


def test_wide_dynamic_range_bit_exact():
    rng = np.random.default_rng(11)
    n = 150000
    mant = rng.uniform(1.0, 10.0, 3 * n)
    expo = rng.integers(-330, 309, 3 * n)
    sign = rng.choice([-1.0, 1.0], 3 * n)
    toks = ["%s%.*ge%d" % ("-" if s < 0 else "", int(p), m, e)
            for m, e, s, p in zip(mant.tolist(), expo.tolist(), sign.tolist(),
                                  rng.integers(1, 19, 3 * n).tolist())]
    # z must stay finite and > -100 for most records: use moderate z
    z = rng.uniform(-150.0, 500.0, n)
    lines = ["%s %s %.17g %d" % (toks[3 * k], toks[3 * k + 1], z[k], k % 251) for k in range(n)]
    # overflowing x values would end the stream: clamp those exponents
    text = "\n".join(lines).replace("e308", "e307").encode()
    _same(text)


def test_device_cloud_feeds_the_dsm_without_leaving_hbm(tmp_path):
    import aerial_mapper_amd as A
    from aerial_mapper_amd import io as AIO, synth
    pts = synth.make_points(60000, 42.0, 3)
    inten = (np.arange(pts.shape[0]) % 200).astype(np.int32)
    f = tmp_path / "cloud.txt"
    with open(f, "w") as fh:
        for (x, y, z), i in zip(pts.tolist(), inten.tolist()):
            fh.write("%.15g %.15g %.15g %d\n" % (x, y, z, i))
    cloud = AIO.load_point_cloud_text(str(f))
    want_xyz, want_int = O.io_load_point_cloud(open(f, "rb").read())
    assert cloud.n == want_xyz.shape[0] == pts.shape[0]
    g = O.make_grid(80.0, 60.0, 0.5)
    st = A.GridMapSettings(0.0, 0.0, 80.0, 60.0, 0.5)
    with A.AerialGridMap(st) as m:
        A.Dsm(A.DsmSettings(), m).process(cloud.xyz, m)
        got = m.get("elevation")
        A.OrthoFromPcl(A.OrthoFromPclSettings()).process(cloud.xyz, cloud.intensities, m)
        got_o = m.get("ortho")
    rc, want, _ = O.dsm_process(want_xyz, g)
    assert rc == O.OK
    import scenarios as S
    S.assert_dsm_close(got, want)
    rc, want_o = O.ortho_from_pcl(want_xyz, want_int, g)
    np.testing.assert_allclose(got_o, want_o, rtol=0, atol=1e-3)
    with pytest.raises(A.AmhipError):
        AIO.load_point_cloud_text("")
