"""CPU tests of the text-format oracle (oracle/amo_io.cc = the reference's own
extraction loops on an in-memory std::istream) and of the host-side pose reader."""
import numpy as np

import oracle_ffi as O


def test_point_cloud_loop_semantics():
    # aerial-mapper-io.cc:327-347: four tokens per record, z > -100 filter,
    # stop at the first failed extraction, incomplete tail dropped
    t = b"1.5 2.25 3 7\n-4e2 .5 -100.0 9\n+1. 2 -99.999 12\n7 8 9\n"
    xyz, inten = O.io_load_point_cloud(t)
    assert xyz.tolist() == [[1.5, 2.25, 3.0], [1.0, 2.0, -99.999]]
    assert inten.tolist() == [7, 12]
    # an intensity with trailing garbage still completes its record; then the stream is dead
    xyz, inten = O.io_load_point_cloud(b"1 2 3 4abc 5 6 7 8\n")
    assert xyz.tolist() == [[1.0, 2.0, 3.0]] and inten.tolist() == [4]
    # a non-number ends the stream before its record
    xyz, _ = O.io_load_point_cloud(b"1 2 3 4\n5 x 7 8\n9 9 9 9\n")
    assert xyz.shape == (1, 3)
    # overflow sets failbit (libstdc++), underflow does not
    xyz, _ = O.io_load_point_cloud(b"1e999 2 3 4\n")
    assert xyz.shape == (0, 3)
    xyz, _ = O.io_load_point_cloud(b"1e-400 4.9e-324 3 4\n")
    assert xyz.tolist() == [[0.0, 5e-324, 3.0]]
    assert O.io_load_point_cloud(b"")[0].shape == (0, 3)
    assert O.io_load_point_cloud(b" \n\t \r\n")[0].shape == (0, 3)


def test_pose_reader_matches_reference_loop(tmp_path):
    from aerial_mapper_amd import io as AIO
    rng = np.random.default_rng(1)
    poses = rng.normal(size=(13, 7)) * [100, 100, 10, 1, 1, 1, 1]
    text = "".join(" ".join("%.17g" % v for v in p) + "\n" for p in poses).encode()
    want = O.io_load_poses(text)
    assert np.array_equal(want, poses)
    f = tmp_path / "poses.txt"
    f.write_bytes(text + b"1 2 3")          # incomplete tail is dropped
    got = AIO.load_poses_text(str(f))
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
