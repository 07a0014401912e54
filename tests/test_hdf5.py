"""The built-in HDF5 subset (articulatory_amd/utils/hdf5.py) on files of its own writer, plus hand-assembled variants of the structures
the reader must follow (continuation blocks, chunked layout, nested group, superblock 1).  No h5py-written fixture exists in this image;
the byte layouts below are written from the HDF5 File Format Specification, field by field."""
import struct

import numpy as np
import pytest

from articulatory_amd.utils import hdf5 as H


def test_round_trip_of_the_reference_s_files(tmp_path):
    rng = np.random.default_rng(0)
    p = str(tmp_path / "stats.h5")
    mean, scale = rng.standard_normal(13).astype(np.float32), rng.random(13).astype(np.float64)
    H.write_hdf5(p, "mean", mean)
    H.write_hdf5(p, "scale", scale)
    feats = rng.standard_normal((57, 12)).astype(np.float32)
    H.write_hdf5(p, "feats", feats)
    H.write_hdf5(p, "ids", np.arange(7, dtype=np.int64))
    assert H.list_hdf5(p) == ["feats", "ids", "mean", "scale"]
    assert np.array_equal(H.read_hdf5(p, "mean"), mean) and H.read_hdf5(p, "scale").dtype == np.float64
    assert np.array_equal(H.read_hdf5(p, "scale"), scale) and np.array_equal(H.read_hdf5(p, "feats"), feats)
    assert np.array_equal(H.read_hdf5(p, "/ids"), np.arange(7))
    with pytest.raises(KeyError):
        H.read_hdf5(p, "wave")
    with pytest.raises(KeyError):
        H.write_hdf5(p, "mean", mean, is_overwrite=False)
    H.write_hdf5(p, "mean", mean * 2)                     # overwrite keeps the others
    assert np.array_equal(H.read_hdf5(p, "mean"), mean * 2) and np.array_equal(H.read_hdf5(p, "feats"), feats)
    raw = open(p, "rb").read()
    assert raw[:8] == b"\x89HDF\r\n\x1a\n" and raw[8] == 0 and struct.unpack_from("<Q", raw, 40)[0] == len(raw)   # superblock 0, EOF address
    with pytest.raises(FileNotFoundError):
        H.read_hdf5(str(tmp_path / "none.h5"), "mean")
    bad = tmp_path / "bad.h5"
    bad.write_bytes(b"not hdf5 at all")
    with pytest.raises(H.HDF5Error):
        H.read_hdf5(str(bad), "mean")


def test_many_datasets_and_register_stats_path(tmp_path):
    p = str(tmp_path / "many.h5")
    data = {f"utt{i:03d}": np.full((i % 5 + 1, 3), i, np.float32) for i in range(40)}
    H.write_file(p, data)
    assert H.list_hdf5(p) == sorted(data)
    for k, v in data.items():
        assert np.array_equal(H.read_hdf5(p, k), v)


def test_reader_follows_continuation_chunks_and_nested_groups(tmp_path):
    """A file assembled by hand: superblock 1, root group -> group "g" -> dataset "x" whose object header continues in a second block,
    chunked (2 x 3 chunks of a 3 x 5 float32 array, v1 chunk B-tree), plus a compact dataset "c"."""
    U = H.UNDEF
    arr = np.arange(15, dtype=np.float32).reshape(3, 5)
    buf = bytearray(4096)
    at = {"root": 104, "bt_root": 160, "heap_root": 704, "heapd_root": 736, "snod_root": 760, "g": 1096, "bt_g": 1152, "heap_g": 1696,
          "heapd_g": 1728, "snod_g": 1760, "x": 2096, "x_cont": 2200, "cbtree": 2300, "chunks": 2800, "c": 3300}
    buf[0:8] = H.SIG
    struct.pack_into("<8B", buf, 8, 1, 0, 0, 0, 0, 8, 8, 0)
    struct.pack_into("<HHI", buf, 16, 4, 16, 0)
    struct.pack_into("<HH", buf, 24, 32, 0)                                  # superblock 1: indexed storage K, reserved
    struct.pack_into("<4Q", buf, 28, 0, U, len(buf), U)
    struct.pack_into("<QQIIQQ", buf, 60, 0, at["root"], 1, 0, at["bt_root"], at["heap_root"])

    def group(hdr, bt, heap, heapd, snod, names):
        h = H._object_header([H._msg(0x0011, struct.pack("<QQ", bt, heap))])
        buf[hdr:hdr + len(h)] = h
        hd, offs = bytearray(8), {}
        for nm in sorted(names):
            offs[nm] = len(hd)
            raw = nm.encode() + b"\0"
            hd += raw + b"\0" * (-len(raw) % 8)
        struct.pack_into("<4sBBHQQ", buf, bt, b"TREE", 0, 0, 1, U, U)
        struct.pack_into("<QQQ", buf, bt + 24, 0, snod, offs[sorted(names)[-1]])
        struct.pack_into("<4sB3xQQQ", buf, heap, b"HEAP", 0, len(hd), 1, heapd)
        buf[heapd:heapd + len(hd)] = hd
        struct.pack_into("<4sBBH", buf, snod, b"SNOD", 1, 0, len(names))
        for i, nm in enumerate(sorted(names)):
            struct.pack_into("<QQII16x", buf, snod + 8 + i * 40, offs[nm], names[nm], 0, 0)

    group(at["root"], at["bt_root"], at["heap_root"], at["heapd_root"], at["snod_root"], {"g": at["g"], "c": at["c"]})
    group(at["g"], at["bt_g"], at["heap_g"], at["heapd_g"], at["snod_g"], {"x": at["x"]})
    # dataset x: first block holds dataspace (version 2) + a continuation message; the second block datatype + chunked layout
    space = struct.pack("<BBBB", 2, 2, 0, 1) + struct.pack("<2Q", 3, 5)
    layout = struct.pack("<BBB", 3, 2, 3) + struct.pack("<Q", at["cbtree"]) + struct.pack("<3I", 2, 3, 4)
    second = H._msg(3, H._datatype(np.float32)) + H._msg(8, layout)
    first = [H._msg(1, space), H._msg(0x0010, struct.pack("<QQ", at["x_cont"], len(second)))]
    h = struct.pack("<BBHII4x", 1, 0, 4, 1, sum(len(m) for m in first)) + b"".join(first)
    buf[at["x"]:at["x"] + len(h)] = h
    buf[at["x_cont"]:at["x_cont"] + len(second)] = second
    # chunk B-tree (leaf): 4 chunks, keys = (size, filter mask, offsets..., 0)
    struct.pack_into("<4sBBHQQ", buf, at["cbtree"], b"TREE", 1, 0, 4, U, U)
    p, cpos = at["cbtree"] + 24, at["chunks"]
    for r0 in (0, 2):
        for c0 in (0, 3):
            chunk = np.zeros((2, 3), np.float32)
            blk = arr[r0:r0 + 2, c0:c0 + 3]
            chunk[:blk.shape[0], :blk.shape[1]] = blk
            struct.pack_into("<II3Q", buf, p, 24, 0, r0, c0, 0)
            struct.pack_into("<Q", buf, p + 32, cpos)
            buf[cpos:cpos + 24] = chunk.tobytes()
            p, cpos = p + 40, cpos + 24
    # compact dataset c
    cdata = np.array([1.5, -2.5], np.float64)
    clayout = struct.pack("<BBH", 3, 0, 16) + cdata.tobytes()
    h = H._object_header([H._msg(1, struct.pack("<BBB5x", 1, 1, 0) + struct.pack("<Q", 2)), H._msg(3, H._datatype(np.float64)), H._msg(8, clayout)])
    buf[at["c"]:at["c"] + len(h)] = h
    path = tmp_path / "hand.h5"
    path.write_bytes(buf)
    if H._h5py() is None:
        assert np.array_equal(H.read_hdf5(str(path), "g/x"), arr)
        assert np.array_equal(H.read_hdf5(str(path), "c"), cdata)
        assert H.list_hdf5(str(path)) == ["c", "g"]


def test_newer_superblocks_fail_loudly(tmp_path):
    if H._h5py() is not None:
        pytest.skip("h5py handles these")
    raw = bytearray(64)
    raw[:8] = H.SIG
    raw[8] = 2
    p = tmp_path / "v2.h5"
    p.write_bytes(raw)
    with pytest.raises(H.HDF5Error, match="superblock version 2"):
        H.read_hdf5(str(p), "mean")


def test_stats_and_dump_dir_through_the_product_paths(tmp_path):
    """register_stats(stats.h5) (hifigan.py:280-296) and the hdf5 dump-dir listing / loading of the decode CLI (decode.py:207-222)."""
    import torch

    from articulatory_amd.bin import decode as D
    from articulatory_amd.models import HiFiGANGenerator
    from conftest import E2W_PARAMS

    mean, scale = np.linspace(-1, 1, 12).astype(np.float32), np.linspace(0.5, 2, 12).astype(np.float32)
    stats = str(tmp_path / "stats.h5")
    H.write_hdf5(stats, "mean", mean)
    H.write_hdf5(stats, "scale", scale)
    g = HiFiGANGenerator(**dict(E2W_PARAMS, channels=32))
    g.register_stats(stats)
    assert torch.equal(g.mean, torch.from_numpy(mean)) and torch.equal(g.scale, torch.from_numpy(scale))
    dump = tmp_path / "dump" / "part1"
    dump.mkdir(parents=True)
    feats = {f"utt{i}": np.random.default_rng(i).standard_normal((10 + i, 12)).astype(np.float32) for i in range(3)}
    for k, v in feats.items():
        H.write_hdf5(str(dump / f"{k}.h5"), "feats", v)
        H.write_hdf5(str(dump / f"{k}.h5"), "wave", np.zeros(80 * len(v), np.float32))
    pairs = D.list_features(dumpdir=str(tmp_path / "dump"), fmt="hdf5")
    assert [u for u, _ in pairs] == sorted(feats)
    assert [D.npy_frames(p) for _, p in pairs] == [10, 11, 12]
    for utt, arr in D.load_features(pairs):
        assert np.array_equal(arr, feats[utt])
    with pytest.raises(ValueError, match="hdf5 or npy"):
        D.list_features(dumpdir=str(tmp_path / "dump"), fmt="mat")


def test_scp_values_npy_h5_and_kaldi_ark(tmp_path):
    """The three feats.scp value forms of the reference's loaders (scp_dataset.py:20-46), incl. a binary Kaldi archive assembled by hand
    (key, "\0B", "FM " / "DM ", "\4" rows, "\4" cols, row-major data)."""
    from articulatory_amd.bin import decode as D
    from articulatory_amd.utils.scp import load_scp_value

    rng = np.random.default_rng(2)
    a, b, c = (rng.standard_normal((n, 5)).astype(np.float32) for n in (7, 9, 4))
    np.save(tmp_path / "a.npy", a)
    H.write_hdf5(str(tmp_path / "b.h5"), "feats", b)
    H.write_hdf5(str(tmp_path / "b.h5"), "other", b * 2)
    ark = tmp_path / "feats.ark"
    offs = {}
    with open(ark, "wb") as f:
        for key, m, tok in (("c", c, b"FM "), ("d", c.astype(np.float64) * 3, b"DM ")):
            f.write(key.encode() + b" ")
            offs[key] = f.tell()
            f.write(b"\0B" + tok + b"\4" + struct.pack("<i", m.shape[0]) + b"\4" + struct.pack("<i", m.shape[1]) + m.tobytes())
    scp = tmp_path / "feats.scp"
    scp.write_text(f"a {tmp_path / 'a.npy'}\nb {tmp_path / 'b.h5'}\nb2 {tmp_path / 'b.h5'}:other\nc {ark}:{offs['c']}\nd {ark}:{offs['d']}\n")
    pairs = D.list_features(feats_scp=str(scp))
    got = dict(D.load_features(pairs))
    assert np.array_equal(got["a"], a) and np.array_equal(got["b"], b) and np.array_equal(got["b2"], b * 2)
    assert np.array_equal(got["c"], c) and np.array_equal(got["d"], c.astype(np.float64) * 3)
    assert [D.npy_frames(p) for _, p in pairs] == [7, 9, 9, 4, 4]
    (tmp_path / "bad.scp").write_text("x /path/x.mat\n")
    with pytest.raises(ValueError, match="Not supported feats.scp type"):
        D.list_features(feats_scp=str(tmp_path / "bad.scp"))
    with pytest.raises(ValueError, match="compressed"):
        with open(tmp_path / "cm.ark", "wb") as f:
            f.write(b"k \0BCM ")
        load_scp_value(f"{tmp_path / 'cm.ark'}:2")
