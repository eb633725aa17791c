"""The data formats either side of the hot path: Map::Save / Map::Load keyframe records
(perfect/src/Map.cc:143-187, :320-430) and the ORB vocabulary files (text / binary).  Host code of liborbfe.so checked
against independently written numpy / pure-Python twins of the reference's field-by-field writers.  CPU; the device
packer is in the gpu-marked test at the end."""
import os
import struct

import numpy as np
import pytest

from orb_slam2_ssd_semantic_amd import KP_DTYPE, VocabularyFile, mapio
from orb_slam2_ssd_semantic_amd.synth import regular_vocabulary

# twin of _WriteKeyFrame's per-feature writes (:363-380): 5 floats, int octave, 32 bytes, unsigned long
REC = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"),
                ("desc", "u1", 32), ("mp", "<u8")])
assert REC.itemsize == 64


def _features(seed, n):
    rng = np.random.default_rng(seed)
    k = np.zeros(n, KP_DTYPE)
    for f in ("x", "y", "size", "angle", "response"):
        k[f] = rng.uniform(0, 640, n).astype(np.float32)
    k["octave"] = rng.integers(0, 8, n)
    k["class_id"] = -1
    d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    mp = rng.integers(0, 5000, n).astype(np.uint64)
    mp[rng.random(n) < 0.3] = mapio.ULONG_MAX
    return k, d, mp


def twin_keyframe(kid, ts, t, q, k, d, mp):
    r = np.zeros(len(k), REC)
    for f in ("x", "y", "size", "angle", "response", "octave"):
        r[f] = k[f]
    r["desc"], r["mp"] = d, mp
    return struct.pack("<Qd3f4fi", kid, ts, *t, *q, len(k)) + r.tobytes()


@pytest.mark.parametrize("n", [0, 1, 7, 1004])
def test_keyframe_block_equals_twin_and_round_trips(n):
    k, d, mp = _features(n, n)
    t, q = np.float32([0.1, -2.5, 3.25]), np.float32([0.5, -0.5, 0.5, 0.5])
    blob = mapio.write_keyframe(12345678901, 1341846313.592026, t, q, k, d, mp)
    assert len(blob) == mapio.keyframe_bytes(n) == 48 + 64 * n
    assert blob == twin_keyframe(12345678901, 1341846313.592026, t, q, k, d, mp)
    kf, used = mapio.read_keyframe(blob + b"tail")
    assert used == len(blob) and kf["id"] == 12345678901 and kf["timestamp"] == 1341846313.592026
    assert np.array_equal(kf["t_cw"], t) and np.array_equal(kf["q_cw"], q)
    assert np.array_equal(kf["kps"].view(np.uint8), k.view(np.uint8)) and np.array_equal(kf["desc"], d)
    assert np.array_equal(kf["mp_index"], mp)
    none = mapio.write_keyframe(1, 0.0, t, q, k, d, None)         # no map points: every index ULONG_MAX (:376-377)
    assert (mapio.read_keyframe(none)[0]["mp_index"] == mapio.ULONG_MAX).all()
    with pytest.raises(Exception):
        mapio.read_keyframe(blob[:-1])                              # truncated block is rejected


def test_map_file_round_trip(tmp_path):
    mps = [(i * 3, (float(i), float(-i), 0.5 * i)) for i in range(40)]
    kfs = []
    for j, n in enumerate((1004, 0, 333)):
        k, d, mp = _features(100 + j, n)
        kfs.append(dict(id=10 + j, timestamp=1.5 * j, t_cw=np.float32([j, 2, 3]), q_cw=np.float32([0, 0, 0, 1]), kps=k, desc=d,
                        mp_index=mp, parent=None if j == 0 else 10, connections=[(10 + (j + 1) % 3, 17 + j)]))
    path = os.path.join(tmp_path, "map.bin")
    mapio.save_map(path, mps, kfs)
    # container layout of Map::Save (:385-430), written here by an independent twin
    twin = struct.pack("<Q", len(mps)) + b"".join(struct.pack("<Qfff", i, *p) for i, p in mps) + struct.pack("<Q", len(kfs))
    twin += b"".join(twin_keyframe(kf["id"], kf["timestamp"], kf["t_cw"], kf["q_cw"], kf["kps"], kf["desc"], kf["mp_index"]) for kf in kfs)
    for kf in kfs:
        twin += struct.pack("<QQ", mapio.ULONG_MAX if kf["parent"] is None else kf["parent"], len(kf["connections"]))
        twin += b"".join(struct.pack("<Qi", c, w) for c, w in kf["connections"])
    assert open(path, "rb").read() == twin
    mps2, kfs2 = mapio.load_map(path)
    assert mps2 == mps and len(kfs2) == 3
    for a, b in zip(kfs, kfs2):
        assert a["id"] == b["id"] and a["parent"] == b["parent"] and a["connections"] == b["connections"]
        assert np.array_equal(a["kps"].view(np.uint8), b["kps"].view(np.uint8)) and np.array_equal(a["desc"], b["desc"])


def _write_voc_text(path, voc, k, L):
    """twin of the text layout loadFromTextFile parses: header, then `parent is_leaf d0..d31 weight` per non-root node"""
    co, ci = voc["child_off"], voc["child_idx"]
    nn = len(voc["node_desc"])
    parent = np.zeros(nn, np.int64)
    for p in range(nn):
        parent[ci[co[p]:co[p + 1]]] = p
    leaf = co[1:] == co[:-1]
    with open(path, "w") as f:
        f.write(f"{k} {L} 0 0\n")
        for i in range(1, nn):
            f.write(f"{parent[i]} {int(leaf[i])} " + " ".join(str(int(b)) for b in voc["node_desc"][i]) + f" {float(voc['weight'][i])!r}\n")
    return parent, leaf


def test_vocabulary_text_and_binary_files(tmp_path):
    k, L = 4, 3
    voc = regular_vocabulary(k, L, seed=11, zero_frac=0.1)
    txt, binp = os.path.join(tmp_path, "ORBvoc.txt"), os.path.join(tmp_path, "ORBvoc.bin")
    parent, leaf = _write_voc_text(txt, voc, k, L)
    vf = VocabularyFile(txt)
    assert (vf.k, vf.L, vf.nnodes, vf.nwords, vf.scoring, vf.weighting) == (k, L, len(parent), int(leaf[1:].sum()), 0, 0)
    arr, extra = vf.arrays()
    for key in ("child_off", "child_idx", "node_desc", "word_id"):
        assert np.array_equal(arr[key][1:] if key == "node_desc" else arr[key], voc[key][1:] if key == "node_desc" else voc[key]), key
    assert np.array_equal(arr["weight"][1:], voc["weight"][1:])          # repr() round-trips doubles exactly
    assert np.array_equal(extra["parent"], parent.astype(np.uint32)) and np.array_equal(extra["is_leaf"][1:], leaf[1:].astype(np.uint8))
    # text -> binary (tool/text2binary.cc): layout against a twin writer, then binary -> arrays
    vf.save_binary(binp)
    blob = open(binp, "rb").read()
    twin = struct.pack("<IIiiii", len(parent), 41, k, L, 0, 0)
    for i in range(1, len(parent)):
        twin += struct.pack("<I", parent[i]) + voc["node_desc"][i].tobytes() + struct.pack("<f", np.float32(voc["weight"][i])) + bytes([int(leaf[i])])
    assert blob == twin
    vb = VocabularyFile(binp)
    ab, _ = vb.arrays()
    assert (vb.k, vb.L, vb.nnodes, vb.nwords) == (k, L, len(parent), vf.nwords)
    for key in ("child_off", "child_idx", "word_id"):
        assert np.array_equal(ab[key], arr[key])
    assert np.array_equal(ab["node_desc"][1:], arr["node_desc"][1:])
    assert np.array_equal(ab["weight"][1:], arr["weight"][1:].astype(np.float32).astype(np.float64))   # weights are float in the binary file
    with pytest.raises(Exception):
        VocabularyFile(os.path.join(tmp_path, "missing.txt"))
    bad = os.path.join(tmp_path, "bad.txt")
    open(bad, "w").write("99 3 0 0\n")
    with pytest.raises(Exception):
        VocabularyFile(bad)                                                # k out of loadFromTextFile's range


def test_loaded_vocabulary_transforms_like_the_oracle(oracle, tmp_path):
    """the arrays a loaded file yields drive the same BoW transform as the arrays it was written from (CPU oracle)"""
    voc = regular_vocabulary(5, 3, seed=2)
    txt = os.path.join(tmp_path, "v.txt")
    _write_voc_text(txt, voc, 5, 3)
    arr, _ = VocabularyFile(txt).arrays()
    desc = np.random.default_rng(0).integers(0, 256, (300, 32), dtype=np.uint8)
    a, b = oracle.bow_transform(voc, desc, 1), oracle.bow_transform(arr, desc, 1)
    for key in a:
        assert np.array_equal(a[key], b[key]), key


@pytest.mark.gpu
def test_device_record_packer_equals_host_writer():
    import torch
    from orb_slam2_ssd_semantic_amd import _ffi
    B, cap = 5, 1088
    rng = np.random.default_rng(5)
    n = np.array([1004, 0, 1088, 17, 500], np.int32)
    kps = np.zeros((B, cap), KP_DTYPE)
    desc = np.zeros((B, cap, 32), np.uint8)
    mp = np.zeros((B, cap), np.uint64)
    for b in range(B):
        k, d, m = _features(50 + b, cap)
        kps[b], desc[b], mp[b] = k, d, m
    dk = torch.from_numpy(kps.view(np.int32).reshape(B, cap, 7)).cuda()
    dd, dn = torch.from_numpy(desc).cuda(), torch.from_numpy(n).cuda()
    dm = torch.from_numpy(mp.view(np.int64)).cuda()
    out = torch.full((B, cap, 64), 0xAB, dtype=torch.uint8, device="cuda")
    for with_mp in (True, False):
        rc = _ffi.lib().orbfe_mapio_pack_records_device(dk.data_ptr(), dd.data_ptr(), dn.data_ptr(), dm.data_ptr() if with_mp else None,
                                                        B, cap, out.data_ptr(), None)
        assert rc == 0
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        for b in range(B):
            ref = mapio.write_keyframe(0, 0.0, [0, 0, 0], [0, 0, 0, 1], kps[b, :n[b]], desc[b, :n[b]], mp[b, :n[b]] if with_mp else None)[48:]
            assert got[b, :n[b]].tobytes() == ref, (b, with_mp)
            assert not got[b, n[b]:].any()


# ---- pinned to the reference's own writer and reader (perfect/src/Map.cc:143-430, sliced and compiled: oracle/_ref) ----
def _random_map(seed, nmp=60, sizes=(1004, 0, 333, 1, 57)):
    rng = np.random.default_rng(seed)
    mps = [(int(i * 7 + 3), tuple(float(np.float32(v)) for v in rng.normal(0, 5, 3))) for i in range(nmp)]
    kfs = []
    for j, n in enumerate(sizes):
        k, d, _ = _features(seed * 100 + j, n)
        mp = rng.integers(0, max(nmp, 1), n).astype(np.uint64)
        mp[rng.random(n) < (0.3 if nmp else 1.0)] = mapio.ULONG_MAX
        q = rng.normal(0, 1, 4).astype(np.float32)
        kfs.append(dict(id=100 + 3 * j, timestamp=1341846313.592026 + 0.033 * j, t_cw=rng.normal(0, 2, 3).astype(np.float32),
                        q_cw=(q / np.linalg.norm(q)).astype(np.float32), kps=k, desc=d, mp_index=mp,
                        parent=None if j == 0 else 100 + 3 * int(rng.integers(0, j)),
                        connections=[(100 + 3 * int(c), int(rng.integers(15, 400))) for c in rng.permutation(len(sizes))[:int(rng.integers(0, len(sizes)))] if c != j]))
    return mps, kfs


def _ref():
    from oracle import ref_ffi as R
    if not R.available():
        pytest.skip("oracle/_ref not built and /root/reference absent")
    return R


@pytest.mark.parametrize("seed,nmp,sizes", [(1, 60, (1004, 0, 333, 1, 57)), (2, 0, (5,)), (3, 7, ()), (4, 400, (2031, 1990))])
def test_map_file_equals_what_the_reference_writer_writes(tmp_path, seed, nmp, sizes):
    """Map::Save + _WriteMapPoint + _WriteKeyFrame -- the reference's code, compiled -- on a random map vs the product's
    writer (orbfe_mapio_write_keyframe blocks inside mapio.save_map's container): the two files are byte-identical, and
    every keyframe block of the reference's file is what orbfe_mapio_write_keyframe returns."""
    R = _ref()
    mps, kfs = _random_map(seed, nmp, sizes)
    ref_path, our_path = os.path.join(tmp_path, "ref.bin"), os.path.join(tmp_path, "orbfe.bin")
    R.map_save(ref_path, mps, kfs)
    mapio.save_map(our_path, mps, kfs)
    blob = open(ref_path, "rb").read()
    assert blob == open(our_path, "rb").read()
    off = 8 + 20 * nmp + 8
    for kf in kfs:
        blk = mapio.write_keyframe(kf["id"], kf["timestamp"], kf["t_cw"], kf["q_cw"], kf["kps"], kf["desc"], kf["mp_index"])
        assert blob[off:off + len(blk)] == blk and len(blk) == mapio.keyframe_bytes(len(kf["kps"]))
        off += len(blk)


@pytest.mark.parametrize("seed,nmp,sizes", [(11, 60, (1004, 0, 333, 1, 57)), (12, 3, (9,)), (13, 400, (2031, 1990))])
def test_reference_reader_reads_our_file_and_we_read_the_reference_file(tmp_path, seed, nmp, sizes):
    """Map::Load + _ReadMapPoint + _ReadKeyFrame (compiled reference code) on a file the PRODUCT wrote == the map that went in ==
    what the product's reader returns for the file the REFERENCE wrote.  Also what the reference does around the bytes:
    the Frame calls per keyframe in order (SetPose, InitializeScaleLevels, UndistortKeyPoints, AssignFeaturesToGrid,
    ComputeBoW, :167-193), mvuRight / mvDepth = -1 (mono only, :186-187), class_id left at -1, map points gaining one
    observation per referencing feature, ComputeDistinctiveDescriptors + UpdateNormalAndDepth once per map point."""
    R = _ref()
    mps, kfs = _random_map(seed, nmp, sizes)
    ref_path, our_path = os.path.join(tmp_path, "ref.bin"), os.path.join(tmp_path, "orbfe.bin")
    mapio.save_map(our_path, mps, kfs)
    R.map_save(ref_path, mps, kfs)
    rmps, rkfs, info = R.map_load(our_path)
    omps, okfs = mapio.load_map(ref_path)
    f32 = lambda p: tuple(float(np.float32(c)) for c in p)
    assert rmps == [(i, f32(p)) for i, p in mps] == omps
    assert len(rkfs) == len(okfs) == len(kfs)
    # the reference indexes std::set<MapPoint*> (heap-ADDRESS order) with the stored indices (:261): file order under the bump
    # allocator (addresses grow with creation order), which is the contract the format is pinned under
    assert np.array_equal(info["mp_set_rank"], np.arange(nmp))
    for a, r, o in zip(kfs, rkfs, okfs):
        for b in (r, o):
            assert b["id"] == a["id"] and b["timestamp"] == a["timestamp"] and b["parent"] == a["parent"] and b["connections"] == a["connections"]
            assert np.array_equal(b["t_cw"], a["t_cw"]) and np.array_equal(b["q_cw"], a["q_cw"]) and np.array_equal(b["desc"], a["desc"])
            for f in ("x", "y", "size", "angle", "response", "octave"):
                assert np.array_equal(b["kps"][f].view(np.uint32), a["kps"][f].view(np.uint32)), f
            assert (b["kps"]["class_id"] == -1).all()
        want = np.array([-1 if m == mapio.ULONG_MAX else mps[int(m)][0] for m in a["mp_index"]], np.int64)
        assert np.array_equal(r["mp_id"], want) and np.array_equal(o["mp_index"], a["mp_index"])
        assert (r["u_right"] == -1).all() and (r["depth"] == -1).all()
    per_kf = "SetPose;InitializeScaleLevels;UndistortKeyPoints;AssignFeaturesToGrid;ComputeBoW;"
    assert info["log"] == per_kf * len(kfs)
    refs = np.zeros(nmp, np.int64)
    for a in kfs:
        for m in a["mp_index"]:
            if m != mapio.ULONG_MAX:
                refs[int(m)] += 1
    assert np.array_equal(info["mp_nobs"], refs) and (info["mp_calls"] == 1).all()
    assert info["next_mp_id"] == (max(i for i, _ in mps) + 1 if mps else 1)


def test_reference_reader_under_glibc_malloc_links_by_heap_address(tmp_path):
    """Not a property of the format but of the reference's reader: `amp = GetAllMapPoints()` (:261) is a std::set<MapPoint*> in
    heap-address order, and a feature's stored index m becomes amp[m].  Under glibc malloc that order is whatever the heap's
    history makes it; the reader is self-consistent with the writer only when it happens to be creation order.  Here: the
    links the compiled reference reads back are exactly amp[m] in BOTH allocator modes, and equal the file's own order (the
    product's reading) whenever the set order is the file order."""
    R = _ref()
    mps, kfs = _random_map(21, 400, (2031, 777))
    path = os.path.join(tmp_path, "m.bin")
    mapio.save_map(path, mps, kfs)
    for bump in (True, False):
        _, rkfs, info = R.map_load(path, bump=bump)
        rank = info["mp_set_rank"]
        assert sorted(rank.tolist()) == list(range(400))
        at_rank = np.zeros(400, np.int64)
        at_rank[rank] = [i for i, _ in mps]                       # amp[m] = the map point whose set position is m
        for a, r in zip(kfs, rkfs):
            want = np.array([-1 if m == mapio.ULONG_MAX else at_rank[int(m)] for m in a["mp_index"]], np.int64)
            assert np.array_equal(r["mp_id"], want), bump
        if bump:
            assert np.array_equal(rank, np.arange(400))
