"""The on-disk map format of the reference's Map::Save / Map::Load (perfect/src/Map.cc:228-430) over the C-ABI.

File = u64 nb_mappoints | nb x {u64 id, f32 x, y, z}                                   (:393-395, _WriteMapPoint :320-327)
       u64 nb_keyframes | nb x keyframe block (orbfe_mapio_write_keyframe, include/orbfe.h)   (:409-411, :330-381)
       per keyframe: u64 parent id (ULONG_MAX = none) | u64 nb_con | nb_con x {u64 id, i32 weight}   (:413-428)
The keyframe blocks -- where the extractor's keypoints and descriptors go -- are produced / parsed by liborbfe.so; the
few container fields around them are plain struct packing here.  Geometry (poses, covisibility) is the caller's.
"""
import ctypes as C
import struct

import numpy as np

from . import _ffi
from ._ffi import KP_DTYPE, check, ptr

ULONG_MAX = 0xFFFFFFFFFFFFFFFF


def keyframe_bytes(n):
    return int(_ffi.lib().orbfe_mapio_keyframe_bytes(int(n)))


def write_keyframe(kf_id, timestamp, t_cw, q_cw, kps, desc, mp_index=None):
    """bytes of one _WriteKeyFrame block.  kps: KP_DTYPE[n], desc: u8[n,32], mp_index: u64[n] or None (no map points)"""
    kps = np.ascontiguousarray(kps, KP_DTYPE)
    desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    n = len(kps)
    assert len(desc) == n
    t = np.ascontiguousarray(t_cw, np.float32)
    q = np.ascontiguousarray(q_cw, np.float32)
    mp = None if mp_index is None else np.ascontiguousarray(mp_index, np.uint64)
    out = np.zeros(keyframe_bytes(n), np.uint8)
    w = C.c_size_t(0)
    check(_ffi.lib().orbfe_mapio_write_keyframe(ptr(out), out.size, int(kf_id), float(timestamp), ptr(t), ptr(q), ptr(kps),
                                                ptr(desc), ptr(mp), n, C.byref(w)), "orbfe_mapio_write_keyframe")
    assert w.value == out.size
    return out.tobytes()


def read_keyframe(buf, offset=0):
    """-> (dict(id, timestamp, t_cw, q_cw, kps, desc, mp_index), bytes consumed)"""
    a = np.frombuffer(buf, np.uint8, offset=offset)
    L = _ffi.lib()
    n = C.c_int32(0)
    used = C.c_size_t(0)
    st = L.orbfe_mapio_read_keyframe(ptr(a), a.size, None, None, None, None, None, None, None, 0, C.byref(n), C.byref(used))
    if st not in (_ffi.ORBFE_OK, _ffi.ORBFE_ERR_CAP):
        check(st, "orbfe_mapio_read_keyframe")
    m = max(n.value, 1)
    kps, desc, mp = np.zeros(m, KP_DTYPE), np.zeros((m, 32), np.uint8), np.zeros(m, np.uint64)
    kid, ts = C.c_uint64(0), C.c_double(0)
    t, q = np.zeros(3, np.float32), np.zeros(4, np.float32)
    check(L.orbfe_mapio_read_keyframe(ptr(a), a.size, C.byref(kid), C.byref(ts), ptr(t), ptr(q), ptr(kps), ptr(desc), ptr(mp), m,
                                      C.byref(n), C.byref(used)), "orbfe_mapio_read_keyframe")
    return dict(id=kid.value, timestamp=ts.value, t_cw=t, q_cw=q, kps=kps[:n.value].copy(), desc=desc[:n.value].copy(),
                mp_index=mp[:n.value].copy()), used.value


def save_map(path, mappoints, keyframes):
    """Map::Save (:385-430).  mappoints: iterable of (id, (x, y, z)); keyframes: list of dicts with the read_keyframe fields
    plus optional parent (id or None) and connections [(id, weight), ...]."""
    with open(path, "wb") as f:
        mappoints = list(mappoints)
        f.write(struct.pack("<Q", len(mappoints)))
        for mid, (x, y, z) in mappoints:
            f.write(struct.pack("<Qfff", mid, x, y, z))
        f.write(struct.pack("<Q", len(keyframes)))
        for kf in keyframes:
            f.write(write_keyframe(kf["id"], kf["timestamp"], kf["t_cw"], kf["q_cw"], kf["kps"], kf["desc"], kf.get("mp_index")))
        for kf in keyframes:
            parent = kf.get("parent")
            con = kf.get("connections", [])
            f.write(struct.pack("<QQ", ULONG_MAX if parent is None else parent, len(con)))
            for cid, wgt in con:
                f.write(struct.pack("<Qi", cid, wgt))


def load_map(path):
    """Map::Load (:228-300) -> (mappoints [(id, (x, y, z))], keyframes [dict])"""
    buf = open(path, "rb").read()
    off = 0
    (nmp,) = struct.unpack_from("<Q", buf, off)
    off += 8
    mps = []
    for _ in range(nmp):
        mid, x, y, z = struct.unpack_from("<Qfff", buf, off)
        off += 20
        mps.append((mid, (x, y, z)))
    (nkf,) = struct.unpack_from("<Q", buf, off)
    off += 8
    kfs = []
    for _ in range(nkf):
        kf, used = read_keyframe(buf, off)
        off += used
        kfs.append(kf)
    for kf in kfs:
        parent, ncon = struct.unpack_from("<QQ", buf, off)
        off += 16
        kf["parent"] = None if parent == ULONG_MAX else parent
        kf["connections"] = []
        for _ in range(ncon):
            cid, wgt = struct.unpack_from("<Qi", buf, off)
            off += 12
            kf["connections"].append((cid, wgt))
    assert off == len(buf), "trailing bytes in map file"
    return mps, kfs


class VocabularyFile:
    """ORBvoc.txt / ORBvoc.bin parsed by liborbfe.so (host only).  `arrays()` gives what ORBVocabulary(...) takes;
    `save_binary(path)` is tool/text2binary.cc's conversion."""

    def __init__(self, path):
        self._L = _ffi.lib()
        self._v = C.c_void_p()
        check(self._L.orbfe_vocfile_load(str(path).encode(), C.byref(self._v)), "orbfe_vocfile_load")
        v = [C.c_int32() for _ in range(6)]
        check(self._L.orbfe_vocfile_info(self._v, *[C.byref(x) for x in v]), "orbfe_vocfile_info")
        self.k, self.L, self.nnodes, self.nwords, self.scoring, self.weighting = (x.value for x in v)

    def close(self):
        if getattr(self, "_v", None):
            self._L.orbfe_vocfile_free(self._v)
            self._v = None

    def __del__(self):
        self.close()

    @property
    def handle(self):
        return self._v

    def arrays(self):
        p = [C.c_void_p() for _ in range(7)]
        check(self._L.orbfe_vocfile_arrays(self._v, *[C.byref(x) for x in p]), "orbfe_vocfile_arrays")
        nn = self.nnodes

        def arr(pp, dt, n):
            if n == 0:
                return np.zeros(0, dt)
            return np.frombuffer((C.c_uint8 * (n * np.dtype(dt).itemsize)).from_address(pp.value), dt).copy()
        return dict(child_off=arr(p[0], np.uint32, nn + 1), child_idx=arr(p[1], np.uint32, nn - 1),
                    node_desc=arr(p[2], np.uint8, nn * 32).reshape(nn, 32), word_id=arr(p[3], np.uint32, nn),
                    weight=arr(p[4], np.float64, nn), L=self.L), dict(parent=arr(p[5], np.uint32, nn), is_leaf=arr(p[6], np.uint8, nn))

    def save_binary(self, path):
        check(self._L.orbfe_vocfile_save_binary(self._v, str(path).encode()), "orbfe_vocfile_save_binary")

    def to_device(self, matcher, device=-1):
        """ORBVocabulary on the device straight from the parsed file"""
        from .matcher import ORBVocabulary
        voc, _ = self.arrays()
        return ORBVocabulary(matcher, device=device, **voc)
