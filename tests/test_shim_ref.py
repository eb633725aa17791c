"""The product's link-level matcher shim (orb_slam2_ssd_semantic_amd/shim/ORBmatcher_orbfe.cc) inside the reference's own
class: compiled against the UNMODIFIED /root/reference/include/ORBmatcher.h and linked with the reference's own
ORBmatcher.cc for every other member (oracle/_ref/libshim_ref.so, recipe oracle/refbuild/Makefile).  CPU: it compiles,
links, loads, DescriptorDistance agrees with the reference's.  GPU: both SearchByBoW overloads called through the
reference's class interface return exactly what the reference's compiled bodies return on the same mock objects."""
import numpy as np
import pytest

from oracle import ref_ffi as R
from test_ref_pin import _bow_case

pytestmark = pytest.mark.skipif(not R.shim_available(), reason="oracle/_ref/libshim_ref.so not built and /root/reference absent")


def test_shim_links_against_the_unmodified_reference_header():
    L = R.shim_lib()
    for name in ("shim_descriptor_distance", "shim_search_by_bow_kf_f", "shim_search_by_bow_kf_kf", "shim_three_maxima"):
        assert getattr(L, name)
    rng = np.random.default_rng(0)
    d = rng.integers(0, 256, (200, 32), dtype=np.uint8)
    for i in range(0, 200, 2):   # host-side entry point: no GPU needed
        assert R.descriptor_distance(d[i], d[i + 1], shim=True) == R.descriptor_distance(d[i], d[i + 1]) == int(np.unpackbits(d[i] ^ d[i + 1]).sum())


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(4))
def test_shim_search_by_bow_equals_reference_bodies(seed):
    rng = np.random.default_rng(300 + seed)
    total = 0
    for it in range(25):
        n1, n2 = int(rng.choice([0, 1, 7, 150, 1000])), int(rng.choice([0, 1, 9, 180, 1000]))
        (d1, v1, a1, fv1), (d2, v2, a2, fv2) = _bow_case(rng, n1, n2, int(rng.choice([1, 4, 30, 120])), 0.8, it % 2)
        nnratio = float(rng.choice([0.6, 0.7, 0.75, 0.9]))
        ori = bool(it % 3)
        rm, rn = R.search_by_bow_kf_f(d1, v1, a1, fv1, d2, a2, fv2, nnratio, ori)
        sm, sn = R.search_by_bow_kf_f(d1, v1, a1, fv1, d2, a2, fv2, nnratio, ori, shim=True)
        assert sn == rn and np.array_equal(sm, rm), ("KF,F", seed, it, n1, n2)
        r12, rn2 = R.search_by_bow_kf_kf(d1, v1, a1, fv1, d2, v2, a2, fv2, nnratio, ori)
        s12, sn2 = R.search_by_bow_kf_kf(d1, v1, a1, fv1, d2, v2, a2, fv2, nnratio, ori, shim=True)
        assert sn2 == rn2 and np.array_equal(s12, r12), ("KF,KF", seed, it, n1, n2)
        total += rn + rn2
    assert total > 50
