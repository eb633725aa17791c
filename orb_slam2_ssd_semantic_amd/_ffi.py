"""ctypes binding of liborbfe.so (the C-ABI declared in include/orbfe.h).

The library is HIP-only: loading works anywhere hipcc's runtime is present, but creating an
extractor / matcher without a GPU fails loudly with ORBFE_ERR_NODEVICE -- there is no fallback.
"""
import ctypes as C
import os

import numpy as np

from . import _build

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28

ORBFE_OK, ORBFE_ERR_ARG, ORBFE_ERR_SIZE, ORBFE_ERR_CAP = 0, -1, -2, -3
ORBFE_ERR_HIP, ORBFE_ERR_NOMEM, ORBFE_ERR_NODEVICE, ORBFE_ERR_STATE = -4, -5, -6, -7
STAGES = ("pyramid", "fast", "octree", "blur", "describe", "total")

# every extern "C" symbol include/orbfe.h declares (tests check the .so exports each of them)
SYMBOLS = (
    "orbfe_version", "orbfe_strerror", "orbfe_last_error", "orbfe_device_count", "orbfe_create", "orbfe_destroy",
    "orbfe_get_scales", "orbfe_get_features_per_level", "orbfe_keypoint_capacity", "orbfe_extract",
    "orbfe_extract_batch", "orbfe_extract_batch_device", "orbfe_get_stream", "orbfe_synchronize", "orbfe_get_level_size",
    "orbfe_get_pyramid_level", "orbfe_tap_blurred_level", "orbfe_tap_candidates", "orbfe_tap_selected",
    "orbfe_set_profiling", "orbfe_get_stage_ms", "orbfe_hamming", "orbfe_matcher_create",
    "orbfe_matcher_destroy", "orbfe_matcher_get_stream", "orbfe_match_bf", "orbfe_match_bf_device", "orbfe_match_bf_frames_device",
    "orbfe_search_by_bow", "orbfe_hamming_csr", "orbfe_assign_grid", "orbfe_features_in_area",
    "orbfe_distinctive_descriptors", "orbfe_distinctive_descriptors_device", "orbfe_stereo_matches", "orbfe_stereo_matches_batch_device", "orbfe_assign_grid_batch_device", "orbfe_features_in_area_device", "orbfe_vocabulary_create", "orbfe_vocabulary_destroy",
    "orbfe_bow_transform", "orbfe_get_overflow", "orbfe_set_fast_mode", "orbfe_get_fast_stats", "orbfe_get_work_counts",
    "orbfe_bow_transform_batch_device", "orbfe_search_by_bow_batch_device", "orbfe_matcher_set_bf_kernel",
    "orbfe_mapio_keyframe_bytes", "orbfe_mapio_write_keyframe", "orbfe_mapio_read_keyframe", "orbfe_mapio_pack_records_device",
    "orbfe_vocfile_load", "orbfe_vocfile_free", "orbfe_vocfile_info", "orbfe_vocfile_arrays", "orbfe_vocfile_save_binary",
    "orbfe_vocabulary_create_from_file", "orbfe_interleaved_to_gray_device", "orbfe_hamming_csr_ex", "orbfe_hamming_csr_device",
    "orbfe_hamming_csr_all", "orbfe_search_by_projection_chi2", "orbfe_window_distances",
    "orbfe_search_by_projection", "orbfe_search_for_triangulation",
    "orbfe_group_shard_range", "orbfe_group_unique_id", "orbfe_group_create_local", "orbfe_group_create_rank", "orbfe_group_destroy",
    "orbfe_group_world", "orbfe_group_capacity", "orbfe_group_frames_padded", "orbfe_group_block_index", "orbfe_group_extract_batch",
    "orbfe_group_extract_shard_device", "orbfe_group_allgather", "orbfe_group_synchronize", "orbfe_group_blocks", "orbfe_group_get_frame",
    "orbfe_group_match", "orbfe_group_match_device", "orbfe_group_owner_rank", "orbfe_group_block_index_of",
    "orbfe_group_create_local_ex", "orbfe_group_members", "orbfe_group_transport", "orbfe_group_get_frame_from", "orbfe_group_get_counts", "orbfe_assign_grid_host", "orbfe_get_pyramid_padded", "orbfe_project_points", "orbfe_proj_queries_local_map", "orbfe_rotation_consistency",
    "orbfe_initialization_resolve", "orbfe_set_option", "orbfe_last_call_reused", "orbfe_matcher_set_projection_kernel", "orbfe_match_bf_blocks_device",
    "orbfe_pipeline_create", "orbfe_pipeline_destroy", "orbfe_pipeline_pipes", "orbfe_pipeline_capacity", "orbfe_pipeline_sub_batch",
    "orbfe_pipeline_extractor", "orbfe_pipeline_matcher", "orbfe_pipeline_extract_match_device", "orbfe_pipeline_join",
    "orbfe_pipeline_synchronize", "orbfe_pipeline_reset_sequence", "orbfe_pipeline_get_overflow", "orbfe_pipeline_extract_match", "orbfe_pipeline_set_host_pipes",
)

# orbfe_set_option (include/orbfe.h ORBFE_OPT_*)
OPTIONS = dict(overlap=1, rows=2, rows_fast=3, rows_blur=4, blur_pieces=5, blur_updown=6, pyr_rows=7, qt_threads_0=8, qt_threads_1=9,
               qt_threads_2=10, debug=11, pyr_fuse=12, fuse_blur_pyr=13, fuse_fast_pyr=14, fuse_fast_pyr_levels=15, blur_rounding=16, reuse_identical_input=17)
PIPE_CONTINUE, PIPE_NO_JOIN = 1, 2


PROJ_QUERY_DTYPE = np.dtype([("u", "<f4"), ("v", "<f4"), ("r", "<f4"), ("min_level", "<i4"), ("max_level", "<i4"),
                             ("ur", "<f4"), ("flags", "<i4"), ("pad", "<i4")])   # orbfe_proj_query
assert PROJ_QUERY_DTYPE.itemsize == 32
PROJ_CLAIMS, PROJ_RIGHT_GATE = 1, 2


class OrbfeParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32), ("max_width", C.c_int32),
                ("max_height", C.c_int32), ("max_batch", C.c_int32), ("device", C.c_int32),
                ("blur_rounding", C.c_int32)]


class OrbfeError(RuntimeError):
    def __init__(self, status, what):
        self.status = status
        super().__init__(f"{what}: status {status}")


_lib = None


def lib():
    """Load (building if needed) liborbfe.so.  torch is imported first when available so that both
    share ONE HIP runtime (same libamdhip64 SONAME) and device pointers can be exchanged."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        import torch  # noqa: F401  (plumbing only: device memory, streams, torch.distributed)
    except Exception:  # pragma: no cover
        pass
    # $ORBFE_LIB: a prebuilt VARIANT of the library (developer A/B runs: kernels compiled with other -D flags, built on the
    # build machine with _build.build_variant so that the GPU box does not spend its minutes compiling)
    path = os.environ.get("ORBFE_LIB") or _build.build()
    _lib = _configure(C.CDLL(path))
    return _lib


def load_variant(path):
    """A second, separately built liborbfe (e.g. ab/liborbfe_dev.so, the -DORBFE_DEVELOPER build) next to the default one in
    the same process: ORBextractor(..., lib=load_variant(path)).  Tests of the developer-only kernel variants use it."""
    lib()   # torch / HIP runtime first, as for the default library
    return _configure(C.CDLL(path))


def _configure(L):
    vp, i32, f32, sz = C.c_void_p, C.c_int32, C.c_float, C.c_size_t
    L.orbfe_version.restype = i32
    L.orbfe_strerror.restype = C.c_char_p
    L.orbfe_strerror.argtypes = [i32]
    L.orbfe_last_error.restype = C.c_char_p
    L.orbfe_device_count.restype = i32
    L.orbfe_create.argtypes = [C.POINTER(OrbfeParams), C.POINTER(vp)]
    L.orbfe_destroy.argtypes = [vp]
    L.orbfe_destroy.restype = None
    L.orbfe_get_scales.argtypes = [vp, vp, vp, vp, vp]
    L.orbfe_get_features_per_level.argtypes = [vp, vp]
    L.orbfe_keypoint_capacity.argtypes = [vp]
    L.orbfe_extract.argtypes = [vp, vp, i32, i32, i32, vp, vp, i32, vp]
    L.orbfe_extract_batch.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, i32, vp]
    L.orbfe_extract_batch_device.argtypes = [vp, vp, i32, i32, i32, i32, sz, vp, vp, i32, vp, vp]
    L.orbfe_synchronize.argtypes = [vp]
    L.orbfe_get_stream.argtypes = [vp]
    L.orbfe_get_stream.restype = vp
    L.orbfe_matcher_get_stream.argtypes = [vp]
    L.orbfe_matcher_get_stream.restype = vp
    L.orbfe_get_level_size.argtypes = [vp, i32, vp, vp]
    L.orbfe_get_pyramid_level.argtypes = [vp, i32, i32, vp, i32, i32]
    L.orbfe_get_pyramid_padded.argtypes = [vp, i32, vp, sz, vp, vp]
    L.orbfe_tap_blurred_level.argtypes = [vp, i32, i32, vp, i32]
    L.orbfe_tap_candidates.argtypes = [vp, i32, i32, vp, i32, vp]
    L.orbfe_tap_selected.argtypes = [vp, i32, i32, vp, i32, vp]
    L.orbfe_get_work_counts.argtypes = [vp, vp]
    L.orbfe_matcher_set_bf_kernel.argtypes = [vp, i32]
    L.orbfe_matcher_set_projection_kernel.argtypes = [vp, i32]
    L.orbfe_mapio_keyframe_bytes.restype = sz
    L.orbfe_mapio_keyframe_bytes.argtypes = [i32]
    L.orbfe_mapio_write_keyframe.argtypes = [vp, sz, C.c_uint64, C.c_double, vp, vp, vp, vp, vp, i32, vp]
    L.orbfe_mapio_read_keyframe.argtypes = [vp, sz, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp]
    L.orbfe_mapio_pack_records_device.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp]
    L.orbfe_vocfile_load.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.orbfe_vocfile_free.argtypes = [vp]
    L.orbfe_vocfile_free.restype = None
    L.orbfe_vocfile_info.argtypes = [vp] + [vp] * 6
    L.orbfe_vocfile_arrays.argtypes = [vp] + [vp] * 7
    L.orbfe_vocfile_save_binary.argtypes = [vp, C.c_char_p]
    L.orbfe_vocabulary_create_from_file.argtypes = [i32, vp, C.POINTER(vp)]
    L.orbfe_interleaved_to_gray_device.argtypes = [vp, i32, i32, i32, i32, sz, i32, vp, i32, sz, vp]
    L.orbfe_bow_transform_batch_device.argtypes = [vp, vp, vp, vp, i32, i32, i32] + [vp] * 10
    L.orbfe_search_by_bow_batch_device.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, f32, i32, i32, i32, vp, vp, vp]
    L.orbfe_get_overflow.argtypes = [vp, vp]
    L.orbfe_set_fast_mode.argtypes = [vp, i32, i32]
    L.orbfe_get_fast_stats.argtypes = [vp, vp, i32]
    L.orbfe_set_profiling.argtypes = [vp, i32]
    L.orbfe_get_stage_ms.argtypes = [vp, vp]
    L.orbfe_hamming.argtypes = [vp, vp]
    L.orbfe_matcher_create.argtypes = [i32, C.POINTER(vp)]
    L.orbfe_matcher_destroy.argtypes = [vp]
    L.orbfe_matcher_destroy.restype = None
    L.orbfe_match_bf.argtypes = [vp, vp, i32, vp, i32, vp, vp, f32, i32, i32, vp, vp, vp, vp]
    L.orbfe_match_bf_device.argtypes = [vp, vp, i32, vp, i32, vp, vp, f32, i32, i32, vp, vp, vp, vp, vp]
    L.orbfe_match_bf_frames_device.argtypes = [vp, vp, vp, vp, i32, vp, vp, i32, f32, i32, i32, vp, vp, vp]
    L.orbfe_search_by_bow.argtypes = ([vp] + [vp, i32, vp, vp, vp, vp, vp, i32] * 2 + [f32, i32, i32, i32, vp, vp])
    L.orbfe_hamming_csr.argtypes = [vp, vp, i32, vp, i32, vp, vp, vp, vp, vp]
    L.orbfe_hamming_csr_ex.argtypes = [vp, vp, i32, vp, i32, vp, vp, vp, vp, vp, vp]
    L.orbfe_hamming_csr_all.argtypes = [vp, vp, i32, vp, i32, vp, vp, vp]
    L.orbfe_hamming_csr_device.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    L.orbfe_project_points.argtypes = [vp] * 5 + [f32] * 9 + [i32, i32] + [vp] * 10
    L.orbfe_proj_queries_local_map.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, f32, vp, vp, vp]
    L.orbfe_rotation_consistency.argtypes = [vp, vp, i32, i32, vp]
    L.orbfe_initialization_resolve.argtypes = [vp, vp, i32, i32, i32, f32, vp, vp]
    L.orbfe_assign_grid_host.argtypes = [vp, i32, f32, f32, f32, f32, vp, vp, vp]
    L.orbfe_assign_grid.argtypes = [vp, vp, i32, f32, f32, f32, f32, vp, vp, vp]
    L.orbfe_vocabulary_create.argtypes = [i32, i32, vp, vp, vp, vp, vp, i32, vp]
    L.orbfe_vocabulary_destroy.argtypes = [vp]
    L.orbfe_vocabulary_destroy.restype = None
    L.orbfe_bow_transform.argtypes = [vp, vp, vp, i32, i32] + [vp] * 10
    L.orbfe_stereo_matches.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp, i32, f32, f32, vp, vp]
    L.orbfe_distinctive_descriptors_device.argtypes = [vp, vp, vp, vp, i32, i32, vp, vp, vp]
    L.orbfe_features_in_area_device.argtypes = [vp, vp, vp, vp, f32, f32, f32, f32, vp, vp, i32, vp, vp, i32, vp]
    L.orbfe_assign_grid_batch_device.argtypes = [vp, vp, vp, i32, i32, f32, f32, f32, f32, vp, vp, vp, vp]
    L.orbfe_stereo_matches_batch_device.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, f32, vp, vp, vp]
    L.orbfe_distinctive_descriptors.argtypes = [vp, vp, i32, vp, vp, i32, vp, vp]
    L.orbfe_features_in_area.argtypes = [vp, vp, vp, i32, vp, vp, f32, f32, f32, f32, vp, vp, i32, vp, vp, i32]
    L.orbfe_search_by_projection.argtypes = [vp, vp, vp, vp, i32, vp, vp, f32, f32, f32, f32, vp, vp, vp, vp, i32, i32, f32, i32, vp, vp, vp]
    L.orbfe_search_by_projection_chi2.argtypes = [vp, vp, vp, vp, i32, vp, vp, f32, f32, f32, f32, vp, vp, vp, i32, vp, vp, i32, i32, f32, i32, vp, vp, vp]
    L.orbfe_window_distances.argtypes = [vp, vp, vp, vp, i32, vp, vp, f32, f32, f32, f32, vp, vp, i32, vp, vp, i32]
    L.orbfe_search_for_triangulation.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp, i32, vp, vp, vp, i32, vp, f32, f32, vp,
                                                 vp, i32, i32, vp]
    L.orbfe_group_shard_range.argtypes = [i32, i32, i32, vp, vp]
    L.orbfe_group_shard_range.restype = None
    L.orbfe_group_unique_id.argtypes = [vp]
    L.orbfe_group_create_local.argtypes = [C.POINTER(OrbfeParams), vp, i32, C.POINTER(vp)]
    L.orbfe_group_create_rank.argtypes = [C.POINTER(OrbfeParams), i32, i32, i32, vp, C.POINTER(vp)]
    L.orbfe_group_destroy.argtypes = [vp]
    L.orbfe_group_destroy.restype = None
    for nm in ("orbfe_group_world", "orbfe_group_capacity", "orbfe_group_frames_padded"):
        getattr(L, nm).argtypes = [vp]
    L.orbfe_group_block_index.argtypes = [vp, i32, i32]
    L.orbfe_group_extract_batch.argtypes = [vp, vp, i32, i32, i32, i32]
    L.orbfe_group_extract_shard_device.argtypes = [vp, i32, vp, i32, i32, i32, i32, sz]
    L.orbfe_group_allgather.argtypes = [vp]
    L.orbfe_group_synchronize.argtypes = [vp]
    L.orbfe_group_blocks.argtypes = [vp, i32, vp, vp, vp, vp]
    L.orbfe_group_get_frame.argtypes = [vp, i32, vp, vp, i32, vp]
    L.orbfe_group_owner_rank.argtypes = [i32, i32, i32]
    L.orbfe_group_block_index_of.argtypes = [i32, i32, i32, i32]
    L.orbfe_group_create_local_ex.argtypes = [C.POINTER(OrbfeParams), vp, i32, i32, C.POINTER(vp)]
    L.orbfe_group_members.argtypes = [vp]
    L.orbfe_group_transport.argtypes = [vp]
    L.orbfe_group_get_frame_from.argtypes = [vp, i32, i32, vp, vp, i32, vp]
    L.orbfe_group_get_counts.argtypes = [vp, i32, vp]
    L.orbfe_group_match.argtypes = [vp, vp, vp, i32, f32, i32, i32, vp, vp]
    L.orbfe_group_match_device.argtypes = [vp, i32, vp, vp, i32, f32, i32, i32, vp, vp]
    L.orbfe_set_option.argtypes = [vp, i32, i32]
    L.orbfe_last_call_reused.argtypes = [vp]
    L.orbfe_last_call_reused.restype = C.c_int32
    L.orbfe_match_bf_blocks_device.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, i32, f32, i32, i32, vp, vp, vp]
    L.orbfe_pipeline_create.argtypes = [C.POINTER(OrbfeParams), i32, C.POINTER(vp)]
    L.orbfe_pipeline_destroy.argtypes = [vp]
    L.orbfe_pipeline_destroy.restype = None
    for nm in ("orbfe_pipeline_pipes", "orbfe_pipeline_capacity", "orbfe_pipeline_sub_batch", "orbfe_pipeline_synchronize",
               "orbfe_pipeline_reset_sequence"):
        getattr(L, nm).argtypes = [vp]
    L.orbfe_pipeline_extractor.argtypes = [vp, i32]
    L.orbfe_pipeline_extractor.restype = vp
    L.orbfe_pipeline_matcher.argtypes = [vp, i32]
    L.orbfe_pipeline_matcher.restype = vp
    L.orbfe_pipeline_extract_match_device.argtypes = [vp, vp, i32, i32, i32, i32, sz, vp, vp, i32, vp, vp, vp, f32, i32, i32, i32, vp]
    L.orbfe_pipeline_join.argtypes = [vp, vp]
    L.orbfe_pipeline_set_host_pipes.argtypes = [vp, i32]
    L.orbfe_pipeline_extract_match.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp, i32, vp, vp, vp, f32, i32, i32, i32]
    L.orbfe_pipeline_get_overflow.argtypes = [vp, vp]
    for name in SYMBOLS:
        f = getattr(L, name)
        if f.restype is C.c_int:  # default -> orbfe_status / int32
            f.restype = i32
    return L


def library_path():
    return _build.LIB


def last_error():
    return lib().orbfe_last_error().decode("utf-8", "replace")


def check(status, what):
    if status != ORBFE_OK:
        raise OrbfeError(status, f"{what}: {lib().orbfe_strerror(status).decode()} ({last_error()})")


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)
