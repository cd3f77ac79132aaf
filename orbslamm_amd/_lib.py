"""ctypes loader for liborbslamm_hip.so (the C ABI of include/orbslamm_hip.h).

The library is the product: there is no Python/CPU compute path behind it.  If
the shared object is missing this module raises -- loudly -- instead of falling
back to anything."""
import ctypes as C
import os
import subprocess

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
SO_PATH = os.path.join(_PKG, "liborbslamm_hip.so")
SRC = os.path.join(_PKG, "csrc", "orbslamm_hip.hip")
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]

ORBX_MAX_LEVELS = 16
ORBX_PROF_MAX = 16
ORBX_OK, ORBX_E_INVALID, ORBX_E_NO_DEVICE, ORBX_E_HIP, ORBX_E_CAPACITY, ORBX_E_UNSUPPORTED = 0, -1, -2, -3, -4, -5

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28


class OrbxParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scaleFactor", C.c_float), ("nlevels", C.c_int32),
                ("iniThFAST", C.c_int32), ("minThFAST", C.c_int32)]


class OrbxProfile(C.Structure):
    _fields_ = [("n", C.c_int32), ("name", C.c_char_p * ORBX_PROF_MAX), ("ms", C.c_double * ORBX_PROF_MAX),
                ("launches", C.c_int64 * ORBX_PROF_MAX)]


class OrbxStreamOpts(C.Structure):
    _fields_ = [("match_prev", C.c_int32), ("nnratio", C.c_float), ("th_low", C.c_int32), ("check_ori", C.c_int32)]


class OrbxBatchView(C.Structure):
    _fields_ = [("B", C.c_int32), ("cap", C.c_int32), ("n", C.c_void_p), ("kps", C.c_void_p), ("desc", C.c_void_p),
                ("match", C.c_void_p), ("nmatch", C.c_void_p)]


class OrbxBatchOut(C.Structure):
    _fields_ = [("kps", C.c_void_p), ("desc", C.c_void_p), ("n", C.c_void_p), ("match", C.c_void_p), ("nmatch", C.c_void_p), ("cap", C.c_int32)]


class OrbmFeatVec(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("node_id", C.c_void_p), ("start", C.c_void_p), ("idx", C.c_void_p)]


class OrbmGrid(C.Structure):
    _fields_ = [("minX", C.c_float), ("minY", C.c_float), ("invW", C.c_float), ("invH", C.c_float),
                ("cols", C.c_int32), ("rows", C.c_int32)]


class OrbmProjParams(C.Structure):
    _fields_ = [("mode", C.c_int32), ("nnratio", C.c_float), ("check_ori", C.c_int32), ("th_dist", C.c_int32)]


# every symbol include/orbslamm_hip.h declares (tests check that all of them resolve)
EXPORTS = [
    "orbx_last_error", "orbx_device_count", "orbx_device_pci_bus_id", "orbx_device_shader_clock_mhz", "orbx_create", "orbx_create_live", "orbx_destroy", "orbx_levels", "orbx_scale_factor",
    "orbx_scale_tables", "orbx_features_per_level", "orbx_umax", "orbx_max_keypoints", "orbx_extract",
    "orbx_extract_batch", "orbx_submit_batch", "orbx_submit_batch_into", "orbx_collect", "orbx_host_alloc", "orbx_collect_view", "orbx_release", "orbx_collect_batch", "orbx_extract_match_batch",
    "orbx_host_alloc_frames", "orbx_host_free", "orbx_host_register", "orbx_host_unregister", "orbx_extract_batch_device", "orbx_device_results", "orbx_download",
    "orbx_pyramid_level", "orbx_compute_stereo_matches", "orbx_level_candidates", "orbx_sync", "orbx_device_alloc", "orbx_device_free", "orbx_upload", "orbx_match_prev_batch_device",
    "orbx_device_matches", "orbx_download_matches", "orbx_reset_stream", "orbx_set_serial", "orbx_profile_enable",
    "orbx_profile_read", "orbx_debug_pair_overlap", "orbx_debug_link_rate", "orbx_debug_stage_rows", "orbx_profile_select", "orbm_create", "orbm_destroy", "orbm_thread_handle", "orbm_alloc_stats", "orbm_distance_matrix", "orbm_match_bruteforce",
    "orbm_search_by_bow", "orbm_search_by_projection", "orbm_search_by_projection_stereo", "orbm_projection_prepare", "orbm_features_in_area", "orbm_window_best",
    "orbm_search_for_initialization", "orbm_search_for_triangulation", "orbm_distinctive_descriptors", "orbm_descriptors_to_text", "orbm_descriptors_from_text", "orbm_undistort_keypoints", "orbm_compute_stereo_from_rgbd", "orbm_frame_create", "orbm_frame_destroy", "orbm_frame_size", "orbm_frame_settle",
    "orbm_frame_download_keys_un", "orbm_search_by_projection_frame", "orbm_frame_compute_bow", "orbm_search_by_bow_frames", "orbm_search_for_initialization_frames", "orbm_window_best_frame", "orbm_search_for_triangulation_frames",
    "orbv_create", "orbv_load_text", "orbv_destroy", "orbv_transform",
    "orbm_last_search_stats", "orbm_frameset_create", "orbm_frameset_destroy", "orbm_frameset_build", "orbm_frameset_build_from_extractor",
    "orbm_frameset_sync", "orbm_frameset_attach", "orbm_frameset_download", "orbm_track_frames", "orbm_track_local_points", "orbm_track_frame_projected", "orbm_track_results", "orbm_track_stats",
    "orbm_frameset_compute_bow", "orbm_frameset_bow_vector", "orbm_bow_frames", "orbm_bow_results",
]


def build(force=False):
    """hipcc the extension in-tree for gfx950 (cross-compiles without a GPU)."""
    deps = [os.path.join(_PKG, "csrc", f) for f in os.listdir(os.path.join(_PKG, "csrc"))]
    deps.append(os.path.join(_ROOT, "include", "orbslamm_hip.h"))
    if not force and os.path.exists(SO_PATH) and all(os.path.getmtime(SO_PATH) >= os.path.getmtime(d) for d in deps):
        return SO_PATH
    cmd = ["hipcc"] + HIPCC_FLAGS + ["-o", SO_PATH, SRC]
    subprocess.check_call(cmd)
    return SO_PATH


_lib = None


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch wheels bundle their own libamdhip64.so.7; if
    this library pulled in /opt/rocm's copy first, a later `import torch` in the same
    process would find the GPU already owned by the other runtime ("No HIP GPUs are
    available").  So when torch is installed, its runtime is loaded (not torch itself)
    before liborbslamm_hip.so, whose NEEDED libamdhip64.so.7 then resolves to it."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError("liborbslamm_hip.so is missing (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                               "there is no CPU fallback for the ORB front-end")
        _preload_hip_runtime()
        # ORBSLAMM_HIP_LIB: load another build of the same library (A/B timing of kernel variants, tools/ab_bench.sh)
        L = C.CDLL(os.environ.get("ORBSLAMM_HIP_LIB") or SO_PATH)
        L.orbx_last_error.restype = C.c_char_p
        L.orbx_scale_factor.restype = C.c_float
        for name in EXPORTS:
            getattr(L, name)
        L.orbx_extract_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t]
        _lib = L
    return _lib


class OrbError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("orbslamm_hip error %d: %s" % (code, msg))
        self.code = code


def check(rc):
    if rc != 0:
        raise OrbError(rc, lib().orbx_last_error().decode(errors="replace"))


def ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def device_pci_bus_id(device):
    """"0000:c1:00.0" or None"""
    buf = C.create_string_buffer(32)
    try:
        return buf.value.decode().lower() if lib().orbx_device_pci_bus_id(int(device), buf, 32) == 0 else None
    except Exception:
        return None


def shader_clock_mhz(device):
    """the shader clock the device runs at now (orbx_device_shader_clock_mhz), or None"""
    v = C.c_float(0)
    try:
        return float(v.value) if lib().orbx_device_shader_clock_mhz(int(device), C.byref(v)) == 0 else None
    except Exception:
        return None


def numa_cpus_of_device(device):
    """(numa node, cpu ids) next to a HIP device, from sysfs; (None, None) where the platform does not say"""
    buf = C.create_string_buffer(32)
    try:
        if lib().orbx_device_pci_bus_id(int(device), buf, 32) != 0:
            return None, None
        bus = buf.value.decode().lower()
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bus).read())
        if node < 0:
            return None, None
        cpus = []
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.extend(range(int(a), int(b or a) + 1))
        return node, cpus
    except (OSError, ValueError):
        return None, None
