"""The C header is valid C99, the C++ adapter compiles and links against the library,
and its host-only getters work (no compute without a GPU)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_is_plain_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "orbslamm_hip.h"\nint main(void){OrbxKeyPoint k; return sizeof(k)==28?0:1;}\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-fsyntax-only", str(src)])


def test_cpp_adapter_builds_and_runs(tmp_path):
    from orbslamm_amd import _lib
    _lib.lib()
    exe = str(tmp_path / "adapter")
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "adapter_compile.cpp"), "-o", exe,
                           "-L", os.path.join(ROOT, "orbslamm_amd"), "-lorbslamm_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "orbslamm_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "adapter ok" in out.stdout and "no CPU fallback" in out.stderr


def test_orbmatcher_dropin_template_instantiates(tmp_path):
    """ORBmatcherT<Frame, KeyFrame, MapPoint> with all eleven reference signatures compiles against mock types that
    carry the reference's member names (no OpenCV, no GPU needed to build; the run is a -m gpu test)"""
    from oracle import binding as ob
    ob.build()
    exe = str(tmp_path / "dropin")
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "matcher_dropin_gpu.cpp"), "-o", exe,
                           "-L", os.path.join(ROOT, "orbslamm_amd"), "-lorbslamm_hip", "-L", os.path.join(ROOT, "oracle"), "-lorb_oracle",
                           "-Wl,-rpath," + os.path.join(ROOT, "orbslamm_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
                           "-Wl,-rpath,/opt/rocm/lib"])
    syms = subprocess.check_output(["nm", "-C", exe]).decode()
    for member in ("SearchByProjection", "SearchByBoW", "SearchForInitialization", "SearchForTriangulation", "SearchBySim3", "Fuse"):
        assert "ORBmatcherT<mock::Frame, mock::KeyFrame, mock::MapPoint>::" + member in syms, member


def test_extractor_dropin_reference_signature_compiles(tmp_path):
    """the -DORBSLAMM_WITH_OPENCV branch of include/ORBextractor_hip.hpp -- the reference's operator()(InputArray,
    InputArray, vector<KeyPoint>&, OutputArray) -- builds against the data-holder cv:: types of tests/cpp/mock_opencv
    (no OpenCV in this image; the run is a -m gpu test)"""
    from oracle import binding as ob
    ob.build()
    exe = str(tmp_path / "adapter_cv")
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-DORBSLAMM_WITH_OPENCV", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "tests", "cpp", "mock_opencv"),
                           os.path.join(ROOT, "tests", "cpp", "adapter_cv_gpu.cpp"), "-o", exe,
                           "-L", os.path.join(ROOT, "orbslamm_amd"), "-lorbslamm_hip", "-L", os.path.join(ROOT, "oracle"), "-lorb_oracle",
                           "-Wl,-rpath," + os.path.join(ROOT, "orbslamm_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
                           "-Wl,-rpath,/opt/rocm/lib"])
    syms = subprocess.check_output(["nm", "-C", exe]).decode()
    assert "iORB_SLAM::ORBextractor::operator()(cv::_InputArray const&, cv::_InputArray const&, std::vector<cv::KeyPoint" in syms


def test_check_vs_opencv_compiles(tmp_path):
    """tools/check_vs_opencv/check_vs_opencv.cpp -- the program that pins the oracle against a real OpenCV + the
    reference's unmodified ORBextractor.cc wherever OpenCV exists -- at least parses and type-checks here: against
    declaration-only cv:: headers (tests/cpp/mock_opencv) and, when the reference checkout is present, its own
    include/ORBextractor.h.  Nothing is built or run (there is no OpenCV in this image); the exporter of the frames it
    reads is run for real."""
    import pytest
    ref_inc = "/root/reference/SingleRobotScenario/include"
    if not os.path.exists(os.path.join(ref_inc, "ORBextractor.h")):
        pytest.skip("no reference checkout: the program includes the reference's own header")
    subprocess.check_call(["g++", "-std=c++11", "-fsyntax-only", "-Wall", "-I", os.path.join(ROOT, "tests", "cpp", "mock_opencv"), "-I", ref_inc,
                           "-I", os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools", "check_vs_opencv", "check_vs_opencv.cpp")])
    subprocess.check_call(["python3", os.path.join(ROOT, "tools", "check_vs_opencv", "export_pgm.py"), str(tmp_path / "frames")])
    names = os.listdir(tmp_path / "frames")
    assert len(names) == 19 and all(n.endswith(".pgm") for n in names)
    head = open(tmp_path / "frames" / "synth_1241x376_0.pgm", "rb").read(16)
    assert head.startswith(b"P5\n1241 376\n255\n")


def test_camera_hub_host_logic_under_thread_sanitizer(tmp_path):
    """include/orbslamm_hub.hpp (the robots' threads sharing one chain) against a mock of the C ABI entries it calls, built with
    -fsanitize=thread: one thread at a time inside the library, every ticket released once and only after its readers copied,
    orbm_track_frames pairs = (frame s, frame s - 1) of the same camera in the slot ring, every thread gets ITS rows --
    lockstep (long wait), no wait, one slow camera, extraction only (tests/cpp/hub_mock_cpu.cpp)"""
    exe = str(tmp_path / "hub_mock")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-g", "-fsanitize=thread", "-pthread", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "hub_mock_cpu.cpp"), "-o", exe])
    out = subprocess.run([exe, "400"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "hub_mock ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
    assert "ThreadSanitizer" not in out.stderr, out.stderr[-4000:]
