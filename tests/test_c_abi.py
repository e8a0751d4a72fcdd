"""The C-ABI libraries load, export every function include/jxl_tiny_amd.h
declares, and fail loudly (no CPU fallback) when no HIP device is present."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

import jxlt_testlib as T


def _declared_functions(header="jxl_tiny_amd.h"):
    text = (T.ROOT / "include" / header).read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(jxlt_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(built):
    """Both headers: the drop-in surface (jxl_tiny_amd.h) and the test / profiling hooks kept apart from it
    (jxl_tiny_amd_testing.h)."""
    declared = _declared_functions()
    testing = _declared_functions("jxl_tiny_amd_testing.h")
    assert len(declared) >= 16 and not set(declared) & set(testing)
    hip, host = built.hip_lib(), built.host_lib()
    for name in declared + testing:
        assert hasattr(hip, name) or hasattr(host, name), name
    assert sorted(built.HIP_SYMBOLS + built.HOST_SYMBOLS) == declared
    assert sorted(built.HIP_SYMBOLS_TESTING + built.HOST_SYMBOLS_TESTING) == testing
    assert not any("debug" in n or n.endswith("_ops") for n in declared)


def test_distance_params_match_oracle(built):
    for d in [0.03, 0.1, 0.299, 0.5, 1.0, 1.25, 1.5, 2.0, 4.0, 8.0, 9.5, 16.0, 25.0]:
        a, b = built.distance_params(d), T.distance_params(d)
        for f, _ in a._fields_:
            assert getattr(a, f) == getattr(b, f), (d, f)


def test_file_header_bits(built):
    h = built.file_header(256, 256)
    assert h[:2] == b"\xff\x0a" and len(h) >= 6
    with pytest.raises(built.JxlTinyError):
        built.file_header(0, 5)


def test_device_count_without_gpu(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    hip = built.hip_lib()
    hip.jxlt_device_count.restype = C.c_int
    assert hip.jxlt_device_count() == 0
    hip.jxlt_bind_thread_near_device.argtypes = [C.c_int]
    assert hip.jxlt_bind_thread_near_device(0) != 0  # no device: an error code, no crash


def test_no_cpu_fallback_without_gpu(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(built.JxlTinyError, match="no HIP device"):
        built.Encoder(0)
    planes = np.zeros((3, 16, 16), np.float32)
    with pytest.raises(built.JxlTinyError):
        built.encode_file(planes, 1.0)
    with pytest.raises(built.JxlTinyError, match="no HIP device"):
        built.BatchEncoder(0, lanes=2)


@pytest.mark.parametrize("host_ingest", [False, True])
def test_cjxl_tiny_without_gpu_reports_an_encoding_failure(built, tmp_path, host_ingest):
    """A readable PFM on a machine without a GPU: the reference's messages for a failed EncodeFile
    (cjxl_main.cc:88-91), not "Error reading PFM input file" (VERDICT r3, weak 12); an unreadable file is
    still reported as such."""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    exe = str(T.ROOT / "libjxl-tiny_amd" / "host" / "cjxl_tiny")
    pfm = tmp_path / "in.pfm"
    T.write_pfm(pfm, T.synthetic_image(40, 24))
    extra = ["--host-ingest"] if host_ingest else []
    r = subprocess.run([exe, str(pfm), str(tmp_path / "out.jxl")] + extra, capture_output=True, text=True)
    assert r.returncode != 0
    assert "Read 40x24 pixels input image." in r.stderr and "Encoding failed." in r.stderr
    assert "no usable HIP device" in r.stderr or "no HIP device" in r.stderr
    assert "Error reading PFM" not in r.stderr and not (tmp_path / "out.jxl").exists()
    r = subprocess.run([exe, str(tmp_path / "missing.pfm")] + extra, capture_output=True, text=True)
    assert r.returncode != 0 and "Error reading PFM input file." in r.stderr


REF_MAIN = Path("/root/reference/encoder/cjxl_main.cc")


def test_the_references_own_caller_compiles_unmodified_against_the_drop_in_headers(built, tmp_path):
    """/root/reference/encoder/cjxl_main.cc, IN PLACE and unmodified, compiles with -I<host> at
    the reference's own language level and links against libjxltiny_host.so -- nothing of the reference is copied.
    (It needs encoder/base/printf_macros.h and <string.h> through encoder/image.h, cjxl_main.cc:10,52,26.)
    Skipped where the reference checkout is absent (the GPU box runs the prebuilt oracle/_ref binary instead)."""
    import subprocess
    if not REF_MAIN.exists():
        pytest.skip("no reference checkout on this machine")
    host = T.ROOT / "libjxl-tiny_amd" / "host"
    exe = tmp_path / "ref_main"
    for std in ("-std=c++11", "-std=c++17"):
        r = subprocess.run(["g++", std, "-O1", "-Wall", "-Werror", "-I" + str(host), "-o", str(exe), str(REF_MAIN),
                            "-L" + str(host), "-ljxltiny_host", "-L" + str(host.parent / "csrc"), "-ljxltiny_hip",
                            "-Wl,-rpath," + str(host), "-Wl,-rpath," + str(host.parent / "csrc")],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe), "-h"], capture_output=True, text=True)
    assert r.returncode == 0 and "[-d distance]" in r.stderr and "--device" not in r.stderr  # the reference's usage text
    # build() made the same binary by the committed recipe (oracle/Makefile: ref-caller)
    assert (T.ROOT / "oracle" / "_ref" / "cjxl_tiny_ref_main").exists()
    import torch
    if torch.cuda.is_available():
        return
    pfm = tmp_path / "in.pfm"
    T.write_pfm(pfm, T.synthetic_image(40, 24))
    r = subprocess.run([str(exe), str(pfm), str(tmp_path / "out.jxl")], capture_output=True, text=True)
    assert r.returncode != 0 and "Read 40x24 pixels input image." in r.stderr and "Encoding failed." in r.stderr
    r = subprocess.run([str(exe), str(tmp_path / "missing.pfm")], capture_output=True, text=True)
    assert r.returncode != 0 and "Error reading PFM input file." in r.stderr
