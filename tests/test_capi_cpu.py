"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and
exports exactly what include/rdis_hip.h declares; without a GPU it fails loudly
instead of computing anything."""
import ctypes as C
import os
import re

import pytest

from rdis_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "rdis_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rdis_hip_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_bound_and_exported():
    names = _declared()
    assert len(names) >= 25
    assert sorted(capi.SYMBOLS) == names          # the python binding covers the whole header
    lib = capi.load_library()
    for n in names:
        assert getattr(lib, n) is not None        # dlsym succeeds
    assert lib.rdis_hip_abi_version() == 1


def test_no_oracle_on_the_product_path():
    """nothing under rdis_amd/ may import, link or execute the oracle"""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "rdis_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", "Makefile")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in src and "rdis_oracle" not in src, os.path.join(dirpath, f)
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dirpath, f)


@pytest.mark.skipif(capi.load_library().rdis_hip_device_count() > 0, reason="a GPU is present")
def test_fails_loudly_without_gpu():
    lib = capi.load_library()
    h = C.c_void_p()
    assert lib.rdis_hip_create(0, C.byref(h)) == -3      # RDIS_HIP_EDEVICE, no fallback
    with pytest.raises(capi.RdisHipError):
        capi.Context(0)


def test_null_arguments_are_rejected():
    lib = capi.load_library()
    assert lib.rdis_hip_create(0, None) == -1
    assert lib.rdis_hip_plan_solve(None, 10, 1e-8) == -1
    assert lib.rdis_hip_eval(None, 0, None, None) == -1


def test_inline_asm_granule_loads_are_waited_for_before_use():
    """grid_sync.hpp / solver_pipe.hpp load granule pairs with an inline-assembly `global_load_dwordx4 ... sc1` the
    compiler's wait counting does not see.  tools/check_async_loads.py disassembles the gfx950 code object
    of the library as built and verifies that no instruction touches such a load's destination registers before the
    explicit `s_waitcnt vmcnt(0)` that follows the batch."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    obj = os.path.join(root, "rdis_amd", "lib", "obj", "rdis_hip.o")
    if not os.path.exists(obj) or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("object file or llvm-objdump not available")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "check_async_loads.py"), obj], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert " 0 touched before" in out.stdout and " 0 reach a branch first" in out.stdout, out.stdout


def test_host_block_cache(tmp_path):
    """rdis_amd/csrc/host_blocks.hpp -- where a plan's host index arrays live (no HIP in it): size classes, a block given
    back is the block taken next, page alignment, the cache's byte limit, vectors over it (tests/cpp/host_blocks_test.cpp)"""
    import subprocess
    exe = str(tmp_path / "host_blocks_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-pthread", "-o", exe,
                           os.path.join(ROOT, "tests", "cpp", "host_blocks_test.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stdout + out.stderr


def test_matrix_form_of_a_trial_agrees_with_the_vector_form(tmp_path):
    """factors.hpp on the host (no GPU): a line-search trial in matrix form -- what the streaming solver evaluates
    (ba_camera_trial / ba_camera_trial_dir / ba_trial_value / ba_trial_slope: rotation matrix and its derivative along
    the direction once per camera and trial point) -- against the vector form every other solver uses (ba_forward +
    ba_slope_dir, the model of src/bundleadjust/BundleAdjustmentFactor.cpp:266-335) on 200 000 random cameras, points,
    directions and observations of ladybug's ranges, with free and fixed cameras and at theta = 0: values to 1e-13,
    slopes to 1e-11 of their terms' size (tests/cpp/factors_forms_test.hip asserts it; measured 8e-14 / 5e-15)."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    exe = str(tmp_path / "factors_forms_test")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-o", exe,
                           os.path.join(ROOT, "tests", "cpp", "factors_forms_test.hip")], stderr=subprocess.DEVNULL)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert "worst relative deviation" in out.stdout


def test_oracle_switches_restate_the_device_factor_arithmetic_bit_for_bit(tmp_path):
    """factors.hpp compiled for the HOST without contraction (what refround_kernels.hip instantiates on the device: the parity
    option's factor arithmetic) against the oracle's plain-C restatement of it -- RO_ARITH_RECIPROCAL | RO_ARITH_SINCOS_ANGLE,
    RO_BA_DERIV_ADJOINT_DEVICE -- on 300 000 random cameras / points / observations of ladybug's ranges, rotation angles up to 40
    radians and theta = 0: value and twelve partials ==, the angle routine == and within one ulp of the C library's
    (tests/cpp/factors_parity_test.hip).  The GPU suite then shows device == oracle per factor and end to end."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    from oracle import oracle as O
    O.build()
    exe = str(tmp_path / "factors_parity_test")
    odir = os.path.join(ROOT, "oracle")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe,
                           os.path.join(ROOT, "tests", "cpp", "factors_parity_test.hip"), "-L" + odir, "-loracle", "-Wl,-rpath," + odir],
                          stderr=subprocess.DEVNULL)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "300000 cases, 0 differences" in out.stdout, out.stdout


def test_optba_entry_of_the_host_library():
    """include/rdis_optba.h: the caller-side C entry (optBA's core over the level driver) is exported by librdis_host.so and
    rejects bad calls without touching a device."""
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "rdis_amd", "host")], stdout=subprocess.DEVNULL)
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "rdis_optba.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(rdis_optba_[a-z_0-9]+)\s*\(", text)))
    assert names == ["rdis_optba_run", "rdis_optba_run_hist"]
    lib = C.CDLL(os.path.join(ROOT, "rdis_amd", "lib", "librdis_host.so"))
    for n in names:
        assert getattr(lib, n) is not None
    out = (C.c_double * 10)()
    run = lib.rdis_optba_run
    run.argtypes = [C.c_char_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    assert run(None, 0, 0, 1, 0, None, None, 0, out, None) == -3              # no file name
    assert run(b"x", 0, 0, 7, 0, None, None, 0, out, None) == -3              # no such schedule
    assert run(b"/nonexistent/problem.txt", 0, 0, 1, 0, None, None, 0, out, None) == -1   # cannot be loaded
