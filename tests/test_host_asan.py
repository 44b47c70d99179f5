"""AddressSanitizer over the HOST half of libynet_hip.so (the GPU pool refuses device ASan): an
instrumented build of every translation unit (`make asan`, device code untouched) driven on the CPU by
tools/asan_host_driver.cpp -- every pure-host entry point over a shape sweep, every launching entry point through the
argument checks that must reject the call.  Needs hipcc (about a minute of compile time); no GPU."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "motion-style-transfer_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
ASAN_DIR = os.path.join(ROOT, "build", "asan")      # (build/ is git- and gpurun-ignored)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_host_paths_are_clean_under_asan(tmp_path):
    r = subprocess.run(["make", "-C", CSRC, "-j8", "asan"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    exe = str(tmp_path / "asan_host_driver")
    clang = "/opt/rocm/lib/llvm/bin/clang++"
    r = subprocess.run([clang if os.path.exists(clang) else shutil.which("clang++") or HIPCC, "-std=c++17", "-O1", "-g",
                        "-fsanitize=address", "-shared-libsan", os.path.join(ROOT, "tools", "asan_host_driver.cpp"),
                        "-L" + ASAN_DIR, "-lynet_hip_asan", "-Wl,-rpath," + ASAN_DIR,
                        "-Wl,-rpath,/opt/rocm/lib", "-o", exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    import glob
    rt = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1",
               LD_LIBRARY_PATH=os.pathsep.join([os.path.dirname(rt[0])] if rt else []) + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
    assert "0 failures" in r.stdout
