"""The host packer (csrc/pack.cpp: the only native code that runs on the CPU) built with -fsanitize=address,undefined and driven
over every network description it supports (GPU sanitizers are not available on the pool; this is the CPU build the round
instructions ask for)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_host_packer_is_clean_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "pack_sanitize")
    cmd = ["g++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "nefes_amd", "csrc"),
           os.path.join(ROOT, "tests", "native", "pack_sanitize_main.cpp"), os.path.join(ROOT, "nefes_amd", "csrc", "pack.cpp"), "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True)
    if b.returncode != 0 and "sanitize" in b.stderr and ("cannot find" in b.stderr or "unrecognized" in b.stderr):
        pytest.skip("this toolchain has no sanitizer runtime")
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert r.stdout.count("pack rc 0") == 16 and "ERROR" not in r.stderr and "runtime error" not in r.stderr
