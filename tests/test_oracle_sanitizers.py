"""The oracle under AddressSanitizer + UndefinedBehaviorSanitizer (the reference relies on Rust's bounds checks and overflow panics; a C++
restatement has neither, so the checker itself is checked): `make -C oracle san` builds oracle/_san/liboracle_san.so, and the oracle's own CPU tests — known
answers, every integrator and sampler, textures, instancing, media, the motion bounds — run over it in a child process with libasan preloaded.  Any
out-of-bounds access, use after free, signed overflow, misaligned or null dereference, or shift out of range aborts the child (-fno-sanitize-recover)."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
FILES = ["tests/test_oracle_kat.py", "tests/test_oracle_integrators.py", "tests/test_oracle_textures.py", "tests/test_motion_bounds.py", "tests/test_volpath.py", "tests/test_pixel_samplers.py",
         "tests/test_instancing.py", "tests/test_materials.py", "tests/test_alpha_masks.py", "tests/test_host.py", "tests/test_bvh_build.py"]


def _libasan():
    p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_oracle_tests_pass_under_asan_and_ubsan():
    asan = _libasan()
    if asan is None:
        pytest.fail("gcc's libasan.so not found: the sanitizer run is part of the CPU suite")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "san"])
    env = dict(os.environ, LD_PRELOAD=asan, ORACLE_LIB=os.path.join(ROOT, "oracle", "_san", "liboracle_san.so"),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:handle_segv=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", PYTHONMALLOC="malloc")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + FILES, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0, tail
    assert "runtime error" not in r.stdout + r.stderr and "AddressSanitizer" not in r.stdout + r.stderr, tail
    assert " passed" in r.stdout, tail
