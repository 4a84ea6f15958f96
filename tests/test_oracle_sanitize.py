"""AddressSanitizer + UndefinedBehaviorSanitizer build of the oracle's C helpers (CPU only: GPU sanitizers are not
available on this pool).  oracle/fast_selftest.c drives every entry point of oracle/fast.c on ragged inputs with
exact-size heap buffers; any out-of-bounds access, signed overflow or misaligned load fails the run."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_oracle_fast_c_is_clean_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "fast_selftest")
    src = [os.path.join(ROOT, "oracle", "fast.c"), os.path.join(ROOT, "oracle", "fast_selftest.c")]
    cmd = ["gcc", "-O1", "-g", "-fopenmp", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-fno-omit-frame-pointer", *src, "-lm", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "asan" in (r.stderr or "").lower():
        pytest.skip("libasan not installed: " + r.stderr[-200:])
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               OMP_NUM_THREADS="4")
    env.pop("LD_PRELOAD", None)
    run = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0, (run.stdout[-1000:], run.stderr[-3000:])
    assert "checksum" in run.stdout
