"""The package's HIP runtime defaults (mmlrec_amd/__init__.py: _runtime_defaults)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(env_value):
    env = dict(os.environ)
    env.pop("HIP_FORCE_DEV_KERNARG", None)
    if env_value is not None:
        env["HIP_FORCE_DEV_KERNARG"] = env_value
    code = ("import os, sys; sys.path.insert(0, %r); import mmlrec_amd; "
            "print(os.environ.get('HIP_FORCE_DEV_KERNARG'), mmlrec_amd.runtime['HIP_FORCE_DEV_KERNARG'], "
            "mmlrec_amd.runtime['set_before_hip_init'])" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout.strip().splitlines()[-1].split()


def test_import_sets_device_kernel_arguments_unless_the_environment_decides():
    assert _run(None) == ["1", "1", "True"]      # the default: kernel-argument blocks in device memory
    assert _run("0") == ["0", "0", "True"]       # an explicit setting wins


def test_bench_and_driver_entry_set_it_before_torch_is_imported():
    for name in ("bench.py", "__graft_entry__.py", os.path.join("tests", "conftest.py")):
        src = open(os.path.join(ROOT, name)).read()
        at = src.index('os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")')
        first_torch = src.find("\nimport torch")
        assert first_torch < 0 or at < first_torch, name
