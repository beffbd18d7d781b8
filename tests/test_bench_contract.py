"""bench.py's launch contract, as far as it can be checked without a GPU: `--gpus N` with N > 1 and no launcher must start
N ranks itself -- or refuse.  It must never print a line that says n_gpus: 1 when more GPUs were asked for."""
import os
import subprocess
import sys

from helpers import ROOT


def _run(args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=300, env=e)


def test_more_gpus_than_the_box_has_is_refused():
    from climt_amd import _hip
    have = _hip.device_count()
    p = _run(["--gpus", str(have + 2), "--steps", "2"])
    assert p.returncode == 2, (p.returncode, p.stderr[-500:])
    assert "refusing" in p.stderr and not any(l.startswith("{") for l in p.stdout.splitlines())


def test_help_names_the_multi_gpu_knobs():
    p = _run(["--help"])
    assert p.returncode == 0
    for flag in ("--gpus", "--gather", "--no-unpack", "--rccl-channels", "--min-seconds", "--config"):
        assert flag in p.stdout, flag


def test_spawned_ranks_fail_fast_and_print_no_line_without_a_gpu():
    """`--gpus 2 --share-device` starts two ranks of this script; on a box without a GPU both die at once and the parent
    must come back promptly with their exit code and no JSON line (never a made-up n_gpus)."""
    from climt_amd import _hip
    if _hip.device_count() > 0:
        import pytest
        pytest.skip("a GPU is present: the ranks would run")
    p = _run(["--gpus", "2", "--share-device", "--dist-backend", "gloo", "--comm", "torch", "--steps", "2"])
    assert p.returncode != 0
    assert not any(l.startswith("{") for l in p.stdout.splitlines())
