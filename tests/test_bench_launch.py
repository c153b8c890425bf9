"""`python bench.py --gpus N` must start its own ranks (the driver runs exactly that command shape): checked here on CPU with the
gloo backend and --dry (rank launch + per-step scatter/gather + the JSON line; no GPU, no kernels).  SURVEY section 8(e)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):       # the parent must not look like a launched rank
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--backend", "gloo", "--dry", "--steps", "3", "--warmup", "1", "--batch", "2", *extra],
                       capture_output=True, text=True, timeout=150, env=e)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p.returncode, lines, p.stderr


@pytest.mark.timeout(180)
@pytest.mark.parametrize("extra", [(), ("--gather-masks",)], ids=["compact", "with_live_masks"])
def test_plain_command_starts_its_own_ranks(lib_built, extra):
    rc, lines, err = _run("--gpus", "2", *extra)
    assert rc == 0 and len(lines) == 1, (rc, lines, err[-2000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks"] == 2 and d["backend"] == "gloo" and d["exchange_ok"] is True
    assert d["config"]["global_batch"] == 4 and d["config"]["parallelism"] == "dp2" and "scatter" in d["exchange"]


@pytest.mark.timeout(180)
def test_launched_ranks_are_accepted_too(lib_built):
    """the torch.distributed.run shape: RANK / WORLD_SIZE already in the environment -> no second launch; a mismatch is refused"""
    rc, lines, _ = _run("--gpus", "1", env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    assert rc == 0 and json.loads(lines[0])["n_gpus"] == 1
    rc, lines, _ = _run("--gpus", "2", env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    assert rc == 2 and not lines
