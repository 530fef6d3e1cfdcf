import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


# ---------------------------------------------------------------------------------------------
# Child-process launcher.  Multi-rank tests start fresh Python processes (one per rank).  On the
# GPU pool a process that has initialised the GPU must not exec another program, and this pytest
# process initialises it as soon as the first GPU test runs -- so a tiny helper is started HERE,
# at import time, before anything touches the GPU (counting devices does not), and all later
# launches go through it.  It never imports torch or the library.
# ---------------------------------------------------------------------------------------------
_LAUNCHER_SRC = r"""
import json, subprocess, sys
for line in sys.stdin:
    req = json.loads(line)
    procs = [subprocess.Popen(c, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              env=req.get("env"), cwd=req.get("cwd")) for c in req["cmds"]]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=req.get("timeout", 900))
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append({"rc": p.returncode, "out": out.decode("utf-8", "replace")[-20000:]})
    sys.stdout.write(json.dumps(outs) + "\n")
    sys.stdout.flush()
"""


class _Launcher(object):
    def __init__(self):
        import subprocess
        self.p = subprocess.Popen([sys.executable, "-c", _LAUNCHER_SRC], stdin=subprocess.PIPE,
                                  stdout=subprocess.PIPE, universal_newlines=True)

    def close(self):
        try:
            self.p.stdin.close()
            self.p.wait(timeout=10)
            self.p.stdout.close()
        except Exception:
            pass

    def run(self, cmds, timeout=900, env=None):
        """Run the commands concurrently; returns [{'rc':..., 'out':...}, ...]."""
        import json
        req = {"cmds": cmds, "timeout": timeout, "cwd": ROOT}
        if env is not None:
            req["env"] = env
        self.p.stdin.write(json.dumps(req) + "\n")
        self.p.stdin.flush()
        return json.loads(self.p.stdout.readline())


def _device_count():
    try:
        import torch
        return torch.cuda.device_count()     # does not initialise the GPU
    except Exception:
        return 0


_LAUNCHER = _Launcher() if _device_count() > 0 else None
if _LAUNCHER is not None:
    import atexit
    atexit.register(_LAUNCHER.close)


@pytest.fixture(scope="session")
def launcher():
    if _LAUNCHER is None:
        pytest.skip("no GPU: the pre-GPU process launcher was not started")
    return _LAUNCHER


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load


def split(flat, lengths):
    out, s = [], 0
    for n in lengths:
        out.append(flat[s:s + int(n)])
        s += int(n)
    return out
