import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(REPO, "tests", "golden")


# ---- two ranks on ONE GPU (tests/test_gpu_two_ranks.py).  The rank processes are FORKED here, at session start, before anything in
# this process has touched the GPU: a fork of a process that has initialised HIP cannot use it, and the GPU boxes refuse an exec
# from one (so no spawn either).  They idle on a pipe until a test sends them a scenario (tests/two_rank_scenarios.py).
_RANKS = {"procs": [], "pipes": []}


def _rank_main(rank, world, conn):
    import traceback
    while True:
        try:
            msg = conn.recv()
        except EOFError:
            return
        if msg is None:
            return
        name, kwargs = msg
        try:
            from tests import two_rank_scenarios
            conn.send(("ok", getattr(two_rank_scenarios, name)(rank, world, **kwargs)))
        except BaseException:  # noqa: BLE001 -- the test reports it
            conn.send(("error", traceback.format_exc()))


def pytest_sessionstart(session):
    expr = session.config.getoption("markexpr", "") or ""
    if "gpu" not in expr or "not gpu" in expr:
        return
    import multiprocessing as mp
    ctx = mp.get_context("fork")
    for rank in range(2):
        parent, child = ctx.Pipe()
        p = ctx.Process(target=_rank_main, args=(rank, 2, child), daemon=True)
        p.start()
        child.close()
        _RANKS["procs"].append(p)
        _RANKS["pipes"].append(parent)


def pytest_sessionfinish(session, exitstatus):
    for c in _RANKS["pipes"]:
        try:
            c.send(None)
        except Exception:
            pass
    for p in _RANKS["procs"]:
        p.join(timeout=20)
        if p.is_alive():
            p.kill()


@pytest.fixture(scope="session")
def two_ranks():
    """run(name, **kwargs): scenario `name` of tests/two_rank_scenarios.py on both rank processes at once; returns their results"""
    if len(_RANKS["procs"]) != 2 or not all(p.is_alive() for p in _RANKS["procs"]):
        pytest.skip("the rank processes are started only by a `-m gpu` session")

    def run(name, timeout=300, **kwargs):
        import socket
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        for c in _RANKS["pipes"]:
            c.send((name, dict(kwargs, port=port)))
        out = []
        for r, c in enumerate(_RANKS["pipes"]):
            if not c.poll(timeout):
                raise AssertionError("rank %d did not answer within %d s" % (r, timeout))
            status, val = c.recv()
            if status != "ok":
                raise AssertionError("rank %d failed:\n%s" % (r, val))
            out.append(val)
        return out
    return run
