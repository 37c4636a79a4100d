"""CPU suite: bench.py bounds its first contact with real RCCL.

`bench.py --gpus N` first runs the nccl process group, the halo probe and the discipline self-check in a short-lived CHILD job of fresh
processes under a wall-clock limit (bench.first_contact_probe); the measuring ranks only execute what that child has survived.  Here the child
is a stub that reports some stages and then HANGS (or dies, or finishes): the parent must come back inside the limit with the fallback the
reached stages justify, and the child's whole process group must be gone."""
import importlib.util
import json
import os
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_first_contact", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


STUB = r"""
import json, os, subprocess, sys, time
stages, then = json.loads(sys.argv[1]), sys.argv[2]
open(sys.argv[3], "w").write(str(os.getpid()))
grandchild = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(3600)"])      # a rank of the job: must die with it
open(sys.argv[3] + ".rank", "w").write(str(grandchild.pid))
print("some unrelated line", flush=True)
for st in stages:
    print("FIRST_CONTACT " + json.dumps(st), flush=True)
    time.sleep(0.05)
if then == "hang":
    time.sleep(3600)
grandchild.kill()
sys.exit(3 if then == "die" else 0)
"""

START, IMPORT, INIT = {"stage": "start"}, {"stage": "import"}, {"stage": "init"}
RCCL, GLOO, SYNC = {"stage": "transport", "transport": "rccl"}, {"stage": "transport", "transport": "gloo"}, {"stage": "sync_ok"}
NATIVE = {"stage": "discipline", "discipline": "native", "why": "2 steps reproduce the synchronous discipline byte for byte"}


def alive(pid):
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    try:       # a zombie is dead for our purposes
        return open("/proc/%d/stat" % pid).read().split(") ")[1][0] != "Z"
    except OSError:
        return False


@pytest.mark.parametrize("stages,then,expect", [
    ([], "hang", ("gloo", "gloo", None)),                                        # nothing proven: not even the nccl process group
    ([START, IMPORT], "hang", ("gloo", "gloo", None)),                           # stuck in init_process_group(nccl)
    ([START, IMPORT, INIT], "hang", ("gloo", "gloo", None)),                     # stuck in the halo probe
    ([START, IMPORT, INIT, GLOO], "hang", ("nccl", "gloo", None)),               # the probe itself fell back to host staging
    ([START, IMPORT, INIT, RCCL], "hang", ("gloo", "gloo", None)),               # the synchronous discipline hung over rccl
    ([START, IMPORT, INIT, RCCL, SYNC], "hang", ("nccl", "rccl", "sync")),       # a faster discipline hung: the proven one
    ([START, IMPORT, INIT, RCCL, SYNC], "die", ("nccl", "rccl", "sync")),        # ... or crashed
    ([START, IMPORT, INIT, RCCL, SYNC, NATIVE], "exit", ("nccl", "rccl", "native")),
    ([START, IMPORT, INIT, RCCL, SYNC, NATIVE], "hang", ("nccl", "rccl", "native")),      # everything reported, then stuck on its way out: proven all the same
])
def test_a_stuck_child_costs_the_limit_and_no_more(bench, tmp_path, stages, then, expect):
    pidfile = str(tmp_path / "pid")
    limits = {k: 1.5 for k in bench.FIRST_CONTACT_ORDER}
    t0 = time.monotonic()
    v = bench.first_contact_probe(4, cmd=[sys.executable, "-c", STUB, json.dumps(stages), then, pidfile], limits=limits, total=6.0)
    took = time.monotonic() - t0
    assert (v["backend"], v["transport"], v["discipline"]) == expect, v
    assert v["reached"] == [s["stage"] for s in stages]
    assert took < 6.0 + 3.0, took
    if then == "hang" and len(stages) < 6:
        assert took >= 1.0 and "killed" in v["how"]
    time.sleep(0.2)
    assert not alive(int(open(pidfile).read())), "the child survived"
    assert not alive(int(open(pidfile + ".rank").read())), "a process of the child's group survived"


def test_the_total_limit_holds_even_if_every_stage_is_slow(bench, tmp_path):
    """stage limits are per stage; a child that dawdles through all of them still ends at the total"""
    slow = STUB.replace("time.sleep(0.05)", "time.sleep(0.8)")
    pidfile = str(tmp_path / "pid")
    t0 = time.monotonic()
    v = bench.first_contact_probe(2, cmd=[sys.executable, "-c", slow, json.dumps([START, IMPORT, INIT, RCCL, SYNC, NATIVE]), "hang", pidfile],
                                  limits={k: 5.0 for k in bench.FIRST_CONTACT_ORDER}, total=2.5)
    assert time.monotonic() - t0 < 2.5 + 3.0
    assert v["discipline"] != "native" and "killed" in v["how"]
    assert not alive(int(open(pidfile).read()))


def test_the_measuring_ranks_take_the_verdict_from_the_environment(bench, monkeypatch):
    """self_launch() probes first and hands the verdict to the ranks it starts; with a verdict already in the environment it does not probe again"""
    import subprocess
    seen = {}

    class Done:
        returncode = 0

    monkeypatch.setattr(subprocess, "run", lambda cmd, **kw: seen.update(cmd=cmd, env=kw.get("env")) or Done())
    monkeypatch.setattr(bench, "first_contact_probe", lambda n, rebalance=0, **kw: {"backend": "nccl", "transport": "rccl", "discipline": "sync", "reached": [], "how": "stub", "n": n,
                                                                                    "rebalance": rebalance})
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--rebalance", "25"])
    monkeypatch.delenv(bench.FIRST_CONTACT_ENV, raising=False)
    monkeypatch.delenv("SPH_BENCH_REHEARSAL", raising=False)
    assert bench.self_launch(8) == 0
    handed = json.loads(seen["env"][bench.FIRST_CONTACT_ENV])
    assert (handed["discipline"], handed["n"], handed["rebalance"]) == ("sync", 8, "25")
    monkeypatch.setenv(bench.FIRST_CONTACT_ENV, json.dumps(handed))
    monkeypatch.setattr(bench, "first_contact_probe", lambda *a, **k: pytest.fail("probed although a verdict was handed over"))
    assert bench.self_launch(8) == 0 and seen["env"] is None


def test_ranks_under_a_foreign_launcher_share_one_probe(bench, monkeypatch, tmp_path):
    """the driver starts the ranks itself (torch.distributed.run ... bench.py --gpus N): local rank 0 probes, the others wait for its file"""
    import tempfile
    import threading
    monkeypatch.setattr(tempfile, "gettempdir", lambda: str(tmp_path))
    monkeypatch.setenv("MASTER_PORT", "29555")
    calls = []

    def probe(n, rebalance=0, **kw):
        calls.append(n)
        time.sleep(0.5)
        return {"backend": "nccl", "transport": "rccl", "discipline": "native", "reached": list(bench.FIRST_CONTACT_ORDER), "how": "stub"}

    monkeypatch.setattr(bench, "first_contact_probe", probe)
    # a leftover of an earlier job with the same launcher pid and port must not be taken for this job's verdict
    stale = os.path.join(str(tmp_path), "sph_first_contact_%d_29555.json" % os.getppid())
    json.dump({"backend": "gloo", "transport": "gloo", "discipline": None, "reached": [], "how": "stale", "written_at": time.time() - 3600}, open(stale, "w"))
    got = {}
    waiter = threading.Thread(target=lambda: got.update(v=bench.first_contact_for_rank(3, 4, 0)))
    waiter.start()
    time.sleep(0.2)
    v0 = bench.first_contact_for_rank(0, 4, 0)
    waiter.join(timeout=10)
    assert calls == [4] and v0["discipline"] == "native" and got["v"]["discipline"] == "native" and got["v"]["how"] == "stub"
