"""bench.py's second-hand numbers are checkable: roofline.traffic and roofline.valu come from committed rocprofv3 --pmc
passes, which store the sha256 of the kernel sources they were taken on; a run on other sources labels them "stale"."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_committed_traffic_is_labelled_against_the_sources(tmp_path):
    from cfd_taichi_amd import build as hip_build
    bench = load_bench()
    sha = hip_build.sources_sha256()
    assert len(sha) == 64 and sha == hip_build.sources_sha256()
    kernels = {"dfsph_div_residual": {"hbm_bytes_per_launch": 1.0e8}}
    cur, old, bare = tmp_path / "cur.json", tmp_path / "old.json", tmp_path / "bare.json"
    cur.write_text(json.dumps({"csrc_sha256": sha, "kernels": kernels}))
    old.write_text(json.dumps({"csrc_sha256": "0" * 64, "kernels": kernels}))
    bare.write_text(json.dumps({"kernels": kernels}))
    assert bench.load_traffic("dfsph_div_residual", path=str(cur)) == (1.0e8, "current")
    assert bench.load_traffic("dfsph_div_residual", path=str(old)) == (1.0e8, "stale")
    assert bench.load_traffic("dfsph_div_residual", path=str(bare)) == (1.0e8, "unknown")
    assert bench.load_traffic("no_such_kernel", path=str(cur)) == (None, None)
    assert bench.load_traffic("dfsph_div_residual", path=str(tmp_path / "absent.json")) == (None, None)


def test_the_digest_follows_the_sources(tmp_path, monkeypatch):
    """Any edit of a kernel source changes the digest (here: a copy of csrc/ with one byte appended to one header)."""
    import shutil
    from cfd_taichi_amd import build as hip_build
    before = hip_build.sources_sha256()
    copy = tmp_path / "csrc"
    shutil.copytree(hip_build.CSRC, copy)
    os.makedirs(tmp_path / "include", exist_ok=True)
    monkeypatch.setattr(hip_build, "CSRC", str(copy))
    monkeypatch.setattr(hip_build, "HEADERS", [h for h in hip_build.HEADERS if not h.startswith("..")])
    same_files = hip_build.sources_sha256()
    with open(copy / "sph_device.h", "a") as f:
        f.write("\n")
    assert hip_build.sources_sha256() != same_files and len(before) == 64


def test_one_traffic_file_per_arithmetic():
    """profiles/pmc_traffic.json (exact) and profiles/pmc_traffic_relaxed.json are the only files bench.py reads traffic from: no path into a
    per-round directory (earlier rounds keep their own historical copies; nothing reads them)."""
    import re
    text = open(os.path.join(ROOT, "bench.py")).read()
    assert not re.search(r'"profiles",\s*"r\d', text) and not re.search(r"profiles/r\d+/pmc_traffic", text)
    for name in ("pmc_traffic.json", "pmc_traffic_relaxed.json"):
        assert os.path.exists(os.path.join(ROOT, "profiles", name)), name
