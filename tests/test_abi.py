"""CPU suite: the C-ABI library builds/loads, exports every symbol include/sph_mi355x.h declares,
and the product fails loudly (no CPU fallback) when there is no GPU."""
import ctypes
import os
import re

import pytest

from cfd_taichi_amd import _native, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "sph_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sph_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_functions() == sorted(_native.EXPORTS)


def test_library_exports_every_declared_symbol():
    lib = _native.load()
    for name in declared_functions():
        assert hasattr(lib, name), name


def test_struct_layouts(tmp_path):
    """ctypes mirrors vs the header, as seen by a C compiler: sizes of all five structs and the offsets of the members where
    mirrors drift most easily (pointers, size_t, function pointers, the first field after a block of doubles)."""
    import subprocess
    probes = [("sizeof(SphConfig)", ctypes.sizeof(_native.SphConfig)), ("sizeof(SphStepStats)", ctypes.sizeof(_native.SphStepStats)),
              ("sizeof(SphSizes)", ctypes.sizeof(_native.SphSizes)), ("sizeof(SphRigid)", ctypes.sizeof(_native.SphRigid)),
              ("sizeof(SphComm)", ctypes.sizeof(_native.SphComm))]
    for struct, cls, fields in (("SphConfig", _native.SphConfig, ["boundary_handle", "max_density_iters", "slab_rebalance_every", "arith", "slab_ghost_layers", "slab_overlap", "reserved"]),
                                ("SphStepStats", _native.SphStepStats, ["capped", "div_first_err", "dt", "lost"]),
                                ("SphRigid", _native.SphRigid, ["points", "vertices", "rho_0", "pos_offset", "attitude_offset", "active"]),
                                ("SphComm", _native.SphComm, ["exchange_counts", "exchange_buffers", "allreduce", "send_left", "recv_right", "capacity", "on_host",
                                                              "stream_ordered", "allreduce_stream", "reduce_buf", "exchange_counts_n", "reduce_capacity"])):
        for f in fields:
            probes.append(("offsetof(%s, %s)" % (struct, f), getattr(cls, f).offset))
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "sph_mi355x.h"\nint main(void){\n' +
                   "".join('printf("%%zu\\n", (size_t)%s);\n' % expr for expr, _ in probes) + "return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert {expr: int(v) for (expr, _), v in zip(probes, out)} == dict(probes)


def test_profile_kernel_names():
    lib = _native.load()
    names = [lib.sph_profile_kernel_name(k).decode() for k in range(lib.sph_profile_kernel_count())]
    assert "build_nl" in names and "dfsph_div_residual" in names and len(set(names)) == len(names)


def _gpu_present():
    try:
        import torch
        return torch.cuda.device_count() > 0
    except Exception:
        return os.path.exists("/dev/kfd")


@pytest.mark.skipif(_gpu_present(), reason="checks the no-GPU failure mode")
def test_no_gpu_is_a_loud_error():
    cfg = _native.config_from_dict(scenes.get("wcsph_tiny_wall"))
    with pytest.raises(_native.SphError) as e:
        _native.Simulation(cfg)
    assert e.value.code == _native.SPH_E_NO_DEVICE


def test_unknown_solver_rejected():
    cfg = scenes.get("wcsph_tiny_wall")
    cfg["solver"]["name"] = "mpm"
    with pytest.raises(NotImplementedError):
        _native.config_from_dict(cfg)
    cfg["solver"]["name"] = "pbf"
    assert _native.config_from_dict(cfg).solver == _native.SOLVER_PBF


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "cfd_taichi_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "sph_oracle" not in text and "liborc" not in text, f


def test_bench_launches_its_own_ranks_without_touching_the_gpu(monkeypatch):
    """`python bench.py --gpus N` without a launcher: a CHILD process running torch.distributed.run on 127.0.0.1,
    started before anything in the parent imports torch; the parent exits with the child's code."""
    import importlib.util
    import subprocess
    import sys
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, **kw):
        seen["cmd"] = cmd
        return Done()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    # (the first-contact child job that self_launch() starts before the ranks has its own suite, tests/test_first_contact.py)
    monkeypatch.setenv(bench.FIRST_CONTACT_ENV, '{"backend": "nccl", "transport": "rccl", "discipline": "sync", "reached": [], "how": "test"}')
    torch_loaded_before = "torch" in sys.modules
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert ("torch" in sys.modules) == torch_loaded_before          # the parent did not import torch on the way


def test_removal_build_patches_still_match_the_kernels():
    """tools/removal_build.py builds its experiment variants from a patched COPY of csrc/ (the product kernels carry no experiment
    switches); a patch is a text replacement that has to match the current source exactly -- checked here so that the recipes
    behind the numbers DESIGN.md quotes do not rot when a kernel changes."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("removal_build", os.path.join(ROOT, "tools", "removal_build.py"))
    rb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rb)
    for name, patches in rb.PATCHES.items():
        for fname, old, new, *want in patches:
            text = open(os.path.join(ROOT, "cfd_taichi_amd", "csrc", fname)).read()
            assert text.count(old) == (want[0] if want else 1), (name, fname, old[:60])
            assert old != new


def test_development_overrides_need_sph_dev():
    """SPH_LIB (which could load ANY .so as the product, the oracle's ABI library included) and the SPH_* knobs inside the
    library are development overrides -- refused / ignored without SPH_DEV=1."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k != "SPH_DEV"}
    env["SPH_LIB"] = os.path.join(ROOT, "oracle", "liborc_abi.so")
    code = "import cfd_taichi_amd; from cfd_taichi_amd import _native; _native.load()"
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "SPH_DEV=1" in r.stderr, r.stderr
    env["SPH_DEV"] = "1"
    env["SPH_LIB"] = os.path.join(ROOT, "cfd_taichi_amd", "libsph_mi355x.so")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # every knob of the library goes through the gate: no bare getenv("SPH_...") is left in the product sources
    csrc = os.path.join(ROOT, "cfd_taichi_amd", "csrc")
    text = "".join(open(os.path.join(csrc, f)).read() for f in sorted(os.listdir(csrc)))
    assert re.findall(r'[^_a-z]getenv\("SPH_(?!DEV")', text) == []


def test_loopback_stand_in_exports_what_the_native_transport_binds():
    """tests/loopback_rccl.hip (test infrastructure: the in-process stand-in for librccl, loaded through the development override SPH_RCCL_LIB)
    must export every entry point the library's RcclApi (csrc/sph_host_transport.h) looks up -- otherwise the GPU suite's loopback tests would fall back to nothing."""
    import ctypes
    import importlib.util
    spec = importlib.util.spec_from_file_location("loopback_worker", os.path.join(ROOT, "tests", "loopback_worker.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    try:
        path = mod.shim_path(build=True)
    except Exception as e:  # noqa: BLE001 - a box without hipcc / the rccl headers: test infrastructure, not the product
        pytest.skip("tests/loopback_rccl.hip does not build here: %s" % e)
    so = ctypes.CDLL(path)
    text = open(os.path.join(ROOT, "cfd_taichi_amd", "csrc", "sph_host_transport.h")).read()          # (a section of sph_mi355x.hip's translation unit)
    bound = sorted(set(re.findall(r'SPH_RCCL_SYM\([A-Za-z]+, "(nccl[A-Za-z]+)"\)', text)))
    assert len(bound) == 9, bound
    for name in bound:
        assert hasattr(so, name), name
    # the override is gated like every other: the loader reads SPH_RCCL_LIB through dev_env
    assert 'dev_env(nullptr, "SPH_RCCL_LIB")' in text
