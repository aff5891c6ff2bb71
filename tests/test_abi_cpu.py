"""CPU-side checks of the boundary: the shared library loads, exports every symbol declared in
include/mpe_hip.h, and the ctypes structure mirrors agree with the header."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "mpe_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mpe_[a-z_]+)\s*\(", text)))


def test_header_declares_expected_entry_points():
    names = _declared()
    for n in ("mpe_create", "mpe_destroy", "mpe_layout", "mpe_last_error", "mpe_step", "mpe_step_auto", "mpe_rollout", "mpe_episode",
              "mpe_observe", "mpe_reset", "mpe_world_step"):
        assert n in names


def test_library_exports_every_declared_symbol():
    from mpe_hip import _lib
    lib = _lib.load()
    for name in _declared():
        assert hasattr(lib, name), name
        assert name in _lib.EXPORTS, "binding is missing a prototype for " + name


def test_struct_mirrors_match_header_sizes():
    from mpe_hip import _lib
    # MpeSpec: 4 int32 + 4 double + 9 pointers + 2 int32 + 7 pointers + 2 int32 + 3 pointers ; MpeLayout: int64 + 10 int32 + 2 * 64 int32
    assert C.sizeof(_lib.MpeSpec) == 16 + 32 + 9 * 8 + 8 + 7 * 8 + 8 + 3 * 8
    assert C.sizeof(_lib.MpeLayout) == 8 + 10 * 4 + 2 * 64 * 4 + 8


def test_create_rejects_bad_arguments_without_a_gpu():
    from mpe_hip import _lib, spread_spec
    lib = _lib.load()
    spec, keep = spread_spec().to_c()
    h = C.c_void_p()
    assert lib.mpe_create(C.byref(spec), 0, _lib.MPE_F32, 0, C.byref(h)) == -1
    assert b"n_envs" in lib.mpe_last_error(None)
    assert lib.mpe_create(C.byref(spec), 8, 7, 0, C.byref(h)) == -1
    assert lib.mpe_create(None, 8, 0, 0, C.byref(h)) == -1


def test_specs_match_oracle_specs():
    """The product's scenario tables and the oracle's were written separately; they must agree."""
    import numpy as np
    import mpe_oracle as orc
    from mpe_hip import make_spec
    for name, kw_p, kw_o in (("simple_spread", {}, {}), ("simple_spread", dict(n_agents=64, n_landmarks=64), dict(A=64, L=64)),
                             ("simple_tag", {}, {}), ("simple_world_comm", {}, {}), ("simple", {}, {}), ("simple_adversary", {}, {}),
                             ("simple_push", {}, {}), ("simple_speaker_listener", {}, {}), ("simple_reference", {}, {}),
                             ("simple_crypto", {}, {})):
        p, o = make_spec(name, **kw_p), orc.make_spec(name, **kw_o)
        assert (p.n_agents, p.n_landmarks, p.dim_c) == (o.A, o.L, o.dim_c)
        for a, b in ((p.size, o.size), (p.mass, o.mass), (p.movable, o.movable), (p.collide, o.collide),
                     (p.sensitivity, o.sens), (p.max_speed, o.max_speed), (p.silent, o.silent),
                     (p.adversary, o.adversary), (p.leader, o.leader)):
            assert np.array_equal(np.asarray(a), np.asarray(b))
        assert (p.dt, p.damping, p.contact_force, p.contact_margin) == (o.dt, o.damping, o.contact_force, o.contact_margin)


def test_library_was_built_from_these_sources():
    """build() notes the hash of the kernel sources next to the library; a library left over from other sources
    would make every GPU test check the wrong code."""
    import bench
    from mpe_hip import _lib
    note = _lib.LIB_PATH + ".sources"
    if not os.path.exists(note):
        pytest.skip("library not built by __graft_entry__.build()")
    assert open(note).read().strip() == bench.source_hash(), "run `python __graft_entry__.py build`"


def test_the_product_has_no_cpu_path():
    """Without a GPU device the batch API refuses to start, and without the built library the package does not load:
    nothing falls back to a CPU implementation."""
    import subprocess
    import sys
    pkg = os.path.join(ROOT, "multiagent-particle-envs_amd")
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import torch, mpe_hip\n"
            "try:\n"
            "    mpe_hip.BatchedEnv('simple_spread', 8, device='cpu')\n"
            "except ValueError as e:\n"
            "    print('REFUSED', e)\n" % pkg)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "REFUSED" in r.stdout and "no CPU path" in r.stdout, r.stdout + r.stderr
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import mpe_hip\n"
            "from mpe_hip import _lib\n"
            "_lib.load()\n" % pkg)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, MPE_HIP_LIB="/nonexistent/libmpe_hip.so"))
    assert r.returncode != 0 and "no CPU fallback" in r.stderr, r.stderr[-1500:]
    # the product package never imports the oracle
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(".py"):
                assert "mpe_oracle" not in open(os.path.join(dirpath, fn)).read(), os.path.join(dirpath, fn)
