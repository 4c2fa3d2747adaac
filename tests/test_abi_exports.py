"""The C-ABI library loads and exports every entry point include/fpc_engine.h declares (no compute
calls, no GPU needed), and the POD layouts match the header."""
import ctypes as C
import os
import re

import fpc_ffi

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(REPO, "include", "fpc_engine.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fpc_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    names = _declared()
    assert len(names) >= 25
    lib = fpc_ffi.lib()                      # raises if the library is missing: no CPU fallback exists
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(fpc_ffi.EXPORTS)
    assert lib.fpc_abi_version() == 7


def test_pod_layouts():
    assert C.sizeof(fpc_ffi.Board) == 288 and C.sizeof(fpc_ffi.Move) == 8 and C.sizeof(fpc_ffi.Tuple) == 1280
    assert fpc_ffi.Tuple.n.offset == 198 and fpc_ffi.Tuple.z.offset == 200 and fpc_ffi.Tuple.flat.offset == 212 and fpc_ffi.Tuple.visits.offset == 724
    assert fpc_ffi.Board.pl.offset == 196 and fpc_ffi.Board.turn.offset == 196 + 64 + 12


def test_static_helpers_need_no_device():
    lib = fpc_ffi.lib()
    assert lib.fpc_action_space_size(14) == 23520 and lib.fpc_num_action_channels(8) == 72
    assert lib.fpc_is_legal_location(14, 3, 0, 0) == 0 and lib.fpc_is_legal_location(14, 3, 3, 0) == 1
    f, t = C.c_int(), C.c_int()
    assert lib.fpc_flat_to_move(8, 50, C.byref(f), C.byref(t)) == 0 and lib.fpc_move_flat_index(8, f.value, t.value) == 50


def test_engine_creation_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        return
    import pytest
    with pytest.raises(RuntimeError, match="no HIP device|fpc_create failed"):
        fpc_ffi.Engine(8, 2, max_games=1, max_sims=1)


def test_a_library_built_from_other_sources_is_refused(tmp_path, monkeypatch):
    """__graft_entry__.build() stores the content hash of the sources beside the .so; fpc_ffi refuses to load a
    library whose recorded hash is not the hash of the sources in the tree (mtimes say nothing after a checkout)."""
    import shutil
    import pytest
    monkeypatch.delenv("FPC_ENGINE_LIB", raising=False)
    fpc_ffi._check_fresh()                                   # the in-tree library is fresh
    fake = tmp_path / "libfpc_engine.so"
    shutil.copy(fpc_ffi.LIB_PATH, fake)
    (tmp_path / "libfpc_engine.so.srchash").write_text("0" * 64 + "\n")
    monkeypatch.setattr(fpc_ffi, "LIB_PATH", str(fake))
    with pytest.raises(RuntimeError, match="stale HIP engine library"):
        fpc_ffi._check_fresh()


def test_an_unloadable_rccl_library_is_an_error_message_not_a_crash():
    """FPC_RCCL_LIB naming a file that cannot be loaded must come back from fpc_comm_available() -- the
    probe tuples.init_comm runs on every rank before anybody enters ncclCommInitRank -- as FPC_ECOMM with a text
    (no silent substitute, and no segfault on the way: dlerror() reads as NULL the second time)."""
    import subprocess
    import sys
    code = ("import ctypes as C, sys\n"
            "L = C.CDLL(%r)\n"
            "L.fpc_comm_available.restype = C.c_int\n"
            "L.fpc_last_error.restype = C.c_char_p; L.fpc_last_error.argtypes = [C.c_void_p]\n"
            "rc = L.fpc_comm_available()\n"
            "msg = (L.fpc_last_error(None) or b'').decode()\n"
            "print(rc, '|', msg)\n"
            "rc2 = L.fpc_comm_available()\n"           # the verdict is cached: same answer, still no crash
            "sys.exit(0 if (rc == rc2 and rc < 0) else 3)\n" % fpc_ffi.LIB_PATH)
    env = dict(os.environ, FPC_RCCL_LIB="/nonexistent/librccl.so")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, (p.returncode, p.stdout, p.stderr)
    assert "FPC_RCCL_LIB=/nonexistent/librccl.so cannot be loaded" in p.stdout
    assert "cannot open shared object file" in p.stdout or "No such file" in p.stdout
