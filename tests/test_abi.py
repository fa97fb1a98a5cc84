"""The C-ABI library loads, exports every symbol include/trl_hip.h declares, and
the ctypes descriptor structs match the C layout.  CPU only (no compute calls)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, "include", "trl_hip.h")


@pytest.fixture(scope="module")
def built_lib():
    sys.path.insert(0, REPO)
    from torchrl_amd import build
    return build.build(verbose=False)


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(trl_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(built_lib):
    handle = ctypes.CDLL(built_lib)
    syms = declared_symbols()
    assert len(syms) >= 14
    for name in syms:
        assert hasattr(handle, name), "libtrl_hip.so does not export %s" % name


def test_python_binding_covers_header(built_lib):
    from torchrl_amd import _C
    assert sorted(_C.SIGNATURES) == declared_symbols()
    assert _C.lib().trl_abi_version() == 1


def test_descriptor_struct_layouts_match_c(tmp_path, built_lib):
    """Compile a probe with gcc that prints sizeof/offsetof, compare with ctypes."""
    from torchrl_amd import _C
    structs = {"trl_rollout_t": _C.RolloutArgs, "trl_ppo_batch_t": _C.PpoBatchArgs, "trl_adam_t": _C.AdamArgs}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "trl_hip.h"', 'int main(void){']
    for cname, cls in structs.items():
        lines.append('printf("%s size %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines.append("return 0;}")
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-I", os.path.join(REPO, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split("\n")
    got = {}
    for ln in out:
        if ln.strip():
            a, b, c = ln.split()
            got[(a, b)] = int(c)
    for cname, cls in structs.items():
        assert got[(cname, "size")] == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert got[(cname, fname)] == getattr(cls, fname).offset, (cname, fname)


def test_missing_library_fails_loudly(monkeypatch, built_lib):
    from torchrl_amd import _C
    monkeypatch.setattr(_C, "_lib", None)
    monkeypatch.setattr(_C, "LIB_PATH", "/nonexistent/libtrl_hip.so")
    with pytest.raises(_C.TrlError, match="no CPU fallback"):
        _C.lib()


def test_cpu_tensor_is_rejected(built_lib):
    import torch
    from torchrl_amd import _C
    with pytest.raises(_C.TrlError, match="no CPU path"):
        _C.dev_ptr(torch.zeros(4), name="x")
