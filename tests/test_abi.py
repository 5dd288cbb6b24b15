"""CPU-side checks of the drop-in boundary: the shared library loads and exports every symbol that
include/vla_native.h declares (no compute calls here: there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "vla_native.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vla_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def lib():
    from vla_adapter_amd import native
    if not os.path.exists(native.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return native.load()


def test_header_and_binding_agree():
    from vla_adapter_amd import native
    assert header_symbols() == native.ABI_SYMBOLS


def test_library_exports_every_declared_symbol(lib):
    for name in header_symbols():
        assert hasattr(lib, name), f"libvla_native.so does not export {name}"
    assert lib.vla_version() == 1


def test_descriptor_layouts_match_header(lib, tmp_path):
    """ctypes mirrors of the descriptor structs must match what a C compiler makes of include/vla_native.h:
    same size and the same offset for every field (checked by compiling a probe with gcc)."""
    import subprocess
    from vla_adapter_amd import native
    structs = {"vla_gemm_desc": native.GemmDesc, "vla_attn_desc": native.AttnDesc, "vla_head_attn_desc": native.HeadAttnDesc}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{os.path.join(ROOT, "include", "vla_native.h")}"', "int main(void) {"]
    for cname, cls in structs.items():
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ["  return 0;", "}"]
    src, exe = tmp_path / "probe.c", tmp_path / "probe"
    src.write_text("\n".join(lines))
    subprocess.run(["gcc", str(src), "-o", str(exe)], check=True)
    got = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"{cname}.{fname}"


def test_argument_validation_without_gpu(lib):
    """Host-side validation rejects bad descriptors before any launch (safe on a CPU-only box)."""
    from vla_adapter_amd import native
    d = native.GemmDesc()
    assert lib.vla_gemm_bf16_nt(None, ctypes.byref(d)) == -1
    assert b"null" in lib.vla_last_error()
    d.A = d.B = d.C = 4096
    d.M, d.N, d.K, d.lda, d.ldb, d.ldc, d.batch = 8, 8, 100, 104, 104, 8, 1
    assert lib.vla_gemm_bf16_nt(None, ctypes.byref(d)) == -1
    assert b"multiple of 64" in lib.vla_last_error()


def test_missing_library_fails_loudly(monkeypatch):
    from vla_adapter_amd import native
    monkeypatch.setattr(native, "_lib", None)
    monkeypatch.setattr(native, "LIB_PATH", "/nonexistent/libvla_native.so")
    with pytest.raises(native.NativeLibraryMissing):
        native.load()
