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
    from vla_adapter_amd import native
    txt = open(os.path.join(ROOT, "include", "vla_native.h")).read()
    assert lib.vla_version() == native.ABI_VERSION == int(re.search(r"#define VLA_ABI_VERSION (\d+)", txt).group(1))
    # the library reports the descriptor sizes it was compiled with; the binding refuses to load on a mismatch (native.load)
    for which, cls in enumerate((native.GemmDesc, native.AttnDesc, native.HeadAttnDesc, native.GemmTnDesc)):
        assert lib.vla_desc_size(which) == ctypes.sizeof(cls), cls.__name__
    assert lib.vla_desc_size(99) == -1


def test_integration_stub_is_current():
    """INTEGRATION.md's ctypes stub is generated from native.py (tools/gen_integration_stub.py); a descriptor that grew without
    the published stub following it would make a maintainer pass a short struct."""
    import subprocess
    import sys
    assert subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_integration_stub.py"), "--check"]).returncode == 0, \
        "INTEGRATION.md stub is stale: run tools/gen_integration_stub.py"


def test_gemm256_routing_predicate_is_host_arithmetic(lib):
    """vla_gemm256_extent_ok: the 256-row kernel's 32-bit per-lane byte offsets must cover every operand row: step
    shapes pass, operands of 4 GiB and more (token-CE lm_head at large batch, row groups with a huge stride) are routed away."""
    from vla_adapter_amd import native
    d = native.GemmDesc()
    d.M, d.N, d.K, d.lda, d.ldb, d.batch = 11264, 9728, 896, 896, 896, 1
    assert lib.vla_gemm256_extent_ok(ctypes.byref(d)) == 1
    d.M, d.lda = 1 << 20, 4096                      # 1 M rows x 8 KiB = 8 GiB of A
    assert lib.vla_gemm256_extent_ok(ctypes.byref(d)) == 0
    d.M, d.lda, d.N, d.ldb = 4096, 896, 151936, 16384      # B: 151936 rows x 32 KiB
    assert lib.vla_gemm256_extent_ok(ctypes.byref(d)) == 0
    d.N, d.ldb = 151936, 896                        # the real lm_head: 272 MB
    assert lib.vla_gemm256_extent_ok(ctypes.byref(d)) == 1
    d.a_group, d.a_group_stride, d.M = 256, 1 << 24, 256 * 200      # row groups 32 MiB apart, 200 of them: 6.4 GiB span
    assert lib.vla_gemm256_extent_ok(ctypes.byref(d)) == 0
    d.fp8 = 1                                       # one byte per element halves every extent
    d.a_group_stride = 1 << 23
    assert lib.vla_gemm256_extent_ok(ctypes.byref(d)) == 1


def test_descriptor_layouts_match_header(lib, tmp_path):
    """ctypes mirrors of the descriptor structs must match what a C compiler makes of include/vla_native.h:
    same size and the same offset for every field (checked by compiling a probe with gcc)."""
    import subprocess
    from vla_adapter_amd import native
    structs = {"vla_gemm_desc": native.GemmDesc, "vla_attn_desc": native.AttnDesc, "vla_head_attn_desc": native.HeadAttnDesc,
               "vla_gemm_tn_desc": native.GemmTnDesc}
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
    t = native.GemmTnDesc()
    assert lib.vla_gemm_bf16_tn(None, ctypes.byref(t)) == -1 and b"null" in lib.vla_last_error()
    t.A = t.B = t.C = 4096
    t.M, t.N1, t.N2, t.lda, t.ldb, t.ldc, t.batch = 100, 12, 8, 16, 8, 8, 1
    assert lib.vla_gemm_bf16_tn(None, ctypes.byref(t)) == -1 and b"multiples of 8" in lib.vla_last_error()
    t.N1, t.a_group, t.a_group_stride = 16, 100, 4096
    assert lib.vla_gemm_bf16_tn(None, ctypes.byref(t)) == -1 and b"multiples of 64 rows" in lib.vla_last_error()


def test_missing_library_fails_loudly(monkeypatch):
    from vla_adapter_amd import native
    monkeypatch.setattr(native, "_lib", None)
    monkeypatch.setattr(native, "LIB_PATH", "/nonexistent/libvla_native.so")
    with pytest.raises(native.NativeLibraryMissing):
        native.load()
