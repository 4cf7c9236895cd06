"""CPU-only checks of the drop-in boundary: the C-ABI library builds, loads, and exports every
symbol include/sprsolve_hip.h declares; the host-side mirror validates arguments; calling into
the library without a GPU fails loudly with a status code (no CPU fallback exists)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from sprsolve_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.lib()


def _declared():
    src = open(os.path.join(ROOT, "include", "sprsolve_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sprs_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(L):
    names = _declared()
    assert len(names) > 80
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_binding_covers_header(L):
    from sprsolve_amd import _lib
    assert sorted(_lib.all_symbols()) == _declared()


def test_no_oracle_in_product():
    """The product path must never import, link or call the oracle."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "sprsolve_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in txt.lower().replace("no cpu fallback", ""), os.path.join(dirpath, f)
    txt = open(os.path.join(ROOT, "include", "sprsolve_hip.h")).read()
    assert "oracle" not in txt.lower()


def test_status_strings(L):
    assert L.sprs_status_str(0) == b"Ok"
    assert b"Dimension mismatch" in L.sprs_status_str(6)
    assert L.sprs_version() >= 100


def test_fails_loudly_without_gpu(L):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    st = L.sprs_ctx_create(0, None, C.byref(h))
    assert st == 102 and not h.value          # SPRS_ERR_NO_DEVICE, never a silent CPU path
    from sprsolve_amd import Context
    from sprsolve_amd.error import BackendError
    with pytest.raises(BackendError):
        Context(0)


def test_null_handles_are_rejected(L):
    assert L.sprs_csr_destroy(None) == 0
    assert L.sprs_bicgstab_destroy(None) == 0
    assert L.sprs_ctx_destroy(None) == 0
    assert L.sprs_csr_rows(None) == -1
    out = C.c_void_p()
    assert L.sprs_bicgstab_create_d(None, 4, C.byref(out)) == 7
    assert L.sprs_mul_vec_dev_d(None, None, None) == 7


def test_header_is_plain_c(tmp_path):
    """The boundary is a C ABI: include/sprsolve_hip.h must compile as strict C99 (and as C++), and a C program must
    link against the library with nothing but the header."""
    import subprocess
    src = tmp_path / "use_header.c"
    src.write_text('#include "sprsolve_hip.h"\n'
                   'int main(void) {\n'
                   '    sprs_ctx *ctx = 0; sprs_csr *a = 0; int no = 0, np = 0; size_t its = 0; double res = 0;\n'
                   '    if (sprs_version() <= 0) return 2;\n'
                   '    if (sprs_csr_stream_format(a, &no, &np) != -1) return 3;      /* null handle */\n'
                   '    if (sprs_csr_destroy(a) != SPRS_OK || sprs_ctx_destroy(ctx) != SPRS_OK) return 4;\n'
                   '    (void)its; (void)res;\n'
                   '    return sprs_status_str(SPRS_BREAKDOWN) ? 0 : 5;\n'
                   '}\n')
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", inc, "-c", str(src), "-o", str(tmp_path / "a.o")])
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I", inc, "-x", "c++", "-c", str(src), "-o", str(tmp_path / "b.o")])
    libdir = os.path.join(ROOT, "sprsolve_amd")
    exe = tmp_path / "use_header"
    subprocess.check_call(["gcc", str(tmp_path / "a.o"), "-o", str(exe), "-L", libdir, "-l:libsprsolve_hip.so",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    assert subprocess.call([str(exe)]) == 0
