"""The Rust binding crate (bindings/rust, source only — no Rust toolchain here) declares the C ABI by
hand in an `extern "C"` block.  This test parses that block and include/sprsolve_hip.h and compares, per function:
name present in the header, arity, and per argument pointer-ness, const-ness of the pointee and the scalar class
(integer width / float width / opaque struct).  It catches the drift a compiler would catch."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

C_SCALAR = {"int": "i32", "int32_t": "i32", "int64_t": "i64", "size_t": "usize", "double": "f64", "float": "f32",
            "void": "void", "char": "char", "sprs_c64": "c64", "sprs_c32": "c32"}
RUST_SCALAR = {"c_int": "i32", "i32": "i32", "i64": "i64", "usize": "usize", "f64": "f64", "f32": "f32", "c_void": "void",
               "c_char": "char", "Complex64": "c64", "Complex32": "c32"}


def _norm_c(arg):
    """'const sprs_c64 *x_host' -> (depth, const_pointee, base)"""
    arg = arg.strip()
    depth = arg.count("*")
    const = bool(re.match(r"const\b", arg))
    toks = re.sub(r"\bconst\b", " ", arg.replace("*", " ")).split()
    base = toks[0]
    if base == "struct":
        base = toks[1]
    return depth, const if depth else False, C_SCALAR.get(base, base)


def _norm_rust(ty):
    """'*const Complex64' / '*mut *mut sprs_csr' / 'usize' -> (depth, const_pointee, base)"""
    ty = ty.strip()
    depth = len(re.findall(r"\*(?:const|mut)\b", ty))
    first = re.match(r"\*(const|mut)\b", ty)
    base = re.sub(r"\*(?:const|mut)\s*", "", ty).strip()
    return depth, bool(first and first.group(1) == "const" and depth == 1), RUST_SCALAR.get(base, base)


def c_prototypes():
    src = open(os.path.join(ROOT, "include", "sprsolve_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w \*]*?)\b(sprs_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3)
        al = [] if args.strip() in ("", "void") else [_norm_c(a) for a in args.split(",")]
        protos[name] = (_norm_c(ret + " r") if "*" in ret else (0, False, C_SCALAR.get(ret.split()[-1], ret)), al)
    return protos


def rust_prototypes():
    src = open(os.path.join(ROOT, "bindings", "rust", "src", "lib.rs")).read()
    blk = re.search(r'extern\s+"C"\s*\{(.*?)\n    \}', src, flags=re.S).group(1)
    protos = {}
    for m in re.finditer(r"pub fn (sprs_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", blk, flags=re.S):
        name, args, ret = m.group(1), m.group(2), (m.group(3) or "()").strip()
        al = [_norm_rust(a.split(":", 1)[1]) for a in args.split(",") if ":" in a]
        protos[name] = (_norm_rust(ret), al)
    return protos


def test_rust_extern_block_matches_header():
    c, r = c_prototypes(), rust_prototypes()
    assert len(r) >= 70, "the extern block was not parsed"
    problems = []
    for name, (rret, rargs) in sorted(r.items()):
        if name not in c:
            problems.append("%s: not declared in include/sprsolve_hip.h" % name)
            continue
        cret, cargs = c[name]
        if len(cargs) != len(rargs):
            problems.append("%s: %d C arguments, %d Rust arguments" % (name, len(cargs), len(rargs)))
            continue
        if (cret[0], cret[2]) != (rret[0], rret[2]):
            problems.append("%s: return %r vs %r" % (name, cret, rret))
        for i, (ca, ra) in enumerate(zip(cargs, rargs)):
            if ca[0] != ra[0]:
                problems.append("%s arg %d: pointer depth %d (C) vs %d (Rust)" % (name, i, ca[0], ra[0]))
            elif ca[2] != ra[2]:
                problems.append("%s arg %d: %s (C) vs %s (Rust)" % (name, i, ca[2], ra[2]))
            elif ca[0] == 1 and ca[1] != ra[1]:
                problems.append("%s arg %d: pointee const-ness differs (C const=%s, Rust const=%s)" % (name, i, ca[1], ra[1]))
    assert not problems, "\n".join(problems)


def test_status_constants_match_header():
    hdr = open(os.path.join(ROOT, "include", "sprsolve_hip.h")).read()
    rs = open(os.path.join(ROOT, "bindings", "rust", "src", "lib.rs")).read()
    cvals = dict((k, int(v)) for k, v in re.findall(r"\b(SPRS_[A-Z_]+)\s*=\s*(\d+)", hdr))
    rvals = dict((k, int(v)) for k, v in re.findall(r"pub const (SPRS_[A-Z_]+): c_int = (\d+);", rs))
    assert rvals and all(cvals.get(k) == v for k, v in rvals.items()), {k: (cvals.get(k), v) for k, v in rvals.items() if cvals.get(k) != v}


def _rust_src():
    return open(os.path.join(ROOT, "bindings", "rust", "src", "lib.rs")).read()


def test_every_symbol_the_wrappers_use_is_declared():
    """A `sys::sprs_*` call (directly or as an argument of the per-scalar dispatch macro) whose symbol is missing from the
    extern block would be a compile error; one missing from the header would be a link error."""
    src = _rust_src()
    declared = set(rust_prototypes())
    header = set(c_prototypes())
    body = src[src.index("pub fn default_ctx"):]
    used = set(re.findall(r"sys::(sprs_[a-z0-9_]+)\s*\(", body))
    for inv in re.findall(r"impl_scalar!\((.*?)\);", body, flags=re.S):
        used |= set(re.findall(r"\b(sprs_[a-z0-9_]+)\b", inv))
    assert len(used) >= 70, len(used)
    assert not (used - declared), "used but not in the extern block: %s" % sorted(used - declared)
    assert not (used - header), "used but not in include/sprsolve_hip.h: %s" % sorted(used - header)
    assert not (declared - used), "declared in the extern block but never used by a wrapper: %s" % sorted(declared - used)


def test_every_d_symbol_has_its_z_sibling():
    """The reference is generic over cauchy::Scalar (f64 and Complex<f64> on the north_star path): whatever the binding
    binds for f64 it must bind for Complex64 too — the four complex integration tests of the reference
    (tests/test_complex_solve.rs:3-88, tests/test_complex_solve2.rs:4-28) need exactly those."""
    r = set(rust_prototypes())
    missing = [n for n in sorted(r) if n.endswith("_d") and n[:-2] + "_z" not in r and not n.startswith("sprs_gauss_seidel")]
    assert not missing, missing          # (Gauss-Seidel is real-only in the library: src/gauss_seidel.rs solves f64 / f32)
    # the two-type preconditioner constructors (precond.rs:6-12: V real while T complex)
    assert {"sprs_diag_precond_create_d", "sprs_diag_precond_create_zd", "sprs_diag_precond_create_z", "sprs_axpy_zd"} <= r


def test_every_f64_symbol_has_its_f32_sibling():
    """The reference is generic over all four cauchy::Scalar types and tests f32 / Complex<f32> BLAS-1 itself
    (src/vecalg.rs:647-658,669-677,771-830): every `_d` wrapper needs its `_s` sibling, every `_z` its `_c`, `_zd` its `_cs`,
    and the generic wrappers must not pin `Real = f64`."""
    r = set(rust_prototypes())
    missing = [n for n in sorted(r) if n.endswith("_d") and n[:-2] + "_s" not in r]
    missing += [n for n in sorted(r) if n.endswith("_z") and n[:-2] + "_c" not in r]
    missing += [n for n in sorted(r) if n.endswith("_zd") and n[:-3] + "_cs" not in r]
    assert not missing, missing
    src = _rust_src()
    assert "Scalar<Real = f64>" not in src
    for needle in ("impl_scalar!(f32,", "impl_scalar!(Complex32,", "impl HipDiag<f32> for f32", "impl HipDiag<f32> for Complex32",
                   "impl HipDiag<Complex32> for Complex32", "impl HipGsScalar for f32", "-> SolveResult<(usize, T::Real)>",
                   "pub fn norm2<T: HipScalar>(x: &DevVec<T>) -> T::Real"):
        assert needle in src, needle


def test_wrappers_cover_the_reference_test_surface():
    src = _rust_src()
    for needle in ("pub struct HipBiCGStab<'data, T: HipScalar>", "pub struct HipMinRes<'data, T: HipScalar>",
                   "pub struct HipCSMinRes<'data, T: HipScalar>", "pub struct HipDiagPrecond<T: HipDiag<V>, V>",
                   "impl HipDiag<f64> for f64", "impl HipDiag<f64> for Complex64", "impl HipDiag<Complex64> for Complex64",
                   "impl_index8!(usize)", "impl HipIndex for i32", "impl HipIndex for u32", "pub struct DevVec<T: HipScalar>",
                   "pub fn mul_vec_dev(&self, v_in: &DevVec<T>, v_out: &mut DevVec<T>)"):
        assert needle in src, needle
    for fn in ("dot", "conj_dot", "norm2", "scale", "rscale", "conj", "axpy", "axpby"):          # src/vecalg.rs:24-144
        assert re.search(r"pub fn %s<T: HipScalar>" % fn, src), fn
    assert src.count("pub fn precond_solve<V>") == 2
    ex = open(os.path.join(ROOT, "bindings", "rust", "examples", "host_loop_bicgstab.rs")).read()
    for call in ("a.mul_vec_dev(", "vecalg::axpby(", "vecalg::conj_dot(", "vecalg::norm2(", "copy_from("):
        assert call in ex, call
