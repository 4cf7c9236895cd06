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
    assert len(r) >= 35, "the extern block was not parsed"
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
