"""bindings/rust is UNVERIFIED source (no Rust toolchain in the build image).  What a text check can establish: ffi.rs declares
exactly the functions include/kryst_hip.h declares, with the header's argument counts, and its #[repr(C)] structs list the
header's fields in the header's order; lib.rs implements kryst's traits for the types INTEGRATION.md promises."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _strip_c(txt):
    return re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", txt, flags=re.S))


def _split_args(s):
    """top-level comma split (function-pointer arguments carry parentheses)"""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "(<[":
            depth += 1
        elif ch in ")>]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return [a for a in out if a and a != "void"]


def header_functions():
    txt = _strip_c(open(os.path.join(ROOT, "include", "kryst_hip.h")).read())
    tail = re.search(r"#define KRYST_SOLVE_ARGS(.*?)\n\n", txt, flags=re.S).group(1).replace("\\\n", " ")
    txt = txt.replace("KRYST_SOLVE_ARGS)", tail.strip() + ")")
    fns = {}
    for m in re.finditer(r"\b(kryst_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.S):
        fns[m.group(1)] = len(_split_args(m.group(2)))
    return fns


def rust_functions():
    txt = re.sub(r"//[^\n]*", "", open(os.path.join(ROOT, "bindings", "rust", "src", "ffi.rs")).read())
    fns = {}
    for m in re.finditer(r"pub fn (kryst_[a-z0-9_]+)\s*\(([^;]*?)\)\s*(?:->\s*[^;]+)?;", txt, flags=re.S):
        fns[m.group(1)] = len(_split_args(m.group(2)))
    # the macro-declared solve entry points: (names ; host) take 12 arguments, (names ; dev) 11
    for m in re.finditer(r"solve_fn!\(([^;]*);\s*(host|dev)\)", txt, flags=re.S):
        for name in re.findall(r"kryst_[a-z0-9_]+", m.group(1)):
            fns[name] = 12 if m.group(2) == "host" else 11
    fns.pop("kryst_", None)
    return fns


def test_ffi_declares_the_whole_header_with_matching_arity():
    h, r = header_functions(), rust_functions()
    assert len(h) >= 70
    assert set(h) == set(r), (sorted(set(h) - set(r)), sorted(set(r) - set(h)))
    assert {k: v for k, v in h.items() if r[k] != v} == {}, {k: (v, r[k]) for k, v in h.items() if r[k] != v}


def test_repr_c_structs_follow_the_header():
    h = _strip_c(open(os.path.join(ROOT, "include", "kryst_hip.h")).read())
    r = open(os.path.join(ROOT, "bindings", "rust", "src", "ffi.rs")).read()
    for cname, rname in (("kryst_params_t", "Params"), ("kryst_stats_t", "Stats")):
        body = re.search(r"typedef struct \{([^{}]*)\}\s*" + cname, h).group(1)
        cfields = [(t.replace("int64_t", "i64").replace("int32_t", "i32").replace("double", "f64"), n.strip())
                   for decl in body.split(";") if decl.strip()
                   for t, names in [re.match(r"\s*(\w+)\s+(.*)", decl.strip(), flags=re.S).groups()]
                   for n in names.split(",")]
        rbody = re.search(r"#\[repr\(C\)\][^{]*pub struct " + rname + r"\s*\{(.*?)\}", r, flags=re.S).group(1)
        rfields = [(t.strip(), n) for n, t in re.findall(r"pub (\w+):\s*([^,\n]+)", rbody)]
        assert cfields == rfields, (cname, cfields, rfields)


def test_lib_implements_the_reference_traits():
    src = open(os.path.join(ROOT, "bindings", "rust", "src", "lib.rs")).read()
    assert "impl MatVec<Vec<f64>> for HipCsrMatrix" in src                                   # src/core/traits.rs:4-7
    for s in ("HipGmresSolver", "HipBiCgStabSolver"):
        assert f"impl LinearSolver<HipCsrMatrix, Vec<f64>> for {s}" in src                     # src/solver/mod.rs:30-52
    assert "impl LinearSolver<HipCsrMatrix, Vec<f64>> for $name" in src and "HipCgSolver, ffi::kryst_cg_solve" in src \
        and "HipPcgSolver, ffi::kryst_pcg_solve" in src
    assert "impl Preconditioner<HipCsrMatrix, Vec<f64>> for $name" in src                      # src/preconditioner/mod.rs:8-13
    for pc in ("HipJacobi", "HipIlu0", "HipChebyshev"):
        assert re.search(r"device_pc!\s*\{[^}]*?" + pc, src, flags=re.S), pc
    for builder in ("with_norm", "with_single_reduction", "with_radius", "with_obj_target", "with_monitor", "clear_history",
                    "with_preconditioning"):
        assert f"pub fn {builder}" in src, builder                                             # cg.rs:64-93, gmres.rs:55-60
    assert "KError::ZeroPivot(unsafe { ffi::kryst_hip_last_error_row() }" in src              # src/error.rs:15-16
