"""CPU: the Rust side of the boundary cannot be compiled in this image (no cargo / rustc), so it is checked as text.

 * rust_shim/apds_sys/src/lib.rs declares EVERY function of include/apds.h with the same name, arity, per-argument pointer depth,
   pointee constness and scalar type, and the same return type (two independent little parsers, one for the C header, one for the
   Rust file - the generator tools/gen_apds_sys.py is not imported here);
 * the committed file is what the generator gives for the committed header (no drift);
 * every `apds_sys::name(...)` call in the shim crates and every binding INTEGRATION.md cites exists, with the arity the header states;
 * the struct layouts the two sides share have the fields the header has.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "apds.h")
SYS = os.path.join(ROOT, "rust_shim", "apds_sys", "src", "lib.rs")

C_SCALARS = {"void": "void", "char": "char", "int": "i32", "float": "f32", "double": "f64", "size_t": "usize", "uint8_t": "u8", "int32_t": "i32",
             "uint32_t": "u32", "int64_t": "i64", "uint64_t": "u64"}
RUST_SCALARS = {"c_void": "void", "c_char": "char", "c_int": "i32", "c_float": "f32", "c_double": "f64", "f32": "f32", "f64": "f64", "usize": "usize", "u8": "u8",
                "i32": "i32", "u32": "u32", "i64": "i64", "u64": "u64"}


def c_type(t):
    """C type text -> (scalar, [pointee-is-const per pointer level, outermost first])."""
    toks = re.findall(r"\*|const|[A-Za-z_][A-Za-z0-9_]*", t)
    base = [x for x in toks if x not in ("*", "const", "struct")]
    assert len(base) == 1, t
    scalar = C_SCALARS.get(base[0], base[0])
    # walk left to right: 'const' before the first '*' (or right after the base) qualifies the base; 'const' after the i-th '*' qualifies that pointer
    quals, cur = [], False
    for x in toks:
        if x == "const":
            cur = True
        elif x == "*":
            quals.append(cur)
            cur = False
    # quals[i] = constness of what pointer level i (innermost first) points TO; outermost first for comparison
    return scalar, list(reversed(quals))


def rust_type(t):
    t = t.strip()
    quals = []
    while True:
        m = re.match(r"^\*(const|mut)\s+(.*)$", t)
        if not m:
            break
        quals.append(m.group(1) == "const")
        t = m.group(2).strip()
    return RUST_SCALARS.get(t, t), quals


def split_top(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [x.strip() for x in out]


def header_functions():
    text = re.sub(r"/\*.*?\*/", " ", open(HEADER).read(), flags=re.S)
    text = text[text.index('extern "C" {'):]
    out = {}
    for m in re.finditer(r"(?m)^([A-Za-z_][A-Za-z0-9_ ]*?[\s\*]+)(apds_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        params = []
        if args not in ("", "void"):
            for a in split_top(args):
                mm = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)$", a)
                params.append(c_type(mm.group(1)))
        out[name] = (c_type(ret) if ret != "void" else None, params)
    return out


def rust_functions():
    text = open(SYS).read()
    block = text[text.index('extern "C" {'):]
    out = {}
    for m in re.finditer(r"pub fn (apds_[a-z0-9_]+)\s*\(([^;]*?)\)\s*(?:->\s*([^;]+?))?\s*;", block, flags=re.S):
        name, args, ret = m.group(1), " ".join(m.group(2).split()), m.group(3)
        params = [rust_type(a.split(":", 1)[1]) for a in split_top(args)] if args.strip() else []
        out[name] = (rust_type(ret) if ret else None, params)
    return out


def test_every_header_function_is_bound_with_the_same_signature():
    hf, rf = header_functions(), rust_functions()
    assert len(hf) >= 70, len(hf)
    assert set(hf) == set(rf), sorted(set(hf) ^ set(rf))
    for name, (ret, params) in hf.items():
        rret, rparams = rf[name]
        assert ret == rret, (name, "return", ret, rret)
        assert len(params) == len(rparams), (name, "arity", len(params), len(rparams))
        for i, (a, b) in enumerate(zip(params, rparams)):
            assert a == b, (name, f"argument {i}", a, b)


def test_committed_bindings_are_what_the_generator_gives():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_apds_sys.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def _call_sites(path):
    text = open(path).read()
    for m in re.finditer(r"apds_sys::(apds_[a-z0-9_]+)\s*\(", text):
        depth, i = 1, m.end()
        while depth and i < len(text):
            depth += text[i] in "([{"
            depth -= text[i] in ")]}"
            i += 1
        yield m.group(1), text[m.end():i - 1]


def test_shim_crates_and_integration_doc_only_use_existing_bindings():
    rf = rust_functions()
    shim_files = []
    for d, _, files in os.walk(os.path.join(ROOT, "rust_shim")):
        shim_files += [os.path.join(d, f) for f in files if f.endswith(".rs") and "apds_sys" not in d]
    assert len(shim_files) >= 4
    used = set()
    for path in shim_files:
        for name, args in _call_sites(path):
            assert name in rf, (path, name)
            assert len(split_top(args)) == len(rf[name][1]), (path, name, "arity", len(split_top(args)), len(rf[name][1]))
            used.add(name)
    # the crate surface the reference exposes is forwarded: extraction, both matchers, the point gather, homography, raster_to_mat, warp, PnP
    for name in ("apds_akaze_extract", "apds_get_knn_matches", "apds_get_bruteforce_matches", "apds_get_points_from_matches", "apds_find_homography",
                 "apds_raster_to_mat", "apds_warp_perspective", "apds_pnp_solver_ransac", "apds_tile_extract_batch", "apds_set_device", "apds_shard_knn", "apds_pipeline_create", "apds_pipeline_submit", "apds_pipeline_poll"):
        assert name in used, name
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for name in set(re.findall(r"apds_sys::(apds_[a-z0-9_]+)", doc)):
        assert name in rf, ("INTEGRATION.md cites a binding that does not exist", name)
    for name, args in _call_sites(os.path.join(ROOT, "INTEGRATION.md")):
        assert len(split_top(args)) == len(rf[name][1]), ("INTEGRATION.md", name, "arity")


def test_shared_struct_layouts():
    text = open(SYS).read()
    kp = re.search(r"pub struct apds_keypoint \{(.*?)\}", text, flags=re.S).group(1)
    assert re.findall(r"pub (\w+): (\w+)", kp) == [("x", "f32"), ("y", "f32"), ("size", "f32"), ("angle", "f32"), ("response", "f32"), ("octave", "i32"),
                                                   ("class_id", "i32")]
    dm = re.search(r"pub struct apds_dmatch \{(.*?)\}", text, flags=re.S).group(1)
    assert re.findall(r"pub (\w+): (\w+)", dm) == [("query_idx", "i32"), ("train_idx", "i32"), ("img_idx", "i32"), ("distance", "f32")]
    assert "pub bytes: [c_char; APDS_COMM_ID_BYTES]" in text and "pub const APDS_COMM_ID_BYTES: usize = 128;" in text
    ht = re.search(r"pub struct apds_host_transport \{(.*?)\n\}", text, flags=re.S).group(1)
    assert ht.index("user") < ht.index("all_gather") < ht.index("all_to_all")
    # the pipeline's plain-data structs: the same field names in the same order as the header (this test's own parse of both texts)
    hdr = re.sub(r"/\*.*?\*/", " ", open(HEADER).read(), flags=re.S)
    for name in ("apds_pipeline_params", "apds_frame_result", "apds_pipeline_counters"):
        cbody = re.search(r"typedef\s+struct\s+%s\s*\{(.*?)\}\s*%s\s*;" % (name, name), hdr, flags=re.S).group(1)
        cnames = []
        for decl in cbody.split(";"):
            decl = decl.strip()
            if decl:
                first, *rest = decl.split(",")
                cnames.append(re.findall(r"[A-Za-z_][A-Za-z0-9_]*", first)[-1] if "[" not in first else re.findall(r"([A-Za-z_][A-Za-z0-9_]*)\s*\[", first)[-1])
                cnames += [re.findall(r"[A-Za-z_][A-Za-z0-9_]*", r)[0] for r in rest]
        rbody = re.search(r"pub struct %s \{(.*?)\n\}" % name, text, flags=re.S).group(1)
        assert re.findall(r"pub (\w+):", rbody) == cnames, (name, cnames)
