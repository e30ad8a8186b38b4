#!/usr/bin/env python3
"""Writes rust_shim/apds_sys/src/lib.rs from include/apds.h: every function of the C ABI as a Rust `extern "C"` declaration, the two
POD structs, the communicator id, the host-transport callback table and the status / enum constants.

    python tools/gen_apds_sys.py            # rewrite the file
    python tools/gen_apds_sys.py --check    # exit 1 if the committed file differs from what the header gives

There is no Rust toolchain in the build image, so the bindings cannot be compiled here; they are generated instead of typed, and
tests/test_rust_bindings.py re-parses BOTH texts independently (its own C and Rust parsers) and compares every signature."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "apds.h")
OUT = os.path.join(ROOT, "rust_shim", "apds_sys", "src", "lib.rs")

BASE = {"void": "c_void", "char": "c_char", "int": "c_int", "float": "f32", "double": "f64", "size_t": "usize", "uint8_t": "u8", "int32_t": "i32",
        "uint32_t": "u32", "int64_t": "i64", "uint64_t": "u64", "apds_keypoint": "apds_keypoint", "apds_dmatch": "apds_dmatch",
        "apds_comm_id": "apds_comm_id", "apds_host_transport": "apds_host_transport", "apds_pipeline_params": "apds_pipeline_params",
        "apds_frame_result": "apds_frame_result", "apds_pipeline_counters": "apds_pipeline_counters", "apds_device_transport": "apds_device_transport"}
POD_STRUCTS = ["apds_pipeline_params", "apds_frame_result", "apds_pipeline_counters"]     # plain-data structs translated field by field
RESERVED = {"type", "match", "ref", "box", "move", "in", "fn", "loop", "mod", "use", "where", "impl", "self", "super", "crate", "dyn", "as"}


def strip_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def rust_type(ctype):
    """'const float* const*' -> '*const *const f32' (pointer levels read right to left)."""
    t = ctype.strip()
    levels = []
    while True:
        m = re.match(r"^(.*?)\*\s*(const)?\s*$", t)
        if not m:
            break
        levels.append("const" if m.group(2) else "mut")     # constness of the POINTER itself (ignored by Rust raw pointers)
        t = m.group(1).strip()
    base_const = bool(re.search(r"\bconst\b", t))
    base = re.sub(r"\bconst\b|\bstruct\b", "", t).strip()
    if base not in BASE:
        raise ValueError(f"unknown C type '{ctype}'")
    out = BASE[base]
    # innermost pointer's pointee constness = base_const; an outer pointer's pointee is the inner pointer, const iff the inner level said const
    n = len(levels)
    for i in range(n - 1, -1, -1):
        pointee_const = base_const if i == n - 1 else levels[i + 1] == "const"
        out = ("*const " if pointee_const else "*mut ") + out
    return out


def split_args(arglist):
    arglist = arglist.strip()
    if arglist in ("", "void"):
        return []
    out = []
    for a in arglist.split(","):
        a = a.strip()
        m = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)$", a)
        ctype, name = m.group(1).strip(), m.group(2)
        out.append((ctype, name))
    return out


def parse_header(text):
    text = strip_comments(text)
    body = text[text.index('extern "C" {') + len('extern "C" {'):]
    funcs = []
    for m in re.finditer(r"(?m)^((?:const\s+)?[A-Za-z_][A-Za-z0-9_]*(?:\s*\*+)?)\s*(apds_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", body):
        funcs.append((m.group(1).strip(), m.group(2), split_args(" ".join(m.group(3).split()))))
    defines = re.findall(r"(?m)^#define\s+(APDS_[A-Z0-9_]+)\s+(\(?-?[0-9]+\)?)\s*$", text)
    enums = []
    for m in re.finditer(r"enum\s*\{([^}]*)\}", text):
        for item in m.group(1).split(","):
            if "=" in item:
                k, v = item.split("=")
                enums.append((k.strip(), v.strip()))
    return funcs, defines, enums


def pod_struct(text, name):
    """`typedef struct name { scalar fields, fixed arrays, void* } name;` -> a #[repr(C)] Rust struct with the same fields in the same order."""
    body = re.search(r"typedef\s+struct\s+%s\s*\{(.*?)\}\s*%s\s*;" % (name, name), strip_comments(text), flags=re.S).group(1)
    fields = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        m = re.match(r"^(.*?)\s*([A-Za-z_][A-Za-z0-9_]*(?:\s*\[\s*\d+\s*\])?(?:\s*,\s*[A-Za-z_][A-Za-z0-9_]*(?:\s*\[\s*\d+\s*\])?)*)$", decl)
        ctype, names = m.group(1).strip(), m.group(2)
        for n in names.split(","):
            n = n.strip()
            arr = re.match(r"^([A-Za-z_][A-Za-z0-9_]*)\s*\[\s*(\d+)\s*\]$", n)
            rt = rust_type(ctype)
            fields.append((arr.group(1), f"[{rt}; {arr.group(2)}]") if arr else (n, rt))
    lines = ["#[repr(C)]", "#[derive(Clone, Copy)]", f"pub struct {name} {{"]
    lines += [f"    pub {n}: {t}," for n, t in fields]
    lines.append("}")
    return "\n".join(lines)


def generate():
    header_text = open(HEADER).read()
    funcs, defines, enums = parse_header(header_text)
    L = []
    L.append("//! Raw bindings of include/apds.h - GENERATED by tools/gen_apds_sys.py from the header (every function of the C ABI); do not edit by")
    L.append("//! hand. The build image has no Rust toolchain: tests/test_rust_bindings.py re-parses this file and the header independently and")
    L.append("//! fails on any signature (name, arity, pointer depth, constness, scalar type, return type) that differs.")
    L.append("#![allow(non_camel_case_types)]")
    L.append("use std::os::raw::{c_char, c_int, c_void};")
    L.append("")
    for k, v in defines:
        v = v.strip("()")
        if k in ("APDS_MAX_POINTS",):
            continue
        if k == "APDS_PIPELINE_NOT_READY":
            L.append(f"pub const {k}: c_int = {v};")
            continue
        ty = "usize" if k in ("APDS_COMM_ID_BYTES", "APDS_DESC_BYTES", "APDS_DESC_STRIDE") else "c_int"
        L.append(f"pub const {k}: {ty} = {v};")
    L.append("pub const APDS_MAX_POINTS: c_int = (1 << APDS_MAX_POINTS_SHIFT) - 1;")
    for k, v in enums:
        L.append(f"pub const {k}: c_int = {v};")
    L.append("")
    L.append("/// cv::KeyPoint layout (28 bytes)")
    L.append("#[repr(C)]\n#[derive(Clone, Copy, Debug, Default)]\npub struct apds_keypoint {\n    pub x: f32,\n    pub y: f32,\n    pub size: f32,\n    pub angle: f32,\n"
             "    pub response: f32,\n    pub octave: i32,\n    pub class_id: i32,\n}")
    L.append("")
    L.append("/// cv::DMatch layout (16 bytes)")
    L.append("#[repr(C)]\n#[derive(Clone, Copy, Debug, Default)]\npub struct apds_dmatch {\n    pub query_idx: i32,\n    pub train_idx: i32,\n    pub img_idx: i32,\n"
             "    pub distance: f32,\n}")
    L.append("")
    L.append("/// communicator id of the sharded matcher (an ncclUniqueId for the RCCL transport)")
    L.append("#[repr(C)]\n#[derive(Clone, Copy)]\npub struct apds_comm_id {\n    pub bytes: [c_char; APDS_COMM_ID_BYTES],\n}")
    L.append("")
    L.append("/// the host program's own communicator as two callbacks on host buffers (APDS_TRANSPORT_HOST)")
    L.append("#[repr(C)]\n#[derive(Clone, Copy)]\npub struct apds_host_transport {\n    pub user: *mut c_void,\n"
             "    pub all_gather: Option<unsafe extern \"C\" fn(user: *mut c_void, send: *const c_void, recv: *mut c_void, bytes_per_rank: usize) -> c_int>,\n"
             "    pub all_to_all: Option<unsafe extern \"C\" fn(user: *mut c_void, send: *const c_void, send_off: *const usize, send_bytes: *const usize, recv: *mut c_void,\n"
             "                                                recv_off: *const usize, recv_bytes: *const usize) -> c_int>,\n}")
    L.append("")
    L.append("/// the host program's communicator as two callbacks on DEVICE buffers, ordered on the stream they are given (APDS_TRANSPORT_DEVICE)")
    L.append("#[repr(C)]\n#[derive(Clone, Copy)]\npub struct apds_device_transport {\n    pub user: *mut c_void,\n"
             "    pub all_gather: Option<unsafe extern \"C\" fn(user: *mut c_void, send_dev: *const c_void, recv_dev: *mut c_void, bytes_per_rank: usize, stream: *mut c_void) -> c_int>,\n"
             "    pub all_to_all: Option<unsafe extern \"C\" fn(user: *mut c_void, send_dev: *const c_void, send_off: *const usize, send_bytes: *const usize, recv_dev: *mut c_void,\n"
             "                                                recv_off: *const usize, recv_bytes: *const usize, stream: *mut c_void) -> c_int>,\n}")
    for name in POD_STRUCTS:
        L.append("")
        L.append(f"/// {name} (include/apds.h): the streamed frame pipeline's plain-data structs, field for field")
        L.append(pod_struct(header_text, name))
    L.append("")
    L.append('extern "C" {')
    for ret, name, args in funcs:
        params = []
        for ctype, an in args:
            if an in RESERVED:
                an += "_"
            params.append(f"{an}: {rust_type(ctype)}")
        sig = f"    pub fn {name}({', '.join(params)})"
        if ret != "void":
            sig += f" -> {rust_type(ret)}"
        L.append(sig + ";")
    L.append("}")
    return "\n".join(L) + "\n"


def main():
    text = generate()
    if "--check" in sys.argv:
        same = os.path.exists(OUT) and open(OUT).read() == text
        print("apds_sys bindings are current" if same else "apds_sys bindings are STALE: run python tools/gen_apds_sys.py")
        return 0 if same else 1
    open(OUT, "w").write(text)
    print(f"wrote {OUT}: {text.count('pub fn ')} functions")
    return 0


if __name__ == "__main__":
    sys.exit(main())
