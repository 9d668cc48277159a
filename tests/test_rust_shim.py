"""integration/rust/src/ffi.rs against include/rt.h, mechanically (there is no Rust toolchain to compile it with):
every function the header declares is declared in ffi.rs with the same number of arguments and the same argument and
return types, and nothing else; every #[repr(C)] struct has the fields, in order, and the size the C struct has."""
import ctypes as C
import os
import re

from raytracer_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "rt.h")).read()
FFI = open(os.path.join(ROOT, "integration", "rust", "src", "ffi.rs")).read()


def strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def c_type_to_rust(t):
    """Canonical Rust spelling of one C parameter / return type of rt.h."""
    t = re.sub(r"\s+", " ", t.strip())
    t = re.sub(r"\[\d*\]$", "*", t)                     # `const double p[3]` decays to a pointer
    t = t.replace(" *", "*").replace("* ", "*")
    m = re.fullmatch(r"(\w+)\*const\*", t)              # `rt_context* const* ctxs`: a constant array of mutable pointers
    if m:
        return "*const *mut " + m.group(1)
    base_map = {"int": "c_int", "double": "f64", "uint8_t": "u8", "uint32_t": "u32", "uint64_t": "u64", "size_t": "usize",
                "char": "c_char", "void": "c_void"}
    m = re.fullmatch(r"(const )?(\w+)(\**)", t)
    assert m, t
    const, base, stars = m.group(1), m.group(2), m.group(3)
    rust = base_map.get(base, base)
    if not stars:
        return rust
    out = ("*const " if const else "*mut ") + rust
    for _ in stars[1:]:
        out = "*mut " + out
    return out


def c_functions():
    text = strip_c_comments(HEADER)
    text = re.sub(r"typedef struct\s*\{.*?\}\s*\w+;", " ", text, flags=re.S)
    out = {}
    for m in re.finditer(r"(?m)^\s*((?:const\s+)?\w+\s*\**)\s*(rt_\w+)\s*\(([^;{]*?)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        params = []
        for a in [x for x in args.split(",") if x.strip() and x.strip() != "void"]:
            a = a.strip()
            mm = re.fullmatch(r"(.*?[\s\*])(\w+)(\[\d*\])?", a)       # type + name (+ array suffix)
            if mm and mm.group(2) not in ("rt_builder", "rt_context", "rt_scene", "rt_camera", "rt_params", "rt_stats", "rt_tree_stats",
                                          "rt_scene_info", "rt_ray_d", "rt_hit") and not a.endswith("*"):
                ty = mm.group(1) + (mm.group(3) or "")
            else:
                ty = a                                                # unnamed parameter: `rt_builder*`
            params.append(c_type_to_rust(ty))
        r = re.sub(r"\s+", " ", ret.strip())
        out[name] = (params, None if r == "void" else c_type_to_rust(r))
    return out


def rust_functions():
    block = FFI[FFI.index('extern "C" {'):]
    out = {}
    for m in re.finditer(r"pub fn (\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", block, flags=re.S):
        params = [p.split(":", 1)[1].strip() for p in m.group(2).split(",") if p.strip()]
        out[m.group(1)] = (params, m.group(3).strip() if m.group(3) else None)
    return out


def test_every_entry_point_is_declared_with_the_same_signature():
    c, r = c_functions(), rust_functions()
    assert len(c) >= 48 and set(c) == set(_lib.PROTOTYPES)             # the header parser sees what the ctypes binding sees
    assert set(r) == set(c), (sorted(set(c) - set(r)), sorted(set(r) - set(c)))
    for name in c:
        assert r[name] == c[name], (name, r[name], c[name])


SIZES = {"u8": 1, "u32": 4, "c_int": 4, "u64": 8, "f64": 8, "usize": 8}


def rust_struct_layouts():
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^\)]*\)\]\s*)?pub struct (\w+)\s*\{(.*?)\}", FFI, flags=re.S):
        fields = []
        for f in re.finditer(r"pub (\w+):\s*([^,]+),", m.group(2)):
            fields.append((f.group(1), f.group(2).strip()))
        out[m.group(1)] = fields
    return out


def layout_size(fields, layouts):
    offset, align = 0, 1
    for _, ty in fields:
        count = 1
        arr = re.fullmatch(r"\[(\w+); (\d+)\]", ty)
        if arr:
            ty, count = arr.group(1), int(arr.group(2))
        if ty in SIZES:
            size = a = SIZES[ty]
        else:
            size, a = layout_size(layouts[ty], layouts)
        offset = (offset + a - 1) // a * a + size * count
        align = max(align, a)
    return (offset + align - 1) // align * align, align


def test_every_struct_has_the_c_layout():
    layouts = rust_struct_layouts()
    for name in ("rt_camera", "rt_scene_info", "rt_ray_d", "rt_hit", "rt_params", "rt_stats", "rt_tree_stats"):
        cs = getattr(_lib, name)
        rust_names = [n for n, _ in layouts[name]]
        c_names = [n.rstrip("_") for n, _ in cs._fields_]
        assert rust_names == c_names, name
        assert layout_size(layouts[name], layouts)[0] == C.sizeof(cs), name
    # and the C structs of the header have the fields the ctypes mirrors have, in order
    text = strip_c_comments(HEADER)
    for m in re.finditer(r"typedef struct\s*\{(.*?)\}\s*(rt_\w+);", text, flags=re.S):
        names = []
        for decl in m.group(1).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                names.append(re.sub(r"\[\d+\]", "", part.strip().split()[-1]).lstrip("*"))
        assert names == [n.rstrip("_") for n, _ in getattr(_lib, m.group(2))._fields_], m.group(2)


def test_the_wrappers_use_only_declared_entry_points():
    declared = set(rust_functions())
    for fn in ("scene.rs", "main.rs"):
        src = open(os.path.join(ROOT, "integration", "rust", "src", fn)).read()
        used = set(re.findall(r"ffi::(rt_\w+)\s*\(", src))
        assert used and used <= declared, (fn, used - declared)
