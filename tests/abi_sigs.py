"""Signature-level comparison of the uncompiled Rust binding (integration/rust/src/amd_ffi.rs) with the C header: every `pub fn pa_*`
prototype (arity, each parameter's type, the return type) and every #[repr(C)] struct (field order, names, types), through a fixed
Rust -> C type map. No Rust toolchain exists in this image; this is what stands between a drifted u32 / u64 or a swapped argument in
amd_ffi.rs and a silent ABI break."""
import re

RUST_SCALARS = {"u8": "uint8_t", "u32": "uint32_t", "u64": "uint64_t", "i32": "int32_t", "i64": "int64_t", "usize": "size_t", "f32": "float", "f64": "double",
                "c_int": "int", "c_char": "char", "c_void": "void",
                "PaIndex": "pa_index", "PaHostIndex": "pa_host_index", "PaFlatIndex": "pa_flat_index", "PaReadResult": "pa_read_result",
                "PaIndexStats": "pa_index_stats", "PaOverflow": "pa_overflow", "PaComm": "pa_comm", "PaRecordStream": "pa_record_stream",
                "PaTxome": "pa_txome"}
C_SIZES = {"uint8_t": 1, "uint32_t": 4, "uint64_t": 8, "int": 4, "size_t": 8, "float": 4, "double": 8, "char": 1}


def rust_type_to_c(t: str) -> str:
    """`*const u32` -> `const uint32_t*`, `*mut *const u32` -> `const uint32_t**`, `u64` -> `uint64_t`"""
    t = t.strip()
    m = re.match(r"\*(const|mut)\s+(.*)$", t)
    if m:
        inner = rust_type_to_c(m.group(2))
        if m.group(1) == "const":
            # const applies to the pointee
            return ("const " + inner + "*") if not inner.endswith("*") else (inner + "const*")   # (`T*const*`: as the header's side is normalised)
        return inner + "*"
    if t not in RUST_SCALARS:
        raise ValueError("unmapped Rust type %r" % t)
    return RUST_SCALARS[t]


def norm_c_type(t: str) -> str:
    """canonical spelling of a C parameter type (name and array suffix already removed)"""
    t = re.sub(r"\bstruct\s+", "", t)
    t = re.sub(r"\s*\*\s*", "*", t.strip())
    t = re.sub(r"\s+", " ", t)
    # `T const*` -> `const T*` for the innermost pointee
    m = re.match(r"^(\w+) const(\*.*)$", t)
    if m:
        t = "const %s%s" % (m.group(1), m.group(2))
    return t


def split_args(s: str):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "(<[":
            depth += 1
        elif ch in ")>]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [a.strip() for a in out]


def c_param_type(p: str) -> str:
    """type of one C parameter declaration: `const uint32_t* d_lens` / `float ms[3]` / `void`"""
    p = p.strip()
    arr = re.search(r"\[[^\]]*\]\s*$", p)
    if arr:
        p = p[: arr.start()].rstrip()
    m = re.match(r"^(.*?)(\b[A-Za-z_]\w*)$", p)            # trailing identifier = the parameter's name
    if m and m.group(1).strip() and m.group(2) not in C_SIZES and not m.group(2).startswith("pa_") and m.group(2) not in ("void",):
        p = m.group(1)
    t = norm_c_type(p)
    return t + "*" if arr else t


def header_prototypes(text: str):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(pa_\w+)\s*\(([^;{}]*)\)\s*;", text):
        ret, name, args = norm_c_type(m.group(1)), m.group(2), m.group(3).strip()
        params = [] if args in ("", "void") else [c_param_type(a) for a in split_args(args)]
        protos[name] = (ret, params)
    return protos


def header_structs(text: str):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            parts = [x.strip() for x in decl.split(",")]          # `uint64_t a, b, c`: one type, several declarators
            mm = re.match(r"^(.*?)(\b\w+)$", parts[0])
            ty = norm_c_type(mm.group(1))
            fields.append((mm.group(2), ty))
            for extra in parts[1:]:
                stars = len(extra) - len(extra.lstrip("* "))
                fields.append((extra.lstrip("* ").strip(), ty.rstrip("*") + "*" * extra.count("*") if stars else ty))
        out[m.group(3)] = fields
    return out


def rust_prototypes(text: str):
    text = re.sub(r"//.*$", "", text, flags=re.M)
    protos = {}
    for m in re.finditer(r"pub fn (pa_\w+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+))?;", text, flags=re.S):
        params = []
        for a in split_args(m.group(2)):
            params.append(rust_type_to_c(a.split(":", 1)[1]))
        ret = rust_type_to_c(m.group(3)) if m.group(3) else "void"
        protos[m.group(1)] = (ret, params)
    return protos


def rust_structs(text: str):
    text = re.sub(r"//.*$", "", text, flags=re.M)
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\](?:\s*#\[[^\]]*\])*\s*pub struct (\w+)\s*\{(.*?)\}", text, flags=re.S):
        fields = []
        for decl in split_args(m.group(2)):
            if not decl or decl.startswith("_private"):
                continue
            name, ty = decl.replace("pub ", "", 1).split(":", 1)
            fields.append((name.strip(), rust_type_to_c(ty)))
        if fields:
            out[RUST_SCALARS.get(m.group(1), m.group(1))] = fields
    return out


def rust_consts(text: str):
    text = re.sub(r"//.*$", "", text, flags=re.M)
    return {m.group(1): int(m.group(2).replace("_", ""), 0) for m in re.finditer(r"pub const (\w+)\s*:\s*\w+\s*=\s*(-?[0-9A-Fa-fx_]+)\s*;", text)}


def header_consts(text: str):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for m in re.finditer(r"#define\s+(PA_\w+)\s+(0x[0-9A-Fa-f]+|\d+)(?:u|ull|ULL|U)?\b", text):
        out[m.group(1)] = int(m.group(2), 0)
    for m in re.finditer(r"\b(PA_\w+)\s*=\s*(-?\d+)", text):
        out[m.group(1)] = int(m.group(2))
    return out


def layout(fields):
    """(offsets by field, sizeof) of a C struct under natural alignment on LP64"""
    off, offs, align = 0, {}, 1
    for name, ty in fields:
        size = 8 if ty.endswith("*") else C_SIZES[ty]
        off = (off + size - 1) // size * size
        offs[name] = off
        off += size
        align = max(align, size)
    return offs, (off + align - 1) // align * align
