"""The `#[repr(C)]` structs of rust_shim/ffi.rs against include/rspt.h WITHOUT a Rust compiler (this image has none): the Rust side's layout is
computed by the repr(C) rules (x86-64: natural alignment, fields in declaration order, size rounded up to the struct's alignment), the C side's
is measured by gcc (sizeof / offsetof of every field, in order).  The two lists are compared position by position (field names may differ in
case), so a maintainer sees a mismatch here before `cargo build` shows it as corrupted scene arrays.
usage: python tools/ffi_layout.py            prints the table and exits non-zero on the first difference (tests/test_abi.py runs it)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRIM = {"f32": (4, 4), "u32": (4, 4), "i32": (4, 4), "u16": (2, 2), "i16": (2, 2), "u8": (1, 1), "i8": (1, 1), "u64": (8, 8), "i64": (8, 8), "f64": (8, 8), "c_int": (4, 4)}
# Rust struct -> C typedef
NAMES = {"RsptBvhNode": "rspt_bvh_node", "RsptPrim": "rspt_prim", "RsptMesh": "rspt_mesh", "RsptMedium": "rspt_medium", "RsptMaterialDesc": "rspt_material_desc",
         "RsptImage": "rspt_image", "RsptTexture": "rspt_texture", "RsptLight": "rspt_light", "RsptEnvMap": "rspt_envmap", "RsptObject": "rspt_object",
         "RsptInstance": "rspt_instance", "RsptSceneDesc": "rspt_scene_desc", "RsptSamplerTables": "rspt_sampler_tables", "RsptRenderDesc": "rspt_render_desc",
         "RsptStats": "rspt_stats"}


def rust_structs(path):
    src = re.sub(r"//[^\n]*", "", open(path).read())
    out = {}
    for m in re.finditer(r"#\[repr\(C\)\][^{;]*?pub struct (\w+)\s*\{(.*?)\}", src, re.S):
        fields = [(f.group(1), f.group(2).strip()) for f in re.finditer(r"pub (\w+)\s*:\s*([^,]+?)\s*(?:,|$)", m.group(2).strip(), re.S)]
        out[m.group(1)] = fields
    return out


def rust_layout(structs, name, cache):
    """[(field, offset, size)], size, align of a repr(C) struct"""
    if name in cache:
        return cache[name]

    def ty(t):
        t = t.strip()
        if t.startswith("*"):
            return 8, 8
        a = re.match(r"\[\s*(.+?)\s*;\s*(\d+)\s*\]$", t)
        if a:
            s, al = ty(a.group(1))
            return s * int(a.group(2)), al
        if t in PRIM:
            return PRIM[t]
        _, s, al = rust_layout(structs, t, cache)
        return s, al
    off, align, rows = 0, 1, []
    for f, t in structs[name]:
        s, a = ty(t)
        off = (off + a - 1) // a * a
        rows.append((f, off, s))
        off += s
        align = max(align, a)
    cache[name] = (rows, (off + align - 1) // align * align, align)
    return cache[name]


def c_fields(header):
    src = re.sub(r"/\*.*?\*/", "", open(header).read(), flags=re.S)
    out = {}
    for m in re.finditer(r"typedef struct\s*\w*\s*\{(.*?)\}\s*(\w+)\s*;", src, re.S):
        names = []
        for decl in m.group(1).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for d in decl.split(","):
                n = re.search(r"(\w+)\s*(?:\[[^\]]*\]\s*)*$", d.strip())
                names.append(n.group(1))
        out[m.group(2)] = names
    return out


def c_layout(header, want):
    fields = c_fields(header)
    body = []
    for c in want:
        body.append('printf("%s %%zu\\n", sizeof(%s));' % (c, c))
        for f in fields[c]:
            body.append('printf("%s.%s %%zu %%zu\\n", offsetof(%s, %s), sizeof(((%s*)0)->%s));' % (c, f, c, f, c, f))
    prog = '#include <stdio.h>\n#include <stddef.h>\n#include "rspt.h"\nint main(void) {\n%s\nreturn 0; }\n' % "\n".join(body)
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "p.c"), "w").write(prog)
        subprocess.check_call(["gcc", "-I", os.path.dirname(header), "-o", os.path.join(td, "p"), os.path.join(td, "p.c")])
        lines = subprocess.check_output([os.path.join(td, "p")]).decode().splitlines()
    out = {}
    for l in lines:
        p = l.split()
        if "." in p[0]:
            c, f = p[0].split(".")
            out[c][0].append((f, int(p[1]), int(p[2])))
        else:
            out[p[0]] = ([], int(p[1]))
    return out


def compare(verbose=False):
    rs = rust_structs(os.path.join(ROOT, "rust_shim", "ffi.rs"))
    missing = [r for r in rs if r not in NAMES]
    cl = c_layout(os.path.join(ROOT, "include", "rspt.h"), [NAMES[r] for r in rs if r in NAMES])
    cache, bad = {}, ["ffi.rs struct %s has no C counterpart listed in tools/ffi_layout.py" % m for m in missing]
    for r, c in NAMES.items():
        if r not in rs:
            bad.append("ffi.rs lacks %s (%s)" % (r, c))
            continue
        rows, size, _ = rust_layout(rs, r, cache)
        crow, csize = cl[c]
        if verbose:
            print("%s / %s: %d B (C %d B), %d fields (C %d)" % (r, c, size, csize, len(rows), len(crow)))
        if size != csize or len(rows) != len(crow):
            bad.append("%s: %d B, %d fields; %s: %d B, %d fields" % (r, size, len(rows), c, csize, len(crow)))
        for (f, o, s), (cf, co, cs) in zip(rows, crow):
            if verbose:
                print("    %-22s @%4d %4d B   | %-22s @%4d %4d B%s" % (f, o, s, cf, co, cs, "" if (o, s) == (co, cs) else "   <-- differs"))
            if (o, s) != (co, cs) or f.lower() != cf.lower():
                bad.append("%s.%s @%d %d B != %s.%s @%d %d B" % (r, f, o, s, c, cf, co, cs))
    return bad


if __name__ == "__main__":
    bad = compare(verbose=True)
    for b in bad:
        print("MISMATCH:", b)
    sys.exit(1 if bad else 0)
