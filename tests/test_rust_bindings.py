"""The Rust side of the boundary (integration/raytracer-hip) cannot be compiled where this repository
is built (no Rust toolchain), so its raw bindings are checked textually against the C headers: every
exported rt_* function is declared in src/sys.rs with the same parameters, every struct of
include/rt_scene.h and include/rt_hip.h has a #[repr(C)] mirror with the same fields in the same
order, and the constants agree.  The patches for the reference's own crates must at least be
well-formed unified diffs that touch the files they claim to."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SYS_RS = os.path.join(ROOT, "integration", "raytracer-hip", "src", "sys.rs")
HEADERS = [os.path.join(ROOT, "include", "rt_scene.h"), os.path.join(ROOT, "include", "rt_hip.h")]

SCALARS = {"int32_t": "i32", "uint32_t": "u32", "uint16_t": "u16", "uint8_t": "u8", "uint64_t": "u64", "int64_t": "i64",
           "double": "f64", "float": "f32", "size_t": "usize", "int": "c_int", "char": "c_char", "void": "c_void"}


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def c_type_to_rust(ctype):
    """'const RtShape *' -> '*const RtShape', 'uint32_t' -> 'u32', 'float *' -> '*mut f32', 'T *const *' -> '*const *mut T'."""
    ctype = " ".join(ctype.replace("*", " * ").split())
    toks = ctype.split(" ")
    # base type with an optional leading const
    const_base = toks[0] == "const"
    if const_base:
        toks = toks[1:]
    base = SCALARS.get(toks[0], toks[0])
    rest = toks[1:]
    out = base
    pointee_const = const_base
    i = 0
    while i < len(rest):
        assert rest[i] == "*", ctype
        out = ("*const " if pointee_const else "*mut ") + out
        pointee_const = False
        if i + 1 < len(rest) and rest[i + 1] == "const":  # '* const': the pointer itself is const -> affects the next level
            pointee_const = True
            i += 1
        i += 1
    return out


def parse_header_structs(text):
    structs = {}
    for m in re.finditer(r"typedef struct (\w+) \{(.*?)\}\s*\1;", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            if "(*" in decl:  # the one function pointer: void (*progress)(...)
                fields.append((re.search(r"\(\*(\w+)\)", decl).group(1), "RtProgressFn"))
                continue
            # 'double x, y, z' | 'float diffuse[3]' | 'const RtShape *shapes' | 'uint32_t tile_w, tile_h'
            head, _, names = decl.rpartition(" ") if "," not in decl else (None, None, None)
            if head is None:
                first, *others = [d.strip() for d in decl.split(",")]
                head, _, name0 = first.rpartition(" ")
                names_list = [name0] + others
            else:
                names_list = [names]
            for name in names_list:
                ctype = head
                while name.startswith("*"):
                    ctype += " *"
                    name = name[1:]
                arr = re.match(r"(\w+)\[(\w+)\]$", name)
                if arr:
                    fields.append((arr.group(1), f"[{c_type_to_rust(ctype)}; {arr.group(2)}]"))
                else:
                    fields.append((name, c_type_to_rust(ctype)))
        structs[m.group(1)] = fields
    return structs


def parse_header_functions(text):
    fns = {}
    for m in re.finditer(r"(?:^|\n)\s*((?:const\s+)?\w+\s*\*?)\s*(rt_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret, name, params = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        args = []
        if params and params != "void":
            for p in params.split(","):
                p = p.strip()
                arr = re.match(r"(.*?)(\w+)\[\w+\]$", p)  # 'uint8_t id_out[RT_COMM_ID_BYTES]' decays to a pointer
                if arr:
                    args.append(c_type_to_rust(arr.group(1).strip() + " *"))
                    continue
                t, _, n = p.rpartition(" ")
                while n.startswith("*"):
                    t += " *"
                    n = n[1:]
                args.append(c_type_to_rust(t))
        fns[name] = (None if ret == "void" else c_type_to_rust(ret), args)
    return fns


def parse_rust(text):
    text = re.sub(r"//[^\n]*", "", text)
    structs = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^\]]*\)\]\s*)?pub struct (\w+) \{(.*?)\n\}", text, flags=re.S):
        fields = []
        for f in re.finditer(r"(?:pub )?(?:r#)?(\w+): ([^,\n]+),", m.group(2)):
            fields.append((f.group(1), f.group(2).strip()))
        structs[m.group(1)] = fields
    fns = {}
    block = re.search(r'extern "C" \{(.*)\n\}', text, flags=re.S).group(1)
    for m in re.finditer(r"pub fn (rt_\w+)\((.*?)\)(?:\s*->\s*([^;]+))?;", block, flags=re.S):
        args = [a.split(":", 1)[1].strip() for a in " ".join(m.group(2).split()).split(", ") if a.strip()]
        fns[m.group(1)] = (m.group(3).strip() if m.group(3) else None, args)
    consts = {m.group(1): m.group(2) for m in re.finditer(r"pub const (\w+): \w+ = ([^;]+);", text)}
    return structs, fns, consts


def load():
    header = "\n".join(strip_comments(open(h).read()) for h in HEADERS)
    return header, parse_rust(open(SYS_RS).read())


def test_every_exported_function_is_bound_with_the_same_signature():
    header, (_, rust_fns, _) = load()
    c_fns = parse_header_functions(header)
    assert len(c_fns) >= 30, sorted(c_fns)
    assert sorted(c_fns) == sorted(rust_fns)
    for name, (ret, args) in c_fns.items():
        r_ret, r_args = rust_fns[name]
        assert r_ret == ret, (name, ret, r_ret)
        assert r_args == args, (name, args, r_args)


def test_every_struct_has_a_repr_c_mirror_with_the_same_fields():
    header, (rust_structs, _, consts) = load()
    c_structs = parse_header_structs(header)
    assert {"RtVec3", "RtCamera", "RtShader", "RtLight", "RtShape", "RtMesh", "RtScene", "RtParseArgs", "RtRenderArgs",
            "RtRayIds", "RtStats", "RtUploadOptions"} <= set(c_structs)
    for name, fields in c_structs.items():
        assert name in rust_structs, name
        got = rust_structs[name]
        assert [f for f, _ in got] == [f for f, _ in fields], (name, got, fields)
        for (fname, ctype), (_, rtype) in zip(fields, got):
            want = re.sub(r"; (RT_\w+)\]", lambda m: f"; {m.group(1)}]", ctype)
            if re.search(r"; \d+\]", want) or "RT_" in want:
                # array lengths: compare numerically through the Rust constants where a macro names them
                n_c = re.search(r"; (\w+)\]", want).group(1)
                n_r = re.search(r"; (\w+)\]", rtype).group(1)
                assert (n_c == n_r) or consts.get(n_c) == n_r or consts.get(n_r) == n_c, (name, fname, want, rtype)
                assert want.split(";")[0] == rtype.split(";")[0], (name, fname, want, rtype)
            else:
                assert rtype == want, (name, fname, want, rtype)
    # opaque handles exist as types
    for handle in ("RtBackend", "RtDeviceScene", "RtMulti", "RtComm"):
        assert handle in rust_structs


def test_constants_agree_with_the_headers():
    header, (_, _, consts) = load()
    for name, value in re.findall(r"#define (RT_\w+) \(?(-?\d+)\)?\s", header):
        assert consts.get(name) == value, (name, value, consts.get(name))
    for body in re.findall(r"enum \{(.*?)\}", header, flags=re.S):
        for name, value in re.findall(r"(RT_\w+) = (\d+)", body):
            assert consts.get(name) == value, (name, value, consts.get(name))
    assert consts["RT_STATIC_SCRATCH_BYTES"].replace(" ", "") == "64<<20"


def test_patches_are_well_formed_and_say_what_they_are():
    pdir = os.path.join(ROOT, "integration", "patches")
    lib = open(os.path.join(pdir, "raytracer-lib-flatten.patch")).read()
    cli = open(os.path.join(pdir, "raytracer-cli-backend.patch")).read()
    for text in (lib, cli):
        assert text.startswith("# NOT COMPILED")
        # every hunk header's line counts match the hunk's body
        for m in re.finditer(r"^@@ -(\d+)(?:,(\d+))? \+(\d+)(?:,(\d+))? @@.*\n((?:[ +\-\\].*\n?)*)", text, flags=re.M):
            old_n = int(m.group(2) or 1)
            new_n = int(m.group(4) or 1)
            body = [l for l in m.group(5).splitlines() if not l.startswith("\\")]
            body = body[: next((i for i, l in enumerate(body) if l.startswith(("--- ", "+++ ", "diff "))), len(body))]
            assert sum(1 for l in body if l[:1] in " -") == old_n, m.group(0)[:80]
            assert sum(1 for l in body if l[:1] in " +") == new_n, m.group(0)[:80]
    # one flatten() per Shape / Shader / Light / Camera implementation, and the new module
    for rel in ("geometry/sphere.rs", "geometry/cuboid.rs", "geometry/triangle.rs", "geometry/mesh.rs", "geometry/instance.rs",
                "shader/lambertian.rs", "shader/blinn_phong.rs", "shader/perfect_mirror.rs", "shader/normal.rs", "shader/null.rs",
                "shader/ggx_mirror.rs", "light/point.rs", "light/ambient.rs", "camera/perspective.rs", "camera/orthographic.rs",
                "flat.rs", "lib.rs", "geometry/mod.rs", "shader/mod.rs", "light/mod.rs", "camera/mod.rs"):
        assert f"+++ b/raytracer-lib/src/{rel}" in lib, rel
    assert lib.count("+    fn flatten(") >= 17
    assert "+++ b/raytracer-cli/src/main.rs" in cli and "Backend::Hip" in cli and "raytracer-hip" in cli
    for f in ("Cargo.toml", "build.rs", "src/sys.rs", "src/lib.rs"):
        head = open(os.path.join(ROOT, "integration", "raytracer-hip", f)).read(600)
        assert "NOT COMPILED" in head, f


REFERENCE = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "raytracer-lib")) or shutil.which("patch") is None,
                    reason="needs the reference's sources and patch(1)")
def test_patches_apply_to_the_reference_sources(tmp_path):
    """Both patches apply without fuzz or rejects to a scratch copy of the files they name, and the
    flattened traits end up with one `flatten` per implementation."""
    pdir = os.path.join(ROOT, "integration", "patches")
    work = tmp_path / "ref"
    for patch in ("raytracer-lib-flatten.patch", "raytracer-cli-backend.patch"):
        text = open(os.path.join(pdir, patch)).read()
        for rel in re.findall(r"^--- a/(\S+)", text, flags=re.M):
            src = os.path.join(REFERENCE, rel)
            dst = work / rel
            if os.path.isfile(src) and not dst.exists():
                dst.parent.mkdir(parents=True, exist_ok=True)
                shutil.copyfile(src, dst)
    for patch in ("raytracer-lib-flatten.patch", "raytracer-cli-backend.patch"):
        r = subprocess.run(["patch", "-p1", "--fuzz=0", "--no-backup-if-mismatch", "-i", os.path.join(pdir, patch)],
                           cwd=work, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "fuzz" not in r.stdout and ".rej" not in r.stdout, r.stdout
    shape_trait = (work / "raytracer-lib/src/geometry/mod.rs").read_text()
    assert "fn flatten(" in shape_trait
    assert "Backend::Hip" in (work / "raytracer-cli/src/main.rs").read_text()
    assert '"raytracer-hip"' in (work / "Cargo.toml").read_text()
