"""The Rust shim crate (rust-shim/) cannot be compiled in this image (no rustc); instead every `extern "C"`
item, constant and #[repr(C)] struct of rust-shim/src/ffi.rs is checked against include/kmeans_hip.h, and the
public surface of rust-shim/src/lib.rs against what the reference's cli/ and examples import."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

C_TO_RUST = {
    "int": "c_int", "void": "()", "uint32_t": "u32", "int32_t": "i32", "uint64_t": "u64", "int64_t": "i64",
    "uint8_t": "u8", "float": "f32", "char": "c_char",
}


def _strip_comments(text, rust=False):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def _c_type_to_rust(t):
    t = t.strip()
    const = "const " in (t + " ") or t.startswith("const")
    t = t.replace("const", " ").strip()
    stars = t.count("*")
    base = t.replace("*", " ").split()
    base = base[0] if base else "void"
    rust = C_TO_RUST.get(base, base)                 # struct names map to themselves
    for i in range(stars):
        # the innermost pointer carries the constness written in C ("const T *" -> *const T)
        rust = ("*const " if (const and i == 0) else "*mut ") + rust
    return rust


def _header_functions():
    text = _strip_comments(open(os.path.join(ROOT, "include", "kmeans_hip.h")).read())
    out = {}
    for m in re.finditer(r"KMG_API\s+([\w\s\*]+?)\b(kmg_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = re.sub(r"\[[^\]]*\]", "*", a.strip())            # array parameters decay to pointers
                toks = a.replace("*", " * ").split()                  # "const uint8_t *rgba": the name is the last token
                params.append(_c_type_to_rust(" ".join(toks[:-1])))
        out[name] = (_c_type_to_rust(ret), params)
    return out


def _rust_functions():
    text = _strip_comments(open(os.path.join(ROOT, "rust-shim", "src", "ffi.rs")).read())
    block = re.search(r'extern\s+"C"\s*\{(.*)\}', text, flags=re.S).group(1)
    out = {}
    for m in re.finditer(r"pub\s+fn\s+(kmg_\w+)\s*\((.*?)\)\s*(->\s*([^;]+))?;", block, flags=re.S):
        name, args, ret = m.group(1), m.group(2), (m.group(4) or "()").strip()
        params = [re.sub(r"\s+", " ", a.split(":", 1)[1].strip()) for a in args.split(",") if ":" in a]
        out[name] = (ret, params)
    return out, text


def test_every_extern_item_matches_the_header():
    header = _header_functions()
    rust, _ = _rust_functions()
    assert {"kmg_processor_create", "kmg_processor_destroy", "kmg_palette", "kmg_find", "kmg_reduce", "kmg_last_error"} <= set(rust)
    for name, (ret, params) in rust.items():
        assert name in header, f"{name} is not declared in include/kmeans_hip.h"
        hret, hparams = header[name]
        assert len(params) == len(hparams), f"{name}: {len(params)} arguments in ffi.rs, {len(hparams)} in the header"
        assert ret == hret, f"{name}: returns {ret} in ffi.rs, {hret} in the header"
        for i, (r, h) in enumerate(zip(params, hparams)):
            assert r == h, f"{name} argument {i}: {r} in ffi.rs, {h} in the header"


def test_constants_and_options_struct_match_the_header():
    header = _strip_comments(open(os.path.join(ROOT, "include", "kmeans_hip.h")).read())
    _, rust = _rust_functions()
    for name in ("KMG_OK", "KMG_ALGO_KMEANS", "KMG_ALGO_OCTREE", "KMG_MODE_REPLACE", "KMG_MODE_DITHER", "KMG_MODE_MELD"):
        c = int(re.search(name + r"\s*=\s*(-?\d+)", header).group(1))
        r = int(re.search(r"pub const " + name + r": c_int = (-?\d+);", rust).group(1))
        assert c == r, name
    cfields = re.search(r"typedef struct kmg_options \{(.*?)\} kmg_options;", header, flags=re.S).group(1)
    cfields = [(_c_type_to_rust(t), n) for t, n in re.findall(r"(\w+)\s+(\w+)\s*;", cfields)]
    rfields = re.search(r"pub struct kmg_options \{(.*?)\}", rust, flags=re.S).group(1)
    rfields = [(t.strip(), n) for n, t in re.findall(r"pub (\w+):\s*([\w\*\s]+?),", rfields)]
    assert cfields == rfields
    assert re.search(r"#\[repr\(C\)\]\s*#\[derive\(Clone, Copy\)\]\s*pub struct kmg_options", rust)


def test_public_surface_of_the_crate():
    lib = open(os.path.join(ROOT, "rust-shim", "src", "lib.rs")).read()
    img = open(os.path.join(ROOT, "rust-shim", "src", "image.rs")).read()
    cargo = open(os.path.join(ROOT, "rust-shim", "Cargo.toml")).read()
    # what cli/src/main.rs:5, cli/src/args.rs:9,115-150 and core/examples/*.rs import from the reference crate
    for item in ("pub struct ImageProcessor", "pub async fn new() -> Result<Self>", "pub async fn palette<C: Container>",
                 "pub async fn find<C: Container>", "pub async fn reduce<C: Container>", "pub use rgb::RGBA8",
                 "pub enum Algorithm", "pub enum ReduceMode", "pub enum ColorSpace", "pub mod image",
                 "unsafe impl Send for ImageProcessor", "unsafe impl Sync for ImageProcessor"):
        assert item in lib, item
    for item in ("pub trait Container", "pub struct Image<C: Container>", "pub fn copied_pixel", "pub fn borrowed_pixel",
                 "pub fn dimensions(&self) -> (u32, u32)", "pub fn into_raw_pixels(self) -> Vec<u8>"):
        assert item in img, item
    assert 'name = "kmeans-color-gpu"' in cargo and 'name = "kmeans_color_gpu"' in cargo
    assert os.path.exists(os.path.join(ROOT, "rust-shim", "build.rs"))
