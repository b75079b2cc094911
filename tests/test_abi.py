"""CPU: the C-ABI shared library loads, exports every symbol include/rspt.h declares, its structs
have the layout the ctypes mirror assumes, and — without a GPU — every GPU entry point fails
loudly instead of falling back to anything."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

from rs_pbrt_amd import abi, lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rspt.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rspt_[a-z_0-9]+)\s*\(", src)))


def test_every_declared_symbol_is_exported():
    L = lib.lib()
    names = declared_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(L, n), n
    assert set(names) == set(lib.EXPORTS)
    assert L.rspt_abi_version() == abi.ABI_VERSION


def test_struct_layouts_match_the_header():
    """compile a probe against include/rspt.h with gcc and compare sizeof/offsetof with ctypes"""
    probe = r'''
#include <stdio.h>
#include <stddef.h>
#include "rspt.h"
#define S(t) printf(#t " %zu\n", sizeof(t))
#define O(t, f) printf(#t "." #f " %zu\n", offsetof(t, f))
int main(void) {
  S(rspt_bvh_node); S(rspt_prim); S(rspt_mesh); S(rspt_bxdf); S(rspt_material); S(rspt_light);
  S(rspt_scene_desc); S(rspt_medium); S(rspt_envmap); S(rspt_image); S(rspt_texture); S(rspt_sampler_tables); S(rspt_render_desc); S(rspt_ray); S(rspt_hit); S(rspt_stats);
  O(rspt_render_desc, filter_table); O(rspt_render_desc, raster_to_camera); O(rspt_render_desc, spp);
  O(rspt_render_desc, max_depth); O(rspt_render_desc, shard_index); O(rspt_render_desc, tables); O(rspt_render_desc, pixel_dimensions); O(rspt_render_desc, n_light_samples); O(rspt_render_desc, strat_x); O(rspt_render_desc, maxmin_c_pixel); O(rspt_render_desc, sample_begin); O(rspt_render_desc, camera_animated); O(rspt_render_desc, camera_time);
  O(rspt_scene_desc, P); O(rspt_scene_desc, materials); O(rspt_scene_desc, lights); O(rspt_stats, alg_bytes);
  O(rspt_bxdf, alpha_x); O(rspt_bxdf, on_a); O(rspt_bxdf, tex_r); O(rspt_bxdf, tex_t); O(rspt_material, bump_tex);
  O(rspt_scene_desc, textures); O(rspt_scene_desc, images); O(rspt_scene_desc, n_images); O(rspt_scene_desc, n_media); O(rspt_scene_desc, media); O(rspt_mesh, medium_inside); O(rspt_medium, g);
  O(rspt_texture, map); O(rspt_texture, image); O(rspt_texture, max_aniso); O(rspt_texture, value); O(rspt_texture, tex2); O(rspt_texture, tex3); O(rspt_texture, world_to_texture); O(rspt_texture, octaves); O(rspt_texture, variation); O(rspt_image, texels);
  return 0; }'''
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "p.c"), "w").write(probe)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(td, "p"), os.path.join(td, "p.c")])
        out = dict(l.split() for l in subprocess.check_output([os.path.join(td, "p")]).decode().splitlines())
    ct = {"rspt_bvh_node": abi.BvhNode, "rspt_prim": abi.Prim, "rspt_mesh": abi.Mesh, "rspt_bxdf": abi.Bxdf, "rspt_material": abi.Material,
          "rspt_light": abi.Light, "rspt_scene_desc": abi.SceneDesc, "rspt_medium": abi.Medium, "rspt_envmap": abi.EnvMap, "rspt_image": abi.Image, "rspt_texture": abi.Texture, "rspt_sampler_tables": abi.SamplerTables, "rspt_render_desc": abi.RenderDesc,
          "rspt_ray": abi.Ray, "rspt_hit": abi.Hit, "rspt_stats": abi.Stats}
    for k, v in out.items():
        if "." in k:
            t, f = k.split(".")
            assert getattr(ct[t], f).offset == int(v), k
        else:
            assert C.sizeof(ct[k]) == int(v), k
    for dt, t in ((abi.NODE_DT, abi.BvhNode), (abi.PRIM_DT, abi.Prim), (abi.MESH_DT, abi.Mesh), (abi.MEDIUM_DT, abi.Medium), (abi.BXDF_DT, abi.Bxdf),
                  (abi.MATERIAL_DT, abi.Material), (abi.TEXTURE_DT, abi.Texture), (abi.LIGHT_DT, abi.Light), (abi.RAY_DT, abi.Ray), (abi.HIT_DT, abi.Hit)):
        assert dt.itemsize == C.sizeof(t)


def _have_gpu():
    try:
        lib.init(0)
        lib.shutdown()
        return True
    except lib.RsptError:
        return False


def test_no_device_means_errors_not_fallbacks():
    if _have_gpu():
        pytest.skip("a GPU is present; the no-device behaviour is checked on CPU-only hosts")
    L = lib.lib()
    assert L.rspt_init(0) == abi.E_NODEVICE and b"HIP" in L.rspt_last_error() or b"device" in L.rspt_last_error()
    h = C.c_void_p()
    sd = abi.SceneDesc()
    assert L.rspt_scene_create(C.addressof(sd), C.addressof(h)) == abi.E_NODEVICE
    rd = abi.RenderDesc()
    film = np.zeros(16, np.float32)
    assert L.rspt_render(None, C.addressof(rd), film.ctypes.data, None) == abi.E_NODEVICE
    assert L.rspt_trace(None, None, 0, None, 0) == abi.E_NODEVICE
    p = C.c_void_p()
    assert L.rspt_dev_alloc(16, C.addressof(p)) == abi.E_NODEVICE
    with pytest.raises(lib.RsptError):
        lib.init(0)
    L.rspt_shutdown()  # harmless without init


def test_python_mirror_fails_loudly_without_gpu():
    if _have_gpu():
        pytest.skip("GPU present")
    from rs_pbrt_amd import integrator, scenes
    sc = scenes.cornell_box(lib.bvh_build)
    integ = integrator.PathIntegrator(camera=scenes.cornell_render_desc(res=16, spp=1))
    with pytest.raises(lib.RsptError):
        integ.render(sc)
    with pytest.raises(ValueError):
        integrator.PathIntegrator()


def test_bvh_build_argument_errors():
    L = lib.lib()
    assert L.rspt_bvh_build(None, None, 0, 4, None, 0, None, 1) == 0  # empty input: zero nodes
    assert L.rspt_bvh_build(None, None, 5, 4, None, 0, None, 1) == abi.E_INVALID
    P = np.zeros((3, 3), np.float32); P[1, 0] = P[2, 1] = 1
    tri = np.array([[0, 1, 2]], np.uint32)
    nodes = np.zeros(0, abi.NODE_DT); order = np.zeros(1, np.uint32)
    assert L.rspt_bvh_build(P.ctypes.data, tri.ctypes.data, 1, 4, nodes.ctypes.data if nodes.size else order.ctypes.data, 0, order.ctypes.data, 1) == abi.E_INVALID
    assert b"nodes_cap" in L.rspt_bvh_last_error()


def test_rust_shim_layouts_match_the_header_without_a_rust_compiler():
    """rust_shim/ffi.rs is never compiled here (no cargo / rustc in the image): its #[repr(C)] structs are laid out by the repr(C) rules in
    tools/ffi_layout.py and compared with gcc's sizeof / offsetof of include/rspt.h, field by field in declaration order; its ABI constant
    and every `pub fn` of its extern block must exist in the library."""
    import re
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import ffi_layout
    assert ffi_layout.compare() == []
    src = open(os.path.join(ROOT, "rust_shim", "ffi.rs")).read()
    assert int(re.search(r"RSPT_ABI_VERSION: c_int = (\d+);", src).group(1)) == abi.ABI_VERSION
    assert "ABI version %d" % abi.ABI_VERSION in src.splitlines()[0]
    fns = re.findall(r"pub fn (rspt_\w+)\(", src)
    assert len(fns) >= 15 and all(f in lib.EXPORTS for f in fns), [f for f in fns if f not in lib.EXPORTS]


def test_the_loaded_library_was_built_from_this_tree():
    """profiles/ are tied to the hash the LIBRARY reports (compiled in by csrc/Makefile); a stale librspt.so — it is git-ignored and shipped prebuilt to the GPU
    box — would report another hash than the sources next to it"""
    assert lib.source_hash() == lib.tree_source_hash(), "librspt.so is stale: rebuild with make -C rs_pbrt_amd/csrc"
    assert len(lib.source_hash()) == 16 and int(lib.source_hash(), 16) >= 0
