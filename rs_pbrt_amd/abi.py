"""ctypes mirror of include/rspt.h (the C ABI of librspt.so) — layouts only, no logic.

Every struct here must stay byte-compatible with the header; tests/test_abi.py checks
sizes against the values the library reports."""
import ctypes as C

ABI_VERSION = 21

OK, E_INVALID, E_NODEVICE, E_HIP, E_UNSUPPORTED, E_NOMEM, E_PEER = 0, -1, -2, -3, -4, -5, -6

BXDF_LAMBERT_R, BXDF_OREN_NAYAR, BXDF_SPECULAR_R, BXDF_SPECULAR_T, BXDF_FRESNEL_SPEC, BXDF_MICROFACET_R, BXDF_LAMBERT_T = 1, 2, 3, 4, 5, 6, 7
BXDF_MICROFACET_T, BXDF_FRESNEL_BLEND = 8, 9
FRESNEL_NOOP, FRESNEL_DIELECTRIC, FRESNEL_CONDUCTOR = 0, 1, 2
MAT_MATTE, MAT_PLASTIC, MAT_MIRROR, MAT_GLASS, MAT_METAL, MAT_SUBSTRATE, MAT_UBER, MAT_TRANSLUCENT, MAT_MIX = range(1, 10)
LOBE_REMAP, LOBE_NODIFF = 1, 2
LIGHT_DIFFUSE_AREA, LIGHT_POINT, LIGHT_SPOT, LIGHT_DISTANT, LIGHT_INFINITE = 1, 2, 3, 4, 5
SAMPLER_SOBOL, SAMPLER_HALTON, SAMPLER_RANDOM, SAMPLER_ZEROTWO, SAMPLER_STRATIFIED, SAMPLER_MAXMINDIST = 1, 2, 3, 4, 5, 6
INTEGRATOR_PATH, INTEGRATOR_AO, INTEGRATOR_DIRECT, INTEGRATOR_VOLPATH = 0, 1, 2, 3
MEDIUM_HOMOGENEOUS, MEDIUM_GRID = 1, 2
DIRECT_SAMPLE_ALL, DIRECT_SAMPLE_ONE = 0, 1
LIGHTS_UNIFORM, LIGHTS_POWER, LIGHTS_SPATIAL = 0, 1, 2
TEX_CONSTANT, TEX_IMAGE, TEX_SCALE, TEX_MIX, TEX_CHECKERBOARD, TEX_DOTS, TEX_FBM, TEX_MARBLE, TEX_WINDY, TEX_WRINKLED = range(1, 11)
MAP_UV, MAP_PLANAR, MAP_SPHERICAL, MAP_CYLINDRICAL, MAP_IDENTITY3D = 1, 2, 3, 4, 5
WRAP_REPEAT, WRAP_BLACK, WRAP_CLAMP = 0, 1, 2
INSTANCING_REFERENCE, INSTANCING_FIXED = 0, 1
MESH_INSTANCE = 0xFFFFFFFF
NO_MATERIAL = 0xFFFFFFFF
MISS = 0xFFFFFFFF


class BvhNode(C.Structure):
    _fields_ = [("bmin", C.c_float * 3), ("bmax", C.c_float * 3), ("offset", C.c_int32),
                ("n_prims", C.c_uint16), ("axis", C.c_uint8), ("pad", C.c_uint8)]


class Prim(C.Structure):
    _fields_ = [("v", C.c_uint32 * 3), ("mesh", C.c_uint32), ("material", C.c_uint32), ("area_light", C.c_int32)]


class Mesh(C.Structure):
    _fields_ = [("has_n", C.c_uint32), ("has_s", C.c_uint32), ("has_uv", C.c_uint32), ("flip", C.c_uint32), ("alpha_tex", C.c_uint32), ("shadow_alpha_tex", C.c_uint32),
                ("medium_inside", C.c_uint32), ("medium_outside", C.c_uint32)]


class Medium(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("sigma_a", C.c_float * 3), ("sigma_s", C.c_float * 3), ("g", C.c_float),
                ("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32), ("pad", C.c_uint32), ("density", C.c_void_p), ("world_to_medium", C.c_float * 16)]


class Bxdf(C.Structure):
    _fields_ = [("type", C.c_uint32), ("fresnel", C.c_uint32), ("r", C.c_float * 3), ("t", C.c_float * 3),
                ("eta_a", C.c_float), ("eta_b", C.c_float), ("alpha_x", C.c_float), ("alpha_y", C.c_float),
                ("c1", C.c_float * 3), ("c2", C.c_float * 3), ("on_a", C.c_float), ("on_b", C.c_float),
                ("sc", C.c_float * 3), ("has_sc", C.c_uint32), ("tex_r", C.c_uint32), ("tex_t", C.c_uint32),
                ("tex_ax", C.c_uint32), ("tex_ay", C.c_uint32), ("remap", C.c_uint32)]


MATERIAL_DESC_FIELDS = ("kind", "kd", "ks", "kr", "kt", "reflect", "transmit", "opacity", "eta", "k", "amount", "sigma", "roughness", "uroughness", "vroughness",
                        "index", "bumpmap", "remap_roughness", "m1", "m2")


class MaterialDesc(C.Structure):
    _fields_ = [(f, C.c_uint32) for f in MATERIAL_DESC_FIELDS]


class Material(C.Structure):
    _fields_ = [("eta", C.c_float), ("first_bxdf", C.c_uint32), ("n_bxdfs", C.c_uint32), ("bump_tex", C.c_uint32)]


class Image(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("n_levels", C.c_uint32), ("channels", C.c_uint32), ("texels", C.c_void_p)]


class Texture(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("mapping", C.c_uint32), ("map", C.c_float * 8), ("image", C.c_uint32), ("trilinear", C.c_uint32),
                ("max_aniso", C.c_float), ("wrap", C.c_uint32), ("value", C.c_float * 3), ("tex1", C.c_uint32), ("tex2", C.c_uint32),
                ("tex3", C.c_uint32), ("world_to_texture", C.c_float * 16), ("octaves", C.c_int32), ("omega", C.c_float),
                ("scale", C.c_float), ("variation", C.c_float)]


class Light(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("prim", C.c_uint32), ("L", C.c_float * 3), ("two_sided", C.c_uint32), ("p", C.c_float * 24)]


class EnvMap(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("n_levels", C.c_uint32), ("pad", C.c_uint32), ("texels", C.c_void_p),
                ("dist_nu", C.c_uint32), ("dist_nv", C.c_uint32), ("dist_func", C.c_void_p)]


class SceneDesc(C.Structure):
    _fields_ = [("nodes", C.c_void_p), ("n_nodes", C.c_uint64),
                ("prims", C.c_void_p), ("n_prims", C.c_uint64),
                ("meshes", C.c_void_p), ("n_meshes", C.c_uint32),
                ("P", C.c_void_p), ("N", C.c_void_p), ("S", C.c_void_p), ("UV", C.c_void_p),
                ("n_vertices", C.c_uint64),
                ("materials", C.c_void_p), ("n_materials", C.c_uint32),
                ("lights", C.c_void_p), ("n_lights", C.c_uint32),
                ("envmaps", C.c_void_p), ("n_envmaps", C.c_uint32),
                ("textures", C.c_void_p), ("n_textures", C.c_uint32),
                ("images", C.c_void_p), ("n_images", C.c_uint32),
                ("objects", C.c_void_p), ("n_objects", C.c_uint32),
                ("instances", C.c_void_p), ("n_instances", C.c_uint32),
                ("n_top_nodes", C.c_uint64), ("n_top_prims", C.c_uint64),
                ("instancing_mode", C.c_uint32), ("n_media", C.c_uint32), ("media", C.c_void_p)]


class SamplerTables(C.Structure):
    _fields_ = [("sobol32", C.c_void_p), ("vdc", C.c_void_p), ("vdc_inv", C.c_void_p), ("halton_perms", C.c_void_p), ("n_halton_perms", C.c_uint64)]


class RenderDesc(C.Structure):
    _fields_ = [("full_res", C.c_int32 * 2), ("crop_px", C.c_int32 * 4), ("sample_bounds", C.c_int32 * 4),
                ("filter_radius", C.c_float * 2), ("filter_table", C.c_float * 256),
                ("max_sample_luminance", C.c_float),
                ("raster_to_camera", C.c_float * 16), ("camera_to_world", C.c_float * 16),
                ("lens_radius", C.c_float), ("focal_distance", C.c_float),
                ("shutter_open", C.c_float), ("shutter_close", C.c_float),
                ("sampler_kind", C.c_uint32), ("spp", C.c_int64),
                ("max_depth", C.c_uint32), ("rr_threshold", C.c_float), ("light_strategy", C.c_uint32),
                ("tile_size", C.c_uint32),
                ("shard_index", C.c_uint32), ("shard_count", C.c_uint32), ("tile_chunk", C.c_uint32),
                ("sample_at_pixel_center", C.c_uint32), ("integrator", C.c_uint32), ("ao_n_samples", C.c_uint32),
                ("ao_cos_sample", C.c_uint32), ("film_reduce", C.c_uint32), ("tables", SamplerTables),
                ("direct_strategy", C.c_uint32), ("pixel_dimensions", C.c_uint32), ("n_light_samples", C.c_void_p),
                ("strat_x", C.c_uint32), ("strat_y", C.c_uint32), ("strat_jitter", C.c_uint32), ("allow_slow_paths", C.c_uint32), ("maxmin_c_pixel", C.c_void_p),
                ("sample_begin", C.c_uint64), ("sample_count", C.c_uint64),
                ("camera_animated", C.c_uint32), ("camera_to_world_end", C.c_float * 16), ("camera_time", C.c_float * 2)]


class Ray(C.Structure):
    _fields_ = [("o", C.c_float * 3), ("d", C.c_float * 3), ("t_max", C.c_float), ("id", C.c_uint32)]


class Hit(C.Structure):
    _fields_ = [("prim", C.c_uint32), ("t", C.c_float), ("b0", C.c_float), ("b1", C.c_float), ("b2", C.c_float)]


class Stats(C.Structure):
    _fields_ = [("t_render_s", C.c_double), ("t_kernels_s", C.c_double), ("t_trace_s", C.c_double),
                ("samples", C.c_uint64), ("rays_closest", C.c_uint64), ("rays_any", C.c_uint64),
                ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64), ("nan_samples", C.c_uint64),
                ("trace_launches", C.c_uint64), ("alg_bytes", C.c_double),
                ("t_trace_closest_s", C.c_double), ("t_trace_any_s", C.c_double), ("t_shade_s", C.c_double),
                ("launches_closest", C.c_uint64), ("launches_any", C.c_uint64), ("truncated_paths", C.c_uint64)]


# numpy dtypes with the same layout (for bulk construction)
import numpy as np  # noqa: E402

NODE_DT = np.dtype([("bmin", "<f4", 3), ("bmax", "<f4", 3), ("offset", "<i4"), ("n_prims", "<u2"), ("axis", "u1"), ("pad", "u1")])
PRIM_DT = np.dtype([("v", "<u4", 3), ("mesh", "<u4"), ("material", "<u4"), ("area_light", "<i4")])
MESH_DT = np.dtype([("has_n", "<u4"), ("has_s", "<u4"), ("has_uv", "<u4"), ("flip", "<u4"), ("alpha_tex", "<u4"), ("shadow_alpha_tex", "<u4"),
                    ("medium_inside", "<u4"), ("medium_outside", "<u4")])
MEDIUM_DT = np.dtype([("kind", "<u4"), ("sigma_a", "<f4", 3), ("sigma_s", "<f4", 3), ("g", "<f4"), ("nx", "<i4"), ("ny", "<i4"), ("nz", "<i4"), ("pad", "<u4"),
                      ("density", "<u8"), ("world_to_medium", "<f4", 16)])
BXDF_DT = np.dtype([("type", "<u4"), ("fresnel", "<u4"), ("r", "<f4", 3), ("t", "<f4", 3), ("eta_a", "<f4"), ("eta_b", "<f4"),
                    ("alpha_x", "<f4"), ("alpha_y", "<f4"), ("c1", "<f4", 3), ("c2", "<f4", 3), ("on_a", "<f4"), ("on_b", "<f4"),
                    ("sc", "<f4", 3), ("has_sc", "<u4"), ("tex_r", "<u4"), ("tex_t", "<u4"), ("tex_ax", "<u4"), ("tex_ay", "<u4"), ("remap", "<u4")])
MATERIAL_DESC_DT = np.dtype([(f, "<u4") for f in MATERIAL_DESC_FIELDS])
MATERIAL_DT = np.dtype([("eta", "<f4"), ("first_bxdf", "<u4"), ("n_bxdfs", "<u4"), ("bump_tex", "<u4")])
TEXTURE_DT = np.dtype([("kind", "<u4"), ("mapping", "<u4"), ("map", "<f4", 8), ("image", "<u4"), ("trilinear", "<u4"), ("max_aniso", "<f4"),
                       ("wrap", "<u4"), ("value", "<f4", 3), ("tex1", "<u4"), ("tex2", "<u4"), ("tex3", "<u4"), ("world_to_texture", "<f4", 16),
                       ("octaves", "<i4"), ("omega", "<f4"), ("scale", "<f4"), ("variation", "<f4")])
LIGHT_DT = np.dtype([("kind", "<u4"), ("prim", "<u4"), ("L", "<f4", 3), ("two_sided", "<u4"), ("p", "<f4", 24)])
OBJECT_DT = np.dtype([("first_node", "<u8"), ("n_nodes", "<u8"), ("first_prim", "<u8"), ("n_prims", "<u8")])
INSTANCE_DT = np.dtype([("object", "<u4"), ("to_world", "<f4", 16), ("from_world", "<f4", 16), ("animated", "<u4"), ("to_world_end", "<f4", 16), ("from_world_end", "<f4", 16), ("time", "<f4", 2)])
RAY_DT = np.dtype([("o", "<f4", 3), ("d", "<f4", 3), ("t_max", "<f4"), ("id", "<u4")])
HIT_DT = np.dtype([("prim", "<u4"), ("t", "<f4"), ("b0", "<f4"), ("b1", "<f4"), ("b2", "<f4")])

assert NODE_DT.itemsize == C.sizeof(BvhNode) == 32
assert MEDIUM_DT.itemsize == C.sizeof(Medium) == 120
assert PRIM_DT.itemsize == C.sizeof(Prim) == 24
assert BXDF_DT.itemsize == C.sizeof(Bxdf) == 116
assert MATERIAL_DESC_DT.itemsize == C.sizeof(MaterialDesc) == 80
assert TEXTURE_DT.itemsize == C.sizeof(Texture) == 160
assert RAY_DT.itemsize == C.sizeof(Ray) == 32
assert HIT_DT.itemsize == C.sizeof(Hit) == 20
assert LIGHT_DT.itemsize == C.sizeof(Light) == 120
