//! src/gpu/ffi.rs — the `extern "C"` block for librspt.so, mirroring include/rspt.h (ABI version 21) one to one.
//! Uncompiled source for a maintainer (the image this repo is built in has no Rust toolchain); struct layouts are checked
//! from the C side by tests/test_abi.py, so a mismatch here shows up as a wrong `size_of` against the table in INTEGRATION.md §2.
#![allow(dead_code)]
use std::os::raw::{c_char, c_int, c_void};

pub const RSPT_ABI_VERSION: c_int = 21;
pub const RSPT_MESH_INSTANCE: u32 = 0xffff_ffff;
pub const RSPT_NO_MATERIAL: u32 = 0xffff_ffff;

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct RsptBvhNode { pub bmin: [f32; 3], pub bmax: [f32; 3], pub offset: i32, pub n_prims: u16, pub axis: u8, pub pad: u8 } // 32 B
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct RsptPrim { pub v: [u32; 3], pub mesh: u32, pub material: u32, pub area_light: i32 } // 24 B
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct RsptMesh { pub has_n: u32, pub has_s: u32, pub has_uv: u32, pub flip: u32, pub alpha_tex: u32, pub shadow_alpha_tex: u32,
                      pub medium_inside: u32, pub medium_outside: u32 }   // 0 = none, else 1 + index into media
#[repr(C)] #[derive(Clone, Copy)]
pub struct RsptMedium { pub kind: u32, pub sigma_a: [f32; 3], pub sigma_s: [f32; 3], pub g: f32,                 // kind 1 = HomogeneousMedium
                        pub nx: i32, pub ny: i32, pub nz: i32, pub pad: u32, pub density: *const f32, pub world_to_medium: [f32; 16] }   // kind 2 = GridDensityMedium
/// rspt_material_desc: kind (1 matte 2 plastic 3 mirror 4 glass 5 metal 6 substrate 7 uber 8 translucent 9 mix) and, per parameter,
/// 0 (absent) or 1 + index of its texture record — a ConstantTexture for a literal value, as TextureParams builds one
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct RsptMaterialDesc { // 80 B
    pub kind: u32, pub kd: u32, pub ks: u32, pub kr: u32, pub kt: u32, pub reflect: u32, pub transmit: u32, pub opacity: u32, pub eta: u32, pub k: u32,
    pub amount: u32, pub sigma: u32, pub roughness: u32, pub uroughness: u32, pub vroughness: u32, pub index: u32, pub bumpmap: u32,
    pub remap_roughness: u32, pub m1: u32, pub m2: u32,
}
#[repr(C)] pub struct RsptImage { pub width: u32, pub height: u32, pub n_levels: u32, pub channels: u32, pub texels: *const f32 }
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct RsptTexture { // 160 B
    pub kind: u32, pub mapping: u32, pub map: [f32; 8], pub image: u32, pub trilinear: u32, pub max_aniso: f32, pub wrap: u32,
    pub value: [f32; 3], pub tex1: u32, pub tex2: u32, pub tex3: u32, pub world_to_texture: [f32; 16], pub octaves: i32,
    pub omega: f32, pub scale: f32, pub variation: f32,
}
#[repr(C)] #[derive(Clone, Copy)]
pub struct RsptLight { pub kind: u32, pub prim: u32, pub l: [f32; 3], pub two_sided: u32, pub p: [f32; 24] } // 120 B
#[repr(C)] pub struct RsptEnvMap { pub width: u32, pub height: u32, pub n_levels: u32, pub pad: u32, pub texels: *const f32,
                                   pub dist_nu: u32, pub dist_nv: u32, pub dist_func: *const f32 }
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct RsptObject { pub first_node: u64, pub n_nodes: u64, pub first_prim: u64, pub n_prims: u64 }
#[repr(C)] #[derive(Clone, Copy)]
pub struct RsptInstance { pub object: u32, pub to_world: [f32; 16], pub from_world: [f32; 16], pub animated: u32, pub to_world_end: [f32; 16], pub from_world_end: [f32; 16], pub time: [f32; 2] } // 272 B
#[repr(C)] pub struct RsptSceneDesc {
    pub nodes: *const RsptBvhNode, pub n_nodes: u64, pub prims: *const RsptPrim, pub n_prims: u64,
    pub meshes: *const RsptMesh, pub n_meshes: u32, pub p: *const f32, pub n: *const f32, pub s: *const f32, pub uv: *const f32,
    pub n_vertices: u64, pub materials: *const RsptMaterialDesc, pub n_materials: u32,
    pub lights: *const RsptLight, pub n_lights: u32, pub envmaps: *const RsptEnvMap, pub n_envmaps: u32,
    pub textures: *const RsptTexture, pub n_textures: u32, pub images: *const RsptImage, pub n_images: u32,
    pub objects: *const RsptObject, pub n_objects: u32, pub instances: *const RsptInstance, pub n_instances: u32,
    pub n_top_nodes: u64, pub n_top_prims: u64, pub instancing_mode: u32, pub n_media: u32, pub media: *const RsptMedium,
}
#[repr(C)] pub struct RsptSamplerTables { pub sobol32: *const u32, pub vdc: *const u64, pub vdc_inv: *const u64,
                                            pub halton_perms: *const u16, pub n_halton_perms: u64 }
#[repr(C)] pub struct RsptRenderDesc {
    pub full_res: [i32; 2], pub crop_px: [i32; 4], pub sample_bounds: [i32; 4], pub filter_radius: [f32; 2], pub filter_table: [f32; 256],
    pub max_sample_luminance: f32, pub raster_to_camera: [f32; 16], pub camera_to_world: [f32; 16], pub lens_radius: f32,
    pub focal_distance: f32, pub shutter_open: f32, pub shutter_close: f32, pub sampler_kind: u32, pub spp: i64, pub max_depth: u32,
    pub rr_threshold: f32, pub light_strategy: u32, pub tile_size: u32, pub shard_index: u32, pub shard_count: u32, pub tile_chunk: u32,
    pub sample_at_pixel_center: u32, pub integrator: u32, pub ao_n_samples: u32, pub ao_cos_sample: u32, pub film_reduce: u32,
    pub tables: RsptSamplerTables,
    pub direct_strategy: u32, pub pixel_dimensions: u32, pub n_light_samples: *const i32,
    pub strat_x: u32, pub strat_y: u32, pub strat_jitter: u32, pub allow_slow_paths: u32, pub maxmin_c_pixel: *const u32,   // pixel samplers
    pub sample_begin: u64, pub sample_count: u64,   // checkpoint / resume: 0, 0 = the whole frame
    pub camera_animated: u32, pub camera_to_world_end: [f32; 16], pub camera_time: [f32; 2],   // AnimatedTransform camera (transform.rs:894-2124)
}
#[repr(C)] #[derive(Default)]
pub struct RsptStats {
    pub t_render_s: f64, pub t_kernels_s: f64, pub t_trace_s: f64, pub samples: u64, pub rays_closest: u64, pub rays_any: u64,
    pub nodes_visited: u64, pub tris_tested: u64, pub nan_samples: u64, pub trace_launches: u64, pub alg_bytes: f64,
    pub t_trace_closest_s: f64, pub t_trace_any_s: f64, pub t_shade_s: f64, pub launches_closest: u64, pub launches_any: u64,
    pub truncated_paths: u64,
}
pub enum RsptSceneOpaque {}

#[link(name = "rspt")]
extern "C" {
    pub fn rspt_abi_version() -> c_int;
    pub fn rspt_init(device: i32) -> c_int;
    pub fn rspt_shutdown();
    pub fn rspt_scene_create(desc: *const RsptSceneDesc, out: *mut *mut RsptSceneOpaque) -> c_int;
    pub fn rspt_scene_destroy(scene: *mut RsptSceneOpaque) -> c_int;
    pub fn rspt_render(scene: *mut RsptSceneOpaque, desc: *const RsptRenderDesc, film_xyzw: *mut f32, stats: *mut RsptStats) -> c_int;
    pub fn rspt_render_device(scene: *mut RsptSceneOpaque, desc: *const RsptRenderDesc, film_dev: *mut c_void, stats: *mut RsptStats) -> c_int;
    pub fn rspt_dev_alloc(bytes: u64, out: *mut *mut c_void) -> c_int;
    pub fn rspt_dev_free(p: *mut c_void) -> c_int;
    pub fn rspt_dev_download(dst_host: *mut c_void, src_dev: *const c_void, bytes: u64) -> c_int;
    pub fn rspt_comm_unique_id(id: *mut u8) -> c_int;                              // 128 bytes
    pub fn rspt_comm_init(rank: i32, world: i32, id: *const u8) -> c_int;
    pub fn rspt_comm_destroy() -> c_int;
    pub fn rspt_comm_library() -> *const c_char;                                   // which librccl the calls above are bound to (ABI 21)
    pub fn rspt_source_hash() -> *const c_char;                                    // hash of the kernel sources compiled into the library (ABI 21)
    // AnimatedTransform::motion_bounds (transform.rs:2147-2210) on the host, for a caller that builds the top-level tree itself; the shim passes rs_pbrt's own BVH and
    // does not need it (ABI 21)
    pub fn rspt_motion_bounds(start_m: *const f32, start_time: f32, end_m: *const f32, end_time: f32, box_min: *const f32, box_max: *const f32,
                              out_min: *mut f32, out_max: *mut f32, flags_out: *mut i32) -> c_int;
    pub fn rspt_last_error() -> *const c_char;
    pub fn rspt_libm(func: u32, x: *const f32, y: *const f32, n: u64, out: *mut f32) -> c_int;   // 0 sin 1 cos 2 ln 3 log2 4 exp 5 acos 6 atan2(x, y) 7 Matrix4x4::inverse of n 4x4 matrices (x, out: 16 n floats); 8 .. 12: the triangle / box / offset_ray_origin / microfacet / vector hooks of rspt.h (16 floats per element)
    pub fn rspt_bvh_build_gpu(p: *const f32, n_vertices: u64, tri_idx: *const u32, n_tris: u64, max_prims_in_node: u32,
                              nodes_out: *mut RsptBvhNode, nodes_cap: u64, ordered_out: *mut u32) -> i64;
}
