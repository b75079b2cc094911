//! src/gpu/refdump.rs — turns a real rs_pbrt run into the fixtures that pin this repo's CPU oracle (the reference
//! ships no tests or golden vectors for the render path, and the image this repo is built in has no Rust toolchain, so the oracle
//! is "parity unpinned" until someone runs this once).  Uncompiled source for a maintainer.
//!
//!     git apply rust_shim/rs_pbrt.patch && cp rust_shim/{gpu,ffi,refdump}.rs src/gpu/   (gpu.rs -> src/gpu/mod.rs)
//!     RSPT_REF_DUMP=out/cornell_matte RSPT_REF_DUMP_LI=1 cargo run --release --bin rs_pbrt -- -t 1 tests/golden/ref_scenes/cornell_matte.pbrt
//!     python tools/ref_to_npz.py out/cornell_matte cornell_matte      ->  tests/golden/ref_cornell_matte.npz
//!     python -m pytest tests/test_reference_fixtures.py               (consumes every tests/golden/ref_*.npz it finds)
//!
//! Files written into $RSPT_REF_DUMP (little endian, no headers; shapes in meta.json):
//!   bvh_nodes.bin   n x 32 B   LinearBVHNode as include/rspt.h lays it out (bounds, offset, n_primitives, axis)
//!   bvh_prims.bin   n x 9 f32  the three world-space vertices of every primitive in BVHAccel.primitives order
//!   hits.bin        for every ray of $RSPT_REF_DUMP/rays.bin (n x 32 B: o, d, t_max, id; optional):
//!                   hit (f32 0/1), t, p.xyz, n.xyz, the hit triangle's 9 vertex coordinates  = 17 f32;  occluded.bin: n x u8 (intersect_p)
//!   li.bin          (RSPT_REF_DUMP_LI=1, run with -t 1) per camera sample: pixel x, y, sample number, p_film.x, p_film.y, L.r, L.g, L.b  = 8 f32
//!   film.bin        Film.pixels after the render: xyz[3], filter_weight_sum per cropped pixel
use crate::accelerators::bvh::BVHAccel;
use crate::core::geometry::{Point3f, Ray, Vector3f};
use crate::core::integrator::SamplerIntegrator;
use crate::core::interaction::SurfaceInteraction;
use crate::core::pbrt::{Float, Spectrum};
use crate::core::primitive::Primitive;
use crate::core::scene::Scene;
use crate::core::shape::Shape;
use std::cell::Cell;
use std::io::Write;
use std::sync::Mutex;

lazy_static::lazy_static! { static ref LI: Mutex<Vec<f32>> = Mutex::new(Vec::new()); }   // (the crate already depends on lazy_static? if not: std::sync::OnceLock)

pub fn dir() -> Option<String> { std::env::var("RSPT_REF_DUMP").ok() }
pub fn want_li() -> bool { dir().is_some() && std::env::var_os("RSPT_REF_DUMP_LI").is_some() }

fn put(path: String, data: &[f32]) {
    let mut f = std::fs::File::create(path).expect("ref dump: create");
    for v in data { f.write_all(&v.to_le_bytes()).unwrap(); }
}
fn tri_vertices(p: &Primitive) -> [f32; 9] {
    if let Primitive::Geometric(g) = p {
        if let Shape::Trngl(t) = &*g.shape {
            let m = t.mesh();
            let vi = &m.vertex_indices[3 * t.id as usize..3 * t.id as usize + 3];
            let (a, b, c) = (m.p[vi[0] as usize], m.p[vi[1] as usize], m.p[vi[2] as usize]);
            return [a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z];
        }
    }
    [f32::NAN; 9]                                                                   // TransformedPrimitive / other shapes
}

/// Called from SamplerIntegrator::render right after preprocess (rs_pbrt.patch).
pub fn dump_scene(_integ: &SamplerIntegrator, scene: &Scene) {
    let d = match dir() { Some(d) => d, None => return };
    std::fs::create_dir_all(&d).unwrap();
    let bvh: &BVHAccel = match &*scene.aggregate { Primitive::BVH(b) => b, _ => return };
    let mut f = std::fs::File::create(format!("{}/bvh_nodes.bin", d)).unwrap();
    for n in &bvh.nodes {
        let b = n.bounds();
        for v in [b.p_min.x, b.p_min.y, b.p_min.z, b.p_max.x, b.p_max.y, b.p_max.z] { f.write_all(&v.to_le_bytes()).unwrap(); }
        f.write_all(&n.offset().to_le_bytes()).unwrap();
        f.write_all(&n.n_primitives().to_le_bytes()).unwrap();
        f.write_all(&[n.axis(), 0u8]).unwrap();
    }
    let mut prims: Vec<f32> = Vec::new();
    for p in &bvh.primitives { prims.extend_from_slice(&tri_vertices(p)); }
    put(format!("{}/bvh_prims.bin", d), &prims);
    if let Ok(bytes) = std::fs::read(format!("{}/rays.bin", d)) {                    // stage-level fixtures: Scene::intersect / intersect_p
        let n = bytes.len() / 32;
        let fl = |i: usize| f32::from_le_bytes([bytes[4 * i], bytes[4 * i + 1], bytes[4 * i + 2], bytes[4 * i + 3]]);
        let mut hits: Vec<f32> = Vec::with_capacity(17 * n);
        let mut occ: Vec<u8> = Vec::with_capacity(n);
        for r in 0..n {
            let mk = || Ray { o: Point3f { x: fl(8 * r), y: fl(8 * r + 1), z: fl(8 * r + 2) }, d: Vector3f { x: fl(8 * r + 3), y: fl(8 * r + 4), z: fl(8 * r + 5) },
                              t_max: Cell::new(fl(8 * r + 6)), time: 0.0 as Float, differential: None, medium: None };
            let ray = mk();
            let mut isect = SurfaceInteraction::default();
            if scene.intersect(&ray, &mut isect) {
                let tv = isect.primitive.map(|p| tri_vertices(unsafe { &*p }.as_primitive_for_shim())).unwrap_or([f32::NAN; 9]); // helper: rs_pbrt.patch
                hits.extend_from_slice(&[1.0, ray.t_max.get(), isect.common.p.x, isect.common.p.y, isect.common.p.z, isect.common.n.x, isect.common.n.y, isect.common.n.z]);
                hits.extend_from_slice(&tv);
            } else { hits.extend_from_slice(&[0.0; 17]); }
            occ.push(scene.intersect_p(&mk()) as u8);
        }
        put(format!("{}/hits.bin", d), &hits);
        std::fs::write(format!("{}/occluded.bin", d), &occ).unwrap();
    }
}

/// Called from the sample loop (rs_pbrt.patch) with what `li` returned, before the NaN / luminance checks.
pub fn record_li(px: i32, py: i32, sample: i64, p_film_x: Float, p_film_y: Float, l: &Spectrum) {
    if want_li() { LI.lock().unwrap().extend_from_slice(&[px as f32, py as f32, sample as f32, p_film_x, p_film_y, l.c[0], l.c[1], l.c[2]]); }
}

/// Called after the tile loop, before write_image (rs_pbrt.patch).
pub fn dump_film(integ: &SamplerIntegrator) {
    let d = match dir() { Some(d) => d, None => return };
    let film = integ.get_camera().get_film();
    let px = film.pixels.read().unwrap();
    let mut out: Vec<f32> = Vec::with_capacity(4 * px.len());
    for p in px.iter() { let (xyz, w) = p.xyz_weight(); out.extend_from_slice(&[xyz[0], xyz[1], xyz[2], w]); }   // getter: rs_pbrt.patch
    put(format!("{}/film.bin", d), &out);
    put(format!("{}/li.bin", d), &LI.lock().unwrap());
    let cb = film.cropped_pixel_bounds; let sb = film.get_sample_bounds();
    std::fs::write(format!("{}/meta.json", d), format!(
        "{{\"full_res\": [{}, {}], \"crop_px\": [{}, {}, {}, {}], \"sample_bounds\": [{}, {}, {}, {}], \"spp\": {}, \"rs_pbrt\": \"{}\"}}\n",
        film.full_resolution.x, film.full_resolution.y, cb.p_min.x, cb.p_min.y, cb.p_max.x, cb.p_max.y, sb.p_min.x, sb.p_min.y, sb.p_max.x, sb.p_max.y,
        integ.get_sampler().get_samples_per_pixel(), env!("CARGO_PKG_VERSION"))).unwrap();
}
