//! src/gpu/mod.rs — the whole Rust side of the MI355X path: flatten `Scene` + `PathIntegrator` into the POD arrays of
//! include/rspt.h, call librspt.so, write the film back.  Uncompiled source for a maintainer (no Rust toolchain in the image this
//! repo is built in); it is written against rs_pbrt v0.9.12 plus the getters of rust_shim/rs_pbrt.patch.  Everything not covered
//! returns Err and `SamplerIntegrator::render` keeps its CPU tile loop (src/core/integrator.rs:70-220).
//!
//! Covered: triangle meshes (Shape::Trngl) under a BVHAccel aggregate, object instances (Primitive::Transformed, static or moving),
//! matte / plastic / mirror / glass (smooth and rough) / metal / substrate / uber / translucent / mix (handed over as their
//! parameters — one texture reference each, rspt_material_desc; the library assembles the lobes), diffuse area / point / spot / distant / infinite lights (the light's own MIP
//! pyramid and Distribution2D image are handed over), homogeneous media, PerspectiveCamera, the Sobol', Halton and the four
//! PCG-backed pixel samplers, the path / ao / directlighting / volpath integrators, any filter (through Film.filter_table).
//! Image / procedural textures (every class of src/textures/, through Texture::describe of rs_pbrt.patch) go over as texture
//! graphs wherever a material or mesh refers to one; any parameter may vary over a surface (include/rspt.h: Kd / Ks / roughness
//! scale the lobes of a list folded once per material, any other varying parameter has the library build the lobe list per hit;
//! mixes may nest).  What the library refuses (more than 8 non-mix materials under one mix, more than 12 varying textures on such a
//! material, more than 8 BxDFs) comes back as RSPT_E_UNSUPPORTED from rspt_scene_create and the scene then keeps the CPU loop.
pub mod ffi;
pub mod refdump;

use self::ffi::*;
use crate::accelerators::bvh::BVHAccel;
use crate::core::camera::Camera;
use crate::core::integrator::SamplerIntegrator;
use crate::core::light::Light;
use crate::core::material::Material;
use crate::core::medium::{Medium, MediumInterface};
use crate::core::pbrt::{Float, Spectrum};
use crate::core::primitive::Primitive;
use crate::core::sampler::Sampler;
use crate::core::scene::Scene;
use crate::core::shape::Shape;
use crate::core::sobolmatrices::{SOBOL_MATRICES_32, VD_C_SOBOL_MATRICES, VD_C_SOBOL_MATRICES_INV};
use crate::core::texture::{TexDesc, Texture, TextureMapping2D, TextureMapping3D};
use crate::core::mipmap::{ImageWrap, MipMap};
use crate::core::transform::Transform;
use crate::samplers::halton::RADICAL_INVERSE_PERMUTATIONS;
use crate::shapes::triangle::TriangleMesh;
use std::collections::HashMap;
use std::ffi::CStr;
use std::sync::Arc;

fn m16(t: &crate::core::transform::Matrix4x4) -> [f32; 16] {
    let mut o = [0.0f32; 16];
    for r in 0..4 { for c in 0..4 { o[4 * r + c] = t.m[r][c]; } }
    o
}
fn rgb(s: &Spectrum) -> [f32; 3] { [s.c[0].max(0.0), s.c[1].max(0.0), s.c[2].max(0.0)] } // Spectrum::clamp_t(0, inf)
fn err() -> String { unsafe { CStr::from_ptr(rspt_last_error()).to_string_lossy().into_owned() } }

/// Float or Spectrum as the ABI stores them (rspt_texture.value, rspt_image.channels)
trait ShimValue { const CHANNELS: u32; fn rgb3(&self) -> [f32; 3]; fn push(&self, out: &mut Vec<f32>); }
impl ShimValue for Float { const CHANNELS: u32 = 1; fn rgb3(&self) -> [f32; 3] { [*self; 3] } fn push(&self, out: &mut Vec<f32>) { out.push(*self); } }
impl ShimValue for Spectrum { const CHANNELS: u32 = 3; fn rgb3(&self) -> [f32; 3] { self.c } fn push(&self, out: &mut Vec<f32>) { out.extend_from_slice(&self.c); } }
const IDENTITY16: [f32; 16] = [1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0];
type SpecTex = Arc<dyn Texture<Spectrum> + Sync + Send>;
type FloatTex = Arc<dyn Texture<Float> + Sync + Send>;

/// Everything that has to stay alive until rspt_scene_create has copied it.
#[derive(Default)]
struct Flat {
    nodes: Vec<RsptBvhNode>, prims: Vec<RsptPrim>, meshes: Vec<RsptMesh>,
    p: Vec<f32>, n: Vec<f32>, s: Vec<f32>, uv: Vec<f32>, any_n: bool, any_s: bool, any_uv: bool,
    materials: Vec<RsptMaterialDesc>, lights: Vec<RsptLight>,
    objects: Vec<RsptObject>, instances: Vec<RsptInstance>,
    envmaps: Vec<RsptEnvMap>, env_texels: Vec<Vec<f32>>, env_dist: Vec<Vec<f32>>,   // the Vecs own what the RsptEnvMap pointers refer to
    textures: Vec<RsptTexture>, images: Vec<RsptImage>, image_texels: Vec<Vec<f32>>,   // image_texels owns what RsptImage.texels points to
    texture_of: HashMap<*const u8, u32>, image_of: HashMap<*const u8, u32>,           // texture / MipMap object -> index (named textures are shared)
    mesh_of: HashMap<*const TriangleMesh, (u32, u32)>,     // mesh -> (mesh index, first vertex)
    material_of: HashMap<*const Material, u32>,
    object_of: HashMap<*const Primitive, u32>,               // instanced aggregate / primitive -> object index
    media: Vec<RsptMedium>,
    medium_of: HashMap<*const Medium, u32>,                  // named medium -> 1 + index (the reference compares these pointers, medium.rs:340-362)
}

impl Flat {
    /// 0 = None, else 1 + index into media (MakeNamedMedium api.rs:953-1037 leaves sigma_a / sigma_s already scaled)
    fn medium(&mut self, m: &Option<Arc<Medium>>) -> Result<u32, String> {
        let m = match m { Some(m) => m, None => return Ok(0) };
        let key = Arc::as_ptr(m);
        if let Some(i) = self.medium_of.get(&key) { return Ok(*i); }
        let md = match &**m {
            Medium::Homogeneous(h) => RsptMedium { kind: 1, sigma_a: rgb(&h.sigma_a), sigma_s: rgb(&h.sigma_s), g: h.g,
                                                   nx: 0, ny: 0, nz: 0, pad: 0, density: std::ptr::null(), world_to_medium: [0.0; 16] },
            // GridDensityMedium (grid.rs:17-56): the density Arc stays alive in the scene for the duration of the call; the library copies it
            Medium::GridDensity(gd) => RsptMedium { kind: 2, sigma_a: rgb(&gd.sigma_a), sigma_s: rgb(&gd.sigma_s), g: gd.g, nx: gd.nx, ny: gd.ny, nz: gd.nz, pad: 0,
                                                    density: gd.density.as_ptr(), world_to_medium: m16(&gd.world_to_medium.m) },
        };
        self.media.push(md);
        self.medium_of.insert(key, self.media.len() as u32);
        Ok(self.media.len() as u32)
    }

    /// One MipMap pyramid as rs_pbrt built it (mipmap.rs:56-196; texels already scaled / inverse-gamma-corrected and flipped by
    /// ImageTexture::new, imagemap.rs:34-96), un-blocked: levels concatenated, row major [t][s]
    fn image<T: ShimValue + num::Zero + Clone + std::ops::Add<T, Output = T> + 'static>(&mut self, mip: &Arc<MipMap<T>>) -> u32 {
        let key = Arc::as_ptr(mip) as *const u8;
        if let Some(i) = self.image_of.get(&key) { return *i; }
        let mut tex: Vec<f32> = Vec::new();
        for lvl in &mip.pyramid {
            for t in 0..lvl.v_size() { for s in 0..lvl.u_size() { lvl[(s, t)].push(&mut tex); } }
        }
        self.image_texels.push(tex);
        let tp = self.image_texels.last().unwrap().as_ptr();
        self.images.push(RsptImage { width: mip.width() as u32, height: mip.height() as u32, n_levels: mip.levels() as u32, channels: T::CHANNELS, texels: tp });
        let i = (self.images.len() - 1) as u32;
        self.image_of.insert(key, i);
        i
    }

    /// A texture graph as rspt_texture records (children first; indices are 0-based here, 1-based where a lobe / material / mesh refers to one)
    fn texture<T: ShimValue + num::Zero + Clone + std::ops::Add<T, Output = T> + 'static>(&mut self, t: &Arc<dyn Texture<T> + Sync + Send>) -> Result<u32, String> {
        let key = Arc::as_ptr(t) as *const u8;
        if let Some(i) = self.texture_of.get(&key) { return Ok(*i); }
        let mut rec = RsptTexture { world_to_texture: IDENTITY16, max_aniso: 8.0, ..RsptTexture::default() };
        fn map2d(m: &TextureMapping2D, rec: &mut RsptTexture) {
            match m {
                TextureMapping2D::UV(u) => { rec.mapping = 1; rec.map[..4].copy_from_slice(&[u.su, u.sv, u.du, u.dv]); }          // texture.rs:91-121
                TextureMapping2D::Planar(p) => { rec.mapping = 2; rec.map = [p.vs.x, p.vs.y, p.vs.z, p.vt.x, p.vt.y, p.vt.z, p.ds, p.dt]; } // :222-257
                TextureMapping2D::Spherical(sp) => { rec.mapping = 3; rec.world_to_texture = m16(&sp.world_to_texture.m); }       // :123-170
                TextureMapping2D::Cylindrical(c) => { rec.mapping = 4; rec.world_to_texture = m16(&c.world_to_texture.m); }       // :172-220
            }
        }
        fn map3d(m: &TextureMapping3D, rec: &mut RsptTexture) {
            match m { TextureMapping3D::Identity(i) => { rec.mapping = 5; rec.world_to_texture = m16(&i.world_to_texture.m); } }   // :259-283
        }
        match t.describe().ok_or("texture class without a GPU form")? {
            TexDesc::Constant(v) => { rec.kind = 1; rec.value = v.rgb3(); }
            TexDesc::Image(m, mip) => {
                rec.kind = 2; map2d(m, &mut rec);
                rec.image = self.image(mip); rec.trilinear = mip.do_trilinear as u32; rec.max_aniso = mip.max_anisotropy;
                rec.wrap = match mip.wrap_mode { ImageWrap::Repeat => 0, ImageWrap::Black => 1, ImageWrap::Clamp => 2 };
            }
            TexDesc::Scale(a, b) => { rec.kind = 3; rec.tex1 = self.texture(a)?; rec.tex2 = self.texture(b)?; }
            TexDesc::Mix(a, b, amount) => { rec.kind = 4; rec.tex1 = self.texture(a)?; rec.tex2 = self.texture(b)?; rec.tex3 = self.texture::<Float>(amount)?; }
            TexDesc::Checkerboard(m, a, b) => { rec.kind = 5; map2d(m, &mut rec); rec.tex1 = self.texture(a)?; rec.tex2 = self.texture(b)?; }
            TexDesc::Dots(m, outside, inside) => { rec.kind = 6; map2d(m, &mut rec); rec.tex1 = self.texture(outside)?; rec.tex2 = self.texture(inside)?; }
            TexDesc::FBm(m, octaves, omega) => { rec.kind = 7; map3d(m, &mut rec); rec.octaves = octaves; rec.omega = omega; }
            TexDesc::Marble(m, octaves, omega, scale, variation) => { rec.kind = 8; map3d(m, &mut rec); rec.octaves = octaves; rec.omega = omega; rec.scale = scale; rec.variation = variation; }
            TexDesc::Windy(m) => { rec.kind = 9; map3d(m, &mut rec); }
            TexDesc::Wrinkled(m, octaves, omega) => { rec.kind = 10; map3d(m, &mut rec); rec.octaves = octaves; rec.omega = omega; }
        }
        self.textures.push(rec);
        let i = (self.textures.len() - 1) as u32;
        self.texture_of.insert(key, i);
        Ok(i)
    }

    /// a float texture referenced as 0 (absent) or 1 + index: bump maps, alpha masks, the *_or_null roughness parameters
    fn bump(&mut self, b: &Option<FloatTex>) -> Result<u32, String> { match b { Some(t) => Ok(self.texture(t)? + 1), None => Ok(0) } }

    /// all triangles of a mesh come from one Shape statement and share its MediumInterface (api.rs:2858-2870)
    fn mesh(&mut self, m: &Arc<TriangleMesh>, mi: &Option<Arc<MediumInterface>>) -> Result<(u32, u32), String> {
        let key = Arc::as_ptr(m);
        if let Some(v) = self.mesh_of.get(&key) { return Ok(*v); }
        let (medium_inside, medium_outside) = match mi { Some(i) => (self.medium(&i.inside)?, self.medium(&i.outside)?), None => (0, 0) };
        let alpha_tex = self.bump(&m.alpha_mask)?; let shadow_alpha_tex = self.bump(&m.shadow_alpha_mask)?;   // same encoding as a bump map: 0 or 1 + index
        let first = (self.p.len() / 3) as u32;
        for q in &m.p { self.p.extend_from_slice(&[q.x, q.y, q.z]); }          // world space already (api.rs:1967-1971)
        for i in 0..m.p.len() {
            if m.n.is_empty() { self.n.extend_from_slice(&[0.0; 3]); } else { self.n.extend_from_slice(&[m.n[i].x, m.n[i].y, m.n[i].z]); }
            if m.s.is_empty() { self.s.extend_from_slice(&[0.0; 3]); } else { self.s.extend_from_slice(&[m.s[i].x, m.s[i].y, m.s[i].z]); }
            if m.uv.is_empty() { self.uv.extend_from_slice(&[0.0; 2]); } else { self.uv.extend_from_slice(&[m.uv[i].x, m.uv[i].y]); }
        }
        self.any_n |= !m.n.is_empty(); self.any_s |= !m.s.is_empty(); self.any_uv |= !m.uv.is_empty();
        self.meshes.push(RsptMesh { has_n: !m.n.is_empty() as u32, has_s: !m.s.is_empty() as u32, has_uv: !m.uv.is_empty() as u32,
                                    flip: (m.reverse_orientation ^ m.transform_swaps_handedness) as u32, // triangle.rs:324
                                    alpha_tex, shadow_alpha_tex,            // "alpha" / "shadowalpha" float textures (triangle.rs:39-40), 1-based
                                    medium_inside, medium_outside });
        let v = ((self.meshes.len() - 1) as u32, first);
        self.mesh_of.insert(key, v);
        Ok(v)
    }

    /// A material as the record the library assembles its lobes from (rspt_material_desc): the kind and one texture reference per
    /// parameter — exactly what the material structs hold (every parameter of src/materials/*.rs is an Arc<dyn Texture>; a literal
    /// value is a ConstantTexture, paramset.rs:622-735).  No recipe lives on this side: Material::compute_scattering_functions is
    /// restated by librspt (csrc/material_assembly.h) and decides there what it can take; a refusal comes back from
    /// rspt_scene_create as RSPT_E_UNSUPPORTED and the scene keeps the CPU loop.
    fn material(&mut self, m: &Option<Arc<Material>>) -> Result<u32, String> {
        let m = match m { Some(m) => m, None => return Ok(RSPT_NO_MATERIAL) };   // path.rs:109-116 passes straight through
        let key = Arc::as_ptr(m);
        if let Some(i) = self.material_of.get(&key) { return Ok(*i); }
        let mut d = RsptMaterialDesc::default();
        macro_rules! spec { ($t:expr) => { self.texture::<Spectrum>($t)? + 1 } }
        macro_rules! flt { ($t:expr) => { self.texture::<Float>($t)? + 1 } }
        match &**m {
            Material::Matte(x) => { d.kind = 1; d.kd = spec!(&x.kd); d.sigma = flt!(&x.sigma); d.bumpmap = self.bump(&x.bump_map)?; }
            Material::Plastic(x) => { d.kind = 2; d.kd = spec!(&x.kd); d.ks = spec!(&x.ks); d.roughness = flt!(&x.roughness);
                                      d.remap_roughness = x.remap_roughness as u32; d.bumpmap = self.bump(&x.bump_map)?; }
            Material::Mirror(x) => { d.kind = 3; d.kr = spec!(&x.kr); d.bumpmap = self.bump(&x.bump_map)?; }
            Material::Glass(x) => { d.kind = 4; d.kr = spec!(&x.kr); d.kt = spec!(&x.kt); d.uroughness = flt!(&x.u_roughness); d.vroughness = flt!(&x.v_roughness);
                                    d.index = flt!(&x.index); d.remap_roughness = x.remap_roughness as u32; d.bumpmap = self.bump(&x.bump_map)?; }
            Material::Metal(x) => { d.kind = 5; d.eta = spec!(&x.eta); d.k = spec!(&x.k); d.roughness = flt!(&x.roughness); d.uroughness = self.bump(&x.u_roughness)?;
                                    d.vroughness = self.bump(&x.v_roughness)?; d.remap_roughness = x.remap_roughness as u32; d.bumpmap = self.bump(&x.bump_map)?; }
            Material::Substrate(x) => { d.kind = 6; d.kd = spec!(&x.kd); d.ks = spec!(&x.ks); d.uroughness = flt!(&x.nu); d.vroughness = flt!(&x.nv);
                                        d.remap_roughness = x.remap_roughness as u32; d.bumpmap = self.bump(&x.bump_map)?; }
            Material::Uber(x) => { d.kind = 7; d.kd = spec!(&x.kd); d.ks = spec!(&x.ks); d.kr = spec!(&x.kr); d.kt = spec!(&x.kt); d.opacity = spec!(&x.opacity);
                                   d.roughness = flt!(&x.roughness); d.uroughness = self.bump(&x.u_roughness)?; d.vroughness = self.bump(&x.v_roughness)?;
                                   d.index = flt!(&x.eta); d.remap_roughness = x.remap_roughness as u32; d.bumpmap = self.bump(&x.bump_map)?; }
            Material::Translucent(x) => { d.kind = 8; d.kd = spec!(&x.kd); d.ks = spec!(&x.ks); d.reflect = spec!(&x.reflect); d.transmit = spec!(&x.transmit);
                                          d.roughness = flt!(&x.roughness); d.remap_roughness = x.remap_roughness as u32; d.bumpmap = self.bump(&x.bump_map)?; }
            Material::Mix(x) => { d.kind = 9; d.amount = spec!(&x.scale); d.m1 = self.material(&Some(x.m1.clone()))?; d.m2 = self.material(&Some(x.m2.clone()))?; }
            _ => return Err("material kind the library has no assembly for (disney / hair / fourier / subsurface / kdsubsurface)".into()),
        }
        self.materials.push(d);
        let i = (self.materials.len() - 1) as u32;
        self.material_of.insert(key, i);
        Ok(i)
    }

    /// One BVHAccel: its LinearBVHNodes (offsets made absolute) and its ordered primitives; returns (first node, n nodes, first prim, n prims)
    fn aggregate(&mut self, bvh: &BVHAccel, lights: &[Arc<Light>], top: bool) -> Result<(u64, u64, u64, u64), String> {
        let node_base = self.nodes.len() as i32; let prim_base = self.prims.len() as i32;
        for n in &bvh.nodes {                                                      // LinearBVHNode getters: rs_pbrt.patch (bvh.rs:77-85)
            let b = n.bounds();
            let leaf = n.n_primitives() > 0;
            self.nodes.push(RsptBvhNode { bmin: [b.p_min.x, b.p_min.y, b.p_min.z], bmax: [b.p_max.x, b.p_max.y, b.p_max.z],
                                          offset: n.offset() + if leaf { prim_base } else { node_base }, n_prims: n.n_primitives(), axis: n.axis(), pad: 0 });
        }
        let mut pending: Vec<(usize, Arc<Primitive>, Transform)> = Vec::new();     // instances: flattened after this aggregate
        for prim in &bvh.primitives {                                              // BVH leaf order (bvh.rs:144-149)
            match &**prim {
                Primitive::Geometric(g) => {
                    let tri = match &*g.shape { Shape::Trngl(t) => t, _ => return Err("non-triangle shape".into()) };
                    let (mesh, first) = self.mesh(tri.mesh(), &g.medium_interface)?; // Triangle.mesh getter: rs_pbrt.patch (triangle.rs:85)
                    let vi = &tri.mesh().vertex_indices[3 * tri.id as usize..3 * tri.id as usize + 3];
                    let area_light = match &g.area_light {                         // the reference compares these pointers (integrator.rs:540-543)
                        Some(al) if top => lights.iter().position(|l| Arc::ptr_eq(l, al)).map(|i| i as i32).unwrap_or(-1),
                        Some(_) => return Err("area light inside an object instance".into()),
                        None => -1,
                    };
                    let material = self.material(&g.material)?;
                    self.prims.push(RsptPrim { v: [first + vi[0], first + vi[1], first + vi[2]], mesh, material, area_light });
                }
                Primitive::Transformed(tp) if top => {
                    // a moving instance goes over with both keys (ABI 20); getters of rs_pbrt.patch: is_animated = actually_animated,
                    // start_transform / end_transform, start_time / end_time (transform.rs:894-911)
                    let a = &tp.primitive_to_world;
                    pending.push((self.prims.len(), tp.primitive.clone(), a.start_transform(), a.end_transform(), a.is_animated(), [a.start_time(), a.end_time()]));
                    self.prims.push(RsptPrim { v: [0; 3], mesh: RSPT_MESH_INSTANCE, material: RSPT_NO_MATERIAL, area_light: -1 });
                }
                _ => return Err("nested aggregate".into()),
            }
        }
        let me = (node_base as u64, bvh.nodes.len() as u64, prim_base as u64, bvh.primitives.len() as u64);
        for (slot, obj, xf, xf_end, animated, times) in pending {
            let key = Arc::as_ptr(&obj);
            let oi = match self.object_of.get(&key) {
                Some(i) => *i,
                None => {
                    let o = match &*obj {
                        Primitive::BVH(b) => { let (fnod, nn, fp, np) = self.aggregate(b, lights, false)?; RsptObject { first_node: fnod, n_nodes: nn, first_prim: fp, n_prims: np } }
                        Primitive::Geometric(_) => {                               // a single primitive: no aggregate (api.rs:3046)
                            let fp = self.prims.len() as u64;
                            let one = BVHAccel::single_for_shim(obj.clone());      // helper in rs_pbrt.patch: primitives = [obj], nodes = []
                            self.aggregate(&one, lights, false)?;
                            RsptObject { first_node: 0, n_nodes: 0, first_prim: fp, n_prims: 1 }
                        }
                        _ => return Err("unsupported instanced primitive".into()),
                    };
                    self.objects.push(o);
                    let i = (self.objects.len() - 1) as u32;
                    self.object_of.insert(key, i);
                    i
                }
            };
            self.prims[slot].v[0] = self.instances.len() as u32;
            self.instances.push(RsptInstance { object: oi, to_world: m16(&xf.m), from_world: m16(&xf.m_inv), animated: animated as u32,
                                               to_world_end: m16(&xf_end.m), from_world_end: m16(&xf_end.m_inv), time: times });
        }
        Ok(me)
    }
}

fn light_record(f: &mut Flat, l: &Light, index: usize) -> Result<RsptLight, String> {
    let mut o = RsptLight { kind: 0, prim: 0, l: [0.0; 3], two_sided: 0, p: [0.0; 24] };
    match l {
        Light::DiffuseArea(a) => {                                                 // diffuse.rs:19-27
            o.kind = 1; o.l = a.l_emit.c; o.two_sided = a.two_sided as u32;
            o.prim = f.prims.iter().position(|p| p.area_light == index as i32).ok_or("area light without primitive")? as u32;
        }
        Light::InfiniteArea(a) => {                                                // infinite.rs:38-392
            o.kind = 5; o.prim = f.envmaps.len() as u32; o.l = [1.0; 3];          // the texels already carry L (infinite.rs:84-96)
            let (l2w, w2l) = (&a.light_to_world.m.m, &a.world_to_light.m.m);
            for r in 0..3 { for c in 0..3 { o.p[3 * r + c] = l2w[r][c]; o.p[9 + 3 * r + c] = w2l[r][c]; } }
            // the MipMap<Spectrum> the light built (power-of-two levels after MipMap::new's resampling, mipmap.rs:56-196),
            // un-blocked level by level, row major [t][s]
            let mut tex: Vec<f32> = Vec::new();
            for lvl in &a.lmap.pyramid {
                for t in 0..lvl.v_size() { for s in 0..lvl.u_size() { tex.extend_from_slice(&lvl[(s, t)].c); } }
            }
            // the scalar image behind the Distribution2D (infinite.rs:120-137) = the func rows of its conditional distributions
            let nv = a.distribution.p_conditional_v.len();
            let nu = a.distribution.p_conditional_v[0].func.len();
            let mut dist: Vec<f32> = Vec::with_capacity(nu * nv);
            for row in &a.distribution.p_conditional_v { dist.extend_from_slice(&row.func); }
            f.env_texels.push(tex); f.env_dist.push(dist);
            let (tp, dp) = (f.env_texels.last().unwrap().as_ptr(), f.env_dist.last().unwrap().as_ptr());   // heap buffers: stable while f lives
            f.envmaps.push(RsptEnvMap { width: a.lmap.width() as u32, height: a.lmap.height() as u32, n_levels: a.lmap.levels() as u32, pad: 0,
                                        texels: tp, dist_nu: nu as u32, dist_nv: nv as u32, dist_func: dp });
        }
        Light::Point(p) => { o.kind = 2; o.l = p.i.c; o.p[..3].copy_from_slice(&[p.p_light.x, p.p_light.y, p.p_light.z]); }   // point.rs:20-68
        Light::Spot(s) => {                                                        // spot.rs:20-110
            o.kind = 3; o.l = s.i.c; o.p[..3].copy_from_slice(&[s.p_light.x, s.p_light.y, s.p_light.z]);
            let w = &s.world_to_light.m.m;
            for r in 0..3 { for c in 0..3 { o.p[3 + 3 * r + c] = w[r][c]; } }
            o.p[12] = s.cos_total_width; o.p[13] = s.cos_falloff_start;
        }
        Light::Distant(d) => { o.kind = 4; o.l = d.l.c; o.p[..3].copy_from_slice(&[d.w_light.x, d.w_light.y, d.w_light.z]); } // distant.rs:25-75
        _ => return Err("light without a GPU form (projection / goniometric)".into()),
    }
    Ok(o)
}

/// What `SamplerIntegrator::render` calls first when RSPT_GPU is set.  Ok(()) = Film.pixels hold the finished frame.
pub fn render_path(integ: &SamplerIntegrator, scene: &Scene) -> Result<(), String> {
    let mut direct_strategy = 0u32; let mut n_light_samples: Vec<i32> = Vec::new();
    let (max_depth, rr_threshold, strategy, integrator_kind, ao_n, ao_cos) = match integ {
        SamplerIntegrator::Path(p) => (p.max_depth(), p.rr_threshold(), p.light_sample_strategy().to_string(), 0u32, 0u32, 0u32), // getters: rs_pbrt.patch (path.rs:30-32)
        SamplerIntegrator::AO(a) => (0, 1.0, "spatial".to_string(), 1u32, a.n_samples as u32, a.cos_sample as u32),
        SamplerIntegrator::DirectLighting(d) => {                                  // getter: rs_pbrt.patch (directlighting.rs:26-34); preprocess has run
            let (all, depth, counts) = d.shim_params();
            direct_strategy = if all { 0 } else { 1 };                            // RSPT_DIRECT_SAMPLE_ALL / _ONE
            n_light_samples = counts.to_vec();                                     // already through Sampler::round_count (:52-53)
            (depth, 1.0, "uniform".to_string(), 2u32, 0u32, 0u32)
        }
        SamplerIntegrator::VolPath(v) => (v.max_depth, v.rr_threshold, v.light_sample_strategy.clone(), 3u32, 0u32, 0u32),   // pub fields (volpath.rs:25-35)
        _ => return Err("integrator without a GPU path (whitted)".into()),
    };
    let bvh = match &*scene.aggregate { Primitive::BVH(b) => b, _ => return Err("aggregate is not a BVH".into()) };
    let mut f = Flat::default();
    let (_, n_top_nodes, _, n_top_prims) = f.aggregate(bvh, &scene.lights, true)?;
    let mut lights: Vec<RsptLight> = Vec::with_capacity(scene.lights.len());
    for (i, l) in scene.lights.iter().enumerate() { lights.push(light_record(&mut f, l, i)?); }
    f.lights = lights;

    // ---- camera / film / sampler (perspective.rs:22-43, film.rs:159-173, sobol.rs:15-20, halton.rs:54-78) ----
    let cam = match &*integ.get_camera() { Camera::Perspective(c) => c.clone_for_shim(), _ => return Err("camera is not perspective".into()) };
    // a moving camera goes over as its two key matrices and their times; the library decomposes and interpolates (AnimatedTransform::interpolate)
    let cam_anim = cam.camera_to_world.is_animated();                               // getter: rs_pbrt.patch (actually_animated)
    let film = cam.film.clone();
    let sb = film.get_sample_bounds(); let cb = film.cropped_pixel_bounds;
    let radius = film.filter.get_radius();
    let mut filter_table = [0.0f32; 256];
    filter_table.copy_from_slice(film.filter_table());                             // getter: rs_pbrt.patch (film.rs:170)
    // (kind, spp, samplepixelcenter, dimensions, xsamples, ysamples, jitter, c_pixel): the pixel samplers' private fields through
    // pub(crate) getters of rs_pbrt.patch; n_sampled_dimensions = samples_1d.len()
    let mut c_pixel = [0u32; 32];
    let (sampler_kind, spp, at_center, pix_dims, sx, sy, jit) = match integ.get_sampler() {
        Sampler::Sobol(s) => (1u32, s.samples_per_pixel, 0u32, 0u32, 0u32, 0u32, 0u32),
        Sampler::Halton(h) => (2u32, h.samples_per_pixel, h.sample_at_pixel_center() as u32, 0, 0, 0, 0),
        Sampler::Random(r) => (3u32, r.samples_per_pixel, 0, 0, 0, 0, 0),
        Sampler::ZeroTwoSequence(z) => (4u32, z.samples_per_pixel, 0, z.n_sampled_dimensions as u32, 0, 0, 0),
        Sampler::Stratified(t) => { let (x, y, j, dims) = t.shim_params(); (5u32, t.samples_per_pixel, 0, dims, x as u32, y as u32, j as u32) }
        Sampler::MaxMinDist(m) => { let (c, dims) = m.shim_params(); c_pixel = c; (6u32, m.samples_per_pixel, 0, dims, 0, 0, 0) }
        _ => return Err("MLT sampler".into()),
    };
    if sampler_kind >= 3 && integrator_kind != 0 { return Err("pixel sampler with an integrator other than path".into()); }
    let mut vdc = vec![0u64; 25 * 52]; let mut vdc_inv = vec![0u64; 26 * 52];       // rows zero-padded to 52 entries
    for (i, row) in VD_C_SOBOL_MATRICES.iter().enumerate() { vdc[i * 52..i * 52 + row.len()].copy_from_slice(row); }
    for (i, row) in VD_C_SOBOL_MATRICES_INV.iter().enumerate() { vdc_inv[i * 52..i * 52 + row.len()].copy_from_slice(row); }
    let rank: u32 = std::env::var("RSPT_RANK").ok().and_then(|v| v.parse().ok()).unwrap_or(0);   // one process per GPU, INTEGRATION.md §5
    let world: u32 = std::env::var("RSPT_WORLD").ok().and_then(|v| v.parse().ok()).unwrap_or(1);
    let rd = RsptRenderDesc {
        full_res: [film.full_resolution.x, film.full_resolution.y],
        crop_px: [cb.p_min.x, cb.p_min.y, cb.p_max.x, cb.p_max.y], sample_bounds: [sb.p_min.x, sb.p_min.y, sb.p_max.x, sb.p_max.y],
        filter_radius: [radius.x, radius.y], filter_table, max_sample_luminance: film.max_sample_luminance(),
        raster_to_camera: m16(&cam.raster_to_camera.m), camera_to_world: m16(&cam.camera_to_world.start_transform().m),
        lens_radius: cam.lens_radius, focal_distance: cam.focal_distance, shutter_open: cam.shutter_open, shutter_close: cam.shutter_close,
        sampler_kind, spp, max_depth, rr_threshold,
        light_strategy: match strategy.as_str() { "uniform" => 0, "power" => 1, _ => 2 },    // lightdistrib.rs:393-418 falls back to spatial
        tile_size: 16, shard_index: rank, shard_count: world, tile_chunk: 1, sample_at_pixel_center: at_center,
        integrator: integrator_kind, ao_n_samples: ao_n, ao_cos_sample: ao_cos, film_reduce: (world > 1) as u32,
        tables: RsptSamplerTables { sobol32: SOBOL_MATRICES_32.as_ptr(), vdc: vdc.as_ptr(), vdc_inv: vdc_inv.as_ptr(),
                                    halton_perms: RADICAL_INVERSE_PERMUTATIONS.as_ptr(), n_halton_perms: RADICAL_INVERSE_PERMUTATIONS.len() as u64 },
        direct_strategy, pixel_dimensions: pix_dims, n_light_samples: if n_light_samples.is_empty() { std::ptr::null() } else { n_light_samples.as_ptr() },
        strat_x: sx, strat_y: sy, strat_jitter: jit, allow_slow_paths: 0 /* a scene the device would render slower than the tile loop comes back as Err: keep the CPU loop */, maxmin_c_pixel: if sampler_kind == 6 { c_pixel.as_ptr() } else { std::ptr::null() },
        sample_begin: 0, sample_count: 0,
        camera_animated: cam_anim as u32, camera_to_world_end: m16(&cam.camera_to_world.end_transform().m),
        camera_time: [cam.camera_to_world.start_time(), cam.camera_to_world.end_time()],
    };
    let sd = RsptSceneDesc {
        nodes: f.nodes.as_ptr(), n_nodes: f.nodes.len() as u64, prims: f.prims.as_ptr(), n_prims: f.prims.len() as u64,
        meshes: f.meshes.as_ptr(), n_meshes: f.meshes.len() as u32, p: f.p.as_ptr(),
        n: if f.any_n { f.n.as_ptr() } else { std::ptr::null() }, s: if f.any_s { f.s.as_ptr() } else { std::ptr::null() },
        uv: if f.any_uv { f.uv.as_ptr() } else { std::ptr::null() }, n_vertices: (f.p.len() / 3) as u64,
        materials: f.materials.as_ptr(), n_materials: f.materials.len() as u32,
        lights: f.lights.as_ptr(), n_lights: f.lights.len() as u32,
        envmaps: if f.envmaps.is_empty() { std::ptr::null() } else { f.envmaps.as_ptr() }, n_envmaps: f.envmaps.len() as u32,
        textures: if f.textures.is_empty() { std::ptr::null() } else { f.textures.as_ptr() }, n_textures: f.textures.len() as u32,
        images: if f.images.is_empty() { std::ptr::null() } else { f.images.as_ptr() }, n_images: f.images.len() as u32,
        objects: f.objects.as_ptr(), n_objects: f.objects.len() as u32, instances: f.instances.as_ptr(), n_instances: f.instances.len() as u32,
        n_top_nodes, n_top_prims, instancing_mode: (std::env::var_os("RSPT_INSTANCING_FIXED").is_some()) as u32,
        n_media: f.media.len() as u32, media: if f.media.is_empty() { std::ptr::null() } else { f.media.as_ptr() },
    };
    unsafe {
        if rspt_abi_version() != RSPT_ABI_VERSION { return Err("librspt.so ABI version mismatch".into()); }
        if rspt_init(std::env::var("RSPT_DEVICE").ok().and_then(|v| v.parse().ok()).unwrap_or(rank as i32)) != 0 { return Err(err()); }
        if world > 1 {                                                             // X1: the id travels through a file the launcher names
            let path = std::env::var("RSPT_COMM_ID_FILE").map_err(|_| "RSPT_COMM_ID_FILE unset")?;
            let mut id = [0u8; 128];
            if rank == 0 {
                if rspt_comm_unique_id(id.as_mut_ptr()) != 0 { return Err(err()); }
                std::fs::write(format!("{}.tmp", path), &id[..]).and_then(|_| std::fs::rename(format!("{}.tmp", path), &path)).map_err(|e| e.to_string())?;
            } else {
                loop { if let Ok(b) = std::fs::read(&path) { if b.len() == 128 { id.copy_from_slice(&b); break; } } std::thread::sleep(std::time::Duration::from_millis(20)); }
            }
            if rspt_comm_init(rank as i32, world as i32, id.as_ptr()) != 0 { return Err(err()); }
        }
        let mut handle: *mut RsptSceneOpaque = std::ptr::null_mut();
        if rspt_scene_create(&sd, &mut handle) != 0 { return Err(err()); }
        let npix = ((cb.p_max.x - cb.p_min.x) * (cb.p_max.y - cb.p_min.y)) as usize;
        let mut xyzw = vec![0.0f32; 4 * npix];
        let mut stats = RsptStats::default();
        let rc = rspt_render(handle, &rd, xyzw.as_mut_ptr(), &mut stats);          // with film_reduce the sum arrives on rank 0
        let msg = if rc != 0 { err() } else { String::new() };
        rspt_scene_destroy(handle);
        if rc != 0 { return Err(msg); }
        if rank != 0 { std::process::exit(0); }                                    // ranks > 0 have delivered their tiles
        // Film.pixels exactly as merge_film_tile leaves them (film.rs:346-371): xyz is ALREADY XYZ (the library applied rgb_to_xyz
        // once per pixel), so write_image must not convert again on the way in — it only reads xyz / filter_weight_sum (film.rs:445-462)
        let mut px = film.pixels.write().unwrap();
        for i in 0..npix { px[i].set_xyz_weight([xyzw[4 * i], xyzw[4 * i + 1], xyzw[4 * i + 2]], xyzw[4 * i + 3]); } // setter: rs_pbrt.patch (film.rs:38-43)
        eprintln!("rspt: {} samples in {:.3} s ({:.1} Msamples/s), {} NaN samples, {} truncated paths", stats.samples, stats.t_render_s,
                  stats.samples as f64 / stats.t_render_s / 1e6, stats.nan_samples, stats.truncated_paths);
    }
    Ok(())
}
