"""ctypes binding of librspt.so (include/rspt.h) — the only way the Python host mirror reaches
the GPU.  There is no fallback: if the shared library is missing, or there is no gfx950 device,
the calls raise RsptError.  Nothing under oracle/ is imported here."""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RSPT_LIB", os.path.join(_HERE, "librspt.so"))  # RSPT_LIB: alternative build (kernel A/B tests)
_LIB = None

EXPORTS = ("rspt_abi_version", "rspt_init", "rspt_shutdown", "rspt_scene_create", "rspt_scene_destroy", "rspt_render",
           "rspt_render_device", "rspt_render_samples", "rspt_trace", "rspt_trace_device", "rspt_dev_alloc", "rspt_dev_free",
           "rspt_dev_upload", "rspt_dev_download", "rspt_last_error", "rspt_last_counters", "rspt_bvh_build", "rspt_bvh_last_error",
           "rspt_bvh_build_gpu", "rspt_bvh_build_bounds", "rspt_comm_unique_id", "rspt_comm_init", "rspt_comm_destroy", "rspt_light_distribution", "rspt_libm",
           "rspt_material_lobes", "rspt_camera_decompose", "rspt_motion_bounds", "rspt_source_hash", "rspt_comm_library")


def tree_source_hash():
    """sha256 (first 16 hex digits) over the kernel / ABI sources under rs_pbrt_amd/csrc + include/rspt.h as they are ON DISK — the recipe
    of csrc/Makefile's SRC_HASH, which compiles the same digest into librspt.so.  Equal to source_hash() exactly when the library was
    built from this tree (tests/test_abi.py checks that)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    names = sorted(os.path.basename(f).encode() for pat in ("*.h", "*.hip", "*.cpp") for f in glob.glob(os.path.join(_HERE, "csrc", pat)))
    for n in names:
        h.update(n)
        h.update(open(os.path.join(_HERE, "csrc", n.decode()), "rb").read())
    h.update(b"rspt.h")
    h.update(open(os.path.join(_HERE, "..", "include", "rspt.h"), "rb").read())
    return h.hexdigest()[:16]


def source_hash():
    """The source hash compiled into the librspt.so that is LOADED (rspt_source_hash): profiles taken on the GPU box
    (tools/refresh_profiles.sh) carry it, and bench.py only quotes a PMC traffic figure whose hash equals the running library's."""
    return lib().rspt_source_hash().decode()


class RsptError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("librspt error %d: %s" % (code, msg))
        self.code = code


def lib():
    """Load librspt.so (built by __graft_entry__.build() / make -C rs_pbrt_amd/csrc)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RsptError(abi.E_NODEVICE, "%s is missing: build it with `make -C rs_pbrt_amd/csrc` (hipcc, gfx950)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        vp, u64, i32, u32 = C.c_void_p, C.c_uint64, C.c_int32, C.c_uint32
        L.rspt_abi_version.restype = C.c_int
        L.rspt_last_error.restype = C.c_char_p
        L.rspt_bvh_last_error.restype = C.c_char_p
        L.rspt_init.argtypes = [i32]
        L.rspt_shutdown.restype = None
        L.rspt_scene_create.argtypes = [vp, vp]
        L.rspt_scene_destroy.argtypes = [vp]
        L.rspt_render.argtypes = [vp, vp, vp, vp]
        L.rspt_render_device.argtypes = [vp, vp, vp, vp]
        L.rspt_render_samples.argtypes = [vp, vp, vp, vp]
        L.rspt_trace.argtypes = [vp, vp, u64, vp, C.c_int]
        L.rspt_trace_device.argtypes = [vp, vp, u64, vp, C.c_int, C.c_int, vp]
        L.rspt_dev_alloc.argtypes = [u64, vp]
        L.rspt_dev_free.argtypes = [vp]
        L.rspt_dev_upload.argtypes = [vp, vp, u64]
        L.rspt_dev_download.argtypes = [vp, vp, u64]
        L.rspt_last_counters.argtypes = [vp]
        L.rspt_bvh_build.restype = C.c_int64
        L.rspt_bvh_build.argtypes = [vp, vp, u64, u32, vp, u64, vp, i32]
        L.rspt_bvh_build_bounds.restype = C.c_int64
        L.rspt_bvh_build_bounds.argtypes = [vp, u64, u32, vp, u64, vp, i32]
        L.rspt_bvh_build_gpu.restype = C.c_int64
        L.rspt_bvh_build_gpu.argtypes = [vp, u64, vp, u64, u32, vp, u64, vp]
        L.rspt_light_distribution.argtypes = [vp, u32, vp, vp, vp, vp, vp]
        L.rspt_libm.argtypes = [u32, vp, vp, C.c_uint64, vp]
        L.rspt_material_lobes.argtypes = [vp, u32, u32, vp, vp]
        L.rspt_camera_decompose.argtypes = [vp, C.c_float, vp, C.c_float, vp, vp]
        L.rspt_motion_bounds.argtypes = [vp, C.c_float, vp, C.c_float, vp, vp, vp, vp, vp]
        L.rspt_source_hash.restype = C.c_char_p
        L.rspt_comm_library.restype = C.c_char_p
        L.rspt_comm_unique_id.argtypes = [vp]
        L.rspt_comm_init.argtypes = [i32, i32, vp]
        _LIB = L
    return _LIB


def _check(rc):
    if rc != 0:
        raise RsptError(rc, (lib().rspt_last_error() or b"").decode("utf-8", "replace"))


MATERIAL_DYNAMIC = 1000


def camera_decompose(rd):
    """What rspt_render makes of a moving camera's two key matrices (rspt_camera_decompose; host only): None when they are equal, else
    (t (2, 3), r (2, 4) xyzw, s (2, 4, 4)) — AnimatedTransform::new's translations, quaternions (second on the shorter arc), scale matrices."""
    a = np.asarray(list(rd.camera_to_world), np.float32); b = np.asarray(list(rd.camera_to_world_end), np.float32)
    animated = C.c_int32(0); trs = np.zeros(46, np.float32)
    _check(lib().rspt_camera_decompose(a.ctypes.data, float(rd.camera_time[0]), b.ctypes.data, float(rd.camera_time[1]), C.addressof(animated), trs.ctypes.data))
    if not animated.value:
        return None
    return trs[:6].reshape(2, 3).copy(), trs[6:14].reshape(2, 4).copy(), trs[14:].reshape(2, 4, 4).copy()


def motion_bounds(start_m, start_time, end_m, end_time, box_min, box_max):
    """AnimatedTransform::motion_bounds (transform.rs:2147-2210) of a box under a moving instance's two keys (rspt_motion_bounds; host only):
    (lo (3,), hi (3,), actually_animated, has_rotation)"""
    a = np.ascontiguousarray(start_m, np.float32).reshape(16); b = np.ascontiguousarray(end_m, np.float32).reshape(16)
    lo = np.ascontiguousarray(box_min, np.float32).reshape(3); hi = np.ascontiguousarray(box_max, np.float32).reshape(3)
    out_lo, out_hi, flags = np.zeros(3, np.float32), np.zeros(3, np.float32), C.c_int32(0)
    _check(lib().rspt_motion_bounds(a.ctypes.data, float(start_time), b.ctypes.data, float(end_time), lo.ctypes.data, hi.ctypes.data,
                                    out_lo.ctypes.data, out_hi.ctypes.data, C.addressof(flags)))
    return out_lo, out_hi, bool(flags.value & 1), bool(flags.value & 2)


def material_lobes(scene, material, allow_multiple_lobes=True):
    """The lobe list librspt assembles for material `material` of a scenes.Scene (rspt_material_lobes; host only, no device):
    (eta, bump_tex, lobes BXDF_DT[]) — tex_* fields are 1 + texture index."""
    mat = abi.Material()
    bx = np.zeros(8, abi.BXDF_DT)
    n = lib().rspt_material_lobes(C.addressof(scene.desc), int(material), int(bool(allow_multiple_lobes)), C.addressof(mat), bx.ctypes.data)
    if n < 0:
        _check(n)
    if n == MATERIAL_DYNAMIC:   # the lobe list depends on texture values at the hit: built per hit on the device
        return float(mat.eta), int(mat.bump_tex), None
    return float(mat.eta), int(mat.bump_tex), bx[:n].copy()


def bvh_build(P, tri, max_prims_in_node=4, threads=0):
    """BVHAccel::new (bvh.rs:96-392) on the host; returns (nodes NODE_DT[], ordered u32[])."""
    P = np.ascontiguousarray(P, np.float32)
    tri = np.ascontiguousarray(tri, np.uint32)
    n = len(tri)
    nodes = np.zeros(max(2 * n, 1), abi.NODE_DT)
    ordered = np.zeros(n, np.uint32)
    k = lib().rspt_bvh_build(P.ctypes.data, tri.ctypes.data, n, max_prims_in_node, nodes.ctypes.data, len(nodes), ordered.ctypes.data, threads)
    if k < 0:
        raise RsptError(int(k), (lib().rspt_bvh_last_error() or b"").decode())
    return nodes[:k].copy(), ordered


def bvh_build_bounds(bounds, max_prims_in_node=4, threads=0):
    """BVHAccel::new over primitives given by their world bounds (n, 6): (nodes, ordered)"""
    bounds = np.ascontiguousarray(bounds, np.float32).reshape(-1, 6)
    n = len(bounds)
    nodes = np.zeros(max(2 * n, 1), abi.NODE_DT)
    ordered = np.zeros(n, np.uint32)
    k = lib().rspt_bvh_build_bounds(bounds.ctypes.data, n, max_prims_in_node, nodes.ctypes.data, len(nodes), ordered.ctypes.data, threads)
    if k < 0:
        raise RsptError(int(k), (lib().rspt_bvh_last_error() or b"").decode())
    return nodes[:k].copy(), ordered


def bvh_build_gpu(P, tri, max_prims_in_node=4):
    """BVHAccel::new on the GPU (level-synchronous; identical output to bvh_build).  Needs init()."""
    P = np.ascontiguousarray(P, np.float32)
    tri = np.ascontiguousarray(tri, np.uint32)
    n = len(tri)
    nodes = np.zeros(max(2 * n, 1), abi.NODE_DT)
    ordered = np.zeros(n, np.uint32)
    k = lib().rspt_bvh_build_gpu(P.ctypes.data, len(P.reshape(-1, 3)), tri.ctypes.data, n, max_prims_in_node, nodes.ctypes.data, len(nodes), ordered.ctypes.data)
    if k < 0:
        raise RsptError(int(k), (lib().rspt_last_error() or b"").decode())
    return nodes[:k].copy(), ordered


_inited_device = None


def init(device=0):
    global _inited_device
    _check(lib().rspt_init(device))
    _inited_device = device


def shutdown():
    global _inited_device
    lib().rspt_shutdown()
    _inited_device = None


LIBM = {"sin": 0, "cos": 1, "log": 2, "log2": 3, "exp": 4, "acos": 5, "atan2": 6}


def libm(fn, x, y=None):
    """rspt_libm: the device's restatement of the host libm's sinf / cosf / logf / log2f / expf / acosf / atan2f(x, y), element by element"""
    x = np.ascontiguousarray(x, np.float32)
    y = None if y is None else np.ascontiguousarray(y, np.float32)
    out = np.empty_like(x)
    _check(lib().rspt_libm(LIBM[fn], x.ctypes.data, None if y is None else y.ctypes.data, x.size, out.ctypes.data))
    return out


def mat4_inverse(m):
    """rspt_libm(RSPT_LIBM_MAT4_INVERSE): Matrix4x4::inverse (transform.rs:128-200) of n row-major 4x4 matrices as the device evaluates it (csrc/mat4_inverse.h)"""
    m = np.ascontiguousarray(m, np.float32).reshape(-1, 16)
    out = np.empty_like(m)
    _check(lib().rspt_libm(7, m.ctypes.data, None, len(m), out.ctypes.data))
    return out


LEAF_GEOM = {"triangle": 8, "box": 9, "offset_ray_origin": 10, "microfacet": 11, "vectors": 12, "area_light": 13}


def leaf_geom(fn, x):
    """rspt_libm codes 8 .. 13 (include/rspt.h): the traversal / shading geometry as the kernels call it — n elements of 16 floats in, 16 floats out"""
    x = np.ascontiguousarray(x, np.float32).reshape(-1, 16)
    out = np.empty_like(x)
    _check(lib().rspt_libm(LEAF_GEOM[fn], x.ctypes.data, None, len(x), out.ctypes.data))
    return out


def leaf_lobe(records, wo, wi, u):
    """rspt_libm code 14: lobe_f / lobe_pdf / lobe_sample_f of BXDF_DT records at (wo, wi, u) -> (n, 13): f(3) pdf | sample value(3) wi(3) pdf sampled_type | get_type"""
    n = len(records)
    x = np.zeros((n, 48), np.float32)
    x[:, :29] = np.ascontiguousarray(records).view(np.float32).reshape(n, 29)
    x[:, 29:32], x[:, 32:35], x[:, 35:37] = wo, wi, u
    out = np.empty_like(x)
    _check(lib().rspt_libm(14, x.ctypes.data, None, n, out.ctypes.data))
    return out[:, :13]


def light_distribution(dscene, strategy, p):
    """rspt_light_distribution: LightDistribution::lookup(p) -> (func[n_lights], cdf[n_lights + 1], n_voxels[3], voxel[3])"""
    n = int(dscene.host.desc.n_lights)
    func, cdf = np.zeros(n, np.float32), np.zeros(n + 1, np.float32)
    nv, vx = np.zeros(3, np.int32), np.zeros(3, np.int32)
    pp = np.ascontiguousarray(p, np.float32)
    _check(lib().rspt_light_distribution(dscene.handle, int(strategy), pp.ctypes.data, func.ctypes.data, cdf.ctypes.data, nv.ctypes.data, vx.ctypes.data))
    return func, cdf, nv, vx


def comm_unique_id():
    """rspt_comm_unique_id: the 128 bytes rank 0 hands to the other ranks"""
    buf = (C.c_uint8 * 128)()
    _check(lib().rspt_comm_unique_id(C.addressof(buf)))
    return bytes(buf)


def comm_init(rank, world, uid):
    """rspt_comm_init (collective over the ranks): the RCCL communicator behind rspt_render_desc.film_reduce"""
    buf = (C.c_uint8 * 128).from_buffer_copy(uid)
    _check(lib().rspt_comm_init(rank, world, C.addressof(buf)))


def comm_destroy():
    lib().rspt_comm_destroy()


def comm_library():
    """the librccl.so rspt_comm_* are bound to (rspt_comm_library): the one the process had mapped already (torch's, where torch was imported
    first), else $RSPT_RCCL_LIB, else the loader's"""
    p = lib().rspt_comm_library()
    if p is None:
        _check(abi.E_UNSUPPORTED)
    return p.decode()


class DeviceScene:
    """Owner of an rspt_scene_t (the flattened Scene + BVHAccel resident in HBM)."""

    def __init__(self, scene):
        self.host = scene  # keeps the numpy arrays alive for the duration of the upload
        h = C.c_void_p()
        _check(lib().rspt_scene_create(C.addressof(scene.desc), C.addressof(h)))
        self.handle = h

    def close(self):
        if self.handle:
            lib().rspt_scene_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def stats_dict(st):
    return {k: getattr(st, k) for k, _ in abi.Stats._fields_}


def render(dscene, rd):
    """rspt_render: returns (film (npix, 4) f32 xyz+weight, stats dict)."""
    npix = (rd.crop_px[2] - rd.crop_px[0]) * (rd.crop_px[3] - rd.crop_px[1])
    film = np.zeros((npix, 4), np.float32)
    st = abi.Stats()
    _check(lib().rspt_render(dscene.handle, C.addressof(rd), film.ctypes.data, C.addressof(st)))
    return film, stats_dict(st)


def render_device(dscene, rd, film_dev_ptr):
    """rspt_render_device: film stays in HBM at film_dev_ptr (npix*4 f32)."""
    st = abi.Stats()
    _check(lib().rspt_render_device(dscene.handle, C.addressof(rd), C.c_void_p(film_dev_ptr), C.addressof(st)))
    return stats_dict(st)


def render_samples(dscene, rd):
    """rspt_render_samples: radiance of every camera sample, (npix, spp, 3)."""
    npix = (rd.crop_px[2] - rd.crop_px[0]) * (rd.crop_px[3] - rd.crop_px[1])
    li = np.zeros((npix, int(rd.spp), 3), np.float32)
    st = abi.Stats()
    _check(lib().rspt_render_samples(dscene.handle, C.addressof(rd), li.ctypes.data, C.addressof(st)))
    return li, stats_dict(st)


def trace(dscene, rays, any_hit=False):
    """rspt_trace: Scene::intersect / intersect_p over a batch of rays (RAY_DT) -> HIT_DT[]."""
    rays = np.ascontiguousarray(rays, abi.RAY_DT)
    out = np.zeros(len(rays), abi.HIT_DT)
    _check(lib().rspt_trace(dscene.handle, rays.ctypes.data, len(rays), out.ctypes.data, int(any_hit)))
    return out


class DeviceBuffer:
    def __init__(self, nbytes):
        p = C.c_void_p()
        _check(lib().rspt_dev_alloc(nbytes, C.addressof(p)))
        self.ptr, self.nbytes = p.value, nbytes

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        _check(lib().rspt_dev_upload(C.c_void_p(self.ptr), arr.ctypes.data, arr.nbytes))

    def download(self, dtype, count):
        out = np.zeros(count, dtype)
        _check(lib().rspt_dev_download(out.ctypes.data, C.c_void_p(self.ptr), out.nbytes))
        return out

    def free(self):
        if self.ptr:
            lib().rspt_dev_free(C.c_void_p(self.ptr))
            self.ptr = None


def trace_device(dscene, rays_buf, n, hits_buf, any_hit=False, repeat=1):
    """rspt_trace_device: returns average kernel milliseconds per launch."""
    ms = C.c_double(0)
    _check(lib().rspt_trace_device(dscene.handle, C.c_void_p(rays_buf.ptr), n, C.c_void_p(hits_buf.ptr), int(any_hit), repeat, C.addressof(ms)))
    return ms.value


def last_counters():
    """(nodes fetched, triangles tested, rays re-traced after LDS stack overflow) of the last trace_device"""
    c = (C.c_uint64 * 3)()
    _check(lib().rspt_last_counters(C.addressof(c)))
    return int(c[0]), int(c[1]), int(c[2])
