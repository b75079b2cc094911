"""Host-side scene set-up: what rs_pbrt's API layer (src/core/api.rs) hands to
`integrator.render` for the BASELINE configs, flattened into the rspt.h structs.

This is host plumbing (runs once per scene, numpy): triangle lists, material lobe
recipes, camera matrices, film/sampler parameters.  It contains
no rendering arithmetic; the BVH is built by whichever builder the caller passes in
(`rs_pbrt_amd.lib().bvh_build` in the product, the oracle's in parity tests)."""
import ctypes as C
import math
import os

import numpy as np

from . import abi

F32 = np.float32
_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "sobol_tables.bin")


# ---------------------------------------------------------------------------------------
# Sobol' tables blob (tools/convert_sobol_tables.py)
# ---------------------------------------------------------------------------------------
class SobolTables:
    def __init__(self, path=_DATA):
        raw = np.fromfile(path, dtype=np.uint8)
        assert raw[:4].tobytes() == b"SBL1", "bad sobol table blob"
        n_dims, msize = np.frombuffer(raw[4:12].tobytes(), dtype="<u4")
        assert (n_dims, msize) == (1024, 52)
        off = 16
        self.sobol32 = np.frombuffer(raw[off:off + 4 * 1024 * 52].tobytes(), dtype="<u4").copy()
        off += 4 * 1024 * 52
        self.vdc = np.frombuffer(raw[off:off + 8 * 25 * 52].tobytes(), dtype="<u8").copy()
        off += 8 * 25 * 52
        self.vdc_inv = np.frombuffer(raw[off:off + 8 * 26 * 52].tobytes(), dtype="<u8").copy()

    def as_struct(self, halton_perms=None):
        t = abi.SamplerTables(self.sobol32.ctypes.data, self.vdc.ctypes.data, self.vdc_inv.ctypes.data, None, 0)
        if halton_perms is not None:
            t.halton_perms, t.n_halton_perms = halton_perms.ctypes.data, len(halton_perms)
        return t


_TABLES = None
_HALTON = {}


def first_primes(n):
    out, v = [], 2
    while len(out) < n:
        if all(v % d for d in range(2, int(v ** 0.5) + 1)):
            out.append(v)
        v += 1
    return out


def halton_permutations(n_dims=256):
    """RADICAL_INVERSE_PERMUTATIONS (src/samplers/halton.rs:19-26) for the first n_dims primes: what
    rs_pbrt's lazy_static builds with compute_radical_inverse_permutations (lowdiscrepancy.rs:2165-2187),
    i.e. identity permutations shuffled (sampling.rs:200-212) with a default-seeded PCG32 (rng.rs:15-83),
    including the reference's bounded-draw threshold (`(!b + 1) & b`, rng.rs:64-73).  Host-side table
    construction, like the Sobol' matrices: the shim hands rs_pbrt's own array through the ABI."""
    if n_dims in _HALTON:
        return _HALTON[n_dims]
    M = (1 << 64) - 1
    state, inc = 0x853C49E6748FEA9B, 0xDA3E39CB94B95BDB
    out = []
    for prime in first_primes(n_dims):
        perm = list(range(prime))
        for i in range(prime):
            b = prime - i
            threshold = ((~b + 1) & 0xFFFFFFFF) & b
            while True:
                old = state
                state = (old * 0x5851F42D4C957F2D + inc) & M
                xs = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
                rot = old >> 59
                r = ((xs >> rot) | (xs << ((-rot) & 31))) & 0xFFFFFFFF
                if r >= threshold:
                    break
            other = i + r % b
            perm[i], perm[other] = perm[other], perm[i]
        out.extend(perm)
    _HALTON[n_dims] = np.array(out, np.uint16)
    return _HALTON[n_dims]


_MAXMIN = None


def maxmin_tables():
    """C_MAX_MIN_DIST (lowdiscrepancy.rs:187): 17 generator matrices of 32 columns (tools/convert_maxmin_table.py)"""
    global _MAXMIN
    if _MAXMIN is None:
        _MAXMIN = np.fromfile(os.path.join(os.path.dirname(_DATA), "maxmin_tables.bin"), dtype="<u4").reshape(17, 32)
    return _MAXMIN


def sobol_tables():
    global _TABLES
    if _TABLES is None:
        _TABLES = SobolTables()
    return _TABLES


# ---------------------------------------------------------------------------------------
# f32 4x4 transforms the way src/core/transform.rs builds them (m and m_inv kept in pairs)
# ---------------------------------------------------------------------------------------
def _mtx_mul(a, b):  # transform.rs:238-250, f32 left-to-right
    r = np.zeros((4, 4), F32)
    for i in range(4):
        for j in range(4):
            r[i, j] = F32(F32(F32(a[i, 0] * b[0, j]) + F32(a[i, 1] * b[1, j])) + F32(a[i, 2] * b[2, j])) + F32(a[i, 3] * b[3, j])
    return r


def _mtx_inverse(m):  # Gauss-Jordan with full pivoting in f32, transform.rs:128-197
    minv = np.array(m, F32).copy()
    indxc, indxr, ipiv = [0] * 4, [0] * 4, [0] * 4
    for i in range(4):
        irow = icol = 0
        big = F32(0)
        for j in range(4):
            if ipiv[j] != 1:
                for k in range(4):
                    if ipiv[k] == 0 and abs(minv[j, k]) >= big:
                        big = abs(minv[j, k]); irow = j; icol = k
        ipiv[icol] += 1
        if irow != icol:
            minv[[irow, icol]] = minv[[icol, irow]]
        indxr[i], indxc[i] = irow, icol
        pivinv = F32(1) / minv[icol, icol]
        minv[icol, icol] = F32(1)
        minv[icol, :] = (minv[icol, :] * pivinv).astype(F32)
        for j in range(4):
            if j != icol:
                save = minv[j, icol]
                minv[j, icol] = F32(0)
                minv[j, :] = (minv[j, :] - (minv[icol, :] * save).astype(F32)).astype(F32)
    for j in (3, 2, 1, 0):
        if indxr[j] != indxc[j]:
            minv[:, [indxr[j], indxc[j]]] = minv[:, [indxc[j], indxr[j]]]
    return minv


def _v3_normalize(v):
    """Vector3f::normalize: v / length, and Vector3f / Float multiplies by the reciprocal (geometry.rs:1271-1279)"""
    x, y, z = (F32(c) for c in v)
    inv = F32(F32(1) / F32(np.sqrt(F32(F32(F32(x * x) + F32(y * y)) + F32(z * z)))))
    return (F32(x * inv), F32(y * inv), F32(z * inv))


def _v3_cross(a, b):
    """vec3_cross_vec3: the products and differences in f64, rounded once (geometry.rs:680-692)"""
    ax, ay, az = (float(F32(c)) for c in a); bx, by, bz = (float(F32(c)) for c in b)
    return (F32(ay * bz - az * by), F32(az * bx - ax * bz), F32(ax * by - ay * bx))


class Transform:
    def __init__(self, m, m_inv=None):
        self.m = np.array(m, F32)
        self.m_inv = np.array(m_inv, F32) if m_inv is not None else _mtx_inverse(self.m)

    def __mul__(self, o):  # transform.rs:869-877
        return Transform(_mtx_mul(self.m, o.m), _mtx_mul(o.m_inv, self.m_inv))

    def inverse(self):
        return Transform(self.m_inv, self.m)

    @staticmethod
    def translate(d):
        m = np.eye(4, dtype=F32); mi = np.eye(4, dtype=F32)
        m[:3, 3] = np.array(d, F32); mi[:3, 3] = -np.array(d, F32)
        return Transform(m, mi)

    @staticmethod
    def scale(x, y, z):
        x, y, z = F32(x), F32(y), F32(z)
        return Transform(np.diag([x, y, z, F32(1)]).astype(F32), np.diag([F32(1) / x, F32(1) / y, F32(1) / z, F32(1)]).astype(F32))

    @staticmethod
    def rotate_y(theta_deg):  # transform.rs:355-367 (m_inv = transpose)
        t = F32(F32(F32(math.pi) / F32(180)) * F32(theta_deg))      # radians() and f32::sin / cos in f32, as the reference (transform.rs:365-367)
        sn, cs = _libm_f32("sinf", t), _libm_f32("cosf", t)
        m = np.array([[cs, 0, sn, 0], [0, 1, 0, 0], [-sn, 0, cs, 0], [0, 0, 0, 1]], F32)
        return Transform(m, m.T.copy())

    @staticmethod
    def identity():
        return Transform(np.eye(4, dtype=F32), np.eye(4, dtype=F32))

    @staticmethod
    def perspective(fov, n, f):  # transform.rs:461-489
        n, f = F32(n), F32(f)
        persp = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, f / (f - n), -f * n / (f - n)], [0, 0, 1, 0]], F32)
        inv_tan = F32(1) / _libm_f32("tanf", F32(F32(F32(F32(math.pi) / F32(180)) * F32(fov)) / F32(2)))      # f32::tan = the platform libm's tanf, not a double tan rounded afterwards
        return Transform.scale(inv_tan, inv_tan, 1) * Transform(persp)

    @staticmethod
    def look_at(pos, look, up):  # transform.rs:414-451 -> world_to_camera (m_inv = camera_to_world); every step in f32 as the reference (tests/test_reference_flow.py holds it to the text)
        def normalize(v):   # Vector3f::normalize: v / length, and Vector3f / Float multiplies by the reciprocal (geometry.rs:1271-1279)
            x, y, z = (F32(c) for c in v)
            inv = F32(F32(1) / F32(np.sqrt(F32(F32(F32(x * x) + F32(y * y)) + F32(z * z)))))
            return (F32(x * inv), F32(y * inv), F32(z * inv))

        def cross(a, b):    # vec3_cross_vec3: the products and differences in f64, rounded once (geometry.rs:460-474)
            ax, ay, az = (float(F32(c)) for c in a); bx, by, bz = (float(F32(c)) for c in b)
            return (F32(ay * bz - az * by), F32(az * bx - ax * bz), F32(ax * by - ay * bx))
        pos32 = tuple(F32(c) for c in pos)
        d = normalize(tuple(F32(F32(l) - p) for l, p in zip(look, pos32)))
        left = normalize(cross(normalize(up), d))
        new_up = cross(d, left)
        c2w = np.eye(4, dtype=F32)
        c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = np.array(left, F32), np.array(new_up, F32), np.array(d, F32), np.array(pos32, F32)
        return Transform(_mtx_inverse(c2w), c2w)


# ---------------------------------------------------------------------------------------
# materials: the parameters of src/materials/*.rs `create`, handed to the library as they are (rspt_material_desc).
# No recipe lives here any more: which lobes a material pushes is decided by librspt (csrc/material_assembly.h) and,
# independently, by the oracle's restatement of compute_scattering_functions (oracle/orc_material.hpp).
# ---------------------------------------------------------------------------------------
class TexRef:
    """A texture of the scene being built (SceneBuilder.*_texture) bound to a material parameter."""

    def __init__(self, index):
        self.index = int(index)


SPECTRUM_PARAMS = ("kd", "ks", "kr", "kt", "reflect", "transmit", "opacity", "eta", "k", "amount")
FLOAT_PARAMS = ("sigma", "roughness", "uroughness", "vroughness", "index", "bumpmap")


def _pbrt_rgb(name, v):
    """a colour parameter of a Material directive: an rgb triple, or a reference to the named texture tools/export_pbrt.py declares
    for texture record `index` ("tex<index>")"""
    if isinstance(v, TexRef):
        return '"texture %s" "tex%d"' % (name, v.index)
    v = np.broadcast_to(np.asarray(v, F32), (3,))
    return '"rgb %s" [%.9g %.9g %.9g]' % (name, v[0], v[1], v[2])


def _pbrt_float(name, v):
    return '"texture %s" "tex%d"' % (name, v.index) if isinstance(v, TexRef) else '"float %s" [%.9g]' % (name, float(v))


def _pbrt_bump(bump):
    return "" if bump is None else ' "texture bumpmap" "tex%d"' % bump.index


def _pbrt_bool(name, v):
    return '"bool %s" ["%s"]' % (name, "true" if v else "false")


def _material(kind, text, remap=True, **params):
    """a material record: kind, the Material directive that makes rs_pbrt build it (tools/export_pbrt.py), and its parameters
    (a number / rgb triple = the ConstantTexture TextureParams would make of it, a TexRef, or None for an absent *_or_null one)"""
    return dict(kind=kind, pbrt=text, remap=bool(remap), params=params)


def material_texrefs(m):
    """(TexRef, "float" | "spectrum") of every parameter of material record m that is bound to a texture of the scene"""
    for name, v in m["params"].items():
        if isinstance(v, TexRef):
            yield v, ("spectrum" if name in SPECTRUM_PARAMS else "float")


def material_scene(m):
    """(geometry-free Scene, material index) for one material record built from literals"""
    sb = SceneBuilder()
    i = sb.add_material(m)
    return sb.materials_only(), i


def matte(kd, sigma=0.0, bump=None):  # matte.rs:26-42
    return _material(abi.MAT_MATTE, 'Material "matte" %s %s%s' % (_pbrt_rgb("Kd", kd), _pbrt_float("sigma", sigma), _pbrt_bump(bump)), kd=kd, sigma=sigma, bumpmap=bump)


def plastic(kd=(0.25,) * 3, ks=(0.25,) * 3, roughness=0.1, remap=True, bump=None):  # plastic.rs:42-56
    return _material(abi.MAT_PLASTIC, 'Material "plastic" %s %s %s %s%s' % (_pbrt_rgb("Kd", kd), _pbrt_rgb("Ks", ks), _pbrt_float("roughness", roughness),
                                                                        _pbrt_bool("remaproughness", remap), _pbrt_bump(bump)), remap, kd=kd, ks=ks, roughness=roughness, bumpmap=bump)


def mirror(kr=(0.9,) * 3, bump=None):  # mirror.rs:24-33
    return _material(abi.MAT_MIRROR, 'Material "mirror" %s%s' % (_pbrt_rgb("Kr", kr), _pbrt_bump(bump)), kr=kr, bumpmap=bump)


def glass(kr=(1.0,) * 3, kt=(1.0,) * 3, index=1.5, uroughness=0.0, vroughness=0.0, remap=True, bump=None, multiple_lobes=None):  # glass.rs:44-82
    """multiple_lobes is ignored (kept for old call sites): the integrator decides (allow_multiple_lobes — true for `path` / `volpath`: one
    FresnelSpecular lobe; false for `directlighting` / `whitted`: SpecularReflection + SpecularTransmission, glass.rs:136-188)"""
    return _material(abi.MAT_GLASS, 'Material "glass" %s %s %s %s %s %s%s' % (_pbrt_rgb("Kr", kr), _pbrt_rgb("Kt", kt), _pbrt_float("uroughness", uroughness),
                                                                          _pbrt_float("vroughness", vroughness), _pbrt_float("index", index), _pbrt_bool("remaproughness", remap), _pbrt_bump(bump)),
                     remap, kr=kr, kt=kt, uroughness=uroughness, vroughness=vroughness, index=index, bumpmap=bump)


def rough_glass(kr=(1.0,) * 3, kt=(1.0,) * 3, uroughness=0.1, vroughness=0.1, index=1.5, remap=True):  # glass.rs:83-211, rough branch
    return glass(kr, kt, index, uroughness, vroughness, remap)


def metal(eta=(0.2004, 0.9240, 1.1022), k=(3.9129, 2.4528, 2.1421), roughness=0.01, remap=True, uroughness=None, vroughness=None, bump=None):  # metal.rs:117-143
    return _material(abi.MAT_METAL, 'Material "metal" %s %s %s%s%s %s%s' % (
        _pbrt_rgb("eta", eta), _pbrt_rgb("k", k), _pbrt_float("roughness", roughness), "" if uroughness is None else " " + _pbrt_float("uroughness", uroughness),
        "" if vroughness is None else " " + _pbrt_float("vroughness", vroughness), _pbrt_bool("remaproughness", remap), _pbrt_bump(bump)),
        remap, eta=eta, k=k, roughness=roughness, uroughness=uroughness, vroughness=vroughness, bumpmap=bump)


def substrate(kd=(0.5,) * 3, ks=(0.5,) * 3, uroughness=0.1, vroughness=0.1, remap=True, bump=None):  # substrate.rs:41-61
    return _material(abi.MAT_SUBSTRATE, 'Material "substrate" %s %s %s %s %s%s' % (_pbrt_rgb("Kd", kd), _pbrt_rgb("Ks", ks), _pbrt_float("uroughness", uroughness),
                                                                               _pbrt_float("vroughness", vroughness), _pbrt_bool("remaproughness", remap), _pbrt_bump(bump)),
                     remap, kd=kd, ks=ks, uroughness=uroughness, vroughness=vroughness, bumpmap=bump)


def uber(kd=(0.25,) * 3, ks=(0.25,) * 3, kr=(0.0,) * 3, kt=(0.0,) * 3, roughness=0.1, uroughness=None, vroughness=None,
         opacity=(1.0,) * 3, index=1.5, remap=True, bump=None):  # uber.rs:60-113
    text = 'Material "uber" %s %s %s %s %s%s%s %s %s %s%s' % (
        _pbrt_rgb("Kd", kd), _pbrt_rgb("Ks", ks), _pbrt_rgb("Kr", kr), _pbrt_rgb("Kt", kt), _pbrt_float("roughness", roughness),
        "" if uroughness is None else " " + _pbrt_float("uroughness", uroughness), "" if vroughness is None else " " + _pbrt_float("vroughness", vroughness),
        _pbrt_rgb("opacity", opacity), _pbrt_float("index", index), _pbrt_bool("remaproughness", remap), _pbrt_bump(bump))
    return _material(abi.MAT_UBER, text, remap, kd=kd, ks=ks, kr=kr, kt=kt, roughness=roughness, uroughness=uroughness, vroughness=vroughness,
                     opacity=opacity, index=index, bumpmap=bump)


def translucent(kd=(0.25,) * 3, ks=(0.25,) * 3, reflect=(0.5,) * 3, transmit=(0.5,) * 3, roughness=0.1, remap=True, bump=None):  # translucent.rs:40-63
    text = 'Material "translucent" %s %s %s %s %s %s%s' % (_pbrt_rgb("Kd", kd), _pbrt_rgb("Ks", ks), _pbrt_rgb("reflect", reflect), _pbrt_rgb("transmit", transmit),
                                                        _pbrt_float("roughness", roughness), _pbrt_bool("remaproughness", remap), _pbrt_bump(bump))
    return _material(abi.MAT_TRANSLUCENT, text, remap, kd=kd, ks=ks, reflect=reflect, transmit=transmit, roughness=roughness, bumpmap=bump)


def mix(m1, m2, amount=(0.5, 0.5, 0.5)):  # api.rs:678-704 (MakeNamedMaterial x 2 + Material "mix", the texture parameter is called "amount"), mixmat.rs:28-41
    named = ("mix", m1["pbrt"], m2["pbrt"], _pbrt_rgb("amount", amount)) if isinstance(m1.get("pbrt"), str) and isinstance(m2.get("pbrt"), str) else None
    m = _material(abi.MAT_MIX, named, amount=amount)
    m["m1"], m["m2"] = m1, m2
    return m



# ---------------------------------------------------------------------------------------
# environment maps: what InfiniteAreaLight::new builds (src/lights/infinite.rs:62-297) — host side
# ---------------------------------------------------------------------------------------
def _env_triangle(level_img, s_, t_):  # MipMap::triangle (mipmap.rs:323-336), wrap Repeat, vectorised in f32
    h, w, _ = level_img.shape
    s = (s_ * F32(w) - F32(0.5)).astype(F32); t = (t_ * F32(h) - F32(0.5)).astype(F32)
    s0 = np.floor(s).astype(np.int64); t0 = np.floor(t).astype(np.int64)
    ds = (s - s0.astype(F32)).astype(F32)[..., None]; dt = (t - t0.astype(F32)).astype(F32)[..., None]
    tex = lambda a, b: level_img[np.mod(b, h), np.mod(a, w)]  # noqa: E731
    one = F32(1)
    tmp1 = tex(s0 + 1, t0 + 1) * (ds * dt); tmp2 = tex(s0 + 1, t0) * (ds * (one - dt))
    tmp3 = tex(s0, t0 + 1) * ((one - ds) * dt); tmp4 = tex(s0, t0) * ((one - ds) * (one - dt))
    return (((tmp4 + tmp3).astype(F32) + tmp2).astype(F32) + tmp1).astype(F32)


def build_envmap(texels):
    """texels (h, w, 3) f32, power-of-two sides (rs_pbrt resamples other sizes first, mipmap.rs:64-148).
    Returns the MIP pyramid (mipmap.rs:154-185) and the scalar image of the sampling distribution
    (infinite.rs:120-137)."""
    img = np.ascontiguousarray(texels, F32)
    h, w, _ = img.shape
    assert w & (w - 1) == 0 and h & (h - 1) == 0, "environment map sides must be powers of two"
    n_levels = 1 + int(math.log2(max(w, h)))
    levels = [img]
    for _ in range(1, n_levels):
        p = levels[-1]
        ph, pw, _ = p.shape
        sh, sw = max(1, ph // 2), max(1, pw // 2)
        ti, si = np.meshgrid(np.arange(sh), np.arange(sw), indexing="ij")
        tex = lambda a, b: p[np.mod(b, ph), np.mod(a, pw)]  # noqa: E731
        acc = (((tex(2 * si, 2 * ti) + tex(2 * si + 1, 2 * ti)).astype(F32) + tex(2 * si, 2 * ti + 1)).astype(F32) + tex(2 * si + 1, 2 * ti + 1)).astype(F32)
        levels.append((acc * F32(0.25)).astype(F32))
    nu, nv = 2 * w, 2 * h
    fwidth = F32(0.5) / F32(min(nu, nv))
    vv, uu = np.meshgrid(np.arange(nv), np.arange(nu), indexing="ij")
    up = ((uu.astype(F32) + F32(0.5)) / F32(nu)).astype(F32); vp = ((vv.astype(F32) + F32(0.5)) / F32(nv)).astype(F32)
    level = F32(n_levels) - F32(1) + F32(math.log2(max(float(fwidth), 1e-8)))
    if level < 0:
        val = _env_triangle(levels[0], up, vp)
    elif level >= n_levels - 1:
        val = np.broadcast_to(levels[-1][0, 0], up.shape + (3,)).astype(F32)
    else:
        il = int(math.floor(level)); delta = F32(level - F32(il))
        val = (_env_triangle(levels[il], up, vp) * (F32(1) - delta) + _env_triangle(levels[il + 1], up, vp) * delta).astype(F32)
    y = (F32(0.212671) * val[..., 0] + F32(0.715160) * val[..., 1]).astype(F32) + F32(0.072169) * val[..., 2]
    sin_row = np.array([_libm_f32("sinf", F32(F32(F32(math.pi) * F32(F32(v) + F32(0.5))) / F32(nv))) for v in range(nv)], F32)      # f32::sin = libm's sinf (numpy's float32 sin differs in the last bit on some rows)
    sin_theta = sin_row[vv]
    dist = np.ascontiguousarray((y.astype(F32) * sin_theta).astype(F32))
    return dict(width=w, height=h, n_levels=n_levels, texels=np.ascontiguousarray(np.concatenate([l.reshape(-1) for l in levels]), F32),
                dist_nu=nu, dist_nv=nv, dist_func=dist)


# ---------------------------------------------------------------------------------------
# image textures: what ImageTexture::new + MipMap::new build (imagemap.rs:34-96, mipmap.rs:56-196) — host side
# ---------------------------------------------------------------------------------------
def _wrap_index(i, n, wrap):
    """index + validity mask for the resampling / pyramid texel fetches (mipmap.rs:88-96, 206-232)"""
    if wrap == abi.WRAP_REPEAT:
        return np.mod(i, n), np.ones(i.shape, bool)
    if wrap == abi.WRAP_CLAMP:
        return np.clip(i, 0, n - 1), np.ones(i.shape, bool)
    return np.clip(i, 0, n - 1), (i >= 0) & (i < n)  # Black: texels outside contribute nothing while resampling


def _lanczos(x, tau=F32(2.0)):  # texture.rs:426-439 in f32
    x = np.abs(x).astype(F32)
    xp = (x * F32(math.pi)).astype(F32)
    with np.errstate(divide="ignore", invalid="ignore"):
        s = (np.sin((xp * tau).astype(F32)).astype(F32) / (xp * tau).astype(F32)).astype(F32)
        lz = (np.sin(xp).astype(F32) / xp).astype(F32)
    out = (s * lz).astype(F32)
    out = np.where(x > F32(1.0), F32(0.0), out)
    return np.where(x < F32(1e-5), F32(1.0), out).astype(F32)


def _resample_weights(old_res, new_res):  # mipmap.rs:298-322
    i = np.arange(new_res)
    center = (((i.astype(F32) + F32(0.5)) * F32(old_res)).astype(F32) / F32(new_res)).astype(F32)
    fw = F32(2.0)
    first = np.floor(((center - fw).astype(F32) + F32(0.5)).astype(F32)).astype(np.int64)
    w = np.zeros((new_res, 4), F32)
    for j in range(4):
        pos = ((first.astype(F32) + F32(j)).astype(F32) + F32(0.5)).astype(F32)
        w[:, j] = _lanczos(((pos - center).astype(F32) / fw).astype(F32))
    tot = (((w[:, 0] + w[:, 1]).astype(F32) + w[:, 2]).astype(F32) + w[:, 3]).astype(F32)
    inv = (F32(1.0) / tot).astype(F32)
    return first, (w * inv[:, None]).astype(F32)


def _pyramid(img, wrap):  # mipmap.rs:154-185
    h, w, _ = img.shape
    n_levels = 1 + int(math.log2(max(w, h)))
    levels = [img]
    for _ in range(1, n_levels):
        p = levels[-1]
        ph, pw, _ = p.shape
        sh, sw = max(1, ph // 2), max(1, pw // 2)
        ti, si = np.meshgrid(np.arange(sh), np.arange(sw), indexing="ij")
        if wrap == abi.WRAP_REPEAT:
            tex = lambda a, b: p[np.mod(b, ph), np.mod(a, pw)]  # noqa: E731
        else:  # Clamp, and Black's clamp-like texel()
            tex = lambda a, b: p[np.clip(b, 0, ph - 1), np.clip(a, 0, pw - 1)]  # noqa: E731
        acc = (((tex(2 * si, 2 * ti) + tex(2 * si + 1, 2 * ti)).astype(F32) + tex(2 * si, 2 * ti + 1)).astype(F32) + tex(2 * si + 1, 2 * ti + 1)).astype(F32)
        levels.append((acc * F32(0.25)).astype(F32))
    return levels


def inverse_gamma(v):  # inverse_gamma_convert_float (spectrum.rs:1865-1871) in f32
    v = np.asarray(v, F32)
    lo = (v / F32(12.92)).astype(F32)
    hi = np.power(((v + F32(0.055)).astype(F32) / F32(1.055)).astype(F32), F32(2.4)).astype(F32)
    return np.where(v <= F32(0.04045), lo, hi).astype(F32)


def build_image(texels, wrap=abi.WRAP_REPEAT, scale=1.0, gamma=False, channels=3):
    """texels (h, w, 3) f32 in [0, 1] as decoded from the file, row 0 = top (the reference flips rows so that
    t = 0 is the bottom, imagemap.rs:63-71).  Returns the pyramid dict of rspt_image.  channels=1 keeps y()
    (float textures, convert_to_float)."""
    img = np.ascontiguousarray(np.asarray(texels, F32)[::-1])
    img = ((inverse_gamma(img) if gamma else img) * F32(scale)).astype(F32)
    if channels == 1:
        img = ((F32(0.212671) * img[..., 0] + F32(0.715160) * img[..., 1]).astype(F32) + F32(0.072169) * img[..., 2]).astype(F32)[..., None]
    h, w, c = img.shape
    pw, ph = 1 << (w - 1).bit_length(), 1 << (h - 1).bit_length()
    if (pw, ph) != (w, h):  # resample to power-of-two resolution, s then t (mipmap.rs:64-148)
        first, wt = _resample_weights(w, pw)
        tmp = np.zeros((h, pw, c), F32)
        for j in range(4):
            idx, ok = _wrap_index(first + j, w, wrap)
            tmp = (tmp + np.where(ok[None, :, None], img[:, idx, :] * wt[None, :, j, None], F32(0))).astype(F32)
        first, wt = _resample_weights(h, ph)
        out = np.zeros((ph, pw, c), F32)
        for j in range(4):
            idx, ok = _wrap_index(first + j, h, wrap)
            out = (out + np.where(ok[:, None, None], tmp[idx, :, :] * wt[:, None, j, None], F32(0))).astype(F32)
        img = np.maximum(out, F32(0))  # Clampable::clamp(0, inf)
    levels = _pyramid(np.ascontiguousarray(img, F32), wrap)
    return dict(width=img.shape[1], height=img.shape[0], n_levels=len(levels), channels=c,
                texels=np.ascontiguousarray(np.concatenate([lv.reshape(-1) for lv in levels]), F32))


# ---------------------------------------------------------------------------------------
# A scene before BVH build: meshes with per-mesh material / emission
# ---------------------------------------------------------------------------------------
class SceneBuilder:
    def __init__(self):
        self.P, self.N, self.UV, self.tris, self.tri_mesh = [], [], [], [], []
        self.image_src = {}   # image index -> the 8-bit source of an image_texture_u8
        self.meshes, self.mesh_material, self.mesh_emit = [], [], []
        self.media = []
        self.materials = []
        self.nv = 0
        self.any_n = self.any_uv = False
        self.delta_lights = []  # point / spot / distant / infinite: appended to Scene.lights after the area lights
        self.envmaps = []
        self.images, self.textures = [], []
        self._const_tex = {}         # literal material parameter -> its ConstantTexture record
        # object instancing (api.rs:3001-3109): meshes added between begin_object / end_object belong to that object
        self.mesh_object = []        # per mesh: object index or -1 (top level)
        self.objects = {}            # name -> index
        self.cur_object = -1
        self.instances = []          # (object index, Transform primitive_to_world)
        self.decl = []               # render_options.primitives order: ("mesh", m) | ("inst", k)

    # ---- object instancing ----
    def begin_object(self, name):  # pbrt_object_begin api.rs:3001-3013
        assert self.cur_object < 0, "ObjectBegin called inside of instance definition"
        self.objects[name] = len(self.objects)
        self.cur_object = self.objects[name]

    def end_object(self):  # pbrt_object_end
        self.cur_object = -1

    def add_instance(self, name, to_world, to_world_end=None, time=(0.0, 1.0)):
        """pbrt_object_instance (api.rs:3024-3109): TransformedPrimitive(object aggregate, AnimatedTransform(CTM[0], start time, CTM[1], end time)).
        to_world_end: the second key of a MOVING instance (ActiveTransform EndTime ... ObjectInstance); its top-level bounds are then
        AnimatedTransform::motion_bounds — the union of the two keys' bounds when the keys differ by translation / scale only
        (transform.rs:2147-2160); with a rotation between the keys the motion of every corner is bounded at the zeros of its velocity
        (bound_point_motion, :2164-2210) — both by librspt's rspt_motion_bounds (csrc/motion_bounds.h)"""
        assert self.cur_object < 0, "ObjectInstance can't be called inside instance definition"
        self.decl.append(("inst", len(self.instances)))
        self.instances.append((self.objects[name], to_world, to_world_end, (float(time[0]), float(time[1]))))

    # ---- textures (api.rs make_texture; src/textures/*.rs) ----
    def _add_texture(self, **kw):
        t = np.zeros((), abi.TEXTURE_DT)
        for k, v in kw.items():
            t[k] = v
        self.textures.append(t)
        return TexRef(len(self.textures) - 1)

    def constant_texture(self, value):
        v = np.broadcast_to(np.asarray(value, F32), (3,))
        return self._add_texture(kind=abi.TEX_CONSTANT, value=v)

    def scale_texture(self, t1, t2):
        return self._add_texture(kind=abi.TEX_SCALE, tex1=t1.index, tex2=t2.index)

    def image_texture(self, texels, mapping="uv", su=1.0, sv=1.0, du=0.0, dv=0.0, v1=(1, 0, 0), v2=(0, 1, 0), trilinear=False,
                      max_aniso=8.0, wrap="repeat", scale=1.0, gamma=False, channels=3, world_to_texture=None):
        """Texture "imagemap" (api.rs get_texture / imagemap.rs): texels (h, w, 3) in file order (row 0 = top)."""
        wm = {"repeat": abi.WRAP_REPEAT, "black": abi.WRAP_BLACK, "clamp": abi.WRAP_CLAMP}[wrap]
        self.images.append(build_image(texels, wm, scale, gamma, channels))
        return self._add_texture(kind=abi.TEX_IMAGE, image=len(self.images) - 1, trilinear=int(trilinear), max_aniso=max_aniso, wrap=wm,
                                 **self._mapping2d(mapping, su, sv, du, dv, v1, v2, world_to_texture))

    def image_texture_u8(self, rgb8, **kw):
        """An image texture whose texels are what ImageTexture::new reads from an 8-bit file: Float::from(u8) / 255 (imagemap.rs:55-62).
        The bytes are kept (image_src) so that tools/export_pbrt.py can write the file for rs_pbrt."""
        rgb8 = np.ascontiguousarray(rgb8, np.uint8)
        t = self.image_texture((rgb8.astype(F32) / F32(255.0)).astype(F32), **kw)
        self.image_src[len(self.images) - 1] = dict(u8=rgb8, scale=float(kw.get("scale", 1.0)), gamma=bool(kw.get("gamma", False)), channels=int(kw.get("channels", 3)))
        return t

    @staticmethod
    def _mapping2d(mapping, su=1.0, sv=1.0, du=0.0, dv=0.0, v1=(1, 0, 0), v2=(0, 1, 0), world_to_texture=None):
        """TextureParams -> TextureMapping2D (api.rs get_texture_mapping_2d): "uv", "planar", "spherical", "cylindrical";
        world_to_texture: 4x4 row-major (the CTM inverse at the Texture directive), identity if None"""
        m = np.zeros(8, F32)
        w2t = np.eye(4, dtype=F32) if world_to_texture is None else np.asarray(world_to_texture, F32).reshape(4, 4)
        if mapping == "uv":
            mk = abi.MAP_UV; m[:4] = (su, sv, du, dv)
        elif mapping == "planar":
            mk = abi.MAP_PLANAR; m[:3] = v1; m[3:6] = v2; m[6:8] = (du, dv)
        else:
            mk = {"spherical": abi.MAP_SPHERICAL, "cylindrical": abi.MAP_CYLINDRICAL}[mapping]
        return dict(mapping=mk, map=m, world_to_texture=w2t.reshape(-1))

    def mix_texture(self, t1, t2, amount):
        """MixTexture (mix.rs): amount is a float texture (TexRef) or a number"""
        amt = amount if isinstance(amount, TexRef) else self.constant_texture(amount)
        return self._add_texture(kind=abi.TEX_MIX, tex1=t1.index, tex2=t2.index, tex3=amt.index)

    def checkerboard_texture(self, t1, t2, mapping="uv", **kw):
        """Checkerboard2DTexture (checkerboard.rs; "dimension" 2, no anti-aliasing in the reference)"""
        return self._add_texture(kind=abi.TEX_CHECKERBOARD, tex1=t1.index, tex2=t2.index, **self._mapping2d(mapping, **kw))

    def dots_texture(self, outside, inside, mapping="uv", **kw):
        return self._add_texture(kind=abi.TEX_DOTS, tex1=outside.index, tex2=inside.index, **self._mapping2d(mapping, **kw))

    def _tex3d(self, kind, world_to_texture=None, **kw):
        w2t = np.eye(4, dtype=F32) if world_to_texture is None else np.asarray(world_to_texture, F32).reshape(4, 4)
        return self._add_texture(kind=kind, mapping=abi.MAP_IDENTITY3D, world_to_texture=w2t.reshape(-1), **kw)

    def fbm_texture(self, octaves=8, omega=0.5, world_to_texture=None):  # fbm.rs
        return self._tex3d(abi.TEX_FBM, world_to_texture, octaves=octaves, omega=omega)

    def wrinkled_texture(self, octaves=8, omega=0.5, world_to_texture=None):  # wrinkled.rs
        return self._tex3d(abi.TEX_WRINKLED, world_to_texture, octaves=octaves, omega=omega)

    def windy_texture(self, world_to_texture=None):  # windy.rs
        return self._tex3d(abi.TEX_WINDY, world_to_texture)

    def marble_texture(self, octaves=8, omega=0.5, scale=1.0, variation=0.2, world_to_texture=None):  # marble.rs
        return self._tex3d(abi.TEX_MARBLE, world_to_texture, octaves=octaves, omega=omega, scale=scale, variation=variation)

    def add_material(self, m):
        """m: a record of matte() .. mix(); returns its index.  The two sides of a mix are added in front of it (MakeNamedMaterial)."""
        if m["kind"] == abi.MAT_MIX:
            m = dict(m, m1_index=self.add_material(m["m1"]), m2_index=self.add_material(m["m2"]))
        self.materials.append(m)
        return len(self.materials) - 1

    def _param_texture(self, v, spectrum):
        """1 + texture index of a material parameter: a literal becomes the ConstantTexture TextureParams::get_*_texture builds"""
        if v is None:
            return 0
        if isinstance(v, TexRef):
            return v.index + 1
        val = np.broadcast_to(np.asarray(v, F32), (3,)).astype(F32) if spectrum else np.array([F32(v), 0, 0], F32)
        key = (bool(spectrum), val.tobytes())
        if key not in self._const_tex:
            self._const_tex[key] = self._add_texture(kind=abi.TEX_CONSTANT, value=val).index
        return self._const_tex[key] + 1

    def materials_only(self):
        """a Scene without geometry carrying this builder's materials, textures and images: what rspt_material_lobes and the oracle's
        material hooks read"""
        mats = self.material_descs()
        return Scene(nodes=np.zeros(0, abi.NODE_DT), prims=np.zeros(0, abi.PRIM_DT), meshes=np.zeros(0, abi.MESH_DT), P=np.zeros((0, 3), F32), N=None, UV=None,
                     materials=mats, lights=np.zeros(0, abi.LIGHT_DT), textures=np.array(self.textures, abi.TEXTURE_DT) if self.textures else None,
                     images=self.images, builder=self)

    def material_descs(self):
        out = np.zeros(len(self.materials), abi.MATERIAL_DESC_DT)
        for i, m in enumerate(self.materials):
            out[i]["kind"] = m["kind"]; out[i]["remap_roughness"] = int(m["remap"])
            for name, v in m["params"].items():
                out[i][name] = self._param_texture(v, name in SPECTRUM_PARAMS)
            if m["kind"] == abi.MAT_MIX:
                out[i]["m1"], out[i]["m2"] = m["m1_index"], m["m2_index"]
        return out

    def add_medium(self, sigma_a=(0.0011, 0.0024, 0.014), sigma_s=(2.55, 3.21, 3.77), g=0.0, scale=1.0):
        """MakeNamedMedium "homogeneous" (api.rs:953-1037; the defaults are the ones at :959-960): returns the handle add_mesh's
        `medium=(inside, outside)` takes (None = no medium)"""
        md = np.zeros((), abi.MEDIUM_DT)
        md["kind"] = abi.MEDIUM_HOMOGENEOUS; md["g"] = F32(g)
        md["sigma_a"] = np.array(sigma_a, F32) * F32(scale); md["sigma_s"] = np.array(sigma_s, F32) * F32(scale)
        self.media.append(md)
        return len(self.media)   # 1 + index, what rspt_mesh.medium_inside / _outside hold

    def add_grid_medium(self, density, p0=(0.0, 0.0, 0.0), p1=(1.0, 1.0, 1.0), sigma_a=(0.0011, 0.0024, 0.014), sigma_s=(2.55, 3.21, 3.77), g=0.0, scale=1.0,
                        medium_to_world=None):
        """MakeNamedMedium "heterogeneous" (api.rs:980-1030): density[nz][ny][nx] between p0 and p1 of medium space; GridDensityMedium::new gets
        medium_to_world * (translate(p0) * scale(p1 - p0)) and keeps its inverse.  Returns the handle add_mesh's `medium=` takes."""
        d = np.ascontiguousarray(density, F32)
        assert d.ndim == 3
        data_to_medium = Transform.translate(p0) * Transform.scale(F32(p1[0]) - F32(p0[0]), F32(p1[1]) - F32(p0[1]), F32(p1[2]) - F32(p0[2]))
        m2w = (Transform.identity() if medium_to_world is None else medium_to_world) * data_to_medium
        md = np.zeros((), abi.MEDIUM_DT)
        md["kind"] = abi.MEDIUM_GRID; md["g"] = F32(g)
        md["sigma_a"] = np.array(sigma_a, F32) * F32(scale); md["sigma_s"] = np.array(sigma_s, F32) * F32(scale)
        md["nz"], md["ny"], md["nx"] = d.shape
        md["density"] = d.ctypes.data
        md["world_to_medium"] = m2w.inverse().m.reshape(-1)
        self._grid_keep = getattr(self, "_grid_keep", []) + [d]   # the ABI copies at rspt_scene_create; until then the array must live
        self.media.append(md)
        return len(self.media)

    def add_mesh(self, P, idx, material, N=None, UV=None, emit=None, two_sided=False, flip=False, alpha=None, shadow_alpha=None, medium=(None, None)):
        """P (nv,3) world-space vertices; idx (nt,3); emit = rgb L or None; alpha / shadow_alpha: float textures (TexRef) of the
        shape's "alpha" / "shadowalpha" parameters (api.rs:1920-1965): where they evaluate to 0 the surface is not there.
        material None: Material "" / "none" (a surface without BSDF: a medium boundary); medium = the graphics state's
        MediumInterface (inside, outside) when the shape is declared (api.rs:2618-2640)."""
        if material is None:
            material = abi.NO_MATERIAL
        P = np.asarray(P, F32).reshape(-1, 3); idx = np.asarray(idx, np.uint32).reshape(-1, 3)
        m = len(self.meshes)
        assert self.cur_object < 0 or emit is None, "Area lights not supported with object instancing (api.rs:2899)"
        self.mesh_object.append(self.cur_object)
        if self.cur_object < 0:
            self.decl.append(("mesh", m))
        assert emit is None or (alpha is None and shadow_alpha is None), "alpha masks on emissive meshes are not supported"
        self.meshes.append((int(N is not None), 0, int(UV is not None), int(flip), 0 if alpha is None else alpha.index + 1, 0 if shadow_alpha is None else shadow_alpha.index + 1,
                            medium[0] or 0, medium[1] or 0))
        self.mesh_material.append(material)
        self.mesh_emit.append(None if emit is None else (np.array(emit, F32), bool(two_sided)))
        self.P.append(P)
        self.N.append(np.asarray(N, F32).reshape(-1, 3) if N is not None else np.zeros_like(P))
        self.UV.append(np.asarray(UV, F32).reshape(-1, 2) if UV is not None else np.zeros((len(P), 2), F32))
        self.any_n |= N is not None; self.any_uv |= UV is not None
        self.tris.append(idx + np.uint32(self.nv))
        self.tri_mesh.append(np.full(len(idx), m, np.uint32))
        self.nv += len(P)
        return m

    def add_quad(self, p, material, **kw):
        return self.add_mesh(np.array(p, F32), [[0, 1, 2], [0, 2, 3]], material, **kw)

    def add_box(self, lo, hi, material, **kw):
        """an axis-aligned box as one mesh of 12 triangles whose geometric normals (cross(p0 - p2, p1 - p2), triangle.rs:293) point
        outwards: what a medium boundary needs (Interaction::get_medium picks `outside` for directions on the normal's side)"""
        x0, y0, z0 = lo; x1, y1, z1 = hi
        P = [(x0, y0, z0), (x1, y0, z0), (x1, y1, z0), (x0, y1, z0), (x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1)]
        quads = [(0, 3, 2, 1), (4, 5, 6, 7), (0, 1, 5, 4), (3, 7, 6, 2), (0, 4, 7, 3), (1, 2, 6, 5)]   # -z +z -y +y -x +x, counter-clockwise seen from outside
        idx = [t for a, b, c, d in quads for t in ((a, b, c), (a, c, d))]
        return self.add_mesh(np.array(P, F32), idx, material, **kw)

    def add_point_light(self, p, I):
        """LightSource "point" (api.rs:773-794, point.rs:30-49): p = from, I = I * scale"""
        lt = np.zeros((), abi.LIGHT_DT)
        lt["kind"] = abi.LIGHT_POINT; lt["L"] = np.array(I, F32); lt["p"][:3] = np.array(p, F32)
        self.delta_lights.append(lt)

    def add_spot_light(self, p_from, p_to, I, coneangle=30.0, conedelta=5.0):
        """LightSource "spot" (api.rs:795-848, spot.rs:30-66): world_to_light rotates `to - from` onto +z"""
        # every step in f32, as api.rs:822-829 / SpotLight::new do it (tests/test_reference_flow.py holds this to the reference's text)
        d = _v3_normalize(tuple(F32(F32(t) - F32(f)) for t, f in zip(p_to, p_from)))
        if abs(d[0]) > abs(d[1]):  # vec3_coordinate_system geometry.rs:779-794: v2 = (..) / sqrt(..) multiplies by the reciprocal, v3 = cross(v1, v2)
            inv = F32(F32(1) / F32(np.sqrt(F32(F32(d[0] * d[0]) + F32(d[2] * d[2])))))
            du = (F32(F32(-d[2]) * inv), F32(F32(0) * inv), F32(d[0] * inv))
        else:
            inv = F32(F32(1) / F32(np.sqrt(F32(F32(d[1] * d[1]) + F32(d[2] * d[2])))))
            du = (F32(F32(0) * inv), F32(d[2] * inv), F32(F32(-d[1]) * inv))
        dv = _v3_cross(d, du)
        lt = np.zeros((), abi.LIGHT_DT)
        lt["kind"] = abi.LIGHT_SPOT; lt["L"] = np.array(I, F32); lt["p"][:3] = np.array(p_from, F32)
        lt["p"][3:12] = np.array([du, dv, d], F32).reshape(-1)
        rad = lambda deg: F32(F32(F32(math.pi) / F32(180)) * F32(deg))      # noqa: E731  radians() pbrt.rs:143-146
        lt["p"][12] = _libm_f32("cosf", rad(coneangle))
        lt["p"][13] = _libm_f32("cosf", rad(F32(F32(coneangle) - F32(conedelta))))
        self.delta_lights.append(lt)

    def add_distant_light(self, p_from, p_to, L):
        """LightSource "distant" (api.rs:889-918, distant.rs:25-40): w_light = normalize(from - to)"""
        w = _v3_normalize(tuple(F32(F32(f) - F32(t)) for f, t in zip(p_from, p_to)))      # in f32 (api.rs:903-906, distant.rs:31-33)
        lt = np.zeros((), abi.LIGHT_DT)
        lt["kind"] = abi.LIGHT_DISTANT; lt["L"] = np.array(L, F32); lt["p"][:3] = np.array(w, F32)
        self.delta_lights.append(lt)

    def add_infinite_light(self, L=(1.0, 1.0, 1.0), image=None, light_to_world=None):
        """LightSource "infinite" (api.rs:855-888, infinite.rs:62-297): constant radiance L (a 1x1 map) or a
        lat-long `image` (h, w, 3) with power-of-two sides, multiplied by L.  light_to_world: 3x3 rotation."""
        env = build_envmap(np.array(L, F32)[None, None, :] * (np.ones((1, 1, 3), F32) if image is None else np.asarray(image, F32)))
        self.envmaps.append(env)
        l2w = np.eye(3, dtype=F32) if light_to_world is None else np.asarray(light_to_world, F32).reshape(3, 3)
        lt = np.zeros((), abi.LIGHT_DT)
        lt["kind"] = abi.LIGHT_INFINITE; lt["prim"] = len(self.envmaps) - 1; lt["L"] = np.array(L, F32)
        lt["p"][:9] = l2w.reshape(-1); lt["p"][9:18] = np.linalg.inv(l2w.astype(np.float64)).astype(F32).reshape(-1)
        self.delta_lights.append(lt)

    def finish(self, bvh_builder, max_prims_in_node=4, instancing="reference"):
        """bvh_builder(P (nv,3) f32, tri (nt,3) u32, max_prims) -> (nodes NODE_DT[], ordered u32[]).
        With object instances the top-level aggregate is built over (top-level triangles in declaration order, then the
        TransformedPrimitives in declaration order) from their world bounds by rspt_bvh_build_bounds; every object with more
        than one triangle gets its own BVHAccel from bvh_builder (api.rs:3046-3094)."""
        P = np.ascontiguousarray(np.concatenate(self.P), F32)
        tri = np.ascontiguousarray(np.concatenate(self.tris), np.uint32)
        tri_mesh = np.concatenate(self.tri_mesh)
        tri_obj = np.array(self.mesh_object, np.int64)[tri_mesh]
        top = np.nonzero(tri_obj < 0)[0]          # input triangle numbers of the top-level shapes
        objects = np.zeros(len(self.objects), abi.OBJECT_DT)
        instances = np.zeros(len(self.instances), abi.INSTANCE_DT)
        if not self.instances:
            assert len(top) == len(tri), "objects without instances are dropped by the reference; not modelled"
            nodes, ordered = bvh_builder(P, tri, max_prims_in_node)
            top_in = ordered                       # BVH slot -> input triangle
            n_top_nodes, n_top_prims = len(nodes), len(tri)
            all_nodes, obj_in = [nodes], []
        else:
            from . import lib
            # each object's own aggregate first (its bounds feed the instances' bounds)
            obj_nodes, obj_in, obj_bound = [], [], []
            for o in range(len(self.objects)):
                t_in = np.nonzero(tri_obj == o)[0]
                assert len(t_in) > 0, "ObjectInstance of an empty object is skipped by the reference (api.rs:3037); not modelled"
                if len(t_in) > 1:
                    on, oo = bvh_builder(P, tri[t_in], max_prims_in_node)
                    lo, hi = on["bmin"][0].copy(), on["bmax"][0].copy()   # BVHAccel::world_bound = nodes[0].bounds (bvh.rs:394-400)
                else:
                    on, oo = np.zeros(0, abi.NODE_DT), np.zeros(1, np.uint32)
                    v = P[tri[t_in[0]]]
                    lo, hi = v.min(0), v.max(0)                             # Triangle::world_bound (triangle.rs:126-133)
                obj_nodes.append(on); obj_in.append(t_in[oo]); obj_bound.append((lo.astype(F32), hi.astype(F32)))
            # the aggregate's input list = render_options.primitives: shapes and instances in declaration order
            first_tri_of_mesh = np.concatenate([[0], np.cumsum([len(t) for t in self.tris])])
            in_tri, in_inst = [], []      # per input primitive: triangle number or -1, instance or -1
            for kind, ref in self.decl:
                if kind == "mesh":
                    nt = len(self.tris[ref])
                    in_tri.append(np.arange(first_tri_of_mesh[ref], first_tri_of_mesh[ref] + nt)); in_inst.append(np.full(nt, -1))
                else:
                    in_tri.append(np.array([-1])); in_inst.append(np.array([ref]))
            in_tri, in_inst = np.concatenate(in_tri).astype(np.int64), np.concatenate(in_inst).astype(np.int64)
            bounds = np.zeros((len(in_tri), 6), F32)
            tsel = in_tri >= 0
            tv = P[tri[in_tri[tsel]]]
            bounds[tsel, :3], bounds[tsel, 3:] = tv.min(1), tv.max(1)
            for k, (o, xf, xf_end, times) in enumerate(self.instances):
                lo, hi = _transform_bounds(xf.m, *obj_bound[o])             # TransformedPrimitive::world_bound (primitive.rs:212-215)
                instances[k]["object"] = o
                instances[k]["to_world"] = xf.m.reshape(-1); instances[k]["from_world"] = xf.m_inv.reshape(-1)
                instances[k]["to_world_end"] = xf.m.reshape(-1); instances[k]["from_world_end"] = xf.m_inv.reshape(-1); instances[k]["time"] = times
                if xf_end is not None and not np.array_equal(xf_end.m, xf.m):   # actually_animated (transform.rs:923)
                    instances[k]["animated"] = 1
                    instances[k]["to_world_end"] = xf_end.m.reshape(-1); instances[k]["from_world_end"] = xf_end.m_inv.reshape(-1)
                    # AnimatedTransform::motion_bounds (transform.rs:2147-2210): the keys' boxes joined, and with a rotation between the keys
                    # the corners' paths bounded at the zeros of their velocity — librspt's host function, as the C caller would use it
                    lo, hi, _, _ = lib.motion_bounds(xf.m, times[0], xf_end.m, times[1], *obj_bound[o])
                row = np.nonzero(in_inst == k)[0][0]
                bounds[row, :3], bounds[row, 3:] = lo, hi
            nodes, ordered = lib.bvh_build_bounds(bounds, max_prims_in_node)
            n_top_nodes, n_top_prims = len(nodes), len(bounds)
            top_in = ordered                       # BVH slot -> input primitive
            all_nodes = [nodes]
            node_base, prim_base = n_top_nodes, n_top_prims
            for o, on in enumerate(obj_nodes):
                on = on.copy()
                leaf = on["n_prims"] > 0
                on["offset"][leaf] += prim_base
                on["offset"][~leaf] += node_base
                objects[o] = (node_base, len(on), prim_base, len(obj_in[o]))
                all_nodes.append(on)
                node_base += len(on); prim_base += len(obj_in[o])
        # prims: the top-level aggregate's (triangles and instances, BVH order), then each object's
        n_prims_total = n_top_prims + sum(len(x) for x in obj_in)
        prims = np.zeros(n_prims_total, abi.PRIM_DT)
        mesh_mat = np.array(self.mesh_material, np.uint32)
        if self.instances:
            is_inst = in_inst[top_in] >= 0
            slot_tri = np.where(is_inst, 0, in_tri[top_in])
        else:
            is_inst = np.zeros(n_top_prims, bool)
            slot_tri = top_in
        prims["v"][:n_top_prims] = tri[slot_tri] if len(tri) else 0
        prims["mesh"][:n_top_prims] = tri_mesh[slot_tri]
        prims["material"][:n_top_prims] = mesh_mat[tri_mesh[slot_tri]]
        prims["area_light"][:n_top_prims] = -1
        if self.instances:
            k = np.nonzero(is_inst)[0]
            prims["v"][k] = 0
            prims["v"][k, 0] = in_inst[top_in[k]].astype(np.uint32)
            prims["mesh"][k] = abi.MESH_INSTANCE; prims["material"][k] = abi.NO_MATERIAL
        base = n_top_prims
        for t_in in obj_in:
            sl = slice(base, base + len(t_in))
            prims["v"][sl] = tri[t_in]; prims["mesh"][sl] = tri_mesh[t_in]; prims["material"][sl] = mesh_mat[tri_mesh[t_in]]; prims["area_light"][sl] = -1
            base += len(t_in)
        # lights in shape-declaration order (one DiffuseAreaLight per emissive triangle, api.rs:2810-2852)
        slot_of_tri = np.full(len(tri), -1, np.int64)
        tri_slots = np.nonzero(~is_inst)[0]
        slot_of_tri[slot_tri[tri_slots]] = tri_slots
        lights_in = []
        emissive_mesh = np.array([e is not None for e in self.mesh_emit], bool)
        for t in np.nonzero(emissive_mesh[tri_mesh])[0]:
            prims["area_light"][slot_of_tri[t]] = len(lights_in)
            lights_in.append((int(t), self.mesh_emit[tri_mesh[t]]))
        lights = np.zeros(len(lights_in) + len(self.delta_lights), abi.LIGHT_DT)
        for i, (t, (L, two)) in enumerate(lights_in):
            lights[i]["kind"] = abi.LIGHT_DIFFUSE_AREA; lights[i]["prim"] = slot_of_tri[t]; lights[i]["L"] = L; lights[i]["two_sided"] = int(two)
        for i, lt in enumerate(self.delta_lights):
            lights[len(lights_in) + i] = lt
        mats = self.material_descs()   # (appends the ConstantTextures of literal parameters to self.textures)
        meshes = np.array(self.meshes, np.uint32).view(abi.MESH_DT).reshape(-1)
        return Scene(nodes=np.concatenate(all_nodes), prims=prims, meshes=meshes, P=P,
                     N=np.ascontiguousarray(np.concatenate(self.N), F32) if self.any_n else None,
                     UV=np.ascontiguousarray(np.concatenate(self.UV), F32) if self.any_uv else None,
                     materials=mats, lights=lights, envmaps=self.envmaps,
                     textures=np.array(self.textures, abi.TEXTURE_DT) if self.textures else None, images=self.images,
                     objects=objects, instances=instances, n_top=(n_top_nodes, n_top_prims), media=np.array(self.media, abi.MEDIUM_DT) if self.media else None,
                     instancing={"reference": abi.INSTANCING_REFERENCE, "fixed": abi.INSTANCING_FIXED}[instancing], builder=self)


def _transform_bounds(m, lo, hi):
    """Transform::transform_bounds (transform.rs:596-660) in f32: the eight corners through transform_point (with its divide by the homogeneous
    weight where that is not exactly 1, :490-516), union"""
    m = np.asarray(m, F32)
    out_lo, out_hi = None, None
    for cx, cy, cz in ((0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 1, 1), (1, 1, 0), (1, 0, 1), (1, 1, 1)):
        x, y, z = (hi if cx else lo)[0], (hi if cy else lo)[1], (hi if cz else lo)[2]
        p = np.array([F32(F32(F32(m[r, 0] * x) + F32(m[r, 1] * y)) + F32(m[r, 2] * z)) + m[r, 3] for r in range(3)], F32)
        wp = F32(F32(F32(m[3, 0] * x) + F32(m[3, 1] * y)) + F32(m[3, 2] * z)) + m[3, 3]
        if wp != F32(1.0):
            p = (F32(F32(1.0) / wp) * p).astype(F32)
        out_lo = p if out_lo is None else np.minimum(out_lo, p)
        out_hi = p if out_hi is None else np.maximum(out_hi, p)
    return out_lo, out_hi


class Scene:
    """Flattened scene arrays + the ctypes rspt_scene_desc pointing at them."""

    def __init__(self, nodes, prims, meshes, P, N, UV, materials, lights, S=None, envmaps=(), textures=None, images=(),
                 objects=None, instances=None, n_top=None, instancing=abi.INSTANCING_REFERENCE, builder=None, media=None):
        self.media = media if media is not None else np.zeros(0, abi.MEDIUM_DT)
        self.builder = builder  # declaration-order view of the scene (tools/export_pbrt.py)
        self.nodes, self.prims, self.meshes, self.P, self.N, self.UV, self.S = nodes, prims, meshes, P, N, UV, S
        self.objects = objects if objects is not None else np.zeros(0, abi.OBJECT_DT)
        self.instances = instances if instances is not None else np.zeros(0, abi.INSTANCE_DT)
        self.n_top = n_top if n_top is not None else (len(nodes), len(prims))
        self.materials, self.lights = materials, lights
        self.envmaps = list(envmaps)
        self.textures = textures if textures is not None else np.zeros(0, abi.TEXTURE_DT)
        self.images = list(images)
        self._img_structs = (abi.Image * max(len(self.images), 1))()
        for i, im in enumerate(self.images):
            self._img_structs[i] = abi.Image(im["width"], im["height"], im["n_levels"], im["channels"], im["texels"].ctypes.data)
        p = lambda a: a.ctypes.data if a is not None and a.size else None  # noqa: E731
        self._env_structs = (abi.EnvMap * max(len(self.envmaps), 1))()
        for i, e in enumerate(self.envmaps):
            self._env_structs[i] = abi.EnvMap(e["width"], e["height"], e["n_levels"], 0, e["texels"].ctypes.data, e["dist_nu"], e["dist_nv"], e["dist_func"].ctypes.data)
        self.desc = abi.SceneDesc(p(nodes), len(nodes), p(prims), len(prims), p(meshes), len(meshes),
                                  p(P), p(N), p(S), p(UV), len(P),
                                  p(materials), len(materials), p(lights), len(lights),
                                  C.addressof(self._env_structs) if self.envmaps else None, len(self.envmaps),
                                  p(self.textures), len(self.textures),
                                  C.addressof(self._img_structs) if self.images else None, len(self.images),
                                  p(self.objects), len(self.objects), p(self.instances), len(self.instances),
                                  self.n_top[0], self.n_top[1], instancing, len(self.media), p(self.media))

    def set_instancing(self, mode):
        """"reference" | "fixed" (rspt_scene_desc.instancing_mode); takes effect at the next DeviceScene / oracle call"""
        self.desc.instancing_mode = {"reference": abi.INSTANCING_REFERENCE, "fixed": abi.INSTANCING_FIXED}[mode]

    @property
    def n_tris(self):
        return len(self.prims)


# ---------------------------------------------------------------------------------------
# film / camera / sampler / integrator parameters -> rspt_render_desc
# ---------------------------------------------------------------------------------------
def box_filter_table(radius=(0.5, 0.5)):  # film.rs:198-211 with BoxFilter::evaluate == 1
    return np.ones(256, F32)


_LIBM = None


def _libm_f32(fn, x):
    """an f32 function of the platform libm (expf, tanf, ...): what Rust's f32 methods call"""
    global _LIBM
    import ctypes
    if _LIBM is None:
        import ctypes.util
        _LIBM = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
    f = getattr(_LIBM, fn)
    f.restype = ctypes.c_float
    f.argtypes = [ctypes.c_float]
    return F32(f(float(F32(x))))


def _expf(x):
    """f32::exp of the reference = the platform libm's expf (not a double exp rounded afterwards: the table's small entries differ by several ulps)"""
    return _libm_f32("expf", x)


def gaussian_filter_table(radius=(2.0, 2.0), alpha=2.0):  # filters/gaussian.rs:20-45, film.rs:198-211 — in f32 throughout, as the reference (tests/test_reference_flow.py holds it to the text)
    rx, ry, a = F32(radius[0]), F32(radius[1]), F32(alpha)
    ex, ey = _expf(F32(F32(-a * rx) * rx)), _expf(F32(F32(-a * ry) * ry))
    t = np.zeros(256, F32)
    for y in range(16):
        for x in range(16):
            px = F32(F32(F32(x + 0.5) * rx) / F32(16)); py = F32(F32(F32(y + 0.5) * ry) / F32(16))
            gx = max(F32(0), F32(_expf(F32(F32(-a * px) * px)) - ex))
            gy = max(F32(0), F32(_expf(F32(F32(-a * py) * py)) - ey))
            t[y * 16 + x] = F32(gx * gy)
    return t


def make_render_desc(xres, yres, spp, look_at, fov, max_depth=5, rr_threshold=1.0, light_strategy=abi.LIGHTS_SPATIAL,
                     crop=(0.0, 1.0, 0.0, 1.0), filter_radius=(0.5, 0.5), filter_table=None, lens_radius=0.0,
                     focal_distance=1e6, max_sample_luminance=float("inf"), shard=(0, 1, 64), sampler="sobol",
                     sample_at_pixel_center=False, integrator="path", ao_samples=64, ao_cos_sample=True, direct_strategy="all", light_samples=None,
                     dimensions=4, strat=(4, 4), jitter=True, sample_range=None, allow_slow_paths=True, look_at_end=None, camera_times=(0.0, 1.0),
                     shutter=(0.0, 1.0), mirror_x=False):
    """allow_slow_paths: True here (tests and bench want the device path whatever its speed); the Rust shim passes 0, so that a pixel-sampler
    frame of few tiles — which the device renders slower than the host's tile loop — comes back as RSPT_E_UNSUPPORTED (include/rspt.h)"""
    rd = abi.RenderDesc()
    rd.allow_slow_paths = int(bool(allow_slow_paths))
    # Integrator "path" (path.rs), "ao" / "ambientocclusion" (api.rs:411; ao.rs: nsamples 64, cossample true) or
    # "directlighting" (api.rs:322-349: strategy "all" | "one", maxdepth 5; light_samples = Light::get_n_samples per light)
    rd.integrator = {"ao": abi.INTEGRATOR_AO, "directlighting": abi.INTEGRATOR_DIRECT, "volpath": abi.INTEGRATOR_VOLPATH}.get(integrator, abi.INTEGRATOR_PATH)
    rd.direct_strategy = {"all": abi.DIRECT_SAMPLE_ALL, "one": abi.DIRECT_SAMPLE_ONE}[direct_strategy]
    if light_samples is not None:
        rd._light_samples = np.ascontiguousarray(light_samples, np.int32)  # kept alive by the desc object
        rd.n_light_samples = rd._light_samples.ctypes.data
    rd.ao_n_samples, rd.ao_cos_sample = int(ao_samples), int(bool(ao_cos_sample))
    rd.full_res[:] = (xres, yres)
    # Film::new film.rs:187-196
    cx0, cx1 = math.ceil(float(F32(xres) * F32(crop[0]))), math.ceil(float(F32(xres) * F32(crop[1])))
    cy0, cy1 = math.ceil(float(F32(yres) * F32(crop[2]))), math.ceil(float(F32(yres) * F32(crop[3])))
    rd.crop_px[:] = (cx0, cy0, cx1, cy1)
    rx, ry = float(filter_radius[0]), float(filter_radius[1])
    # Film::get_sample_bounds film.rs:266-292
    rd.sample_bounds[:] = (math.floor(cx0 + 0.5 - rx), math.floor(cy0 + 0.5 - ry), math.ceil(cx1 - 0.5 + rx), math.ceil(cy1 - 0.5 + ry))
    rd.filter_radius[:] = (rx, ry)
    tbl = box_filter_table() if filter_table is None else np.asarray(filter_table, F32)
    rd.filter_table[:] = tbl.tolist()
    rd.max_sample_luminance = max_sample_luminance
    # PerspectiveCamera::create / new perspective.rs:45-186
    frame = F32(xres) / F32(yres)
    if frame > 1.0:
        sw = (-frame, frame, F32(-1), F32(1))
    else:
        sw = (F32(-1), F32(1), F32(-1) / frame, F32(1) / frame)
    cam2screen = Transform.perspective(fov, 1e-2, 1000.0)
    s2r = Transform.scale(xres, yres, 1) * Transform.scale(F32(1) / (sw[1] - sw[0]), F32(1) / (sw[2] - sw[3]), 1) * Transform.translate((-sw[0], -sw[3], 0))
    r2c = cam2screen.inverse() * s2r.inverse()
    rd.raster_to_camera[:] = r2c.m.reshape(-1).tolist()
    w2c = Transform.look_at(*look_at)
    rd.camera_to_world[:] = w2c.m_inv.reshape(-1).tolist()
    if mirror_x:   # `Scale -1 1 1` in front of LookAt: the CTM Scale * LookAt is world-to-camera, so camera-to-world = LookAt^-1 * Scale(-1, 1, 1)
        assert look_at_end is None
        rd.camera_to_world[:] = (w2c.m_inv.astype(np.float64) @ np.diag([-1.0, 1.0, 1.0, 1.0])).astype(F32).reshape(-1).tolist()
    rd.lens_radius, rd.focal_distance = lens_radius, focal_distance
    rd.shutter_open, rd.shutter_close = float(shutter[0]), float(shutter[1])
    if look_at_end is not None:   # a moving camera (ActiveTransform / TransformTimes blocks around LookAt): AnimatedTransform's two key matrices
        rd.camera_animated = 1
        rd.camera_to_world_end[:] = Transform.look_at(*look_at_end).m_inv.reshape(-1).tolist()
        rd.camera_time[:] = (float(camera_times[0]), float(camera_times[1]))
    if sampler == "halton":  # the reference's default sampler (api.rs:526), halton.rs:163-172
        rd.sampler_kind = abi.SAMPLER_HALTON
        rd.spp = spp
        rd.sample_at_pixel_center = int(sample_at_pixel_center)
    elif sampler in ("random", "02sequence", "lowdiscrepancy", "stratified", "maxmindist"):   # the pixel samplers (make_sampler, api.rs:1690-1720)
        rd.pixel_dimensions = int(dimensions)
        rd.spp = spp
        if sampler == "random":                       # random.rs:53-58
            rd.sampler_kind = abi.SAMPLER_RANDOM
        elif sampler == "stratified":                 # stratified.rs:89-99: spp = xsamples * ysamples
            rd.sampler_kind = abi.SAMPLER_STRATIFIED
            rd.strat_x, rd.strat_y = (int(v) for v in strat)
            rd.strat_jitter = int(bool(jitter))
            rd.spp = rd.strat_x * rd.strat_y
        elif sampler == "maxmindist":                 # maxmin.rs:32-59: spp rounded up to a power of two, at most 2^16
            rd.sampler_kind = abi.SAMPLER_MAXMINDIST
            s = 1
            while s < spp:
                s *= 2
            assert s <= 65536, "No more than 65536 samples per pixel are supported with MaxMinDistSampler"
            rd.spp = s
            rd._c_pixel = np.ascontiguousarray(maxmin_tables()[s.bit_length() - 1])   # kept alive by the desc object
            rd.maxmin_c_pixel = rd._c_pixel.ctypes.data
        else:                                         # zerotwosequence.rs:117-126 (no rounding of spp here, unlike pbrt-v3)
            rd.sampler_kind = abi.SAMPLER_ZEROTWO
    else:
        rd.sampler_kind = abi.SAMPLER_SOBOL
        s = 1
        while s < spp:
            s *= 2  # sobol.rs:38-45 rounds up to a power of two
        rd.spp = s
    if sample_range is not None:   # checkpoint / resume: (first sample, number of samples) of every pixel; films of disjoint ranges add up
        rd.sample_begin, rd.sample_count = int(sample_range[0]), int(sample_range[1])
    rd.max_depth, rd.rr_threshold, rd.light_strategy, rd.tile_size = max_depth, rr_threshold, light_strategy, 16
    rd.shard_index, rd.shard_count, rd.tile_chunk = shard
    n_halton_dims = 5 + 8 * (max_depth + 3)
    if integrator == "volpath":   # 10 dimensions per counted bounce, 2 per pass through a medium boundary (librspt cuts and reports a path that needs more)
        n_halton_dims = min(999, 5 + 10 * (max_depth + 2) + 64 + 8)
    if integrator == "directlighting":  # sample arrays + a full specular tree on the fall-back stream (rs_pbrt_amd/csrc/direct.h)
        nl = len(light_samples) if light_samples is not None else 16
        n_halton_dims = min(999, 5 + 4 * max_depth * nl + ((1 << max_depth) - 1) * (4 * nl + 4) + 4)
    rd.tables = sobol_tables().as_struct(halton_permutations(n_halton_dims) if sampler == "halton" else None)
    return rd


def n_pixels(rd):
    return (rd.crop_px[2] - rd.crop_px[0]) * (rd.crop_px[3] - rd.crop_px[1])


def film_to_rgb(film_xyzw):
    """Film::write_image's linear value before gamma (film.rs:445-462): xyz->rgb, / weight, clamp >= 0."""
    f = np.asarray(film_xyzw, F32).reshape(-1, 4)
    x, y, z, w = f[:, 0], f[:, 1], f[:, 2], f[:, 3]
    rgb = np.stack([F32(3.240479) * x - F32(1.537150) * y - F32(0.498535) * z,
                    F32(-0.969256) * x + F32(1.875991) * y + F32(0.041556) * z,
                    F32(0.055648) * x - F32(0.204043) * y + F32(1.057311) * z], axis=1).astype(F32)
    nz = w != 0
    inv = np.zeros_like(w); inv[nz] = F32(1) / w[nz]
    rgb[nz] = np.maximum(rgb[nz] * inv[nz, None], 0)
    return rgb


# ---------------------------------------------------------------------------------------
# BASELINE scenes
# ---------------------------------------------------------------------------------------
def cornell_box(bvh_builder, variant="matte", fog=None):
    """C1: the public Cornell Box data (32 triangles, one quad light).  variant 'mixed' swaps the
    blocks to mirror / glass and the floor to plastic for BSDF coverage.  fog = (sigma_a, sigma_s, g): a homogeneous medium
    fills the room (a box without material just inside the walls, MediumInterface inside = the fog) — a workload for "volpath";
    the two blocks stand inside it (MediumInterface fog / fog: no transition, rays keep their medium)."""
    sb = SceneBuilder()
    med = sb.add_medium(sigma_a=fog[0], sigma_s=fog[1], g=fog[2]) if fog else None
    white = sb.add_material(matte((0.725, 0.71, 0.68)))
    red = sb.add_material(matte((0.63, 0.065, 0.05)))
    green = sb.add_material(matte((0.14, 0.45, 0.091)))
    short_m, tall_m, floor_m = white, white, white
    if variant in ("mixed", "mixed_two_lobes"):   # _two_lobes: the glass as directlighting / whitted see it (allow_multiple_lobes = false)
        tall_m = sb.add_material(mirror())
        short_m = sb.add_material(glass(multiple_lobes=variant == "mixed"))
        floor_m = sb.add_material(plastic((0.4, 0.4, 0.4), (0.3, 0.3, 0.3), 0.05))
    elif variant == "rough":
        tall_m = sb.add_material(metal(roughness=0.1))
        short_m = sb.add_material(matte((0.5, 0.5, 0.7), sigma=30.0))
        floor_m = sb.add_material(plastic((0.4, 0.4, 0.4), (0.3, 0.3, 0.3), 0.2))
    back_m, left_m, right_m = white, green, red
    if variant == "layered":   # the recipes the other variants leave out: substrate, uber (partly transparent, with Kr and Kt), translucent, rough glass
        floor_m = sb.add_material(substrate((0.4, 0.35, 0.3), (0.3, 0.3, 0.3), 0.05, 0.2))
        short_m = sb.add_material(uber((0.3, 0.4, 0.6), (0.2, 0.2, 0.2), (0.1, 0.1, 0.1), (0.2, 0.2, 0.2), roughness=0.1, uroughness=0.05, opacity=(0.8, 0.8, 0.8), index=1.33))
        tall_m = sb.add_material(rough_glass((0.9, 0.9, 0.9), (0.9, 0.9, 0.9), 0.05, 0.1, 1.5))
        back_m = sb.add_material(translucent((0.4, 0.4, 0.4), (0.2, 0.2, 0.2), (0.6, 0.6, 0.6), (0.3, 0.3, 0.3), 0.15))
        right_m = sb.add_material(mix(matte((0.63, 0.065, 0.05)), plastic((0.1, 0.2, 0.5), (0.4, 0.4, 0.4), 0.08), (0.3, 0.5, 0.7)))
    if variant == "procedural":
        # every procedural texture class of src/textures/ (no image files: tools/export_pbrt.py can hand the scene to rs_pbrt as text):
        # 2-D mappings on the quads' uv, 3-D noise in world space scaled to the room, float textures behind roughness and bump
        c = sb.constant_texture
        w2t = lambda k: Transform.scale(k, k, k).m  # noqa: E731
        floor_m = sb.add_material(matte(sb.checkerboard_texture(c((0.725, 0.71, 0.68)), c((0.2, 0.2, 0.25)), su=6.0, sv=6.0),
                                        bump=sb.scale_texture(sb.windy_texture(world_to_texture=w2t(0.02)), c(3.0))))
        back_m = sb.add_material(matte(sb.marble_texture(octaves=6, omega=0.5, scale=3.0, variation=0.3, world_to_texture=w2t(0.01))))
        left_m = sb.add_material(matte(sb.mix_texture(c((0.14, 0.45, 0.091)), c((0.7, 0.65, 0.1)), sb.fbm_texture(octaves=5, omega=0.6, world_to_texture=w2t(0.02)))))
        right_m = sb.add_material(matte(sb.dots_texture(c((0.63, 0.065, 0.05)), c((0.725, 0.71, 0.68)), su=7.0, sv=7.0)))
        short_m = sb.add_material(plastic(sb.scale_texture(c((0.8, 0.8, 0.75)), sb.wrinkled_texture(octaves=6, omega=0.5, world_to_texture=w2t(0.03))), (0.3, 0.3, 0.3),
                                          sb.scale_texture(sb.wrinkled_texture(octaves=4, omega=0.5, world_to_texture=w2t(0.05)), c(0.25))))
        tall_m = sb.add_material(substrate(sb.checkerboard_texture(c((0.1, 0.3, 0.6)), c((0.6, 0.5, 0.1)), mapping="planar", v1=(0.02, 0, 0), v2=(0, 0.02, 0)),
                                           (0.25, 0.25, 0.25), sb.dots_texture(c(0.05), c(0.3), su=4.0, sv=4.0), 0.1))
    if variant == "imagemap":
        # image textures from 8-bit sources (so that the same texels reach rs_pbrt through a PNG): EWA and trilinear lookups, a
        # resolution that is not a power of two (MipMap::new's Lanczos resampling), the three wrap modes, uv and planar mappings,
        # float images (y()) behind roughness and bump
        rng = np.random.default_rng(7)
        yy, xx = np.mgrid[0:24, 0:40]
        a8 = np.stack([((xx // 5 + yy // 4) % 2) * 200 + 30, xx * 6, 80 + rng.integers(0, 100, (24, 40))], -1).astype(np.uint8)
        b8 = rng.integers(0, 256, (32, 32, 3)).astype(np.uint8)
        floor_m = sb.add_material(matte(sb.image_texture_u8(a8, su=4.0, sv=3.0, du=0.15, dv=0.4)))
        back_m = sb.add_material(matte(sb.image_texture_u8(a8[::-1, ::-1], trilinear=True, wrap="clamp", scale=0.8), sigma=20.0))
        left_m = sb.add_material(matte(sb.scale_texture(sb.image_texture_u8(b8, mapping="planar", v1=(0, 0.004, 0), v2=(0, 0, 0.004), du=0.1, dv=0.2, wrap="black"),
                                                        sb.constant_texture((0.3, 0.9, 0.4)))))
        short_m = sb.add_material(plastic((0.5, 0.45, 0.4), sb.image_texture_u8(b8, su=2.0, sv=2.0, trilinear=True, scale=0.6),
                                          sb.image_texture_u8(a8, channels=1, scale=0.4, su=2.0, sv=2.0, trilinear=True)))
        tall_m = sb.add_material(substrate(sb.image_texture_u8(a8, su=2.0, sv=3.0), (0.2, 0.2, 0.2), 0.1, 0.2,
                                           bump=sb.image_texture_u8(b8, channels=1, scale=4.0, su=3.0, sv=3.0, trilinear=True)))
    uvq = dict(UV=[[0, 0], [1, 0], [1, 1], [0, 1]]) if variant in ("procedural", "imagemap") else {}
    q = lambda p, m, **kw: sb.add_quad(p, m, **dict(uvq, **kw))  # noqa: E731
    q([(552.8, 0, 0), (0, 0, 0), (0, 0, 559.2), (549.6, 0, 559.2)], floor_m)
    q([(556, 548.8, 0), (556, 548.8, 559.2), (0, 548.8, 559.2), (0, 548.8, 0)], white)
    q([(549.6, 0, 559.2), (0, 0, 559.2), (0, 548.8, 559.2), (556, 548.8, 559.2)], back_m)
    q([(0, 0, 559.2), (0, 0, 0), (0, 548.8, 0), (0, 548.8, 559.2)], left_m)
    q([(552.8, 0, 0), (549.6, 0, 559.2), (556, 548.8, 559.2), (556, 548.8, 0)], right_m)
    # light: emits downwards; vertex order chosen so cross(p0-p2, p1-p2) points to -y
    q([(343, 548.7, 227), (343, 548.7, 332), (213, 548.7, 332), (213, 548.7, 227)], white, emit=(17, 12, 4))
    for quads, m in (([[(130, 165, 65), (82, 165, 225), (240, 165, 272), (290, 165, 114)],
                       [(290, 0, 114), (290, 165, 114), (240, 165, 272), (240, 0, 272)],
                       [(130, 0, 65), (130, 165, 65), (290, 165, 114), (290, 0, 114)],
                       [(82, 0, 225), (82, 165, 225), (130, 165, 65), (130, 0, 65)],
                       [(240, 0, 272), (240, 165, 272), (82, 165, 225), (82, 0, 225)]], short_m),
                     ([[(423, 330, 247), (265, 330, 296), (314, 330, 456), (472, 330, 406)],
                       [(423, 0, 247), (423, 330, 247), (472, 330, 406), (472, 0, 406)],
                       [(472, 0, 406), (472, 330, 406), (314, 330, 456), (314, 0, 456)],
                       [(314, 0, 456), (314, 330, 456), (265, 330, 296), (265, 0, 296)],
                       [(265, 0, 296), (265, 330, 296), (423, 330, 247), (423, 0, 247)]], tall_m)):
        for p in quads:
            q(p, m, medium=(med, med))
    if fog:
        sb.add_box((1.3, 0.7, 0.9), (548.1, 547.9, 558.3), None, medium=(med, None))
    return sb.finish(bvh_builder)


# ---- the Cornell box of the reference's documentation (docs/source/getting_started.rst:150-209), recovered from its two renders ----
# rs_pbrt ships two renders of rs-pbrt-test-scenes/pbrt/cornell_box/cornell_box.pbrt (Sampler "sobol" 8 and 256 spp, Integrator "path",
# 500 x 500) but not the scene file.  What the file must say was recovered from the images (tools/recover_cornell_docs.py is the procedure,
# tests/test_reference_pin.py the evidence: with this scene the oracle's 8-spp render has 94 % of its pixels byte for byte equal to the
# reference's PNG).  Every value below sits on a sharp optimum of that agreement:
#  * camera: `Scale -1 1 1` in front of `LookAt 278 273 -800  278 273 -799  0 1 0`, fov 39.146 (the frame edges of the 256-spp image give
#    39.149 +- 0.003; at 39.1445 and 39.148 the byte agreement is already lower).  The mirror is on the camera, not the world: the same
#    picture from a mirrored world has other BSDF frames (ts = cross(ns, ss)) and 16 % equal pixels;
#  * room: the public Cornell data; blocks: exact squares of side 165 — centre (185, 169) turned by 0.29 rad, centre (368, 351) turned by
#    1.27 rad (found as free corners, then as squares with 0.02 steps: 185.0, 169.0, 164.99..165.01, 16.616 deg = 0.2900 rad, ...);
#  * every quad a fan (k, k+1, k+2), (k, k+2, k+3) of its vertex cycle — a triangle's first edge is its dpdu (default uv, triangle.rs:330-345),
#    the axis every cosine-sampled bounce is built on; found per visible triangle among 2 diagonals x 3 rotations x 2 windings;
#  * light: the public rectangle at y = 547.8 (+- 0.05 is worse), fan from its last vertex, L = 100; walls 0.4, red / green / blocks 0.5
#    (+- 0.5 % of any of them halves the number of equal pixels).
CORNELL_DOCS_LOOK_AT = ((278, 273, -800), (278, 273, -799), (0, 1, 0))
CORNELL_DOCS_FOV = 39.14625166082039
_DOCS_ROOM = [([(552.8, 0, 0), (0, 0, 0), (0, 0, 559.2), (549.6, 0, 559.2)], "white", 2),            # floor
              ([(556, 548.8, 0), (556, 548.8, 559.2), (0, 548.8, 559.2), (0, 548.8, 0)], "white", 1),   # ceiling
              ([(549.6, 0, 559.2), (0, 0, 559.2), (0, 548.8, 559.2), (556, 548.8, 559.2)], "white", 3),  # back
              ([(0, 0, 559.2), (0, 0, 0), (0, 548.8, 0), (0, 548.8, 559.2)], "green", 1),
              ([(552.8, 0, 0), (549.6, 0, 559.2), (556, 548.8, 559.2), (556, 548.8, 0)], "red", 1)]
_DOCS_SHORT = ((185.0, 169.0, 165.0, 0.29), 165.0, (1, 2, 0, 2, 0))   # (centre x, centre z, side, angle of the edge a->d in rad), height, fan starts of top + 4 sides
_DOCS_TALL = ((368.0, 351.0, 165.0, 1.27), 330.0, (1, 0, 2, 0, 2))
_DOCS_LIGHT = [(343, 547.8, 227), (343, 547.8, 332), (213, 547.8, 332), (213, 547.8, 227)]


def _docs_fan(k, rev=False):
    c = [(k + i) % 4 for i in range(4)]
    if rev:
        c = [c[0], c[3], c[2], c[1]]
    return [[c[0], c[1], c[2]], [c[0], c[2], c[3]]]


def _docs_square(cx, cz, side, angle):
    """corners a, b, c, d in the order the public data lists a block's top (a->d along `angle`, a->b a quarter turn on)"""
    e = np.array([math.cos(angle), math.sin(angle)]) * side
    f = np.array([-e[1], e[0]])
    a = np.array([cx, cz]) - (e + f) / 2
    return [tuple(a), tuple(a + f), tuple(a + e + f), tuple(a + e)]


def _docs_block_quads(corners, h, tall):
    a, b, c, d = corners
    side = lambda u, v: [(u[0], 0, u[1]), (u[0], h, u[1]), (v[0], h, v[1]), (v[0], 0, v[1])]  # noqa: E731
    order = ((a, d), (d, c), (c, b), (b, a)) if tall else ((d, c), (a, d), (b, a), (c, b))   # the public data lists the two blocks' sides in different orders
    return [[(p[0], h, p[1]) for p in (a, b, c, d)]] + [side(u, v) for u, v in order]


def cornell_box_docs(bvh_builder, mirror_world=False, L=100.0, white=0.4, colour=0.5):
    """The scene behind docs/source/cornell_box_{8,256}_pixelsamples.png as recovered from those images (see above); render it with
    cornell_docs_render_desc().  mirror_world=True negates x and turns the light's cycle round so that it still faces down: with
    cornell_docs_render_desc(mirror_camera=False) that is the same picture with other BSDF frames — the negative control of
    tests/test_reference_pin.py."""
    sb = SceneBuilder()
    m = dict(white=sb.add_material(matte((white,) * 3)), red=sb.add_material(matte((colour, 0, 0))), green=sb.add_material(matte((0, colour, 0))),
             block=sb.add_material(matte((colour,) * 3)))
    sx = -1.0 if mirror_world else 1.0
    quads = [(q, m[k], kf) for q, k, kf in _DOCS_ROOM]
    for sq, h, kfs in (_DOCS_SHORT, _DOCS_TALL):
        quads += [(q, m["block"], kf) for q, kf in zip(_docs_block_quads(_docs_square(*sq), h, h > 200), kfs)]
    for q, mat, kf in quads:
        sb.add_mesh(np.array([(sx * p[0], p[1], p[2]) for p in q], F32), _docs_fan(kf), mat)
    sb.add_mesh(np.array([(sx * p[0], p[1], p[2]) for p in _DOCS_LIGHT], F32), _docs_fan(3, rev=mirror_world), m["white"], emit=(L,) * 3)
    return sb.finish(bvh_builder)


def cornell_docs_render_desc(spp=8, res=500, mirror_camera=True, **kw):
    """Film 500 x 500, Sampler "sobol", Integrator "path" (maxdepth 5), box filter: getting_started.rst:166-172; `Scale -1 1 1` in front of
    LookAt unless mirror_camera=False (then the camera stands at x = -278, for cornell_box_docs(mirror_world=True))."""
    look = CORNELL_DOCS_LOOK_AT if mirror_camera else ((-278, 273, -800), (-278, 273, -799), (0, 1, 0))
    return make_render_desc(res, res, spp, look, CORNELL_DOCS_FOV, mirror_x=mirror_camera, **kw)


CORNELL_FOG = ((0.0002, 0.0002, 0.0003), (0.0016, 0.0016, 0.0014), 0.3)   # per scene unit (the room is 550 units wide): optical depth ~1 across it
CORNELL_LOOK_AT = ((278, 273, -800), (278, 273, 0), (0, 1, 0))
CORNELL_FOV = 39.3


def cornell_render_desc(res=400, spp=64, **kw):
    return make_render_desc(res, res, spp, CORNELL_LOOK_AT, CORNELL_FOV, **kw)


def _splitmix64(seed, n):
    """n outputs of SplitMix64 started at `seed` (vectorised)."""
    with np.errstate(over="ignore"):
        i = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed) + i * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def triangle_soup(bvh_builder, n_tris=1_000_000, seed=0x5EED5EED, extent=0.01, alpha_mask=False):
    """C2: centroids U[-1,1]^3, vertices = centroid + U[-extent,extent]^3, matte 0.5,
    one 1x1 one-sided area light at y=+1.5 facing -y with L = 40.
    alpha_mask: every triangle of the soup carries a "float imagemap" alpha texture (a 64 x 64 disc cut-out over the default uv of a mesh
    without uvs, triangle.rs:97-112) — the foliage case: a candidate hit costs a texture lookup inside the traversal."""
    r = _splitmix64(seed, n_tris * 12)
    u = ((r >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)).reshape(n_tris, 12)
    c = u[:, :3] * 2.0 - 1.0
    off = (u[:, 3:] * 2.0 - 1.0).reshape(n_tris, 3, 3) * extent
    P = (c[:, None, :] + off).astype(F32).reshape(-1, 3)
    idx = np.arange(n_tris * 3, dtype=np.uint32).reshape(-1, 3)
    sb = SceneBuilder()
    grey = sb.add_material(matte((0.5, 0.5, 0.5)))
    mask = None
    if alpha_mask:
        g = (np.arange(64, dtype=np.float64) + 0.5) / 64.0
        disc = (((g[None, :] - 0.62) ** 2 + (g[:, None] - 0.38) ** 2) < 0.33 ** 2).astype(F32)   # centred on the uv triangle (0,0) (1,0) (1,1): about half of it stays
        mask = sb.image_texture(np.repeat(disc[:, :, None], 3, axis=2), channels=1)
    sb.add_mesh(P, idx, grey, alpha=mask)
    sb.add_quad([(0.5, 1.5, -0.5), (0.5, 1.5, 0.5), (-0.5, 1.5, 0.5), (-0.5, 1.5, -0.5)], grey, emit=(40, 40, 40))
    return sb.finish(bvh_builder)


SOUP_LOOK_AT = ((0, 0, -4), (0, 0, 0), (0, 1, 0))
SOUP_FOV = 40.0


def soup_render_desc(res=1024, spp=256, max_depth=8, **kw):
    return make_render_desc(res, res, spp, SOUP_LOOK_AT, SOUP_FOV, max_depth=max_depth, **kw)


def _value_noise(p, seed, octaves=5):
    """Smooth lattice value noise on points p (n,3) (float64), deterministic from `seed`."""
    def h(ix, iy, iz):
        with np.errstate(over="ignore"):
            v = (ix.astype(np.uint64) * np.uint64(73856093)) ^ (iy.astype(np.uint64) * np.uint64(19349663)) ^ (iz.astype(np.uint64) * np.uint64(83492791)) ^ np.uint64(seed)
            v = (v ^ (v >> np.uint64(13))) * np.uint64(0x5BD1E9955BD1E995)
            v = v ^ (v >> np.uint64(15))
        return (v & np.uint64(0xFFFFFF)).astype(np.float64) / float(0xFFFFFF)
    out = np.zeros(len(p)); amp = 1.0; freq = 2.0
    for _ in range(octaves):
        q = p * freq + 64.0
        i = np.floor(q); f = q - i; f = f * f * (3 - 2 * f)
        i = i.astype(np.int64)
        acc = 0
        for dx in (0, 1):
            for dy in (0, 1):
                for dz in (0, 1):
                    w = (f[:, 0] if dx else 1 - f[:, 0]) * (f[:, 1] if dy else 1 - f[:, 1]) * (f[:, 2] if dz else 1 - f[:, 2])
                    acc = acc + w * h(i[:, 0] + dx, i[:, 1] + dy, i[:, 2] + dz)
        out += amp * (acc - 0.5); amp *= 0.5; freq *= 2.0
    return out


def statue_standin(bvh_builder, grid=1466, seed=0x6A4E, textured=False, many_lights=0):
    """C3 stand-in for the off-tree Ganesha scene: grid x grid lat-long sphere
    (2*grid*(grid-1) ~ 4.30 M triangles at 1466) displaced by 5-octave value noise, smooth normals,
    plastic; matte ground; three quad lights.  DECLARED STAND-IN: not the real asset.
    textured=True (the "textured BSDFs" axis of C4): lat-long UVs, a 1024x512 Kd image (EWA),
    a bump map on the body and a checker image on the ground."""
    nu, nv = grid, grid
    th = np.linspace(0, np.pi, nv + 1)[:, None]; ph = np.linspace(0, 2 * np.pi, nu + 1)[None, :]
    d = np.stack([np.sin(th) * np.cos(ph), np.cos(th) * np.ones_like(ph), np.sin(th) * np.sin(ph)], -1).reshape(-1, 3)
    rad = 1.0 + 0.15 * _value_noise(d, seed)
    P = d * rad[:, None]
    # smooth normals from finite differences of the displaced surface
    Pg = P.reshape(nv + 1, nu + 1, 3)
    du = np.roll(Pg, -1, 1) - np.roll(Pg, 1, 1); dv = np.zeros_like(Pg); dv[1:-1] = Pg[2:] - Pg[:-2]; dv[0] = Pg[1] - Pg[0]; dv[-1] = Pg[-1] - Pg[-2]
    N = np.cross(du, dv).reshape(-1, 3)
    ln = np.linalg.norm(N, axis=1); bad = ln < 1e-12
    N[bad] = d[bad]; ln[bad] = 1.0
    N = N / ln[:, None]
    N[(N * d).sum(1) < 0] *= -1
    i, j = np.meshgrid(np.arange(nv), np.arange(nu), indexing="ij")
    a = (i * (nu + 1) + j).reshape(-1); b = a + 1; c = a + (nu + 1); e = c + 1
    idx = np.concatenate([np.stack([a, c, b], 1), np.stack([b, c, e], 1)]).astype(np.uint32)
    # drop the degenerate cap triangles
    Pf = P.astype(F32)
    t0, t1, t2 = Pf[idx[:, 0]], Pf[idx[:, 1]], Pf[idx[:, 2]]
    keep = np.linalg.norm(np.cross(t1 - t0, t2 - t0), axis=1) > 0
    idx = idx[keep]
    sb = SceneBuilder()
    UV = None
    if textured:
        th2 = np.linspace(0, np.pi, 512)[:, None]; ph2 = np.linspace(0, 2 * np.pi, 1024, endpoint=False)[None, :]
        d2 = np.stack([np.sin(th2) * np.cos(ph2), np.cos(th2) * np.ones_like(ph2), np.sin(th2) * np.sin(ph2)], -1).reshape(-1, 3)
        n1 = _value_noise(d2 * 3.0, seed + 1).reshape(512, 1024); n2 = _value_noise(d2 * 9.0, seed + 2).reshape(512, 1024)
        veins = (0.5 + 0.5 * np.sin(12.0 * n1 + 4.0 * n2)).astype(F32)
        img = np.stack([0.25 + 0.55 * veins, 0.22 + 0.45 * veins, 0.2 + 0.35 * veins], -1).astype(F32)
        kd = sb.image_texture(img)
        height = sb.image_texture(np.repeat((0.5 + 0.5 * n2).astype(F32)[..., None], 3, 2), channels=1, scale=0.01, trilinear=True)
        yy, xx = np.mgrid[0:256, 0:256]
        chk = np.where(((xx // 32 + yy // 32) % 2)[..., None] > 0, F32(0.65), F32(0.25)) * np.ones((1, 1, 3), F32)
        body = sb.add_material(plastic(kd, (0.1, 0.1, 0.1), 0.1, bump=height))
        ground = sb.add_material(matte(sb.image_texture(chk.astype(F32), su=6.0, sv=6.0)))
        vj = (np.arange((nv + 1) * (nu + 1)) % (nu + 1)) / float(nu); vi = (np.arange((nv + 1) * (nu + 1)) // (nu + 1)) / float(nv)
        UV = np.stack([vj, 1.0 - vi], 1).astype(F32)
    else:
        body = sb.add_material(plastic((0.4, 0.4, 0.4), (0.1, 0.1, 0.1), 0.1))
        ground = sb.add_material(matte((0.5, 0.5, 0.5)))
    sb.add_mesh(Pf, idx, body, N=N.astype(F32), UV=UV)
    sb.add_quad([(-6, -1.3, -6), (-6, -1.3, 6), (6, -1.3, 6), (6, -1.3, -6)], ground, UV=[[0, 0], [0, 1], [1, 1], [1, 0]] if textured else None)
    if many_lights:  # C4 stand-in: many small area lights instead of three large ones, same total power
        k = int(round(math.sqrt(many_lights)))
        rng = np.random.default_rng(seed)
        for a in range(k):
            for b in range(k):
                cx, cz = -3.5 + 7.0 * (a + 0.5) / k, -3.0 + 6.5 * (b + 0.5) / k
                h = 0.5 * math.sqrt(3.0 / (k * k))  # 3 m^2 of emitter in total, like the three 1 x 1 quads
                L = tuple(float(v) for v in rng.uniform(18, 32, 3))
                sb.add_quad([(cx + h, 3.0, cz - h), (cx + h, 3.0, cz + h), (cx - h, 3.0, cz + h), (cx - h, 3.0, cz - h)], ground, emit=L)
    else:
        for (cx, cz, L) in ((-2.5, -2.0, (30, 28, 24)), (2.5, -2.0, (20, 24, 30)), (0.0, 2.5, (25, 25, 25))):
            sb.add_quad([(cx + 0.5, 3.0, cz - 0.5), (cx + 0.5, 3.0, cz + 0.5), (cx - 0.5, 3.0, cz + 0.5), (cx - 0.5, 3.0, cz - 0.5)], ground, emit=L)
    return sb.finish(bvh_builder)


STATUE_LOOK_AT = ((0, 0.6, -4.2), (0, 0, 0), (0, 1, 0))
STATUE_FOV = 38.0


def read_ply(path):
    """vertices (n, 3) f32, optional normals (n, 3) f32, triangles (m, 3) u32 of a PLY file — ascii or binary_little_endian / big_endian, faces of
    any size triangulated as fans.  What rs_pbrt's `Shape "plymesh"` reads (src/shapes/plymesh.rs:45-340 through ply-rs): x y z [nx ny nz] [u v]
    per vertex, vertex_indices per face.  Only for bench.py's $RSPT_GANESHA_DIR: in the product the mesh arrives flattened from rs_pbrt."""
    f = open(path, "rb")
    if f.readline().strip() != b"ply":
        raise ValueError("%s: not a PLY file" % path)
    fmt, elements = None, []
    while True:
        line = f.readline()
        if not line:
            raise ValueError("%s: header without end_header" % path)
        t = line.decode("ascii", "replace").split()
        if not t or t[0] == "comment" or t[0] == "obj_info":
            continue
        if t[0] == "format":
            fmt = t[1]
        elif t[0] == "element":
            elements.append((t[1], int(t[2]), []))
        elif t[0] == "property":
            elements[-1][2].append(tuple(t[1:]))
        elif t[0] == "end_header":
            break
    types = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4",
             "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}
    end = ">" if fmt == "binary_big_endian" else "<"
    P = N = tri = None
    for name, count, props in elements:
        scalar = all(pr[0] != "list" for pr in props)
        if fmt == "ascii":
            rows = [f.readline().split() for _ in range(count)]
            if name == "vertex":
                cols = {pr[-1]: i for i, pr in enumerate(props)}
                a = np.array(rows, np.float64)
                P = a[:, [cols["x"], cols["y"], cols["z"]]].astype(F32)
                if all(k in cols for k in ("nx", "ny", "nz")):
                    N = a[:, [cols["nx"], cols["ny"], cols["nz"]]].astype(F32)
            elif name == "face":
                faces = [[int(v) for v in r[1:1 + int(r[0])]] for r in rows]
                tri = np.array([[fc[0], fc[i], fc[i + 1]] for fc in faces for i in range(1, len(fc) - 1)], np.uint32)
            continue
        if scalar:
            dt = np.dtype([(pr[-1], end + types[pr[0]]) for pr in props])
            a = np.frombuffer(f.read(dt.itemsize * count), dt, count)
            if name == "vertex":
                P = np.stack([a["x"], a["y"], a["z"]], 1).astype(F32)
                if all(k in a.dtype.names for k in ("nx", "ny", "nz")):
                    N = np.stack([a["nx"], a["ny"], a["nz"]], 1).astype(F32)
            continue
        if name != "face" or len(props) != 1:   # an element with a list property that is not the face list: walk it record by record
            raise ValueError("%s: element %s mixes list and scalar properties" % (path, name))
        ct, it = end + types[props[0][1]], end + types[props[0][2]]
        raw = f.read()
        csz, isz = np.dtype(ct).itemsize, np.dtype(it).itemsize
        first = int(np.frombuffer(raw[:csz], ct)[0])
        rec = csz + first * isz
        if len(raw) >= rec * count and (np.frombuffer(raw[:rec * count], "u1").reshape(count, rec)[:, :csz] == np.frombuffer(raw[:csz], "u1")).all():   # all faces the same size (the usual case): one reshape
            idx = np.frombuffer(np.ascontiguousarray(np.frombuffer(raw[:rec * count], "u1").reshape(count, rec)[:, csz:]).tobytes(), it).reshape(count, first).astype(np.uint32)
            tri = np.concatenate([idx[:, [0, i, i + 1]] for i in range(1, first - 1)], 0)
            f.seek(-(len(raw) - rec * count), 1) if len(raw) > rec * count else None
        else:
            out, off = [], 0
            for _ in range(count):
                k = int(np.frombuffer(raw[off:off + csz], ct)[0]); off += csz
                v = np.frombuffer(raw[off:off + k * isz], it); off += k * isz
                out += [[v[0], v[i], v[i + 1]] for i in range(1, k - 1)]
            tri = np.array(out, np.uint32)
    if P is None or tri is None:
        raise ValueError("%s: no vertex / face element" % path)
    return P, N, tri


def statue_from_ply(bvh_builder, path):
    """C3 with the REAL mesh when $RSPT_GANESHA_DIR holds it: the PLY's triangles, centred and scaled into the
    stand-in's frame (bounding-sphere radius 1.15 at the origin, as the displaced sphere), its normals if it has them, under the stand-in's
    camera, ground, three quad lights and plastic — the asset's own .pbrt (camera, lights, materials) is rs_pbrt's to parse, not this tool's."""
    P, N, tri = read_ply(path)
    lo, hi = P.min(0), P.max(0)
    c = (0.5 * (lo + hi)).astype(F32)
    r = float(np.linalg.norm(P - c, axis=1).max())
    Pf = ((P - c) * F32(1.15 / r)).astype(F32)
    t0, t1, t2 = Pf[tri[:, 0]], Pf[tri[:, 1]], Pf[tri[:, 2]]
    tri = tri[np.linalg.norm(np.cross(t1 - t0, t2 - t0), axis=1) > 0]
    sb = SceneBuilder()
    body = sb.add_material(plastic((0.4, 0.4, 0.4), (0.1, 0.1, 0.1), 0.1))
    ground = sb.add_material(matte((0.5, 0.5, 0.5)))
    sb.add_mesh(Pf, tri, body, N=N)
    sb.add_quad([(-6, -1.3, -6), (-6, -1.3, 6), (6, -1.3, 6), (6, -1.3, -6)], ground)
    for (cx, cz, L) in ((-2.5, -2.0, (30, 28, 24)), (2.5, -2.0, (20, 24, 30)), (0.0, 2.5, (25, 25, 25))):
        sb.add_quad([(cx + 0.5, 3.0, cz - 0.5), (cx + 0.5, 3.0, cz + 0.5), (cx - 0.5, 3.0, cz + 0.5), (cx - 0.5, 3.0, cz - 0.5)], ground, emit=L)
    return sb.finish(bvh_builder), len(tri)


def statue_render_desc(xres=1920, yres=1080, spp=1024, **kw):
    return make_render_desc(xres, yres, spp, STATUE_LOOK_AT, STATUE_FOV, **kw)


# ---------------------------------------------------------------------------------------
# C5 stand-in: instanced geometry under a lat-long sky
# ---------------------------------------------------------------------------------------
def _tree_mesh(grid_u=100, grid_v=50, seed=0x7EE):
    """one "tree": a noise-displaced canopy (2 * grid_u * grid_v triangles) on a thin trunk, ~10 k triangles at 100 x 50"""
    th = np.linspace(0.02, np.pi - 0.02, grid_v + 1)[:, None]; ph = np.linspace(0, 2 * np.pi, grid_u + 1)[None, :]
    d = np.stack([np.sin(th) * np.cos(ph), np.cos(th) * np.ones_like(ph), np.sin(th) * np.sin(ph)], -1).reshape(-1, 3)
    rad = 0.45 + 0.2 * _value_noise(d * 2.0, seed, octaves=3)
    P = d * rad[:, None] * np.array([1.0, 1.4, 1.0]) + np.array([0.0, 1.3, 0.0])
    i, j = np.meshgrid(np.arange(grid_v), np.arange(grid_u), indexing="ij")
    a = (i * (grid_u + 1) + j).reshape(-1); b = a + 1; c = a + (grid_u + 1); e = c + 1
    idx = np.concatenate([np.stack([a, c, b], 1), np.stack([b, c, e], 1)])
    k = 8  # trunk: an open 8-sided prism
    ang = np.linspace(0, 2 * np.pi, k, endpoint=False)
    ring = np.stack([0.06 * np.cos(ang), np.zeros(k), 0.06 * np.sin(ang)], 1)
    T = np.concatenate([ring, ring + np.array([0, 0.9, 0])])
    ti = np.array([[q, (q + 1) % k, k + q] for q in range(k)] + [[(q + 1) % k, k + (q + 1) % k, k + q] for q in range(k)])
    return P.astype(F32), idx.astype(np.uint32), T.astype(F32), ti.astype(np.uint32)


def landscape_standin(bvh_builder, n_side=64, terrain=256, seed=0x1A9D, instancing="reference", tree_grid=(100, 50), moving=False):
    """C5 stand-in for the off-tree Landscape cover scene: n_side^2 ObjectInstances (4096 at 64) of one ~10 k-triangle
    tree object (canopy + trunk, two materials) with per-instance rotation about y, non-uniform scale and translation onto a
    terrain^2-cell height field, under a 64 x 32 lat-long sky with a sun texel (InfiniteAreaLight, importance sampled).
    DECLARED STAND-IN: not the real asset.  instancing: "reference" = rs_pbrt v0.9.12's behaviour (instanced hits carry no
    material: the trees are invisible to camera / bounce rays and cast shadows, Q11) | "fixed".  moving: every instance is a MOVING TransformedPrimitive
    (two keys over the shutter interval 0 .. 1; the render desc then wants shutter=(0, 1))."""
    rng = np.random.default_rng(seed)
    sb = SceneBuilder()
    leaf = sb.add_material(matte((0.12, 0.35, 0.1)))
    bark = sb.add_material(matte((0.3, 0.2, 0.12)))
    soil = sb.add_material(matte((0.35, 0.3, 0.22)))
    g = np.linspace(-1, 1, terrain + 1)
    X, Z = np.meshgrid(g * 40.0, g * 40.0, indexing="xy")
    H = 2.5 * _value_noise(np.stack([X.reshape(-1) / 20.0, np.zeros(X.size), Z.reshape(-1) / 20.0], 1), seed + 1, octaves=4).reshape(X.shape)
    P = np.stack([X, H, Z], -1).reshape(-1, 3)
    i, j = np.meshgrid(np.arange(terrain), np.arange(terrain), indexing="ij")
    a = (i * (terrain + 1) + j).reshape(-1); b = a + 1; c = a + (terrain + 1); e = c + 1
    sb.add_mesh(P.astype(F32), np.concatenate([np.stack([a, c, b], 1), np.stack([b, c, e], 1)]).astype(np.uint32), soil)
    cp, ci, tp, ti = _tree_mesh(*tree_grid)
    sb.begin_object("tree")
    sb.add_mesh(cp, ci, leaf)
    sb.add_mesh(tp, ti, bark)
    sb.end_object()
    cell = 70.0 / n_side
    for r in range(n_side):
        for q in range(n_side):
            x = -35.0 + cell * (q + 0.5) + float(rng.uniform(-0.3, 0.3)) * cell
            z = -35.0 + cell * (r + 0.5) + float(rng.uniform(-0.3, 0.3)) * cell
            fx, fz = (x / 40.0 + 1) * terrain / 2, (z / 40.0 + 1) * terrain / 2
            y = float(H[min(int(fz), terrain), min(int(fx), terrain)]) - 0.05
            s_ = float(rng.uniform(0.7, 1.3))
            xf = Transform.translate((x, y, z)) * Transform.rotate_y(float(rng.uniform(0, 360))) * Transform.scale(s_, s_ * float(rng.uniform(0.8, 1.4)), s_)
            if moving:   # every tree sways over the shutter: the end key is turned 6 degrees about its own axis and shifted by a hand's width — keys that differ by a
                # rotation, so the top-level box of every instance is AnimatedTransform::motion_bounds' per-corner form (rspt_motion_bounds)
                xf_end = Transform.translate((0.1 * s_, 0.0, 0.05 * s_)) * xf * Transform.rotate_y(6.0)
                sb.add_instance("tree", xf, xf_end)
            else:
                sb.add_instance("tree", xf)
    sky = np.zeros((32, 64, 3), F32)
    t = np.linspace(0, 1, 32)[:, None]
    sky[..., 0] = 0.25 + 0.35 * t; sky[..., 1] = 0.4 + 0.3 * t; sky[..., 2] = 0.75 + 0.1 * t
    sky[16:] *= 0.15                      # below the horizon
    sky[6, 20] = (400.0, 380.0, 300.0)    # the sun
    sb.add_infinite_light((1.0, 1.0, 1.0), image=sky)
    return sb.finish(bvh_builder, instancing=instancing)


LANDSCAPE_LOOK_AT = ((0, 9, -42), (0, 1.5, 0), (0, 1, 0))


def landscape_render_desc(xres=1920, yres=1080, spp=4096, **kw):
    return make_render_desc(xres, yres, spp, LANDSCAPE_LOOK_AT, 45.0, **kw)
