"""Host-side mirror of the reference's operator for this path, for Python callers and tests:

    PathIntegrator::new(max_depth, camera, sampler, pixel_bounds, rr_threshold, light_sample_strategy)
                                                                    (src/integrators/path.rs:38-54)
    Integrator::render(&mut self, scene: &Scene, num_threads: u8)   (src/core/integrator.rs:39-46,70)

Same names, same defaults as CreatePathIntegrator (src/core/api.rs:285-321: maxdepth 5,
rrthreshold 1.0, lightsamplestrategy "spatial"), same observable result: render() fills the
film's pixels (xyz + filter_weight_sum, src/core/film.rs:38-43).  Where the reference panics
(no camera) this raises, where it only warns (an unknown light sample strategy falls back to "spatial") this does the same; when the GPU or librspt.so is missing it raises
lib.RsptError — there is no CPU loop here."""
import numpy as np

from . import abi, lib, scenes

_STRATEGIES = {"uniform": abi.LIGHTS_UNIFORM, "power": abi.LIGHTS_POWER, "spatial": abi.LIGHTS_SPATIAL}


class Film:
    """What Film.pixels holds after the render (film.rs:159-173) plus write_image's linear RGB."""

    def __init__(self, rd):
        self.full_resolution = (rd.full_res[0], rd.full_res[1])
        self.cropped_pixel_bounds = tuple(rd.crop_px)
        self.pixels = None  # (h, w, 4): xyz, filter_weight_sum

    @property
    def shape(self):
        x0, y0, x1, y1 = self.cropped_pixel_bounds
        return (y1 - y0, x1 - x0)

    def rgb(self):
        """Film::write_image before gamma / quantisation (film.rs:445-462)."""
        h, w = self.shape
        return scenes.film_to_rgb(self.pixels.reshape(-1, 4)).reshape(h, w, 3)


class PathIntegrator:
    def __init__(self, max_depth=5, camera=None, sampler=None, pixel_bounds=None, rr_threshold=1.0, light_sample_strategy="spatial"):
        """camera: an rspt_render_desc carrying camera + film + sampler parameters
        (scenes.make_render_desc); sampler / pixel_bounds are accepted for signature parity:
        the sampler is the Sobol' sampler configured in the desc, pixel_bounds equal the film's
        sample bounds (the reference ignores the "pixelbounds" parameter too, api.rs:288-304)."""
        if camera is None:
            raise ValueError("Unable to create camera")  # api.rs:467-483 panics the same way
        if light_sample_strategy not in _STRATEGIES:
            # an unknown name does not raise: create_light_sample_distribution (lightdistrib.rs:393-418) warns and falls back to "spatial"
            light_sample_strategy = "spatial"
        self.camera = camera
        self.max_depth = int(max_depth)
        self.rr_threshold = float(rr_threshold)
        self.light_sample_strategy = light_sample_strategy
        self.stats = None

    def _desc(self, shard=None):
        rd = abi.RenderDesc.from_buffer_copy(self.camera)
        rd.max_depth = self.max_depth
        rd.rr_threshold = self.rr_threshold
        rd.light_strategy = _STRATEGIES[self.light_sample_strategy]
        if shard is not None:
            rd.shard_index, rd.shard_count, rd.tile_chunk = shard
        return rd

    def render(self, scene, num_threads=0, shard=None):
        """scene: scenes.Scene (flattened Scene + BVHAccel) or an already uploaded lib.DeviceScene.
        num_threads maps to nothing on the GPU (kept for call compatibility).  Returns the Film."""
        rd = self._desc(shard)
        own = not isinstance(scene, lib.DeviceScene)
        ds = lib.DeviceScene(scene) if own else scene
        try:
            pixels, self.stats = lib.render(ds, rd)
        finally:
            if own:
                ds.close()
        film = Film(rd)
        h, w = film.shape
        film.pixels = np.asarray(pixels, np.float32).reshape(h, w, 4)
        return film


class AOIntegrator(PathIntegrator):
    """AOIntegrator::new(cos_sample, n_samples, camera, sampler, pixel_bounds) (src/integrators/ao.rs:28-45), created
    by the "ao" / "ambientocclusion" integrator name with defaults cossample = true, nsamples = 64 (api.rs:411-440).
    Shares SamplerIntegrator::render with the path integrator; only `li` differs."""

    def __init__(self, cos_sample=True, n_samples=64, camera=None, sampler=None, pixel_bounds=None):
        super().__init__(camera=camera, sampler=sampler, pixel_bounds=pixel_bounds)
        self.cos_sample = bool(cos_sample)
        self.n_samples = int(n_samples)

    def _desc(self, shard=None):
        rd = super()._desc(shard)
        rd.integrator = abi.INTEGRATOR_AO
        rd.ao_n_samples, rd.ao_cos_sample = self.n_samples, int(self.cos_sample)
        return rd


class DirectLightingIntegrator(PathIntegrator):
    """DirectLightingIntegrator::new(strategy, max_depth, camera, sampler, pixel_bounds) (src/integrators/directlighting.rs:38-53), created
    by the "directlighting" integrator name with defaults strategy "all", maxdepth 5 (api.rs:322-349).  Materials must have been
    flattened with allow_multiple_lobes = false (scenes.glass(multiple_lobes=False)): `li` calls compute_scattering_functions(ray, false, ..).
    light_samples: Light::get_n_samples() per light (UniformSampleAll's sample arrays), default 1 each."""

    def __init__(self, strategy="all", max_depth=5, camera=None, sampler=None, pixel_bounds=None, light_samples=None):
        super().__init__(max_depth=max_depth, camera=camera, sampler=sampler, pixel_bounds=pixel_bounds)
        if strategy not in ("all", "one"):
            strategy = "all"  # api.rs:332-341: unknown strategy -> warning, UniformSampleAll
        self.strategy = strategy
        self.light_samples = None if light_samples is None else np.ascontiguousarray(light_samples, np.int32)

    def _desc(self, shard=None):
        rd = super()._desc(shard)
        rd.integrator = abi.INTEGRATOR_DIRECT
        rd.direct_strategy = abi.DIRECT_SAMPLE_ALL if self.strategy == "all" else abi.DIRECT_SAMPLE_ONE
        rd.n_light_samples = self.light_samples.ctypes.data if self.light_samples is not None else None
        return rd


class VolPathIntegrator(PathIntegrator):
    """VolPathIntegrator::new(max_depth, camera, sampler, pixel_bounds, rr_threshold, light_sample_strategy) (src/integrators/volpath.rs:38-55),
    created by the "volpath" integrator name with the path integrator's defaults (api.rs:350-380).  Media come with the scene
    (SceneBuilder.add_medium, add_mesh(medium=(inside, outside))); materials are flattened with allow_multiple_lobes = true as for "path"."""

    def _desc(self, shard=None):
        rd = super()._desc(shard)
        rd.integrator = abi.INTEGRATOR_VOLPATH
        return rd


class Checkpoint:
    """Checkpoint / resume around any of the integrators above ("film buffer + next-sample-index is the whole
    state"; the reference's render loop is one-shot).  Renders the frame in slices of the per-pixel sample index through
    rspt_render_desc.sample_begin / sample_count and keeps the running sum of the films, which IS the film of the samples done so
    far (xyz and filter_weight_sum are sums over samples).  save() / load() carry the state across processes: a run that died at
    sample k resumes at k; a multi-GPU driver re-renders a lost rank's shard the same way (tiles and sample ranges are idempotent).
    Sobol' and Halton only: the pixel samplers chain all samples of a tile."""

    def __init__(self, integrator, shard=None):
        self.integrator, self.shard = integrator, shard
        self.rd = integrator._desc(shard)
        self.next_sample = 0
        npix = (self.rd.crop_px[2] - self.rd.crop_px[0]) * (self.rd.crop_px[3] - self.rd.crop_px[1])
        self.sum = np.zeros((npix, 4), np.float64)   # accumulate wider than the slices so that many small slices do not drift

    @property
    def done(self):
        return self.next_sample >= int(self.rd.spp)

    def step(self, dscene, n_samples):
        """render the next n_samples samples of every pixel (fewer at the end); returns how many were rendered"""
        n = min(int(n_samples), int(self.rd.spp) - self.next_sample)
        if n <= 0:
            return 0
        rd = abi.RenderDesc.from_buffer_copy(self.rd)
        rd.sample_begin, rd.sample_count = self.next_sample, n
        pixels, self.integrator.stats = lib.render(dscene, rd)
        self.sum += pixels
        self.next_sample += n
        return n

    def film(self):
        f = Film(self.rd)
        h, w = f.shape
        f.pixels = self.sum.astype(np.float32).reshape(h, w, 4)
        return f

    def _identity(self, scene_id):
        """what the partial film is a film OF: every render parameter except the sample slice (pointer fields excluded: they are
        per-process), plus the caller's scene identifier — a checkpoint of another scene, integrator, sampler, depth or camera must not
        be summed with this one's samples"""
        import ctypes as C
        import hashlib
        rd = abi.RenderDesc.from_buffer_copy(self.rd)
        rd.sample_begin = rd.sample_count = 0
        rd.film_reduce = 0; rd.allow_slow_paths = 0   # options of HOW the frame is rendered, not of which frame: a resumed run may differ in them
        rd.n_light_samples = None; rd.maxmin_c_pixel = None
        rd.tables = abi.SamplerTables()
        h = hashlib.sha256(bytes(memoryview(rd).cast("B")))
        ls = getattr(self.rd, "_light_samples", None)
        if ls is not None:
            h.update(ls.tobytes())
        h.update(str(scene_id).encode())
        return h.hexdigest()

    @staticmethod
    def _path(path):
        return str(path) if str(path).endswith(".npz") else str(path) + ".npz"

    def save(self, path, scene_id=""):
        """atomic (written beside the target, then renamed over it): a crash during save leaves the previous checkpoint intact"""
        import os
        path = self._path(path)
        tmp = path + ".tmp.npz"
        np.savez(tmp, sum=self.sum, next_sample=self.next_sample, identity=self._identity(scene_id))
        os.replace(tmp, path)

    def load(self, path, scene_id=""):
        z = np.load(self._path(path))
        if "identity" not in z.files or str(z["identity"]) != self._identity(scene_id) or z["sum"].shape != self.sum.shape:   # (no identity: a file of the first checkpoint format)
            raise ValueError("checkpoint belongs to another render (scene / integrator / sampler / camera / film / shard differ)")
        self.sum, self.next_sample = z["sum"].astype(np.float64), int(z["next_sample"])
