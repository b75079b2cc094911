// librspt.so — host side of the C ABI in include/rspt.h: scene upload, wavefront scheduling of
// the gfx950 kernels (kernels.h), film read-back.  gfx950 only; there is no CPU fallback: every
// entry point that needs the GPU fails with RSPT_E_NODEVICE / RSPT_E_HIP when it is not there.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <functional>
#include <limits>
#include <map>
#include <string>
#include <vector>

#include "../../include/rspt.h"
#include "material_assembly.h"
#include "camera_anim.h"
#include "motion_bounds.h"
#include "kernels.h"
#include "trace_w4.h"
#include "trace_w4q.h"
#include "bvh_device.h"
#include "direct.h"
#include "vol.h"
#include "tile_serial.h"
#include "lane_serial.h"
// the heavy kernel templates are compiled in tu_*.hip; here they are only declared (tu_decl.h)
#define RSPT_TU_X extern
#define RSPT_TU_ALL
#include "tu_decl.h"

using namespace rspt;

namespace {

thread_local std::string g_err;
int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIP_TRY(expr)                                                                                          \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) return fail(RSPT_E_HIP, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
    } while (0)

struct Ctx {
    bool inited = false;
    int device = 0;
    int n_cus = 256;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;  // shadow-ray trace launches run beside the closest-hit launches (RSPT_TRACE_STREAMS)
    // path state
    size_t cap = 0;
    PathBuf pb{};
    // round 6, MOVING path state (kernels.h PathBuf::move): the second set of the fields a path carries from position to position (iteration it reads set it & 1: set 0 =
    // pb's own arrays, set 1 = these), the original-slot words of both sets, and the radiance of ended paths by original slot
    struct MoveSet { rspt_ray* ray_cont = nullptr; float4* L_eta = nullptr; float4* beta = nullptr; float4* nee_c1 = nullptr; uint64_t* sobol_index = nullptr;
                     uint32_t* state = nullptr; uint32_t* orig[2] = {nullptr, nullptr}; float4* L_final = nullptr /* the second carried-radiance array (move_pathbuf) */; size_t cap = 0; } mv;
    uint32_t* q[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};  // [parity][active, closest, any]
    QueueCounts* cnt = nullptr;
    uint32_t n_cnt = 0;
    uint32_t* ovf = nullptr;           // queue entries of rays whose traversal stack overflowed (k_trace_fixup's work list)
    size_t ovf_cap = 0;
    uint2* spill = nullptr;            // k_trace_w4's stack rows beyond its LDS column (trace_w4.h)
    size_t spill_threads = 0;
    VolBuf vol{};                      // volpath: per-path medium / shadow-ray state (vol.h)
    size_t vol_cap = 0;
    CamAnim* cam_anim = nullptr;       // a moving camera's key decompositions (dev_scene.h), written per render
    DlBuf dl{};                        // directlighting: per-node arrays (direct.h) + level queues
    uint32_t* dl_queue = nullptr;
    size_t dl_cap = 0;
    uint8_t* bin_keys = nullptr;       // K7b: class of every active-queue entry, the queue sorted by class, per-iteration bin bookkeeping
    uint32_t* q_sorted = nullptr;
    size_t bin_cap = 0;
    BinInfo* bin_info = nullptr;
    uint32_t n_bin_info = 0;
    uint32_t* hit_inst = nullptr;      // per path slot: instance of the continuation ray's hit (scenes with object instances)
    size_t hit_inst_cap = 0;
    float* path_time = nullptr;        // per path slot: Ray.time (scenes with moving instances)
    size_t time_cap = 0;
    unsigned long long* totals = nullptr;  // [0] nodes [1] tris [2] bsdf hits [3] rays closest [4] rays any [5] nan samples
    // sampler tables + filter table
    uint32_t* sobol32 = nullptr;
    uint64_t* vdc = nullptr;
    uint64_t* vdc_inv = nullptr;
    float* filter_table = nullptr;
    uint32_t* primes = nullptr;        // Halton: PRIMES, PRIME_SUMS, RADICAL_INVERSE_PERMUTATIONS prefix
    uint32_t* prime_sums = nullptr;
    uint16_t* halton_perms = nullptr;
    uint64_t n_halton_perms = 0;
    std::vector<uint32_t> host_primes, host_prime_sums;
    // film
    float4* film_own = nullptr;
    float4* film_splat = nullptr;
    float4* film_out = nullptr;
    size_t film_px = 0;
    uint32_t* pix_list = nullptr;
    size_t pix_cap = 0;
    uint32_t* pix_index = nullptr;     // k_film_gather: pixel of the sample-bounds grid -> its position in the pixel list
    size_t pix_index_cap = 0;
    std::vector<hipEvent_t> events;
    uint32_t tex_rows = 0;             // rows of g.pb.tex per path: RSPT_TEX_ROWS, + RSPT_DYN_ROWS once a scene with dynamic materials was rendered
    QueueCounts* look = nullptr;       // pinned host words the render loops read queue lengths into: a device-to-host copy into pageable memory
                                       // goes through a staging buffer and costs milliseconds per look (volpath looks once per pass)
};
Ctx g;

// X1: the one collective of the multi-GPU decomposition, ncclReduce(sum) of the per-rank films onto rank 0
// over xGMI.  RCCL is bound at run time so that single-GPU hosts need no librccl; a process that already has one loaded
// (torch ships its own copy) shares it.
// Before that reduce the ranks agree on whether every one of them got that far (ncclAllReduce(max) of one status word): a rank whose
// render failed contributes 1 from rspt_render's exit path, and all ranks return (RSPT_E_PEER for the healthy ones) instead of
// waiting in a collective that one member never enters.
// The handful of RCCL types the dlopen'ed entry points take, declared here with rccl.h's values (ncclInt32 = 2, ncclFloat32 = 7,
// ncclSum = 0, ncclMax = 2, NCCL_UNIQUE_ID_BYTES = 128) so that a host without the RCCL development headers still builds the library.
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt32 = 2, ncclFloat32 = 7 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclMax = 2 } ncclRedOp_t;
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 0;
    int32_t* status = nullptr;      // device word of the status agreement
    int32_t* status_host = nullptr; // its pinned host twin
    bool status_exchanged = false;  // this rspt_render call has taken part in the agreement
};
Rccl rc_;

int rccl_bind() {
    if (rc_.handle) return RSPT_OK;
    const char* names[] = {getenv("RSPT_RCCL_LIB"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    // one policy for every caller (rspt_comm_library): a librccl the process has mapped ALREADY wins — a second copy of RCCL in one process
    // means two sets of bootstrap threads and IPC state.  /proc/self/maps names it whatever name it was loaded by (torch ships its own copy).
    if (FILE* maps = fopen("/proc/self/maps", "r")) {
        char line[4096];
        while (!h && fgets(line, sizeof line, maps)) {
            char* path = strchr(line, '/');
            if (!path || !strstr(path, "librccl.so")) continue;
            path[strcspn(path, "\n")] = 0;
            h = dlopen(path, RTLD_NOW | RTLD_NOLOAD);
        }
        fclose(maps);
    }
    for (const char* n : names)
        if (!h && n && *n && (h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    for (size_t i = 0; !h && i < sizeof names / sizeof *names; i++)
        if (names[i] && *names[i]) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!h) return fail(RSPT_E_UNSUPPORTED, "librccl.so not found (%s); set RSPT_RCCL_LIB", dlerror());
    *(void**)&rc_.GetUniqueId = dlsym(h, "ncclGetUniqueId");
    *(void**)&rc_.CommInitRank = dlsym(h, "ncclCommInitRank");
    *(void**)&rc_.Reduce = dlsym(h, "ncclReduce");
    *(void**)&rc_.AllReduce = dlsym(h, "ncclAllReduce");
    *(void**)&rc_.CommDestroy = dlsym(h, "ncclCommDestroy");
    *(void**)&rc_.GetErrorString = dlsym(h, "ncclGetErrorString");
    if (!rc_.GetUniqueId || !rc_.CommInitRank || !rc_.Reduce || !rc_.AllReduce || !rc_.CommDestroy || !rc_.GetErrorString)
        return fail(RSPT_E_UNSUPPORTED, "librccl.so lacks an nccl* entry point");
    rc_.handle = h;
    return RSPT_OK;
}
#define RCCL_TRY(expr)                                                                                              \
    do {                                                                                                           \
        ncclResult_t r_ = (expr);                                                                                  \
        if (r_ != ncclSuccess) return fail(RSPT_E_HIP, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr, rc_.GetErrorString(r_)); \
    } while (0)

// The status agreement in front of the film reduce: every rank contributes 0 (ready to reduce) or 1 (failed); returns the maximum.
// The device word and its pinned host twin are allocated by rspt_comm_init, so that a rank that reports a failure allocates nothing on the
// way.  What the agreement covers: every failure that leaves the device usable (argument validation, UNSUPPORTED, out of memory).  After a
// sticky HIP error (a kernel fault) the copies below fail too and this rank cannot enter the collective: its peers are then released by the
// communicator's own time-out / abort, not by this word.
int film_reduce_agree(int32_t mine, int32_t* all) {
    if (!rc_.status) HIP_TRY(hipMalloc((void**)&rc_.status, sizeof(int32_t)));
    if (!rc_.status_host) HIP_TRY(hipHostMalloc((void**)&rc_.status_host, sizeof(int32_t), hipHostMallocDefault));
    *rc_.status_host = mine;
    HIP_TRY(hipMemcpyAsync(rc_.status, rc_.status_host, sizeof mine, hipMemcpyHostToDevice, g.stream));
    rc_.status_exchanged = true;
    RCCL_TRY(rc_.AllReduce(rc_.status, rc_.status, 1, ncclInt32, ncclMax, rc_.comm, g.stream));
    HIP_TRY(hipMemcpyAsync(rc_.status_host, rc_.status, sizeof *all, hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    *all = *rc_.status_host;
    return RSPT_OK;
}

struct LightDist {
    float* func = nullptr;
    float* cdf = nullptr;
    float* func_int = nullptr;
    int32_t nvox[3] = {1, 1, 1};
    int32_t spatial = 0;
    // on-demand voxels: table (one int32 per voxel), the list of voxels claimed in the current round, the counters
    int32_t* table = nullptr;
    uint32_t* new_list = nullptr;
    LightLazy* lazy = nullptr;
    uint32_t max_rows = 0;
};

}  // namespace

struct rspt_scene_s {
    SceneDev dev{};
    const PairNode* pairs = nullptr;  // one per interior LinearBVHNode (trace_wide.h)
    const Wide4Node* w4 = nullptr;    // one per interior LinearBVHNode at even depth (trace_w4.h)
    const uint2* big_leaves = nullptr;
    uint32_t w4_root = 0;
    uint32_t w4_top = 0;              // records kept in LDS by k_trace_w4 (a breadth-first prefix, numbered first)
    bool w4_ok = false;               // false: too large for the ref fields, k_trace_pw serves the scene
    const Quad4Node* w4q = nullptr;   // the same records on an 8-bit grid, for the shadow-ray kernel k_trace_w4q (trace_w4q.h): plain scenes only (no instances, no alpha masks)
    const float4* leaf_boxes = nullptr;  // [2 * first primitive of a leaf]: the leaf's LinearBVHNode bounds, for that kernel's exact leaf test
    int any_q_choice = -1;            // which kernel serves this scene's shadow rays: -1 not measured yet (the plain one until then), 0 k_trace_w4<true>, 1 k_trace_w4q — set by
                                      // the first large shadow-ray launch of a render, which runs both on the same rays and keeps the faster (render_impl tune_any)
    TexTables tex{};                  // textures / images / per-material slots (dev_texture.h); has_textures says whether set
    bool has_textures = false;
    // Lobe lists as material_assembly.h built them, once per value of the integrator's allow_multiple_lobes ([0]: true — path,
    // volpath, ao; [1]: false — directlighting: glass as SpecularReflection + SpecularTransmission, glass.rs:136-188).
    // select_materials() points dev / tex at the set a render needs.
    struct MatSet {
        const rspt_material* materials = nullptr;
        const rspt_bxdf* bxdfs = nullptr;
        const uint32_t* mat_slots = nullptr;
        const uint8_t* mat_flags = nullptr;
        const rspt_mat::DynMaterial* dyn = nullptr;
        bool textured = false, dynamic = false;
    } mat_set[2];
    bool has_dynamic = false;         // (in the selected set) some material's lobe list is built per hit
    uint32_t shade_features = 0;      // SF_* (dev_bsdf.h) of everything the scene can put in front of the shade stage
    uint32_t shade_classes = 1;       // distinct lobe-list shapes among the materials: K7b sorts the shade queue by material only when waves would differ
    void select_materials(bool allow_multiple_lobes) {
        const MatSet& m = mat_set[allow_multiple_lobes ? 0 : 1];
        dev.materials = m.materials; dev.bxdfs = m.bxdfs;
        tex.mat_slots = m.mat_slots; tex.mat_flags = m.mat_flags;
        dev.mat_flags = m.textured ? m.mat_flags : nullptr;
        dev.dyn = m.dyn; tex.dyn = m.dyn;
        has_textures = m.textured;
        has_dynamic = m.dynamic;
    }
    std::vector<void*> allocs;
    bool has_null_material = false;
    uint32_t n_materials = 0;
    bool has_alpha = false;           // some mesh carries an alpha / shadow-alpha mask (Triangle::intersect's alpha tests)
    bool alpha_simple = false;        // ... and every mask is a constant or a uv-mapped image (dev_scene.h AlphaMask): k_trace_w4<.., ALPHA = 2>
    std::vector<uint64_t> image_base; // per image: its first float in the texel pool
    bool has_instances = false;       // object instances: two-level traversal (kernels.h traverse<ANY, true>)
    bool has_animated = false;        // ... some of them moving (dev_scene.h inst_at): the reference-order kernel serves the scene, `path` under Sobol' / Halton only
    std::map<int, LightDist> light_dists;  // by effective strategy
};

namespace {

template <class T>
int dev_alloc(T** out, size_t n) {
    *out = nullptr;
    if (n == 0) return RSPT_OK;
    HIP_TRY(hipMalloc((void**)out, n * sizeof(T)));
    return RSPT_OK;
}
template <class T>
int upload(rspt_scene_s* s, const T* host, size_t n, const T** out) {
    *out = nullptr;
    if (n == 0 || !host) return RSPT_OK;
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, n * sizeof(T)));
    s->allocs.push_back(d);
    HIP_TRY(hipMemcpy(d, host, n * sizeof(T), hipMemcpyHostToDevice));
    *out = (const T*)d;
    return RSPT_OK;
}

uint32_t grid_for(uint32_t waves_per_cu_blocks) { return (uint32_t)g.n_cus * waves_per_cu_blocks; }

size_t env_size(const char* name, size_t dflt) {
    const char* v = getenv(name);
    if (!v || !*v) return dflt;
    return (size_t)strtoull(v, nullptr, 0);
}

void free_move() {
    void* ptrs[] = {g.mv.ray_cont, g.mv.L_eta, g.mv.beta, g.mv.nee_c1, g.mv.sobol_index, g.mv.state, g.mv.orig[0], g.mv.orig[1], g.mv.L_final};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    g.mv = Ctx::MoveSet{};
}
void free_paths() {
    free_move();
    void* ptrs[] = {g.pb.ray_cont, g.pb.ray_mis, g.pb.ray_sh, g.pb.hit_cont, g.pb.hit_mis, g.pb.occluded, g.pb.L_eta, g.pb.beta,
                    g.pb.nee_c1, g.pb.nee_c2, g.pb.nee_beta, g.pb.sobol_index, g.pb.state, g.pb.p_film, g.pb.tex, g.pb.dyn_built,
                    g.q[0][0], g.q[0][1], g.q[0][2], g.q[1][0], g.q[1][1], g.q[1][2]};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (g.path_time) (void)hipFree(g.path_time);   // sized with the path buffers (render_impl), freed with them
    g.path_time = nullptr; g.time_cap = 0;
    g.pb = PathBuf{};
    g.tex_rows = 0;
    for (auto& a : g.q) a[0] = a[1] = a[2] = nullptr;
    g.cap = 0;
}

int ensure_paths(size_t cap) {
    if (g.cap >= cap) return RSPT_OK;
    free_paths();
    int rc;
#define A(field, n) if ((rc = dev_alloc(&field, (n))) != RSPT_OK) return rc
    A(g.pb.ray_cont, cap); A(g.pb.ray_mis, cap); A(g.pb.ray_sh, cap);
    A(g.pb.hit_cont, cap); A(g.pb.hit_mis, cap); A(g.pb.occluded, cap);
    A(g.pb.L_eta, cap); A(g.pb.beta, cap); A(g.pb.nee_c1, cap); A(g.pb.nee_c2, cap); A(g.pb.nee_beta, cap);
    A(g.pb.sobol_index, cap); A(g.pb.state, cap); A(g.pb.p_film, cap);
    for (int par = 0; par < 2; par++) {
        A(g.q[par][0], cap);
        A(g.q[par][1], 2 * cap);  // continuation + MIS rays
        A(g.q[par][2], cap);
    }
#undef A
    g.cap = cap;
    return RSPT_OK;
}

// the second set of the MOVE schedule (116 B per path on top of ensure_paths' 280), sized like the first
int ensure_move(size_t cap) {
    if (g.mv.cap >= cap) return RSPT_OK;
    free_move();
    int rc;
    if ((rc = dev_alloc(&g.mv.ray_cont, cap)) || (rc = dev_alloc(&g.mv.L_eta, cap)) || (rc = dev_alloc(&g.mv.beta, cap)) || (rc = dev_alloc(&g.mv.nee_c1, cap)) ||
        (rc = dev_alloc(&g.mv.sobol_index, cap)) || (rc = dev_alloc(&g.mv.state, cap)) || (rc = dev_alloc(&g.mv.orig[0], cap)) || (rc = dev_alloc(&g.mv.orig[1], cap)) ||
        (rc = dev_alloc(&g.mv.L_final, cap)))
        return rc;
    g.mv.cap = cap;
    return RSPT_OK;
}
// the PathBuf of wavefront iteration `it` under the MOVE schedule, which starts at iteration `first` (RSPT_MOVE_FROM: the iterations before it run the slot-for-life
// kernel on set 0 = pb's own arrays): iteration it >= first reads set (it - first) & 1 and writes the other; at it == first the queue still holds original slots.
// The radiance is the exception: pb.L_eta — by original slot — is what k_film reads (L_final), so the carried radiance alternates between the two arrays of the
// second set and only the first MOVE launch reads it from pb.L_eta, where the slot-for-life launches (or nothing: a fresh batch) left it.
PathBuf move_pathbuf(uint32_t it, uint32_t first) {
    PathBuf b = g.pb;
    const Ctx::MoveSet& m = g.mv;
    b.L_final = g.pb.L_eta;
    if (it < first) return b;
    b.move = 1u;
    b.orig_is_p = it == first ? 1u : 0u;
    it -= first;
    rspt_ray* rc[2] = {g.pb.ray_cont, m.ray_cont}; float4* be[2] = {g.pb.beta, m.beta}; float4* c1[2] = {g.pb.nee_c1, m.nee_c1};
    uint64_t* so[2] = {g.pb.sobol_index, m.sobol_index}; uint32_t* st[2] = {g.pb.state, m.state}; float4* le[2] = {m.L_eta, m.L_final};
    const int r = (int)(it & 1u), w = r ^ 1;
    b.ray_cont = rc[r]; b.beta = be[r]; b.nee_c1 = c1[r]; b.sobol_index = so[r]; b.state = st[r]; b.orig = m.orig[r];
    b.L_eta = it == 0 ? g.pb.L_eta : le[w];   // (written by iteration it - 1 into le[(it - 1) & 1] = le[w])
    b.o_ray_cont = rc[w]; b.o_L_eta = le[r]; b.o_beta = be[w]; b.o_nee_c1 = c1[w]; b.o_sobol_index = so[w]; b.o_state = st[w]; b.o_orig = m.orig[w];
    return b;
}

// per-path rows of k_texture's results, only for scenes with textures (6 x 16 B per path)
int ensure_tex_rows(bool dynamic) {
    const uint32_t rows = RSPT_TEX_ROWS + (dynamic ? RSPT_DYN_ROWS : 0);
    if (g.pb.tex && g.tex_rows >= rows) return RSPT_OK;
    if (g.pb.tex) (void)hipFree(g.pb.tex);
    g.pb.tex = nullptr; g.tex_rows = 0;
    int rc = dev_alloc(&g.pb.tex, g.cap * rows);
    if (rc) return rc;
    g.pb.tex_stride = (uint32_t)g.cap;
    g.tex_rows = rows;
    return RSPT_OK;
}
// dynamic materials: one Built record (8 lobes) per thread of the widest shade-stage launch
int ensure_dyn_built(uint32_t threads) {
    if (g.pb.dyn_built && g.pb.dyn_threads >= threads) return RSPT_OK;
    if (g.pb.dyn_built) (void)hipFree(g.pb.dyn_built);
    g.pb.dyn_built = nullptr; g.pb.dyn_threads = 0;
    int rc = dev_alloc(&g.pb.dyn_built, threads);
    if (rc) return rc;
    g.pb.dyn_threads = threads;
    return RSPT_OK;
}

int ensure_direct(size_t cap) {
    if (g.dl_cap >= cap) return RSPT_OK;
    void* old[] = {g.dl.le_kind, g.dl.w_r, g.dl.w_t, g.dl.l_all, g.dl.ld_acc, g.dl.dim, g.dl.kidx, g.dl.nflags, g.dl.error, g.dl_queue};
    for (void* p : old) if (p) (void)hipFree(p);
    g.dl = DlBuf{}; g.dl_queue = nullptr; g.dl_cap = 0;
    int rc;
    if ((rc = dev_alloc(&g.dl.le_kind, cap)) || (rc = dev_alloc(&g.dl.w_r, cap)) || (rc = dev_alloc(&g.dl.w_t, cap)) || (rc = dev_alloc(&g.dl.l_all, cap)) ||
        (rc = dev_alloc(&g.dl.ld_acc, cap)) || (rc = dev_alloc(&g.dl.dim, cap)) || (rc = dev_alloc(&g.dl.kidx, cap)) || (rc = dev_alloc(&g.dl.nflags, cap)) ||
        (rc = dev_alloc(&g.dl.error, 1)) || (rc = dev_alloc(&g.dl_queue, cap)))
        return rc;
    g.dl_cap = cap;
    return RSPT_OK;
}

int ensure_vol(size_t cap) {
    if (g.vol_cap >= cap) return RSPT_OK;
    void* old[] = {g.vol.medium, g.vol.sh, g.vol.p1_p, g.vol.p1_e, g.vol.p1_n, g.vol.post, g.vol.truncated};
    for (void* p : old) if (p) (void)hipFree(p);
    g.vol = VolBuf{}; g.vol_cap = 0;
    int rc;
    if ((rc = dev_alloc(&g.vol.medium, cap)) || (rc = dev_alloc(&g.vol.sh, cap)) || (rc = dev_alloc(&g.vol.p1_p, cap)) || (rc = dev_alloc(&g.vol.p1_e, cap)) ||
        (rc = dev_alloc(&g.vol.p1_n, cap)) || (rc = dev_alloc(&g.vol.post, cap)) || (rc = dev_alloc(&g.vol.truncated, 1)))
        return rc;
    g.vol_cap = cap;
    return RSPT_OK;
}

int ensure_bins(size_t cap, uint32_t n_iters) {
    int rc;
    if (g.bin_cap < cap) {
        if (g.bin_keys) (void)hipFree(g.bin_keys);
        if (g.q_sorted) (void)hipFree(g.q_sorted);
        g.bin_keys = nullptr; g.q_sorted = nullptr; g.bin_cap = 0;
        if ((rc = dev_alloc(&g.bin_keys, cap)) || (rc = dev_alloc(&g.q_sorted, cap + 64 * RSPT_BIN_K))) return rc;
        g.bin_cap = cap;
    }
    if (g.n_bin_info < n_iters) {
        if (g.bin_info) (void)hipFree(g.bin_info);
        g.bin_info = nullptr; g.n_bin_info = 0;
        if ((rc = dev_alloc(&g.bin_info, n_iters))) return rc;
        g.n_bin_info = n_iters;
    }
    return RSPT_OK;
}

int ensure_hit_inst(size_t n) {
    if (g.hit_inst_cap >= n) return RSPT_OK;
    if (g.hit_inst) (void)hipFree(g.hit_inst);
    g.hit_inst = nullptr; g.hit_inst_cap = 0;
    int rc = dev_alloc(&g.hit_inst, n);
    if (rc) return rc;
    g.hit_inst_cap = n;
    return RSPT_OK;
}

int ensure_overflow_list(size_t n) {
    if (g.ovf_cap >= n) return RSPT_OK;
    if (g.ovf) (void)hipFree(g.ovf);
    g.ovf = nullptr; g.ovf_cap = 0;
    int rc = dev_alloc(&g.ovf, n);
    if (rc) return rc;
    g.ovf_cap = n;
    return RSPT_OK;
}

int ensure_spill(size_t threads) {
    if (g.spill_threads >= threads) return RSPT_OK;
    if (g.spill) (void)hipFree(g.spill);
    g.spill = nullptr; g.spill_threads = 0;
    int rc = dev_alloc(&g.spill, 2 * threads * RSPT_W4_SPILL);  // one set of rows per trace lane
    if (rc) return rc;
    g.spill_threads = threads;
    return RSPT_OK;
}
// persistent trace grid = what is resident: k_trace_w4 fits five 256-thread workgroups per CU (93 VGPRs, 30 KB LDS); more only
// adds workgroups that start at the tail, copy the root-side records and find the queue empty (8 -> 5: C3 +0.6 %)
uint32_t pw_grid() { return grid_for((uint32_t)env_size("RSPT_PW_BLOCKS_PER_CU", 5)); }
// threads of the widest persistent trace launch any selectable shape can make (spill rows are addressed [row][thread]): the default shape's pw_grid() x 256, or the
// 1024 threads per CU of RSPT_W4_SHAPE = 1 / 2 — whichever is larger (with RSPT_PW_BLOCKS_PER_CU < 4 the big shapes indexed past the rows sized for the default)
size_t pw_spill_threads() { return std::max<size_t>((size_t)pw_grid() * RSPT_PW_BLOCK, (size_t)grid_for(1) * 1024u); }

int ensure_counts(uint32_t n) {
    if (g.n_cnt >= n) return RSPT_OK;
    if (g.cnt) (void)hipFree(g.cnt);
    g.cnt = nullptr;
    int rc = dev_alloc(&g.cnt, n);
    if (rc) return rc;
    g.n_cnt = n;
    return RSPT_OK;
}

hipEvent_t get_event(size_t i) {
    while (g.events.size() <= i) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        g.events.push_back(e);
    }
    return g.events[i];
}

__global__ void k_accum_counts(const QueueCounts* cnt, uint32_t n_iter, unsigned long long* totals) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    unsigned long long c = 0, a = 0;
    for (uint32_t i = 0; i < n_iter; i++) { c += cnt[i].closest; a += cnt[i].any; }
    totals[3] += c;
    totals[4] += a;
}

// blockqueue/mod.rs:100-115
uint32_t part1by1(uint32_t x) {
    x &= 0x0000ffff; x = (x ^ (x << 8)) & 0x00ff00ff; x = (x ^ (x << 4)) & 0x0f0f0f0f;
    x = (x ^ (x << 2)) & 0x33333333; return (x ^ (x << 1)) & 0x55555555;
}
uint32_t morton2(uint32_t x, uint32_t y) { return (part1by1(y) << 1) + part1by1(x); }

int round_up_pow2_32(int32_t v) {  // pbrt.rs:188-198
    v -= 1; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16;
    return v + 1;
}

// Light distribution for (scene, strategy): create_light_sample_distribution (lightdistrib.rs:393-418)
int get_light_dist(rspt_scene_s* s, uint32_t strategy, LightDistDev* out, const LightDist** lazy_out = nullptr) {
    const uint32_t nl = s->dev.n_lights;
    *out = LightDistDev{};
    out->nvox[0] = out->nvox[1] = out->nvox[2] = 1;
    if (nl == 0) return RSPT_OK;
    int eff = (strategy == RSPT_LIGHTS_UNIFORM || nl == 1) ? RSPT_LIGHTS_UNIFORM : (strategy == RSPT_LIGHTS_POWER ? RSPT_LIGHTS_POWER : RSPT_LIGHTS_SPATIAL);
    auto it = s->light_dists.find(eff);
    if (it == s->light_dists.end()) {
        LightDist d;
        uint64_t n_vox = 1;
        uint64_t n_rows = 1;
        bool lazy = false;
        if (eff == RSPT_LIGHTS_SPATIAL) {  // SpatialLightDistribution::new (lightdistrib.rs:127-166)
            float diag[3] = {s->dev.wb_max[0] - s->dev.wb_min[0], s->dev.wb_max[1] - s->dev.wb_min[1], s->dev.wb_max[2] - s->dev.wb_min[2]};
            int me = (diag[0] > diag[1] && diag[0] > diag[2]) ? 0 : (diag[1] > diag[2] ? 1 : 2);  // maximum_extent
            float bmax = diag[me];
            for (int i = 0; i < 3; i++) {
                float r = roundf(diag[i] / bmax * 64.0f);
                int32_t v = (r != r) ? 0 : (r >= 2147483648.0f ? 2147483647 : (r <= -2147483648.0f ? (-2147483647 - 1) : (int32_t)r));
                d.nvox[i] = std::max(1, v);
            }
            n_vox = (uint64_t)d.nvox[0] * d.nvox[1] * d.nvox[2];
            d.spatial = 1;
            // Every emissive triangle is a light, so the full table is n_vox x n_lights (64^3 voxels x 10^4 lights = 21 GB and 3 x 10^11
            // light samples before the first pixel).  The reference fills a voxel the first time a path looks it up
            // (lightdistrib.rs:297-384); above RSPT_LIGHT_TABLE_EAGER_BYTES the same is done here, with rows handed out from a pool.
            const uint64_t row_bytes = (2ull * nl + 2ull) * sizeof(float);
            lazy = n_vox * row_bytes > env_size("RSPT_LIGHT_TABLE_EAGER_BYTES", (size_t)1 << 30);
            n_rows = n_vox;
            if (lazy) n_rows = std::max<uint64_t>(1, std::min<uint64_t>(n_vox, env_size("RSPT_LIGHT_TABLE_POOL_BYTES", (size_t)16 << 30) / row_bytes));
            if (n_rows > 0x7fffffffull || n_vox > 0x7fffffffull) return fail(RSPT_E_UNSUPPORTED, "spatial light distribution: %llu voxels", (unsigned long long)n_vox);
        }
        int rc;
        // every allocation is owned by the scene as soon as it exists (a later failure must not leak the earlier ones)
        auto owned = [&](auto** p, size_t n) { int r = dev_alloc(p, n); if (!r && *p) s->allocs.push_back(*p); return r; };
        if ((rc = owned(&d.func, n_rows * nl)) || (rc = owned(&d.cdf, n_rows * (nl + 1))) || (rc = owned(&d.func_int, n_rows))) return rc;
        if (lazy) {
            if ((rc = owned(&d.table, n_vox)) || (rc = owned(&d.new_list, n_rows)) || (rc = owned(&d.lazy, 1))) return rc;
            d.max_rows = (uint32_t)n_rows;
            HIP_TRY(hipMemsetAsync(d.table, 0xff, n_vox * sizeof(int32_t), g.stream));  // -1: no distribution yet
            LightLazy lz{0u, 0u, d.max_rows, 0u};
            HIP_TRY(hipMemcpyAsync(d.lazy, &lz, sizeof lz, hipMemcpyHostToDevice, g.stream));
        } else if (eff == RSPT_LIGHTS_SPATIAL) {
            uint64_t total = n_vox * nl;
            hipLaunchKernelGGL(k_ld_contrib, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, g.stream, s->dev, d.nvox[0], d.nvox[1], d.nvox[2], d.func);
            hipLaunchKernelGGL(k_ld_build, dim3((uint32_t)((n_vox + 255) / 256)), dim3(256), 0, g.stream, (uint32_t)n_vox, nl, 0, d.func, d.cdf, d.func_int);
        } else {
            hipLaunchKernelGGL(k_ld_fixed, dim3((nl + 255) / 256), dim3(256), 0, g.stream, s->dev, eff == RSPT_LIGHTS_POWER ? 1 : 0, d.func);
            hipLaunchKernelGGL(k_ld_build, dim3(1), dim3(64), 0, g.stream, 1u, nl, 1, d.func, d.cdf, d.func_int);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(g.stream));
        it = s->light_dists.emplace(eff, d).first;
    }
    const LightDist& d = it->second;
    out->func = d.func; out->cdf = d.cdf; out->func_int = d.func_int;
    out->nvox[0] = d.nvox[0]; out->nvox[1] = d.nvox[1]; out->nvox[2] = d.nvox[2];
    out->spatial = d.spatial;
    out->table = d.table;
    if (lazy_out) *lazy_out = d.table ? &d : nullptr;
    return RSPT_OK;
}

bool trace_can_overflow(const rspt_scene_s* s);
// upper bound of the queue lengths of the launches that follow, when the host knows one (the null-surface tail of a render looks at
// its queue every 8th iteration and the queues only shrink): a few hundred paths do not need 1280 persistent workgroups each copying
// the root-side records into LDS
uint32_t g_queue_hint = 0xffffffffu;
uint32_t* g_inst_out = nullptr;  // where the next closest-hit launches record the instance of each hit instead of g.hit_inst (volpath's shadow-ray segments)
uint32_t hinted_grid(uint32_t full, uint32_t per_block) {
    if (g_queue_hint == 0xffffffffu) return full;
    const uint64_t need = (2ull * g_queue_hint + per_block - 1) / per_block + 1;
    return (uint32_t)std::min<uint64_t>(full, need);
}

// kernel choice: the persistent-wave kernel (trace_wide.h) unless RSPT_TRACE_KERNEL=0 or the
// reference-order node / triangle counters are wanted (only k_trace counts them)
// lane 0 = the library's main stream; lane 1 = the second stream with its own overflow list and spill rows
// One launch of the traversal stage.  Kernel choice: scenes with object instances or alpha-masked meshes take the <INST, ALPHA>
// instantiations (template flags, so that the plain kernels stay the ones measured in DESIGN.md); counters and RSPT_TRACE_KERNEL=0
// use the reference-order loop; a scene whose records outgrow the four-box reference fields stays on the two-box kernel.
#ifndef RSPT_PW_CHUNK_CAMERA_DEFAULT
#define RSPT_PW_CHUNK_CAMERA_DEFAULT 1024   // measured on the C3 stand-in, same box, alternating: 256 -> 2025 / 2029, 1024 -> 2075 / 2070, 4096 -> 2024 / 2030, 16384 -> 1900 Msamples/s
                                            // (C2: 473.9 / 473.3 / 471.1 at 256 / 1024 / 4096); the incoherent launches keep 256 (512: C3 2061 with the camera launch at 1024)
#endif
#ifndef RSPT_PW_REFILL_CAMERA_DEFAULT
#define RSPT_PW_REFILL_CAMERA_DEFAULT 48   // a wave of coherent camera rays refills when three quarters of its lanes are idle (the incoherent launches: RSPT_PW_REFILL = 16)
#endif
#ifndef RSPT_PW_ENTER
#define RSPT_PW_ENTER 24   // C5 stand-in, every instance moving (profiles/r06_c5_enter_sweep.txt): 1 -> 142, 8 -> 170, 16 -> 182, 24 -> 183, 32 -> 180 Msamples/s
#endif
bool g_camera_launch = false;
int g_any_q_force = -1;   // render_impl's measurement of the two shadow-ray kernels (tune_any): 0 / 1 forces the plain / the quantised one for the launches in between
template <bool ANY, int OUT_MODE, bool INST, bool ALPHA>
void launch_trace_v(int lane, bool count, uint32_t grid, const rspt_scene_s* s, const uint32_t* queue, const uint32_t* count_ptr, uint32_t count_imm, uint32_t* cursor,
                    const rspt_ray* ra, const rspt_ray* rb, float4* oa, float4* ob, uint32_t* occ, rspt_hit* hits, unsigned long long* counters, uint32_t* xcur) {
    const SceneDev& sc = s->dev;
    // RSPT_TRACE_KERNEL: 0 = k_trace (reference-order single-ray loop), 1 = k_trace_pw (persistent waves, two boxes
    // per record), 2 = k_trace_w4 (persistent waves, four boxes per record; default).
    // (A quad-per-ray variant with one coalesced 64-byte fetch per step was measured 35 % slower: the
    //  replicated control flow made it VALU-bound with 16 rays per wave)
    const size_t which = env_size("RSPT_TRACE_KERNEL", 2);
    hipStream_t stream = lane ? g.stream2 : g.stream;
    uint32_t* ovf = g.ovf + (lane ? 2 * g.ovf_cap / 3 : 0);
    uint2* spill = g.spill + (lane ? g.spill_threads * RSPT_W4_SPILL : 0);
    uint32_t* hi = (INST && OUT_MODE == 0 && !ANY) ? (g_inst_out ? g_inst_out : g.hit_inst) : nullptr;
    const bool special = INST || ALPHA;
    // moving instances: k_trace_w4<.., INST, 0, ANIM> (round 5; RSPT_ANIM_W4=0: the reference-order loop with the interpolation, as before)
    const bool anim_w4 = s->has_animated && s->w4_ok && which >= 2 && env_size("RSPT_ANIM_W4", 1) != 0 && env_size("RSPT_INSTANCE_KERNEL", 1) != 0;   // (round 6: the reference-order loop serves moving instances next to masks too, so every A/B switch stays bit-exact)
    const bool slow = count || which == 0 || (s->has_animated && !anim_w4) || (special && (!s->w4_ok || env_size("RSPT_INSTANCE_KERNEL", 1) == 0));
    if (slow) {
        if (INST && s->has_animated)   // moving instances (alone or next to alpha-masked meshes): the reference-order loop with the interpolation (its own instantiations; no node / triangle counters)
            hipLaunchKernelGGL((k_trace<ANY, OUT_MODE, false, true, ALPHA, true>), dim3(grid), dim3(RSPT_TRACE_BLOCK), 0, stream, sc, s->tex, queue, count_ptr, count_imm, ra, rb, oa, ob, occ, hits, counters, hi);
        else if (count)
            hipLaunchKernelGGL((k_trace<ANY, OUT_MODE, true, INST, ALPHA>), dim3(grid), dim3(RSPT_TRACE_BLOCK), 0, stream, sc, s->tex, queue, count_ptr, count_imm, ra, rb, oa, ob, occ, hits, counters, hi);
        else
            hipLaunchKernelGGL((k_trace<ANY, OUT_MODE, false, INST, ALPHA>), dim3(grid), dim3(RSPT_TRACE_BLOCK), 0, stream, sc, s->tex, queue, count_ptr, count_imm, ra, rb, oa, ob, occ, hits, counters, hi);
        return;
    }
    const uint32_t pgrid = hinted_grid(pw_grid(), RSPT_PW_BLOCK);
    // rays a wave claims per global atomic: 256 for the incoherent launches; the camera-ray launch of a batch (pixel-major queue: a chunk is a run of samples of one pixel
    // or its neighbours) takes RSPT_PW_CHUNK_CAMERA (g_camera_launch is set around that launch by the path integrator's loop)
    // refill / leaf-phase thresholds; the camera-ray launch may take its own (RSPT_PW_REFILL_CAMERA / RSPT_PW_LEAF_CAMERA: coherent rays reach their leaves together)
    // (camera launch, C3 stand-in, one box, alternating: refill 16 -> 2079 / 2067 Msamples/s, 32 -> 2083 / 2083, 48 -> 2114 / 2113, 64 -> 2105 / 2101; leaf 16 / 24 / 32 at refill 16: 2061 / 2059 / 2046)
    const int pw_refill = (int)(g_camera_launch ? env_size("RSPT_PW_REFILL_CAMERA", RSPT_PW_REFILL_CAMERA_DEFAULT) : env_size("RSPT_PW_REFILL", RSPT_PW_REFILL));
    const int pw_leaf = (int)(g_camera_launch ? env_size("RSPT_PW_LEAF_CAMERA", env_size("RSPT_PW_LEAF", RSPT_PW_LEAF)) : env_size("RSPT_PW_LEAF", RSPT_PW_LEAF));
    const uint32_t pw_chunk = (uint32_t)std::min<size_t>(std::max<size_t>((g_camera_launch ? env_size("RSPT_PW_CHUNK_CAMERA", RSPT_PW_CHUNK_CAMERA_DEFAULT) : env_size("RSPT_PW_CHUNK", RSPT_PW_CHUNK)) & ~(size_t)63, 64), 1u << 20) |
                              (env_size("RSPT_PW_ADAPT", 1) != 0 ? 0u : 1u);   // (bit 0: the kernels do not shrink the claim on short queues — trace_w4.h)
    grid = hinted_grid(grid, RSPT_TRACE_BLOCK);
    uint32_t* n_overflow = cursor + 2;  // QueueCounts layout: overflow word sits two after its cursor
    const uint32_t spill_rows = (uint32_t)std::min<size_t>(env_size("RSPT_W4_SPILL_ROWS", RSPT_W4_SPILL), RSPT_W4_SPILL);
    if constexpr (INST) {
        if (anim_w4) {   // moving instances; next to alpha-masked meshes the masks in line (ALPHA = 2) where every mask allows it, else through alpha_pass
            // RSPT_PW_ENTER: lanes in front of an instance wait until that many of a wave do (trace_w4.h, the entry phase); it rides in bits 8.. of the leaf threshold
            const int pw_enter = (int)std::min<size_t>(std::max<size_t>(env_size("RSPT_PW_ENTER", RSPT_PW_ENTER), 1), 64);
            auto go = [&](auto kern) {
                hipLaunchKernelGGL(kern, dim3(pgrid), dim3(RSPT_PW_BLOCK), 0, stream, sc, s->tex, s->w4, s->big_leaves, s->w4_root, queue, count_ptr, count_imm, cursor,
                                   ra, rb, oa, ob, occ, hits, n_overflow, ovf, spill, spill_rows, pw_refill, (pw_leaf & 0xff) | (pw_enter << 8), s->w4_top, hi, xcur, pw_chunk);
            };
            if constexpr (ALPHA) { if (s->alpha_simple) go(k_trace_w4<ANY, OUT_MODE, true, 2, true>); else go(k_trace_w4<ANY, OUT_MODE, true, 1, true>); }
            else go(k_trace_w4<ANY, OUT_MODE, true, 0, true>);
            hipLaunchKernelGGL((k_trace_fixup<ANY, OUT_MODE, true, ALPHA, true>), dim3(grid), dim3(RSPT_TRACE_BLOCK), 0, stream, sc, s->tex, n_overflow, ovf, ra, rb, oa, ob, occ, hits, hi);
            return;
        }
    }
    if constexpr (ANY && !INST && !ALPHA) {
        // round 6: shadow rays of plain scenes walk the 64-byte quantised records (trace_w4q.h; RSPT_ANY_Q=0: the plain kernel).  Occlusion flags byte-identical.
        // Which of the two is faster depends on the rays, not on the scene's size: the quantised records win where the plain kernel is bound by L1 lane requests (C2's incoherent
        // shadow rays through a dense soup: +6 % on the frame) and lose where it is bound by VALU issue with half its fetches in LDS (the C3 stand-in's coherent ones: -2.5 %;
        // profiles/r06_any_q_ab.txt).  RSPT_ANY_Q=0 / 1 forces one; otherwise the scene's measured choice (rspt_scene_s::any_q_choice), the plain kernel until it exists.
        const char* q_env = getenv("RSPT_ANY_Q");
        const bool use_q = g_any_q_force >= 0 ? g_any_q_force != 0 : (q_env && *q_env ? atoi(q_env) != 0 : s->any_q_choice > 0);
        if (use_q && which >= 2 && s->w4q && !(s->w4_root & RSPT_REF_LEAF) && ra == rb && env_size("RSPT_W4_SHAPE", RSPT_W4_SHAPE_DEFAULT) == 0 && !xcur && spill_rows == RSPT_W4_SPILL) {
            hipLaunchKernelGGL((k_trace_w4q<OUT_MODE>), dim3(pgrid), dim3(RSPT_PW_BLOCK), 0, stream, sc, s->w4q, s->big_leaves, s->w4_root, queue, count_ptr, count_imm, cursor,
                               ra, occ, hits, reinterpret_cast<uint32_t*>(spill), pw_refill, pw_leaf, s->w4_top,
                               pw_chunk | (env_size("RSPT_ANY_Q_LATE", 1) != 0 ? 2u : 0u) /* bit 1: the exact leaf-box test only behind a triangle hit (trace_w4q.h) */, s->leaf_boxes);
            return;
        }
    }
    if constexpr (!INST && !ALPHA) {
        // RSPT_W4_SHAPE: 0 = five 256-thread workgroups per CU, 56 root-side records in LDS each; 1 = ONE 1024-thread workgroup per CU with 512 records;
        // 2 = two 512-thread workgroups with 256 records each (trace_w4.h BLOCK / TOPCAP; dynamic LDS, 152 KB per CU either way)
        const size_t shape = which >= 2 && s->w4_ok ? env_size("RSPT_W4_SHAPE", RSPT_W4_SHAPE_DEFAULT) : 0;
        if (shape == 1 || shape == 2) {
            auto go = [&](auto kern, uint32_t block, uint32_t topcap, uint32_t per_cu) -> bool {
                const size_t lds = (size_t)8 * RSPT_W4_LDS * block + (size_t)112 * topcap;
                static bool attr_set[2][2][2][3] = {};
                bool& done = attr_set[ANY ? 1 : 0][OUT_MODE ? 1 : 0][0][shape];
                static bool attr_bad[2][2][2][3] = {};
                bool& bad = attr_bad[ANY ? 1 : 0][OUT_MODE ? 1 : 0][0][shape];
                if (!done) { bad = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess; (void)hipGetLastError(); done = true; }
                if (bad) return false;   // the device refuses that much dynamic LDS: the default shape serves the launch
                const uint32_t bgrid = hinted_grid(grid_for(per_cu), block);
                hipLaunchKernelGGL(kern, dim3(bgrid), dim3(block), lds, stream, sc, s->tex, s->w4, s->big_leaves, s->w4_root, queue, count_ptr, count_imm, cursor,
                                   ra, rb, oa, ob, occ, hits, n_overflow, ovf, spill, spill_rows, pw_refill, pw_leaf, s->w4_top, hi, xcur, pw_chunk);
                return true;
            };
            const bool launched = shape == 1 ? go(k_trace_w4<ANY, OUT_MODE, false, 0, false, 1024, 512>, 1024u, 512u, 1u) : go(k_trace_w4<ANY, OUT_MODE, false, 0, false, 512, 256>, 512u, 256u, 2u);
            if (launched) {
                if (trace_can_overflow(s))
                    hipLaunchKernelGGL((k_trace_fixup<ANY, OUT_MODE, INST, ALPHA>), dim3(grid), dim3(RSPT_TRACE_BLOCK), 0, stream, sc, s->tex, n_overflow, ovf, ra, rb, oa, ob, occ, hits, hi);
                return;
            }
        }
    }
    if (ALPHA && s->alpha_simple)   // every mask of the scene is evaluated in line (kernels.h alpha_simple): the traversal keeps its register budget
        hipLaunchKernelGGL((k_trace_w4<ANY, OUT_MODE, INST, ALPHA ? 2 : 0>), dim3(pgrid), dim3(RSPT_PW_BLOCK), 0, stream, sc, s->tex, s->w4, s->big_leaves, s->w4_root, queue, count_ptr, count_imm, cursor,
                           ra, rb, oa, ob, occ, hits, n_overflow, ovf, spill, spill_rows, pw_refill, pw_leaf, s->w4_top, hi, xcur, pw_chunk);
    else if (special || (which >= 2 && s->w4_ok))
        hipLaunchKernelGGL((k_trace_w4<ANY, OUT_MODE, INST, ALPHA ? 1 : 0>), dim3(pgrid), dim3(RSPT_PW_BLOCK), 0, stream, sc, s->tex, s->w4, s->big_leaves, s->w4_root, queue, count_ptr, count_imm, cursor,
                           ra, rb, oa, ob, occ, hits, n_overflow, ovf, spill, spill_rows, pw_refill, pw_leaf, s->w4_top, hi, xcur, pw_chunk);
    else
        hipLaunchKernelGGL((k_trace_pw<ANY, OUT_MODE>), dim3(pgrid), dim3(RSPT_PW_BLOCK), 0, stream, sc, s->pairs, queue, count_ptr, count_imm, cursor, ra, rb, oa, ob, occ, hits, n_overflow, ovf,
                           pw_refill, pw_leaf);
    // with every spill row in use the plain four-box kernel cannot overflow (RSPT_W4_MAX_STACK): no second pass to launch
    if (trace_can_overflow(s))
        hipLaunchKernelGGL((k_trace_fixup<ANY, OUT_MODE, INST, ALPHA>), dim3(grid), dim3(RSPT_TRACE_BLOCK), 0, stream, sc, s->tex, n_overflow, ovf, ra, rb, oa, ob, occ, hits, hi);
}
template <bool ANY, int OUT_MODE>
void launch_trace(int lane, bool count, uint32_t grid, const rspt_scene_s* s, const uint32_t* queue, const uint32_t* count_ptr, uint32_t count_imm, uint32_t* cursor,
                  const rspt_ray* ra, const rspt_ray* rb, float4* oa, float4* ob, uint32_t* occ, rspt_hit* hits, unsigned long long* counters, uint32_t* xcd_cursors = nullptr) {
    // XCD-affine dealing (trace_w4.h), RSPT_XCD_DEAL=1: available where the caller hands eight zeroed cursor words (the path integrator's loop, the trace hook).
    // OFF by default: measured neutral to slightly negative (C2 474.0 -> 471.5 Msamples/s, C3 stand-in 2015.6 -> 2014.7; L2 hit rate of the closest-hit
    // launches 0.624 -> 0.620 by TCC_HIT / TCC_MISS — profiles/r05_xcd_affine_ab.txt): the tree's L2 hits are its root side, which every XCD holds anyway
    uint32_t* xcur = (xcd_cursors && env_size("RSPT_XCD_DEAL", 0) != 0) ? xcd_cursors : nullptr;
#define RSPT_LT(I, A) launch_trace_v<ANY, OUT_MODE, I, A>(lane, count, grid, s, queue, count_ptr, count_imm, cursor, ra, rb, oa, ob, occ, hits, counters, xcur)
    if (s->has_instances) { if (s->has_alpha) RSPT_LT(true, true); else RSPT_LT(true, false); }
    else { if (s->has_alpha) RSPT_LT(false, true); else RSPT_LT(false, false); }
#undef RSPT_LT
}

// does the production trace kernel ever hand rays to k_trace_fixup?  Not the four-box kernel with all its spill rows (RSPT_W4_MAX_STACK)
bool trace_can_overflow(const rspt_scene_s* s) {
    const size_t which = env_size("RSPT_TRACE_KERNEL", 2);
    const size_t rows = std::min<size_t>(env_size("RSPT_W4_SPILL_ROWS", RSPT_W4_SPILL), RSPT_W4_SPILL);
    return s->has_instances || (s->has_alpha && !s->w4_ok) || !(which >= 2 && s->w4_ok) || RSPT_W4_LDS + rows < RSPT_W4_MAX_STACK;  // (two stacked aggregates + leaf continuations can pass the bound)
}
uint32_t trace_grid() { return grid_for((uint32_t)env_size("RSPT_TRACE_BLOCKS_PER_CU", 5)); }

// ---- the shade stage's instantiations (kernels.h k_shade<F>) ----
// A scene is served by the narrowest compiled feature set that covers what it can put in front of the stage (rspt_scene_s.shade_features
// + the sampler): the code for every other lobe type, light kind, texture slot, instance transform and the Halton sampler folds away,
// and with it registers (generic: 212 VGPRs = 2 waves / SIMD).  The arithmetic that remains is the same, so results do not change.
// (the feature sets SV_DIFFUSE / SV_PLASTIC / SV_TEXTURED / SV_GENERIC: tu_decl.h, next to the instantiations they name)
typedef void (*ShadeKernel)(RSPT_SHADE_ARGS);
// natural = the compiler's own register budget; w3 / w4 = built for 3 / 4 waves per SIMD (amdgpu_waves_per_eu: what does not fit 168 / 128
// VGPRs is spilled); dflt = which of the three runs.  Measured on one box (profiles/r03_ab_shade.md; Msamples/s of C2 / the C3 stand-in,
// k_shade seconds per step): generic 212 VGPRs 423 / 1685 (0.161 / 0.578 s); diffuse 158 VGPRs = 3 waves as compiled 459 (0.111 s), forced to
// 4 waves 455; plastic 173 VGPRs as compiled 1749 (0.531 s), 168 + 24 B of spills = 3 waves 1841 (0.470 s), 128 + 152 B = 4 waves 1791.
struct ShadeVariant { uint32_t features; const char* name; ShadeKernel natural, w3, w4; int dflt; ShadeKernel move; /* the MOVE form (kernels.h PathBuf::move), built as this set's default is; nullptr: none */ };
const ShadeVariant g_shade_variants[] = {
    {SV_DIFFUSE, "diffuse", k_shade<SV_DIFFUSE>, k_shade_w<SV_DIFFUSE, 3>, k_shade_w<SV_DIFFUSE, 4>, 3, k_shade_mw<SV_DIFFUSE, 3>},   // (round 4: as compiled it now takes 169 VGPRs = 2 waves — the in-kernel voxel claim of light_row_try
                                                                                                           //  cost the four registers; the 3-wave build fits 168 without scratch: Cornell 875 -> see profiles/r04_*)
    {SV_PLASTIC, "plastic", k_shade<SV_PLASTIC>, k_shade_w<SV_PLASTIC, 3>, k_shade_w<SV_PLASTIC, 4>, 3, k_shade_mw<SV_PLASTIC, 3>},
    {SV_TEXTURED, "textured", k_shade<SV_TEXTURED>, k_shade_w<SV_TEXTURED, 3>, k_shade_w<SV_TEXTURED, 4>, 0, k_shade_m<SV_TEXTURED>},
    {SV_DIFFUSE_H, "diffuse-halton", k_shade<SV_DIFFUSE_H>, k_shade_w<SV_DIFFUSE_H, 3>, k_shade_w<SV_DIFFUSE_H, 3>, 3, k_shade_mw<SV_DIFFUSE_H, 3>},   // (tu_decl.h: the reference's default sampler gets the narrow builds too)
    {SV_PLASTIC_H, "plastic-halton", k_shade<SV_PLASTIC_H>, k_shade_w<SV_PLASTIC_H, 3>, k_shade_w<SV_PLASTIC_H, 3>, 3, k_shade_mw<SV_PLASTIC_H, 3>},
    {SV_TEXTURED_H, "textured-halton", k_shade<SV_TEXTURED_H>, k_shade_w<SV_TEXTURED_H, 3>, k_shade_w<SV_TEXTURED_H, 3>, 3, k_shade_mw<SV_TEXTURED_H, 3>},   // (textured C3 stand-in: 1343 as compiled, 1358 at 3 waves)
    {SV_GENERIC, "generic", k_shade<SV_GENERIC>, k_shade_w<SV_GENERIC, 3>, k_shade_w<SV_GENERIC, 4>, 0, k_shade_m<SV_GENERIC>},
    {SV_DYNAMIC, "dynamic", k_shade<SV_DYNAMIC>, k_shade<SV_DYNAMIC>, k_shade<SV_DYNAMIC>, 0, nullptr},   // (its MOVE form needs 256 VGPRs = one wave per SIMD: dynamic materials keep slots for life)   // + lobe lists built per hit (material_assembly.h)
    {SF_ALL, "moving", k_shade<SF_ALL>, k_shade<SF_ALL>, k_shade<SF_ALL>, 0, nullptr},                     // + moving object instances (dev_scene.h inst_at)
};
// RSPT_SHADE_VARIANT = name forces an instantiation (it must cover the scene), RSPT_SHADE_WAVES = 0 | 3 | 4 one of its builds (A/B)
// move_out (may be null): the MOVE form of the chosen set when it has one and the build asked for is its default (RSPT_SHADE_WAVES A/B runs stay on the slot-for-life kernels)
ShadeKernel shade_kernel_for(uint32_t need, const char** name_out, ShadeKernel* move_out = nullptr) {
    const char* force = getenv("RSPT_SHADE_VARIANT");
    if (move_out) *move_out = nullptr;
    for (const ShadeVariant& v : g_shade_variants) {
        if ((need & ~v.features) != 0) continue;
        if (force && *force && strcmp(force, v.name) != 0 && (v.features | SF_DYNAMIC | SF_ANIM) != SF_ALL) continue;
        if (name_out) *name_out = v.name;
        const size_t waves = env_size("RSPT_SHADE_WAVES", (size_t)v.dflt);
        if (move_out && waves == (size_t)v.dflt) *move_out = v.move;
        return waves == 3 ? v.w3 : (waves == 4 ? v.w4 : v.natural);
    }
    return k_shade<SF_ALL>;
}

#ifndef RSPT_DL_WAVES_DEFAULT
#define RSPT_DL_WAVES_DEFAULT 0
#endif
constexpr int RSPT_DL_RETRY_LANE = -1000;   // batch_direct -> the batch loop: redo this batch with the per-lane form (never leaves render_impl)
int render_impl(rspt_scene_s* s, const rspt_render_desc* d, float* film_host, void* film_dev, float* li_host, rspt_stats* stats) {
    if (!g.inited) return fail(RSPT_E_NODEVICE, "rspt_init has not been called");
    if (!s || !d) return fail(RSPT_E_INVALID, "null scene or render desc");
    s->select_materials(d->integrator != RSPT_INTEGRATOR_DIRECT);  // allow_multiple_lobes: false in DirectLightingIntegrator::li only (directlighting.rs:86)
    const bool halton = d->sampler_kind == RSPT_SAMPLER_HALTON;
    const bool sobol = d->sampler_kind == RSPT_SAMPLER_SOBOL;
    const bool pixel_sampler = d->sampler_kind >= RSPT_SAMPLER_RANDOM && d->sampler_kind <= RSPT_SAMPLER_MAXMINDIST;  // one serial chain per tile: tile_serial.h
    if (!sobol && !halton && !pixel_sampler) return fail(RSPT_E_UNSUPPORTED, "sampler kind %u", d->sampler_kind);
    if (d->spp <= 0 || d->spp > (1ll << 30) || (sobol && (d->spp & (d->spp - 1)) != 0)) return fail(RSPT_E_INVALID, "spp must be in [1, 2^30] (a power of two for sobol)");
    if (d->tile_size == 0 || d->tile_size > 4096) return fail(RSPT_E_INVALID, "bad tile_size");
    if (sobol && (!d->tables.sobol32 || !d->tables.vdc || !d->tables.vdc_inv)) return fail(RSPT_E_INVALID, "null sobol tables");
    if (pixel_sampler) {
        if (d->integrator == RSPT_INTEGRATOR_DIRECT && d->direct_strategy == RSPT_DIRECT_SAMPLE_ALL && (d->sampler_kind == RSPT_SAMPLER_ZEROTWO || d->sampler_kind == RSPT_SAMPLER_MAXMINDIST))
            for (uint32_t i = 0; s && d->n_light_samples && i < s->dev.n_lights; i++)
                if (d->n_light_samples[i] < 1 || (d->n_light_samples[i] & (d->n_light_samples[i] - 1)) != 0)
                    return fail(RSPT_E_INVALID, "directlighting: n_light_samples[%u] must be a power of two with the 02sequence / maxmindist samplers (preprocess passes it through round_count, directlighting.rs:58-60)", i);
        if (d->integrator == RSPT_INTEGRATOR_AO && (d->sampler_kind == RSPT_SAMPLER_ZEROTWO || d->sampler_kind == RSPT_SAMPLER_MAXMINDIST) && (d->ao_n_samples & (d->ao_n_samples - 1)) != 0)
            return fail(RSPT_E_INVALID, "ao: nsamples must be a power of two with the 02sequence / maxmindist samplers (request_2d_array asserts round_count(n) == n, zerotwosequence.rs:187-193)");
        if (d->tile_size > 255) return fail(RSPT_E_UNSUPPORTED, "tile_size > 255 with a pixel sampler");
        if (d->spp > 65536 || d->pixel_dimensions > 64) return fail(RSPT_E_UNSUPPORTED, "pixel sampler: spp > 65536 or more than 64 sampled dimensions");
        if (d->sampler_kind == RSPT_SAMPLER_STRATIFIED && (d->strat_x == 0 || d->strat_y == 0 || (int64_t)d->strat_x * d->strat_y != d->spp))
            return fail(RSPT_E_INVALID, "stratified sampler: spp must equal strat_x * strat_y");
        if (d->sampler_kind == RSPT_SAMPLER_MAXMINDIST && (!d->maxmin_c_pixel || (d->spp & (d->spp - 1)) != 0)) return fail(RSPT_E_INVALID, "maxmindist sampler: spp must be a power of two and maxmin_c_pixel given");
    }
    if (halton && !d->tables.halton_perms) return fail(RSPT_E_INVALID, "null halton permutation table");
    if (!(d->filter_radius[0] > 0.0f) || !(d->filter_radius[1] > 0.0f)) return fail(RSPT_E_INVALID, "bad filter radius");
    if (d->max_depth > 200) return fail(RSPT_E_UNSUPPORTED, "max_depth > 200");
    if (d->integrator != RSPT_INTEGRATOR_PATH && d->integrator != RSPT_INTEGRATOR_AO && d->integrator != RSPT_INTEGRATOR_DIRECT && d->integrator != RSPT_INTEGRATOR_VOLPATH)
        return fail(RSPT_E_UNSUPPORTED, "integrator %u (path, ao, directlighting and volpath only)", d->integrator);
    const bool direct = d->integrator == RSPT_INTEGRATOR_DIRECT;
    const bool volpath = d->integrator == RSPT_INTEGRATOR_VOLPATH;
    // moving instances under the pixel samplers and in the per-lane directlighting: round 6 (k_tile_serial modes 5 - 8, k_lane_dl<.., ANIM>); dynamic materials next to them under a
    // pixel sampler are the one combination left without an instantiation
    if (s->has_animated && pixel_sampler && s->has_dynamic)
        return fail(RSPT_E_UNSUPPORTED, "a scene with a moving object instance AND a dynamic material (a lobe list that depends on a texture value) under a pixel sampler");
    if (direct) {
        if (d->max_depth < 1 || d->max_depth > (uint32_t)RSPT_DL_SERIAL_DEPTH)
            return fail(RSPT_E_UNSUPPORTED, "directlighting: max_depth must be in [1, %d] (the explicit recursion stack of the per-lane form, dl_serial.h)", RSPT_DL_SERIAL_DEPTH);
        if (d->direct_strategy > RSPT_DIRECT_SAMPLE_ONE) return fail(RSPT_E_INVALID, "bad direct_strategy");
        for (uint32_t i = 0; s && d->n_light_samples && i < s->dev.n_lights; i++)
            if (d->n_light_samples[i] < 1 || d->n_light_samples[i] > 4096) return fail(RSPT_E_INVALID, "n_light_samples[%u] out of range", i);
    }
    if (d->integrator == RSPT_INTEGRATOR_AO && (d->ao_n_samples == 0 || d->ao_n_samples > 4096)) return fail(RSPT_E_INVALID, "ao_n_samples must be in [1, 4096]");
    // checkpoint / resume: the pixel samples [smp_begin, smp_end) of every pixel
    const uint64_t smp_begin = d->sample_begin, smp_end = d->sample_count ? d->sample_begin + d->sample_count : (uint64_t)d->spp;
    if (smp_begin >= smp_end || smp_end > (uint64_t)d->spp) return fail(RSPT_E_INVALID, "sample range [%llu, %llu) outside [0, spp)", (unsigned long long)smp_begin, (unsigned long long)smp_end);
    if (pixel_sampler && (smp_begin != 0 || smp_end != (uint64_t)d->spp)) return fail(RSPT_E_UNSUPPORTED, "a pixel sampler renders all samples of a tile in one chain: partial sample ranges are not possible");
    const int32_t* sb = d->sample_bounds;
    const int32_t* cp = d->crop_px;
    if (sb[2] <= sb[0] || sb[3] <= sb[1] || cp[2] <= cp[0] || cp[3] <= cp[1]) return fail(RSPT_E_INVALID, "empty sample or crop bounds");
    if (sb[0] < -32768 || sb[1] < -32768 || sb[2] > 32767 || sb[3] > 32767) return fail(RSPT_E_UNSUPPORTED, "sample bounds outside the 16-bit pixel range");
    const uint32_t shard_count = d->shard_count ? d->shard_count : 1, chunk = d->tile_chunk ? d->tile_chunk : 1;
    if (d->shard_index >= shard_count) return fail(RSPT_E_INVALID, "shard_index >= shard_count");
    // Sobol' needs 5 + 8 dims per bounce < 1024 (sobol.rs:119-124)
    if (5ull + (volpath ? 10ull : 8ull) * (d->max_depth + 2ull) >= 1024ull) return fail(RSPT_E_UNSUPPORTED, "max_depth exceeds the 1024 Sobol' dimensions");

    auto t_start = std::chrono::steady_clock::now();
    HIP_TRY(hipSetDevice(g.device));
    int rc;
    // ---- render constants ----
    RenderDev rd{};
    memcpy(rd.raster_to_camera, d->raster_to_camera, sizeof rd.raster_to_camera);
    memcpy(rd.camera_to_world, d->camera_to_world, sizeof rd.camera_to_world);
    rd.lens_radius = d->lens_radius; rd.focal_distance = d->focal_distance;
    rd.cam_anim = nullptr;
    if (d->camera_animated) {   // AnimatedTransform::new (camera_anim.h); equal key matrices = a camera that does not move
        CamAnim ca;
        if (camanim::camera_keys(d->camera_to_world, d->camera_time[0], d->camera_to_world_end, d->camera_time[1], &ca)) {
            if (!g.cam_anim && (rc = dev_alloc(&g.cam_anim, 1))) return rc;
            HIP_TRY(hipMemcpy(g.cam_anim, &ca, sizeof ca, hipMemcpyHostToDevice));
            rd.cam_anim = g.cam_anim;
        }
    }
    rd.shutter_open = d->shutter_open; rd.shutter_close = d->shutter_close;
    memcpy(rd.sample_bounds, sb, sizeof rd.sample_bounds);
    memcpy(rd.crop_px, cp, sizeof rd.crop_px);
    rd.resolution = round_up_pow2_32(std::max(sb[2] - sb[0], sb[3] - sb[1]));  // sobol.rs:46-47
    rd.log2_res = 31 - __builtin_clz((uint32_t)rd.resolution);
    rd.spp = d->spp; rd.max_depth = d->max_depth; rd.rr_threshold = d->rr_threshold;
    rd.filter_radius[0] = d->filter_radius[0]; rd.filter_radius[1] = d->filter_radius[1];
    rd.max_sample_luminance = d->max_sample_luminance;
    rd.tile_size = d->tile_size;
    if (!g.sobol32) {
        if ((rc = dev_alloc(&g.sobol32, 1024 * 52)) || (rc = dev_alloc(&g.vdc, 25 * 52)) || (rc = dev_alloc(&g.vdc_inv, 26 * 52)) || (rc = dev_alloc(&g.filter_table, 256))) return rc;
    }
    if (sobol) {
        HIP_TRY(hipMemcpyAsync(g.sobol32, d->tables.sobol32, 1024 * 52 * 4, hipMemcpyHostToDevice, g.stream));
        HIP_TRY(hipMemcpyAsync(g.vdc, d->tables.vdc, 25 * 52 * 8, hipMemcpyHostToDevice, g.stream));
        HIP_TRY(hipMemcpyAsync(g.vdc_inv, d->tables.vdc_inv, 26 * 52 * 8, hipMemcpyHostToDevice, g.stream));
    }
    HIP_TRY(hipMemcpyAsync(g.filter_table, d->filter_table, 256 * 4, hipMemcpyHostToDevice, g.stream));
    rd.sobol32 = g.sobol32; rd.vdc = g.vdc; rd.vdc_inv = g.vdc_inv; rd.filter_table = g.filter_table;
    rd.sampler_kind = d->sampler_kind;
    rd.sample_at_pixel_center = d->sample_at_pixel_center;
    uint32_t vol_dim_limit = 1024u;  // NUM_SOBOL_DIMENSIONS (sobol.rs:119-124) or the Halton permutation table the caller brought
    if (halton) {  // HaltonSampler::new (halton.rs:80-131)
        if (!g.primes) {  // PRIMES / PRIME_SUMS (lowdiscrepancy.rs:31-760)
            std::vector<uint32_t> primes, sums;
            for (uint32_t v = 2; primes.size() < 1000; v++) {
                bool is_p = true;
                for (uint32_t q = 2; q * q <= v; q++) if (v % q == 0) { is_p = false; break; }
                if (is_p) primes.push_back(v);
            }
            uint32_t acc = 0;
            for (uint32_t p : primes) { sums.push_back(acc); acc += p; }
            g.host_primes = primes; g.host_prime_sums = sums;
            if ((rc = dev_alloc(&g.primes, 1000)) || (rc = dev_alloc(&g.prime_sums, 1000))) return rc;
            HIP_TRY(hipMemcpy(g.primes, primes.data(), 4000, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(g.prime_sums, sums.data(), 4000, hipMemcpyHostToDevice));
        }
        uint32_t max_dim = 5u + 8u * (d->max_depth + 2u);  // last dimension a path can consume, with slack
        if (volpath) max_dim = std::min(990u, 5u + 10u * (d->max_depth + 2u) + 64u);  // 10 per counted bounce, 2 per uncounted pass through a medium boundary (slack for 32; vol.h cuts a path that needs more)
        if (direct) {  // every dimension the caller's table covers: k_dl_assign reports a camera sample whose stream really ends beyond it
            max_dim = 0;
            while (max_dim + 1u < 1000u && (uint64_t)g.host_prime_sums[max_dim + 1u] + g.host_primes[max_dim + 1u] <= d->tables.n_halton_perms) max_dim++;
        }
        if (max_dim >= 1000u) return fail(RSPT_E_UNSUPPORTED, "max_depth exceeds the 1000 Halton dimensions");
        const uint64_t need = (uint64_t)g.host_prime_sums[max_dim] + g.host_primes[max_dim];
        if (d->tables.n_halton_perms < need) return fail(RSPT_E_INVALID, "halton permutation table has %llu entries, %llu needed for max_depth %u",
                                                         (unsigned long long)d->tables.n_halton_perms, (unsigned long long)need, d->max_depth);
        if (g.n_halton_perms < need) {
            if (g.halton_perms) (void)hipFree(g.halton_perms);
            g.halton_perms = nullptr; g.n_halton_perms = 0;
            if ((rc = dev_alloc(&g.halton_perms, need))) return rc;
            g.n_halton_perms = need;
        }
        HIP_TRY(hipMemcpyAsync(g.halton_perms, d->tables.halton_perms, need * sizeof(uint16_t), hipMemcpyHostToDevice, g.stream));
        rd.halton_perms = g.halton_perms; rd.primes = g.primes; rd.prime_sums = g.prime_sums;
        vol_dim_limit = max_dim;
        const int32_t res[2] = {sb[2] - sb[0], sb[3] - sb[1]};
        for (int i = 0; i < 2; i++) {
            int32_t base = i == 0 ? 2 : 3, scale = 1, exp = 0;
            while (scale < std::min(res[i], 128)) { scale *= base; exp += 1; }
            rd.base_scales[i] = scale; rd.base_exponents[i] = exp;
        }
        rd.sample_stride = (uint64_t)rd.base_scales[0] * (uint64_t)rd.base_scales[1];
        auto mult_inv = [](int64_t a, int64_t n) {  // multiplicative_inverse via extended_gcd (halton.rs:32-52)
            int64_t x0 = 1, x1 = 0, r0 = a, r1 = n;
            while (r1 != 0) {
                int64_t q = r0 / r1, t = r0 - q * r1;
                r0 = r1; r1 = t;
                t = x0 - q * x1; x0 = x1; x1 = t;
            }
            int64_t m = x0 % n;
            return (uint64_t)(m < 0 ? m + n : m);
        };
        rd.mult_inverse[0] = mult_inv(rd.base_scales[1], rd.base_scales[0]);
        rd.mult_inverse[1] = mult_inv(rd.base_scales[0], rd.base_scales[1]);
    }

    LightDistDev ld;
    const LightDist* ld_lazy = nullptr;  // on-demand voxels: a mark / build round in front of every shade launch
    if ((rc = get_light_dist(s, d->light_strategy, &ld, &ld_lazy))) return rc;
    if (volpath && s->dev.n_grid_media && !pixel_sampler)
        return fail(RSPT_E_UNSUPPORTED, "volpath with a grid-density medium under the Sobol' / Halton sampler: every tracking step draws sampler dimensions (the reference panics past "
                                        "dimension 1024 / 1000 within a bounce or two); render it with a pixel sampler (random / 02sequence / stratified / maxmindist)");
    // on-demand voxels under volpath and the pixel samplers: their kernels meet the lookup points only while they run, so they claim missing
    // voxels themselves (dev_scene.h light_row_try) and the step runs again once the claimed rows are built (lightdistrib.rs:276-384 builds on
    // first touch too; a row is a pure function of (voxel, lights), Q18, so when it is built cannot show)
    if (ld_lazy && (volpath || pixel_sampler)) { ld.lazy = reinterpret_cast<LightLazyWords*>(ld_lazy->lazy); ld.new_list = ld_lazy->new_list; }
    static_assert(sizeof(LightLazyWords) == sizeof(LightLazy), "LightLazyWords mirrors LightLazy");
    // the claimed voxels' rows: contributions of every light, the row's distribution, the table entries; returns how many were claimed
    auto build_claimed_rows = [&](uint32_t* n_claimed) -> int {
        LightLazy lz;
        HIP_TRY(hipMemcpyAsync(g.look, ld_lazy->lazy, sizeof lz, hipMemcpyDeviceToHost, g.stream));
        HIP_TRY(hipStreamSynchronize(g.stream));
        memcpy(&lz, g.look, sizeof lz);
        *n_claimed = lz.n_new;
        if (lz.overflow) return fail(RSPT_E_NOMEM, "spatial light distribution: more than %u voxels were touched; raise RSPT_LIGHT_TABLE_POOL_BYTES", lz.max_rows);
        if (lz.n_new == 0) return RSPT_OK;
        const uint32_t lgrid = grid_for(4);
        hipLaunchKernelGGL(k_ld_contrib_list, dim3(lgrid), dim3(256), 0, g.stream, s->dev, ld.nvox[0], ld.nvox[1], ld.nvox[2], ld_lazy->lazy, ld_lazy->new_list, ld_lazy->func);
        hipLaunchKernelGGL(k_ld_build_list, dim3(lgrid), dim3(256), 0, g.stream, s->dev.n_lights, ld_lazy->lazy, ld_lazy->new_list, ld_lazy->func, ld_lazy->cdf, ld_lazy->func_int, ld_lazy->table);
        hipLaunchKernelGGL(k_ld_commit, dim3(1), dim3(1), 0, g.stream, ld_lazy->lazy);
        return RSPT_OK;
    };

    // ---- this shard's pixels: Morton-ordered tiles (blockqueue/mod.rs:23-52), row-major inside a tile ----
    const int32_t ts = (int32_t)d->tile_size;
    const int32_t ext_x = sb[2] - sb[0], ext_y = sb[3] - sb[1];
    const int32_t ntx = (ext_x + ts - 1) / ts, nty = (ext_y + ts - 1) / ts;
    std::vector<std::pair<uint32_t, uint32_t>> blocks((size_t)ntx * nty);
    for (int32_t i = 0; i < ntx * nty; i++) blocks[i] = {(uint32_t)(i % ntx), (uint32_t)(i / ntx)};
    std::stable_sort(blocks.begin(), blocks.end(), [](const std::pair<uint32_t, uint32_t>& a, const std::pair<uint32_t, uint32_t>& b) { return morton2(a.first, a.second) < morton2(b.first, b.second); });
    std::vector<uint32_t> pix;
    pix.reserve((size_t)ext_x * ext_y / shard_count + 1024);
    for (size_t i = 0; i < blocks.size(); i++) {
        if ((i / chunk) % shard_count != d->shard_index) continue;
        int32_t x0 = sb[0] + (int32_t)blocks[i].first * ts, x1 = std::min(x0 + ts, sb[2]);
        int32_t y0 = sb[1] + (int32_t)blocks[i].second * ts, y1 = std::min(y0 + ts, sb[3]);
        for (int32_t y = y0; y < y1; y++)
            for (int32_t x = x0; x < x1; x++) pix.push_back(((uint32_t)(uint16_t)(int16_t)y << 16) | (uint32_t)(uint16_t)(int16_t)x);
    }
    const size_t n_pix = pix.size();
    if (g.pix_cap < n_pix) {
        if (g.pix_list) (void)hipFree(g.pix_list);
        g.pix_list = nullptr; g.pix_cap = 0;
        if ((rc = dev_alloc(&g.pix_list, n_pix))) return rc;
        g.pix_cap = n_pix;
    }
    if (n_pix) HIP_TRY(hipMemcpyAsync(g.pix_list, pix.data(), n_pix * 4, hipMemcpyHostToDevice, g.stream));

    // ---- film ----
    const size_t film_px = (size_t)(cp[2] - cp[0]) * (size_t)(cp[3] - cp[1]);
    if (g.film_px < film_px) {
        for (float4** p : {&g.film_own, &g.film_splat, &g.film_out}) { if (*p) (void)hipFree(*p); *p = nullptr; }
        g.film_px = 0;
        if ((rc = dev_alloc(&g.film_own, film_px)) || (rc = dev_alloc(&g.film_splat, film_px)) || (rc = dev_alloc(&g.film_out, film_px))) return rc;
        g.film_px = film_px;
    }
    HIP_TRY(hipMemsetAsync(g.film_own, 0, film_px * sizeof(float4), g.stream));
    HIP_TRY(hipMemsetAsync(g.film_splat, 0, film_px * sizeof(float4), g.stream));
    // the film stage: k_film, or for filters wider than a pixel the gather form (kernels.h k_film_gather; RSPT_FILM_GATHER=0: the atomic form)
    const int film_k = (int)std::floor(std::max(d->filter_radius[0], d->filter_radius[1]) + 0.5f);
    const bool film_gather = (d->filter_radius[0] > 0.5f || d->filter_radius[1] > 0.5f) && film_k <= 8 && env_size("RSPT_FILM_GATHER", 1) != 0;
    const size_t sb_px = (size_t)(sb[2] - sb[0]) * (size_t)(sb[3] - sb[1]);
    if (film_gather && g.pix_index_cap < sb_px) {
        if (g.pix_index) (void)hipFree(g.pix_index);
        g.pix_index = nullptr; g.pix_index_cap = 0;
        if ((rc = dev_alloc(&g.pix_index, sb_px))) return rc;
        g.pix_index_cap = sb_px;
    }
    auto film_index = [&](const uint32_t* list, uint32_t n) -> int {   // (the list a batch indexes: the render's, or one pass of the pixel samplers)
        if (!film_gather) return RSPT_OK;
        HIP_TRY(hipMemsetAsync(g.pix_index, 0xff, sb_px * sizeof(uint32_t), g.stream));
        if (n) hipLaunchKernelGGL(k_pix_index, dim3((n + 255) / 256), dim3(256), 0, g.stream, list, n, sb[0], sb[1], sb[2] - sb[0], g.pix_index);
        return RSPT_OK;
    };
    float* li_dev = nullptr;
    auto film_stage = [&](const RenderDev& rd_, const Batch& bt, const PathBuf& fpb, const uint32_t* list) {
        if (!film_gather) {
            hipLaunchKernelGGL(k_film, dim3((bt.n_pix + 255) / 256), dim3(256), 0, g.stream, rd_, bt, fpb, list, g.film_own, (float*)g.film_splat, li_dev, g.totals + 5);
            return;
        }
        const size_t s2 = (size_t)(RSPT_FG_T + 2 * film_k) * (RSPT_FG_T + 2 * film_k);
        const size_t fit = (60000 - s2 * 4 - 1024) / (s2 * 24);   // sample rows of (chunk + 1) x 24 bytes per source pixel in < 64 KB of LDS
        const uint32_t chunk = (uint32_t)std::min<size_t>(std::max<size_t>(fit, 2) - 1, 8);
        const size_t lds = s2 * (chunk + 1) * 24 + s2 * 4 + 1024;
        hipLaunchKernelGGL(k_film_gather, dim3((uint32_t)((sb[2] - sb[0] + RSPT_FG_T - 1) / RSPT_FG_T), (uint32_t)((sb[3] - sb[1] + RSPT_FG_T - 1) / RSPT_FG_T)), dim3(256), lds, g.stream,
                           rd_, bt, fpb, g.pix_index, film_k, chunk, g.film_own, li_dev, g.totals + 5);
    };
    struct LiGuard { float** p; ~LiGuard() { if (*p) (void)hipFree(*p); } } li_guard{&li_dev};  // also on the early error returns below
    if (li_host) {
        HIP_TRY(hipMalloc((void**)&li_dev, film_px * (size_t)d->spp * 3 * sizeof(float)));
        HIP_TRY(hipMemsetAsync(li_dev, 0, film_px * (size_t)d->spp * 3 * sizeof(float), g.stream));
    }

    // ---- batches ----
    // paths per wavefront batch: ~280 B of state each, so 2^29 paths = 150 GB of the 288 GB HBM.  Large batches keep the persistent
    // trace kernel's queues long and its launches few (every launch ends in a tail of a few long rays): measured with round 1's
    // kernels, 2^25 / 2^26 / 2^27 / 2^28 paths per batch = 367 / 386 / 398 / 405 Msamples/s on C2 and 1243 / 1400 / 1472 / 1525 on
    // the C3 stand-in; with round 4's, 2^28 -> 2^29 on the C3 stand-in: 1929 -> 1966 (same box; C2's frame is 2^28 paths, one batch
    // either way).  Halving the number of launch tails bought 1.9 %: what is left of them is what a second pipeline of half batches
    // on other streams could still fill — less than that again, which is why shade / trace overlap across batches was not built
    // (DESIGN.md section 10).  When the device cannot give that much (other tenants), the batch is halved until it fits.
    size_t cap = env_size("RSPT_BATCH", (size_t)1 << 29);
    cap = std::min<size_t>(std::max<size_t>(cap, 1024), (size_t)1 << 30);
    const bool counters = env_size("RSPT_COUNTERS", 0) != 0;
    // AOIntegrator: every camera sample carries ao_n_samples shadow rays through the same ray / occlusion arrays
    const bool ao = d->integrator == RSPT_INTEGRATOR_AO;
    const uint32_t ao_n = ao ? d->ao_n_samples : 1u;
    if (ao) cap = std::max<size_t>(cap / ao_n, 1024);
    // directlighting under Sobol' / Halton: the wavefront form (direct.h) unless the render needs what only the per-lane form has
    // (lane_serial.h): textured materials, more than 8 recursion levels; RSPT_DL_FORM=lane forces it (A/B, tests)
    const char* dl_form_env = getenv("RSPT_DL_FORM");
    // levels of the specular tree that can hold nodes (direct.h DlBuf::levels): a scene without specular lobes has the root only
    const bool dl_specular = s->has_dynamic || (s->shade_features & (RSPT_SF_LOBE(RSPT_BXDF_SPECULAR_R) | RSPT_SF_LOBE(RSPT_BXDF_SPECULAR_T) | RSPT_SF_LOBE(RSPT_BXDF_FRESNEL_SPEC))) != 0;
    // round 6: textured materials WITHOUT specular lobes stay in the wavefront form (every node is a camera hit: k_dl_texture); with specular lobes the children's rays carry
    // reflected / refracted differentials, which only the per-lane form tracks.  RSPT_DL_TEX_WAVEFRONT=0: textures always per lane, as before.  (The estimate kernel of the
    // one-round form is the one that reads the texture results: RSPT_DL_ROUNDS=1 or more than 2^16 estimates per node also send a textured scene to the per-lane form.)
    bool dl_rounds_one = env_size("RSPT_DL_ROUNDS", 0) == 0;
    if (dl_rounds_one && d->direct_strategy == RSPT_DIRECT_SAMPLE_ALL && s->dev.n_lights) {
        uint64_t r = 0;
        for (uint32_t j = 0; j < s->dev.n_lights; j++) r += d->n_light_samples ? (uint64_t)std::max<int32_t>(d->n_light_samples[j], 0) : 1u;
        if (r > (1u << 16)) dl_rounds_one = false;
    }
    const bool dl_tex_wave = s->has_textures && !dl_specular && dl_rounds_one && env_size("RSPT_DL_TEX_WAVEFRONT", 1) != 0;
    bool dl_lane = direct && !pixel_sampler && ((s->has_textures && !dl_tex_wave) || d->max_depth > 8 || (dl_form_env && !strcmp(dl_form_env, "lane")));
    const uint32_t dl_levels = (dl_specular || env_size("RSPT_DL_FULL_TREE", 0) != 0) ? (uint32_t)d->max_depth : std::min<uint32_t>((uint32_t)d->max_depth, 1u);
    const uint32_t dl_H = (direct && !pixel_sampler && !dl_lane) ? (1u << dl_levels) : 1u;   // node slots per camera sample (direct.h; in the per-lane forms a lane walks the tree itself)
    // all light estimates of a node in one round (direct.h k_dl_nee_all): R = sum_j n_j virtual slots per node slot in the ray / result arrays; RSPT_DL_ROUNDS=1 = one
    // round per estimate as before (R = 1 for the sizing)
    uint32_t dl_R = 1;
    // (R used to be CLAMPED to 2^20 while the kernels iterate over the unclamped sum — virtual slots would alias.  A sum above 2^16 — hundreds of lights at thousands
    //  of samples each — takes the round-per-estimate form instead, whose slots do not depend on it; n_slots * R then stays far below the RSPT_Q_MIS bit: the batch is <= 2^28 / R)
    bool dl_one_round = direct && !pixel_sampler && !dl_lane && env_size("RSPT_DL_ROUNDS", 0) == 0;
    if (dl_one_round && d->direct_strategy == RSPT_DIRECT_SAMPLE_ALL && s->dev.n_lights) {
        uint64_t r = 0;
        for (uint32_t j = 0; j < s->dev.n_lights; j++) r += d->n_light_samples ? (uint64_t)std::max<int32_t>(d->n_light_samples[j], 0) : 1u;
        if (r > (1u << 16)) dl_one_round = false;
        else dl_R = (uint32_t)std::max<uint64_t>(r, 1u);
    }
    if (direct) cap = std::max<size_t>(std::min<size_t>(cap, (size_t)1 << (dl_lane ? (s->has_dynamic ? 20 : 22) : (dl_R > 1 ? 28 : 26))) / dl_H / dl_R, 1024);   // (per-lane form: texture rows and lobe records per recursion level)
    if (pixel_sampler) cap = std::max<size_t>(blocks.size(), 1024);   // one path slot per tile (tile_serial.h); the samples' results have their own arrays
    // round 6: the path integrator's MOVING path state (kernels.h PathBuf::move) — wherever the scene's shade instantiation has a MOVE form; not with moving instances
    // (their ray times are kept by original slot) and not under the pixel samplers / the other integrators, which keep slots for life.  RSPT_MOVE=0: slots, as before.
    const char* shade_name = "generic";
    ShadeKernel shade_move_k = nullptr;
    const ShadeKernel shade_slot_k = shade_kernel_for(s->shade_features | (halton ? (uint32_t)SF_HALTON : (uint32_t)SF_SOBOL), &shade_name, &shade_move_k);
    const bool move = !volpath && !direct && !ao && !pixel_sampler && !s->has_animated && shade_move_k != nullptr && env_size("RSPT_MOVE", 1) != 0;
    // the first iteration that runs the MOVE kernel.  Iteration 0 reads a dense pixel-major queue whatever the schedule and its stores are dense too (every slot is
    // written): the slot-for-life kernel serves it (the MOVE form is 4 % slower there: 120 B of scratch against 36 at the same 168 VGPRs, profiles/r06_move_ab.txt),
    // and the first MOVE launch reads what it left in place
    const uint32_t move_first = (uint32_t)env_size("RSPT_MOVE_FROM", 1);
    if (!move) free_move();   // (213 GB at the default batch: the second set is not kept for renders that do not use it)
    uint32_t ns = 1;  // samples per pixel per batch: largest power of two with n_pix * ns <= cap
    size_t pix_per_batch = 1;
    for (;;) {
        ns = 1;
        while ((uint64_t)ns * 2 <= (uint64_t)d->spp && (uint64_t)n_pix * ns * 2 <= cap) ns *= 2;
        pix_per_batch = std::max<size_t>(1, std::min(n_pix, cap / ns));
        rc = ensure_paths(std::max<size_t>(pix_per_batch * ns, 1) * ao_n * dl_H * dl_R);
        if (rc == RSPT_OK && direct && !pixel_sampler && !dl_lane) rc = ensure_direct(g.cap);
        if (rc == RSPT_OK && volpath) rc = ensure_vol(g.cap);
        if (rc == RSPT_OK && move) rc = ensure_move(g.cap);
        if (rc == RSPT_OK) break;
        (void)hipGetLastError();  // out of memory: clear the sticky error and try half the batch
        free_paths();
        if (cap <= ((size_t)1 << 20)) return rc;
        cap /= 2;
    }
    // the per-lane directlighting forms: per-level texture rows, the light sample counts on the device, error / truncation words
    struct TmpGuard { std::vector<void*> p; ~TmpGuard() { for (void* q : p) (void)hipFree(q); } } dl_guard;
    float4* dl_tex = nullptr; int32_t* dl_nls = nullptr; uint32_t* dl_words = nullptr; rspt_mat::Built* dl_dyn = nullptr;
    const uint32_t dl_tex_rows = RSPT_TEX_ROWS + (s->has_dynamic ? RSPT_DYN_ROWS : 0);
    const size_t dl_lanes = pixel_sampler ? blocks.size() : pix_per_batch * ns;
    if (direct) {
        if (s->has_textures && (dl_lane || pixel_sampler)) {   // (the wavefront form over textures keeps its results in g.pb.tex; it cannot fall back to the per-lane form: that needs a specular lobe)
            if ((rc = dev_alloc(&dl_tex, (size_t)d->max_depth * dl_tex_rows * std::max<size_t>(dl_lanes, 1)))) return rc;
            dl_guard.p.push_back(dl_tex);
            if (s->has_dynamic) {
                if ((rc = dev_alloc(&dl_dyn, (size_t)d->max_depth * std::max<size_t>(dl_lanes, 1)))) return rc;
                dl_guard.p.push_back(dl_dyn);
            }
        }
        if (!pixel_sampler) {
            if ((rc = dev_alloc(&dl_words, 2))) return rc;
            dl_guard.p.push_back(dl_words);
            if (d->n_light_samples && s->dev.n_lights) {
                if ((rc = dev_alloc(&dl_nls, s->dev.n_lights))) return rc;
                dl_guard.p.push_back(dl_nls);
                HIP_TRY(hipMemcpy(dl_nls, d->n_light_samples, s->dev.n_lights * sizeof(int32_t), hipMemcpyHostToDevice));
            }
        }
    }
    const uint32_t nominal_iters = d->max_depth + 1;
    // a pass through a null-material surface costs a wavefront iteration without counting as a bounce (path.rs:109-116 has no limit
    // on them); the loop below runs until no path is left, and a scene that needs more than RSPT_NULL_PASSES extra iterations is
    // reported, not silently truncated
    const uint32_t max_iters = s->has_null_material ? nominal_iters + (uint32_t)env_size("RSPT_NULL_PASSES", 1024) : nominal_iters;
    if (volpath) HIP_TRY(hipMemsetAsync(g.vol.truncated, 0, sizeof(uint32_t), g.stream));
    // K7b: whole waves of one class (escaped | depth limit | material) for k_shade.  C3 stand-in (two materials): 1396 -> 1686 Msamples/s;
    // C2 (one material: only the escaped paths are separated): 423 -> 425
    // (measured with the specialised shade instantiations, profiles/r03_ab_shade.md: C3 stand-in — plastic statue on a matte ground — 1640 -> 1830
    //  Msamples/s with the bins; C2 — one material, only escaped paths to separate — 468 -> 456: the sort runs when lobe lists differ)
    const bool shade_bins = env_size("RSPT_SHADE_BINS", s->shade_classes > 1 ? 1 : 0) != 0 && !ao && !direct && !volpath;
    if (shade_bins && (rc = ensure_bins(g.cap, max_iters + 10))) return rc;
    // iteration 0 (camera rays) arrives in pixel-major order — neighbouring slots, neighbouring pixels, mostly one material per wave already —
    // and has the longest queue of the batch: it is left unsorted (same box, alternating: C3 stand-in 1855 -> 1906 Msamples/s, textured
    // 1319 -> 1339; RSPT_BIN_FIRST=1 sorts it as before)
    const bool bins_first = env_size("RSPT_BIN_FIRST", 0) != 0;
    if (s->has_instances && (rc = ensure_hit_inst(volpath ? 2 * g.cap : g.cap))) return rc;   // volpath: second half = the hits of the shadow-ray segments
    // moving instances: every path keeps its ray time (the camera sample's), read where an instance's Transform is interpolated
    if (s->has_animated) {
        if (g.time_cap < g.cap) {
            if (g.path_time) (void)hipFree(g.path_time);
            g.path_time = nullptr; g.time_cap = 0;
            if ((rc = dev_alloc(&g.path_time, g.cap))) return rc;
            g.time_cap = g.cap;
        }
    }
    g.pb.time = s->has_animated ? g.path_time : nullptr;
    s->dev.ray_time = g.pb.time;
    s->dev.time_div = 1u;
    g.pb.fresh = 0u;
    g.pb.hit_inst = s->has_instances ? g.hit_inst : nullptr;
    g.vol.hit_inst_tr = (volpath && s->has_instances) ? g.hit_inst + g.cap : nullptr;
    if ((rc = ensure_counts(max_iters + 10)) || (rc = ensure_overflow_list(trace_can_overflow(s) ? 3 * g.cap : 1024)) || (rc = ensure_spill(pw_spill_threads())) ||
        (s->has_textures && (rc = ensure_tex_rows(s->has_dynamic)))) return rc;
    if (!g.totals) { if ((rc = dev_alloc(&g.totals, 8))) return rc; }
    HIP_TRY(hipMemsetAsync(g.totals, 0, 8 * sizeof(unsigned long long), g.stream));

    // LDS copy of the Sobol' matrices for k_shade: dimensions 0 .. 5 + 8 per bounce (+8 of read-ahead), index bits
    // 2*log2(resolution) + log2(spp) (lowdiscrepancy.rs:1014-1043); the checks above keep this under 64 KB
    const uint32_t sob_nd = 5u + 8u * (d->max_depth + 2u) + 8u;
    uint32_t sob_bits = 2u * (uint32_t)rd.log2_res;
    for (int64_t v = d->spp; v > 1; v >>= 1) sob_bits++;
    sob_bits = std::min(52u, sob_bits + 1u);
    if ((size_t)sob_nd * sob_bits * 4 > 64 * 1024) return fail(RSPT_E_UNSUPPORTED, "max_depth %u x %u index bits exceed the LDS Sobol' table", d->max_depth, sob_bits);
    const uint32_t tgrid = trace_grid();
    const ShadeKernel shade_k = shade_slot_k;   // (iterations >= move_first of a MOVE render launch shade_move_k)
    if (getenv("RSPT_VERBOSE")) fprintf(stderr, "rspt: shade stage instantiation '%s'%s (scene features %#x)\n", shade_name, move ? ", moving path state" : "", s->shade_features);
    // one launch fills the chip once: as many 256-thread blocks per CU as the instantiation's registers and the LDS table allow
    int shade_blocks = 2;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&shade_blocks, reinterpret_cast<const void*>(shade_k), 256, sob_nd * sob_bits * sizeof(uint32_t)) != hipSuccess || shade_blocks < 1) {
        (void)hipGetLastError();
        shade_blocks = 2;
    }
    const uint32_t sgrid = grid_for((uint32_t)env_size("RSPT_SHADE_BLOCKS_PER_CU", (size_t)shade_blocks));
    if (getenv("RSPT_VERBOSE")) fprintf(stderr, "rspt: shade stage: %d blocks of 256 per CU\n", shade_blocks);
    // dynamic materials: one Built record per thread of the widest shade-stage launch (k_shade's grid; k_vol_shade runs grid_for(4))
    if (s->has_dynamic && (rc = ensure_dyn_built(std::max(sgrid, grid_for(8)) * 256u))) return rc;
    size_t n_ev = 0;
    const bool two_streams = env_size("RSPT_TRACE_STREAMS", 2) >= 2 && !counters;
    hipEvent_t ev_fork = get_event(n_ev++), ev_join = get_event(n_ev++);
    uint64_t trace_launches = 0;
    hipEvent_t ev_k0 = get_event(n_ev++), ev_k1 = get_event(n_ev++);
    std::vector<std::pair<hipEvent_t, hipEvent_t>> trace_ev;
    // per-launch durations, each pair recorded on the stream its kernel runs on (closest-hit, shadow-ray, shade [+ texture])
    std::vector<std::pair<hipEvent_t, hipEvent_t>> kev[3];
    auto ev_open = [&](int kind, int lane) {
        hipEvent_t a = get_event(n_ev++), b = get_event(n_ev++);
        (void)hipEventRecord(a, lane ? g.stream2 : g.stream);
        kev[kind].push_back({a, b});
    };
    auto ev_close = [&](int kind, int lane) { (void)hipEventRecord(kev[kind].back().second, lane ? g.stream2 : g.stream); };
    HIP_TRY(hipEventRecord(ev_k0, g.stream));
    uint64_t samples = 0, truncated = 0, vol_rays = 0;
    // ---- one batch of each integrator (the wavefront schedules); `it` leaves with the number of queue-counter records used ----
    // Which kernel serves this scene's shadow rays (trace_w4q.h; launch_trace_v): measured once per scene, by the first shadow-ray launch of a batch of >= 2^22 paths — both kernels
    // run on the same rays (their flags are identical), the faster one is kept in rspt_scene_s::any_q_choice.  Returns 1 if it made the launch (twice), 0 if there was nothing to
    // measure (the caller launches as usual), -code on an error.
    auto tune_any = [&](uint32_t batch_n, const uint32_t* queue, uint32_t* count_ptr, uint32_t* cursor, const rspt_ray* rays, uint32_t* occ, uint32_t* xcd) -> int {
        if (!s->w4q || s->any_q_choice >= 0 || getenv("RSPT_ANY_Q") || counters || batch_n < (1u << 22) || env_size("RSPT_ANY_Q_TUNE", 1) == 0) return 0;
        hipEvent_t t0 = get_event(n_ev++), t1 = get_event(n_ev++), t2 = get_event(n_ev++);
        if (hipEventRecord(t0, g.stream) != hipSuccess) return RSPT_E_HIP;
        g_any_q_force = 0;
        launch_trace<true, 0>(0, false, tgrid, s, queue, count_ptr, 0, cursor, rays, rays, nullptr, nullptr, occ, nullptr, g.totals, xcd);
        (void)hipEventRecord(t1, g.stream);
        (void)hipMemsetAsync(cursor, 0, sizeof(uint32_t), g.stream);   // (the persistent kernel's fetch cursor: the second run starts over)
        g_any_q_force = 1;
        launch_trace<true, 0>(0, false, tgrid, s, queue, count_ptr, 0, cursor, rays, rays, nullptr, nullptr, occ, nullptr, g.totals, xcd);
        g_any_q_force = -1;
        (void)hipEventRecord(t2, g.stream);
        if (hipEventSynchronize(t2) != hipSuccess) return RSPT_E_HIP;
        float ms_plain = 0.0f, ms_q = 0.0f;
        (void)hipEventElapsedTime(&ms_plain, t0, t1);
        (void)hipEventElapsedTime(&ms_q, t1, t2);
        s->any_q_choice = ms_q < ms_plain ? 1 : 0;
        if (getenv("RSPT_VERBOSE")) fprintf(stderr, "rspt: shadow rays of this scene: k_trace_w4<any> %.2f ms, k_trace_w4q %.2f ms on the same launch -> %s\n", ms_plain, ms_q, s->any_q_choice ? "the quantised records" : "the plain records");
        return 1;
    };
    auto batch_volpath = [&](const Batch& bt, uint32_t& it) -> int {  // VolPathIntegrator::li (vol.h): the continuation queue doubles as the list of live paths
        const uint32_t dgrid = grid_for(4);
        const uint32_t null_passes = (uint32_t)env_size("RSPT_NULL_PASSES", 1024);
        // LDS table: 10 dimensions per counted pass, 2 per pass through a medium boundary (room for 64 of those), 8 of read-ahead;
        // a path that needs more is cut and counted (rspt_stats.truncated_paths); the reference's own limit is 1024
        const uint32_t vnd = std::min(1024u, 5u + 10u * (d->max_depth + 2u) + 128u) + 8u;
        if ((size_t)vnd * sob_bits * 4 > 64 * 1024) return fail(RSPT_E_UNSUPPORTED, "volpath: max_depth %u x %u index bits exceed the LDS Sobol' table", d->max_depth, sob_bits);
        const uint32_t vlimit = halton ? vol_dim_limit : std::min(1024u, vnd - 8u);
        hipLaunchKernelGGL(k_vol_init, dim3((bt.n + 255) / 256), dim3(256), 0, g.stream, g.vol, bt.n);
        // counters: g.cnt[0 / 1] = the continuation queue of this / the next pass, g.cnt[2 / 3] = the shadow-ray segments
        uint32_t live = bt.n;
        for (uint32_t pass = 0; live > 0; pass++) {
            const int par = pass & 1;
            QueueCounts* cur = &g.cnt[par];
            QueueCounts* nxt = &g.cnt[par ^ 1];
            HIP_TRY(hipMemsetAsync(nxt, 0, sizeof(QueueCounts), g.stream));
            HIP_TRY(hipMemsetAsync(&g.cnt[2], 0, sizeof(QueueCounts), g.stream));
            ev_open(0, 0);
            launch_trace<false, 0>(0, counters, tgrid, s, g.q[par][1], &cur->closest, 0, &cur->cursor_closest, g.pb.ray_cont, g.pb.ray_mis, g.pb.hit_cont, g.pb.hit_mis, nullptr, nullptr, g.totals);
            ev_close(0, 0);
            trace_launches++;
            vol_rays += live;
            ev_open(2, 0);
            if (s->has_textures) hipLaunchKernelGGL(k_texture, dim3(dgrid), dim3(256), 0, g.stream, s->dev, s->tex, rd, g.pb, g.q[par][1], &cur->closest, (const uint32_t*)nullptr, (const BinInfo*)nullptr);
            if (ld.lazy) HIP_TRY(hipMemsetAsync(&cur->active, 0, sizeof(uint32_t), g.stream));   // (k_raygen leaves the batch size there; volpath itself does not use the active queues)
            hipLaunchKernelGGL(s->has_dynamic ? k_vol_shade<true> : k_vol_shade<false>, dim3(dgrid), dim3(256), halton ? 0 : vnd * sob_bits * sizeof(uint32_t), g.stream, s->dev, ld, rd, g.pb, g.vol, g.q[par][1], &cur->closest,
                               g.q[par ^ 1][1], &nxt->closest, g.q[0][2], &g.cnt[2].closest, vlimit, vnd, sob_bits, ld.lazy ? g.q[par][0] : (uint32_t*)nullptr, &cur->active);
            // on-demand light voxels: paths whose voxel had no row were put back (q[par][0], counted in cur->active, zero until here); build the rows, run those paths
            for (uint32_t round = 0; ld.lazy; round++) {
                uint32_t claimed = 0;
                if ((rc = build_claimed_rows(&claimed))) return rc;
                HIP_TRY(hipMemcpyAsync(g.look, cur, sizeof(QueueCounts), hipMemcpyDeviceToHost, g.stream));
                HIP_TRY(hipStreamSynchronize(g.stream));
                const uint32_t n_retry = g.look[0].active;
                if (n_retry == 0) break;
                if (round > 64) return fail(RSPT_E_UNSUPPORTED, "volpath: on-demand light voxels did not settle in 64 rounds (not a device fault: the caller keeps its CPU loop, or asks for the eager table)");
                // the retry queue becomes the input (copied to the other parity's active queue, which volpath does not use either), its counter starts again
                HIP_TRY(hipMemcpyAsync(g.q[par ^ 1][0], g.q[par][0], (size_t)n_retry * sizeof(uint32_t), hipMemcpyDeviceToDevice, g.stream));
                HIP_TRY(hipMemcpyAsync(&cur->any, &cur->active, sizeof(uint32_t), hipMemcpyDeviceToDevice, g.stream));   // (cur->any: the retry run's input length)
                HIP_TRY(hipMemsetAsync(&cur->active, 0, sizeof(uint32_t), g.stream));
                hipLaunchKernelGGL(s->has_dynamic ? k_vol_shade<true> : k_vol_shade<false>, dim3(dgrid), dim3(256), halton ? 0 : vnd * sob_bits * sizeof(uint32_t), g.stream, s->dev, ld, rd, g.pb, g.vol,
                                   g.q[par ^ 1][0], &cur->any, g.q[par ^ 1][1], &nxt->closest, g.q[0][2], &g.cnt[2].closest, vlimit, vnd, sob_bits, g.q[par][0], &cur->active);
            }
            ev_close(2, 0);
            // VisibilityTester::tr: segments until every shadow ray has arrived or is blocked
            // (the first two segments are launched without looking at the queue: most shadow rays cross at most one boundary, an
            //  empty launch costs microseconds, a look costs a stream synchronisation; the look that follows also brings the
            //  next pass's path count)
            QueueCounts* look = g.look;   // (pinned: see Ctx::look)
            bool have_live = false;
            for (uint32_t seg = 0;; seg++) {
                QueueCounts* tc = &g.cnt[2 + (seg & 1u)];
                QueueCounts* tn = &g.cnt[2 + ((seg + 1u) & 1u)];
                QueueCounts c{};
                c.closest = live;   // upper bound while not looking (every live path has at most one shadow ray)
                const bool looked = seg >= 2 || counters;   // (the counting pass wants every queue length)
                if (looked) {
                    HIP_TRY(hipMemcpyAsync(look, g.cnt, 4 * sizeof(QueueCounts), hipMemcpyDeviceToHost, g.stream));
                    HIP_TRY(hipStreamSynchronize(g.stream));
                    c = look[2 + (seg & 1u)];
                    have_live = true;
                    if (c.closest == 0) break;
                }
                if (seg > null_passes) { truncated += c.closest; break; }
                HIP_TRY(hipMemsetAsync(tn, 0, sizeof(QueueCounts), g.stream));
                ev_open(1, 0);
                // (queue entries without the MIS flag over the shadow rays' own arrays, so that the hit's instance is recorded too)
                g_inst_out = g.vol.hit_inst_tr ? g.hit_inst + g.cap : nullptr;
                launch_trace<false, 0>(0, counters, tgrid, s, g.q[seg & 1u][2], &tc->closest, 0, &tc->cursor_closest, g.pb.ray_mis, g.pb.ray_mis, g.pb.hit_mis, g.pb.hit_mis, nullptr, nullptr, g.totals);
                g_inst_out = nullptr;
                ev_close(1, 0);
                trace_launches++;
                if (looked) vol_rays += c.closest;
                hipLaunchKernelGGL(s->has_animated ? k_vol_tr<true> : k_vol_tr<false>, dim3(dgrid), dim3(256), 0, g.stream, s->dev, g.pb, g.vol, g.q[seg & 1u][2], &tc->closest, g.q[(seg + 1u) & 1u][2], &tn->closest);
            }
            if (!have_live) {
                HIP_TRY(hipMemcpyAsync(look, g.cnt, 4 * sizeof(QueueCounts), hipMemcpyDeviceToHost, g.stream));
                HIP_TRY(hipStreamSynchronize(g.stream));
            }
            live = look[par ^ 1].closest;
            if (live && pass >= nominal_iters + null_passes) { truncated += live; break; }
        }
        it = 4;
        return RSPT_OK;
    };
    auto batch_direct = [&](const Batch& bt, uint32_t& it) -> int {  // DirectLightingIntegrator::li (direct.h): specular tree, dimension assignment, light rounds, gather
        const uint32_t nl = s->dev.n_lights, H = dl_H, md = d->max_depth;
        const bool all = d->direct_strategy == RSPT_DIRECT_SAMPLE_ALL;
        const uint32_t n_arrays = all ? 2u * md * nl : 0u;
        // the sample arrays are filled for every pixel sample whether a node uses them or not (GlobalSampler::start_pixel), so they must fit;
        // the regular stream behind them is checked per camera sample by k_dl_assign against what each tree really draws
        const uint32_t dim_limit = halton ? vol_dim_limit + 1u : 1024u;
        if (5ull + 2ull * n_arrays > dim_limit)
            return fail(RSPT_E_UNSUPPORTED, "directlighting: %u sample arrays exceed the sampler's %u dimensions", n_arrays, dim_limit);
        DlBuf dl = g.dl;
        dl.H = H; dl.levels = dl_levels;
        s->dev.time_div = H;   // node h of camera sample s lives in slot s * H + h: its rays carry the sample's time (moving instances); estimate rays: below
        struct TimeDivReset { rspt_scene_s* s; ~TimeDivReset() { s->dev.time_div = 1u; } } time_div_reset{s};
        const size_t n_slots = (size_t)bt.n * H;
        // virtual slots of the estimates (direct.h DlBuf::vs / vr): planes of n_slots, unless a moving instance needs slot -> camera sample by one division (RSPT_DL_PLANES=0: A/B)
        const bool dl_planes = !s->has_animated && env_size("RSPT_DL_PLANES", 1) != 0;
        dl.vs = dl_planes ? 1u : dl_R; dl.vr = dl_planes ? (uint32_t)n_slots : 1u;
        if (s->has_textures) HIP_TRY(hipMemsetAsync(g.pb.state, 0, n_slots * sizeof(uint32_t), g.stream));   // (ST_NO_DIFF marks of k_dl_hit, read by k_dl_texture)
        HIP_TRY(hipMemsetAsync(dl.le_kind, 0, n_slots * sizeof(float4), g.stream));
        HIP_TRY(hipMemsetAsync(dl.l_all, 0, n_slots * sizeof(float4), g.stream));
        HIP_TRY(hipMemsetAsync(dl.ld_acc, 0, n_slots * sizeof(float4), g.stream));
        HIP_TRY(hipMemsetAsync(dl.error, 0, sizeof(uint32_t), g.stream));
        // the camera rays were left in ray_cont[sample]; node slots overlay that array, so move them aside first
        HIP_TRY(hipMemcpyAsync(g.pb.ray_sh, g.pb.ray_cont, (size_t)bt.n * sizeof(rspt_ray), hipMemcpyDeviceToDevice, g.stream));
        // counters: g.cnt[level].closest = nodes of the level, .any = re-trace queue; rounds use g.cnt[md + 1 ..]
        auto level_q = [&](uint32_t l) { return g.dl_queue + (size_t)bt.n * ((1u << l) - 1u); };
        hipLaunchKernelGGL(k_dl_init, dim3((bt.n + 255) / 256), dim3(256), 0, g.stream, bt, g.pb, dl, g.pb.ray_sh, level_q(0), &g.cnt[0].closest);
        const uint32_t dgrid = grid_for(4);
        for (uint32_t l = 0; l < dl_levels; l++) {
            const uint32_t* queue = level_q(l);
            const uint32_t* qcount = &g.cnt[l].closest;
            for (uint32_t round = 0;; round++) {
                QueueCounts* rc_ = &g.cnt[md + 1 + (round & 1u)];  // re-trace queue of this round (null-BSDF hits), double buffered
                HIP_TRY(hipMemsetAsync(rc_, 0, sizeof(QueueCounts), g.stream));
                HIP_TRY(hipMemsetAsync((void*)&g.cnt[l].cursor_closest, 0, 3 * sizeof(uint32_t), g.stream));
                ev_open(0, 0);
                launch_trace<false, 0>(0, false, tgrid, s, queue, qcount, 0, round == 0 ? &g.cnt[l].cursor_closest : &g.cnt[md + 1 + ((round - 1) & 1u)].cursor_closest,
                                       g.pb.ray_cont, g.pb.ray_mis, g.pb.hit_cont, g.pb.hit_mis, nullptr, nullptr, g.totals);
                ev_close(0, 0);
                trace_launches++;
                ev_open(2, 0);
                hipLaunchKernelGGL(k_dl_hit, dim3(dgrid), dim3(256), 0, g.stream, s->dev, rd, g.pb, dl, queue, qcount, g.q[round & 1u][0], &rc_->closest,
                                   level_q(l + 1 < dl_levels ? l + 1 : l), &g.cnt[l + 1].closest, l);
                ev_close(2, 0);
                if (!s->has_null_material) break;
                HIP_TRY(hipMemcpyAsync(g.look, rc_, sizeof(QueueCounts), hipMemcpyDeviceToHost, g.stream));
                HIP_TRY(hipStreamSynchronize(g.stream));
                const QueueCounts c = g.look[0];
                if (c.closest == 0) break;
                if (round >= env_size("RSPT_NULL_PASSES", 1024)) { truncated += c.closest; break; }
                queue = g.q[round & 1u][0];
                qcount = &rc_->closest;
            }
        }
        hipLaunchKernelGGL(k_dl_assign, dim3((bt.n + 255) / 256), dim3(256), 0, g.stream, bt, dl, nl, n_arrays, all ? 1u : 0u, md, dim_limit);
        if (s->has_textures)   // (dl_tex_wave: one level, the roots) the texture stage in front of the estimates
            for (uint32_t l = 0; l < dl_levels; l++)
                hipLaunchKernelGGL(k_dl_texture, dim3(dgrid), dim3(256), 0, g.stream, s->dev, s->tex, rd, g.pb, dl, level_q(l), &g.cnt[l].closest);
        if (nl && dl_one_round) {   // every estimate of a level's nodes in one round: 4 launches per level (direct.h k_dl_nee_all)
            QueueCounts* rc_ = &g.cnt[md + 3];
            for (uint32_t l = 0; l < dl_levels; l++) {
                HIP_TRY(hipMemsetAsync(rc_, 0, sizeof(QueueCounts), g.stream));
                ev_open(2, 0);
                // the estimate kernel per feature set, as k_shade<F>: scenes of Lambert / microfacet-reflection lobes under area lights without instances (C1 - C3) take the
                // narrow build (RSPT_DL_VARIANT=generic forces the other)
                constexpr uint32_t DLV_PLASTIC = SV_PLASTIC | SF_SOBOL | SF_HALTON;
                const bool dl_narrow = (s->shade_features & ~DLV_PLASTIC) == 0 && !(getenv("RSPT_DL_VARIANT") && !strcmp(getenv("RSPT_DL_VARIANT"), "generic"));
                const size_t dl_waves = env_size("RSPT_DL_WAVES", RSPT_DL_WAVES_DEFAULT);   // 3: the narrow build forced to 3 waves per SIMD
                // the Sobol' tables the estimates read, in LDS (direct.h DlSob): the sample arrays' dimensions 5 .. 5 + 2 n_arrays and what the regular stream adds behind them,
                // for indices of 2 log2_res + log2(spp x the longest array) bits; cut to 40 KB (the rest falls back to the global walks); RSPT_DL_LDS_SOBOL=0: as before
                uint32_t dsn = 0, dsb = 0;
                if (!halton && env_size("RSPT_DL_LDS_SOBOL", 1) != 0) {
                    uint64_t longest = 1;
                    for (uint32_t j = 0; all && d->n_light_samples && j < nl; j++) longest = std::max<uint64_t>(longest, (uint64_t)std::max<int32_t>(d->n_light_samples[j], 1));
                    dsb = 2u * (uint32_t)rd.log2_res + 1u;
                    for (uint64_t v = (uint64_t)std::max<int64_t>(d->spp, 1) * longest; v > 1; v >>= 1) dsb++;
                    dsb = std::min(52u, dsb);
                    dsn = std::min<uint32_t>(1024u, 5u + 2u * n_arrays + 8u * (md + 2u));
                    dsn = std::min<uint32_t>(dsn, (40u * 1024u) / (4u * dsb));
                    if (dsn < 16u) dsn = dsb = 0;
                }
                // the lights and the area lights' triangle records in LDS too (direct.h DlSob::lights: their loads must not queue behind the estimates' stores); the count
                // rides in bits 16.. of the index-bits argument.  RSPT_DL_LDS_LIGHTS=0: from global memory as before
                const uint32_t dll = (nl <= DL_LDS_LIGHTS && env_size("RSPT_DL_LDS_LIGHTS", 1) != 0) ? nl : 0u;
                const size_t dl_lds = (dll ? (((size_t)dll * (48 + sizeof(rspt_light)) + 7) / 8) * 8 : 0) + (dsn ? 104 * sizeof(uint64_t) + (size_t)dsn * dsb * sizeof(uint32_t) : 0);
                dsb |= dll << 16;
                hipLaunchKernelGGL(dl_narrow ? (dl_waves == 3 ? k_dl_nee_all_w<DLV_PLASTIC, 3> : k_dl_nee_all<DLV_PLASTIC>) : k_dl_nee_all<SF_ALL>, dim3(dgrid), dim3(256), dl_lds, g.stream, s->dev, rd, bt, g.pb, dl, g.pix_list, level_q(l), &g.cnt[l].closest, (const int32_t*)dl_nls, dl_R,
                                   n_arrays, all ? 1u : 0u, g.q[0][2], &rc_->any, g.q[0][1], &rc_->closest, dsn, dsb);
                ev_close(2, 0);
                ev_open(1, 0);
                s->dev.time_div = H * dl_R;   // estimate r of node slot n sits in virtual slot n * R + r
                {
                    const int tuned = tune_any(bt.n, g.q[0][2], &rc_->any, &rc_->cursor_any, g.pb.ray_sh, g.pb.occluded, nullptr);   // (the scene's first large shadow-ray launch measures the two kernels)
                    if (tuned < 0) return fail(tuned, "the shadow-ray kernel measurement failed");
                    if (!tuned) launch_trace<true, 0>(0, false, tgrid, s, g.q[0][2], &rc_->any, 0, &rc_->cursor_any, g.pb.ray_sh, g.pb.ray_sh, nullptr, nullptr, g.pb.occluded, nullptr, g.totals);
                }
                ev_close(1, 0);
                ev_open(0, 0);
                launch_trace<false, 0>(0, false, tgrid, s, g.q[0][1], &rc_->closest, 0, &rc_->cursor_closest, g.pb.ray_cont, g.pb.ray_mis, g.pb.hit_cont, g.pb.hit_mis, nullptr, nullptr, g.totals);
                s->dev.time_div = H;
                ev_close(0, 0);
                trace_launches += 2;
                hipLaunchKernelGGL(k_dl_nee_resolve_all, dim3(dgrid), dim3(256), 0, g.stream, s->dev, g.pb, dl, level_q(l), &g.cnt[l].closest, (const int32_t*)dl_nls, dl_R, n_arrays, all ? 1u : 0u);
            }
        } else if (nl) {
            QueueCounts* rc_ = &g.cnt[md + 3];
            for (uint32_t l = 0; l < dl_levels; l++) {
                const uint32_t n_lights_round = all ? nl : 1u;
                for (uint32_t j = 0; j < n_lights_round; j++) {
                    const uint32_t n_j = all ? (uint32_t)(d->n_light_samples ? d->n_light_samples[j] : 1) : 1u;
                    for (uint32_t kk = 0; kk < n_j; kk++) {
                        HIP_TRY(hipMemsetAsync(rc_, 0, sizeof(QueueCounts), g.stream));
                        ev_open(2, 0);
                        hipLaunchKernelGGL(k_dl_nee, dim3(dgrid), dim3(256), 0, g.stream, s->dev, rd, bt, g.pb, dl, g.pix_list, level_q(l), &g.cnt[l].closest, j, kk, n_j,
                                           n_arrays, all ? 1u : 0u, g.q[0][2], &rc_->any, g.q[0][1], &rc_->closest);
                        ev_close(2, 0);
                        ev_open(1, 0);
                        launch_trace<true, 0>(0, false, tgrid, s, g.q[0][2], &rc_->any, 0, &rc_->cursor_any, g.pb.ray_sh, g.pb.ray_sh, nullptr, nullptr, g.pb.occluded, nullptr, g.totals);
                        ev_close(1, 0);
                        ev_open(0, 0);
                        launch_trace<false, 0>(0, false, tgrid, s, g.q[0][1], &rc_->closest, 0, &rc_->cursor_closest, g.pb.ray_cont, g.pb.ray_mis, g.pb.hit_cont, g.pb.hit_mis, nullptr, nullptr, g.totals);
                        ev_close(0, 0);
                        trace_launches += 2;
                        hipLaunchKernelGGL(k_dl_nee_resolve, dim3(dgrid), dim3(256), 0, g.stream, s->dev, g.pb, dl, level_q(l), &g.cnt[l].closest, j, kk, n_j, n_arrays, all ? 1u : 0u);
                    }
                }
            }
        }
        hipLaunchKernelGGL(k_dl_gather, dim3((bt.n + 255) / 256), dim3(256), 0, g.stream, bt, g.pb, dl, nl, md);
        uint32_t dl_err = 0;
        HIP_TRY(hipMemcpyAsync(&dl_err, dl.error, sizeof dl_err, hipMemcpyDeviceToHost, g.stream));
        HIP_TRY(hipStreamSynchronize(g.stream));
        if (dl_err == 1u) return RSPT_DL_RETRY_LANE;   // a material with several specular lobes of one kind: the lobe choice depends on a sample value, the tree cannot be traced ahead
        if (dl_err == 3u) return fail(RSPT_E_HIP, "directlighting: a specular bounce in a scene classified as having none");
        if (dl_err) return fail(RSPT_E_UNSUPPORTED, "directlighting: a camera sample draws more than the sampler's %u dimensions (the reference panics there, sobol.rs:119-124)", dim_limit);
        it = md + 4;
        return RSPT_OK;
    };
    auto batch_direct_lane = [&](const Batch& bt, uint32_t& it) -> int {  // the same integrator, one lane per camera sample (lane_serial.h)
        const uint32_t nl = s->dev.n_lights, md = d->max_depth;
        const bool all = d->direct_strategy == RSPT_DIRECT_SAMPLE_ALL;
        const uint32_t n_arrays = all ? 2u * md * nl : 0u;
        const uint32_t dim_limit = halton ? vol_dim_limit + 1u : 1024u;
        if (5ull + 2ull * n_arrays > dim_limit)
            return fail(RSPT_E_UNSUPPORTED, "directlighting: %u sample arrays exceed the sampler's %u dimensions", n_arrays, dim_limit);
        HIP_TRY(hipMemsetAsync(dl_words, 0, 2 * sizeof(uint32_t), g.stream));
        const LaneDesc ln{all ? dl_nls : nullptr, n_arrays, all ? 1u : 0u, dim_limit, dl_tex, (uint32_t)dl_lanes, dl_tex_rows, dl_dyn,
                          s->has_null_material ? (uint32_t)env_size("RSPT_NULL_PASSES", 1024) : 0u, dl_words, dl_words + 1};
        const dim3 lgrid((bt.n + 63u) / 64u);
        ev_open(2, 0);
#define RSPT_LN(I, A) hipLaunchKernelGGL((k_lane_dl<I, A>), lgrid, dim3(64), 0, g.stream, s->dev, s->tex, ld, rd, bt, g.pb, g.pix_list, ln)
        if (s->has_animated) {   // (round 6: the walk interpolates the instances it enters at the sample's ray time, pb.time)
            if (s->has_alpha) hipLaunchKernelGGL((k_lane_dl<true, true, true>), lgrid, dim3(64), 0, g.stream, s->dev, s->tex, ld, rd, bt, g.pb, g.pix_list, ln);
            else hipLaunchKernelGGL((k_lane_dl<true, false, true>), lgrid, dim3(64), 0, g.stream, s->dev, s->tex, ld, rd, bt, g.pb, g.pix_list, ln);
        } else if (s->has_instances) { if (s->has_alpha) RSPT_LN(true, true); else RSPT_LN(true, false); }
        else { if (s->has_alpha) RSPT_LN(false, true); else RSPT_LN(false, false); }
#undef RSPT_LN
        ev_close(2, 0);
        uint32_t w[2] = {0, 0};
        HIP_TRY(hipMemcpyAsync(w, dl_words, sizeof w, hipMemcpyDeviceToHost, g.stream));
        HIP_TRY(hipStreamSynchronize(g.stream));
        if (w[0]) return fail(RSPT_E_UNSUPPORTED, "directlighting: a camera sample draws more than the sampler's %u dimensions (the reference panics there, sobol.rs:119-124)", dim_limit);
        truncated += w[1];
        it = 1;
        return RSPT_OK;
    };
    auto batch_ao = [&](const Batch& bt, uint32_t& it) -> int {  // AOIntegrator::li: closest hit, n shadow rays per hit, sum of the unoccluded terms
        hipEvent_t e0 = get_event(n_ev++), e1 = get_event(n_ev++), e2 = get_event(n_ev++), e3 = get_event(n_ev++);
        HIP_TRY(hipEventRecord(e0, g.stream));
        ev_open(0, 0);
        launch_trace<false, 0>(0, counters, tgrid, s, g.q[0][1], &g.cnt[0].closest, 0, &g.cnt[0].cursor_closest, g.pb.ray_cont, g.pb.ray_mis, g.pb.hit_cont, g.pb.hit_mis, nullptr, nullptr, g.totals);
        ev_close(0, 0);
        HIP_TRY(hipEventRecord(e1, g.stream));
        hipLaunchKernelGGL(s->has_animated ? k_ao_spawn<true> : k_ao_spawn<false>, dim3((bt.n + 255) / 256), dim3(256), 0, g.stream, s->dev, rd, bt, g.pb, g.pix_list, ao_n, d->ao_cos_sample, g.q[0][2], &g.cnt[1]);
        HIP_TRY(hipEventRecord(e2, g.stream));
        ev_open(1, 0);
        s->dev.time_div = ao_n;   // shadow ray k of camera sample i sits in slot i * n + k: its Ray.time is the sample's (moving instances)
        launch_trace<true, 0>(0, counters, tgrid, s, g.q[0][2], &g.cnt[1].any, 0, &g.cnt[1].cursor_any, g.pb.ray_sh, g.pb.ray_sh, nullptr, nullptr, g.pb.occluded, nullptr, g.totals);
        s->dev.time_div = 1u;
        ev_close(1, 0);
        HIP_TRY(hipEventRecord(e3, g.stream));
        trace_ev.push_back({e0, e1}); trace_ev.push_back({e2, e3});
        trace_launches += 2;
        hipLaunchKernelGGL(k_ao_resolve, dim3((bt.n + 255) / 256), dim3(256), 0, g.stream, bt, g.pb, ao_n);
        it = 2;
        return RSPT_OK;
    };
    auto batch_path = [&](const Batch& bt, uint32_t& it) -> int {  // PathIntegrator::li: trace (closest || any) -> [light voxels] -> [bins] -> [textures] -> shade, per bounce
        for (;;) {
            const int par = it & 1;
            if (it > 0) g.pb.fresh = 0u;   // (PathBuf travels by value: the first launches of the batch have carried the flag k_raygen ran with)
            const PathBuf P = move ? move_pathbuf(it, move_first) : g.pb;   // MOVE: the set this iteration reads (written by the previous one's shade launch) and the set it writes
            hipEvent_t e0 = get_event(n_ev++), e1 = get_event(n_ev++);
            HIP_TRY(hipEventRecord(e0, g.stream));
            // the shadow-ray launch does not depend on the closest-hit launch: on a second stream its tail (a few
            // long rays on an otherwise idle chip) overlaps the other launch
            int any_lane = (it > 0 && two_streams) ? 1 : 0;
            bool any_done = false;
            if (it == 1) {   // (the scene's first large shadow-ray launch measures the two kernels: tune_any above)
                ev_open(1, 0);
                const int tuned = tune_any(bt.n, g.q[par][2], &g.cnt[it].any, &g.cnt[it].cursor_any, P.ray_sh, P.occluded, g.cnt[it].xcd_any);
                ev_close(1, 0);
                if (tuned < 0) return fail(tuned, "the shadow-ray kernel measurement failed");
                if (tuned) { any_done = true; any_lane = 0; }
            }
            if (any_lane) {
                HIP_TRY(hipEventRecord(ev_fork, g.stream));
                HIP_TRY(hipStreamWaitEvent(g.stream2, ev_fork, 0));
                ev_open(1, 1);
                launch_trace<true, 0>(1, counters, tgrid, s, g.q[par][2], &g.cnt[it].any, 0, &g.cnt[it].cursor_any, P.ray_sh, P.ray_sh, nullptr, nullptr, P.occluded, nullptr, g.totals, g.cnt[it].xcd_any);
                ev_close(1, 1);
                HIP_TRY(hipEventRecord(ev_join, g.stream2));
            }
            ev_open(0, 0);
            g_camera_launch = it == 0;
            launch_trace<false, 0>(0, counters, tgrid, s, g.q[par][1], &g.cnt[it].closest, 0, &g.cnt[it].cursor_closest, P.ray_cont, P.ray_mis, P.hit_cont, P.hit_mis, nullptr, nullptr, g.totals, g.cnt[it].xcd_closest);
            g_camera_launch = false;
            ev_close(0, 0);
            if (any_lane) HIP_TRY(hipStreamWaitEvent(g.stream, ev_join, 0));
            else if (it > 0 && !any_done) {
                ev_open(1, 0);
                launch_trace<true, 0>(0, counters, tgrid, s, g.q[par][2], &g.cnt[it].any, 0, &g.cnt[it].cursor_any, P.ray_sh, P.ray_sh, nullptr, nullptr, P.occluded, nullptr, g.totals, g.cnt[it].xcd_any);
                ev_close(1, 0);
            }
            HIP_TRY(hipEventRecord(e1, g.stream));
            trace_ev.push_back({e0, e1});
            trace_launches += it > 0 ? 2 : 1;
            ev_open(2, 0);
            const bool bins_now = shade_bins && (it > 0 || bins_first);
            if (bins_now) {  // K7b: whole waves of one class for k_shade
                const uint32_t bgrid = hinted_grid(grid_for(4), 256);
                hipLaunchKernelGGL(k_bin_count, dim3(bgrid), dim3(256), 0, g.stream, s->dev, P, d->max_depth, g.q[par][0], &g.cnt[it], g.bin_keys, &g.bin_info[it]);
                hipLaunchKernelGGL(k_bin_starts, dim3(1), dim3(64), 0, g.stream, &g.bin_info[it], g.q_sorted);
                hipLaunchKernelGGL(k_bin_scatter, dim3(bgrid), dim3(256), 0, g.stream, g.q[par][0], &g.cnt[it], g.bin_keys, &g.bin_info[it], g.q_sorted);
            }
            if (ld_lazy && d->integrator == RSPT_INTEGRATOR_PATH) {
                const uint32_t lgrid = hinted_grid(grid_for(4), 256);
                hipLaunchKernelGGL(k_ld_mark, dim3(lgrid), dim3(256), 0, g.stream, s->dev, ld, P, d->max_depth, g.q[par][0], &g.cnt[it], ld_lazy->lazy, ld_lazy->new_list);
                hipLaunchKernelGGL(k_ld_contrib_list, dim3(lgrid), dim3(256), 0, g.stream, s->dev, ld.nvox[0], ld.nvox[1], ld.nvox[2], ld_lazy->lazy, ld_lazy->new_list, ld_lazy->func);
                hipLaunchKernelGGL(k_ld_build_list, dim3(lgrid), dim3(256), 0, g.stream, s->dev.n_lights, ld_lazy->lazy, ld_lazy->new_list, ld_lazy->func, ld_lazy->cdf, ld_lazy->func_int, ld_lazy->table);
                hipLaunchKernelGGL(k_ld_commit, dim3(1), dim3(1), 0, g.stream, ld_lazy->lazy);
            }
            if (s->has_textures) {
                const bool tex_sorted = bins_now && env_size("RSPT_TEXTURE_SORTED", 1) != 0;
                hipLaunchKernelGGL(k_texture, dim3(sgrid), dim3(256), 0, g.stream, s->dev, s->tex, rd, P, g.q[par][0], &g.cnt[it].active,
                                   tex_sorted ? g.q_sorted : (const uint32_t*)nullptr, tex_sorted ? &g.bin_info[it] : (const BinInfo*)nullptr);
            }
            hipLaunchKernelGGL((move && it >= move_first) ? shade_move_k : shade_k, dim3(hinted_grid(sgrid, 256)), dim3(256), sob_nd * sob_bits * sizeof(uint32_t), g.stream, s->dev, ld, rd, P, g.q[par][0], &g.cnt[it], &g.cnt[it + 1], g.q[par ^ 1][0],
                               g.q[par ^ 1][1], g.q[par ^ 1][2], counters ? g.totals + 2 : nullptr, sob_nd, sob_bits, (uint32_t)g.cap,
                               bins_now ? g.q_sorted : (const uint32_t*)nullptr, bins_now ? &g.bin_info[it] : (const BinInfo*)nullptr);
            ev_close(2, 0);
            it++;
            if (it < nominal_iters) continue;
            // after max_depth + 1 bounces only pending estimates and null-material passes remain
            if (max_iters == nominal_iters) break;
            if (((it - nominal_iters) & 7u) != 0 && it < max_iters) continue;  // look at the queue length every 8th iteration: an empty iteration costs three idle launches, a look costs a stream sync
            HIP_TRY(hipMemcpyAsync(g.look, &g.cnt[it], sizeof(QueueCounts), hipMemcpyDeviceToHost, g.stream));
            HIP_TRY(hipStreamSynchronize(g.stream));
            const QueueCounts c = g.look[0];
            if (c.active == 0 && c.active_tail == 0) break;
            g_queue_hint = c.active + c.active_tail;
            if (it >= max_iters) {  // the reference's loop would still be running (path.rs:109-116 has no limit); these paths keep the radiance gathered so far
                truncated += c.active + c.active_tail;
                if (getenv("RSPT_VERBOSE") && c.active) {  // where the endless paths are: slot, film position and the ray in flight
                    uint32_t slots[4];
                    const uint32_t k = std::min(c.active, 4u);
                    HIP_TRY(hipMemcpy(slots, g.q[it & 1][0], k * sizeof(uint32_t), hipMemcpyDeviceToHost));
                    for (uint32_t j = 0; j < k; j++) {
                        rspt_ray r; float2 pf; float4 hc;
                        const PathBuf D = move ? move_pathbuf(it, move_first) : g.pb;   // (MOVE: queue entries are positions; the film position lives at the original slot)
                        uint32_t og = slots[j];
                        if (move && it > move_first) HIP_TRY(hipMemcpy(&og, D.orig + slots[j], sizeof og, hipMemcpyDeviceToHost));
                        HIP_TRY(hipMemcpy(&r, D.ray_cont + slots[j], sizeof r, hipMemcpyDeviceToHost));
                        HIP_TRY(hipMemcpy(&pf, D.p_film + og, sizeof pf, hipMemcpyDeviceToHost));
                        HIP_TRY(hipMemcpy(&hc, D.hit_cont + slots[j], sizeof hc, hipMemcpyDeviceToHost));
                        uint32_t o[3], dd[3], pr;
                        memcpy(o, r.o, 12); memcpy(dd, r.d, 12); memcpy(&pr, &hc.x, 4);
                        fprintf(stderr, "rspt: endless null-surface path: slot %u film (%.3f, %.3f) ray o %08x %08x %08x d %08x %08x %08x last prim %u\n",
                                slots[j], pf.x, pf.y, o[0], o[1], o[2], dd[0], dd[1], dd[2], pr);
                    }
                }
                break;
            }
        }
        // MOVE: whatever is still queued (paths cut at max_iters; normally nothing) hands its radiance to the film's array
        if (move) hipLaunchKernelGGL(k_move_flush, dim3(grid_for(1)), dim3(256), 0, g.stream, move_pathbuf(it, move_first), g.q[it & 1][0], &g.cnt[it], (uint32_t)g.cap, it > move_first ? 1u : 0u);
        return RSPT_OK;
    };
    auto run_tile_serial = [&]() -> int {  // the pixel samplers: one lane per tile (tile_serial.h)

        // ---- one lane per tile (tile_serial.h) ----
        std::vector<TileRec> tiles;
        for (size_t i = 0; i < blocks.size(); i++) {
            if ((i / chunk) % shard_count != d->shard_index) continue;
            const int32_t x0 = sb[0] + (int32_t)blocks[i].first * ts, x1 = std::min(x0 + ts, sb[2]);
            const int32_t y0 = sb[1] + (int32_t)blocks[i].second * ts, y1 = std::min(y0 + ts, sb[3]);
            tiles.push_back(TileRec{(int16_t)x0, (int16_t)y0, (int16_t)x1, (int16_t)y1, (uint32_t)((int32_t)blocks[i].second * ntx + (int32_t)blocks[i].first), 0u, 0u});
        }
        const uint32_t n_tiles = (uint32_t)tiles.size();
        const uint32_t spp = (uint32_t)d->spp, nd = d->pixel_dimensions;
        // rows of every tile per pass: as many as the sample-result arrays (24 B per sample) allow
        const size_t samp_cap = std::max<size_t>(env_size("RSPT_SERIAL_SAMPLES", (size_t)1 << 28), (size_t)ts * spp);
        int32_t rows = ts;
        while (rows > 1 && (size_t)n_tiles * rows * ts * spp > samp_cap) rows--;
        if ((size_t)n_tiles * rows * ts * spp > ((size_t)1 << 31)) return fail(RSPT_E_UNSUPPORTED, "pixel sampler: %u tiles x %u spp do not fit one pass", n_tiles, spp);
        struct Guard { std::vector<void*> p; ~Guard() { for (void* q : p) (void)hipFree(q); } } guard;
        auto tmp = [&](auto** p, size_t n) { int r = dev_alloc(p, std::max<size_t>(n, 1)); if (!r) guard.p.push_back(*p); return r; };
        TileRec* tiles_d = nullptr; float4* samp_L = nullptr; float2* samp_pf = nullptr; float* a1 = nullptr; float2* a2 = nullptr; uint64_t* rng_state = nullptr; uint64_t* rng_saved = nullptr;
        // the integrator's 2-D sample arrays (request_2d_array in preprocess): ao one of n_samples (ao.rs:47-49); directlighting, strategy all,
        // two per light and recursion level (directlighting.rs:54-70)
        float2* arr = nullptr; uint32_t* arr_sz_d = nullptr; uint32_t* arr_base_d = nullptr; int32_t* nls_d = nullptr;
        std::vector<uint32_t> arr_sz;
        if (ao) arr_sz.push_back(d->ao_n_samples);
        if (direct && d->direct_strategy == RSPT_DIRECT_SAMPLE_ALL)
            for (uint32_t lvl = 0; lvl < d->max_depth; lvl++)
                for (uint32_t j = 0; j < s->dev.n_lights; j++) { const uint32_t n = d->n_light_samples ? (uint32_t)d->n_light_samples[j] : 1u; arr_sz.push_back(n); arr_sz.push_back(n); }
        std::vector<uint32_t> arr_base(arr_sz.size());
        uint32_t arr_total = 0;
        for (size_t a = 0; a < arr_sz.size(); a++) { arr_base[a] = arr_total * spp; arr_total += arr_sz[a]; }
        if ((size_t)arr_total * spp * n_tiles > ((size_t)1 << 31)) return fail(RSPT_E_UNSUPPORTED, "pixel sampler: %u sample-array points x %u spp x %u tiles", arr_total, spp, n_tiles);
        if (!arr_sz.empty()) {
            if ((rc = tmp(&arr, (size_t)arr_total * spp * n_tiles)) || (rc = tmp(&arr_sz_d, arr_sz.size())) || (rc = tmp(&arr_base_d, arr_sz.size()))) return rc;
            HIP_TRY(hipMemcpyAsync(arr_sz_d, arr_sz.data(), arr_sz.size() * 4, hipMemcpyHostToDevice, g.stream));
            HIP_TRY(hipMemcpyAsync(arr_base_d, arr_base.data(), arr_base.size() * 4, hipMemcpyHostToDevice, g.stream));
        }
        if (direct && d->n_light_samples && s->dev.n_lights) {
            if ((rc = tmp(&nls_d, s->dev.n_lights))) return rc;
            HIP_TRY(hipMemcpyAsync(nls_d, d->n_light_samples, s->dev.n_lights * sizeof(int32_t), hipMemcpyHostToDevice, g.stream));
            HIP_TRY(hipStreamSynchronize(g.stream));   // (the caller's array, and the vectors above, must outlive the copies)
        }
        uint32_t* c_pixel_d = nullptr; uint32_t* trunc_d = nullptr; uint32_t* pass_pix = nullptr;
        const size_t max_samples = (size_t)n_tiles * rows * ts * spp;
        if ((rc = tmp(&tiles_d, n_tiles)) || (rc = tmp(&samp_L, max_samples)) || (rc = tmp(&samp_pf, max_samples)) || (rc = tmp(&a1, (size_t)nd * spp * n_tiles)) ||
            (rc = tmp(&a2, (size_t)nd * spp * n_tiles)) || (rc = tmp(&rng_state, 2 * (size_t)n_tiles)) || (rc = tmp(&rng_saved, ld.lazy ? 2 * (size_t)n_tiles : 1)) || (rc = tmp(&c_pixel_d, 32)) || (rc = tmp(&trunc_d, 2)) ||
            (rc = tmp(&pass_pix, (size_t)n_tiles * rows * ts)))
            return rc;
        HIP_TRY(hipMemsetAsync(trunc_d, 0, sizeof(uint32_t), g.stream));
        if (d->sampler_kind == RSPT_SAMPLER_MAXMINDIST) HIP_TRY(hipMemcpyAsync(c_pixel_d, d->maxmin_c_pixel, 32 * sizeof(uint32_t), hipMemcpyHostToDevice, g.stream));
        const PixDesc pd{d->sampler_kind, spp, d->sampler_kind == RSPT_SAMPLER_RANDOM ? 0u : nd, d->strat_x, d->strat_y, d->strat_jitter, c_pixel_d, a1, a2, rng_state, arr, arr_sz_d, arr_base_d, (uint32_t)arr_sz.size(), arr_total, d->ao_cos_sample, nls_d, d->direct_strategy, dl_tex, dl_tex_rows, dl_dyn};
        // lanes per wave: a lane that shares its wave waits whenever the others diverge (measured: four lanes of a wave take four times one lane's
        // time — no overlap at all), so the tiles are spread over waves, up to twice what the chip holds at these kernels' 2 waves / SIMD
        // (256 CUs x 4 SIMDs x 2 = 2048) before doubling up: C3 frame, 8160 tiles, 02sequence — 2048: 21.9, 4096: 24.7, 8192: 21.7 Msamples/s;
        // builds forced to 3 / 4 waves per SIMD (168 / 128 VGPRs, 2.3 / 2.5 KB of scratch) lose to the spills: 17.8 - 21.5
        uint32_t lanes = 1;
        while (lanes < 64 && (n_tiles + lanes - 1) / lanes > (uint32_t)env_size("RSPT_SERIAL_WAVES", 4096)) lanes *= 2;
        if (s->has_dynamic && (rc = ensure_dyn_built(((n_tiles + lanes - 1) / lanes) * 64u))) return rc;   // one lobe record per thread of the launch
        PathBuf fpb = g.pb;
        fpb.L_eta = samp_L; fpb.p_film = samp_pf;
        const uint32_t serial_iters = nominal_iters + 1u + (s->has_null_material ? (uint32_t)env_size("RSPT_NULL_PASSES", 1024) : 0u);
        std::vector<uint32_t> pl;
        for (int32_t r0 = 0; r0 < ts; r0 += rows) {
            const int32_t r1 = std::min(r0 + rows, ts);
            pl.clear();
            for (TileRec& t : tiles) {
                t.pix0 = (uint32_t)pl.size();
                for (int32_t y = t.y0 + r0; y < t.y0 + r1 && y < t.y1; y++)
                    for (int32_t x = t.x0; x < t.x1; x++) pl.push_back(((uint32_t)(uint16_t)(int16_t)y << 16) | (uint32_t)(uint16_t)(int16_t)x);
            }
            if (pl.empty()) continue;
            HIP_TRY(hipStreamSynchronize(g.stream));  // the previous pass still reads tiles_d / pass_pix
            HIP_TRY(hipMemcpyAsync(tiles_d, tiles.data(), n_tiles * sizeof(TileRec), hipMemcpyHostToDevice, g.stream));
            HIP_TRY(hipMemcpyAsync(pass_pix, pl.data(), pl.size() * sizeof(uint32_t), hipMemcpyHostToDevice, g.stream));
            const dim3 grid((n_tiles + lanes - 1) / lanes);
            // on-demand light voxels: a lane claims the voxels it finds without a row (dev_scene.h light_row_try) and goes on with row 0; the claimed rows are
            // built and the rows of the tiles rendered again from the saved generator states, until a run claims nothing — only that run's samples are kept
            // (a wrong row can change how many dimensions an estimate draws, so a run may leave the true paths after its first missing voxel: each
            // round completes at least the first one along every true chain)
            if (ld.lazy) {
                HIP_TRY(hipMemcpyAsync(rng_saved, rng_state, 2 * (size_t)n_tiles * sizeof(uint64_t), hipMemcpyDeviceToDevice, g.stream));
                HIP_TRY(hipMemcpyAsync(trunc_d + 1, trunc_d, sizeof(uint32_t), hipMemcpyDeviceToDevice, g.stream));
            }
            for (uint32_t lazy_round = 0;; lazy_round++) {
            if (ld.lazy && lazy_round > 0) {
                HIP_TRY(hipMemcpyAsync(rng_state, rng_saved, 2 * (size_t)n_tiles * sizeof(uint64_t), hipMemcpyDeviceToDevice, g.stream));
                HIP_TRY(hipMemcpyAsync(trunc_d, trunc_d + 1, sizeof(uint32_t), hipMemcpyDeviceToDevice, g.stream));
            }
            ev_open(2, 0);
#define RSPT_TS(I, A, O) hipLaunchKernelGGL((k_tile_serial<I, A, O>), grid, dim3(64), 0, g.stream, s->dev, s->tex, ld, rd, g.pb, pd, tiles_d, n_tiles, lanes, r0, r1, samp_L, samp_pf, serial_iters, trunc_d)
            if (s->has_animated) {   // (round 6: modes 5 - 8 = path / ao / volpath / directlighting with the instances' Transforms interpolated at the camera sample's time)
                if (ao) { if (s->has_alpha) RSPT_TS(true, true, 6); else RSPT_TS(true, false, 6); }
                else if (volpath) { if (s->has_alpha) RSPT_TS(true, true, 7); else RSPT_TS(true, false, 7); }
                else if (direct) { if (s->has_alpha) RSPT_TS(true, true, 8); else RSPT_TS(true, false, 8); }
                else { if (s->has_alpha) RSPT_TS(true, true, 5); else RSPT_TS(true, false, 5); }
            } else if (ao) {
                if (s->has_instances) { if (s->has_alpha) RSPT_TS(true, true, 1); else RSPT_TS(true, false, 1); }
                else { if (s->has_alpha) RSPT_TS(false, true, 1); else RSPT_TS(false, false, 1); }
            } else if (volpath) {
                if (s->has_instances) { if (s->has_alpha) RSPT_TS(true, true, 2); else RSPT_TS(true, false, 2); }
                else { if (s->has_alpha) RSPT_TS(false, true, 2); else RSPT_TS(false, false, 2); }
            } else if (direct) {
                if (s->has_instances) { if (s->has_alpha) RSPT_TS(true, true, 3); else RSPT_TS(true, false, 3); }
                else { if (s->has_alpha) RSPT_TS(false, true, 3); else RSPT_TS(false, false, 3); }
            } else if (s->has_dynamic) {
                if (s->has_instances) { if (s->has_alpha) RSPT_TS(true, true, 4); else RSPT_TS(true, false, 4); }
                else { if (s->has_alpha) RSPT_TS(false, true, 4); else RSPT_TS(false, false, 4); }
            } else if (s->has_instances) { if (s->has_alpha) RSPT_TS(true, true, 0); else RSPT_TS(true, false, 0); }
            else { if (s->has_alpha) RSPT_TS(false, true, 0); else RSPT_TS(false, false, 0); }
#undef RSPT_TS
            ev_close(2, 0);
            if (!ld.lazy) break;
            uint32_t claimed = 0;
            if ((rc = build_claimed_rows(&claimed))) return rc;
            if (claimed == 0) break;
            if (lazy_round > 64) return fail(RSPT_E_UNSUPPORTED, "pixel sampler: on-demand light voxels did not settle in 64 rounds (not a device fault: the caller keeps its CPU loop, or asks for the eager table)");
            }
            const uint32_t npx = (uint32_t)pl.size();
            Batch bt{0u, npx, 0u, spp, npx * spp};
            samples += bt.n;
            if ((rc = film_index(pass_pix, npx))) return rc;
            film_stage(rd, bt, fpb, pass_pix);
        }
        uint32_t tv = 0;
        HIP_TRY(hipMemcpyAsync(&tv, trunc_d, sizeof tv, hipMemcpyDeviceToHost, g.stream));
        HIP_TRY(hipStreamSynchronize(g.stream));
        truncated += tv;
        return RSPT_OK;
    };
    if (pixel_sampler) {
        size_t my_tiles = 0;
        for (size_t i = 0; i < blocks.size(); i++) my_tiles += (i / chunk) % shard_count == d->shard_index;
        const size_t min_tiles = env_size("RSPT_SERIAL_MIN_TILES", 2048);   // (round 4, statue frame under 02sequence, GPU vs 256 host threads, Msamples/s: 920 tiles 5.6 / 8.9, 2040: 11.2 / 8.0, 4080: 19.3 / 7.8, 8160: 26.7 / 7.3)
        if (!d->allow_slow_paths && my_tiles < min_tiles)
            return fail(RSPT_E_UNSUPPORTED, "a pixel sampler over %zu tiles: one lane per tile is slower than the host's tile loop below ~%zu tiles (set allow_slow_paths to run it anyway)", my_tiles, min_tiles);
    }
    if (pixel_sampler && (rc = run_tile_serial())) return rc;
    if (!pixel_sampler && (rc = film_index(g.pix_list, (uint32_t)n_pix))) return rc;
    for (size_t p0 = 0; !pixel_sampler && p0 < n_pix; p0 += pix_per_batch) {
        const uint32_t npx = (uint32_t)std::min(pix_per_batch, n_pix - p0);
        for (uint32_t s0 = (uint32_t)smp_begin; s0 < (uint32_t)smp_end; s0 += ns) {
            const uint32_t ns_b = std::min(ns, (uint32_t)smp_end - s0);  // Halton spp need not be a power of two
            Batch bt{(uint32_t)p0, npx, s0, ns_b, npx * ns_b};
            samples += bt.n;
            HIP_TRY(hipMemsetAsync(g.cnt, 0, (size_t)g.n_cnt * sizeof(QueueCounts), g.stream));
            if (shade_bins) HIP_TRY(hipMemsetAsync(g.bin_info, 0, (size_t)std::min<uint32_t>(g.n_bin_info, max_iters + 10) * sizeof(BinInfo), g.stream));
            // the path integrator's first shade launch knows what k_raygen would have written into L_eta / beta (PathBuf::fresh); RSPT_FRESH=0: written and read as before
            const bool fresh_ok = !volpath && !direct && !ao && env_size("RSPT_FRESH", 1) != 0;
            g.pb.fresh = fresh_ok ? 1u : 0u;
            hipLaunchKernelGGL(k_raygen, dim3((bt.n + 255) / 256), dim3(256), 0, g.stream, rd, bt, g.pb, g.pix_list, g.q[0][0], g.q[0][1], g.cnt);   // (MOVE or not: iteration 0 lives in set 0 = pb's own arrays, by slot)
            uint32_t it = 0;
            g_queue_hint = 0xffffffffu;
            if (volpath) rc = batch_volpath(bt, it);
            else if (direct) {
                rc = dl_lane ? batch_direct_lane(bt, it) : batch_direct(bt, it);
                if (rc == RSPT_DL_RETRY_LANE) {   // from here on the per-lane form serves this render; this batch starts over
                    dl_lane = true;
                    HIP_TRY(hipMemsetAsync(g.cnt, 0, (size_t)g.n_cnt * sizeof(QueueCounts), g.stream));
                    g.pb.fresh = 0u;
                    hipLaunchKernelGGL(k_raygen, dim3((bt.n + 255) / 256), dim3(256), 0, g.stream, rd, bt, g.pb, g.pix_list, g.q[0][0], g.q[0][1], g.cnt);
                    rc = batch_direct_lane(bt, it);
                }
            }
            else if (ao) rc = batch_ao(bt, it);
            else rc = batch_path(bt, it);
            if (rc) return rc;
            if (getenv("RSPT_QUEUE_LOG") && p0 == 0 && s0 == (uint32_t)smp_begin) {   // the queue lengths of the render's first batch, per wavefront iteration (profiles/rNN_shade_ledger.md)
                std::vector<QueueCounts> qc(it + 1);
                HIP_TRY(hipMemcpyAsync(qc.data(), g.cnt, (it + 1) * sizeof(QueueCounts), hipMemcpyDeviceToHost, g.stream));
                HIP_TRY(hipStreamSynchronize(g.stream));
                for (uint32_t k = 0; k <= it; k++)
                    fprintf(stderr, "rspt: queue it %u: active %u (+ %u that only wait for an estimate) closest %u any %u of %u paths\n", k, qc[k].active, qc[k].active_tail, qc[k].closest, qc[k].any, bt.n);
            }
            if (getenv("RSPT_QUEUE_LOG") && p0 == 0 && s0 == (uint32_t)smp_begin) {   // the queue lengths of the render's first batch, per wavefront iteration (profiles/rNN_shade_ledger.md)
                std::vector<QueueCounts> qc(it + 1);
                HIP_TRY(hipMemcpyAsync(qc.data(), g.cnt, (it + 1) * sizeof(QueueCounts), hipMemcpyDeviceToHost, g.stream));
                HIP_TRY(hipStreamSynchronize(g.stream));
                for (uint32_t k = 0; k <= it; k++)
                    fprintf(stderr, "rspt: queue it %u: active %u (+ %u that only wait for an estimate) closest %u any %u of %u paths\n", k, qc[k].active, qc[k].active_tail, qc[k].closest, qc[k].any, bt.n);
            }
            if (counters) hipLaunchKernelGGL(k_accum_counts, dim3(1), dim3(1), 0, g.stream, g.cnt, it, g.totals);
            film_stage(rd, bt, g.pb, g.pix_list);   // (MOVE: ended paths have written their radiance to pb.L_eta by original slot, move_pathbuf)
        }
    }
    g_queue_hint = 0xffffffffu;
    float4* out_dev = film_dev ? (float4*)film_dev : g.film_out;
    hipLaunchKernelGGL(k_film_resolve, dim3((uint32_t)((film_px + 255) / 256)), dim3(256), 0, g.stream, g.film_own, g.film_splat, out_dev, (uint32_t)film_px);
    if (d->film_reduce) {  // X1: sum of the ranks' films onto rank 0 (a sum, not a gather: tile pixel bounds overlap, film.rs:321-330)
        if (!rc_.comm) return fail(RSPT_E_INVALID, "film_reduce without a communicator (rspt_comm_init)");
        if ((uint32_t)rc_.world != shard_count || (uint32_t)rc_.rank != d->shard_index)
            return fail(RSPT_E_INVALID, "film_reduce: shard %u of %u does not match rank %d of %d", d->shard_index, shard_count, rc_.rank, rc_.world);
        hipError_t pe = hipStreamSynchronize(g.stream);   // a kernel fault of this rank must surface before the agreement, not inside the collective
        if (pe != hipSuccess) return fail(RSPT_E_HIP, "render failed: %s", hipGetErrorString(pe));
        int32_t any_failed = 0;
        if (int rc = film_reduce_agree(0, &any_failed)) return rc;
        if (any_failed) return fail(RSPT_E_PEER, "film_reduce: another rank of the communicator failed its render; no film was summed");
        RCCL_TRY(rc_.Reduce(out_dev, out_dev, film_px * 4, ncclFloat32, ncclSum, 0, rc_.comm, g.stream));
    }
    HIP_TRY(hipEventRecord(ev_k1, g.stream));
    HIP_TRY(hipGetLastError());
    if (film_host) HIP_TRY(hipMemcpyAsync(film_host, out_dev, film_px * sizeof(float4), hipMemcpyDeviceToHost, g.stream));
    if (li_host) HIP_TRY(hipMemcpyAsync(li_host, li_dev, film_px * (size_t)d->spp * 3 * sizeof(float), hipMemcpyDeviceToHost, g.stream));
    hipError_t se = hipStreamSynchronize(g.stream);
    if (se != hipSuccess) return fail(RSPT_E_HIP, "render failed: %s", hipGetErrorString(se));
    if (ld_lazy) {
        LightLazy lz;
        HIP_TRY(hipMemcpy(&lz, ld_lazy->lazy, sizeof lz, hipMemcpyDeviceToHost));
        if (lz.overflow) {
            LightLazy reset{0u, lz.n_rows, lz.max_rows, 0u};
            (void)hipMemcpy(ld_lazy->lazy, &reset, sizeof reset, hipMemcpyHostToDevice);
            return fail(RSPT_E_NOMEM, "spatial light distribution: more than %u voxels were touched; raise RSPT_LIGHT_TABLE_POOL_BYTES", lz.max_rows);
        }
    }
    auto t_end = std::chrono::steady_clock::now();
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->t_render_s = std::chrono::duration<double>(t_end - t_start).count();
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, ev_k0, ev_k1));
        stats->t_kernels_s = ms * 1e-3;
        double tr = 0;
        for (auto& e : trace_ev) {
            float m = 0;
            if (hipEventElapsedTime(&m, e.first, e.second) == hipSuccess) tr += m;
        }
        stats->t_trace_s = tr * 1e-3;
        double ksum[3] = {0, 0, 0};
        for (int k = 0; k < 3; k++)
            for (auto& e : kev[k]) {
                float m = 0;
                if (hipEventElapsedTime(&m, e.first, e.second) == hipSuccess) ksum[k] += m;
            }
        stats->t_trace_closest_s = ksum[0] * 1e-3; stats->t_trace_any_s = ksum[1] * 1e-3; stats->t_shade_s = ksum[2] * 1e-3;
        stats->launches_closest = kev[0].size(); stats->launches_any = kev[1].size();
        if (volpath) {  // paths vol.h cut because the sampler ran out of dimensions
            uint32_t tv = 0;
            HIP_TRY(hipMemcpy(&tv, g.vol.truncated, sizeof tv, hipMemcpyDeviceToHost));
            truncated += tv;
        }
        stats->truncated_paths = truncated;
        stats->samples = samples;
        stats->trace_launches = trace_launches;
        unsigned long long tot[8];
        HIP_TRY(hipMemcpy(tot, g.totals, sizeof tot, hipMemcpyDeviceToHost));
        stats->nan_samples = tot[5];
        if (volpath) { tot[3] = vol_rays; tot[4] = 0; }  // every volpath ray is a closest-hit ray (shadow rays walk segment by segment); counted from the queue lengths the host reads
        if (counters) {
            stats->nodes_visited = tot[0]; stats->tris_tested = tot[1];
            stats->rays_closest = tot[3]; stats->rays_any = tot[4];
            // B_alg = sum(32 N_node + 48 N_tri) + 96 R_closest + 72 R_any + 96 N_bounce + 32 per sample
            stats->alg_bytes = 32.0 * (double)tot[0] + 48.0 * (double)tot[1] + 96.0 * (double)tot[3] + 72.0 * (double)tot[4] + 96.0 * (double)tot[2] + 32.0 * (double)samples;
        }
    }
    return RSPT_OK;
}

}  // namespace

extern "C" {

int rspt_abi_version(void) { return RSPT_ABI_VERSION; }
// Replaces: BVHAccel::new (bvh.rs:96-392), on the device; see bvh_device.h
int64_t rspt_bvh_build_gpu(const float* P, uint64_t n_vertices, const uint32_t* tri_idx, uint64_t n_tris, uint32_t max_prims_in_node,
                           rspt_bvh_node* nodes_out, uint64_t nodes_cap, uint32_t* ordered_out) {
    using namespace rspt::bvhdev;
    if (!g.inited) return fail(RSPT_E_NODEVICE, "rspt_init has not been called");
    if (n_tris == 0) return 0;
    if (!P || !tri_idx || !nodes_out || !ordered_out) return fail(RSPT_E_INVALID, "null argument");
    if (n_tris > 0x3fffffffull) return fail(RSPT_E_UNSUPPORTED, "too many triangles");
    for (uint64_t i = 0; i < 3 * n_tris; i++)
        if (tri_idx[i] >= n_vertices) return fail(RSPT_E_INVALID, "vertex index out of range");
    HIP_TRY(hipSetDevice(g.device));
    const uint32_t n = (uint32_t)n_tris;
    const uint32_t max_prims = max_prims_in_node < 255 ? max_prims_in_node : 255;  // bvh.rs:102
    std::vector<void*> allocs;
    auto cleanup = [&]() { for (void* p : allocs) (void)hipFree(p); };
    auto dalloc = [&](size_t bytes) -> void* { void* p = nullptr; if (hipMalloc(&p, std::max<size_t>(bytes, 16)) != hipSuccess) return nullptr; allocs.push_back(p); return p; };
#define BVD_ALLOC(var, type, count) type* var = (type*)dalloc(sizeof(type) * (size_t)(count)); if (!var) { cleanup(); return fail(RSPT_E_NOMEM, "hipMalloc failed in rspt_bvh_build_gpu"); }
    BVD_ALLOC(P_d, float, 3 * n_vertices)
    BVD_ALLOC(tri_d, uint32_t, 3 * (size_t)n)
    Prims pr[2];
    for (int s2 = 0; s2 < 2; s2++) {
        for (int a = 0; a < 3; a++) {
            BVD_ALLOC(lo, float, n) BVD_ALLOC(hi, float, n) BVD_ALLOC(cc, float, n)
            pr[s2].lo[a] = lo; pr[s2].hi[a] = hi; pr[s2].c[a] = cc;
        }
        BVD_ALLOC(pid, uint32_t, n) BVD_ALLOC(pnode, uint32_t, n)
        pr[s2].prim = pid; pr[s2].node = pnode;
    }
    BVD_ALLOC(nodes, Node, 2 * (size_t)n)
    BVD_ALLOC(level_nodes, uint32_t, 2 * (size_t)n + 2)
    BVD_ALLOC(buckets, Buckets, (size_t)n / 3 + 2)
    BVD_ALLOC(flag, uint32_t, n)
    BVD_ALLOC(scan, uint32_t, n)
    const uint32_t n_scan_blocks = (n + BVD_SCAN_BLOCK * BVD_SCAN_ITEMS - 1) / (BVD_SCAN_BLOCK * BVD_SCAN_ITEMS);
    BVD_ALLOC(block_sums, uint32_t, n_scan_blocks + 1)
    BVD_ALLOC(counters, uint32_t, 8)  // [0] node count, [1] next level size, [2] scan total
    BVD_ALLOC(ordered_d, uint32_t, n)
    BVD_ALLOC(out_d, rspt_bvh_node, 2 * (size_t)n)
#undef BVD_ALLOC
    hipStream_t st = g.stream;
    auto bail = [&](hipError_t e, const char* what) { cleanup(); return (int64_t)fail(RSPT_E_HIP, "%s: %s", what, hipGetErrorString(e)); };
    hipError_t e;
    if ((e = hipMemcpyAsync(P_d, P, sizeof(float) * 3 * n_vertices, hipMemcpyHostToDevice, st)) != hipSuccess) return bail(e, "upload P");
    if ((e = hipMemcpyAsync(tri_d, tri_idx, sizeof(uint32_t) * 3 * (size_t)n, hipMemcpyHostToDevice, st)) != hipSuccess) return bail(e, "upload indices");
    const uint32_t gp = (n + 255) / 256;
    hipLaunchKernelGGL(k_prim_info, dim3(gp), dim3(256), 0, st, P_d, tri_d, n, pr[0]);
    Node root{};
    root.start = 0; root.end = n; root.base = 0; root.child0 = root.child1 = BVD_NONE;
    const uint32_t zero = 0, one = 1;
    (void)hipMemcpyAsync(nodes, &root, sizeof root, hipMemcpyHostToDevice, st);
    (void)hipMemcpyAsync(level_nodes, &zero, sizeof zero, hipMemcpyHostToDevice, st);
    (void)hipMemcpyAsync(&counters[0], &one, sizeof one, hipMemcpyHostToDevice, st);
    std::vector<std::pair<uint32_t, uint32_t>> levels;  // (offset into level_nodes, count)
    uint32_t off = 0, n_level = 1;
    int cur = 0;
    while (n_level > 0) {
        if (levels.size() > 4096) { cleanup(); return fail(RSPT_E_UNSUPPORTED, "BVH deeper than 4096 levels"); }
        levels.push_back({off, n_level});
        const uint32_t* lv = level_nodes + off;
        const uint32_t gl = (n_level + 255) / 256;
        hipLaunchKernelGGL(k_node_reset, dim3(gl), dim3(256), 0, st, nodes, lv, n_level);
        hipLaunchKernelGGL(k_bounds, dim3(gp), dim3(256), 0, st, pr[cur], n, nodes);
        (void)hipMemsetAsync(&counters[3], 0, sizeof(uint32_t), st);
        hipLaunchKernelGGL(k_node_axis, dim3(gl), dim3(256), 0, st, nodes, lv, n_level, buckets, pr[cur], &counters[3]);
        hipLaunchKernelGGL(k_buckets, dim3(gp), dim3(256), 0, st, pr[cur], n, nodes, buckets);
        hipLaunchKernelGGL(k_split, dim3(gl), dim3(256), 0, st, nodes, lv, n_level, buckets, max_prims);
        hipLaunchKernelGGL(k_flags, dim3(gp), dim3(256), 0, st, pr[cur], n, nodes, flag, ordered_d);
        hipLaunchKernelGGL(k_scan_blocks, dim3(n_scan_blocks), dim3(BVD_SCAN_BLOCK), 0, st, flag, scan, block_sums, n);
        hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, st, block_sums, n_scan_blocks, &counters[2]);
        hipLaunchKernelGGL(k_scan_add, dim3(gp), dim3(256), 0, st, scan, block_sums, n);
        (void)hipMemsetAsync(&counters[1], 0, sizeof(uint32_t), st);
        uint32_t* next = level_nodes + off + n_level;
        hipLaunchKernelGGL(k_children, dim3(gl), dim3(256), 0, st, nodes, lv, n_level, scan, &counters[2], n, &counters[0], next, &counters[1]);
        hipLaunchKernelGGL(k_scatter, dim3(gp), dim3(256), 0, st, pr[cur], pr[cur ^ 1], n, nodes, flag, scan);
        uint32_t n_next = 0;
        if ((e = hipMemcpyAsync(&n_next, &counters[1], sizeof n_next, hipMemcpyDeviceToHost, st)) != hipSuccess || (e = hipStreamSynchronize(st)) != hipSuccess) return bail(e, "bvh level");
        off += n_level;
        n_level = n_next;
        cur ^= 1;
    }
    uint32_t total = 0;
    if ((e = hipMemcpy(&total, &counters[0], sizeof total, hipMemcpyDeviceToHost)) != hipSuccess) return bail(e, "node count");
    if (total > nodes_cap) { cleanup(); return fail(RSPT_E_INVALID, "nodes_cap %llu < %u nodes", (unsigned long long)nodes_cap, total); }
    for (size_t l = levels.size(); l-- > 0;) hipLaunchKernelGGL(k_sizes, dim3((levels[l].second + 255) / 256), dim3(256), 0, st, nodes, level_nodes + levels[l].first, levels[l].second);
    for (size_t l = 0; l < levels.size(); l++) hipLaunchKernelGGL(k_indices, dim3((levels[l].second + 255) / 256), dim3(256), 0, st, nodes, level_nodes + levels[l].first, levels[l].second, out_d);
    if ((e = hipMemcpyAsync(nodes_out, out_d, sizeof(rspt_bvh_node) * total, hipMemcpyDeviceToHost, st)) != hipSuccess) return bail(e, "download nodes");
    if ((e = hipMemcpyAsync(ordered_out, ordered_d, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, st)) != hipSuccess) return bail(e, "download order");
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return bail(e, "bvh build");
    cleanup();
    return (int64_t)total;
}

const char* rspt_last_error(void) { return g_err.c_str(); }

int rspt_init(int32_t device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(RSPT_E_NODEVICE, "no HIP device: %s", e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device < 0 || device >= n) return fail(RSPT_E_INVALID, "device %d out of range (have %d)", device, n);
    if (g.inited && g.device == device) return RSPT_OK;
    if (g.inited) rspt_shutdown();
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0 && !getenv("RSPT_ALLOW_ANY_ARCH"))
        return fail(RSPT_E_NODEVICE, "device %d is %s; librspt is built for gfx950 only", device, prop.gcnArchName);
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&g.stream2, hipStreamNonBlocking));
    HIP_TRY(hipHostMalloc((void**)&g.look, 8 * sizeof(QueueCounts), hipHostMallocDefault));
    g.device = device;
    g.n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    g.inited = true;
    return RSPT_OK;
}

void rspt_shutdown(void) {
    if (!g.inited) return;
    (void)rspt_comm_destroy();
    (void)hipSetDevice(g.device);
    (void)hipStreamSynchronize(g.stream);
    free_paths();
    void* ptrs[] = {g.dl.le_kind, g.dl.w_r, g.dl.w_t, g.dl.l_all, g.dl.ld_acc, g.dl.dim, g.dl.kidx, g.dl.nflags, g.dl.error, g.dl_queue, g.bin_keys, g.q_sorted, g.bin_info, g.hit_inst, g.cnt, g.ovf, g.spill, g.totals, g.sobol32, g.vdc, g.vdc_inv, g.filter_table, g.film_own, g.film_splat, g.film_out, g.pix_list, g.pix_index, g.primes, g.prime_sums, g.halton_perms, g.cam_anim};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    for (hipEvent_t e : g.events) (void)hipEventDestroy(e);
    if (g.look) (void)hipHostFree(g.look);
    (void)hipStreamDestroy(g.stream);
    (void)hipStreamDestroy(g.stream2);
    g = Ctx{};
}

int rspt_comm_unique_id(uint8_t id[RSPT_COMM_ID_BYTES]) {
    if (!id) return fail(RSPT_E_INVALID, "null argument");
    int rc = rccl_bind();
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == RSPT_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId u;
    RCCL_TRY(rc_.GetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return RSPT_OK;
}
const char* rspt_comm_library(void) {
    static std::string path;
    if (rccl_bind() != RSPT_OK) return nullptr;
    Dl_info info;
    if (path.empty()) path = (dladdr((void*)rc_.GetUniqueId, &info) && info.dli_fname) ? info.dli_fname : "(unknown)";
    return path.c_str();
}
int rspt_comm_init(int32_t rank, int32_t world, const uint8_t id[RSPT_COMM_ID_BYTES]) {
    if (!g.inited) return fail(RSPT_E_NODEVICE, "rspt_init has not been called");
    if (!id || world < 1 || rank < 0 || rank >= world) return fail(RSPT_E_INVALID, "bad rank / world / id");
    int rc = rccl_bind();
    if (rc) return rc;
    if (rc_.comm) { (void)rc_.CommDestroy(rc_.comm); rc_.comm = nullptr; }
    HIP_TRY(hipSetDevice(g.device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    RCCL_TRY(rc_.CommInitRank(&rc_.comm, world, u, rank));
    rc_.rank = rank; rc_.world = world;
    if (!rc_.status) HIP_TRY(hipMalloc((void**)&rc_.status, sizeof(int32_t)));   // the status agreement's words (film_reduce_agree)
    if (!rc_.status_host) HIP_TRY(hipHostMalloc((void**)&rc_.status_host, sizeof(int32_t), hipHostMallocDefault));
    return RSPT_OK;
}
int rspt_comm_destroy(void) {
    if (rc_.comm) {
        if (g.inited) { (void)hipSetDevice(g.device); (void)hipStreamSynchronize(g.stream); }
        (void)rc_.CommDestroy(rc_.comm);
        rc_.comm = nullptr;
    }
    rc_.rank = rc_.world = 0;
    return RSPT_OK;
}

int rspt_scene_create(const rspt_scene_desc* d, rspt_scene_t* out) {
    if (!g.inited) return fail(RSPT_E_NODEVICE, "rspt_init has not been called");
    if (!d || !out) return fail(RSPT_E_INVALID, "null argument");
    *out = nullptr;
    if (d->n_prims > 0x7fffffffull || d->n_nodes > 0x7fffffffull) return fail(RSPT_E_UNSUPPORTED, "more than 2^31 primitives / nodes");
    if ((d->n_nodes && !d->nodes) || (d->n_prims && (!d->prims || !d->meshes || !d->P)) || (d->n_materials && !d->materials) ||
        (d->n_lights && !d->lights))
        return fail(RSPT_E_INVALID, "null array with non-zero count");
    if ((d->n_nodes == 0) != (d->n_prims == 0)) return fail(RSPT_E_INVALID, "nodes and prims must both be empty or both non-empty");
    // ---- validate indices on the host (a bad scene must not fault the GPU) ----
    bool has_null = false;
    const bool instanced = d->n_instances > 0;
    const uint64_t n_top_prims = instanced ? d->n_top_prims : d->n_prims, n_top_nodes = instanced ? d->n_top_nodes : d->n_nodes;
    if (instanced) {  // objects + instances behind the top-level aggregate
        if (!d->instances || !d->objects || d->n_objects == 0) return fail(RSPT_E_INVALID, "instances without objects");
        if (n_top_prims > d->n_prims || n_top_nodes > d->n_nodes || n_top_nodes == 0) return fail(RSPT_E_INVALID, "bad top-level aggregate range");
        if (d->instancing_mode != RSPT_INSTANCING_REFERENCE && d->instancing_mode != RSPT_INSTANCING_FIXED) return fail(RSPT_E_INVALID, "bad instancing_mode");
        for (uint32_t i = 0; i < d->n_objects; i++) {
            const rspt_object& o = d->objects[i];
            if (o.n_prims == 0 || o.first_prim < n_top_prims || o.first_prim + o.n_prims > d->n_prims) return fail(RSPT_E_INVALID, "object %u: primitive range", i);
            if (o.n_nodes == 0 ? o.n_prims != 1 : (o.first_node < n_top_nodes || o.first_node + o.n_nodes > d->n_nodes)) return fail(RSPT_E_INVALID, "object %u: node range", i);
        }
        for (uint32_t i = 0; i < d->n_instances; i++) {
            const rspt_instance& in = d->instances[i];
            if (in.object >= d->n_objects) return fail(RSPT_E_INVALID, "instance %u: object index out of range", i);
            for (int k = 0; k < 2; k++) {   // (rows 3 other than (0 0 0 1) are served: InstDev::m3 / mi3; a weight of 0 is where the reference asserts, transform.rs:747)
                const float* m = k ? in.from_world : in.to_world;
                if (!std::isfinite(m[12]) || !std::isfinite(m[13]) || !std::isfinite(m[14]) || !std::isfinite(m[15])) return fail(RSPT_E_INVALID, "instance %u: non-finite transform", i);
                for (int j = 0; j < 12; j++) if (!(fabsf(m[j]) < RSPT_INF)) return fail(RSPT_E_INVALID, "instance %u: non-finite transform", i);
            }
        }
    }
    for (uint64_t i = 0; i < d->n_prims; i++) {
        const rspt_prim& p = d->prims[i];
        if (p.mesh == RSPT_MESH_INSTANCE) {
            if (!instanced || i >= n_top_prims || p.v[0] >= d->n_instances) return fail(RSPT_E_INVALID, "prim %llu: bad instance reference", (unsigned long long)i);
            continue;
        }
        if (i >= n_top_prims && p.area_light != -1) return fail(RSPT_E_UNSUPPORTED, "prim %llu: area lights are not supported with object instancing (api.rs:2899)", (unsigned long long)i);
        if (p.v[0] >= d->n_vertices || p.v[1] >= d->n_vertices || p.v[2] >= d->n_vertices) return fail(RSPT_E_INVALID, "prim %llu: vertex index out of range", (unsigned long long)i);
        if (p.mesh >= d->n_meshes) return fail(RSPT_E_INVALID, "prim %llu: mesh index out of range", (unsigned long long)i);
        if (p.material != 0xffffffffu && p.material >= d->n_materials) return fail(RSPT_E_INVALID, "prim %llu: material index out of range", (unsigned long long)i);
        if (p.area_light >= (int64_t)d->n_lights) return fail(RSPT_E_INVALID, "prim %llu: light index out of range", (unsigned long long)i);
        has_null |= p.material == 0xffffffffu;
    }
    bool any_alpha = false;
    for (uint32_t i = 0; i < d->n_meshes; i++) {
        const rspt_mesh& m = d->meshes[i];
        if (m.alpha_tex > d->n_textures || m.shadow_alpha_tex > d->n_textures) return fail(RSPT_E_INVALID, "mesh %u: alpha texture index out of range", i);
        any_alpha |= m.alpha_tex != 0 || m.shadow_alpha_tex != 0;
        if (m.medium_inside > d->n_media || m.medium_outside > d->n_media) return fail(RSPT_E_INVALID, "mesh %u: medium index out of range", i);
    }
    if (d->n_media && !d->media) return fail(RSPT_E_INVALID, "null media");
    for (uint32_t i = 0; i < d->n_media; i++) {
        const rspt_medium& m = d->media[i];
        if (m.kind != RSPT_MEDIUM_HOMOGENEOUS && m.kind != RSPT_MEDIUM_GRID) return fail(RSPT_E_UNSUPPORTED, "medium %u: kind %u (homogeneous and grid-density media)", i, m.kind);
        if (m.kind == RSPT_MEDIUM_GRID) {
            if (m.nx < 1 || m.ny < 1 || m.nz < 1 || (uint64_t)m.nx * m.ny * m.nz > (1ull << 31) || !m.density) return fail(RSPT_E_INVALID, "medium %u: bad density grid", i);
            const float* w = m.world_to_medium;
            if (w[12] != 0.0f || w[13] != 0.0f || w[14] != 0.0f || w[15] != 1.0f) return fail(RSPT_E_UNSUPPORTED, "medium %u: projective world_to_medium", i);
        }
        for (int c = 0; c < 3; c++)
            if (!(m.sigma_a[c] >= 0.0f) || !(m.sigma_s[c] >= 0.0f) || !std::isfinite(m.sigma_a[c]) || !std::isfinite(m.sigma_s[c])) return fail(RSPT_E_INVALID, "medium %u: bad sigma_a / sigma_s", i);
        if (!(m.g > -1.0f && m.g < 1.0f)) return fail(RSPT_E_INVALID, "medium %u: g outside (-1, 1)", i);
    }
    if (any_alpha)
        for (uint64_t i = 0; i < d->n_prims; i++) {
            const rspt_prim& p = d->prims[i];
            if (p.mesh != RSPT_MESH_INSTANCE && p.mesh < d->n_meshes && p.area_light >= 0 && (d->meshes[p.mesh].alpha_tex || d->meshes[p.mesh].shadow_alpha_tex))
                return fail(RSPT_E_UNSUPPORTED, "prim %llu: an emissive mesh with an alpha mask", (unsigned long long)i);
        }
    for (uint32_t i = 0; i < d->n_lights; i++) {
        if (d->lights[i].kind < RSPT_LIGHT_DIFFUSE_AREA || d->lights[i].kind > RSPT_LIGHT_INFINITE) return fail(RSPT_E_UNSUPPORTED, "light %u: unsupported kind %u", i, d->lights[i].kind);
        if (d->lights[i].kind == RSPT_LIGHT_DIFFUSE_AREA && (d->lights[i].prim >= n_top_prims || d->prims[d->lights[i].prim].mesh == RSPT_MESH_INSTANCE))
            return fail(RSPT_E_INVALID, "light %u: prim out of range", i);
        if (d->lights[i].kind == RSPT_LIGHT_INFINITE && d->lights[i].prim >= d->n_envmaps) return fail(RSPT_E_INVALID, "light %u: envmap index out of range", i);
    }
    // textures: constant / imagemap / scale; each material may bind at most RSPT_TEX_SLOTS distinct ones
    if ((d->n_textures && !d->textures) || (d->n_images && !d->images)) return fail(RSPT_E_INVALID, "null texture / image array");
    for (uint32_t i = 0; i < d->n_images; i++) {
        const rspt_image& im = d->images[i];
        auto pow2 = [](uint32_t v) { return v && !(v & (v - 1)); };
        if (!pow2(im.width) || !pow2(im.height) || im.width > 32768 || im.height > 32768) return fail(RSPT_E_INVALID, "image %u: sides must be powers of two <= 32768 (MipMap::new resamples first)", i);
        uint32_t nl = 1;
        for (uint32_t m = std::max(im.width, im.height); m > 1; m >>= 1) nl++;
        if (im.n_levels != nl || nl > 16) return fail(RSPT_E_INVALID, "image %u: n_levels %u, expected %u", i, im.n_levels, nl);
        if (!im.texels || (im.channels != 1 && im.channels != 3)) return fail(RSPT_E_INVALID, "image %u: null texels or channels not 1 / 3", i);
    }
    {   // texture graph: kinds, mappings, child indices, depth (dev_texture.h evaluates at most RSPT_TEX_MAX_DEPTH levels)
        std::vector<int> depth(d->n_textures, 0);  // 0 = not computed, -1 = on the current path (cycle)
        std::function<int(uint32_t)> depth_of = [&](uint32_t i) -> int {
            if (depth[i] > 0) return depth[i];
            if (depth[i] < 0) return 1000;
            depth[i] = -1;
            const rspt_texture& t = d->textures[i];
            int dmax = 0;
            const uint32_t kids[3] = {t.tex1, t.tex2, t.tex3};
            const int n_kids = t.kind == RSPT_TEX_MIX ? 3 : ((t.kind == RSPT_TEX_SCALE || t.kind == RSPT_TEX_CHECKERBOARD || t.kind == RSPT_TEX_DOTS) ? 2 : 0);
            for (int k = 0; k < n_kids; k++) dmax = std::max(dmax, kids[k] < d->n_textures ? depth_of(kids[k]) : 1000);
            return depth[i] = 1 + dmax;
        };
        for (uint32_t i = 0; i < d->n_textures; i++) {
            const rspt_texture& t = d->textures[i];
            if (t.kind < RSPT_TEX_CONSTANT || t.kind > RSPT_TEX_WRINKLED) return fail(RSPT_E_UNSUPPORTED, "texture %u: unsupported kind %u", i, t.kind);
            const bool map2d = t.kind == RSPT_TEX_IMAGE || t.kind == RSPT_TEX_CHECKERBOARD || t.kind == RSPT_TEX_DOTS;
            const bool map3d = t.kind == RSPT_TEX_FBM || t.kind == RSPT_TEX_MARBLE || t.kind == RSPT_TEX_WINDY || t.kind == RSPT_TEX_WRINKLED;
            if (map2d && (t.mapping < RSPT_MAP_UV || t.mapping > RSPT_MAP_CYLINDRICAL)) return fail(RSPT_E_UNSUPPORTED, "texture %u: 2-D mapping %u", i, t.mapping);
            if (map3d && t.mapping != RSPT_MAP_IDENTITY3D) return fail(RSPT_E_UNSUPPORTED, "texture %u: 3-D textures take the identity mapping", i);
            if (map3d && (t.octaves < 0 || t.octaves > 64)) return fail(RSPT_E_INVALID, "texture %u: octaves out of range", i);
            if (t.kind == RSPT_TEX_IMAGE) {
                if (t.image >= d->n_images) return fail(RSPT_E_INVALID, "texture %u: image index out of range", i);
                if (t.wrap > RSPT_WRAP_CLAMP) return fail(RSPT_E_INVALID, "texture %u: bad wrap mode", i);
            }
            const int dep = depth_of(i);
            if (dep >= 1000) return fail(RSPT_E_INVALID, "texture %u: child index out of range or cyclic graph", i);
            if (dep > RSPT_TEX_MAX_DEPTH) return fail(RSPT_E_UNSUPPORTED, "texture %u: graph deeper than %d levels", i, RSPT_TEX_MAX_DEPTH);
        }
    }
    // materials: Material::compute_scattering_functions restated on the host (material_assembly.h), once per allow_multiple_lobes
    std::vector<rspt_material> asm_mats[2];
    std::vector<rspt_bxdf> asm_bx[2];
    std::vector<rspt_mat::DynMaterial> asm_dyn[2];   // [material] when the variant has a dynamic material, else empty
    std::vector<uint8_t> asm_is_dyn[2];
    for (int v = 0; v < 2; v++) {
        rspt_mat::Assembler as(d, v == 0);
        rspt_mat::Lobes lb;
        asm_mats[v].resize(d->n_materials);
        asm_is_dyn[v].assign(d->n_materials, 0);
        for (uint32_t i = 0; i < d->n_materials; i++) {
            if (rspt_mat::Error e = as.assemble(i, &lb)) return fail(e.code, "%s", e.text.c_str());
            lb.mat.first_bxdf = (uint32_t)asm_bx[v].size();
            asm_mats[v][i] = lb.mat;
            asm_bx[v].insert(asm_bx[v].end(), lb.lobes.begin(), lb.lobes.end());
            if (lb.dynamic) {
                if (asm_dyn[v].empty()) asm_dyn[v].resize(d->n_materials);
                asm_dyn[v][i] = lb.dyn;
                asm_is_dyn[v][i] = 1;
            }
        }
    }
    uint32_t shade_features = 0;
    for (int v = 0; v < 2; v++)
        for (const rspt_bxdf& b : asm_bx[v]) {
            shade_features |= RSPT_SF_LOBE(b.type);
            if (b.fresnel == RSPT_FRESNEL_CONDUCTOR) shade_features |= SF_CONDUCTOR;
            if (b.has_sc) shade_features |= SF_SC;
            if (b.tex_r || b.tex_t || b.tex_ax || b.tex_ay) shade_features |= SF_TEX;
        }
    for (int v = 0; v < 2; v++)
        for (const rspt_material& m : asm_mats[v]) if (m.bump_tex) shade_features |= SF_TEX;
    uint32_t shade_classes = 0;
    {   // materials whose lobe lists have the same types in the same order (and the same textured-ness) cost a wave the same
        std::vector<std::string> seen;
        for (uint32_t i = 0; i < d->n_materials; i++) {
            const rspt_material& m = asm_mats[0][i];
            std::string sig = asm_is_dyn[0][i] ? "dyn" : "";
            for (uint32_t l = 0; l < m.n_bxdfs; l++) {
                const rspt_bxdf& b = asm_bx[0][m.first_bxdf + l];
                sig += (char)('a' + b.type); sig += (char)('0' + b.fresnel); sig += (b.tex_r || b.tex_t || b.tex_ax || b.tex_ay) ? 't' : 'c';
            }
            if (m.bump_tex) sig += 'B';
            if (std::find(seen.begin(), seen.end(), sig) == seen.end()) seen.push_back(sig);
        }
        shade_classes = (uint32_t)seen.size();
    }
    if (!asm_dyn[0].empty() || !asm_dyn[1].empty())  // a list built per hit may hold any lobe its material kind can push
        shade_features |= SF_DYNAMIC | SF_TEX | 0x3feu | SF_CONDUCTOR | SF_SC;
    for (uint32_t i = 0; i < d->n_lights; i++) {
        const uint32_t k = d->lights[i].kind;
        shade_features |= k == RSPT_LIGHT_DIFFUSE_AREA ? SF_L_AREA : (k == RSPT_LIGHT_POINT ? SF_L_POINT : (k == RSPT_LIGHT_SPOT ? SF_L_SPOT : (k == RSPT_LIGHT_DISTANT ? SF_L_DISTANT : SF_L_INFINITE)));
    }
    for (uint32_t i = 0; i < d->n_meshes; i++)
        if (d->meshes[i].has_n || d->meshes[i].has_s || d->meshes[i].has_uv) shade_features |= SF_VERTEX;
    if (instanced) shade_features |= SF_INST;
    if (has_null || (instanced && d->instancing_mode == RSPT_INSTANCING_REFERENCE)) shade_features |= SF_NULL;
    if (d->n_envmaps && !d->envmaps) return fail(RSPT_E_INVALID, "null envmaps");
    for (uint32_t i = 0; i < d->n_envmaps; i++) {
        const rspt_envmap& e = d->envmaps[i];
        auto pow2 = [](uint32_t v) { return v && !(v & (v - 1)); };
        if (!pow2(e.width) || !pow2(e.height) || e.width > 32768 || e.height > 32768) return fail(RSPT_E_INVALID, "envmap %u: sides must be powers of two <= 32768", i);
        uint32_t nl = 1;
        for (uint32_t m = std::max(e.width, e.height); m > 1; m >>= 1) nl++;
        if (e.n_levels != nl || nl > 16) return fail(RSPT_E_INVALID, "envmap %u: n_levels %u, expected %u", i, e.n_levels, nl);
        if (!e.texels || !e.dist_func || e.dist_nu == 0 || e.dist_nv == 0) return fail(RSPT_E_INVALID, "envmap %u: null data", i);
    }
    // BVH: child / leaf ranges in bounds, depth <= 64 (the reference's fixed traversal stack, bvh.rs:420); with instances the
    // object's traversal continues on the stack of the top-level one (kernels.h traverse), so the two depths add up
    {
        std::string err;
        auto tree_depth = [&](uint64_t root, uint64_t node_lo, uint64_t node_hi, uint64_t prim_lo, uint64_t prim_hi) -> int {
            std::vector<std::pair<uint32_t, uint32_t>> stack;  // node, depth
            stack.push_back({(uint32_t)root, 1u});
            uint64_t visited = 0;
            uint32_t deepest = 0;
            while (!stack.empty()) {
                auto [ni, depth] = stack.back();
                stack.pop_back();
                if (++visited > node_hi - node_lo) { err = "BVH is not a tree"; return -1; }
                deepest = std::max(deepest, depth);
                if (depth > 64) { err = "BVH deeper than the 64-entry traversal stack"; return -2; }
                const rspt_bvh_node& n = d->nodes[ni];
                if (n.n_prims > 0) {
                    if (n.offset < 0 || (uint64_t)n.offset < prim_lo || (uint64_t)n.offset + n.n_prims > prim_hi) { err = "node " + std::to_string(ni) + ": leaf range out of bounds"; return -1; }
                } else {
                    if (n.axis > 2 || n.offset <= (int64_t)ni || (uint64_t)n.offset >= node_hi || (uint64_t)ni + 1 >= node_hi) { err = "node " + std::to_string(ni) + ": bad children"; return -1; }
                    stack.push_back({(uint32_t)n.offset, depth + 1});
                    stack.push_back({ni + 1, depth + 1});
                }
            }
            return (int)deepest;
        };
        int top_depth = 0, obj_depth = 0;
        if (d->n_nodes) {
            top_depth = tree_depth(0, 0, n_top_nodes, 0, n_top_prims);
            if (top_depth < 0) return fail(top_depth == -2 ? RSPT_E_UNSUPPORTED : RSPT_E_INVALID, "%s", err.c_str());
        }
        for (uint32_t i = 0; instanced && i < d->n_objects; i++) {
            const rspt_object& o = d->objects[i];
            if (o.n_nodes == 0) continue;
            const int dep = tree_depth(o.first_node, o.first_node, o.first_node + o.n_nodes, o.first_prim, o.first_prim + o.n_prims);
            if (dep < 0) return fail(dep == -2 ? RSPT_E_UNSUPPORTED : RSPT_E_INVALID, "object %u: %s", i, err.c_str());
            obj_depth = std::max(obj_depth, dep);
        }
        if (top_depth + obj_depth > 64) return fail(RSPT_E_UNSUPPORTED, "top-level BVH (%d levels) + object BVH (%d levels) deeper than the 64-entry traversal stack", top_depth, obj_depth);
    }
    HIP_TRY(hipSetDevice(g.device));
    rspt_scene_s* s = new rspt_scene_s();
    s->dev.time_div = 1u;
    s->has_null_material = has_null || (instanced && d->instancing_mode == RSPT_INSTANCING_REFERENCE);  // instanced hits pass through like null surfaces (Q11)
    s->has_instances = instanced;
    s->has_alpha = any_alpha;
    s->n_materials = d->n_materials;
    s->shade_features = shade_features;
    s->shade_classes = shade_classes;
    auto bail = [&](int rc) {
        for (void* p : s->allocs) (void)hipFree(p);
        delete s;
        return rc;
    };
    int rc;
    const rspt_bvh_node* nodes_d = nullptr;
    const rspt_mesh* meshes_d = nullptr;
    const float* P_d = nullptr;
    if ((rc = upload(s, d->nodes, d->n_nodes, &nodes_d))) return bail(rc);
    if ((rc = upload(s, d->prims, d->n_prims, &s->dev.prims))) return bail(rc);
    if ((rc = upload(s, d->meshes, d->n_meshes, &meshes_d))) return bail(rc);
    s->dev.meshes = meshes_d;
    {   // media: a grid medium's density goes to the device, its record gets the device pointer and 1 / max(density) (GridDensityMedium::new, grid.rs:44-55)
        std::vector<rspt_medium> media(d->media, d->media + d->n_media);
        for (rspt_medium& m : media) {
            if (m.kind != RSPT_MEDIUM_GRID) continue;
            const size_t n = (size_t)m.nx * m.ny * m.nz;
            float max_density = 0.0f;
            for (size_t k = 0; k < n; k++) max_density = std::fmax(max_density, m.density[k]);   // f32::max
            const float inv_max = 1.0f / max_density;
            memcpy(&m.pad, &inv_max, sizeof inv_max);
            const float* dens_d = nullptr;
            if ((rc = upload(s, m.density, n, &dens_d))) return bail(rc);
            m.density = dens_d;
            s->dev.n_grid_media++;
        }
        if ((rc = upload(s, media.data(), media.size(), &s->dev.media))) return bail(rc);
    }
    s->dev.n_media = d->n_media;
    if ((rc = upload(s, d->P, d->n_vertices * 3, &P_d))) return bail(rc);
    if ((rc = upload(s, d->N, d->N ? d->n_vertices * 3 : 0, &s->dev.N))) return bail(rc);
    if ((rc = upload(s, d->S, d->S ? d->n_vertices * 3 : 0, &s->dev.S))) return bail(rc);
    if ((rc = upload(s, d->UV, d->UV ? d->n_vertices * 2 : 0, &s->dev.UV))) return bail(rc);
    {   // SceneDev::tri_nuv: every primitive's normals and uvs next to each other (80 B per primitive; skipped beyond 16 GB)
        bool any = false;
        for (uint32_t i = 0; i < d->n_meshes; i++) any = any || (d->meshes[i].has_n && d->N) || (d->meshes[i].has_uv && d->UV);
        if (any && d->n_prims && d->n_prims * 80ull <= env_size("RSPT_TRI_NUV_MAX_BYTES", (size_t)16 << 30) && env_size("RSPT_TRI_NUV", 1) != 0) {
            std::vector<float> nuv((size_t)d->n_prims * 20, 0.0f);
            for (uint64_t i = 0; i < d->n_prims; i++) {
                const rspt_prim& pr = d->prims[i];
                if (pr.mesh == RSPT_MESH_INSTANCE) continue;
                const rspt_mesh& me = d->meshes[pr.mesh];
                float* q = nuv.data() + 20 * i;
                for (int k = 0; k < 3; k++) {
                    if (me.has_n && d->N) for (int c = 0; c < 3; c++) q[3 * k + c] = d->N[3 * (size_t)pr.v[k] + c];
                    if (me.has_uv && d->UV) for (int c = 0; c < 2; c++) q[9 + 2 * k + c] = d->UV[2 * (size_t)pr.v[k] + c];
                }
            }
            const float* dev_nuv = nullptr;
            if ((rc = upload(s, nuv.data(), nuv.size(), &dev_nuv))) return bail(rc);
            s->dev.tri_nuv = reinterpret_cast<const float4*>(dev_nuv);
        }
    }
    {   // lobes: texture ids become per-material slot numbers (1 + slot) for k_texture / k_shade
        bool any = false;
        for (int v = 0; v < 2; v++) {
            std::vector<rspt_bxdf>& bx = asm_bx[v];
            std::vector<uint32_t> slots((size_t)d->n_materials * RSPT_TEX_SLOTS, 0xffffffffu);
            std::vector<uint8_t> mflags(d->n_materials, 0);
            bool any_v = false;
            for (uint32_t m = 0; m < d->n_materials; m++) {
                const rspt_material& mat = asm_mats[v][m];
                uint32_t* sl = slots.data() + (size_t)m * RSPT_TEX_SLOTS;
                uint32_t n_sl = 0;
                bool full = false;
                auto slot_of = [&](uint32_t tex_plus_1, uint32_t flags) -> uint32_t {
                    const uint32_t desc = (tex_plus_1 - 1u) | flags;
                    for (uint32_t k = 0; k < n_sl; k++) if (sl[k] == desc) return k + 1u;
                    if (n_sl == RSPT_TEX_SLOTS) { full = true; return 1u; }
                    sl[n_sl++] = desc;
                    return n_sl;
                };
                for (uint32_t l = 0; l < mat.n_bxdfs; l++) {
                    rspt_bxdf& b = bx[mat.first_bxdf + l];
                    const uint32_t nd = (b.remap & RSPT_LOBE_NODIFF) ? RSPT_SLOT_NODIFF : 0u;  // the m2 side of a mix (mixmat.rs:58-69)
                    const uint32_t aflags = RSPT_SLOT_ALPHA | ((b.remap & RSPT_LOBE_REMAP) ? RSPT_SLOT_REMAP : 0u) | nd;
                    if (b.tex_r) { b.tex_r = slot_of(b.tex_r, nd); mflags[m] |= RSPT_MAT_TEXTURED; }
                    if (b.tex_t) { b.tex_t = slot_of(b.tex_t, nd); mflags[m] |= RSPT_MAT_TEXTURED; }
                    if (b.tex_ax) { b.tex_ax = slot_of(b.tex_ax, aflags); mflags[m] |= RSPT_MAT_TEXTURED; }
                    if (b.tex_ay) { b.tex_ay = slot_of(b.tex_ay, aflags); mflags[m] |= RSPT_MAT_TEXTURED; }
                }
                if (full) return bail(fail(RSPT_E_UNSUPPORTED, "material %u binds more than %d distinct textures", m, RSPT_TEX_SLOTS));
                if (asm_is_dyn[v][m]) mflags[m] |= RSPT_MAT_DYNAMIC | RSPT_MAT_TEXTURED;
                if (mat.bump_tex) mflags[m] |= RSPT_MAT_BUMP;
                any_v |= mflags[m] != 0;
            }
            rspt_scene_s::MatSet& ms = s->mat_set[v];
            ms.textured = any_v;
            ms.dynamic = !asm_dyn[v].empty();
            if (ms.dynamic && (rc = upload(s, asm_dyn[v].data(), asm_dyn[v].size(), &ms.dyn))) return bail(rc);
            any |= any_v;
            if ((rc = upload(s, asm_mats[v].data(), asm_mats[v].size(), &ms.materials)) || (rc = upload(s, bx.data(), bx.size(), &ms.bxdfs)) ||
                (rc = upload(s, slots.data(), slots.size(), &ms.mat_slots)) || (rc = upload(s, mflags.data(), mflags.size(), &ms.mat_flags)))
                return bail(rc);
        }
        s->select_materials(true);
        if (any || any_alpha) {  // k_texture / per-path texture rows only when a material is textured; alpha masks just need the tables
            if ((rc = upload(s, d->textures, d->n_textures, &s->tex.textures))) return bail(rc);
            std::vector<ImageDev> imgs(d->n_images);
            std::vector<float> pool;  // every pyramid, back to back
            for (uint32_t i = 0; i < d->n_images; i++) {
                const rspt_image& im = d->images[i];
                ImageDev& o = imgs[i];
                o.width = im.width; o.height = im.height; o.n_levels = im.n_levels; o.channels = im.channels;
                size_t n_tex = 0;
                for (uint32_t l = 0, w = im.width, h = im.height; l < im.n_levels; l++, w = std::max(1u, w / 2), h = std::max(1u, h / 2)) {
                    o.level_offset[l] = (uint32_t)n_tex;
                    n_tex += (size_t)w * h;
                }
                o.texel_base = pool.size();
                s->image_base.push_back(o.texel_base);
                pool.insert(pool.end(), im.texels, im.texels + n_tex * im.channels);
            }
            if ((rc = upload(s, pool.data(), pool.size(), &s->tex.texel_pool))) return bail(rc);
            if ((rc = upload(s, imgs.data(), imgs.size(), &s->tex.images))) return bail(rc);
            float lut[RSPT_EWA_LUT];  // MipMap::new's EWA weights (mipmap.rs:186-192), host expf like the reference's f32::exp
            for (int i = 0; i < RSPT_EWA_LUT; i++) {
                const float alpha = 2.0f, r2 = (float)i / (float)(RSPT_EWA_LUT - 1);
                lut[i] = std::exp(-alpha * r2) - std::exp(-alpha);
            }
            if ((rc = upload(s, lut, (size_t)RSPT_EWA_LUT, &s->tex.ewa_lut))) return bail(rc);
        }
    }
    if ((rc = upload(s, d->lights, d->n_lights, &s->dev.lights))) return bail(rc);
    {   // Scene.infinite_lights (scene.rs:40-43) and their environment maps
        std::vector<uint32_t> inf;
        for (uint32_t i = 0; i < d->n_lights; i++)
            if (d->lights[i].kind == RSPT_LIGHT_INFINITE) inf.push_back(i);
        s->dev.n_infinite = (uint32_t)inf.size();
        if ((rc = upload(s, inf.data(), inf.size(), &s->dev.infinite_lights))) return bail(rc);
        std::vector<EnvMapDev> envs(d->n_envmaps);
        for (uint32_t i = 0; i < d->n_envmaps; i++) {
            const rspt_envmap& e = d->envmaps[i];
            EnvMapDev& m = envs[i];
            m.width = e.width; m.height = e.height; m.n_levels = e.n_levels; m.nu = e.dist_nu; m.nv = e.dist_nv;
            size_t n_tex = 0;
            for (uint32_t l = 0, w = e.width, h = e.height; l < e.n_levels; l++, w = std::max(1u, w / 2), h = std::max(1u, h / 2)) {
                m.level_offset[l] = (uint32_t)n_tex;
                n_tex += (size_t)w * h;
            }
            if ((rc = upload(s, e.texels, n_tex * 3, &m.texels))) return bail(rc);
            // Distribution2D::new (sampling.rs:156-170) = Distribution1D::new (:24-49) per row and for the marginal
            const uint32_t nu = e.dist_nu, nv = e.dist_nv;
            std::vector<float> cdf((size_t)nv * (nu + 1)), fint(nv), mcdf(nv + 1);
            auto dist1d = [](const float* f, uint32_t n, float* c) {
                c[0] = 0.0f;
                for (uint32_t k = 1; k <= n; k++) c[k] = c[k - 1] + f[k - 1] / (float)n;
                float fi = c[n];
                if (fi == 0.0f) for (uint32_t k = 1; k <= n; k++) c[k] = (float)k / (float)n;
                else for (uint32_t k = 1; k <= n; k++) c[k] /= fi;
                return fi;
            };
            for (uint32_t v = 0; v < nv; v++) fint[v] = dist1d(e.dist_func + (size_t)v * nu, nu, cdf.data() + (size_t)v * (nu + 1));
            m.marg_int = dist1d(fint.data(), nv, mcdf.data());
            if ((rc = upload(s, e.dist_func, (size_t)nu * nv, &m.cond_func)) || (rc = upload(s, cdf.data(), cdf.size(), &m.cond_cdf)) ||
                (rc = upload(s, fint.data(), fint.size(), &m.cond_int)) || (rc = upload(s, fint.data(), fint.size(), &m.marg_func)) ||
                (rc = upload(s, mcdf.data(), mcdf.size(), &m.marg_cdf)))
                return bail(rc);
        }
        if ((rc = upload(s, envs.data(), envs.size(), &s->dev.envmaps))) return bail(rc);
    }
    s->dev.nodes = reinterpret_cast<const float4*>(nodes_d);
    s->dev.n_nodes = (uint32_t)d->n_nodes; s->dev.n_prims = (uint32_t)d->n_prims; s->dev.n_lights = d->n_lights;
    if (d->n_nodes) {
        for (int i = 0; i < 3; i++) { s->dev.wb_min[i] = d->nodes[0].bmin[i]; s->dev.wb_max[i] = d->nodes[0].bmax[i]; }
    } else {
        for (int i = 0; i < 3; i++) { s->dev.wb_min[i] = RSPT_FLT_MAX; s->dev.wb_max[i] = -RSPT_FLT_MAX; }  // Bounds3f::default
    }
    // ---- four-box records (trace_w4.h): grandchildren of every interior node at even depth; with object instances every object's
    // aggregate gets its own records behind the top-level ones, and every instance primitive the reference to the rest of its leaf ----
    std::vector<uint32_t> obj_root(d->n_objects, RSPT_NONE), inst_cont_h(d->n_instances, RSPT_NONE);
    if (d->n_nodes > 0) {
        std::vector<uint2> big;
        auto range_ref = [&](uint32_t off, uint32_t cnt) -> uint32_t {
            if (cnt <= 15u && off <= RSPT_W4_OFFSET_MASK) return RSPT_REF_LEAF | ((cnt - 1u) << RSPT_W4_COUNT_SHIFT) | off;
            big.push_back(make_uint2(off, cnt));
            return RSPT_REF_LEAF | (15u << RSPT_W4_COUNT_SHIFT) | (uint32_t)(big.size() - 1);
        };
        auto leaf_ref = [&](uint32_t ni) -> uint32_t {
            const rspt_bvh_node& n = d->nodes[ni];
            if (instanced && ni < n_top_nodes)  // an instance in this leaf: the primitives behind it are reached again through its continuation reference
                for (uint32_t i = 0; i + 1u < n.n_prims; i++) {
                    const rspt_prim& p = d->prims[(uint32_t)n.offset + i];
                    if (p.mesh == RSPT_MESH_INSTANCE) inst_cont_h[p.v[0]] = range_ref((uint32_t)n.offset + i + 1u, n.n_prims - i - 1u);
                }
            return range_ref((uint32_t)n.offset, n.n_prims);
        };
        // the records of the tree rooted at LinearBVHNode `root` (an interior node), depth first, with record-local indices
        auto build_tree = [&](uint32_t root, std::vector<Wide4Node>& recs, std::vector<uint32_t>& rec_axes) {
            const float qnan = std::numeric_limits<float>::quiet_NaN();
            std::vector<std::pair<uint32_t, uint32_t>> todo;  // (LinearBVHNode index, record index)
            recs.clear(); rec_axes.assign(1, 0u);
            recs.emplace_back();
            todo.emplace_back(root, 0u);
            while (!todo.empty()) {
                const auto [ai, ri] = todo.back();
                todo.pop_back();
                const rspt_bvh_node& a = d->nodes[ai];
                uint32_t slot_node[4] = {RSPT_NONE, RSPT_NONE, RSPT_NONE, RSPT_NONE};
                uint32_t axes = a.axis;
                const uint32_t child[2] = {ai + 1u, (uint32_t)a.offset};
                for (int gi = 0; gi < 2; gi++) {
                    const rspt_bvh_node& c = d->nodes[child[gi]];
                    if (c.n_prims != 0) {
                        slot_node[2 * gi] = child[gi];
                    } else {
                        axes |= (uint32_t)c.axis << (2 + 2 * gi);
                        slot_node[2 * gi] = child[gi] + 1u;
                        slot_node[2 * gi + 1] = (uint32_t)c.offset;
                    }
                }
                Wide4Node w{};
                float lo[4][3], hi[4][3];
                uint32_t pending[4], n_pending = 0;
                for (int k = 0; k < 4; k++) {
                    if (slot_node[k] == RSPT_NONE) {
                        for (int c = 0; c < 3; c++) lo[k][c] = hi[k][c] = qnan;
                        w.ref[k] = RSPT_NONE;
                        continue;
                    }
                    const rspt_bvh_node& x = d->nodes[slot_node[k]];
                    for (int c = 0; c < 3; c++) { lo[k][c] = x.bmin[c]; hi[k][c] = x.bmax[c]; }
                    if (x.n_prims != 0) w.ref[k] = leaf_ref(slot_node[k]);
                    else pending[n_pending++] = (uint32_t)k;
                }
                // children records are numbered so that the first slot's subtree follows its parent (pushed last)
                for (uint32_t j = 0; j < n_pending; j++) { w.ref[pending[j]] = (uint32_t)recs.size(); recs.emplace_back(); rec_axes.push_back(0u); }
                for (uint32_t j = n_pending; j-- > 0;) todo.emplace_back(slot_node[pending[j]], w.ref[pending[j]]);
                for (int c = 0; c < 3; c++) {
                    w.b[c] = make_float4(lo[0][c], lo[1][c], hi[0][c], hi[1][c]);
                    w.b[3 + c] = make_float4(lo[2][c], lo[3][c], hi[2][c], hi[3][c]);
                }
                rec_axes[ri] = axes;
                recs[ri] = w;
            }
        };
        // final form of a record: child indices through `new_of` (+ base), the three axes in bits 25..26 of the first three refs (an
        // empty slot has a NaN box, its ref is never read as a ref)
        auto finish_rec = [&](Wide4Node w, uint32_t axes, const std::vector<uint32_t>* new_of, uint32_t base) {
            for (int k = 0; k < 4; k++)
                if (w.ref[k] != RSPT_NONE && !(w.ref[k] & RSPT_REF_LEAF)) w.ref[k] = (new_of ? (*new_of)[w.ref[k]] : w.ref[k]) + base;
            for (int k = 0; k < 3; k++) w.ref[k] = (w.ref[k] & ~RSPT_W4_AXIS_MASK) | (((axes >> (2 * k)) & 3u) << RSPT_W4_AXIS_SHIFT);
            return w;
        };
        std::vector<Wide4Node> recs, local;
        std::vector<uint32_t> axes;
        if (d->nodes[0].n_prims != 0) {
            s->w4_root = leaf_ref(0);
        } else {
            build_tree(0u, local, axes);
            // Renumber: a breadth-first prefix of RSPT_W4_TOP_MAX records goes first (k_trace_w4 keeps its first TOPCAP in LDS: any prefix of a
            // breadth-first order is one), the others keep their depth-first order.
            std::vector<uint32_t> new_of(local.size(), RSPT_NONE), order(1, 0u);
            for (size_t h = 0; h < order.size() && order.size() < RSPT_W4_TOP_MAX; h++)
                for (int k = 0; k < 4; k++) {
                    const uint32_t r = local[order[h]].ref[k];
                    if (r != RSPT_NONE && !(r & RSPT_REF_LEAF) && order.size() < RSPT_W4_TOP_MAX) order.push_back(r);
                }
            for (size_t i = 0; i < order.size(); i++) new_of[order[i]] = (uint32_t)i;
            uint32_t next_index = (uint32_t)order.size();
            for (size_t i = 0; i < local.size(); i++)
                if (new_of[i] == RSPT_NONE) new_of[i] = next_index++;
            recs.resize(local.size());
            for (size_t i = 0; i < local.size(); i++) recs[new_of[i]] = finish_rec(local[i], axes[i], &new_of, 0u);
            s->w4_top = (uint32_t)order.size();
            s->w4_root = 0u;
        }
        for (uint32_t o = 0; instanced && o < d->n_objects; o++) {
            const rspt_object& ob = d->objects[o];
            if (ob.n_nodes == 0) obj_root[o] = range_ref((uint32_t)ob.first_prim, 1u);                       // a lone primitive: no box, no aggregate
            else if (d->nodes[ob.first_node].n_prims != 0) obj_root[o] = leaf_ref((uint32_t)ob.first_node);  // a one-leaf aggregate
            else {
                build_tree((uint32_t)ob.first_node, local, axes);
                const uint32_t base = (uint32_t)recs.size();
                for (size_t i = 0; i < local.size(); i++) recs.push_back(finish_rec(local[i], axes[i], nullptr, base));
                obj_root[o] = base;
            }
        }
        // record indices and big-leaf indices must leave bits 25..30 free; larger scenes stay on the two-box kernel
        s->w4_ok = recs.size() <= RSPT_W4_OFFSET_MASK && big.size() <= RSPT_W4_OFFSET_MASK;
        if (s->w4_ok && !recs.empty() && (rc = upload(s, recs.data(), recs.size(), &s->w4))) return bail(rc);
        if (!big.empty() && (rc = upload(s, big.data(), big.size(), &s->big_leaves))) return bail(rc);
        // ---- the quantised form of the records (trace_w4q.h) for the shadow rays of scenes without instances and alpha masks ----
        bool any_alpha = false;
        for (uint32_t i = 0; i < d->n_meshes; i++) any_alpha = any_alpha || d->meshes[i].alpha_tex || d->meshes[i].shadow_alpha_tex;
        if (s->w4_ok && !recs.empty() && !instanced && !any_alpha && env_size("RSPT_W4Q_BUILD", 1) != 0) {
            std::vector<Quad4Node> q(recs.size());
            bool ok = true;
            for (size_t i = 0; i < recs.size() && ok; i++) {
                const Wide4Node& w = recs[i];
                Quad4Node& o = q[i];
                memset(&o, 0, sizeof o);
                memcpy(o.ref, w.ref, sizeof o.ref);
                uint32_t qlo[3] = {0, 0, 0}, qhi[3] = {0, 0, 0}, meta = 0;
                for (int c = 0; c < 3 && ok; c++) {
                    const float lo4[4] = {w.b[c].x, w.b[c].y, w.b[3 + c].x, w.b[3 + c].y}, hi4[4] = {w.b[c].z, w.b[c].w, w.b[3 + c].z, w.b[3 + c].w};
                    double org = 0.0, top = 0.0;
                    bool have = false;
                    for (int k = 0; k < 4; k++) {
                        if (lo4[k] != lo4[k]) { if (c == 0) meta |= 1u << (24 + k); continue; }   // an empty slot: a NaN box
                        org = have ? std::min(org, (double)lo4[k]) : (double)lo4[k];
                        top = have ? std::max(top, (double)hi4[k]) : (double)hi4[k];
                        have = true;
                    }
                    if (!have) { o.org[c] = 0.0f; meta |= 100u << (8 * c); continue; }
                    if (!std::isfinite(org) || !std::isfinite(top)) { ok = false; break; }
                    // cell = 2^(e - 127), the smallest power of two with 255 cells covering the extent (never below 2^-67, never above 2^60)
                    int e = 60;
                    if (top > org) { int ex = 0; (void)std::frexp((top - org) / 255.0, &ex); e = std::max(60, std::min(187, ex + 127 - 2)); }   // (start two below the answer: the loop then takes <= 3 steps)
                    while (e < 187 && std::ceil((top - org) / std::ldexp(1.0, e - 127)) > 255.0) e++;
                    if (std::ceil((top - org) / std::ldexp(1.0, e - 127)) > 255.0) { ok = false; break; }
                    const double cell = std::ldexp(1.0, e - 127);
                    o.org[c] = (float)org;   // (org is one of the f32 planes: exact)
                    meta |= (uint32_t)e << (8 * c);
                    for (int k = 0; k < 4; k++) {
                        if (lo4[k] != lo4[k]) { qlo[c] |= 255u << (8 * k); continue; }   // empty: lower plane above the upper one (and the empty bit)
                        const double fl = std::floor(((double)lo4[k] - org) / cell), ce = std::ceil(((double)hi4[k] - org) / cell);
                        const uint32_t a = (uint32_t)std::min(255.0, std::max(0.0, fl)), b = (uint32_t)std::min(255.0, std::max(0.0, ce));
                        if (org + a * cell > (double)lo4[k] || org + b * cell < (double)hi4[k]) { ok = false; break; }   // outward, in exact arithmetic
                        qlo[c] |= a << (8 * k); qhi[c] |= b << (8 * k);
                    }
                }
                o.meta = meta; o.qlo[0] = qlo[0]; o.qlo[1] = qlo[1]; o.qlo[2] = qlo[2]; o.qhi_x = qhi[0]; o.qhi_y = qhi[1]; o.qhi_z = qhi[2];
            }
            if (ok) {
                std::vector<float4> lb(2 * (size_t)d->n_prims, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
                for (uint64_t ni = 0; ni < d->n_nodes; ni++) {
                    const rspt_bvh_node& n = d->nodes[ni];
                    if (n.n_prims == 0) continue;
                    lb[2 * (size_t)n.offset] = make_float4(n.bmin[0], n.bmin[1], n.bmin[2], n.bmax[0]);   // the layout box_hit reads (sc.nodes)
                    lb[2 * (size_t)n.offset + 1] = make_float4(n.bmax[1], n.bmax[2], 0.0f, 0.0f);
                }
                if ((rc = upload(s, q.data(), q.size(), &s->w4q)) || (rc = upload(s, lb.data(), lb.size(), &s->leaf_boxes))) return bail(rc);
            }
        }
    }
    if (d->n_prims) {
        float4* tris = nullptr;
        hipError_t e = hipMalloc((void**)&tris, d->n_prims * 3 * sizeof(float4));
        if (e != hipSuccess) return bail(fail(RSPT_E_NOMEM, "triangle records: %s", hipGetErrorString(e)));
        s->allocs.push_back(tris);
        const uint32_t* inst_cont_d = nullptr;
        if ((rc = upload(s, inst_cont_h.data(), inst_cont_h.size(), &inst_cont_d))) return bail(rc);
        // alpha masks in the in-line form (dev_scene.h AlphaMask; kernels.h alpha_simple): possible when every mask of the scene is a ConstantTexture or an
        // ImageTexture under a UVMapping2D with finite parameters, and meshes with uvs have their per-primitive copies (tri_nuv)
        const uint32_t* mesh_mask_d = nullptr;
        if (s->has_alpha && env_size("RSPT_ALPHA_SIMPLE", 1) != 0) {
            std::vector<AlphaEntry> entries;
            std::vector<uint32_t> mesh_mask(d->n_meshes, 0u);
            std::map<std::pair<uint32_t, uint32_t>, uint32_t> seen;
            bool simple = s->image_base.size() == d->n_images;
            auto mask_of = [&](uint32_t tex_plus_1, AlphaMask* m) {
                memset(m, 0, sizeof *m);
                if (!tex_plus_1) return true;
                const rspt_texture& tx = d->textures[tex_plus_1 - 1u];
                if (tx.kind == RSPT_TEX_CONSTANT) { m->kind = 1u; m->value = tx.value[0]; return true; }
                if (tx.kind != RSPT_TEX_IMAGE || tx.mapping != RSPT_MAP_UV || tx.image >= d->n_images) return false;
                for (int k = 0; k < 4; k++) if (!std::isfinite(tx.map[k])) return false;   // (0 * su must be 0: the lookup's zero differentials)
                const rspt_image& im = d->images[tx.image];
                if (!im.n_levels || im.n_levels > 27u) return false;   // (level = n_levels - 1 + log2(1e-8) must be negative: mipmap.rs:236-239)
                m->kind = 2u; m->su = tx.map[0]; m->sv = tx.map[1]; m->du = tx.map[2]; m->dv = tx.map[3];
                m->width = im.width; m->height = im.height; m->wrap = tx.wrap; m->channels = im.channels; m->base = s->image_base[tx.image];
                return true;
            };
            for (uint32_t i = 0; i < d->n_meshes && simple; i++) {
                const rspt_mesh& me = d->meshes[i];
                if (!me.alpha_tex && !me.shadow_alpha_tex) continue;
                if (me.has_uv && d->UV && !s->dev.tri_nuv) { simple = false; break; }
                const auto key = std::make_pair(me.alpha_tex, me.shadow_alpha_tex);
                auto it = seen.find(key);
                if (it == seen.end()) {
                    AlphaEntry e;
                    if (!mask_of(me.alpha_tex, &e.alpha) || !mask_of(me.shadow_alpha_tex, &e.shadow) || entries.size() >= (1u << (32 - MF_MASK_SHIFT))) { simple = false; break; }
                    it = seen.emplace(key, (uint32_t)entries.size()).first;
                    entries.push_back(e);
                }
                mesh_mask[i] = it->second;
            }
            if (simple && !entries.empty()) {
                if ((rc = upload(s, entries.data(), entries.size(), &s->dev.alpha_masks)) || (rc = upload(s, mesh_mask.data(), mesh_mask.size(), &mesh_mask_d))) return bail(rc);
                s->alpha_simple = true;
            }
        }
        hipLaunchKernelGGL(k_build_tris, dim3((uint32_t)((d->n_prims + 255) / 256)), dim3(256), 0, g.stream, s->dev.prims, meshes_d, P_d, (uint32_t)d->n_prims, tris, inst_cont_d, mesh_mask_d);
        e = hipStreamSynchronize(g.stream);
        if (e != hipSuccess) return bail(fail(RSPT_E_HIP, "k_build_tris: %s", hipGetErrorString(e)));
        s->dev.tris = tris;
        if (s->w4_ok && s->w4 && !instanced && (!s->has_alpha || s->alpha_simple) && env_size("RSPT_SERIAL_W4", 1) != 0) {   // the per-lane kernels' traversal (trace_serial.h)
            s->dev.w4 = s->w4; s->dev.w4_big = s->big_leaves; s->dev.w4_root = s->w4_root;
        }
    }
    if (instanced) {  // InstDev records
        std::vector<InstDev> ins(d->n_instances);
        std::vector<InstAnim> anims;
        for (uint32_t i = 0; i < d->n_instances; i++) {
            const rspt_instance& in = d->instances[i];
            const rspt_object& o = d->objects[in.object];
            InstDev& x = ins[i];
            memset(&x, 0, sizeof x);
            memcpy(x.m, in.to_world, sizeof x.m);
            memcpy(x.mi, in.from_world, sizeof x.mi);
            memcpy(x.m3, in.to_world + 12, sizeof x.m3);
            memcpy(x.mi3, in.from_world + 12, sizeof x.mi3);
            x.root_node = o.n_nodes ? (uint32_t)o.first_node : RSPT_MISS;
            x.first_prim = (uint32_t)o.first_prim;
            x.w4_root = obj_root[in.object];
            bool ident = true;  // Transform::is_identity looks at m only (transform.rs:291-308)
            for (int r = 0; r < 4; r++)
                for (int c = 0; c < 4; c++) ident &= in.to_world[4 * r + c] == (r == c ? 1.0f : 0.0f);
            x.identity = ident ? 1u : 0u;
            x.anim = RSPT_MISS;
            if (in.animated) {   // a moving instance (ABI 20): AnimatedTransform::new's decomposition of the two keys, as for a moving camera
                if (!std::isfinite(in.time[0]) || !std::isfinite(in.time[1]) || !(in.time[1] > in.time[0]))
                    return bail(fail(RSPT_E_INVALID, "instance %u: animated with time[1] <= time[0] or a non-finite key time", i));
                for (int k = 0; k < 16; k++)
                    if (!std::isfinite(in.to_world_end[k]) || !std::isfinite(in.from_world_end[k]))
                        return bail(fail(RSPT_E_INVALID, "instance %u: non-finite end key matrix", i));
                // transform_point divides by the homogeneous weight and the reference asserts it is not zero (transform.rs:505): an end key whose
                // rows 3 are all zero can only produce such weights
                if (in.to_world_end[12] == 0.0f && in.to_world_end[13] == 0.0f && in.to_world_end[14] == 0.0f && in.to_world_end[15] == 0.0f)
                    return bail(fail(RSPT_E_INVALID, "instance %u: end key with a zero homogeneous row", i));
                if (in.from_world_end[12] == 0.0f && in.from_world_end[13] == 0.0f && in.from_world_end[14] == 0.0f && in.from_world_end[15] == 0.0f)
                    return bail(fail(RSPT_E_INVALID, "instance %u: end key inverse with a zero homogeneous row", i));
                InstAnim an;
                memset(&an, 0, sizeof an);
                if (camanim::camera_keys(in.to_world, in.time[0], in.to_world_end, in.time[1], &an.keys)) {   // false: equal keys = not actually animated
                    memcpy(an.mi_end, in.from_world_end, sizeof an.mi_end);
                    bool ie = true;
                    for (int r = 0; r < 4; r++)
                        for (int c = 0; c < 4; c++) ie &= in.to_world_end[4 * r + c] == (r == c ? 1.0f : 0.0f);
                    an.identity_end = ie ? 1u : 0u;
                    x.anim = (uint32_t)anims.size();
                    anims.push_back(an);
                }
            }
        }
        if (!anims.empty()) {
            if (s->has_alpha && !s->w4_ok) return bail(fail(RSPT_E_UNSUPPORTED, "moving object instances together with alpha-masked meshes in a scene whose records outgrow the four-box kernel"));
            if ((rc = upload(s, anims.data(), anims.size(), &s->dev.inst_anim))) return bail(rc);
            s->has_animated = true;
            s->shade_features |= SF_ANIM;
        }
        if ((rc = upload(s, ins.data(), ins.size(), &s->dev.inst))) return bail(rc);
        s->dev.n_inst = d->n_instances;
        s->dev.inst_fixed = d->instancing_mode == RSPT_INSTANCING_FIXED ? 1u : 0u;
    }
    if (d->n_nodes > 1 && !instanced) {  // pair records: both children's boxes next to each other (trace_wide.h)
        std::vector<uint32_t> pair_of(d->n_nodes, 0u);
        uint32_t n_pairs = 0;
        for (uint64_t i = 0; i < d->n_nodes; i++)
            if (d->nodes[i].n_prims == 0) pair_of[i] = n_pairs++;
        std::vector<PairNode> pairs(n_pairs);
        for (uint64_t i = 0; i < d->n_nodes; i++) {
            const rspt_bvh_node& n = d->nodes[i];
            if (n.n_prims != 0) continue;
            const uint32_t ci[2] = {(uint32_t)i + 1u, (uint32_t)n.offset};
            const rspt_bvh_node& a = d->nodes[ci[0]];
            const rspt_bvh_node& b = d->nodes[ci[1]];
            PairNode& p = pairs[pair_of[i]];
            p.q0 = make_float4(a.bmin[0], b.bmin[0], a.bmax[0], b.bmax[0]);
            p.q1 = make_float4(a.bmin[1], b.bmin[1], a.bmax[1], b.bmax[1]);
            p.q2 = make_float4(a.bmin[2], b.bmin[2], a.bmax[2], b.bmax[2]);
            p.c0 = a.n_prims ? (ci[0] | RSPT_REF_LEAF) : pair_of[ci[0]];
            p.c1 = b.n_prims ? (ci[1] | RSPT_REF_LEAF) : pair_of[ci[1]];
            p.self = (uint32_t)i;
            p.axis = n.axis;
        }
        if ((rc = upload(s, pairs.data(), pairs.size(), &s->pairs))) return bail(rc);
    }
    *out = s;
    return RSPT_OK;
}

int rspt_scene_destroy(rspt_scene_t s) {
    if (!s) return RSPT_OK;
    if (g.inited) { (void)hipSetDevice(g.device); (void)hipStreamSynchronize(g.stream); }
    for (void* p : s->allocs) (void)hipFree(p);
    delete s;
    return RSPT_OK;
}

namespace {
// A render that asked for the film reduce and failed before reaching it still takes part in the status agreement, so that the other
// ranks return RSPT_E_PEER instead of waiting for it; its own error code and message are kept.
int render_entry(rspt_scene_t s, const rspt_render_desc* d, float* film_host, void* film_dev, float* li_host, rspt_stats* stats) {
    rc_.status_exchanged = false;
    const int rc = (!film_host && !film_dev && !li_host) ? fail(RSPT_E_INVALID, "null output buffer") : render_impl(s, d, film_host, film_dev, li_host, stats);
    if (s && g.inited) { s->dev.ray_time = nullptr; s->dev.time_div = 1u; }   // the per-path times belong to the render that just ended (the buffer is the library's, not the scene's)
    if (rc != RSPT_OK && d && d->film_reduce && rc_.comm && g.inited && !rc_.status_exchanged) {
        const std::string kept = rspt_last_error();
        int32_t all = 0;
        (void)film_reduce_agree(1, &all);
        fail(rc, "%s", kept.c_str());
    }
    return rc;
}
}  // namespace

int rspt_render(rspt_scene_t s, const rspt_render_desc* d, float* film_xyzw, rspt_stats* stats) {
    return render_entry(s, d, film_xyzw, nullptr, nullptr, stats);
}
int rspt_render_device(rspt_scene_t s, const rspt_render_desc* d, void* film_dev, rspt_stats* stats) {
    return render_entry(s, d, nullptr, film_dev, nullptr, stats);
}
int rspt_render_samples(rspt_scene_t s, const rspt_render_desc* d, float* li_rgb, rspt_stats* stats) {
    return render_entry(s, d, nullptr, nullptr, li_rgb, stats);
}

int rspt_material_lobes(const rspt_scene_desc* desc, uint32_t material, uint32_t allow_multiple_lobes, rspt_material* out_material, rspt_bxdf out_bxdfs[8]) {
    if (!desc || !out_material || !out_bxdfs) return fail(RSPT_E_INVALID, "null argument");
    rspt_mat::Assembler as(desc, allow_multiple_lobes != 0);
    rspt_mat::Lobes lb;
    if (rspt_mat::Error e = as.assemble(material, &lb)) return fail(e.code, "%s", e.text.c_str());
    *out_material = lb.mat;
    if (lb.dynamic) return RSPT_MATERIAL_DYNAMIC;
    for (size_t i = 0; i < lb.lobes.size(); i++) out_bxdfs[i] = lb.lobes[i];
    return (int)lb.lobes.size();
}

int rspt_camera_decompose(const float start_m[16], float start_time, const float end_m[16], float end_time, int32_t* animated_out, float trs_out[46]) {
    if (!start_m || !end_m || !animated_out || !trs_out) return fail(RSPT_E_INVALID, "null argument");
    CamAnim ca;
    *animated_out = camanim::camera_keys(start_m, start_time, end_m, end_time, &ca) ? 1 : 0;
    if (*animated_out) {
        memcpy(trs_out, ca.t, sizeof ca.t);
        memcpy(trs_out + 6, ca.r, sizeof ca.r);
        memcpy(trs_out + 14, ca.s, sizeof ca.s);
    }
    return RSPT_OK;
}
int rspt_motion_bounds(const float start_m[16], float start_time, const float end_m[16], float end_time, const float box_min[3], const float box_max[3],
                       float out_min[3], float out_max[3], int32_t* flags_out) {
    if (!start_m || !end_m || !box_min || !box_max || !out_min || !out_max) return fail(RSPT_E_INVALID, "null argument");
    for (int i = 0; i < 16; i++) if (!std::isfinite(start_m[i]) || !std::isfinite(end_m[i])) return fail(RSPT_E_INVALID, "non-finite key matrix");
    for (int i = 0; i < 3; i++) if (!std::isfinite(box_min[i]) || !std::isfinite(box_max[i])) return fail(RSPT_E_INVALID, "non-finite box");
    if (!std::isfinite(start_time) || !std::isfinite(end_time)) return fail(RSPT_E_INVALID, "non-finite key time");
    motion::Keys keys;
    motion::make_keys(start_m, start_time, end_m, end_time, &keys);
    motion::Box b;
    for (int i = 0; i < 3; i++) { b.lo[i] = box_min[i]; b.hi[i] = box_max[i]; }
    bool overflow = false;
    const motion::Box r = motion::motion_bounds(keys, b, &overflow);
    if (overflow) return fail(RSPT_E_UNSUPPORTED, "more than eight motion-derivative zeros for one point: the reference's interval_find_zeros indexes past its array here (transform.rs:2346)");
    for (int i = 0; i < 3; i++) { out_min[i] = r.lo[i]; out_max[i] = r.hi[i]; }
    if (flags_out) *flags_out = (keys.animated ? 1 : 0) | (keys.has_rotation ? 2 : 0);
    return RSPT_OK;
}
}  // extern "C" (reopened below)
namespace {
// stage hook rspt_libm, codes 8 .. 12: the device's traversal / shading geometry AS THE KERNELS CALL IT, one element = 16 floats in, 16 floats out (include/rspt.h) —
// what tests/golden/geom_functions.npz pins by the reference's own text
__global__ void k_leaf_geom(uint32_t fn, const float* __restrict__ x, uint64_t n, float* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a[16], r[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { a[k] = x[16 * i + k]; r[k] = 0.0f; }
    if (fn == RSPT_LIBM_TRIANGLE) {              // Triangle::intersect's watertight test: ray_shear once per ray + tri_test per triangle (dev_scene.h), as traverse<> / k_trace_w4 run it
        const f3 d{a[12], a[13], a[14]};
        const RayShear rs = ray_shear(d);
        float t = 0.0f, b0 = 0.0f, b1 = 0.0f, b2 = 0.0f;
        const bool hit = tri_test(f3{a[0], a[1], a[2]}, f3{a[3], a[4], a[5]}, f3{a[6], a[7], a[8]}, f3{a[9], a[10], a[11]}, rs, a[15], &t, &b0, &b1, &b2);
        r[0] = hit ? 1.0f : 0.0f; r[1] = hit ? t : 0.0f; r[2] = hit ? b0 : 0.0f; r[3] = hit ? b1 : 0.0f; r[4] = hit ? b2 : 0.0f;
    } else if (fn == RSPT_LIBM_BOX) {            // Bounds3f::intersect_p: k_trace's box_hit (kernels.h), and k_trace_w4's choice (trace_w4.h): the pair form for finite reciprocals, else the literal chain
        const f3 o{a[6], a[7], a[8]}, inv{a[9], a[10], a[11]};
        const bool n0 = a[12] != 0.0f, n1 = a[13] != 0.0f, n2 = a[14] != 0.0f;
        r[0] = box_hit(float4{a[0], a[1], a[2], a[3]}, float4{a[4], a[5], 0.0f, 0.0f}, o, inv, n0, n1, n2, a[15]) ? 1.0f : 0.0f;
        if (fabsf(inv.x) < RSPT_INF && fabsf(inv.y) < RSPT_INF && fabsf(inv.z) < RSPT_INF) {
            bool h0, h1; float m0, m1;
            box_pair_hit_m(float4{a[0], a[0], a[3], a[3]}, float4{a[1], a[1], a[4], a[4]}, float4{a[2], a[2], a[5], a[5]}, o.x, o.y, o.z, inv.x, inv.y, inv.z, a[15], &h0, &h1, &m0, &m1);
            r[1] = h0 ? 1.0f : 0.0f; r[2] = h1 ? 1.0f : 0.0f; r[3] = 1.0f;
        } else {
            float m0;
            r[1] = r[2] = box_hit6_m(a[0], a[1], a[2], a[3], a[4], a[5], o, inv, n0, n1, n2, a[15], &m0) ? 1.0f : 0.0f; r[3] = 0.0f;
        }
    } else if (fn == RSPT_LIBM_OFFSET_RAY_ORIGIN) {   // pnt3_offset_ray_origin (dev_math.h): every spawned ray
        const f3 po = offset_ray_origin(f3{a[0], a[1], a[2]}, f3{a[3], a[4], a[5]}, f3{a[6], a[7], a[8]}, f3{a[9], a[10], a[11]});
        r[0] = po.x; r[1] = po.y; r[2] = po.z;
    } else if (fn == RSPT_LIBM_MICROFACET) {     // TrowbridgeReitzDistribution::d / lambda / g1 / g / pdf (dev_bsdf.h)
        const f3 wo{a[0], a[1], a[2]}, wh{a[3], a[4], a[5]};
        r[0] = tr_d(a[6], a[7], wh); r[1] = tr_lambda(a[6], a[7], wo); r[2] = tr_g1(a[6], a[7], wo); r[3] = tr_g(a[6], a[7], wo, wh); r[4] = tr_pdf(a[6], a[7], wo, wh);
    } else if (fn == RSPT_LIBM_AREA_LIGHT) {     // DiffuseAreaLight::sample_li on one emitting triangle without vertex normals (dev_scene.h light_sample_li / tri_sample_ref), radiance (1, 1, 1)
        SceneDev sc{};
        TriRec t; t.p0 = f3{a[0], a[1], a[2]}; t.p1 = f3{a[3], a[4], a[5]}; t.p2 = f3{a[6], a[7], a[8]}; t.material = 0; t.area_light = 0;
        const uint32_t fl = (uint32_t)a[14];
        t.flags = (fl & 2u) ? MF_FLIP : 0u;
        rspt_light lt{}; lt.kind = RSPT_LIGHT_DIFFUSE_AREA; lt.prim = 0; lt.L[0] = lt.L[1] = lt.L[2] = 1.0f; lt.two_sided = (fl & 4u) ? 1u : 0u;
        f3 wi{0.0f, 0.0f, 0.0f}; float pdf = 0.0f; LightSample ls; ls.p = ls.p_err = ls.n = f3{0.0f, 0.0f, 0.0f};
        const rgb li = light_sample_li(sc, lt, f3{a[9], a[10], a[11]}, f2{a[12], a[13]}, &wi, &pdf, &ls, &t);
        r[0] = pdf; r[1] = pdf == 0.0f ? 0.0f : wi.x; r[2] = pdf == 0.0f ? 0.0f : wi.y; r[3] = pdf == 0.0f ? 0.0f : wi.z; r[4] = pdf == 0.0f ? 0.0f : li.r;
        r[5] = ls.p.x; r[6] = ls.p.y; r[7] = ls.p.z; r[8] = ls.n.x; r[9] = ls.n.y; r[10] = ls.n.z; r[11] = ls.p_err.x; r[12] = ls.p_err.y; r[13] = ls.p_err.z;
    } else {                                     // RSPT_LIBM_VECTORS: vec3_cross_vec3, vec3_coordinate_system, refract, cosine_sample_hemisphere (dev_math.h, dev_bsdf.h)
        const f3 u{a[0], a[1], a[2]}, v{a[3], a[4], a[5]};
        const f3 c = cross(u, v);
        f3 v2{0.0f, 0.0f, 0.0f}, v3{0.0f, 0.0f, 0.0f}, wt{0.0f, 0.0f, 0.0f};
        coordinate_system(u, &v2, &v3);
        const bool ok = refract(u, v, a[6], &wt);
        const f3 h = cosine_hemisphere(f2{a[7], a[8]});
        r[0] = c.x; r[1] = c.y; r[2] = c.z; r[3] = v2.x; r[4] = v2.y; r[5] = v2.z; r[6] = v3.x; r[7] = v3.y; r[8] = v3.z;
        r[9] = ok ? wt.x : 0.0f; r[10] = ok ? wt.y : 0.0f; r[11] = ok ? wt.z : 0.0f; r[12] = ok ? 1.0f : 0.0f; r[13] = h.x; r[14] = h.y; r[15] = h.z;
    }
#pragma unroll
    for (int k = 0; k < 16; k++) out[16 * i + k] = r[k];
}
// rspt_libm code 14: one lobe record (the 116 bytes of an rspt_bxdf, as 29 floats) + wo, wi, u -> lobe_f, lobe_pdf, lobe_sample_f (dev_bsdf.h) as the shade kernels call them;
// 48 floats in, 48 floats out per element
__global__ void k_leaf_lobe(const float* __restrict__ x, uint64_t n, float* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    rspt_bxdf b;
    static_assert(sizeof(rspt_bxdf) == 116, "rspt_bxdf is 29 words");
    uint32_t* bw = reinterpret_cast<uint32_t*>(&b);
    for (int k = 0; k < 29; k++) bw[k] = __float_as_uint(x[48 * i + k]);
    const float* q = x + 48 * i + 29;
    const f3 wo{q[0], q[1], q[2]}, wi{q[3], q[4], q[5]};
    const LobeTex lt{nullptr, 0};
    float* o = out + 48 * i;
    for (int k = 0; k < 48; k++) o[k] = 0.0f;
    const rgb f = lobe_f<SF_ALL>(b, lt, wo, wi);
    o[0] = f.r; o[1] = f.g; o[2] = f.b; o[3] = lobe_pdf<SF_ALL>(b, lt, wo, wi);
    f3 w{0.0f, 0.0f, 0.0f}; float pdf = 0.0f; uint32_t st = 255u;
    const rgb sf = lobe_sample_f<SF_ALL>(b, lt, wo, &w, f2{q[6], q[7]}, &pdf, &st, true);
    o[4] = sf.r; o[5] = sf.g; o[6] = sf.b; o[7] = w.x; o[8] = w.y; o[9] = w.z; o[10] = pdf; o[11] = (float)st; o[12] = (float)lobe_type(b.type);
}
}  // namespace
extern "C" {
int rspt_libm(uint32_t fn, const float* x, const float* y, uint64_t n, float* out) {
    if (!g.inited) return fail(RSPT_E_NODEVICE, "rspt_init has not been called");
    if (n == 0) return RSPT_OK;
    if (fn > RSPT_LIBM_LOBE || !x || !out || (fn == RSPT_LIBM_ATAN2 && !y) || n > (fn >= RSPT_LIBM_MAT4_INVERSE ? (1ull << 27) : (1ull << 31))) return fail(RSPT_E_INVALID, "bad function, null argument or more than 2^31 values (2^27 sixteen-float elements)");
    HIP_TRY(hipSetDevice(g.device));
    float *xd = nullptr, *yd = nullptr, *od = nullptr;
    struct Guard { float **a, **b, **c; ~Guard() { for (float** p : {a, b, c}) if (*p) (void)hipFree(*p); } } guard{&xd, &yd, &od};
    int rc;
    const uint64_t per = fn == RSPT_LIBM_LOBE ? 48u : (fn >= RSPT_LIBM_MAT4_INVERSE ? 16u : 1u);   // values per element
    if ((rc = dev_alloc(&xd, n * per)) || (rc = dev_alloc(&od, n * per)) || (y && (rc = dev_alloc(&yd, n)))) return rc;
    HIP_TRY(hipMemcpyAsync(xd, x, n * per * sizeof(float), hipMemcpyHostToDevice, g.stream));
    if (y) HIP_TRY(hipMemcpyAsync(yd, y, n * sizeof(float), hipMemcpyHostToDevice, g.stream));
    if (fn == RSPT_LIBM_LOBE) hipLaunchKernelGGL(k_leaf_lobe, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, g.stream, xd, n, od);
    else if (fn > RSPT_LIBM_MAT4_INVERSE) hipLaunchKernelGGL(k_leaf_geom, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, g.stream, fn, xd, n, od);
    else hipLaunchKernelGGL(k_libm, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, g.stream, fn, xd, yd, n, od);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, od, n * per * sizeof(float), hipMemcpyDeviceToHost, g.stream));
    HIP_TRY(hipStreamSynchronize(g.stream));
    return RSPT_OK;
}

int rspt_light_distribution(rspt_scene_t s, uint32_t strategy, const float p[3], float* func_out, float* cdf_out, int32_t nvox_out[3], int32_t voxel_out[3]) {
    if (!g.inited) return fail(RSPT_E_NODEVICE, "rspt_init has not been called");
    if (!s || !p || !func_out || !cdf_out) return fail(RSPT_E_INVALID, "null argument");
    const uint32_t nl = s->dev.n_lights;
    if (nl == 0) return fail(RSPT_E_INVALID, "the scene has no lights");
    HIP_TRY(hipSetDevice(g.device));
    LightDistDev ld;
    const LightDist* lazy = nullptr;
    int rc = get_light_dist(s, strategy, &ld, &lazy);
    if (rc) return rc;
    // SpatialLightDistribution::lookup's voxel addressing (lightdistrib.rs:276-295), as dev_scene.h light_voxel does it
    int32_t pi[3] = {0, 0, 0};
    uint64_t vox = 0;
    if (ld.spatial) {
        for (int i = 0; i < 3; i++) {
            float o = p[i] - s->dev.wb_min[i];
            if (s->dev.wb_max[i] > s->dev.wb_min[i]) o /= s->dev.wb_max[i] - s->dev.wb_min[i];
            const float f = o * (float)ld.nvox[i];
            const int32_t v = (f != f) ? 0 : (f >= 2147483648.0f ? 2147483647 : (f <= -2147483648.0f ? (-2147483647 - 1) : (int32_t)f));
            pi[i] = v < 0 ? 0 : (v > ld.nvox[i] - 1 ? ld.nvox[i] - 1 : v);
        }
        vox = ((uint64_t)pi[2] * ld.nvox[1] + pi[1]) * ld.nvox[0] + pi[0];
    }
    uint64_t row = vox;
    if (lazy) {  // build the voxel if no path has asked for it yet: one round of the on-demand kernels with a one-entry list
        int32_t r = -1;
        HIP_TRY(hipMemcpy(&r, lazy->table + vox, sizeof r, hipMemcpyDeviceToHost));
        if (r < 0) {
            LightLazy lz;
            HIP_TRY(hipMemcpy(&lz, lazy->lazy, sizeof lz, hipMemcpyDeviceToHost));
            if (lz.n_rows >= lz.max_rows) return fail(RSPT_E_NOMEM, "spatial light distribution: row pool exhausted; raise RSPT_LIGHT_TABLE_POOL_BYTES");
            lz.n_new = 1;
            const uint32_t v32 = (uint32_t)vox;
            HIP_TRY(hipMemcpy(lazy->lazy, &lz, sizeof lz, hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(lazy->new_list, &v32, sizeof v32, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_ld_contrib_list, dim3(4), dim3(256), 0, g.stream, s->dev, ld.nvox[0], ld.nvox[1], ld.nvox[2], lazy->lazy, lazy->new_list, lazy->func);
            hipLaunchKernelGGL(k_ld_build_list, dim3(1), dim3(64), 0, g.stream, nl, lazy->lazy, lazy->new_list, lazy->func, lazy->cdf, lazy->func_int, lazy->table);
            hipLaunchKernelGGL(k_ld_commit, dim3(1), dim3(1), 0, g.stream, lazy->lazy);
            HIP_TRY(hipStreamSynchronize(g.stream));
            HIP_TRY(hipMemcpy(&r, lazy->table + vox, sizeof r, hipMemcpyDeviceToHost));
        }
        row = (uint64_t)r;
    }
    HIP_TRY(hipMemcpy(func_out, ld.func + row * nl, nl * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(cdf_out, ld.cdf + row * (nl + 1), (nl + 1) * sizeof(float), hipMemcpyDeviceToHost));
    for (int i = 0; i < 3; i++) { if (nvox_out) nvox_out[i] = ld.nvox[i]; if (voxel_out) voxel_out[i] = pi[i]; }
    return RSPT_OK;
}

int rspt_trace_device(rspt_scene_t s, const void* rays_dev, uint64_t n, void* out_dev, int any_hit, int repeat, double* ms_per_launch) {
    if (!g.inited) return fail(RSPT_E_NODEVICE, "rspt_init has not been called");
    if (!s || (n && (!rays_dev || !out_dev))) return fail(RSPT_E_INVALID, "null argument");
    if (n > 0xfffffff0ull) return fail(RSPT_E_UNSUPPORTED, "more than 2^32 rays per call");
    if (repeat < 1) repeat = 1;
    HIP_TRY(hipSetDevice(g.device));
    if (!g.totals) { int rc = dev_alloc(&g.totals, 8); if (rc) return rc; }
    const bool counters = env_size("RSPT_COUNTERS", 0) != 0;
    if (counters) HIP_TRY(hipMemsetAsync(g.totals, 0, 8 * sizeof(unsigned long long), g.stream));
    hipEvent_t e0 = get_event(0), e1 = get_event(1);
    HIP_TRY(hipEventRecord(e0, g.stream));
    int rc0 = ensure_counts(4);
    if (!rc0) rc0 = ensure_overflow_list(std::max<size_t>((size_t)n, 1));
    if (!rc0) rc0 = ensure_spill(pw_spill_threads());
    if (rc0) return rc0;
    uint32_t* cursor = &g.cnt[0].cursor_closest;
    for (int r = 0; r < repeat && n; r++) {
        HIP_TRY(hipMemsetAsync(&g.cnt[0], 0, sizeof(QueueCounts), g.stream));
        if (any_hit) launch_trace<true, 1>(0, counters, trace_grid(), s, nullptr, nullptr, (uint32_t)n, cursor, (const rspt_ray*)rays_dev, (const rspt_ray*)rays_dev, nullptr, nullptr, nullptr, (rspt_hit*)out_dev, g.totals, g.cnt[0].xcd_any);
        else launch_trace<false, 1>(0, counters, trace_grid(), s, nullptr, nullptr, (uint32_t)n, cursor, (const rspt_ray*)rays_dev, (const rspt_ray*)rays_dev, nullptr, nullptr, nullptr, (rspt_hit*)out_dev, g.totals, g.cnt[0].xcd_closest);
    }
    HIP_TRY(hipEventRecord(e1, g.stream));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(g.stream));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (ms_per_launch) *ms_per_launch = (double)ms / repeat;
    return RSPT_OK;
}

int rspt_trace(rspt_scene_t s, const rspt_ray* rays, uint64_t n, rspt_hit* out, int any_hit) {
    if (!g.inited) return fail(RSPT_E_NODEVICE, "rspt_init has not been called");
    if (!s || (n && (!rays || !out))) return fail(RSPT_E_INVALID, "null argument");
    if (n == 0) return RSPT_OK;
    HIP_TRY(hipSetDevice(g.device));
    rspt_ray* rd = nullptr;
    rspt_hit* hd = nullptr;
    HIP_TRY(hipMalloc((void**)&rd, n * sizeof(rspt_ray)));
    hipError_t e = hipMalloc((void**)&hd, n * sizeof(rspt_hit));
    if (e != hipSuccess) { (void)hipFree(rd); return fail(RSPT_E_NOMEM, "%s", hipGetErrorString(e)); }
    int rc = RSPT_OK;
    if (hipMemcpy(rd, rays, n * sizeof(rspt_ray), hipMemcpyHostToDevice) != hipSuccess) rc = fail(RSPT_E_HIP, "upload of rays failed");
    if (!rc) rc = rspt_trace_device(s, rd, n, hd, any_hit, 1, nullptr);
    if (!rc && hipMemcpy(out, hd, n * sizeof(rspt_hit), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(RSPT_E_HIP, "download of hits failed");
    (void)hipFree(rd);
    (void)hipFree(hd);
    return rc;
}

// counters of the last rspt_trace_device / rspt_render when RSPT_COUNTERS=1: out[0] nodes visited, out[1] triangles tested
int rspt_last_counters(uint64_t out[3]) {
    if (!g.inited || !g.totals || !g.cnt) return fail(RSPT_E_INVALID, "no counters");
    unsigned long long t[2];
    HIP_TRY(hipMemcpy(t, g.totals, sizeof t, hipMemcpyDeviceToHost));
    QueueCounts c0;
    HIP_TRY(hipMemcpy(&c0, g.cnt, sizeof c0, hipMemcpyDeviceToHost));
    out[0] = t[0]; out[1] = t[1];
    out[2] = (uint64_t)c0.overflow_closest;  // rspt_trace_device keeps its counters in slot 0
    return RSPT_OK;
}

int rspt_dev_alloc(uint64_t bytes, void** out) {
    if (!g.inited) return fail(RSPT_E_NODEVICE, "rspt_init has not been called");
    if (!out) return fail(RSPT_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(g.device));
    HIP_TRY(hipMalloc(out, bytes ? bytes : 1));
    return RSPT_OK;
}
int rspt_dev_free(void* p) {
    if (p) HIP_TRY(hipFree(p));
    return RSPT_OK;
}
int rspt_dev_upload(void* dst, const void* src, uint64_t bytes) {
    if (bytes && (!dst || !src)) return fail(RSPT_E_INVALID, "null argument");
    if (bytes) HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return RSPT_OK;
}
int rspt_dev_download(void* dst, const void* src, uint64_t bytes) {
    if (bytes && (!dst || !src)) return fail(RSPT_E_INVALID, "null argument");
    if (bytes) HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return RSPT_OK;
}

}  // extern "C"
