// C-ABI host side of libcbo_hip.so (declared in include/cbo_hip.h): handle management, the jitchol
// retry ladder, candidate chunking, profiling events.  All arithmetic of the path runs in the HIP
// kernels of kernels_*.hip; there is no CPU fallback here.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <mutex>
#include <string>
#include <map>
#include <utility>
#include <vector>

#include "cbo_internal.h"
#include "schedule_tuner.h"

using namespace cbo;

static thread_local std::string g_err;
static std::atomic<uint64_t> g_fit_stamp{0};

static int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}
namespace cbo {
int set_error(int code, const std::string &msg) { return fail(code, msg); }
}

// Every context that cbo_init handed out and cbo_shutdown has not seen yet.  A context owns CU-masked streams
// (hipExtStreamCreateWithCUMask); when those are still alive while the HIP runtime runs its own exit handlers,
// tools that hook finalisation (rocprofv3) crash inside __cxa_finalize.  The first cbo_init therefore registers an
// atexit handler -- after the runtime's own, so it runs BEFORE them -- that shuts down whatever the caller left
// open: a C or ctypes consumer that exits without cbo_shutdown is safe too.  cbo_shutdown ignores handles that are
// not (or no longer) registered, which also makes a second call on the same handle harmless.
static std::mutex g_live_mutex;
static std::vector<cbo_ctx *> g_live;
static void shutdown_all_at_exit()
{
    for (;;) {
        cbo_ctx *c = nullptr;
        {
            std::lock_guard<std::mutex> lock(g_live_mutex);
            if (g_live.empty()) break;
            c = g_live.back();
        }
        cbo_shutdown(c);
    }
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(CBO_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                \
    } while (0)

constexpr size_t kStageBytes = 64 << 10;

enum Phase { PH_KXX = 0, PH_CHOL, PH_ALPHA, PH_KSTAR, PH_TRSM, PH_ACQ, PH_CONVERT, PH_COUNT };

struct EventPair {
    hipEvent_t a, b;
    int phase;
};

struct cbo_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t side_stream = nullptr;        // look-ahead stream of the Cholesky
    hipStream_t sweep_stream = nullptr;       // the sweep when it is pipelined with the factorisation (lower priority)
    hipStream_t bulk_stream = nullptr;        // its bulk updates (lowest priority)
    std::vector<hipEvent_t> chol_events;
    std::vector<hipEvent_t> pipe_events;      // factorisation -> sweep dependencies
    hipEvent_t ev_join = nullptr, ev_join2 = nullptr, ev_fork = nullptr;
    hipEvent_t region_a = nullptr, region_b = nullptr;
    int pipe_chunk_blocks = 1;        // row blocks per update workgroup (1 since the updates go in K = 512 groups: profiles/r03_schedule_crossover.txt)
    bool pipe_half_lds = true;
    double pipe_tail_frac = -1.0;    // CBO_HIP_PIPE_TAIL: rows (fraction) left to the closing left-looking launch; < 0 = automatic
    int n_cu = 256;
    int n_cu_pipe = 256;             // CUs the pipelined sweep's streams may use (the rest is kept for the factorisation)
    ScheduleTable schedule;          // (padded rows, padded candidates) -> measured schedule of cbo_gp_fit_sweep (schedule_tuner.h)
    int64_t fused_fallbacks = 0;     // factorisations repeated with separate launches after a fused launch gave up
    bool sweep_cache = true;         // CBO_HIP_SWEEP_CACHE=0: never reuse a candidate set's q, mu between sweeps
    bool small_sets = true;          // CBO_HIP_SMALL_SETS=0: cbo_acq_sweep_sets always takes the general path
    int sweep_mode = -1;             // CBO_HIP_SWEEP: 0 = always left-looking, 1 = always right-looking, else automatic
    int overlap_mode = -1;           // CBO_HIP_OVERLAP: 0 = cbo_gp_fit_sweep never overlaps, 1 = always, else automatic
    bool profiling = false;
    std::vector<EventPair> pending;
    EventPair pipe_cur{};                     // the launch pipe_mark is currently bracketing
    std::vector<hipEvent_t> pool;
    cbo_timers timers{};
    // sweep workspaces (grown on demand)
    double *V = nullptr; size_t V_bytes = 0;
    double *W = nullptr; size_t W_bytes = 0;          // -Ky^-1 for the likelihood gradients
    double *gpart = nullptr; size_t gpart_elems = 0;
    double *mupart = nullptr; size_t mupart_elems = 0;   // fp32 sweep: per-row-tile partial sums of K*^T alpha
    // multi-set sweep of small models (cbo_acq_sweep_sets): descriptors, per-workgroup scratch, partial and final winners
    cbo_small_set *sets_host = nullptr; int sets_cap = 0;       // pinned, read by the kernel directly
    cbo_small_result *small_out = nullptr;                      // pinned, written by the kernel directly
    double *small_scratch = nullptr; size_t small_scratch_elems = 0;
    double *small_part_val = nullptr; int64_t *small_part_idx = nullptr; size_t small_part_elems = 0;
    int *small_info = nullptr;                                  // device, sets_cap status words + sets_cap tickets (zero between calls)
    int small_seq = 0;                                          // sequence number of the last multi-set call
    int polled_launches = 0;                                    // launches completed by polling since the last stream sync
    cbo_small_lml_result *lml_out = nullptr;                    // pinned, written by small_lml_kernel
    double *q = nullptr, *mu = nullptr, *mean = nullptr, *var = nullptr, *acq = nullptr; size_t vec_elems = 0;
    double *part_val = nullptr; int64_t *part_idx = nullptr;
    double *best_val = nullptr; int64_t *best_idx = nullptr;   // device
    double *h_best_val = nullptr; int64_t *h_best_idx = nullptr; // pinned host
    int *h_info = nullptr;
    // host-buffer entry points (cbo_gp_predict, _grouped, _gradients, cbo_acq_sweep_host) reuse ONE grow-only candidate
    // set instead of creating and destroying one per call; gradient / export scratch likewise
    cbo_cands *scratch_k = nullptr;
    double *grads = nullptr; size_t grads_elems = 0;
    double *export_buf = nullptr; size_t export_elems = 0;
    // small uploads (cbo_gp_upload_data / cbo_gp_set_data of a few KB, every trial of the reference's loop): one
    // pinned staging buffer the preparation kernel reads directly; `stage_done` guards its reuse
    double *stage = nullptr; hipEvent_t stage_done = nullptr; bool stage_pending = false;
    size_t max_ws_bytes = (size_t)32 << 30;   // V workspace cap: 288 GB of HBM per GPU, one chunk whenever possible
    char name[128] = {0};
};

struct cbo_gp {
    cbo_ctx *ctx = nullptr;
    int64_t n = 0, n_pad = 0, lda = 0;
    int d = 0;
    PointSet X;                      // scaled SoA coordinates
    double *raw = nullptr;           // staging for AoS upload (n*d)
    double *y = nullptr, *ls_dev = nullptr;
    std::vector<double> ls;          // per-dim lengthscales (ard) or single value
    std::vector<double> h_pv;        // host copy of prior variance (diag check of jitchol)
    KernelHyper h{};
    double noise_var = 0.0;
    double *A = nullptr;             // [n_pad][lda] Ky -> U, rhs strip at column n_pad
    double *invDt = nullptr;         // [n_pad/16][16][16]
    double *alpha = nullptr;         // [2*n_pad]
    double *z = nullptr;             // [n_pad] contiguous copy of L^-1 r
    int *info = nullptr;
    bool fitted = false;
    uint64_t fit_stamp = 0;          // unique per successful fit (0 = not fitted); candidates key their cache on it
    bool alpha_ready = false;
    int tries = 0;
    double jitter = 0.0;
    // append-only trial step: the fit this one extends by one observation, and what extending V needs
    uint64_t parent_stamp = 0;
    double append_d = 0.0, append_zn = 0.0;
    double *lvec = nullptr;          // [n_pad] the new column of U, contiguous
    cbo_cands *probe = nullptr;      // the appended point as a one-candidate set (scaled coordinates, prior)
    // backward substitution through the forward kernel (prediction gradients of whole grids): the reversed factor and
    // its diagonal-tile inverses, built on first use after a fit; 1 / lengthscale per dimension (ARD)
    double *T = nullptr, *invT = nullptr, *inv_ls_dev = nullptr;
    uint64_t t_stamp = 0;
    // CBO_DTYPE_F32: fp32 copies of the factor for the sweep (kernels_f32.hip), refreshed by every successful fit
    int dtype = CBO_DTYPE_F64;
    int64_t n32 = 0, ldu32 = 0;
    float *Uf = nullptr, *invF = nullptr;
    uint64_t f32_stamp = 0;          // fit stamp the copies belong to
};

struct cbo_cands {
    cbo_ctx *ctx = nullptr;
    int64_t m = 0, m_pad = 0;
    int d = 0;
    int64_t cap_m_pad = 0; int cap_d = 0; bool cap_prior = false;      // what the buffers below can hold (grow-only)
    double *raw = nullptr;           // AoS (m,d) as uploaded
    double *pm = nullptr, *pv = nullptr;
    bool has_prior = false;          // pm / pv carry this set's prior closures (the buffers may outlive that)
    int64_t index_offset = 0;
    // scaled view for the GP it was last prepared for
    PointSet P;
    const cbo_gp *prepared_for = nullptr;
    std::vector<double> prepared_ls;
    // q = sum V^2 and mu = V^T z of the last sweep, valid while the model's fit stamp is the one recorded here:
    // between refits only the incumbent changes, and EI / cost / arg-max are recomputed from these two vectors
    double *q = nullptr, *mu = nullptr;
    uint64_t fit_stamp = 0;
    // cbo_cands_keep_solution: V = L^-1 K* stays resident so that an appended observation extends it by one row
    bool keep_v = false;
    double *V = nullptr;
    int64_t v_ld = 0, v_rows_cap = 0, v_rows = 0;
    uint64_t v_stamp = 0;
    double *partial = nullptr;       // [64][m_pad] slice sums of the row update
};

// ---- profiling helpers ---------------------------------------------------------------------------
static hipEvent_t get_event(cbo_ctx *c)
{
    if (!c->pool.empty()) {
        hipEvent_t e = c->pool.back();
        c->pool.pop_back();
        return e;
    }
    hipEvent_t e;
    hipEventCreateWithFlags(&e, hipEventDisableSystemFence);   // device-scope ordering is all the stream needs
    return e;
}

struct PhaseScope {
    cbo_ctx *c;
    EventPair p{};
    bool on;
    hipStream_t st;
    PhaseScope(cbo_ctx *ctx, int phase, hipStream_t stream = nullptr)
        : c(ctx), on(ctx->profiling), st(stream ? stream : ctx->stream)
    {
        if (on) {
            p.a = get_event(c);
            p.b = get_event(c);
            p.phase = phase;
            hipEventRecord(p.a, st);
        }
    }
    ~PhaseScope()
    {
        if (on) {
            hipEventRecord(p.b, st);
            c->pending.push_back(p);
        }
    }
};

static void resolve_events(cbo_ctx *c)
{
    if (c->pending.empty()) return;
    hipStreamSynchronize(c->stream);
    hipStreamSynchronize(c->sweep_stream);
    hipStreamSynchronize(c->bulk_stream);
    for (auto &p : c->pending) {
        float ms = 0.f;
        hipEventElapsedTime(&ms, p.a, p.b);
        switch (p.phase) {
            case PH_KXX: c->timers.ms_kxx += ms; break;
            case PH_CHOL: c->timers.ms_chol += ms; break;
            case PH_ALPHA: c->timers.ms_alpha += ms; break;
            case PH_KSTAR: c->timers.ms_kstar += ms; break;
            case PH_TRSM: c->timers.ms_trsm += ms; break;
            case PH_ACQ: c->timers.ms_acq += ms; break;
            case PH_CONVERT: c->timers.ms_f32_convert += ms; break;
        }
        c->pool.push_back(p.a);
        c->pool.push_back(p.b);
    }
    c->pending.clear();
}

// ---- context -------------------------------------------------------------------------------------
static void destroy_ctx(cbo_ctx *c);

extern "C" int cbo_abi_version(void)
{
#ifdef CBO_DIAG_KNOBS
    return CBO_HIP_ABI_DIAG_BASE + CBO_HIP_ABI_VERSION;
#else
    return (f32_debug_mask() != 0 ? CBO_HIP_ABI_DIAG_BASE : 0) + CBO_HIP_ABI_VERSION;
#endif
}
extern "C" const char *cbo_last_error(void) { return g_err.c_str(); }

extern "C" int cbo_device_count(int *count_out)
{
    if (!count_out) return fail(CBO_ERR_INVALID, "count_out is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); n = 0; }
    *count_out = n;
    return CBO_OK;
}

extern "C" int cbo_init(int device_id, cbo_ctx **out)
{
    if (!out) return fail(CBO_ERR_INVALID, "out is NULL");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(CBO_ERR_NO_DEVICE, "no HIP device visible: libcbo_hip has no CPU fallback");
    }
    if (device_id < 0 || device_id >= n) return fail(CBO_ERR_INVALID, "device_id out of range");
    HIP_TRY(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(CBO_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName +
                                           ", this library carries gfx950 code objects only");
    cbo_ctx *c = new cbo_ctx();
    c->device = device_id;
    std::snprintf(c->name, sizeof(c->name), "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    // the factorisation's streams outrank the sweep stream: its kernels are short, few and on the critical path
    int prio_low = 0, prio_high = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_high);
    // The look-ahead stream of the factorisation carries the bulk trailing updates: an ordinary stream one priority below
    // the chain's (keeping CUs away from it, or raising it to the chain's priority, measured neutral to -4 %: NOTES.md).
    if (e == hipSuccess)
        e = hipStreamCreateWithPriority(&c->side_stream, hipStreamNonBlocking, (prio_low + prio_high) / 2);
    // The sweep streams leave a few CUs per XCD to the factorisation: its diagonal-block kernel needs a whole
    // CU's LDS and would otherwise wait behind a queue of half-LDS sweep workgroups that keep every CU partly
    // occupied.  CU-mask bit b is CU b/8 of XCD b%8 on this device (scripts/probes/cumask_probe.hip).
    if (e == hipSuccess) {
        int reserve = 4;
        const char *rv = std::getenv("CBO_HIP_PIPE_RESERVE");
        if (rv) reserve = std::atoi(rv);
        const int n_cu = prop.multiProcessorCount;
        std::vector<uint32_t> mask((size_t)(n_cu + 31) / 32, 0u);
        for (int b = 0; b < n_cu; ++b)
            if (b / 8 >= reserve) mask[(size_t)b / 32] |= 1u << (b % 32);
        bool masked = false;
        c->n_cu_pipe = n_cu;
        if (reserve > 0 && reserve * 8 < n_cu) {
            masked = hipExtStreamCreateWithCUMask(&c->sweep_stream, (uint32_t)mask.size(), mask.data()) == hipSuccess &&
                     hipExtStreamCreateWithCUMask(&c->bulk_stream, (uint32_t)mask.size(), mask.data()) == hipSuccess;
            if (masked) c->n_cu_pipe = n_cu - reserve * 8;
            if (!masked) {                       // no CU masking on this stack: plain lower-priority streams instead
                (void)hipGetLastError();
                if (c->sweep_stream) { hipStreamDestroy(c->sweep_stream); c->sweep_stream = nullptr; }
                if (c->bulk_stream) { hipStreamDestroy(c->bulk_stream); c->bulk_stream = nullptr; }
            }
        }
        if (!masked) {
            e = hipStreamCreateWithPriority(&c->sweep_stream, hipStreamNonBlocking, (prio_low + prio_high) / 2);
            if (e == hipSuccess) e = hipStreamCreateWithPriority(&c->bulk_stream, hipStreamNonBlocking, prio_low);
        }
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming | hipEventDisableSystemFence);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join2, hipEventDisableTiming | hipEventDisableSystemFence);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming | hipEventDisableSystemFence);
    if (e == hipSuccess) e = hipMalloc(&c->part_val, 2048 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&c->part_idx, 2048 * sizeof(int64_t));
    if (e == hipSuccess) e = hipMalloc(&c->best_val, sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&c->best_idx, sizeof(int64_t));
    if (e == hipSuccess) e = hipHostMalloc(&c->h_best_val, sizeof(double));
    if (e == hipSuccess) e = hipHostMalloc(&c->h_best_idx, sizeof(int64_t));
    if (e == hipSuccess) e = hipHostMalloc(&c->h_info, sizeof(int));
    if (e == hipSuccess) e = hipHostMalloc(&c->stage, kStageBytes);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->stage_done, hipEventDisableTiming);
    if (e != hipSuccess) {
        destroy_ctx(c);
        return fail(CBO_ERR_HIP, std::string("cbo_init: ") + hipGetErrorString(e));
    }
    const char *ws = std::getenv("CBO_HIP_WORKSPACE_MB");
    if (ws) c->max_ws_bytes = (size_t)std::atoll(ws) << 20;
    c->n_cu = prop.multiProcessorCount;
    const char *sm = std::getenv("CBO_HIP_SWEEP");
    if (sm) c->sweep_mode = std::atoi(sm);
    const char *sc = std::getenv("CBO_HIP_SWEEP_CACHE");
    if (sc && std::atoi(sc) == 0) c->sweep_cache = false;
    const char *ss = std::getenv("CBO_HIP_SMALL_SETS");
    if (ss && std::atoi(ss) == 0) c->small_sets = false;
    const char *om = std::getenv("CBO_HIP_OVERLAP");
    if (om) c->overlap_mode = std::atoi(om);
    const char *tf = std::getenv("CBO_HIP_PIPE_TAIL");
    if (tf) c->pipe_tail_frac = std::atof(tf);
    {
        static std::once_flag once;
        std::call_once(once, [] { std::atexit(shutdown_all_at_exit); });
        std::lock_guard<std::mutex> lock(g_live_mutex);
        g_live.push_back(c);
    }
    *out = c;
    return CBO_OK;
}

namespace cbo {
hipStream_t ctx_stream(cbo_ctx *c) { return c->stream; }
int ctx_device(cbo_ctx *c) { return c->device; }
}

extern "C" void cbo_shutdown(cbo_ctx *c)
{
    if (!c) return;
    {
        std::lock_guard<std::mutex> lock(g_live_mutex);
        auto it = std::find(g_live.begin(), g_live.end(), c);
        if (it == g_live.end()) return;              // already shut down (or never ours): the pointer is not touched
        g_live.erase(it);
    }
    destroy_ctx(c);
}

static void destroy_ctx(cbo_ctx *c)
{
    hipSetDevice(c->device);
    if (c->scratch_k) { cbo_cands_destroy(c->scratch_k); c->scratch_k = nullptr; }
    hipFree(c->grads); hipFree(c->export_buf);
    if (c->stream) hipStreamSynchronize(c->stream);
    for (auto &p : c->pending) { hipEventDestroy(p.a); hipEventDestroy(p.b); }
    for (auto e : c->pool) hipEventDestroy(e);
    hipFree(c->W); hipFree(c->gpart); hipFree(c->mupart);
    hipHostFree(c->sets_host); hipHostFree(c->small_out); hipHostFree(c->lml_out); hipFree(c->small_scratch); hipFree(c->small_part_val);
    hipFree(c->small_part_idx); hipFree(c->small_info);
    hipFree(c->V); hipFree(c->q); hipFree(c->mu); hipFree(c->mean); hipFree(c->var); hipFree(c->acq);
    hipFree(c->part_val); hipFree(c->part_idx); hipFree(c->best_val); hipFree(c->best_idx);
    hipHostFree(c->h_best_val); hipHostFree(c->h_best_idx); hipHostFree(c->h_info); hipHostFree(c->stage);
    if (c->stage_done) hipEventDestroy(c->stage_done);
    for (auto e : c->chol_events) hipEventDestroy(e);
    for (auto e : c->pipe_events) hipEventDestroy(e);
    if (c->region_a) hipEventDestroy(c->region_a);
    if (c->region_b) hipEventDestroy(c->region_b);
    if (c->ev_join) hipEventDestroy(c->ev_join);
    if (c->ev_join2) hipEventDestroy(c->ev_join2);
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    if (c->sweep_stream) hipStreamDestroy(c->sweep_stream);
    if (c->bulk_stream) hipStreamDestroy(c->bulk_stream);
    if (c->side_stream) hipStreamDestroy(c->side_stream);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int cbo_synchronize(cbo_ctx *c)
{
    if (!c) return fail(CBO_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());            // every stream of the device, the communicator's included
    return CBO_OK;
}

#ifdef CBO_DIAG_KNOBS
// Timing-only (diagnostic build): the LDS-staged update kernel alone, C[klen:n, :] -= U[0:klen, klen:n]^T U[0:klen, :] on
// an n x n array of noise (upper part only when `upper`), `reps` launches between two events -- the bulk trailing
// update of a factorisation at a given K (scripts/update_kernel_timing.py).
__global__ void diag_fill_kernel(double *p, int64_t n)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        p[i] = 1e-3 * (double)((i * 2654435761u) & 1023u) - 0.5;
}
extern "C" int cbo_diag_update_kernel_time(cbo_ctx *c, int n, int klen, int chunk_blocks, int half_lds, int upper, int reps,
                                           double *ms_out)
{
    if (!c || !ms_out) return fail(CBO_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    const int64_t lda = (int64_t)n + 80;
    double *A = nullptr;
    HIP_TRY(hipMalloc(&A, sizeof(double) * (size_t)n * (size_t)lda));
    hipLaunchKernelGGL(diag_fill_kernel, dim3(4096), dim3(256), 0, c->stream, A, (int64_t)n * lda);
    hipEvent_t a, b;
    HIP_TRY(hipEventCreate(&a));
    HIP_TRY(hipEventCreate(&b));
    launch_gemm_update(c->stream, A, lda, A, lda, A, lda, 0, klen, klen, n, n, chunk_blocks, half_lds != 0, upper != 0, nullptr);
    HIP_TRY(hipEventRecord(a, c->stream));
    for (int r = 0; r < reps; ++r)
        launch_gemm_update(c->stream, A, lda, A, lda, A, lda, 0, klen, klen, n, n, chunk_blocks, half_lds != 0, upper != 0, nullptr);
    HIP_TRY(hipEventRecord(b, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, a, b));
    *ms_out = (double)ms / reps;
    hipEventDestroy(a);
    hipEventDestroy(b);
    hipFree(A);
    return CBO_OK;
}
#endif

extern "C" int cbo_set_profiling(cbo_ctx *c, int enabled)
{
    if (!c) return fail(CBO_ERR_INVALID, "ctx is NULL");
    c->profiling = enabled != 0;
    return CBO_OK;
}

extern "C" int cbo_reset_timers(cbo_ctx *c)
{
    if (!c) return fail(CBO_ERR_INVALID, "ctx is NULL");
    resolve_events(c);
    c->timers = cbo_timers{};
    return CBO_OK;
}

extern "C" int cbo_get_timers(cbo_ctx *c, cbo_timers *out)
{
    if (!c || !out) return fail(CBO_ERR_INVALID, "NULL argument");
    resolve_events(c);
    *out = c->timers;
    return CBO_OK;
}

// One event pair on the main stream around a region of calls: device time of the region whatever runs inside
// (several streams join the main stream before every call returns).
extern "C" int cbo_region_begin(cbo_ctx *c)
{
    if (!c) return fail(CBO_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    if (!c->region_a) {
        HIP_TRY(hipEventCreate(&c->region_a));
        HIP_TRY(hipEventCreate(&c->region_b));
    }
    HIP_TRY(hipEventRecord(c->region_a, c->stream));
    return CBO_OK;
}

extern "C" int cbo_region_end(cbo_ctx *c, double *ms_out)
{
    if (!c || !ms_out) return fail(CBO_ERR_INVALID, "NULL argument");
    if (!c->region_a) return fail(CBO_ERR_INVALID, "cbo_region_end without cbo_region_begin");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventRecord(c->region_b, c->stream));
    HIP_TRY(hipEventSynchronize(c->region_b));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, c->region_a, c->region_b));
    *ms_out = ms;
    return CBO_OK;
}

extern "C" int cbo_device_name(cbo_ctx *c, char *buf, int buflen)
{
    if (!c || !buf || buflen <= 0) return fail(CBO_ERR_INVALID, "NULL argument");
    std::snprintf(buf, (size_t)buflen, "%s", c->name);
    return CBO_OK;
}

// ---- GP ------------------------------------------------------------------------------------------
static void free_gp_data(cbo_gp *g)
{
    hipFree(g->X.xs); hipFree(g->X.sq); hipFree(g->X.sv); hipFree(g->X.pm); hipFree(g->X.pv);
    hipFree(g->raw); hipFree(g->y); hipFree(g->A); hipFree(g->invDt); hipFree(g->alpha); hipFree(g->z); hipFree(g->lvec);
    hipFree(g->Uf); hipFree(g->invF); hipFree(g->T); hipFree(g->invT);
    g->Uf = g->invF = nullptr; g->f32_stamp = 0;
    g->T = g->invT = nullptr; g->t_stamp = 0;
    g->X = PointSet{};
    g->raw = g->y = g->A = g->invDt = g->alpha = g->z = g->lvec = nullptr;
    g->parent_stamp = 0;
    g->n = g->n_pad = 0;             // no storage: every entry point that needs data checks g->n
    g->fitted = false;
}

static int upload_gp_data(cbo_gp *g, int64_t n, const double *X, const double *y, const double *pm, const double *pv)
{
    cbo_ctx *c = g->ctx;
    if (n <= 0 || !X || !y) return fail(CBO_ERR_INVALID, "n must be positive and X, y non-NULL");
    if ((pm == nullptr) != (pv == nullptr))
        return fail(CBO_ERR_INVALID, "prior mean and prior variance must be given together");
    if (n > ((int64_t)1 << 30)) return fail(CBO_ERR_INVALID, "n too large");
    const int64_t n_pad = round_up(n, kPadN);
    if (n_pad != g->n_pad || (pv != nullptr) != (g->X.sv != nullptr)) {
        free_gp_data(g);
        g->lda = n_pad + kRhsCols + kLdExtra;
        HIP_TRY(hipMalloc(&g->X.xs, sizeof(double) * g->d * n_pad));
        HIP_TRY(hipMalloc(&g->X.sq, sizeof(double) * n_pad));
        if (pv) {
            HIP_TRY(hipMalloc(&g->X.sv, sizeof(double) * n_pad));
            HIP_TRY(hipMalloc(&g->X.pm, sizeof(double) * n_pad));
            HIP_TRY(hipMalloc(&g->X.pv, sizeof(double) * n_pad));
        }
        HIP_TRY(hipMalloc(&g->raw, sizeof(double) * n_pad * g->d));
        HIP_TRY(hipMalloc(&g->y, sizeof(double) * n_pad));
        HIP_TRY(hipMalloc(&g->A, sizeof(double) * n_pad * g->lda));
        HIP_TRY(hipMalloc(&g->invDt, sizeof(double) * (n_pad / 16) * 256));
        HIP_TRY(hipMalloc(&g->alpha, sizeof(double) * 2 * n_pad));
        HIP_TRY(hipMalloc(&g->z, sizeof(double) * n_pad));
        HIP_TRY(hipMalloc(&g->lvec, sizeof(double) * n_pad));
        if (g->dtype == CBO_DTYPE_F32) {
            g->n32 = round_up(n_pad, kPadN32);
            g->ldu32 = g->n32 + 32;
            HIP_TRY(hipMalloc(&g->Uf, sizeof(float) * (size_t)g->n32 * (size_t)g->ldu32));
            HIP_TRY(hipMalloc(&g->invF, sizeof(float) * (size_t)(g->n32 / 16) * 256));
        }
        g->n_pad = n_pad;            // only now: a failed allocation above leaves the handle empty (n_pad == 0)
    }
    g->n = n;
    g->X.n = n; g->X.ld = n_pad; g->X.d = g->d;
    g->fitted = false;
    const size_t stage_need = sizeof(double) * (size_t)(n * g->d + n + (pv ? 2 * n : 0));
    if (stage_need <= kStageBytes) {
        // small upload: host arrays -> pinned staging -> ONE kernel that reads the staging buffer itself.  The
        // caller's buffers are free as soon as they are copied here; nothing to wait for on the stream.
        if (c->stage_pending) { HIP_TRY(hipEventSynchronize(c->stage_done)); c->stage_pending = false; }
        double *st = c->stage;
        std::memcpy(st, X, sizeof(double) * n * g->d);
        std::memcpy(st + n * g->d, y, sizeof(double) * n);
        g->h_pv.clear();
        if (pv) {
            std::memcpy(st + n * g->d + n, pm, sizeof(double) * n);
            std::memcpy(st + n * g->d + 2 * n, pv, sizeof(double) * n);
            g->h_pv.assign(pv, pv + n);
        }
        launch_prep_points_staged(c->stream, st, n, g->d, g->h.ard ? g->ls_dev : nullptr, pv != nullptr, g->raw, g->y,
                                  g->X.pm, g->X.pv, g->X.xs, n_pad, g->X.sq, g->X.sv);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(c->stage_done, c->stream));
        c->stage_pending = true;
        return CBO_OK;
    }
    HIP_TRY(hipMemcpyAsync(g->raw, X, sizeof(double) * n * g->d, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(g->y, y, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    g->h_pv.clear();
    if (pv) {
        HIP_TRY(hipMemcpyAsync(g->X.pm, pm, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(g->X.pv, pv, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
        g->h_pv.assign(pv, pv + n);
    }
    launch_prep_points(c->stream, g->raw, n, g->d, g->h.ard ? g->ls_dev : nullptr, pv ? g->X.pv : nullptr, g->X.xs,
                       n_pad, g->X.sq, g->X.sv);
    HIP_TRY(hipGetLastError());
    // host buffers are the caller's: make sure the copies are done before returning
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CBO_OK;
}

extern "C" int cbo_gp_create(cbo_ctx *c, int dtype, int64_t n, int d, const double *X, const double *y,
                             const double *pm, const double *pv, double variance, const double *lengthscale,
                             int ard, double noise_var, int zero_diag, cbo_gp **out)
{
    if (!c || !out || !lengthscale) return fail(CBO_ERR_INVALID, "NULL argument");
    if (dtype != CBO_DTYPE_F64 && dtype != CBO_DTYPE_F32) return fail(CBO_ERR_INVALID, "dtype must be CBO_DTYPE_F64 or CBO_DTYPE_F32");
    if (d < 1 || d > CBO_MAX_DIM) return fail(CBO_ERR_INVALID, "d must be in [1, CBO_MAX_DIM]");
    HIP_TRY(hipSetDevice(c->device));
    cbo_gp *g = new cbo_gp();
    g->ctx = c;
    g->d = d;
    g->dtype = dtype;
    g->noise_var = noise_var;
    g->h.variance = variance;
    g->h.ard = ard ? 1 : 0;
    g->h.zero_diag = zero_diag ? 1 : 0;
    g->h.lengthscale = ard ? 1.0 : lengthscale[0];
    g->ls.assign(lengthscale, lengthscale + (ard ? d : 1));
    if (hipMalloc(&g->info, sizeof(int) * (1 + kCholFlagSlots)) != hipSuccess) { delete g; return fail(CBO_ERR_HIP, "hipMalloc info"); }
    if (ard) {
        if (hipMalloc(&g->ls_dev, sizeof(double) * d) != hipSuccess ||
            hipMemcpy(g->ls_dev, lengthscale, sizeof(double) * d, hipMemcpyHostToDevice) != hipSuccess) {
            cbo_gp_destroy(g);
            return fail(CBO_ERR_HIP, "hipMalloc/hipMemcpy lengthscales");
        }
    }
    const int rc = upload_gp_data(g, n, X, y, pm, pv);
    if (rc != CBO_OK) { cbo_gp_destroy(g); return rc; }
    *out = g;
    return CBO_OK;
}

extern "C" void cbo_gp_destroy(cbo_gp *g)
{
    if (g && g->probe) { cbo_cands_destroy(g->probe); g->probe = nullptr; }
    if (!g) return;
    hipSetDevice(g->ctx->device);
    hipStreamSynchronize(g->ctx->stream);
    free_gp_data(g);
    hipFree(g->info);
    hipFree(g->ls_dev); hipFree(g->inv_ls_dev);
    delete g;
}

extern "C" int64_t cbo_gp_n(const cbo_gp *g) { return g ? g->n : -1; }
extern "C" int cbo_gp_dtype(const cbo_gp *g) { return g ? g->dtype : -1; }

extern "C" int cbo_gp_jitter(const cbo_gp *g, int *tries_out, double *jitter_out)
{
    if (!g) return fail(CBO_ERR_INVALID, "gp is NULL");
    if (!g->fitted) return fail(CBO_ERR_NOT_FITTED, "gp is not fitted");
    if (tries_out) *tries_out = g->tries;
    if (jitter_out) *jitter_out = g->jitter;
    return CBO_OK;
}

static void enqueue_factor(cbo_gp *g, double jitter)
{
    cbo_ctx *c = g->ctx;
    {
        PhaseScope ps(c, PH_KXX);
        launch_kxx(c->stream, g->X, g->h, g->noise_var + kGpyDiagJitter, jitter, g->A, g->lda, g->n_pad);
        launch_rhs(c->stream, g->y, g->X.pm, g->n, g->A, g->lda, g->n_pad, g->info, cholesky_info_ints(g->n_pad));
    }
    {
        PhaseScope ps(c, PH_CHOL);
        launch_cholesky(c->stream, c->side_stream, c->chol_events, g->A, g->lda, g->n_pad, g->invDt, g->info, nullptr, true);
    }
}

// Scope of a factorisation's repeat with the separate-launch kernels after a fused launch gave up (kCholFusedTimeout).
struct FusedFallback {
    bool on = false;
    bool active() const { return on; }
    void engage(cbo_ctx *c)
    {
        on = true;
        ++c->fused_fallbacks;
        set_panel_form_override(2);
    }
    ~FusedFallback() { if (on) set_panel_form_override(0); }
};

// GPy util.linalg.jitchol after a failed attempt: first mean(diag) * 1e-6, then x10 per retry, at most 5 retries.
static int next_jitter(cbo_gp *g, int *tries, double *jitter)
{
    if (*tries == 0) {
        // diag of Ky as assembled (jitter-free): variance + v_i + (noise + 1e-8); the kernel's own
        // diagonal differs from this only when zero_diag is off and |x|^2 rounds differently from
        // x.x, i.e. by O(1e-16) relative -- irrelevant for a 1e-6 * mean(diag) jitter.
        // (summed in extended precision and rounded once: numpy's pairwise diagA.mean() is exact for n equal entries,
        //  a sequential double sum of 16384 of them is 4e-13 off)
        long double sum = 0.0L;
        bool nonpos = false;
        for (int64_t i = 0; i < g->n; ++i) {
            const double dv = g->h.variance + (g->h_pv.empty() ? 0.0 : g->h_pv[i]) + (g->noise_var + kGpyDiagJitter);
            if (!(dv > 0.0)) nonpos = true;
            sum += (long double)dv;
        }
        if (nonpos) return fail(CBO_ERR_NONPOS_DIAG, "not pd: non-positive diagonal elements");
        *jitter = (double)(sum / (long double)g->n) * 1e-6;
    } else {
        *jitter *= 10.0;
    }
    ++*tries;
    if (*tries > 5 || !std::isfinite(*jitter))
        return fail(CBO_ERR_NOT_PD, "not positive definite, even with jitter.");
    return CBO_OK;
}

// one attempt at the factorisation with `jitter` on the diagonal: *pd says whether it went through
static int attempt_factor(cbo_gp *g, double jitter, bool *pd)
{
    cbo_ctx *c = g->ctx;
    FusedFallback fallback;
    for (;;) {
        enqueue_factor(g, jitter);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(c->h_info, g->info, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (*c->h_info == kCholFusedTimeout) {
            // a strip of a fused diagonal + panel launch gave up waiting (see potrf_panel_fused_kernel): the same
            // attempt again with the separate-launch kernels -- same bits, no protocol between workgroups
            if (fallback.active())
                return fail(CBO_ERR_HIP, "a fused diagonal + panel launch gave up waiting, and so did the separate-launch repeat");
            fallback.engage(c);
            continue;
        }
        *pd = *c->h_info == 0;
        return CBO_OK;
    }
}

// the factor in g->A is the model's: what every consumer of a fitted model expects beside it
static int adopt_factor(cbo_gp *g, int tries, double jitter)
{
    cbo_ctx *c = g->ctx;
    // contiguous z = L^-1 (y - m) for the sweep: the posterior mean is (L^-1 k*)^T z, so the backward
    // solve for alpha = L^-T z is not on the sweep's path and is materialised on first use (ensure_alpha)
    HIP_TRY(hipMemcpy2DAsync(g->z, sizeof(double), g->A + g->n_pad, sizeof(double) * g->lda, sizeof(double),
                             (size_t)g->n_pad, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipGetLastError());
    g->fitted = true;
    g->fit_stamp = ++g_fit_stamp;
    g->parent_stamp = 0;
    g->alpha_ready = false;
    g->tries = tries;
    g->jitter = jitter;
    if (c->profiling) c->timers.n_fit += 1;
    return CBO_OK;
}

extern "C" int cbo_gp_fit(cbo_gp *g, int *tries_out, double *jitter_out)
{
    if (!g) return fail(CBO_ERR_INVALID, "gp is NULL");
    if (g->n <= 0 || g->n_pad <= 0) return fail(CBO_ERR_INVALID, "gp holds no data (a previous upload failed)");
    cbo_ctx *c = g->ctx;
    HIP_TRY(hipSetDevice(c->device));
    g->fitted = false;
    // GPy util.linalg.jitchol: plain attempt, then mean(diag)*1e-6 jitter, x10 per retry, <= 5 retries.
    double jitter = 0.0;
    int tries = 0;
    for (;;) {
        bool pd = false;
        int rc = attempt_factor(g, jitter, &pd);
        if (rc != CBO_OK) return rc;
        if (pd) break;
        rc = next_jitter(g, &tries, &jitter);
        if (rc != CBO_OK) return rc;
    }
    const int rc = adopt_factor(g, tries, jitter);
    if (rc != CBO_OK) return rc;
    if (tries_out) *tries_out = tries;
    if (jitter_out) *jitter_out = jitter;
    return CBO_OK;
}

// The jitter of level `level` of jitchol's ladder (0: none; k: mean(diag) * 1e-6 * 10^(k-1), by the same repeated
// multiplication as the retries of cbo_gp_fit): CBO_ERR_NONPOS_DIAG / CBO_ERR_NOT_PD as the ladder reports them.
static int ladder_jitter(cbo_gp *g, int level, double *jitter)
{
    *jitter = 0.0;
    int tries = 0;
    while (tries < level) {
        const int rc = next_jitter(g, &tries, jitter);
        if (rc != CBO_OK) return rc;
    }
    return CBO_OK;
}

// ONE level of the ladder, for ranks that walk it side by side (cbo_with_oop_amd/sharding.py, fit_over_ranks: with the
// posterior replicated on G ranks, rank r tries level r while the others try theirs, instead of every rank trying them
// all in turn).  *status: 1 = factored at this level (the model is fitted, tries = level), 0 = not positive definite at
// this level, -1 = the diagonal has non-positive entries (jitchol gives up before its first retry: levels >= 1 only).
// A level beyond jitchol's five retries is CBO_ERR_NOT_PD.
extern "C" int cbo_gp_fit_level(cbo_gp *g, int level, int *status, double *jitter_out)
{
    if (!g || !status) return fail(CBO_ERR_INVALID, "NULL argument");
    if (g->n <= 0 || g->n_pad <= 0) return fail(CBO_ERR_INVALID, "gp holds no data (a previous upload failed)");
    if (level < 0) return fail(CBO_ERR_INVALID, "level must be >= 0");
    cbo_ctx *c = g->ctx;
    HIP_TRY(hipSetDevice(c->device));
    g->fitted = false;
    double jitter = 0.0;
    int rc = ladder_jitter(g, level, &jitter);
    if (rc == CBO_ERR_NONPOS_DIAG) { *status = -1; return CBO_OK; }
    if (rc != CBO_OK) return rc;
    bool pd = false;
    rc = attempt_factor(g, jitter, &pd);
    if (rc != CBO_OK) return rc;
    *status = pd ? 1 : 0;
    if (jitter_out) *jitter_out = jitter;
    return pd ? adopt_factor(g, level, jitter) : CBO_OK;
}

// for cbo_comm_share_factor (cbo_comm.hip): where the factor lives, and its adoption by a rank that received it
namespace cbo {
int gp_factor_view(cbo_gp *g, double **A, int64_t *lda, int64_t *n_pad, double **invDt, cbo_ctx **ctx)
{
    if (!g) return fail(CBO_ERR_INVALID, "gp is NULL");
    if (g->n <= 0 || g->n_pad <= 0) return fail(CBO_ERR_INVALID, "gp holds no data");
    *A = g->A; *lda = g->lda; *n_pad = g->n_pad; *invDt = g->invDt; *ctx = g->ctx;
    return CBO_OK;
}
bool gp_is_fitted_at(const cbo_gp *g, int level) { return g && g->fitted && g->tries == level; }
int gp_adopt_received_factor(cbo_gp *g, int level)
{
    HIP_TRY(hipSetDevice(g->ctx->device));
    double jitter = 0.0;
    const int rc = ladder_jitter(g, level, &jitter);
    if (rc != CBO_OK) return rc;
    return adopt_factor(g, level, jitter);
}
}  // namespace cbo

// GPy's woodbury_vector alpha = Ky^-1 (y - m) = L^-T z (dpotrs): needed by posterior export and by
// prediction gradients, not by predict / the acquisition sweep.
static int ensure_alpha(cbo_gp *g)
{
    if (g->alpha_ready) return CBO_OK;
    cbo_ctx *c = g->ctx;
    bool chained;
    {
        PhaseScope ps(c, PH_ALPHA);
        // one launch (a chain of workgroups); the per-block launches where that form does not apply
        chained = launch_backsolve_chain(c->stream, g->A, g->lda, g->n_pad, g->invDt, g->A + g->n_pad, g->lda,
                                         g->alpha + g->n_pad, g->alpha, g->info);
        if (!chained) launch_backsolve(c->stream, g->A, g->lda, g->n_pad, g->invDt, g->alpha);
    }
    HIP_TRY(hipGetLastError());
    if (chained) {
        // the chain's polls are bounded: a give-up is in the status word, and the solve is repeated the old way
        HIP_TRY(hipMemcpyAsync(c->h_info, g->info, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (*c->h_info != 0) {
            HIP_TRY(hipMemsetAsync(g->info, 0, sizeof(int), c->stream));
            ++c->fused_fallbacks;
            PhaseScope ps(c, PH_ALPHA);
            launch_backsolve(c->stream, g->A, g->lda, g->n_pad, g->invDt, g->alpha);
            HIP_TRY(hipGetLastError());
        }
    }
    g->alpha_ready = true;
    return CBO_OK;
}

extern "C" int cbo_gp_set_data(cbo_gp *g, int64_t n, const double *X, const double *y, const double *pm,
                               const double *pv)
{
    if (!g) return fail(CBO_ERR_INVALID, "gp is NULL");
    HIP_TRY(hipSetDevice(g->ctx->device));
    HIP_TRY(hipStreamSynchronize(g->ctx->stream));
    const int rc = upload_gp_data(g, n, X, y, pm, pv);
    if (rc != CBO_OK) return rc;
    return cbo_gp_fit(g, nullptr, nullptr);
}

extern "C" int cbo_gp_upload_data(cbo_gp *g, int64_t n, const double *X, const double *y, const double *pm,
                                  const double *pv)
{
    if (!g) return fail(CBO_ERR_INVALID, "gp is NULL");
    HIP_TRY(hipSetDevice(g->ctx->device));
    HIP_TRY(hipStreamSynchronize(g->ctx->stream));
    return upload_gp_data(g, n, X, y, pm, pv);      // leaves the model unfitted
}

static int ensure_export(cbo_ctx *c, size_t elems)
{
    if (elems > c->export_elems) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        hipFree(c->export_buf);
        c->export_buf = nullptr; c->export_elems = 0;
        HIP_TRY(hipMalloc(&c->export_buf, sizeof(double) * elems));
        c->export_elems = elems;
    }
    return CBO_OK;
}

extern "C" int cbo_gp_get_posterior(cbo_gp *g, double *L_out, double *alpha_out)
{
    if (!g) return fail(CBO_ERR_INVALID, "gp is NULL");
    if (!g->fitted) return fail(CBO_ERR_NOT_FITTED, "gp is not fitted");
    cbo_ctx *c = g->ctx;
    HIP_TRY(hipSetDevice(c->device));
    if (L_out) {
        int rc = ensure_export(c, (size_t)g->n * (size_t)g->n);
        if (rc != CBO_OK) return rc;
        launch_export_lower(c->stream, g->A, g->lda, g->n, c->export_buf);
        HIP_TRY(hipMemcpyAsync(L_out, c->export_buf, sizeof(double) * g->n * g->n, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    if (alpha_out) {
        const int rc = ensure_alpha(g);
        if (rc != CBO_OK) return rc;
        HIP_TRY(hipMemcpyAsync(alpha_out, g->alpha, sizeof(double) * g->n, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return CBO_OK;
}

extern "C" int cbo_gp_assemble_kxx(cbo_gp *g, double *K_out)
{
    if (!g || !K_out) return fail(CBO_ERR_INVALID, "NULL argument");
    if (g->n <= 0 || g->n_pad <= 0) return fail(CBO_ERR_INVALID, "gp holds no data (a previous upload failed)");
    cbo_ctx *c = g->ctx;
    HIP_TRY(hipSetDevice(c->device));
    // scratch: [n_pad x lda] assembly + [n x n] symmetric export, from the context's grow-only export buffer
    const size_t a_elems = (size_t)g->n_pad * (size_t)g->lda;
    int rc = ensure_export(c, a_elems + (size_t)g->n * (size_t)g->n);
    if (rc != CBO_OK) return rc;
    double *Atmp = c->export_buf, *tmp = c->export_buf + a_elems;
    launch_kxx(c->stream, g->X, g->h, g->noise_var + kGpyDiagJitter, 0.0, Atmp, g->lda, g->n_pad);
    launch_export_sym(c->stream, Atmp, g->lda, g->n, tmp);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(K_out, tmp, sizeof(double) * g->n * g->n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CBO_OK;
}

// ---- candidates ----------------------------------------------------------------------------------
// buffers for m points of dimension d (with prior closures if `prior`): grow-only, nothing happens when they fit
static int cands_reserve(cbo_cands *k, int64_t m, int d, bool prior)
{
    cbo_ctx *c = k->ctx;
    const int64_t m_pad = round_up(m, kStrip);
    if (m_pad > k->cap_m_pad || d > k->cap_d || (prior && !k->cap_prior)) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        const int64_t cap = m_pad > k->cap_m_pad ? m_pad : k->cap_m_pad;
        const int cd = d > k->cap_d ? d : k->cap_d;
        const bool cp = prior || k->cap_prior;
        hipFree(k->raw); hipFree(k->P.xs); hipFree(k->P.sq); hipFree(k->P.sv); hipFree(k->pm); hipFree(k->pv);
        hipFree(k->q); hipFree(k->mu);
        k->raw = k->P.xs = k->P.sq = k->P.sv = k->pm = k->pv = k->q = k->mu = nullptr;
        k->cap_m_pad = 0; k->cap_d = 0; k->cap_prior = false; k->fit_stamp = 0;
        hipError_t e = hipMalloc(&k->raw, sizeof(double) * cap * cd);
        if (e == hipSuccess) e = hipMalloc(&k->P.xs, sizeof(double) * cd * cap);
        if (e == hipSuccess) e = hipMalloc(&k->P.sq, sizeof(double) * cap);
        if (e == hipSuccess && cp) e = hipMalloc(&k->P.sv, sizeof(double) * cap);
        if (e == hipSuccess && cp) e = hipMalloc(&k->pm, sizeof(double) * cap);
        if (e == hipSuccess && cp) e = hipMalloc(&k->pv, sizeof(double) * cap);
        if (e != hipSuccess) return fail(CBO_ERR_HIP, std::string("candidate buffers: ") + hipGetErrorString(e));
        k->cap_m_pad = cap; k->cap_d = cd; k->cap_prior = cp;
    }
    return CBO_OK;
}

// (re)fill a candidate set from host arrays; every cache keyed on its old contents is dropped
static int cands_fill(cbo_cands *k, int64_t m, int d, const double *Xs, const double *pm, const double *pv,
                      int64_t index_offset)
{
    cbo_ctx *c = k->ctx;
    int rc = cands_reserve(k, m, d, pv != nullptr);
    if (rc != CBO_OK) return rc;
    k->m = m; k->d = d; k->index_offset = index_offset;
    k->m_pad = round_up(m, kStrip);
    k->P.n = m; k->P.ld = k->m_pad; k->P.d = d;
    k->has_prior = pv != nullptr;
    k->prepared_for = nullptr; k->prepared_ls.clear();
    k->fit_stamp = 0; k->v_stamp = 0;
    hipError_t e = hipMemcpyAsync(k->raw, Xs, sizeof(double) * m * d, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && pv) e = hipMemcpyAsync(k->pm, pm, sizeof(double) * m, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && pv) e = hipMemcpyAsync(k->pv, pv, sizeof(double) * m, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);           // the host buffers are the caller's
    if (e != hipSuccess) return fail(CBO_ERR_HIP, std::string("candidate upload: ") + hipGetErrorString(e));
    return CBO_OK;
}

static int check_cands_args(cbo_ctx *c, int64_t m, int d, const double *Xs, const double *pm, const double *pv)
{
    if (!c || !Xs) return fail(CBO_ERR_INVALID, "NULL argument");
    if (m <= 0) return fail(CBO_ERR_INVALID, "m must be positive");
    if (d < 1 || d > CBO_MAX_DIM) return fail(CBO_ERR_INVALID, "d must be in [1, CBO_MAX_DIM]");
    if ((pm == nullptr) != (pv == nullptr))
        return fail(CBO_ERR_INVALID, "prior mean and prior variance must be given together");
    return CBO_OK;
}

extern "C" int cbo_cands_create(cbo_ctx *c, int64_t m, int d, const double *Xs, const double *pm, const double *pv,
                                int64_t index_offset, cbo_cands **out)
{
    if (!out) return fail(CBO_ERR_INVALID, "NULL argument");
    int rc = check_cands_args(c, m, d, Xs, pm, pv);
    if (rc != CBO_OK) return rc;
    HIP_TRY(hipSetDevice(c->device));
    cbo_cands *k = new cbo_cands();
    k->ctx = c;
    rc = cands_fill(k, m, d, Xs, pm, pv, index_offset);
    if (rc != CBO_OK) { cbo_cands_destroy(k); return rc; }
    *out = k;
    return CBO_OK;
}

// the context's reusable candidate set for the host-buffer entry points (no allocation once it has grown)
static int scratch_cands(cbo_ctx *c, int64_t m, int d, const double *Xs, const double *pm, const double *pv,
                         cbo_cands **out)
{
    int rc = check_cands_args(c, m, d, Xs, pm, pv);
    if (rc != CBO_OK) return rc;
    HIP_TRY(hipSetDevice(c->device));
    if (!c->scratch_k) { c->scratch_k = new cbo_cands(); c->scratch_k->ctx = c; }
    rc = cands_fill(c->scratch_k, m, d, Xs, pm, pv, 0);
    if (rc != CBO_OK) return rc;
    *out = c->scratch_k;
    return CBO_OK;
}

extern "C" void cbo_cands_destroy(cbo_cands *k)
{
    if (!k) return;
    hipSetDevice(k->ctx->device);
    hipStreamSynchronize(k->ctx->stream);
    hipFree(k->raw); hipFree(k->pm); hipFree(k->pv); hipFree(k->q); hipFree(k->mu); hipFree(k->V); hipFree(k->partial);
    hipFree(k->P.xs); hipFree(k->P.sq); hipFree(k->P.sv);
    delete k;
}

static int ensure_workspaces(cbo_ctx *c, int64_t n_pad, int64_t m_pad, int64_t *chunk_cols, int64_t *ldv,
                             size_t elem = sizeof(double))
{
    // V chunk: as many 64-column strips as fit the workspace budget (at least one strip).  elem = 4: the fp32
    // sweep's workspace (n_pad is then its 256-padded row count), 128 B of row padding either way.
    int64_t cols = m_pad;
    const int64_t max_cols = (int64_t)(c->max_ws_bytes / (elem * (size_t)n_pad)) / kStrip * kStrip;
    if (cols > max_cols) cols = max_cols < kStrip ? kStrip : max_cols;
    const int64_t ld = cols + (int64_t)(128 / elem);
    const size_t need = elem * (size_t)n_pad * (size_t)ld;
    if (need > c->V_bytes) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        hipFree(c->V);
        c->V = nullptr; c->V_bytes = 0;
        HIP_TRY(hipMalloc(&c->V, need));
        c->V_bytes = need;
    }
    if ((size_t)m_pad > c->vec_elems) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        hipFree(c->q); hipFree(c->mu); hipFree(c->mean); hipFree(c->var); hipFree(c->acq);
        c->q = c->mu = c->mean = c->var = c->acq = nullptr; c->vec_elems = 0;
        HIP_TRY(hipMalloc(&c->q, sizeof(double) * m_pad));
        HIP_TRY(hipMalloc(&c->mu, sizeof(double) * m_pad));
        HIP_TRY(hipMalloc(&c->mean, sizeof(double) * m_pad));
        HIP_TRY(hipMalloc(&c->var, sizeof(double) * m_pad));
        HIP_TRY(hipMalloc(&c->acq, sizeof(double) * m_pad));
        c->vec_elems = (size_t)m_pad;
    }
    *chunk_cols = cols;
    *ldv = ld;
    return CBO_OK;
}

// Scale / transpose the candidate coordinates for this GP's lengthscales (GPy ARD divides the inputs).
static int prepare_cands(cbo_gp *g, cbo_cands *k)
{
    cbo_ctx *c = g->ctx;
    if (k->prepared_for == g && k->prepared_ls == g->ls) return CBO_OK;
    launch_prep_points(c->stream, k->raw, k->m, k->d, g->h.ard ? g->ls_dev : nullptr, k->has_prior ? k->pv : nullptr,
                       k->P.xs, k->m_pad, k->P.sq, k->has_prior ? k->P.sv : nullptr);
    HIP_TRY(hipGetLastError());
    k->prepared_for = g;
    k->prepared_ls = g->ls;
    return CBO_OK;
}

// timers of the right-looking sweep: one event pair per launch on the stream it goes to
static void pipe_mark(void *user, hipStream_t st, int begin, double flops)
{
    cbo_ctx *c = static_cast<cbo_ctx *>(user);
    if (!c->profiling) return;
    EventPair &cur = c->pipe_cur;
    if (begin) {
        cur.a = get_event(c);
        cur.b = get_event(c);
        cur.phase = PH_TRSM;
        hipEventRecord(cur.a, st);
        c->timers.n_trsm_launches += 1;
        c->timers.trsm_flops += flops;
    } else {
        hipEventRecord(cur.b, st);
        c->pending.push_back(cur);
    }
}

static SweepPipe make_pipe(cbo_gp *g, double *V, int64_t ldv, int64_t cols, double *q, double *mu)
{
    cbo_ctx *c = g->ctx;
    SweepPipe pipe{};
    pipe.stream = c->sweep_stream;
    pipe.bulk = c->bulk_stream;
    pipe.V = V; pipe.ldv = ldv; pipe.m_pad = cols;
    pipe.zvec = g->z; pipe.q = q; pipe.mu = mu;
    pipe.chunk_blocks = c->pipe_chunk_blocks;
    pipe.half_lds = c->pipe_half_lds;
    pipe.events = &c->pipe_events;
    pipe.mark = pipe_mark; pipe.user = c;
    pipe.tail_begin = (int)g->n_pad;                     // no tail unless the caller sets one
    pipe.group = 0;
    return pipe;
}

// Which schedule for a sweep of `cols` candidate columns against a factor that is already complete?
// The left-looking strip kernel is the more efficient one (no read-modify-write of V, no launch chain) but it
// has one workgroup per 64 columns: its time is whole rounds of n_cu strips.  The right-looking schedule
// (strip kernel on a panel pair, then trsm_update_kernel over strips x row chunks, pair after pair) fills the
// device whatever the column count and costs about a sixth more per column.  Same bits either way.
static bool prefer_right_looking(const cbo_ctx *c, int64_t n_pad, int64_t cols)
{
    if (c->sweep_mode == 0) return false;
    if (c->sweep_mode == 1) return true;
    if (n_pad < 1024) return false;
    const int64_t strips = cols / kStrip;
    const int64_t rounds = (strips + c->n_cu - 1) / c->n_cu;
    return rounds * c->n_cu * 5 >= strips * 6;
}

// ---- the schedule of cbo_gp_fit_sweep is measured on the calls the caller makes: schedule_tuner.h ----------------------

static int enqueue_right_looking(cbo_gp *g, double *V, int64_t ldv, int64_t cols, double *q, double *mu,
                                 bool lower_tri = false)
{
    cbo_ctx *c = g->ctx;
    SweepPipe pipe = make_pipe(g, V, ldv, cols, q, mu);
    pipe.lower_tri = lower_tri;
    HIP_TRY(hipMemsetAsync(q, 0, sizeof(double) * cols, c->stream));
    HIP_TRY(hipMemsetAsync(mu, 0, sizeof(double) * cols, c->stream));
    int p = 0;
    for (int r0 = 0; r0 < (int)g->n_pad; r0 += 256, ++p)
        sweep_pipe_pair(pipe, c->stream, g->A, g->lda, g->invDt, g->n_pad, p, r0,
                        (r0 + 256 <= (int)g->n_pad) ? 256 : 128);
    HIP_TRY(hipEventRecord(c->ev_join, c->sweep_stream));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_join, 0));
    HIP_TRY(hipEventRecord(c->ev_join2, c->bulk_stream));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_join2, 0));
    return CBO_OK;
}

// The candidates' own V buffer (cbo_cands_keep_solution), sized for the model's padded row count.
static int own_solution_buffer(cbo_gp *g, cbo_cands *k, double **V, int64_t *ldv)
{
    cbo_ctx *c = g->ctx;
    if (!k->V || k->v_rows_cap != g->n_pad || k->v_ld != k->m_pad + kLdExtra) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        hipFree(k->V); hipFree(k->partial);
        k->V = nullptr; k->partial = nullptr; k->v_stamp = 0;
        k->v_ld = k->m_pad + kLdExtra;
        HIP_TRY(hipMalloc(&k->V, sizeof(double) * (size_t)g->n_pad * (size_t)k->v_ld));
        HIP_TRY(hipMalloc(&k->partial, sizeof(double) * 64 * (size_t)k->m_pad));
        k->v_rows_cap = g->n_pad;
    }
    *V = k->V;
    *ldv = k->v_ld;
    return CBO_OK;
}

// fp32 sweep (CBO_DTYPE_F32 models): K* in fp64 arithmetic rounded to fp32, substitution on the f32 MFMA, q and mu
// accumulated in fp64.  Left-looking strip kernel only; the fp32 copies of the factor follow the fit lazily.
static int ensure_alpha(cbo_gp *g);

static int enqueue_posterior_f32(cbo_gp *g, cbo_cands *k)
{
    cbo_ctx *c = g->ctx;
    int64_t chunk = 0, ldv = 0;
    int rc = ensure_workspaces(c, g->n32, k->m_pad, &chunk, &ldv, sizeof(float));
    if (rc != CBO_OK) return rc;
    rc = ensure_alpha(g);                     // the mean is K*^T alpha in fp64 (GPy's formula), see kernels_f32.hip
    if (rc != CBO_OK) return rc;
    const size_t part_elems = (size_t)(g->n32 / 64) * (size_t)chunk;
    if (part_elems > c->mupart_elems) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        hipFree(c->mupart);
        c->mupart = nullptr; c->mupart_elems = 0;
        HIP_TRY(hipMalloc(&c->mupart, sizeof(double) * part_elems));
        c->mupart_elems = part_elems;
    }
    if (g->f32_stamp != g->fit_stamp) {
        PhaseScope ps(c, PH_CONVERT);
        launch_factor_to_f32(c->stream, g->A, g->lda, g->n_pad, g->invDt, g->Uf, g->ldu32, g->invF, g->n32);
        g->f32_stamp = g->fit_stamp;
    }
    float *Vf = reinterpret_cast<float *>(c->V);
    k->v_stamp = 0;
    for (int64_t c0 = 0; c0 < k->m_pad; c0 += chunk) {
        const int64_t cols = (k->m_pad - c0 < chunk) ? (k->m_pad - c0) : chunk;
        {
            PhaseScope ps(c, PH_KSTAR);
            launch_kstar_f32(c->stream, g->X, k->P, c0, cols, g->h, Vf, ldv, g->n32, g->alpha, c->mupart, c->mu + c0);
        }
        {
            PhaseScope ps(c, PH_TRSM);
            launch_trsm_strips_f32(c->stream, g->Uf, g->ldu32, g->invF, Vf, ldv, g->n32, cols, c->q + c0);
        }
        if (c->profiling) {
            c->timers.n_trsm_launches += 1;
            c->timers.trsm_flops += (double)g->n32 * (double)g->n32 * (double)cols;
        }
    }
    HIP_TRY(hipGetLastError());
    return CBO_OK;
}

// q = colsum((L^-1 K*)^2), mu = (L^-1 K*)^T z for all candidates, chunk by chunk.
static int enqueue_posterior(cbo_gp *g, cbo_cands *k, bool want_f64_solution = false)
{
    cbo_ctx *c = g->ctx;
    int rc = prepare_cands(g, k);
    if (rc != CBO_OK) return rc;
    // (prediction gradients read V = L^-1 K* back from the fp64 workspace: they stay on the fp64 factor)
    if (g->dtype == CBO_DTYPE_F32 && !want_f64_solution) return enqueue_posterior_f32(g, k);
    int64_t chunk = 0, ldv = 0;
    rc = ensure_workspaces(c, g->n_pad, k->m_pad, &chunk, &ldv);
    if (rc != CBO_OK) return rc;
    double *Vws = c->V;
    k->v_stamp = 0;
    if (k->keep_v && chunk >= k->m_pad) {                    // one chunk: the solution can stay with the candidates
        rc = own_solution_buffer(g, k, &Vws, &ldv);
        if (rc != CBO_OK) return rc;
    }
    for (int64_t c0 = 0; c0 < k->m_pad; c0 += chunk) {
        const int64_t cols = (k->m_pad - c0 < chunk) ? (k->m_pad - c0) : chunk;
        {
            PhaseScope ps(c, PH_KSTAR);
            launch_kstar(c->stream, g->X, k->P, c0, cols, g->h, Vws, ldv, g->n_pad);
        }
        if (prefer_right_looking(c, g->n_pad, cols)) {
            rc = enqueue_right_looking(g, Vws, ldv, cols, c->q + c0, c->mu + c0);
            if (rc != CBO_OK) return rc;
            continue;
        }
        {
            PhaseScope ps(c, PH_TRSM);
            launch_trsm_strips(c->stream, g->A, g->lda, g->invDt, Vws, ldv, g->n_pad, cols, g->z, c->q + c0,
                               c->mu + c0);
        }
        if (c->profiling) {
            c->timers.n_trsm_launches += 1;
            c->timers.trsm_flops += (double)g->n_pad * (double)g->n_pad * (double)cols;
        }
    }
    HIP_TRY(hipGetLastError());
    if (Vws != c->V) { k->v_stamp = g->fit_stamp; k->v_rows = g->n; }
    return CBO_OK;
}

static int check_sweep_args(const cbo_gp *g, const cbo_cands *k, int task)
{
    if (!g || !k) return fail(CBO_ERR_INVALID, "NULL argument");
    if (g->ctx != k->ctx) return fail(CBO_ERR_INVALID, "gp and candidates live on different contexts");
    if (g->d != k->d) return fail(CBO_ERR_INVALID, "gp and candidates have different dimensions");
    if ((g->X.sv != nullptr) && !k->has_prior)
        return fail(CBO_ERR_INVALID, "causal gp needs candidate prior mean/variance");
    if (task != CBO_TASK_MIN && task != CBO_TASK_MAX) return fail(CBO_ERR_INVALID, "task must be 0 (min) or 1 (max)");
    return CBO_OK;
}

// EI / cost and arg-max from q, mu (already on the device), results to the host
// The epilogue of a sweep from q = sum V^2 and mu = V^T z: variance, mean, EI / cost, arg-max, and the copies to the host.
// enqueue_finish only queues (cbo_gp_fit_sweep queues it behind the closing launch, ahead of its one synchronisation:
// the factorisation's status and the winner come back together); complete_finish reads the winner after the stream has
// been synchronised.
// the per-candidate vectors of the epilogue to the caller's host buffers (queued; the caller synchronises)
static int copy_posterior_out(cbo_ctx *c, const cbo_cands *k, double *acq_out, double *mean_out, double *var_out)
{
    if (acq_out) HIP_TRY(hipMemcpyAsync(acq_out, c->acq, sizeof(double) * k->m, hipMemcpyDeviceToHost, c->stream));
    if (mean_out) HIP_TRY(hipMemcpyAsync(mean_out, c->mean, sizeof(double) * k->m, hipMemcpyDeviceToHost, c->stream));
    if (var_out) HIP_TRY(hipMemcpyAsync(var_out, c->var, sizeof(double) * k->m, hipMemcpyDeviceToHost, c->stream));
    return CBO_OK;
}

static int enqueue_finish(cbo_gp *g, cbo_cands *k, double y_best, int task, double ei_jitter, double cost, double *acq_out,
                          double *mean_out, double *var_out, const double *q_src, const double *mu_src,
                          bool with_status = false, bool copy_out = true)
{
    cbo_ctx *c = g->ctx;
    const bool causal = g->X.sv != nullptr;
    AcqParams p;
    p.variance = g->h.variance; p.noise_var = g->noise_var; p.y_best = y_best; p.ei_jitter = ei_jitter; p.cost = cost;
    p.task = task; p.include_noise = 1; p.want_ei = 1;
    const int nb = acq_blocks_for(k->m);
    {
        PhaseScope ps(c, PH_ACQ);
        launch_acq(c->stream, q_src, mu_src, causal ? k->pm : nullptr, causal ? k->pv : nullptr, k->m, p,
                   mean_out ? c->mean : nullptr, var_out ? c->var : nullptr, acq_out ? c->acq : nullptr, c->part_val,
                   c->part_idx, k->index_offset, nb);
        // the winner (and, behind a factorisation, its status word) goes straight to pinned host memory
        launch_argmax_final(c->stream, c->part_val, c->part_idx, nb, c->h_best_val, c->h_best_idx,
                            with_status ? g->info : nullptr, with_status ? c->h_info : nullptr);
    }
    HIP_TRY(hipGetLastError());
    if (copy_out) return copy_posterior_out(c, k, acq_out, mean_out, var_out);
    return CBO_OK;
}

static void complete_finish(cbo_ctx *c, double *best_val, int64_t *best_idx)
{
    if (best_val) *best_val = *c->h_best_val;
    if (best_idx) *best_idx = *c->h_best_idx;
    if (c->profiling) c->timers.n_sweep += 1;
}

static int finish_sweep(cbo_gp *g, cbo_cands *k, double y_best, int task, double ei_jitter, double cost,
                        double *acq_out, double *mean_out, double *var_out, double *best_val, int64_t *best_idx)
{
    cbo_ctx *c = g->ctx;
    // keep q, mu with the candidates (two small device copies): the next sweep of an unchanged model skips the
    // substitution altogether
    const bool cached = k->fit_stamp != 0 && k->fit_stamp == g->fit_stamp;
    if (!cached && c->sweep_cache) {
        if (!k->q) {
            HIP_TRY(hipMalloc(&k->q, sizeof(double) * k->cap_m_pad));
            HIP_TRY(hipMalloc(&k->mu, sizeof(double) * k->cap_m_pad));
        }
        HIP_TRY(hipMemcpyAsync(k->q, c->q, sizeof(double) * k->m_pad, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(k->mu, c->mu, sizeof(double) * k->m_pad, hipMemcpyDeviceToDevice, c->stream));
        k->fit_stamp = g->fit_stamp;
    }
    const double *q_src = cached ? k->q : c->q, *mu_src = cached ? k->mu : c->mu;
    int rc = enqueue_finish(g, k, y_best, task, ei_jitter, cost, acq_out, mean_out, var_out, q_src, mu_src);
    if (rc != CBO_OK) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    complete_finish(c, best_val, best_idx);
    return CBO_OK;
}

// V[n-1, :] for a model that was extended by cbo_gp_append: k(x_new, X*) by the K* kernel with the appended point as
// its only row, then the row update; q, mu (the candidates' cached copies) move along and take the new fit stamp.
static int extend_solution_by_one_row(cbo_gp *g, cbo_cands *k)
{
    cbo_ctx *c = g->ctx;
    int rc = prepare_cands(g, k);
    if (rc != CBO_OK) return rc;
    int64_t chunk = 0, ldv = 0;
    rc = ensure_workspaces(c, g->n_pad, k->m_pad, &chunk, &ldv);      // c->V: scratch for the 64-row K* slab
    if (rc != CBO_OK) return rc;
    if (chunk < k->m_pad) return CBO_OK;                               // cannot happen for a resident V; fall through
    const int64_t row = g->n - 1;
    PointSet one = g->probe->P;                                        // the appended point, scaled as the model's
    one.n = 1;
    launch_kstar(c->stream, one, k->P, 0, k->m_pad, g->h, c->V, ldv, 64);
    launch_append_row(c->stream, k->V, k->v_ld, row, g->lvec, k->m_pad, c->V, g->append_d, g->append_zn, k->partial,
                      k->q, k->mu);
    HIP_TRY(hipGetLastError());
    k->v_rows = g->n;
    k->v_stamp = g->fit_stamp;
    k->fit_stamp = g->fit_stamp;
    return CBO_OK;
}

extern "C" int cbo_cands_keep_solution(cbo_cands *k, int on)
{
    if (!k) return fail(CBO_ERR_INVALID, "cands is NULL");
    k->keep_v = on != 0;
    if (!k->keep_v) {
        hipSetDevice(k->ctx->device);
        hipStreamSynchronize(k->ctx->stream);
        hipFree(k->V); hipFree(k->partial);
        k->V = nullptr; k->partial = nullptr; k->v_stamp = 0; k->v_rows_cap = 0;
    }
    return CBO_OK;
}

// One more observation for a fitted model (the step src/Monitor.py:148-160 + src/CBO.py:224-235 take every trial):
// the factor, z and the resident data grow by one row instead of being rebuilt.  *appended_out = 0 (and nothing
// changed) when the shortcut does not apply -- the current factor carries jitter, the padded size is exhausted, or
// the new pivot is not positive -- and the caller refits with cbo_gp_set_data.
extern "C" int cbo_gp_append(cbo_gp *g, const double *x_new, double y_new, double prior_mean_new, double prior_var_new,
                             int *appended_out)
{
    if (!g || !x_new || !appended_out) return fail(CBO_ERR_INVALID, "NULL argument");
    *appended_out = 0;
    if (!g->fitted) return fail(CBO_ERR_NOT_FITTED, "gp is not fitted");
    if (g->tries != 0 || g->n >= g->n_pad) return CBO_OK;
    if (g->dtype != CBO_DTYPE_F64) return CBO_OK;       // the resident V of the row update is an fp64 object
    cbo_ctx *c = g->ctx;
    HIP_TRY(hipSetDevice(c->device));
    const bool causal = g->X.sv != nullptr;
    // the new point as a one-candidate set (kept with the model: no allocation after the first append)
    if (g->probe) {
        cbo_cands *p = g->probe;
        HIP_TRY(hipMemcpyAsync(p->raw, x_new, sizeof(double) * g->d, hipMemcpyHostToDevice, c->stream));
        if (causal) {
            HIP_TRY(hipMemcpyAsync(p->pm, &prior_mean_new, sizeof(double), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(p->pv, &prior_var_new, sizeof(double), hipMemcpyHostToDevice, c->stream));
        }
        HIP_TRY(hipStreamSynchronize(c->stream));          // x_new and the two scalars are the caller's / this frame's
        p->prepared_for = nullptr;
        p->fit_stamp = 0;
    } else {
        int rc = cbo_cands_create(c, 1, g->d, x_new, causal ? &prior_mean_new : nullptr, causal ? &prior_var_new : nullptr,
                                  0, &g->probe);
        if (rc != CBO_OK) return rc;
    }
    // k(X, x_new) by the K* kernel (64 padded columns, column 0 is the point), then l = L^-1 k by the
    // single-right-hand-side forward solve; l^T l and l^T z on the host (two n-vectors come back)
    int rc = prepare_cands(g, g->probe);
    if (rc != CBO_OK) return rc;
    int64_t chunk = 0, ldv = 0;
    rc = ensure_workspaces(c, g->n_pad, g->probe->m_pad, &chunk, &ldv);
    if (rc != CBO_OK) return rc;
    launch_kstar(c->stream, g->X, g->probe->P, 0, g->probe->m_pad, g->h, c->V, ldv, g->n_pad);
    launch_gather_column(c->stream, c->V, ldv, g->n_pad, g->alpha + g->n_pad);      // work vector (alpha's scratch half)
    const bool chained = launch_forward_chain(c->stream, g->A, g->lda, g->n_pad, g->invDt, g->alpha + g->n_pad, g->lvec,
                                              g->info);
    if (!chained) launch_forward_vec(c->stream, g->A, g->lda, g->n_pad, g->invDt, g->alpha + g->n_pad, g->lvec);
    HIP_TRY(hipGetLastError());
    // l^T z and l^T l by one device reduction (16 bytes come back)
    launch_dot2(c->stream, g->lvec, g->z, g->n, c->part_val);
    HIP_TRY(hipGetLastError());
    double hd[2];
    HIP_TRY(hipMemcpyAsync(hd, c->part_val, sizeof(hd), hipMemcpyDeviceToHost, c->stream));
    if (chained) HIP_TRY(hipMemcpyAsync(c->h_info, g->info, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (chained && *c->h_info != 0) {                       // the chain gave up (bounded polls): the per-block launches
        HIP_TRY(hipMemsetAsync(g->info, 0, sizeof(int), c->stream));
        ++c->fused_fallbacks;
        launch_forward_vec(c->stream, g->A, g->lda, g->n_pad, g->invDt, g->alpha + g->n_pad, g->lvec);
        launch_dot2(c->stream, g->lvec, g->z, g->n, c->part_val);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(hd, c->part_val, sizeof(hd), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    const double h2[2] = {hd[1], hd[0]};                    // {l^T l, l^T z}
    g->alpha_ready = false;                                 // its scratch half was used
    const double kappa = g->h.variance + (causal ? prior_var_new : 0.0) + (g->noise_var + kGpyDiagJitter);
    const double d2 = kappa - h2[0];
    if (!(d2 > 0.0) || !std::isfinite(d2)) return CBO_OK;  // jitchol's business: full refit
    const double d = std::sqrt(d2);
    const double zn = ((y_new - (causal ? prior_mean_new : 0.0)) - h2[1]) / d;
    HIP_TRY(hipMemcpyAsync(g->raw + g->n * g->d, x_new, sizeof(double) * g->d, hipMemcpyHostToDevice, c->stream));
    launch_append_commit(c->stream, g->A, g->lda, g->n, g->n_pad, g->lvec, 1, d, zn, g->z, g->lvec, g->X, g->probe->P,
                         prior_mean_new, prior_var_new, g->y, y_new, g->invDt);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (causal) g->h_pv.push_back(prior_var_new);
    g->n += 1;
    g->X.n = g->n;
    g->append_d = d;
    g->append_zn = zn;
    g->parent_stamp = g->fit_stamp;
    g->fit_stamp = ++g_fit_stamp;
    g->alpha_ready = false;
    *appended_out = 1;
    return CBO_OK;
}

extern "C" int cbo_acq_sweep(cbo_gp *g, cbo_cands *k, double y_best, int task, double ei_jitter, double cost,
                             double *acq_out, double *mean_out, double *var_out, double *best_val, int64_t *best_idx)
{
    int rc = check_sweep_args(g, k, task);
    if (rc != CBO_OK) return rc;
    if (!g->fitted) return fail(CBO_ERR_NOT_FITTED, "gp is not fitted");
    cbo_ctx *c = g->ctx;
    HIP_TRY(hipSetDevice(c->device));
    if (k->keep_v && k->V && k->v_stamp != 0 && k->v_stamp == g->parent_stamp && k->v_rows == g->n - 1 &&
        k->v_rows_cap == g->n_pad && k->fit_stamp == k->v_stamp && k->q) {
        // the model is the one this V belongs to plus one observation: one new row instead of the substitution
        rc = extend_solution_by_one_row(g, k);
        if (rc != CBO_OK) return rc;
    }
    if (!(c->sweep_cache && k->fit_stamp != 0 && k->fit_stamp == g->fit_stamp)) {
        k->fit_stamp = 0;
        rc = enqueue_posterior(g, k);
        if (rc != CBO_OK) return rc;
    } else if ((size_t)k->m_pad > c->vec_elems) {
        int64_t chunk = 0, ldv = 0;                      // mean / var / acq scratch of the epilogue
        rc = ensure_workspaces(c, g->n_pad, k->m_pad, &chunk, &ldv);
        if (rc != CBO_OK) return rc;
    }
    return finish_sweep(g, k, y_best, task, ei_jitter, cost, acq_out, mean_out, var_out, best_val, best_idx);
}

// Refit and sweep in one call, the two overlapped: what CBO.intervene() does for the set it has just
// intervened on (set_data -> refit, then find_next_y_point -> acquisition over the candidates,
// /root/reference/src/Monitor.py:160, src/CBO.py:250-257).  The factorisation is a chain of short kernels
// that leaves most CUs idle; the sweep's rows become solvable panel by panel as the chain advances, so it
// runs right-looking on a second stream underneath (SweepPipe).  Same results as cbo_gp_fit followed by
// cbo_acq_sweep (the per-element operation order is the same; only q and mu are summed panel-wise).
extern "C" int cbo_gp_fit_sweep(cbo_gp *g, cbo_cands *k, double y_best, int task, double ei_jitter, double cost,
                                double *acq_out, double *mean_out, double *var_out, double *best_val,
                                int64_t *best_idx, int *tries_out, double *jitter_out)
{
    int rc = check_sweep_args(g, k, task);
    if (rc != CBO_OK) return rc;
    if (g->n <= 0 || g->n_pad <= 0) return fail(CBO_ERR_INVALID, "gp holds no data (a previous upload failed)");
    cbo_ctx *c = g->ctx;
    HIP_TRY(hipSetDevice(c->device));
    if (g->dtype == CBO_DTYPE_F32) {
        // the fp32 sweep needs the finished fp64 factor (down-converted once): the plain sequence
        rc = cbo_gp_fit(g, tries_out, jitter_out);
        if (rc != CBO_OK) return rc;
        return cbo_acq_sweep(g, k, y_best, task, ei_jitter, cost, acq_out, mean_out, var_out, best_val, best_idx);
    }
    rc = prepare_cands(g, k);
    if (rc != CBO_OK) return rc;
    int64_t chunk = 0, ldv = 0;
    rc = ensure_workspaces(c, g->n_pad, k->m_pad, &chunk, &ldv);
    if (rc != CBO_OK) return rc;
    if (chunk < k->m_pad) {
        // the candidates do not fit one V workspace: no overlap, the plain sequence
        rc = cbo_gp_fit(g, tries_out, jitter_out);
        if (rc != CBO_OK) return rc;
        return cbo_acq_sweep(g, k, y_best, task, ei_jitter, cost, acq_out, mean_out, var_out, best_val, best_idx);
    }
    // The schedule: forced by the environment (CBO_HIP_OVERLAP / CBO_HIP_PIPE_TAIL / CBO_HIP_PIPE_GROUP: diagnostics and
    // scripts/schedule_scan.py), or the one this context has measured for the shape (schedule_choose above).  A model of
    // one panel pair or less has nothing to pipeline.
    using clk = std::chrono::steady_clock;
    const clk::time_point t_call = clk::now();
    auto us_since = [](clk::time_point a) { return std::chrono::duration<double, std::micro>(clk::now() - a).count(); };
    static const int pipe_group_env = [] {
        const char *e = std::getenv("CBO_HIP_PIPE_GROUP");           // 1 = never grouped, G >= 2 = groups of G pairs
        const int v = e ? std::atoi(e) : 0;                          // (groups beyond 4 pairs -- K = 1024 -- are not
        return v < 0 ? 0 : (v > 4 ? 4 : v);                          //  covered by the tests: clamped)
    }();
    const int nb = (int)(g->n_pad / 128);
    const int all_pairs = (nb + 1) / 2;
    const bool forced = c->overlap_mode == 0 || c->overlap_mode == 1 || c->pipe_tail_frac >= 0.0 || pipe_group_env != 0 ||
                        all_pairs < 2;
    ScheduleEntry *entry = nullptr;
    ScheduleChoice choice;
    if (forced) {
        choice.group = pipe_group_env >= 2 ? pipe_group_env : (pipe_group_env == 0 && k->m_pad / kStrip >= c->n_cu_pipe) ? 2 : 0;
        if (c->overlap_mode == 0 || (all_pairs < 2 && c->overlap_mode != 1)) choice.pairs = kSequence;
        else if (c->pipe_tail_frac >= 0.0) {
            int tail_blocks = (int)(c->pipe_tail_frac * nb + 0.5);
            tail_blocks += (nb - tail_blocks) & 1;
            choice.pairs = (nb - (tail_blocks > nb ? nb : tail_blocks)) / 2;
        } else choice.pairs = all_pairs >= 4 ? all_pairs / 4 : 1;    // (only overlap / grouping forced: a quarter of the rows)
    } else {
        entry = &schedule_entry(c->schedule, c->n_cu_pipe, g->n_pad, k->m_pad / kStrip, k->m_pad);
        choice = schedule_choose(*entry, !c->profiling, c->n_cu, c->n_cu_pipe);
        entry->ran_pairs = choice.pairs;
        entry->ran_group = choice.pairs > 0 ? choice.group : 0;
    }
    if (choice.pairs == kSequence) {
        int tries = 0;
        double jitter = 0.0;
        const clk::time_point t_fit = clk::now();
        rc = cbo_gp_fit(g, &tries, &jitter);
        if (tries_out) *tries_out = tries;
        if (jitter_out) *jitter_out = jitter;
        if (rc != CBO_OK) return rc;
        const double fact_us = us_since(t_fit);
        const clk::time_point t_sweep = clk::now();
        rc = cbo_acq_sweep(g, k, y_best, task, ei_jitter, cost, acq_out, mean_out, var_out, best_val, best_idx);
        if (rc == CBO_OK && entry)
            schedule_report(c->n_cu, c->n_cu_pipe, *entry, choice, tries, us_since(t_call) * 1e-3, fact_us, us_since(t_sweep));
        return rc;
    }
    g->fitted = false;
    double *Vws = c->V;
    k->v_stamp = 0;
    if (k->keep_v) {
        rc = own_solution_buffer(g, k, &Vws, &ldv);
        if (rc != CBO_OK) return rc;
    }
    // q = sum V^2 and mu = V^T z go straight to the candidates' own copies when those are kept (the next sweep of the
    // unchanged model starts from them): no device copies between the closing launch and the acquisition kernel
    double *qbuf = c->q, *mubuf = c->mu;
    if (c->sweep_cache) {
        if (!k->q) {
            HIP_TRY(hipMalloc(&k->q, sizeof(double) * k->cap_m_pad));
            HIP_TRY(hipMalloc(&k->mu, sizeof(double) * k->cap_m_pad));
        }
        qbuf = k->q;
        mubuf = k->mu;
        k->fit_stamp = 0;                                  // not valid until this call has succeeded
    }
    const bool speculate = qbuf == k->q;
    SweepPipe pipe = make_pipe(g, Vws, ldv, k->m_pad, qbuf, mubuf);
    pipe.group = choice.group;
    // pairs that do not fill a group go alone ahead of the first one (CBO_HIP_PIPE_LEAD forces the count where the schedule is
    // forced; there the default stays 0: whole groups from the first pair, what the forced schedules of rounds 3-5 meant)
    static const int pipe_lead_env = [] { const char *e = std::getenv("CBO_HIP_PIPE_LEAD"); return e ? std::atoi(e) : -1; }();
    int pairs = choice.pairs;
    pipe.lead = 0;
    if (pipe.group >= 2 && pairs >= 1 && pairs * 256 < (int)g->n_pad)
        pipe.lead = forced ? (pipe_lead_env > 0 ? pipe_lead_env : 0) : (pipe_lead_env >= 0 ? pipe_lead_env : pairs % pipe.group);
    pipe.tail_begin = pairs * 256;
    if (pipe.tail_begin > (int)g->n_pad) pipe.tail_begin = (int)g->n_pad;
    double jitter = 0.0;
    int tries = 0;
    FusedFallback fallback;
    for (;;) {
        // fork: the sweep stream starts after what is queued on the main stream (candidate preparation)
        HIP_TRY(hipEventRecord(c->ev_fork, c->stream));
        HIP_TRY(hipStreamWaitEvent(c->sweep_stream, c->ev_fork, 0));
        // K(X,X) first: it heads the factorisation's chain, K(X,X*) is not needed before the first pair is solved.  (Until
        // round 5 the host queued K(X,X*) and two hipMemsetAsync ahead of it -- a kernel trace showed K(X,X) starting 69 us
        // after K(X,X*), most of it the host's time in those calls; q and mu are now cleared by one small launch.)
        {
            PhaseScope ps(c, PH_KXX);
            launch_kxx(c->stream, g->X, g->h, g->noise_var + kGpyDiagJitter, jitter, g->A, g->lda, g->n_pad);
            launch_rhs(c->stream, g->y, g->X.pm, g->n, g->A, g->lda, g->n_pad, g->info, cholesky_info_ints(g->n_pad));
        }
        {
            PhaseScope ps(c, PH_KSTAR, c->sweep_stream);
            launch_kstar(c->sweep_stream, g->X, k->P, 0, k->m_pad, g->h, Vws, ldv, g->n_pad);
        }
        launch_zero_pair(c->sweep_stream, qbuf, mubuf, k->m_pad);
        {
            PhaseScope ps(c, PH_CHOL);
            launch_cholesky(c->stream, c->side_stream, c->chol_events, g->A, g->lda, g->n_pad, g->invDt, g->info, &pipe, true);
        }
        // join: everything the sweep streams were given is done before the main stream goes on (the last
        // pair has no rows below it, so the bulk stream's last launch precedes the sweep stream's in-panel solve
        // of that pair only through the events: wait for both)
        HIP_TRY(hipEventRecord(c->ev_join, c->sweep_stream));
        HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_join, 0));
        HIP_TRY(hipEventRecord(c->ev_join2, c->bulk_stream));
        HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_join2, 0));
        HIP_TRY(hipGetLastError());
        // the epilogue's KERNELS ride behind the closing launch on the assumption that the factorisation succeeded (they
        // read q, mu where the sweep left them, and write device scratch and the pinned winner record only); the caller's
        // acq / mean / var buffers are written after the status word is known to be good (below): on an error return --
        // CBO_ERR_NOT_PD after the ladder is exhausted -- they are left untouched, as the two-call sequence leaves them
        if (speculate) {
            rc = enqueue_finish(g, k, y_best, task, ei_jitter, cost, acq_out, mean_out, var_out, qbuf, mubuf, true, false);
            if (rc != CBO_OK) return rc;
        } else {
            HIP_TRY(hipMemcpyAsync(c->h_info, g->info, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        }
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (*c->h_info == 0) break;
        if (*c->h_info == kCholFusedTimeout) {                 // as in cbo_gp_fit: the attempt again, separate launches
            if (fallback.active())
                return fail(CBO_ERR_HIP, "a fused diagonal + panel launch gave up waiting, and so did the separate-launch repeat");
            fallback.engage(c);
            continue;
        }
        rc = next_jitter(g, &tries, &jitter);
        if (rc != CBO_OK) return rc;
    }
    g->fitted = true;
    g->fit_stamp = ++g_fit_stamp;
    g->parent_stamp = 0;
    g->alpha_ready = false;
    g->tries = tries;
    g->jitter = jitter;
    if (c->profiling) c->timers.n_fit += 1;
    if (tries_out) *tries_out = tries;
    if (jitter_out) *jitter_out = jitter;
    if (Vws != c->V) { k->v_stamp = g->fit_stamp; k->v_rows = g->n; }
    if (qbuf == k->q) k->fit_stamp = g->fit_stamp;         // the candidates' q, mu are this fit's
    if (speculate) {
        if (acq_out || mean_out || var_out) {
            rc = copy_posterior_out(c, k, acq_out, mean_out, var_out);
            if (rc != CBO_OK) return rc;
            HIP_TRY(hipStreamSynchronize(c->stream));
        }
        complete_finish(c, best_val, best_idx);
        if (entry) schedule_report(c->n_cu, c->n_cu_pipe, *entry, choice, fallback.active() ? -1 : tries, us_since(t_call) * 1e-3, 0.0, 0.0);
        return CBO_OK;
    }
    rc = finish_sweep(g, k, y_best, task, ei_jitter, cost, acq_out, mean_out, var_out, best_val, best_idx);
    if (rc == CBO_OK && entry)
        schedule_report(c->n_cu, c->n_cu_pipe, *entry, choice, fallback.active() ? -1 : tries, us_since(t_call) * 1e-3, 0.0, 0.0);
    return rc;
}

// ---- every exploration set of a trial in one call ------------------------------------------------------------------
// CBO.compute_best_acquisition_values (/root/reference/src/CBO.py:237-260) loops find_next_y_point over the S
// exploration sets.  Sets whose model has at most 128 observations -- every model the reference itself builds
// (10 + <= 40 points, src/ArgumentParser.py:18,25) -- are swept by ONE launch that factors and sweeps inside LDS
// (kernels_chol.hip, small_sets_kernel): no per-set launch chain, no per-set synchronisation, one copy back.  Such a
// model need not be fitted: the launch works from its resident data (cbo_gp_upload_data is enough) and leaves its
// fitted state alone.  A set whose factorisation meets a non-positive pivot there (jitchol's business), a larger
// model, or an fp32 model takes the general path: cbo_gp_fit_sweep when the model is not fitted, cbo_acq_sweep
// otherwise.
// How long the host polls the pinned result record of a one-launch job before it lets the runtime wait on the stream.
// Measured (round 3, CBO_HIP_TRACE_SLOW=1): a blocking hipStreamSynchronize behind such a launch returns after the
// kernel's ~40 us -- except about once in 600-1000 calls, when it returns after 62.75 ms +- 0.03 (a timed wait inside
// the runtime running out: the completion wake-up was missed, the kernel had long finished).  Round 2 polled for
// 32768 reads, which is ~10 us, not the 0.3 ms it was meant to be, so nearly every call ended in that blocking wait.
// The poll is now bounded by the clock: the record of a small job arrives within it and the host never sleeps on the
// stream; only jobs that really take longer fall through to hipStreamSynchronize.
constexpr double kPollBudgetUs = 2000.0;
// A launch whose result was polled is still "in flight" for the runtime: every so many of them the (idle) stream is
// synchronised so that their completion records are reaped; their signals have completed, the call does not sleep.
constexpr int kPolledLaunchesPerSync = 256;
template <class Pred>
static bool poll_until(Pred done, double budget_us)
{
    using clk = std::chrono::steady_clock;
    const clk::time_point t0 = clk::now();
    for (;;) {
        for (int spin = 0; spin < 256; ++spin)
            if (done()) return true;
        if (std::chrono::duration<double, std::micro>(clk::now() - t0).count() > budget_us) return done();
    }
}
static bool polled_launch_needs_sync(cbo_ctx *c)
{
    if (++c->polled_launches < kPolledLaunchesPerSync) return false;
    c->polled_launches = 0;
    return true;
}

// the model half of a one-workgroup kernel's descriptor
static void fill_small_model(cbo_small_set &st, const cbo_gp *g)
{
    const bool causal = g->X.sv != nullptr;
    st.xs = g->X.xs; st.sq = g->X.sq; st.sv = g->X.sv; st.pm = causal ? g->X.pm : nullptr; st.y = g->y;
    st.ld = g->X.ld;
    st.n = (int)g->n; st.d = g->d; st.zero_diag = g->h.zero_diag; st.ard = g->h.ard; st.pad_ = 0;
    st.variance = g->h.variance; st.lengthscale = g->h.lengthscale; st.noise_var = g->noise_var;
    st.diag_add = g->noise_var + kGpyDiagJitter;
    st.stage = nullptr; st.stage_ls = nullptr; st.raw = g->raw; st.pv = g->X.pv;
}

static int ensure_small_buffers(cbo_ctx *c, int n_sets, int blocks)
{
    if (n_sets > c->sets_cap) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        hipHostFree(c->sets_host); hipHostFree(c->small_out); hipFree(c->small_info);
        c->sets_host = nullptr; c->small_out = nullptr; c->small_info = nullptr;
        c->sets_cap = 0;
        const int cap = n_sets < 32 ? 32 : n_sets;
        HIP_TRY(hipHostMalloc(&c->sets_host, sizeof(cbo_small_set) * cap));
        HIP_TRY(hipHostMalloc(&c->small_out, sizeof(cbo_small_result) * cap));
        HIP_TRY(hipMalloc(&c->small_info, sizeof(int) * 2 * cap));
        HIP_TRY(hipMemset(c->small_info, 0, sizeof(int) * 2 * cap));
        std::memset(c->small_out, 0, sizeof(cbo_small_result) * cap);
        c->sets_cap = cap;
    }
    const size_t scratch = small_sets_scratch_doubles(n_sets, blocks), parts = (size_t)n_sets * (size_t)blocks;
    if (scratch > c->small_scratch_elems) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        hipFree(c->small_scratch);
        c->small_scratch = nullptr; c->small_scratch_elems = 0;
        HIP_TRY(hipMalloc(&c->small_scratch, sizeof(double) * scratch));
        c->small_scratch_elems = scratch;
    }
    if (parts > c->small_part_elems) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        hipFree(c->small_part_val); hipFree(c->small_part_idx);
        c->small_part_val = nullptr; c->small_part_idx = nullptr; c->small_part_elems = 0;
        HIP_TRY(hipMalloc(&c->small_part_val, sizeof(double) * parts));
        HIP_TRY(hipMalloc(&c->small_part_idx, sizeof(int64_t) * parts));
        c->small_part_elems = parts;
    }
    return CBO_OK;
}

// staged_set >= 0 (cbo_trial_step): that set's new data sit in the context's staging buffer, its model's host-side state
// is already the new one, and the one-launch path -- which the caller has checked the set takes -- prepares and stores them
static int sweep_sets_impl(int n_sets, cbo_gp *const *gps, cbo_cands *const *cands, const double *y_best, int task,
                           double ei_jitter, const double *costs, double *best_vals, int64_t *best_idxs, int staged_set);

extern "C" int cbo_acq_sweep_sets(int n_sets, cbo_gp *const *gps, cbo_cands *const *cands, const double *y_best, int task,
                                  double ei_jitter, const double *costs, double *best_vals, int64_t *best_idxs)
{
    return sweep_sets_impl(n_sets, gps, cands, y_best, task, ei_jitter, costs, best_vals, best_idxs, -1);
}

static int sweep_sets_impl(int n_sets, cbo_gp *const *gps, cbo_cands *const *cands, const double *y_best, int task,
                           double ei_jitter, const double *costs, double *best_vals, int64_t *best_idxs, int staged_set)
{
    if (n_sets <= 0 || !gps || !cands || !y_best || !costs || !best_vals || !best_idxs)
        return fail(CBO_ERR_INVALID, "bad argument");
    for (int i = 0; i < n_sets; ++i) {
        int rc = check_sweep_args(gps[i], cands[i], task);
        if (rc != CBO_OK) return rc;
        if (gps[i]->ctx != gps[0]->ctx) return fail(CBO_ERR_INVALID, "all sets must live on one context");
        if (gps[i]->n <= 0 || gps[i]->n_pad <= 0) return fail(CBO_ERR_INVALID, "a gp holds no data");
    }
    cbo_ctx *c = gps[0]->ctx;
    HIP_TRY(hipSetDevice(c->device));
    std::vector<int> small;
    int blocks = 1;
    for (int i = 0; i < n_sets; ++i) {
        if (gps[i]->dtype == CBO_DTYPE_F64 && gps[i]->n_pad == kPadN && c->small_sets) {
            small.push_back(i);
            const int b = (int)((cands[i]->m + 63) / 64);
            if (b > blocks) blocks = b;
        }
    }
    std::vector<char> done((size_t)n_sets, 0);
    if (!small.empty() && blocks <= 65535) {
        int rc = ensure_small_buffers(c, (int)small.size(), blocks);
        if (rc != CBO_OK) return rc;
        for (size_t j = 0; j < small.size(); ++j) {
            cbo_gp *g = gps[small[j]];
            cbo_cands *k = cands[small[j]];
            rc = prepare_cands(g, k);
            if (rc != CBO_OK) return rc;
            const bool causal = g->X.sv != nullptr;
            cbo_small_set &st = c->sets_host[j];
            fill_small_model(st, g);
            if (small[j] == staged_set) {
                st.stage = c->stage;
                st.stage_ls = g->h.ard ? g->ls_dev : nullptr;
            }
            st.cxs = k->P.xs; st.csq = k->P.sq; st.csv = causal ? k->P.sv : nullptr;
            st.cpm = causal ? k->pm : nullptr; st.cpv = causal ? k->pv : nullptr;
            st.cld = k->P.ld; st.m = k->m; st.index_offset = k->index_offset;
            st.task = task; st.y_best = y_best[small[j]]; st.ei_jitter = ei_jitter;
            st.cost = costs[small[j]];
        }
        const int ns = (int)small.size();
        if (++c->small_seq == 0) c->small_seq = 1;
        const int seq = c->small_seq;
        // CBO_HIP_TRACE_SLOW=1: a call that takes more than a millisecond says on stderr where the time went
        // (a value above 1 is the threshold in microseconds instead)
        static const bool trace_slow = std::getenv("CBO_HIP_TRACE_SLOW") != nullptr;
        static const double trace_over_us = trace_slow && std::atof(std::getenv("CBO_HIP_TRACE_SLOW")) > 1.0
                                                ? std::atof(std::getenv("CBO_HIP_TRACE_SLOW")) : 1000.0;
        using clk = std::chrono::steady_clock;
        const clk::time_point t_begin = trace_slow ? clk::now() : clk::time_point();
        clk::time_point t_launched, t_polled;
        launch_small_sets(c->stream, c->sets_host, ns, blocks, c->small_scratch, c->small_part_val, c->small_part_idx,
                          c->small_info, c->small_info + c->sets_cap, c->small_out, seq);
        HIP_TRY(hipGetLastError());
        if (trace_slow) t_launched = clk::now();
        // the result records arrive in pinned memory, each closed by the call's sequence number: poll them (kPollBudgetUs),
        // then let the runtime wait (jobs that really take that long)
        {
            const bool all = poll_until([&] {
                for (int j = 0; j < ns; ++j)
                    if (*reinterpret_cast<volatile int *>(&c->small_out[j].seq) != seq) return false;
                return true;
            }, kPollBudgetUs);
            if (trace_slow) t_polled = clk::now();
            const bool reap = polled_launch_needs_sync(c);
            if (!all || c->profiling || reap) HIP_TRY(hipStreamSynchronize(c->stream));
            if (trace_slow) {
                const clk::time_point t_end = clk::now();
                auto us = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
                if (us(t_begin, t_end) > trace_over_us)
                    std::fprintf(stderr, "[cbo] slow cbo_acq_sweep_sets call #%d: launch %.1f us, poll %.1f us (%s), "
                                 "synchronise %.1f us (%s)\n", seq, us(t_begin, t_launched), us(t_launched, t_polled),
                                 all ? "records arrived" : "gave up", us(t_polled, t_end),
                                 !all ? "after the poll gave up" : reap ? "periodic reap" : "none");
            }
            for (int j = 0; j < ns; ++j)
                if (c->small_out[j].seq != seq) return fail(CBO_ERR_HIP, "multi-set sweep: no result record");
            std::atomic_thread_fence(std::memory_order_acquire);
        }
        for (int j = 0; j < ns; ++j) {
            if (c->small_out[j].info != 0) continue;        // not positive definite as assembled: the jitchol ladder below
            best_vals[small[(size_t)j]] = c->small_out[j].best_val;
            best_idxs[small[(size_t)j]] = c->small_out[j].best_idx;
            done[(size_t)small[(size_t)j]] = 1;
        }
    }
    for (int i = 0; i < n_sets; ++i) {
        if (done[(size_t)i]) continue;
        int rc;
        if (!gps[i]->fitted)
            rc = cbo_gp_fit_sweep(gps[i], cands[i], y_best[i], task, ei_jitter, costs[i], nullptr, nullptr, nullptr,
                                  &best_vals[i], &best_idxs[i], nullptr, nullptr);
        else
            rc = cbo_acq_sweep(gps[i], cands[i], y_best[i], task, ei_jitter, costs[i], nullptr, nullptr, nullptr,
                               &best_vals[i], &best_idxs[i]);
        if (rc != CBO_OK) return rc;
    }
    return CBO_OK;
}

// One reference-scale trial in ONE call (src/CBO.py:143-173, CBO.intervene): the model of the set that was intervened on
// last receives its new data (src/CBO.py:224-235 rebuilds it; src/Monitor.py:160), every exploration set is swept
// (src/CBO.py:237-260) and the set to intervene on next is picked (src/CBO.py:269-277) -- cbo_gp_upload_data +
// cbo_acq_sweep_sets + cbo_argmax_sets without the three trips through the caller's language, which at the reference's
// model sizes (one 29 us launch for all sets) cost as much as the device work.
extern "C" int cbo_trial_step(int n_sets, cbo_gp *const *gps, cbo_cands *const *cands, int refit_set, int64_t n,
                              const double *X, const double *y, const double *pm, const double *pv, const double *y_best,
                              int task, double ei_jitter, const double *costs, double *best_vals, int64_t *best_idxs,
                              int *chosen_out)
{
    if (n_sets <= 0 || !gps || !chosen_out) return fail(CBO_ERR_INVALID, "bad argument");
    if (refit_set >= n_sets) return fail(CBO_ERR_INVALID, "refit_set out of range");
    int staged = -1;
    int64_t prev_n = 0;
    bool prev_fitted = false;
    if (refit_set >= 0) {
        cbo_gp *g = gps[refit_set];
        if (!g) return fail(CBO_ERR_INVALID, "gp is NULL");
        if (!cands || !y_best || !costs || !best_vals || !best_idxs) return fail(CBO_ERR_INVALID, "bad argument");
        cbo_ctx *c = g->ctx;
        // The upload folded into the sweep's one launch: when the set takes the one-launch path (as every model the
        // reference builds does) its new data go to the staging buffer and the launch's workgroups prepare the points
        // from there themselves -- no preparation launch, no stream synchronisation (15 of a trial's 57 us).  The same
        // conditions as sweep_sets_impl's, plus: the padded size and the prior-ness do not change, the data fit the buffer.
        bool fuse = c->small_sets && g->dtype == CBO_DTYPE_F64 && g->n_pad == kPadN && n > 0 && n <= kPadN && X && y &&
                    ((pm == nullptr) == (pv == nullptr)) && ((pv != nullptr) == (g->X.sv != nullptr)) &&
                    sizeof(double) * (size_t)(n * g->d + n + (pv ? 2 * n : 0)) <= kStageBytes;
        // (everything the sweep would refuse is refused BEFORE the model's host-side state moves to the new data)
        for (int i = 0; i < n_sets; ++i) {
            const int rc = check_sweep_args(gps[i], cands[i], task);
            if (rc != CBO_OK) return rc;
            if (gps[i]->ctx != c) return fail(CBO_ERR_INVALID, "all sets must live on one context");
            if (i != refit_set && (gps[i]->n <= 0 || gps[i]->n_pad <= 0)) return fail(CBO_ERR_INVALID, "a gp holds no data");
            if (gps[i]->dtype == CBO_DTYPE_F64 && gps[i]->n_pad == kPadN && (cands[i]->m + 63) / 64 > 65535) fuse = false;
            // the same model in two sets: the other copy's workgroups would read the resident arrays while the staged set's
            // first workgroup rewrites them in the same launch -- the plain upload first, then the sweep
            if (i != refit_set && gps[i] == g) fuse = false;
        }
        if (fuse) {
            HIP_TRY(hipSetDevice(c->device));
            if (c->stage_pending) { HIP_TRY(hipEventSynchronize(c->stage_done)); c->stage_pending = false; }
            double *sb = c->stage;
            std::memcpy(sb, X, sizeof(double) * n * g->d);
            std::memcpy(sb + n * g->d, y, sizeof(double) * n);
            g->h_pv.clear();
            if (pv) {
                std::memcpy(sb + n * g->d + n, pm, sizeof(double) * n);
                std::memcpy(sb + n * g->d + 2 * n, pv, sizeof(double) * n);
                g->h_pv.assign(pv, pv + n);
            }
            prev_n = g->n;
            prev_fitted = g->fitted;
            g->n = n;
            g->X.n = n;
            g->fitted = false;
            staged = refit_set;
        } else {
            const int rc = cbo_gp_upload_data(g, n, X, y, pm, pv);      // unfitted: the sweep below refits it
            if (rc != CBO_OK) return rc;
        }
    }
    int rc = sweep_sets_impl(n_sets, gps, cands, y_best, task, ei_jitter, costs, best_vals, best_idxs, staged);
    if (rc != CBO_OK && staged >= 0) {
        // The launch that was to carry the new data into the resident arrays failed (or was never queued): the host-side
        // state already describes the new data, the device arrays may hold either.  Finish the upload from the staging
        // buffer by the plain path -- it is the caller's data either way -- so that model and arrays agree again; if even
        // that fails the model is marked as holding nothing (n = 0: every later call refuses it until new data arrive).
        cbo_gp *g = gps[staged];
        const int up = cbo_gp_upload_data(g, n, X, y, pm, pv);
        if (up != CBO_OK) {
            g->n = 0;
            g->X.n = 0;
            g->fitted = false;
            (void)prev_n;
            (void)prev_fitted;
        }
        return rc;
    }
    if (rc != CBO_OK) return rc;
    return cbo_argmax_sets(best_vals, n_sets, chosen_out);
}

// What the context has measured and chosen for cbo_gp_fit_sweep, one line per shape, as text (scripts/schedule_scan.py,
// profiles/r04_schedule_crossover.txt).  Returns the number of shapes still exploring (0 = every schedule is settled), or a
// CBO_ERR_* code (they are negative); `buf` may be NULL (only the count is wanted).
extern "C" int cbo_schedule_report(cbo_ctx *c, char *buf, int64_t cap)
{
    if (!c) return fail(CBO_ERR_INVALID, "ctx is NULL");      // (CBO_ERR_* codes are negative as they are)
    static const char *names[] = {"cold", "sequence", "base", "neighbours", "climb", "grouping", "settled", "renewing the first split"};
    std::string out;
    int exploring = 0;
    char line[512];
    for (const auto &kv : c->schedule) {
        const ScheduleEntry &e = kv.second;
        if (e.state != ScheduleEntry::SETTLED) ++exploring;
        const int nb = (int)(kv.first.first / 128);
        const double rounds = std::ceil((double)e.strips / c->n_cu);
        std::snprintf(line, sizeof(line), "rows %lld candidates %lld: %s after %d calls; pairs %d of %d (%s), updates %s; "
                      "last call ran pairs %d group %d; "
                      "measured alone: factorisation %.0f us = %.1f us/panel, sweep %.0f us = %.2f us/stage and round;",
                      (long long)kv.first.first, (long long)kv.first.second, names[(int)e.state], e.calls, e.cur, e.all_pairs,
                      e.cur < 0 ? "the plain sequence" : e.cur == 0 ? "overlapped, nothing pipelined" :
                      e.cur == e.all_pairs ? "everything pipelined" : "then one left-looking launch",
                      e.group >= 2 ? "in groups of two pairs" : "pair by pair", e.ran_pairs, e.ran_group, e.fact_alone_us,
                      nb ? e.fact_alone_us / nb : 0.0,
                      e.sweep_alone_us, nb ? e.sweep_alone_us / (rounds * 2.0 * nb * (nb + 1)) : 0.0);
        out += line;
        for (const auto &sv : e.samples) {
            std::snprintf(line, sizeof(line), " g%d/p%d:%.3fms x%d", sv.first.first, sv.first.second, sv.second.ms(), sv.second.count);
            out += line;
        }
        out += "\n";
    }
    if (buf && cap > 0) {
        const size_t n = out.size() < (size_t)cap - 1 ? out.size() : (size_t)cap - 1;
        std::memcpy(buf, out.data(), n);
        buf[n] = 0;
    }
    return exploring;
}

extern "C" int cbo_acq_sweep_host(cbo_gp *g, int64_t m, const double *Xs, const double *pm, const double *pv,
                                  double y_best, int task, double ei_jitter, double cost, double *acq_out,
                                  double *best_val, int64_t *best_idx)
{
    if (!g) return fail(CBO_ERR_INVALID, "gp is NULL");
    cbo_cands *k = nullptr;
    int rc = scratch_cands(g->ctx, m, g->d, Xs, pm, pv, &k);
    if (rc != CBO_OK) return rc;
    return cbo_acq_sweep(g, k, y_best, task, ei_jitter, cost, acq_out, nullptr, nullptr, best_val, best_idx);
}

// posterior mean / variance of m host points into the context's mean / var vectors (device), via the scratch set
static int posterior_of_host_points(cbo_gp *g, int64_t m, const double *Xs, const double *pm, const double *pv,
                                    int include_noise, cbo_cands **k_out)
{
    const bool causal = g->X.sv != nullptr;
    if (causal && (!pm || !pv)) return fail(CBO_ERR_INVALID, "causal gp needs candidate prior mean/variance");
    cbo_ctx *c = g->ctx;
    cbo_cands *k = nullptr;
    int rc = scratch_cands(c, m, g->d, Xs, causal ? pm : nullptr, causal ? pv : nullptr, &k);
    if (rc != CBO_OK) return rc;
    rc = enqueue_posterior(g, k);
    if (rc != CBO_OK) return rc;
    AcqParams p;
    p.variance = g->h.variance; p.noise_var = g->noise_var; p.y_best = 0.0; p.ei_jitter = 0.0; p.cost = 1.0;
    p.task = CBO_TASK_MIN; p.include_noise = include_noise ? 1 : 0; p.want_ei = 0;
    {
        PhaseScope ps(c, PH_ACQ);
        launch_acq(c->stream, c->q, c->mu, causal ? k->pm : nullptr, causal ? k->pv : nullptr, m, p, c->mean, c->var,
                   nullptr, c->part_val, c->part_idx, 0, acq_blocks_for(m));
    }
    HIP_TRY(hipGetLastError());
    *k_out = k;
    return CBO_OK;
}

extern "C" int cbo_gp_predict(cbo_gp *g, int64_t m, const double *Xs, const double *pm, const double *pv,
                              int include_noise, double *mean_out, double *var_out)
{
    if (!g || !mean_out || !var_out) return fail(CBO_ERR_INVALID, "NULL argument");
    if (!g->fitted) return fail(CBO_ERR_NOT_FITTED, "gp is not fitted");
    cbo_ctx *c = g->ctx;
    cbo_cands *k = nullptr;
    int rc = posterior_of_host_points(g, m, Xs, pm, pv, include_noise, &k);
    if (rc != CBO_OK) return rc;
    HIP_TRY(hipMemcpyAsync(mean_out, c->mean, sizeof(double) * m, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(var_out, c->var, sizeof(double) * m, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->profiling) c->timers.n_sweep += 1;
    return CBO_OK;
}

extern "C" int cbo_gp_set_hyper(cbo_gp *g, double variance, const double *lengthscale, double noise_var)
{
    if (!g || !lengthscale) return fail(CBO_ERR_INVALID, "NULL argument");
    cbo_ctx *c = g->ctx;
    HIP_TRY(hipSetDevice(c->device));
    g->h.variance = variance;
    g->noise_var = noise_var;
    g->fitted = false;
    g->alpha_ready = false;
    if (g->h.ard) {
        g->ls.assign(lengthscale, lengthscale + g->d);
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipMemcpy(g->ls_dev, lengthscale, sizeof(double) * g->d, hipMemcpyHostToDevice));
        // GPy ARD scales the inputs: re-derive the scaled SoA coordinates and their squared norms
        launch_prep_points(c->stream, g->raw, g->n, g->d, g->ls_dev, g->X.sv ? g->X.pv : nullptr, g->X.xs, g->n_pad,
                           g->X.sq, g->X.sv);
        HIP_TRY(hipGetLastError());
    } else {
        g->ls.assign(lengthscale, lengthscale + 1);
        g->h.lengthscale = lengthscale[0];
    }
    return CBO_OK;
}

extern "C" int cbo_gp_log_marginal(cbo_gp *g, double *lml_out)
{
    if (!g || !lml_out) return fail(CBO_ERR_INVALID, "NULL argument");
    if (!g->fitted) return fail(CBO_ERR_NOT_FITTED, "gp is not fitted");
    cbo_ctx *c = g->ctx;
    HIP_TRY(hipSetDevice(c->device));
    double h2[2];
    // part_val is a free 2048-double device scratch between sweeps
    launch_lml_terms(c->stream, g->A, g->lda, g->n_pad, g->z, c->part_val);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h2, c->part_val, sizeof(h2), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    // GPy: 0.5 * (-n log(2 pi) - W_logdet - sum(alpha * (Y - m))),  W_logdet = 2 sum log L_ii
    *lml_out = 0.5 * (-(double)g->n * 1.8378770664093453 - 2.0 * h2[1] - h2[0]);
    return CBO_OK;
}

// Gradients of the log marginal likelihood with respect to the kernel variance, the lengthscale(s) and the noise
// variance: 1/2 sum_ij (alpha alpha^T - Ky^-1)_ij dKy_ij/dtheta (GPy ExactGaussianInference: dL_dK ->
// kern.update_gradients_full, dL_dthetaL).  Ky^-1 = L^-T L^-1 on the device: L^-1 by the sweep machinery on
// identity right-hand sides (its q output is diag(Ky^-1)), the product by the GEMM form of the update kernel over
// the non-zero lower-triangular part only, the contraction with dK/dtheta by one pass over the upper tiles.
// host arithmetic shared by both forms of the likelihood gradients: hs = variance sum, lengthscale sums per dimension
static void lml_outputs(const cbo_gp *g, const double *grad_sums, double zz, double logdet, double aa, double tr_w,
                        double *lml_out, double *dvariance_out, double *dlengthscale_out, double *dnoise_out)
{
    *dvariance_out = 0.5 * grad_sums[0] / g->h.variance;
    if (g->h.ard) {
        for (int k = 0; k < g->d; ++k) dlengthscale_out[k] = 0.5 * grad_sums[1 + k] / g->ls[(size_t)k];
    } else {
        double sum = 0.0;
        for (int k = 0; k < g->d; ++k) sum += grad_sums[1 + k];
        dlengthscale_out[0] = 0.5 * sum / g->h.lengthscale;
    }
    *dnoise_out = 0.5 * (aa - tr_w);
    if (lml_out) *lml_out = 0.5 * (-(double)g->n * 1.8378770664093453 - 2.0 * logdet - zz);
}

// Models of at most 128 observations (every model the reference builds): likelihood and gradients in ONE launch, from
// the data and the current hyper-parameters -- no fit beforehand, none left behind.  Returns 1 when done, 0 when the
// general path has to take over (Ky not positive definite as assembled: the jitchol ladder lives there), < 0 on error.
static int small_lml_gradients(cbo_gp *g, double *lml_out, double *dvariance_out, double *dlengthscale_out,
                               double *dnoise_out)
{
    cbo_ctx *c = g->ctx;
    if (!(g->dtype == CBO_DTYPE_F64 && g->n_pad == kPadN && c->small_sets && g->n > 0)) return 0;
    int rc = ensure_small_buffers(c, 1, 2);            // scratch of two workgroup slots >= factor + L^-1
    if (rc != CBO_OK) return rc;
    if (!c->lml_out) {
        if (hipHostMalloc(&c->lml_out, sizeof(cbo_small_lml_result)) != hipSuccess) return fail(CBO_ERR_HIP, "hipHostMalloc");
        std::memset(c->lml_out, 0, sizeof(cbo_small_lml_result));
    }
    cbo_small_set st{};
    fill_small_model(st, g);
    if (++c->small_seq == 0) c->small_seq = 1;
    const int seq = c->small_seq;
    launch_small_lml(c->stream, st, c->small_scratch, c->small_info, c->lml_out, seq);
    if (hipGetLastError() != hipSuccess) return fail(CBO_ERR_HIP, "small_lml_kernel launch");
    const bool ready = poll_until([&] { return *reinterpret_cast<volatile int *>(&c->lml_out->seq) == seq; }, kPollBudgetUs);
    if (!ready || c->profiling || polled_launch_needs_sync(c)) {
        if (hipStreamSynchronize(c->stream) != hipSuccess) return fail(CBO_ERR_HIP, "hipStreamSynchronize");
        if (c->lml_out->seq != seq) return fail(CBO_ERR_HIP, "likelihood kernel: no result record");
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (c->lml_out->info != 0) return 0;
    const double *t = c->lml_out->terms;
    lml_outputs(g, t, t[1 + CBO_MAX_DIM], t[1 + CBO_MAX_DIM + 1], t[1 + CBO_MAX_DIM + 2], t[1 + CBO_MAX_DIM + 3], lml_out,
                dvariance_out, dlengthscale_out, dnoise_out);
    return 1;
}

extern "C" int cbo_gp_lml_gradients(cbo_gp *g, double *lml_out, double *dvariance_out, double *dlengthscale_out,
                                    double *dnoise_out)
{
    if (!g || !dvariance_out || !dlengthscale_out || !dnoise_out) return fail(CBO_ERR_INVALID, "NULL argument");
    cbo_ctx *c = g->ctx;
    HIP_TRY(hipSetDevice(c->device));
    {
        const int done = small_lml_gradients(g, lml_out, dvariance_out, dlengthscale_out, dnoise_out);
        if (done < 0) return done;
        if (done == 1) return CBO_OK;
    }
    if (!g->fitted) {                      // (a small model that was not positive definite as assembled lands here)
        if (g->n_pad != kPadN) return fail(CBO_ERR_NOT_FITTED, "gp is not fitted");
        const int frc = cbo_gp_fit(g, nullptr, nullptr);
        if (frc != CBO_OK) return frc;
    }
    int rc = ensure_alpha(g);
    if (rc != CBO_OK) return rc;
    const int64_t n_pad = g->n_pad;
    int64_t chunk = 0, ldv = 0;
    rc = ensure_workspaces(c, n_pad, n_pad, &chunk, &ldv);
    if (rc != CBO_OK) return rc;
    if (chunk < n_pad) return fail(CBO_ERR_UNSUPPORTED, "likelihood gradients need an n_pad x n_pad workspace (raise CBO_HIP_WORKSPACE_MB)");
    const int64_t ldw = n_pad + kLdExtra;
    const size_t w_bytes = sizeof(double) * (size_t)n_pad * (size_t)ldw;
    const size_t part_elems = (size_t)lml_grad_tiles(n_pad) * (size_t)(1 + g->d);
    if (w_bytes > c->W_bytes || part_elems > c->gpart_elems) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        hipFree(c->W); hipFree(c->gpart);
        c->W = nullptr; c->gpart = nullptr; c->W_bytes = 0; c->gpart_elems = 0;
        HIP_TRY(hipMalloc(&c->W, w_bytes));
        c->W_bytes = w_bytes;
        HIP_TRY(hipMalloc(&c->gpart, sizeof(double) * part_elems));
        c->gpart_elems = part_elems;
    }
    // V = L^-1 (identity right-hand sides), q_j = (Ky^-1)_jj
    launch_set_identity(c->stream, c->V, ldv, n_pad);
    if (prefer_right_looking(c, n_pad, n_pad)) {
        rc = enqueue_right_looking(g, c->V, ldv, n_pad, c->q, c->mu, true);
        if (rc != CBO_OK) return rc;
    } else {
        launch_trsm_strips(c->stream, g->A, g->lda, g->invDt, c->V, ldv, n_pad, n_pad, g->z, c->q, c->mu);
    }
    // -Ky^-1 = -(L^-1)^T L^-1: rows [k0, k0+256) of L^-1 only reach columns < k0+256, so pair p touches the
    // leading (k0+256)^2 block; upper part only
    HIP_TRY(hipMemsetAsync(c->W, 0, w_bytes, c->stream));
    for (int k0 = 0; k0 < (int)n_pad; k0 += 256) {
        const int klen = (k0 + 256 <= (int)n_pad) ? 256 : 128;
        launch_gemm_update(c->stream, c->V, ldv, c->V, ldv, c->W, ldw, k0, klen, 0, k0 + klen, k0 + klen,
                           c->pipe_chunk_blocks, true, true);
    }
    launch_lml_grad(c->stream, g->X, g->h, g->alpha, c->W, ldw, n_pad, c->gpart, c->part_val);
    launch_lml_terms(c->stream, g->A, g->lda, n_pad, g->z, c->part_val + 16);
    HIP_TRY(hipGetLastError());
    // tr(Ky^-1) = sum q and alpha^T alpha by device reductions: part_val[20..23]
    launch_dot2(c->stream, g->alpha, g->alpha, g->n, c->part_val + 20);
    launch_sum(c->stream, c->q, g->n, c->part_val + 22);
    HIP_TRY(hipGetLastError());
    double hs[24];
    HIP_TRY(hipMemcpyAsync(hs, c->part_val, sizeof(hs), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    lml_outputs(g, hs, hs[16], hs[17], hs[20], hs[22], lml_out, dvariance_out, dlengthscale_out, dnoise_out);
    return CBO_OK;
}

// the reversed factor of the backward substitution (kernels_kmat.hip), once per fit
static int ensure_reversed_factor(cbo_gp *g)
{
    cbo_ctx *c = g->ctx;
    if (!g->T) {
        HIP_TRY(hipMalloc(&g->T, sizeof(double) * (size_t)g->n_pad * (size_t)g->lda));
        HIP_TRY(hipMalloc(&g->invT, sizeof(double) * (size_t)(g->n_pad / 16) * 256));
        g->t_stamp = 0;
    }
    if (g->t_stamp != g->fit_stamp) {
        launch_reversed_factor(c->stream, g->A, g->lda, g->n_pad, g->invDt, g->T, g->lda, g->invT);
        HIP_TRY(hipGetLastError());
        g->t_stamp = g->fit_stamp;
    }
    return CBO_OK;
}

// Gradients of the posterior at a batch of points, any size: per workspace chunk, V = L^-1 K* by the forward sweep,
// W = L^-T V by the SAME strip kernel on the reversed system (the factor read backwards is lower triangular again),
// then one pass that forms both gradients from alpha and W.  Nothing is allocated once the workspaces have grown.
extern "C" int cbo_gp_predict_gradients(cbo_gp *g, int64_t m, const double *Xs, const double *pv_s, double *dmean_out,
                                        double *dvar_out)
{
    if (!g || !Xs || !dmean_out || !dvar_out) return fail(CBO_ERR_INVALID, "NULL argument");
    if (m <= 0) return fail(CBO_ERR_INVALID, "m must be positive");
    if (!g->fitted) return fail(CBO_ERR_NOT_FITTED, "gp is not fitted");
    cbo_ctx *c = g->ctx;
    HIP_TRY(hipSetDevice(c->device));
    const bool causal = g->X.sv != nullptr;
    if (causal && !pv_s) return fail(CBO_ERR_INVALID, "causal gp needs the prior variance at the prediction points");
    // the prior mean does not enter the gradients (GPy ignores the mean function there): the variances stand in
    cbo_cands *k = nullptr;
    int rc = scratch_cands(c, m, g->d, Xs, causal ? pv_s : nullptr, causal ? pv_s : nullptr, &k);
    if (rc != CBO_OK) return rc;
    rc = prepare_cands(g, k);
    if (rc == CBO_OK) rc = ensure_alpha(g);
    if (rc == CBO_OK) rc = ensure_reversed_factor(g);
    if (rc != CBO_OK) return rc;
    // two workspaces of the same shape: V (forward) and W (backward); halve the budget so that both fit it
    int64_t chunk = 0, ldv = 0;
    const size_t saved_cap = c->max_ws_bytes;
    c->max_ws_bytes = saved_cap / 2;
    rc = ensure_workspaces(c, g->n_pad, k->m_pad, &chunk, &ldv);
    c->max_ws_bytes = saved_cap;
    if (rc != CBO_OK) return rc;
    const size_t w_bytes = sizeof(double) * (size_t)g->n_pad * (size_t)ldv;
    const size_t grad_elems = 2 * (size_t)k->m_pad * (size_t)g->d;
    if (w_bytes > c->W_bytes || grad_elems > c->grads_elems || (g->h.ard && !g->inv_ls_dev)) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (w_bytes > c->W_bytes) {
            hipFree(c->W);
            c->W = nullptr; c->W_bytes = 0;
            HIP_TRY(hipMalloc(&c->W, w_bytes));
            c->W_bytes = w_bytes;
        }
        if (grad_elems > c->grads_elems) {
            hipFree(c->grads);
            c->grads = nullptr; c->grads_elems = 0;
            HIP_TRY(hipMalloc(&c->grads, sizeof(double) * grad_elems));
            c->grads_elems = grad_elems;
        }
        if (g->h.ard && !g->inv_ls_dev) HIP_TRY(hipMalloc(&g->inv_ls_dev, sizeof(double) * CBO_MAX_DIM));
    }
    if (g->h.ard) {
        double il[CBO_MAX_DIM] = {0};
        for (int i = 0; i < g->d; ++i) il[i] = 1.0 / g->ls[(size_t)i];
        HIP_TRY(hipMemcpyAsync(g->inv_ls_dev, il, sizeof(double) * g->d, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));               // `il` lives in this frame
    }
    double *dmean = c->grads, *dvar = c->grads + (size_t)k->m_pad * g->d;
    for (int64_t c0 = 0; c0 < k->m_pad; c0 += chunk) {
        const int64_t cols = (k->m_pad - c0 < chunk) ? (k->m_pad - c0) : chunk;
        launch_kstar(c->stream, g->X, k->P, c0, cols, g->h, c->V, ldv, g->n_pad);
        launch_trsm_strips(c->stream, g->A, g->lda, g->invDt, c->V, ldv, g->n_pad, cols, nullptr, nullptr, nullptr);
        launch_reverse_rows(c->stream, c->V, ldv, g->n_pad, cols, c->W, ldv);
        launch_trsm_strips(c->stream, g->T, g->lda, g->invT, c->W, ldv, g->n_pad, cols, nullptr, nullptr, nullptr);
        launch_pred_gradients(c->stream, g->X, g->n_pad, k->P, c0, cols, m, g->h, g->inv_ls_dev, g->alpha, c->W, ldv,
                              dmean, dvar);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(dmean_out, dmean, sizeof(double) * m * g->d, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(dvar_out, dvar, sizeof(double) * m * g->d, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CBO_OK;
}

extern "C" int cbo_gp_predict_grouped(cbo_gp *g, int64_t m_groups, int64_t group, const double *Xs, const double *pm,
                                      const double *pv, int include_noise, double *mean_out, double *var_out)
{
    if (!g || !mean_out || !var_out || !Xs) return fail(CBO_ERR_INVALID, "NULL argument");
    if (m_groups <= 0 || group <= 0) return fail(CBO_ERR_INVALID, "m_groups and group must be positive");
    if (!g->fitted) return fail(CBO_ERR_NOT_FITTED, "gp is not fitted");
    cbo_ctx *c = g->ctx;
    const int64_t m = m_groups * group;
    cbo_cands *k = nullptr;
    int rc = posterior_of_host_points(g, m, Xs, pm, pv, include_noise, &k);
    if (rc != CBO_OK) return rc;
    // group means into the (now free) q / mu vectors
    launch_group_mean(c->stream, c->mean, m_groups, group, c->q);
    launch_group_mean(c->stream, c->var, m_groups, group, c->mu);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(mean_out, c->q, sizeof(double) * m_groups, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(var_out, c->mu, sizeof(double) * m_groups, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CBO_OK;
}

// Do-calculus prior of a batch of candidate interventions, inputs built on the device (SURVEY.md §8 f1;
// src/DoCalculus.py:34-89): for candidate c the graph-level GP is evaluated at the n_obs observed input rows with the
// intervened columns overwritten by values[c], and mean / variance are averaged over those rows.  Only `observed`
// (n_obs x d) and `values` (m x n_iv) are uploaded; the m * n_obs prediction points exist on the device only.
extern "C" int cbo_gp_predict_do(cbo_gp *g, int64_t m, int64_t n_obs, const double *observed, int n_iv,
                                 const double *values, const int *iv_index, int include_noise, double *mean_out,
                                 double *var_out)
{
    if (!g || !observed || !values || !iv_index || !mean_out || !var_out) return fail(CBO_ERR_INVALID, "NULL argument");
    if (m <= 0 || n_obs <= 0 || n_iv <= 0) return fail(CBO_ERR_INVALID, "m, n_obs and n_iv must be positive");
    if (!g->fitted) return fail(CBO_ERR_NOT_FITTED, "gp is not fitted");
    if (g->X.sv != nullptr) return fail(CBO_ERR_INVALID, "the do-calculus inputs go to a graph-level (non-causal) gp");
    for (int j = 0; j < g->d; ++j)
        if (iv_index[j] >= n_iv) return fail(CBO_ERR_INVALID, "iv_index refers to a column values does not have");
    cbo_ctx *c = g->ctx;
    HIP_TRY(hipSetDevice(c->device));
    const int64_t total = m * n_obs;
    if (!c->scratch_k) { c->scratch_k = new cbo_cands(); c->scratch_k->ctx = c; }
    cbo_cands *k = c->scratch_k;
    int rc = cands_reserve(k, total, g->d, false);
    if (rc != CBO_OK) return rc;
    // staging for observed | values | iv_index: the export scratch (device), filled by three small copies
    const size_t need = (size_t)(n_obs * g->d) + (size_t)(m * n_iv) + CBO_MAX_DIM;
    rc = ensure_export(c, need);
    if (rc != CBO_OK) return rc;
    double *d_obs = c->export_buf, *d_val = d_obs + n_obs * g->d;
    int *d_idx = reinterpret_cast<int *>(d_val + m * n_iv);
    HIP_TRY(hipMemcpyAsync(d_obs, observed, sizeof(double) * n_obs * g->d, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_val, values, sizeof(double) * m * n_iv, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_idx, iv_index, sizeof(int) * g->d, hipMemcpyHostToDevice, c->stream));
    k->m = total; k->d = g->d; k->index_offset = 0;
    k->m_pad = round_up(total, kStrip);
    k->P.n = total; k->P.ld = k->m_pad; k->P.d = g->d;
    k->has_prior = false;
    k->prepared_for = nullptr; k->prepared_ls.clear();
    k->fit_stamp = 0; k->v_stamp = 0;
    launch_expand_interventions(c->stream, d_obs, n_obs, g->d, d_val, n_iv, d_idx, m, k->raw);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));               // the host buffers are the caller's
    rc = enqueue_posterior(g, k);
    if (rc != CBO_OK) return rc;
    AcqParams p;
    p.variance = g->h.variance; p.noise_var = g->noise_var; p.y_best = 0.0; p.ei_jitter = 0.0; p.cost = 1.0;
    p.task = CBO_TASK_MIN; p.include_noise = include_noise ? 1 : 0; p.want_ei = 0;
    launch_acq(c->stream, c->q, c->mu, nullptr, nullptr, total, p, c->mean, c->var, nullptr, c->part_val, c->part_idx, 0,
               acq_blocks_for(total));
    launch_group_mean(c->stream, c->mean, m, n_obs, c->q);
    launch_group_mean(c->stream, c->var, m, n_obs, c->mu);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(mean_out, c->q, sizeof(double) * m, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(var_out, c->mu, sizeof(double) * m, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CBO_OK;
}

// ---- tiny host-side reductions -------------------------------------------------------------------
static bool host_better(double va, int64_t ia, double vb, int64_t ib)
{
    const bool na = std::isnan(va), nb = std::isnan(vb);
    if (na != nb) return na;
    if (na || va == vb) return ia < ib;
    return va > vb;
}

extern "C" int cbo_argmax_sets(const double *ys, int s, int *idx_out)
{
    if (!ys || !idx_out || s <= 0) return fail(CBO_ERR_INVALID, "bad argument");
    // np.where(ys == np.max(ys))[0][0]  (src/CBO.py:275): np.max propagates NaN, and NaN == NaN is
    // false, so the reference raises IndexError there; we report the first NaN instead.
    int best = 0;
    for (int i = 1; i < s; ++i)
        if (host_better(ys[i], i, ys[best], best)) best = i;
    *idx_out = best;
    return CBO_OK;
}

extern "C" int cbo_argmax_pairs(const double *vals, const int64_t *idxs, int n, double *best_val, int64_t *best_idx)
{
    if (!vals || !idxs || n <= 0 || !best_val || !best_idx) return fail(CBO_ERR_INVALID, "bad argument");
    int b = 0;
    for (int i = 1; i < n; ++i)
        if (host_better(vals[i], idxs[i], vals[b], idxs[b])) b = i;
    *best_val = vals[b];
    *best_idx = idxs[b];
    return CBO_OK;
}

// ---- Monte-Carlo interventional target (f4) ----------------------------------------------------------
struct cbo_sem {
    cbo_ctx *ctx = nullptr;
    cbo_sem_spec spec{};
    int64_t n_draws = 0;
    int n_eps = 0;
    double *eps_cm = nullptr;        // n_eps x n_draws (column of the caller's matrix = contiguous run here)
    // per-call workspaces, grown on demand
    double *values = nullptr, *partial = nullptr, *mean = nullptr;
    int *iv_nodes = nullptr;
    int64_t cap_m = 0;
    int cap_iv = 0;
};

static int check_sem_spec(const cbo_sem_spec *sp, int n_eps)
{
    if (sp->n_nodes < 1 || sp->n_nodes > CBO_SEM_MAX_NODES) return fail(CBO_ERR_INVALID, "sem: n_nodes out of range");
    if (sp->term_begin[0] != 0) return fail(CBO_ERR_INVALID, "sem: term_begin[0] must be 0");
    for (int k = 0; k < sp->n_nodes; ++k) {
        if (sp->term_begin[k + 1] < sp->term_begin[k] || sp->term_begin[k + 1] > CBO_SEM_MAX_TERMS)
            return fail(CBO_ERR_INVALID, "sem: term_begin must be non-decreasing and <= CBO_SEM_MAX_TERMS");
        if (sp->eps_index[k] < -1 || sp->eps_index[k] >= n_eps)
            return fail(CBO_ERR_INVALID, "sem: eps_index out of range");
        for (int t = sp->term_begin[k]; t < sp->term_begin[k + 1]; ++t) {
            if (sp->term_parent[t] < 0 || sp->term_parent[t] >= k)
                return fail(CBO_ERR_INVALID, "sem: a term must read an earlier node (evaluation order)");
            if (sp->term_fn[t] < CBO_FN_ID || sp->term_fn[t] > CBO_FN_SIN)
                return fail(CBO_ERR_INVALID, "sem: unknown term function");
        }
    }
    return CBO_OK;
}

extern "C" int cbo_sem_create(cbo_ctx *c, const cbo_sem_spec *spec, int64_t n_samples, int n_eps, const double *eps,
                              cbo_sem **out)
{
    if (!c || !spec || !eps || !out) return fail(CBO_ERR_INVALID, "NULL argument");
    if (n_samples <= 0 || n_eps < 1 || n_eps > CBO_SEM_MAX_NODES)
        return fail(CBO_ERR_INVALID, "sem: n_samples must be positive and n_eps in [1, CBO_SEM_MAX_NODES]");
    int rc = check_sem_spec(spec, n_eps);
    if (rc != CBO_OK) return rc;
    HIP_TRY(hipSetDevice(c->device));
    cbo_sem *m = new cbo_sem();
    m->ctx = c; m->spec = *spec; m->n_draws = n_samples; m->n_eps = n_eps;
    std::vector<double> cm((size_t)n_samples * n_eps);
    for (int64_t s = 0; s < n_samples; ++s)
        for (int k = 0; k < n_eps; ++k) cm[(size_t)k * n_samples + s] = eps[s * n_eps + k];
    hipError_t e = hipMalloc(&m->eps_cm, sizeof(double) * cm.size());
    if (e == hipSuccess) e = hipMemcpy(m->eps_cm, cm.data(), sizeof(double) * cm.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        cbo_sem_destroy(m);
        return fail(CBO_ERR_HIP, std::string("cbo_sem_create: ") + hipGetErrorString(e));
    }
    *out = m;
    return CBO_OK;
}

extern "C" void cbo_sem_destroy(cbo_sem *m)
{
    if (!m) return;
    hipSetDevice(m->ctx->device);
    hipStreamSynchronize(m->ctx->stream);
    hipFree(m->eps_cm); hipFree(m->values); hipFree(m->partial); hipFree(m->mean); hipFree(m->iv_nodes);
    delete m;
}

extern "C" int cbo_sem_target(cbo_sem *m, int target, int64_t n_iv_sets, int n_iv, const int *iv_nodes,
                              const double *values, double *mean_out)
{
    if (!m || !mean_out) return fail(CBO_ERR_INVALID, "NULL argument");
    if (n_iv_sets <= 0) return fail(CBO_ERR_INVALID, "sem: m must be positive");
    if (target < 0 || target >= m->spec.n_nodes) return fail(CBO_ERR_INVALID, "sem: target node out of range");
    if (n_iv < 0 || n_iv > CBO_SEM_MAX_NODES) return fail(CBO_ERR_INVALID, "sem: n_iv out of range");
    if (n_iv > 0 && (!iv_nodes || !values)) return fail(CBO_ERR_INVALID, "sem: intervention nodes/values missing");
    for (int j = 0; j < n_iv; ++j)
        if (iv_nodes[j] < 0 || iv_nodes[j] >= m->spec.n_nodes)
            return fail(CBO_ERR_INVALID, "sem: intervened node out of range");
    cbo_ctx *c = m->ctx;
    HIP_TRY(hipSetDevice(c->device));
    const int nb = sem_partial_blocks(m->n_draws);
    const int iv_cols = n_iv > 0 ? n_iv : 1;
    if (n_iv_sets > m->cap_m || iv_cols > m->cap_iv) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        hipFree(m->values); hipFree(m->partial); hipFree(m->mean); hipFree(m->iv_nodes);
        m->values = m->partial = m->mean = nullptr; m->iv_nodes = nullptr; m->cap_m = 0; m->cap_iv = 0;
        const int64_t cap = n_iv_sets > m->cap_m ? n_iv_sets : m->cap_m;
        HIP_TRY(hipMalloc(&m->values, sizeof(double) * cap * CBO_SEM_MAX_NODES));
        HIP_TRY(hipMalloc(&m->partial, sizeof(double) * cap * nb));
        HIP_TRY(hipMalloc(&m->mean, sizeof(double) * cap));
        HIP_TRY(hipMalloc(&m->iv_nodes, sizeof(int) * CBO_SEM_MAX_NODES));
        m->cap_m = cap; m->cap_iv = CBO_SEM_MAX_NODES;
    }
    if (n_iv > 0) {
        HIP_TRY(hipMemcpyAsync(m->values, values, sizeof(double) * n_iv_sets * n_iv, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(m->iv_nodes, iv_nodes, sizeof(int) * n_iv, hipMemcpyHostToDevice, c->stream));
    }
    launch_sem_target(c->stream, m->spec, m->eps_cm, m->n_draws, target, n_iv_sets, n_iv, iv_nodes, m->iv_nodes, m->values,
                      m->partial, m->mean);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(mean_out, m->mean, sizeof(double) * n_iv_sets, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CBO_OK;
}

extern "C" int cbo_selftest_mfma(cbo_ctx *c, double *max_abs_err_out)
{
    if (!c) return fail(CBO_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    double err = -1.0;
    if (run_mfma_selftest(c->stream, &err) != 0) return fail(CBO_ERR_HIP, "mfma selftest launch failed");
    if (max_abs_err_out) *max_abs_err_out = err;
    if (err != 0.0) return fail(CBO_ERR_HIP, "fp64 MFMA lane map differs from what the kernels assume");
    double err32 = -1.0;
    if (run_mfma_f32_selftest(c->stream, &err32) != 0) return fail(CBO_ERR_HIP, "f32 mfma selftest launch failed");
    if (max_abs_err_out) *max_abs_err_out = err32 > err ? err32 : err;
    if (err32 != 0.0) return fail(CBO_ERR_HIP, "fp32 MFMA lane map differs from what the kernels assume");
    return CBO_OK;
}
