// Arg-max exchange across the GPUs of one node over RCCL (xGMI), without PyTorch: librccl.so.1 is opened with
// dlopen the first time a communicator is asked for, so a single-GPU user never needs it.  The sweep's only
// exchange step is 16 bytes per rank -- (best acquisition value, best GLOBAL candidate index) -- all-gathered and
// reduced identically on every rank (RCCL has no MAXLOC; lowest index wins ties, NaN is maximal: cbo_argmax_pairs).
// Two ways to form the communicator, both part of the C-ABI (include/cbo_hip.h):
//   * one process per GPU (the layout torch.distributed.run / mpirun produce): rank 0 draws a 128-byte id with
//     cbo_comm_unique_id, the launcher's side channel hands it to the other ranks, everyone calls cbo_comm_init_rank;
//   * one process driving several devices: cbo_comm_init_all over the contexts (ncclCommInitAll); collectives of the
//     whole set are then issued inside one group (cbo_comm_argmax_all).
// Any RCCL failure is reported as CBO_ERR_COMM with RCCL's own message.
#include <dlfcn.h>

#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "cbo_internal.h"

using namespace cbo;

namespace {

typedef struct { char internal[128]; } nccl_unique_id;
typedef void *nccl_comm_t;
enum { kNcclInt64 = 4, kNcclFloat64 = 8, kNcclMax = 2 };

struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(nccl_unique_id *) = nullptr;
    int (*CommInitRank)(nccl_comm_t *, int, nccl_unique_id, int) = nullptr;
    int (*CommInitAll)(nccl_comm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;     // (looked up, not required:
    int (*Recv)(void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;           //  cbo_comm_share_factor only)
    const char *(*GetErrorString)(int) = nullptr;
    std::string load_error;
};

Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // CBO_HIP_RCCL_LIB, when set, is the only name tried
        const char *forced = std::getenv("CBO_HIP_RCCL_LIB");
        const bool use_forced = forced && *forced;
        const char *names[] = {use_forced ? forced : "librccl.so.1", use_forced ? nullptr : "librccl.so"};
        for (const char *n : names) {
            if (!n) continue;
            r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
            r.load_error = dlerror();
        }
        if (!r.handle) return;
        auto sym = [&](const char *name) {
            void *p = dlsym(r.handle, name);
            if (!p) r.load_error = std::string("librccl lacks ") + name;
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        r.Send = reinterpret_cast<decltype(r.Send)>(dlsym(r.handle, "ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(dlsym(r.handle, "ncclRecv"));
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommInitAll || !r.CommDestroy || !r.AllGather || !r.AllReduce ||
            !r.GroupStart || !r.GroupEnd || !r.GetErrorString) {
            dlclose(r.handle);
            r.handle = nullptr;
        }
    });
    return r;
}

int need_rccl()
{
    Rccl &r = rccl();
    if (!r.handle) return set_error(CBO_ERR_COMM, "RCCL is not available: " + r.load_error);
    return CBO_OK;
}

int comm_fail(const char *what, int rc)
{
    return set_error(CBO_ERR_COMM, std::string(what) + ": " + rccl().GetErrorString(rc));
}

int hip_fail(const char *what, hipError_t e)
{
    return set_error(CBO_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

}  // namespace

struct cbo_comm {
    cbo_ctx *ctx = nullptr;
    nccl_comm_t comm = nullptr;
    int world = 1, rank = 0;
    hipStream_t stream = nullptr;          // the exchange's own stream (never the legacy default stream)
    int64_t *d_send = nullptr, *d_recv = nullptr;      // 2 and 2 * world int64: (value bits, index) records
    int64_t *h_send = nullptr, *h_recv = nullptr;      // pinned
};

static int comm_buffers(cbo_comm *m)
{
    hipError_t e = hipSetDevice(ctx_device(m->ctx));
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc(&m->d_send, 2 * sizeof(int64_t));
    if (e == hipSuccess) e = hipMalloc(&m->d_recv, 2 * sizeof(int64_t) * (size_t)m->world);
    if (e == hipSuccess) e = hipHostMalloc(&m->h_send, 2 * sizeof(int64_t));
    if (e == hipSuccess) e = hipHostMalloc(&m->h_recv, 2 * sizeof(int64_t) * (size_t)m->world);
    return e == hipSuccess ? CBO_OK : hip_fail("cbo_comm buffers", e);
}

extern "C" int cbo_comm_unique_id(void *id_out)
{
    if (!id_out) return set_error(CBO_ERR_INVALID, "id_out is NULL");
    int rc = need_rccl();
    if (rc != CBO_OK) return rc;
    nccl_unique_id id;
    const int n = rccl().GetUniqueId(&id);
    if (n != 0) return comm_fail("ncclGetUniqueId", n);
    std::memcpy(id_out, id.internal, sizeof(id.internal));
    return CBO_OK;
}

extern "C" void cbo_comm_destroy(cbo_comm *m)
{
    if (!m) return;
    hipSetDevice(ctx_device(m->ctx));
    if (m->stream) hipStreamSynchronize(m->stream);
    if (m->comm) rccl().CommDestroy(m->comm);
    hipFree(m->d_send); hipFree(m->d_recv);
    hipHostFree(m->h_send); hipHostFree(m->h_recv);
    if (m->stream) hipStreamDestroy(m->stream);
    delete m;
}

extern "C" int cbo_comm_init_rank(cbo_ctx *ctx, int world, int rank, const void *id, cbo_comm **out)
{
    if (!ctx || !id || !out) return set_error(CBO_ERR_INVALID, "NULL argument");
    if (world < 1 || rank < 0 || rank >= world) return set_error(CBO_ERR_INVALID, "rank must be in [0, world)");
    int rc = need_rccl();
    if (rc != CBO_OK) return rc;
    cbo_comm *m = new cbo_comm();
    m->ctx = ctx; m->world = world; m->rank = rank;
    rc = comm_buffers(m);
    if (rc != CBO_OK) { cbo_comm_destroy(m); return rc; }
    nccl_unique_id uid;
    std::memcpy(uid.internal, id, sizeof(uid.internal));
    const int n = rccl().CommInitRank(&m->comm, world, uid, rank);
    if (n != 0) { m->comm = nullptr; cbo_comm_destroy(m); return comm_fail("ncclCommInitRank", n); }
    *out = m;
    return CBO_OK;
}

extern "C" int cbo_comm_init_all(int n, cbo_ctx *const *ctxs, cbo_comm **out)
{
    if (n < 1 || !ctxs || !out) return set_error(CBO_ERR_INVALID, "bad argument");
    int rc = need_rccl();
    if (rc != CBO_OK) return rc;
    std::vector<int> devs((size_t)n);
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i]) return set_error(CBO_ERR_INVALID, "a context is NULL");
        devs[(size_t)i] = ctx_device(ctxs[i]);
    }
    std::vector<nccl_comm_t> comms((size_t)n, nullptr);
    const int e = rccl().CommInitAll(comms.data(), n, devs.data());
    if (e != 0) return comm_fail("ncclCommInitAll", e);
    for (int i = 0; i < n; ++i) out[i] = nullptr;
    for (int i = 0; i < n; ++i) {
        cbo_comm *m = new cbo_comm();
        m->ctx = ctxs[i]; m->world = n; m->rank = i; m->comm = comms[(size_t)i];
        out[i] = m;
        rc = comm_buffers(m);
        if (rc != CBO_OK) {
            for (int j = 0; j <= i; ++j) { cbo_comm_destroy(out[j]); out[j] = nullptr; }
            for (int j = i + 1; j < n; ++j) rccl().CommDestroy(comms[(size_t)j]);
            return rc;
        }
    }
    return CBO_OK;
}

extern "C" int cbo_comm_size(const cbo_comm *m, int *world_out, int *rank_out)
{
    if (!m) return set_error(CBO_ERR_INVALID, "comm is NULL");
    if (world_out) *world_out = m->world;
    if (rank_out) *rank_out = m->rank;
    return CBO_OK;
}

// the three steps of one rank's exchange, enqueued on its stream (no synchronisation)
static int exchange_upload(cbo_comm *m, double val, int64_t idx)
{
    hipError_t e = hipSetDevice(ctx_device(m->ctx));
    if (e != hipSuccess) return hip_fail("hipSetDevice", e);
    std::memcpy(&m->h_send[0], &val, sizeof(double));
    m->h_send[1] = idx;
    e = hipMemcpyAsync(m->d_send, m->h_send, 2 * sizeof(int64_t), hipMemcpyHostToDevice, m->stream);
    return e == hipSuccess ? CBO_OK : hip_fail("exchange upload", e);
}

static int exchange_gather(cbo_comm *m)
{
    const int n = rccl().AllGather(m->d_send, m->d_recv, 2, kNcclInt64, m->comm, m->stream);
    return n == 0 ? CBO_OK : comm_fail("ncclAllGather", n);
}

static int exchange_download(cbo_comm *m)
{
    hipError_t e = hipSetDevice(ctx_device(m->ctx));
    if (e == hipSuccess)
        e = hipMemcpyAsync(m->h_recv, m->d_recv, 2 * sizeof(int64_t) * (size_t)m->world, hipMemcpyDeviceToHost, m->stream);
    return e == hipSuccess ? CBO_OK : hip_fail("exchange download", e);
}

static int finish_exchange(cbo_comm *m, double *best_val, int64_t *best_idx)
{
    hipError_t e = hipSetDevice(ctx_device(m->ctx));
    if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
    if (e != hipSuccess) return hip_fail("exchange wait", e);
    std::vector<double> vals;
    std::vector<int64_t> idxs;
    for (int r = 0; r < m->world; ++r) {
        if (m->h_recv[2 * r + 1] == INT64_MAX) continue;           // a rank whose shard is empty
        double v;
        std::memcpy(&v, &m->h_recv[2 * r], sizeof(double));
        vals.push_back(v);
        idxs.push_back(m->h_recv[2 * r + 1]);
    }
    if (vals.empty()) return set_error(CBO_ERR_INVALID, "every rank's shard was empty");
    return cbo_argmax_pairs(vals.data(), idxs.data(), (int)vals.size(), best_val, best_idx);
}

extern "C" int cbo_comm_argmax(cbo_comm *m, double val, int64_t idx, double *best_val, int64_t *best_idx)
{
    if (!m || !best_val || !best_idx) return set_error(CBO_ERR_INVALID, "NULL argument");
    int rc = exchange_upload(m, val, idx);
    if (rc == CBO_OK) rc = exchange_gather(m);
    if (rc == CBO_OK) rc = exchange_download(m);
    if (rc != CBO_OK) return rc;
    return finish_exchange(m, best_val, best_idx);
}

extern "C" int cbo_comm_argmax_all(int n, cbo_comm *const *comms, const double *vals, const int64_t *idxs,
                                   double *best_val, int64_t *best_idx)
{
    if (n < 1 || !comms || !vals || !idxs || !best_val || !best_idx) return set_error(CBO_ERR_INVALID, "bad argument");
    int rc = CBO_OK;
    for (int i = 0; i < n && rc == CBO_OK; ++i) rc = comms[i] ? exchange_upload(comms[i], vals[i], idxs[i])
                                                               : set_error(CBO_ERR_INVALID, "a communicator is NULL");
    if (rc != CBO_OK) return rc;
    // one process drives every rank: the collectives of all ranks go out as one group (they are launched at
    // ncclGroupEnd), the downloads are enqueued behind them
    int g = rccl().GroupStart();
    if (g != 0) return comm_fail("ncclGroupStart", g);
    for (int i = 0; i < n && rc == CBO_OK; ++i) rc = exchange_gather(comms[i]);
    g = rccl().GroupEnd();
    if (rc != CBO_OK) return rc;
    if (g != 0) return comm_fail("ncclGroupEnd", g);
    for (int i = 0; i < n && rc == CBO_OK; ++i) rc = exchange_download(comms[i]);
    if (rc != CBO_OK) return rc;
    // every rank holds the same gathered records and reduces them identically; rank 0's answer is returned and the
    // others are checked against it
    double v0 = 0.0;
    int64_t i0 = -1;
    for (int i = 0; i < n; ++i) {
        double v;
        int64_t ix;
        rc = finish_exchange(comms[i], &v, &ix);
        if (rc != CBO_OK) return rc;
        if (i == 0) { v0 = v; i0 = ix; }
        else if (ix != i0 || std::memcmp(&v, &v0, sizeof(double)) != 0)
            return set_error(CBO_ERR_COMM, "ranks disagree on the arg-max after the all-gather");
    }
    *best_val = v0;
    *best_idx = i0;
    return CBO_OK;
}

// max over the ranks of one double (bench.py: the slowest rank's time); also serves as a barrier
extern "C" int cbo_comm_max_f64(cbo_comm *m, double value, double *max_out)
{
    if (!m || !max_out) return set_error(CBO_ERR_INVALID, "NULL argument");
    hipError_t e = hipSetDevice(ctx_device(m->ctx));
    if (e != hipSuccess) return hip_fail("hipSetDevice", e);
    double *hs = reinterpret_cast<double *>(m->h_send), *hr = reinterpret_cast<double *>(m->h_recv);
    double *ds = reinterpret_cast<double *>(m->d_send), *dr = reinterpret_cast<double *>(m->d_recv);
    hs[0] = value;
    e = hipMemcpyAsync(ds, hs, sizeof(double), hipMemcpyHostToDevice, m->stream);
    if (e != hipSuccess) return hip_fail("max upload", e);
    const int n = rccl().AllReduce(ds, dr, 1, kNcclFloat64, kNcclMax, m->comm, m->stream);
    if (n != 0) return comm_fail("ncclAllReduce", n);
    e = hipMemcpyAsync(hr, dr, sizeof(double), hipMemcpyDeviceToHost, m->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
    if (e != hipSuccess) return hip_fail("max download", e);
    *max_out = hr[0];
    return CBO_OK;
}

extern "C" int cbo_comm_barrier(cbo_comm *m)
{
    double dummy = 0.0;
    return cbo_comm_max_f64(m, 0.0, &dummy);
}

// One int64 from every rank, in rank order (the ladder's outcome flags, cbo_with_oop_amd/sharding.py fit_over_ranks).
extern "C" int cbo_comm_gather_i64(cbo_comm *m, int64_t value, int64_t *out)
{
    if (!m || !out) return set_error(CBO_ERR_INVALID, "NULL argument");
    hipError_t e = hipSetDevice(ctx_device(m->ctx));
    if (e != hipSuccess) return hip_fail("hipSetDevice", e);
    m->h_send[0] = value;
    e = hipMemcpyAsync(m->d_send, m->h_send, sizeof(int64_t), hipMemcpyHostToDevice, m->stream);
    if (e != hipSuccess) return hip_fail("gather upload", e);
    const int n = rccl().AllGather(m->d_send, m->d_recv, 1, kNcclInt64, m->comm, m->stream);
    if (n != 0) return comm_fail("ncclAllGather", n);
    e = hipMemcpyAsync(m->h_recv, m->d_recv, sizeof(int64_t) * (size_t)m->world, hipMemcpyDeviceToHost, m->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
    if (e != hipSuccess) return hip_fail("gather download", e);
    for (int r = 0; r < m->world; ++r) out[r] = m->h_recv[r];
    return CBO_OK;
}

// The rows [begin, end) of the factor that owner number i of n sends (multiples of 128 rows, the first slices one block
// longer when the blocks do not divide).
static void factor_slice(int64_t n_pad, int n, int i, int64_t *begin, int64_t *end)
{
    const int64_t blocks = n_pad / 128, base = blocks / n, extra = blocks % n;
    const int64_t b0 = (int64_t)i * base + (i < extra ? i : extra);
    *begin = 128 * b0;
    *end = 128 * (b0 + base + (i < extra ? 1 : 0));
}

// The ranks of `owners` hold model g's factor at `level` of the jitchol ladder (each fitted it itself: the posterior
// is replicated); the ranks of `needers` do not (they tried a level that failed).  Every needer receives the factor in
// |owners| row slices, one from each owner -- over xGMI that is |owners| links side by side -- and adopts it (fitted,
// tries = level).  Called by every rank of the communicator with the same lists; a rank in neither list does nothing.
// Rows travel whole (the upper triangle, the unused lower part, the right-hand-side column: lda doubles), with the
// 16x16 diagonal inverses of the same rows.
extern "C" int cbo_comm_share_factor(cbo_comm *m, cbo_gp *g, int level, const int *owners, int n_owners,
                                     const int *needers, int n_needers)
{
    if (!m || !g || !owners || n_owners <= 0 || (n_needers > 0 && !needers))
        return set_error(CBO_ERR_INVALID, "bad argument");
    double *A = nullptr, *invDt = nullptr;
    int64_t lda = 0, n_pad = 0;
    cbo_ctx *ctx = nullptr;
    int rc = gp_factor_view(g, &A, &lda, &n_pad, &invDt, &ctx);
    if (rc != CBO_OK) return rc;
    if (ctx != m->ctx) return set_error(CBO_ERR_INVALID, "the model lives on another context than the communicator");
    int my_owner = -1;
    bool i_need = false;
    for (int i = 0; i < n_owners; ++i) {
        if (owners[i] < 0 || owners[i] >= m->world) return set_error(CBO_ERR_INVALID, "owner rank out of range");
        if (owners[i] == m->rank) my_owner = i;
    }
    for (int i = 0; i < n_needers; ++i) {
        if (needers[i] < 0 || needers[i] >= m->world) return set_error(CBO_ERR_INVALID, "needer rank out of range");
        if (needers[i] == m->rank) i_need = true;
    }
    if (my_owner >= 0 && i_need) return set_error(CBO_ERR_INVALID, "a rank cannot both hold and need the factor");
    // What only THIS rank can know -- it is listed as an owner but does not hold the factor, its RCCL lacks the point-to-point
    // calls -- must not make it leave while its peers enter the send / receive group and wait for it for ever: every rank of
    // the communicator exchanges a status word first (one all-gather of one integer), and all refuse together.
    int64_t status = 0;
    if (my_owner >= 0 && !gp_is_fitted_at(g, level)) status = 1;
    rc = need_rccl();
    if (rc != CBO_OK) return rc;
    if (!rccl().Send || !rccl().Recv) status = 2;
    std::vector<int64_t> all((size_t)m->world, 0);
    rc = cbo_comm_gather_i64(m, status, all.data());
    if (rc != CBO_OK) return rc;
    for (int r = 0; r < m->world; ++r) {
        if (all[(size_t)r] == 1)
            return set_error(CBO_ERR_INVALID, "rank " + std::to_string(r) + " is listed as an owner but does not hold the factor at that level");
        if (all[(size_t)r] == 2)
            return set_error(CBO_ERR_COMM, "the RCCL of rank " + std::to_string(r) + " has no ncclSend / ncclRecv");
    }
    if (n_needers == 0 || (my_owner < 0 && !i_need)) return CBO_OK;     // nobody lacks the factor / this rank is not involved
    hipError_t e = hipSetDevice(ctx_device(m->ctx));
    // the model's own stream has produced (owner) or will consume (needer) the factor: order the exchange behind it
    if (e == hipSuccess) e = hipStreamSynchronize(ctx_stream(m->ctx));
    if (e != hipSuccess) return hip_fail("share_factor: stream", e);
    int n = rccl().GroupStart();
    if (n != 0) return comm_fail("ncclGroupStart", n);
    for (int i = 0; i < n_owners && n == 0; ++i) {
        int64_t r0 = 0, r1 = 0;
        factor_slice(n_pad, n_owners, i, &r0, &r1);
        if (r1 <= r0) continue;
        double *rows = A + r0 * lda, *inv = invDt + (r0 / 16) * 256;
        const size_t n_rows = (size_t)((r1 - r0) * lda), n_inv = (size_t)((r1 - r0) / 16 * 256);
        if (i == my_owner) {
            for (int k = 0; k < n_needers && n == 0; ++k) {
                n = rccl().Send(rows, n_rows, kNcclFloat64, needers[k], m->comm, m->stream);
                if (n == 0) n = rccl().Send(inv, n_inv, kNcclFloat64, needers[k], m->comm, m->stream);
            }
        } else if (i_need) {
            n = rccl().Recv(rows, n_rows, kNcclFloat64, owners[i], m->comm, m->stream);
            if (n == 0) n = rccl().Recv(inv, n_inv, kNcclFloat64, owners[i], m->comm, m->stream);
        }
    }
    const int ge = rccl().GroupEnd();
    if (n != 0) return comm_fail("ncclSend / ncclRecv", n);
    if (ge != 0) return comm_fail("ncclGroupEnd", ge);
    e = hipStreamSynchronize(m->stream);
    if (e != hipSuccess) return hip_fail("share_factor: wait", e);
    return i_need ? gp_adopt_received_factor(g, level) : CBO_OK;
}

// The needer's side of cbo_comm_share_factor with device copies standing in for ncclRecv: `dst` (the same data and
// hyper-parameters as `src`, not fitted at `level`) takes the factor of `src` in the n_owners row slices of factor_slice --
// whole rows of lda doubles and the 16x16 diagonal inverses of the same rows, slice by slice into the places the receives
// write -- and adopts it.  A one-GPU box cannot run the transfer between ranks (with one rank nobody lacks the factor); this
// pins everything around it: the split, the layout of what travels, the adoption (tests/test_parity_gpu.py).
extern "C" int cbo_gp_take_factor_slices(cbo_gp *dst, cbo_gp *src, int level, int n_owners)
{
    if (!dst || !src || n_owners <= 0) return set_error(CBO_ERR_INVALID, "bad argument");
    double *As = nullptr, *invs = nullptr, *Ad = nullptr, *invd = nullptr;
    int64_t ldas = 0, ns = 0, ldad = 0, nd = 0;
    cbo_ctx *cs = nullptr, *cd = nullptr;
    int rc = gp_factor_view(src, &As, &ldas, &ns, &invs, &cs);
    if (rc == CBO_OK) rc = gp_factor_view(dst, &Ad, &ldad, &nd, &invd, &cd);
    if (rc != CBO_OK) return rc;
    if (cs != cd || ldas != ldad || ns != nd) return set_error(CBO_ERR_INVALID, "the two models do not have the same shape and context");
    if (!gp_is_fitted_at(src, level)) return set_error(CBO_ERR_INVALID, "the source does not hold the factor at that level");
    hipError_t e = hipSetDevice(ctx_device(cs));
    if (e == hipSuccess) e = hipStreamSynchronize(ctx_stream(cs));
    for (int i = 0; i < n_owners && e == hipSuccess; ++i) {
        int64_t r0 = 0, r1 = 0;
        factor_slice(ns, n_owners, i, &r0, &r1);
        if (r1 <= r0) continue;
        e = hipMemcpyAsync(Ad + r0 * ldad, As + r0 * ldas, sizeof(double) * (size_t)((r1 - r0) * ldas), hipMemcpyDeviceToDevice, ctx_stream(cs));
        if (e == hipSuccess)
            e = hipMemcpyAsync(invd + (r0 / 16) * 256, invs + (r0 / 16) * 256, sizeof(double) * (size_t)((r1 - r0) / 16 * 256),
                               hipMemcpyDeviceToDevice, ctx_stream(cs));
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx_stream(cs));
    if (e != hipSuccess) return hip_fail("take_factor_slices", e);
    return gp_adopt_received_factor(dst, level);
}
