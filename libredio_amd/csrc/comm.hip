// comm.hip -- the ONE collective of the hot path behind the C ABI (include/redio.h, redio_comm_* / redio_pfb_exchange):
// the regrouping step of the time-sharded channelizer (BASELINE.json configs[3]).  Every rank runs the
// channelizer on its own time slice with the per-destination output layout [group][row][nchan/G]
// (redio_pfb_enqueue(..., ngroups = G)); after the exchange rank g holds [all rows, in rank = time order][its channels].
// RCCL is used directly -- ncclGroupStart; ncclSend / ncclRecv per peer; ncclGroupEnd -- over xGMI's dedicated
// point-to-point links (every pair of GPUs has its own link, so all G-1 transfers of a rank run concurrently).
// librccl.so is loaded on first use (dlopen), so libredio.so itself does not depend on it: a host that never
// shards the channelizer never maps RCCL, and the CPU-side ABI tests load the library without it.
// One process may own several ranks (redio_comm_init_all: the thread-per-block host of src/ratpak.rs:60-185 drives
// every GPU from one process) or one rank per process (redio_comm_init_rank with an id from redio_comm_unique_id
// distributed by the launcher).
#include "../../include/redio.h"
#include <hip/hip_runtime.h>
#include <rccl/rccl.h> // types only: the entry points are resolved with dlsym
#include <dlfcn.h>
#include <mutex>
#include <new>
#include <stdio.h>
#include <string.h>
#include <vector>

namespace {
struct Rccl {
    void *so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
Rccl g_rccl;
std::once_flag g_once;
thread_local char g_last_error[256] = ""; // per calling thread: one block thread per GPU may fail independently
char g_load_error[256] = "";              // why librccl.so could not be used: written once (under g_once), read by every thread after

void load_rccl()
{
    // a library of this name that is already mapped (e.g. the one PyTorch ships) is reused by the loader
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        g_rccl.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.so) break;
    }
    if (!g_rccl.so) { snprintf(g_load_error, sizeof g_load_error, "librccl.so not loadable: %s", dlerror()); return; }
#define RD_SYM(field, name) \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.so, name)); \
    if (!g_rccl.field) { snprintf(g_load_error, sizeof g_load_error, "librccl.so lacks %s", name); return; }
    RD_SYM(GetUniqueId, "ncclGetUniqueId")
    RD_SYM(CommInitRank, "ncclCommInitRank")
    RD_SYM(CommInitAll, "ncclCommInitAll")
    RD_SYM(CommDestroy, "ncclCommDestroy")
    RD_SYM(Send, "ncclSend")
    RD_SYM(Recv, "ncclRecv")
    RD_SYM(GroupStart, "ncclGroupStart")
    RD_SYM(GroupEnd, "ncclGroupEnd")
    RD_SYM(GetErrorString, "ncclGetErrorString")
#undef RD_SYM
    g_rccl.ok = true;
}
const Rccl *rccl()
{
    std::call_once(g_once, load_rccl);
    if (g_rccl.ok) return &g_rccl;
    snprintf(g_last_error, sizeof g_last_error, "%s", g_load_error); // every failing caller gets the text, not only the first thread
    return nullptr;
}
// makes `device` current for the life of a call and puts the caller's device back (the device is per-thread state: a block
// thread that drives several GPUs must not find it changed by an exchange)
struct DeviceScope {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit DeviceScope(int device)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        err = hipSetDevice(device);
    }
    ~DeviceScope() { if (prev >= 0) hipSetDevice(prev); }
};
int nccl_rc(const Rccl *r, ncclResult_t e, const char *what)
{
    if (e == ncclSuccess) return REDIO_OK;
    snprintf(g_last_error, sizeof g_last_error, "%s: %s", what, r->GetErrorString(e));
    return REDIO_ERR_COMM;
}
} // namespace

struct redio_comm {
    ncclComm_t comm;
    int rank, nranks, device;
};

extern "C" const char *redio_comm_last_error(void) { return g_last_error; }

extern "C" int redio_comm_unique_id(void *id128)
{
    if (!id128) return REDIO_ERR_ARG;
    const Rccl *r = rccl();
    if (!r) return REDIO_ERR_COMM;
    static_assert(sizeof(ncclUniqueId) == REDIO_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    const int rc = nccl_rc(r, r->GetUniqueId(&id), "ncclGetUniqueId");
    if (rc) return rc;
    memcpy(id128, &id, sizeof id);
    return REDIO_OK;
}

extern "C" int redio_comm_init_rank(redio_comm **c, int nranks, int rank, const void *id128)
{
    if (!c) return REDIO_ERR_ARG;
    *c = nullptr;
    if (!id128 || nranks < 1 || rank < 0 || rank >= nranks) return REDIO_ERR_ARG;
    const Rccl *r = rccl();
    if (!r) return REDIO_ERR_COMM;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return REDIO_ERR_NO_DEVICE;
    redio_comm *p = new (std::nothrow) redio_comm();
    if (!p) return REDIO_ERR_NOMEM;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    const int rc = nccl_rc(r, r->CommInitRank(&p->comm, nranks, id, rank), "ncclCommInitRank");
    if (rc) { delete p; return rc; }
    p->rank = rank; p->nranks = nranks; p->device = dev;
    *c = p;
    return REDIO_OK;
}

extern "C" int redio_comm_init_all(redio_comm **comms, int ndev, const int *devices)
{
    if (!comms || ndev < 1) return REDIO_ERR_ARG;
    for (int i = 0; i < ndev; ++i) comms[i] = nullptr;
    const Rccl *r = rccl();
    if (!r) return REDIO_ERR_COMM;
    std::vector<int> devs((size_t)ndev);
    for (int i = 0; i < ndev; ++i) devs[(size_t)i] = devices ? devices[i] : i;
    std::vector<ncclComm_t> cs((size_t)ndev);
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    const int rc = nccl_rc(r, r->CommInitAll(cs.data(), ndev, devs.data()), "ncclCommInitAll");
    if (have_cur) hipSetDevice(cur); // the caller's current device is left as it was
    if (rc) return rc;
    for (int i = 0; i < ndev; ++i) {
        redio_comm *p = new (std::nothrow) redio_comm();
        if (!p) {
            for (int j = 0; j < i; ++j) { delete comms[j]; comms[j] = nullptr; }
            for (int j = 0; j < ndev; ++j) r->CommDestroy(cs[(size_t)j]);
            return REDIO_ERR_NOMEM;
        }
        p->comm = cs[(size_t)i]; p->rank = i; p->nranks = ndev; p->device = devs[(size_t)i];
        comms[i] = p;
    }
    return REDIO_OK;
}

extern "C" int redio_comm_destroy(redio_comm *c)
{
    if (!c) return REDIO_OK;
    const Rccl *r = rccl();
    if (r) r->CommDestroy(c->comm);
    delete c;
    return REDIO_OK;
}
extern "C" int redio_comm_rank(const redio_comm *c) { return c ? c->rank : -1; }
extern "C" int redio_comm_size(const redio_comm *c) { return c ? c->nranks : 0; }

// One rank's transfers of an exchange: to every peer q `sendn[q]` floats from `sendp[q]`, from every peer `recvn[q]` floats into
// `recvp[q]`.  Messages are cut into pieces of at most 2^27 floats (512 MiB), one RCCL group per piece index: a single
// ncclSend / ncclRecv pair above 1 GiB was measured to deliver only its first gigabyte (RCCL 2.26.6, send to self on one MI355X;
// tools/rccl_self_probe.py is the probe), and smaller pieces also let the fabric interleave the peers.
constexpr size_t COMM_PIECE = (size_t)1 << 27;
struct PeerXfer { const float *sendp; size_t sendn; float *recvp; size_t recvn; };
static int xfer_group(const Rccl *r, redio_comm *c, const PeerXfer *x, size_t piece, hipStream_t st, bool &any)
{
    int rc = REDIO_OK;
    for (int q = 0; q < c->nranks && rc == REDIO_OK; ++q) {
        const size_t o = piece * COMM_PIECE;
        if (x[q].sendn > o) {
            const size_t n = x[q].sendn - o < COMM_PIECE ? x[q].sendn - o : COMM_PIECE;
            rc = nccl_rc(r, r->Send(x[q].sendp + o, n, ncclFloat, q, c->comm, st), "ncclSend");
            any = true;
        }
        if (rc == REDIO_OK && x[q].recvn > o) {
            const size_t n = x[q].recvn - o < COMM_PIECE ? x[q].recvn - o : COMM_PIECE;
            rc = nccl_rc(r, r->Recv(x[q].recvp + o, n, ncclFloat, q, c->comm, st), "ncclRecv");
            any = true;
        }
    }
    return rc;
}
static size_t xfer_pieces(const PeerXfer *x, int n)
{
    size_t most = 0;
    for (int q = 0; q < n; ++q) {
        most = x[q].sendn > most ? x[q].sendn : most;
        most = x[q].recvn > most ? x[q].recvn : most;
    }
    return (most + COMM_PIECE - 1) / COMM_PIECE;
}
// all transfers of ONE rank on its stream
static int xfer_rank(const Rccl *r, redio_comm *c, const std::vector<PeerXfer> &x, hipStream_t st)
{
    const size_t np = xfer_pieces(x.data(), c->nranks);
    for (size_t k = 0; k < np; ++k) {
        int rc = nccl_rc(r, r->GroupStart(), "ncclGroupStart");
        if (rc) return rc;
        bool any = false;
        rc = xfer_group(r, c, x.data(), k, st, any);
        const int rc2 = nccl_rc(r, r->GroupEnd(), "ncclGroupEnd");
        if (rc || rc2) return rc ? rc : rc2;
    }
    return REDIO_OK;
}

// offsets (in rows) of every rank's block in the regrouped output, and their sum
static size_t row_offsets(const size_t *rows_per_rank, int n, std::vector<size_t> &off)
{
    off.resize((size_t)n);
    size_t tot = 0;
    for (int q = 0; q < n; ++q) { off[(size_t)q] = tot; tot += rows_per_rank[q]; }
    return tot;
}

// The same exchange with the receive placement given explicitly: rank q's rows land at row out_row_offset[q] of d_out.  A slice
// that is analysed in several pieces (so that the exchange of piece i runs beside the analysis of piece i + 1 on another HIP
// stream) places piece i of rank q at q * rows_of_a_whole_slice + i * rows_of_a_piece and so builds the time-ordered result in place.
extern "C" int redio_pfb_exchange_at(redio_comm *c, const void *d_grouped, void *d_out, const size_t *rows_per_rank, const size_t *out_row_offset,
                                     size_t chans_per_rank, void *stream)
{
    if (!c || !rows_per_rank || !out_row_offset || chans_per_rank == 0) return REDIO_ERR_ARG;
    const Rccl *r = rccl();
    if (!r) return REDIO_ERR_COMM;
    const size_t mine = rows_per_rank[c->rank];
    size_t total = 0;
    for (int q = 0; q < c->nranks; ++q) total += rows_per_rank[q];
    if ((mine && !d_grouped) || (total && !d_out)) return REDIO_ERR_ARG;
    const DeviceScope dev_scope(c->device); // the caller's current device is left as it was
    if (dev_scope.err != hipSuccess) return REDIO_ERR_HIP_BASE - (int)dev_scope.err;
    const size_t fl = 2 * chans_per_rank;
    std::vector<PeerXfer> x((size_t)c->nranks);
    for (int q = 0; q < c->nranks; ++q)
        x[(size_t)q] = {(const float *)d_grouped + (size_t)q * mine * fl, mine * fl, (float *)d_out + out_row_offset[q] * fl, rows_per_rank[q] * fl};
    return xfer_rank(r, c, x, (hipStream_t)stream);
}

extern "C" int redio_pfb_exchange(redio_comm *c, const void *d_grouped, void *d_out, const size_t *rows_per_rank, size_t chans_per_rank, void *stream)
{
    if (!c || !rows_per_rank || chans_per_rank == 0) return REDIO_ERR_ARG;
    const Rccl *r = rccl();
    if (!r) return REDIO_ERR_COMM;
    std::vector<size_t> off;
    const size_t total = row_offsets(rows_per_rank, c->nranks, off);
    const size_t mine = rows_per_rank[c->rank];
    if ((mine && !d_grouped) || (total && !d_out)) return REDIO_ERR_ARG;
    const DeviceScope dev_scope(c->device); // the caller's current device is left as it was
    if (dev_scope.err != hipSuccess) return REDIO_ERR_HIP_BASE - (int)dev_scope.err;
    const size_t fl = 2 * chans_per_rank; // floats per row of one group
    std::vector<PeerXfer> x((size_t)c->nranks);
    for (int q = 0; q < c->nranks; ++q) // to rank q: my rows of q's channels (group q of my layout); from rank q: its rows of my channels
        x[(size_t)q] = {(const float *)d_grouped + (size_t)q * mine * fl, mine * fl, (float *)d_out + off[(size_t)q] * fl, rows_per_rank[q] * fl};
    return xfer_rank(r, c, x, (hipStream_t)stream);
}

// every rank of one process in one call (comms from redio_comm_init_all): one RCCL group around all sends and receives
extern "C" int redio_pfb_exchange_all(redio_comm *const *comms, int ndev, const void *const *d_grouped, void *const *d_out,
                                      const size_t *rows_per_rank, size_t chans_per_rank, void *const *streams)
{
    if (!comms || ndev < 1 || !d_grouped || !d_out || !rows_per_rank || chans_per_rank == 0) return REDIO_ERR_ARG;
    const Rccl *r = rccl();
    if (!r) return REDIO_ERR_COMM;
    std::vector<size_t> off;
    row_offsets(rows_per_rank, ndev, off);
    const size_t fl = 2 * chans_per_rank;
    std::vector<std::vector<PeerXfer>> xs((size_t)ndev, std::vector<PeerXfer>((size_t)ndev));
    size_t np = 0;
    for (int g = 0; g < ndev; ++g) {
        redio_comm *c = comms[g];
        if (!c || c->nranks != ndev) return REDIO_ERR_ARG;
        const size_t mine = rows_per_rank[g];
        for (int q = 0; q < ndev; ++q)
            xs[(size_t)g][(size_t)q] = {(const float *)d_grouped[g] + (size_t)q * mine * fl, mine * fl, (float *)d_out[g] + off[(size_t)q] * fl, rows_per_rank[q] * fl};
        const size_t p = xfer_pieces(xs[(size_t)g].data(), ndev);
        np = p > np ? p : np;
    }
    for (size_t k = 0; k < np; ++k) { // one group per piece index around every rank's sends and receives
        int rc = nccl_rc(r, r->GroupStart(), "ncclGroupStart");
        if (rc) return rc;
        for (int g = 0; g < ndev && rc == REDIO_OK; ++g) {
            bool any = false;
            rc = xfer_group(r, comms[g], xs[(size_t)g].data(), k, streams ? (hipStream_t)streams[g] : nullptr, any);
        }
        const int rc2 = nccl_rc(r, r->GroupEnd(), "ncclGroupEnd");
        if (rc || rc2) return rc ? rc : rc2;
    }
    return REDIO_OK;
}
