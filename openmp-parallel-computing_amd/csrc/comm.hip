// comm.hip -- row-shard halo exchange over RCCL (ncclSend/ncclRecv on xGMI) and the sharded pipeline step, host side.
//
// The reference has no distributed layer (OpenMP threads over one shared image, monolithic/src/sobel.c:10-24 and
// parallel_to_grayscale.c:7-16).  Its row-parallel `#pragma omp parallel for` becomes: one process per GPU, each owning
// a contiguous block of rows, and one exchange step -- the row above and the row below a shard come from the
// neighbouring ranks.  At 8 GPUs a 16384^2 frame leaves ~25 us of kernel time per shard, so the
// host cost of the exchange decides the scaling curve; this file issues the two send/recv pairs as one ncclGroup on a
// high-priority side stream straight from C++ (no Python between the calls) and overlaps them either with the interior
// rows of the same frame (gh_sharded_*_step_dev) or, for a stream of frames, with the whole shard of an earlier frame
// (gh_sharded_*_step_pipelined_dev: the halo of frame k+2 travels while frame k runs as one launch).
//
// RCCL is loaded with dlopen at gh_comm_init time: the single-GPU library, the CLIs and the drop-in boundary do not
// depend on librccl.so.
#include "gh_common.h"

#include <dlfcn.h>

#include <cstring>
#include <mutex>

namespace {

// the slice of the RCCL API used here (rccl.h: ncclGetUniqueId / ncclCommInitRank / ncclSend / ncclRecv / ...)
constexpr int kIdBytes = 128;
constexpr size_t kFlagBytes = 512;       // flag (dword 0), trace stamps (bytes 64..127), started (dword 32), standstill counters (dwords 40, 41), warm-up rows (bytes 256..511)
struct NcclId { char internal[kIdBytes]; };
using nccl_comm_t = void *;
enum { kNcclUint8 = 1, kNcclInt32 = 2, kNcclUint64 = 5, kNcclSum = 0, kNcclMax = 2, kNcclMin = 3 };   // ncclDataType_t / ncclRedOp_t values of rccl.h

struct Rccl {
    void *so = nullptr;
    int (*GetUniqueId)(NcclId *) = nullptr;
    int (*CommInitRank)(nccl_comm_t *, int, NcclId, int) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
std::once_flag g_rccl_once;
char g_rccl_err[256];

void load_rccl()
{
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        g_rccl.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.so) break;
    }
    if (!g_rccl.so) {
        snprintf(g_rccl_err, sizeof g_rccl_err, "cannot load librccl.so: %s", dlerror());
        return;
    }
#define GH_SYM(field, name)                                                                          \
    do {                                                                                             \
        *reinterpret_cast<void **>(&g_rccl.field) = dlsym(g_rccl.so, name);                          \
        if (!g_rccl.field) { snprintf(g_rccl_err, sizeof g_rccl_err, "librccl.so lacks %s", name); g_rccl.so = nullptr; return; }  \
    } while (0)
    GH_SYM(GetUniqueId, "ncclGetUniqueId");
    GH_SYM(CommInitRank, "ncclCommInitRank");
    GH_SYM(CommDestroy, "ncclCommDestroy");
    GH_SYM(GroupStart, "ncclGroupStart");
    GH_SYM(GroupEnd, "ncclGroupEnd");
    GH_SYM(Send, "ncclSend");
    GH_SYM(Recv, "ncclRecv");
    GH_SYM(AllReduce, "ncclAllReduce");
    GH_SYM(GetErrorString, "ncclGetErrorString");
#undef GH_SYM
}

int need_rccl()
{
    std::call_once(g_rccl_once, load_rccl);
    if (!g_rccl.so) return gh_set_error(GH_ERR_HIP, "RCCL unavailable: %s", g_rccl_err);
    return GH_OK;
}

#define GH_NCCL_TRY(expr)                                                                               \
    do {                                                                                                \
        const int r_ = (expr);                                                                          \
        if (r_ != 0) return gh_set_error(GH_ERR_HIP, "%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(r_), __FILE__, __LINE__);  \
    } while (0)

struct Comm {
    nccl_comm_t nccl = nullptr;
    int rank = 0, world = 1;
    hipStream_t side = nullptr;          // the exchange runs here, beside the interior rows on the caller's stream
    hipEvent_t ready = nullptr, landed = nullptr;
    int loopback = 0;                    // rehearsal on one GPU (gh_comm_set_loopback): 1 = a middle shard whose two neighbours are this rank
                                         // (ncclSend/ncclRecv to self), 2 = the TOP shard (frame border above, itself below), 3 = the BOTTOM shard
    bool prio_ok = true;                 // the caller's stream last seen does not share the side stream's priority (halo-flag form, see sharded_step)
    hipStream_t prio_stream = nullptr;
    bool prio_seen = false;
    int side_prio = 0;
    // pipelined steps: buffers whose halo rows the side stream has already been asked to fill, and the event that says
    // they have landed.  Requests come one buffer at a time (a halo two or three frames ahead) or as a BATCH -- the halos of
    // several future frames in one ncclGroup, one event for all of them (gh_sharded_prefetch_halos_dev): a pool of kPrefetch
    // events, each shared by the table entries of its batch; `pf_waited` says the caller's stream has already been told to wait
    // for it.  An event is taken for a new request only when none of its entries is still waiting to be consumed (round 2 took
    // them round robin: with batches of 3-4 the single-request fallback recycled the event of the batch just requested for the
    // NEXT frames and forgot it, so every step fell back to its own exchange -- correct pixels, one exchange per step).
    static constexpr int kPrefetch = 8, kEntries = 32;
    const void *pf_buf[kEntries] = {};
    int pf_event_of[kEntries] = {};
    hipEvent_t pf_landed[kPrefetch] = {};
    bool pf_waited[kPrefetch] = {};
    int pf_next = 0, pf_next_entry = 0;
    unsigned long long exchanges = 0;    // ncclGroups issued for halo rows (gh_comm_exchange_count: tests and bench.py read it)
    // halo-flag form of the sharded step (gh_common.h): flag_dev[0] = flag the edge bands poll, flag_dev[32] = started (a cache
    // line of its own), timeouts = host-mapped dword
    uint32_t *flag_dev = nullptr;
    uint32_t *started_sig = nullptr;     // `started` in signal memory (hipMallocSignalMemory) where the runtime has it: hipStreamWaitValue32 can poll it
    uint32_t *timeouts_host = nullptr, *timeouts_dev = nullptr;
    uint32_t epoch = 0;                  // kernel launches that store `started`
    uint32_t xepoch = 0;                 // halo exchanges that set `flag` behind them
    // stream-of-frames step (gh_sharded_*_stream_step_dev): the buffer whose halo rows the previous call sent on their way, and the
    // value `flag` takes once they have landed
    const void *stream_buf = nullptr;
    uint32_t stream_xepoch = 0;
    int step_form = -1;                  // gh_comm_set_step_form: -1 = GH_SHARD_STEP_FORM / the default
    int halo_wait_ms = -1;               // gh_comm_set_halo_wait_ms: -1 = GH_HALO_WAIT_MS / the default (10 s)
    // review of the halo-flag form (halo_flag_review): what the edge bands have reported up to the last look, and the verdict
    uint32_t seen_ticks = 0, seen_waits = 0, review_epoch = 0;
    bool reviewed = false;
    int side_replacements = 0;           // side streams given up because the exchange did not run beside the launch on them
    hipStream_t retired_side[2] = {};
    float exchange_alone_us = 11.f;      // one halo exchange with nothing beside it, timed at gh_comm_init / gh_comm_set_loopback (what the serial form pays per step)
    bool auto_serial = false;            // the exchange was found NOT to run beside the launch: steps whose form is not forced take the serial form
    uint32_t last_avg_wait_ticks = 0;
};

// The stream-of-frames step remembers which buffer's halo rows the previous call sent on their way (stream_buf / stream_xepoch).  Anything
// else that touches halo rows or changes how a step runs -- a single-frame step, a direct exchange, a change of form or of the loopback
// mode -- makes that memory stale: a later stream step on the same address would skip its exchange and wait for an epoch that has
// already passed.  Such calls forget the announcement; the next stream step then exchanges `cur` beside its own launch.
void forget_announced(Comm *c) { c->stream_buf = nullptr; }

// the two neighbours of this rank's shard: rank -/+ 1, -1 = the frame border; the loopback modes stand in for the three kinds of shard
void neighbours(const Comm *c, int &up, int &down)
{
    up = c->rank > 0 ? c->rank - 1 : -1;
    down = c->rank + 1 < c->world ? c->rank + 1 : -1;
    if (c->loopback == 1) up = down = c->rank;                   // middle shard: halo rows = its own first / last row
    else if (c->loopback == 2) { up = -1; down = c->rank; }      // top shard: zero border above, waits for ONE halo row (below)
    else if (c->loopback == 3) { up = c->rank; down = -1; }      // bottom shard: the mirror image
}

// behind the exchange on the side stream: the flag the edge bands poll -- and, on the way, what the edge bands of EARLIER launches reported
// about their waits (gh_halo_wait: device counters at flag + 40 / + 41) goes to the host-mapped words the review reads (plain stores)
__global__ void k_set_halo_flag(uint32_t *flag, uint32_t epoch, uint32_t *host_words)
{
    GH_HALO_TRACE_AT(flag, 2);
    if (threadIdx.x == 0) {
        __hip_atomic_store(flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        host_words[2] = __hip_atomic_load(flag + 40, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        host_words[4] = __hip_atomic_load(flag + 41, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ void k_store_u32(uint32_t *p, uint32_t v)
{
    if (threadIdx.x == 0) __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// in front of the exchange on the side stream: until the shard kernel of this step has begun on the caller's stream (gh_common.h)
__global__ void k_wait_started(const uint32_t *started, uint32_t epoch, unsigned long long ticks, uint32_t *timeouts, const uint32_t *flag)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        const uint32_t v = __hip_atomic_load(started, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((int32_t)(v - epoch) >= 0) { GH_HALO_TRACE_AT(flag, 1); return; }
        if (__builtin_amdgcn_s_memrealtime() - t0 > ticks) {
            if (threadIdx.x == 0) __hip_atomic_fetch_add(timeouts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        __builtin_amdgcn_s_sleep(4);
    }
}

}  // namespace

// GH_SHARD_STEP_FORM in the environment forces a form like gh_comm_set_step_form does (-1: not set)
static int env_step_form()
{
    static const int v = [] {
        const char *e = getenv("GH_SHARD_STEP_FORM");
        if (!e || !*e) return -1;
        const int f = atoi(e);
        return (f == 1 || f == 2 || (GH_LAB && f == 0)) ? f : -1;
    }();
    return v;
}
// the form of the next step where the frame allows the halo-flag form: forced (API, then environment), else 2 until the review says 1
static int step_form_of(const Comm *c)
{
    if (c->step_form >= 0) return c->step_form;
    if (env_step_form() >= 0) return env_step_form();
    return c->auto_serial ? 1 : 2;
}
static bool form_is_forced(const Comm *c) { return c->step_form >= 0 || env_step_form() >= 0; }

// One row to and from each neighbour, nothing beside it: the warm-up of the peer connections and -- timed over a few repetitions -- what the
// serial form of a step pays on top of its kernel.  Collective (every rank of the communicator is in gh_comm_init at the same time).
static int time_exchange_alone(Comm *c, int up, int down)
{
    uint8_t *w = reinterpret_cast<uint8_t *>(c->flag_dev) + 256;
    if (int rc = gh_halo_exchange_dev(c, up, w, w + 64, down, w + 128, w + 192, 64, c->side)) return rc;       // connects the peers
    GH_HIP_TRY(hipStreamSynchronize(c->side));
    hipEvent_t a = nullptr, b = nullptr;
    GH_HIP_TRY(hipEventCreate(&a));
    GH_HIP_TRY(hipEventCreate(&b));
    constexpr int kReps = 8;
    int rc = GH_OK;
    hipError_t e = hipEventRecord(a, c->side);
    for (int i = 0; i < kReps && rc == GH_OK && e == hipSuccess; ++i) rc = gh_halo_exchange_dev(c, up, w, w + 64, down, w + 128, w + 192, 64, c->side);
    if (rc == GH_OK && e == hipSuccess) e = hipEventRecord(b, c->side);
    if (rc == GH_OK && e == hipSuccess) e = hipEventSynchronize(b);
    float ms = 0.f;
    if (rc == GH_OK && e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    if (rc != GH_OK) return rc;
    if (e != hipSuccess) return gh_set_error(GH_ERR_HIP, "timing the halo exchange failed: %s", hipGetErrorString(e));
    c->exchange_alone_us = ms * 1e3f / kReps;
    c->exchanges = 0;
    return GH_OK;
}

extern "C" {

int gh_comm_unique_id(uint8_t id[GH_COMM_ID_BYTES])
{
    GH_REQUIRE(id, "null id");
    static_assert(GH_COMM_ID_BYTES == kIdBytes, "ncclUniqueId is 128 bytes");
    if (int rc = need_rccl()) return rc;
    NcclId nid;
    GH_NCCL_TRY(g_rccl.GetUniqueId(&nid));
    memcpy(id, nid.internal, kIdBytes);
    return GH_OK;
}

int gh_comm_init(const uint8_t id[GH_COMM_ID_BYTES], int rank, int world, gh_comm_t *out)
{
    GH_REQUIRE(id && out, "null pointer");
    GH_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad rank / world");
    if (int rc = gh_ensure_init()) return rc;
    if (int rc = need_rccl()) return rc;
    Comm *c = new Comm;
    c->rank = rank;
    c->world = world;
    c->epoch = c->xepoch = (uint32_t)GH_LAB_INT("GH_HALO_EPOCH_START", 0);      // (probe builds: tests start near the wrap guard of the halo-flag form)
    NcclId nid;
    memcpy(nid.internal, id, kIdBytes);
    const int r = g_rccl.CommInitRank(&c->nccl, world, nid, rank);
    if (r != 0) {
        delete c;
        return gh_set_error(GH_ERR_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, g_rccl.GetErrorString(r));
    }
    // high priority: its own hardware queue (a plain second stream may be mapped onto the caller's queue, which would
    // serialise the exchange with the shard kernels), and the small exchange kernel is not stuck behind a full grid
    const int rc = [&]() -> int {
        int prio_least = 0, prio_greatest = 0;
        GH_HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
        GH_HIP_TRY(hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, prio_greatest));
        c->side_prio = prio_greatest;
        GH_HIP_TRY(hipEventCreateWithFlags(&c->ready, hipEventDisableTiming));
        GH_HIP_TRY(hipEventCreateWithFlags(&c->landed, hipEventDisableTiming));
        for (int i = 0; i < Comm::kPrefetch; ++i) GH_HIP_TRY(hipEventCreateWithFlags(&c->pf_landed[i], hipEventDisableTiming));
        GH_HIP_TRY(hipMalloc((void **)&c->flag_dev, kFlagBytes));
        GH_HIP_TRY(hipMemset(c->flag_dev, 0, kFlagBytes));
        int can_wait = 0, dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, dev) == hipSuccess && can_wait &&
            hipExtMallocWithFlags((void **)&c->started_sig, 8, hipMallocSignalMemory) == hipSuccess) {
            if (hipMemset(c->started_sig, 0, 8) != hipSuccess) { (void)hipFree(c->started_sig); c->started_sig = nullptr; }
        } else {
            c->started_sig = nullptr;
        }
        (void)hipGetLastError();
        GH_HIP_TRY(hipHostMalloc((void **)&c->timeouts_host, 64, hipHostMallocMapped));
        memset(c->timeouts_host, 0, 64);
        GH_HIP_TRY(hipHostGetDevicePointer((void **)&c->timeouts_dev, c->timeouts_host, 0));
        // Peer connections now, not inside the first step: RCCL connects to a peer lazily, inside the first ncclGroupEnd that names it, and that
        // can take seconds and block the host.  The halo-flag form queues a wait for `started` on the side stream BEFORE its exchange and bounds
        // every device-side wait (GH_HALO_WAIT_MS), so a first exchange that connected under such a wait could be taken for a lost one.  One
        // 64-byte row to and from each neighbour here (gh_comm_init is collective: every rank is in this call), then the connections exist.
        if (world > 1)
            if (int rc2 = time_exchange_alone(c, rank > 0 ? rank - 1 : -1, rank + 1 < world ? rank + 1 : -1)) return rc2;
        return GH_OK;
    }();
    if (rc != GH_OK) {
        gh_comm_destroy(c);       // releases whatever was created (the error text of the failed call is kept)
        return rc;
    }
    *out = c;
    return GH_OK;
}

int gh_comm_rank(gh_comm_t comm) { return comm ? static_cast<Comm *>(comm)->rank : -1; }
int gh_comm_world(gh_comm_t comm) { return comm ? static_cast<Comm *>(comm)->world : -1; }

int gh_comm_set_loopback(gh_comm_t comm, int mode)
{
    GH_REQUIRE(comm, "null communicator");
    GH_REQUIRE(mode >= 0 && mode <= 3, "mode must be 0 (off), 1 (middle shard), 2 (top shard) or 3 (bottom shard)");
    Comm *c = static_cast<Comm *>(comm);
    c->loopback = mode;
    forget_announced(c);
    if (mode) {                       // the rehearsal's own figure of an exchange alone (ncclSend / ncclRecv to self)
        int up, down;
        neighbours(c, up, down);
        if (int rc = time_exchange_alone(c, up, down)) return rc;
    }
    return GH_OK;
}

// Which form the single-frame sharded steps of this communicator take: 2 halo flag, 1 serial, 0 events (see sharded_step below);
// -1 = back to GH_SHARD_STEP_FORM / the default.  Every rank must choose alike only in the sense that each rank's own steps are
// self-contained: the forms differ in how a rank orders ITS exchange and ITS kernel, the ncclSend / ncclRecv pairs are the same.
int gh_comm_set_step_form(gh_comm_t comm, int form)
{
    GH_REQUIRE(comm, "null communicator");
    GH_REQUIRE(form >= -1 && form <= 2, "form must be -1, 1 or 2 (0 in probe builds)");
    GH_REQUIRE(GH_LAB || form != 0, "form 0 (events) lost its A/B and exists in probe builds only (make probes)");
    Comm *c = static_cast<Comm *>(comm);
    c->step_form = form;
    c->auto_serial = false;            // (a forced form is the caller's decision; -1 starts the review afresh)
    c->reviewed = false;
    c->review_epoch = c->epoch;
    if (c->timeouts_host) {
        c->seen_ticks = __atomic_load_n(c->timeouts_host + 2, __ATOMIC_ACQUIRE);
        c->seen_waits = __atomic_load_n(c->timeouts_host + 4, __ATOMIC_ACQUIRE);
    }
    forget_announced(c);
    if (c->timeouts_host) __atomic_store_n(c->timeouts_host, 0u, __ATOMIC_RELEASE);      // choosing a form acknowledges waits that gave up earlier
    return GH_OK;
}

// Bound of the device-side waits of the halo-flag form on this communicator, in milliseconds (-1: GH_HALO_WAIT_MS or the default, 10 s).
int gh_comm_set_halo_wait_ms(gh_comm_t comm, int ms)
{
    GH_REQUIRE(comm, "null communicator");
    GH_REQUIRE(ms >= -1, "ms must be >= 0, or -1 for the default");
    static_cast<Comm *>(comm)->halo_wait_ms = ms;
    return GH_OK;
}

int gh_comm_destroy(gh_comm_t comm)
{
    if (!comm) return GH_OK;
    Comm *c = static_cast<Comm *>(comm);
    if (c->side) (void)hipStreamSynchronize(c->side);
    if (c->nccl) (void)g_rccl.CommDestroy(c->nccl);
    if (c->ready) (void)hipEventDestroy(c->ready);
    if (c->landed) (void)hipEventDestroy(c->landed);
    for (int i = 0; i < Comm::kPrefetch; ++i)
        if (c->pf_landed[i]) (void)hipEventDestroy(c->pf_landed[i]);
    if (c->side) (void)hipStreamDestroy(c->side);
    for (hipStream_t rs : c->retired_side)
        if (rs) { (void)hipStreamSynchronize(rs); (void)hipStreamDestroy(rs); }
    if (c->flag_dev) (void)hipFree(c->flag_dev);
    if (c->started_sig) (void)hipFree(c->started_sig);
    if (c->timeouts_host) (void)hipHostFree(c->timeouts_host);
    delete c;
    return GH_OK;
}

// One ncclGroup: my `send_up` row goes to up_peer and its row arrives in recv_up; likewise downwards.  A peer < 0 means
// "no neighbour on that side" (frame border).  Everything is enqueued on `stream`.
int gh_halo_exchange_dev(gh_comm_t comm, int up_peer, const void *send_up, void *recv_up, int down_peer, const void *send_down,
                         void *recv_down, size_t bytes, gh_stream_t stream)
{
    GH_REQUIRE(comm, "null communicator");
    Comm *c = static_cast<Comm *>(comm);
    GH_REQUIRE(up_peer < c->world && down_peer < c->world, "peer out of range");
    forget_announced(c);       // (the stream step re-announces its `next` after its own exchanges)
    if (bytes == 0 || (up_peer < 0 && down_peer < 0)) return GH_OK;
    GH_REQUIRE((up_peer < 0 || (send_up && recv_up)) && (down_peer < 0 || (send_down && recv_down)), "null halo row");
    hipStream_t s = (hipStream_t)stream;
    GH_NCCL_TRY(g_rccl.GroupStart());
    int r = 0;
    if (up_peer >= 0) {
        r |= g_rccl.Send(send_up, bytes, kNcclUint8, up_peer, c->nccl, s);
        r |= g_rccl.Recv(recv_up, bytes, kNcclUint8, up_peer, c->nccl, s);
    }
    if (down_peer >= 0) {
        r |= g_rccl.Send(send_down, bytes, kNcclUint8, down_peer, c->nccl, s);
        r |= g_rccl.Recv(recv_down, bytes, kNcclUint8, down_peer, c->nccl, s);
    }
    const int e = g_rccl.GroupEnd();
    if (r != 0 || e != 0) return gh_set_error(GH_ERR_HIP, "halo exchange failed: %s", g_rccl.GetErrorString(r ? r : e));
    ++c->exchanges;
    return GH_OK;
}

unsigned long long gh_comm_exchange_count(gh_comm_t comm) { return comm ? static_cast<Comm *>(comm)->exchanges : 0; }

#ifdef GH_HALO_TRACE
// trace builds only (make probes): the handshake's 100 MHz timestamps of the LAST step (synchronises the device)
extern "C" __attribute__((visibility("default"))) int gh_comm_halo_trace(gh_comm_t comm, unsigned long long out[8])
{
    GH_REQUIRE(comm && out, "null pointer");
    GH_HIP_TRY(hipDeviceSynchronize());
    GH_HIP_TRY(hipMemcpy(out, static_cast<Comm *>(comm)->flag_dev + 16, 64, hipMemcpyDeviceToHost));
    return GH_OK;
}
#endif

// halo-flag waits that gave up (gh_common.h): 0 unless an exchange took longer than GH_HALO_WAIT_MS beside a launch that was waiting for it
unsigned int gh_comm_halo_timeouts(gh_comm_t comm)
{
    if (!comm) return 0;
    const Comm *c = static_cast<const Comm *>(comm);
    return c->timeouts_host ? __atomic_load_n(c->timeouts_host, __ATOMIC_ACQUIRE) : 0;
}

// Form the next single-frame step of this communicator would take where the frame allows the halo-flag form: 2 or 1 (0 in probe builds) --
// the forced form, else GH_SHARD_STEP_FORM / 2, else 1 once the review found the exchange not overlapping (halo_flag_review).
int gh_comm_step_form_in_use(gh_comm_t comm)
{
    if (!comm) return -1;
    const Comm *c = static_cast<const Comm *>(comm);
    return step_form_of(c);
}

// side streams the review has replaced so far (0 in a healthy process; at most 2, then the serial form)
int gh_comm_side_stream_replacements(gh_comm_t comm) { return comm ? static_cast<const Comm *>(comm)->side_replacements : 0; }

// One halo exchange with nothing beside it, microseconds: timed over eight repetitions when the communicator met its neighbours
// (gh_comm_init; gh_comm_set_loopback for the rehearsal) -- what the serial form of a step pays on top of its kernel.
double gh_comm_exchange_alone_us(gh_comm_t comm) { return comm ? (double)static_cast<const Comm *>(comm)->exchange_alone_us : 0.0; }

// What the edge bands of the halo-flag launches have reported so far: how many of their waits had to stand still and for how many
// microseconds in all (100 MHz clock; the sum wraps after 42 s of standing still).  Diagnostic; reads host-mapped memory.
int gh_comm_halo_wait_stats(gh_comm_t comm, unsigned int *waits, double *microseconds)
{
    GH_REQUIRE(comm, "null communicator");
    const Comm *c = static_cast<const Comm *>(comm);
    if (waits) *waits = c->timeouts_host ? __atomic_load_n(c->timeouts_host + 4, __ATOMIC_ACQUIRE) : 0;
    if (microseconds) *microseconds = c->timeouts_host ? __atomic_load_n(c->timeouts_host + 2, __ATOMIC_ACQUIRE) / 100.0 : 0.0;
    return GH_OK;
}

// After the synchronisation that hands a sharded step's results to the host: did every halo-flag wait of the steps so far see its exchange?
// (The count lives in host-mapped memory: no device call here.)
int gh_comm_check(gh_comm_t comm)
{
    GH_REQUIRE(comm, "null communicator");
    const unsigned int n = gh_comm_halo_timeouts(comm);
    if (n)
        return gh_set_error(GH_ERR_HIP, "%u halo-flag waits gave up after GH_HALO_WAIT_MS: a neighbour's rows did not arrive in time and the edge rows of "
                                        "that sharded step are invalid (gh_comm_set_step_form acknowledges; form 1 waits like plain RCCL)", n);
    return GH_OK;
}

int gh_comm_allreduce_sum_u64_dev(gh_comm_t comm, uint64_t *buf_dev, size_t count, gh_stream_t stream)
{
    GH_REQUIRE(comm && buf_dev, "null pointer");
    Comm *c = static_cast<Comm *>(comm);
    GH_NCCL_TRY(g_rccl.AllReduce(buf_dev, buf_dev, count, kNcclUint64, kNcclSum, c->nccl, (hipStream_t)stream));
    return GH_OK;
}

// min and max of the mean-grey levels over all ranks (old/parallel_to_grayscale.c:12 `reduction(min:min_gray) reduction(max:max_gray)`,
// :36-37): minmax_dev[0] <- min over ranks, minmax_dev[1] <- max over ranks, both in one group.  A rank without rows contributes the
// reference's initial values 256 / -1, the neutral elements.
int gh_comm_allreduce_minmax_i32_dev(gh_comm_t comm, int32_t *minmax_dev, gh_stream_t stream)
{
    GH_REQUIRE(comm && minmax_dev, "null pointer");
    Comm *c = static_cast<Comm *>(comm);
    hipStream_t s = (hipStream_t)stream;
    GH_NCCL_TRY(g_rccl.GroupStart());
    const int r0 = g_rccl.AllReduce(minmax_dev, minmax_dev, 1, kNcclInt32, kNcclMin, c->nccl, s);
    const int r1 = g_rccl.AllReduce(minmax_dev + 1, minmax_dev + 1, 1, kNcclInt32, kNcclMax, c->nccl, s);
    const int e = g_rccl.GroupEnd();
    if (r0 != 0 || r1 != 0 || e != 0) return gh_set_error(GH_ERR_HIP, "min/max all-reduce failed: %s", g_rccl.GetErrorString(r0 ? r0 : (r1 ? r1 : e)));
    return GH_OK;
}

}  // extern "C"

// ---- one pipeline step of a row shard ------------------------------------------------------------------------------
// buf: (rows + 2) rows of `row_bytes`; rows 1..rows are the shard, rows 0 / rows+1 receive the neighbours' edge rows.
// Two forms: serial (exchange, then one launch) and overlapped (exchange on the side stream || interior rows on the caller's
// stream -> first and last row once the halo has landed).
// ---- the halo-flag machinery shared by the single-frame step and the stream-of-frames step (gh_common.h) --------------------------
// The epochs are compared as 32-bit numbers (the command processor's wait: unsigned): long before they could wrap -- 2^31 steps, about a
// day of back-to-back 40 us steps -- both streams are drained once and the words start again from zero.
static int halo_flag_guard(Comm *c, hipStream_t s)
{
    if (c->epoch < 0x7ffffff0u && c->xepoch < 0x7ffffff0u) return GH_OK;
    GH_HIP_TRY(hipStreamSynchronize(s));
    GH_HIP_TRY(hipStreamSynchronize(c->side));
    GH_HIP_TRY(hipMemset(c->flag_dev, 0, kFlagBytes));
    if (c->started_sig) GH_HIP_TRY(hipMemset(c->started_sig, 0, 8));
    c->epoch = c->xepoch = 0;
    c->stream_buf = nullptr;
    c->seen_ticks = c->seen_waits = c->review_epoch = 0;      // (the standstill counters live in flag_dev and were cleared with it)
    c->reviewed = false;
    if (c->timeouts_host) { c->timeouts_host[2] = 0; c->timeouts_host[4] = 0; }
    return GH_OK;
}

// Does the halo-flag form do what it is for?  Its edge bands report how long they stood still in front of their halo row (gh_halo_wait).
// A few microseconds are normal for a small shard (the exchange beside a bandwidth-bound launch takes about as long as a 2048-row kernel);
// tens of them mean the exchange is NOT running beside the launch -- seen when the caller's stream and the side stream land on hardware
// queues of the same command-processor pipe: `started` is then noticed ~30 us late and the flag kernel starts ~40 us behind the RCCL kernel,
// 73 us per step where the serial form takes 40 and the healthy halo-flag form 33 (NOTES.md R4.4).  Which queues a process's streams get
// is the runtime's business, so the communicator looks at the reports every 32 launches; once the average standstill exceeds what the serial
// form would pay instead -- one exchange with nothing beside it, timed at gh_comm_init: 1.25 x that + 2 us (healthy loopback: 4.5 us against
// 16, the slow mapping 19-45 us) -- it first swaps the side stream for a fresh one, which gets another hardware queue (measured: 32 us per
// step again after ONE swap), and after two swaps that did not help goes back to the serial form for good unless the caller forced a form.  gh_comm_step_form_in_use() / gh_comm_halo_wait_stats() say so.
static void halo_flag_review(Comm *c)
{
    if (c->auto_serial || !c->timeouts_host || c->epoch - c->review_epoch < 32) return;
    const bool first_look = !c->reviewed;                 // the first 32 launches are run-in (RCCL's set-up, cold caches): noted, not judged
    c->reviewed = true;
    c->review_epoch = c->epoch;
    const uint32_t ticks = __atomic_load_n(c->timeouts_host + 2, __ATOMIC_ACQUIRE), waits = __atomic_load_n(c->timeouts_host + 4, __ATOMIC_ACQUIRE);
    const uint32_t dt = ticks - c->seen_ticks, dn = waits - c->seen_waits;
    c->seen_ticks = ticks;
    c->seen_waits = waits;
    if (first_look || dn < 8) return;                     // (hardly any band had to wait: nothing to judge)
    c->last_avg_wait_ticks = dt / dn;
    // the halo-flag form costs a step its edge bands' standstill, the serial form one exchange with nothing beside it (timed at init)
    if (c->last_avg_wait_ticks <= (uint32_t)((c->exchange_alone_us * 1.25f + 2.f) * 100.f)) return;
    // Before giving the form up: ANOTHER side stream.  The trouble is the hardware queue the side stream was given (it shares a pipe with the
    // caller's); a second high-priority stream gets another queue.  Up to two replacements, each judged by the next windows; then serial.
    if (c->side_replacements < 2) {
        hipStream_t fresh = nullptr;
        if (hipStreamSynchronize(c->side) == hipSuccess && hipStreamCreateWithPriority(&fresh, hipStreamNonBlocking, c->side_prio) == hipSuccess) {
            c->retired_side[c->side_replacements++] = c->side;      // (destroyed with the communicator: RCCL may still hold per-stream state)
            c->side = fresh;
            c->reviewed = false;                                      // the new stream's first window is run-in again (RCCL sets it up: milliseconds)
            return;
        }
        (void)hipGetLastError();
    }
    if (!form_is_forced(c)) c->auto_serial = true;
}

// The halo-flag form needs the caller's stream and the side stream on DIFFERENT hardware queues: the side stream's wait for `started` would
// otherwise sit in front of the very launch that stores it.  The side stream has the greatest priority, which gives it a queue of its own
// beside normal-priority streams; a caller that passes a stream of that same priority may share its queue -- such steps take the serial form.
static bool caller_stream_ok(Comm *c, hipStream_t s)
{
    if (c->prio_seen && c->prio_stream == s) return c->prio_ok;
    int prio = 0;
    const bool known = s != nullptr && hipStreamGetPriority(s, &prio) == hipSuccess;       // (the NULL stream has the default priority)
    (void)hipGetLastError();
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    c->prio_stream = s;
    c->prio_seen = true;
    c->prio_ok = !(known && prio == c->side_prio && least != greatest);
    return c->prio_ok;
}

// the launch's side of the handshake: a new `started` number; wait_epoch is filled in by whoever knows which exchange the launch needs
static GhHaloWait halo_flag_launch_args(Comm *c)
{
    // The two side-stream operations are one-wave kernels whose waits are BOUNDED (GH_HALO_WAIT_MS): whatever happens to the launch or the
    // exchange, every wave on either stream reaches its exit.  GH_HALO_MEMOPS (probe builds; bit 1: the wait, bit 0: the flag write) hands them
    // to the command processor instead (hipStreamWaitValue32 on signal memory / hipStreamWriteValue32).  Measured (tools/sweep.py
    // shard_memops, 2048-row shard): kernels 38.4 us per step, the wait as a memory operation 37.6, the write as one 42.0 -- but a
    // command-processor wait has no time-out: if the launch it waits for never starts, the side stream is stuck for good.  0.8 us per step
    // does not pay for that, so the kernels are what ships.
    const bool sig_wait = (GH_LAB_INT("GH_HALO_MEMOPS", 0) & 2) && c->started_sig;
    GhHaloWait hw{};
    hw.flag = c->flag_dev;
    hw.started = sig_wait ? c->started_sig : c->flag_dev + 32;
    hw.timeouts = c->timeouts_dev;
    hw.epoch = ++c->epoch;
    hw.ticks = (unsigned long long)(c->halo_wait_ms >= 0 ? c->halo_wait_ms : GH_KNOB_INT("GH_HALO_WAIT_MS", 10000)) * 100000ull;       // 100 MHz constant clock
    return hw;
}

// the side stream's part: behind `started` >= hw.epoch (the launch that carries hw has begun) the halo rows of `buf` are exchanged and
// `flag` takes a new number, returned in *flag_number
static int halo_flag_exchange(Comm *c, const GhHaloWait &hw, uint8_t *buf, size_t row_bytes, int rows, int up, int down, uint32_t *flag_number)
{
    if (hw.started == c->started_sig) {
        GH_HIP_TRY(hipStreamWaitValue32(c->side, c->started_sig, hw.epoch, hipStreamWaitValueGte, 0xffffffffu));
    } else {
        k_wait_started<<<1, 64, 0, c->side>>>(hw.started, hw.epoch, hw.ticks, hw.timeouts, c->flag_dev);
        GH_LAUNCH_CHECK();
    }
    if (int rc = gh_halo_exchange_dev(c, up, buf + row_bytes, buf, down, buf + (size_t)rows * row_bytes, buf + (size_t)(rows + 1) * row_bytes, row_bytes, c->side))
        return rc;
    const uint32_t f = ++c->xepoch;
    if (GH_LAB_INT("GH_HALO_MEMOPS", 0) & 1) {
        GH_HIP_TRY(hipStreamWriteValue32(c->side, c->flag_dev, f, 0));
    } else {
        k_set_halo_flag<<<1, 64, 0, c->side>>>(c->flag_dev, f, c->timeouts_dev);
        GH_LAUNCH_CHECK();
    }
    *flag_number = f;
    return GH_OK;
}

// the shard kernel was not launched: nobody will ever store `started`, and the side stream (a command-processor wait has no time-out)
// would sit in front of the exchange for good -- release it from the caller's stream
static void halo_flag_release(hipStream_t s, const GhHaloWait &hw)
{
    k_store_u32<<<1, 64, 0, s>>>(hw.started, hw.epoch);
    (void)hipGetLastError();
}

template <typename Launch, typename LaunchWait>
static int sharded_step(Comm *c, uint8_t *buf, size_t row_bytes, int rows, hipStream_t s, Launch launch /* (first_row, nrows, top_border, bot_border) */,
                        LaunchWait launch_wait /* (top_border, bot_border, const GhHaloWait &): the whole shard, edge bands wait for the flag */)
{
    int up, down;
    neighbours(c, up, down);
    forget_announced(c);
    if (up < 0 && down < 0) return launch(0, rows, 1, 1);
    uint8_t *send_up = buf + row_bytes, *recv_up = buf, *send_down = buf + (size_t)rows * row_bytes, *recv_down = buf + (size_t)(rows + 1) * row_bytes;
    // Three forms (GH_SHARD_STEP_FORM / gh_comm_set_step_form; loopback rehearsal, 16384-wide shards of 2048 / 4096 / 8192 rows, us per step;
    // tools/sweep.py shard_forms):
    //  2 halo flag (default where the shard runs an LDS-DMA ring kernel): the exchange on the side stream BESIDE one launch over the whole
    //    shard; the two sides meet through two polled dwords (`started`, `flag`: gh_common.h), only the first / last band's workgroups wait --
    //    inside the kernel, in front of their halo row's request.  No event, no extra launch on the caller's stream: 33.7 / 52.7 / 97.4
    //    (the launch alone: 29.3 / 50.4 / 96.0).
    //  1 serial: the exchange on the caller's stream in front of ONE launch -- no side stream, no events; the exchange (one row, ~11 us)
    //    is not hidden: 40.0 / 62.7 / 110.7 (round 2's default).
    //  0 overlapped with events (probe builds only): exchange on the side stream || interior rows, then two one-row launches behind an event --
    //    pays two cross-queue event hand-overs (~20 us each on this runtime) and two extra launches: 58.2 / 66.9 / 116.7.
    halo_flag_review(c);
    int form = step_form_of(c);
    if (form == 2) {                               // frames whose kernel form carries no flag wait (narrow, per-pixel, convolution): serial form
        GhHaloWait probe{};
        probe.pad = 1;
        if (launch_wait(up < 0, down < 0, probe) != GH_OK || !caller_stream_ok(c, s)) form = 1;
    }
    if (form == 1) {
        if (int rc = gh_halo_exchange_dev(c, up, send_up, recv_up, down, send_down, recv_down, row_bytes, s)) return rc;
        return launch(0, rows, up < 0, down < 0);
    }
    if (form == 2) {
        // (a wait of an earlier halo-flag step that gave up -- its edge rows were computed from stale halo rows -- fails every later
        //  halo-flag step of this communicator; gh_comm_set_step_form acknowledges it, e.g. on the way to the serial form)
        if (gh_comm_halo_timeouts(c))
            return gh_set_error(GH_ERR_HIP, "an earlier sharded step gave up waiting for its halo rows (%u waits timed out): results of that step are invalid",
                                gh_comm_halo_timeouts(c));
        // No event in either direction: the side stream polls `started` (stored by the shard kernel's first workgroup: everything
        // enqueued on s before this step has completed by then -- the rows to send are in place, the previous step has read its halo
        // rows), exchanges, and sets `flag`, which the shard kernel's edge bands poll.
        if (int rc = halo_flag_guard(c, s)) return rc;
        GhHaloWait hw = halo_flag_launch_args(c);
        if (int rc = halo_flag_exchange(c, hw, buf, row_bytes, rows, up, down, &hw.wait_epoch)) { halo_flag_release(s, hw); return rc; }
        // Later work on s is ordered behind this launch, which cannot finish before the flag is set, i.e. before the exchange has
        // completed: the caller may overwrite the shard's rows right after the step, as with the other forms.
        const int rc = launch_wait(up < 0, down < 0, hw);
        if (rc != GH_OK) halo_flag_release(s, hw);
        return rc;
    }
#ifdef GH_ENABLE_PROBES
    // the exchange reads rows 1 / rows (written by earlier work on s) and overwrites the halo rows the previous step read
    GH_HIP_TRY(hipEventRecord(c->ready, s));
    GH_HIP_TRY(hipStreamWaitEvent(c->side, c->ready, 0));
    if (int rc = gh_halo_exchange_dev(c, up, send_up, recv_up, down, send_down, recv_down, row_bytes, c->side)) return rc;
    GH_HIP_TRY(hipEventRecord(c->landed, c->side));
    if (form == 0 && rows >= 3) {
        if (int rc = launch(1, rows - 2, 0, 0)) return rc;                       // interior: local rows only
        GH_HIP_TRY(hipStreamWaitEvent(s, c->landed, 0));
        if (int rc = launch(0, 1, up < 0, 0)) return rc;
        return launch(rows - 1, 1, 0, down < 0);
    }
    GH_HIP_TRY(hipStreamWaitEvent(s, c->landed, 0));
    return launch(0, rows, up < 0, down < 0);
#else
    return gh_set_error(GH_ERR_ARG, "GH_SHARD_STEP_FORM=%d: this library has forms 2 (halo flag) and 1 (serial)", form);
#endif
}

// Pipelined form for a stream of frames held in a few buffers: the halo rows of `prefetch` (a frame that will be `cur`
// in a later call) are exchanged on the side stream WHILE `cur` -- whose halo rows were requested by an earlier call --
// is processed by one launch over the whole shard.  Per step the caller's stream sees a single kernel; the exchange
// latency, the cross-queue event latency (~20 us on this runtime when the wait is reached before the signal) and the
// two one-row launches of the form above leave the critical path once the halo is requested two frames ahead.  If
// cur's halo has not been requested (first calls, or the caller changed buffers), it is exchanged first.
static int pf_find(const Comm *c, const void *p)
{
    for (int i = 0; i < Comm::kEntries; ++i)
        if (c->pf_buf[i] == p) return i;
    return -1;
}

// Asks the side stream to fill the halo rows of n frame buffers in ONE ncclGroup (per buffer: my first row up, my last row down,
// the neighbours' rows into rows 0 and rows + 1) and records one event behind it.  Reads the buffers' rows 1 / rows (written by
// earlier work on s) and overwrites their halo rows (last read by earlier work on s): the side stream first waits for an event
// recorded on s here.  Returns the table entry of the FIRST buffer (>= 0) or a negative status.
static int prefetch_halos(Comm *c, uint8_t *const *bufs, int n, size_t row_bytes, int rows, hipStream_t s, int up, int down)
{
    int ev = -1;
    for (int t = 0; t < Comm::kPrefetch && ev < 0; ++t) {            // next event none of whose requests is still unconsumed
        const int cand = (c->pf_next + t) % Comm::kPrefetch;
        bool live = false;
        for (int i = 0; i < Comm::kEntries && !live; ++i) live = c->pf_buf[i] && c->pf_event_of[i] == cand;
        if (!live) ev = cand;
    }
    if (ev < 0) {                                                     // every event still owns requests nobody came for: the oldest goes
        ev = c->pf_next;
        for (int i = 0; i < Comm::kEntries; ++i)
            if (c->pf_buf[i] && c->pf_event_of[i] == ev) c->pf_buf[i] = nullptr;
    }
    c->pf_next = (ev + 1) % Comm::kPrefetch;
    c->pf_waited[ev] = false;
    if (hipEventRecord(c->ready, s) != hipSuccess || hipStreamWaitEvent(c->side, c->ready, 0) != hipSuccess)
        return -gh_set_error(GH_ERR_HIP, "event hand-over to the exchange stream failed");
    int r = g_rccl.GroupStart();
    if (r != 0) return -gh_set_error(GH_ERR_HIP, "ncclGroupStart failed: %s", g_rccl.GetErrorString(r));
    for (int k = 0; k < n; ++k) {
        uint8_t *buf = bufs[k];
        if (up >= 0) {
            r |= g_rccl.Send(buf + row_bytes, row_bytes, kNcclUint8, up, c->nccl, c->side);
            r |= g_rccl.Recv(buf, row_bytes, kNcclUint8, up, c->nccl, c->side);
        }
        if (down >= 0) {
            r |= g_rccl.Send(buf + (size_t)rows * row_bytes, row_bytes, kNcclUint8, down, c->nccl, c->side);
            r |= g_rccl.Recv(buf + (size_t)(rows + 1) * row_bytes, row_bytes, kNcclUint8, down, c->nccl, c->side);
        }
    }
    const int e = g_rccl.GroupEnd();
    if (r != 0 || e != 0) return -gh_set_error(GH_ERR_HIP, "halo exchange failed: %s", g_rccl.GetErrorString(r ? r : e));
    ++c->exchanges;
    if (hipEventRecord(c->pf_landed[ev], c->side) != hipSuccess) return -gh_set_error(GH_ERR_HIP, "hipEventRecord failed");
    int first = -1;
    for (int k = 0; k < n; ++k) {
        int slot = -1;
        for (int t = 0; t < Comm::kEntries && slot < 0; ++t) {        // next free entry, round robin
            const int i = (c->pf_next_entry + t) % Comm::kEntries;
            if (!c->pf_buf[i]) slot = i;
        }
        if (slot < 0) { slot = c->pf_next_entry; }                    // table full of stale requests: overwrite the oldest position
        c->pf_next_entry = (slot + 1) % Comm::kEntries;
        c->pf_buf[slot] = bufs[k];
        c->pf_event_of[slot] = ev;
        if (k == 0) first = slot;
    }
    return first;
}

template <typename Launch>
static int sharded_step_pipelined(Comm *c, uint8_t *cur, uint8_t *prefetch, size_t row_bytes, int rows, hipStream_t s, Launch launch)
{
    int up, down;
    neighbours(c, up, down);
    if (up < 0 && down < 0) return launch(cur, 1, 1);
    int entry = pf_find(c, cur);
    if (entry < 0) {
        uint8_t *one[1] = {cur};
        entry = prefetch_halos(c, one, 1, row_bytes, rows, s, up, down);
        if (entry < 0) return -entry;
    }
    const int ev = c->pf_event_of[entry];
    if (!c->pf_waited[ev]) {                          // one wait per batch: the frames that share the event find it done
        GH_HIP_TRY(hipStreamWaitEvent(s, c->pf_landed[ev], 0));   // cur's halo rows have landed
        c->pf_waited[ev] = true;
    }
    c->pf_buf[entry] = nullptr;
    if (prefetch && prefetch != cur && pf_find(c, prefetch) < 0) {
        // `ready` is recorded BEFORE cur's kernel is launched, so the exchange runs beside it
        uint8_t *one[1] = {prefetch};
        const int ps = prefetch_halos(c, one, 1, row_bytes, rows, s, up, down);
        if (ps < 0) return -ps;
    }
    return launch(cur, up < 0, down < 0);
}

extern "C" {

int gh_sharded_fused_step_pipelined_dev(gh_comm_t comm, uint8_t *cur_rgb_buf, uint8_t *prefetch_rgb_buf, uint8_t *edge, int width, int rows,
                                        int channels, int mode, gh_stream_t stream)
{
    GH_REQUIRE(comm, "null communicator");
    GH_REQUIRE(width >= 0 && rows >= 0, "negative size");
    if (width == 0 || rows == 0) return GH_OK;
    GH_REQUIRE(cur_rgb_buf && edge, "null pointer");
    Comm *c = static_cast<Comm *>(comm);
    const size_t rb = (size_t)width * channels;
    return sharded_step_pipelined(c, cur_rgb_buf, prefetch_rgb_buf, rb, rows, (hipStream_t)stream, [&](uint8_t *buf, int tb, int bb) {
        return gh_gray_sobel_fused_rows_dev(buf + rb, edge, width, rows, channels, mode, tb, bb, stream);
    });
}

int gh_sharded_prefetch_reset(gh_comm_t comm)
{
    GH_REQUIRE(comm, "null communicator");
    Comm *c = static_cast<Comm *>(comm);
    for (int i = 0; i < Comm::kEntries; ++i) c->pf_buf[i] = nullptr;
    c->stream_buf = nullptr;
    return GH_OK;
}

int gh_sharded_prefetch_halos_dev(gh_comm_t comm, uint8_t *const *bufs, int n, size_t row_bytes, int rows, gh_stream_t stream)
{
    GH_REQUIRE(comm && bufs, "null pointer");
    GH_REQUIRE(n >= 1 && n <= Comm::kEntries / 2, "between 1 and 16 buffers per batch");
    GH_REQUIRE(rows >= 1 && row_bytes >= 1, "empty shard");
    for (int k = 0; k < n; ++k) GH_REQUIRE(bufs[k], "null buffer");
    Comm *c = static_cast<Comm *>(comm);
    int up, down;
    neighbours(c, up, down);
    if (up < 0 && down < 0) return GH_OK;             // a single rank: nothing to exchange
    for (int k = 0; k < n; ++k) {                      // a buffer asked for again: its old request is forgotten
        const int i = pf_find(c, bufs[k]);
        if (i >= 0) c->pf_buf[i] = nullptr;
    }
    const int first = prefetch_halos(c, bufs, n, row_bytes, rows, (hipStream_t)stream, up, down);
    return first < 0 ? -first : GH_OK;
}

int gh_sharded_sobel_step_pipelined_dev(gh_comm_t comm, uint8_t *cur_luma_buf, uint8_t *prefetch_luma_buf, uint8_t *edge, int width, int rows,
                                        gh_stream_t stream)
{
    GH_REQUIRE(comm, "null communicator");
    GH_REQUIRE(width >= 0 && rows >= 0, "negative size");
    if (width == 0 || rows == 0) return GH_OK;
    GH_REQUIRE(cur_luma_buf && edge, "null pointer");
    Comm *c = static_cast<Comm *>(comm);
    const size_t rb = (size_t)width;
    return sharded_step_pipelined(c, cur_luma_buf, prefetch_luma_buf, rb, rows, (hipStream_t)stream, [&](uint8_t *buf, int tb, int bb) {
        return gh_sobel_u8_rows_dev(buf + rb, edge, width, rows, tb, bb, stream);
    });
}

}  // extern "C"

// Stream of frames in the halo-flag way: `cur` is filtered by ONE launch whose edge bands wait (in the kernel, in front of their halo row)
// for the exchange that the PREVIOUS call sent on its way when `cur` was its `next`; the halo rows of `next` travel on the side stream
// beside this launch, behind `started` -- no event, no stream-level wait.  By the time a launch needs its halo row the exchange has had a
// whole launch to complete.  cur's halo rows not on their way (first call, another buffer): they are exchanged beside this launch, as in
// the single-frame step.  Frames whose kernel form carries no flag wait: the event-based pipelined step (`fallback`).
template <typename LaunchWait, typename Plain, typename Fallback>
static int sharded_stream_step(Comm *c, uint8_t *cur, uint8_t *next, size_t rb, int rows, hipStream_t s, LaunchWait launch_wait /* (tb, bb, hw) */,
                               Plain plain /* () : no neighbours */, Fallback fallback /* () */)
{
    int up, down;
    neighbours(c, up, down);
    if (up < 0 && down < 0) return plain();
    GhHaloWait probe{};
    probe.pad = 1;
    halo_flag_review(c);
    const int form = step_form_of(c);
    if (form != 2 || launch_wait(up < 0, down < 0, probe) != GH_OK || !caller_stream_ok(c, s)) return fallback();
    if (gh_comm_halo_timeouts(c))
        return gh_set_error(GH_ERR_HIP, "an earlier sharded step gave up waiting for its halo rows (%u waits timed out): results of that step are invalid",
                            gh_comm_halo_timeouts(c));
    if (int rc = halo_flag_guard(c, s)) return rc;
    GhHaloWait hw = halo_flag_launch_args(c);
    if (c->stream_buf == cur) {
        hw.wait_epoch = c->stream_xepoch;                 // on its way since the previous call
    } else if (int rc = halo_flag_exchange(c, hw, cur, rb, rows, up, down, &hw.wait_epoch)) {
        halo_flag_release(s, hw);
        return rc;
    }
    c->stream_buf = nullptr;
    if (next && next != cur) {
        // next's rows 1 / rows are in place (enqueued on s before this call): the exchange may start as soon as this launch has begun
        if (int rc = halo_flag_exchange(c, hw, next, rb, rows, up, down, &c->stream_xepoch)) { halo_flag_release(s, hw); return rc; }
        c->stream_buf = next;
    }
    const int rc = launch_wait(up < 0, down < 0, hw);
    if (rc != GH_OK) halo_flag_release(s, hw);
    return rc;
}

extern "C" {

int gh_sharded_fused_stream_step_dev(gh_comm_t comm, uint8_t *cur_rgb_buf, uint8_t *next_rgb_buf, uint8_t *edge, int width, int rows, int channels,
                                     int mode, gh_stream_t stream)
{
    GH_REQUIRE(comm, "null communicator");
    GH_REQUIRE(width >= 0 && rows >= 0, "negative size");
    if (width == 0 || rows == 0) return GH_OK;
    GH_REQUIRE(cur_rgb_buf && edge, "null pointer");
    Comm *c = static_cast<Comm *>(comm);
    const size_t rb = (size_t)width * channels;
    return sharded_stream_step(c, cur_rgb_buf, next_rgb_buf, rb, rows, (hipStream_t)stream,
        [&](int tb, int bb, const GhHaloWait &hw) { return gh_fused_rows_halo_wait(cur_rgb_buf + rb, edge, width, rows, channels, mode, tb, bb, stream, &hw); },
        [&] { return gh_gray_sobel_fused_rows_dev(cur_rgb_buf + rb, edge, width, rows, channels, mode, 1, 1, stream); },
        [&] { return gh_sharded_fused_step_pipelined_dev(comm, cur_rgb_buf, next_rgb_buf, edge, width, rows, channels, mode, stream); });
}

int gh_sharded_sobel_stream_step_dev(gh_comm_t comm, uint8_t *cur_luma_buf, uint8_t *next_luma_buf, uint8_t *edge, int width, int rows, gh_stream_t stream)
{
    GH_REQUIRE(comm, "null communicator");
    GH_REQUIRE(width >= 0 && rows >= 0, "negative size");
    if (width == 0 || rows == 0) return GH_OK;
    GH_REQUIRE(cur_luma_buf && edge, "null pointer");
    Comm *c = static_cast<Comm *>(comm);
    const size_t rb = (size_t)width;
    return sharded_stream_step(c, cur_luma_buf, next_luma_buf, rb, rows, (hipStream_t)stream,
        [&](int tb, int bb, const GhHaloWait &hw) { return gh_sobel_rows_halo_wait(cur_luma_buf + rb, edge, width, rows, tb, bb, stream, &hw, 0); },
        [&] { return gh_sobel_u8_rows_dev(cur_luma_buf + rb, edge, width, rows, 1, 1, stream); },
        [&] { return gh_sharded_sobel_step_pipelined_dev(comm, cur_luma_buf, next_luma_buf, edge, width, rows, stream); });
}

int gh_sharded_fused_step_dev(gh_comm_t comm, uint8_t *rgb_buf, uint8_t *edge, int width, int rows, int channels, int mode, gh_stream_t stream)
{
    GH_REQUIRE(comm, "null communicator");
    GH_REQUIRE(width >= 0 && rows >= 0, "negative size");
    if (width == 0 || rows == 0) return GH_OK;
    GH_REQUIRE(rgb_buf && edge, "null pointer");
    Comm *c = static_cast<Comm *>(comm);
    const size_t rb = (size_t)width * channels;
    return sharded_step(c, rgb_buf, rb, rows, (hipStream_t)stream, [&](int first, int n, int tb, int bb) {
        return gh_gray_sobel_fused_rows_dev(rgb_buf + (size_t)(first + 1) * rb, edge + (size_t)first * width, width, n, channels, mode, tb, bb, stream);
    }, [&](int tb, int bb, const GhHaloWait &hw) {
        return gh_fused_rows_halo_wait(rgb_buf + rb, edge, width, rows, channels, mode, tb, bb, stream, &hw);
    });
}

// luma: 0 = buf holds luma rows; 1 / 2 = buf holds the previous pass's EDGE rows and the kernel applies lum(e, e, e) on load (IEEE / FMA):
// the halo row exchanged is then the neighbour's raw edge row
static int sharded_sobel_step(gh_comm_t comm, uint8_t *plane_buf, uint8_t *edge, int width, int rows, int luma, gh_stream_t stream)
{
    GH_REQUIRE(comm, "null communicator");
    GH_REQUIRE(width >= 0 && rows >= 0, "negative size");
    if (width == 0 || rows == 0) return GH_OK;
    GH_REQUIRE(plane_buf && edge, "null pointer");
    Comm *c = static_cast<Comm *>(comm);
    const size_t rb = (size_t)width;
    return sharded_step(c, plane_buf, rb, rows, (hipStream_t)stream, [&](int first, int n, int tb, int bb) {
        return gh_sobel_rows_halo_wait(plane_buf + (size_t)(first + 1) * rb, edge + (size_t)first * width, width, n, tb, bb, stream, nullptr, luma);
    }, [&](int tb, int bb, const GhHaloWait &hw) {
        return gh_sobel_rows_halo_wait(plane_buf + rb, edge, width, rows, tb, bb, stream, &hw, luma);
    });
}

int gh_sharded_sobel_step_dev(gh_comm_t comm, uint8_t *luma_buf, uint8_t *edge, int width, int rows, gh_stream_t stream)
{
    return sharded_sobel_step(comm, luma_buf, edge, width, rows, 0, stream);
}

int gh_sharded_sobel_of_gray_step_dev(gh_comm_t comm, uint8_t *edge_buf, uint8_t *edge, int width, int rows, int mode, gh_stream_t stream)
{
    GH_REQUIRE(mode == GH_MODE_IEEE || mode == GH_MODE_FMA, "bad mode");
    return sharded_sobel_step(comm, edge_buf, edge, width, rows, 1 + mode, stream);
}

// 3x3 convolution of a row shard (old/parallel_convolution.c:8-24 on a sharded frame): buf holds
// rows + 2 RGB rows, the halo is one RGB row per neighbour (the taps reach one row up and down); the frame's own top / bottom
// edge keeps the zero border.
int gh_sharded_conv3x3_step_dev(gh_comm_t comm, uint8_t *rgb_buf, uint8_t *dst, int width, int rows, const int32_t kernel[9], int32_t norm,
                                gh_stream_t stream)
{
    GH_REQUIRE(comm, "null communicator");
    GH_REQUIRE(width >= 0 && rows >= 0, "negative size");
    if (width == 0 || rows == 0) return GH_OK;
    GH_REQUIRE(rgb_buf && dst && kernel, "null pointer");
    Comm *c = static_cast<Comm *>(comm);
    const size_t rb = (size_t)width * 3;
    return sharded_step(c, rgb_buf, rb, rows, (hipStream_t)stream, [&](int first, int n, int tb, int bb) {
        return gh_conv3x3_rgb8_rows_dev(rgb_buf + (size_t)(first + 1) * rb, dst + (size_t)first * rb, width, n, kernel, norm, tb, bb, stream);
    }, [&](int, int, const GhHaloWait &) { return GH_NO_FLAG_FORM; });       // (the convolution kernels carry no flag wait: serial form)
}

// `passes` x (gray -> extract -> Sobel -> replicate) on a row shard (monolithic/src/main_with_sobel.c:49-76 on a sharded frame, SURVEY.md
// section 8e row 4): pass k + 1 filters what pass k wrote, so the rows next to a shard edge depend on the neighbour's pass-k edge rows
// -- ONE halo exchange per pass.  Pass 1 exchanges an RGB row and runs the fused kernel (the luma plane is dead after it, as in
// gh_pipeline_gray_sobel_dev); pass k > 1 exchanges the neighbours' raw EDGE rows and runs the plane Sobel with lum(e, e, e) applied on
// load (not the identity: 13 grey levels move in IEEE mode) -- one launch and 2 B/px per pass instead of a luma pass plus a Sobel pass.
// The edge planes of successive passes alternate between the two work planes, each (rows + 2) rows of `width` bytes (rows 1..rows = the
// plane, rows 0 / rows + 1 receive the neighbours' rows); the last pass writes edge_ws, whose rows 1..rows hold its edge plane on return.
// img_buf: rows + 2 rows of width * channels bytes (rows 1..rows = the shard, in place like the reference).
int gh_sharded_pipeline_gray_sobel_dev(gh_comm_t comm, uint8_t *img_buf, int width, int rows, int channels, int passes, int mode,
                                       uint8_t *gray_ws, uint8_t *edge_ws, gh_stream_t stream)
{
    GH_REQUIRE(comm, "null communicator");
    GH_REQUIRE(width >= 0 && rows >= 0, "negative size");
    GH_REQUIRE(channels >= 1 && channels <= 4, "channels must be 1..4");
    GH_REQUIRE(mode == GH_MODE_IEEE || mode == GH_MODE_FMA, "bad mode");
    if (passes < 1) passes = 1;   // main_with_sobel.c:44-45
    if (width == 0 || rows == 0) return GH_OK;
    GH_REQUIRE(img_buf && gray_ws && edge_ws && gray_ws != edge_ws, "null or aliased work planes");
    const size_t npix = (size_t)width * rows, pitch = (size_t)width * channels;
    uint8_t *shard = img_buf + pitch;
    // pass p (0-based) writes plane[(passes - 1 - p) & 1]: the last pass lands in edge_ws
    uint8_t *const plane[2] = {edge_ws, gray_ws};
    for (int p = 0; p < passes; ++p) {
        uint8_t *out = plane[(passes - 1 - p) & 1], *in = plane[(passes - p) & 1];
        int rc;
        if (p == 0 && channels >= 3) {
            rc = gh_sharded_fused_step_dev(comm, img_buf, out + width, width, rows, channels, mode, stream);
        } else if (p == 0) {                       // channels 1 / 2: the plane is channel 0, filtered as it is by every pass
            if ((rc = gh_gray_rgb8_to_plane_dev(shard, in + width, npix, channels, mode, stream))) return rc;
            rc = gh_sharded_sobel_step_dev(comm, in, out + width, width, rows, stream);
        } else if (channels >= 3) {                // the image holds (e, e, e) now: its luma is lum(e, e, e), applied on load
            rc = gh_sharded_sobel_of_gray_step_dev(comm, in, out + width, width, rows, mode, stream);
        } else {
            rc = gh_sharded_sobel_step_dev(comm, in, out + width, width, rows, stream);
        }
        if (rc) return rc;
    }
    return gh_replicate_plane_dev(edge_ws + width, shard, npix, channels, stream);
}

}  // extern "C"
