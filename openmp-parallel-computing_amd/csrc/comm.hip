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
struct NcclId { char internal[kIdBytes]; };
using nccl_comm_t = void *;
enum { kNcclUint8 = 1, kNcclUint64 = 5, kNcclSum = 0 };   // ncclDataType_t / ncclRedOp_t values of rccl.h

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
    bool loopback = false;               // rehearsal on one GPU: both neighbours are this rank (ncclSend/ncclRecv to self)
    // pipelined steps: buffers whose halo rows the side stream has already been asked to fill, and the event that says
    // they have landed.  A few entries, so that a halo can be requested two or three frames ahead.
    static constexpr int kPrefetch = 4;
    const void *pf_buf[kPrefetch] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t pf_landed[kPrefetch] = {nullptr, nullptr, nullptr, nullptr};
    int pf_next = 0;
};

}  // namespace

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
        GH_HIP_TRY(hipEventCreateWithFlags(&c->ready, hipEventDisableTiming));
        GH_HIP_TRY(hipEventCreateWithFlags(&c->landed, hipEventDisableTiming));
        for (int i = 0; i < Comm::kPrefetch; ++i) GH_HIP_TRY(hipEventCreateWithFlags(&c->pf_landed[i], hipEventDisableTiming));
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

int gh_comm_set_loopback(gh_comm_t comm, int on)
{
    GH_REQUIRE(comm, "null communicator");
    static_cast<Comm *>(comm)->loopback = on != 0;
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
    return GH_OK;
}

int gh_comm_allreduce_sum_u64_dev(gh_comm_t comm, uint64_t *buf_dev, size_t count, gh_stream_t stream)
{
    GH_REQUIRE(comm && buf_dev, "null pointer");
    Comm *c = static_cast<Comm *>(comm);
    GH_NCCL_TRY(g_rccl.AllReduce(buf_dev, buf_dev, count, kNcclUint64, kNcclSum, c->nccl, (hipStream_t)stream));
    return GH_OK;
}

}  // extern "C"

// ---- one pipeline step of a row shard ------------------------------------------------------------------------------
// buf: (rows + 2) rows of `row_bytes`; rows 1..rows are the shard, rows 0 / rows+1 receive the neighbours' edge rows.
// Two forms: serial (exchange, then one launch) and overlapped (exchange on the side stream || interior rows on the caller's
// stream -> first and last row once the halo has landed).
template <typename Launch>
static int sharded_step(Comm *c, uint8_t *buf, size_t row_bytes, int rows, hipStream_t s, Launch launch /* (first_row, nrows, top_border, bot_border) */)
{
    int up = c->rank > 0 ? c->rank - 1 : -1, down = c->rank + 1 < c->world ? c->rank + 1 : -1;
    if (c->loopback) up = down = c->rank;      // a middle shard whose neighbours are itself: halo = its own edge rows
    if (up < 0 && down < 0) return launch(0, rows, 1, 1);
    // Serial form (default; GH_SHARD_STEP_SERIAL=0 selects the overlapped form below): the exchange is enqueued on the caller's
    // stream in front of ONE launch over the whole shard -- no side stream, no events, no one-row launches.  The exchange is then
    // not hidden, but it moves one row (~11 us in the loopback rehearsal), while the overlapped form pays two cross-queue event
    // hand-overs and two extra launches: 41 vs 58 us per step for the 2048-row shard of an 8-GPU split, 61 vs 65 us at 4096 rows,
    // 110 vs 116 us at 8192 rows (tools/r2_call_o.sh; the pipelined form, which prefetches halos two frames ahead: 38 us).
    if (GH_TUNE_INT("GH_SHARD_STEP_SERIAL", 1)) {
        if (int rc = gh_halo_exchange_dev(c, up, buf + row_bytes, buf, down, buf + (size_t)rows * row_bytes, buf + (size_t)(rows + 1) * row_bytes,
                                          row_bytes, s))
            return rc;
        return launch(0, rows, up < 0, down < 0);
    }
    // the exchange reads rows 1 / rows (written by earlier work on s) and overwrites the halo rows the previous step read
    GH_HIP_TRY(hipEventRecord(c->ready, s));
    GH_HIP_TRY(hipStreamWaitEvent(c->side, c->ready, 0));
    if (int rc = gh_halo_exchange_dev(c, up, buf + row_bytes, buf, down, buf + (size_t)rows * row_bytes, buf + (size_t)(rows + 1) * row_bytes,
                                      row_bytes, c->side))
        return rc;
    GH_HIP_TRY(hipEventRecord(c->landed, c->side));
    if (rows >= 3) {
        if (int rc = launch(1, rows - 2, 0, 0)) return rc;                       // interior: local rows only
        GH_HIP_TRY(hipStreamWaitEvent(s, c->landed, 0));
        if (int rc = launch(0, 1, up < 0, 0)) return rc;
        return launch(rows - 1, 1, 0, down < 0);
    }
    GH_HIP_TRY(hipStreamWaitEvent(s, c->landed, 0));
    return launch(0, rows, up < 0, down < 0);
}

// Pipelined form for a stream of frames held in a few buffers: the halo rows of `prefetch` (a frame that will be `cur`
// in a later call) are exchanged on the side stream WHILE `cur` -- whose halo rows were requested by an earlier call --
// is processed by one launch over the whole shard.  Per step the caller's stream sees a single kernel; the exchange
// latency, the cross-queue event latency (~20 us on this runtime when the wait is reached before the signal) and the
// two one-row launches of the form above leave the critical path once the halo is requested two frames ahead.  If
// cur's halo has not been requested (first calls, or the caller changed buffers), it is exchanged first.
template <typename Launch>
static int sharded_step_pipelined(Comm *c, uint8_t *cur, uint8_t *prefetch, size_t row_bytes, int rows, hipStream_t s, Launch launch)
{
    int up = c->rank > 0 ? c->rank - 1 : -1, down = c->rank + 1 < c->world ? c->rank + 1 : -1;
    if (c->loopback) up = down = c->rank;
    if (up < 0 && down < 0) return launch(cur, 1, 1);
    auto find = [&](const void *p) {
        for (int i = 0; i < Comm::kPrefetch; ++i)
            if (c->pf_buf[i] == p) return i;
        return -1;
    };
    auto exchange = [&](uint8_t *buf) -> int {      // returns the table slot (>= 0) or a negative status
        // reads buf's rows 1 / rows (written by earlier work on s), overwrites its halo rows (last read by earlier work on s)
        const int slot = c->pf_next;
        c->pf_next = (c->pf_next + 1) % Comm::kPrefetch;
        c->pf_buf[slot] = nullptr;
        if (hipEventRecord(c->ready, s) != hipSuccess || hipStreamWaitEvent(c->side, c->ready, 0) != hipSuccess)
            return -gh_set_error(GH_ERR_HIP, "event hand-over to the exchange stream failed");
        if (int rc = gh_halo_exchange_dev(c, up, buf + row_bytes, buf, down, buf + (size_t)rows * row_bytes, buf + (size_t)(rows + 1) * row_bytes,
                                          row_bytes, c->side))
            return rc < 0 ? rc : -rc;
        if (hipEventRecord(c->pf_landed[slot], c->side) != hipSuccess) return -gh_set_error(GH_ERR_HIP, "hipEventRecord failed");
        c->pf_buf[slot] = buf;
        return slot;
    };
    int slot = find(cur);
    if (slot < 0) {
        slot = exchange(cur);
        if (slot < 0) return -slot;
    }
    GH_HIP_TRY(hipStreamWaitEvent(s, c->pf_landed[slot], 0));   // cur's halo rows have landed
    c->pf_buf[slot] = nullptr;
    if (prefetch && prefetch != cur && find(prefetch) < 0) {
        // `ready` is recorded BEFORE cur's kernel is launched, so the exchange runs beside it
        const int ps = exchange(prefetch);
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
    });
}

int gh_sharded_sobel_step_dev(gh_comm_t comm, uint8_t *luma_buf, uint8_t *edge, int width, int rows, gh_stream_t stream)
{
    GH_REQUIRE(comm, "null communicator");
    GH_REQUIRE(width >= 0 && rows >= 0, "negative size");
    if (width == 0 || rows == 0) return GH_OK;
    GH_REQUIRE(luma_buf && edge, "null pointer");
    Comm *c = static_cast<Comm *>(comm);
    const size_t rb = (size_t)width;
    return sharded_step(c, luma_buf, rb, rows, (hipStream_t)stream, [&](int first, int n, int tb, int bb) {
        return gh_sobel_u8_rows_dev(luma_buf + (size_t)(first + 1) * rb, edge + (size_t)first * width, width, n, tb, bb, stream);
    });
}

}  // extern "C"
