// Multi-GPU decompositions behind the C ABI: the 2-D slab transform and the four-step split of one
// long 1-D transform (include/mfft.h: mfft_plan_2d_dist / mfft_plan_1d_dist), one rank per GPU.
//
// Built on the library's own public entry points (mfft_plan_1d_layout, mfft_plan_1d_columns,
// mfft_exec, mfft_swap_axes, mfft_transpose) for everything local, and on RCCL for the exchange
// steps: every exchange is a group of point-to-point ncclSend / ncclRecv pairs -- what an
// all-to-all is made of -- cut into row chunks, so that a chunk is on the xGMI links (a full mesh:
// every peer pair has its own link, one hop per message) while the next one is being produced or
// the previous one consumed.  Exchanges run on a stream of their own; events order them against
// the caller's stream, on which all kernels run.
//
// RCCL is loaded on first use (dlopen by soname, so a process that already holds a copy -- PyTorch
// ships one -- shares it); single-GPU users of libmfft.so never load it.  A second communicator
// kind, "local", puts several ranks on ONE device inside one process (a host thread per rank,
// exchanges as device copies ordered by events): the multi-rank index algebra, chunking and stream
// ordering run unchanged on a one-GPU box.
#include <dlfcn.h>

#include <cstring>
#include <new>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include <rccl/rccl.h>

#include "common.h"

namespace mfft {

// ---------------------------------------------------------------------------------------------
// RCCL, resolved at run time
// ---------------------------------------------------------------------------------------------
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static int rccl(Rccl** out)
{
    static std::mutex mu;
    static Rccl r;
    std::lock_guard<std::mutex> lock(mu);
    if (!r.handle) {
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) { set_error(std::string("cannot load RCCL: ") + dlerror()); return MFFT_E_UNSUPPORTED; }
        bool ok = true;
        const auto sym = [&](const char* name) { void* p = dlsym(h, name); if (!p) ok = false; return p; };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        if (!ok) { set_error("the RCCL library lacks a required symbol"); return MFFT_E_UNSUPPORTED; }
        r.handle = h;
    }
    *out = &r;
    return MFFT_OK;
}

#define MFFT_NCCL(R, call)                                                                   \
    do {                                                                                     \
        const ncclResult_t r__ = (call);                                                     \
        if (r__ != ncclSuccess) {                                                            \
            set_error(std::string(#call) + " failed: " + (R)->GetErrorString(r__));          \
            return MFFT_E_HIP;                                                               \
        }                                                                                    \
    } while (0)

// ---------------------------------------------------------------------------------------------
// communicators
// ---------------------------------------------------------------------------------------------
// Ranks of a "local" communicator share one device and one process; their host threads meet here.
struct LocalGroup {
    int n = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    std::vector<const void* const*> send;      // per rank: its array of per-destination block pointers
    std::vector<hipEvent_t> ready, done;        // per rank: "my blocks are produced" / "I have copied what I receive"
    void barrier()
    {
        std::unique_lock<std::mutex> lock(mu);
        const uint64_t g = generation;
        if (++arrived == n) { arrived = 0; ++generation; cv.notify_all(); }
        else cv.wait(lock, [&] { return generation != g; });
    }
};

struct Comm {
    int world = 1, rank = 0, device = 0;
    bool local = false;
    ncclComm_t nccl = nullptr;
    std::shared_ptr<LocalGroup> group;

    // Personalised exchange of contiguous blocks of `bytes` bytes: send[p] goes to rank p, recv[p]
    // receives what rank p sends here.  Enqueued on `s`.
    int exchange(const void* const* send, void* const* recv, size_t bytes, hipStream_t s)
    {
        if (bytes == 0) return MFFT_OK;
        if (!local) {
            // the own block goes through the same group (RCCL turns a send to oneself into a copy)
            Rccl* r = nullptr;
            const int rc = rccl(&r);
            if (rc) return rc;
            MFFT_NCCL(r, r->GroupStart());
            for (int p = 0; p < world; ++p) {
                MFFT_NCCL(r, r->Send(send[p], bytes, ncclChar, p, nccl, s));
                MFFT_NCCL(r, r->Recv(recv[p], bytes, ncclChar, p, nccl, s));
            }
            MFFT_NCCL(r, r->GroupEnd());
            return MFFT_OK;
        }
        LocalGroup& g = *group;
        MFFT_HIP(hipEventRecord(g.ready[(size_t)rank], s));
        g.send[(size_t)rank] = send;
        g.barrier();                                   // every rank has published its blocks and its event
        for (int p = 0; p < world; ++p) {
            if (p != rank) MFFT_HIP(hipStreamWaitEvent(s, g.ready[(size_t)p], 0));
            MFFT_HIP(hipMemcpyAsync(recv[p], g.send[(size_t)p][rank], bytes, hipMemcpyDeviceToDevice, s));
        }
        MFFT_HIP(hipEventRecord(g.done[(size_t)rank], s));
        g.barrier();                                   // every rank has enqueued its copies
        for (int p = 0; p < world; ++p)                // my blocks may be overwritten once the peers have read them
            if (p != rank) MFFT_HIP(hipStreamWaitEvent(s, g.done[(size_t)p], 0));
        return MFFT_OK;
    }
};

static size_t elem_size(int dtype) { return dtype == MFFT_C64 ? 8 : 16; }

// largest divisor of `rows` not above the request (1 with one rank)
static int chunk_count(int requested, int64_t rows, int world)
{
    if (world == 1) return 1;
    int64_t c = requested > 0 ? requested : 4;
    if (c > rows) c = rows;
    while (c > 1 && rows % c) --c;
    return (int)(c < 1 ? 1 : c);
}

// ---------------------------------------------------------------------------------------------
// plans
// ---------------------------------------------------------------------------------------------
struct DistPlan {
    enum Kind { SLAB_2D, FOURSTEP_1D } kind = SLAB_2D;
    Comm* comm = nullptr;
    int dtype = 0, device = 0;
    int64_t a0 = 0, a1 = 0;            // 2-D: n0, n1;  1-D: n1, n2
    int option = 0;                    // 2-D: restore_layout;  1-D: natural_order
    int chunks = 1;
    mfft_plan rows = nullptr;          // contiguous-axis transforms of one row chunk
    mfft_plan cols = nullptr;          // strided-axis transforms (column-batched)
    void* buf[2] = {nullptr, nullptr}; // two buffers of the local size
    size_t local_bytes = 0;
    hipStream_t s_comm = nullptr;
    std::vector<hipEvent_t> ev;        // producer -> exchange, exchange -> consumer, per chunk
    std::mutex mu;

    ~DistPlan()
    {
        int prev = 0;
        const bool have = hipGetDevice(&prev) == hipSuccess;
        (void)hipSetDevice(device);
        if (rows) (void)mfft_plan_destroy(rows);
        if (cols) (void)mfft_plan_destroy(cols);
        for (void* b : buf) if (b) (void)hipFree(b);
        for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
        if (s_comm) (void)hipStreamDestroy(s_comm);
        if (have) (void)hipSetDevice(prev);
    }
    hipEvent_t& produced(int c) { return ev[(size_t)(2 * c)]; }
    hipEvent_t& arrived(int c) { return ev[(size_t)(2 * c + 1)]; }
};

static int plan_common(DistPlan* p, size_t local_elems, int nchunks)
{
    p->local_bytes = local_elems * elem_size(p->dtype);
    for (void*& b : p->buf) MFFT_HIP(hipMalloc(&b, p->local_bytes));
    MFFT_HIP(hipStreamCreateWithFlags(&p->s_comm, hipStreamNonBlocking));
    p->ev.assign((size_t)(2 * nchunks + 2), nullptr);
    for (hipEvent_t& e : p->ev) MFFT_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return MFFT_OK;
}

// blocks of one chunked exchange: peer p's block starts at base + p*peer_stride + chunk_off (bytes)
static void block_ptrs(std::vector<const void*>* send, std::vector<void*>* recv, int world, const char* sbase, size_t s_peer,
                       size_t s_off, char* rbase, size_t r_peer, size_t r_off)
{
    send->resize((size_t)world);
    recv->resize((size_t)world);
    for (int p = 0; p < world; ++p) {
        (*send)[(size_t)p] = sbase + (size_t)p * s_peer + s_off;
        (*recv)[(size_t)p] = rbase + (size_t)p * r_peer + r_off;
    }
}

// 2-D: out-of-place on this rank's R = n0/P rows (row-major, n1 columns)
static int exec_2d(DistPlan* p, const char* in, char* out, hipStream_t s)
{
    Comm& cm = *p->comm;
    const int P = cm.world, C = p->chunks;
    const int64_t n0 = p->a0, n1 = p->a1, R = n0 / P, W = n1 / P, Rc = R / C;
    const size_t es = elem_size(p->dtype);
    char* X = static_cast<char*>(p->buf[0]);
    char* Y = static_cast<char*>(p->buf[1]);
    std::vector<const void*> snd;
    std::vector<void*> rcv;
    int rc;
    // row transforms; bin k of local row i is stored at [k / W][i][k % W] of the chunk's send
    // buffer: already packed per destination.  Chunk c travels while chunk c + 1 is transformed and
    // lands as rows [r*R + c*Rc, +Rc) of the (n0 x W) column matrix.
    MFFT_HIP(hipEventRecord(p->arrived(C), s));                 // the buffers are free once earlier work on s is done
    MFFT_HIP(hipStreamWaitEvent(p->s_comm, p->arrived(C), 0));
    if (P == 1) {                                              // nothing to exchange: rows land in the column matrix
        rc = mfft_exec(p->rows, in, Y, 1.0, 0.0, 0.0, 1.0, nullptr, s);
        if (rc) return rc;
        return mfft_exec(p->cols, Y, out, 1.0, 0.0, 0.0, 1.0, nullptr, s);
    }
    for (int c = 0; c < C; ++c) {
        const size_t chunk = (size_t)c * (size_t)Rc * (size_t)n1 * es;
        rc = mfft_exec(p->rows, in + chunk, X + chunk, 1.0, 0.0, 0.0, 1.0, nullptr, s);
        if (rc) return rc;
        MFFT_HIP(hipEventRecord(p->produced(c), s));
        MFFT_HIP(hipStreamWaitEvent(p->s_comm, p->produced(c), 0));
        block_ptrs(&snd, &rcv, P, X + chunk, (size_t)Rc * W * es, 0, Y, (size_t)R * W * es, (size_t)c * Rc * W * es);
        rc = cm.exchange(snd.data(), rcv.data(), (size_t)Rc * W * es, p->s_comm);
        if (rc) return rc;
    }
    MFFT_HIP(hipEventRecord(p->arrived(0), p->s_comm));
    MFFT_HIP(hipStreamWaitEvent(s, p->arrived(0), 0));
    // column transforms on the strided columns where the exchange left them (no transpose)
    if (!p->option)
        return mfft_exec(p->cols, Y, out, 1.0, 0.0, 0.0, 1.0, nullptr, s);
    rc = mfft_exec(p->cols, Y, X, 1.0, 0.0, 0.0, 1.0, nullptr, s);      // X[k0][c] = F[k0][rank*W + c]
    if (rc) return rc;
    // way back: rows [q*R, (q+1)*R) belong to rank q and are contiguous; they travel in row chunks,
    // and a chunk that has arrived ([q][i][c], P x Rc x W) is unpacked into its Rc output rows while
    // the next one is on the links
    MFFT_HIP(hipEventRecord(p->produced(0), s));
    MFFT_HIP(hipStreamWaitEvent(p->s_comm, p->produced(0), 0));
    for (int c = 0; c < C; ++c) {
        const size_t piece = (size_t)Rc * W * es;
        block_ptrs(&snd, &rcv, P, X, (size_t)R * W * es, (size_t)c * piece, Y + (size_t)c * P * piece, piece, 0);
        rc = cm.exchange(snd.data(), rcv.data(), piece, p->s_comm);
        if (rc) return rc;
        MFFT_HIP(hipEventRecord(p->arrived(c), p->s_comm));
        MFFT_HIP(hipStreamWaitEvent(s, p->arrived(c), 0));
        rc = mfft_swap_axes(Y + (size_t)c * P * piece, out + (size_t)c * Rc * n1 * es, P, Rc, W, p->dtype, s);
        if (rc) return rc;
    }
    return MFFT_OK;
}

// 1-D four-step on this rank's n/P consecutive samples
static int exec_1d(DistPlan* p, const char* in, char* out, hipStream_t s)
{
    Comm& cm = *p->comm;
    const int P = cm.world, C = p->chunks;
    const int64_t n1 = p->a0, n2 = p->a1, R1 = n1 / P, W2 = n2 / P, Rc = R1 / C;
    const size_t es = elem_size(p->dtype);
    char* X = static_cast<char*>(p->buf[0]);
    char* Y = static_cast<char*>(p->buf[1]);
    std::vector<const void*> snd;
    std::vector<void*> rcv;
    int rc;
    const size_t piece = (size_t)Rc * W2 * es;                  // one (chunk, peer) block
    const char* cur = in;
    MFFT_HIP(hipEventRecord(p->arrived(C), s));
    MFFT_HIP(hipStreamWaitEvent(p->s_comm, p->arrived(C), 0));
    if (P > 1) {
        // pack: local rows j1 (Rc x P x W2 per chunk) -> [q][i][c]; rank q owns columns [q*W2, (q+1)*W2)
        for (int c = 0; c < C; ++c) {
            const size_t chunk = (size_t)c * (size_t)Rc * (size_t)n2 * es;
            rc = mfft_swap_axes(in + chunk, X + chunk, Rc, P, W2, p->dtype, s);
            if (rc) return rc;
            MFFT_HIP(hipEventRecord(p->produced(c), s));
            MFFT_HIP(hipStreamWaitEvent(p->s_comm, p->produced(c), 0));
            block_ptrs(&snd, &rcv, P, X + chunk, piece, 0, Y, (size_t)R1 * W2 * es, (size_t)c * piece);
            rc = cm.exchange(snd.data(), rcv.data(), piece, p->s_comm);
            if (rc) return rc;
        }
        MFFT_HIP(hipEventRecord(p->arrived(0), p->s_comm));
        MFFT_HIP(hipStreamWaitEvent(s, p->arrived(0), 0));
        cur = Y;                                                // (n1 x W2): cur[j1][c] = x[j1*n2 + rank*W2 + c]
    }
    // n1-point transforms on the strided columns, inter-step twiddle W_n^(k1 * (rank*W2 + c)) folded
    // into their last store:  X[k1][c]
    rc = mfft_exec(p->cols, cur, X, 1.0, 0.0, 0.0, 1.0, nullptr, s);
    if (rc) return rc;
    // Second exchange + n2-point transforms, chunked over this rank's rows k1: the blocks of chunk c
    // (rows [c*Rc, +Rc) from every peer, landing at [q][i][:]) are transformed while chunk c + 1
    // travels.  Sample j2 of local row i sits at [j2 / W2][i][j2 % W2].
    // natural order: bin k2 of row i goes to [k2 / W2][i][k2 % W2] of `Z` (the owner of natural index
    // k2*n1 + k1), and chunk c leaves for its owners -- into the very regions of Y that chunk c of the
    // second exchange occupied and the transform has just consumed -- while chunk c + 1 is transformed.
    const bool natural = p->option != 0;
    const char* rows_in = X;
    if (P > 1) {
        MFFT_HIP(hipEventRecord(p->produced(0), s));
        MFFT_HIP(hipStreamWaitEvent(p->s_comm, p->produced(0), 0));
        rows_in = Y;
    }
    char* Z = natural ? (P > 1 ? out : Y) : out;
    // Order on the exchange stream: second-exchange chunk c + 1 is enqueued before the wait for the
    // transform of chunk c, so it travels during that transform, and third-exchange chunk c travels
    // during the transform of chunk c + 1.  Every rank enqueues the same sequence.
    const auto second_exchange = [&](int c) -> int {
        block_ptrs(&snd, &rcv, P, X, (size_t)R1 * W2 * es, (size_t)c * piece, Y, (size_t)R1 * W2 * es, (size_t)c * piece);
        const int e = cm.exchange(snd.data(), rcv.data(), piece, p->s_comm);
        if (e) return e;
        MFFT_HIP(hipEventRecord(p->arrived(c), p->s_comm));
        return MFFT_OK;
    };
    if (P > 1) {
        rc = second_exchange(0);
        if (rc) return rc;
    }
    for (int c = 0; c < C; ++c) {
        if (P > 1) {
            if (c + 1 < C) {
                rc = second_exchange(c + 1);
                if (rc) return rc;
            }
            MFFT_HIP(hipStreamWaitEvent(s, p->arrived(c), 0));
        }
        // chunk c: signals i in [c*Rc, +Rc); in a segmented buffer signal i starts at i*W2
        const size_t seg_off = (size_t)c * piece;
        const size_t dense_off = (size_t)c * (size_t)Rc * (size_t)n2 * es;
        rc = mfft_exec(p->rows, rows_in + (P > 1 ? seg_off : dense_off), Z + ((natural && P > 1) ? seg_off : dense_off),
                       1.0, 0.0, 0.0, 1.0, nullptr, s);
        if (rc) return rc;
        if (natural && P > 1) {
            MFFT_HIP(hipEventRecord(p->produced(c), s));
            MFFT_HIP(hipStreamWaitEvent(p->s_comm, p->produced(c), 0));
            block_ptrs(&snd, &rcv, P, Z, (size_t)R1 * W2 * es, (size_t)c * piece, Y, (size_t)R1 * W2 * es, (size_t)c * piece);
            rc = cm.exchange(snd.data(), rcv.data(), piece, p->s_comm);
            if (rc) return rc;
        }
    }
    if (!natural) return MFFT_OK;                               // out[i*n2 + k2] = X[(rank*R1 + i) + n1*k2]
    if (P > 1) {
        MFFT_HIP(hipEventRecord(p->arrived(0), p->s_comm));
        MFFT_HIP(hipStreamWaitEvent(s, p->arrived(0), 0));
    }
    // Y[r][i][c] = X[(rank*W2 + c)*n1 + r*R1 + i]: an (n1 x W2) matrix; its transpose is the natural-order chunk
    return mfft_transpose(Y, out, n1, W2, 1, p->dtype, s);
}

}  // namespace mfft

using namespace mfft;

extern "C" {

int mfft_comm_unique_id(void* id128)
{
    if (!id128) { set_error("null argument"); return MFFT_E_INVALID; }
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    Rccl* r = nullptr;
    const int rc = rccl(&r);
    if (rc) return rc;
    ncclUniqueId id;
    MFFT_NCCL(r, r->GetUniqueId(&id));
    memcpy(id128, &id, sizeof id);
    return MFFT_OK;
}

int mfft_comm_init_rank(mfft_comm* out, int world, int rank, const void* id128)
{
    if (!out || !id128 || world < 1 || rank < 0 || rank >= world) { set_error("bad communicator description"); return MFFT_E_INVALID; }
    *out = nullptr;
    if (mfft_device_count() < 0) return MFFT_E_NODEVICE;
    Rccl* r = nullptr;
    const int rc = rccl(&r);
    if (rc) return rc;
    std::unique_ptr<Comm> c(new (std::nothrow) Comm);
    if (!c) { set_error("host allocation failed"); return MFFT_E_NOMEM; }
    c->world = world; c->rank = rank;
    MFFT_HIP(hipGetDevice(&c->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    MFFT_NCCL(r, r->CommInitRank(&c->nccl, world, id, rank));
    *out = reinterpret_cast<mfft_comm>(c.release());
    return MFFT_OK;
}

int mfft_comm_init_local(mfft_comm* out, int nranks)
{
    if (!out || nranks < 1 || nranks > 64) { set_error("1 <= nranks <= 64"); return MFFT_E_INVALID; }
    for (int i = 0; i < nranks; ++i) out[i] = nullptr;
    if (mfft_device_count() < 0) return MFFT_E_NODEVICE;
    int device = 0;
    MFFT_HIP(hipGetDevice(&device));
    auto g = std::make_shared<LocalGroup>();
    g->n = nranks;
    g->send.assign((size_t)nranks, nullptr);
    g->ready.assign((size_t)nranks, nullptr);
    g->done.assign((size_t)nranks, nullptr);
    for (int i = 0; i < nranks; ++i) {
        MFFT_HIP(hipEventCreateWithFlags(&g->ready[(size_t)i], hipEventDisableTiming));
        MFFT_HIP(hipEventCreateWithFlags(&g->done[(size_t)i], hipEventDisableTiming));
    }
    for (int i = 0; i < nranks; ++i) {
        Comm* c = new (std::nothrow) Comm;
        if (!c) { set_error("host allocation failed"); return MFFT_E_NOMEM; }
        c->world = nranks; c->rank = i; c->device = device; c->local = true; c->group = g;
        out[i] = reinterpret_cast<mfft_comm>(c);
    }
    return MFFT_OK;
}

int mfft_comm_info(mfft_comm comm, int* world, int* rank)
{
    if (!comm || !world || !rank) { set_error("null argument"); return MFFT_E_INVALID; }
    const Comm* c = reinterpret_cast<const Comm*>(comm);
    *world = c->world;
    *rank = c->rank;
    return MFFT_OK;
}

int mfft_comm_exchange(mfft_comm comm, const void* const* d_send, void* const* d_recv, int64_t bytes, void* stream)
{
    if (!comm || !d_send || !d_recv || bytes < 0) { set_error("bad argument"); return MFFT_E_INVALID; }
    Comm* c = reinterpret_cast<Comm*>(comm);
    for (int p = 0; p < c->world; ++p)
        if (bytes > 0 && (!d_send[p] || !d_recv[p])) { set_error("null block pointer"); return MFFT_E_INVALID; }
    return c->exchange(d_send, d_recv, (size_t)bytes, static_cast<hipStream_t>(stream));
}

int mfft_comm_destroy(mfft_comm comm)
{
    if (!comm) return MFFT_OK;
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (c->nccl) {
        Rccl* r = nullptr;
        if (rccl(&r) == MFFT_OK) (void)r->CommDestroy(c->nccl);
    }
    if (c->local && c->group && c->group.use_count() == 1) {
        for (hipEvent_t e : c->group->ready) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : c->group->done) if (e) (void)hipEventDestroy(e);
    }
    delete c;
    return MFFT_OK;
}

static int dist_plan_begin(mfft_dist_plan* out, mfft_comm comm, int dtype, std::unique_ptr<DistPlan>* p)
{
    if (!out || !comm) { set_error("null argument"); return MFFT_E_INVALID; }
    *out = nullptr;
    if (dtype != MFFT_C64 && dtype != MFFT_C128) { set_error("dtype must be MFFT_C64 or MFFT_C128"); return MFFT_E_INVALID; }
    p->reset(new (std::nothrow) DistPlan);
    if (!*p) { set_error("host allocation failed"); return MFFT_E_NOMEM; }
    (*p)->comm = reinterpret_cast<Comm*>(comm);
    (*p)->dtype = dtype;
    MFFT_HIP(hipGetDevice(&(*p)->device));
    if ((*p)->device != (*p)->comm->device) { set_error("the communicator belongs to another device"); return MFFT_E_INVALID; }
    return MFFT_OK;
}

int mfft_plan_2d_dist(mfft_dist_plan* out, mfft_comm comm, int64_t n0, int64_t n1, int dtype, int restore_layout,
                      int chunks)
{
    std::unique_ptr<DistPlan> p;
    int rc = dist_plan_begin(out, comm, dtype, &p);
    if (rc) return rc;
    const int P = p->comm->world;
    if (n0 < 1 || n1 < 1 || n0 % P || n1 % P) { set_error("both dimensions must divide by the number of ranks"); return MFFT_E_INVALID; }
    p->kind = DistPlan::SLAB_2D;
    p->a0 = n0; p->a1 = n1; p->option = restore_layout ? 1 : 0;
    const int64_t R = n0 / P, W = n1 / P;
    p->chunks = chunk_count(chunks, R, P);
    const int64_t Rc = R / p->chunks;
    const mfft_layout lout{W, W, Rc * W};
    rc = mfft_plan_1d_layout(&p->rows, n1, n1, Rc, dtype, 0, nullptr, P > 1 ? &lout : nullptr);
    if (rc) return rc;
    rc = mfft_plan_1d_columns(&p->cols, n0, W, 1, dtype, 0, 0);
    if (rc) return rc;                 // shapes the tile passes do not serve: MFFT_E_UNSUPPORTED
    rc = plan_common(p.get(), (size_t)R * (size_t)n1, p->chunks);
    if (rc) return rc;
    *out = reinterpret_cast<mfft_dist_plan>(p.release());
    return MFFT_OK;
}

int mfft_plan_1d_dist(mfft_dist_plan* out, mfft_comm comm, int64_t n1, int64_t n2, int dtype, int natural_order,
                      int chunks)
{
    std::unique_ptr<DistPlan> p;
    int rc = dist_plan_begin(out, comm, dtype, &p);
    if (rc) return rc;
    const int P = p->comm->world;
    if (n1 < 1 || n2 < 1 || n1 % P || n2 % P) { set_error("both factors must divide by the number of ranks"); return MFFT_E_INVALID; }
    if (n1 > (int64_t(1) << 31) / n2) { set_error("n1 * n2 above 2^31"); return MFFT_E_UNSUPPORTED; }
    p->kind = DistPlan::FOURSTEP_1D;
    p->a0 = n1; p->a1 = n2; p->option = natural_order ? 1 : 0;
    const int64_t R1 = n1 / P, W2 = n2 / P;
    p->chunks = chunk_count(chunks, R1, P);
    const int64_t Rc = R1 / p->chunks;
    rc = mfft_plan_1d_columns(&p->cols, n1, W2, 1, dtype, n1 * n2, (int64_t)p->comm->rank * W2);
    if (rc) return rc;
    const mfft_layout seg{W2, W2, R1 * W2};
    rc = mfft_plan_1d_layout(&p->rows, n2, n2, Rc, dtype, 0, P > 1 ? &seg : nullptr, (P > 1 && natural_order) ? &seg : nullptr);
    if (rc) return rc;
    rc = plan_common(p.get(), (size_t)R1 * (size_t)n2, p->chunks);
    if (rc) return rc;
    *out = reinterpret_cast<mfft_dist_plan>(p.release());
    return MFFT_OK;
}

int mfft_dist_exec(mfft_dist_plan plan, const void* d_in, void* d_out, void* stream)
{
    if (!plan || !d_in || !d_out) { set_error("null argument"); return MFFT_E_INVALID; }
    if (d_in == d_out) { set_error("distributed plans run out of place"); return MFFT_E_INVALID; }
    DistPlan* p = reinterpret_cast<DistPlan*>(plan);
    int prev = 0;
    MFFT_HIP(hipGetDevice(&prev));
    if (prev != p->device) MFFT_HIP(hipSetDevice(p->device));
    std::lock_guard<std::mutex> lock(p->mu);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int rc = p->kind == DistPlan::SLAB_2D ? exec_2d(p, static_cast<const char*>(d_in), static_cast<char*>(d_out), s)
                                                : exec_1d(p, static_cast<const char*>(d_in), static_cast<char*>(d_out), s);
    if (prev != p->device) (void)hipSetDevice(prev);
    return rc;
}

int mfft_dist_plan_info(mfft_dist_plan plan, int* chunks, int64_t* workspace_bytes)
{
    if (!plan || !chunks || !workspace_bytes) { set_error("null argument"); return MFFT_E_INVALID; }
    const DistPlan* p = reinterpret_cast<const DistPlan*>(plan);
    *chunks = p->chunks;
    *workspace_bytes = (int64_t)(2 * p->local_bytes);
    return MFFT_OK;
}

int mfft_dist_plan_destroy(mfft_dist_plan plan)
{
    delete reinterpret_cast<DistPlan*>(plan);
    return MFFT_OK;
}

}  // extern "C"
