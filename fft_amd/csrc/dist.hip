// Multi-GPU decompositions behind the C ABI: the 2-D slab transform and the four-step split of one
// long 1-D transform (include/mfft.h: mfft_plan_2d_dist / mfft_plan_1d_dist), one rank per GPU.
//
// Built on the library's own public entry points (mfft_plan_1d_layout, mfft_plan_1d_columns,
// mfft_exec, mfft_swap_axes, mfft_transpose) for everything local, and on RCCL for the exchange
// steps: every exchange is a group of point-to-point ncclSend / ncclRecv pairs -- what an
// all-to-all is made of -- on the xGMI links (a full mesh: every peer pair has its own link, one hop
// per message).  The data is cut two ways so that the links never wait for a local step and no local
// step waits for the whole exchange: into GROUPS OF COLUMNS, each of which is exchanged, transformed
// along the strided axis and sent on as a unit while the next group is still arriving, and -- for the
// group at either end of the pipeline -- into row chunks, so that the step before the first exchange
// and the step after the last one overlap them too.  Exchanges run on a stream of their own, in one
// fixed order on every rank; events order them against the caller's stream, on which all kernels run.
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
            ncclResult_t bad = ncclSuccess;
            const char* what = "";
            for (int p = 0; p < world && bad == ncclSuccess; ++p) {
                bad = r->Send(send[p], bytes, ncclChar, p, nccl, s);
                what = "ncclSend";
                if (bad == ncclSuccess) { bad = r->Recv(recv[p], bytes, ncclChar, p, nccl, s); what = "ncclRecv"; }
            }
            // the group is closed also after a failure: left open on this thread, later RCCL calls would
            // be queued into it and the peers would wait for them
            const ncclResult_t ended = r->GroupEnd();
            if (bad != ncclSuccess) { set_error(std::string(what) + " failed: " + r->GetErrorString(bad)); return MFFT_E_HIP; }
            if (ended != ncclSuccess) { set_error(std::string("ncclGroupEnd failed: ") + r->GetErrorString(ended)); return MFFT_E_HIP; }
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

// Column groups: the largest divisor of this rank's `cols` columns not above the request that leaves groups
// of at least 64 (complex128) / 128 (complex64) columns -- the shortest segments the contiguous-axis kernels
// read and write in place, and at least a 1 KiB run per row of a group.  Default 2: the step behind the last
// exchange (row transforms, unpacking) hides only behind the LAST group's exchange, so more groups shorten
// the window it has (replayed with measured kernel times and 153 GB/s links at eight ranks: 2^31 complex128
// 11.2 ms with one group, 7.6 with two, 8.1 with four; 16384^2 complex64 0.47 / 0.39 / 0.44 ms).
static int group_count(int requested, int64_t cols, int world, int dtype)
{
    if (world == 1) return 1;
    int64_t g = requested > 0 ? requested : 2;
    const int64_t floor_w = dtype == MFFT_C64 ? 128 : 64;
    if (g > cols) g = cols;
    while (g > 1 && (cols % g || cols / g < floor_w)) --g;
    return (int)(g < 1 ? 1 : g);
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
    int chunks = 1;                    // row chunks of the first group's first exchange and the last group's last
    int groups = 1;                    // column groups
    mfft_plan rows = nullptr;          // contiguous-axis transforms of one row chunk
    std::vector<mfft_plan> cols;       // strided-axis transforms (column-batched) of one column group: one plan, or -- 1-D,
                                       // whose fused twiddle depends on the group's first column -- one per group
    void* buf[2] = {nullptr, nullptr}; // two buffers of the local size
    size_t local_bytes = 0;
    hipStream_t s_comm = nullptr;
    std::vector<hipEvent_t> ev;        // stream-to-stream hand-offs of one execution, taken in turn
    size_t ev_next = 0;
    std::mutex mu;

    ~DistPlan()
    {
        int prev = 0;
        const bool have = hipGetDevice(&prev) == hipSuccess;
        (void)hipSetDevice(device);
        if (rows) (void)mfft_plan_destroy(rows);
        for (mfft_plan c : cols) if (c) (void)mfft_plan_destroy(c);
        for (void* b : buf) if (b) (void)hipFree(b);
        for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
        if (s_comm) (void)hipStreamDestroy(s_comm);
        if (have) (void)hipSetDevice(prev);
    }
    mfft_plan col_plan(int g) const { return cols.size() == 1 ? cols[0] : cols[(size_t)g]; }
    // `to` continues only after everything enqueued on `from` so far.  An event may be recorded again once
    // the wait on its previous record has been enqueued, so a fixed pool taken in turn is enough.
    int link(hipStream_t from, hipStream_t to)
    {
        hipEvent_t e = ev[ev_next++ % ev.size()];
        MFFT_HIP(hipEventRecord(e, from));
        MFFT_HIP(hipStreamWaitEvent(to, e, 0));
        return MFFT_OK;
    }
    // an event of its own (recorded now on `from`, waited for later)
    int mark(hipStream_t from, hipEvent_t* out)
    {
        *out = ev[ev_next++ % ev.size()];
        MFFT_HIP(hipEventRecord(*out, from));
        return MFFT_OK;
    }
};

static int plan_common(DistPlan* p, size_t local_elems)
{
    p->local_bytes = local_elems * elem_size(p->dtype);
    for (void*& b : p->buf) MFFT_HIP(hipMalloc(&b, p->local_bytes));
    MFFT_HIP(hipStreamCreateWithFlags(&p->s_comm, hipStreamNonBlocking));
    // one execution takes at most 3 * (groups + chunks) + 8 events; marks stay valid while fewer than the
    // pool's size are taken after them
    p->ev.assign((size_t)(8 * (p->groups + p->chunks) + 16), nullptr);
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

#define MFFT_TRY(call) do { const int rc__ = (call); if (rc__) return rc__; } while (0)

// 2-D: out-of-place on this rank's R = n0/P rows (row-major, n1 columns).
//
// The n1 columns are cut into P*G segments of Wc = n1/(P*G) columns; rank q transforms, as its group g,
// segment s(q, g) = q*G + g when the result stays in column slabs (its columns must be the contiguous
// range [q*W, (q+1)*W)) and g*P + q when the row-slab layout is restored (any assignment serves, and this
// one lets every group's buffers be reused as soon as that group has moved on).
//   rows      bin k of local row i is stored at [k / Wc][i][k % Wc]: every (peer, group) block contiguous
//   exchange  group-major; the first group in row chunks, each leaving as soon as its rows are transformed
//   columns   group g is transformed along the strided axis as soon as it has arrived, while group g + 1
//             travels -- straight into its column block of the result, or (restore_layout) into the send
//             buffer of the way back, which leaves while group g + 1 is transformed
//   unpack    (restore_layout) the last group travels in row chunks; a chunk's rows are unpacked while the
//             next chunk is on the links
static int exec_2d(DistPlan* p, const char* in, char* out, hipStream_t s)
{
    Comm& cm = *p->comm;
    const int P = cm.world, C = p->chunks, G = p->groups;
    const int64_t n0 = p->a0, n1 = p->a1, R = n0 / P, W = n1 / P, Rc = R / C, Wc = W / G;
    const size_t es = elem_size(p->dtype);
    char* X = static_cast<char*>(p->buf[0]);
    char* Y = static_cast<char*>(p->buf[1]);
    hipStream_t sc = p->s_comm;
    std::vector<const void*> snd;
    std::vector<void*> rcv;
    p->ev_next = 0;
    if (P == 1) {                                              // nothing to exchange: rows land in the column matrix
        MFFT_TRY(mfft_exec(p->rows, in, Y, 1.0, 0.0, 0.0, 1.0, nullptr, s));
        return mfft_exec(p->col_plan(0), Y, out, 1.0, 0.0, 0.0, 1.0, nullptr, s);
    }
    const bool restore = p->option != 0;
    const size_t blk = (size_t)R * Wc * es;                    // one (peer, group) block: R rows of Wc columns
    const size_t gsz = (size_t)P * blk;                        // one group's (n0 x Wc) column matrix
    const auto seg_of = [&](int q, int g) { return restore ? (size_t)g * P + q : (size_t)q * G + g; };
    MFFT_TRY(p->link(s, sc));                                  // the buffers are free once earlier work on s is done
    std::vector<hipEvent_t> rows_done((size_t)C), here((size_t)G), back((size_t)C);
    for (int c = 0; c < C; ++c) {
        MFFT_TRY(mfft_exec(p->rows, in + (size_t)c * Rc * n1 * es, X + (size_t)c * Rc * Wc * es, 1.0, 0.0, 0.0, 1.0, nullptr, s));
        MFFT_TRY(p->mark(s, &rows_done[(size_t)c]));
    }
    for (int g = 0; g < G; ++g) {
        const int nck = g == 0 ? C : 1;
        const size_t piece = blk / nck;
        for (int c = 0; c < nck; ++c) {
            if (g == 0) MFFT_HIP(hipStreamWaitEvent(sc, rows_done[(size_t)c], 0));
            snd.resize((size_t)P); rcv.resize((size_t)P);
            for (int q = 0; q < P; ++q) {
                snd[(size_t)q] = X + seg_of(q, g) * blk + (size_t)c * piece;
                rcv[(size_t)q] = Y + (size_t)g * gsz + (size_t)q * blk + (size_t)c * piece;       // rows [q*R + c*R/nck, ...) of group g
            }
            MFFT_TRY(cm.exchange(snd.data(), rcv.data(), piece, sc));
        }
        MFFT_TRY(p->mark(sc, &here[(size_t)g]));
    }
    for (int g = 0; g < G; ++g) {
        MFFT_HIP(hipStreamWaitEvent(s, here[(size_t)g], 0));
        if (!restore) {                                        // out[k0][rank's column g*Wc + w]: rows W elements apart
            MFFT_TRY(mfft_exec(p->col_plan(g), Y + (size_t)g * gsz, out + (size_t)g * Wc * es, 1.0, 0.0, 0.0, 1.0, nullptr, s));
            continue;
        }
        // group g's send blocks have left (its exchange is complete): their region takes the transformed group,
        // whose rows [q*R, (q+1)*R) belong to rank q and are contiguous
        MFFT_TRY(mfft_exec(p->col_plan(g), Y + (size_t)g * gsz, X + (size_t)g * gsz, 1.0, 0.0, 0.0, 1.0, nullptr, s));
        MFFT_TRY(p->link(s, sc));
        const int nck = g == G - 1 ? C : 1;
        const size_t piece = blk / nck;
        for (int c = 0; c < nck; ++c) {
            snd.resize((size_t)P); rcv.resize((size_t)P);
            for (int q = 0; q < P; ++q) {
                snd[(size_t)q] = X + (size_t)g * gsz + (size_t)q * blk + (size_t)c * piece;
                rcv[(size_t)q] = Y + (size_t)g * gsz + (size_t)q * blk + (size_t)c * piece;       // [segment g*P + q][row][w]
            }
            MFFT_TRY(cm.exchange(snd.data(), rcv.data(), piece, sc));
            if (g == G - 1) MFFT_TRY(p->mark(sc, &back[(size_t)c]));
        }
    }
    if (!restore) return MFFT_OK;
    // Y[s][i][w] = F[rank*R + i][s*Wc + w]: out[i][s][w], Rc rows at a time as the last group's chunks arrive
    for (int c = 0; c < C; ++c) {
        MFFT_HIP(hipStreamWaitEvent(s, back[(size_t)c], 0));
        MFFT_TRY(mfft_swap_axes_strided(Y + (size_t)c * Rc * Wc * es, out + (size_t)c * Rc * n1 * es, (int64_t)P * G, Rc, Wc,
                                        R * Wc, Wc, p->dtype, s));
    }
    return MFFT_OK;
}

// 1-D four-step on this rank's n/P consecutive samples: x[j1*n2 + j2], X[k1 + n1*k2].
//
// The n2 columns j2 are cut into P*G segments of Wc columns; rank q transforms segment g*P + q as its group g.
//   pack      group g of local row i, [q][i][w] = x[i*n2 + (g*P + q)*Wc + w]: one strided block copy
//   exchange  group-major; the first group in row chunks, packed and sent one after the other
//   columns   the n1-point transforms of group g (inter-step twiddle folded into their last store) run while
//             group g + 1 travels; their rows k1 in [q*R1, (q+1)*R1) belong to rank q, are contiguous, and
//             leave while group g + 1 is transformed -- into the region group g arrived in, so sample j2 of
//             local row i ends up at [j2 / Wc][i][j2 % Wc], a plain segmented layout
//   rows      the last group travels in row chunks; the n2-point transforms of a chunk's rows read the
//             segments in place while the next chunk is on the links
//   natural   bin k2 of row i goes to [k2 / W2][i][k2 % W2] (the owner of natural index k2*n1 + k1); a
//             chunk leaves for its owners while the next one is transformed; one transpose at the end
static int exec_1d(DistPlan* p, const char* in, char* out, hipStream_t s)
{
    Comm& cm = *p->comm;
    const int P = cm.world, C = p->chunks, G = p->groups;
    const int64_t n1 = p->a0, n2 = p->a1, R1 = n1 / P, W2 = n2 / P, Rc = R1 / C, Wc = W2 / G;
    const size_t es = elem_size(p->dtype);
    char* X = static_cast<char*>(p->buf[0]);
    char* Y = static_cast<char*>(p->buf[1]);
    hipStream_t sc = p->s_comm;
    std::vector<const void*> snd;
    std::vector<void*> rcv;
    const bool natural = p->option != 0;
    p->ev_next = 0;
    if (P == 1) {
        MFFT_TRY(mfft_exec(p->col_plan(0), in, X, 1.0, 0.0, 0.0, 1.0, nullptr, s));
        if (!natural) return mfft_exec(p->rows, X, out, 1.0, 0.0, 0.0, 1.0, nullptr, s);     // out[i*n2 + k2] = X[i + n1*k2]
        MFFT_TRY(mfft_exec(p->rows, X, Y, 1.0, 0.0, 0.0, 1.0, nullptr, s));
        return mfft_transpose(Y, out, n1, n2, 1, p->dtype, s);
    }
    const size_t blk = (size_t)R1 * Wc * es;                   // one (peer, group) block
    const size_t gsz = (size_t)P * blk;                        // one group's (n1 x Wc) column matrix
    MFFT_TRY(p->link(s, sc));
    std::vector<hipEvent_t> here((size_t)G), landed((size_t)C);
    for (int g = 0; g < G; ++g) {
        const int nck = g == 0 ? C : 1;
        const int64_t rows = R1 / nck;
        const size_t piece = blk / nck;
        for (int c = 0; c < nck; ++c) {
            char* pack = X + (size_t)g * gsz + (size_t)c * P * piece;          // [q][i][w], P x rows x Wc
            MFFT_TRY(mfft_swap_axes_strided(in + ((size_t)c * rows * n2 + (size_t)g * P * Wc) * es, pack, rows, P, Wc, n2, Wc, p->dtype, s));
            MFFT_TRY(p->link(s, sc));
            snd.resize((size_t)P); rcv.resize((size_t)P);
            for (int q = 0; q < P; ++q) {
                snd[(size_t)q] = pack + (size_t)q * piece;
                rcv[(size_t)q] = Y + (size_t)g * gsz + (size_t)q * blk + (size_t)c * piece;       // rows j1 = q*R1 + c*rows + i
            }
            MFFT_TRY(cm.exchange(snd.data(), rcv.data(), piece, sc));
        }
        MFFT_TRY(p->mark(sc, &here[(size_t)g]));
    }
    for (int g = 0; g < G; ++g) {
        MFFT_HIP(hipStreamWaitEvent(s, here[(size_t)g], 0));
        MFFT_TRY(mfft_exec(p->col_plan(g), Y + (size_t)g * gsz, X + (size_t)g * gsz, 1.0, 0.0, 0.0, 1.0, nullptr, s));
        MFFT_TRY(p->link(s, sc));
        const int nck = g == G - 1 ? C : 1;
        const size_t piece = blk / nck;
        for (int c = 0; c < nck; ++c) {
            snd.resize((size_t)P); rcv.resize((size_t)P);
            for (int q = 0; q < P; ++q) {
                snd[(size_t)q] = X + (size_t)g * gsz + (size_t)q * blk + (size_t)c * piece;
                rcv[(size_t)q] = Y + (size_t)g * gsz + (size_t)q * blk + (size_t)c * piece;       // [segment g*P + q][row][w]
            }
            MFFT_TRY(cm.exchange(snd.data(), rcv.data(), piece, sc));
            if (g == G - 1) MFFT_TRY(p->mark(sc, &landed[(size_t)c]));
        }
    }
    // Every send out of X has been enqueued before the first receive into it below, and the exchange stream
    // runs in order: X is free for the natural-order exchange to land in.
    const size_t run = (size_t)Rc * W2 * es;                   // a chunk's rows within one peer's (R1 x W2) block
    for (int c = 0; c < C; ++c) {
        MFFT_HIP(hipStreamWaitEvent(s, landed[(size_t)c], 0));
        if (!natural) {
            MFFT_TRY(mfft_exec(p->rows, Y + (size_t)c * Rc * Wc * es, out + (size_t)c * Rc * n2 * es, 1.0, 0.0, 0.0, 1.0, nullptr, s));
            continue;
        }
        MFFT_TRY(mfft_exec(p->rows, Y + (size_t)c * Rc * Wc * es, out + (size_t)c * run, 1.0, 0.0, 0.0, 1.0, nullptr, s));
        MFFT_TRY(p->link(s, sc));
        snd.resize((size_t)P); rcv.resize((size_t)P);
        for (int q = 0; q < P; ++q) {
            snd[(size_t)q] = out + (size_t)q * R1 * W2 * es + (size_t)c * run;
            rcv[(size_t)q] = X + (size_t)q * R1 * W2 * es + (size_t)c * run;
        }
        MFFT_TRY(cm.exchange(snd.data(), rcv.data(), run, sc));
    }
    if (!natural) return MFFT_OK;                               // out[i*n2 + k2] = X[(rank*R1 + i) + n1*k2]
    MFFT_TRY(p->link(sc, s));
    // X[r][i][c] = X[(rank*W2 + c)*n1 + r*R1 + i]: an (n1 x W2) matrix; its transpose is the natural-order chunk
    return mfft_transpose(X, out, n1, W2, 1, p->dtype, s);
}

// Column-batched plans of one group, retried with fewer groups while the tile passes do not serve the width.
// make(g_count, g_index, &plan) builds one plan; `per_group`: one plan per group (else one plan serves all).
template <class Make>
static int build_col_plans(DistPlan* p, int requested, int64_t cols, bool per_group, Make make)
{
    int g = group_count(requested, cols, p->comm->world, p->dtype);
    for (;;) {
        int rc = MFFT_OK;
        const int count = per_group ? g : 1;
        p->cols.assign((size_t)count, nullptr);
        for (int i = 0; i < count && rc == MFFT_OK; ++i) rc = make(g, i, &p->cols[(size_t)i]);
        if (rc == MFFT_OK) { p->groups = g; return MFFT_OK; }
        for (mfft_plan& c : p->cols) { if (c) (void)mfft_plan_destroy(c); c = nullptr; }
        p->cols.clear();
        if (rc != MFFT_E_UNSUPPORTED || g == 1) return rc;      // shapes the tile passes do not serve at all: MFFT_E_UNSUPPORTED
        do { --g; } while (g > 1 && cols % g);
    }
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
    // on a failure part-way nothing is left behind: the handles made so far and the group's events are released
    const auto undo = [&] {
        for (int i = 0; i < nranks; ++i) { delete reinterpret_cast<Comm*>(out[i]); out[i] = nullptr; }
        for (hipEvent_t e : g->ready) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : g->done) if (e) (void)hipEventDestroy(e);
    };
    for (int i = 0; i < nranks; ++i) {
        hipError_t e = hipEventCreateWithFlags(&g->ready[(size_t)i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&g->done[(size_t)i], hipEventDisableTiming);
        if (e != hipSuccess) { undo(); return hip_fail(e, "hipEventCreateWithFlags", __FILE__, __LINE__); }
    }
    for (int i = 0; i < nranks; ++i) {
        Comm* c = new (std::nothrow) Comm;
        if (!c) { undo(); set_error("host allocation failed"); return MFFT_E_NOMEM; }
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
                      int chunks, int col_groups)
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
    rc = build_col_plans(p.get(), col_groups, W, false, [&](int g, int, mfft_plan* plan) {
        // column slabs out: a group's columns are a block of the (n0 x W) result, rows W elements apart
        return mfft_plan_1d_columns_ld(plan, n0, W / g, 1, dtype, 0, 0, restore_layout ? 0 : W);
    });
    if (rc) return rc;
    const int64_t Wc = W / p->groups;
    const mfft_layout lout{Wc, Wc, R * Wc};
    rc = mfft_plan_1d_layout(&p->rows, n1, n1, Rc, dtype, 0, nullptr, P > 1 ? &lout : nullptr);
    if (rc) return rc;
    rc = plan_common(p.get(), (size_t)R * (size_t)n1);
    if (rc) return rc;
    *out = reinterpret_cast<mfft_dist_plan>(p.release());
    return MFFT_OK;
}

int mfft_plan_1d_dist(mfft_dist_plan* out, mfft_comm comm, int64_t n1, int64_t n2, int dtype, int natural_order,
                      int chunks, int col_groups)
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
    const int rank = p->comm->rank;
    rc = build_col_plans(p.get(), col_groups, W2, true, [&](int g, int i, mfft_plan* plan) {
        // group i of this rank is column segment i*P + rank: its first column sets the fused twiddle
        const int64_t wc = W2 / g;
        return mfft_plan_1d_columns(plan, n1, wc, 1, dtype, n1 * n2, ((int64_t)i * P + rank) * wc);
    });
    if (rc) return rc;
    const int64_t Wc = W2 / p->groups;
    const mfft_layout seg_in{Wc, Wc, R1 * Wc}, seg_out{W2, W2, R1 * W2};
    rc = mfft_plan_1d_layout(&p->rows, n2, n2, Rc, dtype, 0, P > 1 ? &seg_in : nullptr, (P > 1 && natural_order) ? &seg_out : nullptr);
    if (rc) return rc;
    rc = plan_common(p.get(), (size_t)R1 * (size_t)n2);
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

int mfft_dist_plan_info(mfft_dist_plan plan, int* chunks, int* col_groups, int64_t* workspace_bytes)
{
    if (!plan || !chunks || !col_groups || !workspace_bytes) { set_error("null argument"); return MFFT_E_INVALID; }
    const DistPlan* p = reinterpret_cast<const DistPlan*>(plan);
    *chunks = p->chunks;
    *col_groups = p->groups;
    *workspace_bytes = (int64_t)(2 * p->local_bytes);
    return MFFT_OK;
}

int mfft_dist_plan_destroy(mfft_dist_plan plan)
{
    delete reinterpret_cast<DistPlan*>(plan);
    return MFFT_OK;
}

}  // extern "C"
