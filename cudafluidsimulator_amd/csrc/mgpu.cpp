// In-process multi-GPU driver (include/sph_mgpu.h): z-slabs of whole cell layers, one per
// MI355X, a one-cell halo exchanged with RCCL send/recv over xGMI every step.  Host logic
// only -- the kernels are libsph_hip.so's, driven through the slab entry points of
// include/sph_c_api.h; there is no reference counterpart (the reference's step,
// simulator.cu:462-546, is single-GPU).
//
// Why z: it is the slowest digit of the flattened cell key (simulator.cu:78-82), so a slab
// is a contiguous range of the key-sorted particle streams and so are its boundary
// layers; gravity acts along y (simulator.cu:270-271) and does not drain slabs.
//
// Order of the combined array before its stable sort -- [halo from below | my migrants
// down | migrants from below | mine | migrants from above | my migrants up | halo from
// above] -- reproduces, inside every cell, the order a stable sort of the previous GLOBAL
// sequence (slabs concatenated by rank) would give, so N slabs equal the single domain bit
// for bit.  The same argument covers re-cutting the slabs (stable filter of that sequence).
//
// Host synchronisations per step: ONE (after exchange A, to read the partition bounds and
// the neighbours' headers); every later size is derived from those headers, and the
// derivation is checked against the sort's own bounds one step later.
#include "sph_mgpu.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_create_error;

struct F4 {
    float x, y, z, w;
};

// Message header = the sender's partition bounds (device-written) + a status word.
//   b[0] #rows with key < (zlo-1) D^2   (migrants down that land deeper than the first layer)
//   b[1] #rows with key <  zlo    D^2   (all migrants down)
//   b[2] #rows with key < (zlo+1) D^2   (end of the lower boundary layer)
//   b[3] #rows with key < (zhi-1) D^2   (start of the upper boundary layer)
//   b[4] #rows with key <  zhi    D^2   (start of the migrants up)
//   b[5] #rows with key < (zhi+1) D^2   (end of the migrants up that land in the first layer)
struct Hdr {
    int b[6];
    int n;
    int status;
};
static_assert(sizeof(Hdr) == 32, "header is 8 ints");

struct Slab {
    int rank = 0, device = 0;
    int zlo = 0, zhi = 0;
    bool has_dn = false, has_up = false;
    sph_handle *h = nullptr;
    hipStream_t s = nullptr;      // compute (owned by h unless shared)
    hipStream_t comm = nullptr;   // exchange B overlaps the interior force sweep
    hipStream_t bnd = nullptr;    // the boundary layers' force sweep (joins the interior's launch)
    hipStream_t copy = nullptr;   // position read-back
    hipEvent_t evDensity = nullptr, evB = nullptr, evBnd = nullptr, evForce = nullptr, evCopy = nullptr;
    hipEvent_t evT[3] = {nullptr, nullptr, nullptr}; // step start, grid done, force done
    hipEvent_t evTx[2] = {nullptr, nullptr};         // STREAMS transport: [compute, exchange] stream reached its sends
    hipEvent_t evRx[2] = {nullptr, nullptr};         //                    ... its receives have landed
    F4 *pos[2] = {nullptr, nullptr}, *vel[2] = {nullptr, nullptr};
    F4 *rx_pos[2] = {nullptr, nullptr}, *rx_vel[2] = {nullptr, nullptr}; // [0] from below, [1] from above
    F4 *ex_pos[2] = {nullptr, nullptr}, *ex_vel[2] = {nullptr, nullptr}; // overflow messages (rare)
    int ex_cap[2] = {0, 0};
    Hdr *hdr_tx = nullptr;        // device
    Hdr *hdr_rx = nullptr;        // device [2]
    int *sortb = nullptr;         // device: bounds of the combined sort (5 ints)
    int *pinned = nullptr;        // host: own Hdr (8) | rx Hdr x2 (16) | sort bounds (8)
    F4 *hostRows = nullptr;       // pinned: owned pos4 rows of the last step
    int hostRowsCount = 0;
    bool rowsStale = true;        // hostRows does not hold the owned rows (fresh upload, re-cut): refill on demand
    bool copyPending = false;
    ncclComm_t comm_nccl = nullptr;
    int cur = 0, off = 0, n_own = 0;
    // per step
    int sbuf = 0, n_comb = 0, i0 = 0, e_lo = 0, s_hi = 0, i1 = 0;
    Hdr mine{}, nb_dn{}, nb_up{};
    bool expectValid = false;
    int expect[4] = {0, 0, 0, 0};
    int status = 0;
};

// Optional host threads, one per local slab (SPH_MGPU_THREADS=1): the ~25 launches of a slab step
// cost the ONE host thread ~0.13 ms per slab, so an in-process run of eight GPUs is bound by the
// host at this problem size (profiles/r02_experiments.md).  The per-slab parts of a step (partition;
// assemble + sort + density; force + read-back) touch one slab each and run on the slab's worker;
// everything that spans slabs (message rounds, the host synchronisation, re-cuts) stays on the
// calling thread, between two joins.  Off by default: one host thread, as the reference has.
struct Workers {
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable wake, done;
    const std::function<int(int)> *job = nullptr; // slab index -> status
    long long generation = 0;
    int pending = 0;
    bool stop = false;
    std::vector<int> rc;
};

} // namespace

struct sph_mgpu {
    SphSettings settings{};
    SphMgpuOptions opt{};
    int n = 0, D = 0, DD = 0;
    int cap = 0, F = 0;
    std::vector<Slab> slabs;       // local slabs, ascending rank
    std::vector<int> cuts;         // world+1 layer cuts
    bool shared_stream = false;    // loopback / self transport: every slab on one stream
    hipStream_t shared = nullptr;
    std::vector<float> hostPos;    // n x 3, id order
    bool hostPosValid = false;
    bool ready = false;
    long long step = 0;
    SphMgpuStats stats{};
    std::string err;
    // step state carried between the phases of one step
    std::chrono::steady_clock::time_point t_begin;
    bool overflow = false;
    int phase = 0;                 // phases of the current step already done (0..3)
    bool clickQueued = false;      // sph_mgpu_queue_click: applied by the step that completes next
    // One process per GPU: a rank whose checks fail must not simply stop -- its neighbours would wait in
    // their next grouped receive for ever (the status word only travels with the NEXT exchange A).
    // A "poisoned" driver finishes the message rounds of the running step with the sizes the headers
    // dictate (payload: whatever the buffers hold), posts ONE more exchange A whose header carries
    // status = 1 -- the farewell -- and only then returns the error; a neighbour that reads status = 1
    // does the same towards ITS other neighbours, one rank per step.  Compute is skipped.
    bool poisoned = false;
    int poisonCode = 0;
    int clickX = 0, clickY = 0;
    std::mutex errMu;              // fail() from worker threads
    Workers *workers = nullptr;    // SPH_MGPU_THREADS=1
};

namespace {

int fail(sph_mgpu *m, int code, const std::string &msg) {
    if (m) {
        std::lock_guard<std::mutex> lk(m->errMu);
        m->err = msg;
    } else {
        g_create_error = msg;
    }
    return code;
}

// some rank of the run lives in another driver object (process): failures must be announced
bool distributed(const sph_mgpu *m) { return (int)m->slabs.size() < m->opt.world; }

// A check failed.  Every rank in this process: report at once.  Otherwise remember the first failure,
// keep the step's message rounds going (see sph_mgpu::poisoned) and report at the end of the step.
int poison(sph_mgpu *m, int code, const std::string &msg) {
    if (!distributed(m)) return fail(m, code, msg);
    if (!m->poisoned) {
        m->poisoned = true;
        m->poisonCode = code;
        (void)fail(m, code, msg);
        for (auto &sl : m->slabs) sl.status = 1;
    }
    return SPH_OK;
}

// Run fn(slab) for every local slab: in rank order on the calling thread, or on the slabs'
// worker threads (all joined before this returns).  First non-zero status wins.
int for_each_slab(sph_mgpu *m, const std::function<int(Slab &)> &fn) {
    Workers *w = m->workers;
    if (!w) {
        for (auto &sl : m->slabs) {
            int rc = fn(sl);
            if (rc) return rc;
        }
        return SPH_OK;
    }
    const std::function<int(int)> job = [&](int k) { return fn(m->slabs[k]); };
    {
        std::unique_lock<std::mutex> lk(w->mu);
        w->job = &job;
        w->pending = (int)w->threads.size();
        std::fill(w->rc.begin(), w->rc.end(), SPH_OK);
        ++w->generation;
        w->wake.notify_all();
        w->done.wait(lk, [&] { return w->pending == 0; });
        w->job = nullptr;
    }
    for (int rc : w->rc)
        if (rc) return rc;
    return SPH_OK;
}

void start_workers(sph_mgpu *m) {
    Workers *w = new Workers();
    const int n = (int)m->slabs.size();
    w->rc.assign(n, SPH_OK);
    for (int k = 0; k < n; ++k)
        w->threads.emplace_back([w, k]() {
            long long seen = 0;
            for (;;) {
                const std::function<int(int)> *job;
                {
                    std::unique_lock<std::mutex> lk(w->mu);
                    w->wake.wait(lk, [&] { return w->stop || w->generation != seen; });
                    if (w->stop) return;
                    seen = w->generation;
                    job = w->job;
                }
                const int rc = (*job)(k);
                {
                    std::lock_guard<std::mutex> lk(w->mu);
                    w->rc[k] = rc;
                    if (--w->pending == 0) w->done.notify_all();
                }
            }
        });
    m->workers = w;
}

void stop_workers(sph_mgpu *m) {
    Workers *w = m->workers;
    if (!w) return;
    {
        std::lock_guard<std::mutex> lk(w->mu);
        w->stop = true;
        w->wake.notify_all();
    }
    for (auto &t : w->threads) t.join();
    delete w;
    m->workers = nullptr;
}

#define HIPM(m, call)                                                                 \
    do {                                                                              \
        hipError_t e__ = (call);                                                      \
        if (e__ != hipSuccess)                                                        \
            return fail((m), SPH_EHIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)
#define NCCLM(m, call)                                                                \
    do {                                                                              \
        ncclResult_t r__ = (call);                                                    \
        if (r__ != ncclSuccess)                                                       \
            return fail((m), SPH_EHIP, std::string(#call) + ": " + ncclGetErrorString(r__)); \
    } while (0)
#define SPHM(m, sl, call)                                                             \
    do {                                                                              \
        int r__ = (call);                                                             \
        if (r__ != SPH_OK)                                                            \
            return fail((m), r__, std::string(#call) + ": " + sph_last_error((sl).h)); \
    } while (0)

// Cut D layers into `world` contiguous slabs of about equal particle counts, every slab
// at least min_layers thick (cuts on layer boundaries).
std::vector<int> partition_layers(const std::vector<long long> &hist, int world, int min_layers) {
    const int D = (int)hist.size();
    std::vector<long long> cum(D + 1, 0);
    for (int z = 0; z < D; ++z) cum[z + 1] = cum[z] + hist[z];
    const long long total = cum[D];
    std::vector<int> cuts{0};
    for (int r = 1; r < world; ++r) {
        const double target = (double)total * r / world;
        int z = (int)(std::lower_bound(cum.begin(), cum.end(), (long long)std::ceil(target)) - cum.begin());
        z = std::min(z, D);
        if (z > 0 && std::fabs((double)cum[z - 1] - target) <= std::fabs((double)cum[z] - target)) --z;
        z = std::max(z, cuts.back() + min_layers);
        z = std::min(z, D - (world - r) * min_layers);
        cuts.push_back(z);
    }
    cuts.push_back(D);
    return cuts;
}

int layer_of(const sph_mgpu *m, float z) {
    int c = (int)(z / m->settings.h); // getGridCell: IEEE divide, truncation (simulator.cu:59)
    return std::min(std::max(c, 0), m->D - 1);
}

void set_geometry(sph_mgpu *m) {
    for (auto &sl : m->slabs) {
        sl.zlo = m->cuts[sl.rank];
        sl.zhi = m->cuts[sl.rank + 1];
        sl.has_dn = sl.rank > 0;
        sl.has_up = sl.rank < m->opt.world - 1;
    }
}

// ---- message layouts (both ends compute them from a header) ----
// DOWN message = [migrants down | lower boundary layer] = rows [0, m1) of the sender's
// partitioned array; the window sent is rows [0, F); rows [F, m1) go in a second message.
// UP message = [upper boundary layer | migrants up] = rows [m2, n); the window is the LAST
// F rows, [n-F, n) ([0, F) if n < F); rows [m2, n-F) -- the FIRST rows -- go in a second one.
struct Layout {
    int payload, offset, extra;
};
Layout down_layout(const Hdr &h, int F) {
    const int payload = h.b[2];
    return {payload, 0, std::max(0, payload - F)};
}
Layout up_layout(const Hdr &h, int F) {
    const int payload = h.n - h.b[3];
    const int extra = std::max(0, payload - F);
    if (extra) return {payload, 0, extra};
    return {payload, h.n >= F ? F - payload : h.b[3], 0};
}

struct Piece {
    const F4 *p, *v;
    int count;
};

// pieces holding payload rows [a, b) of the message received from below (which = 0: an UP
// message) or from above (which = 1: a DOWN message)
void payload_rows(const sph_mgpu *m, const Slab &sl, int which, int a, int b, std::vector<Piece> &out) {
    if (b <= a) return;
    const F4 *rp = sl.rx_pos[which], *rv = sl.rx_vel[which];
    if (which == 0) { // extra rows come FIRST, the window holds the rest
        const Layout L = up_layout(sl.nb_dn, m->F);
        if (a < L.extra) out.push_back({sl.ex_pos[0] + a, sl.ex_vel[0] + a, std::min(b, L.extra) - a});
        if (b > L.extra) {
            const int lo = std::max(a, L.extra) - L.extra;
            out.push_back({rp + L.offset + lo, rv + L.offset + lo, (b - L.extra) - lo});
        }
    } else { // the window holds rows [0, F), the extra rows follow
        const Layout L = down_layout(sl.nb_up, m->F);
        const int inwin = L.payload - L.extra;
        if (a < inwin) out.push_back({rp + L.offset + a, rv + L.offset + a, std::min(b, inwin) - a});
        if (b > inwin) {
            const int lo = std::max(a, inwin) - inwin;
            out.push_back({sl.ex_pos[1] + lo, sl.ex_vel[1] + lo, (b - inwin) - lo});
        }
    }
}

int ensure_extra(sph_mgpu *m, Slab &sl, int which, int rows) {
    if (rows <= sl.ex_cap[which]) return SPH_OK;
    if (sl.ex_pos[which]) (void)hipFree(sl.ex_pos[which]);
    if (sl.ex_vel[which]) (void)hipFree(sl.ex_vel[which]);
    sl.ex_pos[which] = sl.ex_vel[which] = nullptr;
    const int capr = rows + rows / 4 + 1024;
    HIPM(m, hipMalloc(&sl.ex_pos[which], (size_t)capr * sizeof(F4)));
    HIPM(m, hipMalloc(&sl.ex_vel[which], (size_t)capr * sizeof(F4)));
    sl.ex_cap[which] = capr;
    return SPH_OK;
}

Slab *local(sph_mgpu *m, int rank) {
    for (auto &sl : m->slabs)
        if (sl.rank == rank) return &sl;
    return nullptr;
}

// SPH_TRANSPORT_MAILBOX (tests): several one-slab driver objects inside ONE process stand for
// the ranks of a one-process-per-GPU run; a send is a note in this table, the receive copies
// from it once the sending object has posted (the test steps every object phase by phase).
struct Mail { const void *src; size_t bytes; };
std::map<std::pair<int, int>, std::deque<Mail>> g_mail;
// receives of ALL objects waiting for the next phase: they are completed together, by
// whichever object enters the next phase first -- like RCCL, where a send has left the
// sender's buffer before any later kernel of the sender runs
struct PendingRecv { int src_rank, dst_rank; void *dst; size_t bytes; hipStream_t stream; long long epoch; };
std::vector<PendingRecv> g_pending;

// One message of a round.
struct Msg {
    int src_rank, dst_rank;
    const void *src; // valid if the sender is local
    void *dst;       // valid if the receiver is local
    size_t bytes;
};

// Deliver a round of messages.  RCCL: one group of sends and receives on each slab's
// `stream_of`; loopback: device-to-device copies on the (shared) stream.
int deliver(sph_mgpu *m, const std::vector<Msg> &msgs, bool on_comm_stream) {
    if (msgs.empty()) return SPH_OK;
    const int tr = m->opt.transport;
    if (tr == SPH_TRANSPORT_MAILBOX) {
        for (const Msg &g : msgs) {
            if (!g.bytes) continue;
            if (local(m, g.src_rank)) g_mail[{g.src_rank, g.dst_rank}].push_back({g.src, g.bytes});
            if (local(m, g.dst_rank))
                g_pending.push_back({g.src_rank, g.dst_rank, g.dst, g.bytes, m->shared, m->step * 4 + m->phase});
        }
        return SPH_OK;
    }
    if (tr == SPH_TRANSPORT_STREAMS) {
        // what a grouped ncclSend/ncclRecv round does to the participating streams, with copies:
        // a receive starts once the sender's stream has reached the round, and no stream of the
        // round goes on before the messages it sends and receives are through
        const int q = on_comm_stream ? 1 : 0;
        auto stream_of = [&](Slab *sl) { return on_comm_stream ? sl->comm : sl->s; };
        std::vector<Slab *> part;
        for (const Msg &g : msgs) {
            if (!g.src || !g.dst) return fail(m, SPH_ESTATE, "streams transport needs every slab in this process");
            if (!g.bytes) continue;
            for (int r : {g.src_rank, g.dst_rank}) {
                Slab *sl = local(m, r);
                if (std::find(part.begin(), part.end(), sl) == part.end()) part.push_back(sl);
            }
        }
        for (Slab *sl : part) HIPM(m, hipEventRecord(sl->evTx[q], stream_of(sl)));
        for (const Msg &g : msgs) {
            if (!g.bytes) continue;
            Slab *a = local(m, g.src_rank), *b = local(m, g.dst_rank);
            HIPM(m, hipStreamWaitEvent(stream_of(b), a->evTx[q], 0));
            HIPM(m, hipMemcpyAsync(g.dst, g.src, g.bytes, hipMemcpyDeviceToDevice, stream_of(b)));
        }
        for (Slab *sl : part) HIPM(m, hipEventRecord(sl->evRx[q], stream_of(sl)));
        for (const Msg &g : msgs) {
            if (!g.bytes) continue;
            Slab *a = local(m, g.src_rank), *b = local(m, g.dst_rank);
            HIPM(m, hipStreamWaitEvent(stream_of(a), b->evRx[q], 0)); // the sender's buffer is free again
        }
        return SPH_OK;
    }
    if (tr == SPH_TRANSPORT_LOOPBACK) {
        for (const Msg &g : msgs) {
            if (!g.src || !g.dst) return fail(m, SPH_ESTATE, "loopback transport needs every slab in this process");
            if (g.bytes) HIPM(m, hipMemcpyAsync(g.dst, g.src, g.bytes, hipMemcpyDeviceToDevice, m->shared));
        }
        return SPH_OK;
    }
    NCCLM(m, ncclGroupStart());
    if (tr == SPH_TRANSPORT_RCCL_SELF) {
        // one rank, every message goes to itself: sends and receives match in posting order
        Slab &s0 = m->slabs[0];
        for (const Msg &g : msgs)
            if (g.bytes) NCCLM(m, ncclSend(g.src, g.bytes, ncclChar, 0, s0.comm_nccl, m->shared));
        for (const Msg &g : msgs)
            if (g.bytes) NCCLM(m, ncclRecv(g.dst, g.bytes, ncclChar, 0, s0.comm_nccl, m->shared));
    } else {
        for (const Msg &g : msgs) {
            if (!g.bytes) continue;
            if (Slab *a = local(m, g.src_rank)) {
                HIPM(m, hipSetDevice(a->device));
                NCCLM(m, ncclSend(g.src, g.bytes, ncclChar, g.dst_rank, a->comm_nccl, on_comm_stream ? a->comm : a->s));
            }
            if (Slab *b = local(m, g.dst_rank)) {
                HIPM(m, hipSetDevice(b->device));
                NCCLM(m, ncclRecv(g.dst, g.bytes, ncclChar, g.src_rank, b->comm_nccl, on_comm_stream ? b->comm : b->s));
            }
        }
    }
    NCCLM(m, ncclGroupEnd());
    return SPH_OK;
}

// mailbox transport: complete the receives posted in the previous phase
int resolve_mail(sph_mgpu *m) {
    if (m->opt.transport != SPH_TRANSPORT_MAILBOX || g_pending.empty()) return SPH_OK;
    HIPM(m, hipDeviceSynchronize()); // the senders' data is complete (all objects share the device)
    const long long now = m->step * 4 + m->phase; // receives posted in EARLIER phases only
    std::vector<PendingRecv> later;
    for (const auto &r : g_pending) {
        if (r.epoch >= now) {
            later.push_back(r);
            continue;
        }
        auto &q = g_mail[{r.src_rank, r.dst_rank}];
        if (q.empty()) return fail(m, SPH_ESTATE, "mailbox: the sending rank has not run this phase yet");
        const Mail mm = q.front();
        q.pop_front();
        if (mm.bytes != r.bytes) return fail(m, SPH_ESTATE, "mailbox: sender and receiver disagree on a message size");
        HIPM(m, hipMemcpyAsync(r.dst, mm.src, r.bytes, hipMemcpyDeviceToDevice, r.stream));
    }
    g_pending.swap(later);
    HIPM(m, hipDeviceSynchronize());
    return SPH_OK;
}

int free_slab(Slab &sl) {
    (void)hipSetDevice(sl.device);
    if (sl.h) sph_destroy(sl.h);
    for (int b = 0; b < 2; ++b) {
        if (sl.pos[b]) (void)hipFree(sl.pos[b]);
        if (sl.vel[b]) (void)hipFree(sl.vel[b]);
        if (sl.rx_pos[b]) (void)hipFree(sl.rx_pos[b]);
        if (sl.rx_vel[b]) (void)hipFree(sl.rx_vel[b]);
        if (sl.ex_pos[b]) (void)hipFree(sl.ex_pos[b]);
        if (sl.ex_vel[b]) (void)hipFree(sl.ex_vel[b]);
    }
    if (sl.hdr_tx) (void)hipFree(sl.hdr_tx);
    if (sl.hdr_rx) (void)hipFree(sl.hdr_rx);
    if (sl.sortb) (void)hipFree(sl.sortb);
    if (sl.pinned) (void)hipHostFree(sl.pinned);
    if (sl.hostRows) (void)hipHostFree(sl.hostRows);
    for (hipEvent_t e : {sl.evDensity, sl.evB, sl.evBnd, sl.evForce, sl.evCopy, sl.evT[0], sl.evT[1], sl.evT[2],
                         sl.evTx[0], sl.evTx[1], sl.evRx[0], sl.evRx[1]})
        if (e) (void)hipEventDestroy(e);
    if (sl.comm) (void)hipStreamDestroy(sl.comm);
    if (sl.bnd) (void)hipStreamDestroy(sl.bnd);
    if (sl.copy) (void)hipStreamDestroy(sl.copy);
    sl = Slab{};
    return SPH_OK;
}

// (Re)build the slabs' buffers for capacity m->cap / face capacity m->F.
int alloc_slab(sph_mgpu *m, Slab &sl) {
    HIPM(m, hipSetDevice(sl.device));
    SphOptions o{};
    o.struct_size = (int32_t)sizeof o;
    o.device = sl.device;
    o.math_mode = m->opt.math_mode;
    o.sweep = m->opt.sweep;
    o.flags = SPH_FLAG_EXTERNAL_STATE | SPH_FLAG_NO_READBACK;
    o.capacity = m->cap;
    int rc = sph_create(&m->settings, &o, &sl.h);
    if (rc) return fail(m, rc, std::string("sph_create: ") + sph_last_error(nullptr));
    const size_t rows = (size_t)m->cap;
    for (int b = 0; b < 2; ++b) {
        HIPM(m, hipMalloc(&sl.pos[b], rows * sizeof(F4)));
        HIPM(m, hipMalloc(&sl.vel[b], rows * sizeof(F4)));
        HIPM(m, hipMemset(sl.pos[b], 0, rows * sizeof(F4)));
        HIPM(m, hipMemset(sl.vel[b], 0, rows * sizeof(F4)));
        HIPM(m, hipMalloc(&sl.rx_pos[b], (size_t)m->F * sizeof(F4)));
        HIPM(m, hipMalloc(&sl.rx_vel[b], (size_t)m->F * sizeof(F4)));
    }
    SPHM(m, sl, sph_bind_buffers(sl.h, sl.pos[0], sl.vel[0], sl.pos[1], sl.vel[1], m->cap));
    HIPM(m, hipMalloc(&sl.hdr_tx, sizeof(Hdr)));
    HIPM(m, hipMalloc(&sl.hdr_rx, 2 * sizeof(Hdr)));
    HIPM(m, hipMalloc(&sl.sortb, 8 * sizeof(int)));
    HIPM(m, hipMemset(sl.hdr_tx, 0, sizeof(Hdr)));
    HIPM(m, hipMemset(sl.hdr_rx, 0, 2 * sizeof(Hdr)));
    HIPM(m, hipHostMalloc(&sl.pinned, 32 * sizeof(int), hipHostMallocDefault));
    memset(sl.pinned, 0, 32 * sizeof(int));
    HIPM(m, hipHostMalloc(&sl.hostRows, rows * sizeof(F4), hipHostMallocDefault));
    for (hipEvent_t *e : {&sl.evDensity, &sl.evB, &sl.evBnd, &sl.evForce, &sl.evCopy, &sl.evTx[0], &sl.evTx[1],
                          &sl.evRx[0], &sl.evRx[1]})
        HIPM(m, hipEventCreateWithFlags(e, hipEventDisableTiming));
    for (auto &e : sl.evT) HIPM(m, hipEventCreate(&e));
    HIPM(m, hipStreamCreateWithFlags(&sl.copy, hipStreamNonBlocking));
    if (m->shared_stream) {
        SPHM(m, sl, sph_set_stream(sl.h, m->shared));
        sl.s = m->shared;
        sl.comm = nullptr;
    } else {
        sl.s = (hipStream_t)sph_get_stream(sl.h);
        HIPM(m, hipStreamCreateWithFlags(&sl.comm, hipStreamNonBlocking));
        HIPM(m, hipStreamCreateWithFlags(&sl.bnd, hipStreamNonBlocking));
    }
    HIPM(m, hipDeviceSynchronize());
    return SPH_OK;
}

// Hand the global row sequence `p4/v4` (n rows, any order that is the canonical
// sequence: particle-id order at step 0, rank-concatenated sorted order later) out to the
// slabs: stable filter by the z-layer of each row.
int distribute(sph_mgpu *m, const std::vector<F4> &p4, const std::vector<F4> &v4) {
    const int n = (int)p4.size();
    std::vector<long long> hist(m->D, 0);
    std::vector<int> lay(n);
    for (int i = 0; i < n; ++i) {
        lay[i] = layer_of(m, p4[i].z);
        hist[lay[i]]++;
    }
    const int world = m->opt.world;
    if (world * 2 > m->D) return fail(m, SPH_EINVAL, "too many slabs for the grid");
    m->cuts = partition_layers(hist, world, 2);
    std::vector<long long> per(world, 0);
    long long layerMax = 0;
    for (int z = 0; z < m->D; ++z) layerMax = std::max(layerMax, hist[z]);
    for (int r = 0; r < world; ++r)
        for (int z = m->cuts[r]; z < m->cuts[r + 1]; ++z) per[r] += hist[z];
    const long long biggest = *std::max_element(per.begin(), per.end());
    int cap = m->opt.slab_capacity > 0 ? m->opt.slab_capacity : (int)(biggest * 1.6) + 65536;
    int F = m->opt.face_capacity > 0 ? m->opt.face_capacity : (int)(1.25 * (double)layerMax) + 4096;
    F = std::min(F, cap);
    if (biggest + 2 * layerMax > cap && m->opt.slab_capacity > 0)
        return fail(m, SPH_EINVAL, "slab_capacity too small for the largest slab plus its halos");
    // buffers are only re-made when they have to grow (a re-cut keeps them)
    const bool fits = m->slabs[0].h && m->cap >= (int)(biggest * 1.3) + 2 * (int)layerMax &&
                      (m->opt.face_capacity > 0 || m->F >= (int)(1.1 * (double)layerMax)) &&
                      (m->opt.slab_capacity == 0 || m->cap == cap);
    if (!fits) {
        m->cap = cap;
        m->F = F;
        for (auto &sl : m->slabs) {
            const int rank = sl.rank, dev = sl.device;
            ncclComm_t c = sl.comm_nccl;
            free_slab(sl);
            sl.rank = rank;
            sl.device = dev;
            sl.comm_nccl = c;
            int rc = alloc_slab(m, sl);
            if (rc) return rc;
        }
    }
    set_geometry(m);
    std::vector<F4> sp, sv;
    for (auto &sl : m->slabs) {
        sp.clear();
        sv.clear();
        for (int i = 0; i < n; ++i)
            if (lay[i] >= sl.zlo && lay[i] < sl.zhi) {
                sp.push_back(p4[i]);
                sv.push_back(v4[i]);
            }
        if ((int)sp.size() > m->cap) return fail(m, SPH_EINVAL, "slab capacity too small");
        HIPM(m, hipSetDevice(sl.device));
        HIPM(m, hipStreamSynchronize(sl.s));
        HIPM(m, hipStreamSynchronize(sl.copy));
        if (!sp.empty()) {
            // through the slab's pinned read-back buffer (cap rows): a hipMemcpy from pageable
            // memory leaves a deferred unpin behind that stalls the first steps
            const size_t bytes = sp.size() * sizeof(F4);
            memcpy(sl.hostRows, sp.data(), bytes);
            HIPM(m, hipMemcpyAsync(sl.pos[0], sl.hostRows, bytes, hipMemcpyHostToDevice, sl.s));
            HIPM(m, hipStreamSynchronize(sl.s));
            memcpy(sl.hostRows, sv.data(), bytes);
            HIPM(m, hipMemcpyAsync(sl.vel[0], sl.hostRows, bytes, hipMemcpyHostToDevice, sl.s));
            HIPM(m, hipStreamSynchronize(sl.s));
        }
        sl.cur = 0;
        sl.off = 0;
        sl.n_own = (int)sp.size();
        sl.expectValid = false;
        sl.copyPending = false;
        sl.hostRowsCount = 0;
        sl.rowsStale = true; // hostRows served as the upload's staging buffer
        sl.status = 0;
    }
    m->hostPosValid = false;
    return SPH_OK;
}

// the rank-concatenated sequence of the LOCAL slabs' owned rows (device -> host)
int gather_local(sph_mgpu *m, std::vector<F4> &p4, std::vector<F4> &v4) {
    p4.clear();
    v4.clear();
    for (auto &sl : m->slabs) {
        HIPM(m, hipSetDevice(sl.device));
        HIPM(m, hipStreamSynchronize(sl.s));
        const size_t at = p4.size();
        p4.resize(at + sl.n_own);
        v4.resize(at + sl.n_own);
        if (sl.n_own) {
            HIPM(m, hipMemcpy(p4.data() + at, sl.pos[sl.cur] + sl.off, (size_t)sl.n_own * sizeof(F4), hipMemcpyDeviceToHost));
            HIPM(m, hipMemcpy(v4.data() + at, sl.vel[sl.cur] + sl.off, (size_t)sl.n_own * sizeof(F4), hipMemcpyDeviceToHost));
        }
    }
    return SPH_OK;
}

int upload_common(sph_mgpu *m, const float *pos, const float *vel, int n) {
    if (n != m->n) return fail(m, SPH_EINVAL, "particle count differs from settings");
    std::vector<F4> p4((size_t)n), v4((size_t)n);
    const float hh = m->settings.h;
    for (int i = 0; i < n; ++i) {
        const float x = pos[3 * i], y = pos[3 * i + 1], z = pos[3 * i + 2];
        // (range test before any float -> int conversion: that of a NaN / out-of-range value is undefined on the host)
        const float qx = x / hh, qy = y / hh, qz = z / hh, Df = (float)m->D;
        if (!(qx >= 0.f && qx < Df && qy >= 0.f && qy < Df && qz >= 0.f && qz < Df && x >= 0.f && y >= 0.f && z >= 0.f))
            return fail(m, SPH_EINVAL, "position outside the simulation box");
        uint32_t id = (uint32_t)i;
        float idbits;
        memcpy(&idbits, &id, 4);
        p4[i] = {x, y, z, idbits};
        v4[i] = vel ? F4{vel[3 * i], vel[3 * i + 1], vel[3 * i + 2], 0.f} : F4{0.f, 0.f, 0.f, 0.f};
    }
    m->phase = 0; // a fresh state also clears whatever a failed step left half-done
    m->poisoned = false;
    m->overflow = false;
    m->clickQueued = false;
    int rc = distribute(m, p4, v4);
    if (rc) return rc;
    m->ready = true;
    m->step = 0;
    return SPH_OK;
}

int recut(sph_mgpu *m) {
    if ((int)m->slabs.size() != m->opt.world) return SPH_OK; // needs the whole sequence in one process
    std::vector<F4> p4, v4;
    int rc = gather_local(m, p4, v4);
    if (rc) return rc;
    const std::vector<int> before = m->cuts;
    rc = distribute(m, p4, v4);
    if (rc) return rc;
    if (m->cuts != before) m->stats.recuts++;
    return SPH_OK;
}

} // namespace

extern "C" {

const char *sph_mgpu_last_error(const sph_mgpu *m) { return m ? m->err.c_str() : g_create_error.c_str(); }

int sph_mgpu_unique_id(void *out128) {
    if (!out128) return SPH_EINVAL;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return SPH_EHIP;
    static_assert(sizeof(id) == 128, "ncclUniqueId");
    memcpy(out128, &id, sizeof id);
    return SPH_OK;
}

int sph_mgpu_create(const SphSettings *settings, const SphMgpuOptions *options, const void *unique_id128,
                    sph_mgpu **out) {
    if (!settings || !options || !out) return fail(nullptr, SPH_EINVAL, "null argument");
    *out = nullptr;
    SphMgpuOptions o{};
    const size_t sz = options->struct_size > 0 ? (size_t)options->struct_size : sizeof o;
    memcpy(&o, options, std::min(sz, sizeof o));
    if (o.world < 1 || o.rank_count < 1 || o.rank_count > SPH_MGPU_MAX_LOCAL || o.rank_begin < 0 ||
        o.rank_begin + o.rank_count > o.world)
        return fail(nullptr, SPH_EINVAL, "bad world / rank range");
    if (o.transport < SPH_TRANSPORT_LOOPBACK || o.transport > SPH_TRANSPORT_STREAMS)
        return fail(nullptr, SPH_EINVAL, "unknown transport");
    if ((o.transport == SPH_TRANSPORT_LOOPBACK || o.transport == SPH_TRANSPORT_RCCL_SELF ||
         o.transport == SPH_TRANSPORT_STREAMS) && o.rank_count != o.world)
        return fail(nullptr, SPH_EINVAL, "loopback / self transports need every slab in this process");
    if (o.transport == SPH_TRANSPORT_MAILBOX && o.rank_count != 1)
        return fail(nullptr, SPH_EINVAL, "mailbox transport: one slab per driver object");
    if (o.transport == SPH_TRANSPORT_RCCL && o.rank_count != o.world && (o.rank_count != 1 || !unique_id128))
        return fail(nullptr, SPH_EINVAL, "one process per GPU: rank_count = 1 and a unique id");
    if (o.sweep != SPH_SWEEP_LIST && o.sweep != SPH_SWEEP_LDS && o.sweep != SPH_SWEEP_DIRECT)
        return fail(nullptr, SPH_EINVAL, "sweep variant not available in slab mode");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, SPH_ENODEV, "no HIP device: libsph_mgpu has no CPU fallback by design");
    sph_mgpu *m = new (std::nothrow) sph_mgpu();
    if (!m) return fail(nullptr, SPH_ENOMEM, "out of host memory");
    m->settings = *settings;
    m->opt = o;
    m->n = settings->numParticles;
    m->D = (int)settings->numCellsPerDim;
    m->DD = m->D * m->D;
    m->shared_stream = o.transport != SPH_TRANSPORT_RCCL && o.transport != SPH_TRANSPORT_STREAMS;
    m->slabs.resize(o.rank_count);
    for (int k = 0; k < o.rank_count; ++k) {
        m->slabs[k].rank = o.rank_begin + k;
        m->slabs[k].device = o.transport == SPH_TRANSPORT_RCCL ? o.devices[k] : o.devices[0];
        if (m->slabs[k].device < 0 || m->slabs[k].device >= ndev) {
            delete m;
            return fail(nullptr, SPH_EINVAL, "device ordinal out of range");
        }
    }
    int rc = SPH_OK;
    do {
        if (m->shared_stream) {
            if (hipSetDevice(m->slabs[0].device) != hipSuccess ||
                hipStreamCreateWithFlags(&m->shared, hipStreamNonBlocking) != hipSuccess) {
                rc = fail(nullptr, SPH_EHIP, "hipStreamCreate failed");
                break;
            }
        }
        if (o.transport == SPH_TRANSPORT_RCCL_SELF) {
            int dev = m->slabs[0].device;
            ncclComm_t c = nullptr;
            if (ncclCommInitAll(&c, 1, &dev) != ncclSuccess) { rc = fail(nullptr, SPH_EHIP, "ncclCommInitAll failed"); break; }
            for (auto &sl : m->slabs) sl.comm_nccl = c;
        } else if (o.transport == SPH_TRANSPORT_RCCL && o.rank_count == o.world) {
            std::vector<ncclComm_t> comms(o.world);
            std::vector<int> devs(o.world);
            for (int k = 0; k < o.world; ++k) devs[k] = m->slabs[k].device;
            if (ncclCommInitAll(comms.data(), o.world, devs.data()) != ncclSuccess) { rc = fail(nullptr, SPH_EHIP, "ncclCommInitAll failed"); break; }
            for (int k = 0; k < o.world; ++k) m->slabs[k].comm_nccl = comms[k];
        } else if (o.transport == SPH_TRANSPORT_RCCL) {
            ncclUniqueId id;
            memcpy(&id, unique_id128, sizeof id);
            if (hipSetDevice(m->slabs[0].device) != hipSuccess ||
                ncclCommInitRank(&m->slabs[0].comm_nccl, o.world, id, o.rank_begin) != ncclSuccess) {
                rc = fail(nullptr, SPH_EHIP, "ncclCommInitRank failed");
                break;
            }
        }
    } while (0);
    if (rc) {
        sph_mgpu_destroy(m);
        return rc;
    }
    if (const char *e = getenv("SPH_MGPU_THREADS"))
        if (atoi(e) != 0 && m->slabs.size() > 1) start_workers(m);
    *out = m;
    return SPH_OK;
}

void sph_mgpu_destroy(sph_mgpu *m) {
    if (!m) return;
    stop_workers(m);
    for (auto &sl : m->slabs) {
        (void)hipSetDevice(sl.device);
        if (sl.s) (void)hipStreamSynchronize(sl.s);
        if (sl.comm) (void)hipStreamSynchronize(sl.comm);
        if (sl.copy) (void)hipStreamSynchronize(sl.copy);
    }
    ncclComm_t last = nullptr;
    for (auto &sl : m->slabs) {
        if (sl.comm_nccl && sl.comm_nccl != last) {
            last = sl.comm_nccl;
            (void)ncclCommDestroy(sl.comm_nccl);
        }
        sl.comm_nccl = nullptr;
    }
    for (auto &sl : m->slabs) free_slab(sl);
    if (m->shared) (void)hipStreamDestroy(m->shared);
    delete m;
}

int sph_mgpu_setup(sph_mgpu *m) {
    if (!m) return SPH_EINVAL;
    std::vector<float> pos((size_t)std::max(m->n, 1) * 3, 0.f);
    int rc = sph_initial_positions(&m->settings, pos.data());
    if (rc) return fail(m, rc, "sph_initial_positions failed");
    return upload_common(m, pos.data(), nullptr, m->n);
}

int sph_mgpu_upload_state(sph_mgpu *m, const float *pos_xyz, const float *vel_xyz, int n) {
    if (!m || (!pos_xyz && n > 0)) return fail(m, SPH_EINVAL, "null argument");
    return upload_common(m, pos_xyz, vel_xyz, n);
}

} // extern "C"

namespace {

// A step in four phases; between two phases every message posted so far has been handed
// to the transport (RCCL / copies: at once; mailbox: completed at the start of the next).
int step_phase1(sph_mgpu *m, SphTimes *times) {
    const int F = m->F, DD = m->DD;
    m->t_begin = std::chrono::steady_clock::now();

    // ---- 1. partition the owned rows by the z-range of their NEW cell (no host round trip)
    int rcl = for_each_slab(m, [&](Slab &sl) -> int {
        HIPM(m, hipSetDevice(sl.device));
        if (times) HIPM(m, hipEventRecord(sl.evT[0], sl.s));
        const uint32_t thr[6] = {(uint32_t)(std::max(sl.zlo - 1, 0) * DD), (uint32_t)(sl.zlo * DD),
                                 (uint32_t)((sl.zlo + 1) * DD),            (uint32_t)((sl.zhi - 1) * DD),
                                 (uint32_t)(sl.zhi * DD),                  (uint32_t)((sl.zhi + 1) * DD)};
        SPHM(m, sl, sph_slab_partition_async(sl.h, sl.cur, sl.off, sl.n_own, thr, 6, sl.hdr_tx));
        // status word rides in the header: a rank that failed tells its neighbours
        sl.pinned[31] = sl.status;
        HIPM(m, hipMemcpyAsync(&sl.hdr_tx->status, &sl.pinned[31], sizeof(int), hipMemcpyHostToDevice, sl.s));
        HIPM(m, hipMemcpyAsync(sl.pinned, sl.hdr_tx, sizeof(Hdr), hipMemcpyDeviceToHost, sl.s));
        sl.sbuf = sl.cur ^ 1;
        return SPH_OK;
    });
    if (rcl) return rcl;
    // ---- 2. exchange A: header + fixed-size windows, one round
    {
        std::vector<Msg> msgs;
        for (int r = 0; r + 1 < m->opt.world; ++r) {
            Slab *lo = local(m, r), *hi = local(m, r + 1);
            if (!lo && !hi) continue;
            const size_t W = (size_t)F * sizeof(F4);
            // r -> r+1: the UP message of r lands in hi's slot [0] (from below)
            const int w0 = lo ? std::max(lo->n_own - F, 0) : 0;
            msgs.push_back({r, r + 1, lo ? (const void *)lo->hdr_tx : nullptr, hi ? (void *)&hi->hdr_rx[0] : nullptr, sizeof(Hdr)});
            msgs.push_back({r, r + 1, lo ? (const void *)(lo->pos[lo->sbuf] + w0) : nullptr, hi ? (void *)hi->rx_pos[0] : nullptr, W});
            msgs.push_back({r, r + 1, lo ? (const void *)(lo->vel[lo->sbuf] + w0) : nullptr, hi ? (void *)hi->rx_vel[0] : nullptr, W});
            // r+1 -> r: the DOWN message of r+1 lands in lo's slot [1] (from above)
            msgs.push_back({r + 1, r, hi ? (const void *)hi->hdr_tx : nullptr, lo ? (void *)&lo->hdr_rx[1] : nullptr, sizeof(Hdr)});
            msgs.push_back({r + 1, r, hi ? (const void *)hi->pos[hi->sbuf] : nullptr, lo ? (void *)lo->rx_pos[1] : nullptr, W});
            msgs.push_back({r + 1, r, hi ? (const void *)hi->vel[hi->sbuf] : nullptr, lo ? (void *)lo->rx_vel[1] : nullptr, W});
        }
        int rc = deliver(m, msgs, false);
        if (rc) return rc;
    }
    return SPH_OK;
}

int step_phase2(sph_mgpu *m) {
    const int F = m->F;
    int rc0 = resolve_mail(m);
    if (rc0) return rc0;
    // ---- 3. the step's ONE host synchronisation: own bounds + the neighbours' headers
    for (auto &sl : m->slabs) {
        HIPM(m, hipSetDevice(sl.device));
        HIPM(m, hipMemcpyAsync(sl.pinned + 8, sl.hdr_rx, 2 * sizeof(Hdr), hipMemcpyDeviceToHost, sl.s));
    }
    for (auto &sl : m->slabs) {
        HIPM(m, hipSetDevice(sl.device));
        HIPM(m, hipStreamSynchronize(sl.s));
    }
    m->stats.host_syncs++;
    bool overflow = false;
    m->overflow = false;
    for (auto &sl : m->slabs) {
        memcpy(&sl.mine, sl.pinned, sizeof(Hdr));
        memcpy(&sl.nb_dn, sl.pinned + 8, sizeof(Hdr));
        memcpy(&sl.nb_up, sl.pinned + 16, sizeof(Hdr));
        if (!sl.has_dn) sl.nb_dn = Hdr{};
        if (!sl.has_up) sl.nb_up = Hdr{};
        // last step's derived bounds against what the sort actually found
        if (sl.expectValid) {
            const int *got = sl.pinned + 24;
            for (int k = 0; k < 4; ++k)
                if (got[k] != sl.expect[k]) sl.status = 1;
            if (sl.status) {
                int rc = poison(m, SPH_ESTATE,
                                "slab " + std::to_string(sl.rank) +
                                    ": a particle crossed into a neighbour slab beyond its far boundary layer "
                                    "(or out of it) in one step: z-velocity too high for this decomposition");
                if (rc) return rc;
            }
        }
        if ((sl.has_dn && sl.nb_dn.status) || (sl.has_up && sl.nb_up.status)) {
            sl.status = 1;
            int rc = poison(m, SPH_ESTATE, "slab " + std::to_string(sl.rank) + ": a neighbour slab reported a failure");
            if (rc) return rc;
        }
        if (!sl.has_dn && sl.mine.b[1] != 0) { int rc = poison(m, SPH_ESTATE, "particles below the lowest slab"); if (rc) return rc; }
        if (!sl.has_up && sl.mine.b[4] != sl.mine.n) { int rc = poison(m, SPH_ESTATE, "particles above the highest slab"); if (rc) return rc; }
        // (a neighbour that has said farewell takes part in no further round)
        if (sl.has_dn && !sl.nb_dn.status && (down_layout(sl.mine, F).extra || up_layout(sl.nb_dn, F).extra)) overflow = true;
        if (sl.has_up && !sl.nb_up.status && (up_layout(sl.mine, F).extra || down_layout(sl.nb_up, F).extra)) overflow = true;
    }
    // ---- 3b. (rare) a face outgrew its fixed-size message: exact-size second round
    if (overflow) {
        m->overflow = true;
        m->stats.overflow_rounds++;
        std::vector<Msg> msgs;
        for (auto &sl : m->slabs) {
            HIPM(m, hipSetDevice(sl.device));
            if (sl.has_dn) { int rc = ensure_extra(m, sl, 0, up_layout(sl.nb_dn, F).extra); if (rc) return rc; }
            if (sl.has_up) { int rc = ensure_extra(m, sl, 1, down_layout(sl.nb_up, F).extra); if (rc) return rc; }
        }
        for (int r = 0; r + 1 < m->opt.world; ++r) {
            Slab *lo = local(m, r), *hi = local(m, r + 1);
            if (!lo && !hi) continue;
            if ((lo && lo->nb_up.status) || (hi && hi->nb_dn.status)) continue; // that neighbour is gone
            // UP excess of r: rows [b3, b3+extra) of its partitioned array -> hi.ex[0]
            const int exU = lo ? up_layout(lo->mine, F).extra : up_layout(hi->nb_dn, F).extra;
            if (exU) {
                const size_t B = (size_t)exU * sizeof(F4);
                const int at = lo ? lo->mine.b[3] : 0;
                msgs.push_back({r, r + 1, lo ? (const void *)(lo->pos[lo->sbuf] + at) : nullptr, hi ? (void *)hi->ex_pos[0] : nullptr, B});
                msgs.push_back({r, r + 1, lo ? (const void *)(lo->vel[lo->sbuf] + at) : nullptr, hi ? (void *)hi->ex_vel[0] : nullptr, B});
            }
            // DOWN excess of r+1: rows [F, b2) -> lo.ex[1]
            const int exD = hi ? down_layout(hi->mine, F).extra : down_layout(lo->nb_up, F).extra;
            if (exD) {
                const size_t B = (size_t)exD * sizeof(F4);
                msgs.push_back({r + 1, r, hi ? (const void *)(hi->pos[hi->sbuf] + F) : nullptr, lo ? (void *)lo->ex_pos[1] : nullptr, B});
                msgs.push_back({r + 1, r, hi ? (const void *)(hi->vel[hi->sbuf] + F) : nullptr, lo ? (void *)lo->ex_vel[1] : nullptr, B});
            }
        }
        int rc = deliver(m, msgs, false);
        if (rc) return rc;
    }
    return SPH_OK;
}

int step_phase3(sph_mgpu *m, SphTimes *times) {
    const int DD = m->DD;
    int rc0 = resolve_mail(m);
    if (rc0) return rc0;
    // ---- 4. assemble, sort, density
    int rcl = for_each_slab(m, [&](Slab &sl) -> int {
        HIPM(m, hipSetDevice(sl.device));
        const Hdr &me = sl.mine;
        const int m0 = me.b[1], m1 = me.b[2], m2 = me.b[3], m3 = me.b[4], n = me.n;
        const int bnd_from_dn = sl.has_dn ? sl.nb_dn.b[4] - sl.nb_dn.b[3] : 0;
        const int mig_from_dn = sl.has_dn ? sl.nb_dn.n - sl.nb_dn.b[4] : 0;
        const int near_from_dn = sl.has_dn ? sl.nb_dn.b[5] - sl.nb_dn.b[4] : 0;
        const int mig_from_up = sl.has_up ? sl.nb_up.b[1] : 0;
        const int bnd_from_up = sl.has_up ? sl.nb_up.b[2] - sl.nb_up.b[1] : 0;
        const int near_from_up = sl.has_up ? sl.nb_up.b[1] - sl.nb_up.b[0] : 0;
        const int counts[7] = {bnd_from_dn, m0, mig_from_dn, m3 - m0, mig_from_up, n - m3, bnd_from_up};
        int o[8] = {0};
        for (int k = 0; k < 7; ++k) o[k + 1] = o[k] + counts[k];
        sl.n_comb = o[7];
        sl.i0 = bnd_from_dn + m0;
        sl.i1 = sl.n_comb - (bnd_from_up + (n - m3));
        sl.e_lo = sl.i0 + near_from_dn + (m1 - m0);
        sl.s_hi = sl.i1 - (near_from_up + (m3 - m2));
        if (sl.n_comb > m->cap) {
            int rc = poison(m, SPH_ESTATE, "slab capacity exceeded by halo + migrants");
            if (rc) return rc;
        }
        if (sl.i0 > sl.e_lo || sl.e_lo > sl.s_hi || sl.s_hi > sl.i1) {
            int rc = poison(m, SPH_ESTATE, "slab " + std::to_string(sl.rank) + ": inconsistent exchange headers");
            if (rc) return rc;
        }
        if (m->poisoned) return SPH_OK; // the state is lost; only the step's remaining messages matter
        const int s = sl.sbuf, t = s ^ 1;
        std::vector<Piece> pieces;
        std::vector<int> dst;
        auto put_local = [&](int a, int b, int at) {
            if (b > a) { pieces.push_back({sl.pos[s] + a, sl.vel[s] + a, b - a}); dst.push_back(at); }
        };
        auto put_rx = [&](int which, int a, int b, int at) {
            const size_t before = pieces.size();
            payload_rows(m, sl, which, a, b, pieces);
            for (size_t k = before; k < pieces.size(); ++k) { dst.push_back(at); at += pieces[k].count; }
        };
        put_rx(0, 0, bnd_from_dn, o[0]);                          // from below: its upper boundary layer
        put_local(0, m0, o[1]);                                   // my migrants down
        put_rx(0, bnd_from_dn, bnd_from_dn + mig_from_dn, o[2]);  // from below: its migrants up
        put_local(m0, m3, o[3]);                                  // what stays mine
        put_rx(1, 0, mig_from_up, o[4]);                          // from above: its migrants down
        put_local(m3, n, o[5]);                                   // my migrants up
        put_rx(1, mig_from_up, mig_from_up + bnd_from_up, o[6]);  // from above: its lower boundary layer
        if (sl.copyPending) { // the read-back of the last step still reads buffer t
            HIPM(m, hipStreamWaitEvent(sl.s, sl.evCopy, 0));
            sl.copyPending = false;
        }
        for (size_t a = 0; a < pieces.size(); a += 8) {
            const int k = (int)std::min<size_t>(8, pieces.size() - a);
            const void *sp[8], *sv[8];
            int32_t cnt[8], at[8];
            for (int q = 0; q < k; ++q) {
                sp[q] = pieces[a + q].p;
                sv[q] = pieces[a + q].v;
                cnt[q] = pieces[a + q].count;
                at[q] = dst[a + q];
            }
            SPHM(m, sl, sph_slab_copy_segments(sl.h, t, k, sp, sv, cnt, at));
        }
        const uint32_t thr[4] = {(uint32_t)(sl.zlo * DD), (uint32_t)((sl.zlo + 1) * DD),
                                 (uint32_t)((sl.zhi - 1) * DD), (uint32_t)(sl.zhi * DD)};
        SPHM(m, sl, sph_slab_sort_async(sl.h, t, 0, sl.n_comb, thr, 4, sl.sortb));
        HIPM(m, hipMemcpyAsync(sl.pinned + 24, sl.sortb, 4 * sizeof(int), hipMemcpyDeviceToHost, sl.s));
        sl.expect[0] = sl.i0;
        sl.expect[1] = sl.e_lo;
        sl.expect[2] = sl.s_hi;
        sl.expect[3] = sl.i1;
        sl.expectValid = true;
        sl.sbuf = t ^ 1; // the sorted streams
        if (times) HIPM(m, hipEventRecord(sl.evT[1], sl.s));
        SPHM(m, sl, sph_slab_density(sl.h, sl.sbuf, sl.i0, sl.i1, sl.n_comb));
        if (sl.comm) HIPM(m, hipEventRecord(sl.evDensity, sl.s));
        return SPH_OK;
    });
    if (rcl) return rcl;
    // ---- 5. exchange B (rho of the boundary layers, rides in vel4.w) || interior force sweep
    {
        std::vector<Msg> msgs;
        for (int r = 0; r + 1 < m->opt.world; ++r) {
            Slab *lo = local(m, r), *hi = local(m, r + 1);
            if (!lo && !hi) continue;
            // lo's upper boundary layer [s_hi, i1) -> the last rows of hi's lower halo [i0-c, i0)
            const int cU = lo ? lo->i1 - lo->s_hi
                              : (hi->nb_dn.b[4] - hi->nb_dn.b[3]) + (hi->mine.b[1] - hi->mine.b[0]);
            // hi's lower boundary layer [i0, e_lo) -> the first rows of lo's upper halo [i1, i1+c)
            const int cD = hi ? hi->e_lo - hi->i0
                              : (lo->nb_up.b[2] - lo->nb_up.b[1]) + (lo->mine.b[5] - lo->mine.b[4]);
            if (lo && hi) {
                // both ends in this process: the two derivations must agree
                const int cU2 = (hi->nb_dn.b[4] - hi->nb_dn.b[3]) + (hi->mine.b[1] - hi->mine.b[0]);
                const int cD2 = (lo->nb_up.b[2] - lo->nb_up.b[1]) + (lo->mine.b[5] - lo->mine.b[4]);
                if (cU != cU2 || cD != cD2) return fail(m, SPH_ESTATE, "exchange B plans disagree");
            }
            if ((lo && lo->nb_up.status) || (hi && hi->nb_dn.status)) continue; // that neighbour has said farewell
            if (m->poisoned) {
                // sizes as the headers dictate (the healthy neighbour posts the matching calls), payload and
                // destination anywhere inside the buffers: nothing will read them
                const size_t rowB = (m->opt.sweep == SPH_SWEEP_LIST ? 2 : 1) * sizeof(F4);
                if (cU < 0 || cD < 0 || cU > m->cap || cD > m->cap) return fail(m, SPH_ESTATE, "failed step: exchange sizes out of range");
                F4 *bl = lo ? (m->opt.sweep == SPH_SWEEP_LIST ? static_cast<F4 *>(sph_slab_records(lo->h)) : lo->vel[0]) : nullptr;
                F4 *bh = hi ? (m->opt.sweep == SPH_SWEEP_LIST ? static_cast<F4 *>(sph_slab_records(hi->h)) : hi->vel[0]) : nullptr;
                msgs.push_back({r, r + 1, bl, bh, (size_t)cU * rowB});
                msgs.push_back({r + 1, r, bh, bl, (size_t)cD * rowB});
                continue;
            }
            if (m->opt.sweep == SPH_SWEEP_LIST) {
                // the list sweeps read neighbours from the interleaved (pos4, vel4) records, into which the
                // density sweep wrote rho: a boundary layer's records go straight into the neighbour's halo
                // rows (twice the bytes of vel4 alone -- a few hundred KB -- and no patch launch afterwards)
                F4 *rl = lo ? static_cast<F4 *>(sph_slab_records(lo->h)) : nullptr;
                F4 *rh = hi ? static_cast<F4 *>(sph_slab_records(hi->h)) : nullptr;
                msgs.push_back({r, r + 1, lo ? (const void *)(rl + 2 * (size_t)lo->s_hi) : nullptr,
                                hi ? (void *)(rh + 2 * (size_t)(hi->i0 - cU)) : nullptr, (size_t)cU * 2 * sizeof(F4)});
                msgs.push_back({r + 1, r, hi ? (const void *)(rh + 2 * (size_t)hi->i0) : nullptr,
                                lo ? (void *)(rl + 2 * (size_t)lo->i1) : nullptr, (size_t)cD * 2 * sizeof(F4)});
                continue;
            }
            msgs.push_back({r, r + 1, lo ? (const void *)(lo->vel[lo->sbuf] + lo->s_hi) : nullptr,
                            hi ? (void *)(hi->vel[hi->sbuf] + hi->i0 - cU) : nullptr, (size_t)cU * sizeof(F4)});
            msgs.push_back({r + 1, r, hi ? (const void *)(hi->vel[hi->sbuf] + hi->i0) : nullptr,
                            lo ? (void *)(lo->vel[lo->sbuf] + lo->i1) : nullptr, (size_t)cD * sizeof(F4)});
        }
        for (auto &sl : m->slabs)
            if (sl.comm) {
                HIPM(m, hipSetDevice(sl.device));
                if (m->poisoned) HIPM(m, hipEventRecord(sl.evDensity, sl.s)); // (no density sweep recorded it)
                HIPM(m, hipStreamWaitEvent(sl.comm, sl.evDensity, 0));
            }
        int rc = deliver(m, msgs, true);
        if (rc) return rc;
        for (auto &sl : m->slabs)
            if (sl.comm) { HIPM(m, hipSetDevice(sl.device)); HIPM(m, hipEventRecord(sl.evB, sl.comm)); }
    }
    return SPH_OK;
}

// The farewell of a poisoned driver: one more exchange A (header with status = 1 + the fixed-size windows)
// with every neighbour in another process that has not said farewell itself -- exactly the calls that
// neighbour posts in phase 1 of its next step -- then the error of the failed check.
int farewell(sph_mgpu *m) {
    const int F = m->F;
    const size_t W = (size_t)F * sizeof(F4);
    std::vector<Msg> msgs;
    for (auto &sl : m->slabs) {
        HIPM(m, hipSetDevice(sl.device));
        if (sl.bnd) HIPM(m, hipStreamSynchronize(sl.bnd));
        if (sl.comm) HIPM(m, hipStreamSynchronize(sl.comm)); // exchange B of this step is through
        Hdr bye{};
        bye.status = 1;
        memcpy(sl.pinned, &bye, sizeof(Hdr));
        HIPM(m, hipMemcpyAsync(sl.hdr_tx, sl.pinned, sizeof(Hdr), hipMemcpyHostToDevice, sl.s));
    }
    for (int r = 0; r + 1 < m->opt.world; ++r) {
        Slab *lo = local(m, r), *hi = local(m, r + 1);
        if ((lo != nullptr) == (hi != nullptr)) continue; // both here (stopping together) or both elsewhere
        if ((lo && lo->nb_up.status) || (hi && hi->nb_dn.status)) continue; // gone already
        msgs.push_back({r, r + 1, lo ? (const void *)lo->hdr_tx : nullptr, hi ? (void *)&hi->hdr_rx[0] : nullptr, sizeof(Hdr)});
        msgs.push_back({r, r + 1, lo ? (const void *)lo->pos[0] : nullptr, hi ? (void *)hi->rx_pos[0] : nullptr, W});
        msgs.push_back({r, r + 1, lo ? (const void *)lo->vel[0] : nullptr, hi ? (void *)hi->rx_vel[0] : nullptr, W});
        msgs.push_back({r + 1, r, hi ? (const void *)hi->hdr_tx : nullptr, lo ? (void *)&lo->hdr_rx[1] : nullptr, sizeof(Hdr)});
        msgs.push_back({r + 1, r, hi ? (const void *)hi->pos[0] : nullptr, lo ? (void *)lo->rx_pos[1] : nullptr, W});
        msgs.push_back({r + 1, r, hi ? (const void *)hi->vel[0] : nullptr, lo ? (void *)lo->rx_vel[1] : nullptr, W});
    }
    int rc = deliver(m, msgs, false);
    if (rc) return rc;
    if (m->opt.transport != SPH_TRANSPORT_MAILBOX)
        for (auto &sl : m->slabs) {
            HIPM(m, hipSetDevice(sl.device));
            HIPM(m, hipStreamSynchronize(sl.s));
        }
    return m->poisonCode; // (m->err still holds the message of the check that failed)
}

int step_phase4(sph_mgpu *m, SphTimes *times) {
    int rc0 = resolve_mail(m);
    if (rc0) return rc0;
    if (m->poisoned) return farewell(m);
    int rcl = for_each_slab(m, [&](Slab &sl) -> int {
        HIPM(m, hipSetDevice(sl.device));
        const int a = sl.has_dn ? sl.e_lo : sl.i0, b = sl.has_up ? sl.s_hi : sl.i1;
        if (m->shared_stream) {
            // loopback / self transport: exchange B was delivered in stream order, there is
            // nothing to overlap -- the whole slab in ONE launch (one tail instead of two)
            if (m->opt.sweep != SPH_SWEEP_LIST) // (list: exchange B delivered whole records)
                SPHM(m, sl, sph_slab_patch_halo(sl.h, sl.sbuf, sl.i0, sl.i1, sl.n_comb, nullptr));
            SPHM(m, sl, sph_slab_force_ranges(sl.h, sl.sbuf, sl.i0, sl.i0, sl.i1, 0, 0, sl.n_comb, 1, nullptr));
        } else {
        // interior layers: every neighbour is an owned row -> no need to wait for exchange B
        SPHM(m, sl, sph_slab_force_ranges(sl.h, sl.sbuf, sl.i0, a, b, 0, 0, sl.n_comb, 0, nullptr));
        // the two boundary layers in ONE launch once the halo densities are in: on a stream
        // of their own they join the interior's launch on the GPU instead of waiting for its
        // tail (with a shared stream -- loopback -- they simply follow it)
        hipStream_t bs = sl.bnd ? sl.bnd : sl.s;
        if (sl.bnd) HIPM(m, hipStreamWaitEvent(sl.bnd, sl.evB, 0));
        if (m->opt.sweep != SPH_SWEEP_LIST) SPHM(m, sl, sph_slab_patch_halo(sl.h, sl.sbuf, sl.i0, sl.i1, sl.n_comb, bs));
        SPHM(m, sl, sph_slab_force_ranges(sl.h, sl.sbuf, sl.i0, sl.i0, a, b, sl.i1, sl.n_comb, 1, bs));
        if (sl.bnd) {
            HIPM(m, hipEventRecord(sl.evBnd, sl.bnd));
            HIPM(m, hipStreamWaitEvent(sl.s, sl.evBnd, 0));
        }
        }
        sl.cur = sl.sbuf ^ 1;
        sl.off = sl.i0;
        sl.n_own = sl.i1 - sl.i0;
        // the click impulse of Simulator::simulate (simulator.cu:482-489) on the layers this slab owns:
        // new velocities, this step's (pre-integration) grid -- every row of an owned layer is an owned row
        if (m->clickQueued) SPHM(m, sl, sph_slab_apply_click(sl.h, sl.cur, m->clickX, m->clickY, sl.zlo, sl.zhi));
        if (times) HIPM(m, hipEventRecord(sl.evT[2], sl.s));
        // ---- 6. position read-back of the owned rows (simulator.cu:479-480), off the compute stream
        HIPM(m, hipEventRecord(sl.evForce, sl.s));
        HIPM(m, hipStreamWaitEvent(sl.copy, sl.evForce, 0));
        if (sl.n_own)
            HIPM(m, hipMemcpyAsync(sl.hostRows, sl.pos[sl.cur] + sl.off, (size_t)sl.n_own * sizeof(F4),
                                   hipMemcpyDeviceToHost, sl.copy));
        HIPM(m, hipEventRecord(sl.evCopy, sl.copy));
        sl.hostRowsCount = sl.n_own;
        sl.rowsStale = false;
        sl.copyPending = true;
        return SPH_OK;
    });
    if (rcl) return rcl;
    m->clickQueued = false;
    m->hostPosValid = false;
    m->step++;
    m->stats.steps++;
    if (times) {
        double grid = 0, sphu = 0;
        for (auto &sl : m->slabs) {
            HIPM(m, hipSetDevice(sl.device));
            HIPM(m, hipEventSynchronize(sl.evT[2]));
            float g = 0, u = 0;
            HIPM(m, hipEventElapsedTime(&g, sl.evT[0], sl.evT[1]));
            HIPM(m, hipEventElapsedTime(&u, sl.evT[1], sl.evT[2]));
            grid = std::max(grid, (double)g * 1e-3);
            sphu = std::max(sphu, (double)u * 1e-3);
        }
        times->buildGrid += grid;
        times->sphUpdate += sphu;
        const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - m->t_begin).count();
        times->memcpy += std::max(0.0, wall - grid - sphu); // host-visible rest: exchange waits + sync
        times->iters += 1;
    }
    if (m->opt.recut_every > 0 && m->step % m->opt.recut_every == 0) {
        int rc = recut(m);
        if (rc) return rc;
    }
    return SPH_OK;
}

} // namespace

extern "C" {

int sph_mgpu_step_phase(sph_mgpu *m, int phase, SphTimes *times) {
    if (!m) return SPH_EINVAL;
    if (!m->ready) return fail(m, SPH_ESTATE, "setup()/upload_state() must come first");
    if (m->poisoned && m->phase == 0) return m->poisonCode; // (err holds the message; upload_state()/setup() starts over)
    if (phase < 1 || phase > 4 || phase != m->phase + 1) return fail(m, SPH_ESTATE, "step phases run 1, 2, 3, 4");
    int rc = phase == 1 ? step_phase1(m, times) : phase == 2 ? step_phase2(m)
             : phase == 3 ? step_phase3(m, times) : step_phase4(m, times);
    if (rc) {
        if (m->poisoned && phase == 4) m->phase = 0; // farewell sent: every later step reports the same failure
        return rc;
    }
    m->phase = phase == 4 ? 0 : phase;
    return SPH_OK;
}

int sph_mgpu_step(sph_mgpu *m, SphTimes *times) {
    if (!m) return SPH_EINVAL;
    if (m->opt.transport == SPH_TRANSPORT_MAILBOX)
        return fail(m, SPH_ESTATE, "mailbox transport: drive every rank's object with sph_mgpu_step_phase");
    for (int ph = 1; ph <= 4; ++ph) {
        int rc = sph_mgpu_step_phase(m, ph, times);
        if (rc) return rc;
    }
    return SPH_OK;
}

int sph_mgpu_queue_click(sph_mgpu *m, int mouse_x, int mouse_y) {
    if (!m) return SPH_EINVAL;
    if (m->opt.sweep == SPH_SWEEP_LINKED) return fail(m, SPH_ESTATE, "the click impulse is not available with SPH_SWEEP_LINKED");
    if (m->phase != 0) return fail(m, SPH_ESTATE, "queue the click between steps");
    m->clickQueued = true;
    m->clickX = mouse_x;
    m->clickY = mouse_y;
    return SPH_OK;
}

int sph_mgpu_sync(sph_mgpu *m) {
    if (!m) return SPH_EINVAL;
    for (auto &sl : m->slabs) {
        HIPM(m, hipSetDevice(sl.device));
        HIPM(m, hipStreamSynchronize(sl.s));
        if (sl.comm) HIPM(m, hipStreamSynchronize(sl.comm));
        HIPM(m, hipStreamSynchronize(sl.copy));
        if (sl.expectValid) { // the last step's bounds check, now that its copy has landed
            const int *got = sl.pinned + 24;
            for (int k = 0; k < 4; ++k)
                if (got[k] != sl.expect[k]) {
                    sl.status = 1;
                    return fail(m, SPH_ESTATE, "slab " + std::to_string(sl.rank) +
                                                   ": a particle crossed more layers in z than this decomposition allows");
                }
        }
    }
    return SPH_OK;
}

const float *sph_mgpu_positions_host(sph_mgpu *m) {
    if (!m) return nullptr;
    if (m->hostPos.size() != (size_t)m->n * 3) m->hostPos.assign((size_t)m->n * 3, 0.f);
    if (m->hostPosValid) return m->hostPos.data();
    for (auto &sl : m->slabs) {
        if (hipSetDevice(sl.device) != hipSuccess || hipStreamSynchronize(sl.copy) != hipSuccess) {
            m->err = "stream synchronize failed";
            return nullptr;
        }
        if (sl.rowsStale) { // before the first step, or right after a re-cut: the uploaded state itself
            if (hipStreamSynchronize(sl.s) != hipSuccess) {
                m->err = "stream synchronize failed";
                return nullptr;
            }
            if (sl.n_own && hipMemcpy(sl.hostRows, sl.pos[sl.cur] + sl.off, (size_t)sl.n_own * sizeof(F4),
                                      hipMemcpyDeviceToHost) != hipSuccess) {
                m->err = "hipMemcpy failed";
                return nullptr;
            }
            sl.hostRowsCount = sl.n_own;
            sl.rowsStale = false;
        }
        for (int i = 0; i < sl.hostRowsCount; ++i) {
            const F4 &p = sl.hostRows[i];
            uint32_t id;
            memcpy(&id, &p.w, 4);
            if (id >= (uint32_t)m->n) {
                m->err = "corrupt particle id in device state";
                return nullptr;
            }
            float *o = m->hostPos.data() + 3 * (size_t)id;
            o[0] = p.x;
            o[1] = p.y;
            o[2] = p.z;
        }
    }
    m->hostPosValid = true;
    return m->hostPos.data();
}

int sph_mgpu_download_state(sph_mgpu *m, float *pos, float *vel, float *rho, int *written) {
    if (!m) return SPH_EINVAL;
    int rc = sph_mgpu_sync(m);
    if (rc) return rc;
    std::vector<F4> p4, v4;
    if ((rc = gather_local(m, p4, v4))) return rc;
    for (size_t i = 0; i < p4.size(); ++i) {
        uint32_t id;
        memcpy(&id, &p4[i].w, 4);
        if (id >= (uint32_t)m->n) return fail(m, SPH_EHIP, "corrupt particle id in device state");
        if (pos) { pos[3 * id] = p4[i].x; pos[3 * id + 1] = p4[i].y; pos[3 * id + 2] = p4[i].z; }
        if (vel) { vel[3 * id] = v4[i].x; vel[3 * id + 1] = v4[i].y; vel[3 * id + 2] = v4[i].z; }
        if (rho) rho[id] = v4[i].w;
    }
    if (written) *written = (int)p4.size();
    return SPH_OK;
}

int sph_mgpu_get_stats(sph_mgpu *m, SphMgpuStats *out, int reset) {
    if (!m || !out) return SPH_EINVAL;
    int rc = sph_mgpu_sync(m);
    if (rc) return rc;
    m->stats.local_slabs = (int)m->slabs.size();
    for (size_t k = 0; k < m->slabs.size(); ++k) {
        Slab &sl = m->slabs[k];
        HIPM(m, hipSetDevice(sl.device));
        SphKernelTimes kt{};
        SPHM(m, sl, sph_get_kernel_times(sl.h, &kt, reset));
        m->stats.owned[k] = sl.n_own;
        m->stats.grid_s[k] = kt.hash + kt.sort + kt.gather;
        m->stats.density_s[k] = kt.density;
        m->stats.force_s[k] = kt.force;
        m->stats.kernel_s[k] = m->stats.grid_s[k] + kt.density + kt.force;
    }
    *out = m->stats;
    if (reset) {
        const int ls = m->stats.local_slabs;
        m->stats = SphMgpuStats{};
        m->stats.local_slabs = ls;
    }
    return SPH_OK;
}

} // extern "C"
