// Cross-process batching service (include/pnn_service.h): Unix-domain socket server that coalesces the single-block
// PNN requests of many encoder processes into batched calls, and the matching client stub.  Plain POSIX, no HIP here.
//
// The server never blocks on a client: one thread does all socket work with non-blocking sockets and per-client receive /
// send buffers (a request is queued for the GPU only once its last byte has arrived; a client whose reply cannot be written
// for kStallMs or that sends a malformed header is dropped without disturbing the others), worker threads do the backend
// calls -- one per width when the server owns its contexts (pnn_service_run_table), so that passes of different widths
// overlap each other and the socket work.
#include "pnn_service.h"

#include <errno.h>
#include <fcntl.h>
#include <linux/futex.h>
#include <poll.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <sys/epoll.h>
#include <sys/eventfd.h>
#include <sys/prctl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/un.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr uint32_t kMagic = 0x324e4e50u;   // "PNN2"
constexpr uint32_t kWantF32 = 1u;          // flags bit 0: reply with the float prediction (frozen-graph output) instead of Pel
constexpr uint32_t kWantTag = 2u;          // flags bit 1: no inputs; reply = the arithmetic tag of the context that serves `width` (pnn_arithmetic_tag)
constexpr size_t kTagBytes = 256;          // a tag reply's payload: the string, zero-padded
constexpr long kStallMs = 5000;            // a reply that cannot be delivered for this long drops its client
constexpr uint32_t kWantShm = 4u;          // flags bit 2: no inputs; the client asks for the shared-memory request path (see ShmSlot)
struct ReqHeader { uint32_t magic; int32_t width; uint32_t n_above, n_left, flags; };   // followed by the floats
struct RspHeader { int32_t rc; uint32_t n_vals; };                                       // followed by n_vals int32 / float

// ---- the shared-memory request path (round 6) --------------------------------------------------------------------------------------
// Over the socket a request crosses four thread wake-ups and ten system calls (client send / recv, I/O thread epoll + recv, worker
// wake-up, eventfd, I/O thread epoll + read + send: tools/service_load.cpp) and is copied four times before the first kernel sees it;
// round 5's campaigns spent 19-27 CPU-seconds in the service and 95-120 us per 4x4 request inside it for a 40 us call.  Now the socket
// is the CONTROL channel only (connect, hand-shake, teardown: its closing is how the server learns that a client died), and every
// client owns ONE request slot in a memfd both sides map (one outstanding request per client -- HM's calling pattern):
//   client:  writes header + context into its slot, state = kPosted, rings the doorbell of the width's worker (a futex word in a page
//            all clients share; a system call only if that worker sleeps), then sleeps on its slot's state word;
//   worker:  scans the slots for posted requests of its widths (kPosted -> kTaken by compare-and-swap), copies the contexts ONCE into
//            the batch's staging, runs the backend, writes reply header + values into each slot, state = kReady, wakes the client.
// Two wake-ups and two or three system calls per request; the I/O threads shrink to accept / hand-shake / teardown (and still serve
// clients that speak the socket protocol: $PNN_SERVICE_SHM=0 in a client, the tests' raw sockets).  A client that dies with a request
// in its slot: the control socket closes, the slot is unlisted, a worker that already holds it writes its reply into memory nobody
// reads (the mapping lives as long as the last reference).  The server validates the header it reads from a slot exactly as one from a
// socket: what a client scribbles into its own slot can only spoil its own reply.
enum : uint32_t { kSlotIdle = 0, kSlotPosted = 1, kSlotTaken = 2, kSlotReady = 3 };
constexpr size_t kShmInFloats = (size_t)5 * 64 * 64, kShmOutVals = (size_t)64 * 64;    // the largest request: a 64x64 block
struct ShmSlot {
    std::atomic<uint32_t> state;                      // the CLIENT's futex word
    std::atomic<uint32_t> client_sleeps;              // the client is (about to be) asleep on `state`: the server wakes it
    ReqHeader hdr;                                    // client, before state = kSlotPosted
    uint64_t posted_ns;                               // CLOCK_MONOTONIC at the post (the server's queueing statistics)
    alignas(64) RspHeader rsp;                        // server, before state = kSlotReady
    alignas(64) float in[kShmInFloats];               // [n_above | n_left]
    alignas(64) uint32_t out[kShmOutVals];            // int32 Pel or float, as the request asked
};
struct ShmDoor { alignas(64) std::atomic<uint32_t> bell; std::atomic<uint32_t> sleeping; };   // a worker's futex word + "I am (about to be) asleep"
// ... and one BIT per (worker, slot): set by a client behind its post (fetch_or), claimed by the worker that looks (exchange of the
// whole word) -- a worker finds the posted slots in ITS 16 words instead of touching one cache line per connected client (a
// 100-encoder campaign holds 500 slots); the slot's state word stays the truth: a stale bit costs one look, a bit that was never set
// belongs to a client that died.
constexpr int kShmMaxSlots = 1024;
// spin_us: the server's hint to both sides of a request -- with FEW clients connected (a couple of encoders: their waits are what their
// wall clock is made of, and CPUs are idle) a client polls its slot, and an idle worker its posted bits, for that long before going to
// sleep on the futex: a wake-up through the kernel is 5-10 us on each side of a 35-45 us call.  0 with many clients (a campaign: every CPU
// is taken, and a polling client would take one from an encoder that has work).
struct ShmDoors { ShmDoor door[5]; alignas(64) std::atomic<uint64_t> posted[5][kShmMaxSlots / 64]; alignas(64) std::atomic<uint32_t> spin_us; };
constexpr size_t kSlotBytes = (sizeof(ShmSlot) + 4095) / 4096 * 4096, kDoorBytes = (sizeof(ShmDoors) + 4095) / 4096 * 4096;
static_assert(std::atomic<uint32_t>::is_always_lock_free && std::atomic<uint64_t>::is_always_lock_free, "futex words and flags in shared memory");

long futex_wait(std::atomic<uint32_t>* addr, uint32_t expect, long timeout_us)
{
    timespec ts{timeout_us / 1000000, (timeout_us % 1000000) * 1000};
    return syscall(SYS_futex, reinterpret_cast<uint32_t*>(addr), FUTEX_WAIT, expect, &ts, nullptr, 0);   // (not _PRIVATE: the word is shared between processes)
}
long futex_wake(std::atomic<uint32_t>* addr, int n) { return syscall(SYS_futex, reinterpret_cast<uint32_t*>(addr), FUTEX_WAKE, n, nullptr, nullptr, 0); }
uint64_t now_ns() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (uint64_t)t.tv_sec * 1000000000ull + (uint64_t)t.tv_nsec; }

using Clock = std::chrono::steady_clock;

bool read_all(int fd, void* buf, size_t n)      // client side only (blocking socket)
{
    char* p = static_cast<char*>(buf);
    while (n) {
        const ssize_t r = recv(fd, p, n, 0);
        if (r == 0) return false;
        if (r < 0) { if (errno == EINTR) continue; return false; }
        p += r; n -= (size_t)r;
    }
    return true;
}

bool write_all(int fd, const void* buf, size_t n)   // client side only
{
    const char* p = static_cast<const char*>(buf);
    while (n) {
        const ssize_t r = send(fd, p, n, MSG_NOSIGNAL);
        if (r < 0) { if (errno == EINTR) continue; return false; }
        p += r; n -= (size_t)r;
    }
    return true;
}

bool valid_width(int w) { return w == 4 || w == 8 || w == 16 || w == 32 || w == 64; }

bool valid_header(const ReqHeader& h)
{
    if (h.magic != kMagic || !valid_width(h.width) || (h.flags & ~(kWantF32 | kWantTag | kWantShm))) return false;
    if (h.flags & (kWantTag | kWantShm)) return (h.flags == kWantTag || h.flags == kWantShm) && h.n_above == 0 && h.n_left == 0;
    const uint32_t w2 = (uint32_t)(h.width * h.width);
    return (h.n_above == 5 * w2 && h.n_left == 0) || (h.n_above == 3 * w2 && h.n_left == 2 * w2);
}

struct ShmClient {                                     // server side of one client's slot; shared by the registry, the owning I/O thread and the requests in flight
    ShmSlot* slot = nullptr;
    int index = -1;                                   // its byte in every worker's flag array
    std::atomic<bool> dead{false};
    ~ShmClient() { if (slot) munmap(slot, kSlotBytes); }
};

struct Client {
    int fd = -1;
    std::shared_ptr<ShmClient> shm;  // set by the hand-shake: this client's requests arrive through its slot
    int pid = 0;                     // peer process (SO_PEERCRED): an encoder holds one connection per session, one request at a time
    std::vector<char> rx;            // bytes of the request being received
    std::vector<char> tx;            // reply bytes not yet accepted by the socket
    size_t tx_off = 0;
    Clock::time_point tx_since;      // when tx became non-empty
    bool in_flight = false;          // a complete request of this client is queued or being computed
};


int make_addr(const char* path, sockaddr_un* a)
{
    memset(a, 0, sizeof *a);
    a->sun_family = AF_UNIX;
    if (!path || strlen(path) >= sizeof a->sun_path) return PNN_E_ARG;
    strcpy(a->sun_path, path);
    return PNN_OK;
}

void set_nonblocking(int fd)
{
    const int fl = fcntl(fd, F_GETFL, 0);
    if (fl >= 0) fcntl(fd, F_SETFL, fl | O_NONBLOCK);
}

// Input sizes are implicit in pnn_predict_f32_pel ([n][5w^2] for a fully-connected model, [n][3w^2] + [n][2w^2] for a
// convolutional one); the server checks a request's shape against the loaded model's kind before it queues it (Server::kind),
// and this is the second line: a batch of the wrong shape must never reach the copy of `n * 5w^2` floats.
int ctx_backend(void* user, int width, const float* above, const float* left, int n, int32_t* dst, float* out_f32)
{
    int is_fc = 0;
    if (pnn_model_info(static_cast<pnn_ctx*>(user), width, &is_fc, nullptr, nullptr) != PNN_OK) return PNN_E_MODEL;
    if ((is_fc != 0) != (left == nullptr)) return PNN_E_ARG;
    return pnn_predict_f32_pel(static_cast<pnn_ctx*>(user), width, above, left, n, out_f32, dst);
}

}  // namespace

// Client-side prediction cache: HM's rate-distortion search asks for the same block with the same context several times
// (SURVEY 3.2); a repeated request is answered here without a round trip.  Direct-mapped per (width, reply kind), exact
// match on the input bytes -- the same scheme as the in-process cache of pnn_abi.cpp ("cache_mb").
struct ClientCacheEntry { uint64_t hash = 0; bool valid = false; std::vector<char> in, vals; };
struct pnn_client {
    int fd;
    // the shared-memory request path (see ShmSlot above): this client's slot and the workers' doorbells, or NULL on the socket protocol
    void* slot = nullptr;
    void* doors = nullptr;
    int worker_of[5] = {0, 0, 0, 0, 0};
    int index = -1;                                   // this client's bit in the workers' "posted" words
    std::vector<char> buf, rbuf;                      // request / reply bytes
    size_t cache_bytes = 0;                           // 0 = off
    std::vector<ClientCacheEntry> cache[5][2];
    long hits = 0, misses = 0;
};

// Hash of the input bytes for the prediction cache: 8 bytes per step (a byte-wise FNV-1a cost 1.3 us per 8x8 lookup and 80 us
// per 64x64 one -- HM's RD search makes ~100 k lookups per picture); the entry is confirmed with memcmp, so only the spread matters.
static uint64_t hash_bytes(const void* data, size_t bytes, uint64_t h = 0x9e3779b97f4a7c15ull)
{
    const unsigned char* p = static_cast<const unsigned char*>(data);
    size_t i = 0;
    for (; i + 8 <= bytes; i += 8) {
        uint64_t v;
        memcpy(&v, p + i, 8);
        h = (h ^ v) * 0xff51afd7ed558ccdull;
        h ^= h >> 32;
    }
    for (; i < bytes; i++) h = (h ^ p[i]) * 0x100000001b3ull;
    return h ^ (h >> 29);
}

namespace {

// ---- the server ---------------------------------------------------------------------------------------------------------------
// The calling thread does all socket work (accept, receive, reply); `nworkers` worker threads do the backend calls: with one
// worker per width (pnn_service_run_table: five contexts, one model each) the GPU passes of different widths overlap each
// other AND the socket work, and while a worker is busy the requests for its width pile up -- the next batch forms by itself.
// Requests and replies carry a client ID, never a file descriptor (a dropped client's descriptor may be reused at once).
struct Server {
    pnn_service_backend backend;
    static constexpr int kMaxRep = 4;
    void* users[5][kMaxRep];       // backend handle for width index i (4, 8, 16, 32, 64), replica r
    int group[5] = {0, 1, 2, 3, 4};  // worker thread that serves width index i (nworkers == 5; PNN_SERVICE_GROUPS folds widths onto fewer threads)
    int nrep[5] = {1, 1, 1, 1, 1};   // replicas per worker queue (pnn_service_run_table: contexts per width, PNN_SERVICE_REPLICAS)
    int nworkers;                    // 1: one worker serves every width (a single context is not shared between threads); 5: one per width
    int max_batch, window_us;
    volatile int* stop;
    // Per width: 1 = the loaded model is fully-connected (requests must carry n_above = 5w^2, n_left = 0), 0 = convolutional
    // (3w^2 + 2w^2), -2 = no model for this width (every request is refused), -1 = unknown (generic backend: it validates).
    int kind[5] = {-1, -1, -1, -1, -1};
    // The arithmetic tag of what serves width index i (pnn_arithmetic_tag of its context; a generic backend: $PNN_SERVICE_TAG or
    // "backend:unspecified"): a client asks for it once (pnn_client_arithmetic_tag) and compares it with its decoder's -- an encoder
    // behind a service on one arithmetic and a stand-alone decoder on another drift apart silently.  Read-only after start.
    std::string tag[5];

    // One request on its way to a worker: from a socket (the floats were copied out of the receive buffer) or in a client's slot (`shm`:
    // the floats stay where the client wrote them until the batch is staged)
    struct Req {
        uint64_t id; int width; bool want_f32; std::vector<float> above, left; Clock::time_point t_in;
        std::shared_ptr<ShmClient> shm;
        uint32_t na = 0, nl = 0;                      // floats of the two inputs
        const float* A() const { return shm ? shm->slot->in : above.data(); }
        const float* L() const { return shm ? shm->slot->in + na : left.data(); }
    };
    struct Reply { uint64_t id; std::vector<char> bytes; Clock::time_point t_in; int k; };
    // Locks: one per worker queue (with its condition variable), one per I/O thread's reply / new-connection queues, one for
    // the peer accounting (only kept when a batching window is set) and the statistics.  (A single server-wide mutex was taken
    // five times per request by nine threads.)
    std::mutex qmu[5];
    std::vector<Req> queue[5];       // socket requests by worker, under qmu[k]
    // The workers' doorbells (ShmDoors: one futex word + a "sleeping" flag per worker) live in a page every shm client maps; the I/O
    // threads ring them for socket requests too.  The slots of the connected shm clients: an immutable snapshot, replaced under shm_mu
    // at hand-shake / teardown, read by the workers without a lock of their own.
    ShmDoors* doors = nullptr;
    int door_fd = -1;
    // The slots of the connected shm clients by index (the index of their flag bytes): entries are replaced by the I/O threads at
    // hand-shake / teardown (under shm_mu, std::atomic_store) and read by the workers with std::atomic_load.
    std::mutex shm_mu;
    std::shared_ptr<ShmClient> shm_table[kShmMaxSlots];
    std::atomic<int> shm_hi{0};                       // indices >= this were never handed out
    std::atomic<long> shm_clients{0}, shm_requests{0};
    long shm_live = 0;                                // slots listed now (under shm_mu)
    long spin_slots = 20, spin_cfg_us = 200;         // ShmDoors::spin_us = spin_cfg_us while at most spin_slots are listed (an encoder holds five)
    void publish_spin() { doors->spin_us.store(shm_live <= spin_slots ? (uint32_t)spin_cfg_us : 0u, std::memory_order_relaxed); }
    // The wake-ups of a batch's clients are NOT the worker's job: one futex_wake per client is 2-4 us of system call, in front of the
    // worker's next batch (first form of round 6: configs[3] 4.88 -> 5.11 s, the 4x4 / 8x8 workers being what its encoders wait for;
    // profiles/r06_service_transport.txt).  The worker writes the replies and flips the states -- a client that is not asleep yet sees
    // its reply at once -- and hands the sleepers to its WAKER thread: one doorbell per batch instead of one system call per client.
    struct Waker {
        std::mutex mu;
        std::vector<std::shared_ptr<ShmClient>> todo;
        std::atomic<uint32_t> bell{0}, sleeping{0};
        std::thread th;
    } wakers[5];
    void waker_loop(int k)
    {
        { char nm[16]; snprintf(nm, sizeof nm, "pnn-wk%d", nworkers == 1 ? 0 : kServiceWidthsOf(k)); prctl(PR_SET_NAME, nm, 0, 0, 0); }
        Waker& wk = wakers[k];
        std::vector<std::shared_ptr<ShmClient>> mine;
        for (;;) {
            { std::lock_guard<std::mutex> lk(wk.mu); mine.swap(wk.todo); }
            for (auto& sc : mine) futex_wake(&sc->slot->state, 1);
            if (!mine.empty()) { mine.clear(); continue; }
            if (quit_flag.load()) return;
            const uint32_t seen = wk.bell.load(std::memory_order_seq_cst);
            wk.sleeping.store(1, std::memory_order_seq_cst);
            bool any;
            { std::lock_guard<std::mutex> lk(wk.mu); any = !wk.todo.empty(); }
            if (!any && !quit_flag.load()) futex_wait(&wk.bell, seen, 50000);
            wk.sleeping.store(0, std::memory_order_seq_cst);
        }
    }
    void ring(int k)                 // a request for worker k has been posted (queue or slot): wake it if it sleeps
    {
        doors->door[k].bell.fetch_add(1, std::memory_order_seq_cst);
        if (doors->door[k].sleeping.load(std::memory_order_seq_cst)) futex_wake(&doors->door[k].bell, 1);
    }
    std::mutex dmu[8];               // done[t], fresh[t]
    std::mutex mu;                   // peers, statistics
    // Socket work is spread over `nio` I/O threads (round 3): ONE thread doing every recv / send / epoll_wait topped out at
    // ~120 k requests/s -- 24 and 100 HM encoders on one server ran at the same 119 k and 123 k requests/s, and two server
    // PROCESSES on the same GPU finished the 100-picture campaign in 8.9 s instead of 12.9 (profiles/r03_hm_runs.txt).  A
    // connection belongs to one I/O thread for its life (dealt round-robin at accept; the thread's index sits in the low bits
    // of the client ID), the per-width workers hand a reply to the owner's queue and wake it through its own eventfd.
    static constexpr int kMaxIo = 8;
    int nio = 1;
    std::vector<Reply> done[kMaxIo];
    std::vector<int> fresh[kMaxIo];  // accepted descriptors dealt to I/O thread t by the listener (thread 0), under dmu[t]
    std::atomic<int> n_peers{0}, n_waiting_peers{0};   // distinct peer processes connected / with a request in flight (window_us > 0 only; written under `mu`)
    std::atomic<bool> quit_flag{false};
    struct Peer { int conns = 0, in_flight = 0; };
    std::map<int, Peer> peers;       // under `mu`
    int wake_fd[kMaxIo];             // workers (and the listener) -> I/O thread t: an eventfd (one write to raise, ONE read to clear -- a pipe took two)
    std::atomic<uint64_t> next_seq{1};
    std::atomic<long> accepted{0}, refused{0};
    long served = 0, calls = 0, largest = 0;
    double busy_s[5] = {0, 0, 0, 0, 0};   // time inside the backend, per worker (PNN_SERVICE_DEBUG)
    long calls_w[5] = {0, 0, 0, 0, 0}, served_w[5] = {0, 0, 0, 0, 0};   // backend calls / requests per worker
    // Where a request's time inside the server goes (PNN_SERVICE_DEBUG): from its last byte received to its batch taken by the worker
    // (wait_s, under `mu`), and to its reply accepted by the socket (resident_s, per I/O thread: no lock).
    double wait_s[5] = {0, 0, 0, 0, 0}, resident_s[kMaxIo][5] = {};
    long resident_n[kMaxIo][5] = {};
    double shm_resident_s[5] = {0, 0, 0, 0, 0};      // ... of the requests that came through slots: post -> reply written (under `mu`)
    long shm_resident_n[5] = {0, 0, 0, 0, 0};

    static int widx(int w) { return w == 4 ? 0 : w == 8 ? 1 : w == 16 ? 2 : w == 32 ? 3 : 4; }
    static int kServiceWidthsOf(int k) { return 4 << k; }
    int worker_of(int width) const { return nworkers == 1 ? 0 : group[widx(width)]; }

    // One batch on its way through a worker: the requests and their inputs stacked.
    struct Flight {
        std::vector<Req> batch;
        std::vector<float> above, left;
        bool any_f32 = false, any_pel = false;
        double waited = 0;                           // seconds the requests sat in the queue, summed
    };
    // An error reply written straight into a slot (a request that never reaches the backend).
    void slot_reply_error(ShmSlot* sl, int rc)
    {
        sl->rsp = RspHeader{rc, 0u};
        sl->state.store(kSlotReady, std::memory_order_seq_cst);
        if (sl->client_sleeps.load(std::memory_order_seq_cst)) futex_wake(&sl->state, 1);
    }
    // one batch = requests of ONE width and ONE input kind: the socket requests queued for worker k in arrival order (under qmu[k],
    // taken here), then the posted slots of the shm clients (claimed by compare-and-swap: a slot belongs to the worker that took it)
    void take(int k, std::vector<Req>& batch)
    {
        batch.clear();
        int w = 0;
        bool has_left = false;
        {
            std::lock_guard<std::mutex> lk(qmu[k]);
            if (!queue[k].empty()) {
                w = queue[k][0].width;
                has_left = queue[k][0].nl != 0;
                for (size_t i = 0; i < queue[k].size() && (int)batch.size() < max_batch;) {
                    if (queue[k][i].width == w && (queue[k][i].nl != 0) == has_left) {
                        batch.push_back(std::move(queue[k][i]));
                        queue[k].erase(queue[k].begin() + (long)i);
                    } else {
                        ++i;
                    }
                }
            }
        }
        std::atomic<uint64_t>* const flags = doors->posted[k];
        const int hi = shm_hi.load(std::memory_order_acquire);
        for (int wd = 0; wd * 64 < hi && (int)batch.size() < max_batch; wd++) {
            if (!flags[wd].load(std::memory_order_acquire)) continue;
            uint64_t bits = flags[wd].exchange(0, std::memory_order_seq_cst);   // claimed BEFORE the look: a post behind it sets its bit again
            uint64_t leave = 0;                                                  // ... and what is not this batch's goes back
            while (bits) {
                const int b = __builtin_ctzll(bits);
                bits &= bits - 1;
                if ((int)batch.size() >= max_batch) { leave |= 1ull << b; continue; }
                const int idx = wd * 64 + b;
                const std::shared_ptr<ShmClient> sc = std::atomic_load(&shm_table[idx]);
                if (!sc || sc->dead.load(std::memory_order_relaxed)) continue;
                ShmSlot* sl = sc->slot;
                if (sl->state.load(std::memory_order_seq_cst) != kSlotPosted) continue;
                const ReqHeader h = sl->hdr;             // a copy: the client may scribble, the checks below are made on what is used
                const bool well = h.magic == kMagic && valid_width(h.width);
                if (well && worker_of(h.width) != k) {   // rang the wrong bell: passed on
                    doors->posted[worker_of(h.width)][wd].fetch_or(1ull << b, std::memory_order_seq_cst);
                    ring(worker_of(h.width));
                    continue;
                }
                if (well && w && (h.width != w || (h.n_left != 0) != has_left)) { leave |= 1ull << b; continue; }   // this worker's, another batch's
                uint32_t expect = kSlotPosted;
                if (!sl->state.compare_exchange_strong(expect, kSlotTaken, std::memory_order_acq_rel)) continue;
                ++shm_requests;
                if (!valid_header(h) || (h.flags & (kWantTag | kWantShm))) { slot_reply_error(sl, PNN_E_ARG); continue; }
                const int kd = kind[widx(h.width)];
                if (kd == -2 || (kd >= 0 && (kd == 1) != (h.n_left == 0))) { ++refused; slot_reply_error(sl, kd == -2 ? PNN_E_MODEL : PNN_E_ARG); continue; }
                if (!w) { w = h.width; has_left = h.n_left != 0; }
                Req r;
                r.id = 0; r.width = h.width; r.want_f32 = (h.flags & kWantF32) != 0; r.shm = sc; r.na = h.n_above; r.nl = h.n_left;
                const uint64_t now = now_ns(), posted = sl->posted_ns;
                r.t_in = Clock::now() - std::chrono::nanoseconds(posted <= now && now - posted < 60000000000ull ? now - posted : 0);
                batch.push_back(std::move(r));
            }
            if (leave) flags[wd].fetch_or(leave, std::memory_order_seq_cst);
        }
    }
    bool has_pending(int k)          // anything worker k could take right now?  (a stale bit says yes once: take() clears it)
    {
        { std::lock_guard<std::mutex> lk(qmu[k]); if (!queue[k].empty()) return true; }
        const int hi = shm_hi.load(std::memory_order_acquire);
        for (int wd = 0; wd * 64 < hi; wd++) if (doors->posted[k][wd].load(std::memory_order_seq_cst)) return true;
        return false;
    }
    void stage(Flight& f)
    {
        const size_t n = f.batch.size(), na = f.batch[0].na, nl = f.batch[0].nl;
        f.any_f32 = f.any_pel = false;
        f.above.resize(n * na); f.left.resize(n * nl);
        const auto now = Clock::now();
        f.waited = 0;
        for (size_t i = 0; i < n; i++) {
            memcpy(f.above.data() + i * na, f.batch[i].A(), na * 4);
            if (nl) memcpy(f.left.data() + i * nl, f.batch[i].L(), nl * 4);
            (f.batch[i].want_f32 ? f.any_f32 : f.any_pel) = true;
            f.waited += std::chrono::duration<double>(now - f.batch[i].t_in).count();
        }
    }
    // the replies of one finished batch to the I/O threads that own the connections, and the statistics
    void reply(int k, Flight& f, int rc, const int32_t* dst, const float* out, double busy)
    {
        const size_t n = f.batch.size(), w2 = (size_t)f.batch[0].width * f.batch[0].width;
        const int wi = nworkers == 1 ? k : widx(f.batch[0].width);   // the statistics are kept per WIDTH (two widths may share a worker thread)
        std::vector<Reply> replies;
        replies.reserve(n);
        std::vector<std::shared_ptr<ShmClient>> sleepers;
        double shm_resident = 0;
        long shm_n = 0;
        for (size_t i = 0; i < n; i++) {
            const RspHeader rh{rc, rc == 0 ? (uint32_t)w2 : 0u};
            if (f.batch[i].shm) {                      // into the client's slot, by this thread: values, header, state, wake-up
                ShmSlot* sl = f.batch[i].shm->slot;
                if (rc == 0) memcpy(sl->out, f.batch[i].want_f32 ? reinterpret_cast<const void*>(out + i * w2) : reinterpret_cast<const void*>(dst + i * w2), w2 * 4);
                sl->rsp = rh;
                sl->state.store(kSlotReady, std::memory_order_seq_cst);
                if (sl->client_sleeps.load(std::memory_order_seq_cst)) sleepers.push_back(f.batch[i].shm);   // (a client that has not gone to sleep yet re-reads the state first)
                shm_resident += std::chrono::duration<double>(Clock::now() - f.batch[i].t_in).count(); ++shm_n;
                continue;
            }
            replies.emplace_back();
            Reply& rp = replies.back();
            rp.id = f.batch[i].id; rp.t_in = f.batch[i].t_in; rp.k = wi;
            const char* hp = reinterpret_cast<const char*>(&rh);
            rp.bytes.assign(hp, hp + sizeof rh);
            if (rc == 0) {
                const char* pp = f.batch[i].want_f32 ? reinterpret_cast<const char*>(out + i * w2) : reinterpret_cast<const char*>(dst + i * w2);
                rp.bytes.insert(rp.bytes.end(), pp, pp + w2 * 4);
            }
        }
        if (!sleepers.empty()) {                     // to the waker: ONE doorbell for the batch
            Waker& wk = wakers[k];
            { std::lock_guard<std::mutex> lk(wk.mu); for (auto& sc : sleepers) wk.todo.push_back(std::move(sc)); }
            wk.bell.fetch_add(1, std::memory_order_seq_cst);
            if (wk.sleeping.load(std::memory_order_seq_cst)) futex_wake(&wk.bell, 1);
        }
        bool woke[kMaxIo] = {false};
        for (int t = 0; t < nio; t++) {          // replies to their owners, one lock per I/O thread that has any
            bool any = false;
            for (auto& r : replies) any |= (int)(r.id & 15) % nio == t;
            if (!any) continue;
            woke[t] = true;
            std::lock_guard<std::mutex> lk(dmu[t]);
            for (auto& r : replies) if ((int)(r.id & 15) % nio == t) done[t].push_back(std::move(r));
        }
        const uint64_t one = 1;
        for (int t = 0; t < nio; t++) if (woke[t]) (void)!write(wake_fd[t], &one, 8);
        std::lock_guard<std::mutex> lk(mu);
        served += (long)n; ++calls; largest = std::max<long>(largest, (long)n);
        ++calls_w[wi]; served_w[wi] += (long)n; busy_s[wi] += busy; wait_s[wi] += f.waited;
        shm_resident_s[wi] += shm_resident; shm_resident_n[wi] += shm_n;
    }
    // Blocks until worker k has something to do (false: the server stops).  An idle worker sleeps on its doorbell (a futex word the
    // clients and the I/O threads ring); with a batching window it gives stragglers a moment to join -- unless every peer process
    // already waits for an answer (an encoder is single-threaded and blocks on its request: nobody else can arrive).
    bool wait_for_work(int k, std::vector<Req>& batch)
    {
        ShmDoor& d = doors->door[k];
        for (;;) {
            if (quit_flag.load()) return false;
            take(k, batch);
            if (!batch.empty()) {
                if (window_us > 0 && (int)batch.size() < max_batch && n_waiting_peers.load() < n_peers.load()) {
                    const auto until = Clock::now() + std::chrono::microseconds(window_us);
                    while ((int)batch.size() < max_batch && n_waiting_peers.load() < n_peers.load() && !quit_flag.load()) {
                        const auto now = Clock::now();
                        if (now >= until) break;
                        const uint32_t seen = d.bell.load(std::memory_order_seq_cst);
                        d.sleeping.store(1, std::memory_order_seq_cst);
                        if (!has_pending(k)) futex_wait(&d.bell, seen, std::max<long>(1, std::chrono::duration_cast<std::chrono::microseconds>(until - now).count()));
                        d.sleeping.store(0, std::memory_order_seq_cst);
                        take_more(k, batch);
                    }
                    if (quit_flag.load()) return false;
                }
                return true;
            }
            if (const uint32_t spin = doors->spin_us.load(std::memory_order_relaxed)) {   // few clients: poll for the next request before sleeping
                const uint64_t t0 = now_ns();
                bool found = false;
                while (!(found = has_pending(k)) && !quit_flag.load(std::memory_order_relaxed) && now_ns() - t0 < (uint64_t)spin * 1000u) __builtin_ia32_pause();
                if (found) continue;
            }
            const uint32_t seen = d.bell.load(std::memory_order_seq_cst);
            d.sleeping.store(1, std::memory_order_seq_cst);
            if (!has_pending(k) && !quit_flag.load()) futex_wait(&d.bell, seen, 50000);   // (bounded: a lost wake-up costs 50 ms, not the server)
            d.sleeping.store(0, std::memory_order_seq_cst);
        }
    }
    // the batching window: requests of the batch's width and kind that arrived meanwhile join it
    void take_more(int k, std::vector<Req>& batch)
    {
        const int w = batch[0].width;
        const bool has_left = batch[0].nl != 0;
        {
            std::lock_guard<std::mutex> lk(qmu[k]);
            for (size_t i = 0; i < queue[k].size() && (int)batch.size() < max_batch;) {
                if (queue[k][i].width == w && (queue[k][i].nl != 0) == has_left) {
                    batch.push_back(std::move(queue[k][i]));
                    queue[k].erase(queue[k].begin() + (long)i);
                } else {
                    ++i;
                }
            }
        }
    }

    void worker(int k, int r)
    {
        { char nm[16]; snprintf(nm, sizeof nm, "pnn-w%d", nworkers == 1 ? 0 : kServiceWidthsOf(k)); prctl(PR_SET_NAME, nm, 0, 0, 0); }   // (per-thread CPU accounting: campaign.py reads /proc/<pid>/task/*/stat)
        prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0);   // the contexts' "wait_sleep" naps are tens of microseconds: not with the default 50 us of slack
        Flight f;
        std::vector<float> out;
        std::vector<int32_t> dst;
        for (;;) {
            if (!wait_for_work(k, f.batch)) return;
            if (f.batch.empty()) continue;
            stage(f);
            const int w = f.batch[0].width;
            const size_t n = f.batch.size(), w2 = (size_t)w * w;
            if (f.any_pel) dst.resize(n * w2);
            if (f.any_f32) out.resize(n * w2);
            const auto tb0 = Clock::now();
            const int rc = backend(users[widx(w)][r], w, f.above.data(), f.left.empty() ? nullptr : f.left.data(), (int)n, f.any_pel ? dst.data() : nullptr,
                                   f.any_f32 ? out.data() : nullptr);
            reply(k, f, rc, dst.data(), out.data(), std::chrono::duration<double>(Clock::now() - tb0).count());
        }
    }

    void peer_conn(int pid, int d)                   // a connection of `pid` opened (+1) / closed (-1); under `mu`
    {
        Peer& pr = peers[pid];
        pr.conns += d;
        if (pr.conns <= 0) { if (pr.in_flight > 0) --n_waiting_peers; peers.erase(pid); }
        n_peers = (int)peers.size();
    }
    void peer_flight(int pid, int d)                 // a request of `pid` queued (+1) / answered or dropped (-1); under `mu`
    {
        auto it = peers.find(pid);
        if (it == peers.end()) return;
        const bool was = it->second.in_flight > 0;
        it->second.in_flight += d;
        const bool is = it->second.in_flight > 0;
        if (is != was) n_waiting_peers += is ? 1 : -1;
    }

    // The hand-shake of the shared-memory path: a memfd of one slot for this client, listed for the workers; the reply carries the slot's
    // and the doorbell page's descriptors (SCM_RIGHTS) and the worker index of each width.  false = the client is dropped.
    bool shm_handshake(Client& c)
    {
        if (c.shm || !c.tx.empty()) return false;                // once per connection, with nothing else under way
        const int mfd = (int)syscall(SYS_memfd_create, "pnn-slot", 1u /* MFD_CLOEXEC */);
        if (mfd < 0) return false;
        void* mem = ftruncate(mfd, (off_t)kSlotBytes) == 0 ? mmap(nullptr, kSlotBytes, PROT_READ | PROT_WRITE, MAP_SHARED, mfd, 0) : MAP_FAILED;
        if (mem == MAP_FAILED) { close(mfd); return false; }
        auto sc = std::make_shared<ShmClient>();
        sc->slot = new (mem) ShmSlot;                            // (fresh pages are zero: state = kSlotIdle)
        {
            std::lock_guard<std::mutex> lk(shm_mu);
            for (int i = 0; i < kShmMaxSlots && sc->index < 0; i++) if (!shm_table[i]) sc->index = i;
            if (sc->index >= 0) {
                for (int k = 0; k < 5; k++) doors->posted[k][sc->index >> 6].fetch_and(~(1ull << (sc->index & 63)), std::memory_order_seq_cst);   // a bit its previous owner left
                std::atomic_store(&shm_table[sc->index], sc);
                if (sc->index >= shm_hi.load()) shm_hi.store(sc->index + 1, std::memory_order_release);
                ++shm_live;
                publish_spin();
            }
        }
        struct { RspHeader rh; int32_t worker[5]; int32_t index; } body;
        body.rh = RspHeader{sc->index >= 0 ? 0 : PNN_E_NOMEM, 6u};     // more clients than flag bits: this one stays on the socket protocol
        for (int i = 0; i < 5; i++) body.worker[i] = worker_of(4 << i);
        body.index = sc->index;
        if (sc->index < 0) {
            close(mfd);
            ssize_t ns;
            do ns = send(c.fd, &body, sizeof body, MSG_NOSIGNAL); while (ns < 0 && errno == EINTR);
            return ns == (ssize_t)sizeof body;
        }
        iovec iov{&body, sizeof body};
        alignas(cmsghdr) char ctl[CMSG_SPACE(2 * sizeof(int))];
        memset(ctl, 0, sizeof ctl);
        msghdr mh;
        memset(&mh, 0, sizeof mh);
        mh.msg_iov = &iov; mh.msg_iovlen = 1; mh.msg_control = ctl; mh.msg_controllen = sizeof ctl;
        cmsghdr* cm = CMSG_FIRSTHDR(&mh);
        cm->cmsg_level = SOL_SOCKET; cm->cmsg_type = SCM_RIGHTS; cm->cmsg_len = CMSG_LEN(2 * sizeof(int));
        const int fds[2] = {mfd, door_fd};
        memcpy(CMSG_DATA(cm), fds, sizeof fds);
        ssize_t sent;
        do sent = sendmsg(c.fd, &mh, MSG_NOSIGNAL); while (sent < 0 && errno == EINTR);
        close(mfd);                                              // the mapping and the client's descriptor keep the memory
        if (sent != (ssize_t)sizeof body) { shm_unlist(sc); return false; }
        c.shm = sc;
        ++shm_clients;
        return true;
    }
    void shm_unlist(const std::shared_ptr<ShmClient>& sc)
    {
        sc->dead.store(true);
        std::lock_guard<std::mutex> lk(shm_mu);
        if (sc->index >= 0 && std::atomic_load(&shm_table[sc->index]) == sc) {
            std::atomic_store(&shm_table[sc->index], std::shared_ptr<ShmClient>());
            --shm_live;
            publish_spin();
        }
    }

    // One I/O thread: its share of the connections (receive, queue for the workers, reply); thread 0 also owns the listener.
    // epoll, not poll: with hundreds of connections (an encoder holds five) a poll set rebuilt and scanned per wake-up was what
    // bounded the server; event data = client ID (0: listener, 1: this thread's wake eventfd).
    void io_loop(const int t, const int lfd, const int ep)
    {
        if (t > 0) { char nm[16]; snprintf(nm, sizeof nm, "pnn-io%d", t); prctl(PR_SET_NAME, nm, 0, 0, 0); }
        std::map<uint64_t, Client> clients;          // by client ID
        auto ep_ctl = [&](int op, int fd, uint32_t events, uint64_t id) {
            epoll_event ev;
            memset(&ev, 0, sizeof ev);
            ev.events = events; ev.data.u64 = id;
            epoll_ctl(ep, op, fd, &ev);
        };
        auto arm = [&](uint64_t id, Client& c) { ep_ctl(EPOLL_CTL_MOD, c.fd, EPOLLIN | (c.tx.empty() ? 0u : (uint32_t)EPOLLOUT), id); };
        auto adopt = [&](int cfd) {                  // a connection dealt to this thread
            set_nonblocking(cfd);
            Client c;
            c.fd = cfd;
            ucred cred;
            socklen_t len = sizeof cred;
            if (getsockopt(cfd, SOL_SOCKET, SO_PEERCRED, &cred, &len) == 0) c.pid = (int)cred.pid;
            const uint64_t nid = (next_seq.fetch_add(1) << 4) | (uint64_t)t;     // never 0 or 1: the sequence starts at 1
            const int pid = c.pid;
            clients.emplace(nid, std::move(c));
            ep_ctl(EPOLL_CTL_ADD, cfd, EPOLLIN, nid);
            if (window_us > 0) { std::lock_guard<std::mutex> lk(mu); peer_conn(pid, +1); }
        };
        auto drop = [&](uint64_t id) {
            auto it = clients.find(id);
            if (it == clients.end()) return;
            close(it->second.fd);
            const int pid = it->second.pid;
            const bool was_in_flight = it->second.in_flight;
            if (it->second.shm) shm_unlist(it->second.shm);          // a request still posted in its slot is never taken; one already taken is answered into the void
            clients.erase(it);
            if (was_in_flight)
                for (int k = 0; k < nworkers; k++) {
                    std::lock_guard<std::mutex> lk(qmu[k]);
                    queue[k].erase(std::remove_if(queue[k].begin(), queue[k].end(), [id](const Req& r) { return r.id == id; }), queue[k].end());
                }
            if (window_us > 0) {
                std::lock_guard<std::mutex> lk(mu);
                if (was_in_flight) peer_flight(pid, -1);
                peer_conn(pid, -1);
            }
        };
        auto flush = [&](Client& c) {                         // false = broken
            while (c.tx_off < c.tx.size()) {
                const ssize_t r = send(c.fd, c.tx.data() + c.tx_off, c.tx.size() - c.tx_off, MSG_NOSIGNAL);
                if (r < 0) {
                    if (errno == EINTR) continue;
                    if (errno == EAGAIN || errno == EWOULDBLOCK) return true;
                    return false;
                }
                c.tx_off += (size_t)r;
            }
            c.tx.clear(); c.tx_off = 0;
            return true;
        };
        // Moves whatever the socket holds into the client's buffer; queues the request when it is complete.
        // false = client gone or protocol violation.
        auto receive = [&](uint64_t id, Client& c) {
            for (;;) {
                size_t want = sizeof(ReqHeader);
                if (c.rx.size() >= sizeof(ReqHeader)) {
                    ReqHeader h;
                    memcpy(&h, c.rx.data(), sizeof h);
                    if (!valid_header(h)) return false;
                    want = sizeof h + ((size_t)h.n_above + h.n_left) * 4;
                    if (c.rx.size() > want) return false;     // bytes of a second request behind an unanswered one
                    if (c.rx.size() == want) {
                        if (c.in_flight) return false;        // one outstanding request per client
                        if (h.flags & kWantShm) {             // the hand-shake of the shared-memory request path: a slot for this client
                            c.rx.clear();
                            return shm_handshake(c);
                        }
                        if (h.flags & kWantTag) {             // answered here, by the I/O thread: never queued
                            const RspHeader rh{0, (uint32_t)(kTagBytes / 4)};
                            char body[kTagBytes];
                            memset(body, 0, sizeof body);
                            const std::string& tg = tag[widx(h.width)];
                            memcpy(body, tg.data(), std::min(tg.size(), kTagBytes - 1));
                            const char* hp = reinterpret_cast<const char*>(&rh);
                            if (c.tx.empty()) c.tx_since = Clock::now();
                            c.tx.insert(c.tx.end(), hp, hp + sizeof rh);
                            c.tx.insert(c.tx.end(), body, body + kTagBytes);
                            c.rx.clear();
                            return flush(c);
                        }
                        const int kd = kind[widx(h.width)];
                        if (kd == -2 || (kd >= 0 && (kd == 1) != (h.n_left == 0))) {
                            // well-formed, but not the input shape of the model loaded for this width (e.g. an encoder whose
                            // local table lists a convolutional 8x8 model talking to a server with the fully-connected one):
                            // answered with an error code at once, never queued
                            const RspHeader rh{kd == -2 ? PNN_E_MODEL : PNN_E_ARG, 0u};
                            const char* hp = reinterpret_cast<const char*>(&rh);
                            if (c.tx.empty()) c.tx_since = Clock::now();
                            c.tx.insert(c.tx.end(), hp, hp + sizeof rh);
                            c.rx.clear();
                            ++refused;
                            return flush(c);
                        }
                        Req r;
                        r.id = id; r.width = h.width; r.want_f32 = (h.flags & kWantF32) != 0; r.t_in = Clock::now();
                        r.above.resize(h.n_above); r.left.resize(h.n_left); r.na = h.n_above; r.nl = h.n_left;
                        memcpy(r.above.data(), c.rx.data() + sizeof h, (size_t)h.n_above * 4);
                        if (h.n_left) memcpy(r.left.data(), c.rx.data() + sizeof h + (size_t)h.n_above * 4, (size_t)h.n_left * 4);
                        c.in_flight = true;
                        c.rx.clear();
                        const int k = worker_of(h.width);
                        if (window_us > 0) { std::lock_guard<std::mutex> lk(mu); peer_flight(c.pid, +1); }
                        {
                            std::lock_guard<std::mutex> lk(qmu[k]);
                            queue[k].push_back(std::move(r));
                        }
                        ring(k);
                        // the other workers only care when "every peer waits" has just become true (they may be sitting in their window)
                        if (nworkers > 1 && window_us > 0 && n_waiting_peers.load() >= n_peers.load())
                            for (int o = 0; o < nworkers; o++) if (o != k) ring(o);
                        return true;
                    }
                }
                // one outstanding request per client: everything the socket holds belongs to this request, so a header and its
                // payload (one send on the client side) come in with ONE recv instead of two
                const size_t have = c.rx.size();
                const size_t room = have < sizeof(ReqHeader) ? sizeof(ReqHeader) + (size_t)5 * 64 * 4 : want;   // header + an 8x8 FC context
                c.rx.resize(room);
                const ssize_t r = recv(c.fd, c.rx.data() + have, room - have, 0);
                if (r <= 0) {
                    c.rx.resize(have);
                    if (r < 0 && (errno == EAGAIN || errno == EWOULDBLOCK || errno == EINTR)) return true;   // rest comes later
                    return false;                             // closed or broken
                }
                c.rx.resize(have + (size_t)r);
            }
        };
        epoll_event evs[256];
        auto last_sweep = Clock::now();
        std::vector<uint64_t> gone;
        int deal = 0;                                // listener: next I/O thread to get a connection
        while (!__atomic_load_n(const_cast<const int*>(stop), __ATOMIC_ACQUIRE)) {   // (an atomic load of the caller's flag: `volatile` alone is a data race by the letter)
            const int r = epoll_wait(ep, evs, 256, 50);
            gone.clear();
            for (int i = 0; i < r; i++) {
                const uint64_t id = evs[i].data.u64;
                const uint32_t e = evs[i].events;
                if (id == 0) {                                // the listener (thread 0 only): deal the new connections
                    bool woke[kMaxIo] = {false};
                    for (;;) {
                        const int cfd = accept(lfd, nullptr, nullptr);
                        if (cfd < 0) break;
                        ++accepted;
                        const int to = deal;
                        deal = (deal + 1) % nio;
                        if (to == t) { adopt(cfd); continue; }
                        std::lock_guard<std::mutex> lk(dmu[to]);
                        fresh[to].push_back(cfd);
                        woke[to] = true;
                    }
                    const uint64_t one = 1;
                    for (int o = 0; o < nio; o++) if (woke[o]) (void)!write(wake_fd[o], &one, 8);
                } else if (id == 1) {                         // new connections from the listener, replies from the workers
                    uint64_t raised;
                    (void)!read(wake_fd[t], &raised, 8);
                    std::vector<Reply> ready;
                    std::vector<int> mine;
                    {
                        std::lock_guard<std::mutex> lk(dmu[t]);
                        ready.swap(done[t]);
                        mine.swap(fresh[t]);
                    }
                    for (int cfd : mine) adopt(cfd);
                    for (Reply& rp : ready) {
                        auto it = clients.find(rp.id);
                        if (it == clients.end()) continue;    // dropped meanwhile
                        Client& c = it->second;
                        if (c.tx.empty()) c.tx_since = Clock::now();
                        c.tx.insert(c.tx.end(), rp.bytes.begin(), rp.bytes.end());
                        if (c.in_flight) {
                            c.in_flight = false;
                            if (window_us > 0) { std::lock_guard<std::mutex> lk(mu); peer_flight(c.pid, -1); }
                        }
                        if (!flush(c)) gone.push_back(rp.id);
                        else if (!c.tx.empty()) arm(rp.id, c);   // the socket took only part of it: wait for EPOLLOUT
                        resident_s[t][rp.k] += std::chrono::duration<double>(Clock::now() - rp.t_in).count(); ++resident_n[t][rp.k];
                    }
                } else {
                    auto it = clients.find(id);
                    if (it == clients.end()) continue;
                    Client& c = it->second;
                    bool ok = true;
                    if (e & EPOLLOUT) { ok = flush(c); if (ok && c.tx.empty()) arm(id, c); }
                    if (ok && (e & EPOLLIN)) { ok = receive(id, c); if (ok && !c.tx.empty()) arm(id, c); }
                    else if (ok && (e & (EPOLLHUP | EPOLLERR))) ok = false;
                    if (!ok) gone.push_back(id);
                }
            }
            const auto now = Clock::now();
            if (std::chrono::duration_cast<std::chrono::milliseconds>(now - last_sweep).count() >= 250) {   // replies nobody reads
                last_sweep = now;
                for (auto& kv : clients)
                    if (!kv.second.tx.empty() && std::chrono::duration_cast<std::chrono::milliseconds>(now - kv.second.tx_since).count() > kStallMs)
                        gone.push_back(kv.first);
            }
            std::sort(gone.begin(), gone.end());
            gone.erase(std::unique(gone.begin(), gone.end()), gone.end());
            for (uint64_t id : gone) drop(id);
        }
        for (auto& kv : clients) close(kv.second.fd);
    }

    int run(const char* socket_path, long* stats)
    {
        sockaddr_un addr;
        if (make_addr(socket_path, &addr)) return PNN_E_ARG;
        if (const char* e = getenv("PNN_SERVICE_IO_THREADS")) nio = atoi(e);
        nio = std::max(1, std::min(nio, (int)kMaxIo));
        // the doorbell page: the workers' futex words, mapped by every shm client
        door_fd = (int)syscall(SYS_memfd_create, "pnn-doors", 1u /* MFD_CLOEXEC */);
        void* dm = (door_fd >= 0 && ftruncate(door_fd, (off_t)kDoorBytes) == 0) ? mmap(nullptr, kDoorBytes, PROT_READ | PROT_WRITE, MAP_SHARED, door_fd, 0) : MAP_FAILED;
        if (dm == MAP_FAILED) { if (door_fd >= 0) close(door_fd); return PNN_E_IO; }
        doors = new (dm) ShmDoors;
        if (const char* e = getenv("PNN_SERVICE_SPIN_SLOTS")) spin_slots = atol(e);        // poll instead of sleeping while at most this many slots are listed (0: never)
        if (const char* e = getenv("PNN_SERVICE_SPIN_US")) spin_cfg_us = std::max(0L, std::min(atol(e), 1000L));
        { std::lock_guard<std::mutex> lk(shm_mu); publish_spin(); }
        struct DoorGuard { Server* sv; ~DoorGuard() { munmap(sv->doors, kDoorBytes); close(sv->door_fd); sv->doors = nullptr; } } door_guard{this};
        const int lfd = socket(AF_UNIX, SOCK_STREAM, 0);
        if (lfd < 0) return PNN_E_IO;
        unlink(socket_path);
        // every descriptor the loops need exists BEFORE a thread is started: an early error return must not leave joinable
        // threads (std::terminate) or leaked descriptors behind
        int eps[kMaxIo];
        int made = 0;
        bool ok = bind(lfd, reinterpret_cast<sockaddr*>(&addr), sizeof addr) == 0 && listen(lfd, 512) == 0;
        for (; ok && made < nio; made++) {
            wake_fd[made] = -1;
            eps[made] = -1;
            if ((wake_fd[made] = eventfd(0, EFD_NONBLOCK | EFD_CLOEXEC)) < 0) { ok = false; break; }
            eps[made] = epoll_create1(0);
            if (eps[made] < 0) { close(wake_fd[made]); ok = false; break; }
        }
        if (!ok) {
            for (int t = 0; t < made; t++) { close(eps[t]); close(wake_fd[t]); }
            close(lfd);
            unlink(socket_path);
            return PNN_E_IO;
        }
        set_nonblocking(lfd);
        for (int t = 0; t < nio; t++) {
            epoll_event ev;
            memset(&ev, 0, sizeof ev);
            ev.events = EPOLLIN; ev.data.u64 = 1;
            epoll_ctl(eps[t], EPOLL_CTL_ADD, wake_fd[t], &ev);
        }
        {
            epoll_event ev;
            memset(&ev, 0, sizeof ev);
            ev.events = EPOLLIN; ev.data.u64 = 0;
            epoll_ctl(eps[0], EPOLL_CTL_ADD, lfd, &ev);
        }
        std::vector<std::thread> threads;
        for (int k = 0; k < nworkers; k++) {
            for (int r = 0; r < nrep[k]; r++) threads.emplace_back([this, k, r] { worker(k, r); });
            wakers[k].th = std::thread([this, k] { waker_loop(k); });
        }
        std::vector<std::thread> io;
        for (int t = 1; t < nio; t++) io.emplace_back([this, t, lfd, &eps] { io_loop(t, lfd, eps[t]); });
        io_loop(0, lfd, eps[0]);                     // the calling thread: listener + its share of the connections
        for (auto& th : io) th.join();
        quit_flag = true;
        for (int t = 0; t < nio; t++) for (int cfd : fresh[t]) close(cfd);   // dealt but never adopted
        for (int k = 0; k < nworkers; k++) { doors->door[k].bell.fetch_add(1, std::memory_order_seq_cst); futex_wake(&doors->door[k].bell, 64); }
        // whoever still waits in a slot (a server stopped under its clients) is told: the closing control socket says the rest
        for (auto& th : threads) th.join();
        threads.clear();
        for (int k = 0; k < nworkers; k++) { wakers[k].bell.fetch_add(1, std::memory_order_seq_cst); futex_wake(&wakers[k].bell, 1); wakers[k].th.join(); }
        {
            std::lock_guard<std::mutex> lk(shm_mu);
            for (int i = 0; i < kShmMaxSlots; i++) {
                const std::shared_ptr<ShmClient> sc = std::atomic_load(&shm_table[i]);
                if (sc && sc->slot->state.load() == kSlotPosted) slot_reply_error(sc->slot, PNN_E_IO);
                std::atomic_store(&shm_table[i], std::shared_ptr<ShmClient>());
            }
        }
        for (auto& th : threads) th.join();
        for (int t = 0; t < nio; t++) { close(eps[t]); close(wake_fd[t]); }
        close(lfd);
        unlink(socket_path);
        if (stats) { stats[0] = served; stats[1] = calls; stats[2] = largest; stats[3] = accepted.load(); }
        if (getenv("PNN_SERVICE_DEBUG"))
            fprintf(stderr, "[pnn-service] transport: %ld of %ld clients on the shared-memory request path, %ld requests through slots\n", shm_clients.load(), accepted.load(), shm_requests.load());
        if (getenv("PNN_SERVICE_DEBUG"))
            for (int k = 0; k < nworkers; k++)
            {
                double rs = shm_resident_s[k]; long rn = shm_resident_n[k];
                for (int t = 0; t < nio; t++) { rs += resident_s[t][k]; rn += resident_n[t][k]; }
                fprintf(stderr, "[pnn-service] worker %d: %.2f s inside the backend, %ld calls (%.1f us each), %ld requests (%.2f per call); per request %.1f us queued before "
                        "its batch is taken, %.1f us from last byte in to reply out\n", k, busy_s[k], calls_w[k], calls_w[k] ? busy_s[k] * 1e6 / calls_w[k] : 0.0, served_w[k],
                        calls_w[k] ? (double)served_w[k] / calls_w[k] : 0.0, served_w[k] ? wait_s[k] * 1e6 / served_w[k] : 0.0, rn ? rs * 1e6 / rn : 0.0);
            }
        return PNN_OK;
    }
};

}  // namespace

extern "C" {

int pnn_service_run_backend(const char* socket_path, pnn_service_backend backend, void* user, int max_batch, int window_us,
                            volatile int* stop, long* stats)
{
    if (!backend || !stop || max_batch < 1 || window_us < 0) return PNN_E_ARG;
    Server sv;
    sv.backend = backend; sv.nworkers = 1; sv.max_batch = max_batch; sv.window_us = window_us; sv.stop = stop;
    for (auto& ur : sv.users) for (void*& u : ur) u = user;
    sv.nio = 2;
    // $PNN_SERVICE_WORKERS=5: one worker thread per width, as pnn_service_run_table has them -- the backend is then called from five
    // threads at once and must be thread-safe (the race-detector driver tests/tsan_service.cpp: the production thread layout without a GPU)
    if (const char* e = getenv("PNN_SERVICE_WORKERS")) if (atoi(e) == 5) { sv.nworkers = 5; sv.nio = 4; }
    { const char* e = getenv("PNN_SERVICE_TAG"); for (auto& t : sv.tag) t = e ? e : "backend:unspecified"; }
    return sv.run(socket_path, stats);
}

static const int kServiceWidths[5] = {4, 8, 16, 32, 64};

int pnn_service_run(const char* socket_path, pnn_ctx* ctx, int max_batch, int window_us, volatile int* stop, long* stats)
{
    if (!ctx || !stop || max_batch < 1 || window_us < 0) return PNN_E_ARG;
    Server sv;
    sv.backend = ctx_backend; sv.nworkers = 1; sv.max_batch = max_batch; sv.window_us = window_us; sv.stop = stop;
    for (auto& ur : sv.users) for (void*& u : ur) u = ctx;
    for (int k = 0; k < 5; k++) {
        int is_fc = 0;
        sv.kind[k] = pnn_model_info(ctx, kServiceWidths[k], &is_fc, nullptr, nullptr) == PNN_OK ? (is_fc ? 1 : 0) : -2;
        char tg[kTagBytes];
        sv.tag[k] = pnn_arithmetic_tag(ctx, tg, sizeof tg) == PNN_OK ? tg : "unknown";
    }
    return sv.run(socket_path, stats);
}

int pnn_service_run_table(const char* socket_path, const char* model_table_path, int use_pair, float mean, int device, int max_batch,
                          int window_us, volatile int* stop, long* stats)
{
    if (!stop || max_batch < 1 || window_us < 0 || !model_table_path) return PNN_E_ARG;
    // One context per width, each with that width's model only: five worker threads, five streams on one GPU.
    // PNN_SERVICE_REPLICAS = R (1 .. 4) makes it R contexts and workers per width taking alternate batches -- built in round 4 because
    // the 4x4 / 8x8 workers are busy 4.2 s of a 4.9 s Kodak-size campaign (a single-block call is a chain of dependent launches of
    // ~45 us whatever its batch), measured, and left OFF: with 2 / 3 replicas a call takes 61 / 81 us instead of 43 (ten or fifteen host
    // threads launching tiny kernels contend in the runtime and on the device's queues) and the campaign 5.68 / 6.22 s instead of 5.12
    // (configs[4]: 6.97 / 8.07 instead of 5.69); same bitstreams in every case (a block's prediction does not depend on its batch).
    // Round 5, float32, replicas for the busiest widths only ("2,1,1,1,1" / "2,2,1,1,1"): configs[3] 6.76 / 6.69 and 6.52 / 6.38 s against
    // 6.25 / 5.89 s on the same box -- a 4x4 call 67-80 us instead of 52, and the requests of the other widths wait longer.
    int widths[64], pairs[64], chans[64];
    const char* paths[64];
    const int n = pnn_parse_model_table(model_table_path, widths, pairs, chans, paths, 64);
    if (n < 0) return n;
    bool have_pair = false;
    for (int i = 0; i < n; i++) have_pair |= pairs[i] != 0;
    const int want_pair = (have_pair && use_pair) ? 1 : 0;       // TComPrediction.cpp:156
    std::string dir(model_table_path);
    const size_t slash = dir.find_last_of('/');
    dir = slash == std::string::npos ? std::string(".") : dir.substr(0, slash);
    int nrep[5] = {1, 1, 1, 1, 1};                   // PNN_SERVICE_REPLICAS = "R" (every width) or "R4,R8,R16,R32,R64"
    if (const char* e = getenv("PNN_SERVICE_REPLICAS")) {
        int v[5], got = sscanf(e, "%d,%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3], &v[4]);
        for (int k = 0; k < 5; k++) nrep[k] = std::max(1, std::min(Server::kMaxRep, got == 5 ? v[k] : got >= 1 ? v[0] : 1));
    }
    pnn_ctx* ctxs[5][Server::kMaxRep] = {};
    void* qs[4] = {nullptr, nullptr, nullptr, nullptr};   // streams on four different hardware queues (see below)
    int nqs = 0;
    static const int kWidths[5] = {4, 8, 16, 32, 64};
    int rc = PNN_OK;
    for (int k = 0; k < 5 && rc == PNN_OK; k++) {
        const char* hit = nullptr;
        for (int i = 0; i < n; i++) if (widths[i] == kWidths[k] && pairs[i] == want_pair && chans[i] == 0) hit = paths[i];   // later lines win
        if (!hit) { rc = PNN_E_MODEL; break; }
        std::string p(hit);
        if (!p.empty() && p[0] != '/') {
            FILE* f = fopen((dir + "/" + p).c_str(), "rb");
            if (f) { fclose(f); p = dir + "/" + p; }
        }
        for (int r = 0; r < nrep[k] && rc == PNN_OK; r++) {
            rc = pnn_create_empty(&ctxs[k][r], mean, device);
            if (rc == PNN_OK) rc = pnn_load_model_file(ctxs[k][r], p.c_str());
            if (rc == PNN_OK && !getenv("PNN_WAIT_SLEEP")) pnn_set_option(ctxs[k][r], "wait_sleep", 1);   // five workers that spin would hold five CPUs for the length of a campaign
            // (Captured launch chains, option "graphs", are off by default and stay off here: behind five workers the runtime's graph
            // launches contend like its kernel launches do and move their cost to a runtime thread -- configs[3], same box, 6.07 / 6.19 s
            // without against 6.02 / 6.27 s with, service CPU 24.4 / 24.8 -> 25.1 / 27.7 s, a 16x16 call 174 -> 196 us.)
            // the deep weight ring for every small launch that fits one workgroup per CU: inside a campaign the weights arrive from the
            // MALL / HBM (five nets take turns in L2), see pnn_gemm_f32_small.hip
            if (rc == PNN_OK && !getenv("PNN_F32_SMALL_DEEP")) pnn_set_option(ctxs[k][r], "f32_small_deep", 2);
            // Stream priorities per width: PNN_SERVICE_PRIORITIES = five of h / n / l (default: all normal).  Streams of one priority
            // share the runtime's few hardware queues and two busy widths on one queue serialise -- with the two FC widths on high-priority
            // streams a conv 16x16 / 32x32 call takes 114 / 198 us instead of 180 / 258 inside a configs[3] campaign, but a 4x4 call 68
            // instead of 57 and a 64x64 call 760 instead of 370, and the 4x4 worker is the one the encoders wait for: campaign walls
            // 6.3-6.4 s against 6.0-6.4 (hhlll / hhnll / hhhll against nnnnn, same box, round 5).  Left as a knob.
            if (rc == PNN_OK) {
                const char* pr = getenv("PNN_SERVICE_PRIORITIES");
                const char lvl = (pr && strlen(pr) == 5) ? pr[k] : 'n';
                if (lvl != 'n') pnn_set_option(ctxs[k][r], "stream_priority", lvl == 'h' ? -1 : 1);
            }
        }
    }
    if (rc == PNN_OK) {
        Server sv;
        // PNN_SERVICE_GROUPS = "g4,g8,g16,g32,g64" (worker thread 0..4 per width; default 0,1,2,3,4: one thread per width): fewer launching
        // threads contend less, the widths that share a thread wait for each other
        if (const char* e = getenv("PNN_SERVICE_GROUPS")) {
            int v[5];
            if (sscanf(e, "%d,%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3], &v[4]) == 5)
                for (int k = 0; k < 5; k++) { sv.group[k] = std::max(0, std::min(4, v[k])); nrep[k] = 1; }
        }
        // One hardware queue per busy width.  The runtime deals its streams onto 4 hardware queues, and five contexts' streams (behind
        // the null stream the start-up used) end up with the 16x16 and 32x32 workers on ONE queue and the 8x8 and 64x64 workers on
        // another: their calls run in submission order, each waiting for the other's (tools/corun_threads.cpp: a 16x16 and a 32x32 call
        // take 80 / 133 us alone and 273 us EACH side by side; more hardware queues are worse, GPU_MAX_HW_QUEUES = 8: a 4x4 call 248 us).
        // So: four streams MEASURED to sit on four different queues (pnn_streams_on_distinct_queues), one each for the widths that carry
        // the campaign (4, 8, 16), the fourth for 32 and 64 together -- on one worker thread, they would wait for each other anyway.
        // PNN_SERVICE_QUEUES=0: every context on the stream it created, as before.
        bool own_queues = true;
        if (const char* e = getenv("PNN_SERVICE_QUEUES")) own_queues = atoi(e) != 0;
        for (int k = 0; k < 5; k++) own_queues = own_queues && nrep[k] == 1;
        own_queues = own_queues && !getenv("PNN_SERVICE_GROUPS") && !getenv("PNN_SERVICE_PRIORITIES");
        if (own_queues) {
            nqs = pnn_streams_on_distinct_queues(qs, 4);
            // (said out loud: with fewer than four measured queues the service falls back to one stream and thread per width, where two
            // busy widths may share a hardware queue -- a campaign's calls then take visibly longer, and the log should say why)
            fprintf(stderr, "[pnn-service] hardware queues: %d of 4 measured distinct -> %s\n", nqs,
                    nqs == 4 ? "widths 4 / 8 / 16 on a queue of their own, 32 + 64 share the fourth" : "FALLBACK: every context on the stream it created, one thread per width");
            if (nqs == 4) {
                static const int kQueueOf[5] = {0, 1, 2, 3, 3};
                for (int k = 0; k < 5; k++) { pnn_set_option(ctxs[k][0], "stream", (long)qs[kQueueOf[k]]); sv.group[k] = kQueueOf[k]; }
            }
        }
        sv.backend = ctx_backend; sv.nworkers = 5; for (int k = 0; k < 5; k++) sv.nrep[k] = nrep[k]; sv.max_batch = max_batch; sv.window_us = window_us; sv.stop = stop;
        sv.nio = 4;                                  // socket threads (PNN_SERVICE_IO_THREADS overrides)
        for (int k = 0; k < 5; k++) {
            for (int r = 0; r < sv.nrep[k]; r++) sv.users[k][r] = ctxs[k][r];
            int is_fc = 0;
            sv.kind[k] = pnn_model_info(ctxs[k][0], kWidths[k], &is_fc, nullptr, nullptr) == PNN_OK ? (is_fc ? 1 : 0) : -2;
            char tg[kTagBytes];
            sv.tag[k] = pnn_arithmetic_tag(ctxs[k][0], tg, sizeof tg) == PNN_OK ? tg : "unknown";
        }
        rc = sv.run(socket_path, stats);
    }
    for (auto& cr : ctxs) for (pnn_ctx* c : cr) if (c) pnn_destroy(c);
    if (nqs > 0) pnn_streams_release(qs, nqs);
    return rc;
}

int pnn_client_connect(pnn_client** out, const char* socket_path)
{
    if (!out) return PNN_E_ARG;
    *out = nullptr;
    sockaddr_un addr;
    if (make_addr(socket_path, &addr)) return PNN_E_ARG;
    const int fd = socket(AF_UNIX, SOCK_STREAM, 0);
    if (fd < 0) return PNN_E_IO;
    if (connect(fd, reinterpret_cast<sockaddr*>(&addr), sizeof addr) < 0) { close(fd); return PNN_E_IO; }
    pnn_client* c = new pnn_client();
    c->fd = fd;
    const char* e = getenv("PNN_CACHE_MB");
    c->cache_bytes = (size_t)(e ? atol(e) : 64) << 20;
    // The shared-memory request path ($PNN_SERVICE_SHM=0: stay on the socket protocol): ask for a slot; the reply carries two
    // descriptors (the slot, the workers' doorbell page).  Anything unexpected leaves this client on the socket protocol.
    const char* shm = getenv("PNN_SERVICE_SHM");
    if (!shm || atoi(shm) != 0) {
        const ReqHeader h{kMagic, 4, 0u, 0u, kWantShm};
        struct { RspHeader rh; int32_t worker[5]; int32_t index; } body;
        memset(&body, 0, sizeof body);
        if (write_all(fd, &h, sizeof h)) {
            iovec iov{&body, sizeof body};
            alignas(cmsghdr) char ctl[CMSG_SPACE(2 * sizeof(int))];
            msghdr mh;
            memset(&mh, 0, sizeof mh);
            mh.msg_iov = &iov; mh.msg_iovlen = 1; mh.msg_control = ctl; mh.msg_controllen = sizeof ctl;
            ssize_t got;
            do got = recvmsg(fd, &mh, MSG_CMSG_CLOEXEC); while (got < 0 && errno == EINTR);
            int fds[2] = {-1, -1};
            for (cmsghdr* cm = CMSG_FIRSTHDR(&mh); cm; cm = CMSG_NXTHDR(&mh, cm))
                if (cm->cmsg_level == SOL_SOCKET && cm->cmsg_type == SCM_RIGHTS && cm->cmsg_len == CMSG_LEN(2 * sizeof(int))) memcpy(fds, CMSG_DATA(cm), sizeof fds);
            if (got > 0 && got < (ssize_t)sizeof body && !read_all(fd, reinterpret_cast<char*>(&body) + got, sizeof body - (size_t)got)) got = -1;
            if (got > 0 && body.rh.rc == 0 && body.rh.n_vals == 6 && fds[0] >= 0 && fds[1] >= 0 && body.index >= 0 && body.index < kShmMaxSlots) {
                void* sm = mmap(nullptr, kSlotBytes, PROT_READ | PROT_WRITE, MAP_SHARED, fds[0], 0);
                void* dm = mmap(nullptr, kDoorBytes, PROT_READ | PROT_WRITE, MAP_SHARED, fds[1], 0);
                bool ok = sm != MAP_FAILED && dm != MAP_FAILED;
                for (int i = 0; i < 5; i++) { c->worker_of[i] = body.worker[i]; ok = ok && body.worker[i] >= 0 && body.worker[i] < 5; }
                if (ok) { c->slot = sm; c->doors = dm; c->index = body.index; }
                else { if (sm != MAP_FAILED) munmap(sm, kSlotBytes); if (dm != MAP_FAILED) munmap(dm, kDoorBytes); }
            }
            for (int f : fds) if (f >= 0) close(f);
            if (got <= 0) { close(fd); delete c; return PNN_E_IO; }    // the server hung up on the hand-shake
        }
    }
    *out = c;
    return PNN_OK;
}

static int client_call(pnn_client* c, int width, const float* above, const float* left, uint32_t flags, void* vals)
{
    const uint32_t w2 = (uint32_t)(width * width);
    const ReqHeader h{kMagic, width, left ? 3 * w2 : 5 * w2, left ? 2 * w2 : 0u, flags};
    // one send per request: the server sees the whole request in one readable event
    c->buf.resize(sizeof h + ((size_t)h.n_above + h.n_left) * 4);
    memcpy(c->buf.data(), &h, sizeof h);
    memcpy(c->buf.data() + sizeof h, above, (size_t)h.n_above * 4);
    if (left) memcpy(c->buf.data() + sizeof h + (size_t)h.n_above * 4, left, (size_t)h.n_left * 4);
    ClientCacheEntry* slot = nullptr;
    const char* in = c->buf.data() + sizeof h;
    const size_t in_bytes = c->buf.size() - sizeof h;
    uint64_t hash = 0;
    if (c->cache_bytes) {
        const int wi = width == 4 ? 0 : width == 8 ? 1 : width == 16 ? 2 : width == 32 ? 3 : 4;
        auto& table = c->cache[wi][flags & 1u];
        if (table.empty()) table.resize(std::max<size_t>(16, c->cache_bytes / 10 / (in_bytes + (size_t)w2 * 4 + 64)));
        hash = hash_bytes(in, in_bytes);
        slot = &table[hash % table.size()];
        if (slot->valid && slot->hash == hash && slot->in.size() == in_bytes && !memcmp(slot->in.data(), in, in_bytes)) {
            memcpy(vals, slot->vals.data(), (size_t)w2 * 4);
            ++c->hits;
            return PNN_OK;
        }
        ++c->misses;
    }
    if (c->slot) {
        // through the slot: context in place, one doorbell, sleep on the slot's state word until the worker has written the reply
        ShmSlot* sl = static_cast<ShmSlot*>(c->slot);
        const int wk = c->worker_of[width == 4 ? 0 : width == 8 ? 1 : width == 16 ? 2 : width == 32 ? 3 : 4];
        ShmDoor& door = static_cast<ShmDoors*>(c->doors)->door[wk];
        sl->hdr = h;
        memcpy(sl->in, in, in_bytes);
        sl->posted_ns = now_ns();
        sl->client_sleeps.store(0, std::memory_order_relaxed);
        sl->state.store(kSlotPosted, std::memory_order_seq_cst);
        static_cast<ShmDoors*>(c->doors)->posted[wk][c->index >> 6].fetch_or(1ull << (c->index & 63), std::memory_order_seq_cst);
        door.bell.fetch_add(1, std::memory_order_seq_cst);
        if (door.sleeping.load(std::memory_order_seq_cst)) futex_wake(&door.bell, 1);
        int idle_rounds = 0;
        if (const uint32_t spin = static_cast<ShmDoors*>(c->doors)->spin_us.load(std::memory_order_relaxed)) {   // few clients: see ShmDoors::spin_us
            const uint64_t t0 = sl->posted_ns;
            while (sl->state.load(std::memory_order_acquire) != kSlotReady && now_ns() - t0 < (uint64_t)spin * 1000u) __builtin_ia32_pause();
        }
        for (;;) {
            uint32_t st = sl->state.load(std::memory_order_acquire);
            if (st == kSlotReady) break;
            sl->client_sleeps.store(1, std::memory_order_seq_cst);
            st = sl->state.load(std::memory_order_seq_cst);
            if (st == kSlotReady) break;
            if (futex_wait(&sl->state, st, 200000) != 0 && errno == ETIMEDOUT && ++idle_rounds >= 1) {
                // nothing for 200 ms: is the server still there?  (its end of the control socket closes when it dies or drops this client)
                pollfd pf{c->fd, POLLIN, 0};
                char probe;
                if (poll(&pf, 1, 0) > 0 && (pf.revents & (POLLHUP | POLLERR) || (pf.revents & POLLIN && recv(c->fd, &probe, 1, MSG_PEEK | MSG_DONTWAIT) == 0))) {
                    if (sl->state.load(std::memory_order_acquire) == kSlotReady) break;
                    return PNN_E_IO;
                }
            }
        }
        const RspHeader r = sl->rsp;
        if (r.rc == 0 && r.n_vals == w2) memcpy(vals, sl->out, (size_t)w2 * 4);
        sl->state.store(kSlotIdle, std::memory_order_release);
        if (r.rc != 0) return r.rc;
        if (r.n_vals != w2) return PNN_E_IO;
        if (slot) {
            slot->in.assign(in, in + in_bytes);
            slot->vals.assign(static_cast<const char*>(vals), static_cast<const char*>(vals) + (size_t)w2 * 4);
            slot->hash = hash; slot->valid = true;
        }
        return PNN_OK;
    }
    if (!write_all(c->fd, c->buf.data(), c->buf.size())) return PNN_E_IO;
    // the reply in ONE recv where the kernel has it whole (the server sends header and values with one send; an error reply is the
    // header alone, so the first recv must not insist on more than a header).  Its own buffer: `buf` still holds the request's bytes,
    // which become the cache entry's key below.
    RspHeader r;
    c->rbuf.resize(sizeof r + (size_t)w2 * 4);
    size_t got = 0;
    while (got < sizeof r) {
        const ssize_t k = recv(c->fd, c->rbuf.data() + got, c->rbuf.size() - got, 0);
        if (k == 0) return PNN_E_IO;
        if (k < 0) { if (errno == EINTR) continue; return PNN_E_IO; }
        got += (size_t)k;
    }
    memcpy(&r, c->rbuf.data(), sizeof r);
    if (r.rc != 0) return r.rc;
    if (r.n_vals != w2) return PNN_E_IO;
    if (got < c->rbuf.size() && !read_all(c->fd, c->rbuf.data() + got, c->rbuf.size() - got)) return PNN_E_IO;
    memcpy(vals, c->rbuf.data() + sizeof r, (size_t)w2 * 4);
    if (slot) {
        slot->in.assign(in, in + in_bytes);
        slot->vals.assign(static_cast<const char*>(vals), static_cast<const char*>(vals) + (size_t)w2 * 4);
        slot->hash = hash; slot->valid = true;
    }
    return PNN_OK;
}

int pnn_client_predict_pel(pnn_client* c, int width, const float* above, const float* left, int32_t* dst, int dst_stride)
{
    if (!c || !above || !dst || !valid_width(width) || dst_stride < width) return PNN_E_ARG;
    if (dst_stride == width) return client_call(c, width, above, left, 0u, dst);
    std::vector<int32_t> pel((size_t)width * width);
    const int rc = client_call(c, width, above, left, 0u, pel.data());
    if (rc) return rc;
    for (int y = 0; y < width; y++) memcpy(dst + (size_t)y * dst_stride, pel.data() + (size_t)y * width, (size_t)width * 4);
    return PNN_OK;
}

int pnn_client_predict_f32(pnn_client* c, int width, const float* above, const float* left, float* out)
{
    if (!c || !above || !out || !valid_width(width)) return PNN_E_ARG;
    return client_call(c, width, above, left, kWantF32, out);
}

int pnn_client_arithmetic_tag(pnn_client* c, int width, char* out, size_t bytes)
{
    if (!c || !out || bytes == 0 || !valid_width(width)) return PNN_E_ARG;
    const ReqHeader h{kMagic, width, 0u, 0u, kWantTag};
    if (!write_all(c->fd, &h, sizeof h)) return PNN_E_IO;
    RspHeader r;
    if (!read_all(c->fd, &r, sizeof r)) return PNN_E_IO;
    if (r.rc != 0) return r.rc;
    if (r.n_vals != kTagBytes / 4) return PNN_E_IO;
    char body[kTagBytes];
    if (!read_all(c->fd, body, sizeof body)) return PNN_E_IO;
    body[kTagBytes - 1] = 0;
    snprintf(out, bytes, "%s", body);
    return PNN_OK;
}

int pnn_client_cache_stats(const pnn_client* c, long* hits, long* misses)
{
    if (!c) return PNN_E_ARG;
    if (hits) *hits = c->hits;
    if (misses) *misses = c->misses;
    return PNN_OK;
}

void pnn_client_close(pnn_client* c)
{
    if (!c) return;
    if (c->slot) munmap(c->slot, kSlotBytes);
    if (c->doors) munmap(c->doors, kDoorBytes);
    close(c->fd);
    delete c;
}

}  // extern "C"
