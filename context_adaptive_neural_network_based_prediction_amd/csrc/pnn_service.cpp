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
#include <poll.h>
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
#include <condition_variable>
#include <map>
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
struct ReqHeader { uint32_t magic; int32_t width; uint32_t n_above, n_left, flags; };   // followed by the floats
struct RspHeader { int32_t rc; uint32_t n_vals; };                                       // followed by n_vals int32 / float

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
    if (h.magic != kMagic || !valid_width(h.width) || (h.flags & ~(kWantF32 | kWantTag))) return false;
    if (h.flags & kWantTag) return h.flags == kWantTag && h.n_above == 0 && h.n_left == 0;
    const uint32_t w2 = (uint32_t)(h.width * h.width);
    return (h.n_above == 5 * w2 && h.n_left == 0) || (h.n_above == 3 * w2 && h.n_left == 2 * w2);
}

struct Client {
    int fd = -1;
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

    struct Req { uint64_t id; int width; bool want_f32; std::vector<float> above, left; Clock::time_point t_in; };
    struct Reply { uint64_t id; std::vector<char> bytes; Clock::time_point t_in; int k; };
    // Locks: one per worker queue (with its condition variable), one per I/O thread's reply / new-connection queues, one for
    // the peer accounting (only kept when a batching window is set) and the statistics.  (A single server-wide mutex was taken
    // five times per request by nine threads.)
    std::mutex qmu[5];
    std::condition_variable cv[5];
    std::vector<Req> queue[5];       // by worker, under qmu[k]
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
    // one batch = requests of ONE width and ONE input kind, in arrival order; under qmu[k]
    void take(int k, std::vector<Req>& batch)
    {
        batch.clear();
        if (queue[k].empty()) return;
        const int w = queue[k][0].width;
        const bool has_left = !queue[k][0].left.empty();
        for (size_t i = 0; i < queue[k].size() && (int)batch.size() < max_batch;) {
            if (queue[k][i].width == w && !queue[k][i].left.empty() == has_left) {
                batch.push_back(std::move(queue[k][i]));
                queue[k].erase(queue[k].begin() + (long)i);
            } else {
                ++i;
            }
        }
    }
    void stage(Flight& f)
    {
        const size_t n = f.batch.size(), na = f.batch[0].above.size(), nl = f.batch[0].left.size();
        f.any_f32 = f.any_pel = false;
        f.above.resize(n * na); f.left.resize(n * nl);
        const auto now = Clock::now();
        f.waited = 0;
        for (size_t i = 0; i < n; i++) {
            memcpy(f.above.data() + i * na, f.batch[i].above.data(), na * 4);
            if (nl) memcpy(f.left.data() + i * nl, f.batch[i].left.data(), nl * 4);
            (f.batch[i].want_f32 ? f.any_f32 : f.any_pel) = true;
            f.waited += std::chrono::duration<double>(now - f.batch[i].t_in).count();
        }
    }
    // the replies of one finished batch to the I/O threads that own the connections, and the statistics
    void reply(int k, Flight& f, int rc, const int32_t* dst, const float* out, double busy)
    {
        const size_t n = f.batch.size(), w2 = (size_t)f.batch[0].width * f.batch[0].width;
        const int wi = nworkers == 1 ? k : widx(f.batch[0].width);   // the statistics are kept per WIDTH (two widths may share a worker thread)
        std::vector<Reply> replies(n);
        for (size_t i = 0; i < n; i++) {
            const RspHeader rh{rc, rc == 0 ? (uint32_t)w2 : 0u};
            replies[i].id = f.batch[i].id; replies[i].t_in = f.batch[i].t_in; replies[i].k = wi;
            const char* hp = reinterpret_cast<const char*>(&rh);
            replies[i].bytes.assign(hp, hp + sizeof rh);
            if (rc == 0) {
                const char* pp = f.batch[i].want_f32 ? reinterpret_cast<const char*>(out + i * w2) : reinterpret_cast<const char*>(dst + i * w2);
                replies[i].bytes.insert(replies[i].bytes.end(), pp, pp + w2 * 4);
            }
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
    }
    // Blocks until worker k has something to do (false: the server stops).  An idle worker gives stragglers a moment to join --
    // unless every peer process already waits for an answer (an encoder is single-threaded and blocks on its request: nobody
    // else can arrive).
    bool wait_for_work(int k, std::vector<Req>& batch)
    {
        std::unique_lock<std::mutex> lk(qmu[k]);
        cv[k].wait(lk, [&] { return quit_flag.load() || !queue[k].empty(); });
        if (quit_flag.load()) return false;
        if (window_us > 0 && (int)queue[k].size() < max_batch && n_waiting_peers.load() < n_peers.load()) {
            const auto until = Clock::now() + std::chrono::microseconds(window_us);
            cv[k].wait_until(lk, until, [&] { return quit_flag.load() || (int)queue[k].size() >= max_batch || n_waiting_peers.load() >= n_peers.load(); });
            if (quit_flag.load()) return false;
        }
        // the lock was released during the window: the I/O thread may have dropped the only queued request
        // (an encoder killed or timed out mid-window) -- the batch is then empty
        take(k, batch);
        return true;
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
                        r.above.resize(h.n_above); r.left.resize(h.n_left);
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
                        cv[k].notify_one();
                        // the other workers only care when "every peer waits" has just become true (they may be sitting in their window)
                        if (nworkers > 1 && window_us > 0 && n_waiting_peers.load() >= n_peers.load())
                            for (int o = 0; o < nworkers; o++) if (o != k) { std::lock_guard<std::mutex> lk(qmu[o]); cv[o].notify_one(); }
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
        for (int k = 0; k < nworkers; k++)
            for (int r = 0; r < nrep[k]; r++) threads.emplace_back([this, k, r] { worker(k, r); });
        std::vector<std::thread> io;
        for (int t = 1; t < nio; t++) io.emplace_back([this, t, lfd, &eps] { io_loop(t, lfd, eps[t]); });
        io_loop(0, lfd, eps[0]);                     // the calling thread: listener + its share of the connections
        for (auto& th : io) th.join();
        quit_flag = true;
        for (int t = 0; t < nio; t++) for (int cfd : fresh[t]) close(cfd);   // dealt but never adopted
        for (int k = 0; k < nworkers; k++) { std::lock_guard<std::mutex> lk(qmu[k]); cv[k].notify_all(); }
        for (auto& th : threads) th.join();
        for (int t = 0; t < nio; t++) { close(eps[t]); close(wake_fd[t]); }
        close(lfd);
        unlink(socket_path);
        if (stats) { stats[0] = served; stats[1] = calls; stats[2] = largest; stats[3] = accepted.load(); }
        if (getenv("PNN_SERVICE_DEBUG"))
            for (int k = 0; k < nworkers; k++)
            {
                double rs = 0; long rn = 0;
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
    close(c->fd);
    delete c;
}

}  // extern "C"
