// Cross-process batching service (include/pnn_service.h): Unix-domain socket server that coalesces the single-block
// PNN requests of many encoder processes into batched calls, and the matching client stub.  Plain POSIX, no HIP here.
//
// The server is one thread and never blocks on a client: sockets are non-blocking, every client has its own receive and
// send buffer, a request is queued for the GPU only once its last byte has arrived, and a client whose reply cannot be
// written for kStallMs (it stopped reading) or that sends a malformed header is dropped without disturbing the others.
// The batching window is timed with ppoll (microsecond resolution).
#include "pnn_service.h"

#include <errno.h>
#include <fcntl.h>
#include <poll.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/un.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <vector>

namespace {

constexpr uint32_t kMagic = 0x324e4e50u;   // "PNN2"
constexpr uint32_t kWantF32 = 1u;          // flags bit 0: reply with the float prediction (frozen-graph output) instead of Pel
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
    if (h.magic != kMagic || !valid_width(h.width) || (h.flags & ~kWantF32)) return false;
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

struct Pending { int fd; bool want_f32; std::vector<float> above, left; };

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

int ctx_backend(void* user, int width, const float* above, const float* left, int n, int32_t* dst, float* out_f32)
{
    return pnn_predict_f32_pel(static_cast<pnn_ctx*>(user), width, above, left, n, out_f32, dst);
}

}  // namespace

// Client-side prediction cache: HM's rate-distortion search asks for the same block with the same context several times
// (SURVEY 3.2); a repeated request is answered here without a round trip.  Direct-mapped per (width, reply kind), exact
// match on the input bytes -- the same scheme as the in-process cache of pnn_abi.cpp ("cache_mb").
struct ClientCacheEntry { uint64_t hash = 0; bool valid = false; std::vector<char> in, vals; };
struct pnn_client {
    int fd;
    std::vector<char> buf;
    size_t cache_bytes = 0;                           // 0 = off
    std::vector<ClientCacheEntry> cache[5][2];
    long hits = 0, misses = 0;
};

static uint64_t fnv1a64(const void* data, size_t bytes)
{
    const unsigned char* p = static_cast<const unsigned char*>(data);
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < bytes; i++) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

extern "C" {

int pnn_service_run_backend(const char* socket_path, pnn_service_backend backend, void* user, int max_batch, int window_us,
                            volatile int* stop, long* stats)
{
    if (!backend || !stop || max_batch < 1 || window_us < 0) return PNN_E_ARG;
    sockaddr_un addr;
    if (make_addr(socket_path, &addr)) return PNN_E_ARG;
    const int lfd = socket(AF_UNIX, SOCK_STREAM, 0);
    if (lfd < 0) return PNN_E_IO;
    unlink(socket_path);
    if (bind(lfd, reinterpret_cast<sockaddr*>(&addr), sizeof addr) < 0 || listen(lfd, 256) < 0) { close(lfd); return PNN_E_IO; }
    set_nonblocking(lfd);
    std::map<int, Client> clients;
    long served = 0, calls = 0, largest = 0, accepted = 0, dropped = 0;
    // complete requests per (width, has_left): a batch shares one model
    std::map<std::pair<int, int>, std::vector<Pending>> pending;
    size_t n_pending = 0;

    auto drop = [&](int fd) {
        for (auto& kv : pending) {
            auto& v = kv.second;
            const size_t before = v.size();
            v.erase(std::remove_if(v.begin(), v.end(), [fd](const Pending& p) { return p.fd == fd; }), v.end());
            n_pending -= before - v.size();
        }
        close(fd);
        clients.erase(fd);
        ++dropped;
    };
    // Moves whatever the socket holds into the client's buffer; queues the request when it is complete.
    // false = client gone or protocol violation.
    auto receive = [&](Client& c) {
        for (;;) {
            size_t want = sizeof(ReqHeader);
            if (c.rx.size() >= sizeof(ReqHeader)) {
                ReqHeader h;
                memcpy(&h, c.rx.data(), sizeof h);
                if (!valid_header(h)) return false;
                want = sizeof h + ((size_t)h.n_above + h.n_left) * 4;
                if (c.rx.size() == want) {
                    if (c.in_flight) return false;            // one outstanding request per client
                    Pending p;
                    p.fd = c.fd; p.want_f32 = (h.flags & kWantF32) != 0;
                    p.above.resize(h.n_above); p.left.resize(h.n_left);
                    memcpy(p.above.data(), c.rx.data() + sizeof h, (size_t)h.n_above * 4);
                    if (h.n_left) memcpy(p.left.data(), c.rx.data() + sizeof h + (size_t)h.n_above * 4, (size_t)h.n_left * 4);
                    pending[{h.width, h.n_left ? 1 : 0}].push_back(std::move(p));
                    ++n_pending;
                    c.in_flight = true;
                    c.rx.clear();
                    return true;
                }
            }
            const size_t have = c.rx.size();
            c.rx.resize(want);
            const ssize_t r = recv(c.fd, c.rx.data() + have, want - have, 0);
            if (r <= 0) {
                c.rx.resize(have);
                if (r < 0 && (errno == EAGAIN || errno == EWOULDBLOCK || errno == EINTR)) return true;   // rest comes later
                return false;                                 // closed or broken
            }
            c.rx.resize(have + (size_t)r);
        }
    };
    auto flush = [&](Client& c) {                             // false = broken
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
    auto poll_once = [&](long timeout_us) {
        std::vector<pollfd> fds;
        fds.reserve(clients.size() + 1);
        fds.push_back({lfd, POLLIN, 0});
        for (auto& kv : clients) {
            short ev = 0;
            if (!kv.second.in_flight) ev |= POLLIN;           // an answered client may send its next request
            if (!kv.second.tx.empty()) ev |= POLLOUT;
            fds.push_back({kv.first, ev, 0});
        }
        timespec ts{timeout_us / 1000000, (timeout_us % 1000000) * 1000};
        const int r = ppoll(fds.data(), fds.size(), &ts, nullptr);
        std::vector<int> gone;
        if (r > 0) {
            for (size_t i = 1; i < fds.size(); i++) {
                if (!fds[i].revents) continue;
                Client& c = clients[fds[i].fd];
                bool ok = true;
                if (fds[i].revents & POLLOUT) ok = flush(c);
                if (ok && (fds[i].revents & POLLIN)) ok = receive(c);
                else if (ok && (fds[i].revents & (POLLHUP | POLLERR | POLLNVAL))) ok = false;
                if (!ok) gone.push_back(fds[i].fd);
            }
            if (fds[0].revents & POLLIN) {
                for (;;) {
                    const int cfd = accept(lfd, nullptr, nullptr);
                    if (cfd < 0) break;
                    set_nonblocking(cfd);
                    Client c;
                    c.fd = cfd;
                    ucred cred;
                    socklen_t len = sizeof cred;
                    if (getsockopt(cfd, SOL_SOCKET, SO_PEERCRED, &cred, &len) == 0) c.pid = (int)cred.pid;
                    clients.emplace(cfd, std::move(c));
                    ++accepted;
                }
            }
        }
        const auto now = Clock::now();
        for (auto& kv : clients)
            if (!kv.second.tx.empty() && std::chrono::duration_cast<std::chrono::milliseconds>(now - kv.second.tx_since).count() > kStallMs)
                gone.push_back(kv.first);
        std::sort(gone.begin(), gone.end());
        gone.erase(std::unique(gone.begin(), gone.end()), gone.end());
        for (int fd : gone) drop(fd);
    };

    std::vector<float> above, left, out;
    std::vector<int32_t> dst;
    while (!*stop) {
        poll_once(n_pending ? 0 : 50000);
        if (!n_pending) continue;
        // Give stragglers a moment to join the batch -- unless every peer PROCESS already has a request in flight: an encoder is
        // single-threaded and blocks on its request, so nobody else can arrive and waiting would only add latency.
        auto everyone_waits = [&]() {
            std::vector<int> pids, waiting;
            for (const auto& kv : clients) {
                pids.push_back(kv.second.pid);
                if (kv.second.in_flight) waiting.push_back(kv.second.pid);
            }
            std::sort(pids.begin(), pids.end());
            pids.erase(std::unique(pids.begin(), pids.end()), pids.end());
            std::sort(waiting.begin(), waiting.end());
            waiting.erase(std::unique(waiting.begin(), waiting.end()), waiting.end());
            return waiting.size() >= pids.size();
        };
        if (window_us > 0 && (long)n_pending < max_batch && !everyone_waits()) {
            const auto t0 = Clock::now();
            for (;;) {
                const long left_us = window_us - (long)std::chrono::duration_cast<std::chrono::microseconds>(Clock::now() - t0).count();
                if (left_us <= 0 || (long)n_pending >= max_batch || *stop || everyone_waits()) break;
                poll_once(left_us);
            }
        }
        for (auto& kv : pending) {
            std::vector<Pending>& v = kv.second;
            const int w = kv.first.first;
            const size_t w2 = (size_t)w * w;
            while (!v.empty()) {
                const size_t n = std::min<size_t>(v.size(), (size_t)max_batch);
                const size_t na = v[0].above.size(), nl = v[0].left.size();
                bool any_f32 = false, any_pel = false;
                above.resize(n * na); left.resize(n * nl);
                for (size_t i = 0; i < n; i++) {
                    memcpy(above.data() + i * na, v[i].above.data(), na * 4);
                    if (nl) memcpy(left.data() + i * nl, v[i].left.data(), nl * 4);
                    (v[i].want_f32 ? any_f32 : any_pel) = true;
                }
                if (any_pel) dst.resize(n * w2);
                if (any_f32) out.resize(n * w2);
                const int rc = backend(user, w, above.data(), nl ? left.data() : nullptr, (int)n, any_pel ? dst.data() : nullptr,
                                       any_f32 ? out.data() : nullptr);
                ++calls;
                largest = std::max<long>(largest, (long)n);
                for (size_t i = 0; i < n; i++) {
                    auto it = clients.find(v[i].fd);
                    ++served;
                    if (it == clients.end()) continue;
                    Client& c = it->second;
                    const RspHeader rh{rc, rc == 0 ? (uint32_t)w2 : 0u};
                    if (c.tx.empty()) c.tx_since = Clock::now();
                    const char* hp = reinterpret_cast<const char*>(&rh);
                    c.tx.insert(c.tx.end(), hp, hp + sizeof rh);
                    if (rc == 0) {
                        const char* pp = v[i].want_f32 ? reinterpret_cast<const char*>(out.data() + i * w2)
                                                       : reinterpret_cast<const char*>(dst.data() + i * w2);
                        c.tx.insert(c.tx.end(), pp, pp + w2 * 4);
                    }
                    c.in_flight = false;
                    if (!flush(c)) { close(c.fd); clients.erase(it); ++dropped; }
                }
                v.erase(v.begin(), v.begin() + (long)n);
                n_pending -= n;
            }
        }
    }
    for (auto& kv : clients) close(kv.first);
    close(lfd);
    unlink(socket_path);
    if (stats) { stats[0] = served; stats[1] = calls; stats[2] = largest; stats[3] = accepted; }
    (void)dropped;
    return PNN_OK;
}

int pnn_service_run(const char* socket_path, pnn_ctx* ctx, int max_batch, int window_us, volatile int* stop, long* stats)
{
    if (!ctx) return PNN_E_ARG;
    return pnn_service_run_backend(socket_path, ctx_backend, ctx, max_batch, window_us, stop, stats);
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
        hash = fnv1a64(in, in_bytes);
        slot = &table[hash % table.size()];
        if (slot->valid && slot->hash == hash && slot->in.size() == in_bytes && !memcmp(slot->in.data(), in, in_bytes)) {
            memcpy(vals, slot->vals.data(), (size_t)w2 * 4);
            ++c->hits;
            return PNN_OK;
        }
        ++c->misses;
    }
    if (!write_all(c->fd, c->buf.data(), c->buf.size())) return PNN_E_IO;
    RspHeader r;
    if (!read_all(c->fd, &r, sizeof r)) return PNN_E_IO;
    if (r.rc != 0) return r.rc;
    if (r.n_vals != w2) return PNN_E_IO;
    if (!read_all(c->fd, vals, (size_t)w2 * 4)) return PNN_E_IO;
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
