// Cross-process batching service (include/pnn_service.h): Unix-domain socket server that coalesces the single-block
// PNN requests of many encoder processes into batched calls, and the matching client stub.  Plain POSIX, no HIP here.
#include "pnn_service.h"

#include <errno.h>
#include <poll.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/un.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <vector>

namespace {

constexpr uint32_t kMagic = 0x314e4e50u;   // "PNN1"
struct ReqHeader { uint32_t magic; int32_t width; uint32_t n_above, n_left; };   // followed by the floats
struct RspHeader { int32_t rc; uint32_t n_pel; };                                // followed by n_pel int32

bool read_all(int fd, void* buf, size_t n)
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

bool write_all(int fd, const void* buf, size_t n)
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

struct Pending { int fd; std::vector<float> above, left; };

int make_addr(const char* path, sockaddr_un* a)
{
    memset(a, 0, sizeof *a);
    a->sun_family = AF_UNIX;
    if (!path || strlen(path) >= sizeof a->sun_path) return PNN_E_ARG;
    strcpy(a->sun_path, path);
    return PNN_OK;
}

int ctx_backend(void* user, int width, const float* above, const float* left, int n, int32_t* dst)
{
    return pnn_predict_pel(static_cast<pnn_ctx*>(user), width, above, left, n, dst, width);
}

}  // namespace

struct pnn_client { int fd; std::vector<int32_t> pel; };

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
    std::vector<int> clients;
    long served = 0, calls = 0, largest = 0, accepted = 0;
    // pending requests per (width, has_left): a batch shares one model
    std::map<std::pair<int, int>, std::vector<Pending>> pending;
    size_t n_pending = 0;
    auto drop = [&](int fd) {
        close(fd);
        clients.erase(std::remove(clients.begin(), clients.end(), fd), clients.end());
    };
    auto take = [&](int fd) {                        // one request from a readable client; false = client gone / bad request
        ReqHeader h;
        if (!read_all(fd, &h, sizeof h)) return false;
        const long w2 = (long)h.width * h.width;
        if (h.magic != kMagic || !valid_width(h.width) || !((h.n_above == 5 * w2 && h.n_left == 0) || (h.n_above == 3 * w2 && h.n_left == 2 * w2)))
            return false;
        Pending p;
        p.fd = fd;
        p.above.resize(h.n_above);
        p.left.resize(h.n_left);
        if (!read_all(fd, p.above.data(), h.n_above * 4) || (h.n_left && !read_all(fd, p.left.data(), h.n_left * 4))) return false;
        pending[{h.width, h.n_left ? 1 : 0}].push_back(std::move(p));
        ++n_pending;
        return true;
    };
    auto poll_once = [&](int timeout_ms) {           // accept + read whatever is ready; returns number of requests taken
        std::vector<pollfd> fds(clients.size() + 1);
        fds[0] = {lfd, POLLIN, 0};
        for (size_t i = 0; i < clients.size(); i++) fds[i + 1] = {clients[i], POLLIN, 0};
        const int r = poll(fds.data(), fds.size(), timeout_ms);
        if (r <= 0) return 0;
        int took = 0;
        std::vector<int> gone;
        for (size_t i = 1; i < fds.size(); i++) {
            if (!(fds[i].revents & (POLLIN | POLLHUP | POLLERR))) continue;
            bool has_request = false;                // a client with a request in flight sends nothing more until it is answered
            for (auto& kv : pending)
                for (const Pending& p : kv.second) has_request |= p.fd == fds[i].fd;
            if (has_request) { if (fds[i].revents & (POLLHUP | POLLERR)) gone.push_back(fds[i].fd); continue; }
            if (take(fds[i].fd)) ++took; else gone.push_back(fds[i].fd);
        }
        if (fds[0].revents & POLLIN) {
            const int cfd = accept(lfd, nullptr, nullptr);
            if (cfd >= 0) { clients.push_back(cfd); ++accepted; }
        }
        for (int fd : gone) {
            for (auto& kv : pending) {
                auto& v = kv.second;
                const size_t before = v.size();
                v.erase(std::remove_if(v.begin(), v.end(), [fd](const Pending& p) { return p.fd == fd; }), v.end());
                n_pending -= before - v.size();
            }
            drop(fd);
        }
        return took;
    };
    std::vector<float> above, left;
    std::vector<int32_t> dst;
    while (!*stop) {
        poll_once(n_pending ? 0 : 50);
        if (!n_pending) continue;
        if (window_us > 0 && (long)n_pending < max_batch) {   // give stragglers a moment to join the batch
            const auto t0 = std::chrono::steady_clock::now();
            for (;;) {
                const long left_us = window_us - (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
                if (left_us <= 0 || (long)n_pending >= max_batch) break;
                poll_once((int)std::max<long>(1, left_us / 1000));
            }
        }
        for (auto& kv : pending) {
            std::vector<Pending>& v = kv.second;
            const int w = kv.first.first;
            const size_t w2 = (size_t)w * w;
            while (!v.empty()) {
                const size_t n = std::min<size_t>(v.size(), (size_t)max_batch);
                const size_t na = v[0].above.size(), nl = v[0].left.size();
                above.resize(n * na); left.resize(n * nl); dst.resize(n * w2);
                for (size_t i = 0; i < n; i++) {
                    memcpy(above.data() + i * na, v[i].above.data(), na * 4);
                    if (nl) memcpy(left.data() + i * nl, v[i].left.data(), nl * 4);
                }
                const int rc = backend(user, w, above.data(), nl ? left.data() : nullptr, (int)n, dst.data());
                ++calls;
                largest = std::max<long>(largest, (long)n);
                for (size_t i = 0; i < n; i++) {
                    const RspHeader rh{rc, rc == 0 ? (uint32_t)w2 : 0u};
                    const bool ok = write_all(v[i].fd, &rh, sizeof rh) && (rc != 0 || write_all(v[i].fd, dst.data() + i * w2, w2 * 4));
                    if (!ok) drop(v[i].fd);
                    ++served;
                }
                v.erase(v.begin(), v.begin() + (long)n);
                n_pending -= n;
            }
        }
    }
    for (int fd : clients) close(fd);
    close(lfd);
    unlink(socket_path);
    if (stats) { stats[0] = served; stats[1] = calls; stats[2] = largest; stats[3] = accepted; }
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
    *out = new pnn_client{fd, {}};
    return PNN_OK;
}

int pnn_client_predict_pel(pnn_client* c, int width, const float* above, const float* left, int32_t* dst, int dst_stride)
{
    if (!c || !above || !dst || !valid_width(width) || dst_stride < width) return PNN_E_ARG;
    const uint32_t w2 = (uint32_t)(width * width);
    const ReqHeader h{kMagic, width, left ? 3 * w2 : 5 * w2, left ? 2 * w2 : 0u};
    if (!write_all(c->fd, &h, sizeof h) || !write_all(c->fd, above, (size_t)h.n_above * 4) || (left && !write_all(c->fd, left, (size_t)h.n_left * 4)))
        return PNN_E_IO;
    RspHeader r;
    if (!read_all(c->fd, &r, sizeof r)) return PNN_E_IO;
    if (r.rc != 0) return r.rc;
    if (r.n_pel != w2) return PNN_E_IO;
    c->pel.resize(w2);
    if (!read_all(c->fd, c->pel.data(), (size_t)w2 * 4)) return PNN_E_IO;
    for (int y = 0; y < width; y++) memcpy(dst + (size_t)y * dst_stride, c->pel.data() + (size_t)y * width, (size_t)width * 4);
    return PNN_OK;
}

void pnn_client_close(pnn_client* c)
{
    if (!c) return;
    close(c->fd);
    delete c;
}

}  // extern "C"
