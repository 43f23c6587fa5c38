// Race-detector driver of the batching service: `make -C csrc tsan` builds pnn_service.cpp + pnn_host.cpp + this file with
// -fsanitize=thread and runs it.  The PRODUCTION thread layout without a GPU -- five width workers ($PNN_SERVICE_WORKERS=5) and four I/O
// threads behind hand-rolled queues, eventfds and condition variables -- under 32 client threads x N requests of every width, input
// kind and reply kind against a stand-in backend with randomised latencies (0 ... 80 us, sleeping or spinning), with and without a
// batching window; clients that reconnect, ask for the arithmetic tag between requests, vanish with half a request sent, and die with a
// request in flight; a second server that is stopped while clients still hammer it.  Every reply is checked against its request.
// Any ThreadSanitizer report makes the process exit non-zero (TSAN_OPTIONS=halt_on_error=1 exitcode=66 in the Makefile).  CPU only.
//   usage: tsan_service <tmp dir> [requests per client = 10000]
#include "pnn_hip.h"
#include "pnn_service.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <sys/socket.h>
#include <sys/un.h>
#include <time.h>
#include <unistd.h>

extern "C" {
// the GPU entry points pnn_service.cpp references; never called here (pnn_service_run_backend gets a stand-in)
int pnn_predict_f32_pel(pnn_ctx*, int, const float*, const float*, int, float*, int32_t*) { return PNN_E_HIP; }
int pnn_model_info(const pnn_ctx*, int, int*, int*, long*) { return PNN_E_MODEL; }
int pnn_arithmetic_tag(const pnn_ctx*, char* out, size_t bytes) { snprintf(out, bytes, "none"); return PNN_OK; }
int pnn_create_empty(pnn_ctx**, float, int) { return PNN_E_HIP; }
int pnn_load_model_file(pnn_ctx*, const char*) { return PNN_E_HIP; }
int pnn_set_option(pnn_ctx*, const char*, long) { return PNN_E_HIP; }
int pnn_streams_on_distinct_queues(void**, int) { return 0; }
void pnn_streams_release(void**, int) {}
void pnn_destroy(pnn_ctx*) {}
}
namespace pnn { void set_create_error(const std::string&) {} }

#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "CHECK failed at line %d: %s\n", __LINE__, #cond); exit(1); } } while (0)

static unsigned rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

static std::atomic<long> g_backend_calls{0}, g_backend_blocks{0};
static std::atomic<int> g_concurrent{0}, g_max_concurrent{0};

// block i of a batch: every output value = the sum of its inputs (exact in float for these small integers) + its position
static int sum_backend(void*, int width, const float* above, const float* left, int n, int32_t* dst, float* out)
{
    const int now = ++g_concurrent;
    int seen = g_max_concurrent.load();
    while (now > seen && !g_max_concurrent.compare_exchange_weak(seen, now)) {}
    const int w2 = width * width, na = (left ? 3 : 5) * w2, nl = left ? 2 * w2 : 0;
    thread_local unsigned seed = 12345u + (unsigned)width;
    const unsigned us = rnd(seed) % 81;                      // 0 ... 80 us: a GPU call of a few blocks
    if (us) {
        if (rnd(seed) & 1) { timespec ts{0, (long)us * 1000L}; nanosleep(&ts, nullptr); }
        else { timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0); do clock_gettime(CLOCK_MONOTONIC, &t1); while ((t1.tv_sec - t0.tv_sec) * 1000000L + (t1.tv_nsec - t0.tv_nsec) / 1000 < (long)us); }
    }
    for (int b = 0; b < n; b++) {
        float s = 0.f;
        for (int i = 0; i < na; i++) s += above[(size_t)b * na + i];
        for (int i = 0; i < nl; i++) s += left[(size_t)b * nl + i];
        for (int i = 0; i < w2; i++) {
            if (dst) dst[(size_t)b * w2 + i] = (int32_t)s + i;
            if (out) out[(size_t)b * w2 + i] = s + 0.5f * (float)i;
        }
    }
    g_backend_calls++; g_backend_blocks += n;
    --g_concurrent;
    return 0;
}

static int raw_connect(const std::string& sock)
{
    sockaddr_un addr;
    memset(&addr, 0, sizeof addr);
    addr.sun_family = AF_UNIX;
    strcpy(addr.sun_path, sock.c_str());
    const int fd = socket(AF_UNIX, SOCK_STREAM, 0);
    for (int t = 0; t < 1000 && connect(fd, (sockaddr*)&addr, sizeof addr) != 0; t++) usleep(2000);
    return fd;
}

static void campaign(const char* dir, int window_us, int per_client, bool stop_under_load)
{
    const std::string sock = std::string(dir) + "/pnn_tsan_" + std::to_string(window_us) + (stop_under_load ? "s" : "") + ".sock";
    volatile int stop = 0;
    long stats[4] = {0, 0, 0, 0};
    int rc_server = -99;
    g_backend_calls = 0; g_backend_blocks = 0; g_max_concurrent = 0;
    std::thread server([&] { rc_server = pnn_service_run_backend(sock.c_str(), sum_backend, nullptr, 16, window_us, &stop, stats); });
    std::atomic<int> bad{0};
    std::atomic<long> ok_requests{0};
    // three clients in four on the shared-memory request path (a slot per client, the workers' doorbells), every fourth on the socket
    // protocol: connected here, one after the other, because the choice is an environment variable read at connect
    pnn_client* conn[32];
    for (int k = 0; k < 32; k++) {
        setenv("PNN_SERVICE_SHM", (k & 3) == 3 ? "0" : "1", 1);
        conn[k] = nullptr;
        for (int t = 0; t < 1000 && pnn_client_connect(&conn[k], sock.c_str()) != 0; t++) usleep(2000);
    }
    setenv("PNN_SERVICE_SHM", "1", 1);
    auto client = [&](int k) {
        pnn_client* c = conn[k];
        if (!c) { bad++; return; }
        unsigned seed = 100 + k;
        std::vector<float> a, l, f32;
        std::vector<int32_t> pel;
        for (int it = 0; it < per_client; it++) {
            const int w = 4 << (rnd(seed) % 5), w2 = w * w;
            const bool conv = (rnd(seed) & 1) != 0;            // every width with both input kinds (a generic backend accepts either)
            a.resize((conv ? 3 : 5) * w2); l.resize(conv ? 2 * w2 : 0); pel.assign(w2, -1); f32.assign(w2, -1.f);
            float s = 0.f;
            for (auto& v : a) { v = (float)(rnd(seed) % 3); s += v; }
            for (auto& v : l) { v = (float)(rnd(seed) % 3); s += v; }
            const bool want_f32 = (rnd(seed) & 1) != 0;
            const int rc = want_f32 ? pnn_client_predict_f32(c, w, a.data(), conv ? l.data() : nullptr, f32.data())
                                    : pnn_client_predict_pel(c, w, a.data(), conv ? l.data() : nullptr, pel.data(), w);
            if (rc != 0) {                                      // the server is being stopped under this client's feet: expected there only
                if (!stop_under_load) bad++;
                break;
            }
            for (int i = 0; i < w2; i++) if (want_f32 ? f32[i] != s + 0.5f * i : pel[i] != (int32_t)s + i) { bad++; break; }
            ok_requests++;
            if (it % 97 == 11) {                                // the arithmetic tag between two requests
                char tag[64];
                const int trc = pnn_client_arithmetic_tag(c, w, tag, sizeof tag);
                if (trc != 0 ? !stop_under_load : strcmp(tag, "backend:unspecified") != 0) bad++;
                if (trc != 0) break;
            }
            if (it % 1000 == 999 && (k & 3) == 0) {             // a quarter of the clients reconnect now and then (an encoder ends, the next starts)
                pnn_client_close(c);
                c = nullptr;
                if (pnn_client_connect(&c, sock.c_str()) != 0) { if (!stop_under_load) bad++; return; }
            }
        }
        if (c) pnn_client_close(c);
    };
    std::vector<std::thread> ts;
    for (int k = 0; k < 32; k++) ts.emplace_back(client, k);
    // the misbehaving ones, all along: half a request then gone; a whole request, then gone before the reply (dies with a request in flight);
    // a malformed header
    std::thread vandals([&] {
        unsigned seed = 7;
        for (int round = 0; round < (stop_under_load ? 20 : 200); round++) {
            const int kind = round % 3;
            const int fd = raw_connect(sock);
            if (fd < 0) continue;
            const unsigned hdr[5] = {kind == 2 ? 0x12345678u : 0x324e4e50u, 4u, 80u, 0u, 0u};
            (void)!write(fd, hdr, 20);
            if (kind == 0) { float half[40] = {0}; (void)!write(fd, half, sizeof half); usleep(rnd(seed) % 300); }
            if (kind == 1) { float all[80] = {0}; (void)!write(fd, all, sizeof all); usleep(rnd(seed) % 60); }   // ... the reply finds nobody
            close(fd);
            usleep(500 + rnd(seed) % 2000);
        }
    });
    if (stop_under_load) { usleep(300000); __atomic_store_n(const_cast<int*>(&stop), 1, __ATOMIC_RELEASE); }
    for (auto& t : ts) t.join();
    vandals.join();
    __atomic_store_n(const_cast<int*>(&stop), 1, __ATOMIC_RELEASE);
    server.join();
    CHECK(rc_server == 0);
    CHECK(bad == 0);
    if (!stop_under_load) CHECK(ok_requests == 32L * per_client);
    CHECK(stats[0] >= ok_requests.load());                      // + the vandals' complete requests
    CHECK(g_max_concurrent.load() >= 2);                        // the width workers really ran the backend side by side
    printf("tsan_service: window %3d us%s: %ld requests checked, %ld backend calls (%.2f blocks each), up to %d backend calls at once, %ld clients accepted\n", window_us,
           stop_under_load ? ", stopped under load" : "", ok_requests.load(), g_backend_calls.load(), (double)g_backend_blocks.load() / (double)(g_backend_calls.load() ? g_backend_calls.load() : 1),
           g_max_concurrent.load(), stats[3]);
}

int main(int argc, char** argv)
{
    const char* dir = argc > 1 ? argv[1] : "/tmp";
    const int per_client = argc > 2 ? atoi(argv[2]) : 10000;
    setenv("PNN_SERVICE_WORKERS", "5", 1);                      // one worker per width + four I/O threads: pnn_service_run_table's layout
    setenv("PNN_CACHE_MB", "0", 1);                             // every request reaches the server
    campaign(dir, 0, per_client, false);
    campaign(dir, 200, per_client / 4, false);
    campaign(dir, 0, per_client, true);
    printf("tsan_service: ok\n");
    return 0;
}
