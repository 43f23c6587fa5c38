// Host-only load on the batching service (no GPU): what the server's own threads cost per request.  A stand-in backend that sleeps for
// the length of a GPU call; N client threads that each send a 4x4 request (80 floats), "think" for a while and repeat -- the shape of
// 24 HM encoders behind the service.  Prints requests served and the CPU seconds of every server thread (/proc/self/task/*/stat).
//   g++ -O2 -std=c++17 -Iinclude tools/service_load.cpp context_adaptive_neural_network_based_prediction_amd/csrc/pnn_host.cpp \
//       context_adaptive_neural_network_based_prediction_amd/csrc/pnn_service.cpp -o tools/_bin/service_load -lpthread
//   PNN_SERVICE_IO_THREADS=4 tools/_bin/service_load [clients=24] [seconds=3] [call_us=45] [think_us=30]
#include "pnn_hip.h"
#include "pnn_service.h"

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dirent.h>
#include <string>
#include <thread>
#include <vector>
#include <time.h>
#include <unistd.h>

extern "C" {
int pnn_predict_f32_pel(pnn_ctx*, int, const float*, const float*, int, float*, int32_t*) { return PNN_E_HIP; }
int pnn_model_info(const pnn_ctx*, int, int*, int*, long*) { return PNN_E_MODEL; }
int pnn_arithmetic_tag(const pnn_ctx*, char* o, size_t n) { snprintf(o, n, "none"); return PNN_OK; }
int pnn_create_empty(pnn_ctx**, float, int) { return PNN_E_HIP; }
int pnn_load_model_file(pnn_ctx*, const char*) { return PNN_E_HIP; }
int pnn_set_option(pnn_ctx*, const char*, long) { return PNN_E_HIP; }
int pnn_streams_on_distinct_queues(void**, int) { return 0; }
void pnn_streams_release(void**, int) {}
void pnn_destroy(pnn_ctx*) {}
}
namespace pnn { void set_create_error(const std::string&) {} }

static int g_call_us = 45;
static int sleepy_backend(void*, int width, const float*, const float*, int n, int32_t* dst, float* out)
{
    timespec ts{0, g_call_us * 1000L};
    nanosleep(&ts, nullptr);
    for (int i = 0; i < n * width * width; i++) { if (dst) dst[i] = 1; if (out) out[i] = 1.f; }
    return 0;
}

int main(int argc, char** argv)
{
    const int nclients = argc > 1 ? atoi(argv[1]) : 24;
    const double seconds = argc > 2 ? atof(argv[2]) : 3.0;
    g_call_us = argc > 3 ? atoi(argv[3]) : 45;
    const int think_us = argc > 4 ? atoi(argv[4]) : 30;
    const std::string sock = "/tmp/pnn_load_" + std::to_string(getpid()) + ".sock";
    volatile int stop = 0;
    long stats[4] = {0, 0, 0, 0};
    std::thread server([&] { pnn_service_run_backend(sock.c_str(), sleepy_backend, nullptr, 256, 0, &stop, stats); });
    std::atomic<bool> quit{false};
    std::atomic<long> done{0};
    std::vector<std::thread> cl;
    for (int k = 0; k < nclients; k++) cl.emplace_back([&, k] {
        pnn_client* c = nullptr;
        for (int t = 0; t < 500 && pnn_client_connect(&c, sock.c_str()) != 0; t++) usleep(2000);
        if (!c) return;
        std::vector<float> in(80);
        int32_t dst[16];
        unsigned x = 1 + k;
        while (!quit.load()) {
            for (auto& v : in) { x = x * 1664525u + 1013904223u; v = (float)(x >> 24); }   // never the same context twice: no cache hits
            if (pnn_client_predict_pel(c, 4, in.data(), nullptr, dst, 4) != 0) break;
            done++;
            timespec ts{0, think_us * 1000L};
            nanosleep(&ts, nullptr);
        }
        pnn_client_close(c);
    });
    usleep(300000);
    auto read_tasks = [&](std::vector<std::pair<std::string, double>>& out) {
        out.clear();
        DIR* d = opendir("/proc/self/task");
        while (dirent* e = readdir(d)) {
            if (e->d_name[0] == '.') continue;
            char path[128], comm[64] = "";
            snprintf(path, sizeof path, "/proc/self/task/%s/stat", e->d_name);
            FILE* f = fopen(path, "r");
            if (!f) continue;
            char buf[1024];
            if (fgets(buf, sizeof buf, f)) {
                char* l = strchr(buf, '('); char* r = strrchr(buf, ')');
                if (l && r) {
                    snprintf(comm, sizeof comm, "%.*s", (int)(r - l - 1), l + 1);
                    unsigned long ut = 0, st = 0;
                    sscanf(r + 2, "%*c %*d %*d %*d %*d %*d %*u %*u %*u %*u %*u %lu %lu", &ut, &st);
                    out.emplace_back(std::string(comm) + ":" + e->d_name, (double)(ut + st) / sysconf(_SC_CLK_TCK));
                }
            }
            fclose(f);
        }
        closedir(d);
    };
    std::vector<std::pair<std::string, double>> t0, t1;
    read_tasks(t0);
    const long d0 = done.load();
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
    read_tasks(t1);
    const long served = done.load() - d0;
    quit = true;
    for (auto& t : cl) t.join();
    stop = 1;
    server.join();
    double server_cpu = 0;
    printf("%ld requests in %.1f s from %d clients (%.0f k/s); backend call %d us, think %d us; %ld backend calls (%.2f per call)\n", served, seconds, nclients, served / seconds / 1e3,
           g_call_us, think_us, stats[1], stats[1] ? (double)stats[0] / stats[1] : 0.0);
    for (auto& a : t1) {
        if (a.first.compare(0, 4, "pnn-") != 0) continue;
        double before = 0;
        for (auto& b : t0) if (b.first == a.first) before = b.second;
        printf("  %-16s %.2f s CPU = %.2f us per request\n", a.first.c_str(), a.second - before, (a.second - before) / served * 1e6);
        server_cpu += a.second - before;
    }
    printf("  server threads (named pnn-*; I/O thread 0 is the unnamed thread that called the server) %.2f s CPU = %.2f us per request\n", server_cpu, server_cpu / served * 1e6);
    return 0;
}
