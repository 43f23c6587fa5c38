// What the batching service's five width workers do to each other on ONE GPU, without sockets: one host thread and one context per width
// (as pnn_service_run_table keeps them), each issuing back-to-back host calls (pnn_predict_f32_pel) of a typical campaign batch; per
// width the time per call alone and with the other widths running.  tools/corun_threads.py builds the models and runs it.
//   g++ -O2 -std=c++17 -Iinclude tools/corun_threads.cpp -o tools/_bin/corun_threads -Lcontext_adaptive_neural_network_based_prediction_amd -lpnn_hip -lpthread
//   tools/_bin/corun_threads <model table> <precision 0|1> <seconds> [option=value ...]
#include "pnn_hip.h"

#include <time.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

struct Case { int w, n; };
static const Case kCases[5] = {{4, 6}, {8, 3}, {16, 2}, {32, 1}, {64, 1}};   // configs[3]'s mean batches per width

int main(int argc, char** argv)
{
    if (argc < 4) { fprintf(stderr, "usage: corun_threads <table> <precision> <seconds> [option=value ...]\n"); return 2; }
    const int precision = atoi(argv[2]);
    const double seconds = atof(argv[3]);
    int widths[64], pairs[64], chans[64];
    const char* paths[64];
    const int n = pnn_parse_model_table(argv[1], widths, pairs, chans, paths, 64);
    if (n < 5) { fprintf(stderr, "bad table\n"); return 1; }
    std::vector<std::string> files(5);
    for (int i = 0; i < n; i++) for (int k = 0; k < 5; k++) if (widths[i] == kCases[k].w && !pairs[i] && !chans[i]) files[k] = paths[i];
    pnn_ctx* ctx[5];
    int is_fc[5];
    double gap_spin_us = 0, gap_sleep_us = 0;
    for (int k = 0; k < 5; k++) {
        if (pnn_create_empty(&ctx[k], 117.8952234192841f, 0) || pnn_load_model_file(ctx[k], files[k].c_str())) { fprintf(stderr, "model %d: %s\n", kCases[k].w, pnn_last_error(ctx[k])); return 1; }
        pnn_set_option(ctx[k], "precision", precision);
        for (int a = 4; a < argc; a++) {
            std::string s(argv[a]);
            const size_t eq = s.find('=');
            if (s == "fcprio") { pnn_set_option(ctx[k], "stream_priority", k < 2 ? -1 : 1); continue; }   // widths 4 / 8 ahead of the conv widths
            if (eq != std::string::npos && s.substr(0, eq) == "gap_us") { gap_spin_us = atof(s.c_str() + eq + 1); continue; }          // between two calls of a thread: busy-wait
            if (eq != std::string::npos && s.substr(0, eq) == "gap_sleep_us") { gap_sleep_us = atof(s.c_str() + eq + 1); continue; }   // ... or nanosleep (the time of the CALLS is what is printed)
            if (eq != std::string::npos) pnn_set_option(ctx[k], s.substr(0, eq).c_str(), atol(s.c_str() + eq + 1));
        }
        pnn_model_info(ctx[k], kCases[k].w, &is_fc[k], nullptr, nullptr);
    }
    // "queues": as the batching service does -- four streams measured to sit on four hardware queues; 4, 8, 16 one each, 32 and 64 the fourth
    void* qs[4] = {nullptr, nullptr, nullptr, nullptr};
    bool own_queues = false;
    for (int a = 4; a < argc; a++) own_queues |= std::string(argv[a]) == "queues";
    if (own_queues) {
        if (pnn_streams_on_distinct_queues(qs, 4) != 4) { fprintf(stderr, "fewer than four hardware queues\n"); return 1; }
        static const int kQueueOf[5] = {0, 1, 2, 3, 3};
        for (int k = 0; k < 5; k++) pnn_set_option(ctx[k], "stream", (long)qs[kQueueOf[k]]);
        printf("# contexts on streams measured to sit on different hardware queues (4, 8, 16: one each; 32 and 64 share the fourth)\n");
    }
    auto run = [&](unsigned mask, double* us) {
        std::atomic<bool> stop{false};
        std::atomic<int> ready{0};
        int nt = 0;
        for (int k = 0; k < 5; k++) nt += (mask >> k) & 1;
        std::vector<std::thread> th;
        for (int k = 0; k < 5; k++) {
            if (!((mask >> k) & 1)) continue;
            th.emplace_back([&, k] {
                const int w = kCases[k].w, nb = kCases[k].n, w2 = w * w;
                std::mt19937 rng(7 + k);
                std::uniform_real_distribution<float> d(-118.f, 137.f);
                std::vector<float> above((size_t)nb * (is_fc[k] ? 5 : 3) * w2), left((size_t)nb * 2 * w2);
                for (float& v : above) v = d(rng);
                for (float& v : left) v = d(rng);
                std::vector<int32_t> dst((size_t)nb * w2);
                for (int i = 0; i < 100; i++) pnn_predict_f32_pel(ctx[k], w, above.data(), is_fc[k] ? nullptr : left.data(), nb, nullptr, dst.data());
                ready++;
                while (ready.load() < nt) pnn_predict_f32_pel(ctx[k], w, above.data(), is_fc[k] ? nullptr : left.data(), nb, nullptr, dst.data());
                long calls = 0;
                double in_calls = 0;
                while (!stop.load()) {
                    const auto c0 = std::chrono::steady_clock::now();
                    pnn_predict_f32_pel(ctx[k], w, above.data(), is_fc[k] ? nullptr : left.data(), nb, nullptr, dst.data());
                    const auto c1 = std::chrono::steady_clock::now();
                    in_calls += std::chrono::duration<double>(c1 - c0).count(); calls++;
                    if (gap_spin_us > 0) while (std::chrono::duration<double>(std::chrono::steady_clock::now() - c1).count() * 1e6 < gap_spin_us) {}
                    if (gap_sleep_us > 0) { timespec ts{0, (long)(gap_sleep_us * 1e3)}; nanosleep(&ts, nullptr); }
                }
                us[k] = in_calls / (double)std::max(calls, 1L) * 1e6;
            });
        }
        while (ready.load() < nt) std::this_thread::sleep_for(std::chrono::milliseconds(5));
        std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
        stop = true;
        for (auto& t : th) t.join();
    };
    double alone[5] = {0}, all[5] = {0}, small[5] = {0}, no64[5] = {0};
    for (int k = 0; k < 5; k++) run(1u << k, alone);
    run(31u, all);
    run(3u, small);
    run(15u, no64);
    for (int k = 0; k < 5; k++)
        printf("%-5s width %2d %-4s %d blocks per call: alone %6.1f us | all five widths %6.1f us (x %.2f) | 4 + 8 only %6.1f | all but 64 %6.1f\n", precision ? "split" : "f32", kCases[k].w,
               is_fc[k] ? "FC" : "conv", kCases[k].n, alone[k], all[k], all[k] / alone[k], small[k], no64[k]);
    for (int k = 0; k < 5; k++) pnn_destroy(ctx[k]);
    if (own_queues) pnn_streams_release(qs, 4);
    return 0;
}
