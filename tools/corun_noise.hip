// Round 5 diagnostic: what stretches a small FC call when other streams are busy -- the LAUNCHES of the other streams (host runtime, command
// processor) or their WORK on the CUs?  One thread issues FC 4x4 calls of six blocks back to back (pnn_predict_f32_pel); a second thread
// makes noise on its own stream: (1) empty kernels back to back -- launches, no work; (2) one long kernel per ~200 us that keeps N
// workgroups busy with dependent MFMA chains and LDS traffic -- work, hardly any launch; (3) both.  Prints the FC call's time per mode.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/corun_noise.hip -o tools/_bin/corun_noise -Lcontext_adaptive_neural_network_based_prediction_amd -lpnn_hip -Wl,-rpath,$PWD/context_adaptive_neural_network_based_prediction_amd
//   tools/_bin/corun_noise <model table> <seconds>
#include <hip/hip_runtime.h>
#include "pnn_hip.h"

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 12345) *p = 0; }
__global__ __launch_bounds__(256) void busy_kernel(float* out, int iters)
{
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (float)i;
    __syncthreads();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < iters; i++) {
        const float a = lds[(threadIdx.x + i) & 4095], b = lds[(threadIdx.x * 3 + i) & 4095];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    if (acc[0] == 12345.f) out[0] = acc[1];
}

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    const double seconds = atof(argv[2]);
    int widths[64], pairs[64], chans[64];
    const char* paths[64];
    const int n = pnn_parse_model_table(argv[1], widths, pairs, chans, paths, 64);
    std::string file;
    for (int i = 0; i < n; i++) if (widths[i] == 4 && !pairs[i] && !chans[i]) file = paths[i];
    pnn_ctx* ctx = nullptr;
    if (pnn_create_empty(&ctx, 117.8952234192841f, 0) || pnn_load_model_file(ctx, file.c_str())) { fprintf(stderr, "model: %s\n", pnn_last_error(ctx)); return 1; }
    const int nb = 6, w = 4;
    std::vector<float> in((size_t)nb * 80, 3.f);
    std::vector<int32_t> dst((size_t)nb * 16);
    hipStream_t ns;
    hipStreamCreateWithFlags(&ns, hipStreamNonBlocking);
    float* d_out;
    hipMalloc(&d_out, 64);
    const char* names[6] = {"alone", "beside empty kernels back to back (launches, no work)", "beside one long kernel on 64 workgroups (work, no launches)",
                            "beside one long kernel on 512 workgroups", "beside both (empty kernels + 64 busy workgroups)", "beside 4 threads of empty kernels"};
    for (int mode = 0; mode < 6; mode++) {
        std::atomic<bool> stop{false};
        std::vector<std::thread> noise;
        auto launcher = [&](hipStream_t s) { while (!stop.load()) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, (int*)nullptr); if (hipStreamQuery(s) == hipErrorNotReady) { /* keep a few in flight */ } } hipStreamSynchronize(s); };
        auto worker = [&](hipStream_t s, int wgs) { while (!stop.load()) { hipLaunchKernelGGL(busy_kernel, dim3(wgs), dim3(256), 0, s, d_out, 6000); hipStreamSynchronize(s); } };
        std::vector<hipStream_t> extra;
        if (mode == 1 || mode == 4) noise.emplace_back(launcher, ns);
        if (mode == 2) noise.emplace_back(worker, ns, 64);
        if (mode == 3) noise.emplace_back(worker, ns, 512);
        if (mode == 4) { hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking); extra.push_back(s2); noise.emplace_back(worker, s2, 64); }
        if (mode == 5) for (int i = 0; i < 4; i++) { hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking); extra.push_back(s2); noise.emplace_back(launcher, s2); }
        for (int i = 0; i < 300; i++) pnn_predict_f32_pel(ctx, w, in.data(), nullptr, nb, nullptr, dst.data());
        long calls = 0;
        const auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) { pnn_predict_f32_pel(ctx, w, in.data(), nullptr, nb, nullptr, dst.data()); calls++; }
        const double us = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / calls * 1e6;
        stop = true;
        for (auto& t : noise) t.join();
        for (hipStream_t s2 : extra) hipStreamDestroy(s2);
        printf("FC 4x4, 6 blocks per call, exact f32: %6.1f us per call %s\n", us, names[mode]);
        fflush(stdout);
    }
    // The same question for the launches alone: four dependent empty kernels + a wait, as one "call" -- the time inside the four launch calls
    // (host side) and the time until the last one has completed, alone and beside four threads that launch empty kernels on their own streams.
    hipStream_t ps;
    hipStreamCreateWithFlags(&ps, hipStreamNonBlocking);
    hipFunction_t fn = nullptr;
    if (hipGetFuncBySymbol(&fn, (const void*)empty_kernel) != hipSuccess) fn = nullptr;
    for (int api = 0; api < (fn ? 2 : 1); api++)
    for (int nthr = 0; nthr <= 4; nthr++) {
        std::atomic<bool> stop{false};
        std::vector<std::thread> noise;
        std::vector<hipStream_t> extra;
        auto launch = [&](hipStream_t s, int wgs) {
            int* arg = nullptr;
            if (!api) { hipLaunchKernelGGL(empty_kernel, dim3(wgs), dim3(256), 0, s, arg); return; }
            size_t sz = sizeof arg;
            void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &arg, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
            hipModuleLaunchKernel(fn, wgs, 1, 1, 256, 1, 1, 0, s, nullptr, cfg);
        };
        for (int i = 0; i < nthr; i++) { hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking); extra.push_back(s2);
            noise.emplace_back([&stop, &launch, s2] { while (!stop.load()) launch(s2, 1); hipStreamSynchronize(s2); }); }
        double in_launch = 0, total = 0; long calls = 0;
        const auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds * 0.5) {
            const auto a = std::chrono::steady_clock::now();
            for (int i = 0; i < 4; i++) launch(ps, 256);
            const auto b = std::chrono::steady_clock::now();
            hipStreamSynchronize(ps);
            const auto c = std::chrono::steady_clock::now();
            in_launch += std::chrono::duration<double>(b - a).count(); total += std::chrono::duration<double>(c - a).count(); calls++;
        }
        stop = true;
        for (auto& t : noise) t.join();
        for (hipStream_t s2 : extra) hipStreamDestroy(s2);
        printf("4 dependent empty kernels + wait (%s): %5.1f us inside the launch calls, %5.1f us until complete, beside %d threads of empty kernels\n",
               api ? "hipModuleLaunchKernel" : "hipLaunchKernelGGL   ", in_launch / calls * 1e6, total / calls * 1e6, nthr);
    }
    // The same chain as ONE hipGraphLaunch (captured once): does the runtime hand a whole chain to the queue cheaper than kernel by kernel,
    // and does it suffer less from the other threads' launches?  11 dependent kernels = a single-block call of the 16x16 net.
    for (int nk : {4, 11}) {
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        hipStreamBeginCapture(ps, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < nk; i++) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, ps, (int*)nullptr);
        if (hipStreamEndCapture(ps, &graph) != hipSuccess || hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) { printf("graph capture failed\n"); break; }
        for (int use_graph = 0; use_graph < 2; use_graph++)
        for (int nthr : {0, 4}) {
            std::atomic<bool> stop{false};
            std::vector<std::thread> noise;
            std::vector<hipStream_t> extra;
            for (int i = 0; i < nthr; i++) { hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking); extra.push_back(s2);
                noise.emplace_back([&stop, s2] { while (!stop.load()) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s2, (int*)nullptr); hipStreamSynchronize(s2); }); }
            double in_launch = 0, total = 0; long calls = 0;
            const auto t0 = std::chrono::steady_clock::now();
            while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds * 0.5) {
                const auto a = std::chrono::steady_clock::now();
                if (use_graph) hipGraphLaunch(exec, ps);
                else for (int i = 0; i < nk; i++) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, ps, (int*)nullptr);
                const auto b = std::chrono::steady_clock::now();
                while (hipStreamQuery(ps) == hipErrorNotReady) {}
                const auto c = std::chrono::steady_clock::now();
                in_launch += std::chrono::duration<double>(b - a).count(); total += std::chrono::duration<double>(c - a).count(); calls++;
            }
            stop = true;
            for (auto& t : noise) t.join();
            for (hipStream_t s2 : extra) hipStreamDestroy(s2);
            printf("%2d dependent empty kernels (%s): %5.1f us inside the launch call(s), %5.1f us until complete, beside %d threads of empty kernels\n",
                   nk, use_graph ? "one hipGraphLaunch   " : "kernel by kernel     ", in_launch / calls * 1e6, total / calls * 1e6, nthr);
        }
        hipGraphExecDestroy(exec);
        hipGraphDestroy(graph);
    }
    pnn_destroy(ctx);
    return 0;
}
