// Round 5 probe: a chain of dependent kernels written into a user-mode AQL queue by hand (HSA: hsa_queue_create, packets + doorbell)
// against the same chain through the HIP runtime (kernel by kernel, and as one hipGraphLaunch) -- host time of the submission and time
// until the last kernel has completed, alone and beside four threads that launch empty kernels through HIP on their own streams.
// What the batching service's width workers would gain from their own queues.
//   hipcc --genco --no-gpu-bundle-output --offload-arch=gfx950 tools/aql_probe_kernels.hip -o tools/_bin/aql_probe.hsaco
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/aql_probe.cpp -o tools/_bin/aql_probe -lhsa-runtime64
//   tools/_bin/aql_probe tools/_bin/aql_probe.hsaco [seconds per line]
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define HSACHK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { const char* m = nullptr; hsa_status_string(s_, &m); fprintf(stderr, "%s failed: %s\n", #x, m ? m : "?"); return 1; } } while (0)

__global__ void hip_empty(int* p) { if (p && threadIdx.x == 12345) *p = 0; }
__global__ void hip_busy(unsigned ticks, int* p)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
    if (p && threadIdx.x == 12345) *p = 0;
}

static hsa_agent_t g_gpu, g_cpu;
static bool g_have_gpu = false, g_have_cpu = false;
static hsa_amd_memory_pool_t g_kernarg_pool;
static bool g_have_pool = false;

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    const double seconds = argc > 2 ? atof(argv[2]) : 0.5;
    hipFree(nullptr);                                  // the HIP runtime first (it initialises HSA); ours is a second reference
    HSACHK(hsa_init());
    HSACHK(hsa_iterate_agents([](hsa_agent_t a, void*) {
        hsa_device_type_t t;
        hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
        if (t == HSA_DEVICE_TYPE_GPU && !g_have_gpu) { g_gpu = a; g_have_gpu = true; }
        if (t == HSA_DEVICE_TYPE_CPU && !g_have_cpu) { g_cpu = a; g_have_cpu = true; }
        return HSA_STATUS_SUCCESS; }, nullptr));
    if (!g_have_gpu || !g_have_cpu) { fprintf(stderr, "no GPU / CPU agent\n"); return 1; }
    HSACHK(hsa_amd_agent_iterate_memory_pools(g_cpu, [](hsa_amd_memory_pool_t p, void*) {
        hsa_amd_segment_t seg; uint32_t flags = 0;
        hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
        hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
        if (seg == HSA_AMD_SEGMENT_GLOBAL && (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT) && !g_have_pool) { g_kernarg_pool = p; g_have_pool = true; }
        return HSA_STATUS_SUCCESS; }, nullptr));
    if (!g_have_pool) { fprintf(stderr, "no kernarg pool\n"); return 1; }
    // code object
    FILE* f = fopen(argv[1], "rb");
    if (!f) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
    std::vector<char> co;
    { char buf[65536]; size_t n; while ((n = fread(buf, 1, sizeof buf, f)) > 0) co.insert(co.end(), buf, buf + n); fclose(f); }
    hsa_code_object_reader_t reader;
    hsa_executable_t exe;
    HSACHK(hsa_code_object_reader_create_from_memory(co.data(), co.size(), &reader));
    HSACHK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
    HSACHK(hsa_executable_load_agent_code_object(exe, g_gpu, reader, nullptr, nullptr));
    HSACHK(hsa_executable_freeze(exe, nullptr));
    struct Kern { uint64_t object = 0; uint32_t kernarg = 0, group = 0, priv = 0; } k_empty, k_step, k_busy;
    auto get = [&](const char* name, Kern& k) -> int {
        hsa_executable_symbol_t sym;
        HSACHK(hsa_executable_get_symbol_by_name(exe, name, &g_gpu, &sym));
        HSACHK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object));
        HSACHK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.kernarg));
        HSACHK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.group));
        HSACHK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.priv));
        return 0;
    };
    if (get("aql_empty.kd", k_empty) || get("aql_step.kd", k_step) || get("aql_busy.kd", k_busy)) return 1;
    printf("# kernarg segment: aql_empty %u bytes, aql_step %u bytes\n", k_empty.kernarg, k_step.kernarg);
    // variants: AQL_QUEUE_MULTI=1 -> HSA_QUEUE_TYPE_MULTI; AQL_QUEUE_PRIORITY=high|low -> hsa_amd_queue_set_priority
    static const hsa_queue_type32_t qtype = getenv("AQL_QUEUE_MULTI") ? HSA_QUEUE_TYPE_MULTI : HSA_QUEUE_TYPE_SINGLE;
    static const char* qprio = getenv("AQL_QUEUE_PRIORITY");
    auto tune_queue = [](hsa_queue_t* qq) {
        if (qprio) hsa_amd_queue_set_priority(qq, !strcmp(qprio, "high") ? HSA_AMD_QUEUE_PRIORITY_HIGH : !strcmp(qprio, "low") ? HSA_AMD_QUEUE_PRIORITY_LOW : HSA_AMD_QUEUE_PRIORITY_NORMAL);
    };
    static const bool fence_none = getenv("AQL_FENCE_NONE") != nullptr;   // variant: no acquire / release fence on any packet but the chain's last (NOT a correct chain: a measurement of what the fences cost)
    printf("# fences of the packets inside a chain: %s\n", fence_none ? "none" : "agent scope");
    printf("# queues: %s, priority %s\n", qtype == HSA_QUEUE_TYPE_MULTI ? "HSA_QUEUE_TYPE_MULTI" : "HSA_QUEUE_TYPE_SINGLE", qprio ? qprio : "default");
    hsa_queue_t* q = nullptr;
    HSACHK(hsa_queue_create(g_gpu, 1024, qtype, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q));
    tune_queue(q);
    char* kernarg = nullptr;
    HSACHK(hsa_amd_memory_pool_allocate(g_kernarg_pool, 64 * 1024, 0, (void**)&kernarg));
    HSACHK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, kernarg));
    unsigned* word = nullptr;                          // the chain's counter: host-visible, device-writable
    if (hipHostMalloc((void**)&word, 64, hipHostMallocDefault) != hipSuccess) return 1;
    hsa_signal_t done;
    HSACHK(hsa_signal_create(1, 0, nullptr, &done));

    auto submit = [&](const Kern& k, const void* args, size_t bytes, int slot, unsigned grid, unsigned wg, bool last) {
        char* ka = kernarg + 256 * slot;
        memset(ka, 0, k.kernarg < 256 ? k.kernarg : 256);
        memcpy(ka, args, bytes);
        const uint64_t idx = hsa_queue_add_write_index_relaxed(q, 1);
        hsa_kernel_dispatch_packet_t* p = (hsa_kernel_dispatch_packet_t*)q->base_address + (idx & (q->size - 1));
        p->workgroup_size_x = (uint16_t)wg; p->workgroup_size_y = 1; p->workgroup_size_z = 1;
        p->grid_size_x = grid * wg; p->grid_size_y = 1; p->grid_size_z = 1;
        p->private_segment_size = k.priv; p->group_segment_size = k.group;
        p->kernel_object = k.object; p->kernarg_address = ka; p->reserved2 = 0;
        p->completion_signal = last ? done : hsa_signal_t{0};
        const uint16_t header = (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                                           ((fence_none ? HSA_FENCE_SCOPE_NONE : HSA_FENCE_SCOPE_AGENT) << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                                           ((last ? HSA_FENCE_SCOPE_SYSTEM : fence_none ? HSA_FENCE_SCOPE_NONE : HSA_FENCE_SCOPE_AGENT) << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
        const uint16_t setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
        __atomic_store_n((uint32_t*)p, (uint32_t)header | ((uint32_t)setup << 16), __ATOMIC_RELEASE);
        return idx;
    };
    // correctness first: 11 dependent aql_step kernels count the word from 0 to 11
    {
        *word = 0;
        hsa_signal_store_relaxed(done, 1);
        uint64_t idx = 0;
        for (unsigned i = 0; i < 11; i++) { struct { unsigned* p; unsigned e; } a{word, i}; idx = submit(k_step, &a, sizeof a, (int)i, 1, 64, i == 10); }
        hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)idx);
        if (hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, 2000000000ull, HSA_WAIT_STATE_ACTIVE) != 0) { fprintf(stderr, "the chain did not complete\n"); return 1; }
        printf("# 11 dependent packets in order: counter = %u (expected 11)\n", *word);
        if (*word != 11) return 1;
    }
    hipStream_t ps;
    hipStreamCreateWithFlags(&ps, hipStreamNonBlocking);
    for (int nk : {4, 11}) {
        hipGraph_t graph = nullptr; hipGraphExec_t gexec = nullptr;
        hipStreamBeginCapture(ps, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < nk; i++) hipLaunchKernelGGL(hip_empty, dim3(256), dim3(256), 0, ps, (int*)nullptr);
        hipStreamEndCapture(ps, &graph);
        hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0);
        for (int mode = 0; mode < 3; mode++)
        for (int nthr : {0, 4}) {
            std::atomic<bool> stop{false};
            std::vector<std::thread> noise;
            std::vector<hipStream_t> extra;
            for (int i = 0; i < nthr; i++) { hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking); extra.push_back(s2);
                noise.emplace_back([&stop, s2] { while (!stop.load()) hipLaunchKernelGGL(hip_empty, dim3(1), dim3(64), 0, s2, (int*)nullptr); hipStreamSynchronize(s2); }); }
            double in_submit = 0, total = 0; long calls = 0;
            const auto t0 = std::chrono::steady_clock::now();
            while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
                const auto a = std::chrono::steady_clock::now();
                if (mode == 0) for (int i = 0; i < nk; i++) hipLaunchKernelGGL(hip_empty, dim3(256), dim3(256), 0, ps, (int*)nullptr);
                else if (mode == 1) hipGraphLaunch(gexec, ps);
                else {
                    hsa_signal_store_relaxed(done, 1);
                    uint64_t idx = 0;
                    int* null_arg = nullptr;
                    for (int i = 0; i < nk; i++) idx = submit(k_empty, &null_arg, sizeof null_arg, i, 256, 256, i == nk - 1);
                    hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)idx);
                }
                const auto b = std::chrono::steady_clock::now();
                if (mode < 2) { while (hipStreamQuery(ps) == hipErrorNotReady) {} }
                else { while (hsa_signal_load_scacquire(done) >= 1) {} }
                const auto c = std::chrono::steady_clock::now();
                in_submit += std::chrono::duration<double>(b - a).count(); total += std::chrono::duration<double>(c - a).count(); calls++;
            }
            stop = true;
            for (auto& t : noise) t.join();
            for (hipStream_t s2 : extra) hipStreamDestroy(s2);
            printf("%2d dependent empty kernels, %s: %5.1f us to submit, %5.1f us until complete, beside %d threads launching through HIP\n", nk,
                   mode == 0 ? "HIP kernel by kernel     " : mode == 1 ? "one hipGraphLaunch       " : "AQL packets + one doorbell", in_submit / calls * 1e6, total / calls * 1e6, nthr);
            fflush(stdout);
        }
        hipGraphExecDestroy(gexec); hipGraphDestroy(graph);
    }
    // The service's regime: FIVE threads that each run chains of dependent kernels with some work in them (8 us on 64 workgroups per
    // kernel, 4 kernels per chain = an FC call), all five through HIP streams or all five through their own AQL queues.  The time of
    // one thread's chain, alone and beside the other four.
    struct BusyArgs { unsigned ticks; unsigned pad; int* p; };
    for (int aql = 0; aql < 2; aql++)
    for (int nthr : {0, 4}) {
        std::atomic<bool> stop{false};
        std::vector<std::thread> others;
        for (int t = 0; t < nthr; t++) others.emplace_back([&, t] {
            if (!aql) {
                hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
                while (!stop.load()) { for (int i = 0; i < 4; i++) hipLaunchKernelGGL(hip_busy, dim3(64), dim3(256), 0, s2, 800u, (int*)nullptr); while (hipStreamQuery(s2) == hipErrorNotReady) {} }
                hipStreamDestroy(s2);
                return;
            }
            hsa_queue_t* q2 = nullptr; hsa_signal_t d2; char* ka2 = nullptr;
            if (hsa_queue_create(g_gpu, 1024, qtype, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q2) != HSA_STATUS_SUCCESS) return;
            tune_queue(q2);
            hsa_signal_create(1, 0, nullptr, &d2);
            hsa_amd_memory_pool_allocate(g_kernarg_pool, 4096, 0, (void**)&ka2);
            hsa_amd_agents_allow_access(1, &g_gpu, nullptr, ka2);
            while (!stop.load()) {
                hsa_signal_store_relaxed(d2, 1);
                uint64_t idx = 0;
                for (int i = 0; i < 4; i++) {
                    BusyArgs ba{800u, 0u, nullptr};
                    memcpy(ka2 + 256 * i, &ba, sizeof ba);
                    idx = hsa_queue_add_write_index_relaxed(q2, 1);
                    hsa_kernel_dispatch_packet_t* p = (hsa_kernel_dispatch_packet_t*)q2->base_address + (idx & (q2->size - 1));
                    p->workgroup_size_x = 256; p->workgroup_size_y = 1; p->workgroup_size_z = 1;
                    p->grid_size_x = 64 * 256; p->grid_size_y = 1; p->grid_size_z = 1;
                    p->private_segment_size = k_busy.priv; p->group_segment_size = k_busy.group;
                    p->kernel_object = k_busy.object; p->kernarg_address = ka2 + 256 * i; p->reserved2 = 0;
                    p->completion_signal = i == 3 ? d2 : hsa_signal_t{0};
                    const uint16_t header = (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                                                       ((fence_none ? HSA_FENCE_SCOPE_NONE : HSA_FENCE_SCOPE_AGENT) << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                                                       ((i == 3 ? HSA_FENCE_SCOPE_SYSTEM : fence_none ? HSA_FENCE_SCOPE_NONE : HSA_FENCE_SCOPE_AGENT) << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
                    __atomic_store_n((uint32_t*)p, (uint32_t)header | ((uint32_t)(1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS) << 16), __ATOMIC_RELEASE);
                }
                hsa_signal_store_screlease(q2->doorbell_signal, (hsa_signal_value_t)idx);
                while (hsa_signal_load_scacquire(d2) >= 1) {}
            }
            hsa_queue_destroy(q2);
        });
        double in_submit = 0, total = 0; long calls = 0;
        const auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds * 2) {
            const auto a = std::chrono::steady_clock::now();
            if (!aql) for (int i = 0; i < 4; i++) hipLaunchKernelGGL(hip_busy, dim3(64), dim3(256), 0, ps, 800u, (int*)nullptr);
            else {
                hsa_signal_store_relaxed(done, 1);
                uint64_t idx = 0;
                for (int i = 0; i < 4; i++) { BusyArgs ba{800u, 0u, nullptr}; idx = submit(k_busy, &ba, sizeof ba, i, 64, 256, i == 3); }
                hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)idx);
            }
            const auto b = std::chrono::steady_clock::now();
            if (!aql) { while (hipStreamQuery(ps) == hipErrorNotReady) {} } else { while (hsa_signal_load_scacquire(done) >= 1) {} }
            const auto c = std::chrono::steady_clock::now();
            in_submit += std::chrono::duration<double>(b - a).count(); total += std::chrono::duration<double>(c - a).count(); calls++;
        }
        stop = true;
        for (auto& t : others) t.join();
        printf("chain of 4 dependent kernels of 8 us (64 workgroups), %s: %5.1f us to submit, %5.1f us until complete, beside %d threads running the same chains the same way\n",
               aql ? "AQL packets + one doorbell" : "HIP kernel by kernel      ", in_submit / calls * 1e6, total / calls * 1e6, nthr);
        fflush(stdout);
    }
    hsa_queue_destroy(q);
    hsa_shut_down();
    return 0;
}
