// hens_aql.h - direct AQL dispatch of the stepping kernels on a user-mode HSA queue of the context's own (round 4).
//
// Why: a hens_step call of 20 iterations is 40 dependent launches of ~8 us each between two host synchronisations.  Through the
// HIP stream that call costs ~13 us on top of its kernels on an MI355X (tools/probe/call_floor.hip: 3 us of host work per launch,
// every launch a doorbell of its own, a marker packet + signal round trip for the synchronisation) and the kernels themselves run
// ~0.14 us per launch slower while the host is still feeding the queue.  Here the library writes the AQL packets itself:
//   * a packet is 64 bytes of stores + (per call, not per launch) one doorbell - ~0.2 us of host time per launch;
//   * the kernels are the SAME machine code HIP would launch: the gfx950 code object is taken out of this library's own fat binary
//     (clang offload bundle in the .hip_fatbin section of libhipensemble.so) and loaded through the HSA loader;
//   * between the launches of a call the packets carry the barrier bit and agent-scope acquire / release fences - what a HIP
//     stream gives consecutive kernels; the first packet after the host touched the state acquires at system scope, the last one
//     releases at system scope and decrements the context's completion signal, which hens_synchronize spins on.
// Everything that is not a stepping launch (uploads, downloads, the parity API, profiling with HIP events) stays on the HIP stream;
// the two queues are ordered against each other on the host (AqlQueue::drain before HIP work, hipStreamSynchronize before the first
// packet of a call if the stream is busy).
#pragma once
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <dlfcn.h>
#include <algorithm>
#include <elf.h>
#include <cstdlib>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

namespace hens_aql {

struct Kernel {
    uint64_t object = 0;
    uint32_t kernarg_size = 0, group_size = 0, private_size = 0;
    bool resolved = false;
};

// One per HIP device ordinal, shared by the contexts of the process: agent, kernarg pool, loaded code object.
struct Device {
    bool tried = false, ok = false;
    std::string err;
    hsa_agent_t agent{};
    hsa_agent_t cpu_agent{};
    hsa_amd_memory_pool_t kernarg_pool{};       // host memory (fine-grained, KERNARG_INIT)
    hsa_amd_memory_pool_t vram_pool{};          // the GPU's coarse-grained memory: where the kernarg ring lives when the host can map it
    hsa_amd_memory_pool_t vram_fine_pool{};     // ... its fine-grained pool, if it has one
    bool have_pool = false, have_vram = false, have_vram_fine = false;
    hsa_amd_hdp_flush_t hdp{nullptr, nullptr};
    // one loaded executable per code object: the library is built from several translation units (hens_ktable.h), each with a
    // fat binary of its own in .hip_fatbin
    std::vector<hsa_executable_t> exes;
    std::vector<hsa_code_object_reader_t> readers;
    std::vector<std::vector<char>> images;      // the code objects' bytes (the readers refer to them)
    std::unordered_map<std::string, Kernel> kernels;
    std::mutex mu;
};

inline bool fail(Device& d, const char* what, hsa_status_t st = HSA_STATUS_SUCCESS) {
    const char* s = nullptr;
    if (st != HSA_STATUS_SUCCESS) hsa_status_string(st, &s);
    d.err = std::string(what) + (s ? std::string(": ") + s : std::string());
    d.ok = false;
    return false;
}

// the gfx950 code object out of libhipensemble.so's own fat binary
inline bool read_code_object(Device& d) {
    Dl_info info{};
    if (!dladdr(reinterpret_cast<const void*>(&read_code_object), &info) || !info.dli_fname) return fail(d, "dladdr found no library path");
    std::ifstream f(info.dli_fname, std::ios::binary);
    if (!f) return fail(d, "cannot open the library file");
    std::vector<char> so((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (so.size() < sizeof(Elf64_Ehdr) || memcmp(so.data(), ELFMAG, SELFMAG) != 0) return fail(d, "library file is not an ELF image");
    const Elf64_Ehdr* eh = reinterpret_cast<const Elf64_Ehdr*>(so.data());
    if (eh->e_shoff == 0 || eh->e_shoff + (uint64_t)eh->e_shnum * sizeof(Elf64_Shdr) > so.size()) return fail(d, "no section headers");
    const Elf64_Shdr* sh = reinterpret_cast<const Elf64_Shdr*>(so.data() + eh->e_shoff);
    const char* names = so.data() + sh[eh->e_shstrndx].sh_offset;
    const char* fat = nullptr;
    size_t fat_size = 0;
    for (int i = 0; i < eh->e_shnum; ++i)
        if (strcmp(names + sh[i].sh_name, ".hip_fatbin") == 0) { fat = so.data() + sh[i].sh_offset; fat_size = sh[i].sh_size; }
    static const char MAGIC[] = "__CLANG_OFFLOAD_BUNDLE__";
    if (!fat || fat_size < 32) return fail(d, "no .hip_fatbin section");
    // every translation unit's bundle, one after the other (each aligned; found by its magic)
    for (const char* b = fat; b + 32 <= fat + fat_size;) {
        b = static_cast<const char*>(memmem(b, (size_t)(fat + fat_size - b), MAGIC, 24));
        if (!b) break;
        const size_t left = (size_t)(fat + fat_size - b);
        uint64_t n;
        memcpy(&n, b + 24, 8);
        size_t off = 32, end = 32;
        for (uint64_t i = 0; i < n && off + 24 <= left; ++i) {
            uint64_t eo, es, ts;
            memcpy(&eo, b + off, 8); memcpy(&es, b + off + 8, 8); memcpy(&ts, b + off + 16, 8);
            off += 24;
            if (off + ts > left) break;
            const std::string triple(b + off, ts);
            off += ts;
            if (eo + es <= left) end = std::max<size_t>(end, eo + es);
            if (triple.find("amdgcn") != std::string::npos && triple.find("gfx950") != std::string::npos && es > 0 && eo + es <= left)
                d.images.emplace_back(b + eo, b + eo + es);
        }
        b += std::max<size_t>(end, off);
    }
    if (d.images.empty()) return fail(d, "no uncompressed gfx950 code object in the library's fat binaries");
    return true;
}

struct FindAgent { uint32_t bdf, domain; hsa_agent_t out; bool found; };
inline hsa_status_t find_gpu_cb(hsa_agent_t a, void* p) {
    FindAgent* fa = static_cast<FindAgent*>(p);
    hsa_device_type_t t;
    if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS || t != HSA_DEVICE_TYPE_GPU) return HSA_STATUS_SUCCESS;
    uint32_t bdf = 0, dom = 0;
    (void)hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf);
    (void)hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &dom);
    if (bdf == fa->bdf && dom == fa->domain) { fa->out = a; fa->found = true; return HSA_STATUS_INFO_BREAK; }
    return HSA_STATUS_SUCCESS;
}
struct FindPool { hsa_amd_memory_pool_t out; bool found; hsa_agent_t agent; uint32_t want; };
inline hsa_status_t find_vram_pool_cb(hsa_amd_memory_pool_t pool, void* p) {
    hsa_amd_segment_t seg;
    if (hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg) != HSA_STATUS_SUCCESS || seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
    uint32_t flags = 0;
    (void)hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    bool alloc = false;
    (void)hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
    FindPool* fp = static_cast<FindPool*>(p);
    if (getenv("HENS_AQL_STATS")) fprintf(stderr, "[hipensemble] GPU memory pool: global flags 0x%x%s\n", flags, alloc ? "" : " (no runtime allocation)");
    if ((flags & fp->want) && alloc && !fp->found) { fp->out = pool; fp->found = true; }
    return HSA_STATUS_SUCCESS;
}
inline hsa_status_t find_kernarg_pool_cb(hsa_amd_memory_pool_t pool, void* p) {
    hsa_amd_segment_t seg;
    if (hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg) != HSA_STATUS_SUCCESS || seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
    uint32_t flags = 0;
    (void)hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    bool alloc = false;
    (void)hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
    if ((flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT) && alloc) {
        FindPool* fp = static_cast<FindPool*>(p);
        fp->out = pool; fp->found = true;
        return HSA_STATUS_INFO_BREAK;
    }
    return HSA_STATUS_SUCCESS;
}
inline hsa_status_t find_cpu_pool_cb(hsa_agent_t a, void* p) {
    hsa_device_type_t t;
    if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS || t != HSA_DEVICE_TYPE_CPU) return HSA_STATUS_SUCCESS;
    const hsa_status_t st = hsa_amd_agent_iterate_memory_pools(a, find_kernarg_pool_cb, p);
    if (st == HSA_STATUS_INFO_BREAK) static_cast<FindPool*>(p)->agent = a;
    return st == HSA_STATUS_INFO_BREAK ? HSA_STATUS_INFO_BREAK : HSA_STATUS_SUCCESS;
}

inline Device& device(int hip_device) {
    static Device devs[64];
    Device& d = devs[hip_device & 63];
    std::lock_guard<std::mutex> lk(d.mu);
    if (d.tried) return d;
    d.tried = true;
    hsa_status_t st = hsa_init();                                   // (reference counted: HIP holds the runtime already)
    if (st != HSA_STATUS_SUCCESS) { fail(d, "hsa_init", st); return d; }
    int bus = 0, dev = 0, dom = 0;
    if (hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, hip_device) != hipSuccess ||
        hipDeviceGetAttribute(&dev, hipDeviceAttributePciDeviceId, hip_device) != hipSuccess ||
        hipDeviceGetAttribute(&dom, hipDeviceAttributePciDomainID, hip_device) != hipSuccess) { fail(d, "no PCI address for the HIP device"); return d; }
    FindAgent fa{(uint32_t)((bus << 8) | (dev << 3)), (uint32_t)dom, {}, false};
    (void)hsa_iterate_agents(find_gpu_cb, &fa);
    if (!fa.found) { fail(d, "no HSA agent at the HIP device's PCI address"); return d; }
    d.agent = fa.out;
    FindPool fp{{}, false, {}, 0};
    (void)hsa_iterate_agents(find_cpu_pool_cb, &fp);
    if (!fp.found) { fail(d, "no kernarg memory pool"); return d; }
    d.kernarg_pool = fp.out; d.have_pool = true; d.cpu_agent = fp.agent;
    FindPool fv{{}, false, {}, HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED};
    (void)hsa_amd_agent_iterate_memory_pools(d.agent, find_vram_pool_cb, &fv);
    if (fv.found) { d.vram_pool = fv.out; d.have_vram = true; }
    FindPool ff{{}, false, {}, HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_FINE_GRAINED | HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_EXTENDED_SCOPE_FINE_GRAINED};
    (void)hsa_amd_agent_iterate_memory_pools(d.agent, find_vram_pool_cb, &ff);
    if (ff.found) { d.vram_fine_pool = ff.out; d.have_vram_fine = true; }
    (void)hsa_agent_get_info(d.agent, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_HDP_FLUSH, &d.hdp);
    if (!read_code_object(d)) return d;
    for (const std::vector<char>& image : d.images) {
        hsa_code_object_reader_t reader{};
        hsa_executable_t exe{};
        st = hsa_code_object_reader_create_from_memory(image.data(), image.size(), &reader);
        if (st != HSA_STATUS_SUCCESS) { fail(d, "hsa_code_object_reader_create_from_memory", st); return d; }
        d.readers.push_back(reader);
        st = hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe);
        if (st != HSA_STATUS_SUCCESS) { fail(d, "hsa_executable_create_alt", st); return d; }
        d.exes.push_back(exe);
        st = hsa_executable_load_agent_code_object(exe, d.agent, reader, nullptr, nullptr);
        if (st != HSA_STATUS_SUCCESS) { fail(d, "hsa_executable_load_agent_code_object", st); return d; }
        st = hsa_executable_freeze(exe, nullptr);
        if (st != HSA_STATUS_SUCCESS) { fail(d, "hsa_executable_freeze", st); return d; }
    }
    d.ok = true;
    return d;
}

// the kernel HIP launches for host function `host_fn` - by its registered device name - in this library's loaded code object
inline const Kernel* kernel_for(Device& d, const void* host_fn) {
    const char* nm = hipKernelNameRefByPtr(host_fn, nullptr);
    if (!nm) return nullptr;
    std::lock_guard<std::mutex> lk(d.mu);
    Kernel& k = d.kernels[nm];
    if (k.resolved) return k.object ? &k : nullptr;
    k.resolved = true;
    const std::string sym = std::string(nm) + ".kd";
    hsa_executable_symbol_t s{};
    bool found = false;
    for (hsa_executable_t exe : d.exes)
        if (hsa_executable_get_symbol_by_name(exe, sym.c_str(), &d.agent, &s) == HSA_STATUS_SUCCESS && s.handle) { found = true; break; }
    if (!found) return nullptr;
    uint64_t obj = 0;
    uint32_t ka = 0, gs = 0, ps = 0;
    if (hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &obj) != HSA_STATUS_SUCCESS) return nullptr;
    (void)hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &ka);
    (void)hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &gs);
    (void)hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &ps);
    k.object = obj; k.kernarg_size = ka; k.group_size = gs; k.private_size = ps;
    return &k;
}

// code object v5 implicit arguments (256 bytes behind the explicit ones, 8-byte aligned): the few our kernels read
struct ImplicitArgs {
    uint32_t block_count[3];
    uint16_t group_size[3];
    uint16_t remainder[3];
    uint8_t reserved0[16];
    uint64_t global_offset[3];
    uint16_t grid_dims;
    uint8_t reserved1[54];
    uint32_t dynamic_lds_size;
    uint8_t reserved2[132];
};
static_assert(sizeof(ImplicitArgs) == 256, "code object v5 implicit argument block");
static_assert(offsetof(ImplicitArgs, global_offset) == 40 && offsetof(ImplicitArgs, grid_dims) == 64 && offsetof(ImplicitArgs, dynamic_lds_size) == 120,
              "code object v5 implicit argument offsets");

constexpr uint32_t QUEUE_PACKETS = 4096;        // power of two
constexpr uint32_t SLOT_BYTES = 1024;           // one kernarg slot per queue slot (StretchArgs + implicit block = 944 bytes)

struct Queue {
    Device* dev = nullptr;
    hsa_queue_t* q = nullptr;
    hsa_signal_t done{};
    uint32_t qsize = 0;              // packets the queue REALLY holds (a profiler's intercept queue need not honour the request)
    char* kernarg = nullptr;
    bool kernarg_dev = false;        // the ring is device memory mapped into the host (writes cross the BAR: flush before the doorbell)
    bool kernarg_fine = false;
    volatile char* last_ka = nullptr;
    // how argument writes are made visible before a doorbell: 0 sfence only, 1 + HDP flush register write, 2 + read back except at a
    // call's first doorbell (measured: 5.0 vs 6.7 us per call), 3 + read back before EVERY doorbell - the ordered form, what HIP's
    // own runtime does for device-resident kernel arguments, and the default since round 5 (HENS_AQL_FLUSH=2 opts into the shortcut:
    // it relies on the packet fetch over PCIe taking longer than the flush, which holds on this machine and is no guarantee)
    int flush_mode = 3;
    bool acq_agent_ok = true;        // the first packet of a call acquires at agent scope when only this queue touched the state
    uint64_t windex = 0;             // next packet index (single producer: the context's host thread)
    uint64_t rung = 0;               // packets below this index have been handed to the doorbell
    uint64_t call_first = 0;         // index of the current hens_step call's first packet
    uint64_t packets = 0, doorbells = 0;   // statistics
    bool unsignalled = false;        // packets have been queued since the last one that carries the completion signal
    int64_t done_target = 0;         // value of `done` once every signalled packet so far has completed
    bool pending = false;            // packets submitted since the last drain
    bool own_only = false;           // since the last drain only this queue has touched the state (no HIP work, no host upload)
    bool fresh = true;               // the next packet is the first since the host (or the HIP stream) touched the state
    bool norel_next = false;         // the next packet's kernel writes everything a later launch reads through to memory and waits for its
                                     // stores (hens_kernels.h: wt_store, launch_end_wait): no release fence (not on a call's last packet)
    bool nobar_next = false;         // DEV PROBE (timing only, results wrong): the next packet goes out without the barrier bit
    // Dispatch timestamps (hens_set_profiling(ctx, 2), round 6): every packet of a profiled call carries a completion signal of its
    // own and the packet processor stamps the launch's begin and end into it (hsa_amd_profiling_*: what rocprofv3's kernel trace
    // reads) - per-launch durations of the SAME packets, fences and queue the timed calls use, not of a HIP-stream stand-in.
    bool prof = false;
    bool prof_enabled = false;       // the queue stamps packets that carry a completion signal: switched on when the queue is created (as
                                     // HIP's runtime does for its own queues) - switched on later, between two calls, the packet processor
                                     // went on with the queue descriptor it had and the first profiled call read zeros (round 6, session 1)
    int prof_kind = -1;              // tag of the next packet (hens_step: 0 first launch, 2 second launch, 3 one-launch iteration; -1 other)
    std::vector<hsa_signal_t> prof_pool;
    size_t prof_used = 0;
    std::vector<int> prof_kinds;
    std::string err;

    bool create(Device& d) {
        dev = &d;
        hsa_status_t st = hsa_queue_create(d.agent, QUEUE_PACKETS, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q);
        if (st != HSA_STATUS_SUCCESS) { err = "hsa_queue_create failed"; q = nullptr; return false; }
        qsize = q->size;
        if (qsize < 8 || (qsize & (qsize - 1)) != 0) { err = "AQL queue size is not a power of two >= 8"; return false; }
        st = hsa_signal_create(0, 0, nullptr, &done);
        if (st != HSA_STATUS_SUCCESS) { err = "hsa_signal_create failed"; return false; }
        done_target = 0;
        // The kernarg ring lives in DEVICE memory, written by the host through the PCIe BAR (what HIP does on this platform, its
        // HIP_FORCE_DEV_KERNARG): every wave of a launch reads its arguments with scalar loads, and from fine-grained host memory
        // those loads cross PCIe - measured here: 41 us per launch of 4 096 waves instead of 8.  HENS_AQL_HOST_KERNARG=1 = host pool.
        // (measured, round 4: the fine-grained device pool and the coarse-grained one run the kernels alike; HSA's UNCACHED flag
        //  costs 3 us per launch - every wave's argument loads go to memory)
        if (d.have_vram) {
            const bool fine = d.have_vram_fine;
            st = hsa_amd_memory_pool_allocate(fine ? d.vram_fine_pool : d.vram_pool, (size_t)qsize * SLOT_BYTES, 0, reinterpret_cast<void**>(&kernarg));
            kernarg_fine = fine;
            if (st == HSA_STATUS_SUCCESS) {
                st = hsa_amd_agents_allow_access(1, &d.cpu_agent, nullptr, kernarg);
                if (st == HSA_STATUS_SUCCESS) kernarg_dev = true;
                else { (void)hsa_amd_memory_pool_free(kernarg); kernarg = nullptr; }
            } else kernarg = nullptr;
        }
        if (!kernarg) {
            // (no host mapping of device memory - no large BAR: a ring in HOST memory would run every launch at 41 us instead of 8;
            //  the context then keeps its launches on the HIP stream, whose runtime places kernel arguments itself)
            err = "the kernarg ring cannot live in host-mapped device memory on this system";
            return false;
        }
        // first touch of every page of the ring NOW: the host's mapping of device memory is populated by page faults (a fresh 4 KiB
        // page every four packets until the ring has wrapped once - 2 048 iterations; measured: blocks of 20 iterations 10 us
        // slower for as long as every block met fresh pages)
        memset(kernarg, 0, (size_t)qsize * SLOT_BYTES);
        __builtin_ia32_sfence();
        windex = hsa_queue_load_write_index_relaxed(q);
        rung = windex;
        if (!getenv("HENS_AQL_PROF_LATE") && hsa_amd_profiling_set_profiler_enabled(q, 1) == HSA_STATUS_SUCCESS) prof_enabled = true;
        return true;
    }
    bool set_prof(bool on) {
        if (!q) return false;
        if (on == prof) { if (on) { prof_used = 0; prof_kinds.clear(); } return true; }
        if (!drain(30.0)) return false;
        if (on && !prof_enabled) {
            if (hsa_amd_profiling_set_profiler_enabled(q, 1) != HSA_STATUS_SUCCESS) { err = "hsa_amd_profiling_set_profiler_enabled failed"; return false; }
            prof_enabled = true;
        }
        prof = on;
        prof_used = 0;
        prof_kinds.clear();
        return true;
    }
    // begin / end of every profiled packet since set_prof(true), in us after the first one's begin (call after drain)
    bool collect_prof(std::vector<double>& begin_end_us, std::vector<int>& kinds) {
        begin_end_us.clear();
        kinds.clear();
        uint64_t freq = 0;
        if (hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP_FREQUENCY, &freq) != HSA_STATUS_SUCCESS || freq == 0) { err = "no HSA timestamp frequency"; return false; }
        uint64_t t0 = 0;
        for (size_t i = 0; i < prof_used; ++i) {
            hsa_amd_profiling_dispatch_time_t t{};
            if (hsa_amd_profiling_get_dispatch_time(dev->agent, prof_pool[i], &t) != HSA_STATUS_SUCCESS || t.end <= t.start) { err = "hsa_amd_profiling_get_dispatch_time: no timestamps in a profiled packet's signal"; return false; }
            if (i == 0) t0 = t.start;
            begin_end_us.push_back((double)(int64_t)(t.start - t0) * 1e6 / (double)freq);
            begin_end_us.push_back((double)(int64_t)(t.end - t0) * 1e6 / (double)freq);
            kinds.push_back(prof_kinds[i]);
        }
        return true;
    }
    void destroy() {
        if (q) { (void)drain(5.0); (void)hsa_queue_destroy(q); q = nullptr; }
        for (hsa_signal_t s : prof_pool) (void)hsa_signal_destroy(s);
        prof_pool.clear();
        if (done.handle) { (void)hsa_signal_destroy(done); done.handle = 0; }
        if (kernarg) { (void)hsa_amd_memory_pool_free(kernarg); kernarg = nullptr; }
    }

    // One kernel dispatch.  grid in workgroups.  `signal`: this packet ends a call - system-scope release + completion signal.
    // The doorbell is rung by ring() (once per call, or every few packets of a long one).
    bool dispatch(const Kernel& k, uint32_t gx, uint32_t gy, uint32_t gz, uint32_t bx, uint32_t dyn_lds, const void* args, size_t args_size, bool signal) {
        const size_t impl_off = (args_size + 7) & ~(size_t)7;
        const bool implicit = k.kernarg_size > impl_off;
        if (impl_off + (implicit ? sizeof(ImplicitArgs) : 0) > SLOT_BYTES || k.kernarg_size > SLOT_BYTES) { err = "kernel arguments exceed the kernarg slot"; return false; }
        // a slot is free once the packet AFTER its old tenant has been consumed (barrier bit: the old tenant has completed then)
        if (windex + 2 > hsa_queue_load_read_index_relaxed(q) + qsize) {
            ring();
            const auto t0 = std::chrono::steady_clock::now();
            while (windex + 2 > hsa_queue_load_read_index_scacquire(q) + qsize)
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 30.0) { err = "AQL queue stalled for 30 s"; return false; }
        }
        const uint32_t slot = (uint32_t)(windex & (qsize - 1));
        char* ka = kernarg + (size_t)slot * SLOT_BYTES;
        memcpy(ka, args, args_size);
        if (implicit) {
            ImplicitArgs ia{};
            ia.block_count[0] = gx; ia.block_count[1] = gy; ia.block_count[2] = gz;
            ia.group_size[0] = (uint16_t)bx; ia.group_size[1] = 1; ia.group_size[2] = 1;
            ia.grid_dims = gz > 1 ? 3 : (gy > 1 ? 2 : 1);
            ia.dynamic_lds_size = dyn_lds;
            memcpy(ka + impl_off, &ia, std::min(sizeof(ImplicitArgs), (size_t)k.kernarg_size - impl_off));
        }
        hsa_kernel_dispatch_packet_t* p = reinterpret_cast<hsa_kernel_dispatch_packet_t*>(q->base_address) + slot;
        p->workgroup_size_x = (uint16_t)bx; p->workgroup_size_y = 1; p->workgroup_size_z = 1;
        p->reserved0 = 0;
        p->grid_size_x = gx * bx; p->grid_size_y = gy; p->grid_size_z = gz;
        p->private_segment_size = k.private_size;
        p->group_segment_size = k.group_size + dyn_lds;
        p->kernel_object = k.object;
        p->kernarg_address = ka;
        last_ka = ka + (implicit ? impl_off + std::min(sizeof(ImplicitArgs), (size_t)k.kernarg_size - impl_off) : args_size) - 1;
        p->reserved2 = 0;
        if (prof) {                  // (the call's completion signal then rides on a barrier packet behind its last launch: drain())
            if (prof_used == prof_pool.size()) {
                hsa_signal_t s{};
                if (hsa_signal_create(1, 0, nullptr, &s) != HSA_STATUS_SUCCESS) { err = "hsa_signal_create failed (profiling)"; return false; }
                prof_pool.push_back(s);
            }
            hsa_signal_store_relaxed(prof_pool[prof_used], 1);
            p->completion_signal = prof_pool[prof_used++];
            prof_kinds.push_back(prof_kind);
        } else
            p->completion_signal.handle = signal ? done.handle : 0;
        prof_kind = -1;              // (a tag holds for ONE packet, profiled or not)
        uint16_t acq = (fresh && !(acq_agent_ok && own_only)) ? HSA_FENCE_SCOPE_SYSTEM : HSA_FENCE_SCOPE_AGENT;
#ifdef HENS_DEV_BUILD
        // timing probe (NOT correct: another XCD's L2 may hold a stale copy of a row): no acquire fence between the launches of a call
        static const bool acq_none = getenv("HENS_AQL_ACQ_NONE") != nullptr;
        if (acq_none && !fresh) acq = HSA_FENCE_SCOPE_NONE;
#endif
        uint16_t rel = signal ? HSA_FENCE_SCOPE_SYSTEM : HSA_FENCE_SCOPE_AGENT;
#ifdef HENS_DEV_BUILD
        static const bool rel_none = getenv("HENS_AQL_REL_NONE") != nullptr;        // (timing probe, as HENS_AQL_ACQ_NONE below)
        if (rel_none && !signal) rel = HSA_FENCE_SCOPE_NONE;
#endif
        if (norel_next && !signal) rel = HSA_FENCE_SCOPE_NONE;
        norel_next = false;
        const uint16_t barrier = nobar_next ? 0u : 1u;
        nobar_next = false;
        const uint16_t header = (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (barrier << HSA_PACKET_HEADER_BARRIER) |
                                           (acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
        const uint16_t setup = (uint16_t)(3u << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS);
        if (signal && !prof) hsa_signal_add_relaxed(done, 1);         // (the packet's completion takes the 1 back off)
        unsignalled = !signal || prof;
        __atomic_store_n(reinterpret_cast<uint32_t*>(p), (uint32_t)header | ((uint32_t)setup << 16), __ATOMIC_RELEASE);
        windex += 1;
        packets += 1;
        hsa_queue_store_write_index_relaxed(q, windex);
        fresh = false;
        pending = true;
        // (never let one doorbell cover packets on both sides of the ring's end: rocprofv3's intercept queue copies the range a
        //  doorbell announces in one piece and ran off the end of the ring - SIGSEGV at the first wrap-around under --kernel-trace)
        if ((windex & (qsize - 1)) == 0) ring();
        return true;
    }
    void ring() {
        if (rung == windex) return;
        if (kernarg_dev && last_ka) {
            // Argument writes cross the PCIe BAR and the GPU's host data path (HDP): they must have landed in device memory before a
            // wave reads them.  sfence drains the CPU's write-combining buffers; the HDP flush register write travels behind them
            // and ahead of the doorbell (posted writes to one device stay in order).  With the queue backlogged the read-back that
            // waits for the flush costs host time nobody misses; in front of a call's FIRST doorbell - the GPU is idle, and the
            // packet processor still has to fetch the packet over PCIe (>= 1 us) before any wave starts - it is skipped: 1.8 us per call.
            const int mode = flush_mode == 2 ? (rung == call_first ? 1 : 2) : (flush_mode >= 3 ? 2 : flush_mode);
            __builtin_ia32_sfence();
            if (mode >= 1 && dev->hdp.HDP_MEM_FLUSH_CNTL) {
                *reinterpret_cast<volatile uint32_t*>(dev->hdp.HDP_MEM_FLUSH_CNTL) = 1u;
                if (mode >= 2) (void)*reinterpret_cast<volatile uint32_t*>(dev->hdp.HDP_MEM_FLUSH_CNTL);
            } else if (mode >= 1) (void)*last_ka;           // (no flush register: a read through the same path comes back behind the writes)
        }
        hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)(windex - 1));
        doorbells += 1;
        rung = windex;
    }
    // a barrier packet that carries the completion signal (a call that ended without a signalled dispatch: error paths)
    void barrier_signal() {
        const uint32_t slot = (uint32_t)(windex & (qsize - 1));
        while (windex + 2 > hsa_queue_load_read_index_scacquire(q) + qsize) {}
        hsa_barrier_and_packet_t* p = reinterpret_cast<hsa_barrier_and_packet_t*>(q->base_address) + slot;
        memset(reinterpret_cast<char*>(p) + 4, 0, sizeof(*p) - 4);
        p->completion_signal = done;
        hsa_signal_add_relaxed(done, 1);
        const uint16_t header = (uint16_t)((HSA_PACKET_TYPE_BARRIER_AND << HSA_PACKET_HEADER_TYPE) | (1u << HSA_PACKET_HEADER_BARRIER) |
                                           (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                                           (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
        __atomic_store_n(reinterpret_cast<uint32_t*>(p), (uint32_t)header, __ATOMIC_RELEASE);
        windex += 1;
        hsa_queue_store_write_index_relaxed(q, windex);
        unsignalled = false;
    }
    // host waits until every signalled packet has completed (the last packet of every call is one)
    bool drain(double timeout_s) {
        if (!pending) return true;
        if (unsignalled) barrier_signal();
        ring();
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            if (hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_EQ, done_target, 200000, HSA_WAIT_STATE_ACTIVE) == done_target) break;
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) { err = "AQL completion signal timed out"; return false; }
        }
        pending = false;
        fresh = true;
        return true;
    }
};

}  // namespace hens_aql
