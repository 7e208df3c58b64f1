// tools/guard_alloc.cpp -- a guard-page device allocator for torch (torch.cuda.memory.CUDAPluggableAllocator), TOOLS ONLY.
//
// Round 3 ended with an unexplained "Memory access fault by GPU" on the product command.  With torch's caching allocator an
// over-read past the end of a tensor, or a read through a dangling pointer, lands in mapped memory almost always (the
// neighbouring block of a 20-MB..GB segment) and only faults when the tensor happens to be the last block of the last
// mapped segment.  This allocator makes both bugs deterministic:
//   * every allocation is its OWN virtual-memory mapping (hipMemAddressReserve / hipMemCreate / hipMemMap) with an unmapped
//     guard range on both sides; the tensor's END sits on the last mapped byte (GUARD_MODE=tail, default: catches over-reads
//     and over-writes past the end, from 16 bytes on) or its START on the first one (GUARD_MODE=head: catches under-runs);
//   * a freed allocation is unmapped (after a device synchronise, so that work still in flight finishes) and its
//     virtual addresses are NEVER handed out again: any later access through a stale pointer faults;
//   * every allocation / free is appended to GUARD_LOG (text, unbuffered), so tools/guard_report.py can map the address in
//     the fault message to the allocation it belongs to (or runs past).
// Build: hipcc -O2 -shared -fPIC tools/guard_alloc.cpp -o tools/libguard_alloc.so   (tools/guard_run.py does it)
#include <hip/hip_runtime.h>

#include <csignal>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <mutex>
#include <sys/types.h>
#include <unistd.h>
#include <unordered_map>
#include <vector>

namespace {

struct Rec {
    void* va;            // start of the reserved range (guard + mapped + guard)
    size_t reserved;
    void* mapped;        // start of the mapped range
    size_t mapped_bytes;
    hipMemGenericAllocationHandle_t handle;
    size_t size;
    uint64_t serial;
    bool vmm;
};

std::mutex g_mu;
std::unordered_map<void*, Rec> g_live;
std::vector<Rec> g_pending;
size_t g_pending_bytes = 0;
uint64_t g_serial = 0;
int g_log = -2;
size_t g_gran = 0;
size_t g_guard = 0;
int g_head = -1;
int g_vmm_ok = -1;
int g_hold = 0;          // > 0 while a stream capture is open (tools/guard_run.py): no device synchronise in there
long long g_trap = -2;   // GUARD_TRAP_SERIAL: raise SIGUSR1 at that allocation (python's faulthandler prints who asked for it)

void logf(const char* fmt, ...) {
    if (g_log == -2) {
        const char* p = getenv("GUARD_LOG");
        g_log = p ? open(p, O_WRONLY | O_CREAT | O_APPEND, 0644) : -1;
    }
    if (g_log < 0) return;
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    const int n = vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (n > 0) (void)!write(g_log, buf, (size_t)n);
}

size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

void init(int device) {
    if (g_gran) return;
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t g = 0;
    if (hipMemGetAllocationGranularity(&g, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || g == 0) g = 1 << 21;
    g_gran = g;
    g_guard = g < (1u << 21) ? (1u << 21) : g;          // >= 2 MiB of unmapped addresses on both sides
    const char* m = getenv("GUARD_MODE");
    g_head = (m && !strcmp(m, "head")) ? 1 : 0;
    logf("I gran=%zu guard=%zu mode=%s\n", g_gran, g_guard, g_head ? "head" : "tail");
}

void release(const Rec& r) {
    if (r.vmm) {
        hipMemUnmap(r.mapped, r.mapped_bytes);
        hipMemRelease(r.handle);
        // the address range stays reserved for the life of the process: a stale pointer must fault, not alias new memory
    } else {
        hipFree(r.va);
    }
}

void drain_locked() {
    if (g_pending.empty()) return;
    hipDeviceSynchronize();
    for (const Rec& r : g_pending) release(r);
    g_pending.clear();
    g_pending_bytes = 0;
}

}  // namespace

extern "C" void* guard_malloc(ssize_t size_, int device, hipStream_t) {
    std::lock_guard<std::mutex> lk(g_mu);
    hipSetDevice(device);
    init(device);
    const size_t size = size_ > 0 ? (size_t)size_ : 1;
    Rec r;
    memset(&r, 0, sizeof r);
    r.size = size;
    r.serial = ++g_serial;
    r.mapped_bytes = round_up(size, g_gran);
    r.reserved = r.mapped_bytes + 2 * g_guard;
    const size_t tail_off = r.mapped_bytes - round_up(size, 16);      // end of the tensor on the last mapped 16 bytes
    void* user = nullptr;
    if (g_vmm_ok != 0) {
        hipMemAllocationProp prop;
        memset(&prop, 0, sizeof prop);
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = device;
        hipError_t e = hipMemAddressReserve(&r.va, r.reserved, g_gran, nullptr, 0);
        if (e == hipSuccess) e = hipMemCreate(&r.handle, r.mapped_bytes, &prop, 0);
        if (e == hipSuccess) {
            r.mapped = (char*)r.va + g_guard;
            e = hipMemMap(r.mapped, r.mapped_bytes, 0, r.handle, 0);
        }
        if (e == hipSuccess) {
            hipMemAccessDesc d;
            memset(&d, 0, sizeof d);
            d.location.type = hipMemLocationTypeDevice;
            d.location.id = device;
            d.flags = hipMemAccessFlagsProtReadWrite;
            e = hipMemSetAccess(r.mapped, r.mapped_bytes, &d, 1);
        }
        if (e == hipSuccess) {
            r.vmm = true;
            g_vmm_ok = 1;
            user = (char*)r.mapped + (g_head ? 0 : tail_off);
        } else {
            if (g_vmm_ok == 1) {          // worked before: this is memory pressure -> drain the deferred frees and report
                logf("E vmm alloc failed err=%d size=%zu\n", (int)e, size);
                (void)hipGetLastError();
                drain_locked();
                return nullptr;
            }
            logf("I virtual-memory API unavailable (err=%d): falling back to one hipMalloc per tensor, tail-aligned\n", (int)e);
            (void)hipGetLastError();
            g_vmm_ok = 0;
        }
    }
    if (!user) {
        if (hipMalloc(&r.va, r.mapped_bytes) != hipSuccess) return nullptr;
        r.mapped = r.va;
        r.vmm = false;
        user = (char*)r.va + (g_head ? 0 : tail_off);
    }
    g_live[user] = r;
    if (g_trap == -2) {
        const char* t = getenv("GUARD_TRAP_SERIAL");
        g_trap = t ? atoll(t) : -1;
    }
    if ((long long)r.serial == g_trap) raise(SIGUSR1);
    logf("A %llu %p %zu %p %zu\n", (unsigned long long)r.serial, user, size, r.mapped, r.mapped_bytes);
    return user;
}

extern "C" void guard_free(void* ptr, ssize_t, int, hipStream_t) {
    if (!ptr) return;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_live.find(ptr);
    if (it == g_live.end()) {
        logf("E free of unknown pointer %p\n", ptr);
        return;
    }
    logf("F %llu %p\n", (unsigned long long)it->second.serial, ptr);
    g_pending.push_back(it->second);
    g_pending_bytes += it->second.mapped_bytes;
    g_live.erase(it);
    // frees are deferred (work in flight may still use the block, exactly as with a stream-ordered allocator) and released in
    // batches behind ONE device synchronise
    if (!g_hold && (g_pending.size() >= 192 || g_pending_bytes > (size_t(12) << 30))) drain_locked();
}

extern "C" void guard_hold(int delta) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_hold += delta;
}

extern "C" void guard_drain() {
    std::lock_guard<std::mutex> lk(g_mu);
    drain_locked();
}

extern "C" int guard_uses_vmm() { return g_vmm_ok; }
