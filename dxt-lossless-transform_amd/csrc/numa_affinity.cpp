// numa_affinity.cpp -- where the host threads that feed a device run.
//
// dxtlt_transform_sharded starts one worker thread per shard (plus a downloader thread per chunked pipeline).  On the
// two-socket hosts these GPUs sit in, a thread that submits copies for device d from the other socket pays the
// inter-socket hop on every doorbell, every page it pins for the DMA engines and every completion it polls; eight shards
// doing that blindly is what VERDICT r02 flagged.  Each worker therefore binds itself, before it touches the device, to
// the CPUs the kernel reports as local to the device's PCI function (/sys/bus/pci/devices/<bdf>/local_cpulist; falling
// back to numa_node -> /sys/devices/system/node/node<N>/cpulist), intersected with the CPUs the process may use.
//
// Only threads this library creates are bound -- never the caller's.  DXTLT_NUMA_BIND=0 switches it off; DXTLT_SYSFS_ROOT
// replaces "/sys" (tests point it at a mock tree).  Pure host code except dxtlt_device_local_cpulist, which asks HIP for
// the device's bus id.
#include <hip/hip_runtime_api.h>
#include <sched.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <exception>
#include <string>
#include <vector>

#include "../../include/dxtlt_gfx950.h"
#include "host_common.h"

namespace {

std::string sysfs_root()
{
    const char* r = std::getenv("DXTLT_SYSFS_ROOT");
    return (r && *r) ? std::string(r) : std::string("/sys");
}

bool read_line(const std::string& path, std::string* out)
{
    FILE* f = std::fopen(path.c_str(), "r");
    if (!f)
        return false;
    char buf[4096];
    const bool ok = std::fgets(buf, sizeof buf, f) != nullptr;
    std::fclose(f);
    if (!ok)
        return false;
    std::string s(buf);
    while (!s.empty() && (s.back() == '\n' || s.back() == '\r' || s.back() == ' '))
        s.pop_back();
    *out = s;
    return true;
}

// "0-3,8,10-11" -> CPU numbers; false on anything else
bool parse_cpulist(const char* text, std::vector<int>* cpus)
{
    cpus->clear();
    const char* p = text;
    while (*p) {
        char* end = nullptr;
        const long a = std::strtol(p, &end, 10);
        if (end == p || a < 0 || a >= CPU_SETSIZE)
            return false;
        long b = a;
        p = end;
        if (*p == '-') {
            ++p;
            b = std::strtol(p, &end, 10);
            if (end == p || b < a || b >= CPU_SETSIZE)
                return false;
            p = end;
        }
        for (long c = a; c <= b; ++c)
            cpus->push_back((int)c);
        if (*p == ',')
            ++p;
        else if (*p)
            return false;
    }
    return !cpus->empty();
}

std::string lower(std::string s)
{
    for (auto& c : s)
        if (c >= 'A' && c <= 'Z')
            c = (char)(c - 'A' + 'a');
    return s;
}

std::mutex g_cache_mutex;
std::vector<std::string> g_device_cpulist;   // per device ordinal; "" = unknown, "?" = not looked up yet

}  // namespace

extern "C" {

int32_t dxtlt_pci_local_cpulist(const char* pci_bdf, char* out, size_t cap)
{
    if (pci_bdf == nullptr || out == nullptr || cap == 0)
        return 0;
    out[0] = 0;
    const std::string dev = sysfs_root() + "/bus/pci/devices/" + lower(pci_bdf);
    std::string list;
    if (!read_line(dev + "/local_cpulist", &list) || list.empty()) {
        std::string node;
        if (!read_line(dev + "/numa_node", &node) || node.empty() || node[0] == '-')
            return 0;   // the kernel knows no node for this function (-1 on single-node hosts)
        if (!read_line(sysfs_root() + "/devices/system/node/node" + node + "/cpulist", &list) || list.empty())
            return 0;
    }
    std::vector<int> cpus;
    if (!parse_cpulist(list.c_str(), &cpus) || list.size() + 1 > cap)
        return 0;
    std::memcpy(out, list.c_str(), list.size() + 1);
    return (int32_t)list.size();
}

int32_t dxtlt_bind_thread_to_cpulist(const char* cpulist)
{
    std::vector<int> cpus;
    if (cpulist == nullptr || !parse_cpulist(cpulist, &cpus))
        return 0;
    cpu_set_t allowed, want;
    CPU_ZERO(&allowed);
    CPU_ZERO(&want);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0)
        return 0;
    int n = 0;
    for (int c : cpus) {
        if (CPU_ISSET(c, &allowed)) {   // a cpuset / container may leave this process only part of the node
            CPU_SET(c, &want);
            ++n;
        }
    }
    if (n == 0)
        return 0;   // none of the node's CPUs is ours: stay where we are
    return sched_setaffinity(0, sizeof want, &want) == 0 ? n : 0;
}

int32_t dxtlt_device_local_cpulist(int32_t device, char* out, size_t cap)
{
    if (out == nullptr || cap == 0 || device < 0)
        return 0;
    out[0] = 0;
    // a device ordinal the runtime does not know must not size the cache (a huge one would allocate gigabytes, or throw
    // through this extern "C" boundary); nothing below may throw either
    int known = 0;
    if (hipGetDeviceCount(&known) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    if (device >= known)
        return 0;
    try {
    std::string cached = "?";
    {
        std::lock_guard<std::mutex> lk(g_cache_mutex);
        if ((size_t)device < g_device_cpulist.size())
            cached = g_device_cpulist[(size_t)device];
    }
    if (cached == "?") {
        char bdf[64] = {0};
        cached.clear();
        if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device) == hipSuccess) {
            char list[4096];
            if (dxtlt_pci_local_cpulist(bdf, list, sizeof list) > 0)
                cached = list;
        } else {
            (void)hipGetLastError();
        }
        std::lock_guard<std::mutex> lk(g_cache_mutex);
        if ((size_t)device >= g_device_cpulist.size())
            g_device_cpulist.resize((size_t)device + 1, "?");
        g_device_cpulist[(size_t)device] = cached;
    }
    if (cached.empty() || cached.size() + 1 > cap)
        return 0;
    std::memcpy(out, cached.c_str(), cached.size() + 1);
    return (int32_t)cached.size();
    } catch (const std::exception&) {   // std::bad_alloc from the cache
        return 0;
    }
}

}  // extern "C"

int dxtlt_host::bind_this_thread_near_device(int device)
{
    static const bool enabled = [] {
        const char* v = std::getenv("DXTLT_NUMA_BIND");
        return !(v && v[0] == '0');
    }();
    if (!enabled)
        return 0;
    char list[4096];
    if (dxtlt_device_local_cpulist(device, list, sizeof list) <= 0)
        return 0;
    return dxtlt_bind_thread_to_cpulist(list);
}
