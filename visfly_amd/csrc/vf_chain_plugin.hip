// vf_chain_plugin.hip -- registry of the chain plugins (vf_chain_plugin.hpp): shared objects with the register-chained kernels of ONE
// network shape each, compiled on first use by the host side (visfly_amd/_jit.py) for shapes libvisfly_amd.so holds no instance of.
#include "vf_chain_plugin.hpp"

#include <dlfcn.h>

#include <atomic>
#include <mutex>
#include <string>
#include <vector>

namespace vf {
namespace {
struct Loaded {
    std::string path;
    void* handle;
    const ChainPlugin* p;
};
std::mutex g_mu;
std::vector<Loaded>& loaded()
{
    static std::vector<Loaded> v;
    return v;
}
}  // namespace

static std::atomic<long long> g_launches{0};
void chain_plugin_count_launch() { g_launches.fetch_add(1, std::memory_order_relaxed); }

static std::atomic<int> g_enabled{1};

int chain_plugin_count()
{
    if (!g_enabled.load(std::memory_order_relaxed)) return 0;
    std::lock_guard<std::mutex> lk(g_mu);
    return (int)loaded().size();
}

const ChainPlugin* chain_plugin(int i)
{
    std::lock_guard<std::mutex> lk(g_mu);
    return i >= 0 && i < (int)loaded().size() ? loaded()[i].p : nullptr;
}

}  // namespace vf

extern "C" int vf_chain_plugin_load(const char* path)
{
    using namespace vf;
    if (!path) return fail(VF_EINVAL, "vf_chain_plugin_load: null path");
    std::lock_guard<std::mutex> lk(g_mu);
    for (const Loaded& l : loaded())
        if (l.path == path) return VF_OK;
    void* h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) return fail(VF_EINVAL, "vf_chain_plugin_load: %s", dlerror());
    typedef const ChainPlugin* (*entry_t)();
    entry_t entry = reinterpret_cast<entry_t>(dlsym(h, "vf_chain_plugin"));
    const ChainPlugin* p = entry ? entry() : nullptr;
    if (!p || p->abi != kChainPluginAbi || !((p->forward && p->backward && p->ppo_update) || p->ppo_rollout || (p->bptt_rollout && p->bptt_reverse))) {
        dlclose(h);
        return fail(VF_EINVAL, "vf_chain_plugin_load: %s is not a chain plugin of this library build (abi %08x, expected %08x)", path,
                    p ? p->abi : 0u, kChainPluginAbi);
    }
    loaded().push_back(Loaded{path, h, p});
    return VF_OK;
}

extern "C" int vf_chain_plugin_count()
{
    std::lock_guard<std::mutex> lk(vf::g_mu);
    return (int)vf::loaded().size();
}

extern "C" int vf_chain_plugin_set_enabled(int on)
{
    return vf::g_enabled.exchange(on ? 1 : 0);
}

extern "C" const char* vf_chain_plugin_name(int i)
{
    std::lock_guard<std::mutex> lk(vf::g_mu);
    return i >= 0 && i < (int)vf::loaded().size() ? vf::loaded()[i].p->name : nullptr;
}

extern "C" int64_t vf_chain_plugin_launches() { return vf::g_launches.load(std::memory_order_relaxed); }
