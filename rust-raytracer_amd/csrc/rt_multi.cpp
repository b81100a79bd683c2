// rt_multi.cpp -- one frame on several GPUs: the gather of the parts' tiles over RCCL, for a
// single process that drives all devices (RtMulti) and for a launcher with one process per device
// (RtComm).  Both post the same operations: every part that does not already lie in the root's
// gather buffer is ONE ncclSend on its device and ONE ncclRecv on the root's, all inside one
// group, so the seven senders of an eight-GPU node use their seven xGMI links to the root at the
// same time; with a group of frames whose root rotates (rt_comm_gather_frames) every link carries
// tiles in both directions at once.  Nothing else of a frame crosses a link.
//
// librccl is loaded with dlopen when the first communicator is made: a process that renders on one
// GPU never maps it (the library is several hundred megabytes of kernels for every collective).

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>
#include <type_traits>

#include "rt_backend_internal.h"

namespace {

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    std::string error;
};

// The process-wide binding; nullptr (and rt_last_error) when librccl cannot be loaded.
const Rccl *rccl() {
    static std::mutex mu;
    static Rccl api;
    static bool tried = false;
    std::lock_guard<std::mutex> lock(mu);
    if (!tried) {
        tried = true;
        // by soname first: a copy that the process already holds (a framework's) is then the one used
        for (const char *name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
            api.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (api.handle) break;
        }
        if (!api.handle) {
            const char *why = dlerror();  // once: the call hands the message over and clears it
            api.error = std::string("librccl could not be loaded: ") + (why ? why : "unknown error");
        } else {
            bool all = true;
            auto bind = [&](auto &slot, const char *symbol) {
                slot = reinterpret_cast<std::remove_reference_t<decltype(slot)>>(dlsym(api.handle, symbol));
                if (!slot) {
                    all = false;
                    api.error = std::string("librccl lacks ") + symbol;
                }
            };
            bind(api.GetUniqueId, "ncclGetUniqueId");
            bind(api.CommInitRank, "ncclCommInitRank");
            bind(api.CommInitAll, "ncclCommInitAll");
            bind(api.CommDestroy, "ncclCommDestroy");
            bind(api.GroupStart, "ncclGroupStart");
            bind(api.GroupEnd, "ncclGroupEnd");
            bind(api.Send, "ncclSend");
            bind(api.Recv, "ncclRecv");
            bind(api.GetErrorString, "ncclGetErrorString");
            bind(api.GetVersion, "ncclGetVersion");
            if (!all) api.handle = nullptr;
        }
    }
    if (!api.handle) {
        rt::set_error(RT_ERR_DEVICE, api.error);
        return nullptr;
    }
    return &api;
}

#define RT_NCCL(api, call)                                                                                 \
    do {                                                                                                   \
        ncclResult_t r_ = (call);                                                                          \
        if (r_ != ncclSuccess)                                                                             \
            return rt::set_error(RT_ERR_DEVICE, std::string(#call) + ": " + (api)->GetErrorString(r_));   \
    } while (0)

static_assert(RT_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "RT_COMM_ID_BYTES must hold an ncclUniqueId");

}  // namespace

// ------------------------------------------------------------------ one process per GPU

struct RtComm {
    RtBackend *backend = nullptr;
    ncclComm_t comm = nullptr;
    uint32_t rank = 0, world = 1;
};

extern "C" int rt_comm_rccl_version(int *version_out) {
    return rt::guarded([&]() -> int {
        if (!version_out) return rt::set_error(RT_ERR_INVALID, "null argument");
        *version_out = 0;
        const Rccl *api = rccl();
        if (!api) return RT_ERR_DEVICE;
        RT_NCCL(api, api->GetVersion(version_out));
        return RT_OK;
    });
}

extern "C" int rt_comm_id(uint8_t id_out[RT_COMM_ID_BYTES]) {
    return rt::guarded([&]() -> int {
        if (!id_out) return rt::set_error(RT_ERR_INVALID, "null argument");
        const Rccl *api = rccl();
        if (!api) return RT_ERR_DEVICE;
        ncclUniqueId id;
        RT_NCCL(api, api->GetUniqueId(&id));
        std::memcpy(id_out, id.internal, RT_COMM_ID_BYTES);
        return RT_OK;
    });
}

extern "C" int rt_comm_create(RtBackend *be, const uint8_t id_bytes[RT_COMM_ID_BYTES], uint32_t rank, uint32_t world,
                              RtComm **out) {
    return rt::guarded([&]() -> int {
        if (!be || !id_bytes || !out) return rt::set_error(RT_ERR_INVALID, "null argument");
        *out = nullptr;
        if (world == 0 || rank >= world) return rt::set_error(RT_ERR_INVALID, "rank out of range");
        const Rccl *api = rccl();
        if (!api) return RT_ERR_DEVICE;
        RT_HIP(hipSetDevice(be->device));
        ncclUniqueId id;
        std::memcpy(id.internal, id_bytes, RT_COMM_ID_BYTES);
        ncclComm_t comm = nullptr;
        RT_NCCL(api, api->CommInitRank(&comm, (int)world, id, (int)rank));
        RtComm *c = new RtComm();
        c->backend = be;
        c->comm = comm;
        c->rank = rank;
        c->world = world;
        *out = c;
        return RT_OK;
    });
}

extern "C" void rt_comm_destroy(RtComm *c) {
    if (!c) return;
    const Rccl *api = rccl();
    if (api && c->comm) {
        (void)hipSetDevice(c->backend->device);
        (void)api->CommDestroy(c->comm);
    }
    delete c;
}

extern "C" int rt_comm_gather_frames(RtComm *c, const float *parts_dev, size_t floats, uint32_t count, uint32_t first_root,
                                     uint32_t slots, float *gathered_dev, void *hip_stream) {
    return rt::guarded([&]() -> int {
        if (!c || !parts_dev) return rt::set_error(RT_ERR_INVALID, "null argument");
        const uint32_t world = c->world;
        if (first_root >= world) return rt::set_error(RT_ERR_INVALID, "root out of range");
        if ((uint64_t)slots * world < count) return rt::set_error(RT_ERR_INVALID, "slots * world must hold the group's frames");
        if (floats == 0 || count == 0) return RT_OK;
        // position p in the rotation: the root (first_root + p) % world receives the frames p, p + world, ...
        auto frames_at = [&](uint32_t p) { return p < count ? (count - p + world - 1) / world : 0u; };
        const uint32_t my_p = (c->rank + world - first_root) % world;
        const uint32_t mine = frames_at(my_p);
        if (mine && !gathered_dev) return rt::set_error(RT_ERR_INVALID, "a root needs a gather buffer");
        const Rccl *api = rccl();
        if (!api) return RT_ERR_DEVICE;
        RT_HIP(hipSetDevice(c->backend->device));
        hipStream_t stream = hip_stream ? (hipStream_t)hip_stream : c->backend->stream;
        RT_NCCL(api, api->GroupStart());
        ncclResult_t r = ncclSuccess;
        for (uint32_t p = 0; p < world && r == ncclSuccess; ++p) {
            const uint32_t k = frames_at(p);
            if (k) r = api->Send(parts_dev + (size_t)p * slots * floats, (size_t)k * floats, ncclFloat, (int)((first_root + p) % world), c->comm, stream);
        }
        if (mine)
            for (uint32_t peer = 0; peer < world && r == ncclSuccess; ++peer)
                r = api->Recv(gathered_dev + (size_t)peer * slots * floats, (size_t)mine * floats, ncclFloat, (int)peer, c->comm, stream);
        const ncclResult_t end = api->GroupEnd();
        RT_NCCL(api, r);
        RT_NCCL(api, end);
        return RT_OK;
    });
}

extern "C" int rt_comm_gather(RtComm *c, const float *part_dev, size_t floats, float *gathered_dev, uint32_t root,
                              void *hip_stream) {
    if (c && root >= c->world) return rt::set_error(RT_ERR_INVALID, "root out of range");
    if (c && c->rank == root && !gathered_dev) return rt::set_error(RT_ERR_INVALID, "the root needs a gather buffer");
    return rt_comm_gather_frames(c, part_dev, floats, 1, root, 1, gathered_dev, hip_stream);
}

// ------------------------------------------------------------------ one process, several GPUs

namespace {

// A host thread that enqueues one part's kernels whenever the frame thread says so.  The parts'
// enqueues (a dozen launches each) then run side by side instead of one after the other.
struct Worker {
    std::thread thread;
    std::mutex mu;
    std::condition_variable cv;
    bool has_job = false, done = false, quit = false;
    // the job
    RtRenderArgs args{};
    float *dst = nullptr;
    RtStats *stats = nullptr;
    // its outcome
    int rc = RT_OK;
    std::string message;
};

}  // namespace

struct RtMulti {
    uint32_t n = 0;
    std::vector<RtBackend *> backends;
    std::vector<RtDeviceScene *> scenes;
    // one communicator rank per distinct device, in order of first appearance: rank 0 = the root's
    std::vector<int> devices;
    std::vector<ncclComm_t> comms;
    std::vector<int> rank_of_part;           // index into devices / comms
    std::vector<hipStream_t> comm_stream;    // per rank: the stream its sends (root: receives) are enqueued on
    std::vector<hipEvent_t> part_done;       // per part: recorded behind its kernels when its stream is not the rank's
    std::vector<Worker *> workers;
    float *gathered = nullptr;               // root device: n blocks of tiles
    size_t gathered_floats = 0;
};

namespace {

void worker_loop(RtMulti *m, uint32_t k) {
    Worker &w = *m->workers[k];
    for (;;) {
        std::unique_lock<std::mutex> lock(w.mu);
        w.cv.wait(lock, [&] { return w.has_job || w.quit; });
        if (w.quit) return;
        w.has_job = false;
        lock.unlock();
        RtBackend *be = m->backends[k];
        int rc = hipSetDevice(be->device) == hipSuccess ? RT_OK : rt::set_error(RT_ERR_DEVICE, "hipSetDevice failed");
        if (!rc) rc = rt::render_frame(be, m->scenes[k], &w.args, w.dst, nullptr, be->stream, w.stats, false, false);
        if (!rc && m->part_done[k] && hipEventRecord(m->part_done[k], be->stream) != hipSuccess)
            rc = rt::set_error(RT_ERR_DEVICE, "hipEventRecord failed");
        lock.lock();
        w.rc = rc;
        w.message = rc ? rt::last_error_slot() : std::string();  // the slot is per thread
        w.done = true;
        lock.unlock();
        w.cv.notify_all();
    }
}

}  // namespace

extern "C" void rt_multi_destroy(RtMulti *m) {
    if (!m) return;
    for (Worker *w : m->workers) {
        if (!w) continue;
        {
            std::lock_guard<std::mutex> lock(w->mu);
            w->quit = true;
        }
        w->cv.notify_all();
        if (w->thread.joinable()) w->thread.join();
        delete w;
    }
    for (uint32_t k = 0; k < m->n && k < m->backends.size(); ++k) {
        (void)hipSetDevice(m->backends[k]->device);
        (void)hipStreamSynchronize(m->backends[k]->stream);
        if (k < m->part_done.size() && m->part_done[k]) (void)hipEventDestroy(m->part_done[k]);
    }
    const Rccl *api = m->comms.empty() ? nullptr : rccl();
    for (size_t r = 0; r < m->comms.size(); ++r)
        if (api && m->comms[r]) {
            (void)hipSetDevice(m->devices[r]);
            (void)api->CommDestroy(m->comms[r]);
        }
    if (m->gathered) {
        (void)hipSetDevice(m->backends[0]->device);
        (void)hipFree(m->gathered);
    }
    delete m;
}

extern "C" int rt_multi_create(RtBackend *const *bes, RtDeviceScene *const *dss, uint32_t n, RtMulti **out) {
    return rt::guarded([&]() -> int {
        if (!bes || !dss || !out || n == 0) return rt::set_error(RT_ERR_INVALID, "null argument");
        *out = nullptr;
        for (uint32_t k = 0; k < n; ++k) {
            if (!bes[k] || !dss[k]) return rt::set_error(RT_ERR_INVALID, "null backend or scene");
            if (dss[k]->device != bes[k]->device) return rt::set_error(RT_ERR_INVALID, "scene k must lie on backend k's GPU");
            if (dss[k]->width != dss[0]->width || dss[k]->height != dss[0]->height)
                return rt::set_error(RT_ERR_INVALID, "the scenes of the parts differ in size");
            for (uint32_t j = 0; j < k; ++j)
                if (bes[j] == bes[k]) return rt::set_error(RT_ERR_INVALID, "every part needs a backend of its own");
        }
        RtMulti *m = new RtMulti();
        m->n = n;
        m->backends.assign(bes, bes + n);
        m->scenes.assign(dss, dss + n);
        m->rank_of_part.resize(n);
        m->part_done.assign(n, nullptr);
        for (uint32_t k = 0; k < n; ++k) {
            const int dev = bes[k]->device;
            size_t r = std::find(m->devices.begin(), m->devices.end(), dev) - m->devices.begin();
            if (r == m->devices.size()) {
                m->devices.push_back(dev);
                m->comm_stream.push_back(bes[k]->stream);  // the rank's stream = that of its first part
            }
            m->rank_of_part[k] = (int)r;
        }
        auto fail = [&](int rc) {
            rt_multi_destroy(m);
            return rc;
        };
        if (n > 1) {
            const Rccl *api = rccl();
            if (!api) return fail(RT_ERR_DEVICE);
            m->comms.assign(m->devices.size(), nullptr);
            ncclResult_t r = api->CommInitAll(m->comms.data(), (int)m->devices.size(), m->devices.data());
            if (r != ncclSuccess) return fail(rt::set_error(RT_ERR_DEVICE, std::string("ncclCommInitAll: ") + api->GetErrorString(r)));
            for (uint32_t k = 0; k < n; ++k) {
                if (bes[k]->stream == m->comm_stream[(size_t)m->rank_of_part[k]]) continue;
                if (hipSetDevice(bes[k]->device) != hipSuccess ||
                    hipEventCreateWithFlags(&m->part_done[k], hipEventDisableTiming) != hipSuccess)
                    return fail(rt::set_error(RT_ERR_DEVICE, "hipEventCreate failed"));
            }
        }
        m->workers.assign(n, nullptr);
        for (uint32_t k = 0; k < n; ++k) {
            m->workers[k] = new Worker();
            m->workers[k]->thread = std::thread(worker_loop, m, k);
        }
        *out = m;
        return RT_OK;
    });
}

extern "C" int rt_render_multi(RtMulti *m, const RtRenderArgs *args, float *fb_rgb, RtStats *stats_out) {
    return rt::guarded([&]() -> int {
        if (!m || !args || !fb_rgb) return rt::set_error(RT_ERR_INVALID, "null argument");
        const uint32_t n = m->n;
        if (n == 1) {
            RtRenderArgs a = *args;
            a.part_index = 0;
            a.part_count = 1;
            return rt_render(m->backends[0], m->scenes[0], &a, fb_rgb, nullptr, stats_out);
        }
        const auto t_begin = std::chrono::steady_clock::now();
        const uint32_t W = m->scenes[0]->width, H = m->scenes[0]->height;
        RtRenderArgs a0 = *args;
        a0.part_index = 0;
        a0.part_count = n;
        a0.progress = nullptr;
        uint32_t stride = 0;
        int rc = rt_partition_info(W, H, &a0, nullptr, nullptr, &stride);
        if (rc) return rc;
        const uint32_t tw = args->tile_w ? args->tile_w : 32, th = args->tile_h ? args->tile_h : 32;
        const size_t block_floats = (size_t)stride * tw * th * 3;
        const size_t fb_floats = (size_t)W * H * 3;
        RtBackend *root = m->backends[0];

        // ---- buffers: grown on the first frame (or a larger one), kept afterwards
        RT_HIP(hipSetDevice(root->device));
        if (m->gathered_floats < (size_t)n * block_floats) {
            if (m->gathered) (void)hipFree(m->gathered);
            m->gathered = nullptr;
            m->gathered_floats = 0;
            RT_HIP(hipMalloc((void **)&m->gathered, (size_t)n * block_floats * sizeof(float)));
            m->gathered_floats = (size_t)n * block_floats;
        }
        if ((rc = rt::ensure_device_fb(root, fb_floats))) return rc;
        const bool direct = rt::is_pinned_host(fb_rgb);
        if (!direct && (rc = rt::ensure_pinned(root, fb_floats))) return rc;
        for (uint32_t k = 1; k < n; ++k) {
            RT_HIP(hipSetDevice(m->backends[k]->device));
            if ((rc = rt::ensure_device_fb(m->backends[k], block_floats))) return rc;
        }

        // ---- every part enqueues its kernels, side by side; part 0 renders straight into block 0
        std::vector<RtStats> part_stats(stats_out ? n : 0);
        for (uint32_t k = 0; k < n; ++k) {
            Worker &w = *m->workers[k];
            {
                std::lock_guard<std::mutex> lock(w.mu);
                w.args = a0;
                w.args.part_index = k;
                w.dst = k == 0 ? m->gathered : m->backends[k]->fb;
                w.stats = stats_out ? &part_stats[k] : nullptr;
                w.done = false;
                w.has_job = true;
            }
            w.cv.notify_all();
        }
        int first_rc = RT_OK;
        std::string first_msg;
        for (uint32_t k = 0; k < n; ++k) {
            Worker &w = *m->workers[k];
            std::unique_lock<std::mutex> lock(w.mu);
            w.cv.wait(lock, [&] { return w.done; });
            if (w.rc && !first_rc) {
                first_rc = w.rc;
                first_msg = "part " + std::to_string(k) + ": " + w.message;
            }
        }
        if (first_rc) {  // let what was enqueued run out before the buffers are touched again
            for (uint32_t k = 0; k < n; ++k) {
                (void)hipSetDevice(m->backends[k]->device);
                (void)hipStreamSynchronize(m->backends[k]->stream);
            }
            return rt::set_error(first_rc, first_msg);
        }

        // ---- the gather: one group, one send per part and its receive on the root
        const Rccl *api = rccl();
        if (!api) return RT_ERR_DEVICE;
        for (uint32_t k = 1; k < n; ++k)
            if (m->part_done[k]) {
                RT_HIP(hipSetDevice(m->backends[k]->device));
                RT_HIP(hipStreamWaitEvent(m->comm_stream[(size_t)m->rank_of_part[k]], m->part_done[k], 0));
            }
        RT_NCCL(api, api->GroupStart());
        ncclResult_t r = ncclSuccess;
        for (uint32_t k = 1; k < n && r == ncclSuccess; ++k) {
            const int rank = m->rank_of_part[k];
            r = api->Send(m->backends[k]->fb, block_floats, ncclFloat, 0, m->comms[(size_t)rank], m->comm_stream[(size_t)rank]);
            if (r == ncclSuccess)
                r = api->Recv(m->gathered + (size_t)k * block_floats, block_floats, ncclFloat, rank, m->comms[0], m->comm_stream[0]);
        }
        const ncclResult_t end = api->GroupEnd();
        RT_NCCL(api, r);
        RT_NCCL(api, end);

        // ---- image layout and the way home, behind the receives on the root's stream
        RT_HIP(hipSetDevice(root->device));
        if ((rc = rt_untile_device(root, W, H, &a0, m->gathered, root->fb, m->comm_stream[0]))) return rc;
        float *landing = direct ? fb_rgb : root->pinned;
        RT_HIP(hipMemcpyAsync(landing, root->fb, fb_floats * sizeof(float), hipMemcpyDeviceToHost, m->comm_stream[0]));
        if ((rc = rt::finish_stream(root, m->comm_stream[0]))) return rc;
        for (uint32_t k = 1; k < n; ++k) {  // their sends have been received: these waits return at once
            RT_HIP(hipSetDevice(m->backends[k]->device));
            if ((rc = rt::finish_stream(m->backends[k], m->backends[k]->stream))) return rc;
        }
        if (!direct) std::memcpy(fb_rgb, landing, fb_floats * sizeof(float));
        if (stats_out) {
            std::memset(stats_out, 0, sizeof(*stats_out));
            for (const RtStats &p : part_stats) {
                stats_out->primary_rays += p.primary_rays;
                stats_out->secondary_rays += p.secondary_rays;
                stats_out->shadow_rays += p.shadow_rays;
                stats_out->shaded_hits += p.shaded_hits;
                for (int c = 0; c < RT_K_COUNT; ++c) {
                    stats_out->kernel_ms[c] += p.kernel_ms[c];
                    stats_out->kernel_launches[c] += p.kernel_launches[c];
                }
            }
            stats_out->render_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
        }
        return RT_OK;
    });
}
