// host_tree.h -- part of suchtree_hip.hip (included in this order: host_tree.h, host_launch.h, host_path.h,
// host_upload.h).  The per-device staging pipe registry; the tree handle itself is in st_tree.h.
#pragma once
#include "st_tree.h"

// One staging pipe (pinned + device buffers, one stream per slot, copy pool) per GPU, shared by every
// tree of the process on that GPU: SuchLinkedTrees holds two trees, applications hold many,
// and the staging is ~200 MB of pinned memory and up to 15 threads per pipe.  Reference
// counted; the mutex admits one host-path call at a time per device.
struct DevicePipe {
    std::mutex m;
    HostPipe pipe;
    int refs = 0;
};
static std::mutex g_pipes_mutex;
static std::map<int, DevicePipe *> g_pipes;

static DevicePipe *pipe_acquire(int device)
{
    std::lock_guard<std::mutex> g(g_pipes_mutex);
    DevicePipe *&p = g_pipes[device];
    if (!p) p = new DevicePipe();
    p->refs++;
    return p;
}

static void pipe_release(int device)
{
    std::lock_guard<std::mutex> g(g_pipes_mutex);
    auto it = g_pipes.find(device);
    if (it == g_pipes.end()) return;
    if (--it->second->refs > 0) return;
    {
        DeviceScope scope(device);
        it->second->pipe.destroy();
    }
    delete it->second;
    g_pipes.erase(it);
}

