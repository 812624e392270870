// Weight-gradient side stream.  In a backward pass the weight-gradient GEMMs (+ their slab reductions) and the small
// parameter-gradient reductions are off the critical path dout -> dinp: they run on a second, low-priority HIP stream,
// forked/joined with events inside ONE backward call (callers see plain stream semantics), so that the HBM-bound kernels of
// the main chain (LayerNorm / depthwise / SCA backward) and the launch ramps and tails of its GEMMs overlap MFMA-bound
// wgrad work instead of leaving the matrix cores idle.  dcpt_set_side_stream(0) or DCPT_SIDE_STREAM=0 in the environment
// keeps everything on the caller's stream; so does a stream that is being captured into a graph.
#include "side.h"

#include <stdlib.h>

#include <map>
#include <mutex>
#include <utility>

#include "../../include/dcpt_hip.h"

struct Side {
    hipStream_t ss = nullptr;
    hipEvent_t ev[SIDE_EVENTS] = {};
};

namespace {
std::mutex g_side_mu;
std::map<std::pair<int, hipStream_t>, Side*> g_sides;   // keyed by (device, stream): the null-stream handle is the same on every device
int g_side_enabled = -1;

void side_init_locked() {
    if (g_side_enabled < 0) {
        g_side_enabled = dcpt_tuning("DCPT_SIDE_STREAM", 1) ? 1 : 0;   // (the product switch is dcpt_set_side_stream())
    }
}
}  // namespace

Side* side_for(hipStream_t main) {
    std::lock_guard<std::mutex> lk(g_side_mu);
    side_init_locked();
    if (!g_side_enabled) return nullptr;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(main, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return nullptr;   // keep graph captures single-stream
    }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    const auto key = std::make_pair(dev, main);
    auto it = g_sides.find(key);
    if (it != g_sides.end()) return it->second;
    Side* sd = new Side();
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    bool ok = hipStreamCreateWithPriority(&sd->ss, hipStreamNonBlocking, least) == hipSuccess;
    for (int i = 0; ok && i < SIDE_EVENTS; ++i) ok = hipEventCreateWithFlags(&sd->ev[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        delete sd;
        sd = nullptr;
    }
    g_sides[key] = sd;
    return sd;
}

hipStream_t side_stream(Side* sd, hipStream_t main) { return sd ? sd->ss : main; }

int side_fork(Side* sd, int i, hipStream_t main) {
    if (!sd) return DCPT_OK;
    if (i < 0 || i >= SIDE_EVENTS - 1 || hipEventRecord(sd->ev[i], main) != hipSuccess ||
        hipStreamWaitEvent(sd->ss, sd->ev[i], 0) != hipSuccess) {
        dcpt_set_error("side-stream fork failed: %s", hipGetErrorString(hipGetLastError()));
        return DCPT_ERR_HIP;
    }
    return DCPT_OK;
}

int side_join(Side* sd, hipStream_t main) {
    if (!sd) return DCPT_OK;
    hipEvent_t e = sd->ev[SIDE_EVENTS - 1];
    if (hipEventRecord(e, sd->ss) != hipSuccess || hipStreamWaitEvent(main, e, 0) != hipSuccess) {
        dcpt_set_error("side-stream join failed: %s", hipGetErrorString(hipGetLastError()));
        return DCPT_ERR_HIP;
    }
    return DCPT_OK;
}

extern "C" int dcpt_set_side_stream(int on) {
    std::lock_guard<std::mutex> lk(g_side_mu);
    side_init_locked();
    const int prev = g_side_enabled;
    g_side_enabled = on ? 1 : 0;
    return prev;
}
