#include "prof.h"

#include <string.h>

#include <mutex>
#include <vector>

#include "../../include/dcpt_hip.h"

namespace {
struct Rec {
    int cls;
    int64_t M;
    int N, K;
    double flops, bytes;
    hipEvent_t e0, e1;
};
std::mutex g_mu;
bool g_on = false;
int g_every = 1;       // bracket every g_every-th launch
unsigned g_count = 0;
std::vector<Rec> g_recs;         // records of the current session
std::vector<hipEvent_t> g_pool;  // reusable events
size_t g_pool_next = 0;
int g_open = -1;
constexpr size_t MAXREC = 1 << 16;

hipEvent_t get_event() {
    if (g_pool_next < g_pool.size()) return g_pool[g_pool_next++];
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    g_pool.push_back(e);
    g_pool_next++;
    return e;
}
}  // namespace

void prof_begin(hipStream_t s, int cls, int64_t M, int N, int K, double flops, double bytes) {
    if (!g_on) return;
    std::lock_guard<std::mutex> lk(g_mu);
    g_open = -1;
    if ((g_count++ % (unsigned)g_every) != 0) return;
    if (g_recs.size() >= MAXREC) return;
    Rec r{cls, M, N, K, flops, bytes, get_event(), get_event()};
    if (!r.e0 || !r.e1) return;
    hipEventRecord(r.e0, s);
    g_recs.push_back(r);
    g_open = (int)g_recs.size() - 1;
}

void prof_end(hipStream_t s) {
    if (!g_on) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_open >= 0) hipEventRecord(g_recs[g_open].e1, s);
    g_open = -1;
}

extern "C" int dcpt_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_on = on != 0;
    if (g_on) {
        g_every = on > 1 ? on : 1;
        g_count = 0;
        g_recs.clear();
        g_pool_next = 0;
    }
    return DCPT_OK;
}

// Synchronises the recorded events and aggregates per (class, M, N, K).  out: [max_rows][8] =
// {class id, M, N, K, launches, total ms, total flops, total bytes}; returns the number of rows.
extern "C" int dcpt_prof_read(double* out, int max_rows) {
    std::lock_guard<std::mutex> lk(g_mu);
    int n = 0;
    for (const Rec& r : g_recs) {
        if (hipEventSynchronize(r.e1) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
        int j = 0;
        for (; j < n; ++j)
            if ((int)out[j * 8] == r.cls && (int64_t)out[j * 8 + 1] == r.M && (int)out[j * 8 + 2] == r.N &&
                (int)out[j * 8 + 3] == r.K)
                break;
        if (j == n) {
            if (n >= max_rows) continue;
            out[j * 8] = r.cls;
            out[j * 8 + 1] = (double)r.M;
            out[j * 8 + 2] = r.N;
            out[j * 8 + 3] = r.K;
            out[j * 8 + 4] = out[j * 8 + 5] = out[j * 8 + 6] = out[j * 8 + 7] = 0.0;
            n++;
        }
        out[j * 8 + 4] += 1.0;
        out[j * 8 + 5] += ms;
        out[j * 8 + 6] += r.flops;
        out[j * 8 + 7] += r.bytes;
    }
    return n;
}

// ---- launch trace (prof.h) ----
bool g_trace_on = false;
namespace {
struct Tag {
    const char* name;
    long count;
};
std::mutex g_tmu;
std::vector<Tag> g_tags;
}  // namespace

void trace_tag_slow(const char* tag) {
    std::lock_guard<std::mutex> lk(g_tmu);
    for (Tag& t : g_tags)
        if (t.name == tag || strcmp(t.name, tag) == 0) {
            t.count++;
            return;
        }
    g_tags.push_back(Tag{tag, 1});
}

extern "C" int dcpt_trace_enable(int on) {
    std::lock_guard<std::mutex> lk(g_tmu);
    g_trace_on = on != 0;
    if (g_trace_on) g_tags.clear();
    return DCPT_OK;
}

// "tag count\n" per kernel family seen since dcpt_trace_enable(1); returns the bytes the full text needs (incl. the terminating 0)
extern "C" size_t dcpt_trace_read(char* buf, size_t cap) {
    std::lock_guard<std::mutex> lk(g_tmu);
    size_t need = 1, off = 0;
    for (const Tag& t : g_tags) {
        char line[160];
        const int n = snprintf(line, sizeof line, "%s %ld\n", t.name, t.count);
        need += (size_t)n;
        if (buf && off + (size_t)n < cap) {
            memcpy(buf + off, line, (size_t)n);
            off += (size_t)n;
        }
    }
    if (buf && cap) buf[off < cap ? off : cap - 1] = 0;
    return need;
}
