// Node-level driver: one process, one engine (weights replica, streams, workspaces) and one worker thread per GPU,
// data-parallel by utterance. The workers pull batches of utterances from one atomic counter (a rank whose utterances
// decode for longer simply pulls fewer), and the only exchange between GPUs is the gather of fixed-size result records
// on the first device at the end of the job -- RCCL point-to-point over xGMI, loaded at run time so that a process that
// never runs a multi-GPU job carries no collective library.
// The reference has no counterpart: Handy owns one engine behind one mutex
// (src-tauri/src/managers/transcription.rs:39, :383) and its only concurrent caller queues on it
// (src-tauri/src/discord_conversation.rs:470-476). This is what an offline bulk driver binds.
#include <dlfcn.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <thread>

#include "engine.h"

using namespace wmi;

struct wmi_node {
    std::vector<int> devices;
    std::vector<whisper_context *> ctx;
    std::vector<whisper_state *> state;
    int batch = 0;
    int last_gather = 0;            // 0 host memory, 1 RCCL
    std::vector<int> last_batches;  // batches each worker ran in the last job
};

namespace {

// ---- the job queue: batch b = utterances [b * batch, min(n_utts, (b + 1) * batch)) ----------------------------------------
// fn(user, worker, first, count, out): run one batch on worker `worker`, fill out[0..count) (utt, device are set by the
// caller afterwards). A non-zero return marks every record of that batch failed with that code; the job goes on.
int run_queue(int n_workers, int n_utts, int batch, wmi_node_batch_fn fn, void *user, wmi_node_record *out, int *batches_per_worker) {
    if (n_workers < 1 || n_workers > 64 || n_utts < 0 || batch < 1 || !fn || (n_utts > 0 && !out)) return -1;
    const int n_batches = (n_utts + batch - 1) / batch;
    std::atomic<int> next{0};
    std::vector<int> ran(n_workers, 0);
    std::atomic<int> hard_fail{0};
    auto work = [&](int w) {
        for (;;) {
            const int b = next.fetch_add(1, std::memory_order_relaxed);
            if (b >= n_batches) break;
            const int first = b * batch, count = std::min(batch, n_utts - first);
            wmi_node_record *rec = out + first;            // a batch's records have one writer: no lock
            memset(rec, 0, sizeof(wmi_node_record) * (size_t)count);
            int rc = -1;
            try { rc = fn(user, w, first, count, rec); } catch (...) { rc = -1; hard_fail.store(1); }
            for (int i = 0; i < count; i++) {
                rec[i].utt = first + i;
                rec[i].device = w;
                if (rc != 0) { rec[i].rc = rc; rec[i].n_tokens = 0; rec[i].n_segments = 0; }
            }
            ran[w]++;
        }
    };
    std::vector<std::thread> th;
    try {
        for (int w = 1; w < n_workers; w++) th.emplace_back(work, w);
    } catch (...) { hard_fail.store(1); }                   // fewer threads than asked for: the ones that started still drain the queue
    work(0);
    for (auto &t : th) t.join();
    if (batches_per_worker) for (int w = 0; w < n_workers; w++) batches_per_worker[w] = ran[w];
    return hard_fail.load() ? -2 : 0;
}

// ---- RCCL, resolved at run time ---------------------------------------------------------------------------------------
struct Rccl {
    void *h = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool load() {
        if (h) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (h) break;
        }
        if (!h) return false;
        CommInitAll = (decltype(CommInitAll))dlsym(h, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(h, "ncclCommDestroy");
        GroupStart = (decltype(GroupStart))dlsym(h, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(h, "ncclGroupEnd");
        Send = (decltype(Send))dlsym(h, "ncclSend");
        Recv = (decltype(Recv))dlsym(h, "ncclRecv");
        GetErrorString = (decltype(GetErrorString))dlsym(h, "ncclGetErrorString");
        return CommInitAll && CommDestroy && GroupStart && GroupEnd && Send && Recv;
    }
};
Rccl g_rccl;
std::mutex g_rccl_mu;

// Gather over RCCL: worker w's records (rows[w], by utterance index into `all`) travel device w -> device 0 as one padded
// block of `cap` records; device 0 copies the blocks to the host. Returns false (and leaves `out` alone) on any failure:
// the caller then keeps the records it already has in host memory.
bool gather_rccl(const std::vector<int> &devices, const std::vector<std::vector<int>> &rows, const wmi_node_record *all, wmi_node_record *out, int n_utts) {
    const int n = (int)devices.size();
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (!g_rccl.load()) { log_msg(2, "wmi_node_run: librccl not found, records gathered through host memory\n"); return false; }
    size_t cap = 1;
    for (auto &r : rows) cap = std::max(cap, r.size());
    const size_t blk = cap * sizeof(wmi_node_record);
    std::vector<void *> comms(n, nullptr);
    std::vector<void *> send(n, nullptr);
    std::vector<hipStream_t> st(n, nullptr);
    void *recv0 = nullptr;
    std::vector<wmi_node_record> stage(cap), gathered((size_t)n * cap);
    bool ok = g_rccl.CommInitAll(comms.data(), n, devices.data()) == 0;
    if (!ok) log_msg(2, "wmi_node_run: ncclCommInitAll failed, records gathered through host memory\n");
    for (int w = 0; ok && w < n; w++) {
        ok = hipSetDevice(devices[w]) == hipSuccess && hipStreamCreateWithFlags(&st[w], hipStreamNonBlocking) == hipSuccess && hipMalloc(&send[w], blk) == hipSuccess;
        if (!ok) break;
        for (auto &r : stage) { memset(&r, 0, sizeof r); r.utt = -1; }
        for (size_t i = 0; i < rows[w].size(); i++) stage[i] = all[rows[w][i]];
        ok = hipMemcpy(send[w], stage.data(), blk, hipMemcpyHostToDevice) == hipSuccess;
    }
    if (ok) ok = hipSetDevice(devices[0]) == hipSuccess && hipMalloc(&recv0, blk * n) == hipSuccess;
    if (ok) {
        // every rank's call sits inside one group (one thread drives all devices): rank 0 receives from each peer, keeps its own block
        ok = g_rccl.GroupStart() == 0;
        for (int w = 1; ok && w < n; w++) {
            ok = g_rccl.Send(send[w], blk, /*ncclChar*/ 0, 0, comms[w], st[w]) == 0 &&
                 g_rccl.Recv((char *)recv0 + blk * w, blk, 0, w, comms[0], st[0]) == 0;
        }
        ok = (g_rccl.GroupEnd() == 0) && ok;
        for (int w = 0; ok && w < n; w++) ok = hipSetDevice(devices[w]) == hipSuccess && hipStreamSynchronize(st[w]) == hipSuccess;
        if (ok) ok = hipSetDevice(devices[0]) == hipSuccess && hipMemcpy(recv0, send[0], blk, hipMemcpyDeviceToDevice) == hipSuccess &&
                     hipMemcpy(gathered.data(), recv0, blk * n, hipMemcpyDeviceToHost) == hipSuccess;
        if (!ok) log_msg(2, "wmi_node_run: RCCL gather failed, records gathered through host memory\n");
    }
    if (ok) {
        std::vector<char> seen(n_utts, 0);
        int got = 0;
        for (auto &r : gathered)
            if (r.utt >= 0 && r.utt < n_utts && !seen[r.utt]) { seen[r.utt] = 1; got++; }
        ok = got == n_utts;
        if (ok) {
            for (auto &r : gathered)
                if (r.utt >= 0) out[r.utt] = r;
        } else log_msg(2, "wmi_node_run: RCCL gather returned %d of %d records, records gathered through host memory\n", got, n_utts);
    }
    for (int w = 0; w < n; w++) {
        if (hipSetDevice(devices[w]) != hipSuccess) continue;
        if (send[w]) (void)hipFree(send[w]);
        if (st[w]) (void)hipStreamDestroy(st[w]);
    }
    if (recv0 && hipSetDevice(devices[0]) == hipSuccess) (void)hipFree(recv0);
    for (int w = 0; w < n; w++) if (comms[w]) (void)g_rccl.CommDestroy(comms[w]);
    return ok;
}

bool distinct_or_single(const std::vector<int> &devs) {
    for (size_t a = 0; a < devs.size(); a++) for (size_t b = a + 1; b < devs.size(); b++) if (devs[a] == devs[b]) return false;
    return !devs.empty();
}

struct EngineJob {
    wmi_node *node;
    const whisper_full_params *params;
    const float *const *samples;
    const int *n_samples;
};

int engine_batch(void *user, int w, int first, int count, wmi_node_record *out) {
    EngineJob *job = (EngineJob *)user;
    wmi_node *node = job->node;
    if (hipSetDevice(node->devices[w]) != hipSuccess) return -4;
    whisper_state *s = node->state[w];
    const int rc = wmi_full_batch(node->ctx[w], s, *job->params, job->samples + first, job->n_samples + first, count, 0);
    if (rc != 0) return rc;
    for (int u = 0; u < count; u++) {
        wmi_node_record &r = out[u];
        r.rc = wmi_batch_rc(s, u);
        r.lang_id = wmi_batch_lang_id(s, u);
        r.n_segments = wmi_batch_n_segments(s, u);
        int nt = 0;
        for (int g = 0; g < r.n_segments; g++) {
            if (g == 0) r.t0 = wmi_batch_segment_t0(s, u, g);
            r.t1 = wmi_batch_segment_t1(s, u, g);
            const int k = wmi_batch_n_tokens(s, u, g);
            for (int j = 0; j < k; j++, nt++)
                if (nt < WMI_NODE_MAX_TOKENS) r.tokens[nt] = wmi_batch_token_data(s, u, g, j).id;
        }
        r.n_tokens = nt;
    }
    return 0;
}

}  // namespace

extern "C" {

struct whisper_context *wmi_ctx_for_device(const char *path_model, int device) {
    whisper_context_params p = whisper_context_default_params();
    p.gpu_device = device;
    return whisper_init_from_file_with_params_no_state(path_model, p);
}

wmi_node *wmi_node_init(const char *path_model, const int *devices, int n_devices, int batch, int max_decoders, int enc_chunk) {
    try {
        int ndev = 0;
        if (!path_model || batch < 1 || n_devices < 0 || n_devices > 64) return nullptr;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { log_msg(2, "wmi_node_init: no HIP device\n"); return nullptr; }
        int dev_before = 0;            // the calling thread's current device is put back before returning
        (void)hipGetDevice(&dev_before);
        wmi_node *node = new wmi_node();
        node->batch = batch;
        if (n_devices == 0 || !devices) for (int d = 0; d < ndev; d++) node->devices.push_back(d);     // every visible GPU
        else node->devices.assign(devices, devices + n_devices);
        const int n = (int)node->devices.size();
        node->ctx.assign(n, nullptr);
        node->state.assign(n, nullptr);
        // the engines load side by side: each thread reads the file and uploads to its own GPU
        std::vector<std::thread> th;
        auto load = [&](int w) {
            try {
                node->ctx[w] = wmi_ctx_for_device(path_model, node->devices[w]);
                if (node->ctx[w]) node->state[w] = wmi_init_state_batch(node->ctx[w], batch, max_decoders, enc_chunk);
            } catch (...) {}           // the engine stays unset: reported below
        };
        int started = 1;
        try {
            for (; started < n; started++) th.emplace_back(load, started);
        } catch (...) {}               // no more threads to be had: the remaining engines load on this one
        load(0);
        for (auto &t : th) t.join();
        for (int w = started; w < n; w++) load(w);
        (void)hipSetDevice(dev_before);
        for (int w = 0; w < n; w++)
            if (!node->state[w]) { log_msg(2, "wmi_node_init: engine %d (device %d) failed to start\n", w, node->devices[w]); wmi_node_free(node); return nullptr; }
        return node;
    } catch (...) { return nullptr; }
}

void wmi_node_free(wmi_node *node) {
    if (!node) return;
    int dev_before = 0;
    const bool have_dev = hipGetDevice(&dev_before) == hipSuccess;
    for (size_t w = 0; w < node->ctx.size(); w++) {
        (void)hipSetDevice(node->devices[w]);
        whisper_free_state(node->state[w]);
        whisper_free(node->ctx[w]);
    }
    if (have_dev) (void)hipSetDevice(dev_before);
    delete node;
}

int wmi_node_n_devices(wmi_node *node) { return node ? (int)node->devices.size() : 0; }
struct whisper_state *wmi_node_state(wmi_node *node, int i) { return node && i >= 0 && i < (int)node->state.size() ? node->state[i] : nullptr; }
struct whisper_context *wmi_node_context(wmi_node *node, int i) { return node && i >= 0 && i < (int)node->ctx.size() ? node->ctx[i] : nullptr; }
int wmi_node_last_gather(wmi_node *node) { return node ? node->last_gather : 0; }
int wmi_node_last_batches(wmi_node *node, int i) { return node && i >= 0 && i < (int)node->last_batches.size() ? node->last_batches[i] : 0; }

int wmi_node_run_custom(int n_workers, int n_utts, int batch, wmi_node_batch_fn fn, void *user, wmi_node_record *records_out, int *batches_per_worker) {
    try { return run_queue(n_workers, n_utts, batch, fn, user, records_out, batches_per_worker); } catch (...) { return -1; }
}

int wmi_node_run(wmi_node *node, struct whisper_full_params params, const float *const *samples, const int *n_samples, int n_utts, wmi_node_record *records_out) {
    try {
        if (!node || n_utts < 0 || (n_utts > 0 && (!samples || !n_samples || !records_out))) return -1;
        const int n = (int)node->devices.size();
        EngineJob job{node, &params, samples, n_samples};
        node->last_batches.assign(n, 0);
        node->last_gather = 0;
        // worker 0 is the calling thread: its current device is put back when the job is over
        struct DeviceGuard { int d = 0; bool ok; DeviceGuard() : ok(hipGetDevice(&d) == hipSuccess) {} ~DeviceGuard() { if (ok) (void)hipSetDevice(d); } } guard;
        const int rc = run_queue(n, n_utts, node->batch, engine_batch, &job, records_out, node->last_batches.data());
        if (rc != 0) return rc;
        // the gather: with more than one distinct GPU the records of device w travel to device 0 over RCCL (xGMI); they are
        // also in this process's memory already, which is what stands when RCCL is absent or refuses the device list
        bool distinct = n > 1;
        for (int a = 0; a < n; a++) for (int b = a + 1; b < n; b++) if (node->devices[a] == node->devices[b]) distinct = false;
        const char *mode = getenv("WMI_NODE_GATHER");
        const bool force = mode && !strcmp(mode, "rccl") && distinct_or_single(node->devices);      // also with one GPU: a one-rank communicator (self-test)
        if ((force || (distinct && !(mode && !strcmp(mode, "host")))) && n_utts > 0) {
            std::vector<std::vector<int>> rows(n);
            for (int u = 0; u < n_utts; u++) {
                const int w = records_out[u].device;
                if (w >= 0 && w < n) rows[w].push_back(u);
            }
            std::vector<wmi_node_record> tmp(records_out, records_out + n_utts);
            if (gather_rccl(node->devices, rows, tmp.data(), records_out, n_utts)) node->last_gather = 1;
        }
        return 0;
    } catch (...) { return -1; }
}

}  // extern "C"
