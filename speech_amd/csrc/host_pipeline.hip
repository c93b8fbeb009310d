// host_pipeline.hip -- a ragged batch from host memory to host memory (speech_plan_run_host, include/speech_hip.h).
//
// The reference's callers hold their audio in host memory and get their features back there (src/kaldi_mfcc.cpp:286-308);
// one utterance per call leaves the GPU idle while PCIe moves 64 KB per audio-second.  Here the utterances are cut into
// chunks of bounded size; each chunk is packed into a pinned staging buffer by a few host threads, copied to the device,
// run as one batch (speech_batch_run / _pcm16) and copied back, on one of two streams, so that the copies and the host-side
// packing of one chunk overlap the kernel and the copies of the other.  Device and pinned buffers belong to the calling
// thread, only grow, and are reused by later calls (speech_host_release frees them).
#include "speech_internal.hpp"
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace speech {
namespace {

struct CopyJob { void *dst; const void *src; size_t bytes; };

// Host-side gather / scatter of a chunk: pieces of at most 1 MiB handed out to up to 8 threads.
void parallel_copy(const std::vector<CopyJob> &jobs)
{
    size_t total = 0;
    for (const CopyJob &j : jobs) { total += j.bytes; }
    constexpr size_t PIECE = (size_t)1 << 20;
    static const unsigned max_threads = [] { const char *e = getenv("SPEECH_HOST_THREADS"); const int v = e ? atoi(e) : 8; return (unsigned)std::min(64, std::max(1, v)); }();
    unsigned nth = std::min(max_threads, std::max(1u, std::thread::hardware_concurrency()));
    if (total < 4 * PIECE) { nth = 1; }
    if (nth == 1)
    {
        for (const CopyJob &j : jobs) { if (j.bytes) { memcpy(j.dst, j.src, j.bytes); } }
        return;
    }
    std::vector<CopyJob> pieces;
    for (const CopyJob &j : jobs)
    {
        for (size_t o = 0; o < j.bytes; o += PIECE) { pieces.push_back(CopyJob{(char *)j.dst + o, (const char *)j.src + o, std::min(PIECE, j.bytes - o)}); }
    }
    std::atomic<size_t> next(0);
    auto work = [&]() { for (size_t i = next.fetch_add(1); i < pieces.size(); i = next.fetch_add(1)) { memcpy(pieces[i].dst, pieces[i].src, pieces[i].bytes); } };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nth; ++t) { th.emplace_back(work); }
    work();
    for (std::thread &t : th) { t.join(); }
}

struct Slot {
    void *hX = nullptr, *hY = nullptr, *dX = nullptr, *dY = nullptr;
    size_t cap_hX = 0, cap_hY = 0, cap_dX = 0, cap_dY = 0;
    hipStream_t st = nullptr;
    hipEvent_t done = nullptr;
    // the chunk in flight
    speech_batch *batch = nullptr;
    size_t first = 0, count = 0;
    std::vector<size_t> y_off, frames;
};

struct HostPipe {
    int device = -1;
    Slot s[2];
    void release()
    {
        for (Slot &k : s)
        {
            if (k.batch) { speech_batch_destroy(k.batch); }
            if (k.hX) { (void)hipHostFree(k.hX); }
            if (k.hY) { (void)hipHostFree(k.hY); }
            if (k.dX) { (void)hipFree(k.dX); }
            if (k.dY) { (void)hipFree(k.dY); }
            if (k.done) { (void)hipEventDestroy(k.done); }
            if (k.st) { (void)hipStreamDestroy(k.st); }
            k = Slot();
        }
        device = -1;
    }
    ~HostPipe() { release(); }                                              // a worker thread that exits gives its staging back
};

thread_local HostPipe g_pipe;

hipError_t grow(void **p, size_t *cap, size_t bytes, bool pinned)
{
    if (bytes <= *cap) { return hipSuccess; }
    if (*p) { (void)(pinned ? hipHostFree(*p) : hipFree(*p)); *p = nullptr; *cap = 0; }
    const size_t want = bytes + bytes / 8;
    const hipError_t e = pinned ? hipHostMalloc(p, want, hipHostMallocDefault) : hipMalloc(p, want);
    if (e == hipSuccess) { *cap = want; } else { *p = nullptr; }
    return e;
}

#define PIPE_TRY(expr)                                                                                       \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) { return fail(fn, "HIP call failed: %s (%s)", #expr, hipGetErrorString(e_)); } \
    } while (0)

// wait for the slot's chunk and hand its features to the caller's arrays
int drain(const char *fn, Slot &k, void *const *Y, size_t es, size_t D)
{
    if (!k.batch) { return 0; }
    const hipError_t e = hipEventSynchronize(k.done);
    std::vector<CopyJob> jobs;
    if (e == hipSuccess)
    {
        for (size_t u = 0; u < k.count; ++u) { jobs.push_back(CopyJob{Y[k.first + u], (const char *)k.hY + k.y_off[u] * es, k.frames[u] * D * es}); }
        parallel_copy(jobs);
    }
    speech_batch_destroy(k.batch);
    k.batch = nullptr;
    if (e != hipSuccess) { return fail(fn, "kernel execution or copy failed: %s", hipGetErrorString(e)); }
    return 0;
}

int run_host(speech_plan *plan, size_t nutt, const void *const *X, const size_t *n_samples, void *const *Y, int pcm16, size_t chunk_bytes, const unsigned int *ids)
{
    const char *fn = "speech_plan_run_host";
    int dev = 0, dtype = 0; size_t D = 0;
    plan_info(plan, &dev, &dtype, &D);
    const size_t es = dtype == SPEECH_F64 ? 8u : 4u, xs = pcm16 ? 2u : es;
    if (chunk_bytes == 0) { chunk_bytes = (size_t)64 << 20; }
    HostPipe &pipe = g_pipe;
    if (pipe.device != dev) { pipe.release(); pipe.device = dev; }
    for (Slot &k : pipe.s)
    {
        if (!k.st) { PIPE_TRY(hipStreamCreateWithFlags(&k.st, hipStreamNonBlocking)); }
        if (!k.done) { PIPE_TRY(hipEventCreateWithFlags(&k.done, hipEventDisableTiming)); }
    }
    size_t first = 0;
    for (unsigned c = 0; first < nutt; ++c)
    {
        Slot &k = pipe.s[c & 1u];
        if (drain(fn, k, Y, es, D)) { return 1; }
        // the chunk: utterances in order until the staged input reaches chunk_bytes (at least one)
        size_t count = 0, nin = 0;
        while (first + count < nutt && (count == 0 || (nin + n_samples[first + count]) * xs <= chunk_bytes) && count < 0x7FFFFFFFu)
        {
            nin += n_samples[first + count];
            ++count;
        }
        if (first + count > 0xFFFFFFFFull) { return fail(fn, "too many utterances"); }
        if (batch_create(&k.batch, plan, count, n_samples + first, nullptr, nullptr, (uint32_t)first, ids ? ids + first : nullptr)) { return 1; }
        k.first = first; k.count = count;
        k.y_off.resize(count); k.frames.resize(count);
        std::vector<size_t> x_off(count);
        size_t ns = 0, no = 0, nf = 0;
        speech_batch_totals(k.batch, &ns, &no, &nf);
        speech_batch_offsets(k.batch, x_off.data(), k.y_off.data(), k.frames.data());
        PIPE_TRY(grow(&k.hX, &k.cap_hX, ns * xs, true));
        PIPE_TRY(grow(&k.hY, &k.cap_hY, no * es, true));
        PIPE_TRY(grow(&k.dX, &k.cap_dX, ns * xs, false));
        PIPE_TRY(grow(&k.dY, &k.cap_dY, no * es, false));
        std::vector<CopyJob> jobs(count);
        for (size_t u = 0; u < count; ++u) { jobs[u] = CopyJob{(char *)k.hX + x_off[u] * xs, X[first + u], n_samples[first + u] * xs}; }
        parallel_copy(jobs);
        if (ns) { PIPE_TRY(hipMemcpyAsync(k.dX, k.hX, ns * xs, hipMemcpyHostToDevice, k.st)); }
        if (pcm16 ? speech_batch_run_pcm16(k.batch, (const short *)k.dX, k.dY, (void *)k.st, nullptr) : speech_batch_run(k.batch, k.dX, k.dY, (void *)k.st)) { return 1; }
        if (no) { PIPE_TRY(hipMemcpyAsync(k.hY, k.dY, no * es, hipMemcpyDeviceToHost, k.st)); }
        PIPE_TRY(hipEventRecord(k.done, k.st));
        first += count;
    }
    // the two chunks still in flight, older first
    Slot *order[2] = {&pipe.s[0], &pipe.s[1]};
    if (pipe.s[0].batch && pipe.s[1].batch && pipe.s[1].first < pipe.s[0].first) { std::swap(order[0], order[1]); }
    int rc = 0;
    for (Slot *k : order) { rc |= drain(fn, *k, Y, es, D); }
    return rc;
}

}  // namespace
}  // namespace speech

extern "C" {

int speech_plan_run_host(speech_plan *plan, size_t nutt, const void *const *X, const size_t *n_samples, void *const *Y, int pcm16, size_t chunk_bytes,
                         const unsigned int *ids)
{
    using namespace speech;
    const char *fn = "speech_plan_run_host";
    if (!plan) { return fail(fn, "null plan"); }
    if (nutt == 0) { return 0; }
    if (!X || !n_samples || !Y) { return fail(fn, "null argument"); }
    for (size_t u = 0; u < nutt; ++u) { if (!X[u] || !Y[u]) { return fail(fn, "null pointer for utterance %zu", u); } }
    int dev = 0, prev = -1;
    plan_info(plan, &dev, nullptr, nullptr);
    if (hipGetDevice(&prev) != hipSuccess) { prev = -1; }
    if (hipSetDevice(dev) != hipSuccess) { return fail(fn, "cannot select device %d", dev); }
    int rc = run_host(plan, nutt, X, n_samples, Y, pcm16, chunk_bytes, ids);
    if (rc)
    {
        // leave nothing in flight: wait for both streams and drop the chunks' batches
        for (Slot &k : g_pipe.s)
        {
            if (k.st) { (void)hipStreamSynchronize(k.st); }
            if (k.batch) { speech_batch_destroy(k.batch); k.batch = nullptr; }
        }
    }
    if (prev >= 0) { (void)hipSetDevice(prev); }
    return rc;
}

void speech_host_release(void) { speech::g_pipe.release(); }

}  // extern "C"
