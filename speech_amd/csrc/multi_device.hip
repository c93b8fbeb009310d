// multi_device.hip -- a ragged list of utterances from host memory over several GPUs of one node, inside the library
// (speech_run_host_multi, speech_shard_utterances, speech_opts_frames; include/speech_hip.h).
//
// The reference's caller is a C program that holds its audio in host memory and calls one function per utterance
// (src/kaldi_mfcc.cpp:283-308).  Utterances are independent, so the list is dealt to the devices by
// longest-processing-time greedy on the sample counts and every device gets its own host thread, its own plan (tables on that
// device) and its own two-stream pinned staging pipeline (host_pipeline.hip, thread-local).  There is no collective and no
// peer copy; the only cross-device step is the join of the host threads.  Dither is keyed by the utterance's number in the
// caller's list, so the features do not depend on how many devices shared the work.
//
// A device that fails (cannot be selected, plan or staging allocation fails, a HIP call returns an error) does not fail the
// call while another device works: its utterances are dealt again to the devices that finished their own share, once.
#include "speech_internal.hpp"
#include <algorithm>
#include <cfloat>
#include <numeric>
#include <string>
#include <thread>

namespace speech {
namespace {

// Longest first (ties: lower index first), each to the least loaded shard (ties: lowest shard number): the assignment of
// speech_amd.api.shard_utterances, statement for statement, so that a Python rank and a C caller cut a corpus alike.
void lpt_assign(size_t nutt, const size_t *n_samples, const size_t *subset, int nshards, int *shard)
{
    std::vector<size_t> order(nutt);
    if (subset) { std::copy(subset, subset + nutt, order.begin()); } else { std::iota(order.begin(), order.end(), (size_t)0); }
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return n_samples[a] > n_samples[b]; });
    std::vector<unsigned long long> load((size_t)nshards, 0ull);
    for (size_t u : order)
    {
        int r = 0;
        for (int s = 1; s < nshards; ++s) { if (load[(size_t)s] < load[(size_t)r]) { r = s; } }
        shard[u] = r;
        load[(size_t)r] += (unsigned long long)n_samples[u];
    }
}

struct ShardJob {
    int device = 0;
    std::vector<size_t> utts;            // indices into the caller's list, ascending
    int rc = 0;
    std::string err;
};

// one device's share, on its own thread: plan on that device, the two-stream host pipeline, everything released at the end
void run_shard(ShardJob *job, const speech_opts *opts, const void *const *X, const size_t *n_samples, void *const *Y, int pcm16, size_t chunk_bytes,
               const unsigned int *ids)
{
    job->rc = 0; job->err.clear();
    const size_t n = job->utts.size();
    if (n == 0) { return; }
    std::vector<const void *> xs(n); std::vector<void *> ys(n); std::vector<size_t> ns(n); std::vector<unsigned int> id(n);
    for (size_t i = 0; i < n; ++i)
    {
        const size_t u = job->utts[i];
        xs[i] = X[u]; ys[i] = Y[u]; ns[i] = n_samples[u]; id[i] = ids ? ids[u] : (unsigned int)u;
    }
    speech_plan *plan = nullptr;
    if (speech_plan_create(&plan, opts, job->device)) { job->rc = 1; job->err = speech_last_error(); return; }
    if (speech_plan_run_host(plan, n, xs.data(), ns.data(), ys.data(), pcm16, chunk_bytes, id.data())) { job->rc = 1; job->err = speech_last_error(); }
    speech_host_release();               // this thread's staging: freed while its device context is at hand
    speech_plan_destroy(plan);
}

void run_jobs(std::vector<ShardJob> &jobs, const speech_opts *opts, const void *const *X, const size_t *n_samples, void *const *Y, int pcm16,
              size_t chunk_bytes, const unsigned int *ids)
{
    std::vector<std::thread> th;
    for (size_t j = 1; j < jobs.size(); ++j) { th.emplace_back(run_shard, &jobs[j], opts, X, n_samples, Y, pcm16, chunk_bytes, ids); }
    run_shard(&jobs[0], opts, X, n_samples, Y, pcm16, chunk_bytes, ids);       // the caller's thread takes the first share
    for (std::thread &t : th) { t.join(); }
}

}  // namespace
}  // namespace speech

extern "C" {

int speech_shard_utterances(size_t nutt, const size_t *n_samples, int nshards, int *shard)
{
    using namespace speech;
    const char *fn = "speech_shard_utterances";
    if (nshards < 1) { return fail(fn, "nshards must be positive"); }
    if (nutt && (!n_samples || !shard)) { return fail(fn, "null argument"); }
    lpt_assign(nutt, n_samples, nullptr, nshards, shard);
    return 0;
}

int speech_run_host_multi(const speech_opts *opts, int ndev, const int *devices, size_t nutt, const void *const *X, const size_t *n_samples,
                          void *const *Y, int pcm16, size_t chunk_bytes, const unsigned int *ids)
{
    using namespace speech;
    const char *fn = "speech_run_host_multi";
    if (!opts) { return fail(fn, "null options"); }
    if (ndev < 1 || !devices) { return fail(fn, "at least one device is needed"); }
    if (ndev > 64) { return fail(fn, "more than 64 devices"); }
    if (nutt == 0) { return 0; }
    if (!X || !n_samples || !Y) { return fail(fn, "null argument"); }
    if (nutt > 0xFFFFFFFFull) { return fail(fn, "too many utterances"); }
    for (size_t u = 0; u < nutt; ++u) { if (!X[u] || !Y[u]) { return fail(fn, "null pointer for utterance %zu", u); } }
    // what no device could do anything about is found here, before any thread starts: the option set and every utterance's length
    for (size_t u = 0; u < nutt; ++u) { size_t W = 0; if (speech_opts_frames(opts, n_samples[u], &W, nullptr)) { return 1; } }

    std::vector<int> shard(nutt);
    lpt_assign(nutt, n_samples, nullptr, ndev, shard.data());
    std::vector<ShardJob> jobs((size_t)ndev);
    for (int d = 0; d < ndev; ++d) { jobs[(size_t)d].device = devices[d]; }
    for (size_t u = 0; u < nutt; ++u) { jobs[(size_t)shard[u]].utts.push_back(u); }
    run_jobs(jobs, opts, X, n_samples, Y, pcm16, chunk_bytes, ids);

    std::vector<size_t> again;
    std::vector<int> alive;
    std::string first_err;
    for (const ShardJob &j : jobs)
    {
        if (j.rc) { again.insert(again.end(), j.utts.begin(), j.utts.end()); if (first_err.empty()) { first_err = j.err; } }
        else { alive.push_back(j.device); }
    }
    if (again.empty()) { return 0; }
    if (alive.empty()) { return fail(fn, "every device failed; first failure: %s", first_err.c_str()); }
    // the failed devices' utterances, dealt to the devices that finished their own share (no utterance is lost or done twice
    // into different memory: a failed shard's outputs are simply written again)
    std::sort(again.begin(), again.end());
    fprintf(stderr, "speech_run_host_multi: %zu device(s) failed (%s); their %zu utterances go to the %zu working device(s)\n",
            jobs.size() - alive.size(), first_err.c_str(), again.size(), alive.size());
    std::vector<int> shard2(nutt, -1);
    lpt_assign(again.size(), n_samples, again.data(), (int)alive.size(), shard2.data());
    std::vector<ShardJob> jobs2(alive.size());
    for (size_t d = 0; d < alive.size(); ++d) { jobs2[d].device = alive[d]; }
    for (size_t u : again) { jobs2[(size_t)shard2[u]].utts.push_back(u); }
    run_jobs(jobs2, opts, X, n_samples, Y, pcm16, chunk_bytes, ids);
    for (const ShardJob &j : jobs2) { if (j.rc) { return fail(fn, "device %d failed on re-queued utterances: %s", j.device, j.err.c_str()); } }
    return 0;
}

}  // extern "C"
